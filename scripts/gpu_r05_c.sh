#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
T=${1:-r05c}
timeout 1500 python -m pytest tests/test_reference_tables_gpu.py "tests/test_brute_force_gpu.py::test_bit_exact_at_the_baseline_config0_shape" -q --timeout 600 -p no:cacheprovider --durations=25 -x --maxfail=60 > gpurun_out/${T}_tables.log 2>&1
echo "rc=$?"; grep -E "passed|failed" gpurun_out/${T}_tables.log | tail -3; grep -E "^FAILED|^ERROR" gpurun_out/${T}_tables.log | cut -c1-260 | head -70
grep -A30 "slowest" gpurun_out/${T}_tables.log | head -32
