#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
T=${1:-r05s}
timeout 1500 python -m pytest tests/test_reference_tables_gpu.py -k "brute_force_reference or refine_reference or nn_descent_reference" -q --timeout 600 -p no:cacheprovider --durations=8 > gpurun_out/${T}_tests.log 2>&1
echo "rc=$?"; grep -E "passed|failed" gpurun_out/${T}_tests.log | tail -3; grep -E "^FAILED|^ERROR" gpurun_out/${T}_tests.log | cut -c1-200 | head -40; grep -A9 "slowest" gpurun_out/${T}_tests.log | head -12
timeout 300 python -m pytest tests/test_coarse_grouped_gpu.py tests/test_scratch_cache_gpu.py -q --timeout 600 -p no:cacheprovider 2>&1 | tail -3
