#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests/test_bench_shapes_gpu.py tests/test_ivf_flat_gpu.py -q -x --timeout 600 -p no:cacheprovider -k "c2_shape or flat" > gpurun_out/r03k_tests.log 2>&1
echo "tests rc=$?"; grep "passed\|failed\|rror\|^E " gpurun_out/r03k_tests.log | tail -8
timeout 600 python scripts/bench_other.py flat > gpurun_out/r03k_flat.log 2>&1; echo "flat rc=$?"; grep -v "^\[bench\]" gpurun_out/r03k_flat.log | tail -5
