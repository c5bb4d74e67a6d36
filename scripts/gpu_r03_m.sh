#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -X faulthandler -m pytest tests/test_bench_shapes_gpu.py -q -x --timeout 600 -p no:cacheprovider -k "c3 or c2 or c5 or matrix_core" > gpurun_out/r03m_tests.log 2>&1
echo "tests rc=$?"; grep "passed\|failed\|rror\|^E " gpurun_out/r03m_tests.log | tail -6
