#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 24 python -u -m pytest tests -m gpu -n 6 --dist loadfile -v --timeout 60 -p no:cacheprovider > gpurun_out/r03t_tests.log 2>&1
echo "tests rc=$?"; grep -c PASSED gpurun_out/r03t_tests.log; grep "FAILED\|ERROR\|passed\|failed" gpurun_out/r03t_tests.log | tail -8
