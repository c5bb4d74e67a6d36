#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 420 python -X faulthandler -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/r03o_tests.log 2>&1
echo "tests rc=$?"; grep "passed\|failed\|rror\|^E " gpurun_out/r03o_tests.log | tail -12
