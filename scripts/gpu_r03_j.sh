#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -X faulthandler -m pytest tests -m gpu -q -x --timeout 600 -p no:cacheprovider -k "not c4_shape and not nn_descent and not cagra" > gpurun_out/r03j_tests.log 2>&1
echo "tests rc=$?"; grep "passed\|failed\|rror" gpurun_out/r03j_tests.log | tail -5
VARIANTS="LUT=f16,ACC=f32" bash scripts/gpu_r03_f.sh | tail -2
