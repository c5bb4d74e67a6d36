#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/r03n_tests.log 2>&1
echo "tests rc=$?"; grep "passed\|failed\|rror" gpurun_out/r03n_tests.log | tail -5
timeout 300 python scripts/pq_scan_variants.py "LUT=f16,ACC=f32" 2>&1 | grep -v "^\[bench\]" | tail -1
timeout 300 python scripts/pq_scan_variants.py --k 100 "LUT=f16,ACC=f32" "LUT=f16,ACC=f32,S3=0" 2>&1 | grep -v "^\[bench\]" | tail -2
