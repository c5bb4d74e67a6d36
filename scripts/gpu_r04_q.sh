#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests/test_bench_shapes_gpu.py tests/test_ivf_pq_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
timeout 600 python scripts/pq_scan_variants.py --steps 10 "F4=1,LUT=f16,ACC=f32" "F4=1,LUT=f16,ACC=f16" "F4=1,LUT=f32" 2>&1 | grep -v "^\[bench\]" | grep "search" | tail -4
