#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests/test_bench_shapes_gpu.py tests/test_ivf_pq_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -15
timeout 900 python scripts/pq_len_timing.py 2>&1 | grep -v amdgpu.ids | tail -8 | tee gpurun_out/r04n_pq_len.log
