#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests/test_bench_shapes_gpu.py -m gpu -q -p no:cacheprovider -x -k "every_pq_len or c3_shape" 2>&1 | tail -15
timeout 600 python scripts/pq_len_timing.py 2>&1 | grep -v amdgpu.ids | tail -5 | tee gpurun_out/r04l_pq_len.log
