#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
T=${1:-r05q}
timeout 900 python -m pytest tests/test_coarse_grouped_gpu.py tests/test_bench_shapes_gpu.py tests/test_ivf_pq_gpu.py tests/test_ivf_flat_gpu.py tests/test_fuzz_gpu.py -q -x --timeout 600 -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1
echo "rc=$?"; grep -E "passed|failed" gpurun_out/${T}_tests.log | tail -3; grep -E "^FAILED|^ERROR" gpurun_out/${T}_tests.log | cut -c1-260 | head
timeout 900 python scripts/head_rows_sweep.py 100000000 0 > gpurun_out/${T}_sweep.log 2>&1
grep -v amdgpu.ids gpurun_out/${T}_sweep.log | tail -4
