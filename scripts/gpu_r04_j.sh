#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -X faulthandler -m pytest tests/test_select_k_gpu.py tests/test_brute_force_gpu.py tests/test_ivf_pq_gpu.py tests/test_fuzz_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -2
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_j -o kt -- python $GRAFT_REPO_ROOT/scripts/pq_scan_variants.py --steps 10 "F4=1,LUT=f16,ACC=f32" > $GRAFT_REPO_ROOT/gpurun_out/r04j_kt.log 2>&1)
find /tmp/prof_j -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04j_kernel_stats.csv \;
grep "select_k\|dist_tile_kernel<float, float, 0\|pq_filter4\|pq_head" gpurun_out/r04j_kernel_stats.csv | cut -d, -f1-4 | cut -c1-200
grep -v "^\[bench\]" gpurun_out/r04j_kt.log | tail -2
