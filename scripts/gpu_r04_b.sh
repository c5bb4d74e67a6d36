#!/bin/bash
# round 4: filter4 v2 (4-phase ring, term loads two subtiles ahead, unit descriptor prefetch): parity tests, then A/B + ablations
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -X faulthandler -m pytest tests/test_ivf_pq_gpu.py tests/test_bench_shapes_gpu.py tests/test_fuzz_gpu.py tests/test_ivf_flat_gpu.py tests/test_list_shard_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -x > gpurun_out/r04b_tests.log 2>&1
echo "tests rc=$?"; grep "passed\|failed\|rror" gpurun_out/r04b_tests.log | tail -8
timeout 900 python scripts/pq_scan_variants.py --steps 5 "F4=0,LUT=f16,ACC=f32" "F4=1,LUT=f16,ACC=f32" \
  "F4=1,DBG=1024,LUT=f16,ACC=f32" \
  "F4=1,DBG=65536,LUT=f16,ACC=f32" "F4=1,DBG=131072,LUT=f16,ACC=f32" "F4=1,DBG=262144,LUT=f16,ACC=f32" "F4=1,DBG=524288,LUT=f16,ACC=f32" \
  "F4=1,DBG=1048576,LUT=f16,ACC=f32" "F4=1,DBG=2097152,LUT=f16,ACC=f32" > gpurun_out/r04b_variants.log 2>&1
echo "variants rc=$?"; grep -v "^\[bench\]" gpurun_out/r04b_variants.log | grep -v "overflow entries" | tail -30
