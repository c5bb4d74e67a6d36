#!/bin/bash
# round 3: quick loop on the matrix-core tail phase - PQ parity tests + variants
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests/test_bench_shapes_gpu.py tests/test_ivf_pq_gpu.py tests/test_list_shard_gpu.py tests/test_serialize_filter_gpu.py -q -x --timeout 600 -p no:cacheprovider -k "not c4" > gpurun_out/r03e_tests.log 2>&1
echo "tests rc=$?"; grep -v "^  File\|^Extension" gpurun_out/r03e_tests.log | tail -12
timeout 900 python scripts/pq_scan_variants.py ${VARIANTS:-"LUT=f16,ACC=f32" "LUT=f16,ACC=f16" "LUT=f32" "LUT=u8,ACC=f16"} > gpurun_out/r03e_variants.log 2>&1
echo "variants rc=$?"; grep -v "^\[bench\]" gpurun_out/r03e_variants.log | tail -20
