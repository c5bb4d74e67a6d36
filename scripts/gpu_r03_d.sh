#!/bin/bash
# round 3: the matrix-core filter (ivf_pq_scan3.hip) - parity tests, CAGRA optimize tests, scan variants at the bench workload
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1500 python -X faulthandler -m pytest tests/test_bench_shapes_gpu.py tests/test_ivf_pq_gpu.py tests/test_cagra_optimize_gpu.py tests/test_fuzz_gpu.py -q -x --timeout 600 -p no:cacheprovider \
  -k "not c4_shape" > gpurun_out/r03d_tests.log 2>&1
echo "tests rc=$?"; grep -v "^  File\|^Extension" gpurun_out/r03d_tests.log | tail -30
timeout 900 python scripts/pq_scan_variants.py ${VARIANTS:-"LUT=f16,ACC=f32" "LUT=f16,ACC=f32,S3=0" "LUT=f16,ACC=f32,DBG=1024" "LUT=f16,ACC=f16" "LUT=f32" "LUT=u8,ACC=f16"} > gpurun_out/r03d_variants.log 2>&1
echo "variants rc=$?"; grep -v "^\[bench\]" gpurun_out/r03d_variants.log | tail -20
