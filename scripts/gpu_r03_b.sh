#!/bin/bash
# round 3: quick loop - PQ parity tests + scan variants at the bench workload
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bench_shapes_gpu.py tests/test_ivf_pq_gpu.py -q -x -k "c3 or c5 or two_phase or reduced_precision or parity_with_oracle" -p no:cacheprovider > gpurun_out/r03b_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r03b_tests.log
timeout 900 python scripts/pq_scan_variants.py ${VARIANTS:-"LUT=f16,ACC=f32" "LUT=f16,ACC=f16" "LUT=f32" "LUT=u8,ACC=f16" "LUT=f16,ACC=f32,DBG=128"} > gpurun_out/r03b_variants.log 2>&1
echo "variants rc=$?"; grep -v "^\[bench\]" gpurun_out/r03b_variants.log | grep -v "pq_scan stats\|pq_scan2 waits" ; grep "pq_scan stats\|pq_scan2 waits" gpurun_out/r03b_variants.log | tail -2
