#!/bin/bash
# round 3, GPU call 1: LDS gather ceilings, PQ parity tests after the pq_scan2 rewrite, scan variants at the bench workload
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
scripts/bin/lds_gather_bench > gpurun_out/r03_lds_gather_bench.json 2> gpurun_out/r03_lds_gather_bench.err; echo "lds bench rc=$?"
cat gpurun_out/r03_lds_gather_bench.json
timeout 1200 python -m pytest tests/test_bench_shapes_gpu.py tests/test_ivf_pq_gpu.py tests/test_list_shard_gpu.py tests/test_fuzz_gpu.py -q \
  -k "not cagra_walk and not select_k and not brute and not flat and not refine" -p no:cacheprovider > gpurun_out/r03a_tests.log 2>&1
echo "tests rc=$?"; tail -15 gpurun_out/r03a_tests.log
timeout 900 python scripts/pq_scan_variants.py "LUT=f16,ACC=f32" "LUT=f16,ACC=f32,S2=0" "LUT=f16,ACC=f16" "LUT=f16,ACC=f16,S2=0" \
  "LUT=f32" "LUT=f32,S2=0" "LUT=u8,ACC=f16" "LUT=u8,ACC=f16,S2=0" "LUT=f16,ACC=f32,DBG=128" "LUT=f16,ACC=f32,QCAP=256" > gpurun_out/r03a_variants.log 2>&1
echo "variants rc=$?"; cat gpurun_out/r03a_variants.log | grep -v "^\[bench\]"
