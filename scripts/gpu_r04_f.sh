#!/bin/bash
# round 4: chunked survivor buffer (tests, L2 + inner-product timing), then PMC passes over the C4 / C2 search kernels
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -X faulthandler -m pytest tests/test_ivf_pq_gpu.py tests/test_bench_shapes_gpu.py tests/test_fuzz_gpu.py tests/test_list_shard_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -x > gpurun_out/r04f_tests.log 2>&1
echo "tests rc=$?"; grep "passed\|failed\|rror" gpurun_out/r04f_tests.log | tail -8
timeout 600 python scripts/pq_scan_variants.py --steps 5 "F4=0,LUT=f16,ACC=f32" "F4=1,LUT=f16,ACC=f32" > gpurun_out/r04f_l2.log 2>&1
grep -v "^\[bench\]" gpurun_out/r04f_l2.log | tail -3
timeout 600 python scripts/pq_scan_variants.py --steps 5 --metric inner_product "F4=0,LUT=f16,ACC=f32" "F4=1,LUT=f16,ACC=f32" > gpurun_out/r04f_ip.log 2>&1
grep -v "^\[bench\]" gpurun_out/r04f_ip.log | tail -3
timeout 900 python scripts/pmc_c4_c2.py build > gpurun_out/r04f_pmc_build.log 2>&1; tail -2 gpurun_out/r04f_pmc_build.log
W=/tmp/pmc_c4c2; rm -rf $W; mkdir -p $W
i=1
for P in "FETCH_SIZE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES"; do
  (cd /tmp && timeout 300 rocprofv3 --pmc $P --output-format csv -d $W/p$i -o p$i -- python $GRAFT_REPO_ROOT/scripts/pmc_c4_c2.py run > $GRAFT_REPO_ROOT/gpurun_out/r04f_pmc_$i.log 2>&1)
  echo "pmc pass $i rc=$?"; i=$((i+1))
done
python scripts/pmc_c4_c2.py summarize $W gpurun_out/r04f_pmc_c4_c2.json | head -60
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $W/kt -o kt -- python $GRAFT_REPO_ROOT/scripts/pmc_c4_c2.py run > /dev/null 2>&1)
find $W/kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04f_c4_c2_kernel_stats.csv \;
head -12 gpurun_out/r04f_c4_c2_kernel_stats.csv | cut -c1-170
