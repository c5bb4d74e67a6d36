#!/bin/bash
# rocprofv3 --pmc passes (counters only, one group per run) + a kernel trace over 4 searches of the C2 index -> profiles/r06_pmc_c2.json
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp PMC_ONLY_C2=1
python scripts/pmc_c4_c2.py build 2>&1 | grep -v amdgpu.ids | tail -2
OUT=gpurun_out/r06_pmc_c2; rm -rf $OUT; mkdir -p $OUT
i=0
for SET in "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "FETCH_SIZE SQ_VALU_MFMA_BUSY_CYCLES" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $SET -d $GRAFT_REPO_ROOT/$OUT/pass$i -o p --output-format csv -- python $GRAFT_REPO_ROOT/scripts/pmc_c4_c2.py run > $GRAFT_REPO_ROOT/$OUT/pass$i.log 2>&1)
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/trace -o t --output-format csv -- python $GRAFT_REPO_ROOT/scripts/pmc_c4_c2.py run > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1)
python scripts/pmc_c4_c2.py summarize $OUT gpurun_out/r06_pmc_c2.json | tail -60
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r06_c2_kernel_stats.csv
find $OUT -name "*.csv" -size +2000k -delete; find $OUT -name "*.db" -delete
head -12 gpurun_out/r06_c2_kernel_stats.csv
