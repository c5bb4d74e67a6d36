#!/bin/bash
# rocprofv3 --pmc passes (counters only, one group per run) over 4 searches of the PQ-768 shape on the wide path -> profiles/r06_pmc_wide.json
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
OUT=gpurun_out/r06_pmc_wide; rm -rf $OUT; mkdir -p $OUT
i=0
for SET in "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "FETCH_SIZE SQ_VALU_MFMA_BUSY_CYCLES" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --pmc $SET -d $GRAFT_REPO_ROOT/$OUT/pass$i -o p --output-format csv -- python $GRAFT_REPO_ROOT/scripts/pmc_wide.py run > $GRAFT_REPO_ROOT/$OUT/pass$i.log 2>&1)
done
python scripts/pmc_wide.py summarize $OUT gpurun_out/r06_pmc_wide.json | tail -80
find $OUT -name "*.csv" -size +2000k -delete; find $OUT -name "*.db" -delete
