#!/bin/bash
# PMC passes (rocprofv3 --pmc only, one counter group per run, each under its own timeout) over
# scripts/pq_scan_variants.py at the bench.py workload; writes gpurun_out/pmc/<tag>_pmcN.txt summaries.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; TAG=${1:-cur}; VAR="${2:-DBG=0,HEAD=1}"; mkdir -p gpurun_out/pmc
W=/tmp/pmc_work; rm -rf $W; mkdir -p $W
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM"
P3="FETCH_SIZE"
P4="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
i=1
for P in "$P1" "$P2" "$P3" "$P4"; do
  timeout ${PMC_TIMEOUT:-150} rocprofv3 --pmc $P --output-format csv -d $W/pmc$i -o pmc$i -- python scripts/pq_scan_variants.py --steps 2 "$VAR" > gpurun_out/pmc/${TAG}_pmc$i.log 2>&1
  echo "pass $i rc=$?"
  f=$(find $W/pmc$i -name "*counter_collection.csv" | head -1)
  python - "$f" gpurun_out/pmc/${TAG}_pmc$i.txt <<'PY'
import csv, sys, collections
f, out = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0.0, 0])
try:
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:60], r["Counter_Name"])
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
except Exception as e:
    open(out, "w").write(f"error {e}\n"); sys.exit(0)
with open(out, "w") as o:
    for (kn, cn), (v, n) in sorted(agg.items()):
        if "pq_scan" in kn:
            o.write(f"{kn:60s} {cn:24s} sum={v:.6g} dispatches={n} per_dispatch={v/n:.6g}\n")
PY
  cat gpurun_out/pmc/${TAG}_pmc$i.txt
  i=$((i+1))
done
