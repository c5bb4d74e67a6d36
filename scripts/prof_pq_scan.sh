#!/bin/bash
# rocprofv3 kernel trace + PMC passes over a bench.py run; leaves only small summaries under gpurun_out/prof/
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; rm -rf gpurun_out/prof; mkdir -p gpurun_out/prof
ARGS="${BENCH_ARGS:---rows 10000000 --n-lists 2048 --n-probes 32 --steps 3 --warmup 1 --no-cpu-baseline --gt-queries 10}"
W=/tmp/prof_work; rm -rf $W; mkdir -p $W
rocprofv3 --kernel-trace --stats --output-format csv -d $W/kt -o kt -- python bench.py $ARGS > gpurun_out/prof/kt.log 2>&1
find $W/kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof/kernel_stats.csv \;
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
P3="FETCH_SIZE"
P4="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
i=1
for P in "$P1" "$P2" "$P3" "$P4"; do
  rocprofv3 --pmc $P --output-format csv -d $W/pmc$i -o pmc$i -- python bench.py $ARGS > gpurun_out/prof/pmc$i.log 2>&1
  f=$(find $W/pmc$i -name "*counter_collection.csv" | head -1)
  python - "$f" gpurun_out/prof/pmc$i.txt <<'PY'
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
        if "pq_scan" in kn or "select_k" in kn or "dist_mfma" in kn:
            o.write(f"{kn:60s} {cn:24s} sum={v:.6g} dispatches={n} per_dispatch={v/n:.6g}\n")
PY
  i=$((i+1))
done
ls -la gpurun_out/prof
