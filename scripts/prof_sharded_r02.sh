#!/bin/bash
# kernel trace of the list-sharded step with a one-rank communicator (native RCCL all-gather / all-reduce really run)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/prof_r02
W=/tmp/prof_sh; rm -rf $W; mkdir -p $W
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $W/kt -o kt -- python bench.py --steps 5 --warmup 2 --rows 20000000 --n-lists 8192 --force-sharded --no-extras --no-pmc --no-variants --no-cpu-baseline --gt-queries 100 > gpurun_out/prof_r02/sharded_kt.log 2>&1
find $W/kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_r02/sharded_kernel_stats.csv \;
python - <<'PY'
import csv
rows = list(csv.DictReader(open("gpurun_out/prof_r02/sharded_kernel_stats.csv")))
for r in rows:
    n = r["Name"]
    if any(t in n for t in ("nccl", "Nccl", "rccl", "pq_scan", "pack_block", "regroup", "pad_invalid", "select_k")):
        print(n[:100].replace("cuvs_amd::(anonymous namespace)::", ""), r["Calls"], "avg us %.1f" % (float(r["AverageNs"]) / 1e3))
PY
tail -1 gpurun_out/prof_r02/sharded_kt.log | cut -c1-700
