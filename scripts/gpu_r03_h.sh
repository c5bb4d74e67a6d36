#!/bin/bash
# kernel trace of the headline search only (per-search kernel list)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
W=/tmp/prof_r03h; rm -rf $W; mkdir -p $W
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $W/kt -o kt -- python bench.py --steps 10 --warmup 2 --no-extras --no-pmc --no-variants --no-cpu-baseline --gt-queries 100 > gpurun_out/r03h_kt.log 2>&1
find $W/kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/r03h_kernel_stats.csv \;
python - <<'P'
import csv
rows=list(csv.reader(open('gpurun_out/r03h_kernel_stats.csv')))
tot=0
for x in rows[1:]:
    n=x[0]; calls=int(x[1])
    if calls % 12 == 0 and calls <= 120:
        per=float(x[2])/12/1e6; tot+=per
        print(f"{per:8.3f} ms/search  x{calls//12:2d}  {n[:110].replace('cuvs_amd::(anonymous namespace)::','')}")
print("sum", round(tot,3))
P
tail -1 gpurun_out/r03h_kt.log | cut -c1-200
