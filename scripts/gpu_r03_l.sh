#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
W=/tmp/prof_r03l; rm -rf $W; mkdir -p $W
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $W/kt -o kt -- python scripts/bench_other.py flat > gpurun_out/r03l_kt.log 2>&1
find $W/kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/r03l_kernel_stats.csv \;
python - <<'P'
import csv
rows=list(csv.reader(open('gpurun_out/r03l_kernel_stats.csv')))
tot=0
for x in rows[1:]:
    c=int(x[1])
    if c % 7 == 0 and c <= 70:
        per=float(x[2])/7/1e6; tot+=per
        print(f"{per:8.3f} ms/search x{c//7:2d}  {x[0][:110].replace('cuvs_amd::(anonymous namespace)::','')}")
print('sum',round(tot,3))
P
