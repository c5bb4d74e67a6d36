#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
W=/tmp/prof_r03l; rm -rf $W; mkdir -p $W
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $W/kt -o kt -- python scripts/bench_other.py flat > gpurun_out/r03l_kt.log 2>&1
find $W/kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/r03l_kernel_stats.csv \;
python - <<'P'
import csv
rows=list(csv.reader(open('gpurun_out/r03l_kernel_stats.csv')))
for x in rows[1:16]:
    print(f"{float(x[2])/1e6:10.3f} ms total  calls {x[1]:>6}  avg {float(x[3])/1e6:8.3f} ms  {x[0][:100].replace('cuvs_amd::(anonymous namespace)::','')}")
P
