#!/bin/bash
# round 3 evidence: full GPU suite, full bench line, kernel trace of the headline, C2 stand-alone, C5 at 50M rows
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/r03z_tests.log 2>&1
echo "tests rc=$?"; grep "passed\|failed\|rror" gpurun_out/r03z_tests.log | tail -5
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03z_bench.json 2> gpurun_out/r03z_bench.err
echo "bench rc=$?"; tail -2 gpurun_out/r03z_bench.err; grep '^{"metric"' gpurun_out/r03z_bench.json | cut -c1-900
W=/tmp/prof_r03z; rm -rf $W; mkdir -p $W
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $W/kt -o kt -- python bench.py --steps 10 --warmup 2 --no-extras --no-pmc --no-variants --no-cpu-baseline --gt-queries 100 > gpurun_out/r03z_kt.log 2>&1
find $W/kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/r03z_kernel_stats.csv \;
python - <<'P'
import csv
rows=list(csv.reader(open('gpurun_out/r03z_kernel_stats.csv')))
for x in rows[1:]:
    c=int(x[1])
    if c % 12 == 0 and c <= 120 and float(x[2])/12/1e6 > 0.02:
        print(f"{float(x[2])/12/1e6:8.3f} ms/search x{c//12:2d}  {x[0][:100].replace('cuvs_amd::(anonymous namespace)::','')}")
P
timeout 300 python scripts/bench_other.py flat 2>&1 | grep '^{' | cut -c1-400
