#!/bin/bash
# round 4 evidence: full GPU suite, full bench line, kernel trace of the headline, C5 at 50M rows
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
T=${1:-r04z}
timeout 900 python -X faulthandler -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1
echo "tests rc=$?"; grep "passed\|failed\|rror" gpurun_out/${T}_tests.log | tail -5
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$?"; tail -2 gpurun_out/${T}_bench.err; grep '^{"metric"' gpurun_out/${T}_bench.json | cut -c1-400
W=/tmp/prof_${T}; rm -rf $W; mkdir -p $W
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $W/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-extras --no-pmc --no-variants --no-cpu-baseline --gt-queries 100 > $GRAFT_REPO_ROOT/gpurun_out/${T}_kt.log 2>&1)
find $W/kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/${T}_kernel_stats.csv \;
python - $T <<'P'
import csv, sys
rows=list(csv.reader(open(f'gpurun_out/{sys.argv[1]}_kernel_stats.csv')))
for x in rows[1:]:
    c=int(x[1])
    if c % 12 == 0 and c <= 120 and float(x[2])/12/1e6 > 0.02:
        print(f"{float(x[2])/12/1e6:8.3f} ms/search x{c//12:2d}  {x[0][:100].replace('cuvs_amd::(anonymous namespace)::','')}")
P
timeout 600 python bench.py --config c5 --rows 50000000 --steps 10 --warmup 2 > gpurun_out/${T}_c5_50m.json 2> gpurun_out/${T}_c5.err
echo "c5 rc=$?"; grep '^{"metric"' gpurun_out/${T}_c5_50m.json | cut -c1-300
timeout 600 python scripts/pq_len_timing.py 2>&1 | grep -v amdgpu.ids | tail -8 > gpurun_out/${T}_pq_len.log; cat gpurun_out/${T}_pq_len.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $W/cd -o cd -- python $GRAFT_REPO_ROOT/scripts/coarse_dtype_profile.py 16384 > $GRAFT_REPO_ROOT/gpurun_out/${T}_coarse.log 2>&1)
find $W/cd -name "*kernel_stats.csv" -exec cp {} gpurun_out/${T}_coarse_kernel_stats.csv \;
grep "^coarse" gpurun_out/${T}_coarse.log
