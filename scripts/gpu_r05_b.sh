#!/bin/bash
# round 5: kernel timeline of the headline step (gaps between kernels), for the step-time work
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
T=${1:-r05b}
W=/tmp/prof_${T}; rm -rf $W; mkdir -p $W
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $W/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-extras --no-pmc --no-variants --no-cpu-baseline --gt-queries 100 > $GRAFT_REPO_ROOT/gpurun_out/${T}_kt.log 2>&1)
find $W/kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/${T}_kernel_stats.csv \;
TR=$(find $W/kt -name "*kernel_trace.csv" | head -1)
python scripts/kernel_timeline.py $TR pq_filter4_kernel -3 > gpurun_out/${T}_timeline.txt 2>&1
cat gpurun_out/${T}_timeline.txt
grep '^{"metric"' gpurun_out/${T}_kt.log | cut -c1-300
