#!/bin/bash
# round 5 evidence: full GPU suite, full bench line (the driver's command), kernel trace + timeline of the headline
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
T=${1:-r05z}
rm -f gpurun_out/${T}_table_recalls.txt
CUVS_AMD_TABLE_LOG=gpurun_out/${T}_table_recalls.txt timeout 1500 python -X faulthandler -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1
echo "tests rc=$?"; grep -E "passed|failed|rror" gpurun_out/${T}_tests.log | tail -5; grep -E "^FAILED|^ERROR" gpurun_out/${T}_tests.log | cut -c1-200 | head -30
timeout 1500 python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
echo "bench rc=$?"; tail -3 gpurun_out/${T}_bench.err | cut -c1-300; grep '^{"metric"' gpurun_out/${T}_bench.json | cut -c1-600
if [ "$2" != "notrace" ]; then
W=/tmp/prof_${T}; rm -rf $W; mkdir -p $W
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $W/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-extras --no-pmc --no-variants --no-cpu-baseline --gt-queries 100 > $GRAFT_REPO_ROOT/gpurun_out/${T}_kt.log 2>&1)
find $W/kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/${T}_kernel_stats.csv \;
TR=$(find $W/kt -name "*kernel_trace.csv" | head -1)
python scripts/kernel_timeline.py $TR pq_filter4_kernel -3 > gpurun_out/${T}_timeline.txt 2>&1
fi
