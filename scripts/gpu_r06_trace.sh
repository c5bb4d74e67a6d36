#!/bin/bash
# rocprofv3 kernel trace + stats of the headline step (100M rows, the same command as the driver's bench minus the side lines):
# profiles/r06_bench100m_kernel_stats.csv and the timeline of one step (scripts/kernel_timeline.py)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
W=/tmp/r06_trace; rm -rf $W; mkdir -p $W gpurun_out
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $W -o kt -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 3 --no-variants --no-extras --no-pmc --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r06_trace_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/r06_trace_bench.log)
find $W -name "*kernel_stats.csv" -exec cp {} gpurun_out/r06_bench100m_kernel_stats.csv \;
KT=$(find $W -name "*kernel_trace.csv" | head -1)
python scripts/kernel_timeline.py $KT pq_filter4_kernel -3 > gpurun_out/r06_step_timeline.txt 2>&1
head -8 gpurun_out/r06_bench100m_kernel_stats.csv | cut -c1-200
head -50 gpurun_out/r06_step_timeline.txt
