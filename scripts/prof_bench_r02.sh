#!/bin/bash
# round-2 profile evidence: rocprofv3 --kernel-trace --stats of the default bench command (extras and PMC children off,
# so that the trace holds the headline workload only), summaries copied to gpurun_out/prof_r02/
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; rm -rf gpurun_out/prof_r02; mkdir -p gpurun_out/prof_r02
W=/tmp/prof_r02; rm -rf $W; mkdir -p $W
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $W/kt -o kt -- python bench.py --steps 5 --warmup 2 --no-extras --no-pmc --no-variants --no-cpu-baseline --gt-queries 100 > gpurun_out/prof_r02/bench_kt.log 2>&1
find $W/kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_r02/bench100m_kernel_stats.csv \;
head -12 gpurun_out/prof_r02/bench100m_kernel_stats.csv
tail -1 gpurun_out/prof_r02/bench_kt.log | cut -c1-600
