#!/bin/bash
# kernel-trace profile of the side measurements (brute force C1 + 10M, IVF-Flat C2): summary -> gpurun_out/other/
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/other
W=/tmp/kt_other; rm -rf $W; mkdir -p $W
timeout ${1:-100} rocprofv3 --kernel-trace --stats --output-format csv -d $W -o kt -- python scripts/bench_other.py bf flat > gpurun_out/other/bench_other.log 2>&1
grep '^{' gpurun_out/other/bench_other.log
find $W -name "*kernel_stats.csv" -exec cp {} gpurun_out/other/kernel_stats.csv \;
head -12 gpurun_out/other/kernel_stats.csv | cut -c1-200
