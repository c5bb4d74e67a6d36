#!/bin/bash
# kernel-trace profile of the CAGRA side measurement (C4 scaled to 1M x 768 fp16): summary -> gpurun_out/other/
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/other
W=/tmp/kt_cagra; rm -rf $W; mkdir -p $W
timeout ${1:-60} rocprofv3 --kernel-trace --stats --output-format csv -d $W -o kt -- python scripts/bench_other.py cagra > gpurun_out/other/bench_cagra.log 2>&1
grep '^{' gpurun_out/other/bench_cagra.log
find $W -name "*kernel_stats.csv" -exec cp {} gpurun_out/other/cagra_kernel_stats.csv \;
head -8 gpurun_out/other/cagra_kernel_stats.csv | cut -c1-200
