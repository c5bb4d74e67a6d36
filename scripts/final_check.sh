#!/bin/bash
# round-end check on the GPU box: full GPU suite, smoke, kernel-trace profile of the default bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/final
timeout 600 python -X faulthandler -m pytest tests -m gpu -q --timeout 200 --tb=short > gpurun_out/final/pytest.log 2>&1
grep -v "^  File\|^Extension" gpurun_out/final/pytest.log | tail -45
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2
W=/tmp/kt_work; rm -rf $W; mkdir -p $W
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $W -o kt -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --gt-queries 10 > gpurun_out/final/kt.log 2>&1
tail -1 gpurun_out/final/kt.log | cut -c1-400
find $W -name "*kernel_stats.csv" -exec cp {} gpurun_out/final/kernel_stats.csv \;
head -8 gpurun_out/final/kernel_stats.csv | cut -c1-160
