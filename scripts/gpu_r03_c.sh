#!/bin/bash
# round 3, re-entry call: microbenchmarks, full GPU test suite, full bench line, kernel trace of the headline
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
scripts/bin/lds_gather_bench > gpurun_out/r03_lds_gather_bench.json 2> gpurun_out/r03_lds_gather_bench.err; echo "lds bench rc=$?"
scripts/bin/valu_rate_bench > gpurun_out/r03_valu_rate_bench.json 2> gpurun_out/r03_valu_rate_bench.err; echo "valu bench rc=$?"
timeout 900 python -X faulthandler -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider > gpurun_out/r03c_tests.log 2>&1
echo "tests rc=$?"; grep -v "^  File\|^Extension" gpurun_out/r03c_tests.log | tail -15
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03c_bench.json 2> gpurun_out/r03c_bench.err
echo "bench rc=$?"; tail -3 gpurun_out/r03c_bench.err; cut -c1-3000 gpurun_out/r03c_bench.json
W=/tmp/prof_r03; rm -rf $W; mkdir -p $W gpurun_out/prof_r03
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $W/kt -o kt -- python bench.py --steps 5 --warmup 2 --no-extras --no-pmc --no-variants --no-cpu-baseline --gt-queries 100 > gpurun_out/prof_r03/bench_kt.log 2>&1
find $W/kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_r03/bench100m_kernel_stats.csv \;
head -8 gpurun_out/prof_r03/bench100m_kernel_stats.csv | cut -c1-200
