#!/bin/bash
# round 3: full GPU suite + full bench line + kernel trace with the matrix-core tail phase
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -X faulthandler -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/r03g_tests.log 2>&1
echo "tests rc=$?"; grep -v "^  File\|^Extension\|^Running MG\|^MG " gpurun_out/r03g_tests.log | tail -12
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03g_bench.json 2> gpurun_out/r03g_bench.err
echo "bench rc=$?"; tail -3 gpurun_out/r03g_bench.err; cut -c1-2600 gpurun_out/r03g_bench.json
W=/tmp/prof_r03; rm -rf $W; mkdir -p $W gpurun_out/prof_r03g
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $W/kt -o kt -- python bench.py --steps 5 --warmup 2 --no-extras --no-pmc --no-variants --no-cpu-baseline --gt-queries 100 > gpurun_out/prof_r03g/bench_kt.log 2>&1
find $W/kt -name "*kernel_stats.csv" -exec cp {} gpurun_out/prof_r03g/bench100m_kernel_stats.csv \;
grep "pq_\|select_k\|dist_mfma\|pool_merge\|refine\|group_by\|rocprim\|units\|rotate\|labels" gpurun_out/prof_r03g/bench100m_kernel_stats.csv | cut -c1-90,200-330 | head -30
