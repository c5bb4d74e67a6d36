#!/bin/bash
# round 4: full GPU suite, the full bench line (new fields: scan3_equals_lut_scan, corpus_variants, roofline against spec peaks), C5 at 50M rows with shard-local refine
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/r04d_tests.log 2>&1
echo "tests rc=$?"; grep "passed\|failed\|rror" gpurun_out/r04d_tests.log | tail -8
timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/r04d_bench.json 2> gpurun_out/r04d_bench.err
echo "bench rc=$?"; tail -3 gpurun_out/r04d_bench.err; grep '^{"metric"' gpurun_out/r04d_bench.json | cut -c1-600
timeout 600 python bench.py --config c5 --rows 50000000 --steps 5 --warmup 2 > gpurun_out/r04d_c5_50m.json 2> gpurun_out/r04d_c5.err
echo "c5 rc=$?"; tail -2 gpurun_out/r04d_c5.err; grep '^{"metric"' gpurun_out/r04d_c5_50m.json | cut -c1-900
