#!/bin/bash
# round 3: full GPU suite, full bench line, C5 (scaled down) through the sharded path, kernel trace
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1200 python -X faulthandler -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/r03i_tests.log 2>&1
echo "tests rc=$?"; grep -v "^  File\|^Extension\|^Running MG\|^MG " gpurun_out/r03i_tests.log | grep "passed\|failed\|rror" | tail -8
timeout 600 python bench.py --config c5 --rows 50000000 --n-lists 8192 --steps 5 --warmup 2 > gpurun_out/r03i_c5_50m.json 2> gpurun_out/r03i_c5.err
echo "c5 rc=$?"; tail -2 gpurun_out/r03i_c5.err; cat gpurun_out/r03i_c5_50m.json
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r03i_bench.json 2> gpurun_out/r03i_bench.err
echo "bench rc=$?"; tail -3 gpurun_out/r03i_bench.err; cut -c1-3400 gpurun_out/r03i_bench.json
