#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 200 python -X faulthandler -m pytest tests/test_list_shard_gpu.py tests/test_row_shard_gpu.py tests/test_mg_capi_gpu.py -q --timeout 200 -p no:cacheprovider > gpurun_out/r03q_tests.log 2>&1
echo "tests rc=$?"; grep "passed\|failed\|rror\|^E " gpurun_out/r03q_tests.log | tail -6
timeout 150 python scripts/shard_overhead_probe.py --skip-plain 2>&1 | grep -v "^\[bench\]" | tee gpurun_out/r03q_shard_probe.log | grep " ms "
