#!/bin/bash
# partial head: parity + timing sweep over the head's row limit at the bench shape
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
T=${1:-r05p}
timeout 600 python -m pytest tests/test_coarse_grouped_gpu.py tests/test_serialize_format_gpu.py tests/test_list_shard_world2_gpu.py -q --timeout 600 -p no:cacheprovider -x > gpurun_out/${T}_tests.log 2>&1
echo "rc=$?"; grep -E "passed|failed" gpurun_out/${T}_tests.log | tail -3; grep -E "^FAILED|^ERROR" gpurun_out/${T}_tests.log | cut -c1-260 | head
timeout 900 python scripts/head_rows_sweep.py > gpurun_out/${T}_sweep.log 2>&1
grep -v amdgpu.ids gpurun_out/${T}_sweep.log | tail -20
