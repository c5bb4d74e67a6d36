#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
T=${1:-r05w}
timeout 900 python -m pytest tests/test_coarse_grouped_gpu.py tests/test_list_shard_world2_gpu.py tests/test_list_shard_gpu.py tests/test_bench_shapes_gpu.py tests/test_ivf_pq_gpu.py -q -x --timeout 600 -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1
echo "rc=$?"; grep -E "passed|failed" gpurun_out/${T}_tests.log | tail -3; grep -E "^FAILED|^ERROR" gpurun_out/${T}_tests.log | cut -c1-260 | head
timeout 300 python bench.py --gpus 2 --share-devices --rows 4000000 --n-lists 1024 --steps 5 --warmup 2 --no-extras > gpurun_out/${T}_gpus2_shared.json 2> gpurun_out/${T}_gpus2_shared.err
echo "shared rc=$?"; grep '^{"metric"' gpurun_out/${T}_gpus2_shared.json | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['scan_kernels']['phase_ms_per_step'], d['recall_at_10'], d['scan3_equals_lut_scan'])"
timeout 300 python bench.py --config c5 --gpus 2 --share-devices --rows 20000000 --n-lists 4096 --steps 5 --warmup 2 > gpurun_out/${T}_c5_gpus2_shared.json 2> gpurun_out/${T}_c5_gpus2_shared.err
echo "c5 shared rc=$?"; grep '^{"metric"' gpurun_out/${T}_c5_gpus2_shared.json | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['value'], d['ms_per_step'], d['roofline']['scan_kernels']['phase_ms_per_step'], d['recall_at_10'])"
