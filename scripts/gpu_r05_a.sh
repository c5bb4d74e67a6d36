#!/bin/bash
# round 5, first GPU evidence: world > 1 through the host-staged transport (tests + bench self-launch)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
T=${1:-r05a}
timeout 500 python -m pytest tests/test_list_shard_world2_gpu.py tests/test_list_shard_gpu.py tests/test_row_shard_gpu.py -q --timeout 480 -p no:cacheprovider > gpurun_out/${T}_tests.log 2>&1
echo "tests rc=$?"; tail -15 gpurun_out/${T}_tests.log
# --gpus 2 on a one-GPU box: must refuse loudly
python bench.py --gpus 2 --rows 4000000 --steps 3 --warmup 1 > gpurun_out/${T}_gpus2_refused.json 2> gpurun_out/${T}_gpus2_refused.err
echo "refusal rc=$? (expected 2)"; tail -2 gpurun_out/${T}_gpus2_refused.err
# the same with shared devices: the 2-rank path end to end, functional
timeout 600 python bench.py --gpus 2 --share-devices --rows 4000000 --n-lists 1024 --steps 5 --warmup 2 --no-extras > gpurun_out/${T}_gpus2_shared.json 2> gpurun_out/${T}_gpus2_shared.err
echo "shared rc=$?"; tail -5 gpurun_out/${T}_gpus2_shared.err; grep '^{"metric"' gpurun_out/${T}_gpus2_shared.json | cut -c1-1500
timeout 600 python bench.py --config c5 --gpus 2 --share-devices --rows 20000000 --n-lists 4096 --steps 5 --warmup 2 > gpurun_out/${T}_c5_gpus2_shared.json 2> gpurun_out/${T}_c5_gpus2_shared.err
echo "c5 shared rc=$?"; tail -5 gpurun_out/${T}_c5_gpus2_shared.err; grep '^{"metric"' gpurun_out/${T}_c5_gpus2_shared.json | cut -c1-1500
