#!/bin/bash
# round 5: multi-rank functional evidence on the final tree + C5 at 50M rows on one GPU
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
T=${1:-r05u}
timeout 600 python bench.py --gpus 2 --share-devices --rows 4000000 --n-lists 1024 --steps 5 --warmup 2 --no-extras > gpurun_out/${T}_gpus2_shared.json 2> gpurun_out/${T}_gpus2_shared.err
echo "shared rc=$?"; grep '^{"metric"' gpurun_out/${T}_gpus2_shared.json | cut -c1-300
timeout 600 python bench.py --config c5 --gpus 2 --share-devices --rows 20000000 --n-lists 4096 --steps 5 --warmup 2 > gpurun_out/${T}_c5_gpus2_shared.json 2> gpurun_out/${T}_c5_gpus2_shared.err
echo "c5 shared rc=$?"; grep '^{"metric"' gpurun_out/${T}_c5_gpus2_shared.json | cut -c1-300
python bench.py --gpus 2 --rows 4000000 --steps 3 --warmup 1 > /dev/null 2> gpurun_out/${T}_gpus2_refused.err; echo "refusal rc=$? (expected 2)"
timeout 600 python bench.py --config c5 --rows 50000000 --steps 10 --warmup 2 > gpurun_out/${T}_c5_50m.json 2> gpurun_out/${T}_c5.err
echo "c5 50M rc=$?"; grep '^{"metric"' gpurun_out/${T}_c5_50m.json | cut -c1-700
