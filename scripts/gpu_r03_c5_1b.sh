#!/bin/bash
# C5 at full size on ONE GPU: 1B x 96 int8 rows (BASELINE configs[4] is the same index over 8 GPUs)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
ROWS=${ROWS:-1000000000}
timeout ${TMO:-540} python bench.py --config c5 --rows $ROWS --steps 5 --warmup 2 --gt-queries 100 > gpurun_out/r03_c5_${ROWS}.json 2> gpurun_out/r03_c5_${ROWS}.err
echo "c5 rc=$?"; tail -3 gpurun_out/r03_c5_${ROWS}.err; cat gpurun_out/r03_c5_${ROWS}.json
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used\|total" | head -3
