#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
T=${1:-r05r}
timeout 900 python scripts/head_rows_sweep.py 100000000 0 -1 0 -1 > gpurun_out/${T}_sweep.log 2>&1
grep -v amdgpu.ids gpurun_out/${T}_sweep.log | tail -6
