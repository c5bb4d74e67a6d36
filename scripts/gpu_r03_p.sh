#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 200 python scripts/shard_overhead_probe.py 2>&1 | grep -v "^\[bench\]" | tee gpurun_out/r03p_shard_probe.log | tail -12
