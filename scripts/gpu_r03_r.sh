#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 70 python scripts/host_enqueue_probe.py 2>&1 | grep -v "^\[bench\]" | tee gpurun_out/r03r_host_probe.log | grep "ms\|us\|rror" | tail -12
