#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 50 python scripts/host_trace_flat.py 2>&1 | grep -v "^\[bench\]" | tee gpurun_out/r03s_host_trace.log | grep "host trace\|call\|back-to\|rror" | tail -14
