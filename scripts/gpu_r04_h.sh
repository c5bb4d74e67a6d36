#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python scripts/pq_scan_variants.py --steps 5 "F4=0,LUT=f16,ACC=f32" "F4=1,LUT=f16,ACC=f32" "F4=1,LUT=f32" > gpurun_out/r04h_l2.log 2>&1
grep -v "^\[bench\]" gpurun_out/r04h_l2.log | tail -3
timeout 300 python scripts/host_trace_flat.py > gpurun_out/r04h_host_trace.log 2>&1; tail -40 gpurun_out/r04h_host_trace.log
