#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python scripts/pq_scan_variants.py --steps 1 ${VARIANTS:-"LUT=f16,ACC=f32,DBG=128"} > gpurun_out/r03f_variants.log 2>&1
echo "variants rc=$?"; grep -v "^\[bench\]" gpurun_out/r03f_variants.log | tail -14
