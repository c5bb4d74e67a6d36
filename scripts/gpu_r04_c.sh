#!/bin/bash
# round 4: is the filter waiting for memory? cache-hot term / code loads (timing only)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python scripts/pq_scan_variants.py --steps 5 "F4=1,LUT=f16,ACC=f32" \
  "F4=1,DBG=1048576,LUT=f16,ACC=f32" "F4=1,DBG=4194304,LUT=f16,ACC=f32" "F4=1,DBG=5242880,LUT=f16,ACC=f32" "F4=1,DBG=5308416,LUT=f16,ACC=f32" > gpurun_out/r04c_variants.log 2>&1
echo "variants rc=$?"; grep -v "^\[bench\]" gpurun_out/r04c_variants.log | grep -v "overflow entries\|pairs screened" | tail -30
