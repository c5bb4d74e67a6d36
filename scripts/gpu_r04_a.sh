#!/bin/bash
# round 4, first call: the full GPU suite on the new one-wave-per-SIMD filter, then A/B of the filter kernels on the 100M index
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x > gpurun_out/r04a_tests.log 2>&1
echo "tests rc=$?"; grep "passed\|failed\|rror" gpurun_out/r04a_tests.log | tail -8
timeout 900 python scripts/pq_scan_variants.py --steps 5 "S3=0,LUT=f16,ACC=f32" "F4=0,LUT=f16,ACC=f32" "F4=1,LUT=f16,ACC=f32" \
  "F4=1,DBG=1024,LUT=f16,ACC=f32" "F4=0,DBG=1024,LUT=f16,ACC=f32" \
  "F4=1,DBG=65536,LUT=f16,ACC=f32" "F4=1,DBG=131072,LUT=f16,ACC=f32" "F4=1,DBG=262144,LUT=f16,ACC=f32" "F4=1,DBG=524288,LUT=f16,ACC=f32" \
  "F4=1,LUT=f32" "F4=0,LUT=f32" > gpurun_out/r04a_variants.log 2>&1
echo "variants rc=$?"; grep -v "^\[bench\]" gpurun_out/r04a_variants.log | tail -30
