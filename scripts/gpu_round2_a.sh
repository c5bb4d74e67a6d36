#!/bin/bash
# round-2 GPU call A: tests, scan-kernel variants (two-stage vs single stage, all precisions), bench line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/r2a
timeout 600 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r2a/tests.log
timeout 600 python scripts/pq_scan_variants.py --steps 5 "DBG=256,HEAD=1" "DBG=0,HEAD=1" "DBG=8,HEAD=1" "DBG=256,HEAD=1,LUT=f16,ACC=f32" "DBG=0,HEAD=1,LUT=f16,ACC=f32" "DBG=256,HEAD=1,LUT=f32" "DBG=0,HEAD=1,LUT=f32" "DBG=16,HEAD=1" "DBG=17,HEAD=1" "DBG=4,HEAD=1" > gpurun_out/r2a/variants.log 2>&1; echo "variants rc=$?"; cat gpurun_out/r2a/variants.log | grep -v amdgpu.ids
timeout 600 python bench.py --steps 10 --warmup 3 > gpurun_out/r2a/bench.log 2>&1; echo "bench rc=$?"; tail -2 gpurun_out/r2a/bench.log
