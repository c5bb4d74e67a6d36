#!/bin/bash
# brute-force iteration on the GPU: parity tests + C1 / 10M timings
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/iter
timeout 900 python -m pytest tests/test_brute_force_gpu.py -x -q > gpurun_out/iter/bf_tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/iter/bf_tests.log
timeout 900 python scripts/bench_other.py bf 2>&1 | grep -v amdgpu.ids
CUVS_AMD_BF_NO_FUSED_FILTER=1 timeout 900 python scripts/bench_other.py bf 2>&1 | grep -v amdgpu.ids
