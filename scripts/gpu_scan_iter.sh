#!/bin/bash
# quick scan-kernel iteration on the GPU: IVF-PQ parity tests + timing variants at the bench workload
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/iter
timeout 600 python -m pytest tests/test_ivf_pq_gpu.py -x -q > gpurun_out/iter/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/iter/tests.log
timeout 900 python scripts/pq_scan_variants.py --steps 5 "$@" > gpurun_out/iter/variants.log 2>&1; echo "variants rc=$?"; grep -v amdgpu.ids gpurun_out/iter/variants.log
