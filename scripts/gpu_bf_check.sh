#!/bin/bash
# brute-force / select_k parity tests + the C1 kernel timeline (scripts/gpu_c1_prof.sh)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 600 python -X faulthandler -m pytest tests/test_brute_force_gpu.py tests/test_fuzz_gpu.py tests/test_ivf_flat_gpu.py tests/test_select_k_gpu.py tests/test_ivf_pq_gpu.py -m gpu -q -p no:cacheprovider -x 2>&1 | tail -3
bash scripts/gpu_c1_prof.sh 2>&1 | grep "^ms\|^ *[0-9]" | tail -16
