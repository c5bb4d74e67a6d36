#!/bin/bash
# IVF-Flat C2 timing (10M x 128 fp32, n_lists 4096, n_probes 64, batch 10k)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python scripts/bench_other.py flat 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300
