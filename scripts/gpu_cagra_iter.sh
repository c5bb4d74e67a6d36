#!/bin/bash
# CAGRA search timing at 2M x 768 fp16 (latent 24), auto / single / multi
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for algo in auto single_cta multi_cta; do
  CAGRA_ALGO=$algo timeout 900 python scripts/bench_other.py cagra --cagra-rows 2000000 --cagra-latent 24 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-300
done
