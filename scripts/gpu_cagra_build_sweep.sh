#!/bin/bash
# CAGRA kNN-graph builder sweep at 2M x 768 fp16: "lists,probes,kpq" triples (0 = default)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for spec in "$@"; do
  IFS=, read L P K <<< "$spec"
  env ${L:+CUVS_AMD_CAGRA_PQ_LISTS=$L} ${P:+CUVS_AMD_CAGRA_PQ_PROBES=$P} ${K:+CUVS_AMD_CAGRA_KPQ=$K} CAGRA_ALGO=multi_cta timeout 900 python scripts/bench_other.py cagra --cagra-rows 2000000 --cagra-latent 24 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/$spec /" | cut -c1-330
done
