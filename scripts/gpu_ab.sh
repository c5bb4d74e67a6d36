#!/bin/bash
# A/B of two builds of the library on one box: ab/base.so vs ab/new.so (scratch, not committed)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out/ab
for v in base new base new; do
  cp ab/$v.so cuvs_amd/libcuvs_c.so
  echo "== $v"; timeout 600 python scripts/pq_scan_variants.py --steps 5 "$@" 2>&1 | grep -v amdgpu.ids | awk '/pq_scan stats/{c++; if (c%7!=1) next} {print}'
done
