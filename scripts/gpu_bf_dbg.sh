#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cat > /tmp/bf2m.py <<'PY'
import sys, os, torch, time
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench, cuvs_amd
from cuvs_amd.neighbors import brute_force
res = cuvs_amd.common.Resources(); dev = torch.device("cuda:0")
x2 = bench.gen_rows(4_000_000, 128, 1234, dev); q2 = bench.gen_rows(10000, 128, 4321, dev)
idx2 = brute_force.build(x2, resources=res)
for _ in range(3):
    torch.cuda.synchronize(); t = time.perf_counter()
    brute_force.search(idx2, q2, 10, resources=res); res.sync(); torch.cuda.synchronize()
    dt = time.perf_counter() - t
print(os.environ.get("CUVS_AMD_TILE_DBG"), "ms", dt * 1e3, "TF", 2 * 10000 * 4e6 * 128 / dt / 1e12)
PY
for d in "$@"; do CUVS_AMD_TILE_DBG=$d timeout 300 python /tmp/bf2m.py 2>&1 | grep -v amdgpu.ids | tail -1; done
