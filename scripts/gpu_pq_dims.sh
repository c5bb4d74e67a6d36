#!/bin/bash
# IVF-PQ search time across pq_dim at 10M x 96 (off the headline configuration): looking for performance cliffs
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
cat > /tmp/pqd.py <<'PY'
import sys, os, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench, cuvs_amd
from cuvs_amd.neighbors import ivf_pq
res = cuvs_amd.common.Resources(); dev = torch.device("cuda:0")
n, dim, nq = 10_000_000, 96, 10000
x = bench.gen_rows(n, dim, 1234, dev); q = bench.gen_rows(nq, dim, 4321, dev)
for pq_dim, bits in ((32, 8), (48, 8), (64, 8), (96, 8), (48, 5)):
    t0 = time.time()
    idx = ivf_pq.build(ivf_pq.IndexParams(n_lists=4096, pq_dim=pq_dim, pq_bits=bits, kmeans_trainset_fraction=0.05), x, resources=res)
    res.sync(); tb = time.time() - t0
    sp = ivf_pq.SearchParams(n_probes=64, lut_dtype="f16" if hasattr(ivf_pq, "x") else 2, internal_distance_dtype=2) if False else ivf_pq.SearchParams(n_probes=64, lut_dtype=bench.LUTS["f16"], internal_distance_dtype=bench.LUTS["f16"])
    for _ in range(2): ivf_pq.search(sp, idx, q, 20, resources=res)
    res.sync(); torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): ivf_pq.search(sp, idx, q, 20, resources=res)
    res.sync(); torch.cuda.synchronize()
    print("pq_dim %d bits %d: build %.1fs search %.2f ms / 10k queries" % (pq_dim, bits, tb, (time.perf_counter() - t) / 5 * 1e3), flush=True)
    del idx
PY
timeout 900 python /tmp/pqd.py 2>&1 | grep -v amdgpu.ids | tail -8
