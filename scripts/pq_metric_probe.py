#!/usr/bin/env python3
"""IVF-PQ search time of an inner-product / cosine index with and without the matrix-core tail phase, with its statistics.
  python scripts/pq_metric_probe.py [--rows N] [--metric inner_product]"""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=20_000_000)
ap.add_argument("--n-lists", type=int, default=4096)
ap.add_argument("--metric", default="inner_product")
a = ap.parse_args()
import cuvs_amd
from cuvs_amd.neighbors import ivf_pq
dev = torch.device("cuda", 0)
res = cuvs_amd.common.Resources()
x = bench.gen_rows(a.rows, 128, 1234, dev)
q = bench.gen_rows(10000, 128, 4321, dev)
idx = ivf_pq.build(ivf_pq.IndexParams(n_lists=a.n_lists, metric=a.metric, pq_dim=64, kmeans_trainset_fraction=0.05), x, resources=res)
res.sync()
sp = ivf_pq.SearchParams(n_probes=128, lut_dtype=np.float16, internal_distance_dtype=np.float32, max_internal_batch_size=10000)
outs = {}
for env in ({}, {"CUVS_AMD_SCAN_DEBUG": str(1024 + 2048)}, {"CUVS_AMD_PQ_SCAN3": "0"}):
    for k_ in ("CUVS_AMD_SCAN_DEBUG", "CUVS_AMD_PQ_SCAN3"):
        os.environ.pop(k_, None)
    os.environ.update(env)
    r = cuvs_amd.common.Resources()
    for _ in range(2):
        d, i = ivf_pq.search(sp, idx, q, 20, resources=r)
    r.sync(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3):
        d, i = ivf_pq.search(sp, idx, q, 20, resources=r)
    r.sync(); torch.cuda.synchronize()
    print(env, f"{(time.perf_counter() - t0) / 3 * 1e3:.3f} ms", flush=True)
    outs[str(env)] = (d.clone(), i.clone())
v = list(outs.values())
print("same results:", all(torch.equal(v[0][0], o[0]) and torch.equal(v[0][1], o[1]) for o in v[1:]))
