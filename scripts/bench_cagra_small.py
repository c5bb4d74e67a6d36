#!/usr/bin/env python3
"""Small-batch CAGRA latency: one wave per query (SINGLE_CTA) vs the multi-wave walk (MULTI_CTA / AUTO)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, cuvs_amd
from cuvs_amd.neighbors import brute_force, cagra

dev = torch.device("cuda", 0); res = cuvs_amd.common.Resources()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
x = bench.gen_rows(n, 128, 1234, dev, latent=32, n_modes=1); qall = bench.gen_rows(1000, 128, 4321, dev, latent=32, n_modes=1)
idx = cagra.build(cagra.IndexParams(intermediate_graph_degree=64, graph_degree=32), x, resources=res); res.sync()
bf = brute_force.build(x, resources=res); _, gt = brute_force.search(bf, qall, 10, resources=res); res.sync(); gt = gt.cpu().numpy()
for batch in (1, 10, 100, 1000):
    q = qall[:batch].contiguous()
    for algo in ("single_cta", "multi_cta", "auto"):
        sp = cagra.SearchParams(itopk_size=64, algo=algo)
        for _ in range(3): d, i = cagra.search(sp, idx, q, 10, resources=res)
        res.sync(); t0 = time.perf_counter()
        for _ in range(20): d, i = cagra.search(sp, idx, q, 10, resources=res)
        res.sync(); ms = (time.perf_counter() - t0) / 20 * 1e3
        found = i.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        rec = float(np.mean([len(np.intersect1d(f, t)) for f, t in zip(found, gt[:batch])])) / 10
        print(json.dumps({"n": n, "batch": batch, "algo": algo, "ms": round(ms, 4), "qps": round(batch / ms * 1e3, 1), "recall": round(rec, 4)}), flush=True)
