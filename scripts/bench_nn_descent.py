#!/usr/bin/env python3
"""NN-descent graph build: time and graph recall (vs brute force on sampled rows)."""
import ctypes as C, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, cuvs_amd
from cuvs_amd._lib import check, lib
from cuvs_amd.neighbors import brute_force

dev = torch.device("cuda", 0); res = cuvs_amd.common.Resources()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 64
x = bench.gen_rows(n, 128, 1234, dev, latent=32, n_modes=1024)
out = torch.empty((n, K), dtype=torch.int32, device=dev)
fn = lib().cuvsAmdNnDescent; fn.restype = C.c_int
for iters in (5, 10, 20):
    torch.cuda.synchronize(); t0 = time.time()
    check(fn(res.get_c_obj(), C.c_void_p(x.data_ptr()), C.c_int64(n), C.c_int64(128), C.c_uint32(K), C.c_int(0), C.c_int(iters), C.c_void_p(out.data_ptr())))
    res.sync(); dt = time.time() - t0
    rows = torch.randperm(n, device=dev)[:1000]
    bf = brute_force.build(x, resources=res); _, gt = brute_force.search(bf, x[rows], K + 1, resources=res); res.sync()
    g = out[rows].cpu().numpy().view(np.uint32).astype(np.int64); t = gt.cpu().numpy()[:, 1:]
    rec = float(np.mean([len(np.intersect1d(a, b)) for a, b in zip(g, t)])) / K
    print(json.dumps({"n": n, "K": K, "max_iters": iters, "seconds": round(dt, 2), "graph_recall": round(rec, 4)}), flush=True)
