"""IVF-PQ search time off the headline shape (10M x 96, n_lists 4096, n_probes 64, 10k queries, k 20, fp16 LUT / score):
pq_dim / pq_len / pq_bits / k variants through the matrix-core tail phase and, for comparison, on the LUT scan
kernels (CUVS_AMD_PQ_SCAN3=0). Usage: python scripts/pq_len_timing.py"""
import os
import sys
import time

os.environ["CUVS_AMD_DEBUG_SWITCHES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

os.environ["CUVS_AMD_DEBUG_SWITCHES"] = "1"  # (bench.py drops the gate on import: its timed handles run the production configuration)
import cuvs_amd
from cuvs_amd.neighbors import ivf_pq

dev = torch.device("cuda:0")
res = cuvs_amd.common.Resources()
os.environ["CUVS_AMD_PQ_SCAN3"] = "0"
res_lut = cuvs_amd.common.Resources()  # switches are read once per handle
del os.environ["CUVS_AMD_PQ_SCAN3"]
n, nq = 10_000_000, 10000
for dim, pq_dim, bits, k in ((96, 96, 8, 20), (96, 48, 8, 20), (128, 32, 8, 20), (96, 48, 5, 20), (96, 96, 5, 20), (96, 48, 8, 64), (96, 48, 8, 256)):
    x = bench.gen_rows(n, dim, 1234, dev)
    q = bench.gen_rows(nq, dim, 4321, dev)
    idx = ivf_pq.build(ivf_pq.IndexParams(n_lists=4096, pq_dim=pq_dim, pq_bits=bits, kmeans_trainset_fraction=0.05), x, resources=res)
    res.sync()
    sp = ivf_pq.SearchParams(n_probes=64, lut_dtype=bench.LUTS["f16"], internal_distance_dtype=bench.LUTS["f16"])
    out = {}
    for name, r in (("matrix-core tail", res), ("LUT scan", res_lut)):
        for _ in range(2):
            d0, i0 = ivf_pq.search(sp, idx, q, k, resources=r)
        r.sync(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5):
            ivf_pq.search(sp, idx, q, k, resources=r)
        r.sync(); torch.cuda.synchronize()
        out[name] = ((time.perf_counter() - t) / 5 * 1e3, d0, i0)
    same = bool((out["matrix-core tail"][1] == out["LUT scan"][1]).all() and (out["matrix-core tail"][2] == out["LUT scan"][2]).all())
    print(f"dim {dim} pq_dim {pq_dim} x {bits} bit (pq_len {dim // pq_dim}) k {k}: matrix-core tail {out['matrix-core tail'][0]:.2f} ms, "
          f"LUT scan {out['LUT scan'][0]:.2f} ms per 10k queries, identical results: {same}", flush=True)
    del idx, x, q
