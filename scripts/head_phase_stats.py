"""Per-phase cycles of pq_head_kernel on the headline workload (CUVS_AMD_SCAN_DEBUG=2048: header / LUT build / scores / select / output
cycles per item, printed by the library on stderr) and the head kernel's time. Usage: python scripts/head_phase_stats.py [rows]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import cuvs_amd  # noqa: E402
from cuvs_amd._lib import lib  # noqa: E402
from cuvs_amd.neighbors import ivf_pq  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
data = bench.gen_rows(rows, 128, seed=1234, device=dev)
queries = bench.gen_rows(10000, 128, seed=4321, device=dev)
res = cuvs_amd.common.Resources()
index = ivf_pq.build(ivf_pq.IndexParams(n_lists=16384, pq_dim=64, pq_bits=8, kmeans_n_iters=20, kmeans_trainset_fraction=0.02), data, resources=res)
res.sync()
del data
sp = ivf_pq.SearchParams(n_probes=128, lut_dtype=bench.LUTS["f16"], internal_distance_dtype=bench.LUTS["f32"], max_internal_batch_size=10000)
for kk in (20, 60):
    ci = torch.empty((10000, kk), dtype=torch.int64, device=dev)
    cd = torch.empty((10000, kk), dtype=torch.float32, device=dev)
    for name, r in (("default", res), ("stats", bench.comparator_handle(CUVS_AMD_SCAN_DEBUG=2048))):
        for _ in range(3):
            ivf_pq.search(sp, index, queries, kk, neighbors=ci, distances=cd, resources=r)
        lib().cuvsAmdProfileEnable(1)
        for _ in range(5):
            ivf_pq.search(sp, index, queries, kk, neighbors=ci, distances=cd, resources=r)
        r.sync(); torch.cuda.synchronize()
        lib().cuvsAmdProfileEnable(0)
        ph = {}
        for nm in (b"pq_head_kernel", b"pq_filter_kernel", b"pq_rescore_kernel", b"pq_scan_kernel"):
            v = C.c_double(0)
            lib().cuvsAmdProfileCollect(nm, C.byref(v))
            ph[nm.decode()] = round(v.value / 5, 3)
        print(f"k {kk} {name}: {ph}", flush=True)
