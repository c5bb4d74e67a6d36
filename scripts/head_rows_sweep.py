"""Step time of the headline workload as a function of the head phase's row limit (CUVS_AMD_PQ_HEAD_ROWS): ms per step, the
phases' kernel times, survivors per pair. Usage: python scripts/head_rows_sweep.py [rows]"""
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import cuvs_amd  # noqa: E402
from cuvs_amd._lib import lib  # noqa: E402
from cuvs_amd.neighbors import ivf_pq, refine  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda", 0)
data = bench.gen_rows(rows, 128, seed=1234, device=dev)
queries = bench.gen_rows(10000, 128, seed=4321, device=dev)
res0 = cuvs_amd.common.Resources()
index = ivf_pq.build(ivf_pq.IndexParams(n_lists=16384, pq_dim=64, pq_bits=8, kmeans_n_iters=20, kmeans_trainset_fraction=0.02), data, resources=res0)
res0.sync()
kk = 20
ci = torch.empty((10000, kk), dtype=torch.int64, device=dev)
cd = torch.empty((10000, kk), dtype=torch.float32, device=dev)
oi = torch.empty((10000, 10), dtype=torch.int64, device=dev)
od = torch.empty((10000, 10), dtype=torch.float32, device=dev)
sp = ivf_pq.SearchParams(n_probes=128, lut_dtype=bench.LUTS["f16"], internal_distance_dtype=bench.LUTS["f32"], max_internal_batch_size=10000)
ref_i = None
sweep = [int(v) for v in sys.argv[2:]] or [0, 4096, 3072, 2048, 1536, 1024]
for hr in sweep:
    r = bench.comparator_handle(CUVS_AMD_PQ_HEAD_ROWS=hr)

    def step():
        ivf_pq.search(sp, index, queries, kk, neighbors=ci, distances=cd, resources=r)
        refine(data, queries, ci, indices=oi, distances=od, metric="sqeuclidean", resources=r)

    for _ in range(3):
        step()
    lib().cuvsAmdProfileEnable(1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    lib().cuvsAmdProfileEnable(0)
    ph = {}
    for nm in (b"pq_head_kernel", b"pq_filter_kernel", b"pq_rescore_kernel", b"pq_scan_kernel"):
        v = C.c_double(0)
        lib().cuvsAmdProfileCollect(nm, C.byref(v))
        ph[nm.decode()] = round(v.value / 10, 3)
    same = None
    if ref_i is None:
        ref_i, ref_d = ci.clone(), cd.clone()
    else:
        same = bool(torch.equal(ref_i, ci) and torch.equal(ref_d, cd))
    rs = bench.comparator_handle(CUVS_AMD_PQ_HEAD_ROWS=hr, CUVS_AMD_SCAN_DEBUG=1024)
    old = os.dup(2); dn = os.open(os.devnull, os.O_WRONLY); os.dup2(dn, 2)
    try:
        ivf_pq.search(sp, index, queries, kk, neighbors=ci, distances=cd, resources=rs); rs.sync()
    finally:
        os.dup2(old, 2); os.close(dn); os.close(old)
    st = (C.c_uint64 * 6)()
    lib().cuvsAmdIvfPqLastFilterStats6(st)
    print(f"head_rows {hr:5d}: {dt * 1e3:.3f} ms per step, phases {ph}, survivors {st[1]}, overflow {st[5]}, handed back {st[4]}, same_as_whole_list {same}", flush=True)
