#!/usr/bin/env python3
"""Host time to ENQUEUE one search (call returns, nothing synchronised) against its wall time with a synchronise, for
IVF-PQ, IVF-Flat and the shard all-gather + merge; and the cost of one pool allocation / free pair (cuvsRMMAlloc)."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import cuvs_amd
    from cuvs_amd._lib import lib
    from cuvs_amd.neighbors import ivf_flat, ivf_pq, ivf_pq_sharded as sh

    dev = torch.device("cuda", 0)
    res = cuvs_amd.common.Resources()
    rows, nl = 10_000_000, 4096
    data = bench.gen_rows(rows, 128, seed=1234, device=dev)
    queries = bench.gen_rows(10000, 128, seed=4321, device=dev)
    pq = ivf_pq.build(ivf_pq.IndexParams(n_lists=nl, pq_dim=64, pq_bits=8, kmeans_n_iters=10, kmeans_trainset_fraction=0.05),
                      data, resources=res)
    fl = ivf_flat.build(ivf_flat.IndexParams(n_lists=nl, kmeans_n_iters=10, kmeans_trainset_fraction=0.05), data, resources=res)
    res.sync()
    k = 20
    nb = torch.empty((10000, k), dtype=torch.int64, device=dev)
    ds = torch.empty((10000, k), dtype=torch.float32, device=dev)
    mi, md = torch.empty_like(nb), torch.empty_like(ds)
    comm = sh.ShardComm(0, 1, sh.ShardComm.unique_id(), res)
    sp = ivf_pq.SearchParams(n_probes=128, lut_dtype=np.float16, internal_distance_dtype=np.float32, max_internal_batch_size=10000)
    fp = ivf_flat.SearchParams(n_probes=64)
    calls = {
        "ivf_pq.search": lambda: ivf_pq.search(sp, pq, queries, k, neighbors=nb, distances=ds, resources=res),
        "ivf_flat.search": lambda: ivf_flat.search(fp, fl, queries, k, neighbors=nb, distances=ds, resources=res),
        "all_gather_topk": lambda: comm.all_gather_topk(ds, nb, out=(md, mi), resources=res),
    }
    for name, fn in calls.items():
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        enq, tot = [], []
        for _ in range(10):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            enq.append((t1 - t0) * 1e3)
            tot.append((t2 - t0) * 1e3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        b2b = (time.perf_counter() - t0) * 100
        print(f"{name:18s} enqueue {np.median(enq):7.3f} ms  enqueue+drain {np.median(tot):7.3f} ms  back-to-back {b2b:7.3f} ms", flush=True)
    L = lib()
    L.cuvsRMMAlloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p), C.c_size_t]
    L.cuvsRMMFree.argtypes = [C.c_size_t, C.c_void_p, C.c_size_t]
    h = res.get_c_obj()
    for size in (4096, 1 << 20, 1 << 26):
        p = C.c_void_p()
        for _ in range(3):
            L.cuvsRMMAlloc(h, C.byref(p), size); L.cuvsRMMFree(h, p, size)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            L.cuvsRMMAlloc(h, C.byref(p), size); L.cuvsRMMFree(h, p, size)
        t1 = time.perf_counter()
        print(f"pool alloc + free of {size:9d} B: {(t1 - t0) / 200 * 1e6:7.1f} us", flush=True)
    # a mix as in a search: 20 live buffers, freed in reverse
    ps = [C.c_void_p() for _ in range(20)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        for i, p in enumerate(ps):
            L.cuvsRMMAlloc(h, C.byref(p), (i + 1) << 18)
        for p in reversed(ps):
            L.cuvsRMMFree(h, p, 0)
    print(f"20 allocations + 20 frees: {(time.perf_counter() - t0) / 50 * 1e3:7.3f} ms", flush=True)
    comm.close()


if __name__ == "__main__":
    main()
