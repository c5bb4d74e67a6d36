#!/usr/bin/env python3
"""Host timeline (CUVS_AMD_SCAN_DEBUG=8192) of ivf_flat.search at the C2 shape: where the wall time of a call goes."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    import cuvs_amd
    from cuvs_amd.neighbors import ivf_flat

    dev = torch.device("cuda", 0)
    res = cuvs_amd.common.Resources()
    data = bench.gen_rows(10_000_000, 128, seed=1234, device=dev)
    queries = bench.gen_rows(10000, 128, seed=4321, device=dev)
    fl = ivf_flat.build(ivf_flat.IndexParams(n_lists=4096, kmeans_n_iters=10, kmeans_trainset_fraction=0.05), data, resources=res)
    res.sync()
    k = 10
    nb = torch.empty((10000, k), dtype=torch.int64, device=dev)
    ds = torch.empty((10000, k), dtype=torch.float32, device=dev)
    fp = ivf_flat.SearchParams(n_probes=64)
    for _ in range(3):
        ivf_flat.search(fp, fl, queries, k, neighbors=nb, distances=ds, resources=res)
    torch.cuda.synchronize()
    os.environ["CUVS_AMD_SCAN_DEBUG"] = "8192"
    res2 = cuvs_amd.common.Resources()
    for r in (res2, res2, res2, res2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ivf_flat.search(fp, fl, queries, k, neighbors=nb, distances=ds, resources=r)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        print(f"call {(t1 - t0) * 1e3:.3f} ms, + drain {(time.perf_counter() - t1) * 1e3:.3f} ms", flush=True)
    t0 = time.perf_counter()
    for _ in range(4):
        ivf_flat.search(fp, fl, queries, k, neighbors=nb, distances=ds, resources=r)
    torch.cuda.synchronize()
    print(f"back-to-back {(time.perf_counter() - t0) * 250:.3f} ms per search", flush=True)


if __name__ == "__main__":
    main()
