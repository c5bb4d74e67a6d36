#!/usr/bin/env python3
"""Build the bench.py IVF-PQ index once, then time the search under several scan-kernel settings
(environment switches CUVS_AMD_SCAN_DEBUG, CUVS_AMD_PQ_HEAD_PROBES, CUVS_AMD_PQ_SCAN2, CUVS_AMD_PQ_QCAP: read when a handle
is created, so every variant gets its own handle). Prints one line per variant:
search ms (k*refine candidates), pq_scan_kernel ms per search, and whether the results equal those of the first
variant with the same LUT / accumulator types (LUT=f16|f32|u8, ACC=f16|f32; default f16/f16).

  python scripts/pq_scan_variants.py [--rows N --n-lists L --n-probes P] "DBG=8,HEAD=0" "DBG=0,HEAD=0" "DBG=0,HEAD=1,LUT=f16,ACC=f32"
"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

os.environ["CUVS_AMD_DEBUG_SWITCHES"] = "1"  # the gate in front of the switches below (bench.py itself runs without it)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--n-lists", type=int, default=16384)
    ap.add_argument("--n-probes", type=int, default=128)
    ap.add_argument("--batch", type=int, default=10000)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--metric", default="sqeuclidean")
    ap.add_argument("variants", nargs="+")
    args = ap.parse_args()

    import cuvs_amd
    from cuvs_amd._lib import lib
    from cuvs_amd.neighbors import ivf_pq

    dev = torch.device("cuda", 0)
    res = cuvs_amd.common.Resources()
    data = bench.gen_rows(args.rows, 128, seed=1234, device=dev)
    queries = bench.gen_rows(args.batch, 128, seed=4321, device=dev)
    index = ivf_pq.build(ivf_pq.IndexParams(n_lists=args.n_lists, metric=args.metric, pq_dim=64, pq_bits=8, kmeans_n_iters=20,
                                            kmeans_trainset_fraction=0.02), data, resources=res)
    res.sync()
    del data
    dt = {"f16": np.float16, "f32": np.float32, "u8": np.uint8}
    nb = torch.empty((args.batch, args.k), dtype=torch.int64, device=dev)
    ds = torch.empty((args.batch, args.k), dtype=torch.float32, device=dev)
    firsts = {}
    keys = {"DBG": "CUVS_AMD_SCAN_DEBUG", "HEAD": "CUVS_AMD_PQ_HEAD_PROBES", "S2": "CUVS_AMD_PQ_SCAN2", "QCAP": "CUVS_AMD_PQ_QCAP",
            "S3": "CUVS_AMD_PQ_SCAN3", "SCAP": "CUVS_AMD_PQ3_SURV_CAP", "F4": "CUVS_AMD_PQ_FILTER4"}
    for v in args.variants:
        lut, acc = "f16", "f16"
        for name in keys.values():
            os.environ.pop(name, None)
        for kv in v.split(","):
            key, val = kv.split("=")
            if key == "LUT":
                lut = val
            elif key == "ACC":
                acc = val
            else:
                os.environ[keys[key]] = val
        if lut == "f32":
            acc = "f32"
        res = cuvs_amd.common.Resources()  # reads the switches
        sp = ivf_pq.SearchParams(n_probes=args.n_probes, lut_dtype=dt[lut], internal_distance_dtype=dt[acc],
                                 max_internal_batch_size=args.batch)
        first = firsts.get((lut, acc))
        for _ in range(2):
            ivf_pq.search(sp, index, queries, args.k, neighbors=nb, distances=ds, resources=res)
        res.sync()
        lib().cuvsAmdProfileEnable(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ivf_pq.search(sp, index, queries, args.k, neighbors=nb, distances=ds, resources=res)
        res.sync()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / args.steps
        lib().cuvsAmdProfileEnable(0)
        scan = C.c_double(0)
        n = lib().cuvsAmdProfileCollect(b"pq_scan_kernel", C.byref(scan))
        flt, rsc = C.c_double(0), C.c_double(0)
        lib().cuvsAmdProfileCollect(b"pq_filter_kernel", C.byref(flt))
        lib().cuvsAmdProfileCollect(b"pq_rescore_kernel", C.byref(rsc))
        hd, bp = C.c_double(0), C.c_double(0)
        lib().cuvsAmdProfileCollect(b"pq_head_kernel", C.byref(hd))
        lib().cuvsAmdProfileCollect(b"pq_bprep_kernel", C.byref(bp))
        cur = (nb.clone(), ds.clone())
        same = "ref" if first is None else str(bool(torch.equal(cur[0], first[0]) and torch.equal(cur[1], first[1])))
        if first is None:
            firsts[(lut, acc)] = cur
        print(f"{v:24s} search {ms:8.3f} ms  scan {scan.value / args.steps:8.3f} ms ({n // args.steps} launches)  "
              f"bprep {bp.value / args.steps:6.3f} filter {flt.value / args.steps:7.3f} rescore {rsc.value / args.steps:7.3f} head+handback {hd.value / args.steps:7.3f}  same_as_first={same}", flush=True)


if __name__ == "__main__":
    main()
