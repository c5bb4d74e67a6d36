#!/usr/bin/env python3
"""Where does the list-sharded search path spend its extra time on ONE rank? Builds the bench.py IVF-PQ index through the
sharded build (model + extend in chunks) and through the plain build, then times the same search: plain index, sharded
index without a communicator, with the one-rank RCCL communicator attached (coarse search sharded / replicated), and with
the all-gather + merge of the results. Prints wall ms per search and the HIP-event time of every profiled phase."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

NAMES = (b"pq_scan_kernel", b"pq_head_kernel", b"pq_filter_kernel", b"pq_rescore_kernel", b"shard_all_reduce",
         b"shard_all_gather_probes", b"shard_all_gather")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=100_000_000)
    ap.add_argument("--n-lists", type=int, default=16384)
    ap.add_argument("--n-probes", type=int, default=128)
    ap.add_argument("--batch", type=int, default=10000)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--skip-plain", action="store_true")
    args = ap.parse_args()

    import cuvs_amd
    from cuvs_amd._lib import lib
    from cuvs_amd.neighbors import ivf_pq, ivf_pq_sharded as sh

    dev = torch.device("cuda", 0)
    res = cuvs_amd.common.Resources()
    data = bench.gen_rows(args.rows, 128, seed=1234, device=dev)
    queries = bench.gen_rows(args.batch, 128, seed=4321, device=dev)
    kw = dict(n_lists=args.n_lists, pq_dim=64, pq_bits=8, kmeans_n_iters=20, kmeans_trainset_fraction=0.02)
    plain = None if args.skip_plain else ivf_pq.build(ivf_pq.IndexParams(**kw), data, resources=res)
    sidx = sh.build(ivf_pq.IndexParams(add_data_on_build=False, **kw), data, 0, 1, resources=res)
    for r0 in range(0, args.rows, 1 << 24):
        r1 = min(args.rows, r0 + (1 << 24))
        sh.extend(sidx, data[r0:r1], torch.arange(r0, r1, dtype=torch.int64, device=dev), resources=res)
    res.sync()
    del data
    comm = sh.ShardComm(0, 1, sh.ShardComm.unique_id(), res)
    sp = ivf_pq.SearchParams(n_probes=args.n_probes, lut_dtype=np.float16, internal_distance_dtype=np.float32,
                             max_internal_batch_size=args.batch)
    nb = torch.empty((args.batch, args.k), dtype=torch.int64, device=dev)
    ds = torch.empty((args.batch, args.k), dtype=torch.float32, device=dev)
    mi, md = torch.empty_like(nb), torch.empty_like(ds)
    ref = None

    def run(label, index, r, gather):
        nonlocal ref

        def step():
            ivf_pq.search(sp, index, queries, args.k, neighbors=nb, distances=ds, resources=r)
            if gather:
                comm.all_gather_topk(ds, nb, out=(md, mi), resources=r)

        for _ in range(2):
            step()
        r.sync()
        lib().cuvsAmdProfileEnable(1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        r.sync()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / args.steps
        lib().cuvsAmdProfileEnable(0)
        parts = []
        for nm in NAMES:
            v = C.c_double(0)
            n = lib().cuvsAmdProfileCollect(nm, C.byref(v))
            if n:
                parts.append(f"{nm.decode()} {v.value / args.steps:.3f}")
        # the same search without profiling events
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        r.sync()
        torch.cuda.synchronize()
        ms2 = (time.perf_counter() - t0) * 1e3 / args.steps
        cur = nb.clone()
        same = "ref" if ref is None else str(bool(torch.equal(cur, ref)))
        if ref is None:
            ref = cur
        print(f"{label:44s} {ms:7.3f} ms (no events {ms2:7.3f})  same={same}  " + ", ".join(parts), flush=True)

    if plain is not None:
        run("plain build", plain, res, False)
    run("sharded build, no communicator", sidx, res, False)
    sh.attach_comm(sidx, comm)
    run("communicator attached", sidx, res, False)
    run("communicator attached + result all-gather", sidx, res, True)
    os.environ["CUVS_AMD_SHARD_COARSE_REPLICATED"] = "1"
    res2 = cuvs_amd.common.Resources()
    run("communicator attached, replicated coarse", sidx, res2, False)
    sh.attach_comm(sidx, None)
    run("sharded build, no communicator (again)", sidx, res, False)
    comm.close()


if __name__ == "__main__":
    main()
