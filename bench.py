#!/usr/bin/env python3
"""bench.py — QPS of the IVF-PQ search hot path on MI355X (BASELINE.json metric).

Default workload (configs[2], the one the metric is quoted on): IVF-PQ over 100M x 128 fp32 synthetic vectors,
pq_dim=64, pq_bits=8, n_lists=16384, n_probes=128, batch = 10k queries, k = 10, one GPU.
A "step" is one cuvsIvfPqSearch call over one resident batch of 10k queries. Index build, ground truth and
the CPU baseline are outside the timed region; queries/outputs are resident in HBM.

  python bench.py                      # N=1
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N   # replicated index, query-parallel

Prints ONE JSON line (rank 0). Extra keys: recall@10, roofline (dominant kernel = pq_scan_kernel, timed
with HIP events on its launch stream), cpu_baseline (reference CPU path restated in oracle/, bounded sample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def gen_rows(n, dim, seed, device, chunk=1 << 22, latent=32, n_modes=65536):
    """Synthetic corpus with low intrinsic dimension (a 32-d Gaussian mixture embedded in R^dim + small noise):
    isotropic 128-d clusters would make every neighbour equidistant and recall meaningless."""
    g = torch.Generator(device=device).manual_seed(1234)  # structure shared by data and queries
    A = torch.randn(latent, dim, generator=g, device=device) / latent ** 0.5
    modes = torch.randn(n_modes, latent, generator=g, device=device)
    g2 = torch.Generator(device=device).manual_seed(seed)
    out = torch.empty((n, dim), dtype=torch.float32, device=device)
    for r0 in range(0, n, chunk):
        c = min(chunk, n - r0)
        which = torch.randint(0, n_modes, (c,), generator=g2, device=device)
        z = modes[which] + 0.35 * torch.randn(c, latent, generator=g2, device=device)
        out[r0:r0 + c] = z @ A + 0.03 * torch.randn(c, dim, generator=g2, device=device)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--rows", type=int, default=100_000_000, help="dataset rows (default = BASELINE config)")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--n-lists", type=int, default=16384)
    ap.add_argument("--n-probes", type=int, default=128)
    ap.add_argument("--pq-dim", type=int, default=64)
    ap.add_argument("--batch", type=int, default=10000)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--lut", choices=["f32", "f16"], default="f16",
                    help="LUT / internal distance dtype (both are reference search_params settings)")
    ap.add_argument("--refine-ratio", type=int, default=2,
                    help="IVF-PQ returns ratio*k candidates that cuvsRefine re-ranks exactly (reference bench grids "
                         "use refine_ratio 1..4, python/cuvs_bench/.../cuvs_ivf_pq.yaml); 1 disables refinement")
    ap.add_argument("--trainset-fraction", type=float, default=0.02)
    ap.add_argument("--gt-queries", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    import cuvs_amd
    from cuvs_amd._lib import lib
    from cuvs_amd.neighbors import brute_force, ivf_pq, refine

    res = cuvs_amd.common.Resources()

    # ------------------------------------------------------------------ data + index (untimed)
    t0 = time.time()
    data = gen_rows(args.rows, args.dim, seed=1234, device=dev)
    queries = gen_rows(args.batch, args.dim, seed=4321 + rank, device=dev)
    torch.cuda.synchronize()
    log(f"generated {args.rows}x{args.dim} fp32 in {time.time() - t0:.1f}s")
    t0 = time.time()
    ip = ivf_pq.IndexParams(n_lists=args.n_lists, metric="sqeuclidean", pq_dim=args.pq_dim, pq_bits=8,
                            kmeans_n_iters=20, kmeans_trainset_fraction=args.trainset_fraction)
    index = ivf_pq.build(ip, data, resources=res)
    res.sync()
    build_s = time.time() - t0
    log(f"built IVF-PQ index in {build_s:.1f}s")
    lut_np = np.float16 if args.lut == "f16" else np.float32
    sp = ivf_pq.SearchParams(n_probes=args.n_probes, lut_dtype=lut_np, internal_distance_dtype=lut_np,
                             max_internal_batch_size=args.batch)
    neighbors = torch.empty((args.batch, args.k), dtype=torch.int64, device=dev)
    distances = torch.empty((args.batch, args.k), dtype=torch.float32, device=dev)

    kk = args.k * max(1, args.refine_ratio)
    cand_i = torch.empty((args.batch, kk), dtype=torch.int64, device=dev)
    cand_d = torch.empty((args.batch, kk), dtype=torch.float32, device=dev)

    def step():
        if args.refine_ratio > 1:
            ivf_pq.search(sp, index, queries, kk, neighbors=cand_i, distances=cand_d, resources=res)
            refine(data, queries, cand_i, indices=neighbors, distances=distances, metric="sqeuclidean", resources=res)
        else:
            ivf_pq.search(sp, index, queries, args.k, neighbors=neighbors, distances=distances, resources=res)

    # ------------------------------------------------------------------ timed region
    for _ in range(args.warmup):
        step()
    lib().cuvsAmdProfileEnable(1)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t_start
    lib().cuvsAmdProfileEnable(0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    scan_ms = C.c_double(0)
    n_launch = lib().cuvsAmdProfileCollect(b"pq_scan_kernel", C.byref(scan_ms))

    # ------------------------------------------------------------------ recall@10 vs exact search (untimed)
    ng = min(args.gt_queries, args.batch)
    bf = brute_force.build(data, metric="sqeuclidean", resources=res)
    _, gt = brute_force.search(bf, queries[:ng], args.k, resources=res)
    res.sync()
    found, truth = neighbors[:ng].cpu().numpy(), gt.cpu().numpy()
    recall = float(np.mean([len(np.intersect1d(f, t)) for f, t in zip(found, truth)])) / args.k
    del bf

    # ------------------------------------------------------------------ roofline of the dominant kernel
    # algorithmic bytes per launch = sum over (query, probe) pairs of list_len * code bytes
    sizes = index.list_sizes.to(torch.int64)
    centers = index.centers
    cn = (centers * centers).sum(1)
    probe_bytes = 0
    for q0 in range(0, args.batch, 2048):
        qq = queries[q0:q0 + 2048]
        dmat = cn[None, :] - 2.0 * (qq @ centers.T)
        pr = torch.topk(dmat, min(args.n_probes, args.n_lists), dim=1, largest=False).indices
        probe_bytes += int(sizes[pr].sum().item()) * (args.pq_dim * 8 // 8)
    # one search = a small head launch (nearest probe of every query) + the tail launch; both are the same kernel,
    # so bytes and time are averaged over all its launches (sum of bytes / sum of time)
    per_step = max(n_launch, 1) / max(args.steps, 1)
    bytes_per_launch = probe_bytes / per_step
    avg_ms = scan_ms.value / max(n_launch, 1)
    achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic = None
    pmc_busy = {}
    tfile = os.path.join(ROOT, "profiles", "r01b_pq_scan_traffic.json")
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            # PMC pass (profiles/README.md): HBM bytes of the kernel's launches of one search, averaged per launch
            traffic = tj.get("hbm_bytes_per_step", tj.get("hbm_bytes_per_launch"))
            traffic = int(traffic / per_step) if traffic else None
            pmc_busy = {k: tj[k] for k in ("valu_busy", "lds_busy", "lds_bank_conflict_share", "tcc_hit_rate") if k in tj}
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": "pq_scan_kernel", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes_per_launch": int(bytes_per_launch), "avg_launch_ms": round(avg_ms, 3),
                "launches": n_launch, "launches_per_step": per_step, "pmc": pmc_busy,
                "algorithmic_bytes_per_step": probe_bytes, "kernel_ms_per_step": round(avg_ms * per_step, 3),
                "note": "logical code bytes scanned per launch / HIP-event kernel time. The list-major schedule serves every "
                        "list byte fetched from HBM ~45 times from L2 (traffic = measured HBM bytes), so the figure exceeds "
                        "the HBM peak by design; PMC (pmc): neither VALU nor LDS is saturated, the rest is per-item "
                        "serial phases (DESIGN.md 3.1)"}

    # ------------------------------------------------------------------ CPU baseline (rank 0, N=1 only)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle

        sample_rows = min(args.rows, 2_000_000)
        sample_q = 64
        xs = data[:sample_rows].cpu().numpy()
        qs = queries[:sample_q].cpu().numpy()
        t0 = time.perf_counter()
        oracle.exact_knn(qs, xs, args.k)
        dt = time.perf_counter() - t0
        qps_full = sample_q / dt * (sample_rows / args.rows)
        cpu = {"value": round(qps_full, 3), "unit": "queries/s", "cores": oracle.num_threads(), "kind": "port",
               "sample": f"exact kNN (refine_host restatement, OpenMP) of {sample_q} queries over the first "
                         f"{sample_rows} rows took {dt:.2f}s; scaled linearly to {args.rows} rows"}

    if rank == 0:
        total_q = args.batch * args.steps * world
        out = {
            "metric": "QPS @ recall@10>=0.9, 100Mx128 fp32 IVF-PQ, batch=10k",
            "value": round(total_q / elapsed, 1),
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8 codes, " + ("f16" if args.lut == "f16" else "f32") + " LUT/score",
            "data": "synthetic",
            "config": {"workload": f"IVF-PQ {args.rows}x{args.dim} fp32, pq_dim={args.pq_dim} pq_bits=8 "
                                   f"n_lists={args.n_lists} n_probes={args.n_probes} batch={args.batch} k={args.k}",
                       "parallelism": "replicated index, queries split across ranks" if world > 1 else "single GPU",
                       "lut_dtype": args.lut, "refine_ratio": args.refine_ratio, "build_seconds": round(build_s, 1)},
            "recall_at_10": round(recall, 4),
            "roofline": roofline,
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
