#!/usr/bin/env python3
"""bench.py — QPS of the IVF-PQ search hot path on MI355X (BASELINE.json metric).

Default workload (configs[2], the one the metric is quoted on): IVF-PQ over 100M x 128 fp32 synthetic vectors,
pq_dim=64, pq_bits=8, n_lists=16384, n_probes=128, batch = 10k queries, k = 10, one GPU.
A "step" is one cuvsIvfPqSearch (k * refine_ratio candidates) + cuvsRefine over one resident batch of 10k queries.
Index build, ground truth and the CPU baseline are outside the timed region; queries/outputs are resident in HBM.

  python bench.py                      # N=1: headline + the other LUT/score precisions + C1/C2/C4 lines + live PMC passes
  python bench.py --gpus N             # N > 1 with no launcher around it: bench.py starts its own N ranks (one per GPU,
                                       # torch.distributed.run on 127.0.0.1) and fails loudly when the node has fewer than N
                                       # devices. List-sharded index (lists dealt to the ranks by size), every step ends in
                                       # ONE native RCCL all-gather of the per-rank [Q, k] blocks (include/cuvs_amd/shard.h)
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N
                                       # the same under an external launcher (the driver's form)
  python bench.py --gpus N --share-devices
                                       # FUNCTIONAL run of the N-rank path on fewer than N devices (ranks share GPUs, the
                                       # collectives go through the communicator's host-staged transport): not a scaling
                                       # figure - the line says so ("transport", "oversubscribed")
  [torchrun ...] bench.py --config c5 [--rows R]
                                       # BASELINE configs[4]: IVF-PQ rows x 96 int8 (default 1B), list shards dealt by
                                       # size over the ranks, rows generated chunk by chunk (no rank holds the corpus)

Prints ONE JSON line (rank 0):
  value / ms_per_step    the headline variant: fp16 LUT / fp32 scores, the arithmetic of the reference's own bench grid
  config.variants        the same step with the reference-default arithmetic (fp32 LUT / fp32 scores), fp16 / fp16 and
                         the fp8 LUT, each with ms, recall and scan-kernel time
  config.metric_variants the same search on an inner-product and a cosine index of the same rows
  roofline               the scan kernels of one search (head-phase list scan, matrix-core filter, re-score): HIP-event
                         times per kernel, the logical scan rate of SURVEY 8d (code bytes / kernel time) AND the
                         physical fractions from rocprofv3 PMC passes of this same workload: hbm_frac, lds_busy,
                         valu_busy, mfma_busy; `bound` names the busiest pipe, `frac` its busy fraction
  extra                  C1 (brute force 100k x 128), C2 (IVF-Flat 10M x 128), C4 (CAGRA 10M x 768 fp16) - recall,
                         ms, kernel time, fraction of the roofline that bounds each
  cpu_baseline           the reference's CPU exact-search path (refine_host restatement, OpenMP) at the C1 shape
"""
import argparse
import csv
import ctypes as C
import glob
import json
import os
import shutil
import signal
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The comparator handles of this script (LUT-scan check, pruning-off figure, filter counters) select kernels through
# CUVS_AMD_* switches, which the library only looks at behind the gate CUVS_AMD_DEBUG_SWITCHES=1 and only when a handle is
# created: comparator_handle() sets gate + switches for exactly that moment. Every timed handle is created without the gate:
# the production configuration.
os.environ.pop("CUVS_AMD_DEBUG_SWITCHES", None)


def comparator_handle(**switches):
    """a cuvs_amd Resources handle created under the given CUVS_AMD_* switches (and the gate that makes the library read them)"""
    import cuvs_amd

    env = {"CUVS_AMD_DEBUG_SWITCHES": "1", **{k: str(v) for k, v in switches.items()}}
    os.environ.update(env)
    try:
        return cuvs_amd.common.Resources()
    finally:
        for k in env:
            os.environ.pop(k, None)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_TFLOPS = 157.3  # fp32 MFMA peak (spec)
N_CU, N_SIMD = 256, 1024


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks (one process per GPU) under
    torch.distributed.run on the loopback address and pass their output through. The reference starts all its ranks from
    one process too (cpp/src/neighbors/mg/snmg.cuh:283-341, c/include/cuvs/neighbors/mg_ivf_pq.h:152-190)."""
    import socket

    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and not args.share_devices:
        print(f"[bench] --gpus {args.gpus} needs {args.gpus} devices, this node has {n_dev}: refusing to run fewer ranks than asked "
              f"for (a functional run of the {args.gpus}-rank path on shared devices: add --share-devices)", file=sys.stderr, flush=True)
        sys.exit(2)
    if n_dev < 1:
        print("[bench] no GPU visible", file=sys.stderr, flush=True)
        sys.exit(2)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("starting", args.gpus, "ranks:", " ".join(cmd[2:]))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def dist_setup(args):
    """rank / world / device of this process and the control-plane process group (barriers, the max over ranks, the id
    rendezvous; the data path never touches torch.distributed). One rank per GPU; with --share-devices rank r uses device
    r % n_devices, the control plane runs over gloo and the shard communicator over its host-staged transport (RCCL
    refuses two ranks on one device)."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(1, args.gpus):
        print(f"[bench] --gpus {args.gpus} but the launcher started {world} ranks", file=sys.stderr, flush=True)
        sys.exit(2)
    n_dev = torch.cuda.device_count()
    if n_dev < 1 or (local_rank >= n_dev and not args.share_devices):
        print(f"[bench] rank {rank}: local rank {local_rank} has no device of its own ({n_dev} visible); --share-devices runs the ranks "
              f"on shared devices (functional run only)", file=sys.stderr, flush=True)
        sys.exit(2)
    shared = args.share_devices and world > n_dev
    torch.cuda.set_device(local_rank % n_dev)
    dev = torch.device("cuda", local_rank % n_dev)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    return rank, world, dev, dist, shared


def ctl_tensor(values, dtype, dev, shared):
    """a small tensor for a control-plane collective: on the device under nccl, on the host under gloo"""
    return torch.tensor(values, dtype=dtype, device="cpu" if shared else dev)


_REAL_STDOUT = None


def capture_stdout():
    """From here on fd 1 of this process IS stderr: libraries write to it from C / C++ (RCCL's start-up banner through C stdio, flushed
    whenever; gloo's "[Gloo] Rank 0 is connected to ..." notes) - on every rank, and a launcher merges the ranks' stdout. The ONE JSON
    line goes to the saved descriptor (emit_json)."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit_json(obj):
    sys.stdout.flush()
    try:
        C.CDLL(None).fflush(None)
    except Exception:
        pass
    line = (json.dumps(obj) + "\n").encode()
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, line)


def flush_c_stdio_to_stderr():
    """RCCL writes its start-up banner through C stdio whenever it initialises (a buffered write that surfaces at the next flush or at
    exit): stdout carries ONE JSON line, so whatever C stdio holds is flushed with fd 1 pointing at stderr"""
    try:
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)
        try:
            C.CDLL(None).fflush(None)
        finally:
            os.dup2(keep, 1)
            os.close(keep)
    except Exception:
        pass


def make_comm(sh, res, world, shared):
    # (RCCL prints a version banner through C stdio when its first communicator is made: stdout carries ONE JSON line, so fd 1 points
    # at stderr while that happens and until the buffer is flushed)
    sys.stdout.flush()
    keep = os.dup(1)
    os.dup2(2, 1)
    try:
        if world > 1:
            return sh.ShardComm.from_torch(res, host_staged=shared)
        return sh.ShardComm(0, 1, sh.ShardComm.unique_id(), res)
    finally:
        sys.stdout.flush()
        C.CDLL(None).fflush(None)  # (the banner sits in C stdio's buffer: it is written - to stderr - now)
        os.dup2(keep, 1)
        os.close(keep)


def cpu_baseline_line(c1_x=None, c1_q=None, dev=None):
    """The reference's only CPU search code (refine_host: exact distances of candidate rows, OpenMP over queries) restated in
    oracle/ with every row as a candidate, at SURVEY 8d's C1 shape, on this box's host cores. A reported baseline, never
    part of the product path. Bounded: 5 runs of ~0.2 s."""
    import oracle

    oracle.set_num_threads(0)  # every host core (torch.distributed.run pins OMP_NUM_THREADS=1 for its workers)
    if c1_x is None:
        c1_x = gen_rows(100_000, 128, 1234, dev).cpu().numpy()
        c1_q = gen_rows(1000, 128, 4321, dev).cpu().numpy()
    for _ in range(2):  # (page faults, thread start-up, clocks)
        oracle.exact_knn(c1_q, c1_x, 10)
    ts, t_all = [], time.perf_counter()
    while len(ts) < 15 or (time.perf_counter() - t_all < 10.0 and len(ts) < 200):  # >= 15 runs and ~10 s of CPU work, bounded
        t0 = time.perf_counter()
        oracle.exact_knn(c1_q, c1_x, 10)
        ts.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all > 30.0:
            break
    med, best = float(np.median(ts)), float(np.min(ts))
    return {"value": round(1000 / med, 1), "unit": "queries/s", "cores": oracle.num_threads(), "kind": "port",
            "value_best_run": round(1000 / best, 1), "runs": len(ts),
            "gflops": round(2 * 1000 * 100_000 * 128 / med / 1e9, 1),
            "sample": f"C1 shape (SURVEY 8d): exact kNN of 1000 queries over 100000x128 fp32, k=10, the reference's "
                      f"refine_host arithmetic restated in oracle/ (OpenMP, static schedule over the queries), {len(ts)} runs after 2 "
                      f"warm-ups: median {med * 1e3:.1f} ms (value), best {best * 1e3:.1f} ms"}


SPEC_GHZ, MFMA_F16_TFLOPS = 2.4, 2500.0  # MI355X_MICROARCH.md: peak engine clock, dense fp16 MFMA peak


def pq_scan_roofline(index, queries_f32, n_probes, owned, code_bytes, rot_dim, filter_ms, scan_ms, n_launch, steps, phase_ms,
                     early_stop_off_ms=None):
    """Roofline block of the dominant kernel (pq_filter_kernel, one launch per search) of THIS rank's search.
    Algorithmic work of one search, from the index's list sizes and an independent coarse ranking of the batch:
      logical bytes  = sum over (query, probe) pairs of list_len * code bytes (the reference reads a list once per pair)
      unique bytes   = code bytes of every list probed by at least one pair (the lower bound on HBM bytes per batch)
      useful flop    = 2 * rot_dim per (row, query) pair of the tail phase (the screen is a GEMM of decoded rows x residuals)
      gather cycles  = LDS cycles of the decode if no two lanes ever met in a bank: every 32-row subtile of a probed list
                       once per 128 probing queries, 32 ds_read_b32 x 2 cycles each
    `owned`: bool [n_lists], the lists this rank holds (a list shard scans only those). Its speed of light is the largest of
    three floors, each a spec peak: unique code bytes / 8 TB/s, useful fp16 MFMA flop / 2.5 PFLOP/s, conflict-free gather
    cycles / (256 CUs x 2.4 GHz). `frac` = that floor / the kernel's measured duration (HIP events around the launch on the
    handle's stream, this run); `bound` names it."""
    dev = queries_f32.device
    sizes = index.list_sizes.to(torch.int64)
    n_lists = sizes.shape[0]
    centers = index.centers.float()
    cn = (centers * centers).sum(1)
    probe_bytes = 0
    tail_pairs = torch.zeros(n_lists, dtype=torch.int64, device=dev)
    all_pairs = torch.zeros(n_lists, dtype=torch.int64, device=dev)
    for q0 in range(0, queries_f32.shape[0], 2048):
        qq = queries_f32[q0:q0 + 2048]
        dmat = cn[None, :] - 2.0 * (qq @ centers.T)
        pr = torch.topk(dmat, min(n_probes, n_lists), dim=1, largest=False).indices
        probe_bytes += int((sizes[pr] * owned[pr]).sum().item()) * code_bytes
        all_pairs += torch.bincount(pr.reshape(-1), minlength=n_lists)
        tail_pairs += torch.bincount(pr[:, 1:].reshape(-1), minlength=n_lists)  # the nearest probe is the head phase
    tail_pairs, all_pairs = tail_pairs * owned, all_pairs * owned
    unique_bytes = int((sizes * (all_pairs > 0)).sum().item()) * code_bytes
    tail_row_pairs = int((sizes * tail_pairs).sum().item())
    useful_flop = 2.0 * rot_dim * tail_row_pairs
    subtile_decodes = int((((sizes + 31) // 32) * ((tail_pairs + 127) // 128)).sum().item())
    gather_cycles = subtile_decodes * 32 * 2.0
    # one search = a small head launch (nearest probe of every query) + the tail launch: the same work, bytes and
    # time are averaged over all launches (sum of bytes / sum of time)
    per_step = max(n_launch, 1) / max(steps, 1)
    bytes_per_launch = probe_bytes / per_step
    avg_ms = scan_ms / max(n_launch, 1)
    logical = bytes_per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    f_ms = filter_ms
    floors_ms = {"hbm": unique_bytes / (HBM_PEAK_GBS * 1e9) * 1e3, "mfma": useful_flop / (MFMA_F16_TFLOPS * 1e12) * 1e3,
                 "lds": gather_cycles / (N_CU * SPEC_GHZ * 1e9) * 1e3}
    sol_bound = max(floors_ms, key=floors_ms.get)
    sol_frac = floors_ms[sol_bound] / f_ms if f_ms > 0 else None
    if sol_bound == "hbm":
        ach, peak, unit = unique_bytes / max(f_ms, 1e-9) / 1e6, HBM_PEAK_GBS, "GB/s"
    elif sol_bound == "mfma":
        ach, peak, unit = useful_flop / max(f_ms, 1e-9) / 1e9, MFMA_F16_TFLOPS, "TFLOP/s"
    else:
        ach, peak, unit = gather_cycles / max(f_ms, 1e-9) / 1e6, N_CU * SPEC_GHZ, "Gcycles/s"
    return {"bound": sol_bound, "kernel": "pq_filter_kernel (the tail phase's matrix-core screen: dominant kernel, one launch per search)",
            "achieved": round(ach, 1), "peak": peak, "unit": unit, "frac": None if sol_frac is None else round(sol_frac, 4),
            "traffic": None, "avg_launch_ms": round(f_ms, 3),
            "floors_ms": {k: round(v, 3) for k, v in floors_ms.items()},
            "frac_hbm_unique_bytes": round(floors_ms["hbm"] / f_ms, 4) if f_ms > 0 else None,
            "frac_mfma_useful_flop": round(floors_ms["mfma"] / f_ms, 4) if f_ms > 0 else None,
            "frac_lds_conflict_free_gathers": round(floors_ms["lds"] / f_ms, 4) if f_ms > 0 else None,
            "algorithmic": {"unique_code_bytes_per_search": unique_bytes, "logical_code_bytes_per_search": probe_bytes,
                            "tail_row_query_pairs": tail_row_pairs, "useful_mfma_flop": useful_flop,
                            "subtile_decodes": subtile_decodes, "conflict_free_gather_cycles": gather_cycles},
            "scan_kernels": {"names": "pq_head_kernel + pq_bprep_kernel + pq_filter_kernel + pq_rescore_kernel (+ hand-backs)",
                             "launches_per_step": per_step, "kernel_ms_per_step": round(avg_ms * per_step, 3),
                             "phase_ms_per_step": phase_ms, "logical_scan_gbs": round(logical, 1),
                             "frac_hbm_unique_bytes": round(unique_bytes / (HBM_PEAK_GBS * 1e9) / max(avg_ms * per_step * 1e-3, 1e-12), 4),
                             "early_stop_off_kernel_ms_per_step": early_stop_off_ms},
            "note": "frac = max(unique code bytes / 8 TB/s, useful fp16 MFMA flop / 2.5 PFLOP/s, conflict-free LDS gather cycles / "
                    "(256 CUs x 2.4 GHz)) / measured duration of pq_filter_kernel on rank 0: every term is a spec peak and an algorithmic "
                    "quantity computed from the index (this rank's lists) and the batch. logical_scan_gbs (SURVEY 8d: list bytes per pair) "
                    "exceeds the HBM peak by design - a list chunk is fetched once per up to 128 probing queries - and is no utilisation. "
                    "pmc.* (N=1 only) are busy cycles of each pipe over SPEC-clock cycles (2.4 GHz x kernel time), from rocprofv3 --pmc "
                    "passes of this workload"}


def gen_rows(n, dim, seed, device, chunk=1 << 22, latent=32, n_modes=65536, row0=0, out=None, spread=0.35):
    """Synthetic corpus with low intrinsic dimension (a `latent`-d Gaussian mixture embedded in R^dim + small noise):
    isotropic dim-d clusters would make every neighbour equidistant and recall meaningless. Rows are
    generated chunk by chunk from (seed, chunk index), so any row range can be produced on any rank."""
    g = torch.Generator(device=device).manual_seed(1234)  # structure shared by data and queries
    A = torch.randn(latent, dim, generator=g, device=device) / latent ** 0.5
    modes = torch.randn(n_modes, latent, generator=g, device=device)
    if out is None:
        out = torch.empty((n, dim), dtype=torch.float32, device=device)
    assert row0 % chunk == 0
    for r0 in range(0, n, chunk):
        c = min(chunk, n - r0)
        g2 = torch.Generator(device=device).manual_seed(seed * 1000003 + (row0 + r0) // chunk)
        which = torch.randint(0, n_modes, (c,), generator=g2, device=device)
        z = modes[which] + spread * torch.randn(c, latent, generator=g2, device=device)
        out[r0:r0 + c] = (z @ A + 0.03 * torch.randn(c, dim, generator=g2, device=device)).to(out.dtype)
    return out


def gen_rows_survey8d(n, dim, seed, device, n_lists, chunk=1 << 22):
    """SURVEY 8d's generator verbatim: 4 n_lists centres ~ U[-1, 1)^d, points = centre + N(0, 0.1^2 I) (isotropic in all d
    dimensions), counter-based per chunk. The centres come from the DATA seed (1234) for rows and queries alike - queries are
    held-out draws around the same centres."""
    g0 = torch.Generator(device=device)
    g0.manual_seed(1234)
    centres = torch.rand((4 * n_lists, dim), generator=g0, device=device) * 2.0 - 1.0
    out = torch.empty((n, dim), dtype=torch.float32, device=device)
    for c, r0 in enumerate(range(0, n, chunk)):
        r1 = min(n, r0 + chunk)
        g = torch.Generator(device=device)
        g.manual_seed(seed * 1_000_003 + c)
        a = torch.randint(0, centres.shape[0], (r1 - r0,), generator=g, device=device)
        out[r0:r1] = centres[a]
        out[r0:r1].add_(torch.randn((r1 - r0, dim), generator=g, device=device), alpha=0.1)
    return out


def gen_rows_gaussian(n, dim, seed, device, chunk=1 << 22):
    """N(0, I) in all d dimensions: no cluster structure at all - the corpus on which a bound from the nearest list prunes least."""
    out = torch.empty((n, dim), dtype=torch.float32, device=device)
    for c, r0 in enumerate(range(0, n, chunk)):
        r1 = min(n, r0 + chunk)
        g = torch.Generator(device=device)
        g.manual_seed(seed * 1_000_003 + c)
        out[r0:r1] = torch.randn((r1 - r0, dim), generator=g, device=device)
    return out


def exact_topk_fp64(data, q, k, chunk=500_000):
    """Ground truth independent of the library under test: squared L2 in float64 (torch), row chunks, running top-k."""
    qd = q.double()
    best_d = torch.full((q.shape[0], k), float("inf"), dtype=torch.float64, device=q.device)
    best_i = torch.zeros((q.shape[0], k), dtype=torch.int64, device=q.device)
    for r0 in range(0, data.shape[0], chunk):
        xc = data[r0:r0 + chunk].double()
        d2 = (xc * xc).sum(1)[None, :] - 2.0 * (qd @ xc.T)
        v, ii = torch.topk(d2, min(k, xc.shape[0]), dim=1, largest=False)
        cd, ci = torch.cat([best_d, v], 1), torch.cat([best_i, ii + r0], 1)
        o = torch.argsort(cd, dim=1)[:, :k]
        best_d, best_i = torch.gather(cd, 1, o), torch.gather(ci, 1, o)
        del xc, d2
    return best_i


def recall_of(found, truth):
    return float(np.mean([len(np.intersect1d(f, t)) for f, t in zip(found, truth)])) / truth.shape[1]


LUTS = {"f32": np.float32, "f16": np.float16, "fp8": np.uint8}


# ---------------------------------------------------------------------------------------------- PMC passes (rank 0)
PMC_SETS = [
    "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE",
    "FETCH_SIZE SQ_VALU_MFMA_BUSY_CYCLES",
    "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum",
]


def run_pmc_passes(child_args, n_search, timeout_s=240):
    """rocprofv3 --pmc (counters only, one group per run - MI355X_MICROARCH.md 'rocprofv3 PMC slots') over a child that
    rebuilds this workload and runs `n_search` searches; returns per-search sums for pq_scan_kernel or None."""
    if shutil.which("rocprofv3") is None:
        return None
    sums, fsums = {}, {}
    for i, counters in enumerate(PMC_SETS):
        d = tempfile.mkdtemp(prefix="bench_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc", *counters.split(), "--output-format", "csv", "-d", d, "-o", "pmc", "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--pmc-child", str(n_search), *child_args]
        env = dict(os.environ, TMPDIR="/tmp")
        try:
            p = subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=env, cwd="/tmp",
                                 start_new_session=True)
            try:
                p.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)
                log(f"PMC pass {i} timed out")
                return None
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                log(f"PMC pass {i}: rc={p.returncode}, no counter file")
                return None
            n_disp = 0
            for r in csv.DictReader(open(files[0])):
                if any(t in r["Kernel_Name"] for t in ("pq_scan", "pq_head", "pq_bprep", "pq_thr", "pq_filter", "pq_rescore", "pool_merge")):
                    sums[r["Counter_Name"]] = sums.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                    n_disp += 1
                if "pq_filter" in r["Kernel_Name"]:
                    fsums[r["Counter_Name"]] = fsums.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            if n_disp == 0:
                return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    out = {k: v / n_search for k, v in sums.items()}
    out["filter"] = {k: v / n_search for k, v in fsums.items()}
    return out


def pmc_child(args, n_search):
    """The profiled child: same data, same index, `n_search` searches of the headline variant."""
    import cuvs_amd
    from cuvs_amd.neighbors import ivf_pq

    dev = torch.device("cuda", 0)
    res = cuvs_amd.common.Resources()
    data = gen_rows(args.rows, args.dim, seed=1234, device=dev)
    queries = gen_rows(args.batch, args.dim, seed=4321, device=dev)
    index = ivf_pq.build(ivf_pq.IndexParams(n_lists=args.n_lists, pq_dim=args.pq_dim, pq_bits=8, kmeans_n_iters=20,
                                            kmeans_trainset_fraction=args.trainset_fraction), data, resources=res)
    del data
    sp = ivf_pq.SearchParams(n_probes=args.n_probes, lut_dtype=LUTS[args.lut], internal_distance_dtype=LUTS[args.acc],
                             max_internal_batch_size=args.batch)
    kk = args.k * max(1, args.refine_ratio)
    nb = torch.empty((args.batch, kk), dtype=torch.int64, device=dev)
    ds = torch.empty((args.batch, kk), dtype=torch.float32, device=dev)
    for _ in range(n_search):
        ivf_pq.search(sp, index, queries, kk, neighbors=nb, distances=ds, resources=res)
    res.sync()
    torch.cuda.synchronize()


# ---------------------------------------------------------------------------------------------- C5: 1B x 96 int8, list shards
def gen_int8_rows(n, dim, seed, device, row0=0, out=None):
    """int8 rows of the same latent mixture as gen_rows (scaled to the int8 range), generated chunk by chunk from
    (seed, chunk index): any row range can be produced on any rank, nothing but the chunk is ever resident."""
    x = gen_rows(n, dim, seed, device, row0=row0)
    x.mul_(40.0).round_().clamp_(-127, 127)
    y = x.to(torch.int8) if out is None else out.copy_(x)
    del x
    return y


def run_c5(args):
    """BASELINE configs[4]: IVF-PQ over `rows` x 96 int8 vectors, the index sharded by IVF list over the ranks (one process
    per GPU), per-rank top-k all-gathered by RCCL and merged (mg_ivf_pq.h:152-190, snmg.cuh:128-166,248-375). No rank
    ever holds the corpus: rows are generated 2^24 at a time, every rank encodes the chunk's rows that fall into its own
    lists and drops the rest; the lists are dealt to the ranks by size (greedy LPT over the list histogram). Recall is
    measured against an exact search that regenerates the chunks. Search without refinement (the corpus is not kept)."""
    import cuvs_amd
    from cuvs_amd._lib import lib
    from cuvs_amd.neighbors import ivf_pq, ivf_pq_sharded as sh

    rank, world, dev, dist, shared_dev = dist_setup(args)
    res = cuvs_amd.common.Resources()
    rows = args.rows if args.rows != 100_000_000 else 1_000_000_000
    dim, chunk = 96, 1 << 24
    n_lists = args.n_lists if args.n_lists != 16384 else 32768
    nq_total = args.batch * world
    t0 = time.time()
    # model: trained on a sample of the first chunk (the same rows, hence the same model, on every rank)
    first = gen_int8_rows(min(chunk, rows), dim, 1234, dev)
    train = first[:: max(1, first.shape[0] // 2_000_000)].contiguous()
    ip = ivf_pq.IndexParams(n_lists=n_lists, metric="sqeuclidean", pq_dim=args.pq_dim, pq_bits=8, kmeans_n_iters=20,
                            kmeans_trainset_fraction=1.0, add_data_on_build=False)
    comm = make_comm(sh, res, world, shared_dev)
    index = ivf_pq.build(ip, train, resources=res)
    # pass 1: list histogram (every rank counts its share of the chunks, the counts are summed), lists dealt by size
    counts = np.zeros(n_lists, np.uint64)
    n_chunks = (rows + chunk - 1) // chunk
    for c in range(rank, n_chunks, world):
        x = first if c == 0 else gen_int8_rows(min(chunk, rows - c * chunk), dim, 1234, dev, row0=c * chunk)
        counts += sh.list_histogram(index, x, resources=res)
        del x
    if world > 1:
        t = torch.from_numpy(counts.astype(np.int64)).to("cpu" if shared_dev else dev)
        dist.all_reduce(t)
        counts = t.cpu().numpy().astype(np.uint64)
    owners = sh.deal_lists(counts, world)
    sh.set_list_owners(index, owners, rank, world)
    mine = int(counts[owners == rank].sum())
    ratio = max(1, args.refine_ratio)
    owners_t = torch.from_numpy(owners.astype(np.int64)).to(dev)
    # pass 2: every rank sees every chunk and keeps the rows of its own lists: their codes in the index - under LOCAL ids,
    # the position in own_rows - and, for the refinement, the int8 rows themselves with the map local id -> global row
    own_rows = torch.empty((mine if ratio > 1 else 0, dim), dtype=torch.int8, device=dev)
    gmap = torch.empty(mine if ratio > 1 else 0, dtype=torch.int64, device=dev)
    n_local = 0
    for c in range(n_chunks):
        r0 = c * chunk
        x = first if c == 0 else gen_int8_rows(min(chunk, rows - r0), dim, 1234, dev, row0=r0)
        if ratio > 1:
            own = torch.nonzero(owners_t[sh.row_labels(index, x, resources=res)] == rank).flatten()
            xo = x[own]
            own_rows[n_local:n_local + xo.shape[0]] = xo
            gmap[n_local:n_local + xo.shape[0]] = own + r0
            sh.extend(index, xo, torch.arange(n_local, n_local + xo.shape[0], dtype=torch.int64, device=dev), resources=res)
            n_local += xo.shape[0]
            del xo, own
        else:
            sh.extend(index, x, torch.arange(r0, r0 + x.shape[0], dtype=torch.int64, device=dev), resources=res)
        del x
    assert ratio == 1 or n_local == mine, (n_local, mine)
    del first, train
    sh.attach_comm(index, comm)
    res.sync()
    build_s = time.time() - t0
    log(f"C5: built {rows} x {dim} int8 in {build_s:.1f}s; this rank holds {len(index)} rows ({mine} by the histogram)")
    queries = torch.cat([gen_int8_rows(args.batch, dim, 4321 + r, dev) for r in range(world)])
    k = args.k
    kk = k * ratio
    sp = ivf_pq.SearchParams(n_probes=args.n_probes, lut_dtype=LUTS[args.lut], internal_distance_dtype=LUTS[args.acc],
                             max_internal_batch_size=nq_total)
    pq_i = torch.empty((nq_total, kk), dtype=torch.int64, device=dev)
    pq_d = torch.empty((nq_total, kk), dtype=torch.float32, device=dev)
    cand_i = torch.empty((nq_total, k), dtype=torch.int64, device=dev)
    cand_d = torch.empty((nq_total, k), dtype=torch.float32, device=dev)
    out_i, out_d = torch.empty_like(cand_i), torch.empty_like(cand_d)
    from cuvs_amd.neighbors import refine
    INVALID = torch.iinfo(torch.int64).max

    def step():
        if ratio > 1:
            # this rank's kk candidates of every query (local ids), re-ranked exactly against its own int8 rows, then global ids
            ivf_pq.search(sp, index, queries, kk, neighbors=pq_i, distances=pq_d, resources=res)
            refine(own_rows, queries, pq_i, indices=cand_i, distances=cand_d, metric="sqeuclidean", resources=res)
            ok = cand_i != INVALID
            cand_i.copy_(torch.where(ok, gmap[torch.where(ok, cand_i, 0)], cand_i))
        else:
            ivf_pq.search(sp, index, queries, k, neighbors=cand_i, distances=cand_d, resources=res)
        comm.all_gather_topk(cand_d, cand_i, out=(out_d, out_i), resources=res)

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
        t = ctl_tensor([elapsed], torch.float64, dev, shared_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    scan_ms, ag_ms = C.c_double(0), C.c_double(0)
    n_launch = lib().cuvsAmdProfileCollect(b"pq_scan_kernel", C.byref(scan_ms))
    lib().cuvsAmdProfileCollect(b"shard_all_gather", C.byref(ag_ms))
    phase_ms = {}
    for nm in (b"pq_head_kernel", b"pq_filter_kernel", b"pq_rescore_kernel"):
        v = C.c_double(0)
        lib().cuvsAmdProfileCollect(nm, C.byref(v))
        phase_ms[nm.decode()] = round(v.value / max(args.steps, 1), 3)
    # roofline of the dominant kernel on this rank's lists (same floors as the headline line) + the bounded CPU leg
    roofline = cpu = None
    if rank == 0:
        roofline = pq_scan_roofline(index, queries.float(), args.n_probes, owners_t == rank, args.pq_dim, index.pq_dim * index.pq_len,
                                    phase_ms["pq_filter_kernel"], scan_ms.value, n_launch, args.steps, phase_ms)
        if not args.no_cpu_baseline:
            cpu = cpu_baseline_line(dev=dev)
    # recall@k of rank 0's slice against an exact search over regenerated chunks (fp32 arithmetic is exact on int8 values)
    ng = min(args.gt_queries, 200, args.batch)
    recall = None
    if rank == 0:
        qf = queries[:ng].float()
        best_d = torch.full((ng, k), float("inf"), device=dev)
        best_i = torch.zeros((ng, k), dtype=torch.int64, device=dev)
        for c in range(n_chunks):
            x = gen_int8_rows(min(chunk, rows - c * chunk), dim, 1234, dev, row0=c * chunk).float()
            d2 = (x * x).sum(1)[None, :] - 2.0 * (qf @ x.T)
            v, ii = torch.topk(d2, k, dim=1, largest=False)
            cd, ci = torch.cat([best_d, v], 1), torch.cat([best_i, ii + c * chunk], 1)
            o = torch.argsort(cd, dim=1)[:, :k]
            best_d, best_i = torch.gather(cd, 1, o), torch.gather(ci, 1, o)
            del x, d2
        recall = recall_of(out_i[:ng].cpu().numpy(), best_i.cpu().numpy())
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        emit_json({
            "metric": f"QPS, IVF-PQ {rows}x96 int8 list-sharded, batch={args.batch} per GPU", "value": round(nq_total / (ms * 1e-3), 1),
            "unit": "queries/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": f"u8 codes, {args.lut} LUT, {args.acc} score", "data": "synthetic",
            "config": {"workload": f"C5 IVF-PQ {rows}x96 int8, pq_dim={args.pq_dim} pq_bits=8 n_lists={n_lists} n_probes={args.n_probes} "
                                   f"batch={args.batch} per GPU k={k}, " + (f"shard-local refinement of {kk} candidates per query and rank" if ratio > 1 else "no refinement"),
                       "parallelism": f"list shards x{world} dealt by size (LPT), RCCL all-gather of the per-rank top-k",
                       "transport": "host-staged (mapped file): ranks share devices - functional run, not a scaling figure" if shared_dev
                                    else "RCCL (ncclAllGather / ncclAllReduce over xGMI)",
                       "oversubscribed": bool(shared_dev), "refine_ratio": ratio,
                       "rows_on_rank0": len(index), "build_seconds": round(build_s, 1)},
            f"recall_at_{k}": None if recall is None else round(recall, 4),
            "scan_kernel_ms_per_step": round(scan_ms.value / max(args.steps, 1), 3), "scan_launches_per_step": n_launch // max(args.steps, 1),
            "all_gather_merge_ms_per_step": round(ag_ms.value / max(args.steps, 1), 3),
            "roofline": roofline, "cpu_baseline": cpu})
    if world > 1:
        dist.barrier()
    comm.close()
    if world > 1:
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------- extra configs (N=1)
def timeit(fn, steps, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps


def profiled(name, fn, steps, warm):
    from cuvs_amd._lib import lib

    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    lib().cuvsAmdProfileEnable(1)
    t = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / steps
    lib().cuvsAmdProfileEnable(0)
    ms = C.c_double(0)
    n = lib().cuvsAmdProfileCollect(name, C.byref(ms))
    return dt, ms.value / steps, n // max(steps, 1)


def extra_c1(res, dev):
    """C1: brute force L2, 100k x 128 fp32, batch 1k, k = 10 (MFMA fp32 roofline)."""
    from cuvs_amd.neighbors import brute_force

    x = gen_rows(100_000, 128, 1234, dev)
    q = gen_rows(1000, 128, 4321, dev)
    idx = brute_force.build(x, resources=res)
    dt = timeit(lambda: brute_force.search(idx, q, 10, resources=res), 20, 3)
    d, i = brute_force.search(idx, q, 10, resources=res)
    res.sync()
    d2 = torch.cdist(q.double(), x.double()) ** 2  # fp64 ground truth
    gt = torch.topk(d2, 10, dim=1, largest=False).indices
    tf = 2 * 1000 * 100_000 * 128 / dt / 1e12
    xh, qh = x.cpu().numpy(), q.cpu().numpy()
    line = {"config": "C1 brute_force L2 100000x128 fp32 batch=1000 k=10", "ms": round(dt * 1e3, 3),
            "qps": round(1000 / dt, 1), "recall_at_10": round(recall_of(i.cpu().numpy(), gt.cpu().numpy()), 4),
            "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": MFMA_F32_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(tf / MFMA_F32_TFLOPS, 4)}}
    # BASELINE configs[0] asks for bit-exact neighbour indices at exactly this shape: the CPU restatement of the search
    # (oracle.brute_force_knn: canonical norms, k-ordered fma chain, (value, position) select) as the CHECKER of the GPU result
    # (untimed; the oracle is never on the measured path)
    try:
        import oracle

        oracle.set_num_threads(0)
        od, oi = oracle.brute_force_knn(qh, xh, 10)
        line["c1_ids_equal_oracle"] = bool((i.cpu().numpy() == oi).all())
        line["c1_distances_equal_oracle"] = bool((d.cpu().numpy() == od).all())
    except Exception as e:
        line["c1_ids_equal_oracle"] = None
        line["c1_oracle_error"] = repr(e)[:200]
    return line, xh, qh


def extra_c2(res, dev):
    """C2: IVF-Flat 10M x 128 fp32, n_lists 4096, n_probes 64, batch 10k."""
    from cuvs_amd.neighbors import brute_force, ivf_flat

    n, nq = 10_000_000, 10000
    x = gen_rows(n, 128, 1234, dev)
    q = gen_rows(nq, 128, 4321, dev)
    t0 = time.time()
    idx = ivf_flat.build(ivf_flat.IndexParams(n_lists=4096, kmeans_trainset_fraction=0.1), x, resources=res)
    res.sync()
    build_s = time.time() - t0
    sp = ivf_flat.SearchParams(n_probes=64)
    nb = torch.empty((nq, 10), dtype=torch.int64, device=dev)
    dd = torch.empty((nq, 10), dtype=torch.float32, device=dev)
    _, scan_ms, launches = profiled(b"ivf_flat_scan_kernel",
                                    lambda: ivf_flat.search(sp, idx, q, 10, neighbors=nb, distances=dd, resources=res), 5, 2)
    from cuvs_amd._lib import lib
    for nm in (b"flat_filter_kernel", b"flat_rescore_kernel"):
        lib().cuvsAmdProfileCollect(nm, None)
    # the search time without the per-kernel HIP events of profiled()
    dt = timeit(lambda: ivf_flat.search(sp, idx, q, 10, neighbors=nb, distances=dd, resources=res), 10, 2)
    bf = brute_force.build(x, resources=res)
    gt = exact_topk_fp64(x, q[:1000], 10)  # fp64 in torch, independent of this library (eval_neighbours, ann_utils.cuh:222-289)
    r = recall_of(nb[:1000].cpu().numpy(), gt.cpu().numpy())
    # the exact search over the same 10M rows and 10k queries (the ground-truth index): the distance GEMM at scale
    bf_dt = timeit(lambda: brute_force.search(bf, q, 10, resources=res), 2, 1)
    bf_tf = 2 * nq * n * 128 / bf_dt / 1e12
    logical = 64 * (n / 4096) * 512 * nq  # 80 MB of list bytes per query
    # Round 3: the nearest probe of every query runs on ivf_flat_scan_kernel (fp32 rows), the other 63 on the matrix-core
    # filter over the fp16 residual copy of the rows (ivf_pq_scan3.hip, FLAT build) + the fp32 re-scoring of the survivors.
    # Unique bytes the scan phases must fetch: every list once as fp32 (head phase: ~all 4096 lists are some query's nearest)
    # and once as fp16 (tail phase), plus the 4-byte row terms.
    # the same rows as int8 (x 24, rounded, clipped): the matrix-core tail phase serves every row type since round 5; the comparator
    # handle runs the round-4 path for int8 (ivf_flat_scan_kernel for all 64 probes)
    del bf
    int8_line = None
    try:
        xi = torch.empty((n, 128), dtype=torch.int8, device=dev)
        for r0 in range(0, n, 1 << 22):
            xi[r0:r0 + (1 << 22)] = torch.clamp(torch.round(x[r0:r0 + (1 << 22)] * 24.0), -128, 127).to(torch.int8)
        qi = torch.clamp(torch.round(q * 24.0), -128, 127).to(torch.int8)
        idx8 = ivf_flat.build(ivf_flat.IndexParams(n_lists=4096, kmeans_trainset_fraction=0.1), xi, resources=res)
        res.sync()
        dt8 = timeit(lambda: ivf_flat.search(sp, idx8, qi, 10, neighbors=nb, distances=dd, resources=res), 10, 2)
        gt8 = exact_topk_fp64(xi, qi[:1000], 10)
        r8 = recall_of(nb[:1000].cpu().numpy(), gt8.cpu().numpy())
        keep_i, keep_d = nb.clone(), dd.clone()
        res_s = comparator_handle(CUVS_AMD_FLAT_SCAN3=0)
        dt8s = timeit(lambda: ivf_flat.search(sp, idx8, qi, 10, neighbors=nb, distances=dd, resources=res_s), 3, 1)
        int8_line = {"config": "C2 rows as int8 (x 24, rounded): IVF-Flat 10000000x128 int8 n_lists=4096 n_probes=64 batch=10000 k=10",
                     "ms": round(dt8 * 1e3, 3), "qps": round(nq / dt8, 1), "recall_at_10": round(r8, 4),
                     "ms_scan_kernel_only": round(dt8s * 1e3, 3),
                     "equals_scan_kernel": bool(torch.equal(keep_i, nb) and torch.equal(keep_d, dd))}
        del idx8, xi
    except Exception as e:  # (a failed side line must not take the C2 line with it)
        int8_line = {"error": repr(e)[:200]}
    # the same rows under inner product / cosine (round 5: head phase + matrix-core tail phase; the comparator handle runs the
    # one-phase scan of rounds 1-4)
    ip_line = {}
    for mname in ("inner_product", "cosine"):
        try:
            idx_ip = ivf_flat.build(ivf_flat.IndexParams(n_lists=4096, kmeans_trainset_fraction=0.1, metric=mname), x, resources=res)
            res.sync()
            dt_ip = timeit(lambda: ivf_flat.search(sp, idx_ip, q, 10, neighbors=nb, distances=dd, resources=res), 10, 2)
            keep_i, keep_d = nb.clone(), dd.clone()
            res_s = comparator_handle(CUVS_AMD_FLAT_SCAN3=0)
            dt_ips = timeit(lambda: ivf_flat.search(sp, idx_ip, q, 10, neighbors=nb, distances=dd, resources=res_s), 3, 1)
            ip_line[mname] = {"config": f"C2 rows, {mname}: IVF-Flat 10000000x128 fp32 n_lists=4096 n_probes=64 batch=10000 k=10",
                              "ms": round(dt_ip * 1e3, 3), "qps": round(nq / dt_ip, 1), "ms_scan_kernel_only": round(dt_ips * 1e3, 3),
                              "equals_scan_kernel": bool(torch.equal(keep_i, nb) and torch.equal(keep_d, dd))}
            del idx_ip
        except Exception as e:
            ip_line[mname] = {"error": repr(e)[:200]}
    unique = n * 512  # the lower bound on HBM bytes per batch = unique probed-list bytes (this library's own fp16 copy is its cost, not algorithmic work)
    hbm_gbs = unique / (scan_ms * 1e-3) / 1e9
    return {"brute_force_same_data": {"config": "brute_force L2 10000000x128 fp32 batch=10000 k=10", "ms": round(bf_dt * 1e3, 1),
                                      "qps": round(nq / bf_dt, 1),
                                      "roofline": {"bound": "mfma", "achieved": round(bf_tf, 1), "peak": MFMA_F32_TFLOPS,
                                                   "unit": "TFLOP/s", "frac": round(bf_tf / MFMA_F32_TFLOPS, 4)}},
            "int8_rows": int8_line, "other_metrics": ip_line,
            "config": "C2 IVF-Flat 10000000x128 fp32 n_lists=4096 n_probes=64 batch=10000 k=10", "ms": round(dt * 1e3, 3),
            "qps": round(nq / dt, 1), "recall_at_10": round(r, 4), "build_seconds": round(build_s, 1),
            "kernel": "bound-only head phase (flat_filter2_kernel<EMIT> over the nearest lists + select_k + exact bound of the k best) + flat_filter2_kernel + flat_rescore_kernel (all 64 probes)",
            "kernel_ms_per_step": round(scan_ms, 3), "launches_per_step": launches,
            "roofline": {"bound": "hbm", "logical_scan_gbs": round(logical / (scan_ms * 1e-3) / 1e9, 1),
                         "achieved": round(hbm_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_gbs / HBM_PEAK_GBS, 4),
                         "unique_bytes": unique,
                         "note": "frac = algorithmic unique bytes (every row of the index once, 10M x 512 B) / (scan-kernel time x 8 TB/s); "
                                 "what the kernels fetch (profiles/r06_pmc_c2.json): the fp16 residual copy twice (bound-only head phase over "
                                 "the nearest lists 2.50 GB, filter 2.83 GB) and the fp32 rows of the survivors (re-score 2.32 GB) = 7.7 GB. "
                                 "logical_scan_gbs = list bytes per pair and kernel second (SURVEY 8d)"}}


def extra_pq768(res, dev, rows=1_000_000):
    """IVF-PQ at the reference's DEFAULT shape for 768-d rows (ivf_pq.hpp index_params: pq_dim = 0 -> dim / 2 = 384 at pq_len 2, 8-bit codes;
    n_lists 1024), 32 probes, batch 10k, k = 10 - the shape the matrix-core tail phase of rounds 3-5 did not cover (rot_dim beyond its
    decode table; the LUT itself beyond the LDS). Round 6: the wide path (ivf_pq_wide.hip) - the index's decoded fp16 rows through a GEMM
    filter, bound-only head phase - against the LUT scan kernels (CUVS_AMD_PQ_WIDE=0) on the same index: equality of ids and distances,
    ms, recall, the filter's counters and its kernel time against the fp16 MFMA peak and against HBM on the decoded rows it must read."""
    from cuvs_amd._lib import lib
    from cuvs_amd.neighbors import ivf_pq

    nq, k, n_probes, n_lists = 10000, 10, 32, 1024
    x = torch.empty((rows, 768), dtype=torch.float16, device=dev)
    gen_rows(rows, 768, 1234, dev, latent=32, n_modes=4096, out=x, spread=0.7)
    q = torch.empty((nq, 768), dtype=torch.float16, device=dev)
    gen_rows(nq, 768, 4321, dev, latent=32, n_modes=4096, out=q, spread=0.7)
    t0 = time.time()
    idx = ivf_pq.build(ivf_pq.IndexParams(n_lists=n_lists, kmeans_n_iters=10, kmeans_trainset_fraction=0.5), x, resources=res)
    res.sync()
    build_s = time.time() - t0
    sp = ivf_pq.SearchParams(n_probes=n_probes, lut_dtype=np.float16, internal_distance_dtype=np.float32, max_internal_batch_size=nq)
    nb = torch.empty((nq, k), dtype=torch.int64, device=dev)
    dd = torch.empty((nq, k), dtype=torch.float32, device=dev)
    step = lambda r=res: ivf_pq.search(sp, idx, q, k, neighbors=nb, distances=dd, resources=r)
    _, filt_ms, launches = profiled(b"pq_filter_kernel", step, 5, 2)
    for nm in (b"pq_scan_kernel", b"pq_head_kernel", b"pq_rescore_kernel", b"pq_bprep_kernel"):
        lib().cuvsAmdProfileCollect(nm, None)
    dt = timeit(step, 10, 2)
    keep_i, keep_d = nb.clone(), dd.clone()
    gt = exact_topk_fp64(x, q[:1000], k).cpu().numpy()
    rec = recall_of(keep_i[:1000].cpu().numpy(), gt)
    res_l = comparator_handle(CUVS_AMD_PQ_WIDE=0)
    dt_l = timeit(lambda: step(res_l), 3, 1)
    same = bool(torch.equal(keep_i, nb) and torch.equal(keep_d, dd))
    res_st = comparator_handle(CUVS_AMD_SCAN_DEBUG=1024)
    old_err = os.dup(2); devnull = os.open(os.devnull, os.O_WRONLY); os.dup2(devnull, 2)
    try:
        step(res_st); res_st.sync()
    finally:
        os.dup2(old_err, 2); os.close(devnull); os.close(old_err)
    st = (C.c_uint64 * 6)()
    lib().cuvsAmdIvfPqLastFilterStats6(st)
    pairs = int(st[0])
    # the filter launch (tail pairs) + the emit launch (head pairs) are both named pq_filter_kernel; algorithmic work of the tail launch:
    # one fp16 multiply-add per (row, query) pair and rotated dimension; bytes: every probed list's decoded rows once per batch
    flops = 2.0 * pairs * 768
    tf = flops / (filt_ms * 1e-3) / 1e12 if filt_ms > 0 else None
    return {"config": f"IVF-PQ {rows}x768 fp16 rows, the reference's default pq_dim (384 x pq_len 2, 8 bit), n_lists={n_lists} n_probes={n_probes} "
                      f"batch={nq} k={k} lut=f16 acc=f32 (data: bench.gen_rows, 4096 overlapping modes in a 32-d latent space)",
            "path": "wide matrix-core path (ivf_pq_wide.hip): decoded fp16 rows, bound-only head phase, GEMM filter, wave-per-survivor re-score",
            "ms": round(dt * 1e3, 3), "qps": round(nq / dt, 1), "recall_at_10": round(rec, 4), "build_seconds": round(build_s, 1),
            "ms_lut_scan_kernels": round(dt_l * 1e3, 3), "speedup_over_lut_scan": round(dt_l / dt, 2), "equals_lut_scan": same,
            "filter_kernels_ms": round(filt_ms, 3), "filter_launches_per_search": launches,
            "pairs_screened": pairs, "survivors": int(st[1]), "pairs_handed_back": int(st[4]),
            "roofline": {"bound": "mfma", "achieved": round(tf, 1) if tf else None, "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": round(tf / 2500.0, 4) if tf else None,
                         "note": "2 x 768 flops per screened (row, query) pair of the tail launch over the time of BOTH launches "
                                 "(tail + the head pairs' emit pass); fp16 dense MFMA peak 2.5 PF"}}


def extra_flat768(res, dev, rows=1_000_000):
    """IVF-Flat at 768 dimensions (fp16 rows, 1024 lists, 32 probes, batch 10k, k = 10): the tail phase on the wide filter (round 6: DESIGN
    3.1i; before it, the scan kernel alone served every dimension above 256) against the scan kernel alone on the same index."""
    from cuvs_amd.neighbors import ivf_flat

    nq, k = 10000, 10
    x = torch.empty((rows, 768), dtype=torch.float16, device=dev)
    gen_rows(rows, 768, 1234, dev, latent=32, n_modes=4096, out=x, spread=0.7)
    q = torch.empty((nq, 768), dtype=torch.float16, device=dev)
    gen_rows(nq, 768, 4321, dev, latent=32, n_modes=4096, out=q, spread=0.7)
    idx = ivf_flat.build(ivf_flat.IndexParams(n_lists=1024, kmeans_trainset_fraction=0.5), x, resources=res)
    res.sync()
    sp = ivf_flat.SearchParams(n_probes=32)
    nb = torch.empty((nq, k), dtype=torch.int64, device=dev)
    dd = torch.empty((nq, k), dtype=torch.float32, device=dev)
    dt = timeit(lambda: ivf_flat.search(sp, idx, q, k, neighbors=nb, distances=dd, resources=res), 10, 2)
    keep_i, keep_d = nb.clone(), dd.clone()
    rec = recall_of(keep_i[:1000].cpu().numpy(), exact_topk_fp64(x, q[:1000], k).cpu().numpy())
    res_s = comparator_handle(CUVS_AMD_FLAT_SCAN3=0)
    dt_s = timeit(lambda: ivf_flat.search(sp, idx, q, k, neighbors=nb, distances=dd, resources=res_s), 3, 1)
    return {"config": f"IVF-Flat {rows}x768 fp16 n_lists=1024 n_probes=32 batch={nq} k={k} (data: bench.gen_rows, 4096 overlapping modes in a 32-d latent space)",
            "path": "exact head phase on the scan kernel, tail phase on the wide matrix-core filter (ivf_pq_wide.hip) over the fp16 residual copy",
            "ms": round(dt * 1e3, 3), "qps": round(nq / dt, 1), "recall_at_10": round(rec, 4), "ms_scan_kernel_only": round(dt_s * 1e3, 3),
            "speedup_over_scan_kernel": round(dt_s / dt, 2), "equals_scan_kernel": bool(torch.equal(keep_i, nb) and torch.equal(keep_d, dd))}


def extra_cagra128(res, dev, rows=2_000_000):
    """CAGRA build on 128-d fp32 rows (degree 64 / 128): its kNN-graph searches (k = 256 of ~1.4 k-row lists) on the wide path's
    multi-list bounds (round 6) against the LUT scan kernels of rounds 1-5 (CUVS_AMD_PQ_WIDE=0) - build seconds, and the search on both graphs."""
    from cuvs_amd.neighbors import cagra

    nq = 10000
    x = gen_rows(rows, 128, 1234, dev, latent=24, n_modes=4096, spread=0.7)
    q = gen_rows(nq, 128, 4321, dev, latent=24, n_modes=4096, spread=0.7)
    gt = exact_topk_fp64(x, q[:1000], 10).cpu().numpy()
    out = {"config": f"CAGRA build {rows}x128 fp32 intermediate_graph_degree=128 graph_degree=64; search itopk=64 batch={nq} k=10"}
    for name, r in (("wide_path", res), ("lut_scan_kernels", comparator_handle(CUVS_AMD_PQ_WIDE=0))):
        t0 = time.time()
        idx = cagra.build(cagra.IndexParams(intermediate_graph_degree=128, graph_degree=64), x, resources=r)
        r.sync()
        build_s = time.time() - t0
        nb = torch.empty((nq, 10), dtype=torch.int32, device=dev)
        dd = torch.empty((nq, 10), dtype=torch.float32, device=dev)
        sp = cagra.SearchParams(itopk_size=64)
        t = timeit(lambda: cagra.search(sp, idx, q, 10, neighbors=nb, distances=dd, resources=res), 5, 2)
        out[name] = {"build_seconds": round(build_s, 2), "search_ms": round(t * 1e3, 3),
                     "recall_at_10": round(recall_of(nb[:1000].cpu().numpy().astype(np.int64) & 0xFFFFFFFF, gt), 4)}
        del idx
    out["build_speedup"] = round(out["lut_scan_kernels"]["build_seconds"] / max(out["wide_path"]["build_seconds"], 1e-9), 2)
    return out


def extra_c4_family(res, dev, rows=2_000_000, latent=24):
    """CAGRA (degree 64, intermediate 128) on the two ends of the generator family, 2M x 768 fp16 each: ONE cloud (what rounds 2-5 quoted
    C4 on) and 4096 TIGHT modes (spread 0.35: the kNN graph falls apart into components, a walk from random seeds stays in the modes
    it lands in - guarantee_connectivity joins the components but not the walk). itopk 64 and 256, k = 10, 10k queries."""
    from cuvs_amd.neighbors import cagra

    nq = 10000
    out = {"config": f"CAGRA {rows}x768 fp16 graph_degree=64 batch=10000 k=10, generator bench.gen_rows with a {latent}-d latent space", "corpora": []}
    for modes, spread in ((1, 0.35), (4096, 0.35)):
        x = torch.empty((rows, 768), dtype=torch.float16, device=dev)
        gen_rows(rows, 768, 1234, dev, latent=latent, n_modes=modes, out=x, spread=spread)
        q = torch.empty((nq, 768), dtype=torch.float16, device=dev)
        gen_rows(nq, 768, 4321, dev, latent=latent, n_modes=modes, out=q, spread=spread)
        gt = exact_topk_fp64(x, q[:1000], 10, chunk=250_000).cpu().numpy()  # fp64 in torch, independent of this library
        t0 = time.time()
        idx = cagra.build(cagra.IndexParams(intermediate_graph_degree=128, graph_degree=64), x, resources=res)
        res.sync()
        line = {"modes": modes, "spread": spread, "build_seconds": round(time.time() - t0, 1)}
        for itopk in (64, 256):
            sp = cagra.SearchParams(itopk_size=itopk, algo="auto")
            nb = torch.empty((nq, 10), dtype=torch.int32, device=dev)
            dd = torch.empty((nq, 10), dtype=torch.float32, device=dev)
            dt = timeit(lambda: cagra.search(sp, idx, q, 10, neighbors=nb, distances=dd, resources=res), 3, 1)
            rec = recall_of(nb[:1000].cpu().numpy().astype(np.int64) & 0xFFFFFFFF, gt)
            line[f"itopk_{itopk}"] = {"ms": round(dt * 1e3, 3), "qps": round(nq / dt, 1), "recall_at_10": round(rec, 4)}
        out["corpora"].append(line)
        del idx, x, q
        torch.cuda.empty_cache()
    return out


def extra_c4(res, dev, rows, latent, modes=1, spread=0.35):
    """C4: CAGRA rows x 768 fp16, graph_degree 64 (intermediate 128), itopk 64, batch 10k, k = 10."""
    from cuvs_amd.neighbors import brute_force, cagra

    nq = 10000
    x = torch.empty((rows, 768), dtype=torch.float16, device=dev)
    gen_rows(rows, 768, 1234, dev, latent=latent, n_modes=modes, out=x, spread=spread)
    q = torch.empty((nq, 768), dtype=torch.float16, device=dev)
    gen_rows(nq, 768, 4321, dev, latent=latent, n_modes=modes, out=q, spread=spread)
    t0 = time.time()
    idx = cagra.build(cagra.IndexParams(intermediate_graph_degree=128, graph_degree=64), x, resources=res)
    res.sync()
    build_s = time.time() - t0
    nb = torch.empty((nq, 10), dtype=torch.int32, device=dev)
    dd = torch.empty((nq, 10), dtype=torch.float32, device=dev)
    gt = exact_topk_fp64(x, q[:1000], 10, chunk=250_000).cpu().numpy()  # fp64 in torch, independent of this library
    from cuvs_amd._lib import check, lib

    def measured_work(sp):
        """rows scored / graph rows read / walkers of ONE search of the batch (cuvsAmdCagraWorkCounters)."""
        out = (C.c_uint64 * 3)()
        check(lib().cuvsAmdCagraWorkCounters(res.get_c_obj(), 1, out))
        cagra.search(sp, idx, q, 10, neighbors=nb, distances=dd, resources=res)
        check(lib().cuvsAmdCagraWorkCounters(res.get_c_obj(), 0, out))
        return int(out[0]), int(out[1]), int(out[2])

    algos = {}
    # the search algorithm is a search parameter (the reference's bench grids sweep it): multi_cta = one workgroup of W
    # waves per query, single_cta = one wave per query; AUTO follows the reference's rule (single_cta at this batch size)
    for algo in ("multi_cta", "single_cta", "auto"):
        sp = cagra.SearchParams(itopk_size=64, algo=algo)
        t = timeit(lambda: cagra.search(sp, idx, q, 10, neighbors=nb, distances=dd, resources=res), 5, 2)
        res.sync()
        n_dist, n_rows, n_walk = measured_work(sp)
        bytes_q = (n_dist * 768 * 2 + n_rows * 64 * 4) / nq  # n_dist * dim * sizeof(T) + n_iter * degree * 4
        algos[algo] = {"ms": round(t * 1e3, 3), "qps": round(nq / t, 1),
                       "recall_at_10": round(recall_of(nb[:1000].cpu().numpy().astype(np.int64) & 0xFFFFFFFF, gt), 4),
                       "rows_scored_per_query": round(n_dist / nq, 1), "graph_rows_read_per_query": round(n_rows / nq, 1),
                       "walkers_per_query": round(n_walk / nq, 2), "measured_bytes_per_query": int(bytes_q),
                       "gathered_gbs": round(bytes_q * nq / t / 1e9, 1)}
    best = max(algos, key=lambda a: algos[a]["qps"] if algos[a]["recall_at_10"] >= 0.9 else 0.0)
    dt, r = algos[best]["ms"] * 1e-3, algos[best]["recall_at_10"]
    gbs = algos[best]["gathered_gbs"]
    return {"config": f"C4 CAGRA {rows}x768 fp16 graph_degree=64 itopk=64 batch=10000 k=10 algo={best} (data: bench.gen_rows - Gaussian mixture of {modes} modes "
                      f"~ N(0, I) in a {latent}-d latent space, spread {spread} per mode, embedded in R^768 + 0.03 noise)",
            "ms": round(dt * 1e3, 3), "qps": round(nq / dt, 1), "recall_at_10": round(r, 4), "build_seconds": round(build_s, 1),
            "algos": algos,
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(gbs / HBM_PEAK_GBS, 4),
                         "note": "MEASURED gathered bytes (rows scored x 1536 B + graph rows read x 256 B, counted in the "
                                 "kernel: cuvsAmdCagraWorkCounters) per wall second; children already in the visited hash "
                                 "are not scored, so this is below SURVEY 8d's 6.9 MB/query upper bound"}}


# ---------------------------------------------------------------------------------------------------------- main
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
    ap.add_argument("--batch", type=int, default=10000, help="queries per step and GPU")
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--lut", choices=list(LUTS), default="f16", help="headline LUT dtype (search_params.lut_dtype)")
    ap.add_argument("--acc", choices=["f32", "f16"], default="f32",
                    help="headline score dtype (internal_distance_dtype); fp16 LUT / fp32 score is what the reference's own "
                         "bench grid runs (cuvs_ivf_pq.yaml: smemLutDtype half, internalDistanceDtype float)")
    ap.add_argument("--refine-ratio", type=int, default=2,
                    help="IVF-PQ returns ratio*k candidates that cuvsRefine re-ranks exactly (reference bench grids "
                         "use refine_ratio 1..4, python/cuvs_bench/.../cuvs_ivf_pq.yaml); 1 disables refinement")
    ap.add_argument("--trainset-fraction", type=float, default=0.02)
    ap.add_argument("--gt-queries", type=int, default=1000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--no-pmc", action="store_true")
    ap.add_argument("--c4-rows", type=int, default=10_000_000)
    ap.add_argument("--c4-latent", type=int, default=24)
    ap.add_argument("--c4-modes", type=int, default=4096)
    ap.add_argument("--c4-spread", type=float, default=0.7)
    ap.add_argument("--pmc-child", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--config", choices=["c3", "c5"], default="c3",
                    help="c3: the headline (IVF-PQ 100M x 128 fp32); c5: IVF-PQ rows x 96 int8 list-sharded over the ranks "
                         "(BASELINE configs[4]: 1B rows over 8 GPUs; --rows scales it down), rows generated chunk by chunk")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the list-sharded code path (shard build, RCCL all-gather + merge) even with one rank")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak = every rank brings its own batch (N x batch queries per step over the N list shards: the "
                         "driver's SCALE run); strong = ONE batch of `batch` queries per step over the N list shards - BASELINE's metric "
                         "('batch=10k; 1/2/4/8 GPU'), the reference's SHARDED search (cpp/src/neighbors/mg/snmg.cuh:248-375: every rank "
                         "searches the same batch on its shard)")
    ap.add_argument("--share-devices", action="store_true",
                    help="functional run of the N-rank path on fewer than N devices: rank r uses device r %% n_devices, the "
                         "collectives go through the communicator's host-staged transport (RCCL refuses two ranks on one "
                         "device). Never a scaling figure: the JSON line carries transport / oversubscribed")
    args = ap.parse_args()
    if args.lut == "f32":
        args.acc = "f32"
    if args.pmc_child:
        return pmc_child(args, args.pmc_child)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)  # `python bench.py --gpus N`: this process becomes the launcher of N ranks
    capture_stdout()  # (every rank: only emit_json writes to the real stdout)
    if args.config == "c5":
        return run_c5(args)

    rank, world, dev, dist, shared_dev = dist_setup(args)

    import cuvs_amd
    from cuvs_amd._lib import lib
    from cuvs_amd.neighbors import brute_force, ivf_pq, ivf_pq_sharded, refine

    res = cuvs_amd.common.Resources()
    sharded = world > 1 or args.force_sharded
    strong = args.scaling == "strong" and world > 1
    # list-sharded search: every rank sees the whole query batch of a step - world x batch queries (weak), one batch (strong)
    nq_total = args.batch if strong else args.batch * world
    kk = args.k * max(1, args.refine_ratio)

    # ------------------------------------------------------------------ data + index (untimed)
    t0 = time.time()
    data = gen_rows(args.rows, args.dim, seed=1234, device=dev)
    queries = torch.cat([gen_rows(args.batch, args.dim, seed=4321 + r, device=dev) for r in range(1 if strong else world)])
    torch.cuda.synchronize()
    log(f"generated {args.rows}x{args.dim} fp32 in {time.time() - t0:.1f}s")
    t0 = time.time()
    ip = ivf_pq.IndexParams(n_lists=args.n_lists, metric="sqeuclidean", pq_dim=args.pq_dim, pq_bits=8,
                            kmeans_n_iters=20, kmeans_trainset_fraction=args.trainset_fraction,
                            add_data_on_build=not sharded)
    comm = None
    if not sharded:
        index = ivf_pq.build(ip, data, resources=res)
    else:
        # the same model on every rank (same rows, deterministic k-means); each rank then keeps the rows of its lists
        comm = make_comm(ivf_pq_sharded, res, world, shared_dev)
        index = ivf_pq_sharded.build(ip, data, rank, world, resources=res)
        # lists dealt to the ranks by size (greedy LPT over the list histogram): every rank holds the whole synthetic corpus
        # here, so each computes the same histogram and the same table - no communication
        hist = np.zeros(args.n_lists, np.uint64)
        for r0 in range(0, args.rows, 1 << 24):
            ivf_pq_sharded.list_histogram(index, data[r0:min(args.rows, r0 + (1 << 24))], hist, resources=res)
        owners = ivf_pq_sharded.deal_lists(hist, world)
        ivf_pq_sharded.set_list_owners(index, owners, rank, world)
        step_rows = 1 << 24
        for r0 in range(0, args.rows, step_rows):
            r1 = min(args.rows, r0 + step_rows)
            ivf_pq_sharded.extend(index, data[r0:r1], torch.arange(r0, r1, dtype=torch.int64, device=dev), resources=res)
    if sharded:
        ivf_pq_sharded.attach_comm(index, comm)  # head-phase bounds are all-reduced (min) before the tail phase
    res.sync()
    build_s = time.time() - t0
    log(f"built IVF-PQ index in {build_s:.1f}s ({len(index)} rows on this rank)")

    neighbors = torch.empty((nq_total, args.k), dtype=torch.int64, device=dev)
    distances = torch.empty((nq_total, args.k), dtype=torch.float32, device=dev)
    cand_i = torch.empty((nq_total, kk), dtype=torch.int64, device=dev)
    cand_d = torch.empty((nq_total, kk), dtype=torch.float32, device=dev)
    mrg_i, mrg_d = torch.empty_like(cand_i), torch.empty_like(cand_d)
    per_rank = (nq_total + world - 1) // world
    q_lo, q_hi = min(nq_total, rank * per_rank), min(nq_total, (rank + 1) * per_rank)  # the slice of the batch this rank refines

    def make_step(lut, acc, res=res):
        sp = ivf_pq.SearchParams(n_probes=args.n_probes, lut_dtype=LUTS[lut], internal_distance_dtype=LUTS[acc],
                                 max_internal_batch_size=nq_total)

        def step():
            if sharded:
                ivf_pq.search(sp, index, queries, kk, neighbors=cand_i, distances=cand_d, resources=res)
                comm.all_gather_topk(cand_d, cand_i, out=(mrg_d, mrg_i), resources=res)  # native RCCL all-gather + merge
                ci = mrg_i
            else:
                ivf_pq.search(sp, index, queries, kk, neighbors=cand_i, distances=cand_d, resources=res)
                ci = cand_i
            if args.refine_ratio > 1:
                refine(data, queries[q_lo:q_hi], ci[q_lo:q_hi], indices=neighbors[q_lo:q_hi], distances=distances[q_lo:q_hi],
                       metric="sqeuclidean", resources=res)
            else:
                neighbors[q_lo:q_hi].copy_(ci[q_lo:q_hi, :args.k])
        return step

    phase_ms = {}  # per-search HIP-event time of the tail-phase kernels of the last timed() call

    def timed(step, steps, warmup):
        for _ in range(warmup):
            step()
        lib().cuvsAmdProfileEnable(1)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t_start = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t_start
        lib().cuvsAmdProfileEnable(0)
        if world > 1:
            t = ctl_tensor([elapsed], torch.float64, dev, shared_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        scan_ms = C.c_double(0)
        n_launch = lib().cuvsAmdProfileCollect(b"pq_scan_kernel", C.byref(scan_ms))
        for nm in (b"pq_head_kernel", b"pq_filter_kernel", b"pq_rescore_kernel"):
            v = C.c_double(0)
            lib().cuvsAmdProfileCollect(nm, C.byref(v))
            phase_ms[nm.decode()] = v.value / max(steps, 1)
        ag_ms = C.c_double(0)
        lib().cuvsAmdProfileCollect(b"shard_all_gather", C.byref(ag_ms))
        return elapsed, scan_ms.value, n_launch, ag_ms.value

    # ------------------------------------------------------------------ timed region (headline variant)
    elapsed, scan_ms, n_launch, ag_ms = timed(make_step(args.lut, args.acc), args.steps, args.warmup)
    headline_phase_ms = {k: round(v, 3) for k, v in phase_ms.items()}

    # ------------------------------------------------------------------ recall@10 vs exact search (untimed)
    ng = min(args.gt_queries, q_hi - q_lo)
    truth = exact_topk_fp64(data, queries[q_lo:q_lo + ng], args.k).cpu().numpy()  # fp64, not this library's brute force
    recall = recall_of(neighbors[q_lo:q_lo + ng].cpu().numpy(), truth)

    # ------------------------------------------------------------------ bit-level check at the bench scale (untimed): the
    # first 1000 queries searched again by a handle whose tail phase runs the LUT scan kernels (no matrix-core filter) must
    # return the same ids AND the same distances as the headline path (ivf_pq_search.cuh:421-669: same scores, same top-k)
    # A list-sharded index searches collectively: every rank makes the two searches and the merged blocks are compared.
    scan3_equals_lut_scan = None
    if world > 1 or rank == 0:
        nchk = min(1000, args.batch)
        sp_chk = ivf_pq.SearchParams(n_probes=args.n_probes, lut_dtype=LUTS[args.lut], internal_distance_dtype=LUTS[args.acc],
                                     max_internal_batch_size=nq_total)
        a_i = torch.empty((nchk, kk), dtype=torch.int64, device=dev)
        a_d = torch.empty((nchk, kk), dtype=torch.float32, device=dev)
        b_i, b_d = torch.empty_like(a_i), torch.empty_like(a_d)
        res_lut = comparator_handle(CUVS_AMD_PQ_SCAN3=0)  # the switches are read once, when a handle is created
        for r_, o_i, o_d in ((res, a_i, a_d), (res_lut, b_i, b_d)):
            ivf_pq.search(sp_chk, index, queries[:nchk], kk, neighbors=o_i, distances=o_d, resources=r_)
            if sharded:
                g_d, g_i = comm.all_gather_topk(o_d, o_i, resources=r_)
                o_d.copy_(g_d); o_i.copy_(g_i)
            r_.sync()
        torch.cuda.synchronize()
        scan3_equals_lut_scan = bool(torch.equal(a_i, b_i) and torch.equal(a_d, b_d))
        if world > 1:
            t = ctl_tensor([1 if scan3_equals_lut_scan else 0], torch.int64, dev, shared_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            scan3_equals_lut_scan = bool(t.item())
        log(f"scan3 == LUT scan on {nchk} queries of the {args.rows}-row index: {scan3_equals_lut_scan}")
        del res_lut

    # ------------------------------------------------------------------ the other precisions (same step, untimed region)
    variants = []
    early_stop_off_ms = None
    if rank == 0 and world == 1 and not args.no_variants:
        for lut, acc in (("f32", "f32"), ("f16", "f32"), ("f16", "f16"), ("fp8", "f16")):
            if (lut, acc) == (args.lut, args.acc):
                v_el, v_scan, v_n, v_rec = elapsed / args.steps, scan_ms / args.steps, n_launch // args.steps, recall
            else:
                e, s, n, _ = timed(make_step(lut, acc), 5, 1)
                v_el, v_scan, v_n = e / 5, s / 5, n // 5
                v_rec = recall_of(neighbors[:ng].cpu().numpy(), truth)
            variants.append({"lut": lut, "acc": acc, "ms_per_step": round(v_el * 1e3, 3), "qps": round(args.batch / v_el, 1),
                             "recall_at_10": round(v_rec, 4), "scan_kernel_ms_per_step": round(v_scan, 3),
                             "scan_launches_per_step": v_n})
        # data-independent figure: the headline variant with every form of pruning off - no early stop
        # (CUVS_AMD_SCAN_DEBUG=8), no filter stage (CUVS_AMD_PQ_SCAN2=0), no head phase: all 64 gathers of every row
        res_off = comparator_handle(CUVS_AMD_SCAN_DEBUG=8, CUVS_AMD_PQ_SCAN2=0, CUVS_AMD_PQ_SCAN3=0, CUVS_AMD_PQ_HEAD_PROBES=0)
        _, s, _, _ = timed(make_step(args.lut, args.acc, res_off), 3, 1)
        early_stop_off_ms = round(s / 3, 3)
        del res_off

    # ------------------------------------------------------------------ roofline of the dominant kernel (every N: the
    # floors are those of THIS rank's lists - a list shard screens only the probes it owns - against rank 0's kernel time)
    owned = ((torch.from_numpy(owners.astype(np.int64)).to(dev) == rank) if sharded
             else torch.ones(args.n_lists, dtype=torch.bool, device=dev))
    per_step = max(n_launch, 1) / max(args.steps, 1)
    avg_ms = scan_ms / max(n_launch, 1)
    f_ms = headline_phase_ms.get("pq_filter_kernel", 0.0)
    roofline = pq_scan_roofline(index, queries, args.n_probes, owned, args.pq_dim * 8 // 8, 2 * args.pq_dim, f_ms, scan_ms, n_launch,
                                args.steps, headline_phase_ms, early_stop_off_ms)
    unique_bytes = roofline["algorithmic"]["unique_code_bytes_per_search"]
    if rank == 0 and world == 1 and not args.no_pmc:
        t0 = time.time()
        child = ["--rows", str(args.rows), "--dim", str(args.dim), "--n-lists", str(args.n_lists), "--n-probes",
                 str(args.n_probes), "--pq-dim", str(args.pq_dim), "--batch", str(args.batch), "--k", str(args.k),
                 "--lut", args.lut, "--acc", args.acc, "--refine-ratio", str(args.refine_ratio), "--trainset-fraction",
                 str(args.trainset_fraction)]
        pmc = run_pmc_passes(child, n_search=3)
        log(f"PMC passes took {time.time() - t0:.1f}s: {'ok' if pmc else 'unavailable'}")

        def pmc_fracs(c, t_ref):
            """busy fractions of every pipe over SPEC-clock cycles of the un-profiled kernel time t_ref (seconds)"""
            spec_cycles = t_ref * SPEC_GHZ * 1e9
            hbm_bytes = (2.0 * c.get("FETCH_SIZE", 0.0) + c.get("WRITE_SIZE", 0.0)) * 1024.0  # gfx950: FETCH_SIZE x 2
            lds_act = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
            return {"hbm_bytes": int(hbm_bytes), "hbm_frac": round(hbm_bytes / t_ref / 1e9 / HBM_PEAK_GBS, 4),
                    "lds_busy": round(lds_act / (spec_cycles * N_CU), 4),
                    "lds_bank_conflict_share": round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(lds_act, 1.0), 4),
                    "valu_busy": round(c.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (spec_cycles * N_SIMD), 4),
                    "mfma_busy": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (spec_cycles * N_SIMD), 4),
                    "tcc_hit_rate": round(c.get("TCC_HIT_sum", 0.0) / max(c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 0.0), 1.0), 4),
                    "valu_insts": int(c.get("SQ_INSTS_VALU", 0.0)), "lds_insts": int(c.get("SQ_INSTS_LDS", 0.0)),
                    "effective_clock_ghz": round(c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / max(t_ref, 1e-12) / 1e9, 3)}

        if pmc and pmc.get("GRBM_GUI_ACTIVE"):
            roofline["pmc"] = {"scan_kernels_of_one_search": pmc_fracs(pmc, avg_ms * per_step * 1e-3)}
            if pmc.get("filter") and f_ms > 0:
                pf = pmc_fracs(pmc["filter"], f_ms * 1e-3)
                roofline["pmc"]["pq_filter_kernel"] = pf
                roofline["traffic"] = pf["hbm_bytes"]  # HBM bytes of the dominant kernel's launch (FETCH_SIZE x 2 + WRITE_SIZE)
                roofline["traffic_over_unique_bytes"] = round(pf["hbm_bytes"] / max(unique_bytes, 1), 3)
            roofline["pmc_source"] = "live: rocprofv3 --pmc passes of this workload, spawned by this run"

    # ------------------------------------------------------------------ small batches on the same index (the reference bench sweeps
    # the batch size, cpp/bench/ann/src/common/benchmark.hpp:301-345): latency of one search + refine call and the QPS it gives
    batch_sweep = []
    if rank == 0 and world == 1 and not args.no_variants and not sharded:
        for nb in (1, 10, 100, 1000):
            if nb > args.batch:
                continue
            try:
                qs = queries[:nb].contiguous()
                b_i = torch.empty((nb, kk), dtype=torch.int64, device=dev)
                b_d = torch.empty((nb, kk), dtype=torch.float32, device=dev)
                o_i = torch.empty((nb, args.k), dtype=torch.int64, device=dev)
                o_d = torch.empty((nb, args.k), dtype=torch.float32, device=dev)
                sp_b = ivf_pq.SearchParams(n_probes=args.n_probes, lut_dtype=LUTS[args.lut], internal_distance_dtype=LUTS[args.acc],
                                           max_internal_batch_size=nq_total)

                def bstep(r=res):
                    ivf_pq.search(sp_b, index, qs, kk, neighbors=b_i, distances=b_d, resources=r)
                    if args.refine_ratio > 1:
                        refine(data, qs, b_i, indices=o_i, distances=o_d, metric="sqeuclidean", resources=r)

                dt_b = timeit(bstep, 30, 5)
                rec_b = recall_of((o_i if args.refine_ratio > 1 else b_i[:, :args.k])[:min(nb, ng)].cpu().numpy(), truth[:min(nb, ng)])
                line_b = {"batch": nb, "ms_per_call": round(dt_b * 1e3, 3), "qps": round(nb / dt_b, 1), "recall_at_10": round(rec_b, 4)}
                if 10 <= nb < 256:
                    # below 256 queries the search runs ONE phase (no head phase, LUT scan); the comparator forces the two-phase
                    # schedule (head + matrix-core tail) on the same batch: what the threshold is worth
                    res_h = comparator_handle(CUVS_AMD_PQ_HEAD_PROBES=1)
                    keep_i = b_i.clone()
                    dt_h = timeit(lambda: bstep(res_h), 30, 5)
                    line_b["ms_per_call_two_phase_forced"] = round(dt_h * 1e3, 3)
                    line_b["two_phase_ids_equal"] = bool(torch.equal(keep_i, b_i))
                batch_sweep.append(line_b)
            except Exception as e:
                batch_sweep.append({"batch": nb, "error": repr(e)[:200]})

    # ------------------------------------------------------------------ the same workload with inner product / cosine
    # (signed LUT entries: no early stop in a LUT scan; the matrix-core filter works on full-score bounds)
    metric_variants = []
    if rank == 0 and world == 1 and not args.no_variants and not sharded:
        gq = queries[:min(200, args.batch)]
        for metric in ("inner_product", "cosine"):
            try:
                t0 = time.time()
                mi = ivf_pq.build(ivf_pq.IndexParams(n_lists=args.n_lists, metric=metric, pq_dim=args.pq_dim, pq_bits=8, kmeans_n_iters=20,
                                                     kmeans_trainset_fraction=args.trainset_fraction), data, resources=res)
                res.sync()
                b_s = time.time() - t0
                sp = ivf_pq.SearchParams(n_probes=args.n_probes, lut_dtype=LUTS[args.lut], internal_distance_dtype=LUTS[args.acc],
                                         max_internal_batch_size=nq_total)
                m_dt = timeit(lambda: ivf_pq.search(sp, mi, queries, kk, neighbors=cand_i, distances=cand_d, resources=res), 5, 2)
                # ground truth in fp64: inner products (cosine: of the normalised vectors)
                qd = gq.double()
                if metric == "cosine":
                    qd = qd / qd.norm(dim=1, keepdim=True)
                best_v = torch.full((gq.shape[0], kk), -float("inf"), dtype=torch.float64, device=dev)
                best_i = torch.zeros((gq.shape[0], kk), dtype=torch.int64, device=dev)
                for r0 in range(0, args.rows, 500_000):
                    xc = data[r0:r0 + 500_000].double()
                    if metric == "cosine":
                        xc = xc / xc.norm(dim=1, keepdim=True)
                    v, ii = torch.topk(qd @ xc.T, kk, dim=1)
                    cv, ci = torch.cat([best_v, v], 1), torch.cat([best_i, ii + r0], 1)
                    o = torch.argsort(cv, dim=1, descending=True)[:, :kk]
                    best_v, best_i = torch.gather(cv, 1, o), torch.gather(ci, 1, o)
                    del xc
                rec = recall_of(cand_i[:gq.shape[0]].cpu().numpy(), best_i.cpu().numpy())
                metric_variants.append({"metric": metric, "ms_per_search": round(m_dt * 1e3, 3), "qps": round(args.batch / m_dt, 1),
                                        f"recall_at_{kk}_without_refine": round(rec, 4), "build_seconds": round(b_s, 1)})
                del mi
            except Exception as e:
                metric_variants.append({"metric": metric, "error": repr(e)[:300]})
            torch.cuda.empty_cache()

    # ------------------------------------------------------------------ the sharded code path with one rank (every round
    # times shard build, head-bound all-reduce, RCCL all-gather of the [Q, k] blocks and the merge, even without a node)
    sharded_line = None
    if rank == 0 and world == 1 and not args.no_variants and not sharded:
        try:
            t0 = time.time()
            scomm = ivf_pq_sharded.ShardComm(0, 1, ivf_pq_sharded.ShardComm.unique_id(), res)
            sip = ivf_pq.IndexParams(n_lists=args.n_lists, metric="sqeuclidean", pq_dim=args.pq_dim, pq_bits=8, kmeans_n_iters=20,
                                     kmeans_trainset_fraction=args.trainset_fraction, add_data_on_build=False)
            sidx = ivf_pq_sharded.build(sip, data, 0, 1, resources=res)
            for r0 in range(0, args.rows, 1 << 24):
                r1 = min(args.rows, r0 + (1 << 24))
                ivf_pq_sharded.extend(sidx, data[r0:r1], torch.arange(r0, r1, dtype=torch.int64, device=dev), resources=res)
            ivf_pq_sharded.attach_comm(sidx, scomm)
            res.sync()
            sb = time.time() - t0
            ssp = ivf_pq.SearchParams(n_probes=args.n_probes, lut_dtype=LUTS[args.lut], internal_distance_dtype=LUTS[args.acc],
                                      max_internal_batch_size=nq_total)

            def sstep():
                ivf_pq.search(ssp, sidx, queries, kk, neighbors=cand_i, distances=cand_d, resources=res)
                scomm.all_gather_topk(cand_d, cand_i, out=(mrg_d, mrg_i), resources=res)

            sstep()
            lib().cuvsAmdProfileEnable(1)
            s_dt = timeit(sstep, 5, 1)
            lib().cuvsAmdProfileEnable(0)
            agv = C.c_double(0)
            n_ag = lib().cuvsAmdProfileCollect(b"shard_all_gather", C.byref(agv))
            for nm in (b"pq_scan_kernel", b"pq_head_kernel", b"pq_filter_kernel", b"pq_rescore_kernel", b"shard_all_reduce"):
                lib().cuvsAmdProfileCollect(nm, None)
            same = bool(torch.equal(mrg_i, cand_i))
            sharded_line = {"config": "the headline search through the list-sharded path, one rank (native RCCL communicator)",
                            "ms_per_search_incl_all_gather_merge": round(s_dt * 1e3, 3), "all_gather_merge_ms": round(agv.value / max(n_ag, 1), 3),
                            "build_seconds": round(sb, 1), "merged_equals_local": same}
            ivf_pq_sharded.attach_comm(sidx, None)
            del sidx
            scomm.close()
        except Exception as e:
            sharded_line = {"config": "sharded one-rank", "error": repr(e)[:300]}
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ the headline on a second, LESS PRUNABLE corpus
    # (untimed region). The headline corpus has 65536 tight modes in a 32-d latent space: ~1e-4 of the (row, query) pairs of a
    # probed list survive the screen. Real corpora (deep-100M, datasets.yaml) are less clustered: here the same search on
    # 4096 wide modes in a 64-d latent space, with the survivors per pair of both corpora (CUVS_AMD_SCAN_DEBUG=1024 counters).
    def survivors_per_pair(idx_, q_):
        r_st = comparator_handle(CUVS_AMD_SCAN_DEBUG=1024)
        sp_ = ivf_pq.SearchParams(n_probes=args.n_probes, lut_dtype=LUTS[args.lut], internal_distance_dtype=LUTS[args.acc],
                                  max_internal_batch_size=nq_total)
        old_err = os.dup(2)  # the debug handle prints its counters to stderr: keep the log readable
        devnull = os.open(os.devnull, os.O_WRONLY)
        os.dup2(devnull, 2)
        try:
            ivf_pq.search(sp_, idx_, q_, kk, neighbors=cand_i, distances=cand_d, resources=r_st)
            r_st.sync()
        finally:
            os.dup2(old_err, 2); os.close(devnull); os.close(old_err)
        st = (C.c_uint64 * 6)()
        lib().cuvsAmdIvfPqLastFilterStats6(st)
        return {"pairs_screened": int(st[0]), "survivors": int(st[1]), "survivors_per_pair": (st[1] / st[0]) if st[0] else None,
                "subtiles_decoded": int(st[2]), "work_units": int(st[3]), "pairs_handed_back_to_the_lut_scan": int(st[4]),
                "candidates_through_the_overflow_list": int(st[5])}

    # The headline corpus, then the same step (same index parameters, same search parameters) on three others: wider modes; SURVEY
    # 8d's generator verbatim; an isotropic Gaussian with no cluster structure (the least prunable: the worst case of the two-phase
    # path is whatever this line says). Every line: ms per step, QPS, recall@10 vs fp64, kernel ms, survivors per (row, query)
    # pair of the screen, pairs handed back, overflow entries.
    contract_corpus = None

    def contract_point(name, idx_, data_, q_, truth_, ng_, ratios):
        """Sweep of the refine ratio on one corpus; returns the top-level `contract_corpus` entry: the first ratio whose recall@10
        reaches 0.9, timed with the headline's steps / warmup, with its phases and the roofline of its dominant kernel."""
        sweep, chosen = [], None
        for ratio in sorted(set(ratios)):
            kr = args.k * ratio
            c_i = torch.empty((args.batch, kr), dtype=torch.int64, device=dev)
            c_d = torch.empty((args.batch, kr), dtype=torch.float32, device=dev)
            sp_ = ivf_pq.SearchParams(n_probes=args.n_probes, lut_dtype=LUTS[args.lut], internal_distance_dtype=LUTS[args.acc],
                                      max_internal_batch_size=nq_total)

            def step_(c_i=c_i, c_d=c_d, sp_=sp_):
                ivf_pq.search(sp_, idx_, q_, kr, neighbors=c_i, distances=c_d, resources=res)
                refine(data_, q_, c_i, indices=neighbors, distances=distances, metric="sqeuclidean", resources=res)

            st_, wu_ = (args.steps, args.warmup) if chosen is None else (3, 1)
            e_, s_, n_, _ = timed(step_, st_, wu_)
            rec_ = recall_of(neighbors[:ng_].cpu().numpy(), truth_)
            ph_ = {k_: round(v_, 3) for k_, v_ in phase_ms.items()}
            pt = {"refine_ratio": ratio, "ms_per_step": round(e_ / st_ * 1e3, 3), "qps": round(args.batch / (e_ / st_), 1),
                  "recall_at_10": round(rec_, 4), "kernel_ms_per_step": round(s_ / st_, 3), "phase_ms_per_step": ph_}
            sweep.append(pt)
            if chosen is None and rec_ >= 0.9:
                rf = pq_scan_roofline(idx_, q_, args.n_probes, torch.ones(args.n_lists, dtype=torch.bool, device=dev), args.pq_dim,
                                      2 * args.pq_dim, ph_.get("pq_filter_kernel", 0.0), s_, n_, st_, ph_)
                rf.pop("note", None)
                chosen = {"corpus": name, "n_probes": args.n_probes, "refine_ratio": ratio, "k_searched": kr, "value": pt["qps"],
                          "unit": "queries/s", "ms_per_step": pt["ms_per_step"], "steps": st_, "warmup": wu_,
                          "recall_at_10": pt["recall_at_10"], "recall_queries": ng_, "phase_ms_per_step": ph_, "roofline": rf}
            del c_i, c_d
        out_ = chosen or {"corpus": name, "n_probes": args.n_probes, "value": None,
                          "note": "no refine ratio of the sweep reaches recall@10 0.9 at this n_probes"}
        out_["refine_ratio_sweep"] = sweep
        out_["note_sweep"] = ("n_probes 256 changes no recall on this corpus (profiles/r06_contract_sweep.log: the true neighbours' lists are "
                              "among the first 128; what misses them at a small ratio is the PQ ranking of ~1500 near-equidistant rows of a mode)")
        return out_

    corpus_variants = []
    if rank == 0 and world == 1 and not args.no_variants and not sharded:
        try:
            c0 = survivors_per_pair(index, queries)
            corpus_variants.append({"corpus": "headline: 65536 modes, 32-d latent, spread 0.35", "ms_per_step": round(elapsed / args.steps * 1e3, 3),
                                    "recall_at_10": round(recall, 4), "kernel_ms_per_step": round(scan_ms / max(args.steps, 1), 3),
                                    "phase_ms_per_step": headline_phase_ms, **c0})
        except Exception as e:
            corpus_variants.append({"corpus": "headline", "error": repr(e)[:300]})
        q_save = queries
        others = [("4096 modes, 64-d latent, spread 0.7 (wide, overlapping clusters)",
                   lambda n, seed: gen_rows(n, args.dim, seed=seed, device=dev, latent=64, n_modes=4096, spread=0.7)),
                  (f"SURVEY 8d verbatim: {4 * args.n_lists} centres ~ U[-1,1)^{args.dim}, points = centre + N(0, 0.1^2 I)",
                   lambda n, seed: gen_rows_survey8d(n, args.dim, seed, dev, args.n_lists)),
                  (f"isotropic Gaussian N(0, I_{args.dim}): no cluster structure (least prunable)",
                   lambda n, seed: gen_rows_gaussian(n, args.dim, seed, dev))]
        ngv = min(ng, 200)
        for name, gen in others:
            try:
                t0 = time.time()
                del index, data
                torch.cuda.empty_cache()
                data = gen(args.rows, 1234)
                queries = gen(args.batch, 4321)  # make_step() closes over `queries`, `data`, `index`
                index = ivf_pq.build(ip, data, resources=res)
                res.sync()
                b2 = time.time() - t0
                e2, s2, n2, _ = timed(make_step(args.lut, args.acc), 5, 2)
                ph2 = {k_: round(v_, 3) for k_, v_ in phase_ms.items()}
                truth2 = exact_topk_fp64(data, queries[:ngv], args.k).cpu().numpy()
                rec2 = recall_of(neighbors[:ngv].cpu().numpy(), truth2)
                c1 = survivors_per_pair(index, queries)
                # the same step with the tail phase on the LUT scan kernels (the path every shape falls back to): the two-phase path
                # must not be slower than this on any corpus
                res_lut = comparator_handle(CUVS_AMD_PQ_SCAN3=0)
                e3, _, _, _ = timed(make_step(args.lut, args.acc, res_lut), 2, 1)
                del res_lut
                corpus_variants.append({"corpus": name, "ms_per_step": round(e2 / 5 * 1e3, 3), "qps": round(args.batch / (e2 / 5), 1),
                                        "recall_at_10": round(rec2, 4), "kernel_ms_per_step": round(s2 / 5, 3), "phase_ms_per_step": ph2,
                                        "ms_per_step_lut_scan_kernels": round(e3 / 2 * 1e3, 3), "gen_and_build_seconds": round(b2, 1), **c1})
                log(f"corpus variant '{name[:40]}': {e2 / 5 * 1e3:.2f} ms per step, recall {rec2:.4f}")
                if name.startswith("SURVEY 8d"):
                    # the metric's operating point ON THE CONTRACT CORPUS: the smallest refine ratio of the reference's bench grid
                    # style sweep (cuvs_ivf_pq.yaml: refine_ratio) that reaches recall@10 >= 0.9 at the metric's n_probes; every
                    # point = the whole step (search of k * ratio candidates + cuvsRefine), timed like the headline
                    contract_corpus = contract_point(name, index, data, queries, truth2, ngv, (args.refine_ratio, 4, 6, 8, 16))
                elif name.startswith("isotropic"):
                    corpus_variants[-1]["best_reachable"] = (
                        "recall@10 0.18 at n_probes 128 and 0.27 at n_probes 256 for EVERY refine ratio up to 32 (profiles/r06_contract_sweep.log: "
                        "the candidates' own recall equals the refined one): without cluster structure 128 of 16384 lists hold 18 % of the true "
                        "neighbours - the coarse quantizer, not the PQ ranking, is what misses them; no operating point of this index reaches 0.9")
            except Exception as e:
                corpus_variants.append({"corpus": name, "error": repr(e)[:300]})
            torch.cuda.empty_cache()
        queries = q_save

    # ------------------------------------------------------------------ C1 / C2 / C4 lines + CPU baseline (rank 0, N=1)
    extra, cpu = [], None
    if rank == 0 and world == 1:
        try:
            del index, data
        except NameError:  # (the second-corpus block failed between dropping and rebuilding them)
            pass
        torch.cuda.empty_cache()
        c1_x = c1_q = None
        if not args.no_extras:
            # C4's corpus (round 6): 4096 OVERLAPPING modes (spread 0.7: a mode's radius is about the distance to its nearest modes) in
            # the 24-d latent space - a multi-modal corpus on which recall means something; the single cloud of rounds 2-5 (the
            # easiest corpus for a graph walk) and the 4096 TIGHT modes on which no walk from random seeds leaves its mode are the
            # two side lines (2M rows each, profiles/r06_c4_corpus_sweep.log has the whole family)
            for name, fn in (("C1", lambda: extra_c1(res, dev)), ("C2", lambda: extra_c2(res, dev)), ("PQ-768", lambda: extra_pq768(res, dev)), ("Flat-768", lambda: extra_flat768(res, dev)),
                             ("CAGRA-128-build", lambda: extra_cagra128(res, dev)),
                             ("C4", lambda: extra_c4(res, dev, args.c4_rows, args.c4_latent, modes=args.c4_modes, spread=args.c4_spread)),
                             ("C4-corpus-family", lambda: extra_c4_family(res, dev))):
                t0 = time.time()
                try:
                    out = fn()
                    if name == "C1":
                        out, c1_x, c1_q = out
                    extra.append(out)
                except Exception as e:  # an extra line must not take the headline down
                    extra.append({"config": name, "error": repr(e)[:300]})
                torch.cuda.empty_cache()
                log(f"{name} done in {time.time() - t0:.1f}s")
        if not args.no_cpu_baseline:
            cpu = cpu_baseline_line(c1_x, c1_q, dev)
    elif rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline_line(dev=dev)  # N > 1: the same bounded CPU leg on rank 0's host cores (the other ranks wait below)

    if rank == 0:
        total_q = nq_total * args.steps
        out = {
            "metric": "QPS @ recall@10>=0.9, 100Mx128 fp32 IVF-PQ, batch=10k",
            "value": round(total_q / elapsed, 1),
            "unit": "queries/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": f"u8 codes, {args.lut} LUT, {args.acc} score",
            "data": "synthetic",
            "config": {"workload": f"IVF-PQ {args.rows}x{args.dim} fp32, pq_dim={args.pq_dim} pq_bits=8 "
                                   f"n_lists={args.n_lists} n_probes={args.n_probes} batch={args.batch} k={args.k}; rows and held-out queries "
                                   f"from bench.gen_rows (seed 1234 / 4321): Gaussian mixture of 65536 modes ~ N(0, I) in a 32-d latent space, "
                                   f"spread 0.35 per mode, embedded in R^{args.dim} by a fixed random map + N(0, 0.03^2 I) - NOT SURVEY 8d's "
                                   f"generator: the same step on that one is the top-level entry contract_corpus",
                       "parallelism": (f"list-sharded index (lists dealt to the {world} ranks by size, LPT), "
                                       + (f"ONE batch of {args.batch} queries per step searched by every rank on its lists (strong scaling), "
                                          if strong else f"{world} x {args.batch} queries per step, ")
                                       + "one native RCCL all-gather of the [Q,k] blocks per step, every rank refines its slice of the batch") if sharded
                                      else "single GPU",
                       "lut_dtype": args.lut, "internal_distance_dtype": args.acc, "refine_ratio": args.refine_ratio,
                       "build_seconds": round(build_s, 1), "variants": variants, "metric_variants": metric_variants,
                       "sharded_one_rank": sharded_line, "corpus_variants": corpus_variants, "batch_sweep": batch_sweep},
            "recall_at_10": round(recall, 4),
            "contract_corpus": contract_corpus,
            "scan3_equals_lut_scan": scan3_equals_lut_scan,
            "roofline": roofline,
            "cpu_baseline": cpu,
            "extra": extra,
        }
        if sharded:
            out["config"]["all_gather_ms_per_step"] = round(ag_ms / max(args.steps, 1), 3)
            out["config"]["transport"] = ("host-staged (mapped file): ranks share devices - a FUNCTIONAL run of the N-rank path, not a "
                                          "scaling figure") if shared_dev else "RCCL (ncclAllGather / ncclAllReduce over xGMI)"
            out["config"]["oversubscribed"] = bool(shared_dev)
            out["config"]["devices_visible"] = torch.cuda.device_count()
        # RCCL writes its start-up banner through C stdio (the one-rank sharded line initialises a communicator): it goes to stderr,
        # the JSON line is the ONLY line on stdout
        emit_json(out)
    if world > 1:
        dist.barrier()  # rank 0 may still be in its CPU leg: nobody tears the communicators down under it
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
