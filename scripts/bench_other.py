"""Side measurements for DESIGN.md (not the driver's bench): brute force C1, IVF-Flat C2, CAGRA C4-scaled.
usage: python scripts/bench_other.py [bf] [flat] [cagra] [--cagra-rows N]"""
import ctypes as C, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import cuvs_amd
from cuvs_amd._lib import lib
from cuvs_amd.neighbors import brute_force, cagra, ivf_flat

dev = torch.device("cuda", 0)
res = cuvs_amd.common.Resources()
what = [a for a in sys.argv[1:] if a in ("bf", "flat", "cagra")] or ["bf", "flat", "cagra"]
cagra_rows = int(sys.argv[sys.argv.index("--cagra-rows") + 1]) if "--cagra-rows" in sys.argv else 1_000_000
cagra_latent = int(sys.argv[sys.argv.index("--cagra-latent") + 1]) if "--cagra-latent" in sys.argv else 64  # intrinsic dimension of the cloud


def timeit(fn, steps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / steps


def recall(found, truth):
    return float(np.mean([len(np.intersect1d(f, t)) for f, t in zip(found, truth)])) / truth.shape[1]


if "bf" in what:  # C1: 100k x 128 fp32, batch 1k, k=10
    x = bench.gen_rows(100_000, 128, 1234, dev); q = bench.gen_rows(1000, 128, 4321, dev)
    idx = brute_force.build(x, resources=res)
    dt = timeit(lambda: brute_force.search(idx, q, 10, resources=res))
    print(json.dumps({"case": "brute_force 100k x128 batch1k k10", "ms": dt * 1e3, "qps": 1000 / dt,
                      "tflops": 2 * 1000 * 100_000 * 128 / dt / 1e12}))
    x2 = bench.gen_rows(10_000_000, 128, 1234, dev); q2 = bench.gen_rows(10000, 128, 4321, dev)
    idx2 = brute_force.build(x2, resources=res)
    dt = timeit(lambda: brute_force.search(idx2, q2, 10, resources=res), steps=2, warm=1)
    print(json.dumps({"case": "brute_force 10M x128 batch10k k10", "ms": dt * 1e3, "qps": 10000 / dt,
                      "tflops": 2 * 10000 * 1e7 * 128 / dt / 1e12}))
    del idx, idx2, x2

if "flat" in what:  # C2: 10M x 128 fp32, nlist 4096, nprobe 64, batch 10k
    x = bench.gen_rows(10_000_000, 128, 1234, dev); q = bench.gen_rows(10000, 128, 4321, dev)
    t0 = time.time(); idx = ivf_flat.build(ivf_flat.IndexParams(n_lists=4096, kmeans_trainset_fraction=0.1), x, resources=res); res.sync()
    build_s = time.time() - t0
    sp = ivf_flat.SearchParams(n_probes=64)
    nb = torch.empty((10000, 10), dtype=torch.int64, device=dev); dd = torch.empty((10000, 10), dtype=torch.float32, device=dev)
    lib().cuvsAmdProfileEnable(1)
    dt = timeit(lambda: ivf_flat.search(sp, idx, q, 10, neighbors=nb, distances=dd, resources=res))
    lib().cuvsAmdProfileEnable(0)
    ms = C.c_double(0); n = lib().cuvsAmdProfileCollect(b"ivf_flat_scan_kernel", C.byref(ms))
    bf = brute_force.build(x, resources=res); _, gt = brute_force.search(bf, q[:1000], 10, resources=res); res.sync()
    r = recall(nb[:1000].cpu().numpy(), gt.cpu().numpy())
    logical = 64 * (10_000_000 / 4096) * 512 * 10000
    print(json.dumps({"case": "ivf_flat 10M x128 nlist4096 nprobe64 batch10k k10", "ms": dt * 1e3, "qps": 10000 / dt,
                      "recall": r, "build_s": build_s, "scan_ms_per_search": ms.value / 7, "scan_launches": n,
                      "logical_TBps": logical / (ms.value / 7 * 1e-3) / 1e12}))  # timeit: 2 warm-up + 5 timed searches
    del idx, bf, x

if "cagra" in what:  # C4 scaled: N x 768 fp16, degree 64, itopk 64, batch 10k
    n = cagra_rows
    x = bench.gen_rows(n, 768, 1234, dev, latent=cagra_latent, n_modes=1).half(); q = bench.gen_rows(10000, 768, 4321, dev, latent=cagra_latent, n_modes=1).half()  # one broad mode: the kNN graph of well-separated tight modes is disconnected and no graph walk from random seeds can cross modes
    t0 = time.time(); idx = cagra.build(cagra.IndexParams(intermediate_graph_degree=128, graph_degree=64), x, resources=res); res.sync()
    build_s = time.time() - t0
    sp = cagra.SearchParams(itopk_size=64, algo=os.environ.get("CAGRA_ALGO", "auto"))
    nb = torch.empty((10000, 10), dtype=torch.int32, device=dev); dd = torch.empty((10000, 10), dtype=torch.float32, device=dev)
    dt = timeit(lambda: cagra.search(sp, idx, q, 10, neighbors=nb, distances=dd, resources=res))
    bf = brute_force.build(x, resources=res); _, gt = brute_force.search(bf, q[:1000], 10, resources=res); res.sync()
    r = recall(nb[:1000].cpu().numpy().astype(np.int64) & 0xFFFFFFFF, gt.cpu().numpy())
    print(json.dumps({"case": f"cagra {n} x768 fp16 (latent {cagra_latent}) degree64 itopk64 batch10k k10", "ms": dt * 1e3, "qps": 10000 / dt,
                      "recall": r, "build_s": build_s}))
