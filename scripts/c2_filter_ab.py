"""C2 (IVF-Flat 10M x 128 fp32, n_lists 4096, n_probes 64, 10k queries, k 10): the tail phase's filter with 256-query units per
workgroup and the B operands in LDS (flat_filter2_kernel, round 6) against round 3's 64-query units per wave (CUVS_AMD_FLAT_FILTER2=0):
ms per search, per-kernel HIP-event times, ids / distances equal. Usage: python scripts/c2_filter_ab.py [rows] [dtype: f32|i8|f16]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import cuvs_amd  # noqa: E402
from cuvs_amd._lib import lib  # noqa: E402
from cuvs_amd.neighbors import ivf_flat  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dt_name = sys.argv[2] if len(sys.argv) > 2 else "f32"
dev = torch.device("cuda", 0)
nq = 10000
x = bench.gen_rows(n, 128, 1234, dev)
q = bench.gen_rows(nq, 128, 4321, dev)
if dt_name == "i8":
    x, q = torch.clamp(torch.round(x * 24.0), -128, 127).to(torch.int8), torch.clamp(torch.round(q * 24.0), -128, 127).to(torch.int8)
elif dt_name == "f16":
    x, q = x.half(), q.half()
res = cuvs_amd.common.Resources()
idx = ivf_flat.build(ivf_flat.IndexParams(n_lists=4096, kmeans_trainset_fraction=0.1), x, resources=res)
res.sync()
sp = ivf_flat.SearchParams(n_probes=64)
nb = torch.empty((nq, 10), dtype=torch.int64, device=dev)
dd = torch.empty((nq, 10), dtype=torch.float32, device=dev)
ref = None
cases = [("bound-only head + flat_filter2_kernel (default)", res),
         ("exact head on the scan kernel + flat_filter2_kernel", bench.comparator_handle(CUVS_AMD_FLAT_BOUND_HEAD=0)),
         ("exact head + pq_filter_kernel<FLAT> (rounds 3-5)", bench.comparator_handle(CUVS_AMD_FLAT_FILTER2=0)),
         ("default again", res)]
for name, r in cases:
    step = lambda: ivf_flat.search(sp, idx, q, 10, neighbors=nb, distances=dd, resources=r)
    for _ in range(3):
        step()
    dt = bench.timeit(step, 10, 2)
    lib().cuvsAmdProfileEnable(1)
    for _ in range(5):
        step()
    r.sync(); torch.cuda.synchronize()
    lib().cuvsAmdProfileEnable(0)
    ph = {}
    for nm in (b"ivf_flat_scan_kernel", b"flat_head_kernel", b"flat_filter_kernel", b"flat_rescore_kernel"):
        v = C.c_double(0)
        lib().cuvsAmdProfileCollect(nm, C.byref(v))
        ph[nm.decode()] = round(v.value / 5, 3)
    if ref is None:
        ref = (nb.clone(), dd.clone())
    same = bool(torch.equal(ref[0], nb) and torch.equal(ref[1], dd))
    print(f"{name}: {dt * 1e3:.3f} ms per search, kernels {ph}, same_as_first {same}", flush=True)
