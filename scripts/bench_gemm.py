"""Micro-benchmark of the fp32-MFMA distance kernels (dist_mfma_kernel MODE 0 / MODE 1)."""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuvs_amd
from cuvs_amd._lib import check, lib

res = cuvs_amd.common.Resources()
pd = lib().cuvsAmdPairwiseDistance
pd.argtypes = [C.c_size_t, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p]
am = lib().cuvsAmdFusedArgmin
am.argtypes = [C.c_size_t, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]


def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n


for (m, n, d) in [(10000, 16384, 128), (16384, 65536, 128), (8192, 8192, 768), (4096, 4096, 4096)]:
    q = torch.randn(m, d, device="cuda"); x = torch.randn(n, d, device="cuda"); out = torch.empty(m, n, device="cuda")
    dt = t(lambda: check(pd(res.get_c_obj(), q.data_ptr(), m, x.data_ptr(), n, d, 6, out.data_ptr())))
    ref = q @ x.T
    err = (out - ref).abs().max().item()
    dt_t = t(lambda: torch.matmul(q, x.T, out=ref))
    print(f"pairwise IP m={m} n={n} d={d}: {dt*1e3:.3f} ms {2*m*n*d/dt/1e12:.1f} TF (torch/rocBLAS {2*m*n*d/dt_t/1e12:.1f} TF) maxerr {err:.2e}")
for (m, n, d) in [(2_000_000, 16384, 128), (2_000_000, 128, 128), (1_000_000, 4096, 96)]:
    q = torch.randn(m, d, device="cuda"); x = torch.randn(n, d, device="cuda"); lab = torch.empty(m, dtype=torch.int32, device="cuda")
    dt = t(lambda: check(am(res.get_c_obj(), q.data_ptr(), m, x.data_ptr(), n, d, lab.data_ptr())), n=2)
    print(f"fused argmin m={m} n={n} d={d}: {dt*1e3:.2f} ms {2*m*n*d/dt/1e12:.1f} TF")
