"""Cycles per 128 x 128 tile of the grouped coarse GEMM (prologue / main loop / epilogue) at the C3 coarse shape, and the
kernel's time against the fp32 MFMA peak. usage: python scripts/coarse_tile_probe.py"""
import ctypes as C
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cuvs_amd._lib import lib

fn = lib().cuvsAmdPairwiseTopK
fn.argtypes = [C.c_size_t, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
fn.restype = C.c_int
for (m, n, dim, k) in ((10000, 16384, 128, 128), (1000, 100000, 128, 10)):
    q = torch.randn((m, dim), device="cuda"); x = torch.randn((n, dim), device="cuda")
    ov = torch.empty((m, k), device="cuda"); oi = torch.empty((m, k), dtype=torch.int32, device="cuda")
    for dbg in (0, 8, 4):
        r = bench.comparator_handle(CUVS_AMD_TILE_DBG=dbg)
        for grouped in (1, 0):
            if grouped and n % 1 != 0:
                continue
            for _ in range(3):
                rc = fn(r.get_c_obj(), q.data_ptr(), m, x.data_ptr(), n, dim, 0, k, ov.data_ptr(), oi.data_ptr(), grouped)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10):
                fn(r.get_c_obj(), q.data_ptr(), m, x.data_ptr(), n, dim, 0, k, ov.data_ptr(), oi.data_ptr(), grouped)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
            print(f"{m} x {n} x {dim}, k {k}, grouped {grouped}, dbg {dbg}, rc {rc}: {dt * 1e3:.3f} ms per call (GEMM + select + norms, host-synchronous hook), "
                  f"{2.0 * m * n * dim / dt / 1e12:.1f} TFLOP/s", flush=True)
