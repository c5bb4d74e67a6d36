"""GPU: the radix select_k kernel vs oracle.select_k (exact values, indices and tie rule)."""
import ctypes as C

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _gpu_select(vals, k, select_min=True, in_idx=None):
    import torch
    import cuvs_amd
    from cuvs_amd._lib import check, lib

    res = cuvs_amd.common.Resources()
    tv = torch.from_numpy(vals).cuda()
    ti = torch.from_numpy(in_idx).cuda() if in_idx is not None else None
    ov = torch.empty((vals.shape[0], k), dtype=torch.float32, device="cuda")
    oi = torch.empty((vals.shape[0], k), dtype=torch.int64, device="cuda")
    fn = lib().cuvsAmdSelectK
    fn.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    check(fn(res.get_c_obj(), tv.data_ptr(), ti.data_ptr() if ti is not None else None, vals.shape[0], vals.shape[1],
             k, ov.data_ptr(), oi.data_ptr(), int(select_min)))
    res.sync()
    return ov.cpu().numpy(), oi.cpu().numpy()


@pytest.mark.parametrize("rows,ln,k", [(7, 100, 10), (64, 4096, 64), (33, 16384, 128), (5, 100000, 10),
                                       (16, 1280, 10), (3, 257, 256), (4, 5000, 1000), (9, 12, 16), (2, 1, 1)])
@pytest.mark.parametrize("select_min", [True, False])
def test_select_k_random(rows, ln, k, select_min):
    rng = np.random.default_rng(rows * ln + k)
    v = rng.standard_normal((rows, ln)).astype(np.float32)
    gv, gi = _gpu_select(v, k, select_min)
    ov, oi = oracle.select_k(v, k, select_min)
    assert (gi == oi).all()
    assert (gv == ov).all()


def test_select_k_ties_and_specials():
    rng = np.random.default_rng(0)
    v = rng.integers(0, 8, size=(20, 3000)).astype(np.float32)  # heavy ties: earliest positions win
    v[0, :] = 1.0
    v[1, ::2] = -0.0
    v[2, 5] = np.inf
    v[3, 7] = -np.inf
    for smin in (True, False):
        gv, gi = _gpu_select(v, 50, smin)
        ov, oi = oracle.select_k(v, 50, smin)
        assert (gi == oi).all() and (gv == ov).all()


def test_select_k_with_input_indices():
    rng = np.random.default_rng(4)
    v = rng.standard_normal((12, 640)).astype(np.float32)
    ids = rng.integers(0, 1 << 40, size=(12, 640)).astype(np.int64)
    gv, gi = _gpu_select(v, 10, True, ids)
    ov, oi = oracle.select_k(v, 10, True, in_idx=ids)
    assert (gi == oi).all() and (gv == ov).all()


@pytest.mark.parametrize("rows,ln,k", [(3, 20000, 4096), (2, 30000, 8192), (2, 50000, 20000), (1, 3000, 3000)])
def test_select_k_large_k(rows, ln, k):
    """k beyond 2048 (the reference's select_k has no bound): winners in LDS up to 8192, in global scratch beyond."""
    rng = np.random.default_rng(k)
    v = rng.standard_normal((rows, ln)).astype(np.float32)
    v[0, ::7] = 0.25  # ties across the k-th value
    gv, gi = _gpu_select(v, k, True)
    ov, oi = oracle.select_k(v, k, True)
    assert (gi == oi).all() and (gv == ov).all()


@pytest.mark.parametrize("rows,ln,k", [(40, 16384, 128), (17, 4096, 64), (9, 8192, 256), (5, 12288, 8), (6, 16380, 100)])
@pytest.mark.parametrize("select_min", [True, False])
def test_select_k_one_read_kernel(rows, ln, k, select_min):
    """Rows of 4096 .. 16384 elements and k <= 256 (the coarse searches): select_k_minima_kernel holds the row in registers
    and bounds the k-th key by the k-th smallest of 512 group minima. Same values, indices and tie rule as the oracle on
    random rows, on rows with masses of equal keys at the k-th value (the radix kernel takes those rows over) and with
    infinities / signed zeros."""
    rng = np.random.default_rng(ln + k)
    v = rng.standard_normal((rows, ln)).astype(np.float32)
    v[0, :] = 3.0                                   # one value everywhere: every element ties at the bound
    v[1] = rng.integers(0, 4, size=ln)              # four distinct values
    v[2, ::3] = -0.0
    v[2, 1::3] = 0.0
    v[3, :200] = np.inf
    v[4, 100:400] = -np.inf
    v[3, 7::11] = v[3, 5]                           # a few hundred ties somewhere in the row
    gv, gi = _gpu_select(v, k, select_min)
    ov, oi = oracle.select_k(v, k, select_min)
    assert (gi == oi).all(), f"first mismatching row {np.nonzero((gi != oi).any(1))[0][:3]}"
    assert (gv == ov).all()
