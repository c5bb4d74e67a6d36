"""GPU: cuvsRefine vs its CPU twin (bit-exact) and vs exact kNN (refining ALL rows must equal exact search)."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("metric", ["sqeuclidean", "euclidean", "inner_product", "cosine"])
@pytest.mark.parametrize("n,d,m,n_cand,k", [(2000, 64, 50, 40, 10), (500, 7, 20, 100, 100), (3000, 200, 10, 33, 5)])
def test_refine_matches_oracle(metric, n, d, m, n_cand, k):
    import torch
    from cuvs_amd.neighbors import refine

    rng = np.random.default_rng(n + d)
    x = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((m, d)).astype(np.float32)
    cand = rng.integers(0, n, size=(m, n_cand)).astype(np.int64)
    cand[0, 0] = -1  # invalid ids stay in the list at distance = max (refine_host.hpp:440-442)
    cand[1, 1] = n + 5
    gd, gi = refine(torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda(), torch.from_numpy(cand).cuda(), k=k,
                    metric=metric)
    torch.cuda.synchronize()
    od, oi = oracle.refine(x, q, cand, k, metric=metric)
    assert (gi.cpu().numpy() == oi).all()
    assert (gd.cpu().numpy() == od).all()


def test_refine_all_rows_is_exact_search():
    import torch
    from cuvs_amd.neighbors import refine

    rng = np.random.default_rng(0)
    x = rng.standard_normal((512, 32)).astype(np.float32)
    q = rng.standard_normal((16, 32)).astype(np.float32)
    cand = np.tile(np.arange(512, dtype=np.int64), (16, 1))
    gd, gi = refine(torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda(), torch.from_numpy(cand).cuda(), k=10)
    torch.cuda.synchronize()
    td, ti = oracle.exact_knn(q, x, 10)
    assert oracle.recall(gi.cpu().numpy(), ti) == 1.0
    np.testing.assert_allclose(gd.cpu().numpy(), td, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("metric", ["sqeuclidean", "inner_product", "cosine"])
@pytest.mark.parametrize("dtype", [np.float32, np.float16, np.int8])
def test_refine_host_tensors(metric, dtype):
    """All tensors in host memory -> the reference dispatches to refine_host (c/src/neighbors/refine.cpp,
    refine_host.hpp:353-462); here the same arithmetic as the device kernel runs on host threads: identical results."""
    import torch
    from cuvs_amd.neighbors import refine

    rng = np.random.default_rng(3)
    if dtype == np.int8:
        x = rng.integers(-30, 30, size=(1500, 48)).astype(dtype)
        q = rng.integers(-30, 30, size=(40, 48)).astype(dtype)
    else:
        x = rng.standard_normal((1500, 48)).astype(dtype)
        q = rng.standard_normal((40, 48)).astype(dtype)
    cand = rng.integers(0, 1500, size=(40, 50)).astype(np.int64)
    cand[0, 0] = -1
    hd, hi = refine(x, q, cand, k=10, metric=metric)                      # numpy in, numpy out: host path
    gd, gi = refine(torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda(), torch.from_numpy(cand).cuda(), k=10, metric=metric)
    torch.cuda.synchronize()
    assert isinstance(hi, np.ndarray)
    assert (hi == gi.cpu().numpy()).all() and (hd == gd.cpu().numpy()).all()
    od, oi = oracle.refine(x.astype(np.float32), q.astype(np.float32), cand, 10, metric=metric)
    assert (hi == oi).all() and (hd == od).all()


def test_refine_mixed_memory_is_an_error():
    import ctypes as C

    import torch
    import cuvs_amd
    from cuvs_amd._lib import Tensor, lib

    res = cuvs_amd.common.Resources()
    x = np.zeros((10, 4), np.float32)
    q = torch.zeros((2, 4), device="cuda")
    cand = np.zeros((2, 3), np.int64)
    oi, od = np.zeros((2, 2), np.int64), np.zeros((2, 2), np.float32)
    ts = [Tensor(t) for t in (x, q, cand, oi, od)]
    rc = lib().cuvsRefine(res.get_c_obj(), ts[0].ptr, ts[1].ptr, ts[2].ptr, C.c_int(0), ts[3].ptr, ts[4].ptr)
    assert rc == 0  # CUVS_ERROR
    lib().cuvsGetLastErrorText.restype = C.c_char_p
    assert b"all in device memory or all in host memory" in lib().cuvsGetLastErrorText()


def _host_refine_rule(x, q, cand, k, metric):
    """refine_host.hpp:430-460 transcribed with numpy, float64 distances only for RANKING real rows apart from invalid ones:
    every candidate whose id is outside [0, n) gets distance = max and keeps its id; tuples sorted by (distance, id)."""
    fmax = np.finfo(np.float32).max
    out_i = np.empty((q.shape[0], k), np.int64)
    tail = np.empty((q.shape[0], k), bool)
    for r in range(q.shape[0]):
        ids = cand[r]
        bad = (ids < 0) | (ids >= x.shape[0])
        n_good = int((~bad).sum())
        order_bad = np.sort(ids[bad])
        out_i[r, max(0, n_good):] = order_bad[:k - n_good] if n_good < k else []
        tail[r] = np.arange(k) >= n_good
    post = {"sqeuclidean": fmax, "euclidean": np.sqrt(np.float32(fmax)), "inner_product": -fmax, "cosine": fmax}[metric]
    return out_i, tail, np.float32(post)


@pytest.mark.parametrize("metric", ["sqeuclidean", "euclidean", "inner_product", "cosine"])
@pytest.mark.parametrize("host", [False, True])
def test_refine_keeps_out_of_range_candidates_at_max_distance(metric, host):
    """the reference's host refine gives a candidate id outside [0, n) distance = max and sorts it with its OWN id behind the real
    rows (refine_host.hpp:440-442, postprocess :465-505): -1, n, INT64_MAX (the padding of an IVF search) all come back as they
    went in; oracle, device path and host path agree on ids and distances"""
    import torch
    from cuvs_amd.neighbors import refine

    rng = np.random.default_rng(5)
    n, d, m, n_cand, k = 300, 24, 12, 16, 16
    x = rng.standard_normal((n, d)).astype(np.float32)
    q = rng.standard_normal((m, d)).astype(np.float32)
    cand = rng.integers(0, n, size=(m, n_cand)).astype(np.int64)
    cand[0, 3] = -1
    cand[1, [0, 5, 9]] = [np.iinfo(np.int64).max, -7, n]
    cand[2, :] = np.iinfo(np.int64).max           # nothing but padding
    cand[3, 10:] = np.iinfo(np.int64).max
    od, oi = oracle.refine(x, q, cand, k, metric=metric)
    want_i, tail, post = _host_refine_rule(x, q, cand, k, metric)
    assert (oi[tail] == want_i[tail]).all() and (od[tail] == post).all(), "oracle vs the reference's rule"
    assert ((oi[~tail] >= 0) & (oi[~tail] < n)).all()
    if host:
        gd, gi = refine(x, q, cand, k=k, metric=metric)
    else:
        gd, gi = refine(torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda(), torch.from_numpy(cand).cuda(), k=k, metric=metric)
        torch.cuda.synchronize()
        gd, gi = gd.cpu().numpy(), gi.cpu().numpy()
    assert (gi == oi).all()
    assert (gd == od).all()
