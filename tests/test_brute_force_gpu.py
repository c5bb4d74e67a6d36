"""GPU: cuvsBruteForce* through the C ABI vs the oracle — bit-exact ids and distances.

Shapes follow the reference's brute-force tests (cpp/tests/neighbors/ann_brute_force.cuh:167-197,
python/cuvs/cuvs/tests/test_brute_force.py) plus edge cases (odd dims, k > 64, tiling, filters).
"""
import os

import numpy as np
import pytest

import oracle
from tests.golden import reference_fixtures as G

pytestmark = pytest.mark.gpu


def _gen(n, d, q, seed, lo=0.1, hi=2.0):
    # reference generator: uniform [0.1, 2.0) (ann_brute_force.cuh:142-155); stream differs, range same
    rng = np.random.default_rng(seed)
    x = (rng.random((n, d), dtype=np.float32) * (hi - lo) + lo).astype(np.float32)
    qq = (rng.random((q, d), dtype=np.float32) * (hi - lo) + lo).astype(np.float32)
    return x, qq


def _run(x, q, k, metric, prefilter=None, dtype="float32"):
    import torch
    from cuvs_amd.neighbors import brute_force

    tx = torch.from_numpy(x).cuda()
    tq = torch.from_numpy(q).cuda()
    if dtype == "float16":
        tx, tq = tx.half(), tq.half()
    idx = brute_force.build(tx, metric=metric)
    d, i = brute_force.search(idx, tq, k, prefilter=prefilter)
    torch.cuda.synchronize()
    return d.cpu().numpy(), i.cpu().numpy()


@pytest.mark.parametrize("metric", ["sqeuclidean", "euclidean", "cosine", "inner_product"])
@pytest.mark.parametrize("n,d,q,k", [(1000, 32, 100, 10), (10000, 128, 200, 64), (777, 17, 33, 5),
                                     (500, 1, 20, 3), (2049, 100, 129, 100), (300, 3, 7, 1)])
def test_bit_exact_vs_oracle(metric, n, d, q, k):
    x, qq = _gen(n, d, q, seed=n + d)
    gd, gi = _run(x, qq, k, metric)
    od, oi = oracle.brute_force_knn(qq, x, k, metric=metric)
    assert (gi == oi).all(), f"index mismatch rate {(gi != oi).mean()}"
    assert (gd == od).all(), f"distance max abs diff {np.abs(gd - od).max()}"


def test_bit_exact_at_the_baseline_config0_shape():
    """BASELINE configs[0]: brute force L2, 100k x 128 fp32, batch 1k, k = 10 - ids AND distances bit-identical to the CPU
    restatement at exactly that size (the fused threshold-epilogue path: n >= 65536), on bench.py's own corpus generator too."""
    x, qq = _gen(100_000, 128, 1000, seed=2024)
    gd, gi = _run(x, qq, 10, "sqeuclidean")
    od, oi = oracle.brute_force_knn(qq, x, 10)
    assert (gi == oi).all() and (gd == od).all()


def test_recall_vs_reference_cpu_path():
    # vs the refine_host restatement (unexpanded arithmetic): ids may differ only on fp ties
    x, qq = _gen(5000, 64, 100, seed=5)
    gd, gi = _run(x, qq, 16, "sqeuclidean")
    rd, ri = oracle.exact_knn(qq, x, 16)
    assert oracle.recall(gi, ri) > 0.999
    np.testing.assert_allclose(gd, rd, atol=1e-3, rtol=1e-3)  # the reference's own eps (ann_brute_force.cuh:97-105)


def test_half_dataset():
    x, qq = _gen(3000, 64, 50, seed=9)
    xh, qh = x.astype(np.float16), qq.astype(np.float16)
    gd, gi = _run(xh.astype(np.float32), qh.astype(np.float32), 10, "sqeuclidean", dtype="float16")
    od, oi = oracle.brute_force_knn(qh.astype(np.float32), xh.astype(np.float32), 10, clamp_eps=1e-3)
    assert (gi == oi).all() and (gd == od).all()


def test_golden_vectors_and_kat():
    gd, gi = _run(G.CAGRA_C_DATASET, G.CAGRA_C_QUERIES, 1, "sqeuclidean")
    assert (gi[:, 0] == G.CAGRA_C_NEIGHBORS).all()
    np.testing.assert_allclose(gd[:, 0], G.CAGRA_C_DISTANCES, atol=G.CAGRA_C_TOL)
    _, gi = _run(G.BF_KAT_POINTS, G.BF_KAT_POINTS, G.BF_KAT_K, "sqeuclidean")
    assert (G.BF_KAT_LABELS[gi] == G.BF_KAT_LABELS[:, None]).all()


def test_bitset_and_bitmap_prefilter():
    import torch
    from cuvs_amd._lib import BITMAP, BITSET

    x, qq = _gen(1000, 16, 40, seed=11)
    rng = np.random.default_rng(1)
    keep = rng.random(1000) < 0.3
    words = np.packbits(keep, bitorder="little").view(np.uint32) if keep.size % 32 == 0 else None
    if words is None:
        pad = np.zeros((-keep.size) % 32, bool)
        words = np.packbits(np.concatenate([keep, pad]), bitorder="little").view(np.uint32)
    tw = torch.from_numpy(words.view(np.int32)).cuda()
    gd, gi = _run(x, qq, 8, "sqeuclidean", prefilter=(tw, BITSET))
    od, oi = oracle.brute_force_knn(qq, x, 8, keep_bits=words)
    assert (gi == oi).all() and (gd == od).all()
    assert keep[gi].all()
    # golden filtered vector (ann_cagra_c.cu:39-44)
    tw = torch.from_numpy(G.CAGRA_C_FILTER_WORDS.view(np.int32)).cuda()
    gd, gi = _run(G.CAGRA_C_DATASET, G.CAGRA_C_QUERIES, 1, "sqeuclidean", prefilter=(tw, BITSET))
    assert (gi[:, 0] == G.CAGRA_C_NEIGHBORS_FILTERED).all()
    # bitmap: per-query keep masks
    keep2 = rng.random((40, 1000)) < 0.5
    flat = keep2.reshape(-1)
    pad = np.zeros((-flat.size) % 32, bool)
    words2 = np.packbits(np.concatenate([flat, pad]), bitorder="little").view(np.uint32)
    tw2 = torch.from_numpy(words2.view(np.int32)).cuda()
    gd, gi = _run(x, qq, 8, "sqeuclidean", prefilter=(tw2, BITMAP))
    od, oi = oracle.brute_force_knn(qq, x, 8, keep_bits=words2, bitmap=True)
    assert (gi == oi).all() and (gd == od).all()


def test_column_tiling_and_merge(monkeypatch):
    # shrink the workspace so the dataset is cut into many column tiles + a merge select_k
    import torch
    import cuvs_amd
    from cuvs_amd.neighbors import brute_force

    monkeypatch.setenv("CUVS_AMD_WORKSPACE_MB", "1")
    res = cuvs_amd.common.Resources()
    x, qq = _gen(20000, 32, 300, seed=21)
    tx, tq = torch.from_numpy(x).cuda(), torch.from_numpy(qq).cuda()
    idx = brute_force.build(tx, resources=res)
    d, i = brute_force.search(idx, tq, 10, resources=res)
    res.sync()
    od, oi = oracle.brute_force_knn(qq, x, 10)
    assert (i.cpu().numpy() == oi).all() and (d.cpu().numpy() == od).all()


def test_error_convention():
    import torch
    from cuvs_amd._lib import CuvsError
    from cuvs_amd.neighbors import brute_force

    x, qq = _gen(100, 8, 5, seed=1)
    idx = brute_force.build(torch.from_numpy(x).cuda())
    with pytest.raises(CuvsError):
        brute_force.search(idx, torch.from_numpy(qq[:, :4].copy()).cuda(), 3)  # dim mismatch
    with pytest.raises(CuvsError):
        brute_force.build(torch.from_numpy(x).cuda(), metric="l1")  # metric outside the hot path


@pytest.mark.parametrize("metric", ["sqeuclidean", "inner_product", "cosine"])
def test_fused_and_tiled_paths_agree(metric, monkeypatch):
    # CUVS_AMD_BF_FUSED=1 switches k <= 64 searches to the fused distance+top-k kernel (opt-in in round 1)
    x, qq = _gen(30000, 48, 333, seed=77)
    td, ti = _run(x, qq, 33, metric)
    monkeypatch.setenv("CUVS_AMD_BF_FUSED", "1")
    fd, fi = _run(x, qq, 33, metric)
    assert (fi == ti).all() and (fd == td).all()


def test_large_k():
    """k beyond 2048 (no bound in the reference): bit-identical to the oracle."""
    x, qq = _gen(9000, 24, 5, seed=13)
    gd, gi = _run(x, qq, 3000, "sqeuclidean")
    od, oi = oracle.brute_force_knn(qq, x, 3000)
    assert (gi == oi).all() and (gd == od).all()


@pytest.mark.parametrize("metric", ["sqeuclidean", "inner_product"])
@pytest.mark.parametrize("adversarial", [False, True])
def test_running_threshold_path(monkeypatch, metric, adversarial):
    """Datasets wider than one column tile: after the first tile only elements better than a row's current k-th value
    are kept (one pass per tile instead of select_k's four). Identical to the oracle - also when the columns arrive
    in improving order, so that every tile overflows the candidate buffer and falls back to select_k on the tile -
    and identical to the per-tile select + merge it replaces."""
    import torch
    import cuvs_amd
    from cuvs_amd.neighbors import brute_force

    monkeypatch.setenv("CUVS_AMD_WORKSPACE_MB", "1")  # 40 rows x 6400-column tiles
    res = cuvs_amd.common.Resources()
    x, qq = _gen(40000, 16, 40, seed=23)
    if adversarial:  # rows sorted by decreasing distance to the first query: every later tile is better than all before
        key = ((x - qq[0]) ** 2).sum(1) if metric == "sqeuclidean" else -(x @ qq[0])
        x = np.ascontiguousarray(x[np.argsort(-key)])
    x[100] = x[30000]  # exact ties across tiles: the earlier column must win
    tx, tq = torch.from_numpy(x).cuda(), torch.from_numpy(qq).cuda()
    idx = brute_force.build(tx, metric=metric, resources=res)
    d, i = brute_force.search(idx, tq, 12, resources=res)
    res.sync()
    od, oi = oracle.brute_force_knn(qq, x, 12, metric=metric)
    assert (i.cpu().numpy() == oi).all() and (d.cpu().numpy() == od).all()
    monkeypatch.setenv("CUVS_AMD_BF_NO_FUSED_FILTER", "1")  # round 2's first form: distance tiles + a filter pass
    d1, i1 = brute_force.search(idx, tq, 12, resources=res)
    res.sync()
    assert torch.equal(i, i1) and torch.equal(d, d1)
    monkeypatch.setenv("CUVS_AMD_BF_NO_THRESHOLD", "1")
    d2, i2 = brute_force.search(idx, tq, 12, resources=res)
    res.sync()
    assert torch.equal(i, i2) and torch.equal(d, d2)


@pytest.mark.parametrize("host_flags", [False, True])
def test_fused_path_overflow_is_redone_without_a_host_round_trip(monkeypatch, host_flags):
    """One wide column tile (n >= 65536, default workspace) whose columns arrive in improving order for some queries: their
    candidate buffers overflow in the fused epilogue, the row tile's flag is raised and the per-tile select pass - enqueued
    behind the fused one, every launch guarded by that flag on the device - redoes it; identical to the oracle, and to the
    same search with the flags read back by the host (CUVS_AMD_BF_HOST_FLAGS=1, rounds 2-3)."""
    import torch
    import cuvs_amd
    from cuvs_amd.neighbors import brute_force

    if host_flags:
        monkeypatch.setenv("CUVS_AMD_BF_HOST_FLAGS", "1")
    res = cuvs_amd.common.Resources()   # (switches are read when the handle is created)
    x, qq = _gen(90000, 16, 50, seed=29)
    key = ((x - qq[0]) ** 2).sum(1)
    x = np.ascontiguousarray(x[np.argsort(-key)])   # decreasing distance to query 0: every column beats all before it
    tx, tq = torch.from_numpy(x).cuda(), torch.from_numpy(qq).cuda()
    idx = brute_force.build(tx, metric="sqeuclidean", resources=res)
    for _ in range(2):   # twice: the counters and flags of the first call must not leak into the second
        d, i = brute_force.search(idx, tq, 10, resources=res)
        res.sync()
        od, oi = oracle.brute_force_knn(qq, x, 10)
        assert (i.cpu().numpy() == oi).all() and (d.cpu().numpy() == od).all()


@pytest.mark.parametrize("metric,dtype", [("sqeuclidean", "float32"), ("inner_product", "float32"),
                                          ("cosine", "float16"), ("euclidean", "float32")])
@pytest.mark.parametrize("filt", ["none", "bitset", "bitmap"])
def test_fused_threshold_epilogue(metric, dtype, filt):
    """n >= 65536: beyond the first 32768 columns no distance tile is written - the MFMA epilogue appends what beats a
    row's k-th value (pre-filter applied to those only) and a per-row sort by (value, id) keeps k. Identical to the
    oracle, ties included, with every kind of pre-filter."""
    import torch
    from cuvs_amd._lib import BITMAP, BITSET

    n, q, k = 90000, 70, 10
    x, qq = _gen(n, 24, q, seed=31)
    x[200] = x[60000]; x[70001] = x[40]  # exact ties between the first tile and the fused part, both directions
    if dtype == "float16":
        x, qq = x.astype(np.float16).astype(np.float32), qq.astype(np.float16).astype(np.float32)
    rng = np.random.default_rng(5)
    pre, kw = None, {}
    if filt == "bitset":
        keep = rng.random(n) < 0.4
        words = np.packbits(np.concatenate([keep, np.zeros((-n) % 32, bool)]), bitorder="little").view(np.uint32)
        pre, kw = (torch.from_numpy(words.view(np.int32)).cuda(), BITSET), {"keep_bits": words}
    elif filt == "bitmap":
        keep = (rng.random((q, n)) < 0.5).reshape(-1)
        words = np.packbits(np.concatenate([keep, np.zeros((-keep.size) % 32, bool)]), bitorder="little").view(np.uint32)
        pre, kw = (torch.from_numpy(words.view(np.int32)).cuda(), BITMAP), {"keep_bits": words, "bitmap": True}
    gd, gi = _run(x, qq, k, metric, prefilter=pre, dtype=dtype)
    od, oi = oracle.brute_force_knn(qq, x, k, metric=metric, **kw)
    assert (gi == oi).all() and (gd == od).all()


def test_fused_threshold_epilogue_more_than_one_row_tile():
    """More queries than one row tile (16384): the per-row buffers, counters and overflow flags are reused by the second
    row tile. Identical to the oracle."""
    x, qq = _gen(66000, 4, 17000, seed=41)
    gd, gi = _run(x, qq, 5, "sqeuclidean")
    od, oi = oracle.brute_force_knn(qq, x, 5)
    assert (gi == oi).all() and (gd == od).all()


def test_k_beyond_the_number_of_rows_is_padded():
    """The reference accepts k > n (knn_brute_force.cuh / ivf_flat_search.cuh have no such check): the first n slots are the
    exact ranking, the rest are padding (worst distance, an id that is no row)."""
    import torch
    from cuvs_amd.neighbors import brute_force, ivf_flat

    x, q = _gen(40, 16, 7, seed=9)
    idx = brute_force.build(torch.from_numpy(x).cuda())
    d, i = brute_force.search(idx, torch.from_numpy(q).cuda(), 64)
    torch.cuda.synchronize()
    d, i = d.cpu().numpy(), i.cpu().numpy()
    od, oi = oracle.brute_force_knn(q, x, 40)
    assert (i[:, :40] == oi).all() and (d[:, :40] == od).all()
    assert ((i[:, 40:] < 0) | (i[:, 40:] >= 40)).all()
    fidx = ivf_flat.build(ivf_flat.IndexParams(n_lists=4, kmeans_n_iters=5), torch.from_numpy(x).cuda())
    d2, i2 = ivf_flat.search(ivf_flat.SearchParams(n_probes=4), fidx, torch.from_numpy(q).cuda(), 64)
    torch.cuda.synchronize()
    i2 = i2.cpu().numpy()
    assert (np.sort(i2[:, :40], axis=1) == np.arange(40)[None, :]).all()   # every row once (all lists probed)
    assert ((i2[:, 40:] < 0) | (i2[:, 40:] >= 40)).all()
