"""GPU: randomized differential tests against the CPU oracle. A fixed seed draws parameter combinations that the
hand-picked cases do not cover (odd dims, every pq_bits, k on both sides of 64 / 128 / 256, LUT / score types, metrics,
ragged lists); every result must be bit-identical to the oracle searching the same exported index."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

_LUTS = {"f32": np.float32, "f16": np.float16, "fp8": np.uint8}


def _draw_pq(rng):
    d = int(rng.choice([8, 17, 32, 48, 64, 96, 100, 128, 200]))
    pq_dim = int(rng.choice([p for p in (4, 8, 16, 24, 32, 48, 64, 96) if p <= max(4, d)]))
    return dict(n=int(rng.integers(1500, 5000)), d=d, n_lists=int(rng.choice([4, 9, 16, 33])), pq_dim=pq_dim,
                pq_bits=int(rng.choice([4, 5, 6, 7, 8])), k=int(rng.choice([1, 5, 10, 32, 64, 65, 100, 129, 200, 256, 300])),
                n_probes=int(rng.integers(1, 9)), metric=str(rng.choice(["sqeuclidean", "euclidean", "inner_product", "cosine"])),
                lut=str(rng.choice(["f32", "f16", "fp8"])), nq=int(rng.choice([3, 50, 300])),
                codebook=str(rng.choice(["subspace", "subspace", "cluster"])))


@pytest.mark.parametrize("case", range(64))
def test_ivf_pq_random_configuration(case):
    import torch
    from cuvs_amd.neighbors import ivf_pq

    rng = np.random.default_rng(1000 + case)
    c = _draw_pq(rng)
    acc = "f32" if c["lut"] == "f32" else str(rng.choice(["f32", "f16"]))
    x = (rng.random((c["n"], c["d"]), dtype=np.float32) * 1.9 + 0.1)
    x[: c["n"] // 3] += 3.0  # uneven lists
    q = (rng.random((c["nq"], c["d"]), dtype=np.float32) * 1.9 + 0.1)
    q[::2] += 3.0
    index = ivf_pq.build(ivf_pq.IndexParams(n_lists=c["n_lists"], metric=c["metric"], pq_dim=c["pq_dim"], pq_bits=c["pq_bits"],
                                            kmeans_n_iters=5, codebook_kind=c["codebook"]), torch.from_numpy(x).cuda())
    k = min(c["k"], c["n"])
    sp = ivf_pq.SearchParams(n_probes=c["n_probes"], lut_dtype=_LUTS[c["lut"]], internal_distance_dtype=_LUTS[acc])
    gd, gi = ivf_pq.search(sp, index, torch.from_numpy(q).cuda(), k)
    torch.cuda.synchronize()
    ex = ivf_pq.export_for_oracle(index, per_cluster=c["codebook"] == "cluster")
    od, oi = oracle.ivf_pq_search(ex, q, k, c["n_probes"], metric=c["metric"], lut=c["lut"], acc=acc)
    gd, gi = gd.cpu().numpy(), gi.cpu().numpy()
    assert (gi == oi).all(), f"{c} acc={acc}: id mismatch rate {(gi != oi).mean():.4f}"
    assert (gd == od).all(), f"{c} acc={acc}: max |d| diff {np.nanmax(np.abs(gd - od))}"


@pytest.mark.parametrize("case", range(40))
def test_ivf_flat_random_configuration(case):
    import torch
    from cuvs_amd.neighbors import ivf_flat

    rng = np.random.default_rng(2000 + case)
    d = int(rng.choice([1, 7, 16, 33, 64, 100, 257]))
    n, nq = int(rng.integers(1000, 4000)), int(rng.choice([2, 40, 200]))
    dtype = [np.float32, np.float16, np.int8, np.uint8][int(rng.integers(0, 4))]
    metric = str(rng.choice(["sqeuclidean", "euclidean", "inner_product", "cosine"]))
    k = int(rng.choice([1, 10, 64, 65, 128, 200, 300]))
    n_lists, n_probes = int(rng.choice([3, 8, 20])), int(rng.integers(1, 6))
    if dtype in (np.int8, np.uint8):
        lo, hi = (-100, 100) if dtype == np.int8 else (0, 200)
        x, q = rng.integers(lo, hi, size=(n, d)).astype(dtype), rng.integers(lo, hi, size=(nq, d)).astype(dtype)
        scale = 1 / 128 if dtype == np.int8 else 1 / 256
    else:
        x = (rng.random((n, d), dtype=np.float32) * 1.9 + 0.1).astype(dtype)
        q = (rng.random((nq, d), dtype=np.float32) * 1.9 + 0.1).astype(dtype)
        scale = 1.0
    index = ivf_flat.build(ivf_flat.IndexParams(n_lists=n_lists, metric=metric, kmeans_n_iters=5), torch.from_numpy(x).cuda())
    k = min(k, n)
    gd, gi = ivf_flat.search(ivf_flat.SearchParams(n_probes=n_probes), index, torch.from_numpy(q).cuda(), k)
    torch.cuda.synchronize()
    ex = ivf_flat.export_for_oracle(index, dtype)
    od, oi = oracle.ivf_flat_search(ex, q, k, n_probes, metric=metric, coarse_scale=scale)
    gd, gi = gd.cpu().numpy(), gi.cpu().numpy()
    tag = f"d={d} n={n} nq={nq} {np.dtype(dtype).name} {metric} k={k} lists={n_lists} probes={n_probes}"
    assert (gi == oi).all(), f"{tag}: id mismatch rate {(gi != oi).mean():.4f}"
    assert (gd == od).all(), f"{tag}: max |d| diff {np.nanmax(np.abs(gd - od))}"


@pytest.mark.parametrize("case", range(20))
def test_brute_force_random_configuration(case):
    import torch
    from cuvs_amd.neighbors import brute_force

    rng = np.random.default_rng(3000 + case)
    d = int(rng.choice([3, 16, 40, 64, 100, 128]))
    n, nq = int(rng.choice([500, 7000, 70000, 140000])), int(rng.choice([1, 33, 130]))
    k = int(rng.choice([1, 10, 100, 1000]))
    metric = str(rng.choice(["sqeuclidean", "euclidean", "cosine", "inner_product"]))
    x = (rng.random((n, d), dtype=np.float32) * 1.9 + 0.1)
    q = (rng.random((nq, d), dtype=np.float32) * 1.9 + 0.1)
    k = min(k, n)
    idx = brute_force.build(torch.from_numpy(x).cuda(), metric=metric)
    gd, gi = brute_force.search(idx, torch.from_numpy(q).cuda(), k)
    torch.cuda.synchronize()
    od, oi = oracle.brute_force_knn(q, x, k, metric=metric)
    assert (gi.cpu().numpy() == oi).all() and (gd.cpu().numpy() == od).all(), f"d={d} n={n} nq={nq} k={k} {metric}"


@pytest.mark.parametrize("case", range(16))
def test_cagra_walk_random_configuration(case):
    """The single-wave walk on a built graph: dims on both sides of the 8-lane team pass and of the 4-piece load groups,
    every dtype, itopk / search_width / k drawn at random - ids and distances identical to the oracle walk."""
    import torch
    from cuvs_amd.neighbors import cagra

    rng = np.random.default_rng(4000 + case)
    dim = int(rng.choice([4, 24, 33, 96, 128, 200, 256, 520, 768]))
    dtype = [np.float32, np.float16, np.int8, np.uint8][int(rng.integers(0, 4))]
    metric = str(rng.choice(["sqeuclidean", "inner_product", "cosine"]))
    n, nq = int(rng.integers(1200, 3000)), int(rng.choice([5, 60]))
    if dtype in (np.int8, np.uint8):
        lo, hi = (-20, 20) if dtype == np.int8 else (0, 40)
        x, q = rng.integers(lo, hi, size=(n, dim)).astype(dtype), rng.integers(lo, hi, size=(nq, dim)).astype(dtype)
    else:
        x, q = rng.standard_normal((n, dim)).astype(dtype), rng.standard_normal((nq, dim)).astype(dtype)
    degree = int(rng.choice([16, 24, 32]))
    index = cagra.build(cagra.IndexParams(metric=metric, intermediate_graph_degree=2 * degree, graph_degree=degree),
                        torch.from_numpy(x).cuda())
    graph = index.graph.cpu().numpy().view(np.uint32)
    itopk, width, k = int(rng.choice([32, 64, 128])), int(rng.choice([1, 2, 4])), int(rng.choice([1, 10, 32]))
    d, i = cagra.search(cagra.SearchParams(itopk_size=itopk, search_width=width, algo="single_cta"), index,
                        torch.from_numpy(q).cuda(), k)
    torch.cuda.synchronize()
    gi = i.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    od, oi = oracle.cagra_search(x, graph, q, k, itopk_size=itopk, search_width=width, metric=metric)
    tag = f"dim={dim} {np.dtype(dtype).name} {metric} n={n} degree={degree} itopk={itopk} width={width} k={k}"
    assert (gi == oi).all(), f"{tag}: id mismatch rate {(gi != oi).mean():.4f}"
    assert (d.cpu().numpy() == od).all(), tag


@pytest.mark.parametrize("case", range(24))
def test_select_k_random_configuration(case):
    """Row lengths and k across the kernel's internal switch points (64 / 2048 / 8192 winners), heavy ties (values drawn
    from a handful of levels, +-inf, signed zeros), min and max selection: values, positions and tie order equal the
    oracle's."""
    import ctypes as C
    import torch
    import cuvs_amd
    from cuvs_amd._lib import check, lib

    rng = np.random.default_rng(5000 + case)
    rows = int(rng.choice([1, 3, 40]))
    ln = int(rng.choice([1, 7, 64, 65, 1000, 4097, 20000, 70000]))
    k = int(min(ln, rng.choice([1, 2, 10, 64, 65, 300, 2048, 2049, 9000])))
    select_min = bool(rng.integers(0, 2))
    mode = int(rng.integers(0, 3))
    if mode == 0:
        v = rng.standard_normal((rows, ln)).astype(np.float32)
    elif mode == 1:  # few distinct levels: ties everywhere
        v = rng.integers(-3, 4, size=(rows, ln)).astype(np.float32)
    else:
        v = rng.standard_normal((rows, ln)).astype(np.float32)
        v[rng.random((rows, ln)) < 0.1] = np.inf
        v[rng.random((rows, ln)) < 0.1] = -np.inf
        v[rng.random((rows, ln)) < 0.1] = -0.0
    res = cuvs_amd.common.Resources()
    tv = torch.from_numpy(v).cuda()
    ov = torch.empty((rows, k), dtype=torch.float32, device="cuda")
    oi = torch.empty((rows, k), dtype=torch.int64, device="cuda")
    fn = lib().cuvsAmdSelectK
    fn.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    check(fn(res.get_c_obj(), tv.data_ptr(), None, rows, ln, k, ov.data_ptr(), oi.data_ptr(), int(select_min)))
    res.sync()
    ev, ei = oracle.select_k(v, k, select_min)
    tag = f"rows={rows} len={ln} k={k} min={select_min} mode={mode}"
    assert (oi.cpu().numpy() == ei).all(), tag
    assert np.array_equal(ov.cpu().numpy(), ev), tag


@pytest.mark.parametrize("case", range(12))
def test_refine_random_configuration(case):
    import torch
    from cuvs_amd.neighbors import refine

    rng = np.random.default_rng(6000 + case)
    n, d = int(rng.integers(200, 3000)), int(rng.choice([1, 5, 32, 100, 257]))
    m, n_cand = int(rng.choice([1, 17, 200])), int(rng.choice([4, 33, 128]))
    k = int(min(n_cand, rng.choice([1, 4, 10, 64])))
    metric = str(rng.choice(["sqeuclidean", "euclidean", "inner_product", "cosine"]))
    x = (rng.random((n, d), dtype=np.float32) * 1.9 + 0.1)
    q = (rng.random((m, d), dtype=np.float32) * 1.9 + 0.1)
    cand = rng.integers(0, n, size=(m, n_cand)).astype(np.int64)  # duplicates allowed
    gd, gi = refine(torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda(), torch.from_numpy(cand).cuda(), k=k, metric=metric)
    torch.cuda.synchronize()
    od, oi = oracle.refine(x, q, cand, k, metric=metric)
    tag = f"n={n} d={d} m={m} n_cand={n_cand} k={k} {metric}"
    assert (gi.cpu().numpy() == oi).all() and (gd.cpu().numpy() == od).all(), tag
