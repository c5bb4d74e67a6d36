"""GPU: the coarse search's grouped form (dist_tile_kernel MODE 3 + select_k_grouped_kernel, round 5) against the plain
distance matrix + select_k of rounds 1-4 and against the oracle: ids AND values bit-identical, ties included (reference:
select_clusters, ivf_pq_search.cuh:60-168 - a GEMM, then raft::matrix::select_k over the n_lists distances of a query)."""
import ctypes as C

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _topk(q, x, k, metric, grouped, res):
    import torch
    from cuvs_amd._lib import lib

    qt, xt = torch.from_numpy(q).cuda(), torch.from_numpy(x).cuda()
    ov = torch.empty((q.shape[0], k), dtype=torch.float32, device="cuda")
    oi = torch.empty((q.shape[0], k), dtype=torch.int32, device="cuda")
    fn = lib().cuvsAmdPairwiseTopK
    fn.argtypes = [C.c_size_t, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
    fn.restype = C.c_int
    rc = fn(res.get_c_obj(), qt.data_ptr(), q.shape[0], xt.data_ptr(), x.shape[0], q.shape[1], {"l2": 0, "ip": 6}[metric], k,
            ov.data_ptr(), oi.data_ptr(), int(grouped))
    assert rc in (1, 2), "cuvsAmdPairwiseTopK failed"
    return rc, ov.cpu().numpy(), oi.cpu().numpy().astype(np.int64) & 0xFFFFFFFF


@pytest.mark.parametrize("metric", ["l2", "ip"])
@pytest.mark.parametrize("m,n,dim,k", [(300, 4096, 64, 8), (700, 16384, 128, 128), (130, 5000, 32, 100), (257, 8192 + 17, 48, 256),
                                       (64, 40000, 16, 200), (1000, 4100, 128, 32)])
def test_grouped_equals_plain_and_oracle(metric, m, n, dim, k, res):
    rng = np.random.default_rng(m + n + k)
    q = rng.standard_normal((m, dim)).astype(np.float32)
    x = rng.standard_normal((n, dim)).astype(np.float32)
    rc, gv, gi = _topk(q, x, k, metric, True, res)
    assert rc == 1, "the grouped form must take this shape"
    _, pv, pi = _topk(q, x, k, metric, False, res)
    assert (gi == pi).all() and (gv == pv).all()
    if metric == "l2" and m * n <= 700 * 16384:
        od, oi = oracle.brute_force_knn(q, x, k, metric="sqeuclidean")   # the same canonical arithmetic and (value, position) select
        assert (gi == oi).all() and (gv == od).all()


@pytest.mark.parametrize("metric", ["l2", "ip"])
def test_grouped_with_masses_of_ties(metric, res):
    """Duplicate rows by the thousand: more elements tie at the bound than the candidate buffer holds - those rows are re-laid in
    column order and served by the radix kernel; the earliest columns win, as in the plain form."""
    rng = np.random.default_rng(5)
    base = rng.standard_normal((8, 32)).astype(np.float32)
    x = base[rng.integers(0, 8, size=8192)]            # 8 distinct rows, ~1000 copies each
    x[::97] = rng.standard_normal((len(x[::97]), 32)).astype(np.float32)   # plus a few rows of their own
    q = rng.standard_normal((200, 32)).astype(np.float32)
    q[:50] = 0.0                                       # all-zero queries: every inner product ties
    for k in (16, 128):
        _, gv, gi = _topk(q, x, k, metric, True, res)
        _, pv, pi = _topk(q, x, k, metric, False, res)
        assert (gi == pi).all() and (gv == pv).all()


def test_ivf_pq_search_same_with_and_without_the_grouped_coarse_search(monkeypatch):
    """End to end through cuvsIvfPqSearch: n_lists 4096, the default handle (grouped coarse search) against a handle created
    with CUVS_AMD_COARSE_GROUPED=0."""
    import torch
    import cuvs_amd
    from cuvs_amd.neighbors import ivf_pq

    rng = np.random.default_rng(3)
    x = rng.standard_normal((60000, 32)).astype(np.float32)
    q = rng.standard_normal((500, 32)).astype(np.float32)
    xt, qt = torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda()
    for metric in ("sqeuclidean", "inner_product", "cosine"):
        r0 = cuvs_amd.common.Resources()
        index = ivf_pq.build(ivf_pq.IndexParams(n_lists=4096, pq_dim=16, kmeans_n_iters=4, metric=metric), xt, resources=r0)
        sp = ivf_pq.SearchParams(n_probes=64)
        d0, i0 = ivf_pq.search(sp, index, qt, 10, resources=r0)
        r0.sync()
        monkeypatch.setenv("CUVS_AMD_COARSE_GROUPED", "0")
        r1 = cuvs_amd.common.Resources()
        monkeypatch.delenv("CUVS_AMD_COARSE_GROUPED")
        d1, i1 = ivf_pq.search(sp, index, qt, 10, resources=r1)
        r1.sync()
        assert torch.equal(i0, i1) and torch.equal(d0, d1)


@pytest.mark.parametrize("metric,pq_bits,pq_dim,dim,n_lists,n_probes,nq",
                         [("sqeuclidean", 8, 64, 128, 24, 12, 900), ("cosine", 8, 64, 128, 24, 12, 900), ("sqeuclidean", 5, 48, 96, 24, 12, 900),
                          ("sqeuclidean", 8, 32, 128, 24, 12, 900),
                          # 4000 queries x 9 probes over 12 lists: ~2700 pairs per list - segments beyond one wave's LDS region in
                          # the grouping without radix sort (sort_big_segments_kernel)
                          ("sqeuclidean", 8, 32, 64, 12, 9, 4000)])
def test_two_stream_schedule_equals_one_stream(metric, pq_bits, pq_dim, dim, n_lists, n_probes, nq, monkeypatch):
    """Round 5: the head kernel runs straight from the probes while grouping, work units and B operands are made on the helper
    stream (ivf_pq_search.hip `overlap`). Same ids and distances as the one-stream schedule (CUVS_AMD_PQ_OVERLAP=0) and as the
    oracle - on the FIRST search of an index too (the derived tables are built before the fork), over several batches
    (max_internal_batch_size below the batch) and when searches alternate between two handles."""
    import torch
    import cuvs_amd
    from cuvs_amd.neighbors import ivf_pq

    rng = np.random.default_rng(31)
    x = (rng.random((40000, dim), dtype=np.float32) * 1.9 + 0.1)
    q = (rng.random((nq, dim), dtype=np.float32) * 1.9 + 0.1)
    xt, qt = torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda()
    r0 = cuvs_amd.common.Resources()
    monkeypatch.setenv("CUVS_AMD_PQ_OVERLAP", "0")
    r1 = cuvs_amd.common.Resources()
    monkeypatch.delenv("CUVS_AMD_PQ_OVERLAP")
    index = ivf_pq.build(ivf_pq.IndexParams(n_lists=n_lists, pq_dim=pq_dim, pq_bits=pq_bits, kmeans_n_iters=8, metric=metric), xt, resources=r0)
    r0.sync()
    # partial head: only the first 512 (256) rows of a query's nearest list are scored by the head phase, the list's other rows go
    # through the filter; 0: the head phase scores whole lists. Same results all three ways.
    handles = []
    for rows in (512, 256, 0):
        monkeypatch.setenv("CUVS_AMD_PQ_HEAD_ROWS", str(rows))
        handles.append(cuvs_amd.common.Resources())
    monkeypatch.delenv("CUVS_AMD_PQ_HEAD_ROWS")
    for mib in (4096, 300):
        sp = ivf_pq.SearchParams(n_probes=n_probes, max_internal_batch_size=mib, lut_dtype=np.float16)
        d0, i0 = ivf_pq.search(sp, index, qt, 10, resources=r0)   # (first call: derived tables built inside this search)
        d1, i1 = ivf_pq.search(sp, index, qt, 10, resources=r1)
        d2, i2 = ivf_pq.search(sp, index, qt, 10, resources=r0)
        r0.sync(); r1.sync()
        assert torch.equal(i0, i1) and torch.equal(d0, d1)
        assert torch.equal(i0, i2) and torch.equal(d0, d2)
        for rh in handles:
            d3, i3 = ivf_pq.search(sp, index, qt, 10, resources=rh)
            rh.sync()
            assert torch.equal(i0, i3) and torch.equal(d0, d3)
    od, oi = oracle.ivf_pq_search(ivf_pq.export_for_oracle(index), q, 10, n_probes, metric=metric, lut="f16")
    assert (i0.cpu().numpy() == oi).all() and (d0.cpu().numpy() == od).all()


@pytest.mark.parametrize("metric", ["sqeuclidean", "inner_product", "cosine"])
def test_ivf_flat_search_same_with_and_without_the_grouped_coarse_search(metric, monkeypatch):
    """cuvsIvfFlatSearch at n_lists 4096: the default handle (grouped coarse search) against CUVS_AMD_COARSE_GROUPED=0."""
    import torch
    import cuvs_amd
    from cuvs_amd.neighbors import ivf_flat

    rng = np.random.default_rng(4)
    x = rng.standard_normal((50000, 32)).astype(np.float32)
    q = rng.standard_normal((400, 32)).astype(np.float32)
    xt, qt = torch.from_numpy(x).cuda(), torch.from_numpy(q).cuda()
    r0 = cuvs_amd.common.Resources()
    index = ivf_flat.build(ivf_flat.IndexParams(n_lists=4096, kmeans_n_iters=4, metric=metric), xt, resources=r0)
    sp = ivf_flat.SearchParams(n_probes=48)
    d0, i0 = ivf_flat.search(sp, index, qt, 10, resources=r0)
    r0.sync()
    monkeypatch.setenv("CUVS_AMD_COARSE_GROUPED", "0")
    r1 = cuvs_amd.common.Resources()
    monkeypatch.delenv("CUVS_AMD_COARSE_GROUPED")
    d1, i1 = ivf_flat.search(sp, index, qt, 10, resources=r1)
    r1.sync()
    assert torch.equal(i0, i1) and torch.equal(d0, d1)
