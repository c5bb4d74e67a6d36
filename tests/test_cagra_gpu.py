"""GPU: cuvsCagra* through the C ABI — the reference's golden vectors (c/tests/neighbors/ann_cagra_c.cu:31-50),
its recall thresholds (cpp/tests/neighbors/ann_cagra.cuh:1416-1470: n=1000, q=100, k=16, min_recall 0.995) and
graph invariants."""
import numpy as np
import pytest

import oracle
from tests.golden import reference_fixtures as G

pytestmark = pytest.mark.gpu


def _build(x, **kw):
    import torch
    from cuvs_amd.neighbors import cagra

    return cagra.build(cagra.IndexParams(**kw), torch.from_numpy(x).cuda())


def _search(index, q, k, filter=None, **kw):
    import torch
    from cuvs_amd.neighbors import cagra

    d, i = cagra.search(cagra.SearchParams(**kw), index, torch.from_numpy(q).cuda(), k, filter=filter)
    torch.cuda.synchronize()
    return d.cpu().numpy(), i.cpu().numpy().astype(np.int64) & 0xFFFFFFFF


def test_golden_vectors():
    index = _build(G.CAGRA_C_DATASET)
    d, i = _search(index, G.CAGRA_C_QUERIES, 1)
    assert (i[:, 0] == G.CAGRA_C_NEIGHBORS).all()
    np.testing.assert_allclose(d[:, 0], G.CAGRA_C_DISTANCES, atol=G.CAGRA_C_TOL)


def test_golden_vectors_filtered():
    import torch
    from cuvs_amd._lib import BITSET

    index = _build(G.CAGRA_C_DATASET)
    tw = torch.from_numpy(G.CAGRA_C_FILTER_WORDS.view(np.int32)).cuda()
    d, i = _search(index, G.CAGRA_C_QUERIES, 1, filter=(tw, BITSET))
    assert (i[:, 0] == G.CAGRA_C_NEIGHBORS_FILTERED).all()
    np.testing.assert_allclose(d[:, 0], G.CAGRA_C_DISTANCES_FILTERED, atol=G.CAGRA_C_TOL)


@pytest.mark.parametrize("dim", [1, 16])
@pytest.mark.parametrize("metric", ["sqeuclidean", "inner_product"])
def test_recall_reference_config(dim, metric):
    if dim == 1 and metric == "inner_product":
        pytest.skip("1-d inner product ranks every row identically")
    rng = np.random.default_rng(1234)
    x = (rng.random((1000, dim), dtype=np.float32) * 1.9 + 0.1).astype(np.float32)
    q = (rng.random((100, dim), dtype=np.float32) * 1.9 + 0.1).astype(np.float32)
    index = _build(x, metric=metric, intermediate_graph_degree=64, graph_degree=32)
    d, i = _search(index, q, 16, itopk_size=256)
    td, ti = oracle.exact_knn(q, x, 16, metric=metric)
    hits = sum(len(np.intersect1d(a, b)) for a, b in zip(i, ti))
    # the reference counts a distance match as a hit too (eval_neighbours); ids alone must already be close
    assert hits / ti.size >= 0.98, hits / ti.size
    close = np.isclose(d, td, rtol=1e-3, atol=1e-3).mean()
    assert max(hits / ti.size, close) >= 0.995


@pytest.mark.parametrize("dtype", [np.float32, np.float16, np.int8, np.uint8])
def test_dtypes_and_larger_dim(dtype):
    rng = np.random.default_rng(7)
    if dtype in (np.int8, np.uint8):
        x = rng.integers(1, 20, size=(5000, 96)).astype(dtype)
        q = rng.integers(1, 20, size=(200, 96)).astype(dtype)
    else:
        x = rng.standard_normal((5000, 96)).astype(dtype)
        q = rng.standard_normal((200, 96)).astype(dtype)
    index = _build(x, intermediate_graph_degree=64, graph_degree=32)
    d, i = _search(index, q, 10, itopk_size=64)
    _, ti = oracle.exact_knn(q.astype(np.float32), x.astype(np.float32), 10)
    assert oracle.recall(i, ti) > 0.9


def test_graph_invariants_and_int64_neighbors():
    import torch
    from cuvs_amd.neighbors import cagra

    rng = np.random.default_rng(3)
    x = rng.standard_normal((3000, 32)).astype(np.float32)
    index = _build(x, intermediate_graph_degree=48, graph_degree=24)
    assert len(index) == 3000 and index.dim == 32 and index.graph_degree == 24
    g = index.graph.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    assert g.shape == (3000, 24)
    assert (g < 3000).all()                                   # no invalid edges
    assert (g != np.arange(3000)[:, None]).all()              # no self loops
    assert all(len(np.unique(r)) == 24 for r in g[:200])      # no duplicate edges
    q = rng.standard_normal((50, 32)).astype(np.float32)
    nb = torch.empty((50, 10), dtype=torch.int64, device="cuda")
    d, i = cagra.search(cagra.SearchParams(itopk_size=64, algo="single_cta"), index, torch.from_numpy(q).cuda(), 10, neighbors=nb)
    torch.cuda.synchronize()
    _, ti = oracle.exact_knn(q, x, 10)
    assert oracle.recall(i.cpu().numpy(), ti) > 0.95
    # same index through cuvsCagraIndexFromArgs
    idx2 = cagra.from_graph(index.graph, torch.from_numpy(x).cuda())
    d2, i2 = cagra.search(cagra.SearchParams(itopk_size=64, algo="single_cta"), idx2, torch.from_numpy(q).cuda(), 10)
    torch.cuda.synchronize()
    assert ((i2.cpu().numpy().astype(np.int64) & 0xFFFFFFFF) == i.cpu().numpy()).all()


def test_reverse_edges_in_rank_chunks(monkeypatch):
    """The reverse-edge lists are collected a chunk of ranks at a time (so that n * degree may exceed 2^32): any chunk
    size gives the graph of the one-pass sort."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((4000, 24)).astype(np.float32)
    g0 = _build(x, intermediate_graph_degree=32, graph_degree=16).graph.cpu().numpy()
    for chunk in ("1", "5"):
        monkeypatch.setenv("CUVS_AMD_CAGRA_RANK_CHUNK", chunk)
        g1 = _build(x, intermediate_graph_degree=32, graph_degree=16).graph.cpu().numpy()
        assert (g0 == g1).all()


@pytest.mark.parametrize("algo", ["multi_cta", "auto"])
@pytest.mark.parametrize("metric", ["sqeuclidean", "inner_product"])
def test_multi_wave_search_small_batch(algo, metric):
    """MULTI_CTA (search_multi_cta_jit.cuh): several walkers per query sharing a traversed table. AUTO picks it for
    batches that do not fill the GPU (search_plan.cuh:121-131). Not bit-reproducible (claim races), so checked the
    way the reference checks it: recall against exact search, valid ids, sorted distances, no duplicates."""
    rng = np.random.default_rng(11)
    x = rng.standard_normal((20000, 64)).astype(np.float32)
    q = rng.standard_normal((7, 64)).astype(np.float32)
    index = _build(x, metric=metric, intermediate_graph_degree=64, graph_degree=32)
    d, i = _search(index, q, 10, itopk_size=64, algo=algo)
    td, ti = oracle.exact_knn(q, x, 10, metric=metric)
    assert oracle.recall(i, ti) >= 0.9, oracle.recall(i, ti)
    assert (i < 20000).all()
    assert all(len(np.unique(r)) == 10 for r in i)
    order = d if metric == "sqeuclidean" else -d
    assert (np.diff(order, axis=1) >= 0).all()
    # reported distances are the true distances of the reported ids
    xd = x[i]
    true = ((xd - q[:, None, :]) ** 2).sum(-1) if metric == "sqeuclidean" else (xd * q[:, None, :]).sum(-1)
    np.testing.assert_allclose(d, true, rtol=1e-4, atol=1e-4)


def test_multi_kernel_request_runs_the_single_workgroup_walk():
    """MULTI_KERNEL is the reference's mode for itopk lists too large for one CTA (search_plan.cuh:121-131 picks it
    above 512). The 160 KB LDS keeps such a list on chip, so the request is served by the single-workgroup walk:
    identical output to SINGLE_CTA, recall of the reference's thresholds."""
    rng = np.random.default_rng(17)
    x = rng.standard_normal((20000, 48)).astype(np.float32)
    q = rng.standard_normal((50, 48)).astype(np.float32)
    index = _build(x, intermediate_graph_degree=64, graph_degree=32)
    d, i = _search(index, q, 100, itopk_size=768, algo="multi_kernel")
    d1, i1 = _search(index, q, 100, itopk_size=768, algo="single_cta")
    assert (i == i1).all() and (d == d1).all()
    _, ti = oracle.exact_knn(q, x, 100)
    assert oracle.recall(i, ti) >= 0.995, oracle.recall(i, ti)


def test_multi_wave_search_filter_and_dtypes():
    import torch
    from cuvs_amd._lib import BITSET

    rng = np.random.default_rng(12)
    x = rng.integers(-20, 20, size=(8000, 48)).astype(np.int8)
    q = rng.integers(-20, 20, size=(5, 48)).astype(np.int8)
    index = _build(x, intermediate_graph_degree=64, graph_degree=32)
    keep = np.zeros(8000, bool); keep[::2] = True
    words = torch.from_numpy(np.packbits(keep, bitorder="little").view(np.int32)).cuda()
    d, i = _search(index, q, 8, filter=(words, BITSET), itopk_size=128, algo="multi_cta")
    assert (i % 2 == 0).all() and (i < 8000).all()
    _, ti = oracle.exact_knn(q.astype(np.float32), x[::2].astype(np.float32), 8)
    assert oracle.recall(i, ti * 2) >= 0.8
    # k larger than one walker's list: needs several walkers (num_cta_per_query * 32 >= k)
    d, i = _search(index, q, 40, itopk_size=64, algo="multi_cta")
    assert all(len(np.unique(r)) == 40 for r in i)


@pytest.mark.parametrize("algo", ["single_cta", "multi_cta"])
def test_cosine_metric(algo, tmp_path):
    """CosineExpanded (reference: python/cuvs/cuvs/tests/test_cagra.py cosine cases, sklearn brute cosine truth)."""
    import torch
    from cuvs_amd.neighbors import cagra

    rng = np.random.default_rng(21)
    x = (rng.standard_normal((6000, 32)) * rng.uniform(0.3, 4.0, (6000, 1))).astype(np.float32)
    q = rng.standard_normal((40, 32)).astype(np.float32)
    index = _build(x, metric="cosine", intermediate_graph_degree=64, graph_degree=32)
    d, i = _search(index, q, 10, itopk_size=64, algo=algo)
    xf, qf = x.astype(np.float64), q.astype(np.float64)
    cosd = 1.0 - (qf @ xf.T) / (np.linalg.norm(qf, axis=1)[:, None] * np.linalg.norm(xf, axis=1)[None, :])
    truth = np.argsort(cosd, axis=1, kind="stable")[:, :10]
    assert oracle.recall(i, truth) >= 0.95
    np.testing.assert_allclose(d, np.take_along_axis(cosd, i, axis=1), rtol=1e-3, atol=2e-3)
    # norms are rebuilt when the index comes back from a file
    f = str(tmp_path / "cos.bin")
    cagra.save(f, index)
    d2, i2 = cagra.search(cagra.SearchParams(itopk_size=64, algo="single_cta"), cagra.load(f), torch.from_numpy(q).cuda(), 10)
    d1, i1 = cagra.search(cagra.SearchParams(itopk_size=64, algo="single_cta"), index, torch.from_numpy(q).cuda(), 10)
    torch.cuda.synchronize()
    assert torch.equal(i1, i2) and torch.equal(d1, d2)


@pytest.mark.parametrize("metric", ["sqeuclidean", "cosine"])
def test_extend(metric):
    """cuvsCagraExtend (add_nodes.cuh): rows added in chunks stay findable, old rows stay findable, graph stays valid."""
    import torch
    from cuvs_amd.neighbors import cagra

    rng = np.random.default_rng(31)
    x = rng.standard_normal((26000, 32)).astype(np.float32)
    q = rng.standard_normal((300, 32)).astype(np.float32)
    index = cagra.build(cagra.IndexParams(metric=metric, intermediate_graph_degree=64, graph_degree=32),
                        torch.from_numpy(x[:20000]).cuda())
    cagra.extend(index, torch.from_numpy(x[20000:]).cuda(), max_chunk_size=2048)
    assert len(index) == 26000
    g = index.graph.cpu().numpy().view(np.uint32).astype(np.int64)
    assert g.shape == (26000, 32) and (g < 26000).all()
    assert (g[:20000] >= 20000).any() and (g[20000:] >= 20000).any()   # reverse edges and links among new rows
    d, i = cagra.search(cagra.SearchParams(itopk_size=64), index, torch.from_numpy(q).cuda(), 10)
    torch.cuda.synchronize()
    i = i.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    if metric == "cosine":
        xn = x / np.linalg.norm(x, axis=1, keepdims=True)
        truth = np.argsort(-(q @ xn.T), axis=1, kind="stable")[:, :10]
    else:
        _, truth = oracle.exact_knn(q, x, 10)
    assert oracle.recall(i, truth) >= 0.9, oracle.recall(i, truth)
    # the new rows take their fair share of the answers
    assert abs((truth >= 20000).mean() - (i >= 20000).mean()) < 0.04


def test_merge():
    import torch
    from cuvs_amd.neighbors import cagra

    rng = np.random.default_rng(41)
    x = rng.standard_normal((9000, 24)).astype(np.float32)
    q = rng.standard_normal((100, 24)).astype(np.float32)
    p = cagra.IndexParams(intermediate_graph_degree=48, graph_degree=24)
    parts = [cagra.build(p, torch.from_numpy(x[a:b]).cuda()) for a, b in ((0, 4000), (4000, 7000), (7000, 9000))]
    merged = cagra.merge(p, parts)
    assert len(merged) == 9000 and merged.graph_degree == 24
    d, i = cagra.search(cagra.SearchParams(itopk_size=64), merged, torch.from_numpy(q).cuda(), 10)
    torch.cuda.synchronize()
    _, truth = oracle.exact_knn(q, x, 10)
    assert oracle.recall(i.cpu().numpy().astype(np.int64) & 0xFFFFFFFF, truth) >= 0.95


def test_merge_with_a_bitset_filter():
    """cuvsCagraMerge with a BITSET over the concatenated rows (cagra_merge.cuh:94-131): the merged index holds the kept rows, in
    order - searching it equals searching an index built on exactly those rows; a bitmap is refused (:45-46)"""
    import torch
    from cuvs_amd._lib import BITMAP, BITSET
    from cuvs_amd.neighbors import cagra

    rng = np.random.default_rng(43)
    x = rng.standard_normal((6000, 24)).astype(np.float32)
    q = rng.standard_normal((100, 24)).astype(np.float32)
    p = cagra.IndexParams(intermediate_graph_degree=48, graph_degree=24)
    parts = [cagra.build(p, torch.from_numpy(x[a:b]).cuda()) for a, b in ((0, 2500), (2500, 6000))]
    keep = rng.random(6000) < 0.7
    words = np.zeros((6000 + 31) // 32, np.uint32)
    for r in np.nonzero(keep)[0]:
        words[r >> 5] |= np.uint32(1) << np.uint32(r & 31)
    merged = cagra.merge(p, parts, filter=(torch.from_numpy(words.view(np.int32)).cuda(), BITSET))
    assert len(merged) == int(keep.sum())
    kept_rows = x[keep]
    d, i = cagra.search(cagra.SearchParams(itopk_size=64), merged, torch.from_numpy(q).cuda(), 10)
    torch.cuda.synchronize()
    _, truth = oracle.exact_knn(q, kept_rows, 10)
    assert oracle.recall(i.cpu().numpy().astype(np.int64) & 0xFFFFFFFF, truth) >= 0.95
    with pytest.raises(Exception, match="[Bb]itmap"):
        cagra.merge(p, parts, filter=(torch.from_numpy(words.view(np.int32)).cuda(), BITMAP))
