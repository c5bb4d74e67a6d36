"""GPU: bit-level parity of the balanced k-means and of the CAGRA graph walk with their CPU twins."""
import ctypes as C

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


def _gpu_kmeans(x, k, n_iters, hierarchical):
    import torch
    import cuvs_amd
    from cuvs_amd._lib import check, lib

    res = cuvs_amd.common.Resources()
    tx = torch.from_numpy(x).cuda()
    centers = torch.empty((k, x.shape[1]), dtype=torch.float32, device="cuda")
    labels = torch.empty((x.shape[0],), dtype=torch.int32, device="cuda")
    fn = lib().cuvsAmdKMeansBalancedFit
    fn.argtypes = [C.c_size_t, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    check(fn(res.get_c_obj(), tx.data_ptr(), x.shape[0], x.shape[1], k, n_iters, int(hierarchical), centers.data_ptr(),
             labels.data_ptr()))
    res.sync()
    return centers.cpu().numpy(), labels.cpu().numpy().astype(np.uint32)


@pytest.mark.parametrize("n,d,k,hier", [(3000, 16, 12, False), (5000, 32, 64, True), (2000, 5, 30, True),
                                        (4096, 2, 256, False)])
def test_kmeans_bit_exact_and_balanced(n, d, k, hier):
    rng = np.random.default_rng(n + k)
    modes = rng.standard_normal((max(4, k // 3), d)).astype(np.float32) * 3
    x = (modes[rng.integers(0, len(modes), n)] + rng.standard_normal((n, d)).astype(np.float32)).astype(np.float32)
    gc, gl = _gpu_kmeans(x, k, 10, hier)
    oc, ol = oracle.kmeans_balanced_fit(x, k, 10, hier)
    assert (gc == oc).all(), f"centre mismatch: max abs {np.abs(gc - oc).max()}"
    assert (gl == ol).all()
    sizes = np.bincount(gl, minlength=k)
    assert sizes.min() > 0  # balancing leaves no empty cluster (reference: adjust_centers, :464-580)
    assert sizes.max() < 12 * n / k


@pytest.mark.parametrize("dtype", [np.float32, np.float16, np.int8])
@pytest.mark.parametrize("metric", ["sqeuclidean", "inner_product", "cosine"])
def test_cagra_search_walk_matches_oracle(dtype, metric):
    import torch
    from cuvs_amd.neighbors import cagra

    rng = np.random.default_rng(5)
    if dtype == np.int8:
        x = rng.integers(-20, 20, size=(3000, 40)).astype(dtype)
        q = rng.integers(-20, 20, size=(120, 40)).astype(dtype)
    else:
        x = rng.standard_normal((3000, 40)).astype(dtype)
        q = rng.standard_normal((120, 40)).astype(dtype)
    index = cagra.build(cagra.IndexParams(metric=metric, intermediate_graph_degree=48, graph_degree=24),
                        torch.from_numpy(x).cuda())
    graph = index.graph.cpu().numpy().view(np.uint32)
    for itopk, width in [(64, 1), (96, 2)]:
        d, i = cagra.search(cagra.SearchParams(itopk_size=itopk, search_width=width, algo="single_cta"), index, torch.from_numpy(q).cuda(), 10)
        torch.cuda.synchronize()
        gi = i.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        od, oi = oracle.cagra_search(x, graph, q, 10, itopk_size=itopk, search_width=width, metric=metric)
        assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.4f}"
        assert (d.cpu().numpy() == od).all()


@pytest.mark.parametrize("samplings,width", [(1, 1), (4, 1), (4, 3), (16, 1), (1, 12)])
def test_cagra_num_random_samplings_and_wide_search(samplings, width):
    """num_random_samplings (device_common_jit.cuh:60-83: every seed slot keeps the nearest of that many pseudo-random
    nodes) and search_width beyond 8: ids and distances identical to the oracle walk; more samplings never lower the recall
    of a walk that starts in the wrong place (clustered rows, few iterations)."""
    import torch
    from cuvs_amd.neighbors import cagra

    rng = np.random.default_rng(11)
    centres = rng.standard_normal((12, 32)).astype(np.float32) * 4
    x = (centres[rng.integers(0, 12, 4000)] + 0.3 * rng.standard_normal((4000, 32))).astype(np.float32)
    q = (centres[rng.integers(0, 12, 96)] + 0.3 * rng.standard_normal((96, 32))).astype(np.float32)
    index = cagra.build(cagra.IndexParams(intermediate_graph_degree=32, graph_degree=16), torch.from_numpy(x).cuda())
    graph = index.graph.cpu().numpy().view(np.uint32)
    sp = cagra.SearchParams(itopk_size=64, search_width=width, algo="single_cta", num_random_samplings=samplings)
    d, i = cagra.search(sp, index, torch.from_numpy(q).cuda(), 10)
    torch.cuda.synchronize()
    gi = i.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    od, oi = oracle.cagra_search(x, graph, q, 10, itopk_size=64, search_width=width, num_random_samplings=samplings)
    assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.4f}"
    assert (d.cpu().numpy() == od).all()
    # the multi-wave walk takes the parameter too (claim races: checked by recall)
    _, ti = oracle.exact_knn(q, x, 10)
    dm, im = cagra.search(cagra.SearchParams(itopk_size=64, algo="multi_cta", num_random_samplings=samplings), index,
                          torch.from_numpy(q).cuda(), 10)
    torch.cuda.synchronize()
    assert oracle.recall(im.cpu().numpy().astype(np.int64) & 0xFFFFFFFF, ti) > 0.9


@pytest.mark.parametrize("dtype,dim", [(np.float16, 768), (np.float32, 600), (np.float16, 1024)])
def test_cagra_search_walk_matches_oracle_large_dim(dtype, dim):
    """The BASELINE C4 row shape (768 fp16) and its neighbours: rows longer than one team pass (dim > 512), so the
    distance loop runs more than once per row. Bit-identical to the oracle walk on the same graph."""
    import torch
    from cuvs_amd.neighbors import cagra

    rng = np.random.default_rng(dim)
    lat = rng.standard_normal((2500, 24)).astype(np.float32)
    A = (rng.standard_normal((24, dim)) / 5).astype(np.float32)
    x = (lat @ A + 0.02 * rng.standard_normal((2500, dim))).astype(dtype)
    q = ((rng.standard_normal((80, 24)).astype(np.float32)) @ A).astype(dtype)
    index = cagra.build(cagra.IndexParams(intermediate_graph_degree=64, graph_degree=32), torch.from_numpy(x).cuda())
    graph = index.graph.cpu().numpy().view(np.uint32)
    d, i = cagra.search(cagra.SearchParams(itopk_size=64, algo="single_cta"), index, torch.from_numpy(q).cuda(), 10)
    torch.cuda.synchronize()
    gi = i.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    od, oi = oracle.cagra_search(x, graph, q, 10, itopk_size=64, search_width=1)
    assert (gi == oi).all(), f"id mismatch rate {(gi != oi).mean():.4f}"
    assert (d.cpu().numpy() == od).all()
    _, ti = oracle.exact_knn(q.astype(np.float32), x.astype(np.float32), 10)
    assert oracle.recall(gi, ti) > 0.95
