"""GPU: the reference bindings' known answers through the C ABI (Java k=3 vectors for brute force - plain and with
a bitset prefilter - and CAGRA; the Rust bindings' "a dataset row is its own nearest neighbour" tests)."""
import numpy as np
import pytest

from tests.golden import reference_fixtures as G
from tests.test_reference_kats import check_maps, keep_words

pytestmark = pytest.mark.gpu


def test_java_brute_force_k3():
    import torch
    from cuvs_amd._lib import BITSET
    from cuvs_amd.neighbors import brute_force

    idx = brute_force.build(torch.from_numpy(G.CAGRA_C_DATASET).cuda(), metric="sqeuclidean")
    q = torch.from_numpy(G.CAGRA_C_QUERIES).cuda()
    d, i = brute_force.search(idx, q, 3)
    torch.cuda.synchronize()
    check_maps(G.BF_JAVA_K3, i.cpu().numpy(), d.cpu().numpy())
    tw = torch.from_numpy(keep_words(G.BF_JAVA_K3_FILTER_KEEP, 4).view(np.int32)).cuda()
    d, i = brute_force.search(idx, q, 3, prefilter=(tw, BITSET))
    torch.cuda.synchronize()
    check_maps(G.BF_JAVA_K3_FILTERED, i.cpu().numpy(), d.cpu().numpy())


@pytest.mark.parametrize("algo", ["auto", "single_cta", "multi_cta"])
def test_java_cagra_k3(algo):
    import torch
    from cuvs_amd.neighbors import cagra

    idx = cagra.build(cagra.IndexParams(), torch.from_numpy(G.CAGRA_C_DATASET).cuda())
    d, i = cagra.search(cagra.SearchParams(algo=algo), idx, torch.from_numpy(G.CAGRA_C_QUERIES).cuda(), 3)
    torch.cuda.synchronize()
    check_maps(G.CAGRA_JAVA_K3, i.cpu().numpy().astype(np.int64) & 0xFFFFFFFF, d.cpu().numpy())


def _uniform(name):
    c = G.RUST_SELF_NEIGHBOR_CASES[name]
    return c, np.random.default_rng(7).random((c["n"], c["dim"]), dtype=np.float32)


def test_rust_self_neighbor_brute_force():
    import torch
    from cuvs_amd.neighbors import brute_force

    c, x = _uniform("brute_force")
    xt = torch.from_numpy(x).cuda()
    _, i = brute_force.search(brute_force.build(xt, metric="sqeuclidean"), xt[:4], c["k"])
    assert (i.cpu().numpy()[:, 0] == np.arange(4)).all()


def test_rust_self_neighbor_ivf_pq_repeated_search():
    import torch
    from cuvs_amd.neighbors import ivf_pq

    c, x = _uniform("ivf_pq")
    xt = torch.from_numpy(x).cuda()
    index = ivf_pq.build(ivf_pq.IndexParams(n_lists=c["n_lists"]), xt)
    for _ in range(3):  # rust/cuvs/src/ivf_pq/index.rs:160-215: one index, searched repeatedly
        _, i = ivf_pq.search(ivf_pq.SearchParams(), index, xt[:4], c["k"])
        assert (i.cpu().numpy()[:, 0] == np.arange(4)).all()


def test_rust_self_neighbor_ivf_flat():
    import torch
    from cuvs_amd.neighbors import ivf_flat

    c, x = _uniform("ivf_flat")
    xt = torch.from_numpy(x).cuda()
    index = ivf_flat.build(ivf_flat.IndexParams(n_lists=c["n_lists"]), xt)
    _, i = ivf_flat.search(ivf_flat.SearchParams(), index, xt[:4], c["k"])
    assert (i.cpu().numpy()[:, 0] == np.arange(4)).all()


def test_rust_self_neighbor_cagra():
    import torch
    from cuvs_amd.neighbors import cagra

    c, x = _uniform("cagra")
    xt = torch.from_numpy(x).cuda()
    index = cagra.build(cagra.IndexParams(), xt)
    _, i = cagra.search(cagra.SearchParams(), index, xt[:4], c["k"])
    assert ((i.cpu().numpy().astype(np.int64) & 0xFFFFFFFF)[:, 0] == np.arange(4)).all()


def test_go_cagra_filtering():
    """go/cagra/cagra_test.go:160-362 (TestCagraFiltering): 1024 x 16 uniform rows, default index / search parameters,
    k = 4. Without a filter the first four rows find themselves; with an allow-list of rows 512..1023 no result is below
    512 for those queries, and rows 512..515 find themselves at distance < 1e-3."""
    import torch
    from cuvs_amd._lib import BITSET
    from cuvs_amd.neighbors import cagra

    rng = np.random.default_rng(160)
    x = rng.random((1024, 16), dtype=np.float32)
    xt = torch.from_numpy(x).cuda()
    index = cagra.build(cagra.IndexParams(), xt)
    nb = torch.empty((4, 4), dtype=torch.int32, device="cuda")
    d, i = cagra.search(cagra.SearchParams(), index, xt[:4], 4, neighbors=nb)
    torch.cuda.synchronize()
    got = i.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    assert (got[:, 0] == np.arange(4)).all() and (np.abs(d.cpu().numpy()[:, 0]) < 1e-3).all()
    keep = np.zeros(1024, bool)
    keep[512:] = True
    words = torch.from_numpy(np.packbits(keep, bitorder="little").view(np.int32).copy()).cuda()
    d, i = cagra.search(cagra.SearchParams(), index, xt[:4], 4, neighbors=nb, filter=(words, BITSET))
    torch.cuda.synchronize()
    got = i.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    assert (got >= 512).all()
    d, i = cagra.search(cagra.SearchParams(), index, xt[512:516], 4, neighbors=nb, filter=(words, BITSET))
    torch.cuda.synchronize()
    got = i.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    assert (got[:, 0] == np.arange(512, 516)).all() and (np.abs(d.cpu().numpy()[:, 0]) < 1e-3).all()
