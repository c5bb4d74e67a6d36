"""CPU: independent cross-checks of the oracle's IVF-PQ / IVF-Flat / CAGRA / k-means restatements against
straightforward numpy (float64) re-derivations on hand-built indexes. These keep the oracle honest without a GPU."""
import numpy as np
import pytest

import oracle


def _toy_pq_index(rng, n=600, dim=16, n_lists=6, pq_dim=4, pq_bits=8):
    pq_len = dim // pq_dim
    x = rng.standard_normal((n, dim)).astype(np.float32)
    centers = x[rng.choice(n, n_lists, replace=False)].copy()
    labels = np.argmin(((x[:, None, :] - centers[None]) ** 2).sum(-1), axis=1)
    rotation = np.eye(dim, dtype=np.float32)
    centers_rot = centers @ rotation.T
    book = 1 << pq_bits
    pqc = (rng.standard_normal((pq_dim, pq_len, book)) * 0.5).astype(np.float32)
    resid = (x @ rotation.T - centers_rot[labels]).astype(np.float32)
    codes_all = oracle.pq_encode(resid, pqc, pq_bits)  # [n, pq_dim]
    codes, ids, sizes = [], [], []
    for L in range(n_lists):
        rows = np.nonzero(labels == L)[0]
        c = codes_all[rows]
        if pq_bits != 8:  # pack little-endian bitstream
            bits = np.zeros((len(rows), pq_dim * pq_bits), np.uint8)
            for s in range(pq_dim):
                for b in range(pq_bits):
                    bits[:, s * pq_bits + b] = (c[:, s] >> b) & 1
            pad = (-bits.shape[1]) % 8
            bits = np.pad(bits, ((0, 0), (0, pad)))
            c = np.packbits(bits, axis=1, bitorder="little")
        codes.append(np.ascontiguousarray(c, dtype=np.uint8))
        ids.append(rows.astype(np.int64))
        sizes.append(len(rows))
    ex = dict(centers=centers, centers_rot=centers_rot, rotation=rotation, pq_centers=pqc,
              list_sizes=np.array(sizes, np.uint32), codes=codes, ids=ids, pq_bits=pq_bits, pq_dim=pq_dim, pq_len=pq_len)
    return x, ex, labels, codes_all


@pytest.mark.parametrize("pq_bits", [8, 5])
@pytest.mark.parametrize("metric", ["sqeuclidean", "inner_product"])
def test_ivf_pq_oracle_vs_numpy(pq_bits, metric):
    rng = np.random.default_rng(pq_bits)
    x, ex, labels, codes_all = _toy_pq_index(rng, pq_bits=pq_bits)
    q = rng.standard_normal((20, x.shape[1])).astype(np.float32)
    k, n_probes = 7, 3
    od, oi = oracle.ivf_pq_search(ex, q, k, n_probes, metric=metric)
    # numpy float64 re-derivation: reconstruct every vector from its code and rank inside the probed lists
    pqc = ex["pq_centers"].astype(np.float64)
    recon = ex["centers_rot"][labels].astype(np.float64).copy()
    for s in range(ex["pq_dim"]):
        recon[:, s * ex["pq_len"]:(s + 1) * ex["pq_len"]] += pqc[s][:, codes_all[:, s]].T
    for qi in range(len(q)):
        qq = q[qi].astype(np.float64)
        c = ex["centers"].astype(np.float64)
        coarse = ((qq - c) ** 2).sum(1) if metric == "sqeuclidean" else -(c @ qq)
        probes = np.argsort(coarse, kind="stable")[:n_probes]
        cand = np.nonzero(np.isin(labels, probes))[0]
        dist = ((recon[cand] - qq) ** 2).sum(1) if metric == "sqeuclidean" else -(recon[cand] @ qq)
        want = cand[np.argsort(dist, kind="stable")[:k]]
        got = oi[qi]
        assert len(np.intersect1d(got, want)) >= k - 1  # fp32 vs fp64 may flip one near-tie
        ref_d = np.sort(dist)[:k] * (1 if metric == "sqeuclidean" else -1)
        np.testing.assert_allclose(np.sort(od[qi]) if metric == "sqeuclidean" else np.sort(od[qi])[::-1], ref_d,
                                   rtol=1e-4, atol=1e-4)


def test_ivf_flat_oracle_vs_numpy():
    rng = np.random.default_rng(1)
    n, dim, n_lists = 800, 12, 8
    x = rng.standard_normal((n, dim)).astype(np.float32)
    centers = x[:n_lists].copy()
    labels = np.argmin(((x[:, None, :] - centers[None]) ** 2).sum(-1), axis=1)
    rows = [x[labels == L] for L in range(n_lists)]
    ids = [np.nonzero(labels == L)[0].astype(np.int64) for L in range(n_lists)]
    ex = dict(centers=centers, list_sizes=np.array([len(r) for r in rows], np.uint32), rows=rows, ids=ids)
    q = rng.standard_normal((25, dim)).astype(np.float32)
    od, oi = oracle.ivf_flat_search(ex, q, 5, 3)
    for qi in range(len(q)):
        probes = np.argsort(((q[qi] - centers) ** 2).sum(1), kind="stable")[:3]
        cand = np.nonzero(np.isin(labels, probes))[0]
        d = ((x[cand].astype(np.float64) - q[qi]) ** 2).sum(1)
        want = cand[np.argsort(d, kind="stable")[:5]]
        assert (oi[qi] == want).all()
        np.testing.assert_allclose(od[qi], np.sort(d)[:5], rtol=1e-5)
    # probing every list == exact search
    od, oi = oracle.ivf_flat_search(ex, q, 5, n_lists)
    td, ti = oracle.exact_knn(q, x, 5)
    assert (oi == ti).all()


def test_cagra_walk_on_complete_graph_is_exact():
    # with a complete graph every node is one hop away: the walk must return the exact kNN
    rng = np.random.default_rng(2)
    n, dim = 40, 8
    x = rng.standard_normal((n, dim)).astype(np.float32)
    graph = np.array([[j for j in range(n) if j != i] for i in range(n)], dtype=np.uint32)
    q = rng.standard_normal((10, dim)).astype(np.float32)
    od, oi = oracle.cagra_search(x, graph, q, 5, itopk_size=64)
    td, ti = oracle.exact_knn(q, x, 5)
    assert (oi == ti).all()
    np.testing.assert_allclose(od, td, rtol=1e-5, atol=1e-6)
    # bitset filter: only even ids may be returned
    keep = np.zeros(64, bool); keep[0:n:2] = True
    words = np.packbits(keep, bitorder="little").view(np.uint32)
    od, oi = oracle.cagra_search(x, graph, q, 5, itopk_size=64, filter_words=words)
    assert (oi % 2 == 0).all()


@pytest.mark.parametrize("hier", [False, True])
def test_kmeans_oracle_is_balanced_and_sane(hier):
    rng = np.random.default_rng(3)
    modes = rng.standard_normal((10, 6)).astype(np.float32) * 4
    x = (modes[rng.integers(0, 10, 3000)] + rng.standard_normal((3000, 6))).astype(np.float32)
    k = 36 if hier else 10
    c, lab = oracle.kmeans_balanced_fit(x, k, 10, hier)
    sizes = np.bincount(lab, minlength=k)
    assert sizes.sum() == 3000 and sizes.min() > 0
    # labels are the L2 argmin of the returned centres
    d = ((x[:, None, :].astype(np.float64) - c[None].astype(np.float64)) ** 2).sum(-1)
    assert (np.argmin(d, 1) == lab).mean() > 0.999
    # clustering beats k random points as centres
    rand_c = x[rng.choice(3000, k, replace=False)]
    rand_cost = ((x[:, None, :] - rand_c[None]) ** 2).sum(-1).min(1).sum()
    assert d.min(1).sum() < rand_cost
