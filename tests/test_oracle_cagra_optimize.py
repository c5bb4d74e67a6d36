"""CPU checks of the graph::optimize restatement (oracle/oracle_cagra_optimize.c): the prune against a literal
transcription of kern_fused_prune's counting rule (graph_core.cuh:251-272), the properties the connectivity pass
guarantees (graph_core.cuh:1186-1581: one component, forest edges protected in the merged rows)."""
import numpy as np
import pytest
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components

import oracle


def clustered_knn(n_clusters, per, K, seed, dim=8):
    """kNN graph of well separated clusters: every list stays inside its cluster -> n_clusters components."""
    rng = np.random.default_rng(seed)
    centers = rng.normal(size=(n_clusters, dim)) * 50.0
    x = (centers[:, None, :] + rng.normal(size=(n_clusters, per, dim))).reshape(-1, dim).astype(np.float32)
    rng.shuffle(x)
    d = ((x[:, None, :] - x[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d, np.inf)
    return np.argsort(d, axis=1, kind="stable")[:, :K].astype(np.uint32), x


def components(graph):
    n, deg = graph.shape
    src = np.repeat(np.arange(n), deg)
    dst = graph.reshape(-1).astype(np.int64)
    ok = dst < n
    m = coo_matrix((np.ones(ok.sum()), (src[ok], dst[ok])), shape=(n, n))
    return connected_components(m, directed=True, connection="weak")[0], connected_components(m, directed=True, connection="strong")[0]


def prune_literal(knn, degree):
    n, K = knn.shape
    out = np.full((n, degree), 0xFFFFFFFF, np.uint32)
    for a in range(n):
        det = np.where(knn[a] == a, K, 0).astype(np.int64)
        for kad in range(K - 1):
            d = knn[a, kad]
            if d >= n:
                continue
            for b in knn[d]:
                hit = np.nonzero(knn[a, kad + 1:] == b)[0]
                if len(hit):
                    det[kad + 1 + hit[0]] += 1
        det = np.minimum(det, 0xFFFF)
        det[knn[a] >= n] = 0xFFFF
        for i in range(degree):
            tags = (det << 16) | np.arange(K)
            tags[det >= 0xFFFF] = 1 << 40
            b = tags.argmin()
            if tags[b] >= (1 << 40):
                break
            out[a, i] = knn[a, b]
            det[knn[a] == knn[a, b]] = 0xFFFF
    return out


def test_prune_and_merge_without_connectivity():
    knn, _ = clustered_knn(3, 40, 16, 0)
    g, left = oracle.cagra_optimize(knn, 8, False)
    assert left == 0
    pruned = prune_literal(knn, 8)
    # the first degree/2 entries of a merged row are the protected pruned edges, untouched by the reverse edges
    assert (g[:, :4] == pruned[:, :4]).all()
    # every row keeps distinct, valid ids
    for r in g:
        v = r[r < len(knn)]
        assert len(np.unique(v)) == len(v) == 8


@pytest.mark.parametrize("n_clusters,per,K,degree", [(8, 30, 16, 8), (40, 12, 8, 6), (5, 64, 32, 16)])
def test_connectivity_pass_joins_every_cluster(n_clusters, per, K, degree):
    knn, _ = clustered_knn(n_clusters, per, K, n_clusters)
    weak0, _ = components(knn)
    assert weak0 == n_clusters  # the kNN graph alone is disconnected
    g0, _ = oracle.cagra_optimize(knn, degree, False)
    assert components(g0)[0] == n_clusters
    mst, cnt, left = oracle.cagra_mst(knn, degree)
    assert left == 1
    n = len(knn)
    # the forest edges are stored in both directions: its rows alone are strongly connected
    assert components(mst) == (1, 1)
    assert cnt.max() <= degree
    # n - 1 undirected edges would be a tree; edges granted in the same round may close cycles (as in the reference,
    # whose rounds add every candidate edge that joined two components at round start), but a node adds at most one
    # edge per round and stops asking once its component is the only one left
    und = {(min(i, j), max(i, j)) for i in range(n) for j in mst[i, :cnt[i]]}
    assert n - 1 <= len(und) < 2 * n
    g, left = oracle.cagra_optimize(knn, degree, True)
    assert left == 1
    assert components(g) == (1, 1)
    for i in range(n):  # protected: every forest edge of a node is in its final row
        assert set(mst[i, :cnt[i]]) <= set(g[i])


def test_connected_input_keeps_short_forest_edges():
    rng = np.random.default_rng(5)
    x = rng.normal(size=(300, 4)).astype(np.float32)
    d = ((x[:, None] - x[None]) ** 2).sum(-1)
    np.fill_diagonal(d, np.inf)
    knn = np.argsort(d, axis=1, kind="stable")[:, :16].astype(np.uint32)
    mst, cnt, left = oracle.cagra_mst(knn, 8)
    assert left == 1
    # forest edges come from low ranks of the kNN lists: every outgoing edge (front slots) is one of the node's neighbours
    ranks = [int(np.nonzero(knn[i] == mst[i, 0])[0][0]) for i in range(300) if cnt[i] and mst[i, 0] in knn[i]]
    assert np.mean(ranks) < 4
