"""CPU: pin the oracle against the reference's golden vectors and its Python oracles (scipy / sklearn)."""
import numpy as np
import pytest
from scipy.spatial.distance import cdist

import oracle
from tests.golden import reference_fixtures as G


def test_cagra_c_golden_exact_knn():
    # ann_cagra_c.cu:31-50 is an exact top-1 problem: both oracle kNN flavours must reproduce it
    for fn in (oracle.exact_knn, oracle.brute_force_knn):
        d, i = fn(G.CAGRA_C_QUERIES, G.CAGRA_C_DATASET, 1)
        assert (i[:, 0] == G.CAGRA_C_NEIGHBORS).all()
        np.testing.assert_allclose(d[:, 0], G.CAGRA_C_DISTANCES, atol=G.CAGRA_C_TOL)


def test_cagra_c_golden_filtered():
    d, i = oracle.brute_force_knn(G.CAGRA_C_QUERIES, G.CAGRA_C_DATASET, 1, keep_bits=G.CAGRA_C_FILTER_WORDS)
    assert (i[:, 0] == G.CAGRA_C_NEIGHBORS_FILTERED).all()
    np.testing.assert_allclose(d[:, 0], G.CAGRA_C_DISTANCES_FILTERED, atol=G.CAGRA_C_TOL)


def test_brute_force_label_kat():
    # cpp/tests/neighbors/brute_force.cu:169-185
    for fn in (oracle.exact_knn, oracle.brute_force_knn):
        _, i = fn(G.BF_KAT_POINTS, G.BF_KAT_POINTS, G.BF_KAT_K)
        assert (G.BF_KAT_LABELS[i] == G.BF_KAT_LABELS[:, None]).all()


@pytest.mark.parametrize("metric,sp", [("sqeuclidean", "sqeuclidean"), ("euclidean", "euclidean"),
                                       ("cosine", "cosine"), ("inner_product", None)])
@pytest.mark.parametrize("dim", [1, 3, 32, 100])
def test_knn_matches_scipy(metric, sp, dim):
    # python/cuvs/cuvs/tests/test_brute_force.py:88-103 uses scipy cdist with atol=rtol=1e-3
    rng = np.random.default_rng(dim)
    x = rng.random((777, dim), dtype=np.float32) + 0.1
    q = rng.random((31, dim), dtype=np.float32) + 0.1
    k = 10
    ref = cdist(q, x, sp) if sp else -(q.astype(np.float64) @ x.T.astype(np.float64))
    ri = np.argsort(ref, axis=1, kind="stable")[:, :k]
    rd = np.take_along_axis(ref, ri, 1)
    if sp is None:
        rd = -rd
    for fn in (oracle.exact_knn, oracle.brute_force_knn):
        d, i = fn(q, x, k, metric=metric)
        np.testing.assert_allclose(d, rd, atol=1e-3, rtol=1e-3)
        # ids may differ only where distances tie within fp32 noise
        bad = i != ri
        if bad.any():
            dd = np.take_along_axis(ref, i, 1)
            assert np.allclose(np.abs(dd[bad]), np.abs(np.take_along_axis(ref, ri, 1)[bad]), atol=1e-4, rtol=1e-4)


def test_select_k_rule():
    rng = np.random.default_rng(3)
    v = rng.integers(0, 20, size=(40, 300)).astype(np.float32)  # many ties
    ov, oi = oracle.select_k(v, 17)
    ri = np.argsort(v, axis=1, kind="stable")[:, :17]  # (value, position) lexicographic
    assert (oi == ri).all()
    assert (ov == np.take_along_axis(v, ri, 1)).all()
    ov, oi = oracle.select_k(v, 17, select_min=False)
    ri = np.argsort(-v, axis=1, kind="stable")[:, :17]
    assert (oi == ri).all()
    # k > len pads
    ov, oi = oracle.select_k(v[:, :5], 8)
    assert (oi[:, 5:] == -1).all() and (ov[:, 5:] == np.finfo(np.float32).max).all()


def test_canonical_dot_is_fma_chain():
    # the arithmetic contract shared with the MFMA kernel: a k-ordered fp32 fma chain
    rng = np.random.default_rng(0)
    a = rng.standard_normal((1, 64)).astype(np.float32)
    b = rng.standard_normal((1, 64)).astype(np.float32)
    got = oracle.pairwise(a, b, metric="inner_product")[0, 0]
    acc = np.float32(0)
    for k in range(64):
        acc = np.float32(np.float64(a[0, k]) * np.float64(b[0, k]) + np.float64(acc))  # exact product, one rounding
    assert got == acc


def test_refine_cos_zero_denominator_is_distance_one():
    """refine_host.hpp:333-350: denom > 0 ? 1 - dot / denom : 1 (a zero vector is at cosine distance 1 from everything)"""
    x = np.zeros((3, 8), np.float32)
    x[1] = 1.0
    q = np.ones((1, 8), np.float32)
    d, i = oracle.exact_knn(q, x, 3, metric="cosine")
    assert i[0, 0] == 1 and abs(d[0, 0]) < 1e-6
    assert (d[0, 1:] == 1.0).all() and sorted(i[0, 1:]) == [0, 2]


def test_oracle_refine_keeps_out_of_range_candidates():
    """refine_host.hpp:440-442: id >= n_rows -> distance = max, the id stays; sorted (distance, id)"""
    x = np.arange(12, dtype=np.float32).reshape(6, 2)
    q = np.zeros((1, 2), np.float32)
    cand = np.array([[5, -1, 2, np.iinfo(np.int64).max, 6, 0]], np.int64)
    d, i = oracle.refine(x, q, cand, 6, metric="sqeuclidean")
    assert list(i[0]) == [0, 2, 5, -1, 6, np.iinfo(np.int64).max]
    assert (d[0, 3:] == np.finfo(np.float32).max).all()
    d, i = oracle.refine(x, q, cand, 6, metric="inner_product")
    assert (d[0, 3:] == -np.finfo(np.float32).max).all()
