"""CPU: the reference bindings' known answers (Java k=3 vectors, Rust self-neighbour properties) against the oracle."""
import numpy as np
import pytest

import oracle
from tests.golden import reference_fixtures as G


def check_maps(expected, ids, dists, tol=G.JAVA_TOL):
    """CuVSTestCase.java:111-140: same id set per query, |d - d_exp| < tol per id (Maps are unordered)."""
    for q, exp in enumerate(expected):
        got = {int(i): float(d) for i, d in zip(ids[q], dists[q])}
        assert set(got) == set(exp), (q, got, exp)
        for key, val in exp.items():
            assert abs(got[key] - val) < tol, (q, key, got[key], val)


def keep_words(keep, n):
    bits = np.zeros(((n + 31) // 32) * 32, bool)
    bits[list(keep)] = True
    return np.packbits(bits, bitorder="little").view(np.uint32)


@pytest.mark.parametrize("fn", ["exact_knn", "brute_force_knn"])
def test_java_brute_force_k3(fn):
    d, i = getattr(oracle, fn)(G.CAGRA_C_QUERIES, G.CAGRA_C_DATASET, 3)
    check_maps(G.BF_JAVA_K3, i, d)
    # the CAGRA IT holds the same answer computed by the graph search's (q - x)^2 arithmetic
    check_maps(G.CAGRA_JAVA_K3, i, d)


def test_java_brute_force_k3_filtered():
    words = keep_words(G.BF_JAVA_K3_FILTER_KEEP, 4)
    d, i = oracle.brute_force_knn(G.CAGRA_C_QUERIES, G.CAGRA_C_DATASET, 3, keep_bits=words)
    check_maps(G.BF_JAVA_K3_FILTERED, i, d)


def test_rust_brute_force_self_neighbor():
    c = G.RUST_SELF_NEIGHBOR_CASES["brute_force"]
    x = np.random.default_rng(0).random((c["n"], c["dim"]), dtype=np.float32)
    _, i = oracle.brute_force_knn(x[:4], x, c["k"])
    assert (i[:, 0] == np.arange(4)).all()
