"""CPU: the k-means parameter objects of the C ABI and the Lloyd oracle against the reference's known answer."""
import ctypes as C

import numpy as np

import oracle
from tests.golden import reference_fixtures as fx


def test_params_defaults_match_the_reference():
    # cpp/include/cuvs/cluster/kmeans.hpp:26-163 via c/src/cluster/kmeans.cpp:228-249
    from cuvs_amd.cluster import kmeans

    p = kmeans.KMeansParams()
    assert (p.metric, p.n_clusters, p.init_method, p.max_iter, p.n_init) == ("sqeuclidean", 8, "KMeansPlusPlus", 300, 1)
    assert (p.tol, p.oversampling_factor, p.batch_samples, p.batch_centroids) == (1e-4, 2.0, 1 << 15, 0)
    assert (p.hierarchical, p.hierarchical_n_iters, p.streaming_batch_size, p.init_size) == (False, 20, 0, 0)
    q = kmeans.KMeansParams(n_clusters=5, init_method="Array", hierarchical=True, hierarchical_n_iters=7, tol=1e-6)
    assert (q.n_clusters, q.init_method, q.hierarchical, q.hierarchical_n_iters, q.tol) == (5, "Array", True, 7, 1e-6)
    try:
        kmeans.KMeansParams(hierarchical_n_iters=3)
    except ValueError:
        pass
    else:
        raise AssertionError("hierarchical_n_iters without hierarchical must be rejected (kmeans.pyx:156-160)")


def test_v2_params_have_no_inertia_check_slot():
    from cuvs_amd._lib import lib

    class V2(C.Structure):
        _fields_ = [("metric", C.c_int), ("n_clusters", C.c_int), ("init", C.c_int), ("max_iter", C.c_int),
                    ("tol", C.c_double), ("n_init", C.c_int), ("oversampling_factor", C.c_double),
                    ("batch_samples", C.c_int), ("batch_centroids", C.c_int), ("hierarchical", C.c_bool),
                    ("hierarchical_n_iters", C.c_int), ("streaming_batch_size", C.c_int64), ("init_size", C.c_int64)]

    p = C.POINTER(V2)()
    assert lib().cuvsKMeansParamsCreate_v2(C.byref(p)) == 1
    v = p.contents
    assert (v.n_clusters, v.max_iter, v.hierarchical, v.hierarchical_n_iters, v.batch_samples) == (8, 300, False, 20, 1 << 15)
    assert lib().cuvsKMeansParamsDestroy_v2(p) == 1
    assert lib().cuvsKMeansParamsCreate(None) == 0  # CUVS_ERROR + text, no abort


def test_lloyd_oracle_reproduces_the_reference_known_answer():
    c, labels, inertia, n_iter = oracle.kmeans_lloyd(fx.KMEANS_C_DATASET, fx.KMEANS_C_INIT_CENTROIDS, 100, 1e-6)
    assert np.abs(c - fx.KMEANS_C_CENTROIDS).max() <= fx.KMEANS_C_TOL
    assert (labels == fx.KMEANS_C_LABELS).all()
    assert abs(inertia - fx.KMEANS_C_INERTIA) <= fx.KMEANS_C_TOL
    assert n_iter > 0


def test_lloyd_oracle_agrees_with_sklearn():
    from sklearn.cluster import KMeans

    rng = np.random.default_rng(3)
    centres = rng.uniform(-20, 20, size=(6, 5))
    x = (centres[rng.integers(0, 6, 3000)] + rng.standard_normal((3000, 5))).astype(np.float32)
    init = x[:6].copy()
    c, labels, inertia, _ = oracle.kmeans_lloyd(x, init, 300, 1e-6)
    sk = KMeans(n_clusters=6, init=init, n_init=1, max_iter=300, tol=0, algorithm="lloyd").fit(x.astype(np.float64))
    assert abs(inertia - sk.inertia_) <= 1e-3 * sk.inertia_
    assert np.abs(c - sk.cluster_centers_).max() < 5e-2
