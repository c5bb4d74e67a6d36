"""GPU: cuvsKMeans* through the C ABI — the reference's known answer, Lloyd vs the CPU restatement, the balanced
(hierarchical) path bit for bit against the hook the IVF builders use, error texts."""
import ctypes as C

import numpy as np
import pytest

import oracle
from tests.golden import reference_fixtures as fx

pytestmark = pytest.mark.gpu

REL_TOL = 2e-4  # fp32 sums in a different (fixed) order than the float64 restatement


def _blobs(n, d, k, seed, spread=20.0):
    rng = np.random.default_rng(seed)
    centres = rng.uniform(-spread, spread, size=(k, d))
    which = rng.integers(0, k, n)
    return (centres[which] + rng.standard_normal((n, d))).astype(np.float32), centres, which


@pytest.mark.parametrize("host_data", [False, True])
def test_reference_known_answer(host_data):
    # c/tests/cluster/kmeans_c.cu: test_fit_predict (:104-175) and test_fit_host (:178-229)
    import torch
    from cuvs_amd.cluster import kmeans

    params = kmeans.KMeansParams(n_clusters=2, max_iter=100, tol=1e-6, init_method="Array",
                                 streaming_batch_size=4 if host_data else 0)
    x = fx.KMEANS_C_DATASET if host_data else torch.from_numpy(fx.KMEANS_C_DATASET).cuda()
    cent = torch.from_numpy(fx.KMEANS_C_INIT_CENTROIDS).cuda()
    out = kmeans.fit(params, x, centroids=cent)
    assert np.abs(out.centroids.cpu().numpy() - fx.KMEANS_C_CENTROIDS).max() <= fx.KMEANS_C_TOL
    assert out.n_iter > 0 and abs(out.inertia - fx.KMEANS_C_INERTIA) <= fx.KMEANS_C_TOL
    xd = torch.from_numpy(fx.KMEANS_C_DATASET).cuda()
    labels, inertia = kmeans.predict(params, xd, out.centroids, normalize_weight=False)
    assert (labels.cpu().numpy() == fx.KMEANS_C_LABELS).all()
    assert abs(inertia - fx.KMEANS_C_INERTIA) <= fx.KMEANS_C_TOL
    assert abs(kmeans.cluster_cost(xd, out.centroids) - fx.KMEANS_C_INERTIA) <= fx.KMEANS_C_TOL


def test_known_answer_through_the_v2_entry_points():
    import torch
    import cuvs_amd
    from cuvs_amd._lib import Tensor, check, lib

    class V2(C.Structure):
        _fields_ = [("metric", C.c_int), ("n_clusters", C.c_int), ("init", C.c_int), ("max_iter", C.c_int),
                    ("tol", C.c_double), ("n_init", C.c_int), ("oversampling_factor", C.c_double),
                    ("batch_samples", C.c_int), ("batch_centroids", C.c_int), ("hierarchical", C.c_bool),
                    ("hierarchical_n_iters", C.c_int), ("streaming_batch_size", C.c_int64), ("init_size", C.c_int64)]

    res = cuvs_amd.common.Resources()
    p = C.POINTER(V2)()
    check(lib().cuvsKMeansParamsCreate_v2(C.byref(p)))
    p.contents.n_clusters, p.contents.max_iter, p.contents.tol, p.contents.init = 2, 100, 1e-6, 2
    x = torch.from_numpy(fx.KMEANS_C_DATASET).cuda()
    cent = torch.from_numpy(fx.KMEANS_C_INIT_CENTROIDS).cuda()
    labels = torch.empty(8, dtype=torch.int32, device="cuda")
    tx, tc, tl = Tensor(x), Tensor(cent), Tensor(labels)
    inertia, n_iter, pin = C.c_double(-1), C.c_int(-1), C.c_double(-1)
    check(lib().cuvsKMeansFit_v2(res.get_c_obj(), p, tx.ptr, None, tc.ptr, C.byref(inertia), C.byref(n_iter)))
    check(lib().cuvsKMeansPredict_v2(res.get_c_obj(), p, tx.ptr, None, tc.ptr, tl.ptr, C.c_bool(False), C.byref(pin)))
    res.sync()
    assert np.abs(cent.cpu().numpy() - fx.KMEANS_C_CENTROIDS).max() <= fx.KMEANS_C_TOL
    assert (labels.cpu().numpy() == fx.KMEANS_C_LABELS).all()
    assert n_iter.value > 0 and abs(inertia.value - 4.0) <= 1e-4 and abs(pin.value - 4.0) <= 1e-4
    check(lib().cuvsKMeansParamsDestroy_v2(p))


@pytest.mark.parametrize("n,d,k,spread", [(5000, 16, 12, 20.0), (20000, 64, 40, 20.0), (3000, 3, 5, 20.0),
                                          (70000, 128, 64, 20.0), (8000, 8, 10, 3.0), (6000, 6, 12, 5.0)])
def test_lloyd_from_given_centroids_matches_the_restatement(n, d, k, spread):
    import torch
    from cuvs_amd.cluster import kmeans

    x, centres, _ = _blobs(n, d, k, seed=n + k, spread=spread)  # spread 3-5: overlapping blobs, 6-8 iterations
    # one perturbed seed per blob: a short, well-conditioned trajectory (no near-tie decides which optimum is reached)
    init = (centres + 0.5 * np.random.default_rng(1).standard_normal(centres.shape)).astype(np.float32)
    # tol 1e-9: both sides run until the assignment stops changing (shift exactly 0), not to a rounding-sensitive ratio
    oc, ol, oin, oit = oracle.kmeans_lloyd(x, init, 100, 1e-9)
    assert oit < 100
    params = kmeans.KMeansParams(n_clusters=k, max_iter=100, tol=1e-9, init_method="Array")
    out = kmeans.fit(params, torch.from_numpy(x).cuda(), centroids=torch.from_numpy(init).cuda())
    gc = out.centroids.cpu().numpy()
    assert abs(out.n_iter - oit) <= 1
    assert np.abs(gc - oc).max() <= REL_TOL * max(1.0, np.abs(oc).max())
    assert abs(out.inertia - oin) <= REL_TOL * oin
    labels, inertia = kmeans.predict(params, torch.from_numpy(x).cuda(), out.centroids)
    assert (labels.cpu().numpy() != ol).mean() < 1e-3  # rows equidistant within fp32 rounding may flip
    assert abs(inertia - oin) <= REL_TOL * oin
    assert abs(kmeans.cluster_cost(torch.from_numpy(x).cuda(), out.centroids) - oin) <= REL_TOL * oin


def test_sample_weights_act_like_repeated_rows():
    import torch
    from cuvs_amd.cluster import kmeans

    x, centres, _ = _blobs(4000, 8, 6, seed=11)
    rng = np.random.default_rng(12)
    reps = rng.integers(1, 4, size=len(x))
    init = (centres + 0.5 * rng.standard_normal(centres.shape)).astype(np.float32)
    params = kmeans.KMeansParams(n_clusters=6, max_iter=100, tol=1e-9, init_method="Array")
    a = kmeans.fit(params, torch.from_numpy(x).cuda(), centroids=torch.from_numpy(init).cuda(),
                   sample_weights=torch.from_numpy(reps.astype(np.float32)).cuda())
    b = kmeans.fit(params, torch.from_numpy(np.repeat(x, reps, axis=0)).cuda(), centroids=torch.from_numpy(init).cuda())
    assert np.abs(a.centroids.cpu().numpy() - b.centroids.cpu().numpy()).max() <= 1e-3
    oc, _, oin, _ = oracle.kmeans_lloyd(x, init, 100, 1e-9, sample_weights=reps)
    assert np.abs(a.centroids.cpu().numpy() - oc).max() <= 1e-3
    assert abs(a.inertia - oin) <= REL_TOL * oin  # weights rescaled to sum to n on both sides


@pytest.mark.parametrize("init", ["KMeansPlusPlus", "Random"])
def test_seeded_fits_find_the_blobs(init):
    import torch
    from cuvs_amd.cluster import kmeans

    x, centres, which = _blobs(12000, 8, 6, seed=21, spread=40.0)
    ideal = sum(((x[which == c] - x[which == c].mean(0)) ** 2).sum() for c in range(6))
    params = kmeans.KMeansParams(n_clusters=6, init_method=init, n_init=8 if init == "Random" else 4, max_iter=100)
    out = kmeans.fit(params, torch.from_numpy(x).cuda())
    assert out.n_iter > 0
    c = out.centroids.cpu().numpy()
    assert len({tuple(r) for r in np.round(c, 3)}) == 6
    if init == "KMeansPlusPlus":
        assert out.inertia <= 1.05 * ideal  # every blob got its centroid
    else:
        assert out.inertia <= float(((x - x.mean(0)) ** 2).sum())  # better than one centre; local optima allowed
    again = kmeans.fit(params, torch.from_numpy(x).cuda())
    assert (again.centroids.cpu().numpy() == c).all()  # counter-based seeding + ordered sums: reproducible


@pytest.mark.parametrize("n,d,k", [(5000, 32, 64), (2000, 5, 30)])
def test_hierarchical_is_the_balanced_kmeans_of_the_ivf_builders(n, d, k):
    import torch
    from cuvs_amd.cluster import kmeans
    from tests.test_kmeans_cagra_parity_gpu import _gpu_kmeans

    rng = np.random.default_rng(n + k)
    modes = rng.standard_normal((max(4, k // 3), d)).astype(np.float32) * 3
    x = (modes[rng.integers(0, len(modes), n)] + rng.standard_normal((n, d)).astype(np.float32)).astype(np.float32)
    hook_c, hook_l = _gpu_kmeans(x, k, 10, True)
    oc, ol = oracle.kmeans_balanced_fit(x, k, 10, True)
    params = kmeans.KMeansParams(n_clusters=k, hierarchical=True, hierarchical_n_iters=10)
    out = kmeans.fit(params, torch.from_numpy(x).cuda())
    gc = out.centroids.cpu().numpy()
    assert (gc == hook_c).all() and (gc == oc).all()
    assert out.n_iter == 10
    want = float(((x.astype(np.float64) - oc[ol].astype(np.float64)) ** 2).sum())
    assert abs(out.inertia - want) <= REL_TOL * want
    labels, inertia = kmeans.predict(params, torch.from_numpy(x).cuda(), out.centroids)
    assert (labels.cpu().numpy().view(np.uint32) == ol).all() and inertia == 0  # kmeans.cpp:163-178


def test_error_texts():
    import torch
    from cuvs_amd._lib import CuvsError
    from cuvs_amd.cluster import kmeans

    x = torch.randn(100, 4, device="cuda")
    with pytest.raises(CuvsError, match="float64"):
        kmeans.fit(kmeans.KMeansParams(n_clusters=3), x.double(), centroids=torch.empty(3, 4, device="cuda"))
    hp = kmeans.KMeansParams(n_clusters=3, hierarchical=True)
    with pytest.raises(CuvsError, match="sample_weight cannot be used with hierarchical"):
        kmeans.fit(hp, x, sample_weights=torch.ones(100, device="cuda"))
    with pytest.raises(CuvsError, match="not supported with host data"):
        kmeans.fit(hp, x.cpu().numpy())
    with pytest.raises(CuvsError, match="n_clusters"):
        kmeans.fit(kmeans.KMeansParams(n_clusters=3), x, centroids=torch.empty(4, 4, device="cuda"))
    with pytest.raises(CuvsError, match="less than n_clusters"):
        kmeans.fit(kmeans.KMeansParams(n_clusters=300, init_method="Random"), x)
    with pytest.raises(CuvsError, match="metric"):
        kmeans.fit(kmeans.KMeansParams(n_clusters=3, metric="inner_product"), x)
    with pytest.raises(CuvsError, match="device memory"):
        kmeans.predict(kmeans.KMeansParams(n_clusters=3), x.cpu().numpy(), torch.empty(3, 4, device="cuda"))
