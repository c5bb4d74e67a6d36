"""Metric names -> cuvsDistanceType (reference: python/cuvs/cuvs/distance/distance.pyx:17-41)."""
DISTANCE_TYPES = {
    "l2": 1,
    "sqeuclidean": 0,
    "l2_unexpanded": 4,        # cuvsDistanceType L2Unexpanded / L2SqrtUnexpanded: the same values as the expanded forms,
    "l2_sqrt_unexpanded": 5,   # computed as a sum of squared differences in the reference
    "euclidean": 1,
    "l1": 3,
    "cityblock": 3,
    "inner_product": 6,
    "chebyshev": 7,
    "canberra": 8,
    "cosine": 2,
    "lp": 9,
    "correlation": 10,
    "jaccard": 11,
    "hellinger": 12,
    "braycurtis": 14,
    "jensenshannon": 15,
    "hamming": 16,
    "kl_divergence": 17,
    "minkowski": 9,
    "russellrao": 18,
    "dice": 19,
    "bitwise_hamming": 20,
}
DISTANCE_NAMES = {v: k for k, v in DISTANCE_TYPES.items()}

SUPPORTED_DISTANCES = ["euclidean", "l2", "sqeuclidean", "inner_product", "cosine"]  # what the HIP library builds


def pairwise_distance(X, Y, out=None, metric="euclidean", p=2.0, resources=None):
    """Distances between the rows of X (m, k) and Y (n, k) -> (m, n) float32 on the device
    (reference: python/cuvs/cuvs/distance/distance.pyx:48-131). X, Y and out share one layout: row-major, or
    column-major as in ``x.t().contiguous().t()``; a missing ``out`` takes X's layout."""
    import ctypes as C

    import torch

    from .._lib import Tensor, check, lib
    from ..common import Resources

    if X.shape[1] != Y.shape[1]:
        raise ValueError("Inputs must have same number of columns. a=%s, b=%s" % (X.shape[1], Y.shape[1]))
    if metric not in DISTANCE_TYPES:
        raise ValueError("metric %s is not supported" % metric)
    if X.dtype != Y.dtype:
        raise ValueError("Inputs must have the same dtypes")
    m, n = X.shape[0], Y.shape[0]
    if out is None:
        if X.is_contiguous():
            out = torch.empty((m, n), dtype=torch.float32, device=X.device)
        else:
            out = torch.empty((n, m), dtype=torch.float32, device=X.device).t()
    sync = resources is None
    resources = resources if resources is not None else Resources()
    tx, ty, to = Tensor(X), Tensor(Y), Tensor(out)
    check(lib().cuvsPairwiseDistance(resources.get_c_obj(), tx.ptr, ty.ptr, to.ptr, C.c_int(DISTANCE_TYPES[metric]),
                                     C.c_float(p)))
    if sync:
        resources.sync()
    return out
