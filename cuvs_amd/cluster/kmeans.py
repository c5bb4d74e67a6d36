"""k-means over the C ABI (reference: python/cuvs/cuvs/cluster/kmeans/kmeans.pyx).

``KMeansParams(hierarchical=True)`` is the balanced hierarchical k-means that trains the IVF coarse quantizers.
"""
import ctypes as C
from collections import namedtuple

import numpy as np
import torch

from .._lib import Tensor, check, lib
from ..common import auto_sync_resources
from ..distance import DISTANCE_NAMES, DISTANCE_TYPES

INIT_METHOD_TYPES = {"KMeansPlusPlus": 0, "Random": 1, "Array": 2}
INIT_METHOD_NAMES = {v: k for k, v in INIT_METHOD_TYPES.items()}


class _CParams(C.Structure):
    """struct cuvsKMeansParams (include/cuvs/cluster/kmeans.h)."""

    _fields_ = [
        ("metric", C.c_int),
        ("n_clusters", C.c_int),
        ("init", C.c_int),
        ("max_iter", C.c_int),
        ("tol", C.c_double),
        ("n_init", C.c_int),
        ("oversampling_factor", C.c_double),
        ("batch_samples", C.c_int),
        ("batch_centroids", C.c_int),
        ("inertia_check", C.c_bool),
        ("hierarchical", C.c_bool),
        ("hierarchical_n_iters", C.c_int),
        ("streaming_batch_size", C.c_int64),
        ("init_size", C.c_int64),
    ]


class KMeansParams:
    """Same keywords and defaults as the reference's KMeansParams (kmeans.pyx:48-214)."""

    def __init__(self, *, metric=None, n_clusters=None, init_method=None, max_iter=None, tol=None, n_init=None,
                 oversampling_factor=None, batch_samples=None, batch_centroids=None, inertia_check=None,
                 init_size=None, streaming_batch_size=None, hierarchical=None, hierarchical_n_iters=None):
        self._p = C.POINTER(_CParams)()
        check(lib().cuvsKMeansParamsCreate(C.byref(self._p)))
        p = self._p.contents
        if metric is not None:
            p.metric = DISTANCE_TYPES[metric]
        if init_method is not None:
            p.init = INIT_METHOD_TYPES[init_method]
        for name, v in (("n_clusters", n_clusters), ("max_iter", max_iter), ("tol", tol), ("n_init", n_init),
                        ("oversampling_factor", oversampling_factor), ("batch_samples", batch_samples),
                        ("batch_centroids", batch_centroids), ("init_size", init_size),
                        ("streaming_batch_size", streaming_batch_size), ("hierarchical", hierarchical)):
            if v is not None:
                setattr(p, name, v)
        if hierarchical_n_iters is not None:
            if not p.hierarchical:
                raise ValueError("Setting hierarchical_n_iters requires `hierarchical` to be also set to True")
            p.hierarchical_n_iters = hierarchical_n_iters

    def __del__(self):
        if getattr(self, "_p", None):
            lib().cuvsKMeansParamsDestroy(self._p)
            self._p = None

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        p = self._p.contents
        if name == "metric":
            return DISTANCE_NAMES[p.metric]
        if name == "init_method":
            return INIT_METHOD_NAMES[p.init]
        return getattr(p, name)


FitOutput = namedtuple("FitOutput", "centroids inertia n_iter")
PredictOutput = namedtuple("PredictOutput", "labels inertia")


def _matrix(x):
    """Device tensors stay where they are; host arrays are handed over as host DLPack tensors."""
    if isinstance(x, torch.Tensor):
        return x.contiguous()
    a = np.asarray(x)
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("X must have C contiguous layout")
    return a


@auto_sync_resources
def fit(params, X, centroids=None, sample_weights=None, resources=None):
    """Returns FitOutput(centroids [n_clusters, dim] on the device, inertia, n_iter)."""
    x = _matrix(X)
    if centroids is None:
        centroids = torch.empty((params.n_clusters, x.shape[1]), dtype=torch.float32, device="cuda")
    tx, tc = Tensor(x), Tensor(centroids)
    tw = Tensor(_matrix(sample_weights)) if sample_weights is not None else None
    inertia, n_iter = C.c_double(0), C.c_int(0)
    check(lib().cuvsKMeansFit(resources.get_c_obj(), params._p, tx.ptr, tw.ptr if tw else None, tc.ptr,
                              C.byref(inertia), C.byref(n_iter)))
    return FitOutput(centroids, inertia.value, n_iter.value)


@auto_sync_resources
def predict(params, X, centroids, sample_weights=None, labels=None, normalize_weight=True, resources=None):
    """Returns PredictOutput(labels int32 [n] on the device, inertia)."""
    x = _matrix(X)
    if labels is None:
        labels = torch.empty((x.shape[0],), dtype=torch.int32, device="cuda")
    tx, tc, tl = Tensor(x), Tensor(centroids), Tensor(labels)
    tw = Tensor(_matrix(sample_weights)) if sample_weights is not None else None
    inertia = C.c_double(0)
    check(lib().cuvsKMeansPredict(resources.get_c_obj(), params._p, tx.ptr, tw.ptr if tw else None, tc.ptr, tl.ptr,
                                  C.c_bool(normalize_weight), C.byref(inertia)))
    return PredictOutput(labels, inertia.value)


@auto_sync_resources
def cluster_cost(X, centroids, resources=None):
    """Sum of squared distances of the rows of X to their closest centroid."""
    tx, tc = Tensor(_matrix(X)), Tensor(centroids)
    cost = C.c_double(0)
    check(lib().cuvsKMeansClusterCost(resources.get_c_obj(), tx.ptr, tc.ptr, C.byref(cost)))
    return cost.value
