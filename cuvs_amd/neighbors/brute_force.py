"""Brute-force kNN (reference: python/cuvs/cuvs/neighbors/brute_force/brute_force.pyx:34-262)."""
import ctypes as C

import torch

from .._lib import DLDataType, Tensor, check, lib
from ..common import auto_sync_resources
from ..distance import DISTANCE_TYPES
from ._util import as_device, make_filter, out_buffers


class _CIndex(C.Structure):
    _fields_ = [("addr", C.c_size_t), ("dtype", DLDataType)]


class Index:
    def __init__(self):
        self._p = C.POINTER(_CIndex)()
        check(lib().cuvsBruteForceIndexCreate(C.byref(self._p)))
        self.trained = False
        self._keep = None  # the index is a non-owning view of the dataset

    def __del__(self):
        try:
            if self._p:
                lib().cuvsBruteForceIndexDestroy(self._p)
        except Exception:
            pass

    def __repr__(self):
        return "Index(type=BruteForce)"


@auto_sync_resources
def build(dataset, metric="sqeuclidean", metric_arg=2.0, resources=None):
    """dataset: [n, dim] float32/float16, device (torch) or host (numpy; uploaded)."""
    ds = as_device(dataset)
    idx = Index()
    t = Tensor(ds)
    check(
        lib().cuvsBruteForceBuild(
            resources.get_c_obj(), t.ptr, C.c_int(DISTANCE_TYPES[metric]), C.c_float(metric_arg), idx._p
        )
    )
    idx._keep = ds
    idx.trained = True
    return idx


@auto_sync_resources
def search(index, queries, k, neighbors=None, distances=None, resources=None, prefilter=None):
    """Returns (distances [m,k] float32, neighbors [m,k] int64) like the reference."""
    if not index.trained:
        raise ValueError("Index needs to be built before calling search.")
    q = as_device(queries)
    neighbors, distances = out_buffers(q.shape[0], k, neighbors, distances)
    flt, keep = make_filter(prefilter)
    tq, tn, td = Tensor(q), Tensor(neighbors), Tensor(distances)
    fn = lib().cuvsBruteForceSearch
    fn.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, type(flt)]
    check(fn(resources.get_c_obj(), index._p, tq.ptr, tn.ptr, td.ptr, flt))
    del keep
    return distances, neighbors


@auto_sync_resources
def save(filename, index, resources=None):
    check(lib().cuvsBruteForceSerialize(resources.get_c_obj(), C.c_char_p(filename.encode()), index._p))


@auto_sync_resources
def load(filename, resources=None):
    idx = Index()
    check(lib().cuvsBruteForceDeserialize(resources.get_c_obj(), C.c_char_p(filename.encode()), idx._p))
    idx.trained = True
    return idx
