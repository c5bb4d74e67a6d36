"""Single-process multi-GPU indexes over the C ABI (reference: python/cuvs/cuvs/neighbors/mg/{ivf_flat,ivf_pq,cagra}).

One process drives every GPU of a ``MultiGpuResources`` handle; datasets, queries and results are HOST arrays, as in
the reference. The one-process-per-GPU path over torch.distributed/RCCL is ``cuvs_amd.mg``.

    res = mg.MultiGpuResources()                       # all visible GPUs
    index = mg.build("ivf_pq", ivf_pq.IndexParams(n_lists=1024), dataset, mode="sharded", resources=res)
    distances, neighbors = mg.search(ivf_pq.SearchParams(n_probes=32), index, queries, 10)
"""
import ctypes as C

import numpy as np

from .._lib import Tensor, check, lib

_ALGOS = {"ivf_flat": "IvfFlat", "ivf_pq": "IvfPq", "cagra": "Cagra"}
DISTRIBUTION_MODES = {"replicated": 0, "sharded": 1}
REPLICATED_SEARCH_MODES = {"load_balancer": 0, "round_robin": 1}
SHARDED_MERGE_MODES = {"merge_on_root_rank": 0, "tree_merge": 1}


class _CIndexParams(C.Structure):
    """struct cuvsMultiGpu*IndexParams (mg_*.h): the base params are borrowed from the single-GPU params object."""

    _fields_ = [("base_params", C.c_void_p), ("mode", C.c_int)]


class _CSearchParams(C.Structure):
    _fields_ = [("base_params", C.c_void_p), ("search_mode", C.c_int), ("merge_mode", C.c_int),
                ("n_rows_per_batch", C.c_int64)]


class _CIndex(C.Structure):
    _fields_ = [("addr", C.c_size_t), ("code", C.c_uint8), ("bits", C.c_uint8), ("lanes", C.c_uint16)]


class MultiGpuResources:
    """cuvsMultiGpuResourcesCreate[WithDeviceIds] (reference: cuvs.common.MultiGpuResources)."""

    def __init__(self, device_ids=None):
        self._h = C.c_size_t(0)
        if device_ids is None:
            check(lib().cuvsMultiGpuResourcesCreate(C.byref(self._h)))
        else:
            ids = Tensor(np.ascontiguousarray(device_ids, dtype=np.int32))
            check(lib().cuvsMultiGpuResourcesCreateWithDeviceIds(C.byref(self._h), ids.ptr))

    def get_c_obj(self):
        return self._h

    def __del__(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().cuvsMultiGpuResourcesDestroy(self._h)
            self._h = C.c_size_t(0)


class Index:
    def __init__(self, algo, resources):
        self.algo, self.resources = algo, resources
        self._p = C.POINTER(_CIndex)()
        check(getattr(lib(), f"cuvsMultiGpu{_ALGOS[algo]}IndexCreate")(C.byref(self._p)))

    def _fn(self, name):
        return getattr(lib(), f"cuvsMultiGpu{_ALGOS[self.algo]}{name}")

    def __del__(self):
        if getattr(self, "_p", None):
            self._fn("IndexDestroy")(self._p)
            self._p = None


def _host(a):
    a = np.asarray(a)
    if not a.flags["C_CONTIGUOUS"]:
        raise ValueError("multi-GPU entry points take C-contiguous host arrays")
    return a


def build(algo, index_params, dataset, mode="sharded", resources=None):
    """index_params: the single-GPU IndexParams of `algo` (ivf_flat / ivf_pq / cagra)."""
    resources = resources if resources is not None else MultiGpuResources()
    index = Index(algo, resources)
    p = _CIndexParams(C.cast(index_params._p, C.c_void_p), DISTRIBUTION_MODES[mode])
    ds = Tensor(_host(dataset))
    check(index._fn("Build")(resources.get_c_obj(), C.byref(p), ds.ptr, index._p))
    return index


def extend(index, new_vectors, new_indices=None):
    tv = Tensor(_host(new_vectors))
    ti = Tensor(np.ascontiguousarray(new_indices, dtype=np.int64)) if new_indices is not None else None
    check(index._fn("Extend")(index.resources.get_c_obj(), index._p, tv.ptr, ti.ptr if ti is not None else None))
    return index


def search(search_params, index, queries, k, search_mode="load_balancer", merge_mode="tree_merge",
           n_rows_per_batch=1 << 20):
    """Returns (distances [m, k] float32, neighbors [m, k] int64) as numpy arrays."""
    q = _host(queries)
    neighbors = np.empty((q.shape[0], k), dtype=np.int64)
    distances = np.empty((q.shape[0], k), dtype=np.float32)
    p = _CSearchParams(C.cast(search_params._p, C.c_void_p), REPLICATED_SEARCH_MODES[search_mode],
                       SHARDED_MERGE_MODES[merge_mode], n_rows_per_batch)
    tq, tn, td = Tensor(q), Tensor(neighbors), Tensor(distances)
    check(index._fn("Search")(index.resources.get_c_obj(), C.byref(p), index._p, tq.ptr, tn.ptr, td.ptr))
    return distances, neighbors


def save(filename, index):
    check(index._fn("Serialize")(index.resources.get_c_obj(), index._p, filename.encode()))


def load(algo, filename, resources):
    index = Index(algo, resources)
    check(index._fn("Deserialize")(resources.get_c_obj(), filename.encode(), index._p))
    return index


def distribute(algo, filename, resources):
    """Load a single-GPU index file onto every GPU of `resources` (replicated)."""
    index = Index(algo, resources)
    check(index._fn("Distribute")(resources.get_c_obj(), filename.encode(), index._p))
    return index
