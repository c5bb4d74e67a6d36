"""IVF-Flat (reference: python/cuvs/cuvs/neighbors/ivf_flat/ivf_flat.pyx)."""
import ctypes as C

import numpy as np
import torch

from .._lib import DLDataType, DLManagedTensor, Tensor, check, cuvsFilter, lib, view_to_torch
from ..common import auto_sync_resources
from ..distance import DISTANCE_TYPES
from ._util import as_device, make_filter, out_buffers


class _CIndexParams(C.Structure):
    _fields_ = [
        ("metric", C.c_int),
        ("metric_arg", C.c_float),
        ("add_data_on_build", C.c_bool),
        ("n_lists", C.c_uint32),
        ("kmeans_n_iters", C.c_uint32),
        ("kmeans_trainset_fraction", C.c_double),
        ("adaptive_centers", C.c_bool),
        ("conservative_memory_allocation", C.c_bool),
    ]


class _CSearchParams(C.Structure):
    _fields_ = [("n_probes", C.c_uint32)]


class _CIndex(C.Structure):
    _fields_ = [("addr", C.c_size_t), ("dtype", DLDataType)]


class IndexParams:
    def __init__(self, *, n_lists=1024, metric="sqeuclidean", metric_arg=2.0, kmeans_n_iters=20,
                 kmeans_trainset_fraction=0.5, add_data_on_build=True, adaptive_centers=False,
                 conservative_memory_allocation=False):
        self._p = C.POINTER(_CIndexParams)()
        check(lib().cuvsIvfFlatIndexParamsCreate(C.byref(self._p)))
        p = self._p.contents
        p.metric = DISTANCE_TYPES[metric]
        p.metric_arg = metric_arg
        p.add_data_on_build = add_data_on_build
        p.n_lists = n_lists
        p.kmeans_n_iters = kmeans_n_iters
        p.kmeans_trainset_fraction = kmeans_trainset_fraction
        p.adaptive_centers = adaptive_centers
        p.conservative_memory_allocation = conservative_memory_allocation
        self.metric = metric

    def __del__(self):
        try:
            lib().cuvsIvfFlatIndexParamsDestroy(self._p)
        except Exception:
            pass


class SearchParams:
    def __init__(self, *, n_probes=20):
        self._p = C.POINTER(_CSearchParams)()
        check(lib().cuvsIvfFlatSearchParamsCreate(C.byref(self._p)))
        self._p.contents.n_probes = n_probes

    @property
    def n_probes(self):
        return self._p.contents.n_probes

    def __del__(self):
        try:
            lib().cuvsIvfFlatSearchParamsDestroy(self._p)
        except Exception:
            pass


class Index:
    def __init__(self):
        self._p = C.POINTER(_CIndex)()
        check(lib().cuvsIvfFlatIndexCreate(C.byref(self._p)))
        self.trained = False
        self._dtype = None

    def __del__(self):
        try:
            lib().cuvsIvfFlatIndexDestroy(self._p)
        except Exception:
            pass

    def _scalar(self, fn):
        v = C.c_int64(0)
        check(getattr(lib(), fn)(self._p, C.byref(v)))
        return v.value

    n_lists = property(lambda self: self._scalar("cuvsIvfFlatIndexGetNLists"))
    dim = property(lambda self: self._scalar("cuvsIvfFlatIndexGetDim"))

    @property
    def centers(self):
        m = DLManagedTensor()
        check(lib().cuvsIvfFlatIndexGetCenters(self._p, C.byref(m)))
        return view_to_torch(m, "cuda")


def _prep(ds):
    if isinstance(ds, torch.Tensor):
        return ds.contiguous()
    return np.ascontiguousarray(ds)


@auto_sync_resources
def build(index_params, dataset, resources=None):
    ds = _prep(dataset)
    idx = Index()
    t = Tensor(ds)
    check(lib().cuvsIvfFlatBuild(resources.get_c_obj(), index_params._p, t.ptr, idx._p))
    idx.trained = True
    idx._dtype = ds.dtype
    return idx


@auto_sync_resources
def extend(index, new_vectors, new_indices, resources=None):
    tv = Tensor(_prep(new_vectors))
    ti = None if new_indices is None else Tensor(new_indices)
    check(lib().cuvsIvfFlatExtend(resources.get_c_obj(), tv.ptr, ti.ptr if ti is not None else None, index._p))
    return index


@auto_sync_resources
def search(search_params, index, queries, k, neighbors=None, distances=None, resources=None, filter=None):
    if not index.trained:
        raise ValueError("Index needs to be built before calling search.")
    q = as_device(queries)
    neighbors, distances = out_buffers(q.shape[0], k, neighbors, distances)
    flt, keep = make_filter(filter)
    tq, tn, td = Tensor(q), Tensor(neighbors), Tensor(distances)
    fn = lib().cuvsIvfFlatSearch
    fn.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, cuvsFilter]
    check(fn(resources.get_c_obj(), search_params._p, index._p, tq.ptr, tn.ptr, td.ptr, flt))
    del keep
    return distances, neighbors


def export_for_oracle(index, dtype, resources=None):
    """Host copy of centers and every list (rows in original dtype + source ids); tests only."""
    from ..common import Resources

    resources = resources or Resources()
    n_lists, dim = index.n_lists, index.dim
    tdt = {np.dtype("float32"): torch.float32, np.dtype("float16"): torch.float16, np.dtype("int8"): torch.int8,
           np.dtype("uint8"): torch.uint8}[np.dtype(dtype)]
    rows, ids, sizes = [], [], []
    for L in range(n_lists):
        sz = C.c_uint32(0)
        check(lib().cuvsAmdIvfFlatListSize(index._p, C.c_uint32(L), C.byref(sz)))
        out = torch.empty((sz.value, dim), dtype=tdt, device="cuda")
        oid = torch.empty((sz.value,), dtype=torch.int64, device="cuda")
        if sz.value:
            check(lib().cuvsAmdIvfFlatUnpackList(resources.get_c_obj(), index._p, C.c_uint32(L),
                                                 C.c_void_p(out.data_ptr()), C.c_void_p(oid.data_ptr())))
        resources.sync()
        rows.append(out.cpu().numpy())
        ids.append(oid.cpu().numpy())
        sizes.append(sz.value)
    return dict(centers=index.centers.cpu().numpy(), list_sizes=np.array(sizes, np.uint32), rows=rows, ids=ids)


@auto_sync_resources
def save(filename, index, resources=None):
    check(lib().cuvsIvfFlatSerialize(resources.get_c_obj(), C.c_char_p(filename.encode()), index._p))


@auto_sync_resources
def load(filename, resources=None):
    idx = Index()
    check(lib().cuvsIvfFlatDeserialize(resources.get_c_obj(), C.c_char_p(filename.encode()), idx._p))
    idx.trained = True
    return idx
