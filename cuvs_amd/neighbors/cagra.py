"""CAGRA (reference: python/cuvs/cuvs/neighbors/cagra/cagra.pyx)."""
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
        ("intermediate_graph_degree", C.c_size_t),
        ("graph_degree", C.c_size_t),
        ("build_algo", C.c_int),
        ("nn_descent_niter", C.c_size_t),
        ("compression", C.c_void_p),
        ("graph_build_params", C.c_void_p),
    ]


class _CSearchParams(C.Structure):
    _fields_ = [
        ("max_queries", C.c_size_t),
        ("itopk_size", C.c_size_t),
        ("max_iterations", C.c_size_t),
        ("algo", C.c_int),
        ("team_size", C.c_size_t),
        ("search_width", C.c_size_t),
        ("min_iterations", C.c_size_t),
        ("thread_block_size", C.c_size_t),
        ("hashmap_mode", C.c_int),
        ("hashmap_min_bitlen", C.c_size_t),
        ("hashmap_max_fill_rate", C.c_float),
        ("num_random_samplings", C.c_uint32),
        ("rand_xor_mask", C.c_uint64),
        ("persistent", C.c_bool),
        ("persistent_lifetime", C.c_float),
        ("persistent_device_usage", C.c_float),
    ]


class _CIndex(C.Structure):
    _fields_ = [("addr", C.c_size_t), ("dtype", DLDataType)]


_BUILD_ALGOS = {"auto": 0, "ivf_pq": 1, "nn_descent": 2, "iterative_cagra_search": 3}


class IndexParams:
    def __init__(self, *, metric="sqeuclidean", intermediate_graph_degree=128, graph_degree=64, build_algo="ivf_pq",
                 nn_descent_niter=20, guarantee_connectivity=False):
        # guarantee_connectivity: cagra::index_params::guarantee_connectivity (cagra.hpp:193; C++-only in the reference,
        # here a switch on the handle: cuvsAmdCagraSetGuaranteeConnectivity)
        self.guarantee_connectivity = bool(guarantee_connectivity)
        self._p = C.POINTER(_CIndexParams)()
        check(lib().cuvsCagraIndexParamsCreate(C.byref(self._p)))
        p = self._p.contents
        p.metric = DISTANCE_TYPES[metric]
        p.intermediate_graph_degree = intermediate_graph_degree
        p.graph_degree = graph_degree
        p.nn_descent_niter = nn_descent_niter
        self._algo = _BUILD_ALGOS[build_algo]
        self.metric = metric

    def __del__(self):
        try:
            lib().cuvsCagraIndexParamsDestroy(self._p)
        except Exception:
            pass


class SearchParams:
    def __init__(self, *, max_queries=0, itopk_size=64, max_iterations=0, algo="auto", team_size=0, search_width=1,
                 min_iterations=0, thread_block_size=0, hashmap_mode="auto", hashmap_min_bitlen=0,
                 hashmap_max_fill_rate=0.5, num_random_samplings=1, rand_xor_mask=0x128394):
        self._p = C.POINTER(_CSearchParams)()
        check(lib().cuvsCagraSearchParamsCreate(C.byref(self._p)))
        p = self._p.contents
        p.max_queries = max_queries
        p.itopk_size = itopk_size
        p.max_iterations = max_iterations
        p.algo = {"single_cta": 0, "multi_cta": 1, "multi_kernel": 2, "auto": 100}[algo]
        p.team_size = team_size
        p.search_width = search_width
        p.min_iterations = min_iterations
        p.thread_block_size = thread_block_size
        p.hashmap_mode = {"hash": 0, "small": 1, "auto": 100}[hashmap_mode]
        p.hashmap_min_bitlen = hashmap_min_bitlen
        p.hashmap_max_fill_rate = hashmap_max_fill_rate
        p.num_random_samplings = num_random_samplings
        p.rand_xor_mask = rand_xor_mask

    def __del__(self):
        try:
            lib().cuvsCagraSearchParamsDestroy(self._p)
        except Exception:
            pass


class Index:
    def __init__(self):
        self._p = C.POINTER(_CIndex)()
        check(lib().cuvsCagraIndexCreate(C.byref(self._p)))
        self.trained = False
        self._keep = None

    def __del__(self):
        try:
            lib().cuvsCagraIndexDestroy(self._p)
        except Exception:
            pass

    def _scalar(self, fn):
        v = C.c_int64(0)
        check(getattr(lib(), fn)(self._p, C.byref(v)))
        return v.value

    dim = property(lambda self: self._scalar("cuvsCagraIndexGetDims"))
    graph_degree = property(lambda self: self._scalar("cuvsCagraIndexGetGraphDegree"))

    def __len__(self):
        return self._scalar("cuvsCagraIndexGetSize")

    @property
    def graph(self):
        m = DLManagedTensor()
        check(lib().cuvsCagraIndexGetGraph(self._p, C.byref(m)))
        return view_to_torch(m, "cuda")


@auto_sync_resources
def build(index_params, dataset, resources=None):
    ds = dataset.contiguous() if isinstance(dataset, torch.Tensor) else np.ascontiguousarray(dataset)
    idx = Index()
    t = Tensor(ds)
    index_params._p.contents.build_algo = max(index_params._algo, 1) if index_params._algo != 3 else 3
    check(lib().cuvsAmdCagraSetGuaranteeConnectivity(resources.get_c_obj(), C.c_int(int(index_params.guarantee_connectivity))))
    try:
        check(lib().cuvsCagraBuild(resources.get_c_obj(), index_params._p, t.ptr, idx._p))
    finally:
        lib().cuvsAmdCagraSetGuaranteeConnectivity(resources.get_c_obj(), C.c_int(0))
    index_params._p.contents.build_algo = 1  # keep Destroy's graph_build_params bookkeeping valid
    idx._keep = ds  # the index views a device dataset
    idx.trained = True
    return idx


class _CExtendParams(C.Structure):
    _fields_ = [("max_chunk_size", C.c_uint32)]


@auto_sync_resources
def extend(index, additional_dataset, max_chunk_size=0, resources=None):
    """cuvsCagraExtend: add rows to a built index (reference: python/cuvs/cuvs/neighbors/cagra extend). The index
    owns its dataset afterwards."""
    ds = additional_dataset.contiguous() if isinstance(additional_dataset, torch.Tensor) else np.ascontiguousarray(additional_dataset)
    p = C.POINTER(_CExtendParams)()
    check(lib().cuvsCagraExtendParamsCreate(C.byref(p)))
    try:
        p.contents.max_chunk_size = max_chunk_size
        t = Tensor(ds)
        check(lib().cuvsCagraExtend(resources.get_c_obj(), p, t.ptr, index._p))
    finally:
        lib().cuvsCagraExtendParamsDestroy(p)
    index._keep = None
    return index


@auto_sync_resources
def optimize(knn_graph, graph_degree, guarantee_connectivity=False, resources=None):
    """cuvs::neighbors::cagra::helpers::optimize (cagra_optimize.hpp): kNN graph [n, K] uint32 -> search graph
    [n, graph_degree] uint32 on the device (cuvsAmdCagraOptimize)."""
    g = knn_graph if isinstance(knn_graph, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(knn_graph).astype(np.uint32).view(np.int32))
    g = g.contiguous().to(torch.int32)
    out = torch.empty((g.shape[0], graph_degree), dtype=torch.int32, device="cuda")
    tk, to = Tensor(g), Tensor(out)
    tk.m.dl_tensor.dtype.code = 1  # uint32
    to.m.dl_tensor.dtype.code = 1
    check(lib().cuvsAmdCagraOptimize(resources.get_c_obj(), tk.ptr, to.ptr, C.c_int(int(guarantee_connectivity))))
    return out


@auto_sync_resources
def from_graph(graph, dataset, metric="sqeuclidean", resources=None):
    """cuvsCagraIndexFromArgs: index from an existing [n, degree] uint32 graph."""
    g = as_device(graph).to(torch.int32) if not isinstance(graph, torch.Tensor) else graph
    ds = as_device(dataset)
    idx = Index()
    tg = Tensor(g)
    tg.m.dl_tensor.dtype.code = 1  # uint32
    td = Tensor(ds)
    check(lib().cuvsCagraIndexFromArgs(resources.get_c_obj(), C.c_int(DISTANCE_TYPES[metric]), tg.ptr, td.ptr, idx._p))
    idx._keep = ds
    idx.trained = True
    return idx


@auto_sync_resources
def search(search_params, index, queries, k, neighbors=None, distances=None, resources=None, filter=None):
    """Returns (distances [m,k] float32, neighbors [m,k] uint32 stored in an int32 tensor, like the reference)."""
    if not index.trained:
        raise ValueError("Index needs to be built before calling search.")
    q = as_device(queries)
    neighbors, distances = out_buffers(q.shape[0], k, neighbors, distances, idx_dtype=torch.int32)
    flt, keep = make_filter(filter)
    tq, tn, td = Tensor(q), Tensor(neighbors), Tensor(distances)
    if neighbors.dtype == torch.int32:
        tn.m.dl_tensor.dtype.code = 1  # uint32
    fn = lib().cuvsCagraSearch
    fn.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, cuvsFilter]
    check(fn(resources.get_c_obj(), search_params._p, index._p, tq.ptr, tn.ptr, td.ptr, flt))
    del keep
    return distances, neighbors


@auto_sync_resources
def save(filename, index, include_dataset=True, resources=None):
    check(lib().cuvsCagraSerialize(resources.get_c_obj(), C.c_char_p(filename.encode()), index._p,
                                   C.c_bool(include_dataset)))


@auto_sync_resources
def load(filename, resources=None):
    idx = Index()
    check(lib().cuvsCagraDeserialize(resources.get_c_obj(), C.c_char_p(filename.encode()), idx._p))
    idx.trained = True
    return idx


@auto_sync_resources
def merge(index_params, indices, resources=None, filter=None):
    """cuvsCagraMerge: one index over the concatenated datasets of `indices` (ids shifted by the preceding sizes); filter: None or a
    bitset over the concatenated rows (uint32 words on the device, BITSET) - only rows whose bit is set are kept (cagra_merge.cuh:94-131)."""
    out = Index()
    arr = (C.POINTER(_CIndex) * len(indices))(*[ix._p for ix in indices])
    flt, keep = make_filter(filter)
    fn = lib().cuvsCagraMerge
    fn.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, cuvsFilter, C.c_void_p]
    index_params._p.contents.build_algo = max(index_params._algo, 1) if index_params._algo != 3 else 3
    check(fn(resources.get_c_obj(), index_params._p, arr, C.c_size_t(len(indices)), flt, out._p))
    del keep
    index_params._p.contents.build_algo = 1
    out.trained = True
    return out
