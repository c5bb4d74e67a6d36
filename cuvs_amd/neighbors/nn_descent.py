"""NN-descent all-neighbours graph (reference: python/cuvs/cuvs/neighbors/nn_descent/nn_descent.pyx over
c/include/cuvs/neighbors/nn_descent.h)."""
import ctypes as C

import numpy as np
import torch

from .._lib import DLDataType, Tensor, check, lib
from ..common import auto_sync_resources
from ..distance import DISTANCE_TYPES


class _CParams(C.Structure):
    _fields_ = [
        ("metric", C.c_int),
        ("metric_arg", C.c_float),
        ("graph_degree", C.c_size_t),
        ("intermediate_graph_degree", C.c_size_t),
        ("max_iterations", C.c_size_t),
        ("termination_threshold", C.c_float),
        ("return_distances", C.c_bool),
        ("dist_comp_dtype", C.c_int),
    ]


class _CIndex(C.Structure):
    _fields_ = [("addr", C.c_size_t), ("dtype", DLDataType)]


class IndexParams:
    def __init__(self, *, metric="sqeuclidean", graph_degree=64, intermediate_graph_degree=128, max_iterations=20,
                 termination_threshold=0.0001, return_distances=True):
        self._p = C.POINTER(_CParams)()
        check(lib().cuvsNNDescentIndexParamsCreate(C.byref(self._p)))
        p = self._p.contents
        p.metric = DISTANCE_TYPES[metric]
        p.graph_degree = graph_degree
        p.intermediate_graph_degree = intermediate_graph_degree
        p.max_iterations = max_iterations
        p.termination_threshold = termination_threshold
        p.return_distances = return_distances

    def __del__(self):
        try:
            lib().cuvsNNDescentIndexParamsDestroy(self._p)
        except Exception:
            pass


class Index:
    def __init__(self):
        self._p = C.POINTER(_CIndex)()
        check(lib().cuvsNNDescentIndexCreate(C.byref(self._p)))
        self.trained = False
        self._shape = None
        self._res = None

    def __del__(self):
        try:
            lib().cuvsNNDescentIndexDestroy(self._p)
        except Exception:
            pass

    def _copy_out(self, fn, dtype):
        from ..common import Resources

        res = self._res or Resources()
        out = torch.empty(self._shape, dtype=dtype, device="cuda")
        t = Tensor(out)
        if dtype == torch.int32:
            t.m.dl_tensor.dtype.code = 1  # the graph is uint32; torch has no such dtype
        check(getattr(lib(), fn)(res.get_c_obj(), self._p, t.ptr))
        res.sync()
        return out

    @property
    def graph(self):
        """uint32 [n, graph_degree] (returned as an int32 torch tensor with the same bits)."""
        return self._copy_out("cuvsNNDescentIndexGetGraph", torch.int32)

    @property
    def distances(self):
        return self._copy_out("cuvsNNDescentIndexGetDistances", torch.float32)


@auto_sync_resources
def build(index_params, dataset, graph=None, resources=None):
    """dataset: torch (device) or numpy (host) [n, dim] float32/float16/int8/uint8. Returns an Index."""
    ds = dataset.contiguous() if isinstance(dataset, torch.Tensor) else np.ascontiguousarray(dataset)
    idx = Index()
    t = Tensor(ds)
    tg = Tensor(graph) if graph is not None else None
    if tg is not None:
        tg.m.dl_tensor.dtype.code = 1  # uint32
    check(lib().cuvsNNDescentBuild(resources.get_c_obj(), index_params._p, t.ptr, tg.ptr if tg is not None else None,
                                   idx._p))
    idx.trained = True
    idx._shape = (ds.shape[0], int(index_params._p.contents.graph_degree))
    idx._res = resources
    return idx
