"""Exact re-ranking (reference: python/cuvs/cuvs/neighbors/refine.pyx)."""
import ctypes as C

import torch

from .._lib import Tensor, check, lib
from ..common import auto_sync_resources
from ..distance import DISTANCE_TYPES
from ._util import as_device, out_buffers


@auto_sync_resources
def refine(dataset, queries, candidates, k=None, indices=None, distances=None, metric="sqeuclidean", resources=None):
    """Returns (distances [m,k], indices [m,k]) of the k best candidates by exact distance. Host inputs (numpy arrays /
    CPU tensors for dataset, queries and candidates) take the library's host path, as in the reference
    (refine.pyx: `_refine_host`); anything on the GPU takes the device path."""
    import numpy as np

    def on_host(x):
        return isinstance(x, np.ndarray) or (isinstance(x, torch.Tensor) and not x.is_cuda)

    if on_host(dataset) and on_host(queries) and on_host(candidates):
        ds, q = np.ascontiguousarray(dataset), np.ascontiguousarray(queries)
        cand = np.ascontiguousarray(np.asarray(candidates), dtype=np.int64)
        if k is None:
            k = indices.shape[1] if indices is not None else cand.shape[1]
        indices = np.empty((q.shape[0], k), np.int64) if indices is None else indices
        distances = np.empty((q.shape[0], k), np.float32) if distances is None else distances
        td, tq, tc, ti, tdd = Tensor(ds), Tensor(q), Tensor(cand), Tensor(indices), Tensor(distances)
        check(lib().cuvsRefine(resources.get_c_obj(), td.ptr, tq.ptr, tc.ptr, C.c_int(DISTANCE_TYPES[metric]), ti.ptr,
                               tdd.ptr))
        return distances, indices
    ds, q = as_device(dataset), as_device(queries)
    cand = as_device(candidates, torch.int64)
    if k is None:
        k = indices.shape[1] if indices is not None else cand.shape[1]
    indices, distances = out_buffers(q.shape[0], k, indices, distances)
    td, tq, tc, ti, tdd = Tensor(ds), Tensor(q), Tensor(cand), Tensor(indices), Tensor(distances)
    check(lib().cuvsRefine(resources.get_c_obj(), td.ptr, tq.ptr, tc.ptr, C.c_int(DISTANCE_TYPES[metric]), ti.ptr,
                           tdd.ptr))
    return distances, indices
