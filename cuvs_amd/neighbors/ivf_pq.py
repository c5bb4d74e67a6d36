"""IVF-PQ (reference: python/cuvs/cuvs/neighbors/ivf_pq/ivf_pq.pyx — IndexParams :40-236, Index :239-475,
build :477, SearchParams :667, search, extend)."""
import ctypes as C

import numpy as np
import torch

from .._lib import DLDataType, DLManagedTensor, Tensor, check, lib, view_to_torch
from ..common import auto_sync_resources
from ..distance import DISTANCE_TYPES
from ._util import as_device, make_filter, out_buffers

_HIP_DT = {np.dtype("float32"): 0, np.dtype("float16"): 2, np.dtype("int8"): 3, np.dtype("uint8"): 8}


class _CIndexParams(C.Structure):
    _fields_ = [
        ("metric", C.c_int),
        ("metric_arg", C.c_float),
        ("add_data_on_build", C.c_bool),
        ("n_lists", C.c_uint32),
        ("kmeans_n_iters", C.c_uint32),
        ("kmeans_trainset_fraction", C.c_double),
        ("pq_bits", C.c_uint32),
        ("pq_dim", C.c_uint32),
        ("codebook_kind", C.c_int),
        ("force_random_rotation", C.c_bool),
        ("conservative_memory_allocation", C.c_bool),
        ("max_train_points_per_pq_code", C.c_uint32),
        ("codes_layout", C.c_int),
    ]


class _CSearchParams(C.Structure):
    _fields_ = [
        ("n_probes", C.c_uint32),
        ("lut_dtype", C.c_int),
        ("internal_distance_dtype", C.c_int),
        ("coarse_search_dtype", C.c_int),
        ("max_internal_batch_size", C.c_uint32),
        ("preferred_shmem_carveout", C.c_double),
    ]


class _CIndex(C.Structure):
    _fields_ = [("addr", C.c_size_t), ("dtype", DLDataType)]


class IndexParams:
    def __init__(self, *, n_lists=1024, metric="sqeuclidean", metric_arg=2.0, kmeans_n_iters=20,
                 kmeans_trainset_fraction=0.5, pq_bits=8, pq_dim=0, codebook_kind="subspace",
                 force_random_rotation=False, add_data_on_build=True, conservative_memory_allocation=False,
                 max_train_points_per_pq_code=256, codes_layout="interleaved"):
        self._p = C.POINTER(_CIndexParams)()
        check(lib().cuvsIvfPqIndexParamsCreate(C.byref(self._p)))
        p = self._p.contents
        p.metric = DISTANCE_TYPES[metric]
        p.metric_arg = metric_arg
        p.add_data_on_build = add_data_on_build
        p.n_lists = n_lists
        p.kmeans_n_iters = kmeans_n_iters
        p.kmeans_trainset_fraction = kmeans_trainset_fraction
        p.pq_bits = pq_bits
        p.pq_dim = pq_dim
        p.codebook_kind = {"subspace": 0, "cluster": 1}[codebook_kind]
        p.force_random_rotation = force_random_rotation
        p.conservative_memory_allocation = conservative_memory_allocation
        p.max_train_points_per_pq_code = max_train_points_per_pq_code
        p.codes_layout = {"flat": 0, "interleaved": 1}[codes_layout]
        self.metric = metric

    def __del__(self):
        try:
            lib().cuvsIvfPqIndexParamsDestroy(self._p)
        except Exception:
            pass

    def __getattr__(self, name):
        if name in dict(_CIndexParams._fields_) and name != "metric":
            return getattr(self._p.contents, name)
        raise AttributeError(name)


class SearchParams:
    def __init__(self, *, n_probes=20, lut_dtype=np.float32, internal_distance_dtype=np.float32,
                 coarse_search_dtype=np.float32, max_internal_batch_size=4096):
        self._p = C.POINTER(_CSearchParams)()
        check(lib().cuvsIvfPqSearchParamsCreate(C.byref(self._p)))
        p = self._p.contents
        p.n_probes = n_probes
        p.lut_dtype = _HIP_DT[np.dtype(lut_dtype)]
        p.internal_distance_dtype = _HIP_DT[np.dtype(internal_distance_dtype)]
        p.coarse_search_dtype = _HIP_DT[np.dtype(coarse_search_dtype)]
        p.max_internal_batch_size = max_internal_batch_size

    def __del__(self):
        try:
            lib().cuvsIvfPqSearchParamsDestroy(self._p)
        except Exception:
            pass

    @property
    def n_probes(self):
        return self._p.contents.n_probes


class Index:
    def __init__(self):
        self._p = C.POINTER(_CIndex)()
        check(lib().cuvsIvfPqIndexCreate(C.byref(self._p)))
        self.trained = False

    def __del__(self):
        try:
            lib().cuvsIvfPqIndexDestroy(self._p)
        except Exception:
            pass

    def _scalar(self, fn):
        v = C.c_int64(0)
        check(getattr(lib(), fn)(self._p, C.byref(v)))
        return v.value

    n_lists = property(lambda self: self._scalar("cuvsIvfPqIndexGetNLists"))
    dim = property(lambda self: self._scalar("cuvsIvfPqIndexGetDim"))
    pq_dim = property(lambda self: self._scalar("cuvsIvfPqIndexGetPqDim"))
    pq_len = property(lambda self: self._scalar("cuvsIvfPqIndexGetPqLen"))
    pq_bits = property(lambda self: self._scalar("cuvsIvfPqIndexGetPqBits"))

    @property
    def codes_layout(self):
        """list_layout of the index: "flat" or "interleaved" (cuvsAmdIvfPqIndexGetCodesLayout)"""
        v = C.c_int(0)
        check(lib().cuvsAmdIvfPqIndexGetCodesLayout(self._p, C.byref(v)))
        return {0: "flat", 1: "interleaved"}[v.value]

    def __len__(self):
        return self._scalar("cuvsIvfPqIndexGetSize")

    def _tensor(self, fn, *args):
        m = DLManagedTensor()
        check(getattr(lib(), fn)(self._p, *args, C.byref(m)))
        t = m.dl_tensor
        if t.strides:  # strided view (centers): copy through a padded tensor
            shape = [t.shape[i] for i in range(t.ndim)]
            ld = t.strides[0]
            t.shape[1] = ld
            t.strides = None
            full = view_to_torch(m, "cuda")
            return full[:, : shape[1]].contiguous()
        return view_to_torch(m, "cuda")

    centers = property(lambda self: self._tensor("cuvsIvfPqIndexGetCenters"))
    centers_padded = property(lambda self: self._tensor("cuvsIvfPqIndexGetCentersPadded"))
    pq_centers = property(lambda self: self._tensor("cuvsIvfPqIndexGetPqCenters"))
    centers_rot = property(lambda self: self._tensor("cuvsIvfPqIndexGetCentersRot"))
    rotation_matrix = property(lambda self: self._tensor("cuvsIvfPqIndexGetRotationMatrix"))
    list_sizes = property(lambda self: self._tensor("cuvsIvfPqIndexGetListSizes"))

    def list_indices(self, label):
        m = DLManagedTensor()
        check(lib().cuvsIvfPqIndexGetListIndices(self._p, C.c_uint32(label), C.byref(m)))
        return view_to_torch(m, "cuda")

    def list_data(self, label, n_rows=0, offset=0, resources=None):
        """Contiguous bit-packed codes [n_rows, ceil(pq_dim*pq_bits/8)] of one list."""
        from ..common import Resources

        resources = resources or Resources()
        if n_rows == 0:
            n_rows = int(self.list_sizes[label].item()) - offset
        bpr = (self.pq_dim * self.pq_bits + 7) // 8
        out = torch.empty((n_rows, bpr), dtype=torch.uint8, device="cuda")
        t = Tensor(out)
        check(lib().cuvsIvfPqIndexUnpackContiguousListData(resources.get_c_obj(), self._p, t.ptr, C.c_uint32(label),
                                                            C.c_uint32(offset)))
        resources.sync()
        return out


@auto_sync_resources
def build(index_params, dataset, resources=None):
    """dataset: [n, dim] float32/float16/int8/uint8; device (torch) or host (numpy)."""
    ds = dataset if isinstance(dataset, (torch.Tensor, np.ndarray)) else np.asarray(dataset)
    if isinstance(ds, torch.Tensor):
        ds = ds.contiguous()
    else:
        ds = np.ascontiguousarray(ds)
    idx = Index()
    t = Tensor(ds)
    check(lib().cuvsIvfPqBuild(resources.get_c_obj(), index_params._p, t.ptr, idx._p))
    idx.trained = True
    return idx


@auto_sync_resources
def extend(index, new_vectors, new_indices, resources=None):
    tv = Tensor(new_vectors if isinstance(new_vectors, torch.Tensor) else np.ascontiguousarray(new_vectors))
    ti = None if new_indices is None else Tensor(new_indices)
    check(lib().cuvsIvfPqExtend(resources.get_c_obj(), tv.ptr, ti.ptr if ti is not None else None, index._p))
    return index


@auto_sync_resources
def search(search_params, index, queries, k, neighbors=None, distances=None, resources=None, filter=None):
    """Returns (distances [m,k] float32, neighbors [m,k] int64). filter: None or (uint32 words on the device, BITSET) - a
    bitset over source ids, 1 keeps the row (the C++ overload with a sample filter, ivf_pq.hpp:1818-1828; the reference's
    C / Python layers do not expose it: include/cuvs_amd/extensions.h)."""
    if not index.trained:
        raise ValueError("Index needs to be built before calling search.")
    q = as_device(queries)
    neighbors, distances = out_buffers(q.shape[0], k, neighbors, distances)
    tq, tn, td = Tensor(q), Tensor(neighbors), Tensor(distances)
    if filter is None:
        check(lib().cuvsIvfPqSearch(resources.get_c_obj(), search_params._p, index._p, tq.ptr, tn.ptr, td.ptr))
    else:
        flt, keep = make_filter(filter)
        check(lib().cuvsAmdIvfPqSearchFiltered(resources.get_c_obj(), search_params._p, index._p, tq.ptr, tn.ptr, td.ptr, flt))
        del keep
    return distances, neighbors


def export_for_oracle(index, per_cluster=False):
    """Host copy of everything the CPU oracle needs to search the SAME index (tests only)."""
    sizes = index.list_sizes.cpu().numpy().astype(np.uint32)
    codes, ids = [], []
    bpr = (index.pq_dim * index.pq_bits + 7) // 8
    for L in range(index.n_lists):
        if sizes[L]:
            codes.append(index.list_data(L).cpu().numpy())
            ids.append(index.list_indices(L).cpu().numpy())
        else:
            codes.append(np.zeros((0, bpr), np.uint8))
            ids.append(np.zeros((0,), np.int64))
    return dict(
        centers=index.centers.cpu().numpy(),
        centers_rot=index.centers_rot.cpu().numpy(),
        rotation=index.rotation_matrix.cpu().numpy(),
        pq_centers=index.pq_centers.cpu().numpy(),
        list_sizes=sizes,
        codes=codes,
        ids=ids,
        pq_bits=index.pq_bits,
        pq_dim=index.pq_dim,
        pq_len=index.pq_len,
        per_cluster=per_cluster,  # codebook_kind="cluster": pq_centers is [n_lists, pq_len, 2^pq_bits]
    )


@auto_sync_resources
def save(filename, index, resources=None):
    check(lib().cuvsIvfPqSerialize(resources.get_c_obj(), C.c_char_p(filename.encode()), index._p))


@auto_sync_resources
def load(filename, resources=None):
    idx = Index()
    check(lib().cuvsIvfPqDeserialize(resources.get_c_obj(), C.c_char_p(filename.encode()), idx._p))
    idx.trained = True
    return idx


@auto_sync_resources
def transform(index, dataset, resources=None):
    """cuvsIvfPqTransform: (labels [n] uint32 stored in an int32 tensor, codes [n, ceil(pq_dim * pq_bits / 8)] uint8) of
    device rows: the list each row would go to and its PQ code as a contiguous bitstream."""
    ds = dataset.contiguous()
    n = ds.shape[0]
    bpr = (index.pq_dim * index.pq_bits + 7) // 8
    labels = torch.empty((n,), dtype=torch.int32, device=ds.device)
    codes = torch.empty((n, bpr), dtype=torch.uint8, device=ds.device)
    td, tl, tc = Tensor(ds), Tensor(labels), Tensor(codes)
    tl.m.dl_tensor.dtype.code = 1  # uint32
    check(lib().cuvsIvfPqTransform(resources.get_c_obj(), index._p, td.ptr, tl.ptr, tc.ptr))
    return labels, codes
