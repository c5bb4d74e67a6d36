"""ctypes binding of the C ABI in include/cuvs/**.h (libcuvs_c.so built by `make`).

This is the host-side stub a reference binding would use: the reference's Python layer builds
``DLManagedTensor`` structs from ``__cuda_array_interface__`` (python/cuvs/cuvs/common/cydlpack.pyx:66-140)
and calls the ``cuvs*`` C functions; here the same structs are built with ctypes from torch tensors /
numpy arrays. torch is used for device memory only.

The library MUST be present: there is no CPU fallback (the CPU oracle under oracle/ is test-only).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcuvs_c.so")


class CuvsError(RuntimeError):
    """Raised when a cuvs* call returns CUVS_ERROR (text from cuvsGetLastErrorText)."""


# ---- DLPack structs (include/dlpack/dlpack.h)
kDLCPU, kDLCUDA, kDLCUDAHost, kDLROCM = 1, 2, 3, 10
kDLInt, kDLUInt, kDLFloat = 0, 1, 2


class DLDevice(C.Structure):
    _fields_ = [("device_type", C.c_int), ("device_id", C.c_int32)]


class DLDataType(C.Structure):
    _fields_ = [("code", C.c_uint8), ("bits", C.c_uint8), ("lanes", C.c_uint16)]


class DLTensor(C.Structure):
    _fields_ = [
        ("data", C.c_void_p),
        ("device", DLDevice),
        ("ndim", C.c_int32),
        ("dtype", DLDataType),
        ("shape", C.POINTER(C.c_int64)),
        ("strides", C.POINTER(C.c_int64)),
        ("byte_offset", C.c_uint64),
    ]


class DLManagedTensor(C.Structure):
    pass


_DELETER = C.CFUNCTYPE(None, C.POINTER(DLManagedTensor))
DLManagedTensor._fields_ = [
    ("dl_tensor", DLTensor),
    ("manager_ctx", C.c_void_p),
    ("deleter", _DELETER),
]


class cuvsFilter(C.Structure):
    _fields_ = [("addr", C.c_size_t), ("type", C.c_int)]


NO_FILTER, BITSET, BITMAP = 0, 1, 2

_lib = None


def lib():
    """Load libcuvs_c.so (fails loudly if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: run `make` (or __graft_entry__.build()) first. "
                "cuvs_amd has no CPU fallback."
            )
        _lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        _lib.cuvsGetLastErrorText.restype = C.c_char_p
    return _lib


def check(status):
    if status != 1:  # CUVS_SUCCESS
        msg = lib().cuvsGetLastErrorText()
        raise CuvsError(msg.decode() if msg else "cuvs call failed")


_NP_DT = {
    np.dtype("float32"): (kDLFloat, 32),
    np.dtype("float16"): (kDLFloat, 16),
    np.dtype("float64"): (kDLFloat, 64),
    np.dtype("int8"): (kDLInt, 8),
    np.dtype("uint8"): (kDLUInt, 8),
    np.dtype("int32"): (kDLInt, 32),
    np.dtype("uint32"): (kDLUInt, 32),
    np.dtype("int64"): (kDLInt, 64),
    np.dtype("uint64"): (kDLUInt, 64),
}


class Tensor:
    """Owns a DLManagedTensor describing `obj` (torch tensor on cuda/cpu, or numpy array)."""

    def __init__(self, obj):
        import torch

        self.obj = obj
        if isinstance(obj, torch.Tensor):
            npdt = np.dtype(str(obj.dtype).replace("torch.", ""))
            ptr = obj.data_ptr()
            shape = tuple(obj.shape)
            strides = tuple(obj.stride())
            if obj.is_cuda:
                dev = DLDevice(kDLCUDA, obj.device.index or 0)
            else:
                dev = DLDevice(kDLCPU, 0)
        else:
            obj = np.asarray(obj)
            self.obj = obj
            npdt = obj.dtype
            ptr = obj.ctypes.data
            shape = obj.shape
            strides = tuple(s // obj.itemsize for s in obj.strides)
            dev = DLDevice(kDLCPU, 0)
        code, bits = _NP_DT[np.dtype(npdt)]
        nd = len(shape)
        self._shape = (C.c_int64 * nd)(*shape)
        self._strides = (C.c_int64 * nd)(*strides)
        self.m = DLManagedTensor()
        t = self.m.dl_tensor
        t.data = ptr
        t.device = dev
        t.ndim = nd
        t.dtype = DLDataType(code, bits, 1)
        t.shape = self._shape
        t.strides = self._strides
        t.byte_offset = 0
        self.m.manager_ctx = None
        self.m.deleter = _DELETER(0)

    @property
    def ptr(self):
        return C.byref(self.m)

    @property
    def addr(self):
        return C.addressof(self.m)


def view_to_torch(m, device):
    """Copy a (non-owning, device) DLManagedTensor view filled by a getter into a fresh torch tensor."""
    import torch

    t = m.dl_tensor
    shape = [t.shape[i] for i in range(t.ndim)]
    rev = {v: k for k, v in _NP_DT.items()}
    npdt = rev[(t.dtype.code, t.dtype.bits)]
    tdt = getattr(torch, str(npdt)) if str(npdt) != "uint32" else torch.int32
    out = torch.empty(shape, dtype=tdt, device=device)
    n = out.numel() * out.element_size()
    if n:
        hip = hip_runtime()
        hip.hipMemcpy(C.c_void_p(out.data_ptr()), C.c_void_p(t.data), C.c_size_t(n), 4)  # hipMemcpyDefault
    if m.deleter:
        m.deleter(C.pointer(m))
    if str(npdt) == "uint32":
        out = out.view(torch.int32)
    return out


_hip = None


def hip_runtime():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so", mode=C.RTLD_GLOBAL)
    return _hip
