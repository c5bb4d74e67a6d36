"""Shared helpers for the neighbors modules."""
import ctypes as C

import torch

from .._lib import NO_FILTER, Tensor, cuvsFilter


def as_device(x, dtype=None):
    """torch tensor on the current GPU (numpy / host tensors are uploaded)."""
    if not isinstance(x, torch.Tensor):
        x = torch.as_tensor(x)
    if not x.is_cuda:
        x = x.cuda()
    if dtype is not None and x.dtype != dtype:
        x = x.to(dtype)
    return x.contiguous()


def make_filter(prefilter):
    """prefilter: None or (tensor_of_uint32_words, BITSET|BITMAP). Returns (cuvsFilter, keepalive)."""
    if prefilter is None:
        return cuvsFilter(0, NO_FILTER), None
    bits, ftype = prefilter
    if isinstance(bits, torch.Tensor) and bits.dtype == torch.int32:
        t = Tensor(bits)
        t.m.dl_tensor.dtype.code = 1  # reinterpret as uint32
    else:
        t = Tensor(bits)
    return cuvsFilter(t.addr, ftype), t


def out_buffers(m, k, neighbors, distances, idx_dtype=torch.int64):
    if neighbors is None:
        neighbors = torch.empty((m, k), dtype=idx_dtype, device="cuda")
    if distances is None:
        distances = torch.empty((m, k), dtype=torch.float32, device="cuda")
    return neighbors, distances
