"""Resources handle (reference: python/cuvs/cuvs/common/resources.pyx)."""
import ctypes as C
import functools

from .._lib import check, lib


class Resources:
    """Owns a cuvsResources_t. `stream` may be a torch.cuda.Stream or a raw hipStream_t value."""

    def __init__(self, stream=None):
        self._h = C.c_size_t(0)
        check(lib().cuvsResourcesCreate(C.byref(self._h)))
        if stream is None:
            # share torch's current stream so tensors produced by torch ops are ordered before our kernels
            import torch

            if torch.cuda.is_available():
                stream = torch.cuda.current_stream()
        if stream is not None:
            raw = getattr(stream, "cuda_stream", stream)
            check(lib().cuvsStreamSet(self._h, C.c_void_p(raw)))

    def sync(self):
        check(lib().cuvsStreamSync(self._h))

    def get_c_obj(self):
        return self._h

    @property
    def stream(self):
        s = C.c_void_p(0)
        check(lib().cuvsStreamGet(self._h, C.byref(s)))
        return s.value

    def __del__(self):
        try:
            if self._h.value:
                lib().cuvsResourcesDestroy(self._h)
                self._h = C.c_size_t(0)
        except Exception:
            pass


def auto_sync_resources(f):
    """Create a Resources when none is given and sync before returning
    (reference: cuvs/common/resources.pyx auto_sync_resources)."""

    @functools.wraps(f)
    def wrapper(*args, resources=None, **kwargs):
        sync = resources is None
        resources = resources if resources is not None else Resources()
        ret = f(*args, resources=resources, **kwargs)
        if sync:
            resources.sync()
        return ret

    return wrapper
