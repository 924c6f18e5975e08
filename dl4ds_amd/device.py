"""Device (HBM) buffers owned through the C ABI -- numpy in / numpy out, no torch in the product path."""
import ctypes
import numpy as np
from . import _lib


class DeviceArray:
    """fp32 (or raw-byte) buffer in HBM with a numpy-like shape."""

    def __init__(self, shape, dtype=np.float32):
        self.shape = tuple(int(s) for s in np.atleast_1d(shape)) if not isinstance(shape, tuple) else tuple(shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        p = ctypes.c_void_p()
        _lib.check(_lib.lib().dl4ds_malloc(ctypes.byref(p), max(self.nbytes, 4)))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, a):
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype)
        d.copy_from(a)
        return d

    @classmethod
    def zeros(cls, shape, dtype=np.float32):
        d = cls(shape, dtype)
        _lib.check(_lib.lib().dl4ds_memset(d.ptr, 0, d.nbytes))
        return d

    def copy_from(self, a):
        a = np.ascontiguousarray(a, dtype=self.dtype)
        assert a.nbytes == self.nbytes, (a.shape, self.shape)
        _lib.check(_lib.lib().dl4ds_memcpy_h2d(self.ptr, a.ctypes.data, self.nbytes))

    def numpy(self):
        out = np.empty(self.shape, self.dtype)
        _lib.check(_lib.lib().dl4ds_memcpy_d2h(out.ctypes.data, self.ptr, self.nbytes))
        return out

    def free(self):
        if getattr(self, 'ptr', None):
            try:
                _lib.lib().dl4ds_free(self.ptr)
            except Exception:
                pass
            self.ptr = None

    def __del__(self):
        self.free()


def sync():
    _lib.check(_lib.lib().dl4ds_sync())
