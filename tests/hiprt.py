"""The few HIP runtime calls the GPU tests need for device memory of their own (ctypes on the runtime the
library under test is linked against; no torch in the test processes)."""
import ctypes as C

import numpy as np

_rt = None


def rt():
    global _rt
    if _rt is None:
        from lanecv import _abi
        _abi.lib()                                    # liblanecv.so first: the runtime it links is then the one found
        L = C.CDLL("libamdhip64.so.7")
        L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        L.hipFree.argtypes = [C.c_void_p]
        L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        L.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        _rt = L
    return _rt


def _ok(rc, what):
    if rc != 0:
        raise RuntimeError("%s failed with HIP error %d" % (what, rc))


class DeviceBuffer:
    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        self.nbytes = int(nbytes)
        _ok(rt().hipMalloc(C.byref(self.ptr), max(1, self.nbytes)), "hipMalloc")

    def upload(self, a):
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        _ok(rt().hipMemcpy(self.ptr, a.ctypes.data_as(C.c_void_p), a.nbytes, 1), "hipMemcpy H2D")

    def download(self, shape, dtype=np.uint8):
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        _ok(rt().hipMemcpy(out.ctypes.data_as(C.c_void_p), self.ptr, out.nbytes, 2), "hipMemcpy D2H")
        return out

    def free(self):
        if self.ptr:
            rt().hipFree(self.ptr)
            self.ptr = C.c_void_p()


def synchronize():
    _ok(rt().hipDeviceSynchronize(), "hipDeviceSynchronize")


def free_bytes():
    f, t = C.c_size_t(), C.c_size_t()
    _ok(rt().hipMemGetInfo(C.byref(f), C.byref(t)), "hipMemGetInfo")
    return f.value
