"""Device buffers for GPU tests through the HIP runtime the render core already loaded (ctypes, by soname: the loader
hands back the same library).  torch is avoided on purpose: inside one pytest process torch's bundled runtime would be
a second HIP runtime next to the library's (it then reports "No HIP GPUs are available")."""
import ctypes as C

import numpy as np


class Hip:
    def __init__(self):
        self.L = C.CDLL("libamdhip64.so.7")
        self.L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.L.hipFree.argtypes = [C.c_void_p]
        self.L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.L.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        self.L.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        self.bufs = []

    def malloc(self, nbytes):
        p = C.c_void_p()
        assert self.L.hipMalloc(C.byref(p), max(1, nbytes)) == 0
        assert self.L.hipMemset(p, 0xAB, max(1, nbytes)) == 0
        self.bufs.append(p)
        return p

    def upload(self, arr):
        a = np.ascontiguousarray(arr)
        p = self.malloc(a.nbytes)
        assert self.L.hipMemcpy(p, a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0     # hipMemcpyHostToDevice
        return p

    def download(self, p, dtype, n):
        out = np.zeros(n, dtype=dtype)
        assert self.L.hipMemcpy(out.ctypes.data_as(C.c_void_p), p, out.nbytes, 2) == 0   # hipMemcpyDeviceToHost
        return out

    def free_bytes(self):
        f, t = C.c_size_t(), C.c_size_t()
        assert self.L.hipMemGetInfo(C.byref(f), C.byref(t)) == 0
        return f.value

    def release(self):
        for p in self.bufs:
            self.L.hipFree(p)
        self.bufs = []
