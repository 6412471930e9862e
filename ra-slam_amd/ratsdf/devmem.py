"""Device memory without PyTorch: hipMalloc / hipMemcpy through the HIP runtime that libratsdf.so itself links
(ctypes on libamdhip64).  For callers that only need images resident in HBM -- bench.py at N = 1, tools/ -- and
should run on the system ROCm runtime a C / C++ caller of the library gets: a process that imports torch first runs
libratsdf.so on the HIP runtime bundled with the PyTorch wheel instead (one HIP runtime per process).  The engine's
frames/s are the same on both (same-box A/B); what differs is the host side: 1.0 vs
1.9 us to enqueue a frame, and no `import torch` in front of the run.

`TorchLike` offers the three torch calls such callers use -- torch.from_numpy(a).to(dev), tensor.data_ptr(),
torch.cuda.synchronize() -- so the same code runs with either."""
import ctypes as C

import numpy as np

_hip = None


def runtime_library_present():
    """Is there a HIP runtime library to bind by name?  Looks, loads nothing (a caller that has to fall back to
    PyTorch must not have mapped another HIP runtime first)."""
    import ctypes.util
    import os
    return bool(ctypes.util.find_library("amdhip64")) or os.path.exists("/opt/rocm/lib/libamdhip64.so")


def _rt():
    global _hip
    if _hip is None:
        from . import library
        library()   # libratsdf.so first: its libamdhip64 is the process's HIP runtime
        err = None
        for name in ("libamdhip64.so.7", "libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
            try:   # (by its SONAME the loader hands back the copy libratsdf.so has already mapped)
                _hip = C.CDLL(name)
                break
            except OSError as e:
                err = e
        if _hip is None:
            raise ImportError(f"ratsdf.devmem: no HIP runtime library found ({err})")
        _hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _hip.hipFree.argtypes = [C.c_void_p]
        _hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        _hip.hipSetDevice.argtypes = [C.c_int]
        _hip.hipGetDeviceCount.argtypes = [C.POINTER(C.c_int)]
    return _hip


def _chk(err, what):
    if err != 0:
        raise RuntimeError(f"{what}: HIP error {err}")


class DeviceArray:
    """`nbytes` of device memory holding a copy of a host array; freed with the object."""

    def __init__(self, host, device=0):
        rt = _rt()
        a = np.ascontiguousarray(host)
        self.shape, self.dtype, self.nbytes = a.shape, a.dtype, a.nbytes
        _chk(rt.hipSetDevice(int(device)), "hipSetDevice")
        p = C.c_void_p()
        _chk(rt.hipMalloc(C.byref(p), max(a.nbytes, 1)), "hipMalloc")
        self._p = p
        _chk(rt.hipMemcpy(p, a.ctypes.data, a.nbytes, 1), "hipMemcpy H2D")   # synchronous

    def data_ptr(self):
        return self._p.value

    def numpy(self):
        out = np.empty(self.shape, dtype=self.dtype)
        _chk(_rt().hipMemcpy(out.ctypes.data, self._p, self.nbytes, 2), "hipMemcpy D2H")
        return out

    def __del__(self):
        try:
            if self._p:
                _rt().hipFree(self._p)
                self._p = None
        except Exception:
            pass


class _Pending:
    def __init__(self, a):
        self.a = a

    def to(self, dev):
        return DeviceArray(self.a, dev if isinstance(dev, int) else getattr(dev, "index", 0) or 0)


class _Cuda:
    @staticmethod
    def is_available():
        n = C.c_int(0)
        return _rt().hipGetDeviceCount(C.byref(n)) == 0 and n.value > 0

    @staticmethod
    def synchronize():
        _chk(_rt().hipDeviceSynchronize(), "hipDeviceSynchronize")

    @staticmethod
    def empty_cache():
        pass


class TorchLike:
    """the sliver of the torch API bench.py's single-GPU legs use"""
    cuda = _Cuda

    @staticmethod
    def from_numpy(a):
        return _Pending(a)
