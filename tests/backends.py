"""Two ways to run the SAME test bodies against the C ABI of include/trx.h:

  emu : tests/hipemu build of the kernel sources, numpy host buffers (CPU, kernel-logic check; marker `emu`)
  gpu : torcwa_amd/libtrx.so (gfx950), torch CUDA buffers on a real MI355X (marker `gpu`)
"""
import numpy as np
import pytest


class EmuBackend:
    name = "emu"

    def __init__(self):
        from tests.emu import emu_lib
        self.lib = emu_lib()
        self.stream = None

    def dev(self, a):
        return np.ascontiguousarray(a).copy()

    def empty(self, shape, dtype):
        return np.zeros(shape, dtype=dtype)

    def ptr(self, h):
        return h.ctypes.data

    def host(self, h):
        return np.array(h)

    def sync(self):
        pass


class GpuBackend:
    name = "gpu"

    def __init__(self):
        import torch
        from torcwa_amd._lib import lib
        assert torch.cuda.is_available(), "gpu tests need a GPU"
        self.torch = torch
        self.lib = lib()
        self.stream = torch.cuda.current_stream().cuda_stream

    def dev(self, a):
        return self.torch.from_numpy(np.ascontiguousarray(a)).cuda()

    def empty(self, shape, dtype):
        return self.torch.zeros(shape, dtype=getattr(self.torch, np.dtype(dtype).name), device="cuda")

    def ptr(self, h):
        return h.data_ptr()

    def host(self, h):
        self.torch.cuda.synchronize()
        return h.cpu().numpy()

    def sync(self):
        self.torch.cuda.synchronize()


_cache = {}


def get_backend(name):
    if name not in _cache:
        _cache[name] = EmuBackend() if name == "emu" else GpuBackend()
    return _cache[name]


BACKENDS = [pytest.param("emu", marks=pytest.mark.emu), pytest.param("gpu", marks=pytest.mark.gpu)]


def dtcode(dtype):
    return 1 if np.dtype(dtype) in (np.dtype(np.complex128), np.dtype(np.float64)) else 0
