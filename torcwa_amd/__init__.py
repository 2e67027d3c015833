"""torcwa_amd: MI355X-native RCWA inner solver, drop-in for the torcwa.rcwa hot path (kch3782/torcwa 0.1.4.2).

    import torcwa_amd as torcwa
    sim = torcwa.rcwa(freq=1/532., order=[15, 15], L=[300., 300.])        # same API as the reference

The heavy numerics are hand-written HIP kernels for gfx950 behind the C ABI of include/trx.h (torcwa_amd/libtrx.so,
built by `python torcwa_amd/csrc/build.py`).  There is no CPU fallback.
"""
from .torch_eig import Eig
from .geometry import geometry, rcwa_geo
from . import materials
from .rcwa import rcwa
from .batched import BatchedRCWA
from .engine import Engine, NumericalError
from ._lib import TrxError

__version__ = "0.1.0"
__all__ = ["Eig", "geometry", "rcwa_geo", "rcwa", "BatchedRCWA", "Engine", "NumericalError", "TrxError", "materials", "__version__"]
