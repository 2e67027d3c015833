"""Bind the CPU kernel-logic emulator build of libtrx (tests only; see tests/hipemu/hip/hip_runtime.h)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

_emu = None


def emu_lib():
    global _emu
    if _emu is None:
        from tests.hipemu.build_emu import build_emu
        from torcwa_amd._lib import TrxLib
        _emu = TrxLib(build_emu())
    return _emu


def ptr(a):
    return a.ctypes.data


def dt(a):
    return 1 if a.dtype in (np.complex128, np.float64) else 0
