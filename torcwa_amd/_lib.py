"""ctypes binding of libtrx.so (the gfx950 HIP library behind include/trx.h).

The product loads ONLY `torcwa_amd/libtrx.so` and fails loudly if it is missing: there is no CPU fallback.
`TrxLib(path)` is also used by the tests to bind the kernel-logic emulator build (tests/hipemu), which is test
infrastructure and never loaded from here.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_int, c_long, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtrx.so")

C64, C128 = 0, 1
OP_N, OP_T, OP_C = 0, 1, 2

_SIGS = {
    "trx_version": (c_int, []),
    "trx_strerror": (c_char_p, [c_int]),
    "trx_convmat_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int, c_int]),
    "trx_convmat": (c_int, [c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "trx_gemm": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_long, c_void_p, c_int, c_long,
                         c_void_p, c_void_p, c_int, c_long, c_int, c_void_p]),
    "trx_lu_solve": (c_int, [c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "trx_inverse_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "trx_inverse": (c_int, [c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "trx_eig_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "trx_build_pq": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "trx_layer_smatrix_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "trx_layer_smatrix_ws_bytes_lean": (c_size_t, [c_int, c_int, c_int]),
    "trx_layer_smatrix": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "trx_eig_backward_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "trx_eig_backward": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "trx_hmodes_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "trx_hmodes": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "trx_redheffer_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "trx_redheffer": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "trx_redheffer_halfspace_ws_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    "trx_redheffer_halfspace": (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "trx_build_a_ws_bytes": (c_size_t, [c_int, c_int, c_int]),
    "trx_build_a": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "trx_tuning": (c_int, [c_char_p, c_int]),
    "trx_prof_enable": (c_int, [c_int]),
    "trx_prof_reset": (c_int, []),
    "trx_prof_get": (c_int, [c_int, c_void_p]),
    "trx_prof_tag_name": (c_char_p, [c_int]),
    "trx_eig": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    "trx_eig_last_fallback": (c_int, []),
    "trx_eig_ws_bytes_opts": (c_size_t, [c_int, c_int, c_int, ctypes.c_uint]),
    "trx_eig_opts": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p, ctypes.c_uint]),
}


class TrxError(RuntimeError):
    pass


class TrxLib:
    """Typed handle on a libtrx build.  All pointer arguments are raw addresses (ints)."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise TrxError(
                f"{path} not found: the HIP extension is not built. Run `python torcwa_amd/csrc/build.py` "
                "(hipcc --offload-arch=gfx950). torcwa_amd has no CPU fallback.")
        self.path = path
        self.dll = ctypes.CDLL(path)
        for name, (res, args) in _SIGS.items():
            fn = getattr(self.dll, name)      # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
            setattr(self, name[4:], fn)

    def check(self, rc):
        if rc != 0:
            raise TrxError(f"libtrx error {rc}: {self.strerror(rc).decode()}")


_lib = None


def lib() -> TrxLib:
    global _lib
    if _lib is None:
        _lib = TrxLib(LIB_PATH)
    return _lib


def exported_symbols():
    return sorted(_SIGS)
