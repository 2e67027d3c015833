"""Thin, batched Python face of the C ABI in include/trx.h.

`Engine` owns nothing but a library handle and a device: every call allocates its outputs/workspaces as torch
tensors on that device (PyTorch is used only for device memory and streams) and passes raw pointers plus the current
HIP stream to libtrx.  There is no CPU fallback: the default engine binds torcwa_amd/libtrx.so (gfx950) and requires
a CUDA/ROCm device.  (The test-suite can inject the kernel-logic emulator build with host tensors; see tests/.)
"""
import ctypes
import os
import threading

import torch

from . import _lib

_CODE = {torch.complex64: _lib.C64, torch.complex128: _lib.C128}
_REAL = {torch.complex64: torch.float32, torch.complex128: torch.float64}


class NumericalError(RuntimeError):
    pass


def _phase(name):
    """Wall-clock bracket of an Engine method for the per-phase table of bench.py: when `engine.profile_phases` is on, a pair of HIP events
    is recorded on the current stream around the call (no synchronisation; Engine.phase_report() reads them after the timed region)."""
    def deco(fn):
        def wrapped(self, *a, **kw):
            if not self.profile_phases or self.device.type != "cuda":
                return fn(self, *a, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream(self.device))
            try:
                return fn(self, *a, **kw)
            finally:
                e1.record(torch.cuda.current_stream(self.device))
                self._phase_events.append((name, e0, e1))
        wrapped.__name__, wrapped.__doc__ = fn.__name__, fn.__doc__
        return wrapped
    return deco


class Engine:
    def __init__(self, lib=None, device=None):
        if lib is None:
            lib = _lib.lib()                       # raises TrxError if libtrx.so has not been built
            if device is None:
                device = torch.device("cuda")
            device = torch.device(device)
            if device.type != "cuda" or not torch.cuda.is_available():
                raise _lib.TrxError("torcwa_amd runs on an MI355X (ROCm) device only; no CPU path exists. "
                                    f"Requested device: {device}, torch.cuda.is_available()={torch.cuda.is_available()}")
        self.lib = lib
        self.device = torch.device(device if device is not None else "cpu")
        self.check_info = True
        self._fail_acc = {}            # per host thread: device-side count of non-zero info entries seen while check_info is False
        self.last_eig_fallback = 0
        self._tls = threading.local()
        self.profile_phases = False    # bench.py: event pairs around the phases of a layer-solve (see _phase)
        self._phase_events = []

    def phase_report(self, reset=True):
        """{phase: summed milliseconds} of the calls bracketed since the last reset (synchronises the device)."""
        out = {}
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        for name, e0, e1 in self._phase_events:
            out[name] = out.get(name, 0.0) + e0.elapsed_time(e1)
        if reset:
            self._phase_events = []
        return out

    # -- helpers ---------------------------------------------------------------------------------------
    @property
    def stream(self):
        if self.device.type == "cuda":
            return torch.cuda.current_stream(self.device).cuda_stream
        return None

    def _ws(self, nbytes):
        return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=self.device)

    def _ints(self, n):
        return torch.zeros(int(n), dtype=torch.int32, device=self.device)

    @staticmethod
    def _c(t):
        return t if t.is_contiguous() else t.contiguous()

    def failures(self):
        """Number of batch entries that reported a numerical failure since the last call (one device sync).  The counters are
        kept per host thread (each thread accumulates on its own stream); they are combined here after a device-wide sync."""
        acc, self._fail_acc = self._fail_acc, {}
        if not acc:
            return 0
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        return sum(int(c) for c in acc.values())

    def _check(self, *tensors):
        """Raw pointers carry no type information: refuse operands on another device or of mixed / non-complex dtypes here,
        instead of an invalid device access or garbage inside the kernel."""
        dt = tensors[0].dtype
        for t in tensors:
            if t.device != self.device and not (t.device.type == self.device.type and self.device.index is None):
                raise ValueError(f"libtrx operand on {t.device}, engine on {self.device}")
            if t.dtype != dt or dt not in _CODE:
                raise TypeError(f"libtrx operands must share one complex dtype (got {[str(x.dtype) for x in tensors]})")

    def _info(self, info, what):
        if not self.check_info:        # deferred, sync-free accounting (throughput runs); read with failures()
            c = (info != 0).sum()
            tid = threading.get_ident()
            prev = self._fail_acc.get(tid)
            self._fail_acc[tid] = c if prev is None else prev + c
            return
        if self.check_info:
            bad = int((info != 0).sum())
            if bad:
                raise NumericalError(f"{what}: {bad} of {info.numel()} batch entries reported a numerical failure "
                                     f"(info={info[info != 0][:8].tolist()})")

    # -- a5 --------------------------------------------------------------------------------------------
    @_phase("assembly (convolution matrices, E^-1, A = PQ)")
    def convmat(self, grid, ox, oy, dtype):
        """[B,nx,ny] real/complex grid -> [B,N,N] convolution matrix (torcwa/rcwa.py:1183-1204)."""
        B, nx, ny = grid.shape
        cplx = grid.is_complex()
        grid = self._c(grid.to(dtype if cplx else _REAL[dtype]))
        N = (2 * ox + 1) * (2 * oy + 1)
        out = torch.empty((B, N, N), dtype=dtype, device=self.device)
        nws = self.lib.convmat_ws_bytes(_CODE[dtype], B, nx, ny, ox, oy)
        ws = self._ws(nws)
        self.lib.check(self.lib.convmat(_CODE[dtype], int(cplx), grid.data_ptr(), B, nx, ny, ox, oy, out.data_ptr(),
                                        ws.data_ptr(), nws, self.stream))
        return out

    # -- dense blocks ----------------------------------------------------------------------------------
    def gemm(self, A, Bm, *, opA=0, opB=0, alpha=1.0, beta=0.0, out=None):
        """Batched C = alpha op(A) op(B) + beta C for contiguous [B,*,*] operands."""
        A, Bm = self._c(A), self._c(Bm)
        self._check(A, Bm)
        dt = A.dtype
        Bt = A.shape[0]
        m = A.shape[1] if opA == 0 else A.shape[2]
        k = A.shape[2] if opA == 0 else A.shape[1]
        n = Bm.shape[2] if opB == 0 else Bm.shape[1]
        if out is None:
            out = torch.empty((Bt, m, n), dtype=dt, device=self.device)
        ctype = ctypes.c_double if dt == torch.complex128 else ctypes.c_float
        al = (ctype * 2)(complex(alpha).real, complex(alpha).imag)
        be = (ctype * 2)(complex(beta).real, complex(beta).imag)
        self.lib.check(self.lib.gemm(_CODE[dt], opA, opB, m, n, k, ctypes.addressof(al), A.data_ptr(), A.shape[2],
                                     A.shape[1] * A.shape[2], Bm.data_ptr(), Bm.shape[2], Bm.shape[1] * Bm.shape[2],
                                     ctypes.addressof(be), out.data_ptr(), n, m * n, Bt, self.stream))
        return out

    @_phase("assembly (convolution matrices, E^-1, A = PQ)")
    def inverse(self, A):
        """Returns inv(A) for [B,n,n] (A is not modified)."""
        self._check(A)
        A = A.clone()
        B, n, _ = A.shape
        piv, info = self._ints(B * n), self._ints(B)
        nws = self.lib.inverse_ws_bytes(_CODE[A.dtype], n, B)
        ws = self._ws(nws)
        self.lib.check(self.lib.inverse(_CODE[A.dtype], A.data_ptr(), n, B, piv.data_ptr(), info.data_ptr(), ws.data_ptr(), nws, self.stream))
        self._info(info, "inverse")
        return A

    @_phase("other (gemm / solve)")
    def solve(self, A, Bm):
        """Returns X with A X = B ([B,n,n], [B,n,r]); inputs are not modified."""
        self._check(A, Bm)
        A, X = A.clone(), Bm.clone()
        B, n, _ = A.shape
        piv, info = self._ints(B * n), self._ints(B)
        self.lib.check(self.lib.lu_solve(_CODE[A.dtype], A.data_ptr(), n, X.data_ptr(), X.shape[2], B, piv.data_ptr(), info.data_ptr(), self.stream))
        self._info(info, "lu_solve")
        return X

    # -- a7 --------------------------------------------------------------------------------------------
    @_phase("eigendecomposition (trx_eig)")
    def eig(self, A, destroy=False, refine_steps=0, route=0):
        """(w [B,n], V [B,n,n]) with A V = V diag(w) (torcwa/torch_eig.py:14).

        refine_steps: Newton steps of libtrx's mixed-precision route (complex128 input of at least 256 rows: fp32 eigendecomposition
        refined in fp64, include/trx.h "eig_refine"); 0 = the library default (2: the accuracy class of the all-fp64 pipeline; measured on
        MI355X, one step is NOT enough for the 1e-5 gate of a complex64 problem at order [15,15] -- the fp32 start of this pipeline leaves
        max |E| ~ 2e-2 ... 2e-1 there).
        route: 0 = the library's automatic choice (mixed precision for complex128 input with n >= 256 and batch >= 8), 1 = all-fp64 (all-fp32 for
        complex64 input), 3 = mixed wherever n >= 8 (include/trx.h "eig_vec").  After the call `eig_fallback_of_last_call()` gives the number of
        matrices the mixed route redid in fp64 (per calling thread)."""
        self._check(A)
        if int(route) not in (0, 1, 3):
            raise ValueError("eig route must be 0 (automatic), 1 (one precision) or 3 (mixed); 2 was the removed inverse-iteration route")
        opts = (int(refine_steps) & 0xF) | ((int(route) & 0xF) << 4)          # per-call option word of trx_eig_opts (no process-global knob is touched: thread-safe)
        # (No route policy lives here any more: matrices the mixed-precision route cannot certify -- clusters of close eigenvalues beyond the
        # refinement's exact treatment -- are redone in fp64 INSIDE the library, as a sub-batch; results do not depend on call history.)
        A = self._c(A) if destroy else A.clone()
        B, n, _ = A.shape
        dt = A.dtype
        w = torch.empty((B, n), dtype=dt, device=self.device)
        V = torch.empty((B, n, n), dtype=dt, device=self.device)
        info = self._ints(B)
        nws = self.lib.eig_ws_bytes_opts(_CODE[dt], n, B, opts)
        ws = self._ws(nws)
        self.lib.check(self.lib.eig_opts(_CODE[dt], A.data_ptr(), w.data_ptr(), V.data_ptr(), n, B, info.data_ptr(), ws.data_ptr(), nws, self.stream, opts))
        fb = int(self.lib.eig_last_fallback())        # matrices of THIS call redone in fp64 (thread-local in the library: the calling thread's last trx_eig)
        self._tls.eig_fallback = fb                   # per host thread: the solver that called reads its own call's count (eig_fallback_of_last_call)
        self.last_eig_fallback = fb                   # diagnostic only: shared by every thread of a multi-stream sweep
        self._info(info, "eig")
        return w, V

    def eig_fallback_of_last_call(self):
        """Matrices the calling thread's last `eig` redid in fp64 (0 if this thread has not called it)."""
        return getattr(self._tls, "eig_fallback", 0)

    @_phase("adjoint (eig backward, solves)")
    def eig_backward(self, w, V, gw, gV, broadening):
        """gA of the reference's broadened eig adjoint (torcwa/torch_eig.py:19-44) in one library call."""
        B, n, _ = V.shape
        dt = V.dtype
        gA = torch.empty((B, n, n), dtype=dt, device=self.device)
        piv, info = self._ints(B * n), self._ints(B)
        nws = self.lib.eig_backward_ws_bytes(_CODE[dt], n, B)
        ws = self._ws(nws)
        self.lib.check(self.lib.eig_backward(_CODE[dt], self._c(w).data_ptr(), self._c(V).data_ptr(), self._c(gw.to(dt)).data_ptr(),
                                             self._c(gV.to(dt)).data_ptr(), float(broadening), n, B, gA.data_ptr(), piv.data_ptr(), info.data_ptr(),
                                             ws.data_ptr(), nws, self.stream))
        self._info(info, "eig_backward")
        return gA

    # -- a6 / a8 / a9 ----------------------------------------------------------------------------------
    @_phase("assembly (convolution matrices, E^-1, A = PQ)")
    def build_pq(self, E, Einv, M, Minv, kx, ky):
        B, N, _ = E.shape
        dt = E.dtype
        P = torch.empty((B, 2 * N, 2 * N), dtype=dt, device=self.device)
        Q = torch.empty_like(P)
        self.lib.check(self.lib.build_pq(_CODE[dt], self._c(E).data_ptr(), self._c(Einv).data_ptr(), self._c(M).data_ptr(), self._c(Minv).data_ptr(),
                                         self._c(kx).data_ptr(), self._c(ky).data_ptr(), N, B, P.data_ptr(), Q.data_ptr(), self.stream))
        return P, Q

    @_phase("layer S-matrix (V = P^-1 W Kz, trx_layer_smatrix)")
    def hmodes(self, E, mu, kx, ky, W, kz):
        """V = P^-1 W diag(kz) for homogeneous mu [B] via the rank-N structure of P (include/trx.h: trx_hmodes)."""
        B, n, _ = W.shape
        N = n // 2
        dt = W.dtype
        V = torch.empty((B, n, n), dtype=dt, device=self.device)
        piv, info = self._ints(B * N), self._ints(B)
        nws = self.lib.hmodes_ws_bytes(_CODE[dt], N, B)
        ws = self._ws(nws)
        self.lib.check(self.lib.hmodes(_CODE[dt], self._c(E).data_ptr(), self._c(mu.to(dt)).data_ptr(), self._c(kx).data_ptr(), self._c(ky).data_ptr(),
                                       self._c(W).data_ptr(), self._c(kz).data_ptr(), N, B, V.data_ptr(), piv.data_ptr(), info.data_ptr(),
                                       ws.data_ptr(), nws, self.stream))
        self._info(info, "hmodes")
        return V

    @_phase("layer S-matrix (V = P^-1 W Kz, trx_layer_smatrix)")
    def layer_smatrix(self, P, Q, W, kzfac, vfinv, phase, *, use_q=False, want_c=True, V=None):
        """Layer S-matrix (torcwa/rcwa.py:1244-1281).  vfinv: [4,B,N]; returns S11, S21, V, Cplus, Cminus.
        V given (from hmodes): the H-field modes are taken as input (use_q = 2 of the C ABI)."""
        B, n, _ = W.shape
        N = n // 2
        dt = W.dtype
        S = torch.empty((2, B, n, n), dtype=dt, device=self.device)     # S11 | S21 contiguous: without coupling coefficients they double
        S11, S21 = S[0], S[1]                                           # as the kernel's last scratch block (include/trx.h)
        if V is not None:
            V, use_q = self._c(V), 2
        else:
            V = torch.empty_like(S11)
        cp = torch.empty_like(S11) if want_c else None
        cm = torch.empty_like(S11) if want_c else None
        piv, info = self._ints(3 * B * n), self._ints(3 * B)
        nws = (self.lib.layer_smatrix_ws_bytes if want_c else self.lib.layer_smatrix_ws_bytes_lean)(_CODE[dt], N, B)
        ws = self._ws(nws)
        W, kzfac, vfinv, phase = self._c(W), self._c(kzfac), self._c(vfinv), self._c(phase)
        self.lib.check(self.lib.layer_smatrix(
            _CODE[dt], self._c(P).data_ptr() if P is not None else None, self._c(Q).data_ptr() if Q is not None else None,
            W.data_ptr(), kzfac.data_ptr(), vfinv.data_ptr(), phase.data_ptr(), int(use_q), N, B, S11.data_ptr(), S21.data_ptr(),
            V.data_ptr(), cp.data_ptr() if want_c else None, cm.data_ptr() if want_c else None, piv.data_ptr(), info.data_ptr(),
            ws.data_ptr(), nws, self.stream))
        self._info(info, "layer_smatrix")
        return S11, S21, V, cp, cm

    @_phase("Redheffer star products")
    def redheffer(self, Sm, Sn):
        """Star product of two S-matrices given as lists [S11,S21,S12,S22] of [B,n,n] (torcwa/rcwa.py:1283-1306).
        Returns (Sout list, X1, X2, Y1, Y2) with the four C-propagation factors."""
        Sm = [self._c(t) for t in Sm]
        Sn = [self._c(t) for t in Sn]
        B, n, _ = Sm[0].shape
        dt = Sm[0].dtype
        out = [torch.empty((B, n, n), dtype=dt, device=self.device) for _ in range(4)]
        XY = torch.empty((2, B, n, 2 * n), dtype=dt, device=self.device)
        piv, info = self._ints(B * n), self._ints(B)
        nws = self.lib.redheffer_ws_bytes(_CODE[dt], n, B)
        ws = self._ws(nws)
        arr = ctypes.c_void_p * 4
        pm, pn, po = arr(*[t.data_ptr() for t in Sm]), arr(*[t.data_ptr() for t in Sn]), arr(*[t.data_ptr() for t in out])
        self.lib.check(self.lib.redheffer(_CODE[dt], ctypes.addressof(pm), ctypes.addressof(pn), ctypes.addressof(po), XY.data_ptr(), n, B,
                                          piv.data_ptr(), info.data_ptr(), ws.data_ptr(), nws, self.stream))
        self._info(info, "redheffer")
        X, Y = XY[0], XY[1]
        return out, X[:, :, :n], X[:, :, n:], Y[:, :, :n], Y[:, :, n:]


    @_phase("Redheffer star products")
    def redheffer_halfspace(self, side, bd, S, want_xy=True):
        """Star product with a block-diagonal half-space S-matrix (side 0: Sin * S, side 1: S * Sout).
        bd: [4,4,B,N] diagonals (see include/trx.h).  want_xy=False (no coupling lists to propagate) lets side 0 use the
        cheaper right-solve algebra; the factor slots of the result are then None."""
        S = [self._c(t) for t in S]
        B, n, _ = S[0].shape
        N = n // 2
        dt = S[0].dtype
        bd = self._c(bd.to(dt))
        out = [torch.empty((B, n, n), dtype=dt, device=self.device) for _ in range(4)]
        lean = (int(side) == 0) and not want_xy
        XY = None if lean else torch.empty((2, B, n, 2 * n), dtype=dt, device=self.device)
        piv, info = self._ints(B * n), self._ints(B)
        nws = self.lib.redheffer_halfspace_ws_bytes(_CODE[dt], N, B, int(side), 0 if lean else 1)
        ws = self._ws(nws)
        arr = ctypes.c_void_p * 4
        ps, po = arr(*[t.data_ptr() for t in S]), arr(*[t.data_ptr() for t in out])
        self.lib.check(self.lib.redheffer_halfspace(_CODE[dt], int(side), bd.data_ptr(), ctypes.addressof(ps), ctypes.addressof(po),
                                                    None if lean else XY.data_ptr(), N, B, piv.data_ptr(), info.data_ptr(), ws.data_ptr(), nws, self.stream))
        self._info(info, "redheffer_halfspace")
        if lean:
            return out, None, None, None, None
        X, Y = XY[0], XY[1]
        return out, X[:, :, :n], X[:, :, n:], Y[:, :, :n], Y[:, :, n:]

    @_phase("assembly (convolution matrices, E^-1, A = PQ)")
    def build_a(self, E, Einv, mu, kx, ky):
        """A = P Q for homogeneous mu [B] via the block structure (two N^3 GEMMs)."""
        B, N, _ = E.shape
        dt = E.dtype
        A = torch.empty((B, 2 * N, 2 * N), dtype=dt, device=self.device)
        nws = self.lib.build_a_ws_bytes(_CODE[dt], N, B)
        ws = self._ws(nws)
        self.lib.check(self.lib.build_a(_CODE[dt], self._c(E).data_ptr(), self._c(Einv).data_ptr(), self._c(mu.to(dt)).data_ptr(), self._c(kx).data_ptr(),
                                        self._c(ky).data_ptr(), N, B, A.data_ptr(), ws.data_ptr(), nws, self.stream))
        return A


_default = None


def default_engine():
    global _default
    if _default is None:
        _default = Engine()
    return _default
