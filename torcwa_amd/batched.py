"""Batched RCWA solver: B independent sweep points (frequency / angle / geometry) advance in lock-step.

This is the MI355X-native restatement of the hot path of torcwa.rcwa (kch3782/torcwa 0.1.4.2, torcwa/rcwa.py).  The
reference is un-batched (one object per sweep point, a Python `for` loop over points, e.g. example/Example1.ipynb
cell "lamb0 sweep"); here the sweep axis is the leading tensor dimension and every heavy operation is one batched
libtrx call (include/trx.h).  The drop-in class `torcwa_amd.rcwa` is the B=1 view of this class.

All per-point scalars (`freq`, angles, homogeneous eps/mu, thickness) may be python scalars or [B] tensors.
"""
import warnings

import torch

from .engine import Engine, default_engine
from . import autograd_ops as ag
from .torch_eig import Eig

# torcwa/rcwa.py:5 -- the reference's pi (typo in the 9th decimal) is part of its observable behaviour
PI_REF = 3.141592652589793

_DIRS = {"f": "forward", "forward": "forward", "b": "backward", "backward": "backward"}
_PORTS = {"t": "transmission", "transmission": "transmission", "r": "reflection", "reflection": "reflection"}
_SBLOCK = {("forward", "transmission"): 0, ("forward", "reflection"): 1, ("backward", "reflection"): 2, ("backward", "transmission"): 3}
# (numerator side, denominator side) of the power normalisation for S[k]            rcwa.py:377-388
_KZ_SIDES = {0: ("out", "in"), 1: ("in", "in"), 2: ("out", "out"), 3: ("in", "out")}


class BlockDiag2:
    """2x2 block matrix whose four N x N blocks are diagonal (Vf, Vi, Vo, Sin, Sout all have this form,
    rcwa.py:1143-1181): stored as four [B,N] diagonals, O(N) algebra instead of dense n^3."""

    def __init__(self, d11, d12, d21, d22):
        self.d = (d11, d12, d21, d22)

    def __add__(self, o):
        return BlockDiag2(*[a + b for a, b in zip(self.d, o.d)])

    def __sub__(self, o):
        return BlockDiag2(*[a - b for a, b in zip(self.d, o.d)])

    def __neg__(self):
        return BlockDiag2(*[-a for a in self.d])

    def scale(self, s):
        return BlockDiag2(*[s * a for a in self.d])

    def __matmul__(self, o):
        a, b, c, d = self.d
        e, f, g, h = o.d
        return BlockDiag2(a * e + b * g, a * f + b * h, c * e + d * g, c * f + d * h)

    def inv(self):
        a, b, c, d = self.d
        det = a * d - b * c
        return BlockDiag2(d / det, -b / det, -c / det, a / det)

    def dense(self):
        a, b, c, d = self.d
        top = torch.cat((torch.diag_embed(a), torch.diag_embed(b)), dim=2)
        bot = torch.cat((torch.diag_embed(c), torch.diag_embed(d)), dim=2)
        return torch.cat((top, bot), dim=1)


def _halfspace_V(kx, ky, epsmu):
    """E->H map of a homogeneous half space, eps*mu = epsmu ([B] or scalar)        rcwa.py:1143-1147"""
    kz = torch.sqrt(epsmu - kx ** 2 - ky ** 2)
    kz = torch.where(torch.imag(kz) < 0, torch.conj(kz), kz)
    return BlockDiag2(-ky * kx / kz, -kz - ky ** 2 / kz, kz + kx ** 2 / kz, kx * ky / kz)


class BatchedRCWA:
    def __init__(self, freq, order, L, *, batch=None, dtype=torch.complex64, device=None, stable_eig_grad=True,
                 avoid_Pinv_instability=False, max_Pinv_instability=0.005, precision="high", engine=None,
                 keep_coupling=True, fold_layers=False, eig_route="auto", route_hint=None):
        if dtype != torch.complex64 and dtype != torch.complex128:                      # rcwa.py:37-41
            warnings.warn("Invalid simulation data type. Set as torch.complex64.", UserWarning)
            dtype = torch.complex64
        self._dtype = dtype
        # eigensolver route of THIS solver: "auto" | "mixed" | "fp64" (see _eig_call); route_hint: a dict shared by the chunks of one sweep call
        self.eig_route = eig_route
        self._route_hint = route_hint if route_hint is not None else {}
        self.engine = engine if engine is not None else (default_engine() if device is None else Engine(device=device))
        self._device = self.engine.device
        # precision="high": c64 problems are computed in complex128 internally (the reference's own c64 path is only
        # ~1e-3 accurate at order 15; the <=1e-5 parity gate needs fp64 in eig and LU).
        self._cdtype = torch.complex128 if (precision == "high" or dtype == torch.complex128) else torch.complex64
        self._rdtype = torch.float64 if self._cdtype == torch.complex128 else torch.float32
        self.stable_eig_grad = bool(stable_eig_grad)
        self.avoid_Pinv_instability = avoid_Pinv_instability is True
        self.max_Pinv_instability = max_Pinv_instability if self.avoid_Pinv_instability else None
        self.Pinv_instability = [] if self.avoid_Pinv_instability else None
        self.Qinv_instability = [] if self.avoid_Pinv_instability else None
        self.keep_coupling = keep_coupling
        # fold_layers (sweep drivers; needs keep_coupling=False): every layer's S-matrix is folded into the running cascade as soon as it
        # exists and is then dropped together with the layer's convolution matrix, so a K-layer stack holds ONE layer at a time instead
        # of K (configs[2]: 4 layers at n = 3698).  The per-layer attributes of such a solver are None.
        self.fold_layers = bool(fold_layers) and not keep_coupling
        self._running = None
        self._n_folded = 0            # layers 0 .. _n_folded-1 live in _running; the rest are stored layers

        if batch is None:
            batch = freq.numel() if (torch.is_tensor(freq) and freq.dim() > 0) else (len(freq) if isinstance(freq, (list, tuple)) else 1)
        self.B = int(batch)
        self.freq = self._bvec(freq)                                                    # [B] complex
        self.omega = (2 * PI_REF) * torch.real(self.freq)                               # rcwa.py:61  [B] real
        self.order = [int(order[0]), int(order[1])]
        self.order_x = torch.arange(-self.order[0], self.order[0] + 1, dtype=torch.int64, device=self._device)
        self.order_y = torch.arange(-self.order[1], self.order[1] + 1, dtype=torch.int64, device=self._device)
        self.order_N = len(self.order_x) * len(self.order_y)
        self.L = L
        self.Gx_norm = 1 / (L[0] * self.freq)                                           # rcwa.py:72
        self.Gy_norm = 1 / (L[1] * self.freq)
        one = torch.ones(self.B, dtype=self._cdtype, device=self._device)
        self.eps_in, self.mu_in, self.eps_out, self.mu_out = one, one.clone(), one.clone(), one.clone()
        self.has_in = self.has_out = False
        self.layer_N = 0
        self.thickness = []
        self.eps_conv, self.mu_conv = [], []
        self.P, self.Q = [], []
        self.kz_norm, self.E_eigvec, self.H_eigvec = [], [], []
        self.Cplus, self.Cminus = [], []
        self.layer_S11, self.layer_S21 = [], []

    # ---- helpers -----------------------------------------------------------------------------------------
    def _bvec(self, v, dtype=None):
        """python scalar / 0-d / [B] tensor -> [B] tensor of the compute dtype."""
        dt = dtype if dtype is not None else self._cdtype
        if torch.is_tensor(v):
            t = v.to(device=self._device, dtype=dt)
        else:                      # python scalars / lists: build directly in the target precision (no fp32 detour)
            t = torch.as_tensor(v, dtype=dt, device=self._device)
        if t.dim() == 0:
            t = t.expand(self.B).clone()
        return t.reshape(self.B)

    @property
    def n(self):
        return 2 * self.order_N

    # ---- a2 / a3 -----------------------------------------------------------------------------------------
    def add_input_layer(self, eps=1., mu=1.):                                           # rcwa.py:95-107
        self.eps_in, self.mu_in, self.has_in = self._bvec(eps), self._bvec(mu), True

    def add_output_layer(self, eps=1., mu=1.):                                          # rcwa.py:109-121
        self.eps_out, self.mu_out, self.has_out = self._bvec(eps), self._bvec(mu), True

    def set_incident_angle(self, inc_ang, azi_ang, angle_layer="input"):                # rcwa.py:123-144
        self.inc_ang, self.azi_ang = self._bvec(inc_ang), self._bvec(azi_ang)
        if angle_layer in ("i", "in", "input"):
            self.angle_layer = "input"
        elif angle_layer in ("o", "out", "output"):
            self.angle_layer = "output"
        else:
            warnings.warn("Invalid angle layer. Set as input layer.", UserWarning)
            self.angle_layer = "input"
        self._kvectors()

    def _kvectors(self):                                                                # rcwa.py:1124-1181
        em = self.eps_in * self.mu_in if self.angle_layer == "input" else self.eps_out * self.mu_out
        nref = torch.real(torch.sqrt(em))
        self.kx0_norm = nref * torch.sin(self.inc_ang) * torch.cos(self.azi_ang)
        self.ky0_norm = nref * torch.sin(self.inc_ang) * torch.sin(self.azi_ang)
        kx = self.kx0_norm[:, None] + self.order_x[None, :] * self.Gx_norm[:, None]     # [B, 2ox+1]
        ky = self.ky0_norm[:, None] + self.order_y[None, :] * self.Gy_norm[:, None]     # [B, 2oy+1]
        self.kx_norm, self.ky_norm = kx, ky
        self.Kx_norm_dn = kx[:, :, None].expand(-1, -1, ky.shape[1]).reshape(self.B, -1).contiguous()   # x-major
        self.Ky_norm_dn = ky[:, None, :].expand(-1, kx.shape[1], -1).reshape(self.B, -1).contiguous()
        kxd, kyd = self.Kx_norm_dn, self.Ky_norm_dn
        self._Vf = _halfspace_V(kxd, kyd, 1.0)
        self._Vfinv = self._Vf.inv()
        self._Sin = self._Sout = None
        if self.has_in:                                                                 # rcwa.py:1149-1164
            self._Vi = _halfspace_V(kxd, kyd, (self.eps_in * self.mu_in)[:, None])
            T = (self._Vf + self._Vi).inv()
            D = self._Vf - self._Vi
            self._Sin = [(T @ self._Vi).scale(2), -(T @ D), T @ D, (T @ self._Vf).scale(2)]
        if self.has_out:                                                                # rcwa.py:1166-1181
            self._Vo = _halfspace_V(kxd, kyd, (self.eps_out * self.mu_out)[:, None])
            T = (self._Vf + self._Vo).inv()
            D = self._Vf - self._Vo
            self._Sout = [(T @ self._Vf).scale(2), T @ D, -(T @ D), (T @ self._Vo).scale(2)]

    # ---- a4-a8 -------------------------------------------------------------------------------------------
    def add_layer(self, thickness, eps=1., mu=1.):                                      # rcwa.py:146-170
        eng, N, B, cdt = self.engine, self.order_N, self.B, self._cdtype
        eps_h, mu_h = self._is_homogeneous(eps), self._is_homogeneous(mu)
        eye = torch.eye(N, dtype=cdt, device=self._device)
        # differentiable path (Examples 4-6): every O(n^3) primitive is an autograd.Function over the same HIP kernels
        diff = torch.is_grad_enabled() and any(torch.is_tensor(v) and v.requires_grad for v in
                                               (thickness, eps, mu, self.freq, self.Kx_norm_dn, self.Ky_norm_dn))
        self._diff = getattr(self, "_diff", False) or diff

        def conv(v, homog):
            if homog:
                s = self._bvec(v)
                return s[:, None, None] * eye, (1 / s)[:, None, None] * eye, s
            g = torch.as_tensor(v, device=self._device)
            if g.dim() == 2:
                g = g[None].expand(B, -1, -1)
            if diff:
                return ag.ConvMatFn.apply(g.contiguous(), self.order[0], self.order[1], cdt, eng), None, None
            C = eng.convmat(g.contiguous(), self.order[0], self.order[1], cdt)          # rcwa.py:1183-1204
            return C, None, None

        if eps_h and mu_h and not diff and not self.keep_coupling:
            self._add_homogeneous_layer_bd(thickness, self._bvec(eps), self._bvec(mu))
            self._fold_last_layer()
            return
        # Sweep drivers (keep_coupling=False) with a homogeneous mu never read P, Q, the dense mu matrices or, after the layer's
        # S-matrix, the mode matrices W, V: A = PQ and V = P^-1 W Kz come from E directly (trx_build_a / trx_hmodes).  Not building
        # / not keeping them takes 4 of ~16 n^2-sized tensors per sweep point out of the peak.
        lean = (not diff) and (not self.keep_coupling) and mu_h and (not eps_h) and (not self.avoid_Pinv_instability)
        E, Einv, eps_s = conv(eps, eps_h)
        if lean:
            M, Minv, mu_s = None, None, self._bvec(mu)
        else:
            M, Minv, mu_s = conv(mu, mu_h)
        self.eps_conv.append(E)
        self.mu_conv.append(M)
        self.layer_N += 1
        d = self._bvec(thickness, self._rdtype)
        self.thickness.append(d)
        kxd, kyd = self.Kx_norm_dn, self.Ky_norm_dn
        inv = (lambda A: ag.InverseFn.apply(A, eng)) if diff else eng.inverse
        if Einv is None:
            Einv = inv(E)
        if Minv is None and M is not None:
            Minv = inv(M)
        if diff:
            P, Q = self._pq_torch(E, Einv, M, Minv, kxd, kyd)
        elif lean:
            P = Q = None
        else:
            P, Q = eng.build_pq(E, Einv, M, Minv, kxd, kyd)
        if eps_h and mu_h:                                                              # rcwa.py:1206-1222
            W = torch.eye(2 * N, dtype=cdt, device=self._device).expand(B, -1, -1).contiguous()
            kz = torch.sqrt((eps_s * mu_s)[:, None] - kxd ** 2 - kyd ** 2)
            kz = torch.where(torch.imag(kz) < 0, torch.conj(kz), kz)
            kz = torch.cat((kz, kz), dim=1)
        else:                                                                           # rcwa.py:1224-1242
            if diff:
                Eig.engine = eng
                A = ag.GemmFn.apply(P, Q, eng)
                # stable_eig_grad=False is the reference's plain torch.linalg.eig branch (rcwa.py:1238): its backward is never
                # broadened.  The choice is bound to this graph node (not to the process-global at backward time).
                lam, W = Eig.apply(A) if self.stable_eig_grad else Eig.apply(A, Eig.UNBROADENED)
            else:
                # A = P Q (rcwa.py:1236): with homogeneous mu the block structure needs two N^3 products, not one (2N)^3
                A = eng.build_a(E, Einv, mu_s, kxd, kyd) if mu_h else eng.gemm(P, Q)
                del Einv
                # mixed-precision eigensolver: two Newton steps for a complex64 problem (1e-5 gate), three for complex128 (engine.eig)
                lam, W = self._eig_call(A, refine_steps=3 if self._dtype == torch.complex128 else 2)      # torch_eig.py:14
                del A
            kz = torch.sqrt(lam)
            kz = torch.where(torch.imag(kz) < 0, -kz, kz)                               # rcwa.py:1241
        self.P.append(P)
        self.Q.append(Q)
        self.kz_norm.append(kz)
        self.E_eigvec.append(W)
        self._mu_scalar = mu_s if mu_h else None       # homogeneous mu: V = P^-1 W Kz from the rank-N structure of P (trx_hmodes)
        if diff:
            self._solve_layer_smatrix_diff()
        else:
            self._solve_layer_smatrix()
        self._fold_last_layer()

    def _fold_last_layer(self):
        """Streaming cascade: fold the layer just added into the running star product and drop it.  Decided PER LAYER: a layer is folded
        only while the stack so far is a plain (non-differentiable) prefix 0 .. i-1 that has itself been folded; from the first
        differentiable layer on, layers stay stored and solve_global_smatrix continues the cascade from the folded prefix over them
        (a global early return here used to leave such layers out of the cascade altogether)."""
        i = self.layer_N - 1
        if not self.fold_layers or getattr(self, "_diff", False) or self._n_folded != i:
            return
        S = self._layer_S(i)
        if self._running is None:
            self._running = S
        else:
            self._running, _ = self._star(self._running, S, [[], []], [[], []])
        self.layer_S11[i] = self.layer_S21[i] = None
        self.eps_conv[i] = None
        self._n_folded = i + 1

    def _add_homogeneous_layer_bd(self, thickness, eps_s, mu_s):
        """Homogeneous layer without field bookkeeping (keep_coupling=False): every operator of rcwa.py:1206-1222 and 1244-1281
        is 2x2-block-diagonal (W = I, V = P^-1 Kz = Q Kz^-1 in closed form), so the layer S-matrix is four [B,N] diagonals per
        block and costs O(N) instead of the dense eigen/LU path; the cascade then uses the half-space star product.  The dense
        per-layer attributes (P, Q, E_eigvec, H_eigvec, eps_conv, mu_conv) are not materialised for such a layer (None)."""
        kxd, kyd = self.Kx_norm_dn, self.Ky_norm_dn
        d = self._bvec(thickness, self._rdtype)
        epsmu = (eps_s * mu_s)[:, None]
        kz = torch.sqrt(epsmu - kxd ** 2 - kyd ** 2)
        kz = torch.where(torch.imag(kz) < 0, torch.conj(kz), kz)                        # rcwa.py:1218
        x = torch.exp(1j * (self.omega * d)[:, None] * kz)                              # rcwa.py:1246, [B,N] (same for both field components)
        V = _halfspace_V(kxd, kyd, epsmu).scale(1 / mu_s[:, None])                      # Q Kz^-1 = P^-1 Kz
        one, zero = torch.ones_like(kz), torch.zeros_like(kz)
        I = BlockDiag2(one, zero, zero, one)
        F = self._Vfinv @ V
        A_, B_ = I + F, (I - F).scale(x)
        Tip, Tim = (A_ + B_).inv(), (A_ - B_).inv()
        cp, cm = Tip + Tim, Tip - Tim
        self.layer_S11.append(cp.scale(x) + cm)                                         # rcwa.py:1276 with W = I
        self.layer_S21.append(cp + cm.scale(x) - I)                                     # rcwa.py:1277
        self.layer_N += 1
        self.thickness.append(d)
        self.kz_norm.append(torch.cat((kz, kz), dim=1))
        for lst in (self.eps_conv, self.mu_conv, self.P, self.Q, self.E_eigvec, self.H_eigvec, self.Cplus, self.Cminus):
            lst.append(None)

    @staticmethod
    def _pq_torch(E, Ei, M, Mi, kx, ky):
        """P, Q (rcwa.py:1226-1232) with broadcasting instead of dense diagonal products (differentiable)."""
        kxi, kyi, kxj, kyj = kx[:, :, None], ky[:, :, None], kx[:, None, :], ky[:, None, :]
        P = torch.cat((torch.cat((kxi * Ei * kyj, M - kxi * Ei * kxj), dim=2),
                       torch.cat((kyi * Ei * kyj - M, -(kyi * Ei * kxj)), dim=2)), dim=1)
        Q = torch.cat((torch.cat((-(kxi * Mi * kyj), kxi * Mi * kxj - E), dim=2),
                       torch.cat((E - kyi * Mi * kyj, kyi * Mi * kxj), dim=2)), dim=1)
        return P, Q

    def _solve_layer_smatrix_diff(self):
        """Layer S-matrix (rcwa.py:1244-1281) from differentiable primitives; same lean algebra as trx_layer_smatrix."""
        eng, N, n = self.engine, self.order_N, self.n
        P, W, kz, d = self.P[-1], self.E_eigvec[-1], self.kz_norm[-1], self.thickness[-1]
        X = torch.exp(1j * (self.omega * d)[:, None] * kz)                              # [B, n]
        WKz = W * kz[:, None, :]
        bad = None
        if self.avoid_Pinv_instability:                                                 # rcwa.py:1249-1262 (metrics are detached diagnostics)
            with torch.no_grad():
                Pd, Qd = P.detach(), self.Q[-1].detach()
                I = torch.eye(n, dtype=self._cdtype, device=self._device)
                Pinv = eng.inverse(Pd)
                ins = torch.maximum(torch.amax(torch.abs(eng.gemm(Pd, Pinv) - I), dim=(1, 2)),
                                    torch.amax(torch.abs(eng.gemm(Pinv, Pd) - I), dim=(1, 2)))
                qins = torch.amax(torch.abs(eng.gemm(Qd, eng.inverse(Qd)) - I), dim=(1, 2))
            self.Pinv_instability.append(ins)
            self.Qinv_instability.append(qins)
            bad = ins >= self.max_Pinv_instability
            bad = bad if bool(bad.any()) else None
        if bad is None:
            V = ag.SolveFn.apply(P, WKz, eng)                                           # P^-1 W Kz
        else:
            # V = Q W Kz^-1 for the ill-conditioned points.  The reference routes the graph of such a point through that branch only;
            # in a batch both branches are evaluated, so the discarded P-solve of a bad point must not be able to produce Inf / NaN
            # (0 * NaN in SolveFn.backward would poison the gradient of every point): those points solve with the identity instead
            sel = bad[:, None, None]
            I = torch.eye(n, dtype=self._cdtype, device=self._device)
            V = torch.where(sel, ag.GemmFn.apply(self.Q[-1], (W / kz[:, None, :]).contiguous(), eng),
                            ag.SolveFn.apply(torch.where(sel, I, P), WKz, eng))
        p11, p12, p21, p22 = [t.to(self._cdtype)[:, :, None] for t in self._Vfinv.d]
        F = torch.cat((p11 * V[:, :N] + p12 * V[:, N:], p21 * V[:, :N] + p22 * V[:, N:]), dim=1)    # Vf^-1 V
        A_, B_ = W + F, (W - F) * X[:, None, :]
        Tip, Tim = ag.InverseFn.apply(A_ + B_, eng), ag.InverseFn.apply(A_ - B_, eng)
        cp, cm = Tip + Tim, Tip - Tim
        I = torch.eye(n, dtype=self._cdtype, device=self._device)
        S11 = ag.GemmFn.apply(W, X[:, :, None] * cp + cm, eng)
        S21 = ag.GemmFn.apply(W, cp + X[:, :, None] * cm, eng) - I
        self.H_eigvec.append(V)
        self.layer_S11.append(S11)
        self.layer_S21.append(S21)
        self.Cplus.append(cp)
        self.Cminus.append(cm)

    def _RS_prod_diff(self, Sm, Sn, Cm, Cn):
        """Star product (rcwa.py:1283-1306) from differentiable primitives (one factorisation, push-through identity)."""
        eng, n = self.engine, self.n
        I = torch.eye(n, dtype=self._cdtype, device=self._device)
        mm = lambda a, b: ag.GemmFn.apply(a.contiguous(), b.contiguous(), eng)
        K = I - mm(Sm[2], Sn[1])
        X12 = ag.SolveFn.apply(K, torch.cat((Sm[0], mm(Sm[2], Sn[3])), dim=2).contiguous(), eng)
        X1, X2 = X12[:, :, :n], X12[:, :, n:]
        Y1, Y2 = mm(Sn[1], X1), Sn[3] + mm(Sn[1], X2)
        S = [mm(Sn[0], X1), Sm[1] + mm(Sm[3], Y1), Sn[2] + mm(Sn[0], X2), mm(Sm[3], Y2)]
        C = [[], []]
        for m in range(len(Cm[0])):
            C[0].append(Cm[0][m] + mm(Cm[1][m], Y1))
            C[1].append(mm(Cm[1][m], Y2))
        for k in range(len(Cn[0])):
            C[0].append(mm(Cn[0][k], X1))
            C[1].append(Cn[1][k] + mm(Cn[0][k], X2))
        return S, C

    def _eig_call(self, A, refine_steps):
        """trx_eig with this solver's route policy.  "fp64" / "mixed": as named.  "auto": the library's mixed-precision route, and -- scoped to THIS
        solver object (or to the one sweep call whose chunks share `route_hint`), never beyond -- once a call had to redo matrices in fp64
        (clusters of close eigenvalues beyond the refinement's exact treatment: symmetric meta-atoms, the dense spectra of large orders), the
        following eigenproblems of the same size go to the all-fp64 route directly instead of paying for another failed attempt.  The hint is
        created with the solver / sweep call and dies with it: results never depend on what the process solved before."""
        eng = self.engine
        key = int(A.shape[-1])
        route = {"auto": 0, "mixed": 3, "fp64": 1}[self.eig_route]
        if self.eig_route == "auto" and self._route_hint.get(key):
            route = 1
        lam, W = eng.eig(A, destroy=True, refine_steps=refine_steps, route=route)
        if self.eig_route == "auto" and route == 0 and eng.eig_fallback_of_last_call() > 0:
            self._route_hint[key] = True
        return lam, W

    def _is_homogeneous(self, v):
        if isinstance(v, (float, complex)):
            return True
        if isinstance(v, int):                     # the reference raises AttributeError here (rcwa.py:156)
            raise AttributeError("'int' object has no attribute 'dim'")
        t = torch.as_tensor(v)
        # rcwa.py:156-157: a 0-d tensor or a 1-D tensor of length 1.  Batched extension: a 1-D tensor of length B is one homogeneous value per
        # sweep point; any other 1-D tensor is NOT homogeneous, exactly as in the reference (it then fails in the grid path like there).
        return t.dim() == 0 or (t.dim() == 1 and t.shape[0] in (1, self.B))

    def _solve_layer_smatrix(self):                                                     # rcwa.py:1244-1281
        eng = self.engine
        P, Q, W, kz, d = self.P[-1], self.Q[-1], self.E_eigvec[-1], self.kz_norm[-1], self.thickness[-1]
        phase = torch.exp(1j * (self.omega * d)[:, None] * kz)                          # rcwa.py:1246
        vfinv = torch.stack(self._Vfinv.d, dim=0).to(self._cdtype).contiguous()         # [4,B,N]
        if not self.avoid_Pinv_instability:
            if self._mu_scalar is not None:                                             # rcwa.py:1264, structured (include/trx.h: trx_hmodes)
                Vh = eng.hmodes(self.eps_conv[-1], self._mu_scalar, self.Kx_norm_dn, self.Ky_norm_dn, W, kz)
                S11, S21, V, cp, cm = eng.layer_smatrix(None, None, W, kz, vfinv, phase, want_c=self.keep_coupling, V=Vh)
            else:
                S11, S21, V, cp, cm = eng.layer_smatrix(P, None, W, kz, vfinv, phase, use_q=False, want_c=self.keep_coupling)
        else:                                                                           # rcwa.py:1249-1262
            n = self.n
            I = torch.eye(n, dtype=self._cdtype, device=self._device)
            Pinv = eng.inverse(P)
            ins1 = torch.amax(torch.abs(eng.gemm(P, Pinv) - I), dim=(1, 2))
            ins2 = torch.amax(torch.abs(eng.gemm(Pinv, P) - I), dim=(1, 2))
            qins = torch.amax(torch.abs(eng.gemm(Q, eng.inverse(Q)) - I), dim=(1, 2))   # computed twice in the reference
            self.Pinv_instability.append(torch.maximum(ins1, ins2))
            self.Qinv_instability.append(qins)
            bad = self.Pinv_instability[-1] >= self.max_Pinv_instability
            S11, S21, V, cp, cm = eng.layer_smatrix(P, None, W, kz, vfinv, phase, use_q=False, want_c=self.keep_coupling)
            if bool(bad.any()):
                alt = eng.layer_smatrix(None, Q, W, 1 / kz, vfinv, phase, use_q=True, want_c=self.keep_coupling)
                sel = bad[:, None, None]
                S11, S21, V = torch.where(sel, alt[0], S11), torch.where(sel, alt[1], S21), torch.where(sel, alt[2], V)
                if self.keep_coupling:
                    cp, cm = torch.where(sel, alt[3], cp), torch.where(sel, alt[4], cm)
        if not self.keep_coupling and not self.avoid_Pinv_instability and self._mu_scalar is not None:
            self.E_eigvec[-1] = None          # sweep drivers: the mode matrices are not read again (see add_layer)
            V = None
        self.H_eigvec.append(V)
        self.layer_S11.append(S11)
        self.layer_S21.append(S21)
        self.Cplus.append(cp)
        self.Cminus.append(cm)

    # ---- a9 / a10 ----------------------------------------------------------------------------------------
    def _RS_prod(self, Sm, Sn, Cm, Cn):                                                 # rcwa.py:1283-1306
        eng = self.engine
        if getattr(self, "_diff", False):
            return self._RS_prod_diff(Sm, Sn, Cm, Cn)
        S, X1, X2, Y1, Y2 = eng.redheffer(Sm, Sn)
        C = [[], []]
        for m in range(len(Cm[0])):
            C[0].append(Cm[0][m] + eng.gemm(Cm[1][m], Y1.contiguous()))
            C[1].append(eng.gemm(Cm[1][m], Y2.contiguous()))
        for k in range(len(Cn[0])):
            C[0].append(eng.gemm(Cn[0][k], X1.contiguous()))
            C[1].append(Cn[1][k] + eng.gemm(Cn[0][k], X2.contiguous()))
        return S, C

    def _RS_halfspace(self, side, Sbd, S, C):
        """Star product with Sin (side 0) / Sout (side 1), whose blocks are 2x2-block-diagonal: O(n^2) products with them."""
        eng = self.engine
        if getattr(self, "_diff", False):
            dense = [blk.dense().to(self._cdtype) for blk in Sbd]
            return self._RS_prod_diff(dense, S, [[], []], C) if side == 0 else self._RS_prod_diff(S, dense, C, [[], []])
        bd = torch.stack([torch.stack(blk.d, dim=0) for blk in Sbd], dim=0).to(self._cdtype).contiguous()   # [4,4,B,N]
        Sn, X1, X2, Y1, Y2 = eng.redheffer_halfspace(side, bd, S, want_xy=len(C[0]) > 0)
        Cn = [[], []]
        if side == 0:                       # C belongs to the right operand (rcwa.py:1302-1304)
            for k in range(len(C[0])):
                Cn[0].append(eng.gemm(C[0][k], X1.contiguous()))
                Cn[1].append(C[1][k] + eng.gemm(C[0][k], X2.contiguous()))
        else:                               # C belongs to the left operand (rcwa.py:1298-1300)
            for m in range(len(C[0])):
                Cn[0].append(C[0][m] + eng.gemm(C[1][m], Y1.contiguous()))
                Cn[1].append(eng.gemm(C[1][m], Y2.contiguous()))
        return Sn, Cn

    @staticmethod
    def _is_bd(S):
        return isinstance(S[0], BlockDiag2)

    @staticmethod
    def _RS_bd_bd(Sm, Sn):
        """Star product of two block-diagonal S-matrices (rcwa.py:1287-1296), O(N)."""
        one, zero = torch.ones_like(Sm[0].d[0]), torch.zeros_like(Sm[0].d[0])
        I = BlockDiag2(one, zero, zero, one)
        t1 = (I - Sm[2] @ Sn[1]).inv()
        t2 = (I - Sn[1] @ Sm[2]).inv()
        return [Sn[0] @ t1 @ Sm[0], Sm[1] + Sm[3] @ t2 @ Sn[1] @ Sm[0], Sn[2] + Sn[0] @ t1 @ Sm[2] @ Sn[3], Sm[3] @ t2 @ Sn[3]]

    def _star(self, Sm, Sn, Cm, Cn):
        """Sm * Sn for any mix of dense ([B,n,n] tensors) and block-diagonal (BlockDiag2) operands."""
        bm, bn = self._is_bd(Sm), self._is_bd(Sn)
        if bm and bn:
            return self._RS_bd_bd(Sm, Sn), [[], []]
        if bm:
            return self._RS_halfspace(0, Sm, Sn, Cn)
        if bn:
            return self._RS_halfspace(1, Sn, Sm, Cm)
        return self._RS_prod(Sm, Sn, Cm, Cn)

    def _layer_S(self, i):
        # the layer S-matrix is symmetric under port exchange: S22 = S11, S12 = S21
        return [self.layer_S11[i], self.layer_S21[i], self.layer_S21[i], self.layer_S11[i]]

    def _layer_C(self, i):
        if not self.keep_coupling or self.Cplus[i] is None:
            return [[], []]
        return [[torch.cat((self.Cplus[i], self.Cminus[i]), dim=1)], [torch.cat((self.Cminus[i], self.Cplus[i]), dim=1)]]

    def solve_global_smatrix(self):                                                     # rcwa.py:173-211
        n, B = self.n, self.B
        self._zero_layer_S = False
        first = 1                                                                       # first stored layer still to be folded in
        if self._n_folded > 0:
            S, C, first = self._running, [[], []], self._n_folded
        elif self.layer_N > 0:
            S = self._layer_S(0)
            C = self._layer_C(0)
        else:
            I = torch.eye(n, dtype=self._cdtype, device=self._device).expand(B, -1, -1).contiguous()
            Z = torch.zeros((B, n, n), dtype=self._cdtype, device=self._device)
            S = [I, Z, Z.clone(), I.clone()]
            C = [[], []]
            self._zero_layer_S = not (self.has_in or self.has_out)     # reference stores 1-D zeros (rcwa.py:187-188)
        for i in range(first, self.layer_N):
            S, C = self._star(S, self._layer_S(i), C, self._layer_C(i))
        if self.has_in:                                                                 # rcwa.py:198-202
            S, C = self._star(self._Sin, S, [[], []], C)
        if self.has_out:                                                                # rcwa.py:204-208
            S, C = self._star(S, self._Sout, C, [[], []])
        if self._is_bd(S):                                                              # only homogeneous media: densify for the read-out
            S = [blk.dense().to(self._cdtype) for blk in S]
        self.S = S
        self.C = C

    # ---- a11 ---------------------------------------------------------------------------------------------
    def _matching_indices(self, orders):                                                # rcwa.py:1115-1122
        ox, oy = self.order
        orders[orders[:, 0] < -ox, 0] = -ox
        orders[orders[:, 0] > ox, 0] = ox
        orders[orders[:, 1] < -oy, 1] = -oy
        orders[orders[:, 1] > oy, 1] = oy
        return len(self.order_y) * (orders[:, 0] + ox) + orders[:, 1] + oy

    def _kz_real(self, side, evan, abs_when_evanescent=False):
        em = self.eps_in * self.mu_in if side == "in" else self.eps_out * self.mu_out
        kzc = torch.sqrt(em[:, None] - self.Kx_norm_dn ** 2 - self.Ky_norm_dn ** 2)
        ev = torch.abs(torch.real(kzc) / torch.imag(kzc)) < evan
        repl = torch.abs(torch.real(kzc)) if abs_when_evanescent else torch.zeros_like(torch.real(kzc))
        kz = torch.where(ev, repl, torch.real(kzc))
        return torch.cat((kz, kz), dim=1)                                               # [B, n]

    def S_parameters(self, orders, *, direction="forward", port="transmission", polarization="xx", ref_order=[0, 0],
                     power_norm=True, evanscent=1e-3):                                  # rcwa.py:300-524
        dev = self._device
        orders = torch.as_tensor(orders, dtype=torch.int64, device=dev).reshape([-1, 2])
        if direction in _DIRS:
            direction = _DIRS[direction]
        else:
            warnings.warn("Invalid propagation direction. Set as forward.", UserWarning)
            direction = "forward"
        if port in _PORTS:
            port = _PORTS[port]
        else:
            warnings.warn("Invalid port. Set as tramsmission.", UserWarning)
            port = "transmission"
        if polarization not in ("xx", "yx", "xy", "yy", "pp", "sp", "ps", "ss"):
            warnings.warn("Invalid polarization. Set as xx.", UserWarning)
            polarization = "xx"
        ref_order = torch.as_tensor(ref_order, dtype=torch.int64, device=dev).reshape([1, 2])
        oi = self._matching_indices(orders)
        ri = self._matching_indices(ref_order)
        N = self.order_N
        k = _SBLOCK[(direction, port)]
        Sk = self.S[k]
        num_side, den_side = _KZ_SIDES[k]

        if polarization in ("xx", "yx", "xy", "yy"):
            if polarization[0] == "y":
                oi = oi + N
            if polarization[1] == "y":
                ri = ri + N
            val = Sk[:, oi, ri[0]]                                                       # [B, M]
            if power_norm:
                kzn, kzd = self._kz_real(num_side, evanscent), self._kz_real(den_side, evanscent)
                kxr = torch.cat((torch.real(self.Kx_norm_dn),) * 2, dim=1)
                kyr = torch.cat((torch.real(self.Ky_norm_dn),) * 2, dim=1)
                pn = kxr if polarization[0] == "x" else kyr                              # rcwa.py:368-375
                pd = kxr if polarization[1] == "x" else kyr
                norm = torch.sqrt((1 + (pn[:, oi] / kzn[:, oi]) ** 2) / (1 + (pd[:, ri] / kzd[:, ri]) ** 2))
                norm = norm * torch.sqrt(kzn[:, oi] / kzd[:, ri])
                val = val * norm
            val = torch.where(torch.isinf(val), torch.zeros_like(val), val)
            val = torch.where(torch.isnan(val), torch.zeros_like(val), val)
            return val.to(self._dtype)

        # ps basis                                                                         rcwa.py:410-521
        osign, rsign = {0: (1, 1), 1: (-1, 1), 2: (1, -1), 3: (-1, -1)}[k]
        em_in, em_out = self.eps_in * self.mu_in, self.eps_out * self.mu_out
        ok2 = {0: em_out, 1: em_in, 2: em_out, 3: em_in}[k]
        rk2 = {0: em_in, 1: em_in, 2: em_out, 3: em_out}[k]

        def angles(idx, k2, sign):
            kx_, ky_ = self.Kx_norm_dn[:, idx], self.Ky_norm_dn[:, idx]
            kt = torch.sqrt(kx_ ** 2 + ky_ ** 2)
            kzc = torch.sqrt(k2[:, None] - kx_ ** 2 - ky_ ** 2)
            kzs = sign * torch.abs(torch.real(kzc))
            ev = torch.abs(torch.real(kzc) / torch.imag(kzc)) < evanscent
            return torch.atan2(torch.real(kt), kzs), torch.atan2(torch.real(ky_), torch.real(kx_)), ev

        o_inc, o_azi, o_ev = angles(oi, ok2, osign)
        r_inc, r_azi, r_ev = angles(ri, rk2, rsign)
        r0 = int(ri[0])
        xx, xy = Sk[:, oi, r0], Sk[:, oi, r0 + N]
        yx, yy = Sk[:, oi + N, r0], Sk[:, oi + N, r0 + N]
        zero = torch.zeros_like(xx)
        xx, xy, yx, yy = (torch.where(o_ev, zero, t) for t in (xx, xy, yx, yy))
        co, so, ci = torch.cos(o_azi), torch.sin(o_azi), torch.cos(o_inc)
        cr, sr, cri = torch.cos(r_azi), torch.sin(r_azi), torch.cos(r_inc)
        if polarization == "pp":
            val = co / ci * cri * cr * xx + so / ci * cri * cr * yx + co / ci * cri * sr * xy + so / ci * cri * sr * yy
        elif polarization == "ps":
            val = co / ci * (-1) * sr * xx + so / ci * (-1) * sr * yx + co / ci * cr * xy + so / ci * cr * yy
        elif polarization == "sp":
            val = -so * cri * cr * xx + co * cri * cr * yx + -so * cri * sr * xy + co * cri * sr * yy
        else:
            val = -so * (-1) * sr * xx + co * (-1) * sr * yx + -so * cr * xy + co * cr * yy
        val = torch.where(torch.isinf(val), torch.zeros_like(val), val)
        val = torch.where(torch.isnan(val), torch.zeros_like(val), val)
        if power_norm:
            kz_in = self._kz_real("in", evanscent)
            kz_out = self._kz_real("out", evanscent, abs_when_evanescent=True)          # rcwa.py:495
            kzn = kz_out if num_side == "out" else kz_in
            kzd = kz_out if den_side == "out" else kz_in
            val = val * torch.sqrt(kzn[:, oi] / kzd[:, ri])
        val = torch.where(r_ev, torch.zeros_like(val), val)                             # rcwa.py:462-464 (per point)
        return val.to(self._dtype)

    def diffraction_angle(self, orders, *, layer="output", unit="radian"):               # rcwa.py:214-262
        orders = torch.as_tensor(orders, dtype=torch.int64, device=self._device).reshape([-1, 2])
        if layer in ("i", "in", "input"):
            layer = "input"
        elif layer in ("o", "out", "output"):
            layer = "output"
        else:
            warnings.warn("Invalid layer. Set as output layer.", UserWarning)
            layer = "output"
        if unit in ("r", "rad", "radian"):
            unit = "radian"
        elif unit in ("d", "deg", "degree"):
            unit = "degree"
        else:
            warnings.warn("Invalid unit. Set as radian.", UserWarning)
            unit = "radian"
        idx = self._matching_indices(orders)
        eps = self.eps_in if layer == "input" else self.eps_out
        mu = self.mu_in if layer == "input" else self.mu_out
        kx, ky = self.Kx_norm_dn[:, idx], self.Ky_norm_dn[:, idx]
        kt = torch.sqrt(kx ** 2 + ky ** 2)
        kz = torch.sqrt((eps * mu)[:, None] - kx ** 2 - ky ** 2)
        inc = torch.atan2(torch.real(kt), torch.real(kz))
        azi = torch.atan2(torch.real(ky), torch.real(kx))
        if unit == "degree":
            inc, azi = (180. / PI_REF) * inc, (180. / PI_REF) * azi
        return inc, azi
