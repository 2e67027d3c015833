"""CPU oracle for the RCWA layer-solve hot path (TEST INFRASTRUCTURE, not product code).

This is a restatement, in a functional style and on torch-CPU, of the algorithm of
kch3782/torcwa 0.1.4.2 (`torcwa/rcwa.py`, `torcwa/torch_eig.py`).  It keeps the *same
operation sequence* as the reference (1 eig / 12 inv / 48 matmul / 1 fft2 per patterned
layer-solve with an input half-space) so that it can also serve as the timed
"reference CPU path" (`bench.py` cpu_baseline, kind="port").

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
this module.  The product (`torcwa_amd`) never does.

Parity status: PINNED.  `tests/golden/make_golden.py` imports the real reference from
/root/reference in the build container and stores its outputs in `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks this oracle against them (<=1e-12 rel in complex128).

Every function cites the reference lines it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import torch

# torcwa/rcwa.py:5 -- the reference's value of pi has a typo in the 9th decimal
# (3.141592652589793 instead of 3.141592653589793).  Parity needs the same constant.
PI_REF = 3.141592652589793


def _c(x, dtype):
    return torch.as_tensor(x, dtype=dtype)


# ----------------------------------------------------------------------------------------
# a1: simulation constants                                   (torcwa/rcwa.py:9-93)
# ----------------------------------------------------------------------------------------
@dataclass
class Setup:
    freq: object                      # user's freq object (python float or real 0-d tensor)
    order: Sequence[int]
    L: Sequence[float]
    dtype: torch.dtype = torch.complex128
    eps_in: object = 1.0
    mu_in: object = 1.0
    eps_out: object = 1.0
    mu_out: object = 1.0
    has_in: bool = False              # add_input_layer called  (rcwa.py:107 creates self.Sin)
    has_out: bool = False             # add_output_layer called (rcwa.py:121 creates self.Sout)
    inc_ang: object = 0.0
    azi_ang: object = 0.0
    angle_layer: str = "input"
    # filled by kvectors()
    kx: Optional[torch.Tensor] = None
    ky: Optional[torch.Tensor] = None
    Vf: Optional[torch.Tensor] = None
    Vi: Optional[torch.Tensor] = None
    Vo: Optional[torch.Tensor] = None
    Sin: Optional[List[torch.Tensor]] = None
    Sout: Optional[List[torch.Tensor]] = None

    @property
    def N(self):
        return (2 * self.order[0] + 1) * (2 * self.order[1] + 1)

    @property
    def omega(self):
        # rcwa.py:61 -- built from the user's freq object and the typo'd pi
        return 2 * PI_REF * self.freq

    @property
    def mx(self):
        return torch.arange(-self.order[0], self.order[0] + 1, dtype=torch.int64)   # rcwa.py:66

    @property
    def my(self):
        return torch.arange(-self.order[1], self.order[1] + 1, dtype=torch.int64)   # rcwa.py:67


def _halfspace_V(kx, ky, epsmu):
    """E->H map of a homogeneous half-space with eps*mu = epsmu (rcwa.py:1143-1147, 1151-1155)."""
    kz = torch.sqrt(epsmu - kx ** 2 - ky ** 2)
    kz = torch.where(torch.imag(kz) < 0, torch.conj(kz), kz)
    top = torch.hstack((torch.diag(-ky * kx / kz), torch.diag(-kz - ky ** 2 / kz)))
    bot = torch.hstack((torch.diag(kz + kx ** 2 / kz), torch.diag(kx * ky / kz)))
    # reference builds column blocks tmp1=[a;b], tmp2=[c;d] and hstacks them: [[a,c],[b,d]]
    return torch.vstack((top, bot))


def kvectors(s: Setup) -> Setup:
    """a3: set_incident_angle -> _kvectors (rcwa.py:123-144, 1124-1181)."""
    dt = s.dtype
    freq_c = _c(s.freq, dt)
    Gx = 1 / (s.L[0] * freq_c)                                     # rcwa.py:72
    Gy = 1 / (s.L[1] * freq_c)
    inc = _c(s.inc_ang, dt)
    azi = _c(s.azi_ang, dt)
    if s.angle_layer == "input":
        nref = torch.real(torch.sqrt(_c(s.eps_in, dt) * _c(s.mu_in, dt)))           # :1126
    else:
        nref = torch.real(torch.sqrt(_c(s.eps_out, dt) * _c(s.mu_out, dt)))         # :1129
    kx0 = nref * torch.sin(inc) * torch.cos(azi)
    ky0 = nref * torch.sin(inc) * torch.sin(azi)
    kxv = kx0 + s.mx * Gx                                          # :1133
    kyv = ky0 + s.my * Gy
    gx, gy = torch.meshgrid(kxv, kyv, indexing="ij")               # :1136
    s.kx = gx.reshape(-1)
    s.ky = gy.reshape(-1)
    s.Vf = _halfspace_V(s.kx, s.ky, 1.0)                           # :1143-1147
    if s.has_in:                                                   # :1149-1164
        s.Vi = _halfspace_V(s.kx, s.ky, _c(s.eps_in, dt) * _c(s.mu_in, dt))
        T = torch.linalg.inv(s.Vf + s.Vi)
        D = s.Vf - s.Vi
        s.Sin = [2 * (T @ s.Vi), -(T @ D), T @ D, 2 * (T @ s.Vf)]
    if s.has_out:                                                  # :1166-1181
        s.Vo = _halfspace_V(s.kx, s.ky, _c(s.eps_out, dt) * _c(s.mu_out, dt))
        T = torch.linalg.inv(s.Vf + s.Vo)
        D = s.Vf - s.Vo
        s.Sout = [2 * (T @ s.Vf), T @ D, -(T @ D), 2 * (T @ s.Vo)]
    return s


# ----------------------------------------------------------------------------------------
# a5: Fourier factorisation (Laurent rule)                    (torcwa/rcwa.py:1183-1204)
# ----------------------------------------------------------------------------------------
def conv_matrix(grid: torch.Tensor, order) -> torch.Tensor:
    """E[i,j] = fft2(grid)[(m_i-m_j) mod nx, (n_i-n_j) mod ny]/(nx*ny), i=(m+ox)(2oy+1)+(n+oy)."""
    nx, ny = grid.shape
    mx = torch.arange(-order[0], order[0] + 1, dtype=torch.int64)
    my = torch.arange(-order[1], order[1] + 1, dtype=torch.int64)
    gm, gn = torch.meshgrid(mx, my, indexing="ij")
    m = gm.reshape(-1)
    n = gn.reshape(-1)
    coef = torch.fft.fft2(grid) / (nx * ny)                        # :1194
    dm = m[:, None] - m[None, :]                                   # python negative index == mod
    dn = n[:, None] - n[None, :]
    re = torch.real(coef)[dm, dn]                                  # :1199
    im = torch.imag(coef)[dm, dn]                                  # :1200
    return torch.complex(re, im)


def is_homogeneous(v) -> bool:
    """rcwa.py:156-157 (python int is rejected there with AttributeError; mirrored)."""
    return (type(v) == float) or (type(v) == complex) or (v.dim() == 0) or (v.dim() == 1 and v.shape[0] == 1)


# ----------------------------------------------------------------------------------------
# a6: layer eigenproblem                                      (torcwa/rcwa.py:1206-1242)
# ----------------------------------------------------------------------------------------
def pq_patterned(E, M, kx, ky):
    """P, Q of a patterned layer (rcwa.py:1226-1232)."""
    Kx, Ky = torch.diag(kx), torch.diag(ky)
    KK = torch.vstack((Kx, Ky))
    Z = torch.zeros_like(M)
    P = torch.vstack((torch.hstack((Z, M)), torch.hstack((-M, Z)))) + (KK @ torch.linalg.inv(E)) @ torch.hstack((Ky, -Kx))
    Z = torch.zeros_like(E)
    Q = torch.vstack((torch.hstack((Z, -E)), torch.hstack((E, Z)))) + (KK @ torch.linalg.inv(M)) @ torch.hstack((-Ky, Kx))
    return P, Q


def pq_homogeneous(E, M, eps, mu, kx, ky):
    """P, Q of a homogeneous layer (rcwa.py:1208-1214)."""
    Kx, Ky = torch.diag(kx), torch.diag(ky)
    KK = torch.vstack((Kx, Ky))
    Z = torch.zeros_like(M)
    P = torch.vstack((torch.hstack((Z, M)), torch.hstack((-M, Z)))) + 1 / eps * (KK @ torch.hstack((Ky, -Kx)))
    Q = torch.vstack((torch.hstack((Z, -E)), torch.hstack((E, Z)))) + 1 / mu * (KK @ torch.hstack((-Ky, Kx)))
    return P, Q


def modes_patterned(P, Q):
    """(kz, W) = eig(P Q), kz = sqrt(lambda) flipped to Im>=0 (rcwa.py:1235-1242, torch_eig.py:14)."""
    lam, W = torch.linalg.eig(P @ Q)
    kz = torch.sqrt(lam)
    kz = torch.where(torch.imag(kz) < 0, -kz, kz)                  # :1241 (note: -kz, not conj)
    return kz, W


def modes_homogeneous(eps, mu, kx, ky, dtype):
    """W = I, analytic kz (rcwa.py:1216-1219)."""
    n = 2 * kx.shape[0]
    W = torch.eye(n, dtype=dtype)
    kz = torch.sqrt(eps * mu - kx ** 2 - ky ** 2)
    kz = torch.where(torch.imag(kz) < 0, torch.conj(kz), kz)       # :1218 (conj here)
    return torch.cat((kz, kz)), W


# ----------------------------------------------------------------------------------------
# a8: layer S-matrix                                          (torcwa/rcwa.py:1244-1281)
# ----------------------------------------------------------------------------------------
@dataclass
class Layer:
    thickness: object
    E: torch.Tensor
    M: torch.Tensor
    P: torch.Tensor
    Q: torch.Tensor
    kz: torch.Tensor
    W: torch.Tensor
    V: Optional[torch.Tensor] = None
    Cf: Optional[torch.Tensor] = None
    Cb: Optional[torch.Tensor] = None
    S: Optional[List[torch.Tensor]] = None
    Pinv_instability: Optional[torch.Tensor] = None
    Qinv_instability: Optional[torch.Tensor] = None


def layer_smatrix(lay: Layer, Vf, omega, *, avoid_Pinv_instability=False, max_Pinv_instability=0.005):
    dt = lay.W.dtype
    n = lay.W.shape[0]
    I = torch.eye(n, dtype=dt)
    Kz = torch.diag(lay.kz)
    X = torch.diag(torch.exp(1.j * omega * lay.kz * lay.thickness))            # :1246
    Pinv = torch.linalg.inv(lay.P)                                              # :1248
    if avoid_Pinv_instability:                                                  # :1249-1262
        a = torch.max(torch.abs(lay.P @ Pinv - I))
        b = torch.max(torch.abs(Pinv @ lay.P - I))
        c = torch.max(torch.abs(lay.Q @ torch.linalg.inv(lay.Q) - I))           # computed twice in the ref
        lay.Pinv_instability = torch.maximum(a, b)
        lay.Qinv_instability = torch.maximum(c, c)
        if lay.Pinv_instability < max_Pinv_instability:
            lay.V = Pinv @ (lay.W @ Kz)
        else:
            lay.V = lay.Q @ (lay.W @ torch.linalg.inv(Kz))
    else:
        lay.V = Pinv @ (lay.W @ Kz)                                             # :1264
    # The reference evaluates inv(Vf) four times (:1266-1267); kept for op-count fidelity.
    A1 = lay.W + torch.linalg.inv(Vf) @ lay.V
    B1 = (lay.W - torch.linalg.inv(Vf) @ lay.V) @ X
    B2 = (lay.W - torch.linalg.inv(Vf) @ lay.V) @ X
    A2 = lay.W + torch.linalg.inv(Vf) @ lay.V
    big = torch.hstack((torch.vstack((A1, B1)), torch.vstack((B2, A2))))       # :1268
    Z = torch.zeros((n, n), dtype=dt)
    lay.Cf = torch.linalg.inv(big) @ torch.vstack((2 * I, Z))                   # :1271
    lay.Cb = torch.linalg.inv(big) @ torch.vstack((Z, 2 * I))                   # :1273
    WX = lay.W @ X
    S11 = WX @ lay.Cf[:n] + lay.W @ lay.Cf[n:]                                  # :1276
    S21 = lay.W @ lay.Cf[:n] + (lay.W @ X) @ lay.Cf[n:] - I                     # :1277
    S12 = (lay.W @ X) @ lay.Cb[:n] + lay.W @ lay.Cb[n:] - I                     # :1279
    S22 = lay.W @ lay.Cb[:n] + (lay.W @ X) @ lay.Cb[n:]                         # :1281
    lay.S = [S11, S21, S12, S22]
    return lay


def add_layer(s: Setup, thickness, eps=1.0, mu=1.0, **kw) -> Layer:
    """a4: add_layer orchestration (rcwa.py:146-170)."""
    dt = s.dtype
    N = s.N
    he, hm = is_homogeneous(eps), is_homogeneous(mu)
    E = eps * torch.eye(N, dtype=dt) if he else conv_matrix(eps, s.order)      # :159
    M = mu * torch.eye(N, dtype=dt) if hm else conv_matrix(mu, s.order)        # :160
    if he and hm:
        P, Q = pq_homogeneous(E, M, eps, mu, s.kx, s.ky)
        kz, W = modes_homogeneous(eps, mu, s.kx, s.ky, dt)
    else:
        P, Q = pq_patterned(E, M, s.kx, s.ky)
        kz, W = modes_patterned(P, Q)
    lay = Layer(thickness=thickness, E=E, M=M, P=P, Q=Q, kz=kz, W=W)
    return layer_smatrix(lay, s.Vf, s.omega, **kw)


# ----------------------------------------------------------------------------------------
# a9/a10: Redheffer star product and global assembly          (torcwa/rcwa.py:173-211, 1283-1306)
# ----------------------------------------------------------------------------------------
def redheffer(Sm, Sn, Cm, Cn):
    n = Sm[0].shape[0]
    I = torch.eye(n, dtype=Sm[0].dtype)
    t1 = torch.linalg.inv(I - Sm[2] @ Sn[1])                                    # :1287
    t2 = torch.linalg.inv(I - Sn[1] @ Sm[2])                                    # :1288
    S11 = Sn[0] @ (t1 @ Sm[0])
    S21 = Sm[1] + Sm[3] @ (t2 @ (Sn[1] @ Sm[0]))
    S12 = Sn[2] + Sn[0] @ (t1 @ (Sm[2] @ Sn[3]))
    S22 = Sm[3] @ (t2 @ Sn[3])
    C = [[], []]
    for k in range(len(Cm[0])):                                                 # :1298-1300
        C[0].append(Cm[0][k] + Cm[1][k] @ (t2 @ (Sn[1] @ Sm[0])))
        C[1].append(Cm[1][k] @ (t2 @ Sn[3]))
    for k in range(len(Cn[0])):                                                 # :1302-1304
        C[0].append(Cn[0][k] @ (t1 @ Sm[0]))
        C[1].append(Cn[1][k] + Cn[0][k] @ (t1 @ (Sm[2] @ Sn[3])))
    return [S11, S21, S12, S22], C


def global_smatrix(s: Setup, layers: List[Layer]):
    dt = s.dtype
    n = 2 * s.N
    if layers:
        S = list(layers[0].S)
        C = [[layers[0].Cf], [layers[0].Cb]]
    else:                                                                       # :185-190 (1-D zeros)
        S = [torch.eye(n, dtype=dt), torch.zeros(n, dtype=dt), torch.zeros(n, dtype=dt), torch.eye(n, dtype=dt)]
        C = [[], []]
    for lay in layers[1:]:
        S, C = redheffer(S, lay.S, C, [[lay.Cf], [lay.Cb]])
    if s.has_in:
        S, C = redheffer(s.Sin, S, [[], []], C)
    if s.has_out:
        S, C = redheffer(S, s.Sout, C, [[], []])
    return S, C


# ----------------------------------------------------------------------------------------
# a11: read-out                                               (torcwa/rcwa.py:300-524, 1115-1122)
# ----------------------------------------------------------------------------------------
def matching_indices(orders: torch.Tensor, order) -> torch.Tensor:
    """Clamp (in place, like the reference) and map (m,n) -> (2oy+1)(m+ox)+(n+oy) (rcwa.py:1115-1122)."""
    ox, oy = int(order[0]), int(order[1])
    orders[orders[:, 0] < -ox, 0] = -ox
    orders[orders[:, 0] > ox, 0] = ox
    orders[orders[:, 1] < -oy, 1] = -oy
    orders[orders[:, 1] > oy, 1] = oy
    return (2 * oy + 1) * (orders[:, 0] + ox) + orders[:, 1] + oy


_DIR = {"f": "forward", "forward": "forward", "b": "backward", "backward": "backward"}
_PORT = {"t": "transmission", "transmission": "transmission", "r": "reflection", "reflection": "reflection"}
_SIDX = {("forward", "transmission"): 0, ("forward", "reflection"): 1,
         ("backward", "reflection"): 2, ("backward", "transmission"): 3}


def _kz_real_side(s: Setup, which: str, evan, *, abs_when_evanescent=False):
    dt = s.dtype
    em = _c(s.eps_in, dt) * _c(s.mu_in, dt) if which == "in" else _c(s.eps_out, dt) * _c(s.mu_out, dt)
    kzc = torch.sqrt(em - s.kx ** 2 - s.ky ** 2)
    ev = torch.abs(torch.real(kzc) / torch.imag(kzc)) < evan
    repl = torch.abs(torch.real(kzc)) if abs_when_evanescent else torch.real(torch.zeros_like(kzc))
    kz = torch.where(ev, repl, torch.real(kzc))
    return torch.hstack((kz, kz))


def s_parameters(s: Setup, S, orders, *, direction="forward", port="transmission", polarization="xx",
                 ref_order=(0, 0), power_norm=True, evanscent=1e-3):
    dt = s.dtype
    N = s.N
    orders = torch.as_tensor(orders, dtype=torch.int64).reshape(-1, 2).clone()
    ref = torch.as_tensor(ref_order, dtype=torch.int64).reshape(1, 2).clone()
    direction = _DIR.get(direction, "forward")
    port = _PORT.get(port, "transmission")
    if polarization not in ("xx", "yx", "xy", "yy", "pp", "sp", "ps", "ss"):
        polarization = "xx"
    oi = matching_indices(orders, s.order)
    ri = matching_indices(ref, s.order)
    k = _SIDX[(direction, port)]
    num_side, den_side = {0: ("out", "in"), 1: ("in", "in"), 2: ("out", "out"), 3: ("in", "out")}[k]

    if polarization in ("xx", "yx", "xy", "yy"):                                # :346-408
        if polarization[0] == "y":
            oi = oi + N
        if polarization[1] == "y":
            ri = ri + N
        if power_norm:
            kzn = _kz_real_side(s, num_side, evanscent)
            kzd = _kz_real_side(s, den_side, evanscent)
            kxr = torch.hstack((torch.real(s.kx), torch.real(s.kx)))
            kyr = torch.hstack((torch.real(s.ky), torch.real(s.ky)))
            pn = kxr if polarization[0] == "x" else kyr
            pd = kxr if polarization[1] == "x" else kyr
            # note: the reference picks numerator by *output* letter?  rcwa.py:368-375:
            #  'xx': (Kx,Kx)  'xy': (Kx,Ky)  'yx': (Ky,Kx)  'yy': (Ky,Ky)  -> (first letter, second letter)
            norm = torch.sqrt((1 + (pn[oi] / kzn[oi]) ** 2) / (1 + (pd[ri] / kzd[ri]) ** 2))
            norm = norm * torch.sqrt(kzn[oi] / kzd[ri])
        else:
            norm = 1.0
        out = S[k][oi, ri] * norm
        out = torch.where(torch.isinf(out), torch.zeros_like(out), out)
        out = torch.where(torch.isnan(out), torch.zeros_like(out), out)
        return out

    # ps basis                                                                   :410-521
    osign, rsign = {0: (1, 1), 1: (-1, 1), 2: (1, -1), 3: (-1, -1)}[k]
    em_in = _c(s.eps_in, dt) * _c(s.mu_in, dt)
    em_out = _c(s.eps_out, dt) * _c(s.mu_out, dt)
    ok2 = {0: em_out, 1: em_in, 2: em_out, 3: em_in}[k]
    rk2 = {0: em_in, 1: em_in, 2: em_out, 3: em_out}[k]

    def angles(idx, k2, sign):
        kx_, ky_ = s.kx[idx], s.ky[idx]
        kt = torch.sqrt(kx_ ** 2 + ky_ ** 2)
        kzc = torch.sqrt(k2 - kx_ ** 2 - ky_ ** 2)
        kzs = sign * torch.abs(torch.real(kzc))
        ev = torch.abs(torch.real(kzc) / torch.imag(kzc)) < evanscent
        return torch.atan2(torch.real(kt), kzs), torch.atan2(torch.real(ky_), torch.real(kx_)), ev

    o_inc, o_azi, o_ev = angles(oi, ok2, osign)
    r_inc, r_azi, r_ev = angles(ri, rk2, rsign)
    xx = S[k][oi, ri]
    xy = S[k][oi, ri + N]
    yx = S[k][oi + N, ri]
    yy = S[k][oi + N, ri + N]
    zero = torch.zeros_like(xx)
    xx, xy, yx, yy = (torch.where(o_ev, zero, t) for t in (xx, xy, yx, yy))
    if bool(r_ev):
        return torch.zeros_like(xx)
    co, so, ci = torch.cos(o_azi), torch.sin(o_azi), torch.cos(o_inc)
    cr, sr, cri = torch.cos(r_azi), torch.sin(r_azi), torch.cos(r_inc)
    if polarization == "pp":
        out = co / ci * cri * cr * xx + so / ci * cri * cr * yx + co / ci * cri * sr * xy + so / ci * cri * sr * yy
    elif polarization == "ps":
        out = co / ci * (-1) * sr * xx + so / ci * (-1) * sr * yx + co / ci * cr * xy + so / ci * cr * yy
    elif polarization == "sp":
        out = -so * cri * cr * xx + co * cri * cr * yx + -so * cri * sr * xy + co * cri * sr * yy
    else:  # ss
        out = -so * (-1) * sr * xx + co * (-1) * sr * yx + -so * cr * xy + co * cr * yy
    if power_norm:
        kz_in = _kz_real_side(s, "in", evanscent)
        kz_out = _kz_real_side(s, "out", evanscent, abs_when_evanescent=True)   # :495 quirk
        kzn = kz_out if num_side == "out" else kz_in
        kzd = kz_out if den_side == "out" else kz_in
        norm = torch.sqrt(kzn[oi] / kzd[ri])
    else:
        norm = 1.0
    out = torch.where(torch.isinf(out), torch.zeros_like(out), out)
    out = torch.where(torch.isnan(out), torch.zeros_like(out), out)
    return out * norm


# ----------------------------------------------------------------------------------------
# convenience drivers used by tests and by bench.py's cpu_baseline
# ----------------------------------------------------------------------------------------
def rectangle_density(nx, ny, Lx, Ly, Wx, Wy, Cx, Cy, theta=0.0, edge_sharpness=1000.0, dtype=torch.float64):
    """Sigmoid level-set rectangle on the cell-centre grid (torcwa/geometry.py:39-46, 87-100)."""
    x = (Lx / nx) * (torch.arange(nx, dtype=dtype) + 0.5)
    y = (Ly / ny) * (torch.arange(ny, dtype=dtype) + 0.5)
    xg, yg = torch.meshgrid(x, y, indexing="ij")
    th = torch.as_tensor(theta, dtype=dtype)
    u = ((xg - Cx) * torch.cos(th) + (yg - Cy) * torch.sin(th)) / (Wx / 2.0)
    v = (-(xg - Cx) * torch.sin(th) + (yg - Cy) * torch.cos(th)) / (Wy / 2.0)
    return torch.sigmoid(edge_sharpness * (1.0 - torch.maximum(torch.abs(u), torch.abs(v))))


def solve_stack(freq, order, L, layers, *, dtype=torch.complex128, eps_in=None, eps_out=None,
                inc_ang=0.0, azi_ang=0.0, angle_layer="input", **kw):
    """One sweep point: layers = [(thickness, eps[, mu]), ...].  Returns (Setup, [Layer], S, C)."""
    s = Setup(freq=freq, order=order, L=L, dtype=dtype)
    if eps_in is not None:
        s.eps_in, s.has_in = eps_in, True
    if eps_out is not None:
        s.eps_out, s.has_out = eps_out, True
    s.inc_ang, s.azi_ang, s.angle_layer = inc_ang, azi_ang, angle_layer
    kvectors(s)
    lays = [add_layer(s, *l, **kw) for l in layers]
    S, C = global_smatrix(s, lays)
    return s, lays, S, C
