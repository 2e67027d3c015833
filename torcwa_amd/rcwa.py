"""`rcwa`: drop-in for torcwa.rcwa (kch3782/torcwa 0.1.4.2, torcwa/rcwa.py:7-1305) on MI355X.

Same constructor, methods, keyword spellings (`evanscent`), string aliases, warnings-with-fallback behaviour and
public attributes as the reference; the numerics run through the batched HIP path (`BatchedRCWA` with B = 1, i.e.
libtrx kernels behind include/trx.h).  Attributes are exposed un-batched and in the simulation dtype.
"""
import torch

from .batched import BatchedRCWA, PI_REF
from .fields import FieldMixin

pi = PI_REF


class rcwa(FieldMixin):
    def __init__(self, freq, order, L, *, dtype=torch.complex64, device=None, stable_eig_grad=True,
                 avoid_Pinv_instability=False, max_Pinv_instability=0.005, precision="high", engine=None):
        self._b = BatchedRCWA(freq, order, L, batch=1, dtype=dtype, device=device, stable_eig_grad=stable_eig_grad,
                              avoid_Pinv_instability=avoid_Pinv_instability, max_Pinv_instability=max_Pinv_instability,
                              precision=precision, engine=engine)
        self._dtype = self._b._dtype
        self._device = self._b._device
        self.freq = torch.as_tensor(freq, dtype=self._dtype, device=self._device)      # rcwa.py:60
        self.omega = 2 * pi * freq                                                       # rcwa.py:61 (user's freq object)
        self.order = order
        self.L = torch.as_tensor(L, dtype=self._dtype, device=self._device)                  # rcwa.py:62: a complex tensor, not the list
        self.stable_eig_grad = self._b.stable_eig_grad
        self.avoid_Pinv_instability = self._b.avoid_Pinv_instability
        self.max_Pinv_instability = self._b.max_Pinv_instability

    # ---- forwarding of the solver steps ---------------------------------------------------------------------
    def add_input_layer(self, eps=1., mu=1.):
        self._b.add_input_layer(eps, mu)
        self.Sin = []                      # existence is the reference's flag (rcwa.py:107); filled lazily below

    def add_output_layer(self, eps=1., mu=1.):
        self._b.add_output_layer(eps, mu)
        self.Sout = []

    def set_incident_angle(self, inc_ang, azi_ang, angle_layer="input"):
        self._b.set_incident_angle(inc_ang, azi_ang, angle_layer)
        self.inc_ang, self.azi_ang, self.angle_layer = self._b.inc_ang[0], self._b.azi_ang[0], self._b.angle_layer
        if hasattr(self, "Sin"):
            self.Sin = [self._u(b.dense()) for b in self._b._Sin]
        if hasattr(self, "Sout"):
            self.Sout = [self._u(b.dense()) for b in self._b._Sout]

    def add_layer(self, thickness, eps=1., mu=1.):
        def prep(v):
            if isinstance(v, (float, complex)):
                return v
            # the reference tests v.dim() directly (python ints raise AttributeError there, rcwa.py:156)
            if v.dim() == 0 or (v.dim() == 1 and v.shape[0] == 1):
                return v.reshape(1)
            return v
        self._b.add_layer(thickness, prep(eps), prep(mu))

    def solve_global_smatrix(self):
        self._b.solve_global_smatrix()

    def S_parameters(self, orders, *, direction="forward", port="transmission", polarization="xx", ref_order=[0, 0],
                     power_norm=True, evanscent=1e-3):
        return self._b.S_parameters(orders, direction=direction, port=port, polarization=polarization, ref_order=ref_order,
                                    power_norm=power_norm, evanscent=evanscent)[0]

    def diffraction_angle(self, orders, *, layer="output", unit="radian"):
        inc, azi = self._b.diffraction_angle(orders, layer=layer, unit=unit)
        return inc[0], azi[0]

    def _matching_indices(self, orders):
        return self._b._matching_indices(orders)

    def return_layer(self, layer_num, nx=100, ny=100):                                  # rcwa.py:264-298
        """eps(x,y), mu(x,y) of a layer recovered from the truncated Fourier series held in its convolution matrix.
        Harmonic (i, j), |i| <= 2ox, |j| <= 2oy, is read from the first column / first row of the Toeplitz matrix and
        placed at [i mod nx, j mod ny] (the reference's negative-index wrap); one gather + one inverse FFT per material."""
        import numpy as np
        ox, oy = self.order
        wy = 2 * oy + 1
        if 2 * ox >= nx or 2 * oy >= ny:
            raise IndexError("index %d is out of bounds for a %d x %d grid" % (2 * max(ox, oy), nx, ny))
        i, j = np.broadcast_arrays(np.arange(-2 * ox, 2 * ox + 1)[:, None], np.arange(-2 * oy, 2 * oy + 1)[None, :])
        row = np.where(i >= 0, np.where(j >= 0, i * wy + j, i * wy), np.where(j >= 0, j, 0)).ravel()
        col = np.where(i >= 0, np.where(j >= 0, 0, -j), np.where(j >= 0, -i * wy, -i * wy - j)).ravel()
        flat = ((i % nx) * ny + (j % ny)).ravel()
        # a coarse grid (nx < 4ox+1) makes harmonics collide: the reference's loop order lets the LAST write win
        _, first_rev = np.unique(flat[::-1], return_index=True)
        keep = np.sort(len(flat) - 1 - first_rev)
        dev = self._device
        row_t, col_t, flat_t = (torch.as_tensor(a[keep], dtype=torch.int64, device=dev) for a in (row, col, flat))
        outs = []
        for conv in (self._b.eps_conv[layer_num][0], self._b.mu_conv[layer_num][0]):
            f = torch.zeros(nx * ny, dtype=conv.dtype, device=dev)
            f[flat_t] = conv[row_t, col_t]
            outs.append((torch.fft.ifftn(f.reshape(nx, ny)) * nx * ny).to(self._dtype))
        return outs[0], outs[1]

    # ---- un-batched attribute views ----------------------------------------------------------------------------
    def _u(self, t):
        return None if t is None else t[0].to(self._dtype)

    def _ul(self, lst):
        return [self._u(t) for t in lst]

    @property
    def S(self):
        S = self._ul(self._b.S)
        if getattr(self._b, "_zero_layer_S", False):        # rcwa.py:187-188 stores 1-D zeros in the no-layer case
            n = 2 * self.order_N
            S[1] = torch.zeros(n, dtype=self._dtype, device=self._device)
            S[2] = torch.zeros(n, dtype=self._dtype, device=self._device)
        return S

    @property
    def C(self):
        return [self._ul(self._b.C[0]), self._ul(self._b.C[1])]

    order_N = property(lambda self: self._b.order_N)
    order_x = property(lambda self: self._b.order_x)
    order_y = property(lambda self: self._b.order_y)
    layer_N = property(lambda self: self._b.layer_N)
    thickness = property(lambda self: [t[0] for t in self._b.thickness])
    Gx_norm = property(lambda self: self._b.Gx_norm[0].to(self._dtype))
    Gy_norm = property(lambda self: self._b.Gy_norm[0].to(self._dtype))
    eps_in = property(lambda self: self._b.eps_in[0].to(self._dtype))
    mu_in = property(lambda self: self._b.mu_in[0].to(self._dtype))
    eps_out = property(lambda self: self._b.eps_out[0].to(self._dtype))
    mu_out = property(lambda self: self._b.mu_out[0].to(self._dtype))
    kx0_norm = property(lambda self: self._b.kx0_norm[0])
    ky0_norm = property(lambda self: self._b.ky0_norm[0])
    kx_norm = property(lambda self: self._u(self._b.kx_norm))
    ky_norm = property(lambda self: self._u(self._b.ky_norm))
    Kx_norm_dn = property(lambda self: self._u(self._b.Kx_norm_dn))
    Ky_norm_dn = property(lambda self: self._u(self._b.Ky_norm_dn))
    Kx_norm = property(lambda self: torch.diag(self.Kx_norm_dn))
    Ky_norm = property(lambda self: torch.diag(self.Ky_norm_dn))
    Vf = property(lambda self: self._u(self._b._Vf.dense()))
    Vi = property(lambda self: self._u(self._b._Vi.dense()))
    Vo = property(lambda self: self._u(self._b._Vo.dense()))
    eps_conv = property(lambda self: self._ul(self._b.eps_conv))
    mu_conv = property(lambda self: self._ul(self._b.mu_conv))
    P = property(lambda self: self._ul(self._b.P))
    Q = property(lambda self: self._ul(self._b.Q))
    kz_norm = property(lambda self: self._ul(self._b.kz_norm))
    E_eigvec = property(lambda self: self._ul(self._b.E_eigvec))
    H_eigvec = property(lambda self: self._ul(self._b.H_eigvec))
    layer_S11 = property(lambda self: self._ul(self._b.layer_S11))
    layer_S21 = property(lambda self: self._ul(self._b.layer_S21))
    layer_S12 = property(lambda self: self._ul(self._b.layer_S21))     # S12 == S21 for a single layer
    layer_S22 = property(lambda self: self._ul(self._b.layer_S11))     # S22 == S11
    Cf = property(lambda self: [self._u(torch.cat((p, m), dim=1)) for p, m in zip(self._b.Cplus, self._b.Cminus)])
    Cb = property(lambda self: [self._u(torch.cat((m, p), dim=1)) for p, m in zip(self._b.Cplus, self._b.Cminus)])
    Pinv_instability = property(lambda self: None if self._b.Pinv_instability is None else [t[0] for t in self._b.Pinv_instability])
    Qinv_instability = property(lambda self: None if self._b.Qinv_instability is None else [t[0] for t in self._b.Qinv_instability])
