"""Sources and field maps of the drop-in `rcwa` (torcwa/rcwa.py:526-1112), re-expressed as batched GEMMs.

The reference evaluates fields with a Python loop over z samples and ~30 dense n x n products per sample
(rcwa.py:637-764).  The same quantities are, per layer,

    Exy(z) = W ( c+ . e^{i w kz z} + c- . e^{i w kz (d - z)} ),   Hxy(z) = V ( c+ . e^{...} - c- . e^{...} ),
    Ez = E^-1 (Ky Hx - Kx Hy),   Hz = M^-1 (Kx Ey - Ky Ex),       [c+; c-] = C_layer E_i,

so all z samples of a layer are ONE [n,n] x [n,nz] product each (libtrx GEMM), and the spatial synthesis
F(x,z) = sum_mn F_mn(z) e^{i w (kx x + ky y)} is one more [nx,N] x [N,nz] product.  Results are identical to the
reference's (same formulas, rcwa.py lines cited below); eigen-mode ordering does not matter because the sums run over
all modes.
"""
import warnings

import torch


class FieldMixin:
    # ---- sources ---------------------------------------------------------------------------------------------
    def source_planewave(self, *, amplitude=[1., 0.], direction="forward", notation="xy"):          # rcwa.py:526-537
        self.source_fourier(amplitude=amplitude, orders=[0, 0], direction=direction, notation=notation)

    def source_fourier(self, *, amplitude, orders, direction="forward", notation="xy"):             # rcwa.py:539-596
        b = self._b
        cdt, dev, N = b._cdtype, self._device, b.order_N
        amplitude = torch.as_tensor(amplitude, dtype=cdt, device=dev).reshape([-1, 2])
        orders = torch.as_tensor(orders, dtype=torch.int64, device=dev).reshape([-1, 2])
        if direction in ("f", "forward"):
            direction = "forward"
        elif direction in ("b", "backward"):
            direction = "backward"
        else:
            warnings.warn("Invalid source direction. Set as forward.", UserWarning)
            direction = "forward"
        if notation not in ("xy", "ps"):
            warnings.warn("Invalid amplitude notation. Set as xy notation.", UserWarning)
            notation = "xy"
        idx = b._matching_indices(orders)
        self.source_direction = direction
        E_i = torch.zeros([2 * N], dtype=cdt, device=dev)
        E_i[idx] = amplitude[:, 0]
        E_i[idx + N] = amplitude[:, 1]
        if notation == "ps":                                                                       # rcwa.py:575-594
            eps, mu, sign = (b.eps_in[0], b.mu_in[0], 1) if direction == "forward" else (b.eps_out[0], b.mu_out[0], -1)
            kx, ky = b.Kx_norm_dn[0], b.Ky_norm_dn[0]
            kt = torch.sqrt(kx ** 2 + ky ** 2)
            kz = sign * torch.abs(torch.real(torch.sqrt(eps * mu - kx ** 2 - ky ** 2)))
            inc = torch.atan2(torch.real(kt), kz)
            azi = torch.atan2(torch.real(ky), torch.real(kx))
            Ep, Es = E_i[:N], E_i[N:]
            Ex = torch.cos(inc) * torch.cos(azi) * Ep - torch.sin(azi) * Es
            Ey = torch.cos(inc) * torch.sin(azi) * Ep + torch.cos(azi) * Es
            E_i = torch.cat((Ex, Ey)).to(cdt)
        self._E_i = E_i.reshape(-1, 1)

    @property
    def E_i(self):
        return self._E_i.to(self._dtype)

    # ---- helpers ---------------------------------------------------------------------------------------------
    def _mm(self, A, B):
        """[r,k] @ [k,c] on the HIP GEMM; through the autograd wrapper when the simulation was built on the differentiable
        path, so that field-based figures of merit can be back-propagated like in the reference (plain torch ops there)."""
        if getattr(self._b, "_diff", False) and (A.requires_grad or B.requires_grad):
            from . import autograd_ops as ag
            return ag.GemmFn.apply(A[None].contiguous(), B[None].contiguous(), self._b.engine)[0]
        return self._b.engine.gemm(A[None].contiguous(), B[None].contiguous())[0]

    def _bd_apply(self, bd, X):
        """(2x2-block-diagonal operator) @ X for X [n, c]."""
        N = self._b.order_N
        d0, d1, d2, d3 = [t[0][:, None] for t in bd.d]
        return torch.cat((d0 * X[:N] + d1 * X[N:], d2 * X[:N] + d3 * X[N:]), dim=0)

    def _layer_of(self, z_axis):
        """Layer index of every z sample (rcwa.py:623-634): -1 input, 0..L-1 internal, L output."""
        b = self._b
        dev = self._device
        th = [float(t[0]) for t in b.thickness]
        zp = torch.zeros(len(th), device=dev, dtype=z_axis.dtype)
        for i in range(len(th)):
            zp[i:] += th[i]
        zm = torch.zeros_like(zp)
        if len(th) > 0:
            zm[1:] = zp[:-1]
        layer = torch.zeros(len(z_axis), dtype=torch.int64, device=dev)
        layer[z_axis < 0.] = -1
        for i in range(len(zp)):
            layer[z_axis > zp[i]] += 1
        return layer, zp, zm

    def _coeffs(self, layer_num, zprop):
        """Fourier coefficients (Ex,Ey,Ez,Hx,Hy,Hz), each [N, nz], of layer `layer_num` at in-layer offsets zprop [nz]."""
        b = self._b
        cdt, N = b._cdtype, b.order_N
        n = 2 * N
        om = b.omega[0]
        kx, ky = b.Kx_norm_dn[0], b.Ky_norm_dn[0]
        E_i = self._E_i
        zprop = zprop.to(b._rdtype).reshape(1, -1)
        fwd = self.source_direction == "forward"
        if layer_num == -1 or layer_num == b.layer_N:                                            # rcwa.py:639-696
            if layer_num == -1:
                eps, mu = b.eps_in[0], b.mu_in[0]
                Vh = b._Vi if b.has_in else b._Vf
                kz = torch.sqrt(eps * mu - kx ** 2 - ky ** 2)
                kz = torch.where(torch.imag(kz) > 0, torch.conj(kz), kz)
            else:
                eps, mu = b.eps_out[0], b.mu_out[0]
                Vh = b._Vo if b.has_out else b._Vf
                kz = torch.sqrt(eps * mu - kx ** 2 - ky ** 2)
                kz = torch.where(torch.imag(kz) < 0, torch.conj(kz), kz)
            kz2 = torch.cat((kz, kz)).reshape(-1, 1)
            ph = torch.exp(1j * om * kz2 * zprop)                                                 # [n, nz]
            zero = torch.zeros((n, zprop.shape[1]), dtype=cdt, device=self._device)
            S = b.S
            if layer_num == -1 and fwd:
                Ep = E_i * ph
                Em = self._mm(S[1][0], E_i) * torch.conj(ph)
                Hp, Hm = self._bd_apply(Vh, Ep), -self._bd_apply(Vh, Em)
            elif layer_num == -1:
                Ep, Hp = zero, zero
                Em = self._mm(S[3][0], E_i) * torch.conj(ph)
                Hm = -self._bd_apply(Vh, Em)
            elif fwd:
                Ep = self._mm(S[0][0], E_i) * ph
                Hp = self._bd_apply(Vh, Ep)
                Em, Hm = zero, zero
            else:
                Ep = self._mm(S[2][0], E_i) * ph
                Hp = self._bd_apply(Vh, Ep)
                Em = E_i * torch.conj(ph)
                Hm = -self._bd_apply(Vh, Em)
            Exy, Hxy = Ep + Em, Hp + Hm
            Ex, Ey, Hx, Hy = Exy[:N], Exy[N:], Hxy[:N], Hxy[N:]
            Hz = (kx[:, None] * Ey - ky[:, None] * Ex) / mu
            Ez = (ky[:, None] * Hx - kx[:, None] * Hy) / eps
            return Ex, Ey, Ez, Hx, Hy, Hz
        # internal layer                                                                           rcwa.py:708-755
        Cl = b.C[0][layer_num][0] if fwd else b.C[1][layer_num][0]                                # [2n, n]
        C = self._mm(Cl, E_i)                                                                     # [2n, 1]
        cp, cm = C[:n], C[n:]
        kzl = b.kz_norm[layer_num][0].reshape(-1, 1)
        d = b.thickness[layer_num][0]
        Pp = cp * torch.exp(1j * om * kzl * zprop)
        Pm = cm * torch.exp(1j * om * kzl * (d - zprop))
        Exy = self._mm(b.E_eigvec[layer_num][0], Pp + Pm)
        Hxy = self._mm(b.H_eigvec[layer_num][0], Pp - Pm)
        Ex, Ey, Hx, Hy = Exy[:N], Exy[N:], Hxy[:N], Hxy[N:]
        inv = self._conv_inverses(layer_num)
        Hz = self._mm(inv[1], kx[:, None] * Ey - ky[:, None] * Ex)
        Ez = self._mm(inv[0], ky[:, None] * Hx - kx[:, None] * Hy)
        return Ex, Ey, Ez, Hx, Hy, Hz

    def _conv_inverses(self, layer_num):
        cache = self.__dict__.setdefault("_conv_inv_cache", {})
        key = (layer_num, id(self._b.eps_conv[layer_num]))
        if key not in cache:
            eng = self._b.engine
            E, M = self._b.eps_conv[layer_num], self._b.mu_conv[layer_num]
            if getattr(self._b, "_diff", False) and (E.requires_grad or M.requires_grad):
                from . import autograd_ops as ag
                return ag.InverseFn.apply(E, eng)[0], ag.InverseFn.apply(M, eng)[0]      # graph-bound: not cached
            cache[key] = (eng.inverse(E)[0], eng.inverse(M)[0])
        return cache[key]

    def _plane(self, which, axis, z_axis, fixed):
        """field_xz (which='x': axis=x samples, fixed=y) / field_yz (which='y')."""
        b = self._b
        cdt = b._cdtype
        om = b.omega[0]
        kx, ky = b.Kx_norm_dn[0], b.Ky_norm_dn[0]
        axis = axis.to(self._device).reshape(-1, 1).to(b._rdtype)
        z_axis = z_axis.to(self._device).reshape(-1)
        layer, zp, zm = self._layer_of(z_axis)
        if which == "x":
            phase = torch.exp(1j * om * (kx[None, :] * axis + ky[None, :] * fixed)).to(cdt)       # [nx, N]
        else:
            phase = torch.exp(1j * om * (kx[None, :] * fixed + ky[None, :] * axis)).to(cdt)       # [ny, N]
        out = [torch.zeros((axis.shape[0], z_axis.shape[0]), dtype=cdt, device=self._device) for _ in range(6)]
        for ln in sorted(set(layer.tolist())):
            sel = torch.nonzero(layer == ln).reshape(-1)
            z = z_axis[sel]
            if ln == -1:
                zprop = torch.where(z <= 0., z, torch.zeros_like(z))
            elif ln == b.layer_N:
                zprop = z if len(zp) == 0 else torch.where(z - zp[-1] >= 0., z - zp[-1], torch.zeros_like(z))
            else:
                zprop = z - zm[ln]
            coeffs = self._coeffs(ln, zprop)
            for k in range(6):
                out[k][:, sel] = self._mm(phase, coeffs[k].to(cdt))
        out = [t.to(self._dtype) for t in out]
        return out[:3], out[3:]

    # ---- public API ------------------------------------------------------------------------------------------
    def field_xz(self, x_axis, z_axis, y):                                                        # rcwa.py:598-775
        if type(x_axis) != torch.Tensor or type(z_axis) != torch.Tensor:
            warnings.warn("x and z axis must be torch.Tensor type. Return None.", UserWarning)
            return None
        return self._plane("x", x_axis, z_axis, y)

    def field_yz(self, y_axis, z_axis, x):                                                        # rcwa.py:777-957
        if type(y_axis) != torch.Tensor or type(z_axis) != torch.Tensor:
            warnings.warn("y and z axis must be torch.Tensor type. Return None.", UserWarning)
            return None
        return self._plane("y", y_axis, z_axis, x)

    def field_xy(self, layer_num, x_axis, y_axis, z_prop=0.):                                     # rcwa.py:959-1112
        b = self._b
        if type(layer_num) != int:
            warnings.warn('Parameter "layer_num" must be int type. Return None.', UserWarning)
            return None
        if layer_num < -1 or layer_num > b.layer_N:
            warnings.warn("Layer number is out of range. Return None.", UserWarning)
            return None
        if type(x_axis) != torch.Tensor or type(y_axis) != torch.Tensor:
            warnings.warn("x and y axis must be torch.Tensor type. Return None.", UserWarning)
            return None
        cdt = b._cdtype
        om = b.omega[0]
        kx, ky = b.Kx_norm_dn[0], b.Ky_norm_dn[0]
        x = x_axis.to(self._device).reshape(-1, 1).to(b._rdtype)
        y = y_axis.to(self._device).reshape(1, -1).to(b._rdtype)
        if layer_num == -1:
            z_prop = z_prop if z_prop <= 0. else 0.
        elif layer_num == b.layer_N:
            z_prop = z_prop if z_prop >= 0. else 0.
        coeffs = self._coeffs(layer_num, torch.as_tensor([float(z_prop)], device=self._device))
        Px = torch.exp(1j * om * kx[None, :] * x).to(cdt)                                         # [nx, N]
        Py = torch.exp(1j * om * ky[:, None] * y).to(cdt)                                         # [N, ny]
        out = [self._mm(Px * c.reshape(1, -1).to(cdt), Py).to(self._dtype) for c in coeffs]
        return out[:3], out[3:]
