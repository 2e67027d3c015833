"""Dispersion tables for the sweep drivers: batched `material(lambda[B])` lookup.

`aSiH` mirrors the reference's example helper `Materials.aSiH` (example/Materials.py:5-52): n + ik of hydrogenated amorphous
silicon by CUBIC interpolation of a measured table (192-999 nm, 1 nm step), clamped to the end values outside the table, with
the reference's finite-difference derivative (dl = 0.005 nm) as its backward.  Differences by design: the table is a committed
array (`data/asih_nk_table.npz`) instead of a text file opened relative to the CWD on every call, the spline is built once, and
the wavelength may be a [B] tensor on the GPU (one lookup for a whole sweep).

The interpolant is the not-a-knot cubic spline (what scipy's `interp1d(kind='cubic')` constructs), restated here in its
second-derivative form; tests pin it to the reference's own outputs (tests/golden/material_asih.npz, asih_table.npz).
"""
import os

import numpy as np
import torch

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


class CubicTable:
    """Not-a-knot cubic spline through (x_i, y_i), y complex; evaluation is batched torch code on any device."""

    def __init__(self, x, y):
        x = np.asarray(x, dtype=np.float64)
        y = np.asarray(y, dtype=np.complex128)
        n = len(x)
        h = np.diff(x)
        A = np.zeros((n, n))
        rhs = np.zeros(n, dtype=np.complex128)
        for i in range(1, n - 1):                      # continuity of the first derivative at the interior knots
            A[i, i - 1], A[i, i], A[i, i + 1] = h[i - 1], 2 * (h[i - 1] + h[i]), h[i]
            rhs[i] = 6 * ((y[i + 1] - y[i]) / h[i] - (y[i] - y[i - 1]) / h[i - 1])
        # not-a-knot: the third derivative is continuous across the second and the second-to-last knot
        A[0, 0], A[0, 1], A[0, 2] = h[1], -(h[0] + h[1]), h[0]
        A[n - 1, n - 3], A[n - 1, n - 2], A[n - 1, n - 1] = h[n - 2], -(h[n - 3] + h[n - 2]), h[n - 3]
        M = np.linalg.solve(A, rhs)                    # second derivatives at the knots
        self.x0, self.x1 = float(x[0]), float(x[-1])
        self._np = (x, y, M)
        self._dev = {}

    def _tensors(self, device):
        key = str(device)
        if key not in self._dev:
            x, y, M = self._np
            self._dev[key] = tuple(torch.as_tensor(a, device=device) for a in (x, y, M))
        return self._dev[key]

    def __call__(self, xq):
        """xq: real float64 tensor of any shape -> complex128 values; clamped to the end values outside [x0, x1]."""
        x, y, M = self._tensors(xq.device)
        xc = xq.clamp(self.x0, self.x1)
        i = (torch.searchsorted(x, xc, right=True) - 1).clamp(0, len(x) - 2)
        h = x[i + 1] - x[i]
        a, b = (x[i + 1] - xc) / h, (xc - x[i]) / h
        v = a * y[i] + b * y[i + 1] + ((a ** 3 - a) * M[i] + (b ** 3 - b) * M[i + 1]) * (h * h) / 6
        v = torch.where(xq < self.x0, y[0], v)
        return torch.where(xq > self.x1, y[-1], v)


_asih = None


def _asih_table():
    global _asih
    if _asih is None:
        z = np.load(os.path.join(_DATA, "asih_nk_table.npz"))
        _asih = CubicTable(z["lam"], z["n"] + 1j * z["k"])
    return _asih


def asih_nk(wavelength):
    """n + ik of a-Si:H at `wavelength` (nm): scalar / [B] tensor -> complex128 tensor of the same shape."""
    w = torch.as_tensor(wavelength)
    return _asih_table()(torch.real(w).to(torch.float64))


class aSiH(torch.autograd.Function):
    """`aSiH.apply(wavelength)` -> n + ik (example/Materials.py:5-52); wavelength may be 0-d (as in the reference) or [B]."""

    @staticmethod
    def forward(ctx, wavelength, dl=0.005):
        w = torch.real(wavelength.detach()).to(torch.float64)
        tab = _asih_table()
        ctx.dnk_dl = (tab(w + dl) - tab(w - dl)) / (2 * dl)                  # Materials.py:29-46
        wide = wavelength.dtype in (torch.float64, torch.complex128)         # Materials.py:48-49
        return tab(w).to(torch.complex128 if wide else torch.complex64)

    @staticmethod
    def backward(ctx, grad_output):
        grad = 2 * torch.real(torch.conj(grad_output) * ctx.dnk_dl)          # Materials.py:52-53
        return grad, None
