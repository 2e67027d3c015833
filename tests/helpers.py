"""Shared helpers for the parity tests (test infrastructure)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

ORDERS_PROBE = [[0, 0], [1, 0], [-1, 0], [0, 1], [0, -1], [-1, -1], [2, 1], [99, -99]]
POLS = ["xx", "yx", "xy", "yy", "pp", "sp", "ps", "ss"]
DIRPORT = [("forward", "transmission"), ("forward", "reflection"), ("backward", "reflection"), ("backward", "transmission")]

CASES = ["fresnel_0", "fresnel_30", "fresnel_60", "example1_o3", "example1_o5", "example2_o4",
         "example1_1_o4", "asym_o32", "asym_o32_avoidPinv"]


def load_case(name, dtype):
    z = np.load(os.path.join(GOLDEN, f"{name}_{dtype}.npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def case_inputs(g, dtype):
    """Rebuild the exact inputs the reference was given (see tests/golden/make_golden.py:run_case)."""
    cdt = torch.complex128 if dtype == "c128" else torch.complex64
    rdt = torch.float64 if dtype == "c128" else torch.float32
    layers = []
    for li in range(int(g["n_layers"])):
        vals = []
        for nm in ("eps", "mu"):
            if f"L{li}_{nm}_grid" in g:
                t = torch.from_numpy(g[f"L{li}_{nm}_grid"])
                t = t.to(cdt if torch.is_complex(t) else rdt)
                vals.append(t)
            else:
                v = complex(g[f"L{li}_{nm}_scalar"])
                vals.append(v.real if v.imag == 0 else v)
        layers.append((float(g[f"L{li}_thickness"]), vals[0], vals[1]))
    kw = dict(freq=float(g["freq"]), order=[int(v) for v in g["order"]], L=[float(v) for v in g["L"]],
              layers=layers, dtype=cdt, inc_ang=float(g["inc"]), azi_ang=float(g["azi"]),
              angle_layer=str(g["angle_layer"]))
    if bool(g["has_in"]):
        kw["eps_in"] = float(np.real(g["eps_in"]))
    if bool(g["has_out"]):
        kw["eps_out"] = float(np.real(g["eps_out"]))
    return kw


def relerr(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    d = np.linalg.norm((a - b).ravel())
    s = np.linalg.norm(b.ravel())
    return d / s if s > 0 else d


def multiset_dist(a, b):
    """Relative distance between two complex multisets after greedy nearest matching (order-independent; robust to
    pairs like x+iy / x-iy whose sort order flips on a 1e-16 change of the real part)."""
    a = np.asarray(a, dtype=np.complex128).ravel()
    b = list(np.asarray(b, dtype=np.complex128).ravel())
    assert len(a) == len(b)
    worst = 0.0
    for z in a:
        j = int(np.argmin(np.abs(np.array(b) - z)))
        worst = max(worst, abs(b[j] - z))
        b.pop(j)
    return worst / max(np.abs(a).max(), 1e-300)


def config5_density(nx=700, ny=300, beta=6.0):
    """Deterministic stand-in for Example 6's blurred, tanh-projected random density (same recipe as
    tests/golden/make_golden.py:config5_density; the fixture stores a checksum and a sub-sample of it)."""
    x = (np.arange(nx) + 0.5) / nx
    y = (np.arange(ny) + 0.5) / ny
    X, Y = np.meshgrid(x, y, indexing="ij")
    f = (0.50 + 0.22 * np.cos(2 * np.pi * (1 * X) + 0.3) * np.cos(2 * np.pi * 1 * Y) + 0.17 * np.cos(2 * np.pi * (2 * X) + 1.1)
         + 0.12 * np.cos(2 * np.pi * (3 * X) + 2.0) * np.cos(2 * np.pi * 2 * Y) + 0.08 * np.cos(2 * np.pi * (5 * X) + 0.7) * np.cos(2 * np.pi * 1 * Y))
    rho = 0.5 + np.tanh(2 * beta * f - beta) / (2 * np.tanh(beta))
    return rho.astype(np.float32)
