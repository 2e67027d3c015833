"""Gradients through the HIP path (config 5 of BASELINE.json at a small order) against golden vectors produced by the
reference's autograd (tests/golden/grad_o32.npz; make_golden.py --grad): FoM = sum_pol |t_(1,0),pol|^2 of a 2-layer stack,
d FoM / d density (40x36 grid) and d FoM / d thickness, for stable_eig_grad True (broadening 1e-10 and None) and False.
Tolerance 1e-6 relative (broadening makes the gradient approximate by design)."""
import os

import numpy as np
import pytest
import torch

from tests.backends import BACKENDS
from tests.helpers import GOLDEN
from tests.test_pipeline import make_engine


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("stable,bp,tag", [(True, 1e-10, "stable1_bpe-10"), (True, None, "stable1_bpnone"), (False, 1e-10, "stable0_bpe-10")])
def test_gradient_matches_reference(backend, stable, bp, tag):
    import torcwa_amd
    eng = make_engine(backend)
    g = np.load(os.path.join(GOLDEN, "grad_o32.npz"))
    eps_si = complex(g["eps_si"])
    old = torcwa_amd.Eig.broadening_parameter
    torcwa_amd.Eig.broadening_parameter = bp
    try:
        rho = torch.from_numpy(g["rho"]).to(eng.device).requires_grad_(True)
        thick = torch.tensor(300., dtype=torch.float64, device=eng.device, requires_grad=True)
        sim = torcwa_amd.rcwa(freq=1 / 532., order=[3, 2], L=[700., 300.], dtype=torch.complex128, engine=eng, stable_eig_grad=stable)
        sim.add_input_layer(eps=1.46 ** 2)
        sim.set_incident_angle(inc_ang=0., azi_ang=0.)
        sim.add_layer(thickness=thick, eps=rho * eps_si + (1. - rho))
        sim.add_layer(thickness=80., eps=2.25)
        sim.solve_global_smatrix()
        t = [sim.S_parameters(orders=[1, 0], direction="forward", port="transmission", polarization=p, ref_order=[0, 0]) for p in ("xx", "yx", "xy", "yy")]
        fom = sum(torch.abs(v) ** 2 for v in t)
        fom.sum().backward()
    finally:
        torcwa_amd.Eig.broadening_parameter = old
    fom_v, fom_ref = float(fom.detach().reshape(-1)[0]), float(np.asarray(g[f"{tag}_fom"]).reshape(-1)[0])
    assert abs(fom_v - fom_ref) / fom_ref < 1e-9
    gr = rho.grad.cpu().numpy()
    ref = g[f"{tag}_grad_rho"]
    assert np.abs(gr - ref).max() / np.abs(ref).max() < 1e-6
    gt_ref = float(np.asarray(g[f"{tag}_grad_thick"]).reshape(-1)[0])
    assert abs(float(thick.grad.reshape(-1)[0]) - gt_ref) / abs(gt_ref) < 1e-6


@pytest.mark.parametrize("backend", BACKENDS)
def test_gradient_through_forced_q_branch(backend):
    """avoid_Pinv_instability on the differentiable path (rcwa.py:1249-1262): with a threshold below rounding noise every layer takes
    V = Q W Kz^-1; mathematically the same V, so FoM and gradient must agree with the P-solve branch -- and the discarded P-solve
    (evaluated on the identity for those points) must leave no trace in the gradient."""
    import torcwa_amd
    eng = make_engine(backend)
    g = np.load(os.path.join(GOLDEN, "grad_o32.npz"))
    eps_si = complex(g["eps_si"])
    res = []
    for kw in ({}, {"avoid_Pinv_instability": True, "max_Pinv_instability": 1e-17}):
        rho = torch.from_numpy(g["rho"]).to(eng.device).requires_grad_(True)
        sim = torcwa_amd.rcwa(freq=1 / 532., order=[3, 2], L=[700., 300.], dtype=torch.complex128, engine=eng, **kw)
        sim.add_input_layer(eps=1.46 ** 2)
        sim.set_incident_angle(inc_ang=0., azi_ang=0.)
        sim.add_layer(thickness=300., eps=rho * eps_si + (1. - rho))
        sim.solve_global_smatrix()
        t = sim.S_parameters(orders=[1, 0], direction="forward", port="transmission", polarization="xx", ref_order=[0, 0])
        fom = (torch.abs(t) ** 2).sum()
        fom.backward()
        res.append((float(fom.detach()), rho.grad.cpu().numpy()))
        if kw:
            assert len(sim.Pinv_instability) == 1 and float(sim.Pinv_instability[0]) < 1e-10
    (f0, g0), (f1, g1) = res
    assert np.isfinite(g1).all()
    assert abs(f1 - f0) / abs(f0) < 1e-9
    assert np.abs(g1 - g0).max() / np.abs(g0).max() < 1e-6


def _geo(eng):
    import torcwa_amd
    return torcwa_amd.geometry(Lx=300., Ly=300., nx=120, ny=120, edge_sharpness=60., dtype=torch.float64, device=eng.device)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("tag,stable,bp", [("exact", False, 1e-10), ("bpe-10", True, 1e-10), ("bpnone", True, None)])
def test_shape_derivative_of_a_cylinder(backend, tag, stable, bp):
    """example/Example4.ipynb at a small order: |txx|^2 differentiated through the level-set geometry with respect to the radius of a
    cylinder (C4v: degenerate mode pairs), exact and stabilised eigen-gradient, against the reference's autograd
    (tests/golden/shape_grad.npz; make_golden.py --shape-grad).  1e-6 relative, as the other gradient cases."""
    import torcwa_amd
    eng = make_engine(backend)
    g = np.load(os.path.join(GOLDEN, "shape_grad.npz"))
    old = torcwa_amd.Eig.broadening_parameter
    torcwa_amd.Eig.broadening_parameter = bp
    try:
        for R0 in ((97,) if backend == "emu" and tag != "bpe-10" else (88, 97)):          # emulator time budget
            R = torch.tensor(float(R0), dtype=torch.float64, device=eng.device, requires_grad=True)
            sim = torcwa_amd.rcwa(freq=1 / 473., order=[3, 3], L=[300., 300.], dtype=torch.complex128, engine=eng, stable_eig_grad=stable)
            sim.add_input_layer(eps=1.46 ** 2)
            sim.set_incident_angle(inc_ang=0., azi_ang=0.)
            m = _geo(eng).circle(R=R, Cx=150., Cy=150.)
            sim.add_layer(thickness=600., eps=m * 2.0709 ** 2 + (1. - m))
            sim.solve_global_smatrix()
            txx = sim.S_parameters(orders=[0, 0], direction="forward", port="transmission", polarization="xx", ref_order=[0, 0])
            (torch.abs(txx) ** 2).sum().backward()
            ref_t = complex(np.asarray(g[f"circle_{tag}_R{R0}_txx"]).reshape(-1)[0])
            ref_g = float(np.asarray(g[f"circle_{tag}_R{R0}_grad"]).reshape(-1)[0])
            assert abs(complex(txx.detach().reshape(-1)[0]) - ref_t) / abs(ref_t) < 1e-9
            # scale of the derivative: d|txx|^2/dR reaches 1.14 / nm at R = 97 and passes through zero near R = 88
            assert abs(float(R.grad) - ref_g) < 1e-6 * max(abs(ref_g), 1e-2), (float(R.grad), ref_g)
    finally:
        torcwa_amd.Eig.broadening_parameter = old


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("th", [0.0, 0.3])
def test_shape_derivative_of_a_rectangle(backend, th):
    """example/Example5.ipynb's objective |tyy - txx| differentiated with respect to (Wx, Wy) and the rotation angle of the rectangle
    (theta = 0: mirror-symmetric, d/dtheta vanishes; theta = 0.3: no symmetry left), against the reference's autograd."""
    import torcwa_amd
    eng = make_engine(backend)
    g = np.load(os.path.join(GOLDEN, "shape_grad.npz"))
    eps_si = complex(g["eps_si"])
    W = torch.tensor([180., 100.], dtype=torch.float64, device=eng.device, requires_grad=True)
    theta = torch.tensor(th, dtype=torch.float64, device=eng.device, requires_grad=True)
    sim = torcwa_amd.rcwa(freq=1 / 532., order=[3, 3], L=[300., 300.], dtype=torch.complex128, engine=eng)
    sim.add_input_layer(eps=1.46 ** 2)
    sim.set_incident_angle(inc_ang=0., azi_ang=0.)
    m = _geo(eng).rectangle(Wx=W[0], Wy=W[1], Cx=150., Cy=150., theta=theta)
    sim.add_layer(thickness=250., eps=m * eps_si + (1. - m))
    sim.solve_global_smatrix()
    txx = sim.S_parameters(orders=[0, 0], direction="forward", port="transmission", polarization="xx", ref_order=[0, 0])
    tyy = sim.S_parameters(orders=[0, 0], direction="forward", port="transmission", polarization="yy", ref_order=[0, 0])
    torch.abs(tyy - txx).sum().backward()
    key = f"rect_th{int(round(th * 10))}"
    for v, name in ((txx, "_txx"), (tyy, "_tyy")):
        ref = complex(np.asarray(g[key + name]).reshape(-1)[0])
        assert abs(complex(v.detach().reshape(-1)[0]) - ref) / abs(ref) < 1e-9
    refW = np.asarray(g[key + "_gradW"])
    assert np.abs(W.grad.cpu().numpy() - refW).max() / np.abs(refW).max() < 1e-6
    ref_th = float(np.asarray(g[key + "_gradtheta"]).reshape(-1)[0])
    assert abs(float(theta.grad) - ref_th) < 1e-6 * max(abs(ref_th), np.abs(refW).max())
