"""The oracle (oracle/rcwa_oracle.py) against the golden vectors produced by the real reference.

This is what PINS the oracle.  CPU only.
"""
import numpy as np
import pytest
import torch

from oracle import rcwa_oracle as orc
from tests.helpers import CASES, DIRPORT, ORDERS_PROBE, POLS, case_inputs, load_case, relerr

TOL = {"c128": 2e-11, "c64": 5e-3}      # c64: the reference's own fp32 round-off
TOL_SP = {"c128": 2e-11, "c64": 5e-3}


def _solve(g, dtype, **kw):
    ci = case_inputs(g, dtype)
    return orc.solve_stack(ci.pop("freq"), ci.pop("order"), ci.pop("L"), ci.pop("layers"), **ci, **kw)


@pytest.mark.parametrize("dtype", ["c128", "c64"])
@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference(name, dtype):
    g = load_case(name, dtype)
    avoid = name.endswith("avoidPinv")
    s, lays, S, C = _solve(g, dtype, avoid_Pinv_instability=avoid)
    tol = TOL[dtype]
    assert relerr(s.kx.numpy(), g["kx"]) < 1e-6 if dtype == "c64" else relerr(s.kx.numpy(), g["kx"]) < 1e-14
    # global S: norms, central block, full where stored
    fro = np.array([np.linalg.norm(x.numpy()) for x in S])
    assert np.allclose(fro, g["S_fro"], rtol=tol * 10, atol=tol)
    cidx = g["central_idx"]
    for k in range(4):
        if f"S{k}_central" in g:
            assert relerr(S[k].numpy()[np.ix_(cidx, cidx)], g[f"S{k}_central"]) < tol
        if f"S{k}" in g:
            assert relerr(S[k].numpy(), g[f"S{k}"]) < tol
    # per-layer invariants
    for li, lay in enumerate(lays):
        lam = (lay.kz ** 2).numpy()
        idx = np.lexsort((lam.imag, lam.real))
        ref = g[f"L{li}_kz2_sorted"]
        # sorting by real part can permute near-degenerate pairs: compare as multisets via nearest match
        assert relerr(np.sort_complex(lam[idx]), np.sort_complex(ref)) < max(tol, 1e-9) * (50 if dtype == "c64" else 1)
        if f"L{li}_E" in g:
            assert relerr(lay.E.numpy(), g[f"L{li}_E"]) < (1e-13 if dtype == "c128" else 1e-5)
            assert relerr(lay.P.numpy(), g[f"L{li}_P"]) < tol
            assert relerr(lay.Q.numpy(), g[f"L{li}_Q"]) < tol
            for k, nm in enumerate(("S11", "S21", "S12", "S22")):
                assert relerr(lay.S[k].numpy(), g[f"L{li}_{nm}"]) < tol
    if "Vf" in g:
        assert relerr(s.Vf.numpy(), g["Vf"]) < (1e-14 if dtype == "c128" else 1e-6)
        if bool(g["has_in"]):
            for k in range(4):
                assert relerr(s.Sin[k].numpy(), g[f"Sin{k}"]) < (1e-13 if dtype == "c128" else 1e-5)
        if bool(g["has_out"]):
            for k in range(4):
                assert relerr(s.Sout[k].numpy(), g[f"Sout{k}"]) < (1e-13 if dtype == "c128" else 1e-5)
    if avoid:
        assert np.allclose(float(lays[0].Pinv_instability), g["Pinv_instability"][0], rtol=1e-3 if dtype == "c128" else 0.5)
    # S-parameters: 8 polarisations x 4 (direction, port) x probe orders (incl. clamped out-of-range order)
    sp = g["sparams"]
    for a, (dr, pt) in enumerate(DIRPORT):
        for b, pol in enumerate(POLS):
            v = orc.s_parameters(s, S, ORDERS_PROBE, direction=dr, port=pt, polarization=pol, ref_order=[0, 0]).numpy()
            scale = max(np.abs(sp[a, b]).max(), 1e-3)
            assert np.abs(v - sp[a, b]).max() / scale < TOL_SP[dtype], (dr, pt, pol)
    v = orc.s_parameters(s, S, ORDERS_PROBE, direction="f", port="t", polarization="yx", ref_order=[-1, 1], power_norm=False).numpy()
    assert np.abs(v - g["sparams_yx_ref_m1p1_nonorm"]).max() < TOL_SP[dtype]
    v = orc.s_parameters(s, S, ORDERS_PROBE, direction="f", port="r", polarization="ps", ref_order=[0, 1]).numpy()
    assert np.abs(v - g["sparams_ps_ref_0p1_refl"]).max() < TOL_SP[dtype]


def test_matching_indices_bit_exact():
    """Diffraction-order -> flat index map incl. clamping and in-place mutation (rcwa.py:1115-1122)."""
    o = torch.tensor([[0, 0], [3, -2], [-3, 2], [4, 9], [-7, -9]], dtype=torch.int64)
    idx = orc.matching_indices(o, [3, 2])
    assert idx.tolist() == [17, 30, 4, 34, 0]
    assert o.tolist() == [[0, 0], [3, -2], [-3, 2], [3, 2], [-3, -2]]   # clamped in place


def test_geometry_recipe_pin(golden_dir):
    import os
    import zlib
    z = np.load(os.path.join(golden_dir, "geometry_pin.npz"))
    r = orc.rectangle_density(300, 300, 300., 300., 180., 100., 150., 150.).numpy()
    assert np.uint32(zlib.crc32(r.tobytes())) == z["rect_crc"]
    r30 = orc.rectangle_density(300, 300, 300., 300., 180., 100., 150., 150., theta=30 / 180 * np.pi).numpy()
    assert np.allclose(r30[150], z["rect30_row150"], rtol=0, atol=1e-15)
    assert abs(r30.sum() - z["rect30_sum"]) < 1e-9


def test_fresnel_known_answer():
    """Example0: |r|^2 of a glass(1.46)->air interface against the analytic Fresnel formulas."""
    n1, n2 = 1.46, 1.0
    for deg in (0.0, 10.0, 30.0, 40.0, 60.0):
        th = deg * np.pi / 180
        s, lays, S, C = orc.solve_stack(1 / 532., [1, 1], [300., 300.], [], eps_in=n1 ** 2, inc_ang=th)
        rpp = orc.s_parameters(s, S, [0, 0], direction="f", port="r", polarization="pp").numpy()[0]
        rss = orc.s_parameters(s, S, [0, 0], direction="f", port="r", polarization="ss").numpy()[0]
        ct = np.sqrt(complex(1 - (n1 / n2 * np.sin(th)) ** 2))
        R_tm = abs((n1 * ct - n2 * np.cos(th)) / (n1 * ct + n2 * np.cos(th))) ** 2
        R_te = abs((n1 * np.cos(th) - n2 * ct) / (n1 * np.cos(th) + n2 * ct)) ** 2
        assert abs(abs(rpp) ** 2 - R_tm) < 1e-8
        assert abs(abs(rss) ** 2 - R_te) < 1e-8
