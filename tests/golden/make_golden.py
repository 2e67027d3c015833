#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING the real reference (kch3782/torcwa 0.1.4.2).

Run only in the build container, where /root/reference exists:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Writes small .npz fixtures next to this file.  Nothing of the reference's source travels:
the fixtures are inputs + the reference's numerical outputs.  The GPU box only reads the .npz.
"""
import os
import sys
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, REF)
sys.dont_write_bytecode = True
import torcwa  # noqa: E402  (the reference)

assert torcwa.__version__ == "0.1.4.2"
torch.set_num_threads(8)

ORDERS_PROBE = [[0, 0], [1, 0], [-1, 0], [0, 1], [0, -1], [-1, -1], [2, 1], [99, -99]]   # last one clamps
POLS = ["xx", "yx", "xy", "yy", "pp", "sp", "ps", "ss"]
DIRPORT = [("forward", "transmission"), ("forward", "reflection"), ("backward", "reflection"), ("backward", "transmission")]


def asih_nk(lams):
    """n+ik of a-Si:H from the reference's example helper (example/Materials.py), c128."""
    cwd = os.getcwd()
    os.chdir(os.path.join(REF, "example"))
    sys.path.insert(0, os.path.join(REF, "example"))
    import Materials
    out = []
    for lam in lams:
        out.append(complex(Materials.aSiH.apply(torch.tensor(float(lam), dtype=torch.float64))))
    os.chdir(cwd)
    return np.array(out, dtype=np.complex128)


def ref_geometry(nx, ny, Lx, Ly, dtype):
    g = torcwa.geometry(Lx=Lx, Ly=Ly, nx=nx, ny=ny, edge_sharpness=1000., dtype=dtype, device=torch.device("cpu"))
    g.grid()
    return g


def _via_f32(v):
    """Round a grid through float32 / complex64 and return it in float64 / complex128 (a float32-representable input)."""
    if not torch.is_tensor(v):
        return v
    return v.to(torch.complex64).to(torch.complex128) if torch.is_complex(v) else v.to(torch.float32).to(torch.float64)


def run_case(name, *, freq, order, L, layers, dtype, eps_in=None, eps_out=None, inc=0.0, azi=0.0,
             angle_layer="input", full_S=False, extra=None, avoid=False, tag=None, max_pinv=0.005):
    """layers: list of (thickness, eps, mu) with eps/mu python scalars or torch grids.
    dtype "c128f32": the reference runs in complex128 on grids that were rounded through float32 / complex64, i.e. on exactly
    the values a complex64 user hands over -- the gate of the complex64-I/O product run (<= 1e-5, no representation slack)."""
    if dtype == "c128f32":
        layers = [(d, _via_f32(eps), _via_f32(mu)) for (d, eps, mu) in layers]
    cdt = torch.complex64 if dtype == "c64" else torch.complex128
    rdt = torch.float32 if dtype == "c64" else torch.float64
    sim = torcwa.rcwa(freq=freq, order=order, L=L, dtype=cdt, device=torch.device("cpu"),
                      stable_eig_grad=False, avoid_Pinv_instability=avoid, max_Pinv_instability=max_pinv)
    if eps_in is not None:
        sim.add_input_layer(eps=eps_in)
    if eps_out is not None:
        sim.add_output_layer(eps=eps_out)
    sim.set_incident_angle(inc_ang=inc, azi_ang=azi, angle_layer=angle_layer)
    for (d, eps, mu) in layers:
        e = eps.to(cdt if torch.is_complex(eps) else rdt) if torch.is_tensor(eps) else eps
        m = mu.to(cdt if torch.is_complex(mu) else rdt) if torch.is_tensor(mu) else mu
        sim.add_layer(thickness=d, eps=e, mu=m)
    sim.solve_global_smatrix()

    out = {"freq": np.float64(freq), "order": np.array(order), "L": np.array(L, dtype=np.float64),
           "inc": np.float64(inc), "azi": np.float64(azi), "angle_layer": angle_layer,
           "has_in": eps_in is not None, "has_out": eps_out is not None,
           "eps_in": np.complex128(eps_in if eps_in is not None else 1.0),
           "eps_out": np.complex128(eps_out if eps_out is not None else 1.0),
           "n_layers": len(layers)}
    for li, (d, eps, mu) in enumerate(layers):
        out[f"L{li}_thickness"] = np.float64(d)
        for nm, v in (("eps", eps), ("mu", mu)):
            if torch.is_tensor(v):
                out[f"L{li}_{nm}_grid"] = v.numpy()
            else:
                out[f"L{li}_{nm}_scalar"] = np.complex128(v)
    N = sim.order_N
    n = 2 * N
    out["kx"] = sim.Kx_norm_dn.numpy()
    out["ky"] = sim.Ky_norm_dn.numpy()
    # S-parameters: 8 pols x 4 (dir, port) x probe orders
    sp = np.zeros((len(DIRPORT), len(POLS), len(ORDERS_PROBE)), dtype=np.complex128)
    for a, (dr, pt) in enumerate(DIRPORT):
        for b, pol in enumerate(POLS):
            if sim.S[1].dim() == 1 and (dr, pt) in (("forward", "reflection"), ("backward", "reflection")) and eps_in is None and eps_out is None:
                continue
            v = sim.S_parameters(orders=[list(o) for o in ORDERS_PROBE], direction=dr, port=pt, polarization=pol, ref_order=[0, 0])
            sp[a, b] = v.numpy()
    out["sparams"] = sp
    # same with ref_order=(1,0)-ish and no power norm for the xy branch (exercise index/normalisation paths)
    v = sim.S_parameters(orders=[list(o) for o in ORDERS_PROBE], direction="f", port="t", polarization="yx", ref_order=[-1, 1], power_norm=False)
    out["sparams_yx_ref_m1p1_nonorm"] = v.numpy()
    v = sim.S_parameters(orders=[list(o) for o in ORDERS_PROBE], direction="f", port="r", polarization="ps", ref_order=[0, 1])
    out["sparams_ps_ref_0p1_refl"] = v.numpy()
    S = [x.numpy() for x in sim.S]
    out["S_fro"] = np.array([np.linalg.norm(x) for x in S])
    # central sub-block |m|,|n|<=1, both polarisations
    oy = order[1]
    cidx = [(2 * oy + 1) * (m + order[0]) + (q + oy) for m in (-1, 0, 1) if abs(m) <= order[0] for q in (-1, 0, 1) if abs(q) <= oy]
    cidx = np.array(cidx + [c + N for c in cidx])
    out["central_idx"] = cidx
    for k in range(4):
        if S[k].ndim == 2:
            out[f"S{k}_central"] = S[k][np.ix_(cidx, cidx)]
            if full_S:
                out[f"S{k}"] = S[k]
    for li in range(len(layers)):
        kz = sim.kz_norm[li].numpy()
        lam = kz ** 2
        idx = np.lexsort((lam.imag, lam.real))
        out[f"L{li}_kz2_sorted"] = lam[idx]
        if full_S:
            out[f"L{li}_E"] = sim.eps_conv[li].numpy()
            out[f"L{li}_P"] = sim.P[li].numpy()
            out[f"L{li}_Q"] = sim.Q[li].numpy()
            out[f"L{li}_S11"] = sim.layer_S11[li].numpy()
            out[f"L{li}_S21"] = sim.layer_S21[li].numpy()
            out[f"L{li}_S12"] = sim.layer_S12[li].numpy()
            out[f"L{li}_S22"] = sim.layer_S22[li].numpy()
    if full_S:
        out["Vf"] = sim.Vf.numpy()
        if eps_in is not None:
            for k in range(4):
                out[f"Sin{k}"] = sim.Sin[k].numpy()
        if eps_out is not None:
            for k in range(4):
                out[f"Sout{k}"] = sim.Sout[k].numpy()
    if avoid:
        out["Pinv_instability"] = np.array([float(x) for x in sim.Pinv_instability])
        out["Qinv_instability"] = np.array([float(x) for x in sim.Qinv_instability])
    # diffraction angles (a small piece of host glue, rcwa.py:214-262)
    ia, aa = sim.diffraction_angle(orders=[list(o) for o in ORDERS_PROBE], layer="output", unit="degree")
    out["diff_inc_deg"], out["diff_azi_deg"] = ia.numpy(), aa.numpy()
    if extra:
        out.update(extra(sim))
    path = os.path.join(HERE, f"{name}_{tag or dtype}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}_{tag or dtype}: n={n}  txx00={sp[0, 0, 0]:.12g}  -> {os.path.getsize(path) / 1024:.0f} KiB")
    return sim


def main():
    lam_tab = np.concatenate([np.linspace(400., 700., 128), [532., 650.]])
    nk = asih_nk(lam_tab)
    np.savez_compressed(os.path.join(HERE, "asih_table.npz"), lam=lam_tab, nk=nk)
    eps_si = {532.: complex(nk[-2] ** 2), 650.: complex(nk[-1] ** 2)}
    print("eps_Si(532) =", eps_si[532.], " eps_Si(650) =", eps_si[650.])

    for dtype in ("c128", "c64", "c128f32"):
        if "--f32only" in sys.argv and dtype != "c128f32":
            continue
        rdt = torch.float32 if dtype == "c64" else torch.float64
        # --- Fresnel, no internal layer (Example0): glass -> air, 0/30/60 degrees ------------
        for ang in (0, 30, 60):
            run_case(f"fresnel_{ang}", freq=1 / 532., order=[2, 2], L=[300., 300.], layers=[], dtype=dtype,
                     eps_in=1.46 ** 2, inc=ang * np.pi / 180, full_S=(ang == 30))
        # --- Example1 rectangle, 1 layer, 300x300 grid --------------------------------------
        g = ref_geometry(300, 300, 300., 300., torch.float64)
        rect = g.rectangle(Wx=180., Wy=100., Cx=150., Cy=150.)
        eps1 = rect * eps_si[532.] + (1. - rect)
        for o in ([3, 3], [5, 5]):
            # store the density recipe, not the 300x300 grid (checksum pins it)
            sim = run_case(f"example1_o{o[0]}", freq=1 / 532., order=o, L=[300., 300.], dtype=dtype,
                           layers=[(300., eps1, 1.0)], eps_in=1.46 ** 2, full_S=(o[0] == 3))
        # --- Example2: 15 deg oblique incidence, square 120 ---------------------------------
        sq = g.square(W=120., Cx=150., Cy=150.)
        eps2 = sq * eps_si[532.] + (1. - sq)
        run_case("example2_o4", freq=1 / 532., order=[4, 4], L=[300., 300.], dtype=dtype,
                 layers=[(300., eps2, 1.0)], eps_in=1.46 ** 2, inc=15 * np.pi / 180)
        # --- Example1-1: literal 6-layer stack at 650 nm ------------------------------------
        su8 = 1.6 ** 2
        lays = []
        for th in (0., 30., 60.):
            r = g.rectangle(Wx=180., Wy=100., Cx=150., Cy=150., theta=th / 180 * np.pi)
            lays.append((200., r * eps_si[650.] + (1. - r) * su8, 1.0))
            lays.append((100., su8, 1.0))
        run_case("example1_1_o4", freq=1 / 650., order=[4, 4], L=[300., 300.], dtype=dtype, layers=lays, eps_in=1.46 ** 2)
        # --- asymmetric case: order [3,2], L=[320,410], 40x36 complex grids, patterned eps AND mu,
        #     oblique + azimuth, input and output half-spaces, angle referenced to the output layer.
        gen = torch.Generator().manual_seed(20260927)
        ge = torch.rand(40, 36, generator=gen, dtype=torch.float64)
        gm = torch.rand(40, 36, generator=gen, dtype=torch.float64)
        epsg = (1.0 + 5.0 * ge) + 1j * (0.3 * ge)
        mug = (1.0 + 0.4 * gm) + 0j
        gen2 = torch.Generator().manual_seed(7)
        g2 = torch.rand(40, 36, generator=gen2, dtype=torch.float64)
        eps_real = 1.0 + 3.0 * g2
        run_case("asym_o32", freq=1 / 600., order=[3, 2], L=[320., 410.], dtype=dtype,
                 layers=[(150., epsg, mug), (80., 2.25, 1.0), (120., eps_real, 1.0)],
                 eps_in=2.1, eps_out=1.7, inc=20 * np.pi / 180, azi=35 * np.pi / 180, angle_layer="output", full_S=True)
        run_case("asym_o32_avoidPinv", freq=1 / 600., order=[3, 2], L=[320., 410.], dtype=dtype,
                 layers=[(150., epsg, mug)], eps_in=2.1, inc=20 * np.pi / 180, azi=35 * np.pi / 180, avoid=True)

    # geometry recipe pin: the oracle's rectangle_density must reproduce this
    g = ref_geometry(300, 300, 300., 300., torch.float64)
    rect = g.rectangle(Wx=180., Wy=100., Cx=150., Cy=150.).numpy()
    rect30 = g.rectangle(Wx=180., Wy=100., Cx=150., Cy=150., theta=30 / 180 * np.pi).numpy()
    np.savez_compressed(os.path.join(HERE, "geometry_pin.npz"),
                        rect_sum=rect.sum(), rect_crc=np.uint32(zlib.crc32(rect.tobytes())),
                        rect_row150=rect[150], rect30_sum=rect30.sum(), rect30_row150=rect30[150])


if __name__ == "__main__" and not any(a in sys.argv for a in ("--fields", "--fields-larger", "--grad", "--fullsize", "--fullorder", "--geometry", "--rayleigh", "--shape-grad")):
    main()


# ---------------------------------------------------------------------------------------------------------
# field-map golden vectors: sources + field_xz / field_yz / field_xy
# ---------------------------------------------------------------------------------------------------------
def field_case(name, *, freq, order, L, layers, eps_in=None, eps_out=None, inc=0.0, azi=0.0, angle_layer="input", dtype="c128"):
    cdt = torch.complex128
    sim = torcwa.rcwa(freq=freq, order=order, L=L, dtype=cdt, device=torch.device("cpu"), stable_eig_grad=False)
    if eps_in is not None:
        sim.add_input_layer(eps=eps_in)
    if eps_out is not None:
        sim.add_output_layer(eps=eps_out)
    sim.set_incident_angle(inc_ang=inc, azi_ang=azi, angle_layer=angle_layer)
    for (d, eps, mu) in layers:
        sim.add_layer(thickness=d, eps=eps, mu=mu)
    sim.solve_global_smatrix()
    tot = sum(d for d, _, _ in layers)
    x = torch.tensor([3.0, 77.5, 150.0, 211.0, 299.0], dtype=torch.float64)
    y = torch.tensor([10.0, 120.0, 205.5], dtype=torch.float64)
    zs = [-120.0, -1.0, 0.0, 0.5]
    acc = 0.0
    for d, _, _ in layers:
        zs += [acc + 0.3 * d, acc + d]
        acc += d
    zs += [tot + 0.5, tot + 90.0]
    z = torch.tensor(zs, dtype=torch.float64)
    out = {"x": x.numpy(), "y": y.numpy(), "z": z.numpy()}
    srcs = [("planewave_xy_f", dict(kind="pw", amplitude=[1.0, 0.5j], direction="forward", notation="xy")),
            ("planewave_ps_b", dict(kind="pw", amplitude=[0.3, 1.0], direction="backward", notation="ps")),
            ("fourier_xy_f", dict(kind="fo", amplitude=[[1.0, 0.2], [0.1j, 0.4]], orders=[[0, 0], [1, -1]], direction="f", notation="xy"))]
    for sname, kw in srcs:
        kw = dict(kw)
        kind = kw.pop("kind")
        if kind == "pw":
            sim.source_planewave(**kw)
        else:
            sim.source_fourier(**kw)
        out[f"{sname}_Ei"] = sim.E_i.numpy()
        E, H = sim.field_xz(x, z, 133.0)
        out[f"{sname}_xz"] = np.stack([t.numpy() for t in E + H])
        E, H = sim.field_yz(y, z, 41.0)
        out[f"{sname}_yz"] = np.stack([t.numpy() for t in E + H])
        for ln, zp in [(-1, -35.0), (0, 0.4 * layers[0][0] if layers else 0.0), (len(layers) - 1, 10.0), (len(layers), 25.0)]:
            if ln < -1 or (not layers and ln in (0, len(layers) - 1) and ln != len(layers)):
                continue
            if ln >= 0 and ln < len(layers) or ln in (-1, len(layers)):
                E, H = sim.field_xy(int(ln), x, y, zp)
                out[f"{sname}_xy_L{ln}"] = np.stack([t.numpy() for t in E + H])
                out[f"{sname}_xy_L{ln}_zprop"] = np.float64(zp)
    path = os.path.join(HERE, f"fields_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"fields_{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def main_fields():
    lam_tab = np.load(os.path.join(HERE, "asih_table.npz"))
    nk532 = complex(lam_tab["nk"][-2])
    eps_si = nk532 ** 2
    # small grids (the field fixtures reuse the inputs stored in the S-matrix fixtures of the same name)
    z = np.load(os.path.join(HERE, "example1_o3_c128.npz"))
    eps1 = torch.from_numpy(z["L0_eps_grid"])
    field_case("example1_o3", freq=1 / 532., order=[3, 3], L=[300., 300.], layers=[(300., eps1, 1.0)], eps_in=1.46 ** 2)
    z = np.load(os.path.join(HERE, "asym_o32_c128.npz"))
    lays = [(150., torch.from_numpy(z["L0_eps_grid"]), torch.from_numpy(z["L0_mu_grid"])), (80., 2.25, 1.0),
            (120., torch.from_numpy(z["L2_eps_grid"]), 1.0)]
    field_case("asym_o32", freq=1 / 600., order=[3, 2], L=[320., 410.], layers=lays, eps_in=2.1, eps_out=1.7,
               inc=20 * np.pi / 180, azi=35 * np.pi / 180, angle_layer="output")


def _layers_of(z):
    lays = []
    for li in range(int(z["n_layers"])):
        vals = []
        for nm in ("eps", "mu"):
            if f"L{li}_{nm}_grid" in z.files:
                vals.append(torch.from_numpy(z[f"L{li}_{nm}_grid"]))
            else:
                v = complex(z[f"L{li}_{nm}_scalar"])
                vals.append(v.real if v.imag == 0 else v)
        lays.append((float(z[f"L{li}_thickness"]), vals[0], vals[1]))
    return lays


def main_fields_larger():
    """Field maps at larger orders, on the inputs of existing S-matrix fixtures: the single layer at [5,5] (n = 242) and the 6-layer
    stack of config 3 at [8,8] (n = 578; fields inside every one of its patterned and homogeneous layers)."""
    for name, src in (("example1_o5", "example1_o5_c128"), ("config3_o8_l500", "config3_o8_l500_c128f32")):
        z = np.load(os.path.join(HERE, src + ".npz"))
        field_case(name, freq=float(z["freq"]), order=[int(v) for v in z["order"]], L=[float(v) for v in z["L"]], layers=_layers_of(z),
                   eps_in=float(np.real(z["eps_in"])) if bool(z["has_in"]) else None, eps_out=float(np.real(z["eps_out"])) if bool(z["has_out"]) else None,
                   inc=float(z["inc"]), azi=float(z["azi"]), angle_layer=str(z["angle_layer"]))


if __name__ == "__main__" and "--fields" in sys.argv:
    main_fields()
if __name__ == "__main__" and "--fields-larger" in sys.argv:
    main_fields_larger()


# ---------------------------------------------------------------------------------------------------------
# gradient golden vectors (config 5 of BASELINE.json at a small order): FoM, dFoM/d(density), dFoM/d(thickness)
# ---------------------------------------------------------------------------------------------------------
def main_grad():
    lam_tab = np.load(os.path.join(HERE, "asih_table.npz"))
    eps_si = complex(lam_tab["nk"][-2]) ** 2
    gen = torch.Generator().manual_seed(333)
    rho0 = torch.rand(40, 36, generator=gen, dtype=torch.float64)
    rho0 = (rho0 + torch.flip(rho0, dims=[1])) / 2          # symmetrised like Example6
    out = {"rho": rho0.numpy(), "eps_si": np.complex128(eps_si)}
    for stable, bp in ((True, 1e-10), (True, None), (False, 1e-10)):
        torcwa.Eig.broadening_parameter = bp
        rho = rho0.clone().requires_grad_(True)
        thick = torch.tensor(300., dtype=torch.float64, requires_grad=True)
        sim = torcwa.rcwa(freq=1 / 532., order=[3, 2], L=[700., 300.], dtype=torch.complex128, device=torch.device("cpu"),
                          stable_eig_grad=stable)
        sim.add_input_layer(eps=1.46 ** 2)
        sim.set_incident_angle(inc_ang=0., azi_ang=0.)
        eps = rho * eps_si + (1. - rho)
        sim.add_layer(thickness=thick, eps=eps)
        sim.add_layer(thickness=80., eps=2.25)
        sim.solve_global_smatrix()
        t1xx = sim.S_parameters(orders=[1, 0], direction="forward", port="transmission", polarization="xx", ref_order=[0, 0])
        t1yx = sim.S_parameters(orders=[1, 0], direction="forward", port="transmission", polarization="yx", ref_order=[0, 0])
        t1xy = sim.S_parameters(orders=[1, 0], direction="forward", port="transmission", polarization="xy", ref_order=[0, 0])
        t1yy = sim.S_parameters(orders=[1, 0], direction="forward", port="transmission", polarization="yy", ref_order=[0, 0])
        fom = torch.abs(t1xx) ** 2 + torch.abs(t1yx) ** 2 + torch.abs(t1xy) ** 2 + torch.abs(t1yy) ** 2
        fom.sum().backward()
        tag = f"stable{int(stable)}_bp{'none' if bp is None else 'e-10'}"
        out[f"{tag}_fom"] = fom.detach().numpy()
        out[f"{tag}_t1xx"] = t1xx.detach().numpy()
        out[f"{tag}_grad_rho"] = rho.grad.numpy()
        out[f"{tag}_grad_thick"] = thick.grad.numpy()
        print(tag, "FoM", float(fom), "sum dFoM/drho", float(rho.grad.sum()), "dFoM/dd", float(thick.grad))
    torcwa.Eig.broadening_parameter = 1e-10
    np.savez_compressed(os.path.join(HERE, "grad_o32.npz"), **out)


if __name__ == "__main__" and "--grad" in sys.argv:
    main_grad()


# ---------------------------------------------------------------------------------------------------------
# full-size golden vectors: the configs of BASELINE.json at their real Fourier orders
# ---------------------------------------------------------------------------------------------------------
def _eps_si_c64(lam):
    """eps of a-Si:H at `lam` as the complex64 value a complex64 user of the reference computes (Materials.aSiH.apply(l)**2)."""
    nk = asih_nk([lam])[0]
    return complex(torch.tensor(nk, dtype=torch.complex64) ** 2)


def _grid_c64(dens32, eps_core, eps_bg=1.0):
    """complex64 permittivity grid exactly as the notebooks build it: geo*eps_core + (1-geo)*eps_bg in complex64 on the CPU."""
    e = torch.tensor(eps_core, dtype=torch.complex64)
    return (dens32 * e + (1. - dens32) * eps_bg).to(torch.complex64)


def main_fullsize(which):
    """All inputs are float32 / complex64-representable, so ONE complex128 reference run is the gate of both the complex128
    product run (<= 1e-9) and the complex64-I/O product run (<= 1e-5)."""
    g32 = ref_geometry(300, 300, 300., 300., torch.float32)
    glass = 1.46 ** 2
    if "config2" in which:
        # config 2: Example-1 rectangle, one layer, order [15,15], three wavelengths of the 128-point sweep
        rect = g32.rectangle(Wx=180., Wy=100., Cx=150., Cy=150.)
        for lam in (400., 532., 700.):
            eps = _grid_c64(rect, _eps_si_c64(lam))
            run_case(f"config2_o15_l{int(lam)}", freq=1 / lam, order=[15, 15], L=[300., 300.], dtype="c128f32", tag="c128f32",
                     layers=[(300., eps, 1.0)], eps_in=glass)
    if "config4" in which:
        # config 4: Example-3 style (Wx, Wy, lambda) grid, order [15,15]; a 2 x 2 x 2 corner-ish sample of the 16^3 sweep
        Wv = np.linspace(50., 250., 16)
        Lv = np.linspace(400., 700., 16)
        for iw in (2, 13):
            for jw in (4, 11):
                for kl in (1, 14):
                    wx, wy, lam = float(np.float32(Wv[iw])), float(np.float32(Wv[jw])), float(Lv[kl])
                    rect = g32.rectangle(Wx=wx, Wy=wy, Cx=150., Cy=150.)
                    eps = _grid_c64(rect, _eps_si_c64(lam))
                    run_case(f"config4_o15_w{iw}_{jw}_l{kl}", freq=1 / lam, order=[15, 15], L=[300., 300.], dtype="c128f32", tag="c128f32",
                             layers=[(300., eps, 1.0)], eps_in=glass,
                             extra=lambda sim, wx=wx, wy=wy, lam=lam: {"Wx": np.float64(wx), "Wy": np.float64(wy), "lam": np.float64(lam)})
    if "config3" in which:
        # config 3: the literal 6-layer stack of Example1-1 (3 rotated rectangles in SU8 + 3 SU8 spacers), order [8,8]
        su8 = 1.6 ** 2
        for lam in (650., 500.):
            lays = []
            for th in (0., 30., 60.):
                r = g32.rectangle(Wx=180., Wy=100., Cx=150., Cy=150., theta=th / 180 * np.pi)
                lays.append((200., _grid_c64(r, _eps_si_c64(lam), su8), 1.0))
                lays.append((100., su8, 1.0))
            run_case(f"config3_o8_l{int(lam)}", freq=1 / lam, order=[8, 8], L=[300., 300.], dtype="c128f32", tag="c128f32",
                     layers=lays, eps_in=glass)
    if "config5" in which:
        main_config5()


def config5_density(nx=700, ny=300, beta=6.0):
    """Deterministic stand-in for Example 6's blurred, tanh-projected random density (the notebook draws it on the CUDA RNG,
    which is not reproducible): a closed-form sum of cosines, symmetric under y -> Ly - y like the notebook's (rho + fliplr)/2,
    projected with the notebook's tanh formula, rounded to float32.  tests/helpers.py holds the same recipe."""
    x = (np.arange(nx) + 0.5) / nx
    y = (np.arange(ny) + 0.5) / ny
    X, Y = np.meshgrid(x, y, indexing="ij")
    f = (0.50 + 0.22 * np.cos(2 * np.pi * (1 * X) + 0.3) * np.cos(2 * np.pi * 1 * Y) + 0.17 * np.cos(2 * np.pi * (2 * X) + 1.1)
         + 0.12 * np.cos(2 * np.pi * (3 * X) + 2.0) * np.cos(2 * np.pi * 2 * Y) + 0.08 * np.cos(2 * np.pi * (5 * X) + 0.7) * np.cos(2 * np.pi * 1 * Y))
    rho = 0.5 + np.tanh(2 * beta * f - beta) / (2 * np.tanh(beta))
    return rho.astype(np.float32)


def main_config5(order=(15, 8), name="config5_o15_8_c128"):
    """config 5: Example-6 geometry L=[700,300], 700x300 grid, order [15,8] (the notebook's own order) or [25,25] (BASELINE.json's),
    complex128, FoM = sum over the four polarisation pairs of |t_(1,0)|^2, gradient w.r.t. the density through the stabilised Eig."""
    lam = 532.
    eps_si = complex(asih_nk([lam])[0] ** 2)
    rho0 = torch.from_numpy(config5_density().astype(np.float64))
    out = {"eps_si": np.complex128(eps_si), "rho_sum": np.float64(rho0.sum()), "rho_sub": rho0[::70, ::30].numpy(), "lam": np.float64(lam)}
    torcwa.Eig.broadening_parameter = 1e-10
    rho = rho0.clone().requires_grad_(True)
    sim = torcwa.rcwa(freq=1 / lam, order=list(order), L=[700., 300.], dtype=torch.complex128, device=torch.device("cpu"), stable_eig_grad=True)
    sim.add_input_layer(eps=1.46 ** 2)
    sim.set_incident_angle(inc_ang=0., azi_ang=0.)
    sim.add_layer(thickness=300., eps=rho * eps_si + (1. - rho))
    sim.solve_global_smatrix()
    ts = {p: sim.S_parameters(orders=[1, 0], direction="forward", port="transmission", polarization=p, ref_order=[0, 0]) for p in ("xx", "yy", "xy", "yx")}
    fom = sum(torch.abs(t) ** 2 for t in ts.values())
    fom.sum().backward()
    gr = rho.grad.numpy()
    out.update(fom=fom.detach().numpy(), grad_sum=np.float64(gr.sum()), grad_l2=np.float64(np.linalg.norm(gr)), grad_sub=gr[::7, ::3].copy(),
               **{f"t1{p}": t.detach().numpy() for p, t in ts.items()})
    lam2 = sim.kz_norm[0].detach().numpy() ** 2
    out["L0_kz2_sorted"] = lam2[np.lexsort((lam2.imag, lam2.real))]
    out["order"] = np.array(order)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, ": FoM", float(fom), "sum grad", gr.sum(), "|grad|", np.linalg.norm(gr))


if __name__ == "__main__" and "--fullsize" in sys.argv:
    sel = [a for a in sys.argv[1:] if a.startswith("config")] or ["config2", "config3", "config4", "config5"]
    main_fullsize(sel)


# ---------------------------------------------------------------------------------------------------------
# configs 3 and 5 at BASELINE.json's OWN Fourier orders ([21,21]: n = 3698; [25,25]: n = 5202) -- one-off runs of the reference
# (about 15 and 40 minutes on the 8-core build container):  python tests/golden/make_golden.py --fullorder [config3] [config5]
# ---------------------------------------------------------------------------------------------------------
def main_fullorder(which):
    import time
    if "config3" in which:
        # the throughput stack of config 3 (bench.py:make_inputs_stack): four 200 nm layers, the 180 x 100 rectangle rotated by
        # 0 / 30 / 60 / 90 degrees in SU-8, glass input; one wavelength of the 64-point sweep (index 33 of the 128-entry table)
        t0 = time.time()
        g32 = ref_geometry(300, 300, 300., 300., torch.float32)
        lam = float(np.linspace(400., 700., 128)[66])
        lays = []
        for th in (0., 30., 60., 90.):
            r = g32.rectangle(Wx=180., Wy=100., Cx=150., Cy=150., theta=th / 180 * np.pi)
            lays.append((200., _grid_c64(r, _eps_si_c64(lam), 1.6 ** 2), 1.0))
        run_case("config3_o21_l%d" % int(lam), freq=1 / lam, order=[21, 21], L=[300., 300.], dtype="c128f32", tag="c128f32",
                 layers=lays, eps_in=1.46 ** 2, extra=lambda sim, lam=lam: {"lam": np.float64(lam)})
        print("config3 at [21,21]: %.0f s" % (time.time() - t0))
    if "config5" in which:
        t0 = time.time()
        main_config5(order=(25, 25), name="config5_o25_c128")
        print("config5 at [25,25]: %.0f s" % (time.time() - t0))


if __name__ == "__main__" and "--fullorder" in sys.argv:
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", "8")))
    main_fullorder([a for a in sys.argv[1:] if a.startswith("config")] or ["config3", "config5"])


# ---------------------------------------------------------------------------------------------------------
# geometry / return_layer / material-table golden vectors
# ---------------------------------------------------------------------------------------------------------
def main_geometry():
    out = {}
    kw = dict(Lx=320., Ly=210., nx=48, ny=40, edge_sharpness=37.)
    calls = [("circle", dict(R=60., Cx=150., Cy=100.)), ("ellipse", dict(Rx=90., Ry=40., Cx=170., Cy=90., theta=0.4)),
             ("square", dict(W=110., Cx=140., Cy=120., theta=0.2)), ("rectangle", dict(Wx=180., Wy=70., Cx=160., Cy=105., theta=-0.7)),
             ("rhombus", dict(Wx=200., Wy=120., Cx=150., Cy=100., theta=0.3)),
             ("super_ellipse", dict(Wx=190., Wy=100., Cx=155., Cy=95., theta=0.5, power=3.))]
    for dt, tag in ((torch.float64, "f64"), (torch.float32, "f32")):
        g = torcwa.geometry(dtype=dt, device=torch.device("cpu"), **kw)
        g.grid()
        out[f"x_{tag}"], out[f"y_{tag}"] = g.x.numpy(), g.y.numpy()
        shapes = {}
        for nm, a in calls:
            shapes[nm] = getattr(g, nm)(**a)
            out[f"inst_{nm}_{tag}"] = shapes[nm].numpy()
        out[f"inst_union_{tag}"] = g.union(shapes["circle"], shapes["rectangle"]).numpy()
        out[f"inst_intersection_{tag}"] = g.intersection(shapes["ellipse"], shapes["rhombus"]).numpy()
        out[f"inst_difference_{tag}"] = g.difference(shapes["square"], shapes["circle"]).numpy()
        # legacy class-level API (the notebooks' torcwa.rcwa_geo)
        G = torcwa.rcwa_geo
        G.dtype, G.device = dt, torch.device("cpu")
        G.Lx, G.Ly, G.nx, G.ny, G.edge_sharpness = kw["Lx"], kw["Ly"], kw["nx"], kw["ny"], kw["edge_sharpness"]
        G.grid()
        cs = {}
        for nm, a in calls:
            cs[nm] = getattr(G, nm)(**a)
            out[f"cls_{nm}_{tag}"] = cs[nm].numpy()
        out[f"cls_union_{tag}"] = G.union(cs["circle"], cs["rectangle"]).numpy()
        out[f"cls_intersection_{tag}"] = G.intersection(cs["ellipse"], cs["rhombus"]).numpy()
        out[f"cls_difference_{tag}"] = G.difference(cs["square"], cs["circle"]).numpy()
    np.savez_compressed(os.path.join(HERE, "geometry_shapes.npz"), **out)
    print("geometry_shapes:", len(out), "arrays")

    # return_layer (rcwa.py:264-298) of the asymmetric case (patterned eps AND mu) on a 21 x 17 grid and the default 100 x 100
    z = np.load(os.path.join(HERE, "asym_o32_c128.npz"))
    sim = torcwa.rcwa(freq=float(z["freq"]), order=[3, 2], L=[320., 410.], dtype=torch.complex128, device=torch.device("cpu"))
    sim.add_input_layer(eps=2.1)
    sim.set_incident_angle(inc_ang=0.1, azi_ang=0.2)
    sim.add_layer(thickness=150., eps=torch.from_numpy(z["L0_eps_grid"]), mu=torch.from_numpy(z["L0_mu_grid"]))
    sim.add_layer(thickness=80., eps=2.25)
    o = {}
    for nm, (nx, ny) in (("a", (21, 17)), ("b", (100, 100))):
        e, m = sim.return_layer(0, nx=nx, ny=ny)
        o[f"L0_eps_{nm}"], o[f"L0_mu_{nm}"] = e.numpy(), m.numpy()
    e, m = sim.return_layer(1, nx=21, ny=17)
    o["L1_eps_a"], o["L1_mu_a"] = e.numpy(), m.numpy()
    np.savez_compressed(os.path.join(HERE, "return_layer_asym_o32.npz"), **o)

    # material helper (example/Materials.py:5-52): n+ik at in-range, out-of-range and knot wavelengths, in both dtypes, and the
    # finite-difference derivative its backward uses
    cwd = os.getcwd()
    os.chdir(os.path.join(REF, "example"))
    sys.path.insert(0, os.path.join(REF, "example"))
    import Materials
    lams = np.concatenate([[150., 191.9, 192., 192.003, 192.5, 193., 400.25, 532., 650.123, 998., 998.997, 999., 1200.], np.linspace(200.3, 990.7, 40)])
    nk128, nk64, dn = [], [], []
    for lam in lams:
        w = torch.tensor(float(lam), dtype=torch.float64, requires_grad=True)
        v = Materials.aSiH.apply(w)
        nk128.append(complex(v))
        (v.real * 2.0 + v.imag * 3.0).backward()
        dn.append(float(w.grad))
        nk64.append(complex(Materials.aSiH.apply(torch.tensor(float(lam), dtype=torch.float32))))
    os.chdir(cwd)
    np.savez_compressed(os.path.join(HERE, "material_asih.npz"), lam=lams, nk128=np.array(nk128), nk64=np.array(nk64, dtype=np.complex64),
                        grad_2re_3im=np.array(dn))
    print("material_asih:", len(lams), "wavelengths")


if __name__ == "__main__" and "--geometry" in sys.argv:
    main_geometry()


# ---------------------------------------------------------------------------------------------------------
# Rayleigh anomaly (kz -> 0 of a diffraction order, rcwa.py:1143-1147, 1262): just off and exactly on the anomaly
# ---------------------------------------------------------------------------------------------------------
def main_rayleigh():
    lam_tab = np.load(os.path.join(HERE, "asih_table.npz"))
    eps_si = complex(lam_tab["nk"][-2]) ** 2
    g = ref_geometry(64, 64, 300., 300., torch.float64)
    rect = g.rectangle(Wx=180., Wy=100., Cx=150., Cy=150.)
    eps1 = rect * eps_si + (1. - rect)
    # free-space order (1,0) grazes at lambda = L = 300 nm under normal incidence: Kz0 = sqrt(1 - (lambda/L)^2)
    for tag, lam in (("above", 300.0001), ("below", 299.9999)):
        run_case(f"rayleigh_{tag}", freq=1 / lam, order=[3, 3], L=[300., 300.], dtype="c128", layers=[(300., eps1, 1.0)], eps_in=1.46 ** 2)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sim = run_case("rayleigh_exact", freq=1 / 300., order=[3, 3], L=[300., 300.], dtype="c128", layers=[(300., eps1, 1.0)], eps_in=1.46 ** 2)
    print("exact anomaly: non-finite entries of S11:", int((~torch.isfinite(torch.view_as_real(sim.S[0]))).sum()), "of", sim.S[0].numel() * 2)


    # avoid_Pinv_instability with a threshold below the rounding noise of P P^-1: the reference takes its V = Q W Kz^-1 branch
    # (rcwa.py:1259-1262) on every layer -- the fixture that exercises that branch
    z = np.load(os.path.join(HERE, "asym_o32_c128.npz"))
    lays = [(150., torch.from_numpy(z["L0_eps_grid"]), torch.from_numpy(z["L0_mu_grid"])), (80., 2.25, 1.0), (120., torch.from_numpy(z["L2_eps_grid"]), 1.0)]
    run_case("asym_o32_forceQ", freq=1 / 600., order=[3, 2], L=[320., 410.], dtype="c128", layers=lays, eps_in=2.1, eps_out=1.7,
             inc=20 * np.pi / 180, azi=35 * np.pi / 180, angle_layer="output", avoid=True, max_pinv=1e-17, full_S=True)


if __name__ == "__main__" and "--rayleigh" in sys.argv:
    main_rayleigh()


# ---------------------------------------------------------------------------------------------------------
# shape derivatives: the FoM differentiated through the level-set geometry (example/Example4.ipynb: d|txx|^2 / dR of a cylinder,
# exact and stabilised eigen-gradient; example/Example5.ipynb: d|tyy - txx| / d(Wx, Wy) of a rectangle), at a small order
# ---------------------------------------------------------------------------------------------------------
def main_shape_grad():
    out = {}
    geo = torcwa.geometry(Lx=300., Ly=300., nx=120, ny=120, edge_sharpness=60., dtype=torch.float64, device=torch.device("cpu"))
    geo.grid()
    # cylinder (C4v-symmetric: degenerate mode pairs -- the case the broadened adjoint exists for)
    for tag, stable, bp in (("exact", False, 1e-10), ("bpe-10", True, 1e-10), ("bpnone", True, None)):
        torcwa.Eig.broadening_parameter = bp
        for R0 in (88., 97.):
            R = torch.tensor(R0, dtype=torch.float64, requires_grad=True)
            sim = torcwa.rcwa(freq=1 / 473., order=[3, 3], L=[300., 300.], dtype=torch.complex128, device=torch.device("cpu"), stable_eig_grad=stable)
            sim.add_input_layer(eps=1.46 ** 2)
            sim.set_incident_angle(inc_ang=0., azi_ang=0.)
            m = geo.circle(R=R, Cx=150., Cy=150.)
            sim.add_layer(thickness=600., eps=m * 2.0709 ** 2 + (1. - m))
            sim.solve_global_smatrix()
            txx = sim.S_parameters(orders=[0, 0], direction="forward", port="transmission", polarization="xx", ref_order=[0, 0])
            T = torch.abs(txx) ** 2
            T.backward()
            out[f"circle_{tag}_R{int(R0)}_txx"] = txx.detach().numpy()
            out[f"circle_{tag}_R{int(R0)}_grad"] = R.grad.numpy()
            print("circle", tag, R0, "Txx", float(T), "dTxx/dR", float(R.grad))
    torcwa.Eig.broadening_parameter = 1e-10
    # rectangle, rotated so that no mirror symmetry is left (simple spectrum)
    lam_tab = np.load(os.path.join(HERE, "asih_table.npz"))
    eps_si = complex(lam_tab["nk"][-2]) ** 2
    out["eps_si"] = np.complex128(eps_si)
    for th in (0.0, 0.3):
        W = torch.tensor([180., 100.], dtype=torch.float64, requires_grad=True)
        theta = torch.tensor(th, dtype=torch.float64, requires_grad=True)
        sim = torcwa.rcwa(freq=1 / 532., order=[3, 3], L=[300., 300.], dtype=torch.complex128, device=torch.device("cpu"))
        sim.add_input_layer(eps=1.46 ** 2)
        sim.set_incident_angle(inc_ang=0., azi_ang=0.)
        m = geo.rectangle(Wx=W[0], Wy=W[1], Cx=150., Cy=150., theta=theta)
        sim.add_layer(thickness=250., eps=m * eps_si + (1. - m))
        sim.solve_global_smatrix()
        txx = sim.S_parameters(orders=[0, 0], direction="forward", port="transmission", polarization="xx", ref_order=[0, 0])
        tyy = sim.S_parameters(orders=[0, 0], direction="forward", port="transmission", polarization="yy", ref_order=[0, 0])
        delta = torch.abs(tyy - txx)
        delta.sum().backward()
        key = f"rect_th{int(round(th * 10))}"
        out[key + "_txx"], out[key + "_tyy"] = txx.detach().numpy(), tyy.detach().numpy()
        out[key + "_gradW"], out[key + "_gradtheta"] = W.grad.numpy(), theta.grad.numpy()
        print("rectangle theta", th, "delta", float(delta), "d/dW", W.grad.numpy(), "d/dtheta", float(theta.grad))
    np.savez_compressed(os.path.join(HERE, "shape_grad.npz"), **out)


if __name__ == "__main__" and "--shape-grad" in sys.argv:
    main_shape_grad()
