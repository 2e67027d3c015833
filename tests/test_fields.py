"""Sources and field maps (torcwa/rcwa.py:526-1112) of the drop-in class against golden vectors from the reference.

Fixtures: tests/golden/fields_*.npz (make_golden.py --fields, --fields-larger: orders [3,3], [3,2], [5,5] and the 6-layer stack of
config 3 at [8,8]): E_i of three sources (plane wave xy/forward, plane wave
ps/backward, Fourier source), field_xz / field_yz on z samples in every region incl. layer boundaries, field_xy in the
input half-space, first/last layer and output half-space.  Tolerance 1e-8 relative to the largest field value
(c128 run vs c128 golden; eigen-decomposition conditioning limits it).
"""
import numpy as np
import pytest
import torch

from tests.backends import BACKENDS
from tests.helpers import GOLDEN, case_inputs, load_case
from tests.test_pipeline import make_engine, run_case

SRCS = {"planewave_xy_f": ("pw", dict(amplitude=[1.0, 0.5j], direction="forward", notation="xy")),
        "planewave_ps_b": ("pw", dict(amplitude=[0.3, 1.0], direction="backward", notation="ps")),
        "fourier_xy_f": ("fo", dict(amplitude=[[1.0, 0.2], [0.1j, 0.4]], orders=[[0, 0], [1, -1]], direction="f", notation="xy"))}


# (fixture, dtype tag of the S-matrix fixture holding its inputs, runs on the CPU emulator): the 6-layer stack at [8,8] (three patterned
# layers of n = 578) takes ten minutes on the emulator -- validated there once (TRX_TEST_SLOW_EMU=1), run on the GPU only
FIELD_CASES = [("example1_o3", "c128", True), ("asym_o32", "c128", True), ("example1_o5", "c128", True), ("config3_o8_l500", "c128f32", False)]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("name,tag,emu_ok", FIELD_CASES, ids=[c[0] for c in FIELD_CASES])
def test_fields_against_reference(backend, name, tag, emu_ok):
    import os
    if backend == "emu" and not emu_ok and not os.environ.get("TRX_TEST_SLOW_EMU"):
        pytest.skip("too slow for the CPU suite (validated on the emulator once, runs on the GPU)")
    eng = make_engine(backend)
    g = load_case(name, tag)
    tol = 1e-8 if emu_ok else 1e-7          # the 6-layer stack passes 1e-8 on the emulator; a decade of slack for another summation order
    f = np.load(os.path.join(GOLDEN, f"fields_{name}.npz"))
    sim = run_case(eng, g, "c128")
    x, y, z = (torch.from_numpy(f[k]) for k in ("x", "y", "z"))
    nl = int(g["n_layers"])
    for sname, (kind, kw) in SRCS.items():
        if kind == "pw":
            sim.source_planewave(**kw)
        else:
            sim.source_fourier(**kw)
        assert np.abs(sim.E_i.cpu().numpy() - f[f"{sname}_Ei"]).max() < 1e-13
        for plane, args in (("xz", (x, z, 133.0)), ("yz", (y, z, 41.0))):
            E, H = getattr(sim, "field_" + plane)(*args)
            got = np.stack([t.cpu().numpy() for t in E + H])
            ref = f[f"{sname}_{plane}"]
            assert got.shape == ref.shape
            assert np.abs(got - ref).max() / np.abs(ref).max() < tol, (sname, plane)
        for ln in (-1, 0, nl - 1, nl):
            key = f"{sname}_xy_L{ln}"
            if key not in f:
                continue
            E, H = sim.field_xy(int(ln), x, y, float(f[key + "_zprop"]))
            got = np.stack([t.cpu().numpy() for t in E + H])
            assert np.abs(got - f[key]).max() / np.abs(f[key]).max() < tol, (sname, ln)


@pytest.mark.parametrize("backend", BACKENDS)
def test_field_argument_checks(backend):
    eng = make_engine(backend)
    g = load_case("fresnel_30", "c128")
    sim = run_case(eng, g, "c128")
    sim.source_planewave(amplitude=[1., 0.])
    with pytest.warns(UserWarning):
        assert sim.field_xz([0.0], torch.zeros(1), 0.0) is None
    with pytest.warns(UserWarning):
        assert sim.field_xy(0.5, torch.zeros(1), torch.zeros(1)) is None
    with pytest.warns(UserWarning):
        assert sim.field_xy(7, torch.zeros(1), torch.zeros(1)) is None
    with pytest.warns(UserWarning):
        sim.source_planewave(direction="sideways")
    assert sim.source_direction == "forward"
