"""Observable API behaviour of the drop-in class that a torcwa script can depend on:
warn-and-fallback on invalid string options, homogeneity dispatch rules, zero-layer quirks, argument validation."""
import numpy as np
import pytest
import torch

from tests.backends import BACKENDS
from tests.test_pipeline import make_engine


def _sim(eng, **kw):
    import torcwa_amd
    return torcwa_amd.rcwa(freq=1 / 500., order=[2, 1], L=[300., 260.], dtype=torch.complex128, engine=eng, **kw)


@pytest.mark.parametrize("backend", BACKENDS)
def test_invalid_strings_warn_and_fall_back(backend):
    import torcwa_amd
    eng = make_engine(backend)
    with pytest.warns(UserWarning):                         # rcwa.py:37-41
        s = torcwa_amd.rcwa(freq=1 / 500., order=[1, 1], L=[300., 300.], dtype=torch.float32, engine=eng)
    assert s._dtype == torch.complex64
    sim = _sim(eng)
    sim.add_input_layer(eps=2.0)
    with pytest.warns(UserWarning):                         # rcwa.py:141
        sim.set_incident_angle(0.1, 0.0, angle_layer="sideways")
    assert sim.angle_layer == "input"
    sim.add_layer(100., 2.5)
    sim.solve_global_smatrix()
    ref = sim.S_parameters([[0, 0]])
    for kw in (dict(direction="up"), dict(port="x"), dict(polarization="zz")):
        with pytest.warns(UserWarning):                     # rcwa.py:325, 333, 337
            v = sim.S_parameters([[0, 0]], **kw)
        assert torch.allclose(v, ref)
    with pytest.warns(UserWarning):                         # rcwa.py:234, 242
        a = sim.diffraction_angle([[0, 0]], layer="middle", unit="grad")
    b = sim.diffraction_angle([[0, 0]], layer="o", unit="r")
    assert torch.allclose(a[0], b[0]) and torch.allclose(a[1], b[1])
    for al in ("f", "forward"):
        assert torch.allclose(sim.S_parameters([[0, 0]], direction=al, port="t"), ref)


@pytest.mark.parametrize("backend", BACKENDS)
def test_homogeneity_dispatch_rules(backend):
    """python float/complex, 0-d tensor and 1-element 1-D tensor are homogeneous; python int is rejected like the
    reference (rcwa.py:156-157 evaluates `eps.dim()` on it)."""
    eng = make_engine(backend)
    vals = []
    for eps in (2.25, complex(2.25, 0.0), torch.tensor(2.25, dtype=torch.float64), torch.tensor([2.25], dtype=torch.float64)):
        sim = _sim(eng)
        sim.add_input_layer(eps=1.5)
        sim.set_incident_angle(0.2, 0.1)
        sim.add_layer(120., eps)
        sim.solve_global_smatrix()
        vals.append(sim.S_parameters([[0, 0], [1, 0]], polarization="pp").cpu().numpy())
        assert torch.allclose(sim.E_eigvec[0], torch.eye(sim.E_eigvec[0].shape[0], dtype=torch.complex128, device=sim.E_eigvec[0].device))
    for v in vals[1:]:
        assert np.abs(v - vals[0]).max() < 1e-12
    sim = _sim(eng)
    sim.set_incident_angle(0., 0.)
    with pytest.raises(AttributeError):
        sim.add_layer(100., 2)
    # a homogeneous layer given as a constant GRID goes through the patterned path and must agree with the analytic one
    sim = _sim(eng)
    sim.add_input_layer(eps=1.5)
    sim.set_incident_angle(0.2, 0.1)
    sim.add_layer(120., torch.full((16, 12), 2.25, dtype=torch.float64, device=eng.device))
    sim.solve_global_smatrix()
    assert np.abs(sim.S_parameters([[0, 0], [1, 0]], polarization="pp").cpu().numpy() - vals[0]).max() < 1e-10


@pytest.mark.parametrize("backend", BACKENDS)
def test_zero_layer_and_shapes(backend):
    eng = make_engine(backend)
    sim = _sim(eng)
    sim.set_incident_angle(0., 0.)
    sim.solve_global_smatrix()
    n = 2 * sim.order_N
    S = sim.S
    assert S[0].shape == (n, n) and S[1].shape == (n,) and S[2].shape == (n,)      # rcwa.py:185-190 (1-D zeros)
    assert torch.allclose(S[0], torch.eye(n, dtype=torch.complex128, device=S[0].device))
    assert sim.order_N == 15 and sim.order_x.tolist() == [-2, -1, 0, 1, 2] and sim.order_y.tolist() == [-1, 0, 1]
    assert sim.Kx_norm.shape == (15, 15) and sim.Vf.shape == (n, n)
    out = sim.S_parameters([[0, 0], [1, 1], [7, 7]])
    assert out.shape == (3,) and out.dtype == torch.complex128


@pytest.mark.parametrize("backend", BACKENDS)
def test_grid_too_small_is_an_error(backend):
    from torcwa_amd._lib import TrxError
    eng = make_engine(backend)
    sim = _sim(eng)
    sim.set_incident_angle(0., 0.)
    with pytest.raises(TrxError):          # needs nx > 2*ox (the reference would raise IndexError at rcwa.py:1199)
        sim.add_layer(100., torch.ones(4, 12, dtype=torch.float64, device=eng.device) * 2.0)


@pytest.mark.parametrize("backend", BACKENDS)
def test_numerical_failure_is_reported(backend):
    """A singular permittivity convolution matrix (eps == 0 everywhere) must surface as an error, not as silent garbage."""
    import torcwa_amd
    eng = make_engine(backend)
    sim = _sim(eng)
    sim.set_incident_angle(0., 0.)
    with pytest.raises(torcwa_amd.NumericalError):
        sim.add_layer(100., torch.zeros(16, 12, dtype=torch.float64, device=eng.device))
