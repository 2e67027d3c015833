"""Size-independent properties at BASELINE.json's full sizes (GPU only): the oracle would need minutes per point at
order [15,15] / [21,21], so these check physics invariants of the HIP path instead.

* energy conservation: for a lossless grating (real permittivities) and p- or s-polarised incidence, the power carried
  by all propagating transmitted + reflected orders sums to 1 (ps basis, power_norm=True);
* batch consistency: B sweep points solved in lock-step equal the same points solved one at a time;
* info == 0 (eigensolver converged) at n = 1922 and n = 3698.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def lossless_stack(order, B, lam, engine=None, dtype=torch.complex128):
    import torcwa_amd
    from torcwa_amd.sweep import rectangle_density
    dev = torch.device("cuda")
    dens = rectangle_density(300, 300, 300., 300., 180., 100., 150., 150., dtype=torch.float64, device=dev)
    eps_core = torch.linspace(5.0, 7.0, B, dtype=torch.float64, device=dev)
    grids = dens[None] * eps_core[:, None, None] + (1. - dens[None])
    sim = torcwa_amd.BatchedRCWA(torch.full((B,), 1.0 / lam, dtype=torch.float64), order, [300., 300.], dtype=dtype, keep_coupling=False)
    sim.add_input_layer(eps=2.1)
    sim.set_incident_angle(0.15, 0.3)
    sim.add_layer(220., grids)
    sim.solve_global_smatrix()
    return sim


def power_sum(sim, pol_in):
    orders = [[m, n] for m in range(-2, 3) for n in range(-2, 3)]
    tot = 0.0
    for port in ("transmission", "reflection"):
        for pol_out in ("p", "s"):
            v = sim.S_parameters(orders, direction="forward", port=port, polarization=pol_out + pol_in, ref_order=[0, 0])
            tot = tot + (torch.abs(v.to(torch.complex128)) ** 2).sum(dim=1)
    return tot.cpu().numpy()


@pytest.mark.parametrize("order,B", [([15, 15], 4), ([21, 21], 2)])
def test_energy_conservation_full_size(order, B):
    sim = lossless_stack(order, B, lam=480.0)
    for pol in ("p", "s"):
        p = power_sum(sim, pol)
        assert np.abs(p - 1.0).max() < 1e-8, (order, pol, p)


def test_batch_equals_single_at_order_15():
    B = 3
    simB = lossless_stack([15, 15], B, lam=520.0)
    tB = simB.S_parameters([[0, 0], [1, 0], [0, -1]], polarization="xx").cpu().numpy()
    import torcwa_amd
    from torcwa_amd.sweep import rectangle_density
    dev = torch.device("cuda")
    dens = rectangle_density(300, 300, 300., 300., 180., 100., 150., 150., dtype=torch.float64, device=dev)
    eps_core = torch.linspace(5.0, 7.0, B, dtype=torch.float64)
    b = 1
    sim = torcwa_amd.rcwa(freq=1.0 / 520.0, order=[15, 15], L=[300., 300.], dtype=torch.complex128)
    sim.add_input_layer(eps=2.1)
    sim.set_incident_angle(0.15, 0.3)
    sim.add_layer(220., dens * float(eps_core[b]) + (1. - dens))
    sim.solve_global_smatrix()
    t1 = sim.S_parameters([[0, 0], [1, 0], [0, -1]], polarization="xx").cpu().numpy()
    assert np.abs(t1 - tB[b]).max() < 1e-9


def test_native_c64_precision_is_reference_class():
    """precision='native' (fp32 arithmetic) lands in the accuracy class of the reference's own complex64 run
    (~1e-3 at order 15), precision='high' stays at the 1e-5 gate."""
    import torcwa_amd
    from torcwa_amd.sweep import asih_eps_table, rectangle_density, solve_single_layer_sweep
    dev = torch.device("cuda")
    lam, eps_si = asih_eps_table()
    idx = [10, 64, 120]
    dens = rectangle_density(300, 300, 300., 300., 180., 100., 150., 150., dtype=torch.float32, device=dev)
    eps_t = torch.as_tensor(eps_si[idx], dtype=torch.complex64, device=dev)
    grids = (dens[None] * eps_t[:, None, None] + (1. - dens[None])).contiguous()
    freq = torch.as_tensor(1.0 / lam[idx], dtype=torch.float64, device=dev)
    kw = dict(eps_in=1.46 ** 2, dtype=torch.complex64)
    hi = solve_single_layer_sweep(freq, grids, 300., [15, 15], [300., 300.], precision="high", **kw).cpu().numpy()
    ref = solve_single_layer_sweep(freq, grids.to(torch.complex128), 300., [15, 15], [300., 300.], eps_in=1.46 ** 2, dtype=torch.complex128).cpu().numpy()
    nat = solve_single_layer_sweep(freq, grids, 300., [15, 15], [300., 300.], precision="native", **kw).cpu().numpy()
    assert np.abs(hi - ref).max() / np.abs(ref).max() < 1e-5
    assert np.abs(nat - ref).max() / np.abs(ref).max() < 2e-2
    print("native-c64 rel err:", np.abs(nat - ref).max() / np.abs(ref).max(), " high:", np.abs(hi - ref).max() / np.abs(ref).max())
