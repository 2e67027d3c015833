"""Ad-hoc GPU diagnostic (not a test): error of precision="native" (fp32 arithmetic) at order [15,15] for three wavelengths against the
complex128 result, printed for the current knob environment (TRX_QR_FUSE, TRX_LU_SUB ...)."""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from torcwa_amd.sweep import asih_eps_table, rectangle_density, solve_single_layer_sweep
dev = torch.device("cuda")
lam, eps_si = asih_eps_table()
idx = [10, 64, 120]
dens = rectangle_density(300, 300, 300., 300., 180., 100., 150., 150., dtype=torch.float32, device=dev)
eps_t = torch.as_tensor(eps_si[idx], dtype=torch.complex64, device=dev)
grids = (dens[None] * eps_t[:, None, None] + (1. - dens[None])).contiguous()
freq = torch.as_tensor(1.0 / lam[idx], dtype=torch.float64, device=dev)
ref = solve_single_layer_sweep(freq, grids.to(torch.complex128), 300., [15, 15], [300., 300.], eps_in=1.46 ** 2, dtype=torch.complex128).cpu().numpy()
for rep in range(3):
    nat = solve_single_layer_sweep(freq, grids, 300., [15, 15], [300., 300.], precision="native", eps_in=1.46 ** 2, dtype=torch.complex64).cpu().numpy()
    print({k: os.environ.get(k) for k in ("TRX_QR_FUSE", "TRX_LU_SUB")}, "native abs err per point", np.abs(nat - ref).ravel(), "rel to max", float(np.abs(nat - ref).max() / np.abs(ref).max()), flush=True)
