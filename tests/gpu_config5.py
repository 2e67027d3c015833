"""Config 5 of BASELINE.json (Example6-style topology optimisation step) on the GPU: forward + adjoint at a given order,
complex128, stabilised eig gradient; directional-derivative check of the adjoint gradient.  Ad-hoc script (not pytest)."""
import sys, time
import numpy as np
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torcwa_amd

order = [int(sys.argv[1]), int(sys.argv[1])] if len(sys.argv) > 1 else [25, 25]
dev = torch.device('cuda')
L = [700., 300.]
nx, ny = 700, 300
gen = torch.Generator().manual_seed(333)
rho0 = torch.rand(nx, ny, generator=gen, dtype=torch.float64)
rho0 = (rho0 + torch.flip(rho0, dims=[1])) / 2
# deterministic smooth density: Gaussian blur (radius 20) via FFT, like the notebook
kx = torch.fft.fftfreq(nx, d=1.0)[:, None]; ky = torch.fft.fftfreq(ny, d=1.0)[None, :]
blur = torch.exp(-2 * (np.pi * 20.0) ** 2 * (kx ** 2 + ky ** 2) / 4)
rho0 = torch.real(torch.fft.ifft2(torch.fft.fft2(rho0) * blur)).clamp(0, 1).to(dev)
eps_si = 12.011610263133004 + 0.525912014756j

def fom_of(rho):
    sim = torcwa_amd.rcwa(freq=1 / 532., order=order, L=L, dtype=torch.complex128, device=dev, stable_eig_grad=True)
    sim.add_input_layer(eps=1.46 ** 2)
    sim.set_incident_angle(inc_ang=0., azi_ang=0.)
    sim.add_layer(thickness=300., eps=rho * eps_si + (1. - rho))
    sim.solve_global_smatrix()
    t = [sim.S_parameters(orders=[1, 0], direction='forward', port='transmission', polarization=p, ref_order=[0, 0]) for p in ('xx', 'yx', 'xy', 'yy')]
    return sum(torch.abs(v) ** 2 for v in t).sum()

rho = rho0.clone().requires_grad_(True)
torch.cuda.synchronize(); t0 = time.time()
fom = fom_of(rho)
torch.cuda.synchronize(); t1 = time.time()
fom.backward()
torch.cuda.synchronize(); t2 = time.time()
g = rho.grad
print(f"order {order} n={2*(2*order[0]+1)**2}: forward {t1-t0:.2f}s backward {t2-t1:.2f}s FoM={float(fom):.9e} |grad|={float(g.norm()):.6e} peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB", flush=True)
d = torch.randn(nx, ny, generator=torch.Generator().manual_seed(1), dtype=torch.float64).to(dev)
h = 1e-4
with torch.no_grad():
    fp, fm = fom_of(rho0 + h * d), fom_of(rho0 - h * d)
fd = float(fp - fm) / (2 * h)
ad = float((g * d).sum())
print(f"directional derivative: adjoint {ad:.9e}  central-FD {fd:.9e}  rel.diff {abs(ad-fd)/abs(fd):.2e}")
