"""One large matrix (config-5 geometry, batch 1): forward time against the QR knobs.  Ad-hoc script (not pytest).
usage: python tests/gpu_single_matrix_knobs.py [order=25]"""
import sys, time, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torcwa_amd
from torcwa_amd._lib import lib

order = [int(sys.argv[1])] * 2 if len(sys.argv) > 1 else [25, 25]
dev = torch.device('cuda')
nx, ny = 700, 300
gen = torch.Generator().manual_seed(333)
rho = torch.rand(nx, ny, generator=gen, dtype=torch.float64)
rho = (rho + torch.flip(rho, dims=[1])) / 2
kx = torch.fft.fftfreq(nx, d=1.0)[:, None]; ky = torch.fft.fftfreq(ny, d=1.0)[None, :]
blur = torch.exp(-2 * (np.pi * 20.0) ** 2 * (kx ** 2 + ky ** 2) / 4)
rho = torch.real(torch.fft.ifft2(torch.fft.fft2(rho) * blur)).clamp(0, 1).to(dev)
eps_si = 12.011610263133004 + 0.525912014756j


def forward():
    sim = torcwa_amd.rcwa(freq=1 / 532., order=order, L=[700., 300.], dtype=torch.complex128, device=dev)
    sim.add_input_layer(eps=1.46 ** 2)
    sim.set_incident_angle(inc_ang=0., azi_ang=0.)
    sim.add_layer(thickness=300., eps=rho * eps_si + (1. - rho))
    sim.solve_global_smatrix()
    return sim.S_parameters(orders=[1, 0], direction='forward', port='transmission', polarization='xx', ref_order=[0, 0])


with torch.no_grad():
    forward()
    torch.cuda.synchronize()
    L = lib()
    for knobs in [dict(), dict(qr_chains=2), dict(qr_chains=3), dict(qr_aed=64), dict(qr_chains=3, qr_aed=64), dict(qr_chains=2, qr_aed=64)]:
        for k, v in knobs.items():
            assert L.tuning(k.encode(), v) == 0
        t0 = time.time()
        t = forward()
        torch.cuda.synchronize()
        dt = time.time() - t0
        print("n=%d %-32s forward %.2f s   txx=%.9f%+.9fj" % (2 * (2 * order[0] + 1) ** 2, knobs, dt, t.real.item(), t.imag.item()), flush=True)
        for k in knobs:
            L.tuning(k.encode(), 0)
