"""Config 3 of BASELINE.json (throughput variant): Example1-1 style stack of 4 patterned layers (rectangle
180x100 rotated by 0/30/60/90 deg in SU8, 200 nm each), order [21,21] (n = 3698), lambda sweep, glass input.
Ad-hoc GPU script (not pytest): reports layer-solves/s and the Redheffer share."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torcwa_amd
from torcwa_amd.sweep import asih_eps_table, rectangle_density, solve_stack_sweep

order = int(sys.argv[1]) if len(sys.argv) > 1 else 21
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dev = torch.device('cuda')
lam, eps_si = asih_eps_table()
idx = np.linspace(0, 127, B).round().astype(int)
freq = torch.as_tensor(1.0 / lam[idx], dtype=torch.float64, device=dev)
eps_t = torch.as_tensor(eps_si[idx], dtype=torch.complex64, device=dev)
su8 = 1.6 ** 2
layers = []
for th in (0., 30., 60., 90.):
    d = rectangle_density(300, 300, 300., 300., 180., 100., 150., 150., theta=th / 180 * np.pi, dtype=torch.float32, device=dev)
    layers.append((200., (d[None] * eps_t[:, None, None] + (1. - d[None]) * su8).contiguous()))
eng = torcwa_amd.Engine(device=dev)
torch.cuda.synchronize(); t0 = time.time()
out = solve_stack_sweep(freq, layers, [order, order], [300., 300.], eps_in=1.46 ** 2, dtype=torch.complex64, engine=eng, chunk=chunk,
                        orders=[(0, 0)], polarization="xx", check_info=False)
torch.cuda.synchronize(); dt = time.time() - t0
print(f"config3: order [{order},{order}] n={2*(2*order+1)**2} B={B} chunk={chunk}: {dt:.1f} s -> {4*B/dt:.2f} layer-solves/s; failures={eng.failures()}; "
      f"txx[0]={complex(out[0,0]):.6f} peak mem {torch.cuda.max_memory_allocated()/2**30:.0f} GiB")
