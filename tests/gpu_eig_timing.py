"""Ad-hoc timing of trx_eig on the GPU (not a pytest file)."""
import sys, time
import numpy as np
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torcwa_amd._lib import lib
L = lib()
torch.manual_seed(0)
for (n, batch, dt) in [(242, 8, torch.complex128), (450, 8, torch.complex128), (962, 8, torch.complex128), (1922, 4, torch.complex128), (1922, 4, torch.complex64)]:
    A = torch.randn(batch, n, n, dtype=dt, device='cuda')
    A0 = A.clone()
    w = torch.empty(batch, n, dtype=dt, device='cuda'); V = torch.empty(batch, n, n, dtype=dt, device='cuda')
    info = torch.zeros(batch, dtype=torch.int32, device='cuda')
    code = 1 if dt == torch.complex128 else 0
    nws = L.eig_ws_bytes(code, n, batch)
    ws = torch.empty(nws, dtype=torch.uint8, device='cuda')
    torch.cuda.synchronize(); t0 = time.time()
    rc = L.eig(code, A.data_ptr(), w.data_ptr(), V.data_ptr(), n, batch, info.data_ptr(), ws.data_ptr(), nws, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize(); t1 = time.time()
    res = (A0 @ V - V * w[:, None, :]).abs().max().item() / A0.abs().max().item()
    print(f"n={n} batch={batch} {dt} rc={rc} info={info.tolist()} time={t1-t0:.3f}s per-matrix={(t1-t0)/batch*1e3:.1f}ms resid={res:.2e}", flush=True)
