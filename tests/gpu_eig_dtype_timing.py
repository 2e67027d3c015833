"""Ad-hoc GPU timing (not pytest): trx_eig in complex64 vs complex128 at the bench shape (n = 1922), batch given on the command line."""
import sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torcwa_amd._lib import lib
L = lib()
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1922
torch.manual_seed(0)
A0 = torch.randn(batch, n, n, dtype=torch.complex128, device='cuda')
for dt, code in ((torch.complex64, 0), (torch.complex128, 1), (torch.complex64, 0)):
    A = A0.to(dt).clone()
    w = torch.empty(batch, n, dtype=dt, device='cuda'); V = torch.empty(batch, n, n, dtype=dt, device='cuda')
    info = torch.zeros(batch, dtype=torch.int32, device='cuda')
    nws = L.eig_ws_bytes(code, n, batch)
    ws = torch.empty(nws, dtype=torch.uint8, device='cuda')
    torch.cuda.synchronize(); t0 = time.time()
    rc = L.eig(code, A.data_ptr(), w.data_ptr(), V.data_ptr(), n, batch, info.data_ptr(), ws.data_ptr(), nws, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize(); t1 = time.time()
    res = ((A0[:2].to(dt) @ V[:2] - V[:2] * w[:2, None, :]).abs().max() / A0[:2].abs().max()).item()
    print(f"{dt}: eig {t1-t0:.3f} s  rc={rc} fails={int((info!=0).sum())} resid={res:.2e}", flush=True)
    del A, w, V, ws
