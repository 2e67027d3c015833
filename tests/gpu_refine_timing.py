"""Ad-hoc GPU timing (not pytest): cost of a Newton step of the mixed-precision eigensolver at the bench shape."""
import sys, time, os, ctypes
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torcwa_amd._lib import lib
L = lib()
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = 1922
torch.manual_seed(0)
A0 = torch.randn(batch, n, n, dtype=torch.complex128, device='cuda')
w = torch.empty(batch, n, dtype=torch.complex128, device='cuda'); V = torch.empty(batch, n, n, dtype=torch.complex128, device='cuda')
info = torch.zeros(batch, dtype=torch.int32, device='cuda')
for vec, steps in ((1, 0), (3, 1), (3, 2), (3, 3)):
    assert L.tuning(b"eig_vec", vec) == 0 and L.tuning(b"eig_refine", steps) == 0
    nws = L.eig_ws_bytes(1, n, batch)
    ws = torch.empty(nws, dtype=torch.uint8, device='cuda')
    A = A0.clone()
    L.prof_enable(1); L.prof_reset()
    torch.cuda.synchronize(); t0 = time.time()
    rc = L.eig(1, A.data_ptr(), w.data_ptr(), V.data_ptr(), n, batch, info.data_ptr(), ws.data_ptr(), nws, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize(); t1 = time.time()
    res = ((A0[:2] @ V[:2] - V[:2] * w[:2, None, :]).abs().max() / A0[:2].abs().max()).item()
    tags = []
    for tag in range(9):
        buf = (ctypes.c_double * 7)(); L.prof_get(tag, ctypes.addressof(buf))
        if buf[1] > 0: tags.append("%s %.0fms" % (L.prof_tag_name(tag).decode().split('<')[0][:14] + ('NN' if tag == 0 else ''), buf[4] / buf[1] * buf[0]))
    print(f"eig_vec {vec} steps {steps}: {t1-t0:.3f} s resid {res:.2e} fails {int((info!=0).sum())} ws {nws/2**30:.1f} GiB | " + ", ".join(tags), flush=True)
    del ws
