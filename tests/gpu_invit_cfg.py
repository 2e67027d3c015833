"""Ad-hoc GPU timing of the inverse-iteration kernel layouts (knob invit_cfg) at the bench shape: n = 1922, batch B (not a pytest file).
usage: python tests/gpu_invit_cfg.py [batch] [n]"""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torcwa_amd._lib import lib
L = lib()
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1922
torch.manual_seed(0)
A0 = torch.randn(batch, n, n, dtype=torch.complex128, device='cuda')
w = torch.empty(batch, n, dtype=torch.complex128, device='cuda'); V = torch.empty(batch, n, n, dtype=torch.complex128, device='cuda')
info = torch.zeros(batch, dtype=torch.int32, device='cuda')
nws = L.eig_ws_bytes(1, n, batch)
ws = torch.empty(nws, dtype=torch.uint8, device='cuda')
L.prof_enable(1)
for cfgs in (sys.argv[3].split(',') if len(sys.argv) > 3 else "0,2,3,4".split(',')):
    parts = [int(v) for v in cfgs.split(':')] + [0, 0]
    cfg, xcd, dbg = parts[0], parts[1], parts[2]
    if dbg:      # the skip bits exist only in a -DTRX_INVIT_DEBUG build of eig_invit.hip (they change results: not a knob of the release library)
        assert L.tuning(b"invit_dbg", dbg) == 0, "rebuild libtrx with -DTRX_INVIT_DEBUG for the timing-experiment bits"
    assert L.tuning(b"invit_cfg", cfg) == 0 and L.tuning(b"eig_vec", 2) == 0 and L.tuning(b"invit_xcd", xcd) == 0
    A = A0.clone()
    L.prof_reset()
    torch.cuda.synchronize(); t0 = time.time()
    rc = L.eig(1, A.data_ptr(), w.data_ptr(), V.data_ptr(), n, batch, info.data_ptr(), ws.data_ptr(), nws, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize(); t1 = time.time()
    buf = (ctypes.c_double * 6)()
    L.prof_get(8, ctypes.addressof(buf))
    res = ((A0[:2] @ V[:2] - V[:2] * w[:2, None, :]).abs().max() / A0[:2].abs().max()).item()
    print(f"cfg {cfg} xcd-knob {xcd} dbg {dbg}: eig {t1-t0:.3f} s, invit_solve {buf[4]:.1f} ms over {int(buf[1])} launch(es) = {8*n**3*batch/buf[4]/1e9:.2f} TF-equivalent fp64 vector; rc={rc} fails={int((info!=0).sum())} resid={res:.2e}", flush=True)
