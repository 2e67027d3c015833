"""Ad-hoc GEMM micro-benchmark on the GPU (not a pytest file): TFLOP/s of trx_gemm for the shapes of the hot path."""
import ctypes, sys, time
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torcwa_amd._lib import lib
L = lib()
dt = torch.complex128
one = (ctypes.c_double * 2)(1.0, 0.0); zero = (ctypes.c_double * 2)(0.0, 0.0)
def run(m, n, k, batch, beta=0.0, reps=5):
    A = torch.randn(batch, m, k, dtype=dt, device='cuda'); B = torch.randn(batch, k, n, dtype=dt, device='cuda'); C = torch.randn(batch, m, n, dtype=dt, device='cuda')
    be = (ctypes.c_double * 2)(beta, 0.0)
    st = torch.cuda.current_stream().cuda_stream
    def call():
        return L.gemm(1, 0, 0, m, n, k, ctypes.addressof(one), A.data_ptr(), k, m * k, B.data_ptr(), n, k * n, ctypes.addressof(be), C.data_ptr(), n, m * n, batch, st)
    call(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): call()
    torch.cuda.synchronize()
    dtm = (time.perf_counter() - t0) / reps
    fl = 8.0 * m * n * k * batch
    by = 16.0 * batch * (m * k + k * n + m * n * (2 if beta else 1))
    print(f"m={m:5d} n={n:5d} k={k:5d} batch={batch:3d} beta={beta}: {dtm*1e3:8.3f} ms  {fl/dtm/1e12:6.1f} TFLOP/s  {by/dtm/1e9:7.0f} GB/s(algorithmic)", flush=True)
if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "hot":          # the large fp64 shapes of the bench step
        run(1922, 1922, 1922, 128)
        run(1666, 1666, 256, 128, beta=1.0)                 # LU trailing update (outer block 256)
        run(1922, 3844, 1922, 128)                          # 2n right-hand sides
        run(961, 961, 961, 128)
        run(4096, 4096, 4096, 4)
        sys.exit(0)
    run(1922, 1922, 1922, 32)
    run(1922, 1922, 1922, 128)
    run(1922, 1922, 32, 128, beta=1.0)
    run(1922, 1922, 64, 128, beta=1.0)
    run(1922, 3844, 32, 128, beta=1.0)
    run(32, 1922, 1922, 128)
    run(1922, 32, 1922, 128)
    run(961, 961, 961, 128)
    run(4096, 4096, 4096, 4)
