"""One trx_gemm shape, a few repetitions -- target of the rocprofv3 --pmc passes.   usage: gpu_gemm_pmc.py [m n k batch [beta]]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_gemm_bench import run
if len(sys.argv) >= 5:
    m, n, k, b = (int(x) for x in sys.argv[1:5])
    run(m, n, k, b, beta=float(sys.argv[5]) if len(sys.argv) > 5 else 0.0, reps=3)
else:
    run(1922, 1922, 1922, 32, reps=3)
    run(1922, 1922, 128, 128, beta=1.0, reps=3)
