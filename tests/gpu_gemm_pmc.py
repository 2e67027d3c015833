"""One trx_gemm shape, a few repetitions -- target of the rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE)."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.gpu_gemm_bench import run
run(1922, 1922, 1922, 32, reps=3)
run(1922, 1922, 128, 128, beta=1.0, reps=3)
