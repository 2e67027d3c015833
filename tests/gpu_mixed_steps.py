"""Ad-hoc GPU diagnostic (not a test): eigen-residuals of the mixed route after 1 .. 4 Newton steps on the three matrices of
tests/test_eig.py::test_eig_mixed_precision_route (n = 600: random, spread spectrum, all-double spectrum)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.backends import get_backend
from tests.test_eig import run_eig, _set_knobs
be = get_backend("gpu")
RNG = np.random.default_rng(99)
n = 600
A = (RNG.standard_normal((3, n, n)) + 1j * RNG.standard_normal((3, n, n))).astype(np.complex128)
A[1] = 0.3 * A[1] + np.diag(np.linspace(-9, 9, n)).astype(np.complex128)
Q, _ = np.linalg.qr(RNG.standard_normal((n, n)) + 1j * RNG.standard_normal((n, n)))
lam = np.repeat(3.0 * (RNG.standard_normal(n // 2) + 1j * RNG.standard_normal(n // 2)), 2)
A[2] = (Q * lam[None, :]) @ Q.conj().T
for steps in (1, 2, 3, 4):
    _set_knobs(be, eig_vec=3, eig_refine=steps)
    w, V, info = run_eig(be, A)
    _set_knobs(be, eig_vec=0, eig_refine=0)
    print(steps, [float(np.abs(A[b] @ V[b] - V[b] * w[b][None, :]).max() / np.abs(A[b]).max()) for b in range(3)], info, "cond", [float(np.linalg.cond(V[b])) for b in range(3)], flush=True)
