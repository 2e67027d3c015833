"""Ad-hoc GPU diagnostic (not a test): the eigensolver and the LU solve over shapes the suites do not visit -- odd / even n around the block
sizes (32-column panels, 128-column Hessenberg groups, 256-row LU blocks, the 1094 / 1971-row limits of the LDS-resident panels), batches on
both sides of the automatic switches (8: mixed route, 24 / 48 / 64: chains per sweep and iteration groups), both precisions.  Prints one line
per case; exits non-zero on the first residual outside the solver's accuracy class."""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.backends import get_backend, dtcode
from tests.test_eig import run_eig
be = get_backend("gpu")
bad = 0
for n, batch in ((31, 5), (64, 9), (97, 3), (128, 49), (129, 24), (130, 8), (255, 7), (256, 8), (257, 8), (300, 65), (385, 25), (511, 2), (640, 12), (1001, 3), (1100, 9)):
    rng = np.random.default_rng(7 * n + batch)
    A = (rng.standard_normal((batch, n, n)) + 1j * rng.standard_normal((batch, n, n))).astype(np.complex128)
    A[0] = 0.2 * A[0] + np.diag(np.linspace(-7, 7, n))
    for dtype, tol in ((np.complex128, 2e-12), (np.complex64, 2e-5)):
        Ad = A.astype(dtype)
        w, V, info = run_eig(be, Ad)
        A128 = Ad.astype(np.complex128)
        res = max(np.abs(A128[b] @ V[b] - V[b] * w[b][None, :]).max() / (np.abs(A128[b]).max() * n ** 0.5) for b in range(batch))
        ok = (info == 0).all() and res < tol
        bad += not ok
        print("eig", dtype.__name__, "n", n, "batch", batch, "residual %.2e" % res, "info", int(np.abs(info).max()), "OK" if ok else "FAIL", flush=True)
for n, batch, nrhs in ((255, 3, 5), (257, 2, 300), (513, 2, 1026), (1094, 2, 7), (1095, 2, 7), (1500, 2, 64), (1971, 1, 9), (1972, 1, 9), (2500, 1, 33)):
    rng = np.random.default_rng(11 * n)
    for dtype, tol in ((np.complex128, 1e-14), (np.complex64, 5e-6)):
        A = (rng.standard_normal((batch, n, n)) + 1j * rng.standard_normal((batch, n, n))).astype(dtype)
        B = (rng.standard_normal((batch, n, nrhs)) + 1j * rng.standard_normal((batch, n, nrhs))).astype(dtype)
        dA, dB = be.dev(A), be.dev(B)
        piv, info = be.empty((batch, n), np.int32), be.dev(np.full((batch,), -7, dtype=np.int32))
        rc = be.lib.lu_solve(dtcode(dtype), be.ptr(dA), n, be.ptr(dB), nrhs, batch, be.ptr(piv), be.ptr(info), be.stream)
        X = be.host(dB).astype(np.complex128)
        A128 = A.astype(np.complex128)
        berr = max(np.abs(A128[b] @ X[b] - B[b]).max() / (np.abs(A128[b]).sum(axis=1).max() * np.abs(X[b]).max()) for b in range(batch))
        ok = rc == 0 and (be.host(info) == 0).all() and berr < tol
        bad += not ok
        print("lu_solve", dtype.__name__, "n", n, "batch", batch, "nrhs", nrhs, "backward error %.2e" % berr, "OK" if ok else "FAIL", flush=True)
sys.exit(1 if bad else 0)
