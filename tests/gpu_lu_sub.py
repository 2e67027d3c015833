"""Ad-hoc GPU diagnostic (not a test): LU factors with the sub-blocked panels (knob lu_sub = 0) against the column-by-column panels (1), fp32 and
fp64, with and without the row-split panel: pivots, factor differences, backward error."""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.backends import get_backend, dtcode
be = get_backend("gpu")
for dtype in (np.complex64, np.complex128):
    for n, batch in ((1922, 2), (961, 2), (700, 3)):
        rng = np.random.default_rng(n)
        A = (rng.standard_normal((batch, n, n)) + 1j * rng.standard_normal((batch, n, n))).astype(dtype)
        B = (rng.standard_normal((batch, n, 4)) + 1j * rng.standard_normal((batch, n, 4))).astype(dtype)
        out = {}
        for sub in (1, 0):
            for split in (1, 0):
                be.lib.tuning(b"lu_sub", sub); be.lib.tuning(b"lu_split", split)
                dA, dB = be.dev(A), be.dev(B)
                piv, info = be.empty((batch, n), np.int32), be.dev(np.full((batch,), -7, dtype=np.int32))
                assert be.lib.lu_solve(dtcode(dtype), be.ptr(dA), n, be.ptr(dB), 4, batch, be.ptr(piv), be.ptr(info), be.stream) == 0
                LU, X, pv = be.host(dA), be.host(dB).astype(np.complex128), be.host(piv)
                A128 = A.astype(np.complex128)
                berr = max(np.abs(A128[b] @ X[b] - B[b]).max() / (np.abs(A128[b]).sum(axis=1).max() * np.abs(X[b]).max()) for b in range(batch))
                out[(sub, split)] = (LU, pv)
                ref = out[(1, 1)]
                print(dtype.__name__, n, "lu_sub", sub, "lu_split", split, "berr %.2e" % berr, "Lmax %.3f" % np.abs(np.tril(LU[0], -1)).max(),
                      "pivots == old one-workgroup:", bool((pv == ref[1]).all()), "factor diff %.2e" % (np.abs(LU - ref[0]).max() / np.abs(ref[0]).max()), flush=True)
be.lib.tuning(b"lu_sub", 0); be.lib.tuning(b"lu_split", 0)
