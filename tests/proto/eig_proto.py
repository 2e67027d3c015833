"""numpy prototype of the batched non-Hermitian eigensolver implemented in torcwa_amd/csrc/eig_*.hip.

TEST INFRASTRUCTURE: a readable, slow, single-matrix statement of the exact algorithm the HIP kernels implement
(blocked Householder Hessenberg reduction -> small-bulge multi-shift QR with per-window accumulation -> triangular
eigenvector back-substitution -> back-transform + normalisation), used to validate the algorithm and to cross-check
kernel intermediates in the emulator tests.  Not used by the product.
"""
import numpy as np


# ------------------------------------------------------------------------------------------------------
# Householder reflector (LAPACK zlarfg convention): H = I - tau v v^H,  H^H x = beta e1, v[0] = 1, beta real
# ------------------------------------------------------------------------------------------------------
def larfg(x):
    alpha = x[0]
    xnorm = np.linalg.norm(x[1:])
    if xnorm == 0.0 and alpha.imag == 0.0:
        v = np.zeros_like(x)
        v[0] = 1.0
        return alpha.real, 0.0 + 0.0j, v
    beta = -np.copysign(np.sqrt(abs(alpha) ** 2 + xnorm ** 2), alpha.real)
    tau = complex((beta - alpha.real) / beta, -alpha.imag / beta)
    v = x / (alpha - beta)
    v[0] = 1.0
    return beta, tau, v


def hessenberg_blocked(A, nb=8):
    """A = Q H Q^H.  Returns (H, Q).  Same panel algebra as LAPACK zgehrd/zlahr2."""
    A = A.astype(np.complex128).copy()
    n = A.shape[0]
    Q = np.eye(n, dtype=np.complex128)
    p0 = 0
    while p0 < n - 2:
        ib = min(nb, n - 2 - p0)
        V = np.zeros((n, ib), dtype=np.complex128)
        Y = np.zeros((n, ib), dtype=np.complex128)
        T = np.zeros((ib, ib), dtype=np.complex128)
        R = slice(p0 + 1, n)
        for c in range(ib):
            j = p0 + c
            if c > 0:
                b = A[R, j]
                b = b - Y[R, :c] @ np.conj(V[j, :c])
                w = V[R, :c].conj().T @ b
                w = T[:c, :c].conj().T @ w
                b = b - V[R, :c] @ w
                A[R, j] = b
            beta, tau, v = larfg(A[j + 1:, j].copy())
            V[j + 1:, c] = v
            A[j + 1, j] = beta
            A[j + 2:, j] = 0.0
            y = A[R, j + 1:] @ v                      # the big BLAS-2 stream
            t = V[j + 1:, :c].conj().T @ v
            y = y - Y[R, :c] @ t
            Y[R, c] = tau * y
            T[:c, c] = -tau * (T[:c, :c] @ t)
            T[c, c] = tau
        # top rows of Y
        Y[:p0 + 1, :] = (A[:p0 + 1, p0 + 1:] @ V[p0 + 1:, :]) @ T
        # right update: top rows (panel + trailing columns), then rows R (trailing columns only)
        A[:p0 + 1, p0 + 1:] -= Y[:p0 + 1, :] @ V[p0 + 1:, :].conj().T
        A[R, p0 + ib:] -= Y[R, :] @ V[p0 + ib:, :].conj().T
        # left update of the trailing columns
        W = V[R, :].conj().T @ A[R, p0 + ib:]
        A[R, p0 + ib:] -= V[R, :] @ (T.conj().T @ W)
        # accumulate Q <- Q (I - V T V^H)
        Q[:, R] -= ((Q[:, R] @ V[R, :]) @ T) @ V[R, :].conj().T
        p0 += ib
    return A, Q


# ------------------------------------------------------------------------------------------------------
# Givens-type rotation used by the bulge chase:  G = [[c, s], [-conj(s), c]] (c real) with G @ [f, g]^T = [r, 0]^T
# ------------------------------------------------------------------------------------------------------
def rotg(f, g):
    if g == 0:
        return 1.0, 0.0 + 0.0j, f
    if f == 0:
        return 0.0, np.conj(g) / abs(g), abs(g) + 0.0j
    d = np.hypot(abs(f), abs(g))
    c = abs(f) / d
    s = (f / abs(f)) * np.conj(g) / d
    r = (f / abs(f)) * d
    return c, s, r


def small_schur(H, want_u=True, maxit=30):
    """Single-shift (Wilkinson) QR on a small upper Hessenberg matrix.  Returns (T, U, ok) with H = U T U^H."""
    H = H.astype(np.complex128).copy()
    m = H.shape[0]
    U = np.eye(m, dtype=np.complex128)
    eps = np.finfo(np.float64).eps
    ihi = m - 1
    its = 0
    total = 0
    while ihi > 0:
        # deflation scan
        l = ihi
        while l > 0:
            s = abs(H[l - 1, l - 1].real) + abs(H[l - 1, l - 1].imag) + abs(H[l, l].real) + abs(H[l, l].imag)
            if s == 0:
                s = 1.0
            if abs(H[l, l - 1].real) + abs(H[l, l - 1].imag) <= eps * s:
                H[l, l - 1] = 0.0
                break
            l -= 1
        if l == ihi:
            ihi -= 1
            its = 0
            continue
        its += 1
        total += 1
        if its > maxit:
            return H, U, False
        # shift
        if its % 10 == 0:
            sig = H[ihi, ihi] + 0.75 * abs(H[ihi, ihi - 1].real)
        else:
            a, b, c_, d = H[ihi - 1, ihi - 1], H[ihi - 1, ihi], H[ihi, ihi - 1], H[ihi, ihi]
            tr = 0.5 * (a + d)
            det = (a - tr) * (d - tr) - b * c_
            sq = np.sqrt(-det + 0j)
            e1, e2 = tr + sq, tr - sq
            sig = e1 if abs(e1 - d) < abs(e2 - d) else e2
        # one implicit single-shift sweep on [l, ihi]
        f, g = H[l, l] - sig, H[l + 1, l]
        for p in range(l, ihi):
            if p > l:
                f, g = H[p, p - 1], H[p + 1, p - 1]
            c, s, r = rotg(f, g)
            G = np.array([[c, s], [-np.conj(s), c]])
            lo = max(p - 1, l) if p > l else l
            H[p:p + 2, lo:] = G @ H[p:p + 2, lo:]
            if p > l:
                H[p + 1, p - 1] = 0.0
            hi_row = min(p + 2, ihi) + 1
            H[:hi_row, p:p + 2] = H[:hi_row, p:p + 2] @ G.conj().T
            if want_u:
                U[:, p:p + 2] = U[:, p:p + 2] @ G.conj().T
    return H, U, True


def multishift_qr(H, Z, ns=4, w=16, nmin=12, max_sweeps=None, stats=None):
    """Small-bulge multi-shift QR without AED; windows of size w, chain of ns single-shift bulges spaced 2 apart.
    Updates H (-> upper triangular T) and Z in place.  Returns ok."""
    n = H.shape[0]
    eps = np.finfo(np.float64).eps
    ihi = n - 1
    sweeps = 0
    stall = 0
    if max_sweeps is None:
        max_sweeps = 30 * n
    while ihi >= 0:
        # ---- deflation scan on the subdiagonal (small-subdiagonal criterion, LAPACK zlahqr style)
        if ihi == 0:
            break
        for i in range(ihi, 0, -1):
            s = abs(H[i - 1, i - 1].real) + abs(H[i - 1, i - 1].imag) + abs(H[i, i].real) + abs(H[i, i].imag)
            if s == 0:
                s = 1.0
            if abs(H[i, i - 1].real) + abs(H[i, i - 1].imag) <= eps * s:
                H[i, i - 1] = 0.0
        while ihi > 0 and H[ihi, ihi - 1] == 0:
            ihi -= 1
            stall = 0
        if ihi == 0:
            break
        ilo = ihi
        while ilo > 0 and H[ilo, ilo - 1] != 0:
            ilo -= 1
        m = ihi - ilo + 1
        if m <= nmin:
            T, U, ok = small_schur(H[ilo:ihi + 1, ilo:ihi + 1])
            if not ok:
                return False
            H[ilo:ihi + 1, ilo:ihi + 1] = np.triu(T)
            H[ilo:ihi + 1, ihi + 1:] = U.conj().T @ H[ilo:ihi + 1, ihi + 1:]
            H[:ilo, ilo:ihi + 1] = H[:ilo, ilo:ihi + 1] @ U
            Z[:, ilo:ihi + 1] = Z[:, ilo:ihi + 1] @ U
            ihi = ilo - 1
            stall = 0
            continue
        sweeps += 1
        stall += 1
        if sweeps > max_sweeps:
            return False
        k = min(ns, m // 2)
        # ---- shifts: eigenvalues of the trailing k x k block (exceptional shifts when stalled)
        Tt, _, ok = small_schur(H[ihi - k + 1:ihi + 1, ihi - k + 1:ihi + 1], want_u=False)
        shifts = np.diag(Tt).copy()
        if stall % 6 == 0:
            shifts = shifts + 0.75 * abs(H[ihi, ihi - 1]) * np.exp(2j * np.pi * np.arange(k) / k)
        # ---- chase: global step tau; bulge s sits at p = ilo + tau - 2 s, active while ilo <= p <= ihi-1
        tau = 0
        tau_last = (ihi - 1 - ilo) + 2 * (k - 1)
        while tau <= tau_last:
            w0 = max(ilo, ilo + tau - 2 * (k - 1) - 1)
            w1 = min(w0 + w, ihi + 1)
            if w1 == ihi + 1:
                tau_end = tau_last
            else:
                tau_end = w1 - 3 - ilo
            assert tau_end >= tau, (tau, tau_end, w0, w1)
            ww = w1 - w0
            Hw = H[w0:w1, w0:w1].copy()
            U = np.eye(ww, dtype=np.complex128)
            for t in range(tau, tau_end + 1):
                rots = []
                for s in range(k):
                    p = ilo + t - 2 * s
                    if p < ilo or p > ihi - 1:
                        continue
                    q = p - w0
                    if p == ilo:
                        f, g = Hw[q, q] - shifts[s], Hw[q + 1, q]
                    else:
                        f, g = Hw[q, q - 1], Hw[q + 1, q - 1]
                    c, sn, r = rotg(f, g)
                    rots.append((q, c, sn, p == ilo))
                for (q, c, sn, first) in rots:          # all left rotations (disjoint rows)
                    G = np.array([[c, sn], [-np.conj(sn), c]])
                    lo = q if first else q - 1
                    Hw[q:q + 2, lo:] = G @ Hw[q:q + 2, lo:]
                    if not first:
                        Hw[q + 1, q - 1] = 0.0
                for (q, c, sn, first) in rots:          # all right rotations (disjoint columns)
                    G = np.array([[c, sn], [-np.conj(sn), c]])
                    hi = min(q + 2, ww - 1) + 1
                    Hw[:hi, q:q + 2] = Hw[:hi, q:q + 2] @ G.conj().T
                    U[:, q:q + 2] = U[:, q:q + 2] @ G.conj().T
            H[w0:w1, w0:w1] = Hw
            H[w0:w1, w1:] = U.conj().T @ H[w0:w1, w1:]
            H[:w0, w0:w1] = H[:w0, w0:w1] @ U
            Z[:, w0:w1] = Z[:, w0:w1] @ U
            tau = tau_end + 1
    if stats is not None:
        stats["sweeps"] = sweeps
    return True


def trevc_upper(T):
    """Right eigenvectors X of an upper-triangular T (columns), by back substitution with LAPACK-style
    perturbation of tiny pivots.  T X = X diag(T)."""
    n = T.shape[0]
    eps = np.finfo(np.float64).eps
    smlnum = np.finfo(np.float64).tiny * (n / eps)
    X = np.zeros((n, n), dtype=np.complex128)
    tnorm = np.abs(T).sum(axis=0).max() if n else 0.0
    for k in range(n):
        lam = T[k, k]
        smin = max(eps * (abs(lam.real) + abs(lam.imag)), smlnum, eps * tnorm * 0)
        smin = max(eps * (abs(lam.real) + abs(lam.imag)), smlnum)
        x = np.zeros(n, dtype=np.complex128)
        x[k] = 1.0
        x[:k] = -T[:k, k]
        for i in range(k - 1, -1, -1):
            d = T[i, i] - lam
            if abs(d.real) + abs(d.imag) < smin:
                d = smin + 0j
            x[i] = x[i] / d
            x[:i] -= x[i] * T[:i, i]
        X[:, k] = x
    return X


def eig(A, nb=8, ns=4, w=16, nmin=12, stats=None):
    A = np.asarray(A, dtype=np.complex128)
    H, Q = hessenberg_blocked(A, nb=nb)
    H = np.triu(H, -1)
    Z = Q.copy()
    ok = multishift_qr(H, Z, ns=ns, w=w, nmin=nmin, stats=stats)
    T = np.triu(H)
    X = trevc_upper(T)
    V = Z @ X
    V = V / np.linalg.norm(V, axis=0, keepdims=True)
    return np.diag(T).copy(), V, ok


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for n in (5, 17, 40, 90):
        A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
        H, Q = hessenberg_blocked(A, nb=8)
        print(n, "hess resid", np.abs(Q @ H @ Q.conj().T - A).max(), "below-subdiag", np.abs(np.tril(H, -2)).max(),
              "orth", np.abs(Q.conj().T @ Q - np.eye(n)).max())
        st = {}
        w, V, ok = eig(A, stats=st)
        print("   eig ok", ok, "resid", np.abs(A @ V - V * w).max(), "sweeps", st.get("sweeps"),
              "eigval err", np.abs(np.sort_complex(w) - np.sort_complex(np.linalg.eigvals(A))).max())


# ------------------------------------------------------------------------------------------------------
# Aggressive early deflation (Braman/Byers/Mathias; LAPACK zlaqr3 simplified) -- prototype for eig_qr.hip
# ------------------------------------------------------------------------------------------------------
def _swap_adjacent(T, V, k):
    """Swap T[k,k] and T[k+1,k+1] of an upper-triangular T by one rotation (ztrexc for complex)."""
    a, b = T[k, k], T[k + 1, k + 1]
    c, s, _ = rotg(T[k, k + 1], b - a)
    G = np.array([[c, s], [-np.conj(s), c]])
    T[k:k + 2, k:] = G @ T[k:k + 2, k:]
    T[:k + 2, k:k + 2] = T[:k + 2, k:k + 2] @ G.conj().T
    T[k + 1, k] = 0.0
    V[:, k:k + 2] = V[:, k:k + 2] @ G.conj().T


def aed_step(H, Z, ilo, ihi, nw, max_moves=None):
    """One AED on the trailing nw x nw window of the active block.  Returns (nd, shifts)."""
    n = H.shape[0]
    eps = np.finfo(np.float64).eps
    smlnum = np.finfo(np.float64).tiny * (n / eps)
    nw = min(nw, ihi - ilo + 1)
    kw = ihi - nw + 1
    s = H[kw, kw - 1] if kw > ilo else 0.0
    T, V, ok = small_schur(H[kw:ihi + 1, kw:ihi + 1])
    T = np.triu(T)
    if not ok:
        return 0, np.diag(T).copy()
    ns = nw
    ilst = 0
    moves = 0
    while ilst < ns:
        foo = abs(T[ns - 1, ns - 1].real) + abs(T[ns - 1, ns - 1].imag)
        if foo == 0:
            foo = abs(s)
        spike = abs((s * V[0, ns - 1]).real) + abs((s * V[0, ns - 1]).imag)
        if spike <= max(smlnum, eps * foo):
            ns -= 1                       # deflatable
        else:
            if max_moves is not None and moves >= max_moves:
                break                      # limited reordering: stop at the first undeflatable eigenvalue after max_moves
            for k in range(ns - 2, ilst - 1, -1):     # move it to position ilst
                _swap_adjacent(T, V, k)
            ilst += 1
            moves += 1
    if ns == 0:
        s = 0.0
    nd = nw - ns
    if nd == 0:
        return 0, np.diag(T).copy()       # nothing deflated: H untouched, window eigenvalues are the shifts
    if ns > 1 and s != 0:
        # reflector reducing the spike s*conj(V[0,0:ns]) to a multiple of e1, then Hessenberg restore of T[0:ns,0:ns]
        work = np.conj(V[0, :ns]).copy()
        beta, tau, v = larfg(work)
        Hh = np.eye(ns, dtype=np.complex128) - tau * np.outer(v, v.conj())
        # T <- Hh^H T Hh on the leading ns rows/cols (all nw columns for the left side), V <- V Hh
        T[:ns, :] = Hh.conj().T @ T[:ns, :]
        T[:, :ns] = T[:, :ns] @ Hh
        V[:, :ns] = V[:, :ns] @ Hh
        Th, Qh = hessenberg_blocked(T[:ns, :ns], nb=4)
        T[:ns, ns:] = Qh.conj().T @ T[:ns, ns:]
        T[:ns, :ns] = np.triu(Th, -1)
        V[:, :ns] = V[:, :ns] @ Qh
    if kw > ilo:
        H[kw, kw - 1] = s * np.conj(V[0, 0])
    H[kw:ihi + 1, kw:ihi + 1] = T
    H[kw:ihi + 1, ihi + 1:] = V.conj().T @ H[kw:ihi + 1, ihi + 1:]
    H[:kw, kw:ihi + 1] = H[:kw, kw:ihi + 1] @ V
    Z[:, kw:ihi + 1] = Z[:, kw:ihi + 1] @ V
    return nd, np.diag(T)[:ns].copy()


def multishift_qr_aed(H, Z, ns=4, w=16, nmin=12, nw=None, stats=None, nibble=14, max_moves=None):
    """multishift_qr with an AED step before every sweep."""
    n = H.shape[0]
    eps = np.finfo(np.float64).eps
    nw = nw or (3 * ns) // 2
    ihi = n - 1
    sweeps = aeds = 0
    shift_rows = 0
    stall = 0
    while ihi > 0:
        for i in range(ihi, 0, -1):
            sc = abs(H[i - 1, i - 1].real) + abs(H[i - 1, i - 1].imag) + abs(H[i, i].real) + abs(H[i, i].imag)
            if abs(H[i, i - 1].real) + abs(H[i, i - 1].imag) <= eps * (sc if sc else 1.0):
                H[i, i - 1] = 0.0
        while ihi > 0 and H[ihi, ihi - 1] == 0:
            ihi -= 1
            stall = 0
        if ihi == 0:
            break
        ilo = ihi
        while ilo > 0 and H[ilo, ilo - 1] != 0:
            ilo -= 1
        m = ihi - ilo + 1
        if m <= nmin:
            T, U, ok = small_schur(H[ilo:ihi + 1, ilo:ihi + 1])
            H[ilo:ihi + 1, ilo:ihi + 1] = np.triu(T)
            H[ilo:ihi + 1, ihi + 1:] = U.conj().T @ H[ilo:ihi + 1, ihi + 1:]
            H[:ilo, ilo:ihi + 1] = H[:ilo, ilo:ihi + 1] @ U
            Z[:, ilo:ihi + 1] = Z[:, ilo:ihi + 1] @ U
            ihi = ilo - 1
            continue
        aeds += 1
        nd, shifts = aed_step(H, Z, ilo, ihi, nw, max_moves)
        if nd > 0:
            stall = 0
            if nibble is None or nd * 100 >= nibble * min(nw, m):      # enough deflation: skip the sweep, AED again
                continue
        stall += 1
        # recompute active block after AED (ihi may have moved)
        while ihi > 0 and H[ihi, ihi - 1] == 0:
            ihi -= 1
        ilo = ihi
        while ilo > 0 and H[ilo, ilo - 1] != 0:
            ilo -= 1
        m = ihi - ilo + 1
        if m <= nmin:
            continue
        k = min(ns, m // 2, len(shifts))
        if k < 1:
            continue
        # use the k shifts of smallest |.| distance?  LAPACK takes the trailing ones; keep the last k
        shifts = shifts[-k:]
        if stall % 6 == 0 and stall > 0:
            shifts = shifts + 0.75 * abs(H[ihi, ihi - 1]) * np.exp(2j * np.pi * np.arange(k) / k)
        sweeps += 1
        shift_rows += k * m
        tau = 0
        tau_last = (ihi - 1 - ilo) + 2 * (k - 1)
        while tau <= tau_last:
            w0 = max(ilo, ilo + tau - 2 * (k - 1) - 1)
            w1 = min(w0 + w, ihi + 1)
            tau_end = tau_last if w1 == ihi + 1 else w1 - 3 - ilo
            ww = w1 - w0
            Hw = H[w0:w1, w0:w1].copy()
            U = np.eye(ww, dtype=np.complex128)
            for t in range(tau, tau_end + 1):
                rots = []
                for s_ in range(k):
                    p = ilo + t - 2 * s_
                    if p < ilo or p > ihi - 1:
                        continue
                    q = p - w0
                    if p == ilo:
                        f, g = Hw[q, q] - shifts[s_], Hw[q + 1, q]
                    else:
                        f, g = Hw[q, q - 1], Hw[q + 1, q - 1]
                    c, sn, r = rotg(f, g)
                    rots.append((q, c, sn, p == ilo))
                for (q, c, sn, first) in rots:
                    G = np.array([[c, sn], [-np.conj(sn), c]])
                    lo = q if first else q - 1
                    Hw[q:q + 2, lo:] = G @ Hw[q:q + 2, lo:]
                    if not first:
                        Hw[q + 1, q - 1] = 0.0
                for (q, c, sn, first) in rots:
                    G = np.array([[c, sn], [-np.conj(sn), c]])
                    hi = min(q + 2, ww - 1) + 1
                    Hw[:hi, q:q + 2] = Hw[:hi, q:q + 2] @ G.conj().T
                    U[:, q:q + 2] = U[:, q:q + 2] @ G.conj().T
            H[w0:w1, w0:w1] = Hw
            H[w0:w1, w1:] = U.conj().T @ H[w0:w1, w1:]
            H[:w0, w0:w1] = H[:w0, w0:w1] @ U
            Z[:, w0:w1] = Z[:, w0:w1] @ U
            tau = tau_end + 1
        if sweeps > 30 * n:
            return False
    if stats is not None:
        stats.update(sweeps=sweeps, aeds=aeds, shift_rows=shift_rows)
    return True
