"""numpy prototype of eigenvectors of an upper Hessenberg matrix by inverse iteration, one O(n^2) solve per eigenvalue
with O(n) state: the statement of the algorithm of torcwa_amd/csrc/eig_invit.hip (vectorised here over ALL eigenvalues at once).

TEST INFRASTRUCTURE (prototype + cross-check of kernel intermediates); never imported by the product.

Algorithm, per eigenvalue lam (LAPACK zhsein/zlaein solve the same system with row operations and a stored factor; here the
elimination runs over COLUMNS from the bottom, so that the triangular solve can be interleaved with it and nothing but two
vectors is kept):

    M = H - lam I  (upper Hessenberg).  For j = n-1 .. 1:  the running column q (rows 0..j; "column j" after the operations so
    far) and p = M[0:j+1, j-1] are combined so that row j of the other one vanishes -- with the larger of |q_j|, |p_j| as the
    pivot (a column interchange otherwise):   f = pivot column,  g' = g - (g_j / f_j) f.   f is column j of the triangular
    factor R (M C = R, C = product of the column operations): y_j = b_j / f_j,  b[0:j] -= f[0:j] y_j, and g' carries on.
    x = C y is the O(n) recurrence  x_j = y_j - m_j x_{j-1}  (with the interchanges) run upwards afterwards.
"""
import numpy as np


def start_vector(n, k):
    """deterministic start vector of eigenvalue number k: entries of modulus ~1 with a k-dependent phase pattern, so that equal
    eigenvalues get independent vectors of their common eigenspace"""
    i = np.arange(n, dtype=np.float64)
    ph = 2 * np.pi * np.modf((i + 1.0) * 0.6180339887498949 * (k + 1.0) + 0.137 * (k + 1.0))[0]
    return np.exp(1j * ph)


def invit_all(H, lam, tries=1):
    """Columns X[:, k] ~ eigenvector of the upper Hessenberg H for lam[k].  Returns (X, growth) with growth = ||x|| / ||b||."""
    n = H.shape[0]
    nl = lam.shape[0]
    hn = np.abs(H).sum(axis=0).max()                      # 1-norm
    eps3 = np.finfo(np.float64).eps * hn
    Bm = np.stack([start_vector(n, k) for k in range(nl)])           # [nl, n]
    b = Bm.copy()
    q = np.tile(H[:, n - 1][None, :], (nl, 1)).astype(np.complex128)  # running column, rows 0..n-1
    q[:, n - 1] -= lam
    y = np.zeros((nl, n), dtype=np.complex128)
    mm = np.zeros((nl, n), dtype=np.complex128)
    sw = np.zeros((nl, n), dtype=bool)
    for j in range(n - 1, 0, -1):
        p = np.tile(H[: j + 1, j - 1][None, :], (nl, 1)).astype(np.complex128)
        p[:, j - 1] -= lam
        qq = q[:, : j + 1]
        swap = np.abs(p[:, j].real) + np.abs(p[:, j].imag) > np.abs(qq[:, j].real) + np.abs(qq[:, j].imag)
        f = np.where(swap[:, None], p, qq)
        g = np.where(swap[:, None], qq, p)
        piv = f[:, j].copy()
        tiny = np.abs(piv) < eps3
        piv[tiny] = eps3
        m = g[:, j] / piv
        g = g - m[:, None] * f
        yj = b[:, j] / piv
        y[:, j] = yj
        b[:, :j] -= f[:, :j] * yj[:, None]
        mm[:, j] = m
        sw[:, j] = swap
        q = g[:, :j]
    piv = q[:, 0].copy()
    tiny = np.abs(piv) < eps3
    piv[tiny] = eps3
    y[:, 0] = b[:, 0] / piv
    # x = C_{n-1} ... C_1 y ;  C_j = [interchange of coordinates (j-1, j) if swapped] . (I - m e_j e_{j-1}^T)
    # column op  col_{j-1} <- col_{j-1} - m col_j  is right-multiplication by E = I - m e_j e_{j-1}^T:  (E z)_j = z_j - m z_{j-1}
    # with an interchange first: M P E = ..., so C_j = P_j E_j and x = C_{n-1} (... (C_1 y))
    x = y.copy()
    for j in range(1, n):
        xj1, xj = x[:, j - 1].copy(), x[:, j].copy()
        xj = xj - mm[:, j] * xj1                     # E_j
        s = sw[:, j]
        x[:, j - 1] = np.where(s, xj, xj1)           # P_j
        x[:, j] = np.where(s, xj1, xj)
    growth = np.linalg.norm(x, axis=1) / np.linalg.norm(Bm, axis=1)
    return x.T.copy(), growth


if __name__ == "__main__":
    import sys
    rng = np.random.default_rng(0)
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    from scipy.linalg import hessenberg
    A = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n))
    H, Q = hessenberg(A, calc_q=True)
    lam = np.linalg.eigvals(H)
    X, gr = invit_all(H, lam)
    X /= np.linalg.norm(X, axis=0)
    res = np.linalg.norm(H @ X - X * lam[None, :], axis=0) / np.linalg.norm(H, 2)
    print("n", n, "max residual", res.max(), "min growth", gr.min(), "cond(X)", np.linalg.cond(X))
