"""numpy statement of the Newton refinement of torcwa_amd/csrc/eig_refine.hip on the bench operator, used to decide in WHICH PRECISION
each step has to run.  TEST INFRASTRUCTURE (prototype); never imported by the product.

    python tests/proto/refine_proto.py [order] [lambda_nm]

Start: LAPACK cgeev (true single precision, through scipy) eigenpairs of A = P Q of the bench layer (a-Si:H rectangle 180 x 100 nm, 300 nm cell, glass input).
Step in precision p: G = V^-1 (A V), lambda = diag G, pairs with |G_ij| + |G_ji| > 0.1 |lambda_j - lambda_i| are coupled (their connected
components are diagonalised exactly from their block of G), F_ij = G_ij / (lambda_j - lambda_i) elsewhere, V <- V (I + F) R.
Schedules compared: 64,64 (the library default), 32,64 (first step in fp32), 32,64,64, 64 alone, 32 alone.
Reported: eigen-residual max_j ||A v_j - lambda_j v_j|| / (||A||_F ||v_j||), eigenvalue error against LAPACK zgeev (matched greedily),
coupled indices and largest component per step.
"""
import sys
import os

import numpy as np
import scipy.linalg as sl          # numpy.linalg computes complex64 problems in DOUBLE and rounds the result; scipy calls cgeev / cgesv
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import rcwa_oracle as orc          # noqa: E402


def bench_operator(order, lam_nm, grid=300):
    from torcwa_amd.materials import asih_nk
    eps_si = complex(asih_nk(torch.tensor([lam_nm], dtype=torch.float64))[0] ** 2)
    dens = orc.rectangle_density(grid, grid, 300., 300., 180., 100., 150., 150., dtype=torch.float32)
    g = (dens * np.complex64(eps_si) + (1. - dens)).to(torch.complex64).to(torch.complex128)
    s = orc.Setup(freq=1.0 / lam_nm, order=[order, order], L=[300., 300.], dtype=torch.complex128)
    s.eps_in, s.has_in = 1.46 ** 2, True
    orc.kvectors(s)
    E = orc.conv_matrix(g, s.order)
    M = torch.eye(s.N, dtype=torch.complex128)
    P, Q = orc.pq_patterned(E, M, s.kx, s.ky)
    return (P @ Q).numpy()


def components(n, pairs):
    lab = np.arange(n)
    changed = True
    while changed:
        changed = False
        for i, j in pairs:
            m = min(lab[i], lab[j])
            if lab[i] != m or lab[j] != m:
                lab[i] = lab[j] = m
                changed = True
    return lab


def newton_step(A, V, prec):
    ct = np.complex64 if prec == 32 else np.complex128
    A_, V_ = A.astype(ct), V.astype(ct)
    G = sl.solve(V_, A_ @ V_)
    assert G.dtype == ct
    lam = np.diag(G).copy()
    n = G.shape[0]
    a1 = np.abs(G.real) + np.abs(G.imag)
    d = lam[None, :] - lam[:, None]
    gap = np.abs(d.real) + np.abs(d.imag)
    coupled = (a1 + a1.T) > 0.1 * gap
    np.fill_diagonal(coupled, False)
    idx = np.nonzero(coupled.any(axis=1))[0]
    R = np.eye(n, dtype=ct)
    incl = np.zeros((n, n), dtype=bool)
    big = 0
    if len(idx):
        sub = coupled[np.ix_(idx, idx)]
        pairs = [(p, q) for p, q in zip(*np.nonzero(np.triu(sub)))]
        lab = components(len(idx), pairs)
        for rep in np.unique(lab):
            mem = idx[lab == rep]
            big = max(big, len(mem))
            mu, X = np.linalg.eig(G[np.ix_(mem, mem)].astype(np.complex128))
            X = X / np.linalg.norm(X, axis=0)
            R[np.ix_(mem, mem)] = X.astype(ct)
            lam[mem] = mu.astype(ct)
            incl[np.ix_(mem, mem)] = True
    with np.errstate(divide="ignore", invalid="ignore"):
        F = G / d
    F[incl] = 0
    np.fill_diagonal(F, 0)
    Vn = V_ @ ((np.eye(n, dtype=ct) + F) @ R)
    Vn = Vn / np.linalg.norm(Vn, axis=0)
    eoff = (a1 - np.diag(np.diag(a1))).max()
    return lam.astype(np.complex128), Vn.astype(np.complex128), dict(coupled=len(idx), largest=big, max_offdiag=float(eoff))


def residual(A, lam, V):
    r = A @ V - V * lam[None, :]
    return float((np.linalg.norm(r, axis=0) / np.linalg.norm(V, axis=0)).max() / np.linalg.norm(A))


def eigenvalue_error(lam, ref):
    ref = ref.copy()
    err = 0.0
    for v in lam:                      # greedy nearest match (the spectrum has close pairs, so match and remove)
        k = int(np.argmin(np.abs(ref - v)))
        err = max(err, abs(ref[k] - v) / max(abs(v), 1.0))
        ref[k] = np.inf
    return err


def main():
    order = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    lam_nm = float(sys.argv[2]) if len(sys.argv) > 2 else 400.0
    A = bench_operator(order, lam_nm)
    n = A.shape[0]
    ref = np.linalg.eigvals(A)
    w32, V32 = sl.eig(A.astype(np.complex64))
    assert V32.dtype == np.complex64
    print("order [%d,%d], n = %d, lambda = %.0f nm; |eig| up to %.3g; fp32 start: residual %.2e, eigenvalue error %.2e"
          % (order, order, n, lam_nm, np.abs(ref).max(), residual(A, w32.astype(np.complex128), V32.astype(np.complex128)), eigenvalue_error(w32, ref)))
    for sched in ((64, 64), (32, 64), (32, 64, 64), (64,), (32,), (32, 32, 64)):
        lam, V = w32.astype(np.complex128), V32.astype(np.complex128)
        notes = []
        for p in sched:
            lam, V, info = newton_step(A, V, p)
            notes.append("fp%d: %d coupled (largest %d), max|E| %.1e" % (p, info["coupled"], info["largest"], info["max_offdiag"]))
        # the eigenvalues the NEXT scan would see are diag(V^-1 A V); report the ones the last step left (lam) and the residual of (lam, V)
        print("  steps %-12s residual %.2e  eigenvalue error %.2e | %s" % (",".join(map(str, sched)), residual(A, lam, V), eigenvalue_error(lam, ref), "; ".join(notes)))


if __name__ == "__main__":
    main()
