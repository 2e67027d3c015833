#!/usr/bin/env python3
"""CPU model (numpy / LAPACK) of the Newton refinement of trx_eig's mixed route (torcwa_amd/csrc/eig_refine.hip), used to choose the
arithmetic of each piece before spending GPU time:  LAPACK-complex64 eigenpairs of a bench operator (the role of the fp32 pipeline), then
  full   per step  G = V^-1 (A V) with an fp64 LU of the CURRENT V, V <- V (I + F) in fp64            (rounds 3-5)
  cheap  per step  R = A V - V diag(lambda) in fp64;  E = V0^-1 R with ONE LU of the fp32 start V0 (fp32 factors, fp32 solves);
                   lambda += diag E;  F = E_ij / (lambda_j - lambda_i);  V += fp32(V) fp32(F) accumulated into fp64
Prints eigen-residual max_j |A v_j - lambda_j v_j| / |A|_F-ish and the eigenvalue error against LAPACK zgeev after every step, plus the
coupling statistics (|G_ij| + |G_ji| > 0.1 |lambda_i - lambda_j|: connected components) the cluster solver has to deal with.

    python profiles/scripts/refine_model.py [--order 15] [--lam 550] [--wx 180 --wy 100] [--theta 0] [--steps 3]
"""
import argparse
import os
import sys
import time

import numpy as np
import scipy.linalg as sla
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import rcwa_oracle as orc  # noqa: E402   (analysis script: test infrastructure only)


def operator(order, lam, wx, wy, theta, bg=1.0):
    from torcwa_amd.materials import asih_nk
    eps_si = complex(asih_nk(torch.tensor([lam], dtype=torch.float64))[0] ** 2)
    dens = orc.rectangle_density(300, 300, 300., 300., wx, wy, 150., 150., theta=theta).to(torch.float32).to(torch.float64)
    grid = (dens * torch.tensor(eps_si, dtype=torch.complex64) + (1. - dens) * bg).to(torch.complex64).to(torch.complex128)
    s = orc.Setup(freq=1.0 / lam, order=[order, order], L=[300., 300.], dtype=torch.complex128, eps_in=1.46 ** 2, has_in=True)
    s = orc.kvectors(s)
    E = orc.conv_matrix(grid, [order, order])
    M = torch.eye(E.shape[0], dtype=torch.complex128)
    P, Q = orc.pq_patterned(E, M, s.kx, s.ky)
    return (P @ Q).numpy()


def resid(A, V, lam, nA):
    R = A @ V - V * lam[None, :]
    return float(np.abs(R).max() / nA), float((np.linalg.norm(R, axis=0) / np.linalg.norm(V, axis=0)).max() / nA)


def match_err(lam, ref):
    ref = ref.copy()
    o = np.argsort(lam.real)
    ro = np.argsort(ref.real)
    # nearest matching on sorted lists is fragile for complex spectra: greedy on a KD-free O(n^2) distance matrix
    D = np.abs(lam[:, None] - ref[None, :])
    return float(D.min(axis=1).max() / np.abs(ref).max())


def components(G, lam, rho=0.1):
    n = G.shape[0]
    C = np.abs(G.real) + np.abs(G.imag)
    C = C + C.T
    gap = np.abs((lam[None, :] - lam[:, None]).real) + np.abs((lam[None, :] - lam[:, None]).imag)
    cp = C > rho * gap
    np.fill_diagonal(cp, False)
    idx = np.nonzero(cp.any(axis=1))[0]
    lab = {int(i): int(i) for i in idx}

    def find(i):
        while lab[i] != i:
            lab[i] = lab[lab[i]]
            i = lab[i]
        return i
    ii, jj = np.nonzero(cp)
    for a, b in zip(ii, jj):
        ra, rb = find(int(a)), find(int(b))
        if ra != rb:
            lab[max(ra, rb)] = min(ra, rb)
    sizes = {}
    for i in idx:
        r = find(int(i))
        sizes[r] = sizes.get(r, 0) + 1
    hist = {}
    for v in sizes.values():
        hist[v] = hist.get(v, 0) + 1
    return len(idx), dict(sorted(hist.items()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--order", type=int, default=15)
    ap.add_argument("--lam", type=float, default=550.)
    ap.add_argument("--wx", type=float, default=180.)
    ap.add_argument("--wy", type=float, default=100.)
    ap.add_argument("--theta", type=float, default=0.)
    ap.add_argument("--bg", type=float, default=1.0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--threads", type=int, default=2)
    ap.add_argument("--no-ref", action="store_true")
    ap.add_argument("--backward", type=float, default=0.0, help="instead: the fp32 start = exact eigenpairs of A + dA, |dA| = this x |A| (backward error model of the GPU's fp32 pipeline)")
    ap.add_argument("--noise", type=float, default=0.0, help="relative random perturbation of the fp32 start (the GPU's fp32 pipeline leaves ~1e-5)")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    t0 = time.time()
    A = operator(a.order, a.lam, a.wx, a.wy, a.theta * np.pi / 180, a.bg)
    A, _ = sla.matrix_balance(A, permute=False)
    n = A.shape[0]
    nA = np.abs(A).max()
    print("n", n, "max|A| %.3e" % nA, "build %.1f s" % (time.time() - t0))
    if a.backward > 0:
        rng = np.random.default_rng(2)
        dA = (rng.standard_normal(A.shape) + 1j * rng.standard_normal(A.shape)) * (a.backward * np.linalg.norm(A, 2) / (2 * np.sqrt(n)))
        w32, V32 = np.linalg.eig(A + dA)
        w32, V32 = w32.astype(np.complex64), V32.astype(np.complex64)
    else:
        w32, V32 = np.linalg.eig(A.astype(np.complex64))
    if a.noise > 0:
        rng = np.random.default_rng(1)
        V32 = V32 + a.noise * np.abs(V32).max() * (rng.standard_normal(V32.shape) + 1j * rng.standard_normal(V32.shape)).astype(np.complex64) / np.sqrt(n)
        w32 = w32 * (1 + a.noise * (rng.standard_normal(n) + 1j * rng.standard_normal(n))).astype(np.complex64)
    V32 = (V32 / np.linalg.norm(V32, axis=0)[None, :]).astype(np.complex64)
    print("fp32 start: cond(V) %.2e" % np.linalg.cond(V32.astype(np.complex128)), "resid (max entry, max col) %.2e %.2e" % resid(A, V32.astype(np.complex128), w32.astype(np.complex128), nA))
    ref = None
    if not a.no_ref:
        ref = np.linalg.eigvals(A)
        print("eigenvalue error of the fp32 start %.2e" % match_err(w32.astype(np.complex128), ref))

    # ---- full (rounds 3-5) ----
    V = V32.astype(np.complex128)
    for it in range(a.steps):
        G = np.linalg.solve(V, A @ V)
        lam = np.diag(G).copy()
        if it == 0:
            print("coupled indices / component-size histogram after the first G:", components(G, lam))
        gap = lam[None, :] - lam[:, None]
        np.fill_diagonal(gap, 1.0)
        F = G / gap
        np.fill_diagonal(F, 0.0)
        print("  full  step %d: max|E| %.2e  max|F| %.2e" % (it + 1, np.abs(G - np.diag(lam)).max(), np.abs(F).max()))
        V = V + V @ F
        V /= np.linalg.norm(V, axis=0)[None, :]
        lam2 = np.einsum("ij,ij->j", V.conj(), A @ V)
        print("  full  step %d: resid %.2e %.2e" % ((it + 1,) + resid(A, V, lam, nA)), "eigval err %.2e" % (match_err(lam, ref) if ref is not None else -1))

    # ---- cheap ----
    for solve_prec, prod_prec in (("f32", "f32"), ("f64-stale", "f32")):
        V = V32.astype(np.complex128)
        lam = w32.astype(np.complex128)
        if solve_prec == "f32":
            lu = sla.lu_factor(V32)
        else:
            lu = sla.lu_factor(V)
        for it in range(a.steps + 1):
            R = A @ V - V * lam[None, :]
            if solve_prec == "f32":
                sc = np.abs(R).max()
                E = sla.lu_solve(lu, (R / sc).astype(np.complex64)).astype(np.complex128) * sc
            else:
                E = sla.lu_solve(lu, R)
            lam = lam + np.diag(E)
            gap = lam[None, :] - lam[:, None]
            np.fill_diagonal(gap, 1.0)
            F = E / gap
            np.fill_diagonal(F, 0.0)
            if prod_prec == "f32":
                V = V + (V.astype(np.complex64) @ F.astype(np.complex64)).astype(np.complex128)
            else:
                V = V + V @ F
            nv = np.linalg.norm(V, axis=0)
            V /= nv[None, :]
            print("  cheap[%s solve, %s product] step %d: max|E| %.2e max|F| %.2e resid %.2e %.2e" % ((solve_prec, prod_prec, it + 1, np.abs(E).max(), np.abs(F).max()) + resid(A, V, lam, nA)),
                  "eigval err %.2e" % (match_err(lam, ref) if ref is not None else -1))


if __name__ == "__main__":
    main()
