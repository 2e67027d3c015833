"""trx_eig (batched general complex eigendecomposition) against numpy.linalg.eig.

Eigenpair order/phase is unspecified (as in LAPACK), so the checks are invariants: eigenvalue multiset, residual
||A V - V diag(w)||, unit column norms, conditioning of V comparable to LAPACK's.  Runs through the CPU emulator
(`emu`) and on the GPU (`gpu`).
"""
import numpy as np
import pytest

from tests.backends import BACKENDS, dtcode, get_backend

RNG = np.random.default_rng(99)


def run_eig(be, A):
    batch, n, _ = A.shape
    dtype = A.dtype
    dA = be.dev(A)
    w, V = be.empty((batch, n), dtype), be.empty((batch, n, n), dtype)
    info = be.dev(np.full((batch,), -1, dtype=np.int32))
    nws = be.lib.eig_ws_bytes(dtcode(dtype), n, batch)
    ws = be.empty((nws,), np.uint8)
    rc = be.lib.eig(dtcode(dtype), be.ptr(dA), be.ptr(w), be.ptr(V), n, batch, be.ptr(info), be.ptr(ws), nws, be.stream)
    assert rc == 0
    return be.host(w), be.host(V), be.host(info)


def match_eigs(w, wref):
    """max distance after greedy nearest matching (robust to ordering)."""
    wref = list(wref)
    worst = 0.0
    for z in w:
        j = int(np.argmin([abs(z - r) for r in wref]))
        worst = max(worst, abs(z - wref[j]))
        wref.pop(j)
    return worst


def check(A, w, V, info, tol):
    A128 = A.astype(np.complex128)
    for b in range(A.shape[0]):
        assert info[b] == 0
        scale = np.abs(A128[b]).max() * A.shape[1] ** 0.5
        res = np.abs(A128[b] @ V[b] - V[b] * w[b][None, :]).max() / scale
        assert res < tol, res
        assert np.allclose(np.linalg.norm(V[b], axis=0), 1.0, atol=1e-5 if A.dtype == np.complex64 else 1e-12)
        wref = np.linalg.eigvals(A128[b])
        assert match_eigs(w[b], wref) / np.abs(wref).max() < tol * 100
        assert np.linalg.cond(V[b].astype(np.complex128)) < 1e3 * np.linalg.cond(np.linalg.eig(A128[b])[1])


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype,tol", [(np.complex128, 1e-13), (np.complex64, 5e-6)])
@pytest.mark.parametrize("n", [3, 5, 33, 40, 70, 101])
def test_eig_random(backend, dtype, tol, n):
    if backend == "emu" and n > 70 and dtype == np.complex64:
        pytest.skip("emulator time budget")
    be = get_backend(backend)
    batch = 2
    A = (RNG.standard_normal((batch, n, n)) + 1j * RNG.standard_normal((batch, n, n))).astype(dtype)
    A[1] = A[1] + 3.0 * np.diag(np.arange(n)).astype(dtype)      # second matrix: spread spectrum, early deflations
    w, V, info = run_eig(be, A)
    check(A, w, V, info, tol)


@pytest.mark.parametrize("vec", [0, 1])
@pytest.mark.parametrize("backend", BACKENDS)
def test_eig_degenerate_and_structured(backend, vec):
    """Exactly repeated eigenvalues (symmetric meta-atoms give degenerate mode pairs), a triangular input, and a
    block-diagonal input that deflates in the middle."""
    be = get_backend(backend)
    n = 24
    Q, _ = np.linalg.qr(RNG.standard_normal((n, n)) + 1j * RNG.standard_normal((n, n)))
    lam = np.repeat(RNG.standard_normal(n // 2) + 1j * RNG.standard_normal(n // 2), 2)
    A0 = (Q * lam[None, :]) @ Q.conj().T                       # normal matrix, every eigenvalue double
    A1 = np.triu(RNG.standard_normal((n, n)) + 1j * RNG.standard_normal((n, n)))
    A2 = np.zeros((n, n), dtype=np.complex128)
    A2[:10, :10] = RNG.standard_normal((10, 10))
    A2[10:, 10:] = RNG.standard_normal((14, 14)) + 1j * RNG.standard_normal((14, 14))
    A = np.stack([A0, A1, A2]).astype(np.complex128)
    try:
        assert be.lib.tuning(b"eig_vec", vec) == 0
        w, V, info = run_eig(be, A)
    finally:
        be.lib.tuning(b"eig_vec", 0)
    for b in range(3):
        assert info[b] == 0
        res = np.abs(A[b] @ V[b] - V[b] * w[b][None, :]).max() / np.abs(A[b]).max()
        assert res < 1e-12
        assert match_eigs(w[b], np.linalg.eigvals(A[b])) < 1e-10
    assert np.linalg.cond(V[0]) < 1e6        # degenerate pairs must still give independent eigenvectors


@pytest.mark.parametrize("backend", BACKENDS)
def test_eig_two_iteration_groups(backend):
    """batch >= 8 splits the QR phase into two groups on two streams (uneven halves here); matrices of very different
    convergence speed (nearly diagonal, random, nearly triangular) make one group finish before the other."""
    be = get_backend(backend)
    n, batch = 70, 9
    A = (RNG.standard_normal((batch, n, n)) + 1j * RNG.standard_normal((batch, n, n))).astype(np.complex128)
    for b in range(0, 4):
        A[b] = 1e-3 * A[b] + np.diag(np.arange(1, n + 1)).astype(np.complex128)      # group 0: nearly diagonal
    A[8] = np.triu(A[8]) + 1e-6 * A[7] + np.diag(2.0 * np.arange(n))
    w, V, info = run_eig(be, A)
    check(A, w, V, info, 1e-12)


@pytest.mark.gpu
def test_eig_four_iteration_groups_gpu():
    """batch >= 64: the QR phase runs as four iteration groups on four streams with the 64-wide AED window (the bench shape's
    code path) -- exercised here at a size the test budget allows."""
    be = get_backend("gpu")
    n, batch = 150, 66
    A = (RNG.standard_normal((batch, n, n)) + 1j * RNG.standard_normal((batch, n, n))).astype(np.complex128)
    for b in range(0, batch, 5):
        A[b] = 1e-2 * A[b] + np.diag(np.linspace(-3, 3, n)).astype(np.complex128)      # a few fast-converging members per group
    w, V, info = run_eig(be, A)
    check(A, w, V, info, 1e-12)


def _set_knobs(be, **kw):
    for k, v in kw.items():
        assert be.lib.tuning(k.encode(), int(v)) == 0, k


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("knobs", [dict(qr_groups=3, slab_spw=4), dict(qr_groups=1, slab_spw=1, qr_aed=32), dict(qr_chains=1), dict(qr_chains=2, qr_groups=2), dict(slab_band=1), dict(eig_vec=1), dict(eig_vec=1, qr_groups=3, slab_band=1), dict(qr_chains=2, slab_spw=2), dict(qr_chains=2, qr_fuse=1), dict(qr_chains=3, qr_fuse=2)])
def test_eig_tuning_knobs(backend, knobs):
    """The tuning knobs of the QR phase (iteration groups, strips per wave, AED window, bulge chains per sweep; include/trx.h:
    trx_tuning) select different code paths, not results."""
    if backend == "emu" and knobs in (dict(qr_chains=1), dict(eig_vec=1), dict(eig_vec=1, qr_groups=3, slab_band=1), dict(qr_chains=2, slab_spw=2), dict(qr_chains=3, qr_fuse=2)):
        pytest.skip("emulator time budget: one chain and Schur vectors are the automatic choices at this size, the combinations are covered knob by knob")
    be = get_backend(backend)
    n, batch = (76, 8) if backend == "emu" else (90, 9)          # batch >= 8: two iteration groups by default
    A = (RNG.standard_normal((batch, n, n)) + 1j * RNG.standard_normal((batch, n, n))).astype(np.complex128)
    try:
        _set_knobs(be, **knobs)
        w, V, info = run_eig(be, A)
    finally:
        _set_knobs(be, **{k: 0 for k in knobs})
        _set_knobs(be, qr_nibble=100, qr_moves=12)
    check(A, w, V, info, 1e-12)
    assert be.lib.tuning(b"no_such_knob", 1) != 0 and be.lib.tuning(b"qr_chains", 9) != 0


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("spw", [4, 1])
def test_eig_deferred_right_update(backend, spw):
    """Sweeps of several window steps: the left update runs right behind each step (1 or 4 strips per wave), the right update of H and the
    update of Z once per sweep over the whole link log (apply_links_kernel<1>: every row strip walks through all links whose window lies
    below it).  A matrix that deflates in the middle (active blocks that end anywhere) and a dense one."""
    if backend == "emu" and spw == 1:
        pytest.skip("emulator time budget: one strip per wave is the default of the other emulator tests")
    be = get_backend(backend)
    n = 140 if backend == "emu" else 333
    A = (RNG.standard_normal((3, n, n)) + 1j * RNG.standard_normal((3, n, n))).astype(np.complex128)
    A[2] = 0.2 * A[2] + np.diag(np.linspace(-9, 9, n)).astype(np.complex128)
    A[1][n // 2:, :n // 2] = 0                                     # block upper triangular: two independent active blocks
    try:
        _set_knobs(be, slab_spw=spw, qr_chains=1)             # one chain per sweep: the form with super-steps (automatic above batch 48 only)
        w, V, info = run_eig(be, A)
    finally:
        _set_knobs(be, slab_spw=0, qr_chains=0)
    check(A, w, V, info, 1e-12)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("group", [1, 2, 3, 4])
def test_eig_hessenberg_delayed_right_updates(backend, group):
    """Hessenberg reduction with the right updates of Z and of the rows above a panel delayed over groups of 1 - 4 panels (eig_hess.hip:
    merged block reflector, applied as one rank-32 g update).  group = 1 is the per-panel form of rounds 1 - 5; all four must give the same eigenpairs to
    the solver's accuracy class, in both precisions.  Sizes: several full groups + a ragged last group + a ragged last panel."""
    if backend == "emu" and group in (2, 4):
        pytest.skip("emulator time budget: 4 is the default of every other emulator test, 3 covers the asymmetric merge")
    be = get_backend(backend)
    n = 139 if backend == "emu" else 333          # 139: panels at 0, 32, ..., 128 (5, the last with 9 columns); 333: 11 panels
    A = (RNG.standard_normal((2, n, n)) + 1j * RNG.standard_normal((2, n, n))).astype(np.complex128)
    A[1] = 0.3 * A[1] + np.diag(np.linspace(-5, 5, n)).astype(np.complex128)
    try:
        _set_knobs(be, hess_group=group, eig_vec=1)
        w, V, info = run_eig(be, A)
        fp32_too = backend != "emu" or group != 1      # emulator time budget: the per-panel form in one precision
        if fp32_too:
            w32, V32, info32 = run_eig(be, A.astype(np.complex64))
    finally:
        _set_knobs(be, hess_group=0, eig_vec=0)
    check(A, w, V, info, 1e-13)
    if fp32_too:
        check(A.astype(np.complex64), w32, V32, info32, 5e-6)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("knobs", [dict(qr_super=4), dict(qr_super=8, slab_spw=2), dict(qr_super=1), dict(qr_super=3, slab_band=1), dict(qr_super=4, qr_fuse=1)])
def test_eig_super_steps_fp32(backend, knobs):
    """fp32 QR phase (first stage of the mixed-precision route; complex64 problems under precision="native"): a launch of the window kernel
    takes the chain through up to 8 windows, applying each window's unitary itself to the band of columns the following windows slide
    over; the left update beyond the band is one launch per super-step over its links, the right / Z update one launch per sweep.  Sizes
    with several super-steps per sweep; a spread spectrum (early deflations move the active block) next to a dense matrix."""
    if backend == "emu" and knobs not in (dict(qr_super=4), dict(qr_super=4, qr_fuse=1)):
        pytest.skip("emulator time budget: the default super-step length only (fused launches, and the left update as its own launch)")
    be = get_backend(backend)
    n = 200 if backend == "emu" else 450
    A = (RNG.standard_normal((2, n, n)) + 1j * RNG.standard_normal((2, n, n))).astype(np.complex64)
    A[1] = (0.3 * A[1] + np.diag(np.linspace(-12, 12, n))).astype(np.complex64)
    try:
        _set_knobs(be, qr_chains=1, **knobs)                  # one chain per sweep (automatic above batch 48 only): super-steps need it
        w, V, info = run_eig(be, A)
    finally:
        _set_knobs(be, qr_chains=0, **{k: 0 for k in knobs})
    check(A, w, V, info, 5e-6)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("chains", [3, 2, 1])
def test_eig_multiple_bulge_chains(backend, chains):
    """Active blocks long enough for three bulge chains per sweep (chain c follows chain c-1 one window behind; 48 of the AED
    window's eigenvalues as shifts): a dense random matrix (slow deflation: the chains run the full length many times) and a
    matrix with a spread diagonal (fast deflation: chains are re-planned as the block shrinks).  Same results with 1, 2, 3 chains."""
    if backend == "emu" and chains != 3:
        pytest.skip("emulator time budget: the three-chain path only")
    be = get_backend(backend)
    n = 420 if backend == "gpu" else 232          # emulator: two chains fit (one per 128 rows beyond the first 96)
    A = (RNG.standard_normal((2, n, n)) + 1j * RNG.standard_normal((2, n, n))).astype(np.complex128)
    A[1] = 0.3 * A[1] + np.diag(np.linspace(-20, 20, n)).astype(np.complex128)
    try:
        _set_knobs(be, qr_chains=chains)
        w, V, info = run_eig(be, A)
    finally:
        _set_knobs(be, qr_chains=0)
    check(A, w, V, info, 1e-12)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype,tol", [(np.complex128, 1e-13), (np.complex64, 5e-6)])
@pytest.mark.parametrize("fold,spacing", [(36, 3e-9), (36, 0.0), (70, 0.0)])
def test_eig_large_cluster_of_equal_eigenvalues(backend, dtype, tol, fold, spacing):
    """A normal matrix with a 36- / 70-fold (nearly) equal eigenvalue through the Schur pipeline (knob eig_vec = 1).  Inside such a cluster the
    AED's reordering meets rotation inputs like |f| ~ 1e-100 next to |g| ~ 1e-60: |f|^2 (|f|^2 + |g|^2) underflowed in the fast rotation
    generator, 1 / sqrt(0) = inf, and the whole result was NaN with info = n (rounds 1 - 5; found in round 6 through the fp64 fallback of the
    mixed route).  The generator now rescales by a power of two when the product leaves the fp64 range."""
    be = get_backend(backend)
    n = 100 if backend == "emu" else 300
    rng = np.random.default_rng(5)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
    lam = 2.0 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    lam[:fold] = (1.5 - 0.5j) + spacing * np.arange(fold)
    A = ((Q * lam[None, :]) @ Q.conj().T)[None].astype(dtype)
    try:
        _set_knobs(be, eig_vec=1)
        w, V, info = run_eig(be, A)
    finally:
        _set_knobs(be, eig_vec=0)
    assert info[0] == 0
    res = np.abs(A[0].astype(np.complex128) @ V[0] - V[0] * w[0][None, :]).max() / np.abs(A[0]).max()
    assert res < tol * 10, res
    assert match_eigs(w[0], lam) / np.abs(lam).max() < tol * 100


@pytest.mark.parametrize("backend", BACKENDS)
def test_eig_nonfinite_input_fails_fast(backend):
    """A NaN in the input cannot converge: info reports the failure (LAPACK style) instead of iterating to the sweep limit."""
    be = get_backend(backend)
    n = 80
    A = (RNG.standard_normal((2, n, n)) + 1j * RNG.standard_normal((2, n, n))).astype(np.complex128)
    A[1, 3, 5] = np.nan
    w, V, info = run_eig(be, A)
    assert info[0] == 0 and info[1] > 0
    assert np.abs(A[0] @ V[0] - V[0] * w[0][None, :]).max() < 1e-10


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("seed", [1, 2, 4])
def test_eig_balances_badly_scaled_input(backend, seed):
    """zgebal parity (torch.linalg.eig -> zgeev balances first): A = D A0 D^-1 with D = 2^k, k in [-24, 24], has the eigenvalues
    of the well-scaled A0, but entries spread over 28 orders of magnitude; an unbalanced QR iteration loses the small eigenvalues
    to the norm of A (eps * 2^48 ~ 6e-2), the balanced one recovers them like LAPACK does.  Also: a matrix that is already
    balanced must come through bit-identical scaling (D = I), and a batch mixes both kinds."""
    be = get_backend(backend)
    n = 48
    # own generator: the three seeds are the ones (of 0 .. 7) on which the incrementally maintained row / column sums of round 6 first lost
    # their digits to cancellation (eigenvalues 1e-13 .. 4e-13 off, 3e-10 in the suite's shared stream) before stale sums were recomputed
    rng = np.random.default_rng(seed)
    A0 = rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)) + np.diag(4.0 * np.arange(n))
    k = rng.integers(-24, 25, size=n)
    D = 2.0 ** k
    A = np.stack([(D[:, None] * A0) / D[None, :], A0]).astype(np.complex128)
    w, V, info = run_eig(be, A)
    assert info[0] == 0 and info[1] == 0
    wref, Vref = np.linalg.eig(A0)
    assert match_eigs(w[0], wref) / np.abs(wref).max() < 1e-13
    assert match_eigs(w[1], wref) / np.abs(wref).max() < 1e-13
    # eigenvectors of the scaled matrix: columns of D V0, unit 2-norm (zgebak + normalisation)
    for j in range(n):
        i = int(np.argmin(np.abs(wref - w[0][j])))
        v = D * Vref[:, i]
        v /= np.linalg.norm(v)
        assert abs(abs(np.vdot(v, V[0][:, j])) - 1.0) < 1e-9, j
    assert np.allclose(np.linalg.norm(V[0], axis=0), 1.0, atol=1e-12)
    check(A[1:], w[1:], V[1:], info[1:], 1e-13)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("steps", [3, 2, 1])
def test_eig_mixed_precision_route(backend, steps):
    """Knob eig_vec = 3: fp32 eigendecomposition refined to fp64 by Newton steps (large GEMMs + one LU; eig_refine.hip), the default route
    for n >= 256.  Two steps reach the accuracy class of the all-fp64 pipeline on simple spectra (same 1e-13 residual gate), three on a
    spectrum of 300 exact pairs; one step leaves ~1e-10.  Matrix 1 has a spread spectrum, matrix 2 exactly repeated eigenvalues (pairs: 2 x 2 blocks
    diagonalised in closed form)."""
    be = get_backend(backend)
    n = 70 if backend == "emu" else 600
    A = (RNG.standard_normal((3, n, n)) + 1j * RNG.standard_normal((3, n, n))).astype(np.complex128)
    A[1] = 0.3 * A[1] + np.diag(np.linspace(-9, 9, n)).astype(np.complex128)
    Q, _ = np.linalg.qr(RNG.standard_normal((n, n)) + 1j * RNG.standard_normal((n, n)))
    lam = np.repeat(3.0 * (RNG.standard_normal(n // 2) + 1j * RNG.standard_normal(n // 2)), 2)
    A[2] = (Q * lam[None, :]) @ Q.conj().T                          # normal, every eigenvalue double
    try:
        _set_knobs(be, eig_vec=3, eig_refine=steps)
        w, V, info = run_eig(be, A)
    finally:
        _set_knobs(be, eig_vec=0, eig_refine=0)
    check(A[:2], w[:2], V[:2], info[:2], 1e-13 if steps >= 2 else 3e-8)
    res = np.abs(A[2] @ V[2] - V[2] * w[2][None, :]).max() / np.abs(A[2]).max()
    # one step: first-order accurate in the fp32 start's error (5e-8 seen); two steps (what a complex64 problem gets): the 300 rotated pairs of
    # the all-double spectrum converge with the refreshed fp32 inverse, 2e-12 seen on MI355X; three steps (complex128 callers): 1e-14
    assert info[2] == 0 and res < {3: 1e-12, 2: 1e-11, 1: 2e-7}[steps]
    assert np.linalg.cond(V[2]) < 1e6


@pytest.mark.parametrize("backend", BACKENDS)
def test_eig_mixed_precision_fallback(backend):
    """Coupled clusters: a triple and a ten-fold eigenvalue are diagonalised exactly inside the refinement (small dense solver, clusters of up
    to 32 members; rounds 3 - 5 stopped at 8 and redid the ten-fold case in fp64).  Same result quality either way, no error."""
    be = get_backend(backend)
    n = 40 if backend == "emu" else 300
    Q, _ = np.linalg.qr(RNG.standard_normal((n, n)) + 1j * RNG.standard_normal((n, n)))
    lam = 2.0 * (RNG.standard_normal(n) + 1j * RNG.standard_normal(n))
    lam[5] = lam[17] = lam[29]
    lam2 = lam.copy()
    lam2[:10] = 1.5 - 0.5j
    A = np.stack([(Q * lam[None, :]) @ Q.conj().T, RNG.standard_normal((n, n)) + 1j * RNG.standard_normal((n, n)), (Q * lam2[None, :]) @ Q.conj().T]).astype(np.complex128)
    for sl in (slice(0, 2), slice(2, 3)):            # [triple + random]: resolved in the refinement; [ten-fold]: falls back
        try:
            _set_knobs(be, eig_vec=3)
            w, V, info = run_eig(be, A[sl])
        finally:
            _set_knobs(be, eig_vec=0)
        for b in range(w.shape[0]):
            assert info[b] == 0
            assert np.abs(A[sl][b] @ V[b] - V[b] * w[b][None, :]).max() / np.abs(A[sl][b]).max() < 1e-12
        assert np.linalg.cond(V[0]) < 1e6


@pytest.mark.gpu
def test_eig_two_host_threads_small_batches():
    """Two host threads, each on its own HIP stream, call trx_eig with batches below 8 (one iteration group, run on the CALLER's stream):
    every call borrows a lane (pinned progress summary + events) from the process-wide pool and must hand it back only after the
    look-ahead iteration it queued has drained -- otherwise the other thread's call can pick the lane up and read a stale 'no active
    matrix left' summary, ending its QR phase early with info == 0.  Results are checked on every call."""
    import threading
    import torch
    from torcwa_amd.engine import Engine
    eng = Engine()
    dev = eng.device
    rng = np.random.default_rng(4711)
    mats = [torch.from_numpy((rng.standard_normal((3, n, n)) + 1j * rng.standard_normal((3, n, n))).astype(np.complex128)).to(dev) for n in (70, 101, 88, 120)]
    errors, worst = [], [0.0, 0.0]

    def worker(t):
        try:
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.device(dev), torch.cuda.stream(st):
                for it in range(10):
                    A = mats[(2 * it + t) % len(mats)]
                    w, V = eng.eig(A)
                    res = (torch.linalg.norm(A @ V - V * w[:, None, :], dim=(1, 2)) / torch.linalg.norm(A, dim=(1, 2))).max()
                    worst[t] = max(worst[t], float(res))
                st.synchronize()
        except BaseException as e:      # noqa: BLE001 - reported on the test's thread
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    assert max(worst) < 1e-12, worst


def run_eig_opts(be, A, opts):
    batch, n, _ = A.shape
    dA = be.dev(A)
    w, V = be.empty((batch, n), A.dtype), be.empty((batch, n, n), A.dtype)
    info = be.dev(np.full((batch,), -1, dtype=np.int32))
    nws = be.lib.eig_ws_bytes_opts(dtcode(A.dtype), n, batch, opts)
    assert nws > 0
    ws = be.empty((nws,), np.uint8)
    assert be.lib.eig_opts(dtcode(A.dtype), be.ptr(dA), be.ptr(w), be.ptr(V), n, batch, be.ptr(info), be.ptr(ws), nws, be.stream, opts) == 0
    return be.host(w), be.host(V), be.host(info)


def _residual(A, w, V):
    return max(np.abs(A[b] @ V[b] - V[b] * w[b][None, :]).max() / (np.abs(A[b]).max() * A.shape[1] ** 0.5) for b in range(A.shape[0]))


@pytest.mark.parametrize("backend", BACKENDS)
def test_eig_opts_per_call(backend):
    """trx_eig_opts: route and Newton steps of ONE call (bits 0-3 steps, bits 4-7 route), no process-global knob involved: the mixed route
    with 1 step and with 2 steps give their own residual classes, the workspace size follows the route, invalid option words are refused,
    and a plain trx_eig afterwards still runs the knobs' defaults (all-fp64 for this small batch)."""
    be = get_backend(backend)
    n = 70 if backend == "emu" else 400
    A = (RNG.standard_normal((2, n, n)) + 1j * RNG.standard_normal((2, n, n))).astype(np.complex128)
    mixed1, mixed2, schur = 1 | (3 << 4), 2 | (3 << 4), 1 << 4
    r1 = _residual(A, *run_eig_opts(be, A, mixed1)[:2])
    r2 = _residual(A, *run_eig_opts(be, A, mixed2)[:2])
    rs = _residual(A, *run_eig_opts(be, A, schur)[:2])
    assert 1e-12 < r1 < 3e-8 and r2 < 1e-13 and rs < 1e-13, (r1, r2, rs)
    assert be.lib.eig_ws_bytes_opts(1, n, 2, mixed2) > be.lib.eig_ws_bytes_opts(1, n, 2, schur) == be.lib.eig_ws_bytes(1, n, 2)
    assert be.lib.eig_ws_bytes_opts(1, n, 2, 5) == 0 and be.lib.eig_ws_bytes_opts(1, n, 2, 4 << 4) == 0 and be.lib.eig_ws_bytes_opts(1, n, 2, 1 << 8) == 0
    w, V, info = run_eig(be, A)
    check(A, w, V, info, 1e-13)


@pytest.mark.gpu
def test_eig_opts_two_threads():
    """Two host threads in one process, one solving with the mixed route and ONE Newton step (what a complex64 user gets), the other with
    THREE (complex128 / differentiable path), each on its own stream, through Engine.eig: every call must come out in the residual class
    of its OWN step count.  With the process-global knob of round 3 (`trx_tuning("eig_refine")` before every call) the two overwrote each
    other between the knob call and trx_eig."""
    import threading
    import torch
    from torcwa_amd.engine import Engine
    eng = Engine()
    dev = eng.device
    rng = np.random.default_rng(815)
    n, batch = 300, 8                               # batch >= 8 and n >= 256: the automatic route is the mixed one
    A = torch.from_numpy((rng.standard_normal((batch, n, n)) + 1j * rng.standard_normal((batch, n, n))).astype(np.complex128)).to(dev)
    errors, res = [], [[], []]

    def worker(t):
        try:
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.device(dev), torch.cuda.stream(st):
                for _ in range(6):
                    w, V = eng.eig(A, refine_steps=1 if t == 0 else 3)
                    r = (torch.linalg.norm(A @ V - V * w[:, None, :], dim=(1, 2)) / torch.linalg.norm(A, dim=(1, 2))).max()
                    res[t].append(float(r))
                st.synchronize()
        except BaseException as e:      # noqa: BLE001 - reported on the test's thread
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
    assert all(1e-12 < r < 1e-6 for r in res[0]), res[0]        # one step: first-order accurate in the fp32 start
    assert all(r < 1e-13 for r in res[1]), res[1]               # three steps: the all-fp64 class


@pytest.mark.parametrize("backend", BACKENDS)
def test_eig_partial_fallback_is_per_matrix(backend):
    """A batch in which ONE matrix has a 36-fold eigenvalue (beyond the refinement's exact cluster treatment): the mixed route keeps its
    refined results for the other matrices and redoes only the flagged one in fp64, as a compact sub-batch inside the same workspace
    (trx_eig_last_fallback = number of matrices redone, on the calling thread).  No host-side route memory: the same call sequence in any
    order gives the same answers, and a batch without hard matrices reports 0 right after one with."""
    from tests.test_pipeline import make_engine
    import torch
    eng = make_engine(backend)
    be = get_backend(backend)
    n = 40 if backend == "emu" else 300
    rng = np.random.default_rng(5)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)) + 1j * rng.standard_normal((n, n)))
    lam = 2.0 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
    # beyond the exact treatment (clusters of up to 32 members): 36 eigenvalues that the fp32 stage cannot tell apart -- exactly equal at the
    # emulator's size, 3e-9 apart on the GPU (distinct in fp64: the fp64 pipeline that redoes the matrix sees a simple spectrum)
    lam[:36] = (1.5 - 0.5j) + (0.0 if n == 40 else 3e-9) * np.arange(36)
    hard = (Q * lam[None, :]) @ Q.conj().T
    A = (rng.standard_normal((6, n, n)) + 1j * rng.standard_normal((6, n, n))).astype(np.complex128)
    A[4] = hard                                         # 1 of 6 flagged: sub-batch of one
    A[1] = 0.3 * A[1] + np.diag(np.linspace(-9, 9, n))
    try:
        _set_knobs(be, eig_vec=3)                    # mixed wherever n >= 8 (the automatic route needs n >= 256 and batch >= 8)
        w, V, info = run_eig(be, A)
        assert be.lib.eig_last_fallback() == 1
        for b in range(6):
            assert info[b] == 0
            assert np.abs(A[b] @ V[b] - V[b] * w[b][None, :]).max() / np.abs(A[b]).max() < 1e-12, b
            assert np.allclose(np.linalg.norm(V[b], axis=0), 1.0, atol=1e-12)
        w, V, info = run_eig(be, A[:4])              # no hard matrix: nothing redone, nothing remembered
        assert be.lib.eig_last_fallback() == 0
        A2 = A.copy(); A2[0] = hard; A2[2] = hard    # 3 of 6 flagged: more than a third -> the whole batch is redone
        w, V, info = run_eig(be, A2)
        assert be.lib.eig_last_fallback() == 6
        for b in range(6):
            assert info[b] == 0 and np.abs(A2[b] @ V[b] - V[b] * w[b][None, :]).max() / np.abs(A2[b]).max() < 1e-12, b
        # a 20-fold eigenvalue is INSIDE the exact cluster treatment (diagonalised by the small dense solver): nothing is redone
        lam3 = 2.0 * (rng.standard_normal(n) + 1j * rng.standard_normal(n))
        lam3[:20] = -0.7 + 1.1j
        A3 = A.copy()
        A3[4] = (Q * lam3[None, :]) @ Q.conj().T
        w, V, info = run_eig(be, A3)
        assert be.lib.eig_last_fallback() == 0
        for b in range(6):
            assert info[b] == 0 and np.abs(A3[b] @ V[b] - V[b] * w[b][None, :]).max() / np.abs(A3[b]).max() < 1e-12, b
        assert np.linalg.cond(V[4]) < 1e6
        At = torch.from_numpy(A).to(eng.device) if backend == "gpu" else None
        if At is not None:                           # the engine reports the same count and keeps no route state
            eng.eig(At)
            assert eng.last_eig_fallback == 1 and not hasattr(eng, "_eig_route_hint")
    finally:
        _set_knobs(be, eig_vec=0)
