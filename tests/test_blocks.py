"""Dense building blocks of libtrx (convmat, GEMM, LU solve, inverse) against numpy restatements.

Every test body runs twice: through the CPU kernel-logic emulator (`-m emu`, part of the CPU suite) and on a real
MI355X through torcwa_amd/libtrx.so (`-m gpu`).  Tolerances: c128 ~1e-11..1e-13 rel, c64 ~1e-5..2e-3 rel (fp32
round-off of an n~100 LU), stated per test.
"""
import numpy as np
import pytest

from tests.backends import BACKENDS, dtcode, get_backend

RNG = np.random.default_rng(1234)


def crand(shape, dtype):
    return (RNG.standard_normal(shape) + 1j * RNG.standard_normal(shape)).astype(dtype)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype,tol", [(np.complex128, 1e-13), (np.complex64, 2e-5)])
@pytest.mark.parametrize("opA,opB", [(0, 0), (1, 0), (2, 0), (0, 1), (0, 2), (2, 2)])
@pytest.mark.parametrize("m,n,k", [(70, 67, 37), (70, 20, 37), (20, 150, 37), (32, 129, 64), (100, 32, 16), (65, 130, 3), (130, 70, 100)])
def test_gemm(backend, dtype, tol, opA, opB, m, n, k):
    """(70,67,37): general 64x64 tile with ragged edges; n <= 32: the 64x32 tile; m <= 32: the 32x128 tile; (65,130,3): a single partial
    K slab; (130,70,100): several tiles and slabs."""
    be = get_backend(backend)
    batch = 2
    A = crand((batch, m, k) if opA == 0 else (batch, k, m), dtype)
    B = crand((batch, k, n) if opB == 0 else (batch, n, k), dtype)
    C0 = crand((batch, m, n), dtype)
    al, bt = np.array([0.7 - 0.2j], dtype=dtype), np.array([-0.3 + 0.5j], dtype=dtype)
    dA, dB, dC = be.dev(A), be.dev(B), be.dev(C0)
    rc = be.lib.gemm(dtcode(dtype), opA, opB, m, n, k, al.ctypes.data, be.ptr(dA), A.shape[2], A.shape[1] * A.shape[2],
                     be.ptr(dB), B.shape[2], B.shape[1] * B.shape[2], bt.ctypes.data, be.ptr(dC), n, m * n, batch, be.stream)
    assert rc == 0
    f = {0: lambda x: x, 1: lambda x: x.transpose(0, 2, 1), 2: lambda x: x.conj().transpose(0, 2, 1)}
    ref = al[0] * (f[opA](A).astype(np.complex128) @ f[opB](B).astype(np.complex128)) + bt[0] * C0
    assert np.abs(be.host(dC) - ref).max() / np.abs(ref).max() < tol


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("opA,opB,m,n,k,beta", [
    (0, 0, 2 * 128 + 2, 2 * 96 + 2, 77, 0.0),       # remainders of 2 rows / columns: peeled off for the narrow tiles
    (0, 2, 2 * 128 + 40, 2 * 96 + 50, 64, 1.0),     # remainders above 32: ragged large tiles; beta != 0; K a multiple of the slab
    (2, 0, 330, 259, 131, 1.0),                     # A conjugate-transposed; 17 slabs with a K tail of 3
    (1, 1, 321, 270, 70, 0.0),
])
@pytest.mark.parametrize("dtype,tol", [(np.complex128, 1e-12), (np.complex64, 2e-5)])
def test_gemm_large_tile(backend, opA, opB, m, n, k, beta, dtype, tol):
    """The large tiles against numpy, with the thin-remainder peel of gemm.hip in play: gemm_big.hip (fp64: 128 x 96 on 8 waves, direct-to-LDS
    ring) and gemm_f32_big_kernel (fp32: 128 x 128, a wave owns 64 x 64)."""
    be = get_backend(backend)
    batch = 2
    A = crand((batch, m, k) if opA == 0 else (batch, k, m), dtype)
    B = crand((batch, k, n) if opB == 0 else (batch, n, k), dtype)
    C0 = crand((batch, m, n), dtype)
    al, bt = np.array([0.7 - 0.2j], dtype=dtype), np.array([beta * (-0.3 + 0.5j)], dtype=dtype)
    dA, dB, dC = be.dev(A), be.dev(B), be.dev(C0)
    rc = be.lib.gemm(dtcode(dtype), opA, opB, m, n, k, al.ctypes.data, be.ptr(dA), A.shape[2], A.shape[1] * A.shape[2],
                     be.ptr(dB), B.shape[2], B.shape[1] * B.shape[2], bt.ctypes.data, be.ptr(dC), n, m * n, batch, be.stream)
    assert rc == 0
    f = {0: lambda x: x, 1: lambda x: x.transpose(0, 2, 1), 2: lambda x: x.conj().transpose(0, 2, 1)}
    A128, B128 = A.astype(np.complex128), B.astype(np.complex128)
    ref = al[0].astype(np.complex128) * (f[opA](A128) @ f[opB](B128)) + bt[0].astype(np.complex128) * C0.astype(np.complex128)
    assert np.abs(be.host(dC) - ref).max() / np.abs(ref).max() < tol


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype,tol", [(np.complex128, 1e-11), (np.complex64, 2e-3)])
@pytest.mark.parametrize("n,nrhs", [(5, 3), (32, 32), (33, 7), (97, 130), (150, 20), (290, 40), (530, 16)])
def test_lu_solve(backend, dtype, tol, n, nrhs):
    be = get_backend(backend)
    batch = 2
    A = crand((batch, n, n), dtype)
    A[1, :, 0] *= 1e-3          # force non-trivial pivoting
    B = crand((batch, n, nrhs), dtype)
    dA, dB = be.dev(A), be.dev(B)
    piv, info = be.empty((batch, n), np.int32), be.dev(np.full((batch,), -7, dtype=np.int32))
    rc = be.lib.lu_solve(dtcode(dtype), be.ptr(dA), n, be.ptr(dB), nrhs, batch, be.ptr(piv), be.ptr(info), be.stream)
    assert rc == 0 and (be.host(info) == 0).all()
    X = np.linalg.solve(A.astype(np.complex128), B.astype(np.complex128))
    assert np.abs(be.host(dB) - X).max() / np.abs(X).max() < tol


@pytest.mark.parametrize("backend", BACKENDS)
def test_lu_singular_info(backend):
    be = get_backend(backend)
    n = 40
    A = crand((1, n, n), np.complex128)
    A[0, :, 5] = 0.0
    A[0, 5, :] = 0.0
    dA, dB = be.dev(A), be.dev(crand((1, n, 2), np.complex128))
    piv, info = be.empty((1, n), np.int32), be.empty((1,), np.int32)
    assert be.lib.lu_solve(1, be.ptr(dA), n, be.ptr(dB), 2, 1, be.ptr(piv), be.ptr(info), be.stream) == 0
    assert be.host(info)[0] > 0


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype,tol", [(np.complex128, 1e-11), (np.complex64, 2e-3)])
def test_inverse(backend, dtype, tol):
    be = get_backend(backend)
    n, batch = 45, 3
    A = crand((batch, n, n), dtype)
    dA = be.dev(A)
    piv, info = be.empty((batch, n), np.int32), be.empty((batch,), np.int32)
    nws = be.lib.inverse_ws_bytes(dtcode(dtype), n, batch)
    ws = be.empty((nws,), np.uint8)
    assert be.lib.inverse(dtcode(dtype), be.ptr(dA), n, batch, be.ptr(piv), be.ptr(info), be.ptr(ws), nws, be.stream) == 0
    ref = np.linalg.inv(A.astype(np.complex128))
    assert np.abs(be.host(dA) - ref).max() / np.abs(ref).max() < tol


def _convmat_ref(g, ox, oy):
    """Independent numpy restatement of torcwa/rcwa.py:1183-1204 (fft2 + negative-index gather)."""
    nx, ny = g.shape
    mm, nn = np.meshgrid(np.arange(-ox, ox + 1), np.arange(-oy, oy + 1), indexing="ij")
    m, n_ = mm.reshape(-1), nn.reshape(-1)
    c = np.fft.fft2(g.astype(np.complex128)) / (nx * ny)
    return c[m[:, None] - m[None, :], n_[:, None] - n_[None, :]]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype,tol", [(np.complex128, 1e-13), (np.complex64, 1e-6)])
@pytest.mark.parametrize("cplx", [0, 1])
def test_convmat(backend, dtype, tol, cplx):
    be = get_backend(backend)
    batch, nx, ny, ox, oy = 2, 20, 14, 3, 2
    rdt = np.float64 if dtype == np.complex128 else np.float32
    g = crand((batch, nx, ny), dtype) if cplx else RNG.standard_normal((batch, nx, ny)).astype(rdt)
    N = (2 * ox + 1) * (2 * oy + 1)
    dg, out = be.dev(g), be.empty((batch, N, N), dtype)
    nws = be.lib.convmat_ws_bytes(dtcode(dtype), batch, nx, ny, ox, oy)
    ws = be.empty((nws,), np.uint8)
    assert be.lib.convmat(dtcode(dtype), cplx, be.ptr(dg), batch, nx, ny, ox, oy, be.ptr(out), be.ptr(ws), nws, be.stream) == 0
    o = be.host(out)
    for b in range(batch):
        ref = _convmat_ref(g[b], ox, oy)
        assert np.abs(o[b] - ref).max() / np.abs(ref).max() < tol


@pytest.mark.parametrize("backend", BACKENDS)
def test_convmat_index_map_is_exact(backend):
    """A grid made of a single Fourier harmonic (p,q) gives a convolution matrix that is exactly the indicator of
    m_i-m_j == p, n_i-n_j == q: pins the integer index arithmetic independent of floating point."""
    be = get_backend(backend)
    nx, ny, ox, oy = 16, 12, 2, 1
    x, y = np.meshgrid(np.arange(nx), np.arange(ny), indexing="ij")
    for (p, q) in [(1, 0), (0, -1), (-3, 2), (4, -2)]:
        g = np.exp(2j * np.pi * (p * x / nx + q * y / ny))
        N = (2 * ox + 1) * (2 * oy + 1)
        dg, out = be.dev(g[None]), be.empty((1, N, N), np.complex128)
        nws = be.lib.convmat_ws_bytes(1, 1, nx, ny, ox, oy)
        ws = be.empty((nws,), np.uint8)
        assert be.lib.convmat(1, 1, be.ptr(dg), 1, nx, ny, ox, oy, be.ptr(out), be.ptr(ws), nws, be.stream) == 0
        o = be.host(out)[0]
        mm, nn = np.meshgrid(np.arange(-ox, ox + 1), np.arange(-oy, oy + 1), indexing="ij")
        m, n_ = mm.reshape(-1), nn.reshape(-1)
        expect = ((m[:, None] - m[None, :]) == p) & ((n_[:, None] - n_[None, :]) == q)
        assert (np.abs(o) > 0.5).tolist() == expect.tolist()
        assert np.abs(o - expect).max() < 1e-13


@pytest.mark.parametrize("backend", BACKENDS)
def test_convmat_rejects_small_grid(backend):
    be = get_backend(backend)
    g, out = be.dev(np.zeros((1, 6, 6))), be.empty((1, 49, 49), np.complex128)
    ws = be.empty((1 << 16,), np.uint8)
    assert be.lib.convmat(1, 0, be.ptr(g), 1, 6, 6, 3, 3, be.ptr(out), be.ptr(ws), 1 << 16, be.stream) == -2


def _bd_dense(d):
    """[4 diagonals d11,d12,d21,d22][N] -> dense 2N x 2N block-diagonal operator."""
    return np.block([[np.diag(d[0]), np.diag(d[1])], [np.diag(d[2]), np.diag(d[3])]])


def _star(Sm, Sn):
    """Redheffer star product, the reference's formulas (torcwa/rcwa.py:1287-1296), blocks ordered [S11,S21,S12,S22]."""
    n = Sm[0].shape[0]
    I = np.eye(n)
    t1 = np.linalg.inv(I - Sm[2] @ Sn[1])
    t2 = np.linalg.inv(I - Sn[1] @ Sm[2])
    return [Sn[0] @ t1 @ Sm[0], Sm[1] + Sm[3] @ t2 @ Sn[1] @ Sm[0], Sn[2] + Sn[0] @ t1 @ Sm[2] @ Sn[3], Sm[3] @ t2 @ Sn[3]]


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype,tol", [(np.complex128, 1e-11), (np.complex64, 2e-4)])
@pytest.mark.parametrize("side,want_xy", [(0, 1), (0, 0), (1, 1)])
def test_redheffer_halfspace(backend, dtype, tol, side, want_xy):
    """trx_redheffer_halfspace against the dense star product: both sides, and the lean side-0 path without coupling factors."""
    be = get_backend(backend)
    N, batch = 37, 2
    n = 2 * N
    bd = 0.4 * crand((4, 4, batch, N), dtype)
    S = [0.3 * crand((batch, n, n), dtype) for _ in range(4)]
    dbd, dS = be.dev(bd), [be.dev(x) for x in S]
    out = [be.empty((batch, n, n), dtype) for _ in range(4)]
    XY = be.empty((2, batch, n, 2 * n), dtype) if want_xy else None
    piv, info = be.empty((batch, n), np.int32), be.dev(np.full((batch,), -7, dtype=np.int32))
    nws = be.lib.redheffer_halfspace_ws_bytes(dtcode(dtype), N, batch, side, want_xy)
    ws = be.empty((nws,), np.uint8)
    import ctypes
    arr = ctypes.c_void_p * 4
    ps, po = arr(*[be.ptr(x) for x in dS]), arr(*[be.ptr(x) for x in out])
    rc = be.lib.redheffer_halfspace(dtcode(dtype), side, be.ptr(dbd), ctypes.addressof(ps), ctypes.addressof(po), be.ptr(XY) if want_xy else None,
                                    N, batch, be.ptr(piv), be.ptr(info), be.ptr(ws), nws, be.stream)
    assert rc == 0 and (be.host(info) == 0).all()
    for b in range(batch):
        D = [_bd_dense(bd[k, :, b].astype(np.complex128)) for k in range(4)]
        Sd = [x[b].astype(np.complex128) for x in S]
        ref = _star(D, Sd) if side == 0 else _star(Sd, D)
        for k in range(4):
            assert np.abs(be.host(out[k])[b] - ref[k]).max() / np.abs(ref[k]).max() < tol, (side, want_xy, k)


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype,tol", [(np.complex128, 1e-10), (np.complex64, 5e-4)])
def test_hmodes_structured(backend, dtype, tol):
    """trx_hmodes (rank-N structure of P) against the dense V = inv(P) W diag(kz) of torcwa/rcwa.py:1226-1228, 1248, 1264."""
    be = get_backend(backend)
    N, batch = 41, 2
    n = 2 * N
    E = (crand((batch, N, N), np.complex128) * 0.2 + 3.0 * np.eye(N)).astype(dtype)
    mu = np.array([1.0 + 0.0j, 1.3 - 0.2j]).astype(dtype)
    kx, ky = crand((batch, N), dtype), crand((batch, N), dtype)
    W, kz = crand((batch, n, n), dtype), crand((batch, n), dtype)
    V = be.empty((batch, n, n), dtype)
    piv, info = be.empty((batch, N), np.int32), be.dev(np.full((batch,), -7, dtype=np.int32))
    nws = be.lib.hmodes_ws_bytes(dtcode(dtype), N, batch)
    ws = be.empty((nws,), np.uint8)
    dE, dmu, dkx, dky, dW, dkz = [be.dev(x) for x in (E, mu, kx, ky, W, kz)]          # keep the device buffers alive across the call
    rc = be.lib.hmodes(dtcode(dtype), be.ptr(dE), be.ptr(dmu), be.ptr(dkx), be.ptr(dky), be.ptr(dW), be.ptr(dkz), N, batch, be.ptr(V),
                       be.ptr(piv), be.ptr(info), be.ptr(ws), nws, be.stream)
    assert rc == 0 and (be.host(info) == 0).all()
    for b in range(batch):
        Kx, Ky = np.diag(kx[b].astype(np.complex128)), np.diag(ky[b].astype(np.complex128))
        M = mu[b].astype(np.complex128) * np.eye(N)
        Z = np.zeros((N, N))
        P = np.block([[Z, M], [-M, Z]]) + np.vstack((Kx, Ky)) @ np.linalg.inv(E[b].astype(np.complex128)) @ np.hstack((Ky, -Kx))
        ref = np.linalg.solve(P, W[b].astype(np.complex128) * kz[b].astype(np.complex128)[None, :])
        assert np.abs(be.host(V)[b] - ref).max() / np.abs(ref).max() < tol


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype,tol", [(np.complex128, 1e-11), (np.complex64, 2e-3)])
@pytest.mark.parametrize("n,batch", [(200, 1), (333, 2), (530, 1), (300, 9)])
def test_lu_row_split_panel(backend, dtype, tol, n, batch):
    """The row-split panel (few large matrices: W workgroups per matrix, one launch per panel column, implicit pivoting inside the
    panel) against numpy, and against the one-workgroup panel: same pivots (no exact ties in random data), same factors."""
    be = get_backend(backend)
    rng = np.random.default_rng(4000 + n)          # own generator: the data must not depend on which tests ran before
    A = (rng.standard_normal((batch, n, n)) + 1j * rng.standard_normal((batch, n, n))).astype(dtype)
    A[0, :, 0] *= 1e-3
    A[0, 7, :] *= 50.0              # a row that wins several pivot searches in a row
    B = (rng.standard_normal((batch, n, 9)) + 1j * rng.standard_normal((batch, n, 9))).astype(dtype)
    res = []
    for split in (128, 1):          # 128: split while >= 128 rows remain; 1: never
        assert be.lib.tuning(b"lu_split", split) == 0 and be.lib.tuning(b"lu_split_batch", 16) == 0
        try:
            dA, dB = be.dev(A), be.dev(B)
            piv, info = be.empty((batch, n), np.int32), be.dev(np.full((batch,), -7, dtype=np.int32))
            rc = be.lib.lu_solve(dtcode(dtype), be.ptr(dA), n, be.ptr(dB), 9, batch, be.ptr(piv), be.ptr(info), be.stream)
        finally:
            be.lib.tuning(b"lu_split", 0)
            be.lib.tuning(b"lu_split_batch", 0)
        assert rc == 0 and (be.host(info) == 0).all()
        res.append((be.host(dA), be.host(dB), be.host(piv)))
    X = np.linalg.solve(A.astype(np.complex128), B.astype(np.complex128))
    for LU, Xg, piv in res:
        assert np.abs(Xg - X).max() / np.abs(X).max() < tol
    assert (res[0][2] == res[1][2]).all()
    # same pivots, so the factors agree to rounding: reductions in another order differ by a few n eps (fp32: 1.6e-5 measured at n = 300;
    # fp64: 1.06e-13 at n = 530 on MI355X)
    assert np.abs(res[0][0] - res[1][0]).max() / np.abs(res[1][0]).max() < 20 * n * np.finfo(dtype).eps


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [np.complex128, np.complex64])
def test_lu_panel_sub_blocks(backend, dtype):
    """The one-workgroup panel with LDS-resident sub-blocks of 8 columns (knob lu_sub = 0; 4 columns for panels too tall for 8: forced
    everywhere by lu_sub = 2) against the column-by-column panel of rounds 1 - 5 (lu_sub = 1): same pivots, factors equal to rounding, the
    solve within the backward-error class.  Several outer blocks and a ragged last panel."""
    be = get_backend(backend)
    n, batch = (300, 2) if backend == "emu" else (1500, 3)
    rng = np.random.default_rng(77)
    A = (rng.standard_normal((batch, n, n)) + 1j * rng.standard_normal((batch, n, n))).astype(dtype)
    A[0, :, 5] *= 1e-3
    Bm = (rng.standard_normal((batch, n, 7)) + 1j * rng.standard_normal((batch, n, 7))).astype(dtype)
    res = {}
    for sub in (1, 0, 2):
        assert be.lib.tuning(b"lu_sub", sub) == 0 and be.lib.tuning(b"lu_split", 1) == 0          # never the row-split kernels here
        try:
            dA, dB = be.dev(A), be.dev(Bm)
            piv, info = be.empty((batch, n), np.int32), be.dev(np.full((batch,), -7, dtype=np.int32))
            assert be.lib.lu_solve(dtcode(dtype), be.ptr(dA), n, be.ptr(dB), 7, batch, be.ptr(piv), be.ptr(info), be.stream) == 0
        finally:
            be.lib.tuning(b"lu_sub", 0); be.lib.tuning(b"lu_split", 0)
        assert (be.host(info) == 0).all()
        res[sub] = (be.host(dA), be.host(dB), be.host(piv))
    tol = 20 * n * np.finfo(dtype).eps              # factors of the same pivot sequence: the update order differs, errors grow with n (3.4e-4 seen in fp32 at n = 1500)
    A128, B128 = A.astype(np.complex128), Bm.astype(np.complex128)
    for sub in (0, 2):
        assert (res[sub][2] == res[1][2]).all()
        assert np.abs(res[sub][0] - res[1][0]).max() / np.abs(res[1][0]).max() < tol
    for sub in (1, 0, 2):
        X = res[sub][1].astype(np.complex128)
        berr = np.abs(A128 @ X - B128).max() / (np.abs(A128).max() * np.abs(X).max() * n)
        assert berr < (1e-15 if dtype == np.complex128 else 1e-6), (sub, berr)


@pytest.mark.parametrize("backend", BACKENDS)
def test_lu_row_split_singular_info(backend):
    be = get_backend(backend)
    n = 300
    A = crand((1, n, n), np.complex128)
    A[0, :, 5] = 0.0
    dA, dB = be.dev(A), be.dev(crand((1, n, 2), np.complex128))
    piv, info = be.empty((1, n), np.int32), be.empty((1,), np.int32)
    assert be.lib.tuning(b"lu_split", 128) == 0
    try:
        assert be.lib.lu_solve(1, be.ptr(dA), n, be.ptr(dB), 2, 1, be.ptr(piv), be.ptr(info), be.stream) == 0
    finally:
        be.lib.tuning(b"lu_split", 0)
    assert be.host(info)[0] == 6


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("where", ["first", "last"])
def test_lu_row_split_nan_matrix_stays_in_bounds(backend, where):
    """A NaN-filled matrix in the batch (what the mixed-precision eigensolver's refinement hands to lu_factor when the fp32 stage broke
    down): no row passes the pivot search, the recorded pivots must still be rows of that matrix.  Before the fix piv[col] = n was
    recorded and the interchange kernels swapped with row n: row 0 of the next matrix, or memory behind the last one (guard row here)."""
    be = get_backend(backend)
    n, nrhs, batch = 300, 3, 2
    bad = 0 if where == "first" else batch - 1
    A = crand((batch, n, n), np.complex128)
    A[bad] = np.nan
    B = crand((batch, n, nrhs), np.complex128)
    flatA = np.concatenate([A.reshape(-1), np.full(n, 7.25 + 0j)])          # guard row behind the last matrix
    flatB = np.concatenate([B.reshape(-1), np.full(nrhs, 7.25 + 0j)])
    dA, dB = be.dev(flatA), be.dev(flatB)
    piv, info = be.empty((batch, n), np.int32), be.empty((batch,), np.int32)
    assert be.lib.tuning(b"lu_split", 128) == 0 and be.lib.tuning(b"lu_split_batch", 16) == 0
    try:
        assert be.lib.lu_solve(1, be.ptr(dA), n, be.ptr(dB), nrhs, batch, be.ptr(piv), be.ptr(info), be.stream) == 0
    finally:
        be.lib.tuning(b"lu_split", 0)
        be.lib.tuning(b"lu_split_batch", 0)
    hp, hi, hB, hA = be.host(piv), be.host(info), be.host(dB), be.host(dA)
    assert hi[bad] != 0 and hi[1 - bad] == 0
    assert (hp >= 0).all() and (hp < n).all()
    assert (hA[batch * n * n:] == 7.25).all() and (hB[batch * n * nrhs:] == 7.25).all()
    good = 1 - bad
    X = np.linalg.solve(A[good], B[good])
    Xg = hB[:batch * n * nrhs].reshape(batch, n, nrhs)[good]
    assert np.abs(Xg - X).max() / np.abs(X).max() < 1e-10


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("dtype", [np.complex128, np.complex64])
@pytest.mark.parametrize("n,batch,split", [(530, 2, 0), (777, 1, 128), (1100, 8, 0)])
def test_lu_outer_blocks_deferred_interchanges(backend, dtype, n, batch, split):
    """Several outer blocks (256 columns) + a short tail: a block's row interchanges reach the columns outside the block only when the block
    is finished (lu_swap_range_kernel with all kb pivots), inside the block per panel.  Solve checked by its backward error (independent of
    the conditioning of the random matrix, which an fp32 forward-error bound is not), the pivot vector by being a valid LAPACK interchange
    sequence that reproduces P A = L U.  Own random generator: the data must not depend on which tests ran before."""
    be = get_backend(backend)
    if backend == "emu" and (n > 800 or dtype == np.complex64):
        pytest.skip("emulator: the two smaller sizes in complex128 cover the schedule (the CPU suite stays within minutes)")
    rng = np.random.default_rng(1000 + n)
    A = (rng.standard_normal((batch, n, n)) + 1j * rng.standard_normal((batch, n, n))).astype(dtype)
    A[0, :, 3] *= 1e-3
    B = (rng.standard_normal((batch, n, 5)) + 1j * rng.standard_normal((batch, n, 5))).astype(dtype)
    assert be.lib.tuning(b"lu_split", split) == 0
    try:
        dA, dB = be.dev(A), be.dev(B)
        piv, info = be.empty((batch, n), np.int32), be.dev(np.full((batch,), -7, dtype=np.int32))
        assert be.lib.lu_solve(dtcode(dtype), be.ptr(dA), n, be.ptr(dB), 5, batch, be.ptr(piv), be.ptr(info), be.stream) == 0
    finally:
        be.lib.tuning(b"lu_split", 0)
    assert (be.host(info) == 0).all()
    LU, X, pv = be.host(dA).astype(np.complex128), be.host(dB).astype(np.complex128), be.host(piv)
    A128 = A.astype(np.complex128)
    tol = 1e-13 if dtype == np.complex128 else 1e-5
    for b in range(batch):
        assert ((pv[b] >= np.arange(n)) & (pv[b] < n)).all()
        PA = A128[b].copy()
        for r in range(n):
            if pv[b][r] != r:
                PA[[r, pv[b][r]]] = PA[[pv[b][r], r]]
        Lm, Um = np.tril(LU[b], -1) + np.eye(n), np.triu(LU[b])
        assert np.abs(Lm).max() <= 2 ** 0.5 + 1e-6          # partial pivoting by |re| + |im| (LAPACK cabs1): moduli up to sqrt(2)
        assert np.abs(Lm @ Um - PA).max() / np.abs(PA).max() < 50 * n * (2.2e-16 if dtype == np.complex128 else 1.2e-7)
        berr = np.abs(A128[b] @ X[b] - B[b].astype(np.complex128)).max() / (np.abs(A128[b]).sum(axis=1).max() * np.abs(X[b]).max())
        assert berr < tol, (b, berr)


@pytest.mark.gpu
@pytest.mark.parametrize("opA,opB", [(0, 0), (2, 0), (0, 2)])
def test_gemm_large_tile_hot_shape_gpu(opA, opB):
    """The hot shape of the bench step (1922 = 15 x 128 + 2 = 20 x 96 + 2: both peels, 241 slabs with a K tail of 2) through the default
    large-tile kernel against the 64 x 64 kernel of gemm.hip (knob gemm_big = 4) and, on a row / column sample, against numpy."""
    be = get_backend("gpu")
    import torch
    m = n = k = 1922
    batch = 3
    g = torch.Generator(device="cuda").manual_seed(7)
    A = torch.randn(batch, m, k, dtype=torch.complex128, device="cuda", generator=g)
    B = torch.randn(batch, k, n, dtype=torch.complex128, device="cuda", generator=g)
    C0 = torch.randn(batch, m, n, dtype=torch.complex128, device="cuda", generator=g)
    al, bt = np.array([0.7 - 0.2j]), np.array([-0.3 + 0.5j])
    outs = []
    for knob in (0, 4):
        C = C0.clone()
        try:
            assert be.lib.tuning(b"gemm_big", knob) == 0
            rc = be.lib.gemm(1, opA, opB, m, n, k, al.ctypes.data, A.data_ptr(), k, m * k, B.data_ptr(), n, k * n, bt.ctypes.data, C.data_ptr(), n, m * n,
                             batch, be.stream)
        finally:
            be.lib.tuning(b"gemm_big", 0)
        assert rc == 0
        outs.append(C)
    torch.cuda.synchronize()
    scale = float(outs[1].abs().max())
    assert float((outs[0] - outs[1]).abs().max()) / scale < 1e-12
    f = {0: lambda x: x, 2: lambda x: x.conj().transpose(1, 2)}
    rows = torch.tensor([0, 1, 127, 128, 1919, 1920, 1921], device="cuda")
    ref = al[0] * (f[opA](A)[:, rows, :] @ f[opB](B)) + bt[0] * C0[:, rows, :]
    assert float((outs[0][:, rows, :] - ref).abs().max()) / scale < 1e-12
