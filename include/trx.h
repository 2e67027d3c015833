/* libtrx -- C ABI of the MI355X-native RCWA layer-solve hot path.
 *
 * The reference (kch3782/torcwa 0.1.4.2) has no FFI of its own: the hot path is a chain of torch.* calls inside
 * torcwa/rcwa.py and the single-op seam torcwa/torch_eig.py (`Eig.apply`, used at rcwa.py:1236).  This header is
 * the boundary a maintainer would bind those call sites to (see INTEGRATION.md for the ctypes stub).  Every entry
 * point cites the reference lines it replaces.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes; no torch / HIP types.  `stream` is a hipStream_t passed as void*.
 *  - All matrices are row-major, interleaved complex (re,im), layout-identical to torch.complex64 / complex128;
 *    batched tensors are [batch, rows, cols] contiguous unless a leading dimension / stride is given.
 *  - All pointers are DEVICE pointers (HBM).  The library never allocates device memory and is stream-ordered and
 *    re-entrant: the caller owns every buffer including the workspace, whose size is returned by the matching
 *    *_ws_bytes() function.  Every entry point is asynchronous EXCEPT trx_eig, whose QR iteration is convergence-driven:
 *    per outer iteration (about n / 16 of them per sub-batch) the host reads a 12-byte progress summary from pinned memory, one
 *    iteration after it was produced, and for batch >= 8 it runs the QR phase of 2-4 sub-batches on internal non-blocking streams
 *    (forked from and joined back into `stream` with events) so that their latency-bound steps overlap; on the mixed-precision route it
 *    also reads the per-matrix flag words after the refinement.  Those streams and events come from
 *    a process-wide pool created on first use (nothing is created or destroyed per call), and no environment variable is read per call.
 *  - Return value: 0 = ok, <0 = TRX_ERR_* (bad argument / launch failure).  Numerical failures (singular pivot,
 *    eigensolver non-convergence) are reported LAPACK-style in the device-resident `info[batch]` array.
 */
#ifndef TRX_H_
#define TRX_H_
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { TRX_C64 = 0, TRX_C128 = 1 };          /* dtype */
enum { TRX_OP_N = 0, TRX_OP_T = 1, TRX_OP_C = 2 };
enum {
    TRX_OK = 0,
    TRX_ERR_DTYPE = -1,
    TRX_ERR_ARG = -2,
    TRX_ERR_WORKSPACE = -3,
    TRX_ERR_LAUNCH = -4,
    TRX_ERR_UNSUPPORTED = -5
};

int trx_version(void);                        /* major*10000 + minor*100 + patch */
const char* trx_strerror(int code);

/* ---- Fourier factorisation: torcwa/rcwa.py:1183-1204 (`_material_conv`: fft2 -> Toeplitz gather) --------------
 * out[b,i,j] = c[b, (m_i-m_j) mod nx, (n_i-n_j) mod ny],  c = DFT2(grid[b]) / (nx*ny),
 * i = (m+ox)*(2*oy+1) + (n+oy).  Only the (4ox+1)(4oy+1) needed coefficients are computed (pruned DFT).
 * grid: [batch, nx, ny] real (grid_is_complex=0) or complex, in the real/complex type of `dtype`.
 * Requires nx > 2*ox and ny > 2*oy (same as the reference's negative-index wrap). */
size_t trx_convmat_ws_bytes(int dtype, int batch, int nx, int ny, int ox, int oy);
int trx_convmat(int dtype, int grid_is_complex, const void* grid, int batch, int nx, int ny, int ox, int oy,
                void* out, void* ws, size_t ws_bytes, void* stream);

/* ---- dense complex building blocks (the torch.matmul / torch.linalg.inv call sites, rcwa.py:1157-1304) -------- */
/* C = alpha*op(A)*op(B) + beta*C, batched with element strides; alpha/beta point to HOST complex scalars. */
int trx_gemm(int dtype, int opA, int opB, int m, int n, int k, const void* alpha, const void* A, int lda,
             long strideA, const void* B, int ldb, long strideB, const void* beta, void* C, int ldc, long strideC,
             int batch, void* stream);
/* Solve A X = B in place (partial-pivot LU; A is overwritten by its factors, B by X).
 * piv: int[batch*n] device scratch; info: int[batch] device (0 ok, k>0: zero pivot at column k). */
int trx_lu_solve(int dtype, void* A, int n, void* B, int nrhs, int batch, int* piv, int* info, void* stream);
/* A <- inverse(A) (LU + solve against the identity); ws: batch*n*n elements. */
size_t trx_inverse_ws_bytes(int dtype, int n, int batch);
int trx_inverse(int dtype, void* A, int n, int batch, int* piv, int* info, void* ws, size_t ws_bytes, void* stream);

/* ---- eigendecomposition: torcwa/torch_eig.py:12-17 (`Eig.forward` -> torch.linalg.eig), rcwa.py:1236/1238 -----
 * A [batch,n,n] general complex, DESTROYED.  w [batch,n] eigenvalues, V [batch,n,n] right eigenvectors in the
 * COLUMNS of V (A V = V diag(w)), each column scaled to unit 2-norm (LAPACK geev convention).  Order of the
 * eigenpairs is unspecified (as in LAPACK).  info[b] = 0 ok, >0: number of eigenvalues that failed to converge.
 * Size limit: n*n*sizeof(element) < 4 GiB (n < 16384 for complex128), else TRX_ERR_ARG. */
size_t trx_eig_ws_bytes(int dtype, int n, int batch);
int trx_eig(int dtype, void* A, void* w, void* V, int n, int batch, int* info, void* ws, size_t ws_bytes,
            void* stream);
/* The same with PER-CALL options instead of the process-global knobs "eig_refine" / "eig_vec" (which remain the defaults of trx_eig):
 *   opts bits 0-3: Newton steps of the mixed-precision route, 1-4 (0 = the knob's value);  bits 4-7: eigenvector route, 1-3 as knob "eig_vec"
 *   (0 = the knob's value); other bits must be 0.  The options live only on the calling thread for the duration of the call, so solvers on
 *   different host threads -- or a complex64 and a complex128 solver of one process -- cannot change each other's route or step count
 *   (torcwa_amd.Engine.eig uses these entry points; tests/test_eig.py::test_eig_opts_two_threads).  The workspace size depends on the route:
 *   size it with trx_eig_ws_bytes_opts and the SAME opts. */
/* Number of matrices of the calling thread's last trx_eig / trx_eig_opts that the mixed-precision route could not certify and redid with the
 * all-fp64 pipeline (a cluster of close eigenvalues beyond the refinement's exact treatment: symmetric meta-atoms, dense spectra of large
 * orders; an fp32 result too far off; a singular eigenvector matrix).  Up to a third of the batch is redone as a compact sub-batch inside
 * the same workspace, beyond that the whole batch (the count is then `batch`); 0 = nothing was redone.  Diagnostic only: results do not
 * depend on it, and the library keeps no state between calls. */
int trx_eig_last_fallback(void);
size_t trx_eig_ws_bytes_opts(int dtype, int n, int batch, unsigned opts);
int trx_eig_opts(int dtype, void* A, void* w, void* V, int n, int batch, int* info, void* ws, size_t ws_bytes, void* stream, unsigned opts);

/* Tuning knobs of libtrx (no reference counterpart).  Results do not depend on any of them (tests/test_eig.py, tests/test_blocks.py); they
 * select code paths.  Knobs are process-global and unsynchronised: trx_tuning must not race with a running trx_eig / trx_lu_solve
 * (per-call alternative for the eigensolver's route and refinement depth: trx_eig_opts).  The
 * environment variables named below are read ONCE per process as initial values.  Returns TRX_OK, or TRX_ERR_ARG for an unknown key or a
 * value out of range.  Unless stated otherwise value 0 = automatic (chosen from n and the batch size).
 *   QR phase of trx_eig
 *   "qr_groups"   1-8   iteration groups on their own streams (TRX_QR_GROUPS)        auto: 4 (batch >= 64), 2 (batch >= 8), 1
 *   "qr_aed"      16-64 aggressive-early-deflation window (TRX_QR_AED)                auto: 64; 48 for one chain per sweep below batch 64
 *   "qr_chains"   1-3   bulge chains per sweep (TRX_QR_CHAINS)                        auto: 3 for batch <= 2, 2 up to batch 48, 1 above (one chain =
 *                       the form with super-steps and fused launches, which pays when the batch fills the chip)
 *   "qr_nibble"   0-100 LITERAL percentage, default 100 (TRX_QR_NIBBLE): an AED that deflated less than this share of its window is
 *                       followed by a sweep in the same outer iteration; 0 switches that sweep off.  No automatic value.
 *   "qr_moves"    0-64  LITERAL bound, default 12 (TRX_QR_MOVES): undeflatable eigenvalues an AED moves out of the way; 0 = no reordering.
 *   "qr_rotb"     1 = the in-LDS Schur solver of the AED broadcasts each rotation with ds_bpermute (round-3 code); default: v_readlane (TRX_QR_ROTB)
 *   "qr_super"    1-8   window steps per launch of the chase kernel (one chain per sweep: TRX_QR_SUPER)     auto: 4 (fp32), 8 (fp64).  The workgroup applies each
 *                       window's unitary itself to the band of columns the next windows slide over; the left update beyond the band is one
 *                       launch per super-step, the right update of H and the update of Z one launch per sweep (link log of the sweep)
 *   "qr_defer"    1 = right update of H and update of Z after every super-step instead of once per sweep (TRX_QR_DEFER)   auto: once per sweep
 *                       when the sweep has one chain; with 2-3 chains the following chain reads the rows, so it is per step
 *   "qr_fuse"     1 = every update of a launch's links as its own launch, as in round 5 (TRX_QR_FUSE)
 *                       auto: one chain per sweep -- the NEXT chase launch carries far workgroups that apply the left update beyond the columns the
 *                       chase reaches; several chains -- the NEXT step's chase launch carries the right / Z update of the step (two strips per
 *                       rider wave: TRX_QR_RSPW, environment only)
 *   "slab_spw"    1, 2, 4  strips per wave of the left update (TRX_SLAB_SPW)           auto: 1
 *   "slab_band"   1 = dense window unitary always (TRX_SLAB_BAND)                     auto: skip the structurally zero blocks of a chase unitary
 *   Eigenvector route of trx_eig
 *   "eig_vec"     1 = all-fp64 pipeline (all-fp32 for complex64 input) with Schur vectors, 3 = mixed precision wherever n >= 8: fp32
 *                       eigendecomposition refined to fp64 by Newton steps (TRX_EIG_VEC)
 *                       auto: mixed precision for complex128 input of n >= 256 AND batch >= 8, else Schur vectors; trx_eig_ws_bytes depends on
 *                       this knob (per call and race-free: trx_eig_opts).  (2, inverse iteration on the Hessenberg matrix, was removed in round 5.)
 *   "eig_refine"  1-4   Newton steps of the mixed-precision route; 0 = default (2: eigen-residual ~1e-11 ||A||, what a complex64 caller's
 *                       1e-5 needs with five digits to spare; 3 reach the all-fp64 pipeline's 1e-13 -- torcwa_amd.Engine asks for 3 per call
 *                       (trx_eig_opts) whenever the caller's own dtype is complex128)
 *   GEMM (trx_gemm and every product inside the library)
 *   "gemm_big"    4 = large-tile complex128 kernel (128 x 96 on 8 waves; outputs of at least 2 x 2 tiles, k >= 64) OFF: the 64 x 64 tile everywhere
 *                       (TRX_GEMM_BIG)                                                                              auto: on
 *   LU (trx_lu_solve, trx_inverse and everything built on them)
 *   "lu_split"    rows: a panel is factored by several workgroups per matrix while at least this many rows remain (TRX_LU_SPLIT); 1 = never;
 *                       0 = automatic: only panels too tall for the LDS-resident one-workgroup kernel (fp64: above 1971 rows, fp32: 3942)
 *   "lu_split_batch"  largest batch that uses the row-split panel (TRX_LU_SPLIT_BATCH); 0 = any batch
 *   "lu_sub"      1 = panels column by column as in rounds 1 - 5, 2 = sub-blocks of 4 columns everywhere (TRX_LU_SUB)
 *                       auto: sub-blocks of 8 columns, of 4 for panels too tall for 8 (same pivots)
 *   Hessenberg reduction
 *   "hess_group"  1-4   panels whose right updates of Z and of the rows above the panel are merged into one block reflector and applied
 *                       together (TRX_HESS_GROUP); 1 = every panel on its own as in rounds 1 - 5                      auto: 4
 *   TRX_HESS_RPW=2 (environment only) streams two rows per wave and pass in the BLAS-2 kernel instead of four. */
int trx_tuning(const char* key, int value);

/* Adjoint of the eigendecomposition: torcwa/torch_eig.py:19-44 (`Eig.backward`, the Lorentzian-broadened formula)
 *   gA = (V^H)^-1 (diag(gw) + conj(F) o (V^H gV)) V^H,   F_ij = conj(w_j - w_i) / (|w_j - w_i|^2 + broadening), F_ii = 0.
 * w [batch,n], V [batch,n,n] as returned by trx_eig; gw [batch,n], gV [batch,n,n] incoming gradients; gA [batch,n,n] output.
 * broadening: `Eig.broadening_parameter` (1e-10 by default); pass the smallest positive number of the dtype to reproduce the
 * reference's un-broadened branch (torch_eig.py:27-31).  piv: int[batch*n], info: int[batch] (LU of V^H). */
size_t trx_eig_backward_ws_bytes(int dtype, int n, int batch);
int trx_eig_backward(int dtype, const void* w, const void* V, const void* gw, const void* gV, double broadening, int n, int batch,
                     void* gA, int* piv, int* info, void* ws, size_t ws_bytes, void* stream);

/* ---- layer eigenproblem assembly: torcwa/rcwa.py:1224-1232 (`_eigen_decomposition`, P and Q) -------------------
 * P = [[Kx Ei Ky, M - Kx Ei Kx],[Ky Ei Ky - M, -Ky Ei Kx]],  Q = [[-Kx Mi Ky, Kx Mi Kx - E],[E - Ky Mi Ky, Ky Mi Kx]]
 * E, Einv, Mu, Muinv: [batch,N,N] (Einv = inverse of the permittivity convolution matrix, etc.);
 * kx, ky: [batch,N] complex (the diagonals of Kx_norm, Ky_norm, rcwa.py:1138-1141); P, Q: [batch,2N,2N]. */
int trx_build_pq(int dtype, const void* E, const void* Einv, const void* Mu, const void* Muinv, const void* kx,
                 const void* ky, int N, int batch, void* P, void* Q, void* stream);

/* ---- layer scattering matrix: torcwa/rcwa.py:1244-1281 (`_solve_layer_smatrix`) --------------------------------
 * Inputs  W [batch,n,n] eigenvectors (E_eigvec), n = 2N;
 *         kzfac [batch,n]: kz (use_q=0: V = P^-1 W diag(kz), rcwa.py:1264) or 1/kz (use_q=1: V = Q W diag(1/kz), :1262);
 *         use_q=2: V is an INPUT (already computed, e.g. by trx_hmodes); P, Q and kzfac are not read;
 *         vfinv [4,batch,N]: the four diagonals (p11,p12,p21,p22) of Vf^-1 = [[p11,p12],[p21,p22]] (Vf: rcwa.py:1143-1147);
 *         phase [batch,n] = exp(i*omega*kz*thickness) (rcwa.py:1246).
 * Outputs S11, S21 [batch,n,n] (the layer's S22 == S11 and S12 == S21 identically), V [batch,n,n] (H_eigvec),
 *         optional Cplus, Cminus [batch,n,n]: Cf = [Cplus; Cminus], Cb = [Cminus; Cplus] (rcwa.py:1271-1274).
 * piv: int[3*batch*n], info: int[3*batch] (slot 0..B-1: P factorisation; B..3B-1: the two n x n inverses).
 * Workspace: trx_layer_smatrix_ws_bytes (6 matrices per point); when Cplus == NULL and S11 | S21 are ONE contiguous
 * [2*batch,n,n] block (S21 == S11 + batch*n*n) the outputs double as scratch and trx_layer_smatrix_ws_bytes_lean (4) suffices. */
size_t trx_layer_smatrix_ws_bytes(int dtype, int N, int batch);
size_t trx_layer_smatrix_ws_bytes_lean(int dtype, int N, int batch);
/* H-field modes V = P^-1 W diag(kz) (rcwa.py:1248, 1264) of a layer with HOMOGENEOUS mu, from the rank-N structure
 * P = mu J + [Kx; Ky] E^-1 [Ky, -Kx] (rcwa.py:1226-1228): one N x N factorisation of E - (Kx^2 + Ky^2)/mu and a 2N-column
 * solve replace the LU of the 2N x 2N matrix P (0.29 n^3 instead of 1.33 n^3 complex MACs).  E [batch,N,N] permittivity
 * convolution matrix (NOT its inverse), mu [batch], kx, ky [batch,N], W [batch,n,n], kz [batch,n]; V [batch,n,n] output.
 * piv: int[batch*N], info: int[batch]; ws: trx_hmodes_ws_bytes. */
size_t trx_hmodes_ws_bytes(int dtype, int N, int batch);
int trx_hmodes(int dtype, const void* E, const void* mu, const void* kx, const void* ky, const void* W, const void* kz, int N, int batch,
               void* V, int* piv, int* info, void* ws, size_t ws_bytes, void* stream);
int trx_layer_smatrix(int dtype, const void* P, const void* Q, const void* W, const void* kzfac, const void* vfinv,
                      const void* phase, int use_q, int N, int batch, void* S11, void* S21, void* V, void* Cplus,
                      void* Cminus, int* piv, int* info, void* ws, size_t ws_bytes, void* stream);

/* ---- Redheffer star product: torcwa/rcwa.py:1283-1306 (`_RS_prod`) -----------------------------------------------
 * Sm, Sn, Sout: HOST arrays of 4 device pointers in the reference's order [S11, S21, S12, S22], each [batch,n,n];
 * outputs must not alias inputs.  XY [batch,n,2n] x 2 receives X = [t1 Sm11 | t1 Sm12 Sn22] and
 * Y = [t2 Sn21 Sm11 | t2 Sn22] -- exactly the four products the reference needs to propagate the mode-coupling
 * coefficients C (rcwa.py:1297-1304), so the caller can do that lazily.  piv: int[batch*n], info: int[batch]. */
size_t trx_redheffer_ws_bytes(int dtype, int n, int batch);
int trx_redheffer(int dtype, const void* const* Sm, const void* const* Sn, void* const* Sout, void* XY, int n, int batch,
                  int* piv, int* info, void* ws, size_t ws_bytes, void* stream);

/* Star product with a HALF-SPACE operand (the Sin / Sout coupling steps of solve_global_smatrix, rcwa.py:198-208).
 * The four blocks of Sin / Sout are 2x2-block-diagonal (rcwa.py:1157-1181), so they are passed as diagonals
 * bd[4 blocks S11,S21,S12,S22][4 diagonals d11,d12,d21,d22][batch][N] and every product with them is O(n^2).
 * side = 0: half-space on the left (Sin * S);  side = 1: on the right (S * Sout).  Other arguments as trx_redheffer.
 * XY may be NULL for side = 0 when the coupling factors are not needed (no C lists to propagate): the product is then
 * formed with right-solves (4.33 n^3 instead of 6.33 n^3 complex MACs) and needs the larger workspace reported by
 * trx_redheffer_halfspace_ws_bytes(dtype, N, batch, side, want_xy = 0). */
size_t trx_redheffer_halfspace_ws_bytes(int dtype, int N, int batch, int side, int want_xy);
int trx_redheffer_halfspace(int dtype, int side, const void* bd, const void* const* S, void* const* Sout, void* XY, int N, int batch,
                            int* piv, int* info, void* ws, size_t ws_bytes, void* stream);

/* A = P Q (rcwa.py:1236) for a layer with homogeneous mu[batch], from its block structure (two N^3 GEMMs instead of
 * one (2N)^3): A = [[mu E - Ky^2 - Kx Gx, KxKy - Kx Gy],[KxKy - Ky Gx, mu E - Kx^2 - Ky Gy]], G* = Einv (K* E). */
size_t trx_build_a_ws_bytes(int dtype, int N, int batch);
int trx_build_a(int dtype, const void* E, const void* Einv, const void* mu, const void* kx, const void* ky, int N, int batch, void* A,
                void* ws, size_t ws_bytes, void* stream);

/* ---- measurement aid (no reference counterpart): HIP-event timing of the dominant kernels --------------------
 * trx_prof_enable(1) makes the instrumented launch sites record hipEvents on the launch stream.  Sampling is systematic and
 * uniform over the run: every stride-th launch of a tag is timed; when the pool (2048 event pairs per tag) is full every
 * other sample is dropped and the stride doubles.  trx_prof_get(tag, out[7]) waits for those events and returns
 * {launches, timed_launches, algorithmic flops of the timed launches, algorithmic bytes of the timed launches,
 * milliseconds of the timed launches, flops of ALL launches, bytes of ALL launches} (the last two are exact sums, not samples).  Tags: 0 gemm N,N; 1 gemm other ops; 2 QR prepare (AED);
 * 3 QR off-window update; 4 QR window chase; 5 Hessenberg gemv; 6 Hessenberg reflector column; 7 LU panel. */
int trx_prof_enable(int on);
int trx_prof_reset(void);
int trx_prof_get(int tag, double* out);
const char* trx_prof_tag_name(int tag);

#ifdef __cplusplus
}
#endif
#endif /* TRX_H_ */
