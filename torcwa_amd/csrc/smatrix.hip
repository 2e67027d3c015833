// Layer eigenproblem assembly, layer scattering matrix and Redheffer star product (batched, row-major complex).
// Replaces torcwa/rcwa.py:1224-1232 (`_eigen_decomposition`: P, Q), :1244-1281 (`_solve_layer_smatrix`) and
// :1283-1306 (`_RS_prod`) with the lean formulation of SURVEY.md section 7.2, which is algebraically identical:
//
//   * Kx, Ky, Kz, X=exp(i w kz d), Vf are (block-)diagonal: they are never materialised, every product with them
//     is a fused row/column scaling inside an assembly kernel.
//   * [[A,B],[B,A]]^-1 of rcwa.py:1268-1274 (a 2n x 2n inverse, evaluated twice) is replaced by two n x n
//     inverses:  c+ + c- = 2 (A+B)^-1,  c+ - c- = 2 (A-B)^-1,  and S22 = S11, S12 = S21, Cb = swap(Cf).
//   * The star product needs ONE LU (push-through identity (I-BA)^-1 B = B (I-AB)^-1) instead of two inverses.
#include "common.hpp"

namespace trx {
namespace {

// P = [[Kx Ei Ky, M - Kx Ei Kx], [Ky Ei Ky - M, -Ky Ei Kx]],  Q = [[-Kx Mi Ky, Kx Mi Kx - E], [E - Ky Mi Ky, Ky Mi Kx]]
template <class T>
__global__ __launch_bounds__(256) void build_pq_kernel(const cx<T>* __restrict__ E, const cx<T>* __restrict__ Ei,
                                                       const cx<T>* __restrict__ M, const cx<T>* __restrict__ Mi,
                                                       const cx<T>* __restrict__ kx, const cx<T>* __restrict__ ky, int N,
                                                       cx<T>* __restrict__ P, cx<T>* __restrict__ Q) {
    const int b = blockIdx.z, i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    const long o = ((long)b * N + i) * N + j;
    const cx<T> e = E[o], ei = Ei[o], m = M[o], mi = Mi[o];
    const cx<T> kxi = kx[(long)b * N + i], kyi = ky[(long)b * N + i], kxj = kx[(long)b * N + j], kyj = ky[(long)b * N + j];
    const int n = 2 * N;
    cx<T>* Pb = P + (long)b * n * n;
    cx<T>* Qb = Q + (long)b * n * n;
    const long r0 = (long)i * n + j, r1 = (long)(i + N) * n + j;
    Pb[r0] = kxi * ei * kyj;
    Pb[r0 + N] = m - kxi * ei * kxj;
    Pb[r1] = kyi * ei * kyj - m;
    Pb[r1 + N] = -(kyi * ei * kxj);
    Qb[r0] = -(kxi * mi * kyj);
    Qb[r0 + N] = kxi * mi * kxj - e;
    Qb[r1] = e - kyi * mi * kyj;
    Qb[r1 + N] = kyi * mi * kxj;
}

// out[i,j] = in[i,j] * s[j]
template <class T>
__global__ __launch_bounds__(256) void scale_cols_kernel(const cx<T>* __restrict__ in, const cx<T>* __restrict__ s, int n, cx<T>* __restrict__ out) {
    const int b = blockIdx.z, i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const long o = ((long)b * n + i) * n + j;
    out[o] = in[o] * s[(long)b * n + j];
}

// F = Vf^-1 V (Vf^-1 2x2-block-diagonal, rows i and i+N coupled);  Tp = (W+F) + (W-F) X,  Tm = (W+F) - (W-F) X
template <class T>
__global__ __launch_bounds__(256) void layer_T_kernel(const cx<T>* __restrict__ W, const cx<T>* __restrict__ V,
                                                      const cx<T>* __restrict__ p11, const cx<T>* __restrict__ p12,
                                                      const cx<T>* __restrict__ p21, const cx<T>* __restrict__ p22,
                                                      const cx<T>* __restrict__ x, int N, cx<T>* __restrict__ Tp, cx<T>* __restrict__ Tm) {
    const int b = blockIdx.z, i = blockIdx.y;           // i in [0, N)
    const int n = 2 * N;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const long o0 = ((long)b * n + i) * n + j, o1 = ((long)b * n + i + N) * n + j;
    const cx<T> v0 = V[o0], v1 = V[o1], w0 = W[o0], w1 = W[o1];
    const long d = (long)b * N + i;
    const cx<T> f0 = p11[d] * v0 + p12[d] * v1;
    const cx<T> f1 = p21[d] * v0 + p22[d] * v1;
    const cx<T> xj = x[(long)b * n + j];
    const cx<T> a0 = w0 + f0, b0 = (w0 - f0) * xj;
    const cx<T> a1 = w1 + f1, b1 = (w1 - f1) * xj;
    Tp[o0] = a0 + b0; Tm[o0] = a0 - b0;
    Tp[o1] = a1 + b1; Tm[o1] = a1 - b1;
}

// G1 = (I+X) Tip, G2 = (I-X) Tim (row scalings);  optionally c+ = Tip + Tim, c- = Tip - Tim
template <class T>
__global__ __launch_bounds__(256) void layer_G_kernel(const cx<T>* __restrict__ Tip, const cx<T>* __restrict__ Tim, const cx<T>* __restrict__ x,
                                                      int n, cx<T>* __restrict__ G1, cx<T>* __restrict__ G2, cx<T>* __restrict__ cp, cx<T>* __restrict__ cm) {
    const int b = blockIdx.z, i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const long o = ((long)b * n + i) * n + j;
    const cx<T> xi = x[(long)b * n + i];
    const cx<T> tp = Tip[o], tm = Tim[o];
    const cx<T> one(T(1), T(0));
    if (cp) { cp[o] = tp + tm; cm[o] = tp - tm; }
    G1[o] = (one + xi) * tp;
    G2[o] = (one - xi) * tm;
}

// S11 = Mp - Mm,  S21 = Mp + Mm - I
template <class T>
__global__ __launch_bounds__(256) void layer_S_kernel(const cx<T>* Mp, const cx<T>* Mm, int n, cx<T>* S11, cx<T>* S21) {      // may run in place (S11 == Mp, S21 == Mm)
    const int b = blockIdx.z, i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const long o = ((long)b * n + i) * n + j;
    const cx<T> p = Mp[o], m = Mm[o];
    S11[o] = p - m;
    cx<T> s = p + m;
    if (i == j) s.x -= T(1);
    S21[o] = s;
}

// dst[b, i, dcol0 + j] = (identity ? delta_ij : 0) + alpha * src[b, i, j]    (strided block copy / init)
template <class T>
__global__ __launch_bounds__(256) void block_copy_kernel(const cx<T>* __restrict__ src, int lds, long ss, cx<T>* __restrict__ dst, int ldd, long sd,
                                                         int rows, int cols, T alpha, int add_identity) {
    const int b = blockIdx.z, i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= cols || i >= rows) return;
    cx<T> v(T(0), T(0));
    if (src) v = alpha * src[(long)b * ss + (long)i * lds + j];
    if (add_identity && i == j) v.x += T(1);
    dst[(long)b * sd + (long)i * ldd + j] = v;
}

// out[b] = in[b]^T  (n x n blocks with leading dimensions ldi / ldo and batch strides si / so; tiled through LDS so that
// both the read and the write are coalesced).  Tile 32 rows x 64 columns per 256-thread workgroup: 8 independent 16-byte (complex128) loads
// per thread in flight, rows of 1 KB read and 512 B written contiguously; row stride 65 elements keeps the transposed b128 reads of a 16-lane
// group on distinct banks.  (The 32 x 32 tile of rounds 1 - 4 moved 2.6 TB/s: 5.7 ms per call at the bench shape, 8 calls per step.)
constexpr int TRR = 32, TRC = 64;
template <class T>
__global__ __launch_bounds__(256) void transpose_kernel(const cx<T>* __restrict__ in, int ldi, long si, cx<T>* __restrict__ out, int ldo, long so, int n) {
    __shared__ cx<T> tile[TRR][TRC + 1];
    in += (long)blockIdx.z * si;
    out += (long)blockIdx.z * so;
    const int c0 = blockIdx.x * TRC, r0 = blockIdx.y * TRR;
    const int t = threadIdx.y * blockDim.x + threadIdx.x;        // launched as (32, 8)
    {
        const int tx = t & (TRC - 1), ty = t / TRC;              // 64 columns x 4 rows per pass
        cx<T> v[TRR / 4];
#pragma unroll
        for (int i = 0; i < TRR / 4; ++i) {
            const int r = r0 + ty + 4 * i, c = c0 + tx;
            v[i] = in[(long)(r < n ? r : n - 1) * ldi + (c < n ? c : n - 1)];          // clamped: all loads in flight, no branch
        }
#pragma unroll
        for (int i = 0; i < TRR / 4; ++i) tile[ty + 4 * i][tx] = v[i];
    }
    __syncthreads();
    {
        const int tx = t & (TRR - 1), ty = t / TRR;              // 32 columns (= input rows) x 8 rows (= input columns) per pass
#pragma unroll
        for (int i = 0; i < TRC / 8; ++i) {
            const int r = c0 + ty + 8 * i, c = r0 + tx;          // transposed block
            if (r < n && c < n) out[(long)r * ldo + c] = tile[tx][ty + 8 * i];
        }
    }
}

// Rp = W (I + X), Rm = W (I - X)   (column scalings by the layer phase)
template <class T>
__global__ __launch_bounds__(256) void layer_R_kernel(const cx<T>* __restrict__ W, const cx<T>* __restrict__ x, int n, cx<T>* __restrict__ Rp, cx<T>* __restrict__ Rm) {
    const int b = blockIdx.z, i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const long o = ((long)b * n + i) * n + j;
    const cx<T> w = W[o], xj = x[(long)b * n + j];
    const cx<T> wx = w * xj;
    Rp[o] = w + wx;
    Rm[o] = w - wx;
}

template <class T>
int build_pq_t(hipStream_t s, const void* E, const void* Ei, const void* M, const void* Mi, const void* kx, const void* ky, int N, int batch, void* P, void* Q) {
    TRX_LAUNCH((build_pq_kernel<T>), dim3(cdiv_i(N, 256), N, batch), dim3(256), 0, s, (const cx<T>*)E, (const cx<T>*)Ei, (const cx<T>*)M, (const cx<T>*)Mi,
               (const cx<T>*)kx, (const cx<T>*)ky, N, (cx<T>*)P, (cx<T>*)Q);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

// ---- V = P^-1 (W Kz) through the rank-N structure of P (homogeneous mu) -------------------------------------------------
// P = mu J + [Kx; Ky] E^-1 [Ky, -Kx],  J = [[0, I], [-I, 0]]   (rcwa.py:1226-1228 with a scalar mu).  Woodbury:
//   P^-1 B = C - (1/mu) [-Ky Y; Kx Y],   C = (1/mu) [-B2; B1],   (E - (Kx^2 + Ky^2)/mu) Y = Ky C1 - Kx C2
// i.e. one N x N factorisation and one N x 2N-column solve (n^3/24 + n^3/4 complex MACs) instead of the LU of the 2N x 2N
// matrix P and a 2N-column solve (n^3/3 + n^3).  E (not E^-1) is the matrix that gets factorised.
template <class T>
__global__ __launch_bounds__(256) void hm_inner_kernel(const cx<T>* __restrict__ E, const cx<T>* __restrict__ mu, const cx<T>* __restrict__ kx,
                                                       const cx<T>* __restrict__ ky, int N, cx<T>* __restrict__ Mi) {
    const int b = blockIdx.z, i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    const long o = ((long)b * N + i) * N + j;
    cx<T> v = E[o];
    if (i == j) {
        const cx<T> x = kx[(long)b * N + i], y = ky[(long)b * N + i];
        v -= cdiv(x * x + y * y, mu[b]);
    }
    Mi[o] = v;
}
// D[i, j] = (-ky_i B2[i,j] - kx_i B1[i,j]) / mu,   B = W Kz  (column j scaled by kz_j)
template <class T>
__global__ __launch_bounds__(256) void hm_rhs_kernel(const cx<T>* __restrict__ W, const cx<T>* __restrict__ kz, const cx<T>* __restrict__ mu,
                                                     const cx<T>* __restrict__ kx, const cx<T>* __restrict__ ky, int N, cx<T>* __restrict__ D) {
    const int b = blockIdx.z, i = blockIdx.y, n = 2 * N;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const cx<T>* Wb = W + (long)b * n * n;
    const cx<T> k = kz[(long)b * n + j];
    const cx<T> b1 = Wb[(long)i * n + j] * k, b2 = Wb[(long)(i + N) * n + j] * k;
    const cx<T> x = kx[(long)b * N + i], y = ky[(long)b * N + i];
    D[((long)b * N + i) * n + j] = cdiv(-(y * b2) - x * b1, mu[b]);
}
// V_top = (-B2 + ky Y)/mu,  V_bot = (B1 - kx Y)/mu
template <class T>
__global__ __launch_bounds__(256) void hm_finish_kernel(const cx<T>* __restrict__ W, const cx<T>* __restrict__ kz, const cx<T>* __restrict__ mu,
                                                        const cx<T>* __restrict__ kx, const cx<T>* __restrict__ ky, const cx<T>* __restrict__ Y, int N,
                                                        cx<T>* __restrict__ V) {
    const int b = blockIdx.z, i = blockIdx.y, n = 2 * N;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const cx<T>* Wb = W + (long)b * n * n;
    cx<T>* Vb = V + (long)b * n * n;
    const cx<T> k = kz[(long)b * n + j];
    const cx<T> b1 = Wb[(long)i * n + j] * k, b2 = Wb[(long)(i + N) * n + j] * k;
    const cx<T> x = kx[(long)b * N + i], y = ky[(long)b * N + i];
    const cx<T> yv = Y[((long)b * N + i) * n + j];
    Vb[(long)i * n + j] = cdiv(y * yv - b2, mu[b]);
    Vb[(long)(i + N) * n + j] = cdiv(b1 - x * yv, mu[b]);
}
template <class T>
int hmodes_t(hipStream_t s, const cx<T>* E, const cx<T>* mu, const cx<T>* kx, const cx<T>* ky, const cx<T>* W, const cx<T>* kz, int N, int batch,
             cx<T>* V, int* piv, int* info, cx<T>* ws) {
    const int n = 2 * N;
    const long NN = (long)N * N, Nn = (long)N * n;
    cx<T>* Mi = ws;                         // [B,N,N]
    cx<T>* D = ws + (long)batch * NN;       // [B,N,n]
    const dim3 blk(256);
    TRX_LAUNCH((hm_inner_kernel<T>), dim3(cdiv_i(N, 256), N, batch), blk, 0, s, E, mu, kx, ky, N, Mi);
    TRX_LAUNCH((hm_rhs_kernel<T>), dim3(cdiv_i(n, 256), N, batch), blk, 0, s, W, kz, mu, kx, ky, N, D);
    int rc = lu_factor<T>(s, Mi, N, NN, N, piv, batch, info); if (rc) return rc;
    rc = lu_solve<T>(s, Mi, N, NN, N, piv, D, n, Nn, n, batch); if (rc) return rc;
    TRX_LAUNCH((hm_finish_kernel<T>), dim3(cdiv_i(n, 256), N, batch), blk, 0, s, W, kz, mu, kx, ky, (const cx<T>*)D, N, V);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

template <class T>
int layer_smatrix_t(hipStream_t s, const cx<T>* P, const cx<T>* Q, const cx<T>* W, const cx<T>* kz, const cx<T>* pv, const cx<T>* x, int use_q,
                    int N, int batch, cx<T>* S11, cx<T>* S21, cx<T>* V, cx<T>* cp, cx<T>* cm, int* piv, int* info, cx<T>* ws) {
    const int n = 2 * N;
    const long nn = (long)n * n, bn = (long)batch * nn;
    const cx<T> one(T(1), T(0)), zero(T(0), T(0));
    const dim3 g(cdiv_i(n, 256), n, batch), blk(256);
    cx<T>* T2 = ws;              // [2B, n, n]: Tp | Tm, inverted in place
    cx<T>* G = ws + 2 * bn;      // [2B, n, n]: scratch (LU copy of P / inverse workspace / G1 | G2)
    // [2B, n, n]: Mp | Mm.  Without coupling coefficients the last step (layer_S_kernel) is elementwise, so when the caller's
    // S11 | S21 are one contiguous [2B,n,n] block they serve as this buffer and the workspace is 4 instead of 6 matrices per point
    cx<T>* Mx = (!cp && S21 == S11 + bn) ? S11 : ws + 4 * bn;
    int rc;
    if (use_q == 2) {
        // V supplied by the caller (trx_hmodes)
        if (hipMemsetAsync(info, 0, sizeof(int) * batch, s) != hipSuccess) return TRX_ERR_LAUNCH;
    } else if (!use_q) {
        // V = P^-1 (W Kz)                                                       (rcwa.py:1248, 1264)
        if (hipMemcpyAsync(G, P, sizeof(cx<T>) * bn, hipMemcpyDeviceToDevice, s) != hipSuccess) return TRX_ERR_LAUNCH;
        TRX_LAUNCH((scale_cols_kernel<T>), g, blk, 0, s, W, kz, n, V);
        rc = lu_factor<T>(s, G, n, nn, n, piv, batch, info); if (rc) return rc;
        rc = lu_solve<T>(s, G, n, nn, n, piv, V, n, nn, n, batch); if (rc) return rc;
    } else {
        // V = Q W Kz^-1   (kz holds 1/kz on this path)                         (rcwa.py:1262)
        TRX_LAUNCH((scale_cols_kernel<T>), g, blk, 0, s, W, kz, n, G);
        rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, n, n, one, Q, n, nn, G, n, nn, zero, V, n, nn, batch); if (rc) return rc;
        if (hipMemsetAsync(info, 0, sizeof(int) * batch, s) != hipSuccess) return TRX_ERR_LAUNCH;
    }
    const long bN = (long)batch * N;
    TRX_LAUNCH((layer_T_kernel<T>), dim3(cdiv_i(n, 256), N, batch), blk, 0, s, W, (const cx<T>*)V, pv, pv + bN, pv + 2 * bN, pv + 3 * bN, x, N, T2, T2 + bn);
    int* piv2 = piv + (long)batch * n;               // caller provides 3*batch*n ints
    int* info2 = info + batch;                       // and 3*batch info slots
    if (!cp) {
        // Coupling coefficients not requested: M+ = W(I+X) Tp^-1 and M- = W(I-X) Tm^-1 are RIGHT solves, done as such from the factors of
        // Tp | Tm (lu_solve_right: X U = B, then X L = ., then the column permutation) -- 2.67 n^3 instead of the 4.67 n^3 cMAC of the
        // explicit-inverse route.  (Rounds 1 - 5 solved the transposed systems from the left: three tiled transposes of [2B,n,n] per call,
        // 20 ms of a 128-point step.)
        TRX_LAUNCH((layer_R_kernel<T>), g, blk, 0, s, W, x, n, Mx, Mx + bn);                             // Mx = W(I+X) | W(I-X)
        rc = lu_factor<T>(s, T2, n, nn, n, piv2, 2 * batch, info2); if (rc) return rc;
        rc = lu_solve_right<T>(s, T2, n, nn, n, piv2, Mx, n, nn, n, 2 * batch, reinterpret_cast<int*>(G)); if (rc) return rc;     // Mx = M+ | M-
        TRX_LAUNCH((layer_S_kernel<T>), g, blk, 0, s, (const cx<T>*)Mx, (const cx<T>*)(Mx + bn), n, S11, S21);
        TRX_CHECK_LAUNCH();
        return TRX_OK;
    }
    // invert Tp and Tm as one batch of 2B
    {
        rc = lu_factor<T>(s, T2, n, nn, n, piv2, 2 * batch, info2); if (rc) return rc;
        TRX_LAUNCH((block_copy_kernel<T>), dim3(cdiv_i(n, 256), n, 2 * batch), blk, 0, s, (const cx<T>*)nullptr, n, nn, G, n, nn, n, n, T(0), 1);
        rc = lu_solve<T>(s, T2, n, nn, n, piv2, G, n, nn, n, 2 * batch); if (rc) return rc;
    }
    // G now holds Tip | Tim; form G1 | G2 into T2 (factors no longer needed)
    TRX_LAUNCH((layer_G_kernel<T>), g, blk, 0, s, (const cx<T>*)G, (const cx<T>*)(G + bn), x, n, T2, T2 + bn, cp, cm);
    rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, n, n, one, W, n, nn, T2, n, nn, zero, Mx, n, nn, batch); if (rc) return rc;
    rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, n, n, one, W, n, nn, T2 + bn, n, nn, zero, Mx + bn, n, nn, batch); if (rc) return rc;
    TRX_LAUNCH((layer_S_kernel<T>), g, blk, 0, s, (const cx<T>*)Mx, (const cx<T>*)(Mx + bn), n, S11, S21);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

template <class T>
int redheffer_t(hipStream_t s, const cx<T>* const* Sm, const cx<T>* const* Sn, cx<T>* const* O, cx<T>* XY, int n, int batch, int* piv, int* info, cx<T>* ws) {
    // index convention of the reference: [0]=S11, [1]=S21, [2]=S12, [3]=S22            (rcwa.py:1284)
    const long nn = (long)n * n, bn = (long)batch * nn;
    const cx<T> one(T(1), T(0)), mone(T(-1), T(0)), zero(T(0), T(0));
    const dim3 g(cdiv_i(n, 256), n, batch), blk(256);
    cx<T>* K = ws;                 // [B,n,n]
    cx<T>* X = XY;                 // [B,n,2n]  X1 | X2
    cx<T>* Y = XY + 2 * bn;        // [B,n,2n]  Y1 | Y2
    int rc;
    // K = I - Sm12 Sn21
    TRX_LAUNCH((block_copy_kernel<T>), g, blk, 0, s, (const cx<T>*)nullptr, n, nn, K, n, nn, n, n, T(0), 1);
    rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, n, n, mone, Sm[2], n, nn, Sn[1], n, nn, one, K, n, nn, batch); if (rc) return rc;
    // RHS = [Sm11 | Sm12 Sn22]
    TRX_LAUNCH((block_copy_kernel<T>), g, blk, 0, s, Sm[0], n, nn, X, 2 * n, 2 * nn, n, n, T(1), 0);
    rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, n, n, one, Sm[2], n, nn, Sn[3], n, nn, zero, X + n, 2 * n, 2 * nn, batch); if (rc) return rc;
    rc = lu_factor<T>(s, K, n, nn, n, piv, batch, info); if (rc) return rc;
    rc = lu_solve<T>(s, K, n, nn, n, piv, X, 2 * n, 2 * nn, 2 * n, batch); if (rc) return rc;
    // S11 = Sn11 X1 ; S12 = Sn12 + Sn11 X2
    rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, n, n, one, Sn[0], n, nn, X, 2 * n, 2 * nn, zero, O[0], n, nn, batch); if (rc) return rc;
    TRX_LAUNCH((block_copy_kernel<T>), g, blk, 0, s, Sn[2], n, nn, O[2], n, nn, n, n, T(1), 0);
    rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, n, n, one, Sn[0], n, nn, X + n, 2 * n, 2 * nn, one, O[2], n, nn, batch); if (rc) return rc;
    // Y1 = Sn21 X1 ; Y2 = Sn22 + Sn21 X2       (= t2 Sn21 Sm11 and t2 Sn22 of rcwa.py:1292-1294, 1299-1300)
    rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, n, n, one, Sn[1], n, nn, X, 2 * n, 2 * nn, zero, Y, 2 * n, 2 * nn, batch); if (rc) return rc;
    TRX_LAUNCH((block_copy_kernel<T>), g, blk, 0, s, Sn[3], n, nn, Y + n, 2 * n, 2 * nn, n, n, T(1), 0);
    rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, n, n, one, Sn[1], n, nn, X + n, 2 * n, 2 * nn, one, Y + n, 2 * n, 2 * nn, batch); if (rc) return rc;
    // S21 = Sm21 + Sm22 Y1 ; S22 = Sm22 Y2
    TRX_LAUNCH((block_copy_kernel<T>), g, blk, 0, s, Sm[1], n, nn, O[1], n, nn, n, n, T(1), 0);
    rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, n, n, one, Sm[3], n, nn, Y, 2 * n, 2 * nn, one, O[1], n, nn, batch); if (rc) return rc;
    rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, n, n, one, Sm[3], n, nn, Y + n, 2 * n, 2 * nn, zero, O[3], n, nn, batch); if (rc) return rc;
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}


// ---- 2x2-block-diagonal operators (Sin/Sout blocks, rcwa.py:1157-1181): D = [[d0, d1], [d2, d3]], each d* a diagonal ----
// out = alpha * (D X) [+ I] [+ Y]     (row combination: rows i and i+N of X)
template <class T>
__global__ __launch_bounds__(256) void bd_rowcomb_kernel(const cx<T>* __restrict__ d, long dstride, const cx<T>* __restrict__ X, int ldx, long sx,
                                                         const cx<T>* __restrict__ Y, int ldy, long sy, cx<T>* __restrict__ out, int ldo, long so,
                                                         int N, int ncols, T alpha, int add_identity) {
    const int b = blockIdx.z, i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ncols) return;
    const long di = (long)b * N + i;
    const cx<T> d0 = d[di], d1 = d[dstride + di], d2 = d[2 * dstride + di], d3 = d[3 * dstride + di];
    const cx<T> x0 = X[(long)b * sx + (long)i * ldx + j], x1 = X[(long)b * sx + (long)(i + N) * ldx + j];
    cx<T> o0 = alpha * (d0 * x0 + d1 * x1), o1 = alpha * (d2 * x0 + d3 * x1);
    if (add_identity) { if (j == i) o0.x += T(1); if (j == i + N) o1.x += T(1); }
    if (Y) { o0 += Y[(long)b * sy + (long)i * ldy + j]; o1 += Y[(long)b * sy + (long)(i + N) * ldy + j]; }
    out[(long)b * so + (long)i * ldo + j] = o0;
    out[(long)b * so + (long)(i + N) * ldo + j] = o1;
}
// out = alpha * (X D) [+ I]           (column combination: columns j and j+N of X)
template <class T>
__global__ __launch_bounds__(256) void bd_colcomb_kernel(const cx<T>* __restrict__ d, long dstride, const cx<T>* __restrict__ X, int ldx, long sx,
                                                         cx<T>* __restrict__ out, int ldo, long so, int N, int nrows, T alpha, int add_identity) {
    const int b = blockIdx.z, i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;      // j in [0, N)
    if (j >= N || i >= nrows) return;
    const long dj = (long)b * N + j;
    const cx<T> d0 = d[dj], d1 = d[dstride + dj], d2 = d[2 * dstride + dj], d3 = d[3 * dstride + dj];
    const cx<T> x0 = X[(long)b * sx + (long)i * ldx + j], x1 = X[(long)b * sx + (long)i * ldx + j + N];
    cx<T> o0 = alpha * (x0 * d0 + x1 * d2), o1 = alpha * (x0 * d1 + x1 * d3);
    if (add_identity) { if (i == j) o0.x += T(1); if (i == j + N) o1.x += T(1); }
    out[(long)b * so + (long)i * ldo + j] = o0;
    out[(long)b * so + (long)i * ldo + j + N] = o1;
}
// out = dense(D) [+ Y]
template <class T>
__global__ __launch_bounds__(256) void bd_dense_kernel(const cx<T>* __restrict__ d, long dstride, const cx<T>* __restrict__ Y, int ldy, long sy,
                                                       cx<T>* __restrict__ out, int ldo, long so, int N) {
    const int b = blockIdx.z, i = blockIdx.y;                 // i in [0, 2N)
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = 2 * N;
    if (j >= n) return;
    cx<T> v(T(0), T(0));
    const int ii = i < N ? i : i - N, jj = j < N ? j : j - N;
    if (ii == jj) v = d[(long)((i < N ? 0 : 2) + (j < N ? 0 : 1)) * dstride + (long)b * N + ii];
    if (Y) v += Y[(long)b * sy + (long)i * ldy + j];
    out[(long)b * so + (long)i * ldo + j] = v;
}

// Star product with a HALF-SPACE operand whose four blocks are 2x2-block-diagonal (Sin on the left, side = 0, or
// Sout on the right, side = 1): every product with a block-diagonal factor is an O(n^2) row/column combination.
// bd: [4 blocks (S11,S21,S12,S22)][4 diagonals][B][N].
template <class T>
int redheffer_halfspace_t(hipStream_t s, int side, const cx<T>* bd, const cx<T>* const* S, cx<T>* const* O, cx<T>* XY, int N, int batch,
                          int* piv, int* info, cx<T>* ws) {
    const int n = 2 * N;
    const long nn = (long)n * n, bn = (long)batch * nn, bN = (long)batch * N;
    const cx<T> one(T(1), T(0)), zero(T(0), T(0));
    const dim3 blk(256), gN(cdiv_i(n, 256), N, batch), gn(cdiv_i(n, 256), n, batch), gc(cdiv_i(N, 256), n, batch);
    const cx<T>*D11 = bd, *D21 = bd + 4 * bN, *D12 = bd + 8 * bN, *D22 = bd + 12 * bN;
    cx<T>* K = ws;                 // [B,n,n]
    cx<T>* X = XY;                 // [B,n,2n]  X1 | X2
    cx<T>* Y = XY + 2 * bn;        // [B,n,2n]  Y1 | Y2
    int rc;
    if (side == 0 && !XY) {
        // Coupling factors not requested: with D = half-space blocks, K = I - D12 Sn21, T1 = Sn11 K^-1, T2 = Sn21 K^-1 (RIGHT solves,
        // done as one left solve of the transposed system with 2n right-hand sides) and M = D12 Sn22:
        //   S11 = T1 D11,   S12 = Sn12 + T1 M,   S21 = D21 + D22 T2 D11,   S22 = D22 (Sn22 + T2 M)
        // (push-through: (I - Sn21 D12)^-1 = I + T2 D12).  One LU, one 2n-column solve, two n^3 products = 4.33 n^3 complex MACs
        // instead of 6.33; every other step is an O(n^2) combination with block-diagonal factors or a tiled transpose.
        // Scratch that dies before its host is written lives in the OUTPUT blocks (K in S12's, K^T and later M in S22's), so the
        // workspace is the two [B,n,2n] solve buffers only: 4 instead of 6 matrices per point.
        K = O[2];
        cx<T>* Kt = O[3];              // [B,n,n]
        cx<T>* Xs = ws;                // [B,n,2n]  [Sn11^T | Sn21^T] -> solution; later two [B,n,n] temporaries
        cx<T>* Ys = ws + 2 * bn;       // [B,n,2n]  [T1 | T2]
        const dim3 tg(cdiv_i(n, TRC), cdiv_i(n, TRR), batch), tb(32, 8);
        TRX_LAUNCH((bd_rowcomb_kernel<T>), gN, blk, 0, s, D12, bN, S[1], n, nn, (const cx<T>*)nullptr, 0, 0L, K, n, nn, N, n, T(-1), 1);
        TRX_LAUNCH((transpose_kernel<T>), tg, tb, 0, s, (const cx<T>*)K, n, nn, Kt, n, nn, n);
        TRX_LAUNCH((transpose_kernel<T>), tg, tb, 0, s, S[0], n, nn, Xs, 2 * n, 2 * nn, n);
        TRX_LAUNCH((transpose_kernel<T>), tg, tb, 0, s, S[1], n, nn, Xs + n, 2 * n, 2 * nn, n);
        rc = lu_factor<T>(s, Kt, n, nn, n, piv, batch, info); if (rc) return rc;
        rc = lu_solve<T>(s, Kt, n, nn, n, piv, Xs, 2 * n, 2 * nn, 2 * n, batch); if (rc) return rc;
        TRX_LAUNCH((transpose_kernel<T>), tg, tb, 0, s, (const cx<T>*)Xs, 2 * n, 2 * nn, Ys, 2 * n, 2 * nn, n);
        TRX_LAUNCH((transpose_kernel<T>), tg, tb, 0, s, (const cx<T>*)(Xs + n), 2 * n, 2 * nn, Ys + n, 2 * n, 2 * nn, n);
        cx<T>* M = Kt;                 // the factors of K^T are no longer needed after the solve (and S22, their host, is written last)
        cx<T>* Ta = Xs;                // [B,n,n] temporaries in the solve buffer
        cx<T>* Tb = Xs + bn;
        TRX_LAUNCH((bd_rowcomb_kernel<T>), gN, blk, 0, s, D12, bN, S[3], n, nn, (const cx<T>*)nullptr, 0, 0L, M, n, nn, N, n, T(1), 0);
        // S11 = T1 D11
        TRX_LAUNCH((bd_colcomb_kernel<T>), gc, blk, 0, s, D11, bN, (const cx<T>*)Ys, 2 * n, 2 * nn, O[0], n, nn, N, n, T(1), 0);
        // S12 = Sn12 + T1 M
        TRX_LAUNCH((block_copy_kernel<T>), gn, blk, 0, s, S[2], n, nn, O[2], n, nn, n, n, T(1), 0);
        rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, n, n, one, Ys, 2 * n, 2 * nn, M, n, nn, one, O[2], n, nn, batch); if (rc) return rc;
        // S21 = D21 + D22 (T2 D11)
        TRX_LAUNCH((bd_colcomb_kernel<T>), gc, blk, 0, s, D11, bN, (const cx<T>*)(Ys + n), 2 * n, 2 * nn, Ta, n, nn, N, n, T(1), 0);
        TRX_LAUNCH((bd_dense_kernel<T>), gn, blk, 0, s, D21, bN, (const cx<T>*)nullptr, 0, 0L, O[1], n, nn, N);
        TRX_LAUNCH((bd_rowcomb_kernel<T>), gN, blk, 0, s, D22, bN, (const cx<T>*)Ta, n, nn, (const cx<T>*)O[1], n, nn, O[1], n, nn, N, n, T(1), 0);
        // S22 = D22 (Sn22 + T2 M)
        TRX_LAUNCH((block_copy_kernel<T>), gn, blk, 0, s, S[3], n, nn, Tb, n, nn, n, n, T(1), 0);
        rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, n, n, one, Ys + n, 2 * n, 2 * nn, M, n, nn, one, Tb, n, nn, batch); if (rc) return rc;
        TRX_LAUNCH((bd_rowcomb_kernel<T>), gN, blk, 0, s, D22, bN, (const cx<T>*)Tb, n, nn, (const cx<T>*)nullptr, 0, 0L, O[3], n, nn, N, n, T(1), 0);
    } else if (side == 0) {
        // Sm = half-space (block diagonal), Sn = S (dense)
        // K = I - Sm12 Sn21 ;  RHS = [Sm11 | Sm12 Sn22]
        TRX_LAUNCH((bd_rowcomb_kernel<T>), gN, blk, 0, s, D12, bN, S[1], n, nn, (const cx<T>*)nullptr, 0, 0L, K, n, nn, N, n, T(-1), 1);
        TRX_LAUNCH((bd_dense_kernel<T>), gn, blk, 0, s, D11, bN, (const cx<T>*)nullptr, 0, 0L, X, 2 * n, 2 * nn, N);
        TRX_LAUNCH((bd_rowcomb_kernel<T>), gN, blk, 0, s, D12, bN, S[3], n, nn, (const cx<T>*)nullptr, 0, 0L, X + n, 2 * n, 2 * nn, N, n, T(1), 0);
        rc = lu_factor<T>(s, K, n, nn, n, piv, batch, info); if (rc) return rc;
        rc = lu_solve<T>(s, K, n, nn, n, piv, X, 2 * n, 2 * nn, 2 * n, batch); if (rc) return rc;
        // S11 = Sn11 X1 ; S12 = Sn12 + Sn11 X2 ; Y1 = Sn21 X1 ; Y2 = Sn22 + Sn21 X2
        rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, n, n, one, S[0], n, nn, X, 2 * n, 2 * nn, zero, O[0], n, nn, batch); if (rc) return rc;
        TRX_LAUNCH((block_copy_kernel<T>), gn, blk, 0, s, S[2], n, nn, O[2], n, nn, n, n, T(1), 0);
        rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, n, n, one, S[0], n, nn, X + n, 2 * n, 2 * nn, one, O[2], n, nn, batch); if (rc) return rc;
        rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, n, n, one, S[1], n, nn, X, 2 * n, 2 * nn, zero, Y, 2 * n, 2 * nn, batch); if (rc) return rc;
        TRX_LAUNCH((block_copy_kernel<T>), gn, blk, 0, s, S[3], n, nn, Y + n, 2 * n, 2 * nn, n, n, T(1), 0);
        rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, n, n, one, S[1], n, nn, X + n, 2 * n, 2 * nn, one, Y + n, 2 * n, 2 * nn, batch); if (rc) return rc;
        // S21 = Sm21 + Sm22 Y1 ; S22 = Sm22 Y2          (row combinations)
        TRX_LAUNCH((bd_dense_kernel<T>), gn, blk, 0, s, D21, bN, (const cx<T>*)nullptr, 0, 0L, O[1], n, nn, N);
        TRX_LAUNCH((bd_rowcomb_kernel<T>), gN, blk, 0, s, D22, bN, (const cx<T>*)Y, 2 * n, 2 * nn, (const cx<T>*)O[1], n, nn, O[1], n, nn, N, n, T(1), 0);
        TRX_LAUNCH((bd_rowcomb_kernel<T>), gN, blk, 0, s, D22, bN, (const cx<T>*)(Y + n), 2 * n, 2 * nn, (const cx<T>*)nullptr, 0, 0L, O[3], n, nn, N, n, T(1), 0);
    } else {
        // Sm = S (dense), Sn = half-space (block diagonal)
        // K = I - Sm12 Sn21 (column combination) ;  RHS = [Sm11 | Sm12 Sn22]
        TRX_LAUNCH((bd_colcomb_kernel<T>), gc, blk, 0, s, D21, bN, S[2], n, nn, K, n, nn, N, n, T(-1), 1);
        TRX_LAUNCH((block_copy_kernel<T>), gn, blk, 0, s, S[0], n, nn, X, 2 * n, 2 * nn, n, n, T(1), 0);
        TRX_LAUNCH((bd_colcomb_kernel<T>), gc, blk, 0, s, D22, bN, S[2], n, nn, X + n, 2 * n, 2 * nn, N, n, T(1), 0);
        rc = lu_factor<T>(s, K, n, nn, n, piv, batch, info); if (rc) return rc;
        rc = lu_solve<T>(s, K, n, nn, n, piv, X, 2 * n, 2 * nn, 2 * n, batch); if (rc) return rc;
        // S11 = Sn11 X1 ; S12 = Sn12 + Sn11 X2 ; Y1 = Sn21 X1 ; Y2 = Sn22 + Sn21 X2   (all row combinations)
        TRX_LAUNCH((bd_rowcomb_kernel<T>), gN, blk, 0, s, D11, bN, (const cx<T>*)X, 2 * n, 2 * nn, (const cx<T>*)nullptr, 0, 0L, O[0], n, nn, N, n, T(1), 0);
        TRX_LAUNCH((bd_dense_kernel<T>), gn, blk, 0, s, D12, bN, (const cx<T>*)nullptr, 0, 0L, O[2], n, nn, N);
        TRX_LAUNCH((bd_rowcomb_kernel<T>), gN, blk, 0, s, D11, bN, (const cx<T>*)(X + n), 2 * n, 2 * nn, (const cx<T>*)O[2], n, nn, O[2], n, nn, N, n, T(1), 0);
        TRX_LAUNCH((bd_rowcomb_kernel<T>), gN, blk, 0, s, D21, bN, (const cx<T>*)X, 2 * n, 2 * nn, (const cx<T>*)nullptr, 0, 0L, Y, 2 * n, 2 * nn, N, n, T(1), 0);
        TRX_LAUNCH((bd_dense_kernel<T>), gn, blk, 0, s, D22, bN, (const cx<T>*)nullptr, 0, 0L, Y + n, 2 * n, 2 * nn, N);
        TRX_LAUNCH((bd_rowcomb_kernel<T>), gN, blk, 0, s, D21, bN, (const cx<T>*)(X + n), 2 * n, 2 * nn, (const cx<T>*)(Y + n), 2 * n, 2 * nn, Y + n, 2 * n, 2 * nn, N, n, T(1), 0);
        // S21 = Sm21 + Sm22 Y1 ; S22 = Sm22 Y2
        TRX_LAUNCH((block_copy_kernel<T>), gn, blk, 0, s, S[1], n, nn, O[1], n, nn, n, n, T(1), 0);
        rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, n, n, one, S[3], n, nn, Y, 2 * n, 2 * nn, one, O[1], n, nn, batch); if (rc) return rc;
        rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, n, n, one, S[3], n, nn, Y + n, 2 * n, 2 * nn, zero, O[3], n, nn, batch); if (rc) return rc;
    }
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

// A = P Q for a layer with homogeneous mu (rcwa.py:1236), from the block structure
//   A = [[mu E - Ky^2 - Kx Gx,  KxKy - Kx Gy], [KxKy - Ky Gx,  mu E - Kx^2 - Ky Gy]],  Gx = E^-1 (Kx E), Gy = E^-1 (Ky E)
// i.e. two N^3 products instead of one (2N)^3 product.
template <class T>
__global__ __launch_bounds__(256) void scale_rows_kernel(const cx<T>* __restrict__ in, const cx<T>* __restrict__ s, int N, cx<T>* __restrict__ out) {
    const int b = blockIdx.z, i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    const long o = ((long)b * N + i) * N + j;
    out[o] = s[(long)b * N + i] * in[o];
}
template <class T>
__global__ __launch_bounds__(256) void assemble_a_kernel(const cx<T>* __restrict__ E, const cx<T>* __restrict__ Gx, const cx<T>* __restrict__ Gy,
                                                         const cx<T>* __restrict__ mu, const cx<T>* __restrict__ kx, const cx<T>* __restrict__ ky, int N,
                                                         cx<T>* __restrict__ A) {
    const int b = blockIdx.z, i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    const long o = ((long)b * N + i) * N + j;
    const cx<T> kxi = kx[(long)b * N + i], kyi = ky[(long)b * N + i];
    const cx<T> e = mu[b] * E[o], gx = Gx[o], gy = Gy[o];
    const int n = 2 * N;
    cx<T>* Ab = A + (long)b * n * n;
    cx<T> a11 = e - kxi * gx, a12 = -(kxi * gy), a21 = -(kyi * gx), a22 = e - kyi * gy;
    if (i == j) { a11 -= kyi * kyi; a22 -= kxi * kxi; a12 += kxi * kyi; a21 += kxi * kyi; }
    Ab[(long)i * n + j] = a11;
    Ab[(long)i * n + j + N] = a12;
    Ab[(long)(i + N) * n + j] = a21;
    Ab[(long)(i + N) * n + j + N] = a22;
}
template <class T>
int build_a_t(hipStream_t s, const cx<T>* E, const cx<T>* Ei, const cx<T>* mu, const cx<T>* kx, const cx<T>* ky, int N, int batch, cx<T>* A, cx<T>* ws) {
    const long NN = (long)N * N, bNN = (long)batch * NN;
    const cx<T> one(T(1), T(0)), zero(T(0), T(0));
    const dim3 g(cdiv_i(N, 256), N, batch), blk(256);
    cx<T>*Sx = ws, *Gx = ws + bNN, *Gy = ws + 2 * bNN;
    TRX_LAUNCH((scale_rows_kernel<T>), g, blk, 0, s, E, kx, N, Sx);
    int rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, N, N, N, one, Ei, N, NN, Sx, N, NN, zero, Gx, N, NN, batch); if (rc) return rc;
    TRX_LAUNCH((scale_rows_kernel<T>), g, blk, 0, s, E, ky, N, Sx);
    rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, N, N, N, one, Ei, N, NN, Sx, N, NN, zero, Gy, N, NN, batch); if (rc) return rc;
    TRX_LAUNCH((assemble_a_kernel<T>), g, blk, 0, s, E, (const cx<T>*)Gx, (const cx<T>*)Gy, mu, kx, ky, N, A);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

}  // namespace
}  // namespace trx

using namespace trx;

extern "C" int trx_build_pq(int dtype, const void* E, const void* Einv, const void* Mu, const void* Muinv, const void* kx,
                            const void* ky, int N, int batch, void* P, void* Q, void* stream) {
    if (!E || !Einv || !Mu || !Muinv || !kx || !ky || !P || !Q || N <= 0 || batch <= 0) return TRX_ERR_ARG;
    hipStream_t s = trx::api_stream(stream);
    if (dtype == TRX_C64) return build_pq_t<float>(s, E, Einv, Mu, Muinv, kx, ky, N, batch, P, Q);
    if (dtype == TRX_C128) return build_pq_t<double>(s, E, Einv, Mu, Muinv, kx, ky, N, batch, P, Q);
    return TRX_ERR_DTYPE;
}

extern "C" size_t trx_layer_smatrix_ws_bytes(int dtype, int N, int batch) {
    return (size_t)(dtype == TRX_C128 ? 16 : 8) * 6 * (size_t)batch * (2 * (size_t)N) * (2 * (size_t)N);
}
extern "C" size_t trx_layer_smatrix_ws_bytes_lean(int dtype, int N, int batch) {
    return (size_t)(dtype == TRX_C128 ? 16 : 8) * 4 * (size_t)batch * (2 * (size_t)N) * (2 * (size_t)N);
}

extern "C" int trx_layer_smatrix(int dtype, const void* P, const void* Q, const void* W, const void* kzfac, const void* vfinv,
                                 const void* phase, int use_q, int N, int batch, void* S11, void* S21, void* V, void* Cplus,
                                 void* Cminus, int* piv, int* info, void* ws, size_t ws_bytes, void* stream) {
    if (!W || !kzfac || !vfinv || !phase || !S11 || !S21 || !V || !piv || !info || !ws || N <= 0 || batch <= 0) return TRX_ERR_ARG;
    if (use_q < 0 || use_q > 2 || (use_q == 1 && !Q) || (use_q == 0 && !P)) return TRX_ERR_ARG;
    if ((Cplus == nullptr) != (Cminus == nullptr)) return TRX_ERR_ARG;
    {
        const size_t esz = dtype == TRX_C128 ? 16 : 8, blk = esz * (size_t)batch * (2 * (size_t)N) * (2 * (size_t)N);
        const bool lean = !Cplus && (char*)S21 == (char*)S11 + blk;       // outputs double as the last scratch block
        if (ws_bytes < (lean ? trx_layer_smatrix_ws_bytes_lean(dtype, N, batch) : trx_layer_smatrix_ws_bytes(dtype, N, batch))) return TRX_ERR_WORKSPACE;
    }
    hipStream_t s = trx::api_stream(stream);
    if (dtype == TRX_C64)
        return layer_smatrix_t<float>(s, (const cx<float>*)P, (const cx<float>*)Q, (const cx<float>*)W, (const cx<float>*)kzfac, (const cx<float>*)vfinv,
                                      (const cx<float>*)phase, use_q, N, batch, (cx<float>*)S11, (cx<float>*)S21, (cx<float>*)V, (cx<float>*)Cplus,
                                      (cx<float>*)Cminus, piv, info, (cx<float>*)ws);
    if (dtype == TRX_C128)
        return layer_smatrix_t<double>(s, (const cx<double>*)P, (const cx<double>*)Q, (const cx<double>*)W, (const cx<double>*)kzfac, (const cx<double>*)vfinv,
                                       (const cx<double>*)phase, use_q, N, batch, (cx<double>*)S11, (cx<double>*)S21, (cx<double>*)V, (cx<double>*)Cplus,
                                       (cx<double>*)Cminus, piv, info, (cx<double>*)ws);
    return TRX_ERR_DTYPE;
}

extern "C" size_t trx_hmodes_ws_bytes(int dtype, int N, int batch) {
    return (size_t)(dtype == TRX_C128 ? 16 : 8) * 3 * (size_t)batch * N * N;
}

extern "C" int trx_hmodes(int dtype, const void* E, const void* mu, const void* kx, const void* ky, const void* W, const void* kz, int N, int batch,
                          void* V, int* piv, int* info, void* ws, size_t ws_bytes, void* stream) {
    if (!E || !mu || !kx || !ky || !W || !kz || !V || !piv || !info || !ws || N <= 0 || batch <= 0) return TRX_ERR_ARG;
    if (ws_bytes < trx_hmodes_ws_bytes(dtype, N, batch)) return TRX_ERR_WORKSPACE;
    hipStream_t s = trx::api_stream(stream);
    if (dtype == TRX_C64)
        return hmodes_t<float>(s, (const cx<float>*)E, (const cx<float>*)mu, (const cx<float>*)kx, (const cx<float>*)ky, (const cx<float>*)W,
                               (const cx<float>*)kz, N, batch, (cx<float>*)V, piv, info, (cx<float>*)ws);
    if (dtype == TRX_C128)
        return hmodes_t<double>(s, (const cx<double>*)E, (const cx<double>*)mu, (const cx<double>*)kx, (const cx<double>*)ky, (const cx<double>*)W,
                                (const cx<double>*)kz, N, batch, (cx<double>*)V, piv, info, (cx<double>*)ws);
    return TRX_ERR_DTYPE;
}

extern "C" size_t trx_redheffer_ws_bytes(int dtype, int n, int batch) {
    return (size_t)(dtype == TRX_C128 ? 16 : 8) * (size_t)batch * (size_t)n * n;
}

extern "C" int trx_redheffer(int dtype, const void* const* Sm, const void* const* Sn, void* const* Sout, void* XY, int n, int batch,
                             int* piv, int* info, void* ws, size_t ws_bytes, void* stream) {
    if (!Sm || !Sn || !Sout || !XY || !piv || !info || !ws || n <= 0 || batch <= 0) return TRX_ERR_ARG;
    for (int k = 0; k < 4; ++k)
        if (!Sm[k] || !Sn[k] || !Sout[k]) return TRX_ERR_ARG;
    if (ws_bytes < trx_redheffer_ws_bytes(dtype, n, batch)) return TRX_ERR_WORKSPACE;
    hipStream_t s = trx::api_stream(stream);
    if (dtype == TRX_C64) return redheffer_t<float>(s, (const cx<float>* const*)Sm, (const cx<float>* const*)Sn, (cx<float>* const*)Sout, (cx<float>*)XY, n, batch, piv, info, (cx<float>*)ws);
    if (dtype == TRX_C128) return redheffer_t<double>(s, (const cx<double>* const*)Sm, (const cx<double>* const*)Sn, (cx<double>* const*)Sout, (cx<double>*)XY, n, batch, piv, info, (cx<double>*)ws);
    return TRX_ERR_DTYPE;
}

extern "C" size_t trx_redheffer_halfspace_ws_bytes(int dtype, int N, int batch, int side, int want_xy) {
    const size_t nn = (size_t)(dtype == TRX_C128 ? 16 : 8) * (size_t)batch * (size_t)(2 * N) * (size_t)(2 * N);
    return (side == 0 && !want_xy) ? 4 * nn : nn;
}

extern "C" int trx_redheffer_halfspace(int dtype, int side, const void* bd, const void* const* S, void* const* Sout, void* XY, int N, int batch,
                                       int* piv, int* info, void* ws, size_t ws_bytes, void* stream) {
    if (!bd || !S || !Sout || !piv || !info || !ws || N <= 0 || batch <= 0 || (side != 0 && side != 1)) return TRX_ERR_ARG;
    if (!XY && side == 1) return TRX_ERR_ARG;            // the side-1 algebra produces the factors anyway: XY is its scratch
    for (int k = 0; k < 4; ++k)
        if (!S[k] || !Sout[k]) return TRX_ERR_ARG;
    if (ws_bytes < trx_redheffer_halfspace_ws_bytes(dtype, N, batch, side, XY != nullptr)) return TRX_ERR_WORKSPACE;
    hipStream_t s = trx::api_stream(stream);
    if (dtype == TRX_C64) return redheffer_halfspace_t<float>(s, side, (const cx<float>*)bd, (const cx<float>* const*)S, (cx<float>* const*)Sout, (cx<float>*)XY, N, batch, piv, info, (cx<float>*)ws);
    if (dtype == TRX_C128) return redheffer_halfspace_t<double>(s, side, (const cx<double>*)bd, (const cx<double>* const*)S, (cx<double>* const*)Sout, (cx<double>*)XY, N, batch, piv, info, (cx<double>*)ws);
    return TRX_ERR_DTYPE;
}

extern "C" size_t trx_build_a_ws_bytes(int dtype, int N, int batch) {
    return (size_t)(dtype == TRX_C128 ? 16 : 8) * 3 * (size_t)batch * N * N;
}

extern "C" int trx_build_a(int dtype, const void* E, const void* Einv, const void* mu, const void* kx, const void* ky, int N, int batch, void* A,
                           void* ws, size_t ws_bytes, void* stream) {
    if (!E || !Einv || !mu || !kx || !ky || !A || !ws || N <= 0 || batch <= 0) return TRX_ERR_ARG;
    if (ws_bytes < trx_build_a_ws_bytes(dtype, N, batch)) return TRX_ERR_WORKSPACE;
    hipStream_t s = trx::api_stream(stream);
    if (dtype == TRX_C64) return build_a_t<float>(s, (const cx<float>*)E, (const cx<float>*)Einv, (const cx<float>*)mu, (const cx<float>*)kx, (const cx<float>*)ky, N, batch, (cx<float>*)A, (cx<float>*)ws);
    if (dtype == TRX_C128) return build_a_t<double>(s, (const cx<double>*)E, (const cx<double>*)Einv, (const cx<double>*)mu, (const cx<double>*)kx, (const cx<double>*)ky, N, batch, (cx<double>*)A, (cx<double>*)ws);
    return TRX_ERR_DTYPE;
}
