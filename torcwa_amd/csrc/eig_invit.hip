// Eigenvectors of the Hessenberg matrix by inverse iteration -- the eigenvector stage of the replacement for torch.linalg.eig
// (torcwa/torch_eig.py:14) when the QR phase runs for EIGENVALUES ONLY (LAPACK's zhsein route instead of zhseqr('V') + ztrevc).
//
// Why: with Schur vectors every window step of the QR iteration updates 2n - 64 rows / columns (H right of and above the window,
// all of Z); for eigenvalues alone only the active diagonal block matters, a third of the work on average, and no Z.  The vectors
// then cost one O(n^2) solve (H - lam I) x = b per eigenvalue -- n^3 complex MACs per matrix, all of them independent.
//
// One solve, with O(n) state.  LAPACK's zlaein factors H - lam I with row operations and keeps the factor (n^2 per eigenvalue).  Here
// the elimination runs over COLUMNS from the bottom, so the triangular solve can be interleaved with it and only two vectors live:
//     M = H - lam I.   For j = n-1 .. 1:   q = running column j (rows 0..j),  p = M[0:j+1, j-1].
//         pivot = the larger of |q_j|, |p_j| (column interchange if it is p);  f = pivot column,  g' = g - (g_j / f_j) f   (row j of g' = 0)
//         f is column j of R (M C = R upper triangular, C = the column operations):   y_j = b_j / f_j,   b[0:j] -= f[0:j] y_j;   q <- g'
//     y_0 = b_0 / q_0;   x = C y  is the O(n) recurrence  z_j <- z_j - m_j z_{j-1}, interchange (j-1, j)  run upwards  (invit_back_kernel).
// The singularity of M (lam is an eigenvalue to working accuracy) shows up as a tiny pivot, replaced by eps ||H|| as in zlaein; x then
// grows by ~1 / (eps ||H||) along the eigenvector.  b is a fixed pseudo-random vector that differs from eigenvalue to eigenvalue, so that
// equal eigenvalues (degenerate pairs of symmetric meta-atoms) get independent vectors of their common eigenspace.
//
// Mapping to gfx950.  Lanes run along the ROWS of the two vectors (q, b in registers: ISL slots of 64 WPL lanes per eigenvalue, WPL waves
// per eigenvalue chosen by n), 16 / WPL eigenvalues per workgroup share every column of H: the workgroup stages column j-1 (a row of the
// transposed copy Ht, contiguous) into LDS while step j computes on its double-buffered predecessor; the pivot row of the next step is
// published through LDS by the lane that owns it.  One barrier per step.  All fp64 vector FMAs: the matrix cores have nothing to offer a
// recurrence whose multiplier changes with every step, and on this chip the fp64 vector rate equals the matrix rate.
#include "eig.hpp"
#include <limits>
#include "prof.hpp"

namespace trx {
namespace {

constexpr int IVT = 1024;     // threads per workgroup (16 waves: 4 per SIMD, i.e. at most 128 VGPRs)
// slots (rows per lane): the two vectors of one eigenvalue take 2 x ISL complex numbers per lane = 64 VGPRs in either precision
template <class T> struct IvSlots { static constexpr int value = sizeof(T) == 8 ? 8 : 16; };
constexpr int IVW = IVT / 64; // waves per workgroup

template <class T>
struct IvPivot {
    cx<T> q, b;
};

// start vector of eigenvalue k, row i: modulus in [0.5, 1.5], pseudo-random phase-like pattern (integer hash, no trigonometry)
template <class T>
__device__ __forceinline__ cx<T> invit_start(int i, int k) {
    unsigned h = (unsigned)i * 2654435761u ^ ((unsigned)k * 2246822519u + 0x9e3779b9u);
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    const T a = (T)(int)(h & 0xffff) * (T)(1.0 / 65536.0) - T(0.5);
    const T c = (T)(int)(h >> 16) * (T)(1.0 / 65536.0) - T(0.5);
    return cx<T>(T(1) + a, c);
}

// out[b] = in[b]^T (plain transpose), 32x32 tiles
template <class T>
__global__ __launch_bounds__(256) void invit_transpose_kernel(const cx<T>* __restrict__ in, cx<T>* __restrict__ out, int n) {
    __shared__ cx<T> tile[32][33];
    in += (long)blockIdx.z * n * n;
    out += (long)blockIdx.z * n * n;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + threadIdx.x;
        if (r < n && c < n) tile[i][threadIdx.x] = in[(long)r * n + c];
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int r = c0 + i, c = r0 + threadIdx.x;
        if (r < n && c < n) out[(long)r * n + c] = tile[threadIdx.x][i];
    }
}

// hnorm[b] = infinity norm of the (Hessenberg) matrix b: one workgroup per matrix, a wave per row
template <class T>
__global__ __launch_bounds__(256) void invit_norm_kernel(const cx<T>* __restrict__ Hall, int n, T* __restrict__ hnorm) {
    __shared__ T red[4];
    const cx<T>* H = Hall + (long)blockIdx.x * n * n;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    T best = T(0);
    for (int r = w; r < n; r += 4) {
        T s = T(0);
        for (int c = (r > 0 ? r - 1 : 0) + lane; c < n; c += 64) s += abs1(H[(long)r * n + c]);
        s = wave_sum(s);
        best = s > best ? s : best;
    }
    if (lane == 0) red[w] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        T m = red[0];
        for (int i = 1; i < 4; ++i) m = red[i] > m ? red[i] : m;
        hnorm[blockIdx.x] = m;
    }
}

// w[b, i] = A[b, i, i]
template <class T>
__global__ __launch_bounds__(256) void invit_diag_kernel(const cx<T>* __restrict__ Aall, cx<T>* __restrict__ w, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) w[(long)blockIdx.y * n + i] = Aall[((long)blockIdx.y * n + i) * n + i];
}

// The steps j = jhi .. jlo of one eigenvalue's elimination, all with rows 0..j-1 inside the slots 0..S (S compile time: the register
// arrays q, b are only ever indexed by constants).
template <class T, int WPL, int S>
__device__ __forceinline__ void invit_block(int jhi, int jlo, cx<T> (&q)[IvSlots<T>::value], cx<T> (&b)[IvSlots<T>::value], const cx<T>* __restrict__ Ht, int n, cx<T>* pbuf, int pstride,
                                            IvPivot<T>* pv, int ll, int L, cx<T> lam, T eps3, bool writer, cx<T>* __restrict__ Yk, cx<T>* __restrict__ Mk,
                                            unsigned char* __restrict__ SWk, bool single) {
    constexpr int ISL = IvSlots<T>::value, LW = 64 * WPL, LPW = IVW / WPL, PF = (LW * ISL + IVT - 1) / IVT;
    const int t = threadIdx.x;
    for (int j = jhi; j >= jlo; --j) {
        const int par = j & 1;
        const cx<T>* P = pbuf + (single ? 0 : par * pstride);          // column j-1 of H: rows 0..j
        // prefetch column j-2 (rows 0..j-1) for the next step
        cx<T> pf[PF];
        const cx<T>* src = Ht + (long)(j >= 2 ? j - 2 : 0) * n;
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int e = t + IVT * k;
            pf[k] = (j >= 2 && e < j) ? src[e] : cx<T>(T(0), T(0));
        }
        const IvPivot<T> pvv = pv[par * LPW + ll];
        const cx<T> pj = P[j];
        const int swap = __builtin_amdgcn_readfirstlane((int)(abs1(pj) > abs1(pvv.q)));
        cx<T> piv = swap ? pj : pvv.q;
        const cx<T> oth = swap ? pvv.q : pj;
        if (abs1(piv) < eps3) piv = cx<T>(eps3, T(0));
        const cx<T> rp = crecip(piv);
        const cx<T> m = oth * rp, yj = pvv.b * rp;
        const int Lp = (j - 1) - S * LW;                                 // lane that owns row j-1 (slot S)
        if (swap) {
#pragma unroll
            for (int s = 0; s <= S; ++s) {
                const int i = s * LW + L;
                cx<T> pi = P[i < j ? i : j];
                if (s == S && L == Lp) pi = pi - lam;
                cx<T> qn = q[s], bn = b[s];
                cfma(qn, -m, pi);                                       // g' = q - m p
                cfma(bn, -pi, yj);                                      // b -= p y_j
                q[s] = qn; b[s] = bn;
            }
        } else {
#pragma unroll
            for (int s = 0; s <= S; ++s) {
                const int i = s * LW + L;
                cx<T> pi = P[i < j ? i : j];
                if (s == S && L == Lp) pi = pi - lam;
                const cx<T> qo = q[s];
                cx<T> bn = b[s];
                cfma(pi, -m, qo);                                       // g' = p - m q
                cfma(bn, -qo, yj);                                      // b -= q y_j
                q[s] = pi; b[s] = bn;
            }
        }
        if (writer) {
            Yk[(long)j * n] = yj;
            Mk[(long)j * n] = m;
            SWk[(long)j * n] = (unsigned char)swap;
        }
        if (single) __syncthreads();                                    // one column buffer: everybody is done reading it
        if (L == Lp) { IvPivot<T> o; o.q = q[S]; o.b = b[S]; pv[(par ^ 1) * LPW + ll] = o; }
        cx<T>* Pn = pbuf + (single ? 0 : (par ^ 1) * pstride);
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int e = t + IVT * k;
            if (j >= 2 && e < j) Pn[e] = pf[k];
        }
        __syncthreads();
    }
}

// grid (ceil(n / LPW), batch).  Y[j, k], M[j, k], SW[j, k]: y_j, multiplier and interchange flag of step j of eigenvalue k.
template <class T, int WPL>
__global__ __launch_bounds__(IVT) void invit_solve_kernel(const cx<T>* __restrict__ Ht_all, int n, const cx<T>* __restrict__ lam_all, const T* __restrict__ hnorm,
                                                          cx<T>* __restrict__ Yall, cx<T>* __restrict__ Mall, unsigned char* __restrict__ SWall, int single) {
    TRX_DYN_SMEM(smem);
    constexpr int ISL = IvSlots<T>::value, LW = 64 * WPL, LPW = IVW / WPL;
    const int pstride = n;
    cx<T>* pbuf = reinterpret_cast<cx<T>*>(smem);                              // [2 or 1][n]
    IvPivot<T>* pv = reinterpret_cast<IvPivot<T>*>(pbuf + (single ? 1 : 2) * (size_t)n);     // [2][LPW]
    const int bm = blockIdx.y, t = threadIdx.x;
    const cx<T>* Ht = Ht_all + (long)bm * n * n;
    const int ll = t / LW, L = t - ll * LW;                                    // eigenvalue within the workgroup, lane within the eigenvalue
    int k = blockIdx.x * LPW + ll;
    const bool valid = k < n;
    if (!valid) k = n - 1;                                                      // a padding group repeats the last eigenvalue and writes nothing
    const cx<T> lam = lam_all[(long)bm * n + k];
    T eps3 = eps_of<T>::value * hnorm[bm];
    if (!(eps3 > eps_of<T>::safmin)) eps3 = eps_of<T>::safmin;
    const bool writer = valid && L == 0;
    cx<T>* Yk = Yall + (long)bm * n * n + k;
    cx<T>* Mk = Mall + (long)bm * n * n + k;
    unsigned char* SWk = SWall + (long)bm * n * n + k;
    cx<T> q[ISL], b[ISL];
    // column n-1 of M and the start vector
    {
        const cx<T>* src = Ht + (long)(n - 1) * n;
#pragma unroll
        for (int s = 0; s < ISL; ++s) {
            const int i = s * LW + L;
            cx<T> v(T(0), T(0)), bv(T(0), T(0));
            if (i < n) { v = src[i]; bv = invit_start<T>(i, k); }
            if (i == n - 1) v = v - lam;
            q[s] = v; b[s] = bv;
        }
    }
    if (n == 1) {
        if (writer) { Yk[0] = cx<T>(T(1), T(0)); }
        return;
    }
    // stage column n-2 (rows 0..n-1) and publish row n-1
    {
        const int par = (n - 1) & 1;
        cx<T>* Pn = pbuf + (single ? 0 : par * pstride);
        const cx<T>* src = Ht + (long)(n - 2) * n;
        for (int e = t; e < n; e += IVT) Pn[e] = src[e];
        const int so = (n - 1) / LW, Lo = (n - 1) - so * LW;
        if (L == Lo) {
            IvPivot<T> o;
            o.q = cx<T>(T(0), T(0)); o.b = cx<T>(T(0), T(0));
#pragma unroll
            for (int s = 0; s < ISL; ++s)
                if (s == so) { o.q = q[s]; o.b = b[s]; }
            pv[par * LPW + ll] = o;
        }
    }
    __syncthreads();
    const int stop = (n - 2) / LW;
    for (int S = stop; S >= 0; --S) {
        const int jlo = S * LW + 1;
        const int jhi = (S + 1) * LW < n - 1 ? (S + 1) * LW : n - 1;
#define TRX_IV_CASE(SS) case SS: invit_block<T, WPL, SS>(jhi, jlo, q, b, Ht, n, pbuf, pstride, pv, ll, L, lam, eps3, writer, Yk, Mk, SWk, single != 0); break;
        switch (S) {
            TRX_IV_CASE(0) TRX_IV_CASE(1) TRX_IV_CASE(2) TRX_IV_CASE(3) TRX_IV_CASE(4) TRX_IV_CASE(5) TRX_IV_CASE(6) TRX_IV_CASE(7)
            default:
                if constexpr (ISL > 8) {
                    switch (S) {
                        TRX_IV_CASE(8) TRX_IV_CASE(9) TRX_IV_CASE(10) TRX_IV_CASE(11) TRX_IV_CASE(12) TRX_IV_CASE(13) TRX_IV_CASE(14) TRX_IV_CASE(15)
                        default: break;
                    }
                }
                break;
        }
#undef TRX_IV_CASE
    }
    // j = 0: the last pivot carries the singularity
    if (writer) {
        const IvPivot<T> pvv = pv[0 * LPW + ll];
        cx<T> piv = pvv.q;
        if (abs1(piv) < eps3) piv = cx<T>(eps3, T(0));
        Yk[0] = pvv.b * crecip(piv);
    }
}

// x = C y, one thread per eigenvalue (coalesced across eigenvalues), in place on Y; then the column is scaled to unit maximum modulus
// (the growth of the solve is ~ 1 / (eps ||H||): kept away from the GEMM that follows, whatever the arithmetic type).
template <class T>
__global__ __launch_bounds__(256) void invit_back_kernel(cx<T>* __restrict__ Yall, const cx<T>* __restrict__ Mall, const unsigned char* __restrict__ SWall, int n) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    cx<T>* Y = Yall + (long)blockIdx.y * n * n + k;
    const cx<T>* M = Mall + (long)blockIdx.y * n * n + k;
    const unsigned char* SW = SWall + (long)blockIdx.y * n * n + k;
    cx<T> c = Y[0];
    T big = T(0);
    for (int j = 1; j < n; ++j) {
        cx<T> zj = Y[(long)j * n];
        cfma(zj, -M[(long)j * n], c);
        cx<T> out;
        if (SW[(long)j * n]) { out = zj; } else { out = c; c = zj; }
        Y[(long)(j - 1) * n] = out;
        const T a = abs1(out);
        big = a > big ? a : big;
    }
    Y[(long)(n - 1) * n] = c;
    { const T a = abs1(c); big = a > big ? a : big; }
    const T sc = (big > T(0) && big < std::numeric_limits<T>::infinity()) ? T(1) / big : T(1);
    for (int j = 0; j < n; ++j) Y[(long)j * n] = sc * Y[(long)j * n];
}

template <class T, int WPL>
int launch_solve(hipStream_t s, const cx<T>* Ht, int n, const cx<T>* w, const T* hnorm, cx<T>* Y, cx<T>* M, unsigned char* SW, int batch) {
    constexpr int LPW = IVW / WPL;
    // double-buffered column staging when two columns fit into LDS next to the pivot slots, else one buffer and a second barrier per step
    const size_t two = sizeof(cx<T>) * 2 * (size_t)n + sizeof(IvPivot<T>) * 2 * LPW, one = sizeof(cx<T>) * (size_t)n + sizeof(IvPivot<T>) * 2 * LPW;
    const int single = two > 150 * 1024;
    const size_t sm = single ? one : two;
    if (set_max_dyn_smem((const void*)invit_solve_kernel<T, WPL>, sm)) return TRX_ERR_LAUNCH;
    ProfScope prof(PROF_INVIT, s, 8.0 * (double)n * n * n * batch, 0.0);
    TRX_LAUNCH((invit_solve_kernel<T, WPL>), dim3(cdiv_i(n, LPW), batch), dim3(IVT), sm, s, Ht, n, w, hnorm, Y, M, SW, single);
    return TRX_OK;
}

}  // namespace

// A holds T's diagonal (eigenvalues) after the eigenvalue-only QR phase; Ht the transposed copy of the Hessenberg matrix taken before it;
// Z the unitary of the Hessenberg reduction.  Writes w and V (unit 2-norm columns, balancing undone).
template <class T>
int invit_vectors(hipStream_t s, const EigBuffers<T>& B, int n, int batch, cx<T>* w, cx<T>* V) {
    const cx<T> one(T(1), T(0)), zero(T(0), T(0));
    const long nn = (long)n * n;
    TRX_LAUNCH((invit_diag_kernel<T>), dim3(cdiv_i(n, 256), batch), dim3(256), 0, s, (const cx<T>*)B.A, w, n);
    cx<T>* Y = B.X;                    // y, then x (eigenvectors of H), [n rows, n eigenvalues]
    cx<T>* M = V;                      // multipliers: the output buffer is free until the back-transform writes it
    int rc;
    const int rows1 = 64 * IvSlots<T>::value;          // rows one wave holds
    if (n <= rows1) rc = launch_solve<T, 1>(s, B.Ht, n, w, B.hnorm, Y, M, B.SW, batch);
    else if (n <= 2 * rows1) rc = launch_solve<T, 2>(s, B.Ht, n, w, B.hnorm, Y, M, B.SW, batch);
    else if (n <= 4 * rows1) rc = launch_solve<T, 4>(s, B.Ht, n, w, B.hnorm, Y, M, B.SW, batch);
    else if (n <= 8 * rows1) rc = launch_solve<T, 8>(s, B.Ht, n, w, B.hnorm, Y, M, B.SW, batch);
    else if (n <= 16 * rows1) rc = launch_solve<T, 16>(s, B.Ht, n, w, B.hnorm, Y, M, B.SW, batch);
    else return TRX_ERR_ARG;
    if (rc) return rc;
    TRX_LAUNCH((invit_back_kernel<T>), dim3(cdiv_i(n, 256), batch), dim3(256), 0, s, Y, (const cx<T>*)M, (const unsigned char*)B.SW, n);
    rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, n, n, one, B.Z, n, nn, Y, n, nn, zero, V, n, nn, batch);
    if (rc) return rc;
    return finish_vectors<T>(s, B, n, batch, V);
}

// transposed copy + norm of the Hessenberg matrix (before the QR phase overwrites it)
template <class T>
int invit_prepare(hipStream_t s, const EigBuffers<T>& B, int n, int batch) {
    TRX_LAUNCH((invit_transpose_kernel<T>), dim3(cdiv_i(n, 32), cdiv_i(n, 32), batch), dim3(32, 8), 0, s, (const cx<T>*)B.A, B.Ht, n);
    TRX_LAUNCH((invit_norm_kernel<T>), dim3(batch), dim3(256), 0, s, (const cx<T>*)B.A, n, B.hnorm);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

template int invit_vectors<float>(hipStream_t, const EigBuffers<float>&, int, int, cx<float>*, cx<float>*);
template int invit_vectors<double>(hipStream_t, const EigBuffers<double>&, int, int, cx<double>*, cx<double>*);
template int invit_prepare<float>(hipStream_t, const EigBuffers<float>&, int, int);
template int invit_prepare<double>(hipStream_t, const EigBuffers<double>&, int, int);

}  // namespace trx
