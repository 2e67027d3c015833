// Batched blocked Householder reduction to upper Hessenberg form, A = Q H Q^H, with Q accumulated into Z.
// First stage of the eigendecomposition that replaces torch.linalg.eig (torcwa/torch_eig.py:14, rcwa.py:1236-1238).
//
// Panel algebra follows LAPACK zgehrd/zlahr2 (compact WY: Q_panel = I - V T V^H, Y = A V T).  The batch is
// processed in lock-step, so every launch streams `batch` trailing matrices at once:
//   hess_col_kernel   one workgroup per matrix: finalises Y/T of the previous panel column, updates the current
//                     column with the pending block reflector, generates its Householder vector.
//   hess_gemv_kernel  the irreducible BLAS-2 stream y = A[R, j+1:n] v  (one wave per row, coalesced row reads,
//                     v broadcast from LDS)  -- HBM-bound, n^3/3 element reads per matrix in total.
//   gemm              all block updates (right/left trailing updates and the Z accumulation).
#include "eig.hpp"
#include <cstdlib>
#include "prof.hpp"

namespace trx {
namespace {

constexpr int HNB = EigPlan::HNB;   // panel width
constexpr int HCT = 512;            // threads of the reflector kernel (one workgroup per matrix; 1024 measured no faster)
constexpr int HRG = HCT / HNB;      // row groups of its HNB x HRG thread grid

template <class T>
__global__ __launch_bounds__(HCT) void hess_col_kernel(cx<T>* __restrict__ Aall, int n, int p0, int ib, int c,
                                                        cx<T>* __restrict__ Vall, cx<T>* __restrict__ Yall,
                                                        cx<T>* __restrict__ Tall, cx<T>* __restrict__ tau_all) {
    TRX_DYN_SMEM(smem);
    cx<T>* bcol = reinterpret_cast<cx<T>*>(smem);     // [n]   current column (rows p0+1..n-1 at index r-(p0+1))
    cx<T>* part = bcol + n;                            // [HRG][HNB] partial sums
    cx<T>* vec = part + HRG * HNB;                     // [HNB]  t / w vectors
    T* red = reinterpret_cast<T*>(vec + HNB);          // [16] scalar reduction scratch
    const int b = blockIdx.x;
    cx<T>* A = Aall + (long)b * n * n;
    cx<T>* V = Vall + (long)b * n * HNB;
    cx<T>* Y = Yall + (long)b * n * HNB;
    cx<T>* Tm = Tall + (long)b * HNB * HNB;
    cx<T>* tau = tau_all + (long)b * HNB;
    const int t = threadIdx.x;
    const int cc = t & (HNB - 1), rg = t / HNB;        // HNB x HRG thread grid
    const int r0 = p0 + 1;                             // first row of R
    const int nr = n - r0;

    // ---- part 1: finalise column cp = c-1 of Y and T -------------------------------------------------------
    if (c > 0) {
        const int cp = c - 1, jp = p0 + cp;
        const cx<T> tau_p = tau[cp];
        // t[q] = sum_{r > jp} conj(V[r,q]) * V[r,cp],  q < cp
        cx<T> acc(T(0), T(0));
        if (cc < cp)
            for (int r = jp + 1 + rg; r < n; r += HRG) cfma_conj(acc, V[(long)r * HNB + cc], V[(long)r * HNB + cp]);
        part[rg * HNB + cc] = acc;
        __syncthreads();
        if (t < HNB) {
            cx<T> s(T(0), T(0));
            for (int g = 0; g < HRG; ++g) s += part[g * HNB + t];
            vec[t] = (t < cp) ? s : cx<T>(T(0), T(0));
        }
        __syncthreads();
        // Y[r,cp] = tau * (Yraw[r,cp] - sum_q Y[r,q] t[q]),  r in R
        for (int r = r0 + t; r < n; r += blockDim.x) {
            cx<T> y = Y[(long)r * HNB + cp];
            for (int q = 0; q < cp; ++q) cfma(y, -Y[(long)r * HNB + q], vec[q]);
            Y[(long)r * HNB + cp] = tau_p * y;
        }
        // T[0:cp, cp] = -tau * T[0:cp,0:cp] t ;  T[cp,cp] = tau ; below-diagonal entries stay zero
        if (t < HNB) {
            cx<T> s(T(0), T(0));
            if (t < cp) {
                for (int q = t; q < cp; ++q) cfma(s, Tm[t * HNB + q], vec[q]);
                s = -(tau_p * s);
            } else if (t == cp) {
                s = tau_p;
            }
            Tm[t * HNB + cp] = s;
        }
        __syncthreads();
    }
    if (c >= ib) return;

    // ---- part 2: update column j with the pending reflectors of this panel, then build its reflector -------
    const int j = p0 + c;
    for (int i = t; i < nr; i += blockDim.x) bcol[i] = A[(long)(r0 + i) * n + j];
    __syncthreads();
    if (c > 0) {
        // b -= Y[R,0:c] conj(V[j,0:c])
        for (int i = t; i < nr; i += blockDim.x) {
            cx<T> v = bcol[i];
            for (int q = 0; q < c; ++q) cfma(v, -Y[(long)(r0 + i) * HNB + q], conj(V[(long)j * HNB + q]));
            bcol[i] = v;
        }
        __syncthreads();
        // w = V[R,0:c]^H b
        cx<T> acc(T(0), T(0));
        if (cc < c)
            for (int i = rg; i < nr; i += HRG) cfma_conj(acc, V[(long)(r0 + i) * HNB + cc], bcol[i]);
        part[rg * HNB + cc] = acc;
        __syncthreads();
        if (t < HNB) {
            cx<T> s(T(0), T(0));
            for (int g = 0; g < HRG; ++g) s += part[g * HNB + t];
            part[t] = (t < c) ? s : cx<T>(T(0), T(0));       // reuse row 0 of part as w
        }
        __syncthreads();
        // w <- T^H w   (T upper triangular:  (T^H w)[q] = sum_{p<=q} conj(T[p,q]) w[p])
        if (t < HNB) {
            cx<T> s(T(0), T(0));
            if (t < c)
                for (int p = 0; p <= t; ++p) cfma_conj(s, Tm[p * HNB + t], part[p]);
            vec[t] = s;
        }
        __syncthreads();
        // b -= V[R,0:c] w
        for (int i = t; i < nr; i += blockDim.x) {
            cx<T> v = bcol[i];
            for (int q = 0; q < c; ++q) cfma(v, -V[(long)(r0 + i) * HNB + q], vec[q]);
            bcol[i] = v;
        }
        __syncthreads();
    }
    // reflector for x = b[rows j+1 .. n-1]  (LAPACK zlarfg: H = I - tau v v^H, H^H x = beta e1)
    const int x0 = j + 1 - r0;                         // index of alpha inside bcol
    T ss = T(0);
    for (int i = x0 + 1 + t; i < nr; i += blockDim.x) ss += norm2(bcol[i]);
    ss = wave_sum(ss);
    if ((t & 63) == 0) red[t >> 6] = ss;
    __syncthreads();
    T xn2 = T(0);
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) xn2 += red[w];
    const cx<T> alpha = bcol[x0];
    cx<T> tau_c, scale;
    T beta;
    if (xn2 == T(0) && alpha.y == T(0)) {
        tau_c = cx<T>(T(0), T(0));
        scale = cx<T>(T(0), T(0));
        beta = alpha.x;
    } else {
        const T nrm = sqrt(norm2(alpha) + xn2);
        beta = (alpha.x >= T(0)) ? -nrm : nrm;
        tau_c = cx<T>((beta - alpha.x) / beta, -alpha.y / beta);
        scale = crecip(cx<T>(alpha.x - beta, alpha.y));
    }
    __syncthreads();
    if (t == 0) tau[c] = tau_c;
    // write back: rows <= j keep the updated values, A[j+1,j] = beta, below = 0; V[:,c] = [0...0, 1, v]
    for (int i = t; i < nr; i += blockDim.x) {
        const int r = r0 + i;
        cx<T> a, v;
        if (r <= j) { a = bcol[i]; v = cx<T>(T(0), T(0)); }
        else if (r == j + 1) { a = cx<T>(beta, T(0)); v = cx<T>(T(1), T(0)); }
        else { a = cx<T>(T(0), T(0)); v = bcol[i] * scale; }
        A[(long)r * n + j] = a;
        V[(long)r * HNB + c] = v;
    }
    for (int r = t; r < r0; r += blockDim.x) V[(long)r * HNB + c] = cx<T>(T(0), T(0));
}

// Yraw[r, c] = sum_{q>j} A[r, q] * V[q, c]   for r in [r0, n)
// RPW rows per wave and pass: the v element read from LDS serves all of them, and RPW row loads per lane are in flight (2: measured
// 5.07 TB/s in situ at the bench shape; 4 doubles the bytes in flight per CU, which is what an 8 TB/s stream with ~1 us of latency asks for)
template <class T, int RPW>
__global__ __launch_bounds__(512) void hess_gemv_kernel(const cx<T>* __restrict__ Aall, int n, int r0, int j, int c,
                                                         const cx<T>* __restrict__ Vall, cx<T>* __restrict__ Yall, int rows_per_block) {
    TRX_DYN_SMEM(smem);
    cx<T>* v = reinterpret_cast<cx<T>*>(smem);        // [n - j - 1]
    const int b = blockIdx.y;
    const cx<T>* A = Aall + (long)b * n * n;
    const cx<T>* V = Vall + (long)b * n * HNB;
    cx<T>* Y = Yall + (long)b * n * HNB;
    const int len = n - j - 1;
    for (int i = threadIdx.x; i < len; i += blockDim.x) v[i] = V[(long)(j + 1 + i) * HNB + c];
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int rbeg = r0 + blockIdx.x * rows_per_block;
    for (int rr = RPW * wid; rr < rows_per_block; rr += RPW * nw) {
        const int ra = rbeg + rr;
        if (ra >= n) break;
        const cx<T>* row[RPW];
        cx<T> acc[RPW];
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            const bool ok = (rr + q < rows_per_block) && (ra + q < n);
            row[q] = A + (long)(ok ? ra + q : ra) * n + j + 1;       // a missing row re-reads the first one (its sum is not stored)
            acc[q] = cx<T>(T(0), T(0));
        }
        for (int i = lane; i < len; i += 64) {
            const cx<T> vi = v[i];
            cx<T> a[RPW];
#pragma unroll
            for (int q = 0; q < RPW; ++q) a[q] = row[q][i];
#pragma unroll
            for (int q = 0; q < RPW; ++q) cfma(acc[q], a[q], vi);
        }
#pragma unroll
        for (int q = 0; q < RPW; ++q) {
            acc[q].x = wave_sum(acc[q].x); acc[q].y = wave_sum(acc[q].y);
            if (lane == 0 && (rr + q < rows_per_block) && (ra + q < n)) Y[(long)(ra + q) * HNB + c] = acc[q];
        }
    }
}

// YV[r, 0:HNB] = Y[r,:], YV[r, HNB:2HNB] = V[r,:]   for rows r >= r0 (operand of the fused rank-2*HNB update)
template <class T>
__global__ __launch_bounds__(256) void pack_yv_kernel(const cx<T>* __restrict__ Yall, const cx<T>* __restrict__ Vall, cx<T>* __restrict__ YVall, int n, int r0) {
    const int b = blockIdx.y;
    const int r = r0 + blockIdx.x * 4 + (threadIdx.x >> 6), c = threadIdx.x & 63;
    if (r >= n) return;
    const cx<T>* src = (c < HNB) ? Yall : Vall;
    YVall[((long)b * n + r) * 2 * HNB + c] = src[((long)b * n + r) * HNB + (c & (HNB - 1))];
}
// BC[c, j] = conj(V[row0 + j, c])   (first HNB rows of the [2*HNB, n]-strided operand of the fused update)
template <class T>
__global__ __launch_bounds__(256) void conj_transpose_panel_kernel(const cx<T>* __restrict__ Vall, cx<T>* __restrict__ BCall, int n, int row0, int mt) {
    const int b = blockIdx.z, c = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= mt) return;
    BCall[(long)b * 2 * HNB * n + (long)c * n + j] = conj(Vall[((long)b * n + row0 + j) * HNB + c]);
}

template <class T>
__global__ __launch_bounds__(256) void set_identity_batched(cx<T>* __restrict__ X, int n) {
    cx<T>* M = X + (long)blockIdx.z * n * n;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j < n) M[(long)i * n + j] = cx<T>(i == j ? T(1) : T(0), T(0));
}

}  // namespace

static int hess_rpw() {
    static const int v = [] { const char* e = getenv("TRX_HESS_RPW"); const int x = e ? atoi(e) : 0; return x == 4 ? 4 : 2; }();
    return v;
}

template <class T>
int hessenberg(hipStream_t s, const EigBuffers<T>& B, int n, int batch) {
    const cx<T> one(T(1), T(0)), mone(T(-1), T(0)), zero(T(0), T(0));
    const long nn = (long)n * n, sV = (long)n * HNB, sT = HNB * HNB, sW = (long)HNB * n;
    cx<T>*A = B.A, *Z = B.Z, *V = B.Vp, *Y = B.Yp, *Tm = B.Tp, *W = B.W1, *W2 = B.W2;
    TRX_LAUNCH((set_identity_batched<T>), dim3(cdiv_i(n, 256), n, batch), dim3(256), 0, s, Z, n);
    const size_t sm_col = sizeof(cx<T>) * ((size_t)n + HRG * HNB + HNB) + sizeof(T) * 16;
    if (set_max_dyn_smem((const void*)hess_col_kernel<T>, sm_col) || set_max_dyn_smem((const void*)hess_gemv_kernel<T, 2>, sizeof(cx<T>) * (size_t)n) ||
        set_max_dyn_smem((const void*)hess_gemv_kernel<T, 4>, sizeof(cx<T>) * (size_t)n))
        return TRX_ERR_LAUNCH;
    const int rpw = hess_rpw();
    for (int p0 = 0; p0 < n - 2; p0 += HNB) {
        const int ib = (n - 2 - p0 < HNB) ? (n - 2 - p0) : HNB;
        const int r0 = p0 + 1, nr = n - r0;
        if (hipMemsetAsync(Tm, 0, sizeof(cx<T>) * sT * batch, s) != hipSuccess) return TRX_ERR_LAUNCH;
        for (int c = 0; c <= ib; ++c) {
            { ProfScope prof(PROF_HESS_COL, s, 0, 0);
              TRX_LAUNCH((hess_col_kernel<T>), dim3(batch), dim3(HCT), sm_col, s, A, n, p0, ib, c, V, Y, Tm, B.tau); }
            if (c < ib) {
                const int j = p0 + c;
                // 64 rows per 4-wave workgroup fill the chip when the batch supplies the workgroups (batch 128 at n = 1922: 3840); one or
                // two large matrices do not (n = 5202, batch 1: 82 workgroups on 256 CUs, and 83 KB of LDS for v leaves room for one
                // workgroup per CU): there 16 rows per 8-wave workgroup give 4x the workgroups and twice the waves per CU
                const bool few = (long)cdiv_i(nr, 64) * batch < 256;
                const int rpb = few ? 16 : 64, gthreads = few ? 512 : 256;
                // algorithmic traffic of the BLAS-2 stream: the (n-r0) x (n-j-1) trailing block is read once per matrix
                ProfScope prof(PROF_HESS_GEMV, s, 8.0 * (double)nr * (n - j - 1) * batch, (double)sizeof(cx<T>) * nr * (double)(n - j - 1) * batch);
                if (rpw == 4 && !few)
                    TRX_LAUNCH((hess_gemv_kernel<T, 4>), dim3(cdiv_i(nr, rpb), batch), dim3(gthreads), sizeof(cx<T>) * (size_t)(n - j - 1), s,
                               (const cx<T>*)A, n, r0, j, c, (const cx<T>*)V, Y, rpb);
                else
                    TRX_LAUNCH((hess_gemv_kernel<T, 2>), dim3(cdiv_i(nr, rpb), batch), dim3(gthreads), sizeof(cx<T>) * (size_t)(n - j - 1), s,
                               (const cx<T>*)A, n, r0, j, c, (const cx<T>*)V, Y, rpb);
            }
        }
        int rc;
        const int mt = n - p0 - ib;       // trailing columns
        // (1) Ytop = (A[0:r0, r0:n] V[r0:n,:]) T
        rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, r0, ib, nr, one, A + r0, n, nn, V + (long)r0 * HNB, HNB, sV, zero, W, HNB, sW, batch); if (rc) return rc;
        rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, r0, ib, ib, one, W, HNB, sW, Tm, HNB, sT, zero, Y, HNB, sV, batch); if (rc) return rc;
        // (2) A[0:r0, r0:n] -= Ytop V[r0:n,:]^H
        rc = gemm<T>(s, TRX_OP_N, TRX_OP_C, r0, nr, ib, mone, Y, HNB, sV, V + (long)r0 * HNB, HNB, sV, one, A + r0, n, nn, batch); if (rc) return rc;
        if (mt > 0) {
            cx<T>* At = A + (long)r0 * n + p0 + ib;           // A[R, p0+ib:n]
            if (ib == HNB) {
                // Right and left block-reflector updates of the trailing block fused into ONE rank-2*HNB GEMM:
                //   A <- A - Y Vt^H - V T^H W,   W = V^H (A - Y Vt^H) = V^H A - (V^H Y) Vt^H     (Vt = V[p0+ib:n,:])
                //   =>  A <- A - [Y | V] [Vt^H ; T^H W]
                // Three passes over the trailing block (read for V^H A, read+write for the update) instead of five.
                const long sYV = (long)n * 2 * HNB, sBC = (long)2 * HNB * n;
                cx<T>*YV = B.YV, *BC = B.BC, *Sm = B.Sm;
                TRX_LAUNCH((pack_yv_kernel<T>), dim3(cdiv_i(nr, 4), batch), dim3(256), 0, s, (const cx<T>*)Y, (const cx<T>*)V, YV, n, r0);
                TRX_LAUNCH((conj_transpose_panel_kernel<T>), dim3(cdiv_i(mt, 256), HNB, batch), dim3(256), 0, s, (const cx<T>*)V, BC, n, p0 + ib, mt);
                rc = gemm<T>(s, TRX_OP_C, TRX_OP_N, ib, mt, nr, one, V + (long)r0 * HNB, HNB, sV, At, n, nn, zero, W, n, sW, batch); if (rc) return rc;          // W0 = V^H A
                rc = gemm<T>(s, TRX_OP_C, TRX_OP_N, ib, ib, nr, one, V + (long)r0 * HNB, HNB, sV, Y + (long)r0 * HNB, HNB, sV, zero, Sm, HNB, sT, batch); if (rc) return rc;   // S = V^H Y
                rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, ib, mt, ib, mone, Sm, HNB, sT, BC, n, sBC, one, W, n, sW, batch); if (rc) return rc;                        // W = W0 - S Vt^H
                rc = gemm<T>(s, TRX_OP_C, TRX_OP_N, ib, mt, ib, one, Tm, HNB, sT, W, n, sW, zero, BC + (long)HNB * n, n, sBC, batch); if (rc) return rc;          // T^H W
                rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, nr, mt, 2 * HNB, mone, YV + (long)r0 * 2 * HNB, 2 * HNB, sYV, BC, n, sBC, one, At, n, nn, batch); if (rc) return rc;
            } else {
                // (3) A[R, p0+ib:n] -= Y[R,:] V[p0+ib:n,:]^H
                rc = gemm<T>(s, TRX_OP_N, TRX_OP_C, nr, mt, ib, mone, Y + (long)r0 * HNB, HNB, sV, V + (long)(p0 + ib) * HNB, HNB, sV, one, At, n, nn, batch); if (rc) return rc;
                // (4) W = V[R,:]^H A[R, p0+ib:n] ; (5) W2 = T^H W ; (6) A[R,..] -= V[R,:] W2
                rc = gemm<T>(s, TRX_OP_C, TRX_OP_N, ib, mt, nr, one, V + (long)r0 * HNB, HNB, sV, At, n, nn, zero, W, n, sW, batch); if (rc) return rc;
                rc = gemm<T>(s, TRX_OP_C, TRX_OP_N, ib, mt, ib, one, Tm, HNB, sT, W, n, sW, zero, W2, n, sW, batch); if (rc) return rc;
                rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, nr, mt, ib, mone, V + (long)r0 * HNB, HNB, sV, W2, n, sW, one, At, n, nn, batch); if (rc) return rc;
            }
        }
        // (7-9) Z[:, R] -= ((Z[:, R] V[R,:]) T) V[R,:]^H
        rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, ib, nr, one, Z + r0, n, nn, V + (long)r0 * HNB, HNB, sV, zero, W, HNB, sW, batch); if (rc) return rc;
        rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, ib, ib, one, W, HNB, sW, Tm, HNB, sT, zero, W2, HNB, sW, batch); if (rc) return rc;
        rc = gemm<T>(s, TRX_OP_N, TRX_OP_C, n, nr, ib, mone, W2, HNB, sW, V + (long)r0 * HNB, HNB, sV, one, Z + r0, n, nn, batch); if (rc) return rc;
    }
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

template int hessenberg<float>(hipStream_t, const EigBuffers<float>&, int, int);
template int hessenberg<double>(hipStream_t, const EigBuffers<double>&, int, int);

}  // namespace trx
