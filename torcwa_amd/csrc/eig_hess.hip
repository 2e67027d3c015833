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
#include <string>
#include "prof.hpp"

namespace trx {
namespace {

constexpr int HNB = EigPlan::HNB;   // panel width
constexpr int HCT = 512;            // threads of the reflector kernel (one workgroup per matrix; 1024 measured no faster)
constexpr int HRG = HCT / HNB;      // row groups of its HNB x HRG thread grid

// ---- panel column c (matrix column j = p0 + c), fused formulation --------------------------------------------------------------------
// zlahr2 per column:  b = a_j - Y conj(V[j,:]);  b -= V T^H (V^H b);  reflector v of b[j+1:];  y = tau (A v - Y (V^H v));  T[:,c] = -tau T (V^H v).
// The straightforward mapping (one workgroup per matrix doing all of it around the wide A v launch) cost seven passes over the rows and four
// block reductions per column: 100 us per column at n = 1922 and 166 us at n = 5202, all of it single-workgroup latency (a fifth of the
// Hessenberg phase at batch 128, 18 % of a single large matrix's whole solve).  Here everything that is local to a row rides on the wide
// launch, which streams the rows anyway:
//   hess_gemv_kernel(c)  rows r of its block:  Yraw = A[r, j+1:] v;   Y[r,c] = tau_c (Yraw - Y[r,0:c] t_c)            (t_c = V^H v, from the column kernel)
//                        and for the NEXT column:  b[r] = A[r, j+1] - Y[r,0:c+1] conj(V[j+1,0:c+1]) -> Bcol;   partial sums of V^H b -> wpart
//   hess_col_kernel(c)   one workgroup per matrix:  w = T^H (sum of the partials);  b = Bcol - V w  (one pass, b kept in LDS);  one combined
//                        reduction for |b[j+2:]|^2 and u = V[j+2:,:]^H b;  reflector;  write-back pass;  t_c = conj(V[j+1,:]) + scale u;  T[:,c].
template <class T>
__global__ __launch_bounds__(HCT) void hess_col_kernel(cx<T>* __restrict__ Aall, int n, int p0, int c, int nwg,
                                                        cx<T>* __restrict__ Vall, cx<T>* __restrict__ Tall, cx<T>* __restrict__ tau_all,
                                                        cx<T>* __restrict__ tvec_all, const cx<T>* __restrict__ Bcol_all, const cx<T>* __restrict__ wpart_all) {
    TRX_DYN_SMEM(smem);
    cx<T>* bcol = reinterpret_cast<cx<T>*>(smem);     // [n]   current column (rows p0+1..n-1 at index r-(p0+1))
    cx<T>* part = bcol + n;                            // [HRG][HNB] partial sums
    cx<T>* vec = part + HRG * HNB;                     // [HNB]  w, then u
    T* red = reinterpret_cast<T*>(vec + HNB);          // [16] scalar reduction scratch
    const int b = blockIdx.x;
    cx<T>* A = Aall + (long)b * n * n;
    cx<T>* V = Vall + (long)b * n * HNB;
    cx<T>* Tm = Tall + (long)b * HNB * HNB;
    cx<T>* tau = tau_all + (long)b * HNB;
    cx<T>* tvec = tvec_all + (long)b * HNB;
    const cx<T>* Bcol = Bcol_all + (long)b * n;
    const cx<T>* wpart = wpart_all + (long)b * nwg * HNB;
    const int t = threadIdx.x;
    const int cc = t & (HNB - 1), rg = t / HNB;        // HNB x HRG thread grid
    const int r0 = p0 + 1, nr = n - r0, j = p0 + c;
    // 1. w = T^H (V^H b), the partial sums of V^H b come from the wide launch of the previous column
    if (c > 0) {
        cx<T> acc(T(0), T(0));
        if (cc < c)
            for (int g = rg; g < nwg; g += HRG) acc += wpart[(long)g * HNB + cc];
        part[rg * HNB + cc] = acc;
        __syncthreads();
        if (t < HNB) {
            cx<T> sraw(T(0), T(0));
            for (int g = 0; g < HRG; ++g) sraw += part[g * HNB + t];
            vec[t] = (t < c) ? sraw : cx<T>(T(0), T(0));
        }
        __syncthreads();
        cx<T> wq(T(0), T(0));
        if (t < c)
            for (int p = 0; p <= t; ++p) cfma_conj(wq, Tm[p * HNB + t], vec[p]);      // (T^H w)[q] = sum_{p<=q} conj(T[p,q]) w[p]
        __syncthreads();
        if (t < HNB) vec[t] = wq;
        __syncthreads();
    }
    // 2. b = Bcol - V w (first column of a panel: the column of A itself), kept in LDS
    for (int i = t; i < nr; i += blockDim.x) {
        cx<T> v = (c > 0) ? Bcol[r0 + i] : A[(long)(r0 + i) * n + j];
        for (int q = 0; q < c; ++q) cfma(v, -V[(long)(r0 + i) * HNB + q], vec[q]);
        bcol[i] = v;
    }
    __syncthreads();
    // 3. one reduction round:  |b[j+2:]|^2  and  u[q] = sum_{r >= j+2} conj(V[r,q]) b[r]
    const int x0 = j + 1 - r0;                         // index of alpha inside bcol
    T ss = T(0);
    for (int i = x0 + 1 + t; i < nr; i += blockDim.x) ss += norm2(bcol[i]);
    ss = wave_sum(ss);
    if ((t & 63) == 0) red[t >> 6] = ss;
    {
        cx<T> acc(T(0), T(0));
        if (cc < c)
            for (int i = x0 + 1 + rg; i < nr; i += HRG) cfma_conj(acc, V[(long)(r0 + i) * HNB + cc], bcol[i]);
        part[rg * HNB + cc] = acc;
    }
    __syncthreads();
    T xn2 = T(0);
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) xn2 += red[w];
    const cx<T> alpha = bcol[x0];
    cx<T> tau_c, scale;
    T beta;
    // |alpha|^2 + |x|^2 below safmin / eps: the squares have underflowed (or are about to) and beta = sqrt(.) would be 0 or garbage -- tau = 0 / 0
    // -> NaN in every later column.  It happens for real: inside a cluster of m nearly equal eigenvalues the subdiagonals of the reduction decay
    // like (gap)^k (36 eigenvalues 3e-9 apart: 1e-153 after 18 columns; found by the round-6 fallback test).  LAPACK's zlarfg rescales there;
    // such a column is zero to 1e-146 of anything representable next to it, so it is treated AS zero: H = I, beta = 0, nothing below.
    const bool vanishing = !(norm2(alpha) + xn2 >= eps_of<T>::safmin / eps_of<T>::value);
    if (vanishing || (xn2 == T(0) && alpha.y == T(0))) {
        tau_c = cx<T>(T(0), T(0));
        scale = cx<T>(T(0), T(0));
        beta = vanishing ? T(0) : alpha.x;
    } else {
        const T nrm = sqrt(norm2(alpha) + xn2);
        beta = (alpha.x >= T(0)) ? -nrm : nrm;
        tau_c = cx<T>((beta - alpha.x) / beta, -alpha.y / beta);
        scale = crecip(cx<T>(alpha.x - beta, alpha.y));
    }
    // t_c = V^H v = conj(V[j+1,:]) + scale u   (v = [0 .. 0, 1, scale b[j+2:]]);   T[0:c,c] = -tau T t_c,  T[c,c] = tau
    if (t < HNB) {
        cx<T> u(T(0), T(0));
        for (int g = 0; g < HRG; ++g) u += part[g * HNB + t];
        cx<T> tq(T(0), T(0));
        if (t < c) tq = conj(V[(long)(j + 1) * HNB + t]) + scale * u;
        vec[t] = tq;
        tvec[t] = tq;
    }
    if (t == 0) tau[c] = tau_c;
    __syncthreads();
    if (t < HNB) {
        cx<T> sT(T(0), T(0));
        if (t < c) {
            for (int q = t; q < c; ++q) cfma(sT, Tm[t * HNB + q], vec[q]);
            sT = -(tau_c * sT);
        } else if (t == c) {
            sT = tau_c;
        }
        if (t <= c) Tm[t * HNB + c] = sT;
    }
    // 4. write back: rows <= j keep the updated values, A[j+1,j] = beta, below = 0; V[:,c] = [0...0, 1, v]
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

// Wide launch of column c.  RPW rows per wave and pass: the v element read from LDS serves all of them, and RPW row loads per lane are in
// flight (2: 5.07 TB/s in situ at the bench shape; 4: +0.7 % / +2.7 % of the whole step at batch 128 / 16).  The row-local tail (Y final,
// next column's b, partial V^H b) handles two rows at a time on the two halves of the wave (lane & 31 = panel column q).
template <class T, int RPW>
__global__ __launch_bounds__(512) void hess_gemv_kernel(const cx<T>* __restrict__ Aall, int n, int r0, int j, int c, int next,
                                                         const cx<T>* __restrict__ Vall, cx<T>* __restrict__ Yall, const cx<T>* __restrict__ tau_all,
                                                         const cx<T>* __restrict__ tvec_all, cx<T>* __restrict__ Bcol_all, cx<T>* __restrict__ wpart_all,
                                                         int rows_per_block) {
    TRX_DYN_SMEM(smem);
    static_assert(RPW % 2 == 0 && HNB == 32, "two rows per half-wave round");
    cx<T>* v = reinterpret_cast<cx<T>*>(smem);        // [n - j - 1]
    cx<T>* wred = v + (n - j - 1) + 2;                 // [waves][HNB] partial V^H b of each wave (v may sit one element in: see `off`)
    const int b = blockIdx.y;
    const cx<T>* A = Aall + (long)b * n * n;
    const cx<T>* V = Vall + (long)b * n * HNB;
    cx<T>* Y = Yall + (long)b * n * HNB;
    const int len = n - j - 1;
    // fp32, even n: the row stream is read as PAIRS of elements (16 bytes per lane, like the fp64 stream) -- every row of A starts at an even
    // element index when n is even, so the pairs i = off, off + 2, ... with off = (j + 1) & 1 are 16-byte aligned in every row; v sits in LDS
    // shifted by off so that its pairs are aligned too.  (8-byte loads kept the fp32 stream at 4.3 TB/s in situ where the fp64 one reaches 5.2.)
    const bool pairs = sizeof(T) == 4 && (n & 1) == 0;
    const int off = pairs ? ((j + 1) & 1) : 0;
    for (int i = threadIdx.x; i < len; i += blockDim.x) v[i + off] = V[(long)(j + 1 + i) * HNB + c];
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int q = lane & 31, half = lane >> 5;
    const int rbeg = r0 + blockIdx.x * rows_per_block;
    const cx<T> tau_c = tau_all[(long)b * HNB + c];
    const cx<T> tq = (q < c) ? tvec_all[(long)b * HNB + q] : cx<T>(T(0), T(0));
    const cx<T> vjn = (next && q <= c) ? conj(V[(long)(j + 1) * HNB + q]) : cx<T>(T(0), T(0));       // conj(V[j+1, q]); V[j+1, c] = 1
    cx<T> wacc(T(0), T(0));                            // this lane's share of sum_r conj(V[r,q]) b[r]
    for (int rr = RPW * wid; rr < rows_per_block; rr += RPW * nw) {
        const int ra = rbeg + rr;
        if (ra >= n) break;
        const cx<T>* row[RPW];
        cx<T> acc[RPW];
#pragma unroll
        for (int k = 0; k < RPW; ++k) {
            const bool ok = (rr + k < rows_per_block) && (ra + k < n);
            row[k] = A + (long)(ok ? ra + k : ra) * n + j + 1;       // a missing row re-reads the first one (its sum is not stored)
            acc[k] = cx<T>(T(0), T(0));
        }
        // inputs of the row-local tail, requested BEFORE the row stream so that they have landed when it ends (two rows per round, one per
        // half-wave; lane & 31 = panel column q)
        cx<T> yq_[RPW / 2], vq_[RPW / 2], an_[RPW / 2];
#pragma unroll
        for (int kp = 0; kp < RPW / 2; ++kp) {
            const int k = 2 * kp + half, r = ra + k;
            const bool ok = (rr + k < rows_per_block) && (r < n);
            yq_[kp] = (ok && q < c) ? Y[(long)r * HNB + q] : cx<T>(T(0), T(0));
            vq_[kp] = (ok && next && q <= c) ? V[(long)r * HNB + q] : cx<T>(T(0), T(0));
            an_[kp] = (ok && next) ? A[(long)r * n + j + 1] : cx<T>(T(0), T(0));
        }
        // fp32: two 64-wide column chunks per round -- with 8-byte elements one chunk keeps only 1 KB per wave in flight (hipcc waits for every
        // load right behind it: ISA of round 3), half of what the fp64 stream has, and the fp32 gemv ran at 4.25 TB/s against 5.1 TB/s
        constexpr int UNR = sizeof(T) == 4 ? 2 : 1;
        int i = lane;
        if (pairs) {
            if constexpr (sizeof(T) == 4) {
                typedef float pair_t __attribute__((ext_vector_type(4)));          // two complex elements
                const int npair = (len - off) >> 1;
                const pair_t* vp = reinterpret_cast<const pair_t*>(v + 2 * off);   // pair p = elements off + 2 p, off + 2 p + 1; LDS slot 2 p + 2 off
                int p = lane;
                for (; p + 64 < npair; p += 128) {                                 // two 128-column chunks per round: 2 x RPW 16-byte loads in flight
                    pair_t a0[RPW], a1[RPW];
#pragma unroll
                    for (int k = 0; k < RPW; ++k) {
                        const pair_t* rp = reinterpret_cast<const pair_t*>(row[k] + off);
                        a0[k] = rp[p]; a1[k] = rp[p + 64];
                    }
                    const pair_t v0 = vp[p], v1 = vp[p + 64];
#pragma unroll
                    for (int k = 0; k < RPW; ++k) {
                        cfma(acc[k], cx<T>(a0[k].x, a0[k].y), cx<T>(v0.x, v0.y)); cfma(acc[k], cx<T>(a0[k].z, a0[k].w), cx<T>(v0.z, v0.w));
                        cfma(acc[k], cx<T>(a1[k].x, a1[k].y), cx<T>(v1.x, v1.y)); cfma(acc[k], cx<T>(a1[k].z, a1[k].w), cx<T>(v1.z, v1.w));
                    }
                }
                for (; p < npair; p += 64) {
                    const pair_t v0 = vp[p];
#pragma unroll
                    for (int k = 0; k < RPW; ++k) {
                        const pair_t a0 = reinterpret_cast<const pair_t*>(row[k] + off)[p];
                        cfma(acc[k], cx<T>(a0.x, a0.y), cx<T>(v0.x, v0.y)); cfma(acc[k], cx<T>(a0.z, a0.w), cx<T>(v0.z, v0.w));
                    }
                }
                // the element in front of the first pair and the one behind the last
                if (off && lane == 0) {
#pragma unroll
                    for (int k = 0; k < RPW; ++k) cfma(acc[k], row[k][0], v[off]);
                }
                if (((len - off) & 1) && lane == 1) {
#pragma unroll
                    for (int k = 0; k < RPW; ++k) cfma(acc[k], row[k][len - 1], v[len - 1 + off]);
                }
            }
            i = len;                                                               // nothing left for the element loops below
        }
        for (; i + 64 * (UNR - 1) < len; i += 64 * UNR) {
            cx<T> a[UNR][RPW], vi[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u)
#pragma unroll
                for (int k = 0; k < RPW; ++k) a[u][k] = row[k][i + 64 * u];
#pragma unroll
            for (int u = 0; u < UNR; ++u) vi[u] = v[i + 64 * u + off];
#pragma unroll
            for (int u = 0; u < UNR; ++u)
#pragma unroll
                for (int k = 0; k < RPW; ++k) cfma(acc[k], a[u][k], vi[u]);
        }
        for (; i < len; i += 64) {
            const cx<T> vi = v[i + off];
#pragma unroll
            for (int k = 0; k < RPW; ++k) cfma(acc[k], row[k][i], vi);
        }
#pragma unroll
        for (int k = 0; k < RPW; ++k) { acc[k].x = wave_sum(acc[k].x); acc[k].y = wave_sum(acc[k].y); }
#pragma unroll
        for (int kp = 0; kp < RPW / 2; ++kp) {
            const int k = 2 * kp + half;
            const int r = ra + k;
            const bool ok = (rr + k < rows_per_block) && (r < n);
            const cx<T> yraw = half ? acc[2 * kp + 1] : acc[2 * kp];
            const cx<T> yq = yq_[kp];
            cx<T> corr = yq * tq;
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) { corr.x += __shfl_xor(corr.x, o); corr.y += __shfl_xor(corr.y, o); }       // within the half-wave
            const cx<T> yfin = tau_c * (yraw - corr);
            if (ok && q == 0) Y[(long)r * HNB + c] = yfin;
            if (next) {
                cx<T> term = (q < c ? yq : (q == c ? yfin : cx<T>(T(0), T(0)))) * vjn;
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) { term.x += __shfl_xor(term.x, o); term.y += __shfl_xor(term.y, o); }
                const cx<T> bnew = ok ? an_[kp] - term : cx<T>(T(0), T(0));
                if (ok && q == 0) Bcol_all[(long)b * n + r] = bnew;
                if (ok && q <= c) cfma_conj(wacc, vq_[kp], bnew);
            }
        }
    }
    if (next) {
        // the two halves of a wave, then the waves of the workgroup
        wacc.x += __shfl_xor(wacc.x, 32); wacc.y += __shfl_xor(wacc.y, 32);
        if (lane < HNB) wred[wid * HNB + lane] = wacc;
        __syncthreads();
        if (threadIdx.x < HNB) {
            cx<T> sW(T(0), T(0));
            for (int w = 0; w < nw; ++w) sW += wred[w * HNB + threadIdx.x];
            wpart_all[((long)b * gridDim.x + blockIdx.x) * HNB + threadIdx.x] = sW;
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


// Delayed right updates (round 6): the reflector blocks of HG consecutive panels are kept side by side, Vg = [V_0 | V_1 | ...] (explicit zeros
// above each panel's first row and in unused columns), with the diagonal blocks of the merged triangular factor Tg.
template <class T>
__global__ __launch_bounds__(256) void pack_group_kernel(const cx<T>* __restrict__ Vall, const cx<T>* __restrict__ Tall, cx<T>* __restrict__ Vg_all,
                                                         cx<T>* __restrict__ Tg_all, int n, int g_r0, int r0, int ib, int gp) {
    constexpr int GK = EigPlan::HGK;
    const int b = blockIdx.y;
    const int r = g_r0 + blockIdx.x * 8 + (threadIdx.x >> 5), c = threadIdx.x & 31;
    if (r < n) Vg_all[((long)b * n + r) * GK + gp * HNB + c] = (r >= r0 && c < ib) ? Vall[((long)b * n + r) * HNB + c] : cx<T>(T(0), T(0));
    if (blockIdx.x == 0) {
        cx<T>* Tg = Tg_all + (long)b * GK * GK;
        const cx<T>* Tm = Tall + (long)b * HNB * HNB;
        for (int i = threadIdx.x >> 5; i < HNB; i += 8) {
            // row block gp of Tg: zero left of the diagonal block (and right of it: filled by the merge), the panel's T on the diagonal
            for (int q = 0; q < EigPlan::HG; ++q)
                Tg[(long)(gp * HNB + i) * GK + q * HNB + c] = (q == gp && i < ib && c < ib) ? Tm[i * HNB + c] : cx<T>(T(0), T(0));
        }
    }
}

template <class T>
__global__ __launch_bounds__(256) void set_identity_batched(cx<T>* __restrict__ X, int n) {
    cx<T>* M = X + (long)blockIdx.z * n * n;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j < n) M[(long)i * n + j] = cx<T>(i == j ? T(1) : T(0), T(0));
}

}  // namespace

// rows per wave and pass of the wide launch (TRX_HESS_RPW = 2 / 4 forces one).  Measured with the fused row-local tail (whole step, round 3):
// batch 128: 2 rows 29.47-29.73, 4 rows 29.22-29.45 layer-solves/s;  batch 16: 2 rows 13.37, 4 rows 13.64
// Round 6, fp32 stream read as 16-byte pairs: 2 against 4 rows at batch 128 measured twice, 36.60 / 36.66 / 36.88 against 36.76 / 36.61
// layer-solves/s -- no difference, the rule stands (profiles/r06_ab/r6u..., r6v...).
static int hess_rpw(int batch) {
    static const int v = [] { const char* e = getenv("TRX_HESS_RPW"); const int x = e ? atoi(e) : 0; return (x == 2 || x == 4) ? x : 0; }();
    return v ? v : (batch >= 64 ? 2 : 4);
}

// Panels per group of the delayed right updates (TRX_HESS_GROUP / trx_tuning("hess_group", g)): 0 automatic (4), 1 = every panel on its own as in
// rounds 1 - 5.
static int hess_group_env() { const char* e = getenv("TRX_HESS_GROUP"); const int v = e ? atoi(e) : 0; return (v >= 0 && v <= EigPlan::HG) ? v : 0; }
static int g_hess_group = hess_group_env();
int hess_set_knob(const char* key, int value) {
    if (std::string(key) != "hess_group" || value < 0 || value > EigPlan::HG) return TRX_ERR_ARG;
    g_hess_group = value;
    return TRX_OK;
}

// The right updates of a panel's block reflector touch two regions that never feed back into the reduction: the rows of A above the panel and
// all of Z.  Per panel they are rank-32 updates that read and write the region once each (plus one read for X V): bound by HBM, and half of
// the traffic of the phase's block updates.  They are therefore DELAYED: the reflectors of up to HG = 4 consecutive panels are merged,
//     (I - V_a T_a V_a^H)(I - V_b T_b V_b^H) = I - [V_a V_b] [[T_a, -T_a (V_a^H V_b) T_b], [0, T_b]] [V_a V_b]^H,
// and applied at once as a rank-128 update: a quarter of the passes, and products the matrix cores bound.  Rows of A between the group's
// first row g_r0 and a later panel's first row are "above" that panel only: those (at most 96) rows keep the per-panel update.
template <class T>
static int hess_flush_group(hipStream_t s, const EigBuffers<T>& B, int n, int batch, int g_r0, int np) {
    constexpr int GK = EigPlan::HGK;
    const cx<T> one(T(1), T(0)), mone(T(-1), T(0)), zero(T(0), T(0));
    const long nn = (long)n * n, sVg = (long)n * GK, sTg = (long)GK * GK;
    cx<T>*Vg = B.Vg, *Tg = B.Tg, *Gg = B.Gg, *Wg = B.Wg, *W2g = B.W2g, *A = B.A, *Z = B.Z;
    const int kk = np * HNB, nr = n - g_r0;
    int rc;
    if (np > 1) {
        // G = Vg^H Vg (only the blocks above the diagonal are used), then the off-diagonal blocks of the merged T, pairs first
        rc = gemm<T>(s, TRX_OP_C, TRX_OP_N, kk, kk, nr, one, Vg + (long)g_r0 * GK, GK, sVg, Vg + (long)g_r0 * GK, GK, sVg, zero, Gg, GK, sTg, batch); if (rc) return rc;
        auto merge = [&](int a0, int ka, int b0, int kb) {      // Tg[a, b] = -Tg[a, a] G[a, b] Tg[b, b]   (W2g: scratch)
            int r = gemm<T>(s, TRX_OP_N, TRX_OP_N, ka, kb, kb, one, Gg + (long)a0 * GK + b0, GK, sTg, Tg + (long)b0 * GK + b0, GK, sTg, zero, W2g, GK, sVg, batch);
            if (r) return r;
            return gemm<T>(s, TRX_OP_N, TRX_OP_N, ka, kb, ka, mone, Tg + (long)a0 * GK + a0, GK, sTg, W2g, GK, sVg, zero, Tg + (long)a0 * GK + b0, GK, sTg, batch);
        };
        rc = merge(0, HNB, HNB, HNB); if (rc) return rc;
        if (np == 3) { rc = merge(0, 2 * HNB, 2 * HNB, HNB); if (rc) return rc; }
        if (np == 4) {
            rc = merge(2 * HNB, HNB, 3 * HNB, HNB); if (rc) return rc;
            rc = merge(0, 2 * HNB, 2 * HNB, 2 * HNB); if (rc) return rc;
        }
    }
    // rows of A above the group: A[0:g_r0, g_r0:n] <- A[0:g_r0, g_r0:n] (I - Vg Tg Vg^H)
    if (g_r0 > 0) {
        rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, g_r0, kk, nr, one, A + g_r0, n, nn, Vg + (long)g_r0 * GK, GK, sVg, zero, Wg, GK, sVg, batch); if (rc) return rc;
        rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, g_r0, kk, kk, one, Wg, GK, sVg, Tg, GK, sTg, zero, W2g, GK, sVg, batch); if (rc) return rc;
        rc = gemm<T>(s, TRX_OP_N, TRX_OP_C, g_r0, nr, kk, mone, W2g, GK, sVg, Vg + (long)g_r0 * GK, GK, sVg, one, A + g_r0, n, nn, batch); if (rc) return rc;
    }
    // Z[:, g_r0:n] <- Z[:, g_r0:n] (I - Vg Tg Vg^H)
    rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, kk, nr, one, Z + g_r0, n, nn, Vg + (long)g_r0 * GK, GK, sVg, zero, Wg, GK, sVg, batch); if (rc) return rc;
    rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, kk, kk, one, Wg, GK, sVg, Tg, GK, sTg, zero, W2g, GK, sVg, batch); if (rc) return rc;
    rc = gemm<T>(s, TRX_OP_N, TRX_OP_C, n, nr, kk, mone, W2g, GK, sVg, Vg + (long)g_r0 * GK, GK, sVg, one, Z + g_r0, n, nn, batch); if (rc) return rc;
    return TRX_OK;
}

// (Paired launches -- the batch in two halves half a column step apart, ONE launch streaming for one half and running the reflector step of
// the other, no second stream -- were built and measured in round 6: the reflector kernel does hide (its tag disappears), but a launch that
// streams half the batch takes 170 us where the full batch takes 295, and the phase stays at 881 - 887 ms (batch 32: 279 -> 296 ms).
// Removed; profiles/r06_ab/r6t_hessenberg_paired_launches.txt.)
// (Two to four sub-batches on side streams, half a panel out of phase -- one's block updates and reflector kernels under the other's gemv
// stream -- were measured in round 6 and removed: the phase takes 882 ms as one batch of 128 and 914 / 946 / 1001 ms as 2 / 3 / 4 sub-batches;
// two chip-filling gemv grids share the chip instead of overlapping.  profiles/r06_ab/cumask.txt)
template <class T>
int hessenberg(hipStream_t s, const EigBuffers<T>& B, int n, int batch) {
    const cx<T> one(T(1), T(0)), mone(T(-1), T(0)), zero(T(0), T(0));
    const long nn = (long)n * n, sV = (long)n * HNB, sT = HNB * HNB, sW = (long)HNB * n;
    cx<T>*A = B.A, *Z = B.Z, *V = B.Vp, *Y = B.Yp, *Tm = B.Tp, *W = B.W1, *W2 = B.W2;
    TRX_LAUNCH((set_identity_batched<T>), dim3(cdiv_i(n, 256), n, batch), dim3(256), 0, s, Z, n);
    const size_t sm_col = sizeof(cx<T>) * ((size_t)n + HRG * HNB + HNB) + sizeof(T) * 16;
    const size_t sm_gemv = sizeof(cx<T>) * ((size_t)n + 2 + 8 * HNB);
    if (set_max_dyn_smem((const void*)hess_col_kernel<T>, sm_col) || set_max_dyn_smem((const void*)hess_gemv_kernel<T, 2>, sm_gemv) ||
        set_max_dyn_smem((const void*)hess_gemv_kernel<T, 4>, sm_gemv))
        return TRX_ERR_LAUNCH;
    const int rpw = hess_rpw(batch);
    cx<T>* Bcol = W;            // [B, n]        next column with the pending right update applied (the GEMM scratch is free during the column loop)
    cx<T>* wpart = W2;          // [B, nwg, HNB] partial sums of V^H b, one row per workgroup of the wide launch
    const int hg = g_hess_group ? g_hess_group : EigPlan::HG;        // panels per group of the delayed right updates (1: none delayed)
    int g_r0 = 1, gp = 0;                                            // first row of the current group, panels it holds
    for (int p0 = 0; p0 < n - 2; p0 += HNB) {
        const int ib = (n - 2 - p0 < HNB) ? (n - 2 - p0) : HNB;
        const int r0 = p0 + 1, nr = n - r0;
        if (hipMemsetAsync(Tm, 0, sizeof(cx<T>) * sT * batch, s) != hipSuccess) return TRX_ERR_LAUNCH;
        // 64 rows per 4-wave workgroup fill the chip when the batch supplies the workgroups (batch 128 at n = 1922: 3840); one or
        // two large matrices do not (n = 5202, batch 1: 82 workgroups on 256 CUs, and 83 KB of LDS for v leaves room for one
        // workgroup per CU): there 16 rows per 8-wave workgroup give 4x the workgroups and twice the waves per CU
        const bool few = (long)cdiv_i(nr, 64) * batch < 256;
        const int rpb = few ? 16 : 64, gthreads = few ? 512 : 256;
        const int nwg = cdiv_i(nr, rpb);
        for (int c = 0; c < ib; ++c) {
            const int j = p0 + c;
            { ProfScope prof(PROF_HESS_COL, s, 0, 0);
              TRX_LAUNCH((hess_col_kernel<T>), dim3(batch), dim3(HCT), sm_col, s, A, n, p0, c, nwg, V, Tm, B.tau, B.tvec, (const cx<T>*)Bcol, (const cx<T>*)wpart); }
            const int next = (c + 1 < ib) ? 1 : 0;
            const size_t smg = sizeof(cx<T>) * ((size_t)(n - j - 1) + 2 + (gthreads / 64) * HNB);
            // algorithmic traffic of the BLAS-2 stream: the (n-r0) x (n-j-1) trailing block is read once per matrix
            ProfScope prof(PROF_HESS_GEMV, s, 8.0 * (double)nr * (n - j - 1) * batch, (double)sizeof(cx<T>) * nr * (double)(n - j - 1) * batch);
            if (rpw == 4 && !few)
                TRX_LAUNCH((hess_gemv_kernel<T, 4>), dim3(nwg, batch), dim3(gthreads), smg, s, (const cx<T>*)A, n, r0, j, c, next, (const cx<T>*)V, Y,
                           (const cx<T>*)B.tau, (const cx<T>*)B.tvec, Bcol, wpart, rpb);
            else
                TRX_LAUNCH((hess_gemv_kernel<T, 2>), dim3(nwg, batch), dim3(gthreads), smg, s, (const cx<T>*)A, n, r0, j, c, next, (const cx<T>*)V, Y,
                           (const cx<T>*)B.tau, (const cx<T>*)B.tvec, Bcol, wpart, rpb);
        }
        int rc;
        const int mt = n - p0 - ib;       // trailing columns
        if (gp == 0) g_r0 = r0;
        const int t0 = hg > 1 ? g_r0 : 0, tr = r0 - t0;      // rows above the panel that are updated now: all of them (hg == 1) or those inside the group
        if (tr > 0) {
            // (1) Ytop = (A[t0:r0, r0:n] V[r0:n,:]) T
            rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, tr, ib, nr, one, A + (long)t0 * n + r0, n, nn, V + (long)r0 * HNB, HNB, sV, zero, W, HNB, sW, batch); if (rc) return rc;
            rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, tr, ib, ib, one, W, HNB, sW, Tm, HNB, sT, zero, Y + (long)t0 * HNB, HNB, sV, batch); if (rc) return rc;
            // (2) A[t0:r0, r0:n] -= Ytop V[r0:n,:]^H
            rc = gemm<T>(s, TRX_OP_N, TRX_OP_C, tr, nr, ib, mone, Y + (long)t0 * HNB, HNB, sV, V + (long)r0 * HNB, HNB, sV, one, A + (long)t0 * n + r0, n, nn, batch); if (rc) return rc;
        }
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
        if (hg == 1) {
            // (7-9) Z[:, R] -= ((Z[:, R] V[R,:]) T) V[R,:]^H
            rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, ib, nr, one, Z + r0, n, nn, V + (long)r0 * HNB, HNB, sV, zero, W, HNB, sW, batch); if (rc) return rc;
            rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, ib, ib, one, W, HNB, sW, Tm, HNB, sT, zero, W2, HNB, sW, batch); if (rc) return rc;
            rc = gemm<T>(s, TRX_OP_N, TRX_OP_C, n, nr, ib, mone, W2, HNB, sW, V + (long)r0 * HNB, HNB, sV, one, Z + r0, n, nn, batch); if (rc) return rc;
        } else {
            // the panel joins its group; the rows of A above the group and Z get the merged reflector when the group is full (or the reduction ends)
            TRX_LAUNCH((pack_group_kernel<T>), dim3(cdiv_i(n - g_r0, 8), batch), dim3(256), 0, s, (const cx<T>*)V, (const cx<T>*)Tm, B.Vg, B.Tg, n, g_r0, r0, ib, gp);
            ++gp;
            if (gp == hg || p0 + HNB >= n - 2) {
                rc = hess_flush_group<T>(s, B, n, batch, g_r0, gp); if (rc) return rc;
                gp = 0;
            }
        }
    }
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

template int hessenberg<float>(hipStream_t, const EigBuffers<float>&, int, int);
template int hessenberg<double>(hipStream_t, const EigBuffers<double>&, int, int);

}  // namespace trx
