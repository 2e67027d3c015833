// Batched partial-pivot LU factorisation and multi-RHS solves (row-major, interleaved complex).
// Replaces the torch.linalg.inv call sites of the reference hot path (torcwa/rcwa.py:1157, 1174, 1226, 1230,
// 1248, 1266-1267, 1271, 1273, 1287-1288): the MI355X build never forms an explicit inverse unless the
// algorithm needs one, it factors once and solves.
//
// Blocked right-looking algorithm, panel width NB:
//   lu_panel_kernel      one workgroup per matrix factors the (n-k0) x jb panel (pivot search = block reduction)
//   lu_swap_range_kernel applies row interchanges to two column windows (inside the outer block per panel, outside it per block)
//   trsm_lower_kernel    U12 = L11^-1 A12      (one thread per column, L11 broadcast from LDS)
//   gemm                 A22 -= L21 U12
#include "common.hpp"
#include <cstdlib>
#include "prof.hpp"
#include <mutex>
#include <string>

namespace trx {
namespace {

constexpr int NB = 32;

template <class T>
__global__ __launch_bounds__(512) void lu_panel_kernel(cx<T>* __restrict__ Aall, int lda, long sA, int n, int k0, int jb,
                                                        int* __restrict__ piv_all, int* __restrict__ info_all) {
    __shared__ cx<T> prow[NB];
    __shared__ cx<T> rinv[NB];
    __shared__ T red_v[8];
    __shared__ int red_i[8];
    __shared__ int s_piv;
    const int b = blockIdx.x;
    cx<T>* A = Aall + (long)b * sA;
    int* piv = piv_all + (long)b * n;
    const int t = threadIdx.x, nt = blockDim.x;
    const int lane = t & 63, wid = t >> 6, nw = nt >> 6;

    for (int j = 0; j < jb; ++j) {
        const int col = k0 + j;
        // 1. pivot search (LAPACK i?amax convention: max |re|+|im|, first occurrence)
        T best = T(-1);
        int bi = col;
        for (int i = col + t; i < n; i += nt) {
            T v = abs1(A[(long)i * lda + col]);
            if (v > best) { best = v; bi = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            T ov = __shfl_xor(best, o);
            int oi = __shfl_xor(bi, o);
            if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
        }
        if (lane == 0) { red_v[wid] = best; red_i[wid] = bi; }
        __syncthreads();
        if (t == 0) {
            T bv = red_v[0];
            int bidx = red_i[0];
            for (int w = 1; w < nw; ++w)
                if (red_v[w] > bv || (red_v[w] == bv && red_i[w] < bidx)) { bv = red_v[w]; bidx = red_i[w]; }
            s_piv = bidx;
            piv[col] = bidx;
            if (!(bv > T(0)) && info_all[b] == 0) info_all[b] = col + 1;
        }
        __syncthreads();
        const int p = s_piv;
        // 2. swap rows col <-> p inside the panel, and publish the pivot row
        if (t < jb) {
            cx<T> a = A[(long)col * lda + k0 + t];
            if (p != col) {
                cx<T> c = A[(long)p * lda + k0 + t];
                A[(long)p * lda + k0 + t] = a;
                A[(long)col * lda + k0 + t] = c;
                a = c;
            }
            prow[t] = a;
            if (t == j) rinv[j] = (a.x != T(0) || a.y != T(0)) ? crecip(a) : cx<T>(T(0), T(0));
        }
        __syncthreads();
        // 3. rank-1 update of the remaining panel columns; the multipliers stay unscaled until the end
        const int ncol = jb - j - 1;
        if (ncol > 0) {
            const cx<T> ri = rinv[j];
            const int cpart = t & 15, rpart = t >> 4, rstep = nt >> 4;
            for (int i = col + 1 + rpart; i < n; i += rstep) {
                cx<T>* row = A + (long)i * lda + k0;
                const cx<T> l = row[j] * ri;
                for (int c = j + 1 + cpart; c < jb; c += 16) row[c] -= l * prow[c];
            }
        }
        __syncthreads();
    }
    // 4. scale the multipliers: L[i, j] = A[i, j] / pivot_j for i > k0 + j
    {
        const int cpart = t & 31, rpart = t >> 5, rstep = nt >> 5;
        if (cpart < jb) {
            const cx<T> ri = rinv[cpart];
            for (int i = k0 + rpart; i < n; i += rstep)
                if (i > k0 + cpart) {
                    cx<T>* p = A + (long)i * lda + k0 + cpart;
                    *p = (*p) * ri;
                }
        }
    }
}

// ---- one-workgroup panel with LDS-resident sub-blocks ----------------------------------------------------------------------------
// lu_panel_kernel walks the panel once per column from L2: a pivot search pass and a rank-1 pass, each a chain of L2 round trips of ONE
// workgroup (14 us per column, 440 us per 32-column panel at n = 1922 -- 140 ms of a 128-point step over its LU factorisations).  Here the
// panel is cut into sub-blocks of PSB columns; the (rows x PSB) sub-block is loaded into LDS ONCE, its PSB columns are factored there (pivot
// search, row swaps, rank-1 updates: LDS round trips and barriers only; the swaps reach the other columns of the panel in global memory), the
// multipliers are scaled in LDS, the sub-block goes back, and the rest of the panel receives the sub-block's contribution in ONE rank-PSB pass
// with the multipliers still in LDS.  Same pivots, same arithmetic up to the order of the updates.  Used while rows x PSB fits the LDS budget.
// Sub-blocks of 8 columns while rows x 9 elements fit (fp64: 1094 rows, fp32: 2189), of 4 columns up to twice that (fp64: 1971 rows -- the
// whole n = 1922 factorisation of the bench shape -- fp32: 3942); taller panels go to the row-split kernels below.
constexpr int PLT = 512;
template <class T, int PSB>
__global__ __launch_bounds__(PLT) void lu_panel_lds_kernel(cx<T>* __restrict__ Aall, int lda, long sA, int n, int k0, int jb,
                                                            int* __restrict__ piv_all, int* __restrict__ info_all) {
    TRX_DYN_SMEM(smem);
    constexpr int SLD = PSB + 1;                                  // row stride of the LDS sub-block (elements)
    cx<T>* S = reinterpret_cast<cx<T>*>(smem);                    // [rows][SLD]
    __shared__ cx<T> Ub[PSB][NB];                                 // U rows of the sub-block in the rest columns
    __shared__ cx<T> rinv[NB];
    __shared__ T red_v[PLT / 64];
    __shared__ int red_i[PLT / 64];
    __shared__ int s_piv;
    const int b = blockIdx.x;
    cx<T>* A = Aall + (long)b * sA;
    int* piv = piv_all + (long)b * n;
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
    for (int s0 = 0; s0 < jb; s0 += PSB) {
        const int s1 = s0 + PSB < jb ? s0 + PSB : jb, ks = s1 - s0;
        const int rtop = k0 + s0, m = n - rtop;                   // rows of this sub-block: rtop .. n - 1
        // 1. sub-block -> LDS
        for (int e = t; e < m * PSB; e += PLT) {
            const int i = e / PSB, c = e - i * PSB;
            if (c < ks) S[i * SLD + c] = A[(long)(rtop + i) * lda + k0 + s0 + c];
        }
        __syncthreads();
        for (int j = s0; j < s1; ++j) {
            const int jl = j - s0, col = k0 + j, il = j - s0;     // column inside the sub-block; its diagonal row inside S is il
            // 2. pivot search over S[il.., jl] (LAPACK i?amax convention: max |re| + |im|, first occurrence)
            T best = T(-1);
            int bi = col;
            for (int i = il + t; i < m; i += PLT) {
                const T v = abs1(S[i * SLD + jl]);
                if (v > best) { best = v; bi = rtop + i; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const T ov = __shfl_xor(best, o);
                const int oi = __shfl_xor(bi, o);
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            if (lane == 0) { red_v[wid] = best; red_i[wid] = bi; }
            __syncthreads();
            if (t == 0) {
                T bv = red_v[0];
                int bidx = red_i[0];
                for (int w = 1; w < PLT / 64; ++w)
                    if (red_v[w] > bv || (red_v[w] == bv && red_i[w] < bidx)) { bv = red_v[w]; bidx = red_i[w]; }
                s_piv = bidx;
                piv[col] = bidx;
                if (!(bv > T(0)) && info_all[b] == 0) info_all[b] = col + 1;
            }
            __syncthreads();
            const int p = s_piv;
            // 3. swap rows col <-> p: inside the sub-block in LDS, in the other columns of the panel in global memory
            if (p != col) {
                if (t < ks) {
                    const cx<T> a = S[il * SLD + t], c = S[(p - rtop) * SLD + t];
                    S[il * SLD + t] = c; S[(p - rtop) * SLD + t] = a;
                } else if (t >= 64 && t < 64 + jb) {
                    const int c = t - 64;
                    if (c < s0 || c >= s1) {
                        const cx<T> a = A[(long)col * lda + k0 + c], v = A[(long)p * lda + k0 + c];
                        A[(long)p * lda + k0 + c] = a; A[(long)col * lda + k0 + c] = v;
                    }
                }
            }
            __syncthreads();
            const cx<T> pv = S[il * SLD + jl];
            const cx<T> ri = (pv.x != T(0) || pv.y != T(0)) ? crecip(pv) : cx<T>(T(0), T(0));
            if (t == 0) rinv[j] = ri;
            // 4. rank-1 update of the remaining columns of the sub-block (multipliers unscaled until the sub-block is done)
            const int ncol = s1 - j - 1;
            if (ncol > 0) {
                for (int e = t; e < (m - il - 1) * ncol; e += PLT) {
                    const int i = il + 1 + e / ncol, c = jl + 1 + e % ncol;
                    S[i * SLD + c] -= (S[i * SLD + jl] * ri) * S[il * SLD + c];
                }
            }
            __syncthreads();
        }
        // 5. scale the multipliers in LDS and write the sub-block back
        for (int e = t; e < m * PSB; e += PLT) {
            const int i = e / PSB, c = e - i * PSB;
            if (c < ks) {
                cx<T> v = S[i * SLD + c];
                if (i > c) { v = v * rinv[s0 + c]; S[i * SLD + c] = v; }
                A[(long)(rtop + i) * lda + k0 + s0 + c] = v;
            }
        }
        __syncthreads();
        if (s1 < jb) {
            // 6. U rows of the sub-block in the rest columns: row k gets the rows above it (unit lower triangle of the scaled multipliers)
            if (t < jb - s1) {
                const int c = s1 + t;
                for (int k = 0; k < ks; ++k) {
                    cx<T> v = A[(long)(rtop + k) * lda + k0 + c];
                    for (int kk = 0; kk < k; ++kk) v -= S[k * SLD + kk] * Ub[kk][c];
                    Ub[k][c] = v;
                    A[(long)(rtop + k) * lda + k0 + c] = v;
                }
            }
            __syncthreads();
            // 7. ... and the rows below in ONE pass: a[i][c] -= sum_k l[i][k] U[k][c]
            const int nrest = jb - s1;
            for (int e = t; e < (m - ks) * nrest; e += PLT) {
                const int i = ks + e / nrest, c = s1 + e % nrest;
                cx<T>* a = A + (long)(rtop + i) * lda + k0 + c;
                cx<T> v = *a;
#pragma unroll
                for (int k = 0; k < PSB; ++k) if (k < ks) v -= S[i * SLD + k] * Ub[k][c];
                *a = v;
            }
            __syncthreads();
        }
    }
}

// opt-in to the large dynamic LDS of lu_panel_lds_kernel: once per device and dtype, a failure is remembered (the old kernel then serves)
constexpr size_t PANEL_LDS_MAX = 154 * 1024;          // + < 5 KB of static LDS (U rows, reciprocals, reduction scratch) = the CU's 160 KB
template <class T, int PSB>
static bool panel_lds_ready() {
    static std::mutex mu;
    static int state[64];               // 0 = not yet set, 1 = ok, 2 = failed
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    int& st = state[dev & 63];
    if (st == 0) st = set_max_dyn_smem((const void*)lu_panel_lds_kernel<T, PSB>, PANEL_LDS_MAX) ? 2 : 1;
    return st == 1;
}

// ---- row-split panel for a FEW LARGE matrices ---------------------------------------------------------------------------------
// One workgroup per matrix streams the whole (n-k0) x jb panel through one CU jb times: 1.35 ms per panel at n = 5202, a quarter of
// the forward + adjoint step of a single [25,25] solve.  With few matrices the panel is instead cut into row blocks, W workgroups
// per matrix, ONE launch per panel column, and the pivoting is IMPLICIT inside the panel (no row is moved until the panel is done),
// which is what makes a launch free of cross-workgroup hazards:
//   lu_split_cand_kernel    candidates of column 0: each row block's largest |a| (LAPACK's |re|+|im|) -> cand[0][w] = row
//   lu_split_col_kernel(j)  every workgroup reduces the W candidates of column j to the pivot row p (the same result everywhere; block
//                           0 records it in piv[k0+j]), reads row p -- which nobody writes in this launch: a pivot row is frozen
//                           from the moment it is chosen -- and eliminates column j from its own rows that are not pivots yet
//                           (rank-1 update of columns j+1..jb-1), collecting on the way its candidate
//                           for column j+1 -> cand[(j+1)&1][w].  The rows chosen so far are the <= jb entries piv[k0..k0+j).
//   lu_split_scale_kernel   scales the multipliers (kept unscaled while the columns are eliminated, like in the one-workgroup kernel)
//   lu_split_final_kernel   turns the list of chosen rows into LAPACK's sequential interchanges (piv[k0+j] = position swapped with
//                           position k0+j) and applies them to the panel columns; lu_swap_range_kernel then does the other columns as
//                           usual.  Row k0+j ends up as [multipliers of columns < j | U row j]: the standard layout.
// The pivot of each column is the same largest-magnitude element as in the one-workgroup kernel (exact ties may resolve to another
// row: the candidates are ordered by their original row index, not by their current position).
// The candidate lists live in the not-yet-written tail of the pivot array (piv[k0+jb ...), 2 W ints), so the C ABI is unchanged;
// the split path is used while that tail is long enough and the panel tall enough (rows >= lu_split rows, default 1024).
constexpr int LSW_MAX = 64;             // workgroups per matrix
constexpr int LST = 256;                // threads per workgroup
constexpr int LSB = 8;                  // columns of a sub-block of the panel (the rank-1 updates stay inside it; the rest of the panel is updated per sub-block)

template <class T>
__device__ __forceinline__ void lu_split_rows(int n, int k0, int W, int w, int& r0, int& r1) {
    const int rows = n - k0, per = (rows + W - 1) / W;
    r0 = k0 + w * per;
    r1 = r0 + per < n ? r0 + per : n;
    if (r0 > n) r0 = n;
}

// block reduction of (value, row): larger value wins, ties -> smaller row; result valid in thread 0
template <class T>
__device__ __forceinline__ void lu_best_reduce(T& best, int& bi, T* red_v, int* red_i) {
    const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const T ov = __shfl_xor(best, o);
        const int oi = __shfl_xor(bi, o);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) { red_v[wid] = best; red_i[wid] = bi; }
    __syncthreads();
    if (t == 0) {
        for (int q = 1; q < LST / 64; ++q)
            if (red_v[q] > best || (red_v[q] == best && red_i[q] < bi)) { best = red_v[q]; bi = red_i[q]; }
    }
}

template <class T>
__global__ __launch_bounds__(LST) void lu_split_cand_kernel(const cx<T>* __restrict__ Aall, int lda, long sA, int n, int k0, int jb, int W,
                                                             int* __restrict__ piv_all) {
    __shared__ T red_v[LST / 64];
    __shared__ int red_i[LST / 64];
    const int b = blockIdx.y, w = blockIdx.x;
    const cx<T>* A = Aall + (long)b * sA;
    int* cand = piv_all + (long)b * n + k0 + jb;          // [2][W]
    int r0, r1;
    lu_split_rows<T>(n, k0, W, w, r0, r1);
    T best = T(-1);
    int bi = n;
    for (int r = r0 + threadIdx.x; r < r1; r += LST) {
        const T v = abs1(A[(long)r * lda + k0]);
        if (v > best) { best = v; bi = r; }
    }
    lu_best_reduce<T>(best, bi, red_v, red_i);
    if (threadIdx.x == 0) cand[w] = bi;                   // n = "no row" (empty block)
}

template <class T>
__global__ __launch_bounds__(LST) void lu_split_col_kernel(cx<T>* __restrict__ Aall, int lda, long sA, int n, int k0, int jb, int j, int W,
                                                            int* __restrict__ piv_all, int* __restrict__ info_all, int cend) {
    __shared__ cx<T> prow[NB];
    __shared__ int chosen[NB];
    __shared__ T red_v[LST / 64];
    __shared__ int red_i[LST / 64];
    __shared__ int s_p;
    const int b = blockIdx.y, w = blockIdx.x, t = threadIdx.x;
    cx<T>* A = Aall + (long)b * sA;
    int* piv = piv_all + (long)b * n;
    const int* cand = piv + k0 + jb + (j & 1) * W;
    int* cand_next = piv + k0 + jb + ((j + 1) & 1) * W;
    const int col = k0 + j;
    // pivot of column j: the largest of the W candidates (every workgroup computes the same)
    {
        T best = T(-1);
        int bi = n;
        if (t < W) {
            const int r = cand[t];
            if (r < n) { best = abs1(A[(long)r * lda + col]); bi = r; }
        }
        lu_best_reduce<T>(best, bi, red_v, red_i);
        if (t == 0) {
            if (bi >= n) {
                // no candidate at all: every remaining entry of the column failed `v > best` (NaN input).  Take the first row that is not
                // chosen yet (one of k0 .. k0 + j is free), so that the recorded pivot is always a row of THIS matrix -- piv[col] = n would make
                // the interchange kernels swap with row n, i.e. with the next matrix of the batch or past the buffer.  `best` stays -1:
                // info is raised below as for a zero pivot.
                for (int r = k0; r <= k0 + j; ++r) {
                    bool used = false;
                    for (int q = 0; q < j; ++q) used |= (piv[k0 + q] == r);
                    if (!used) { bi = r; break; }
                }
            }
            s_p = bi;
            if (w == 0) {
                piv[col] = bi;                            // chosen ROW (turned into an interchange by lu_split_final_kernel)
                if (!(best > T(0)) && info_all[b] == 0) info_all[b] = col + 1;
            }
        }
        __syncthreads();
    }
    const int p = s_p;
    if (t < j) chosen[t] = piv[k0 + t];
    if (t == j) chosen[j] = p;
    if (t < jb && p < n) prow[t] = A[(long)p * lda + k0 + t];
    __syncthreads();
    if (p >= n) { if (t == 0) cand_next[w] = n; return; }           // no row left (cannot happen while col < n)
    const cx<T> pv = prow[j];
    const cx<T> ri = (pv.x != T(0) || pv.y != T(0)) ? crecip(pv) : cx<T>(T(0), T(0));
    int r0, r1;
    lu_split_rows<T>(n, k0, W, w, r0, r1);
    // Columns j + 1 .. cend - 1 only: the rest of the panel gets this column's contribution later, eight columns at a time
    // (lu_split_sub_kernel).  8 lanes per row (one column each), 32 rows per pass, LUR passes in flight: the loads of LUR x 32 rows are
    // issued before the first result is needed (one pass at a time the loop was a chain of dependent L2 round trips).
    const int cpart = t & (LSB - 1), rpart = t / LSB;
    T best = T(-1);
    int bi = n;
    constexpr int LUR = 4;
    const int c1 = j + 1 + cpart;
    const bool c1on = c1 < cend;
    const cx<T> u1 = c1on ? prow[c1] : cx<T>(T(0), T(0));
    for (int rb = r0 + rpart; rb < r1; rb += LUR * (LST / LSB)) {
        cx<T> lv[LUR], a1[LUR];
        bool on[LUR];
#pragma unroll
        for (int u = 0; u < LUR; ++u) {
            const int r = rb + u * (LST / LSB);
            bool done = r >= r1;
            for (int q = 0; q <= j; ++q) done |= (chosen[q] == r);
            on[u] = !done;
            const cx<T>* row = A + (long)(done ? r0 : r) * lda + k0;          // (a valid address for the rows that sit out)
            lv[u] = row[j];
            a1[u] = row[c1on ? c1 : j];
        }
#pragma unroll
        for (int u = 0; u < LUR; ++u) {
            if (!on[u] || !c1on) continue;
            const int r = rb + u * (LST / LSB);
            cx<T>* row = A + (long)r * lda + k0;
            const cx<T> l = lv[u] * ri;             // the multiplier itself stays unscaled in memory until lu_split_scale_kernel
            const cx<T> v = a1[u] - l * u1;
            row[c1] = v;
            if (cpart == 0) { const T a = abs1(v); if (a > best) { best = a; bi = r; } }
            for (int c = c1 + LSB; c < cend; c += LSB) row[c] -= l * prow[c];        // (only with knob lu_sub = 1: the whole panel column by column)
        }
    }
    if (j + 1 < cend) {                              // (at the end of a sub-block the candidates of the next column come from lu_split_sub_kernel)
        lu_best_reduce<T>(best, bi, red_v, red_i);
        if (t == 0) cand_next[w] = bi;
    }
}

// U rows of the sub-block [s0, s1) in the panel columns [s1, jb), into LDS: U[k][c] = a[p_k][c] - sum_{s0 <= k' < k} l[p_k][k'] U[k'][c] with the
// (unscaled) multipliers a[p_k][k'] / u_k'k' -- the frozen pivot rows have not seen each other yet in these columns.  One thread per column.
template <class T>
__device__ __forceinline__ void lu_split_ublock(const cx<T>* __restrict__ A, int lda, int k0, int jb, int s0, int s1, const int* chosen, const cx<T>* rinv,
                                                cx<T> (*Ub)[NB]) {
    const int t = threadIdx.x;
    const int c = s1 + t;
    if (c < jb) {
        for (int k = s0; k < s1; ++k) {
            const cx<T>* row = A + (long)chosen[k] * lda + k0;
            cx<T> v = row[c];
            for (int kk = s0; kk < k; ++kk) v -= (row[kk] * rinv[kk]) * Ub[kk - s0][c];
            Ub[k - s0][c] = v;
        }
    }
}

// The sub-block [s0, s1) is factored: its contribution to the columns [s1, jb) of the panel in ONE pass (rank s1 - s0) over the rows that are
// not pivots yet, and the candidates of column s1.  Pivot rows are only read.
template <class T>
__global__ __launch_bounds__(LST) void lu_split_sub_kernel(cx<T>* __restrict__ Aall, int lda, long sA, int n, int k0, int jb, int s0, int s1, int W,
                                                            int* __restrict__ piv_all) {
    __shared__ cx<T> Ub[LSB][NB];
    __shared__ cx<T> rinv[NB];
    __shared__ int chosen[NB];
    __shared__ T red_v[LST / 64];
    __shared__ int red_i[LST / 64];
    const int b = blockIdx.y, w = blockIdx.x, t = threadIdx.x;
    cx<T>* A = Aall + (long)b * sA;
    int* piv = piv_all + (long)b * n;
    int* cand_next = piv + k0 + jb + (s1 & 1) * W;
    if (t < s1) chosen[t] = piv[k0 + t];
    __syncthreads();
    if (t >= s0 && t < s1) {
        const int p = chosen[t];
        cx<T> d(T(0), T(0));
        if (p < n) d = A[(long)p * lda + k0 + t];
        rinv[t] = (d.x != T(0) || d.y != T(0)) ? crecip(d) : cx<T>(T(0), T(0));
    }
    __syncthreads();
    lu_split_ublock<T>(A, lda, k0, jb, s0, s1, chosen, rinv, Ub);
    __syncthreads();
    int r0, r1;
    lu_split_rows<T>(n, k0, W, w, r0, r1);
    // 8 lanes per row, columns s1 + cpart, + 8, + 16; 32 rows per pass
    const int cpart = t & (LSB - 1), rpart = t / LSB;
    const int ks = s1 - s0;
    T best = T(-1);
    int bi = n;
    for (int r = r0 + rpart; r < r1; r += LST / LSB) {
        bool done = false;
        for (int q = 0; q < s1; ++q) done |= (chosen[q] == r);
        if (done) continue;
        cx<T>* row = A + (long)r * lda + k0;
        cx<T> l[LSB];
#pragma unroll
        for (int k = 0; k < LSB; ++k) l[k] = k < ks ? row[s0 + k] * rinv[s0 + k] : cx<T>(T(0), T(0));
#pragma unroll
        for (int cc = 0; cc < NB / LSB; ++cc) {
            const int c = s1 + cpart + LSB * cc;
            if (c >= jb) continue;
            cx<T> v = row[c];
#pragma unroll
            for (int k = 0; k < LSB; ++k) v -= l[k] * Ub[k][c];
            row[c] = v;
            if (c == s1) { const T a = abs1(v); if (a > best) { best = a; bi = r; } }
        }
    }
    lu_best_reduce<T>(best, bi, red_v, red_i);
    if (t == 0) cand_next[w] = bi;
}

// ... and the U rows themselves: the sub-block's pivot rows in the columns [s1, jb) (a launch of its own: lu_split_sub_kernel reads them)
template <class T>
__global__ __launch_bounds__(64) void lu_split_urow_kernel(cx<T>* __restrict__ Aall, int lda, long sA, int n, int k0, int jb, int s0, int s1,
                                                            const int* __restrict__ piv_all) {
    __shared__ cx<T> Ub[LSB][NB];
    __shared__ cx<T> rinv[NB];
    __shared__ int chosen[NB];
    const int b = blockIdx.x, t = threadIdx.x;
    cx<T>* A = Aall + (long)b * sA;
    const int* piv = piv_all + (long)b * n;
    if (t < s1) chosen[t] = piv[k0 + t];
    __syncthreads();
    if (t >= s0 && t < s1) {
        const int p = chosen[t];
        cx<T> d(T(0), T(0));
        if (p < n) d = A[(long)p * lda + k0 + t];
        rinv[t] = (d.x != T(0) || d.y != T(0)) ? crecip(d) : cx<T>(T(0), T(0));
    }
    __syncthreads();
    lu_split_ublock<T>(A, lda, k0, jb, s0, s1, chosen, rinv, Ub);
    __syncthreads();
    const int c = s1 + t;
    if (c < jb)
        for (int k = s0; k < s1; ++k) A[(long)chosen[k] * lda + k0 + c] = Ub[k - s0][c];
}

// L[r, j] = a[r, j] / u_jj for every column j at which row r was not a pivot yet (rows still at their original positions)
template <class T>
__global__ __launch_bounds__(LST) void lu_split_scale_kernel(cx<T>* __restrict__ Aall, int lda, long sA, int n, int k0, int jb, int W,
                                                              const int* __restrict__ piv_all) {
    __shared__ cx<T> rinv[NB];
    __shared__ int chosen[NB];
    const int b = blockIdx.y, w = blockIdx.x, t = threadIdx.x;
    cx<T>* A = Aall + (long)b * sA;
    const int* piv = piv_all + (long)b * n;
    if (t < jb) {
        const int p = piv[k0 + t];
        chosen[t] = p;
        cx<T> d(T(0), T(0));
        if (p < n) d = A[(long)p * lda + k0 + t];          // u_tt: the owner of row p scales only its columns < t
        rinv[t] = (d.x != T(0) || d.y != T(0)) ? crecip(d) : cx<T>(T(0), T(0));
    }
    __syncthreads();
    int r0, r1;
    lu_split_rows<T>(n, k0, W, w, r0, r1);
    const int cpart = t & 31, rpart = t >> 5;
    for (int r = r0 + rpart; r < r1; r += LST / 32) {
        int cend = jb;
        for (int q = 0; q < jb; ++q) if (chosen[q] == r) cend = q;
        if (cpart < cend) {
            cx<T>* e = A + (long)r * lda + k0 + cpart;
            *e = (*e) * rinv[cpart];
        }
    }
}

// Bookkeeping and row moves of this kernel were one thread's linear searches plus, per panel column, a chain of jb dependent swaps in global
// memory (200 us per launch at EVERY batch size: 0.12 s of a config-5 step, 2 % of a batch-16 step).  Now: the map "original row -> current
// position" of the <= 2 jb rows that move lives in LDS with ONE ENTRY PER LANE, each of the jb sequential steps finds its two entries by
// ballot; the moves themselves are not replayed swap by swap: no row has moved physically yet, so the final map says where every affected row
// goes -- all sources are staged in LDS (2 jb rows x jb columns, coalesced), then written to their destinations.  Same interchange sequence,
// same result, bit for bit.
static_assert(2 * NB <= 64, "lu_split_final_kernel keeps one map entry per lane of ONE 64-wide wavefront (gfx950: wave64 only)");
template <class T>
__global__ __launch_bounds__(64) void lu_split_final_kernel(cx<T>* __restrict__ Aall, int lda, long sA, int n, int k0, int jb, int* __restrict__ piv_all) {
    __shared__ int key[2 * NB], pos[2 * NB];             // original row -> current position, for the rows that have moved (entry l on lane l)
    __shared__ int seq[NB];
    __shared__ int nk_s;
    __shared__ cx<T> stage[2 * NB][NB + 1];
    const int b = blockIdx.x, t = threadIdx.x;
    cx<T>* A = Aall + (long)b * sA;
    int* piv = piv_all + (long)b * n;
    if (t == 0) nk_s = 0;
    __syncthreads();
    for (int j = 0; j < jb; ++j) {
        const int target = piv[k0 + j];                   // original row chosen for column j (uniform: every lane reads the same word)
        const int here = k0 + j;
        const int nk = nk_s;
        const int kt = t < nk ? key[t] : -1, pt = t < nk ? pos[t] : -1;
        const unsigned long long m_tgt = __ballot(kt == target);       // where the target row sits now
        const unsigned long long m_here = __ballot(pt == here);        // which original row sits at position `here`
        const int lt = m_tgt ? __builtin_ctzll(m_tgt) : -1, lh = m_here ? __builtin_ctzll(m_here) : -1;
        const int q = lt >= 0 ? __shfl(pt, lt) : target;
        const int o = lh >= 0 ? __shfl(kt, lh) : here;
        __syncthreads();
        if (t == 0) {
            seq[j] = q;
            if (q != here) {
                int nn = nk;
                if (lt >= 0) pos[lt] = here; else { key[nn] = target; pos[nn] = here; ++nn; }          // put(target, here)
                // put(o, q): o's entry -- it may be the one just appended (o == target cannot happen: q != here)
                int lo = lh;
                if (lo < 0) { for (int z = nk; z < nn; ++z) if (key[z] == o) lo = z; }
                if (lo >= 0) pos[lo] = q; else { key[nn] = o; pos[nn] = q; ++nn; }
                nk_s = nn;
            }
        }
        __syncthreads();
    }
    if (t < jb) piv[k0 + t] = seq[t];
    // every entry (key -> pos) with key != pos is a row of the panel that moves from its original place `key` to `pos`
    const int nk = nk_s;
    for (int e = t; e < nk * jb; e += 64) {
        const int l = e / jb, c = e - l * jb;
        stage[l][c] = A[(long)key[l] * lda + k0 + c];
    }
    __syncthreads();
    for (int e = t; e < nk * jb; e += 64) {
        const int l = e / jb, c = e - l * jb;
        if (pos[l] != key[l]) A[(long)pos[l] * lda + k0 + c] = stage[l][c];
    }
}

// Apply the interchanges piv[k0 .. k0+np) to the columns [a0, a1) and [b0, b1) (the two column windows either side of an outer block).
template <class T>
__global__ __launch_bounds__(256) void lu_swap_range_kernel(cx<T>* __restrict__ Aall, int lda, long sA, int a0, int a1, int b0, int b1, int k0, int np,
                                                             const int* __restrict__ piv_all, int n) {
    const int b = blockIdx.y;
    cx<T>* A = Aall + (long)b * sA;
    const int* piv = piv_all + (long)b * n;
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= (a1 - a0) + (b1 - b0)) return;
    c = c < a1 - a0 ? a0 + c : b0 + (c - (a1 - a0));
    for (int j = 0; j < np; ++j) {
        const int r = k0 + j, p = piv[r];
        if (p != r) {
            cx<T> t = A[(long)r * lda + c];
            A[(long)r * lda + c] = A[(long)p * lda + c];
            A[(long)p * lda + c] = t;
        }
    }
}

// Apply ALL interchanges (rows 0..n-1, in order) to the right-hand sides.
template <class T>
__global__ __launch_bounds__(256) void rhs_permute_kernel(cx<T>* __restrict__ Ball, int ldb, long sB, int nrhs, const int* __restrict__ piv_all, int n) {
    const int b = blockIdx.y;
    cx<T>* B = Ball + (long)b * sB;
    const int* piv = piv_all + (long)b * n;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nrhs) return;
    for (int r = 0; r < n; ++r) {
        const int p = piv[r];
        if (p != r) {
            cx<T> a = B[(long)r * ldb + c];
            B[(long)r * ldb + c] = B[(long)p * ldb + c];
            B[(long)p * ldb + c] = a;
        }
    }
}

// X = L^-1 B (UNIT lower, UPPER=false) or X = U^-1 B (non-unit upper, UPPER=true); the jb x jb triangle sits at
// Tm (leading dimension ldt); B is jb x ncols, in place; one thread per column.
template <class T, bool UPPER>
__global__ __launch_bounds__(256) void trsm_kernel(const cx<T>* __restrict__ Tall, int ldt, long sT, int jb,
                                                   cx<T>* __restrict__ Ball, int ldb, long sB, int ncols) {
    __shared__ cx<T> Ts[NB][NB + 1];
    const int b = blockIdx.y;
    const cx<T>* Tm = Tall + (long)b * sT;
    cx<T>* B = Ball + (long)b * sB;
    for (int e = threadIdx.x; e < jb * jb; e += blockDim.x) {
        const int r = e / jb, c = e % jb;
        cx<T> v = Tm[(long)r * ldt + c];
        if (UPPER && r == c) v = crecip(v);
        Ts[r][c] = v;
    }
    __syncthreads();
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= ncols) return;
    cx<T> x[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) x[i] = (i < jb) ? B[(long)i * ldb + c] : cx<T>(T(0), T(0));
    if (!UPPER) {
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if (i < jb) {
                cx<T> s = x[i];
#pragma unroll
                for (int q = 0; q < NB; ++q)
                    if (q < i) cfma(s, -Ts[i][q], x[q]);
                x[i] = s;
            }
        }
    } else {
#pragma unroll
        for (int i = NB - 1; i >= 0; --i) {
            if (i < jb) {
                cx<T> s = x[i];
#pragma unroll
                for (int q = NB - 1; q >= 0; --q)
                    if (q > i && q < jb) cfma(s, -Ts[i][q], x[q]);
                x[i] = s * Ts[i][i];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
        if (i < jb) B[(long)i * ldb + c] = x[i];
}

// X = B U^-1 (non-unit upper, UPPER) or X = B L^-1 (UNIT lower) for the jb x jb triangle at Tm: B is nrows x jb, in place; one thread per ROW
// (a row's jb elements are contiguous: each lane reads and writes whole cache lines of its own).
template <class T, bool UPPER>
__global__ __launch_bounds__(256) void trsm_right_kernel(const cx<T>* __restrict__ Tall, int ldt, long sT, int jb,
                                                         cx<T>* __restrict__ Ball, int ldb, long sB, int nrows) {
    __shared__ cx<T> Ts[NB][NB + 1];
    const int b = blockIdx.y;
    const cx<T>* Tm = Tall + (long)b * sT;
    cx<T>* B = Ball + (long)b * sB;
    for (int e = threadIdx.x; e < jb * jb; e += blockDim.x) {
        const int r = e / jb, c = e % jb;
        cx<T> v = Tm[(long)r * ldt + c];
        if (UPPER && r == c) v = crecip(v);
        Ts[r][c] = v;
    }
    __syncthreads();
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    cx<T>* row = B + (long)r * ldb;
    cx<T> x[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) x[i] = (i < jb) ? row[i] : cx<T>(T(0), T(0));
    if (UPPER) {                 // x U = b: columns ascending, x_j = (b_j - sum_{i<j} x_i U_ij) / U_jj
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (j < jb) {
                cx<T> sacc = x[j];
#pragma unroll
                for (int i = 0; i < NB; ++i)
                    if (i < j) cfma(sacc, -Ts[i][j], x[i]);
                x[j] = sacc * Ts[j][j];
            }
        }
    } else {                     // x L = b, unit diagonal: columns descending, x_j = b_j - sum_{i>j} x_i L_ij
#pragma unroll
        for (int j = NB - 1; j >= 0; --j) {
            if (j < jb) {
                cx<T> sacc = x[j];
#pragma unroll
                for (int i = NB - 1; i >= 0; --i)
                    if (i > j && i < jb) cfma(sacc, -Ts[i][j], x[i]);
                x[j] = sacc;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
        if (i < jb) row[i] = x[i];
}

// idx[c] = the column of Z that ends up in column c of X = Z P (P = P_{n-1} ... P_0, the row interchanges of P A = L U): the column swaps
// (k, piv[k]) applied for k = n-1 down to 0 to the identity.  One thread per matrix, the index array in LDS.
__global__ __launch_bounds__(64) void perm_compose_kernel(const int* __restrict__ piv_all, int n, int* __restrict__ idx_all) {
    TRX_DYN_SMEM(smem);
    int* idx = reinterpret_cast<int*>(smem);
    const int b = blockIdx.x;
    const int* piv = piv_all + (long)b * n;
    for (int c = threadIdx.x; c < n; c += 64) idx[c] = c;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int k = n - 1; k >= 0; --k) {
            const int p = piv[k];
            if (p != k) { const int a = idx[k]; idx[k] = idx[p]; idx[p] = a; }
        }
    __syncthreads();
    for (int c = threadIdx.x; c < n; c += 64) idx_all[(long)b * n + c] = idx[c];
}

// X[r, c] = Z[r, idx[c]] in place: a workgroup takes rows one at a time through LDS (coalesced both ways, the gather happens in LDS)
template <class T>
__global__ __launch_bounds__(256) void col_permute_kernel(cx<T>* __restrict__ Ball, int ldb, long sB, int nrows, int n, const int* __restrict__ idx_all, int rows_per_block) {
    TRX_DYN_SMEM(smem);
    cx<T>* rowbuf = reinterpret_cast<cx<T>*>(smem);            // [n]
    const int b = blockIdx.y;
    cx<T>* B = Ball + (long)b * sB;
    const int* idx = idx_all + (long)b * n;
    const int r0 = blockIdx.x * rows_per_block;
    for (int r = r0; r < r0 + rows_per_block && r < nrows; ++r) {
        cx<T>* row = B + (long)r * ldb;
        for (int c = threadIdx.x; c < n; c += blockDim.x) rowbuf[c] = row[c];
        __syncthreads();
        for (int c = threadIdx.x; c < n; c += blockDim.x) row[c] = rowbuf[idx[c]];
        __syncthreads();
    }
}

}  // namespace

// Two-level blocking: panels of NB columns are factored one at a time, but the update of everything to the right of
// the current OUTER block (NBO = 8 panels) is delayed until the whole outer block is done, so the dominant trailing
// update is a rank-256 GEMM (compute-bound on the matrix cores) instead of four rank-32 updates (HBM-bound: each
// re-reads and re-writes the trailing matrix).
constexpr int NBO = 8 * NB;

static int lu_sub_env() { const char* e = getenv("TRX_LU_SUB"); const int v = e ? atoi(e) : 0; return (v >= 0 && v <= 2) ? v : 0; }
static int g_lu_sub = lu_sub_env();       // trx_tuning("lu_sub", v): 0 = panels in sub-blocks of 8 (4: tall panels) columns (default), 1 = column by column as in rounds 1 - 5, 2 = sub-blocks of 4 everywhere (tests)
static int lu_split_rows_env() { const char* e = getenv("TRX_LU_SPLIT"); const int v = e ? atoi(e) : 0; return v >= 0 ? v : 0; }
static int g_lu_split_rows = lu_split_rows_env();   // 0 = automatic (panels too tall for the LDS-resident kernel); trx_tuning("lu_split", rows) / TRX_LU_SPLIT; 1 = never split
static int lu_split_batch_env() { const char* e = getenv("TRX_LU_SPLIT_BATCH"); return e ? atoi(e) : 0; }
static int g_lu_split_batch = lu_split_batch_env();          // 0 = automatic (any batch): largest batch the split panel is used for; trx_tuning("lu_split_batch", b)
int lu_set_knob(const char* key, int value) {
    const std::string k(key);
    if (value < 0 || value > (1 << 20)) return TRX_ERR_ARG;
    if (k == "lu_split") g_lu_split_rows = value;
    else if (k == "lu_split_batch") g_lu_split_batch = value;
    else if (k == "lu_sub") { if (value > 2) return TRX_ERR_ARG; g_lu_sub = value; }
    else return TRX_ERR_ARG;
    return TRX_OK;
}

// ---- outer blocks --------------------------------------------------------------------------------------------------------------
// An outer block's panels apply their row interchanges inside the block's columns only; the columns left and right of the block receive all
// kb interchanges at once when the block is finished (lu_swap_range_kernel).  (A look-ahead on top of this split -- the next block's panels
// on a side stream under the trailing update of the remaining columns -- was built and measured in round 4: bit-identical factors, but a
// wash in time at every batch size, and its GPU test was not deterministic; removed.  profiles/r04_ab/r4_lu_look.txt)
// Panels of the outer block [K0, Kend) on stream s; row interchanges applied inside the block's columns only.
template <class T>
int lu_block_panels(hipStream_t s, cx<T>* A, int lda, long sA, int n, int K0, int Kend, int* piv, int batch, int* info) {
    const cx<T> one(T(1), T(0)), mone(T(-1), T(0));
    auto at = [&](int r, int c) { return A + (long)r * lda + c; };
    for (int c0 = K0; c0 < Kend; c0 += NB) {
        const int jb = (Kend - c0 < NB) ? (Kend - c0) : NB;
        // few large matrices: row-split panel (see above); needs 2 W ints of the pivot array's unwritten tail
        const int rows = n - c0;
        // one workgroup per matrix with the sub-block in LDS whenever it fits (sub-blocks of 8 columns, of 4 for panels twice as tall):
        // ~5 - 9 us per column at any batch size, where the row-split kernels take 35 (fp64, batch 128: 21 per column launch + 12 for
        // the sub-block passes + the scaling).  lu_split = rows (explicit): the row-split panel from that height on, as in rounds 3 - 5.
        const int psb = g_lu_sub == 1 ? 0 : (g_lu_sub != 2 && sizeof(cx<T>) * (size_t)rows * 9 <= PANEL_LDS_MAX && panel_lds_ready<T, 8>()) ? 8
                        : (sizeof(cx<T>) * (size_t)rows * 5 <= PANEL_LDS_MAX && panel_lds_ready<T, 4>()) ? 4 : 0;
        const int split_min = g_lu_split_rows ? g_lu_split_rows : (psb ? (1 << 30) : 1024);
        // workgroups per matrix: about 512 per launch over the batch, at least 64 rows each (a larger batch already supplies
        // workgroups, but the one-workgroup panel still leaves half of the CUs idle at batch 128: knob lu_split_batch)
        const int split_batch = g_lu_split_batch ? g_lu_split_batch : (1 << 20);
        int W = rows / 64 < LSW_MAX ? rows / 64 : LSW_MAX;
        const int wcap = 512 / batch > 2 ? 512 / batch : 2;
        if (W > wcap) W = wcap;
        if (batch <= split_batch && g_lu_split_rows != 1 && rows >= split_min && W >= 2 && n - c0 - jb >= 2 * W) {
            ProfScope prof(PROF_LU_PANEL, s, 0, 0);
            TRX_LAUNCH((lu_split_cand_kernel<T>), dim3(W, batch), dim3(LST), 0, s, (const cx<T>*)A, lda, sA, n, c0, jb, W, piv);
            const int lsb = g_lu_sub == 1 ? jb : LSB;
            for (int s0 = 0; s0 < jb; s0 += lsb) {
                const int s1 = s0 + lsb < jb ? s0 + lsb : jb;
                for (int j = s0; j < s1; ++j)
                    TRX_LAUNCH((lu_split_col_kernel<T>), dim3(W, batch), dim3(LST), 0, s, A, lda, sA, n, c0, jb, j, W, piv, info, s1);
                if (s1 < jb) {
                    TRX_LAUNCH((lu_split_sub_kernel<T>), dim3(W, batch), dim3(LST), 0, s, A, lda, sA, n, c0, jb, s0, s1, W, piv);
                    TRX_LAUNCH((lu_split_urow_kernel<T>), dim3(batch), dim3(64), 0, s, A, lda, sA, n, c0, jb, s0, s1, (const int*)piv);
                }
            }
            TRX_LAUNCH((lu_split_scale_kernel<T>), dim3(W, batch), dim3(LST), 0, s, A, lda, sA, n, c0, jb, W, (const int*)piv);
            TRX_LAUNCH((lu_split_final_kernel<T>), dim3(batch), dim3(64), 0, s, A, lda, sA, n, c0, jb, piv);
        } else {
            ProfScope prof(PROF_LU_PANEL, s, 0, 0);
            if (psb == 8)
                TRX_LAUNCH((lu_panel_lds_kernel<T, 8>), dim3(batch), dim3(PLT), sizeof(cx<T>) * (size_t)rows * 9, s, A, lda, sA, n, c0, jb, piv, info);
            else if (psb == 4)
                TRX_LAUNCH((lu_panel_lds_kernel<T, 4>), dim3(batch), dim3(PLT), sizeof(cx<T>) * (size_t)rows * 5, s, A, lda, sA, n, c0, jb, piv, info);
            else
                TRX_LAUNCH((lu_panel_kernel<T>), dim3(batch), dim3(512), 0, s, A, lda, sA, n, c0, jb, piv, info);
        }
        // the panel's interchanges on the OTHER columns of this outer block (left: factors of the block's earlier panels; right: columns
        // of the block still to be factored); the columns outside the block follow when the block is done
        const int inblock = (Kend - K0) - jb;
        if (inblock > 0)
            TRX_LAUNCH((lu_swap_range_kernel<T>), dim3(cdiv_i(inblock, 256), batch), dim3(256), 0, s, A, lda, sA, K0, c0, c0 + jb, Kend, c0, jb, (const int*)piv, n);
        const int wcols = Kend - (c0 + jb);       // columns of the outer block still to be factored
        if (wcols > 0) {
            TRX_LAUNCH((trsm_kernel<T, false>), dim3(cdiv_i(wcols, 256), batch), dim3(256), 0, s, (const cx<T>*)at(c0, c0), lda, sA, jb,
                       at(c0, c0 + jb), lda, sA, wcols);
            const int rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n - c0 - jb, wcols, jb, mone, at(c0 + jb, c0), lda, sA, at(c0, c0 + jb), lda, sA, one,
                                   at(c0 + jb, c0 + jb), lda, sA, batch);
            if (rc) return rc;
        }
    }
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

// Triangular solve with the diagonal block [r0, r1) of a factored matrix (UPPER: U; otherwise unit-lower L) on the rows [r0, r1) of B, in
// place, by recursive halving down to NB-row leaves (trsm_kernel).  Same arithmetic as marching through the block leaf by leaf with a
// rank-NB update of all rows still to come after each leaf -- what rounds 1 - 5 did -- but that form reads and writes the remaining rows of
// B once per leaf (3.5 block heights of traffic per block; the rank-32 fp64 updates ran at the HBM roof: 247 ms of a 128-point step,
// profiles/r05_pmc_bench.txt), where the halving form touches every row once per LEVEL (1.5 block heights) and gives the upper levels
// K = 128 / 64 products that are bound by the matrix cores instead.
template <class T, bool UPPER>
static int tri_block_solve(hipStream_t s, const cx<T>* LU, int lda, long sA, int r0, int r1, cx<T>* B, int ldb, long sB, int nrhs, int batch) {
    const cx<T> one(T(1), T(0)), mone(T(-1), T(0));
    const int m = r1 - r0;
    if (m <= 0) return TRX_OK;
    if (m <= NB) {
        TRX_LAUNCH((trsm_kernel<T, UPPER>), dim3(cdiv_i(nrhs, 256), batch), dim3(256), 0, s, LU + (long)r0 * lda + r0, lda, sA, m,
                   B + (long)r0 * ldb, ldb, sB, nrhs);
        return TRX_OK;
    }
    int h = NB;                                   // largest NB * 2^k below m: full blocks split 128 | 128, 64 | 64, 32 | 32
    while (2 * h < m) h *= 2;
    const int mid = r0 + h;
    int rc;
    if (!UPPER) {
        if ((rc = tri_block_solve<T, false>(s, LU, lda, sA, r0, mid, B, ldb, sB, nrhs, batch))) return rc;
        if ((rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, r1 - mid, nrhs, h, mone, LU + (long)mid * lda + r0, lda, sA, B + (long)r0 * ldb, ldb, sB, one,
                          B + (long)mid * ldb, ldb, sB, batch))) return rc;
        return tri_block_solve<T, false>(s, LU, lda, sA, mid, r1, B, ldb, sB, nrhs, batch);
    }
    if ((rc = tri_block_solve<T, true>(s, LU, lda, sA, mid, r1, B, ldb, sB, nrhs, batch))) return rc;
    if ((rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, h, nrhs, r1 - mid, mone, LU + (long)r0 * lda + mid, lda, sA, B + (long)mid * ldb, ldb, sB, one,
                      B + (long)r0 * ldb, ldb, sB, batch))) return rc;
    return tri_block_solve<T, true>(s, LU, lda, sA, r0, mid, B, ldb, sB, nrhs, batch);
}

// U rows of the outer block [K0, Kend) for the columns [c_lo, c_hi) right of it (trsm + in-block update per panel), then the rank-kb
// update of the rows below the block in those columns.
template <class T>
int lu_block_update(hipStream_t s, cx<T>* A, int lda, long sA, int n, int K0, int Kend, int c_lo, int c_hi, int batch) {
    const cx<T> one(T(1), T(0)), mone(T(-1), T(0));
    auto at = [&](int r, int c) { return A + (long)r * lda + c; };
    const int tcols = c_hi - c_lo, kb = Kend - K0;
    if (tcols <= 0) return TRX_OK;
    int rc;
    rc = tri_block_solve<T, false>(s, A, lda, sA, K0, Kend, at(0, c_lo), lda, sA, tcols, batch);
    if (rc) return rc;
    if (n - Kend > 0) {
        rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n - Kend, tcols, kb, mone, at(Kend, K0), lda, sA, at(K0, c_lo), lda, sA, one, at(Kend, c_lo), lda, sA, batch);
        if (rc) return rc;
    }
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

// (Two sub-batches on side streams -- one's panel chains under the other's trailing update -- were measured in round 6: 33.61 against 33.43
// layer-solves/s at batch 128, 15.76 against 15.98 at batch 16: noise, removed; profiles/r06_ab/cumask.txt.)
template <class T>
int lu_factor(hipStream_t s, cx<T>* A, int lda, long sA, int n, int* piv, int batch, int* info) {
    if (n <= 0 || batch <= 0) return TRX_OK;
    if (hipMemsetAsync(info, 0, sizeof(int) * batch, s) != hipSuccess) return TRX_ERR_LAUNCH;
    for (int K0 = 0; K0 < n; K0 += NBO) {
        const int kb = (n - K0 < NBO) ? (n - K0) : NBO;
        const int Kend = K0 + kb;
        int rc = lu_block_panels<T>(s, A, lda, sA, n, K0, Kend, piv, batch, info);
        if (rc) return rc;
        // the block's interchanges on the columns left and right of it
        const int outside = K0 + (n - Kend);
        if (outside > 0)
            TRX_LAUNCH((lu_swap_range_kernel<T>), dim3(cdiv_i(outside, 256), batch), dim3(256), 0, s, A, lda, sA, 0, K0, Kend, n, K0, kb, (const int*)piv, n);
        rc = lu_block_update<T>(s, A, lda, sA, n, K0, Kend, Kend, n, batch);
        if (rc) return rc;
    }
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

template <class T>
int lu_solve(hipStream_t s, const cx<T>* LU, int lda, long sA, int n, const int* piv, cx<T>* B, int ldb, long sB,
             int nrhs, int batch) {
    if (n <= 0 || nrhs <= 0 || batch <= 0) return TRX_OK;
    const cx<T> one(T(1), T(0)), mone(T(-1), T(0));
    const dim3 cg(cdiv_i(nrhs, 256), batch);
    auto lu = [&](int r, int c) { return LU + (long)r * lda + c; };
    auto bb = [&](int r) { return B + (long)r * ldb; };
    int rc;
    TRX_LAUNCH((rhs_permute_kernel<T>), cg, dim3(256), 0, s, B, ldb, sB, nrhs, piv, n);
    for (int K0 = 0; K0 < n; K0 += NBO) {            // forward: L Y = P B
        const int kb = (n - K0 < NBO) ? (n - K0) : NBO;
        const int Kend = K0 + kb;
        rc = tri_block_solve<T, false>(s, LU, lda, sA, K0, Kend, B, ldb, sB, nrhs, batch);
        if (rc) return rc;
        if (n - Kend > 0) {
            rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n - Kend, nrhs, kb, mone, lu(Kend, K0), lda, sA, bb(K0), ldb, sB, one, bb(Kend), ldb, sB, batch);
            if (rc) return rc;
        }
    }
    const int lastK = ((n - 1) / NBO) * NBO;
    for (int K0 = lastK; K0 >= 0; K0 -= NBO) {       // backward: U X = Y
        const int kb = (n - K0 < NBO) ? (n - K0) : NBO;
        const int Kend = K0 + kb;
        rc = tri_block_solve<T, true>(s, LU, lda, sA, K0, Kend, B, ldb, sB, nrhs, batch);
        if (rc) return rc;
        if (K0 > 0) {
            rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, K0, nrhs, kb, mone, lu(0, K0), lda, sA, bb(K0), ldb, sB, one, bb(0), ldb, sB, batch);
            if (rc) return rc;
        }
    }
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

// Right-sided counterpart of tri_block_solve: X U = B / X L = B on the COLUMNS [r0, r1) of B (nrows rows), by the same recursive halving.
template <class T, bool UPPER>
static int tri_block_solve_right(hipStream_t s, const cx<T>* LU, int lda, long sA, int r0, int r1, cx<T>* B, int ldb, long sB, int nrows, int batch) {
    const cx<T> one(T(1), T(0)), mone(T(-1), T(0));
    const int m = r1 - r0;
    if (m <= 0) return TRX_OK;
    if (m <= NB) {
        TRX_LAUNCH((trsm_right_kernel<T, UPPER>), dim3(cdiv_i(nrows, 256), batch), dim3(256), 0, s, LU + (long)r0 * lda + r0, lda, sA, m, B + r0, ldb, sB, nrows);
        return TRX_OK;
    }
    int h = NB;
    while (2 * h < m) h *= 2;
    const int mid = r0 + h;
    int rc;
    if (UPPER) {       // columns ascending: X[:, mid:r1] needs X[:, r0:mid] U[r0:mid, mid:r1]
        if ((rc = tri_block_solve_right<T, true>(s, LU, lda, sA, r0, mid, B, ldb, sB, nrows, batch))) return rc;
        if ((rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, nrows, r1 - mid, h, mone, B + r0, ldb, sB, LU + (long)r0 * lda + mid, lda, sA, one, B + mid, ldb, sB, batch))) return rc;
        return tri_block_solve_right<T, true>(s, LU, lda, sA, mid, r1, B, ldb, sB, nrows, batch);
    }
    // columns descending: X[:, r0:mid] needs X[:, mid:r1] L[mid:r1, r0:mid]
    if ((rc = tri_block_solve_right<T, false>(s, LU, lda, sA, mid, r1, B, ldb, sB, nrows, batch))) return rc;
    if ((rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, nrows, h, r1 - mid, mone, B + mid, ldb, sB, LU + (long)mid * lda + r0, lda, sA, one, B + r0, ldb, sB, batch))) return rc;
    return tri_block_solve_right<T, false>(s, LU, lda, sA, r0, mid, B, ldb, sB, nrows, batch);
}

// X = B A^-1 from the factors of P A = L U (lu_factor), in place on B [nrows x n]:  X = B U^-1 L^-1 P.  idx: n ints per matrix of scratch.
// Replaces the transposed-system route of rounds 1 - 5 (A^T factored, B^T solved from the left, the result transposed back: three tiled
// transposes per right solve); same blocking as lu_solve with rows and columns exchanged.
template <class T>
int lu_solve_right(hipStream_t s, const cx<T>* LU, int lda, long sA, int n, const int* piv, cx<T>* B, int ldb, long sB, int nrows, int batch, int* idx) {
    if (n <= 0 || nrows <= 0 || batch <= 0) return TRX_OK;
    const cx<T> one(T(1), T(0)), mone(T(-1), T(0));
    int rc;
    for (int K0 = 0; K0 < n; K0 += NBO) {            // Y U = B: column blocks ascending
        const int kb = (n - K0 < NBO) ? (n - K0) : NBO, Kend = K0 + kb;
        if ((rc = tri_block_solve_right<T, true>(s, LU, lda, sA, K0, Kend, B, ldb, sB, nrows, batch))) return rc;
        if (n - Kend > 0 &&
            (rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, nrows, n - Kend, kb, mone, B + K0, ldb, sB, LU + (long)K0 * lda + Kend, lda, sA, one, B + Kend, ldb, sB, batch))) return rc;
    }
    const int lastK = ((n - 1) / NBO) * NBO;
    for (int K0 = lastK; K0 >= 0; K0 -= NBO) {       // Z L = Y: column blocks descending
        const int kb = (n - K0 < NBO) ? (n - K0) : NBO, Kend = K0 + kb;
        if ((rc = tri_block_solve_right<T, false>(s, LU, lda, sA, K0, Kend, B, ldb, sB, nrows, batch))) return rc;
        if (K0 > 0 &&
            (rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, nrows, K0, kb, mone, B + K0, ldb, sB, LU + (long)K0 * lda, lda, sA, one, B, ldb, sB, batch))) return rc;
    }
    // X = Z P
    const size_t smi = sizeof(int) * (size_t)n, smr = sizeof(cx<T>) * (size_t)n;
    if (set_max_dyn_smem((const void*)perm_compose_kernel, smi) || set_max_dyn_smem((const void*)col_permute_kernel<T>, smr)) return TRX_ERR_LAUNCH;
    TRX_LAUNCH(perm_compose_kernel, dim3(batch), dim3(64), smi, s, piv, n, idx);
    const int rpb = 4;
    TRX_LAUNCH((col_permute_kernel<T>), dim3(cdiv_i(nrows, rpb), batch), dim3(256), smr, s, B, ldb, sB, nrows, n, (const int*)idx, rpb);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

template int lu_factor<float>(hipStream_t, cx<float>*, int, long, int, int*, int, int*);
template int lu_factor<double>(hipStream_t, cx<double>*, int, long, int, int*, int, int*);
template int lu_solve<float>(hipStream_t, const cx<float>*, int, long, int, const int*, cx<float>*, int, long, int, int);
template int lu_solve<double>(hipStream_t, const cx<double>*, int, long, int, const int*, cx<double>*, int, long, int, int);
template int lu_solve_right<float>(hipStream_t, const cx<float>*, int, long, int, const int*, cx<float>*, int, long, int, int, int*);
template int lu_solve_right<double>(hipStream_t, const cx<double>*, int, long, int, const int*, cx<double>*, int, long, int, int, int*);

}  // namespace trx
