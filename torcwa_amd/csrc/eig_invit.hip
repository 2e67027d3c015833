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
#include <cstdlib>
#include <limits>
#include <string>
#include "prof.hpp"

namespace trx {

static int invit_cfg_env() {
    const char* e = getenv("TRX_INVIT_CFG");
    const int v = e ? atoi(e) : 0;
    return (v >= 0 && v <= 6) ? v : 0;
}
static int g_invit_cfg = invit_cfg_env(), g_invit_min_wpl = 0, g_invit_ring = 0, g_invit_xcd = 0, g_invit_dbg = 0;      // xcd: 0 = XCD-aware launches, 1 = plain 2-D grid      // trx_tuning("invit_cfg" / "invit_wpl", v): kernel layout, minimum waves per eigenvalue (tests)
int invit_set_knob(const char* key, int value) {
    const std::string k(key);
    if (k == "invit_cfg" && value >= 0 && value <= 6) { g_invit_cfg = value; return TRX_OK; }
#ifdef TRX_INVIT_DEBUG
    // timing experiments of round 3 only: the bits SKIP barriers and parts of the solve step, i.e. they change results -- not a tuning knob of
    // the library (include/trx.h: "results do not depend on any knob"), compiled in only with -DTRX_INVIT_DEBUG
    if (k == "invit_dbg" && value >= 0 && value <= 15) { g_invit_dbg = value; return TRX_OK; }
#endif
    if (k == "invit_xcd" && value >= 0 && value <= 1) { g_invit_xcd = value; return TRX_OK; }
    if (k == "invit_ring" && value >= 0 && value <= 3) { g_invit_ring = value; return TRX_OK; }
    if (k == "invit_wpl" && (value == 0 || value == 1 || value == 2 || value == 4 || value == 8)) { g_invit_min_wpl = value; return TRX_OK; }
    return TRX_ERR_ARG;
}

namespace {

// what the lane that owns the pivot row of a step publishes for everybody: multiplier, y_j and the interchange flag of that step
template <class T>
struct IvPivot {
    cx<T> m, y;
    int swap, pad[3];
};

// 1 / z for a pivot: |z| >= eps ||H|| and the entries of a balanced operator are far from the ends of the exponent range, so |z|^2 is
// formed directly (one division instead of the three of the overflow-safe quotient)
__device__ __forceinline__ double fast_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);          // v_rcp_f64, then two Newton steps (the full IEEE division sequence is three times as long)
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ float fast_rcp(float d) {
    float r = __builtin_amdgcn_rcpf(d);
    r = fmaf(fmaf(-d, r, 1.0f), r, r);
    return r;
}
template <class T>
__device__ __forceinline__ cx<T> pivot_recip(cx<T> z) {
    const T r = fast_rcp(norm2(z));
    return cx<T>(z.x * r, -z.y * r);
}

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

// grid (ceil(n / LPW), batch).  Y[j, k], M[j, k], SW[j, k]: y_j, multiplier and interchange flag of step j of eigenvalue k.
template <class T, int WPL, int SL, int NT, int D>
__global__ __launch_bounds__(NT) void invit_solve_kernel(const cx<T>* __restrict__ Ht_all, int n, const cx<T>* __restrict__ lam_all, const T* __restrict__ hnorm,
                                                         cx<T>* __restrict__ Yall, cx<T>* __restrict__ Mall, unsigned char* __restrict__ SWall, int ring) {
    TRX_DYN_SMEM(smem);
    constexpr int LW = 64 * WPL, LPW = (NT / 64) / WPL;
    const int pstride = n;
    cx<T>* pbuf = reinterpret_cast<cx<T>*>(smem);                              // [2 or 1][n]
    IvPivot<T>* pv = reinterpret_cast<IvPivot<T>*>(pbuf + (size_t)ring * n);     // [2][LPW]
    // LDS ring of `ring` columns (column c in slot c mod ring).  3: the step's column AND the next one are resident, so the subdiagonal the
    // next pivot competes with is read from LDS (a global load per step would sit on the critical path with its full latency);
    // 2: double buffer, that one element comes from memory; 1: one buffer and a second barrier per step (very large n only).
    const int La = ring == 3 ? 2 : 1;                                           // columns resident ahead of the one in use
    auto slot = [&](int c) __attribute__((always_inline)) { return pbuf + (ring == 1 ? 0 : (c % ring) * pstride); };
    const int bm = blockIdx.y, t = threadIdx.x;
    const cx<T>* Ht = Ht_all + (long)bm * n * n;
    const int ll = t / LW, L = t - ll * LW;                                    // eigenvalue within the workgroup, lane within the eigenvalue
    int k = blockIdx.x * LPW + ll;
    const bool valid = k < n;
    if (!valid) k = n - 1;                                                      // a padding group repeats the last eigenvalue and writes nothing
    const cx<T> lam = lam_all[(long)bm * n + k];
    T eps3 = eps_of<T>::value * hnorm[bm];
    if (!(eps3 > eps_of<T>::safmin)) eps3 = eps_of<T>::safmin;
    cx<T>* Yk = Yall + (long)bm * n * n + k;
    cx<T>* Mk = Mall + (long)bm * n * n + k;
    unsigned char* SWk = SWall + (long)bm * n * n + k;
    cx<T> q[SL], b[SL];
    // column n-1 of M and the start vector
    {
        const cx<T>* src = Ht + (long)(n - 1) * n;
#pragma unroll
        for (int s = 0; s < SL; ++s) {
            const int i = s * LW + L;
            cx<T> v(T(0), T(0)), bv(T(0), T(0));
            if (i < n) { v = src[i]; bv = invit_start<T>(i, k); }
            if (i == n - 1) v = v - lam;
            q[s] = v; b[s] = bv;
        }
    }
    // stage column n-2 (rows 0..n-1); the owner of row n-1 publishes the first pivot
    {
        const int par = (n - 1) & 1;
        for (int a = 0; a < La; ++a) {
            const int c = n - 2 - a;
            if (c >= 0) {
                cx<T>* Pn = slot(c);
                const cx<T>* src = Ht + (long)c * n;
                for (int e = t; e < n; e += NT) Pn[e] = src[e];
            }
        }
        const int so = (n - 1) / LW, Lo = (n - 1) - so * LW;
        if (L == Lo) {
            cx<T> qn(T(0), T(0)), bn(T(0), T(0));
#pragma unroll
            for (int s = 0; s < SL; ++s)
                if (s == so) { qn = q[s]; bn = b[s]; }
            cx<T> sub(T(0), T(0));
            if (n >= 2) sub = Ht[(long)(n - 2) * n + n - 1];
            const int sw = (n >= 2) && (abs1(sub) > abs1(qn));
            cx<T> piv = sw ? sub : qn;
            const cx<T> oth = sw ? qn : sub;
            if (abs1(piv) < eps3) piv = cx<T>(eps3, T(0));
            const cx<T> rp = pivot_recip(piv);
            IvPivot<T> o;
            o.m = oth * rp; o.y = bn * rp; o.swap = sw; o.pad[0] = o.pad[1] = o.pad[2] = 0;
            pv[par * LPW + ll] = o;
            if (valid) {
                Yk[(long)(n - 1) * n] = o.y;
                Mk[(long)(n - 1) * n] = o.m;
                SWk[(long)(n - 1) * n] = (unsigned char)sw;
            }
        }
    }
    if (n == 1) return;
    __syncthreads();
    // The pivot arithmetic of step j (interchange decision, reciprocal, multiplier, y_j) is done ONCE, by the lane that owns row j, at the
    // end of step j+1 -- right after it has updated that row -- and published through LDS; that lane also writes the step's record.
    // Rows 0..j-1 live in the slots 0..(j-1)/LW: the slot loop is fully unrolled (the register arrays are only indexed by constants) with a
    // wave-uniform guard per slot.  The -lam of the diagonal entry of column j-1 (row j-1) only ever matters to the pivot of the next
    // step, so it is applied to the owner's copy of that row instead of to the column.
    // Column prefetch: global memory -> registers -> LDS, D register buffers deep.  Step number k = n-1-j loads column j-1-D into buffer
    // k mod D at its start and, at its end, writes column j-2 (loaded D-1 steps earlier into buffer (k+1) mod D) to LDS for the next step:
    // a load has D-1 steps (each ~1 us: an L2 round trip) to land.  The step loop is unrolled by D so that the buffers are compile-time.
    constexpr int PF = (LW * SL + NT - 1) / NT;
    static_assert(D >= 1 && D <= 3, "prefetch depth");
    cx<T> pf0[PF], pf1[PF], pf2[PF];                                     // named buffers (an array indexed by the step would live in scratch)
    auto fetch = [&](int col, cx<T> (&dst)[PF]) __attribute__((always_inline)) {
        // unconditional loads from clamped (always valid) addresses: a predicated load into zero-initialised registers makes the compiler
        // wait for ALL outstanding loads first (vmcnt(0) in front of the initialisation), which would undo the prefetch
        const int c = col > 0 ? col : 0;                                  // column c: rows 0..c+1
        const cx<T>* srcp = Ht + (long)c * n;
#pragma unroll
        for (int kk = 0; kk < PF; ++kk) {
            const int e = t + NT * kk;
            const cx<T> v = srcp[e <= c + 1 ? e : c + 1];
            dst[kk].x = v.x; dst[kk].y = v.y;                             // member-wise: a struct copy becomes a memcpy that pins the buffer to scratch
        }
    };
    if (D >= 2) fetch(n - 2 - La, pf1);                                  // prologue: the columns the first D-1 steps hand to LDS
    if (D >= 3) fetch(n - 3 - La, pf2);
    auto step = [&](const int j, cx<T> (&ld)[PF], cx<T> (&wr)[PF]) __attribute__((always_inline)) {
        // H[j-1, j-2]: the subdiagonal the next step's pivot competes with.  A wave-uniform address: a scalar load on its own counter
        // (as a vector load it would sit BEHIND this step's prefetch in the in-order vector-memory queue and drag it along)
        cx<T> sub(T(0), T(0));
        if (ring != 3) sub = Ht[(long)(j >= 2 ? j - 2 : 0) * n + (j >= 2 ? j - 1 : 0)];
        fetch(j - D - La, ld);
        const int par = j & 1;
        const int Sact = (j - 1) / LW;
        const cx<T>* P = slot(j - 1);                                   // column j-1 of H: rows 0..j
        const bool owner = L == (j - 1) - Sact * LW;                    // lane that owns row j-1 (slot Sact)
        const IvPivot<T>& pvv = pv[par * LPW + ll];
        const cx<T> m = pvv.m, yj = pvv.y;
        const int swap = __builtin_amdgcn_readfirstlane(pvv.swap);
        if (swap) {
#pragma unroll
            for (int s = 0; s < SL; ++s) {
                if (s <= Sact) {
                    const int i = s * LW + L;
                    const cx<T> pi = P[i < j ? i : j];
                    cx<T> qn = q[s], bn = b[s];
                    cfma(qn, -m, pi);                                   // g' = q - m p
                    cfma(bn, -pi, yj);                                  // b -= p y_j
                    q[s] = qn; b[s] = bn;
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < SL; ++s) {
                if (s <= Sact) {
                    const int i = s * LW + L;
                    cx<T> pi = P[i < j ? i : j];
                    const cx<T> qo = q[s];
                    cx<T> bn = b[s];
                    cfma(pi, -m, qo);                                   // g' = p - m q
                    cfma(bn, -qo, yj);                                  // b -= q y_j
                    q[s] = pi; b[s] = bn;
                }
            }
        }
        if (ring == 1) __syncthreads();                                 // one column buffer: everybody is done reading it
        if (owner) {
            if (ring == 3 && j >= 2) sub = slot(j - 2)[j - 1];
            // pivot of step j-1 (row j-1 is final now); for j = 1 it is the last pivot, which carries the singularity
            cx<T> qn(T(0), T(0)), bn(T(0), T(0));
#pragma unroll
            for (int s = 0; s < SL; ++s)
                if (s == Sact) { qn = q[s]; bn = b[s]; }
            if (swap) { cfma(qn, m, lam); cfma(bn, lam, yj); }          // the column entry was p - lam:  q - m (p - lam),  b - (p - lam) y_j
            else qn = qn - lam;                                         //                                (p - lam) - m q
            const int sw = (j >= 2) && (abs1(sub) > abs1(qn));
            cx<T> piv = sw ? sub : qn;
            const cx<T> oth = sw ? qn : sub;
            if (abs1(piv) < eps3) piv = cx<T>(eps3, T(0));
            const cx<T> rp = pivot_recip(piv);
            IvPivot<T> o;
            o.m = oth * rp; o.y = bn * rp; o.swap = sw; o.pad[0] = o.pad[1] = o.pad[2] = 0;
            pv[(par ^ 1) * LPW + ll] = o;
            if (valid) {
                Yk[(long)(j - 1) * n] = o.y;
                Mk[(long)(j - 1) * n] = o.m;
                SWk[(long)(j - 1) * n] = (unsigned char)sw;
            }
        }
        const int wc = j - 1 - La;                                      // column handed to LDS now: rows 0..wc+1
        if (wc >= 0) {
            cx<T>* Pn = slot(wc);
#pragma unroll
            for (int kk = 0; kk < PF; ++kk) {
                const int e = t + NT * kk;
                if (e <= wc + 1) Pn[e] = cx<T>(wr[kk].x, wr[kk].y);
            }
        }
        __syncthreads();
    };
    // step number k = n-1-j loads into buffer k mod D and writes from buffer (k+1) mod D
    for (int j = n - 1; j >= 1; j -= D) {
        if constexpr (D == 1) {
            step(j, pf0, pf0);
        } else if constexpr (D == 2) {
            step(j, pf0, pf1);
            if (j - 1 >= 1) step(j - 1, pf1, pf0);
        } else {
            step(j, pf0, pf1);
            if (j - 1 >= 1) step(j - 1, pf1, pf2);
            if (j - 2 >= 1) step(j - 2, pf2, pf0);
        }
    }
}

// ---- the same elimination with the columns of H streamed by direct global -> LDS loads ---------------------------------------------
// Measured on MI355X (n = 1922, batch 128): with the columns prefetched through registers a step took 2.2 us whatever the depth -- the
// compiler makes every loop iteration wait for all outstanding loads (a load destination is reused as an address register), so a step
// pays a full L2 / HBM round trip.  Here a ring of RING columns lives in LDS; at the start of step j every wave issues its share of
// column j-RING as global_load_lds (no destination registers, not tracked by the compiler) and, before the step's barrier, waits only
// until the columns of the NEXT step have landed (vmcnt counts the younger loads that may stay in flight): a column has RING-2 steps
// to arrive.  The step's record (y_j, multiplier, interchange flag) goes to an LDS log that the workgroup flushes every IVF steps, so the
// step loop issues no other vector-memory operation that would disturb the count.
constexpr int IVF = 32;

template <int K> __device__ __forceinline__ void iv_wait_vmcnt() {
    static_assert(K == 0 || K == 1 || K == 2 || K == 4 || K == 8 || K == 16, "wait count");
    if constexpr (K == 0) TRX_WAIT_VMCNT(0);
    else if constexpr (K == 1) TRX_WAIT_VMCNT(1);
    else if constexpr (K == 2) TRX_WAIT_VMCNT(2);
    else if constexpr (K == 4) TRX_WAIT_VMCNT(4);
    else if constexpr (K == 8) TRX_WAIT_VMCNT(8);
    else TRX_WAIT_VMCNT(16);
}

template <class T, int WPL, int SL, int NT, int RING>
__global__ __launch_bounds__(NT) void invit_solve_dma_kernel(const cx<T>* __restrict__ Ht_all, int n, const cx<T>* __restrict__ lam_all, const T* __restrict__ hnorm,
                                                             cx<T>* __restrict__ Yall, cx<T>* __restrict__ Mall, unsigned char* __restrict__ SWall,
                                                             int batch, int mb, int MB, int gb, int dbg) {
    TRX_DYN_SMEM(smem);
    static_assert(RING == 3 || RING == 4, "ring depth");
    static_assert(sizeof(cx<T>) == 16, "16-byte elements: one direct-to-LDS load per lane and element");
    constexpr int NW = NT / 64, LW = 64 * WPL, LPW = NW / WPL;
    constexpr int PFW = (LW * SL + NT - 1) / NT;                               // 64-element chunks of a column per wave
    constexpr int CS = PFW * NT;                                                // ring slot, in elements (>= n)
    cx<T>* ring = reinterpret_cast<cx<T>*>(smem);                              // [RING][CS]
    IvPivot<T>* pv = reinterpret_cast<IvPivot<T>*>(ring + (size_t)RING * CS);  // [2][LPW]
    cx<T>* ylog = reinterpret_cast<cx<T>*>(pv + 2 * LPW);                      // [2][IVF][LPW]
    cx<T>* mlog = ylog + 2 * IVF * LPW;                                         // [2][IVF][LPW]
    int* swlog = reinterpret_cast<int*>(mlog + 2 * IVF * LPW);                  // [2][IVF][LPW]
    // XCD-aware launch geometry (MB > 0): a 1-D grid whose workgroup w runs on XCD w mod 8 (round-robin dispatch).  All workgroups of an
    // XCD take eigenvalue groups of the SAME matrix and start together, so they walk down its columns in step and the XCD's L2 serves each
    // column to all of them -- measured: with one workgroup per (group, matrix) in dispatch order the kernel moved 3.6 TB through the
    // fabric per 128-matrix batch (every workgroup streams its matrix by itself) and ran at that limit whatever the inner loop looked like.
    int bm, grp;
    if (MB > 0) {
        const int x = blockIdx.x & 7, sl = blockIdx.x >> 3;
        bm = mb + x % MB;
        grp = gb + sl + (int)(gridDim.x >> 3) * (x / MB);
    } else {
        bm = blockIdx.y; grp = blockIdx.x;
    }
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    if (bm >= batch || grp * LPW >= n) return;
    const cx<T>* Ht = Ht_all + (long)bm * n * n;
    const int ll = t / LW, L = t - ll * LW;
    int k = grp * LPW + ll;
    const bool valid = k < n;
    if (!valid) k = n - 1;
    const cx<T> lam = lam_all[(long)bm * n + k];
    T eps3 = eps_of<T>::value * hnorm[bm];
    if (!(eps3 > eps_of<T>::safmin)) eps3 = eps_of<T>::safmin;
    auto slot = [&](int c) __attribute__((always_inline)) { return ring + (c % RING) * CS; };
    // this wave's share of column c (a row of Ht: entries beyond c+1 are exact zeros, so the whole row may be loaded)
    auto issue = [&](int c) __attribute__((always_inline)) {
        const cx<T>* src = Ht + (long)c * n;
        cx<T>* dst = slot(c);
#pragma unroll
        for (int kk = 0; kk < PFW; ++kk) {
            const int ch = wave + NW * kk, e = ch * 64 + lane;
            TRX_LDS_DMA16(src + (e < n ? e : n - 1), dst + ch * 64);
        }
    };
    cx<T> q[SL], b[SL];
    {
        const cx<T>* src = Ht + (long)(n - 1) * n;
#pragma unroll
        for (int s = 0; s < SL; ++s) {
            const int i = s * LW + L;
            cx<T> v(T(0), T(0)), bv(T(0), T(0));
            if (i < n) { v = src[i]; bv = invit_start<T>(i, k); }
            if (i == n - 1) v = v - lam;
            q[s] = v; b[s] = bv;
        }
    }
    for (int a = 2; a <= RING; ++a)
        if (n - a >= 0) issue(n - a);                                          // columns n-2 .. n-RING
    // first pivot (row n-1): log entry 0
    {
        const int so = (n - 1) / LW, Lo = (n - 1) - so * LW;
        if (L == Lo) {
            cx<T> qn(T(0), T(0)), bn(T(0), T(0));
#pragma unroll
            for (int s = 0; s < SL; ++s)
                if (s == so) { qn = q[s]; bn = b[s]; }
            cx<T> sub(T(0), T(0));
            if (n >= 2) sub = Ht[(long)(n - 2) * n + n - 1];
            const int sw = (n >= 2) && (abs1(sub) > abs1(qn));
            cx<T> piv = sw ? sub : qn;
            const cx<T> oth = sw ? qn : sub;
            if (abs1(piv) < eps3) piv = cx<T>(eps3, T(0));
            const cx<T> rp = pivot_recip(piv);
            IvPivot<T> o;
            o.m = oth * rp; o.y = bn * rp; o.swap = sw; o.pad[0] = o.pad[1] = o.pad[2] = 0;
            pv[((n - 1) & 1) * LPW + ll] = o;
            ylog[ll] = o.y; mlog[ll] = o.m; swlog[ll] = sw;
        }
    }
    iv_wait_vmcnt<0>();
    __syncthreads();
    // log entry f of block blk is row n-1-(blk*IVF+f); a block is flushed by the first IVF*LPW threads once it is complete
    auto flush = [&](int blk, int cnt) __attribute__((always_inline)) {
        if (t < IVF * LPW) {
            const int f = t / LPW, l2 = t - f * LPW;
            const int kk = grp * LPW + l2, r = n - 1 - (blk * IVF + f);
            if (f < cnt && kk < n) {
                const int o = ((blk & 1) * IVF + f) * LPW + l2;
                const long g = (long)bm * n * n + (long)r * n + kk;
                Yall[g] = ylog[o]; Mall[g] = mlog[o]; SWall[g] = (unsigned char)swlog[o];
            }
        }
    };
    // Order inside a step (it is latency, not arithmetic, that bounds it: 2-4 waves per SIMD, one barrier per step): all LDS reads of the
    // step are issued first; the slots are then processed from the highest active one DOWN, because that is where the pivot row j-1 lives:
    // its owner computes and publishes the next pivot (a reciprocal: the longest dependent chain of the step) right after that slot, while
    // every wave, its own included, still has the lower slots to do.
    for (int j = n - 1; j >= 1; --j) {
        const int par = j & 1;
        const int Sact = (j - 1) / LW;
        if (j - RING >= 0 && !(dbg & 4)) issue(j - RING);
        const cx<T>* P = slot(j - 1);                                   // column j-1 of H: rows 0..j
        const bool owner = L == (j - 1) - Sact * LW && !(dbg & 2);      // lane that owns row j-1 (slot Sact)
        const IvPivot<T>& pvv = pv[par * LPW + ll];
        const cx<T> m = pvv.m, yj = pvv.y;
        const int swap = __builtin_amdgcn_readfirstlane(pvv.swap);
        cx<T> sub(T(0), T(0));
        if (owner && j >= 2) sub = slot(j - 2)[j - 1];                  // H[j-1, j-2] (that column landed a step ago)
        cx<T> pis[SL];
        if (dbg & 1) { if (!(dbg & 8)) __syncthreads(); continue; }
#pragma unroll
        for (int s = SL - 1; s >= 0; --s) {
            pis[s] = cx<T>(T(0), T(0));
            if (s <= Sact) { const int i = s * LW + L; pis[s] = P[i < j ? i : j]; }
        }
        __builtin_amdgcn_sched_barrier(0);
        const int rec = n - j;                                          // number of the record written in this step (row j-1)
        // new carry = alpha q + beta p,  b -= gam q + del p   with (alpha, beta, gam, del) = (-m, 1, y_j, 0) or, interchanged, (1, -m, 0, y_j):
        // one branch-free body for both cases (the extra products with exact 0 / 1 cost less than two code paths and their register moves)
        const cx<T> one(T(1), T(0)), zero(T(0), T(0));
        const cx<T> alpha = swap ? one : -m, beta = swap ? -m : one, gam = swap ? zero : yj, del = swap ? yj : zero;
#pragma unroll
        for (int s = SL - 1; s >= 0; --s) {
            if (s <= Sact) {
                const cx<T> pi = pis[s], qo = q[s];
                cx<T> qn = alpha * qo, bn = b[s];
                cfma(qn, beta, pi);
                cfma(bn, -gam, qo);
                cfma(bn, -del, pi);
                q[s] = qn; b[s] = bn;
                if (s == Sact && owner) {
                    // pivot of step j-1 from row j-1, which is final now (the column entry was p - lam: applied to this copy only)
                    cfma(qn, -beta, lam);
                    cfma(bn, del, lam);
                    const int sw = (j >= 2) && (abs1(sub) > abs1(qn));
                    cx<T> piv = sw ? sub : qn;
                    const cx<T> oth = sw ? qn : sub;
                    if (abs1(piv) < eps3) piv = cx<T>(eps3, T(0));
                    const cx<T> rp = pivot_recip(piv);
                    IvPivot<T> o;
                    o.m = oth * rp; o.y = bn * rp; o.swap = sw; o.pad[0] = o.pad[1] = o.pad[2] = 0;
                    pv[(par ^ 1) * LPW + ll] = o;
                    const int lo = (((rec / IVF) & 1) * IVF + rec % IVF) * LPW + ll;
                    ylog[lo] = o.y; mlog[lo] = o.m; swlog[lo] = sw;
                }
            }
        }
        // the next step reads column j-2 and, for its pivot, column j-3: everything older than this step's loads (RING 4) has to be in
        iv_wait_vmcnt<(RING == 4 ? PFW : 0)>();
        if (!(dbg & 8)) __syncthreads();
        if (rec % IVF == IVF - 1) flush(rec / IVF, IVF);
    }
    if (n % IVF != 0) flush((n - 1) / IVF, n % IVF);                    // records 0 .. n-1: the last, partial block
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

template <class T, int WPL, int SL, int NT, int D>
int launch_solve(hipStream_t s, const cx<T>* Ht, int n, const cx<T>* w, const T* hnorm, cx<T>* Y, cx<T>* M, unsigned char* SW, int batch) {
    if constexpr (WPL * 64 > NT) {
        return TRX_ERR_ARG;
    } else {
        constexpr int LPW = (NT / 64) / WPL;
        // double-buffered column staging when two columns fit into LDS next to the pivot slots, else one buffer and a second barrier per step
        // as many columns of H in LDS as fit (3, 2 or 1), next to the pivot slots
        const size_t col = sizeof(cx<T>) * (size_t)n, fix = sizeof(IvPivot<T>) * 2 * LPW;
        const int ring = (g_invit_ring >= 1 && g_invit_ring <= 3 && g_invit_ring * col + fix <= 150 * 1024) ? g_invit_ring
                         : (3 * col + fix <= 150 * 1024 ? 3 : (2 * col + fix <= 150 * 1024 ? 2 : 1));
        const size_t sm = ring * col + fix;
        if (set_max_dyn_smem((const void*)invit_solve_kernel<T, WPL, SL, NT, D>, sm)) return TRX_ERR_LAUNCH;
        ProfScope prof(PROF_INVIT, s, 8.0 * (double)n * n * n * batch, 0.0);
        TRX_LAUNCH((invit_solve_kernel<T, WPL, SL, NT, D>), dim3(cdiv_i(n, LPW), batch), dim3(NT), sm, s, Ht, n, w, hnorm, Y, M, SW, ring);
        return TRX_OK;
    }
}

template <class T, int WPL, int SL, int NT, int RING>
int launch_solve_dma(hipStream_t s, const cx<T>* Ht, int n, const cx<T>* w, const T* hnorm, cx<T>* Y, cx<T>* M, unsigned char* SW, int batch) {
    if constexpr (WPL * 64 > NT) {
        return TRX_ERR_ARG;
    } else {
        constexpr int NW = NT / 64, LW = 64 * WPL, LPW = NW / WPL, PFW = (LW * SL + NT - 1) / NT, CS = PFW * NT;
        const size_t sm = sizeof(cx<T>) * (size_t)RING * CS + sizeof(IvPivot<T>) * 2 * LPW + (2 * sizeof(cx<T>) + sizeof(int)) * 2 * IVF * LPW;
        if (set_max_dyn_smem((const void*)invit_solve_dma_kernel<T, WPL, SL, NT, RING>, sm)) return TRX_ERR_LAUNCH;
        ProfScope prof(PROF_INVIT, s, 8.0 * (double)n * n * n * batch, 0.0);
        const int groups = cdiv_i(n, LPW);
        if (g_invit_xcd == 1) {
            TRX_LAUNCH((invit_solve_dma_kernel<T, WPL, SL, NT, RING>), dim3(groups, batch), dim3(NT), sm, s, Ht, n, w, hnorm, Y, M, SW, batch, 0, 0, 0, g_invit_dbg);
            return TRX_OK;
        }
        // one workgroup per CU and launch: 8 XCDs x (CUs / 8) slots; MB matrices side by side, 8 / MB XCDs per matrix
        int dev = 0, cus = 256;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 8) cus = 256;
        const int slots = cus / 8;
        const int MB = batch >= 8 ? 8 : (batch >= 4 ? 4 : (batch >= 2 ? 2 : 1));
        const int gstep = slots * (8 / MB);
        for (int mb = 0; mb < batch; mb += MB)
            for (int gb = 0; gb < groups; gb += gstep)
                TRX_LAUNCH((invit_solve_dma_kernel<T, WPL, SL, NT, RING>), dim3(8 * slots), dim3(NT), sm, s, Ht, n, w, hnorm, Y, M, SW, batch, mb, MB, gb, g_invit_dbg);
        return TRX_OK;
    }
}
// LDS bytes of the direct-to-LDS variant for waves-per-eigenvalue wpl
template <class T, int SL, int NT>
size_t dma_lds_bytes(int wpl, int ringn) {
    const int NW = NT / 64, LW = 64 * wpl, LPW = NW / wpl, PFW = (LW * SL + NT - 1) / NT, CS = PFW * NT;
    return sizeof(cx<T>) * (size_t)ringn * CS + sizeof(IvPivot<T>) * 2 * LPW + (2 * sizeof(cx<T>) + sizeof(int)) * 2 * IVF * LPW;
}
template <class T, int SL, int NT>
int dispatch_solve_dma(hipStream_t s, const cx<T>* Ht, int n, const cx<T>* w, const T* hnorm, cx<T>* Y, cx<T>* M, unsigned char* SW, int batch, int min_wpl, int ringn) {
    const int rows1 = 64 * SL;
    int wpl = 1;
    while (wpl * rows1 < n || wpl < min_wpl) wpl *= 2;
    if (wpl * 64 > NT || dma_lds_bytes<T, SL, NT>(wpl, ringn) > 156 * 1024) return -1;          // does not fit: the caller falls back
#define TRX_IV_DMA(W) case W: return ringn == 4 ? launch_solve_dma<T, W, SL, NT, 4>(s, Ht, n, w, hnorm, Y, M, SW, batch) : launch_solve_dma<T, W, SL, NT, 3>(s, Ht, n, w, hnorm, Y, M, SW, batch);
    switch (wpl) {
        TRX_IV_DMA(1) TRX_IV_DMA(2) TRX_IV_DMA(4) TRX_IV_DMA(8) TRX_IV_DMA(16)
        default: return -1;
    }
#undef TRX_IV_DMA
}

// waves per eigenvalue: the smallest power of two whose lanes x slots cover n
template <class T, int SL, int NT, int D>
int dispatch_solve(hipStream_t s, const cx<T>* Ht, int n, const cx<T>* w, const T* hnorm, cx<T>* Y, cx<T>* M, unsigned char* SW, int batch, int min_wpl) {
    const int rows1 = 64 * SL;
    int wpl = 1;
    while (wpl * rows1 < n || wpl < min_wpl) wpl *= 2;
    switch (wpl) {
        case 1: return launch_solve<T, 1, SL, NT, D>(s, Ht, n, w, hnorm, Y, M, SW, batch);
        case 2: return launch_solve<T, 2, SL, NT, D>(s, Ht, n, w, hnorm, Y, M, SW, batch);
        case 4: return launch_solve<T, 4, SL, NT, D>(s, Ht, n, w, hnorm, Y, M, SW, batch);
        case 8: return launch_solve<T, 8, SL, NT, D>(s, Ht, n, w, hnorm, Y, M, SW, batch);
        case 16: return launch_solve<T, 16, SL, NT, D>(s, Ht, n, w, hnorm, Y, M, SW, batch);
        default: return TRX_ERR_ARG;
    }
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
    // layout (knob invit_cfg): the two vectors of an eigenvalue take 2 x SL complex numbers per lane (SL = 8 in fp64, 16 in fp32: 64 VGPRs);
    //   0 / 1: 512 threads (<= 256 VGPRs), column prefetch 2 steps deep   2: 3 steps deep   3: 1 step deep   4: 1024 threads (<= 128 VGPRs), 2 deep
    const int cfg = g_invit_cfg, mw = g_invit_min_wpl;
    constexpr int SLT = sizeof(T) == 8 ? 8 : 16;
    rc = -1;
    if constexpr (sizeof(cx<T>) == 16) {
        // direct-to-LDS column ring (fp64; an fp32 element is 8 bytes, half a lane's load): 5 = 1024 threads, 6 = 512 threads, ring 4 if it fits
        if (cfg == 5 || cfg == 6) {
            const int ringn = g_invit_ring == 3 ? 3 : 4;
            rc = cfg == 5 ? dispatch_solve_dma<T, SLT, 1024>(s, B.Ht, n, w, B.hnorm, Y, M, B.SW, batch, mw, ringn)
                          : dispatch_solve_dma<T, SLT, 512>(s, B.Ht, n, w, B.hnorm, Y, M, B.SW, batch, mw, ringn);
            if (rc == -1 && ringn == 4)
                rc = cfg == 5 ? dispatch_solve_dma<T, SLT, 1024>(s, B.Ht, n, w, B.hnorm, Y, M, B.SW, batch, mw, 3)
                              : dispatch_solve_dma<T, SLT, 512>(s, B.Ht, n, w, B.hnorm, Y, M, B.SW, batch, mw, 3);
        }
    }
    const bool need1024 = n > 8 * 64 * SLT;                            // more rows than eight waves of slots hold
    if (rc != -1) { /* launched (or failed) above */ }
    else if (need1024) rc = dispatch_solve<T, SLT, 1024, 2>(s, B.Ht, n, w, B.hnorm, Y, M, B.SW, batch, mw);
    else if (cfg == 2) rc = dispatch_solve<T, SLT, 512, 3>(s, B.Ht, n, w, B.hnorm, Y, M, B.SW, batch, mw);
    else if (cfg == 3) rc = dispatch_solve<T, SLT, 512, 1>(s, B.Ht, n, w, B.hnorm, Y, M, B.SW, batch, mw);
    else if (cfg == 4) rc = dispatch_solve<T, SLT, 1024, 2>(s, B.Ht, n, w, B.hnorm, Y, M, B.SW, batch, mw);
    else rc = dispatch_solve<T, SLT, 512, 2>(s, B.Ht, n, w, B.hnorm, Y, M, B.SW, batch, mw);
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
