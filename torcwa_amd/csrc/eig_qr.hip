// Batched small-bulge multi-shift QR iteration on upper Hessenberg matrices, H = Z T Z^H (Schur form), with the
// unitary accumulated into Z.  Second stage of the replacement for torch.linalg.eig (torcwa/torch_eig.py:14).
//
// Design for MI355X.  The sequential part of the QR algorithm (generating rotations) only ever touches a narrow
// diagonal window, so it runs as ONE workgroup per matrix entirely out of LDS (qr_window_kernel: a chain of up to
// QNS single-shift bulges, 2 rows apart, is chased through a QW x QW window while the window's unitary U is
// accumulated in LDS).  Everything off the window is updated afterwards with U by wide, embarrassingly parallel
// slab kernels (apply_left / apply_right) that stream H and Z through LDS once per window step.  Per-matrix
// progress (active block, shifts, chase position) lives in device memory, so one fixed launch schedule serves the
// whole batch; matrices that have nothing to do in a step see an empty window and exit.
//
//   qr_prepare_kernel  (1 wave / matrix)  deflation scan, active-block bookkeeping, aggressive early deflation on the
//                                         trailing window (in-LDS single-shift QR), shifts for up to QKC chains; blocks
//                                         <= QNMIN are finished here by the same in-LDS QR with U accumulated.
//   qr_window_kernel   (1 workgroup per matrix AND chain) one window step of each bulge chain.
//   apply_window_kernel<PART 0>  H[w0:w1, w1:n] <- U^H H[w0:w1, w1:n]                                   (left updates, all chains)
//   apply_window_kernel<PART 1>  H[0:w0, w0:w1] <- H[0:w0, w0:w1] U ;  Z[:, w0:w1] <- Z[:, w0:w1] U     (right updates, all chains)
//
// Several bulge chains per sweep.  A sweep sends up to QKC chains of QNS shifts each down the active block, chain c following
// chain c-1 at a distance of at least one window (a chain moves only if its new window ends above the last bulge of the chain
// ahead), so that one window step advances all of them: their windows are disjoint diagonal blocks (independent workgroups), the
// left updates of different chains touch disjoint rows and the right updates disjoint columns.  A block that two chains both
// reach (rows of the upper window x columns of the lower one) gets U_a^H from the left in PART 0 and U_b from the right in PART 1
// -- the two commute, and the two launches order them.  Up to 48 of the AED window's eigenvalues are thus used per sweep
// instead of 16: a third of the AED calls and of the latency-bound window steps for the same number of shifts.
#include "eig.hpp"
#include <cstdlib>
#include <limits>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "mfma.hpp"
#include "prof.hpp"

namespace trx {
namespace {

constexpr int QW = EigPlan::QW, QNS = EigPlan::QNS, QKC = EigPlan::QKC;
constexpr int SM = 64;             // largest matrix the in-LDS single-wave routines handle (one lane per column)
// The small in-LDS matrices have the RUNTIME leading dimension SLD = sm + 1, sm = the largest size the launch will meet (AED
// window / small-block threshold): at the default 48 the prepare kernel needs 79 KB of LDS instead of 135 KB, which lets a
// slab-update workgroup (74 KB) share its CU -- see the note on co-residency at hessenberg_qr.
constexpr int QAED = EigPlan::QAED;   // aggressive-early-deflation window
constexpr int QAED_MOVES = 12;        // undeflatable eigenvalues moved out of the way per AED

template <class T>
struct Rot {
    T c;
    cx<T> s, r;
};

// Rotations G = [[c, s], [-conj(s), c]] (c real) with G [f; g] = [r; 0]; see rotg_fast below.
// rows:  (x, y) <- (c x + s y, -conj(s) x + c y)
template <class T>
__device__ __forceinline__ void rot_rows(const Rot<T>& R, cx<T>& x, cx<T>& y) {
    const cx<T> nx = R.c * x + R.s * y;
    const cx<T> ny = R.c * y - conj(R.s) * x;
    x = nx; y = ny;
}
// columns (right-multiplication by G^H):  (x, y) <- (c x + conj(s) y, -s x + c y)
template <class T>
__device__ __forceinline__ void rot_cols(const Rot<T>& R, cx<T>& x, cx<T>& y) {
    const cx<T> nx = R.c * x + conj(R.s) * y;
    const cx<T> ny = R.c * y - R.s * x;
    x = nx; y = ny;
}

// 1 / sqrt(x) for the rotation generator.  rsqrt(float) resolves to the DOUBLE overload in HIP: the fp32 kernels of round 3 carried, per
// call, v_cvt_f64_f32 + v_rsq_f64 + a five-instruction fp64 refinement + v_cvt_f32_f64 -- twice per rotation, on the critical path of every
// chain step.  A plain v_rsq_f32 (1 ulp, or refined in fp32) is cheaper but leaves the rotations unitary to only ~5 ulp, and the
// eigen-refinement behind the fp32 solver feels that (exactly degenerate spectra: 1e-8 instead of 1e-14 after two Newton steps, emulator).
// So fp32 takes ONE unrefined v_rsq_f64 of the fp64 argument: good to ~2^-26, i.e. correctly rounded for a float, no under / overflow of
// the product |f|^2 d^2, five dependent instructions.
__device__ __forceinline__ double fast_rsqrt(double x) { return rsqrt(x); }
__device__ __forceinline__ float fast_rsqrt_f64arg(double x) { return (float)__builtin_amdgcn_rsq(x); }

// Fast rotation generator for the chase (no hypot/divide chain: two rsqrt).  |f|^2+|g|^2 cannot overflow here: the
// entries of a balanced RCWA operator are O(1e3) and negligible subdiagonals were flushed to zero by the deflation scan.
template <class T>
__device__ __forceinline__ Rot<T> rotg_fast(cx<T> f, cx<T> g) {
    Rot<T> R;
    const T ag2 = norm2(g), af2 = norm2(f);
    if (ag2 == T(0)) { R.c = T(1); R.s = cx<T>(T(0), T(0)); R.r = f; return R; }
    if (af2 == T(0)) { const T ig = (sizeof(T) == 8) ? (T)fast_rsqrt((double)ag2) : (T)fast_rsqrt_f64arg((double)ag2); R.c = T(0); R.s = ig * conj(g); R.r = cx<T>(ag2 * ig, T(0)); return R; }
    const T n2 = af2 + ag2;
    T tu;                                          // 1 / (|f| d)
    if (sizeof(T) == 8) {
        tu = (T)fast_rsqrt((double)(af2 * n2));    // one rsqrt on the critical path (fp64 range: |f|^2 d^2 cannot under/overflow here)
    } else {
        tu = (T)fast_rsqrt_f64arg((double)af2 * (double)n2);      // fp32: the product is formed in fp64 (it could underflow in fp32)
    }
    R.c = af2 * tu;                                // |f| / d
    R.s = tu * (f * conj(g));                      // (f/|f|) conj(g) / d
    R.r = (n2 * tu) * f;                           // (f/|f|) d
    return R;
}

// Synchronisation of the single-wave in-LDS routines below.  In a 64-thread workgroup they use the block barrier; when the same code
// runs on wave 0 of a larger workgroup (multi-wave AED kernel) a block barrier would wait for the other waves, and none is needed: one
// wave's LDS operations execute in program order, so only the compiler has to be kept from reordering them (wave_sync).
struct BlockSync { __device__ __forceinline__ void operator()() const { __syncthreads(); } };
struct WaveSync { __device__ __forceinline__ void operator()() const { wave_sync(); } };

// Schur form of an m x m (m <= SM = 64) upper Hessenberg matrix held in LDS, by EXPLICITLY shifted QR iterations executed
// by ONE wave.  Lane c owns column c while Q^H is applied from the left (the rotation that zeroes H[r+1,r] is generated by
// lane r as soon as its column has received the previous rotations, and broadcast through LDS), and row c while Q is
// applied from the right (all rotations are known by then, so the lanes run independently).  Compared with a rotation-by-
// rotation implicit chase this keeps every lane busy and needs no block barrier inside a QR iteration.
// On return Hs is upper triangular; if Us != nullptr it holds U with H_in = U T U^H.  Returns false if not converged.
// ihi0 / lstop: work on the leading (ihi0 + 1) x (ihi0 + 1) part only and stop as soon as everything below row lstop has deflated
// (defaults: the whole matrix); *ihi_out receives the index the loop stopped at.
// BC: the inputs (f, g) of rotation r are broadcast from lane r with v_readlane and EVERY lane generates the rotation, instead of lane r
// generating it and five ds_bpermute shuffles distributing the result (one LDS round trip less on the chain of every rotation).
template <class T, class SY = BlockSync, bool BC = false>
__device__ bool small_schur(cx<T>* Hs, int m, cx<T>* Us, Rot<T>* rots, const int SLD, long long* dbg = nullptr, int ihi0 = -1, int lstop = 0, int* ihi_out = nullptr) {
    const SY sync;
    const int lane = threadIdx.x;
    long long t_left = 0, t_right = 0, t_u = 0, t_pre = 0, tt = 0, n_it = 0, n_rot = 0;
    if (dbg) tt = clock64();
    const T ulp = eps_of<T>::value;
    int ihi = ihi0 >= 0 ? ihi0 : m - 1, its = 0, total = 0;
    while (ihi > lstop) {
        // deflation: flush negligible subdiagonals of [1, ihi] to zero, find the active block [l, ihi]
        int small = 0;
        if (lane >= 1 && lane <= ihi) {
            const cx<T> sub = Hs[lane * SLD + lane - 1];
            T sc = abs1(Hs[(lane - 1) * SLD + lane - 1]) + abs1(Hs[lane * SLD + lane]);
            if (sc == T(0)) sc = T(1);
            small = (abs1(sub) <= ulp * sc);
            if (small) Hs[lane * SLD + lane - 1] = cx<T>(T(0), T(0));
        }
        const unsigned long long mask = __ballot(small);
        const int l = mask ? (63 - __builtin_clzll(mask)) : 0;
        sync();
        if (l == ihi) { --ihi; its = 0; continue; }
        ++its; ++total;
        if (total > 40 * m) { if (ihi_out) *ihi_out = ihi; return false; }
        cx<T> sig;
        {
            const cx<T> a = Hs[(ihi - 1) * SLD + ihi - 1], bq = Hs[(ihi - 1) * SLD + ihi], cq = Hs[ihi * SLD + ihi - 1], d = Hs[ihi * SLD + ihi];
            if (its % 10 == 0) {
                sig = d + cx<T>(T(0.75) * fabs(cq.x), T(0));
            } else {
                const cx<T> tr = T(0.5) * (a + d);
                const cx<T> det = (a - tr) * (d - tr) - bq * cq;
                const cx<T> sq = csqrt(-det);
                const cx<T> e1 = tr + sq, e2 = tr - sq;
                sig = (abs1(e1 - d) < abs1(e2 - d)) ? e1 : e2;
            }
        }
        sync();
        if (lane >= l && lane <= ihi) Hs[lane * SLD + lane] -= sig;               // H_act - sig I
        sync();
        if (dbg) { const long long t1 = clock64(); t_pre += t1 - tt; tt = t1; n_it += 1; n_rot += ihi - l; }
        // left phase: (H_act - sig I) = Q R, rows l..ihi of all columns >= l.  Lane c walks down its column with the
        // running row value carried in registers (one LDS read + one write per rotation, next row prefetched); the
        // rotation generated by lane r is broadcast with wave shuffles instead of an LDS round trip.
        {
            const bool mine = (lane >= l && lane < m);
            cx<T> xcur = mine ? Hs[l * SLD + lane] : cx<T>(T(0), T(0));
            cx<T> ynext = (mine && l + 1 <= ihi) ? Hs[(l + 1) * SLD + lane] : cx<T>(T(0), T(0));
            for (int r = l; r < ihi; ++r) {
                const cx<T> ypre = (mine && r + 2 <= ihi) ? Hs[(r + 2) * SLD + lane] : cx<T>(T(0), T(0));   // prefetch
                Rot<T> R;
                if constexpr (BC) {
                    R = rotg_fast(bcast_lane(xcur, r), bcast_lane(ynext, r));
                } else {
                    R.c = T(1); R.s = cx<T>(T(0), T(0)); R.r = cx<T>(T(0), T(0));
                    if (lane == r) R = rotg_fast(xcur, ynext);
                    R.c = __shfl(R.c, r);
                    R.s.x = __shfl(R.s.x, r); R.s.y = __shfl(R.s.y, r);
                    R.r.x = __shfl(R.r.x, r); R.r.y = __shfl(R.r.y, r);
                }
                if (lane == r) rots[r] = R;
                if (lane >= r && lane < m) {
                    cx<T> x = xcur, y = ynext;
                    rot_rows(R, x, y);
                    if (lane == r) { x = R.r; y = cx<T>(T(0), T(0)); Hs[(r + 1) * SLD + lane] = y; }
                    Hs[r * SLD + lane] = x;
                    xcur = y;
                }
                ynext = ypre;
            }
            if (lane >= ihi && lane < m) Hs[ihi * SLD + lane] = xcur;
        }
        sync();
        if (dbg) { const long long t1 = clock64(); t_left += t1 - tt; tt = t1; }
        // right phase: H <- R Q (row i is touched by the rotations r >= i-1), U <- U Q (all rows); each lane walks along
        // its own row with the running column value in registers
        {
        if (lane <= ihi) {
            const int i = lane;
            const int r0 = (i - 1 > l) ? i - 1 : l;
            if (r0 < ihi) {
                cx<T> x = Hs[i * SLD + r0];
                for (int r = r0; r < ihi; ++r) {
                    cx<T> y = Hs[i * SLD + r + 1];
                    rot_cols(rots[r], x, y);
                    Hs[i * SLD + r] = x;
                    x = y;
                }
                Hs[i * SLD + ihi] = x;
            }
        }
        if (dbg) { sync(); const long long t1 = clock64(); t_right += t1 - tt; tt = t1; }
        if (Us && lane < m && l < ihi) {
            cx<T> x = Us[lane * SLD + l];
            for (int r = l; r < ihi; ++r) {
                cx<T> y = Us[lane * SLD + r + 1];
                rot_cols(rots[r], x, y);
                Us[lane * SLD + r] = x;
                x = y;
            }
            Us[lane * SLD + ihi] = x;
        }
        sync();
        if (dbg) { const long long t1 = clock64(); t_u += t1 - tt; tt = t1; }
        }
        if (lane >= l && lane <= ihi) Hs[lane * SLD + lane] += sig;
        sync();
    }
    if (dbg && lane == 0) { dbg[0] += t_pre; dbg[1] += t_left; dbg[2] += t_right; dbg[3] += t_u; dbg[4] += n_it; dbg[5] += n_rot; }
    if (ihi_out) *ihi_out = ihi;
    return true;
}

// Swap the adjacent diagonal entries k, k+1 of the upper-triangular Ts (order m <= 64) by one rotation, accumulating into
// Vs (LAPACK ztrexc for complex Schur forms).  One wave.
template <class T, class SY = BlockSync>
__device__ void schur_swap(cx<T>* Ts, cx<T>* Vs, int m, int k, const int SLD) {
    const SY sync;
    const int lane = threadIdx.x;
    const cx<T> a = Ts[k * SLD + k], bq = Ts[(k + 1) * SLD + k + 1], x = Ts[k * SLD + k + 1];
    const Rot<T> R = rotg_fast(x, bq - a);
    sync();
    if (lane >= k && lane < m) {                                   // rows k, k+1: lane = column
        cx<T> u = Ts[k * SLD + lane], v = Ts[(k + 1) * SLD + lane];
        rot_rows(R, u, v);
        Ts[k * SLD + lane] = u; Ts[(k + 1) * SLD + lane] = v;
    }
    if (lane < m) {                                                 // V columns k, k+1: lane = row
        cx<T> u = Vs[lane * SLD + k], v = Vs[lane * SLD + k + 1];
        rot_cols(R, u, v);
        Vs[lane * SLD + k] = u; Vs[lane * SLD + k + 1] = v;
    }
    sync();
    if (lane <= k + 1) {                                            // columns k, k+1 of T: lane = row
        cx<T> u = Ts[lane * SLD + k], v = Ts[lane * SLD + k + 1];
        rot_cols(R, u, v);
        if (lane == k + 1) u = cx<T>(T(0), T(0));
        Ts[lane * SLD + k] = u; Ts[lane * SLD + k + 1] = v;
    }
    sync();
}

// Householder reflector for x[0:len] held in LDS at stride `inc` (LAPACK zlarfg): H = I - tau v v^H, H^H x = beta e1.
// All lanes compute redundantly; v (v[0] = 1) is written to vw[0:len] by the lanes; returns tau, beta.  len <= 64.
template <class T, class SY = BlockSync>
__device__ void small_larfg(const cx<T>* x, int inc, int len, cx<T>* vw, cx<T>& tau, T& beta) {
    const SY sync;
    const int lane = threadIdx.x;
    const cx<T> alpha = x[0];
    T xn2 = (lane >= 1 && lane < len) ? norm2(x[lane * inc]) : T(0);
    xn2 = wave_sum(xn2);
    cx<T> scale;
    if (xn2 == T(0) && alpha.y == T(0)) {
        tau = cx<T>(T(0), T(0)); beta = alpha.x; scale = cx<T>(T(0), T(0));
    } else {
        const T nrm = sqrt(norm2(alpha) + xn2);
        beta = (alpha.x >= T(0)) ? -nrm : nrm;
        tau = cx<T>((beta - alpha.x) / beta, -alpha.y / beta);
        scale = crecip(cx<T>(alpha.x - beta, alpha.y));
    }
    sync();
    if (lane < len) vw[lane] = (lane == 0) ? cx<T>(T(1), T(0)) : x[lane * inc] * scale;
    sync();
}

// Apply H = I - tau v v^H (v on rows/cols [o, o+len)) to the m x m Ts from both sides (Ts <- H^H Ts H, left side on
// columns >= c0, right side on rows < nrows_t) and to Vs from the right (all m rows).  One wave, m <= 64.
template <class T, class SY = BlockSync>
__device__ void small_apply_reflector(cx<T>* Ts, cx<T>* Vs, int m, int nrows_t, int o, int len, int c0, const cx<T>* vw, cx<T> tau, const int SLD) {
    const SY sync;
    const int lane = threadIdx.x;
    if (lane >= c0 && lane < m) {                                   // left: lane = column
        cx<T> w(T(0), T(0));
        for (int i = 0; i < len; ++i) cfma_conj(w, vw[i], Ts[(o + i) * SLD + lane]);
        const cx<T> f = conj(tau) * w;
        for (int i = 0; i < len; ++i) Ts[(o + i) * SLD + lane] -= vw[i] * f;
    }
    sync();
    if (lane < nrows_t) {                                           // right on T: lane = row
        cx<T> w(T(0), T(0));
        for (int i = 0; i < len; ++i) cfma(w, Ts[lane * SLD + o + i], vw[i]);
        const cx<T> f = tau * w;
        for (int i = 0; i < len; ++i) Ts[lane * SLD + o + i] -= f * conj(vw[i]);
    }
    if (lane < m) {                                                 // right on V: lane = row
        cx<T> w(T(0), T(0));
        for (int i = 0; i < len; ++i) cfma(w, Vs[lane * SLD + o + i], vw[i]);
        const cx<T> f = tau * w;
        for (int i = 0; i < len; ++i) Vs[lane * SLD + o + i] -= f * conj(vw[i]);
    }
    sync();
}

template <class T>
__global__ __launch_bounds__(64) void qr_init_kernel(QrState* __restrict__ st, int n) {
    if (threadIdx.x == 0) {
        QrState s;
        s.ilo = 0; s.ihi = n - 1; s.nch = 0; s.mode = QR_IDLE; s.stall = 0; s.sweeps = 0; s.fail = 0; s.strip_next = 0;
        for (int c = 0; c < QKC; ++c) { s.k[c] = 0; s.tau[c][0] = s.tau[c][1] = 1; s.tau_last[c] = 0; s.w0[c] = 0; s.w1[c] = 0; }
        st[blockIdx.x] = s;
    }
}

__device__ __forceinline__ void clear_windows(QrState& st) {
#pragma unroll
    for (int c = 0; c < QKC; ++c) { st.w0[c] = 0; st.w1[c] = 0; }
}

// Lay out the chains of a sweep over an active block of m rows with `avail` shifts at hand: one chain per 128 rows of room
// beyond the first window (a follower needs ~3 window steps = 93 rows of distance), QNS shifts per chain, at most max_chains.
// Returns the total number of shifts; chain c (0 = first to run) gets k[c] of them.
__device__ __forceinline__ int plan_chains(QrState& st, int ilo, int ihi, int avail, int max_chains) {
    const int m = ihi - ilo + 1;
    int room = 1 + (m > 96 ? (m - 96) / 128 : 0);
    if (room > max_chains) room = max_chains;
    int ktot = avail < QNS * room ? avail : QNS * room;
    if (ktot < 0) ktot = 0;
    st.nch = (ktot + QNS - 1) / QNS;
#pragma unroll
    for (int c = 0; c < QKC; ++c) {
        int kc = ktot - QNS * c;
        kc = kc < 0 ? 0 : (kc > QNS ? QNS : kc);
        st.k[c] = kc;
        if (kc > 0) { st.tau[c][0] = st.tau[c][1] = 0; st.tau_last[c] = (ihi - 1 - ilo) + 2 * (kc - 1); }
        else { st.tau[c][0] = st.tau[c][1] = 1; st.tau_last[c] = 0; }
    }
    return ktot;
}
// position (0 = top of the shift list, ktot-1 = bottom-most eigenvalue) of shift s of chain c: chain 0 takes the bottom QNS
__device__ __forceinline__ int chain_shift_pos(const QrState& st, int ktot, int c, int s) { return ktot - QNS * c - st.k[c] + s; }

template <class T, bool BC>
__global__ __launch_bounds__(64) void qr_prepare_kernel(cx<T>* __restrict__ Aall, long mstride, int n, QrState* __restrict__ stall_,
                                                        cx<T>* __restrict__ Uall, cx<T>* __restrict__ shifts_all,
                                                        int* __restrict__ summary, int max_sweeps, int aed_w, int nibble, int aed_moves, int par, int max_chains,
                                                        int sm, int wantz, long long* dbg_all = nullptr) {
    TRX_DYN_SMEM(smem);
    long long* dbg = (dbg_all && blockIdx.x == 0) ? dbg_all : nullptr;       // cycle counters of matrix 0 (TRX_QR_DEBUG)
    long long tk0 = dbg ? clock64() : 0;
    const int SLD = sm + 1;                                      // sm: largest small matrix of this launch (AED window = small-block threshold)
    const int QNMIN = sm;                                        // active blocks up to this size are finished right here
    cx<T>* Hs = reinterpret_cast<cx<T>*>(smem);                  // [sm][SLD]
    cx<T>* Us = Hs + sm * SLD;                                   // [sm][SLD]
    cx<T>* vwork = Us + sm * SLD;                                // [sm]
    cx<T>* wk = vwork + sm;                                      // [sm]
    Rot<T>* rots = reinterpret_cast<Rot<T>*>(wk + sm);           // [sm]
    QrState& sst = *reinterpret_cast<QrState*>(rots + sm);
    const int b = blockIdx.x, lane = threadIdx.x;
    cx<T>* H = Aall + (long)b * mstride;
    if (lane == 0) sst = stall_[b];
    __syncthreads();
    QrState st = sst;
    if (st.mode == QR_DONE) return;
    if (st.mode == QR_CHASE) {
        // safety net: a chain that has not reached the bottom yet (the host's step count is a bound, not a guarantee) -- ask for more steps
        bool unfinished = false;
        for (int c = 0; c < st.nch; ++c) unfinished = unfinished || (st.tau[c][par] <= st.tau_last[c]);
        if (unfinished) {
            if (lane == 0) { atomicAdd(&summary[0], 1); atomicOr(&summary[2], 2); atomicMax(&summary[1], st.ihi + 1); }
            return;
        }
    }
    const T ulp = eps_of<T>::value;
    // 1. deflation scan (negligible subdiagonals -> exact zeros).  A non-finite entry (NaN/Inf input, e.g. from a singular
    //    convolution matrix upstream) can never deflate: report the whole active block as failed instead of iterating to the
    //    sweep limit.
    int bad = 0;
    for (int i = 1 + lane; i <= st.ihi; i += 64) {
        const cx<T> sub = H[(long)i * n + i - 1];
        const cx<T> dg = H[(long)i * n + i];
        if (!(abs1(sub) + abs1(dg) < std::numeric_limits<T>::infinity())) bad = 1;
        if (sub.x != T(0) || sub.y != T(0)) {
            T s = abs1(H[(long)(i - 1) * n + i - 1]) + abs1(H[(long)i * n + i]);
            if (s == T(0)) s = T(1);
            if (abs1(sub) <= ulp * s) H[(long)i * n + i - 1] = cx<T>(T(0), T(0));
        }
    }
    __syncthreads();
    if (__any(bad)) {
        if (lane == 0) { st.fail += st.ihi + 1; st.mode = QR_DONE; clear_windows(st); stall_[b] = st; }
        return;
    }
    // 2. new ihi = largest i in [1, ihi] with a non-zero subdiagonal (0 if none)
    int ihi = 0;
    for (int base = st.ihi; base >= 1; base -= 64) {
        const int i = base - lane;
        int pred = 0;
        if (i >= 1) { const cx<T> sub = H[(long)i * n + i - 1]; pred = (sub.x != T(0) || sub.y != T(0)); }
        const unsigned long long mask = __ballot(pred);
        if (mask) { ihi = base - __builtin_ctzll(mask); break; }
    }
    if (ihi <= 0) {
        if (lane == 0) { st.ihi = 0; st.mode = QR_DONE; clear_windows(st); stall_[b] = st; }
        return;
    }
    // 3. ilo = largest i in [1, ihi-1] with a zero subdiagonal (0 if none)
    int ilo = 0;
    for (int base = ihi - 1; base >= 1; base -= 64) {
        const int i = base - lane;
        int pred = 0;
        if (i >= 1) { const cx<T> sub = H[(long)i * n + i - 1]; pred = (sub.x == T(0) && sub.y == T(0)); }
        const unsigned long long mask = __ballot(pred);
        if (mask) { ilo = base - __builtin_ctzll(mask); break; }
    }
    if (ihi != st.ihi || ilo != st.ilo) st.stall = 0;
    st.ihi = ihi; st.ilo = ilo;
    const int m = ihi - ilo + 1;
    if (m <= QNMIN) {
        // finish this block in LDS, publish its unitary for the off-block update
        for (int e = lane; e < m * m; e += 64) {
            const int r = e / m, c = e - r * m;
            Hs[r * SLD + c] = (r <= c + 1) ? H[(long)(ilo + r) * n + ilo + c] : cx<T>(T(0), T(0));
            Us[r * SLD + c] = cx<T>(r == c ? T(1) : T(0), T(0));
        }
        __syncthreads();
        const bool ok = small_schur<T, BlockSync, BC>(Hs, m, Us, rots, SLD);
        __syncthreads();
        cx<T>* U = Uall + (long)b * QKC * QW * QW;            // chain slot 0 carries the unitary of a finished block / AED window
        for (int e = lane; e < m * m; e += 64) {
            const int r = e / m, c = e - r * m;
            H[(long)(ilo + r) * n + ilo + c] = (r <= c) ? Hs[r * SLD + c] : cx<T>(T(0), T(0));
            U[r * QW + c] = Us[r * SLD + c];
        }
        if (lane == 0) {
            clear_windows(st);
            if (wantz) { st.w0[0] = ilo; st.w1[0] = ihi + 1; }        // eigenvalues only: nothing outside the finished block needs its unitary
            st.mode = QR_SMALL_PENDING; st.ihi = ilo - 1; st.ilo = 0; st.stall = 0;
            if (!ok) st.fail += m;
            stall_[b] = st;
            atomicAdd(&summary[0], 1);
            atomicOr(&summary[2], 1);
            atomicMax(&summary[1], st.ihi + 1);
        }
        return;
    }
    // ---- aggressive early deflation on the trailing nw x nw window (Braman/Byers/Mathias; LAPACK zlaqr3) ----------
    if (st.sweeps >= max_sweeps) {
        if (lane == 0) { st.fail += ihi + 1; st.mode = QR_DONE; clear_windows(st); stall_[b] = st; }
        return;
    }
    const int nw = aed_w;                       // aed_w <= QAED <= QNMIN < m here, so the window is strictly inside the active block
    const int kw = ihi - nw + 1;
    cx<T> spike = H[(long)kw * n + kw - 1];
    for (int e = lane; e < nw * nw; e += 64) {
        const int r = e / nw, c = e - r * nw;
        Hs[r * SLD + c] = (r <= c + 1) ? H[(long)(kw + r) * n + kw + c] : cx<T>(T(0), T(0));
        Us[r * SLD + c] = cx<T>(r == c ? T(1) : T(0), T(0));
    }
    __syncthreads();
    if (dbg && lane == 0) { const long long t1 = clock64(); dbg[6] += t1 - tk0; tk0 = t1; }       // scans + window load
    const bool okw = small_schur<T, BlockSync, BC>(Hs, nw, Us, rots, SLD, dbg);
    __syncthreads();
    if (dbg && lane == 0) { const long long t1 = clock64(); dbg[7] += t1 - tk0; tk0 = t1; }       // Schur total
    int ns = nw;
    if (okw) {
        const T smlnum = eps_of<T>::safmin * ((T)n / ulp);
        // LAPACK moves every undeflatable eigenvalue to the top of the window (O(nw^2) sequential swaps).  Moving at most
        // QAED_MOVES of them keeps the deflation yield (prototype: same shift-row count as full reordering at 12 moves)
        // at a fraction of the latency; the unexamined eigenvalues simply stay undeflated.
        int ilst = 0, moves = 0;
        while (ilst < ns) {
            T foo = abs1(Hs[(ns - 1) * SLD + ns - 1]);
            if (foo == T(0)) foo = abs1(spike);
            const T sp = abs1(spike) * abs1(Us[ns - 1]);            // |s| |V[0, ns-1]|
            const T thr = (smlnum > ulp * foo) ? smlnum : ulp * foo;
            if (sp <= thr) {
                --ns;                                                // deflatable
            } else {
                if (moves >= aed_moves) break;
                for (int kk = ns - 2; kk >= ilst; --kk) schur_swap<T>(Hs, Us, nw, kk, SLD);
                ++ilst; ++moves;
            }
        }
    }
    const int nd = nw - ns;
    if (dbg && lane == 0) { const long long t1 = clock64(); dbg[8] += t1 - tk0; tk0 = t1; dbg[11] += 1; }     // spike test + reordering
    cx<T>* sh = shifts_all + (long)b * QKC * QNS;
    if (nd == 0) {
        // nothing deflates: H is untouched; the window's eigenvalues (bottom k of them) are the shifts of a full sweep
        int avail = m / 2 < nw ? m / 2 : nw;
        const int k = plan_chains(st, ilo, ihi, avail, max_chains);
        {
            const int c = lane / QNS, sidx = lane - c * QNS;
            if (c < st.nch && sidx < st.k[c]) {
                const int pos = nw - k + chain_shift_pos(st, k, c, sidx);
                cx<T> sv = Hs[pos * SLD + pos];
                if (st.stall > 0 && (st.stall % 6) == 0) {
                    const T mag = T(0.75) * cabs(H[(long)ihi * n + ihi - 1]);
                    const T ang = T(6.283185307179586) * (T)lane / (T)k;
                    sv = sv + cx<T>(mag * (T)cos(ang), mag * (T)sin(ang));
                }
                sh[c * QNS + sidx] = sv;
            }
        }
        if (lane == 0) {
            st.mode = QR_CHASE;
            st.stall += 1; st.sweeps += 1; clear_windows(st);
            stall_[b] = st;
            atomicAdd(&summary[0], 1);
            atomicMax(&summary[1], ihi + 1);
        }
        return;
    }
    // nd > 0: commit the window in Schur/Hessenberg form and publish V for the off-window update
    if (ns == 0) spike = cx<T>(T(0), T(0));
    // "nibble" rule: after a deflation of less than `nibble` % of the window the undeflated window eigenvalues are used
    // as shifts of a sweep in the SAME outer iteration (first window slot applies V, the following ones chase).
    const int m2 = (ihi - nd) - ilo + 1;
    int kch = 0;
    QrState stc = st;                       // chain layout of the sweep that follows the AED (committed below if it is taken)
    if (nd * 100 < nibble * nw && m2 > QNMIN && ns >= 2) {
        kch = plan_chains(stc, ilo, ihi - nd, (m2 / 2 < ns) ? m2 / 2 : ns, max_chains);
        const int c = lane / QNS, sidx = lane - c * QNS;
        if (kch >= 2 && c < stc.nch && sidx < stc.k[c]) {
            const int pos = ns - kch + chain_shift_pos(stc, kch, c, sidx);
            sh[c * QNS + sidx] = Hs[pos * SLD + pos];                                 // eigenvalues, before the restore below
        }
    }
    if (ns > 1 && (spike.x != T(0) || spike.y != T(0))) {
        // reflector that maps the spike s*conj(V[0,0:ns]) onto e1, then return T[0:ns,0:ns] to Hessenberg form
        cx<T>* vw = vwork;
        if (lane < ns) wk[lane] = conj(Us[lane]);
        __syncthreads();
        cx<T> tau; T beta;
        small_larfg<T>(wk, 1, ns, vw, tau, beta);
        small_apply_reflector<T>(Hs, Us, nw, ns, 0, ns, 0, vw, tau, SLD);
        for (int jc = 0; jc + 2 < ns; ++jc) {
            small_larfg<T>(Hs + (jc + 1) * SLD + jc, SLD, ns - jc - 1, vw, tau, beta);
            if (lane == 0) Hs[(jc + 1) * SLD + jc] = cx<T>(beta, T(0));
            if (lane >= 1 && lane < ns - jc - 1) Hs[(jc + 1 + lane) * SLD + jc] = cx<T>(T(0), T(0));
            __syncthreads();
            small_apply_reflector<T>(Hs, Us, nw, ns, jc + 1, ns - jc - 1, jc + 1, vw, tau, SLD);
        }
    }
    __syncthreads();
    if (dbg && lane == 0) { const long long t1 = clock64(); dbg[9] += t1 - tk0; tk0 = t1; }       // Hessenberg restore
    {
        cx<T>* U = Uall + (long)b * QKC * QW * QW;
        for (int e = lane; e < nw * nw; e += 64) {
            const int r = e / nw, c = e - r * nw;
            cx<T> v = Hs[r * SLD + c];
            if (r > c + 1 || (r == c + 1 && r >= ns)) v = cx<T>(T(0), T(0));
            H[(long)(kw + r) * n + kw + c] = v;
            U[r * QW + c] = Us[r * SLD + c];
        }
        if (lane == 0) {
            H[(long)kw * n + kw - 1] = spike * conj(Us[0]);
            if (kch >= 2) {
                const int sw = st.sweeps;
                st = stc;                      // chains planned above (k, tau, tau_last per chain)
                st.mode = QR_AED_CHASE;
                st.ihi = ihi - nd;
                st.sweeps = sw + 1;
                atomicMax(&summary[1], st.ihi + 1);
            } else {
                st.mode = QR_SMALL_PENDING;
            }
            clear_windows(st);
            st.w0[0] = kw; st.w1[0] = ihi + 1; st.stall = 0;
            stall_[b] = st;
            atomicAdd(&summary[0], 1);
            atomicOr(&summary[2], 1);
            atomicMax(&summary[1], st.ihi + 1);
            if (dbg) { const long long t1 = clock64(); dbg[10] += t1 - tk0; }                      // write-back
        }
    }
}

// Window of a chain whose chase step is tau: starts one row above its last bulge (clamped to the active block), QW wide.
__device__ __forceinline__ void chain_window(int ilo, int ihi, int k, int tau, int tau_last, int& w0, int& w1, int& tau_end) {
    w0 = ilo + tau - 2 * (k - 1) - 1;
    if (w0 < ilo) w0 = ilo;
    w1 = w0 + QW;
    if (w1 > ihi + 1) w1 = ihi + 1;
    tau_end = (w1 == ihi + 1) ? tau_last : (w1 - 3 - ilo);
}

// One window step of one bulge chain (blockIdx.x = chain, blockIdx.y = matrix).  Thread layout: QNS groups of LPB lanes, group s owns bulge s: its lanes compute the
// rotation redundantly (no broadcast barrier), then stride over the window's columns (left rotation) and, after one barrier,
// over its rows and the rows of U (right rotation).  Two barriers per chain step.  LPB = 64 (one wave per bulge, 1024
// threads): a chain step is bound by the fp64 vector issue time of the 2 + 4 rotated element pairs per lane-quartet (cycle
// counters, TRX_QR_DEBUG: 16 lanes per bulge spent 3600 cycles per step, ~2600 of them in the two rotation phases), so the
// widest mapping the 64-wide window admits is the fastest one.
// Two phases in ONE window-sized LDS buffer: (1) the chase on the H window, every rotation logged (c, s: 24 bytes); H written
// back; (2) the same buffer becomes U = I and the log is replayed onto it (bulge s's wave rotates its two columns, one barrier
// per chain step).  With H and U side by side the kernel needed 133 KB of LDS; now 66.5 KB + the log.  (The hope that a smaller
// footprint would let window launches start next to the slab-update workgroups of the other iteration groups did not
// materialise: a slab launch occupies every workgroup slot of the chip whatever is left of a CU's LDS -- DESIGN.md section 9.)
constexpr int LPB = 64;                    // lanes per bulge
constexpr int WTHREADS = QNS * LPB;        // threads of the window kernel
constexpr int WIT = QW / LPB;              // element pairs per lane and phase
constexpr int WMAXS = 96;                  // chain steps per launch (rotation log: WMAXS x QNS entries = 37 KB): the first window of a sweep
                                           // chases 62 steps and the last one up to ~94, so no launch is split (48 split 7 % of them)
template <class T> struct RotCS { T c; cx<T> s; };
// DBG: cycle counters of matrix 0, chain 0 (TRX_QR_DEBUG); the production instantiation carries none of it (it must stay within
// 64 VGPRs: 4 of its waves share a SIMD's 512 registers with one 240-register wave of a slab-update workgroup).
template <class T, bool DBG>
__global__ __launch_bounds__(WTHREADS) void qr_window_kernel(cx<T>* __restrict__ Aall, long mstride, int n, QrState* __restrict__ st_all,
                                                        cx<T>* __restrict__ Uall, const cx<T>* __restrict__ shifts_all, int par, long long* dbg_all = nullptr) {
    TRX_DYN_SMEM(smem);
    long long* dbg = (DBG && dbg_all && blockIdx.y == 0 && blockIdx.x == 0 && threadIdx.x == 0) ? dbg_all : nullptr;
    long long tk0 = dbg ? clock64() : 0;
    constexpr int LD = QW + 1;
    cx<T>* Hw = reinterpret_cast<cx<T>*>(smem);      // [QW][LD]   phase 1: H window;  phase 2: U
    RotCS<T>* rlog = reinterpret_cast<RotCS<T>*>(Hw + QW * LD);      // [WMAXS][QNS]
    QrState& sst = *reinterpret_cast<QrState*>(rlog + WMAXS * QNS);
    const int b = blockIdx.y, ch = blockIdx.x, t = threadIdx.x;
    if (t == 0) { sst = st_all[b]; if (ch == 0) st_all[b].strip_next = 0; }
    __syncthreads();
    const QrState& st = sst;           // read in place (LDS): a register copy indexed by the chain number would live in scratch
    // Every block writes only the fields of its own chain (and chain 0's block the mode); reads of the other chains' chase
    // positions go to the [par] copy, which nobody writes in this step.
    if (st.mode == QR_SMALL_PENDING) { if (t == 0 && ch == 0) st_all[b].mode = QR_SMALL_APPLIED; return; }      // this slot applies the block's unitary
    if (st.mode == QR_AED_CHASE) { if (t == 0 && ch == 0) st_all[b].mode = QR_CHASE; return; }                  // this slot applies the AED unitary
    const int tau0 = st.tau[ch][par];
    bool move = (st.mode == QR_CHASE) && ch < st.nch && tau0 <= st.tau_last[ch];
    const int k = st.k[ch], ilo = st.ilo, ihi = st.ihi;
    int w0 = 0, w1 = 0, tau_end = 0;
    if (move) {
        chain_window(ilo, ihi, k, tau0, st.tau_last[ch], w0, w1, tau_end);
        if (tau_end > tau0 + WMAXS - 1) tau_end = tau0 + WMAXS - 1;      // first and last window of a sweep: several launches
        if (ch > 0 && st.tau[ch - 1][par] <= st.tau_last[ch - 1]) {
            // the chain ahead is still under way: its last bulge sits at the start of ITS window (whether or not it moves in
            // this step); this chain may only work strictly above it
            int p0, p1, pe;
            chain_window(ilo, ihi, st.k[ch - 1], st.tau[ch - 1][par], st.tau_last[ch - 1], p0, p1, pe);
            if (w1 > p0) move = false;
        }
    }
    if (!move) {
        if (t == 0) {
            if (st.w0[ch] != 0 || st.w1[ch] != 0) { st_all[b].w0[ch] = 0; st_all[b].w1[ch] = 0; }
            st_all[b].tau[ch][par ^ 1] = tau0;
        }
        return;
    }
    cx<T>* H = Aall + (long)b * mstride;
    const int ww = w1 - w0;
    {
        // window load: QW*QW/WTHREADS independent (clamped) global loads per thread in flight, then the LDS fill
        constexpr int RPT = QW * QW / WTHREADS, RSTEP = WTHREADS / QW;      // rows per thread, row stride between them
        const int c = t & (QW - 1), r4 = t / QW;
        const int cc = c < ww ? c : ww - 1;
        cx<T> hv[RPT];
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int r = r4 + RSTEP * i;
            hv[i] = H[(long)(w0 + (r < ww ? r : ww - 1)) * n + w0 + cc];
        }
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int r = r4 + RSTEP * i;
            if (r < ww && c < ww) Hw[r * LD + c] = hv[i];
        }
    }
    const int sb = t / LPB, j = t & (LPB - 1);       // bulge index, lane within the group
    const cx<T> shift = (sb < k) ? shifts_all[((long)b * QKC + ch) * QNS + sb] : cx<T>(T(0), T(0));
    __syncthreads();
    if (dbg) { const long long t1 = clock64(); dbg[12] += t1 - tk0; tk0 = t1; }
    // Every phase of a chain step touches WIT element pairs per lane: the loops are fully unrolled with clamped LDS reads issued
    // up front and guarded writes, so a phase costs one LDS round trip.  (Computing the next rotation right after the H part of
    // the right phase, to overlap it with the U part, was tried and is not faster: hipcc serialises the two and the second
    // barrier absorbs the skew.)
    for (int tau = tau0; tau <= tau_end; ++tau) {
        const int p = ilo + tau - 2 * sb;
        const bool active = (sb < k) && (p >= ilo) && (p <= ihi - 1);
        const int q = p - w0;
        const bool first = (p == ilo);
        Rot<T> R;
        if (active) {
            cx<T> f, g;
            if (first) { f = Hw[q * LD + q] - shift; g = Hw[(q + 1) * LD + q]; }
            else { f = Hw[q * LD + q - 1]; g = Hw[(q + 1) * LD + q - 1]; }
            R = rotg_fast(f, g);
        } else { R.c = T(1); R.s = cx<T>(T(0), T(0)); R.r = cx<T>(T(0), T(0)); }
        if (j == 0) { rlog[(tau - tau0) * QNS + sb].c = R.c; rlog[(tau - tau0) * QNS + sb].s = R.s; }
        wave_sync();                                  // all lanes have read (f, g) before any lane overwrites them
        if (dbg) { const long long t1 = clock64(); dbg[16] += t1 - tk0; tk0 = t1; }
        if (active) {
            const int lo = first ? q : q - 1;
            cx<T> x[WIT], y[WIT];
#pragma unroll
            for (int it = 0; it < WIT; ++it) {
                const int col = lo + j + LPB * it;
                const int cl = col < ww ? col : ww - 1;
                x[it] = Hw[q * LD + cl]; y[it] = Hw[(q + 1) * LD + cl];
            }
#pragma unroll
            for (int it = 0; it < WIT; ++it) {
                const int col = lo + j + LPB * it;
                rot_rows(R, x[it], y[it]);
                if (!first && col == q - 1) { x[it] = R.r; y[it] = cx<T>(T(0), T(0)); }
                if (col < ww) { Hw[q * LD + col] = x[it]; Hw[(q + 1) * LD + col] = y[it]; }
            }
        }
        if (dbg) { const long long t1 = clock64(); dbg[17] += t1 - tk0; tk0 = t1; }
        __syncthreads();
        if (dbg) { const long long t1 = clock64(); dbg[18] += t1 - tk0; tk0 = t1; }
        if (active) {
            const int hi = (q + 2 < ww - 1) ? q + 2 : ww - 1;
            cx<T> xh[WIT], yh[WIT];
#pragma unroll
            for (int it = 0; it < WIT; ++it) {
                const int row = j + LPB * it;
                const int rh = row <= hi ? row : hi;
                xh[it] = Hw[rh * LD + q]; yh[it] = Hw[rh * LD + q + 1];
            }
#pragma unroll
            for (int it = 0; it < WIT; ++it) {
                const int row = j + LPB * it;
                rot_cols(R, xh[it], yh[it]);
                if (row <= hi) { Hw[row * LD + q] = xh[it]; Hw[row * LD + q + 1] = yh[it]; }
            }
        }
        if (dbg) { const long long t1 = clock64(); dbg[19] += t1 - tk0; tk0 = t1; }
        __syncthreads();
        if (dbg) { const long long t1 = clock64(); dbg[20] += t1 - tk0; tk0 = t1; }
    }
    if (dbg) { dbg[13] += dbg[16] + dbg[17] + dbg[18] + dbg[19] + dbg[20] - dbg[13]; dbg[15] += tau_end - tau0 + 1; tk0 = clock64(); }
    cx<T>* U = Uall + ((long)b * QKC + ch) * QW * QW;
    constexpr int RPT = QW * QW / WTHREADS, RSTEP = WTHREADS / QW;
    {
        // phase 1 done: the window goes back to H, the buffer becomes U = I
        const int c = t & (QW - 1), r4 = t / QW;
        cx<T> hv[RPT];
#pragma unroll
        for (int i = 0; i < RPT; ++i) hv[i] = Hw[(r4 + RSTEP * i) * LD + c];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int r = r4 + RSTEP * i;
            if (r < ww && c < ww) H[(long)(w0 + r) * n + w0 + c] = hv[i];
            Hw[r * LD + c] = cx<T>(r == c ? T(1) : T(0), T(0));
        }
    }
    __syncthreads();
    // phase 2: replay the logged rotations onto U (right multiplications; within a chain step the bulges own disjoint column pairs)
    for (int tau = tau0; tau <= tau_end; ++tau) {
        const int p = ilo + tau - 2 * sb;
        const bool active = (sb < k) && (p >= ilo) && (p <= ihi - 1);
        if (active) {
            const int q = p - w0;
            Rot<T> R;
            R.c = rlog[(tau - tau0) * QNS + sb].c; R.s = rlog[(tau - tau0) * QNS + sb].s;
            cx<T> xu[WIT], yu[WIT];
#pragma unroll
            for (int it = 0; it < WIT; ++it) {
                const int row = j + LPB * it;
                const int ru = row < ww ? row : ww - 1;
                xu[it] = Hw[ru * LD + q]; yu[it] = Hw[ru * LD + q + 1];
            }
#pragma unroll
            for (int it = 0; it < WIT; ++it) {
                const int row = j + LPB * it;
                rot_cols(R, xu[it], yu[it]);
                if (row < ww) { Hw[row * LD + q] = xu[it]; Hw[row * LD + q + 1] = yu[it]; }
            }
        }
        __syncthreads();
    }
    {
        const int c = t & (QW - 1), r4 = t / QW;
        if (c < ww) {
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int r = r4 + RSTEP * i;
                if (r < ww) U[r * QW + c] = Hw[r * LD + c];
            }
        }
    }
    if (t == 0) {
        st_all[b].tau[ch][par ^ 1] = tau_end + 1; st_all[b].w0[ch] = w0; st_all[b].w1[ch] = w1;
    }
    if (dbg) dbg[14] += clock64() - tk0;
}

// Off-window updates on the matrix cores: H[w0:w1, w1:n) <- U^H H[w0:w1, w1:n)   (left),  H[0:w0, w0:w1) <- H[0:w0, w0:w1) U  and
// Z[:, w0:w1) <- Z[:, w0:w1) U   (right), with the window unitary U (ww x ww, ww <= 64) of the last window step.
//
// Streaming design: the work of one matrix is cut into STRIPS of 16 columns (left) or 16 rows (right); one wave owns a strip,
// i.e. a 64 x 16 / 16 x 64 output block = four 16x16 MFMA tiles with the full K = ww.  U is staged ONCE per workgroup into LDS
// (split re/im planes) and serves as the A operand of the left update (U^H: the conjugation is folded into the signs of the
// four real MFMAs) and as the B operand of the right update; the streamed operand goes global memory -> registers directly
// in MFMA fragment layout (no LDS staging, no barrier after the prologue), and the next strip of the wave is prefetched into
// registers while the current one is multiplied.  The k index of an MFMA step is permuted (k = 16c + 4*(lane>>4) + j, kstep()) so that
// a lane of the right update reads 4 consecutive elements of its row; both operands use the same permutation.
// Algorithmic intensity: 8*16*64*64 flops per 2*16 KiB moved = 16 flop/B, i.e. the update sits at the MFMA/HBM balance point
// (78.6 TF / 16 = 4.9 TB/s): every byte is touched exactly once per window step.
constexpr int MLD = 72;           // LDS plane row stride: element U[k][c] at [k*MLD + c]
// k index a lane of k-group lk (= lane >> 4) supplies at MFMA step (h, cc, j): k = kstep(h, cc, j) + KLS * lk.  Permuted order
// (16 c + 4 lk + j): the four loads (j) of a lane of the right update are 64 contiguous bytes.  The natural order 4 step + lk
// (the four k-groups of ONE instruction contiguous instead) was measured too: 28.0 vs 28.4 solves/s, so the permutation stays.
constexpr int KLS = 4;
__device__ __forceinline__ constexpr int kstep(int h, int cc, int j) { return 16 * (2 * h + cc) + j; }

template <class T>
struct SlabStrip {       // wave-uniform description of one strip
    cx<T>* X;            // matrix the strip lives in (H or Z of this batch entry)
    int side;            // 0 = left update (16 columns a0.. of rows w0..w1), 1 = right update (16 rows a0.. of columns w0..w1)
    int a0, lim;         // first column / row of the strip, end of the valid column / row range
};

// Regions of the off-window update of one window step.  wantz (Schur form): left update on all columns right of the window, right update
// on all rows above it, and the window's columns of Z.  Eigenvalues only: the left update stops at the end of the active block, the right
// update starts at its first row, Z is not touched -- everything outside the active diagonal block is irrelevant to the eigenvalues.
struct SlabRanges {
    int nL, nR, nZ;      // strips (16 columns / rows) of the left update, the right update of H, the update of Z
    int lc0, lclim;      // left update: first column, end of the column range
    int rr0;             // right update of H: first row (the range ends at w0)
};
__device__ __forceinline__ SlabRanges slab_ranges(const QrState& st, int n, int w0, int w1, int wantz) {
    SlabRanges r;
    r.lc0 = w1;
    if (wantz) { r.lclim = n; r.rr0 = 0; r.nZ = (n + 15) >> 4; }
    else { r.lclim = st.ihi + 1 < n ? st.ihi + 1 : n; r.rr0 = st.ilo < w0 ? st.ilo : w0; r.nZ = 0; }
    r.nL = r.lclim > w1 ? (r.lclim - w1 + 15) >> 4 : 0;
    r.nR = (w0 - r.rr0 + 15) >> 4;
    return r;
}

// strip g of PART 0 (left update: nL strips of 16 columns of H right of the window), PART 1 (right updates: nR strips of 16
// rows of H above the window, then the strips of Z) or PART 2 (everything: left | right-H | Z)
template <class T, int PART>
__device__ __forceinline__ SlabStrip<T> slab_locate(int g, const SlabRanges& R, cx<T>* H, cx<T>* Z, int n, int w0) {
    SlabStrip<T> d;
    if (PART == 2) { if (g < R.nL) { d.X = H; d.side = 0; d.a0 = R.lc0 + 16 * g; d.lim = R.lclim; return d; } g -= R.nL; }
    if (PART == 0) { d.X = H; d.side = 0; d.a0 = R.lc0 + 16 * g; d.lim = R.lclim; }
    else if (g < R.nR) { d.X = H; d.side = 1; d.a0 = R.rr0 + 16 * g; d.lim = w0; }
    else { d.X = Z; d.side = 1; d.a0 = 16 * (g - R.nR); d.lim = n; }
    return d;
}

// registers x[4cc + j] <- streamed operand element k = kstep(h, cc, j) + lk (half h of the strip's K range) of this lane's
// column (left) / row (right).  Out-of-range coordinates are CLAMPED to a valid element of the same matrix and the value is
// used as is: for k >= ww it meets a zero row of the padded U in LDS, and a lane whose row / column lies outside the region
// only feeds output elements that are never stored.  (No select after the load: the loaded registers have no consumer until
// the MFMAs of the next strip, so the loads stay in flight behind the current strip's arithmetic.)
// FULL (the window has all QW rows / columns: no clamp on k): the address is split into ONE lane-dependent 32-bit offset per strip
// and a wave-uniform part per load that goes into the scalar base, so that 16 loads cost one address register instead of 16.
template <class T, bool FULL = false>
__device__ __forceinline__ void slab_load_half(const SlabStrip<T>& d, int h, int n, int w0, int ww, int lane, cx<T> (&x)[8]) {
    const int lr = lane & 15, lk = lane >> 4;
    const int a = d.a0 + lr;
    const int ac = a < d.lim ? a : d.lim - 1;
    if constexpr (FULL) {
        const unsigned ksu = (d.side == 0 ? (unsigned)n : 1u) * (unsigned)sizeof(cx<T>);                 // wave-uniform
        const unsigned lane_off = (d.side == 0 ? (unsigned)w0 * n + ac : (unsigned)ac * n + w0) * (unsigned)sizeof(cx<T>) + (unsigned)(KLS * lk) * ksu;
        const char* base = reinterpret_cast<const char*>(d.X);
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const char* pb = base + (size_t)kstep(h, cc, j) * ksu;                                        // scalar
                x[4 * cc + j] = *reinterpret_cast<const cx<T>*>(pb + lane_off);
            }
        return;
    }
    // 32-bit BYTE offsets from the wave-uniform matrix base (scalar base + 32-bit vector offset addressing; one address
    // register per access instead of two): requires n*n*sizeof(cx<T>) < 4 GiB, i.e. n < 16384 for complex128 (checked on the host)
    const unsigned p0 = (d.side == 0 ? (unsigned)w0 * n + ac : (unsigned)ac * n + w0) * (unsigned)sizeof(cx<T>);
    const unsigned ks = (d.side == 0 ? n : 1) * (unsigned)sizeof(cx<T>);
    const char* base = reinterpret_cast<const char*>(d.X);
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = kstep(h, cc, j) + KLS * lk;
            x[4 * cc + j] = *reinterpret_cast<const cx<T>*>(base + (p0 + (unsigned)(k < ww ? k : ww - 1) * ks));
        }
}

template <class T, int SIDE>
__device__ __forceinline__ void slab_multiply_half(const T* __restrict__ Ur, const T* __restrict__ Ui, int h, int lane, const cx<T> (&x)[8],
                                                   typename Mfma<T>::acc_t (&accR)[4], typename Mfma<T>::acc_t (&accI)[4]) {
    const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int off = (kstep(h, cc, j) + KLS * lk) * MLD + lr;
            const T xr = x[4 * cc + j].x, xi = x[4 * cc + j].y;
            T ur[4], ui[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { ur[q] = Ur[off + 16 * q]; ui[q] = Ui[off + 16 * q]; }
            if (SIDE == 1) {          // C = X U:      Cr += xr ur - xi ui,  Ci += xr ui + xi ur          (A = x, B = u)
                const T nxi = -xi;
#pragma unroll
                for (int q = 0; q < 4; ++q) accR[q] = Mfma<T>::mma(xr, ur[q], accR[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) accI[q] = Mfma<T>::mma(xr, ui[q], accI[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) accR[q] = Mfma<T>::mma(nxi, ui[q], accR[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) accI[q] = Mfma<T>::mma(xi, ur[q], accI[q]);
            } else {                  // C = U^H X:    Cr += ur xr + ui xi,  Ci += ur xi - ui xr          (A = conj(u)^T, B = x)
                const T nxr = -xr;
#pragma unroll
                for (int q = 0; q < 4; ++q) accR[q] = Mfma<T>::mma(ur[q], xr, accR[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) accI[q] = Mfma<T>::mma(ur[q], xi, accI[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) accR[q] = Mfma<T>::mma(ui[q], xi, accR[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) accI[q] = Mfma<T>::mma(ui[q], nxr, accI[q]);
            }
        }
}

// One strip: issue the loads of both K halves, multiply, store.  There is no explicit cross-strip prefetch: the two
// workgroups (8 waves) resident on a CU run out of phase, so one wave's load latency hides behind the MFMAs of its SIMD
// neighbour (an explicit register double-buffer pushes the kernel over 256 VGPRs and the spill reloads, which share the
// vector-memory counter with the prefetch, then serialise everything -- measured on the ISA).
// 3M variant of one K half for the tile pair (2 pp, 2 pp + 1): three real MFMAs per k-step and tile (mfma.hpp).
//   right update (C = X U):    P1 += xr ur,  P2 += xi ui,  P3 += (xr + xi)(ur + ui);   Cr = P1 - P2,  Ci = P3 - P1 - P2
//   left update  (C = U^H X):  P1 += ur xr,  P2 += ui xi,  P3 += (ur - ui)(xr + xi);   Cr = P1 + P2,  Ci = P3 - P1 + P2
//
// BAND: the unitary of a chain step is a product of at most QNS bulge passes, each an ascending sequence of adjacent-column
// rotations, i.e. an upper Hessenberg matrix (rotations of different bulges that are interleaved in time act two or more columns apart
// and commute into that order): U has at most QNS = 16 nonzero subdiagonals.  With 16-wide k chunks c and 16-wide output tiles t
// (t indexes the COLUMNS of U on either side) the block (c, t) is structurally zero for c >= t + 2 -- (2,0), (3,0), (3,1): 3 of the
// 16 blocks, 19 % of the matrix-core work.  The conditions fold at compile time (h, pp, cc, q are unrolled constants); whether a
// given U has the band is decided by the kernel when it stages U into LDS (dense AED / small-block unitaries use the same kernel).
template <class T, int SIDE, bool BAND = false>
__device__ __forceinline__ void slab_multiply_half_3m(const T* __restrict__ Ur, const T* __restrict__ Ui, int h, int pp, int lane, const cx<T> (&x)[8],
                                                      typename Mfma<T>::acc_t (&p1)[2], typename Mfma<T>::acc_t (&p2)[2], typename Mfma<T>::acc_t (&p3)[2]) {
    const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int chunk = 2 * h + cc;
            const bool use0 = !(BAND && chunk >= 2 * pp + 2), use1 = !(BAND && chunk >= 2 * pp + 3);       // tiles 2pp, 2pp + 1
            if (!use0 && !use1) continue;
            const int off = (kstep(h, cc, j) + KLS * lk) * MLD + lr + 32 * pp;
            const T xr = x[4 * cc + j].x, xi = x[4 * cc + j].y;
            const T xs = xr + xi;
            T ur[2], ui[2], us[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (!(q == 0 ? use0 : use1)) continue;
                ur[q] = Ur[off + 16 * q]; ui[q] = Ui[off + 16 * q];
                us[q] = SIDE == 1 ? ur[q] + ui[q] : ur[q] - ui[q];
            }
            if (SIDE == 1) {
#pragma unroll
                for (int q = 0; q < 2; ++q) if (q == 0 ? use0 : use1) p1[q] = Mfma<T>::mma(xr, ur[q], p1[q]);
#pragma unroll
                for (int q = 0; q < 2; ++q) if (q == 0 ? use0 : use1) p2[q] = Mfma<T>::mma(xi, ui[q], p2[q]);
#pragma unroll
                for (int q = 0; q < 2; ++q) if (q == 0 ? use0 : use1) p3[q] = Mfma<T>::mma(xs, us[q], p3[q]);
            } else {
#pragma unroll
                for (int q = 0; q < 2; ++q) if (q == 0 ? use0 : use1) p1[q] = Mfma<T>::mma(ur[q], xr, p1[q]);
#pragma unroll
                for (int q = 0; q < 2; ++q) if (q == 0 ? use0 : use1) p2[q] = Mfma<T>::mma(ui[q], xi, p2[q]);
#pragma unroll
                for (int q = 0; q < 2; ++q) if (q == 0 ? use0 : use1) p3[q] = Mfma<T>::mma(us[q], xs, p3[q]);
            }
            // bound the hoisting of the U fragment reads to the next two k-steps: unbounded, hipcc keeps the fragments of a whole K half
            // live (96 registers), which the register double buffer of the streamed operand has no room for
            if (j & 1) __builtin_amdgcn_sched_barrier(0);
        }
}

// FULL: `w0` is the origin of a full QW-wide window frame and `ww` packs the range of frame indices that belong to the real window
// (klo | khi << 8): only those rows (left update) / columns (right update) are stored, the identity-padded rest is left untouched.
template <class T, bool FULL = false>
__device__ __forceinline__ void slab_store_pair(const SlabStrip<T>& d, int n, int w0, int ww, int lane, int pp, const typename Mfma<T>::acc_t (&accR)[2],
                                                const typename Mfma<T>::acc_t (&accI)[2]) {
    const int lr = lane & 15;
    char* base = reinterpret_cast<char*>(d.X);       // scalar base + 32-bit byte offsets, as in slab_load_half
    if constexpr (FULL) {
        // one lane-dependent offset; the (r, q) part of every store is wave-uniform and moves into the scalar base
        const int cr0 = Mfma<T>::crow(lane, 0), rs = Mfma<T>::crow(0, 1) - Mfma<T>::crow(0, 0);
        const unsigned esz = (unsigned)sizeof(cx<T>);
        const int klo = ww & 255, khi = ww >> 8;
        if (d.side == 1) {
            const unsigned lane_off = ((unsigned)(d.a0 + cr0) * n + w0 + lr) * esz;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool okr = d.a0 + cr0 + r * rs < d.lim;
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const int k = 16 * (2 * pp + q2) + lr;
                    char* pb = base + ((size_t)(r * rs) * n + 16 * (2 * pp + q2)) * esz;
                    if (okr && k >= klo && k < khi) *reinterpret_cast<cx<T>*>(pb + lane_off) = cx<T>(accR[q2][r], accI[q2][r]);
                }
            }
        } else {
            const unsigned lane_off = ((unsigned)(w0 + cr0) * n + d.a0 + lr) * esz;
            const bool okc = d.a0 + lr < d.lim;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q2 = 0; q2 < 2; ++q2) {
                    const int i = 16 * (2 * pp + q2) + cr0 + r * rs;
                    char* pb = base + (size_t)(16 * (2 * pp + q2) + r * rs) * n * esz;
                    if (okc && i >= klo && i < khi) *reinterpret_cast<cx<T>*>(pb + lane_off) = cx<T>(accR[q2][r], accI[q2][r]);
                }
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int cr = Mfma<T>::crow(lane, r);
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
            const int q = 2 * pp + q2;
            const cx<T> v(accR[q2][r], accI[q2][r]);
            if (d.side == 1) {
                const int row = d.a0 + cr, k = 16 * q + lr;
                if (row < d.lim && k < ww) *reinterpret_cast<cx<T>*>(base + ((unsigned)row * n + w0 + k) * (unsigned)sizeof(cx<T>)) = v;
            } else {
                const int i = 16 * q + cr, col = d.a0 + lr;
                if (i < ww && col < d.lim) *reinterpret_cast<cx<T>*>(base + ((unsigned)(w0 + i) * n + col) * (unsigned)sizeof(cx<T>)) = v;
            }
        }
    }
}

template <class T, int SIDE, bool FULL = false, bool BAND = false>
__device__ __forceinline__ void slab_compute(const T* __restrict__ Ur, const T* __restrict__ Ui, const SlabStrip<T>& d, int n, int w0, int ww, int lane,
                                             const cx<T> (&xa)[8], const cx<T> (&xb)[8]);

template <class T, int SIDE, bool BAND = false>
__device__ __forceinline__ void slab_strip(const T* __restrict__ Ur, const T* __restrict__ Ui, const SlabStrip<T>& d, int n, int w0, int ww, int lane) {
    cx<T> xa[8], xb[8];
    slab_load_half<T>(d, 0, n, w0, ww, lane, xa);
    slab_load_half<T>(d, 1, n, w0, ww, lane, xb);
    __builtin_amdgcn_sched_barrier(0);       // keep all 16 loads of the strip in flight ahead of the first MFMA (hipcc otherwise sinks them to ~3 deep)
    slab_compute<T, SIDE, false, BAND>(Ur, Ui, d, n, w0, ww, lane, xa, xb);
}

// multiply + store of one strip whose streamed operand is already in (or on its way to) registers
template <class T, int SIDE, bool FULL, bool BAND>
__device__ __forceinline__ void slab_compute(const T* __restrict__ Ur, const T* __restrict__ Ui, const SlabStrip<T>& d, int n, int w0, int ww, int lane,
                                             const cx<T> (&xa)[8], const cx<T> (&xb)[8]) {
    if constexpr (sizeof(T) == 8) {
        // fp64: 3M product, two of the four output tiles at a time (the streamed operand stays in registers for both passes, the
        // U fragments of the second pass are other columns of the same LDS planes): 48 accumulator registers instead of 64, a
        // quarter fewer MFMAs -- the update sits at the HBM / matrix-core balance point, so this moves it onto the HBM side
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) {
            typename Mfma<T>::acc_t p1[2], p2[2], p3[2];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) { p1[q][r] = T(0); p2[q][r] = T(0); p3[q][r] = T(0); }
            slab_multiply_half_3m<T, SIDE, BAND>(Ur, Ui, 0, pp, lane, xa, p1, p2, p3);
            slab_multiply_half_3m<T, SIDE, BAND>(Ur, Ui, 1, pp, lane, xb, p1, p2, p3);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const T a = p1[q][r], b = p2[q][r], c = p3[q][r];
                    p1[q][r] = SIDE == 1 ? a - b : a + b;               // real part
                    p2[q][r] = SIDE == 1 ? c - a - b : c - a + b;       // imaginary part
                }
            slab_store_pair<T, FULL>(d, n, w0, ww, lane, pp, p1, p2);
        }
        return;
    }
    typename Mfma<T>::acc_t accR[4], accI[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) { accR[q][r] = T(0); accI[q][r] = T(0); }
    slab_multiply_half<T, SIDE>(Ur, Ui, 0, lane, xa, accR, accI);
    slab_multiply_half<T, SIDE>(Ur, Ui, 1, lane, xb, accR, accI);
    slab_store<T>(d, n, w0, ww, lane, accR, accI);
}

// Result register r of tile q:  right update: strip row crow(lane, r), window column 16q + (lane&15);
//                               left update:  window row 16q + crow(lane, r), strip column lane&15.
template <class T>
__device__ __forceinline__ void slab_store(const SlabStrip<T>& d, int n, int w0, int ww, int lane, const typename Mfma<T>::acc_t (&accR)[4],
                                           const typename Mfma<T>::acc_t (&accI)[4]) {
    const int lr = lane & 15;
    char* base = reinterpret_cast<char*>(d.X);       // scalar base + 32-bit byte offsets, as in slab_load_half
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int cr = Mfma<T>::crow(lane, r);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const cx<T> v(accR[q][r], accI[q][r]);
            if (d.side == 1) {
                const int row = d.a0 + cr, k = 16 * q + lr;
                if (row < d.lim && k < ww) *reinterpret_cast<cx<T>*>(base + ((unsigned)row * n + w0 + k) * (unsigned)sizeof(cx<T>)) = v;
            } else {
                const int i = 16 * q + cr, col = d.a0 + lr;
                if (i < ww && col < d.lim) *reinterpret_cast<cx<T>*>(base + ((unsigned)(w0 + i) * n + col) * (unsigned)sizeof(cx<T>)) = v;
            }
        }
    }
}

// Off-window updates of a window step.  With ONE chain per sweep (the default) a single launch does everything (PART 2: the
// regions are disjoint).  With several chains two launches order the left updates (PART 0) before the right updates of H and Z
// (PART 1), see the note at the top.  blockIdx.y = matrix, blockIdx.x = chain * nslab + strip group; a workgroup takes 4*SPW
// consecutive strips of ONE chain (its 4 waves interleaved, so that they stream neighbouring rows).
// SPW = strips per wave (a workgroup covers 4*SPW strips): 1 gives the shortest dependent chain per launch and the most
// workgroups (what matters when several iteration groups keep the GPU busy with small launches); larger values amortise the
// U prologue (64 KiB from L2 per workgroup) over more streamed data.
// PART 2 claims its strips DYNAMICALLY (per-matrix counter QrState::strip_next, reset by the window kernel of the step): the
// latency-bound kernels of the other iteration groups hold whole CUs (133 KB of LDS each), so a launch sized to fill the chip
// exactly would otherwise run a second, nearly empty round of workgroups on the CUs that are left; with dynamic claiming a
// workgroup that starts late finds the counter exhausted and leaves, and the launch ends when the work does.
template <class T, int SPW, int PART>
__global__ __launch_bounds__(256, 2) void apply_window_kernel(cx<T>* __restrict__ Aall, cx<T>* __restrict__ Zall, long mstride, int n,
                                                           QrState* __restrict__ st_all, const cx<T>* __restrict__ Uall,
                                                           unsigned* __restrict__ work, int nslab, int dynamic, int band_on, int wantz) {
    TRX_DYN_SMEM(smem);
    T* Ur = reinterpret_cast<T*>(smem);      // [QW][MLD]
    T* Ui = Ur + QW * MLD;
    const int b = blockIdx.y;
    const int ch = blockIdx.x / nslab, gx = blockIdx.x - ch * nslab;
    const int w0 = st_all[b].w0[ch], w1 = st_all[b].w1[ch];
    const int ww = w1 - w0;
    if (ww <= 0) return;
    const SlabRanges RG = slab_ranges(st_all[b], n, w0, w1, wantz);
    const int nL = RG.nL, nR = RG.nR, nZ = RG.nZ;
    const int S = PART == 0 ? nL : (PART == 1 ? nR + nZ : nL + nR + nZ);
    if (S <= 0) return;
    const int g0 = gx * (4 * SPW);
    if (PART == 2 && dynamic) { if (*(volatile int*)&st_all[b].strip_next >= S) return; }
    else if (g0 >= S) return;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);      // strip descriptors live in scalar registers
    if (PART >= 1 && gx == 0 && t == 0)              // algorithmic work of this chain's update (both parts), in units of 4096 complex MACs
        atomicAdd(work, (unsigned)(((long)ww * ww * (wantz ? 2L * n - ww : (long)(RG.lclim > w1 ? RG.lclim - w1 : 0) + (w0 - RG.rr0))) >> 12));
    const cx<T>* U = Uall + ((long)b * QKC + ch) * QW * QW;
    int dense = 0;                            // any nonzero in the blocks the banded product skips?
    for (int e = t; e < QW * QW; e += 256) {
        const int k = e >> 6, c = e & 63;
        cx<T> u(T(0), T(0));
        if (k < ww && c < ww) u = U[k * QW + c];
        Ur[k * MLD + c] = u.x; Ui[k * MLD + c] = u.y;
        if ((k >> 4) >= (c >> 4) + 2 && (u.x != T(0) || u.y != T(0))) dense = 1;
    }
    int* wdense = reinterpret_cast<int*>(Ui + QW * MLD);          // [4] per-wave votes, behind the planes
    { const int wd = __any(dense); if (lane == 0) wdense[t >> 6] = wd; }
    __syncthreads();
    const bool band = sizeof(T) == 8 && band_on && !(wdense[0] | wdense[1] | wdense[2] | wdense[3]);
    cx<T>* H = Aall + (long)b * mstride;
    cx<T>* Z = Zall + (long)b * mstride;
    if (PART == 2 && dynamic) {
        // dynamic claiming; the next claim is issued before the current strip is processed, so that the round trip of the atomic
        // (1-2 us) hides behind a strip's worth of loads and MFMAs (a wave over-claims once at the end: harmless)
        auto claim = [&]() {
            int g = 0;
            if (lane == 0) g = atomicAdd(&st_all[b].strip_next, 1);
            return __builtin_amdgcn_readfirstlane(g);
        };
        for (int g = claim(); g < S;) {
            const int gn = claim();
            const SlabStrip<T> d = slab_locate<T, PART>(g, RG, H, Z, n, w0);
            if (band) {
                if (d.side == 0) slab_strip<T, 0, true>(Ur, Ui, d, n, w0, ww, lane);
                else slab_strip<T, 1, true>(Ur, Ui, d, n, w0, ww, lane);
            } else {
                if (d.side == 0) slab_strip<T, 0>(Ur, Ui, d, n, w0, ww, lane);
                else slab_strip<T, 1>(Ur, Ui, d, n, w0, ww, lane);
            }
            g = gn;
        }
        return;
    }
    for (int i = 0; i < SPW; ++i) {
        const int g = g0 + wave + 4 * i;
        if (g >= S) break;
        const SlabStrip<T> d = slab_locate<T, PART>(g, RG, H, Z, n, w0);
        const bool left = PART == 0 || (PART == 2 && d.side == 0);
        if (band) {
            if (left) slab_strip<T, 0, true>(Ur, Ui, d, n, w0, ww, lane);
            else slab_strip<T, 1, true>(Ur, Ui, d, n, w0, ww, lane);
        } else {
            if (left) slab_strip<T, 0>(Ur, Ui, d, n, w0, ww, lane);
            else slab_strip<T, 1>(Ur, Ui, d, n, w0, ww, lane);
        }
    }
}

// One 16x16 output tile (tile q of the strip) with the 3M product over the full K = QW, then its store: the unit of work of the
// software-pipelined kernel below.  A single tile keeps only 24 accumulator registers live next to the two 64-register operand
// buffers; the streamed operand is reused from registers by all four tiles, the U fragments come from LDS per tile.
template <class T, int SIDE>
__device__ __forceinline__ void slab_tile_3m(const T* __restrict__ Ur, const T* __restrict__ Ui, const SlabStrip<T>& d, int n, int ws0, int krange, int lane, int q,
                                             const cx<T> (&xa)[8], const cx<T> (&xb)[8]) {
    const int lr = lane & 15, lk = lane >> 4;
    typename Mfma<T>::acc_t p1, p2, p3;
#pragma unroll
    for (int r = 0; r < 4; ++r) { p1[r] = T(0); p2[r] = T(0); p3[r] = T(0); }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int off = (kstep(h, cc, j) + KLS * lk) * MLD + lr + 16 * q;
                const cx<T> xv = h == 0 ? xa[4 * cc + j] : xb[4 * cc + j];
                const T xs = xv.x + xv.y;
                const T ur = Ur[off], ui = Ui[off];
                if (SIDE == 1) {          // C = X U
                    p1 = Mfma<T>::mma(xv.x, ur, p1);
                    p2 = Mfma<T>::mma(xv.y, ui, p2);
                    p3 = Mfma<T>::mma(xs, ur + ui, p3);
                } else {                  // C = U^H X
                    p1 = Mfma<T>::mma(ur, xv.x, p1);
                    p2 = Mfma<T>::mma(ui, xv.y, p2);
                    p3 = Mfma<T>::mma(ur - ui, xs, p3);
                }
            }
            __builtin_amdgcn_sched_barrier(0);       // bounds the hoisting of the U fragment reads (4 k-steps at a time)
        }
    // Cr, Ci from the three products, then the store of the tile (frame indices outside [klo, khi) are identity padding: skipped)
    const int cr0 = Mfma<T>::crow(lane, 0), rs = Mfma<T>::crow(0, 1) - Mfma<T>::crow(0, 0);
    const unsigned esz = (unsigned)sizeof(cx<T>);
    const int klo = krange & 255, khi = krange >> 8;
    char* base = reinterpret_cast<char*>(d.X);
    if (SIDE == 1) {
        const unsigned lane_off = ((unsigned)(d.a0 + cr0) * n + ws0 + lr) * esz;
        const int k = 16 * q + lr;
        const bool okk = k >= klo && k < khi;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const T a = p1[r], bb = p2[r], c = p3[r];
            char* pb = base + ((size_t)(r * rs) * n + 16 * q) * esz;
            if (okk && d.a0 + cr0 + r * rs < d.lim) *reinterpret_cast<cx<T>*>(pb + lane_off) = cx<T>(a - bb, c - a - bb);
        }
    } else {
        const unsigned lane_off = ((unsigned)(ws0 + cr0) * n + d.a0 + lr) * esz;
        const bool okc = d.a0 + lr < d.lim;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const T a = p1[r], bb = p2[r], c = p3[r];
            const int i = 16 * q + cr0 + r * rs;
            char* pb = base + (size_t)(16 * q + r * rs) * n * esz;
            if (okc && i >= klo && i < khi) *reinterpret_cast<cx<T>*>(pb + lane_off) = cx<T>(a + bb, c - a + bb);
        }
    }
}
template <class T, int SIDE>
__device__ __forceinline__ void slab_strip_tiles(const T* __restrict__ Ur, const T* __restrict__ Ui, const SlabStrip<T>& d, int n, int ws0, int krange, int lane,
                                                 const cx<T> (&xa)[8], const cx<T> (&xb)[8]) {
#pragma unroll 1
    for (int q = 0; q < 4; ++q) slab_tile_3m<T, SIDE>(Ur, Ui, d, n, ws0, krange, lane, q, xa, xb);
}

// Software-pipelined variant of the single-launch update (fp64, one chain, n >= 2 QW): the loads of the NEXT strip are issued
// before the current one is multiplied (register double buffer: 2 x 64 operand registers + 48 accumulators of the 3M tile pairs),
// so the matrix cores do not wait out an HBM round trip per strip -- alone on the chip the one-strip-at-a-time kernel ran at 39 %
// of the time its MFMAs need (rocprofv3 timeline: 100 us for a launch whose matrix-core work is 39 us).  To keep ONE code path in
// the register budget every window is treated as a full QW-wide frame: its origin is moved up when it would stick out of the
// matrix (ws0 = min(w0, n - QW)) and U is embedded in an identity of size QW; loads need no clamp on the window index, and the
// stores skip the identity rows / columns (slab_store_pair<FULL>), so nothing outside the real window is rewritten.
template <class T>
__global__ __launch_bounds__(256, 2) void apply_window_pipe_kernel(cx<T>* __restrict__ Aall, cx<T>* __restrict__ Zall, long mstride, int n,
                                                                QrState* __restrict__ st_all, const cx<T>* __restrict__ Uall, unsigned* __restrict__ work, int wantz) {
    TRX_DYN_SMEM(smem);
    T* Ur = reinterpret_cast<T*>(smem);      // [QW][MLD]
    T* Ui = Ur + QW * MLD;
    const int b = blockIdx.y;
    const int w0 = st_all[b].w0[0], w1 = st_all[b].w1[0];
    const int ww = w1 - w0;
    if (ww <= 0) return;
    const SlabRanges RG = slab_ranges(st_all[b], n, w0, w1, wantz);
    const int S = RG.nL + RG.nR + RG.nZ;
    if (S <= 0) return;
    const int t = threadIdx.x, lane = t & 63;
    if (blockIdx.x == 0 && t == 0)
        atomicAdd(work, (unsigned)(((long)ww * ww * (wantz ? 2L * n - ww : (long)(RG.lclim > w1 ? RG.lclim - w1 : 0) + (w0 - RG.rr0))) >> 12));
    const int ws0 = w0 < n - QW ? w0 : n - QW;          // origin of the QW-wide frame
    const int sh = w0 - ws0;                            // the real window sits at frame indices [sh, sh + ww)
    const int krange = sh | ((sh + ww) << 8);
    const cx<T>* U = Uall + (long)b * QKC * QW * QW;
    for (int e = t; e < QW * QW; e += 256) {
        const int k = e >> 6, c = e & 63;
        const int kk = k - sh, cc = c - sh;
        cx<T> u(k == c ? T(1) : T(0), T(0));
        if (kk >= 0 && kk < ww && cc >= 0 && cc < ww) u = U[kk * QW + cc];
        Ur[k * MLD + c] = u.x; Ui[k * MLD + c] = u.y;
    }
    __syncthreads();
    cx<T>* H = Aall + (long)b * mstride;
    cx<T>* Z = Zall + (long)b * mstride;
    // STATIC strip assignment here (wave w of workgroup x takes strips x*4*spw + w + 4 i): a dynamic claim is an atomic with return,
    // and hipcc waits for it with vmcnt(0) right where it is issued (its wave-level atomic optimiser reads the result back at once),
    // which also waits out every prefetch load in flight -- seen in the ISA, and measured: no gain from the pipeline with it.  The
    // prefetch is unconditional (a wave past its last strip re-reads that strip and discards it), so there is no branch between
    // the issue of the loads and the MFMAs that hide them; the waits in front of the MFMAs are vmcnt(16..25), i.e. they leave the
    // 16 loads of the next strip in flight.
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int spw = (S + 4 * (int)gridDim.x - 1) / (4 * (int)gridDim.x);          // strips per wave
    const int gbase = blockIdx.x * 4 * spw + wave;
    const int gend = (blockIdx.x + 1) * 4 * spw < S ? (blockIdx.x + 1) * 4 * spw : S;     // this workgroup's strips: [blockIdx.x*4*spw, gend)
    if (gbase >= gend) return;
    SlabStrip<T> d0 = slab_locate<T, 2>(gbase, RG, H, Z, n, w0), d1 = d0;
    cx<T> xa0[8], xb0[8], xa1[8], xb1[8];
    slab_load_half<T, true>(d0, 0, n, ws0, QW, lane, xa0);
    slab_load_half<T, true>(d0, 1, n, ws0, QW, lane, xb0);
    for (int g = gbase;; g += 8) {
        const int gb = g + 4, gc = g + 8;
        d1 = slab_locate<T, 2>(gb < gend ? gb : g, RG, H, Z, n, w0);
        slab_load_half<T, true>(d1, 0, n, ws0, QW, lane, xa1);
        slab_load_half<T, true>(d1, 1, n, ws0, QW, lane, xb1);
        __builtin_amdgcn_sched_barrier(0);
        if (d0.side == 0) slab_strip_tiles<T, 0>(Ur, Ui, d0, n, ws0, krange, lane, xa0, xb0);
        else slab_strip_tiles<T, 1>(Ur, Ui, d0, n, ws0, krange, lane, xa0, xb0);
        if (gb >= gend) break;
        d0 = slab_locate<T, 2>(gc < gend ? gc : gb, RG, H, Z, n, w0);
        slab_load_half<T, true>(d0, 0, n, ws0, QW, lane, xa0);
        slab_load_half<T, true>(d0, 1, n, ws0, QW, lane, xb0);
        __builtin_amdgcn_sched_barrier(0);
        if (d1.side == 0) slab_strip_tiles<T, 0>(Ur, Ui, d1, n, ws0, krange, lane, xa1, xb1);
        else slab_strip_tiles<T, 1>(Ur, Ui, d1, n, ws0, krange, lane, xa1, xb1);
        if (gc >= gend) break;
    }
}

template <class T>
__global__ void qr_collect_info_kernel(const QrState* __restrict__ st, int* __restrict__ info, int batch, int ngroups) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;       // matrix b belongs to group b % ngroups, at position b / ngroups of it
    if (b >= batch) return;
    const int g = b % ngroups, local = b / ngroups;
    int b0 = 0;
    for (int h = 0; h < g; ++h) b0 += (batch - h + ngroups - 1) / ngroups;       // state slots are group-contiguous
    info[b] = st[b0 + local].fail;
}

}  // namespace

// ---- host-side runtime shared by all calls: tuning knobs resolved ONCE, internal streams / events pooled --------------------
struct QrKnobs {
    int groups = 0, spw = 0, aed = 0, nibble = 100, moves = QAED_MOVES, chains = 0, dyn = 0, wgs = 0, pipe = 0, band = 0, rotb = 0;      // dyn: 0 auto, 1 static strips, 2 dynamic; wgs: workgroups per slab launch
    bool debug = false;
};
static QrKnobs& qr_knobs() {
    static QrKnobs k = [] {
        QrKnobs q;
        auto geti = [](const char* name, int lo, int hi, int dflt) {
            const char* e = getenv(name);
            if (!e) return dflt;
            const int v = atoi(e);
            return (v >= lo && v <= hi) ? v : dflt;
        };
        q.groups = geti("TRX_QR_GROUPS", 1, 8, 0);            // 0 = by batch size
        q.spw = geti("TRX_SLAB_SPW", 1, 4, 0);
        if (q.spw == 3) q.spw = 0;
        q.aed = geti("TRX_QR_AED", 16, QAED, 0);
        q.nibble = geti("TRX_QR_NIBBLE", 0, 100, 100);
        q.moves = geti("TRX_QR_MOVES", 0, QAED, QAED_MOVES);
        q.chains = geti("TRX_QR_CHAINS", 1, QKC, 0);
        q.dyn = geti("TRX_SLAB_DYN", 0, 2, 0);
        q.wgs = geti("TRX_SLAB_WGS", 32, 4096, 0);
        q.pipe = geti("TRX_SLAB_PIPE", 0, 2, 0);
        q.band = geti("TRX_SLAB_BAND", 0, 2, 0);              // 0 / 2: skip the structurally zero blocks of a chain unitary, 1: dense product always
        q.rotb = geti("TRX_QR_ROTB", 0, 2, 0);               // 1: rotations of the in-LDS Schur solver broadcast by ds_bpermute (round-3 code), else v_readlane
        q.debug = getenv("TRX_QR_DEBUG") != nullptr;
        return q;
    }();
    return k;
}

// Non-blocking streams and timing-less events for the iteration groups, created on first use and kept for the life of the
// process (per device); a call checks out what it needs and hands it back, so concurrent callers never share one.
struct QrLane {
    hipStream_t s = nullptr;
    hipEvent_t ev = nullptr;           // fork / join / start stagger
    hipEvent_t evs[2] = {nullptr, nullptr};   // "summary of iteration k has landed in hsum[k % 2]"
    int* hsum = nullptr;               // pinned host memory, 2 x 4 ints
    int dev = -1;
    bool has_stream = false;
};
static std::mutex g_lane_mu;
static std::vector<QrLane> g_lane_free;
static bool lane_checkout(int dev, bool want_stream, QrLane& out) {
    {
        std::lock_guard<std::mutex> lock(g_lane_mu);
        for (size_t i = 0; i < g_lane_free.size(); ++i)
            if (g_lane_free[i].dev == dev && g_lane_free[i].has_stream == want_stream) { out = g_lane_free[i]; g_lane_free.erase(g_lane_free.begin() + i); return true; }
    }
    out = QrLane();
    out.dev = dev;
    out.has_stream = want_stream;
    if (want_stream && hipStreamCreateWithFlags(&out.s, hipStreamNonBlocking) != hipSuccess) return false;
    if (hipEventCreateWithFlags(&out.ev, hipEventDisableTiming) != hipSuccess) return false;
    if (hipEventCreateWithFlags(&out.evs[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&out.evs[1], hipEventDisableTiming) != hipSuccess) return false;
    if (hipHostMalloc((void**)&out.hsum, sizeof(int) * 8, hipHostMallocDefault) != hipSuccess) return false;
    return true;
}
static void lane_return(const QrLane& l) {
    std::lock_guard<std::mutex> lock(g_lane_mu);
    g_lane_free.push_back(l);
}

// trx_tuning(): the environment variables only provide the defaults (read once); this sets a knob explicitly.  0 = automatic.
int qr_set_knob(const char* key, int value) {
    QrKnobs& k = qr_knobs();
    const std::string s(key);
    int* slot = nullptr;
    int lo = 0, hi = 0;
    if (s == "qr_groups") { slot = &k.groups; hi = 8; }
    else if (s == "slab_spw") { slot = &k.spw; hi = 4; }
    else if (s == "qr_aed") { slot = &k.aed; hi = QAED; }
    else if (s == "qr_nibble") { slot = &k.nibble; hi = 100; }
    else if (s == "qr_moves") { slot = &k.moves; hi = QAED; }
    else if (s == "qr_rotb") { slot = &k.rotb; hi = 2; }
    else if (s == "qr_chains") { slot = &k.chains; hi = QKC; }
    else if (s == "slab_dyn") { slot = &k.dyn; hi = 2; }
    else if (s == "slab_wgs") { slot = &k.wgs; hi = 4096; }
    else if (s == "slab_pipe") { slot = &k.pipe; hi = 2; }
    else if (s == "slab_band") { slot = &k.band; hi = 2; }
    else return TRX_ERR_ARG;
    if (value < lo || value > hi || (slot == &k.spw && value == 3) || (slot == &k.aed && value != 0 && value < 16)) return TRX_ERR_ARG;
    *slot = value;
    return TRX_OK;
}

template <class T>
int hessenberg_qr(hipStream_t s, const EigBuffers<T>& B, int n, int batch, int* info, int wantz) {
    constexpr int LD = QW + 1;
    if ((double)n * n * sizeof(cx<T>) >= 4294967296.0) return TRX_ERR_ARG;      // slab kernel: 32-bit byte offsets inside one matrix
    const QrKnobs& K = qr_knobs();
    const size_t smw = sizeof(cx<T>) * QW * LD + sizeof(RotCS<T>) * WMAXS * QNS + sizeof(QrState);
    const size_t sma = sizeof(T) * 2 * QW * MLD + 16;        // + the four per-wave band votes
    auto smp_of = [](int sm) { return sizeof(cx<T>) * (2 * (size_t)sm * (sm + 1) + 2 * sm) + sizeof(Rot<T>) * sm + sizeof(QrState); };
    // opt-in to > 64 KB of dynamic LDS: per (device, dtype), once; a failure is remembered so that no later call launches anyway
    static std::mutex attr_mu;
    static int attr_state[64][2];        // 0 = not yet set, 1 = ok, 2 = failed
    int attr_rc = 0, dev_attr = 0;
    (void)hipGetDevice(&dev_attr);
    {
        std::lock_guard<std::mutex> lock(attr_mu);
        int& stt = attr_state[dev_attr & 63][sizeof(T) == 8];
        if (stt == 0) {
            const int r = set_max_dyn_smem((const void*)qr_window_kernel<T, false>, smw) || set_max_dyn_smem((const void*)qr_window_kernel<T, true>, smw) || set_max_dyn_smem((const void*)apply_window_kernel<T, 1, 0>, sma) ||
                  set_max_dyn_smem((const void*)apply_window_kernel<T, 1, 1>, sma) || set_max_dyn_smem((const void*)apply_window_kernel<T, 2, 0>, sma) ||
                  set_max_dyn_smem((const void*)apply_window_kernel<T, 2, 1>, sma) || set_max_dyn_smem((const void*)apply_window_kernel<T, 4, 0>, sma) ||
                  set_max_dyn_smem((const void*)apply_window_kernel<T, 4, 1>, sma) || set_max_dyn_smem((const void*)apply_window_kernel<T, 1, 2>, sma) ||
                  set_max_dyn_smem((const void*)apply_window_kernel<T, 2, 2>, sma) || set_max_dyn_smem((const void*)apply_window_kernel<T, 4, 2>, sma) || set_max_dyn_smem((const void*)apply_window_pipe_kernel<T>, sma) ||
                  set_max_dyn_smem((const void*)qr_prepare_kernel<T, false>, smp_of(SM)) || set_max_dyn_smem((const void*)qr_prepare_kernel<T, true>, smp_of(SM));
            stt = r ? 2 : 1;
        }
        attr_rc = stt == 2;
    }
    if (attr_rc) return TRX_ERR_LAUNCH;
    TRX_LAUNCH((qr_init_kernel<T>), dim3(batch), dim3(64), 0, s, B.st, n);
    if (hipMemsetAsync(B.summary, 0, sizeof(int) * 64, s) != hipSuccess) return TRX_ERR_LAUNCH;
    const int max_sweeps = 30 * n + 100;
    // strips per wave of the slab kernel: small batches are latency bound and keep the shorter per-launch chain
    const int spw = K.spw ? K.spw : (batch >= 64 ? 4 : 2);
    const int nstrip = cdiv_i(n, 16);
    const int nslabL = cdiv_i(nstrip + 1, 4 * spw);          // workgroups per matrix and chain: left strips <= n/16 + 1
    const int nslabR = cdiv_i(2 * nstrip + 1, 4 * spw);      // right-H strips <= n/16 + 1, Z strips = n/16
    const int adv = QW - 2 * QNS - 1;                        // guaranteed chain advance per window step
    // Bulge chains per sweep.  Measured on MI355X (n = 1922): 2 / 3 chains cut the outer iterations by only 31 / 36 % (the AED's
    // deflation yield, not the shift count, paces the iteration) while the slab work grows by a third, and the two-launch
    // update they need costs 10 % on its own: 26.5 (1 chain, one launch) vs 22.1 / 21.9 solves/s at batch 128, 12.1 vs 11.3 at
    // batch 16.  One chain is the default for batches; a single large matrix (the topology-optimisation case, n = 5202, batch 1) has
    // no slab work to protect and gains from the shorter chain: 5.62 s (1 chain, AED 48) -> 5.16 s (3 chains) -> 4.79 s (3 chains,
    // AED 64) for the whole forward solve (profiles/r02_single_matrix_knobs.txt).
    const int kc = K.chains ? K.chains : (batch <= 2 ? QKC : 1);
    // software-pipelined slab kernel: knob slab_pipe = 2 switches it on.  Measured on MI355X at batch 128 it is NOT faster than the
    // one-strip-at-a-time kernel with dynamically claimed strips (27.4-28.0 vs 28.4 solves/s): a launch of the latter already
    // overlaps the loads of one wave with the MFMAs of its SIMD neighbour, and the static strips the pipeline needs bring back
    // the tail of a launch that dynamic claiming removes.  Kept as an option (tests/test_eig.py runs both).
    const bool pipe = sizeof(T) == 8 && n >= 2 * QW && K.pipe == 2;
    const int band_on = K.band != 1;

    // The batch is split into groups that iterate out of phase on their own streams: the latency-bound kernels of one group
    // (AED / shift preparation: one wave per matrix; window chase: one workgroup per matrix and chain) run while the slab updates
    // of the other groups fill the matrix cores.  One host thread drives all of them round-robin; per visit it reads the 16-byte
    // summary of the group's last prepare (normally long finished: the prepare is queued right behind the group's sweep),
    // queues the next sweep and the prepare after it, and moves on.
    constexpr int MAXG = 8;
    struct Group {
        QrLane lane;           // stream (null: the caller's) + event
        hipStream_t s;
        int b0, nb;
        int* summary;          // device, 8 ints: two slots {[0] active matrices, [1] bound of the remaining blocks, [2] flags}; [3] slab work (cumulative)
        bool done;
        unsigned work;
        int par;               // parity of the next window step (double-buffered chase positions)
        int g;                 // group index = index of its first matrix
        int issued, read;      // outer iterations queued / summaries read
    };
    // 4 groups = the number of hardware queues a HIP process gets by default; beyond that streams share queues and serialise
    int ngroups = K.groups ? K.groups : (batch >= 64 ? 4 : (batch >= 8 ? 2 : 1));
    if (ngroups > batch) ngroups = batch;
    Group grp[MAXG];
    int rc = TRX_OK, dev = 0, nlanes = 0;
    (void)hipGetDevice(&dev);
    QrLane fork;
    if (ngroups > 1) {
        if (!lane_checkout(dev, false, fork) || hipEventRecord(fork.ev, s) != hipSuccess) return TRX_ERR_LAUNCH;
    }
    for (int g = 0; g < ngroups; ++g) {
        Group& G = grp[g];
        // Group g takes the matrices g, g + ngroups, g + 2 ngroups, ... (NOT a contiguous block): the cost of a matrix varies
        // smoothly along a sweep (wavelength, geometry), so contiguous blocks finish at very different times and the last
        // group runs alone; interleaved, every group sees the same mix.  State, U and shift slots stay group-contiguous (b0).
        G.nb = (batch - g + ngroups - 1) / ngroups;
        G.b0 = g == 0 ? 0 : grp[g - 1].b0 + grp[g - 1].nb;
        G.summary = B.summary + 8 * g;
        G.done = false;
        G.issued = 0;
        G.read = 0;
        G.work = 0;
        G.par = 0;
        G.g = g;
        if (!lane_checkout(dev, g > 0, G.lane)) { rc = TRX_ERR_LAUNCH; break; }
        ++nlanes;
        G.s = g == 0 ? s : G.lane.s;
        if (g > 0 && hipStreamWaitEvent(G.s, fork.ev, 0) != hipSuccess) { rc = TRX_ERR_LAUNCH; break; }
    }
    // AED window: 64 deflates most per call (fewest sweeps, least slab work) but costs 3 ms of single-wave latency; at small
    // batches, where nothing is throughput bound, a smaller window shortens the chain
    const long mstride = (long)ngroups * n * n;              // distance between consecutive matrices of one group
    const int aed_w = K.aed ? K.aed : ((batch >= 64 || batch <= 2) ? QAED : 48);     // measured: batch 128: 64 -> 28.4, 48 -> 27.8 solves/s; batch 16: 48; batch 1 with 3 chains: 64
    const size_t smp = smp_of(aed_w);
    // LAPACK skips the sweep when AED deflated more than 14 % of the window ("nibble") because there the sweep is the expensive
    // part.  Here the AED is, so every AED that leaves an active block is followed by a sweep in the same iteration.
    const int nibble = K.nibble, aed_moves = K.moves;
    const bool qr_debug = K.debug;                                        // cycle breakdown of the prepare / window kernels (matrix 0) to stderr
    long long* dbg_dev = reinterpret_cast<long long*>(B.summary + 64);    // 24 counters behind the group summaries
    // One outer iteration of a group = the window steps of its sweep followed by the next prepare (deflation scan, AED, shifts)
    // and the copy of that prepare's 16-byte summary into pinned host memory.  The host runs ONE ITERATION AHEAD of what it has
    // read: iteration k+1 is queued with a step count from summary k-1 -- summary[1] is an upper bound (ihi + 1) for every block
    // the matrices of the group can still work on, and it only shrinks; a step a chain does not need is a kernel that exits at
    // once.  So a stream never drains while the host is busy with another group (with one queued iteration per group and a
    // blocking read the groups ran mostly ONE AT A TIME: rocprofv3 timeline, profiles/).
    auto issue_prepare = [&](Group& G, int slot) -> bool {
        int* sum = G.summary + 4 * slot;
        if (hipMemsetAsync(sum, 0, sizeof(int) * 3, G.s) != hipSuccess) return false;            // G.summary[3] (slab work) keeps accumulating
        { ProfScope prof(PROF_QR_PREPARE, G.s, 0, 0);
          if (K.rotb == 1)
              TRX_LAUNCH((qr_prepare_kernel<T, false>), dim3(G.nb), dim3(64), smp, G.s, B.A + (long)G.g * n * n, mstride, n, B.st + G.b0, B.U + (long)G.b0 * QKC * QW * QW,
                         B.shifts + (long)G.b0 * QKC * QNS, sum, max_sweeps, aed_w, nibble, aed_moves, G.par, kc, aed_w, wantz,
                         (qr_debug && G.b0 == 0) ? dbg_dev : (long long*)nullptr);
          else
              TRX_LAUNCH((qr_prepare_kernel<T, true>), dim3(G.nb), dim3(64), smp, G.s, B.A + (long)G.g * n * n, mstride, n, B.st + G.b0, B.U + (long)G.b0 * QKC * QW * QW,
                         B.shifts + (long)G.b0 * QKC * QNS, sum, max_sweeps, aed_w, nibble, aed_moves, G.par, kc, aed_w, wantz,
                         (qr_debug && G.b0 == 0) ? dbg_dev : (long long*)nullptr); }
        if (hipMemcpyAsync(G.lane.hsum + 4 * slot, sum, sizeof(int) * 3, hipMemcpyDeviceToHost, G.s) != hipSuccess) return false;
        return hipEventRecord(G.lane.evs[slot], G.s) == hipSuccess;
    };
    auto issue_sweep = [&](Group& G, int bound) {
        cx<T>* Ag = B.A + (long)G.g * n * n;
        cx<T>* Zg = B.Z + (long)G.g * n * n;
        cx<T>* Ug = B.U + (long)G.b0 * QKC * QW * QW;
        const cx<T>* shg = B.shifts + (long)G.b0 * QKC * QNS;
        QrState* stg = B.st + G.b0;
        // window steps: the first chain needs (m + 2 QNS) / adv steps (+ one slot that applies the AED unitary); every further chain
        // enters about 3 steps behind the one ahead.  `bound` is one iteration old, i.e. already a step or so generous, and a sweep
        // that still falls short is finished by the next iteration's steps (flag 2 of the summary).
        const int nwin = bound > 0 ? cdiv_i(bound + 2 * QNS, adv) + 2 + 4 * (kc - 1) : 1;
        unsigned* wk = (unsigned*)(G.summary + 3);
        for (int q = 0; q < nwin; ++q) {
            { ProfScope p(PROF_QR_WINDOW, G.s, 0, 0);
              if (qr_debug && G.b0 == 0) TRX_LAUNCH((qr_window_kernel<T, true>), dim3(kc, G.nb), dim3(WTHREADS), smw, G.s, Ag, mstride, n, stg, Ug, shg, G.par, dbg_dev);
              else TRX_LAUNCH((qr_window_kernel<T, false>), dim3(kc, G.nb), dim3(WTHREADS), smw, G.s, Ag, mstride, n, stg, Ug, shg, G.par, (long long*)nullptr); }
            G.par ^= 1;
            { ProfScope p(PROF_QR_APPLY_RIGHT, G.s, 0, 0);
              // single-launch variant: strips are claimed dynamically, so the workgroup count per matrix only has to fill the chip
              // (about two workgroups per CU over the group), whatever the group size
              // (groups of fewer than 16 matrices are latency bound and nothing competes for their CUs: static strips, measured faster)
              const int dyn = K.dyn ? K.dyn - 1 : (G.nb >= 16);
              int wgm = cdiv_i(2 * nstrip + 2, 4 * spw);
              if (dyn) {
                  wgm = (K.wgs ? K.wgs : 512) / G.nb;
                  wgm = wgm < 1 ? 1 : (wgm > 32 ? 32 : wgm);      // > 32 per matrix: the 64 KB U prologue of each workgroup dominates (measured)
                  if (wgm > cdiv_i(2 * nstrip + 2, 4)) wgm = cdiv_i(2 * nstrip + 2, 4);
              }
              const dim3 gl(kc * nslabL, G.nb), gr(kc * nslabR, G.nb), ga(wgm, G.nb);
              if (kc == 1 && dyn && pipe) {
                  if constexpr (sizeof(T) == 8) TRX_LAUNCH((apply_window_pipe_kernel<T>), ga, dim3(256), sma, G.s, Ag, Zg, mstride, n, stg, (const cx<T>*)Ug, wk, wantz);
              } else if (kc == 1) {
                  if (spw == 1) TRX_LAUNCH((apply_window_kernel<T, 1, 2>), ga, dim3(256), sma, G.s, Ag, Zg, mstride, n, stg, (const cx<T>*)Ug, wk, wgm, dyn, band_on, wantz);
                  else if (spw == 2) TRX_LAUNCH((apply_window_kernel<T, 2, 2>), ga, dim3(256), sma, G.s, Ag, Zg, mstride, n, stg, (const cx<T>*)Ug, wk, wgm, dyn, band_on, wantz);
                  else TRX_LAUNCH((apply_window_kernel<T, 4, 2>), ga, dim3(256), sma, G.s, Ag, Zg, mstride, n, stg, (const cx<T>*)Ug, wk, wgm, dyn, band_on, wantz);
              } else if (spw == 1) {
                  TRX_LAUNCH((apply_window_kernel<T, 1, 0>), gl, dim3(256), sma, G.s, Ag, Zg, mstride, n, stg, (const cx<T>*)Ug, wk, nslabL, 0, band_on, wantz);
                  TRX_LAUNCH((apply_window_kernel<T, 1, 1>), gr, dim3(256), sma, G.s, Ag, Zg, mstride, n, stg, (const cx<T>*)Ug, wk, nslabR, 0, band_on, wantz);
              } else if (spw == 2) {
                  TRX_LAUNCH((apply_window_kernel<T, 2, 0>), gl, dim3(256), sma, G.s, Ag, Zg, mstride, n, stg, (const cx<T>*)Ug, wk, nslabL, 0, band_on, wantz);
                  TRX_LAUNCH((apply_window_kernel<T, 2, 1>), gr, dim3(256), sma, G.s, Ag, Zg, mstride, n, stg, (const cx<T>*)Ug, wk, nslabR, 0, band_on, wantz);
              } else {
                  TRX_LAUNCH((apply_window_kernel<T, 4, 0>), gl, dim3(256), sma, G.s, Ag, Zg, mstride, n, stg, (const cx<T>*)Ug, wk, nslabL, 0, band_on, wantz);
                  TRX_LAUNCH((apply_window_kernel<T, 4, 1>), gr, dim3(256), sma, G.s, Ag, Zg, mstride, n, stg, (const cx<T>*)Ug, wk, nslabR, 0, band_on, wantz);
              } }
        }
    };
    if (!rc && qr_debug && hipMemsetAsync(dbg_dev, 0, sizeof(long long) * 24, s) != hipSuccess) rc = TRX_ERR_LAUNCH;
    // iteration 0 = the first prepare (chained: group g starts when group g-1 has finished its own, so that the groups start out of
    // phase); iteration 1 = the first sweep, queued at once with the full matrix as its bound
    for (int g = 0; g < ngroups && !rc; ++g) {
        Group& G = grp[g];
        if (g > 0 && hipStreamWaitEvent(G.s, grp[g - 1].lane.ev, 0) != hipSuccess) rc = TRX_ERR_LAUNCH;
        if (!rc && !issue_prepare(G, 0)) rc = TRX_ERR_LAUNCH;
        if (!rc && ngroups > 1 && hipEventRecord(G.lane.ev, G.s) != hipSuccess) rc = TRX_ERR_LAUNCH;
        G.issued = 1;
        G.read = 0;
    }
    for (int g = 0; g < ngroups && !rc; ++g) {
        Group& G = grp[g];
        issue_sweep(G, n);
        if (!issue_prepare(G, 1)) rc = TRX_ERR_LAUNCH;
        G.issued = 2;
    }
    // The host serves whichever group's summary has landed (event query), never a fixed round-robin with a blocking wait: a group
    // the GPU happened to favour would otherwise drain its queue and idle until the host had waited out the slower ones.
    int live = rc ? 0 : ngroups;
    long served = 0;
    int idle_passes = 0;
    while (live > 0 && served < (long)ngroups * (64L * n + 1000)) {
        bool any = false;
        for (int g = 0; g < ngroups && !rc; ++g) {
            Group& G = grp[g];
            if (G.done) continue;
            const int slot = G.read & 1;
            const hipError_t q = hipEventQuery(G.lane.evs[slot]);
            if (q == hipErrorNotReady) continue;
            if (q != hipSuccess) { rc = TRX_ERR_LAUNCH; break; }
            any = true;
            ++served;
            const int active = G.lane.hsum[4 * slot], bound = G.lane.hsum[4 * slot + 1];
            ++G.read;
            if (active == 0) { G.done = true; --live; continue; }       // (the iteration already queued behind it finds nothing to do)
            // summary G.read-1 is in: queue iteration G.issued (slot parity = G.issued & 1 = slot, free again now)
            issue_sweep(G, bound);
            if (!issue_prepare(G, G.issued & 1)) { rc = TRX_ERR_LAUNCH; break; }
            ++G.issued;
        }
        if (rc) break;
        if (any) { idle_passes = 0; continue; }
        // nothing ready: block on the group that is furthest behind (its summary is the next to arrive, by and large)
        if (++idle_passes < 64) { std::this_thread::yield(); continue; }
        idle_passes = 0;
        int gmin = -1;
        for (int g = 0; g < ngroups; ++g)
            if (!grp[g].done && (gmin < 0 || grp[g].read < grp[gmin].read)) gmin = g;
        if (gmin >= 0 && hipEventSynchronize(grp[gmin].lane.evs[grp[gmin].read & 1]) != hipSuccess) { rc = TRX_ERR_LAUNCH; break; }
    }
    (void)hipGetLastError();          // hipEventQuery leaves hipErrorNotReady as the thread's last error
    double work = 0;
    for (int g = 0; g < nlanes; ++g) {
        Group& G = grp[g];
        if (!rc && prof_enabled() && hipMemcpyAsync(&G.work, G.summary + 3, sizeof(unsigned), hipMemcpyDeviceToHost, G.s) == hipSuccess && hipStreamSynchronize(G.s) == hipSuccess)
            work += G.work;
        if (g > 0) {
            // join: the caller's stream waits for everything queued on the group's stream; a pooled stream goes back idle
            if (hipEventRecord(G.lane.ev, G.s) != hipSuccess || hipStreamWaitEvent(s, G.lane.ev, 0) != hipSuccess) rc = rc ? rc : TRX_ERR_LAUNCH;
            if (hipStreamSynchronize(G.s) != hipSuccess) rc = rc ? rc : TRX_ERR_LAUNCH;
        } else if (G.issued > 0) {
            // the host loop runs one iteration ahead: the last iteration queued on the CALLER's stream (empty sweep, prepare, summary
            // copy into this lane's pinned slot, event) may still be pending.  The lane must not go back to the pool before it has
            // landed -- a concurrent trx_eig on another host thread could check it out and read the stale summary as its own.
            if (hipEventSynchronize(G.lane.evs[(G.issued - 1) & 1]) != hipSuccess) rc = rc ? rc : TRX_ERR_LAUNCH;
        }
        lane_return(G.lane);
    }
    if (ngroups > 1) lane_return(fork);
    if (rc) return rc;
    if (qr_debug) {
        long long h[24];
        if (hipMemcpyAsync(h, dbg_dev, sizeof(h), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess)
            fprintf(stderr, "libtrx qr_prepare cycles (matrix 0, %lld AED calls): scan+load %lld | schur %lld (pre %lld left %lld right %lld U %lld; %lld iterations, %lld rotations) | "
                            "reorder %lld | restore %lld | store %lld || window kernel: load %lld chase %lld (%lld chain steps) store %lld; per phase: rotg %lld left %lld barrier %lld right %lld barrier %lld\n",
                    h[11], h[6], h[7], h[0], h[1], h[2], h[3], h[4], h[5], h[8], h[9], h[10], h[12], h[13], h[15], h[14], h[16], h[17], h[18], h[19], h[20]);
    }
    if (prof_enabled()) prof_add_work(PROF_QR_APPLY_RIGHT, 8.0 * 4096.0 * work, 0.0);
    TRX_LAUNCH((qr_collect_info_kernel<T>), dim3(cdiv_i(batch, 64)), dim3(64), 0, s, (const QrState*)B.st, info, batch, ngroups);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

template int hessenberg_qr<float>(hipStream_t, const EigBuffers<float>&, int, int, int*, int);
template int hessenberg_qr<double>(hipStream_t, const EigBuffers<double>&, int, int, int*, int);

}  // namespace trx
