// Batched small-bulge multi-shift QR iteration on upper Hessenberg matrices, H = Z T Z^H (Schur form), with the
// unitary accumulated into Z.  Second stage of the replacement for torch.linalg.eig (torcwa/torch_eig.py:14).
//
// Design for MI355X.  The sequential part of the QR algorithm (generating rotations) only ever touches a narrow
// diagonal window, so it runs as ONE workgroup per matrix entirely out of LDS (qr_window_kernel: a chain of up to
// QNS single-shift bulges, 2 rows apart, is chased through a QW x QW window while the window's unitary U is
// accumulated in LDS).  Everything off the window is updated with U by wide, embarrassingly parallel kernels on the
// matrix cores (apply_links_kernel).  Per-matrix progress (active block, shifts, chase position) lives in device memory,
// so one fixed launch schedule serves the whole batch; matrices that have nothing to do in a step see an empty window and exit.
//
//   qr_prepare_kernel  (1 wave / matrix)  deflation scan, active-block bookkeeping, aggressive early deflation on the
//                                         trailing window (in-LDS single-shift QR), shifts for up to QKC chains; blocks
//                                         <= QNMIN are finished here by the same in-LDS QR with U accumulated.
//   qr_window_kernel   (1 workgroup per matrix AND chain) one window step of each bulge chain; the window unitary and
//                                         its position go into the LINK LOG of the sweep (slot = window step).
//   apply_links_kernel<0>   per window step:  H[w0:w1, w1:n) <- U^H H[w0:w1, w1:n)   -- the left update, the only part the NEXT
//                                         window step depends on (its new columns);
//   apply_links_kernel<1>   ONCE per sweep:   H[0:w0, w0:w1) <- H[0:w0, w0:w1) U,  Z[:, w0:w1) <- Z[:, w0:w1) U  for every link of the
//                                         log, in order.  No window step of the same sweep reads what these touch (rows ABOVE a window),
//                                         left and right multiplications commute, so two thirds of the off-window work leave the
//                                         window -> update -> window chain and run as one large launch per sweep, each row strip walking
//                                         through the links with its 16 x 64 block hot in L2.
//
// Several bulge chains per sweep.  A sweep sends up to QKC chains of QNS shifts each down the active block, chain c following
// chain c-1 at a distance of at least one window (a chain moves only if its new window ends above the last bulge of the chain
// ahead), so that one window step advances all of them: their windows are disjoint diagonal blocks (independent workgroups), the
// left updates of different chains touch disjoint rows and the right updates disjoint columns.  A block that two chains both
// reach (rows of the upper window x columns of the lower one) gets U_a^H from the left and U_b from the right -- the two commute.
// Up to 48 of the AED window's eigenvalues are thus used per sweep instead of 16 (default for one or two matrices).
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
// chain step.  A plain v_rsq_f32 is cheaper still but leaves the rotations unitary to only a few fp32 ulp.  fp32 takes ONE unrefined v_rsq_f64
// of the fp64 argument: no under / overflow of the product |f|^2 d^2, five dependent instructions.  Its accuracy is NOT claimed beyond what
// the ISA documents for v_rsq_f64 (about single precision, ~1e-7 relative -- the CPU emulator computes an exact 1/sqrt and cannot vouch for the
// hardware): the rotations are therefore unitary to fp32 rounding, no better, and that is all the fp32 stage needs -- its results go through
// Newton refinement in fp64 (mixed route) or are a precision="native" answer held to the fp32 gates of the GPU suite (residual 5e-6,
// tests/test_eig.py::test_eig_random[complex64], test_eig_super_steps_fp32).
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
    // 1 / (|f| d) from ONE rsqrt of |f|^2 d^2, the product formed in fp64 (in fp32 it could underflow).  Outside a safe range of the product
    // -- the result would overflow the working precision, or the product itself left the fp64 range -- both inputs are rescaled by a power of
    // two (exact) first.  Rare but real: inside a large cluster of (nearly) equal eigenvalues -- 36 of them 3e-9 apart -- the AED's reordering
    // meets |f| ~ 1e-100 next to |g| ~ 1e-60 (fp32: 1e-25 / 1e-15); the product underflowed, rsqrt(0) = inf, and the NaN spread over the whole
    // matrix with info = n (rounds 1 - 5; found in round 6 through the fp64 fallback of the mixed route).
    const double prod = (double)af2 * (double)n2;
    const double plo = sizeof(T) == 8 ? 1e-290 : 1e-60, phi = sizeof(T) == 8 ? 1e290 : 1e60;
    const bool unsafe = !(prod >= plo && prod <= phi);              // (decided beside the rsqrt chain; the branch sits behind the fast formulas)
    const T tu = sizeof(T) == 8 ? (T)fast_rsqrt(prod) : (T)fast_rsqrt_f64arg(prod);
    R.c = af2 * tu;                                // |f| / d
    R.s = tu * (f * conj(g));                      // (f/|f|) conj(g) / d
    R.r = (n2 * tu) * f;                           // (f/|f|) d
    if (__builtin_expect(unsafe, 0)) {
        const T big = fmax(fmax(fabs(f.x), fabs(f.y)), fmax(fabs(g.x), fabs(g.y)));
        if (!(big < std::numeric_limits<T>::infinity())) { R.c = T(1); R.s = cx<T>(T(0), T(0)); R.r = f; return R; }       // non-finite input: left to the callers' checks
        int ex;
        (void)frexp((double)big, &ex);
        const T sc = (T)ldexp(1.0, -ex), isc = (T)ldexp(1.0, ex);
        const cx<T> fs = sc * f, gs = sc * g;
        const T a2 = norm2(fs), g2 = norm2(gs), m2 = a2 + g2;
        if (a2 == T(0)) { const T ng = sqrt(g2); R.c = T(0); R.s = (T(1) / ng) * conj(gs); R.r = cx<T>(ng * isc, T(0)); return R; }
        const T t2 = T(1) / sqrt(a2 * m2);
        R.c = a2 * t2;
        R.s = t2 * (fs * conj(gs));
        R.r = ((m2 * t2) * isc) * fs;
    }
    return R;
}

// Schur form of an m x m (m <= SM = 64) upper Hessenberg matrix held in LDS, by EXPLICITLY shifted QR iterations executed
// by ONE wave.  Lane c owns column c while Q^H is applied from the left (the rotation that zeroes H[r+1,r] is generated by
// lane r as soon as its column has received the previous rotations, and broadcast through LDS), and row c while Q is
// applied from the right (all rotations are known by then, so the lanes run independently).  Compared with a rotation-by-
// rotation implicit chase this keeps every lane busy and needs no block barrier inside a QR iteration.
// On return Hs is upper triangular; if Us != nullptr it holds U with H_in = U T U^H.  Returns false if not converged.
// BC: the inputs (f, g) of rotation r are broadcast from lane r with v_readlane and EVERY lane generates the rotation, instead of lane r
// generating it and five ds_bpermute shuffles distributing the result (one LDS round trip less on the chain of every rotation).
template <class T, bool BC = false>
__device__ bool small_schur(cx<T>* Hs, int m, cx<T>* Us, Rot<T>* rots, const int SLD, long long* dbg = nullptr, int* nrot_out = nullptr) {
    const auto sync = [] { __syncthreads(); };
    const int lane = threadIdx.x;
    long long t_left = 0, t_right = 0, t_u = 0, t_pre = 0, tt = 0, n_it = 0, n_rot = 0;
    if (dbg) tt = clock64();
    const T ulp = eps_of<T>::value;
    int ihi = m - 1, its = 0, total = 0, nrot = 0;
    while (ihi > 0) {
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
        if (total > 40 * m) { if (nrot_out) *nrot_out = nrot; return false; }
        nrot += ihi - l;
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
    if (nrot_out) *nrot_out = nrot;
    return true;
}

// Swap the adjacent diagonal entries k, k+1 of the upper-triangular Ts (order m <= 64) by one rotation, accumulating into
// Vs (LAPACK ztrexc for complex Schur forms).  One wave.
template <class T>
__device__ void schur_swap(cx<T>* Ts, cx<T>* Vs, int m, int k, const int SLD) {
    const auto sync = [] { __syncthreads(); };
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
template <class T>
__device__ void small_larfg(const cx<T>* x, int inc, int len, cx<T>* vw, cx<T>& tau, T& beta) {
    const auto sync = [] { __syncthreads(); };
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
template <class T>
__device__ void small_apply_reflector(cx<T>* Ts, cx<T>* Vs, int m, int nrows_t, int o, int len, int c0, const cx<T>* vw, cx<T> tau, const int SLD) {
    const auto sync = [] { __syncthreads(); };
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
        s.ilo = 0; s.ihi = n - 1; s.nch = 0; s.mode = QR_IDLE; s.stall = 0; s.sweeps = 0; s.fail = 0;
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
                                                        int sm, int* __restrict__ counters, long long* dbg_all = nullptr) {
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
        int nrot = 0;
        const bool ok = small_schur<T, BC>(Hs, m, Us, rots, SLD, nullptr, &nrot);
        if (lane == 0) atomicAdd(&counters[1], nrot);           // dependent rotations of the in-LDS Schur solver (latency model of bench.py)
        __syncthreads();
        cx<T>* U = Uall + (long)b * QW * QW;                  // the unitary of a finished block / AED window (a DENSE link of the next sweep)
        for (int e = lane; e < m * m; e += 64) {
            const int r = e / m, c = e - r * m;
            H[(long)(ilo + r) * n + ilo + c] = (r <= c) ? Hs[r * SLD + c] : cx<T>(T(0), T(0));
            U[r * QW + c] = Us[r * SLD + c];
        }
        if (lane == 0) {
            clear_windows(st);
            st.w0[0] = ilo; st.w1[0] = ihi + 1;
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
    int nrot = 0;
    const bool okw = small_schur<T, BC>(Hs, nw, Us, rots, SLD, dbg, &nrot);
    if (lane == 0) atomicAdd(&counters[1], nrot);
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
        cx<T>* U = Uall + (long)b * QW * QW;
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

// Off-window updates on the matrix cores: H[w0:w1, w1:n) <- U^H H[w0:w1, w1:n)   (left),  H[0:w0, w0:w1) <- H[0:w0, w0:w1) U  and
// Z[:, w0:w1) <- Z[:, w0:w1) U   (right), with the window unitary U (ww x ww, ww <= 64) of a link of the log.
//
// Streaming design: the work of one matrix is cut into STRIPS of 16 columns (left) or 16 rows (right); one wave owns a strip,
// i.e. a 64 x 16 / 16 x 64 output block = four 16x16 MFMA tiles with the full K = ww.  U is staged ONCE per workgroup and link into LDS
// (split re/im planes) and serves as the A operand of the left update (U^H: the conjugation is folded into the signs of the
// four real MFMAs) and as the B operand of the right update; the streamed operand goes global memory -> registers directly
// in MFMA fragment layout (no LDS staging).  The k index of an MFMA step is permuted (k = 16c + 4*(lane>>4) + j, kstep()) so that
// a lane of the right update reads 4 consecutive elements of its row; both operands use the same permutation.
// Algorithmic intensity: 8*16*64*64 flops per 2*16 KiB moved = 16 flop/B in fp64 (32 in fp32); the deferred right update re-reads, per link,
// a block whose left 33 columns it wrote itself one link earlier (L2-hot), so about half of that traffic reaches HBM.
#ifndef TRX_QR_VAR
#define TRX_QR_VAR 0      // experiment switch of the fp32 update arithmetic (profiles/scripts/build_qr_variants.sh); 0 = the shipped code
#endif
constexpr int MLD = TRX_QR_VAR == 3 ? 72 : 68;           // LDS plane row stride: element U[k][c] at [k*MLD + c].  The k-groups of a fragment read are 4 rows apart (kstep below):
                                  // 4 * 68 = 272 elements = 16 banks (fp32, ds_read_b32: 32 banks) / 32 banks (fp64, ds_read_b64: 64 banks) -- no two
                                  // of the 32 lanes an LDS cycle serves share a bank (72 put every k-group on the same banks: 2-way conflicts)
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

// registers x[4cc + j] <- streamed operand element k = kstep(h, cc, j) + lk (half h of the strip's K range) of this lane's
// column (left) / row (right).  Out-of-range coordinates are CLAMPED to a valid element of the same matrix and the value is
// used as is: for k >= ww it meets a zero row of the padded U in LDS, and a lane whose row / column lies outside the region
// only feeds output elements that are never stored.  (No select after the load: the loaded registers have no consumer until
// the MFMAs of the next strip, so the loads stay in flight behind the current strip's arithmetic.)
template <class T, int SIDE = -1>      // SIDE 0 / 1: compile-time side of the strip (no address arithmetic for the other one), -1: d.side
__device__ __forceinline__ void slab_load_half(const SlabStrip<T>& d, int h, int n, int w0, int ww, int lane, cx<T> (&x)[8]) {
    const int side = SIDE >= 0 ? SIDE : d.side;
    const int lr = lane & 15, lk = lane >> 4;
    const int a = d.a0 + lr;
    const int ac = a < d.lim ? a : d.lim - 1;
    // 32-bit BYTE offsets from the wave-uniform matrix base (scalar base + 32-bit vector offset addressing; one address
    // register per access instead of two): requires n*n*sizeof(cx<T>) < 4 GiB, i.e. n < 16384 for complex128 (checked on the host)
    const unsigned p0 = (side == 0 ? (unsigned)w0 * n + ac : (unsigned)ac * n + w0) * (unsigned)sizeof(cx<T>);
    const unsigned ks = (side == 0 ? n : 1) * (unsigned)sizeof(cx<T>);
    const char* base = reinterpret_cast<const char*>(d.X);
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = kstep(h, cc, j) + KLS * lk;
            x[4 * cc + j] = *reinterpret_cast<const cx<T>*>(base + (p0 + (unsigned)(k < ww ? k : ww - 1) * ks));
        }
}

// BAND (see slab_multiply_half_3m): output tile q skips the k chunks c >= q + 2 of a chase unitary (compile-time conditions: h, cc, q are
// unrolled constants)
template <class T, int SIDE, bool BAND = false>
__device__ __forceinline__ void slab_multiply_half(const T* __restrict__ Ur, const T* __restrict__ Ui, int h, int lane, const cx<T> (&x)[8],
                                                   typename Mfma<T>::acc_t (&accR)[4], typename Mfma<T>::acc_t (&accI)[4]) {
    const int lr = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int chunk = 2 * h + cc;
            const int off = (kstep(h, cc, j) + KLS * lk) * MLD + lr;
            const T xr = x[4 * cc + j].x, xi = x[4 * cc + j].y;
            T ur[4], ui[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (BAND && chunk >= q + 2) continue;
                ur[q] = Ur[off + 16 * q]; ui[q] = Ui[off + 16 * q];
            }
            if (SIDE == 1) {          // C = X U:      Cr += xr ur - xi ui,  Ci += xr ui + xi ur          (A = x, B = u)
                const T nxi = -xi;
#pragma unroll
                for (int q = 0; q < 4; ++q) if (!(BAND && chunk >= q + 2)) accR[q] = Mfma<T>::mma(xr, ur[q], accR[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) if (!(BAND && chunk >= q + 2)) accI[q] = Mfma<T>::mma(xr, ui[q], accI[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) if (!(BAND && chunk >= q + 2)) accR[q] = Mfma<T>::mma(nxi, ui[q], accR[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) if (!(BAND && chunk >= q + 2)) accI[q] = Mfma<T>::mma(xi, ur[q], accI[q]);
            } else {                  // C = U^H X:    Cr += ur xr + ui xi,  Ci += ur xi - ui xr          (A = conj(u)^T, B = x)
                const T nxr = -xr;
#pragma unroll
                for (int q = 0; q < 4; ++q) if (!(BAND && chunk >= q + 2)) accR[q] = Mfma<T>::mma(ur[q], xr, accR[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) if (!(BAND && chunk >= q + 2)) accI[q] = Mfma<T>::mma(ur[q], xi, accI[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) if (!(BAND && chunk >= q + 2)) accR[q] = Mfma<T>::mma(ui[q], xi, accR[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) if (!(BAND && chunk >= q + 2)) accI[q] = Mfma<T>::mma(ui[q], nxr, accI[q]);
            }
        }
}

// 4M product of one K half for the tile pair (2 pp, 2 pp + 1).  BAND (see slab_multiply_half_3m): output tile q skips the k chunks
// c >= q + 2 of a chase unitary (compile-time conditions: h, cc, pp are unrolled constants).
template <class T, int SIDE, bool BAND = false>
__device__ __forceinline__ void slab_multiply_half_pair(const T* __restrict__ Ur, const T* __restrict__ Ui, int h, int pp, int lane, const cx<T> (&x)[8],
                                                        typename Mfma<T>::acc_t (&accR)[2], typename Mfma<T>::acc_t (&accI)[2]) {
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
            T ur[2], ui[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (!(q == 0 ? use0 : use1)) continue;
                ur[q] = Ur[off + 16 * q]; ui[q] = Ui[off + 16 * q];
            }
            if (SIDE == 1) {          // C = X U:      Cr += xr ur - xi ui,  Ci += xr ui + xi ur          (A = x, B = u)
                const T nxi = -xi;
#pragma unroll
                for (int q = 0; q < 2; ++q) if (q == 0 ? use0 : use1) accR[q] = Mfma<T>::mma(xr, ur[q], accR[q]);
#pragma unroll
                for (int q = 0; q < 2; ++q) if (q == 0 ? use0 : use1) accI[q] = Mfma<T>::mma(xr, ui[q], accI[q]);
#pragma unroll
                for (int q = 0; q < 2; ++q) if (q == 0 ? use0 : use1) accR[q] = Mfma<T>::mma(nxi, ui[q], accR[q]);
#pragma unroll
                for (int q = 0; q < 2; ++q) if (q == 0 ? use0 : use1) accI[q] = Mfma<T>::mma(xi, ur[q], accI[q]);
            } else {                  // C = U^H X:    Cr += ur xr + ui xi,  Ci += ur xi - ui xr          (A = conj(u)^T, B = x)
                const T nxr = -xr;
#pragma unroll
                for (int q = 0; q < 2; ++q) if (q == 0 ? use0 : use1) accR[q] = Mfma<T>::mma(ur[q], xr, accR[q]);
#pragma unroll
                for (int q = 0; q < 2; ++q) if (q == 0 ? use0 : use1) accI[q] = Mfma<T>::mma(ur[q], xi, accI[q]);
#pragma unroll
                for (int q = 0; q < 2; ++q) if (q == 0 ? use0 : use1) accR[q] = Mfma<T>::mma(ui[q], xi, accR[q]);
#pragma unroll
                for (int q = 0; q < 2; ++q) if (q == 0 ? use0 : use1) accI[q] = Mfma<T>::mma(ui[q], nxr, accI[q]);
            }
            // bound the hoisting of the U fragment reads to the next two k-steps (register budget)
            if (TRX_QR_VAR == 0 && (j & 1)) __builtin_amdgcn_sched_barrier(0);
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

template <class T, int SIDE = -1>
__device__ __forceinline__ void slab_store_pair(const SlabStrip<T>& d, int n, int w0, int ww, int lane, int pp, const typename Mfma<T>::acc_t (&accR)[2],
                                                const typename Mfma<T>::acc_t (&accI)[2]) {
    const int lr = lane & 15;
    char* base = reinterpret_cast<char*>(d.X);       // scalar base + 32-bit byte offsets, as in slab_load_half
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int cr = Mfma<T>::crow(lane, r);
#pragma unroll
        for (int q2 = 0; q2 < 2; ++q2) {
            const int q = 2 * pp + q2;
            const cx<T> v(accR[q2][r], accI[q2][r]);
            if ((SIDE >= 0 ? SIDE : d.side) == 1) {
                const int row = d.a0 + cr, k = 16 * q + lr;
                if (row < d.lim && k < ww) *reinterpret_cast<cx<T>*>(base + ((unsigned)row * n + w0 + k) * (unsigned)sizeof(cx<T>)) = v;
            } else {
                const int i = 16 * q + cr, col = d.a0 + lr;
                if (i < ww && col < d.lim) *reinterpret_cast<cx<T>*>(base + ((unsigned)(w0 + i) * n + col) * (unsigned)sizeof(cx<T>)) = v;
            }
        }
    }
}

template <class T, int SIDE>
__device__ __forceinline__ void slab_store(const SlabStrip<T>& d, int n, int w0, int ww, int lane, const typename Mfma<T>::acc_t (&accR)[4],
                                           const typename Mfma<T>::acc_t (&accI)[4]);
template <class T, int SIDE, bool BAND = false>
__device__ __forceinline__ void slab_compute(const T* __restrict__ Ur, const T* __restrict__ Ui, const SlabStrip<T>& d, int n, int w0, int ww, int lane,
                                             const cx<T> (&xa)[8], const cx<T> (&xb)[8]);

template <class T, int SIDE, bool BAND = false>
__device__ __forceinline__ void slab_strip(const T* __restrict__ Ur, const T* __restrict__ Ui, const SlabStrip<T>& d, int n, int w0, int ww, int lane) {
    cx<T> xa[8], xb[8];
    slab_load_half<T, SIDE>(d, 0, n, w0, ww, lane, xa);
    slab_load_half<T, SIDE>(d, 1, n, w0, ww, lane, xb);
    __builtin_amdgcn_sched_barrier(0);       // keep all 16 loads of the strip in flight ahead of the first MFMA (hipcc otherwise sinks them to ~3 deep)
    slab_compute<T, SIDE, BAND>(Ur, Ui, d, n, w0, ww, lane, xa, xb);
}

// multiply + store of one strip whose streamed operand is already in (or on its way to) registers
template <class T, int SIDE, bool BAND>
__device__ __forceinline__ void slab_compute(const T* __restrict__ Ur, const T* __restrict__ Ui, const SlabStrip<T>& d, int n, int w0, int ww, int lane,
                                             const cx<T> (&xa)[8], const cx<T> (&xb)[8]) {
    if constexpr (sizeof(T) == 8 || TRX_QR_VAR == 4) {
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
            slab_store_pair<T, SIDE>(d, n, w0, ww, lane, pp, p1, p2);
        }
        return;
    }
    if constexpr (TRX_QR_VAR == 2 || TRX_QR_VAR == 3) {
        typename Mfma<T>::acc_t accR[4], accI[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) { accR[q][r] = T(0); accI[q][r] = T(0); }
        slab_multiply_half<T, SIDE, BAND>(Ur, Ui, 0, lane, xa, accR, accI);
        slab_multiply_half<T, SIDE, BAND>(Ur, Ui, 1, lane, xb, accR, accI);
        slab_store<T, SIDE>(d, n, w0, ww, lane, accR, accI);
        return;
    }
    // fp32: 4M product (its error budget is the tight one), two of the four output tiles at a time: 16 accumulator registers, so that the
    // kernel fits 128 registers and two of its workgroups share a CU with a window workgroup
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
        typename Mfma<T>::acc_t accR[2], accI[2];
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) { accR[q][r] = T(0); accI[q][r] = T(0); }
        slab_multiply_half_pair<T, SIDE, BAND>(Ur, Ui, 0, pp, lane, xa, accR, accI);
        slab_multiply_half_pair<T, SIDE, BAND>(Ur, Ui, 1, pp, lane, xb, accR, accI);
        slab_store_pair<T, SIDE>(d, n, w0, ww, lane, pp, accR, accI);
    }
}

// Result register r of tile q:  right update: strip row crow(lane, r), window column 16q + (lane&15);
//                               left update:  window row 16q + crow(lane, r), strip column lane&15.
template <class T, int SIDE>
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
            if ((SIDE >= 0 ? SIDE : d.side) == 1) {
                const int row = d.a0 + cr, k = 16 * q + lr;
                if (row < d.lim && k < ww) *reinterpret_cast<cx<T>*>(base + ((unsigned)row * n + w0 + k) * (unsigned)sizeof(cx<T>)) = v;
            } else {
                const int i = 16 * q + cr, col = d.a0 + lr;
                if (i < ww && col < d.lim) *reinterpret_cast<cx<T>*>(base + ((unsigned)(w0 + i) * n + col) * (unsigned)sizeof(cx<T>)) = v;
            }
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

// Window steps of one bulge chain (blockIdx.x = chain, blockIdx.y = matrix).  Thread layout: QNS groups of LPB lanes, group s owns bulge s: its lanes compute the
// rotation redundantly (no broadcast barrier), then stride over the window's columns (left rotation) and, after one barrier,
// over its rows and the rows of U (right rotation).  Two barriers per chain step.  LPB = 64 (one wave per bulge, 1024
// threads): a chain step is bound by the vector issue time of the 2 + 4 rotated element pairs per lane-quartet (cycle
// counters, TRX_QR_DEBUG: 16 lanes per bulge spent 3600 cycles per step, ~2600 of them in the two rotation phases), so the
// widest mapping the 64-wide window admits is the fastest one.
// Two phases in ONE window-sized LDS buffer: (1) the chase on the H window, every rotation logged (c, s: 24 bytes); H written
// back; (2) the same buffer becomes U = I and the log is replayed onto it (bulge s's wave rotates its two columns, one barrier
// per chain step).
//
// SUPER-STEPS (one chain per sweep): a launch takes the chain through `nsteps` consecutive windows.  Window step s + 1 needs, of
// everything off window s, only the NEW COLUMNS it slides over to have received U_s^H from the left -- so the workgroup applies U_s^H itself
// to the band [w1_s, E) right of its window (E = end of the band all steps of the launch share, <= nsteps * 31 + ... columns; one 16-column
// strip per wave on the matrix cores, U_s converted in place to the split re/im planes the strip code reads), publishes (w0, w1, E) in
// the link and moves on; the left update of the columns from E on is ONE launch per super-step over all its links (apply_links_kernel<0>),
// the right / Z update one launch per sweep.  A sweep over 1922 rows is then ~16 launches of each kind instead of 62 window -> update pairs.
constexpr int LPB = 64;                    // lanes per bulge
constexpr int WTHREADS = QNS * LPB;        // threads of the window kernel
constexpr int WIT = QW / LPB;              // element pairs per lane and phase
constexpr int WMAXS = 96;                  // chain steps per window (rotation log: WMAXS x QNS entries = 37 KB): the first window of a sweep
                                           // chases 62 steps and the last one up to ~94, so no window is split (48 split 7 % of them)
constexpr int QSUPER = 8;                  // most window steps per launch
template <class T> struct RotCS { T c; cx<T> s; };

// Left update of ONE 16-column strip of the band by one wave, lean in registers (the window kernel runs 4 waves per SIMD): the strip's
// 64 x 16 block goes to registers in MFMA fragment layout (slab_load_half), then one 16 x 16 output tile at a time
// (8 accumulator registers) with the banded structure of a chase unitary (tile q needs the k chunks <= q + 1 only).
template <class T>
__device__ __forceinline__ void band_left_strip(const T* __restrict__ Ur, const T* __restrict__ Ui, const SlabStrip<T>& d, int n, int w0, int ww, int lane, bool band) {
    const int lr = lane & 15, lk = lane >> 4;
    char* base = reinterpret_cast<char*>(d.X);
    // x[4 c + j] <- H[w0 + k, a0 + lr], k = 16 c + 4 lk + j (clamped as in slab_load_half: beyond ww it meets zero rows of the planes).  One
    // address register: the offset advances load by load (sched_barrier keeps address arithmetic and load together; the loads stay in flight)
    cx<T> x[16];
    {
        const int a = d.a0 + lr;
        const int ac = a < d.lim ? a : d.lim - 1;
        const unsigned ks = (unsigned)n * (unsigned)sizeof(cx<T>);
        const unsigned p0 = ((unsigned)w0 * n + ac) * (unsigned)sizeof(cx<T>);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 16 * c + KLS * lk + j;
                x[4 * c + j] = *reinterpret_cast<const cx<T>*>(base + (p0 + (unsigned)(k < ww ? k : ww - 1) * ks));
                __builtin_amdgcn_sched_barrier(0);
            }
    }
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        typename Mfma<T>::acc_t accR, accI;
#pragma unroll
        for (int r = 0; r < 4; ++r) { accR[r] = T(0); accI[r] = T(0); }
        const int cmax = band ? q + 1 : 3;              // last k chunk with a nonzero block in tile column q
        const T* ur = Ur + (KLS * lk) * MLD + lr + 16 * q;
        const T* ui = Ui + (KLS * lk) * MLD + lr + 16 * q;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c > cmax) continue;                      // (wave-uniform)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const T a = ur[(16 * c + j) * MLD], bq = ui[(16 * c + j) * MLD];
                const cx<T> xv = x[4 * c + j];
                // C = U^H X:    Cr += ur xr + ui xi,  Ci += ur xi - ui xr     (the 3M form -- P1 = ur xr, P2 = ui xi, P3 = (ur - ui)(xr + xi) -- was
                // measured in the fused launches in round 6: QR phase 919 / 920 against 904 / 906 ms; profiles/r06_ab/r6n_left_update_3m.txt)
                accR = Mfma<T>::mma(a, xv.x, accR);
                accI = Mfma<T>::mma(a, xv.y, accI);
                accR = Mfma<T>::mma(bq, xv.y, accR);
                accI = Mfma<T>::mma(bq, -xv.x, accI);
                if (j & 1) __builtin_amdgcn_sched_barrier(0);        // bounds the hoisting of the fragment reads (register budget)
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 16 * q + Mfma<T>::crow(lane, r), col = d.a0 + lr;
            if (i < ww && col < d.lim) *reinterpret_cast<cx<T>*>(base + ((unsigned)(w0 + i) * n + col) * (unsigned)sizeof(cx<T>)) = cx<T>(accR[r], accI[r]);
        }
    }
}

// Right update of ONE 16-row strip by one wave, the mirror image of band_left_strip: rows a0 .. a0 + 15 of the columns w0 .. w0 + ww of X (H above
// the window, or Z) times U.  The strip's 16 x 64 block sits in registers as the A operand (lane: row lr, k = 16 c + 4 lk + j), U comes from the
// planes as the B operand, one 16 x 16 output tile at a time; a chase unitary has nonzero blocks only for k chunks <= q + 1 of column tile q.
template <class T>
__device__ __forceinline__ void band_right_strip(const T* __restrict__ Ur, const T* __restrict__ Ui, const SlabStrip<T>& d, int n, int w0, int ww, int lane, bool band) {
    const int lr = lane & 15, lk = lane >> 4;
    char* base = reinterpret_cast<char*>(d.X);
    cx<T> x[16];
    {
        const int r = d.a0 + lr;
        const int rc = r < d.lim ? r : d.lim - 1;
        const unsigned p0 = ((unsigned)rc * n + w0) * (unsigned)sizeof(cx<T>);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = 16 * c + KLS * lk + j;
                x[4 * c + j] = *reinterpret_cast<const cx<T>*>(base + (p0 + (unsigned)(k < ww ? k : ww - 1) * (unsigned)sizeof(cx<T>)));
            }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
        typename Mfma<T>::acc_t accR, accI;
#pragma unroll
        for (int r = 0; r < 4; ++r) { accR[r] = T(0); accI[r] = T(0); }
        const int cmax = band ? q + 1 : 3;
        const T* ur = Ur + (KLS * lk) * MLD + lr + 16 * q;
        const T* ui = Ui + (KLS * lk) * MLD + lr + 16 * q;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (c > cmax) continue;                      // (wave-uniform)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const T a = ur[(16 * c + j) * MLD], bq = ui[(16 * c + j) * MLD];
                const cx<T> xv = x[4 * c + j];
                // C = X U:    Cr += xr ur - xi ui,  Ci += xr ui + xi ur
                accR = Mfma<T>::mma(xv.x, a, accR);
                accI = Mfma<T>::mma(xv.x, bq, accI);
                accR = Mfma<T>::mma(-xv.y, bq, accR);
                accI = Mfma<T>::mma(xv.y, a, accI);
                if (j & 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = d.a0 + Mfma<T>::crow(lane, r), col = 16 * q + lr;
            if (row < d.lim && col < ww) *reinterpret_cast<cx<T>*>(base + ((unsigned)row * n + w0 + col) * (unsigned)sizeof(cx<T>)) = cx<T>(accR[r], accI[r]);
        }
    }
}

// Right / Z update of ONE chase link (several chains per sweep: one window step per launch) by the 16 waves of a workgroup of the window
// kernel: U into the planes, then strips g0, g0 + 1, ... < g1 of the list [H rows 0 .. w0 in 16-row strips | Z rows 0 .. n], one per wave.
template <class T>
__device__ void right_link_strips(cx<T>* __restrict__ H, cx<T>* __restrict__ Z, int n, const QrLink& l, const cx<T>* __restrict__ U, T* Ur, T* Ui, int* vote,
                                  int g0, int g1, int band_on, unsigned* __restrict__ work) {
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr int UPT = QW * QW / WTHREADS;
    const int kind = __builtin_amdgcn_readfirstlane(l.kind), w0 = __builtin_amdgcn_readfirstlane(l.w0), w1 = __builtin_amdgcn_readfirstlane(l.w1);
    const int ww = w1 - w0;
    if (kind != QRL_CHASE || ww <= 0) return;
    const int nH = (w0 + 15) >> 4, nZ = (n + 15) >> 4;
    const int lim = g1 < nH + nZ ? g1 : nH + nZ;
    if (g0 >= lim) return;                                   // (workgroup-uniform)
    cx<T> ureg[UPT];
#pragma unroll
    for (int q = 0; q < UPT; ++q) {
        const int el = t + WTHREADS * q, k = el >> 6, c = el & 63;
        ureg[q] = U[(k < ww ? k : ww - 1) * QW + (c < ww ? c : ww - 1)];
    }
    if (t == 0) vote[0] = 0;
    __syncthreads();
    int dense = 0;
#pragma unroll
    for (int q = 0; q < UPT; ++q) {
        const int el = t + WTHREADS * q, k = el >> 6, c = el & 63;
        cx<T> u = ureg[q];
        if (k >= ww || c >= ww) u = cx<T>(T(0), T(0));
        Ur[k * MLD + c] = u.x; Ui[k * MLD + c] = u.y;
        if ((k >> 4) >= (c >> 4) + 2 && (u.x != T(0) || u.y != T(0))) dense = 1;
    }
    if (dense) vote[0] = 1;
    __syncthreads();
    const bool band = band_on && vote[0] == 0;
    int mine = 0;
    for (int g = g0 + wave; g < lim; g += WTHREADS / 64) {
        SlabStrip<T> d;
        d.side = 1;
        if (g < nH) { d.X = H; d.a0 = 16 * g; d.lim = w0; }
        else { d.X = Z; d.a0 = 16 * (g - nH); d.lim = n; }
        band_right_strip<T>(Ur, Ui, d, n, w0, ww, lane, band);
        ++mine;
    }
    if (lane == 0 && mine > 0) atomicAdd(work, (unsigned)(((long)ww * ww * 16 * mine) >> 12));       // units of 4096 complex MACs
}

// Fused launches (qr_window_kernel): band end of a launch whose predecessor's band ended at e_prev -- every window of the launch and the first
// one of the next end left of it, because the predecessor's band already reached the first window of this launch and a window step advances
// by at most QW - 2 k - 1 columns -- and the number of 16-column strips between the two, which the chase workgroup brings up to date itself.
__device__ __forceinline__ int fused_band_end(int e_prev, int k, int nsteps, int n) {
    const int e = e_prev + nsteps * (QW - 2 * k - 1);
    return e > n ? n : e;
}
__device__ __forceinline__ int fused_catchup_strips(int e_prev, int k, int nsteps, int n) {
    const int e = fused_band_end(e_prev, k, nsteps, n);
    return e > e_prev ? (e - e_prev + 15) >> 4 : 0;
}

// U^H from the left for the chase links lks[0], lks[kc], ... (pnq of them, one chain) on the 16-column strips a0 = e + 16 g of the columns right
// of their launch's band (e: recorded in the links), for g = g0 + wave, + gs, ... < g1; one strip per wave and pass, by the lean band routine
// of the window kernel.  All 1024 threads of the workgroup; U is staged per link into split planes at the start of the dynamic LDS.
template <class T>
__device__ void left_links_strips(cx<T>* __restrict__ H, int n, const QrLink* __restrict__ lks, int kc, const cx<T>* __restrict__ Ulog, int pnq, T* Ur, T* Ui,
                                  int* vote, int g0, int g1, int gs, int band_on, unsigned* __restrict__ work) {
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr int UPT = QW * QW / WTHREADS;
    // (Fetching the NEXT link's window unitary into registers under the current link's strips was measured in round 6: QR phase 927 / 932 against
    // 894 / 894 ms at batch 128, 627 against 607 at batch 64 -- removed; profiles/r06_ab/r6o_u_prefetch_and_chains.txt)
    for (int i = 0; i < pnq; ++i) {
        const QrLink l = lks[(long)i * kc];
        const int kind = __builtin_amdgcn_readfirstlane(l.kind), w0 = __builtin_amdgcn_readfirstlane(l.w0), w1 = __builtin_amdgcn_readfirstlane(l.w1);
        const int e = __builtin_amdgcn_readfirstlane(l.e);
        const int ww = w1 - w0;
        if (kind != QRL_CHASE || ww <= 0) continue;
        const int nL = n > e ? (n - e + 15) >> 4 : 0;
        const int lim = g1 < nL ? g1 : nL;
        if (g0 >= lim) continue;                             // (workgroup-uniform)
        const cx<T>* U = Ulog + (long)i * kc * QW * QW;
        cx<T> ureg[UPT];
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
            const int el = t + WTHREADS * q, k = el >> 6, c = el & 63;
            ureg[q] = U[(k < ww ? k : ww - 1) * QW + (c < ww ? c : ww - 1)];
        }
        if (t == 0) vote[i & 1] = 0;
        __syncthreads();                                     // the previous link's readers are done with the planes
        int dense = 0;
#pragma unroll
        for (int q = 0; q < UPT; ++q) {
            const int el = t + WTHREADS * q, k = el >> 6, c = el & 63;
            cx<T> u = ureg[q];
            if (k >= ww || c >= ww) u = cx<T>(T(0), T(0));
            Ur[k * MLD + c] = u.x; Ui[k * MLD + c] = u.y;
            if ((k >> 4) >= (c >> 4) + 2 && (u.x != T(0) || u.y != T(0))) dense = 1;
        }
        if (dense) vote[i & 1] = 1;
        __syncthreads();
        const bool band = band_on && vote[i & 1] == 0;
        int mine = 0;
        for (int g = g0 + wave; g < lim; g += gs) {
            SlabStrip<T> d;
            d.X = H; d.side = 0; d.a0 = e + 16 * g; d.lim = n;
            band_left_strip<T>(Ur, Ui, d, n, w0, ww, lane, band);
            ++mine;
        }
        if (lane == 0 && mine > 0) atomicAdd(work, (unsigned)(((long)ww * ww * 16 * mine) >> 12));       // units of 4096 complex MACs
    }
    __syncthreads();                                         // planes free again
}

// DBG: cycle counters of matrix 0, chain 0 (TRX_QR_DEBUG); the production instantiation carries none of it.  (A 64-register build of the fp32
// kernel -- so that its 4 waves per SIMD fit next to update workgroups -- spilled 45 registers in the band update and changed nothing:
// profiles/r05_ab/r5h_occupancy.txt.)
template <class T, bool DBG>
__global__ __launch_bounds__(WTHREADS) void qr_window_kernel(cx<T>* __restrict__ Aall, long mstride, int n, QrState* __restrict__ st_all,
                                                        cx<T>* __restrict__ Ulog_all, QrLink* __restrict__ links_all, const cx<T>* __restrict__ shifts_all,
                                                        int par, int nslot, int kc, int slot0, int nsteps, int band_on, int* __restrict__ counters, long long* dbg_all,
                                                        int prev_q0, int prev_nq, unsigned* __restrict__ work, cx<T>* __restrict__ Zall) {
    // FUSED LAUNCH (one chain per sweep, prev_nq > 0): besides the chase workgroups (blockIdx.x < kc) the grid carries FAR workgroups
    // (blockIdx.x >= kc) that apply the left update of the PREVIOUS launch's links to the columns this launch's chase does not touch -- the
    // work that sat, as a launch of its own, between two chase launches of the chain in rounds 4 - 5 (76 us alone, 207 us in situ, eleven
    // times per iteration) now runs beside the chase, on other compute units.  The columns the chase does reach -- [e_prev, e_this), e = the
    // band end of a launch -- get the previous links from the chase workgroup itself before its first window (catch-up), so that every
    // strip still sees the links in order.  No stream, no event, one launch fewer per super-step.
    TRX_DYN_SMEM(smem);
    constexpr int LD = QW + 1;
    cx<T>* Hw = reinterpret_cast<cx<T>*>(smem);      // [QW][LD]   phase 1: H window;  phase 2: U
    RotCS<T>* rlog = reinterpret_cast<RotCS<T>*>(Hw + QW * LD);      // [WMAXS][QNS]
    QrState& sst = *reinterpret_cast<QrState*>(rlog + WMAXS * QNS);
    int* sflag = reinterpret_cast<int*>(&sst + 1);                   // [0] "U has a nonzero outside the band"; [1], [2]: the same for the links of left_links_strips
    T* Ur = reinterpret_cast<T*>(smem);              // [QW][MLD] x 2: split planes of U for the band update, over Hw | rlog (both dead by then)
    T* Ui = Ur + QW * MLD;
    static_assert(sizeof(T) * 2 * QW * MLD <= sizeof(cx<T>) * QW * LD + sizeof(RotCS<T>) * WMAXS * QNS, "planes fit over window + log");
    if ((int)blockIdx.x >= kc && kc > 1) {
        // Several chains per sweep (one window step per launch): the extra workgroups apply the PREVIOUS step's links from the right to H and to
        // Z.  That update touches rows above and columns of the previous windows only; this launch's chase workgroups work on windows that
        // start at or below the end of their own previous window and, for a follower, end at or above the start of the previous window of the
        // chain ahead (the gate in the chase loop below) -- disjoint from all of it -- and the left update of the previous step, with which it
        // shares blocks, was its own launch in between.  One launch fewer on the chain of every window step (rounds 3 - 5: window, left, right).
        const int b = blockIdx.y, fx = (int)blockIdx.x - kc, per = ((int)gridDim.x - kc) / kc;
        const int ch = fx / per, part = fx - ch * per;
        const QrLink l = links_all[((long)b * nslot + prev_q0) * kc + ch];
        const int spr = 16 * ((2 * ((n + 15) >> 4) + 16 * per - 1) / (16 * per));        // strips per rider: the whole list over `per` riders, in rounds of 16
        right_link_strips<T>(Aall + (long)b * mstride, Zall + (long)b * mstride, n, l, Ulog_all + (((long)b * nslot + prev_q0) * kc + ch) * QW * QW, Ur, Ui,
                             sflag + 1, spr * part, spr * (part + 1), band_on, work + 4);
        return;
    }
    if ((int)blockIdx.x >= kc) {
        const int b = blockIdx.y, fx = (int)blockIdx.x - kc, nfar = (int)gridDim.x - kc;
        if (threadIdx.x == 0) sst = st_all[b];
        __syncthreads();
        // (Nothing here may depend on the chase position: the chase workgroup of this launch rewrites it when it is done, and a far workgroup
        // can be scheduled late.  The split of the strips is a function of the previous launch's band end and of constants of the sweep.)
        const QrLink* pl = links_all + ((long)b * nslot + prev_q0) * kc;                   // (chain 0: kc == 1 in fused launches)
        const int gc = fused_catchup_strips(pl[0].e, sst.k[0], nsteps, n);                   // strips the chase workgroup catches up on itself
        left_links_strips<T>(Aall + (long)b * mstride, n, pl, kc, Ulog_all + ((long)b * nslot + prev_q0) * kc * QW * QW, prev_nq, Ur, Ui, sflag + 1,
                             gc + 16 * fx, 1 << 30, 16 * nfar, band_on, work + 7);
        return;
    }
    long long* dbg = (DBG && dbg_all && blockIdx.y == 0 && blockIdx.x == 0 && threadIdx.x == 0) ? dbg_all : nullptr;
    long long tk0 = dbg ? clock64() : 0;
    const int b = blockIdx.y, ch = blockIdx.x, t = threadIdx.x;
    if (t == 0) sst = st_all[b];
    __syncthreads();
    // the entries of the link log of these window steps: EVERY exit path writes them (the update kernels walk over all slots of the sweep)
    QrLink* links = links_all + ((long)b * nslot + slot0) * kc + ch;         // step s: links[s * kc]
    const QrState& st = sst;           // read in place (LDS): a register copy indexed by the chain number would live in scratch
    // Every block writes only the fields of its own chain (and chain 0's block the mode); reads of the other chains' chase
    // positions go to the [par] copy, which nobody writes in this launch (several chains: one window step per launch).
    if (st.mode == QR_SMALL_PENDING || st.mode == QR_AED_CHASE) {
        // this slot applies the unitary of the finished block / the AED window (written by the prepare kernel): a dense link, updated on all sides at once
        if (t == 0) {
            QrLink l; l.w0 = 0; l.w1 = 0; l.kind = QRL_NONE; l.e = 0;
            for (int s = 1; s < nsteps; ++s) links[s * kc] = l;
            if (ch == 0) { l.w0 = st.w0[0]; l.w1 = st.w1[0]; l.kind = QRL_DENSE; l.e = l.w1; st_all[b].mode = st.mode == QR_SMALL_PENDING ? QR_SMALL_APPLIED : QR_CHASE; }
            links[0] = l;
        }
        return;
    }
    const int k = st.k[ch], ilo = st.ilo, ihi = st.ihi, tau_last = st.tau_last[ch];
    const bool chasing = (st.mode == QR_CHASE) && ch < st.nch;
    int tau_cur = st.tau[ch][par];
    int band_e = 0;                                  // end of the band this launch keeps up to date itself (0: not decided yet)
    cx<T>* H = Aall + (long)b * mstride;
    const int sb = t / LPB, j = t & (LPB - 1);       // bulge index, lane within the group
    const cx<T> shift = (sb < k) ? shifts_all[((long)b * QKC + ch) * QNS + sb] : cx<T>(T(0), T(0));
    constexpr int RPT = QW * QW / WTHREADS, RSTEP = WTHREADS / QW;      // window elements per thread, row stride between them
    if (prev_nq > 0 && kc == 1) {
        // catch-up of a fused launch: the previous launch's links on the strips between its band end and ours (see the kernel head), whether
        // or not the chain still moves; our own band then ends where the far workgroups begin
        const QrLink* pl = links_all + ((long)b * nslot + prev_q0) * kc + ch;
        if (pl[0].kind == QRL_CHASE) {
            const int e_prev = pl[0].e;
            const int gc = fused_catchup_strips(e_prev, st.k[0], nsteps, n);
            left_links_strips<T>(H, n, pl, kc, Ulog_all + (((long)b * nslot + prev_q0) * kc + ch) * QW * QW, prev_nq, Ur, Ui, sflag + 1, 0, gc, WTHREADS / 64, band_on, work + 7);
            band_e = fused_band_end(e_prev, st.k[0], nsteps, n);
        }
    }
    for (int s = 0; s < nsteps; ++s) {
        const int tau0 = tau_cur;
        bool move = chasing && tau0 <= tau_last;
        int w0 = 0, w1 = 0, tau_end = 0;
        if (move) {
            chain_window(ilo, ihi, k, tau0, tau_last, w0, w1, tau_end);
            if (tau_end > tau0 + WMAXS - 1) tau_end = tau0 + WMAXS - 1;      // first and last window of a sweep: several steps
            if (ch > 0 && st.tau[ch - 1][par] <= st.tau_last[ch - 1]) {
                // the chain ahead is still under way: its last bulge sits at the start of ITS window (whether or not it moves in
                // this step); this chain may only work strictly above it
                int p0, p1, pe;
                chain_window(ilo, ihi, st.k[ch - 1], st.tau[ch - 1][par], st.tau_last[ch - 1], p0, p1, pe);
                if (w1 > p0) move = false;
            }
            if (ch > 0 && prev_nq > 0) {
                // the previous step's right / Z update rides in THIS launch (extra workgroups, see the kernel head): it writes the columns of the
                // PREVIOUS window of the chain ahead in the rows above it, so this chain's window must also end above that window's start
                const QrLink pa = links_all[((long)b * nslot + prev_q0) * kc + ch - 1];
                if (pa.kind == QRL_CHASE && w1 > pa.w0) move = false;
            }
        }
        if (!move) {
            if (t == 0) {
                QrLink l; l.w0 = 0; l.w1 = 0; l.kind = QRL_NONE; l.e = 0;
                for (int s2 = s; s2 < nsteps; ++s2) links[s2 * kc] = l;
            }
            break;
        }
        const int ww = w1 - w0;
        if (band_e == 0) {
            // in-kernel band: one chain per sweep.  Every window of this launch ends at most (QW - 2 k - 1) columns further right
            // than the one before, and the first window of the NEXT launch as well: all of them lie left of E.
            band_e = w1;
            if (kc == 1) { band_e = w1 + nsteps * (QW - 2 * k - 1); if (band_e > n) band_e = n; }
        }
        {
            // window load: QW*QW/WTHREADS independent (clamped) global loads per thread in flight, then the LDS fill
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
        __syncthreads();
        if (dbg) { const long long t1 = clock64(); dbg[12] += t1 - tk0; tk0 = t1; }
        // Every phase of a chain step touches WIT element pairs per lane: the loops are fully unrolled with clamped LDS reads issued
        // up front and guarded writes, so a phase costs one LDS round trip.  (Computing the next rotation right after the H part of
        // the right phase, to overlap it with the U part, was tried and is not faster: hipcc serialises the two and the second
        // barrier absorbs the skew.)
        // (Forwarding the next rotation's inputs from this wave's own right rotation -- two v_readlane instead of the LDS read behind the step's
        // second barrier -- was measured in round 6: QR phase 939 against 919 ms at batch 128, 440 against 416 ms at batch 16.  Not in the tree.)
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
        cx<T>* U = Ulog_all + (((long)b * nslot + slot0 + s) * kc + ch) * QW * QW;
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
        const bool do_band = band_e > w1;            // (workgroup-uniform)
        {
            // U goes to the log; for the band update also, through registers, into split planes over the same LDS (zero outside ww x ww)
            const int c = t & (QW - 1), r4 = t / QW;
            cx<T> hv[RPT];
#pragma unroll
            for (int i = 0; i < RPT; ++i) hv[i] = Hw[(r4 + RSTEP * i) * LD + c];
            if (c < ww) {
#pragma unroll
                for (int i = 0; i < RPT; ++i) {
                    const int r = r4 + RSTEP * i;
                    if (r < ww) U[r * QW + c] = hv[i];
                }
            }
            if (do_band) {
                if (t == 0) *sflag = 0;
                __syncthreads();                         // every thread holds its part of U; the buffer may be overwritten
                int dense = 0;
#pragma unroll
                for (int i = 0; i < RPT; ++i) {
                    const int r = r4 + RSTEP * i;
                    cx<T> v = hv[i];
                    if (r >= ww || c >= ww) v = cx<T>(T(0), T(0));
                    Ur[r * MLD + c] = v.x; Ui[r * MLD + c] = v.y;
                    if ((r >> 4) >= (c >> 4) + 2 && (v.x != T(0) || v.y != T(0))) dense = 1;
                }
                if (dense) *sflag = 1;
                __syncthreads();
            }
        }
        if (t == 0) {
            QrLink l; l.w0 = w0; l.w1 = w1; l.kind = QRL_CHASE; l.e = band_e;
            links[s * kc] = l;
        }
        if (do_band) {
            // left update of the band [w1, band_e): one 16-column strip per wave (16 waves)
            const bool band = band_on && *sflag == 0;       // (knob slab_band: dense product although the unitary is banded)
            const int wv = __builtin_amdgcn_readfirstlane(t >> 6);
            for (int a0 = w1 + 16 * wv; a0 < band_e; a0 += 16 * (WTHREADS / 64)) {
                SlabStrip<T> d;
                d.X = H; d.side = 0; d.a0 = a0; d.lim = band_e;
                band_left_strip<T>(Ur, Ui, d, n, w0, ww, t & 63, band);
            }
        }
        if (t == 0) atomicAdd(&counters[0], tau_end - tau0 + 1);      // chain steps of this matrix (latency model of bench.py)
        tau_cur = tau_end + 1;
        if (dbg) { dbg[14] += clock64() - tk0; tk0 = clock64(); }
        if (s + 1 < nsteps) __syncthreads();         // window write-back + band update visible to the next step's loads; LDS free again
    }
    if (t == 0) st_all[b].tau[ch][par ^ (nsteps & 1)] = tau_cur;
}

// Waves per SIMD the update kernel is compiled for: fp64 2 (<= 256 registers), fp32 3 (<= 168).  (4, i.e. <= 128 registers, spilled 26 - 47
// and was not faster; a software-pipelined fp32 variant -- double-buffered U, two strips per wave, 240 registers -- was slower than this kernel
// once its staging loads were batched: profiles/r05_ab/.)
#define APPLY_MIN_WG(T) (sizeof(T) == 4 ? 3 : 2)

// The update kernels walk over LINKS of the sweep's log (QrLink: window [w0, w1), kind, e = first column the left update still has to
// reach; the window unitary sits in the same slot of the U log, that of a dense link in the per-matrix buffer the prepare kernel writes).
//
// MODE 0 -- right behind a super-step, the CHASE links among [q0, q0 + nq) of chain blockIdx.x / units:
//     H[w0:w1, e:n) <- U^H H[w0:w1, e:n)            (the left update: the next window's new columns are among these)
// MODE 2 -- slot 0 of a sweep, the DENSE link if there is one (unitary of an AED window / a finished block: the chase that follows reads rows
//     above that window, so nothing of it can wait):  the left update + H[0:w0, w0:w1) <- H[0:w0, w0:w1) U + Z[:, w0:w1) <- Z[:, w0:w1) U
//     A workgroup takes 4 * spw consecutive strips (its 4 waves interleaved, so that they stream neighbouring columns / rows).
//     (Separate instantiations because the side of a strip is then a compile-time constant in MODE 0 and 1: no address arithmetic for the
//     other side, ~40 registers less -- the fp32 kernels run 4 instead of 2 waves per SIMD.)
// MODE 1 -- links [q0, q0 + nq) x all chains, in order, chase links only (`units` = parts: bit 0 rows of H, bit 1 rows of Z):
//     H[0:w0, w0:w1) <- H[0:w0, w0:w1) U   and   Z[:, w0:w1) <- Z[:, w0:w1) U.
//     A workgroup owns 64 ROWS (one 16-row strip per wave) of Z (the first blocks) or of H and takes them through every link whose
//     window lies below them: U_q is staged once per link, the strip's 16 x 64 block moves 31 columns to the right from link to link.
//     ONE chain per sweep: once per sweep for both parts -- nothing else touches these rows x columns during the sweep (left updates act
//     on rows INSIDE a window, and no window of the sweep comes back to rows above an earlier one), and left / right multiplications
//     commute, so the result is that of the interleaved order.  SEVERAL chains: a following chain's window does come back to rows the
//     chain ahead has updated from the right, so the H part runs after every window step (nq = 1) and only Z waits for the end of the sweep.
template <class T, int MODE>
__global__ __launch_bounds__(256, APPLY_MIN_WG(T)) void apply_links_kernel(cx<T>* __restrict__ Aall, cx<T>* __restrict__ Zall, long mstride, int n,
                                                          const QrLink* __restrict__ links_all, const cx<T>* __restrict__ Ulog_all,
                                                          const cx<T>* __restrict__ Udense_all, unsigned* __restrict__ work, int nslot, int kc,
                                                          int q0, int nq, int spw, int units, int band_on) {
    TRX_DYN_SMEM(smem);
    T* Ur = reinterpret_cast<T*>(smem);      // [QW][MLD]
    T* Ui = Ur + QW * MLD;
    int* wdense = reinterpret_cast<int*>(Ui + QW * MLD);          // [4] per-wave votes, behind the planes
    QrLink* lks = reinterpret_cast<QrLink*>(wdense + 4);          // this launch's link records (one coalesced read instead of a dependent global read per link)
    const int b = blockIdx.y;
    const int t = threadIdx.x, lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);      // strip descriptors live in scalar registers
    const int ch0 = MODE != 1 ? (int)blockIdx.x / units : 0;
    const int gx = MODE != 1 ? (int)blockIdx.x - ch0 * units : (int)blockIdx.x;
    const int unitsZ = (MODE == 1 && (units & 2)) ? (n + 63) >> 6 : 0;
    const bool isZ = MODE == 1 && gx < unitsZ;
    const int row0 = MODE == 1 ? 64 * (isZ ? gx : gx - unitsZ) : 0;     // MODE 1: first of this workgroup's 64 rows
    cx<T>* H = Aall + (long)b * mstride;
    cx<T>* Z = Zall + (long)b * mstride;
    for (int i = t; i < nq * kc; i += 256) lks[i] = links_all[((long)b * nslot + q0) * kc + i];
    __syncthreads();
    bool staged = false;                                              // LDS holds a U some wave may still be reading
    for (int qq = q0; qq < q0 + nq; ++qq)
        for (int ch = (MODE != 1 ? ch0 : 0); ch < (MODE != 1 ? ch0 + 1 : kc); ++ch) {
            const QrLink* lp = lks + (qq - q0) * kc + ch;
            const int kind = __builtin_amdgcn_readfirstlane(lp->kind);
            if (kind != (MODE == 2 ? QRL_DENSE : QRL_CHASE)) continue;
            const int w0 = __builtin_amdgcn_readfirstlane(lp->w0), w1 = __builtin_amdgcn_readfirstlane(lp->w1);
            const int e = __builtin_amdgcn_readfirstlane(lp->e);
            const int ww = w1 - w0;
            if (ww <= 0) continue;
            // strips of this link that fall to this workgroup (everything here is workgroup-uniform)
            int nL = 0, nR = 0, nZ = 0, S = 0, g0 = 0;
            if (MODE != 1) {
                nL = n > e ? (n - e + 15) >> 4 : 0;
                if (MODE == 2) { nR = (w0 + 15) >> 4; nZ = (n + 15) >> 4; }
                S = nL + nR + nZ;
                g0 = gx * (4 * spw);
                if (g0 >= S) continue;
            } else {
                if (row0 >= (isZ ? n : w0)) continue;
            }
            if (gx == 0 && t == 0)      // algorithmic work of this link's update, in units of 4096 complex MACs
                atomicAdd(work, (unsigned)(((long)ww * ww * (MODE == 1 ? (long)((units & 1) ? w0 : 0) + ((units & 2) ? n : 0) : (MODE == 2 ? 2L * n - ww : (long)(n > e ? n - e : 0)))) >> 12));
            const cx<T>* U = MODE == 2 ? Udense_all + (long)b * QW * QW : Ulog_all + (((long)b * nslot + qq) * kc + ch) * QW * QW;
            // Per link: (1) ALL global loads up front -- the 16 U elements of this thread and the first strip's 64 x 16 block (both only depend on
            // what this wave itself stored for the previous link) -- (2) barrier: the previous link's readers are done with the planes, (3) U
            // registers -> planes, (4) barrier, (5) MFMAs + stores.  (As a loop of load -> LDS store the staging was 16 dependent L2 round trips,
            // ~25 us per link on the critical path of the link chain: measured, profiles/r05_ab/r5i_pmc_qr_updates.txt.)
            cx<T> ureg[QW * QW / 256];
#pragma unroll
            for (int i = 0; i < QW * QW / 256; ++i) {
                const int el = t + 256 * i, k = el >> 6, c = el & 63;
                ureg[i] = U[(k < ww ? k : ww - 1) * QW + (c < ww ? c : ww - 1)];            // clamped: no branch around the load
            }
            // first strip of this wave
            SlabStrip<T> d;
            bool on = false;
            if (MODE == 1) {
                d.X = isZ ? Z : H; d.side = 1; d.a0 = row0 + 16 * wave; d.lim = isZ ? n : w0;
                on = d.a0 < d.lim;
            } else {
                const int g = g0 + wave;
                on = g < S;
                if (MODE == 0 || g < nL) { d.X = H; d.side = 0; d.a0 = e + 16 * g; d.lim = n; }
                else if (g < nL + nR) { d.X = H; d.side = 1; d.a0 = 16 * (g - nL); d.lim = w0; }
                else { d.X = Z; d.side = 1; d.a0 = 16 * (g - nL - nR); d.lim = n; }
            }
            cx<T> xa[8], xb[8];
            if (on) {
                slab_load_half<T, MODE == 2 ? -1 : (MODE == 1 ? 1 : 0)>(d, 0, n, w0, ww, lane, xa);
                slab_load_half<T, MODE == 2 ? -1 : (MODE == 1 ? 1 : 0)>(d, 1, n, w0, ww, lane, xb);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (staged) __syncthreads();                          // every wave is done with the previous link's planes
            int dense = 0;                                        // any nonzero in the blocks the banded product skips?
#pragma unroll
            for (int i = 0; i < QW * QW / 256; ++i) {
                const int el = t + 256 * i, k = el >> 6, c = el & 63;
                cx<T> u = ureg[i];
                if (k >= ww || c >= ww) u = cx<T>(T(0), T(0));
                Ur[k * MLD + c] = u.x; Ui[k * MLD + c] = u.y;
                if ((k >> 4) >= (c >> 4) + 2 && (u.x != T(0) || u.y != T(0))) dense = 1;
            }
            { const int wd = __any(dense); if (lane == 0) wdense[t >> 6] = wd; }
            __syncthreads();
            staged = true;
            const bool band = MODE != 2 && band_on && !(wdense[0] | wdense[1] | wdense[2] | wdense[3]);
            if (on) {
                if (MODE == 1) {
                    if (band) slab_compute<T, 1, true>(Ur, Ui, d, n, w0, ww, lane, xa, xb);
                    else slab_compute<T, 1, false>(Ur, Ui, d, n, w0, ww, lane, xa, xb);
                } else if (MODE == 0) {
                    if (band) slab_compute<T, 0, true>(Ur, Ui, d, n, w0, ww, lane, xa, xb);
                    else slab_compute<T, 0, false>(Ur, Ui, d, n, w0, ww, lane, xa, xb);
                } else {
                    if (d.side == 0) slab_compute<T, 0, false>(Ur, Ui, d, n, w0, ww, lane, xa, xb);
                    else slab_compute<T, 1, false>(Ur, Ui, d, n, w0, ww, lane, xa, xb);
                }
            }
            // further strips of this wave (MODE 0 / 2 with more than one strip per wave)
            if (MODE != 1) {
                for (int i = 1; i < spw; ++i) {
                    const int g = g0 + wave + 4 * i;
                    if (g >= S) break;
                    SlabStrip<T> d2;
                    if (MODE == 0 || g < nL) { d2.X = H; d2.side = 0; d2.a0 = e + 16 * g; d2.lim = n; }
                    else if (g < nL + nR) { d2.X = H; d2.side = 1; d2.a0 = 16 * (g - nL); d2.lim = w0; }
                    else { d2.X = Z; d2.side = 1; d2.a0 = 16 * (g - nL - nR); d2.lim = n; }
                    if (MODE == 0) {
                        if (band) slab_strip<T, 0, true>(Ur, Ui, d2, n, w0, ww, lane);
                        else slab_strip<T, 0>(Ur, Ui, d2, n, w0, ww, lane);
                    } else {
                        if (d2.side == 0) slab_strip<T, 0>(Ur, Ui, d2, n, w0, ww, lane);
                        else slab_strip<T, 1>(Ur, Ui, d2, n, w0, ww, lane);
                    }
                }
            }
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
    int groups = 0, spw = 0, aed = 0, nibble = 100, moves = QAED_MOVES, chains = 0, band = 0, rotb = 0, super = 0, defer = 0, fuse = 0;
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
        q.band = geti("TRX_SLAB_BAND", 0, 2, 0);              // 0 / 2: skip the structurally zero blocks of a chain unitary, 1: dense product always
        q.rotb = geti("TRX_QR_ROTB", 0, 1, 0);               // 1: rotations of the in-LDS Schur solver broadcast by ds_bpermute (round-3 code), else v_readlane
        q.super = geti("TRX_QR_SUPER", 1, QSUPER, 0);       // window steps per launch (fp32, one chain per sweep); 0 = automatic
        q.defer = geti("TRX_QR_DEFER", 0, 2, 0);              // right / Z update: 0 automatic, 1 behind every (super-)step, 2 once per sweep
        q.fuse = geti("TRX_QR_FUSE", 0, 2, 0);                // fused launches (chase + far part of the previous launch's left update): 0 automatic (on), 1 off, 2 on
        q.debug = getenv("TRX_QR_DEBUG") != nullptr;
        return q;
    }();
    return k;
}

// Bulge chains per sweep and slots of the link log: the workspace layout (eig.hip) and the solver must agree on both.
// Bulge chains per sweep.  Automatic: 3 for one or two matrices, 2 up to batch 48, 1 above.  One chain per sweep is the form with super-steps,
// fused launches and the right / Z update once per sweep -- throughput machinery that pays when the batch fills the chip; below that the QR
// phase is its chain of dependent launches, and two chains (32 shifts per sweep, AED window 64) halve the number of sweeps.  Measured in round 6
// (layer-solves/s, one chain / two chains with the 64-wide window): batch 4: 4.03 / 5.24, 8: 10.43 / 11.73, 16: 16.76 / 17.85, 24: 20.95 / 21.83,
// 32: 24.13 / 24.89, 48: 27.91 / 28.04 (profiles/r06_ab/r6p_chains_small_batches.txt, r6q_chains_aed_small_batches.txt).
int qr_chains_for(int batch) { const QrKnobs& K = qr_knobs(); return K.chains ? K.chains : (batch <= 2 ? QKC : (batch <= 48 ? 2 : 1)); }
int qr_log_slots(int n) { return cdiv_i(n + 2 * QNS, QW - 2 * QNS - 1) + 2 + 4 * (QKC - 1) + 1; }

// Non-blocking streams and timing-less events for the iteration groups, created on first use and kept for the life of the
// process (per device); a call checks out what it needs and hands it back, so concurrent callers never share one.
struct QrLane {
    hipStream_t s = nullptr;           // the group's stream (null: the caller's)
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
    else if (s == "qr_rotb") { slot = &k.rotb; hi = 1; }
    else if (s == "qr_chains") { slot = &k.chains; hi = QKC; }
    else if (s == "slab_band") { slot = &k.band; hi = 2; }
    else if (s == "qr_super") { slot = &k.super; hi = QSUPER; }
    else if (s == "qr_defer") { slot = &k.defer; hi = 2; }
    else if (s == "qr_fuse") { slot = &k.fuse; hi = 2; }
    else return TRX_ERR_ARG;
    if (value < lo || value > hi || (slot == &k.spw && value == 3) || (slot == &k.aed && value != 0 && value < 16)) return TRX_ERR_ARG;
    *slot = value;
    return TRX_OK;
}

template <class T>
int hessenberg_qr(hipStream_t s, const EigBuffers<T>& B, int n, int batch, int* info) {
    constexpr int LD = QW + 1;
    if ((double)n * n * sizeof(cx<T>) >= 4294967296.0) return TRX_ERR_ARG;      // update kernels: 32-bit byte offsets inside one matrix
    const QrKnobs& K = qr_knobs();
    const size_t smw = sizeof(cx<T>) * QW * LD + sizeof(RotCS<T>) * WMAXS * QNS + sizeof(QrState) + 16;      // + the band flag
    const int kc = qr_chains_for(batch);
    const int nslot = qr_log_slots(n);
    // update kernel: the two planes of U, the four per-wave band votes, the launch's link records
    const size_t sma = sizeof(T) * 2 * QW * MLD + 16 + sizeof(QrLink) * (size_t)nslot * kc;
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
            const size_t sma_max = 160 * 1024 - 512;     // (the link records depend on n: opt in to the largest size once)
            const int r = set_max_dyn_smem((const void*)qr_window_kernel<T, false>, smw) || set_max_dyn_smem((const void*)qr_window_kernel<T, true>, smw) ||
                  set_max_dyn_smem((const void*)apply_links_kernel<T, 0>, sma_max) || set_max_dyn_smem((const void*)apply_links_kernel<T, 1>, sma_max) ||
                  set_max_dyn_smem((const void*)apply_links_kernel<T, 2>, sma_max) ||
                  set_max_dyn_smem((const void*)qr_prepare_kernel<T, false>, smp_of(SM)) || set_max_dyn_smem((const void*)qr_prepare_kernel<T, true>, smp_of(SM));
            stt = r ? 2 : 1;
        }
        attr_rc = stt == 2;
    }
    if (attr_rc) return TRX_ERR_LAUNCH;
    TRX_LAUNCH((qr_init_kernel<T>), dim3(batch), dim3(64), 0, s, B.st, n);
    if (hipMemsetAsync(B.summary, 0, sizeof(int) * 128, s) != hipSuccess) return TRX_ERR_LAUNCH;
    const int max_sweeps = 30 * n + 100;
    // strips per wave of the per-step left update: small batches are latency bound and keep the shorter per-launch chain
    const int spw = K.spw ? K.spw : 1;          // measured: 1 is best at every batch size (32.9 vs 32.4 layer-solves/s at batch 128 with 2)
    const int nstrip = cdiv_i(n, 16);
    const int adv = QW - 2 * QNS - 1;                        // guaranteed chain advance per window step
    // Bulge chains per sweep.  Measured on MI355X (n = 1922): 2 / 3 chains cut the outer iterations by only 31 / 36 % (the AED's
    // deflation yield, not the shift count, paces the iteration) while the update work grows by a third: one chain is the default for
    // batches; a single large matrix (the topology-optimisation case, n = 5202, batch 1) has no update work to protect and gains from the
    // shorter chain: 5.62 s (1 chain, AED 48) -> 5.16 s (3 chains) -> 4.79 s (3 chains, AED 64) for the whole forward solve
    // (profiles/r02_single_matrix_knobs.txt).
    const int band_on = K.band != 1;
    // window steps per launch: super-steps need one chain per sweep (several chains advance in lock-step, one window step per launch)
    const int super = kc == 1 ? (K.super ? K.super : (sizeof(T) == 4 ? 4 : 8)) : 1;          // measured: fp32 4 (2 / 8 within 0.5 %), fp64 8 (28.4 vs 28.1 layer-solves/s on the all-fp64 route)
    const bool defer = kc == 1 && K.defer != 1;
    const bool fuse = defer && K.fuse != 1;                  // fused launches: see qr_window_kernel
    static const int fspw_env = [] { const char* e = getenv("TRX_QR_FSPW"); const int v = e ? atoi(e) : 0; return (v >= 0 && v <= 8) ? v : 0; }();
    // Strips per wave of a far workgroup (TRX_QR_FSPW, environment only).  Measured at batch 128 (layer-solves/s; QR phase): 1 strip 36.85 (895 ms), 2:
    // 37.40 (844 ms), 4: 37.82 (810 ms); batch 64: 31.81 / 31.86 / 31.97 -- fewer, longer far workgroups stop queueing for the compute units the
    // chase workgroups of the four groups leave, and still end before the chase does (profiles/r06_ab/r6za_far_strips_per_wave.txt).
    const int fspw = fspw_env ? fspw_env : 4;
    const int far_wgs = cdiv_i(cdiv_i(n, 16), fspw * (WTHREADS / 64));       // far workgroups per matrix: 16 x fspw strips each, all strips in one pass
    // several chains per sweep: the right / Z update of a window step rides in the NEXT step's chase launch (qr_window_kernel; knob qr_fuse: 1 off).
    // Measured (layer-solves/s, own launch / riding with 1 / 2 / 4 strips per rider wave): batch 8: 11.81 / 12.49 / 12.49 / 11.62, 16: 18.38 / 19.07 /
    // 19.49 / 18.44, 24: 22.36 / 22.16 / 23.89 / 22.75, 32: 25.46 / 24.89 / 26.83 / 25.79, 48: 28.63 / 27.81 / 29.14 / 29.22, config 5 (one n = 5202
    // matrix, three chains): 4.11 / 3.77 / 3.75 / 4.53 s per step -- one strip per wave floods the chip with 1024-thread riders once many matrices
    // are in flight, four leave the riders on the launch's critical path (profiles/r06_ab/r6y..., r6z...).
    const bool rfuse = kc > 1 && !defer && K.fuse != 1;
    static const int rspw_env = [] { const char* e = getenv("TRX_QR_RSPW"); const int v = e ? atoi(e) : 0; return (v >= 0 && v <= 8) ? v : 0; }();
    const int rspw = rspw_env ? rspw_env : 2;                        // strips per wave of a rider (TRX_QR_RSPW, environment only)
    const int rz_wgs = cdiv_i(2 * cdiv_i(n, 16), rspw * (WTHREADS / 64));    // riders per chain: [H rows above the window | Z rows], 16 x rspw strips each

    // The batch is split into groups that iterate out of phase on their own streams: the latency-bound kernels of one group
    // (AED / shift preparation: one wave per matrix; window chase: one workgroup per matrix and chain) run while the updates
    // of the other groups fill the matrix cores.  One host thread drives all of them; per visit it reads the 12-byte
    // summary of the group's last prepare (normally long finished: the prepare is queued right behind the group's sweep),
    // queues the next sweep and the prepare after it, and moves on.
    constexpr int MAXG = 8;
    struct Group {
        QrLane lane;           // streams + events
        hipStream_t s;
        int b0, nb;
        int* summary;          // device, 16 ints: two slots of {[0] active matrices, [1] bound of the remaining blocks, [2] flags}; [3] / [7]: work of the
                               // per-step / the deferred updates (cumulative, in units of 4096 complex MACs); [8] chain steps of the window kernel, [9] rotations of the AED's Schur solver,
                               // [10] left-update work done INSIDE the window launches (far workgroups + catch-up of the fused launches; same units as [3])
        bool done;
        unsigned work[5];
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
        G.summary = B.summary + 16 * g;
        G.done = false;
        G.issued = 0;
        G.read = 0;
        G.work[0] = G.work[1] = G.work[2] = G.work[3] = G.work[4] = 0;
        G.par = 0;
        G.g = g;
        if (!lane_checkout(dev, g > 0, G.lane)) { rc = TRX_ERR_LAUNCH; break; }
        ++nlanes;
        G.s = g == 0 ? s : G.lane.s;
        if (g > 0 && hipStreamWaitEvent(G.s, fork.ev, 0) != hipSuccess) { rc = TRX_ERR_LAUNCH; break; }
    }
    // AED window: 64 deflates most per call (fewest sweeps, least update work) but costs 3 ms of single-wave latency; at small
    // batches, where nothing is throughput bound, a smaller window shortens the chain
    const long mstride = (long)ngroups * n * n;              // distance between consecutive matrices of one group
    const int aed_w = K.aed ? K.aed : ((batch >= 64 || kc >= 2) ? QAED : 48);     // measured: batch 128: 64 -> 28.4, 48 -> 27.8 solves/s; batch 16: 48; batch 1 with 3 chains: 64
    const size_t smp = smp_of(aed_w);
    // LAPACK skips the sweep when AED deflated more than 14 % of the window ("nibble") because there the sweep is the expensive
    // part.  Here the AED is, so every AED that leaves an active block is followed by a sweep in the same iteration.
    const int nibble = K.nibble, aed_moves = K.moves;
    const bool qr_debug = K.debug;                                        // cycle breakdown of the prepare / window kernels (matrix 0) to stderr
    long long* dbg_dev = reinterpret_cast<long long*>(B.summary + 128);   // 24 counters behind the group summaries
    // One outer iteration of a group = the window steps of its sweep followed by the next prepare (deflation scan, AED, shifts)
    // and the copy of that prepare's 12-byte summary into pinned host memory.  The host runs ONE ITERATION AHEAD of what it has
    // read: iteration k+1 is queued with a step count from summary k-1 -- summary[1] is an upper bound (ihi + 1) for every block
    // the matrices of the group can still work on, and it only shrinks; a step a chain does not need is a kernel that exits at
    // once.  So a stream never drains while the host is busy with another group.
    auto issue_prepare = [&](Group& G, int slot) -> bool {
        int* sum = G.summary + 4 * slot;
        if (hipMemsetAsync(sum, 0, sizeof(int) * 3, G.s) != hipSuccess) return false;            // G.summary[3] (update work) keeps accumulating
        { ProfScope prof(PROF_QR_PREPARE, G.s, 0, 0);
          if (K.rotb == 1)
              TRX_LAUNCH((qr_prepare_kernel<T, false>), dim3(G.nb), dim3(64), smp, G.s, B.A + (long)G.g * n * n, mstride, n, B.st + G.b0, B.U + (long)G.b0 * QW * QW,
                         B.shifts + (long)G.b0 * QKC * QNS, sum, max_sweeps, aed_w, nibble, aed_moves, G.par, kc, aed_w, G.summary + 8,
                         (qr_debug && G.b0 == 0) ? dbg_dev : (long long*)nullptr);
          else
              TRX_LAUNCH((qr_prepare_kernel<T, true>), dim3(G.nb), dim3(64), smp, G.s, B.A + (long)G.g * n * n, mstride, n, B.st + G.b0, B.U + (long)G.b0 * QW * QW,
                         B.shifts + (long)G.b0 * QKC * QNS, sum, max_sweeps, aed_w, nibble, aed_moves, G.par, kc, aed_w, G.summary + 8,
                         (qr_debug && G.b0 == 0) ? dbg_dev : (long long*)nullptr); }
        if (hipMemcpyAsync(G.lane.hsum + 4 * slot, sum, sizeof(int) * 3, hipMemcpyDeviceToHost, G.s) != hipSuccess) return false;
        return hipEventRecord(G.lane.evs[slot], G.s) == hipSuccess;
    };
    auto issue_sweep = [&](Group& G, int bound) -> bool {
        cx<T>* Ag = B.A + (long)G.g * n * n;
        cx<T>* Zg = B.Z + (long)G.g * n * n;
        const cx<T>* Ud = B.U + (long)G.b0 * QW * QW;
        cx<T>* Ul = B.Ulog + (long)G.b0 * nslot * kc * QW * QW;
        QrLink* lk = B.links + (long)G.b0 * nslot * kc;
        const cx<T>* shg = B.shifts + (long)G.b0 * QKC * QNS;
        QrState* stg = B.st + G.b0;
        // window steps: the first chain needs (m + 2 QNS) / adv steps (+ one slot that applies the AED unitary); every further chain
        // enters about 3 steps behind the one ahead.  `bound` is one iteration old, i.e. already a step or so generous, and a sweep
        // that still falls short is finished by the next iteration's steps (flag 2 of the summary).
        int nwin = bound > 0 ? cdiv_i(bound + 2 * QNS, adv) + 2 + 4 * (kc - 1) : 1;
        if (nwin > nslot) nwin = nslot;
        unsigned* wk = (unsigned*)(G.summary + 3);
        // slot 0 on its own: it may carry the dense link of an AED window / a finished block (all sides at once: up to 3 n / 16 + 3 strips),
        // which must be applied before the chase starts; then super-steps of `super` window steps + ONE left update over their links
        int prev_q = 0, prev_ns = 0;
        for (int q = 0; q < nwin;) {
            const int ns = q == 0 ? 1 : (nwin - q < super ? nwin - q : super);
            { ProfScope p(PROF_QR_WINDOW, G.s, 0, 0);
              const bool dbgk = qr_debug && G.b0 == 0;
              long long* dbp = dbgk ? dbg_dev : (long long*)nullptr;
              // fused launch: the far part of the previous launch's left update rides along (not behind slot 0, whose left update is a launch of its own)
              // several chains: the right / Z update of the previous step rides along instead (rz_wgs workgroups per chain)
              const bool ride = (kc == 1 ? fuse : rfuse) && q > 1;
              const int pq = ride ? prev_q : 0, pn = ride ? prev_ns : 0;
              const int gx = kc + (pn > 0 ? (kc == 1 ? far_wgs : kc * rz_wgs) : 0);
              if (dbgk) TRX_LAUNCH((qr_window_kernel<T, true>), dim3(gx, G.nb), dim3(WTHREADS), smw, G.s, Ag, mstride, n, stg, Ul, lk, shg, G.par, nslot, kc, q, ns, band_on, G.summary + 8, dbp, pq, pn, wk, Zg);
              else TRX_LAUNCH((qr_window_kernel<T, false>), dim3(gx, G.nb), dim3(WTHREADS), smw, G.s, Ag, mstride, n, stg, Ul, lk, shg, G.par, nslot, kc, q, ns, band_on, G.summary + 8, dbp, pq, pn, wk, Zg);
            }
            G.par ^= (ns & 1);
            { ProfScope p(PROF_QR_APPLY_LEFT, G.s, 0, 0);
              if (q == 0) {      // the dense link of slot 0, if there is one: left | right-H | Z, up to 3 n / 16 + 3 strips
                  const int du = cdiv_i(3 * nstrip + 3, 4 * spw);
                  TRX_LAUNCH((apply_links_kernel<T, 2>), dim3(du, G.nb), dim3(256), sma, G.s, Ag, Zg, mstride, n, (const QrLink*)lk, (const cx<T>*)Ul, Ud, wk, nslot, kc, 0, 1, spw, du, band_on);
              }
              const int units = cdiv_i(nstrip + 1, 4 * spw);
              // fused: the left update of a launch's links is done by the NEXT launch (catch-up + far workgroups) or, for the last launch of
              // the sweep, by the launch behind the loop
              if (!fuse || q == 0)
                  TRX_LAUNCH((apply_links_kernel<T, 0>), dim3(kc * units, G.nb), dim3(256), sma, G.s, Ag, Zg, mstride, n, (const QrLink*)lk, (const cx<T>*)Ul, Ud, wk, nslot, kc, q, ns, spw, units, band_on);
              // several chains: the right update of H cannot wait (the following chain's windows read rows the chain ahead has passed), and Z goes
              // with it: one or two matrices have no throughput to protect, and a deferred walk over ~170 links is a 3 ms latency chain per sweep
              // (measured: config 5 4.26 s with Z deferred against 3.98 s in round 4's two-launch form)
              // (rfuse: the NEXT window launch does it; slot 0 excepted -- it may hold a dense link, whose update is a launch of its own, so the
              // launch of step 1 carries nothing, as in the one-chain form, and step 0's chase links are applied here)
              if (!defer && (!rfuse || q == 0))
                  TRX_LAUNCH((apply_links_kernel<T, 1>), dim3(2 * cdiv_i(n, 64), G.nb), dim3(256), sma, G.s, Ag, Zg, mstride, n, (const QrLink*)lk, (const cx<T>*)Ul, Ud, wk, nslot, kc, q, ns, 1, 3, band_on); }
            prev_q = q; prev_ns = ns;
            q += ns;
        }
        if (rfuse && prev_q > 0) {
            // the right / Z update of the sweep's last step
            ProfScope p(PROF_QR_APPLY_LEFT, G.s, 0, 0);
            TRX_LAUNCH((apply_links_kernel<T, 1>), dim3(2 * cdiv_i(n, 64), G.nb), dim3(256), sma, G.s, Ag, Zg, mstride, n, (const QrLink*)lk, (const cx<T>*)Ul, Ud, wk, nslot, kc, prev_q, prev_ns, 1, 3, band_on);
        }
        if (fuse && prev_q > 0) {
            // the left update of the sweep's last launch
            ProfScope p(PROF_QR_APPLY_LEFT, G.s, 0, 0);
            const int units = cdiv_i(nstrip + 1, 4 * spw);
            TRX_LAUNCH((apply_links_kernel<T, 0>), dim3(kc * units, G.nb), dim3(256), sma, G.s, Ag, Zg, mstride, n, (const QrLink*)lk, (const cx<T>*)Ul, Ud, wk, nslot, kc, prev_q, prev_ns, spw, units, band_on);
        }
        // deferred right / Z update of all chase links of this sweep: ONE launch.  In line on the group's stream: the prepare kernel that
        // follows may place its AED window on rows these links' updates still have to reach (the active block can end anywhere after a
        // deflation), so it cannot run beside it.
        if (defer) {
            ProfScope p(PROF_QR_APPLY_RIGHT, G.s, 0, 0);
            TRX_LAUNCH((apply_links_kernel<T, 1>), dim3(2 * cdiv_i(n, 64), G.nb), dim3(256), sma, G.s, Ag, Zg, mstride, n, (const QrLink*)lk, (const cx<T>*)Ul, Ud, wk + 4, nslot, kc, 0, nwin, 1, 3, band_on);
        }
        return true;
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
        if (!issue_sweep(G, n) || !issue_prepare(G, 1)) rc = TRX_ERR_LAUNCH;
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
            if (!issue_sweep(G, bound) || !issue_prepare(G, G.issued & 1)) { rc = TRX_ERR_LAUNCH; break; }
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
    double work[5] = {0, 0, 0, 0, 0};
    for (int g = 0; g < nlanes; ++g) {
        Group& G = grp[g];
        if (!rc && prof_enabled() && hipMemcpyAsync(&G.work[0], G.summary + 3, sizeof(unsigned), hipMemcpyDeviceToHost, G.s) == hipSuccess &&
            hipMemcpyAsync(&G.work[1], G.summary + 7, sizeof(unsigned), hipMemcpyDeviceToHost, G.s) == hipSuccess &&
            hipMemcpyAsync(&G.work[2], G.summary + 8, 3 * sizeof(unsigned), hipMemcpyDeviceToHost, G.s) == hipSuccess && hipStreamSynchronize(G.s) == hipSuccess) {
            for (int i = 0; i < 5; ++i) work[i] += G.work[i];
        }
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
    if (prof_enabled()) {      // algorithmic work of the update kernels: 8 flops per complex MAC, every element of a 64-wide slab read and written once
        prof_add_work(PROF_QR_APPLY_LEFT, 8.0 * 4096.0 * work[0], 2.0 * sizeof(cx<T>) * 4096.0 * work[0] / QW);
        prof_add_work(PROF_QR_APPLY_RIGHT, 8.0 * 4096.0 * work[1], 2.0 * sizeof(cx<T>) * 4096.0 * work[1] / QW);
        // latency kernels: dependent steps per MATRIX (the matrices of a launch run side by side): chain steps / rotations summed over all matrices / batch
        // (the window tag's BYTES slot carries the flops of the left updates its launches did themselves -- far workgroups and catch-up of the
        // fused launches; the left-update tag holds only what the stand-alone launches did, so that its work and its event time match)
        prof_add_work(PROF_QR_WINDOW, work[2] / batch, 8.0 * 4096.0 * work[4]);
        prof_add_work(PROF_QR_PREPARE, work[3] / batch, 0.0);
    }
    TRX_LAUNCH((qr_collect_info_kernel<T>), dim3(cdiv_i(batch, 64)), dim3(64), 0, s, (const QrState*)B.st, info, batch, ngroups);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

template int hessenberg_qr<float>(hipStream_t, const EigBuffers<float>&, int, int, int*);
template int hessenberg_qr<double>(hipStream_t, const EigBuffers<double>&, int, int, int*);

}  // namespace trx
