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
//   qr_prepare_kernel  (1 wave / matrix)  deflation scan, active-block bookkeeping, shifts = eigenvalues of the
//                                         trailing k x k block (in-LDS single-shift QR); blocks <= QNMIN are
//                                         finished here by the same in-LDS QR with U accumulated.
//   qr_window_kernel   (256 thr / matrix) one window step of the bulge chain.
//   apply_window_kernel  H[w0:w1, w1:n] <- U^H H[w0:w1, w1:n] ;  H[0:w0, w0:w1] <- H[0:w0, w0:w1] U ;
//                        Z[:, w0:w1] <- Z[:, w0:w1] U        (one launch, disjoint slabs, MFMA)
#include "eig.hpp"
#include "mfma.hpp"
#include "prof.hpp"

namespace trx {
namespace {

constexpr int QW = EigPlan::QW, QNS = EigPlan::QNS, QNMIN = EigPlan::QNMIN;
constexpr int SLD = QNMIN + 1;    // leading dimension of the small in-LDS matrices

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

// Fast rotation generator for the chase (no hypot/divide chain: two rsqrt).  |f|^2+|g|^2 cannot overflow here: the
// entries of a balanced RCWA operator are O(1e3) and negligible subdiagonals were flushed to zero by the deflation scan.
template <class T>
__device__ __forceinline__ Rot<T> rotg_fast(cx<T> f, cx<T> g) {
    Rot<T> R;
    const T ag2 = norm2(g), af2 = norm2(f);
    if (ag2 == T(0)) { R.c = T(1); R.s = cx<T>(T(0), T(0)); R.r = f; return R; }
    if (af2 == T(0)) { const T ig = rsqrt(ag2); R.c = T(0); R.s = ig * conj(g); R.r = cx<T>(ag2 * ig, T(0)); return R; }
    const T n2 = af2 + ag2;
    const T u = rsqrt(n2), tt = rsqrt(af2);        // 1/d, 1/|f|
    R.c = af2 * tt * u;                            // |f| / d
    R.s = (tt * u) * (f * conj(g));                // (f/|f|) conj(g) / d
    R.r = (n2 * u * tt) * f;                       // (f/|f|) d
    return R;
}

// Single-shift QR (Wilkinson shift) of an m x m (m <= QNMIN) upper Hessenberg matrix held in LDS, executed by ONE
// wave (blockDim.x == 64).  On return Hs is upper triangular (Schur form); if Us != nullptr it holds U with
// H_in = U T U^H.  Returns false if it did not converge.
template <class T>
__device__ bool small_schur(cx<T>* Hs, int m, cx<T>* Us) {
    const int lane = threadIdx.x;
    const T ulp = eps_of<T>::value;
    int ihi = m - 1, its = 0, total = 0;
    while (ihi > 0) {
        int l = ihi;
        while (l > 0) {
            T s = abs1(Hs[(l - 1) * SLD + l - 1]) + abs1(Hs[l * SLD + l]);
            if (s == T(0)) s = T(1);
            if (abs1(Hs[l * SLD + l - 1]) <= ulp * s) break;
            --l;
        }
        __syncthreads();
        if (l > 0 && lane == 0) Hs[l * SLD + l - 1] = cx<T>(T(0), T(0));
        __syncthreads();
        if (l == ihi) { --ihi; its = 0; continue; }
        ++its; ++total;
        if (total > 40 * m) return false;
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
        for (int p = l; p < ihi; ++p) {
            cx<T> f, g;
            if (p == l) { f = Hs[l * SLD + l] - sig; g = Hs[(l + 1) * SLD + l]; }
            else { f = Hs[p * SLD + p - 1]; g = Hs[(p + 1) * SLD + p - 1]; }
            const Rot<T> R = rotg_fast(f, g);
            __syncthreads();
            if (lane < 32) {
                const int lo = (p == l) ? p : p - 1;
                const int col = lo + lane;
                if (col < m) {
                    cx<T> x = Hs[p * SLD + col], y = Hs[(p + 1) * SLD + col];
                    rot_rows(R, x, y);
                    if (p != l && col == p - 1) { x = R.r; y = cx<T>(T(0), T(0)); }
                    Hs[p * SLD + col] = x; Hs[(p + 1) * SLD + col] = y;
                }
            } else if (Us) {
                const int row = lane - 32;
                if (row < m) {
                    cx<T> x = Us[row * SLD + p], y = Us[row * SLD + p + 1];
                    rot_cols(R, x, y);
                    Us[row * SLD + p] = x; Us[row * SLD + p + 1] = y;
                }
            }
            __syncthreads();
            if (lane < 32) {
                const int hi = (p + 2 < ihi) ? p + 2 : ihi;
                const int row = lane;
                if (row <= hi) {
                    cx<T> x = Hs[row * SLD + p], y = Hs[row * SLD + p + 1];
                    rot_cols(R, x, y);
                    Hs[row * SLD + p] = x; Hs[row * SLD + p + 1] = y;
                }
            }
            __syncthreads();
        }
    }
    return true;
}

// Swap the adjacent diagonal entries k, k+1 of the upper-triangular Ts (order m) by one rotation, accumulating into Vs
// (LAPACK ztrexc for complex Schur forms).  One wave.
template <class T>
__device__ void schur_swap(cx<T>* Ts, cx<T>* Vs, int m, int k) {
    const int lane = threadIdx.x;
    const cx<T> a = Ts[k * SLD + k], bq = Ts[(k + 1) * SLD + k + 1], x = Ts[k * SLD + k + 1];
    const Rot<T> R = rotg_fast(x, bq - a);
    __syncthreads();
    if (lane < 32) {
        const int col = k + lane;
        if (col < m) {
            cx<T> u = Ts[k * SLD + col], v = Ts[(k + 1) * SLD + col];
            rot_rows(R, u, v);
            Ts[k * SLD + col] = u; Ts[(k + 1) * SLD + col] = v;
        }
    } else {
        const int row = lane - 32;
        if (row < m) {
            cx<T> u = Vs[row * SLD + k], v = Vs[row * SLD + k + 1];
            rot_cols(R, u, v);
            Vs[row * SLD + k] = u; Vs[row * SLD + k + 1] = v;
        }
    }
    __syncthreads();
    if (lane <= k + 1) {
        cx<T> u = Ts[lane * SLD + k], v = Ts[lane * SLD + k + 1];
        rot_cols(R, u, v);
        if (lane == k + 1) u = cx<T>(T(0), T(0));
        Ts[lane * SLD + k] = u; Ts[lane * SLD + k + 1] = v;
    }
    __syncthreads();
}

// Householder reflector for x[0:len] held in LDS at stride `inc` (LAPACK zlarfg): H = I - tau v v^H, H^H x = beta e1.
// All lanes compute redundantly; v (v[0] = 1) is written to vw[0:len] by the lanes; returns tau, beta.
template <class T>
__device__ void small_larfg(const cx<T>* x, int inc, int len, cx<T>* vw, cx<T>& tau, T& beta) {
    const int lane = threadIdx.x;
    const cx<T> alpha = x[0];
    T xn2 = T(0);
    for (int i = 1; i < len; ++i) xn2 += norm2(x[i * inc]);
    cx<T> scale;
    if (xn2 == T(0) && alpha.y == T(0)) {
        tau = cx<T>(T(0), T(0)); beta = alpha.x; scale = cx<T>(T(0), T(0));
    } else {
        const T nrm = sqrt(norm2(alpha) + xn2);
        beta = (alpha.x >= T(0)) ? -nrm : nrm;
        tau = cx<T>((beta - alpha.x) / beta, -alpha.y / beta);
        scale = crecip(cx<T>(alpha.x - beta, alpha.y));
    }
    __syncthreads();
    if (lane < len) vw[lane] = (lane == 0) ? cx<T>(T(1), T(0)) : x[lane * inc] * scale;
    __syncthreads();
}

// Apply H = I - tau v v^H (v on rows/cols [o, o+len)) to the m x m Ts from both sides (Ts <- H^H Ts H, left side on
// columns >= c0) and to Vs from the right.  One wave: lanes 0..31 own a column (left) / a row of Ts (right), lanes
// 32..63 a row of Vs.
template <class T>
__device__ void small_apply_reflector(cx<T>* Ts, cx<T>* Vs, int m, int nrows_t, int o, int len, int c0, const cx<T>* vw, cx<T> tau) {
    const int lane = threadIdx.x;
    if (lane < 32) {
        const int c = lane;
        if (c >= c0 && c < m) {
            cx<T> w(T(0), T(0));
            for (int i = 0; i < len; ++i) cfma_conj(w, vw[i], Ts[(o + i) * SLD + c]);
            const cx<T> f = conj(tau) * w;
            for (int i = 0; i < len; ++i) Ts[(o + i) * SLD + c] -= vw[i] * f;
        }
    }
    __syncthreads();
    {
        cx<T>* Mx = (lane < 32) ? Ts : Vs;
        const int r = (lane < 32) ? lane : lane - 32;
        const int lim = (lane < 32) ? nrows_t : m;
        if (r < lim) {
            cx<T> w(T(0), T(0));
            for (int i = 0; i < len; ++i) cfma(w, Mx[r * SLD + o + i], vw[i]);
            const cx<T> f = tau * w;
            for (int i = 0; i < len; ++i) Mx[r * SLD + o + i] -= f * conj(vw[i]);
        }
    }
    __syncthreads();
}

template <class T>
__global__ __launch_bounds__(64) void qr_init_kernel(QrState* __restrict__ st, int n) {
    if (threadIdx.x == 0) {
        QrState s;
        s.ilo = 0; s.ihi = n - 1; s.k = 0; s.tau = 0; s.tau_last = -1; s.mode = QR_IDLE; s.stall = 0; s.sweeps = 0;
        s.w0 = 0; s.w1 = 0; s.fail = 0; s.pad = 0;
        st[blockIdx.x] = s;
    }
}

template <class T>
__global__ __launch_bounds__(64) void qr_prepare_kernel(cx<T>* __restrict__ Aall, int n, QrState* __restrict__ stall_,
                                                        cx<T>* __restrict__ Uall, cx<T>* __restrict__ shifts_all,
                                                        int* __restrict__ summary, int max_sweeps) {
    __shared__ cx<T> Hs[QNMIN * SLD];
    __shared__ cx<T> Us[QNMIN * SLD];
    __shared__ cx<T> vwork[QNMIN];
    __shared__ cx<T> wk[QNMIN];
    __shared__ QrState sst;
    const int b = blockIdx.x, lane = threadIdx.x;
    cx<T>* H = Aall + (long)b * n * n;
    if (lane == 0) sst = stall_[b];
    __syncthreads();
    QrState st = sst;
    if (st.mode == QR_DONE) return;
    const T ulp = eps_of<T>::value;
    // 1. deflation scan (negligible subdiagonals -> exact zeros)
    for (int i = 1 + lane; i <= st.ihi; i += 64) {
        const cx<T> sub = H[(long)i * n + i - 1];
        if (sub.x != T(0) || sub.y != T(0)) {
            T s = abs1(H[(long)(i - 1) * n + i - 1]) + abs1(H[(long)i * n + i]);
            if (s == T(0)) s = T(1);
            if (abs1(sub) <= ulp * s) H[(long)i * n + i - 1] = cx<T>(T(0), T(0));
        }
    }
    __syncthreads();
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
        if (lane == 0) { st.ihi = 0; st.mode = QR_DONE; st.w0 = st.w1 = 0; stall_[b] = st; }
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
        const bool ok = small_schur<T>(Hs, m, Us);
        __syncthreads();
        cx<T>* U = Uall + (long)b * QW * QW;
        for (int e = lane; e < m * m; e += 64) {
            const int r = e / m, c = e - r * m;
            H[(long)(ilo + r) * n + ilo + c] = (r <= c) ? Hs[r * SLD + c] : cx<T>(T(0), T(0));
            U[r * QW + c] = Us[r * SLD + c];
        }
        if (lane == 0) {
            st.w0 = ilo; st.w1 = ihi + 1; st.mode = QR_SMALL_PENDING; st.ihi = ilo - 1; st.ilo = 0; st.stall = 0;
            if (!ok) st.fail += m;
            stall_[b] = st;
            atomicAdd(&summary[0], 1);
            atomicOr(&summary[2], 1);
        }
        return;
    }
    // ---- aggressive early deflation on the trailing nw x nw window (Braman/Byers/Mathias; LAPACK zlaqr3) ----------
    if (st.sweeps >= max_sweeps) {
        if (lane == 0) { st.fail += ihi + 1; st.mode = QR_DONE; st.w0 = st.w1 = 0; stall_[b] = st; }
        return;
    }
    const int nw = QNMIN;                       // m > QNMIN here, so the window is strictly inside the active block
    const int kw = ihi - nw + 1;
    cx<T> spike = H[(long)kw * n + kw - 1];
    for (int e = lane; e < nw * nw; e += 64) {
        const int r = e / nw, c = e - r * nw;
        Hs[r * SLD + c] = (r <= c + 1) ? H[(long)(kw + r) * n + kw + c] : cx<T>(T(0), T(0));
        Us[r * SLD + c] = cx<T>(r == c ? T(1) : T(0), T(0));
    }
    __syncthreads();
    const bool okw = small_schur<T>(Hs, nw, Us);
    __syncthreads();
    int ns = nw;
    if (okw) {
        const T smlnum = eps_of<T>::safmin * ((T)n / ulp);
        int ilst = 0;
        while (ilst < ns) {
            T foo = abs1(Hs[(ns - 1) * SLD + ns - 1]);
            if (foo == T(0)) foo = abs1(spike);
            const T sp = abs1(spike) * abs1(Us[ns - 1]);            // |s| |V[0, ns-1]|
            const T thr = (smlnum > ulp * foo) ? smlnum : ulp * foo;
            if (sp <= thr) {
                --ns;                                                // deflatable
            } else {
                for (int kk = ns - 2; kk >= ilst; --kk) schur_swap<T>(Hs, Us, nw, kk);
                ++ilst;
            }
        }
    }
    const int nd = nw - ns;
    cx<T>* sh = shifts_all + (long)b * QNS;
    if (nd == 0) {
        // nothing deflates: H is untouched; the window's eigenvalues (bottom k of them) are the shifts of a full sweep
        const int k = (m / 2 < QNS) ? m / 2 : QNS;
        if (lane < k) {
            cx<T> sv = Hs[(nw - k + lane) * SLD + nw - k + lane];
            if (st.stall > 0 && (st.stall % 6) == 0) {
                const T mag = T(0.75) * cabs(H[(long)ihi * n + ihi - 1]);
                const T ang = T(6.283185307179586) * (T)lane / (T)k;
                sv = sv + cx<T>(mag * (T)cos(ang), mag * (T)sin(ang));
            }
            sh[lane] = sv;
        }
        if (lane == 0) {
            st.k = k; st.tau = 0; st.tau_last = (ihi - 1 - ilo) + 2 * (k - 1); st.mode = QR_CHASE;
            st.stall += 1; st.sweeps += 1; st.w0 = st.w1 = 0;
            stall_[b] = st;
            atomicAdd(&summary[0], 1);
            atomicMax(&summary[1], m);
        }
        return;
    }
    // nd > 0: commit the window in Schur/Hessenberg form and publish V for the off-window update
    if (ns == 0) spike = cx<T>(T(0), T(0));
    // LAPACK's "nibble" rule: after a small deflation (< 14 % of the window) the undeflated window eigenvalues are used
    // as shifts of a sweep in the SAME outer iteration (first window slot applies V, the following ones chase).
    const int m2 = (ihi - nd) - ilo + 1;
    int kch = 0;
    if (nd * 100 < 14 * nw && m2 > QNMIN && ns >= 2) {
        kch = (m2 / 2 < QNS) ? m2 / 2 : QNS;
        if (kch > ns) kch = ns;
        if (lane < kch) sh[lane] = Hs[(ns - kch + lane) * SLD + ns - kch + lane];     // eigenvalues, before the restore below
    }
    if (ns > 1 && (spike.x != T(0) || spike.y != T(0))) {
        // reflector that maps the spike s*conj(V[0,0:ns]) onto e1, then return T[0:ns,0:ns] to Hessenberg form
        cx<T>* vw = vwork;
        if (lane < ns) wk[lane] = conj(Us[lane]);
        __syncthreads();
        cx<T> tau; T beta;
        small_larfg<T>(wk, 1, ns, vw, tau, beta);
        small_apply_reflector<T>(Hs, Us, nw, ns, 0, ns, 0, vw, tau);
        for (int jc = 0; jc + 2 < ns; ++jc) {
            small_larfg<T>(Hs + (jc + 1) * SLD + jc, SLD, ns - jc - 1, vw, tau, beta);
            if (lane == 0) Hs[(jc + 1) * SLD + jc] = cx<T>(beta, T(0));
            if (lane >= 1 && lane < ns - jc - 1) Hs[(jc + 1 + lane) * SLD + jc] = cx<T>(T(0), T(0));
            __syncthreads();
            small_apply_reflector<T>(Hs, Us, nw, ns, jc + 1, ns - jc - 1, jc + 1, vw, tau);
        }
    }
    __syncthreads();
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
            st.w0 = kw; st.w1 = ihi + 1; st.stall = 0;
            if (kch >= 2) {
                st.mode = QR_AED_CHASE;
                st.ihi = ihi - nd;
                st.k = kch; st.tau = 0; st.tau_last = (st.ihi - 1 - ilo) + 2 * (kch - 1);
                st.sweeps += 1;
                atomicMax(&summary[1], m2);
            } else {
                st.mode = QR_SMALL_PENDING;
            }
            stall_[b] = st;
            atomicAdd(&summary[0], 1);
            atomicOr(&summary[2], 1);
        }
    }
}

// One window step of the bulge chain.  Thread layout: 16 groups of 16 lanes, group s owns bulge s: its lanes
// compute the rotation redundantly (no broadcast barrier), then stride over the window's columns (left rotation)
// and, after one barrier, over its rows and the rows of U (right rotation).  Two barriers per chain step.
template <class T>
__global__ __launch_bounds__(256) void qr_window_kernel(cx<T>* __restrict__ Aall, int n, QrState* __restrict__ st_all,
                                                        cx<T>* __restrict__ Uall, const cx<T>* __restrict__ shifts_all) {
    TRX_DYN_SMEM(smem);
    constexpr int LD = QW + 1;
    cx<T>* Hw = reinterpret_cast<cx<T>*>(smem);      // [QW][LD]
    cx<T>* Uw = Hw + QW * LD;                          // [QW][LD]
    QrState& sst = *reinterpret_cast<QrState*>(Uw + QW * LD);
    const int b = blockIdx.x, t = threadIdx.x;
    if (t == 0) sst = st_all[b];
    __syncthreads();
    const QrState st = sst;
    if (st.mode == QR_SMALL_PENDING) { if (t == 0) st_all[b].mode = QR_SMALL_APPLIED; return; }
    if (st.mode == QR_AED_CHASE) { if (t == 0) st_all[b].mode = QR_CHASE; return; }      // this slot applies the AED unitary
    if (st.mode != QR_CHASE || st.tau > st.tau_last) {
        if (t == 0 && (st.w0 != 0 || st.w1 != 0)) { st_all[b].w0 = 0; st_all[b].w1 = 0; }
        return;
    }
    cx<T>* H = Aall + (long)b * n * n;
    const int k = st.k, ilo = st.ilo, ihi = st.ihi;
    int w0 = ilo + st.tau - 2 * (k - 1) - 1;
    if (w0 < ilo) w0 = ilo;
    int w1 = w0 + QW;
    if (w1 > ihi + 1) w1 = ihi + 1;
    const int ww = w1 - w0;
    const int tau_end = (w1 == ihi + 1) ? st.tau_last : (w1 - 3 - ilo);
    {
        const int c = t & 63, r4 = t >> 6;
        for (int r = r4; r < ww; r += 4) {
            if (c < ww) {
                Hw[r * LD + c] = H[(long)(w0 + r) * n + w0 + c];
                Uw[r * LD + c] = cx<T>(r == c ? T(1) : T(0), T(0));
            }
        }
    }
    const int sb = t >> 4, j = t & 15;               // bulge index, lane within the group
    const cx<T> shift = (sb < k) ? shifts_all[(long)b * QNS + sb] : cx<T>(T(0), T(0));
    __syncthreads();
    for (int tau = st.tau; tau <= tau_end; ++tau) {
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
        }
        wave_sync();                                  // all lanes have read (f, g) before any lane overwrites them
        if (active) {
            const int lo = first ? q : q - 1;
            for (int col = lo + j; col < ww; col += 16) {
                cx<T> x = Hw[q * LD + col], y = Hw[(q + 1) * LD + col];
                rot_rows(R, x, y);
                if (!first && col == q - 1) { x = R.r; y = cx<T>(T(0), T(0)); }
                Hw[q * LD + col] = x; Hw[(q + 1) * LD + col] = y;
            }
        }
        __syncthreads();
        if (active) {
            const int hi = (q + 2 < ww - 1) ? q + 2 : ww - 1;
            for (int row = j; row <= hi; row += 16) {
                cx<T> x = Hw[row * LD + q], y = Hw[row * LD + q + 1];
                rot_cols(R, x, y);
                Hw[row * LD + q] = x; Hw[row * LD + q + 1] = y;
            }
            for (int row = j; row < ww; row += 16) {
                cx<T> x = Uw[row * LD + q], y = Uw[row * LD + q + 1];
                rot_cols(R, x, y);
                Uw[row * LD + q] = x; Uw[row * LD + q + 1] = y;
            }
        }
        __syncthreads();
    }
    cx<T>* U = Uall + (long)b * QW * QW;
    {
        const int c = t & 63, r4 = t >> 6;
        for (int r = r4; r < ww; r += 4) {
            if (c < ww) {
                H[(long)(w0 + r) * n + w0 + c] = Hw[r * LD + c];
                U[r * QW + c] = Uw[r * LD + c];
            }
        }
    }
    if (t == 0) { st_all[b].tau = tau_end + 1; st_all[b].w0 = w0; st_all[b].w1 = w1; }
}

// Slab updates on the matrix cores.  Both kernels multiply a 64-wide slab by the window unitary U (ww <= 64) with
// cmma_tile_strided (mfma.hpp): operands are staged as split re/im planes in two K-chunks of 32 so that two
// workgroups fit in a CU's 160 KiB of LDS; each of the 4 waves owns 16 rows x 64 columns of the output slab.
constexpr int KC = 32;            // K chunk
constexpr int ALD = KC + 2;       // k-contiguous plane stride   (element (major,k) at [major*ALD + k])
constexpr int MLD = 72;           // major-contiguous plane stride (element (major,k) at [k*MLD + major])
constexpr int APLANE = (64 * ALD > KC * MLD) ? 64 * ALD : KC * MLD;

// H[w0:w1, cs:cs+64) <- U^H H[w0:w1, cs:cs+64),  cs = w1 + 64*blockIdx.x
template <class T>
__device__ __forceinline__ void apply_left_body(char* smem, int bx, int b, cx<T>* __restrict__ Aall, int n, const QrState* __restrict__ st_all,
                                                const cx<T>* __restrict__ Uall) {
    T* Ar = reinterpret_cast<T*>(smem);       // (U^H)[i][k] = conj(U[k][i]) : i-contiguous  [k*MLD + i]
    T* Ai = Ar + APLANE;
    T* Br = Ai + APLANE;                      // X[k][col] : col-contiguous  [k*MLD + col]
    T* Bi = Br + APLANE;
    const int w0 = st_all[b].w0, w1 = st_all[b].w1;
    const int ww = w1 - w0;
    const int cs = w1 + 64 * bx;
    if (ww <= 0 || cs >= n) return;
    const int nc = (n - cs < 64) ? n - cs : 64;
    cx<T>* H = Aall + (long)b * n * n;
    const cx<T>* U = Uall + (long)b * QW * QW;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    typename Mfma<T>::acc_t accR[4], accI[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) { accR[j][r] = T(0); accI[j][r] = T(0); }
    for (int kc = 0; kc < ww; kc += KC) {
        for (int e = t; e < KC * 64; e += 256) {
            const int kk = e >> 6, c = e & 63;
            const int k = kc + kk;
            cx<T> u(T(0), T(0)), x(T(0), T(0));
            if (k < ww) {
                if (c < ww) u = U[k * QW + c];
                if (c < nc) x = H[(long)(w0 + k) * n + cs + c];
            }
            Ar[kk * MLD + c] = u.x; Ai[kk * MLD + c] = -u.y;
            Br[kk * MLD + c] = x.x; Bi[kk * MLD + c] = x.y;
        }
        __syncthreads();
        cmma_tile_strided<T, 4>(Ar, Ai, 1, MLD, 16 * wave, Br, Bi, MLD, 1, 0, KC, accR, accI);
        __syncthreads();
    }
    const int cl = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * wave + Mfma<T>::crow(lane, r);
        if (row >= ww) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 16 * j + cl;
            if (c < nc) H[(long)(w0 + row) * n + cs + c] = cx<T>(accR[j][r], accI[j][r]);
        }
    }
}

// X[rs:rs+64, w0:w1) <- X[rs:rs+64, w0:w1) U  for X = H (rows < w0; blockIdx.x < nslab) and X = Z (all rows)
template <class T>
__device__ __forceinline__ void apply_right_body(char* smem, int bx, int b, cx<T>* __restrict__ Aall, cx<T>* __restrict__ Zall, int n, int nslab,
                                                 const QrState* __restrict__ st_all, const cx<T>* __restrict__ Uall) {
    T* Ar = reinterpret_cast<T*>(smem);       // X[row][k] : k-contiguous  [row*ALD + k]
    T* Ai = Ar + APLANE;
    T* Br = Ai + APLANE;                      // U[k][col] : col-contiguous  [k*MLD + col]
    T* Bi = Br + APLANE;
    const int w0 = st_all[b].w0, w1 = st_all[b].w1;
    const int ww = w1 - w0;
    if (ww <= 0) return;
    const bool isZ = bx >= nslab;
    const int rs = 64 * (isZ ? bx - nslab : bx);
    const int rend = isZ ? n : w0;
    if (rs >= rend) return;
    const int nr = (rend - rs < 64) ? rend - rs : 64;
    cx<T>* X = (isZ ? Zall : Aall) + (long)b * n * n;
    const cx<T>* U = Uall + (long)b * QW * QW;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    typename Mfma<T>::acc_t accR[4], accI[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) { accR[j][r] = T(0); accI[j][r] = T(0); }
    for (int kc = 0; kc < ww; kc += KC) {
        for (int e = t; e < 64 * KC; e += 256) {
            {   // X slab: lanes run along k (contiguous in global memory)
                const int row = e / KC, kk = e - row * KC;
                const int k = kc + kk;
                cx<T> x(T(0), T(0));
                if (row < nr && k < ww) x = X[(long)(rs + row) * n + w0 + k];
                Ar[row * ALD + kk] = x.x; Ai[row * ALD + kk] = x.y;
            }
            {   // U chunk: lanes run along the column
                const int kk = e >> 6, c = e & 63;
                const int k = kc + kk;
                cx<T> u(T(0), T(0));
                if (k < ww && c < ww) u = U[k * QW + c];
                Br[kk * MLD + c] = u.x; Bi[kk * MLD + c] = u.y;
            }
        }
        __syncthreads();
        cmma_tile_strided<T, 4>(Ar, Ai, ALD, 1, 16 * wave, Br, Bi, MLD, 1, 0, KC, accR, accI);
        __syncthreads();
    }
    const int cl = lane & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 16 * wave + Mfma<T>::crow(lane, r);
        if (row >= nr) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = 16 * j + cl;
            if (c < ww) X[(long)(rs + row) * n + w0 + c] = cx<T>(accR[j][r], accI[j][r]);
        }
    }
}

// One launch per window step updates everything off the window: blockIdx.x in [0, nslab) are the column slabs of the
// left update, [nslab, 3 nslab) the row slabs of the right update of H and of Z (the three regions are disjoint).
template <class T>
__global__ __launch_bounds__(256) void apply_window_kernel(cx<T>* __restrict__ Aall, cx<T>* __restrict__ Zall, int n, int nslab,
                                                           const QrState* __restrict__ st_all, const cx<T>* __restrict__ Uall) {
    TRX_DYN_SMEM(smem);
    const int b = blockIdx.y, bx = blockIdx.x;
    if (bx < nslab) apply_left_body<T>(smem, bx, b, Aall, n, st_all, Uall);
    else apply_right_body<T>(smem, bx - nslab, b, Aall, Zall, n, nslab, st_all, Uall);
}

template <class T>
__global__ void qr_collect_info_kernel(const QrState* __restrict__ st, int* __restrict__ info, int batch) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch) info[b] = st[b].fail;
}

}  // namespace

template <class T>
int hessenberg_qr(hipStream_t s, const EigBuffers<T>& B, int n, int batch, int* info) {
    constexpr int LD = QW + 1;
    const size_t sm2 = sizeof(cx<T>) * 2 * QW * LD;
    const size_t smw = sm2 + sizeof(QrState);
    const size_t sma = sizeof(T) * 4 * APLANE;
    if (set_max_dyn_smem((const void*)qr_window_kernel<T>, smw) || set_max_dyn_smem((const void*)apply_window_kernel<T>, sma))
        return TRX_ERR_LAUNCH;
    TRX_LAUNCH((qr_init_kernel<T>), dim3(batch), dim3(64), 0, s, B.st, n);
    const int max_sweeps = 30 * n + 100;
    const int nslab = cdiv_i(n, 64);
    const int adv = QW - 2 * QNS - 1;                 // guaranteed chain advance per window step
    int summary[4];
    for (long outer = 0; outer < 64L * n + 1000; ++outer) {
        if (hipMemsetAsync(B.summary, 0, sizeof(int) * 4, s) != hipSuccess) return TRX_ERR_LAUNCH;
        TRX_LAUNCH((qr_prepare_kernel<T>), dim3(batch), dim3(64), 0, s, B.A, n, B.st, B.U, B.shifts, B.summary, max_sweeps);
        if (hipMemcpyAsync(summary, B.summary, sizeof(int) * 4, hipMemcpyDeviceToHost, s) != hipSuccess) return TRX_ERR_LAUNCH;
        if (hipStreamSynchronize(s) != hipSuccess) return TRX_ERR_LAUNCH;
        if (summary[0] == 0) break;
        const int nwin = summary[1] > 0 ? cdiv_i(summary[1] + 2 * QNS, adv) + 2 : 1;
        for (int q = 0; q < nwin; ++q) {
            { ProfScope p(PROF_QR_WINDOW, s, 0, 0);
              TRX_LAUNCH((qr_window_kernel<T>), dim3(batch), dim3(256), smw, s, B.A, n, B.st, B.U, (const cx<T>*)B.shifts); }
            { ProfScope p(PROF_QR_APPLY_RIGHT, s, 0, 0);
              TRX_LAUNCH((apply_window_kernel<T>), dim3(3 * nslab, batch), dim3(256), sma, s, B.A, B.Z, n, nslab, (const QrState*)B.st, (const cx<T>*)B.U); }
        }
    }
    TRX_LAUNCH((qr_collect_info_kernel<T>), dim3(cdiv_i(batch, 64)), dim3(64), 0, s, (const QrState*)B.st, info, batch);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

template int hessenberg_qr<float>(hipStream_t, const EigBuffers<float>&, int, int, int*);
template int hessenberg_qr<double>(hipStream_t, const EigBuffers<double>&, int, int, int*);

}  // namespace trx
