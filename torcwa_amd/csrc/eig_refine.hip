// Mixed-precision route of trx_eig (fp64 problems, n >= 256: the default): the eigendecomposition is computed in fp32 by the same
// pipeline (balancing -> Hessenberg -> multi-shift QR -> Schur vectors: half the bytes, twice the matrix-core rate) and then REFINED to
// fp64 accuracy by Newton steps on the eigendecomposition (first-order perturbation theory around the current pairs):
//
//     R = A V - V Lambda                      fp64 GEMM: the ONE piece that needs fp64 (the two terms cancel to 1e-5 ... 1e-13 of |A|)
//     E = V0^-1 R                             fp32: ONE LU of the fp32 start V0, factored once, reused by every step
//     lambda_i += E_ii;    F_ij = E_ij / (lambda_j - lambda_i)  (i != j)
//     V += V F                                fp64 GEMM (in column halves: the product lands in the buffer E has just left)
//
// Why fp32 suffices for E: it is a CORRECTION.  It only has to be right to a few digits RELATIVE TO ITSELF (it is 1e-5 |A| in the first
// step, 1e-8 in the second), and V0^-1 instead of the current V^-1 is off by |F| ~ 1e-3: the iteration with the stale fp32 inverse
// converges linearly with that factor on top of the quadratic term.  (The product V F is NOT taken in fp32, although F is small as a rule:
// between two nearly degenerate eigenvalues F_ij stays O(0.1) at every step -- harmless, the two vectors span the same invariant subspace --
// and 0.1 x the fp32 rounding of V would put 6e-9 of garbage into such a pair; seen on the emulator with an all-double spectrum.)
// CPU model on the bench operators at n = 1922
// (profiles/scripts/refine_model.py, fp32 start perturbed to the GPU pipeline's 3e-5): eigen-residual 2.5e-5 -> 1.4e-8 -> 1.0e-11 ->
// 3.1e-14 |A| against 1.4e-8 -> 3.9e-12 -> 2.1e-15 for the all-fp64 Newton step of rounds 3 - 5 (G = V^-1 A V with an fp64 LU of the
// current V per step, V <- V (I + F) in fp64), which cost 3.33 n^3 complex fp64 MACs per step where this one costs 2 (+ 1 in fp32,
// + n^3 / 3 in fp32 once) and has no LU -- a latency chain of panel kernels -- inside the step.
//
// A pair whose coupling is not small against its gap (|E_ij| + |E_ji| > 0.1 |lambda_j - lambda_i|; equal eigenvalues included) is
// COUPLED: the individual vectors are ill determined there and the formula is not applied.  The connected components of the coupling
// graph (the degenerate mode pairs of symmetric meta-atoms: ~480 pairs at n = 1922 for a square; otherwise a few accidental pairs and
// triples) are diagonalised exactly from their blocks of Lambda + E by a small dense solver, and V's cluster columns are rotated by the
// result in fp64 after the first-order update.  A component of more than RCM indices, more clusters / edges than the tables hold, a
// defective block, an off-diagonal part that is not small, or a singular V0 flag the matrix, and trx_eig then redoes THOSE matrices with
// the all-fp64 pipeline as a compact sub-batch (the balanced input is kept intact for that; eig.hip).
// Steps: knob eig_refine (library default 2; torcwa_amd asks for 3 on behalf of complex128 problems and of the differentiable path).
#include "eig.hpp"
#include <cstdlib>
#include <limits>
#include <string>
#include <vector>
#include "prof.hpp"

namespace trx {
namespace {

template <class TI, class TO>
__global__ __launch_bounds__(256) void cvt_kernel(const cx<TI>* __restrict__ in, cx<TO>* __restrict__ out, long count) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) { const cx<TI> v = in[i]; out[i] = cx<TO>((TO)v.x, (TO)v.y); }
}

// E32 <- fp32(R - V diag(w)):  R = A V from the fp64 GEMM
template <class T>
__global__ __launch_bounds__(256) void refine_resid_kernel(const cx<T>* __restrict__ R, const cx<T>* __restrict__ V, const cx<T>* __restrict__ w,
                                                           cx<float>* __restrict__ E, int n) {
    const int b = blockIdx.z, i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const long e = ((long)b * n + i) * n + j;
    const cx<T> r = R[e] - V[e] * w[(long)b * n + j];
    E[e] = cx<float>((float)r.x, (float)r.y);
}

// V[:, c0 : c0 + wc] += P   (P: [B, n, wc])
template <class T>
__global__ __launch_bounds__(256) void refine_add_kernel(cx<T>* __restrict__ V, const cx<T>* __restrict__ P, int n, int c0, int wc) {
    const int b = blockIdx.z, i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= wc) return;
    V[((long)b * n + i) * n + c0 + j] += P[((long)b * n + i) * wc + j];
}

// eoff[b] = max off-diagonal |E_ij| (abs1), lmax[b] = max |lambda_i|;  lam[b, i] = d0[b, i] = lam[b, i] + E_ii.  RSPLIT workgroups per
// matrix (a row range each) leave partial maxima, refine_scan_reduce_kernel combines them.
constexpr int RSPLIT = 32;
template <class T>
__global__ __launch_bounds__(256) void refine_scan_kernel(const cx<float>* __restrict__ Eall, int n, cx<T>* __restrict__ lam, cx<T>* __restrict__ d0, T* __restrict__ part) {
    __shared__ T red[2][4];
    const int b = blockIdx.y, sp = blockIdx.x;
    const cx<float>* E = Eall + (long)b * n * n;
    const int r0 = (int)((long)n * sp / RSPLIT), r1 = (int)((long)n * (sp + 1) / RSPLIT);
    T eo = T(0), lm = T(0);
    for (long e = (long)r0 * n + threadIdx.x; e < (long)r1 * n; e += blockDim.x) {
        const int i = (int)(e / n), j = (int)(e - (long)i * n);
        const cx<float> v = E[e];
        if (i == j) {
            const cx<T> l = lam[(long)b * n + i] + cx<T>((T)v.x, (T)v.y);
            lam[(long)b * n + i] = l; d0[(long)b * n + i] = l;
            const T a = abs1(l);
            lm = (a > lm || !(a == a)) ? a : lm;
        } else {
            const T a = (T)abs1(v);
            eo = (a > eo || !(a == a)) ? a : eo;                      // a NaN sticks
        }
    }
    // NaN-propagating maximum: wave_max keeps whichever operand wins `w > v`, so a NaN held by only some lanes could drop out
    const bool has_nan = __any(!(eo == eo) || !(lm == lm));
    eo = wave_max(has_nan ? T(0) : eo);
    if (has_nan) eo = (T)__builtin_nan("");
    lm = wave_max(has_nan ? T(0) : lm);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = eo; red[1][threadIdx.x >> 6] = lm; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) { eo = (red[0][w] > eo || !(red[0][w] == red[0][w])) ? red[0][w] : eo; lm = red[1][w] > lm ? red[1][w] : lm; }
        part[((long)b * RSPLIT + sp) * 2] = eo; part[((long)b * RSPLIT + sp) * 2 + 1] = lm;
    }
}
template <class T>
__global__ void refine_scan_reduce_kernel(const T* __restrict__ part, T* __restrict__ eoff, T* __restrict__ lmax, int* __restrict__ flags, int batch) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    T eo = T(0), lm = T(0);
    for (int sp = 0; sp < RSPLIT; ++sp) {
        const T e = part[((long)b * RSPLIT + sp) * 2], l = part[((long)b * RSPLIT + sp) * 2 + 1];
        eo = (e > eo || !(e == e)) ? e : eo;
        if (!(eo == eo)) break;                                       // NaN: final
        lm = l > lm ? l : lm;
    }
    eoff[b] = eo; lmax[b] = lm;
    if (!(eo <= T(1e-2) * lm)) atomicOr(&flags[b], 1);                // the current pairs are not close enough (or not finite)
}

// ---- the coupling graph ----------------------------------------------------------------------------------------------------------------
// Which pairs (i, j) can NOT take the first-order formula: those whose coupling is not small against their gap,
//     |E_ij| + |E_ji| > rho |lambda_j - lambda_i|      (rho = 0.1; equal eigenvalues included).
// A global distance threshold would lump together every pair of close eigenvalues -- at n = 1922 the fp32 start leaves max |E| = 4e-4 and
// the spectrum has gaps of 2e-3 -- although close eigenvalues are, as a rule, hardly coupled at all: with this criterion the bench operator
// has NO coupled pair, a square meta-atom exactly its degenerate pairs.  A coupled pair (i < j) marks both indices and appends an edge to the
// matrix's list (capacity REC; the count keeps running so that an overflow is seen).
constexpr int RCM = 32;       // largest cluster diagonalised here
constexpr int RCK = 1024;     // most clusters per matrix
constexpr int REC = REFINE_EDGE_CAP;     // most coupled pairs (edges) per matrix
constexpr int RCX = 32768;    // complex entries of the clusters' eigenvector blocks per matrix (sum of size^2)
template <class T>
struct RefineClusters {       // per matrix (at most REFINE_CLUSTER_BYTES)
    int ncl, ncoupled, largest, pad;
    int size[RCK];
    int xoff[RCK];            // start of the cluster's size x size eigenvector block in X (row-major, leading dimension = size)
    int member[RCK][RCM];
    cx<T> X[RCX];             // columns = eigenvectors of the cluster's block of Lambda + E, in the basis of its members
    cx<T> W[2 * RCX];         // work space of the small solver: the cluster's block and its Schur vectors, at 2 * xoff
};
static_assert(sizeof(RefineClusters<double>) <= REFINE_CLUSTER_BYTES, "cluster table");

// Tiled: a workgroup takes the 32 x 32 tile pair (I, J), J >= I, of E -- E[I, J] and E[J, I], both read row-wise (coalesced) into LDS -- and
// tests its 1024 index pairs (the one-thread-per-index walk of the first version read E[i, :] with a stride of n between lanes: 9.9 ms per
// call at the bench shape against ~1 ms for one pass over E).
constexpr int RCT = 32;
template <class T>
__global__ __launch_bounds__(256) void refine_cluster_kernel(const cx<float>* __restrict__ Eall, const cx<T>* __restrict__ lam, int n, int* __restrict__ coupled,
                                                             int* __restrict__ edges, int* __restrict__ ecount) {
    __shared__ float a_ij[RCT][RCT + 1], a_ji[RCT][RCT + 1];          // |E[I0 + r, J0 + c]|_1 and |E[J0 + r, I0 + c]|_1
    __shared__ cx<T> li[RCT], lj[RCT];
    const int b = blockIdx.z, I0 = blockIdx.y * RCT, J0 = blockIdx.x * RCT;
    if (J0 < I0) return;                                               // (workgroup-uniform) upper triangle of tile pairs only
    const cx<float>* E = Eall + (long)b * n * n;
    const int t = threadIdx.x, c = t & (RCT - 1), r0 = t / RCT;        // 8 rows per pass
    for (int r = r0; r < RCT; r += 256 / RCT) {
        const int i = I0 + r, j = J0 + c;
        a_ij[r][c] = (i < n && j < n) ? abs1(E[(long)i * n + j]) : 0.f;
        const int i2 = J0 + r, j2 = I0 + c;
        a_ji[r][c] = (i2 < n && j2 < n) ? abs1(E[(long)i2 * n + j2]) : 0.f;
    }
    if (t < RCT) li[t] = I0 + t < n ? lam[(long)b * n + I0 + t] : cx<T>(T(0), T(0));
    else if (t < 2 * RCT) lj[t - RCT] = J0 + t - RCT < n ? lam[(long)b * n + J0 + t - RCT] : cx<T>(T(0), T(0));
    __syncthreads();
    const T rho = T(0.1);
    for (int r = r0; r < RCT; r += 256 / RCT) {
        const int i = I0 + r, j = J0 + c;
        if (i >= n || j >= n || j <= i) continue;
        const T cpl = (T)(a_ij[r][c] + a_ji[c][r]);                   // |E_ij| + |E_ji|
        if (!(cpl <= rho * abs1(lj[c] - li[r]))) {
            coupled[(long)b * n + i] = 1;
            coupled[(long)b * n + j] = 1;
            const int e = atomicAdd(&ecount[b], 1);
            if (e < REC) { edges[((long)b * REC + e) * 2] = i; edges[((long)b * REC + e) * 2 + 1] = j; }
        }
    }
}

// Schur form + eigenvectors of a dense m x m block (m <= RCM), serial: Hessenberg by Givens rotations, explicitly shifted QR
// iterations with deflation, triangular back-substitution, X = Z Y with unit columns.  Bm, Z, X: m x m, leading dimension m, in
// global memory (a 32 x 32 block per lane does not belong in private memory).  Returns false if it did
// not converge or the eigenvector matrix is numerically singular (a defective block): the caller then leaves the cluster alone and flags
// the matrix.
template <class T>
__device__ bool small_dense_eig(cx<T>* Bm, int m, cx<T>* Z, cx<T>* mu, cx<T>* X) {
    auto at = [&](cx<T>* M_, int r, int c) -> cx<T>& { return M_[r * m + c]; };
    for (int r = 0; r < m; ++r)
        for (int c = 0; c < m; ++c) at(Z, r, c) = cx<T>(r == c ? T(1) : T(0), T(0));
    // rotation G = [[c, s], [-conj(s), c]] with G [f; g] = [r; 0]
    auto rotg = [&](cx<T> f, cx<T> g, T& c, cx<T>& sn) {
        const T ag = cabs(g);
        if (ag == T(0)) { c = T(1); sn = cx<T>(T(0), T(0)); return; }
        const T af = cabs(f);
        if (af == T(0)) { c = T(0); sn = (T(1) / ag) * conj(g); return; }
        const T d = sqrt(af * af + ag * ag);
        c = af / d;
        sn = (T(1) / (af * d)) * (f * conj(g));
    };
    auto rot_rows = [&](int r1, int r2, T c, cx<T> sn, int c0) {       // rows r1, r2 of Bm <- G rows
        for (int q = c0; q < m; ++q) {
            const cx<T> x = at(Bm, r1, q), y = at(Bm, r2, q);
            at(Bm, r1, q) = c * x + sn * y;
            at(Bm, r2, q) = c * y - conj(sn) * x;
        }
    };
    auto rot_cols = [&](cx<T>* M_, int c1, int c2, T c, cx<T> sn, int rmax) {   // columns <- columns G^H
        for (int r = 0; r < rmax; ++r) {
            const cx<T> x = at(M_, r, c1), y = at(M_, r, c2);
            at(M_, r, c1) = c * x + conj(sn) * y;
            at(M_, r, c2) = c * y - sn * x;
        }
    };
    // Hessenberg form: zero column c below the subdiagonal with rotations in the planes (r-1, r)
    for (int c = 0; c + 2 < m; ++c)
        for (int r = m - 1; r >= c + 2; --r) {
            T cs; cx<T> sn;
            rotg(at(Bm, r - 1, c), at(Bm, r, c), cs, sn);
            rot_rows(r - 1, r, cs, sn, c);
            at(Bm, r, c) = cx<T>(T(0), T(0));
            rot_cols(Bm, r - 1, r, cs, sn, m);
            rot_cols(Z, r - 1, r, cs, sn, m);
        }
    // shifted QR iterations on the active block [l, hi]
    const T ulp = eps_of<T>::value;
    int hi = m - 1, its = 0, total = 0;
    while (hi > 0) {
        int l = hi;
        while (l > 0) {
            T sc = abs1(at(Bm, l - 1, l - 1)) + abs1(at(Bm, l, l));
            if (sc == T(0)) sc = T(1);
            if (abs1(at(Bm, l, l - 1)) <= ulp * sc) { at(Bm, l, l - 1) = cx<T>(T(0), T(0)); break; }
            --l;
        }
        if (l == hi) { --hi; its = 0; continue; }
        if (++total > 60 * m) return false;
        ++its;
        cx<T> sig;
        {
            const cx<T> a = at(Bm, hi - 1, hi - 1), bq = at(Bm, hi - 1, hi), cq = at(Bm, hi, hi - 1), d = at(Bm, hi, hi);
            if (its % 10 == 0) sig = d + cx<T>(T(0.75) * abs1(cq), T(0));
            else {
                const cx<T> tr = T(0.5) * (a + d);
                const cx<T> sq = csqrt((a - tr) * (d - tr) * T(-1) + bq * cq);
                const cx<T> e1 = tr + sq, e2 = tr - sq;
                sig = (abs1(e1 - d) < abs1(e2 - d)) ? e1 : e2;
            }
        }
        // implicit single-shift sweep (bulge chase with rotations)
        cx<T> f = at(Bm, l, l) - sig, g = at(Bm, l + 1, l);
        for (int r = l; r < hi; ++r) {
            T cs; cx<T> sn;
            if (r > l) { f = at(Bm, r, r - 1); g = at(Bm, r + 1, r - 1); }
            rotg(f, g, cs, sn);
            rot_rows(r, r + 1, cs, sn, r > l ? r - 1 : l);
            if (r > l) at(Bm, r + 1, r - 1) = cx<T>(T(0), T(0));
            rot_cols(Bm, r, r + 1, cs, sn, (r + 2 < m ? r + 2 : m - 1) + 1);
            rot_cols(Z, r, r + 1, cs, sn, m);
        }
    }
    // eigenvectors of the triangular factor, X = Z Y
    T tn = T(0);
    for (int r = 0; r < m; ++r)
        for (int c = r; c < m; ++c) { const T a = abs1(at(Bm, r, c)); tn = a > tn ? a : tn; }
    const T smin = (tn > T(0) ? tn : T(1)) * ulp;
    for (int k = 0; k < m; ++k) {
        cx<T> y[RCM];
        const cx<T> lam = at(Bm, k, k);
        mu[k] = lam;
        for (int r = 0; r < m; ++r) y[r] = cx<T>(T(0), T(0));
        y[k] = cx<T>(T(1), T(0));
        for (int r = k - 1; r >= 0; --r) {
            cx<T> sacc(T(0), T(0));
            for (int q = r + 1; q <= k; ++q) cfma(sacc, at(Bm, r, q), y[q]);
            cx<T> d = at(Bm, r, r) - lam;
            if (abs1(d) < smin) d = cx<T>(smin, T(0));
            y[r] = cdiv(-sacc, d);
        }
        T nrm = T(0);
        for (int r = 0; r < m; ++r) {
            cx<T> v(T(0), T(0));
            for (int q = 0; q <= k; ++q) cfma(v, at(Z, r, q), y[q]);
            X[r * m + k] = v;
            nrm += norm2(v);
        }
        const T inv = nrm > T(0) ? T(1) / sqrt(nrm) : T(1);
        for (int r = 0; r < m; ++r) X[r * m + k] = inv * X[r * m + k];
    }
    // conditioning guard: smallest pivot of unit-column X under elimination with partial pivoting, on a copy (Z is free now).  (A bound
    // on |det X| -- rounds 3 - 5, clusters of at most 8 -- shrinks geometrically with m for perfectly conditioned random bases.)
    for (int r = 0; r < m; ++r)
        for (int c = 0; c < m; ++c) at(Z, r, c) = X[r * m + c];
    T pmin = T(1);
    for (int c = 0; c < m; ++c) {
        int pr = c; T best = abs1(at(Z, c, c));
        for (int r = c + 1; r < m; ++r) { const T a = abs1(at(Z, r, c)); if (a > best) { best = a; pr = r; } }
        if (!(best > T(0))) return false;
        if (pr != c) for (int q = 0; q < m; ++q) { const cx<T> tmp = at(Z, c, q); at(Z, c, q) = at(Z, pr, q); at(Z, pr, q) = tmp; }
        pmin = best < pmin ? best : pmin;
        for (int r = c + 1; r < m; ++r) {
            const cx<T> fct = cdiv(at(Z, r, c), at(Z, c, c));
            for (int q = c; q < m; ++q) cfma(at(Z, r, q), -fct, at(Z, c, q));
        }
    }
    return pmin >= T(1e-5);
}

// One 64-thread workgroup per matrix: connected components of the coupling graph (label propagation over the edge list, labels of all n
// indices in LDS), cluster numbering and member lists, exact diagonalisation of every component's block of Lambda + E (one thread per
// component), cluster tables for the build / rotation kernels.  clus[b, i] = RCM * cluster + position, or -1.
// Dynamic LDS: 2 n ints (labels, cluster ids).
template <class T>
__global__ __launch_bounds__(64) void refine_solve_clusters_kernel(const cx<float>* __restrict__ Eall, int n, cx<T>* __restrict__ lam, const cx<T>* __restrict__ d0all,
                                                                    const int* __restrict__ coupled, const int* __restrict__ edges, const int* __restrict__ ecount,
                                                                    int* __restrict__ clus, RefineClusters<T>* __restrict__ tab, int* __restrict__ flags, int* __restrict__ nrot) {
    TRX_DYN_SMEM(smem);
    int* lab = reinterpret_cast<int*>(smem);      // [n]
    int* cid = lab + n;                            // [n]  cluster number of a representative
    __shared__ int changed_s, bad_s;
    const int b = blockIdx.x, t = threadIdx.x;
    const cx<float>* E = Eall + (long)b * n * n;
    const cx<T>* d0 = d0all + (long)b * n;
    const int* cp = coupled + (long)b * n;
    const int* ed = edges + (long)b * REC * 2;
    int* cl = clus + (long)b * n;
    RefineClusters<T>& Tb = tab[b];
    for (int i = t; i < n; i += 64) { lab[i] = i; cl[i] = -1; }
    if (t == 0) { bad_s = 0; Tb.ncl = 0; Tb.ncoupled = 0; Tb.largest = 0; }
    __syncthreads();
    const int ne = ecount[b];
    if (ne == 0) return;
    if (ne > REC) { if (t == 0) { atomicOr(&flags[b], 2 | 32); Tb.ncoupled = -ne; } return; }
    // components: label propagation over the edges (a label only decreases; the representative of a component is its smallest index)
    for (int sweep = 0; sweep < n; ++sweep) {
        if (t == 0) changed_s = 0;
        __syncthreads();
        for (int e = t; e < ne; e += 64) {
            const int p = ed[2 * e], q = ed[2 * e + 1];
            const int lp = lab[p], lq = lab[q];
            if (lp < lq) { atomicMin(&lab[q], lp); changed_s = 1; }
            else if (lq < lp) { atomicMin(&lab[p], lq); changed_s = 1; }
        }
        __syncthreads();
        const int ch = changed_s;
        __syncthreads();
        if (!ch) break;
    }
    // cluster numbering and member lists (serial over the indices: ascending, so a representative is numbered before its members).  A
    // component beyond RCM members is still counted to its end: the statistics (TRX_EIG_DEBUG) want the true size.
    if (t == 0) {
        int ncl = 0, nc = 0, big = 0, xo = 0, bad = 0;
        for (int i = 0; i < n; ++i) {
            if (!cp[i]) continue;
            ++nc;
            const int r = lab[i];
            if (r == i) {
                if (ncl == RCK) { bad = 128; break; }
                cid[i] = ncl; Tb.size[ncl] = 0; ++ncl;
            }
            const int c = cid[r];
            if (Tb.size[c] >= RCM) bad |= 64;
            else Tb.member[c][Tb.size[c]] = i;
            ++Tb.size[c];
            big = Tb.size[c] > big ? Tb.size[c] : big;
        }
        Tb.ncl = ncl; Tb.ncoupled = nc; Tb.largest = big;
        for (int c = 0; c < ncl && !bad; ++c) {
            if (Tb.size[c] < 2) { bad = 128; break; }
            Tb.xoff[c] = xo;
            xo += Tb.size[c] * Tb.size[c];
            if (xo > RCX) { bad = 128; break; }
        }
        bad_s = bad;
    }
    __syncthreads();
    if (bad_s) { if (t == 0) { atomicOr(&flags[b], 2 | bad_s); Tb.ncl = 0; } return; }
    // every cluster: exact eigendecomposition of its block of Lambda + E
    const int ncl = Tb.ncl;
    if (t == 0 && ncl > 0) atomicAdd(nrot, ncl);          // "V's cluster columns get rotated in this step" (the host then refreshes the LU)
    for (int c = t; c < ncl; c += 64) {
        cx<T> mu[RCM];
        const int m = Tb.size[c];
        cx<T>* Bm = Tb.W + 2 * Tb.xoff[c];
        cx<T>* Z = Bm + m * m;
        for (int r = 0; r < m; ++r)
            for (int q = 0; q < m; ++q) {
                const int i = Tb.member[c][r], j = Tb.member[c][q];
                const cx<float> e = E[(long)i * n + j];
                Bm[r * m + q] = (r == q) ? d0[i] : cx<T>((T)e.x, (T)e.y);
            }
        if (!small_dense_eig<T>(Bm, m, Z, mu, Tb.X + Tb.xoff[c])) { atomicOr(&flags[b], 2 | 256); continue; }
        for (int r = 0; r < m; ++r) { const int i = Tb.member[c][r]; cl[i] = RCM * c + r; lam[(long)b * n + i] = mu[r]; }
    }
}

// F (fp64, out of place) from E (fp32):  F_kj = E_kj / (d_j - d_k) for k != j outside the clusters (d = Lambda + diag E, saved in d0 by the
// scan), 0 on the diagonal and between two members of one cluster (no first-order correction inside a cluster: its rotation follows).
template <class T>
__global__ __launch_bounds__(256) void refine_build_kernel(const cx<float>* __restrict__ Eall, cx<T>* __restrict__ Fall, int n, const cx<T>* __restrict__ d0all,
                                                           const int* __restrict__ clus) {
    const int b = blockIdx.z, k = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const long e = ((long)b * n + k) * n + j;
    const cx<T>* d0 = d0all + (long)b * n;
    const int ck = clus[(long)b * n + k], cj = clus[(long)b * n + j];
    if (k == j || (ck >= 0 && cj >= 0 && ck / RCM == cj / RCM)) { Fall[e] = cx<T>(T(0), T(0)); return; }
    const cx<float> v = Eall[e];
    Fall[e] = cdiv(cx<T>((T)v.x, (T)v.y), d0[j] - d0[k]);
}

// V[:, members] <- V[:, members] X_c for every cluster c (fp64; one thread per row of V)
template <class T>
__global__ __launch_bounds__(256) void refine_rotate_clusters_kernel(cx<T>* __restrict__ Vall, int n, const int* __restrict__ clus, const RefineClusters<T>* __restrict__ tab) {
    const int b = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
    const RefineClusters<T>& Tb = tab[b];
    const int ncl = Tb.ncl;
    if (k >= n || ncl == 0) return;
    cx<T>* Vrow = Vall + ((long)b * n + k) * n;
    for (int c = 0; c < ncl; ++c) {
        const int m = Tb.size[c];
        if (clus[(long)b * n + Tb.member[c][0]] < 0) continue;          // the small solver gave up on this cluster (matrix flagged)
        const cx<T>* X = Tb.X + Tb.xoff[c];
        cx<T> in[RCM];
        for (int r = 0; r < m; ++r) in[r] = Vrow[Tb.member[c][r]];
        for (int pos = 0; pos < m; ++pos) {
            cx<T> v(T(0), T(0));
            for (int r = 0; r < m; ++r) cfma(v, in[r], X[r * m + pos]);
            Vrow[Tb.member[c][pos]] = v;
        }
    }
}

// flags[b] |= 1 where the fp32 eigensolver reported unconverged eigenvalues / the LU of the fp32 start hit a zero pivot
__global__ void refine_fold_info_kernel(int* __restrict__ flags, const int* __restrict__ info32, int batch) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch && info32[b] != 0) atomicOr(&flags[b], 1);
}

// any[0] = number of matrices with a flag
__global__ void refine_count_flags_kernel(const int* __restrict__ flags, int* __restrict__ any, int batch) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch && flags[b] != 0) atomicAdd(any, 1);
}

}  // namespace

static int g_refine_steps = 2;
int refine_set_knob(const char* key, int value) {
    if (std::string(key) != "eig_refine" || value < 0 || value > 4) return TRX_ERR_ARG;
    g_refine_steps = value == 0 ? 2 : value;
    return TRX_OK;
}
int refine_steps() { return g_refine_steps; }

// A: balanced fp64 input (kept intact); V32 / w32: its fp32 eigendecomposition.  On return w, V hold the refined fp64 eigenpairs of A
// (unit 2-norm is restored by the caller together with the undo of the balancing); *host_any = number of flagged matrices (their results are
// not to be used: host_bad[b] != 0, one entry per matrix).  When more than a third of the batch is flagged after the first scan the
// remaining steps are skipped (the caller redoes the whole batch).
template <class T>
int eig_refine(hipStream_t s, const RefineBuffers<T>& R, const cx<T>* A, const cx<float>* V32, const cx<float>* w32, cx<T>* w, cx<T>* V, int n, int batch, int steps,
               int* host_any, int* host_bad) {
    static const bool debug = getenv("TRX_EIG_DEBUG") != nullptr;          // read once per process (never per call)
    const cx<T> one(T(1), T(0)), zero(T(0), T(0));
    const long nn = (long)n * n;
    const long cntV = nn * batch, cntw = (long)n * batch;
    const size_t sm_cl = sizeof(int) * 2 * (size_t)n;
    if (set_max_dyn_smem((const void*)refine_solve_clusters_kernel<T>, sm_cl)) return TRX_ERR_LAUNCH;
    cx<T>* Pw = reinterpret_cast<cx<T>*>(R.E32);       // [B, n, wc]: product V F of one column block (E's buffer, free once F has been built)
    const int wc = n / 2 > 0 ? n / 2 : 1;               // column block: n * wc fp64 elements fit the n * n fp32 elements of E
    // the fp32 start: V <- fp64(V32), w <- fp64(w32); its LU (fp32, on a copy: V32 itself lies where R is about to be written)
    TRX_LAUNCH((cvt_kernel<float, T>), dim3(cdiv_i(cntV, 256)), dim3(256), 0, s, V32, V, cntV);
    TRX_LAUNCH((cvt_kernel<float, T>), dim3(cdiv_i(cntw, 256)), dim3(256), 0, s, w32, w, cntw);
    if (hipMemsetAsync(R.flags, 0, sizeof(int) * (batch + 2), s) != hipSuccess) return TRX_ERR_LAUNCH;
    TRX_LAUNCH(refine_fold_info_kernel, dim3(cdiv_i(batch, 64)), dim3(64), 0, s, R.flags, (const int*)R.linfo, batch);     // R.linfo: info of the fp32 solve on entry
    if (hipMemcpyAsync(R.LU32, V32, sizeof(cx<float>) * cntV, hipMemcpyDeviceToDevice, s) != hipSuccess) return TRX_ERR_LAUNCH;
    int rc = lu_factor<float>(s, R.LU32, n, nn, n, R.piv, batch, R.linfo); if (rc) return rc;
    TRX_LAUNCH(refine_fold_info_kernel, dim3(cdiv_i(batch, 64)), dim3(64), 0, s, R.flags, (const int*)R.linfo, batch);     // a singular V0
    for (int it = 0; it < steps; ++it) {
        // E = V0^-1 (A V - V Lambda)
        rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, n, n, one, A, n, nn, V, n, nn, zero, R.Rb, n, nn, batch); if (rc) return rc;
        TRX_LAUNCH((refine_resid_kernel<T>), dim3(cdiv_i(n, 256), n, batch), dim3(256), 0, s, (const cx<T>*)R.Rb, (const cx<T>*)V, (const cx<T>*)w, R.E32, n);
        rc = lu_solve<float>(s, R.LU32, n, nn, n, R.piv, R.E32, n, nn, n, batch); if (rc) return rc;
        // eigenvalue update, size of the correction, coupling graph, clusters
        TRX_LAUNCH((refine_scan_kernel<T>), dim3(RSPLIT, batch), dim3(256), 0, s, (const cx<float>*)R.E32, n, w, R.d0, R.scan_part);
        TRX_LAUNCH((refine_scan_reduce_kernel<T>), dim3(cdiv_i(batch, 64)), dim3(64), 0, s, (const T*)R.scan_part, R.eoff, R.lmax, R.flags, batch);
        if (hipMemsetAsync(R.partner, 0, sizeof(int) * cntw, s) != hipSuccess || hipMemsetAsync(R.ecount, 0, sizeof(int) * batch, s) != hipSuccess) return TRX_ERR_LAUNCH;
        TRX_LAUNCH((refine_cluster_kernel<T>), dim3(cdiv_i(n, RCT), cdiv_i(n, RCT), batch), dim3(256), 0, s, (const cx<float>*)R.E32, (const cx<T>*)w, n, R.partner, R.edges, R.ecount);
        TRX_LAUNCH((refine_solve_clusters_kernel<T>), dim3(batch), dim3(64), sm_cl, s, (const cx<float>*)R.E32, n, w, (const cx<T>*)R.d0, (const int*)R.partner,
                   (const int*)R.edges, (const int*)R.ecount, R.clus, (RefineClusters<T>*)R.tab, R.flags, R.flags + batch + 1);
        int host_cnt[2] = {0, 0};                         // {flagged matrices, clusters rotated in this step}
        if (it + 1 < steps) {
            // A matrix the scheme cannot certify (far-off start, a cluster beyond the exact treatment, singular V0) is known after the FIRST
            // scan.  Flagged matrices are redone by the all-fp64 pipeline afterwards (as a sub-batch); when more than a third of the batch is
            // flagged the caller redoes the whole batch instead, so the remaining Newton steps would be thrown away: stop here.
            TRX_LAUNCH(refine_count_flags_kernel, dim3(cdiv_i(batch, 64)), dim3(64), 0, s, (const int*)R.flags, R.flags + batch, batch);
            if (hipMemcpyAsync(host_cnt, R.flags + batch, 2 * sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess) return TRX_ERR_LAUNCH;
            if (hipStreamSynchronize(s) != hipSuccess) return TRX_ERR_LAUNCH;
            if (hipMemsetAsync(R.flags + batch, 0, 2 * sizeof(int), s) != hipSuccess) return TRX_ERR_LAUNCH;
            *host_any = host_cnt[0];
            if (it == 0 && 3 * host_cnt[0] > batch && !debug) {
                for (int b = 0; b < batch; ++b) host_bad[b] = 1;
                return TRX_OK;
            }
        }
        // V += V F (F into the residual's buffer, the product block by block into E's), then the cluster columns' rotations
        TRX_LAUNCH((refine_build_kernel<T>), dim3(cdiv_i(n, 256), n, batch), dim3(256), 0, s, (const cx<float>*)R.E32, R.Rb, n, (const cx<T>*)R.d0, (const int*)R.clus);
        for (int c0 = 0; c0 < n; c0 += wc) {
            const int w = n - c0 < wc ? n - c0 : wc;
            rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, w, n, one, V, n, nn, R.Rb + c0, n, nn, zero, Pw, w, (long)n * w, batch); if (rc) return rc;
            TRX_LAUNCH((refine_add_kernel<T>), dim3(cdiv_i(w, 256), n, batch), dim3(256), 0, s, V, (const cx<T>*)Pw, n, c0, w);
        }
        TRX_LAUNCH((refine_rotate_clusters_kernel<T>), dim3(cdiv_i(n, 256), batch), dim3(256), 0, s, V, n, (const int*)R.clus, (const RefineClusters<T>*)R.tab);
        if (host_cnt[1] > 0) {
            // Cluster columns were ROTATED (an O(1) change of basis inside their invariant subspaces): V0^-1 is no approximate inverse of
            // the new V on those rows any more, and the clusters would converge linearly with an O(1) factor (seen: an all-double spectrum
            // halved its error per step).  Refresh the fp32 LU from the current V -- only then, and for the whole batch: matrices without
            // clusters (the bench operators) never pay for a second factorisation.
            TRX_LAUNCH((cvt_kernel<T, float>), dim3(cdiv_i(cntV, 256)), dim3(256), 0, s, (const cx<T>*)V, R.LU32, cntV);
            rc = lu_factor<float>(s, R.LU32, n, nn, n, R.piv, batch, R.linfo); if (rc) return rc;
            TRX_LAUNCH(refine_fold_info_kernel, dim3(cdiv_i(batch, 64)), dim3(64), 0, s, R.flags, (const int*)R.linfo, batch);
        }
    }
    {
        // per-matrix verdict: flags accumulated over the steps
        std::vector<int> hf(batch);
        if (hipMemcpyAsync(hf.data(), R.flags, sizeof(int) * batch, hipMemcpyDeviceToHost, s) != hipSuccess) return TRX_ERR_LAUNCH;
        if (hipStreamSynchronize(s) != hipSuccess) return TRX_ERR_LAUNCH;
        int cnt = 0;
        for (int b = 0; b < batch; ++b) { host_bad[b] = hf[b] != 0; cnt += host_bad[b]; }
        *host_any = cnt;
        if (debug) {
            // statistics of the LAST step's coupling graphs: per matrix {clusters, coupled indices (negative: edge-list overflow), largest
            // component}, the first four ints of its table
            std::vector<T> he(batch), hm(batch);
            (void)hipMemcpy(he.data(), R.eoff, sizeof(T) * batch, hipMemcpyDeviceToHost);
            (void)hipMemcpy(hm.data(), R.lmax, sizeof(T) * batch, hipMemcpyDeviceToHost);
            int f1 = 0, f32 = 0, f64 = 0, f128 = 0, f256 = 0, worst = 0, most = 0;
            std::string line, hist;
            int sizes[8] = {0, 0, 0, 0, 0, 0, 0, 0};       // largest component: 0, 2, 3-4, 5-8, 9-16, 17-32, 33-64, > 64
            for (int b = 0; b < batch; ++b) {
                int st[4] = {0, 0, 0, 0};
                (void)hipMemcpy(st, (const char*)R.tab + (size_t)b * sizeof(RefineClusters<T>), sizeof(st), hipMemcpyDeviceToHost);     // (tab[b]: the struct's own stride)
                f1 += (hf[b] & 1) != 0; f32 += (hf[b] & 32) != 0; f64 += (hf[b] & 64) != 0; f128 += (hf[b] & 128) != 0; f256 += (hf[b] & 256) != 0;
                worst = st[2] > worst ? st[2] : worst;
                most = st[1] > most ? st[1] : most;
                const int g = st[2];
                sizes[g == 0 ? 0 : g <= 2 ? 1 : g <= 4 ? 2 : g <= 8 ? 3 : g <= 16 ? 4 : g <= 32 ? 5 : g <= 64 ? 6 : 7] += 1;
                if (hf[b] != 0 && line.size() < 600) line += " " + std::to_string(b) + ":f" + std::to_string(hf[b]) + "/" + std::to_string(st[2]) + "/" + std::to_string(st[1]);
            }
            for (int i = 0; i < 8; ++i) hist += " " + std::to_string(sizes[i]);
            fprintf(stderr, "libtrx eig_refine: n %d batch %d steps %d: flagged %d (far-off / singular %d, edge overflow %d, cluster > %d: %d, table overflow %d, small solver %d) | "
                            "largest component %d, most coupled indices %d | matrices by largest component [0, 2, 3-4, 5-8, 9-16, 17-32, 33-64, > 64]:%s | "
                            "last step: max|E| %.3e max|lambda| %.3e (matrix 0), %.3e %.3e (matrix %d) | flagged (matrix:flags/largest/coupled):%s\n",
                    n, batch, steps, cnt, f1, f32, RCM, f64, f128, f256, worst, most, hist.c_str(), (double)he[0], (double)hm[0], (double)he[batch - 1], (double)hm[batch - 1], batch - 1,
                    line.c_str());
        }
    }
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

template int eig_refine<double>(hipStream_t, const RefineBuffers<double>&, const cx<double>*, const cx<float>*, const cx<float>*, cx<double>*, cx<double>*, int, int, int, int*, int*);

}  // namespace trx
