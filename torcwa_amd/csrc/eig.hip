// trx_eig: batched general complex eigendecomposition A V = V diag(w) -- the MI355X replacement for the
// torch.linalg.eig call behind torcwa's `Eig.apply` (torcwa/torch_eig.py:12-17, used at rcwa.py:1236/1238).
// Pipeline (the stages of LAPACK zgeev: gebal -> gehrd/unghr -> hseqr -> trevc -> gebak -> normalise): balancing (eig_bal.hip) ->
// Hessenberg reduction (eig_hess.hip) -> multi-shift QR to Schur form (eig_qr.hip) -> triangular eigenvectors + back-transform
// + undo of the balancing + unit-norm scaling (eig_vec.hip).
#include "eig.hpp"
#include "prof.hpp"
#include <cstdlib>
#include <string>
#include <vector>

namespace trx {

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

static int eig_vec_env() {
    const char* e = getenv("TRX_EIG_VEC");
    const int v = e ? atoi(e) : 0;
    return (v == 0 || v == 1 || v == 3) ? v : 0;
}
static int g_eig_vec = eig_vec_env();      // trx_tuning("eig_vec", v): 0 automatic, 1 all-fp64 (all-fp32 for complex64 input) Schur vectors, 3 mixed wherever n >= 8
// Per-call options (trx_eig_opts / trx_eig_ws_bytes_opts): route and Newton steps of THIS call, held thread-locally for the duration of the
// call on the calling host thread -- the process-global knobs are only the defaults, so two threads (or a complex64 and a complex128 solver
// in one process) can no longer overwrite each other's setting between trx_tuning and trx_eig.  -1 = not set (use the knob).
static thread_local int tl_eig_vec = -1, tl_refine = -1;
static thread_local int tl_last_fallback = 0;      // the calling thread's last trx_eig: number of matrices the mixed route redid in fp64 (trx_eig_last_fallback)
static inline int cur_eig_vec() { return tl_eig_vec >= 0 ? tl_eig_vec : g_eig_vec; }
struct EigCallOpts {
    EigCallOpts(unsigned opts) { const int r = opts & 0xF, v = (opts >> 4) & 0xF; tl_refine = r ? r : -1; tl_eig_vec = v ? v : -1; }
    ~EigCallOpts() { tl_refine = -1; tl_eig_vec = -1; }
};
// mixed-precision route (fp32 eigendecomposition + Newton refinement in fp64, eig_refine.hip): fp64 problems of at least 256 rows
// Automatic: batches of at least 8 (measured, n = 1922: +6 % of the whole layer-solve step at batch 16, 64 and 128; a single n = 5202
// matrix, whose fp64 solve is a latency chain that fp32 does not shorten, loses 60 % to the extra refinement work).
bool eig_uses_mixed(int n, int batch, size_t elem) { return elem == 8 && ((cur_eig_vec() == 3 && n >= 8) || (cur_eig_vec() == 0 && n >= 256 && batch >= 8)); }
int eig_set_knob(const char* key, int value) {
    if (std::string(key) != "eig_vec" || (value != 0 && value != 1 && value != 3)) return TRX_ERR_ARG;
    g_eig_vec = value;
    return TRX_OK;
}


template <class T> size_t eig_ws_bytes_t(int n, int batch);
static size_t eig_ws_bytes_f32(int n, int batch) { return eig_ws_bytes_t<float>(n, batch); }

// Mixed route, on top of the all-fp64 layout: Z holds the fp32 LU of the fp32 start and the fp32 correction matrix E, X the fp64 residual
// A V (then F); extra: pivots / flags / coupling graph / cluster tables / the saved diagonal +
// whatever the fp32 pool (A32, the fp32 eigensolver's own workspace, V32, w32) needs beyond Z | X, which it overlaps: the pool is dead before
// the refinement first writes them, except V32 / w32, which sit at its END (behind Z in any case; they are consumed before X is written).
static size_t mixed_pool_bytes(int n, int batch) {
    const size_t B = batch, N = n;
    return al256(8 * B * N * N) + al256(eig_ws_bytes_f32(n, batch)) + al256(8 * B * N * N) + al256(8 * B * N);
}
static size_t mixed_extra_bytes(int n, int batch) {
    const size_t B = batch, N = n, e = 16;
    const size_t zx = 2 * al256(e * B * N * N), pool = mixed_pool_bytes(n, batch);
    size_t tot = 0;
    if (pool > zx) tot += al256(pool - zx);                                    // spill of the fp32 pool beyond Z | X
    tot += al256(sizeof(int) * B * N) * 3 + al256(sizeof(int) * (B + 2)) * 2 + al256(sizeof(int) * B) + al256(sizeof(int) * 2 * REFINE_EDGE_CAP * B) +
           al256(8 * B) * 2 + al256(8 * 64 * B) + al256(REFINE_CLUSTER_BYTES * B) + al256(e * B * N);
    return tot;
}

template <class T>
size_t eig_ws_bytes_t(int n, int batch) {
    const size_t e = sizeof(cx<T>), B = batch, N = n;
    size_t tot = 0;
    tot += al256(e * B * N * N) * 2;                                  // Z, X
    tot += al256(e * B * N * EigPlan::HNB) * 2;                       // Vp, Yp
    tot += al256(e * B * EigPlan::HNB * EigPlan::HNB);                // Tp
    tot += al256(e * B * EigPlan::HNB * N) * 2;                       // W1, W2
    tot += al256(e * B * N * 2 * EigPlan::HNB) * 2;                   // YV, BC
    tot += al256(e * B * EigPlan::HNB * EigPlan::HNB);                // Sm
    tot += al256(e * B * N * EigPlan::HGK) * 3 + al256(e * B * EigPlan::HGK * EigPlan::HGK) * 2;   // Vg, Wg, W2g, Tg, Gg
    tot += al256(e * B * EigPlan::HNB) * 2;                           // tau, tvec
    tot += al256(e * B * EigPlan::QW * EigPlan::QW);                  // U (dense link)
    {   // link log of a sweep: window unitaries + link records
        const size_t slots = (size_t)qr_log_slots(n) * qr_chains_for(batch);
        tot += al256(e * B * slots * EigPlan::QW * EigPlan::QW) + al256(sizeof(QrLink) * B * slots);
    }
    tot += al256(e * B * EigPlan::QKC * EigPlan::QNS);                // shifts
    tot += al256(sizeof(T) * B * N) + al256(sizeof(T) * 3 * B * N) + al256(sizeof(int) * 2 * B);   // balancing: D, scratch, flags
    tot += al256(sizeof(QrState) * B);
    tot += al256(sizeof(int) * 128 + sizeof(long long) * 24);
    if (eig_uses_mixed(n, batch, sizeof(T))) tot += mixed_extra_bytes(n, batch);
    return tot;
}

template <class T>
void eig_carve(EigBuffers<T>& Bf, void* A, void* ws, int n, int batch) {
    const size_t e = sizeof(cx<T>), B = batch, N = n;
    char* p = (char*)ws;
    auto take = [&](size_t bytes) { char* q = p; p += al256(bytes); return q; };
    Bf.A = (cx<T>*)A;
    Bf.Z = (cx<T>*)take(e * B * N * N);
    Bf.X = (cx<T>*)take(e * B * N * N);
    Bf.mixed_pool = nullptr;
    Bf.mixed_pool_bytes = 0;
    if (eig_uses_mixed(n, batch, sizeof(T))) {
        // Z | X | spill: the fp32 pool of the mixed route overlaps Z (later the second eigenvector buffer) and X (later G)
        const size_t zx = 2 * al256(e * B * N * N), pool = mixed_pool_bytes(n, batch);
        if (pool > zx) (void)take(pool - zx);
        Bf.mixed_pool = (char*)Bf.Z;
        Bf.mixed_pool_bytes = pool > zx ? pool : zx;
    }
    Bf.Vp = (cx<T>*)take(e * B * N * EigPlan::HNB);
    Bf.Yp = (cx<T>*)take(e * B * N * EigPlan::HNB);
    Bf.Tp = (cx<T>*)take(e * B * EigPlan::HNB * EigPlan::HNB);
    Bf.W1 = (cx<T>*)take(e * B * EigPlan::HNB * N);
    Bf.W2 = (cx<T>*)take(e * B * EigPlan::HNB * N);
    Bf.YV = (cx<T>*)take(e * B * N * 2 * EigPlan::HNB);
    Bf.BC = (cx<T>*)take(e * B * N * 2 * EigPlan::HNB);
    Bf.Sm = (cx<T>*)take(e * B * EigPlan::HNB * EigPlan::HNB);
    Bf.Vg = (cx<T>*)take(e * B * N * EigPlan::HGK);
    Bf.Wg = (cx<T>*)take(e * B * N * EigPlan::HGK);
    Bf.W2g = (cx<T>*)take(e * B * N * EigPlan::HGK);
    Bf.Tg = (cx<T>*)take(e * B * EigPlan::HGK * EigPlan::HGK);
    Bf.Gg = (cx<T>*)take(e * B * EigPlan::HGK * EigPlan::HGK);
    Bf.tau = (cx<T>*)take(e * B * EigPlan::HNB);
    Bf.tvec = (cx<T>*)take(e * B * EigPlan::HNB);
    Bf.U = (cx<T>*)take(e * B * EigPlan::QW * EigPlan::QW);
    {
        const size_t slots = (size_t)qr_log_slots(n) * qr_chains_for(batch);
        Bf.Ulog = (cx<T>*)take(e * B * slots * EigPlan::QW * EigPlan::QW);
        Bf.links = (QrLink*)take(sizeof(QrLink) * B * slots);
    }
    Bf.shifts = (cx<T>*)take(e * B * EigPlan::QKC * EigPlan::QNS);
    Bf.bal_d = (T*)take(sizeof(T) * B * N);
    Bf.bal_w = (T*)take(sizeof(T) * 3 * B * N);
    Bf.bal_flags = (int*)take(sizeof(int) * 2 * B);
    Bf.st = (QrState*)take(sizeof(QrState) * B);
    Bf.summary = (int*)take(sizeof(int) * 128 + sizeof(long long) * 24);  // up to 8 iteration groups x 16 ints
    Bf.r_piv = Bf.r_linfo = Bf.r_flags = Bf.r_partner = Bf.r_clus = Bf.r_edges = Bf.r_ecount = nullptr;
    Bf.r_eoff = Bf.r_lmax = Bf.r_scan = nullptr;
    Bf.r_tab = nullptr;
    Bf.r_d0 = nullptr;
    if (eig_uses_mixed(n, batch, sizeof(T))) {
        Bf.r_piv = (int*)take(sizeof(int) * B * N);
        Bf.r_partner = (int*)take(sizeof(int) * B * N);
        Bf.r_clus = (int*)take(sizeof(int) * B * N);
        Bf.r_linfo = (int*)take(sizeof(int) * (B + 2));
        Bf.r_flags = (int*)take(sizeof(int) * (B + 2));
        Bf.r_ecount = (int*)take(sizeof(int) * B);
        Bf.r_edges = (int*)take(sizeof(int) * 2 * REFINE_EDGE_CAP * B);
        Bf.r_eoff = (T*)take(8 * B);
        Bf.r_lmax = (T*)take(8 * B);
        Bf.r_scan = (T*)take(8 * 64 * B);
        Bf.r_tab = take(REFINE_CLUSTER_BYTES * B);
        Bf.r_d0 = (cx<T>*)take(e * B * N);
    }
}

namespace {
__global__ __launch_bounds__(256) void cvt_c128_c64_kernel(const cx<double>* __restrict__ in, cx<float>* __restrict__ out, long count) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) { const cx<double> v = in[i]; out[i] = cx<float>((float)v.x, (float)v.y); }
}
int eig_mixed_convert(hipStream_t s, const cx<double>* in, cx<float>* out, long count) {
    TRX_LAUNCH(cvt_c128_c64_kernel, dim3(cdiv_i(count, 256)), dim3(256), 0, s, in, out, count);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

template <class T>
__global__ __launch_bounds__(256) void clear_below_subdiag_kernel(cx<T>* __restrict__ Aall, int n) {
    cx<T>* A = Aall + (long)blockIdx.z * n * n;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j < n && i > j + 1) A[(long)i * n + j] = cx<T>(T(0), T(0));
}

// all-fp64 (or all-fp32) pipeline after the balancing
template <class T>
int eig_after_balance(hipStream_t s, const EigBuffers<T>& B, void* w, void* V, int n, int batch, int* info);

// gather / scatter of whole matrices and rows by an index list (sub-batch of the mixed route's fallback)
template <class T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T* __restrict__ src, T* __restrict__ dst, const int* __restrict__ idx, long row, int to_compact) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= row) return;
    const long j = blockIdx.y, b = idx[j];
    if (to_compact) dst[j * row + i] = src[b * row + i];
    else dst[b * row + i] = src[j * row + i];
}

// All-fp64 solve of the matrices bad[b] != 0 (nf of them) as ONE compact sub-batch inside the caller's workspace: the sub-batch's own buffers
// (eig_carve for nf matrices) at the front, its input copies / outputs / index list at the tail.  TRX_ERR_WORKSPACE: no room (the caller then
// redoes the whole batch).  bal_d: the scaling of the whole batch (its rows are gathered for the undo of the balancing).  Synchronises the
// stream (host-side index list) and allocates host vectors: like the rest of trx_eig it cannot run under stream capture.
template <class T>
int eig_redo_subset(hipStream_t s, void* A, void* w, void* V, int n, int batch, int* info, void* ws, size_t ws_bytes, const T* bal_d, const std::vector<int>& bad, int nf) {
    const size_t e = sizeof(cx<T>), N = n, NF = nf;
    struct RouteGuard { int save; RouteGuard() : save(tl_eig_vec) { tl_eig_vec = 1; } ~RouteGuard() { tl_eig_vec = save; } } guard;      // the sub-batch: all-fp64 layout and route
    const size_t front = eig_ws_bytes_t<T>(n, nf);
    const size_t tail = al256(e * NF * N * N) * 2 + al256(e * NF * N) + al256(sizeof(T) * NF * N) + al256(sizeof(int) * NF) * 2;
    if (front + tail > ws_bytes) return TRX_ERR_WORKSPACE;
    char* p = (char*)ws + ws_bytes - tail;
    p = (char*)(((size_t)p) & ~(size_t)255);
    if (p < (char*)ws + front) return TRX_ERR_WORKSPACE;
    auto take = [&](size_t bytes) { char* q = p; p += al256(bytes); return q; };
    // the small pieces first (they sit at the very end of the workspace, in what the refinement's tables occupied), then the big ones
    char* p_save = p;
    p = p_save + al256(e * NF * N * N) * 2;
    cx<T>* wsub = (cx<T>*)take(e * NF * N);
    T* dsub = (T*)take(sizeof(T) * NF * N);
    int* idx = (int*)take(sizeof(int) * NF);
    int* isub = (int*)take(sizeof(int) * NF);
    p = p_save;
    // The first gather reads the scaling rows of the WHOLE batch (bal_d, full-batch layout) and writes dsub; the sub-batch's big copies
    // (Asub, Vsub) may lie over bal_d -- they are written AFTER that gather, on the same stream -- but the small pieces must not: they sit in
    // what the refinement's tables occupied, behind bal_d in the layout.  Checked here instead of argued: no room -> the caller
    // redoes the whole batch.
    if ((const char*)wsub < (const char*)(bal_d + (size_t)batch * n)) return TRX_ERR_WORKSPACE;
    cx<T>* Asub = (cx<T>*)take(e * NF * N * N);
    cx<T>* Vsub = (cx<T>*)take(e * NF * N * N);
    std::vector<int> hidx;
    for (int b = 0; b < batch; ++b) if (bad[b]) hidx.push_back(b);
    if ((int)hidx.size() != nf) return TRX_ERR_ARG;
    if (hipMemcpyAsync(idx, hidx.data(), sizeof(int) * NF, hipMemcpyHostToDevice, s) != hipSuccess) return TRX_ERR_LAUNCH;
    if (hipStreamSynchronize(s) != hipSuccess) return TRX_ERR_LAUNCH;          // (hidx is a host temporary)
    TRX_LAUNCH((gather_rows_kernel<T>), dim3(cdiv_i(n, 256), nf), dim3(256), 0, s, bal_d, dsub, (const int*)idx, (long)n, 1);
    TRX_LAUNCH((gather_rows_kernel<cx<T>>), dim3(cdiv_i((long)n * n, 256), nf), dim3(256), 0, s, (const cx<T>*)A, Asub, (const int*)idx, (long)n * n, 1);
    EigBuffers<T> S;
    eig_carve<T>(S, Asub, ws, n, nf);
    if (hipMemcpyAsync(S.bal_d, dsub, sizeof(T) * NF * N, hipMemcpyDeviceToDevice, s) != hipSuccess) return TRX_ERR_LAUNCH;
    if (hipMemsetAsync(isub, 0, sizeof(int) * NF, s) != hipSuccess) return TRX_ERR_LAUNCH;
    int rc = eig_after_balance<T>(s, S, wsub, Vsub, n, nf, isub);
    if (rc) return rc;
    TRX_LAUNCH((gather_rows_kernel<cx<T>>), dim3(cdiv_i(n, 256), nf), dim3(256), 0, s, (const cx<T>*)wsub, (cx<T>*)w, (const int*)idx, (long)n, 0);
    TRX_LAUNCH((gather_rows_kernel<cx<T>>), dim3(cdiv_i((long)n * n, 256), nf), dim3(256), 0, s, (const cx<T>*)Vsub, (cx<T>*)V, (const int*)idx, (long)n * n, 0);
    TRX_LAUNCH((gather_rows_kernel<int>), dim3(1, nf), dim3(256), 0, s, (const int*)isub, info, (const int*)idx, 1L, 0);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

template <class T>
int eig_t(hipStream_t s, void* A, void* w, void* V, int n, int batch, int* info, void* ws, size_t ws_bytes) {
    EigBuffers<T> B;
    eig_carve<T>(B, A, ws, n, batch);
    if (hipMemsetAsync(info, 0, sizeof(int) * batch, s) != hipSuccess) return TRX_ERR_LAUNCH;
    int rc;
    { ProfScope ph(PROF_PH_BALANCE, s, 0, 0);
      rc = balance<T>(s, B, n, batch); }       // A <- D^-1 A D (zgebal 'S'); undone on the eigenvectors in schur_vectors
    if (rc) return rc;
    if constexpr (sizeof(T) == 8) {
        if (eig_uses_mixed(n, batch, sizeof(T))) {
            // Mixed-precision route (eig_refine.hip): the balanced matrix stays intact in fp64; its fp32 copy goes through the fp32 pipeline
            // (which balances once more: a no-op up to rounding) and the result is refined by Newton steps made of fp64 GEMMs and one LU.
            const size_t Bn = batch, N = n;
            char* p = B.mixed_pool;
            cx<float>* A32 = (cx<float>*)p; p += al256(8 * Bn * N * N);
            void* ws32 = p; p += al256(eig_ws_bytes_f32(n, batch));
            cx<float>* V32 = (cx<float>*)p; p += al256(8 * Bn * N * N);           // behind Z: A32 and the fp32 solver's Z32 alone fill it
            cx<float>* w32 = (cx<float>*)p;
            rc = eig_mixed_convert(s, (const cx<double>*)A, A32, (long)Bn * N * N);
            if (rc) return rc;
            rc = eig_t<float>(s, A32, w32, V32, n, batch, B.r_linfo, ws32, eig_ws_bytes_f32(n, batch));      // its info reaches eig_refine in R.linfo and is folded into the flags there
            if (rc) return rc;
            RefineBuffers<T> R;
            // Z (dead A32 / Z32 of the fp32 solve by now) takes the LU of the fp32 start and E; X takes the fp64 residual.  V32 lies at the END of
            // the pool (behind Z, inside X and the spill): it is copied / converted before X is first written.
            R.Rb = B.X; R.LU32 = (cx<float>*)B.Z; R.E32 = R.LU32 + Bn * N * N; R.d0 = B.r_d0; R.piv = B.r_piv; R.linfo = B.r_linfo; R.flags = B.r_flags;
            R.eoff = B.r_eoff; R.lmax = B.r_lmax; R.scan_part = B.r_scan; R.partner = B.r_partner; R.clus = B.r_clus; R.edges = B.r_edges; R.ecount = B.r_ecount; R.tab = B.r_tab;
            int any = 0;
            std::vector<int> bad(batch, 0);
            // a failed fp32 solve shows up as non-finite input of the refinement (flag 1)
            { ProfScope ph(PROF_PH_REFINE, s, 0, 0);
              rc = eig_refine<T>(s, R, (const cx<T>*)A, V32, w32, (cx<T>*)w, (cx<T>*)V, n, batch, tl_refine > 0 ? tl_refine : refine_steps(), &any, bad.data()); }
            if (rc) return rc;
            tl_last_fallback = any;
            if (!any) return finish_vectors<T>(s, B, n, batch, (cx<T>*)V);
            // Some matrices have a cluster beyond the exact treatment, an fp32 result that is too far off, or a singular V: redo THOSE in fp64
            // as a compact sub-batch (A is still the balanced input, the scaling D is kept) when they are few and the workspace has room for
            // the sub-batch next to its own buffers; otherwise the whole batch.
            if (3 * any <= batch) {
                rc = finish_vectors<T>(s, B, n, batch, (cx<T>*)V);          // everybody (the flagged ones are overwritten below); last reader of B
                if (rc) return rc;
                rc = eig_redo_subset<T>(s, A, w, V, n, batch, info, ws, ws_bytes, B.bal_d, bad, any);
                if (rc != TRX_ERR_WORKSPACE) return rc;
            }
            tl_last_fallback = batch;
        }
    }
    return eig_after_balance<T>(s, B, w, V, n, batch, info);
}

template <class T>
int eig_after_balance(hipStream_t s, const EigBuffers<T>& B, void* w, void* V, int n, int batch, int* info) {
    int rc;
    { ProfScope ph(PROF_PH_HESSENBERG, s, 0, 0);
      rc = hessenberg<T>(s, B, n, batch);
      if (rc) return rc;
      TRX_LAUNCH((clear_below_subdiag_kernel<T>), dim3(cdiv_i(n, 256), n, batch), dim3(256), 0, s, B.A, n); }
    // Schur vectors: T = Z^H H Z with Z accumulated, eigenvectors of T by blocked back-substitution, back-transform.  (An eigenvalues-only
    // QR phase + inverse iteration on H was built in round 3, lost at every batch size -- profiles/r03_invit_route.txt -- and was removed.)
    { ProfScope ph(PROF_PH_QR, s, 0, 0);
      rc = hessenberg_qr<T>(s, B, n, batch, info); }
    if (rc) return rc;
    ProfScope ph(PROF_PH_VECTORS, s, 0, 0);
    return schur_vectors<T>(s, B, n, batch, (cx<T>*)w, (cx<T>*)V);
}
}  // namespace

// ---- adjoint of the eigendecomposition: torcwa/torch_eig.py:19-44 (the reference's Lorentzian-broadened formula) ------------
//   s_ij = w_j - w_i,  F = conj(s) / (|s|^2 + eps), F_ii = 0
//   gA = (V^H)^-1 (diag(gw) + conj(F) o (V^H gV)) V^H
// inner <- diag(gw) + conj(F) o M   (in place on M = V^H gV)
template <class T>
__global__ __launch_bounds__(256) void eigbwd_inner_kernel(const cx<T>* __restrict__ w, const cx<T>* __restrict__ gw, cx<T>* __restrict__ M, int n, T eps) {
    const int b = blockIdx.z, i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const cx<T>* wb = w + (long)b * n;
    cx<T>* p = M + ((long)b * n + i) * n + j;
    if (i == j) { *p = gw[(long)b * n + i]; return; }
    const cx<T> sij = wb[j] - wb[i];
    const T den = norm2(sij) + eps;
    *p = cx<T>(sij.x / den, sij.y / den) * (*p);          // conj(F_ij) = s_ij / (|s_ij|^2 + eps)
}
// out[b] = conj(in[b])^T
template <class T>
__global__ __launch_bounds__(256) void conj_transpose_kernel(const cx<T>* __restrict__ in, cx<T>* __restrict__ out, int n) {
    __shared__ cx<T> tile[32][33];
    in += (long)blockIdx.z * n * n;
    out += (long)blockIdx.z * n * n;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + threadIdx.x;
        if (r < n && c < n) tile[i][threadIdx.x] = conj(in[(long)r * n + c]);
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += 8) {
        const int r = c0 + i, c = r0 + threadIdx.x;
        if (r < n && c < n) out[(long)r * n + c] = tile[threadIdx.x][i];
    }
}
template <class T>
int eig_backward_t(hipStream_t s, const cx<T>* w, const cx<T>* V, const cx<T>* gw, const cx<T>* gV, double eps, int n, int batch, cx<T>* gA,
                   int* piv, int* info, cx<T>* ws) {
    const long nn = (long)n * n;
    const cx<T> one(T(1), T(0)), zero(T(0), T(0));
    cx<T>* M = ws;                             // [B,n,n]  V^H gV -> inner
    cx<T>* XH = ws + (long)batch * nn;         // [B,n,n]  V^H (factored in place)
    int rc = gemm<T>(s, TRX_OP_C, TRX_OP_N, n, n, n, one, V, n, nn, gV, n, nn, zero, M, n, nn, batch); if (rc) return rc;
    TRX_LAUNCH((eigbwd_inner_kernel<T>), dim3(cdiv_i(n, 256), n, batch), dim3(256), 0, s, w, gw, M, n, (T)eps);
    rc = gemm<T>(s, TRX_OP_N, TRX_OP_C, n, n, n, one, M, n, nn, V, n, nn, zero, gA, n, nn, batch); if (rc) return rc;      // inner V^H
    TRX_LAUNCH((conj_transpose_kernel<T>), dim3(cdiv_i(n, 32), cdiv_i(n, 32), batch), dim3(32, 8), 0, s, V, XH, n);
    rc = lu_factor<T>(s, XH, n, nn, n, piv, batch, info); if (rc) return rc;
    rc = lu_solve<T>(s, XH, n, nn, n, piv, gA, n, nn, n, batch); if (rc) return rc;
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

template size_t eig_ws_bytes_t<float>(int, int);
template size_t eig_ws_bytes_t<double>(int, int);
template void eig_carve<float>(EigBuffers<float>&, void*, void*, int, int);
template void eig_carve<double>(EigBuffers<double>&, void*, void*, int, int);

}  // namespace trx

using trx::cx;

extern "C" size_t trx_eig_backward_ws_bytes(int dtype, int n, int batch) {
    return (size_t)(dtype == TRX_C128 ? 16 : 8) * 2 * (size_t)batch * n * n;
}

extern "C" int trx_eig_backward(int dtype, const void* w, const void* V, const void* gw, const void* gV, double broadening, int n, int batch,
                                void* gA, int* piv, int* info, void* ws, size_t ws_bytes, void* stream) {
    if (!w || !V || !gw || !gV || !gA || !piv || !info || !ws || n <= 0 || batch <= 0 || !(broadening >= 0.0)) return TRX_ERR_ARG;
    if (ws_bytes < trx_eig_backward_ws_bytes(dtype, n, batch)) return TRX_ERR_WORKSPACE;
    hipStream_t s = trx::api_stream(stream);
    if (dtype == TRX_C64)
        return trx::eig_backward_t<float>(s, (const cx<float>*)w, (const cx<float>*)V, (const cx<float>*)gw, (const cx<float>*)gV, broadening, n, batch,
                                     (cx<float>*)gA, piv, info, (cx<float>*)ws);
    if (dtype == TRX_C128)
        return trx::eig_backward_t<double>(s, (const cx<double>*)w, (const cx<double>*)V, (const cx<double>*)gw, (const cx<double>*)gV, broadening, n, batch,
                                      (cx<double>*)gA, piv, info, (cx<double>*)ws);
    return TRX_ERR_DTYPE;
}

extern "C" int trx_tuning(const char* key, int value) {
    if (!key) return TRX_ERR_ARG;
    int rc = trx::qr_set_knob(key, value);
    if (rc != TRX_OK) rc = trx::lu_set_knob(key, value);
    if (rc != TRX_OK) rc = trx::eig_set_knob(key, value);
    if (rc != TRX_OK) rc = trx::hess_set_knob(key, value);
    if (rc != TRX_OK) rc = trx::gemm_set_knob(key, value);
    if (rc != TRX_OK) rc = trx::refine_set_knob(key, value);
    return rc;
}

extern "C" size_t trx_eig_ws_bytes(int dtype, int n, int batch) {
    if (n <= 0 || batch <= 0) return 0;
    return dtype == TRX_C128 ? trx::eig_ws_bytes_t<double>(n, batch) : trx::eig_ws_bytes_t<float>(n, batch);
}

extern "C" int trx_eig_last_fallback(void) { return trx::tl_last_fallback; }

static inline bool eig_opts_valid(unsigned opts) {       // refine steps 0 - 4; route 0 (automatic), 1 (all one precision), 3 (mixed): 2 was the removed inverse-iteration route
    const unsigned r = opts & 0xF, v = (opts >> 4) & 0xF;
    return r <= 4 && (v == 0 || v == 1 || v == 3) && !(opts >> 8);
}

extern "C" size_t trx_eig_ws_bytes_opts(int dtype, int n, int batch, unsigned opts) {
    if (!eig_opts_valid(opts)) return 0;
    trx::EigCallOpts guard(opts);
    return trx_eig_ws_bytes(dtype, n, batch);
}

extern "C" int trx_eig_opts(int dtype, void* A, void* w, void* V, int n, int batch, int* info, void* ws, size_t ws_bytes, void* stream, unsigned opts) {
    if (!eig_opts_valid(opts)) return TRX_ERR_ARG;
    trx::EigCallOpts guard(opts);
    return trx_eig(dtype, A, w, V, n, batch, info, ws, ws_bytes, stream);
}

extern "C" int trx_eig(int dtype, void* A, void* w, void* V, int n, int batch, int* info, void* ws, size_t ws_bytes,
                       void* stream) {
    if (!A || !w || !V || !info || !ws || n <= 0 || batch <= 0) return TRX_ERR_ARG;
    if (dtype != TRX_C64 && dtype != TRX_C128) return TRX_ERR_DTYPE;
    if (ws_bytes < trx_eig_ws_bytes(dtype, n, batch)) return TRX_ERR_WORKSPACE;
    hipStream_t s = trx::api_stream(stream);
    trx::tl_last_fallback = 0;
    return dtype == TRX_C128 ? trx::eig_t<double>(s, A, w, V, n, batch, info, ws, ws_bytes) : trx::eig_t<float>(s, A, w, V, n, batch, info, ws, ws_bytes);
}
