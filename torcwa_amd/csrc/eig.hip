// trx_eig: batched general complex eigendecomposition A V = V diag(w) -- the MI355X replacement for the
// torch.linalg.eig call behind torcwa's `Eig.apply` (torcwa/torch_eig.py:12-17, used at rcwa.py:1236/1238).
// Pipeline: Hessenberg reduction (eig_hess.hip) -> multi-shift QR to Schur form (eig_qr.hip) -> triangular
// eigenvectors + back-transform + unit-norm scaling (eig_vec.hip).
#include "eig.hpp"

namespace trx {

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

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
    tot += al256(e * B * EigPlan::HNB);                               // tau
    tot += al256(e * B * EigPlan::QW * EigPlan::QW);                  // U
    tot += al256(e * B * EigPlan::QNS);                               // shifts
    tot += al256(sizeof(QrState) * B);
    tot += al256(sizeof(int) * 64 + sizeof(long long) * 24);
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
    Bf.Vp = (cx<T>*)take(e * B * N * EigPlan::HNB);
    Bf.Yp = (cx<T>*)take(e * B * N * EigPlan::HNB);
    Bf.Tp = (cx<T>*)take(e * B * EigPlan::HNB * EigPlan::HNB);
    Bf.W1 = (cx<T>*)take(e * B * EigPlan::HNB * N);
    Bf.W2 = (cx<T>*)take(e * B * EigPlan::HNB * N);
    Bf.YV = (cx<T>*)take(e * B * N * 2 * EigPlan::HNB);
    Bf.BC = (cx<T>*)take(e * B * N * 2 * EigPlan::HNB);
    Bf.Sm = (cx<T>*)take(e * B * EigPlan::HNB * EigPlan::HNB);
    Bf.tau = (cx<T>*)take(e * B * EigPlan::HNB);
    Bf.U = (cx<T>*)take(e * B * EigPlan::QW * EigPlan::QW);
    Bf.shifts = (cx<T>*)take(e * B * EigPlan::QNS);
    Bf.st = (QrState*)take(sizeof(QrState) * B);
    Bf.summary = (int*)take(sizeof(int) * 64 + sizeof(long long) * 24);   // up to 8 iteration groups x 8 ints
}

namespace {
template <class T>
__global__ __launch_bounds__(256) void clear_below_subdiag_kernel(cx<T>* __restrict__ Aall, int n) {
    cx<T>* A = Aall + (long)blockIdx.z * n * n;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j < n && i > j + 1) A[(long)i * n + j] = cx<T>(T(0), T(0));
}

template <class T>
int eig_t(hipStream_t s, void* A, void* w, void* V, int n, int batch, int* info, void* ws) {
    EigBuffers<T> B;
    eig_carve<T>(B, A, ws, n, batch);
    if (hipMemsetAsync(info, 0, sizeof(int) * batch, s) != hipSuccess) return TRX_ERR_LAUNCH;
    int rc = hessenberg<T>(s, B, n, batch);
    if (rc) return rc;
    TRX_LAUNCH((clear_below_subdiag_kernel<T>), dim3(cdiv_i(n, 256), n, batch), dim3(256), 0, s, B.A, n);
    rc = hessenberg_qr<T>(s, B, n, batch, info);
    if (rc) return rc;
    return schur_vectors<T>(s, B, n, batch, (cx<T>*)w, (cx<T>*)V);
}
}  // namespace

template size_t eig_ws_bytes_t<float>(int, int);
template size_t eig_ws_bytes_t<double>(int, int);
template void eig_carve<float>(EigBuffers<float>&, void*, void*, int, int);
template void eig_carve<double>(EigBuffers<double>&, void*, void*, int, int);

}  // namespace trx

extern "C" size_t trx_eig_ws_bytes(int dtype, int n, int batch) {
    if (n <= 0 || batch <= 0) return 0;
    return dtype == TRX_C128 ? trx::eig_ws_bytes_t<double>(n, batch) : trx::eig_ws_bytes_t<float>(n, batch);
}

extern "C" int trx_eig(int dtype, void* A, void* w, void* V, int n, int batch, int* info, void* ws, size_t ws_bytes,
                       void* stream) {
    if (!A || !w || !V || !info || !ws || n <= 0 || batch <= 0) return TRX_ERR_ARG;
    if (dtype != TRX_C64 && dtype != TRX_C128) return TRX_ERR_DTYPE;
    if (ws_bytes < trx_eig_ws_bytes(dtype, n, batch)) return TRX_ERR_WORKSPACE;
    hipStream_t s = trx::api_stream(stream);
    return dtype == TRX_C128 ? trx::eig_t<double>(s, A, w, V, n, batch, info, ws) : trx::eig_t<float>(s, A, w, V, n, batch, info, ws);
}
