// C-ABI entry points of libtrx that are thin dispatchers over the templated internals (see include/trx.h).
#include "common.hpp"

using namespace trx;

extern "C" int trx_version(void) { return 100; }   // 0.1.0

extern "C" const char* trx_strerror(int code) {
    switch (code) {
        case TRX_OK: return "ok";
        case TRX_ERR_DTYPE: return "unsupported dtype (expected TRX_C64 or TRX_C128)";
        case TRX_ERR_ARG: return "invalid argument";
        case TRX_ERR_WORKSPACE: return "workspace too small";
        case TRX_ERR_LAUNCH: return "kernel launch failed";
        case TRX_ERR_UNSUPPORTED: return "unsupported problem size or option";
        default: return code > 0 ? "numerical failure (see info[])" : "unknown error";
    }
}

namespace {
template <class T>
__global__ __launch_bounds__(256) void set_identity_kernel(cx<T>* __restrict__ X, int n) {
    cx<T>* M = X + (long)blockIdx.z * n * n;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j < n) M[(long)i * n + j] = cx<T>(i == j ? T(1) : T(0), T(0));
}

template <class T>
int inverse_t(hipStream_t s, cx<T>* A, int n, int batch, int* piv, int* info, cx<T>* ws) {
    const long nn = (long)n * n;
    int rc = lu_factor<T>(s, A, n, nn, n, piv, batch, info);
    if (rc) return rc;
    TRX_LAUNCH((set_identity_kernel<T>), dim3(cdiv_i(n, 256), n, batch), dim3(256), 0, s, ws, n);
    rc = lu_solve<T>(s, A, n, nn, n, piv, ws, n, nn, n, batch);
    if (rc) return rc;
    if (hipMemcpyAsync(A, ws, sizeof(cx<T>) * nn * batch, hipMemcpyDeviceToDevice, s) != hipSuccess) return TRX_ERR_LAUNCH;
    return TRX_OK;
}
}  // namespace

extern "C" int trx_gemm(int dtype, int opA, int opB, int m, int n, int k, const void* alpha, const void* A, int lda,
                        long strideA, const void* B, int ldb, long strideB, const void* beta, void* C, int ldc,
                        long strideC, int batch, void* stream) {
    if (!alpha || !beta || !A || !B || !C) return TRX_ERR_ARG;
    if (m < 0 || n < 0 || k < 0 || batch < 0) return TRX_ERR_ARG;
    hipStream_t s = trx::api_stream(stream);
    if (dtype == TRX_C64)
        return gemm<float>(s, opA, opB, m, n, k, *(const cx<float>*)alpha, (const cx<float>*)A, lda, strideA, (const cx<float>*)B,
                           ldb, strideB, *(const cx<float>*)beta, (cx<float>*)C, ldc, strideC, batch);
    if (dtype == TRX_C128)
        return gemm<double>(s, opA, opB, m, n, k, *(const cx<double>*)alpha, (const cx<double>*)A, lda, strideA, (const cx<double>*)B,
                            ldb, strideB, *(const cx<double>*)beta, (cx<double>*)C, ldc, strideC, batch);
    return TRX_ERR_DTYPE;
}

extern "C" int trx_lu_solve(int dtype, void* A, int n, void* B, int nrhs, int batch, int* piv, int* info, void* stream) {
    if (!A || !B || !piv || !info || n < 0 || nrhs < 0 || batch < 0) return TRX_ERR_ARG;
    hipStream_t s = trx::api_stream(stream);
    const long nn = (long)n * n, nr = (long)n * nrhs;
    if (dtype == TRX_C64) {
        int rc = lu_factor<float>(s, (cx<float>*)A, n, nn, n, piv, batch, info);
        return rc ? rc : lu_solve<float>(s, (const cx<float>*)A, n, nn, n, piv, (cx<float>*)B, nrhs, nr, nrhs, batch);
    }
    if (dtype == TRX_C128) {
        int rc = lu_factor<double>(s, (cx<double>*)A, n, nn, n, piv, batch, info);
        return rc ? rc : lu_solve<double>(s, (const cx<double>*)A, n, nn, n, piv, (cx<double>*)B, nrhs, nr, nrhs, batch);
    }
    return TRX_ERR_DTYPE;
}

extern "C" size_t trx_inverse_ws_bytes(int dtype, int n, int batch) {
    return (size_t)(dtype == TRX_C128 ? 16 : 8) * (size_t)n * n * batch;
}

extern "C" int trx_inverse(int dtype, void* A, int n, int batch, int* piv, int* info, void* ws, size_t ws_bytes, void* stream) {
    if (!A || !piv || !info || !ws || n < 0 || batch < 0) return TRX_ERR_ARG;
    if (ws_bytes < trx_inverse_ws_bytes(dtype, n, batch)) return TRX_ERR_WORKSPACE;
    hipStream_t s = trx::api_stream(stream);
    if (dtype == TRX_C64) return inverse_t<float>(s, (cx<float>*)A, n, batch, piv, info, (cx<float>*)ws);
    if (dtype == TRX_C128) return inverse_t<double>(s, (cx<double>*)A, n, batch, piv, info, (cx<double>*)ws);
    return TRX_ERR_DTYPE;
}

