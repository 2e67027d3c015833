// Batched complex GEMM, row-major interleaved complex: C = alpha*op(A)*op(B) + beta*C.
// Replaces the torch.matmul call sites of the reference hot path (torcwa/rcwa.py:1161-1164, 1226-1232,
// 1236, 1260-1281, 1287-1304).
//
// v1 kernel: LDS-tiled 64x64x16, 256 threads, 4x4 complex accumulators per thread, register-prefetched
// global->LDS staging.  (The MFMA kernel lives in gemm_mfma.hip and is selected by gemm<T>() when enabled.)
#include "common.hpp"

namespace trx {

namespace {
constexpr int BM = 64, BN = 64, BK = 16, TM = 4, TN = 4;

template <class T, int OPA, int OPB>
__global__ __launch_bounds__(256) void gemm_kernel(int m, int n, int k, cx<T> alpha, const cx<T>* __restrict__ A,
                                                   int lda, long sA, const cx<T>* __restrict__ B, int ldb, long sB,
                                                   cx<T> beta, cx<T>* __restrict__ C, int ldc, long sC,
                                                   const GemmDesc* __restrict__ desc) {
    __shared__ cx<T> As[BK][BM + 1];
    __shared__ cx<T> Bs[BK][BN + 1];
    const int b = blockIdx.z;
    A += (long)b * sA;
    B += (long)b * sB;
    C += (long)b * sC;
    if (desc) {
        const GemmDesc d = desc[b];
        m = d.m; n = d.n; k = d.k;
        A += d.offA; B += d.offB; C += d.offC;
    }
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    if (m0 >= m || n0 >= n) return;
    const int t = threadIdx.x;
    const int tx = t & 15, ty = t >> 4;

    cx<T> acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = cx<T>(T(0), T(0));

    cx<T> ra[4], rb[4];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = t + 256 * r;
            // A tile: BM x BK
            int row, kk;
            if (OPA == TRX_OP_N) { row = e >> 4; kk = e & 15; } else { kk = e >> 6; row = e & 63; }
            cx<T> v(T(0), T(0));
            if (m0 + row < m && k0 + kk < k) {
                if (OPA == TRX_OP_N) v = A[(long)(m0 + row) * lda + k0 + kk];
                else { v = A[(long)(k0 + kk) * lda + m0 + row]; if (OPA == TRX_OP_C) v = conj(v); }
            }
            ra[r] = v;
            // B tile: BK x BN
            int col, kb;
            if (OPB == TRX_OP_N) { kb = e >> 6; col = e & 63; } else { col = e >> 4; kb = e & 15; }
            cx<T> w(T(0), T(0));
            if (n0 + col < n && k0 + kb < k) {
                if (OPB == TRX_OP_N) w = B[(long)(k0 + kb) * ldb + n0 + col];
                else { w = B[(long)(n0 + col) * ldb + k0 + kb]; if (OPB == TRX_OP_C) w = conj(w); }
            }
            rb[r] = w;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = t + 256 * r;
            int row, kk, col, kb;
            if (OPA == TRX_OP_N) { row = e >> 4; kk = e & 15; } else { kk = e >> 6; row = e & 63; }
            if (OPB == TRX_OP_N) { kb = e >> 6; col = e & 63; } else { col = e >> 4; kb = e & 15; }
            As[kk][row] = ra[r];
            Bs[kb][col] = rb[r];
        }
    };

    load_tiles(0);
    for (int k0 = 0; k0 < k; k0 += BK) {
        store_tiles();
        __syncthreads();
        if (k0 + BK < k) load_tiles(k0 + BK);
#pragma unroll
        for (int kk = 0; kk < BK; ++kk) {
            cx<T> a[TM], bb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[kk][ty + 16 * i];
#pragma unroll
            for (int j = 0; j < TN; ++j) bb[j] = Bs[kk][tx + 16 * j];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) cfma(acc[i][j], a[i], bb[j]);
        }
        __syncthreads();
    }
    const bool has_beta = (beta.x != T(0)) || (beta.y != T(0));
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = m0 + ty + 16 * i;
        if (row >= m) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + tx + 16 * j;
            if (col >= n) continue;
            cx<T> v = alpha * acc[i][j];
            cx<T>* p = C + (long)row * ldc + col;
            if (has_beta) v += beta * (*p);
            *p = v;
        }
    }
}

template <class T, int OPA>
int launch_b(hipStream_t s, int opB, dim3 grid, int m, int n, int k, cx<T> alpha, const cx<T>* A, int lda, long sA,
             const cx<T>* B, int ldb, long sB, cx<T> beta, cx<T>* C, int ldc, long sC, const GemmDesc* desc) {
    switch (opB) {
        case TRX_OP_N: TRX_LAUNCH((gemm_kernel<T, OPA, TRX_OP_N>), grid, dim3(256), 0, s, m, n, k, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, desc); break;
        case TRX_OP_T: TRX_LAUNCH((gemm_kernel<T, OPA, TRX_OP_T>), grid, dim3(256), 0, s, m, n, k, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, desc); break;
        case TRX_OP_C: TRX_LAUNCH((gemm_kernel<T, OPA, TRX_OP_C>), grid, dim3(256), 0, s, m, n, k, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, desc); break;
        default: return TRX_ERR_ARG;
    }
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}
}  // namespace

template <class T>
int gemm(hipStream_t s, int opA, int opB, int m, int n, int k, cx<T> alpha, const cx<T>* A, int lda, long sA,
         const cx<T>* B, int ldb, long sB, cx<T> beta, cx<T>* C, int ldc, long sC, int batch, const GemmDesc* desc) {
    if (m <= 0 || n <= 0 || batch <= 0) return TRX_OK;
    dim3 grid(cdiv_i(n, BN), cdiv_i(m, BM), batch);
    switch (opA) {
        case TRX_OP_N: return launch_b<T, TRX_OP_N>(s, opB, grid, m, n, k, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, desc);
        case TRX_OP_T: return launch_b<T, TRX_OP_T>(s, opB, grid, m, n, k, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, desc);
        case TRX_OP_C: return launch_b<T, TRX_OP_C>(s, opB, grid, m, n, k, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, desc);
        default: return TRX_ERR_ARG;
    }
}

template int gemm<float>(hipStream_t, int, int, int, int, int, cx<float>, const cx<float>*, int, long, const cx<float>*, int, long, cx<float>, cx<float>*, int, long, int, const GemmDesc*);
template int gemm<double>(hipStream_t, int, int, int, int, int, cx<double>, const cx<double>*, int, long, const cx<double>*, int, long, cx<double>, cx<double>*, int, long, int, const GemmDesc*);

}  // namespace trx
