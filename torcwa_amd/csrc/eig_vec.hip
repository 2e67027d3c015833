// Eigenvectors from the Schur form: blocked triangular back-substitution, back-transform, normalisation.
// Third stage of the replacement for torch.linalg.eig (torcwa/torch_eig.py:14): LAPACK's trevc + gebak + the
// unit-2-norm scaling of geev, reorganised so that almost all work is GEMM.
//
//   T X = X diag(T), X upper triangular with unit diagonal.  Block rows are processed bottom-up:
//     R = T[I, i1:n] X[i1:n, i1:n]                       (GEMM, written into X[I, i1:n])
//     (T[I,I] - lambda_k) x_I = -R[:,k]  for every k      (trevc_block_kernel: one thread per eigenvector column,
//                                                          T[I,I] broadcast from LDS, tiny pivots perturbed as in
//                                                          LAPACK ztrevc)
//   V = Z X (GEMM), columns scaled to unit 2-norm.
#include "eig.hpp"

namespace trx {
namespace {

constexpr int VNB = EigPlan::VNB;

template <class T>
__global__ __launch_bounds__(256) void trevc_init_kernel(const cx<T>* __restrict__ Tall, cx<T>* __restrict__ Xall, cx<T>* __restrict__ wall, int n) {
    const int b = blockIdx.z, i = blockIdx.y;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    Xall[((long)b * n + i) * n + j] = cx<T>(i == j ? T(1) : T(0), T(0));
    if (i == j) wall[(long)b * n + i] = Tall[((long)b * n + i) * n + i];
}

template <class T>
__global__ __launch_bounds__(256) void trevc_block_kernel(const cx<T>* __restrict__ Tall, cx<T>* __restrict__ Xall, int n, int i0, int nbi, T smlnum) {
    __shared__ cx<T> Ts[VNB][VNB + 1];
    const int b = blockIdx.y;
    const cx<T>* Tm = Tall + (long)b * n * n;
    cx<T>* X = Xall + (long)b * n * n;
    for (int e = threadIdx.x; e < nbi * nbi; e += blockDim.x) {
        const int r = e / nbi, c = e - r * nbi;
        Ts[r][c] = Tm[(long)(i0 + r) * n + i0 + c];
    }
    __syncthreads();
    const int k = i0 + blockIdx.x * blockDim.x + threadIdx.x;     // eigenvector (column) index
    if (k >= n) return;
    const int i1 = i0 + nbi;
    const cx<T> lam = Tm[(long)k * n + k];
    T smin = eps_of<T>::value * abs1(lam);
    if (smin < smlnum) smin = smlnum;
    const int top = (k < i1) ? (k - i0) : nbi;                    // rows [0, top) of the block are unknowns
    cx<T> x[VNB];
#pragma unroll
    for (int i = 0; i < VNB; ++i) {
        cx<T> v(T(0), T(0));
        if (i < top) v = (k < i1) ? -Ts[i][k - i0] : -X[(long)(i0 + i) * n + k];
        x[i] = v;
    }
#pragma unroll
    for (int i = VNB - 1; i >= 0; --i) {
        if (i < top) {
            cx<T> s = x[i];
#pragma unroll
            for (int q = VNB - 1; q >= 0; --q)
                if (q > i && q < top) cfma(s, -Ts[i][q], x[q]);
            cx<T> d = Ts[i][i] - lam;
            if (abs1(d) < smin) d = cx<T>(smin, T(0));
            x[i] = cdiv(s, d);
        }
    }
#pragma unroll
    for (int i = 0; i < VNB; ++i)
        if (i < nbi) {
            cx<T> v(T(0), T(0));
            if (i < top) v = x[i];
            else if (k < i1 && i == k - i0) v = cx<T>(T(1), T(0));
            X[(long)(i0 + i) * n + k] = v;
        }
}

// V <- D V (undo of the balancing, zgebak: row r times d[r]) and unit column 2-norms in one pass: each block owns 64 columns and
// walks all rows (coalesced along rows)
template <class T>
__global__ __launch_bounds__(256) void colnorm_scale_kernel(cx<T>* __restrict__ Vall, int n, const T* __restrict__ dall) {
    __shared__ T part[4][64];
    const int b = blockIdx.y;
    cx<T>* V = Vall + (long)b * n * n;
    const T* d = dall + (long)b * n;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    T s = T(0);
    if (c < n)
        for (int r = rg; r < n; r += 4) s += d[r] * d[r] * norm2(V[(long)r * n + c]);
    part[rg][threadIdx.x & 63] = s;
    __syncthreads();
    const T tot = part[0][threadIdx.x & 63] + part[1][threadIdx.x & 63] + part[2][threadIdx.x & 63] + part[3][threadIdx.x & 63];
    const T sc = tot > T(0) ? T(1) / sqrt(tot) : T(1);
    if (c < n)
        for (int r = rg; r < n; r += 4) V[(long)r * n + c] = (sc * d[r]) * V[(long)r * n + c];
}

}  // namespace

template <class T>
int schur_vectors(hipStream_t s, const EigBuffers<T>& B, int n, int batch, cx<T>* w, cx<T>* V) {
    const cx<T> one(T(1), T(0)), zero(T(0), T(0));
    const long nn = (long)n * n;
    TRX_LAUNCH((trevc_init_kernel<T>), dim3(cdiv_i(n, 256), n, batch), dim3(256), 0, s, (const cx<T>*)B.A, B.X, w, n);
    const T smlnum = eps_of<T>::safmin * ((T)n / eps_of<T>::value);
    const int last = ((n - 1) / VNB) * VNB;
    for (int i0 = last; i0 >= 0; i0 -= VNB) {
        const int nbi = (n - i0 < VNB) ? n - i0 : VNB;
        const int i1 = i0 + nbi;
        if (i1 < n) {
            int rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, nbi, n - i1, n - i1, one, B.A + (long)i0 * n + i1, n, nn, B.X + (long)i1 * n + i1, n, nn,
                             zero, B.X + (long)i0 * n + i1, n, nn, batch, nullptr, 1);     // X[i1:, i1:] is upper triangular
            if (rc) return rc;
        }
        TRX_LAUNCH((trevc_block_kernel<T>), dim3(cdiv_i(n - i0, 256), batch), dim3(256), 0, s, (const cx<T>*)B.A, B.X, n, i0, nbi, smlnum);
    }
    int rc = gemm<T>(s, TRX_OP_N, TRX_OP_N, n, n, n, one, B.Z, n, nn, B.X, n, nn, zero, V, n, nn, batch, nullptr, 1);    // X upper triangular: half the K range
    if (rc) return rc;
    return finish_vectors<T>(s, B, n, batch, V);
}

template <class T>
int finish_vectors(hipStream_t s, const EigBuffers<T>& B, int n, int batch, cx<T>* V) {
    TRX_LAUNCH((colnorm_scale_kernel<T>), dim3(cdiv_i(n, 64), batch), dim3(256), 0, s, V, n, (const T*)B.bal_d);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}

template int finish_vectors<float>(hipStream_t, const EigBuffers<float>&, int, int, cx<float>*);
template int finish_vectors<double>(hipStream_t, const EigBuffers<double>&, int, int, cx<double>*);
template int schur_vectors<float>(hipStream_t, const EigBuffers<float>&, int, int, cx<float>*, cx<float>*);
template int schur_vectors<double>(hipStream_t, const EigBuffers<double>&, int, int, cx<double>*, cx<double>*);

}  // namespace trx
