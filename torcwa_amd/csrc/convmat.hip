// Fourier factorisation (Laurent rule): permittivity grid -> block-Toeplitz convolution matrix.
// Replaces torcwa/rcwa.py:1183-1204 (`_material_conv`: torch.fft.fft2 + advanced-index gather).
//
// Only the (4ox+1) x (4oy+1) Fourier coefficients that the Toeplitz gather can address are needed, so instead of
// a full nx x ny FFT the kernels evaluate a pruned, separable DFT directly (rows then columns), with exact integer
// phase reduction ((dn*y) mod ny) and twiddles/accumulation in fp64 for both dtypes, then scatter the coefficients
// into out[b,i,j] = c[b, m_i-m_j, n_i-n_j], i = (m+ox)(2oy+1)+(n+oy)  -- the same index map as rcwa.py:1187-1200.
#include "common.hpp"

namespace trx {
namespace {

typedef cx<double> zc;

// T1[b, x, q] = sum_y g[b,x,y] * exp(-2 pi i (q-2oy) y / ny)
template <class T, bool CPLX>
__global__ __launch_bounds__(128) void dft_rows_kernel(const T* __restrict__ grid, int nx, int ny, int oy, zc* __restrict__ T1) {
    TRX_DYN_SMEM(smem);
    zc* tw = reinterpret_cast<zc*>(smem);          // [ny]
    zc* row = tw + ny;                             // [ny]
    const int x = blockIdx.x, b = blockIdx.y;
    const int nq = 4 * oy + 1;
    const T* g = grid + ((long)b * nx + x) * (long)ny * (CPLX ? 2 : 1);
    for (int y = threadIdx.x; y < ny; y += blockDim.x) {
        double s, c;
        sincospi(-2.0 * (double)y / (double)ny, &s, &c);
        tw[y] = zc(c, s);
        row[y] = CPLX ? zc((double)g[2 * y], (double)g[2 * y + 1]) : zc((double)g[y], 0.0);
    }
    __syncthreads();
    for (int q = threadIdx.x; q < nq; q += blockDim.x) {
        int step = (q - 2 * oy) % ny;
        if (step < 0) step += ny;
        zc acc(0.0, 0.0);
        int idx = 0;
        for (int y = 0; y < ny; ++y) {
            cfma(acc, row[y], tw[idx]);
            idx += step;
            if (idx >= ny) idx -= ny;
        }
        T1[((long)b * nx + x) * nq + q] = acc;
    }
}

// coef[b, p, q] = (1/(nx ny)) sum_x T1[b,x,q] * exp(-2 pi i (p-2ox) x / nx)
__global__ __launch_bounds__(128) void dft_cols_kernel(const zc* __restrict__ T1, int nx, int ny, int ox, int oy, zc* __restrict__ coef) {
    TRX_DYN_SMEM(smem);
    zc* tw = reinterpret_cast<zc*>(smem);          // [nx]
    zc* colv = tw + nx;                            // [nx]
    const int q = blockIdx.x, b = blockIdx.y;
    const int nq = 4 * oy + 1, np = 4 * ox + 1;
    for (int x = threadIdx.x; x < nx; x += blockDim.x) {
        double s, c;
        sincospi(-2.0 * (double)x / (double)nx, &s, &c);
        tw[x] = zc(c, s);
        colv[x] = T1[((long)b * nx + x) * nq + q];
    }
    __syncthreads();
    const double scale = 1.0 / ((double)nx * (double)ny);
    for (int p = threadIdx.x; p < np; p += blockDim.x) {
        int step = (p - 2 * ox) % nx;
        if (step < 0) step += nx;
        zc acc(0.0, 0.0);
        int idx = 0;
        for (int x = 0; x < nx; ++x) {
            cfma(acc, colv[x], tw[idx]);
            idx += step;
            if (idx >= nx) idx -= nx;
        }
        coef[((long)b * np + p) * nq + q] = scale * acc;
    }
}

template <class T>
__global__ __launch_bounds__(256) void toeplitz_kernel(const zc* __restrict__ coef, int ox, int oy, cx<T>* __restrict__ out) {
    const int b = blockIdx.z;
    const int wy = 2 * oy + 1, N = (2 * ox + 1) * wy;
    const int nq = 4 * oy + 1, np = 4 * ox + 1;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= N) return;
    const int mi = i / wy, ni = i - mi * wy;
    const int mj = j / wy, nj = j - mj * wy;
    const zc v = coef[((long)b * np + (mi - mj + 2 * ox)) * nq + (ni - nj + 2 * oy)];
    out[((long)b * N + i) * N + j] = cx<T>((T)v.x, (T)v.y);
}

template <class T>
int convmat_t(int cplx, const void* grid, int batch, int nx, int ny, int ox, int oy, void* out, void* ws, hipStream_t s) {
    const int nq = 4 * oy + 1, np = 4 * ox + 1;
    zc* T1 = reinterpret_cast<zc*>(ws);
    zc* coef = T1 + (long)batch * nx * nq;
    const size_t sm1 = sizeof(zc) * 2 * (size_t)ny, sm2 = sizeof(zc) * 2 * (size_t)nx;
    if (cplx) TRX_LAUNCH((dft_rows_kernel<T, true>), dim3(nx, batch), dim3(128), sm1, s, (const T*)grid, nx, ny, oy, T1);
    else      TRX_LAUNCH((dft_rows_kernel<T, false>), dim3(nx, batch), dim3(128), sm1, s, (const T*)grid, nx, ny, oy, T1);
    TRX_LAUNCH(dft_cols_kernel, dim3(nq, batch), dim3(128), sm2, s, (const zc*)T1, nx, ny, ox, oy, coef);
    const int N = (2 * ox + 1) * (2 * oy + 1);
    TRX_LAUNCH((toeplitz_kernel<T>), dim3(cdiv_i(N, 256), N, batch), dim3(256), 0, s, (const zc*)coef, ox, oy, (cx<T>*)out);
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}
}  // namespace
}  // namespace trx

extern "C" size_t trx_convmat_ws_bytes(int dtype, int batch, int nx, int ny, int ox, int oy) {
    (void)dtype; (void)ny;
    return sizeof(trx::zc) * ((size_t)batch * nx * (4 * oy + 1) + (size_t)batch * (4 * ox + 1) * (4 * oy + 1));
}

extern "C" int trx_convmat(int dtype, int grid_is_complex, const void* grid, int batch, int nx, int ny, int ox, int oy,
                           void* out, void* ws, size_t ws_bytes, void* stream) {
    if (!grid || !out || !ws) return TRX_ERR_ARG;
    if (batch <= 0 || ox < 0 || oy < 0 || nx <= 2 * ox || ny <= 2 * oy) return TRX_ERR_ARG;
    if (ws_bytes < trx_convmat_ws_bytes(dtype, batch, nx, ny, ox, oy)) return TRX_ERR_WORKSPACE;
    if ((size_t)16 * 2 * (size_t)(nx > ny ? nx : ny) > 64 * 1024) return TRX_ERR_UNSUPPORTED;
    hipStream_t s = trx::api_stream(stream);
    if (dtype == TRX_C64) return trx::convmat_t<float>(grid_is_complex, grid, batch, nx, ny, ox, oy, out, ws, s);
    if (dtype == TRX_C128) return trx::convmat_t<double>(grid_is_complex, grid, batch, nx, ny, ox, oy, out, ws, s);
    return TRX_ERR_DTYPE;
}
