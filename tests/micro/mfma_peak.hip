// Ad-hoc microbenchmark (not part of the library): the sustained rate of v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 with
// register-only operands, and the shader clock it runs at.  Gives the practical ceiling that the GEMM kernels are priced
// against next to the 78.6 / 157.3 TF datasheet peaks (MI355X_MICROARCH.md).
//   hipcc --offload-arch=gfx950 -O3 -o tests/micro/_build/mfma_peak tests/micro/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_f64_loop(double* out, long long* clk, int iters) {
    f64x4 acc[NACC];
    const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = f64x4{0, 0, 0, 0};
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));   // the builtin in a loop makes hipcc bounce the tile through v_accvgpr moves
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    double s = 0;
#pragma unroll
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

template <int NACC>
__global__ __launch_bounds__(256) void mfma_f32_loop(float* out, int iters) {
    f32x4 acc[NACC];
    const float a = 1.0f + threadIdx.x * 1e-6f, b = 1.0f - threadIdx.x * 1e-6f;
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = f32x4{0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < NACC; ++j) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(a), "v"(b));
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clockRate %d kHz\n", prop.name, cus, prop.clockRate);
    double* out; long long* clk;
    CK(hipMalloc(&out, sizeof(double) * 256 * cus * 8));
    CK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int NACC = 8;
    for (int wgs_per_cu = 1; wgs_per_cu <= 4; wgs_per_cu *= 2) {
        for (int rep = 0; rep < 3; ++rep) {
            const int iters = 100000;                      // ~0.1-0.4 s per launch: long enough for the clocks to settle
            const int grid = cus * wgs_per_cu;
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((mfma_f64_loop<NACC>), dim3(grid), dim3(256), 0, 0, out, clk, iters);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
            const double flops = 2.0 * 16 * 16 * 4 * NACC * (double)iters * 4 * grid;
            printf("f64 16x16x4: %d WG/CU (= waves/SIMD), %7.2f ms, %6.1f TFLOP/s; block 0: %.0f shader cycles per MFMA-wave, shader clock %.0f MHz\n",
                   wgs_per_cu, ms, flops / ms * 1e-9, (double)h[0] / ((double)iters * NACC), (double)h[0] / ((double)h[1] / 100.0));
        }
    }
    for (int wgs_per_cu = 1; wgs_per_cu <= 4; wgs_per_cu *= 2) {
        const int iters = 40000, grid = cus * wgs_per_cu;
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((mfma_f32_loop<NACC>), dim3(grid), dim3(256), 0, 0, (float*)out, iters);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double flops = 2.0 * 16 * 16 * 4 * NACC * (double)iters * 4 * grid;
            printf("f32 16x16x4: %d WG/CU, %7.2f ms, %6.1f TFLOP/s\n", wgs_per_cu, ms, flops / ms * 1e-9);
        }
    }
    return 0;
}
