// Micro-benchmark (MI355X): where does the fp64 GEMM lose the matrix pipe?  The 3M complex GEMM kernel of libtrx sustains 0.57-0.59 of the
// fp64 MFMA peak in ISSUED flops (73-76 TF-equivalent in the 8-flops-per-complex-MAC count) whether its operands are staged through
// registers or through a direct-to-LDS ring, while a register-only MFMA loop reaches 0.986 (mfma_peak.hip).  This ladder adds the
// ingredients of the GEMM's inner loop one at a time around the SAME 12 MFMAs per k-step (2 x 2 tiles of 16 x 16, three real products
// each), 4 waves per workgroup, 1 or 2 workgroups per CU:
//   level 0  register operands
//   level 1  + operands read from LDS every k-step (8 ds_read_b64 per wave: 2 A fragments + 2 B fragments, real and imaginary planes)
//   level 2  + the 3M operand sums (Ar + Ai, Br + Bi: 4 v_add_f64)
//   level 3  + two workgroup barriers per K slab of 4 k-steps
//   level 4  + the slab's global loads (64 x 16 + 16 x 64 complex128 per workgroup = 32 KB) and LDS stores, register-staged: loads issued
//            before the MFMAs, first used after them (a true prefetch)
//   level 5  = level 4 with the loaded registers touched right behind the loads, as the round-3 GEMM kernel does (its zeroing select of
//            out-of-range elements sits there: hipcc waits for the whole slab before the MFMAs)
//   level 6  = level 4 with every k-step's MFMAs issued twice: the arithmetic intensity of a tile twice as large (half the bytes per flop) --
//            if the utilisation jumps, the 64 x 64 tile is bound by the operand traffic (385 GB per 1922^3 x 128 GEMM at 3.9 TB/s), not by latency
//   hipcc --offload-arch=gfx950 -O3 -o tests/micro/_build/mfma_ladder tests/micro/mfma_ladder.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double f64x4 __attribute__((ext_vector_type(4)));
constexpr int LDP = 66;                 // padded leading dimension of the LDS planes (doubles)

template <int LEVEL>
__global__ __launch_bounds__(256, 2) void ladder_kernel(const double2* __restrict__ src, double* __restrict__ out, int slabs, long src_elems) {
    __shared__ double Ar[16 * LDP], Ai[16 * LDP], Br[16 * LDP], Bi[16 * LDP];        // one K slab: 16 (k) x 64 (m or n), split planes
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;                             // this wave's 32 x 32 part of the 64 x 64 tile
    for (int e = t; e < 16 * LDP; e += 256) { Ar[e] = 1.0 + e * 1e-9; Ai[e] = 0.5 - e * 1e-9; Br[e] = 0.25 + e * 1e-9; Bi[e] = 2.0 - e * 1e-9; }
    __syncthreads();
    f64x4 acc[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) acc[j] = f64x4{0, 0, 0, 0};
    double ar[2] = {1.0 + lane * 1e-9, 1.0 - lane * 1e-9}, ai[2] = {0.5, 0.25}, br[2] = {2.0, 1.5}, bi[2] = {0.75, 1.25};
    const int kk = lane >> 4, mm = lane & 15;                                          // fragment layout of v_mfma_f64_16x16x4_f64: k = lane / 16, m (or n) = lane % 16
    long g = ((long)blockIdx.x * 4096 + t) % (src_elems - 8 * 256);
    double2 stage[8];
    for (int s = 0; s < slabs; ++s) {
        asm volatile("" ::: "memory");                    // the LDS operands are re-read every slab (no hoisting out of the loop)
        if (LEVEL >= 4) {
#pragma unroll
            for (int q = 0; q < 8; ++q) stage[q] = src[g + q * 256];                   // 8 x 16 B per thread = the slab's 32 KB per workgroup
            g += 2048; if (g >= src_elems - 8 * 256) g -= (src_elems - 8 * 256);
            if (LEVEL == 5) {
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(stage[q].x), "+v"(stage[q].y));       // first use right here
            }
            __builtin_amdgcn_sched_barrier(0);      // keep the loads AHEAD of the MFMAs (hipcc otherwise sinks them to their first use)
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (LEVEL >= 1) {
                const int row = ks * 4 + kk;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    ar[i] = Ar[row * LDP + wm + 16 * i + mm]; ai[i] = Ai[row * LDP + wm + 16 * i + mm];
                    br[i] = Br[row * LDP + wn + 16 * i + mm]; bi[i] = Bi[row * LDP + wn + 16 * i + mm];
                }
            }
            double as[2], bs[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { as[i] = LEVEL >= 2 ? ar[i] + ai[i] : ar[i]; bs[i] = LEVEL >= 2 ? br[i] + bi[i] : br[i]; }
#pragma unroll
            for (int rep = 0; rep < (LEVEL == 6 ? 2 : 1); ++rep)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[3 * (2 * i + j) + 0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[i], br[j], acc[3 * (2 * i + j) + 0], 0, 0, 0);
                    acc[3 * (2 * i + j) + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai[i], bi[j], acc[3 * (2 * i + j) + 1], 0, 0, 0);
                    acc[3 * (2 * i + j) + 2] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[i], bs[j], acc[3 * (2 * i + j) + 2], 0, 0, 0);
                }
        }
        if (LEVEL >= 4) __builtin_amdgcn_sched_barrier(0);
        if (LEVEL >= 3) __syncthreads();
        if (LEVEL >= 4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {                                              // A part: element e = q * 256 + t of the 16 x 64 slab
                const int e = q * 256 + t, r = e >> 6, c = e & 63;
                Ar[r * LDP + c] = stage[q].x; Ai[r * LDP + c] = stage[q].y;
                Br[r * LDP + c] = stage[4 + q].x; Bi[r * LDP + c] = stage[4 + q].y;
            }
        }
        if (LEVEL >= 3) __syncthreads();
    }
    double sum = 0;
#pragma unroll
    for (int j = 0; j < 12; ++j) sum += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
    out[(long)blockIdx.x * 256 + t] = sum;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <int LEVEL>
static int run(int cus, const double2* src, long src_elems, double* out, const char* what) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int per_cu = 1; per_cu <= 2; ++per_cu) {
        const int grid = cus * per_cu, slabs = 20000;
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((ladder_kernel<LEVEL>), dim3(grid), dim3(256), 0, 0, src, out, slabs, src_elems);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        const double flops = 2.0 * 16 * 16 * 4 * 12 * 4.0 * slabs * 4 * grid * (LEVEL == 6 ? 2 : 1);      // real MFMA flops
        printf("level %d (%s), %d workgroup(s) per CU: %8.2f ms, %6.1f TFLOP/s issued = %.3f of 78.6\n", LEVEL, what, per_cu, best, flops / best * 1e-9, flops / best * 1e-9 / 78.6);
    }
    return 0;
}
int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const long src_elems = 64L << 20;                  // 1 GiB of complex128: larger than the caches
    double2* src; double* out;
    CK(hipMalloc(&src, sizeof(double2) * src_elems));
    CK(hipMemset(src, 0, sizeof(double2) * src_elems));
    CK(hipMalloc(&out, sizeof(double) * 256 * cus * 2));
    if (run<0>(cus, src, src_elems, out, "register operands")) return 1;
    if (run<1>(cus, src, src_elems, out, "+ LDS operand reads")) return 1;
    if (run<2>(cus, src, src_elems, out, "+ 3M operand sums")) return 1;
    if (run<3>(cus, src, src_elems, out, "+ 2 barriers per K slab")) return 1;
    if (run<4>(cus, src, src_elems, out, "+ global loads and LDS stores of the slab")) return 1;
    if (run<5>(cus, src, src_elems, out, "same, loads consumed before the MFMAs")) return 1;
    if (run<6>(cus, src, src_elems, out, "level 4 at twice the flops per byte")) return 1;
    return 0;
}
