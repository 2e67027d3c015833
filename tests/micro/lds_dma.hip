// Micro-test of the gfx950 direct global -> LDS load (global_load_lds_dwordx4) issued from inline assembly, as used by the column
// ring of the inverse-iteration kernel (eig_invit.hip, removed in round 5): where do the 16 bytes of lane L land (M0 base + L * 16 expected), does s_waitcnt vmcnt(N)
// with N loads still in flight order it against a later ds_read, and what does a deep ring of them cost per step.
//   hipcc --offload-arch=gfx950 -O3 tests/micro/lds_dma.hip -o tests/micro/_build/lds_dma && tests/micro/_build/lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void dma16(const void* gsrc_lane, unsigned lds_wave_base) {
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc_lane), "s"(lds_wave_base) : "memory", "m0");
}

__global__ __launch_bounds__(256) void place_kernel(const double2* __restrict__ src, double2* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2* buf = reinterpret_cast<double2*>(smem);
    const int t = threadIdx.x;
    for (int i = t; i < 512; i += 256) buf[i] = make_double2(-1.0, -1.0);
    __syncthreads();
    // lane t loads src[1000 - t] (a lane-dependent, non-monotone address) ; wave w writes to buf + 64 * w + 128 (offset base)
    const unsigned base = (unsigned)(size_t)buf + (unsigned)(128 + 64 * (t >> 6)) * 16u;
    dma16(src + (1000 - t), __builtin_amdgcn_readfirstlane(base));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = t; i < 512; i += 256) out[i] = buf[i];
}

// ring of R columns, column = 256 * 16 B per workgroup step; compute = a few dependent FMAs on the data D steps old
__global__ __launch_bounds__(256) void ring_kernel(const double2* __restrict__ src, double2* out, int steps, long stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double2* ring = reinterpret_cast<double2*>(smem);   // [4][256]
    const int t = threadIdx.x;
    const unsigned wbase = (unsigned)(size_t)ring + (unsigned)(t & ~63) * 16u;
    const double2* s0 = src + (long)(blockIdx.x & 7) * stride + (long)(blockIdx.x >> 3) * 4096;
    double2 acc = make_double2(0, 0);
    for (int j = 0; j < 3; ++j) dma16(s0 + (long)j * 256 + t, __builtin_amdgcn_readfirstlane(wbase + j * 4096));
    for (int j = 0; j < steps; ++j) {
        dma16(s0 + (long)(j + 3) * 256 + t, __builtin_amdgcn_readfirstlane(wbase + ((j + 3) & 3) * 4096));
        asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        __syncthreads();
        const double2 v = ring[(j & 3) * 256 + ((t * 7 + 3) & 255)];
        acc.x = fma(v.x, 1.0000001, acc.x); acc.y += v.y;
        __syncthreads();
    }
    out[blockIdx.x * 256 + t] = acc;
}

int main() {
    const int N = 1 << 22;
    std::vector<double2> h(N);
    for (int i = 0; i < N; ++i) h[i] = make_double2((double)i, 0.5 * i);
    double2 *d, *o;
    hipMalloc(&d, sizeof(double2) * (size_t)N * 8);
    hipMalloc(&o, sizeof(double2) * 256 * 1024);
    for (int r = 0; r < 8; ++r) hipMemcpy(d + (size_t)r * N, h.data(), sizeof(double2) * N, hipMemcpyHostToDevice);
    place_kernel<<<1, 256, 512 * 16>>>(d, o);
    if (hipDeviceSynchronize() != hipSuccess) { printf("place_kernel failed\n"); return 2; }
    std::vector<double2> ho(512);
    hipMemcpy(ho.data(), o, sizeof(double2) * 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 512; ++i) {
        double expect = -1.0;
        if (i >= 128 && i < 384) { const int t = i - 128; expect = 1000 - t; }
        if (ho[i].x != expect) { if (bad < 8) printf("placement mismatch at %d: got %g expected %g\n", i, ho[i].x, expect); ++bad; }
    }
    printf("placement: %s (%d mismatches)\n", bad ? "UNEXPECTED" : "lane L -> M0 base + 16 L, as assumed", bad);
    fflush(stdout);
    // ring: correctness of the ordering + time per step
    const int steps = 4000, wgs = 1024;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        ring_kernel<<<wgs, 256, 4 * 4096>>>(d, o, steps, (long)N);
        hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<double2> hr(256);
    hipMemcpy(hr.data(), o, sizeof(double2) * 256, hipMemcpyDeviceToHost);
    // reference for workgroup 0, thread 0: sum over j of src[j*256 + 3].x * 1.0000001
    double ref = 0; for (int j = 0; j < steps; ++j) ref = fma((double)(j * 256 + 3), 1.0000001, ref);
    printf("ring: thread 0 acc %.6f reference %.6f (%s); %d workgroups x %d steps in %.3f ms = %.1f ns per step per round of 256 CUs, %.2f TB/s\n", hr[0].x, ref,
           hr[0].x == ref ? "ok" : "MISMATCH", wgs, steps, ms, ms * 1e6 / steps / (wgs / 256.0), (double)wgs * steps * 4096 / ms / 1e9);
    return bad != 0;
}
