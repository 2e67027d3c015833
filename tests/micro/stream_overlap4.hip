// Micro-benchmark (MI355X): G streams, each running the chain [window-like kernel (32 wg x 1024 thr, 80 KB LDS, 56 us) ; slab-like
// kernel (W workgroups x 256 thr, 74 KB LDS, T us)] x iters -- the launch pattern of the QR phase of trx_eig.  Reports the wall time
// against the single-stream time: perfect overlap of G chains would keep it constant.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
__global__ void spin_kernel(long long ticks, float* out) {
    extern __shared__ char smem[];
    const long long t0 = wall_clock64();
    float acc = threadIdx.x;
    while (wall_clock64() - t0 < ticks) acc = acc * 1.000001f + 0.5f;
    if (acc == -1.f) out[0] = acc + smem[0];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    float* out; hipMalloc(&out, 4);
    hipStream_t st[8];
    for (int i = 0; i < 8; ++i) hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
    hipFuncSetAttribute((const void*)spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 200;
    struct Cfg { int wgs; long long ticksA; };
    const Cfg cfgs[] = {{448, 8000}, {224, 8000}, {112, 8000}, {448, 2000}};
    for (const Cfg& c : cfgs)
        for (int G : {1, 2, 4, 8}) {
            hipDeviceSynchronize();
            const double t0 = now();
            for (int i = 0; i < iters; ++i)
                for (int g = 0; g < G; ++g) {
                    hipLaunchKernelGGL(spin_kernel, dim3(32), dim3(1024), 80 * 1024, st[g], 5600LL, out);
                    hipLaunchKernelGGL(spin_kernel, dim3(c.wgs), dim3(256), 74 * 1024, st[g], c.ticksA, out);
                }
            hipDeviceSynchronize();
            printf("slab %4d wg x %3lld us, %d streams: %.1f ms  (one chain alone: %.1f ms)\n", c.wgs, c.ticksA / 100, G, (now() - t0) * 1e3, iters * (56 + c.ticksA / 100.0) * 1e-3);
        }
    return 0;
}
