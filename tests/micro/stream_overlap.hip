// Micro-benchmark (MI355X): do short kernels of different HIP streams overlap, and what decides it?
// Stream A runs a chain of "slab-update like" kernels (many 256-thread workgroups, LDS lA, ~tA us each), stream B a chain of
// "window like" kernels (32 workgroups of 1024 threads, LDS lB, ~tB us each).  Reported: wall time of both chains run alone and
// together.  Build: hipcc --offload-arch=gfx950 -O3 -o _build/stream_overlap stream_overlap.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void spin_kernel(long long cycles, float* out) {
    extern __shared__ char smem[];
    const long long t0 = wall_clock64();
    float acc = threadIdx.x;
    while (wall_clock64() - t0 < cycles) acc = acc * 1.000001f + 0.5f;
    if (acc == -1.f) out[0] = acc + smem[0];
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
    float* out; hipMalloc(&out, 4);
    hipStream_t sa, sb, sbh;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStreamCreateWithPriority(&sbh, hipStreamNonBlocking, hi);
    hipFuncSetAttribute((const void*)spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 300;
    const long long cycA = 8000, cycB = 5600;          // clock64 ticks at 100 MHz: 80 us, 56 us
    struct Cfg { const char* name; int gridA; int ldsA; int ldsB; bool prio; };
    const Cfg cfgs[] = {{"A 512wg x 74KB | B 32wg x 133KB", 512, 74 * 1024, 133 * 1024, false},
                        {"A 512wg x 74KB | B 32wg x  80KB", 512, 74 * 1024, 80 * 1024, false},
                        {"A 512wg x 74KB | B 32wg x  80KB, B high priority", 512, 74 * 1024, 80 * 1024, true},
                        {"A 448wg x 74KB | B 32wg x  80KB", 448, 74 * 1024, 80 * 1024, false},
                        {"A 2048wg x 74KB (short wgs: 20 us) | B 32wg x 80KB", 2048, 74 * 1024, 80 * 1024, false},
                        {"A 2048wg x 74KB (short wgs: 20 us) | B 32wg x 80KB, B high priority", 2048, 74 * 1024, 80 * 1024, true},
                        {"A 256wg x 74KB | B 32wg x 133KB", 256, 74 * 1024, 133 * 1024, false}};
    for (const Cfg& c : cfgs) {
        hipStream_t b = c.prio ? sbh : sb;
        const long long ca = c.gridA == 2048 ? cycA / 4 : cycA;
        auto runA = [&]() { for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(spin_kernel, dim3(c.gridA), dim3(256), c.ldsA, sa, ca, out); };
        auto runB = [&]() { for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(spin_kernel, dim3(32), dim3(1024), c.ldsB, b, cycB, out); };
        hipDeviceSynchronize();
        double t0 = now(); runA(); hipDeviceSynchronize(); const double ta = now() - t0;
        t0 = now(); runB(); hipDeviceSynchronize(); const double tb = now() - t0;
        t0 = now();
        for (int i = 0; i < iters; ++i) {
            hipLaunchKernelGGL(spin_kernel, dim3(c.gridA), dim3(256), c.ldsA, sa, ca, out);
            hipLaunchKernelGGL(spin_kernel, dim3(32), dim3(1024), c.ldsB, b, cycB, out);
        }
        hipDeviceSynchronize();
        const double tab = now() - t0;
        printf("%-78s alone A %.1f ms, B %.1f ms; together %.1f ms (sum %.1f, max %.1f)\n", c.name, ta * 1e3, tb * 1e3, tab * 1e3, (ta + tb) * 1e3, (ta > tb ? ta : tb) * 1e3);
    }
    return 0;
}
