// Micro-benchmark (MI355X): the HOST LOOP of trx_eig's QR phase with spin kernels in place of the real ones.  G iteration groups, each on
// its own stream, each iteration = nwin x [chase-like kernel (32 wg x 1024 thr, 52 KB LDS, 50 us) ; slab-like kernel (W wg x 256 thr,
// 35 KB LDS, 58 us)] followed by [12-byte memset ; AED-like kernel (32 wg x 64 thr, 68 KB LDS, 2.45 ms) ; 12-byte copy to pinned host
// memory ; event].  The host polls the events and queues iteration k+2 of a group when the summary of iteration k has landed (one
// iteration ahead), exactly like hessenberg_qr().  Perfect overlap keeps the wall time at one group's chain; the traces of rounds 2 and 3
// show every group busy only 40 % of the phase.  Variants tell the causes apart:
//     group_loop <G> <W> <null0> <nocopy>     null0 = 1: group 0 on the null stream (as under PyTorch); nocopy = 1: no memset / memcpy
//                                              (the summary kernel-written into mapped host memory)
// Prints wall time, one chain alone, and per group: kernels' busy time (from events around each iteration).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
__global__ void spin_kernel(long long ticks, int* out, int value) {
    extern __shared__ char smem[];
    const long long t0 = wall_clock64();
    float acc = threadIdx.x;
    while (wall_clock64() - t0 < ticks) acc = acc * 1.000001f + 0.5f;
    if (acc == -1.f) out[1] = (int)acc + smem[0];
    if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = value;
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 4, W = argc > 2 ? atoi(argv[2]) : 512, null0 = argc > 3 ? atoi(argv[3]) : 1, nocopy = argc > 4 ? atoi(argv[4]) : 0;
    const int iters = 30, nwin = 30;
    const long long us = 100;                       // wall_clock64 ticks per microsecond (100 MHz)
    hipFuncSetAttribute((const void*)spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    struct Grp { hipStream_t s; hipEvent_t ev[2]; int* dsum; int* hsum; int issued, read; };
    std::vector<Grp> g(G);
    for (int i = 0; i < G; ++i) {
        if (i == 0 && null0) g[i].s = nullptr; else hipStreamCreateWithFlags(&g[i].s, hipStreamNonBlocking);
        hipEventCreateWithFlags(&g[i].ev[0], hipEventDisableTiming); hipEventCreateWithFlags(&g[i].ev[1], hipEventDisableTiming);
        hipMalloc(&g[i].dsum, 64); hipHostMalloc(&g[i].hsum, 64, hipHostMallocDefault);
        g[i].issued = g[i].read = 0;
    }
    auto issue = [&](Grp& a, int it) {
        const int slot = it & 1;
        for (int q = 0; q < nwin; ++q) {
            hipLaunchKernelGGL(spin_kernel, dim3(32), dim3(1024), 52 * 1024, a.s, 50 * us, (int*)nullptr, 0);
            hipLaunchKernelGGL(spin_kernel, dim3(W), dim3(256), 35 * 1024, a.s, 58 * us, (int*)nullptr, 0);
        }
        if (nocopy) {
            hipLaunchKernelGGL(spin_kernel, dim3(32), dim3(64), 68 * 1024, a.s, 2450 * us, a.hsum + 4 * slot, it + 1);
        } else {
            hipMemsetAsync(a.dsum + 4 * slot, 0, 12, a.s);
            hipLaunchKernelGGL(spin_kernel, dim3(32), dim3(64), 68 * 1024, a.s, 2450 * us, a.dsum + 4 * slot, it + 1);
            hipMemcpyAsync(a.hsum + 4 * slot, a.dsum + 4 * slot, 12, hipMemcpyDeviceToHost, a.s);
        }
        hipEventRecord(a.ev[slot], a.s);
    };
    hipDeviceSynchronize();
    const double t0 = now();
    double t_issue = 0, t_block = 0;
    for (int i = 0; i < G; ++i) { issue(g[i], 0); issue(g[i], 1); g[i].issued = 2; }
    int live = G, idle = 0;
    while (live > 0) {
        bool any = false;
        for (int i = 0; i < G; ++i) {
            Grp& a = g[i];
            if (a.read >= iters) continue;
            if (hipEventQuery(a.ev[a.read & 1]) != hipSuccess) continue;
            any = true;
            ++a.read;
            if (a.read >= iters) { --live; continue; }
            if (a.issued < iters) { const double ti = now(); issue(a, a.issued); ++a.issued; t_issue += now() - ti; }
        }
        if (any) { idle = 0; continue; }
        if (++idle < 64) { std::this_thread::yield(); continue; }
        idle = 0;
        int gm = -1;
        for (int i = 0; i < G; ++i) if (g[i].read < iters && (gm < 0 || g[i].read < g[gm].read)) gm = i;
        if (gm >= 0) { const double tb = now(); hipEventSynchronize(g[gm].ev[g[gm].read & 1]); t_block += now() - tb; }
    }
    hipDeviceSynchronize();
    (void)hipGetLastError();
    const double wall = (now() - t0) * 1e3, chain = iters * (nwin * 0.108 + 2.45);
    printf("groups %d, slab %4d wg, group 0 on %s, %s: wall %.1f ms, one chain alone %.1f ms -> %.2fx; host: issuing %.1f ms, blocked in hipEventSynchronize %.1f ms\n",
           G, W, null0 ? "the null stream" : "a created stream", nocopy ? "summary written by the kernel into mapped host memory" : "memset + memcpy per iteration",
           wall, chain, wall / chain, t_issue * 1e3, t_block * 1e3);
    return 0;
}
