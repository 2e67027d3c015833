// Ad-hoc probe (not part of the library): which compute units does a stream created with hipExtStreamCreateWithCUMask use, as a
// function of the mask bits?  Every workgroup of a 4096-block launch records (XCC id, SE id, CU id) from the hardware registers; the host
// prints, per mask, the number of distinct CUs per XCD.  Answers the bit order question of torcwa_amd's CU-partitioned sweep streams
// (bit i -> which XCD / CU) before any measurement relies on it.
//   hipcc --offload-arch=gfx950 -O3 -o tests/micro/_build/cumask_probe tests/micro/cumask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

__global__ __launch_bounds__(256) void where_kernel(unsigned* out, int spin) {
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    // keep the workgroup resident for a while so that the launch spreads over every CU the queue may use
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
}

static void run(const char* name, const std::vector<unsigned>& mask) {
    hipStream_t s;
    if (hipExtStreamCreateWithCUMask(&s, (unsigned)mask.size(), mask.data()) != hipSuccess) { printf("%s: stream creation failed\n", name); return; }
    const int nb = 4096;
    unsigned* d;
    hipMalloc(&d, sizeof(unsigned) * 2 * nb);
    hipMemsetAsync(d, 0xff, sizeof(unsigned) * 2 * nb, s);
    hipLaunchKernelGGL(where_kernel, dim3(nb), dim3(256), 0, s, d, 20000);
    std::vector<unsigned> h(2 * nb);
    hipMemcpyAsync(h.data(), d, sizeof(unsigned) * 2 * nb, hipMemcpyDeviceToHost, s);
    hipStreamSynchronize(s);
    std::set<unsigned> cus[8];
    int blocks[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < nb; ++b) {
        const unsigned xcc = h[2 * b] & 0xf, hw = h[2 * b + 1];
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;      // HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
        if (xcc < 8) { cus[xcc].insert((se << 8) | (sh << 4) | cu); blocks[xcc]++; }
    }
    printf("%-34s CUs per XCD:", name);
    int tot = 0;
    for (int x = 0; x < 8; ++x) { printf(" %2zu", cus[x].size()); tot += (int)cus[x].size(); }
    printf("  (total %3d)  blocks per XCD:", tot);
    for (int x = 0; x < 8; ++x) printf(" %4d", blocks[x]);
    printf("\n");
    hipFree(d);
    hipStreamDestroy(s);
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    printf("device %s, %d CUs\n", p.name, p.multiProcessorCount);
    const int words = 8;
    std::vector<unsigned> m(words, 0xffffffffu);
    run("all 256 bits", m);
    for (int w = 0; w < words; ++w) { std::vector<unsigned> a(words, 0u); a[w] = 0xffffffffu; char nm[64]; snprintf(nm, sizeof nm, "bits [%d, %d)", 32 * w, 32 * w + 32); run(nm, a); }
    { std::vector<unsigned> a(words, 0u); a[0] = 0xffu; run("bits [0, 8)", a); }
    { std::vector<unsigned> a(words, 0u); a[0] = 0xff00u; run("bits [8, 16)", a); }
    { std::vector<unsigned> a(words, 0u); a[0] = 0x1u; run("bit 0", a); }
    { std::vector<unsigned> a(words, 0u); a[0] = 0x2u; run("bit 1", a); }
    { std::vector<unsigned> a(words, 0u); a[0] = 0x100u; run("bit 8", a); }
    { std::vector<unsigned> a(words, 0xffffffffu); a[0] = 0u; run("all but bits [0, 32)", a); }
    { std::vector<unsigned> a(words, 0xffffffffu); a[0] = 0u; a[1] = 0u; run("all but bits [0, 64)", a); }
    { std::vector<unsigned> a(words, 0u); for (int i = 0; i < 256; i += 8) a[i / 32] |= 1u << (i % 32); run("every 8th bit", a); }
    { std::vector<unsigned> a(words, 0u); for (int i = 0; i < 256; ++i) if ((i % 32) < 4) a[i / 32] |= 1u << (i % 32); run("bits with i mod 32 < 4", a); }
    return 0;
}
