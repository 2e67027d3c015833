// Shared device/host helpers for libtrx (MI355X / gfx950 RCWA layer-solve kernels).
#pragma once
#include <cstdio>
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/trx.h"

#ifndef TRX_DYN_SMEM
// Dynamic LDS region, 16-byte aligned (cdna_hip_programming.md Guideline 17: no static LDS in front of it).
#define TRX_DYN_SMEM(name) extern __shared__ __attribute__((aligned(16))) char name[]
#endif

#ifndef TRX_LDS_DMA16
// Direct global -> LDS load of 16 bytes per lane (gfx950 global_load_lds_dwordx4): lane L's 16 bytes land at lds_wave_base + 16 L
// (lds_wave_base wave-uniform; placement and ordering pinned on hardware by tests/micro/lds_dma.hip).  No destination registers, and --
// being inline assembly -- invisible to the compiler's own wait-count bookkeeping: the kernel waits with TRX_WAIT_VMCNT(n) ("at most
// n younger vector-memory operations of this wave still in flight") before a barrier that publishes the data.  (The compiler's builtin
// for the same instruction makes it wait for ALL of them in front of every later LDS read, which would undo the prefetch.)
#define TRX_LDS_DMA16(gsrc_lane, lds_wave_base)                                                                          \
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"((const void*)(gsrc_lane)),   \
                 "s"(__builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_wave_base)))                                    \
                 : "memory", "m0")
// The same with a wave-uniform 64-bit base (SGPR pair) + a 32-bit byte offset per lane: no 64-bit vector address arithmetic per load.
#define TRX_LDS_DMA16_S(sbase, voff32, lds_wave_base)                                                                                 \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"((unsigned)(voff32)), "s"((const void*)(sbase)),   \
                 "s"(__builtin_amdgcn_readfirstlane((unsigned)(size_t)(lds_wave_base)))                                               \
                 : "memory", "m0")
#define TRX_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
// the same with any constant expression 0 .. 63 (gfx9 encodes a 6-bit vmcnt)
#define TRX_WAIT_VMCNT_IMM(expr) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(expr) : "memory")
#endif
// the same with a compile-time constant expression (0 .. 15)
#define TRX_WAIT_VMCNT_N(expr)                                                                     \
    do {                                                                                           \
        constexpr int n__ = (expr);                                                                \
        static_assert(n__ >= 0 && n__ <= 15, "vmcnt immediate");                                   \
        if constexpr (n__ == 0) TRX_WAIT_VMCNT(0); else if constexpr (n__ == 1) TRX_WAIT_VMCNT(1); \
        else if constexpr (n__ == 2) TRX_WAIT_VMCNT(2); else if constexpr (n__ == 3) TRX_WAIT_VMCNT(3); \
        else if constexpr (n__ == 4) TRX_WAIT_VMCNT(4); else if constexpr (n__ == 5) TRX_WAIT_VMCNT(5); \
        else if constexpr (n__ == 6) TRX_WAIT_VMCNT(6); else if constexpr (n__ == 7) TRX_WAIT_VMCNT(7); \
        else if constexpr (n__ == 8) TRX_WAIT_VMCNT(8); else if constexpr (n__ == 9) TRX_WAIT_VMCNT(9); \
        else if constexpr (n__ == 10) TRX_WAIT_VMCNT(10); else if constexpr (n__ == 11) TRX_WAIT_VMCNT(11); \
        else if constexpr (n__ == 12) TRX_WAIT_VMCNT(12); else if constexpr (n__ == 13) TRX_WAIT_VMCNT(13); \
        else if constexpr (n__ == 14) TRX_WAIT_VMCNT(14); else TRX_WAIT_VMCNT(15);                  \
    } while (0)

#define TRX_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(HIP_KERNEL_NAME(kernel), grid, block, shmem, stream, __VA_ARGS__)

namespace trx {

// ---- interleaved complex (layout == torch.complex64 / complex128) -----------------------------------
template <class T>
struct alignas(2 * sizeof(T)) cx {
    T x, y;
    __host__ __device__ cx() = default;
    __host__ __device__ constexpr cx(T re, T im = T(0)) : x(re), y(im) {}
};
template <class T> __host__ __device__ __forceinline__ cx<T> operator+(cx<T> a, cx<T> b) { return {a.x + b.x, a.y + b.y}; }
template <class T> __host__ __device__ __forceinline__ cx<T> operator-(cx<T> a, cx<T> b) { return {a.x - b.x, a.y - b.y}; }
template <class T> __host__ __device__ __forceinline__ cx<T> operator-(cx<T> a) { return {-a.x, -a.y}; }
template <class T> __host__ __device__ __forceinline__ cx<T> operator*(cx<T> a, cx<T> b) {
    return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
template <class T> __host__ __device__ __forceinline__ cx<T> operator*(T s, cx<T> a) { return {s * a.x, s * a.y}; }
template <class T> __host__ __device__ __forceinline__ cx<T> operator*(cx<T> a, T s) { return {s * a.x, s * a.y}; }
template <class T> __host__ __device__ __forceinline__ cx<T>& operator+=(cx<T>& a, cx<T> b) { a.x += b.x; a.y += b.y; return a; }
template <class T> __host__ __device__ __forceinline__ cx<T>& operator-=(cx<T>& a, cx<T> b) { a.x -= b.x; a.y -= b.y; return a; }
template <class T> __host__ __device__ __forceinline__ cx<T> conj(cx<T> a) { return {a.x, -a.y}; }
template <class T> __host__ __device__ __forceinline__ T norm2(cx<T> a) { return a.x * a.x + a.y * a.y; }
template <class T> __host__ __device__ __forceinline__ T abs1(cx<T> a) { return fabs(a.x) + fabs(a.y); }   // LAPACK cabs1
template <class T> __host__ __device__ __forceinline__ T cabs(cx<T> a) { return hypot(a.x, a.y); }
// acc += a*b with explicit FMAs (4 real FMAs)
template <class T> __host__ __device__ __forceinline__ void cfma(cx<T>& acc, cx<T> a, cx<T> b) {
    acc.x = fma(a.x, b.x, acc.x);
    acc.x = fma(-a.y, b.y, acc.x);
    acc.y = fma(a.x, b.y, acc.y);
    acc.y = fma(a.y, b.x, acc.y);
}
// acc += conj(a)*b
template <class T> __host__ __device__ __forceinline__ void cfma_conj(cx<T>& acc, cx<T> a, cx<T> b) {
    acc.x = fma(a.x, b.x, acc.x);
    acc.x = fma(a.y, b.y, acc.x);
    acc.y = fma(a.x, b.y, acc.y);
    acc.y = fma(-a.y, b.x, acc.y);
}
// Smith-style robust complex division a/b
template <class T> __host__ __device__ __forceinline__ cx<T> cdiv(cx<T> a, cx<T> b) {
    if (fabs(b.x) >= fabs(b.y)) {
        T r = b.y / b.x, d = b.x + b.y * r;
        return {(a.x + a.y * r) / d, (a.y - a.x * r) / d};
    } else {
        T r = b.x / b.y, d = b.x * r + b.y;
        return {(a.x * r + a.y) / d, (a.y * r - a.x) / d};
    }
}
template <class T> __host__ __device__ __forceinline__ cx<T> crecip(cx<T> b) { return cdiv(cx<T>(T(1), T(0)), b); }
// principal square root
template <class T> __host__ __device__ __forceinline__ cx<T> csqrt(cx<T> z) {
    T r = hypot(z.x, z.y);
    if (r == T(0)) return {T(0), T(0)};
    T u = sqrt(T(0.5) * (r + fabs(z.x)));
    T v = z.y / (T(2) * u);
    if (z.x >= T(0)) return {u, v};
    return cx<T>((T)fabs(v), (T)copysign(u, z.y));
}
template <class T> __host__ __device__ __forceinline__ cx<T> cexp(cx<T> z) {
    T e = exp(z.x), s, c;
    s = sin(z.y);
    c = cos(z.y);
    return cx<T>(e * c, e * s);
}

template <class T> struct eps_of;
template <> struct eps_of<float> { static constexpr float value = 1.1920929e-07f; static constexpr float safmin = 1.17549435e-38f; };
template <> struct eps_of<double> { static constexpr double value = 2.220446049250313e-16; static constexpr double safmin = 2.2250738585072014e-308; };

// ---- wave / block reductions -------------------------------------------------------------------------
template <class T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
template <class T> __device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { T w = __shfl_xor(v, o); v = w > v ? w : v; }
    return v;
}

// Value of lane `l` (WAVE-UNIFORM) in every lane: v_readlane_b32 per dword, a register-to-scalar move (no LDS round trip like __shfl's
// ds_bpermute).
template <class T> __device__ __forceinline__ T bcast_lane(T v, int l) {
    static_assert(sizeof(T) % 4 == 0, "dword multiple");
    int w[sizeof(T) / 4];
    __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
    for (int i = 0; i < (int)(sizeof(T) / 4); ++i) w[i] = __builtin_amdgcn_readlane(w[i], l);
    T r;
    __builtin_memcpy(&r, w, sizeof(T));
    return r;
}

// Ordering point inside a wave: lanes of one wavefront execute in lock-step, so this emits no instruction on gfx950; it
// stops the compiler from moving LDS accesses across it (and is a rendezvous in the CPU kernel-logic emulator).
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_wave_barrier(); }

static inline int cdiv_i(long a, long b) { return (int)((a + b - 1) / b); }

// Kernels that stage more than the default 64 KiB of dynamic LDS (gfx950 has 160 KiB per CU) must opt in.
static inline int set_max_dyn_smem(const void* kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return 0;
    return hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess;
}

// ---- host-side error helper ---------------------------------------------------------------------------
#define TRX_CHECK_LAUNCH()                                  \
    do {                                                    \
        hipError_t e__ = hipGetLastError();                 \
        if (e__ != hipSuccess) {                            \
            fprintf(stderr, "libtrx: HIP error '%s' at %s:%d\n", hipGetErrorString(e__), __FILE__, __LINE__); \
            return TRX_ERR_LAUNCH;                          \
        }                                                   \
    } while (0)

// Entry of every extern "C" function: clears a stale sticky error of the calling thread (e.g. hipErrorNotReady left behind by
// an event or stream query of the host framework) so that TRX_CHECK_LAUNCH reports only errors of this library's own calls.
inline hipStream_t api_stream(void* stream) {
    (void)hipGetLastError();
    return (hipStream_t)stream;
}

// ---- internal entry points (defined across the .hip files) -------------------------------------------
struct GemmDesc {            // optional per-batch override (device array), used by the eigensolver
    long offA, offB, offC;   // element offsets added to the batch base pointers
    int m, n, k;
    int pad;
};

template <class T>
int gemm(hipStream_t s, int opA, int opB, int m, int n, int k, cx<T> alpha, const cx<T>* A, int lda, long sA,
         const cx<T>* B, int ldb, long sB, cx<T> beta, cx<T>* C, int ldc, long sC, int batch,
         const GemmDesc* desc = nullptr, int b_upper = 0);     // b_upper: op(B) is upper triangular (k == rows of op(B) indexed like its columns)

template <class T>
int lu_factor(hipStream_t s, cx<T>* A, int lda, long sA, int n, int* piv, int batch, int* info);
int lu_set_knob(const char* key, int value);          // trx_tuning("lu_split", rows)
// large-tile fp64 kernel (gemm_big.hip)
void gemm_big_tile(int* bm, int* bn);
int gemm_big(hipStream_t s, int opA, int opB, int m, int n, int k, cx<double> alpha, const cx<double>* A, int lda, long sA, const cx<double>* B,
             int ldb, long sB, cx<double> beta, cx<double>* C, int ldc, long sC, int batch, int b_upper);
int gemm_set_knob(const char* key, int value);        // trx_tuning("gemm_big", 0 / 4)
template <class T>
int lu_solve(hipStream_t s, const cx<T>* LU, int lda, long sA, int n, const int* piv, cx<T>* B, int ldb, long sB,
             int nrhs, int batch);
// X = B A^-1 in place on B [nrows x n] from the factors of lu_factor; idx: n ints per matrix of scratch (lu.hip)
template <class T>
int lu_solve_right(hipStream_t s, const cx<T>* LU, int lda, long sA, int n, const int* piv, cx<T>* B, int ldb, long sB, int nrows, int batch, int* idx);

}  // namespace trx
