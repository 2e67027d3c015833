// hipemu: a minimal single-threaded, fiber-based executor for HIP kernels -- TEST INFRASTRUCTURE ONLY.
//
// Purpose: there is no GPU in the build container and GPU minutes are scarce, so kernel *logic*
// (index arithmetic, barriers, wave shuffles, MFMA fragment maps) is exercised on the CPU by compiling the
// unmodified product sources (torcwa_amd/csrc/*.hip) with clang++ against this header instead of the real
// <hip/hip_runtime.h>.  Each thread of a workgroup is a fiber; __syncthreads() and wave-collective ops are
// scheduler yields.  It is never built into or loaded by the product (torcwa_amd loads only libtrx.so, the
// gfx950 build) and it is not a compatibility layer: nothing in the product sources is conditional on it
// except the dynamic-LDS declaration macro TRX_DYN_SMEM.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define HIPEMU 1
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __restrict__ __restrict
#define TRX_DYN_SMEM(name) char* name = ::hipemu::g().dyn_smem

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;
typedef int hipError_t;
static const hipError_t hipSuccess = 0;
inline hipError_t hipGetLastError() { return 0; }
inline hipError_t hipPeekAtLastError() { return 0; }
inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return 0; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 16 };
inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 16; return 0; }      // a small "chip": 8 XCDs x 2 CUs
static const unsigned hipHostMallocDefault = 0;
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = malloc(n); return *p ? 0 : 2; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return 0; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memmove(d, s, n); return 0; }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
    for (size_t i = 0; i < h; ++i) memmove((char*)d + i * dp, (const char*)s + i * sp, w);
    return 0;
}

typedef void* hipEvent_t;
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return 0; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
static const hipError_t hipErrorNotReady = 600;
inline hipError_t hipEventQuery(hipEvent_t) { return 0; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }
// streams execute synchronously in the emulator: extra streams and cross-stream waits are no-ops
static const unsigned hipStreamNonBlocking = 1, hipEventDisableTiming = 2;
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return 0; }
inline hipError_t hipStreamDestroy(hipStream_t) { return 0; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = nullptr; return 0; }
inline hipError_t hipEventDestroy(hipEvent_t) { return 0; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return 0; }

namespace hipemu {

extern "C" void hipemu_switch(void** save_sp, void* new_sp);

enum State { READY = 0, AT_BARRIER = 1, AT_WAVE = 2, DONE = 3 };

struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    State st = DONE;
    // wave-op mailboxes (double-buffered by per-wave op parity)
    alignas(16) unsigned char box[2][32];
};

struct Global {
    dim3 threadIdx_, blockIdx_, blockDim_, gridDim_;
    char* dyn_smem = nullptr;
    std::vector<Fiber> fibers;
    void* sched_sp = nullptr;
    int cur = -1;
    int nthreads = 0;
    std::function<void()> body;
    std::vector<int> wave_parity;     // per wave
    size_t stack_bytes = 256 * 1024;
    long n_launch = 0;
};
inline Global& g() {
    static Global G;
    return G;
}

inline void yield(State s) {
    Global& G = g();
    Fiber& f = G.fibers[G.cur];
    f.st = s;
    hipemu_switch(&f.sp, G.sched_sp);
}

inline void trampoline() {
    Global& G = g();
    G.body();
    G.fibers[G.cur].st = DONE;
    hipemu_switch(&G.fibers[G.cur].sp, G.sched_sp);
    abort();
}

inline void set_ids(int t) {
    Global& G = g();
    unsigned bx = G.blockDim_.x, by = G.blockDim_.y;
    G.threadIdx_.x = t % bx;
    G.threadIdx_.y = (t / bx) % by;
    G.threadIdx_.z = t / (bx * by);
    G.cur = t;
}

inline void run_block() {
    Global& G = g();
    const int nt = G.nthreads;
    if ((int)G.fibers.size() < nt) {
        size_t old = G.fibers.size();
        G.fibers.resize(nt);
        for (size_t i = old; i < (size_t)nt; ++i) G.fibers[i].stack = (char*)aligned_alloc(64, G.stack_bytes);
    }
    for (int t = 0; t < nt; ++t) {
        Fiber& f = G.fibers[t];
        void** sp = (void**)(f.stack + G.stack_bytes);
        *(--sp) = nullptr;                 // fake return address: trampoline entry sees rsp % 16 == 8
        *(--sp) = (void*)&trampoline;
        for (int k = 0; k < 6; ++k) *(--sp) = nullptr;
        f.sp = sp;
        f.st = READY;
    }
    const int nw = (nt + 63) / 64;
    G.wave_parity.assign(nw, 0);
    for (;;) {
        bool any_ready = false;
        for (int t = 0; t < nt; ++t) {
            if (G.fibers[t].st == READY) {
                any_ready = true;
                set_ids(t);
                hipemu_switch(&G.sched_sp, G.fibers[t].sp);
            }
        }
        if (any_ready) continue;
        // wave ops: release a wave when none of its lanes is READY (they are all at the op, at a barrier or done)
        bool released = false;
        for (int w = 0; w < nw; ++w) {
            bool has = false;
            for (int l = w * 64; l < nt && l < (w + 1) * 64; ++l) has |= (G.fibers[l].st == AT_WAVE);
            if (has) {
                for (int l = w * 64; l < nt && l < (w + 1) * 64; ++l)
                    if (G.fibers[l].st == AT_WAVE) G.fibers[l].st = READY;
                G.wave_parity[w] ^= 1;
                released = true;
            }
        }
        if (released) continue;
        bool any_bar = false, all_done = true;
        for (int t = 0; t < nt; ++t) {
            any_bar |= (G.fibers[t].st == AT_BARRIER);
            all_done &= (G.fibers[t].st == DONE);
        }
        if (all_done) break;
        if (any_bar) {
            for (int t = 0; t < nt; ++t)
                if (G.fibers[t].st == AT_BARRIER) G.fibers[t].st = READY;
            continue;
        }
        fprintf(stderr, "hipemu: deadlock\n");
        abort();
    }
}

template <class F>
inline void launch(dim3 grid, dim3 block, size_t shmem, F&& f) {
    Global& G = g();
    G.n_launch++;
    G.gridDim_ = grid;
    G.blockDim_ = block;
    G.nthreads = block.x * block.y * block.z;
    G.body = std::function<void()>(f);
    std::vector<char> smem(shmem + 64);
    G.dyn_smem = (char*)(((uintptr_t)smem.data() + 63) & ~(uintptr_t)63);
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                G.blockIdx_ = dim3(bx, by, bz);
                run_block();
            }
}

// ---- wave collectives -------------------------------------------------------------------------------
inline int lane_id() { return g().cur & 63; }
inline int wave_base() { return g().cur & ~63; }

template <class T>
inline void wave_post(const T& v) {
    static_assert(sizeof(T) <= 32, "mailbox too small");
    Global& G = g();
    int par = G.wave_parity[G.cur >> 6];
    memcpy(G.fibers[G.cur].box[par], &v, sizeof(T));
    yield(AT_WAVE);
}
template <class T>
inline T wave_read(int lane) {   // valid right after wave_post (parity was flipped on release)
    Global& G = g();
    int par = G.wave_parity[G.cur >> 6] ^ 1;
    T v;
    int idx = wave_base() + lane;
    if (idx >= G.nthreads) idx = G.cur;
    memcpy(&v, G.fibers[idx].box[par], sizeof(T));
    return v;
}
}  // namespace hipemu

#define threadIdx (::hipemu::g().threadIdx_)
#define blockIdx (::hipemu::g().blockIdx_)
#define blockDim (::hipemu::g().blockDim_)
#define gridDim (::hipemu::g().gridDim_)
static const int warpSize = 64;

inline void __syncthreads() { ::hipemu::yield(::hipemu::AT_BARRIER); }
inline void __threadfence() {}
inline void __threadfence_block() {}
inline void __builtin_amdgcn_s_barrier() { ::hipemu::yield(::hipemu::AT_BARRIER); }

template <class T>
inline T __shfl(T v, int src, int width = 64) {
    ::hipemu::wave_post(v);
    int l = ::hipemu::lane_id();
    int s = (l & ~(width - 1)) + (src & (width - 1));
    return ::hipemu::wave_read<T>(s);
}
template <class T>
inline T __shfl_xor(T v, int mask, int width = 64) {
    ::hipemu::wave_post(v);
    int l = ::hipemu::lane_id();
    int s = l ^ mask;
    if ((s & ~(width - 1)) != (l & ~(width - 1))) s = l;
    return ::hipemu::wave_read<T>(s);
}
template <class T>
inline T __shfl_down(T v, unsigned d, int width = 64) {
    ::hipemu::wave_post(v);
    int l = ::hipemu::lane_id();
    int s = l + (int)d;
    if ((s & ~(width - 1)) != (l & ~(width - 1))) s = l;
    return ::hipemu::wave_read<T>(s);
}
template <class T>
inline T __shfl_up(T v, unsigned d, int width = 64) {
    ::hipemu::wave_post(v);
    int l = ::hipemu::lane_id();
    int s = l - (int)d;
    if (s < 0 || (s & ~(width - 1)) != (l & ~(width - 1))) s = l;
    return ::hipemu::wave_read<T>(s);
}
inline void __builtin_amdgcn_wave_barrier() { ::hipemu::wave_post<int>(0); }
inline void __builtin_amdgcn_sched_barrier(int) {}
inline long long clock64() { return 0; }               // cycle counter: only used by optional debug timers
inline long long wall_clock64() { return 0; }      // compiler scheduling fence: nothing to emulate
inline unsigned long long __ballot(int pred) {
    ::hipemu::wave_post<int>(pred ? 1 : 2);          // 2 = participated, false; 0 = stale/not participating
    unsigned long long m = 0;
    // only lanes that posted this round hold fresh data; lanes that did not participate keep old values, so a
    // kernel must call __ballot convergently (same rule as on hardware for meaningful results)
    for (int l = 0; l < 64; ++l)
        if (::hipemu::wave_base() + l < ::hipemu::g().nthreads && ::hipemu::wave_read<int>(l) == 1) m |= 1ull << l;
    return m;
}
inline int __all(int pred) {
    ::hipemu::wave_post<int>(pred ? 1 : 2);
    for (int l = 0; l < 64; ++l)
        if (::hipemu::wave_base() + l < ::hipemu::g().nthreads && ::hipemu::wave_read<int>(l) == 2) return 0;
    return 1;
}
inline int __any(int pred) { return __ballot(pred) != 0; }
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
template <class T>
inline T __builtin_amdgcn_readfirstlane(T v) { return __shfl(v, 0); }
inline int __builtin_amdgcn_readlane(int v, int l) { return __shfl(v, l); }

// ---- MFMA (fragment maps per /opt/skills/guides/cdna_hip_programming.md section 3) --------------------
typedef double hipemu_f64x4 __attribute__((ext_vector_type(4)));
typedef float hipemu_f32x4 __attribute__((ext_vector_type(4)));
typedef float hipemu_f32x16 __attribute__((ext_vector_type(16)));

// v_mfma_f64_16x16x4_f64: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D: col=l&15, row=(l>>4)+4*r
inline hipemu_f64x4 __builtin_amdgcn_mfma_f64_16x16x4f64(double a, double b, hipemu_f64x4 c, int, int, int) {
    struct AB { double a, b; } ab{a, b};
    ::hipemu::wave_post(ab);
    int l = ::hipemu::lane_id();
    hipemu_f64x4 d = c;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) + 4 * r, col = l & 15;
        double acc = c[r];
        for (int k = 0; k < 4; ++k) {
            double av = ::hipemu::wave_read<AB>(row + 16 * k).a;
            double bv = ::hipemu::wave_read<AB>(col + 16 * k).b;
            acc = std::fma(av, bv, acc);
        }
        d[r] = acc;
    }
    return d;
}
// v_mfma_f32_16x16x4_f32: same A/B maps; D: col=l&15, row=(l>>4)*4+r
inline hipemu_f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, hipemu_f32x4 c, int, int, int) {
    struct AB { float a, b; } ab{a, b};
    ::hipemu::wave_post(ab);
    int l = ::hipemu::lane_id();
    hipemu_f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) * 4 + r, col = l & 15;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) {
            float av = ::hipemu::wave_read<AB>(row + 16 * k).a;
            float bv = ::hipemu::wave_read<AB>(col + 16 * k).b;
            acc = std::fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    return d;
}
// v_mfma_f32_32x32x2_f32: A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D: col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5)
inline hipemu_f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, hipemu_f32x16 c, int, int, int) {
    struct AB { float a, b; } ab{a, b};
    ::hipemu::wave_post(ab);
    int l = ::hipemu::lane_id();
    hipemu_f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), col = l & 31;
        float acc = c[r];
        for (int k = 0; k < 2; ++k) {
            float av = ::hipemu::wave_read<AB>(row + 32 * k).a;
            float bv = ::hipemu::wave_read<AB>(col + 32 * k).b;
            acc = std::fmaf(av, bv, acc);
        }
        d[r] = acc;
    }
    return d;
}

// ---- atomics (single OS thread) ----------------------------------------------------------------------
template <class T>
inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T>
inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T>
inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T>
inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <class T>
inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }

// ---- device math -------------------------------------------------------------------------------------
inline void sincos(double x, double* s, double* c) { *s = std::sin(x); *c = std::cos(x); }
inline void sincosf(float x, float* s, float* c) { *s = std::sin(x); *c = std::cos(x); }
inline void sincospi(double x, double* s, double* c) { *s = std::sin(M_PI * x); *c = std::cos(M_PI * x); }
inline double __builtin_amdgcn_rcp(double x) { return 1.0 / x; }
inline double __builtin_amdgcn_rsq(double x) { return 1.0 / std::sqrt(x); }
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
inline double rsqrt(double x) { return 1.0 / std::sqrt(x); }
inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }

// direct global -> LDS load (torcwa_amd/csrc/common.hpp): the emulator copies at once, so the wait is empty
#define TRX_LDS_DMA16(gsrc_lane, lds_wave_base) std::memcpy((char*)(lds_wave_base) + 16 * (threadIdx.x & 63), (const void*)(gsrc_lane), 16)
#define TRX_LDS_DMA16_S(sbase, voff32, lds_wave_base) std::memcpy((char*)(lds_wave_base) + 16 * (threadIdx.x & 63), (const char*)(sbase) + (unsigned)(voff32), 16)
#define TRX_WAIT_VMCNT(n) ((void)0)
#define TRX_WAIT_VMCNT_IMM(expr) ((void)0)

#define HIP_KERNEL_NAME(...) __VA_ARGS__
namespace hipemu {
template <class K, class... A>
inline void launch_k(dim3 grid, dim3 block, size_t shmem, K kernel, A... args) {
    launch(grid, block, shmem, [=]() mutable { kernel(args...); });
}
}  // namespace hipemu
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    ::hipemu::launch_k(dim3(grid), dim3(block), (size_t)(shmem), kernel, __VA_ARGS__)
using std::copysign;
using std::cos;
using std::exp;
using std::fabs;
using std::fma;
using std::hypot;
using std::sin;
using std::sqrt;
