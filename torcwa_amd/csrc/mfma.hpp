// CDNA4 matrix-core tile primitives for complex GEMM-shaped work (f64 and f32 inputs, exact arithmetic).
//
// v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32: one wave computes a 16x16 tile, K = 4 per instruction.
// Operand maps (cdna_hip_programming.md section 3):  A[i = lane&15][k = lane>>4],  B[k = lane>>4][j = lane&15];
// result register r of a lane is C[row][col = lane&15] with row = (lane>>4) + 4r (f64) or 4(lane>>4) + r (f32).
// On MI355X the f64/f32 matrix rate equals the vector rate (78.6 / 157.3 TF), so the win over a VALU tile is not
// peak but operand economy: one LDS read per lane feeds 16x16x4 MACs, so the matrix pipe -- not LDS -- is the limit.
//
// A complex product is four real MFMAs per k-step:  Cr += Ar Br - Ai Bi,  Ci += Ar Bi + Ai Br.
// Operands are staged in LDS as SPLIT re/im planes, k-major:  plane[k][LD] with LD % 32 == 16 makes the b64
// fragment reads of a 32-lane group bank-conflict free (rows cover 32 banks, k toggles the other half).
#pragma once
#include "common.hpp"

namespace trx {

typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <class T> struct Mfma;
template <> struct Mfma<double> {
    typedef f64x4 acc_t;
    static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <> struct Mfma<float> {
    typedef f32x4 acc_t;
    static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) * 4 + r; }
};

// One wave accumulates a 16 x (16*NT) complex tile over `kcount` (multiple of 4) k-values.
//   A planes: element (row, k) at [row*sAr + k*sAk], rows arow0 .. arow0+15
//   B planes: element (k, col) at [k*sBk + col*sBc], cols bcol0 .. bcol0+16*NT-1
// Conflict-free strides for b64 reads: k-contiguous (s?k = 1) with a row/col stride = 2 (mod 32), e.g. 18 or 34;
// or row/col-contiguous with a k stride = 16 (mod 32), e.g. 80.
template <class T, int NT>
__device__ __forceinline__ void cmma_tile_strided(const T* __restrict__ Ar, const T* __restrict__ Ai, int sAr, int sAk, int arow0,
                                                  const T* __restrict__ Br, const T* __restrict__ Bi, int sBk, int sBc, int bcol0, int kcount,
                                                  typename Mfma<T>::acc_t (&accR)[NT], typename Mfma<T>::acc_t (&accI)[NT]) {
    const int lane = threadIdx.x & 63;
    const int lr = lane & 15, lk = lane >> 4;
    const int aoff = (arow0 + lr) * sAr + lk * sAk;
    const int boff = lk * sBk + (bcol0 + lr) * sBc;
    for (int k0 = 0; k0 < kcount; k0 += 4) {
        const T ar = Ar[aoff + k0 * sAk];
        const T ai = Ai[aoff + k0 * sAk];
        const T nai = -ai;
        T br[NT], bi[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            br[j] = Br[boff + k0 * sBk + 16 * j * sBc];
            bi[j] = Bi[boff + k0 * sBk + 16 * j * sBc];
        }
        // issue order keeps 2*NT independent MFMAs between two updates of the same accumulator
        // (no per-MFMA predicates here: conditional MFMAs make hipcc shuffle the whole accumulator file through
        //  v_accvgpr_mov and quintuple the kernel time -- measured)
#pragma unroll
        for (int j = 0; j < NT; ++j) accR[j] = Mfma<T>::mma(ar, br[j], accR[j]);
#pragma unroll
        for (int j = 0; j < NT; ++j) accI[j] = Mfma<T>::mma(ar, bi[j], accI[j]);
#pragma unroll
        for (int j = 0; j < NT; ++j) accR[j] = Mfma<T>::mma(nai, bi[j], accR[j]);
#pragma unroll
        for (int j = 0; j < NT; ++j) accI[j] = Mfma<T>::mma(ai, br[j], accI[j]);
    }
}

// Same tile with the 3M complex product (three real MFMAs per k-step instead of four):
//   P1 += Ar Br,  P2 += Ai Bi,  P3 += (Ar + Ai)(Br + Bi);      Cr = P1 - P2,  Ci = P3 - P1 - P2   (cmma3_finish)
// The operand sums are formed in registers (one VALU add per fragment, co-issued with the matrix pipe), so LDS traffic and
// layout are those of the 4M tile and the matrix-core work drops by a quarter.  Normwise the error bound is the 4M one
// (eps * (|Ar|+|Ai|)(|Br|+|Bi|) per term); only the RELATIVE accuracy of an imaginary part much smaller than its real part
// is weaker, which the S-matrix algebra never relies on.  Used for the fp64 path (the c128 parity gate, 1e-9, is held with it:
// tests/test_blocks.py, test_pipeline.py); the fp32 path keeps 4M, its budget being the tighter one.
template <class T, int NT>
__device__ __forceinline__ void cmma3_tile_strided(const T* __restrict__ Ar, const T* __restrict__ Ai, int sAr, int sAk, int arow0,
                                                   const T* __restrict__ Br, const T* __restrict__ Bi, int sBk, int sBc, int bcol0, int kcount,
                                                   typename Mfma<T>::acc_t (&p1)[NT], typename Mfma<T>::acc_t (&p2)[NT], typename Mfma<T>::acc_t (&p3)[NT]) {
    const int lane = threadIdx.x & 63;
    const int lr = lane & 15, lk = lane >> 4;
    const int aoff = (arow0 + lr) * sAr + lk * sAk;
    const int boff = lk * sBk + (bcol0 + lr) * sBc;
    for (int k0 = 0; k0 < kcount; k0 += 4) {
        const T ar = Ar[aoff + k0 * sAk];
        const T ai = Ai[aoff + k0 * sAk];
        const T as = ar + ai;
        T br[NT], bi[NT], bs[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            br[j] = Br[boff + k0 * sBk + 16 * j * sBc];
            bi[j] = Bi[boff + k0 * sBk + 16 * j * sBc];
            bs[j] = br[j] + bi[j];
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) p1[j] = Mfma<T>::mma(ar, br[j], p1[j]);
#pragma unroll
        for (int j = 0; j < NT; ++j) p2[j] = Mfma<T>::mma(ai, bi[j], p2[j]);
#pragma unroll
        for (int j = 0; j < NT; ++j) p3[j] = Mfma<T>::mma(as, bs[j], p3[j]);
    }
}

}  // namespace trx
