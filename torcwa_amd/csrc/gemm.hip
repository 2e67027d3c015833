// Batched complex GEMM, row-major interleaved complex: C = alpha*op(A)*op(B) + beta*C, on the CDNA4 matrix cores.
// Replaces the torch.matmul call sites of the reference hot path (torcwa/rcwa.py:1161-1164, 1226-1232,
// 1236, 1260-1281, 1287-1304) and serves every block update of the LU and eigensolver kernels.
//
// Kernel: K-slab BK = 16 or 32, 256 threads = 4 waves arranged WR x (4/WR); a wave owns 16 rows x 16*NT columns (NT MFMA tiles, complex
// accumulators).  Three block tiles cover the shapes of the hot path:
//   <WR=4,NT=4>  64 x 64    the general case
//   <WR=4,NT=2>  64 x 32    outputs at most 32 columns wide (panel products Z V, A V of the Hessenberg reduction)
//   <WR=2,NT=4>  32 x 128   outputs at most 32 rows high   (V^H A of the Hessenberg reduction, the back-substitution blocks)
// so that a 32-wide panel no longer pays for a half-empty 64-wide tile.  Operands are staged through LDS as split re/im
// planes whose in-LDS orientation follows the operand's global orientation, so that both the global->LDS copy and the MFMA
// fragment reads are contiguous / bank-conflict free (see mfma.hpp); the next K-slab is prefetched into registers while
// the current one is multiplied.
#include "mfma.hpp"
#include <cstdlib>
#include <string>
#include "prof.hpp"

namespace trx {

namespace {
// K-slab depth.  32: with the 3M product a 16-deep slab is only 48 MFMAs per wave (~1.3 us), shorter than the latency of the
// register prefetch of the next slab under load; 32 doubles the distance and halves the barriers per flop (measured below).
// major-contiguous plane: element (major, k) at [k*ldm + major], ldm = 16 (mod 32) and >= the tile extent
constexpr int ldm_of(int extent) { return extent <= 64 ? 80 : 144; }
// k-contiguous plane: element (major, k) at [major*(BK+2) + k]
constexpr int plane_of(int extent, bool k_contig, int bk) { return k_contig ? extent * (bk + 2) : bk * ldm_of(extent); }

template <class T, int OPA, int OPB, int WR, int NT, int BK>
__global__ __launch_bounds__(256, 2) void gemm_mfma_kernel(int m, int n, int k, cx<T> alpha, const cx<T>* __restrict__ A, int lda, long sA,
                                                        const cx<T>* __restrict__ B, int ldb, long sB, cx<T> beta, cx<T>* __restrict__ C,
                                                        int ldc, long sC, const GemmDesc* __restrict__ desc, int b_upper) {
    constexpr int LDK = BK + 2;
    constexpr int WC = 4 / WR;                   // waves along N
    constexpr int BM = 16 * WR, BN = 16 * NT * WC;
    constexpr int LDMA = ldm_of(BM), LDMB = ldm_of(BN);
    constexpr int RA = BM * BK / 256, RB = BN * BK / 256;    // elements per thread and slab of the A / B tile
    __shared__ T Ar[plane_of(BM, OPA == TRX_OP_N, BK)];
    __shared__ T Ai[plane_of(BM, OPA == TRX_OP_N, BK)];
    __shared__ T Br[plane_of(BN, OPB != TRX_OP_N, BK)];
    __shared__ T Bi[plane_of(BN, OPB != TRX_OP_N, BK)];
    const int b = blockIdx.z;
    A += (long)b * sA;
    B += (long)b * sB;
    C += (long)b * sC;
    if (desc) {
        const GemmDesc d = desc[b];
        m = d.m; n = d.n; k = d.k;
        A += d.offA; B += d.offB; C += d.offC;
    }
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    if (m0 >= m || n0 >= n) return;
    if (b_upper && n0 + BN < k) k = n0 + BN;     // op(B) upper triangular: rows below the diagonal of this column tile are zero
    const int t = threadIdx.x;
    constexpr bool A_KC = (OPA == TRX_OP_N);     // A's k index is contiguous in global memory
    constexpr bool B_KC = (OPB != TRX_OP_N);     // B's k index is contiguous in global memory
    constexpr int sAr = A_KC ? LDK : 1, sAk = A_KC ? 1 : LDMA;
    constexpr int sBc = B_KC ? LDK : 1, sBk = B_KC ? 1 : LDMB;

    cx<T> ra[RA], rb[RB];
    // Branch-free tile loads: out-of-range coordinates are clamped to a valid address, so the global_load_dwordx4 of a slab issue
    // back-to-back instead of each sitting in its own exec branch.  The loaded registers are NOT touched here: the zeroing of the
    // out-of-range elements (and the conjugation) happens in store_tiles(), one K slab of MFMAs later -- a select right behind the load
    // made hipcc wait for the whole slab (s_waitcnt vmcnt(7..0)) BEFORE the MFMAs it was meant to run under, i.e. the "prefetch" exposed
    // the full load latency once per slab and wave (ISA of round 3).
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            const int e = t + 256 * r;
            const int row = A_KC ? (e / BK) : (e % BM), kk = A_KC ? (e % BK) : (e / BM);
            const int gr = (m0 + row < m) ? m0 + row : m - 1, gk = (k0 + kk < k) ? k0 + kk : k - 1;
            // 32-bit element offsets from the block-uniform base (scalar base + vector offset addressing: one register per load
            // instead of two); the host checks that a matrix spans less than 2^31 elements
            ra[r] = (OPA == TRX_OP_N) ? A[(unsigned)(gr * lda + gk)] : A[(unsigned)(gk * lda + gr)];
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int e = t + 256 * r;
            const int col = B_KC ? (e / BK) : (e % BN), kk = B_KC ? (e % BK) : (e / BN);
            const int gc = (n0 + col < n) ? n0 + col : n - 1, gk = (k0 + kk < k) ? k0 + kk : k - 1;
            rb[r] = (OPB == TRX_OP_N) ? B[(unsigned)(gk * ldb + gc)] : B[(unsigned)(gc * ldb + gk)];
        }
    };
    auto store_tiles = [&](int k0) {             // k0: the slab the registers were loaded for
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            const int e = t + 256 * r;
            const int row = A_KC ? (e / BK) : (e % BM), ka = A_KC ? (e % BK) : (e / BM);
            const bool ok = (m0 + row < m) && (k0 + ka < k);
            cx<T> v = ra[r];
            if (OPA == TRX_OP_C) v.y = -v.y;
            Ar[row * sAr + ka * sAk] = ok ? v.x : T(0); Ai[row * sAr + ka * sAk] = ok ? v.y : T(0);
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int e = t + 256 * r;
            const int col = B_KC ? (e / BK) : (e % BN), kb = B_KC ? (e % BK) : (e / BN);
            const bool ok = (n0 + col < n) && (k0 + kb < k);
            cx<T> v = rb[r];
            if (OPB == TRX_OP_C) v.y = -v.y;
            Br[col * sBc + kb * sBk] = ok ? v.x : T(0); Bi[col * sBc + kb * sBk] = ok ? v.y : T(0);
        }
    };

    // fp64: 3M complex product (P1 = Ar Br, P2 = Ai Bi, P3 = (Ar+Ai)(Br+Bi); see mfma.hpp); fp32: 4M (accR, accI; accX unused)
    constexpr bool M3 = sizeof(T) == 8;
    typename Mfma<T>::acc_t accR[NT], accI[NT], accX[M3 ? NT : 1];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) { accR[j][r] = T(0); accI[j][r] = T(0); if (M3) accX[M3 ? j : 0][r] = T(0); }

    const int wave = t >> 6, lane = t & 63;
    const int arow0 = 16 * (wave % WR), bcol0 = 16 * NT * (wave / WR);
    const bool has_beta = (beta.x != T(0)) || (beta.y != T(0));
    const int col_l = lane & 15;
    load_tiles(0);
    for (int k0 = 0; k0 < k; k0 += BK) {
        store_tiles(k0);
        __syncthreads();
        if (k0 + BK < k) load_tiles(k0 + BK);
        __builtin_amdgcn_sched_barrier(0);          // the next slab's loads are issued here, ahead of this slab's MFMAs, and first used after them
        // 16 k-values at a time: bounds the unrolled fragment prefetch (a 32-deep unroll spills)
#pragma unroll 1
        for (int kh = 0; kh < BK; kh += 16) {
            // a wave whose 16-row band lies entirely below the matrix (the last tile row of m = 1922 = 30 x 64 + 2 has two valid rows, of
            // m = 961 one) issues no MFMAs: wave-uniform, so no per-MFMA predicate; it still loads, stores and meets the barriers
            if (m0 + arow0 >= m) continue;
            if constexpr (M3) cmma3_tile_strided<T, NT>(Ar + kh * sAk, Ai + kh * sAk, sAr, sAk, arow0, Br + kh * sBk, Bi + kh * sBk, sBk, sBc, bcol0, 16, accR, accI, accX);
            else cmma_tile_strided<T, NT>(Ar + kh * sAk, Ai + kh * sAk, sAr, sAk, arow0, Br + kh * sBk, Bi + kh * sBk, sBk, sBc, bcol0, 16, accR, accI);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
    }
    if constexpr (M3) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const T p1 = accR[j][r], p2 = accI[j][r];
                accR[j][r] = p1 - p2;                       // Cr = Ar Br - Ai Bi
                accI[j][r] = accX[j][r] - p1 - p2;          // Ci = (Ar+Ai)(Br+Bi) - Ar Br - Ai Bi
            }
    }
    // C tile of this lane (beta != 0): all loads are issued back to back with clamped addresses (the guards sit at the
    // store), so a rank-32/64 update pays the read latency of C once instead of once per element behind an exec branch.
    cx<T> cv[4][NT];
    if (has_beta) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + arow0 + Mfma<T>::crow(lane, r);
            const int rc = row < m ? row : m - 1;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int col = n0 + bcol0 + 16 * j + col_l;
                cv[r][j] = C[(long)rc * ldc + (col < n ? col : n - 1)];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = m0 + arow0 + Mfma<T>::crow(lane, r);
        if (row >= m) continue;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int col = n0 + bcol0 + 16 * j + col_l;
            if (col >= n) continue;
            cx<T> v = alpha * cx<T>(accR[j][r], accI[j][r]);
            if (has_beta) v += beta * cv[r][j];
            C[(long)row * ldc + col] = v;
        }
    }
}

// (A 128 x 128 tile for large fp32 products -- same kernel shape, a wave owning 64 x 64, 249 registers -- was measured in round 6 and removed:
// gemm<N,N> fp32 stayed at 0.55 of the fp32 matrix peak on the refinement's LU / solve products, and the rank-32 / 64 updates of the Hessenberg
// reduction it also caught lost parallelism: that phase went from 909 to 982 ms.  profiles/r06_ab/r6i_ninth_call.txt)
// (An XCD-aware tile order -- XCD c takes a contiguous eighth of the tiles, column panels of 4 / 8 tiles inside a matrix, for both kernels --
// and the large tile on single full rows of tiles were measured in round 6 and removed: 35.85 / 35.88 layer-solves/s with the remap against
// 36.14 / 35.98 in the plain blockIdx order, every GEMM tag within 1 %; neither kernel is bound by its L2 misses at these shapes.
// profiles/r06_ab/r6m_xcd_tile_order.txt)
// Large-tile fp64 kernel of gemm_big.hip (128 x 96 on 8 waves): trx_tuning("gemm_big", v) / TRX_GEMM_BIG, v = 0 automatic (= on), 4 = off (64 x 64
// tile of this file everywhere).  Measured on MI355X at 1922^3 x 128 (profiles/r04_ab/r4_gemm_big.txt): 73.2 (off) / 85.2 TF-equivalent.
static int gemm_big_env() { const char* e = getenv("TRX_GEMM_BIG"); const int v = e ? atoi(e) : 0; return (v == 0 || v == 4) ? v : 0; }
static int g_gemm_big = gemm_big_env();

template <class T, int OPA, int OPB>
void launch_shape(hipStream_t s, int shape, int batch, int m, int n, int k, cx<T> alpha, const cx<T>* A, int lda, long sA,
                  const cx<T>* B, int ldb, long sB, cx<T> beta, cx<T>* C, int ldc, long sC, const GemmDesc* desc, int b_upper) {
    // K-slab depth 16 everywhere.  A 32-deep slab for the general tile (half the barriers per flop, twice the prefetch distance) was
    // measured SLOWER on MI355X with the 3M product (70.9 vs 72.6 TF at 1922^3 x 128: 256 VGPRs, one spill); the template keeps it.
    if (shape == 1)
        TRX_LAUNCH((gemm_mfma_kernel<T, OPA, OPB, 4, 2, 16>), dim3(cdiv_i(n, 32), cdiv_i(m, 64), batch), dim3(256), 0, s, m, n, k, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, desc, b_upper);
    else if (shape == 2)
        TRX_LAUNCH((gemm_mfma_kernel<T, OPA, OPB, 2, 4, 16>), dim3(cdiv_i(n, 128), cdiv_i(m, 32), batch), dim3(256), 0, s, m, n, k, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, desc, b_upper);
    else {
        TRX_LAUNCH((gemm_mfma_kernel<T, OPA, OPB, 4, 4, 16>), dim3(cdiv_i(n, 64), cdiv_i(m, 64), batch), dim3(256), 0, s, m, n, k, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, desc, b_upper);
    }
}

template <class T, int OPA>
int launch_b(hipStream_t s, int opB, int shape, int batch, int m, int n, int k, cx<T> alpha, const cx<T>* A, int lda, long sA,
             const cx<T>* B, int ldb, long sB, cx<T> beta, cx<T>* C, int ldc, long sC, const GemmDesc* desc, int b_upper) {
    switch (opB) {
        case TRX_OP_N: launch_shape<T, OPA, TRX_OP_N>(s, shape, batch, m, n, k, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, desc, b_upper); break;
        case TRX_OP_T: launch_shape<T, OPA, TRX_OP_T>(s, shape, batch, m, n, k, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, desc, b_upper); break;
        case TRX_OP_C: launch_shape<T, OPA, TRX_OP_C>(s, shape, batch, m, n, k, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, desc, b_upper); break;
        default: return TRX_ERR_ARG;
    }
    TRX_CHECK_LAUNCH();
    return TRX_OK;
}
}  // namespace

int gemm_set_knob(const char* key, int value) {
    if (std::string(key) == "gemm_big") {
        if (value != 0 && value != 4) return TRX_ERR_ARG;
        g_gemm_big = value;
        return TRX_OK;
    }
    return TRX_ERR_ARG;
}

template <class T>
int launch_dispatch(hipStream_t s, int opA, int opB, int shape, int batch, int m, int n, int k, cx<T> alpha, const cx<T>* A, int lda, long sA,
                    const cx<T>* B, int ldb, long sB, cx<T> beta, cx<T>* C, int ldc, long sC, const GemmDesc* desc = nullptr, int b_upper = 0);

template <class T>
int gemm(hipStream_t s, int opA, int opB, int m, int n, int k, cx<T> alpha, const cx<T>* A, int lda, long sA,
         const cx<T>* B, int ldb, long sB, cx<T> beta, cx<T>* C, int ldc, long sC, int batch, const GemmDesc* desc, int b_upper) {
    if (m <= 0 || n <= 0 || batch <= 0) return TRX_OK;
    if (k <= 0 && !desc) return TRX_ERR_ARG;          // callers never pass an empty inner dimension
    {   // operand tiles are addressed with 32-bit element offsets inside one matrix
        const long ra = (opA == TRX_OP_N ? m : k), rb = (opB == TRX_OP_N ? k : n);
        if (ra * lda >= 2147483647L || rb * ldb >= 2147483647L || (long)m * ldc >= 2147483647L) return TRX_ERR_UNSUPPORTED;
    }
    // block-tile shape: 64x32 for narrow outputs, 32x128 for flat ones (per-batch descriptors keep the general tile: their
    // sizes are only known on the device)
    const int shape = desc ? 0 : (n <= 32 ? 1 : (m <= 32 ? 2 : 0));
    // algorithmic work of this launch: 8 real flops per complex MAC; bytes = A + B read once, C written (+read if beta)
    const double macs = (double)m * n * k * batch * (b_upper ? 0.5 : 1.0);
    const double el = (double)sizeof(cx<T>) * batch;
    const bool nn = opA == TRX_OP_N && opB == TRX_OP_N;
    ProfScope prof(sizeof(T) == 8 ? (nn ? PROF_GEMM_NN : PROF_GEMM_OTHER) : (nn ? PROF_GEMM_NN_F32 : PROF_GEMM_OTHER_F32), s, desc ? 0.0 : 8.0 * macs,
                   desc ? 0.0 : el * ((double)m * k + (double)k * n + (double)m * n * ((beta.x != T(0) || beta.y != T(0)) ? 2 : 1)));
    if constexpr (sizeof(T) == 8) {
        // Large-tile kernel (gemm_big.hip) for large fp64 outputs.  A thin remainder (at most 32 rows / columns beyond a multiple of the
        // block tile: 1922 = 15 x 128 + 2 = 20 x 96 + 2) is peeled off for the flat / narrow tiles of this file instead of costing a
        // whole row / column of almost empty large tiles.
        int bm = 0, bn = 0;
        const int big = g_gemm_big != 4;
        if (big) gemm_big_tile(&bm, &bn);
        // (its direct loads address an operand with 32-bit BYTE offsets from a scalar base: the operand's extent must stay below 4 GiB)
        const long exA = (long)(opA == TRX_OP_N ? m : k) * lda, exB = (long)(opB == TRX_OP_N ? k : n) * ldb;
        if (big && !desc && !b_upper && m >= 2 * bm && n >= 2 * bn && k >= 64 && exA < (1L << 28) && exB < (1L << 28)) {
            const int rm = (m % bm) <= 32 ? m % bm : 0, rn = (n % bn) <= 32 ? n % bn : 0;
            const int mm = m - rm, nm = n - rn;
            int rc = gemm_big(s, opA, opB, mm, nm, k, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, batch, 0);
            if (rc != TRX_OK) return rc;
            if (rm) {      // bottom rows, all columns
                const cx<T>* Ar = A + (opA == TRX_OP_N ? (long)mm * lda : (long)mm);
                rc = launch_dispatch<T>(s, opA, opB, 2, batch, rm, n, k, alpha, Ar, lda, sA, B, ldb, sB, beta, C + (long)mm * ldc, ldc, sC);
                if (rc != TRX_OK) return rc;
            }
            if (rn) {      // right columns of the rows above
                const cx<T>* Bc = B + (opB == TRX_OP_N ? (long)nm : (long)nm * ldb);
                rc = launch_dispatch<T>(s, opA, opB, 1, batch, mm, rn, k, alpha, A, lda, sA, Bc, ldb, sB, beta, C + nm, ldc, sC);
                if (rc != TRX_OK) return rc;
            }
            return TRX_OK;
        }
    }
    return launch_dispatch<T>(s, opA, opB, shape, batch, m, n, k, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, desc, b_upper);
}

template <class T>
int launch_dispatch(hipStream_t s, int opA, int opB, int shape, int batch, int m, int n, int k, cx<T> alpha, const cx<T>* A, int lda, long sA,
                    const cx<T>* B, int ldb, long sB, cx<T> beta, cx<T>* C, int ldc, long sC, const GemmDesc* desc, int b_upper) {
    switch (opA) {
        case TRX_OP_N: return launch_b<T, TRX_OP_N>(s, opB, shape, batch, m, n, k, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, desc, b_upper);
        case TRX_OP_T: return launch_b<T, TRX_OP_T>(s, opB, shape, batch, m, n, k, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, desc, b_upper);
        case TRX_OP_C: return launch_b<T, TRX_OP_C>(s, opB, shape, batch, m, n, k, alpha, A, lda, sA, B, ldb, sB, beta, C, ldc, sC, desc, b_upper);
        default: return TRX_ERR_ARG;
    }
}

template int gemm<float>(hipStream_t, int, int, int, int, int, cx<float>, const cx<float>*, int, long, const cx<float>*, int, long, cx<float>, cx<float>*, int, long, int, const GemmDesc*, int);
template int gemm<double>(hipStream_t, int, int, int, int, int, cx<double>, const cx<double>*, int, long, const cx<double>*, int, long, cx<double>, cx<double>*, int, long, int, const GemmDesc*, int);

}  // namespace trx
