// Internal interfaces of the batched non-Hermitian eigensolver (eig_hess.hip, eig_qr.hip, eig_vec.hip, eig.hip).
#pragma once
#include "common.hpp"

namespace trx {

struct EigPlan {
    static constexpr int HNB = 32;     // Hessenberg panel width
    static constexpr int HG = 4;       // panels per group of the delayed right updates (eig_hess.hip)
    static constexpr int HGK = HG * HNB;
    static constexpr int QW = 64;      // QR window size (rows/cols staged in LDS)
    static constexpr int QNS = 16;     // shifts (= bulges) per CHAIN, spaced 2 rows apart
    static constexpr int QKC = 3;      // bulge chains per sweep (each in its own window, at least one window apart)
    static constexpr int QNMIN = 64;   // active blocks up to this size are finished by the in-LDS explicit-shift QR (one wave)
    static constexpr int QAED = 64;    // aggressive-early-deflation window (<= QNMIN, <= 64: one lane per column)
    static constexpr int VNB = 32;     // block height of the triangular eigenvector back-substitution
};

// per-matrix iteration state of the QR phase (device resident)
struct QrState {
    int ilo, ihi;                 // active block (inclusive)
    int nch;                      // bulge chains of the running sweep (chain 0 runs ahead, chain c follows chain c-1)
    int k[EigPlan::QKC];          // shifts of each chain
    int tau[EigPlan::QKC][2];     // chase step of each chain: bulge s of chain c sits at p = ilo + tau - 2 s.  Double-buffered by the
                                  // parity of the window step: a step reads [par] and writes [par ^ 1] (chain c looks at chain c-1)
    int tau_last[EigPlan::QKC];
    int mode;                     // see QR_* below
    int stall, sweeps;
    int w0[EigPlan::QKC], w1[EigPlan::QKC];   // [0]: window of the pending dense link (AED window / finished block), set by the prepare kernel
    int fail;                     // number of unconverged eigenvalues on failure
};
enum { QR_CHASE = 0, QR_SMALL_PENDING = 1, QR_SMALL_APPLIED = 2, QR_IDLE = 3, QR_DONE = 4, QR_AED_CHASE = 5 };
// One entry of a sweep's link log (slot = window step, per chain): the window [w0, w1) whose unitary the update kernels apply.
//   QRL_CHASE: unitary of a window step (U log, same slot): left update from column e on right behind the step, right / Z update deferred
//   QRL_DENSE: unitary of an AED window / a finished small block (per-matrix buffer U): all sides at once
struct QrLink { int w0, w1, kind, e; };
enum { QRL_NONE = 0, QRL_CHASE = 1, QRL_DENSE = 2 };
// chains per sweep of a batch and slots of the link log for order n (eig_qr.hip; the workspace layout depends on both)
int qr_chains_for(int batch);
int qr_log_slots(int n);

template <class T>
struct EigBuffers {
    cx<T>* A;      // [B,n,n] input, becomes H then T
    cx<T>* Z;      // [B,n,n] accumulated unitary
    cx<T>* X;      // [B,n,n] triangular eigenvectors
    cx<T>* Vp;     // [B,n,HNB] panel reflectors (dense, explicit zeros/ones)
    cx<T>* Yp;     // [B,n,HNB]
    cx<T>* Tp;     // [B,HNB,HNB]
    cx<T>* W1;     // [B,HNB*n]
    cx<T>* W2;     // [B,HNB*n]
    cx<T>* YV;     // [B,n,2*HNB]   [Y | V] operand of the fused trailing update
    cx<T>* BC;     // [B,2*HNB,n]   [Vt^H ; T^H W]
    cx<T>* Sm;     // [B,HNB,HNB]   V^H Y
    cx<T>* Vg;     // [B,n,HGK]     reflectors of a group of HG panels side by side (delayed right updates of Z and of the rows above the group)
    cx<T>* Wg;     // [B,n,HGK]     X Vg
    cx<T>* W2g;    // [B,n,HGK]     X Vg Tg
    cx<T>* Tg;     // [B,HGK,HGK]   merged triangular factor of the group
    cx<T>* Gg;     // [B,HGK,HGK]   Vg^H Vg
    cx<T>* tau;    // [B,HNB]
    cx<T>* tvec;   // [B,HNB]  V^H v of the current panel column
    cx<T>* U;      // [B,QW,QW] unitary of the last AED window / finished small block (dense link)
    cx<T>* Ulog;   // [B,slots,chains,QW,QW] window unitaries of the running sweep (link log)
    QrLink* links; // [B,slots,chains]
    cx<T>* shifts; // [B,QKC,QNS]
    T* bal_d;      // [B,n] balancing scale D (A_balanced = D^-1 A D)
    T* bal_w;      // [2,B,n] balancing scratch: row norms, column norms (sized for 3)
    int* bal_flags; // [B] per-matrix "needs balancing" flags (sized for 2B)
    QrState* st;   // [B]
    int* summary;  // [128]: 16 ints per iteration group of the QR phase (up to 8 groups)
    // mixed-precision route (fp64 only; null otherwise)
    char* mixed_pool;          // fp32 pool = Z | X | spill
    size_t mixed_pool_bytes;
    int *r_piv, *r_linfo, *r_flags, *r_partner, *r_clus, *r_edges, *r_ecount;
    T *r_eoff, *r_lmax, *r_scan;
    void* r_tab;
    cx<T>* r_d0;               // [B,n] Lambda + diag E
};

// buffers of the mixed-precision route (eig_refine.hip): fp32 eigendecomposition + Newton refinement to fp64
template <class T>
struct RefineBuffers {
    cx<T>* Rb;          // [B,n,n]  fp64: A V; once the residual has been taken from it: F
    cx<float>* LU32;    // [B,n,n]  LU factors of the fp32 start V0 (factored once, used by every step)
    cx<float>* E32;     // [B,n,n]  fp32(A V - V Lambda) -> E = V0^-1 (.); once F has been built from it: the product V F, one column block (fp64) at a time
    cx<T>* d0;          // [B,n]    Lambda + diag E of the step (the eigenvalue array itself is overwritten for cluster members)
    int* piv;           // [B,n]    pivots of LU32
    int* linfo;         // [B]      info of the fp32 eigensolve on entry, then of the LU
    int* flags;         // [B + 1]  per matrix: 1 = off-diagonal part not small / fp32 solve failed / V0 singular, 2 = a cluster the exact treatment does not take; [B] = count of flagged matrices, [B + 1] = clusters rotated in the running step
    T* eoff;            // [B] max off-diagonal |E_ij|
    T* lmax;            // [B] max |lambda_i|
    T* scan_part;       // [B, 2 * 32] partial maxima of the scan (its workgroups per matrix)
    int* partner;       // [B,n]    1 = index is coupled to another one
    int* clus;          // [B,n]    RCM * cluster + position, or -1
    int* edges;         // [B, REFINE_EDGE_CAP, 2] coupled pairs
    int* ecount;        // [B]
    void* tab;          // [B] cluster tables (RefineClusters<T>, REFINE_CLUSTER_BYTES each)
};
constexpr size_t REFINE_CLUSTER_BYTES = 1728 * 1024;
constexpr int REFINE_EDGE_CAP = 16384;
template <class T> int eig_refine(hipStream_t s, const RefineBuffers<T>& R, const cx<T>* A, const cx<float>* V32, const cx<float>* w32, cx<T>* w, cx<T>* V,
                                  int n, int batch, int steps, int* host_any, int* host_bad);
int refine_set_knob(const char* key, int value);
int refine_steps();

template <class T> size_t eig_ws_bytes_t(int n, int batch);
template <class T> void eig_carve(EigBuffers<T>& B, void* A, void* ws, int n, int batch);

template <class T> int balance(hipStream_t s, const EigBuffers<T>& B, int n, int batch);
template <class T> int hessenberg(hipStream_t s, const EigBuffers<T>& B, int n, int batch);
// Schur form: T in A, unitary accumulated into Z
template <class T> int hessenberg_qr(hipStream_t s, const EigBuffers<T>& B, int n, int batch, int* info);
int qr_set_knob(const char* key, int value);
int eig_set_knob(const char* key, int value);
int hess_set_knob(const char* key, int value);
template <class T> int schur_vectors(hipStream_t s, const EigBuffers<T>& B, int n, int batch, cx<T>* w, cx<T>* V);
// V <- D V (undo of the balancing) with unit 2-norm columns
template <class T> int finish_vectors(hipStream_t s, const EigBuffers<T>& B, int n, int batch, cx<T>* V);

}  // namespace trx
