// Optional in-library kernel timing with HIP events (used by bench.py for the live roofline figure).
// When enabled, instrumented launch sites bracket the kernel with hipEventRecord on the SAME stream the kernel is
// launched on; trx_prof_get() synchronises those events and returns launches / algorithmic flops / bytes / time.
#pragma once
#include "common.hpp"

namespace trx {
enum ProfTag { PROF_GEMM_NN = 0, PROF_GEMM_OTHER = 1, PROF_QR_PREPARE = 2, PROF_QR_APPLY_RIGHT = 3, PROF_QR_WINDOW = 4,
               PROF_HESS_GEMV = 5, PROF_HESS_COL = 6, PROF_LU_PANEL = 7, PROF_QR_APPLY_LEFT = 8, PROF_GEMM_NN_F32 = 9, PROF_GEMM_OTHER_F32 = 10,
               // wall-clock PHASES of trx_eig (one event pair per call on the caller's stream; the QR phase from fork to join of its iteration groups)
               PROF_PH_BALANCE = 11, PROF_PH_HESSENBERG = 12, PROF_PH_QR = 13, PROF_PH_VECTORS = 14, PROF_PH_REFINE = 15, PROF_NTAGS = 16 };      // the fp32 GEMMs (first stage of the
               // mixed-precision eigensolver, precision="native") are counted apart from the fp64 ones: other peak, other roofline

bool prof_enabled();
// returns an event-slot handle (>= 0) or -1 when this launch is not sampled; records the start event.  Sampling is systematic:
// every `stride`-th launch of a tag is timed; when the pool of event pairs is full, every other sample is dropped and the
// stride doubles, so the timed launches stay spread uniformly over the whole run whatever its length.
int prof_begin(int tag, hipStream_t s, double flops, double bytes);
void prof_end(int tag, int slot, hipStream_t s);
// adds algorithmic work that is only known after the launches ran (data-dependent kernels count it on the device)
void prof_add_work(int tag, double flops, double bytes);

struct ProfScope {
    int tag, slot;
    hipStream_t s;
    ProfScope(int tag_, hipStream_t s_, double flops, double bytes) : tag(tag_), slot(-1), s(s_) {
        if (prof_enabled()) slot = prof_begin(tag, s, flops, bytes);
    }
    ~ProfScope() {
        if (slot >= 0) prof_end(tag, slot, s);
    }
};
}  // namespace trx
