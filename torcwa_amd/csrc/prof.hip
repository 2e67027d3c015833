#include "prof.hpp"

#include <mutex>
#include <vector>

namespace trx {
namespace {
constexpr int POOL = 2048;
struct TagData {
    std::vector<hipEvent_t> start, stop;
    std::vector<double> sflops, sbytes;       // algorithmic work of each sampled launch
    int used = 0;          // event pairs in use
    int open = 0;          // scopes that hold a slot of this tag and have not recorded their stop event yet
    long stride = 1;       // every stride-th launch is sampled (starts at 16 for the two kernels launched ~18000 times per eig call:
                           // an event pair around EVERY such launch cost 2 % of the step time, measured)
    double launches = 0;   // all launches seen while enabled
    double flops = 0, bytes = 0;           // summed over ALL launches
};
TagData g_tags[PROF_NTAGS];
bool g_on = false;
std::mutex g_mu;
}  // namespace

bool prof_enabled() { return g_on; }

int prof_begin(int tag, hipStream_t s, double flops, double bytes) {
    std::lock_guard<std::mutex> lock(g_mu);
    TagData& t = g_tags[tag];
    const long idx = (long)t.launches;
    t.launches += 1;
    t.flops += flops;
    t.bytes += bytes;
    if (idx % t.stride != 0) return -1;
    if (t.used >= POOL) {
        // pool full: keep the samples of the even multiples of the stride (slots 0, 2, 4, ...), double the stride.  The compaction moves
        // event pairs to other slots, so it waits until no scope (of another host thread) still holds a slot handle of this tag.
        if (t.open > 0) return -1;
        for (int i = 0; 2 * i < t.used; ++i) {
            std::swap(t.start[i], t.start[2 * i]);
            std::swap(t.stop[i], t.stop[2 * i]);
            t.sflops[i] = t.sflops[2 * i];
            t.sbytes[i] = t.sbytes[2 * i];
        }
        t.used = (t.used + 1) / 2;
        t.stride *= 2;
        if (idx % t.stride != 0) return -1;
    }
    if ((int)t.start.size() <= t.used) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
        t.start.push_back(a);
        t.stop.push_back(b);
        t.sflops.push_back(0);
        t.sbytes.push_back(0);
    }
    const int slot = t.used++;
    t.sflops[slot] = flops;
    t.sbytes[slot] = bytes;
    hipEventRecord(t.start[slot], s);
    t.open += 1;
    return slot;
}

void prof_end(int tag, int slot, hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_mu);
    hipEventRecord(g_tags[tag].stop[slot], s);
    g_tags[tag].open -= 1;
}

void prof_add_work(int tag, double flops, double bytes) {
    std::lock_guard<std::mutex> lock(g_mu);
    g_tags[tag].flops += flops;
    g_tags[tag].bytes += bytes;
}
}  // namespace trx

using namespace trx;

extern "C" int trx_prof_enable(int on) {
    g_on = on != 0;
    return TRX_OK;
}

extern "C" int trx_prof_reset(void) {
    for (int i = 0; i < PROF_NTAGS; ++i) {
        TagData& t = g_tags[i];
        t.used = 0;
        t.open = 0;
        t.stride = (i == PROF_QR_WINDOW || i == PROF_QR_APPLY_LEFT) ? 16 : 1;
        t.launches = t.flops = t.bytes = 0;
    }
    return TRX_OK;
}

// out[7] = {launches, timed_launches, flops_timed, bytes_timed, ms_timed, flops_all, bytes_all}
extern "C" int trx_prof_get(int tag, double* out) {
    if (tag < 0 || tag >= PROF_NTAGS || !out) return TRX_ERR_ARG;
    TagData& t = g_tags[tag];
    double ms = 0, fl = 0, by = 0;
    for (int i = 0; i < t.used; ++i) {
        if (hipEventSynchronize(t.stop[i]) != hipSuccess) return TRX_ERR_LAUNCH;
        float e = 0;
        if (hipEventElapsedTime(&e, t.start[i], t.stop[i]) != hipSuccess) return TRX_ERR_LAUNCH;
        ms += e;
        fl += t.sflops[i];
        by += t.sbytes[i];
    }
    out[0] = t.launches; out[1] = t.used; out[2] = fl; out[3] = by; out[4] = ms; out[5] = t.flops; out[6] = t.bytes;
    return TRX_OK;
}

extern "C" const char* trx_prof_tag_name(int tag) {
    static const char* names[PROF_NTAGS] = {"gemm<N,N>", "gemm<other ops>", "qr_prepare_kernel", "apply_links_kernel<1>",
                                            "qr_window_kernel", "hess_gemv_kernel", "hess_col_kernel", "lu_panel_kernel", "apply_links_kernel<0>",
                                            "gemm<N,N> fp32", "gemm<other ops> fp32",
                                            "phase:balance", "phase:hessenberg", "phase:qr", "phase:schur_vectors", "phase:refinement"};
    return (tag >= 0 && tag < PROF_NTAGS) ? names[tag] : "?";
}
