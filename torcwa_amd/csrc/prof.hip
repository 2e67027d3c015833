#include "prof.hpp"

#include <mutex>
#include <vector>

namespace trx {
namespace {
struct TagData {
    std::vector<hipEvent_t> start, stop;
    int used = 0;          // event pairs recorded
    double launches = 0;   // all launches seen while enabled (also those beyond the pool)
    double flops = 0, bytes = 0;           // summed over ALL launches
    double flops_timed = 0, bytes_timed = 0;   // summed over the timed (event-bracketed) launches
};
TagData g_tags[PROF_NTAGS];
bool g_on = false;
std::mutex g_mu;
constexpr int POOL = 4096;
}  // namespace

bool prof_enabled() { return g_on; }

int prof_begin(int tag, hipStream_t s, double flops, double bytes) {
    std::lock_guard<std::mutex> lock(g_mu);
    TagData& t = g_tags[tag];
    t.launches += 1;
    t.flops += flops;
    t.bytes += bytes;
    if (t.used >= POOL) return -1;
    if ((int)t.start.size() <= t.used) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return -1;
        t.start.push_back(a);
        t.stop.push_back(b);
    }
    const int slot = t.used++;
    t.flops_timed += flops;
    t.bytes_timed += bytes;
    hipEventRecord(t.start[slot], s);
    return slot;
}

void prof_end(int tag, int slot, hipStream_t s) {
    std::lock_guard<std::mutex> lock(g_mu);
    hipEventRecord(g_tags[tag].stop[slot], s);
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
        t.launches = t.flops = t.bytes = t.flops_timed = t.bytes_timed = 0;
    }
    return TRX_OK;
}

// out[6] = {launches, timed_launches, flops_timed, bytes_timed, ms_timed, flops_all}
extern "C" int trx_prof_get(int tag, double* out) {
    if (tag < 0 || tag >= PROF_NTAGS || !out) return TRX_ERR_ARG;
    TagData& t = g_tags[tag];
    double ms = 0;
    for (int i = 0; i < t.used; ++i) {
        if (hipEventSynchronize(t.stop[i]) != hipSuccess) return TRX_ERR_LAUNCH;
        float e = 0;
        if (hipEventElapsedTime(&e, t.start[i], t.stop[i]) != hipSuccess) return TRX_ERR_LAUNCH;
        ms += e;
    }
    out[0] = t.launches; out[1] = t.used; out[2] = t.flops_timed; out[3] = t.bytes_timed; out[4] = ms; out[5] = t.flops;
    return TRX_OK;
}

extern "C" const char* trx_prof_tag_name(int tag) {
    static const char* names[PROF_NTAGS] = {"gemm_mfma_kernel<N,N>", "gemm_mfma_kernel<other ops>", "(retired tag)", "apply_window_kernel",
                                            "qr_window_kernel", "hess_gemv_kernel"};
    return (tag >= 0 && tag < PROF_NTAGS) ? names[tag] : "?";
}
