// Error reporting and ABI version of libmssr_hip.so.
#include "common.hpp"
#include <atomic>

namespace mssr {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace mssr

// ---- launch timing for bench.py's roofline: HIP events around every GEMM launch, on its stream
namespace mssr {
struct ProfState {
    bool on = false;
    int cap = 0;
    std::atomic<int> n{0};
    hipEvent_t* ev = nullptr;          // 2 per record
    mssr_prof_record* rec = nullptr;
};
static ProfState g_prof;

static thread_local int tl_armed = -1;

int prof_begin(hipStream_t s) {
    ProfState& P = g_prof;
    if (!P.on) return -1;
    const int i = P.n.fetch_add(1);
    if (i >= P.cap) return -1;
    tl_armed = i;
    return i;
}
bool prof_active() { return g_prof.on; }
bool prof_take(hipEvent_t* start, hipEvent_t* stop) {
    if (tl_armed < 0) return false;
    ProfState& P = g_prof;
    const int i = tl_armed;
    tl_armed = -1;
    if (!P.on || i >= P.cap) return false;
    *start = P.ev[2 * i];
    *stop = P.ev[2 * i + 1];
    return true;
}
void prof_end(int i, hipStream_t s, int kernel, int in_dt, int out_dt, int tile, int64_t batch, int64_t m, int64_t n,
              int64_t k, int ksplit, int ln) {
    if (i < 0) return;
    ProfState& P = g_prof;
    tl_armed = -1;                  // a path that launched nothing profiled leaves ms = -1
    mssr_prof_record& r = P.rec[i];
    r.kernel = kernel; r.in_dtype = in_dt; r.out_dtype = out_dt; r.tile = tile;
    r.batch = batch; r.m = m; r.n = n; r.k = k; r.ms = -1.f; r.ksplit = ksplit; r.ln = ln;
}
}  // namespace mssr

extern "C" int mssr_prof_start(int max_records) {
    using namespace mssr;
    ProfState& P = g_prof;
    MSSR_REQUIRE(!P.on, "prof_start: already recording");
    MSSR_REQUIRE(max_records > 0 && max_records <= (1 << 20), "prof_start: bad record count");
    delete[] P.rec;                  // records of the previous start/stop cycle (kept for mssr_prof_get until now)
    P.ev = new hipEvent_t[2 * (size_t)max_records];
    P.rec = new mssr_prof_record[max_records];
    for (int i = 0; i < 2 * max_records; ++i) {
        if (hipEventCreate(&P.ev[i]) != hipSuccess) {
            for (int j = 0; j < i; ++j) (void)hipEventDestroy(P.ev[j]);
            delete[] P.ev; delete[] P.rec; P.ev = nullptr; P.rec = nullptr;
            set_error("prof_start: hipEventCreate failed");
            return MSSR_ERR_LAUNCH;
        }
    }
    P.cap = max_records; P.n = 0; P.on = true;
    return MSSR_OK;
}

// Stops recording, waits for the device and resolves the durations.  Returns the record count (< 0: error).
extern "C" int mssr_prof_stop(void) {
    using namespace mssr;
    ProfState& P = g_prof;
    if (!P.on) return 0;
    P.on = false;
    (void)hipDeviceSynchronize();
    if (P.n > P.cap) P.n = P.cap;
    for (int i = 0; i < P.n; ++i) {
        float ms = -1.f;
        if (hipEventElapsedTime(&ms, P.ev[2 * i], P.ev[2 * i + 1]) != hipSuccess) ms = -1.f;
        P.rec[i].ms = ms;
    }
    for (int i = 0; i < 2 * P.cap; ++i) (void)hipEventDestroy(P.ev[i]);
    delete[] P.ev; P.ev = nullptr;
    return P.n;
}

extern "C" int mssr_prof_get(int i, mssr_prof_record* out) {
    using namespace mssr;
    MSSR_REQUIRE(out && g_prof.rec && i >= 0 && i < g_prof.n, "prof_get: no record %d", i);
    *out = g_prof.rec[i];
    return MSSR_OK;
}

extern "C" const char* mssr_last_error(void) { return mssr::g_err; }
extern "C" int mssr_abi_version(void) { return 2; }
