// Library-level entry points of include/tobac_flow_hip.h: version, error string, device count.
#include "tf_common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void tf_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int tf_version(void) { return 100; }
extern "C" const char *tf_last_error(void) { return g_err; }
extern "C" int tf_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { tf_set_error("hipGetDeviceCount: %s", hipGetErrorString(e)); return 0; }
    return n;
}

// ---- kernel timing ---------------------------------------------------------------------------------
#include <vector>
#include <mutex>
bool g_tf_prof_on = false;
namespace {
struct Rec { int id; double bytes; hipEvent_t a, b; };
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_free;
std::mutex g_mu;
const char *g_names[TFK_COUNT] = {"to8bit_pair", "fb_gaussian_blur", "fb_resize", "fb_polyexp", "fb_update_matrices",
    "fb_blur_solve", "fb_iteration_fused", "smooth_flow", "convolve", "sobel", "ws_setup", "ws_relax_sweep", "ws_labels", "vr_prepare", "vr_system", "vr_sor", "binary_morph"};
hipEvent_t get_event() {
    if (!g_free.empty()) { hipEvent_t e = g_free.back(); g_free.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}
}
void tf_prof_record(int id, double bytes, hipStream_t s, bool start) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (start) { Rec r{id, bytes, get_event(), get_event()}; (void)hipEventRecord(r.a, s); g_recs.push_back(r); }
    else {
        for (size_t i = g_recs.size(); i-- > 0;) if (g_recs[i].id == id) { (void)hipEventRecord(g_recs[i].b, s); break; }
    }
}
extern "C" int tf_profile_enable(int on) { g_tf_prof_on = on != 0; return TF_OK; }
extern "C" int tf_profile_kernel_count(void) { return TFK_COUNT; }
extern "C" const char *tf_profile_kernel_name(int id) { return (id >= 0 && id < TFK_COUNT) ? g_names[id] : ""; }
// fills calls[TFK_COUNT], ms[TFK_COUNT], bytes[TFK_COUNT]; clears the recorded events
extern "C" int tf_profile_collect(int64_t *calls, double *ms, double *bytes) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (int i = 0; i < TFK_COUNT; i++) { calls[i] = 0; ms[i] = 0; bytes[i] = 0; }
    for (auto &r : g_recs) {
        float t = 0;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            calls[r.id]++; ms[r.id] += t; bytes[r.id] += r.bytes;
        }
        g_free.push_back(r.a); g_free.push_back(r.b);
    }
    g_recs.clear();
    return TF_OK;
}

// The library keeps no device memory of its own (every workspace is the caller's); the pooled resources are the HIP
// events of the timing facility and the host scratch of the watershed's reference-order replays (csrc/watershed.hip).
// tf_shutdown() turns timing off and releases both; the library stays usable.
void tf_ws_host_pool_release();
extern "C" int tf_shutdown(void) {
    tf_ws_host_pool_release();
    std::lock_guard<std::mutex> lk(g_mu);
    g_tf_prof_on = false;
    for (auto &r : g_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : g_free) (void)hipEventDestroy(e);
    g_recs.clear(); g_free.clear();
    return TF_OK;
}
