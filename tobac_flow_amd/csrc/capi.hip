// Library-level entry points of include/tobac_flow_hip.h: version, error string, device count.
#include "tf_common.h"
#include <stdarg.h>
#include <algorithm>

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

// ---- the practical HBM ceiling: a plain copy, 16 bytes per lane per access (bench.py's `practical_peak`) ---------------
// A workgroup copies tiles of UNROLL x 256 consecutive 16-byte words (all loads of a tile in flight before its first store);
// the grid is 8 workgroups per CU, grid-stride over the tiles (cdna_hip_programming.md, guideline 11).
template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256)
k_copy16(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16)
{
    const size_t tile = (size_t)UNROLL * 256, n_tiles = n16 / tile;
    for (size_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        const size_t base = t * tile + threadIdx.x;
        uint4 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; k++) {
            if (NT) { typedef unsigned int u4 __attribute__((ext_vector_type(4))); const u4 r = __builtin_nontemporal_load((const u4 *)(src + base + k * 256)); v[k] = make_uint4(r.x, r.y, r.z, r.w); }
            else v[k] = src[base + k * 256];
        }
#pragma unroll
        for (int k = 0; k < UNROLL; k++) {
            if (NT) { typedef unsigned int u4 __attribute__((ext_vector_type(4))); u4 r; r.x = v[k].x; r.y = v[k].y; r.z = v[k].z; r.w = v[k].w; __builtin_nontemporal_store(r, (u4 *)(dst + base + k * 256)); }
            else dst[base + k * 256] = v[k];
        }
    }
    for (size_t i = n_tiles * tile + (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
// variant: 0 = the default form; 1 .. : development forms (tools/copy_bw.py picks the default)
extern "C" int tf_copy16_variant(const void *src, void *dst, size_t bytes, void *stream, int variant)
{
    TF_REQUIRE(src && dst && bytes >= 16 && bytes % 16 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "tf_copy16: 16-byte aligned buffers of a multiple of 16 bytes");
    const size_t n16 = bytes / 16;
    hipStream_t s = (hipStream_t)stream;
    const uint4 *a = (const uint4 *)src; uint4 *b = (uint4 *)dst;
    const unsigned g8 = (unsigned)std::min<size_t>((n16 + 255) / 256, 256 * 8), g16 = (unsigned)std::min<size_t>((n16 + 255) / 256, 256 * 16);
    switch (variant) {
    case 1: hipLaunchKernelGGL((k_copy16<4, false>), dim3(g16), dim3(256), 0, s, a, b, n16); break;
    case 2: hipLaunchKernelGGL((k_copy16<8, false>), dim3(g8), dim3(256), 0, s, a, b, n16); break;
    case 3: hipLaunchKernelGGL((k_copy16<4, true>), dim3(g8), dim3(256), 0, s, a, b, n16); break;
    case 4: hipLaunchKernelGGL((k_copy16<8, true>), dim3(g8), dim3(256), 0, s, a, b, n16); break;
    case 5: hipLaunchKernelGGL((k_copy16<2, false>), dim3(g8), dim3(256), 0, s, a, b, n16); break;
    case 6: hipLaunchKernelGGL((k_copy16<1, false>), dim3(g16), dim3(256), 0, s, a, b, n16); break;
    case 7: hipLaunchKernelGGL((k_copy16<4, false>), dim3(g8), dim3(256), 0, s, a, b, n16); break;
    default: hipLaunchKernelGGL((k_copy16<4, true>), dim3(g8), dim3(256), 0, s, a, b, n16); break;      // nontemporal loads and stores: 5.9 TB/s over 2 x 1 GiB (plain: 5.6; torch copy_: 5.1; tools/copy_bw.py)
    }
    TF_CHECK_LAUNCH();
    return TF_OK;
}
extern "C" int tf_copy16(const void *src, void *dst, size_t bytes, void *stream) { return tf_copy16_variant(src, dst, bytes, stream, 0); }
