// Shared host/device helpers for the gfx950 kernels (internal; the public surface is
// include/tobac_flow_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/tobac_flow_hip.h"

void tf_set_error(const char *fmt, ...);

#define TF_CHECK_HIP(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            tf_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return TF_EHIP;                                                             \
        }                                                                               \
    } while (0)

#define TF_CHECK_LAUNCH()                                                               \
    do {                                                                                \
        hipError_t _e = hipGetLastError();                                              \
        if (_e != hipSuccess) {                                                         \
            tf_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return TF_EHIP;                                                             \
        }                                                                               \
    } while (0)

#define TF_REQUIRE(cond, msg)                                                           \
    do {                                                                                \
        if (!(cond)) { tf_set_error("%s", msg); return TF_EINVAL; }                     \
    } while (0)

static inline size_t tf_align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// bump allocator over the caller's workspace
struct TfArena {
    char *base; size_t size, used;
    TfArena(void *p, size_t n) : base((char *)p), size(n), used(0) {}
    template <typename T> T *take(size_t count) {
        size_t off = tf_align_up(used, 256);
        size_t end = off + count * sizeof(T);
        if (end > size || base == nullptr) { used = size + 1; return nullptr; }
        used = end;
        return (T *)(base + off);
    }
    bool ok() const { return used <= size; }
};

// ---- once-per-DEVICE initialisation (constant tables, function attributes are per-device state) --------------------
// tf_first_use_on_device(slot): true exactly once per (slot, current device), under a mutex: the caller performs its
// initialisation inside `if (guard.first) { ... guard.done(); }` while the lock is held, so a second thread or a second
// device never sees a half-initialised table (a process-global `static bool done` left the second device
// with an all-zero Lanczos table).
#include <mutex>
#define TF_MAX_DEVICES 64
struct TfDeviceOnce {
    std::mutex mu; bool done_[TF_MAX_DEVICES] = {};
    struct Guard {
        TfDeviceOnce &o; std::unique_lock<std::mutex> lk; int dev; bool first;
        explicit Guard(TfDeviceOnce &o_) : o(o_), lk(o_.mu), dev(0), first(false) {
            if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= TF_MAX_DEVICES) dev = 0;
            first = !o.done_[dev];
        }
        void done() { o.done_[dev] = true; }
    };
};

// ---- device helpers ------------------------------------------------------------------------
// cvRound(float): round half to even (the host reference uses cvtss2si under the default MXCSR)
__device__ __forceinline__ int tf_cvround(float v) { return __float2int_rn(v); }
__device__ __forceinline__ int tf_cvfloor(float v) { return __float2int_rd(v); }
__device__ __forceinline__ int tf_sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
__device__ __forceinline__ int tf_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---- built-in kernel timing (HIP events on the launch stream) ----------------------------------
// Used by bench.py for the roofline figure: tf_profile_enable(1) makes every annotated launch record
// a start/stop event pair together with its ALGORITHMIC byte count; tf_profile_get() synchronises and
// returns calls / total ms / total algorithmic bytes per kernel id.  Disabled: zero overhead.
enum TfKernelId {
    TFK_TO8BIT = 0, TFK_FB_BLUR, TFK_FB_RESIZE, TFK_FB_POLYEXP, TFK_FB_MATRICES, TFK_FB_BLUR_SOLVE, TFK_FB_ITER,
    TFK_SMOOTH, TFK_CONVOLVE, TFK_SOBEL, TFK_WS_SETUP, TFK_WS_RELAX, TFK_WS_LABELS, TFK_VR_PREPARE, TFK_VR_SYSTEM, TFK_VR_SOR,
    TFK_MORPH,
    TFK_COUNT
};
extern bool g_tf_prof_on;
void tf_prof_record(int id, double bytes, hipStream_t s, bool start);
struct TfProfScope {
    int id; double bytes; hipStream_t s; bool on;
    TfProfScope(int id_, double bytes_, hipStream_t s_) : id(id_), bytes(bytes_), s(s_), on(g_tf_prof_on) { if (on) tf_prof_record(id, bytes, s, true); }
    ~TfProfScope() { if (on) tf_prof_record(id, bytes, s, false); }
};
