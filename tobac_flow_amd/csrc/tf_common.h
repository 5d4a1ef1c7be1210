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

// ---- device helpers ------------------------------------------------------------------------
// cvRound(float): round half to even (the host reference uses cvtss2si under the default MXCSR)
__device__ __forceinline__ int tf_cvround(float v) { return __float2int_rn(v); }
__device__ __forceinline__ int tf_cvfloor(float v) { return __float2int_rd(v); }
__device__ __forceinline__ int tf_sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
__device__ __forceinline__ int tf_clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
