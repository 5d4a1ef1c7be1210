// to_8bit(linear_norm(data[i:i+2]), 0, 1) for one frame pair, exactly as
// /root/reference/tobac_flow/flow.py:411-414 composes
// /root/reference/tobac_flow/utils/normalisation_utils.py:59-72 (linear_norm) and :10-33 (to_8bit).
// Pass 1: joint NaN-ignoring min/max of both frames (grid-stride, float4 loads, wave shuffle +
// one atomic pair per block on order-preserving integer keys).  Pass 2: map.  Algorithmic
// traffic: 2 x (4 r + 4 r + 1 w) = 18 B per pixel pair.
#include "tf_common.h"

__device__ __forceinline__ unsigned f2key(float v) {       // monotone float -> uint
    unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

__global__ void k_minmax_init(unsigned *mm) { mm[0] = 0xFFFFFFFFu; mm[1] = 0u; mm[2] = 0u; }

__device__ __forceinline__ void mm_take(float v, float &lo, float &hi, unsigned &seen) {
    if (v == v) { lo = fminf(lo, v); hi = fmaxf(hi, v); seen = 1; }
}

// NaN-ignoring min / max of one frame: scalar head up to 16-byte alignment, float4 body with two loads in flight per
// thread, scalar tail.  min / max are exact and order-independent, so any traversal gives the same pair.
__device__ __forceinline__ void mm_span(const float *__restrict__ p, int64_t n, int64_t tid, int64_t nthreads,
                                        float &lo, float &hi, unsigned &seen)
{
    int64_t head = (int64_t)((16 - ((uintptr_t)p & 15)) & 15) / 4;
    if (head > n) head = n;
    const int64_t nvec = (n - head) / 4, tail0 = head + nvec * 4;
    if (tid < head) mm_take(p[tid], lo, hi, seen);
    if (tid < n - tail0) mm_take(p[tail0 + tid], lo, hi, seen);
    const float4 *v = (const float4 *)(p + head);
    int64_t i = tid;
    for (; i + nthreads < nvec; i += 2 * nthreads) {
        const float4 x = v[i], y = v[i + nthreads];
        mm_take(x.x, lo, hi, seen); mm_take(x.y, lo, hi, seen); mm_take(x.z, lo, hi, seen); mm_take(x.w, lo, hi, seen);
        mm_take(y.x, lo, hi, seen); mm_take(y.y, lo, hi, seen); mm_take(y.z, lo, hi, seen); mm_take(y.w, lo, hi, seen);
    }
    if (i < nvec) {
        const float4 x = v[i];
        mm_take(x.x, lo, hi, seen); mm_take(x.y, lo, hi, seen); mm_take(x.z, lo, hi, seen); mm_take(x.w, lo, hi, seen);
    }
}

__global__ void __launch_bounds__(256)
k_minmax(const float *__restrict__ a, const float *__restrict__ b, int64_t n, unsigned *__restrict__ mm)
{
    float lo = INFINITY, hi = -INFINITY; unsigned seen = 0;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (int64_t)gridDim.x * blockDim.x;
    mm_span(a, n, tid, nthreads, lo, hi, seen);
    mm_span(b, n, tid, nthreads, lo, hi, seen);
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); seen |= __shfl_xor(seen, o);
    }
    // one result per workgroup, and an atomic only when it can still change the global value: thousands of
    // atomics on the three words of one cache line used to cost more than reading the two frames
    __shared__ float s_lo[4], s_hi[4]; __shared__ unsigned s_seen[4];
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_lo[w] = lo; s_hi[w] = hi; s_seen[w] = seen; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; k++) { lo = fminf(lo, s_lo[k]); hi = fmaxf(hi, s_hi[k]); seen |= s_seen[k]; }
        if (seen) {
            const unsigned klo = f2key(lo), khi = f2key(hi);
            if (klo < __hip_atomic_load(&mm[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMin(&mm[0], klo);
            if (khi > __hip_atomic_load(&mm[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&mm[1], khi);
            if (__hip_atomic_load(&mm[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) atomicOr(&mm[2], 1u);
        }
    }
}

__device__ __forceinline__ float norm255(float x, float vmin, float factor) {
    // linear_norm: (x - vmin) * factor, np.minimum(.,1), np.maximum(.,0) (NaN propagates);
    // to_8bit(., 0, 1): (t - 0) * 255.0 in float32
    float t = (x - vmin) * factor;
    t = (t != t) ? t : fminf(t, 1.f);
    t = (t != t) ? t : fmaxf(t, 0.f);
    return (t - 0.f) * 255.f;
}

__global__ void __launch_bounds__(256)
k_to8bit(const float *__restrict__ a, const float *__restrict__ b, int64_t n, const unsigned *__restrict__ mm,
         uint8_t *__restrict__ oa, uint8_t *__restrict__ ob)
{
    // nanmin/nanmax of an all-NaN pair are NaN -> `vmax > vmin` is False -> factor 0
    float vmin = mm[2] ? key2f(mm[0]) : NAN, vmax = mm[2] ? key2f(mm[1]) : NAN;
    float factor = (vmax > vmin) ? 1.f / (vmax - vmin) : 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float u0 = norm255(a[i], vmin, factor), u1 = norm255(b[i], vmin, factor);
        bool f0 = isfinite(u0), f1 = isfinite(u1);
        if (!f0) u0 = 127.f;
        if (!f1) u1 = 127.f;
        if (!f0) u0 = u1;            // out[0][~fin0] = out[1][~fin0]
        if (!f1) u1 = u0;            // out[1][~fin1] = out[0][~fin1]
        oa[i] = (uint8_t)(int)u0;    // astype("uint8") truncates; values are within [0, 255]
        ob[i] = (uint8_t)(int)u1;
    }
}

extern "C" size_t tf_to8bit_workspace_bytes(int64_t, int64_t) { return 256; }

extern "C" int tf_to8bit_pair(const float *frame0, const float *frame1, int64_t H, int64_t W,
                              uint8_t *out0, uint8_t *out1, void *ws, size_t ws_bytes, void *stream)
{
    TF_REQUIRE(frame0 && frame1 && out0 && out1 && ws, "tf_to8bit_pair: null pointer");
    TF_REQUIRE(H > 0 && W > 0, "tf_to8bit_pair: bad shape");
    if (ws_bytes < 256) { tf_set_error("tf_to8bit_pair: workspace too small"); return TF_ENOMEM; }
    hipStream_t s = (hipStream_t)stream;
    unsigned *mm = (unsigned *)ws;
    const int64_t n = H * W;
    int blocks = (int)((n + 255) / 256); if (blocks > 2048) blocks = 2048;
    TfProfScope ps(TFK_TO8BIT, 18.0 * n, s);
    hipLaunchKernelGGL(k_minmax_init, dim3(1), dim3(1), 0, s, mm);
    hipLaunchKernelGGL(k_minmax, dim3(blocks), dim3(256), 0, s, frame0, frame1, n, mm);
    hipLaunchKernelGGL(k_to8bit, dim3(blocks), dim3(256), 0, s, frame0, frame1, n, mm, out0, out1);
    TF_CHECK_LAUNCH();
    return TF_OK;
}
