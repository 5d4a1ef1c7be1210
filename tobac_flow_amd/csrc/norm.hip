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

__global__ void __launch_bounds__(256)
k_minmax(const float *__restrict__ a, const float *__restrict__ b, int64_t n, unsigned *__restrict__ mm)
{
    float lo = INFINITY, hi = -INFINITY; unsigned seen = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n; i += stride) {
        float v = i < n ? a[i] : b[i - n];
        if (v == v) { lo = fminf(lo, v); hi = fmaxf(hi, v); seen = 1; }
    }
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o)); hi = fmaxf(hi, __shfl_xor(hi, o)); seen |= __shfl_xor(seen, o);
    }
    if ((threadIdx.x & 63) == 0 && seen) {
        atomicMin(&mm[0], f2key(lo)); atomicMax(&mm[1], f2key(hi)); atomicOr(&mm[2], 1u);
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
    int blocks = (int)((2 * n + 255) / 256); if (blocks > 2048) blocks = 2048;
    TfProfScope ps(TFK_TO8BIT, 18.0 * n, s);
    hipLaunchKernelGGL(k_minmax_init, dim3(1), dim3(1), 0, s, mm);
    hipLaunchKernelGGL(k_minmax, dim3(blocks), dim3(256), 0, s, frame0, frame1, n, mm);
    hipLaunchKernelGGL(k_to8bit, dim3(blocks), dim3(256), 0, s, frame0, frame1, n, mm, out0, out1);
    TF_CHECK_LAUNCH();
    return TF_OK;
}
