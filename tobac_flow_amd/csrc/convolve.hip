// Semi-Lagrangian gather operators for gfx950:
//   tf_convolve  -- /root/reference/tobac_flow/convolve.py:248-348 (+ :8-86, :89-144, :147-245)
//                   with the reductions the reference passes as `func` fused in:
//                   sobel.py:32-86, flow.py:180-184 (diff), detection.py nanmean / any
//   tf_warp_flow -- /root/reference/tobac_flow/utils/flow_utils.py:80-99
//   tf_smooth_flow_step -- /root/reference/tobac_flow/flow.py:530-568
//   tf_flow_finalize    -- /root/reference/tobac_flow/flow.py:425-426, :60-61
//
// One thread per output pixel; a 64 x 4 workgroup covers 64 consecutive columns of 4 rows so
// that the field loads of the 3x3 same-step taps and of the (flow-displaced, locally coherent)
// t-1 / t+1 taps coalesce and hit L1/L2.  The 27-row [n_struct, H, W] float64 temporary of the
// reference is never materialised: each tap is reduced as soon as it is gathered.  HBM traffic
// per pixel is the algorithmic 4 (field, each frame reused by 3 outputs through L2/MALL) + 16
// (two flow vectors) + 4 or 8 (output) bytes.
#define TF_LANCZOS_TABLE_DEFINED
#include <hip/hip_runtime.h>
__constant__ float c_tf_lanczos[32][8];
#include "remap_dev.h"
#include <type_traits>
#include <stdlib.h>
#include <math.h>
#include <float.h>

// interpolateLanczos4 (imgwarp.cpp) at the 32 fractional positions of the remap table, uploaded on first use
static int ensure_lanczos_table()
{
    static TfDeviceOnce once;                      // __constant__ memory is per device
    TfDeviceOnce::Guard guard(once);
    if (!guard.first) return TF_OK;
    static const double s45 = 0.70710678118654752440084436210485;
    static const double cs[8][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
    float tab[32][8];
    for (int k = 0; k < 32; k++) {
        const float x = (float)k * (1.f / 32.f);
        float *c = tab[k];
        if (x < FLT_EPSILON) { for (int i = 0; i < 8; i++) c[i] = 0; c[3] = 1; continue; }
        float sum = 0;
        const double y0 = -(x + 3) * M_PI * 0.25, s0 = sin(y0), c0 = cos(y0);
        for (int i = 0; i < 8; i++) {
            const double y = -(x + 3 - i) * M_PI * 0.25;
            c[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
            sum += c[i];
        }
        sum = 1.f / sum;
        for (int i = 0; i < 8; i++) c[i] *= sum;
    }
    TF_CHECK_HIP(hipMemcpyToSymbol(HIP_SYMBOL(c_tf_lanczos), tab, sizeof(tab)));
    guard.done();
    return TF_OK;
}

struct ConvTaps {
    int nb, ns, nf;        // taps taken from t-1 (backward flow), t, t+1 (forward flow)
    int8_t ox[27], oy[27]; // (x, y) offset of every tap, in the reference's stack order
    int8_t wx[27], wy[27], wt[27]; // Sobel weights per stack slot (full 27-tap structure only)
};

template <typename TS> struct StackTraits;
template <> struct StackTraits<float> { typedef float In; };
template <> struct StackTraits<double> { typedef float In; };
template <> struct StackTraits<int32_t> { typedef int32_t In; };

template <typename TO> __device__ __forceinline__ void store_out(void *out, int out_type, int64_t idx, TO v) {
    if (out_type == TF_F32) ((float *)out)[idx] = (float)v;
    else if (out_type == TF_F64) ((double *)out)[idx] = (double)v;
    else ((int32_t *)out)[idx] = (int32_t)v;
}

template <typename T> __device__ __forceinline__ bool is_nan_t(T v) { return v != v; }
template <> __device__ __forceinline__ bool is_nan_t<int32_t>(int32_t) { return false; }

// gather one tap of the stack (slot i) for output pixel (x, y) of frame t
template <int METHOD, typename TS>
__device__ __forceinline__ TS gather_tap(const typename StackTraits<TS>::In *__restrict__ data,
                                         const float *__restrict__ fwd, const float *__restrict__ bwd,
                                         int64_t T, int H, int W, int64_t t, int y, int x,
                                         const ConvTaps &tp, int i, typename StackTraits<TS>::In fillv,
                                         float fbx, float fby, float ffx, float ffy)
{
    typedef typename StackTraits<TS>::In In;
    const int ox = tp.ox[i], oy = tp.oy[i];
    const int64_t plane = (int64_t)H * W;
    if (i >= tp.nb && i < tp.nb + tp.ns) {                 // same step: integer gather, OOB -> fill
        int xx = x + ox, yy = y + oy;
        if (xx < 0 || yy < 0 || xx >= W || yy >= H) return (TS)fillv;
        return (TS)data[t * plane + (int64_t)yy * W + xx];
    }
    const bool prev = i < tp.nb;
    const int64_t tt = prev ? t - 1 : t + 1;
    const float mx = tf_loc(prev ? fbx : ffx, ox, x), my = tf_loc(prev ? fby : ffy, oy, y);
    if (tt < 0 || tt >= T) {                             // all-fill neighbour frame (convolve.py:307-314)
        if constexpr (std::is_same<In, int32_t>::value) return (TS)fillv;
        else return (TS)tf_remap_const<METHOD>(H, W, mx, my, (float)fillv);
    }
    const In *img = data + tt * plane;
    if constexpr (std::is_same<In, int32_t>::value) {
        return (TS)tf_remap_nearest<int32_t>(img, H, W, mx, my, fillv);
    } else {
        return (TS)tf_remap<METHOD>(img, H, W, mx, my, fillv);
    }
}

template <int METHOD, typename TS>
__global__ void __launch_bounds__(256)
k_convolve(const typename StackTraits<TS>::In *__restrict__ data, const float *__restrict__ fwd,
           const float *__restrict__ bwd, int64_t T, int H, int W, ConvTaps tp, double fill, int func,
           void *__restrict__ out, int out_type, int64_t t0)
{
    typedef typename StackTraits<TS>::In In;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    const int64_t t = t0 + blockIdx.z;
    if (x >= W || y >= H) return;
    const int64_t plane = (int64_t)H * W, pix = t * plane + (int64_t)y * W + x;
    const In fillv = (In)fill;
    const int n = tp.nb + tp.ns + tp.nf;
    float fbx = 0, fby = 0, ffx = 0, ffy = 0;
    if (tp.nb) { float2 f = ((const float2 *)bwd)[pix]; fbx = f.x; fby = f.y; }
    if (tp.nf) { float2 f = ((const float2 *)fwd)[pix]; ffx = f.x; ffy = f.y; }
    const In centre = data[pix];

    if (func == TF_FUNC_STACK) {
        for (int i = 0; i < n; i++) {
            TS v = gather_tap<METHOD, TS>(data, fwd, bwd, T, H, W, t, y, x, tp, i, fillv, fbx, fby, ffx, ffy);
            store_out<TS>(out, out_type, (int64_t)i * T * plane + pix, v);
        }
        return;
    }
    if (func >= TF_FUNC_SOBEL && func <= TF_FUNC_SOBEL_DOWNHILL) {
        // sobel.py:32-86: x = stack - stack[13] (in the STACK dtype), optional fmax/fmin with 0,
        // three nansum(x * w) in float64 in stack order, sqrt(gx^2 + gy^2 + gt^2)
        if constexpr (!std::is_same<TS, int32_t>::value) {
            const TS c = (TS)centre;
            double gx = 0, gy = 0, gt = 0;
            for (int i = 0; i < n; i++) {
                TS v = gather_tap<METHOD, TS>(data, fwd, bwd, T, H, W, t, y, x, tp, i, fillv, fbx, fby, ffx, ffy);
                TS d = v - c;
                if (func == TF_FUNC_SOBEL_UPHILL) d = (d != d) ? (TS)0 : (d > (TS)0 ? d : (TS)0);
                else if (func == TF_FUNC_SOBEL_DOWNHILL) d = (d != d) ? (TS)0 : (d < (TS)0 ? d : (TS)0);
                const double dd = (double)d;
                double px = dd * (double)tp.wx[i], py = dd * (double)tp.wy[i], pt = dd * (double)tp.wt[i];
                if (px == px) gx += px;
                if (py == py) gy += py;
                if (pt == pt) gt += pt;
            }
            double m = gx * gx;
            m += gy * gy;
            m += gt * gt;
            double r = sqrt(m);
            if (is_nan_t(centre)) r = fill;
            store_out<double>(out, out_type, pix, r);
        }
        return;
    }
    if (func == TF_FUNC_NANMEAN || func == TF_FUNC_NANMAX) {
        if constexpr (!std::is_same<TS, int32_t>::value) {
            TS s = 0; int cnt = 0; TS mx = 0; bool any = false;
            for (int i = 0; i < n; i++) {
                TS v = gather_tap<METHOD, TS>(data, fwd, bwd, T, H, W, t, y, x, tp, i, fillv, fbx, fby, ffx, ffy);
                if (v == v) { s += v; cnt++; mx = (!any || v > mx) ? v : mx; any = true; }
            }
            double r;
            if (func == TF_FUNC_NANMEAN) r = (double)s / (double)cnt;      // 0/0 -> NaN like np.nanmean
            else r = any ? (double)mx : (double)NAN;
            if (is_nan_t(centre)) r = fill;
            store_out<double>(out, out_type, pix, r);
        }
        return;
    }
    if (func == TF_FUNC_DIFF) {
        // flow.py:180-184 with a (prev, same, next) 3-tap structure
        if constexpr (!std::is_same<TS, int32_t>::value) {
            TS x0 = gather_tap<METHOD, TS>(data, fwd, bwd, T, H, W, t, y, x, tp, 0, fillv, fbx, fby, ffx, ffy);
            TS x1 = gather_tap<METHOD, TS>(data, fwd, bwd, T, H, W, t, y, x, tp, 1, fillv, fbx, fby, ffx, ffy);
            TS x2 = gather_tap<METHOD, TS>(data, fwd, bwd, T, H, W, t, y, x, tp, 2, fillv, fbx, fby, ffx, ffy);
            TS a = x2 - x1, b = x1 - x0;
            TS s = (a == a ? a : (TS)0) + (b == b ? b : (TS)0);
            int cnt = (isfinite((double)x2) ? 1 : 0) + (isfinite((double)x0) ? 1 : 0);
            double r = (double)s / (double)(cnt > 1 ? cnt : 1);
            if (is_nan_t(centre)) r = fill;
            store_out<double>(out, out_type, pix, r);
        }
        return;
    }
    if (func == TF_FUNC_ANY) {
        bool any = false;
        for (int i = 0; i < n; i++) {
            TS v = gather_tap<METHOD, TS>(data, fwd, bwd, T, H, W, t, y, x, tp, i, fillv, fbx, fby, ffx, ffy);
            any = any || (v != (TS)0);                       // NaN counts as True, like np.any
        }
        double r = any ? 1.0 : 0.0;
        if (is_nan_t(centre)) r = fill;
        store_out<double>(out, out_type, pix, r);
        return;
    }
}

// ---- dedicated 27-tap semi-Lagrangian Sobel --------------------------------------------------------
// Same arithmetic as k_convolve + TF_FUNC_SOBEL*, specialised for the full 3x3x3 structure:
//  * the 9 taps of a warped plane normally share one sub-pixel phase and sit on consecutive integer
//    positions, so their 4x4 (cubic) / 2x2 (linear) footprints are slices of ONE 6x6 / 4x4 patch:
//    the patch and the 16 / 4 interpolation weights are fetched / built once instead of 9 times,
//    each tap is still the reference's row-major 16-term (4-term) sum -> bit-identical results;
//    planes whose taps do not line up (float rounding of the coordinates) or that touch the image
//    border take the generic per-tap path;
//  * Sobel weights are compile-time constants: zero-weight products are skipped (adding 0 is exact),
//    unit weights need no multiply.
// Measured and NOT adopted (round 2, 12 x 5424^2, fused edge-field variant, 9.9 ms as it stands): staging the two
// warped-frame regions and the same-step tile of a workgroup in LDS (64 x 4 pixels per workgroup: 18.3 ms; 64 x 16:
// 12.3 ms; the same structure with the LDS path switched off: 10.7 ms).  The patch loads hit L1 / L2 and are not what
// bounds the kernel: per pixel it issues 944 vector instructions (round-3 counters, profiles/round3_sobel_pmc.txt: the two
// bicubic planes' 2 x 9 x 16-term row-major sums, which admit no sharing between taps, ~245 float64 operations for the 27
// differences and their weighted sums, coordinates / alignment tests / coefficients), and 1.74e9 issued wave instructions
// per launch against 1.72e9 issue slots: the kernel runs AT the VALU issue limit; its time is its instruction count.
// Round 3, measured and not kept: both warped planes in one pass with every multiply / add of the two planes a packed
// instruction (v_pk_mul_f32 / v_pk_add_f32): 889 instructions, 108 registers, 9.82 vs 9.9 ms -- packed instructions issue
// at ~5 instead of ~4.4 cycles (tools/microbench/pk_rate.hip), the count drops by 6 %, the time by 1 %.
// Also measured: requesting everything a pixel reads before the arithmetic starts (both flows, the same-step
// neighbourhood and BOTH warped patches: two dependent memory round trips instead of five) costs 166 VGPRs / three
// waves per SIMD and runs in 11.1 ms; with only the first patch requested early (114 VGPRs, four waves) 10.0 ms;
// as it stands (96 VGPRs, five waves) 9.9 ms: the kernel follows its occupancy, not its load latency.
template <int METHOD>
__device__ __forceinline__ void sobel_plane_taps(const float *__restrict__ img, int H, int W, int x, int y,
                                                 float flx, float fly, float cval, float (&tap)[9])
{
    if (METHOD == TF_INTERP_NEAREST || METHOD == TF_INTERP_LANCZOS) {      // (a 10 x 10 patch does not fit the registers)
#pragma unroll
        for (int k = 0; k < 9; k++)
            tap[k] = tf_remap<METHOD>(img, H, W, tf_loc(flx, k % 3 - 1, x), tf_loc(fly, k / 3 - 1, y), cval);
        return;
    }
    int fx[3], fy[3];
#pragma unroll
    for (int o = 0; o < 3; o++) { fx[o] = tf_cvround(tf_loc(flx, o - 1, x) * 32.f); fy[o] = tf_cvround(tf_loc(fly, o - 1, y) * 32.f); }
    const bool aligned = fx[0] + 32 == fx[1] && fx[1] + 32 == fx[2] && fy[0] + 32 == fy[1] && fy[1] + 32 == fy[2];
    constexpr int R = METHOD == TF_INTERP_CUBIC ? 4 : 2;       // footprint size
    constexpr int P = R + 2;                                   // patch size
    const int bx = (fx[0] >> 5) - (R == 4 ? 1 : 0), by = (fy[0] >> 5) - (R == 4 ? 1 : 0);
    // every tap must take the reference's "patch fully inside" branch:
    //   cubic: 0 <= sx - 1 <= W - 4 for sx = sx0 .. sx0 + 2   <=>  bx >= 0 and bx + 6 <= W   (bx = sx0 - 1)
    //   linear: 0 <= sx <= W - 2                               <=>  bx >= 0 and bx + 4 <= W   (bx = sx0)
    if (aligned && bx >= 0 && by >= 0 && bx + P <= W && by + P <= H) {
        float wt[R * R];
        if (R == 4) {
            float cx[4], cy[4];
            tf_cubic_coeffs((float)(fx[0] & 31) * (1.f / 32.f), cx);
            tf_cubic_coeffs((float)(fy[0] & 31) * (1.f / 32.f), cy);
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) wt[i * 4 + j] = cy[i] * cx[j];
        } else {
            const float ax = (float)(fx[0] & 31) * (1.f / 32.f), ay = (float)(fy[0] & 31) * (1.f / 32.f);
            wt[0] = (1.f - ay) * (1.f - ax); wt[1] = (1.f - ay) * ax; wt[2] = ay * (1.f - ax); wt[3] = ay * ax;
        }
        float pt[P][P];
#pragma unroll
        for (int r = 0; r < P; r++)
#pragma unroll
            for (int c = 0; c < P; c++) pt[r][c] = img[(int64_t)(by + r) * W + bx + c];
#pragma unroll
        for (int k = 0; k < 9; k++) {
            const int oy = k / 3, ox = k % 3;
            float sum = pt[oy][ox] * wt[0];
#pragma unroll
            for (int q = 1; q < R * R; q++) sum = sum + pt[oy + q / R][ox + q % R] * wt[q];
            tap[k] = sum;
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < 9; k++)
        tap[k] = tf_remap<METHOD>(img, H, W, tf_loc(flx, k % 3 - 1, x), tf_loc(fly, k / 3 - 1, y), cval);
}

template <typename TS, int DIR>
__device__ __forceinline__ void sobel_accumulate(TS v, TS c, int wx, int wy, int wt, double &gx, double &gy, double &gt)
{
    TS d = v - c;
    if (DIR == TF_FUNC_SOBEL_UPHILL) d = (d != d) ? (TS)0 : (d > (TS)0 ? d : (TS)0);
    else if (DIR == TF_FUNC_SOBEL_DOWNHILL) d = (d != d) ? (TS)0 : (d < (TS)0 ? d : (TS)0);
    const double dd = (double)d;
    if (DIR == TF_FUNC_SOBEL && dd != dd) return;            // nansum skips NaN products
    // (inf * 0 is NaN in the reference and is skipped there too; inf * w stays in the sum)
    // The Sobel weights are +-1, +-2, +-4: dd * w is exact in double, so one fused multiply-add returns exactly the
    // value of the reference's multiply followed by its add (a product that carries no rounding cannot be "contracted")
    if (wx) gx = (wx == 1 || wx == -1) ? gx + (wx > 0 ? dd : -dd) : __fma_rn(dd, (double)wx, gx);
    if (wy) gy = (wy == 1 || wy == -1) ? gy + (wy > 0 ? dd : -dd) : __fma_rn(dd, (double)wy, gy);
    if (wt) gt = (wt == 1 || wt == -1) ? gt + (wt > 0 ? dd : -dd) : __fma_rn(dd, (double)wt, gt);
}

// EDGE: the tail of detection.py:638-642 (get_combined_edge_field) is applied to the float64 magnitude before it is
// stored -- edges[edges > 0] += 1; edges -= field; edges[isnan(field)] = inf -- exactly k_edge_field's expressions.
template <int METHOD, typename TS, int DIR, bool EDGE = false>
__global__ void __launch_bounds__(256)
k_sobel27(const float *__restrict__ data, const float *__restrict__ fwd, const float *__restrict__ bwd,
          int64_t T, int H, int W, double fill, void *__restrict__ out, int out_type, int64_t t0)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    const int64_t t = t0 + blockIdx.z;
    if (x >= W || y >= H) return;
    const int64_t plane = (int64_t)H * W, pix = t * plane + (int64_t)y * W + x;
    const float fillf = (float)fill;
    const float centre = data[pix];
    const TS c = (TS)centre;
    double gx = 0, gy = 0, gt = 0;
    const int a3[3] = {1, 2, 1}, d3[3] = {-1, 0, 1};
    float tap[9];
    // plane 0: previous frame through the backward flow (stack slots 0..8)
    {
        const float2 f = ((const float2 *)bwd)[pix];
        if (t > 0) sobel_plane_taps<METHOD>(data + (t - 1) * plane, H, W, x, y, f.x, f.y, fillf, tap);
        else {
#pragma unroll
            for (int k = 0; k < 9; k++) tap[k] = tf_remap_const<METHOD>(H, W, tf_loc(f.x, k % 3 - 1, x), tf_loc(f.y, k / 3 - 1, y), fillf);
        }
    }
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int r = k / 3, cc = k % 3;
        sobel_accumulate<TS, DIR>((TS)tap[k], c, a3[0] * a3[r] * d3[cc], a3[cc] * a3[0] * d3[r], a3[r] * a3[cc] * d3[0], gx, gy, gt);
    }
    // plane 1: same step, integer offsets, out of image -> fill
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int r = k / 3, cc = k % 3;
        const int xx = x + cc - 1, yy = y + r - 1;
        const float v = (xx < 0 || yy < 0 || xx >= W || yy >= H) ? fillf : data[t * plane + (int64_t)yy * W + xx];
        sobel_accumulate<TS, DIR>((TS)v, c, a3[1] * a3[r] * d3[cc], a3[cc] * a3[1] * d3[r], a3[r] * a3[cc] * d3[1], gx, gy, gt);
    }
    // plane 2: next frame through the forward flow
    {
        const float2 f = ((const float2 *)fwd)[pix];
        if (t + 1 < T) sobel_plane_taps<METHOD>(data + (t + 1) * plane, H, W, x, y, f.x, f.y, fillf, tap);
        else {
#pragma unroll
            for (int k = 0; k < 9; k++) tap[k] = tf_remap_const<METHOD>(H, W, tf_loc(f.x, k % 3 - 1, x), tf_loc(f.y, k / 3 - 1, y), fillf);
        }
    }
#pragma unroll
    for (int k = 0; k < 9; k++) {
        const int r = k / 3, cc = k % 3;
        sobel_accumulate<TS, DIR>((TS)tap[k], c, a3[2] * a3[r] * d3[cc], a3[cc] * a3[2] * d3[r], a3[r] * a3[cc] * d3[2], gx, gy, gt);
    }
    double m = gx * gx;
    m += gy * gy;
    m += gt * gt;
    double r = sqrt(m);
    if (centre != centre) r = fill;
    if (EDGE) {
        double e = r;
        if (e > 0) e += 1.0;
        e = e - (double)centre;
        if (centre != centre) e = INFINITY;
        store_out<double>(out, out_type, pix, e);
    } else {
        store_out<double>(out, out_type, pix, r);
    }
}

template <int METHOD, typename TS>
static void launch_sobel27(int func, dim3 grid, dim3 block, hipStream_t s, const float *d, const float *fwd, const float *bwd,
                           int64_t T, int H, int W, double fill, void *out, int out_type, int64_t t0)
{
    if (func == TF_FUNC_SOBEL_UPHILL)
        hipLaunchKernelGGL((k_sobel27<METHOD, TS, TF_FUNC_SOBEL_UPHILL>), grid, block, 0, s, d, fwd, bwd, T, H, W, fill, out, out_type, t0);
    else if (func == TF_FUNC_SOBEL_DOWNHILL)
        hipLaunchKernelGGL((k_sobel27<METHOD, TS, TF_FUNC_SOBEL_DOWNHILL>), grid, block, 0, s, d, fwd, bwd, T, H, W, fill, out, out_type, t0);
    else
        hipLaunchKernelGGL((k_sobel27<METHOD, TS, TF_FUNC_SOBEL>), grid, block, 0, s, d, fwd, bwd, T, H, W, fill, out, out_type, t0);
}

static int build_taps(const uint8_t *structure, ConvTaps &tp) {
    // stack order = plane 0 taps (C order) | plane 1 | plane 2; offsets as (x, y) = (col-1, row-1)
    // (convolve.py:212,224,234: np.where(structure[k])[..., ::-1] - centre)
    int k = 0;
    tp.nb = tp.ns = tp.nf = 0;
    for (int p = 0; p < 3; p++)
        for (int r = 0; r < 3; r++)
            for (int c = 0; c < 3; c++)
                if (structure[p * 9 + r * 3 + c]) {
                    tp.ox[k] = (int8_t)(c - 1); tp.oy[k] = (int8_t)(r - 1);
                    // sobel.py:7-29: S[t,y,x] = a[t] a[y] d[x]; transpose([1,2,0]) -> d along y;
                    // transpose([2,0,1]) -> d along t  (a = [1,2,1], d = [-1,0,1])
                    static const int a[3] = {1, 2, 1}, d[3] = {-1, 0, 1};
                    tp.wx[k] = (int8_t)(a[p] * a[r] * d[c]);
                    tp.wy[k] = (int8_t)(a[c] * a[p] * d[r]);
                    tp.wt[k] = (int8_t)(a[r] * a[c] * d[p]);
                    k++;
                    if (p == 0) tp.nb++; else if (p == 1) tp.ns++; else tp.nf++;
                }
    return k;
}

template <typename TS>
static int launch_convolve(const void *data, int64_t T, int H, int W, const float *fwd, const float *bwd,
                           const ConvTaps &tp, int interp, double fill, int func, void *out, int out_type,
                           int64_t t0, int64_t t1, hipStream_t s)
{
    typedef typename StackTraits<TS>::In In;
    const bool is_sobel = func >= TF_FUNC_SOBEL && func <= TF_FUNC_SOBEL_DOWNHILL;
    const double out_b = (out_type == TF_F64 ? 8.0 : 4.0) * (func == TF_FUNC_STACK ? (tp.nb + tp.ns + tp.nf) : 1);
    TfProfScope ps(is_sobel ? TFK_SOBEL : TFK_CONVOLVE, (4.0 + 16.0 + out_b) * (double)H * W * (double)(t1 - t0), s);
    dim3 block(64, 4, 1), grid((W + 63) / 64, (H + 3) / 4, (unsigned)(t1 - t0));
    const In *d = (const In *)data;
    if constexpr (!std::is_same<TS, int32_t>::value) {
        if (is_sobel && !getenv("TF_SOBEL_GENERIC")) {
            if (interp == TF_INTERP_NEAREST) launch_sobel27<TF_INTERP_NEAREST, TS>(func, grid, block, s, d, fwd, bwd, T, H, W, fill, out, out_type, t0);
            else if (interp == TF_INTERP_LANCZOS) launch_sobel27<TF_INTERP_LANCZOS, TS>(func, grid, block, s, d, fwd, bwd, T, H, W, fill, out, out_type, t0);
            else if (interp == TF_INTERP_LINEAR) launch_sobel27<TF_INTERP_LINEAR, TS>(func, grid, block, s, d, fwd, bwd, T, H, W, fill, out, out_type, t0);
            else launch_sobel27<TF_INTERP_CUBIC, TS>(func, grid, block, s, d, fwd, bwd, T, H, W, fill, out, out_type, t0);
            TF_CHECK_LAUNCH();
            return TF_OK;
        }
    }
    switch (interp) {
    case TF_INTERP_NEAREST:
        hipLaunchKernelGGL((k_convolve<TF_INTERP_NEAREST, TS>), grid, block, 0, s, d, fwd, bwd, T, H, W, tp, fill, func, out, out_type, t0); break;
    case TF_INTERP_LINEAR:
        hipLaunchKernelGGL((k_convolve<TF_INTERP_LINEAR, TS>), grid, block, 0, s, d, fwd, bwd, T, H, W, tp, fill, func, out, out_type, t0); break;
    case TF_INTERP_LANCZOS:
        hipLaunchKernelGGL((k_convolve<TF_INTERP_LANCZOS, TS>), grid, block, 0, s, d, fwd, bwd, T, H, W, tp, fill, func, out, out_type, t0); break;
    default:
        hipLaunchKernelGGL((k_convolve<TF_INTERP_CUBIC, TS>), grid, block, 0, s, d, fwd, bwd, T, H, W, tp, fill, func, out, out_type, t0); break;
    }
    TF_CHECK_LAUNCH();
    return TF_OK;
}

extern "C" int tf_convolve(const void *data, int data_type, int64_t T, int64_t H, int64_t W,
                           const float *fwd, const float *bwd, const uint8_t *structure_host,
                           int interp, double fill, int func, void *out, int out_type,
                           int64_t t0, int64_t t1, void *stream)
{
    TF_REQUIRE(data && fwd && bwd && structure_host && out, "tf_convolve: null pointer");
    TF_REQUIRE(T > 0 && H > 0 && W > 0 && H < (1 << 15) && W < (1 << 15), "tf_convolve: bad shape");
    TF_REQUIRE(interp >= TF_INTERP_NEAREST && interp <= TF_INTERP_LANCZOS, "tf_convolve: bad interp");
    if (interp == TF_INTERP_LANCZOS) { const int rc = ensure_lanczos_table(); if (rc) return rc; }
    TF_REQUIRE(func >= TF_FUNC_STACK && func <= TF_FUNC_NANMAX, "tf_convolve: bad func");
    TF_REQUIRE(data_type == TF_F32 || data_type == TF_I32, "tf_convolve: data_type must be f32 or i32");
    TF_REQUIRE(out_type >= TF_F32 && out_type <= TF_I32, "tf_convolve: bad out_type");
    TF_REQUIRE(data_type != TF_I32 || interp == TF_INTERP_NEAREST, "tf_convolve: int32 data needs nearest");
    TF_REQUIRE(t0 >= 0 && t1 <= T && t0 <= t1, "tf_convolve: bad frame range");
    if (t0 == t1) return TF_OK;
    ConvTaps tp;
    int n = build_taps(structure_host, tp);
    TF_REQUIRE(n > 0, "tf_convolve: empty structure");
    if (func == TF_FUNC_DIFF)
        TF_REQUIRE(tp.nb == 1 && tp.ns == 1 && tp.nf == 1, "tf_convolve: DIFF needs one tap per time plane");
    if (func >= TF_FUNC_SOBEL && func <= TF_FUNC_SOBEL_DOWNHILL)
        TF_REQUIRE(n == 27, "tf_convolve: SOBEL needs the full 27-tap structure");
    hipStream_t s = (hipStream_t)stream;
    if (data_type == TF_I32) {
        TF_REQUIRE(func == TF_FUNC_STACK || func == TF_FUNC_ANY, "tf_convolve: int32 data supports STACK / ANY");
        return launch_convolve<int32_t>(data, T, (int)H, (int)W, fwd, bwd, tp, interp, fill, func, out, out_type, t0, t1, s);
    }
    // the stack dtype is the reference's `dtype` argument: float32 unless the output is float64
    if (out_type == TF_F64)
        return launch_convolve<double>(data, T, (int)H, (int)W, fwd, bwd, tp, interp, fill, func, out, out_type, t0, t1, s);
    return launch_convolve<float>(data, T, (int)H, (int)W, fwd, bwd, tp, interp, fill, func, out, out_type, t0, t1, s);
}

// ---- single-image warp and forward/backward smoothing -----------------------------------------
template <int METHOD>
__global__ void __launch_bounds__(256)
k_warp(const float *__restrict__ img, const float *__restrict__ flow, int H, int W, float *__restrict__ out)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    const int64_t pix = (int64_t)y * W + x;
    float2 f = ((const float2 *)flow)[pix];
    out[pix] = tf_remap<METHOD>(img, H, W, tf_loc(f.x, 0, x), tf_loc(f.y, 0, y), NAN);
}

// two-component sampler: flow images are interleaved (H, W, 2); the reference warps the two components
// as separate (H, W) images with the SAME map (flow.py:545-546), so both are sampled here with one set of
// coordinates / weights and float2 loads -- per component the arithmetic is exactly tf_remap's.
__device__ __forceinline__ float2 f2_mul(float2 a, float w) { return make_float2(a.x * w, a.y * w); }
__device__ __forceinline__ float2 f2_add(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 f2_sub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

template <int METHOD>
__device__ __forceinline__ float2 sample_flow2(const float *__restrict__ f2, int H, int W, float mx, float my)
{
    const float2 cval = make_float2(NAN, NAN);
    auto at = [&](int yy, int xx) { return ((const float2 *)f2)[(int64_t)yy * W + xx]; };
    if (METHOD == TF_INTERP_NEAREST) {
        int sx = tf_sat_short(tf_cvround(mx)), sy = tf_sat_short(tf_cvround(my));
        return ((unsigned)sx < (unsigned)W && (unsigned)sy < (unsigned)H) ? at(sy, sx) : cval;
    }
    int fx = tf_cvround(mx * 32.f), fy = tf_cvround(my * 32.f);
    int sx = tf_sat_short(fx >> 5), sy = tf_sat_short(fy >> 5);
    if (METHOD == TF_INTERP_LINEAR) {
        float ax = (float)(fx & 31) * (1.f / 32.f), ay = (float)(fy & 31) * (1.f / 32.f);
        float w0 = (1.f - ay) * (1.f - ax), w1 = (1.f - ay) * ax, w2 = ay * (1.f - ax), w3 = ay * ax;
        int w1lim = W - 1 > 0 ? W - 1 : 0, h1lim = H - 1 > 0 ? H - 1 : 0;
        if ((unsigned)sx < (unsigned)w1lim && (unsigned)sy < (unsigned)h1lim)
            return f2_add(f2_add(f2_add(f2_mul(at(sy, sx), w0), f2_mul(at(sy, sx + 1), w1)), f2_mul(at(sy + 1, sx), w2)), f2_mul(at(sy + 1, sx + 1), w3));
        if (sx >= W || sx + 1 < 0 || sy >= H || sy + 1 < 0) return cval;
        bool okx0 = sx >= 0 && sx < W, okx1 = sx + 1 >= 0 && sx + 1 < W, oky0 = sy >= 0 && sy < H, oky1 = sy + 1 >= 0 && sy + 1 < H;
        float2 v0 = (okx0 && oky0) ? at(sy, sx) : cval, v1 = (okx1 && oky0) ? at(sy, sx + 1) : cval;
        float2 v2 = (okx0 && oky1) ? at(sy + 1, sx) : cval, v3 = (okx1 && oky1) ? at(sy + 1, sx + 1) : cval;
        return f2_add(f2_add(f2_add(f2_mul(v0, w0), f2_mul(v1, w1)), f2_mul(v2, w2)), f2_mul(v3, w3));
    }
    if (METHOD == TF_INTERP_LANCZOS) {
        const int bx = sx - 3, by = sy - 3;
        const float *wx = c_tf_lanczos[fx & 31], *wy = c_tf_lanczos[fy & 31];
        int w1lim = W - 7 > 0 ? W - 7 : 0, h1lim = H - 7 > 0 ? H - 7 : 0;
        if ((unsigned)bx < (unsigned)w1lim && (unsigned)by < (unsigned)h1lim) {
            float2 sum = make_float2(0.f, 0.f);
            for (int r = 0; r < 8; r++) {
                const float wr = wy[r];
                float2 row = f2_mul(at(by + r, bx), wr * wx[0]);
#pragma unroll
                for (int j = 1; j < 8; j++) row = f2_add(row, f2_mul(at(by + r, bx + j), wr * wx[j]));
                sum = f2_add(sum, row);
            }
            return sum;
        }
        if (bx >= W || bx + 8 <= 0 || by >= H || by + 8 <= 0) return cval;
        float2 sum = f2_mul(cval, 1.f);
        for (int i = 0; i < 8; i++) {
            int yi = by + i;
            if (yi < 0 || yi >= H) continue;
            for (int j = 0; j < 8; j++) {
                int xj = bx + j;
                if (xj >= 0 && xj < W) sum = f2_add(sum, f2_mul(f2_sub(at(yi, xj), cval), wy[i] * wx[j]));
            }
        }
        return sum;
    }
    float cx[4], cy[4];
    tf_cubic_coeffs((float)(fx & 31) * (1.f / 32.f), cx);
    tf_cubic_coeffs((float)(fy & 31) * (1.f / 32.f), cy);
    int bx = sx - 1, by = sy - 1;
    int w1lim = W - 3 > 0 ? W - 3 : 0, h1lim = H - 3 > 0 ? H - 3 : 0;
    if ((unsigned)bx < (unsigned)w1lim && (unsigned)by < (unsigned)h1lim) {
        float2 sum = f2_mul(at(by, bx), cy[0] * cx[0]);
#pragma unroll
        for (int q = 1; q < 16; q++) sum = f2_add(sum, f2_mul(at(by + q / 4, bx + q % 4), cy[q / 4] * cx[q % 4]));
        return sum;
    }
    if (bx >= W || bx + 4 <= 0 || by >= H || by + 4 <= 0) return cval;
    float2 sum = f2_mul(cval, 1.f);
    for (int i = 0; i < 4; i++) {
        int yi = by + i;
        if (yi < 0 || yi >= H) continue;
        for (int j = 0; j < 4; j++) {
            int xj = bx + j;
            if (xj >= 0 && xj < W) sum = f2_add(sum, f2_mul(f2_sub(at(yi, xj), cval), cy[i] * cx[j]));
        }
    }
    return sum;
}

__device__ __forceinline__ float nanmean2(float a, float b) {
    // np.nanmean([a, b], 0) in float32: nansum / count (0/0 -> NaN)
    float s = (a == a ? a : 0.f) + (b == b ? b : 0.f);
    int cnt = (a == a) + (b == b);
    // numpy divides in double and rounds to float.  By 2 and by 1 that is the float operation, bit for bit (scaling by a power of
    // two is exact, the one rounding -- a result in the subnormal range -- is the same correctly rounded step either way); 0 / 0
    // is NaN.  (Round 6: the double-precision division was 140 of the kernel's 491 instructions per pixel, four of them per pixel.)
    return cnt == 2 ? s * 0.5f : (cnt == 1 ? s : __builtin_nanf(""));
}

__device__ __forceinline__ float flow_clip(float v, float maxv) { return (v != v) ? v : fminf(fmaxf(v, -maxv), maxv); }

template <int METHOD>
__global__ void __launch_bounds__(256)
k_smooth(const float *__restrict__ fwd, const float *__restrict__ bwd, int H, int W,
         float *__restrict__ fo, float *__restrict__ bo, float maxv)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    const int64_t pix = (int64_t)y * W + x;
    const float2 f = ((const float2 *)fwd)[pix], b = ((const float2 *)bwd)[pix];
    const float fmx = tf_loc(f.x, 0, x), fmy = tf_loc(f.y, 0, y);
    const float bmx = tf_loc(b.x, 0, x), bmy = tf_loc(b.y, 0, y);
    float2 r;
    const float2 wb = sample_flow2<METHOD>(bwd, H, W, fmx, fmy);
    // maxv: create_flow's clip (flow.py:60-63, np.minimum(np.maximum(v, -max), max): NaN propagates) applied to what is
    // stored; +inf = no clip (tf_smooth_flow_step, and every pass but the last of tf_smooth_flow_step_clip's callers)
    r.x = flow_clip(nanmean2(f.x, -wb.x), maxv);
    r.y = flow_clip(nanmean2(f.y, -wb.y), maxv);
    ((float2 *)fo)[pix] = r;
    const float2 wf = sample_flow2<METHOD>(fwd, H, W, bmx, bmy);
    r.x = flow_clip(nanmean2(b.x, -wf.x), maxv);
    r.y = flow_clip(nanmean2(b.y, -wf.y), maxv);
    ((float2 *)bo)[pix] = r;
}

extern "C" int tf_warp_flow(const float *img, const float *flow, int64_t H, int64_t W, int interp,
                            float *out, void *stream)
{
    TF_REQUIRE(img && flow && out, "tf_warp_flow: null pointer");
    TF_REQUIRE(H > 0 && W > 0 && H < (1 << 15) && W < (1 << 15), "tf_warp_flow: bad shape");
    TF_REQUIRE(interp >= 0 && interp <= 3, "tf_warp_flow: bad interp");
    if (interp == TF_INTERP_LANCZOS) { const int rc = ensure_lanczos_table(); if (rc) return rc; }
    dim3 block(64, 4), grid((W + 63) / 64, (H + 3) / 4);
    hipStream_t s = (hipStream_t)stream;
    if (interp == 0) hipLaunchKernelGGL(k_warp<0>, grid, block, 0, s, img, flow, (int)H, (int)W, out);
    else if (interp == 1) hipLaunchKernelGGL(k_warp<1>, grid, block, 0, s, img, flow, (int)H, (int)W, out);
    else if (interp == 3) hipLaunchKernelGGL(k_warp<3>, grid, block, 0, s, img, flow, (int)H, (int)W, out);
    else hipLaunchKernelGGL(k_warp<2>, grid, block, 0, s, img, flow, (int)H, (int)W, out);
    TF_CHECK_LAUNCH();
    return TF_OK;
}

static int smooth_flow_step(const float *fwd, const float *bwd, int64_t H, int64_t W, int interp,
                            float *fwd_out, float *bwd_out, float max_value, void *stream)
{
    TF_REQUIRE(fwd && bwd && fwd_out && bwd_out, "tf_smooth_flow_step: null pointer");
    TF_REQUIRE(fwd_out != fwd && fwd_out != bwd && bwd_out != fwd && bwd_out != bwd, "tf_smooth_flow_step: outputs alias inputs");
    TF_REQUIRE(H > 0 && W > 0 && H < (1 << 15) && W < (1 << 15), "tf_smooth_flow_step: bad shape");
    TF_REQUIRE(interp >= 0 && interp <= 3, "tf_smooth_flow_step: bad interp");
    TF_REQUIRE(max_value >= 0.f, "tf_smooth_flow_step_clip: max_value must be >= 0 (+inf: no clip)");
    if (interp == TF_INTERP_LANCZOS) { const int rc = ensure_lanczos_table(); if (rc) return rc; }
    dim3 block(64, 4), grid((W + 63) / 64, (H + 3) / 4);
    hipStream_t s = (hipStream_t)stream;
    TfProfScope ps(TFK_SMOOTH, 48.0 * (double)H * W, s);
    if (interp == 0) hipLaunchKernelGGL(k_smooth<0>, grid, block, 0, s, fwd, bwd, (int)H, (int)W, fwd_out, bwd_out, max_value);
    else if (interp == 1) hipLaunchKernelGGL(k_smooth<1>, grid, block, 0, s, fwd, bwd, (int)H, (int)W, fwd_out, bwd_out, max_value);
    else if (interp == 3) hipLaunchKernelGGL(k_smooth<3>, grid, block, 0, s, fwd, bwd, (int)H, (int)W, fwd_out, bwd_out, max_value);
    else hipLaunchKernelGGL(k_smooth<2>, grid, block, 0, s, fwd, bwd, (int)H, (int)W, fwd_out, bwd_out, max_value);
    TF_CHECK_LAUNCH();
    return TF_OK;
}

extern "C" int tf_smooth_flow_step(const float *fwd, const float *bwd, int64_t H, int64_t W, int interp,
                                   float *fwd_out, float *bwd_out, void *stream)
{
    return smooth_flow_step(fwd, bwd, H, W, interp, fwd_out, bwd_out, INFINITY, stream);
}

extern "C" int tf_smooth_flow_step_clip(const float *fwd, const float *bwd, int64_t H, int64_t W, int interp,
                                        float *fwd_out, float *bwd_out, float max_value, void *stream)
{
    return smooth_flow_step(fwd, bwd, H, W, interp, fwd_out, bwd_out, max_value, stream);
}

__global__ void k_flow_finalize(float *__restrict__ fwd, float *__restrict__ bwd, int64_t T, int64_t plane2, float maxv)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // index inside one (H, W, 2) frame
    const int64_t t = blockIdx.y;
    if (i >= plane2) return;
    const int64_t idx = t * plane2 + i;
    float f = fwd[idx], b = bwd[idx];
    if (t == T - 1) f = -b;                       // flow.py:425  forward[-1] = -backward[-1]
    if (t == 0) b = -f;                           // flow.py:426  backward[0] = -forward[0] (after :425)
    // np.minimum(np.maximum(v, -max), max): NaN propagates
    f = (f != f) ? f : fminf(fmaxf(f, -maxv), maxv);
    b = (b != b) ? b : fminf(fmaxf(b, -maxv), maxv);
    fwd[idx] = f; bwd[idx] = b;
}

extern "C" int tf_flow_finalize(float *fwd, float *bwd, int64_t T, int64_t H, int64_t W, float max_value, void *stream)
{
    TF_REQUIRE(fwd && bwd, "tf_flow_finalize: null pointer");
    TF_REQUIRE(T > 0 && H > 0 && W > 0 && T < 65536, "tf_flow_finalize: bad shape");
    const int64_t plane2 = H * W * 2;
    dim3 block(256), grid((unsigned)((plane2 + 255) / 256), (unsigned)T);
    hipLaunchKernelGGL(k_flow_finalize, grid, block, 0, (hipStream_t)stream, fwd, bwd, T, plane2, max_value);
    TF_CHECK_LAUNCH();
    return TF_OK;
}

// The two END frames only (flow.py:425-426), for stacks whose interior was clipped where it was produced
// (tf_smooth_flow_step_clip): forward[T - 1] = -backward[T - 1], backward[0] = -forward[0], clipped like the rest.
// The full pass reads and writes both flow arrays once more -- 136 GB for 144 x 5424^2.
__global__ void k_flow_finalize_ends(float *__restrict__ fwd, float *__restrict__ bwd, int64_t T, int64_t plane2, float maxv)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= plane2) return;
    const int64_t last = (T - 1) * plane2 + i;
    const float f_last = flow_clip(-bwd[last], maxv);
    fwd[last] = f_last;
    bwd[i] = flow_clip(-(T == 1 ? f_last : fwd[i]), maxv);
}

extern "C" int tf_flow_finalize_ends(float *fwd, float *bwd, int64_t T, int64_t H, int64_t W, float max_value, void *stream)
{
    TF_REQUIRE(fwd && bwd, "tf_flow_finalize_ends: null pointer");
    TF_REQUIRE(T > 0 && H > 0 && W > 0 && T < 65536, "tf_flow_finalize_ends: bad shape");
    const int64_t plane2 = H * W * 2;
    hipLaunchKernelGGL(k_flow_finalize_ends, dim3((unsigned)((plane2 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, fwd, bwd, T, plane2, max_value);
    TF_CHECK_LAUNCH();
    return TF_OK;
}

// ---- combined edge field (detection.py:638-642), fused elementwise -----------------------------------
//   edges[edges > 0] += 1; edges = edges - field; edges[isnan(field)] = inf   (float64 arithmetic as in the
//   reference), optionally rounded once to float32 -- the cast watershed.py:64-65 applies anyway.
__global__ void __launch_bounds__(256)
k_edge_field(const double *__restrict__ sob, const float *__restrict__ field, int64_t n, void *__restrict__ out, int out_type)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double e = sob[i];
    const float f = field[i];
    if (e > 0) e += 1.0;
    e = e - (double)f;
    if (f != f) e = INFINITY;
    if (out_type == TF_F64) ((double *)out)[i] = e; else ((float *)out)[i] = (float)e;
}

extern "C" int tf_sobel_edge_field(const float *field, int64_t T, int64_t H, int64_t W, const float *fwd, const float *bwd,
                                   int interp, void *out, int out_type, void *stream)
{
    TF_REQUIRE(field && fwd && bwd && out, "tf_sobel_edge_field: null pointer");
    TF_REQUIRE(T > 0 && H > 0 && W > 0 && H < (1 << 15) && W < (1 << 15), "tf_sobel_edge_field: bad shape");
    TF_REQUIRE(interp >= TF_INTERP_NEAREST && interp <= TF_INTERP_LANCZOS, "tf_sobel_edge_field: bad interp");
    if (interp == TF_INTERP_LANCZOS) { const int rc = ensure_lanczos_table(); if (rc) return rc; }
    TF_REQUIRE(out_type == TF_F32 || out_type == TF_F64, "tf_sobel_edge_field: out_type must be f32 or f64");
    hipStream_t s = (hipStream_t)stream;
    const double fill = NAN;                                  // Flow.sobel's default fill_value
    TfProfScope ps(TFK_SOBEL, (4.0 + 16.0 + (out_type == TF_F64 ? 8.0 : 4.0)) * (double)H * W * (double)T, s);
    dim3 block(64, 4, 1), grid((unsigned)((W + 63) / 64), (unsigned)((H + 3) / 4), (unsigned)T);
    // the Sobel stack is float64 (Flow.sobel(dtype=None)); only the finished edge value is rounded to out_type
    if (interp == TF_INTERP_NEAREST)
        hipLaunchKernelGGL((k_sobel27<TF_INTERP_NEAREST, double, TF_FUNC_SOBEL_UPHILL, true>), grid, block, 0, s, field, fwd, bwd, T, (int)H, (int)W, fill, out, out_type, (int64_t)0);
    else if (interp == TF_INTERP_LINEAR)
        hipLaunchKernelGGL((k_sobel27<TF_INTERP_LINEAR, double, TF_FUNC_SOBEL_UPHILL, true>), grid, block, 0, s, field, fwd, bwd, T, (int)H, (int)W, fill, out, out_type, (int64_t)0);
    else if (interp == TF_INTERP_LANCZOS)
        hipLaunchKernelGGL((k_sobel27<TF_INTERP_LANCZOS, double, TF_FUNC_SOBEL_UPHILL, true>), grid, block, 0, s, field, fwd, bwd, T, (int)H, (int)W, fill, out, out_type, (int64_t)0);
    else
        hipLaunchKernelGGL((k_sobel27<TF_INTERP_CUBIC, double, TF_FUNC_SOBEL_UPHILL, true>), grid, block, 0, s, field, fwd, bwd, T, (int)H, (int)W, fill, out, out_type, (int64_t)0);
    TF_CHECK_LAUNCH();
    return TF_OK;
}

extern "C" int tf_edge_field(const double *sobel, const float *field, int64_t n, void *out, int out_type, void *stream)
{
    TF_REQUIRE(sobel && field && out && n > 0, "tf_edge_field: bad arguments");
    TF_REQUIRE(out_type == TF_F32 || out_type == TF_F64, "tf_edge_field: out_type must be f32 or f64");
    TfProfScope ps(TFK_CONVOLVE, (8.0 + 4.0 + (out_type == TF_F64 ? 8.0 : 4.0)) * (double)n, (hipStream_t)stream);
    hipLaunchKernelGGL(k_edge_field, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, sobel, field, n, out, out_type);
    TF_CHECK_LAUNCH();
    return TF_OK;
}
