// Device-side cv2.remap semantics (map = absolute float coordinates, BORDER_CONSTANT) as the
// reference calls it: /root/reference/tobac_flow/convolve.py:65-84 and
// /root/reference/tobac_flow/utils/flow_utils.py:90-98.
//  * non-nearest modes quantise the coordinate to 1/32 px (cvRound(x*32); >>5 / &31) and take
//    the weights of that bin; weights and accumulation are float, taps row-major;
//  * a patch that straddles the border takes the border value for outside taps (bilinear) or
//    cval + SUM (S - cval) * w (bicubic): a NaN border value poisons the result;
//  * nearest uses cvRound (half to even).
// Built with -ffp-contract=off: every expression below is evaluated as written.
#pragma once
#include "tf_common.h"

__device__ __forceinline__ void tf_cubic_coeffs(float x, float *c) {
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

// absolute sampling coordinate exactly as numpy builds it (convolve.py:56-63):
//   locs = flow + offset (float32)  ;  locs += grid (int64)  ->  float32(float64(locs) + grid)
__device__ __forceinline__ float tf_loc(float flow, int off, int grid) {
    float l = flow + (float)off;
    return (float)((double)l + (double)grid);
}

template <typename T>
__device__ __forceinline__ T tf_remap_nearest(const T *__restrict__ img, int h, int w, float mx, float my, T cval) {
    int sx = tf_sat_short(tf_cvround(mx)), sy = tf_sat_short(tf_cvround(my));
    return ((unsigned)sx < (unsigned)w && (unsigned)sy < (unsigned)h) ? img[(int64_t)sy * w + sx] : cval;
}

__device__ __forceinline__ float tf_remap_linear(const float *__restrict__ img, int h, int w, float mx, float my, float cval) {
    int fx = tf_cvround(mx * 32.f), fy = tf_cvround(my * 32.f);
    int sx = tf_sat_short(fx >> 5), sy = tf_sat_short(fy >> 5);
    float ax = (float)(fx & 31) * (1.f / 32.f), ay = (float)(fy & 31) * (1.f / 32.f);
    float wx0 = 1.f - ax, wx1 = ax, wy0 = 1.f - ay, wy1 = ay;
    float w0 = wy0 * wx0, w1 = wy0 * wx1, w2 = wy1 * wx0, w3 = wy1 * wx1;
    int w1lim = w - 1 > 0 ? w - 1 : 0, h1lim = h - 1 > 0 ? h - 1 : 0;
    if ((unsigned)sx < (unsigned)w1lim && (unsigned)sy < (unsigned)h1lim) {
        const float *S = img + (int64_t)sy * w + sx;
        return S[0] * w0 + S[1] * w1 + S[w] * w2 + S[w + 1] * w3;
    }
    if (sx >= w || sx + 1 < 0 || sy >= h || sy + 1 < 0) return cval;
    bool okx0 = sx >= 0 && sx < w, okx1 = sx + 1 >= 0 && sx + 1 < w;
    bool oky0 = sy >= 0 && sy < h, oky1 = sy + 1 >= 0 && sy + 1 < h;
    float v0 = (okx0 && oky0) ? img[(int64_t)sy * w + sx] : cval;
    float v1 = (okx1 && oky0) ? img[(int64_t)sy * w + sx + 1] : cval;
    float v2 = (okx0 && oky1) ? img[(int64_t)(sy + 1) * w + sx] : cval;
    float v3 = (okx1 && oky1) ? img[(int64_t)(sy + 1) * w + sx + 1] : cval;
    return v0 * w0 + v1 * w1 + v2 * w2 + v3 * w3;
}

__device__ __forceinline__ float tf_remap_cubic(const float *__restrict__ img, int h, int w, float mx, float my, float cval) {
    int fx = tf_cvround(mx * 32.f), fy = tf_cvround(my * 32.f);
    int sx = tf_sat_short(fx >> 5), sy = tf_sat_short(fy >> 5);
    float cx[4], cy[4];
    tf_cubic_coeffs((float)(fx & 31) * (1.f / 32.f), cx);
    tf_cubic_coeffs((float)(fy & 31) * (1.f / 32.f), cy);
    int bx = sx - 1, by = sy - 1;
    int w1lim = w - 3 > 0 ? w - 3 : 0, h1lim = h - 3 > 0 ? h - 3 : 0;
    if ((unsigned)bx < (unsigned)w1lim && (unsigned)by < (unsigned)h1lim) {
        const float *S = img + (int64_t)by * w + bx;
        float sum = S[0] * (cy[0] * cx[0]) + S[1] * (cy[0] * cx[1]) + S[2] * (cy[0] * cx[2]) + S[3] * (cy[0] * cx[3]);
#pragma unroll
        for (int i = 1; i < 4; i++) {
            const float *R = S + (int64_t)i * w;
            sum = sum + R[0] * (cy[i] * cx[0]);
            sum = sum + R[1] * (cy[i] * cx[1]);
            sum = sum + R[2] * (cy[i] * cx[2]);
            sum = sum + R[3] * (cy[i] * cx[3]);
        }
        return sum;
    }
    if (bx >= w || bx + 4 <= 0 || by >= h || by + 4 <= 0) return cval;
    float sum = cval * 1.f;
    for (int i = 0; i < 4; i++) {
        int yi = by + i;
        if (yi < 0 || yi >= h) continue;
        for (int j = 0; j < 4; j++) {
            int xj = bx + j;
            if (xj >= 0 && xj < w) sum += (img[(int64_t)yi * w + xj] - cval) * (cy[i] * cx[j]);
        }
    }
    return sum;
}

// cv2.INTER_LANCZOS4 (imgwarp.cpp remapLanczos4): 8 x 8 taps around (sx - 3, sy - 3), 1-D weights from a 32-entry
// table (interpolateLanczos4 at f = k / 32, built on the host with the C library's sin / cos exactly as OpenCV builds
// it, uploaded once), 2-D weight = wy * wx in float; the sum goes ROW BY ROW: sum += (S0 w0 + S1 w1 + ... + S7 w7).
// Border (BORDER_CONSTANT): fully outside -> cval; straddling -> cval + SUM (S - cval) w over the inside taps.
#ifndef TF_LANCZOS_TABLE_DEFINED
extern __constant__ float c_tf_lanczos[32][8];
#endif

__device__ __forceinline__ float tf_remap_lanczos(const float *__restrict__ img, int h, int w, float mx, float my, float cval) {
    int fx = tf_cvround(mx * 32.f), fy = tf_cvround(my * 32.f);
    int sx = tf_sat_short(fx >> 5) - 3, sy = tf_sat_short(fy >> 5) - 3;
    const float *wx = c_tf_lanczos[fx & 31], *wy = c_tf_lanczos[fy & 31];
    int w1lim = w - 7 > 0 ? w - 7 : 0, h1lim = h - 7 > 0 ? h - 7 : 0;
    if ((unsigned)sx < (unsigned)w1lim && (unsigned)sy < (unsigned)h1lim) {
        const float *S = img + (int64_t)sy * w + sx;
        float sum = 0.f;
        for (int r = 0; r < 8; r++, S += w) {
            const float wr = wy[r];
            sum += S[0] * (wr * wx[0]) + S[1] * (wr * wx[1]) + S[2] * (wr * wx[2]) + S[3] * (wr * wx[3])
                 + S[4] * (wr * wx[4]) + S[5] * (wr * wx[5]) + S[6] * (wr * wx[6]) + S[7] * (wr * wx[7]);
        }
        return sum;
    }
    if (sx >= w || sx + 8 <= 0 || sy >= h || sy + 8 <= 0) return cval;
    float sum = cval * 1.f;
    for (int i = 0; i < 8; i++) {
        int yi = sy + i;
        if (yi < 0 || yi >= h) continue;
        for (int j = 0; j < 8; j++) {
            int xj = sx + j;
            if (xj >= 0 && xj < w) sum += (img[(int64_t)yi * w + xj] - cval) * (wy[i] * wx[j]);
        }
    }
    return sum;
}

template <int METHOD>
__device__ __forceinline__ float tf_remap(const float *__restrict__ img, int h, int w, float mx, float my, float cval) {
    if (METHOD == TF_INTERP_NEAREST) return tf_remap_nearest<float>(img, h, w, mx, my, cval);
    if (METHOD == TF_INTERP_LINEAR) return tf_remap_linear(img, h, w, mx, my, cval);
    if (METHOD == TF_INTERP_LANCZOS) return tf_remap_lanczos(img, h, w, mx, my, cval);
    return tf_remap_cubic(img, h, w, mx, my, cval);
}

// The reference substitutes a missing neighbour frame (t-1 before the first frame, t+1 after the last) by an image
// that is CONSTANT = fill_value and still runs cv2.remap on it (convolve.py:307-314).  For NaN that is NaN; for a
// numeric fill the interpolation of a constant image is fill * (sum of weights), which is not exactly fill in
// float.  This evaluates tf_remap on such a constant image without touching memory (same branches, same order).
template <int METHOD>
__device__ __forceinline__ float tf_remap_const(int h, int w, float mx, float my, float fill) {
    if (METHOD == TF_INTERP_NEAREST || fill != fill) return fill;
    int fx = tf_cvround(mx * 32.f), fy = tf_cvround(my * 32.f);
    int sx = tf_sat_short(fx >> 5), sy = tf_sat_short(fy >> 5);
    if (METHOD == TF_INTERP_LINEAR) {
        if (sx >= w || sx + 1 < 0 || sy >= h || sy + 1 < 0) {
            int w1lim = w - 1 > 0 ? w - 1 : 0, h1lim = h - 1 > 0 ? h - 1 : 0;
            if (!((unsigned)sx < (unsigned)w1lim && (unsigned)sy < (unsigned)h1lim)) return fill;
        }
        float ax = (float)(fx & 31) * (1.f / 32.f), ay = (float)(fy & 31) * (1.f / 32.f);
        float w0 = (1.f - ay) * (1.f - ax), w1 = (1.f - ay) * ax, w2 = ay * (1.f - ax), w3 = ay * ax;
        return fill * w0 + fill * w1 + fill * w2 + fill * w3;      // inside and border taps all read `fill`
    }
    if (METHOD == TF_INTERP_LANCZOS) {
        int bx = sx - 3, by = sy - 3;
        int w1lim = w - 7 > 0 ? w - 7 : 0, h1lim = h - 7 > 0 ? h - 7 : 0;
        if ((unsigned)bx < (unsigned)w1lim && (unsigned)by < (unsigned)h1lim) {
            const float *wx = c_tf_lanczos[fx & 31], *wy = c_tf_lanczos[fy & 31];
            float sum = 0.f;
            for (int r = 0; r < 8; r++) {
                const float wr = wy[r];
                sum += fill * (wr * wx[0]) + fill * (wr * wx[1]) + fill * (wr * wx[2]) + fill * (wr * wx[3])
                     + fill * (wr * wx[4]) + fill * (wr * wx[5]) + fill * (wr * wx[6]) + fill * (wr * wx[7]);
            }
            return sum;
        }
        return fill;       // fully outside -> cval; straddling -> cval + sum (fill - cval) * w = fill
    }
    int bx = sx - 1, by = sy - 1;
    int w1lim = w - 3 > 0 ? w - 3 : 0, h1lim = h - 3 > 0 ? h - 3 : 0;
    if ((unsigned)bx < (unsigned)w1lim && (unsigned)by < (unsigned)h1lim) {
        float cx[4], cy[4];
        tf_cubic_coeffs((float)(fx & 31) * (1.f / 32.f), cx);
        tf_cubic_coeffs((float)(fy & 31) * (1.f / 32.f), cy);
        float sum = fill * (cy[0] * cx[0]);
#pragma unroll
        for (int q = 1; q < 16; q++) sum = sum + fill * (cy[q / 4] * cx[q % 4]);
        return sum;
    }
    return fill;           // fully outside -> cval; straddling -> cval + sum (fill - cval) * w = fill
}
