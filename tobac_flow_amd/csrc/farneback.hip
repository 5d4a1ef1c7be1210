// Pyramidal Farnebaeck dense optical flow for one frame pair, both directions, on gfx950.
//
// Replaces cv2.optflow.createOptFlow_Farneback().calc(prev, next, None) / .calc(next, prev, None)
// as issued by /root/reference/tobac_flow/flow.py:511,516 (factory
// /root/reference/tobac_flow/utils/flow_utils.py:52-53).  Algorithm = OpenCV's
// modules/video/src/optflowgf.cpp with its defaults:
//   per level k (coarse -> fine):  u8 -> f32, GaussianBlur of the FULL-RES image (REFLECT_101),
//   INTER_LINEAR resize to the level size, polynomial expansion (n = 5, sigma = 1.1) -> 5-channel R,
//   flow init (zeros | upsampled previous level * 2), UpdateMatrices, then numIters x
//   [13x13 box filter of M in double -> per-pixel 2x2 solve -> flow ; UpdateMatrices].
//
// MI355X design notes (what the file does today; history and measurements: DESIGN.md section 4)
//   * both directions of a pair share the Gaussian pyramid and the polynomial expansion of the two images, and a
//     BATCH of pairs goes through every launch (blockIdx.z) so that the coarse levels fill the chip;
//   * the blur is evaluated only at the pixels the resize samples (k_fb_blur_rows_sampled / _cols_resize; fused 3 x 3
//     forms for the two finest levels);
//   * the polynomial expansion R is stored as float4 {y, x, yy, xx} + a float plane {xy}: a bilinear gather is five wide
//     loads per corner pair instead of twenty scalar ones (k_fb_polyexp / k_fb_polyexp5, LDS tiles);
//   * ONE kernel per iteration (k_fb_iter): UpdateMatrices -> 13 x 13 box sums in double -> 2 x 2 solve; each thread walks
//     down a column with the last 13 rows of M in a register ring and OpenCV's running column sum, the column sums of a row
//     group go to LDS, and (round 4) OpenCV's running ROW sum is carried through them by 25 chain lanes and from column strip
//     to column strip through tagged words in global memory: the flow is bit-identical to the oracle's.  The 5-channel M of
//     OpenCV never exists in HBM.  (k_fb_iter_tree, round 3: window sums as a tree, within 7e-5 px; TF_FB_ROW_SUMS_TREE=1.)
//     The unfused pair k_fb_update_matrices + k_fb_blur_solve (LDS tiles, one channel at a time) serves window sizes other
//     than 13;
//   * all of these are HBM / L2- or latency-bound stencils: no MFMA.
#include "tf_common.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <atomic>

#define FB_MAX_KSIZE 255
#define FB_MAX_POLY_N 8

__device__ __forceinline__ int fb_reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}

// ---- Gaussian blur (cv::GaussianBlur on CV_32F: row pass then symmetric column pass) -----------
struct FbKernel { int ksize; float k[FB_MAX_KSIZE]; };

template <typename TIn>
__global__ void __launch_bounds__(256)
k_fb_blur_rows(const TIn *__restrict__ src, int H, int W, const FbKernel kk, float *__restrict__ dst, int64_t bs_src, int64_t bs_dst)
{
    src += (int64_t)blockIdx.z * bs_src; dst += (int64_t)blockIdx.z * bs_dst;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    const int ksize = kk.ksize, r = ksize >> 1;
    const float *k = kk.k;
    const TIn *S = src + (int64_t)y * W;
    float s;
    if (ksize == 3) {
        s = (float)S[x] * k[1] + ((float)S[fb_reflect101(x - 1, W)] + (float)S[fb_reflect101(x + 1, W)]) * k[0];
    } else if (ksize == 5) {
        s = (float)S[x] * k[2] + ((float)S[fb_reflect101(x - 1, W)] + (float)S[fb_reflect101(x + 1, W)]) * k[1]
          + ((float)S[fb_reflect101(x - 2, W)] + (float)S[fb_reflect101(x + 2, W)]) * k[0];
    } else {
        s = k[0] * (float)S[fb_reflect101(x - r, W)];
        if (x - r >= 0 && x + r < W) { for (int i = 1; i < ksize; i++) s += k[i] * (float)S[x - r + i]; }
        else { for (int i = 1; i < ksize; i++) s += k[i] * (float)S[fb_reflect101(x - r + i, W)]; }
    }
    dst[(int64_t)y * W + x] = s;
}

__global__ void __launch_bounds__(256)
k_fb_blur_cols(const float *__restrict__ src, int H, int W, const FbKernel kk, float *__restrict__ dst, int64_t bs_src, int64_t bs_dst)
{
    src += (int64_t)blockIdx.z * bs_src; dst += (int64_t)blockIdx.z * bs_dst;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    const int r = kk.ksize >> 1;
    const float *k = kk.k;
    float s = k[r] * src[(int64_t)y * W + x];
    if (y - r >= 0 && y + r < H) {
        for (int i = 1; i <= r; i++) s += k[r + i] * (src[(int64_t)(y + i) * W + x] + src[(int64_t)(y - i) * W + x]);
    } else {
        for (int i = 1; i <= r; i++)
            s += k[r + i] * (src[(int64_t)fb_reflect101(y + i, H) * W + x] + src[(int64_t)fb_reflect101(y - i, H) * W + x]);
    }
    dst[(int64_t)y * W + x] = s;
}

// 3 x 3 Gaussian blur in one pass (the kernel size of pyramid levels 0 and 1 with the default pyr_scale): every thread
// forms the three row sums its column pass needs, with exactly the expressions of k_fb_blur_rows / k_fb_blur_cols, so the
// result is bit-identical to the two-pass form while the float intermediate (4 B written + 4 B read per pixel) never
// exists: 1 B read + 4 B written per pixel instead of 13.
template <typename TIn>
__global__ void __launch_bounds__(256)
k_fb_blur3_fused(const TIn *__restrict__ src, int H, int W, const FbKernel kk, float *__restrict__ dst, int64_t bs_src, int64_t bs_dst)
{
    // a thread owns FOUR consecutive pixels of a row: 3 x 6 source pixels feed them (18 loads instead of 36), one
    // 16-byte store when the row start allows it
    src += (int64_t)blockIdx.z * bs_src; dst += (int64_t)blockIdx.z * bs_dst;
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4, y = blockIdx.y * 4 + threadIdx.y;
    if (x0 >= W || y >= H) return;
    const float *k = kk.k;
    const int yu = fb_reflect101(y - 1, H), yd = fb_reflect101(y + 1, H);
    const TIn *S0 = src + (int64_t)y * W, *Su = src + (int64_t)yu * W, *Sd = src + (int64_t)yd * W;
    int xs[6];
#pragma unroll
    for (int j = 0; j < 6; j++) xs[j] = fb_reflect101(min(x0 - 1 + j, W), W);      // columns x0 - 1 .. x0 + 4 (past-the-end ones unused)
    float v0[6], vu[6], vd[6];
    // uint8 rows of whole, aligned words (the production case): the six bytes of a row come from three word loads -- the
    // word of the four pixels and its two neighbours -- instead of six byte loads
    bool words = false;
    if (sizeof(TIn) == 1) {
        words = (W & 3) == 0 && (bs_src & 3) == 0 && ((uintptr_t)src & 3) == 0 && x0 >= 4 && x0 + 8 <= W;
        if (words) {
            auto row6 = [&](const TIn *S, float (&v)[6]) {
                const uint32_t *p = (const uint32_t *)(S + x0);
                const uint32_t l = p[-1], c = p[0], r = p[1];
                v[0] = (float)(l >> 24); v[1] = (float)(c & 0xffu); v[2] = (float)((c >> 8) & 0xffu); v[3] = (float)((c >> 16) & 0xffu);
                v[4] = (float)(c >> 24); v[5] = (float)(r & 0xffu);
            };
            row6(S0, v0); row6(Su, vu); row6(Sd, vd);
        }
    }
    if (!words) {
#pragma unroll
        for (int j = 0; j < 6; j++) { v0[j] = (float)S0[xs[j]]; vu[j] = (float)Su[xs[j]]; vd[j] = (float)Sd[xs[j]]; }
    }
    float o[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const float r0 = v0[e + 1] * k[1] + (v0[e] + v0[e + 2]) * k[0];
        const float ru = vu[e + 1] * k[1] + (vu[e] + vu[e + 2]) * k[0];
        const float rd = vd[e + 1] * k[1] + (vd[e] + vd[e + 2]) * k[0];
        float s = k[1] * r0;
        s += k[2] * (rd + ru);                                // k_fb_blur_cols: k[r + 1] * (tmp[y + 1] + tmp[y - 1])
        o[e] = s;
    }
    float *D = dst + (int64_t)y * W + x0;
    if (x0 + 3 < W && (((uintptr_t)D) & 15) == 0) *(float4 *)D = make_float4(o[0], o[1], o[2], o[3]);
    else {
#pragma unroll
        for (int e = 0; e < 4; e++) if (x0 + e < W) D[e] = o[e];
    }
}

// 3 x 3 Gaussian blur followed by the exact-2x INTER_AREA resize (pyramid level 1), in one pass: a thread of the
// half-resolution grid forms the four blurred values of its 2 x 2 block with the expressions of k_fb_blur3_fused and
// averages them in k_fb_resize_area2's order.  Bit-identical to blur-then-resize; the full-resolution float image of
// this level (4 B written + 4 B read per pixel) never exists.
template <typename TIn>
__global__ void __launch_bounds__(256)
k_fb_blur3_area2(const TIn *__restrict__ src, int H, int W, const FbKernel kk, float *__restrict__ dst, int dh, int dw,
                 int64_t bs_src, int64_t bs_dst)
{
    src += (int64_t)blockIdx.z * bs_src; dst += (int64_t)blockIdx.z * bs_dst;
    const int dx = blockIdx.x * 64 + threadIdx.x, dy = blockIdx.y * 4 + threadIdx.y;
    if (dx >= dw || dy >= dh) return;
    const float *k = kk.k;
    const int x0 = 2 * dx, y0 = 2 * dy;
    const int xs[4] = {fb_reflect101(x0 - 1, W), x0, x0 + 1, fb_reflect101(x0 + 2, W)};
    const int ys[4] = {fb_reflect101(y0 - 1, H), y0, y0 + 1, fb_reflect101(y0 + 2, H)};
    float r[4][2];                                            // row sums of rows ys[j] at columns x0, x0 + 1
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const TIn *S = src + (int64_t)ys[j] * W;
        const float v0 = (float)S[xs[0]], v1 = (float)S[xs[1]], v2 = (float)S[xs[2]], v3 = (float)S[xs[3]];
        r[j][0] = v1 * k[1] + (v0 + v2) * k[0];               // column x0:     S[x] k1 + (S[x-1] + S[x+1]) k0
        r[j][1] = v2 * k[1] + (v1 + v3) * k[0];               // column x0 + 1
    }
    float b[2][2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        float s0 = k[1] * r[1][c]; s0 += k[2] * (r[2][c] + r[0][c]);        // row y0:     k1 tmp[y] + k2 (tmp[y+1] + tmp[y-1])
        float s1 = k[1] * r[2][c]; s1 += k[2] * (r[3][c] + r[1][c]);        // row y0 + 1
        b[0][c] = s0; b[1][c] = s1;
    }
    float sum = 0;
    sum += b[0][0] + b[0][1] + b[1][0] + b[1][1];
    dst[(int64_t)dy * dw + dx] = sum * 0.25f;
}

// ---- cv::resize INTER_LINEAR (cn interleaved channels), optional post-scale ---------------------
__global__ void __launch_bounds__(256)
k_fb_resize_linear(const float *__restrict__ src, int sh, int sw, int cn, float *__restrict__ dst, int dh, int dw,
                   double scale_x, double scale_y, float post, int64_t bs_src, int64_t bs_dst)
{
    src += (int64_t)blockIdx.z * bs_src; dst += (int64_t)blockIdx.z * bs_dst;
    const int dx = blockIdx.x * 64 + threadIdx.x, dy = blockIdx.y * 4 + threadIdx.y;
    if (dx >= dw || dy >= dh) return;
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = tf_cvfloor(fx); fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = tf_cvfloor(fy); fy -= sy;
    const int sy0 = tf_clampi(sy, 0, sh - 1), sy1 = tf_clampi(sy + 1, 0, sh - 1);
    const int sx1 = sx + 1 < sw ? sx + 1 : sx;
    const float ax0 = 1.f - fx, ax1 = fx, ay0 = 1.f - fy, ay1 = fy;
    for (int c = 0; c < cn; c++) {
        float r0 = src[((int64_t)sy0 * sw + sx) * cn + c] * ax0 + src[((int64_t)sy0 * sw + sx1) * cn + c] * ax1;
        float r1 = src[((int64_t)sy1 * sw + sx) * cn + c] * ax0 + src[((int64_t)sy1 * sw + sx1) * cn + c] * ax1;
        float v = r0 * ay0 + r1 * ay1;
        dst[((int64_t)dy * dw + dx) * cn + c] = post == 1.f ? v : v * post;
    }
}

// the same for a two-channel image (the flow upsample between pyramid levels): one 8-byte access per tap and per
// output instead of two 4-byte ones; per channel the expressions of k_fb_resize_linear
__global__ void __launch_bounds__(256)
k_fb_resize_linear2(const float2 *__restrict__ src, int sh, int sw, float2 *__restrict__ dst, int dh, int dw,
                    double scale_x, double scale_y, float post, int64_t bs_src, int64_t bs_dst)
{
    src += (int64_t)blockIdx.z * bs_src; dst += (int64_t)blockIdx.z * bs_dst;
    const int dx = blockIdx.x * 64 + threadIdx.x, dy = blockIdx.y * 4 + threadIdx.y;
    if (dx >= dw || dy >= dh) return;
    float fx = (float)((dx + 0.5) * scale_x - 0.5);
    int sx = tf_cvfloor(fx); fx -= sx;
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
    float fy = (float)((dy + 0.5) * scale_y - 0.5);
    int sy = tf_cvfloor(fy); fy -= sy;
    const int sy0 = tf_clampi(sy, 0, sh - 1), sy1 = tf_clampi(sy + 1, 0, sh - 1);
    const int sx1 = sx + 1 < sw ? sx + 1 : sx;
    const float ax0 = 1.f - fx, ax1 = fx, ay0 = 1.f - fy, ay1 = fy;
    const float2 a = src[(int64_t)sy0 * sw + sx], b = src[(int64_t)sy0 * sw + sx1];
    const float2 c = src[(int64_t)sy1 * sw + sx], d = src[(int64_t)sy1 * sw + sx1];
    const float r0x = a.x * ax0 + b.x * ax1, r1x = c.x * ax0 + d.x * ax1;
    const float r0y = a.y * ax0 + b.y * ax1, r1y = c.y * ax0 + d.y * ax1;
    const float vx = r0x * ay0 + r1x * ay1, vy = r0y * ay0 + r1y * ay1;
    dst[(int64_t)dy * dw + dx] = post == 1.f ? make_float2(vx, vy) : make_float2(vx * post, vy * post);
}

// exact 2x decimation: OpenCV switches INTER_LINEAR to the INTER_AREA fast path
__global__ void __launch_bounds__(256)
k_fb_resize_area2(const float *__restrict__ src, int sw, float *__restrict__ dst, int dh, int dw, int64_t bs_src, int64_t bs_dst)
{
    src += (int64_t)blockIdx.z * bs_src; dst += (int64_t)blockIdx.z * bs_dst;
    const int dx = blockIdx.x * 64 + threadIdx.x, dy = blockIdx.y * 4 + threadIdx.y;
    if (dx >= dw || dy >= dh) return;
    const float *S = src + (int64_t)(2 * dy) * sw + 2 * dx;
    float sum = 0;
    sum += S[0] + S[1] + S[sw] + S[sw + 1];
    dst[(int64_t)dy * dw + dx] = sum * 0.25f;
}

// ---- GaussianBlur + INTER_LINEAR resize, evaluated only where the resize samples it -----------------
// For pyramid level k >= 1 OpenCV blurs the FULL-RES image and then reads just two blurred pixels
// per axis for every level pixel.  These two kernels compute exactly those values, with the same
// arithmetic order as the full blur followed by resize (row filter first, then symmetric column
// filter, then the bilinear combination):
//   pass 1  rowf[y][dx] = float2(row filter at column sx(dx), row filter at column sx(dx)+1), all H rows
//   pass 2  level(dy, dx) from the column filter at rows sy(dy), sy(dy)+1 of rowf
// Work drops from H*W*ksize*2 to about H*w*ksize*2 + h*w*ksize*4 multiply-adds per image.
struct FbResizeGeom { int sh, sw, dh, dw; double scale_x, scale_y; };

__device__ __forceinline__ void fb_resize_coord(int d, double scale, int slen, int &s0, float &f) {
    // cv::resize INTER_LINEAR x-coordinate rule (the y rule differs only in the clamping, done by the caller)
    f = (float)((d + 0.5) * scale - 0.5);
    s0 = tf_cvfloor(f); f -= s0;
}

template <typename TIn>
__global__ void __launch_bounds__(256)
k_fb_blur_rows_sampled(const TIn *__restrict__ src, FbResizeGeom g, const FbKernel kk, float2 *__restrict__ rowf, int64_t bs_src, int64_t bs_dst)
{
    src += (int64_t)blockIdx.z * bs_src; rowf += (int64_t)blockIdx.z * bs_dst;
    const int dx = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (dx >= g.dw || y >= g.sh) return;
    int sx; float fx;
    fb_resize_coord(dx, g.scale_x, g.sw, sx, fx);
    if (sx < 0) sx = 0;
    if (sx >= g.sw - 1) sx = g.sw - 1;
    const int sx1 = sx + 1 < g.sw ? sx + 1 : sx;
    const int ksize = kk.ksize, r = ksize >> 1, W = g.sw;
    const float *k = kk.k;
    const TIn *S = src + (int64_t)y * W;
    float2 out;
    if (sizeof(TIn) == 1 && ksize > 5 && sx1 == sx + 1 && sx - r >= 0 && sx1 + r + 4 < W) {
        // interior, 8-bit source (round 6): the ksize + 1 bytes of the two overlapping windows come as ALIGNED 32-bit words --
        // one load and one v_alignbyte per four taps instead of four byte loads (the kernel was bound by its memory
        // instructions, a byte each) -- and feed the same two accumulation chains in the same order.  (The last word may reach
        // up to three bytes past the window: still inside the row, hence `+ 4` in the condition.)
        const uint8_t *P = (const uint8_t *)S + sx - r;
        const unsigned sh = (unsigned)((uintptr_t)P & 3);
        const unsigned *wp = (const unsigned *)((uintptr_t)P & ~(uintptr_t)3);
        unsigned lo = wp[0], hi = wp[1];
        unsigned q = __builtin_amdgcn_alignbyte(hi, lo, sh);             // bytes P[0 .. 3]
        float s0 = k[0] * (float)(q & 0xffu), cur = (float)((q >> 8) & 0xffu);
        float s1 = k[0] * cur;
        // from here on the NEXT values P[i + 1 ..] are fetched in quads that start at byte sh + 2 of the word stream
        const unsigned sh2 = (sh + 2) & 3;
        int j = (int)((sh + 2) >> 2);                                      // word that holds P[2]
        lo = j ? hi : lo;
        int i = 1;
        for (; i + 3 < ksize; i += 4) {
            hi = wp[j + 1];
            q = __builtin_amdgcn_alignbyte(hi, lo, sh2);                   // bytes P[i + 1 .. i + 4]
            const float n0 = (float)(q & 0xffu), n1 = (float)((q >> 8) & 0xffu), n2 = (float)((q >> 16) & 0xffu), n3 = (float)(q >> 24);
            s0 += k[i] * cur;     s1 += k[i] * n0;
            s0 += k[i + 1] * n0;  s1 += k[i + 1] * n1;
            s0 += k[i + 2] * n1;  s1 += k[i + 2] * n2;
            s0 += k[i + 3] * n2;  s1 += k[i + 3] * n3;
            cur = n3; lo = hi; j++;
        }
        for (; i < ksize; i++) {
            const float nxt = (float)P[i + 1];
            s0 += k[i] * cur;
            s1 += k[i] * nxt;
            cur = nxt;
        }
        out.x = s0; out.y = s1;
    } else if (ksize > 5 && sx1 == sx + 1 && sx - r >= 0 && sx1 + r < W) {
        // interior: the two windows overlap in ksize - 1 pixels -> one pass over ksize + 1 pixels feeding two
        // independent accumulation chains (each chain keeps the reference's left-to-right order)
        const TIn *P = S + sx - r;
        float prev = (float)P[0];
        float s0 = k[0] * prev, s1;
        float cur = (float)P[1];
        s1 = k[0] * cur;
        // four pixels per trip: the loads of a trip are independent of its accumulations (which keep their order)
        int i = 1;
        for (; i + 7 < ksize; i += 8) {                        // wide kernels (pyramid levels >= 3): eight per trip
            float nn[8];
#pragma unroll
            for (int u = 0; u < 8; u++) nn[u] = (float)P[i + 1 + u];
            s0 += k[i] * cur; s1 += k[i] * nn[0];
#pragma unroll
            for (int u = 1; u < 8; u++) { s0 += k[i + u] * nn[u - 1]; s1 += k[i + u] * nn[u]; }
            cur = nn[7];
        }
        for (; i + 3 < ksize; i += 4) {
            const float n0 = (float)P[i + 1], n1 = (float)P[i + 2], n2 = (float)P[i + 3], n3 = (float)P[i + 4];
            s0 += k[i] * cur;     s1 += k[i] * n0;
            s0 += k[i + 1] * n0;  s1 += k[i + 1] * n1;
            s0 += k[i + 2] * n1;  s1 += k[i + 2] * n2;
            s0 += k[i + 3] * n2;  s1 += k[i + 3] * n3;
            cur = n3;
        }
        for (; i < ksize; i++) {
            const float nxt = (float)P[i + 1];
            s0 += k[i] * cur;
            s1 += k[i] * nxt;
            cur = nxt;
        }
        out.x = s0; out.y = s1;
    } else {
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int x = c ? sx1 : sx;
        float s;
        if (ksize == 3) {
            s = (float)S[x] * k[1] + ((float)S[fb_reflect101(x - 1, W)] + (float)S[fb_reflect101(x + 1, W)]) * k[0];
        } else if (ksize == 5) {
            s = (float)S[x] * k[2] + ((float)S[fb_reflect101(x - 1, W)] + (float)S[fb_reflect101(x + 1, W)]) * k[1]
              + ((float)S[fb_reflect101(x - 2, W)] + (float)S[fb_reflect101(x + 2, W)]) * k[0];
        } else {
            s = k[0] * (float)S[fb_reflect101(x - r, W)];
            if (x - r >= 0 && x + r < W) { for (int i = 1; i < ksize; i++) s += k[i] * (float)S[x - r + i]; }
            else { for (int i = 1; i < ksize; i++) s += k[i] * (float)S[fb_reflect101(x - r + i, W)]; }
        }
        if (c) out.y = s; else out.x = s;
    }
    }
    rowf[(int64_t)y * g.dw + dx] = out;
}

// The same pass with the source row segment of a workgroup staged in LDS (uint8 images, ksize > 5): 64 outputs x 4 rows
// per 256-thread workgroup read `64 * stride + ksize` source bytes per row ONCE, as aligned 32-bit words issued together
// (byte loads with the reflected border where the segment touches the image edge), instead of ksize + 1 byte loads per
// output in a dependent loop.  The LDS row is padded by one word per `unit` bytes (unit = the largest power of two <=
// stride) so that the lanes of a wave, which read bytes `stride` apart, fall into different banks.  Every output is the
// left-to-right sum of k_fb_blur_rows_sampled, bit for bit.
#define FBL_ROW_BYTES 4096            /* staged bytes per row the kernel accepts (before padding) */
#define FBL_ROW_WORDS (FBL_ROW_BYTES / 4 + FBL_ROW_BYTES / 32 + 8)
__global__ void __launch_bounds__(256)
k_fb_blur_rows_sampled_lds(const uint8_t *__restrict__ src, FbResizeGeom g, const FbKernel kk, float2 *__restrict__ rowf,
                           int64_t bs_src, int64_t bs_dst, int unit_shift, int word_env)
{
    __shared__ unsigned s_row[4][FBL_ROW_WORDS];
    src += (int64_t)blockIdx.z * bs_src; rowf += (int64_t)blockIdx.z * bs_dst;
    const int lane = threadIdx.x, ry = threadIdx.y;
    const int dx = blockIdx.x * 64 + lane, y0 = blockIdx.y * 4, y = y0 + ry;
    const int ksize = kk.ksize, r = ksize >> 1, W = g.sw;
    int sx; float fx;
    fb_resize_coord(min(dx, g.dw - 1), g.scale_x, g.sw, sx, fx);
    if (sx < 0) sx = 0;
    if (sx >= g.sw - 1) sx = g.sw - 1;
    const int sx1 = sx + 1 < g.sw ? sx + 1 : sx;
    // segment of this workgroup: source columns [lo, hi] (before reflection), lo rounded down to a word
    const int n_out = min(64, g.dw - blockIdx.x * 64);
    const int lo = (__shfl(sx, 0) - r) & ~3, hi = __shfl(sx1, n_out - 1) + r;
    const int span = hi - lo + 1;                                       // host guarantees span <= FBL_ROW_BYTES
    const int n_words = (span + 3) >> 2;
    const bool inside = lo >= 0 && lo + 4 * n_words <= W && ((W & 3) == 0) && ((bs_src & 3) == 0) && ((((uintptr_t)src) & 3) == 0);
    {
        // every wave stages the row it is going to read (round 6: the first form dealt the 4 x n_words words out over the 256
        // threads with `i / n_words` -- two runtime integer divisions per word and thread, 34 of them unrolled: more VALU work
        // than the 95-tap sums themselves, SQ_INSTS_VALU said); all loads of a thread first, the LDS stores afterwards
        constexpr int NV = (FBL_ROW_BYTES / 4 + 1 + 63) / 64;
        unsigned v[NV];
        const int yy = min(y0 + ry, g.sh - 1);
        const uint8_t *S = src + (int64_t)yy * W;
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const int wd = lane + 64 * j;
            v[j] = 0;
            if (wd < n_words) {
                if (inside) v[j] = *(const unsigned *)(S + lo + 4 * wd);
                else {
                    const int c = lo + 4 * wd;
                    v[j] = (unsigned)S[fb_reflect101(c, W)] | ((unsigned)S[fb_reflect101(c + 1, W)] << 8) |
                           ((unsigned)S[fb_reflect101(c + 2, W)] << 16) | ((unsigned)S[fb_reflect101(c + 3, W)] << 24);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < NV; j++) {
            const int wd = lane + 64 * j;
            if (wd < n_words) s_row[ry][wd + ((4 * wd) >> unit_shift)] = v[j];
        }
    }
    __syncthreads();
    if (dx >= g.dw || y >= g.sh) return;
    const float *k = kk.k;
    const uint8_t *L = (const uint8_t *)s_row[ry];
    // the word reads may touch the word after the last staged one (its value is never used: alignbyte drops it for the bytes
    // that matter, but the index must be inside the LDS row); TF_FB_BLUR_BYTES=1: the byte reads of rounds 2 - 5
    const bool word_reads = word_env && (((span + 3) >> 2) + 2 + (((span + 3) & ~3) >> unit_shift) < FBL_ROW_WORDS);
    // byte a of the segment lives at a + 4 * (a >> unit_shift)
    auto at = [&](int a) { return (float)L[a + ((a >> unit_shift) << 2)]; };
    const int b0 = sx - r - lo, b1 = sx1 - r - lo;
    float s0, s1;
    static_assert(FBL_ROW_WORDS > 0, "");
    if (b1 == b0 + 1 && (word_reads)) {
        // (round 6) the same two chains fed from 32-bit LDS reads: byte a of the segment lives in word (a >> 2) + (a >> unit_shift),
        // a quad that starts at an arbitrary byte is v_alignbyte of two neighbouring words -- one LDS read per four taps
        // instead of four byte reads (the kernel was bound by its LDS instructions)
        const unsigned *Lw = s_row[ry];
        auto word = [&](int a4) { return Lw[(a4 >> 2) + (a4 >> unit_shift)]; };     // a4: a multiple of four
        const int a0 = b0 & ~3;
        const unsigned sh = (unsigned)(b0 & 3);
        unsigned lo = word(a0), hi = word(a0 + 4);
        unsigned q = __builtin_amdgcn_alignbyte(hi, lo, sh);               // bytes b0 .. b0 + 3
        float cur = (float)((q >> 8) & 0xffu);
        s0 = k[0] * (float)(q & 0xffu); s1 = k[0] * cur;
        const unsigned sh2 = (sh + 2) & 3;
        int a = a0 + (int)((sh + 2) & ~3u);                                  // word that holds byte b0 + 2
        lo = (sh + 2) >= 4 ? hi : lo;
        int i = 1;
        for (; i + 3 < ksize; i += 4) {
            hi = word(a + 4);
            q = __builtin_amdgcn_alignbyte(hi, lo, sh2);                     // bytes b0 + i + 1 .. b0 + i + 4
            const float n0 = (float)(q & 0xffu), n1 = (float)((q >> 8) & 0xffu), n2 = (float)((q >> 16) & 0xffu), n3 = (float)(q >> 24);
            s0 += k[i] * cur;     s1 += k[i] * n0;
            s0 += k[i + 1] * n0;  s1 += k[i + 1] * n1;
            s0 += k[i + 2] * n1;  s1 += k[i + 2] * n2;
            s0 += k[i + 3] * n2;  s1 += k[i + 3] * n3;
            cur = n3; lo = hi; a += 4;
        }
        for (; i < ksize; i++) { const float nxt = at(b0 + i + 1); s0 += k[i] * cur; s1 += k[i] * nxt; cur = nxt; }
    } else if (b1 == b0 + 1) {
        float cur = at(b0 + 1);
        s0 = k[0] * at(b0); s1 = k[0] * cur;
        int i = 1;
        for (; i + 7 < ksize; i += 8) {
            float nn[8];
#pragma unroll
            for (int u = 0; u < 8; u++) nn[u] = at(b0 + i + 1 + u);
            s0 += k[i] * cur; s1 += k[i] * nn[0];
#pragma unroll
            for (int u = 1; u < 8; u++) { s0 += k[i + u] * nn[u - 1]; s1 += k[i + u] * nn[u]; }
            cur = nn[7];
        }
        for (; i < ksize; i++) { const float nxt = at(b0 + i + 1); s0 += k[i] * cur; s1 += k[i] * nxt; cur = nxt; }
    } else {
        s0 = k[0] * at(b0); s1 = k[0] * at(b1);
        for (int i = 1; i < ksize; i++) { s0 += k[i] * at(b0 + i); s1 += k[i] * at(b1 + i); }
    }
    rowf[(int64_t)y * g.dw + dx] = make_float2(s0, s1);
}

__global__ void __launch_bounds__(256)
k_fb_blur_cols_resize(const float2 *__restrict__ rowf, FbResizeGeom g, const FbKernel kk, float *__restrict__ dst, int64_t bs_src, int64_t bs_dst)
{
    rowf += (int64_t)blockIdx.z * bs_src; dst += (int64_t)blockIdx.z * bs_dst;
    const int dx = blockIdx.x * 64 + threadIdx.x, dy = blockIdx.y * 4 + threadIdx.y;
    if (dx >= g.dw || dy >= g.dh) return;
    int sx, sy; float fx, fy;
    fb_resize_coord(dx, g.scale_x, g.sw, sx, fx);
    if (sx < 0) { fx = 0; sx = 0; }
    if (sx >= g.sw - 1) { fx = 0; sx = g.sw - 1; }
    fb_resize_coord(dy, g.scale_y, g.sh, sy, fy);
    const int H = g.sh, w = g.dw, r = kk.ksize >> 1;
    const float *k = kk.k;
    const int rows[2] = {tf_clampi(sy, 0, H - 1), tf_clampi(sy + 1, 0, H - 1)};
    float b[2][2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const int y = rows[q];
        const float2 c0 = rowf[(int64_t)y * w + dx];
        float s0 = k[r] * c0.x, s1 = k[r] * c0.y;
        if (y - r >= 0 && y + r < H) {
            int i = 1;
            for (; i + 3 <= r; i += 4) {                       // loads of a trip first, accumulation order unchanged
                float2 p[4], m[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { p[u] = rowf[(int64_t)(y + i + u) * w + dx]; m[u] = rowf[(int64_t)(y - i - u) * w + dx]; }
#pragma unroll
                for (int u = 0; u < 4; u++) { s0 += k[r + i + u] * (p[u].x + m[u].x); s1 += k[r + i + u] * (p[u].y + m[u].y); }
            }
            for (; i <= r; i++) {
                const float2 p = rowf[(int64_t)(y + i) * w + dx], m = rowf[(int64_t)(y - i) * w + dx];
                s0 += k[r + i] * (p.x + m.x); s1 += k[r + i] * (p.y + m.y);
            }
        } else {
            for (int i = 1; i <= r; i++) {
                const float2 p = rowf[(int64_t)fb_reflect101(y + i, H) * w + dx], m = rowf[(int64_t)fb_reflect101(y - i, H) * w + dx];
                s0 += k[r + i] * (p.x + m.x); s1 += k[r + i] * (p.y + m.y);
            }
        }
        b[q][0] = s0; b[q][1] = s1;
    }
    const float ax0 = 1.f - fx, ax1 = fx, ay0 = 1.f - fy, ay1 = fy;
    const float r0 = b[0][0] * ax0 + b[0][1] * ax1;
    const float r1 = b[1][0] * ax0 + b[1][1] * ax1;
    dst[(int64_t)dy * g.dw + dx] = r0 * ay0 + r1 * ay1;
}

// ---- polynomial expansion ------------------------------------------------------------------------
struct FbPoly { int n; float g[FB_MAX_POLY_N + 1], xg[FB_MAX_POLY_N + 1], xxg[FB_MAX_POLY_N + 1]; double ig11, ig03, ig33, ig55; };

// Fused vertical + horizontal pass on an LDS tile: 64 x 16 outputs per 256-thread workgroup, the
// (64 + 2n) x (16 + 2n) image tile is staged once (replicate borders), the three vertical moments
// t0 = sum g (I_up + I_dn), t1 = sum xg (I_dn - I_up), t2 = sum xxg (I_up + I_dn) are kept in LDS for all
// 64 + 2n columns, the horizontal pass accumulates in double exactly like OpenCV and writes R
// (float4 {y, x, yy, xx} + float {xy}).  HBM traffic = 4 B read + 20 B written per level pixel.
#define FBP_W 64
#define FBP_H 16
// development switch: TF_FB_POLYEXP_GENERIC=1 routes polyN = 5 through the generic kernel too (same results)
static bool fb_polyexp_generic() { static const bool v = getenv("TF_FB_POLYEXP_GENERIC") != nullptr; return v; }
#define FBP_MAXN FB_MAX_POLY_N
__global__ void __launch_bounds__(256)
k_fb_polyexp(const float *__restrict__ I, int H, int W, FbPoly pp, float *__restrict__ R, int64_t plane,
             int64_t bs_I, int64_t bs_R)
{
    __shared__ float tI[(FBP_H + 2 * FBP_MAXN) * (FBP_W + 2 * FBP_MAXN)];
    __shared__ float tT[3][FBP_H * (FBP_W + 2 * FBP_MAXN)];
    I += (int64_t)blockIdx.z * bs_I; R += (int64_t)blockIdx.z * bs_R;
    const int n = pp.n, tw = FBP_W + 2 * n, th = FBP_H + 2 * n;
    const int bx = blockIdx.x * FBP_W, by = blockIdx.y * FBP_H;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    for (int i = tid; i < tw * th; i += 256) {
        const int ty = i / tw, tx = i - ty * tw;
        tI[i] = I[(int64_t)tf_clampi(by + ty - n, 0, H - 1) * W + tf_clampi(bx + tx - n, 0, W - 1)];
    }
    __syncthreads();
    for (int i = tid; i < tw * FBP_H; i += 256) {
        const int oy = i / tw, tx = i - oy * tw;
        const int y = by + oy;                                   // image row of this output row
        const float *col = tI + (oy + n) * tw + tx;
        float r0 = col[0] * pp.g[0], r1 = 0.f, r2 = 0.f;
        for (int k = 1; k <= n; k++) {
            // rows are clamped in IMAGE space (replicate): the tile already holds clamp(y +- k)
            const float s0 = col[-k * tw], s1 = col[k * tw];
            const float p = s0 + s1;
            r0 = r0 + pp.g[k] * p;
            r1 = r1 + pp.xg[k] * (s1 - s0);
            r2 = r2 + pp.xxg[k] * p;
        }
        (void)y;
        tT[0][i] = r0; tT[1][i] = r1; tT[2][i] = r2;
    }
    __syncthreads();
    const int x = bx + threadIdx.x;
    if (x >= W) return;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int oy = threadIdx.y * 4 + r, y = by + oy;
        if (y >= H) continue;
        const float *a = tT[0] + oy * tw + threadIdx.x + n, *b = tT[1] + oy * tw + threadIdx.x + n, *c = tT[2] + oy * tw + threadIdx.x + n;
        float g0 = pp.g[0];
        double b1 = a[0] * g0, b2 = 0, b3 = b[0] * g0, b4 = 0, b5 = c[0] * g0, b6 = 0;
        for (int k = 1; k <= n; k++) {
            const double tg = a[k] + a[-k];
            g0 = pp.g[k];
            b1 += tg * g0; b4 += tg * pp.xxg[k];
            b2 += (a[k] - a[-k]) * pp.xg[k];
            b3 += (b[k] + b[-k]) * g0;
            b6 += (b[k] - b[-k]) * pp.xg[k];
            b5 += (c[k] + c[-k]) * g0;
        }
        const int64_t o = (int64_t)y * W + x;
        ((float4 *)R)[o] = make_float4((float)(b3 * pp.ig11), (float)(b2 * pp.ig11),
                                       (float)(b1 * pp.ig03 + b5 * pp.ig33), (float)(b1 * pp.ig03 + b4 * pp.ig33));
        R[4 * plane + o] = (float)(b6 * pp.ig55);
    }
}

// The same kernel for polyN = 5 (the default), register-blocked: a work item of the vertical pass owns one tile column
// for FOUR output rows (14 tile values feed 4 x 11 taps), a thread of the horizontal pass owns FOUR consecutive output
// columns of one row (14 values of each moment, fetched as three 16-byte and one 8-byte LDS reads from rows padded to
// a multiple of four floats).  LDS reads per output drop from 11 + 33 to 3.5 + 10.5; every output is still formed by the
// expressions of k_fb_polyexp in the same order (bit-identical, tests compare the two).
// Round 6: the tile HEIGHT is a template parameter (16 = the default; 32: the halo rows, the vertical pass' 10 extra rows and
// the two barriers are paid once per 32 output rows -- image reads 1.88 x -> 1.52 x the tile, 3 instead of 5 workgroups of LDS
// per CU: 3 % faster with nothing beside it, 30 % slower in the timed region of bench.py, where the floods' kernels share the
// CUs -- TF_FB_POLYEXP_TH=32 opts in, profiles/round6_polyexp_notes.txt has the measurements).
#define FBP5_TS (FBP_W + 2 * 5 + 2)      /* row stride of the moment tiles: 76 floats, 16-byte aligned rows */
template <int FBP5_H>
__global__ void __launch_bounds__(256)
k_fb_polyexp5(const float *__restrict__ I, int H, int W, FbPoly pp, float *__restrict__ R, int64_t plane,
              int64_t bs_I, int64_t bs_R)
{
    constexpr int n = 5, tw = FBP_W + 2 * n, th = FBP5_H + 2 * n;
    __shared__ float tI[th * tw];
    __shared__ __attribute__((aligned(16))) float tT[3][FBP5_H * FBP5_TS];
    I += (int64_t)blockIdx.z * bs_I; R += (int64_t)blockIdx.z * bs_R;
    const int bx = blockIdx.x * FBP_W, by = blockIdx.y * FBP5_H;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    {
        // all loads of a thread first, the LDS stores afterwards: a store right behind its load makes every trip of the
        // loop a full memory round trip (the rolled form ran eight of them back to back)
        constexpr int NL = (tw * th + 255) / 256;
        float v[NL];
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const int i = tid + 256 * j, ty = i / tw, tx = i - ty * tw;
            v[j] = I[(int64_t)tf_clampi(by + ty - n, 0, H - 1) * W + tf_clampi(bx + tx - n, 0, W - 1)];   // ty past the tile: clamped, unused
        }
#pragma unroll
        for (int j = 0; j < NL; j++) { const int i = tid + 256 * j; if (i < tw * th) tI[i] = v[j]; }
    }
    __syncthreads();
    float g[n + 1], xg[n + 1], xxg[n + 1];
#pragma unroll
    for (int k = 0; k <= n; k++) { g[k] = pp.g[k]; xg[k] = pp.xg[k]; xxg[k] = pp.xxg[k]; }
    for (int i = tid; i < tw * (FBP5_H / 4); i += 256) {
        const int rg = i / tw, tx = i - rg * tw;
        float v[14];
#pragma unroll
        for (int j = 0; j < 14; j++) v[j] = tI[(4 * rg + j) * tw + tx];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float r0 = v[r + n] * g[0], r1 = 0.f, r2 = 0.f;
#pragma unroll
            for (int k = 1; k <= n; k++) {
                const float s0 = v[r + n - k], s1 = v[r + n + k];
                const float p = s0 + s1;
                r0 = r0 + g[k] * p;
                r1 = r1 + xg[k] * (s1 - s0);
                r2 = r2 + xxg[k] * p;
            }
            const int o = (4 * rg + r) * FBP5_TS + tx;
            tT[0][o] = r0; tT[1][o] = r1; tT[2][o] = r2;
        }
    }
    __syncthreads();
    const int x4 = (tid & 15) * 4;
    if (bx + x4 >= W) return;
#pragma unroll
    for (int part = 0; part < FBP5_H / 16; part++) {
    const int oy = (tid >> 4) + 16 * part, y = by + oy;
    if (y >= H) break;
    float a[14], b[14], c[14];
    {
        const float *pa = tT[0] + oy * FBP5_TS + x4, *pb = tT[1] + oy * FBP5_TS + x4, *pc = tT[2] + oy * FBP5_TS + x4;
#pragma unroll
        for (int q = 0; q < 3; q++) {
            const float4 va = ((const float4 *)pa)[q], vb = ((const float4 *)pb)[q], vc = ((const float4 *)pc)[q];
            a[4 * q] = va.x; a[4 * q + 1] = va.y; a[4 * q + 2] = va.z; a[4 * q + 3] = va.w;
            b[4 * q] = vb.x; b[4 * q + 1] = vb.y; b[4 * q + 2] = vb.z; b[4 * q + 3] = vb.w;
            c[4 * q] = vc.x; c[4 * q + 1] = vc.y; c[4 * q + 2] = vc.z; c[4 * q + 3] = vc.w;
        }
        const float2 ea = ((const float2 *)pa)[6], eb = ((const float2 *)pb)[6], ec = ((const float2 *)pc)[6];
        a[12] = ea.x; a[13] = ea.y; b[12] = eb.x; b[13] = eb.y; c[12] = ec.x; c[13] = ec.y;
    }
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const int x = bx + x4 + e;
        if (x >= W) break;
        const int m = e + n;                      // centre of this output inside the 14-value windows
        float g0 = g[0];
        double b1 = a[m] * g0, b2 = 0, b3 = b[m] * g0, b4 = 0, b5 = c[m] * g0, b6 = 0;
#pragma unroll
        for (int k = 1; k <= n; k++) {
            const double tg = a[m + k] + a[m - k];
            g0 = g[k];
            // tg and the coefficients are float values: their product is exact in double, so ONE fused multiply-add returns
            // exactly the value of the reference's multiply followed by its add
            b1 = __fma_rn(tg, (double)g0, b1); b4 = __fma_rn(tg, (double)xxg[k], b4);
            b2 += (a[m + k] - a[m - k]) * xg[k];
            b3 += (b[m + k] + b[m - k]) * g0;
            b6 += (b[m + k] - b[m - k]) * xg[k];
            b5 += (c[m + k] + c[m - k]) * g0;
        }
        const int64_t o = (int64_t)y * W + x;
        ((float4 *)R)[o] = make_float4((float)(b3 * pp.ig11), (float)(b2 * pp.ig11),
                                       (float)(b1 * pp.ig03 + b5 * pp.ig33), (float)(b1 * pp.ig03 + b4 * pp.ig33));
        R[4 * plane + o] = (float)(b6 * pp.ig55);
    }
    }
}

static void fb_launch_polyexp(const float *I, int h, int w, const FbPoly &pp, float *R, int64_t plane, int64_t bs_I, int64_t bs_R, int B, hipStream_t s)
{
    static const int th_env = getenv("TF_FB_POLYEXP_TH") ? atoi(getenv("TF_FB_POLYEXP_TH")) : 0;      // development switch: 16 / 32
    const dim3 block(64, 4);
    if (pp.n != 5 || fb_polyexp_generic())
        hipLaunchKernelGGL(k_fb_polyexp, dim3((w + FBP_W - 1) / FBP_W, (h + FBP_H - 1) / FBP_H, B), block, 0, s, I, h, w, pp, R, plane, bs_I, bs_R);
    else if (th_env == 32)                                   // 64 x 32 tiles: 3 % faster alone, 30 % SLOWER beside the floods (round 6, measured): opt-in
        hipLaunchKernelGGL(k_fb_polyexp5<32>, dim3((w + FBP_W - 1) / FBP_W, (h + 31) / 32, B), block, 0, s, I, h, w, pp, R, plane, bs_I, bs_R);
    else
        hipLaunchKernelGGL(k_fb_polyexp5<16>, dim3((w + FBP_W - 1) / FBP_W, (h + 15) / 16, B), block, 0, s, I, h, w, pp, R, plane, bs_I, bs_R);
}

// ---- FarnebackUpdateMatrices ---------------------------------------------------------------------
__device__ __forceinline__ void fb_matrix_at(const float *__restrict__ R0, const float *__restrict__ R1,
                                             const float *__restrict__ flow, int H, int W, int64_t plane,
                                             int x, int y, float &m0, float &m1, float &m2, float &m3, float &m4)
{
    const int64_t o = (int64_t)y * W + x;
    const float2 fl = ((const float2 *)flow)[o];
    const float dx = fl.x, dy = fl.y;
    float fx = x + dx, fy = y + dy;
    const int x1 = tf_cvfloor(fx), y1 = tf_cvfloor(fy);
    float r2, r3, r4, r5, r6;
    fx -= x1; fy -= y1;
    const float4 q0 = ((const float4 *)R0)[o];
    const float q04 = R0[4 * plane + o];
    if ((unsigned)x1 < (unsigned)(W - 1) && (unsigned)y1 < (unsigned)(H - 1)) {
        const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
        const int64_t q = (int64_t)y1 * W + x1;
        const float4 *P4 = (const float4 *)R1 + q;
        const float *P1 = R1 + 4 * plane + q;
        const float4 c00 = P4[0], c01 = P4[1], c10 = P4[W], c11 = P4[W + 1];
        const float e00 = P1[0], e01 = P1[1], e10 = P1[W], e11 = P1[W + 1];
        r2 = a00 * c00.x + a01 * c01.x + a10 * c10.x + a11 * c11.x;
        r3 = a00 * c00.y + a01 * c01.y + a10 * c10.y + a11 * c11.y;
        r4 = a00 * c00.z + a01 * c01.z + a10 * c10.z + a11 * c11.z;
        r5 = a00 * c00.w + a01 * c01.w + a10 * c10.w + a11 * c11.w;
        r6 = a00 * e00 + a01 * e01 + a10 * e10 + a11 * e11;
        r4 = (q0.z + r4) * 0.5f;
        r5 = (q0.w + r5) * 0.5f;
        r6 = (q04 + r6) * 0.25f;
    } else {
        r2 = r3 = 0.f;
        r4 = q0.z; r5 = q0.w; r6 = q04 * 0.5f;
    }
    r2 = (q0.x - r2) * 0.5f;
    r3 = (q0.y - r3) * 0.5f;
    r2 += r4 * dy + r6 * dx;
    r3 += r6 * dy + r5 * dx;
    if ((unsigned)(x - 5) >= (unsigned)(W - 10) || (unsigned)(y - 5) >= (unsigned)(H - 10)) {
        const float border[5] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
        const float scale = (x < 5 ? border[x] : 1.f) * (x >= W - 5 ? border[W - x - 1] : 1.f) *
                            (y < 5 ? border[y] : 1.f) * (y >= H - 5 ? border[H - y - 1] : 1.f);
        r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
    }
    m0 = r4 * r4 + r6 * r6;
    m1 = (r4 + r5) * r6;
    m2 = r5 * r5 + r6 * r6;
    m3 = r4 * r2 + r6 * r3;
    m4 = r6 * r2 + r5 * r3;
}

__global__ void __launch_bounds__(256)
k_fb_update_matrices(const float *__restrict__ R0, const float *__restrict__ R1, const float *__restrict__ flow,
                     int H, int W, int64_t plane, float *__restrict__ M)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H) return;
    float m0, m1, m2, m3, m4;
    fb_matrix_at(R0, R1, flow, H, W, plane, x, y, m0, m1, m2, m3, m4);
    const int64_t o = (int64_t)y * W + x;
    M[o] = m0; M[plane + o] = m1; M[2 * plane + o] = m2; M[3 * plane + o] = m3; M[4 * plane + o] = m4;
}

// ---- fused iteration: UpdateMatrices -> 13x13 box sums (double) -> 2x2 solve -----------------------
// One launch = one Farnebaeck iteration of BOTH directions for a batch of frame pairs.  A workgroup owns a strip of
// FBI_OW = 116 output columns (+ 6 halo columns each side) over the WHOLE height, with FBI_T = 128 threads (two waves)
// per direction: 256 threads when both directions run (round 3), so that they share the R rows they both read.
// Thread j walks DOWN its column: at every row it evaluates M = UpdateMatrices(R0, R1, flow_old) in
// registers, keeps the last 13 rows of M in a register ring and the running 13-row column sums in
// double (OpenCV's recurrence, float-rounded differences included); the five column sums go to an LDS row, from
// which thread t takes (row t / 29, quad t % 29): four 13-wide window sums that share their ten middle terms, four
// 2x2 solves, flow_new.  The 5-plane matrix M never exists in HBM: per level pixel
// the kernel moves R0 (20 B) + R1 (20 B) + flow in (8 B) + flow out (8 B).
// The per-row evaluation is BRANCH-FREE (out-of-image gathers read a valid dummy address and are
// discarded by selects) so that the loads of FBI_NB rows are in flight together, and the flow of the next
// row group is fetched before the LDS phase of the current one.  Occupancy: two waves per SIMD = two four-wave workgroups per CU.
#define FBI_M 6
#define FBI_WIN (2 * FBI_M + 1)
#ifndef FBI_T
#define FBI_T 128                   // threads = evaluated columns per workgroup
#endif
#define FBI_OW (FBI_T - 2 * FBI_M)
// R[img] / fin[q] / fout[q] are the pointers of batch item 0; item b adds b * the matching stride
struct FbIterArgs {
    const float *R[2]; const float *fin[2]; float *fout[2]; int dir[2]; int nd, nx; int64_t bs_R, bs_fin[2], bs_fout[2];
    // sequential row sums (k_fb_iter): hand-over words of batch item 0 (item b adds b * bs_hand words), this launch's tag, its ticket counter
    unsigned long long *hand; int64_t bs_hand; unsigned epoch; int *ticket;
    int abl;                        // timing aid (TF_FBI_SEQ_ABLATE): 1 no wait for the left neighbour, 2 no chain, 4 no solve, 16 hand-over without the chain (all wrong flows); 64 no hand-over stores (the chains right of strip 0 starve: tests of TF_ESTARVED); 8 count waiting chains, 128 no priority for the chain wave, 256 priority 1 instead of 3 (right flows)
    int xcd_lists;                  // 1: one ticket list per XCD (pairs dealt round robin), 0: one list
    int nq;                         // 1: a workgroup holds all directions of its strip; 2: one direction per workgroup, directions take tickets
    int nb, nxg, slack_rows;        // pairs in the launch; strips per column group (ticket order); rows a strip lets its left neighbour get ahead before it starts
    int *starved;                   // host-visible word (fb_starved_word): set by a chain whose left neighbour's words never arrived -- its row is NaN, the call is an error
    int poll_limit;                 // polls before such a chain gives up (1 << 22: several seconds; TF_FBI_POLL_LIMIT: tests)
};
#define FBI_G 5                     // rows per group (13 = 4 + 4 + 5)
#ifndef FBI_NB
#define FBI_NB 2                    // rows whose gathers are in flight together.  Round 3, four-wave workgroups: 2 / 3 / 4 / 5 rows ->
#endif                              // 16.2 / 16.8 / 18.9 / 22.4 ms per level-0 launch of 21 pairs (4: 10 spilled registers, 5: more); round 2's
                                    // two-wave workgroups did not care (116.3 / 116.1 / 116.3 ms per step for 2 / 4 / 5)

// Addressing: per-thread 64-bit byte offsets from per-item base pointers.  (A variant with wave-uniform bases + 32-bit
// offsets, the global_load "saddr + voffset" form, saves 7 % of the VALU instructions and runs 12 % SLOWER at 5424^2:
// measured, not pursued.)
typedef int64_t fb_off_t;

struct FbIterCtx {
    const char *R0, *R0e, *fin; char *fout;                      // R*: float4 plane {y, x, yy, xx};  R*e: float plane {xy}
    // the four corners of the bilinear gather share ONE per-thread offset: the corner displacement (+1 element,
    // +1 row, both) is folded into four wave-uniform base pointers
    const char *R1c[4], *R1ec[4];
    int H, W; int j, dj, tg, tq, x_strip, xc, y0, y1; float xscale; bool xborder;
    bool strip_xborder;             // wave-uniform: some column of this strip lies in OpenCV's border band (xborder of any thread)
    // sequential row sums: strip index / count, the (row of the group, channel) this lane scans, hand-over slots of the left
    // neighbour (read) and of this strip (written), the launch's tag
    int sx, nx, sr, sch; const unsigned long long *hin; unsigned long long *hout; unsigned epoch; int abl; int *spins; int *starved; int poll_limit;
    int rsr, rsch; bool full;       // (row, channel) of a right-half chain lane (j - 32); a full strip of FBI_OW output columns
};

struct FbTaps { float4 q0, c00, c01, c10, c11; float2 e0, e1; float q04, dx, dy; };

template <typename T, typename OFF>
__device__ __forceinline__ T fb_ld(const char *base, OFF byte_off) { return *(const T *)(base + byte_off); }

__device__ __forceinline__ float2 fb_iter_flow_at(const FbIterCtx &c, int s)
{
    typedef fb_off_t off_t;
    const off_t o = (off_t)tf_clampi(s, 0, c.H - 1) * (off_t)c.W + (off_t)c.xc;
    return fb_ld<float2>(c.fin, o * 8);
}

// issue every load of one M evaluation (same arithmetic as fb_matrix_at, branch-free)
template <int ABL = 0>
__device__ __forceinline__ void fb_taps_load(const FbIterCtx &c, int s, float2 fl, FbTaps &t)
{
    typedef fb_off_t off_t;
    const int y = tf_clampi(s, 0, c.H - 1);
    const off_t o = ABL == 2 ? (off_t)c.xc : (off_t)y * (off_t)c.W + (off_t)c.xc;      // ABL 2: every gather from row 0
    t.dx = fl.x; t.dy = fl.y;
    const float fx = c.xc + t.dx, fy = y + t.dy;
    const int x1 = tf_cvfloor(fx), y1 = tf_cvfloor(fy);
    const bool inb = (unsigned)x1 < (unsigned)(c.W - 1) && (unsigned)y1 < (unsigned)(c.H - 1);
    // out of the image: the patch at element 0 is read instead (always valid, see the base pointers) and discarded
    const off_t q = ABL == 2 ? (off_t)(inb ? x1 : 0) : (inb ? (off_t)y1 * (off_t)c.W + (off_t)x1 : (off_t)0);
    const off_t q16 = q * 16, q4 = q * 4;
    t.q0 = fb_ld<float4>(c.R0, o * 16);
    t.q04 = fb_ld<float>(c.R0e, o * 4);
    t.c00 = fb_ld<float4>(c.R1c[0], q16); t.c01 = fb_ld<float4>(c.R1c[1], q16);
    t.c10 = fb_ld<float4>(c.R1c[2], q16); t.c11 = fb_ld<float4>(c.R1c[3], q16);
    t.e0.x = fb_ld<float>(c.R1ec[0], q4); t.e0.y = fb_ld<float>(c.R1ec[1], q4);
    t.e1.x = fb_ld<float>(c.R1ec[2], q4); t.e1.y = fb_ld<float>(c.R1ec[3], q4);
}

__device__ __forceinline__ void fb_taps_eval(const FbIterCtx &c, int s, const FbTaps &t, float (&m)[5])
{
    // the (cheap) coordinate arithmetic is redone here instead of being carried in registers while loads fly
    const int y = tf_clampi(s, 0, c.H - 1);
    const float dx = t.dx, dy = t.dy;
    float fx = c.xc + dx, fy = y + dy;
    const int x1 = tf_cvfloor(fx), y1 = tf_cvfloor(fy);
    fx -= x1; fy -= y1;
    const bool inb = (unsigned)x1 < (unsigned)(c.W - 1) && (unsigned)y1 < (unsigned)(c.H - 1);
    const float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
    float r2 = a00 * t.c00.x + a01 * t.c01.x + a10 * t.c10.x + a11 * t.c11.x;
    float r3 = a00 * t.c00.y + a01 * t.c01.y + a10 * t.c10.y + a11 * t.c11.y;
    float r4 = a00 * t.c00.z + a01 * t.c01.z + a10 * t.c10.z + a11 * t.c11.z;
    float r5 = a00 * t.c00.w + a01 * t.c01.w + a10 * t.c10.w + a11 * t.c11.w;
    float r6 = a00 * t.e0.x + a01 * t.e0.y + a10 * t.e1.x + a11 * t.e1.y;
    r4 = (t.q0.z + r4) * 0.5f;
    r5 = (t.q0.w + r5) * 0.5f;
    r6 = (t.q04 + r6) * 0.25f;
    r2 = inb ? r2 : 0.f; r3 = inb ? r3 : 0.f;
    r4 = inb ? r4 : t.q0.z; r5 = inb ? r5 : t.q0.w; r6 = inb ? r6 : t.q04 * 0.5f;
    r2 = (t.q0.x - r2) * 0.5f;
    r3 = (t.q0.y - r3) * 0.5f;
    r2 += r4 * dy + r6 * dx;
    r3 += r6 * dy + r5 * dx;
    // border attenuation: (x factors, fixed per thread) * (y factors, uniform).  It is applied under OpenCV's own
    // test, (unsigned)(x - 5) >= (unsigned)(W - 10) || (unsigned)(y - 5) >= (unsigned)(H - 10), whose unsigned
    // wrap-around leaves some border columns / rows of images narrower than 10 pixels UNscaled: reproduced as is
    // (round 5: the test is wave-uniform -- interior rows of interior strips, i.e. nearly every evaluation, skip the scale's
    // selects and five multiplications by exactly 1, which change no bit)
    if (c.strip_xborder || (unsigned)(y - 5) >= (unsigned)(c.H - 10)) {
        const int yb = c.H - 1 - y;
        const float b0 = y == 0 ? 0.14f : (y == 1 ? 0.14f : 0.4472f);
        const float b1 = yb == 0 ? 0.14f : (yb == 1 ? 0.14f : 0.4472f);
        const bool in_border = c.xborder || (unsigned)(y - 5) >= (unsigned)(c.H - 10);
        const float scale = in_border ? c.xscale * (y < 5 ? b0 : 1.f) * (yb < 5 ? b1 : 1.f) : 1.f;
        r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
    }
    m[0] = r4 * r4 + r6 * r6;
    m[1] = (r4 + r5) * r6;
    m[2] = r5 * r5 + r6 * r6;
    m[3] = r4 * r2 + r6 * r3;
    m[4] = r6 * r2 + r5 * r3;
}

// LDS row of column sums: column i lives at i + i/4 (one pad per four) so that both access patterns are
// conflict-free: the vertical phase writes consecutive i, the horizontal phase reads i = 4 q + k (stride 5).
#define FBI_VS (FBI_T + FBI_T / 4)
#define FBI_Q (FBI_OW / 4)          // 29 quads of 4 outputs per strip row

// horizontal phase for one (row g, quad q): 16 column sums per channel -> four 13-wide window sums ->
// four 2x2 solves.  The four windows share the ten middle terms.
__device__ __forceinline__ void fb_iter_quad(const FbIterCtx &c, int yo, int g, int q, const double *vrow)
{
    typedef fb_off_t off_t;
    const int x0 = c.x_strip + 4 * q;
    if (yo < c.y0 || yo >= c.y1 || x0 >= c.W) return;
    const double *base = vrow + (g * 5) * FBI_VS + 5 * q;
    double sum[5][4];
#pragma unroll
    for (int ch = 0; ch < 5; ch++) {
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = base[ch * FBI_VS + k + (k >> 2)];
        // balanced tree over the ten shared terms
        const double T = (((v[3] + v[4]) + (v[5] + v[6])) + ((v[7] + v[8]) + (v[9] + v[10]))) + (v[11] + v[12]);
        const double lo = v[1] + v[2], hi = v[13] + v[14];
        sum[ch][0] = v[0] + lo + T;
        sum[ch][1] = lo + T + v[13];
        sum[ch][2] = v[2] + T + hi;
        sum[ch][3] = T + hi + v[15];
    }
    float2 *out = (float2 *)(c.fout + ((off_t)yo * (off_t)c.W + (off_t)x0) * 8);
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const double g11 = sum[0][u], g12 = sum[1][u], g22 = sum[2][u], h1 = sum[3][u], h2 = sum[4][u];
        // flow = (G h)/(det G + 1e-3) with G, h the window MEANS: evaluated on the window SUMS with the
        // regulariser scaled by 169^2 instead (saves five multiplies); hardware reciprocal + one Newton step
        const double det = g11 * g22 - g12 * g12 + 1e-3 * (double)(FBI_WIN * FBI_WIN) * (double)(FBI_WIN * FBI_WIN);
        double idet = __builtin_amdgcn_rcp(det);
        idet = idet * (2.0 - det * idet);
        float2 f;
        f.x = (float)((g11 * h2 - g12 * h1) * idet);
        f.y = (float)((g22 * h1 - g12 * h2) * idet);
        if (x0 + u < c.W) out[u] = f;
    }
}

// one group of G consecutive window rows s0 .. s0+G-1 with STATIC ring slots K0 .. K0+G-1:
//   1. evaluate M for NB rows at a time (their loads in flight together), slide the 13-row column sums,
//      park them in LDS,
//   2. fetch the flow of the next group's GN rows,
//   3. one barrier, then the horizontal phase: thread t takes (row t / 29, quad t % 29); a 5-row group has 145 items,
//      so its threads 0 .. 16 take a second one.
// ABL 1 (development aid, env TF_FBI_ABLATE=1): synthetic M instead of the gathers; ABL 2: the gathers read row 0 only
// (same instructions, cache-resident data).
template <int K0, int G, int GN, int NB, int ABL>
__device__ __forceinline__ void fb_iter_group(const FbIterCtx &c, int s0, float (&ring)[FBI_WIN][5], double (&S)[5],
                                              float2 (&fl)[FBI_G], double *vrow)
{
#pragma unroll
    for (int g0 = 0; g0 < G; g0 += NB) {
        FbTaps t[NB];
        float m[NB][5];
#pragma unroll
        for (int r = 0; r < NB; r++)
            if (g0 + r < G && ABL != 1) fb_taps_load<ABL>(c, s0 + g0 + r, fl[g0 + r], t[r]);
#pragma unroll
        for (int r = 0; r < NB; r++)
            if (g0 + r < G) {
                const int g = g0 + r, s = s0 + g;
                if (ABL == 1) { m[r][0] = (float)s; m[r][1] = (float)c.xc; m[r][2] = 1.f; m[r][3] = 2.f; m[r][4] = (float)(s + c.xc); }
                else fb_taps_eval(c, s, t[r], m[r]);
#pragma unroll
                for (int ch = 0; ch < 5; ch++) {
                    S[ch] += (double)(m[r][ch] - ring[K0 + g][ch]); ring[K0 + g][ch] = m[r][ch];
                    vrow[(g * 5 + ch) * FBI_VS + c.dj] = S[ch];
                }
            }
    }
    if (ABL != 1) {
#pragma unroll
        for (int g = 0; g < GN; g++) fl[g] = fb_iter_flow_at(c, s0 + G + g);
    }
    __syncthreads();
    if (c.j < G * FBI_Q) fb_iter_quad(c, s0 + c.tg - FBI_M, c.tg, c.tq, vrow);
    if (G * FBI_Q > FBI_T && c.j < G * FBI_Q - FBI_T) {
        const int item = c.j + FBI_T, tg = item / FBI_Q;
        fb_iter_quad(c, s0 + tg - FBI_M, tg, item - tg * FBI_Q, vrow);
    }
    __syncthreads();
}

// Workgroup width (round 2): 128 threads = two waves.  The occupancy is two waves per SIMD whatever the width, i.e. FOUR
// independent barrier domains per CU instead of the two of 256-thread workgroups: the phases of a workgroup (gathers,
// matrix arithmetic, LDS column sums, window sums + solve) run one after the other, so what overlaps them is the number
// of workgroups in different phases.  12 x 5424^2 step: 132.4 ms (256 threads, 5 % halo columns) -> 115.4 ms (128
// threads, 10 %) -> 126.2 ms (64 threads, 23 %).  The rows-in-flight count FBI_NB no longer matters (2 / 4 / 5: 116.3 /
// 116.1 / 116.3 ms): with every gather redirected to a cache-resident row (TF_FBI_ABLATE=2) the 256-thread kernel
// took 117.9 instead of 132.5 ms and without gathers and matrix arithmetic (TF_FBI_ABLATE=1) 51.7 ms -- the kernel is
// bound by the SUM of its VALU (about 55 ms at full issue rate), L1 (22 ms) and LDS (22 ms) work, not by HBM latency.
// Occupancy is two waves per SIMD by construction (65 ring registers + the 72 of the horizontal phase; 64 KB of LDS):
// the launch bound says so, which lets the compiler schedule for the 256-register budget (5 % faster than the
// default bound).  Forcing three waves spills the ring (+70 %); row groups 4 + 4 + 4 + 1 with 51 KB of LDS cost
// 2 % for the extra barrier pair and gain nothing while the registers hold the kernel at two waves.
// Workgroup = (pair, strip of FBI_OW columns), all rows, both directions (two waves each).
//
// Why whole columns.  OpenCV's vertical running sum, term for term (FarnebackUpdateFlow_Blur; oracle/c/farneback.c:263-281):
//   vsum  = M[0] * (m + 2)  [float product]  + M[1] + ... + M[m-1]          (rows clamped to H - 1)
//   row y:  vsum += (double)( M[min(y + m, H - 1)] - M[max(y - m - 1, 0)] )   the difference ROUNDED TO FLOAT
// The float rounding of every difference stays in vsum for all rows below it -- a drift that depends on the whole
// column above a pixel, not on its 13-row window.  It is of the order of 1e-7 of the largest M in the column, which
// in a low-texture spot below a textured one moves the solution by > 1e-4 (round 2: max 1.5e-4 at 5424^2 with
// row strips that restarted the sum every <= 512 rows; now 7e-5, tests/test_gpu_fullsize.py).  So the rows of a chain
// are processed in order by one workgroup, and the parallelism of a launch is strips x directions x PAIRS.
// What was measured on the way (profiles/round3_fb_iter_notes.txt; 15 pairs at 5424^2, one launch):
//   row strips of round 2 (15 840 independent workgroups)           10.7 ms
//   whole columns, 1410 workgroups on 1024 resident slots            14.1 ms = exactly two rounds: a workgroup does not run
//       faster when its CU is half empty, and the L2 / HBM counters equal those of the row-strip kernel
//   chains cut into 8 segments handed over through memory, persistent workgroups on a ready queue (one queue, or one
//       per XCD): balanced, but 14.1 - 14.4 ms: 16 - 21 % more HBM fetches and 60 - 90 % more L1 stall cycles eat the gain
//   start delays that de-phase the workgroups: no effect
// So the remedy is not in the kernel but in the batch: the host layer sizes a launch to a whole number of rounds
// (tf_farneback_batch_hint: 21 pairs = 1974 chains = 96 % of two rounds at 5424^2), which a stack of frames allows.
template <int NB, int ABL>
__global__ void __launch_bounds__(2 * FBI_T, 2)
k_fb_iter_tree(FbIterArgs a, int H, int W, int64_t plane)
{
    // BOTH directions of a strip in ONE workgroup: waves 0 - 1 walk the column strip for direction dir[0], waves 2 - 3 for
    // dir[1], in step (they share the barriers).  Each direction reads its own expansion row by row and gathers from the
    // other's -- the very rows the other direction is reading: with the two in one workgroup every R row is fetched from
    // HBM once instead of twice (as separate workgroups they drift apart, and an XCD's L2 turns over every few
    // microseconds at this kernel's rate: measured traffic was that of NO sharing, 1.6 - 1.9 x the compulsory bytes).
    // Occupancy is unchanged: 4 waves and 64 KB of LDS per workgroup, two workgroups per CU.
    __shared__ double vrow_all[2][FBI_G * 5 * FBI_VS];
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x / FBI_T); // wave-uniform: 0 / 1 (one direction only: 128 threads, q = 0)
    double *vrow = vrow_all[q];
    const int b = blockIdx.x / a.nx, sx = blockIdx.x - b * a.nx;
    const int d = a.dir[q];                                            // 0: prev -> next, 1: next -> prev
    FbIterCtx c;
    const float *R0 = a.R[d] + b * a.bs_R, *R1 = a.R[1 - d] + b * a.bs_R;
    c.R0 = (const char *)R0; c.R0e = (const char *)(R0 + 4 * plane);
    {
        // corner displacements in elements; a level narrower / shorter than two pixels has no in-image patch at all
        // (inb is never true): all four corners then alias element 0 so that the discarded reads stay in bounds
        const bool patch = W >= 2 && H >= 2;
        const int64_t dcorner[4] = {0, patch ? 1 : 0, patch ? W : 0, patch ? (int64_t)W + 1 : 0};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            c.R1c[k] = (const char *)R1 + dcorner[k] * 16;
            c.R1ec[k] = (const char *)(R1 + 4 * plane) + dcorner[k] * 4;
        }
    }
    c.fin = (const char *)(a.fin[q] + b * a.bs_fin[q]); c.fout = (char *)(a.fout[q] + b * a.bs_fout[q]);
    c.H = H; c.W = W;
    c.j = threadIdx.x - q * FBI_T;
    c.dj = c.j + (c.j >> 2);
    c.tg = c.j / FBI_Q; c.tq = c.j - c.tg * FBI_Q;
    c.x_strip = sx * FBI_OW;
    c.xc = tf_clampi(c.x_strip + c.j - FBI_M, 0, W - 1);              // column this thread evaluates M for (replicate border)
    c.y0 = 0;
    c.y1 = H;                                                         // output rows [0, H)
    {
        const float border[5] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
        const int xb = W - 1 - c.xc;
        float lo = 1.f, hi = 1.f;
#pragma unroll
        for (int k = 0; k < 5; k++) { lo = c.xc == k ? border[k] : lo; hi = xb == k ? border[k] : hi; }
        c.xscale = lo * hi;
        c.xborder = (unsigned)(c.xc - 5) >= (unsigned)(W - 10);
        c.strip_xborder = W < 10 || c.x_strip - FBI_M < 5 || c.x_strip + FBI_T - FBI_M - 1 > W - 6;     // (columns x_strip - 6 .. x_strip + 121, clamped)
    }
    float ring[FBI_WIN][5];
    double S[5];
    float2 fl[FBI_G];
    {
        // rows 0 .. m-1 -> ring slots m+1 .. 2m; slots 0 .. m+1 hold row 0 (OpenCV's max(y - m - 1, 0) for the rows above the
        // image); three rows at a time (their loads in flight together, few registers live next to the ring)
#pragma unroll
        for (int r0 = 0; r0 < FBI_M; r0 += 3) {
            float2 f0[3];
            FbTaps t[3];
            float mm[3][5];
#pragma unroll
            for (int r = 0; r < 3; r++) f0[r] = (ABL != 1) ? fb_iter_flow_at(c, r0 + r) : make_float2(0.f, 0.f);
#pragma unroll
            for (int r = 0; r < 3; r++) if (ABL != 1) fb_taps_load<ABL>(c, r0 + r, f0[r], t[r]);
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const int row = r0 + r;
                if (ABL == 1) { mm[r][0] = (float)row; mm[r][1] = (float)c.xc; mm[r][2] = 1.f; mm[r][3] = 2.f; mm[r][4] = (float)(row + c.xc); }
                else fb_taps_eval(c, row, t[r], mm[r]);
#pragma unroll
                for (int ch = 0; ch < 5; ch++) {
                    if (row == 0) {
                        S[ch] = (double)(mm[r][ch] * (float)(FBI_M + 2));
#pragma unroll
                        for (int k = 0; k <= FBI_M + 1; k++) ring[k][ch] = mm[r][ch];       // rows -(m+1) .. 0
                    } else {
                        S[ch] += (double)mm[r][ch];
                        ring[FBI_M + 1 + row][ch] = mm[r][ch];
                    }
                }
            }
        }
    }
    const int s_last = H - 1 + FBI_M;                                  // window rows needed (inclusive; clamped to H - 1)
#pragma unroll
    for (int g = 0; g < FBI_G; g++) fl[g] = (ABL != 1) ? fb_iter_flow_at(c, FBI_M + g) : make_float2(0.f, 0.f);
    // row s enters ring slot (s - m) mod 13, which holds row s - 13 = (output row) - m - 1: OpenCV's srow0.
    // rows past s_last are evaluated (clamped, harmless) but never produce output
    for (int base = FBI_M; base <= s_last; base += FBI_WIN) {
        fb_iter_group<0, 4, 4, NB, ABL>(c, base, ring, S, fl, vrow);
        fb_iter_group<4, 4, 5, NB, ABL>(c, base + 4, ring, S, fl, vrow);
        fb_iter_group<8, 5, 4, NB, ABL>(c, base + 8, ring, S, fl, vrow);
    }
}

// ---- the same iteration with OpenCV's ROW sums too (round 4): bit-identical flow ---------------------------------------------
// FarnebackUpdateFlow_Blur forms the 13-wide window sums of a row as a RUNNING sum in double, left to right over the
// whole row (oracle/c/farneback.c:282-292):
//   g  = vsum[0] * (m + 2) + vsum[1] + ... + vsum[m - 1];     x = 0 .. W-1:   g += vsum[x + m] - vsum[x - m - 1]
// (columns left / right of the image replicate the border).  Every rounding of that chain stays in g for the rest of the
// row, so no other order of the additions reproduces it (k_fb_iter_tree above: max 7e-5 px away at 5424^2, which the
// refinement's 1/32-px remap bins amplify to 0.02 px in the composed flow).  A floating-point chain is sequential by
// definition; what is parallel is the number of chains -- rows x 5 channels x strips x directions x pairs:
//   * a workgroup still owns a strip of FBI_OW columns over all rows and forms OpenCV's column sums as before;
//   * per row group, lane (row r, channel ch) of each direction's first wave walks the strip's 116 columns:
//     g += V[x + 6] - V[x - 7] out of LDS, and leaves g in the slot whose column sum is no longer needed (25 chains side by
//     side; the other lanes have nothing to do for ~1 us: the price of the order);
//   * the chain ENTERS the strip with the g and the V[x_strip - 7] its left neighbour left at the same row: strips hand
//     10 doubles per row to the right through global memory -- a skewed pipeline, strip k one row group behind strip
//     k - 1.  Every double travels as two 64-bit words (launch tag << 32 | half): a word is valid iff its tag is this
//     launch's, so there is no flag to order against the data and no fence -- relaxed agent-scope atomics only.  The words
//     live in the (idle) blur scratch of the pair, zeroed once per pyramid level; tags count the level's iterations;
//   * workgroups take their (pair, strip, direction) from a TICKET counter in arrival order, strips of a pair left to right
//     (column group by column group when the launch needs several rounds of resident workgroups): a workgroup only ever
//     waits for a lower ticket, which is running or done -- no deadlock whatever the dispatch order;
//   * the 2 x 2 solve is OpenCV's expression on the window MEANS, with a true division (k_fb_iter_tree: scaled
//     regulariser, reciprocal + Newton step).
#define FBI_VS2 (FBI_T + 9)         // LDS row stride in doubles, odd: lanes (r, ch) of a scan hit different banks (137 * 2 mod 64 = 18)
#define FBI_HDR 16384               // bytes of ticket counters in front of a batch's hand-over words: 16 ints per launch of a level (8 used: one per XCD)
#define FBI_HW 20                   // hand-over words per row and strip: (g, V[next strip's x - 7]) x 5 channels x two halves, stored as four planes of H x 5

__device__ __forceinline__ unsigned long long fb_hand_ld(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void fb_hand_st(unsigned long long *p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// the hand-over words of one chain (row yo, channel ch), as loaded: valid iff all four tags are this launch's
struct FbHand { unsigned long long w0, w1, w2, w3; };
__device__ __forceinline__ void fb_hand_load(const FbIterCtx &c, int yo, int ch, FbHand &h)
{
    // four planes of H x 5 words (g low / high half, subtrahend low / high half): the 25 lanes of a row group read 25
    // CONSECUTIVE words per instruction (four cache lines instead of thirteen)
    const unsigned long long *pg = c.hin + (int64_t)yo * 5 + ch;
    const int64_t pl = (int64_t)c.H * 5;
    h.w0 = fb_hand_ld(pg); h.w1 = fb_hand_ld(pg + pl); h.w2 = fb_hand_ld(pg + 2 * pl); h.w3 = fb_hand_ld(pg + 3 * pl);
}
__device__ __forceinline__ bool fb_hand_valid(const FbIterCtx &c, const FbHand &h)
{
    return (unsigned)(h.w0 >> 32) == c.epoch && (unsigned)(h.w1 >> 32) == c.epoch && (unsigned)(h.w2 >> 32) == c.epoch && (unsigned)(h.w3 >> 32) == c.epoch;
}

// how a chain enters its strip: the running sum g left of column x_strip and the first subtrahend V[x_strip - 7] -- OpenCV's
// start of a row for strip 0 (row: LDS, slot i = column x_strip + i - 6), the left neighbour's hand-over words otherwise
// (`h`: as fetched at the top of the row group -- the neighbour is normally ahead: they are there -- polled here if not)
__device__ __forceinline__ void fb_chain_enter(const FbIterCtx &c, int yo, int ch, const double *row, FbHand h, double &g, double &sub)
{
    if (c.sx == 0 || (c.abl & 1)) {
        const double v0 = row[FBI_M];                                  // column 0
        g = v0 * (double)(FBI_M + 2);
#pragma unroll
        for (int x = 1; x < FBI_M; x++) g += row[FBI_M + x];           // (columns past W - 1 hold the clamped column: OpenCV's replicated border)
        sub = v0;                                                      // vsum[-m - 1] = vsum[0]
    } else {
        bool ok = fb_hand_valid(c, h);
        // (the left neighbour holds a lower ticket: it is running or done.  The poll count is bounded all the same -- several
        // seconds -- so that the grid drains whatever happens.  A chain that gives up continues with NaN AND says so: it sets
        // the launch's host-visible `starved` word, which turns the call that owns the launch into TF_ESTARVED -- a device
        // slowed down enough for this, by a profiler's serialisation or a preempted queue, must not hand out NaN rows with rc 0)
        if (!ok && c.spins) atomicAdd(c.spins, 1);                     // (development aid: chains that found their words missing at scan time)
        for (int spin = 0; !ok && spin < c.poll_limit; spin++) {
            if (c.spins) atomicAdd(c.spins + 1, 1);
            __builtin_amdgcn_s_sleep(1);
            fb_hand_load(c, yo, ch, h);
            ok = fb_hand_valid(c, h);
        }
        g = __longlong_as_double((long long)((h.w1 << 32) | (h.w0 & 0xffffffffull)));
        sub = __longlong_as_double((long long)((h.w3 << 32) | (h.w2 & 0xffffffffull)));
        if (!ok) {
            g = __longlong_as_double(0x7ff8000000000000ll);
            if (c.starved) __hip_atomic_store(c.starved, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// (g, V[next strip's x - 7]) for the strip to the right
__device__ __forceinline__ void fb_hand_store(const FbIterCtx &c, int yo, int ch, double g, double sub)
{
    if (c.sx >= c.nx - 1 || (c.abl & 64)) return;                    // (64: test aid -- the words never leave: every chain to the right starves)
    unsigned long long *pg = c.hout + (int64_t)yo * 5 + ch;
    const int64_t pl = (int64_t)c.H * 5;
    const unsigned long long tag = (unsigned long long)c.epoch << 32;
    const unsigned long long ug = (unsigned long long)__double_as_longlong(g), us = (unsigned long long)__double_as_longlong(sub);
    fb_hand_st(pg, tag | (ug & 0xffffffffull)); fb_hand_st(pg + pl, tag | (ug >> 32));
    fb_hand_st(pg + 2 * pl, tag | (us & 0xffffffffull)); fb_hand_st(pg + 3 * pl, tag | (us >> 32));
}

// NSTEP steps of a chain, straight-line: step s adds base[s + 12] - base[s - 1] (base[-1] = `sub` on entry) and leaves g in
// base[s].  A column sum is the minuend of step s and the subtrahend of step s + 13: chunks of WIN = 13 steps keep the last
// 13 minuends in registers, so every LDS slot is read ONCE (a lone wave pays 4 - 8 ns per instruction whatever it is --
// tools/microbench/chain_lds.hip: dependent add 5, subtraction 4, LDS read of two values 8, LDS write of two 5 ns -- so the
// chain is bound by its instruction count, not by the adds).  The reads of chunk c + 1 are issued before the chain of chunk
// c, whose writes go to slots below them.  Returns in `sub` the column sum base[NSTEP - 1] held on entry: the first
// subtrahend of whoever continues the chain.
// one chunk of WIN steps from slot i0: `mn` holds the minuends base[i0 + 12 + k], `prev` the subtrahends base[i0 - 1 + k].  Once the
// differences are formed `prev` is dead: the NEXT chunk's minuends are read into it (during the dependent additions), so two
// chunks with the two arrays swapped are one loop trip without a single register move (round 5: the rotating form
// prev <- mn <- nxt cost 25 v_mov_b64 per 13 steps, a third of the instructions of a loop that is bound by their number)
__device__ __forceinline__ void fb_chain_chunk(double *base, int i0, int last, double (&prev)[FBI_WIN], const double (&mn)[FBI_WIN], double &g)
{
    double d[FBI_WIN];
#pragma unroll
    for (int k = 0; k < FBI_WIN; k++) d[k] = mn[k] - prev[k];
#pragma unroll
    for (int k = 0; k < FBI_WIN; k++) prev[k] = base[min(i0 + FBI_WIN + 2 * FBI_M + k, last)];   // (a last chunk may be shorter: clamped, unused)
#pragma unroll
    for (int k = 0; k < FBI_WIN; k++) { g += d[k]; base[i0 + k] = g; }
}

template <int NSTEP>
__device__ __forceinline__ void fb_chain_run(double *base, double &g, double &sub)
{
    constexpr int CH = FBI_WIN, NFULL = NSTEP / CH, TAIL = NSTEP - NFULL * CH, LAST = NSTEP + 2 * FBI_M - 1;
    static_assert(NFULL >= 2 && NFULL % 2 == 0, "chain chunks come in pairs");
    double pa[CH], pb[CH];
    pa[0] = sub;
#pragma unroll
    for (int k = 1; k < CH; k++) pa[k] = base[k - 1];
#pragma unroll
    for (int k = 0; k < CH; k++) pb[k] = base[2 * FBI_M + k];
#pragma unroll 1
    for (int i0 = 0; i0 < NFULL * CH; i0 += 2 * CH) {
        fb_chain_chunk(base, i0, LAST, pa, pb, g);
        fb_chain_chunk(base, i0 + CH, LAST, pb, pa, g);
    }
    if (TAIL > 0) {
        double d[TAIL > 0 ? TAIL : 1];
#pragma unroll
        for (int k = 0; k < TAIL; k++) d[k] = pb[k] - pa[k];
#pragma unroll
        for (int k = 0; k < TAIL; k++) { g += d[k]; base[NFULL * CH + k] = g; }
    }
    sub = pa[TAIL];
}

// one whole chain by one lane: row `row` of output row yo, channel ch
__device__ __forceinline__ void fb_iter_scan(const FbIterCtx &c, int yo, int ch, double *row, FbHand h)
{
    const int n_out = min(FBI_OW, c.W - c.x_strip);                    // (wave-uniform)
    double g, sub;
    fb_chain_enter(c, yo, ch, row, h, g, sub);
    if (c.abl & 16) {                                                  // timing aid: hand-over without the chain (wrong flows)
    } else if (n_out == FBI_OW) fb_chain_run<FBI_OW>(row, g, sub);
    else {
        for (int i = 0; i < n_out; i++) {                              // the ragged last strip
            const double mnv = row[i + 2 * FBI_M], nxv = row[i];
            g += mnv - sub;
            row[i] = g;                                                // window sum of column x_strip + i, in the slot of V[x - 6] (read above)
            sub = nxv;
        }
    }
    fb_hand_store(c, yo, ch, g, sub);
}

// OpenCV's solve for output pixel (yo, x_strip + j) from the five window sums in LDS
// (rows5: the five channels' LDS rows, `stride` doubles apart; col: the pixel's slot in them)
__device__ __forceinline__ void fb_iter_solve(const FbIterCtx &c, int yo, const double *rows5, int stride = FBI_VS2, int col = -1)
{
    typedef fb_off_t off_t;
    if (col < 0) col = c.j;
    const double scale = 1. / (double)(FBI_WIN * FBI_WIN);
    const double g11 = rows5[col] * scale, g12 = rows5[stride + col] * scale, g22 = rows5[2 * stride + col] * scale;
    const double h1 = rows5[3 * stride + col] * scale, h2 = rows5[4 * stride + col] * scale;
    const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
    float2 f;
    f.x = (float)((g11 * h2 - g12 * h1) * idet);
    f.y = (float)((g22 * h1 - g12 * h2) * idet);
    *(float2 *)(c.fout + ((off_t)yo * (off_t)c.W + (off_t)(c.x_strip + c.j)) * 8) = f;
}

template <int K0, int G, int GN, int NB, int ABL, int VS = FBI_VS2>
__device__ __forceinline__ void fb_iter_group_seq(const FbIterCtx &c, int s0, float (&ring)[FBI_WIN][5], double (&S)[5],
                                                  float2 (&fl)[FBI_G], double *vrow)
{
    // the chain lanes ask for their left neighbour's hand-over words now: the answer arrives during the column phase
    FbHand hand = {0ull, 0ull, 0ull, 0ull};
    const bool chain_lane = c.j < G * 5 && s0 + c.sr - FBI_M < c.H;
    if (chain_lane && c.sx > 0) fb_hand_load(c, s0 + c.sr - FBI_M, c.sch, hand);
#pragma unroll
    for (int g0 = 0; g0 < G; g0 += NB) {
        FbTaps t[NB];
        float m[NB][5];
#pragma unroll
        for (int r = 0; r < NB; r++)
            if (g0 + r < G && ABL != 1) fb_taps_load<ABL>(c, s0 + g0 + r, fl[g0 + r], t[r]);
#pragma unroll
        for (int r = 0; r < NB; r++)
            if (g0 + r < G) {
                const int g = g0 + r, s = s0 + g;
                if (ABL == 1) { m[r][0] = (float)s; m[r][1] = (float)c.xc; m[r][2] = 1.f; m[r][3] = 2.f; m[r][4] = (float)(s + c.xc); }
                else fb_taps_eval(c, s, t[r], m[r]);
#pragma unroll
                for (int ch = 0; ch < 5; ch++) {
                    S[ch] += (double)(m[r][ch] - ring[K0 + g][ch]); ring[K0 + g][ch] = m[r][ch];
                    vrow[(g * 5 + ch) * VS + c.j] = S[ch];
                }
            }
    }
    if (ABL != 1) {
#pragma unroll
        for (int g = 0; g < GN; g++) fl[g] = fb_iter_flow_at(c, s0 + G + g);
    }
    __syncthreads();
    if (chain_lane && !(c.abl & 2)) {                                  // 20 or 25 chains, lanes of the direction's first wave
        // the chain is the serial stretch of the workgroup -- one wave, bound by its instruction count, everybody else at the
        // barrier -- while the SIMD's other wave (another workgroup's column phase) has work for both issue slots: the
        // chain wave takes priority in the arbitration for its duration (level-0 launch of 21 pairs: 19.7 -> 18.3 ms)
        if (!(c.abl & 128)) { if (c.abl & 256) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(3); }
        fb_iter_scan(c, s0 + c.sr - FBI_M, c.sch, vrow + c.j * VS, hand);             // (row r, channel ch) = LDS row r * 5 + ch = j
        if (!(c.abl & 128)) __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
    if (c.j < FBI_OW && c.x_strip + c.j < c.W && !(c.abl & 4)) {
#pragma unroll
        for (int r = 0; r < G; r++) {
            const int yo = s0 + r - FBI_M;
            if (yo < c.H) fb_iter_solve(c, yo, vrow + (r * 5) * VS, VS);
        }
    }
    __syncthreads();
}

// ---- THE CHAIN IN TWO PARTS, ONE ROW GROUP APART (HP form of the kernel) --------------------------------------------------------
// A chain uses 20 - 25 of its wave's 64 lanes for 116 steps, and a lone wave pays for instructions, not for lanes.  So the
// wave chains TWO parts at once: lanes 0 .. 24 the LEFT part (steps 0 .. 63) of the current row group, lanes 32 .. 56 the
// RIGHT part (steps 64 .. 115) of the PREVIOUS one, which starts from the g the left part reached a group earlier -- 64 steps
// per row group instead of 116, the same additions in the same order.  (64 + 52, not 58 + 58: the pixels of the left part are
// then solved by the workgroup's first wave, those of the right part by its second, without divergence.)  What the right
// part needs of its group outlives the group's LDS rows in a side buffer: the column sums of slots 63 .. 127 (65 per row and
// channel: slot 63 written by the left chain itself, which replaces it by its last sum; slots 64 .. 127 copied by the thread
// that has just solved the pixel whose entry it overwrites) and the left part's g (gmid).  The right-part pixels are solved
// one group late; the strip's hand-over words leave one group late too (a lag, not a cost).  LDS: 25 x (129 + 65 + 1)
// doubles = 39 000 B per direction -- four workgroups still fit a CU's 160 KB.  After the last group one more chain phase
// flushes the right part.
#define FBI_HS (FBI_T + 1)           // row stride of the column sums (129)
#define FBI_HL 64                    // steps of the left part
#define FBI_HRN (FBI_OW - FBI_HL)    // steps of the right part (52)
#define FBI_HR (FBI_T - FBI_HL + 1)  // side-buffer row: slots 63 .. 127 (65)
#define FBI_HP_DOUBLES (FBI_G * 5 * (FBI_HS + FBI_HR + 1))
static_assert(FBI_T == 128 && FBI_HRN == 4 * FBI_WIN && FBI_HL == FBI_HRN + FBI_WIN - 1 && FBI_G * 5 <= 32, "two parts of a strip in one wave");

// FBI_HRN steps by every active lane, FBI_HL - FBI_HRN more by the lanes with `left` (fb_chain_run for two lengths in one
// instruction stream); base[-1] = `sub` on entry; `sub` returns the column sum the last step's slot held
__device__ __forceinline__ void fb_chain_run_lr(double *base, double &g, double &sub, bool left)
{
    constexpr int CH = FBI_WIN, NFULL = FBI_HRN / CH, TAIL = FBI_HL - FBI_HRN;
    static_assert(NFULL % 2 == 0, "chain chunks come in pairs");
    const int last = (left ? FBI_HL : FBI_HRN) + 2 * FBI_M - 1;
    double pa[CH], pb[CH];
    pa[0] = sub;
#pragma unroll
    for (int k = 1; k < CH; k++) pa[k] = base[k - 1];
#pragma unroll
    for (int k = 0; k < CH; k++) pb[k] = base[2 * FBI_M + k];
#pragma unroll 1
    for (int i0 = 0; i0 < NFULL * CH; i0 += 2 * CH) {
        fb_chain_chunk(base, i0, last, pa, pb, g);
        fb_chain_chunk(base, i0 + CH, last, pb, pa, g);
    }
    sub = pa[0];                                                       // right part: slot 115, the next strip's first subtrahend
    if (left) {
        double d[TAIL];
#pragma unroll
        for (int k = 0; k < TAIL; k++) d[k] = pb[k] - pa[k];
#pragma unroll
        for (int k = 0; k < TAIL; k++) { g += d[k]; base[NFULL * CH + k] = g; }
        sub = pa[TAIL];                                                // left part: slot 63, the right part's first subtrahend
    }
}

// the right part of row group (s0p, PG rows) by lanes 32 .. 56, the left part of (s0, G rows) by lanes 0 .. 24; G = 0: flush
// (gR, subR: what the previous group's left part left for the right lanes -- fb_right_entry, read BEFORE the barrier in front of
// this phase: the left lanes overwrite both entries at the end of the phase, and the order of the two roles inside one wave
// is not something to leave to the code layout)
__device__ __forceinline__ void fb_right_entry(const FbIterCtx &c, const double *lds, double &gR, double &subR)
{
    const double *Rb = lds + FBI_G * 5 * FBI_HS, *gmid = Rb + FBI_G * 5 * FBI_HR;
    const int jr = c.j - 32;
    gR = 0.; subR = 0.;
    if (jr >= 0 && jr < FBI_G * 5) { gR = gmid[jr]; subR = Rb[jr * FBI_HR]; }
}
template <int G, int PG>
__device__ __forceinline__ void fb_iter_chain_parts(const FbIterCtx &c, int s0, int s0p, bool has_prev, FbHand hand, double *lds, double gR, double subR)
{
    double *M = lds, *Rb = lds + FBI_G * 5 * FBI_HS, *gmid = Rb + FBI_G * 5 * FBI_HR;
    const int jr = c.j - 32;
    const bool chainL = G > 0 && c.j < G * 5 && s0 + c.sr - FBI_M < c.H;
    const bool chainR = PG > 0 && has_prev && jr >= 0 && jr < PG * 5 && s0p + c.rsr - FBI_M < c.H;
    if ((chainL || chainR) && !(c.abl & 2)) {
        if (!(c.abl & 128)) __builtin_amdgcn_s_setprio(3);             // (the serial stretch of the workgroup: see fb_iter_group_seq)
        double g, sub, *base;
        if (chainR) { base = Rb + jr * FBI_HR + 1; g = gR; sub = subR; }
        else { base = M + c.j * FBI_HS; fb_chain_enter(c, s0 + c.sr - FBI_M, c.sch, base, hand, g, sub); }
        fb_chain_run_lr(base, g, sub, !chainR);
        if (chainR) fb_hand_store(c, s0p + c.rsr - FBI_M, c.rsch, g, sub);
        else { gmid[c.j] = g; Rb[c.j * FBI_HR] = sub; }                // (the right lanes hold both entries in registers since before the barrier)
        if (!(c.abl & 128)) __builtin_amdgcn_s_setprio(0);
    }
}

template <int K0, int G, int GN, int NB, int ABL, int PG>
__device__ __forceinline__ void fb_iter_group_half(const FbIterCtx &c, int s0, float (&ring)[FBI_WIN][5], double (&S)[5],
                                                   float2 (&fl)[FBI_G], double *lds, bool has_prev)
{
    double *M = lds, *Rb = lds + FBI_G * 5 * FBI_HS;
    FbHand hand = {0ull, 0ull, 0ull, 0ull};
    if (c.j < G * 5 && s0 + c.sr - FBI_M < c.H && c.sx > 0) fb_hand_load(c, s0 + c.sr - FBI_M, c.sch, hand);
#pragma unroll
    for (int g0 = 0; g0 < G; g0 += NB) {
        FbTaps t[NB];
        float m[NB][5];
#pragma unroll
        for (int r = 0; r < NB; r++)
            if (g0 + r < G && ABL != 1) fb_taps_load<ABL>(c, s0 + g0 + r, fl[g0 + r], t[r]);
#pragma unroll
        for (int r = 0; r < NB; r++)
            if (g0 + r < G) {
                const int g = g0 + r, s = s0 + g;
                if (ABL == 1) { m[r][0] = (float)s; m[r][1] = (float)c.xc; m[r][2] = 1.f; m[r][3] = 2.f; m[r][4] = (float)(s + c.xc); }
                else fb_taps_eval(c, s, t[r], m[r]);
#pragma unroll
                for (int ch = 0; ch < 5; ch++) {
                    S[ch] += (double)(m[r][ch] - ring[K0 + g][ch]); ring[K0 + g][ch] = m[r][ch];
                    M[(g * 5 + ch) * FBI_HS + c.j] = S[ch];
                }
            }
    }
    if (ABL != 1) {
#pragma unroll
        for (int g = 0; g < GN; g++) fl[g] = fb_iter_flow_at(c, s0 + G + g);
    }
    double gR, subR;
    fb_right_entry(c, lds, gR, subR);
    __syncthreads();
    fb_iter_chain_parts<G, PG>(c, s0, s0 - PG, has_prev, hand, lds, gR, subR);
    __syncthreads();
    const bool in_x = c.x_strip + c.j < c.W;
    if (c.j < FBI_HL) {                                                // first wave: the left-part pixels of this group
        if (in_x && !(c.abl & 4)) {
#pragma unroll
            for (int r = 0; r < G; r++) {
                const int yo = s0 + r - FBI_M;
                if (yo < c.H) fb_iter_solve(c, yo, M + (r * 5) * FBI_HS, FBI_HS, c.j);
            }
        }
    } else {                                                           // second wave: the right-part pixels of the previous group ...
        if (PG > 0 && has_prev && c.j < FBI_OW && in_x && !(c.abl & 4)) {
#pragma unroll
            for (int r = 0; r < (PG > 0 ? PG : 1); r++) {
                const int yo = s0 - PG + r - FBI_M;
                if (yo < c.H) fb_iter_solve(c, yo, Rb + (r * 5) * FBI_HR, FBI_HR, c.j - (FBI_HL - 1));
            }
        }
        // ... then this group's column sums of slots 64 .. 127 into the side buffer, each thread the entry it has just read
        const double *src = M + c.j;
        double *dst = Rb + c.j - (FBI_HL - 1);
#pragma unroll 1
        for (int r = 0; r < G; r++) {                                  // (five entries at a time: the copy must not cost registers)
            double v[5];
#pragma unroll
            for (int ch = 0; ch < 5; ch++) v[ch] = src[(r * 5 + ch) * FBI_HS];
#pragma unroll
            for (int ch = 0; ch < 5; ch++) dst[(r * 5 + ch) * FBI_HR] = v[ch];
        }
    }
    __syncthreads();
}

// NDW = directions per workgroup: 2 = both directions of a strip share a workgroup (and the R rows they read), 1 = every
// (strip, direction) is a two-wave workgroup of its own, four per CU (a.nq = 2 then: directions take tickets of their own)
// HP = 1: the chain in two parts one row group apart (full strips; a ragged last strip chains whole, on the same LDS rows)
template <int NB, int ABL, int NDW, int HP>
__global__ void __launch_bounds__(NDW * FBI_T, 2)
k_fb_iter(FbIterArgs a, int H, int W, int64_t plane)
{
    __shared__ double vrow_all[NDW][HP ? FBI_HP_DOUBLES : FBI_G * 5 * FBI_VS2];
    // TICKETS, ONE LIST PER XCD.  Pair b belongs to the list of XCD b mod 8 -- with all its strips, both directions and all
    // column groups -- and a workgroup takes the next item of ITS XCD's list (s_getreg XCC_ID): the two directions of a
    // strip, which read each other's expansion rows, then run on the same XCD within microseconds and share them in its
    // L2 (one workgroup per direction otherwise fetches every R row from HBM twice), and the strips of a pair hand their
    // words to a neighbour on the same XCD.  A workgroup whose own list is exhausted takes from the next XCD's (21 pairs
    // on 8 XCDs: three of them own two pairs, five own three).  Inside a list tickets run column group by column group,
    // pair by pair, strips left to right: a workgroup only ever waits for a lower ticket of the same list, whose owner
    // is running or done whichever XCD it came from -- no deadlock whatever the dispatch order.
    __shared__ int s_ticket[2];
    if (threadIdx.x == 0) {
        const int xcc = a.xcd_lists ? (int)(__builtin_amdgcn_s_getreg(6164) & 7u) : 0;      // HW_REG_XCC_ID (id 20), bits 3:0
        const int n_lists = a.xcd_lists ? 8 : 1;
        int found = -1, y = 0;
        for (int k = 0; k < n_lists && found < 0; k++) {
            y = (xcc + k) & (n_lists - 1);
            const int n_pairs = a.nb > y ? (a.nb - y + n_lists - 1) / n_lists : 0;
            if (n_pairs == 0) continue;
            const int t = atomicAdd(a.ticket + y, 1);
            if (t < n_pairs * a.nx * a.nq) found = t;                  // (the grid has exactly one workgroup per item: some list has one left)
        }
        s_ticket[0] = found; s_ticket[1] = y;
    }
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(s_ticket[0]), list = __builtin_amdgcn_readfirstlane(s_ticket[1]);
    if (ticket < 0) return;                                            // (cannot happen: more workgroups than items)
    const int n_lists = a.xcd_lists ? 8 : 1, list_pairs = (a.nb - list + n_lists - 1) / n_lists;
    const int qw = __builtin_amdgcn_readfirstlane(threadIdx.x / FBI_T); // wave-uniform: which half of the workgroup (0 when it has one direction)
    double *vrow = vrow_all[qw];
    // ticket -> (pair, strip): column groups of nxg strips, group by group; inside a group pair by pair, strips left to right
    // (a strip's left neighbour always holds a lower ticket).  A launch that needs several rounds of resident workgroups
    // runs one column group per round: the start delays below then add up over nxg strips, not over all of them.
    const int per_group = list_pairs * a.nxg * a.nq;
    const int cg = ticket / per_group, tr = ticket - cg * per_group, n_cg = min(a.nxg, a.nx - cg * a.nxg);
    const int bi = tr / (n_cg * a.nq), tr2 = tr - bi * n_cg * a.nq, sxl = tr2 / a.nq;
    const int b = list + n_lists * bi;
    const int sx = cg * a.nxg + sxl;
    const int q = NDW == 2 ? qw : tr2 - sxl * a.nq;                    // index into the launch's directions
    const int d = a.dir[q];                                            // 0: prev -> next, 1: next -> prev
    FbIterCtx c;
    const float *R0 = a.R[d] + b * a.bs_R, *R1 = a.R[1 - d] + b * a.bs_R;
    c.R0 = (const char *)R0; c.R0e = (const char *)(R0 + 4 * plane);
    {
        const bool patch = W >= 2 && H >= 2;
        const int64_t dcorner[4] = {0, patch ? 1 : 0, patch ? W : 0, patch ? (int64_t)W + 1 : 0};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            c.R1c[k] = (const char *)R1 + dcorner[k] * 16;
            c.R1ec[k] = (const char *)(R1 + 4 * plane) + dcorner[k] * 4;
        }
    }
    c.fin = (const char *)(a.fin[q] + b * a.bs_fin[q]); c.fout = (char *)(a.fout[q] + b * a.bs_fout[q]);
    c.H = H; c.W = W;
    c.j = threadIdx.x - qw * FBI_T;
    c.dj = c.j; c.tg = 0; c.tq = 0;
    c.sr = c.j / 5; c.sch = c.j - c.sr * 5;
    c.rsr = c.j >= 32 ? (c.j - 32) / 5 : 0; c.rsch = c.j >= 32 ? (c.j - 32) - c.rsr * 5 : 0;
    c.full = W - sx * FBI_OW >= FBI_OW;
    c.sx = sx; c.nx = a.nx; c.epoch = a.epoch; c.abl = a.abl; c.spins = (a.abl & 8) ? a.ticket - 16 * (a.epoch - 1) + FBI_HDR / 4 - 4 : nullptr;
    c.starved = a.starved; c.poll_limit = a.poll_limit;
    {
        // hand-over slots of (pair b, direction q, strip): [q][strip 0 .. nx - 2][row][FBI_HW words]
        unsigned long long *hb = a.hand + b * a.bs_hand + (int64_t)q * (a.nx - 1) * H * FBI_HW;
        c.hout = hb + (int64_t)sx * H * FBI_HW;
        c.hin = hb + (int64_t)(sx > 0 ? sx - 1 : 0) * H * FBI_HW;
    }
    c.x_strip = sx * FBI_OW;
    c.xc = tf_clampi(c.x_strip + c.j - FBI_M, 0, W - 1);              // column this thread evaluates M for (replicate border)
    c.y0 = 0;
    c.y1 = H;
    {
        const float border[5] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
        const int xb = W - 1 - c.xc;
        float lo = 1.f, hi = 1.f;
#pragma unroll
        for (int k = 0; k < 5; k++) { lo = c.xc == k ? border[k] : lo; hi = xb == k ? border[k] : hi; }
        c.xscale = lo * hi;
        c.xborder = (unsigned)(c.xc - 5) >= (unsigned)(W - 10);
        c.strip_xborder = W < 10 || c.x_strip - FBI_M < 5 || c.x_strip + FBI_T - FBI_M - 1 > W - 6;     // (columns x_strip - 6 .. x_strip + 121, clamped)
    }
    float ring[FBI_WIN][5];
    double S[5];
    float2 fl[FBI_G];
    {
        // rows 0 .. m-1 -> ring slots m+1 .. 2m; slots 0 .. m+1 hold row 0 (as in k_fb_iter_tree)
#pragma unroll
        for (int r0 = 0; r0 < FBI_M; r0 += 3) {
            float2 f0[3];
            FbTaps t[3];
            float mm[3][5];
#pragma unroll
            for (int r = 0; r < 3; r++) f0[r] = (ABL != 1) ? fb_iter_flow_at(c, r0 + r) : make_float2(0.f, 0.f);
#pragma unroll
            for (int r = 0; r < 3; r++) if (ABL != 1) fb_taps_load<ABL>(c, r0 + r, f0[r], t[r]);
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const int row = r0 + r;
                if (ABL == 1) { mm[r][0] = (float)row; mm[r][1] = (float)c.xc; mm[r][2] = 1.f; mm[r][3] = 2.f; mm[r][4] = (float)(row + c.xc); }
                else fb_taps_eval(c, row, t[r], mm[r]);
#pragma unroll
                for (int ch = 0; ch < 5; ch++) {
                    if (row == 0) {
                        S[ch] = (double)(mm[r][ch] * (float)(FBI_M + 2));
#pragma unroll
                        for (int k = 0; k <= FBI_M + 1; k++) ring[k][ch] = mm[r][ch];       // rows -(m+1) .. 0
                    } else {
                        S[ch] += (double)mm[r][ch];
                        ring[FBI_M + 1 + row][ch] = mm[r][ch];
                    }
                }
            }
        }
    }
    const int s_last = H - 1 + FBI_M;
#pragma unroll
    for (int g = 0; g < FBI_G; g++) fl[g] = (ABL != 1) ? fb_iter_flow_at(c, FBI_M + g) : make_float2(0.f, 0.f);
    if (sx > 0 && a.slack_rows > 0 && !(a.abl & 1)) {
        // SLACK.  Strips that start together run in lockstep: each finds its left neighbour's words missing at every row
        // group, polls, and passes every hiccup on to all strips right of it (measured: 55 % of the chains waited, the launch
        // took 1.5 x the time of its parts).  So a strip starts only when its left neighbour is slack_rows ahead: the
        // words are then there when the row group asks for them, and a neighbour's hiccup is absorbed by the lead.
        if (threadIdx.x == 0) {
            const unsigned long long *p = c.hin + (int64_t)min(a.slack_rows, H - 1) * 5;
            // (a wait that gives up here is not an error: the slack is a scheduling hint, every chain below still enters through
            // fb_chain_enter, which is where a missing word is polled for and -- if it never comes -- reported)
            for (int spin = 0; spin < a.poll_limit && (unsigned)(fb_hand_ld(p) >> 32) != a.epoch; spin++) __builtin_amdgcn_s_sleep(8);
        }
        __syncthreads();
    }
    if (HP && c.full) {
        for (int base = FBI_M; base <= s_last; base += FBI_WIN) {
            fb_iter_group_half<0, 4, 4, NB, ABL, 5>(c, base, ring, S, fl, vrow, base > FBI_M);
            fb_iter_group_half<4, 4, 5, NB, ABL, 4>(c, base + 4, ring, S, fl, vrow, true);
            fb_iter_group_half<8, 5, 4, NB, ABL, 4>(c, base + 8, ring, S, fl, vrow, true);
        }
        // flush: the right part of the last group (its rows past H - 1 produce nothing)
        const int s0p = FBI_M + ((s_last - FBI_M) / FBI_WIN) * FBI_WIN + 8;
        const FbHand none = {0ull, 0ull, 0ull, 0ull};
        double gR, subR;
        fb_right_entry(c, vrow, gR, subR);                              // (no left part in the flush: nobody writes these entries)
        fb_iter_chain_parts<0, 5>(c, s0p + 5, s0p, true, none, vrow, gR, subR);
        __syncthreads();
        if (c.j >= FBI_HL && c.j < FBI_OW && c.x_strip + c.j < c.W && !(c.abl & 4)) {
#pragma unroll
            for (int r = 0; r < 5; r++) {
                const int yo = s0p + r - FBI_M;
                if (yo < c.H) fb_iter_solve(c, yo, vrow + FBI_G * 5 * FBI_HS + (r * 5) * FBI_HR, FBI_HR, c.j - (FBI_HL - 1));
            }
        }
        return;
    }
    for (int base = FBI_M; base <= s_last; base += FBI_WIN) {
        fb_iter_group_seq<0, 4, 4, NB, ABL, (HP ? FBI_HS : FBI_VS2)>(c, base, ring, S, fl, vrow);
        fb_iter_group_seq<4, 4, 5, NB, ABL, (HP ? FBI_HS : FBI_VS2)>(c, base + 4, ring, S, fl, vrow);
        fb_iter_group_seq<8, 5, 4, NB, ABL, (HP ? FBI_HS : FBI_VS2)>(c, base + 8, ring, S, fl, vrow);
    }
}

// ---- FarnebackUpdateFlow_Blur: box filter of M (double) + 2x2 solve ------------------------------
// Tile: 64 x 16 outputs per 256-thread workgroup (each thread: one column, 4 rows); per channel the
// (64 + 2m) x (16 + 2m) float tile is staged in LDS, reduced to column sums (double) and then row sums.
#define FBT_W 64
#define FBT_H 16
#define FB_MAX_M 8
__global__ void __launch_bounds__(256)
k_fb_blur_solve(const float *__restrict__ M, int H, int W, int64_t plane, int m, float *__restrict__ flow)
{
    __shared__ float tile[(FBT_H + 2 * FB_MAX_M) * (FBT_W + 2 * FB_MAX_M)];
    __shared__ double vs[FBT_H * (FBT_W + 2 * FB_MAX_M)];
    const int tw = FBT_W + 2 * m, th = FBT_H + 2 * m;
    const int bx = blockIdx.x * FBT_W, by = blockIdx.y * FBT_H;
    const int tid = threadIdx.y * 64 + threadIdx.x;
    double acc[5][4];
#pragma unroll
    for (int c = 0; c < 5; c++) {
        const float *Mc = M + c * plane;
        for (int i = tid; i < tw * th; i += 256) {
            const int ty = i / tw, tx = i - ty * tw;
            const int gy = tf_clampi(by + ty - m, 0, H - 1), gx = tf_clampi(bx + tx - m, 0, W - 1);
            tile[i] = Mc[(int64_t)gy * W + gx];
        }
        __syncthreads();
        for (int i = tid; i < tw * FBT_H; i += 256) {           // column sums over 2m+1 rows
            const int oy = i / tw, tx = i - oy * tw;
            double s = 0;
            for (int j = 0; j <= 2 * m; j++) s += (double)tile[(oy + j) * tw + tx];
            vs[i] = s;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int oy = threadIdx.y * 4 + r;
            double s = 0;
            for (int j = 0; j <= 2 * m; j++) s += vs[oy * tw + threadIdx.x + j];
            acc[c][r] = s;
        }
        __syncthreads();
    }
    const double scale = 1. / ((2 * m + 1) * (2 * m + 1));
    const int x = bx + threadIdx.x;
    if (x >= W) return;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int y = by + threadIdx.y * 4 + r;
        if (y >= H) continue;
        const double g11 = acc[0][r] * scale, g12 = acc[1][r] * scale, g22 = acc[2][r] * scale;
        const double h1 = acc[3][r] * scale, h2 = acc[4][r] * scale;
        const double idet = 1. / (g11 * g22 - g12 * g12 + 1e-3);
        float2 f;
        f.x = (float)((g11 * h2 - g12 * h1) * idet);
        f.y = (float)((g22 * h1 - g12 * h2) * idet);
        ((float2 *)flow)[(int64_t)y * W + x] = f;
    }
}

__global__ void __launch_bounds__(256) k_fb_zero(float *__restrict__ p, int64_t count, int64_t bs) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) p[(int64_t)blockIdx.z * bs + i] = 0.f;
}

// ---- host side -----------------------------------------------------------------------------------
extern "C" void tf_farneback_default_params(tf_farneback_params *p) {
    p->num_levels = 5; p->pyr_scale = 0.5; p->win_size = 13; p->num_iters = 10; p->poly_n = 5; p->poly_sigma = 1.1;
    p->chain_form = TF_FB_CHAIN_DEFAULT; p->status_slot = 0;
}

static void fb_gaussian_kernel(int n, double sigma, FbKernel *out) {
    // cv::getGaussianKernel(n, sigma, CV_32F)
    static const float small_tab[4][7] = {{1.f}, {0.25f, 0.5f, 0.25f}, {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f},
        {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f}};
    const float *fixed = (n % 2 == 1 && n <= 7 && sigma <= 0) ? small_tab[n >> 1] : nullptr;
    const double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    const double scale2X = -0.5 / (sigmaX * sigmaX);
    double sum = 0;
    out->ksize = n;
    for (int i = 0; i < n; i++) {
        const double x = i - (n - 1) * 0.5;
        out->k[i] = (float)(fixed ? (double)fixed[i] : exp(scale2X * x * x));
        sum += out->k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) out->k[i] = (float)(out->k[i] * sum);
}

static void fb_prepare_poly(int n, double sigma, FbPoly *pp) {
    // FarnebackPrepareGaussian: 1-D kernels + the four entries of inv(G) that are used
    if (sigma < FLT_EPSILON) sigma = n * 0.3;
    float gb[2 * FB_MAX_POLY_N + 1]; float *g = gb + n;
    double s = 0;
    for (int x = -n; x <= n; x++) { g[x] = (float)exp(-x * x / (2 * sigma * sigma)); s += g[x]; }
    s = 1. / s;
    pp->n = n;
    for (int x = -n; x <= n; x++) g[x] = (float)(g[x] * s);
    for (int x = 0; x <= n; x++) { pp->g[x] = g[x]; pp->xg[x] = (float)(x * g[x]); pp->xxg[x] = (float)(x * x * g[x]); }
    double G00 = 0, G11 = 0, G33 = 0, G55 = 0;
    for (int y = -n; y <= n; y++)
        for (int x = -n; x <= n; x++) {
            G00 += g[y] * g[x]; G11 += g[y] * g[x] * x * x;
            G33 += g[y] * g[x] * x * x * x * x; G55 += g[y] * g[x] * x * x * y * y;
        }
    // inv(G) by Gauss-Jordan elimination with partial pivoting in double, operation for operation as the oracle does it
    // (oracle/c/farneback.c inv6): the four entries are used as double factors of every expansion coefficient, and a
    // closed-form block inverse (rounds 1 - 3) differs from the elimination in the last digits -- enough to flip the float
    // rounding of one coefficient in ~10^5, which the sequential box sums then carry down a whole column.
    double m[6][12];
    for (int i = 0; i < 6; i++) for (int j = 0; j < 12; j++) m[i][j] = 0.;
    m[0][0] = G00; m[1][1] = m[2][2] = m[0][3] = m[0][4] = m[3][0] = m[4][0] = G11;
    m[3][3] = m[4][4] = G33; m[3][4] = m[4][3] = m[5][5] = G55;
    for (int i = 0; i < 6; i++) m[i][i + 6] = 1.;
    for (int c = 0; c < 6; c++) {
        int piv = c;
        for (int r = c + 1; r < 6; r++) if (fabs(m[r][c]) > fabs(m[piv][c])) piv = r;
        if (piv != c) for (int j = 0; j < 12; j++) std::swap(m[c][j], m[piv][j]);
        const double d = 1. / m[c][c];
        for (int j = 0; j < 12; j++) m[c][j] *= d;
        for (int r = 0; r < 6; r++) {
            if (r == c) continue;
            const double f = m[r][c];
            if (f != 0) for (int j = 0; j < 12; j++) m[r][j] -= f * m[c][j];
        }
    }
    pp->ig11 = m[1][7]; pp->ig03 = m[0][9]; pp->ig33 = m[3][9]; pp->ig55 = m[5][11];
}

static int fb_levels(int64_t H, int64_t W, const tf_farneback_params *p) {
    int k; double scale = 1;
    for (k = 0; k < p->num_levels; k++) { scale *= p->pyr_scale; if (W * scale < 32 || H * scale < 32) break; }
    return k;
}

// floats of blur scratch the iteration kernel borrows per pair at a level of h x w: 1 KB of ticket counters + the strips'
// hand-over words (both directions)
static size_t fb_hand_floats(int64_t h, int64_t w) {
    const int64_t nx = (w + FBI_OW - 1) / FBI_OW;
    return FBI_HDR / 4 + 2 * (size_t)(2 * (nx - 1) * h * FBI_HW);
}
static size_t fb_pair_floats(int64_t H, int64_t W, bool fused) {
    // per pair: tmp (n + 2H + 64), blur, I, R[2] (5n each), 2 flow scratch (2n each); the 5-plane matrix M (5n) only for the
    // unfused fallback (window sizes other than 13): the fused iteration never stores it
    const size_t n = (size_t)H * W;
    return tf_align_up(std::max(n + 2 * (size_t)H + 64, fb_hand_floats(H, W)), 64) + 2 * tf_align_up(n, 64) + (fused ? 10 : 15) * tf_align_up(n, 64) + 2 * tf_align_up(2 * n, 64);
}

extern "C" size_t tf_farneback_workspace_bytes_batch(int64_t B, int64_t H, int64_t W, const tf_farneback_params *p)
{
    if (B <= 0 || H <= 0 || W <= 0 || !p) return 0;
    return (size_t)B * fb_pair_floats(H, W, p->win_size == FBI_WIN) * sizeof(float) + 8192;
}

// Pairs per tf_farneback_batch call that fill the GPU best.  The iteration kernel runs one workgroup per (pair, strip,
// direction) over all rows, four workgroups are resident per CU, and a workgroup does not speed up when its CU is
// half empty: a launch costs ceil(chains / resident) rounds.  Returns the batch size <= max_pairs (and within
// max_bytes of workspace, 0 = no limit) with the best fill of its last round, the larger one on ties within 2 %.
extern "C" int64_t tf_farneback_batch_hint(int64_t H, int64_t W, const tf_farneback_params *p, int64_t max_pairs, size_t max_bytes)
{
    if (H <= 0 || W <= 0 || !p || max_pairs < 1) return 0;
    int dev = 0, n_cu = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) n_cu = 256;
    const int64_t slots = (int64_t)n_cu * 8 / (FBI_T / 64);                     // two waves per SIMD; one (strip, direction) per workgroup of FBI_T threads
    const size_t per_pair_bytes = fb_pair_floats(H, W, p->win_size == FBI_WIN) * sizeof(float);
    int64_t cap = max_pairs;
    if (max_bytes > 0 && (int64_t)(max_bytes / per_pair_bytes) < cap) cap = (int64_t)(max_bytes / per_pair_bytes);
    if (cap < 1) cap = 1;
    if (cap > 1024) cap = 1024;
    // A launch of the iteration kernel costs whole ROUNDS of resident workgroups, each as long as the level is high, at
    // EVERY pyramid level: B pairs cost sum_l ceil(B strips_l / slots) rows_l and do sum_l (B strips_l / slots) rows_l of
    // work.  The batch with the best ratio wins, the larger one among those within 1 % of it (at 5424^2 on 256 CUs: 42
    // pairs, 94.6 %: full rounds at levels 0, 1 AND 2; 54 pairs: 92.1 %, 43 pairs: 85 % -- level 1 spills into a third round).
    const int levels = fb_levels(H, W, p);
    auto efficiency = [&](int64_t B) {
        double cost = 0, work = 0, scale = 1;
        for (int k = 0; k <= levels; k++) {
            const int64_t w = (int64_t)lrint(W * scale), h = (int64_t)lrint(H * scale);
            const int64_t chains = 2 * B * ((w + FBI_OW - 1) / FBI_OW), rounds = (chains + slots - 1) / slots;
            cost += (double)rounds * (double)h; work += (double)chains / (double)slots * (double)h;
            scale *= p->pyr_scale;
        }
        return work / cost;
    };
    double best_eff = 0;
    for (int64_t B = 1; B <= cap; B++) best_eff = std::max(best_eff, efficiency(B));
    int64_t pick = 1;
    for (int64_t B = cap; B >= 1; B--) if (efficiency(B) >= best_eff - 0.01) { pick = B; break; }
    return pick;
}

// ---- starved chains -> TF_ESTARVED ---------------------------------------------------------------------------------------
// A ring of pinned, device-visible status words per device.  A chain of k_fb_iter that gives up on its left neighbour's
// hand-over words (fb_chain_enter) stores 1 into the word its launch was handed; the host reads it once the launches are
// known to have finished and turns it into an error: rows of NaN never leave with TF_OK.
// Round 6: WHICH word a launch is handed is the caller's -- tf_farneback_params.status_slot.  Slot 0 is the
// device's shared word (round 5's only one): every call with status_slot 0 reports and clears it on entry, tf_farneback_check()
// reads it.  A caller that runs several flows side by side on one device (detect_stack_windows' flood thread beside the next
// stack's flow; two host threads) acquires a slot per flow (tf_farneback_status_acquire): its calls write, report on entry and
// are checked on THAT word only, so one flow's check can neither consume nor be blamed for another flow's starved chain.
#define FB_STATUS_SLOTS 256
static std::mutex fb_starved_mu;
static int *fb_status_ring[TF_MAX_DEVICES] = {};
static unsigned char fb_status_busy[TF_MAX_DEVICES][FB_STATUS_SLOTS] = {};
static int fb_current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= TF_MAX_DEVICES) dev = 0;
    return dev;
}
static int *fb_status_ring_locked(int dev, bool create)
{
    if (!fb_status_ring[dev] && create) {
        void *p = nullptr;
        if (hipHostMalloc(&p, FB_STATUS_SLOTS * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        for (int k = 0; k < FB_STATUS_SLOTS; k++) ((volatile int *)p)[k] = 0;
        fb_status_ring[dev] = (int *)p;
        fb_status_busy[dev][0] = 1;              // the shared word is never handed out
    }
    return fb_status_ring[dev];
}
static int *fb_starved_word(bool create, int slot = 0)
{
    if (slot < 0 || slot >= FB_STATUS_SLOTS) slot = 0;
    const int dev = fb_current_device();
    std::lock_guard<std::mutex> lk(fb_starved_mu);
    int *ring = fb_status_ring_locked(dev, create);
    return ring ? ring + slot : nullptr;
}
static int fb_report_starved(const char *who, int slot = 0)
{
    int *w = fb_starved_word(false, slot);
    if (!w || __atomic_load_n(w, __ATOMIC_ACQUIRE) == 0) return TF_OK;
    __atomic_store_n(w, 0, __ATOMIC_RELEASE);
    tf_set_error("%s: a row-sum chain of the Farneback iteration kernel gave up waiting for its left neighbour's hand-over words "
                 "(device stalled: profiler serialisation, preempted queue?) -- the flow of that launch holds NaN rows and must be recomputed", who);
    return TF_ESTARVED;
}
extern "C" int tf_farneback_check(void) { return fb_report_starved("tf_farneback_check"); }
extern "C" int tf_farneback_status_acquire(void)
{
    const int dev = fb_current_device();
    std::lock_guard<std::mutex> lk(fb_starved_mu);
    int *ring = fb_status_ring_locked(dev, true);
    if (!ring) return 0;
    for (int k = 1; k < FB_STATUS_SLOTS; k++)
        if (!fb_status_busy[dev][k]) { fb_status_busy[dev][k] = 1; __atomic_store_n(ring + k, 0, __ATOMIC_RELEASE); return k; }
    return 0;                                    // all taken: the caller shares the device's word (round 5 behaviour)
}
extern "C" void tf_farneback_status_release(int slot)
{
    if (slot <= 0 || slot >= FB_STATUS_SLOTS) return;
    const int dev = fb_current_device();
    std::lock_guard<std::mutex> lk(fb_starved_mu);
    fb_status_busy[dev][slot] = 0;
}
extern "C" int tf_farneback_status_check(int slot)
{
    if (slot < 0 || slot >= FB_STATUS_SLOTS) { tf_set_error("tf_farneback_status_check: no such slot"); return TF_EINVAL; }
    return fb_report_starved(slot ? "tf_farneback_status_check" : "tf_farneback_check", slot);
}
// (test hooks: what the kernel does when it gives up, done from the host -- the host-side path can be tested without a device
// that stalls)
extern "C" int tf_farneback_debug_set_starved_slot(int slot)
{
    if (slot < 0 || slot >= FB_STATUS_SLOTS) { tf_set_error("tf_farneback_debug_set_starved_slot: no such slot"); return TF_EINVAL; }
    int *w = fb_starved_word(true, slot);
    if (!w) { tf_set_error("tf_farneback_debug_set_starved: no pinned word (no HIP device?)"); return TF_EHIP; }
    __atomic_store_n(w, 1, __ATOMIC_RELEASE);
    return TF_OK;
}
extern "C" int tf_farneback_debug_set_starved(void) { return tf_farneback_debug_set_starved_slot(0); }

// Workgroups of the iteration kernel's full-resolution launch for B pairs (both directions), and how many of them the
// device holds at once: a launch costs whole rounds of resident workgroups, so a caller that may cut a batch into parts
// (tf_farneback_batch_phase) does so only while a part still fills a round.
// resident two-wave workgroups of the current device (cached per device index: ADVICE r4 -- one `static int` kept the first
// device's count for every device)
static int fb_resident_slots()
{
    static std::atomic<int> cache[TF_MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= TF_MAX_DEVICES) dev = 0;
    int v = cache[dev].load(std::memory_order_relaxed);
    if (v == 0) {
        int n_cu = 256;
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n_cu <= 0) { (void)hipGetLastError(); n_cu = 256; }
        v = n_cu * 8 / (FBI_T / 64);
        cache[dev].store(v, std::memory_order_relaxed);
    }
    return v;
}
extern "C" int64_t tf_farneback_iteration_workgroups(int64_t H, int64_t W, const tf_farneback_params *p, int64_t B, int64_t *resident_out)
{
    if (H <= 0 || W <= 0 || !p || B < 1) return 0;
    if (resident_out) *resident_out = fb_resident_slots();
    return 2 * B * ((W + FBI_OW - 1) / FBI_OW);
}

extern "C" size_t tf_farneback_workspace_bytes(int64_t H, int64_t W, const tf_farneback_params *p)
{
    return tf_farneback_workspace_bytes_batch(1, H, W, p);
}

// scratch of one phase: every array holds B items back to back; bs_* = the items' stride in floats
struct FbScratch { float *tmp, *blur, *I, *R[2], *M, *fbuf[2]; int64_t bs_tmp, bs_n, bs_R, bs_f; };
static size_t fb_scratch_floats(size_t tmp_floats, size_t plane, bool fused) {
    return tf_align_up(tmp_floats, 64) + 2 * tf_align_up(plane, 64) + (fused ? 10 : 15) * tf_align_up(plane, 64) + 2 * tf_align_up(2 * plane, 64);
}
static bool fb_carve(TfArena &ar, int B, size_t tmp_floats, size_t plane, bool fused, FbScratch *S) {
    S->bs_tmp = (int64_t)tf_align_up(tmp_floats, 64); S->bs_n = (int64_t)tf_align_up(plane, 64);
    S->bs_R = 5 * S->bs_n; S->bs_f = (int64_t)tf_align_up(2 * plane, 64);
    S->tmp = ar.take<float>(S->bs_tmp * B); S->blur = ar.take<float>(S->bs_n * B); S->I = ar.take<float>(S->bs_n * B);
    S->R[0] = ar.take<float>(S->bs_R * B); S->R[1] = ar.take<float>(S->bs_R * B);
    S->M = fused ? nullptr : ar.take<float>(S->bs_R * B);          // unfused fallback only
    S->fbuf[0] = ar.take<float>(S->bs_f * B); S->fbuf[1] = ar.take<float>(S->bs_f * B);
    return ar.ok();
}
// level k's image size and the largest plane / row-blur scratch among the levels k_lo .. k_hi
static void fb_level_size(int H, int W, const tf_farneback_params *p, int k, int *h, int *w) {
    double scale = 1; for (int i = 0; i < k; i++) scale *= p->pyr_scale;
    *w = (int)lrint(W * scale); *h = (int)lrint(H * scale);
}
static void fb_phase_sizes(int H, int W, const tf_farneback_params *p, int k_hi, int k_lo, size_t *tmp_floats, size_t *plane) {
    size_t t = 0, pl = 0;
    for (int k = k_lo; k <= k_hi; k++) {
        int h, w; fb_level_size(H, W, p, k, &h, &w);
        pl = std::max(pl, (size_t)h * w);
        // the two-pass blur of a full-size level keeps a whole image (+ rows), the sampled blur H rows of w float2
        t = std::max(t, k == 0 ? (size_t)H * W + 2 * (size_t)H + 64 : std::max((size_t)H * W / 4 + 2 * (size_t)H + 64, 2 * (size_t)H * w + 64));
        t = std::max(t, fb_hand_floats(h, w));
    }
    if (k_lo == 0) t = std::max(t, std::max((size_t)H * W + 2 * (size_t)H + 64, fb_hand_floats(H, W)));
    *tmp_floats = t; *plane = pl;
}

// pyramid levels k_hi .. k_lo for B pairs (pointers at the first pair of the range); cur / pw / ph carry the state of the
// levels above over to the next call
static int fb_run_levels(const uint8_t *prev, const uint8_t *next, int B, int64_t img_stride, int H, int W, const tf_farneback_params *p,
                         float *const out[2], int64_t flow_stride, const FbScratch &S, int k_hi, int k_lo, int cur[2], int *pw_io, int *ph_io,
                         const FbPoly &pp, hipStream_t s)
{
    const size_t n = (size_t)H * W;
    float *tmp = S.tmp, *blur = S.blur, *I = S.I, *M = S.M;
    float *const R[2] = {S.R[0], S.R[1]};
    const int64_t bs_tmp = S.bs_tmp, bs_n = S.bs_n, bs_R = S.bs_R;
    const uint8_t *img[2] = {prev, next};
    int nd = 0, dirs[2];
    for (int d = 0; d < 2; d++) if (out[d]) dirs[nd++] = d;
    // per direction two ping-pong slots: slot 0 = the caller's output (stride flow_stride), slot 1 = scratch
    float *slot[2][2]; int64_t slot_bs[2][2];
    for (int d = 0; d < 2; d++) { slot[d][0] = out[d]; slot_bs[d][0] = flow_stride; slot[d][1] = S.fbuf[d]; slot_bs[d][1] = S.bs_f; }
    int pw = *pw_io, ph = *ph_io;
    const dim3 block(64, 4);
    const dim3 gfull((W + 63) / 64, (H + 3) / 4, B);
    const bool fused = p->win_size == FBI_WIN;
    for (int k = k_hi; k >= k_lo; k--) {
        double scale = 1; for (int i = 0; i < k; i++) scale *= p->pyr_scale;
        const double sigma = (1. / scale - 1) * 0.5;
        int smooth_sz = (int)lrint(sigma * 5) | 1; if (smooth_sz < 3) smooth_sz = 3;
        TF_REQUIRE(smooth_sz <= FB_MAX_KSIZE, "tf_farneback: blur kernel too large");
        const int w = (int)lrint(W * scale), h = (int)lrint(H * scale);
        const int64_t plane = (int64_t)w * h;
        const dim3 glev((w + 63) / 64, (h + 3) / 4, B);
        FbKernel hk; fb_gaussian_kernel(smooth_sz, sigma, &hk);
        for (int i = 0; i < 2; i++) {
            const float *Ik = blur; int64_t bs_Ik = bs_n;
            const double rsx = 1. / ((double)w / W), rsy = 1. / ((double)h / H);
            const int irx = (int)(rsx + 0.5), iry = (int)(rsy + 0.5);
            const bool same = (w == W && h == H);
            const bool area2 = !same && fabs(rsx - irx) < DBL_EPSILON && fabs(rsy - iry) < DBL_EPSILON && irx == 2 && iry == 2;
            static const bool two_pass = getenv("TF_FB_BLUR_TWOPASS") != nullptr;            // development aid
            if (area2 && hk.ksize == 3 && !two_pass && W >= 2 && H >= 2 && 2 * w <= W && 2 * h <= H) {
                TfProfScope ps(TFK_FB_BLUR, (1.0 * n + 4.0 * plane) * B, s);                   // u8 r + quarter-size f32 w
                hipLaunchKernelGGL(k_fb_blur3_area2<uint8_t>, glev, block, 0, s, img[i], H, W, hk, I, h, w, img_stride, bs_n);
                Ik = I;
            } else if (same || area2) {
                {
                    if (hk.ksize == 3 && !two_pass) {
                        TfProfScope ps(TFK_FB_BLUR, 5.0 * n * B, s);  // u8 r + f32 w
                        hipLaunchKernelGGL(k_fb_blur3_fused<uint8_t>, dim3((W + 255) / 256, (H + 3) / 4, B), block, 0, s, img[i], H, W, hk, blur, img_stride, bs_n);
                    } else {
                        TfProfScope ps(TFK_FB_BLUR, 13.0 * n * B, s); // u8 r + f32 w, then f32 r + f32 w
                        hipLaunchKernelGGL(k_fb_blur_rows<uint8_t>, gfull, block, 0, s, img[i], H, W, hk, tmp, img_stride, bs_tmp);
                        hipLaunchKernelGGL(k_fb_blur_cols, gfull, block, 0, s, tmp, H, W, hk, blur, bs_tmp, bs_n);
                    }
                }
                if (area2) {
                    TfProfScope ps(TFK_FB_RESIZE, (4.0 * n + 4.0 * plane) * B, s);
                    hipLaunchKernelGGL(k_fb_resize_area2, glev, block, 0, s, blur, W, I, h, w, bs_n, bs_n);
                    Ik = I;
                }
            } else {
                // blur + resize fused on the sampled columns / rows (tmp holds rowf: H * w float2 <= n + 2H floats)
                TfProfScope ps(TFK_FB_BLUR, (1.0 * n + 8.0 * (double)H * w * 2 + 4.0 * plane) * B, s);
                FbResizeGeom rg; rg.sh = H; rg.sw = W; rg.dh = h; rg.dw = w; rg.scale_x = rsx; rg.scale_y = rsy;
                // LDS-staged form when a workgroup's source segment fits (64 outputs * stride + ksize bytes per row)
                static const bool no_lds = getenv("TF_FB_BLUR_NO_LDS") != nullptr;                 // development aid
                const int64_t seg = (int64_t)(64 * rsx) + hk.ksize + 8;
                // (measured at 5424^2, 8 images: stride 32 / 16: 958 -> 282 / 554 -> 322 us; stride 8 / 4: 355 -> 415 / 380 -> 673 us --
                // short kernels gain nothing from staging and pay for the barrier: LDS form from stride 12 on)
                if (!no_lds && hk.ksize > 5 && rsx >= 12. && seg <= FBL_ROW_BYTES && W >= hk.ksize) {
                    int unit_shift = 2;                                 // pad unit = largest power of two <= stride, >= 8 bytes;
                    while ((2 << unit_shift) <= (int)rsx) unit_shift++; // unit 4 (shift 2) would pad every word: then no padding
                    if (unit_shift < 3) unit_shift = 30;
                    static const int word_env = getenv("TF_FB_BLUR_BYTES") ? 0 : 1;                // development aid: byte reads
                    hipLaunchKernelGGL(k_fb_blur_rows_sampled_lds, dim3((w + 63) / 64, (H + 3) / 4, B), block, 0, s, img[i], rg, hk,
                                       (float2 *)tmp, img_stride, bs_tmp / 2, unit_shift, word_env);
                } else
                    hipLaunchKernelGGL(k_fb_blur_rows_sampled<uint8_t>, dim3((w + 63) / 64, (H + 3) / 4, B), block, 0, s, img[i], rg, hk,
                                       (float2 *)tmp, img_stride, bs_tmp / 2);
                hipLaunchKernelGGL(k_fb_blur_cols_resize, glev, block, 0, s, (const float2 *)tmp, rg, hk, I, bs_tmp / 2, bs_n);
                Ik = I;
            }
            {
                TfProfScope ps(TFK_FB_POLYEXP, 24.0 * plane * B, s);   // fused-ideal: 4 r + 20 w per level pixel
                fb_launch_polyexp(Ik, h, w, pp, R[i], plane, bs_Ik, bs_R, B, s);
            }
        }
        TF_CHECK_LAUNCH();
        // level's initial flow
        for (int q = 0; q < nd; q++) {
            const int d = dirs[q];
            if (cur[d] < 0) {
                // choose the start slot so that the result of level 0 lands in slot 0 (the caller's buffer):
                // every level flips the slot num_iters times (+1 for the upsample on all but the coarsest level);
                // the unfused fallback iterates in place (no flips from iterations)
                const int flips_per_level = (fused ? p->num_iters : 0) + 1;
                const int total = (k + 1) * flips_per_level - 1;
                cur[d] = (total % 2 == 0) ? 0 : 1;
                hipLaunchKernelGGL(k_fb_zero, dim3((unsigned)((plane * 2 + 255) / 256), 1, B), dim3(256), 0, s,
                                   slot[d][cur[d]], plane * 2, slot_bs[d][cur[d]]);
            } else {
                const int src = cur[d], dst = 1 - cur[d];
                TfProfScope ps(TFK_FB_RESIZE, (8.0 * pw * ph + 8.0 * plane) * B, s);
                const double sx = 1. / ((double)w / pw), sy = 1. / ((double)h / ph);
                if (slot_bs[d][src] % 2 == 0 && slot_bs[d][dst] % 2 == 0)
                    hipLaunchKernelGGL(k_fb_resize_linear2, glev, block, 0, s, (const float2 *)slot[d][src], ph, pw, (float2 *)slot[d][dst],
                                       h, w, sx, sy, (float)(1. / p->pyr_scale), slot_bs[d][src] / 2, slot_bs[d][dst] / 2);
                else                                          // odd caller stride: items are not 8-byte aligned
                    hipLaunchKernelGGL(k_fb_resize_linear, glev, block, 0, s, slot[d][src], ph, pw, 2, slot[d][dst], h, w, sx, sy,
                                       (float)(1. / p->pyr_scale), slot_bs[d][src], slot_bs[d][dst]);
                cur[d] = dst;
            }
        }
        TF_CHECK_LAUNCH();
        if (fused) {
            // whole columns per workgroup (k_fb_iter): parallelism = strips x directions x pairs
            const int nx = (w + FBI_OW - 1) / FBI_OW;
            const dim3 gi((unsigned)(nx * B), 1, 1), bi(FBI_T * nd);
            FbIterArgs ia;
            ia.R[0] = R[0]; ia.R[1] = R[1]; ia.bs_R = bs_R; ia.nd = nd; ia.nx = nx;
            // development switch: TF_FB_ROW_SUMS_TREE=1 -> the round-3 kernel (window sums as a tree: within 1e-4 px of OpenCV's order, not identical)
            static const bool tree = getenv("TF_FB_ROW_SUMS_TREE") != nullptr;
            // TF_FBI_TWO_PART_CHAIN=1: the chain in two parts one row group apart (HP form; same flows).  Alone on the GPU it is the
            // faster form (level-0 launch of 21 pairs: 18.0 against 19.7 ms); it is NOT the default because its 39 KB of LDS per
            // workgroup fill the CU (4 x 39 = 156 of 160 KB): in the pipelined benchmark the floods of the finished windows run
            // beside the flow in what the one-lane form leaves free (4 x 27 KB), and with the two-part form they displace
            // iteration workgroups instead -- the kernel then reads 1 985 against 1 860 ms per config-F step, the step is the same
            // (4.75 s either way, back to back on one box)
            // (tf_farneback_params.chain_form: PER CALL -- the host layer asks for the two-part form when nothing is going to run
            // beside this call's flow; a process-wide switch, as in round 4, let one thread's plain create_flow turn it on under
            // another thread's pipelined one)
            static const char *chain_env = getenv("TF_FBI_TWO_PART_CHAIN");
            const bool whole_chain = chain_env ? atoi(chain_env) == 0 : p->chain_form != TF_FB_CHAIN_TWO_PART;
            // sequential row sums: the strips' hand-over words and the launches' ticket counters live in the blur scratch,
            // idle from the polynomial expansion of this level to the blur of the next: [1 KB of counters][words] per pair
            const size_t hand_words = (size_t)nd * (size_t)(nx - 1) * (size_t)h * FBI_HW;
            static const int seq_abl = getenv("TF_FBI_SEQ_ABLATE") ? atoi(getenv("TF_FBI_SEQ_ABLATE")) : 0;
            ia.abl = seq_abl;
            {
                // column groups: one per round of resident workgroups (two 4-wave workgroups per CU)
                const int slots = fb_resident_slots();                       // resident two-wave workgroups (one direction each) of THIS device
                // one direction per workgroup (four two-wave workgroups per CU) unless TF_FBI_JOIN_DIRECTIONS=1: the chains make a
                // workgroup latency-bound for a third of its time, and four independent workgroups per CU overlap those
                // stretches better than two (config F: 2.05 s of k_fb_iter per step against 2.25 s), at the price of reading the
                // R rows once per direction again (round 3 joined the directions for that: -23 % HBM bytes, same time)
                static const bool join_env = getenv("TF_FBI_JOIN_DIRECTIONS") != nullptr;
                ia.nq = (!join_env && nd == 2) ? 2 : 1;
                static const bool one_list_env = getenv("TF_FBI_ONE_TICKET_LIST") != nullptr;
                ia.xcd_lists = one_list_env ? 0 : 1;
                const int rounds = (int)(((int64_t)nx * B * 2 + slots - 1) / slots);   // (a joined workgroup counts as two)
                static const int slack_env = getenv("TF_FBI_SLACK_ROWS") ? atoi(getenv("TF_FBI_SLACK_ROWS")) : -1;
                static const int groups_env = getenv("TF_FBI_COLUMN_GROUPS") ? atoi(getenv("TF_FBI_COLUMN_GROUPS")) : 0;
                const int n_groups = groups_env > 0 ? std::min(groups_env, nx) : std::min(rounds, nx);
                ia.nb = B; ia.nxg = (nx + n_groups - 1) / n_groups; ia.slack_rows = slack_env >= 0 ? slack_env : 0;
            }
            ia.hand = (unsigned long long *)((char *)tmp + FBI_HDR); ia.bs_hand = bs_tmp / 2; ia.ticket = (int *)tmp; ia.epoch = 0;
            static const int poll_env = getenv("TF_FBI_POLL_LIMIT") ? atoi(getenv("TF_FBI_POLL_LIMIT")) : -1;
            ia.poll_limit = poll_env >= 0 ? poll_env : (1 << 22);
            ia.starved = tree ? nullptr : fb_starved_word(true, p->status_slot);
            if (!tree && !ia.starved) { tf_set_error("tf_farneback: no pinned status word for the iteration kernel (hipHostMalloc failed)"); return TF_EHIP; }
            if (!tree) {
                TF_REQUIRE(FBI_HDR + hand_words * 8 <= (size_t)bs_tmp * sizeof(float), "tf_farneback: blur scratch too small for the strips' hand-over words");
                TF_REQUIRE(p->num_iters <= 250, "tf_farneback: more than 250 iterations per level");
                TF_CHECK_HIP(hipMemsetAsync(tmp, 0, FBI_HDR, s));
                for (int b = 0; b < B && hand_words > 0; b++) TF_CHECK_HIP(hipMemsetAsync((char *)(tmp + (int64_t)b * bs_tmp) + FBI_HDR, 0, hand_words * 8, s));
            }
            for (int it = 0; it < p->num_iters; it++) {
                for (int q = 0; q < nd; q++) {
                    const int d = dirs[q];
                    ia.dir[q] = d;
                    ia.fin[q] = slot[d][cur[d]]; ia.bs_fin[q] = slot_bs[d][cur[d]];
                    ia.fout[q] = slot[d][1 - cur[d]]; ia.bs_fout[q] = slot_bs[d][1 - cur[d]];
                }
                {
                    TfProfScope ps(TFK_FB_ITER, 56.0 * plane * nd * B, s);
                    static const int abl = getenv("TF_FBI_ABLATE") ? atoi(getenv("TF_FBI_ABLATE")) : 0;
                    ia.epoch = (unsigned)(it + 1); ia.ticket = (int *)tmp + 16 * it;
                    if (tree) {
                        if (abl == 1) hipLaunchKernelGGL((k_fb_iter_tree<FBI_NB, 1>), gi, bi, 0, s, ia, h, w, plane);
                        else if (abl == 2) hipLaunchKernelGGL((k_fb_iter_tree<FBI_NB, 2>), gi, bi, 0, s, ia, h, w, plane);
                        else hipLaunchKernelGGL((k_fb_iter_tree<FBI_NB, 0>), gi, bi, 0, s, ia, h, w, plane);
                    } else if (abl == 1) hipLaunchKernelGGL((k_fb_iter<FBI_NB, 1, 2, 0>), gi, bi, 0, s, ia, h, w, plane);
                    else if (abl == 2) hipLaunchKernelGGL((k_fb_iter<FBI_NB, 2, 2, 0>), gi, bi, 0, s, ia, h, w, plane);
                    else if (whole_chain) {
                        if (ia.nq == 2) hipLaunchKernelGGL((k_fb_iter<FBI_NB, 0, 1, 0>), dim3(gi.x * 2), dim3(FBI_T), 0, s, ia, h, w, plane);
                        else if (nd == 1) hipLaunchKernelGGL((k_fb_iter<FBI_NB, 0, 1, 0>), gi, bi, 0, s, ia, h, w, plane);
                        else hipLaunchKernelGGL((k_fb_iter<FBI_NB, 0, 2, 0>), gi, bi, 0, s, ia, h, w, plane);
                    } else if (ia.nq == 2) hipLaunchKernelGGL((k_fb_iter<FBI_NB, 0, 1, 1>), dim3(gi.x * 2), dim3(FBI_T), 0, s, ia, h, w, plane);
                    else if (nd == 1) hipLaunchKernelGGL((k_fb_iter<FBI_NB, 0, 1, 1>), gi, bi, 0, s, ia, h, w, plane);
                    else hipLaunchKernelGGL((k_fb_iter<FBI_NB, 0, 2, 1>), gi, bi, 0, s, ia, h, w, plane);
                }
                for (int q = 0; q < nd; q++) cur[dirs[q]] = 1 - cur[dirs[q]];
            }
            TF_CHECK_LAUNCH();
            if (!tree && (seq_abl & 8)) {                               // development aid: how often did a chain have to wait?
                int h_sp[2] = {0, 0};
                TF_CHECK_HIP(hipMemcpyAsync(h_sp, (int *)tmp + FBI_HDR / 4 - 4, sizeof(h_sp), hipMemcpyDeviceToHost, s));
                TF_CHECK_HIP(hipStreamSynchronize(s));
                fprintf(stderr, "k_fb_iter %d x %d, %d pairs, %d launches: %d chains waited (of %lld), %d polls\n", h, w, B, p->num_iters, h_sp[0],
                        (long long)B * nd * (nx - 1) * h * 5 * p->num_iters, h_sp[1]);
            }
        } else {
            // generic window size: separate UpdateMatrices / box-filter+solve kernels, one pair at a time, in place
            const dim3 glev1((w + 63) / 64, (h + 3) / 4, 1);
            const dim3 gt((w + FBT_W - 1) / FBT_W, (h + FBT_H - 1) / FBT_H);
            for (int b = 0; b < B; b++)
                for (int q = 0; q < nd; q++) {
                    const int d = dirs[q];
                    float *flow = slot[d][cur[d]] + (int64_t)b * slot_bs[d][cur[d]];
                    const float *R0 = R[d] + (int64_t)b * bs_R, *R1 = R[1 - d] + (int64_t)b * bs_R;
                    float *Mb = M + (int64_t)b * bs_R;
                    {
                        TfProfScope ps(TFK_FB_MATRICES, 68.0 * plane, s);
                        hipLaunchKernelGGL(k_fb_update_matrices, glev1, block, 0, s, R0, R1, flow, h, w, plane, Mb);
                    }
                    for (int it = 0; it < p->num_iters; it++) {
                        {
                            TfProfScope ps(TFK_FB_BLUR_SOLVE, 28.0 * plane, s);
                            hipLaunchKernelGGL(k_fb_blur_solve, gt, block, 0, s, Mb, h, w, plane, p->win_size / 2, flow);
                        }
                        if (it < p->num_iters - 1) {
                            TfProfScope ps(TFK_FB_MATRICES, 68.0 * plane, s);
                            hipLaunchKernelGGL(k_fb_update_matrices, glev1, block, 0, s, R0, R1, flow, h, w, plane, Mb);
                        }
                    }
                }
            TF_CHECK_LAUNCH();
        }
        pw = w; ph = h;
    }
    *pw_io = pw; *ph_io = ph;
    return TF_OK;
}

// levels >= FB_SPLIT_LEVEL of a split batch run for all its pairs at once, the finer ones in parts
#define FB_SPLIT_LEVEL 2
extern "C" size_t tf_farneback_workspace_bytes_split(int64_t B, int64_t parts, int64_t H, int64_t W, const tf_farneback_params *p)
{
    if (B <= 0 || H <= 0 || W <= 0 || !p) return 0;
    const int levels = fb_levels(H, W, p);
    const bool fused = p->win_size == FBI_WIN;
    if (parts <= 1 || levels < FB_SPLIT_LEVEL || B < 2 || p->pyr_scale != 0.5) return tf_farneback_workspace_bytes_batch(B, H, W, p);
    if (parts > B) parts = B;
    size_t t = 0, pl = 0;
    fb_phase_sizes((int)H, (int)W, p, levels, FB_SPLIT_LEVEL, &t, &pl);
    const size_t coarse = (size_t)B * fb_scratch_floats(t, pl, fused) * sizeof(float) + 8192;
    const size_t fine = tf_farneback_workspace_bytes_batch((B + parts - 1) / parts, H, W, p);
    return coarse > fine ? coarse : fine;
}

extern "C" int tf_farneback_batch_split(const uint8_t *prev, const uint8_t *next, int64_t B64, int64_t parts64, int64_t img_stride,
                                        int64_t H64, int64_t W64, const tf_farneback_params *p,
                                        float *flow_fwd, float *flow_bwd, int64_t flow_stride,
                                        void *ws, size_t ws_bytes, void *stream)
{
    TF_REQUIRE(prev && next && p && ws, "tf_farneback: null pointer");
    TF_REQUIRE(flow_fwd || flow_bwd, "tf_farneback: both outputs are NULL");
    TF_REQUIRE(B64 >= 1 && B64 <= 1024 && parts64 >= 1, "tf_farneback: bad batch size");
    TF_REQUIRE(H64 > 0 && W64 > 0 && H64 < (1 << 15) && W64 < (1 << 15), "tf_farneback: bad shape");
    TF_REQUIRE(img_stride >= H64 * W64 && flow_stride >= H64 * W64 * 2, "tf_farneback: strides smaller than one frame");
    TF_REQUIRE(p->poly_n >= 1 && p->poly_n <= FB_MAX_POLY_N, "tf_farneback: poly_n out of range");
    TF_REQUIRE(p->win_size >= 1 && p->win_size / 2 <= FB_MAX_M, "tf_farneback: win_size out of range");
    TF_REQUIRE(p->num_iters >= 1 && p->num_levels >= 0 && p->pyr_scale > 0 && p->pyr_scale < 1, "tf_farneback: bad params");
    if (ws_bytes < tf_farneback_workspace_bytes_split(B64, parts64, H64, W64, p)) { tf_set_error("tf_farneback: workspace too small"); return TF_ENOMEM; }
    if (const int rc0 = fb_report_starved("tf_farneback (an earlier call's launches)", p->status_slot)) return rc0;
    const int H = (int)H64, W = (int)W64, B = (int)B64;
    hipStream_t s = (hipStream_t)stream;
    const bool fused = p->win_size == FBI_WIN;
    FbPoly pp; fb_prepare_poly(p->poly_n, p->poly_sigma, &pp);
    const int levels = fb_levels(H, W, p);
    float *const out[2] = {flow_fwd, flow_bwd};
    int cur[2] = {-1, -1};                       // slot index that holds the current flow of direction d
    int pw = 0, ph = 0;
    int parts = (int)(parts64 > B ? B : parts64);
    if (levels < FB_SPLIT_LEVEL || B < 2 || p->pyr_scale != 0.5) parts = 1;     // (only then is every coarse level a sampled blur: no full-size plane)
    if (parts == 1) {
        TfArena ar(ws, ws_bytes);
        FbScratch S;
        if (!fb_carve(ar, B, std::max((size_t)H * W + 2 * (size_t)H + 64, fb_hand_floats(H, W)), (size_t)H * W, fused, &S)) { tf_set_error("tf_farneback: workspace too small"); return TF_ENOMEM; }
        const int rc = fb_run_levels(prev, next, B, img_stride, H, W, p, out, flow_stride, S, levels, 0, cur, &pw, &ph, pp, s);
        if (rc) return rc;
    } else {
        // SPLIT BATCH (round 4).  A launch of the iteration kernel costs whole rounds of resident workgroups; at the coarse
        // levels a round holds the strips of ~40 pairs, at full resolution those of ~10 -- but the scratch of a pair is sized by
        // its full-resolution planes.  So the levels >= 2 run for ALL pairs of the batch at once, with scratch strides of the
        // level-2 plane (1 / 16 of a full one), and leave their flow in the caller's output frames; the two finest levels then
        // run part by part on full-size scratch for B / parts pairs.  Same kernels on the same data: the results are those
        // of the unsplit batch.  (Every level flips the ping-pong slot num_iters + 1 times, and two levels remain: the flow
        // of level 2 is in slot 0, the caller's buffer, whatever the parity of num_iters.)
        {
            size_t t = 0, pl = 0;
            fb_phase_sizes(H, W, p, levels, FB_SPLIT_LEVEL, &t, &pl);
            TfArena ar(ws, ws_bytes);
            FbScratch S;
            if (!fb_carve(ar, B, t, pl, fused, &S)) { tf_set_error("tf_farneback: workspace too small"); return TF_ENOMEM; }
            const int rc = fb_run_levels(prev, next, B, img_stride, H, W, p, out, flow_stride, S, levels, FB_SPLIT_LEVEL, cur, &pw, &ph, pp, s);
            if (rc) return rc;
            for (int d = 0; d < 2; d++) if (out[d] && cur[d] != 0) { tf_set_error("tf_farneback: internal slot parity error (split)"); return TF_EINVAL; }
        }
        const int per = (B + parts - 1) / parts;
        for (int b0 = 0; b0 < B; b0 += per) {
            const int Bp = B - b0 < per ? B - b0 : per;
            TfArena ar(ws, ws_bytes);                                 // (the coarse phase's scratch is dead: stream order)
            FbScratch S;
            if (!fb_carve(ar, Bp, std::max((size_t)H * W + 2 * (size_t)H + 64, fb_hand_floats(H, W)), (size_t)H * W, fused, &S)) { tf_set_error("tf_farneback: workspace too small"); return TF_ENOMEM; }
            float *const outp[2] = {flow_fwd ? flow_fwd + (int64_t)b0 * flow_stride : nullptr, flow_bwd ? flow_bwd + (int64_t)b0 * flow_stride : nullptr};
            int curp[2] = {cur[0], cur[1]}, pwp = pw, php = ph;
            const int rc = fb_run_levels(prev + (int64_t)b0 * img_stride, next + (int64_t)b0 * img_stride, Bp, img_stride, H, W, p, outp, flow_stride, S,
                                         FB_SPLIT_LEVEL - 1, 0, curp, &pwp, &php, pp, s);
            if (rc) return rc;
            if (b0 + per >= B) { cur[0] = curp[0]; cur[1] = curp[1]; }
        }
    }
    for (int d = 0; d < 2; d++)
        if (out[d] && cur[d] != 0) { tf_set_error("tf_farneback: internal slot parity error"); return TF_EINVAL; }
    return TF_OK;
}

extern "C" int tf_farneback_batch(const uint8_t *prev, const uint8_t *next, int64_t B64, int64_t img_stride,
                                  int64_t H64, int64_t W64, const tf_farneback_params *p,
                                  float *flow_fwd, float *flow_bwd, int64_t flow_stride,
                                  void *ws, size_t ws_bytes, void *stream)
{
    return tf_farneback_batch_split(prev, next, B64, 1, img_stride, H64, W64, p, flow_fwd, flow_bwd, flow_stride, ws, ws_bytes, stream);
}

// The two halves of a split batch as calls of their own, for a caller that does something between the parts (create_flow
// refines and smooths a part and hands its frames out before the next part's finest levels run):
//   phase 1  pyramid levels >= 2 for B pairs; their flow is left in the output frames
//   phase 2  levels 1 and 0 for B pairs whose output frames hold the flow of level 2 (a part of the pairs of phase 1)
extern "C" int tf_farneback_can_split(int64_t H, int64_t W, const tf_farneback_params *p)
{
    return (p && H > 0 && W > 0 && fb_levels(H, W, p) >= FB_SPLIT_LEVEL && p->pyr_scale == 0.5) ? 1 : 0;
}
extern "C" size_t tf_farneback_workspace_bytes_phase(int64_t B, int64_t H, int64_t W, const tf_farneback_params *p, int phase)
{
    if (B <= 0 || H <= 0 || W <= 0 || !p || !tf_farneback_can_split(H, W, p)) return 0;
    if (phase == 2) return tf_farneback_workspace_bytes_batch(B, H, W, p);
    size_t t = 0, pl = 0;
    fb_phase_sizes((int)H, (int)W, p, fb_levels(H, W, p), FB_SPLIT_LEVEL, &t, &pl);
    return (size_t)B * fb_scratch_floats(t, pl, p->win_size == FBI_WIN) * sizeof(float) + 8192;
}
extern "C" int tf_farneback_batch_phase(const uint8_t *prev, const uint8_t *next, int64_t B64, int64_t img_stride,
                                        int64_t H64, int64_t W64, const tf_farneback_params *p,
                                        float *flow_fwd, float *flow_bwd, int64_t flow_stride,
                                        void *ws, size_t ws_bytes, void *stream, int phase)
{
    TF_REQUIRE(prev && next && p && ws, "tf_farneback: null pointer");
    TF_REQUIRE(flow_fwd || flow_bwd, "tf_farneback: both outputs are NULL");
    TF_REQUIRE(B64 >= 1 && B64 <= 1024 && (phase == 1 || phase == 2), "tf_farneback: bad batch size / phase");
    TF_REQUIRE(H64 > 0 && W64 > 0 && H64 < (1 << 15) && W64 < (1 << 15), "tf_farneback: bad shape");
    TF_REQUIRE(img_stride >= H64 * W64 && flow_stride >= H64 * W64 * 2, "tf_farneback: strides smaller than one frame");
    TF_REQUIRE(p->poly_n >= 1 && p->poly_n <= FB_MAX_POLY_N, "tf_farneback: poly_n out of range");
    TF_REQUIRE(p->win_size >= 1 && p->win_size / 2 <= FB_MAX_M, "tf_farneback: win_size out of range");
    TF_REQUIRE(p->num_iters >= 1 && p->num_levels >= 0 && p->pyr_scale > 0 && p->pyr_scale < 1, "tf_farneback: bad params");
    TF_REQUIRE(tf_farneback_can_split(H64, W64, p), "tf_farneback_batch_phase: this geometry does not split (tf_farneback_can_split)");
    if (ws_bytes < tf_farneback_workspace_bytes_phase(B64, H64, W64, p, phase)) { tf_set_error("tf_farneback: workspace too small"); return TF_ENOMEM; }
    if (const int rc0 = fb_report_starved("tf_farneback (an earlier call's launches)", p->status_slot)) return rc0;
    const int H = (int)H64, W = (int)W64, B = (int)B64;
    hipStream_t s = (hipStream_t)stream;
    const bool fused = p->win_size == FBI_WIN;
    FbPoly pp; fb_prepare_poly(p->poly_n, p->poly_sigma, &pp);
    const int levels = fb_levels(H, W, p);
    float *const out[2] = {flow_fwd, flow_bwd};
    TfArena ar(ws, ws_bytes);
    FbScratch S;
    if (phase == 1) {
        size_t t = 0, pl = 0;
        fb_phase_sizes(H, W, p, levels, FB_SPLIT_LEVEL, &t, &pl);
        if (!fb_carve(ar, B, t, pl, fused, &S)) { tf_set_error("tf_farneback: workspace too small"); return TF_ENOMEM; }
        int cur[2] = {-1, -1}, pw = 0, ph = 0;
        const int rc = fb_run_levels(prev, next, B, img_stride, H, W, p, out, flow_stride, S, levels, FB_SPLIT_LEVEL, cur, &pw, &ph, pp, s);
        if (rc) return rc;
        for (int d = 0; d < 2; d++) if (out[d] && cur[d] != 0) { tf_set_error("tf_farneback: internal slot parity error (phase 1)"); return TF_EINVAL; }
        return TF_OK;
    }
    if (!fb_carve(ar, B, std::max((size_t)H * W + 2 * (size_t)H + 64, fb_hand_floats(H, W)), (size_t)H * W, fused, &S)) { tf_set_error("tf_farneback: workspace too small"); return TF_ENOMEM; }
    int cur[2] = {0, 0}, pw = 0, ph = 0;                              // the flow of level FB_SPLIT_LEVEL sits in slot 0 = the output frames
    fb_level_size(H, W, p, FB_SPLIT_LEVEL, &ph, &pw);
    const int rc = fb_run_levels(prev, next, B, img_stride, H, W, p, out, flow_stride, S, FB_SPLIT_LEVEL - 1, 0, cur, &pw, &ph, pp, s);
    if (rc) return rc;
    for (int d = 0; d < 2; d++) if (out[d] && cur[d] != 0) { tf_set_error("tf_farneback: internal slot parity error (phase 2)"); return TF_EINVAL; }
    return TF_OK;
}

extern "C" int tf_farneback_pair(const uint8_t *prev, const uint8_t *next, int64_t H, int64_t W,
                                 const tf_farneback_params *p, float *flow_fwd, float *flow_bwd,
                                 void *ws, size_t ws_bytes, void *stream)
{
    return tf_farneback_batch(prev, next, 1, H * W, H, W, p, flow_fwd, flow_bwd, H * W * 2, ws, ws_bytes, stream);
}

// Diagnostic / test entry: the full-resolution level's image (3 x 3 Gaussian of the uint8 frame) and its polynomial
// expansion, as the pyramid loop computes them -- so that the stages below the iteration can be compared with the oracle
// one by one (tests/test_gpu_parity.py).  blur_out: H * W floats or NULL; R_out: 5 * H * W floats, the library's layout
// (H * W float4 {r0, r1, r2, r3} followed by one plane r4; OpenCV's 5 interleaved channels in the same order).
extern "C" int tf_farneback_expansion(const uint8_t *img, int64_t H64, int64_t W64, const tf_farneback_params *p,
                                      float *blur_out, float *R_out, void *stream)
{
    TF_REQUIRE(img && p && R_out, "tf_farneback_expansion: null pointer");
    TF_REQUIRE(H64 > 0 && W64 > 0 && H64 < (1 << 15) && W64 < (1 << 15), "tf_farneback_expansion: bad shape");
    TF_REQUIRE(p->poly_n >= 1 && p->poly_n <= FB_MAX_POLY_N, "tf_farneback_expansion: poly_n out of range");
    const int H = (int)H64, W = (int)W64;
    hipStream_t s = (hipStream_t)stream;
    FbPoly pp; fb_prepare_poly(p->poly_n, p->poly_sigma, &pp);
    FbKernel hk; fb_gaussian_kernel(3, 0.0, &hk);
    float *blur = blur_out;
    if (!blur) TF_CHECK_HIP(hipMallocAsync((void **)&blur, (size_t)H * W * sizeof(float), s));
    const dim3 block(64, 4);
    hipLaunchKernelGGL(k_fb_blur3_fused<uint8_t>, dim3((W + 255) / 256, (H + 3) / 4, 1), block, 0, s, img, H, W, hk, blur, (int64_t)H * W, (int64_t)H * W);
    fb_launch_polyexp((const float *)blur, H, W, pp, R_out, (int64_t)H * W, (int64_t)H * W, (int64_t)5 * H * W, 1, s);
    TF_CHECK_LAUNCH();
    if (!blur_out) TF_CHECK_HIP(hipFreeAsync(blur, s));
    return TF_OK;
}
