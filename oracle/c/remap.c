/* ORACLE (test infrastructure, never shipped or measured as the product):
 * CPU restatement of cv2.remap as the reference calls it
 *   /root/reference/tobac_flow/convolve.py:65-84      (batched multi-offset warp)
 *   /root/reference/tobac_flow/utils/flow_utils.py:90-98 (single-image warp)
 * i.e. map1 = CV_32FC2 absolute coordinates, map2 = None, interpolation in
 * {INTER_NEAREST, INTER_LINEAR, INTER_CUBIC, INTER_LANCZOS4}, BORDER_CONSTANT with a scalar border value.
 *
 * OpenCV is a third-party dependency that is ABSENT from /root/reference and from this
 * image (environment.yml:15 `opencv`, unpinned).  This file restates the published
 * algorithm of modules/imgproc/src/imgwarp.cpp (OpenCV 4.x) from upstream knowledge
 * (SURVEY.md Appendix A.3):
 *   - non-nearest modes first quantise coordinates to 1/32 px: sx = cvRound(x*32),
 *     integer part sx >> 5, table index sx & 31;
 *   - bilinear weights (1-f, f), bicubic weights with A = -0.75, 2-D weight = product,
 *     all in float; accumulation in float, taps row-major;
 *   - BORDER_CONSTANT: patch fully inside -> plain sum; fully outside -> border value;
 *     straddling -> outside taps take the border value (bilinear) or
 *     sum = cval + SUM (S - cval) * w over inside taps (bicubic);
 *   - INTER_NEAREST: ix = cvRound(x) (half to even), outside -> border value;
 *   - INTER_LANCZOS4 (interp 3): 8 x 8 taps from (sx - 3, sy - 3), 1-D weights interpolateLanczos4(k / 32) (a = 4 Lanczos
 *     window evaluated with the sin / cos addition trick, normalised to unit sum in float; f < FLT_EPSILON -> delta),
 *     row sums added row by row; border like bicubic.
 * PARITY STATUS: pinned only by the reference's own known-answer tests
 * (tests/test_flow.py:94-161: identity, integer shifts, exact half-pixel mean); cubic /
 * fractional-nearest / border behaviour are "parity unpinned" (no cv2 in this image).
 */
#include <math.h>
#include <stdint.h>

#define INTER_BITS 5
#define INTER_TAB_SIZE 32

static inline int cv_round_f(float v) { return (int)lrintf(v); }       /* round half to even */
static inline int16_t sat_short(int v) { return (int16_t)(v < -32768 ? -32768 : (v > 32767 ? 32767 : v)); }

static void cubic_coeffs(float x, float *c) {
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
}

static void lanczos4_coeffs(float x, float *coeffs) {
    static const double s45 = 0.70710678118654752440084436210485;
    static const double cs[8][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
    if (x < 1.1920928955078125e-07f) { for (int i = 0; i < 8; i++) coeffs[i] = 0; coeffs[3] = 1; return; }
    float sum = 0;
    double y0 = -(x + 3) * 3.14159265358979323846 * 0.25, s0 = sin(y0), c0 = cos(y0);
    for (int i = 0; i < 8; i++) {
        double y = -(x + 3 - i) * 3.14159265358979323846 * 0.25;
        coeffs[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
        sum += coeffs[i];
    }
    sum = 1.f / sum;
    for (int i = 0; i < 8; i++) coeffs[i] *= sum;
}

/* interp: 0 nearest, 1 linear, 2 cubic, 3 lanczos4.  map: rows*cols*2 floats (x, y). */
void oracle_remap_f32(const float *img, int h, int w, const float *map, int64_t rows, int64_t cols,
                      int interp, float cval, float *dst)
{
    float lin[INTER_TAB_SIZE][2], cub[INTER_TAB_SIZE][4], lan[INTER_TAB_SIZE][8];
    for (int i = 0; i < INTER_TAB_SIZE; i++) {
        float f = (float)i * (1.f / INTER_TAB_SIZE);
        lin[i][0] = 1.f - f; lin[i][1] = f;
        cubic_coeffs(f, cub[i]);
        lanczos4_coeffs(f, lan[i]);
    }
    for (int64_t r = 0; r < rows; r++)
        for (int64_t c = 0; c < cols; c++) {
            const float mx = map[(r * cols + c) * 2], my = map[(r * cols + c) * 2 + 1];
            float *d = dst + r * cols + c;
            if (interp == 0) {
                int sx = sat_short(cv_round_f(mx)), sy = sat_short(cv_round_f(my));
                *d = ((unsigned)sx < (unsigned)w && (unsigned)sy < (unsigned)h) ? img[(int64_t)sy * w + sx] : cval;
                continue;
            }
            int fx = cv_round_f(mx * (float)INTER_TAB_SIZE), fy = cv_round_f(my * (float)INTER_TAB_SIZE);
            int sx = sat_short(fx >> INTER_BITS), sy = sat_short(fy >> INTER_BITS);
            int ax = fx & (INTER_TAB_SIZE - 1), ay = fy & (INTER_TAB_SIZE - 1);
            if (interp == 1) {
                float wt[4] = { lin[ay][0] * lin[ax][0], lin[ay][0] * lin[ax][1],
                                lin[ay][1] * lin[ax][0], lin[ay][1] * lin[ax][1] };
                int w1 = w - 1 > 0 ? w - 1 : 0, h1 = h - 1 > 0 ? h - 1 : 0;
                if ((unsigned)sx < (unsigned)w1 && (unsigned)sy < (unsigned)h1) {
                    const float *S = img + (int64_t)sy * w + sx;
                    *d = S[0] * wt[0] + S[1] * wt[1] + S[w] * wt[2] + S[w + 1] * wt[3];
                } else if (sx >= w || sx + 1 < 0 || sy >= h || sy + 1 < 0) {
                    *d = cval;
                } else {
                    int sx0 = sx, sx1 = sx + 1, sy0 = sy, sy1 = sy + 1;
                    int okx0 = sx0 >= 0 && sx0 < w, okx1 = sx1 >= 0 && sx1 < w;
                    int oky0 = sy0 >= 0 && sy0 < h, oky1 = sy1 >= 0 && sy1 < h;
                    float v0 = (okx0 && oky0) ? img[(int64_t)sy0 * w + sx0] : cval;
                    float v1 = (okx1 && oky0) ? img[(int64_t)sy0 * w + sx1] : cval;
                    float v2 = (okx0 && oky1) ? img[(int64_t)sy1 * w + sx0] : cval;
                    float v3 = (okx1 && oky1) ? img[(int64_t)sy1 * w + sx1] : cval;
                    *d = v0 * wt[0] + v1 * wt[1] + v2 * wt[2] + v3 * wt[3];
                }
            } else if (interp == 3) {
                float wt[64];
                for (int i = 0; i < 8; i++) for (int j = 0; j < 8; j++) wt[i * 8 + j] = lan[ay][i] * lan[ax][j];
                int bx = sx - 3, by = sy - 3;
                int w1 = w - 7 > 0 ? w - 7 : 0, h1 = h - 7 > 0 ? h - 7 : 0;
                if ((unsigned)bx < (unsigned)w1 && (unsigned)by < (unsigned)h1) {
                    const float *S = img + (int64_t)by * w + bx;
                    const float *wp = wt;
                    float sum = 0;
                    for (int r = 0; r < 8; r++, S += w, wp += 8)
                        sum += S[0] * wp[0] + S[1] * wp[1] + S[2] * wp[2] + S[3] * wp[3]
                             + S[4] * wp[4] + S[5] * wp[5] + S[6] * wp[6] + S[7] * wp[7];
                    *d = sum;
                } else if (bx >= w || bx + 8 <= 0 || by >= h || by + 8 <= 0) {
                    *d = cval;
                } else {
                    float sum = cval * 1.f;
                    for (int i = 0; i < 8; i++) {
                        int yi = by + i;
                        if (yi < 0 || yi >= h) continue;
                        for (int j = 0; j < 8; j++) {
                            int xj = bx + j;
                            if (xj >= 0 && xj < w) sum += (img[(int64_t)yi * w + xj] - cval) * wt[i * 8 + j];
                        }
                    }
                    *d = sum;
                }
            } else {
                float wt[16];
                for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) wt[i * 4 + j] = cub[ay][i] * cub[ax][j];
                int bx = sx - 1, by = sy - 1;
                int w1 = w - 3 > 0 ? w - 3 : 0, h1 = h - 3 > 0 ? h - 3 : 0;
                if ((unsigned)bx < (unsigned)w1 && (unsigned)by < (unsigned)h1) {
                    const float *S = img + (int64_t)by * w + bx;
                    float sum = S[0] * wt[0] + S[1] * wt[1] + S[2] * wt[2] + S[3] * wt[3]
                              + S[w] * wt[4] + S[w + 1] * wt[5] + S[w + 2] * wt[6] + S[w + 3] * wt[7]
                              + S[2 * w] * wt[8] + S[2 * w + 1] * wt[9] + S[2 * w + 2] * wt[10] + S[2 * w + 3] * wt[11]
                              + S[3 * w] * wt[12] + S[3 * w + 1] * wt[13] + S[3 * w + 2] * wt[14] + S[3 * w + 3] * wt[15];
                    *d = sum;
                } else if (bx >= w || bx + 4 <= 0 || by >= h || by + 4 <= 0) {
                    *d = cval;
                } else {
                    float sum = cval * 1.f;
                    for (int i = 0; i < 4; i++) {
                        int yi = by + i;
                        if (yi < 0 || yi >= h) continue;
                        for (int j = 0; j < 4; j++) {
                            int xj = bx + j;
                            if (xj >= 0 && xj < w) sum += (img[(int64_t)yi * w + xj] - cval) * wt[i * 4 + j];
                        }
                    }
                    *d = sum;
                }
            }
        }
}

/* INTER_NEAREST on int32 images (label warps, /root/reference/tobac_flow/label.py:135-137) */
void oracle_remap_nearest_i32(const int32_t *img, int h, int w, const float *map, int64_t rows, int64_t cols,
                              int32_t cval, int32_t *dst)
{
    for (int64_t i = 0; i < rows * cols; i++) {
        int sx = sat_short(cv_round_f(map[i * 2])), sy = sat_short(cv_round_f(map[i * 2 + 1]));
        dst[i] = ((unsigned)sx < (unsigned)w && (unsigned)sy < (unsigned)h) ? img[(int64_t)sy * w + sx] : cval;
    }
}
