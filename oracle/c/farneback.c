/* ORACLE (test infrastructure, never shipped or measured as the product):
 * CPU restatement of the dense optical flow the reference obtains from
 *   cv2.optflow.createOptFlow_Farneback().calc(prev, next, None)
 *   (/root/reference/tobac_flow/utils/flow_utils.py:52-53, /root/reference/tobac_flow/flow.py:511,516)
 * with OpenCV's defaults numLevels=5, pyrScale=0.5, fastPyramids=false, winSize=13,
 * numIters=10, polyN=5, polySigma=1.1, flags=0 (box window).
 *
 * OpenCV (conda `opencv`, UNPINNED in /root/reference/environment.yml:15) is a third-party
 * dependency absent from /root/reference and from this image.  This file restates the
 * published algorithm of modules/video/src/optflowgf.cpp (+ GaussianBlur / resize from
 * imgproc) from upstream knowledge, keeping OpenCV's loop structure (float/double placement,
 * replicate borders, running box sums in double, striped matrix refresh == two passes);
 * see SURVEY.md Appendix A.1.
 *
 * PARITY STATUS: "parity unpinned" -- the reference holds no golden vector for Farnebaeck
 * (its flow tests use the DIS model, tests/test_flow.py:198-360) and cv2 cannot be run
 * here.  What IS checked: analytic translations recover the shift; the HIP path matches
 * this restatement to <= 1e-4 px.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int cv_round_d(double v) { return (int)lrint(v); }
static inline int cv_floor_f(float v) { int i = (int)v; return i - (v < (float)i); }
static inline int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) { if (p < 0) p = -p; else p = 2 * len - 2 - p; }
    return p;
}

/* cv::getGaussianKernel(n, sigma, CV_32F) */
static void gaussian_kernel(int n, double sigma, float *k) {
    static const float small_tab[4][7] = {
        {1.f}, {0.25f, 0.5f, 0.25f}, {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f},
        {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f}};
    const float *fixed = (n % 2 == 1 && n <= 7 && sigma <= 0) ? small_tab[n >> 1] : 0;
    double sigmaX = sigma > 0 ? sigma : ((n - 1) * 0.5 - 1) * 0.3 + 0.8;
    double scale2X = -0.5 / (sigmaX * sigmaX), sum = 0;
    for (int i = 0; i < n; i++) {
        double x = i - (n - 1) * 0.5;
        double t = fixed ? (double)fixed[i] : exp(scale2X * x * x);
        k[i] = (float)t; sum += k[i];
    }
    sum = 1. / sum;
    for (int i = 0; i < n; i++) k[i] = (float)(k[i] * sum);
}

/* cv::GaussianBlur(src f32, ksize x ksize, sigma) with BORDER_REFLECT_101: separable, row pass
 * first (sequential taps; symmetric-small formula for ksize <= 5), then symmetric column pass. */
static void gaussian_blur(const float *src, int h, int w, int ksize, double sigma, float *dst) {
    float *k = (float *)malloc(ksize * sizeof(float));
    gaussian_kernel(ksize, sigma, k);
    int r = ksize / 2;
    float *tmp = (float *)malloc((size_t)h * w * sizeof(float));
    for (int y = 0; y < h; y++) {
        const float *S = src + (size_t)y * w;
        for (int x = 0; x < w; x++) {
            float s;
            if (ksize == 3)
                s = S[x] * k[1] + (S[reflect101(x - 1, w)] + S[reflect101(x + 1, w)]) * k[0];
            else if (ksize == 5)
                s = S[x] * k[2] + (S[reflect101(x - 1, w)] + S[reflect101(x + 1, w)]) * k[1]
                  + (S[reflect101(x - 2, w)] + S[reflect101(x + 2, w)]) * k[0];
            else {
                s = k[0] * S[reflect101(x - r, w)];
                for (int i = 1; i < ksize; i++) s += k[i] * S[reflect101(x - r + i, w)];
            }
            tmp[(size_t)y * w + x] = s;
        }
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float s = k[r] * tmp[(size_t)y * w + x];
            for (int i = 1; i <= r; i++)
                s += k[r + i] * (tmp[(size_t)reflect101(y + i, h) * w + x] + tmp[(size_t)reflect101(y - i, h) * w + x]);
            dst[(size_t)y * w + x] = s;
        }
    free(tmp); free(k);
}

/* cv::resize(src, dst, Size(dw, dh), 0, 0, INTER_LINEAR) for cn-channel float images.
 * Exact 2x decimation takes OpenCV's INTER_AREA fast path (sum of the 2x2 block * 0.25). */
static void resize_linear(const float *src, int sh, int sw, int cn, float *dst, int dh, int dw) {
    double inv_x = (double)dw / sw, inv_y = (double)dh / sh;
    double scale_x = 1. / inv_x, scale_y = 1. / inv_y;
    int isx = (int)(scale_x >= 0 ? scale_x + 0.5 : scale_x - 0.5), isy = (int)(scale_y >= 0 ? scale_y + 0.5 : scale_y - 0.5);
    int area_fast = fabs(scale_x - isx) < DBL_EPSILON && fabs(scale_y - isy) < DBL_EPSILON;
    if (area_fast && isx == 2 && isy == 2) {
        for (int y = 0; y < dh; y++)
            for (int x = 0; x < dw; x++)
                for (int c = 0; c < cn; c++) {
                    const float *S = src + ((size_t)(2 * y) * sw + 2 * x) * cn + c;
                    float sum = 0;
                    sum += S[0] + S[cn] + S[(size_t)sw * cn] + S[(size_t)sw * cn + cn];
                    dst[((size_t)y * dw + x) * cn + c] = sum * 0.25f;
                }
        return;
    }
    int *xofs = (int *)malloc(dw * sizeof(int)), *yofs = (int *)malloc(dh * sizeof(int));
    float *ax = (float *)malloc(dw * 2 * sizeof(float)), *ay = (float *)malloc(dh * 2 * sizeof(float));
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5) * scale_x - 0.5);
        int sx = cv_floor_f(fx); fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx; ax[dx * 2] = 1.f - fx; ax[dx * 2 + 1] = fx;
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5) * scale_y - 0.5);
        int sy = cv_floor_f(fy); fy -= sy;
        /* OpenCV keeps the vertical weight and clamps the two ROW INDICES instead (below) */
        yofs[dy] = sy; ay[dy * 2] = 1.f - fy; ay[dy * 2 + 1] = fy;
    }
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = yofs[dy] < 0 ? 0 : (yofs[dy] > sh - 1 ? sh - 1 : yofs[dy]);
        int sy1 = yofs[dy] + 1 < 0 ? 0 : (yofs[dy] + 1 > sh - 1 ? sh - 1 : yofs[dy] + 1);
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx], sx1 = sx + 1 < sw ? sx + 1 : sx;
            for (int c = 0; c < cn; c++) {
                float r0 = src[((size_t)sy0 * sw + sx) * cn + c] * ax[dx * 2] + src[((size_t)sy0 * sw + sx1) * cn + c] * ax[dx * 2 + 1];
                float r1 = src[((size_t)sy1 * sw + sx) * cn + c] * ax[dx * 2] + src[((size_t)sy1 * sw + sx1) * cn + c] * ax[dx * 2 + 1];
                dst[((size_t)dy * dw + dx) * cn + c] = r0 * ay[dy * 2] + r1 * ay[dy * 2 + 1];
            }
        }
    }
    free(xofs); free(yofs); free(ax); free(ay);
}

/* 6x6 symmetric positive definite inverse by Gauss-Jordan in double (G.inv(DECOMP_CHOLESKY)) */
static void inv6(double G[6][6], double inv[6][6]) {
    double a[6][12];
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) { a[i][j] = G[i][j]; a[i][j + 6] = i == j; }
    for (int c = 0; c < 6; c++) {
        int p = c; for (int r = c + 1; r < 6; r++) if (fabs(a[r][c]) > fabs(a[p][c])) p = r;
        if (p != c) for (int j = 0; j < 12; j++) { double t = a[c][j]; a[c][j] = a[p][j]; a[p][j] = t; }
        double d = 1. / a[c][c];
        for (int j = 0; j < 12; j++) a[c][j] *= d;
        for (int r = 0; r < 6; r++) if (r != c) { double f = a[r][c]; if (f != 0) for (int j = 0; j < 12; j++) a[r][j] -= f * a[c][j]; }
    }
    for (int i = 0; i < 6; i++) for (int j = 0; j < 6; j++) inv[i][j] = a[i][j + 6];
}

void oracle_farneback_prepare_gaussian(int n, double sigma, float *g, float *xg, float *xxg, double *ig) {
    /* FarnebackPrepareGaussian; g/xg/xxg point at the centre tap */
    if (sigma < FLT_EPSILON) sigma = n * 0.3;
    double s = 0.;
    for (int x = -n; x <= n; x++) { g[x] = (float)exp(-x * x / (2 * sigma * sigma)); s += g[x]; }
    s = 1. / s;
    for (int x = -n; x <= n; x++) { g[x] = (float)(g[x] * s); xg[x] = (float)(x * g[x]); xxg[x] = (float)(x * x * g[x]); }
    double G[6][6]; memset(G, 0, sizeof(G));
    for (int y = -n; y <= n; y++)
        for (int x = -n; x <= n; x++) {
            G[0][0] += g[y] * g[x];
            G[1][1] += g[y] * g[x] * x * x;
            G[3][3] += g[y] * g[x] * x * x * x * x;
            G[5][5] += g[y] * g[x] * x * x * y * y;
        }
    G[2][2] = G[0][3] = G[0][4] = G[3][0] = G[4][0] = G[1][1];
    G[4][4] = G[3][3];
    G[3][4] = G[4][3] = G[5][5];
    double inv[6][6]; inv6(G, inv);
    ig[0] = inv[1][1]; ig[1] = inv[0][3]; ig[2] = inv[3][3]; ig[3] = inv[5][5];
}

static void poly_exp(const float *src, int height, int width, float *dst, int n, double sigma) {
    float *kbuf = (float *)malloc((n * 6 + 3) * sizeof(float));
    float *g = kbuf + n, *xg = g + n * 2 + 1, *xxg = xg + n * 2 + 1;
    float *rowbuf = (float *)malloc((size_t)(width + n * 2) * 3 * sizeof(float)), *row = rowbuf + n * 3;
    double ig[4];
    oracle_farneback_prepare_gaussian(n, sigma, g, xg, xxg, ig);
    double ig11 = ig[0], ig03 = ig[1], ig33 = ig[2], ig55 = ig[3];
    for (int y = 0; y < height; y++) {
        float g0 = g[0], g1, g2;
        const float *srow0 = src + (size_t)y * width, *srow1;
        float *drow = dst + (size_t)y * width * 5;
        for (int x = 0; x < width; x++) { row[x * 3] = srow0[x] * g0; row[x * 3 + 1] = row[x * 3 + 2] = 0.f; }
        for (int k = 1; k <= n; k++) {
            g0 = g[k]; g1 = xg[k]; g2 = xxg[k];
            srow0 = src + (size_t)(y - k > 0 ? y - k : 0) * width;
            srow1 = src + (size_t)(y + k < height - 1 ? y + k : height - 1) * width;
            for (int x = 0; x < width; x++) {
                float p = srow0[x] + srow1[x];
                float t0 = row[x * 3] + g0 * p;
                float t1 = row[x * 3 + 1] + g1 * (srow1[x] - srow0[x]);
                float t2 = row[x * 3 + 2] + g2 * p;
                row[x * 3] = t0; row[x * 3 + 1] = t1; row[x * 3 + 2] = t2;
            }
        }
        for (int x = 0; x < n * 3; x++) { row[-1 - x] = row[2 - x]; row[width * 3 + x] = row[width * 3 + x - 3]; }
        for (int x = 0; x < width; x++) {
            g0 = g[0];
            double b1 = row[x * 3] * g0, b2 = 0, b3 = row[x * 3 + 1] * g0, b4 = 0, b5 = row[x * 3 + 2] * g0, b6 = 0;
            for (int k = 1; k <= n; k++) {
                double tg = row[(x + k) * 3] + row[(x - k) * 3];
                g0 = g[k];
                b1 += tg * g0; b4 += tg * xxg[k];
                b2 += (row[(x + k) * 3] - row[(x - k) * 3]) * xg[k];
                b3 += (row[(x + k) * 3 + 1] + row[(x - k) * 3 + 1]) * g0;
                b6 += (row[(x + k) * 3 + 1] - row[(x - k) * 3 + 1]) * xg[k];
                b5 += (row[(x + k) * 3 + 2] + row[(x - k) * 3 + 2]) * g0;
            }
            drow[x * 5 + 1] = (float)(b2 * ig11);
            drow[x * 5] = (float)(b3 * ig11);
            drow[x * 5 + 3] = (float)(b1 * ig03 + b4 * ig33);
            drow[x * 5 + 2] = (float)(b1 * ig03 + b5 * ig33);
            drow[x * 5 + 4] = (float)(b6 * ig55);
        }
    }
    free(kbuf); free(rowbuf);
}

static void update_matrices(const float *R0_, const float *R1, const float *flow_, float *matM,
                            int height, int width, int y0, int y1) {
    enum { BORDER = 5 };
    static const float border[BORDER] = {0.14f, 0.14f, 0.4472f, 0.4472f, 0.4472f};
    size_t step1 = (size_t)width * 5;
    for (int y = y0; y < y1; y++) {
        const float *flow = flow_ + (size_t)y * width * 2, *R0 = R0_ + (size_t)y * width * 5;
        float *M = matM + (size_t)y * width * 5;
        for (int x = 0; x < width; x++) {
            float dx = flow[x * 2], dy = flow[x * 2 + 1];
            float fx = x + dx, fy = y + dy;
            int x1 = cv_floor_f(fx), y1_ = cv_floor_f(fy);
            float r2, r3, r4, r5, r6;
            fx -= x1; fy -= y1_;
            if ((unsigned)x1 < (unsigned)(width - 1) && (unsigned)y1_ < (unsigned)(height - 1)) {
                const float *ptr = R1 + (size_t)y1_ * step1 + (size_t)x1 * 5;
                float a00 = (1.f - fx) * (1.f - fy), a01 = fx * (1.f - fy), a10 = (1.f - fx) * fy, a11 = fx * fy;
                r2 = a00 * ptr[0] + a01 * ptr[5] + a10 * ptr[step1] + a11 * ptr[step1 + 5];
                r3 = a00 * ptr[1] + a01 * ptr[6] + a10 * ptr[step1 + 1] + a11 * ptr[step1 + 6];
                r4 = a00 * ptr[2] + a01 * ptr[7] + a10 * ptr[step1 + 2] + a11 * ptr[step1 + 7];
                r5 = a00 * ptr[3] + a01 * ptr[8] + a10 * ptr[step1 + 3] + a11 * ptr[step1 + 8];
                r6 = a00 * ptr[4] + a01 * ptr[9] + a10 * ptr[step1 + 4] + a11 * ptr[step1 + 9];
                r4 = (R0[x * 5 + 2] + r4) * 0.5f;
                r5 = (R0[x * 5 + 3] + r5) * 0.5f;
                r6 = (R0[x * 5 + 4] + r6) * 0.25f;
            } else {
                r2 = r3 = 0.f;
                r4 = R0[x * 5 + 2]; r5 = R0[x * 5 + 3]; r6 = R0[x * 5 + 4] * 0.5f;
            }
            r2 = (R0[x * 5] - r2) * 0.5f;
            r3 = (R0[x * 5 + 1] - r3) * 0.5f;
            r2 += r4 * dy + r6 * dx;
            r3 += r6 * dy + r5 * dx;
            if ((unsigned)(x - BORDER) >= (unsigned)(width - BORDER * 2) ||
                (unsigned)(y - BORDER) >= (unsigned)(height - BORDER * 2)) {
                float scale = (x < BORDER ? border[x] : 1.f) * (x >= width - BORDER ? border[width - x - 1] : 1.f) *
                              (y < BORDER ? border[y] : 1.f) * (y >= height - BORDER ? border[height - y - 1] : 1.f);
                r2 *= scale; r3 *= scale; r4 *= scale; r5 *= scale; r6 *= scale;
            }
            M[x * 5] = r4 * r4 + r6 * r6;
            M[x * 5 + 1] = (r4 + r5) * r6;
            M[x * 5 + 2] = r5 * r5 + r6 * r6;
            M[x * 5 + 3] = r4 * r2 + r6 * r3;
            M[x * 5 + 4] = r6 * r2 + r5 * r3;
        }
    }
}

static void update_flow_blur(const float *R0, const float *R1, float *flow_, float *matM,
                             int height, int width, int block_size, int update) {
    int m = block_size / 2, y0 = 0, y1;
    int min_update_stripe = (1 << 10) / width > block_size ? (1 << 10) / width : block_size;
    double scale = 1. / (block_size * block_size);
    double *vbuf = (double *)malloc((size_t)(width + m * 2 + 2) * 5 * sizeof(double)), *vsum = vbuf + (m + 1) * 5;
    const float *srow0 = matM;
    for (int x = 0; x < width * 5; x++) vsum[x] = srow0[x] * (m + 2);
    for (int y = 1; y < m; y++) {
        srow0 = matM + (size_t)(y < height - 1 ? y : height - 1) * width * 5;
        for (int x = 0; x < width * 5; x++) vsum[x] += srow0[x];
    }
    for (int y = 0; y < height; y++) {
        double g11, g12, g22, h1, h2;
        float *flow = flow_ + (size_t)y * width * 2;
        srow0 = matM + (size_t)(y - m - 1 > 0 ? y - m - 1 : 0) * width * 5;
        const float *srow1 = matM + (size_t)(y + m < height - 1 ? y + m : height - 1) * width * 5;
        for (int x = 0; x < width * 5; x++) vsum[x] += srow1[x] - srow0[x];
        for (int x = 0; x < (m + 1) * 5; x++) { vsum[-1 - x] = vsum[4 - x]; vsum[width * 5 + x] = vsum[width * 5 + x - 5]; }
        g11 = vsum[0] * (m + 2); g12 = vsum[1] * (m + 2); g22 = vsum[2] * (m + 2);
        h1 = vsum[3] * (m + 2); h2 = vsum[4] * (m + 2);
        for (int x = 1; x < m; x++) {
            g11 += vsum[x * 5]; g12 += vsum[x * 5 + 1]; g22 += vsum[x * 5 + 2]; h1 += vsum[x * 5 + 3]; h2 += vsum[x * 5 + 4];
        }
        for (int x = 0; x < width; x++) {
            g11 += vsum[(x + m) * 5] - vsum[(x - m) * 5 - 5];
            g12 += vsum[(x + m) * 5 + 1] - vsum[(x - m) * 5 - 4];
            g22 += vsum[(x + m) * 5 + 2] - vsum[(x - m) * 5 - 3];
            h1 += vsum[(x + m) * 5 + 3] - vsum[(x - m) * 5 - 2];
            h2 += vsum[(x + m) * 5 + 4] - vsum[(x - m) * 5 - 1];
            double g11_ = g11 * scale, g12_ = g12 * scale, g22_ = g22 * scale, h1_ = h1 * scale, h2_ = h2 * scale;
            double idet = 1. / (g11_ * g22_ - g12_ * g12_ + 1e-3);
            flow[x * 2] = (float)((g11_ * h2_ - g12_ * h1_) * idet);
            flow[x * 2 + 1] = (float)((g22_ * h1_ - g12_ * h2_) * idet);
        }
        y1 = y == height - 1 ? height : y - block_size;
        if (update && (y1 == height || y1 >= y0 + min_update_stripe)) {
            update_matrices(R0, R1, flow_, matM, height, width, y0, y1);
            y0 = y1;
        }
    }
    free(vbuf);
}

/* prev/next: uint8 (h, w); flow_out: float (h, w, 2) = (dx, dy).  Returns number of pyramid
 * resolutions processed, or -1 on allocation failure. */
int oracle_farneback(const uint8_t *prev, const uint8_t *next, int h, int w, float *flow_out,
                     int num_levels, double pyr_scale, int win_size, int num_iters, int poly_n, double poly_sigma)
{
    const int min_size = 32;
    const uint8_t *img[2] = {prev, next};
    int levels = num_levels, k;
    double scale;
    for (k = 0, scale = 1; k < levels; k++) {
        scale *= pyr_scale;
        if (w * scale < min_size || h * scale < min_size) break;
    }
    levels = k;
    size_t npix = (size_t)h * w;
    float *fimg = (float *)malloc(npix * sizeof(float)), *blur = (float *)malloc(npix * sizeof(float));
    float *I = (float *)malloc(npix * sizeof(float));
    float *R[2] = {(float *)malloc(npix * 5 * sizeof(float)), (float *)malloc(npix * 5 * sizeof(float))};
    float *M = (float *)malloc(npix * 5 * sizeof(float));
    float *prevFlow = 0, *flow = 0;
    int pw = 0, ph = 0, nres = 0;
    if (!fimg || !blur || !I || !R[0] || !R[1] || !M) return -1;
    for (k = levels; k >= 0; k--) {
        int i;
        for (i = 0, scale = 1; i < k; i++) scale *= pyr_scale;
        double sigma = (1. / scale - 1) * 0.5;
        int smooth_sz = cv_round_d(sigma * 5) | 1;
        if (smooth_sz < 3) smooth_sz = 3;
        int width = cv_round_d(w * scale), height = cv_round_d(h * scale);
        flow = (k > 0) ? (float *)malloc((size_t)width * height * 2 * sizeof(float)) : flow_out;
        if (!prevFlow) memset(flow, 0, (size_t)width * height * 2 * sizeof(float));
        else {
            resize_linear(prevFlow, ph, pw, 2, flow, height, width);
            float s = (float)(1. / pyr_scale);
            for (size_t j = 0; j < (size_t)width * height * 2; j++) flow[j] *= s;
        }
        for (i = 0; i < 2; i++) {
            for (size_t j = 0; j < npix; j++) fimg[j] = (float)img[i][j];
            gaussian_blur(fimg, h, w, smooth_sz, sigma, blur);
            if (width == w && height == h) memcpy(I, blur, npix * sizeof(float));
            else resize_linear(blur, h, w, 1, I, height, width);
            poly_exp(I, height, width, R[i], poly_n, poly_sigma);
        }
        update_matrices(R[0], R[1], flow, M, height, width, 0, height);
        for (i = 0; i < num_iters; i++)
            update_flow_blur(R[0], R[1], flow, M, height, width, win_size, i < num_iters - 1);
        if (prevFlow) free(prevFlow);
        prevFlow = flow; pw = width; ph = height; nres++;
    }
    free(fimg); free(blur); free(I); free(R[0]); free(R[1]); free(M);
    return nres;
}

/* Exposed pieces so tests can compare individual stages of the HIP path. */
void oracle_gaussian_blur(const float *src, int h, int w, int ksize, double sigma, float *dst) { gaussian_blur(src, h, w, ksize, sigma, dst); }
void oracle_resize_linear(const float *src, int sh, int sw, int cn, float *dst, int dh, int dw) { resize_linear(src, sh, sw, cn, dst, dh, dw); }
void oracle_poly_exp(const float *src, int h, int w, float *dst, int n, double sigma) { poly_exp(src, h, w, dst, n, sigma); }
void oracle_update_matrices(const float *R0, const float *R1, const float *flow, float *M, int h, int w) { update_matrices(R0, R1, flow, M, h, w, 0, h); }
void oracle_update_flow_blur(const float *R0, const float *R1, float *flow, float *M, int h, int w, int block, int update) { update_flow_blur(R0, R1, flow, M, h, w, block, update); }
