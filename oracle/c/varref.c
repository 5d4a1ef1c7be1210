/* ORACLE (test infrastructure, never shipped or measured as the product):
 * CPU restatement of cv2.VariationalRefinement.create().calc(I0, I1, flow) as the reference calls it
 *   /root/reference/tobac_flow/flow.py:359        vr_model = cv2.VariationalRefinement.create()
 *   /root/reference/tobac_flow/flow.py:513-519    one calc() per direction when vr_steps > 0
 * with OpenCV's defaults (fixedPointIterations 5, sorIterations 5, alpha 20, delta 5, gamma 10, omega 1.6;
 * internal zeta 0.1, epsilon 0.001).
 *
 * OpenCV is a third-party dependency ABSENT from /root/reference and from this image (environment.yml:15 `opencv`,
 * unpinned).  This file restates the published algorithm of modules/video/src/variational_refinement.cpp
 * (OpenCV 4.x; Brox et al. 2004 warping-based refinement as used by DISOpticalFlow) from upstream knowledge:
 *   prepareBuffers   I1 -> float, warped by the flow with remap(INTER_LINEAR, BORDER_REPLICATE) (coordinates quantised
 *                    to 1/32 px like every non-nearest remap); averaged image (I0 + warped) / 2; Iz = warped - I0;
 *                    Ix, Iy of the averaged image, Ixz, Iyz of Iz, Ixx, Ixy of Ix, Iyy of Iy, all with
 *                    Sobel(ksize = 1, scale = 1, BORDER_REPLICATE) = the unnormalised central difference
 *   fixed point loop data term (colour + gradient constancy, robust weights), smoothness term (horizontal pass with
 *                    the weights, then vertical pass), sorIterations red-black SOR sweeps (red = even y + x first),
 *                    tempW = W + dW
 * The accumulation order of the smoothness contributions into A11 / A22 / b1 / b2 follows OpenCV's passes (red
 * elements before black ones, horizontal before vertical; each element adds to itself and to its right / lower
 * neighbour), which makes the order at a pixel depend on its colour.
 * PARITY STATUS: **parity unpinned** -- no cv2 in this image and the reference holds no vectors for this stage; the
 * tests check the method's own identities (tests/test_oracle_known_answers.py) and the HIP kernels against this file.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define INTER_BITS 5
#define INTER_TAB_SIZE 32

static inline int vr_round(float v) { return (int)lrintf(v); }
static inline int vr_sat_short(int v) { return v < -32768 ? -32768 : (v > 32767 ? 32767 : v); }
static inline int vr_clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* remap(src float, INTER_LINEAR, BORDER_REPLICATE) at (x + u, y + v) */
static void warp_replicate(const float *src, int H, int W, const float *u, const float *v, float *dst)
{
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const int64_t i = (int64_t)y * W + x;
            const float mx = (float)x + u[i], my = (float)y + v[i];
            const int fx = vr_round(mx * (float)INTER_TAB_SIZE), fy = vr_round(my * (float)INTER_TAB_SIZE);
            const int sx = vr_sat_short(fx >> INTER_BITS), sy = vr_sat_short(fy >> INTER_BITS);
            const int ax = fx & (INTER_TAB_SIZE - 1), ay = fy & (INTER_TAB_SIZE - 1);
            const float tx1 = (float)ax * (1.f / INTER_TAB_SIZE), tx0 = 1.f - tx1;
            const float ty1 = (float)ay * (1.f / INTER_TAB_SIZE), ty0 = 1.f - ty1;
            const float w0 = ty0 * tx0, w1 = ty0 * tx1, w2 = ty1 * tx0, w3 = ty1 * tx1;
            const int x0 = vr_clamp(sx, 0, W - 1), x1 = vr_clamp(sx + 1, 0, W - 1);
            const int y0 = vr_clamp(sy, 0, H - 1), y1 = vr_clamp(sy + 1, 0, H - 1);
            dst[i] = src[(int64_t)y0 * W + x0] * w0 + src[(int64_t)y0 * W + x1] * w1
                   + src[(int64_t)y1 * W + x0] * w2 + src[(int64_t)y1 * W + x1] * w3;
        }
}

/* Sobel(ksize = 1, scale = 1, BORDER_REPLICATE): dir 0 = d/dx, 1 = d/dy */
static void central_diff(const float *src, int H, int W, int dir, float *dst)
{
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            float a, b;
            if (dir == 0) { a = src[(int64_t)y * W + vr_clamp(x + 1, 0, W - 1)]; b = src[(int64_t)y * W + vr_clamp(x - 1, 0, W - 1)]; }
            else { a = src[(int64_t)vr_clamp(y + 1, 0, H - 1) * W + x]; b = src[(int64_t)vr_clamp(y - 1, 0, H - 1) * W + x]; }
            dst[(int64_t)y * W + x] = a - b;
        }
}

typedef struct {
    int fixed_point_iterations, sor_iterations;
    float alpha, delta, gamma, omega, zeta, epsilon;
} vr_params;

/* flow: (H, W, 2) float32, (dx, dy), refined in place.  Returns 0, or -1 on allocation failure. */
int oracle_variational_refinement(const uint8_t *I0, const uint8_t *I1, int H, int W, float *flow,
                                  int fixed_point_iterations, int sor_iterations,
                                  float alpha, float delta, float gamma, float omega)
{
    const int64_t N = (int64_t)H * W;
    const float zeta = 0.1f, epsilon = 0.001f;
    const int n_planes = 24;
    float *buf = (float *)malloc((size_t)N * n_planes * sizeof(float));
    if (!buf) return -1;
    float *Wu = buf, *Wv = buf + N, *I1f = buf + 2 * N, *warped = buf + 3 * N, *avg = buf + 4 * N;
    float *Ix = buf + 5 * N, *Iy = buf + 6 * N, *Iz = buf + 7 * N, *Ixx = buf + 8 * N, *Ixy = buf + 9 * N, *Iyy = buf + 10 * N;
    float *Ixz = buf + 11 * N, *Iyz = buf + 12 * N;
    float *A11 = buf + 13 * N, *A12 = buf + 14 * N, *A22 = buf + 15 * N, *b1 = buf + 16 * N, *b2 = buf + 17 * N;
    float *wt = buf + 18 * N, *cu = buf + 19 * N, *cv = buf + 20 * N, *du = buf + 21 * N, *dv = buf + 22 * N;
    for (int64_t i = 0; i < N; i++) { Wu[i] = flow[2 * i]; Wv[i] = flow[2 * i + 1]; I1f[i] = (float)I1[i]; }
    warp_replicate(I1f, H, W, Wu, Wv, warped);
    for (int64_t i = 0; i < N; i++) { avg[i] = ((float)I0[i] + warped[i]) * 0.5f; Iz[i] = warped[i] - (float)I0[i]; }
    central_diff(avg, H, W, 0, Ix);  central_diff(avg, H, W, 1, Iy);
    central_diff(Iz, H, W, 0, Ixz);  central_diff(Iz, H, W, 1, Iyz);
    central_diff(Ix, H, W, 0, Ixx);  central_diff(Ix, H, W, 1, Ixy);
    central_diff(Iy, H, W, 1, Iyy);
    memcpy(cu, Wu, (size_t)N * sizeof(float)); memcpy(cv, Wv, (size_t)N * sizeof(float));
    memset(du, 0, (size_t)N * sizeof(float)); memset(dv, 0, (size_t)N * sizeof(float));

    const float zeta_squared = zeta * zeta, epsilon_squared = epsilon * epsilon;
    const float gamma2 = gamma / 2, delta2 = delta / 2, alpha2 = alpha / 4;
    for (int it = 0; it < fixed_point_iterations; it++) {
        /* ComputeDataTerm */
        for (int64_t j = 0; j < N; j++) {
            float derivNorm = Ix[j] * Ix[j] + Iy[j] * Iy[j] + zeta_squared;
            const float Ik1z = Iz[j] + Ix[j] * du[j] + Iy[j] * dv[j];
            float weight = delta2 / sqrtf(Ik1z * Ik1z / derivNorm + epsilon_squared);
            A11[j] = weight * (Ix[j] * Ix[j] / derivNorm) + zeta_squared;
            A12[j] = weight * (Ix[j] * Iy[j] / derivNorm);
            A22[j] = weight * (Iy[j] * Iy[j] / derivNorm) + zeta_squared;
            b1[j] = -weight * (Iz[j] * Ix[j] / derivNorm);
            b2[j] = -weight * (Iz[j] * Iy[j] / derivNorm);
            derivNorm = Ixx[j] * Ixx[j] + Ixy[j] * Ixy[j] + zeta_squared;
            const float derivNorm2 = Iyy[j] * Iyy[j] + Ixy[j] * Ixy[j] + zeta_squared;
            const float Ik1zx = Ixz[j] + Ixx[j] * du[j] + Ixy[j] * dv[j];
            const float Ik1zy = Iyz[j] + Ixy[j] * du[j] + Iyy[j] * dv[j];
            weight = gamma2 / sqrtf(Ik1zx * Ik1zx / derivNorm + Ik1zy * Ik1zy / derivNorm2 + epsilon_squared);
            A11[j] += weight * (Ixx[j] * Ixx[j] / derivNorm + Ixy[j] * Ixy[j] / derivNorm2);
            A12[j] += weight * (Ixx[j] * Ixy[j] / derivNorm + Ixy[j] * Iyy[j] / derivNorm2);
            A22[j] += weight * (Ixy[j] * Ixy[j] / derivNorm + Iyy[j] * Iyy[j] / derivNorm2);
            b1[j] += -weight * (Ixx[j] * Ixz[j] / derivNorm + Ixy[j] * Iyz[j] / derivNorm2);
            b2[j] += -weight * (Ixy[j] * Ixz[j] / derivNorm + Iyy[j] * Iyz[j] / derivNorm2);
        }
        /* ComputeSmoothnessTermHorPass: red elements, then black ones */
        for (int colour = 0; colour < 2; colour++)
            for (int y = 0; y < H; y++)
                for (int x = (y + colour) & 1; x < W; x += 2) {
                    const int64_t j = (int64_t)y * W + x;
                    const int64_t jr = x + 1 < W ? j + 1 : j, jd = y + 1 < H ? j + W : j;   /* replicated borders */
                    float ux = cu[jr] - cu[j], vx = cv[jr] - cv[j], uy = cu[jd] - cu[j], vy = cv[jd] - cv[j];
                    wt[j] = alpha2 / sqrtf(ux * ux + vx * vx + uy * uy + vy * vy + epsilon_squared);
                    if (x + 1 >= W) continue;               /* the rightmost element only gets its weight */
                    ux = wt[j] * (Wu[jr] - Wu[j]);
                    vx = wt[j] * (Wv[jr] - Wv[j]);
                    b1[j] += ux; A11[j] += wt[j]; b2[j] += vx; A22[j] += wt[j];
                    b1[jr] -= ux; A11[jr] += wt[j]; b2[jr] -= vx; A22[jr] += wt[j];
                }
        /* ComputeSmoothnessTermVertPass: the last row has no lower neighbour */
        for (int colour = 0; colour < 2; colour++)
            for (int y = 0; y + 1 < H; y++)
                for (int x = (y + colour) & 1; x < W; x += 2) {
                    const int64_t j = (int64_t)y * W + x, jd = j + W;
                    const float uy = wt[j] * (Wu[jd] - Wu[j]);
                    const float vy = wt[j] * (Wv[jd] - Wv[j]);
                    b1[j] += uy; A11[j] += wt[j]; b2[j] += vy; A22[j] += wt[j];
                    b1[jd] -= uy; A11[jd] += wt[j]; b2[jd] -= vy; A22[jd] += wt[j];
                }
        /* RedBlackSOR: out-of-image neighbours carry dW = 0 */
        for (int s = 0; s < sor_iterations; s++)
            for (int colour = 0; colour < 2; colour++)
                for (int y = 0; y < H; y++)
                    for (int x = (y + colour) & 1; x < W; x += 2) {
                        const int64_t j = (int64_t)y * W + x;
                        const float wl = x > 0 ? wt[j - 1] : 0.f, wu_ = y > 0 ? wt[j - W] : 0.f;
                        const float dul = x > 0 ? du[j - 1] : 0.f, dur = x + 1 < W ? du[j + 1] : 0.f;
                        const float duu = y > 0 ? du[j - W] : 0.f, dud = y + 1 < H ? du[j + W] : 0.f;
                        const float dvl = x > 0 ? dv[j - 1] : 0.f, dvr = x + 1 < W ? dv[j + 1] : 0.f;
                        const float dvu = y > 0 ? dv[j - W] : 0.f, dvd = y + 1 < H ? dv[j + W] : 0.f;
                        const float sigmaU = wl * dul + wt[j] * dur + wu_ * duu + wt[j] * dud;
                        const float sigmaV = wl * dvl + wt[j] * dvr + wu_ * dvu + wt[j] * dvd;
                        du[j] += omega * ((sigmaU + b1[j] - dv[j] * A12[j]) / A11[j] - du[j]);
                        dv[j] += omega * ((sigmaV + b2[j] - du[j] * A12[j]) / A22[j] - dv[j]);
                    }
        for (int64_t j = 0; j < N; j++) { cu[j] = Wu[j] + du[j]; cv[j] = Wv[j] + dv[j]; }
    }
    for (int64_t i = 0; i < N; i++) { flow[2 * i] = cu[i]; flow[2 * i + 1] = cv[i]; }
    free(buf);
    return 0;
}
