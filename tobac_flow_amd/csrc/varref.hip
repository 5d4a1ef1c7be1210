// Variational refinement of a dense flow field on gfx950.
//
// Replaces cv2.VariationalRefinement.create().calc(I0, I1, flow) as the reference issues it once per direction when
// vr_steps > 0 (/root/reference/tobac_flow/flow.py:359, 513-519), with OpenCV's defaults: fixedPointIterations 5,
// sorIterations 5, alpha 20, delta 5, gamma 10, omega 1.6 (internal zeta 0.1, epsilon 0.001) --
// modules/video/src/variational_refinement.cpp, restated from the published algorithm (SURVEY.md Appendix A.2;
// PARITY UNPINNED: no OpenCV in the build image, the reference holds no vectors for this stage).
//
//   k_vr_prepare   I1 warped by the flow (bilinear, coordinates quantised to 1/32 px, replicated border), averaged
//                  image, Iz, and all first / second central differences (Sobel ksize 1, replicated border) on ONE LDS
//                  tile with a 2-pixel halo: the eight derivative planes are written once (32 B / px), nothing else
//                  touches HBM
//   k_vr_system    smoothness weights of the current flow W + dW (forward differences, replicated border; each
//                  weight shared with the right / lower neighbour through a lane shuffle / LDS) and
//                  the 2x2 system of every pixel: data term (robust colour- and gradient-constancy weights) plus the
//                  smoothness contributions of its four edges, accumulated in OpenCV's pass order (red before black,
//                  horizontal before vertical -- the order at a pixel depends on its colour)
//   k_vr_sor       one half sweep (one colour) of red-black SOR on dW
// Every float expression is evaluated as written (-ffp-contract=off, correctly rounded divide / sqrt), so the result
// is bit-identical to the oracle's C restatement (oracle/c/varref.c).
//
// HBM layout (planar, per flow direction): D1 = float4 {Ix, Iy, Ixz, Iyz}, D2 = float4 {Ixx, Ixy, Iyy, Iz},
// S = float4 {A11, A22, b1, b2}, A12 float, wt float, W float2 (the input flow), dW float2.
#include "tf_common.h"
#include <stdlib.h>
#include <string.h>

struct VrP { float alpha2, delta2, gamma2, omega, zeta2, eps2; };
// strides between the images of a batch (tf_varref_batch; all zero-cost for one image): frames in pixels, the caller's flow
// array in float2, the workspace planes in elements
struct VrB { int64_t img, flow, plane; };

__device__ __forceinline__ int vr_clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

#define VR_TW 64
#define VR_TH 16
#define VR_LW (VR_TW + 4)
#define VR_LH (VR_TH + 4)

__global__ void __launch_bounds__(256)
k_vr_prepare(const uint8_t *__restrict__ I0, const uint8_t *__restrict__ I1, const float2 *__restrict__ flow,
             int H, int W, float4 *__restrict__ D1, float4 *__restrict__ D2, VrB bs)
{
    __shared__ float s_avg[VR_LH][VR_LW], s_iz[VR_LH][VR_LW], s_ix[VR_LH][VR_LW], s_iy[VR_LH][VR_LW];
    {   // image blockIdx.z of a batch (tf_varref_batch): the same tile code on that image's arrays
        const int64_t b = blockIdx.z;
        I0 += b * bs.img; I1 += b * bs.img; flow += b * bs.flow; D1 += b * bs.plane; D2 += b * bs.plane;
    }
    const int x0 = blockIdx.x * VR_TW - 2, y0 = blockIdx.y * VR_TH - 2;
    // warped / averaged image and Iz on the tile + 2 halo (in-image positions only; neighbours are clamped later).
    // Three stages so that a thread's loads are in flight together: flow + I0 of all its positions, then the four I1
    // taps of all of them (their addresses need the flow), then the arithmetic and the LDS stores.
    {
        constexpr int NL = (VR_LW * VR_LH + 255) / 256;
        float2 f[NL]; float i0[NL]; bool ok[NL];
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const int i = threadIdx.x + 256 * j, ly = i / VR_LW, lx = i - ly * VR_LW;
            const int x = x0 + lx, y = y0 + ly;
            ok[j] = i < VR_LW * VR_LH && x >= 0 && y >= 0 && x < W && y < H;
            const int64_t p = ok[j] ? (int64_t)y * W + x : 0;
            f[j] = flow[p]; i0[j] = (float)I0[p];
        }
        float v[NL][4], wgt[NL][4];
#pragma unroll
        for (int j = 0; j < NL; j++) {
            const int i = threadIdx.x + 256 * j, ly = i / VR_LW, lx = i - ly * VR_LW;
            const int x = x0 + lx, y = y0 + ly;
            const float mx = (float)x + f[j].x, my = (float)y + f[j].y;
            const int fx = tf_cvround(mx * 32.f), fy = tf_cvround(my * 32.f);
            const int sx = tf_sat_short(fx >> 5), sy = tf_sat_short(fy >> 5);
            const int ax = fx & 31, ay = fy & 31;
            const float tx1 = (float)ax * (1.f / 32.f), tx0 = 1.f - tx1, ty1 = (float)ay * (1.f / 32.f), ty0 = 1.f - ty1;
            wgt[j][0] = ty0 * tx0; wgt[j][1] = ty0 * tx1; wgt[j][2] = ty1 * tx0; wgt[j][3] = ty1 * tx1;
            const int xa = vr_clampi(sx, W - 1), xb = vr_clampi(sx + 1, W - 1), ya = vr_clampi(sy, H - 1), yb = vr_clampi(sy + 1, H - 1);
            v[j][0] = (float)I1[(int64_t)ya * W + xa]; v[j][1] = (float)I1[(int64_t)ya * W + xb];
            v[j][2] = (float)I1[(int64_t)yb * W + xa]; v[j][3] = (float)I1[(int64_t)yb * W + xb];
        }
#pragma unroll
        for (int j = 0; j < NL; j++) {
            if (!ok[j]) continue;
            const int i = threadIdx.x + 256 * j, ly = i / VR_LW, lx = i - ly * VR_LW;
            const float warped = v[j][0] * wgt[j][0] + v[j][1] * wgt[j][1] + v[j][2] * wgt[j][2] + v[j][3] * wgt[j][3];
            s_avg[ly][lx] = (i0[j] + warped) * 0.5f;
            s_iz[ly][lx] = warped - i0[j];
        }
    }
    __syncthreads();
    // first differences of the averaged image on the tile + 1 halo (replicated border = clamped neighbour coordinates)
    for (int i = threadIdx.x; i < VR_LW * VR_LH; i += 256) {
        const int ly = i / VR_LW, lx = i - ly * VR_LW;
        if (lx < 1 || ly < 1 || lx >= VR_LW - 1 || ly >= VR_LH - 1) continue;
        const int x = x0 + lx, y = y0 + ly;
        if (x < 0 || y < 0 || x >= W || y >= H) continue;
        const int xr = vr_clampi(x + 1, W - 1) - x0, xl = vr_clampi(x - 1, W - 1) - x0;
        const int yd = vr_clampi(y + 1, H - 1) - y0, yu = vr_clampi(y - 1, H - 1) - y0;
        s_ix[ly][lx] = s_avg[ly][xr] - s_avg[ly][xl];
        s_iy[ly][lx] = s_avg[yd][lx] - s_avg[yu][lx];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < VR_TW * VR_TH; i += 256) {
        const int ty = i / VR_TW, tx = i - ty * VR_TW;
        const int lx = tx + 2, ly = ty + 2;
        const int x = x0 + lx, y = y0 + ly;
        if (x >= W || y >= H) continue;
        const int xr = vr_clampi(x + 1, W - 1) - x0, xl = vr_clampi(x - 1, W - 1) - x0;
        const int yd = vr_clampi(y + 1, H - 1) - y0, yu = vr_clampi(y - 1, H - 1) - y0;
        const float ixz = s_iz[ly][xr] - s_iz[ly][xl], iyz = s_iz[yd][lx] - s_iz[yu][lx];
        const float ixx = s_ix[ly][xr] - s_ix[ly][xl], ixy = s_ix[yd][lx] - s_ix[yu][lx];
        const float iyy = s_iy[yd][lx] - s_iy[yu][lx];
        const int64_t p = (int64_t)y * W + x;
        D1[p] = make_float4(s_ix[ly][lx], s_iy[ly][lx], ixz, iyz);
        D2[p] = make_float4(ixx, ixy, iyy, s_iz[ly][lx]);
    }
}

// ---- correctly rounded division with the reciprocal work shared between numerators -----------------------------------
// `n / d` in IEEE arithmetic costs the GPU eleven instructions: two range scalings, v_rcp, one Newton step on the
// reciprocal, q = n r, two residual corrections (the last one the final, correctly rounding FMA) and a fix-up for
// special values.  The reciprocal and its Newton step depend on d alone; the scalings and the fix-up do nothing when n
// is 0 or 2^-100 < |n| < 2^96 |d| and the quotient is a normal number.  vr_div_shared performs exactly the remaining
// five operations and the fix-up (which also gives -0 / d its sign), so its result is the hardware division's bit for bit in that range.  It is used only where the
// range is guaranteed: the derivative planes of k_vr_prepare are exact multiples of 2^-11 of magnitude <= 1020 (sums
// and differences of uint8 values and of bilinear samples with 1/32-quantised weights, all exact in float), so a
// product of two of them is 0 or lies in [2^-22, 2^20], and the denominators lie in [zeta^2, 2^22].
// tf_selftest_shared_divide compares the two forms on random operands of that range (tests/test_gpu_parity.py).
struct VrRcp { float d, r; };
__device__ __forceinline__ VrRcp vr_rcp_refined(float d)
{
    float r = __builtin_amdgcn_rcpf(d);
    const float e = __fmaf_rn(-d, r, 1.f);
    r = __fmaf_rn(e, r, r);
    VrRcp k; k.d = d; k.r = r;
    return k;
}
__device__ __forceinline__ float vr_div_shared(float n, VrRcp k)
{
#ifdef VR_PLAIN_DIVIDE                 /* A/B build: the hardware division */
    return n / k.d;
#endif
    float q = n * k.r;
    float e = __fmaf_rn(-k.d, q, n);
    q = __fmaf_rn(e, k.r, q);
    e = __fmaf_rn(-k.d, q, n);
    return __builtin_amdgcn_div_fixupf(__fmaf_rn(e, k.r, q), k.d, n);     // the fix-up keeps the sign of a zero numerator
}

__global__ void __launch_bounds__(256)
k_vr_selftest_divide(unsigned long long seed, int64_t count, unsigned long long *__restrict__ mismatches)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    // splitmix64 -> operands shaped like the kernel's: products of two multiples of 2^-11 over a sum of squares + zeta^2
    auto next = [](unsigned long long &z) { z += 0x9e3779b97f4a7c15ull; unsigned long long x = z; x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
                                            x = (x ^ (x >> 27)) * 0x94d049bb133111ebull; return x ^ (x >> 31); };
    unsigned long long st = seed + 0x632be59bd9b4e019ull * (unsigned long long)i;
    float v[4];
    for (int k = 0; k < 4; k++) {
        const unsigned long long r = next(st);
        const int mag = (int)(r % 2088961u) - 1044480;                   // -1020 * 1024 .. 1020 * 1024, in units of 2^-11... / 2
        const int shift = (int)((r >> 40) % 12);                          // small values as often as large ones
        v[k] = (float)(mag >> shift) * (1.f / 2048.f);
    }
    const float d = v[0] * v[0] + v[1] * v[1] + 0.1f * 0.1f, n = v[2] * v[3];
    const float fast = vr_div_shared(n, vr_rcp_refined(d)), ieee = n / d;
    if (__float_as_uint(fast) != __float_as_uint(ieee)) atomicAdd(mismatches, 1ull);
}

// smoothness weight from the current flow at a pixel (c0), its right (cr) and its lower (cd) neighbour
// FAST (TF_VR_FAST_DIVIDE, opt-in): hardware reciprocal square root / reciprocal (1 ulp) and reciprocal-multiply instead
// of the correctly rounded `/` and sqrtf -- not bit-identical to OpenCV's arithmetic any more, within 1e-4 px of it
// (tests/test_gpu_parity.py).  The default (FAST = false) evaluates every expression as written.
template <bool FAST = false>
__device__ __forceinline__ float vr_weight(float2 c0, float2 cr, float2 cd, const VrP &P)
{
    const float ux = cr.x - c0.x, vx = cr.y - c0.y, uy = cd.x - c0.x, vy = cd.y - c0.y;
    const float q = ux * ux + vx * vx + uy * uy + vy * vy + P.eps2;
    return FAST ? P.alpha2 * __builtin_amdgcn_rsqf(q) : P.alpha2 / sqrtf(q);
}
template <bool FAST> __device__ __forceinline__ float vr_quot(float n, VrRcp k) { return FAST ? n * k.r : vr_div_shared(n, k); }
template <bool FAST> __device__ __forceinline__ float vr_over_sqrt(float a, float q) { return FAST ? a * __builtin_amdgcn_rsqf(q) : a / sqrtf(q); }

__device__ __forceinline__ float2 vr_add2(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }

__global__ void __launch_bounds__(256)
k_vr_weights(const float2 *__restrict__ Wf, const float2 *__restrict__ dW, int H, int W, VrP P, float *__restrict__ wt)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const int64_t j = (int64_t)y * W + x;
    const int64_t jr = x + 1 < W ? j + 1 : j, jd = y + 1 < H ? j + W : j;
    const float2 z = make_float2(0.f, 0.f);                            // dW == nullptr: the first iteration's dW = 0
    const float2 w0 = Wf[j], d0 = dW ? dW[j] : z, wr = Wf[jr], dr = dW ? dW[jr] : z, wd = Wf[jd], dd = dW ? dW[jd] : z;
    wt[j] = vr_weight(vr_add2(w0, d0), vr_add2(wr, dr), vr_add2(wd, dd), P);
}

// WEIGHTS = true: the smoothness weights are formed here instead of in a k_vr_weights pass (same expression, so the same
// bits): every thread computes the weight of ITS pixel (written to `wt` for the SOR kernel), the left neighbour's weight
// comes from the neighbouring lane (lane 0 computes it itself), the upper neighbour's from the wave above through LDS
// (the first wave of the 64 x 4 block computes the row above itself): 1.27 weights per pixel instead of a pass of
// 16 B read + 4 B written and a 4 B read here.
template <bool WEIGHTS, bool FAST = false>
__global__ void __launch_bounds__(256)
k_vr_system(const float4 *__restrict__ D1, const float4 *__restrict__ D2, const float2 *__restrict__ Wf,
            const float2 *__restrict__ dW, const float *wt_in, int H, int W, VrP P,
            float4 *__restrict__ S, float *__restrict__ A12o, float *wt_out, VrB bs)
{
    __shared__ float s_w[4][64];
    {
        const int64_t b = blockIdx.z;
        D1 += b * bs.plane; D2 += b * bs.plane; Wf += b * bs.flow; S += b * bs.plane; A12o += b * bs.plane;
        if (dW) dW += b * bs.plane;
        if (wt_in) wt_in += b * bs.plane;
        if (wt_out) wt_out += b * bs.plane;
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x = blockIdx.x * 64 + lane, y = blockIdx.y * 4 + wv;
    const bool in = x < W && y < H;
    if (!WEIGHTS && !in) return;
    const int64_t j = in ? (int64_t)y * W + x : 0;
    const float2 z2 = make_float2(0.f, 0.f);
    const bool has_r = x + 1 < W, has_l = x > 0, has_d = y + 1 < H, has_u = y > 0;
    // EVERY load of the thread is issued here, before the first use: a missing neighbour reads the pixel itself (the
    // value the replicated border stands for), so the loads are unconditional and in flight together, and the
    // derivative planes arrive while the weights are being formed.  (Loads behind the weights' barrier, or behind
    // per-neighbour branches, cost the kernel six dependent memory round trips.)
    const int64_t jr = (in && has_r) ? j + 1 : j, jl = (in && has_l) ? j - 1 : j;
    const int64_t jd = (in && has_d) ? j + W : j, ju = (in && has_u) ? j - W : j;
    const int64_t jdl = (in && has_l && has_d) ? j + W - 1 : jl, jur = (in && has_u && has_r) ? j - W + 1 : ju;
    const float4 d1 = D1[j], d2 = D2[j];
    const float2 w0 = Wf[j], wr = Wf[jr], wlf = Wf[jl], wd = Wf[jd], wuf = Wf[ju];
    float2 d = z2, dr = z2, dd = z2, dl = z2, ddl = z2, dup = z2, dur = z2, wdl = z2, wur = z2;
    if (dW) { d = dW[j]; if (WEIGHTS) { dr = dW[jr]; dd = dW[jd]; dl = dW[jl]; ddl = dW[jdl]; } }
    if (WEIGHTS) {
        wdl = Wf[jdl];
        if (wv == 0) { wur = Wf[jur]; if (dW) { dup = dW[ju]; dur = dW[jur]; } }
    }
    float wp, wl, wu;
    if (WEIGHTS) {
        const float2 cO = vr_add2(w0, d);
        wp = vr_weight<FAST>(cO, vr_add2(wr, dr), vr_add2(wd, dd), P);      // missing neighbour = the pixel itself
        if (in) wt_out[j] = wp;
        s_w[wv][lane] = wp;
        wl = __shfl_up(wp, 1);
        if (lane == 0 && in && has_l) {
            // weight of (x - 1, y): its right neighbour is this pixel, its lower one (x - 1, y + 1) or itself
            wl = vr_weight<FAST>(vr_add2(wlf, dl), cO, vr_add2(wdl, ddl), P);
        }
        float wu0 = 0.f;
        if (wv == 0 && in && has_u) {
            // weight of (x, y - 1): its lower neighbour is this pixel, its right one (x + 1, y - 1) or itself
            wu0 = vr_weight<FAST>(vr_add2(wuf, dup), vr_add2(wur, dur), cO, P);
        }
        __syncthreads();
        wu = wv == 0 ? wu0 : s_w[wv > 0 ? wv - 1 : 0][lane];
        if (!in) return;
        wl = has_l ? wl : 0.f; wu = has_u ? wu : 0.f;
    } else {
        wp = wt_in[j]; wl = has_l ? wt_in[j - 1] : 0.f; wu = has_u ? wt_in[j - W] : 0.f;
    }
    const float Ix = d1.x, Iy = d1.y, Ixz = d1.z, Iyz = d1.w, Ixx = d2.x, Ixy = d2.y, Iyy = d2.z, Iz = d2.w;
    const float du = d.x, dv = d.y;
    // ComputeDataTerm.  The fifteen quotients whose numerator is a product of two derivative values share the
    // reciprocal work of their three denominators (vr_div_shared: bit-identical to `/` in their range, see there)
    float derivNorm = Ix * Ix + Iy * Iy + P.zeta2;
    const VrRcp k1 = vr_rcp_refined(derivNorm);
    const float Ik1z = Iz + Ix * du + Iy * dv;
    float weight = vr_over_sqrt<FAST>(P.delta2, (FAST ? Ik1z * Ik1z * k1.r : Ik1z * Ik1z / derivNorm) + P.eps2);
    float A11 = weight * vr_quot<FAST>(Ix * Ix, k1) + P.zeta2;
    float A12 = weight * vr_quot<FAST>(Ix * Iy, k1);
    float A22 = weight * vr_quot<FAST>(Iy * Iy, k1) + P.zeta2;
    float b1 = -weight * vr_quot<FAST>(Iz * Ix, k1);
    float b2 = -weight * vr_quot<FAST>(Iz * Iy, k1);
    derivNorm = Ixx * Ixx + Ixy * Ixy + P.zeta2;
    const float derivNorm2 = Iyy * Iyy + Ixy * Ixy + P.zeta2;
    const VrRcp k2 = vr_rcp_refined(derivNorm), k3 = vr_rcp_refined(derivNorm2);
    const float Ik1zx = Ixz + Ixx * du + Ixy * dv;
    const float Ik1zy = Iyz + Ixy * du + Iyy * dv;
    weight = vr_over_sqrt<FAST>(P.gamma2, (FAST ? Ik1zx * Ik1zx * k2.r + Ik1zy * Ik1zy * k3.r : Ik1zx * Ik1zx / derivNorm + Ik1zy * Ik1zy / derivNorm2) + P.eps2);
    A11 += weight * (vr_quot<FAST>(Ixx * Ixx, k2) + vr_quot<FAST>(Ixy * Ixy, k3));
    A12 += weight * (vr_quot<FAST>(Ixx * Ixy, k2) + vr_quot<FAST>(Ixy * Iyy, k3));
    A22 += weight * (vr_quot<FAST>(Ixy * Ixy, k2) + vr_quot<FAST>(Iyy * Iyy, k3));
    b1 += -weight * (vr_quot<FAST>(Ixx * Ixz, k2) + vr_quot<FAST>(Ixy * Iyz, k3));
    b2 += -weight * (vr_quot<FAST>(Ixy * Ixz, k2) + vr_quot<FAST>(Iyy * Iyz, k3));
    // smoothness: each edge (p, right) / (p, down) carries the weight of its upper-left end
    // own edge terms as the horizontal / vertical passes form them at p, neighbour edge terms as they form them at the
    // left / upper neighbour
    const float own_ux = wp * (wr.x - w0.x), own_vx = wp * (wr.y - w0.y);
    const float lft_ux = wl * (w0.x - wlf.x), lft_vx = wl * (w0.y - wlf.y);
    const float own_uy = wp * (wd.x - w0.x), own_vy = wp * (wd.y - w0.y);
    const float up_uy = wu * (w0.x - wuf.x), up_vy = wu * (w0.y - wuf.y);
    const bool red = ((x + y) & 1) == 0;
    if (red) {
        if (has_r) { b1 += own_ux; A11 += wp; b2 += own_vx; A22 += wp; }
        if (has_l) { b1 -= lft_ux; A11 += wl; b2 -= lft_vx; A22 += wl; }
        if (has_d) { b1 += own_uy; A11 += wp; b2 += own_vy; A22 += wp; }
        if (has_u) { b1 -= up_uy; A11 += wu; b2 -= up_vy; A22 += wu; }
    } else {
        if (has_l) { b1 -= lft_ux; A11 += wl; b2 -= lft_vx; A22 += wl; }
        if (has_r) { b1 += own_ux; A11 += wp; b2 += own_vx; A22 += wp; }
        if (has_u) { b1 -= up_uy; A11 += wu; b2 -= up_vy; A22 += wu; }
        if (has_d) { b1 += own_uy; A11 += wp; b2 += own_vy; A22 += wp; }
    }
    S[j] = make_float4(A11, A22, b1, b2);
    A12o[j] = A12;
}

// one colour of one red-black SOR sweep: thread -> the pixel of that colour in its pixel pair
__global__ void __launch_bounds__(256)
k_vr_sor(const float4 *__restrict__ S, const float *__restrict__ A12, const float *__restrict__ wt, int H, int W, int colour,
         float omega, float2 *__restrict__ dW)
{
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int x = 2 * (blockIdx.x * 64 + (threadIdx.x & 63)) + ((y + colour) & 1);
    if (x >= W || y >= H) return;
    const int64_t j = (int64_t)y * W + x;
    const float2 z = make_float2(0.f, 0.f);
    const float wp = wt[j], wl = x > 0 ? wt[j - 1] : 0.f, wu = y > 0 ? wt[j - W] : 0.f;
    const float2 dl = x > 0 ? dW[j - 1] : z, dr = x + 1 < W ? dW[j + 1] : z, du_ = y > 0 ? dW[j - W] : z, dd = y + 1 < H ? dW[j + W] : z;
    const float4 s = S[j];
    const float a12 = A12[j];
    float2 d = dW[j];
    const float sigmaU = wl * dl.x + wp * dr.x + wu * du_.x + wp * dd.x;
    const float sigmaV = wl * dl.y + wp * dr.y + wu * du_.y + wp * dd.y;
    d.x += omega * ((sigmaU + s.z - d.y * a12) / s.x - d.x);
    d.y += omega * ((sigmaV + s.w - d.x * a12) / s.y - d.y);
    dW[j] = d;
}

// ---- all sorIterations red-black sweeps of one fixed-point iteration in ONE pass over HBM -------------------------
// A half sweep only reads the four neighbours of the other colour, so 2 * sorIterations half sweeps of a tile need a
// halo of 2 * sorIterations pixels and nothing else.  One 512-thread workgroup owns a 108 x 84 tile + VRT_HALO = a
// 128 x 104 region: wave w owns the rows w, w + 8, ... (13 of them), lane l the pixel pair (2l, 2l + 1) of each.  Every
// thread keeps the system (A11, A22, b1, b2, A12) and the weight of ITS pixels in registers for the whole kernel, sorted
// by COLOUR (which pixel of a pair is red depends on the row parity = the wave parity, a per-wave constant); only dW and
// the weights of the left / upper neighbours live in LDS (12 B / px, 156 KB), in OpenCV's own red-black split: the
// pixels of one colour of a row are contiguous, so a wave's neighbour reads are unit-stride, all LDS addresses are one
// per-wave base + compile-time offsets, and row activity is a scalar branch.  The outermost ring is never updated and
// invalid values creep inwards one pixel per half sweep -- they stop short of the tile; half sweep s therefore skips the
// rows closer than s to the region's edge (their values can no longer reach the tile).
// Same expressions in the same order as k_vr_sor: bit-identical results (tests compare the two paths).
// dW_in == nullptr stands for dW = 0 (first fixed-point iteration); Wadd != nullptr (last iteration) makes the kernel
// store W + dW, the refined flow, instead of dW -- dW_out may then be the flow array itself (this kernel never reads W
// except at the pixel a thread is about to write).
// Measured and not adopted: delaying the odd workgroups of the first round by half a tile's time, so that loads and sweeps
// of different CUs overlap instead of running in lockstep: 5.63 / 5.70 / 5.74 / 5.63 ms per refinement for 0 / 5 / 10 / 20 us.
// Measured alternatives (12 x 5424^2, ms per step for all 110 launches): 128 x 64 tile, thread -> pair q = t + 512 k
// (row and pair parity vary inside a wave: per-pixel index arithmetic, activity predicates and value selects between the
// two pixels of a pair), 256 VGPRs: 93.2; the same with 1024 threads / 7 pairs / 128 VGPRs and a small spill: 95.9.
// Round 3, measured and not adopted (packed float instructions, v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: two operations
// per lane, 1.8 x the scalar float rate in isolation, tools/microbench/pk_rate.hip):
//  * the (u, v) components of a pixel as one register pair, the two sigma sums packed (8 instead of 16 instructions per
//    update; the own weight then has to come from LDS, or the register pairs spill): 5.36 instead of 5.46 ms for ONE
//    refinement on an idle GPU, but 299 instead of 269 ms of kernel time per 48-frame step inside bench.py, back to back
//    with the other stages (with the weight in registers and one spilled register: 338 ms) -- under sustained load the
//    packed form is the slower one;
//  * two ROWS of a thread per packed instruction, the six multiply-adds of the two rows' divisions packed as well (72
//    instead of 2 x 49 instructions per pair of updates on paper): transposes between the (u, v) layout in LDS and the
//    row-pair layout in registers, wait states between dependent packed operations and 20 spilled registers: 6.30 ms.
// What an update costs is its two sequential IEEE divisions (2 x 11 instructions, the reciprocal at quarter rate): v needs
// the new u, so they cannot be paired, and a per-pixel reciprocal kept across the ten half sweeps would need 52 more
// registers than the 229 the kernel has.
#define VRT_W 108
#define VRT_HALO 10
#define VRT_RW (VRT_W + 2 * VRT_HALO)
#define VRT_PW (VRT_RW / 2)
static_assert(VRT_PW == 64, "one lane per pixel pair of a region row");
// Tile HEIGHT and workgroup size are template parameters since round 6: <84, 512> is the kernel described
// above -- one 8-wave workgroup per CU, whose load phase (HBM-bound) and sweep phase (VALU-bound) alternate with nothing
// resident to overlap them; <32, 256> is the same code on half-height tiles (region 128 x 52, 78 KB of LDS, 4 waves x 13 rows:
// the same registers per thread), TWO workgroups per CU, so that one loads while the other sweeps -- at the price of more halo
// (region / tile = 1.93 instead of 1.47).  Measured: profiles/round6_vr_sor_notes.txt.
template <int TH, int NT> struct VrTile {
    static constexpr int H = TH, RH = TH + 2 * VRT_HALO, THREADS = NT, WAVES = NT / 64, K = RH / WAVES;
    static constexpr int LDS_BYTES = 2 * RH * VRT_PW * 12;
    static_assert(RH % WAVES == 0, "whole rows per wave");
};
typedef VrTile<84, 512> VrTileFull;
typedef VrTile<32, 256> VrTileHalf;
#define VRT_H (TILE::H)
#define VRT_RH (TILE::RH)
#define VRT_THREADS (TILE::THREADS)
#define VRT_WAVES (TILE::WAVES)
#define VRT_K (TILE::K)

template <bool FAST, typename TILE>
__global__ void __launch_bounds__(TILE::THREADS, 2)
k_vr_sor_tile(const float4 *__restrict__ S, const float *__restrict__ A12, const float *__restrict__ wt, int H, int W,
              int n_half, float omega, const float2 *__restrict__ dW_in, const float2 *Wadd, float2 *dW_out, VrB bs, int stagger)
{
    extern __shared__ __align__(16) unsigned char vr_lds[];
    if (stagger > 0 && (blockIdx.x & 1)) {                 // (experiment: odd workgroups start `stagger` x 64 clocks late)
        for (int i = 0; i < stagger; i++) __builtin_amdgcn_s_sleep(1);
    }
    {
        const int64_t b = blockIdx.z;
        S += b * bs.plane; A12 += b * bs.plane; wt += b * bs.plane;
        if (dW_in) dW_in += b * bs.plane;
        // the last fixed-point iteration writes the refined flow into the caller's array (its stride), the others dW into a plane
        if (Wadd) { Wadd += b * bs.flow; dW_out += b * bs.flow; } else dW_out += b * bs.plane;
    }
    float2 *l_dw = (float2 *)vr_lds;                                   // [2][VRT_RH][VRT_PW]
    float *l_wt = (float *)(vr_lds + 2 * VRT_RH * VRT_PW * 8);         // [2][VRT_RH][VRT_PW]
    const int x0 = blockIdx.x * VRT_W - VRT_HALO, y0 = blockIdx.y * VRT_H - VRT_HALO;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), cp = threadIdx.x & 63;
    // [ci]: image colour (0 red, 1 black).  A pixel of image colour ci has local parity lp = (r + c) & 1 = ci ^ par0 and
    // sits in column 2 cp + e with e = (lp + r) & 1; r = wv + 8 k has the parity of wv
    const int par0 = (x0 + y0) & 1;
    int e_[2], own_[2];
    bool colact[2];
#pragma unroll
    for (int ci = 0; ci < 2; ci++) {
        const int lp = ci ^ par0;
        e_[ci] = (lp + wv) & 1;
        own_[ci] = (lp * VRT_RH + wv) * VRT_PW + cp;                   // + k * VRT_WAVES * VRT_PW
        const int c = 2 * cp + e_[ci], x = x0 + c;
        colact[ci] = c > 0 && c < VRT_RW - 1 && x >= 0 && x < W;
    }
    float a11[VRT_K][2], a22[VRT_K][2], b1[VRT_K][2], b2[VRT_K][2], a12[VRT_K][2], wp[VRT_K][2];
    // Load phase, ordered so that MANY loads are in flight per wave: an LDS store right behind its own load makes the
    // compiler wait for everything issued before it (the first version ran 26 dependent round trips per thread).
    // 1. dW of one colour at a time into temporaries, then into LDS; 2. all systems and weights straight into their
    // registers; 3. the weights into LDS.
#pragma unroll
    for (int ci = 0; ci < 2; ci++) {
        float2 t[VRT_K];
        const int x = x0 + 2 * cp + e_[ci];
#pragma unroll
        for (int k = 0; k < VRT_K; k++) {
            const int y = y0 + wv + k * VRT_WAVES;
            const bool in = x >= 0 && y >= 0 && x < W && y < H;
            t[k] = make_float2(0.f, 0.f);
            if (dW_in) {                                       // uniform; nullptr = the first iteration's dW = 0
                const float2 v = dW_in[in ? (int64_t)y * W + x : 0];
                if (in) t[k] = v;
            }
        }
#pragma unroll
        for (int k = 0; k < VRT_K; k++) l_dw[own_[ci] + k * VRT_WAVES * VRT_PW] = t[k];
    }
#pragma unroll
    for (int k = 0; k < VRT_K; k++) {
        const int y = y0 + wv + k * VRT_WAVES;
#pragma unroll
        for (int ci = 0; ci < 2; ci++) {
            const int x = x0 + 2 * cp + e_[ci];
            const bool in = x >= 0 && y >= 0 && x < W && y < H;
            const int64_t p = in ? (int64_t)y * W + x : 0;
            const float4 sv = S[p];
            const float av = A12[p], wv_ = wt[p];
            a11[k][ci] = in ? sv.x : 1.f; a22[k][ci] = in ? sv.y : 1.f; b1[k][ci] = in ? sv.z : 0.f; b2[k][ci] = in ? sv.w : 0.f;
            if (FAST) { a11[k][ci] = __builtin_amdgcn_rcpf(a11[k][ci]); a22[k][ci] = __builtin_amdgcn_rcpf(a22[k][ci]); }   // divide once, multiply ten times
            a12[k][ci] = in ? av : 0.f;
            wp[k][ci] = in ? wv_ : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < VRT_K; k++) {
#pragma unroll
        for (int ci = 0; ci < 2; ci++) l_wt[own_[ci] + k * VRT_WAVES * VRT_PW] = wp[k][ci];
    }
    __syncthreads();
    for (int s = 0; s < n_half; s += 2) {
#pragma unroll
        for (int ci = 0; ci < 2; ci++) {
            // neighbours live in the other colour's plane: left = pair cp - (1 - e), right = pair cp + e, up / down = cp
            const int opb = own_[ci ^ 1], lft = opb - (1 - e_[ci]), rgt = opb + e_[ci];
            const int reach = s + ci + 1;                              // 1-based index of this half sweep
#pragma unroll
            for (int k = 0; k < VRT_K; k++) {
                const int r = wv + k * VRT_WAVES, y = y0 + r, o = k * VRT_WAVES * VRT_PW;
                if (r < reach || r > VRT_RH - 1 - reach || y < 0 || y >= H) continue;      // wave-uniform
                if (colact[ci]) {
                    const float2 dl = l_dw[lft + o], dr = l_dw[rgt + o];
                    const float2 du_ = l_dw[opb + o - VRT_PW], dd = l_dw[opb + o + VRT_PW];
                    const float wl = l_wt[lft + o], wu = l_wt[opb + o - VRT_PW];
                    const float w = wp[k][ci];
                    float2 d = l_dw[own_[ci] + o];
                    const float sigmaU = wl * dl.x + w * dr.x + wu * du_.x + w * dd.x;
                    const float sigmaV = wl * dl.y + w * dr.y + wu * du_.y + w * dd.y;
                    if (FAST) {
                        d.x += omega * ((sigmaU + b1[k][ci] - d.y * a12[k][ci]) * a11[k][ci] - d.x);
                        d.y += omega * ((sigmaV + b2[k][ci] - d.x * a12[k][ci]) * a22[k][ci] - d.y);
                    } else {
                        d.x += omega * ((sigmaU + b1[k][ci] - d.y * a12[k][ci]) / a11[k][ci] - d.x);
                        d.y += omega * ((sigmaV + b2[k][ci] - d.x * a12[k][ci]) / a22[k][ci] - d.y);
                    }
                    l_dw[own_[ci] + o] = d;
                }
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int k = 0; k < VRT_K; k++) {
        const int r = wv + k * VRT_WAVES, y = y0 + r;
        if (r < VRT_HALO || r >= VRT_HALO + VRT_H || y >= H) continue;
#pragma unroll
        for (int ci = 0; ci < 2; ci++) {
            const int c = 2 * cp + e_[ci], x = x0 + c;
            if (c < VRT_HALO || c >= VRT_HALO + VRT_W || x >= W) continue;
            const int64_t p = (int64_t)y * W + x;
            float2 d = l_dw[own_[ci] + k * VRT_WAVES * VRT_PW];
            if (Wadd) { const float2 w = Wadd[p]; d = make_float2(w.x + d.x, w.y + d.y); }   // last iteration: the refined flow
            dW_out[p] = d;
        }
    }
}

__global__ void __launch_bounds__(256)
k_vr_finish(const float2 *Wf, const float2 *__restrict__ dW, int64_t n, float2 *out)     // out may be Wf
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 w = Wf[i], d = dW[i];
    out[i] = make_float2(w.x + d.x, w.y + d.y);
}

extern "C" void tf_varref_default_params(tf_varref_params *p)
{
    p->fixed_point_iterations = 5; p->sor_iterations = 5;
    p->alpha = 20.f; p->delta = 5.f; p->gamma = 10.f; p->omega = 1.6f;
}

static size_t vr_plane(int64_t n, int64_t B) { return B > 1 ? tf_align_up((size_t)n, 64) : (size_t)n; }

extern "C" size_t tf_varref_workspace_bytes_batch(int64_t B, int64_t H, int64_t W)
{
    if (H <= 0 || W <= 0 || B <= 0) return 0;
    const size_t n = vr_plane(H * W, B) * (size_t)B;
    // D1, D2, S (float4), A12, wt (float), dW and its ping-pong partner (float2)
    return 3 * tf_align_up(n * 16, 256) + 2 * tf_align_up(n * 4, 256) + 2 * tf_align_up(n * 8, 256) + 4096;
}
extern "C" size_t tf_varref_workspace_bytes(int64_t H, int64_t W) { return tf_varref_workspace_bytes_batch(1, H, W); }

// the refinement of B images in one set of launches (grid z = image): I0 + b * img_stride, I1 + b * img_stride, flow +
// b * flow_stride (floats).  Per tile the code is the single image's, so the bits are too; what changes is the launch: at
// 1500 x 2500 a k_vr_sor_tile launch for ONE image is 282 tiles on 256 CUs -- one full round and a second one that is 10 %
// full -- and a frame pair costs 2 x 11 launches; 23 pairs at once are 6486 tiles (25.3 rounds) in 11 launches per direction.
static int vr_run(const uint8_t *I0, const uint8_t *I1, int64_t B, int64_t img_stride, int64_t H, int64_t W, const tf_varref_params *params,
                  float *flow, int64_t flow_stride, int flags, void *ws, size_t ws_bytes, void *stream)
{
    TF_REQUIRE((flags & ~(TF_VR_FAST_DIVIDE | TF_VR_FAST_SOR)) == 0, "tf_varref_ex: unknown flag");
    const bool fast = (flags & TF_VR_FAST_DIVIDE) != 0;                  // system assembly AND the sweeps
    const bool fast_sor = fast || (flags & TF_VR_FAST_SOR) != 0;        // the sweeps only
    TF_REQUIRE(I0 && I1 && flow && ws, "tf_varref: null pointer");
    TF_REQUIRE(H > 0 && W > 0 && H < 32768 && W < 32768, "tf_varref: bad shape");
    TF_REQUIRE(B >= 1 && B <= 65535, "tf_varref: bad batch size");
    TF_REQUIRE(B == 1 || (img_stride >= H * W && flow_stride >= 2 * H * W && flow_stride % 2 == 0), "tf_varref_batch: strides smaller than one image / odd flow stride");
    tf_varref_params dp;
    if (!params) { tf_varref_default_params(&dp); params = &dp; }
    TF_REQUIRE(params->fixed_point_iterations >= 0 && params->sor_iterations >= 0, "tf_varref: bad iteration counts");
    hipStream_t s = (hipStream_t)stream;
    // the fused SOR kernel covers up to VRT_HALO half sweeps; TF_VR_SOR_SWEEPS=1 selects the one-launch-per-half-sweep
    // form (same results, kept as the reference for the fused one and for larger sorIterations)
    static const bool force_sweeps = getenv("TF_VR_SOR_SWEEPS") != nullptr;
    static const bool weights_pass_env = getenv("TF_VR_WEIGHTS_PASS") != nullptr;   // development aid: separate pass
    const bool tiled = !force_sweeps && 2 * params->sor_iterations <= VRT_HALO;
    const bool tile_path = tiled && params->sor_iterations > 0;
    if (B > 1 && (!tile_path || weights_pass_env)) {
        // the forms without a batch dimension (per-half-sweep kernels, the separate weights pass): image by image
        for (int64_t b = 0; b < B; b++)
            if (const int rc = vr_run(I0 + b * img_stride, I1 + b * img_stride, 1, 0, H, W, params, flow + b * flow_stride, 0, flags, ws, ws_bytes, stream)) return rc;
        return TF_OK;
    }
    const int64_t n = H * W;
    const int64_t np = (int64_t)vr_plane(n, B);
    TfArena ar(ws, ws_bytes);
    float4 *D1 = ar.take<float4>(np * B), *D2 = ar.take<float4>(np * B), *S = ar.take<float4>(np * B);
    float *A12 = ar.take<float>(np * B), *wt = ar.take<float>(np * B);
    float2 *dW = ar.take<float2>(np * B), *dW2 = ar.take<float2>(np * B);
    if (!ar.ok()) { tf_set_error("tf_varref: workspace too small"); return TF_ENOMEM; }
    VrP P;
    P.alpha2 = params->alpha / 4; P.delta2 = params->delta / 2; P.gamma2 = params->gamma / 2; P.omega = params->omega;
    P.zeta2 = 0.1f * 0.1f; P.eps2 = 0.001f * 0.001f;
    const VrB bs = {B > 1 ? img_stride : 0, B > 1 ? flow_stride / 2 : 0, B > 1 ? np : 0};
    const unsigned Z = (unsigned)B;
    const double nb = (double)n * (double)B;
    // W is the caller's flow array itself: it is only read until the last kernel replaces it by W + dW
    const float2 *Wf = (const float2 *)flow;
    const int iH = (int)H, iW = (int)W;
    {
        TfProfScope ps(TFK_VR_PREPARE, (1.0 + 1.0 + 8.0 + 32.0) * nb, s);
        hipLaunchKernelGGL(k_vr_prepare, dim3((iW + VR_TW - 1) / VR_TW, (iH + VR_TH - 1) / VR_TH, Z), dim3(256), 0, s,
                           I0, I1, Wf, iH, iW, D1, D2, bs);
    }
    TF_CHECK_LAUNCH();
    const dim3 g1((iW + 63) / 64, (iH + 3) / 4, Z), g2(((iW + 1) / 2 + 63) / 64, (iH + 3) / 4);
    if (tiled) {
        static TfDeviceOnce once;                  // function attributes are per device
        TfDeviceOnce::Guard guard(once);
        if (guard.first) {
            TF_CHECK_HIP(hipFuncSetAttribute((const void *)k_vr_sor_tile<false, VrTileFull>, hipFuncAttributeMaxDynamicSharedMemorySize, VrTileFull::LDS_BYTES));
            TF_CHECK_HIP(hipFuncSetAttribute((const void *)k_vr_sor_tile<true, VrTileFull>, hipFuncAttributeMaxDynamicSharedMemorySize, VrTileFull::LDS_BYTES));
            TF_CHECK_HIP(hipFuncSetAttribute((const void *)k_vr_sor_tile<false, VrTileHalf>, hipFuncAttributeMaxDynamicSharedMemorySize, VrTileHalf::LDS_BYTES));
            TF_CHECK_HIP(hipFuncSetAttribute((const void *)k_vr_sor_tile<true, VrTileHalf>, hipFuncAttributeMaxDynamicSharedMemorySize, VrTileHalf::LDS_BYTES));
            guard.done();
        }
    }
    // dW = 0 at the start: the tiled path passes "no dW" to the first iteration's kernels, the sweep path needs the array
    const float2 *dW_cur = nullptr;
    if (!tile_path) { TF_CHECK_HIP(hipMemsetAsync(dW, 0, (size_t)n * 8, s)); dW_cur = dW; }
    bool flow_done = false;
    for (int it = 0; it < params->fixed_point_iterations; it++) {
        {
            // algorithmic bytes: D1 + D2 32, W 8, dW 8 read; S 16, A12 4, weight 4 written (no dW in the first iteration)
            const bool weights_pass = weights_pass_env && !fast;
            TfProfScope ps(TFK_VR_SYSTEM, (32.0 + 8.0 + (dW_cur ? 8.0 : 0.0) + 16.0 + 4.0 + 4.0) * nb, s);
            if (weights_pass) {
                hipLaunchKernelGGL(k_vr_weights, g1, dim3(256), 0, s, Wf, dW_cur, iH, iW, P, wt);
                hipLaunchKernelGGL((k_vr_system<false, false>), g1, dim3(256), 0, s, (const float4 *)D1, (const float4 *)D2, Wf,
                                   dW_cur, (const float *)wt, iH, iW, P, S, A12, (float *)nullptr, bs);
            } else if (fast)
                hipLaunchKernelGGL((k_vr_system<true, true>), g1, dim3(256), 0, s, (const float4 *)D1, (const float4 *)D2, Wf,
                                   dW_cur, (const float *)nullptr, iH, iW, P, S, A12, wt, bs);
            else
                hipLaunchKernelGGL((k_vr_system<true, false>), g1, dim3(256), 0, s, (const float4 *)D1, (const float4 *)D2, Wf,
                                   dW_cur, (const float *)nullptr, iH, iW, P, S, A12, wt, bs);
        }
        TF_CHECK_LAUNCH();
        if (tile_path) {
            // algorithmic bytes: system 20 + weight 4 + dW 8 read, dW 8 written, once per fixed-point iteration (no dW
            // to read in the first; W 8 more to read in the last, which writes the refined flow)
            const bool last = it == params->fixed_point_iterations - 1;
            TfProfScope ps(TFK_VR_SOR, (20.0 + 4.0 + (dW_cur ? 8.0 : 0.0) + (last ? 8.0 : 0.0) + 8.0) * nb, s);
            float2 *dst = last ? (float2 *)flow : (dW_cur == dW ? dW2 : dW);
            // TF_VR_TILE=half: half-height tiles, two 4-wave workgroups per CU (VrTileHalf); TF_VR_STAGGER=<n>: odd workgroups
            // start n x 64 clocks late (development switches of the round-6 experiment; default: the full tile, no stagger)
            static const bool half_env = getenv("TF_VR_TILE") && !strcmp(getenv("TF_VR_TILE"), "half");
            static const int stagger_env = getenv("TF_VR_STAGGER") ? atoi(getenv("TF_VR_STAGGER")) : 0;
            const float2 *wadd = last ? Wf : (const float2 *)nullptr;
#define VR_LAUNCH_TILE(FASTV, TILET)                                                                                              \
            hipLaunchKernelGGL((k_vr_sor_tile<FASTV, TILET>), dim3((iW + VRT_W - 1) / VRT_W, (iH + TILET::H - 1) / TILET::H, Z),      \
                               dim3(TILET::THREADS), TILET::LDS_BYTES, s, (const float4 *)S, (const float *)A12, (const float *)wt,  \
                               iH, iW, 2 * params->sor_iterations, P.omega, dW_cur, wadd, dst, bs, stagger_env)
            if (half_env) { if (fast_sor) VR_LAUNCH_TILE(true, VrTileHalf); else VR_LAUNCH_TILE(false, VrTileHalf); }
            else { if (fast_sor) VR_LAUNCH_TILE(true, VrTileFull); else VR_LAUNCH_TILE(false, VrTileFull); }
#undef VR_LAUNCH_TILE
            dW_cur = dst;
            flow_done = last;
        } else {
            TfProfScope ps(TFK_VR_SOR, (20.0 + 4.0 + 8.0 + 8.0) * (double)n * params->sor_iterations, s);
            for (int k = 0; k < params->sor_iterations; k++) {
                hipLaunchKernelGGL(k_vr_sor, g2, dim3(256), 0, s, (const float4 *)S, (const float *)A12, (const float *)wt, iH, iW, 0, P.omega, dW);
                hipLaunchKernelGGL(k_vr_sor, g2, dim3(256), 0, s, (const float4 *)S, (const float *)A12, (const float *)wt, iH, iW, 1, P.omega, dW);
            }
        }
        TF_CHECK_LAUNCH();
    }
    if (!flow_done) {
        // no tiled last iteration (sweep path, or no iterations at all; B == 1 here): flow = W + dW, with dW = 0 if nothing ran
        for (int64_t b = 0; b < B; b++) {
            if (!dW_cur) { TF_CHECK_HIP(hipMemsetAsync(dW, 0, (size_t)n * 8, s)); dW_cur = dW; }
            hipLaunchKernelGGL(k_vr_finish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, Wf + b * bs.flow, dW_cur + b * bs.plane, n, (float2 *)flow + b * bs.flow);
            TF_CHECK_LAUNCH();
        }
    }
    return TF_OK;
}

extern "C" int tf_varref_ex(const uint8_t *I0, const uint8_t *I1, int64_t H, int64_t W, const tf_varref_params *params,
                            float *flow, int flags, void *ws, size_t ws_bytes, void *stream)
{
    return vr_run(I0, I1, 1, 0, H, W, params, flow, 0, flags, ws, ws_bytes, stream);
}

extern "C" int tf_varref(const uint8_t *I0, const uint8_t *I1, int64_t H, int64_t W, const tf_varref_params *params,
                         float *flow, void *ws, size_t ws_bytes, void *stream)
{
    return tf_varref_ex(I0, I1, H, W, params, flow, 0, ws, ws_bytes, stream);
}

extern "C" int tf_varref_batch(const uint8_t *I0, const uint8_t *I1, int64_t B, int64_t img_stride, int64_t H, int64_t W,
                               const tf_varref_params *params, float *flow, int64_t flow_stride, int flags,
                               void *ws, size_t ws_bytes, void *stream)
{
    if (ws_bytes < tf_varref_workspace_bytes_batch(B, H, W)) {
        // (a caller that sized for one image still gets its images refined -- one after the other)
        TF_REQUIRE(B >= 1 && ws_bytes >= tf_varref_workspace_bytes(H, W), "tf_varref_batch: workspace too small");
        for (int64_t b = 0; b < B; b++)
            if (const int rc = vr_run(I0 + b * img_stride, I1 + b * img_stride, 1, 0, H, W, params, flow + b * flow_stride, 0, flags, ws, ws_bytes, stream)) return rc;
        return TF_OK;
    }
    return vr_run(I0, I1, B, img_stride, H, W, params, flow, flow_stride, flags, ws, ws_bytes, stream);
}

// development aid: number of operand pairs (out of `count`, drawn from the range vr_div_shared is used in) for which the
// shared-reciprocal division differs from the hardware's correctly rounded one.  Must be 0.
extern "C" int tf_selftest_shared_divide(int64_t count, uint64_t seed, uint64_t *mismatches_host, void *stream)
{
    TF_REQUIRE(mismatches_host && count > 0, "tf_selftest_shared_divide: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *d_m = nullptr;
    TF_CHECK_HIP(hipMalloc(&d_m, 8));
    TF_CHECK_HIP(hipMemsetAsync(d_m, 0, 8, s));
    hipLaunchKernelGGL(k_vr_selftest_divide, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, (unsigned long long)seed, count, d_m);
    unsigned long long h = 0;
    hipError_t e = hipMemcpyAsync(&h, d_m, 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d_m);
    TF_CHECK_HIP(e);
    *mismatches_host = h;
    return TF_OK;
}
