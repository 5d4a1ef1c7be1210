// scipy.ndimage glue of the detection recipes on the GPU, bit-exact with SciPy:
//   tf_binary_morph   ndi.binary_erosion / binary_dilation with a 3x3x3 structuring element, `iterations`
//                     and `border_value` (detection.py:80-93, 105-119, 509, 551-553, 571-573, 608-616)
//   tf_linearise      utils/normalisation_utils.py:36-56 linearise_field
//   tf_label_extent   analysis.py:15-35 find_object_lengths + :38-63 mask_labels (per-label t-extent / hit flag)
//   tf_apply_lut      utils/label_utils.py:265-309 remap_labels' final gather
// All are one-pass HBM-bound stencils / reductions on uint8 / int32 volumes.
#include "tf_common.h"
#include <initializer_list>
#include <string.h>
#include <stdlib.h>

struct MorphTaps { int n; int8_t dt[27], dy[27], dx[27]; };

// op 0: erosion  out = AND_d in[p + d]   (out of volume -> border)
// op 1: dilation out = OR_d  in[p - d]   (SciPy reflects the structure; out of volume -> border)
__global__ void __launch_bounds__(256)
k_binary_morph(const uint8_t *__restrict__ in, int64_t T, int H, int W, MorphTaps tp, int op, int border,
               uint8_t *__restrict__ out)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    const int64_t t = blockIdx.z;
    if (x >= W || y >= H) return;
    const int64_t plane = (int64_t)H * W;
    bool r = op == 0;
    for (int i = 0; i < tp.n; i++) {
        const int s = op == 0 ? 1 : -1;
        const int64_t tt = t + s * tp.dt[i];
        const int yy = y + s * tp.dy[i], xx = x + s * tp.dx[i];
        bool v;
        if (tt < 0 || tt >= T || yy < 0 || yy >= H || xx < 0 || xx >= W) v = border != 0;
        else v = in[tt * plane + (int64_t)yy * W + xx] != 0;
        if (op == 0) { if (!v) { r = false; break; } }
        else if (v) { r = true; break; }
    }
    out[t * plane + (int64_t)y * W + x] = r ? 1 : 0;
}

// Four pixels per thread (one 32-bit word of the uint8 volume; W % 4 == 0).  The taps are grouped by (dt, dy) row: a row
// contributes its centre word and, for dx = -1 / +1, the word shifted by one byte with the neighbour word's edge byte
// (or the border value at the volume's edge) -- three word loads per row instead of up to twelve byte loads.
// Bytes are normalised to 0 / 1 first, so erosion = bitwise AND and dilation = bitwise OR over the taps.
struct MorphRows { int n; int8_t dt[9], dy[9]; uint8_t dxmask[9]; };   // dxmask bit 0: dx = -1, bit 1: dx = 0, bit 2: dx = +1

__device__ __forceinline__ uint32_t morph_norm(uint32_t w) {
    w |= w >> 4; w |= w >> 2; w |= w >> 1;
    return w & 0x01010101u;
}

__global__ void __launch_bounds__(256)
k_binary_morph4(const uint32_t *__restrict__ in, int64_t T, int H, int W4, MorphRows rw, int op, int border,
                uint32_t *__restrict__ out)
{
    const int x4 = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    const int64_t t = blockIdx.z;
    if (x4 >= W4 || y >= H) return;
    const int64_t plane4 = (int64_t)H * W4;
    const uint32_t bw = border ? 0x01010101u : 0u, bb = border ? 1u : 0u;
    uint32_t r = op == 0 ? 0x01010101u : 0u;
    const int s = op == 0 ? 1 : -1;                   // SciPy reflects the structure for the dilation
    for (int i = 0; i < rw.n; i++) {
        const int64_t tt = t + s * rw.dt[i];
        const int yy = y + s * rw.dy[i];
        uint32_t c = bw, l = bb, rt = bb;             // centre word; byte left of it; byte right of it
        if (tt >= 0 && tt < T && yy >= 0 && yy < H) {
            const uint32_t *row = in + tt * plane4 + (int64_t)yy * W4;
            c = morph_norm(row[x4]);
            if (rw.dxmask[i] & 5) {
                if (x4 > 0) l = morph_norm(row[x4 - 1]) >> 24;
                if (x4 + 1 < W4) rt = morph_norm(row[x4 + 1]) & 1u;
            }
        }
        const uint32_t m = rw.dxmask[i];
        // pixel p of the word sees in[x + s * dx]: dx = -1 (bit 0) is the left neighbour for the erosion, the right one for the dilation
        const uint32_t left = (c << 8) | l, right = (c >> 8) | (rt << 24);
        const uint32_t lo = s == 1 ? left : right, hi = s == 1 ? right : left;
        if (op == 0) {
            if (m & 1) r &= lo;
            if (m & 2) r &= c;
            if (m & 4) r &= hi;
        } else {
            if (m & 1) r |= lo;
            if (m & 2) r |= c;
            if (m & 4) r |= hi;
        }
    }
    out[t * plane4 + (int64_t)y * W4 + x4] = r;
}

// Sixteen pixels per thread (one uint4; W % 16 == 0, buffers 16-byte aligned): per (dt, dy) row one 16-byte load plus,
// for dx = -1 / +1, the two neighbour words for the edge bytes -- a quarter of the load instructions of the word form
// (2.2 -> 0.87 ms per call on a 16 x 5424^2 window).
__global__ void __launch_bounds__(256)
k_binary_morph16(const uint4 *__restrict__ in, int64_t T, int H, int W16, MorphRows rw, int op, int border,
                 uint4 *__restrict__ out)
{
    // Round 6: a 1-D grid whose workgroup -> (tile, t) mapping is XCD-aware.  Workgroup L runs on XCD L % 8 (round-robin
    // dispatch); within an XCD consecutive workgroups take consecutive time steps t of ONE (x, y) tile, so the up to three
    // planes a tap row reads (t - 1, t, t + 1) are requested by workgroups that share an L2 and run back to back.  With the
    // (x, y, t) grid of rounds 2 - 5 a plane was read again a whole plane's worth of workgroups later -- from the fabric:
    // 2.18 x the algorithmic bytes by FETCH_SIZE for the 3 x 3 x 3 structure.
    // A tile is 1024 x 8 pixels: every thread forms TWO output rows (y, y + 4), so that the two halo rows of a tile are read
    // per eight output rows instead of per four (1.25 x instead of 1.5 x the rows; the rows the two halves share come from L1).
    const int tiles_x = (W16 + 63) / 64, tiles_y = (H + 7) / 8;
    const int64_t n_tiles = (int64_t)tiles_x * tiles_y, L = blockIdx.x;
    int64_t slot = L >> 3, tile = (slot / T) * 8 + (L & 7), t = slot % T;
    if (border & 2) {                                 // (A/B switch TF_MORPH_GRID=plane: the (x, y, t) order of rounds 2 - 5)
        const int64_t padded = (n_tiles + 7) / 8 * 8;
        tile = L % padded; t = L / padded;
        border &= 1;
    }
    if (tile >= n_tiles) return;
    const int x16 = (int)(tile % tiles_x) * 64 + threadIdx.x;
    if (x16 >= W16) return;
    const int64_t plane16 = (int64_t)H * W16;
    const uint32_t bw = border ? 0x01010101u : 0u, bb = border ? 1u : 0u;
    const uint32_t init = op == 0 ? 0x01010101u : 0u;
    const int s = op == 0 ? 1 : -1;                   // SciPy reflects the structure for the dilation
    for (int half = 0; half < 2; half++) {
    const int y = (int)(tile / tiles_x) * 8 + threadIdx.y + 4 * half;
    if (y >= H) break;
    uint32_t r[4] = {init, init, init, init};
    for (int i = 0; i < rw.n; i++) {
        const int64_t tt = t + s * rw.dt[i];
        const int yy = y + s * rw.dy[i];
        uint32_t c[4] = {bw, bw, bw, bw}, l = bb, rt = bb;
        if (tt >= 0 && tt < T && yy >= 0 && yy < H) {
            const uint4 *row = in + tt * plane16 + (int64_t)yy * W16;
            const uint4 v = row[x16];
            c[0] = morph_norm(v.x); c[1] = morph_norm(v.y); c[2] = morph_norm(v.z); c[3] = morph_norm(v.w);
            if (rw.dxmask[i] & 5) {
                const uint32_t *words = (const uint32_t *)row;
                if (x16 > 0) l = morph_norm(words[4 * x16 - 1]) >> 24;
                if (x16 + 1 < W16) rt = morph_norm(words[4 * x16 + 4]) & 1u;
            }
        }
        const uint32_t m = rw.dxmask[i];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t lb = k == 0 ? l : c[k - 1] >> 24, rb = k == 3 ? rt : c[k + 1] & 1u;
            const uint32_t left = (c[k] << 8) | lb, right = (c[k] >> 8) | (rb << 24);
            const uint32_t lo = s == 1 ? left : right, hi = s == 1 ? right : left;
            if (op == 0) {
                if (m & 1) r[k] &= lo;
                if (m & 2) r[k] &= c[k];
                if (m & 4) r[k] &= hi;
            } else {
                if (m & 1) r[k] |= lo;
                if (m & 2) r[k] |= c[k];
                if (m & 4) r[k] |= hi;
            }
        }
    }
    out[t * plane16 + (int64_t)y * W16 + x16] = make_uint4(r[0], r[1], r[2], r[3]);
    }
}

extern "C" int tf_binary_morph(const uint8_t *in, int64_t T, int64_t H, int64_t W, const uint8_t *structure_host,
                               int op, int iterations, int border_value, uint8_t *out, uint8_t *tmp, void *stream)
{
    TF_REQUIRE(in && out && structure_host, "tf_binary_morph: null pointer");
    TF_REQUIRE(T > 0 && H > 0 && W > 0 && T < 65536 && H < (1 << 15) && W < (1 << 15), "tf_binary_morph: bad shape");
    TF_REQUIRE(op == 0 || op == 1, "tf_binary_morph: op must be 0 (erosion) or 1 (dilation)");
    TF_REQUIRE(iterations >= 1, "tf_binary_morph: iterations must be >= 1");
    TF_REQUIRE(iterations == 1 || tmp, "tf_binary_morph: iterations > 1 needs a tmp buffer");
    TF_REQUIRE(out != in && tmp != in && (tmp != out || iterations == 1), "tf_binary_morph: buffers alias");
    MorphTaps tp; tp.n = 0;
    for (int p = 0; p < 3; p++) for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++)
        if (structure_host[p * 9 + r * 3 + c]) { tp.dt[tp.n] = (int8_t)(p - 1); tp.dy[tp.n] = (int8_t)(r - 1); tp.dx[tp.n] = (int8_t)(c - 1); tp.n++; }
    TF_REQUIRE(tp.n > 0, "tf_binary_morph: empty structure");
    hipStream_t s = (hipStream_t)stream;
    dim3 block(64, 4), grid((unsigned)((W + 63) / 64), (unsigned)((H + 3) / 4), (unsigned)T);
    MorphRows rw; rw.n = 0;
    for (int p = 0; p < 3; p++) for (int r = 0; r < 3; r++) {
        uint8_t m = 0;
        for (int c = 0; c < 3; c++) if (structure_host[p * 9 + r * 3 + c]) m |= (uint8_t)(1 << c);
        if (m) { rw.dt[rw.n] = (int8_t)(p - 1); rw.dy[rw.n] = (int8_t)(r - 1); rw.dxmask[rw.n] = m; rw.n++; }
    }
    // word form: rows of whole words, every buffer 4-byte aligned
    const bool words = W % 4 == 0 && ((uintptr_t)in % 4 == 0) && ((uintptr_t)out % 4 == 0) && (!tmp || (uintptr_t)tmp % 4 == 0);
    const dim3 grid4((unsigned)((W / 4 + 63) / 64), (unsigned)((H + 3) / 4), (unsigned)T);
    const bool quads = W % 16 == 0 && ((uintptr_t)in % 16 == 0) && ((uintptr_t)out % 16 == 0) && (!tmp || (uintptr_t)tmp % 16 == 0);
    // tiles rounded up to a multiple of 8 (one per XCD and slot group), times T time steps
    const int64_t tiles16 = (int64_t)((W / 16 + 63) / 64) * ((H + 7) / 8);
    TF_REQUIRE(!quads || ((tiles16 + 7) / 8) * 8 * T < (1ll << 31), "tf_binary_morph: volume too large for the 1-D grid");
    const dim3 grid16((unsigned)(((tiles16 + 7) / 8) * 8 * T), 1, 1);
    const uint8_t *src = in;
    for (int it = 0; it < iterations; it++) {
        // ping-pong so that the last iteration writes `out`
        uint8_t *dst = ((iterations - 1 - it) % 2 == 0) ? out : tmp;
        TfProfScope ps(TFK_MORPH, 2.0 * (double)T * H * W, s);
        static const bool plane_grid = getenv("TF_MORPH_GRID") && !strcmp(getenv("TF_MORPH_GRID"), "plane");
        if (quads) hipLaunchKernelGGL(k_binary_morph16, grid16, block, 0, s, (const uint4 *)src, T, (int)H, (int)(W / 16), rw, op, (border_value ? 1 : 0) | (plane_grid ? 2 : 0), (uint4 *)dst);   // (grid16: 1-D, XCD-aware mapping inside)
        else if (words) hipLaunchKernelGGL(k_binary_morph4, grid4, block, 0, s, (const uint32_t *)src, T, (int)H, (int)(W / 4), rw, op, border_value, (uint32_t *)dst);
        else hipLaunchKernelGGL(k_binary_morph, grid, block, 0, s, src, T, (int)H, (int)W, tp, op, border_value, dst);
        src = dst;
    }
    TF_CHECK_LAUNCH();
    return TF_OK;
}

// linearise_field: clip((f - lo) / (hi - lo), 0, 1), reversed thresholds flip the ramp (float32 like numpy on float32)
__global__ void __launch_bounds__(256)
k_linearise(const float *__restrict__ f, int64_t n, float lo, float hi, int flip, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = (f[i] - lo) / (hi - lo);
    v = (v != v) ? v : fminf(v, 1.f);          // np.minimum / np.maximum propagate NaN
    v = (v != v) ? v : fmaxf(v, 0.f);
    out[i] = flip ? 1.f - v : v;
}

// Four elements per thread (16-byte accesses) for the one-pass elementwise kernels of this file: with one 4-byte
// element per thread they moved ~3.5 TB/s.  tf_vec4_ok: every 4-byte-element array 16-byte aligned, every byte array
// 4-byte aligned; the last thread handles the n % 4 tail element by element.
static bool tf_vec4_ok(std::initializer_list<const void *> words, std::initializer_list<const void *> bytes)
{
    for (const void *p : words) if ((uintptr_t)p % 16) return false;
    for (const void *p : bytes) if ((uintptr_t)p % 4) return false;
    return true;
}

__global__ void __launch_bounds__(256)
k_linearise4(const float *__restrict__ f, int64_t n, float lo, float hi, int flip, float *__restrict__ out)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    auto one = [&](float x) {
        float v = (x - lo) / (hi - lo);
        v = (v != v) ? v : fminf(v, 1.f);
        v = (v != v) ? v : fmaxf(v, 0.f);
        return flip ? 1.f - v : v;
    };
    if (i + 3 < n) {
        const float4 x = *(const float4 *)(f + i);
        *(float4 *)(out + i) = make_float4(one(x.x), one(x.y), one(x.z), one(x.w));
    } else {
        for (int64_t j = i; j < n; j++) out[j] = one(f[j]);
    }
}

extern "C" int tf_linearise(const float *field, int64_t n, double lower, double upper, float *out, void *stream)
{
    TF_REQUIRE(field && out && n > 0, "tf_linearise: bad arguments");
    TF_REQUIRE(lower != upper, "tf_linearise: lower and upper thresholds must have different values");
    int flip = 0;
    if (lower > upper) { const double t = lower; lower = upper; upper = t; flip = 1; }
    if (tf_vec4_ok({field, out}, {}))
        hipLaunchKernelGGL(k_linearise4, dim3((unsigned)(((n + 3) / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, field, n,
                           (float)lower, (float)upper, flip, out);
    else
    hipLaunchKernelGGL(k_linearise, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, field, n,
                       (float)lower, (float)upper, flip, out);
    TF_CHECK_LAUNCH();
    return TF_OK;
}

// per-label first / last time step and "overlaps the mask" flag: tmin/tmax int32[n_labels + 1], hit u8[n_labels + 1]
__global__ void __launch_bounds__(256)
k_label_extent(const int32_t *__restrict__ labels, const uint8_t *__restrict__ mask, int64_t plane, int n_labels,
               int *__restrict__ tmin, int *__restrict__ tmax, uint8_t *__restrict__ hit)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int t = blockIdx.y;
    if (i >= plane) return;
    const int64_t p = (int64_t)t * plane + i;
    const int32_t l = labels[p];
    if (l <= 0 || l > n_labels) return;
    if (tmin[l] > t) atomicMin(&tmin[l], t);
    if (tmax[l] < t) atomicMax(&tmax[l], t);
    if (mask && mask[p] && !hit[l]) hit[l] = 1;
}

extern "C" int tf_label_extent(const int32_t *labels, const uint8_t *mask, int64_t T, int64_t H, int64_t W, int n_labels,
                               int *tmin, int *tmax, uint8_t *hit, void *stream)
{
    TF_REQUIRE(labels && tmin && tmax && hit && n_labels >= 0, "tf_label_extent: bad arguments");
    TF_REQUIRE(T > 0 && T < 65536 && H > 0 && W > 0, "tf_label_extent: bad shape");
    hipStream_t s = (hipStream_t)stream;
    TF_CHECK_HIP(hipMemsetAsync(tmin, 0x7f, (size_t)(n_labels + 1) * sizeof(int), s));
    TF_CHECK_HIP(hipMemsetAsync(tmax, 0xff, (size_t)(n_labels + 1) * sizeof(int), s));       // -1
    TF_CHECK_HIP(hipMemsetAsync(hit, 0, (size_t)(n_labels + 1), s));
    const int64_t plane = H * W;
    hipLaunchKernelGGL(k_label_extent, dim3((unsigned)((plane + 255) / 256), (unsigned)T), dim3(256), 0, s,
                       labels, mask, plane, n_labels, tmin, tmax, hit);
    TF_CHECK_LAUNCH();
    return TF_OK;
}

__global__ void __launch_bounds__(256)
k_apply_lut(const int32_t *__restrict__ labels, int64_t n, const int32_t *__restrict__ lut, int n_lut, int32_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t l = labels[i];
    out[i] = (l >= 0 && l < n_lut) ? lut[l] : 0;
}

extern "C" int tf_apply_lut(const int32_t *labels, int64_t n, const int32_t *lut, int n_lut, int32_t *out, void *stream)
{
    TF_REQUIRE(labels && lut && out && n > 0 && n_lut > 0, "tf_apply_lut: bad arguments");
    hipLaunchKernelGGL(k_apply_lut, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, labels, n, lut, n_lut, out);
    TF_CHECK_LAUNCH();
    return TF_OK;
}

// the same, ids <= 0 kept as they are (the window stitch: background seeds -1 and unlabelled 0 pass through)
__global__ void __launch_bounds__(256)
k_apply_lut_keep(const int32_t *__restrict__ labels, int64_t n, const int32_t *__restrict__ lut, int n_lut, int32_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t l = labels[i];
    out[i] = l <= 0 ? l : (l < n_lut ? lut[l] : 0);
}

__global__ void __launch_bounds__(256)
k_apply_lut_keep4(const int32_t *__restrict__ labels, int64_t n, const int32_t *__restrict__ lut, int n_lut, int32_t *__restrict__ out)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    auto one = [&](int32_t l) { return l <= 0 ? l : (l < n_lut ? lut[l] : 0); };
    if (i + 3 < n) {
        const int4 l = *(const int4 *)(labels + i);
        *(int4 *)(out + i) = make_int4(one(l.x), one(l.y), one(l.z), one(l.w));
    } else {
        for (int64_t j = i; j < n; j++) out[j] = one(labels[j]);
    }
}

extern "C" int tf_apply_lut_keep_nonpositive(const int32_t *labels, int64_t n, const int32_t *lut, int n_lut, int32_t *out, void *stream)
{
    TF_REQUIRE(labels && lut && out && n > 0 && n_lut > 0, "tf_apply_lut_keep_nonpositive: bad arguments");
    if (tf_vec4_ok({labels, out}, {}))
        hipLaunchKernelGGL(k_apply_lut_keep4, dim3((unsigned)(((n + 3) / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, labels, n, lut, n_lut, out);
    else
    hipLaunchKernelGGL(k_apply_lut_keep, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, labels, n, lut, n_lut, out);
    TF_CHECK_LAUNCH();
    return TF_OK;
}

// ---- the elementwise glue of detect_anvils' seeds in two passes (detection.py:547-561, 590-617) ----------------------------
// tf_field_masks: ge1 = field >= 1 (markers = field >= 1, :551-552), le0 = (field <= 0) | isnan(field) (the mask
// get_watershed_mask erodes, :608-609), isnan = isnan(field) (:607, :616): three byte masks from one read of the field.
// tf_merge_seeds: seeds = (bg | isnan) ? -1 : comp  (`mask[wh_field_nan] = True; eroded_markers[mask] = -1`, :616, :558).
__global__ void __launch_bounds__(256)
k_field_masks(const float *__restrict__ f, int64_t n, uint8_t *__restrict__ ge1, uint8_t *__restrict__ le0, uint8_t *__restrict__ isn)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = f[i];
    const bool nan = v != v;
    ge1[i] = v >= 1.f; le0[i] = (v <= 0.f) || nan; isn[i] = nan;
}
__global__ void __launch_bounds__(256)
k_merge_seeds(const int32_t *__restrict__ comp, const uint8_t *__restrict__ bg, const uint8_t *__restrict__ isn, int64_t n,
              int32_t *__restrict__ seeds)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    seeds[i] = (bg[i] | isn[i]) ? -1 : comp[i];
}
__global__ void __launch_bounds__(256)
k_field_masks4(const float *__restrict__ f, int64_t n, uint8_t *__restrict__ ge1, uint8_t *__restrict__ le0, uint8_t *__restrict__ isn)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 3 < n) {
        const float4 x = *(const float4 *)(f + i);
        const float v[4] = {x.x, x.y, x.z, x.w};
        uint32_t a = 0, b = 0, c = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool nan = v[k] != v[k];
            a |= (uint32_t)(v[k] >= 1.f) << (8 * k); b |= (uint32_t)((v[k] <= 0.f) || nan) << (8 * k); c |= (uint32_t)nan << (8 * k);
        }
        *(uint32_t *)(ge1 + i) = a; *(uint32_t *)(le0 + i) = b; *(uint32_t *)(isn + i) = c;
    } else {
        for (int64_t j = i; j < n; j++) {
            const float v = f[j];
            const bool nan = v != v;
            ge1[j] = v >= 1.f; le0[j] = (v <= 0.f) || nan; isn[j] = nan;
        }
    }
}
__global__ void __launch_bounds__(256)
k_merge_seeds4(const int32_t *__restrict__ comp, const uint8_t *__restrict__ bg, const uint8_t *__restrict__ isn, int64_t n,
               int32_t *__restrict__ seeds)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 3 < n) {
        const int4 c = *(const int4 *)(comp + i);
        const uint32_t m = *(const uint32_t *)(bg + i) | *(const uint32_t *)(isn + i);
        *(int4 *)(seeds + i) = make_int4((m & 0xffu) ? -1 : c.x, (m & 0xff00u) ? -1 : c.y, (m & 0xff0000u) ? -1 : c.z, (m & 0xff000000u) ? -1 : c.w);
    } else {
        for (int64_t j = i; j < n; j++) seeds[j] = (bg[j] | isn[j]) ? -1 : comp[j];
    }
}
extern "C" int tf_field_masks(const float *field, int64_t n, uint8_t *ge1, uint8_t *le0_or_nan, uint8_t *isnan_out, void *stream)
{
    TF_REQUIRE(field && ge1 && le0_or_nan && isnan_out && n > 0, "tf_field_masks: bad arguments");
    if (tf_vec4_ok({field}, {ge1, le0_or_nan, isnan_out}))
        hipLaunchKernelGGL(k_field_masks4, dim3((unsigned)(((n + 3) / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, field, n, ge1, le0_or_nan, isnan_out);
    else
    hipLaunchKernelGGL(k_field_masks, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, field, n, ge1, le0_or_nan, isnan_out);
    TF_CHECK_LAUNCH();
    return TF_OK;
}
extern "C" int tf_merge_seeds(const int32_t *comp, const uint8_t *bg, const uint8_t *isnan_in, int64_t n, int32_t *seeds, void *stream)
{
    TF_REQUIRE(comp && bg && isnan_in && seeds && n > 0, "tf_merge_seeds: bad arguments");
    if (tf_vec4_ok({comp, seeds}, {bg, isnan_in}))
        hipLaunchKernelGGL(k_merge_seeds4, dim3((unsigned)(((n + 3) / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, comp, bg, isnan_in, n, seeds);
    else
    hipLaunchKernelGGL(k_merge_seeds, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, comp, bg, isnan_in, n, seeds);
    TF_CHECK_LAUNCH();
    return TF_OK;
}

// ---- connected-component labelling: scipy.ndimage.label(input, structure) ---------------------------------
// Union-find on the GPU.  Roots are the smallest raster index of each component, so numbering components by
// ascending root (an exclusive scan over the root flags) reproduces SciPy's numbering, which labels
// components in the order their first pixel is met in a raster scan.  `structure` must be centro-symmetric
// (SciPy requires that too); only the "forward" half of its offsets is needed for the unions.
#include "tf_prim.h"

__device__ __forceinline__ int ccl_find(int *__restrict__ parent, int p) {
    int q = parent[p];
    while (q != p) { p = q; q = parent[p]; }
    return p;
}

__device__ __forceinline__ void ccl_union(int *__restrict__ parent, int a, int b) {
    for (;;) {
        a = ccl_find(parent, a); b = ccl_find(parent, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }        // a > b: attach the larger root to the smaller
        const int old = atomicMin(&parent[a], b);
        if (old == a) return;
        a = old;                                             // somebody else re-rooted a meanwhile: retry from there
    }
}

__global__ void __launch_bounds__(256)
k_ccl_init(const uint8_t *__restrict__ in, int64_t n, int *__restrict__ parent) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) parent[i] = in[i] ? (int)i : -1;
}
__global__ void __launch_bounds__(256)
k_ccl_init4(const uint8_t *__restrict__ in, int64_t n, int *__restrict__ parent) {     // four pixels per thread (tf_vec4_ok)
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 3 < n) {
        const uint32_t m = *(const uint32_t *)(in + i);
        const int b = (int)i;
        *(int4 *)(parent + i) = make_int4((m & 0xffu) ? b : -1, (m & 0xff00u) ? b + 1 : -1, (m & 0xff0000u) ? b + 2 : -1, (m & 0xff000000u) ? b + 3 : -1);
    } else {
        for (int64_t j = i; j < n; j++) parent[j] = in[j] ? (int)j : -1;
    }
}

// RUNS (round 5).  With one union per pixel and forward tap a big component -- the background that binary_fill_holes labels:
// 95 % of a frame -- costs ~2 atomics and two root searches per pixel on ever longer chains (168 ms per 16 x 5424^2 volume, the
// largest kernel of detect_cores).  When the structure holds the horizontal tap (every connectivity does), the pixels of a
// horizontal run are connected whatever else happens, so: (i) the initial parent of a pixel is the HEAD of its run inside
// the wave's 64-pixel row segment (one ballot: no union, no atomic; the head is the smallest index, as the numbering needs);
// (ii) a run that continues across a segment boundary is joined there once; (iii) a tap without a horizontal component (dy
// or dt only) joins p and q only where the overlap of the two runs STARTS -- if the left neighbours of both are set they are
// joined by the thread to the left, and p, q hang on them through their runs.  Unions per volume fall from ~2 per pixel to
// ~2 per run overlap.  (iv) A tap WITH a horizontal component (connectivity 2, 3) is skipped where its sibling without it does
// the job: (x, y) -> (x + dx, y') is implied by (x, y) -> (x, y') plus the run of row y' when pixel (x, y') is set and the
// structure holds that sibling tap (`sibling` bit i).
__global__ void __launch_bounds__(256)
k_ccl_init_runs(const uint8_t *__restrict__ in, int64_t T, int H, int W, int *__restrict__ parent)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    const int64_t t = blockIdx.z;
    const bool inside = x < W && y < H;
    const int64_t plane = (int64_t)H * W, p = t * plane + (int64_t)y * W + x;
    const bool on = inside && in[p] != 0;
    const unsigned long long m = __ballot(on);                      // (a wave = one row segment: blockDim.x == 64)
    if (!inside) return;
    if (!on) { parent[p] = -1; return; }
    const int lane = threadIdx.x;
    const unsigned long long zeros_below = ~m & ((1ull << lane) - 1ull);
    const int head = zeros_below ? 64 - __clzll((long long)zeros_below) : 0;
    parent[p] = (int)(p - (lane - head));
}

template <bool RUNS>
__global__ void __launch_bounds__(256)
k_ccl_union(const uint8_t *__restrict__ in, int64_t T, int H, int W, MorphTaps tp, unsigned sibling, int *__restrict__ parent)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    const int64_t t = blockIdx.z;
    if (x >= W || y >= H) return;
    const int64_t plane = (int64_t)H * W, p = t * plane + (int64_t)y * W + x;
    if (!in[p]) return;
    const bool left = RUNS && x > 0 && in[p - 1] != 0;
    if (RUNS && left && threadIdx.x == 0) ccl_union(parent, (int)p, (int)(p - 1));        // the run crosses a segment boundary
    for (int i = 0; i < tp.n; i++) {
        if (RUNS && tp.dt[i] == 0 && tp.dy[i] == 0) continue;       // the horizontal tap: runs
        const int64_t tt = t + tp.dt[i];
        const int yy = y + tp.dy[i], xx = x + tp.dx[i];
        if (tt < 0 || tt >= T || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        const int64_t q = tt * plane + (int64_t)yy * W + xx;
        if (!in[q]) continue;
        if (RUNS && tp.dx[i] == 0 && left && in[q - 1]) continue;   // not the start of the two runs' overlap
        if (RUNS && tp.dx[i] != 0 && ((sibling >> i) & 1u) && in[q - tp.dx[i]]) continue;    // (x, y') is set: the sibling tap + row y' run
        ccl_union(parent, (int)p, (int)q);
    }
}

__global__ void __launch_bounds__(256)
k_ccl_flatten(int64_t n, int *__restrict__ parent, uint8_t *__restrict__ isroot) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int p = parent[i];
    if (p < 0) { isroot[i] = 0; return; }
    const int r = ccl_find(parent, (int)i);
    parent[i] = r;
    isroot[i] = r == (int)i;
}

// Numbering the roots in ascending order = an exclusive scan of the root flags, of which only the values AT the roots are
// ever read.  Instead of a device-wide scan that writes one int per voxel (1.9 GB for a 16 x 5424^2 window, 1.3 ms): the
// roots of every 256-voxel block are counted, the block counts are scanned (n / 256 values), and a second pass over the
// flags adds a block-local prefix (ballots + popcounts) and stores the rank of the roots only.
// (one wave per 256-voxel block, a thread = one word of four flags: the counts and prefixes come from four ballots)
__global__ void __launch_bounds__(256)
k_ccl_count_roots(int64_t n, const uint8_t *__restrict__ isroot, int *__restrict__ block_count) {
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), i0 = b * 256 + (threadIdx.x & 63) * 4;
    if (b * 256 >= n) return;                                      // (wave-uniform)
    const uint32_t w = i0 < n ? *(const uint32_t *)(isroot + i0) : 0u;      // (the array is padded to a multiple of 256 bytes)
    int c = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) c += __popcll(__ballot(i0 + j < n && ((w >> (8 * j)) & 0xffu) != 0));
    if ((threadIdx.x & 63) == 0) block_count[b] = c;
}
__global__ void __launch_bounds__(256)
k_ccl_rank_roots(int64_t n, const uint8_t *__restrict__ isroot, const int *__restrict__ block_base, int *__restrict__ rank) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6), i0 = b * 256 + lane * 4;
    if (b * 256 >= n) return;                                      // (wave-uniform)
    const uint32_t w = i0 < n ? *(const uint32_t *)(isroot + i0) : 0u;
    const unsigned long long below = (1ull << lane) - 1ull;
    bool f[4];
    int before = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) { f[j] = i0 + j < n && ((w >> (8 * j)) & 0xffu) != 0; before += __popcll(__ballot(f[j]) & below); }
    if (!(f[0] || f[1] || f[2] || f[3])) return;
    int r = block_base[b] + before;
#pragma unroll
    for (int j = 0; j < 4; j++) if (f[j]) rank[i0 + j] = r++;
}

__global__ void __launch_bounds__(256)
k_ccl_number(int64_t n, const int *__restrict__ parent, const int *__restrict__ rank, int32_t *__restrict__ labels) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = parent[i];
    labels[i] = r < 0 ? 0 : rank[r] + 1;
}
// four voxels per thread (tf_vec4_ok): 16-byte loads and stores, the rank gathers of a quad in flight together
__global__ void __launch_bounds__(256)
k_ccl_number4(int64_t n, const int *__restrict__ parent, const int *__restrict__ rank, int32_t *__restrict__ labels) {
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 3 < n) {
        const int4 r = *(const int4 *)(parent + i);
        const int a = rank[r.x < 0 ? 0 : r.x], b = rank[r.y < 0 ? 0 : r.y], c = rank[r.z < 0 ? 0 : r.z], d = rank[r.w < 0 ? 0 : r.w];
        *(int4 *)(labels + i) = make_int4(r.x < 0 ? 0 : a + 1, r.y < 0 ? 0 : b + 1, r.z < 0 ? 0 : c + 1, r.w < 0 ? 0 : d + 1);
    } else {
        for (int64_t j = i; j < n; j++) { const int r = parent[j]; labels[j] = r < 0 ? 0 : rank[r] + 1; }
    }
}

extern "C" size_t tf_label_workspace_bytes(int64_t T, int64_t H, int64_t W)
{
    if (T <= 0 || H <= 0 || W <= 0) return 0;
    const int64_t n = T * H * W;
    size_t scan = 0;
    const int64_t nblk = (n + 255) / 256;
    (void)tf_exclusive_sum(nullptr, scan, (const int *)nullptr, (int *)nullptr, (size_t)(nblk > 0x7fffffff ? 0x7fffffff : nblk));
    return tf_align_up((size_t)n * 4, 256) * 2 + tf_align_up((size_t)n, 256) + 2 * tf_align_up((size_t)nblk * 4, 256) + tf_align_up(scan, 256) + 4096;
}

// labels: int32 (T, H, W) out; n_labels_host: number of components.  Synchronises the stream.
extern "C" int tf_label(const uint8_t *in, int64_t T, int64_t H, int64_t W, const uint8_t *structure_host,
                        int32_t *labels, int *n_labels_host, void *ws, size_t ws_bytes, void *stream)
{
    TF_REQUIRE(in && labels && structure_host && ws, "tf_label: null pointer");
    TF_REQUIRE(T > 0 && H > 0 && W > 0 && T < 65536 && H < (1 << 15) && W < (1 << 15), "tf_label: bad shape");
    const int64_t n = T * H * W;
    TF_REQUIRE(n < 0x7fffffffll, "tf_label: volume too large for 32-bit parents (use time windows)");
    if (ws_bytes < tf_label_workspace_bytes(T, H, W)) { tf_set_error("tf_label: workspace too small"); return TF_ENOMEM; }
    MorphTaps tp; tp.n = 0;
    for (int p = 0; p < 3; p++) for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) {
        const int i = p * 9 + r * 3 + c;
        TF_REQUIRE((structure_host[i] != 0) == (structure_host[26 - i] != 0), "tf_label: structuring element must be symmetric");
        if (structure_host[i] && i > 13) { tp.dt[tp.n] = (int8_t)(p - 1); tp.dy[tp.n] = (int8_t)(r - 1); tp.dx[tp.n] = (int8_t)(c - 1); tp.n++; }
    }
    hipStream_t s = (hipStream_t)stream;
    TfArena ar(ws, ws_bytes);
    int *parent = ar.take<int>(n), *rank = ar.take<int>(n);
    uint8_t *isroot = ar.take<uint8_t>(n);
    const int64_t nblk = (n + 255) / 256;
    int *block_count = ar.take<int>(nblk), *block_base = ar.take<int>(nblk);
    size_t scan_bytes = 0;
    (void)tf_exclusive_sum(nullptr, scan_bytes, (const int *)nullptr, (int *)nullptr, (size_t)nblk);
    char *scan_tmp = ar.take<char>(scan_bytes ? scan_bytes : 1);
    if (!ar.ok()) { tf_set_error("tf_label: workspace too small"); return TF_ENOMEM; }
    const unsigned nb = (unsigned)((n + 255) / 256);
    dim3 block(64, 4), grid((unsigned)((W + 63) / 64), (unsigned)((H + 3) / 4), (unsigned)T);
    bool runs = false;                                              // the structure holds the horizontal tap (0, 0, +1)
    for (int i = 0; i < tp.n; i++) runs = runs || (tp.dt[i] == 0 && tp.dy[i] == 0 && tp.dx[i] == 1);
    static const bool no_runs_env = getenv("TF_CCL_NO_RUNS") != nullptr;                 // development switch: the plain form (same labels)
    if (no_runs_env) runs = false;
    if (runs) hipLaunchKernelGGL(k_ccl_init_runs, grid, block, 0, s, in, T, (int)H, (int)W, parent);
    else if (tf_vec4_ok({parent}, {in})) hipLaunchKernelGGL(k_ccl_init4, dim3((unsigned)(((n + 3) / 4 + 255) / 256)), dim3(256), 0, s, in, n, parent);
    else hipLaunchKernelGGL(k_ccl_init, dim3(nb), dim3(256), 0, s, in, n, parent);
    unsigned sibling = 0;                                           // bit i: tap i has a horizontal component and (dt, dy, 0) is a tap too
    for (int i = 0; i < tp.n; i++)
        for (int j = 0; j < tp.n && tp.dx[i] != 0; j++)
            if (tp.dx[j] == 0 && tp.dt[j] == tp.dt[i] && tp.dy[j] == tp.dy[i]) sibling |= 1u << i;
    if (tp.n && runs) hipLaunchKernelGGL(k_ccl_union<true>, grid, block, 0, s, in, T, (int)H, (int)W, tp, sibling, parent);
    else if (tp.n) hipLaunchKernelGGL(k_ccl_union<false>, grid, block, 0, s, in, T, (int)H, (int)W, tp, sibling, parent);
    hipLaunchKernelGGL(k_ccl_flatten, dim3(nb), dim3(256), 0, s, n, parent, isroot);
    TF_CHECK_LAUNCH();
    const unsigned nb4 = (unsigned)((nblk + 3) / 4);             // four 256-voxel blocks (waves) per workgroup
    hipLaunchKernelGGL(k_ccl_count_roots, dim3(nb4), dim3(256), 0, s, n, (const uint8_t *)isroot, block_count);
    TF_CHECK_HIP(tf_exclusive_sum(scan_tmp, scan_bytes, (const int *)block_count, block_base, (size_t)nblk, s));
    hipLaunchKernelGGL(k_ccl_rank_roots, dim3(nb4), dim3(256), 0, s, n, (const uint8_t *)isroot, (const int *)block_base, rank);
    if (tf_vec4_ok({parent, labels}, {})) hipLaunchKernelGGL(k_ccl_number4, dim3((unsigned)(((n + 3) / 4 + 255) / 256)), dim3(256), 0, s, n, parent, rank, labels);
    else hipLaunchKernelGGL(k_ccl_number, dim3(nb), dim3(256), 0, s, n, parent, rank, labels);
    TF_CHECK_LAUNCH();
    int last_base = 0, last_count = 0;
    TF_CHECK_HIP(hipMemcpyAsync(&last_base, block_base + nblk - 1, sizeof(int), hipMemcpyDeviceToHost, s));
    TF_CHECK_HIP(hipMemcpyAsync(&last_count, block_count + nblk - 1, sizeof(int), hipMemcpyDeviceToHost, s));
    TF_CHECK_HIP(hipStreamSynchronize(s));
    if (n_labels_host) *n_labels_host = last_base + last_count;
    return TF_OK;
}

// ---- scipy.ndimage.correlate1d with a SYMMETRIC kernel, mode 'reflect' (the pass gaussian_filter is made of) ---------
// ndi.gaussian_filter(field, (0, sigma, sigma)) at tobac_flow/detection.py:65, 137-138, 150 = one such pass per axis with
// sigma > 0, in axis order, each pass rounding to the array's dtype.  SciPy's arithmetic, reproduced bit for bit
// (checked against SciPy 1.15 for float32 / float64, rows shorter than the radius included):
//   every sample converted to double;  tmp = x[c] * w[r];  for j = r .. 1 (OUTERMOST pair first):
//   tmp += (x[c - j] + x[c + j]) * w[r + j];  indices reflected half-sample symmetrically (d c b a | a b c d | d c b a).
// One thread per output; along y / t neighbouring threads stay coalesced in x, along x the taps are L1 hits.
#define TF_C1D_MAX_RADIUS 64
struct Corr1dW { double w[TF_C1D_MAX_RADIUS + 1]; int r; };          // w[j] = weight at distance j from the centre

__device__ __forceinline__ int64_t tf_reflect_index(int64_t i, int64_t n)
{
    if (n == 1) return 0;
    const int64_t p = 2 * n;
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - 1 - i;
}

template <typename TS>
__global__ void __launch_bounds__(256)
k_correlate1d_sym(const TS *__restrict__ in, int64_t T, int H, int W, int axis, Corr1dW kw, TS *__restrict__ out)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    const int64_t t = blockIdx.z;
    if (x >= W || y >= H) return;
    const int64_t plane = (int64_t)H * W;
    const int64_t o = t * plane + (int64_t)y * W + x;
    const int64_t n = axis == 0 ? T : (axis == 1 ? H : W);
    const int64_t c = axis == 0 ? t : (axis == 1 ? y : x);
    const int64_t stride = axis == 0 ? plane : (axis == 1 ? W : 1);
    const TS *line = in + (o - c * stride);                             // element 0 of this thread's line
    double tmp = (double)line[c * stride] * kw.w[0];
    for (int j = kw.r; j >= 1; j--) {
        const double a = (double)line[tf_reflect_index(c - j, n) * stride];
        const double b = (double)line[tf_reflect_index(c + j, n) * stride];
        tmp += (a + b) * kw.w[j];
    }
    out[o] = (TS)tmp;
}

extern "C" int tf_correlate1d_sym(const void *in, int type, int64_t T, int64_t H, int64_t W, int axis,
                                  const double *weights_host, int radius, void *out, void *stream)
{
    TF_REQUIRE(in && out && weights_host, "tf_correlate1d_sym: null pointer");
    TF_REQUIRE(in != out, "tf_correlate1d_sym: in-place operation is not supported");
    TF_REQUIRE(T > 0 && H > 0 && W > 0 && H < (1 << 30) && W < (1 << 30) && T < 65536, "tf_correlate1d_sym: bad shape");
    TF_REQUIRE(axis >= 0 && axis <= 2, "tf_correlate1d_sym: axis must be 0, 1 or 2");
    TF_REQUIRE(radius >= 0 && radius <= TF_C1D_MAX_RADIUS, "tf_correlate1d_sym: radius out of range");
    TF_REQUIRE(type == TF_F32 || type == TF_F64, "tf_correlate1d_sym: type must be TF_F32 or TF_F64");
    Corr1dW kw; kw.r = radius;
    for (int j = 0; j <= radius; j++) {
        // weights_host has 2r+1 entries; symmetric by contract
        TF_REQUIRE(weights_host[radius - j] == weights_host[radius + j], "tf_correlate1d_sym: kernel is not symmetric");
        kw.w[j] = weights_host[radius + j];
    }
    hipStream_t s = (hipStream_t)stream;
    const dim3 block(64, 4, 1), grid((unsigned)((W + 63) / 64), (unsigned)((H + 3) / 4), (unsigned)T);
    if (type == TF_F32) hipLaunchKernelGGL(k_correlate1d_sym<float>, grid, block, 0, s, (const float *)in, T, (int)H, (int)W, axis, kw, (float *)out);
    else hipLaunchKernelGGL(k_correlate1d_sym<double>, grid, block, 0, s, (const double *)in, T, (int)H, (int)W, axis, kw, (double *)out);
    TF_CHECK_LAUNCH();
    return TF_OK;
}

// ---- scipy.ndimage.grey_erosion / grey_dilation with a flat footprint, mode 'reflect' ---------------------------------
// ndi.grey_opening(x, footprint=cross) at tobac_flow/detection.py:106-108 = erosion then dilation.  SciPy's
// min_or_max_filter visits the footprint's true cells in C order, starts from the FIRST cell's value and replaces it
// only on a strict comparison (v < tmp / v > tmp): a NaN in the first cell sticks, a NaN elsewhere is ignored.  That
// order is kept here (bit-exact incl. NaN placement, checked against SciPy 1.15).  Restricted to point-symmetric
// footprints of a 3x3x3 box: for those SciPy's mirrored dilation footprint visits the same cells in the same order.
template <typename TS>
__global__ void __launch_bounds__(256)
k_grey_morph(const TS *__restrict__ in, int64_t T, int H, int W, MorphTaps tp, int op, TS *__restrict__ out)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    const int64_t t = blockIdx.z;
    if (x >= W || y >= H) return;
    const int64_t plane = (int64_t)H * W;
    TS tmp = 0;
    for (int i = 0; i < tp.n; i++) {
        const int64_t tt = tf_reflect_index(t + tp.dt[i], T);
        const int64_t yy = tf_reflect_index(y + tp.dy[i], H), xx = tf_reflect_index(x + tp.dx[i], W);
        const TS v = in[tt * plane + yy * W + xx];
        if (i == 0) tmp = v;
        else if (op == 0) { if (v < tmp) tmp = v; }
        else { if (v > tmp) tmp = v; }
    }
    out[t * plane + (int64_t)y * W + x] = tmp;
}

extern "C" int tf_grey_morph(const void *in, int type, int64_t T, int64_t H, int64_t W, const uint8_t *footprint_host,
                             int op, void *out, void *stream)
{
    TF_REQUIRE(in && out && footprint_host, "tf_grey_morph: null pointer");
    TF_REQUIRE(in != out, "tf_grey_morph: in-place operation is not supported");
    TF_REQUIRE(T > 0 && H > 0 && W > 0 && H < (1 << 30) && W < (1 << 30) && T < 65536, "tf_grey_morph: bad shape");
    TF_REQUIRE(op == 0 || op == 1, "tf_grey_morph: op must be 0 (erosion) or 1 (dilation)");
    TF_REQUIRE(type == TF_F32 || type == TF_F64, "tf_grey_morph: type must be TF_F32 or TF_F64");
    MorphTaps tp; tp.n = 0;
    for (int i = 0; i < 27; i++) {
        TF_REQUIRE((footprint_host[i] != 0) == (footprint_host[26 - i] != 0), "tf_grey_morph: footprint must be point-symmetric");
        if (footprint_host[i]) { tp.dt[tp.n] = (int8_t)(i / 9 - 1); tp.dy[tp.n] = (int8_t)((i / 3) % 3 - 1); tp.dx[tp.n] = (int8_t)(i % 3 - 1); tp.n++; }
    }
    TF_REQUIRE(tp.n > 0, "tf_grey_morph: empty footprint");
    hipStream_t s = (hipStream_t)stream;
    const dim3 block(64, 4, 1), grid((unsigned)((W + 63) / 64), (unsigned)((H + 3) / 4), (unsigned)T);
    if (type == TF_F32) hipLaunchKernelGGL(k_grey_morph<float>, grid, block, 0, s, (const float *)in, T, (int)H, (int)W, tp, op, (float *)out);
    else hipLaunchKernelGGL(k_grey_morph<double>, grid, block, 0, s, (const double *)in, T, (int)H, (int)W, tp, op, (double *)out);
    TF_CHECK_LAUNCH();
    return TF_OK;
}
