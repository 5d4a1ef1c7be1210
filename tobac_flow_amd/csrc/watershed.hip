// Semi-Lagrangian marker-controlled watershed on gfx950 -- wavefront-parallel priority flood.
//
// Replaces /root/reference/tobac_flow/watershed.py:17-168 and the sequential heap flood
// /root/reference/tobac_flow/_watershed.pyx:222-344 (compactness = 0, wsl = False).
//
// The reference pops pixels in (value, age) order from one binary heap and labels a pixel when it
// is first pushed (:330-337).  That pop order has a closed form ("Watershed"):
//   K2(n) = (l, g)   l = flood level at which n pops = min over directed paths from a marker of
//                        the max value on the path;  g = FIFO generation inside that level
//   chain(n) = [K2(n), K2(entry(n)), K2(entry(entry(n))), ..., marker push index]
// where entry(n) is the lower-level pixel that pushed the first pixel of n's same-level run.
// The first in-neighbour to pop -- the one whose label n takes -- is the in-neighbour with the
// lexicographically smallest chain.  Every component of the chain is the fixpoint of a MONOTONE
// min-relaxation over the directed neighbour graph (flow-displaced in t), so it is computed with
// chaotic 64-bit atomicMin sweeps, one phase per chain level:
//   phase A   K2 and M1(n) = min K2 over in-neighbours
//   phase k   C_k(n) = k-th chain element (k = 1 .. depth-1), candidates must match levels < k
//   phase R   root marker index among fully matching candidates; label(n) = markers[R(n)]
// Ties between equal-valued MARKERS (age 0 in the reference, heap-internal order there) are
// broken by push order = raster index.
//
// Fast path: phase A, then phase R with candidates matched on K2 only.  If afterwards every
// candidate edge p -> n joins equal labels, the labelling does not depend on any tie-break and is
// final (induction over strata); otherwise the chain phases run.  Tie-free fields and the
// detect_anvils plateau structure normally finish on the fast path.
//
// Data layout: the volume itself only carries cid int32[N] (compact id of floodable pixels,
// -2-id for boundary markers, -1 for everything else) and cls u8[N].  All keys live in COMPACT
// arrays over the R = (#floodable + #boundary markers) relevant pixels, ids in TILE order (WS_TILE, below; raster order in the raveled form):
//   pix u64[R] raster index (top bit: marker), val u32[R] ordered field key, nbr int32[R][n_nbr] compact ids of the
//   floodable out-neighbours (flow displacement already applied), K2, M1, C_1.., Rt u64[R].
// Sweeps are frontier driven: two ping-pong queues of compact ids hold the pixels whose key just
// changed; a sweep costs work proportional to the frontier, not to the volume.
#include "tf_common.h"
#include <string.h>
#include <time.h>
#include <stdio.h>
#include <stdlib.h>
#include "tf_prim.h"
#include <new>

typedef unsigned long long u64;
#define WS_INF 0xFFFFFFFFFFFFFFFFull
#define WS_MARKER_BIT 0x8000000000000000ull
#define WS_MAX_NBR 26
#define WS_MAX_DEPTH TF_WS_MAX_DEPTH
#define WS_BATCH 32
#define WS_K2(c, i) ((c).KM[2 * (int64_t)(i)])
#define WS_M1(c, i) ((c).KM[2 * (int64_t)(i) + 1])

struct WsGeom {
    int64_t T; int H, W; int64_t plane;
    int n_nbr;
    int8_t dt[WS_MAX_NBR], dy[WS_MAX_NBR], dx[WS_MAX_NBR];
    int n_tx;              // tiles per row of tiles (tiled compact ids, below)
};

// COMPACT IDS IN TILE ORDER (round 3).  The relevant pixels are numbered by an exclusive scan of their flags.  In raster
// order a pixel's y neighbours lie a row's worth of ids away and its t neighbours a frame's worth (megabytes in the key
// arrays): the floodable set of a detect_anvils field is made of bands a few pixels wide, so a sweep touched one or two
// useful entries per cache line it pulled -- L2 hit rate 35 %, ~1 KB fetched from HBM per relevant pixel and phase.
// The scan therefore runs over the flags in the order (tile row, tile column, t, y in tile, x in tile) with 16 x 16
// pixel tiles through ALL frames: the six (or 26) neighbours of a pixel, flow displacement included, then carry ids a few
// hundred apart.  Nothing depends on the order of the ids: the tie-break of last resort is the RASTER index of the root
// marker, which the root phase carries as such (default mode) or as the reference's pop rank (TF_WS_REFERENCE_ORDER).
#ifndef WS_TILE
#define WS_TILE 16
#endif
__device__ __forceinline__ int64_t ws_vpos(const WsGeom &g, int64_t t, int y, int x) {
    const int64_t tile = (int64_t)(y / WS_TILE) * g.n_tx + x / WS_TILE;
    return ((tile * g.T + t) * WS_TILE + (y % WS_TILE)) * WS_TILE + (x % WS_TILE);
}
static int64_t ws_virtual_voxels(int64_t T, int64_t H, int64_t W) {
    return T * ((H + WS_TILE - 1) / WS_TILE * WS_TILE) * ((W + WS_TILE - 1) / WS_TILE * WS_TILE);
}

struct WsC {               // compact arrays
    int64_t R; int n_nbr;
    const u64 *pix; const unsigned *val; const int *nbr;
    u64 *KM;               // K2 and M1 of a pixel side by side ({K2, M1}[R], 16-byte aligned): phase A reads both with one load
    u64 *C[WS_MAX_DEPTH], *Rt;
    int *Llo, *Lhi;        // smallest / largest label among the roots of all fully matching candidates (root phase)
    const u64 *emask;      // per pixel p, bit i: out-edge i is a CANDIDATE edge (K2[p] == M1[nbr i]); bit 32 + i: nbr i is an entry
};

__device__ __forceinline__ unsigned ws_ordkey(float v) {
    v = v + 0.0f;                                    // -0.0 -> +0.0 (they compare equal in the reference)
    unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ u64 ws_load(const u64 *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Ordering by data dependency.  The in-queue flag of a pixel must be cleared BEFORE its key is read (otherwise a
// decrease that lands between the read and the clear is never propagated).  Both are relaxed agent-scope accesses to
// different addresses, which the memory system may perform out of order; so the key's address takes the exchange's
// return value (0 or 1) as a term the compiler cannot fold: ws_after(x) is 0 for every value the flag can hold.
__device__ __forceinline__ int ws_after(int was) { asm volatile("" : "+v"(was)); return was >> 2; }
__device__ __forceinline__ int ws_round_flow(float f) {
    // np.round(flow).astype(int32), watershed.py:121-141 (half to even); NaN -> 0
    return (f == f) ? __float2int_rn(f) : 0;
}

// neighbour of p = (t, y, x) through slot i, or -1 (watershed.pyx:310-313 without the padding)
__device__ __forceinline__ int64_t ws_neighbour(const WsGeom &g, int64_t t, int y, int x, int fx, int fy, int bx, int by, int i) {
    const int dt = g.dt[i];
    int yy = y + g.dy[i], xx = x + g.dx[i];
    if (dt == 1) { xx += fx; yy += fy; }
    else if (dt == -1) { xx += bx; yy += by; }
    const int64_t tt = t + dt;
    if (tt < 0 || tt >= g.T || (unsigned)yy >= (unsigned)g.H || (unsigned)xx >= (unsigned)g.W) return -1;
    return tt * g.plane + (int64_t)yy * g.W + xx;
}

// cls: 0 = not floodable, 1 = floodable (mask set, no marker), 2 = marker
__global__ void __launch_bounds__(256)
k_ws_classify(const int32_t *__restrict__ markers, const int8_t *__restrict__ mask, int64_t n, uint8_t *__restrict__ cls)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t m = markers[i];
    cls[i] = m != 0 ? 2 : ((mask ? mask[i] != 0 : true) ? 1 : 0);
}
// four voxels per thread (markers 16-byte aligned, mask and cls 4-byte aligned): 16-byte loads, one word store
__global__ void __launch_bounds__(256)
k_ws_classify4(const int32_t *__restrict__ markers, const int8_t *__restrict__ mask, int64_t n, uint8_t *__restrict__ cls)
{
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    if (i + 3 < n) {
        const int4 m = *(const int4 *)(markers + i);
        const uint32_t k = mask ? *(const uint32_t *)(mask + i) : 0xffffffffu;
        const int mm[4] = {m.x, m.y, m.z, m.w};
        uint32_t out = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) out |= (uint32_t)(mm[j] != 0 ? 2 : (((k >> (8 * j)) & 0xffu) ? 1 : 0)) << (8 * j);
        *(uint32_t *)(cls + i) = out;
    } else {
        for (int64_t j = i; j < n; j++) cls[j] = markers[j] != 0 ? 2 : ((mask ? mask[j] != 0 : true) ? 1 : 0);
    }
}

// flag = 1 for floodable pixels and for markers with at least one floodable out-neighbour.
// NN = the neighbour count as a compile-time constant (6 / 18 / 26: connectivity 1 / 2 / 3), 0 = any count.  With a run-time
// count every g.dt[i] / dy / dx is a scalar load from the kernel arguments followed by a wait and a branch (the generic form
// compiled to 97 branches and 49 scalar loads); unrolled, the table sits in scalar registers and a marker's NN neighbour
// classes are in flight together.
template <int NN>
__global__ void __launch_bounds__(256)
k_ws_relevant(const uint8_t *__restrict__ cls, const float *__restrict__ fwd, const float *__restrict__ bwd, WsGeom g,
              uint8_t *__restrict__ flag)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    const int64_t t = blockIdx.z;
    if (x >= g.W || y >= g.H) return;
    const int64_t p = t * g.plane + (int64_t)y * g.W + x;
    const uint8_t c = cls[p];
    uint8_t f = c == 1;
    if (c == 2) {
        const float2 fw = ((const float2 *)fwd)[p], bw = ((const float2 *)bwd)[p];
        const int fx = ws_round_flow(fw.x), fy = ws_round_flow(fw.y), bx = ws_round_flow(bw.x), by = ws_round_flow(bw.y);
        if (NN > 0) {
            uint8_t v[NN > 0 ? NN : 1];
#pragma unroll
            for (int j = 0; j < NN; j++) {               // branch-free: a missing neighbour reads p itself and is discarded
                const int64_t n = ws_neighbour(g, t, y, x, fx, fy, bx, by, j);
                const uint8_t vn = cls[n >= 0 ? n : p];
                v[j] = n >= 0 ? vn : 0;
            }
#pragma unroll
            for (int j = 0; j < NN; j++) f |= v[j] == 1;
        } else {
            // eight neighbour classes in flight at a time (a marker inside a marker region has to look at all of them)
            for (int i0 = 0; i0 < g.n_nbr && !f; i0 += 8) {
                uint8_t v[8];
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int64_t n = i0 + j < g.n_nbr ? ws_neighbour(g, t, y, x, fx, fy, bx, by, i0 + j) : -1;
                    v[j] = n >= 0 ? cls[n] : 0;
                }
#pragma unroll
                for (int j = 0; j < 8; j++) f |= v[j] == 1;
            }
        }
    }
    flag[ws_vpos(g, t, y, x)] = f;               // (the padding positions of edge tiles were zeroed by the caller)
}

// Connectivity 1 (the six face neighbours), four pixels per thread (W % 4 == 0, cls / flag 4-byte and the flows 16-byte
// aligned): the classes of a pixel quad, of the quads above and below and of the two edge bytes decide the four in-plane
// neighbours without further loads; only markers that found no floodable in-plane neighbour read their two flow-displaced
// t -+ 1 neighbours.  17 loads per quad at most instead of 4 x 10.
__global__ void __launch_bounds__(256)
k_ws_relevant6x4(const uint8_t *__restrict__ cls, const float *__restrict__ fwd, const float *__restrict__ bwd, WsGeom g,
                 uint8_t *__restrict__ flag)
{
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4, y = blockIdx.y * 4 + threadIdx.y;
    const int64_t t = blockIdx.z;
    if (x0 >= g.W || y >= g.H) return;
    const int64_t p0 = t * g.plane + (int64_t)y * g.W + x0;
    const uint32_t c = *(const uint32_t *)(cls + p0);
    const uint32_t is1 = c & 0x01010101u & ~((c >> 1) & 0x01010101u);          // bytes equal to 1 (classes are 0 / 1 / 2)
    const uint32_t is2 = (c >> 1) & 0x01010101u;                                // bytes equal to 2
    uint32_t rel = is1;
    if (is2) {
        // floodable (class 1) flags of the in-plane neighbours, one byte per pixel of the quad
        const uint32_t up = y > 0 ? *(const uint32_t *)(cls + p0 - g.W) : 0u, dn = y + 1 < g.H ? *(const uint32_t *)(cls + p0 + g.W) : 0u;
        const uint32_t lb = x0 > 0 ? cls[p0 - 1] : 0u, rb = x0 + 4 < g.W ? cls[p0 + 4] : 0u;
        const uint32_t one = 0x01010101u;
        auto ones = [&](uint32_t w) { return w & one & ~((w >> 1) & one); };
        uint32_t hit = ones(up) | ones(dn) | ones((c << 8) | lb) | ones((c >> 8) | (rb << 24));
        uint32_t need = is2 & ~hit;                                             // markers still undecided: look at t -+ 1
        rel |= is2 & hit;
        if (need) {
            const float4 fa = *(const float4 *)(fwd + 2 * p0), fb = *(const float4 *)(fwd + 2 * p0 + 4);
            const float4 ba = *(const float4 *)(bwd + 2 * p0), bb = *(const float4 *)(bwd + 2 * p0 + 4);
            const float fx[4] = {fa.x, fa.z, fb.x, fb.z}, fy[4] = {fa.y, fa.w, fb.y, fb.w};
            const float bx[4] = {ba.x, ba.z, bb.x, bb.z}, by[4] = {ba.y, ba.w, bb.y, bb.w};
            uint8_t vn[4], vp[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                // missing neighbours (outside the volume) and decided pixels read the pixel itself and are discarded
                const bool want = (need >> (8 * j)) & 1u;
                const int xn = x0 + j + ws_round_flow(fx[j]), yn = y + ws_round_flow(fy[j]);
                const int xp = x0 + j + ws_round_flow(bx[j]), yp = y + ws_round_flow(by[j]);
                const bool okn = want && t + 1 < g.T && (unsigned)xn < (unsigned)g.W && (unsigned)yn < (unsigned)g.H;
                const bool okp = want && t > 0 && (unsigned)xp < (unsigned)g.W && (unsigned)yp < (unsigned)g.H;
                const uint8_t a = cls[okn ? (t + 1) * g.plane + (int64_t)yn * g.W + xn : p0 + j];
                const uint8_t b = cls[okp ? (t - 1) * g.plane + (int64_t)yp * g.W + xp : p0 + j];
                vn[j] = okn ? a : 0; vp[j] = okp ? b : 0;
            }
#pragma unroll
            for (int j = 0; j < 4; j++) rel |= (uint32_t)(vn[j] == 1 || vp[j] == 1) << (8 * j);
        }
    }
    *(uint32_t *)(flag + ws_vpos(g, t, y, x0)) = rel;        // (x0 % 4 == 0: the quad lies in one tile row; padding positions zeroed by the caller)
}

// the same numbering for the (t, y, x) entry points, whose scan runs in tile order
__global__ void __launch_bounds__(256)
k_ws_cid_tiled(const uint8_t *__restrict__ cls, const uint8_t *__restrict__ flag, const int *__restrict__ scan, WsGeom g,
               int *__restrict__ cid)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    const int64_t t = blockIdx.z;
    if (x >= g.W || y >= g.H) return;
    const int64_t p = t * g.plane + (int64_t)y * g.W + x, k = ws_vpos(g, t, y, x);
    int c = -1;
    if (flag[k]) c = cls[p] == 1 ? scan[k] : -2 - scan[k];
    cid[p] = c;
}

// The same ids WITHOUT a device-wide scan over one int per voxel (1.9 GB written and read back per 16 x 5424^2 window): a
// scan position block of 256 = one 16 x 16 tile of one frame.  k_ws_count_tiles counts the flags of every tile-frame, the
// counts are scanned (NV / 256 values, 64-bit: the total is checked against 2^30 afterwards), and k_ws_cid_tiles -- one
// wave per tile-frame -- adds the prefix inside the tile (ballots + popcounts).
// One WAVE per tile-frame, a thread = four consecutive pixels of a tile row (one word of flags, one int4 of ids): the
// prefix inside the tile comes from four ballots, no LDS, no barrier.
__global__ void __launch_bounds__(256)
k_ws_count_tiles(const uint8_t *__restrict__ flag, int64_t n_tiles, long long *__restrict__ count)
{
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n_tiles) return;                                      // (wave-uniform)
    const int lane = threadIdx.x & 63;
    const uint32_t w = *(const uint32_t *)(flag + b * 256 + lane * 4);
    int c = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) c += __popcll(__ballot(((w >> (8 * j)) & 0xffu) != 0));
    if (lane == 0) count[b] = c;
}
__global__ void __launch_bounds__(256)
k_ws_cid_tiles(const uint8_t *__restrict__ cls, const uint8_t *__restrict__ flag, const long long *__restrict__ base, int64_t n_tiles,
               WsGeom g, int *__restrict__ cid)
{
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= n_tiles) return;                                      // (wave-uniform)
    const int lane = threadIdx.x & 63;
    const int64_t tile = b / g.T, t = b - tile * g.T;
    const int ty = (int)(tile / g.n_tx), tx = (int)(tile - (int64_t)ty * g.n_tx);
    const int y = ty * WS_TILE + lane / 4, x0 = tx * WS_TILE + (lane & 3) * 4;
    const uint32_t w = *(const uint32_t *)(flag + b * 256 + lane * 4);      // (padding positions of edge tiles hold 0)
    const unsigned long long below = (1ull << lane) - 1ull;
    bool f[4]; unsigned long long m[4];
    int before = 0;                                                // flags of the lanes below, all four byte positions
#pragma unroll
    for (int j = 0; j < 4; j++) { f[j] = ((w >> (8 * j)) & 0xffu) != 0; m[j] = __ballot(f[j]); before += __popcll(m[j] & below); }
    if (y >= g.H || x0 >= g.W) return;
    const int64_t p0 = t * g.plane + (int64_t)y * g.W + x0;
    const int id0 = (int)base[b] + before;
    int c[4], run = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        c[j] = -1;
        if (f[j]) { const int id = id0 + run; c[j] = cls[p0 + j] == 1 ? id : -2 - id; run++; }
    }
    if (x0 + 3 < g.W && (((uintptr_t)(cid + p0)) & 15) == 0) *(int4 *)(cid + p0) = make_int4(c[0], c[1], c[2], c[3]);
    else {
#pragma unroll
        for (int j = 0; j < 4; j++) if (x0 + j < g.W) cid[p0 + j] = c[j];
    }
}
static_assert(WS_TILE * WS_TILE == 256, "one scan block = one tile of one frame");

// cid from the exclusive scan of flag: id (floodable), -2-id (boundary marker), -1 otherwise
__global__ void __launch_bounds__(256)
k_ws_cid(const uint8_t *__restrict__ cls, const uint8_t *__restrict__ flag, const int *__restrict__ scan, int64_t n,
         int *__restrict__ cid)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int c = -1;
    if (flag[i]) c = cls[i] == 1 ? scan[i] : -2 - scan[i];
    cid[i] = c;
}

template <int NN>
__global__ void __launch_bounds__(256)
k_ws_compact(const float *__restrict__ field, const float *__restrict__ fwd, const float *__restrict__ bwd,
             const int *__restrict__ cid, WsGeom g, u64 *__restrict__ pix, unsigned *__restrict__ val,
             int *__restrict__ nbr, u64 *__restrict__ KM, int *__restrict__ nan_flag)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    const int64_t t = blockIdx.z;
    if (x >= g.W || y >= g.H) return;
    const int64_t p = t * g.plane + (int64_t)y * g.W + x;
    const int c = cid[p];
    if (c == -1) return;
    const bool marker = c < 0;
    const int64_t id = marker ? -2 - c : c;
    const float2 fw = ((const float2 *)fwd)[p], bw = ((const float2 *)bwd)[p];
    const int fx = ws_round_flow(fw.x), fy = ws_round_flow(fw.y), bx = ws_round_flow(bw.x), by = ws_round_flow(bw.y);
    const float fv = field[p];
    if (fv != fv) *nan_flag = 1;                 // (every writer stores the same value)
    const unsigned v = ws_ordkey(fv);
    pix[id] = (u64)p | (marker ? WS_MARKER_BIT : 0ull);
    val[id] = v;
    if (NN > 0) {                                // (see k_ws_relevant: table in scalar registers, all loads in flight together)
        int cn[NN > 0 ? NN : 1];
#pragma unroll
        for (int i = 0; i < NN; i++) {
            const int64_t n = ws_neighbour(g, t, y, x, fx, fy, bx, by, i);
            const int cv = cid[n >= 0 ? n : p];
            cn[i] = n >= 0 ? cv : -1;
        }
#pragma unroll
        for (int i = 0; i < NN; i++) nbr[id * NN + i] = cn[i] < 0 ? -1 : cn[i];
    } else {
        for (int i = 0; i < g.n_nbr; i++) {
            const int64_t n = ws_neighbour(g, t, y, x, fx, fy, bx, by, i);
            int cn = -1;
            if (n >= 0) { cn = cid[n]; if (cn < 0) cn = -1; }
            nbr[id * g.n_nbr + i] = cn;
        }
    }
    *(ulonglong2 *)&KM[2 * id] = make_ulonglong2(marker ? ((u64)v << 32) : WS_INF, WS_INF);
}

// ---- the reference's own raveled form (tf_watershed_raveled) -------------------------------------------------------
// neighbour i of flat index p:  p + structure[i] + fwd_loc[i] * fwd_off[p] + bwd_loc[i] * bwd_off[p]
// (_watershed.pyx:310-313).  The reference relies on a zero-masked padding ring to stay inside the arrays; here every
// neighbour index is range-checked as well.
struct WsRavel {
    int64_t n; int n_nbr;
    int64_t structure[WS_MAX_NBR]; int32_t floc[WS_MAX_NBR], bloc[WS_MAX_NBR];
    const int32_t *foff, *boff;
};
__device__ __forceinline__ int64_t wsr_neighbour(const WsRavel &g, int64_t p, int fo, int bo, int i) {
    const int64_t n = p + g.structure[i] + (int64_t)g.floc[i] * fo + (int64_t)g.bloc[i] * bo;
    return (n >= 0 && n < g.n) ? n : -1;
}

// seeds are the entries of marker_locations (ascending = np.flatnonzero order, checked): flag them
__global__ void __launch_bounds__(256)
k_wsr_seeds(const int64_t *__restrict__ locs, int64_t n_locs, int64_t n, uint8_t *__restrict__ seed, int *__restrict__ bad)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_locs) return;
    const int64_t p = locs[i];
    if (p < 0 || p >= n || (i > 0 && locs[i - 1] >= p)) { *bad = 1; return; }
    seed[p] = 1;
}

// cls: 2 seed; 1 floodable (mask set, output 0); 0 anything else (masked out, or pre-labelled without being a seed)
__global__ void __launch_bounds__(256)
k_wsr_classify(const uint8_t *__restrict__ seed, const int32_t *__restrict__ output, const int8_t *__restrict__ mask, int64_t n,
               uint8_t *__restrict__ cls, int *__restrict__ bad)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool sd = seed[i] != 0;
    if (sd && output[i] == 0) *bad = 2;          // "output must already contain nonzero entries at all the seed locations"
    cls[i] = sd ? 2 : ((mask[i] != 0 && output[i] == 0) ? 1 : 0);
}

__global__ void __launch_bounds__(256)
k_wsr_relevant(const uint8_t *__restrict__ cls, WsRavel g, uint8_t *__restrict__ flag)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= g.n) return;
    const uint8_t c = cls[p];
    uint8_t f = c == 1;
    if (c == 2) {
        const int fo = g.foff[p], bo = g.boff[p];
        for (int i = 0; i < g.n_nbr && !f; i++) { const int64_t n = wsr_neighbour(g, p, fo, bo, i); f |= n >= 0 && cls[n] == 1; }
    }
    flag[p] = f;
}

__global__ void __launch_bounds__(256)
k_wsr_compact(const float *__restrict__ image, const int *__restrict__ cid, WsRavel g, u64 *__restrict__ pix,
              unsigned *__restrict__ val, int *__restrict__ nbr, u64 *__restrict__ KM, int *__restrict__ nan_flag)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= g.n) return;
    const int c = cid[p];
    if (c == -1) return;
    const bool marker = c < 0;
    const int64_t id = marker ? -2 - c : c;
    const int fo = g.foff[p], bo = g.boff[p];
    const float fv = image[p];
    if (fv != fv) *nan_flag = 1;
    const unsigned v = ws_ordkey(fv);
    pix[id] = (u64)p | (marker ? WS_MARKER_BIT : 0ull);
    val[id] = v;
    for (int i = 0; i < g.n_nbr; i++) {
        const int64_t n = wsr_neighbour(g, p, fo, bo, i);
        int cn = -1;
        if (n >= 0) { cn = cid[n]; if (cn < 0) cn = -1; }
        nbr[id * g.n_nbr + i] = cn;
    }
    *(ulonglong2 *)&KM[2 * id] = make_ulonglong2(marker ? ((u64)v << 32) : WS_INF, WS_INF);
}

// in place: output[i] = label of its root seed for every flooded pixel (seeds and everything else untouched)
__global__ void __launch_bounds__(256)
k_wsr_labels(const int *__restrict__ cid, const u64 *__restrict__ Rt, const u64 *__restrict__ pix, int ranked, int32_t *output, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = cid[i];
    if (c >= 0 && output[i] == 0) { const u64 r = Rt[c]; if (r != WS_INF) output[i] = output[ranked ? (pix[r & 0xFFFFFFFFull] & ~WS_MARKER_BIT) : r]; }
}

// chain arrays of a marker: C_k = 0 for every k; root key: rank == nullptr (default): the marker's RASTER index (= the
// reference's marker_locations order); otherwise (pop rank the reference's heap gives the marker << 32) | compact id
// (ws_reference_ranks below).  The label kernels are told which of the two a root key is (`ranked`).
__global__ void __launch_bounds__(256)
k_ws_init_level(const u64 *__restrict__ pix, u64 *__restrict__ dst, int64_t R, int is_root, const int *__restrict__ rank)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const u64 px = pix[i];
    const bool marker = (px & WS_MARKER_BIT) != 0ull;
    const u64 root = rank ? (((u64)(unsigned)rank[i] << 32) | (u64)(unsigned)i) : (px & ~WS_MARKER_BIT);
    dst[i] = marker ? (is_root ? root : 0ull) : WS_INF;
}

// root phase start: a marker's label set is its own label, everything else starts empty
__global__ void __launch_bounds__(256)
k_ws_init_labelset(const u64 *__restrict__ pix, const int32_t *__restrict__ markers, int *__restrict__ lo,
                   int *__restrict__ hi, int64_t R)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    const u64 px = pix[i];
    const bool marker = (px & WS_MARKER_BIT) != 0ull;
    const int32_t l = marker ? markers[px & ~WS_MARKER_BIT] : 0;
    lo[i] = marker ? l : 0x7fffffff;
    hi[i] = marker ? l : (int)0x80000000;
}

// ---- frontier queues ---------------------------------------------------------------------------
// Two modes.  WITHOUT in-queue flags (inq == nullptr, the default): a pixel is appended whenever a relaxation STRICTLY
// lowers its key -- no lost update by construction (the append follows the atomic that lowered the key), and a pixel
// lowered twice in one sweep simply sits in the queue twice (the second visit finds nothing to do).  That removes
// ~4 of the ~10 returning atomics a processed pixel costs, which is what bounds the sweeps.  The queue length then has
// no a-priori bound: if a sweep overflows its queue the host re-seeds the phase with a full scan (every finite key is
// pushed again; the fixpoint is the same).  WITH flags (TF_WS_DEDUPE=1, the round-1 scheme):
// A sweep processes the queue of pixels whose key changed (qin) and appends every pixel whose key it
// lowers to qout.  inq[n] = 1 while n sits in a queue and has not been processed since: the flag is
// cleared BEFORE the pixel's key is read, so a later decrease always re-queues it (no lost update); a
// pixel can appear at most twice per queue, hence the 2R capacity.  Appends are aggregated per wave
// (one atomicAdd per 64 lanes).
// LOCAL ROUNDS.  A sweep used to advance every front by exactly one pixel: a flood that has to cross a few hundred
// pixels of plateau costs that many launches, each a few microseconds of work behind ~20 us of launch latency.  A
// workgroup now keeps the pixels it has just lowered in LDS and relaxes THEM as well, for up to `rounds` further
// rounds, before it hands what is left to the global queue: its piece of the front moves several pixels per launch.
// Measured (12 x 5424^2, 17.6 M relevant pixels): 576 -> 96 launches per call with 8 rounds, 128 with 4 -- and the same
// 32 ms either way.  The sweeps are NOT launch-bound: a call issues ~10 returning 64-bit atomics per processed pixel
// (K2 and M1 minima, in-queue flags), ~2 x 10^8 per phase, and runs at the chip's rate for scattered atomics
// (~2 x 10^10 / s, MI355X_MICROARCH.md "Global float atomics": 64 lanes in 64 rows).  Fewer launches is what is kept.
// Chaotic relaxation of a monotone system reaches the same fixpoint in any order, and the in-queue flags work as
// before (a staged pixel has its flag set; whoever processes it clears the flag before reading the key).
#ifndef WS_LDS_CAP
#define WS_LDS_CAP 4096
#endif
struct WsStage { int cnt[2]; int base; int buf[2][WS_LDS_CAP]; };

// stage an append in LDS buffer w (one LDS atomic per appended id); a full buffer spills straight to the global queue
__device__ __forceinline__ void ws_stage(WsStage &st, int w, bool enq, int n, int *__restrict__ qout, int *__restrict__ cnt_out, int qcap) {
    if (enq) {
        const int pos = atomicAdd(&st.cnt[w], 1);
        if (pos < WS_LDS_CAP) st.buf[w][pos] = n;
        else { const int g = atomicAdd(cnt_out, 1); if (g < qcap) qout[g] = n; }
    }
}
// flush buffer w with ONE global atomicAdd; must be reached by every thread
__device__ __forceinline__ void ws_flush(WsStage &st, int w, int *__restrict__ qout, int *__restrict__ cnt_out, int qcap) {
    __syncthreads();
    const int n = min(st.cnt[w], WS_LDS_CAP);
    if (n > 0) {
        if (threadIdx.x == 0) st.base = atomicAdd(cnt_out, n);
        __syncthreads();
        const int base = st.base;
        for (int i = threadIdx.x; i < n; i += blockDim.x) { const int pos = base + i; if (pos < qcap) qout[pos] = st.buf[w][i]; }
        __syncthreads();
        if (threadIdx.x == 0) st.cnt[w] = 0;
    }
    __syncthreads();
}

// ---- phase A: K2 and M1 ------------------------------------------------------------------------
// Memory-level parallelism.  A queue entry used to walk its out-edges one at a time: neighbour id -> its value
// and keys -> atomics, about twenty DEPENDENT round trips per entry, and a large sweep ran at the latency of that
// chain (0.45 ns per entry = a few hundred GB/s).  The edges of one pixel are independent, so they are now handled
// WS_NB at a time in three waves of independent accesses: all neighbour ids (issued together with the in-queue
// exchange), then all neighbour values / keys, then all atomics.  Same operations, same decisions on the values the
// atomics return; the plain reads were already only pre-filters (keys only decrease, a stale value is larger).
#define WS_NB 8

// one queue entry of phase A: pop p, relax its out-edges WS_NB at a time, stage the pixels whose key it lowered
// NN = 6 (round 5): the six face neighbours as a compile-time count -- one trip of SIX slots instead of eight with two dead
// ones (a dead slot still cost its registers and two loads of element 0 per entry): k_ws_sweep_a 80 -> 64 VGPRs; NN = 0:
// any neighbour count, eight slots per trip
template <int NN>
__device__ __forceinline__ void ws_entry_a(const WsC &c, WsStage &st, int w, bool act, int p, int *__restrict__ inq,
                                           int *__restrict__ qout, int *__restrict__ cnt_out, int qcap)
{
    constexpr int NB = NN ? NN : WS_NB;
    const int n_nbr = NN ? NN : c.n_nbr;
    const int *np = c.nbr + (int64_t)p * n_nbr;
    u64 kp = WS_INF;
    for (int s0 = 0; s0 < n_nbr; s0 += NB) {
        int n[NB];
#pragma unroll
        for (int j = 0; j < NB; j++) n[j] = (act && s0 + j < n_nbr) ? np[s0 + j] : -1;
        if (s0 == 0 && act) {
            // relaxed L2 atomics only: the key is loaded after the exchange has returned (the flag is cleared BEFORE
            // the key is read, so a later decrease re-queues the pixel); the id loads above are already in flight
            const int dep = inq ? ws_after(atomicExch(&inq[p], 0)) : 0;
            kp = ws_load(&WS_K2(c, p) + 2 * dep);
        }
        const u64 lp = kp >> 32;
        u64 vn[NB], m1[NB], k2[NB];
#pragma unroll
        for (int j = 0; j < NB; j++) {
            const int q = n[j] >= 0 ? n[j] : 0;
            vn[j] = c.val[q];
            const ulonglong2 km = *(const ulonglong2 *)&WS_K2(c, q);
            k2[j] = km.x; m1[j] = km.y;
        }
        u64 cand[NB], old[NB];
#pragma unroll
        for (int j = 0; j < NB; j++) {
            cand[j] = vn[j] > lp ? ((vn[j] << 32) | 1ull) : (vn[j] == lp ? kp + 1ull : kp);
            old[j] = 0ull;                                             // "no improvement"
            if (n[j] >= 0) {
                // keys only decrease, so a (possibly stale, i.e. larger) plain read is a safe pre-filter
                if (kp < m1[j]) atomicMin(&WS_M1(c, n[j]), kp);
                if (cand[j] < k2[j]) old[j] = atomicMin(&WS_K2(c, n[j]), cand[j]);
            }
        }
        int was_q[NB];                                              // raw returns: consumed only after all are issued
#pragma unroll
        for (int j = 0; j < NB; j++) {
            was_q[j] = 1;
            if (n[j] >= 0 && cand[j] < old[j]) was_q[j] = inq ? atomicExch(&inq[n[j]], 1) : 0;
        }
#pragma unroll
        for (int j = 0; j < NB; j++) ws_stage(st, w, was_q[j] == 0, n[j], qout, cnt_out, qcap);
    }
}

template <int NN>
__global__ void __launch_bounds__(256)
k_ws_sweep_a(WsC c, const int *__restrict__ qin, const int *__restrict__ cnt_in, int *__restrict__ qout,
             int *__restrict__ cnt_out, int *__restrict__ inq, int qcap, int rounds)
{
    __shared__ WsStage st;
    if (threadIdx.x == 0) { st.cnt[0] = 0; st.cnt[1] = 0; }
    __syncthreads();
    // qin == nullptr: first sweep of the phase = scan of all relevant pixels (no seed queue needed)
    const int64_t n_in = qin ? (int64_t)min(*cnt_in, qcap) : c.R;
    const int64_t n_pad = (n_in + 255) & ~255ll;
    for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < n_pad; i0 += (int64_t)gridDim.x * 256) {
        const int64_t i = i0 + threadIdx.x;
        const bool act = i < n_in;
        const int p = act ? (qin ? qin[i] : (int)i) : 0;
        int w = 0;
        ws_entry_a<NN>(c, st, w, act, p, inq, qout, cnt_out, qcap);
        for (int r = 0; r < rounds; r++) {                     // local rounds: relax what this workgroup has just lowered
            __syncthreads();
            const int n = min(st.cnt[w], WS_LDS_CAP);
            if (n == 0) break;                                 // (uniform: read after the barrier)
            for (int j0 = 0; j0 < n; j0 += 256) {
                const int j = j0 + threadIdx.x;
                const bool a2 = j < n;
                ws_entry_a<NN>(c, st, w ^ 1, a2, a2 ? st.buf[w][j] : 0, inq, qout, cnt_out, qcap);
            }
            __syncthreads();
            if (threadIdx.x == 0) st.cnt[w] = 0;
            w ^= 1;
        }
        ws_flush(st, w, qout, cnt_out, qcap);
    }
}

// After phase A the keys K2 and M1 are final, and with them which edges p -> n are candidate edges (K2[p] == M1[n]: p is
// among the first in-neighbours of n to pop) and which pixels are entries (first pixel of a same-level run).  Every later
// phase -- chain levels, root, the exactness check -- walks candidate edges only; they used to re-derive that per visit
// from M1[n], K2[n] and val[n] of ALL neighbours (three scattered loads per edge, six edges per pixel at connectivity 1,
// one or two of them candidates).  One pass stores the answer: 8 B per relevant pixel.
__global__ void __launch_bounds__(256)
k_ws_edge_masks(WsC c, u64 *__restrict__ emask)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= c.R) return;
    const u64 kp = WS_K2(c, p);
    u64 m = 0ull;
    if (kp != WS_INF) {
        const int *np = c.nbr + p * c.n_nbr;
        for (int i0 = 0; i0 < c.n_nbr; i0 += WS_NB) {
            int n[WS_NB]; u64 m1[WS_NB], kn[WS_NB]; unsigned vn[WS_NB];
#pragma unroll
            for (int j = 0; j < WS_NB; j++) n[j] = i0 + j < c.n_nbr ? np[i0 + j] : -1;
#pragma unroll
            for (int j = 0; j < WS_NB; j++) {
                const int q = n[j] >= 0 ? n[j] : 0;
                const ulonglong2 km = *(const ulonglong2 *)&WS_K2(c, q);
                kn[j] = km.x; m1[j] = km.y; vn[j] = c.val[q];
            }
#pragma unroll
            for (int j = 0; j < WS_NB; j++)
                if (n[j] >= 0 && m1[j] == kp) {
                    m |= 1ull << (i0 + j);
                    if ((kn[j] >> 32) == (u64)vn[j] && (kn[j] & 0xFFFFFFFFull) == 1ull) m |= 1ull << (32 + i0 + j);
                }
        }
    }
    emask[p] = m;
}

// ---- phase k >= 1 (chain level k) and phase R (k == depth) ----------------------------------------
// For edge p -> n with K2[p] == M1[n]:
//   n is an ENTRY (first pixel of a same-level run, pushed from a lower level or by a level marker)
//     iff K2[n] = (value(n), 1): its chain is [K2 n, chain(p)]   -> offered_j = C_{j-1}[p]
//   otherwise n continues p's run / descent: chain = [K2 n, tail(p)] -> offered_j = C_j[p]
// The candidate must agree with n on every level j < k.
__device__ __forceinline__ int ws_load_i(const int *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int NN>
__device__ __forceinline__ void ws_entry_chain(const WsC &c, int k, int depth, u64 *__restrict__ dst, WsStage &st, int w,
                                               bool act0, int p, int *__restrict__ inq,
                                               int *__restrict__ qout, int *__restrict__ cnt_out, int qcap)
{
    constexpr int NB = NN ? NN : WS_NB;
    const int n_nbr = NN ? NN : c.n_nbr;
    const int *np = c.nbr + (int64_t)p * n_nbr;
    const bool root = k == depth;
    u64 kp = WS_INF, own = WS_INF, em = 0ull;
    int own_lo = 0x7fffffff, own_hi = (int)0x80000000;   // label set of p (root phase only)
    u64 cp[WS_MAX_DEPTH];                        // C_j[p], j < k: final since their own phases, read once per entry
    for (int s0 = 0; s0 < n_nbr; s0 += NB) {
        if (s0 == 0 && act0) {
            // the in-queue flag is cleared BEFORE p's own keys are read (the loads take the exchange's
            // return value as an address term), so a later decrease re-queues p: no lost update
            const int dep = inq ? ws_after(atomicExch(&inq[p], 0)) : 0;
            em = c.emask[p];                     // final since phase A (0 for a pixel phase A never reached)
            kp = WS_K2(c, p);
            own = ws_load(&dst[p] + dep);
            if (root) { own_lo = ws_load_i(&c.Llo[p] + dep); own_hi = ws_load_i(&c.Lhi[p] + dep); }
            for (int j = 1; j < k; j++) cp[j] = c.C[j][p];
        }
        // candidate edges only: one or two of the six at connectivity 1
        int n[NB];
#pragma unroll
        for (int j = 0; j < NB; j++) n[j] = (s0 + j < n_nbr && ((em >> (s0 + j)) & 1ull)) ? np[s0 + j] : -1;
        u64 dn[NB];
        int lon[NB], hin[NB];
#pragma unroll
        for (int j = 0; j < NB; j++) {
            dn[j] = WS_INF; lon[j] = 0; hin[j] = 0;
            if (n[j] >= 0) { dn[j] = dst[n[j]]; if (root) { lon[j] = c.Llo[n[j]]; hin[j] = c.Lhi[n[j]]; } }
        }
        u64 offered[NB], old[NB];
        int olo[NB], ohi[NB];              // raw atomic returns of the label-set relaxations
#pragma unroll
        for (int j = 0; j < NB; j++) {
            offered[j] = WS_INF; old[j] = 0ull; olo[j] = (int)0x80000000; ohi[j] = 0x7fffffff;
            if (n[j] >= 0) {
                const bool entry = ((em >> (32 + s0 + j)) & 1ull) != 0ull;
                bool match = true;
                for (int l = 1; l < k && match; l++) {
                    const u64 offered_l = entry ? (l == 1 ? kp : cp[l - 1]) : cp[l];
                    match = offered_l == c.C[l][n[j]];
                }
                if (match) {
                    if (root) offered[j] = own;                                       // root: copied along every edge
                    else offered[j] = entry ? (k == 1 ? kp : cp[k - 1]) : own;
                    if (offered[j] != WS_INF && offered[j] < dn[j]) old[j] = atomicMin(&dst[n[j]], offered[j]);
                    if (root) {
                        // label set of n = union over its fully matching candidates (monotone min / max relaxations)
                        if (own_lo < lon[j]) olo[j] = atomicMin(&c.Llo[n[j]], own_lo);
                        if (own_hi > hin[j]) ohi[j] = atomicMax(&c.Lhi[n[j]], own_hi);
                    }
                }
            }
        }
        int was_q[NB];
#pragma unroll
        for (int j = 0; j < NB; j++) {
            was_q[j] = 1;
            const bool improved = offered[j] < old[j] || (root && (own_lo < olo[j] || own_hi > ohi[j]));
            if (improved) was_q[j] = inq ? atomicExch(&inq[n[j]], 1) : 0;
        }
#pragma unroll
        for (int j = 0; j < NB; j++) ws_stage(st, w, was_q[j] == 0, n[j], qout, cnt_out, qcap);
    }
}

template <int NN>
__global__ void __launch_bounds__(256)
k_ws_sweep_chain(WsC c, int k, int depth, const int *__restrict__ qin, const int *__restrict__ cnt_in,
                 int *__restrict__ qout, int *__restrict__ cnt_out, int *__restrict__ inq, int qcap, int rounds)
{
    __shared__ WsStage st;
    if (threadIdx.x == 0) { st.cnt[0] = 0; st.cnt[1] = 0; }
    __syncthreads();
    const int64_t n_in = qin ? (int64_t)min(*cnt_in, qcap) : c.R;
    const int64_t n_pad = (n_in + 255) & ~255ll;
    u64 *dst = k == depth ? c.Rt : c.C[k];
    for (int64_t i0 = (int64_t)blockIdx.x * 256; i0 < n_pad; i0 += (int64_t)gridDim.x * 256) {
        const int64_t i = i0 + threadIdx.x;
        const bool act0 = i < n_in;
        const int p = act0 ? (qin ? qin[i] : (int)i) : 0;
        int w = 0;
        ws_entry_chain<NN>(c, k, depth, dst, st, w, act0, p, inq, qout, cnt_out, qcap);
        for (int r = 0; r < rounds; r++) {
            __syncthreads();
            const int n = min(st.cnt[w], WS_LDS_CAP);
            if (n == 0) break;
            for (int j0 = 0; j0 < n; j0 += 256) {
                const int j = j0 + threadIdx.x;
                const bool a2 = j < n;
                ws_entry_chain<NN>(c, k, depth, dst, st, w ^ 1, a2, a2 ? st.buf[w][j] : 0, inq, qout, cnt_out, qcap);
            }
            __syncthreads();
            if (threadIdx.x == 0) st.cnt[w] = 0;
            w ^= 1;
        }
        ws_flush(st, w, qout, cnt_out, qcap);
    }
}

// ---- exactness check after a root phase at depth d ------------------------------------------------------------
// The root phase also computed, for every reached pixel n, [Llo(n), Lhi(n)]: the range of labels over the roots of
// ALL candidates whose chain agrees with n's on the d compared levels.  Llo == Lhi: whichever of those candidates the
// reference pops first, n gets this label -- the result does not depend on any tie-break.  Llo != Lhi: the decision
// was made by the last-resort rule (smallest root index), which is the reference's order only if ...
//   * the compared chains are COMPLETE (they ended in a marker key within d levels: low word of C_{d-1}(n) is 0) --
//     then the candidates tie on everything down to EQUAL-VALUED MARKERS, which the reference pops in an order fixed
//     by the array mechanics of its binary heap (_watershed.pyx:278-284 pushes every marker with age 0; :67-152).
//     No order-free formulation reproduces that; the library uses the push order (raster index) and REPORTS the pixel;
//   * otherwise the chains were cut off at d levels (depth exhausted) and the tie is an artefact: the caller deepens.
// An ORIGIN is a pixel where two label sets meet (a fully matching edge p -> n with a different set at p); every
// other ambiguous pixel inherited its set.  org[n]: bit 0 = origin with complete chains, bit 1 = origin cut off.
__global__ void __launch_bounds__(256)
k_ws_origins(WsC c, int depth, int *__restrict__ org)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= c.R) return;
    const u64 kp = WS_K2(c, p);
    if (kp == WS_INF || c.Rt[p] == WS_INF) return;
    const int lo = c.Llo[p], hi = c.Lhi[p];
    u64 cp[WS_MAX_DEPTH];
    for (int j = 1; j < depth; j++) cp[j] = c.C[j][p];
    const u64 em = c.emask[p];
    for (int i = 0; i < c.n_nbr; i++) {
        if (!((em >> i) & 1ull)) continue;                   // candidate edges only
        const int n = c.nbr[p * c.n_nbr + i];
        const bool entry = ((em >> (32 + i)) & 1ull) != 0ull;
        bool match = true;
        for (int l = 1; l < depth && match; l++) {
            const u64 offered_l = entry ? (l == 1 ? kp : cp[l - 1]) : cp[l];
            match = offered_l == c.C[l][n];
        }
        if (!match) continue;
        if (c.Llo[n] != lo || c.Lhi[n] != hi) {
            const bool complete = depth >= 2 && (c.C[depth - 1][n] & 0xFFFFFFFFull) == 0ull;
            atomicOr(&org[n], complete ? 1 : 2);
        }
    }
}

// cnt[0] = pixels whose label depends on a last-resort tie-break, cnt[1] = origins between equal-valued markers,
// cnt[2] = origins whose chains were cut off at the current depth
__global__ void __launch_bounds__(256)
k_ws_count_ambiguous(WsC c, const int *__restrict__ org, unsigned long long *__restrict__ cnt)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int a = 0, t = 0, e = 0;
    if (i < c.R && c.Rt[i] != WS_INF && !(c.pix[i] & WS_MARKER_BIT)) {
        a = c.Llo[i] != c.Lhi[i];
        const int o = org[i];
        t = o & 1; e = (o >> 1) & 1;
    }
    const unsigned long long ma = __ballot(a), mt = __ballot(t), me = __ballot(e);
    if ((threadIdx.x & 63) == 0) {
        if (ma) atomicAdd(&cnt[0], (unsigned long long)__popcll(ma));
        if (mt) atomicAdd(&cnt[1], (unsigned long long)__popcll(mt));
        if (me) atomicAdd(&cnt[2], (unsigned long long)__popcll(me));
    }
}

// labels; optionally the per-voxel report: bit 0 = label depends on a last-resort tie-break (TF_WS_AMB_DEPENDS),
// bit 1 = origin between equal-valued markers (TF_WS_AMB_MARKER_TIE), bit 2 = origin cut off at the final depth
// (TF_WS_AMB_DEPTH)
__global__ void __launch_bounds__(256)
k_ws_labels(const int32_t *__restrict__ markers, const int *__restrict__ cid, const u64 *__restrict__ Rt,
            const u64 *__restrict__ pix, int ranked, const int *__restrict__ lo, const int *__restrict__ hi, const int *__restrict__ org,
            int32_t *__restrict__ labels, uint8_t *__restrict__ amb, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t l = markers[i];
    const int c = cid[i];
    uint8_t a = 0;
    if (l == 0 && c >= 0) {
        const u64 r = Rt[c];
        if (r != WS_INF) {
            l = markers[ranked ? (pix[r & 0xFFFFFFFFull] & ~WS_MARKER_BIT) : r];   // root key: raster index, or (rank << 32) | compact id
            if (amb) a = (uint8_t)((lo[c] != hi[c] ? 1 : 0) | ((org[c] & 3) << 1));
        }
    }
    labels[i] = l;
    if (amb) amb[i] = a;
}

// ---- reference order of equal-valued markers (TF_WS_REFERENCE_ORDER): what the device hands the host replay ----------
// SEED NUMBERS.  Seed k of the reference (marker_locations order = raster order) enters its heap at position k.  The
// numbers come from per-block counts: one workgroup per 256 voxels counts its seeds and its SMALL seeds (key <= vmax: the
// only ones the sparse replay follows), the two count arrays are scanned in 64 bits (a whole config-F stack flooded in one
// call has more than 2^31 seeds), and the list kernels add the prefix inside the block (ballots + popcounts).  No array
// over the voxels is written for this (round 3 flagged the seeds in one pass over the volume and scanned one int per
// voxel: 2.4 GB written and read back per 16 x 5424^2 window).
__global__ void __launch_bounds__(256)
k_ws_seed_counts(const uint8_t *__restrict__ cls, const float *__restrict__ field, int64_t n, unsigned vmax,
                 int *__restrict__ n_seed, int *__restrict__ n_small, int *__restrict__ any_below)
{
    __shared__ int part[2][4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool sd = i < n && cls[i] == 2;
    const unsigned key = sd ? ws_ordkey(field[i]) : 0u;
    const bool sm = sd && key <= vmax;
    if (sm && key < vmax) *any_below = 1;                              // (every writer stores 1) a seed BELOW the tie value: not one value class
    const unsigned long long ms = __ballot(sd), mm = __ballot(sm);
    if ((threadIdx.x & 63) == 0) { part[0][threadIdx.x >> 6] = __popcll(ms); part[1][threadIdx.x >> 6] = __popcll(mm); }
    __syncthreads();
    if (threadIdx.x == 0) {
        n_seed[blockIdx.x] = part[0][0] + part[0][1] + part[0][2] + part[0][3];
        n_small[blockIdx.x] = part[1][0] + part[1][1] + part[1][2] + part[1][3];
    }
}
// the small seeds in seed order: heap position k (64-bit), value key, id in the exported sub-graph (-1: nobody floods from it)
__global__ void __launch_bounds__(256)
k_ws_small_list(const uint8_t *__restrict__ cls, const int *__restrict__ cid, const float *__restrict__ field, int64_t n, unsigned vmax,
                const long long *__restrict__ base_seed, const long long *__restrict__ base_small, const int *__restrict__ subid,
                long long *__restrict__ out_k, unsigned *__restrict__ out_val, int *__restrict__ out_id)
{
    __shared__ int part[2][4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    unsigned key = 0;
    const bool sd = i < n && cls[i] == 2;
    bool sm = false;
    if (sd) { key = ws_ordkey(field[i]); sm = key <= vmax; }
    const unsigned long long ms = __ballot(sd), mm = __ballot(sm);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { part[0][wave] = __popcll(ms); part[1][wave] = __popcll(mm); }
    __syncthreads();
    if (!sm) return;
    const unsigned long long below = (1ull << lane) - 1ull;
    long long k = base_seed[blockIdx.x] + __popcll(ms & below), slot = base_small[blockIdx.x] + __popcll(mm & below);
    for (int w = 0; w < wave; w++) { k += part[0][w]; slot += part[1][w]; }
    const int c = cid[i];
    const int id = c <= -2 ? -2 - c : -1;
    out_k[slot] = k; out_val[slot] = key; out_id[slot] = (id >= 0 && subid) ? subid[id] : id;
}
// dense form: seeds number c0 .. c0 + cap - 1 as the 8-byte heap entries of ws_replay.h (value key | seed flag | id + 1; one
// anonymous LARGE entry for every seed above the tie value): the host sifts them up in place
__global__ void __launch_bounds__(256)
k_ws_seed_list(const uint8_t *__restrict__ cls, const int *__restrict__ cid, const float *__restrict__ field, int64_t n,
               const long long *__restrict__ base_seed, int64_t c0, int64_t cap, const int *__restrict__ subid, unsigned vmax,
               u64 *__restrict__ out_entry)
{
    __shared__ int part[4];
    const long long b0 = base_seed[blockIdx.x];
    if (b0 >= c0 + cap || b0 + 256 <= c0) return;                  // (uniform) no seed of this block falls into the chunk
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool sd = i < n && cls[i] == 2;
    const unsigned long long ms = __ballot(sd);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) part[wave] = __popcll(ms);
    __syncthreads();
    if (!sd) return;
    long long k = b0 + __popcll(ms & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; w++) k += part[w];
    k -= c0;
    if (k < 0 || k >= cap) return;
    const int c = cid[i];
    int id = c <= -2 ? -2 - c : -1;
    const unsigned v = ws_ordkey(field[i]);
    if (id >= 0 && subid && v <= vmax) id = subid[id];              // (a seed above the tie value is not in the sub-graph: its entry is anonymous)
    out_entry[k] = v > vmax ? WS_INF : (((u64)v << 32) | 0x80000000ull | (u64)(unsigned)(id + 1));
}
// THE DENSE FORM IN 2 BITS PER SEED.  When the tie value lies at or above the value most seeds share, every one of those seeds
// has to be in the replay's heap -- 450 M entries, 3.6 GB over PCIe -- but nearly all of them are one of two things: the
// MODAL BALLAST entry D (the background's value, not relevant: nobody floods from it) or LARGE.  The device therefore sends
// two bitmaps over the seed numbers (is D / is LARGE) and the entries of the other seeds in seed order (~ 1 % of them); the
// host's heap build regenerates seed k's entry as it reaches position k (ws_replay.h, WsSeedCodes).
// D is the most frequent ballast entry among 256 samples spread over the volume (any choice is correct, a poor one only sends
// more exceptions).
__global__ void __launch_bounds__(256)
k_ws_entry_sample(const uint8_t *__restrict__ cls, const int *__restrict__ cid, const float *__restrict__ field, int64_t n, unsigned vmax,
                  u64 *__restrict__ out)
{
    const int64_t step = n / 256 > 0 ? n / 256 : 1;
    int64_t i = (int64_t)threadIdx.x * step;
    u64 e = 0ull;
    for (int j = 0; j < 512 && i < n; j++, i++) {
        if (cls[i] != 2 || cid[i] != -1) continue;
        const unsigned v = ws_ordkey(field[i]);
        if (v <= vmax) { e = ((u64)v << 32) | 0x80000000ull; break; }
    }
    out[threadIdx.x] = e;
}
struct WsSeedCode { bool seed, d, l, x; unsigned key; int id; };
__device__ __forceinline__ WsSeedCode ws_seed_code(const uint8_t *cls, const int *cid, const float *field, int64_t i, int64_t n, unsigned vmax, u64 D) {
    WsSeedCode r; r.seed = i < n && cls[i] == 2; r.d = r.l = r.x = false; r.key = 0; r.id = -1;
    if (r.seed) {
        r.key = ws_ordkey(field[i]);
        const int c = cid[i];
        r.id = c <= -2 ? -2 - c : -1;
        r.l = r.key > vmax;
        r.d = !r.l && r.id < 0 && D != 0ull && r.key == (unsigned)(D >> 32);
        r.x = !r.l && !r.d;
    }
    return r;
}
__global__ void __launch_bounds__(256)
k_ws_code_counts(const uint8_t *__restrict__ cls, const int *__restrict__ cid, const float *__restrict__ field, int64_t n, unsigned vmax, u64 D,
                 int *__restrict__ n_exc)
{
    __shared__ int part[4];
    const WsSeedCode r = ws_seed_code(cls, cid, field, (int64_t)blockIdx.x * 256 + threadIdx.x, n, vmax, D);
    const unsigned long long m = __ballot(r.x);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) n_exc[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
// bits_d / bits_l: zeroed by the caller, written when write_bits; exceptions number c0 .. c0 + cap - 1 -> exc[0 .. cap)
__global__ void __launch_bounds__(256)
k_ws_seed_codes(const uint8_t *__restrict__ cls, const int *__restrict__ cid, const float *__restrict__ field, int64_t n, unsigned vmax, u64 D,
                const long long *__restrict__ base_seed, const long long *__restrict__ base_exc, const int *__restrict__ subid,
                int write_bits, unsigned *__restrict__ bits_d, unsigned *__restrict__ bits_l, int64_t c0, int64_t cap, u64 *__restrict__ exc)
{
    __shared__ int part_s[4], part_x[4];
    __shared__ unsigned lb_d[9], lb_l[9];
    if (!write_bits) { const long long b0 = base_exc[blockIdx.x]; if (b0 >= c0 + cap || b0 + 256 <= c0) return; }     // (uniform)
    if (threadIdx.x < 9) { lb_d[threadIdx.x] = 0u; lb_l[threadIdx.x] = 0u; }
    const WsSeedCode r = ws_seed_code(cls, cid, field, (int64_t)blockIdx.x * 256 + threadIdx.x, n, vmax, D);
    const unsigned long long ms = __ballot(r.seed), mx = __ballot(r.x);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { part_s[wave] = __popcll(ms); part_x[wave] = __popcll(mx); }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
    int sl = __popcll(ms & below), xl = __popcll(mx & below);
    for (int w = 0; w < wave; w++) { sl += part_s[w]; xl += part_x[w]; }
    if (write_bits) {
        if (r.d) atomicOr(&lb_d[sl >> 5], 1u << (sl & 31));
        if (r.l) atomicOr(&lb_l[sl >> 5], 1u << (sl & 31));
        __syncthreads();
        if (threadIdx.x < 18) {                                       // the block's 256-bit strings, shifted to its first seed number
            const int t = threadIdx.x % 9;
            const unsigned *lb = threadIdx.x < 9 ? lb_d : lb_l;
            unsigned *bits = threadIdx.x < 9 ? bits_d : bits_l;
            const long long kb = base_seed[blockIdx.x];
            const int sh = (int)(kb & 31);
            const unsigned lo = t < 8 ? lb[t] : 0u, hi = t > 0 ? lb[t - 1] : 0u;
            const unsigned word = sh ? ((lo << sh) | (hi >> (32 - sh))) : lo;
            if (word) atomicOr(&bits[(kb >> 5) + t], word);
        }
    }
    if (r.x) {
        const long long slot = base_exc[blockIdx.x] + xl - c0;
        if (slot >= 0 && slot < cap) {
            int id = r.id;
            if (id >= 0 && subid) id = subid[id];
            exc[slot] = ((u64)r.key << 32) | 0x80000000ull | (u64)(unsigned)(id + 1);
        }
    }
}
// THE SUB-GRAPH THE REPLAY CAN REACH.  The replay pops an item only while its key is at or below the tie value: P = the
// relevant markers with a key <= vmax and the floodable pixels with a key < vmax.  It looks at the out-neighbours of what
// it pops (pushed or not yet, small or large): Q = P + the out-neighbours of P.  On a detect_anvils window P is the rim of
// the saturated cores, ~1 % of the relevant pixels: only Q's keys and P's neighbour rows cross PCIe (round 3 sent the whole
// neighbour table, 28 B per relevant pixel, ~1 GB per 16 x 5424^2 window).
__device__ __forceinline__ bool ws_poppable(const WsC &c, int64_t i, unsigned vmax) {
    const unsigned v = c.val[i];
    return (c.pix[i] & WS_MARKER_BIT) ? v <= vmax : v < vmax;
}
__global__ void __launch_bounds__(256)
k_ws_sub_mark(WsC c, unsigned vmax, uint8_t *__restrict__ in_q, int *__restrict__ any_flooded_small)       // in_q zeroed by the caller
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.R || !ws_poppable(c, i, vmax)) return;
    in_q[i] = 1;
    if (!(c.pix[i] & WS_MARKER_BIT)) *any_flooded_small = 1;          // a floodable pixel below the tie value: pushed, it is a small item of another key
    const int *np = c.nbr + i * c.n_nbr;
    for (int j = 0; j < c.n_nbr; j++) { const int n = np[j]; if (n >= 0) in_q[n] = 1; }     // (every writer stores 1)
}
__global__ void __launch_bounds__(256)
k_ws_sub_export(WsC c, unsigned vmax, const uint8_t *__restrict__ in_q, const int *__restrict__ subid,
                unsigned *__restrict__ out_val, int *__restrict__ out_nbr)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.R || !in_q[i]) return;
    const int64_t q = subid[i];
    out_val[q] = c.val[i];
    const bool pop = ws_poppable(c, i, vmax);
    const int *np = c.nbr + i * c.n_nbr;
    for (int j = 0; j < c.n_nbr; j++) { const int n = pop ? np[j] : -1; out_nbr[q * c.n_nbr + j] = n >= 0 ? subid[n] : -1; }
}
// ---- THE REPLAY IN CLOSED FORM, ON THE DEVICE, WHEN THE SMALL ITEMS ARE ONE VALUE CLASS (round 4) ----------------------------
// The usual case on a detect_anvils field: the tie value is the SMALLEST marker value (-1: the saturated cores), so every
// small item is a seed of that one value (age 0) and no flooded pixel can be small.  Equal keys never swap, and then the
// reference heap's mechanics have a closed form:
//   BUILD.  Seed k enters at position k and rises through its large ancestors until its parent is a small seed: it ends at
//     the FIRST position of its root-to-k chain that no earlier small seed holds (the small seeds are an ancestor-closed set).
//     So node v is taken by the first small seed inside subtree(v) that arrives after the seed that took v's parent --
//     t(v) = min { j in subtree(v) : j > t(parent v) } -- which is one binary search in the sorted list of small seed numbers
//     per tree level of the subtree (the positions of subtree(v) at level L are one contiguous range).
//   POPS.  A popped root is replaced by the last item of the array -- a large one, as long as no small seed sits in the last S
//     positions -- which sinks along "left child if small, else right child if small" and moves every seed on that path up
//     one node: the seeds leave the heap in the PRE-ORDER of the tree they form (ws_replay.h, "the tree of equal seeds").
//     Pushes are all large (a flooded pixel is at or above the tie value) and only lengthen the array.
// The tree is built level by level (one launch per level, <= 30), subtree sizes bottom-up, pre-order indices top-down; the
// pre-order index IS the pop rank.  ~2 ms per 16 x 5424^2 window instead of a 0.2 s host pass, no export at all.  The host
// replay remains for every other case (several small values, flooded pixels below the tie value, small seeds at the very
// end of the array) and as the cross-check (TF_WS_REFERENCE_HOST=1; tests/test_gpu_reference_order.py).
struct WsTie { const long long *k; const int *sid; int S; long long M; long long *pos; int *tj, *c0, *c1, *size, *pre; int *n_nodes; };
__device__ __forceinline__ int ws_tie_lower_bound(const long long *k, int lo, int hi, long long a) {
    while (lo < hi) { const int mid = lo + ((hi - lo) >> 1); if (k[mid] < a) lo = mid + 1; else hi = mid; }
    return lo;
}
__global__ void __launch_bounds__(256)
k_ws_tie_level(WsTie t, int lo, int hi)
{
    const int n = lo + (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (n >= hi) return;
    const long long p = t.pos[n];
    const int j = t.tj[n];
    const long long kprev = t.k[j];
#pragma unroll
    for (int side = 0; side < 2; side++) {
        const long long v = 2 * p + 1 + side;
        int jj = -1;
        for (int sh = 0; sh < 62; sh++) {                              // the positions of subtree(v), level by level
            const long long lo_l = ((v + 1) << sh) - 1;
            if (lo_l >= t.M) break;
            const long long hi_l = lo_l + (1ll << sh) - 1;
            if (hi_l <= kprev) continue;                               // seeds there arrived before the parent's
            const long long a = lo_l > kprev + 1 ? lo_l : kprev + 1;
            const int idx = ws_tie_lower_bound(t.k, j + 1, t.S, a);
            if (idx < t.S && t.k[idx] <= hi_l) { jj = idx; break; }    // (ranges and seed numbers both ascend: the first hit is the earliest arrival)
        }
        int m = -1;
        if (jj >= 0) { m = atomicAdd(t.n_nodes, 1); if (m < t.S) { t.pos[m] = v; t.tj[m] = jj; } else m = -1; }
        if (side == 0) t.c0[n] = m; else t.c1[n] = m;
    }
}
__global__ void __launch_bounds__(256)
k_ws_tie_sizes(WsTie t, int lo, int hi)
{
    const int n = lo + (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (n >= hi) return;
    const int a = t.c0[n], b = t.c1[n];
    t.size[n] = 1 + (a >= 0 ? t.size[a] : 0) + (b >= 0 ? t.size[b] : 0);
}
__global__ void __launch_bounds__(256)
k_ws_tie_pre(WsTie t, int lo, int hi)
{
    const int n = lo + (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (n >= hi) return;
    const int a = t.c0[n], b = t.c1[n], p = t.pre[n];
    if (a >= 0) t.pre[a] = p + 1;
    if (b >= 0) t.pre[b] = p + 1 + (a >= 0 ? t.size[a] : 0);
}
__global__ void __launch_bounds__(256)
k_ws_tie_default_ranks(int64_t R, int S, int *__restrict__ rank)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < R) rank[i] = (int)(S + i);                                // markers that do not pop: after all that do (no label depends on their order)
}
__global__ void __launch_bounds__(256)
k_ws_tie_scatter(WsTie t, int *__restrict__ rank)
{
    const int n = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (n >= t.S) return;
    const int id = t.sid[t.tj[n]];
    if (id >= 0) rank[id] = t.pre[n];
}

// pop ranks back on the compact set: a marker the replay popped gets its rank, every other one keeps its place after all
// of those (no label depends on their order)
__global__ void __launch_bounds__(256)
k_ws_scatter_ranks(const uint8_t *__restrict__ in_q, const int *__restrict__ subid, const int *__restrict__ rank_q, int n_ranked,
                   int64_t R, int *__restrict__ rank)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R) return;
    int r = -1;
    if (!in_q) r = rank_q[i];                                     // identity ids: the sub-graph is the whole compact set
    else if (in_q[i]) r = rank_q[subid[i]];
    rank[i] = r >= 0 ? r : (int)(n_ranked + i);
}
// largest marker value (ordered key) below which the order of equal-valued markers decides a label: every origin
// with complete chains ties down to markers of ONE value, the value of its own root
__global__ void __launch_bounds__(256)
k_ws_tie_value_max(WsC c, const int *__restrict__ org, const float *__restrict__ field, int ranked, unsigned *__restrict__ vmax)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.R || !(org[i] & 1)) return;
    const u64 r = c.Rt[i];                               // root key: the raster index of the root marker, or (pop rank << 32) | compact id
    if (r != WS_INF) atomicMax(vmax, ws_ordkey(field[ranked ? (c.pix[r & 0xFFFFFFFFull] & ~WS_MARKER_BIT) : r]));
}

struct WsU8ToInt { __host__ __device__ __forceinline__ int operator()(uint8_t v) const { return (int)v; } };
typedef rocprim::transform_iterator<const uint8_t *, WsU8ToInt, int> WsFlagIter;

// The flag scan runs in chunks of at most 2^30 voxels (the scan primitive counts its items in an int), each continuing
// from the total of the chunks before it: volumes beyond 2^31 voxels -- config F's 144 full-disk frames as ONE exact
// flood -- only need the RELEVANT pixel count to fit the int32 compact ids.
#define WS_SCAN_CHUNK (1ll << 30)
struct WsIntToLL { __host__ __device__ __forceinline__ long long operator()(int v) const { return (long long)v; } };
typedef rocprim::transform_iterator<const int *, WsIntToLL, long long> WsCountIter;
static size_t ws_scan_temp_bytes(int64_t n) {
    size_t bytes = 0, b2 = 0, b3 = 0;
    WsFlagIter it((const uint8_t *)nullptr, WsU8ToInt());
    // size queries only (null temp storage): the flag scan over n positions, and the 64-bit scans of per-block counts
    // (tile ids; seed numbers of the reference-order export) -- on a small volume the latter need MORE than the former
    (void)tf_exclusive_sum_from(nullptr, bytes, it, (int *)nullptr, 0, (size_t)(n > WS_SCAN_CHUNK ? WS_SCAN_CHUNK : n));
    const int64_t nb = (n + 255) / 256 < 0x7fffffffll ? (n + 255) / 256 : 0x7ffffffell;
    (void)tf_exclusive_sum(nullptr, b2, (const long long *)nullptr, (long long *)nullptr, (size_t)nb);
    (void)tf_exclusive_sum(nullptr, b3, WsCountIter((const int *)nullptr, WsIntToLL()), (long long *)nullptr, (size_t)nb);
    if (b2 > bytes) bytes = b2;
    if (b3 > bytes) bytes = b3;
    return bytes;
}

// scan[i] = number of flagged voxels before i; *total = number of flagged voxels.  Synchronises the stream.
static int ws_scan_flags(const uint8_t *flag, int *scan, int64_t N, void *tmp, size_t tmp_bytes, hipStream_t s, int64_t *total)
{
    int64_t carry = 0;
    for (int64_t off = 0; off < N; off += WS_SCAN_CHUNK) {
        const int64_t n = N - off < WS_SCAN_CHUNK ? N - off : WS_SCAN_CHUNK;
        size_t tb = tmp_bytes;
        WsFlagIter it(flag + off, WsU8ToInt());
        TF_CHECK_HIP(tf_exclusive_sum_from(tmp, tb, it, scan + off, (int)carry, (size_t)n, s));
        int last_scan = 0; uint8_t last_flag = 0;
        TF_CHECK_HIP(hipMemcpyAsync(&last_scan, scan + off + n - 1, sizeof(int), hipMemcpyDeviceToHost, s));
        TF_CHECK_HIP(hipMemcpyAsync(&last_flag, flag + off + n - 1, 1, hipMemcpyDeviceToHost, s));
        TF_CHECK_HIP(hipStreamSynchronize(s));
        carry = (int64_t)last_scan + last_flag;
        // a chunk adds at most 2^30: the running total cannot wrap an int before this test sees it
        TF_REQUIRE(carry <= 0x3fffff00ll, "tf_watershed: more than 2^30 relevant pixels in one call (use time windows)");
    }
    *total = carry;
    return TF_OK;
}

static size_t ws_full_bytes(int64_t N, int64_t NV) {
    // cls, cid over the N voxels; flag, scan over the NV >= N scan positions (tile order pads edge tiles); scan temp, flags
    return tf_align_up((size_t)N, 256) + tf_align_up((size_t)NV, 256) + tf_align_up((size_t)N * 4, 256) + tf_align_up((size_t)NV * 4, 256)
         + tf_align_up(ws_scan_temp_bytes(NV), 256) + tf_align_up((size_t)((N + 255) / 256) * 24 + 64, 256) + 4096;   // ... and the per-block seed counts / bases of the reference-order export
}
static size_t ws_compact_bytes(int64_t R, int n_nbr, int depth) {
    // pix + val + nbr + keys + queues
    return tf_align_up((size_t)R * 8, 256) + tf_align_up((size_t)R * 4, 256) + tf_align_up((size_t)R * 4 * n_nbr, 256)
         + (size_t)(depth + 2) * tf_align_up((size_t)R * 8, 256)          // K2, M1, C_1..C_{d-1}, Rt
         + 2 * tf_align_up((size_t)R * 8 + 256, 256) + tf_align_up((size_t)R * 4, 256)           // two queues (2R ints), inq
         + 5 * tf_align_up((size_t)R * 4, 256) + tf_align_up((size_t)R * 8, 256) + 4096;         // Llo, Lhi, org, rank, sub-graph ids; edge masks
}

extern "C" size_t tf_watershed_workspace_bytes(int64_t T, int64_t H, int64_t W, int n_nbr, int chain_depth, int64_t max_relevant)
{
    if (T <= 0 || H <= 0 || W <= 0 || chain_depth < 1 || chain_depth > WS_MAX_DEPTH || n_nbr < 1 || n_nbr > WS_MAX_NBR) return 0;
    const int64_t N = T * H * W;
    if (max_relevant <= 0 || max_relevant > N) max_relevant = N;
    return ws_full_bytes(N, ws_virtual_voxels(T, H, W)) + ws_compact_bytes(max_relevant, n_nbr, chain_depth);
}

struct WsQueues { int *q[2]; int *cnt; int *inq; int qcap; int *h_cnt; int64_t *processed; };

// Run one relaxation phase to its fixpoint.  phase_k = 0: K2/M1; otherwise chain level k (k == depth: root).
// alg_bytes: the algorithmic bytes the timing facility books on this phase's first batch of sweeps.  The flood's figure is
// SURVEY section 8(d)'s: 29 B per voxel for ONE ideal sweep over the volume (field 4 + markers 4 + labels 4 + mask 1 + flows
// 16) -- booked once per flood, on phase A; the other phases add time, not algorithmic bytes (the roofline entry of
// `ws_relax_sweep` then says what the sweeps cost against the single pass an ideal flood would need).
static int ws_run_phase(const WsC &c, int phase_k, int depth, const WsQueues &Q, hipStream_t s, int64_t max_sweeps,
                        int64_t *sweeps_out, double alg_bytes = 0.0)
{
    const unsigned nbR = (unsigned)((c.R + 255) / 256);
    const unsigned nb = nbR < 2048u ? nbR : 2048u;
    // development switches (same labels either way): TF_WS_LOCAL_ROUNDS=<n> (0 = one front step per launch),
    // TF_WS_DEDUPE=1 (in-queue flags: every pixel at most once per queue, the round-1 scheme)
    static const int rounds = getenv("TF_WS_LOCAL_ROUNDS") ? atoi(getenv("TF_WS_LOCAL_ROUNDS")) : 4;
    static const bool dedupe = getenv("TF_WS_DEDUPE") != nullptr;
    int *inq = dedupe ? Q.inq : nullptr;
    TF_CHECK_HIP(hipMemsetAsync(Q.cnt, 0, (WS_BATCH + 1) * sizeof(int), s));
    if (dedupe) TF_CHECK_HIP(hipMemsetAsync(Q.inq, 0, (size_t)c.R * sizeof(int), s));
    int64_t sweeps = 0;
    int parity = 0;
    bool first = true;                // the next launch scans every relevant pixel instead of reading a queue
    unsigned grid_hint = nb;          // sized from the frontier seen at the end of the previous batch
    for (;;) {
        const bool scan_batch = first;
        // one timing scope per batch of launches, not per launch: ~1000 event pairs per call cost more than the
        // small sweeps themselves (measured: 8.5 ms per 12x5424^2 step); the scope therefore includes dispatch gaps
        {
            TfProfScope ps(TFK_WS_RELAX, alg_bytes, s);                // stop event recorded right after the last launch
            alg_bytes = 0.0;                                           // (booked once, on the first batch)
            for (int b = 0; b < WS_BATCH; b++) {
                const int *qin = first ? nullptr : Q.q[parity];
                const unsigned blocks = first ? nbR : grid_hint;
                if (phase_k == 0) {
                    if (c.n_nbr == 6) hipLaunchKernelGGL(k_ws_sweep_a<6>, dim3(blocks), dim3(256), 0, s, c, qin, Q.cnt + b, Q.q[parity ^ 1], Q.cnt + b + 1, inq, Q.qcap, rounds);
                    else hipLaunchKernelGGL(k_ws_sweep_a<0>, dim3(blocks), dim3(256), 0, s, c, qin, Q.cnt + b, Q.q[parity ^ 1], Q.cnt + b + 1, inq, Q.qcap, rounds);
                } else {
                    if (c.n_nbr == 6) hipLaunchKernelGGL(k_ws_sweep_chain<6>, dim3(blocks), dim3(256), 0, s, c, phase_k, depth, qin, Q.cnt + b, Q.q[parity ^ 1], Q.cnt + b + 1, inq, Q.qcap, rounds);
                    else hipLaunchKernelGGL(k_ws_sweep_chain<0>, dim3(blocks), dim3(256), 0, s, c, phase_k, depth, qin, Q.cnt + b, Q.q[parity ^ 1], Q.cnt + b + 1, inq, Q.qcap, rounds);
                }
                parity ^= 1;
                first = false;
            }
        }
        TF_CHECK_LAUNCH();
        TF_CHECK_HIP(hipMemcpyAsync(Q.h_cnt, Q.cnt, (WS_BATCH + 1) * sizeof(int), hipMemcpyDeviceToHost, s));
        TF_CHECK_HIP(hipStreamSynchronize(s));
        bool done = false, overflow = false;
        // slot b holds the size of the queue consumed by sweep b (slot 0 of a batch that starts with a full scan is
        // unused)
        static const bool trace = getenv("TF_WS_TRACE") != nullptr;      // development aid: frontier size per sweep
        if (trace) {
            fprintf(stderr, "ws_trace phase %d sweeps %lld:", phase_k, (long long)sweeps);
            for (int b = 0; b <= WS_BATCH; b++) fprintf(stderr, " %d", Q.h_cnt[b]);
            fprintf(stderr, "\n");
        }
        for (int b = (scan_batch ? 1 : 0); b <= WS_BATCH; b++) overflow = overflow || Q.h_cnt[b] > Q.qcap;
        if (overflow && dedupe) { tf_set_error("tf_watershed: frontier queue overflow"); return TF_EHIP; }
        for (int b = (scan_batch ? 1 : 0); b <= WS_BATCH && !overflow; b++) {
            if (Q.h_cnt[b] == 0) { done = true; break; }
            if (b < WS_BATCH) *Q.processed += Q.h_cnt[b];
        }
        sweeps += WS_BATCH;
        {   // frontiers shrink slowly: 4x the last frontier (in 256-pixel chunks) is a generous grid for the next batch
            const int64_t last = Q.h_cnt[WS_BATCH];
            int64_t gb = (last * 4 + 255) / 256;
            if (gb < 64) gb = 64;
            if (gb > (int64_t)nb) gb = nb;
            grid_hint = (unsigned)gb;
        }
        if (done) break;
        if (sweeps > max_sweeps) { tf_set_error("tf_watershed: phase %d did not converge in %lld sweeps", phase_k, (long long)sweeps); return TF_ENOCONV; }
        if (overflow) {
            // a queue was cut off: appended pixels were dropped.  Every key is still a valid upper bound (relaxations only
            // lower keys), so a scan of ALL relevant pixels re-seeds the fronts and the phase carries on to the same fixpoint
            first = true;
            TF_CHECK_HIP(hipMemsetAsync(Q.cnt, 0, (WS_BATCH + 1) * sizeof(int), s));
            continue;
        }
        // carry the last count into slot 0, clear the rest
        TF_CHECK_HIP(hipMemcpyAsync(Q.cnt, Q.cnt + WS_BATCH, sizeof(int), hipMemcpyDeviceToDevice, s));
        TF_CHECK_HIP(hipMemsetAsync(Q.cnt + 1, 0, WS_BATCH * sizeof(int), s));
    }
    *sweeps_out = sweeps;
    return TF_OK;
}

// ---- the reference's pop order of equal-valued markers: the host replays live in ws_replay.h (pure C++, also built by
// tools/replay_check for CPU-side checks); here: their scratch, what the device exports for them, and the job around them
// host scratch of the replays: plain memory kept between calls (a plain-form replay touches ~6 GB: as fresh mallocs that
// is ~1.5 M page faults per call) and pinned memory for the device -> host exports; grow-only, freed by tf_shutdown()
#include <vector>
namespace {
struct WsHostBuf { void *p; size_t bytes; bool pinned; };
std::mutex g_ws_host_mu;
std::vector<WsHostBuf> g_ws_host_free, g_ws_host_lent;
}
static WsHostBuf ws_host_take(size_t bytes, bool pinned)
{
    if (bytes == 0) bytes = 64;
    {
        std::lock_guard<std::mutex> lk(g_ws_host_mu);
        int best = -1;
        for (size_t i = 0; i < g_ws_host_free.size(); i++) {
            const WsHostBuf &b = g_ws_host_free[i];
            if (b.pinned == pinned && b.bytes >= bytes && (best < 0 || b.bytes < g_ws_host_free[best].bytes)) best = (int)i;
        }
        if (best >= 0) { WsHostBuf b = g_ws_host_free[best]; g_ws_host_free.erase(g_ws_host_free.begin() + best); return b; }
    }
    WsHostBuf b{nullptr, bytes + bytes / 8, pinned};                // a little slack: the next window is rarely the same size
    if (pinned) { if (hipHostMalloc(&b.p, b.bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); b.p = nullptr; } }
    else b.p = malloc(b.bytes);
    if (!b.p) b.bytes = 0;
    return b;
}
static void ws_host_give(WsHostBuf &b)
{
    if (!b.p) return;
    std::lock_guard<std::mutex> lk(g_ws_host_mu);
    g_ws_host_free.push_back(b);
    b.p = nullptr; b.bytes = 0;
}
// malloc / free look-alikes over the pool for ws_replay.h (WSR_ALLOC / WSR_FREE)
static void *ws_pool_alloc(size_t bytes)
{
    WsHostBuf b = ws_host_take(bytes, false);
    if (!b.p) return nullptr;
    std::lock_guard<std::mutex> lk(g_ws_host_mu);
    g_ws_host_lent.push_back(b);
    return b.p;
}
static void ws_pool_free(void *p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_ws_host_mu);
    for (size_t i = 0; i < g_ws_host_lent.size(); i++)
        if (g_ws_host_lent[i].p == p) { g_ws_host_free.push_back(g_ws_host_lent[i]); g_ws_host_lent.erase(g_ws_host_lent.begin() + i); return; }
}
void tf_ws_host_pool_release()                                      // tf_shutdown()
{
    std::lock_guard<std::mutex> lk(g_ws_host_mu);
    for (auto &b : g_ws_host_free) { if (b.pinned) (void)hipHostFree(b.p); else free(b.p); }
    g_ws_host_free.clear();
}
#define WSR_ALLOC(bytes) ws_pool_alloc(bytes)
#define WSR_FREE(p) ws_pool_free(p)
#include "ws_replay.h"

// ---- one flood = a JOB in three parts ---------------------------------------------------------------------------------
//   begin   (device)  classification / compaction, phase A, chain + root phases at increasing depth until the exactness
//                     check finds no origin whose chains were cut off; with TF_WS_REFERENCE_ORDER and labels that hang on
//                     the order of equal-valued markers: the tie value, the seed numbers and the sub-graph the replay can
//                     reach, copied into pinned host memory
//   replay  (host)    the reference heap's mechanics for the pop ranks -- no HIP call, any thread: the replays of several
//                     windows run side by side while the device floods the next ones
//   finish  (device)  pop ranks up, root phase repeated with them, labels written
// tf_watershed_ex2 and friends run the three back to back; tf_watershed_begin / _replay / _finish expose them.
struct tf_ws_job {
    const float *field; const int32_t *markers; const int8_t *mask;
    int64_t T, H, W, N, NV, R;
    int n_nbr, depth0, depth_max, flags;
    hipStream_t s;
    bool raveled;
    WsGeom g;
    uint8_t *cls, *flag; int *scan, *cid; char *scan_tmp; size_t scan_bytes;
    int *d_flags; unsigned long long *d_cnt; char *seed_blocks;
    WsC c; WsQueues Q; int *org, *rank_dev;
    int h_cnt[WS_BATCH + 8];
    int depth; int64_t max_sweeps;
    unsigned long long h_amb[4];
    int64_t st[TF_WS_NSTATS];
    // reference order
    bool need_replay, replay_done, applied, sparse, plain, identity, coded;
    bool speculative;            // the export ran before the root phase, for a tie value the caller guessed (spec_vmax)
    bool root_pending;           // finish starts with the root phase (begin has run phase A and the chain levels below `depth`)
    bool sweeps_pending;         // TF_WS_DEFER_SWEEPS: begin returned after the set-up and the export; phase A and the chain levels are tf_watershed_sweeps' (or finish's)
    bool speculate_fast; int levels_done;
    unsigned true_vmax; bool has_tie, spec_hit; int *subid;
    bool ranked, late_export; double ms_detour;
    bool device_ranks;           // the small items were one value class: rank_dev filled on the device (k_ws_tie_*), no host replay
    u64 code_d; int64_t code_words, n_exc;
    int64_t M, S, nQ; unsigned vmax;
    WsHostBuf hb_val, hb_nbr, hb_rank, hb_sk, hb_sval, hb_sid;
    int64_t popped; int n_ranked; int replay_rc;
    double ms_export, ms_replay;
};

static double ws_now_ms() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static bool ws_env(const char *name) { return getenv(name) != nullptr; }

static void ws_job_free(tf_ws_job *j)
{
    if (!j) return;
    ws_host_give(j->hb_val); ws_host_give(j->hb_nbr); ws_host_give(j->hb_rank);
    ws_host_give(j->hb_sk); ws_host_give(j->hb_sval); ws_host_give(j->hb_sid);
    delete j;
}

// device -> pinned host, waits
static int ws_to_host(void *dst, const void *src, size_t bytes, hipStream_t s)
{
    if (bytes == 0) return TF_OK;
    TF_CHECK_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s));
    return TF_OK;
}

// begin, part 2: what the host replay needs (see tf_ws_job)
static int ws_job_export(tf_ws_job *j)
{
    const double t_enter = ws_now_ms();
    j->device_ranks = false;
    hipStream_t s = j->s;
    const WsC &c = j->c;
    const int64_t R = j->R, N = j->N;
    const int nn = j->n_nbr;
    const unsigned nbr_blocks = (unsigned)((R + 255) / 256);
    static const bool ref_debug = ws_env("TF_WS_REF_DEBUG");           // development aid: where the detour's time goes
    // A/B and test aid (read per call: the tests toggle it): TF_WS_REFERENCE_DENSE=1 the dense form whatever the seed counts, =2 the plain form
    const char *dense_env = getenv("TF_WS_REFERENCE_DENSE");
    const bool force_dense = dense_env != nullptr;
    j->plain = dense_env && atoi(dense_env) == 2;
    // the largest marker value at which the order of equal-valued markers decides a label (the caller's: j->vmax)
    const unsigned h_vmax = j->vmax;
    // seed numbers: per-block counts of seeds / small seeds and their 64-bit scans, in the idle flag array
    const int64_t nb256 = (N + 255) / 256;
    TF_REQUIRE(nb256 < 0x7fffffffll, "tf_watershed: too many voxels for the seed numbering");
    int *n_seed = (int *)j->seed_blocks, *n_small = n_seed + nb256;
    long long *base_seed = (long long *)(((uintptr_t)(n_small + nb256) + 15) & ~(uintptr_t)15), *base_small = base_seed + nb256;
    size_t need = 0;
    (void)tf_exclusive_sum(nullptr, need, WsCountIter((const int *)nullptr, WsIntToLL()), (long long *)nullptr, (size_t)nb256);
    TF_REQUIRE(need <= j->scan_bytes, "tf_watershed: scan scratch too small for the seed numbering");
    int *d_any = j->d_flags + WS_BATCH + 6;                             // [0] a seed below the tie value, [1] a floodable pixel below it
    TF_CHECK_HIP(hipMemsetAsync(d_any, 0, 2 * sizeof(int), s));
    hipLaunchKernelGGL(k_ws_seed_counts, dim3((unsigned)nb256), dim3(256), 0, s, (const uint8_t *)j->cls, j->field, N, h_vmax, n_seed, n_small, d_any);
    TF_CHECK_LAUNCH();
    size_t tb = j->scan_bytes;
    TF_CHECK_HIP(tf_exclusive_sum(j->scan_tmp, tb, WsCountIter(n_seed, WsIntToLL()), base_seed, (size_t)nb256, s));
    tb = j->scan_bytes;
    TF_CHECK_HIP(tf_exclusive_sum(j->scan_tmp, tb, WsCountIter(n_small, WsIntToLL()), base_small, (size_t)nb256, s));
    long long h_last[4] = {0, 0, 0, 0}; int h_lastc[2] = {0, 0};
    TF_CHECK_HIP(hipMemcpyAsync(&h_last[0], base_seed + nb256 - 1, sizeof(long long), hipMemcpyDeviceToHost, s));
    TF_CHECK_HIP(hipMemcpyAsync(&h_last[1], base_small + nb256 - 1, sizeof(long long), hipMemcpyDeviceToHost, s));
    TF_CHECK_HIP(hipMemcpyAsync(&h_lastc[0], n_seed + nb256 - 1, sizeof(int), hipMemcpyDeviceToHost, s));
    TF_CHECK_HIP(hipMemcpyAsync(&h_lastc[1], n_small + nb256 - 1, sizeof(int), hipMemcpyDeviceToHost, s));
    // the sub-graph: Q = poppable pixels + their out-neighbours (flags in the idle scan array, ids in the idle in-queue flags)
    uint8_t *in_q = (uint8_t *)j->scan;
    int *subid = j->subid;
    TF_CHECK_HIP(hipMemsetAsync(in_q, 0, (size_t)R, s));
    hipLaunchKernelGGL(k_ws_sub_mark, dim3(nbr_blocks), dim3(256), 0, s, c, h_vmax, in_q, d_any + 1);
    TF_CHECK_LAUNCH();
    int h_any[2] = {0, 0};
    TF_CHECK_HIP(hipMemcpyAsync(h_any, d_any, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    int64_t nQ = 0;
    {
        const int rc = ws_scan_flags(in_q, subid, R, j->scan_tmp, j->scan_bytes, s, &nQ);       // synchronises
        if (rc) return rc;
    }
    j->M = h_last[0] + h_lastc[0]; j->S = h_last[1] + h_lastc[1];
    {
        // ONE VALUE CLASS?  Then the pop ranks have a closed form and are computed here, on the device (k_ws_tie_*).
        const bool force_host = ws_env("TF_WS_REFERENCE_HOST");                          // A/B and test aid (read per call: the tests toggle it)
        const int64_t S = j->S;
        char *tie_base = (char *)j->scan + tf_align_up((size_t)R, 256);
        const size_t tie_room = (size_t)j->NV * 4 > tf_align_up((size_t)R, 256) ? (size_t)j->NV * 4 - tf_align_up((size_t)R, 256) : 0;
        const size_t tie_need = tf_align_up((size_t)S * 8, 256) + 5 * tf_align_up((size_t)S * 4, 256) + 256;
        if (ref_debug)
            fprintf(stderr, "reference order: closed form? seed below the tie value %d, floodable pixel below it %d, S %lld, R %lld, room %zu of %zu\n",
                    h_any[0], h_any[1], (long long)S, (long long)R, tie_room, tie_need);
        if (!force_host && !force_dense && !h_any[0] && !h_any[1] && S > 0 && S <= R && S < 0x3fffffffll && tie_need <= tie_room) {
            long long *stg_k = (long long *)c.Rt; unsigned *stg_val = (unsigned *)c.Llo; int *stg_id = c.Lhi;
            hipLaunchKernelGGL(k_ws_small_list, dim3((unsigned)nb256), dim3(256), 0, s, (const uint8_t *)j->cls, (const int *)j->cid, j->field, N, h_vmax,
                               (const long long *)base_seed, (const long long *)base_small, (const int *)nullptr, stg_k, stg_val, stg_id);
            TF_CHECK_LAUNCH();
            long long k_max = 0;
            TF_CHECK_HIP(hipMemcpyAsync(&k_max, stg_k + S - 1, sizeof(long long), hipMemcpyDeviceToHost, s));
            TF_CHECK_HIP(hipStreamSynchronize(s));
            if (ref_debug && k_max >= j->M - S) fprintf(stderr, "reference order: closed form refused: small seed at heap position %lld of %lld\n", k_max, (long long)j->M);
            if (k_max < j->M - S) {                                     // no small seed among the last S positions: every replacement item is large
                TfArena ta(tie_base, tie_room);
                WsTie t;
                t.k = stg_k; t.sid = stg_id; t.S = (int)S; t.M = j->M;
                t.pos = ta.take<long long>(S); t.tj = ta.take<int>(S); t.c0 = ta.take<int>(S); t.c1 = ta.take<int>(S);
                t.size = ta.take<int>(S); t.pre = ta.take<int>(S); t.n_nodes = ta.take<int>(4);
                if (ta.ok()) {
                    const long long h_pos0 = 0; const int h_init[2] = {1, 0};      // node 0 = the root, taken by the first small seed; pre-order index 0
                    TF_CHECK_HIP(hipMemcpyAsync(t.pos, &h_pos0, sizeof(long long), hipMemcpyHostToDevice, s));
                    TF_CHECK_HIP(hipMemsetAsync(t.tj, 0, sizeof(int), s));
                    TF_CHECK_HIP(hipMemsetAsync(t.pre, 0, sizeof(int), s));
                    TF_CHECK_HIP(hipMemcpyAsync(t.n_nodes, h_init, sizeof(int), hipMemcpyHostToDevice, s));
                    int offs[80]; int n_lev = 0, lo = 0, hi = 1;
                    offs[0] = 0;
                    while (hi > lo && n_lev < 78) {
                        hipLaunchKernelGGL(k_ws_tie_level, dim3((unsigned)((hi - lo + 255) / 256)), dim3(256), 0, s, t, lo, hi);
                        TF_CHECK_LAUNCH();
                        int h_n = 0;
                        TF_CHECK_HIP(hipMemcpyAsync(&h_n, t.n_nodes, sizeof(int), hipMemcpyDeviceToHost, s));
                        TF_CHECK_HIP(hipStreamSynchronize(s));
                        offs[++n_lev] = hi;
                        lo = hi; hi = h_n < (int)S ? h_n : (int)S;
                        if (h_n > (int)S) { lo = hi = -1; break; }      // (cannot happen: every seed takes exactly one node)
                    }
                    if (hi == (int)S && lo == hi) {                     // every small seed has its node
                        for (int l = n_lev - 1; l >= 0; l--)
                            hipLaunchKernelGGL(k_ws_tie_sizes, dim3((unsigned)((offs[l + 1] - offs[l] + 255) / 256)), dim3(256), 0, s, t, offs[l], offs[l + 1]);
                        for (int l = 0; l < n_lev; l++)
                            hipLaunchKernelGGL(k_ws_tie_pre, dim3((unsigned)((offs[l + 1] - offs[l] + 255) / 256)), dim3(256), 0, s, t, offs[l], offs[l + 1]);
                        hipLaunchKernelGGL(k_ws_tie_default_ranks, dim3(nbr_blocks), dim3(256), 0, s, R, (int)S, j->rank_dev);
                        hipLaunchKernelGGL(k_ws_tie_scatter, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, s, t, j->rank_dev);
                        TF_CHECK_LAUNCH();
                        TF_CHECK_HIP(hipStreamSynchronize(s));
                        j->device_ranks = true;
                        j->need_replay = false;
                        j->popped = S; j->n_ranked = (int)S;
                        j->ms_export = ws_now_ms() - t_enter;
                        if (ref_debug)
                            fprintf(stderr, "reference order: %lld seeds, %lld at the tie value (key %u) and nothing below it: pop ranks on the device "
                                    "(tree of %d levels), %.1f ms\n", (long long)j->M, (long long)S, h_vmax, n_lev, j->ms_export);
                        return TF_OK;
                    }
                }
            }
        }
    }
    // compact export if the translated rows fit the staging room (the two idle frontier queues), else the whole compact
    // set with its own ids
    j->identity = nQ * nn > 2 * R + 64;
    if (j->identity) nQ = R;
    j->nQ = nQ;
    // the small seeds are staged in Rt / Llo / Lhi (idle until the root phase is repeated): R entries each
    j->sparse = !force_dense && j->S <= R;
    const double t_numbered = ws_now_ms();
    j->hb_val = ws_host_take((size_t)nQ * sizeof(unsigned), true);
    j->hb_nbr = ws_host_take((size_t)nQ * nn * sizeof(int), true);
    j->hb_rank = ws_host_take((size_t)nQ * sizeof(int), true);
    // sparse: heap position, key, id of the small seeds (the dense form takes its buffers below)
    if (j->sparse) {
        j->hb_sk = ws_host_take((size_t)j->S * sizeof(long long), true);
        j->hb_sval = ws_host_take((size_t)j->S * sizeof(unsigned), true); j->hb_sid = ws_host_take((size_t)j->S * sizeof(int), true);
    }
    if (!j->hb_val.p || !j->hb_nbr.p || !j->hb_rank.p || (j->sparse && (!j->hb_sk.p || !j->hb_sval.p || !j->hb_sid.p))) {
        tf_set_error("tf_watershed: out of host memory for the reference-order replay");
        return TF_ENOMEM;
    }
    if (j->identity) {
        if (ws_to_host(j->hb_val.p, c.val, (size_t)R * sizeof(unsigned), s) || ws_to_host(j->hb_nbr.p, c.nbr, (size_t)R * nn * sizeof(int), s)) return TF_EHIP;
    } else {
        unsigned *stg_val = (unsigned *)j->Q.q[0]; int *stg_nbr = j->Q.q[1];
        hipLaunchKernelGGL(k_ws_sub_export, dim3(nbr_blocks), dim3(256), 0, s, c, h_vmax, (const uint8_t *)in_q, (const int *)subid, stg_val, stg_nbr);
        TF_CHECK_LAUNCH();
        if (ws_to_host(j->hb_val.p, stg_val, (size_t)nQ * sizeof(unsigned), s) || ws_to_host(j->hb_nbr.p, stg_nbr, (size_t)nQ * nn * sizeof(int), s)) return TF_EHIP;
    }
    const int *subid_or_null = j->identity ? nullptr : subid;
    if (j->sparse) {
        long long *stg_k = (long long *)c.Rt; unsigned *stg_val = (unsigned *)c.Llo; int *stg_id = c.Lhi;
        hipLaunchKernelGGL(k_ws_small_list, dim3((unsigned)nb256), dim3(256), 0, s, (const uint8_t *)j->cls, (const int *)j->cid, j->field, N, h_vmax,
                           (const long long *)base_seed, (const long long *)base_small, subid_or_null, stg_k, stg_val, stg_id);
        TF_CHECK_LAUNCH();
        if (ws_to_host(j->hb_sk.p, stg_k, (size_t)j->S * sizeof(long long), s) || ws_to_host(j->hb_sval.p, stg_val, (size_t)j->S * sizeof(unsigned), s) ||
            ws_to_host(j->hb_sid.p, stg_id, (size_t)j->S * sizeof(int), s)) return TF_EHIP;
        TF_CHECK_HIP(hipStreamSynchronize(s));
    } else {
        // DENSE form: every seed has to be in the replay's heap (more small seeds than staging room, or TF_WS_REFERENCE_DENSE).
        TF_CHECK_HIP(hipStreamSynchronize(s));                          // the sub-graph has left the queues
        // 2 bits per seed + the exceptions (k_ws_seed_codes) if the two bitmaps fit a frontier queue ...
        const int64_t W = (j->M + 31) / 32 + 8;                         // words per bitmap
        j->coded = 2 * W <= 2 * R + 64 && !ws_env("TF_WS_REFERENCE_NO_CODES");
        if (j->coded) {
            u64 *d_sample = (u64 *)j->Q.q[1];
            hipLaunchKernelGGL(k_ws_entry_sample, dim3(1), dim3(256), 0, s, (const uint8_t *)j->cls, (const int *)j->cid, j->field, N, h_vmax, d_sample);
            TF_CHECK_LAUNCH();
            u64 h_sample[256];
            TF_CHECK_HIP(hipMemcpyAsync(h_sample, d_sample, sizeof(h_sample), hipMemcpyDeviceToHost, s));
            TF_CHECK_HIP(hipStreamSynchronize(s));
            u64 D = 0ull; int best = 0;
            for (int a = 0; a < 256; a++) {
                if (!h_sample[a]) continue;
                int cnt = 0;
                for (int b = 0; b < 256; b++) cnt += h_sample[b] == h_sample[a];
                if (cnt > best) { best = cnt; D = h_sample[a]; }
            }
            j->code_d = D;
            int *n_exc = n_small; long long *base_exc = base_small;      // (the small-seed counts have served)
            hipLaunchKernelGGL(k_ws_code_counts, dim3((unsigned)nb256), dim3(256), 0, s, (const uint8_t *)j->cls, (const int *)j->cid, j->field, N, h_vmax, D, n_exc);
            TF_CHECK_LAUNCH();
            tb = j->scan_bytes;
            TF_CHECK_HIP(tf_exclusive_sum(j->scan_tmp, tb, WsCountIter(n_exc, WsIntToLL()), base_exc, (size_t)nb256, s));
            long long h_e = 0; int h_ec = 0;
            TF_CHECK_HIP(hipMemcpyAsync(&h_e, base_exc + nb256 - 1, sizeof(long long), hipMemcpyDeviceToHost, s));
            TF_CHECK_HIP(hipMemcpyAsync(&h_ec, n_exc + nb256 - 1, sizeof(int), hipMemcpyDeviceToHost, s));
            TF_CHECK_HIP(hipStreamSynchronize(s));
            const int64_t E = h_e + h_ec;
            j->code_words = W; j->n_exc = E;
            j->hb_sval = ws_host_take((size_t)2 * W * sizeof(unsigned), true);
            j->hb_sid = ws_host_take((size_t)(E > 0 ? E : 1) * sizeof(u64), true);
            j->hb_sk = ws_host_take((size_t)(j->M + nQ + 1) * sizeof(u64), false);       // the heap itself: plain memory, filled by the replay
            if (!j->hb_sval.p || !j->hb_sid.p || !j->hb_sk.p) { tf_set_error("tf_watershed: out of host memory for the reference-order replay"); return TF_ENOMEM; }
            unsigned *d_bits = (unsigned *)j->Q.q[0];
            TF_CHECK_HIP(hipMemsetAsync(d_bits, 0, (size_t)2 * W * sizeof(unsigned), s));
            const int64_t cap = R + 32;
            bool first = true;
            for (int64_t c0 = 0; first || c0 < E; c0 += cap, first = false) {
                const int64_t cnt = E - c0 < cap ? E - c0 : cap;
                u64 *stg = (u64 *)j->Q.q[1];
                hipLaunchKernelGGL(k_ws_seed_codes, dim3((unsigned)nb256), dim3(256), 0, s, (const uint8_t *)j->cls, (const int *)j->cid, j->field, N, h_vmax, D,
                                   (const long long *)base_seed, (const long long *)base_exc, subid_or_null, first ? 1 : 0, d_bits, d_bits + W, c0, cap, stg);
                TF_CHECK_LAUNCH();
                if (cnt > 0 && ws_to_host((u64 *)j->hb_sid.p + c0, stg, (size_t)cnt * sizeof(u64), s)) return TF_EHIP;
                if (first && ws_to_host(j->hb_sval.p, d_bits, (size_t)2 * W * sizeof(unsigned), s)) return TF_EHIP;
                TF_CHECK_HIP(hipStreamSynchronize(s));                  // (the staging queue is written again by the next chunk)
            }
        } else {
            // ... else as the 8-byte heap entries themselves, through the two frontier queues in turn, R entries at a time
            j->hb_sk = ws_host_take((size_t)(j->M + nQ + 1) * sizeof(u64), true);
            if (!j->hb_sk.p) { tf_set_error("tf_watershed: out of host memory for the reference-order replay"); return TF_ENOMEM; }
            const int64_t cap = R + 32;
            int turn = 0;
            for (int64_t c0 = 0; c0 < j->M; c0 += cap, turn ^= 1) {
                const int64_t cnt = j->M - c0 < cap ? j->M - c0 : cap;
                u64 *stg = (u64 *)j->Q.q[turn];
                hipLaunchKernelGGL(k_ws_seed_list, dim3((unsigned)nb256), dim3(256), 0, s, (const uint8_t *)j->cls, (const int *)j->cid, j->field, N,
                                   (const long long *)base_seed, c0, cap, subid_or_null, h_vmax, stg);
                TF_CHECK_LAUNCH();
                if (ws_to_host((u64 *)j->hb_sk.p + c0, stg, (size_t)cnt * sizeof(u64), s)) return TF_EHIP;
                if (turn) TF_CHECK_HIP(hipStreamSynchronize(s));        // (the stream orders kernel k + 2 after copy k anyway; bound the queue of copies)
            }
            TF_CHECK_HIP(hipStreamSynchronize(s));
        }
    }
    j->need_replay = true;
    j->ms_export = ws_now_ms() - t_enter;
    if (ref_debug)
        fprintf(stderr, "reference order: %lld seeds, %lld at or below the tie value (key %u), %lld relevant pixels, sub-graph %lld%s, %s form; "
                "tie value + numbering %.1f ms, export %.1f ms\n", (long long)j->M, (long long)j->S, h_vmax, (long long)R, (long long)nQ,
                j->identity ? " (whole compact set)" : "", j->sparse ? "sparse" : (j->coded ? "dense (2 bits per seed + exceptions)" : "dense"), t_numbered - t_enter, ws_now_ms() - t_numbered);
    if (ref_debug && !j->sparse && j->coded)
        fprintf(stderr, "reference order: modal ballast entry %016llx, %lld exceptions\n", (unsigned long long)j->code_d, (long long)j->n_exc);
    return TF_OK;
}

// chain levels C_{levels_done + 1} .. C_{depth - 1} (final once computed)
static int ws_chain_levels(tf_ws_job *j)
{
    const unsigned nbr_blocks = (unsigned)((j->R + 255) / 256);
    for (int k = j->levels_done + 1; k < j->depth; k++) {
        hipLaunchKernelGGL(k_ws_init_level, dim3(nbr_blocks), dim3(256), 0, j->s, j->c.pix, j->c.C[k], j->R, 0, (const int *)nullptr);
        TF_CHECK_LAUNCH();
        int64_t sw = 0;
        const int rc = ws_run_phase(j->c, k, j->depth_max + 1, j->Q, j->s, j->max_sweeps, &sw);      // k < "depth": a chain level
        if (rc) return rc;
        j->st[2 + (k < 3 ? k - 1 : 2)] += sw;
        j->levels_done = k;
    }
    return TF_OK;
}

// the root phase at the job's depth (root keys: raster indices, or pop ranks from rank_dev) + the exactness check -> h_amb
static int ws_root_and_check(tf_ws_job *j, bool ranked)
{
    hipStream_t s = j->s;
    const WsC &c = j->c;
    const int64_t R = j->R;
    const unsigned nbr_blocks = (unsigned)((R + 255) / 256);
    hipLaunchKernelGGL(k_ws_init_level, dim3(nbr_blocks), dim3(256), 0, s, c.pix, c.Rt, R, 1, ranked ? (const int *)j->rank_dev : (const int *)nullptr);
    hipLaunchKernelGGL(k_ws_init_labelset, dim3(nbr_blocks), dim3(256), 0, s, c.pix, j->markers, c.Llo, c.Lhi, R);
    TF_CHECK_LAUNCH();
    int64_t sw = 0;
    const int rc = ws_run_phase(c, j->depth, j->depth, j->Q, s, j->max_sweeps, &sw);              // k == depth: the root phase
    if (rc) return rc;
    if (j->speculate_fast && j->depth == 1) j->st[1] = sw; else j->st[2 + (j->depth < 3 ? j->depth - 1 : 2)] += sw;
    j->st[12] += 1;
    TF_CHECK_HIP(hipMemsetAsync(j->org, 0, (size_t)R * sizeof(int), s));
    TF_CHECK_HIP(hipMemsetAsync(j->d_cnt, 0, 4 * sizeof(unsigned long long), s));
    hipLaunchKernelGGL(k_ws_origins, dim3(nbr_blocks), dim3(256), 0, s, c, j->depth, j->org);
    hipLaunchKernelGGL(k_ws_count_ambiguous, dim3(nbr_blocks), dim3(256), 0, s, c, j->org, j->d_cnt);
    TF_CHECK_LAUNCH();
    TF_CHECK_HIP(hipMemcpyAsync(j->h_amb, j->d_cnt, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    TF_CHECK_HIP(hipStreamSynchronize(s));
    return TF_OK;
}

// begin, part 3: phase A, the edge masks, the speculative root phase or the chain levels below the job's depth.  Part of
// tf_watershed_begin unless the caller asked for TF_WS_DEFER_SWEEPS: it then runs them through tf_watershed_sweeps (or leaves
// them to finish) -- none of it reads the flow fields, the seeds' set-up of the NEXT window can run first, and the host replay
// of an export on a guessed tie value is under way meanwhile.
static int ws_job_sweeps(tf_ws_job *j)
{
    if (!j->sweeps_pending) return TF_OK;
    j->sweeps_pending = false;
    int64_t *st = j->st;
    const WsC &c = j->c;
    hipStream_t s = j->s;
    const unsigned nbr_blocks = (unsigned)((j->R + 255) / 256);
    int rc = ws_run_phase(c, 0, j->depth_max, j->Q, s, j->max_sweeps, &st[0], 29.0 * (double)j->N);
    if (rc) return rc;
    {
        TfProfScope ps(TFK_WS_SETUP, 0.0, s);
        hipLaunchKernelGGL(k_ws_edge_masks, dim3(nbr_blocks), dim3(256), 0, s, c, (u64 *)c.emask);
    }
    TF_CHECK_LAUNCH();
    // Speculative start: a root phase on K2 alone (depth 1).  If its exactness check finds no origin at all the
    // labelling cannot depend on any tie-break and is final (tie-free fields).  The caller's hint skips it for
    // inputs known to contain exact plateaus: the labels are the same either way, only the work differs.
    j->speculate_fast = !((j->flags & TF_WS_SKIP_FAST_PATH) && j->depth0 > 1);
    st[5] = j->speculate_fast ? 0 : -1;
    j->depth = j->depth0;
    j->root_pending = true;
    if (j->speculate_fast) {
        j->depth = 1;
        rc = ws_root_and_check(j, false);
        if (rc) return rc;
        st[5] = j->h_amb[2] != 0;
        if (j->h_amb[2] == 0 || j->depth_max <= 1) j->root_pending = false;        // final (at depth 1 every origin counts as cut off)
        else j->depth = j->depth0;
    }
    if (j->root_pending) {
        rc = ws_chain_levels(j);                                                    // C_1 .. C_{depth - 1}; the root phase is finish's
        if (rc) return rc;
    }
    st[8] = j->depth; st[9] = (int64_t)j->h_amb[0]; st[10] = (int64_t)j->h_amb[1]; st[11] = (int64_t)j->h_amb[2];
    return TF_OK;
}

// `rv` != nullptr: the raveled form (tf_watershed_raveled): `field` = image, `markers` = `labels` = output (in place),
// seeds = rv_locs; T, H, W, fwd, bwd, nbr_host unused.
static int ws_job_begin(tf_ws_job *j, const float *field, const int32_t *markers, const int8_t *mask,
                        const float *fwd, const float *bwd, int64_t T, int64_t H, int64_t W,
                        const int8_t *nbr_host, int n_nbr, int depth0, int depth_max, int flags,
                        void *ws, size_t ws_bytes, void *stream,
                        const WsRavel *rv = nullptr, const int64_t *rv_locs = nullptr, int64_t rv_n_locs = 0, int64_t spec_key = -1)
{
    int64_t *st = j->st;
    for (int i = 0; i < TF_WS_NSTATS; i++) st[i] = 0;
    TF_REQUIRE((flags & ~(TF_WS_SKIP_FAST_PATH | TF_WS_REFERENCE_ORDER | TF_WS_DEFER_SWEEPS)) == 0, "tf_watershed: unknown flag");
    TF_REQUIRE(field && markers && ws, "tf_watershed: null pointer");
    if (!rv) {
        TF_REQUIRE(fwd && bwd && nbr_host, "tf_watershed: null pointer");
        TF_REQUIRE(T > 0 && H > 0 && W > 0 && H < (1 << 15) && W < (1 << 15) && T < 65536, "tf_watershed: bad shape");
    }
    TF_REQUIRE(n_nbr > 0 && n_nbr <= WS_MAX_NBR, "tf_watershed: bad neighbour count");
    TF_REQUIRE(depth0 >= 1 && depth0 <= depth_max && depth_max <= WS_MAX_DEPTH, "tf_watershed: bad chain_depth");
    const int64_t N = rv ? rv->n : T * H * W;
    // the raveled twin addresses voxels through int32 strides like the reference; the (T, H, W) entry points index in
    // 64 bits and are bounded by memory (and by 2^30 RELEVANT pixels, checked after the scan)
    TF_REQUIRE(N > 0 && (rv ? N <= 0x7fffffffll : N <= (1ll << 36)), "tf_watershed: too many voxels per call (use time windows)");
    const int64_t NV = rv ? N : ws_virtual_voxels(T, H, W);           // scan positions (tile order pads edge tiles)
    if (ws_bytes < ws_full_bytes(N, NV) + ws_compact_bytes(1, n_nbr, depth_max)) { tf_set_error("tf_watershed: workspace too small"); return TF_ENOMEM; }
    hipStream_t s = (hipStream_t)stream;
    WsGeom &g = j->g;
    g.T = T; g.H = (int)H; g.W = (int)W; g.plane = H * W; g.n_nbr = n_nbr; g.n_tx = (int)((W + WS_TILE - 1) / WS_TILE);
    for (int i = 0; i < n_nbr && !rv; i++) {
        g.dt[i] = nbr_host[i * 3]; g.dy[i] = nbr_host[i * 3 + 1]; g.dx[i] = nbr_host[i * 3 + 2];
        TF_REQUIRE(abs(g.dt[i]) <= 1 && abs(g.dy[i]) <= 1 && abs(g.dx[i]) <= 1, "tf_watershed: neighbour offset out of range");
    }
    j->field = field; j->markers = markers; j->mask = mask; j->T = T; j->H = H; j->W = W; j->N = N; j->NV = NV;
    j->n_nbr = n_nbr; j->depth0 = depth0; j->depth_max = depth_max; j->flags = flags; j->s = s; j->raveled = rv != nullptr;
    TfArena ar(ws, ws_bytes);
    uint8_t *cls = ar.take<uint8_t>(N), *flag = ar.take<uint8_t>(NV);
    int *scan = ar.take<int>(NV), *cid = ar.take<int>(N);
    const size_t scan_bytes = ws_scan_temp_bytes(NV);
    char *scan_tmp = ar.take<char>(scan_bytes ? scan_bytes : 1);
    int *d_flags = ar.take<int>(WS_BATCH + 8);
    unsigned long long *d_cnt = ar.take<unsigned long long>(4);
    j->seed_blocks = ar.take<char>((size_t)((N + 255) / 256) * 24 + 64);
    if (!ar.ok()) { tf_set_error("tf_watershed: workspace too small"); return TF_ENOMEM; }
    j->cls = cls; j->flag = flag; j->scan = scan; j->cid = cid; j->scan_tmp = scan_tmp; j->scan_bytes = scan_bytes;
    j->d_flags = d_flags; j->d_cnt = d_cnt;

    dim3 block(64, 4, 1), grid(rv ? 1 : (g.W + 63) / 64, rv ? 1 : (g.H + 3) / 4, rv ? 1 : (unsigned)T);
    const unsigned nb1 = (unsigned)((N + 255) / 256);
    int64_t R = 0;
    bool tiles = false;
    {
        TfProfScope ps(TFK_WS_SETUP, 29.0 * (double)N, s);
        if (rv) {
            // `flag` doubles as the seed bitmap until the relevance pass overwrites it (cls is final by then)
            int *d_bad = d_flags + WS_BATCH + 5;
            TF_CHECK_HIP(hipMemsetAsync(d_bad, 0, sizeof(int), s));
            TF_CHECK_HIP(hipMemsetAsync(flag, 0, (size_t)N, s));
            if (rv_n_locs > 0)
                hipLaunchKernelGGL(k_wsr_seeds, dim3((unsigned)((rv_n_locs + 255) / 256)), dim3(256), 0, s, rv_locs, rv_n_locs, N, flag, d_bad);
            hipLaunchKernelGGL(k_wsr_classify, dim3(nb1), dim3(256), 0, s, (const uint8_t *)flag, markers, mask, N, cls, d_bad);
            TF_CHECK_LAUNCH();
            int h_bad = 0;
            TF_CHECK_HIP(hipMemcpyAsync(&h_bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, s));
            TF_CHECK_HIP(hipStreamSynchronize(s));
            TF_REQUIRE(h_bad != 1, "tf_watershed_raveled: marker_locations must be strictly ascending indices into the arrays (np.flatnonzero order)");
            TF_REQUIRE(h_bad != 2, "tf_watershed_raveled: output must be non-zero at every marker location (_watershed.pyx:241-244)");
            hipLaunchKernelGGL(k_wsr_relevant, dim3(nb1), dim3(256), 0, s, (const uint8_t *)cls, *rv, flag);
        } else {
            if ((uintptr_t)markers % 16 == 0 && (uintptr_t)cls % 4 == 0 && (!mask || (uintptr_t)mask % 4 == 0))
                hipLaunchKernelGGL(k_ws_classify4, dim3((unsigned)(((N + 3) / 4 + 255) / 256)), dim3(256), 0, s, markers, mask, N, cls);
            else hipLaunchKernelGGL(k_ws_classify, dim3(nb1), dim3(256), 0, s, markers, mask, N, cls);
            if (NV > N) TF_CHECK_HIP(hipMemsetAsync(flag, 0, (size_t)NV, s));   // padding positions of the edge tiles
            bool faces = n_nbr == 6;                        // the six face neighbours, in any order?
            for (int i = 0; faces && i < 6; i++) faces = abs(g.dt[i]) + abs(g.dy[i]) + abs(g.dx[i]) == 1;
            for (int i = 0; faces && i < 6; i++) for (int k = 0; k < i; k++) faces = faces && !(g.dt[i] == g.dt[k] && g.dy[i] == g.dy[k] && g.dx[i] == g.dx[k]);
            if (faces && W % 4 == 0 && (uintptr_t)cls % 4 == 0 && (uintptr_t)flag % 4 == 0 && (uintptr_t)fwd % 16 == 0 && (uintptr_t)bwd % 16 == 0)
                hipLaunchKernelGGL(k_ws_relevant6x4, dim3((unsigned)((W / 4 + 63) / 64), (unsigned)((H + 3) / 4), (unsigned)T), block, 0, s,
                                   (const uint8_t *)cls, fwd, bwd, g, flag);
            else if (n_nbr == 6) hipLaunchKernelGGL(k_ws_relevant<6>, grid, block, 0, s, cls, fwd, bwd, g, flag);
            else if (n_nbr == 18) hipLaunchKernelGGL(k_ws_relevant<18>, grid, block, 0, s, cls, fwd, bwd, g, flag);
            else if (n_nbr == 26) hipLaunchKernelGGL(k_ws_relevant<26>, grid, block, 0, s, cls, fwd, bwd, g, flag);
            else hipLaunchKernelGGL(k_ws_relevant<0>, grid, block, 0, s, cls, fwd, bwd, g, flag);
        }
        TF_CHECK_LAUNCH();
        // tiled ids: per-tile counts + a scan of the counts (k_ws_cid_tiles adds the prefix inside a tile); the `scan` array
        // (NV ints) holds the 64-bit counts and bases (2 x NV / 256 x 8 bytes)
        const int64_t n_tiles = NV / 256;
        size_t tile_scan_bytes = 0;
        if (!rv) (void)tf_exclusive_sum(nullptr, tile_scan_bytes, (const long long *)nullptr, (long long *)nullptr, (size_t)n_tiles);
        tiles = !rv && NV % 256 == 0 && n_tiles > 0 && n_tiles < 0x7fffffffll && tile_scan_bytes <= scan_bytes;
        if (tiles) {
            long long *t_count = (long long *)scan, *t_base = t_count + n_tiles;
            hipLaunchKernelGGL(k_ws_count_tiles, dim3((unsigned)((n_tiles + 3) / 4)), dim3(256), 0, s, (const uint8_t *)flag, n_tiles, t_count);
            size_t tb = scan_bytes;
            TF_CHECK_HIP(tf_exclusive_sum(scan_tmp, tb, (const long long *)t_count, t_base, (size_t)n_tiles, s));
            long long h_last[2] = {0, 0};
            TF_CHECK_HIP(hipMemcpyAsync(&h_last[0], t_base + n_tiles - 1, sizeof(long long), hipMemcpyDeviceToHost, s));
            TF_CHECK_HIP(hipMemcpyAsync(&h_last[1], t_count + n_tiles - 1, sizeof(long long), hipMemcpyDeviceToHost, s));
            TF_CHECK_HIP(hipStreamSynchronize(s));
            R = h_last[0] + h_last[1];
            TF_REQUIRE(R <= 0x3fffff00ll, "tf_watershed: more than 2^30 relevant pixels in one call (use time windows)");
        } else {
            const int rc_scan = ws_scan_flags(flag, scan, NV, scan_tmp, scan_bytes, s, &R);
            if (rc_scan) return rc_scan;
        }
    }
    st[6] = R;
    j->R = R;
    if (ws_bytes < ws_full_bytes(N, NV) + ws_compact_bytes(R > 0 ? R : 1, n_nbr, depth_max)) {
        tf_set_error("tf_watershed: workspace too small for %lld relevant pixels", (long long)R);
        return TF_ENOMEM;
    }
    if (rv) hipLaunchKernelGGL(k_ws_cid, dim3(nb1), dim3(256), 0, s, cls, flag, scan, N, cid);
    else if (tiles) hipLaunchKernelGGL(k_ws_cid_tiles, dim3((unsigned)((NV / 256 + 3) / 4)), dim3(256), 0, s, (const uint8_t *)cls, (const uint8_t *)flag,
                                       (const long long *)((long long *)scan + NV / 256), NV / 256, g, cid);
    else hipLaunchKernelGGL(k_ws_cid_tiled, grid, block, 0, s, (const uint8_t *)cls, (const uint8_t *)flag, (const int *)scan, g, cid);
    TF_CHECK_LAUNCH();
    WsC &c = j->c; memset(&c, 0, sizeof(c));
    c.R = R; c.n_nbr = n_nbr;
    j->org = nullptr; j->rank_dev = nullptr;
    j->depth = 0;
    if (R > 0) {
        u64 *pix = ar.take<u64>(R); unsigned *val = ar.take<unsigned>(R); int *nbr = ar.take<int>(R * n_nbr);
        c.pix = pix; c.val = val; c.nbr = nbr;
        c.KM = ar.take<u64>(2 * R);
        for (int k = 1; k < depth_max; k++) c.C[k] = ar.take<u64>(R);
        c.Rt = ar.take<u64>(R);
        c.Llo = ar.take<int>(R); c.Lhi = ar.take<int>(R); j->org = ar.take<int>(R);
        j->rank_dev = ar.take<int>(R);
        u64 *emask = ar.take<u64>(R);
        c.emask = emask;
        WsQueues &Q = j->Q;
        Q.qcap = (int)(2 * R < 0x7fffff00ll ? 2 * R : 0x7fffff00ll);
        Q.q[0] = ar.take<int>(2 * R + 64); Q.q[1] = ar.take<int>(2 * R + 64); Q.inq = ar.take<int>(R);
        j->subid = ar.take<int>(R);
        Q.cnt = d_flags; Q.h_cnt = j->h_cnt; Q.processed = &st[7];
        if (!ar.ok()) { tf_set_error("tf_watershed: workspace too small"); return TF_ENOMEM; }
        int *d_nan = d_flags + WS_BATCH + 4;
        TF_CHECK_HIP(hipMemsetAsync(d_nan, 0, sizeof(int), s));
        {
            TfProfScope ps(TFK_WS_SETUP, 0.0, s);
            if (rv) hipLaunchKernelGGL(k_wsr_compact, dim3(nb1), dim3(256), 0, s, field, (const int *)cid, *rv, pix, val, nbr, c.KM, d_nan);
            else if (n_nbr == 6) hipLaunchKernelGGL(k_ws_compact<6>, grid, block, 0, s, field, fwd, bwd, cid, g, pix, val, nbr, c.KM, d_nan);
            else if (n_nbr == 18) hipLaunchKernelGGL(k_ws_compact<18>, grid, block, 0, s, field, fwd, bwd, cid, g, pix, val, nbr, c.KM, d_nan);
            else if (n_nbr == 26) hipLaunchKernelGGL(k_ws_compact<26>, grid, block, 0, s, field, fwd, bwd, cid, g, pix, val, nbr, c.KM, d_nan);
            else hipLaunchKernelGGL(k_ws_compact<0>, grid, block, 0, s, field, fwd, bwd, cid, g, pix, val, nbr, c.KM, d_nan);
        }
        TF_CHECK_LAUNCH();
        {   // the reference's `smaller()` (_watershed.pyx:161-164) is not an order on NaN: its heap then pops in an order
            // that depends on the array state, and no closed form exists.  Refuse instead of returning something else.
            int h_nan = 0;
            TF_CHECK_HIP(hipMemcpyAsync(&h_nan, d_nan, sizeof(int), hipMemcpyDeviceToHost, s));
            TF_CHECK_HIP(hipStreamSynchronize(s));
            if (h_nan) {
                tf_set_error("tf_watershed: the field is NaN at a floodable pixel or at a seed next to one; the pop order of the "
                             "reference's heap is undefined for NaN keys (detection.get_combined_edge_field maps NaN to +inf)");
                return TF_EINVAL;
            }
        }
        const unsigned nbr_blocks = (unsigned)((R + 255) / 256);
        // a front advances at least one pixel per sweep, so a phase needs at most "longest flood path" sweeps: bounded by
        // T + H + W times a detour factor on a (T, H, W) grid; the raveled form knows no shape, there the only safe bound
        // is the number of relevant pixels itself (a snake-shaped mask floods one pixel per sweep)
        const int64_t max_sweeps = rv ? 4096 + 2 * R : 4096 + 512 * (T + H + W);
        j->max_sweeps = max_sweeps;
        if ((flags & TF_WS_REFERENCE_ORDER) && spec_key >= 0) {
            // SPECULATIVE EXPORT (round 4).  The replay needs nothing the relaxation phases compute -- seeds, keys and the
            // neighbour table exist now -- except the tie value, and any value at or above the true one gives the ranks of
            // every marker that can matter.  With the caller's guess (the previous window's tie value) the export happens
            // here, the host replay runs beside phase A and the chain phases, and finish runs ONE root phase, with the pop
            // ranks; it checks the guess against the tie value it then finds and falls back on the export-after-the-root-phase
            // order if the guess was too low.
            j->vmax = (unsigned)spec_key;
            j->speculative = true;
            const int rc_x = ws_job_export(j);
            if (rc_x) return rc_x;
        }
        (void)nbr_blocks;
        j->sweeps_pending = true;
        if (flags & TF_WS_DEFER_SWEEPS) return TF_OK;                               // phase A and the chain levels: tf_watershed_sweeps / finish
        return ws_job_sweeps(j);
    }
    st[8] = j->depth; st[9] = (int64_t)j->h_amb[0]; st[10] = (int64_t)j->h_amb[1]; st[11] = (int64_t)j->h_amb[2];
    return TF_OK;
}

// host only: no HIP call, any thread
static int ws_job_replay(tf_ws_job *j)
{
    if (!j->need_replay || j->replay_done) return j->replay_rc;
    const double t0 = ws_now_ms();
    int n_ranked = 0;
    double phase[2] = {0.0, 0.0};
    if (j->sparse)
        j->popped = ws_reference_ranks_sparse(j->M, j->S, (const long long *)j->hb_sk.p, (const unsigned *)j->hb_sval.p, (const int *)j->hb_sid.p,
                                              j->nQ, (const unsigned *)j->hb_val.p, (const int *)j->hb_nbr.p, j->n_nbr, j->vmax, (int *)j->hb_rank.p, &n_ranked, phase);
    else {
        const WsSeedCodes codes{(const uint32_t *)j->hb_sval.p, (const uint32_t *)j->hb_sval.p + (j->coded ? j->code_words : 0), j->code_d, (const u64 *)j->hb_sid.p};
        if (j->plain) {
            if (j->coded) wsr_expand(j->M, codes, (u64 *)j->hb_sk.p);
            j->popped = ws_reference_ranks_plain(j->M, (const u64 *)j->hb_sk.p, nullptr, j->nQ, (const unsigned *)j->hb_val.p,
                                                 (const int *)j->hb_nbr.p, j->n_nbr, j->vmax, (int *)j->hb_rank.p, &n_ranked);
        } else
            j->popped = ws_reference_ranks_dense(j->M, (u64 *)j->hb_sk.p, j->nQ, (const unsigned *)j->hb_val.p, (const int *)j->hb_nbr.p,
                                                 j->n_nbr, j->vmax, (int *)j->hb_rank.p, &n_ranked, phase, j->coded ? &codes : nullptr);
    }
    j->n_ranked = n_ranked;
    j->replay_done = true;
    j->ms_replay = ws_now_ms() - t0;
    if (j->popped < 0) { tf_set_error("tf_watershed: out of host memory for the reference-order replay"); j->replay_rc = TF_ENOMEM; }
    static const bool ref_debug = ws_env("TF_WS_REF_DEBUG");
    if (ref_debug) fprintf(stderr, "reference order: %s%s replay %.1f ms (%lld pops, %d markers ranked; heap build %.1f ms, pops %.1f ms)\n", j->speculative ? "speculative " : "",
                           j->sparse ? "sparse" : (j->plain ? "plain" : "dense"), j->ms_replay, (long long)j->popped, n_ranked, phase[0], phase[1]);
    return j->replay_rc;
}

// pop ranks of the finished replay -> rank_dev (one entry per relevant pixel)
static int ws_upload_ranks(tf_ws_job *j)
{
    hipStream_t s = j->s;
    const unsigned nbr_blocks = (unsigned)((j->R + 255) / 256);
    int *stg_rank = j->Q.q[0];
    TF_CHECK_HIP(hipMemcpyAsync(stg_rank, j->hb_rank.p, (size_t)j->nQ * sizeof(int), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_ws_scatter_ranks, dim3(nbr_blocks), dim3(256), 0, s, j->identity ? (const uint8_t *)nullptr : (const uint8_t *)j->scan,
                       (const int *)j->subid, (const int *)stg_rank, j->n_ranked, j->R, j->rank_dev);
    TF_CHECK_LAUNCH();
    TF_CHECK_HIP(hipStreamSynchronize(s));                              // (the staging queue is a frontier queue again right after)
    return TF_OK;
}

static int ws_job_finish(tf_ws_job *j, int32_t *labels, uint8_t *amb_out)
{
    TF_REQUIRE(labels, "tf_watershed: null pointer");
    hipStream_t s = j->s;
    int64_t *st = j->st;
    const WsC &c = j->c;
    const int64_t R = j->R, N = j->N;
    const unsigned nb1 = (unsigned)((N + 255) / 256);
    static const bool ref_debug = ws_env("TF_WS_REF_DEBUG");
    bool ranked = j->ranked;                                             // root keys are (pop rank << 32) | compact id
    if (R > 0 && j->sweeps_pending) {                                    // TF_WS_DEFER_SWEEPS and nobody has called tf_watershed_sweeps
        const int rc_s = ws_job_sweeps(j);
        if (rc_s) return rc_s;
    }
    if (R > 0 && j->late_export) {
        // second entry: the replay of the export that followed the root phase has run (or runs here): ranks up, root phase again
        j->late_export = false;
        int rc = ws_job_replay(j);
        if (rc) return rc;
        const double t0 = ws_now_ms();
        rc = ws_upload_ranks(j);
        if (rc) return rc;
        ranked = j->ranked = true;
        rc = ws_root_and_check(j, true);                                 // (the check's counts do not depend on the order)
        if (rc) return rc;
        j->applied = true;
        j->ms_detour += j->ms_export + j->ms_replay + (ws_now_ms() - t0);
        st[13] = j->popped; st[14] = j->sparse ? j->S : j->M;
        st[15] = (int64_t)(j->ms_detour * 1000.0);
    } else if (R > 0 && j->root_pending) {
        j->root_pending = false;
        const unsigned nbr_blocks = (unsigned)((R + 255) / 256);
        int rc;
        double &ms_detour = j->ms_detour;
        if (j->need_replay) {                                            // the speculative export's replay: ranks before the root phase
            rc = ws_job_replay(j);                                       // (a no-op if the caller has run it)
            if (rc) return rc;
            const double t0 = ws_now_ms();
            rc = ws_upload_ranks(j);
            if (rc) return rc;
            ranked = j->ranked = true;
            ms_detour = j->ms_export + j->ms_replay + (ws_now_ms() - t0);
        } else if (j->device_ranks) {                                    // the speculative export found one value class: rank_dev is filled
            ranked = j->ranked = true;
            ms_detour = j->ms_export;
        }
        for (;;) {
            rc = ws_root_and_check(j, ranked);
            if (rc) return rc;
            if (j->h_amb[2] == 0 || j->depth >= j->depth_max) break;
            j->depth += 1;                                               // ties left by the cut-off: one more chain level
            rc = ws_chain_levels(j);
            if (rc) return rc;
        }
        if ((j->flags & TF_WS_REFERENCE_ORDER) && j->h_amb[1] > 0 && j->h_amb[2] == 0) {
            // Labels hang on the order of equal-valued markers.  The largest marker value at which they do:
            unsigned *d_vmax = (unsigned *)(j->d_cnt + 3);
            TF_CHECK_HIP(hipMemsetAsync(d_vmax, 0, sizeof(unsigned), s));
            hipLaunchKernelGGL(k_ws_tie_value_max, dim3(nbr_blocks), dim3(256), 0, s, c, (const int *)j->org, j->field, ranked ? 1 : 0, d_vmax);
            TF_CHECK_LAUNCH();
            unsigned h_vmax = 0;
            TF_CHECK_HIP(hipMemcpyAsync(&h_vmax, d_vmax, sizeof(unsigned), hipMemcpyDeviceToHost, s));
            TF_CHECK_HIP(hipStreamSynchronize(s));
            j->true_vmax = h_vmax; j->has_tie = true;
            if (ranked && h_vmax <= j->vmax) {
                j->applied = true;                                       // the guess covered it: these ARE the labels
                j->spec_hit = true;
            } else {
                // the host replay (ws_reference_ranks*) gives the reference's order, and the root phase is repeated with the
                // pop rank in place of the raster index.  Chain levels, origins and label sets do not depend on that order,
                // only the choice among tying candidates does.
                if (ref_debug && ranked) fprintf(stderr, "reference order: the guessed tie value (key %u) was too low (true key %u): export again\n", j->vmax, h_vmax);
                ws_host_give(j->hb_val); ws_host_give(j->hb_nbr); ws_host_give(j->hb_rank);
                ws_host_give(j->hb_sk); ws_host_give(j->hb_sval); ws_host_give(j->hb_sid);
                j->need_replay = false; j->replay_done = false; j->speculative = false; j->replay_rc = TF_OK;
                j->vmax = h_vmax;
                rc = ws_job_export(j);
                if (rc) return rc;
                if (j->device_ranks) {                                   // closed form on the device: no host pass, root phase again at once
                    ranked = j->ranked = true;
                    rc = ws_root_and_check(j, true);                     // (the check's counts do not depend on the order)
                    if (rc) return rc;
                    j->applied = true;
                    ms_detour += j->ms_export;
                    st[13] = j->popped; st[14] = j->S; st[15] = (int64_t)(ms_detour * 1000.0);
                    goto labels_out;
                }
                // the replay is the caller's to run (any thread); finish is entered again afterwards
                j->late_export = true;
                st[8] = j->depth; st[9] = (int64_t)j->h_amb[0]; st[10] = (int64_t)j->h_amb[1]; st[11] = (int64_t)j->h_amb[2];
                return TF_WS_REPLAY_PENDING;
            }
        }
        if (j->need_replay && j->replay_done) {
            st[13] = j->popped; st[14] = j->sparse ? j->S : j->M;
            st[15] = (int64_t)(ms_detour * 1000.0);
        } else if (j->device_ranks) {
            st[13] = j->popped; st[14] = j->S; st[15] = (int64_t)(ms_detour * 1000.0);
        }
    }
labels_out:
    st[8] = j->depth; st[9] = (int64_t)j->h_amb[0]; st[10] = (int64_t)j->h_amb[1]; st[11] = (int64_t)j->h_amb[2];
    {
        TfProfScope ps(TFK_WS_LABELS, 12.0 * (double)N, s);
        if (j->raveled) { if (R > 0) hipLaunchKernelGGL(k_wsr_labels, dim3(nb1), dim3(256), 0, s, (const int *)j->cid, (const u64 *)c.Rt, c.pix, ranked ? 1 : 0, labels, N); }
        else hipLaunchKernelGGL(k_ws_labels, dim3(nb1), dim3(256), 0, s, j->markers, j->cid, c.Rt, c.pix, ranked ? 1 : 0, c.Llo, c.Lhi, j->org, labels,
                                R > 0 ? amb_out : nullptr, N);
    }
    TF_CHECK_LAUNCH();
    if (R == 0 && amb_out) TF_CHECK_HIP(hipMemsetAsync(amb_out, 0, (size_t)N, s));
    TF_CHECK_HIP(hipStreamSynchronize(s));
    if (st[11] > 0) {
        tf_set_error("tf_watershed: %lld pixel(s) still tie at chain depth %d (the compared chains are cut off); "
                     "labels written, but they may differ from the reference there: raise max_depth",
                     (long long)st[11], j->depth);
        return TF_EDEPTH;
    }
    // reference order applied: the labels ARE the reference's, whatever its heap did with the equal-valued markers
    return (st[9] > 0 && !j->applied) ? TF_WS_AMBIGUOUS : TF_OK;
}

static tf_ws_job *ws_job_new()
{
    tf_ws_job *j = new (std::nothrow) tf_ws_job;
    if (!j) return nullptr;
    memset((void *)j, 0, sizeof(*j));
    return j;
}

static int ws_run(const float *field, const int32_t *markers, const int8_t *mask,
                  const float *fwd, const float *bwd, int64_t T, int64_t H, int64_t W,
                  const int8_t *nbr_host, int n_nbr, int depth0, int depth_max, int flags, int32_t *labels,
                  uint8_t *amb_out, void *ws, size_t ws_bytes, int64_t *st, void *stream,
                  const WsRavel *rv = nullptr, const int64_t *rv_locs = nullptr, int64_t rv_n_locs = 0)
{
    TF_REQUIRE(!(flags & TF_WS_DEFER_SWEEPS), "tf_watershed: unknown flag (TF_WS_DEFER_SWEEPS is tf_watershed_begin's)");
    for (int i = 0; i < TF_WS_NSTATS; i++) st[i] = 0;
    TF_REQUIRE(labels, "tf_watershed: null pointer");
    tf_ws_job *j = ws_job_new();
    if (!j) { tf_set_error("tf_watershed: out of host memory"); return TF_ENOMEM; }
    int rc = ws_job_begin(j, field, markers, mask, fwd, bwd, T, H, W, nbr_host, n_nbr, depth0, depth_max, flags, ws, ws_bytes, stream, rv, rv_locs, rv_n_locs);
    if (rc == TF_OK) {
        rc = ws_job_finish(j, labels, amb_out);
        if (rc == TF_WS_REPLAY_PENDING) rc = ws_job_finish(j, labels, amb_out);     // (runs the replay itself)
    }
    for (int i = 0; i < TF_WS_NSTATS; i++) st[i] = j->st[i];
    ws_job_free(j);
    return rc;
}

extern "C" int tf_watershed_begin(const float *field, const int32_t *markers, const int8_t *mask,
                                  const float *fwd, const float *bwd, int64_t T, int64_t H, int64_t W,
                                  const int8_t *nbr_host, int n_nbr, int chain_depth, int max_depth, int flags,
                                  int64_t guessed_tie_key, void *ws, size_t ws_bytes, int64_t *stats_host, void *stream, void **job_out)
{
    TF_REQUIRE(job_out, "tf_watershed_begin: null pointer");
    *job_out = nullptr;
    TF_REQUIRE(guessed_tie_key >= -1 && guessed_tie_key <= 0xFFFFFFFFll, "tf_watershed_begin: guessed_tie_key must be -1 or a 32-bit ordered key");
    tf_ws_job *j = ws_job_new();
    if (!j) { tf_set_error("tf_watershed: out of host memory"); return TF_ENOMEM; }
    const int rc = ws_job_begin(j, field, markers, mask, fwd, bwd, T, H, W, nbr_host, n_nbr, chain_depth, max_depth, flags, ws, ws_bytes, stream,
                                nullptr, nullptr, 0, guessed_tie_key);
    if (stats_host) for (int i = 0; i < TF_WS_NSTATS; i++) stats_host[i] = j->st[i];
    if (rc != TF_OK) { ws_job_free(j); return rc; }
    *job_out = j;
    return TF_OK;
}
extern "C" int tf_watershed_needs_replay(const void *job) { return job && ((const tf_ws_job *)job)->need_replay ? 1 : 0; }
extern "C" int tf_watershed_sweeps(void *job, int64_t *stats_host)
{
    TF_REQUIRE(job, "tf_watershed_sweeps: null job");
    tf_ws_job *j = (tf_ws_job *)job;
    const int rc = j->R > 0 ? ws_job_sweeps(j) : TF_OK;
    if (stats_host) for (int i = 0; i < TF_WS_NSTATS; i++) stats_host[i] = j->st[i];
    return rc;
}
extern "C" int tf_watershed_replay(void *job)
{
    TF_REQUIRE(job, "tf_watershed_replay: null job");
    return ws_job_replay((tf_ws_job *)job);
}
extern "C" int tf_watershed_job_info(const void *job, int64_t *info);
extern "C" int tf_watershed_finish(void *job, int32_t *labels, uint8_t *ambiguous, int64_t *stats_host, int64_t *info_host)
{
    TF_REQUIRE(job, "tf_watershed_finish: null job");
    tf_ws_job *j = (tf_ws_job *)job;
    const int rc = ws_job_finish(j, labels, ambiguous);
    if (stats_host) for (int i = 0; i < TF_WS_NSTATS; i++) stats_host[i] = j->st[i];
    if (info_host) (void)tf_watershed_job_info(j, info_host);
    if (rc == TF_WS_REPLAY_PENDING) return rc;                            // the job lives on: replay, then finish again
    ws_job_free(j);
    return rc;
}
extern "C" void tf_watershed_abandon(void *job) { ws_job_free((tf_ws_job *)job); }
extern "C" int tf_watershed_set_stream(void *job, void *stream)
{
    TF_REQUIRE(job, "tf_watershed_set_stream: null job");
    ((tf_ws_job *)job)->s = (hipStream_t)stream;
    return TF_OK;
}
extern "C" int tf_watershed_job_info(const void *job, int64_t *info)
{
    TF_REQUIRE(job && info, "tf_watershed_job_info: null pointer");
    const tf_ws_job *j = (const tf_ws_job *)job;
    info[0] = j->need_replay ? (j->sparse ? 1 : 2) : (j->device_ranks ? 3 : 0);
    info[1] = j->M; info[2] = j->S; info[3] = j->nQ; info[4] = j->R;
    info[5] = (int64_t)(j->ms_export * 1000.0); info[6] = j->replay_done ? (int64_t)(j->ms_replay * 1000.0) : -1;
    info[7] = j->has_tie ? (int64_t)j->true_vmax : -1;
    info[8] = j->speculative ? 1 : 0; info[9] = j->spec_hit ? 1 : 0; info[10] = (j->need_replay || j->device_ranks) ? (int64_t)j->vmax : -1; info[11] = j->device_ranks ? 1 : 0;
    return TF_OK;
}

extern "C" int tf_watershed_ex2(const float *field, const int32_t *markers, const int8_t *mask,
                                const float *fwd, const float *bwd, int64_t T, int64_t H, int64_t W,
                                const int8_t *nbr_host, int n_nbr, int chain_depth, int max_depth, int flags,
                                int32_t *labels, uint8_t *ambiguous, void *ws, size_t ws_bytes,
                                int64_t *stats_host, void *stream)
{
    int64_t st[TF_WS_NSTATS];
    const int rc = ws_run(field, markers, mask, fwd, bwd, T, H, W, nbr_host, n_nbr, chain_depth, max_depth, flags, labels,
                          ambiguous, ws, ws_bytes, st, stream);
    if (stats_host) for (int i = 0; i < TF_WS_NSTATS; i++) stats_host[i] = st[i];
    return rc;
}

extern "C" int tf_watershed_ex(const float *field, const int32_t *markers, const int8_t *mask,
                               const float *fwd, const float *bwd, int64_t T, int64_t H, int64_t W,
                               const int8_t *nbr_host, int n_nbr, int chain_depth, int flags, int32_t *labels,
                               void *ws, size_t ws_bytes, int64_t *stats_host, void *stream)
{
    int64_t st[TF_WS_NSTATS];
    const int rc = ws_run(field, markers, mask, fwd, bwd, T, H, W, nbr_host, n_nbr, chain_depth, chain_depth, flags, labels,
                          nullptr, ws, ws_bytes, st, stream);
    if (stats_host) for (int i = 0; i < 8; i++) stats_host[i] = st[i];
    return rc;
}

extern "C" int tf_watershed(const float *field, const int32_t *markers, const int8_t *mask,
                            const float *fwd, const float *bwd, int64_t T, int64_t H, int64_t W,
                            const int8_t *nbr_host, int n_nbr, int chain_depth, int32_t *labels,
                            void *ws, size_t ws_bytes, int64_t *stats_host, void *stream)
{
    return tf_watershed_ex(field, markers, mask, fwd, bwd, T, H, W, nbr_host, n_nbr, chain_depth, 0, labels,
                           ws, ws_bytes, stats_host, stream);
}

// The reference's only native seam, argument for argument (tobac_flow/_watershed.pyx:222-233), on the GPU.
extern "C" size_t tf_watershed_raveled_workspace_bytes(int64_t n, int n_structure, int max_depth, int64_t max_relevant)
{
    if (n <= 0 || n > 0x7fffffffll || max_depth < 1 || max_depth > WS_MAX_DEPTH || n_structure < 1 || n_structure > WS_MAX_NBR) return 0;
    if (max_relevant <= 0 || max_relevant > n) max_relevant = n;
    return ws_full_bytes(n, n) + ws_compact_bytes(max_relevant, n_structure, max_depth);
}

extern "C" int tf_watershed_raveled_ex(const float *image, int64_t n, const int64_t *marker_locations, int64_t n_markers,
                                       const int64_t *structure_host, int n_structure,
                                       const int32_t *forward_offset, const int32_t *backward_offset,
                                       const int32_t *forward_offset_locations_host, const int32_t *backward_offset_locations_host,
                                       const int8_t *mask, const int32_t *strides_host, int n_strides, double compactness,
                                       int32_t *output, int wsl, int max_depth, int flags, void *ws, size_t ws_bytes,
                                       int64_t *stats_host, void *stream)
{
    TF_REQUIRE((flags & ~TF_WS_REFERENCE_ORDER) == 0, "tf_watershed_raveled_ex: unknown flag");
    TF_REQUIRE(image && structure_host && forward_offset && backward_offset && forward_offset_locations_host &&
               backward_offset_locations_host && mask && output && ws && (marker_locations || n_markers == 0),
               "tf_watershed_raveled: null pointer");
    (void)strides_host; (void)n_strides;         // only used by the compact-watershed branch of the reference
    TF_REQUIRE(compactness == 0.0 && !wsl, "tf_watershed_raveled: compactness > 0 and watershed lines are the dead branches of the "
               "reference's call path (watershed.py:151-164 passes 0 and False) and are not built");
    TF_REQUIRE(n_structure > 0 && n_structure <= WS_MAX_NBR, "tf_watershed_raveled: bad neighbour count");
    WsRavel rv; rv.n = n; rv.n_nbr = n_structure; rv.foff = forward_offset; rv.boff = backward_offset;
    for (int i = 0; i < n_structure; i++) {
        rv.structure[i] = structure_host[i]; rv.floc[i] = forward_offset_locations_host[i]; rv.bloc[i] = backward_offset_locations_host[i];
    }
    int64_t st[TF_WS_NSTATS];
    const int d0 = max_depth < 3 ? max_depth : 3;
    const int rc = ws_run(image, output, mask, nullptr, nullptr, 0, 0, 0, nullptr, n_structure, d0, max_depth, flags, output,
                          nullptr, ws, ws_bytes, st, stream, &rv, marker_locations, n_markers);
    if (stats_host) for (int i = 0; i < TF_WS_NSTATS; i++) stats_host[i] = st[i];
    return rc;
}

extern "C" int tf_watershed_raveled(const float *image, int64_t n, const int64_t *marker_locations, int64_t n_markers,
                                    const int64_t *structure_host, int n_structure,
                                    const int32_t *forward_offset, const int32_t *backward_offset,
                                    const int32_t *forward_offset_locations_host, const int32_t *backward_offset_locations_host,
                                    const int8_t *mask, const int32_t *strides_host, int n_strides, double compactness,
                                    int32_t *output, int wsl, int max_depth, void *ws, size_t ws_bytes,
                                    int64_t *stats_host, void *stream)
{
    return tf_watershed_raveled_ex(image, n, marker_locations, n_markers, structure_host, n_structure, forward_offset,
                                   backward_offset, forward_offset_locations_host, backward_offset_locations_host, mask,
                                   strides_host, n_strides, compactness, output, wsl, max_depth, 0, ws, ws_bytes, stats_host, stream);
}
