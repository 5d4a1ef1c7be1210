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
// Data layout (all in the caller's workspace, N = T*H*W):
//   state u8[N] (0 off / 1 floodable / 2 marker), off int16x4[N] rounded (fx, fy, bx, by),
//   K2, M1, C_1..C_{depth-1}, R, pushed : uint64[N].
// Every sweep is one launch over the volume; a pixel re-pushes only when its own key changed
// since its last push (pushed[]), so converged regions cost one 8-byte compare per sweep.
#include "tf_common.h"
#include <string.h>
#include <stdlib.h>

typedef unsigned long long u64;
#define WS_INF 0xFFFFFFFFFFFFFFFFull
#define WS_NEVER 0xFFFFFFFFFFFFFFFEull
#define WS_MAX_NBR 26
#define WS_MAX_DEPTH 8
#define WS_BATCH 8

struct WsGeom {
    int64_t T; int H, W; int64_t plane;
    int n_nbr;
    int8_t dt[WS_MAX_NBR], dy[WS_MAX_NBR], dx[WS_MAX_NBR];
};

struct WsArrays {
    const float *field; const uint8_t *state; const short4 *off;
    u64 *K2, *M1, *C[WS_MAX_DEPTH], *R, *pushed;
};

__device__ __forceinline__ u64 ws_ordkey(float v) {
    v = v + 0.0f;                                    // -0.0 -> +0.0 (they compare equal in the reference)
    unsigned u = __float_as_uint(v);
    return (u64)((u & 0x80000000u) ? ~u : (u | 0x80000000u));
}
__device__ __forceinline__ u64 ws_load(const u64 *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int ws_round_flow(float f) {
    // np.round(flow).astype(int32), watershed.py:121-141 (half to even); NaN -> 0
    return (f == f) ? __float2int_rn(f) : 0;
}

__global__ void __launch_bounds__(256)
k_ws_init(const float *__restrict__ field, const int32_t *__restrict__ markers, const int8_t *__restrict__ mask,
          const float *__restrict__ fwd, const float *__restrict__ bwd, WsGeom g,
          uint8_t *__restrict__ state, short4 *__restrict__ off, u64 *__restrict__ K2, u64 *__restrict__ M1)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    const int64_t t = blockIdx.z;
    if (x >= g.W || y >= g.H) return;
    const int64_t p = t * g.plane + (int64_t)y * g.W + x;
    const int32_t m = markers[p];
    const bool on = mask ? mask[p] != 0 : true;
    state[p] = m != 0 ? 2 : (on ? 1 : 0);
    float2 f = ((const float2 *)fwd)[p], b = ((const float2 *)bwd)[p];
    off[p] = make_short4((short)ws_round_flow(f.x), (short)ws_round_flow(f.y),
                         (short)ws_round_flow(b.x), (short)ws_round_flow(b.y));
    K2[p] = m != 0 ? (ws_ordkey(field[p]) << 32) : WS_INF;
    M1[p] = WS_INF;
}

__global__ void __launch_bounds__(256) k_ws_fill(u64 *__restrict__ a, int64_t n, u64 v) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = v;
}

// chain arrays of a marker: C_k = 0 for every k; R = raster index
__global__ void __launch_bounds__(256)
k_ws_init_level(const uint8_t *__restrict__ state, u64 *__restrict__ Ck, u64 *__restrict__ pushed, int64_t n, int is_root) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Ck[i] = state[i] == 2 ? (is_root ? (u64)i : 0ull) : WS_INF;
    pushed[i] = WS_NEVER;
}

// neighbour of p = (t, y, x) through slot i, or -1 (watershed.pyx:310-313 without the padding)
__device__ __forceinline__ int64_t ws_neighbour(const WsGeom &g, int64_t t, int y, int x, short4 o, int i) {
    const int dt = g.dt[i];
    int yy = y + g.dy[i], xx = x + g.dx[i];
    if (dt == 1) { xx += o.x; yy += o.y; }
    else if (dt == -1) { xx += o.z; yy += o.w; }
    const int64_t tt = t + dt;
    if (tt < 0 || tt >= g.T || (unsigned)yy >= (unsigned)g.H || (unsigned)xx >= (unsigned)g.W) return -1;
    return tt * g.plane + (int64_t)yy * g.W + xx;
}

// ---- phase A: K2 and M1 ------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_ws_relax_a(WsGeom g, WsArrays a, int *__restrict__ changed)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    const int64_t t = blockIdx.z;
    if (x >= g.W || y >= g.H) return;
    const int64_t p = t * g.plane + (int64_t)y * g.W + x;
    if (a.state[p] == 0) return;
    const u64 kp = ws_load(&a.K2[p]);
    if (kp == WS_INF || kp == a.pushed[p]) return;
    a.pushed[p] = kp;
    const u64 lp = kp >> 32;
    const short4 o = a.off[p];
    bool ch = false;
    for (int i = 0; i < g.n_nbr; i++) {
        const int64_t n = ws_neighbour(g, t, y, x, o, i);
        if (n < 0 || a.state[n] != 1) continue;
        const u64 vn = ws_ordkey(a.field[n]);
        const u64 cand = vn > lp ? ((vn << 32) | 1ull) : (vn == lp ? kp + 1ull : kp);
        const u64 old = atomicMin(&a.K2[n], cand);
        ch |= cand < old;
        atomicMin(&a.M1[n], kp);
    }
    if (ch) *changed = 1;
}

// ---- phase k >= 1 (chain level k) and phase R (k == depth) ----------------------------------------
// For edge p -> n with K2[p] == M1[n]:
//   n is an ENTRY (first pixel of a same-level run, pushed from a lower level or by a level marker)
//     iff K2[n] = (value(n), 1): its chain is [K2 n, chain(p)]   -> offered_j = C_{j-1}[p]
//   otherwise n continues p's run / descent: chain = [K2 n, tail(p)] -> offered_j = C_j[p]
// The candidate must agree with n on every level j < k.
__global__ void __launch_bounds__(256)
k_ws_relax_chain(WsGeom g, WsArrays a, int k, int depth, int *__restrict__ changed)
{
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    const int64_t t = blockIdx.z;
    if (x >= g.W || y >= g.H) return;
    const int64_t p = t * g.plane + (int64_t)y * g.W + x;
    if (a.state[p] == 0) return;
    const u64 kp = a.K2[p];                      // final since phase A
    if (kp == WS_INF) return;
    u64 *dst = k == depth ? a.R : a.C[k];
    const u64 own = ws_load(&dst[p]);
    if (own == a.pushed[p]) return;              // nothing new to offer (first visit: pushed = NEVER)
    a.pushed[p] = own;
    const short4 o = a.off[p];
    bool ch = false;
    for (int i = 0; i < g.n_nbr; i++) {
        const int64_t n = ws_neighbour(g, t, y, x, o, i);
        if (n < 0 || a.state[n] != 1) continue;
        if (a.M1[n] != kp) continue;
        const u64 kn = a.K2[n];
        const bool entry = (kn >> 32) == ws_ordkey(a.field[n]) && (kn & 0xFFFFFFFFull) == 1ull;
        bool match = true;
        for (int j = 1; j < k && match; j++) {
            const u64 offered_j = entry ? (j == 1 ? kp : a.C[j - 1][p]) : a.C[j][p];
            match = offered_j == a.C[j][n];
        }
        if (!match) continue;
        u64 offered;
        if (k == depth) offered = own;                                    // root: copied along every edge
        else offered = entry ? (k == 1 ? kp : a.C[k - 1][p]) : own;
        if (offered == WS_INF) continue;
        const u64 old = atomicMin(&dst[n], offered);
        ch |= offered < old;
    }
    if (ch) *changed = 1;
}

__global__ void __launch_bounds__(256)
k_ws_labels(const int32_t *__restrict__ markers, const uint8_t *__restrict__ state, const u64 *__restrict__ R,
            int32_t *__restrict__ labels, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t s = state[i];
    int32_t l = 0;
    if (s == 2) l = markers[i];
    else if (s == 1) { const u64 r = R[i]; if (r != WS_INF) l = markers[r]; }
    labels[i] = l;
}

extern "C" size_t tf_watershed_workspace_bytes(int64_t T, int64_t H, int64_t W, int chain_depth)
{
    if (T <= 0 || H <= 0 || W <= 0 || chain_depth < 1 || chain_depth > WS_MAX_DEPTH) return 0;
    const size_t n = (size_t)T * H * W;
    // state + off + (K2, M1, C_1..C_{d-1}, R, pushed) + flags, each 256-byte aligned
    return tf_align_up(n, 256) + tf_align_up(n * 8, 256) + (size_t)(chain_depth + 3) * tf_align_up(n * 8, 256) + 4096;
}

static int ws_run_phase(const WsGeom &g, const WsArrays &a, int phase_k, int depth, int *d_flags, int *h_flags,
                        hipStream_t s, int64_t max_sweeps, int64_t *sweeps_out)
{
    dim3 block(64, 4, 1), grid((g.W + 63) / 64, (g.H + 3) / 4, (unsigned)g.T);
    int64_t sweeps = 0;
    for (;;) {
        TF_CHECK_HIP(hipMemsetAsync(d_flags, 0, WS_BATCH * sizeof(int), s));
        for (int b = 0; b < WS_BATCH; b++) {
            if (phase_k == 0) hipLaunchKernelGGL(k_ws_relax_a, grid, block, 0, s, g, a, d_flags + b);
            else hipLaunchKernelGGL(k_ws_relax_chain, grid, block, 0, s, g, a, phase_k, depth, d_flags + b);
        }
        TF_CHECK_LAUNCH();
        TF_CHECK_HIP(hipMemcpyAsync(h_flags, d_flags, WS_BATCH * sizeof(int), hipMemcpyDeviceToHost, s));
        TF_CHECK_HIP(hipStreamSynchronize(s));
        bool done = false;
        for (int b = 0; b < WS_BATCH; b++) { sweeps++; if (!h_flags[b]) { done = true; break; } }
        if (done) break;
        if (sweeps > max_sweeps) { tf_set_error("tf_watershed: phase %d did not converge in %lld sweeps", phase_k, (long long)sweeps); return TF_ENOCONV; }
    }
    *sweeps_out = sweeps;
    return TF_OK;
}

extern "C" int tf_watershed(const float *field, const int32_t *markers, const int8_t *mask,
                            const float *fwd, const float *bwd, int64_t T, int64_t H, int64_t W,
                            const int8_t *nbr_host, int n_nbr, int chain_depth, int32_t *labels,
                            void *ws, size_t ws_bytes, int64_t *stats_host, void *stream)
{
    TF_REQUIRE(field && markers && fwd && bwd && nbr_host && labels && ws, "tf_watershed: null pointer");
    TF_REQUIRE(T > 0 && H > 0 && W > 0 && H < (1 << 15) && W < (1 << 15) && T < 65536, "tf_watershed: bad shape");
    TF_REQUIRE(n_nbr > 0 && n_nbr <= WS_MAX_NBR, "tf_watershed: bad neighbour count");
    TF_REQUIRE(chain_depth >= 1 && chain_depth <= WS_MAX_DEPTH, "tf_watershed: bad chain_depth");
    if (ws_bytes < tf_watershed_workspace_bytes(T, H, W, chain_depth)) { tf_set_error("tf_watershed: workspace too small"); return TF_ENOMEM; }
    hipStream_t s = (hipStream_t)stream;
    WsGeom g; g.T = T; g.H = (int)H; g.W = (int)W; g.plane = H * W; g.n_nbr = n_nbr;
    for (int i = 0; i < n_nbr; i++) {
        g.dt[i] = nbr_host[i * 3]; g.dy[i] = nbr_host[i * 3 + 1]; g.dx[i] = nbr_host[i * 3 + 2];
        TF_REQUIRE(abs(g.dt[i]) <= 1 && abs(g.dy[i]) <= 1 && abs(g.dx[i]) <= 1, "tf_watershed: neighbour offset out of range");
    }
    const int64_t N = T * H * W;
    TfArena ar(ws, ws_bytes);
    uint8_t *state = ar.take<uint8_t>(N);
    short4 *off = ar.take<short4>(N);
    WsArrays a; memset(&a, 0, sizeof(a));
    a.field = field; a.state = state; a.off = off;
    a.K2 = ar.take<u64>(N); a.M1 = ar.take<u64>(N);
    for (int k = 1; k < chain_depth; k++) a.C[k] = ar.take<u64>(N);
    a.R = ar.take<u64>(N); a.pushed = ar.take<u64>(N);
    int *d_flags = ar.take<int>(WS_BATCH);
    if (!ar.ok()) { tf_set_error("tf_watershed: workspace too small"); return TF_ENOMEM; }
    int h_flags[WS_BATCH];

    dim3 block(64, 4, 1), grid((g.W + 63) / 64, (g.H + 3) / 4, (unsigned)T);
    const unsigned nb1 = (unsigned)((N + 255) / 256);
    hipLaunchKernelGGL(k_ws_init, grid, block, 0, s, field, markers, mask, fwd, bwd, g, state, off, a.K2, a.M1);
    hipLaunchKernelGGL(k_ws_fill, dim3(nb1), dim3(256), 0, s, a.pushed, N, WS_NEVER);
    TF_CHECK_LAUNCH();
    const int64_t max_sweeps = 64 + 8 * (T + H + W) * 8;
    int64_t st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int rc = ws_run_phase(g, a, 0, chain_depth, d_flags, h_flags, s, max_sweeps * 64, &st[0]);
    if (rc) return rc;
    for (int k = 1; k <= chain_depth; k++) {
        u64 *dst = k == chain_depth ? a.R : a.C[k];
        hipLaunchKernelGGL(k_ws_init_level, dim3(nb1), dim3(256), 0, s, state, dst, a.pushed, N, k == chain_depth ? 1 : 0);
        TF_CHECK_LAUNCH();
        int64_t sw = 0;
        rc = ws_run_phase(g, a, k, chain_depth, d_flags, h_flags, s, max_sweeps * 64, &sw);
        if (rc) return rc;
        st[k < 7 ? k : 7] += sw;
    }
    hipLaunchKernelGGL(k_ws_labels, dim3(nb1), dim3(256), 0, s, markers, state, a.R, labels, N);
    TF_CHECK_LAUNCH();
    TF_CHECK_HIP(hipStreamSynchronize(s));
    if (stats_host) for (int i = 0; i < 8; i++) stats_host[i] = st[i];
    return TF_OK;
}
