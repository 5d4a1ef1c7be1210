// Label-overlap graphs on gfx950: the counting half of the reference's flow-aware labelling and of its
// cross-window linking.
//
//   tf_pair_counts        histogram of coinciding (a[i], b[i]) label pairs of two int32 volumes -- what the reference
//                         computes per label with np.bincount / np.unique over that label's pixels
//                         (tobac_flow/utils/label_utils.py:352-376 find_overlapping_labels, tobac_flow/linking.py:33-47
//                         find_overlaps)
//   tf_label_sizes        np.bincount(labels.ravel()) (label.py:139, linking.py:62-65)
//   tf_flow_link_overlap  tobac_flow/label.py:249-321 flow_link_overlap (and :127-175, the second half of flow_label)
//   tf_flow_label         tobac_flow/label.py:84-175 flow_label with subsegment_shrink = 0
//   tf_window_overlap_pairs  tobac_flow/linking.py:49-93 / :96-140: which labels of two windows are the same object,
//                         judged on the frames both windows hold
//
// Pair counting.  Labels are spatially coherent, so in raster order the key stream (a << 32 | b) consists of long
// runs.  One run-length pass (rocPRIM) turns N voxels into n_runs (key, length) records -- typically N / 50 -- which
// are radix-sorted by key and reduced by key.  HBM traffic: 8 B read per voxel once; everything after that works on
// the run records.  The label GRAPH (a few thousand nodes) is walked on the host in the reference's own visiting
// order, because that order decides the numbering (label.py:145-170: first come, first served).
#include "tf_common.h"
#include "tf_prim.h"
#include <vector>
#include <algorithm>

typedef unsigned long long u64;
#define PC_INVALID 0xFFFFFFFFFFFFFFFFull

struct PairKey {
    const int32_t *a, *b;
    int min_b;                                   // 1: pairs need b > 0; 0: b >= 0 is kept (b == 0 counts towards sizes)
    __host__ __device__ __forceinline__ u64 operator()(int64_t i) const {
        const int32_t x = a[i], y = b[i];
        return (x > 0 && y >= min_b) ? (((u64)(uint32_t)x << 32) | (uint32_t)y) : PC_INVALID;
    }
};
typedef rocprim::counting_iterator<int64_t> PcCount;
typedef rocprim::transform_iterator<PcCount, PairKey, u64> PcKeyIter;

// number of runs of the key stream (runs of the invalid key included)
// (grid-stride over the stream, one atomic per workgroup: with one atomic per wave the 0.9 M returning atomics of a
// 2 x 5424^2 overlap on ONE address took 0.7 ms, five times the time of reading the two label frames)
#define PC_COUNT_BLOCKS 2048
__global__ void __launch_bounds__(256)
k_pc_count_runs(PairKey key, int64_t n, unsigned long long *__restrict__ n_runs)
{
    __shared__ unsigned part[4];
    unsigned mine = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        mine += (i == 0 || key(i) != key(i - 1)) ? 1u : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mine += __shfl_down(mine, d);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned total = part[0] + part[1] + part[2] + part[3];
        if (total) atomicAdd(n_runs, (unsigned long long)total);
    }
}

__global__ void __launch_bounds__(256)
k_pc_widen(const int *__restrict__ in, int64_t n, int64_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[i];
}

__global__ void __launch_bounds__(256)
k_pc_unpack(const u64 *__restrict__ keys, const int64_t *__restrict__ cnt, int64_t n, int32_t *__restrict__ oa,
            int32_t *__restrict__ ob, int64_t *__restrict__ oc)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 k = keys[i];
    oa[i] = (int32_t)(k >> 32); ob[i] = (int32_t)(k & 0xFFFFFFFFull); oc[i] = cnt[i];
}

struct PcSizes { size_t rle, sort, reduce, runs_bytes; };
static PcSizes pc_temp_sizes(int64_t n, int64_t max_runs)
{
    PcSizes z; z.rle = z.sort = z.reduce = 0;
    PairKey pk{nullptr, nullptr, 1};
    PcKeyIter it(PcCount(0), pk);
    (void)tf_run_length_encode(nullptr, z.rle, it, (u64 *)nullptr, (int *)nullptr, (int *)nullptr, (size_t)n);
    (void)tf_sort_pairs(nullptr, z.sort, (const u64 *)nullptr, (u64 *)nullptr, (const int64_t *)nullptr, (int64_t *)nullptr, (size_t)max_runs);
    (void)tf_sum_by_key(nullptr, z.reduce, (const u64 *)nullptr, (u64 *)nullptr, (const int64_t *)nullptr, (int64_t *)nullptr, (int *)nullptr, (size_t)max_runs);
    return z;
}

extern "C" size_t tf_pair_counts_workspace_bytes(int64_t n, int64_t max_runs)
{
    if (n <= 0 || n > 0x7fffffffll) return 0;
    if (max_runs <= 0 || max_runs > n) max_runs = n;
    const PcSizes z = pc_temp_sizes(n, max_runs);
    const size_t temp = std::max(z.rle, std::max(z.sort, z.reduce));
    // run keys + run lengths (int) + widened lengths, sorted keys + sorted lengths, reduced keys + sums, counters
    return tf_align_up(temp, 256) + 3 * tf_align_up((size_t)max_runs * 8, 256) + 3 * tf_align_up((size_t)max_runs * 8, 256)
         + tf_align_up((size_t)max_runs * 4, 256) + 4096;
}

// Device-side result (keys sorted by (a, b), int64 counts) left in the workspace; host copies / unpacks as needed.
struct PcResult { const u64 *keys; const int64_t *counts; int64_t n_pairs; };

// number of runs of the (a, b) key stream: what the scratch of pc_run has to hold
static int pc_count(const int32_t *a, const int32_t *b, int64_t n, int min_b, void *ws, size_t ws_bytes, hipStream_t s,
                    int64_t *runs)
{
    TF_REQUIRE(a && b && n > 0 && n <= 0x7fffffffll, "tf_pair_counts: bad arguments");
    if (ws_bytes < 64) { tf_set_error("tf_pair_counts: workspace too small"); return TF_ENOMEM; }
    unsigned long long *d_cnt = (unsigned long long *)ws;
    PairKey pk{a, b, min_b};
    TF_CHECK_HIP(hipMemsetAsync(d_cnt, 0, sizeof(unsigned long long), s));
    hipLaunchKernelGGL(k_pc_count_runs, dim3((unsigned)std::min<int64_t>((n + 255) / 256, PC_COUNT_BLOCKS)), dim3(256), 0, s, pk, n, d_cnt);
    TF_CHECK_LAUNCH();
    unsigned long long h = 0;
    TF_CHECK_HIP(hipMemcpyAsync(&h, d_cnt, sizeof(h), hipMemcpyDeviceToHost, s));
    TF_CHECK_HIP(hipStreamSynchronize(s));
    *runs = (int64_t)h;
    return TF_OK;
}

// does a workspace of ws_bytes hold pc_run's scratch for `runs` runs of an n-voxel stream?
static bool pc_fits(int64_t n, int64_t runs, size_t ws_bytes)
{
    const PcSizes z = pc_temp_sizes(n, runs);
    const size_t temp = std::max(z.rle, std::max(z.sort, z.reduce));
    const size_t need = 256 + tf_align_up(temp ? temp : 1, 256) + 3 * tf_align_up((size_t)runs * 8, 256)
                      + 3 * tf_align_up((size_t)runs * 8, 256) + tf_align_up((size_t)runs * 4, 256) + 2048;
    return need <= ws_bytes;
}

static int pc_run(const int32_t *a, const int32_t *b, int64_t n, int min_b, void *ws, size_t ws_bytes, hipStream_t s,
                  PcResult *res, int64_t *runs_needed)
{
    TF_REQUIRE(a && b && n > 0 && n <= 0x7fffffffll, "tf_pair_counts: bad arguments");
    PairKey pk{a, b, min_b};
    TfArena ar(ws, ws_bytes);
    unsigned long long *d_cnt = ar.take<unsigned long long>(8);
    if (!ar.ok()) { tf_set_error("tf_pair_counts: workspace too small"); return TF_ENOMEM; }
    TF_CHECK_HIP(hipMemsetAsync(d_cnt, 0, 8 * sizeof(unsigned long long), s));
    hipLaunchKernelGGL(k_pc_count_runs, dim3((unsigned)std::min<int64_t>((n + 255) / 256, PC_COUNT_BLOCKS)), dim3(256), 0, s, pk, n, d_cnt);
    TF_CHECK_LAUNCH();
    unsigned long long h_runs = 0;
    TF_CHECK_HIP(hipMemcpyAsync(&h_runs, d_cnt, sizeof(h_runs), hipMemcpyDeviceToHost, s));
    TF_CHECK_HIP(hipStreamSynchronize(s));
    const int64_t runs = (int64_t)h_runs;
    if (runs_needed) *runs_needed = runs;
    const PcSizes z = pc_temp_sizes(n, runs);
    const size_t temp = std::max(z.rle, std::max(z.sort, z.reduce));
    char *tmp = ar.take<char>(temp ? temp : 1);
    u64 *rk = ar.take<u64>(runs), *sk = ar.take<u64>(runs), *uk = ar.take<u64>(runs);
    int *rl = ar.take<int>(runs);
    int64_t *rl64 = ar.take<int64_t>(runs), *sl = ar.take<int64_t>(runs), *us = ar.take<int64_t>(runs);
    if (!ar.ok()) {
        tf_set_error("tf_pair_counts: workspace too small for %lld runs", (long long)runs);
        return TF_ENOMEM;
    }
    int *d_nruns = (int *)(d_cnt + 2), *d_npairs = (int *)(d_cnt + 4);
    PcKeyIter it(PcCount(0), pk);
    size_t tb = temp;
    TF_CHECK_HIP(tf_run_length_encode(tmp, tb, it, rk, rl, d_nruns, (size_t)n, s));
    hipLaunchKernelGGL(k_pc_widen, dim3((unsigned)((runs + 255) / 256)), dim3(256), 0, s, rl, runs, rl64);
    TF_CHECK_LAUNCH();
    tb = temp;
    TF_CHECK_HIP(tf_sort_pairs(tmp, tb, (const u64 *)rk, sk, (const int64_t *)rl64, sl, (size_t)runs, s));
    tb = temp;
    TF_CHECK_HIP(tf_sum_by_key(tmp, tb, (const u64 *)sk, uk, (const int64_t *)sl, us, d_npairs, (size_t)runs, s));
    int h_npairs = 0; u64 last_key = 0;
    TF_CHECK_HIP(hipMemcpyAsync(&h_npairs, d_npairs, sizeof(int), hipMemcpyDeviceToHost, s));
    TF_CHECK_HIP(hipStreamSynchronize(s));
    if (h_npairs > 0) {
        TF_CHECK_HIP(hipMemcpyAsync(&last_key, uk + (h_npairs - 1), sizeof(u64), hipMemcpyDeviceToHost, s));
        TF_CHECK_HIP(hipStreamSynchronize(s));
        if (last_key == PC_INVALID) h_npairs -= 1;            // the invalid key sorts last
    }
    res->keys = uk; res->counts = us; res->n_pairs = h_npairs;
    return TF_OK;
}

extern "C" int tf_pair_counts(const int32_t *a, const int32_t *b, int64_t n, int include_b_zero,
                              int32_t *out_a, int32_t *out_b, int64_t *out_count, int64_t max_pairs,
                              int64_t *n_pairs_host, void *ws, size_t ws_bytes, void *stream)
{
    TF_REQUIRE(n_pairs_host && ws, "tf_pair_counts: null pointer");
    hipStream_t s = (hipStream_t)stream;
    PcResult r; int64_t runs = 0;
    const int rc = pc_run(a, b, n, include_b_zero ? 0 : 1, ws, ws_bytes, s, &r, &runs);
    if (rc == TF_ENOMEM) { *n_pairs_host = runs; return rc; }
    if (rc) return rc;
    *n_pairs_host = r.n_pairs;
    if (r.n_pairs > max_pairs) { tf_set_error("tf_pair_counts: %lld pairs, room for %lld", (long long)r.n_pairs, (long long)max_pairs); return TF_ENOMEM; }
    if (r.n_pairs > 0) {
        TF_REQUIRE(out_a && out_b && out_count, "tf_pair_counts: null output");
        hipLaunchKernelGGL(k_pc_unpack, dim3((unsigned)((r.n_pairs + 255) / 256)), dim3(256), 0, s, r.keys, r.counts, r.n_pairs,
                           out_a, out_b, out_count);
        TF_CHECK_LAUNCH();
    }
    return TF_OK;
}

// ---- rank of (a[i], b[i]) among the sorted distinct pairs ----------------------------------------------------------
// tobac_flow/utils/label_utils.py:183-200 make_step_labels: the non-zero mask of a label volume is split into the pieces
// connected within a time step (flat_label = tf_label with the structure's t planes zeroed), every piece into the
// original labels it contains, ids contiguous from 1 in the order (piece, label) -- i.e. the RANK of the voxel's
// (piece, label) pair among the distinct pairs, which tf_pair_counts returns sorted.  A voxel's rank is a binary search
// in that list (a few thousand entries: L2-resident); consecutive voxels mostly repeat the pair, so a thread keeps the
// last answer.  4 + 4 B read, 4 B written per voxel.
// ALIGNED: all three volumes are 16-byte aligned (wide loads / stores); otherwise -- a view such as labels[1:] of a volume
// whose H * W is not a multiple of 4 -- the same quads with scalar accesses.
template <bool ALIGNED>
__global__ void __launch_bounds__(256)
k_pair_rank(const int32_t *__restrict__ a, const int32_t *__restrict__ b, int64_t n, const int32_t *__restrict__ pa,
            const int32_t *__restrict__ pb, int n_pairs, int32_t *__restrict__ out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
    for (int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i0 < n; i0 += stride) {
        int32_t va[4], vb[4], r[4];
        if (ALIGNED && i0 + 4 <= n) {
            const int4 qa = *(const int4 *)(a + i0), qb = *(const int4 *)(b + i0);
            va[0] = qa.x; va[1] = qa.y; va[2] = qa.z; va[3] = qa.w; vb[0] = qb.x; vb[1] = qb.y; vb[2] = qb.z; vb[3] = qb.w;
        } else {
            for (int j = 0; j < 4; j++) { va[j] = i0 + j < n ? a[i0 + j] : 0; vb[j] = i0 + j < n ? b[i0 + j] : 0; }
        }
        u64 last_key = PC_INVALID; int32_t last_rank = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            int32_t rank = 0;
            if (va[j] > 0 && vb[j] > 0) {
                const u64 key = ((u64)(uint32_t)va[j] << 32) | (uint32_t)vb[j];
                if (key == last_key) rank = last_rank;
                else {
                    int lo = 0, hi = n_pairs;                         // first pair >= key
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        const u64 km = ((u64)(uint32_t)pa[mid] << 32) | (uint32_t)pb[mid];
                        if (km < key) lo = mid + 1; else hi = mid;
                    }
                    if (lo < n_pairs && pa[lo] == va[j] && pb[lo] == vb[j]) rank = lo + 1;
                    last_key = key; last_rank = rank;
                }
            }
            r[j] = rank;
        }
        if (ALIGNED && i0 + 4 <= n) *(int4 *)(out + i0) = make_int4(r[0], r[1], r[2], r[3]);
        else for (int j = 0; j < 4 && i0 + j < n; j++) out[i0 + j] = r[j];
    }
}

extern "C" int tf_pair_rank(const int32_t *a, const int32_t *b, int64_t n, const int32_t *pairs_a, const int32_t *pairs_b,
                            int64_t n_pairs, int32_t *out, void *stream)
{
    TF_REQUIRE(a && b && out && n > 0 && n_pairs >= 0 && n_pairs < (1ll << 31) && (n_pairs == 0 || (pairs_a && pairs_b)), "tf_pair_rank: bad arguments");
    TF_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 3) == 0, "tf_pair_rank: volumes must be 4-byte aligned");
    const bool aligned = (((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0;
    hipStream_t s = (hipStream_t)stream;
    const int64_t quads = (n + 3) / 4;
    const unsigned grid = (unsigned)std::min<int64_t>((quads + 255) / 256, 256 * 32);
    if (aligned) hipLaunchKernelGGL(k_pair_rank<true>, dim3(grid), dim3(256), 0, s, a, b, n, pairs_a, pairs_b, (int)n_pairs, out);
    else hipLaunchKernelGGL(k_pair_rank<false>, dim3(grid), dim3(256), 0, s, a, b, n, pairs_a, pairs_b, (int)n_pairs, out);
    TF_CHECK_LAUNCH();
    return TF_OK;
}

// ---- np.bincount ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_label_sizes(const int32_t *__restrict__ labels, int64_t n, int64_t n_labels, unsigned long long *__restrict__ sizes)
{
    // one atomic per run of equal labels inside a thread's 8 consecutive voxels
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i0 >= n) return;
    int32_t cur = -1; unsigned long long run = 0;
    for (int j = 0; j < 8 && i0 + j < n; j++) {
        const int32_t l = labels[i0 + j];
        if (l == cur) { run++; continue; }
        if (run && cur >= 0 && cur <= n_labels) atomicAdd(&sizes[cur], run);
        cur = l; run = 1;
    }
    if (run && cur >= 0 && cur <= n_labels) atomicAdd(&sizes[cur], run);
}

extern "C" int tf_label_sizes(const int32_t *labels, int64_t n, int64_t n_labels, int64_t *sizes, void *stream)
{
    TF_REQUIRE(labels && sizes && n > 0 && n_labels >= 0, "tf_label_sizes: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    TF_CHECK_HIP(hipMemsetAsync(sizes, 0, (size_t)(n_labels + 1) * sizeof(int64_t), s));
    hipLaunchKernelGGL(k_label_sizes, dim3((unsigned)((n + 2047) / 2048)), dim3(256), 0, s, labels, n, n_labels,
                       (unsigned long long *)sizes);
    TF_CHECK_LAUNCH();
    return TF_OK;
}

__global__ void __launch_bounds__(256)
k_label_max(const int32_t *__restrict__ labels, int64_t n, int *__restrict__ out)
{
    int m = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) m = max(m, labels[i]);
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(out, m);
}

static int label_max(const int32_t *labels, int64_t n, int *d_tmp, hipStream_t s, int *out)
{
    TF_CHECK_HIP(hipMemsetAsync(d_tmp, 0, sizeof(int), s));
    const unsigned nb = (unsigned)std::min<int64_t>((n + 255) / 256, 2048);
    hipLaunchKernelGGL(k_label_max, dim3(nb), dim3(256), 0, s, labels, n, d_tmp);
    TF_CHECK_LAUNCH();
    TF_CHECK_HIP(hipMemcpyAsync(out, d_tmp, sizeof(int), hipMemcpyDeviceToHost, s));
    TF_CHECK_HIP(hipStreamSynchronize(s));
    return TF_OK;
}

// ---- flow_link_overlap ------------------------------------------------------------------------------------------
// label.py:268-321.  back / forward = the per-step labels of t-1 / t+1 warped to t with the backward / forward
// flow (nearest neighbour; Flow.convolve with the t-planes of `structure` reduced to their centre column, i.e.
// structure * [1, 0, 1]).  For label L with n pixels, a warped label M != 0 is a neighbour iff
//   count(L, M) > absolute_overlap  and  count(L, M) >= overlap * min(n, size(M))          (label_utils.py:352-376)
// Groups: labels in ascending order; an unvisited label opens a group and absorbs, breadth first, the unvisited
// neighbours of every member -- forward neighbours (ascending) before backward ones (label.py:208-245).
extern "C" size_t tf_flow_link_workspace_bytes(int64_t T, int64_t H, int64_t W, int64_t max_runs)
{
    const int64_t N = T * H * W;
    if (T <= 0 || H <= 0 || W <= 0 || N > 0x7fffffffll) return 0;
    return 2 * tf_align_up((size_t)N * 4, 256) + tf_pair_counts_workspace_bytes(N, max_runs) + 4096;
}

static void link_groups(int n_lab, const std::vector<int64_t> &sizes, const std::vector<u64> keys[2],
                        const std::vector<int64_t> cnts[2], double overlap, int64_t absolute_overlap, std::vector<int32_t> &lut)
{
    // adjacency per direction (0 = forward, 1 = backward), CSR over the sorted keys
    std::vector<int64_t> start[2];
    std::vector<int32_t> nb[2];
    for (int d = 0; d < 2; d++) {
        start[d].assign(n_lab + 2, 0);
        for (size_t i = 0; i < keys[d].size(); i++) {
            const int32_t a = (int32_t)(keys[d][i] >> 32), b = (int32_t)(keys[d][i] & 0xFFFFFFFFull);
            if (a < 1 || a > n_lab || b < 1 || b > n_lab) continue;
            const int64_t c = cnts[d][i];
            const int64_t m = std::min(sizes[a], sizes[b]);
            if (c > absolute_overlap && (double)c >= overlap * (double)m) { nb[d].push_back(b); start[d][a + 1]++; }
        }
        for (int l = 1; l <= n_lab + 1; l++) start[d][l] += start[d][l - 1];
    }
    std::vector<char> seen(n_lab + 1, 0);
    lut.assign(n_lab + 1, 0);
    std::vector<int32_t> stack;
    int32_t n_groups = 0;
    for (int label = 1; label <= n_lab; label++) {
        if (seen[label]) continue;
        n_groups++;
        stack.clear(); stack.push_back(label); seen[label] = 1;
        for (size_t i = 0; i < stack.size(); i++) {
            const int cur = stack[i];
            if (sizes[cur] == 0) continue;                      // label.py:218: labels without pixels have no neighbours
            for (int d = 0; d < 2; d++)
                for (int64_t e = start[d][cur]; e < start[d][cur + 1]; e++) {
                    const int32_t m = nb[d][e];
                    if (!seen[m]) { seen[m] = 1; stack.push_back(m); }
                }
        }
        for (int32_t m : stack) lut[m] = sizes[m] > 0 ? n_groups : 0;
    }
}

extern "C" int tf_flow_link_overlap(const int32_t *flat_labels, const float *fwd, const float *bwd,
                                    int64_t T, int64_t H, int64_t W, const uint8_t *structure_host,
                                    double overlap, int64_t absolute_overlap, int32_t *out, int *n_objects_host,
                                    void *ws, size_t ws_bytes, void *stream)
{
    TF_REQUIRE(flat_labels && fwd && bwd && structure_host && out && ws, "tf_flow_link_overlap: null pointer");
    const int64_t N = T * H * W;
    TF_REQUIRE(T > 0 && H > 0 && W > 0 && N <= 0x7fffffffll, "tf_flow_link_overlap: bad shape");
    hipStream_t s = (hipStream_t)stream;
    TfArena ar(ws, ws_bytes);
    int32_t *warped = ar.take<int32_t>(2 * N);
    int *d_tmp = ar.take<int>(64);
    if (!ar.ok()) { tf_set_error("tf_flow_link_overlap: workspace too small"); return TF_ENOMEM; }
    // structure * [1, 0, 1] along t (label.py:266)
    uint8_t st[27];
    for (int i = 0; i < 27; i++) st[i] = (i / 9 == 1) ? 0 : (structure_host[i] ? 1 : 0);
    int n_taps_prev = 0, n_taps_next = 0;
    for (int i = 0; i < 9; i++) { n_taps_prev += st[i]; n_taps_next += st[18 + i]; }
    TF_REQUIRE(n_taps_prev == 1 && n_taps_next == 1, "tf_flow_link_overlap: the structure must have exactly one tap in each "
               "of its t-1 and t+1 planes (the reference unpacks Flow.convolve's stack into back_labels, forward_labels)");
    int rc = tf_convolve(flat_labels, TF_I32, T, H, W, fwd, bwd, st, TF_INTERP_NEAREST, 0.0, TF_FUNC_STACK, warped, TF_I32, 0, T, stream);
    if (rc) return rc;
    int n_lab = 0;
    rc = label_max(flat_labels, N, d_tmp, s, &n_lab);
    if (rc) return rc;
    std::vector<int64_t> sizes(n_lab + 1, 0);
    std::vector<u64> keys[2];
    std::vector<int64_t> cnts[2];
    char *rest = (char *)ws + tf_align_up(ar.used, 256);
    const size_t rest_bytes = ws_bytes - tf_align_up(ar.used, 256);
    if (n_lab > 0) {
        // scratch check for all three counting passes at once, so that ONE retry with the reported run count suffices
        int64_t need = 0;
        for (int d = 0; d < 3; d++) {
            const int32_t *b = d == 0 ? warped + N : (d == 1 ? warped : flat_labels);
            int64_t runs = 0;
            rc = pc_count(flat_labels, b, N, 1, rest, rest_bytes, s, &runs);
            if (rc) return rc;
            need = std::max(need, runs);
        }
        if (!pc_fits(N, need, rest_bytes)) {
            if (n_objects_host) *n_objects_host = (int)std::min<int64_t>(need, 0x7fffffff);
            tf_set_error("tf_flow_link_overlap: workspace too small for %lld label runs", (long long)need);
            return TF_ENOMEM;
        }
        // bincount of the labels themselves = pair counts of (label, label)
        for (int d = 0; d < 3; d++) {
            const int32_t *b = d == 0 ? warped + N /* forward */ : (d == 1 ? warped /* back */ : flat_labels);
            PcResult r; int64_t runs = 0;
            rc = pc_run(flat_labels, b, N, 1, rest, rest_bytes, s, &r, &runs);
            if (rc) { if (rc == TF_ENOMEM && n_objects_host) *n_objects_host = (int)std::min<int64_t>(runs, 0x7fffffff); return rc; }
            std::vector<u64> k(r.n_pairs); std::vector<int64_t> c(r.n_pairs);
            if (r.n_pairs) {
                TF_CHECK_HIP(hipMemcpyAsync(k.data(), r.keys, r.n_pairs * sizeof(u64), hipMemcpyDeviceToHost, s));
                TF_CHECK_HIP(hipMemcpyAsync(c.data(), r.counts, r.n_pairs * sizeof(int64_t), hipMemcpyDeviceToHost, s));
                TF_CHECK_HIP(hipStreamSynchronize(s));
            }
            if (d < 2) { keys[d].swap(k); cnts[d].swap(c); }
            else for (int64_t i = 0; i < r.n_pairs; i++) sizes[(int32_t)(k[i] >> 32)] = c[i];
        }
    }
    std::vector<int32_t> lut;
    link_groups(n_lab, sizes, keys, cnts, overlap, absolute_overlap, lut);
    int32_t n_groups = 0;
    for (int32_t v : lut) n_groups = std::max(n_groups, v);
    if (n_objects_host) *n_objects_host = n_groups;
    // relabel: the LUT goes through the (now free) pair-count scratch
    int32_t *d_lut = (int32_t *)rest;
    if (rest_bytes < (size_t)(n_lab + 1) * 4) { tf_set_error("tf_flow_link_overlap: workspace too small"); return TF_ENOMEM; }
    TF_CHECK_HIP(hipMemcpyAsync(d_lut, lut.data(), (size_t)(n_lab + 1) * 4, hipMemcpyHostToDevice, s));
    rc = tf_apply_lut(flat_labels, N, d_lut, n_lab + 1, out, stream);
    if (rc) return rc;
    TF_CHECK_HIP(hipStreamSynchronize(s));                       // lut (host vector) is read by the copy above
    return TF_OK;
}

extern "C" size_t tf_flow_label_workspace_bytes(int64_t T, int64_t H, int64_t W, int64_t max_runs)
{
    const size_t a = tf_flow_link_workspace_bytes(T, H, W, max_runs), b = tf_label_workspace_bytes(T, H, W);
    if (!a || !b) return 0;
    return tf_align_up((size_t)(T * H * W) * 4, 256) + std::max(a, b) + 256;
}

extern "C" int tf_flow_label(const uint8_t *mask, const float *fwd, const float *bwd, int64_t T, int64_t H, int64_t W,
                             const uint8_t *structure_host, double overlap, int64_t absolute_overlap,
                             int32_t *labels, int *n_objects_host, void *ws, size_t ws_bytes, void *stream)
{
    TF_REQUIRE(mask && structure_host && labels && ws, "tf_flow_label: null pointer");
    const int64_t N = T * H * W;
    TF_REQUIRE(T > 0 && H > 0 && W > 0 && N <= 0x7fffffffll, "tf_flow_label: bad shape");
    TfArena ar(ws, ws_bytes);
    int32_t *flat = ar.take<int32_t>(N);
    if (!ar.ok()) { tf_set_error("tf_flow_label: workspace too small"); return TF_ENOMEM; }
    char *rest = (char *)ws + tf_align_up(ar.used, 256);
    const size_t rest_bytes = ws_bytes - tf_align_up(ar.used, 256);
    uint8_t st[27];                                              // label_utils.py:170-172: no connection across t
    for (int i = 0; i < 27; i++) st[i] = (i / 9 == 1) ? structure_host[i] : 0;
    int n_flat = 0;
    int rc = tf_label(mask, T, H, W, st, flat, &n_flat, rest, rest_bytes, stream);
    if (rc) return rc;
    return tf_flow_link_overlap(flat, fwd, bwd, T, H, W, structure_host, overlap, absolute_overlap, labels, n_objects_host,
                                rest, rest_bytes, stream);
}

// ---- cross-window linking -----------------------------------------------------------------------------------------
// linking.py:49-93 (cores) / :96-140 (anvils): `left` and `right` are the labels two consecutive windows assign to
// the SAME frames (the caller has already dropped the first and the last common frame, linking.py:55-56).  For every
// left label L (n pixels in these frames) and right label M != 0 (size(M) pixels in these frames):
//   linked  iff  count >= atol (count > 0 when atol == 0)  and  (rtol <= 0 or max(count / n, count / size(M)) >= rtol)
// Output: the linked (L, M) pairs sorted by (L, M), on the HOST (they are few and feed the union-find / all-gather).
extern "C" int tf_window_overlap_pairs(const int32_t *left, const int32_t *right, int64_t n, int64_t atol, double rtol,
                                       int32_t *pairs_host, int64_t max_pairs, int64_t *n_pairs_host,
                                       void *ws, size_t ws_bytes, void *stream)
{
    TF_REQUIRE(left && right && n_pairs_host && ws && n > 0, "tf_window_overlap_pairs: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    PcResult r; int64_t runs = 0, runs2 = 0;
    int rc = pc_count(left, right, n, 0, ws, ws_bytes, s, &runs);
    if (!rc) rc = pc_count(right, right, n, 1, ws, ws_bytes, s, &runs2);
    if (rc) return rc;
    if (!pc_fits(n, std::max(runs, runs2), ws_bytes)) {
        *n_pairs_host = std::max(runs, runs2);
        tf_set_error("tf_window_overlap_pairs: workspace too small for %lld label runs", (long long)*n_pairs_host);
        return TF_ENOMEM;
    }
    rc = pc_run(left, right, n, 0, ws, ws_bytes, s, &r, &runs);           // right == 0 kept: it counts towards n
    if (rc == TF_ENOMEM) { *n_pairs_host = runs; return rc; }
    if (rc) return rc;
    std::vector<u64> k(r.n_pairs); std::vector<int64_t> c(r.n_pairs);
    if (r.n_pairs) {
        TF_CHECK_HIP(hipMemcpyAsync(k.data(), r.keys, r.n_pairs * sizeof(u64), hipMemcpyDeviceToHost, s));
        TF_CHECK_HIP(hipMemcpyAsync(c.data(), r.counts, r.n_pairs * sizeof(int64_t), hipMemcpyDeviceToHost, s));
        TF_CHECK_HIP(hipStreamSynchronize(s));
    }
    // size of every right label over ALL voxels of these frames (linking.py:62-65), i.e. also where left == 0
    PcResult rr;
    rc = pc_run(right, right, n, 1, ws, ws_bytes, s, &rr, &runs);
    if (rc == TF_ENOMEM) { *n_pairs_host = runs; return rc; }
    if (rc) return rc;
    std::vector<u64> rk(rr.n_pairs); std::vector<int64_t> rc_(rr.n_pairs);
    if (rr.n_pairs) {
        TF_CHECK_HIP(hipMemcpyAsync(rk.data(), rr.keys, rr.n_pairs * sizeof(u64), hipMemcpyDeviceToHost, s));
        TF_CHECK_HIP(hipMemcpyAsync(rc_.data(), rr.counts, rr.n_pairs * sizeof(int64_t), hipMemcpyDeviceToHost, s));
        TF_CHECK_HIP(hipStreamSynchronize(s));
    }
    auto right_size = [&](uint32_t m) -> int64_t {
        const u64 key = ((u64)m << 32) | m;
        auto it = std::lower_bound(rk.begin(), rk.end(), key);
        return (it != rk.end() && *it == key) ? std::max<int64_t>(rc_[it - rk.begin()], 1) : 1;
    };
    int64_t out = 0;
    for (size_t i = 0; i < k.size();) {
        size_t j = i; int64_t n_left = 0;
        const uint32_t L = (uint32_t)(k[i] >> 32);
        while (j < k.size() && (uint32_t)(k[j] >> 32) == L) { n_left += c[j]; j++; }
        for (size_t e = i; e < j; e++) {
            const uint32_t M = (uint32_t)(k[e] & 0xFFFFFFFFull);
            if (M == 0) continue;
            const int64_t cnt = c[e];
            bool ok = atol > 0 ? cnt >= atol : cnt > 0;
            if (ok && rtol > 0) ok = std::max((double)cnt / (double)n_left, (double)cnt / (double)right_size(M)) >= rtol;
            if (!ok) continue;
            if (out < max_pairs && pairs_host) { pairs_host[2 * out] = (int32_t)L; pairs_host[2 * out + 1] = (int32_t)M; }
            out++;
        }
        i = j;
    }
    *n_pairs_host = out;
    if (out > max_pairs) { tf_set_error("tf_window_overlap_pairs: %lld pairs, room for %lld", (long long)out, (long long)max_pairs); return TF_ENOMEM; }
    return TF_OK;
}

// ---- slice_labels (utils/label_utils.py:312-349): one id per (label, time step) --------------------------------------
// The reference shifts every step's labels by the running sum of the steps' largest labels, then renumbers the ids that
// occur densely in ascending order (bincount -> LUT).  Here: per-step maxima (one atomicMax per wave), the T offsets on
// the host, presence flags of the shifted ids, an exclusive scan as the LUT, one write pass.
__global__ void __launch_bounds__(256)
k_sl_step_max(const int32_t *__restrict__ labels, int64_t hw, int *__restrict__ step_max)
{
    const int32_t *L = labels + (int64_t)blockIdx.y * hw;
    int m = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (int64_t)gridDim.x * blockDim.x) m = max(m, L[i]);
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(step_max + blockIdx.y, m);
}

__global__ void __launch_bounds__(256)
k_sl_mark(const int32_t *__restrict__ labels, int64_t hw, const int *__restrict__ offset, int *__restrict__ present)
{
    const int32_t *L = labels + (int64_t)blockIdx.y * hw;
    const int off = offset[blockIdx.y];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = L[i];
        // shifted id 0 is the background: it takes part in the dense renumbering when (and only when) it occurs
        if (v > 0) { if (!present[off + v]) present[off + v] = 1; }
        else if (v == 0 && !present[0]) present[0] = 1;
    }
}

__global__ void __launch_bounds__(256)
k_sl_apply(const int32_t *__restrict__ labels, int64_t hw, const int *__restrict__ offset, const int *__restrict__ rank,
           int32_t *__restrict__ out)
{
    const int64_t base = (int64_t)blockIdx.y * hw;
    const int off = offset[blockIdx.y];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = labels[base + i];
        // rank = number of present shifted ids below this one, the background id 0 included if any voxel is 0 -- in a
        // volume WITHOUT background the reference's LUT (np.arange over the ids that occur) sends the smallest label to 0
        out[base + i] = v > 0 ? rank[off + v] : 0;
    }
}

static size_t sl_scan_bytes(int64_t n_ids)
{
    size_t b = 0;
    (void)tf_exclusive_sum(nullptr, b, (const int *)nullptr, (int *)nullptr, (size_t)std::min<int64_t>(n_ids, INT32_MAX));
    return b;
}

// id_capacity: the largest sum of per-step maxima the workspace is sized for (0 = unknown: size for T * 2^20)
extern "C" size_t tf_slice_labels_workspace_bytes(int64_t T, int64_t id_capacity)
{
    if (T <= 0) return 0;
    const int64_t ids = (id_capacity > 0 ? id_capacity : T * (int64_t)(1 << 20)) + 1;
    return 2 * tf_align_up((size_t)T * 4, 256) + 2 * tf_align_up((size_t)ids * 4, 256) + tf_align_up(sl_scan_bytes(ids), 256) + 4096;
}

extern "C" int tf_slice_labels(const int32_t *labels, int64_t T, int64_t hw, int32_t *out, int64_t *n_step_labels_host,
                               void *ws, size_t ws_bytes, void *stream)
{
    TF_REQUIRE(labels && out && n_step_labels_host && ws, "tf_slice_labels: null pointer");
    TF_REQUIRE(T > 0 && hw > 0 && T <= 65535, "tf_slice_labels: bad shape");
    hipStream_t s = (hipStream_t)stream;
    TfArena ar(ws, ws_bytes);
    int *d_max = ar.take<int>(T), *d_off = ar.take<int>(T);
    if (!ar.ok()) { tf_set_error("tf_slice_labels: workspace too small"); return TF_ENOMEM; }
    TF_CHECK_HIP(hipMemsetAsync(d_max, 0, (size_t)T * 4, s));
    const dim3 grid((unsigned)std::min<int64_t>((hw + 255) / 256, 1024), (unsigned)T);
    hipLaunchKernelGGL(k_sl_step_max, grid, dim3(256), 0, s, labels, hw, d_max);
    TF_CHECK_LAUNCH();
    std::vector<int> h_max(T), h_off(T);
    TF_CHECK_HIP(hipMemcpyAsync(h_max.data(), d_max, (size_t)T * 4, hipMemcpyDeviceToHost, s));
    TF_CHECK_HIP(hipStreamSynchronize(s));
    int64_t total = 0;
    for (int64_t t = 0; t < T; t++) { h_off[t] = (int)total; total += h_max[t]; }
    TF_REQUIRE(total < INT32_MAX, "tf_slice_labels: the shifted ids overflow int32 (as they would in the reference)");
    *n_step_labels_host = total;                              // what a retry must be sized for, should the workspace be short
    const int64_t ids = total + 1;
    int *d_present = ar.take<int>(ids), *d_rank = ar.take<int>(ids);
    size_t scan_bytes = sl_scan_bytes(ids);
    void *d_scan = ar.take<unsigned char>(scan_bytes);
    if (!ar.ok()) { tf_set_error("tf_slice_labels: workspace too small for %lld shifted ids", (long long)total); return TF_ENOMEM; }
    TF_CHECK_HIP(hipMemcpyAsync(d_off, h_off.data(), (size_t)T * 4, hipMemcpyHostToDevice, s));
    TF_CHECK_HIP(hipMemsetAsync(d_present, 0, (size_t)ids * 4, s));
    hipLaunchKernelGGL(k_sl_mark, grid, dim3(256), 0, s, labels, hw, (const int *)d_off, d_present);
    TF_CHECK_LAUNCH();
    TF_CHECK_HIP(tf_exclusive_sum(d_scan, scan_bytes, (const int *)d_present, d_rank, (size_t)ids, s));
    hipLaunchKernelGGL(k_sl_apply, grid, dim3(256), 0, s, labels, hw, (const int *)d_off, (const int *)d_rank, out);
    TF_CHECK_LAUNCH();
    int last_rank = 0, last_present = 0;
    TF_CHECK_HIP(hipMemcpyAsync(&last_rank, d_rank + total, 4, hipMemcpyDeviceToHost, s));
    TF_CHECK_HIP(hipMemcpyAsync(&last_present, d_present + total, 4, hipMemcpyDeviceToHost, s));
    TF_CHECK_HIP(hipStreamSynchronize(s));                    // h_off must outlive the copy; the count is returned on the host
    *n_step_labels_host = (int64_t)last_rank + last_present - 1;       // ids that occur, minus the one that maps to 0
    return TF_OK;
}

// ---- per-label (weighted) statistics (analysis.py:204-245, 293-376) ----------------------------------------------------
// weighted_statistics_on_labels / get_stats_for_labels evaluate, for every label, the weighted mean, the weighted
// standard deviation about it, and the largest / smallest value where the weight is positive -- NaN values dropped.
// Two passes over the volume, double accumulators, one atomic per RUN of equal labels in a thread's 16 consecutive
// voxels (labels are spatially coherent): pass 1 sums w (all voxels / voxels with a value), w x, and the extrema;
// pass 2 sums w (x - mean)^2.  acc layout per label id (0 .. n_labels): {sum w all, sum w valued, sum w x, sum w dev^2}
// doubles + {max key, min key} as order-preserving 32-bit keys.
__device__ __forceinline__ unsigned ls_key(float v) { const unsigned b = __float_as_uint(v); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }
__device__ __forceinline__ float ls_unkey(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

template <int PASS>
__global__ void __launch_bounds__(256)
k_label_stats(const int32_t *__restrict__ labels, const float *__restrict__ field, const float *__restrict__ weights, int64_t n,
              int64_t n_labels, double *__restrict__ acc, unsigned *__restrict__ ext)
{
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    if (i0 >= n) return;
    int32_t cur = -1;
    double s0 = 0, s1 = 0, s2 = 0, mean = 0; unsigned kmax = 0, kmin = 0xffffffffu;
    auto flush = [&]() {
        if (cur < 1 || cur > n_labels) return;
        if (PASS == 1) {
            if (s0 != 0) atomicAdd(&acc[4 * cur + 0], s0);
            if (s1 != 0) atomicAdd(&acc[4 * cur + 1], s1);
            if (s2 != 0) atomicAdd(&acc[4 * cur + 2], s2);
            if (kmax != 0) atomicMax(&ext[2 * cur + 0], kmax);
            if (kmin != 0xffffffffu) atomicMin(&ext[2 * cur + 1], kmin);
        } else if (s2 != 0) atomicAdd(&acc[4 * cur + 3], s2);
    };
    for (int j = 0; j < 16 && i0 + j < n; j++) {
        const int32_t l = labels[i0 + j];
        if (l != cur) {
            flush();
            cur = l; s0 = s1 = s2 = 0; kmax = 0; kmin = 0xffffffffu;
            if (PASS == 2 && l >= 1 && l <= n_labels) mean = acc[4 * l + 2] / acc[4 * l + 1];
        }
        if (l < 1 || l > n_labels) continue;
        const float x = field[i0 + j], w = weights ? weights[i0 + j] : 1.f;
        const bool has = !(x != x);
        if (PASS == 1) {
            if (!(w != w)) s0 += (double)w;                   // np.nansum(weights) over the whole region
            if (has) {
                s1 += (double)w; s2 += (double)w * (double)x;
                if (w > 0.f) { const unsigned k = ls_key(x); kmax = max(kmax, k); kmin = min(kmin, k); }
            }
        } else if (has) { const double d = (double)x - mean; s2 += (double)w * (d * d); }
    }
    flush();
}

__global__ void __launch_bounds__(256)
k_label_stats_finish(const double *__restrict__ acc, const unsigned *__restrict__ ext, int64_t n_labels, double *__restrict__ out)
{
    const int64_t l = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + 1;
    if (l > n_labels) return;
    const double w_all = acc[4 * l], w_val = acc[4 * l + 1], nan = __longlong_as_double(0x7ff8000000000000ll);
    double *o = out + 6 * (l - 1);
    o[0] = w_all; o[1] = w_val;
    o[2] = acc[4 * l + 2] / w_val;                            // 0 / 0 = NaN where nothing carries weight
    o[3] = sqrt(acc[4 * l + 3] / w_val);
    o[4] = ext[2 * l] ? (double)ls_unkey(ext[2 * l]) : nan;
    o[5] = ext[2 * l + 1] != 0xffffffffu ? (double)ls_unkey(ext[2 * l + 1]) : nan;
}

extern "C" size_t tf_label_stats_workspace_bytes(int64_t n_labels)
{
    if (n_labels < 0) return 0;
    return tf_align_up((size_t)(n_labels + 1) * 32, 256) + tf_align_up((size_t)(n_labels + 1) * 8, 256) + 1024;
}

extern "C" int tf_label_stats(const int32_t *labels, const float *field, const float *weights, int64_t n, int64_t n_labels,
                              double *out, void *ws, size_t ws_bytes, void *stream)
{
    TF_REQUIRE(labels && field && ws && n > 0 && n_labels >= 0, "tf_label_stats: bad arguments");
    if (n_labels == 0) return TF_OK;
    TF_REQUIRE(out, "tf_label_stats: null output");
    hipStream_t s = (hipStream_t)stream;
    TfArena ar(ws, ws_bytes);
    double *acc = ar.take<double>(4 * (n_labels + 1));
    unsigned *ext = ar.take<unsigned>(2 * (n_labels + 1));
    if (!ar.ok()) { tf_set_error("tf_label_stats: workspace too small"); return TF_ENOMEM; }
    TF_CHECK_HIP(hipMemsetAsync(acc, 0, (size_t)(n_labels + 1) * 32, s));
    TF_CHECK_HIP(hipMemsetAsync(ext, 0, (size_t)(n_labels + 1) * 8, s));
    {
        // min keys start at all-ones: every second 32-bit word
        TF_CHECK_HIP(hipMemset2DAsync((char *)ext + 4, 8, 0xff, 4, (size_t)(n_labels + 1), s));
    }
    const dim3 grid((unsigned)((n + 4095) / 4096));
    hipLaunchKernelGGL(k_label_stats<1>, grid, dim3(256), 0, s, labels, field, weights, n, n_labels, acc, ext);
    hipLaunchKernelGGL(k_label_stats<2>, grid, dim3(256), 0, s, labels, field, weights, n, n_labels, acc, ext);
    hipLaunchKernelGGL(k_label_stats_finish, dim3((unsigned)((n_labels + 255) / 256)), dim3(256), 0, s, (const double *)acc,
                       (const unsigned *)ext, n_labels, out);
    TF_CHECK_LAUNCH();
    return TF_OK;
}
