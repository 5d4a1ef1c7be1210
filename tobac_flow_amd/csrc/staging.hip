// Host containers <-> HBM (include/tobac_flow_hip.h, "host staging"): what the reference's numpy / xarray containers cost a
// drop-in caller is PCIe time, and pageable memory moves at a third of the link's rate.  This file is the plumbing that keeps
// the link busy:
//   * a pool of PINNED host blocks -- made from huge pages touched by the host threads and registered with the runtime (22 ms
//     for 1.88 GB where hipHostMalloc takes 250 ms), kept by size class and handed out again: results are downloaded straight
//     into such a block, which the Python layer wraps as the numpy array it returns (no second host copy);
//   * tf_upload: a pageable source is copied by a pool of host threads, chunk by chunk, into a ring of pinned slots, each chunk
//     followed at once by its own asynchronous DMA -- host memcpy and DMA pipelined, the call returns when the source has been
//     read; a pinned source (a block of the pool) goes out in one DMA;
//   * a 128-bit content checksum that the host threads compute while they copy (or alone: tf_hash_host) and a kernel computes
//     on device memory (tf_hash_dev), word for word the same function: the Python layer keys its cache of device twins with
//     it, so that an array presented again -- the same object, or another temporary with the same content (`wvd - swd`
//     evaluated twice by scripts/dcc_detect_goes.py:227,241) -- is recognised by its content, not trusted by its address.
// Reference: the containers of tobac_flow/decorators.py:21-61, scripts/dcc_detect_goes.py:164-303 (numpy / DataArray in and
// out of every entry point); SURVEY.md section 8(f) rank 4 ("pinned-host staging to HBM").
#include "tf_common.h"
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <string.h>
#include <thread>
#include <vector>
#include <sched.h>
#include <sys/mman.h>

typedef unsigned long long u64;

// ---- the checksum -------------------------------------------------------------------------------------------------------
// The buffer as 16-byte blocks (w0, w1) of little-endian 64-bit words, the last block zero-padded; block i contributes
//   fold(mul128(w0 ^ a_i, w1 ^ b_i))  to h[0]  and  fold(mul128(w0 ^ c_i, w1 ^ d_i))  to h[1]      (sums mod 2^64)
// with fold(hi, lo) = hi ^ lo and the four key streams k_i = K + (i + 1) * STEP (odd steps).  Sums commute: any split of the
// blocks over host threads or GPU lanes gives the same two words.  One 64 x 64 -> 128 multiply per 8 bytes: memory-bound on
// both sides.  The byte length is folded in at the end.  (The NH / "mum" construction with counter keys: not cryptographic --
// it guards against a mutated or recycled array, not against an adversary.)
#define TF_HK_A 0x9E3779B97F4A7C15ull
#define TF_HK_B 0xC2B2AE3D27D4EB4Full
#define TF_HK_C 0xD6E8FEB86659FD93ull
#define TF_HK_D 0xA0761D6478BD642Full
#define TF_HS_A 0xE7037ED1A0B428DBull
#define TF_HS_B 0x8EBC6AF09C88C6E3ull
#define TF_HS_C 0x589965CC75374CC3ull
#define TF_HS_D 0x1D8E4E27C47D124Full

__host__ __device__ static inline void tf_hash_block(u64 w0, u64 w1, u64 i, u64 &h0, u64 &h1)
{
    const u64 n = i + 1;
    const u64 a = w0 ^ (TF_HK_A + n * TF_HS_A), b = w1 ^ (TF_HK_B + n * TF_HS_B);
    const u64 c = w0 ^ (TF_HK_C + n * TF_HS_C), d = w1 ^ (TF_HK_D + n * TF_HS_D);
#if defined(__HIP_DEVICE_COMPILE__)
    h0 += (a * b) ^ __umul64hi(a, b);
    h1 += (c * d) ^ __umul64hi(c, d);
#else
    const unsigned __int128 p = (unsigned __int128)a * b, q = (unsigned __int128)c * d;
    h0 += (u64)p ^ (u64)(p >> 64);
    h1 += (u64)q ^ (u64)(q >> 64);
#endif
}
__host__ __device__ static inline void tf_hash_finish(u64 bytes, u64 &h0, u64 &h1)
{
    tf_hash_block(bytes, ~bytes, 0xFFFFFFFFFFFFull, h0, h1);
}

// blocks [i0, i1) of a host buffer of `bytes` bytes (the last block may be partial)
static void tf_hash_host_range(const unsigned char *p, size_t bytes, size_t i0, size_t i1, u64 &h0, u64 &h1)
{
    const size_t full = bytes / 16;
    size_t i = i0;
    for (const size_t e = std::min(i1, full); i < e; i++) {
        u64 w[2];
        memcpy(w, p + i * 16, 16);
        tf_hash_block(w[0], w[1], i, h0, h1);
    }
    if (i < i1 && i == full && bytes % 16) {
        u64 w[2] = {0, 0};
        memcpy(w, p + i * 16, bytes % 16);
        tf_hash_block(w[0], w[1], i, h0, h1);
    }
}

__global__ void __launch_bounds__(256)
k_hash128(const unsigned char *__restrict__ p, size_t bytes, u64 *__restrict__ out)
{
    const size_t full = bytes / 16, n_blocks = (bytes + 15) / 16;
    u64 h0 = 0, h1 = 0;
    const bool aligned = ((uintptr_t)p & 15) == 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_blocks; i += (size_t)gridDim.x * 256) {
        u64 w0 = 0, w1 = 0;
        if (i < full && aligned) {
            const ulonglong2 v = *(const ulonglong2 *)(p + i * 16);
            w0 = v.x; w1 = v.y;
        } else {
            const size_t n = i < full ? 16 : bytes % 16;
            unsigned char b[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            for (size_t k = 0; k < n; k++) b[k] = p[i * 16 + k];
            for (int k = 7; k >= 0; k--) { w0 = (w0 << 8) | b[k]; w1 = (w1 << 8) | b[8 + k]; }
        }
        tf_hash_block(w0, w1, i, h0, h1);
    }
    for (int off = 32; off > 0; off >>= 1) { h0 += __shfl_down(h0, off, 64); h1 += __shfl_down(h1, off, 64); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&out[0], h0); atomicAdd(&out[1], h1); }
}

// ---- host threads ---------------------------------------------------------------------------------------------------------
namespace {
struct Workers {
    std::mutex run_mu;                        // one job at a time
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> threads;
    std::function<void(int)> job;
    int n = 0, active = 0;
    unsigned long gen = 0;
    bool stop = false;

    static int wanted()
    {
        if (const char *e = getenv("TF_STAGING_THREADS")) { const int v = atoi(e); if (v >= 1) return std::min(v, 64); }
        int cpus = (int)std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) cpus = CPU_COUNT(&set);
        return std::max(1, std::min(cpus, 16));
    }
    void start()
    {
        if (n) return;
        n = wanted();
        for (int i = 0; i < n; i++) threads.emplace_back([this, i] { loop(i); });
    }
    void loop(int i)
    {
        unsigned long seen = 0;
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_work.wait(lk, [&] { return stop || gen != seen; });
            if (stop) return;
            seen = gen;
            lk.unlock();
            job(i);
            lk.lock();
            if (--active == 0) cv_done.notify_all();
        }
    }
    // fn(worker index) on every worker; returns when all are done
    void run(const std::function<void(int)> &fn)
    {
        std::lock_guard<std::mutex> one(run_mu);
        std::unique_lock<std::mutex> lk(mu);
        start();
        job = fn; active = n; gen++;
        cv_work.notify_all();
        cv_done.wait(lk, [&] { return active == 0; });
        job = nullptr;
    }
    ~Workers()
    {
        { std::lock_guard<std::mutex> lk(mu); stop = true; }
        cv_work.notify_all();
        for (auto &t : threads) if (t.joinable()) t.join();
    }
};
Workers &workers() { static Workers *w = new Workers(); return *w; }      // (never destroyed: threads may outlive static destructors)

// ---- pinned blocks ----------------------------------------------------------------------------------------------------------
struct HostPool {
    std::mutex mu;
    std::map<uintptr_t, size_t> live;                  // blocks handed out: base -> capacity
    std::multimap<size_t, void *> cached;              // blocks kept for reuse: capacity -> base
    size_t live_bytes = 0, cached_bytes = 0;
};
HostPool &pool() { static HostPool *p = new HostPool(); return *p; }
// A pinned block is made by mmap + MADV_HUGEPAGE, touched by the host threads, then registered with the runtime
// (hipHostRegister).  hipHostMalloc of 1.88 GB takes 250 - 275 ms on the box this was measured on (tools/microbench/pin_cost.hip:
// it faults and pins 460 000 4-KiB pages on one thread); the same block from 2-MiB pages, touched by 16 threads, takes 19 ms and
// registers in 3 ms, and the DMA engine writes it at the same 57 GB/s.  Without transparent huge pages the pages are still
// touched in parallel (137 + 33 ms).  If any step fails the block comes from hipHostMalloc as before.
struct PinnedHow { void *raw; size_t raw_len; bool registered; };
std::mutex g_how_mu;
std::map<uintptr_t, PinnedHow> g_how;
void *pinned_new(size_t cap)
{
    static const bool plain_env = getenv("TF_STAGING_HIPHOSTMALLOC") != nullptr;      // development switch: the runtime's own allocation
    const size_t huge = (size_t)2 << 20, len = cap + huge;
    if (!plain_env) {
        void *raw = mmap(nullptr, len, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (raw != MAP_FAILED) {
            unsigned char *q = (unsigned char *)(((uintptr_t)raw + huge - 1) & ~(uintptr_t)(huge - 1));
            (void)madvise(q, cap, MADV_HUGEPAGE);
            const size_t piece = (size_t)8 << 20;
            std::atomic<size_t> next(0);
            workers().run([&](int) {
                for (;;) {
                    const size_t o = next.fetch_add(piece);
                    if (o >= cap) break;
                    for (size_t i = o, e = std::min(cap, o + piece); i < e; i += 4096) q[i] = 0;
                }
            });
            if (hipHostRegister(q, cap, hipHostRegisterDefault) == hipSuccess) {
                std::lock_guard<std::mutex> lk(g_how_mu);
                g_how[(uintptr_t)q] = PinnedHow{raw, len, true};
                return q;
            }
            (void)hipGetLastError();
            munmap(raw, len);
        }
    }
    void *p = nullptr;
    if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    std::lock_guard<std::mutex> lk(g_how_mu);
    g_how[(uintptr_t)p] = PinnedHow{nullptr, 0, false};
    return p;
}
void pinned_delete(void *p)
{
    PinnedHow how{nullptr, 0, false};
    {
        std::lock_guard<std::mutex> lk(g_how_mu);
        auto it = g_how.find((uintptr_t)p);
        if (it != g_how.end()) { how = it->second; g_how.erase(it); }
    }
    if (how.registered) { (void)hipHostUnregister(p); munmap(how.raw, how.raw_len); }
    else (void)hipHostFree(p);
}
size_t size_class(size_t bytes)
{
    // small: powers of two from 64 KiB; large (> 64 MiB): multiples of 32 MiB (a 1.88 GB volume wastes < 2 %)
    if (bytes <= (64u << 20)) { size_t c = 64u << 10; while (c < bytes) c <<= 1; return c; }
    return tf_align_up(bytes, (size_t)32 << 20);
}

// ---- the ring of pinned slots a pageable upload is staged through -----------------------------------------------------------------
struct Ring {
    std::mutex mu;                              // one staged transfer at a time per process
    unsigned char *base = nullptr;
    size_t slot_bytes = 0;
    int n_slots = 0;
    std::vector<hipEvent_t> ev;                 // ev[s]: the DMA that last read slot s
    std::vector<char> used;
    int dev = -1;
};
Ring &ring() { static Ring *r = new Ring(); return *r; }
int ring_prepare(Ring &r, int dev)
{
    if (r.base && r.dev == dev) return TF_OK;
    if (r.base) {                               // another device: events are per device, the slots are not
        for (auto e : r.ev) (void)hipEventDestroy(e);
        r.ev.clear();
    }
    if (!r.base) {
        const char *e = getenv("TF_STAGING_CHUNK_MB");
        r.slot_bytes = (size_t)std::max(1, e ? atoi(e) : 8) << 20;
        r.n_slots = 32;
        void *p = pinned_new(r.slot_bytes * r.n_slots);
        if (!p) {
            tf_set_error("tf_upload: no pinned memory for the staging ring (%zu MiB)", (r.slot_bytes * r.n_slots) >> 20);
            return TF_EHIP;
        }
        r.base = (unsigned char *)p;
    }
    r.ev.resize(r.n_slots);
    r.used.assign(r.n_slots, 0);
    for (int s = 0; s < r.n_slots; s++) TF_CHECK_HIP(hipEventCreateWithFlags(&r.ev[s], hipEventDisableTiming));
    r.dev = dev;
    return TF_OK;
}
}  // namespace

extern "C" int tf_host_alloc(size_t bytes, void **ptr_out)
{
    TF_REQUIRE(ptr_out && bytes > 0, "tf_host_alloc: bad arguments");
    const size_t cap = size_class(bytes);
    HostPool &hp = pool();
    {
        std::lock_guard<std::mutex> lk(hp.mu);
        auto it = hp.cached.find(cap);
        if (it != hp.cached.end()) {
            void *p = it->second;
            hp.cached.erase(it);
            hp.cached_bytes -= cap;
            hp.live[(uintptr_t)p] = cap;
            hp.live_bytes += cap;
            *ptr_out = p;
            return TF_OK;
        }
    }
    void *p = pinned_new(cap);
    if (!p) {
        // the cache may hold what is missing: give it back and try once more
        std::vector<void *> drop;
        {
            std::lock_guard<std::mutex> lk(hp.mu);
            for (auto &kv : hp.cached) drop.push_back(kv.second);
            hp.cached.clear();
            hp.cached_bytes = 0;
        }
        for (void *q : drop) pinned_delete(q);
        p = pinned_new(cap);
    }
    if (!p) {
        tf_set_error("tf_host_alloc: no pinned block of %zu bytes (mmap + hipHostRegister and hipHostMalloc both failed)", cap);
        return TF_ENOMEM;
    }
    std::lock_guard<std::mutex> lk(hp.mu);
    hp.live[(uintptr_t)p] = cap;
    hp.live_bytes += cap;
    *ptr_out = p;
    return TF_OK;
}

extern "C" int tf_host_free(void *ptr)
{
    if (!ptr) return TF_OK;
    HostPool &hp = pool();
    std::unique_lock<std::mutex> lk(hp.mu);
    auto it = hp.live.find((uintptr_t)ptr);
    TF_REQUIRE(it != hp.live.end(), "tf_host_free: not a block of tf_host_alloc");
    const size_t cap = it->second;
    hp.live.erase(it);
    hp.live_bytes -= cap;
    hp.cached.emplace(cap, ptr);
    hp.cached_bytes += cap;
    // the pool keeps at most TF_PINNED_CACHE_GB (default 16) of FREE blocks: a process that once returned a 15 GB flow array
    // does not hold it pinned for ever; the largest go first (a block is cheap to make again: 14 ms per 1.88 GB)
    static const size_t keep = (size_t)((getenv("TF_PINNED_CACHE_GB") ? atof(getenv("TF_PINNED_CACHE_GB")) : 16.0) * 1e9);
    std::vector<void *> drop;
    while (hp.cached_bytes > keep && !hp.cached.empty()) {
        auto last = std::prev(hp.cached.end());
        drop.push_back(last->second);
        hp.cached_bytes -= last->first;
        hp.cached.erase(last);
    }
    lk.unlock();
    for (void *q : drop) pinned_delete(q);
    return TF_OK;
}

extern "C" int tf_host_is_pinned(const void *ptr, size_t bytes)
{
    if (!ptr) return 0;
    HostPool &hp = pool();
    std::lock_guard<std::mutex> lk(hp.mu);
    auto it = hp.live.upper_bound((uintptr_t)ptr);
    if (it == hp.live.begin()) return 0;
    --it;
    return (uintptr_t)ptr >= it->first && (uintptr_t)ptr + bytes <= it->first + it->second ? 1 : 0;
}

extern "C" int tf_host_pool_stats(int64_t *live_bytes, int64_t *cached_bytes)
{
    HostPool &hp = pool();
    std::lock_guard<std::mutex> lk(hp.mu);
    if (live_bytes) *live_bytes = (int64_t)hp.live_bytes;
    if (cached_bytes) *cached_bytes = (int64_t)hp.cached_bytes;
    return TF_OK;
}

// cached (free) blocks of the size class `bytes` falls into
extern "C" int tf_host_pool_spare(size_t bytes)
{
    if (!bytes) return 0;
    HostPool &hp = pool();
    std::lock_guard<std::mutex> lk(hp.mu);
    return (int)hp.cached.count(size_class(bytes));
}

extern "C" int tf_host_pool_trim(size_t keep_bytes)
{
    HostPool &hp = pool();
    std::vector<void *> drop;
    {
        std::lock_guard<std::mutex> lk(hp.mu);
        while (hp.cached_bytes > keep_bytes && !hp.cached.empty()) {
            auto it = std::prev(hp.cached.end());             // largest first
            drop.push_back(it->second);
            hp.cached_bytes -= it->first;
            hp.cached.erase(it);
        }
    }
    for (void *q : drop) pinned_delete(q);
    return TF_OK;
}

extern "C" int tf_hash_host(const void *src, size_t bytes, uint64_t *hash_out)
{
    TF_REQUIRE(src && hash_out && bytes > 0, "tf_hash_host: bad arguments");
    const unsigned char *p = (const unsigned char *)src;
    const size_t n_blocks = (bytes + 15) / 16;
    u64 h0 = 0, h1 = 0;
    if (bytes < (4u << 20)) {
        tf_hash_host_range(p, bytes, 0, n_blocks, h0, h1);
    } else {
        const size_t piece = (size_t)1 << 16;                 // blocks per piece (1 MiB)
        std::atomic<size_t> next(0);
        std::mutex acc;
        workers().run([&](int) {
            u64 a = 0, b = 0;
            for (;;) {
                const size_t i0 = next.fetch_add(piece);
                if (i0 >= n_blocks) break;
                tf_hash_host_range(p, bytes, i0, std::min(n_blocks, i0 + piece), a, b);
            }
            std::lock_guard<std::mutex> lk(acc);
            h0 += a; h1 += b;
        });
    }
    tf_hash_finish(bytes, h0, h1);
    hash_out[0] = h0; hash_out[1] = h1;
    return TF_OK;
}

extern "C" int tf_hash_dev(const void *src_dev, size_t bytes, uint64_t *hash_out_host, void *stream)
{
    TF_REQUIRE(src_dev && hash_out_host && bytes > 0, "tf_hash_dev: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    u64 *d = nullptr;
    TF_CHECK_HIP(hipMalloc((void **)&d, 16));
    hipError_t e = hipMemsetAsync(d, 0, 16, s);
    if (e == hipSuccess) {
        const size_t n_blocks = (bytes + 15) / 16;
        const unsigned grid = (unsigned)std::min<size_t>((n_blocks + 255) / 256, 256 * 8);
        hipLaunchKernelGGL(k_hash128, dim3(grid), dim3(256), 0, s, (const unsigned char *)src_dev, bytes, d);
        e = hipGetLastError();
    }
    u64 h[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(d);
    if (e != hipSuccess) { tf_set_error("tf_hash_dev: %s", hipGetErrorString(e)); return TF_EHIP; }
    tf_hash_finish(bytes, h[0], h[1]);
    hash_out_host[0] = h[0]; hash_out_host[1] = h[1];
    return TF_OK;
}

extern "C" int tf_upload(void *dst_dev, const void *src_host, size_t bytes, uint64_t *hash_out, void *stream)
{
    TF_REQUIRE(dst_dev && src_host && bytes > 0, "tf_upload: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const unsigned char *src = (const unsigned char *)src_host;
    if (tf_host_is_pinned(src_host, bytes)) {
        // a block of the pool: the DMA engine reads it where it lies
        TF_CHECK_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, s));
        if (hash_out) return tf_hash_host(src_host, bytes, hash_out);
        return TF_OK;
    }
    if (bytes < (1u << 20)) {                                   // small: the runtime's own staging does as well
        TF_CHECK_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, s));
        TF_CHECK_HIP(hipStreamSynchronize(s));                  // (a pageable source must not be touched by the caller before the copy has read it)
        if (hash_out) return tf_hash_host(src_host, bytes, hash_out);
        return TF_OK;
    }
    // (Measured and not adopted, round 6: hipHostRegister of the SOURCE + one DMA instead of the ring -- 37 ms against 40 ms per
    // 1.88 GB when the source is backed by huge pages, as numpy's own large arrays are; 60 ms when it is not, e.g. an array that
    // came out of torch's CPU allocator.  The ring costs the same whatever the source's pages are.)
    int dev = 0;
    TF_CHECK_HIP(hipGetDevice(&dev));
    Ring &r = ring();
    std::lock_guard<std::mutex> one(r.mu);
    if (const int rc = ring_prepare(r, dev)) return rc;
    const size_t chunk = r.slot_bytes, n_chunks = (bytes + chunk - 1) / chunk;
    std::atomic<size_t> next(0);
    std::atomic<int> failed(0);
    std::mutex acc;
    u64 h0 = 0, h1 = 0;
    const bool want_hash = hash_out != nullptr;
    workers().run([&](int) {
        if (hipSetDevice(dev) != hipSuccess) { failed = 1; return; }
        u64 a = 0, b = 0;
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= n_chunks || failed.load()) break;
            const int slot = (int)(k % (size_t)r.n_slots);
            const size_t off = k * chunk, len = std::min(chunk, bytes - off);
            // chunks are claimed in order, so the previous user of this slot (chunk k - n_slots) was claimed -- and its DMA
            // enqueued or about to be -- before this one: wait until that worker has recorded its event, then for the DMA
            if (k >= (size_t)r.n_slots) {
                while (__atomic_load_n(&r.used[slot], __ATOMIC_ACQUIRE) != (char)(1 + ((k / r.n_slots - 1) & 1))) {
                    if (failed.load()) return;
                    std::this_thread::yield();
                }
                if (hipEventSynchronize(r.ev[slot]) != hipSuccess) { failed = 1; return; }
            }
            unsigned char *stage = r.base + (size_t)slot * r.slot_bytes;
            // copy (and checksum, while the lines are in cache) in pieces of 256 KiB
            for (size_t o = 0; o < len; o += (size_t)256 << 10) {
                const size_t m = std::min((size_t)256 << 10, len - o);
                memcpy(stage + o, src + off + o, m);
                if (want_hash) tf_hash_host_range(src, bytes, (off + o) / 16, (off + o + m + 15) / 16, a, b);
            }
            if (hipMemcpyAsync((unsigned char *)dst_dev + off, stage, len, hipMemcpyHostToDevice, s) != hipSuccess ||
                hipEventRecord(r.ev[slot], s) != hipSuccess) { failed = 1; return; }
            __atomic_store_n(&r.used[slot], (char)(1 + ((k / r.n_slots) & 1)), __ATOMIC_RELEASE);
        }
        if (want_hash) { std::lock_guard<std::mutex> lk(acc); h0 += a; h1 += b; }
    });
    // the ring is reused by the next transfer: its DMAs must have read the slots (the caller's stream stays asynchronous for
    // everything enqueued after this call; only this host thread waits, for the last few chunks' DMA)
    hipError_t e = hipSuccess;
    for (int sl = 0; sl < r.n_slots && (size_t)sl < n_chunks; sl++) {
        const hipError_t e1 = hipEventSynchronize(r.ev[sl]);
        if (e1 != hipSuccess) e = e1;
    }
    std::fill(r.used.begin(), r.used.end(), 0);
    if (failed.load() || e != hipSuccess) {
        (void)hipGetLastError();
        tf_set_error("tf_upload: a staged copy failed (%s)", e != hipSuccess ? hipGetErrorString(e) : "hipMemcpyAsync / hipEventRecord on a worker thread");
        return TF_EHIP;
    }
    if (want_hash) { tf_hash_finish(bytes, h0, h1); hash_out[0] = h0; hash_out[1] = h1; }
    return TF_OK;
}

extern "C" int tf_download(void *dst_host, const void *src_dev, size_t bytes, void *stream)
{
    TF_REQUIRE(dst_host && src_dev && bytes > 0, "tf_download: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    // a pinned destination (a block of the pool) is written by the DMA engine at the link's rate; any other destination goes
    // through the runtime's staging.  Either way the call returns when the data is there.
    TF_CHECK_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, s));
    TF_CHECK_HIP(hipStreamSynchronize(s));
    return TF_OK;
}
