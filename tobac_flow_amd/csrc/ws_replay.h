// The reference heap's pop order of equal-valued markers: the host replays of TF_WS_REFERENCE_ORDER (include/tobac_flow_hip.h).
// Pure C++ (no HIP): included by watershed.hip and by tools/replay_check.cpp, which compares the three forms below on
// random instances and times them on a config-F-sized one -- on the CPU, without a GPU box.
//
// The reference pushes every marker with age 0 (_watershed.pyx:278-284), so markers of equal value compare equal
// (`smaller`, :161-164) and pop in an order that is a by-product of its binary heap's array mechanics (:67-152): where
// the sift-up / sift-down loops happen to leave them.  That order depends on EVERY push and pop before it -- a marker
// of another value pushed in between, or a pixel flooded from a lower marker, moves tied items up and down the array
// (two tied markers A, B and one smaller item X pushed between them pop X, B, A; without X: A, B) -- so it cannot be
// derived from the tied markers alone: the mechanics have to be replayed with every item in place.  These routines are
// that replay over the sub-graph of the compact flood graph the device exports (k_ws_sub_*): the same push / pop / sift
// rules, item for item, with the keys only (no labels are computed here -- the device flood does that, with the pop rank
// returned here as the last component of its chain comparison).  They stop as soon as every marker whose rank can matter
// has popped: at the first top item above `vmax`, the largest marker value at which the device found chains that tie down
// to equal-valued markers of different labels.
// Ids are sub-graph ids (nQ of them): `val` their value keys, `nbr` the rows of floodable out-neighbours (rows of ids the
// replay can pop are filled, see k_ws_sub_export), seeds carry -1 when nobody floods from them (ballast the heap needs).
// rank[id] = pop rank of a marker, -1 if it did not pop; *n_ranked = markers popped.  Return: the pops, -1: out of memory.
#pragma once
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <utility>
#include <vector>

#ifndef WSR_ALLOC            /* large scratch: watershed.hip routes these to its pool of host buffers */
#define WSR_ALLOC(bytes) malloc(bytes)
#define WSR_FREE(p) free(p)
#endif

typedef unsigned long long wsr_u64;

// ---- seed entries as the device writes them (k_ws_seed_list) and the dense replay keeps them: 8 bytes -----------------
//   high word  the ordered value key; 0xFFFFFFFF for an item that can never pop before the replay ends (LARGE: value above
//              vmax, or equal to it with age != 0): all LARGE items are one anonymous value, equal to itself
//   low word   seeds (age 0): 0x80000000 | (id + 1), id = sub-graph id or -1;  pushed pixels: their age (1 .. 2^31 - 1; the id
//              of the pixel pushed with age a is kept in a table)
#define WSR_LARGE 0xFFFFFFFFFFFFFFFFull
static inline wsr_u64 wsr_seed_entry(unsigned v, int id, unsigned vmax) {
    return v > vmax ? WSR_LARGE : (((wsr_u64)v << 32) | 0x80000000ull | (wsr_u64)(unsigned)(id + 1));
}
// `smaller` (_watershed.pyx:161-164): value, then age; two seeds of one value are EQUAL (both age 0)
static inline bool wsr_smaller(wsr_u64 a, wsr_u64 b) {
    const unsigned ha = (unsigned)(a >> 32), hb = (unsigned)(b >> 32);
    if (ha != hb) return ha < hb;
    const unsigned la = (unsigned)a, lb = (unsigned)b;
    const unsigned ea = (la >> 31) ? 0u : la, eb = (lb >> 31) ? 0u : lb;          // effective age
    return ea < eb;
}

struct WsRefItem { unsigned v; int32_t age; int32_t id; };
static inline bool ws_ref_smaller(const WsRefItem &a, const WsRefItem &b) { return a.v != b.v ? a.v < b.v : a.age < b.age; }

// ---- PLAIN form: every seed pushed, every item kept as the reference keeps it (value, age, id): the statement of the
// mechanics the other two are checked against (TF_WS_REFERENCE_DENSE=2)
static int64_t ws_reference_ranks_plain(int64_t M, const wsr_u64 *seeds, const unsigned *seed_val_all, int64_t nQ, const unsigned *val,
                                        const int *nbr, int n_nbr, unsigned vmax, int *rank, int *n_ranked_out)
{
    // `seeds`: the 8-byte entries (LARGE ones have lost their value: any value above vmax does, they never pop);
    // seed_val_all: optional true values (the harness passes them: the plain form then orders the large items exactly like
    // the reference does, which must not matter)
    WsRefItem *h = (WsRefItem *)WSR_ALLOC((size_t)(M + nQ + 1) * sizeof(WsRefItem));     // every pixel is pushed at most once
    uint8_t *state = (uint8_t *)calloc((size_t)(nQ > 0 ? nQ : 1), 1);                      // 1: already pushed
    if (!h || !state) { if (h) WSR_FREE(h); free(state); return -1; }
    int64_t items = 0;
    auto push = [&](const WsRefItem &e) {                                              // _watershed.pyx:120-152
        int64_t child = items;
        h[child] = e;
        items += 1;
        while (child > 0) {
            const int64_t parent = (child + 1) / 2 - 1;
            if (ws_ref_smaller(h[child], h[parent])) { const WsRefItem t = h[parent]; h[parent] = h[child]; h[child] = t; child = parent; }
            else break;
        }
    };
    for (int64_t i = 0; i < M; i++) {                                                  // :278-284, marker_locations order
        const wsr_u64 e = seeds[i];
        const unsigned v = seed_val_all ? seed_val_all[i] : (unsigned)(e >> 32);
        push(WsRefItem{v, 0, e == WSR_LARGE ? -1 : (int)((unsigned)e & 0x7fffffffu) - 1});
    }
    for (int64_t i = 0; i < nQ; i++) rank[i] = -1;
    int64_t age = 1, popped = 0;
    int n_ranked = 0;
    while (items > 0) {
        const WsRefItem e = h[0];
        // all markers of value <= vmax pop before the first item that is above vmax or a flooded pixel AT vmax
        if (e.v > vmax || (e.v == vmax && e.age != 0)) break;
        items -= 1;                                                                    // :67-111
        if (items > 0) {
            h[0] = h[items];
            int64_t i = 0, smallest = 0;
            for (;;) {
                const int64_t l = 2 * i + 1, r = 2 * i + 2;
                if (l < items) {
                    if (ws_ref_smaller(h[l], h[i])) smallest = l;
                    if (r < items && ws_ref_smaller(h[r], h[smallest])) smallest = r;
                } else break;
                if (smallest == i) break;
                const WsRefItem t = h[i]; h[i] = h[smallest]; h[smallest] = t;
                i = smallest;
            }
        }
        popped++;
        if (e.id < 0) continue;
        if (e.age == 0) rank[e.id] = n_ranked++;                                       // a marker: its pop rank
        const int *np = nbr + (int64_t)e.id * n_nbr;
        for (int k = 0; k < n_nbr; k++) {                                              // :308-341 (mask / already labelled: not pushed)
            const int n = np[k];
            if (n < 0 || state[n]) continue;
            state[n] = 1;
            age += 1;
            push(WsRefItem{val[n], (int32_t)age, n});                                  // Py_ssize_t -> int32 store, :338
        }
    }
    *n_ranked_out = n_ranked;
    WSR_FREE(h); free(state);
    return popped;
}

// ---- SPARSE form (round 3): THE SAME REPLAY WITHOUT THE LARGE ITEMS.  Call an item SMALL when the loop above would still
// pop it -- v < vmax, or v == vmax with age 0 -- and LARGE otherwise.  A small item compares smaller than every large one,
// the heap order keeps every descendant of a large node large, and the loop stops at the first large top: so whatever the
// large items do among themselves (which of them a sift moves where) never moves a small item, and the trajectory of the
// small items depends on the large ones only through the POSITIONS they occupy.  The replay therefore keeps the small items
// alone -- an occupancy bitmap over the heap positions plus a position -> item table -- and treats every other position
// below `items` as an anonymous large item: pushing a large item just lengthens the heap; a small item sifting up walks
// through unoccupied ancestors without a comparison; a large item sifting down from the root follows the smaller of its
// small children until both children are large.  With S small seeds out of M (S / M = 0.3 % on detect_anvils fields) the
// replay costs O(S log M) bit tests instead of M pushes, and only the small seeds cross PCIe.
struct WsPosMap {                                                       // open addressing, linear probing, backward-shift deletion
    int64_t *key = nullptr; WsRefItem *val = nullptr; size_t cap = 0, n = 0;
    ~WsPosMap() { free(key); free(val); }
    static size_t hash(int64_t k) { uint64_t x = (uint64_t)k * 0x9E3779B97F4A7C15ull; return (size_t)(x ^ (x >> 29)); }
    bool init(size_t want) {
        cap = 1024; while (cap < want) cap <<= 1;
        key = (int64_t *)malloc(cap * sizeof(int64_t)); val = (WsRefItem *)malloc(cap * sizeof(WsRefItem));
        if (!key || !val) return false;
        for (size_t i = 0; i < cap; i++) key[i] = -1;
        n = 0;
        return true;
    }
    bool grow() {
        WsPosMap b; if (!b.init(cap * 2)) return false;
        for (size_t i = 0; i < cap; i++) if (key[i] >= 0) b.put_nogrow(key[i], val[i]);
        std::swap(key, b.key); std::swap(val, b.val); std::swap(cap, b.cap); std::swap(n, b.n);
        return true;
    }
    void put_nogrow(int64_t k, const WsRefItem &v) {
        size_t i = hash(k) & (cap - 1);
        while (key[i] >= 0 && key[i] != k) i = (i + 1) & (cap - 1);
        if (key[i] < 0) n++;
        key[i] = k; val[i] = v;
    }
    bool put(int64_t k, const WsRefItem &v) { if ((n + 1) * 2 > cap && !grow()) return false; put_nogrow(k, v); return true; }
    WsRefItem get(int64_t k) const {                                  // the key must be present
        size_t i = hash(k) & (cap - 1);
        while (key[i] != k) i = (i + 1) & (cap - 1);
        return val[i];
    }
    void erase(int64_t k) {                                           // the key must be present
        size_t i = hash(k) & (cap - 1);
        while (key[i] != k) i = (i + 1) & (cap - 1);
        size_t j = i;
        for (;;) {
            j = (j + 1) & (cap - 1);
            if (key[j] < 0) break;
            const size_t h = hash(key[j]) & (cap - 1);
            // the entry at j may move to the hole at i unless its home slot lies cyclically in (i, j]
            if ((i <= j) ? (i < h && h <= j) : (i < h || h <= j)) continue;
            key[i] = key[j]; val[i] = val[j]; i = j;
        }
        key[i] = -1; n--;
    }
};

static int64_t ws_reference_ranks_sparse(int64_t M, int64_t S, const long long *sk, const unsigned *sval, const int *sid, int64_t nQ,
                                         const unsigned *val, const int *nbr, int n_nbr, unsigned vmax, int *rank, int *n_ranked_out,
                                         double *phase_ms = nullptr)
{
    auto now_ms = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    const double t_start = now_ms();
    // an item = (key, id) with key = (v << 32) | age: `smaller` (:161-164) is the order of the keys.  LARGE = the all-ones
    // key: larger than every small item (value keys stop at that of +inf, NaN fields are refused), equal to itself -- a
    // sift never swaps two of them, which is all the replay needs to know about the large items.
    typedef wsr_u64 u64;
    struct Item { u64 key; int32_t id; int32_t pad; };
    const Item LARGE{~0ull, -1, 0};
    const int64_t max_items = M + nQ + 1;
    // WHERE THE SMALL ITEMS LIVE (round 4).  Most of them are one and the same item: the MODAL BALLAST seed -- the value most
    // small seeds share, age 0, nobody floods from it (the interior of the saturated cores of a detect_anvils field: 95 % of
    // the small seeds).  A position that holds it is a bit in `occ` and nothing else; every other small item (seeds with an id,
    // other values, pushed pixels) has its bit in `expl` as well and its item in a hash table keyed by position.  Round 3 kept
    // a 16-byte item per position for the top 4 S positions (128 MB for 1.9 M seeds, cold) and hashed the rest -- more than half
    // of the seeds, because seeds come in runs of consecutive numbers whose chains share every ancestor but the last few.
    uint64_t *occ = (uint64_t *)calloc((size_t)((max_items + 63) / 64), sizeof(uint64_t));    // position holds a small item
    uint64_t *expl = (uint64_t *)calloc((size_t)((max_items + 63) / 64), sizeof(uint64_t));   // ... which is not the modal ballast item
    uint8_t *state = (uint8_t *)calloc((size_t)(nQ > 0 ? nQ : 1), 1);                          // 1: already pushed
    WsPosMap deep;
    if (!occ || !expl || !state || !deep.init(1 << 16)) { free(occ); free(expl); free(state); return -1; }
    u64 modal = ~0ull;                                                                          // key of the modal ballast item (none: ~0)
    {
        unsigned sample[257]; int ns = 0;
        const int64_t step = S / 256 > 0 ? S / 256 : 1;
        for (int64_t j = 0; j < S && ns < 256; j += step) if (sid[j] < 0) sample[ns++] = sval[j];
        int best = 0;
        for (int a = 0; a < ns; a++) { int c = 0; for (int b = 0; b < ns; b++) c += sample[b] == sample[a]; if (c > best) { best = c; modal = (u64)sample[a] << 32; } }
    }
    bool oom = false;
    unsigned long long path_key = 0;     // the saved path of the pop loop (below): value class it follows, and whether it still holds
    bool path_valid = false;
    // the bitmaps cover ALL positions: a small item on its way up walks through unoccupied ancestors on bit tests alone
    auto has = [&](int64_t p) { return (occ[p >> 6] >> (p & 63)) & 1ull; };
    auto load = [&](int64_t p) -> Item {
        if (!has(p)) return LARGE;
        if (!((expl[p >> 6] >> (p & 63)) & 1ull)) return Item{modal, -1, 0};
        const WsRefItem e = deep.get(p);
        return Item{((u64)e.v << 32) | (uint32_t)e.age, e.id, 0};
    };
    auto store = [&](int64_t p, const Item &e) {
        const bool was = has(p), is = e.key != ~0ull;
        if (is != was) occ[p >> 6] ^= 1ull << (p & 63);
        const bool was_x = (expl[p >> 6] >> (p & 63)) & 1ull, is_x = is && !(e.key == modal && e.id < 0);
        if (is_x != was_x) expl[p >> 6] ^= 1ull << (p & 63);
        if (!is_x) { if (was_x) deep.erase(p); return; }
        if (!deep.put(p, WsRefItem{(unsigned)(e.key >> 32), (int32_t)(e.key & 0xffffffffu), e.id})) oom = true;
    };
    // _watershed.pyx:120-152 for a small item entering at position `child` (a large one only lengthens the heap)
    bool hint_on = true;                 // (the build only)
    uint64_t hint_c1 = 0; int hint_depth = -1, hint_d = 0;
    auto push_small = [&](int64_t child, const Item &e) {
        if (child > 0 && !has((child + 1) / 2 - 1)) {
            // The small items are an ancestor-closed set, so the unoccupied ancestors of `child` are the LOWER part of its
            // chain: the item rises through all of them without a comparison.  Find the first unoccupied position of the
            // chain from the root down -- ~21 tests in the hot top of the bitmap instead of ~8 cold ones from the bottom
            // (the sparse build was 0.8 us per seed this way round).
            const uint64_t c1 = (uint64_t)child + 1;
            const int depth = 63 - __builtin_clzll(c1);
            int d = 0;
            if (hint_on && hint_depth == depth) {
                // during the build (no pop in between) the chain of the previous seed is occupied down to the level it
                // reached, and seeds come in runs of consecutive numbers: the levels this chain shares with that one need no test
                const uint64_t x = c1 ^ hint_c1;
                const int shared = x ? depth - (63 - __builtin_clzll(x)) : depth + 1;      // levels 0 .. shared - 1 coincide
                d = shared < hint_d + 1 ? shared : hint_d + 1;
                if (d > depth) d = depth;
            }
            while (d < depth && has((int64_t)(c1 >> (depth - d)) - 1)) d++;
            child = (int64_t)(c1 >> (depth - d)) - 1;
            hint_c1 = c1; hint_depth = depth; hint_d = d;
        } else hint_depth = -1;
        while (child > 0) {
            const int64_t parent = (child + 1) / 2 - 1;
            if (has(parent)) {
                const Item pe = load(parent);
                if (!(e.key < pe.key)) break;
                if (pe.key == path_key) path_valid = false;             // a seed of the tracked tree moves down: drop the saved path
                store(child, pe);
                store(parent, LARGE);                                    // (rewritten by the next step or by the final store)
            }
            child = parent;                                              // a large parent moves down: nothing to record
        }
        store(child, e);
    };
    const double t_alloc = now_ms();
    // (round 5: the bitmap words of a seed's position, of its parent and of its grandparent -- the cold bottom of its chain; the
    // levels above are shared with its neighbours and stay cached -- are requested a few seeds ahead: the build was paced by
    // those misses, two 56 MB bitmaps per 16 x 5424^2 window)
    constexpr int64_t AHEAD = 12;
    for (int64_t j = 0; j < S; j++) {
        if (j + AHEAD < S) {
            int64_t q = (int64_t)sk[j + AHEAD];
            for (int lvl = 0; lvl < 3 && q > 0; lvl++, q = (q + 1) / 2 - 1) {
                __builtin_prefetch(&occ[q >> 6], 1, 1);
                __builtin_prefetch(&expl[q >> 6], 1, 1);
            }
        }
        push_small((int64_t)sk[j], Item{(u64)sval[j] << 32, sid[j], 0});     // seed k enters at position k, age 0
    }
    hint_on = false;
    if (phase_ms) phase_ms[0] = now_ms() - t_start;
    if (getenv("WSR_DEBUG")) fprintf(stderr, "sparse: scratch %.1f ms, %lld pushes %.1f ms, table of items with an id / another value: %zu entries\n", t_alloc - t_start, (long long)S, now_ms() - t_alloc, deep.n);
    int64_t items = M;
    for (int64_t i = 0; i < nQ; i++) rank[i] = -1;
    int64_t age = 1, popped = 0;
    int n_ranked = 0;
    // The tree of seeds of the root's value (see the dense form below): a larger item taken from the end sinks along the
    // path "left child if it is such a seed, else the right one if it is" to a leaf of that tree and every seed on the path
    // moves up one node -- which changes nothing for seeds nobody floods from (one and the same item).  The path is kept
    // between pops; only the seeds with an id on it are moved.  (Round 4: a pop walked the ~21 levels of 1.9 M equal seeds
    // through a 128 MB array, 290 ns each.)
    std::vector<int64_t> path;       // path[0] = 0
    std::vector<int> rel;            // levels d >= 1 of path items with an id, ascending
    auto sift_down_from = [&](int64_t i, const Item &x) {                // :67-111 from the hole at i
        for (;;) {
            const int64_t l = 2 * i + 1, r = 2 * i + 2;
            if (l >= items) break;
            int64_t smallest = i;
            Item cur = x;
            const Item le = load(l);
            if (le.key < cur.key) { smallest = l; cur = le; }
            if (r < items) { const Item re = load(r); if (re.key < cur.key) { smallest = r; cur = re; } }
            if (smallest == i) break;
            store(i, cur);
            i = smallest;
        }
        store(i, x);
    };
    while (items > 0 && !oom) {
        const Item e = load(0);
        if (e.key == ~0ull) break;                                       // a large top ends the replay
        items -= 1;                                                      // :67-111
        if (items > 0) {
            const Item x = load(items);
            store(items, LARGE);
            const bool seed_root = (e.key & 0xffffffffull) == 0;
            if (seed_root && x.key == e.key) store(0, x);               // equal to both children at most: stays at the root
            else if (seed_root) {
                if (!path_valid || path_key != e.key || path.back() >= items) { path.clear(); path.push_back(0); rel.clear(); path_key = e.key; path_valid = true; }
                for (;;) {
                    const int64_t i = path.back(), l = 2 * i + 1, r = l + 1;
                    int64_t c = -1;
                    Item ce = LARGE;
                    if (l < items && has(l)) { ce = load(l); if (ce.key == e.key) c = l; }
                    if (c < 0 && r < items && has(r)) { ce = load(r); if (ce.key == e.key) c = r; }
                    if (c < 0) break;
                    path.push_back(c);
                    if (ce.id >= 0) rel.push_back((int)path.size() - 1);
                }
                const int L = (int)path.size() - 1;
                if (L >= 1) {
                    store(0, Item{e.key, -1, 0});
                    size_t w = 0;
                    for (size_t q = 0; q < rel.size(); q++) {
                        const int d = rel[q];
                        store(path[d - 1], load(path[d]));
                        store(path[d], Item{e.key, -1, 0});
                        if (d - 1 >= 1) rel[w++] = d - 1;
                    }
                    rel.resize(w);
                }
                sift_down_from(path[L], x);
                if (L >= 1) path.pop_back(); else path_valid = false;
            } else { path_valid = false; sift_down_from(0, x); }
        } else store(0, LARGE);
        popped++;
        if (e.id < 0) continue;
        if ((e.key & 0xffffffffull) == 0) rank[e.id] = n_ranked++;      // a marker (age 0): its pop rank
        const int *np = nbr + (int64_t)e.id * n_nbr;
        for (int k = 0; k < n_nbr; k++) {                                // :308-341
            const int n = np[k];
            if (n < 0 || state[n]) continue;
            state[n] = 1;
            age += 1;
            const unsigned v = val[n];
            if (v < vmax) push_small(items, Item{((u64)v << 32) | (uint32_t)age, n, 0});      // (v == vmax with age != 0 is large)
            items += 1;
        }
    }
    if (phase_ms) phase_ms[1] = now_ms() - t_start - phase_ms[0];
    *n_ranked_out = n_ranked;
    free(occ); free(expl); free(state);
    return oom ? -1 : popped;
}

// ---- DENSE form (round 4): every seed in place, for a tie value at or above the value most seeds share (on a
// detect_anvils field: the 420 M background seeds of a 16 x 5424^2 window at exactly 0 -- every one of them is SMALL then,
// and has to pop before the tied markers do).  Items are the 8-byte entries above, in the array the device export filled
// (seed k at position k: the reference's heap before its first sift-up), with room for the pushes behind them.
//   * The heap is built IN PLACE by the reference's sift-up, arrival by arrival; an arrival that equals its parent -- a
//     background seed under a background seed -- is one comparison.
//   * A popped seed whose replacement (the last item) is a seed of the same value stays at the root: runs of such items at
//     the end of the array pop in one backward scan (the reference does the same, one sift-down of zero steps each).
//   * SEEDS OF THE ROOT'S VALUE FORM A TREE at the top of the heap (heap order; they compare equal, so a sift-down never
//     swaps two of them): a larger item taken from the end sinks along the path "left child if it is such a seed, else the
//     right one if it is" to a leaf of that tree, and every seed on the path moves up one node.  Seeds nobody floods from
//     are one and the same entry, so that move changes nothing for them: the path is kept between pops as a stack
//     (the next path is the same down to the parent of the leaf that just left the tree), only the few seeds with an id on
//     it are moved, and the sinking item continues by the general rule from the leaf on.  A pop costs O(1) amortised
//     instead of the 28 swaps of a 2^28-item heap.
// Same pushes, pops and sift decisions as the plain form, item for item (tools/replay_check.cpp, test_gpu_reference_order).
// The seed entries in 2 bits + exceptions (k_ws_seed_codes): most seeds are either the MODAL ballast entry `D` (the
// background's value, nobody floods from it) or LARGE; bit k of bits_d / bits_l says so for seed k, every other seed's entry
// follows in `exc`, in seed order.  3.6 GB of entries per 16 x 5424^2 window become 2 x 56 MB + ~30 MB over PCIe.
struct WsSeedCodes { const uint32_t *bits_d, *bits_l; wsr_u64 d; const wsr_u64 *exc; };
static inline wsr_u64 wsr_next_entry(const WsSeedCodes &c, int64_t k, int64_t &j) {
    const uint32_t m = 1u << (k & 31);
    if (c.bits_d[k >> 5] & m) return c.d;
    if (c.bits_l[k >> 5] & m) return WSR_LARGE;
    return c.exc[j++];
}
static void wsr_expand(int64_t M, const WsSeedCodes &c, wsr_u64 *h) {
    int64_t j = 0;
    for (int64_t k = 0; k < M; k++) h[k] = wsr_next_entry(c, k, j);
}

// `codes` != nullptr: h[0 .. M) is written here, from the codes, as the build goes (no pass of its own)
static int64_t ws_reference_ranks_dense(int64_t M, wsr_u64 *h, int64_t nQ, const unsigned *val, const int *nbr, int n_nbr,
                                        unsigned vmax, int *rank, int *n_ranked_out, double *phase_ms = nullptr,
                                        const WsSeedCodes *codes = nullptr)
{
    typedef wsr_u64 u64;
    uint8_t *state = (uint8_t *)calloc((size_t)(nQ > 0 ? nQ : 1), 1);                          // 1: already pushed
    int *pushed_id = (int *)malloc((size_t)(nQ + 2) * sizeof(int));                           // id of the pixel pushed with age a
    if (!state || !pushed_id) { free(state); free(pushed_id); return -1; }
    auto now_ms = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; };
    const double t_start = now_ms();
    // build: _watershed.pyx:120-152 for arrival k = 1 .. M - 1 (position k holds seed k already)
    int64_t j_exc = 0;
    if (codes && M > 0) h[0] = wsr_next_entry(*codes, 0, j_exc);
    for (int64_t k = 1; k < M; k++) {
        u64 e;
        if (codes) { e = wsr_next_entry(*codes, k, j_exc); h[k] = e; } else e = h[k];
        int64_t parent = (k - 1) >> 1;
        const u64 pe = h[parent];
        if (e == pe || !wsr_smaller(e, pe)) continue;
        int64_t child = k;
        h[child] = pe; child = parent;
        while (child > 0) {
            parent = (child - 1) >> 1;
            const u64 q = h[parent];
            if (!wsr_smaller(e, q)) break;
            h[child] = q; child = parent;
        }
        h[child] = e;
    }
    const double t_built = now_ms();
    for (int64_t i = 0; i < nQ; i++) rank[i] = -1;
    int64_t items = M, age = 1, popped = 0;
    int n_ranked = 0;
    // the path through the tree of seeds of the root's value (class = entry >> 31: value + seed flag)
    std::vector<int64_t> path;       // path[0] = 0
    std::vector<int> rel;            // levels d >= 1 of path entries that carry an id (not the ballast entry), ascending
    u64 path_class = 0;
    bool path_valid = false;
    auto push = [&](u64 e) {         // :120-152
        int64_t child = items;
        items += 1;
        while (child > 0) {
            const int64_t parent = (child - 1) >> 1;
            const u64 q = h[parent];
            if (!wsr_smaller(e, q)) break;
            if ((q >> 31) == path_class) path_valid = false;          // a seed of the tracked tree moves down: drop the saved path
            h[child] = q; child = parent;
        }
        h[child] = e;
    };
    // the reference's sift-down of item x from the hole at position i (:67-111, swap form <=> hole form)
    auto sift_down_from = [&](int64_t i, u64 x) {
        for (;;) {
            const int64_t l = 2 * i + 1, r = l + 1;
            if (l >= items) break;
            int64_t smallest = i;
            u64 cur = x;
            const u64 le = h[l];
            if (wsr_smaller(le, cur)) { smallest = l; cur = le; }
            if (r < items) { const u64 re = h[r]; if (wsr_smaller(re, cur)) { smallest = r; cur = re; } }
            if (smallest == i) break;
            h[i] = cur;
            i = smallest;
        }
        h[i] = x;
    };
    while (items > 0) {
        u64 e = h[0];
        if (e == WSR_LARGE) break;
        const bool seed_root = ((unsigned)e >> 31) != 0;
        const u64 cls = e >> 31;                                       // meaningful for a seed root
        if (seed_root) {
            // runs of ballast seeds of this value at the end of the array: each pops and hands the root to the next
            const u64 ballast = (cls << 31);                           // id + 1 == 0
            if (e == ballast && items > 1) {
                // the common case in blocks of eight: entries that ARE the ballast entry (the loop below takes the rest)
                int64_t n = items;
                while (n - 8 >= 1) {
                    const u64 *q = h + n - 8;
                    const u64 diff = (q[0] ^ ballast) | (q[1] ^ ballast) | (q[2] ^ ballast) | (q[3] ^ ballast) |
                                     (q[4] ^ ballast) | (q[5] ^ ballast) | (q[6] ^ ballast) | (q[7] ^ ballast);
                    if (diff) break;
                    n -= 8;
                }
                popped += items - n; items = n;                          // (e stays the ballast entry)
            }
            while (e == ballast && items > 1 && (h[items - 1] >> 31) == cls) { items -= 1; popped += 1; e = h[items]; }
            h[0] = e;
        }
        items -= 1;                                                    // e pops (:67-111)
        if (items > 0) {
            const u64 x = h[items];
            if (seed_root && (x >> 31) == cls) h[0] = x;               // equal to both children at most: stays at the root
            else if (seed_root) {
                // x is larger than every seed of the root's value: it sinks along their tree
                if (!path_valid || path_class != cls || path.back() >= items) { path.clear(); path.push_back(0); rel.clear(); path_class = cls; path_valid = true; }
                for (;;) {                                             // extend the saved path to a leaf of the tree
                    const int64_t i = path.back(), l = 2 * i + 1, r = l + 1;
                    int64_t c = -1;
                    if (l < items && (h[l] >> 31) == cls) c = l;
                    else if (r < items && (h[r] >> 31) == cls) c = r;
                    if (c < 0) break;
                    path.push_back(c);
                    if (((unsigned)h[c] & 0x7fffffffu) != 0) rel.push_back((int)path.size() - 1);
                }
                const int L = (int)path.size() - 1;
                if (L >= 1) {
                    h[0] = cls << 31;                                  // every seed on the path moves up one node
                    size_t w = 0;
                    for (size_t q = 0; q < rel.size(); q++) {
                        const int d = rel[q];
                        h[path[d - 1]] = h[path[d]];
                        h[path[d]] = cls << 31;
                        if (d - 1 >= 1) rel[w++] = d - 1;
                    }
                    rel.resize(w);
                }
                sift_down_from(path[L], x);                            // from the leaf on: the general rule
                if (L >= 1) path.pop_back(); else path_valid = false;
            } else { path_valid = false; sift_down_from(0, x); }
        }
        popped++;
        int id;
        const unsigned lo = (unsigned)e;
        if (lo >> 31) { id = (int)(lo & 0x7fffffffu) - 1; if (id >= 0) rank[id] = n_ranked++; }      // a marker (age 0): its pop rank
        else id = pushed_id[lo];
        if (id < 0) continue;
        const int *np = nbr + (int64_t)id * n_nbr;
        for (int k = 0; k < n_nbr; k++) {                              // :308-341
            const int n = np[k];
            if (n < 0 || state[n]) continue;
            state[n] = 1;
            age += 1;
            const unsigned v = val[n];
            if (v < vmax) { pushed_id[age] = n; push(((u64)v << 32) | (u64)(unsigned)age); }         // (v == vmax with age != 0 is large)
            else { h[items] = WSR_LARGE; items += 1; }
        }
    }
    if (phase_ms) { phase_ms[0] = t_built - t_start; phase_ms[1] = now_ms() - t_built; }
    *n_ranked_out = n_ranked;
    free(state); free(pushed_id);
    return popped;
}
