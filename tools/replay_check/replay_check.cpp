// CPU check + timing of the three host replays of csrc/ws_replay.h (no GPU needed).
//   replay_check random N      N random small instances: plain == sparse == dense (ranks, pops)
//   replay_check big M         one config-F-like instance with M seeds: timings of the dense (and plain) form
#include "../../tobac_flow_amd/csrc/ws_replay.h"
#include <stdio.h>
#include <time.h>
#include <random>
#include <algorithm>

static double now_ms() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }

struct Inst {
    int64_t M, nQ; int n_nbr; unsigned vmax;
    std::vector<unsigned> seed_val; std::vector<int> seed_id;     // per seed
    std::vector<unsigned> val; std::vector<int> nbr;              // per sub-graph id
};

static bool run_all(const Inst &I, bool with_plain, bool verbose)
{
    const int64_t M = I.M, nQ = I.nQ;
    std::vector<int> r_plain(nQ), r_sparse(nQ), r_dense(nQ);
    int n_plain = -1, n_sparse = -1, n_dense = -1;
    int64_t p_plain = -2, p_sparse = -2, p_dense = -2;
    std::vector<wsr_u64> entries(M + nQ + 1);
    for (int64_t k = 0; k < M; k++) entries[k] = wsr_seed_entry(I.seed_val[k], I.seed_id[k], I.vmax);
    double t0 = now_ms();
    if (with_plain) p_plain = ws_reference_ranks_plain(M, entries.data(), I.seed_val.data(), nQ, I.val.data(), I.nbr.data(), I.n_nbr, I.vmax, r_plain.data(), &n_plain);
    double t1 = now_ms();
    std::vector<long long> sk; std::vector<unsigned> sv; std::vector<int> sid;
    for (int64_t k = 0; k < M; k++) if (I.seed_val[k] <= I.vmax) { sk.push_back(k); sv.push_back(I.seed_val[k]); sid.push_back(I.seed_id[k]); }
    double t2 = now_ms();
    double phs[2] = {0, 0};
    p_sparse = ws_reference_ranks_sparse(M, (int64_t)sk.size(), sk.data(), sv.data(), sid.data(), nQ, I.val.data(), I.nbr.data(), I.n_nbr, I.vmax, r_sparse.data(), &n_sparse, phs);
    double t3 = now_ms();
    if (verbose) printf("sparse: build %.1f ms, pops %.1f ms\n", phs[0], phs[1]);
    // the 2-bit codes + exceptions the device sends: modal ballast entry = the most frequent entry with id -1
    std::vector<uint32_t> bd((M + 31) / 32 + 1, 0), bl((M + 31) / 32 + 1, 0);
    std::vector<wsr_u64> exc;
    wsr_u64 D = 0;
    {
        std::vector<wsr_u64> sample;
        for (int64_t k = 0; k < M; k += std::max<int64_t>(1, M / 256)) if (entries[k] != WSR_LARGE && ((unsigned)entries[k] & 0x7fffffffu) == 0) sample.push_back(entries[k]);
        std::sort(sample.begin(), sample.end());
        size_t best = 0;
        for (size_t i = 0; i < sample.size();) { size_t e2 = i; while (e2 < sample.size() && sample[e2] == sample[i]) e2++; if (e2 - i > best) { best = e2 - i; D = sample[i]; } i = e2; }
        for (int64_t k = 0; k < M; k++) {
            if (D && entries[k] == D) bd[k >> 5] |= 1u << (k & 31);
            else if (entries[k] == WSR_LARGE) bl[k >> 5] |= 1u << (k & 31);
            else exc.push_back(entries[k]);
        }
    }
    WsSeedCodes codes{bd.data(), bl.data(), D, exc.data()};
    std::vector<wsr_u64> h2(M + nQ + 1);
    double ph[2] = {0, 0};
    p_dense = ws_reference_ranks_dense(M, h2.data(), nQ, I.val.data(), I.nbr.data(), I.n_nbr, I.vmax, r_dense.data(), &n_dense, ph, &codes);
    double t4 = now_ms();
    {   // and from the expanded array: the same
        std::vector<int> r2(nQ); int n2 = -1;
        const int64_t p2 = ws_reference_ranks_dense(M, entries.data(), nQ, I.val.data(), I.nbr.data(), I.n_nbr, I.vmax, r2.data(), &n2);
        if (p2 != p_dense || n2 != n_dense || r2 != r_dense) { printf("dense from codes != dense from entries\n"); return false; }
    }
    if (verbose) printf("dense: build %.1f ms, pops %.1f ms\n", ph[0], ph[1]);
    if (verbose) printf("M %lld nQ %lld small seeds %zu: plain %.1f ms (%lld pops), sparse %.1f ms (%lld pops), dense %.1f ms (%lld pops, %d ranked)\n",
                        (long long)M, (long long)nQ, sk.size(), t1 - t0, (long long)p_plain, t3 - t2, (long long)p_sparse, t4 - t3, (long long)p_dense, n_dense);
    bool ok = p_sparse == p_dense && n_sparse == n_dense && r_sparse == r_dense;
    if (with_plain) ok = ok && p_plain == p_dense && n_plain == n_dense && r_plain == r_dense;
    return ok;
}

static Inst random_instance(std::mt19937_64 &g)
{
    Inst I;
    auto U = [&](int a, int b) { return (int)(a + g() % (uint64_t)(b - a + 1)); };
    I.n_nbr = U(1, 6);
    const int n_levels = U(1, 5);                                  // few distinct values: many ties
    std::vector<unsigned> levels(n_levels);
    for (auto &v : levels) v = 0x80000000u + (unsigned)U(0, 40) * 1000u;
    std::sort(levels.begin(), levels.end());
    I.M = U(1, 300);
    const int n_flood = U(0, 200);
    const int n_rel = U(0, (int)std::min<int64_t>(I.M, 60));       // seeds somebody floods from
    I.nQ = n_flood + n_rel;
    I.val.resize(I.nQ); I.nbr.assign((size_t)I.nQ * I.n_nbr, -1);
    const int modal = U(0, n_levels - 1);
    auto pick = [&]() { return (g() % 100 < 60) ? levels[modal] : levels[U(0, n_levels - 1)]; };
    I.seed_val.resize(I.M); I.seed_id.assign(I.M, -1);
    for (int64_t k = 0; k < I.M; k++) I.seed_val[k] = pick();
    // ids [0, n_flood) floodable, [n_flood, nQ) relevant seeds
    std::vector<int64_t> ks(I.M);
    for (int64_t k = 0; k < I.M; k++) ks[k] = k;
    std::shuffle(ks.begin(), ks.end(), g);
    for (int j = 0; j < n_rel; j++) { I.seed_id[ks[j]] = n_flood + j; I.val[n_flood + j] = I.seed_val[ks[j]]; }
    for (int j = 0; j < n_flood; j++) I.val[j] = (g() % 100 < 50) ? levels[modal] : levels[U(0, n_levels - 1)];
    if (n_flood > 0)
        for (int64_t q = 0; q < I.nQ; q++)
            for (int j = 0; j < I.n_nbr; j++) if (g() % 100 < 70) I.nbr[q * I.n_nbr + j] = U(0, n_flood - 1);
    I.vmax = levels[U(0, n_levels - 1)];
    return I;
}

int main(int argc, char **argv)
{
    if (argc >= 3 && !strcmp(argv[1], "random")) {
        const int N = atoi(argv[2]);
        std::mt19937_64 g(argc >= 4 ? atoll(argv[3]) : 12345);
        for (int i = 0; i < N; i++) {
            Inst I = random_instance(g);
            if (!run_all(I, true, false)) { printf("MISMATCH at instance %d (M %lld nQ %lld n_nbr %d)\n", i, (long long)I.M, (long long)I.nQ, I.n_nbr); run_all(I, true, true); return 1; }
        }
        printf("%d random instances: plain == sparse == dense\n", N);
        return 0;
    }
    if (argc >= 3 && !strcmp(argv[1], "big")) {
        // a 2-D scene with the statistics of a detect_anvils window: background seeds at value 0 (most of the image), cores
        // (seeds at -1, interior = ballast, rim relevant), a floodable ring around each core with values > 0, ring / background
        // boundary seeds with values > 0 (LARGE), a few seeds at small positive values (the tie value)
        const int64_t W = atoll(argv[2]), H = W;
        const bool with_plain = argc >= 4 && atoi(argv[3]) != 0;
        std::mt19937_64 g(7);
        std::vector<uint8_t> cls((size_t)W * H, 0);                 // 0 background, 1 core marker, 2 floodable ring
        std::vector<float> fv((size_t)W * H, 0.0f);
        const int n_blobs = (int)(W * H / 65536);
        for (int b = 0; b < n_blobs; b++) {
            const int cx = (int)(g() % W), cy = (int)(g() % H), rc = 4 + (int)(g() % 24), rr = rc + 3 + (int)(g() % 12);
            for (int y = std::max(0, cy - rr); y < std::min<int64_t>(H, cy + rr + 1); y++)
                for (int x = std::max(0, cx - rr); x < std::min<int64_t>(W, cx + rr + 1); x++) {
                    const int d2 = (x - cx) * (x - cx) + (y - cy) * (y - cy);
                    const size_t p = (size_t)y * W + x;
                    if (d2 <= rc * rc) { cls[p] = 1; fv[p] = -1.0f; }
                    else if (d2 <= rr * rr && cls[p] != 1) { cls[p] = 2; fv[p] = 0.2f + 0.001f * (float)(g() % 1000); }
                }
        }
        auto key = [](float v) { unsigned u; memcpy(&u, &v, 4); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); };
        // background pixels next to a ring are "large" seeds (sobel > 0 there), some core pixels carry the tie value
        const int dx[4] = {0, -1, 1, 0}, dy[4] = {-1, 0, 0, 1};
        Inst I; I.n_nbr = 4;
        std::vector<int> id((size_t)W * H, -1);
        int64_t nQ = 0;
        for (int64_t y = 0; y < H; y++) for (int64_t x = 0; x < W; x++) {
            const size_t p = (size_t)y * W + x;
            bool near_ring = false;
            for (int j = 0; j < 4; j++) { const int64_t xx = x + dx[j], yy = y + dy[j]; if (xx >= 0 && yy >= 0 && xx < W && yy < H && cls[(size_t)yy * W + xx] == 2) near_ring = true; }
            if (cls[p] == 0 && near_ring) fv[p] = 1.0f + 0.001f * (float)(g() % 1000);
            if (cls[p] == 1 && near_ring && g() % 50 == 0) fv[p] = 0.00390625f;     // the tie value, on a few relevant core seeds
            if (cls[p] == 2 || near_ring) id[p] = (int)nQ++;
        }
        I.nQ = nQ; I.val.resize(nQ); I.nbr.assign((size_t)nQ * 4, -1);
        for (int64_t y = 0; y < H; y++) for (int64_t x = 0; x < W; x++) {
            const size_t p = (size_t)y * W + x;
            if (cls[p] != 2) { I.seed_val.push_back(key(fv[p])); I.seed_id.push_back(id[p]); }
            if (id[p] >= 0) {
                I.val[id[p]] = key(fv[p]);
                for (int j = 0; j < 4; j++) { const int64_t xx = x + dx[j], yy = y + dy[j]; if (xx >= 0 && yy >= 0 && xx < W && yy < H && cls[(size_t)yy * W + xx] == 2) I.nbr[(size_t)id[p] * 4 + j] = id[(size_t)yy * W + xx]; }
            }
        }
        I.M = (int64_t)I.seed_val.size();
        I.vmax = (argc >= 5 && atoi(argv[4]) != 0) ? key(-1.0f) : key(0.00390625f);     // 4th argument != 0: the usual case, ties among the cores only
        printf("big: %lld x %lld, M %lld seeds, nQ %lld\n", (long long)W, (long long)H, (long long)I.M, (long long)nQ);
        const bool ok = run_all(I, with_plain, true);
        printf(ok ? "agree\n" : "MISMATCH\n");
        return ok ? 0 : 1;
    }
    printf("usage: replay_check random N [seed] | big WIDTH [with_plain [cores_only]]\n");
    return 2;
}
