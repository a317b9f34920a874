// coherence_sim.cpp — CPU model of the persistent-wave shadow march (k_shadow1) for studying wave coherence.
//
// Reads the shadow segments of one tile (tools/sim/dump_shadow_segments.py), orders them like the GPU job list
// ([depth][NEE sample][slot]) or by an alternative key, and replays the wave scheduling: 64 lanes per wave, every lane owns one
// segment, a finished lane takes the next segment of its wave's 256-entry chunk.  For every trip it records, per fold
// iteration, whether ANY lane of the wave enters the sphere-fold block (what the hardware pays) and how many lanes needed it.
// Statistics only: arithmetic is plain float, not the bit-exact device sequence.
//
//   g++ -O2 -o /tmp/coherence_sim tools/sim/coherence_sim.cpp && /tmp/coherence_sim /tmp/seg.bin [order]
//   order: 0 GPU list order, 1 Morton(start) within (depth), 2 (light = end point cluster, Morton(start)), 3 first-eval fold mask,
//          4 random shuffle within depth (worst case)
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

struct Seg { float depth, s, a[3], b[3]; };

static const int ITER = 12;
static const float L = 1.0f, MRS = 0.01f * 0.01f, FRS = 1.9f * 1.9f, SCALE = -2.1f;

// MandelBox::dist + per-iteration fold mask (bit i = lane folds in iteration i)
static float dist_mask(const float p0[3], uint32_t* mask) {
    float p[3] = {p0[0], p0[1], p0[2]}, dr = 1.0f;
    uint32_t m = 0;
    for (int i = 0; i < ITER; i++) {
        for (int c = 0; c < 3; c++) { float cl = std::min(std::max(p[c], -L), L); p[c] = cl * 2.0f - p[c]; }
        float r2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
        if (r2 < FRS) { m |= 1u << i; float q = FRS / std::max(r2, MRS); for (int c = 0; c < 3; c++) p[c] *= q; dr *= q; }
        for (int c = 0; c < 3; c++) p[c] = p[c] * SCALE + p0[c];
        dr = -dr * SCALE + 1.0f;
    }
    *mask = m;
    return std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2]) / std::fabs(dr);
}

static uint32_t part1by2(uint32_t x) { x &= 0x3FF; x = (x | (x << 16)) & 0x30000FF; x = (x | (x << 8)) & 0x300F00F; x = (x | (x << 4)) & 0x30C30C3; x = (x | (x << 2)) & 0x9249249; return x; }
static uint32_t morton(const float a[3]) {
    uint32_t q[3];
    for (int c = 0; c < 3; c++) { float v = (a[c] + 6.0f) / 12.0f; v = std::min(std::max(v, 0.0f), 0.999999f); q[c] = (uint32_t)(v * 1024.0f); }
    return part1by2(q[0]) | (part1by2(q[1]) << 1) | (part1by2(q[2]) << 2);
}

struct Lane { bool has = false, first = false; float st[3], dir[3], maxd, t; int m; uint32_t last = 0xFFF; };

int main(int argc, char** argv) {
    if (argc < 2) return 1;
    const int order = argc > 2 ? atoi(argv[2]) : 0;
    const int regroup = argc > 3 ? atoi(argv[3]) : 0; // 1: ideal regrouping bound (lanes of ALL waves re-sorted by fold count every trip)
    const int period = argc > 4 ? atoi(argv[4]) : 1;  // 2/3: the 256 lanes of a block (4 waves) are re-sorted every `period` trips by the PREVIOUS evaluation's fold mask (2: popcount, 3: highest folding iteration)
    long step = 0;
    FILE* f = fopen(argv[1], "rb");
    fseek(f, 0, SEEK_END); size_t n = ftell(f) / sizeof(Seg); fseek(f, 0, SEEK_SET);
    std::vector<Seg> segs(n);
    if (fread(segs.data(), sizeof(Seg), n, f) != n) return 2;
    fclose(f);
    // drop NaN / padding lanes (the GPU never enqueues them)
    segs.erase(std::remove_if(segs.begin(), segs.end(), [](const Seg& s) { for (int c = 0; c < 3; c++) if (!(s.a[c] == s.a[c]) || !(s.b[c] == s.b[c])) return true; return false; }), segs.end());
    n = segs.size();
    std::vector<uint32_t> idx(n);
    for (size_t i = 0; i < n; i++) idx[i] = (uint32_t)i;
    std::vector<uint64_t> key(n);
    std::mt19937 rng(1);
    for (size_t i = 0; i < n; i++) {
        const Seg& s = segs[i];
        uint64_t k = (uint64_t)s.depth << 56;
        if (order == 0) k |= (uint64_t)s.s << 48; // [depth][sample][slot]; slot order = dump order
        else if (order == 1) k |= (uint64_t)morton(s.a) << 8;
        else if (order == 2) k |= ((uint64_t)(morton(s.b) >> 21) << 40) | ((uint64_t)morton(s.a) << 8);
        else if (order == 3) { uint32_t m; dist_mask(s.a, &m); k |= (uint64_t)m << 32 | morton(s.a); }
        else if (order == 4) k |= (uint64_t)(rng() & 0xFFFFFF) << 8;
        key[i] = k;
    }
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return key[x] < key[y]; });

    // replay per depth (one kernel launch per depth)
    const int W = 64, CH = 256; // waves in flight, chunk size
    const float c0 = 0.0001f * 0.5f, c1 = 0.00001f * 0.5f;
    double trips = 0, lane_evals = 0, wave_entries = 0, lane_folds = 0, jobs = 0, ideal_entries = 0;
    double hist[65] = {0}; // block entries by the number of lanes that fold in that (wave, iteration)
    size_t lo = 0;
    while (lo < n) {
        size_t hi = lo;
        while (hi < n && segs[idx[hi]].depth == segs[idx[lo]].depth) hi++;
        size_t head = lo;
        std::vector<Lane> lanes(W * 64);
        std::vector<size_t> cur(W, 0), end(W, 0);
        bool any = true;
        while (any) {
            any = false;
            step++;
            if (regroup >= 2 && step % period == 0)
                for (int b = 0; b < W / 4; b++) {
                    Lane* bl = &lanes[b * 256];
                    auto keyf = [&](const Lane& x) { if (!x.has) return 99; return regroup == 2 ? __builtin_popcount(x.last) : (x.last ? 32 - __builtin_clz(x.last) : 0); };
                    std::stable_sort(bl, bl + 256, [&](const Lane& x, const Lane& y) { return keyf(x) < keyf(y); });
                }
            std::vector<uint32_t> all_masks;
            for (int w = 0; w < W; w++) {
                Lane* ln = &lanes[w * 64];
                for (int l = 0; l < 64; l++) { // refill
                    if (ln[l].has) continue;
                    if (cur[w] == end[w]) { if (head >= hi) continue; cur[w] = head; end[w] = std::min(head + CH, hi); head = end[w]; }
                    const Seg& s = segs[idx[cur[w]++]];
                    Lane& L2 = ln[l];
                    float d[3], len = 0;
                    for (int c = 0; c < 3; c++) { d[c] = s.b[c] - s.a[c]; len += d[c] * d[c]; }
                    len = std::sqrt(len);
                    for (int c = 0; c < 3; c++) { L2.st[c] = s.a[c]; L2.dir[c] = d[c] / len; }
                    L2.maxd = len; L2.has = true; L2.first = true; L2.m = 0; L2.t = 0; L2.last = 0xFFF;
                    jobs++;
                }
                uint32_t wave_or = 0; int act = 0; int per_it[ITER] = {0};
                for (int l = 0; l < 64; l++) {
                    Lane& L2 = ln[l];
                    if (!L2.has) continue;
                    act++;
                    float p[3];
                    for (int c = 0; c < 3; c++) p[c] = L2.first ? L2.st[c] : L2.st[c] + L2.dir[c] * L2.t;
                    uint32_t m; float d = dist_mask(p, &m);
                    wave_or |= m; lane_folds += __builtin_popcount(m); L2.last = m;
                    for (int i = 0; i < ITER; i++) per_it[i] += (m >> i) & 1;
                    all_masks.push_back(m);
                    bool done = false;
                    if (L2.first) { L2.t = d; L2.first = false; if (L2.t > L2.maxd || d != d) done = true; }
                    else if (std::fabs(d) < std::max(c0, c1 * L2.t)) done = true;
                    else { L2.t += d; L2.m++; if (L2.m == 100 || L2.t > L2.maxd) done = true; }
                    if (done) L2.has = false;
                }
                if (act) { trips++; lane_evals += act; wave_entries += __builtin_popcount(wave_or); any = true; for (int i = 0; i < ITER; i++) if (per_it[i]) hist[per_it[i]]++; }
            }
            if (regroup && !all_masks.empty()) { // ideal: sort all live lanes by fold count, regroup into waves of 64
                std::sort(all_masks.begin(), all_masks.end(), [](uint32_t a, uint32_t b) { int pa = __builtin_popcount(a), pb = __builtin_popcount(b); return pa != pb ? pa < pb : a < b; });
                for (size_t i = 0; i < all_masks.size(); i += 64) { uint32_t o = 0; for (size_t j = i; j < std::min(i + 64, all_masks.size()); j++) o |= all_masks[j]; ideal_entries += __builtin_popcount(o); }
            }
        }
        lo = hi;
    }
    printf("order %d: jobs %.0f  trips %.0f  lane util %.3f  evals/job %.1f  fold-block entries per trip (wave) %.2f / %d   folds per lane-eval %.2f", order, jobs, trips,
           lane_evals / (trips * 64), lane_evals / jobs, wave_entries / trips, ITER, lane_folds / lane_evals);
    if (regroup) printf("   ideal regroup %.2f", ideal_entries / trips);
    printf("\n");
    double cum = 0;
    printf("block entries by folding-lane count (cumulative share): ");
    for (int c = 1; c <= 64; c++) { cum += hist[c]; if (c == 1 || c == 2 || c == 4 || c == 8 || c == 12 || c == 15 || c == 16 || c == 24 || c == 32 || c == 48 || c == 64) printf("<=%d: %.3f  ", c, cum / wave_entries); }
    printf("\n");
    return 0;
}
