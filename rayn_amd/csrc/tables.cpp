// tables.cpp — host-side builders of the three tables the GPU path consumes:
//   rayn_build_rd_tables  = Samples::new_rd                       (src/sampler.rs:18-37)
//   rayn_build_scramble   = SmallRng::seed_from_u64(pixel).gen()  (src/film.rs:460-461)
//   rayn_build_fis_table  = FilterImportanceSampler::new + CDF     (src/filter.rs:187-220, src/math.rs:136-191)
// The arithmetic of quasi-rd (git ce117035) and rand 0.7.2 / rand_pcg 0.2.1 is not vendored in the
// reference; it is restated from the published algorithms (Roberts' R_d sequence; PCG XSL-RR 128/64 MCG).
// A rayn host that already owns these tables passes its own buffers to rayn_hip_render_frame instead.
#include <stdint.h>

#include <vector>

#include "../../include/rayn_detmath.h"
#include "../../include/rayn_hip.h"

namespace {

typedef unsigned __int128 u128;

inline u128 make128(uint64_t hi, uint64_t lo) { return ((u128)hi << 64) | (u128)lo; }

// alpha_j = phi_d^-j as 0.128 fixed point (tools/gen_rd_constants.py)
const uint64_t ALPHA_1D[2] = {0x9e3779b97f4a7c15ULL, 0xf39cc0605cedc834ULL};
const uint64_t ALPHA_2D_X[2] = {0xc13fa9a902a6328fULL, 0x434ff71b2d97724bULL};
const uint64_t ALPHA_2D_Y[2] = {0x91e10da5c79e7b1cULL, 0xd438a0a8e6c9c0fcULL};

// frac(1/2 + alpha * index), top 24 bits as f32 in [0,1)
inline float rd_point(const uint64_t alpha[2], u128 index) {
    u128 x = ((u128)1 << 127) + make128(alpha[0], alpha[1]) * index; // mod 2^128 == frac()
    return (float)(uint32_t)(x >> 104) * (1.0f / 16777216.0f);
}

// quasi_rd::Sequence::new_with_offset(dim, offset).fill_with_samples_f32(out)
void rd_fill(float* out, uint32_t count, uint32_t dim, uint64_t offset) {
    for (uint32_t k = 0; k < count; k++) {
        u128 index = (u128)offset + 1u + k;
        if (dim == 1) out[k] = rd_point(ALPHA_1D, index);
        else {
            out[2 * k + 0] = rd_point(ALPHA_2D_X, index);
            out[2 * k + 1] = rd_point(ALPHA_2D_Y, index);
        }
    }
}

inline uint32_t rotr32(uint32_t v, uint32_t r) { r &= 31; return r ? (v >> r) | (v << (32 - r)) : v; }
inline uint64_t rotr64(uint64_t v, uint32_t r) { r &= 63; return r ? (v >> r) | (v << (64 - r)) : v; }

// rand_core 0.5 SeedableRng::seed_from_u64 -> Pcg64Mcg::from_seed -> next_u32 -> Standard f32
float small_rng_first_f32(uint64_t seed) {
    uint32_t w[4];
    uint64_t st = seed;
    for (int i = 0; i < 4; i++) { // PCG32 expansion of the u64 seed into 16 seed bytes
        st = st * 6364136223846793005ULL + 11634580027462260723ULL;
        w[i] = rotr32((uint32_t)(((st >> 18) ^ st) >> 27), (uint32_t)(st >> 59));
    }
    u128 state = make128(((uint64_t)w[3] << 32) | w[2], ((uint64_t)w[1] << 32) | w[0]) | 1u; // Mcg128Xsl64::new
    state *= make128(0x2360ED051FC65DA4ULL, 0x4385DF649FCCF645ULL);
    uint64_t out = rotr64((uint64_t)(state >> 64) ^ (uint64_t)state, (uint32_t)(state >> 122));
    return (float)((uint32_t)out >> 8) * (1.0f / 16777216.0f);
}

// Lanczos' sinc, src/filter.rs:159-168
float lanczos_sinc(float x) {
    x = x < 0.0f ? -x : x;
    if (x <= 0.00001f) return 1.0f;
    const float pix = 3.14159265358979323846f * x;
    return dm_sinf(pix) / pix;
}

// Filter::evaluate of the four filters.  p0/p1: Mitchell-Netravali (b, c); LanczosSinc (tau, -).
float eval_filter(uint32_t kind, float radius, float p0, float p1, float p) {
    float ap = p < 0.0f ? -p : p;
    if (kind == RAYN_FILTER_BOX) return ap > radius ? 0.0f : 1.0f; // BoxFilter, src/filter.rs:131-139
    if (kind == RAYN_FILTER_MITCHELL) { // MitchellNetravaliFilter::evaluate, src/filter.rs:74-91
        const float b = p0, c = p1;
        float x = 2.0f * p / radius;
        x = x < 0.0f ? -x : x;
        if (x >= 2.0f) return 0.0f;
        if (x > 1.0f)
            return ((-b - 6.0f * c) * x * x * x + (6.0f * b + 30.0f * c) * x * x + (-12.0f * b - 48.0f * c) * x + (8.0f * b + 24.0f * c)) * (1.0f / 6.0f);
        return ((12.0f - 9.0f * b - 6.0f * c) * x * x * x + (-18.0f + 12.0f * b + 6.0f * c) * x * x + (6.0f - 2.0f * b)) * (1.0f / 6.0f);
    }
    if (kind == RAYN_FILTER_LANCZOS) { // LanczosSincFilter::evaluate, src/filter.rs:176-184
        if (ap > radius) return 0.0f;
        const float lanczos = lanczos_sinc(ap / p0);
        return lanczos_sinc(ap) * lanczos;
    }
    // BlackmanHarrisFilter, src/filter.rs:29-49
    const float pi = 3.14159265358979323846f;
    if (ap > radius) return 0.0f;
    float q = p / radius;
    float x = (q < 0.0f ? -q : q) * 0.5f + 0.5f;
    return 0.35875f - 0.48829f * dm_cosf((pi * 2.0f) * x) + 0.14128f * dm_cosf((pi * 4.0f) * x) + 0.01168f * dm_cosf((pi * 6.0f) * x);
}

} // namespace

extern "C" {

uint32_t rayn_sets_1d(uint32_t max_bounces, uint32_t volume_marches) { return 1u + (max_bounces + 1u) * (3u + volume_marches); }
uint32_t rayn_sets_2d(uint32_t max_bounces, uint32_t volume_marches) { return 2u + (max_bounces + 1u) * (12u + 8u * volume_marches); }

int rayn_build_rd_tables(uint32_t spp, uint32_t sets_1d, uint32_t sets_2d, uint64_t frame, float* samples_1d, float* samples_2d) {
    if (!samples_1d || !samples_2d || spp == 0) return RAYN_ERR_INVALID_ARG;
    for (uint32_t i = 0; i < sets_1d; i++) rd_fill(samples_1d + (size_t)spp * i, spp, 1, (frame + i) << 32);
    for (uint32_t i = 0; i < sets_2d; i++) rd_fill(samples_2d + (size_t)spp * 2 * i, spp, 2, (frame + sets_1d + i) << 32);
    return RAYN_OK;
}

int rayn_build_scramble(uint32_t width, uint32_t height, float* scramble) {
    if (!scramble) return RAYN_ERR_INVALID_ARG;
    for (uint32_t y = 0; y < height; y++)
        for (uint32_t x = 0; x < width; x++) scramble[x + (size_t)y * width] = small_rng_first_f32((uint64_t)(x + y * width));
    return RAYN_OK;
}

int rayn_build_fis_table(uint32_t filter_kind, float radius, float* table512) {
    if (filter_kind > RAYN_FILTER_BOX) return RAYN_ERR_INVALID_ARG; // the parameterised filters go through _ex
    return rayn_build_fis_table_ex(filter_kind, radius, 0.0f, 0.0f, table512);
}

int rayn_build_fis_table_ex(uint32_t filter_kind, float radius, float param0, float param1, float* table512) {
    if (!table512 || filter_kind > RAYN_FILTER_LANCZOS) return RAYN_ERR_INVALID_ARG;
    const uint32_t N = RAYN_FIS_TABLE_SIZE;
    std::vector<float> pos(N), w(N), cdf(N);
    float sum = 0.0f;
    for (uint32_t n = 0; n < N; n++) { // CDF::insert, src/math.rs:153-156
        float t = (float)n / (float)(N - 1);
        pos[n] = 0.0f * (1.0f - t) + radius * t; // 0.0.lerp(f_rad, t)
        w[n] = eval_filter(filter_kind, radius, param0, param1, pos[n]);
        sum += w[n];
    }
    float run = 0.0f; // CDF::prepare, src/math.rs:158-181
    for (uint32_t n = 0; n < N; n++) { w[n] /= sum; run += w[n]; cdf[n] = run; }
    for (uint32_t n = N; n-- > 0;) { cdf[n] = 1.0f; if (w[n] > 0.0f) break; }
    for (uint32_t n = 0; n < N; n++) { // CDF::sample: first density >= u, src/math.rs:183-190
        float u = (float)n / (float)(N - 1);
        float v = 0.0f;
        for (uint32_t k = 0; k < N; k++) if (cdf[k] >= u) { v = pos[k]; break; }
        table512[n] = v;
    }
    return RAYN_OK;
}

uint32_t rayn_tile_count(uint32_t width, uint32_t height, uint32_t tile_w, uint32_t tile_h) {
    if (!tile_w || !tile_h) return 0;
    return ((width + width % tile_w) / tile_w) * ((height + height % tile_h) / tile_h); // src/film.rs:399-404
}

} // extern "C"
