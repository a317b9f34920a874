// kernels.h — host-visible launch wrappers of kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_scene.h"

namespace rayn {

constexpr uint32_t SCAN_NC_BIN = 16; // = RAYN_MAX_HITABLES: classes of the bin scan
// k_resolve_blk's 32-bit sort key = depth:7 | offset of the termination slot inside the tile's binned segment of that depth.  The depth field needs 7 bits
// (max_bounces <= 120), which leaves 25 for the offset; a tile's segment holds at most max_tile_pixels * spp slots + the x4 bin padding and the x64 tail.
// resolve_keys_fit() is checked by the host for every frame share (run_worker): relaxing the host's tile / spp limits cannot silently spill the offset into the depth bits
// (which would change the summation order) - it becomes an error instead.
constexpr uint32_t RESOLVE_KEY_SHIFT = 25;
static_assert(120u < (1u << (32 - RESOLVE_KEY_SHIFT)) - 1u, /* MAX_BOUNCES below */ "depth field of the resolve key too narrow for max_bounces <= 120 (0x7F.. is NOKEY's prefix)");
constexpr bool resolve_keys_fit(uint32_t max_tile_pixels, uint32_t spp) {
    return (unsigned long long)max_tile_pixels * spp + SCAN_NC_BIN * 3ull + 64ull < (1ull << RESOLVE_KEY_SHIFT);
}
// the limits validate() enforces (rayn_hip.hip): tiles of at most 1024 pixels; k_resolve_blk serves 512 < spp <= 4096.  Under them the check cannot fail -
// it guards a future relaxation of either limit (then: a compile error here, or the per-frame error in run_worker for the combination that does not fit)
constexpr uint32_t MAX_TILE_PIXELS = 1024, MAX_SPP_RESOLVE_BLK = 4096, MAX_BOUNCES = 120;
static_assert(resolve_keys_fit(MAX_TILE_PIXELS, MAX_SPP_RESOLVE_BLK), "k_resolve_blk's 25-bit slot offset does not cover the largest tile x spp the host accepts");
static_assert(!resolve_keys_fit(MAX_TILE_PIXELS, 8u * MAX_SPP_RESOLVE_BLK), "resolve_keys_fit must reject a segment beyond 2^25 slots");


// Path pool (device pointers): one slot per camera path, records of 16 bytes so that a scattered
// access costs one 128-bit gather per record instead of four 32-bit ones (the shade kernels were
// bound by the address path of the vector memory unit, not by bandwidth).
struct Pool {
    float4* geo0;  // origin.xyz, dir.x
    float4* geo1;  // dir.y, dir.z, hit_t, bits: hit object (8) | sample index << 8
    float4* col0;  // radiance.rgb, throughput.r
    float4* col1;  // throughput.g, throughput.b, bits: film pixel index, ray time
    float4* aov;   // depth-0 WorldNormal sample xyz, bits: depth-0 object (OBJ_NONE = no Alpha/Normal sample)
    uint32_t* term_key; // binned slot at which the path emitted its Color/Background sample
    uint8_t* term_info; // depth (bits 0..6) | Background flag (bit 7); TERM_NONE = no sample emitted (dropped ray)
};

// per-slot NEE records + the shadow-job list (see k_shade_setup)
struct Nee {
    float* x;      // [4][3][cap] surface samples: unoccluded contribution (Le*f)*transmission
    float* vtr;    // [ns-4][cap] volume samples: transmission sample point -> light (x = Le * 1/(4 pi) * vtr is rebuilt by k_shade_finish)
    float* pdf;    // [ns][cap]
    float* aux;    // [ns-4][cap] volume samples: exp(-rho_t * sample distance)
    uint8_t* vis;  // [ns][cap]   HitableStore::test_occluded: 0 occluded, 1 visible, 2 SDF march pending.  The shadow kernels only write
                   //             VISIBLE results (1): ~90 % of the marched segments of the shipped scene are occluded, and a pending mark left
                   //             in place reads as occluded in k_shade_finish - a tenth of the scattered byte stores (r2: 151 GB written per
                   //             config-3 frame for 40 GB of results)
    unsigned long long* vpicks; // [cap] light index of every volume sample, 4 bits each (sample 4 + k at bits 4k..4k+3)
    float* T;      // [cap]       volume transmission of the segment
    float* t0;     // [cap]       lane-0 ray time of the packet (only written / read when a hitable is time-sequenced)
    float* nthr;   // [3][cap]    throughput of the spawned ray
    uint8_t* flags; // [cap]      bit0 alive, bit1 surface NEE, bit2 volume NEE
    size_t cap;
    uint32_t* job_ref; // [jobcap] dense list of pending [sample*cap + slot] indices (k_shadow_list)
    float2* job_geo;   // [jobcap][3] pending shadow segments (start.xy | start.z, end.x | end.yz) at [sample*cap + slot]: 24 B
    size_t jobcap;
};

// Device-resident queue sizes and counters of one worker.  The depth loop never reads a queue size back to the host: every
// kernel takes its element count from here, grids are sized for the batch's upper bound and loop (grid-stride) or exit.
struct DCtl {
    uint32_t q_groups;     // ray queue size in 64-slot groups   (k_raygen, then k_tile_prefix of the repack)
    uint32_t q_valid;      // valid entries in it
    uint32_t b_groups;     // binned queue size in groups        (k_tile_prefix of the bin stage)
    uint32_t b_valid;      // hits = valid binned entries
    uint32_t head_extend;  // persistent-kernel queue heads, reset on device between uses
    uint32_t job_count;    // pending shadow segments of this depth (k_shadow_list)
    uint32_t head_shadow;
    uint32_t overflow;     // set by k_tile_prefix when a stage's output would not fit its queue (bit 0 bin, bit 1 repack): the scatter is
                           // skipped, the stage's size becomes 0 and the host reports an internal error at the end of the share
    // statistics of the whole frame share (read back once, at the end)
    unsigned long long segments, shaded_slots, entries_sum, next_sum, shadow_jobs, _pad2;
};

// the host brackets the three shading kernels with its profiling events through these hooks
struct ShadeHooks {
    void* user;
    void (*before_fn)(void*, int);
    void (*after_fn)(void*, int);
    void before(int i) const { if (before_fn) before_fn(user, i); }
    void after(int i) const { if (after_fn) after_fn(user, i); }
};

// launch tuning of the persistent march kernels
struct Tuning {
    uint32_t persistent_blocks = 256 * 8; // 256 CUs x 8 blocks of 4 waves = 32 waves per CU
    uint32_t refill_min_extend = 16;      // parked lanes before a wave runs epilogue + queue fetch
    uint32_t refill_min_shadow = 8;
    uint32_t prefetch_min_extend = 32;    // fast path: lanes without a spare ray before the bulk queue fetch
    uint32_t prefetch_min_shadow = 32;
    bool fast_path = true;                // single-SDF specialisations k_extend1 / k_shadow1
    bool sdf_templates = true;            // per-SDF-kind instantiations of k_extend1 / k_shadow1 (off: the generic ones that read the kind from the object)
    bool box12s = true;                   // r6: the instantiation for the MandelBox in its shipped shape (12 iterations + verified short division known at compile time; RAYN_HIP_BOX12S=0 keeps the per-kind one)
    int sdf_kind = -1;                    // RAYN_SDF_* of the scene's single TracedSDF (-1: none / several), set per frame by the host: picks the instantiation
    // march_bulb.h: the shadow-march kernel written for a single-Mandelbulb scene (K rays per lane, rounds of refill / orbits / epilogues)
    bool bulb_path = true;                // use it when the scene's one SDF is a Mandelbulb ...
    bool bulb = false;                    // ... which the host decides per frame (render_device); never set by a caller
    uint32_t bulb_steps = 2;              // orbit steps per trip of the orbit phase (1 or 2; r6: 2 since rcp_sqrt_rn made a step cheaper than a trip's bookkeeping is worth - 1 / 2 / 3 / 4 steps: 643 / 613 / 641 / 677 ms of k_shadow_bulb per eighth of bulb3, profiles/r06_exp_bulb_tuning.txt)
    uint32_t bulb_rays = 3;               // rays per lane of k_shadow_bulb (2, 3 or 4)
    uint32_t bulb_orbit_min = 24;         // the orbit phase of a round ends when at most this many lanes are still inside an orbit
    uint32_t bulb_prefetch_min = 32;      // free ray slots of a wave before the bulk queue fetch
};

// sample tables: the reference layout (src/sampler.rs:11-15) + a per-(depth, sample) packed copy built at
// frame start: record = [3+VM 1-D sets | pad to 8 | 12+8*VM 2-D components] of that depth, 16-byte aligned,
// so a shading point fetches its ~20 random numbers with five 128-bit gathers instead of ~20 32-bit ones.
struct Tables {
    const float* __restrict__ s1d; const float* __restrict__ s2d; const float* __restrict__ fis;
    const float4* __restrict__ rec; uint32_t rec_stride; // float4 per record; record index = depth * spp + sample
};

} // namespace rayn

// The kernels are compiled twice (kernels.hip with RAYN_FMA_POLICY=0 -> namespace rayn_p0, =1 -> rayn_p1):
// mul_add unfused (the reference's default x86-64 build) or fused (rayn built with +fma).
#define RAYN_DECLARE_LAUNCHERS(NS)                                                                 \
    namespace NS {                                                                                 \
    using namespace rayn;                                                                          \
    void launch_pack_tables(hipStream_t s, Tables tab, float4* out, uint32_t spp, uint32_t depths, uint32_t n1, uint32_t n2); \
    void launch_raygen(hipStream_t s, const DScene* sc, Tables tab, const float* scramble, const DTile* tiles, const uint32_t* pgrp_tile, \
                       Pool pool, uint32_t* q, uint32_t n_pool, DCtl* ctl);                         \
    void launch_extend(hipStream_t s, bool count, const DScene* sc, uint32_t depth, const uint32_t* q, uint32_t max_entries, Pool pool, \
                       uint8_t* ent_obj, int single_sdf, DCtl* ctl, unsigned long long* evals, const Tuning& tun); \
    void launch_group_hist(hipStream_t s, uint32_t nclass, const uint8_t* ent_obj, uint32_t max_entries, const DCtl* ctl, uint8_t* grp_cnt); \
    void launch_scan_tile(hipStream_t s, uint32_t n_tiles, uint32_t nclass, uint32_t stride, uint32_t pad, const uint8_t* grp_cnt, \
                          const uint32_t* tgb, const uint32_t* tgc, uint32_t* grp_base, uint32_t* grp_tile, uint32_t* tile_total, \
                          uint32_t* tile_valid, uint32_t* tile_cls_cnt, const DCtl* ctl);           \
    void launch_tile_prefix(hipStream_t s, uint32_t n_tiles, const uint32_t* tile_total, const uint32_t* tile_valid, uint32_t* tile_out_base, \
                            uint32_t* ogb, uint32_t* ogc, DCtl* ctl, int stage, uint32_t nclass, uint32_t pad, const uint32_t* tile_cls_cnt, \
                            uint32_t* tile_cls_base, uint32_t cap_groups, uint32_t* base_hist);     \
    void launch_bin_scatter(hipStream_t s, uint32_t nclass, const uint32_t* q, const uint8_t* ent_obj, const uint32_t* grp_base, \
                            const uint32_t* grp_tile, const uint32_t* tile_out_base, uint32_t max_entries, uint32_t* bq, uint32_t n_tiles, \
                            const uint32_t* tile_cls_cnt, const uint32_t* tile_total, const uint32_t* tile_cls_base, const DCtl* ctl); \
    void launch_shade(hipStream_t s, bool count, const DScene* sc, Tables tab, const float* scramble, uint32_t depth, const uint32_t* bq, \
                      uint32_t max_slots, Pool pool, Nee nee, uint32_t ns, bool has_sdf, int single_sdf, unsigned long long* alive_mask, uint8_t* bgrp_cnt, DCtl* ctl, \
                      unsigned long long* evals, ShadeHooks hooks, const Tuning& tun);              \
    void launch_compact_scatter(hipStream_t s, const uint32_t* bq, const unsigned long long* alive_mask, const uint32_t* grp_base, const uint32_t* grp_tile, \
                                const uint32_t* tile_out_base, uint32_t max_slots, uint32_t* qn, uint32_t n_tiles, const uint32_t* tile_total, \
                                const DCtl* ctl);                                                   \
    void launch_unpack_tiles(hipStream_t s, const DTile* tiles, uint32_t n_tiles, uint32_t width, float* color, float* alpha, \
                             float* background, float* normal, const float* packed, size_t packed_pixels); \
    void launch_batch_setup(hipStream_t s, const DTile* tiles, uint32_t n_tiles, uint32_t* pgrp_tile, uint32_t* tgb, uint32_t* tgc); \
    void launch_resolve(hipStream_t s, const DScene* sc, const DTile* tiles, uint32_t n_tiles, uint32_t max_tile_pixels, uint32_t spp, Pool pool, \
                        float* out_color, float* out_alpha, float* out_background, float* out_normal, const uint32_t* base_hist, uint32_t hist_stride); \
    void launch_probe_dist(hipStream_t s, const DScene* sc, uint32_t hit_index, const float* pts, float* out, uint32_t n); \
    void launch_shadow_march(hipStream_t s, bool count, const DScene* sc, Nee nee, uint32_t max_jobs, int single_sdf, DCtl* ctl, unsigned long long* evals, const Tuning& tun); \
    void launch_probe_detmath(hipStream_t s, uint32_t op, const float* a, const float* b, float* out, uint32_t n); \
    void launch_verify_short_div(hipStream_t s, float n, uint32_t lo_bits, uint32_t count, uint32_t* bad); \
    }
RAYN_DECLARE_LAUNCHERS(rayn_p0)
RAYN_DECLARE_LAUNCHERS(rayn_p1)

namespace rayn {
struct KernelSet {
    decltype(&rayn_p0::launch_pack_tables) pack_tables;
    decltype(&rayn_p0::launch_raygen) raygen;
    decltype(&rayn_p0::launch_extend) extend;
    decltype(&rayn_p0::launch_group_hist) group_hist;
    decltype(&rayn_p0::launch_scan_tile) scan_tile;
    decltype(&rayn_p0::launch_tile_prefix) tile_prefix;
    decltype(&rayn_p0::launch_bin_scatter) bin_scatter;
    decltype(&rayn_p0::launch_shade) shade;
    decltype(&rayn_p0::launch_compact_scatter) compact_scatter;
    decltype(&rayn_p0::launch_batch_setup) batch_setup;
    decltype(&rayn_p0::launch_resolve) resolve;
    decltype(&rayn_p0::launch_probe_dist) probe_dist;
    decltype(&rayn_p0::launch_shadow_march) shadow_march;
    decltype(&rayn_p0::launch_probe_detmath) probe_detmath;
};
#define RAYN_KERNEL_SET(NS) KernelSet{&NS::launch_pack_tables, &NS::launch_raygen, &NS::launch_extend, &NS::launch_group_hist, &NS::launch_scan_tile, &NS::launch_tile_prefix, &NS::launch_bin_scatter, &NS::launch_shade, &NS::launch_compact_scatter, &NS::launch_batch_setup, &NS::launch_resolve, &NS::launch_probe_dist, &NS::launch_shadow_march, &NS::launch_probe_detmath}
inline KernelSet kernel_set(int fma_policy) {
    if (fma_policy) return RAYN_KERNEL_SET(rayn_p1);
    return RAYN_KERNEL_SET(rayn_p0);
}
} // namespace rayn
