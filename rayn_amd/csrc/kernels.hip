// kernels.hip — wavefront kernels of the rayn hot path for gfx950 (MI355X).
//
// Data layout in HBM (per batch of film tiles; DESIGN.md "Data layout"):
//   path pool   SoA, one slot per camera path, pixel-major / sample-minor inside a tile, tiles
//               64-aligned:  origin xyz, dir xyz, time, radiance rgb, throughput rgb, hit_t, pixel,
//               sample, term_key, depth-0 normal + object.  A path never moves; queues hold indices.
//   ray queue   u32 pool indices, dense per tile in the reference's order, each tile's segment padded
//               to a multiple of 64 (one wavefront) with INVALID.  'grp' = one 64-slot group.
//   binned q    the queue after the stable per-tile partition by hit object (object-major, insertion
//               order, every object bin padded to x4 = the f32x4 packets of HitStore::process_hits,
//               src/hitable.rs:94-134).  Four consecutive slots are one reference packet.
// Order is semantic: partition and compaction are STABLE prefix sums built from wave
// ballots (mbcnt) + a block scan — never atomics-append.
#include <hip/hip_runtime.h>

#include <stdlib.h>

#include <algorithm>

#include "device_core.h"
#include "kernels.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "kernels.hip is written for gfx950 (MI355X): k_resolve_blk<512, 8> / k_resolve_huge use 64 KB / 130 KB of the CU's 160 KB LDS"
#endif

namespace RAYN_KNS {

RD uint32_t lane_id() { return threadIdx.x & 63u; }
// number of set bits of a 64-bit wave mask below this lane
RD uint32_t mbcnt(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Per-(depth, sample) packed copy of the sample tables (see Tables in kernels.h).  One thread per record.
__global__ void __launch_bounds__(256) k_pack_tables(Tables tab, float4* __restrict__ out, uint32_t spp, uint32_t depths, uint32_t n1, uint32_t n2) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= spp * depths) return;
    const uint32_t depth = r / spp, s = r % spp;
    float* dst = (float*)(out + (size_t)r * tab.rec_stride);
    for (uint32_t k = 0; k < 8; k++) dst[k] = k < n1 ? tab.s1d[s + spp * (1 + k + depth * n1)] : 0.0f; // sets 1+k+depth*n1, src/film.rs:572
    for (uint32_t c = 0; c < n2; c++) dst[8 + c] = tab.s2d[(c & 1) + s * 2 + spp * 2 * (2 + (c >> 1) + depth * (n2 / 2))]; // src/film.rs:580-587
}

// Per-batch bookkeeping derived from the tile list ON DEVICE (the host uploads all tile lists once per frame): the tile of
// every 64-slot pool group and the tile's initial group range in the ray queue.  One block per tile.
__global__ void __launch_bounds__(256) k_batch_setup(const DTile* __restrict__ tiles, uint32_t* __restrict__ pgrp_tile,
                                                      uint32_t* __restrict__ tgb, uint32_t* __restrict__ tgc) {
    const uint32_t k = blockIdx.x;
    const DTile t = tiles[k];
    const uint32_t g0 = t.pool_base >> 6, groups = (t.n_paths + 63u) >> 6;
    if (threadIdx.x == 0) { tgb[k] = g0; tgc[k] = groups; }
    for (uint32_t g = threadIdx.x; g < groups; g += 256) pgrp_tile[g0 + g] = k;
}

// ------------------------------------------------------------------------------------------------
// a8: tile ray-gen loop, src/film.rs:456-529 (+ sample_uv :695-709, Camera::get_rays).
// One thread per pool slot.  Pool order inside a tile: x outer, y inner, sample innermost.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_raygen(const DScene* __restrict__ scp, Tables tab, const float* __restrict__ scramble,
                                                 const DTile* __restrict__ tiles, const uint32_t* __restrict__ pgrp_tile,
                                                 Pool pool, uint32_t* __restrict__ q, uint32_t n_pool, DCtl* __restrict__ ctl) {
    const uint32_t P = blockIdx.x * blockDim.x + threadIdx.x;
    if (P >= n_pool) return;
    if (P == 0) { ctl->q_groups = n_pool >> 6; ctl->q_valid = n_pool; ctl->head_extend = 0; } // n_pool is a multiple of 64
    const DScene& sc = *scp;
    const DTile tile = tiles[pgrp_tile[P >> 6]];
    const uint32_t p = P - tile.pool_base;
    pool.term_info[P] = (uint8_t)TERM_NONE;
    pool.aov[P] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(OBJ_NONE));
    if (p >= tile.n_paths) { q[P] = INVALID; return; }
    const uint32_t spp = sc.spp;
    const uint32_t s = p % spp, lpix = p / spp;
    const uint32_t lx = lpix / tile.eh, ly = lpix % tile.eh;
    const uint32_t x = tile.x0 + lx, y = tile.y0 + ly;
    const uint32_t pix = x + y * sc.width;
    const float scr = scramble[pix];
    // sample_uv
    float u0 = sample_2d(tab, spp, 0, s, scr, 0), u1 = sample_2d(tab, spp, 1, s, scr, 0);
    float fx = fis_sample(tab.fis, u0), fy = fis_sample(tab.fis, u1);
    float scx = ((float)x + 0.5f) + fx, scy = ((float)y + 0.5f) + fy;
    float uvx = sc.ndc_x * scx, uvy = sc.ndc_y * scy;
    float time = sc.time_start + sc.time_range * sample_1d(tab, spp, s, scr, 0);
    float l0 = sample_2d(tab, spp, 0, s, scr, 1), l1 = sample_2d(tab, spp, 1, s, scr, 1);
    const float t0 = sc.cam.animated ? sc.time_start + sc.time_range * sample_1d(tab, spp, s & ~3u, scr, 0) : time;
    f3 o, d;
    camera_ray(sc.cam, uvx, uvy, l0, l1, t0, &o, &d);
    pool.geo0[P] = make_float4(o.x, o.y, o.z, d.x);
    pool.geo1[P] = make_float4(d.y, d.z, 0.0f, __uint_as_float(OBJ_NONE | (s << 8)));
    pool.col0[P] = make_float4(0.0f, 0.0f, 0.0f, 1.0f); // WRay::new: radiance 0, throughput 1
    pool.col1[P] = make_float4(1.0f, 1.0f, __uint_as_float(pix), time);
    q[P] = P;
}

// ------------------------------------------------------------------------------------------------
// a10/a11/a13: HitableStore::add_hits (src/hitable.rs:170-210) for one queue entry per thread.
// Writes t + object into the pool, the object per entry, and the per-group object histogram that
// the bin scan consumes (wave ballot, no atomics).
// ------------------------------------------------------------------------------------------------
// Ray time of lane 0 of the reference packet a queue entry belongs to (queue chunks of 4; a tile's segment starts at a
// multiple of 64 and is padded only at its end, like the reference's spawned_wrays, src/film.rs:608-625).  Only needed
// when a Sphere centre is time-sequenced.
RD float packet_time(const DScene& sc, const uint32_t* __restrict__ q, const Pool& pool, uint32_t ent) {
    if (!sc.anim_spheres) return 0.0f;
    const uint32_t P0 = q[ent & ~3u];
    return P0 == INVALID ? dm_nanf() : pool.col1[P0].w;
}

// Persistent waves: march lengths vary 1..257 per ray, so a thread-per-ray launch idles most lanes.
// Here every lane owns a small state machine; a lane whose ray is finished immediately takes the
// next queue entry (wave-local chunk of 256 entries, refilled with ONE atomic per chunk), so every
// loop iteration evaluates the SDF on (almost) all 64 lanes.  Results are written by entry/pool
// index, so the fetch order never influences the output.
constexpr uint32_t CHUNK = 256;
constexpr uint32_t ENDGAME_ENTRIES = 256 * 32 * 64; // about one ray per resident lane of the chip

template <bool COUNT>
__global__ void __launch_bounds__(256) k_extend(const DScene* __restrict__ scp, uint32_t depth, const uint32_t* __restrict__ q,
                                                 DCtl* __restrict__ ctl, Pool pool, uint8_t* __restrict__ ent_obj,
                                                 uint32_t REFILL_MIN, unsigned long long* __restrict__ evals_out) {
    const DScene& sc = *scp;
    const uint32_t n_entries = ctl->q_groups << 6;
    uint32_t* const head = &ctl->head_extend;
    const uint32_t lane = lane_id();
    const Thr th = make_thr(sc, depth);
    const uint32_t nh = sc.n_hitables;
    const float c0 = 0.00005f * sc.detail_scale, c1 = 0.05f * sc.detail_scale;
    uint32_t cur = 0, end = 0; // wave-uniform chunk window
    bool exhausted = false;
    // per-lane ray state
    bool has = false, first = false, nan = false;
    uint32_t ent = 0, P = 0, k = 0, id = OBJ_NONE, m = 0, sbits = 0;
    EvalCtr evals;
    f3 o = f3{0, 0, 0}, d = f3{0, 0, 0};
    float closest = 0.0f, t = 0.0f, t0 = 0.0f;
    // fold over the hitables (src/hitable.rs:177-198) up to the next TracedSDF; finish the ray at the end
    auto advance = [&]() {
        while (k < nh) {
            const DHitable& hh = sc.h[k];
            if (hh.kind != RAYN_HITABLE_SPHERE) { first = true; return; }
            float ts = sphere_hit(hh, o, d, closest, t0);
            if (ts < closest) { closest = ts; id = k; }
            k++;
        }
        // hit_t + object byte into geo1.zw (the sample index in the upper bits is preserved)
        *(float2*)(&pool.geo1[P].z) = make_float2(closest, __uint_as_float((sbits & ~0xFFu) | id));
        ent_obj[ent] = (uint8_t)id;
        has = false;
    };
    // 'has': the lane owns a ray; 'marching': it is inside a TracedSDF march.  A lane that leaves its march
    // parks (has && !marching) until at least REFILL_MIN lanes are parked/idle: the sphere epilogue, the
    // result stores and the queue fetch then run once for many lanes instead of once per finishing lane.
    bool marching = false;
    for (;;) {
        const uint64_t parked = __ballot(!marching);
        if (parked != 0 && ((uint32_t)__popcll(parked) >= REFILL_MIN || exhausted)) {
            if (has && !marching) { advance(); marching = has; } // epilogue of the finished march
            for (;;) { // refill idle lanes until every lane is inside a march (or the queue is empty)
                const uint64_t need = __ballot(!has);
                if (need == 0) break;
                if (cur == end) {
                    if (exhausted) break;
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(head, CHUNK);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base >= n_entries) { exhausted = true; break; }
                    cur = base;
                    end = min(base + CHUNK, n_entries);
                }
                const uint32_t rank = mbcnt(need), avail = end - cur;
                if (!has && rank < avail) {
                    ent = cur + rank;
                    P = q[ent];
                    if (P == INVALID) ent_obj[ent] = (uint8_t)OBJ_NONE;
                    else {
                        const float4 g0 = pool.geo0[P], g1 = pool.geo1[P];
                        o = f3{g0.x, g0.y, g0.z};
                        d = f3{g0.w, g1.x, g1.y};
                        sbits = __float_as_uint(g1.w);
                        t0 = packet_time(sc, q, pool, ent);
                        closest = sc.t_max; id = OBJ_NONE; k = 0; has = true;
                        advance();
                        marching = has;
                    }
                }
                cur += min((uint32_t)__popcll(need), avail);
            }
        }
        const uint64_t act = __ballot(marching);
        if (act == 0) {
            if (__ballot(has) == 0 && exhausted) break;
            continue;
        }
        // all lanes of this step evaluate the same SDF object (scenes normally hold exactly one)
        const uint32_t ku = (uint32_t)__builtin_amdgcn_readlane((int)k, (int)__builtin_ctzll(act));
        const DHitable& h = sc.h[ku];
        if (marching && k == ku) { // TracedSDF::hit, src/sdf.rs:59-83, one evaluation per loop trip
            const f3 ol = o - sphere_center(h, t0); // the SDF's frame (extension: TracedSDF origin, zero in the reference)
            const f3 p = first ? ol : muladd3(d, t, ol);
            const float dist = sdf_dist<COUNT>(h, p, evals, sdf_scale(h, t0));
            bool done;
            if (first) { t = dist; nan = dist != dist; first = false; m = 0; done = sc.max_marches == 0; }
            else {
                const bool hit = __builtin_fabsf(dist) < fmaxs(c0, c1 * thr_at(th, t));
                const bool gt = t > closest;
                done = hit || nan || gt;
                if (!done) { t = t + dist; m++; done = m == sc.max_marches; }
            }
            if (done) {
                if (t < closest) { closest = t; id = k; }
                k++;
                marching = false;
            }
        }
    }
    if (COUNT && evals.n) { atomicAdd(evals_out, (unsigned long long)evals.n); atomicAdd(evals_out + 4, (unsigned long long)evals.it); }
}

// ------------------------------------------------------------------------------------------------
// Fast path of k_extend for scenes with exactly ONE TracedSDF (every shipped scene).
//  * The SDF parameters are wave-uniform and loaded once.
//  * Sphere::hit returns an intrinsic value v (t1 if t1 > 1e-4 else t2) that is "valid" iff v <= t_max,
//    and the fold (src/hitable.rs:177-198) accepts it iff v < closest-so-far.  The fold is therefore the
//    first-wins minimum over [spheres before the SDF..., SDF march(t_max = closest of those), spheres
//    after...]: the candidates of ALL spheres can be computed at fetch time, leaving two compares after
//    the march.
//  * Every lane keeps a prefetched NEXT ray (with its sphere candidates) in registers; a lane that ends
//    its march stores the result and starts the next ray in the same loop trip.  The queue fetch runs in
//    bulk when >= PREFETCH_MIN lanes have used up their spare ray, so lanes idle only at the very end.
// ------------------------------------------------------------------------------------------------
template <bool COUNT, int SDFK>
__global__ void __launch_bounds__(256) k_extend1(const DScene* __restrict__ scp, uint32_t depth, uint32_t ks, const uint32_t* __restrict__ q,
                                                  DCtl* __restrict__ ctl, Pool pool, uint8_t* __restrict__ ent_obj,
                                                  uint32_t PREFETCH_MIN, unsigned long long* __restrict__ evals_out) {
    const DScene& sc = *scp;
    const uint32_t n_entries = ctl->q_groups << 6;
    uint32_t* const head = &ctl->head_extend;
    const uint32_t lane = lane_id();
    const Thr th = make_thr(sc, depth);
    const uint32_t nh = sc.n_hitables, max_marches = sc.max_marches;
    const DHitable h = sc.h[ks]; // uniform copy
    const float c0 = 0.00005f * sc.detail_scale, c1 = 0.05f * sc.detail_scale;
    uint32_t cur = 0, end = 0;
    bool exhausted = false, endgame = false;
    // current ray
    bool c_has = false, first = false, nan = false;
    uint32_t c_P = 0, c_ent = 0, c_ids = 0, m = 0;
    EvalCtr evals;
    f3 o = f3{0, 0, 0}, d = f3{0, 0, 0};
    f3 pt = f3{0, 0, 0}; // the current ray's next march point (unused under -DRAYN_MARCH_POINT_SELECT)
    float c_pre = 0.0f, c_post = 0.0f, t = 0.0f;
    float c_scale = h.scale, n_scale = h.scale; // MandelBox scale at the packet time (extension; h.scale itself in the reference's case)
    // prefetched next ray
    bool n_has = false;
    uint32_t n_P = 0, n_ent = 0, n_ids = 0;
    f3 n_o = f3{0, 0, 0}, n_d = f3{0, 0, 0};
    float n_pre = 0.0f, n_post = 0.0f;
    for (;;) {
        const uint64_t lack = __ballot(!n_has);
        const uint64_t idle = __ballot(!c_has && !n_has);
        // endgame: once the queue is nearly drained no more spare rays are hoarded (a spare held by a lane that is still
        // inside a long march would wait while other lanes idle) - idle lanes then fetch one ray at a time
        if (!exhausted && (endgame ? idle != 0 : ((uint32_t)__popcll(lack) >= PREFETCH_MIN || (uint32_t)__popcll(idle) >= 4u))) {
            for (;;) {
                const uint64_t need = endgame ? __ballot(!n_has && !c_has) : __ballot(!n_has);
                if (need == 0) break;
                if (cur == end) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(head, CHUNK);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base >= n_entries) { exhausted = true; break; }
                    cur = base;
                    end = min(base + CHUNK, n_entries);
                    endgame = n_entries - base < ENDGAME_ENTRIES;
                }
                const uint32_t rank = mbcnt(need), avail = end - cur;
                if ((need >> lane) & 1ull) if (rank < avail) {
                    n_ent = cur + rank;
                    n_P = q[n_ent];
                    if (n_P == INVALID) ent_obj[n_ent] = (uint8_t)OBJ_NONE;
                    else {
                        const float4 g0 = pool.geo0[n_P];
                        const float2 g1 = *(const float2*)(&pool.geo1[n_P].x);
                        n_o = f3{g0.x, g0.y, g0.z};
                        n_d = f3{g0.w, g1.x, g1.y};
                        const float t0 = packet_time(sc, q, pool, n_ent);
                        float closest = sc.t_max;
                        uint32_t id = OBJ_NONE;
                        for (uint32_t k = 0; k < ks; k++) { // spheres before the SDF: the true fold
                            float ts = sphere_hit(sc.h[k], n_o, n_d, closest, t0);
                            if (ts < closest) { closest = ts; id = k; }
                        }
                        n_pre = closest;
                        uint32_t idp = OBJ_NONE;
                        for (uint32_t k = ks + 1; k < nh; k++) { // spheres after it: candidates (see header)
                            float ts = sphere_hit(sc.h[k], n_o, n_d, closest, t0);
                            if (ts < closest) { closest = ts; idp = k; }
                        }
                        n_post = closest;
                        n_ids = id | (idp << 8);
                        n_o = n_o - sphere_center(h, t0); // march in the SDF's frame (extension; zero origin in the reference)
                        n_scale = sdf_scale(h, t0);
                        n_has = true;
                    }
                }
                cur += min((uint32_t)__popcll(need), avail);
            }
        }
        if (!c_has && n_has) { // start the spare ray
            c_has = true; n_has = false;
            o = n_o; d = n_d; c_pre = n_pre; c_post = n_post; c_ids = n_ids; c_P = n_P; c_ent = n_ent; c_scale = n_scale;
            first = true;
#ifndef RAYN_MARCH_POINT_SELECT
            pt = n_o; // the first evaluation is at the origin
#endif
        }
        if (__ballot(c_has) == 0) {
            if (exhausted) break;
            continue;
        }
        if (c_has) { // TracedSDF::hit, src/sdf.rs:59-83, one evaluation per loop trip
#ifdef RAYN_MARCH_POINT_SELECT
            const f3 p = first ? o : muladd3(d, t, o);
#else
            const f3 p = pt; // r6: the march point is STATE - set at promotion (the origin) and at the end of a trip that goes on - instead of a per-trip select between the origin and o + d t
#endif
            const float dist = sdf_dist<COUNT, SDFK>(h, p, evals, c_scale);
            bool done;
            if (first) { t = dist; nan = dist != dist; first = false; m = 0; done = max_marches == 0; }
            else {
                const bool hit = __builtin_fabsf(dist) < fmaxs(c0, c1 * thr_at(th, t));
                const bool gt = t > c_pre;
                done = hit || nan || gt;
                if (!done) { t = t + dist; m++; done = m == max_marches; }
            }
            if (done) {
                float closest = c_pre;
                uint32_t id = c_ids & 0xFFu;
                if (t < closest) { closest = t; id = ks; }
                const uint32_t idp = c_ids >> 8;
                if (idp != OBJ_NONE && c_post < closest) { closest = c_post; id = idp; }
                pool.geo1[c_P].z = closest;
                ((uint8_t*)&pool.geo1[c_P].w)[0] = (uint8_t)id; // low byte of the bits word = hit object
                ent_obj[c_ent] = (uint8_t)id;
                c_has = false;
            }
#ifndef RAYN_MARCH_POINT_SELECT
            else pt = muladd3(d, t, o); // the next trip's point (Ray::point_at: the same operations the select form ran at the top of that trip)
#endif
        }
    }
    if (COUNT && evals.n) { atomicAdd(evals_out, (unsigned long long)evals.n); atomicAdd(evals_out + 4, (unsigned long long)evals.it); }
}

constexpr uint32_t QUEUE_UNROLL = 8; // groups a wave of a queue-streaming kernel has in flight per trip (r4: 4 -> 8, bin -3 %, repack -3 %)

// per-group object histogram of the extend results (input of the bin scan)
// sum over every quad of 4 lanes (two DPP quad-permute butterflies, no LDS): all four lanes end up with the quad's total
RD uint32_t quad_total(uint32_t n) {
    n += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)n, 0xB1, 0xF, 0xF, true); // quad_perm:[1,0,3,2]
    n += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)n, 0x4E, 0xF, 0xF, true); // quad_perm:[2,3,0,1]
    return n;
}
RD uint32_t count_eq4(uint32_t v, uint32_t pat) {
    const uint32_t x = v ^ pat;
    return (uint32_t)__popc(~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu)); // bytes of v equal to the class byte (exact zero-byte test)
}
__global__ void __launch_bounds__(256) k_group_hist(uint32_t nclass, const uint8_t* __restrict__ ent_obj, const DCtl* __restrict__ ctl,
                                                     uint8_t* __restrict__ grp_cnt) {
    // r4: every lane takes SIXTEEN consecutive entries (one 128-bit load), so a quad of lanes is one 64-entry group and a wave's load covers
    // 16 groups (1 KB instead of the 64 B of a byte per lane: the byte version was latency-bound at 0.8 TB/s with eight such loads in
    // flight).  Per class: the bytes equal to c are found with an exact zero-byte test on v ^ (c * 0x01010101), counted per lane (v_bcnt)
    // and summed over the quad with two DPP adds - no ballots, no scalar work.  Lane 3 of each quad packs the group's 16 class counts
    // into 16 bytes and stores them with one 128-bit store.  (A dword per lane with a 16-lane DPP row scan was 6 % slower.)
    const uint32_t n_q = ctl->q_groups << 2; // 16-byte pieces of the object bytes (n_entries is a multiple of 64)
    const uint4* __restrict__ obj16 = (const uint4*)ent_obj;
    const uint32_t stride = gridDim.x * blockDim.x;
    constexpr uint32_t HIST_UNROLL = 2;
    for (uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < n_q; i0 += stride * HIST_UNROLL) { // quads are in or out together
        uint4 v[HIST_UNROLL];
#pragma unroll
        for (uint32_t u = 0; u < HIST_UNROLL; u++) { const uint32_t i = i0 + u * stride; v[u] = i < n_q ? obj16[i] : make_uint4(~0u, ~0u, ~0u, ~0u); }
#pragma unroll
        for (uint32_t u = 0; u < HIST_UNROLL; u++) {
            const uint32_t i = i0 + u * stride;
            uint32_t acc[4] = {0u, 0u, 0u, 0u};
            for (uint32_t c = 0; c < nclass; c++) {
                const uint32_t pat = c * 0x01010101u;
                const uint32_t n = quad_total(count_eq4(v[u].x, pat) + count_eq4(v[u].y, pat) + count_eq4(v[u].z, pat) + count_eq4(v[u].w, pat));
                const uint32_t sh = n << (8u * (c & 3u));
                if ((c >> 2) == 0) acc[0] |= sh; else if ((c >> 2) == 1) acc[1] |= sh; else if ((c >> 2) == 2) acc[2] |= sh; else acc[3] |= sh;
            }
            if (i < n_q && (threadIdx.x & 3u) == 3u) *(uint4*)(grp_cnt + (size_t)(i >> 2) * SCAN_NC_BIN) = make_uint4(acc[0], acc[1], acc[2], acc[3]);
        }
    }
}
// ------------------------------------------------------------------------------------------------
// a14 (bins) / a26 (repack): per-tile stable offsets.  One block per tile walks the tile's groups
// and turns per-group class counts into tile-relative output bases:
//     out slot = class_offset[c] + (#entries of class c in earlier groups) [+ rank inside the group]
// class_offset pads every class to a multiple of 'pad' (4 for object bins = packets; 1 for the
// survivor repack, whose x4 padding rays in the reference are invalid and never binned).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_scan_tile(uint32_t nclass, uint32_t stride, uint32_t pad, const uint8_t* __restrict__ grp_cnt,
                                                    const uint32_t* __restrict__ tile_grp_begin, const uint32_t* __restrict__ tile_grp_count,
                                                    uint32_t* __restrict__ grp_base, uint32_t* __restrict__ grp_tile,
                                                    uint32_t* __restrict__ tile_total, uint32_t* __restrict__ tile_valid,
                                                    uint32_t* __restrict__ tile_cls_cnt, const DCtl* __restrict__ ctl) {
    // After an overflow (k_tile_prefix; cannot happen by the host's sizing) the group ranges of the tiles describe a queue that was
    // never written: nothing downstream may index with them.  The flag is sticky, every later stage of the share sees size 0.
    if (ctl->overflow) return;
    // Each WAVE owns whole classes (c = wave, wave + 4, ..) and walks the tile's groups 64 at a time with a wave scan and a
    // scalar running total: no block barrier inside the walk (the block-wide version spent its time in __syncthreads).
    // grp_base stays CLASS-relative; the class offsets are added by the scatter through tile_cls_base (k_tile_prefix).
    __shared__ uint32_t s_tot[SCAN_NC_BIN];
    const uint32_t k = blockIdx.x;
    const uint32_t gb = tile_grp_begin[k], gc = tile_grp_count[k];
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    for (uint32_t gi = threadIdx.x; gi < gc; gi += 256) grp_tile[gb + gi] = k;
    for (uint32_t c = wave; c < nclass; c += 4) {
        uint32_t run = 0;
        for (uint32_t base = 0; base < gc; base += 64 * QUEUE_UNROLL) {
            uint32_t v[QUEUE_UNROLL];
#pragma unroll
            for (uint32_t u = 0; u < QUEUE_UNROLL; u++) {
                const uint32_t gi = base + u * 64 + lane;
                v[u] = gi < gc ? grp_cnt[(size_t)(gb + gi) * stride + c] : 0u;
            }
#pragma unroll
            for (uint32_t u = 0; u < QUEUE_UNROLL; u++) {
                const uint32_t gi = base + u * 64 + lane;
                uint32_t inc = v[u];
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const uint32_t n = __shfl_up(inc, off);
                    if (lane >= (uint32_t)off) inc += n;
                }
                if (gi < gc) grp_base[(size_t)(gb + gi) * stride + c] = run + inc - v[u];
                run += (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
            }
        }
        if (lane == 0) s_tot[c] = run;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t off = 0, valid = 0;
        for (uint32_t c = 0; c < nclass; c++) {
            valid += s_tot[c];
            tile_cls_cnt[k * SCAN_NC_BIN + c] = s_tot[c];
            off += (s_tot[c] + pad - 1) / pad * pad;
        }
        tile_total[k] = off;
        tile_valid[k] = valid;
    }
}

// prefix over the tiles of the batch: where each tile's (64-padded) output segment starts.  The totals (output groups,
// valid entries) go to the device control block - stage 0 = bin (binned queue size; resets the shadow job list),
// stage 1 = repack (next ray queue size; resets the extend queue head) - together with the frame statistics.
__global__ void __launch_bounds__(1024) k_tile_prefix(uint32_t n_tiles, const uint32_t* __restrict__ tile_total,
                                                       const uint32_t* __restrict__ tile_valid, uint32_t* __restrict__ tile_out_base,
                                                       uint32_t* __restrict__ out_grp_begin, uint32_t* __restrict__ out_grp_count,
                                                       DCtl* __restrict__ ctl, int stage, uint32_t nclass, uint32_t pad,
                                                       const uint32_t* __restrict__ tile_cls_cnt, uint32_t* __restrict__ tile_cls_base, uint32_t cap_groups,
                                                       uint32_t* __restrict__ base_hist) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_vsum[16];
    __shared__ uint32_t s_run, s_valid;
    if (threadIdx.x == 0) { s_run = 0; s_valid = 0; }
    __syncthreads();
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    for (uint32_t base = 0; base < n_tiles; base += 1024) {
        const uint32_t k = base + threadIdx.x;
        const bool in = k < n_tiles;
        uint32_t groups = in ? (tile_total[k] + 63u) / 64u : 0u;
        uint32_t valid = in ? tile_valid[k] : 0u;
        uint32_t inc = groups, vs = valid;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t n = __shfl_up(inc, off);
            if (lane >= (uint32_t)off) inc += n;
            vs += __shfl_xor(vs, off); // butterfly sum (all lanes get the wave total)
        }
        if (lane == 63) s_wave[wave] = inc;
        if (lane == 0) s_vsum[wave] = vs;
        __syncthreads();
        uint32_t wbase = 0, tot = 0, vtot = 0;
        for (uint32_t w = 0; w < 16; w++) {
            uint32_t wt = s_wave[w];
            if (w < wave) wbase += wt;
            tot += wt;
            vtot += s_vsum[w];
        }
        const uint32_t run = s_run;
        if (in) {
            uint32_t begin = run + wbase + inc - groups;
            out_grp_begin[k] = begin;
            out_grp_count[k] = groups;
            tile_out_base[k] = begin * 64u;
            if (base_hist) base_hist[k] = begin * 64u; // bin stage: where the tile's binned segment starts at THIS depth (kept for the film resolve's 32-bit keys)
            uint32_t off = begin * 64u; // where each class bin of the tile starts (bins padded to 'pad' slots)
            for (uint32_t c = 0; c < nclass; c++) {
                tile_cls_base[k * SCAN_NC_BIN + c] = off;
                const uint32_t cnt = tile_cls_cnt[k * SCAN_NC_BIN + c];
                off += (cnt + pad - 1) / pad * pad;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) { s_run = run + tot; s_valid += vtot; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        // cannot happen by the host's sizing; never write past a queue - and once it has happened every later stage of the share is
        // empty (the group ranges written above may exceed the arrays they index: k_scan_tile and the scatters return on the flag)
        if (s_run > cap_groups) ctl->overflow |= 1u << stage;
        if (ctl->overflow) { s_run = 0; s_valid = 0; }
        if (stage == 0) {
            ctl->entries_sum += (unsigned long long)ctl->q_groups << 6;
            ctl->b_groups = s_run; ctl->b_valid = s_valid;
            ctl->segments += s_valid; ctl->shaded_slots += (unsigned long long)s_run << 6;
            ctl->job_count = 0; ctl->head_shadow = 0;
        } else {
            ctl->q_groups = s_run; ctl->q_valid = s_valid;
            ctl->next_sum += (unsigned long long)s_run << 6;
            ctl->head_extend = 0;
        }
    }
}

// INVALID padding of a tile's output segment, written by the scatter kernels themselves (no memset of the whole queue): every
// class bin is padded to a multiple of 'pad' slots and the tile's segment to a multiple of 64.  One block per tile (strided).
RD void write_tile_padding(uint32_t n_tiles, uint32_t nclass, uint32_t pad, const uint32_t* __restrict__ tile_cls_cnt,
                           const uint32_t* __restrict__ tile_total, const uint32_t* __restrict__ tile_out_base, uint32_t* __restrict__ out) {
    for (uint32_t k = blockIdx.x; k < n_tiles; k += gridDim.x) {
        const uint32_t base = tile_out_base[k], total = tile_total[k];
        if (pad > 1 && threadIdx.x < nclass) { // pad <= 4: at most 3 slots per class
            uint32_t off = 0, cnt = 0;
            for (uint32_t c = 0; c <= threadIdx.x; c++) {
                off += (cnt + pad - 1) / pad * pad;
                cnt = tile_cls_cnt[k * SCAN_NC_BIN + c];
            }
            for (uint32_t e = cnt; e < (cnt + pad - 1) / pad * pad; e++) out[base + off + e] = INVALID;
        }
        if (threadIdx.x >= 64 && threadIdx.x < 128) { // tail of the tile's segment
            const uint32_t e = total + (threadIdx.x - 64);
            if (e < ((total + 63u) & ~63u)) out[base + e] = INVALID;
        }
    }
}

// stable scatter of the queue into object bins (a14)
__global__ void __launch_bounds__(256) k_bin_scatter(uint32_t nclass, const uint32_t* __restrict__ q, const uint8_t* __restrict__ ent_obj,
                                                      const uint32_t* __restrict__ grp_base, const uint32_t* __restrict__ grp_tile,
                                                      const uint32_t* __restrict__ tile_out_base, const DCtl* __restrict__ ctl,
                                                      uint32_t* __restrict__ bq, uint32_t n_tiles, const uint32_t* __restrict__ tile_cls_cnt,
                                                      const uint32_t* __restrict__ tile_total, const uint32_t* __restrict__ tile_cls_base) {
    if (ctl->overflow) return;
    write_tile_padding(n_tiles, nclass, 4, tile_cls_cnt, tile_total, tile_out_base, bq);
    const uint32_t n_entries = ctl->q_groups << 6;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < n_entries; i0 += stride * QUEUE_UNROLL) {
        uint32_t obj[QUEUE_UNROLL], ref[QUEUE_UNROLL], tbase[QUEUE_UNROLL], gbase[QUEUE_UNROLL];
#pragma unroll
        for (uint32_t u = 0; u < QUEUE_UNROLL; u++) { // first wave of loads: object, queue entry, tile of the group
            const uint32_t i = i0 + u * stride;
            const bool in = i < n_entries;
            obj[u] = in ? ent_obj[i] : OBJ_NONE;
            ref[u] = in ? q[i] : INVALID;
            tbase[u] = grp_tile[__builtin_amdgcn_readfirstlane((int)(in ? i >> 6 : 0u))]; // per group = wave-uniform: a scalar load
        }
#pragma unroll
        for (uint32_t u = 0; u < QUEUE_UNROLL; u++) { // second wave: the two dependent lookups
            const uint32_t i = i0 + u * stride;
            const bool live = obj[u] != OBJ_NONE;
            gbase[u] = live ? grp_base[(i >> 6) * SCAN_NC_BIN + obj[u]] : 0u;
            tbase[u] = live ? tile_cls_base[tbase[u] * SCAN_NC_BIN + obj[u]] : 0u;
        }
#pragma unroll
        for (uint32_t u = 0; u < QUEUE_UNROLL; u++) {
            if (i0 + u * stride >= n_entries) break; // wave-uniform
            uint32_t rank = 0;
            for (uint32_t c = 0; c < nclass; c++) {
                uint64_t m = __ballot(obj[u] == c);
                if (obj[u] == c) rank = mbcnt(m);
            }
            if (obj[u] != OBJ_NONE) bq[tbase[u] + gbase[u] + rank] = ref[u];
        }
    }
}

// stable compaction of the survivors of a shade pass into the next ray queue (a26)
// (survivor flags arrive as ONE 64-bit ballot per 64-slot group, written by k_shade_setup: an eighth of the bytes of a flag per slot,
// and the rank inside the group is mbcnt of that mask - no per-slot byte load, no ballot here)
__global__ void __launch_bounds__(256) k_compact_scatter(const uint32_t* __restrict__ bq, const unsigned long long* __restrict__ alive_mask,
                                                          const uint32_t* __restrict__ grp_base, const uint32_t* __restrict__ grp_tile,
                                                          const uint32_t* __restrict__ tile_out_base, const DCtl* __restrict__ ctl,
                                                          uint32_t* __restrict__ qn, uint32_t n_tiles, const uint32_t* __restrict__ tile_total) {
    if (ctl->overflow) return;
    write_tile_padding(n_tiles, 1, 1, nullptr, tile_total, tile_out_base, qn);
    const uint32_t n_slots = ctl->b_groups << 6;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t j0 = blockIdx.x * blockDim.x + threadIdx.x; j0 < n_slots; j0 += stride * QUEUE_UNROLL) {
        uint32_t a[QUEUE_UNROLL], ref[QUEUE_UNROLL], tbase[QUEUE_UNROLL], gbase[QUEUE_UNROLL];
        unsigned long long am[QUEUE_UNROLL];
#pragma unroll
        for (uint32_t u = 0; u < QUEUE_UNROLL; u++) {
            const uint32_t j = j0 + u * stride;
            const bool in = j < n_slots; // wave-uniform: a wave's 64 slots are one group
            // everything that is per GROUP is wave-uniform: survivor mask, tile, output base -> scalar loads; one vector load (bq) per slot
            const uint32_t g = (uint32_t)__builtin_amdgcn_readfirstlane((int)(in ? j >> 6 : 0u));
            am[u] = in ? alive_mask[g] : 0ull;
            a[u] = (uint32_t)(am[u] >> (j & 63u)) & 1u;
            ref[u] = in ? bq[j] : INVALID; // (loading survivors only - a load that waits for the mask - was 50 % SLOWER: the scalar mask load then sits in front of every vector load)
            tbase[u] = grp_tile[g];
            gbase[u] = grp_base[g];
        }
#pragma unroll
        for (uint32_t u = 0; u < QUEUE_UNROLL; u++) tbase[u] = tile_out_base[__builtin_amdgcn_readfirstlane((int)tbase[u])];
#pragma unroll
        for (uint32_t u = 0; u < QUEUE_UNROLL; u++) {
            if (j0 + u * stride >= n_slots) break; // wave-uniform
            const uint32_t rank = mbcnt(am[u]);
            if (a[u]) qn[tbase[u] + gbase[u] + rank] = ref[u];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// a15-a24: shading = three kernels per depth.
//   k_shade_setup  (one thread per binned slot): get_shading_info, emission, the light samples of
//                  NEE (surface 4x, volume 4xVM), BSDF scatter, roulette, AOVs, new ray.  For every
//                  NEE sample it stores the UNOCCLUDED contribution x = (Le*f)*transmission and its
//                  pdf, resolves the analytic-sphere part of HitableStore::test_occluded inline and
//                  appends a shadow job when an SDF march is still needed.
//   k_shadow       (persistent waves over the job list): TracedSDF::occluded, src/sdf.rs:25-57.
//   k_shade_finish (one thread per slot): radiance += ((x*occluded)/pdf) * throughput * ... in the
//                  reference's order (src/integrator.rs:91-92,128-129), then publishes the new throughput.
// Slots 4k..4k+3 of the binned queue are one reference packet; the only cross-lane data are the light
// indices each lane draws from ITS OWN 1-D sample (src/integrator.rs:76-82,100-110), exchanged with
// wave shuffles.  Padding lanes use sample 0 / scramble 0 (Ray::new_invalid, src/ray.rs:54-66).
// ------------------------------------------------------------------------------------------------
RD bool all_zero(f3 v) { return v.x == 0.0f && v.y == 0.0f && v.z == 0.0f; }
constexpr uint32_t VOL_MEMO_LIGHTS = 7; // per-light volume terms memoised in LDS (3 floats per light and thread)

constexpr int SETUP_WAVES = 6; // waves per SIMD the register budget of k_shade_setup is set for (80 VGPRs; 5 / 7 / 8 measured slower)
// One slot per thread over a grid sized for the batch's upper bound (surplus blocks exit at once; a grid-stride loop carries
// j and the count across a body that already spills: +44 B of scratch, 22 % slower).
template <bool COUNT>
__global__ void __launch_bounds__(256, SETUP_WAVES) k_shade_setup(const DScene* __restrict__ scp, Tables tab, const float* __restrict__ scramble,
                                                      uint32_t depth, const uint32_t* __restrict__ bq, const DCtl* __restrict__ ctl, Pool pool, Nee nee,
                                                      unsigned long long* __restrict__ alive_mask, uint8_t* __restrict__ bgrp_cnt,
                                                      unsigned long long* __restrict__ evals_out) {
    const DScene& sc = *scp;
    const uint32_t lane = lane_id();
    const uint32_t n_slots = ctl->b_groups << 6;
    EvalCtr evals;
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_slots) return; // whole waves: n_slots % 64 == 0
    const uint32_t P = bq[j];
    const bool valid = P != INVALID;
    const uint32_t spp = sc.spp, nl = sc.n_lights, VM = sc.vm;
    uint32_t sample = 0, pix = 0;
    float scr = 0.0f;
    float4 g0 = make_float4(0, 0, 0, 0), g1 = g0, c0 = g0, c1 = g0;
    if (valid) {
        g0 = pool.geo0[P]; g1 = pool.geo1[P]; c0 = pool.col0[P]; c1 = pool.col1[P];
        sample = __float_as_uint(g1.w) >> 8; pix = __float_as_uint(c1.z); scr = scramble[pix];
    }
    // lane 0 of the packet supplies the time of closure-sequenced sphere centres (get_shading_info, occluded)
    const float t0 = sc.anim_spheres ? __shfl(c1.w, (int)(lane & ~3u)) : 0.0f;
    // this lane's random numbers of this depth (raw table values; the pixel scramble is added per use)
    // Random numbers (and light picks) are only consumed by packets of light-receiving objects, or by every segment when
    // the volume scatters; bins are object-major, so whole waves of sky / emissive hits skip the table fetch.
    const bool recv_l = valid && sc.m[sc.h[__float_as_uint(g1.w) & 0xFFu].material].receives_light != 0;
    const bool pk_recv = __shfl((int)recv_l, (int)(lane & ~3u)) != 0; // lane 0 of a bin packet is always a real hit
    const bool wave_needs_samples = sc.has_scatter || __ballot(pk_recv) != 0;
    const float4* rec = tab.rec + (size_t)(depth * spp + sample) * tab.rec_stride;
    float4 r1d = make_float4(0, 0, 0, 0);
    float r1d4 = 0.0f;
    if (wave_needs_samples) { r1d = rec[0]; r1d4 = rec[1].x; }
    auto s1 = [&](uint32_t k) { // samples_1d[k] of this depth, src/film.rs:568-574 + Samples::sample_1d
        const float raw = k == 0 ? r1d.x : (k == 1 ? r1d.y : (k == 2 ? r1d.z : (k == 3 ? r1d.w : r1d4)));
        return dm_fractf(raw + scr);
    };
    // light picks: each lane draws an index from ITS 1-D sample; the packet's four picks are packed
    // 4 bits each (n_lights <= 16) so the rolled loops below need no register arrays.
    uint32_t surf_picks = 0;
    unsigned long long vol_picks = 0;
    if (nl > 0 && wave_needs_samples) {
        const uint32_t base = lane & ~3u;
        for (uint32_t k = 0; k < 1 + VM; k++) {
            uint32_t mine = light_index(s1(k), nl);
            uint32_t packed = 0;
#pragma unroll
            for (uint32_t i = 0; i < 4; i++) packed |= (uint32_t)__shfl(mine, base + i) << (4 * i);
            if (k == 0) surf_picks = packed;
            else vol_picks |= (unsigned long long)packed << (16 * (k - 1));
        }
    }
    const size_t cap = nee.cap;
    const bool scene_has_sdf = sc.n_sdf > 0;
    bool is_alive = false;
    uint32_t flags = 0;
    // shadow jobs live IN PLACE: vis[s][slot] = 2 marks "SDF march pending" and the segment is parked at
    // job_geo[3*(s*cap + slot)..]; k_shadow_list collects the pending (sample, slot) pairs.  (A compacted job list would need
    // one atomic per wave per sample on a single counter - that alone cost 0.5 s per frame.)
    auto park_job = [&](uint32_t s, f3 a, f3 b) {
        const size_t idx = s * cap + j; // 24 contiguous bytes per segment (the packet time of a moving SDF is per slot: nee.t0)
        nee.job_geo[3 * idx] = make_float2(a.x, a.y);
        nee.job_geo[3 * idx + 1] = make_float2(a.z, b.x);
        nee.job_geo[3 * idx + 2] = make_float2(b.y, b.z);
    };
    // analytic spheres of test_occluded (every factor is exactly 0 or 1 -> order independent)
    auto spheres_visible = [&](f3 a, f3 b) {
        f3 dir = b - a;
        const float dist = mag(dir);
        dir = div_by_mag(dir, dist);
        for (uint32_t k = 0; k < sc.n_hitables; k++)
            if (sc.h[k].kind == RAYN_HITABLE_SPHERE && sphere_occluded_dir(sc.h[k], a, dir, dist, t0) == 0.0f) return false;
        return true;
    };
    f3 o = f3{0, 0, 0}, d = f3{0, 0, 0}, rad = f3{0, 0, 0}, thr = f3{0, 0, 0}, point = f3{0, 0, 0}, normal = f3{0, 0, 1};
    float t = 0.0f, offset_by = 0.0f, vol_T = 1.0f;
    uint32_t obj = 0;
    bool receives = false;
    if (valid) {
        o = f3{g0.x, g0.y, g0.z};
        d = f3{g0.w, g1.x, g1.y};
        rad = f3{c0.x, c0.y, c0.z};
        thr = f3{c0.w, c1.x, c1.y};
        t = g1.z;
        obj = __float_as_uint(g1.w) & 0xFFu;
        const DHitable& h = sc.h[obj];
        point = muladd3(d, t, o); // WHit::point -> Ray::point_at
        if (h.kind == RAYN_HITABLE_SPHERE) { // src/sphere.rs:73-86
            normal = normalized(point - sphere_center(h, t0));
            offset_by = 0.0f;
        } else { // src/sdf.rs:85-101
            Thr th = make_thr(sc, depth);
            float hps = fmaxs(0.0001f, sc.detail_scale * thr_at(th, t));
            normal = sdf_normal<COUNT>(h, point - sphere_center(h, t0), hps, evals, sdf_scale(h, t0));
            offset_by = hps;
        }
        vol_T = sc.has_extinct ? dmf_expf(-sc.rho_t * t) : 1.0f;
        receives = sc.m[h.material].receives_light != 0;
        rad = rad + bsdf_le(sc.m[h.material], -d) * thr * vol_T;
        pool.col0[P] = make_float4(rad.x, rad.y, rad.z, c0.w); // throughput.r stays: k_shade_finish needs the old one
    }
    const DMaterial& mat = sc.m[sc.h[obj].material];
    const f3 wo = -d;
    // ---- r6: segments whose throughput is EXACTLY (0, 0, 0) ------------------------------------------------------------------------
    // DielectricBSDF::scatter zeroes f when the picked specular lobe lands below the horizon (src/material.rs:240-243) and roulette cannot
    // end a path before depth 3 (src/integrator.rs:147-156), so 9 % of the shaded segments of config 3 (24 % at depth 2, 33 % at depth 3;
    // zero is absorbing: 0 * f stays 0, a NaN product keeps the old throughput, src/integrator.rs:181) arrive with throughput 0.  Every NEE
    // term of such a segment is   rad + ((x * occluded) / pdf) * thr * corr * (vol_T | rho_s * aux)   (src/integrator.rs:91-92,128-129):
    // with |x| <= 2^60 and |pdf| >= 2^-60 (infinite too; negative too: a march that ends inside the Mandelbulb returns a negative hit distance and with it
    // a negative equi-angular pdf - 15 % of the zero-throughput samples of bulb3) the quotient is FINITE for occluded = 0 and for 1, finite * (+-0) is +-0, it stays +-0
    // through the finite factors that follow, and rad + (+-0) == rad bit for bit (rad is never -0: it starts at +0 and a sum is -0 only
    // when both terms are).  The visibility of such a sample cannot reach the film: no sphere tests, no parked segment, no shadow march
    // (vis stays 1; k_shade_finish adds its exact zero).  The path itself goes on: its hit object still decides packet membership
    // (src/hitable.rs:100-133).  A sample outside the bounds (NaN, overflow: a degenerate half vector, a light of absurd power) is tested
    // and marched as before, so inf * 0 / NaN cases keep their bits (FILM case "huge_lights").
    constexpr float ELIDE_MAX_X = 1.1529215e18f /* 2^60 */, ELIDE_MIN_PDF = 8.6736174e-19f /* 2^-60 */, FINITE_MAX = 3.40282347e+38f;
    const bool zero_w = valid && all_zero(thr) && __builtin_fabsf(vol_T) <= FINITE_MAX && __builtin_fabsf(sc.rho_s) <= FINITE_MAX;
    uint32_t n_elided_jobs = 0, n_oob = 0; // COUNT builds only: the shadow segments the elision did not park; samples of zero-weight slots outside the bounds
    // ---- surface NEE, surface_sample_one_light src/integrator.rs:207-240
    const bool do_surf = valid && receives && nl > 0;
    {
        if (do_surf) { flags |= 2u; nee.T[j] = vol_T; }
        for (uint32_t i = 0; i < 4; i++) {
            if (!do_surf) nee.vis[i * cap + j] = 1; // nothing pending for this (sample, slot)
            else {
                const DLight& L = sc.l[(surf_picks >> (4 * i)) & 15u];
                const float4 rs = rec[2 + (i >> 1)]; // 2-D components 2i, 2i+1
                float u0 = dm_fractf(((i & 1) ? rs.z : rs.x) + scr), u1 = dm_fractf(((i & 1) ? rs.w : rs.y) + scr);
                f3 end_point; float pdf;
                light_sample(L, u0, u1, point, &end_point, &pdf);
                f3 wi = end_point - point;
                float dist = mag(wi);
                wi = div_by_mag(wi, dist);
                f3 occlude_point = point + normal * signum(dot(normal, wi)) * offset_by;
                const float ndw = dot(normal, wi);
                const float cosw = fmaxs(ndw, 0.0f);
                f3 f;
                // light below the horizon: bsdf.f(..) * 0 is +0 whenever f is finite, which it is for finite
                // inputs with a non-degenerate half vector -> skip the pow/normalize of BSDF::f (bit-identical)
                if (cosw == 0.0f && ndw == ndw && mag_sq(wo + wi) > 0.0f) f = f3{0.0f, 0.0f, 0.0f};
                else f = bsdf_f(mat, wo, wi, normal) * cosw;
                float tr = sc.has_extinct ? dmf_expf(-sc.rho_t * dist) : 1.0f;
                f3 x = L.emission * f * tr;
                nee.x[(i * 3 + 0) * cap + j] = x.x; nee.x[(i * 3 + 1) * cap + j] = x.y; nee.x[(i * 3 + 2) * cap + j] = x.z;
                nee.pdf[i * cap + j] = pdf;
                uint8_t vis = 1;
                // x == 0 (light below the horizon): (x*occluded)/pdf is the same zero for occluded 0 or 1 -> no test needed
                const bool zw = zero_w && __builtin_fabsf(x.x) <= ELIDE_MAX_X && __builtin_fabsf(x.y) <= ELIDE_MAX_X && __builtin_fabsf(x.z) <= ELIDE_MAX_X && __builtin_fabsf(pdf) >= ELIDE_MIN_PDF;
                if (COUNT) { if (zw && !all_zero(x) && scene_has_sdf && spheres_visible(occlude_point, end_point)) n_elided_jobs++; if (zero_w && !zw) n_oob++; }
                if (!all_zero(x) && !zw) {
                    if (!spheres_visible(occlude_point, end_point)) vis = 0;
                    else if (scene_has_sdf) { vis = 2; park_job(i, occlude_point, end_point); }
                }
                nee.vis[i * cap + j] = vis;
            }
        }
    }
    // ---- volume NEE, volume_sample_one_light src/integrator.rs:242-281 (runs for every segment)
    const bool do_vol = valid && sc.has_scatter && nl > 0;
    if (sc.has_scatter) {
        if (do_vol) flags |= 4u;
        const float vsample = s1(1); // samples_1d[1] for every march
        // Every march of a segment uses the SAME 1-D sample (src/integrator.rs:100-131), so the equi-angular distance, its pdf
        // and the transmittance to it depend on the light alone: with fewer lights than samples (5 vs 4*VM = 8 in the shipped
        // scene) they are computed once per light into this thread's LDS slots and looked up per sample - same inputs, same
        // functions, same bits, 3/8 fewer atan2/tan/exp calls.
        extern __shared__ float s_vol[]; // [VOL_MEMO_LIGHTS][3][256]
        const bool memo = nl <= VOL_MEMO_LIGHTS && nl < 4u * VM;
        if (memo)
            for (uint32_t li = 0; li < nl; li++) {
                float vd = 0.0f, vp = 0.0f, va = 1.0f;
                if (do_vol) {
                    light_sample_volume(sc.l[li], vsample, o, d, t, &vd, &vp);
                    va = sc.has_extinct ? dmf_expf(-sc.rho_t * vd) : 1.0f;
                }
                s_vol[(li * 3 + 0) * 256 + threadIdx.x] = vd;
                s_vol[(li * 3 + 1) * 256 + threadIdx.x] = vp;
                s_vol[(li * 3 + 2) * 256 + threadIdx.x] = va;
            }
        for (uint32_t march = 0; march < VM; march++) {
            for (uint32_t i = 0; i < 4; i++) {
                const uint32_t s = 4 + 4 * march + i;
                if (!do_vol) nee.vis[s * cap + j] = 1;
                else {
                    const uint32_t li = (uint32_t)(vol_picks >> (16 * march + 4 * i)) & 15u;
                    const DLight& L = sc.l[li];
                    const float4 rs = rec[4 + 2 * march + (i >> 1)]; // comps 8+8*march+2i, +1
                    float u0 = dm_fractf(((i & 1) ? rs.z : rs.x) + scr), u1 = dm_fractf(((i & 1) ? rs.w : rs.y) + scr);
                    float vdist, vpdf, vaux;
                    if (memo) {
                        vdist = s_vol[(li * 3 + 0) * 256 + threadIdx.x];
                        vpdf = s_vol[(li * 3 + 1) * 256 + threadIdx.x];
                        vaux = s_vol[(li * 3 + 2) * 256 + threadIdx.x];
                    } else {
                        light_sample_volume(L, vsample, o, d, t, &vdist, &vpdf);
                        vaux = sc.has_extinct ? dmf_expf(-sc.rho_t * vdist) : 1.0f;
                    }
                    f3 sp = o + d * vdist;
                    f3 end_point; float lpdf;
                    light_sample(L, u0, u1, sp, &end_point, &lpdf);
                    float dl = mag(end_point - sp);
                    float tr = sc.has_extinct ? dmf_expf(-sc.rho_t * dl) : 1.0f;
                    // x = L.emission * f * tr is rebuilt by k_shade_finish from the light index and tr (same operations, same bits)
                    nee.vtr[(s - 4) * cap + j] = tr;
                    nee.pdf[s * cap + j] = vpdf * lpdf;
                    nee.aux[(s - 4) * cap + j] = vaux;
                    uint8_t vis = 1;
                    // zero-weight slot (see above): x = Le * (1/(4 pi) * tr) with the constant factor <= 1 and tr the only variable, pdf = vpdf * lpdf, then * rho_s * aux
                    const bool zw = zero_w && __builtin_fabsf(L.emission.x * tr) <= ELIDE_MAX_X && __builtin_fabsf(L.emission.y * tr) <= ELIDE_MAX_X &&
                                    __builtin_fabsf(L.emission.z * tr) <= ELIDE_MAX_X && __builtin_fabsf(vpdf * lpdf) >= ELIDE_MIN_PDF && __builtin_fabsf(vaux) <= FINITE_MAX;
                    if (COUNT) { if (zw && scene_has_sdf && spheres_visible(sp, end_point)) n_elided_jobs++; if (zero_w && !zw) n_oob++; }
                    if (zw) {}
                    else if (!spheres_visible(sp, end_point)) vis = 0;
                    else if (scene_has_sdf) { vis = 2; park_job(s, sp, end_point); }
                    nee.vis[s * cap + j] = vis;
                }
            }
        }
    }
    if (do_vol) nee.vpicks[j] = vol_picks;
    if (sc.anim_spheres) nee.t0[j] = t0;
    // ---- BSDF sample, roulette, AOVs, termination (src/integrator.rs:134-203)
    if (valid) {
        if (receives) {
            const Basis basis = orthonormal_basis(normal);
            const float4 rb = rec[4 + 2 * VM]; // comps 8+8*VM .. +3
            float s3 = s1(3), s4 = s1(4);
            Scatter se = bsdf_scatter(mat, wo, normal, basis, s3, dm_fractf(rb.x + scr), dm_fractf(rb.y + scr), dm_fractf(rb.z + scr), dm_fractf(rb.w + scr));
            float ndl = __builtin_fabsf(dot(se.wi, normal));
            f3 nthr = thr * vol_T * se.f * ndl / se.pdf;
            float rr = 0.0f;
            if (depth > 2) {
                rr = fmaxs(1.0f - component_max(thr), 0.05f);
                nthr = nthr / (1.0f - rr);
            }
            if (depth == 0) { // Alpha + WorldNormal AOVs, src/integrator.rs:161-169
                pool.aov[P] = make_float4(normal.x, normal.y, normal.z, __uint_as_float(obj));
            }
            if (depth >= sc.max_bounces || s4 < rr) {
                pool.term_key[P] = j; pool.term_info[P] = (uint8_t)depth; // ChannelSample::Color(ray.radiance)
            } else {
                if (!any_nan(nthr)) thr = nthr;
                f3 no = point + normal * signum(dot(normal, se.wi)) * offset_by; // create_rays, src/hitable.rs:42-47
                pool.geo0[P] = make_float4(no.x, no.y, no.z, se.wi.x);
                *(float2*)(&pool.geo1[P].x) = make_float2(se.wi.y, se.wi.z);
                // the finish kernel still needs the OLD throughput: park the new one beside the slot
                nee.nthr[j] = thr.x; nee.nthr[cap + j] = thr.y; nee.nthr[2 * cap + j] = thr.z;
                is_alive = true;
            }
        } else {
            pool.term_key[P] = j; pool.term_info[P] = (uint8_t)(depth | (depth == 0 ? 0x80u : 0u)); // Background at depth 0, else Color
        }
        nee.flags[j] = (uint8_t)(flags | (is_alive ? 1u : 0u));
    }
    const uint64_t m = __ballot(is_alive); // the group's survivors: count for the repack scan, mask for the scatter's ranks
    if (lane == 0) { alive_mask[j >> 6] = m; bgrp_cnt[j >> 6] = (uint8_t)__popcll(m); }
    if (COUNT && evals.n) { atomicAdd(evals_out, (unsigned long long)evals.n); atomicAdd(evals_out + 4, (unsigned long long)evals.it); }
    if (COUNT) { // elision accounting: [3] zero-weight slots, [7] shadow segments they would have parked, [8] their samples outside the bounds (evals_out = &evals[1])
        const uint64_t em = __ballot(zero_w);
        uint32_t nj = n_elided_jobs, no = n_oob;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { nj += (uint32_t)__shfl_xor((int)nj, off); no += (uint32_t)__shfl_xor((int)no, off); }
        if (lane == 0 && em) {
            atomicAdd(evals_out + 2, (unsigned long long)__popcll(em)); atomicAdd(evals_out + 6, (unsigned long long)nj); atomicAdd(evals_out + 7, (unsigned long long)no);
        }
    }
}

// Dense list of the pending (sample, slot) pairs.  A block scans 256 * SCAN_ITEMS ids per trip; each wave takes whole 64-id groups
// (n_slots is a multiple of 64, so a group lies inside one sample and its sample index is wave-uniform: no per-item
// division), counts its pending items with ballots, the block reserves its range with ONE atomic and every (wave, group) writes
// its refs densely in lane order - coalesced stores, and consecutive list entries point at consecutive segments, which keeps
// the shadow kernel's fetch contiguous.  List order is irrelevant to the result: it is written back by index.
// Two things bounded this kernel at 1.5 TB/s in r1, and only fixing BOTH helped (ablations: without the stores, or without the
// loads, it ran exactly as fast): (1) the appends go through one counter and a single address takes ~90-100 atomics/us
// (MI355X_MICROARCH.md "dequeue") - 16 ids per thread meant 13.4 M atomics per config-3 frame = the kernel's 140 ms; 64 ids per
// thread = 16 K ids per atomic; (2) a uniform branch around every visibility load serialised them (load - wait - ballot, once
// per group) - all loads of a trip are now issued before the first ballot.  Together: 16.5 -> 7.7 ms per 1/8 share of config 3
// (3.5 TB/s); either one alone: 16.5 -> 15.5-16.2 ms.
constexpr uint32_t SCAN_ITEMS = 64;
__global__ void __launch_bounds__(256) k_shadow_list(Nee nee, uint32_t ns, DCtl* __restrict__ ctl) {
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_base;
    const uint32_t n_slots = ctl->b_groups << 6;
    const uint32_t n_ids = ns * n_slots;
    uint32_t* const job_count = &ctl->job_count;
    const size_t cap = nee.cap;
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    for (uint32_t block_first = blockIdx.x * (256 * SCAN_ITEMS); block_first < n_ids; block_first += gridDim.x * (256 * SCAN_ITEMS)) {
    // All SCAN_ITEMS visibility loads of a wave are issued before the first ballot waits for one (a uniform branch around every
    // load used to serialise them: load - wait - ballot, 16 times per chunk, which made the kernel latency-bound at 1.5 TB/s):
    // groups beyond the end load slot 0 and are masked afterwards.
    uint32_t refs[SCAN_ITEMS];
    uint8_t vis[SCAN_ITEMS];
#pragma unroll
    for (uint32_t r = 0; r < SCAN_ITEMS; r++) {
        const uint32_t g = (uint32_t)__builtin_amdgcn_readfirstlane((int)(block_first + (r * 4 + wave) * 64)); // first id of the group
        const bool in = g < n_ids;
        const uint32_t gc = in ? g : 0u;
        const uint32_t s = gc / n_slots;                                       // scalar
        refs[r] = (uint32_t)(s * cap + (gc - s * n_slots) + lane);             // < 2^32, checked on the host
        vis[r] = nee.vis[refs[r]];
        if (!in) refs[r] = INVALID;
    }
    uint32_t wave_total = 0; // wave-uniform
#pragma unroll
    for (uint32_t r = 0; r < SCAN_ITEMS; r++) {
        if (vis[r] != 2) refs[r] = INVALID;
        wave_total += (uint32_t)__popcll(__ballot(refs[r] != INVALID));
    }
    if (lane == 0) s_wave[wave] = wave_total;
    __syncthreads();
    uint32_t before = 0, tot = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4; w++) {
        const uint32_t wt = s_wave[w];
        if (w < wave) before += wt;
        tot += wt;
    }
    if (threadIdx.x == 0) s_base = tot ? atomicAdd(job_count, tot) : 0u;
    __syncthreads();
    uint32_t w = s_base + before;
#pragma unroll
    for (uint32_t r = 0; r < SCAN_ITEMS; r++) {
        const uint64_t m = __ballot(refs[r] != INVALID);
        if (refs[r] != INVALID) nee.job_ref[w + mbcnt(m)] = refs[r];
        w += (uint32_t)__popcll(m);
    }
    __syncthreads(); // s_wave / s_base are reused by the next chunk
    }
}

// TracedSDF::occluded (src/sdf.rs:25-57) for the pending shadow segments, persistent waves (see k_extend).
template <bool COUNT>
__global__ void __launch_bounds__(256) k_shadow(const DScene* __restrict__ scp, Nee nee, DCtl* __restrict__ ctl,
                                                 uint32_t REFILL_MIN, unsigned long long* __restrict__ evals_out) {
    const DScene& sc = *scp;
    const uint32_t lane = lane_id();
    const uint32_t n_jobs = ctl->job_count, nh = sc.n_hitables;
    uint32_t* const head = &ctl->head_shadow;
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->shadow_jobs += n_jobs;
    const float c0 = 0.0001f * sc.detail_scale, c1 = 0.00001f * sc.detail_scale;
    uint32_t cur = 0, end = 0;
    bool exhausted = false;
    bool has = false, first = false, nan = false;
    uint32_t ref = 0, k = 0, m = 0;
    EvalCtr evals;
    f3 start = f3{0, 0, 0}, dir = f3{0, 0, 0}, wa = f3{0, 0, 0}, wb = f3{0, 0, 0};
    float max_dist = 0.0f, t = 0.0f, jt0 = 0.0f;
    auto next_sdf = [&]() { // advance k to the next TracedSDF; none left -> the segment is visible
        while (k < nh && sc.h[k].kind == RAYN_HITABLE_SPHERE) k++;
        if (k >= nh) { nee.vis[ref] = 1; has = false; }
        else { // the segment in this SDF's frame (TracedSDF::occluded works on start - origin, end - origin)
            const f3 origin = sphere_center(sc.h[k], jt0);
            start = wa - origin;
            dir = (wb - origin) - start;
            max_dist = mag(dir);
            dir = div_by_mag(dir, max_dist);
            first = true;
        }
    };
    for (;;) {
        const uint64_t idle = __ballot(!has);
        if (idle != 0 && ((uint32_t)__popcll(idle) >= REFILL_MIN) && !exhausted) {
            for (;;) {
                const uint64_t need = __ballot(!has);
                if (need == 0) break;
                if (cur == end) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(head, CHUNK);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base >= n_jobs) { exhausted = true; break; }
                    cur = base;
                    end = min(base + CHUNK, n_jobs);
                }
                const uint32_t rank = mbcnt(need), avail = end - cur;
                if (!has && rank < avail) {
                    ref = nee.job_ref[cur + rank]; // [sample][slot]
                    const float2 j0 = nee.job_geo[3 * (size_t)ref], j1 = nee.job_geo[3 * (size_t)ref + 1], j2 = nee.job_geo[3 * (size_t)ref + 2];
                    const float4 ja = make_float4(j0.x, j0.y, j1.x, j1.y), jb = make_float4(j2.x, j2.y, sc.anim_spheres ? nee.t0[ref % (uint32_t)nee.cap] : 0.0f, 0.0f);
                    wa = f3{ja.x, ja.y, ja.z};
                    wb = f3{ja.w, jb.x, jb.y};
                    jt0 = jb.z;
                    k = 0; has = true;
                    next_sdf();
                }
                cur += min((uint32_t)__popcll(need), avail);
            }
        }
        const uint64_t act = __ballot(has);
        if (act == 0) break;
        const uint32_t ku = (uint32_t)__builtin_amdgcn_readlane((int)k, (int)__builtin_ctzll(act));
        const DHitable& h = sc.h[ku];
        if (has && k == ku) {
            const f3 p = first ? start : muladd3(dir, t, start);
            const float dist = sdf_dist<COUNT>(h, p, evals, sdf_scale(h, jt0));
            int res = -1; // -1 keep marching, 0 occluded, 1 this SDF does not occlude
            if (first) {
                t = dist; nan = dist != dist; first = false; m = 0;
                if (sc.max_vis_marches == 0) res = ((dist < 0.0001f) && !((dist > max_dist) || nan)) ? 0 : 1;
                else if ((t > max_dist) || nan) res = 1;
            } else {
                if (__builtin_fabsf(dist) < fmaxs(c0, c1 * t)) res = 0;
                else {
                    t = t + dist; m++;
                    if (m == sc.max_vis_marches || (t > max_dist) || nan) res = 1;
                }
            }
            if (res == 0) has = false; // occluded: the byte keeps its 'pending' mark, which k_shade_finish reads as occluded
            else if (res == 1) { k++; next_sdf(); }
        }
    }
    if (COUNT && evals.n) { atomicAdd(evals_out, (unsigned long long)evals.n); atomicAdd(evals_out + 4, (unsigned long long)evals.it); }
}

// Fast path of k_shadow for scenes with exactly one TracedSDF: uniform SDF parameters, and a
// prefetched NEXT segment per lane (see k_extend1).  (A variant that scans the visibility bytes itself instead of consuming
// k_shadow_list's job list was measured 1-4 % slower - window logic + 8 VGPRs in the hot loop; tools/variants/README.md.)
template <bool COUNT, int SDFK>
__global__ void __launch_bounds__(256) k_shadow1(const DScene* __restrict__ scp, uint32_t ks, Nee nee, DCtl* __restrict__ ctl,
                                                  uint32_t PREFETCH_MIN, unsigned long long* __restrict__ evals_out) {
    const DScene& sc = *scp;
    const uint32_t lane = lane_id();
    const uint32_t n_jobs = ctl->job_count, max_vis = sc.max_vis_marches;
    uint32_t* const head = &ctl->head_shadow;
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->shadow_jobs += n_jobs;
    const DHitable h = sc.h[ks];
    const float c0 = 0.0001f * sc.detail_scale, c1 = 0.00001f * sc.detail_scale;
    uint32_t cur = 0, end = 0;
    bool exhausted = false, endgame = false;
    bool c_has = false, first = false, nan = false, n_has = false;
    uint32_t ref = 0, n_ref = 0, m = 0;
    EvalCtr evals;
    f3 start = f3{0, 0, 0}, dir = f3{0, 0, 0}, n_start = f3{0, 0, 0}, n_dir = f3{0, 0, 0};
    f3 pt = f3{0, 0, 0}; // the current segment's next march point (unused under -DRAYN_MARCH_POINT_SELECT)
    float max_dist = 0.0f, n_max = 0.0f, t = 0.0f;
    float c_scale = h.scale, n_scale = h.scale; // MandelBox scale at the packet time (extension)
    for (;;) {
        const uint64_t lack = __ballot(!n_has);
        const uint64_t idle = __ballot(!c_has && !n_has);
        // endgame: once the queue is nearly drained no more spare rays are hoarded (a spare held by a lane that is still
        // inside a long march would wait while other lanes idle) - idle lanes then fetch one ray at a time
        if (!exhausted && (endgame ? idle != 0 : ((uint32_t)__popcll(lack) >= PREFETCH_MIN || (uint32_t)__popcll(idle) >= 4u))) {
            for (;;) {
                const uint64_t need = endgame ? __ballot(!n_has && !c_has) : __ballot(!n_has);
                if (need == 0) break;
                if (cur == end) {
                    uint32_t base = 0;
                    if (lane == 0) base = atomicAdd(head, CHUNK);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base >= n_jobs) { exhausted = true; break; }
                    cur = base;
                    end = min(base + CHUNK, n_jobs);
                    endgame = n_jobs - base < ENDGAME_ENTRIES;
                }
                const uint32_t rank = mbcnt(need), avail = end - cur;
                if (((need >> lane) & 1ull) && rank < avail) {
                    n_ref = nee.job_ref[cur + rank];
                    const float2 j0 = nee.job_geo[3 * (size_t)n_ref], j1 = nee.job_geo[3 * (size_t)n_ref + 1], j2 = nee.job_geo[3 * (size_t)n_ref + 2];
                    const float jt0 = sc.anim_spheres ? nee.t0[n_ref % (uint32_t)nee.cap] : 0.0f;
                    const f3 origin = sphere_center(h, jt0); // TracedSDF origin at the packet time (extension; zero in the reference)
                    n_scale = sdf_scale(h, jt0);
                    n_start = f3{j0.x, j0.y, j1.x} - origin;
                    const f3 e = f3{j1.y, j2.x, j2.y} - origin;
                    n_dir = e - n_start;
                    n_max = mag(n_dir);
                    n_dir = div_by_mag(n_dir, n_max);
                    n_has = true;
                }
                cur += min((uint32_t)__popcll(need), avail);
            }
        }
        if (!c_has && n_has) {
            c_has = true; n_has = false;
            start = n_start; dir = n_dir; max_dist = n_max; ref = n_ref; c_scale = n_scale;
            first = true;
#ifndef RAYN_MARCH_POINT_SELECT
            pt = n_start; // the first evaluation is at the segment start
#endif
        }
        if (__ballot(c_has) == 0) {
            if (exhausted) break;
            continue;
        }
        if (c_has) { // TracedSDF::occluded, src/sdf.rs:25-57
#ifdef RAYN_MARCH_POINT_SELECT
            const f3 p = first ? start : muladd3(dir, t, start);
#else
            const f3 p = pt; // (see k_extend1)
#endif
            const float dist = sdf_dist<COUNT, SDFK>(h, p, evals, c_scale);
            int res = -1; // -1 keep marching, 0 occluded, 1 visible
            if (first) {
                t = dist; nan = dist != dist; first = false; m = 0;
                if (max_vis == 0) res = ((dist < 0.0001f) && !((dist > max_dist) || nan)) ? 0 : 1;
                else if ((t > max_dist) || nan) res = 1;
            } else {
                if (__builtin_fabsf(dist) < fmaxs(c0, c1 * t)) res = 0;
                else {
                    t = t + dist; m++;
                    if (m == max_vis || (t > max_dist) || nan) res = 1;
                }
            }
            if (res >= 0) { if (res == 1) nee.vis[ref] = 1; c_has = false; } // only VISIBLE results are written (see Nee::vis)
#ifndef RAYN_MARCH_POINT_SELECT
            else pt = muladd3(dir, t, start);
#endif
        }
    }
    if (COUNT && evals.n) { atomicAdd(evals_out, (unsigned long long)evals.n); atomicAdd(evals_out + 4, (unsigned long long)evals.it); }
}

#include "march_bulb.h"

__global__ void __launch_bounds__(256) k_shade_finish(const DScene* __restrict__ scp, const uint32_t* __restrict__ bq, const DCtl* __restrict__ ctl,
                                                       Pool pool, Nee nee) {
    const DScene& sc = *scp;
    const uint32_t n_slots = ctl->b_groups << 6;
    for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < n_slots; j += gridDim.x * blockDim.x) {
    const uint32_t P = bq[j];
    if (P == INVALID) continue;
    const uint32_t flags = nee.flags[j], nl = sc.n_lights, VM = sc.vm;
    const size_t cap = nee.cap;
    if (flags & 6u) {
        const float4 c0 = pool.col0[P];
        const float2 c1 = *(const float2*)(&pool.col1[P].x);
        f3 rad = f3{c0.x, c0.y, c0.z};
        const f3 thr = f3{c0.w, c1.x, c1.y};
        if (flags & 2u) { // src/integrator.rs:82-93
            const float corr = (float)nl / 4.0f, vol_T = nee.T[j];
            for (uint32_t i = 0; i < 4; i++) {
                const f3 x = f3{nee.x[(i * 3 + 0) * cap + j], nee.x[(i * 3 + 1) * cap + j], nee.x[(i * 3 + 2) * cap + j]};
                const float occ = nee.vis[i * cap + j] == 1 ? 1.0f : 0.0f;
                const f3 li = x * occ / nee.pdf[i * cap + j];
                rad = rad + li * thr * corr * vol_T;
            }
        }
        if (flags & 4u) { // src/integrator.rs:99-131
            const float corr = (float)nl / 4.0f / (float)VM;
            const unsigned long long vpicks = nee.vpicks[j];
            for (uint32_t s = 4; s < 4 + 4 * VM; s++) {
                // volume_sample_one_light's li * f * transmission (src/integrator.rs:269-279): f = 1/(4 pi) is a constant and
                // k_shade_setup kept the transmittance, so x is rebuilt here with the same two multiplies
                const float f = 1.0f / (4.0f * PI_F);
                const f3 x = sc.l[(uint32_t)(vpicks >> (4 * (s - 4))) & 15u].emission * f * nee.vtr[(s - 4) * cap + j];
                const float occ = nee.vis[s * cap + j] == 1 ? 1.0f : 0.0f;
                const f3 li = x * occ / nee.pdf[s * cap + j];
                rad = rad + li * thr * corr * sc.rho_s * nee.aux[(s - 4) * cap + j];
            }
        }
        if (flags & 1u) { // survivor: publish the new throughput together with the radiance
            pool.col0[P] = make_float4(rad.x, rad.y, rad.z, nee.nthr[j]);
            *(float2*)(&pool.col1[P].x) = make_float2(nee.nthr[cap + j], nee.nthr[2 * cap + j]);
        } else pool.col0[P] = make_float4(rad.x, rad.y, rad.z, c0.w);
    } else if (flags & 1u) {
        pool.col0[P].w = nee.nthr[j];
        *(float2*)(&pool.col1[P].x) = make_float2(nee.nthr[cap + j], nee.nthr[2 * cap + j]);
    }
    } // grid-stride loop
}

// ------------------------------------------------------------------------------------------------
// a25/a27: Tile::add_sample + tile_finished (src/film.rs:54-61,82-98,167-172,660-691).
// The reference adds a pixel's samples serially in emission order = (depth, packet slot); float
// addition is not associative, so the resolve kernels replay exactly that order: the pixel's spp paths are sorted
// by termination key (bitonic network in registers + wave shuffles, below) and three lanes (r,g,b) accumulate them
// sequentially.  Alpha/WorldNormal are depth-0 samples, ordered (object, sample).
// ------------------------------------------------------------------------------------------------
// where a tile's local pixel (x-major inside the tile, like the path pool) lives in the output film
RD size_t film_pixel(const DTile& t, uint32_t lpix, uint32_t width) {
    if (t.film_packed) return (size_t)t.film_base + lpix;
    const uint32_t lx = lpix / t.eh, ly = lpix % t.eh;
    return (size_t)(t.x0 + lx) + (size_t)(t.y0 + ly) * width;
}

// serial float sum of src[0], src[stride], .. (cnt terms) in index order (one lane = one channel); loads are issued eight at a time
RD float serial_sum(const float* src, uint32_t stride, uint32_t cnt) {
    float a = 0.0f;
    uint32_t e = 0;
    for (; e + 8 <= cnt; e += 8) {
        float v[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) v[u] = src[(e + u) * stride];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) a += v[u];
    }
    for (; e < cnt; e++) a += src[e * stride];
    return a;
}

// ------------------------------------------------------------------------------------------------
// k_resolve_reg (spp <= 1024): one wave per pixel holds KPL = n_sort / 64 sort keys per lane, REGISTER-resident.  Bitonic steps whose partner lies in the same lane are compare-exchanges
// between registers; the others exchange register r with lane ^ (jj / KPL) through a wave shuffle - no LDS round trip and no
// barrier per step (the LDS version spent 55 x 8 dependent LDS round trips per sort at 1024 spp: 0.9 TB/s, 11 % of the HBM roof).
// LDS only stages the sorted samples for the three serial-sum lanes.
// ------------------------------------------------------------------------------------------------
RD unsigned long long shfl_xor_key(unsigned long long v, uint32_t m) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, (int)m), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), (int)m);
    return ((unsigned long long)hi << 32) | lo;
}
RD uint32_t shfl_xor_key(uint32_t v, uint32_t m) { return (uint32_t)__shfl_xor((int)v, (int)m); }

// ascending bitonic sort of 64 * KPL keys, element e = lane * KPL + r.  The stage (k) and the cross-lane sub-step (jj) loops
// stay ROLLED (a fully unrolled u64 network at KPL = 16 was 70 KB of code and ran out of the instruction cache); only the
// per-register work is unrolled, and selects are written so that no branch separates a shuffle from its use.
template <typename K, uint32_t KPL>
RD void bitonic_sort_reg(K (&key)[KPL]) {
    const uint32_t lane = lane_id();
    constexpr uint32_t N = 64 * KPL;
#pragma nounroll
    for (uint32_t k = 2; k <= N; k <<= 1) {
        const bool asc_lane = ((lane * KPL) & k) == 0; // direction of this lane's elements when k >= KPL
#pragma nounroll
        for (uint32_t jj = k >> 1; jj >= KPL && jj > 0; jj >>= 1) { // partner = the same register of lane ^ (jj / KPL)
            const uint32_t lm = jj / KPL;
            const bool keep_min = asc_lane == ((lane & lm) == 0);
            K o[KPL];
#pragma unroll
            for (uint32_t r = 0; r < KPL; r++) o[r] = shfl_xor_key(key[r], lm);
#pragma unroll
            for (uint32_t r = 0; r < KPL; r++) {
                const bool take = (o[r] < key[r]) == keep_min; // equal keys: either copy is the same value
                key[r] = take ? o[r] : key[r];
            }
        }
#pragma unroll
        for (uint32_t jj = KPL / 2; jj > 0; jj >>= 1) { // partner in the same lane: registers r and r | jj
            if (jj < k) {                                 // wave-uniform: this sub-step belongs to stage k
#pragma unroll
                for (uint32_t r = 0; r < KPL; r++) {
                    if ((r & jj) == 0) {
                        const bool asc = k < KPL ? (r & k) == 0 : asc_lane;
                        const K a = key[r], b = key[r | jj];
                        const bool sw = (a > b) == asc;
                        key[r] = sw ? b : a;
                        key[r | jj] = sw ? a : b;
                    }
                }
            }
        }
    }
}

// keys loaded sample-major (register r of lane l = sample r * 64 + l): true when they are already in non-decreasing sample order
RD unsigned long long shfl_key(unsigned long long v, uint32_t src) {
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, (int)src), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), (int)src);
    return ((unsigned long long)hi << 32) | lo;
}
RD uint32_t shfl_key(uint32_t v, uint32_t src) { return (uint32_t)__shfl((int)v, (int)src); }
template <typename K, uint32_t KPL>
RD bool is_sorted_reg(const K (&key)[KPL]) {
    const uint32_t lane = lane_id();
    bool ok = true;
#pragma unroll
    for (uint32_t r = 0; r < KPL; r++) {
        // successor of sample r * 64 + lane: lane + 1 of the same register, or lane 0 of the next register (none after the last)
        const K same = shfl_key(key[r], (lane + 1) & 63u);
        const K wrap = r + 1 < KPL ? shfl_key(key[r + 1 < KPL ? r + 1 : r], 0u) : (K)~(K)0;
        ok = ok && !(key[r] > (lane == 63 ? wrap : same));
    }
    return __ballot(!ok) == 0;
}

// serial float sum of src[lo .. hi) in index order; 128-bit LDS reads once the index is 4-aligned (one LDS instruction per four terms:
// the serial-sum phase is ISSUE-bound - a handful of lanes, one instruction per 4 cycles - so instructions per term are what it costs)
RD float serial_sum_range(const float* src, uint32_t lo, uint32_t hi) {
    float a = 0.0f;
    uint32_t e = lo;
    for (; e < hi && (e & 3u); e++) a += src[e];
    for (; e + 8 <= hi; e += 8) {
        const float4 v0 = *(const float4*)(src + e), v1 = *(const float4*)(src + e + 4);
        a += v0.x; a += v0.y; a += v0.z; a += v0.w;
        a += v1.x; a += v1.y; a += v1.z; a += v1.w;
    }
    for (; e < hi; e++) a += src[e];
    return a;
}

template <uint32_t KPL>
__global__ void __launch_bounds__(64) k_resolve_reg(const DScene* __restrict__ scp, const DTile* __restrict__ tiles, Pool pool,
                                                     float* __restrict__ out_color, float* __restrict__ out_alpha,
                                                     float* __restrict__ out_background, float* __restrict__ out_normal) {
    constexpr uint32_t n_sort = 64 * KPL;
    // r4: the sorted samples are staged CHANNEL-major (R | G | B, then normal x | y | z) + one flag byte, so that a serial-sum lane reads four
    // terms with one 128-bit LDS instruction (see serial_sum_range and k_resolve_blk: this phase is bound by instruction issue)
    __shared__ __attribute__((aligned(16))) float ch[3][n_sort];
    __shared__ uint8_t s_flag[n_sort];
    constexpr unsigned long long NOKEY = ~0ull;
    const DScene& sc = *scp;
    const DTile tile = tiles[blockIdx.y];
    const uint32_t lpix = blockIdx.x;
    if (lpix >= tile.ew * tile.eh) return;
    const uint32_t spp = sc.spp, lane = threadIdx.x;
    const uint32_t P0 = tile.pool_base + lpix * spp;
    const size_t fi = film_pixel(tile, lpix, sc.width);
    const float n = (float)spp;
    // ---- Color / Background in (depth, slot) order: key = depth:7 | slot:32 | background:1 | sample:12
    unsigned long long key[KPL];
#pragma unroll
    for (uint32_t r = 0; r < KPL; r++) {
        const uint32_t i = r * 64 + lane;
        unsigned long long k = NOKEY;
        if (i < spp) {
            const uint32_t info = pool.term_info[P0 + i];
            if (info != TERM_NONE)
                k = ((unsigned long long)(info & 0x7Fu) << 45) | ((unsigned long long)pool.term_key[P0 + i] << 13) | ((info >> 7) << 12) | i;
        }
        key[r] = k;
    }
    const bool sorted = is_sorted_reg<unsigned long long, KPL>(key); // sky-only pixels arrive sorted (sample-major layout)
    if (!sorted) bitonic_sort_reg<unsigned long long, KPL>(key);   // now element lane * KPL + r
    uint32_t cnt = 0, nb = 0;
#pragma unroll
    for (uint32_t r = 0; r < KPL; r++) {
        cnt += (uint32_t)__popcll(__ballot(key[r] != NOKEY));
        nb += (uint32_t)__popcll(__ballot(key[r] != NOKEY && (((uint32_t)key[r] >> 12) & 1u)));
    }
    // Background samples are emitted at depth 0 only, i.e. they sort first: when they are exactly the first nb entries (always, unless
    // max_bounces == 0 puts Color samples at depth 0 too) the two accumulators are plain serial sums over [0, nb) and [nb, cnt)
    bool prefix_ok = true;
#pragma unroll
    for (uint32_t r = 0; r < KPL; r++) {
        if (key[r] != NOKEY) {
            const uint32_t e = sorted ? r * 64 + lane : lane * KPL + r, lo = (uint32_t)key[r];
            const float4 c = pool.col0[P0 + (lo & 0xFFFu)];
            const uint32_t bg = (lo >> 12) & 1u;
            ch[0][e] = c.x; ch[1][e] = c.y; ch[2][e] = c.z;
            s_flag[e] = (uint8_t)bg;
            prefix_ok = prefix_ok && (bg != 0) == (e < nb);
        }
    }
    const bool split = __ballot(!prefix_ok) == 0;
    __syncthreads();
    if (split) {
        if (lane < 6) { // lanes 0..2: Color r, g, b over [nb, cnt); lanes 3..5: Background over [0, nb)
            const uint32_t c = lane % 3u, isb = lane / 3u;
            const float v = serial_sum_range(ch[c], isb ? 0u : nb, isb ? nb : cnt) / n;
            if (isb) out_background[3 * fi + c] = v; else out_color[3 * fi + c] = v;
        }
    } else if (lane < 3) { // general order: two independent serial chains with a flag test per term (the accumulator a sample does not
        float c = 0.0f, b = 0.0f; // belong to is simply not touched)
        for (uint32_t e = 0; e < cnt; e++) {
            const float v = ch[lane][e];
            if (s_flag[e]) b += v; else c += v;
        }
        out_color[3 * fi + lane] = c / n;
        out_background[3 * fi + lane] = b / n;
    }
    __syncthreads();
    // ---- Alpha / WorldNormal: depth-0 packets are object-major, then queue (= sample) order: key = object:16 | sample:16
    uint32_t k32[KPL];
#pragma unroll
    for (uint32_t r = 0; r < KPL; r++) {
        const uint32_t i = r * 64 + lane;
        uint32_t k = INVALID;
        if (i < spp) { const uint32_t ob = __float_as_uint(pool.aov[P0 + i].w); if (ob != OBJ_NONE) k = (ob << 16) | i; }
        k32[r] = k;
    }
    const bool sorted0 = is_sorted_reg<uint32_t, KPL>(k32); // the common case: one object per pixel
    if (!sorted0) bitonic_sort_reg<uint32_t, KPL>(k32);
    uint32_t cnt0 = 0;
#pragma unroll
    for (uint32_t r = 0; r < KPL; r++) cnt0 += (uint32_t)__popcll(__ballot(k32[r] != INVALID));
#pragma unroll
    for (uint32_t r = 0; r < KPL; r++)
        if (k32[r] != INVALID) {
            const uint32_t e = sorted0 ? r * 64 + lane : lane * KPL + r;
            const float4 a = pool.aov[P0 + (k32[r] & 0xFFFFu)];
            ch[0][e] = a.x; ch[1][e] = a.y; ch[2][e] = a.z;
        }
    __syncthreads();
    // Alpha adds 1.0 per depth-0 surface sample: every partial sum is an integer < 2^24, so the serial sum is the count
    if (lane == 3) out_alpha[fi] = (float)cnt0 / n;
    else if (lane < 3) out_normal[3 * fi + lane] = serial_sum_range(ch[lane], 0u, cnt0) / n;
}

// ------------------------------------------------------------------------------------------------
// k_resolve_blk for 512 < spp <= 4096 (configs 3 / 4 / 5): the register-resident sort across the 2 / 4 / 8 waves of a block, EIGHT
// keys per lane, element e = tid * 8 + r.  Sub-steps inside a lane are register compare-exchanges, sub-steps inside a wave are
// shuffles, and only the sub-steps whose partner sits in another wave (jj >= 512: 1 / 3 / 6 of the 55 / 66 / 78) go through LDS.
// Eight keys per lane instead of r2's sixteen (one wave per pixel at 1024 spp, four at 4096): the sort is a chain of dependent
// shuffles, i.e. latency-bound, and 16 u64 keys + 16 exchange registers cost 118-131 VGPRs - with 16 B of LDS per sample that
// left 2.5 (1024 spp) / 2 (4096 spp) waves per SIMD.  Half the keys per lane = ~75 VGPRs, half the chain length per lane and
// twice the waves per pixel: 5 / 4 waves per SIMD on the same LDS.
// ------------------------------------------------------------------------------------------------
template <typename K, uint32_t KPL, uint32_t NT = 256>
RD void bitonic_sort_block4(K (&key)[KPL], K* exch /* [NT * KPL] */) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    constexpr uint32_t N = NT * KPL;
#pragma nounroll
    for (uint32_t k = 2; k <= N; k <<= 1) {
        const bool asc_t = ((tid * KPL) & k) == 0; // direction of this thread's elements when k >= KPL
#pragma nounroll
        for (uint32_t jj = k >> 1; jj >= KPL && jj > 0; jj >>= 1) {
            const uint32_t tm = jj / KPL; // partner thread = tid ^ tm, same register
            const bool keep_min = asc_t == ((tid & tm) == 0);
            K o[KPL];
            if (tm >= 64) { // partner in another wave: through LDS
#pragma unroll
                for (uint32_t r = 0; r < KPL; r++) exch[r * NT + tid] = key[r];
                __syncthreads();
#pragma unroll
                for (uint32_t r = 0; r < KPL; r++) o[r] = exch[r * NT + (tid ^ tm)];
                __syncthreads();
            } else {
#pragma unroll
                for (uint32_t r = 0; r < KPL; r++) o[r] = shfl_xor_key(key[r], tm);
            }
#pragma unroll
            for (uint32_t r = 0; r < KPL; r++) {
                const bool take = (o[r] < key[r]) == keep_min;
                key[r] = take ? o[r] : key[r];
            }
        }
#pragma unroll
        for (uint32_t jj = KPL / 2; jj > 0; jj >>= 1) {
            if (jj < k) {
#pragma unroll
                for (uint32_t r = 0; r < KPL; r++) {
                    if ((r & jj) == 0) {
                        const bool asc = k < KPL ? (r & k) == 0 : asc_t;
                        const K a = key[r], b = key[r | jj];
                        const bool sw = (a > b) == asc;
                        key[r] = sw ? b : a;
                        key[r | jj] = sw ? a : b;
                    }
                }
            }
        }
    }
    (void)lane;
}

template <uint32_t NT, uint32_t KPL>
__global__ void __launch_bounds__(NT) k_resolve_blk(const DScene* __restrict__ scp, const DTile* __restrict__ tiles, Pool pool,
                                                     float* __restrict__ out_color, float* __restrict__ out_alpha,
                                                     float* __restrict__ out_background, float* __restrict__ out_normal,
                                                     const uint32_t* __restrict__ base_hist, uint32_t hist_stride) {
    constexpr uint32_t n_sort = NT * KPL; // 1024 (2 waves x 8 keys), 2048 (4 x 8) or 4096 (8 x 8)
    // 13 B per sample: the sorted samples channel-major (R | G | B, then normal x | y | z) for the serial-sum lanes + one flag byte; the
    // sort's cross-wave exchange, the sorted key array of the rank search and the sortedness check use the R plane before the staging starts
    __shared__ __attribute__((aligned(16))) float ch[3][n_sort];
    __shared__ uint8_t s_flag[n_sort];
    __shared__ uint32_t s_meta[4]; // [0] valid keys, [1] "some pair is out of order", [2] Background samples, [3] "the Background samples are not a prefix of the order"
    uint32_t* exch32 = (uint32_t*)ch[0];
    const DScene& sc = *scp;
    const DTile tile = tiles[blockIdx.y];
    const uint32_t lpix = blockIdx.x;
    if (lpix >= tile.ew * tile.eh) return;
    const uint32_t spp = sc.spp, tid = threadIdx.x;
    const uint32_t P0 = tile.pool_base + lpix * spp;
    const size_t fi = film_pixel(tile, lpix, sc.width);
    const float n = (float)spp;
    // ---- Color / Background in (depth, slot) order.  r4: the sort runs on 32-bit keys WITHOUT a payload - depth:7 | slot relative to
    // where the tile's binned segment began at that depth (k_tile_prefix keeps those bases per depth; a tile's segment is at most
    // 1024 pixels x 4096 spp + padding = 2^22 + 112 slots, the offset field has RESOLVE_KEY_SHIFT = 25 bits, checked at compile time against the host's limits and per frame: resolve_keys_fit, kernels.h) - and every sample then finds its rank in the sorted key array by binary search (keys of a pixel are
    // distinct: a slot holds one path).  Half the registers, one shuffle + v_cmp + v_cndmask per compare-exchange instead of two shuffles,
    // a 64-bit compare and two selects.
    constexpr uint32_t NOKEY = 0xFFFFFFFFu; // keys compare UNSIGNED; depth <= 120 (validate()) puts every real key below 121 << 25 = 0xF2000000 < NOKEY (depths >= 64 do set bit 31)
    uint32_t ko[KPL];
    uint32_t mine = 0, bgm = 0;
#pragma unroll
    for (uint32_t r = 0; r < KPL; r++) { // sample-major: register r of thread t = sample r * NT + t (coalesced)
        const uint32_t i = r * NT + tid;
        uint32_t k = NOKEY;
        if (i < spp) {
            const uint32_t info = pool.term_info[P0 + i];
            if (info != TERM_NONE) {
                const uint32_t d = info & 0x7Fu;
                k = (d << RESOLVE_KEY_SHIFT) | (pool.term_key[P0 + i] - base_hist[d * hist_stride + blockIdx.y]);
                bgm |= (info >> 7) << r;
            }
        }
        ko[r] = k;
        mine += k != NOKEY;
        exch32[i] = k;
    }
    if (tid < 4) s_meta[tid] = 0;
    __syncthreads();
    bool ok = true;
#pragma unroll
    for (uint32_t r = 0; r < KPL; r++) { const uint32_t i = r * NT + tid; ok = ok && (i + 1 >= n_sort || !(ko[r] > exch32[i + 1])); }
    if (mine) atomicAdd(&s_meta[0], mine);
    if (bgm) atomicAdd(&s_meta[2], (uint32_t)__popc(bgm));
    if (!ok) s_meta[1] = 1;
    __syncthreads(); // also orders the LDS reads above before the exchange buffer is reused
    const uint32_t cnt = s_meta[0], nb = s_meta[2];
    const bool sorted = s_meta[1] == 0;
    uint32_t rank[KPL];
#pragma unroll
    for (uint32_t r = 0; r < KPL; r++) rank[r] = r * NT + tid; // already in order (sky-only pixels): a sample's rank is its index
    if (!sorted) {
        uint32_t ks[KPL];
#pragma unroll
        for (uint32_t r = 0; r < KPL; r++) ks[r] = ko[r];
        bitonic_sort_block4<uint32_t, KPL, NT>(ks, exch32); // now element tid * KPL + r
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < KPL; r++) exch32[tid * KPL + r] = ks[r];
        __syncthreads();
#pragma unroll
        for (uint32_t r = 0; r < KPL; r++) { // rank = number of sorted keys below this sample's key (branch-free lower bound, n_sort = 2^m)
            uint32_t pos = 0;
#pragma unroll
            for (uint32_t step = n_sort / 2; step > 0; step >>= 1) pos += exch32[pos + step - 1] < ko[r] ? step : 0u;
            rank[r] = pos;
        }
    }
    __syncthreads(); // every rank is known before the staging overwrites the key array
    // Background samples are emitted at depth 0 only (src/integrator.rs:151-160), i.e. they sort first: when they are exactly the first nb
    // entries of the order (always, unless max_bounces == 0 puts Color samples at depth 0 too) the two accumulators are two plain serial
    // sums over [0, nb) and [nb, cnt) - no per-term flag test in the issue-bound loop
    bool prefix_ok = true;
#pragma unroll
    for (uint32_t r = 0; r < KPL; r++) {
        if (ko[r] != NOKEY) {
            const float4 c = pool.col0[P0 + r * NT + tid];
            const uint32_t bg = (bgm >> r) & 1u;
            ch[0][rank[r]] = c.x; ch[1][rank[r]] = c.y; ch[2][rank[r]] = c.z;
            s_flag[rank[r]] = (uint8_t)bg;
            prefix_ok = prefix_ok && (bg != 0) == (rank[r] < nb);
        }
    }
    if (!prefix_ok) s_meta[3] = 1;
    __syncthreads();
    if (s_meta[3] == 0) {
        if (tid < 6) { // lanes 0..2: Color r, g, b over [nb, cnt); lanes 3..5: Background over [0, nb)
            const uint32_t c = tid % 3u, isb = tid / 3u;
            const float v = serial_sum_range(ch[c], isb ? 0u : nb, isb ? nb : cnt) / n;
            if (isb) out_background[3 * fi + c] = v; else out_color[3 * fi + c] = v;
        }
    } else if (tid < 3) { // general order: two independent chains with a flag test per term (see k_resolve_reg)
        float c = 0.0f, b = 0.0f;
        for (uint32_t e = 0; e < cnt; e++) {
            const float v = ch[tid][e];
            if (s_flag[e]) b += v; else c += v;
        }
        out_color[3 * fi + tid] = c / n;
        out_background[3 * fi + tid] = b / n;
    }
    __syncthreads();
    // ---- Alpha / WorldNormal: key = object:16 | sample:16
    uint32_t k32[KPL];
    mine = 0;
#pragma unroll
    for (uint32_t r = 0; r < KPL; r++) {
        const uint32_t i = r * NT + tid;
        uint32_t k = INVALID;
        if (i < spp) { const uint32_t ob = __float_as_uint(pool.aov[P0 + i].w); if (ob != OBJ_NONE) k = (ob << 16) | i; }
        k32[r] = k;
        mine += k != INVALID;
        exch32[i] = k;
    }
    if (tid < 2) s_meta[tid] = 0;
    __syncthreads();
    ok = true;
#pragma unroll
    for (uint32_t r = 0; r < KPL; r++) { const uint32_t i = r * NT + tid; ok = ok && (i + 1 >= n_sort || !(k32[r] > exch32[i + 1])); }
    if (mine) atomicAdd(&s_meta[0], mine);
    if (!ok) s_meta[1] = 1;
    __syncthreads();
    const uint32_t cnt0 = s_meta[0];
    const bool sorted0 = s_meta[1] == 0;
    if (!sorted0) bitonic_sort_block4<uint32_t, KPL, NT>(k32, exch32);
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < KPL; r++)
        if (k32[r] != INVALID) {
            const uint32_t e = sorted0 ? r * NT + tid : tid * KPL + r;
            const float4 a = pool.aov[P0 + (k32[r] & 0xFFFFu)];
            ch[0][e] = a.x; ch[1][e] = a.y; ch[2][e] = a.z;
        }
    __syncthreads();
    if (tid == 3) out_alpha[fi] = (float)cnt0 / n;
    else if (tid < 3) out_normal[3 * fi + tid] = serial_sum_range(ch[tid], 0u, cnt0) / n;
}

// ------------------------------------------------------------------------------------------------
// k_resolve for 4096 < spp <= 16384 (beyond BASELINE's configurations; the reference has no limit): the same sort across the
// SIXTEEN waves of a 1024-thread block, 16 keys per lane.  128 KB of dynamic LDS serve in turn as the sort's cross-wave exchange
// buffer and as the staging area of the serial sums, which run in two rounds because (r, g) and (b, flag) of 16384 samples do
// not fit together: round one stages (r, g) for lanes 0 / 1, round two b for lane 2; the background flags of the sorted order sit
// in a 2 KB bit table.  Sample index: 14 bits of the sort key.
// ------------------------------------------------------------------------------------------------
RD uint32_t flag_bit(const unsigned long long* bits, uint32_t e) { return (uint32_t)(bits[e >> 6] >> (e & 63u)) & 1u; }
// serial sums of one channel in index order, colour and background kept apart by the flag table (two independent chains, see k_resolve_reg)
RD void serial_sum_split(const float* src, uint32_t stride, const unsigned long long* bits, uint32_t cnt, float& c_out, float& b_out) {
    float c = 0.0f, b = 0.0f;
    uint32_t e = 0;
    for (; e + 8 <= cnt; e += 8) {
        float v[8];
        uint32_t f[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) { v[u] = src[(e + u) * stride]; f[u] = flag_bit(bits, e + u); }
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) {
            const bool bg = f[u] != 0;
            c += bg ? 0.0f : v[u];
            b += bg ? v[u] : 0.0f;
        }
    }
    for (; e < cnt; e++) {
        if (flag_bit(bits, e)) b += src[e * stride]; else c += src[e * stride];
    }
    c_out = c; b_out = b;
}

__global__ void __launch_bounds__(1024) k_resolve_huge(const DScene* __restrict__ scp, const DTile* __restrict__ tiles, Pool pool,
                                                        float* __restrict__ out_color, float* __restrict__ out_alpha,
                                                        float* __restrict__ out_background, float* __restrict__ out_normal) {
    constexpr uint32_t NT = 1024, KPL = 16, n_sort = NT * KPL; // 16384
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_dyn[]; // n_sort * 8 bytes
    __shared__ unsigned long long s_flags[n_sort / 64];
    __shared__ uint32_t s_cnt;
    unsigned long long* exch64 = (unsigned long long*)lds_dyn;
    uint32_t* exch32 = (uint32_t*)lds_dyn;
    float2* rg = (float2*)lds_dyn;
    float* one = (float*)lds_dyn;
    constexpr unsigned long long NOKEY = ~0ull;
    const DScene& sc = *scp;
    const DTile tile = tiles[blockIdx.y];
    const uint32_t lpix = blockIdx.x;
    if (lpix >= tile.ew * tile.eh) return;
    const uint32_t spp = sc.spp, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t P0 = tile.pool_base + lpix * spp;
    const size_t fi = film_pixel(tile, lpix, sc.width);
    const float n = (float)spp;
    // ---- Color / Background in (depth, slot) order
    unsigned long long key[KPL];
    uint32_t mine = 0;
#pragma unroll
    for (uint32_t r = 0; r < KPL; r++) { // sample-major: register r of thread t = sample r * NT + t (coalesced)
        const uint32_t i = r * NT + tid;
        unsigned long long k = NOKEY;
        if (i < spp) {
            const uint32_t info = pool.term_info[P0 + i];
            if (info != TERM_NONE)
                k = ((unsigned long long)(info & 0x7Fu) << 47) | ((unsigned long long)pool.term_key[P0 + i] << 15) | ((info >> 7) << 14) | i;
        }
        key[r] = k;
        mine += k != NOKEY;
        exch64[i] = k;
    }
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    bool ok = true;
#pragma unroll
    for (uint32_t r = 0; r < KPL; r++) { const uint32_t i = r * NT + tid; ok = ok && (i + 1 >= n_sort || !(key[r] > exch64[i + 1])); }
    if (mine) atomicAdd(&s_cnt, mine);
    const bool sorted = __syncthreads_or(!ok) == 0; // also orders the LDS reads above before the exchange buffer is reused
    const uint32_t cnt = s_cnt;
    if (!sorted) bitonic_sort_block4<unsigned long long, KPL, NT>(key, exch64); // now element tid * KPL + r
    __syncthreads();
    // background flags of the sorted order (NOKEY elements sort last and are never read)
    if (sorted) {
#pragma unroll
        for (uint32_t r = 0; r < KPL; r++) { // element r * NT + tid: the 64 lanes of a wave fill one word
            const unsigned long long m = __ballot(key[r] != NOKEY && (((uint32_t)key[r] >> 14) & 1u));
            if (lane == 0) s_flags[r * (NT / 64) + wave] = m;
        }
    } else {
        uint32_t m16 = 0;
#pragma unroll
        for (uint32_t r = 0; r < KPL; r++) m16 |= (key[r] != NOKEY && (((uint32_t)key[r] >> 14) & 1u)) ? (1u << r) : 0u;
        ((unsigned short*)s_flags)[tid] = (unsigned short)m16; // elements tid * 16 .. + 15
    }
    // round one: (r, g)
#pragma unroll
    for (uint32_t r = 0; r < KPL; r++)
        if (key[r] != NOKEY) {
            const uint32_t e = sorted ? r * NT + tid : tid * KPL + r;
            const float4 c = pool.col0[P0 + ((uint32_t)key[r] & 0x3FFFu)];
            rg[e] = make_float2(c.x, c.y);
        }
    __syncthreads();
    if (tid < 2) {
        float c, b;
        serial_sum_split((const float*)rg + tid, 2, s_flags, cnt, c, b);
        out_color[3 * fi + tid] = c / n;
        out_background[3 * fi + tid] = b / n;
    }
    __syncthreads();
    // round two: b
#pragma unroll
    for (uint32_t r = 0; r < KPL; r++)
        if (key[r] != NOKEY) {
            const uint32_t e = sorted ? r * NT + tid : tid * KPL + r;
            one[e] = pool.col0[P0 + ((uint32_t)key[r] & 0x3FFFu)].z;
        }
    __syncthreads();
    if (tid == 0) {
        float c, b;
        serial_sum_split(one, 1, s_flags, cnt, c, b);
        out_color[3 * fi + 2] = c / n;
        out_background[3 * fi + 2] = b / n;
    }
    __syncthreads();
    // ---- Alpha / WorldNormal: key = object:16 | sample:16
    uint32_t k32[KPL];
    mine = 0;
#pragma unroll
    for (uint32_t r = 0; r < KPL; r++) {
        const uint32_t i = r * NT + tid;
        uint32_t k = INVALID;
        if (i < spp) { const uint32_t ob = __float_as_uint(pool.aov[P0 + i].w); if (ob != OBJ_NONE) k = (ob << 16) | i; }
        k32[r] = k;
        mine += k != INVALID;
        exch32[i] = k;
    }
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    ok = true;
#pragma unroll
    for (uint32_t r = 0; r < KPL; r++) { const uint32_t i = r * NT + tid; ok = ok && (i + 1 >= n_sort || !(k32[r] > exch32[i + 1])); }
    if (mine) atomicAdd(&s_cnt, mine);
    const bool sorted0 = __syncthreads_or(!ok) == 0;
    const uint32_t cnt0 = s_cnt;
    if (!sorted0) bitonic_sort_block4<uint32_t, KPL, NT>(k32, exch32);
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < KPL; r++)
        if (k32[r] != INVALID) {
            const uint32_t e = sorted0 ? r * NT + tid : tid * KPL + r;
            const float4 a = pool.aov[P0 + (k32[r] & 0xFFFFu)];
            rg[e] = make_float2(a.x, a.y);
        }
    __syncthreads();
    if (tid == 3) out_alpha[fi] = (float)cnt0 / n;
    else if (tid < 2) out_normal[3 * fi + tid] = serial_sum((const float*)rg + tid, 2, cnt0) / n;
    __syncthreads();
#pragma unroll
    for (uint32_t r = 0; r < KPL; r++)
        if (k32[r] != INVALID) {
            const uint32_t e = sorted0 ? r * NT + tid : tid * KPL + r;
            one[e] = pool.aov[P0 + (k32[r] & 0xFFFFu)].z;
        }
    __syncthreads();
    if (tid == 0) out_normal[3 * fi + 2] = serial_sum(one, 1, cnt0) / n;
}

// ------------------------------------------------------------------------------------------------
// Multi-device film assembly (no reference counterpart; tiles are independent, src/film.rs:439-627): a peer device resolves
// the tiles it rendered straight into a PLANAR packed film (DTile::film_packed: Color 3N | Alpha N | Background 3N |
// WorldNormal 3N floats for its N owned pixels, tile after tile, pixel-major inside a tile like the path pool), that buffer
// crosses xGMI with ONE peer copy, and device 0 scatters it into the caller's film with this kernel.  film_base of the DTile
// = the tile's first pixel in the packed planes.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_unpack_tiles(const DTile* __restrict__ tiles, uint32_t width, float* __restrict__ color,
                                                       float* __restrict__ alpha, float* __restrict__ background, float* __restrict__ normal,
                                                       const float* __restrict__ p_color, const float* __restrict__ p_alpha,
                                                       const float* __restrict__ p_background, const float* __restrict__ p_normal) {
    const DTile t = tiles[blockIdx.x];
    const uint32_t npx = t.ew * t.eh;
    for (uint32_t i = threadIdx.x; i < npx * 10u; i += 256) {
        const uint32_t lpix = i / 10u, c = i - lpix * 10u;
        const uint32_t lx = lpix / t.eh, ly = lpix - lx * t.eh;
        const size_t fi = (size_t)(t.x0 + lx) + (size_t)(t.y0 + ly) * width, pi = (size_t)t.film_base + lpix;
        if (c < 3) color[3 * fi + c] = p_color[3 * pi + c];
        else if (c == 3) alpha[fi] = p_alpha[pi];
        else if (c < 7) background[3 * fi + (c - 4)] = p_background[3 * pi + (c - 4)];
        else normal[3 * fi + (c - 7)] = p_normal[3 * pi + (c - 7)];
    }
}

// ------------------------------------------------------------------------------------------------
// test probes (called through the C ABI by tests only): per-lane primitives on arbitrary inputs.  (Closest hit and occlusion have no
// probe kernel: rayn_hip_probe_extend / rayn_hip_probe_shadow launch the PRODUCT march kernels on a synthetic queue, rayn_hip.hip.)
// ------------------------------------------------------------------------------------------------
__global__ void k_probe_dist(const DScene* __restrict__ scp, uint32_t hit_index, const float* __restrict__ pts, float* __restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    EvalCtr ev;
    out[i] = sdf_dist<false>(scp->h[hit_index], f3{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]}, ev, scp->h[hit_index].scale);
}
__global__ void k_probe_detmath(uint32_t op, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r;
    switch (op) {
    case 0: r = dmf_expf(a[i]); break;
    case 1: { float c_; dmf_sincosf(a[i], &r, &c_); }; break;
    case 2: { float s_; dmf_sincosf(a[i], &s_, &r); }; break;
    case 3: r = dmf_tanf(a[i]); break;
    case 4: r = dmf_atan2f(a[i], b[i]); break;
    case 5: r = dmf_powf(a[i], b[i]); break;
    case 6: r = div_nr(a[i], b[i]); break;
    case 7: r = a[i] / b[i]; break;
    case 8: r = sqrt_rn(a[i]); break;
    case 9: case 10: case 11: case 12: { // a holds n xyz triples
        const f3 v = f3{a[3 * i], a[3 * i + 1], a[3 * i + 2]};
        const float m = mag(v);
        const f3 q = div_by_mag(v, m);
        r = op == 9 ? q.x : (op == 10 ? q.y : (op == 11 ? q.z : m));
        break;
    }
    case 13: { // exhaustive sweep: sqrt_rn against the IEEE sqrt over the 65536 bit patterns from bits(a[i]); returns the mismatch count
        const uint32_t base = __float_as_uint(a[i]);
        uint32_t bad = 0;
        for (uint32_t j = 0; j < 65536u; j++) {
            const float x = __uint_as_float(base + j);
            const float s0 = sqrt_rn(x), s1 = __builtin_sqrtf(x);
            bad += (__float_as_uint(s0) != __float_as_uint(s1)) && !(s0 != s0 && s1 != s1);
        }
        r = (float)bad;
        break;
    }
    case 14: r = dmf_logf(a[i]); break; // the Mandelbulb estimator's logarithm as the kernels evaluate it
    case 15: { // exhaustive sweep: rcp_sqrt_rn against 1 / IEEE sqrt (IEEE '/') over the 65536 bit patterns from bits(a[i]); returns the mismatch count
        const uint32_t base = __float_as_uint(a[i]);
        uint32_t bad = 0;
        for (uint32_t j = 0; j < 65536u; j++) {
            const float x = __uint_as_float(base + j);
            const float r0 = rcp_sqrt_rn(x), r1 = 1.0f / __builtin_sqrtf(x);
            bad += (__float_as_uint(r0) != __float_as_uint(r1)) && !(r0 != r0 && r1 != r1);
        }
        r = (float)bad;
        break;
    }
    case 16: r = rcp_sqrt_rn(a[i]); break;
    default: r = a[i] / b[i]; break;
    }
    out[i] = r;
}

// Scene-upload check behind DHitable::fast_div == 2: div_short(n, d) against the IEEE quotient for every float d with
// bit pattern in [lo_bits, lo_bits + count).
__global__ void __launch_bounds__(256) k_verify_short_div(float n, uint32_t lo_bits, uint32_t count, uint32_t* __restrict__ bad) {
    uint32_t local = 0;
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < count; j += (uint64_t)gridDim.x * blockDim.x) {
        const float d = __uint_as_float(lo_bits + (uint32_t)j);
        local += __float_as_uint(div_short(n, d)) != __float_as_uint(n / d);
    }
    if (local) atomicAdd(bad, local);
}

// ---- launch wrappers (declared in kernels.h) -----------------------------------------------------
static inline dim3 grid_for(uint32_t n, uint32_t block) { return dim3((n + block - 1) / block); }


void launch_pack_tables(hipStream_t s, Tables tab, float4* out, uint32_t spp, uint32_t depths, uint32_t n1, uint32_t n2) {
    hipLaunchKernelGGL(k_pack_tables, grid_for(spp * depths, 256), dim3(256), 0, s, tab, out, spp, depths, n1, n2);
}
// grid of a grid-stride kernel over at most max_items items (the host's upper bound of a device-resident count)
static inline dim3 stride_grid(uint32_t max_items, uint32_t per_block, uint32_t cap_blocks) {
    const uint32_t need = (max_items + per_block - 1) / per_block;
    return dim3(std::max<uint32_t>(1u, std::min<uint32_t>(need, cap_blocks)));
}
constexpr uint32_t STREAM_BLOCKS = 256 * 8; // light streaming kernels: 8 blocks of 256 threads per CU

void launch_batch_setup(hipStream_t s, const DTile* tiles, uint32_t n_tiles, uint32_t* pgrp_tile, uint32_t* tgb, uint32_t* tgc) {
    hipLaunchKernelGGL(k_batch_setup, dim3(n_tiles), dim3(256), 0, s, tiles, pgrp_tile, tgb, tgc);
}
void launch_raygen(hipStream_t s, const DScene* sc, Tables tab, const float* scramble, const DTile* tiles, const uint32_t* pgrp_tile,
                   Pool pool, uint32_t* q, uint32_t n_pool, DCtl* ctl) {
    hipLaunchKernelGGL(k_raygen, grid_for(n_pool, 256), dim3(256), 0, s, sc, tab, scramble, tiles, pgrp_tile, pool, q, n_pool, ctl);
}
void launch_extend(hipStream_t s, bool count, const DScene* sc, uint32_t depth, const uint32_t* q, uint32_t max_entries, Pool pool,
                   uint8_t* ent_obj, int single_sdf, DCtl* ctl, unsigned long long* evals, const Tuning& tun) {
    const dim3 grid = stride_grid(max_entries, 256, tun.persistent_blocks);
    if (single_sdf >= 0 && tun.fast_path) { // (a single-Mandelbulb scene keeps k_extend1 for its closest-hit marches: the k_shadow_bulb scheme measured slower here, tools/variants/r6_k_extend_bulb.h)
        // one instantiation per SDF kind of the scene's single TracedSDF (Tuning::sdf_kind, set per frame by the host); the counting variants stay generic
#define RAYN_EXTEND1(C, KIND) hipLaunchKernelGGL((k_extend1<C, KIND>), grid, dim3(256), 0, s, sc, depth, (uint32_t)single_sdf, q, ctl, pool, ent_obj, tun.prefetch_min_extend, evals)
        if (count) RAYN_EXTEND1(true, -1);
        else if (tun.sdf_kind == SDFK_MANDELBOX_12S) RAYN_EXTEND1(false, SDFK_MANDELBOX_12S);
        else if (tun.sdf_kind == RAYN_SDF_MANDELBOX) RAYN_EXTEND1(false, RAYN_SDF_MANDELBOX);
        else if (tun.sdf_kind == RAYN_SDF_MANDELBULB) RAYN_EXTEND1(false, RAYN_SDF_MANDELBULB);
        else RAYN_EXTEND1(false, -1);
#undef RAYN_EXTEND1
    } else if (count) hipLaunchKernelGGL(k_extend<true>, grid, dim3(256), 0, s, sc, depth, q, ctl, pool, ent_obj, tun.refill_min_extend, evals);
    else hipLaunchKernelGGL(k_extend<false>, grid, dim3(256), 0, s, sc, depth, q, ctl, pool, ent_obj, tun.refill_min_extend, evals);
}
void launch_group_hist(hipStream_t s, uint32_t nclass, const uint8_t* ent_obj, uint32_t max_entries, const DCtl* ctl, uint8_t* grp_cnt) {
    hipLaunchKernelGGL(k_group_hist, stride_grid(max_entries / 16, 256, STREAM_BLOCKS), dim3(256), 0, s, nclass, ent_obj, ctl, grp_cnt);
}
void launch_scan_tile(hipStream_t s, uint32_t n_tiles, uint32_t nclass, uint32_t stride, uint32_t pad, const uint8_t* grp_cnt,
                      const uint32_t* tgb, const uint32_t* tgc, uint32_t* grp_base, uint32_t* grp_tile, uint32_t* tile_total,
                      uint32_t* tile_valid, uint32_t* tile_cls_cnt, const DCtl* ctl) {
    hipLaunchKernelGGL(k_scan_tile, dim3(n_tiles), dim3(256), 0, s, nclass, stride, pad, grp_cnt, tgb, tgc, grp_base, grp_tile, tile_total, tile_valid, tile_cls_cnt, ctl);
}
void launch_tile_prefix(hipStream_t s, uint32_t n_tiles, const uint32_t* tile_total, const uint32_t* tile_valid, uint32_t* tile_out_base,
                        uint32_t* ogb, uint32_t* ogc, DCtl* ctl, int stage, uint32_t nclass, uint32_t pad, const uint32_t* tile_cls_cnt,
                        uint32_t* tile_cls_base, uint32_t cap_groups, uint32_t* base_hist) {
    hipLaunchKernelGGL(k_tile_prefix, dim3(1), dim3(1024), 0, s, n_tiles, tile_total, tile_valid, tile_out_base, ogb, ogc, ctl, stage, nclass, pad, tile_cls_cnt,
                       tile_cls_base, cap_groups, base_hist);
}
void launch_bin_scatter(hipStream_t s, uint32_t nclass, const uint32_t* q, const uint8_t* ent_obj, const uint32_t* grp_base,
                        const uint32_t* grp_tile, const uint32_t* tile_out_base, uint32_t max_entries, uint32_t* bq, uint32_t n_tiles,
                        const uint32_t* tile_cls_cnt, const uint32_t* tile_total, const uint32_t* tile_cls_base, const DCtl* ctl) {
    hipLaunchKernelGGL(k_bin_scatter, stride_grid(max_entries, 256, STREAM_BLOCKS), dim3(256), 0, s, nclass, q, ent_obj, grp_base, grp_tile, tile_out_base, ctl, bq,
                       n_tiles, tile_cls_cnt, tile_total, tile_cls_base);
}
// the shadow-march kernel of the scene over the job list in nee / ctl (at most max_jobs entries): k_shade's second stage, and rayn_hip_probe_shadow
void launch_shadow_march(hipStream_t s, bool count, const DScene* sc, Nee nee, uint32_t max_jobs, int single_sdf, DCtl* ctl, unsigned long long* evals, const Tuning& tun) {
    const dim3 grid = stride_grid(max_jobs, 256, tun.persistent_blocks);
    if (single_sdf >= 0 && tun.fast_path && tun.bulb) {
#define RAYN_SHADOW_BULB(C, KK, SS) hipLaunchKernelGGL((k_shadow_bulb<C, KK, SS>), grid, dim3(256), 0, s, sc, (uint32_t)single_sdf, nee, ctl, tun.bulb_orbit_min, tun.bulb_prefetch_min, evals + 2)
        if (count) { // the instrumented kernels run the SAME shape as the product ones (their stage-occupancy counters are quoted for it)
            if (tun.bulb_rays == 2) { if (tun.bulb_steps == 2) RAYN_SHADOW_BULB(true, 2, 2); else RAYN_SHADOW_BULB(true, 2, 1); }
            else if (tun.bulb_rays == 3) { if (tun.bulb_steps == 2) RAYN_SHADOW_BULB(true, 3, 2); else RAYN_SHADOW_BULB(true, 3, 1); }
            else { if (tun.bulb_steps == 2) RAYN_SHADOW_BULB(true, 4, 2); else RAYN_SHADOW_BULB(true, 4, 1); }
        }
        else if (tun.bulb_rays == 2) { if (tun.bulb_steps == 2) RAYN_SHADOW_BULB(false, 2, 2); else RAYN_SHADOW_BULB(false, 2, 1); }
        else if (tun.bulb_rays == 3) { if (tun.bulb_steps == 2) RAYN_SHADOW_BULB(false, 3, 2); else RAYN_SHADOW_BULB(false, 3, 1); }
        else { if (tun.bulb_steps == 2) RAYN_SHADOW_BULB(false, 4, 2); else RAYN_SHADOW_BULB(false, 4, 1); }
#undef RAYN_SHADOW_BULB
    } else if (single_sdf >= 0 && tun.fast_path) {
#define RAYN_SHADOW1(C, KIND) hipLaunchKernelGGL((k_shadow1<C, KIND>), grid, dim3(256), 0, s, sc, (uint32_t)single_sdf, nee, ctl, tun.prefetch_min_shadow, evals + 2)
        if (count) RAYN_SHADOW1(true, -1);
        else if (tun.sdf_kind == SDFK_MANDELBOX_12S) RAYN_SHADOW1(false, SDFK_MANDELBOX_12S);
        else if (tun.sdf_kind == RAYN_SDF_MANDELBOX) RAYN_SHADOW1(false, RAYN_SDF_MANDELBOX);
        else RAYN_SHADOW1(false, -1); // (sphere SDF; a Mandelbulb scene with the k_shadow_bulb path switched off)
#undef RAYN_SHADOW1
    } else if (count) hipLaunchKernelGGL(k_shadow<true>, grid, dim3(256), 0, s, sc, nee, ctl, tun.refill_min_shadow, evals + 2);
    else hipLaunchKernelGGL(k_shadow<false>, grid, dim3(256), 0, s, sc, nee, ctl, tun.refill_min_shadow, evals + 2);
}
void launch_shade(hipStream_t s, bool count, const DScene* sc, Tables tab, const float* scramble, uint32_t depth, const uint32_t* bq,
                  uint32_t max_slots, Pool pool, Nee nee, uint32_t ns, bool has_sdf, int single_sdf, unsigned long long* alive_mask, uint8_t* bgrp_cnt, DCtl* ctl,
                  unsigned long long* evals, ShadeHooks hooks, const Tuning& tun) {
    hooks.before(0);
    const uint32_t shmem = ns > 4 ? VOL_MEMO_LIGHTS * 3 * 256 * 4 : 0; // ns > 4: the volume scatters (volume NEE samples exist)
    const dim3 sgrid = grid_for(max_slots, 256);
    if (count) hipLaunchKernelGGL(k_shade_setup<true>, sgrid, dim3(256), shmem, s, sc, tab, scramble, depth, bq, ctl, pool, nee, alive_mask, bgrp_cnt, evals + 1);
    else hipLaunchKernelGGL(k_shade_setup<false>, sgrid, dim3(256), shmem, s, sc, tab, scramble, depth, bq, ctl, pool, nee, alive_mask, bgrp_cnt, evals + 1);
    hooks.after(0);
    if (has_sdf) {
        hooks.before(1);
        hipLaunchKernelGGL(k_shadow_list, stride_grid(ns * max_slots, 256 * SCAN_ITEMS, STREAM_BLOCKS), dim3(256), 0, s, nee, ns, ctl);
        launch_shadow_march(s, count, sc, nee, ns * max_slots, single_sdf, ctl, evals, tun);
        hooks.after(1);
    }
    hooks.before(2);
    hipLaunchKernelGGL(k_shade_finish, stride_grid(max_slots, 256, STREAM_BLOCKS), dim3(256), 0, s, sc, bq, ctl, pool, nee);
    hooks.after(2);
}
void launch_compact_scatter(hipStream_t s, const uint32_t* bq, const unsigned long long* alive_mask, const uint32_t* grp_base, const uint32_t* grp_tile,
                            const uint32_t* tile_out_base, uint32_t max_slots, uint32_t* qn, uint32_t n_tiles, const uint32_t* tile_total,
                            const DCtl* ctl) {
    hipLaunchKernelGGL(k_compact_scatter, stride_grid(max_slots, 256, STREAM_BLOCKS), dim3(256), 0, s, bq, alive_mask, grp_base, grp_tile, tile_out_base, ctl, qn, n_tiles, tile_total);
}
void launch_unpack_tiles(hipStream_t s, const DTile* tiles, uint32_t n_tiles, uint32_t width, float* color, float* alpha, float* background,
                         float* normal, const float* packed, size_t packed_pixels) {
    if (!n_tiles) return;
    const float *pc = packed, *pa = pc + 3 * packed_pixels, *pb = pa + packed_pixels, *pn = pb + 3 * packed_pixels;
    hipLaunchKernelGGL(k_unpack_tiles, dim3(n_tiles), dim3(256), 0, s, tiles, width, color, alpha, background, normal, pc, pa, pb, pn);
}
void launch_resolve(hipStream_t s, const DScene* sc, const DTile* tiles, uint32_t n_tiles, uint32_t max_tile_pixels, uint32_t spp, Pool pool,
                    float* out_color, float* out_alpha, float* out_background, float* out_normal, const uint32_t* base_hist, uint32_t hist_stride) {
    const dim3 grid(max_tile_pixels, n_tiles);
    if (spp <= 1024) { // register-resident sort, one wave per pixel
        if (spp <= 64) hipLaunchKernelGGL(k_resolve_reg<1>, grid, dim3(64), 0, s, sc, tiles, pool, out_color, out_alpha, out_background, out_normal);
        else if (spp <= 128) hipLaunchKernelGGL(k_resolve_reg<2>, grid, dim3(64), 0, s, sc, tiles, pool, out_color, out_alpha, out_background, out_normal);
        else if (spp <= 256) hipLaunchKernelGGL(k_resolve_reg<4>, grid, dim3(64), 0, s, sc, tiles, pool, out_color, out_alpha, out_background, out_normal);
        else if (spp <= 512) hipLaunchKernelGGL(k_resolve_reg<8>, grid, dim3(64), 0, s, sc, tiles, pool, out_color, out_alpha, out_background, out_normal);
        else hipLaunchKernelGGL((k_resolve_blk<128, 8>), grid, dim3(128), 0, s, sc, tiles, pool, out_color, out_alpha, out_background, out_normal, base_hist, hist_stride); // two waves per pixel
        return;
    }
    if (spp <= 4096) { // four / eight waves per pixel, 8 keys per lane (config 5: 4096 spp)
        if (spp <= 2048) hipLaunchKernelGGL((k_resolve_blk<256, 8>), grid, dim3(256), 0, s, sc, tiles, pool, out_color, out_alpha, out_background, out_normal, base_hist, hist_stride);
        else hipLaunchKernelGGL((k_resolve_blk<512, 8>), grid, dim3(512), 0, s, sc, tiles, pool, out_color, out_alpha, out_background, out_normal, base_hist, hist_stride);
        return;
    }
    // 4096 < spp <= 16384 (the host rejects more): sixteen waves per pixel, 128 KB of dynamic LDS
    constexpr int huge_lds = 16384 * 8;
    (void)hipFuncSetAttribute((const void*)k_resolve_huge, hipFuncAttributeMaxDynamicSharedMemorySize, huge_lds); // per function AND device
    hipLaunchKernelGGL(k_resolve_huge, grid, dim3(1024), huge_lds, s, sc, tiles, pool, out_color, out_alpha, out_background, out_normal);
}
void launch_probe_dist(hipStream_t s, const DScene* sc, uint32_t hit_index, const float* pts, float* out, uint32_t n) {
    hipLaunchKernelGGL(k_probe_dist, grid_for(n, 256), dim3(256), 0, s, sc, hit_index, pts, out, n);
}
void launch_verify_short_div(hipStream_t s, float n, uint32_t lo_bits, uint32_t count, uint32_t* bad) {
    hipLaunchKernelGGL(k_verify_short_div, dim3(4096), dim3(256), 0, s, n, lo_bits, count, bad);
}
void launch_probe_detmath(hipStream_t s, uint32_t op, const float* a, const float* b, float* out, uint32_t n) {
    hipLaunchKernelGGL(k_probe_detmath, grid_for(n, 256), dim3(256), 0, s, op, a, b, out, n);
}

} // namespace RAYN_KNS
