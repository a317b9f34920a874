// march_bulb.h — the two march kernels written for the power-8 Mandelbulb (EXTENSION: the fractal BASELINE.json names; the reference's only fractal
// is the MandelBox, src/sdf.rs:104-141).  Included by kernels.hip inside its namespace; same march semantics as k_extend1 / k_shadow1
// (TracedSDF::hit src/sdf.rs:59-83, TracedSDF::occluded src/sdf.rs:25-57, the first-wins fold of src/hitable.rs:177-198), same results bit for bit.
//
// Why a kernel of its own.  A Mandelbulb evaluation is an ORBIT of 1..iterations steps that ends when |w|^2 > 256: on the metric's named workload
// (bulb3) a lane runs 4.4 steps per evaluation, but a wave runs to its slowest lane - 8 - so with one evaluation per loop trip (k_shadow1) 58 % of
// the lanes are enabled on average (PMC, profiles/r05_pmc_hbm_bulb3.json).  Making one trip one orbit STEP of a single ray does not help by itself:
// what follows an orbit (binary64 logarithm, exact sqrt, IEEE division, march test, next point: ~250 VALU instructions) would then run for the few
// lanes that happen to finish in that trip.  So every lane owns TWO rays and the evaluation is cut in two stages that run at their own occupancy:
//   orbit stage     ONE register set per lane (w, |w|^2, dz, step count, the point) holding the orbit of whichever of the lane's rays needs one;
//                   a trip runs STEPS orbit steps on every lane whose set is loaded - a lane whose orbit ends loads its other ray's next trip;
//   epilogue stage  rays whose orbit has ended wait (two floats) until EPI_MIN lanes hold one, then distance + march step + next point run
//                   for all of them at once.
// The rays of a lane take turns in the orbit registers, so the steps run at (nearly) full occupancy and the epilogue at >= EPI_MIN / 64.  Ray state lives in
// registers (north_star's "LDS-staged march state" was measured slower on these kernels); the selects between the two rays of a lane
// are v_cndmask moves in the epilogue / load blocks, never in the orbit step.
#pragma once

constexpr uint32_t BP_EMPTY = 0, BP_NEED_ORBIT = 1, BP_IN_ORBIT = 2, BP_NEED_EPI = 3; // phase of a ray slot
constexpr uint32_t BC_FIRST = 1u << 16, BC_NAN = 1u << 17, BC_COUNT = 0xFFFFu;        // march flags + march count of a ray

RD f3 sel3(bool c, f3 a, f3 b) { return f3{c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z}; }

// ------------------------------------------------------------------------------------------------
// TracedSDF::occluded for the pending shadow segments of a single-Mandelbulb scene (cf. k_shadow1)
// ------------------------------------------------------------------------------------------------
template <bool COUNT, uint32_t STEPS>
__global__ void __launch_bounds__(256) k_shadow_bulb(const DScene* __restrict__ scp, uint32_t ks, Nee nee, DCtl* __restrict__ ctl,
                                                      uint32_t PREFETCH_MIN, uint32_t EPI_MIN, unsigned long long* __restrict__ evals_out) {
    const DScene& sc = *scp;
    const uint32_t lane = lane_id();
    const uint32_t n_jobs = ctl->job_count, max_vis = sc.max_vis_marches;
    uint32_t* const head = &ctl->head_shadow;
    if (blockIdx.x == 0 && threadIdx.x == 0) ctl->shadow_jobs += n_jobs;
    const DHitable h = sc.h[ks];
    const uint32_t iterations = h.iterations; // >= 1 (the host takes the generic kernels otherwise)
    const float c0 = 0.0001f * sc.detail_scale, c1 = 0.00001f * sc.detail_scale;
    uint32_t cur = 0, end = 0;
    bool exhausted = false, endgame = false;
    EvalCtr evals;
    uint32_t n_orbit_trips = 0, n_epi_passes = 0; // COUNT builds: executions of the two stages by this wave (x 64 = lane slots offered)
    // the lane's two rays: segment start / unit direction / length in the SDF's frame, march distance, job ref, march count + flags, and three
    // floats that hold the next point (BP_NEED_ORBIT) or |w|^2, dz, orbit steps (BP_NEED_EPI)
    uint32_t ph0 = BP_EMPTY, ph1 = BP_EMPTY, ref0 = 0, ref1 = 0, cnt0 = 0, cnt1 = 0;
    f3 st0 = f3{0, 0, 0}, st1 = st0, dr0 = st0, dr1 = st0, x0 = st0, x1 = st0;
    float mx0 = 0.0f, mx1 = 0.0f, t0 = 0.0f, t1 = 0.0f;
    // the orbit registers
    bool o_valid = false, o_slot = false;
    f3 o_p = f3{0, 0, 0};
    BulbOrbit o = BulbOrbit{f3{0, 0, 0}, 0.0f, 1.0f};
    uint32_t o_it = 0;
    for (;;) {
        // ---- (1) queue fetch: in bulk, when enough lanes have a free ray slot (cf. k_shadow1's spare ray)
        {
            const uint64_t lack = __ballot(ph0 == BP_EMPTY || ph1 == BP_EMPTY);
            const uint64_t idle = __ballot(ph0 == BP_EMPTY && ph1 == BP_EMPTY);
            if (!exhausted && (endgame ? idle != 0 : ((uint32_t)__popcll(lack) >= PREFETCH_MIN || (uint32_t)__popcll(idle) >= 4u))) {
                for (;;) {
                    const uint64_t need = endgame ? __ballot(ph0 == BP_EMPTY && ph1 == BP_EMPTY) : __ballot(ph0 == BP_EMPTY || ph1 == BP_EMPTY);
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
                        const uint32_t n_ref = nee.job_ref[cur + rank];
                        const float2 j0 = nee.job_geo[3 * (size_t)n_ref], j1 = nee.job_geo[3 * (size_t)n_ref + 1], j2 = nee.job_geo[3 * (size_t)n_ref + 2];
                        const float jt0 = sc.anim_spheres ? nee.t0[n_ref % (uint32_t)nee.cap] : 0.0f;
                        const f3 origin = sphere_center(h, jt0); // TracedSDF origin at the packet time (extension; zero in the reference)
                        const f3 n_start = f3{j0.x, j0.y, j1.x} - origin;
                        const f3 e = f3{j1.y, j2.x, j2.y} - origin;
                        f3 n_dir = e - n_start;
                        const float n_max = mag(n_dir);
                        n_dir = div_by_mag(n_dir, n_max);
                        if (ph0 == BP_EMPTY) { st0 = n_start; dr0 = n_dir; mx0 = n_max; ref0 = n_ref; cnt0 = BC_FIRST; x0 = n_start; t0 = 0.0f; ph0 = BP_NEED_ORBIT; }
                        else { st1 = n_start; dr1 = n_dir; mx1 = n_max; ref1 = n_ref; cnt1 = BC_FIRST; x1 = n_start; t1 = 0.0f; ph1 = BP_NEED_ORBIT; }
                    }
                    cur += min((uint32_t)__popcll(need), avail);
                }
            }
        }
        // ---- (2) epilogue stage: distance of a finished orbit + one step of TracedSDF::occluded, src/sdf.rs:25-57
        {
            const bool e0 = ph0 == BP_NEED_EPI, e1 = ph1 == BP_NEED_EPI;
            const uint64_t epi = __ballot(e0 || e1);
            const uint64_t orbitable = __ballot(o_valid || ph0 == BP_NEED_ORBIT || ph1 == BP_NEED_ORBIT);
            if (epi != 0 && ((uint32_t)__popcll(epi) >= EPI_MIN || orbitable == 0)) {
                if (COUNT) n_epi_passes++;
                if (e0 || e1) {
                    const bool s1 = !e0; // the ray this pass serves
                    const f3 xs = sel3(s1, x1, x0);
                    const float dist = bulb_finish(xs.x, xs.y);
                    if (COUNT) { evals.n++; evals.it += __float_as_uint(xs.z); }
                    float t = s1 ? t1 : t0;
                    const float max_dist = s1 ? mx1 : mx0;
                    uint32_t cnt = s1 ? cnt1 : cnt0;
                    bool nan = (cnt & BC_NAN) != 0;
                    int res = -1; // -1 keep marching, 0 occluded, 1 visible
                    if (cnt & BC_FIRST) {
                        t = dist; nan = dist != dist; cnt = nan ? BC_NAN : 0u;
                        if (max_vis == 0) res = ((dist < 0.0001f) && !((dist > max_dist) || nan)) ? 0 : 1;
                        else if ((t > max_dist) || nan) res = 1;
                    } else {
                        if (__builtin_fabsf(dist) < fmaxs(c0, c1 * t)) res = 0;
                        else {
                            t = t + dist; cnt++;
                            if ((cnt & BC_COUNT) == max_vis || (t > max_dist) || nan) res = 1;
                        }
                    }
                    uint32_t ph = BP_NEED_ORBIT;
                    f3 xn = xs;
                    if (res >= 0) { if (res == 1) nee.vis[s1 ? ref1 : ref0] = 1; ph = BP_EMPTY; } // only VISIBLE results are written (see Nee::vis)
                    else xn = muladd3(sel3(s1, dr1, dr0), t, sel3(s1, st1, st0));                // the next point of the march
                    if (s1) { t1 = t; cnt1 = cnt; ph1 = ph; x1 = xn; }
                    else { t0 = t; cnt0 = cnt; ph0 = ph; x0 = xn; }
                }
            }
        }
        // ---- (3) free orbit registers take the lane's next waiting point
        if (!o_valid && (ph0 == BP_NEED_ORBIT || ph1 == BP_NEED_ORBIT)) {
            o_slot = ph0 != BP_NEED_ORBIT;
            o_p = sel3(o_slot, x1, x0);
            o = bulb_begin(o_p);
            o_it = 0;
            o_valid = true;
            if (o_slot) ph1 = BP_IN_ORBIT; else ph0 = BP_IN_ORBIT;
        }
        if (__ballot(o_valid) == 0) {
            if (exhausted && __ballot(ph0 != BP_EMPTY || ph1 != BP_EMPTY) == 0) break;
            continue;
        }
        // ---- (4) orbit stage
#pragma unroll
        for (uint32_t s = 0; s < STEPS; s++) {
            if (COUNT) n_orbit_trips++;
            if (o_valid) {
                bulb_step(o, o_p);
                o_it++;
                if (o.m > BULB_BAILOUT || o_it == iterations) { // the orbit has ended: park (|w|^2, dz) with its ray
                    const f3 xe = f3{o.m, o.dz, __uint_as_float(o_it)};
                    if (o_slot) { x1 = xe; ph1 = BP_NEED_EPI; } else { x0 = xe; ph0 = BP_NEED_EPI; }
                    o_valid = false;
                }
            }
        }
    }
    if (COUNT && evals.n) { atomicAdd(evals_out, (unsigned long long)evals.n); atomicAdd(evals_out + 4, (unsigned long long)evals.it); }
    if (COUNT && lane == 0) { atomicAdd(evals_out + 10, 64ull * n_orbit_trips); atomicAdd(evals_out + 11, 64ull * n_epi_passes); } // evals_out = &evals[2]: [12], [13]
}

// ------------------------------------------------------------------------------------------------
// HitableStore::add_hits for a single-Mandelbulb scene (cf. k_extend1: sphere candidates folded at fetch time, first-wins minimum after the march)
// ------------------------------------------------------------------------------------------------
template <bool COUNT, uint32_t STEPS>
__global__ void __launch_bounds__(256) k_extend_bulb(const DScene* __restrict__ scp, uint32_t depth, uint32_t ks, const uint32_t* __restrict__ q,
                                                      DCtl* __restrict__ ctl, Pool pool, uint8_t* __restrict__ ent_obj,
                                                      uint32_t PREFETCH_MIN, uint32_t EPI_MIN, unsigned long long* __restrict__ evals_out) {
    const DScene& sc = *scp;
    const uint32_t n_entries = ctl->q_groups << 6;
    uint32_t* const head = &ctl->head_extend;
    const uint32_t lane = lane_id();
    const Thr th = make_thr(sc, depth);
    const uint32_t nh = sc.n_hitables, max_marches = sc.max_marches;
    const DHitable h = sc.h[ks];
    const uint32_t iterations = h.iterations;
    const float c0 = 0.00005f * sc.detail_scale, c1 = 0.05f * sc.detail_scale;
    uint32_t cur = 0, end = 0;
    bool exhausted = false, endgame = false;
    EvalCtr evals;
    uint32_t n_orbit_trips = 0, n_epi_passes = 0;
    // the lane's two rays: origin (SDF frame) / direction, march distance, closest-so-far of the spheres before / after the SDF + their ids, pool slot,
    // queue entry, march count + flags, and the three floats of the waiting point / finished orbit
    uint32_t ph0 = BP_EMPTY, ph1 = BP_EMPTY, P0 = 0, P1 = 0, en0 = 0, en1 = 0, cnt0 = 0, cnt1 = 0, ids0 = 0, ids1 = 0;
    f3 og0 = f3{0, 0, 0}, og1 = og0, dr0 = og0, dr1 = og0, x0 = og0, x1 = og0;
    float pre0 = 0.0f, pre1 = 0.0f, post0 = 0.0f, post1 = 0.0f, t0 = 0.0f, t1 = 0.0f;
    bool o_valid = false, o_slot = false;
    f3 o_p = f3{0, 0, 0};
    BulbOrbit o = BulbOrbit{f3{0, 0, 0}, 0.0f, 1.0f};
    uint32_t o_it = 0;
    for (;;) {
        // ---- (1) queue fetch
        {
            const uint64_t lack = __ballot(ph0 == BP_EMPTY || ph1 == BP_EMPTY);
            const uint64_t idle = __ballot(ph0 == BP_EMPTY && ph1 == BP_EMPTY);
            if (!exhausted && (endgame ? idle != 0 : ((uint32_t)__popcll(lack) >= PREFETCH_MIN || (uint32_t)__popcll(idle) >= 4u))) {
                for (;;) {
                    const uint64_t need = endgame ? __ballot(ph0 == BP_EMPTY && ph1 == BP_EMPTY) : __ballot(ph0 == BP_EMPTY || ph1 == BP_EMPTY);
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
                    if (((need >> lane) & 1ull) && rank < avail) {
                        const uint32_t n_ent = cur + rank;
                        const uint32_t n_P = q[n_ent];
                        if (n_P == INVALID) ent_obj[n_ent] = (uint8_t)OBJ_NONE;
                        else {
                            const float4 g0 = pool.geo0[n_P];
                            const float2 g1 = *(const float2*)(&pool.geo1[n_P].x);
                            f3 n_o = f3{g0.x, g0.y, g0.z};
                            const f3 n_d = f3{g0.w, g1.x, g1.y};
                            const float tt0 = packet_time(sc, q, pool, n_ent);
                            float closest = sc.t_max;
                            uint32_t id = OBJ_NONE;
                            for (uint32_t k = 0; k < ks; k++) { // spheres before the SDF: the true fold
                                float ts = sphere_hit(sc.h[k], n_o, n_d, closest, tt0);
                                if (ts < closest) { closest = ts; id = k; }
                            }
                            const float n_pre = closest;
                            uint32_t idp = OBJ_NONE;
                            for (uint32_t k = ks + 1; k < nh; k++) { // spheres after it: candidates (see k_extend1)
                                float ts = sphere_hit(sc.h[k], n_o, n_d, closest, tt0);
                                if (ts < closest) { closest = ts; idp = k; }
                            }
                            const float n_post = closest;
                            const uint32_t n_ids = id | (idp << 8);
                            n_o = n_o - sphere_center(h, tt0); // march in the SDF's frame (extension; zero origin in the reference)
                            if (ph0 == BP_EMPTY) { og0 = n_o; dr0 = n_d; pre0 = n_pre; post0 = n_post; ids0 = n_ids; P0 = n_P; en0 = n_ent; cnt0 = BC_FIRST; x0 = n_o; t0 = 0.0f; ph0 = BP_NEED_ORBIT; }
                            else { og1 = n_o; dr1 = n_d; pre1 = n_pre; post1 = n_post; ids1 = n_ids; P1 = n_P; en1 = n_ent; cnt1 = BC_FIRST; x1 = n_o; t1 = 0.0f; ph1 = BP_NEED_ORBIT; }
                        }
                    }
                    cur += min((uint32_t)__popcll(need), avail);
                }
            }
        }
        // ---- (2) epilogue stage: distance of a finished orbit + one step of TracedSDF::hit, src/sdf.rs:59-83
        {
            const bool e0 = ph0 == BP_NEED_EPI, e1 = ph1 == BP_NEED_EPI;
            const uint64_t epi = __ballot(e0 || e1);
            const uint64_t orbitable = __ballot(o_valid || ph0 == BP_NEED_ORBIT || ph1 == BP_NEED_ORBIT);
            if (epi != 0 && ((uint32_t)__popcll(epi) >= EPI_MIN || orbitable == 0)) {
                if (COUNT) n_epi_passes++;
                if (e0 || e1) {
                    const bool s1 = !e0;
                    const f3 xs = sel3(s1, x1, x0);
                    const float dist = bulb_finish(xs.x, xs.y);
                    if (COUNT) { evals.n++; evals.it += __float_as_uint(xs.z); }
                    float t = s1 ? t1 : t0;
                    const float c_pre = s1 ? pre1 : pre0;
                    uint32_t cnt = s1 ? cnt1 : cnt0;
                    bool nan = (cnt & BC_NAN) != 0;
                    bool done;
                    if (cnt & BC_FIRST) { t = dist; nan = dist != dist; cnt = nan ? BC_NAN : 0u; done = max_marches == 0; }
                    else {
                        const bool hit = __builtin_fabsf(dist) < fmaxs(c0, c1 * thr_at(th, t));
                        const bool gt = t > c_pre;
                        done = hit || nan || gt;
                        if (!done) { t = t + dist; cnt++; done = (cnt & BC_COUNT) == max_marches; }
                    }
                    uint32_t ph = BP_NEED_ORBIT;
                    f3 xn = xs;
                    if (done) {
                        const uint32_t c_ids = s1 ? ids1 : ids0, c_P = s1 ? P1 : P0;
                        float closest = c_pre;
                        uint32_t id = c_ids & 0xFFu;
                        if (t < closest) { closest = t; id = ks; }
                        const uint32_t idp = c_ids >> 8;
                        const float c_post = s1 ? post1 : post0;
                        if (idp != OBJ_NONE && c_post < closest) { closest = c_post; id = idp; }
                        pool.geo1[c_P].z = closest;
                        ((uint8_t*)&pool.geo1[c_P].w)[0] = (uint8_t)id; // low byte of the bits word = hit object
                        ent_obj[s1 ? en1 : en0] = (uint8_t)id;
                        ph = BP_EMPTY;
                    } else xn = muladd3(sel3(s1, dr1, dr0), t, sel3(s1, og1, og0));
                    if (s1) { t1 = t; cnt1 = cnt; ph1 = ph; x1 = xn; }
                    else { t0 = t; cnt0 = cnt; ph0 = ph; x0 = xn; }
                }
            }
        }
        // ---- (3) free orbit registers take the lane's next waiting point
        if (!o_valid && (ph0 == BP_NEED_ORBIT || ph1 == BP_NEED_ORBIT)) {
            o_slot = ph0 != BP_NEED_ORBIT;
            o_p = sel3(o_slot, x1, x0);
            o = bulb_begin(o_p);
            o_it = 0;
            o_valid = true;
            if (o_slot) ph1 = BP_IN_ORBIT; else ph0 = BP_IN_ORBIT;
        }
        if (__ballot(o_valid) == 0) {
            if (exhausted && __ballot(ph0 != BP_EMPTY || ph1 != BP_EMPTY) == 0) break;
            continue;
        }
        // ---- (4) orbit stage
#pragma unroll
        for (uint32_t s = 0; s < STEPS; s++) {
            if (COUNT) n_orbit_trips++;
            if (o_valid) {
                bulb_step(o, o_p);
                o_it++;
                if (o.m > BULB_BAILOUT || o_it == iterations) {
                    const f3 xe = f3{o.m, o.dz, __uint_as_float(o_it)};
                    if (o_slot) { x1 = xe; ph1 = BP_NEED_EPI; } else { x0 = xe; ph0 = BP_NEED_EPI; }
                    o_valid = false;
                }
            }
        }
    }
    if (COUNT && evals.n) { atomicAdd(evals_out, (unsigned long long)evals.n); atomicAdd(evals_out + 4, (unsigned long long)evals.it); }
    if (COUNT && lane == 0) { atomicAdd(evals_out + 9, 64ull * n_orbit_trips); atomicAdd(evals_out + 10, 64ull * n_epi_passes); } // evals_out = &evals[0]: [9], [10]
}
