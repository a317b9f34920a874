// march_bulb.h — the shadow-march kernel written for the power-8 Mandelbulb (EXTENSION: the fractal BASELINE.json names; the reference's only fractal
// is the MandelBox, src/sdf.rs:104-141).  Included by kernels.hip inside its namespace; same march semantics as k_shadow1 (TracedSDF::occluded,
// src/sdf.rs:25-57), same results bit for bit (device_core.h: bulb_begin / bulb_step / bulb_finish ARE mandelbulb_dist, cut in three).
//
// Why a kernel of its own.  A Mandelbulb evaluation is an ORBIT of 1..iterations steps that ends when |w|^2 > 256: on the metric's named workload
// (bulb3) a lane runs 4.4 steps per evaluation, but a wave runs to its slowest lane - 8 - so with one evaluation per loop trip (k_shadow1) 58 % of
// the lanes are enabled on average (PMC, profiles/r05_pmc_hbm_bulb3.json).  Making one trip one orbit STEP of a single ray does not help by itself:
// what follows an orbit (binary64 logarithm, exact sqrt, IEEE division, march test, next point: ~250 VALU instructions) would then run for the few
// lanes that happen to finish in that trip.  The evaluation is therefore cut in two stages that run at their own occupancy, over MORE rays than lanes:
// every lane owns K rays, and the wave works in rounds (k_shadow_bulb below).  Measured on bulb3 (profiles/r06_exp_bulb.txt): k_shadow1 6 568 ms per
// frame -> 5 503 ms (K = 3; orbit stage 88 %, epilogue stage 83 % occupied).  What was tried on the way, all bit-identical (tools/variants/README.md):
// two rays per lane taking turns in one set of orbit registers, epilogues batched at 40 lanes (6 684 / 6 222 ms with one / two steps per trip: the
// selects between the two rays and the per-trip bookkeeping ate the gain); rounds that drain every orbit before the epilogues (5 860 ms at K = 4, orbit
// stage 79 %); K = 2 / 4 with carry-over (5 817 / 6 098 ms: these kernels want their 7-8 waves per SIMD - 84 VGPRs at K = 4 cost more than the fuller
// stages give); a refill pipelined over three rounds (6 178 ms); the same scheme for the closest-hit marches (k_extend1 stays: 1 366 vs 1 425 ms).
#pragma once

constexpr uint32_t BC_FIRST = 1u << 16, BC_NAN = 1u << 17, BC_COUNT = 0xFFFFu;        // march flags + march count of a ray

// ------------------------------------------------------------------------------------------------
// v2 (rounds): every lane owns K rays in registers; the wave works in ROUNDS of three phases
//   A  refill    empty ray slots take the next entries of the job queue (bulk, one atomic per 256 entries)
//   B  orbits    the 64 K waiting points of the wave are a job list in LDS (one float4 per job: point in, |w|^2 / dz / steps out).  ANY lane
//                runs ANY job: a lane whose orbit ends stores the result and pulls the next unstarted job (ballot + mbcnt on a wave-uniform
//                counter), so the orbit steps run at full occupancy.  The phase does NOT wait for the slowest orbits: once every job has been started and
//                fewer than ORBIT_MIN lanes are still inside one, it ends - those lanes keep their orbit in registers and carry on in the next round
//                (their jobs are marked in flight; their rays simply skip this round's epilogue)
//   C  epilogues K unrolled passes, pass k = ray k of every lane: distance from the orbit's result, one step of the march, the next point back
//                into the job list - no selects (pass k touches only the registers of ray k) and all lanes busy
// The LDS traffic is one 128-bit write + one 128-bit read per evaluation and lane phase; the job list is private to its wave (LDS operations of a wave
// execute in order: no barrier, the compiler fences below only pin the program order).
// ------------------------------------------------------------------------------------------------
// .w of a job: a point waiting for its orbit, an orbit some lane is running, a finished orbit's result, the job of an empty ray slot
constexpr uint32_t BJ_POINT = 0u, BJ_INFLIGHT = 1u, BJ_RESULT = 2u, BJ_INVALID = 0xFFFFFFFFu;

template <bool COUNT, uint32_t K, uint32_t STEPS>
__global__ void __launch_bounds__(256, K == 2 ? 8 : (K == 3 ? 7 : 6)) k_shadow_bulb(const DScene* __restrict__ scp, uint32_t ks, Nee nee, DCtl* __restrict__ ctl,
                                                      uint32_t ORBIT_MIN, uint32_t REFILL_MIN, unsigned long long* __restrict__ evals_out) {
    __shared__ float4 s_jobs[4][64 * K];
    float4* const jobs = s_jobs[threadIdx.x >> 6];
#ifndef RAYN_BULB_LOGTAB_GLOBAL
    // the logarithm's table (128 x { 1 / c_i, ln c_i }, 2 KB) next to the job lists: an epilogue pass waits for one 128-bit table read per ray, and from LDS that is
    // a tenth of the latency of the read-only global path
    __shared__ __attribute__((aligned(16))) double s_logtab[256];
    s_logtab[threadIdx.x] = RAYN_LOGTAB[threadIdx.x];
    __syncthreads();
    const double* const logtab = s_logtab;
#else
    const double* const logtab = RAYN_LOGTAB;
#endif
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
    // the lane's K rays: segment start / unit direction / length in the SDF's frame, march distance, job ref, march count + flags
    f3 st[K], dr[K];
    float mx[K], t[K];
    uint32_t ref[K], cnt[K];
    bool act[K];     // the slot holds a live ray
#pragma unroll
    for (uint32_t k = 0; k < K; k++) { st[k] = dr[k] = f3{0, 0, 0}; mx[k] = t[k] = 0.0f; ref[k] = cnt[k] = 0; act[k] = false; }
    // the orbit registers of the lane (an orbit may span rounds)
    bool o_valid = false;
    uint32_t my = 0, o_it = 0;
    f3 o_p = f3{0, 0, 0};
    BulbOrbit o = BulbOrbit{f3{0, 0, 0}, 0.0f, 1.0f};
    for (;;) {
        // ---- A: refill the empty ray slots (near the end of the queue only slot 0: no hoarding while other waves idle) - in bulk, once REFILL_MIN
        // slots of the wave are empty: a fetch is two dependent gathers (job ref -> segment) that the whole wave waits for.  (Pipelining the
        // fetch over three rounds - ref, then segment, then use, each issued a round ahead into the dead registers of the empty slot - was
        // measured SLOWER, 6.18 vs 5.49 s of k_shadow_bulb per bulb3 frame: slots sit empty for two more rounds, tools/variants/README.md.)
        uint32_t n_act = 0;
#pragma unroll
        for (uint32_t k = 0; k < K; k++) n_act += (uint32_t)__popcll(__ballot(act[k]));
        const bool refill = !exhausted && (64u * K - n_act >= REFILL_MIN || n_act < 64u);
#pragma unroll
        for (uint32_t k = 0; k < K; k++) {
            if (refill && !exhausted && !(endgame && k > 0)) {
                for (;;) {
                    const uint64_t need = __ballot(!act[k]);
                    if (need == 0) break;
                    if (cur == end) {
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(head, CHUNK);
                        base = __builtin_amdgcn_readfirstlane(base);
                        if (base >= n_jobs) { exhausted = true; break; }
                        cur = base;
                        end = min(base + CHUNK, n_jobs);
                        endgame = n_jobs - base < ENDGAME_ENTRIES * K;
                    }
                    const uint32_t rank = mbcnt(need), avail = end - cur;
                    if (!act[k] && rank < avail) {
                        const uint32_t n_ref = nee.job_ref[cur + rank];
                        const float2 j0 = nee.job_geo[3 * (size_t)n_ref], j1 = nee.job_geo[3 * (size_t)n_ref + 1], j2 = nee.job_geo[3 * (size_t)n_ref + 2];
                        const float jt0 = sc.anim_spheres ? nee.t0[n_ref % (uint32_t)nee.cap] : 0.0f;
                        const f3 origin = sphere_center(h, jt0); // TracedSDF origin at the packet time (extension; zero in the reference)
                        st[k] = f3{j0.x, j0.y, j1.x} - origin;
                        const f3 e = f3{j1.y, j2.x, j2.y} - origin;
                        const f3 d = e - st[k];
                        mx[k] = mag(d);
                        dr[k] = div_by_mag(d, mx[k]);
                        ref[k] = n_ref; cnt[k] = BC_FIRST; t[k] = 0.0f; act[k] = true;
                        jobs[lane + 64u * k] = make_float4(st[k].x, st[k].y, st[k].z, __uint_as_float(BJ_POINT)); // the first evaluation is at the segment start
                    }
                    cur += min((uint32_t)__popcll(need), avail);
                }
            }
        }
        uint32_t total = 0; // live rays of the wave
#pragma unroll
        for (uint32_t k = 0; k < K; k++) {
            if (!act[k]) jobs[lane + 64u * k] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(BJ_INVALID));
            total += (uint32_t)__popcll(__ballot(act[k]));
        }
        if (total == 0) break; // queue exhausted and every ray finished
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- B: orbits
        {
            const uint32_t leave = min(ORBIT_MIN, total >> 2); // the phase ends once at most this many lanes are still inside an orbit (few rays: drain them)
            uint32_t next = 0; // first job not yet offered to a lane (wave-uniform)
            for (;;) {
                const uint64_t need = __ballot(!o_valid);
                if (need != 0 && next < 64u * K) {
                    if (!o_valid) {
                        const uint32_t jn = next + mbcnt(need);
                        if (jn < 64u * K) {
                            const float4 j = jobs[jn];
                            if (__float_as_uint(j.w) == BJ_POINT) { // (a job in flight belongs to the lane that carried it over; an invalid one to an empty ray slot)
                                my = jn;
                                ((float*)&jobs[jn])[3] = __uint_as_float(BJ_INFLIGHT);
                                o_p = f3{j.x, j.y, j.z};
                                o = bulb_begin(o_p);
                                o_it = 0;
                                o_valid = true;
                            }
                        }
                    }
                    next += (uint32_t)__popcll(need);
                }
                const uint64_t busy = __ballot(o_valid);
                if (next >= 64u * K && (uint32_t)__popcll(busy) <= leave) break;
                if (busy == 0) continue;
#pragma unroll
                for (uint32_t s = 0; s < STEPS; s++) {
                    if (COUNT) n_orbit_trips++;
                    if (o_valid) {
                        bulb_step(o, o_p);
                        o_it++;
                        if (o.m > BULB_BAILOUT || o_it == iterations) { // the orbit has ended: its result replaces the point in the job list
                            jobs[my] = make_float4(o.m, o.dz, __uint_as_float(o_it), __uint_as_float(BJ_RESULT));
                            o_valid = false;
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- C: distance + one step of TracedSDF::occluded (src/sdf.rs:25-57) for ray k of every lane
        auto epilogue = [&](const uint32_t k) { // k is a literal at every call: pass k touches the registers of ray k only
            if (__ballot(act[k]) == 0) return;
            if (COUNT) n_epi_passes++;
            float4 r = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (act[k]) r = jobs[lane + 64u * k];
            if (act[k] && __float_as_uint(r.w) == BJ_RESULT) { // (an orbit still in flight: next round)
                const float dist = bulb_finish_inl(r.x, r.y, logtab);
                if (COUNT) { evals.n++; evals.it += __float_as_uint(r.z); }
                bool nan = (cnt[k] & BC_NAN) != 0;
                int res = -1; // -1 keep marching, 0 occluded, 1 visible
                if (cnt[k] & BC_FIRST) {
                    t[k] = dist; nan = dist != dist; cnt[k] = nan ? BC_NAN : 0u;
                    if (max_vis == 0) res = ((dist < 0.0001f) && !((dist > mx[k]) || nan)) ? 0 : 1;
                    else if ((t[k] > mx[k]) || nan) res = 1;
                } else {
                    if (__builtin_fabsf(dist) < fmaxs(c0, c1 * t[k])) res = 0;
                    else {
                        t[k] = t[k] + dist; cnt[k]++;
                        if ((cnt[k] & BC_COUNT) == max_vis || (t[k] > mx[k]) || nan) res = 1;
                    }
                }
                if (res >= 0) { if (res == 1) nee.vis[ref[k]] = 1; act[k] = false; } // only VISIBLE results are written (see Nee::vis)
                else {
                    const f3 pn = muladd3(dr[k], t[k], st[k]); // the next point of the march
                    jobs[lane + 64u * k] = make_float4(pn.x, pn.y, pn.z, __uint_as_float(BJ_POINT));
                }
            }
        };
        epilogue(0);
        if (K > 1) epilogue(1);
        if (K > 2) epilogue(K > 2 ? 2 : 0);
        if (K > 3) epilogue(K > 3 ? 3 : 0);
    }
    if (COUNT && evals.n) { atomicAdd(evals_out, (unsigned long long)evals.n); atomicAdd(evals_out + 4, (unsigned long long)evals.it); }
    if (COUNT && lane == 0) { atomicAdd(evals_out + 10, 64ull * n_orbit_trips); atomicAdd(evals_out + 11, 64ull * n_epi_passes); } // evals_out = &evals[2]: [12], [13]
}

