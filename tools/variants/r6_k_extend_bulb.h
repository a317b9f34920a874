// r6 experiment, NOT in the product: the closest-hit counterpart of k_shadow_bulb (rayn_amd/csrc/march_bulb.h), same rounds of refill / orbits / epilogues.
// Bit-identical (the seven Mandelbulb parity cases incl. the bulb3 / bulb4 / bulb5 tile digests) and SLOWER than the generic k_extend1 on the metric's named
// workload (bulb3, whole frame, profiles/r06_exp_bulb.txt): k_extend1 1 366 ms; K = 2 rays per lane 1 425 ms (72 VGPRs, 7 waves per SIMD; orbit stage 82 %,
// epilogue stage 76 % occupied); K = 3 1 429 ms (80 VGPRs + 5 spills, 6 waves per SIMD; 88 % / 83 %).  A camera / bounce ray's orbits are shorter (3.7 steps
// per evaluation against a shadow ray's 4.4: most of its march is far from the surface) and its epilogue carries more state (13 registers per ray against 10),
// so the decoupling gains less than the lost waves per SIMD cost.  To re-measure: paste into march_bulb.h, add the launch to launch_extend (kernels.hip).
// ------------------------------------------------------------------------------------------------
// HitableStore::add_hits for a single-Mandelbulb scene, same rounds (cf. k_extend1: sphere candidates folded at fetch time, first-wins minimum after the march)
// ------------------------------------------------------------------------------------------------
template <bool COUNT, uint32_t K, uint32_t STEPS>
__global__ void __launch_bounds__(256, K == 2 ? 7 : 6) k_extend_bulb(const DScene* __restrict__ scp, uint32_t depth, uint32_t ks, const uint32_t* __restrict__ q,
                                                      DCtl* __restrict__ ctl, Pool pool, uint8_t* __restrict__ ent_obj,
                                                      uint32_t ORBIT_MIN, uint32_t REFILL_MIN, unsigned long long* __restrict__ evals_out) {
    __shared__ float4 s_jobs[4][64 * K];
    float4* const jobs = s_jobs[threadIdx.x >> 6];
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
    // the lane's K rays: origin (SDF frame) / direction, march distance, closest-so-far of the spheres before / after the SDF + their ids, pool slot,
    // queue entry, march count + flags
    f3 og[K], dr[K];
    float t[K], pre[K], post[K];
    uint32_t ids[K], Pk[K], en[K], cnt[K];
    bool act[K];
#pragma unroll
    for (uint32_t k = 0; k < K; k++) { og[k] = dr[k] = f3{0, 0, 0}; t[k] = pre[k] = post[k] = 0.0f; ids[k] = Pk[k] = en[k] = cnt[k] = 0; act[k] = false; }
    bool o_valid = false;
    uint32_t my = 0, o_it = 0;
    f3 o_p = f3{0, 0, 0};
    BulbOrbit o = BulbOrbit{f3{0, 0, 0}, 0.0f, 1.0f};
    for (;;) {
        // ---- A: refill (see k_shadow_bulb)
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
                        if (base >= n_entries) { exhausted = true; break; }
                        cur = base;
                        end = min(base + CHUNK, n_entries);
                        endgame = n_entries - base < ENDGAME_ENTRIES * K;
                    }
                    const uint32_t rank = mbcnt(need), avail = end - cur;
                    if (!act[k] && rank < avail) {
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
                            for (uint32_t kk = 0; kk < ks; kk++) { // spheres before the SDF: the true fold
                                float ts = sphere_hit(sc.h[kk], n_o, n_d, closest, tt0);
                                if (ts < closest) { closest = ts; id = kk; }
                            }
                            pre[k] = closest;
                            uint32_t idp = OBJ_NONE;
                            for (uint32_t kk = ks + 1; kk < nh; kk++) { // spheres after it: candidates (see k_extend1)
                                float ts = sphere_hit(sc.h[kk], n_o, n_d, closest, tt0);
                                if (ts < closest) { closest = ts; idp = kk; }
                            }
                            post[k] = closest;
                            ids[k] = id | (idp << 8);
                            n_o = n_o - sphere_center(h, tt0); // march in the SDF's frame (extension; zero origin in the reference)
                            og[k] = n_o; dr[k] = n_d; Pk[k] = n_P; en[k] = n_ent; cnt[k] = BC_FIRST; t[k] = 0.0f; act[k] = true;
                            jobs[lane + 64u * k] = make_float4(n_o.x, n_o.y, n_o.z, __uint_as_float(BJ_POINT));
                        }
                    }
                    cur += min((uint32_t)__popcll(need), avail);
                }
            }
        }
        uint32_t total = 0;
#pragma unroll
        for (uint32_t k = 0; k < K; k++) {
            if (!act[k]) jobs[lane + 64u * k] = make_float4(0.0f, 0.0f, 0.0f, __uint_as_float(BJ_INVALID));
            total += (uint32_t)__popcll(__ballot(act[k]));
        }
        if (total == 0) { if (exhausted) break; continue; } // (a fetch may have met INVALID entries only)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- B: orbits (see k_shadow_bulb)
        {
            const uint32_t leave = min(ORBIT_MIN, total >> 2);
            uint32_t next = 0;
            for (;;) {
                const uint64_t need = __ballot(!o_valid);
                if (need != 0 && next < 64u * K) {
                    if (!o_valid) {
                        const uint32_t jn = next + mbcnt(need);
                        if (jn < 64u * K) {
                            const float4 j = jobs[jn];
                            if (__float_as_uint(j.w) == BJ_POINT) {
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
                        if (o.m > BULB_BAILOUT || o_it == iterations) {
                            jobs[my] = make_float4(o.m, o.dz, __uint_as_float(o_it), __uint_as_float(BJ_RESULT));
                            o_valid = false;
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // ---- C: distance + one step of TracedSDF::hit (src/sdf.rs:59-83) for ray k of every lane
        auto epilogue = [&](const uint32_t k) {
            if (__ballot(act[k]) == 0) return;
            if (COUNT) n_epi_passes++;
            float4 r = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (act[k]) r = jobs[lane + 64u * k];
            if (act[k] && __float_as_uint(r.w) == BJ_RESULT) {
                const float dist = bulb_finish(r.x, r.y);
                if (COUNT) { evals.n++; evals.it += __float_as_uint(r.z); }
                bool nan = (cnt[k] & BC_NAN) != 0;
                bool done;
                if (cnt[k] & BC_FIRST) { t[k] = dist; nan = dist != dist; cnt[k] = nan ? BC_NAN : 0u; done = max_marches == 0; }
                else {
                    const bool hit = __builtin_fabsf(dist) < fmaxs(c0, c1 * thr_at(th, t[k]));
                    const bool gt = t[k] > pre[k];
                    done = hit || nan || gt;
                    if (!done) { t[k] = t[k] + dist; cnt[k]++; done = (cnt[k] & BC_COUNT) == max_marches; }
                }
                if (done) {
                    float closest = pre[k];
                    uint32_t id = ids[k] & 0xFFu;
                    if (t[k] < closest) { closest = t[k]; id = ks; }
                    const uint32_t idp = ids[k] >> 8;
                    if (idp != OBJ_NONE && post[k] < closest) { closest = post[k]; id = idp; }
                    pool.geo1[Pk[k]].z = closest;
                    ((uint8_t*)&pool.geo1[Pk[k]].w)[0] = (uint8_t)id; // low byte of the bits word = hit object
                    ent_obj[en[k]] = (uint8_t)id;
                    act[k] = false;
                } else {
                    const f3 pn = muladd3(dr[k], t[k], og[k]);
                    jobs[lane + 64u * k] = make_float4(pn.x, pn.y, pn.z, __uint_as_float(BJ_POINT));
                }
            }
        };
        epilogue(0);
        if (K > 1) epilogue(1);
        if (K > 2) epilogue(K > 2 ? 2 : 0);
    }
    if (COUNT && evals.n) { atomicAdd(evals_out, (unsigned long long)evals.n); atomicAdd(evals_out + 4, (unsigned long long)evals.it); }
    if (COUNT && lane == 0) { atomicAdd(evals_out + 9, 64ull * n_orbit_trips); atomicAdd(evals_out + 10, 64ull * n_epi_passes); } // evals_out = &evals[0]: [9], [10]
}
