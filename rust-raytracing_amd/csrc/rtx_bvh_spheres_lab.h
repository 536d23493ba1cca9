// rtx_bvh_spheres_lab.h -- the experiments on the sphere path that lost (profiles/LAB_NOTEBOOK.md R3.2, R3.4, R3.7): stage 2 as a
// wave-local pool of ray slots, stage 2 with two rays per lane, a counting sort of the survivors between the stages -- and round 4's
// stage 2 over ray slots in LDS (R4.3).  All of them
// render the product's bits.  Compiled only with -DRTX_LAB (librtx_hip_lab.so), included by rtx_bvh_spheres.hip inside namespace
// rtx, after the product kernels whose types and constants it uses (SphQueue, SphSurvivor, kSphStack, the walks of rtx_traverse.h).
#pragma once

// ---- stage 2 as a wave-local pool -----------------------------------------------------------------------------------------
// Stage 2's lock-step form (MODE 2 above) issues 2.6 lane-slots per lane-instruction it needs: every round of a wave lasts
// as long as the longest of its 64 walks (21 node visits on average, ~55 for the slowest lane: lane utilisation 0.39,
// profiles/r03_bench_n1.json).  Refilling a lane the moment its walk ends needs a segment that is READY to walk, and making one
// ready is the f64 phase (exact tests, ray_hit, set-up: ~1300 wave-instructions whether 1 or 64 lanes take part) -- which is
// why the schedules that served waiting lanes in small groups lost.  Here the two are decoupled INSIDE the wave:
//   * a wave owns kPoolSlots = 128 ray slots in device memory (f64 path state, the walk's f32 parameters, the candidates);
//     a slot is READY (set up, waiting for a lane), WALKING (a lane owns it), DONE (its walk ended, candidates stored) or dead;
//   * every iteration idle lanes take READY slots (ballot + mbcnt over a wave-local list in LDS: no atomics) and all walking
//     lanes do one sphere_step -- the walk runs with (almost) all lanes;
//   * when 64 slots are DONE (or the lanes starve) the whole wave runs ONE f64 phase over 64 DONE slots -- lane i serves slot
//     done[i], not the ray it is walking --: exact tests, ray_hit, then the next segment's set-up (-> READY), or the sample
//     store and a fresh survivor from stage 1's queue into the same slot.  The f64 phase runs with all lanes.
// Same functions in the same order per ray: same bits.  No barrier, no atomic besides the queue chunk grab; every
// iteration either walks, or consumes DONE slots, or ends the wave.
constexpr int kPoolSlots = 128;
constexpr int kPoolStack = kSphStack - 1;         // LDS stack entries per lane (one row less than MODE 2: the row pays for the lists)
constexpr uint32_t kPoolFresh = 1u << 10, kPoolNoWalk = 1u << 9, kPoolOverflow = 1u << 8;
#ifndef RTX_POOL_SERVE
#define RTX_POOL_SERVE 24
#endif
#ifndef RTX_POOL_WAIT
#define RTX_POOL_WAIT 8
#endif
constexpr uint32_t kPoolWait = RTX_POOL_WAIT;      // lanes whose walk has ended wait until this many have, then they are served together
constexpr uint32_t kPoolServe = RTX_POOL_SERVE;    // idle lanes (with nothing READY) that trigger an f64 phase before 64 slots are DONE
constexpr int kPoolF64 = 13, kPoolU32 = 28;       // fields per slot
struct SphPool {
    double *f;              // field k of slot i: f[k * stride + i]   (0-2 pos, 3-5 dir, 6-8 result, 9-11 light, 12 rng key)
    uint32_t *u;            // 0 ridx, 1 bounce, 2 qcnt | flags, 3-6 candidate index, 7-10 candidate t_lo, 11 best_up,
    size_t stride;          // 12-17 Ray32 ix iy iz nx ny nz, 18 slack, 19-27 SphereRay px py pz dx dy dz Kg c0 K
};

size_t bvh_spheres_pool2_bytes(int n_cus)
{
    const size_t slots = (size_t)n_cus * kSphWavesPerSimd * (kBvhThreads / 64) * kPoolSlots;
    return slots * (kPoolF64 * sizeof(double) + kPoolU32 * sizeof(uint32_t)) + 512;
}

template <bool SPILL, bool Q3>
__global__ __launch_bounds__(kBvhThreads, kSphWavesPerSimd) void trace_sph_pool_kernel(const SceneView *__restrict__ svp,
                                                                              const RowsView *__restrict__ rvp,
                                                                              double *__restrict__ samples, Counters *__restrict__ ctr,
                                                                              unsigned long long *__restrict__ work_counter,
                                                                              const float4 *__restrict__ nodes, const LeafArrays la,
                                                                              uint32_t *__restrict__ spill, uint32_t spill_entries,
                                                                              const SphQueue sq, const SphPool pool)
{
    constexpr int STACK = kPoolStack;
    const SceneView &sv = *svp;
    const RowsView &rv = *rvp;
    __shared__ uint32_t lds_stack[STACK + 1][kBvhThreads];
    __shared__ uint32_t lds_q[2 * kSphQueue][kBvhThreads];
    __shared__ uint8_t lds_lists[kBvhThreads >> 6][2][kPoolSlots];     // per wave: [0] DONE slots, [1] READY slots
    uint32_t *const lq = &lds_q[0][0];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    uint8_t *const done_list = &lds_lists[tid >> 6][0][0], *const ready_list = &lds_lists[tid >> 6][1][0];
    const size_t spill_stride = (size_t)gridDim.x * kBvhThreads, glane = (size_t)blockIdx.x * kBvhThreads + tid;
    const size_t base = ((size_t)blockIdx.x * (kBvhThreads >> 6) + (tid >> 6)) * kPoolSlots;     // this wave's slots
    const uint32_t bounce_limit = sv.max_bounces >= 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)sv.max_bounces + 1u;
    const unsigned long long n_rays = *sq.count < sq.capacity ? *sq.count : sq.capacity;
    // (volatile: a slot is written by one lane and read by another of the same wave later on; the accesses must reach the
    //  memory system as written, in order, and not be kept in registers or a stale line)
    volatile double *const pf = pool.f;
    volatile uint32_t *const pu = pool.u;
    const size_t ps = pool.stride;

    unsigned long long wave_next = 0, wave_end = 0;
    bool queue_empty = false;
    uint32_t n_done = kPoolSlots, n_ready = 0;                         // wave-uniform
    for (uint32_t s = lane; s < (uint32_t)kPoolSlots; s += 64u) {      // every slot starts DONE + FRESH: the first f64 phases fill the pool
        done_list[s] = (uint8_t)s;
        pu[2 * ps + base + s] = kPoolFresh;
    }
    bool walking = false;
    uint32_t slot = 0, node = kNone, sp = 0, qcnt = 0, nbox = 0, nleaf = 0;
    bool overflow = false;
    float best_up = 0.f;
    Ray32S q;
    SphereRay sr;
    q.ix = q.iy = q.iz = 1.f; q.nx = q.ny = q.nz = q.e = 0.f;
    sr.px = sr.py = sr.pz = sr.dx = sr.dy = sr.dz = sr.Kg = sr.K = 0.f; sr.c0 = __builtin_inff();
    unsigned long long segs = 0, box_tests = 0, leaf_filters = 0, exact = 0;

    for (;;) {
        // ---- idle lanes take READY slots (the list's tail)
        unsigned long long idle_mask = __ballot(!walking);
        uint32_t n_idle = (uint32_t)__popcll(idle_mask);
        if (n_idle != 0u && n_ready != 0u) {
            const uint32_t k = bvh_mbcnt(idle_mask);
            if (!walking && k < n_ready) {
                slot = ready_list[n_ready - 1u - k];
                const size_t i = base + slot;
                q.ix = __uint_as_float(pu[12 * ps + i]); q.iy = __uint_as_float(pu[13 * ps + i]); q.iz = __uint_as_float(pu[14 * ps + i]);
                q.nx = __uint_as_float(pu[15 * ps + i]); q.ny = __uint_as_float(pu[16 * ps + i]); q.nz = __uint_as_float(pu[17 * ps + i]);
                q.e = __uint_as_float(pu[18 * ps + i]);
                sr.px = __uint_as_float(pu[19 * ps + i]); sr.py = __uint_as_float(pu[20 * ps + i]); sr.pz = __uint_as_float(pu[21 * ps + i]);
                sr.dx = __uint_as_float(pu[22 * ps + i]); sr.dy = __uint_as_float(pu[23 * ps + i]); sr.dz = __uint_as_float(pu[24 * ps + i]);
                sr.Kg = __uint_as_float(pu[25 * ps + i]); sr.c0 = __uint_as_float(pu[26 * ps + i]); sr.K = __uint_as_float(pu[27 * ps + i]);
                node = sv.bvh_root; sp = 0; qcnt = 0; overflow = false; best_up = __builtin_inff();
                walking = true;
            }
            n_ready -= n_idle < n_ready ? n_idle : n_ready;
            idle_mask = __ballot(!walking);
            n_idle = (uint32_t)__popcll(idle_mask);
        }
        // ---- the f64 phase, for up to 64 DONE slots, when it runs full -- or the walk is starving
        if (n_done >= 64u || (n_done != 0u && n_ready == 0u && (n_idle == 64u || n_idle >= kPoolServe))) {
            const uint32_t take = n_done < 64u ? n_done : 64u;
            const bool have = lane < take;
            const uint32_t my = have ? (uint32_t)done_list[n_done - 1u - lane] : 0u;
            n_done -= take;
            const size_t i = base + my;
            uint32_t fl = have ? pu[2 * ps + i] : 0u;
            RayState r;
            uint32_t ridx = 0;
            bool go = false;                                            // this slot has a segment to set up
            // (a) a slot whose walk ended: closest_object's exact part + ray_hit
            if (have && (fl & kPoolFresh) == 0u) {
                r.pos = mk(pf[0 * ps + i], pf[1 * ps + i], pf[2 * ps + i]);
                r.dir = mk(pf[3 * ps + i], pf[4 * ps + i], pf[5 * ps + i]);
                r.result = mk(pf[6 * ps + i], pf[7 * ps + i], pf[8 * ps + i]);
                r.light = mk(pf[9 * ps + i], pf[10 * ps + i], pf[11 * ps + i]);
                r.key = (uint64_t)__double_as_longlong(pf[12 * ps + i]);
                ridx = pu[0 * ps + i];
                r.bounce = pu[1 * ps + i];
                r.draw = 6u + 2u * r.bounce;
                const float bu = __uint_as_float(pu[11 * ps + i]);
                const RayX rx = make_rayx(r.pos, r.dir);
                Hit h;
                hit_init(h);
                ++segs;
                if ((fl & (kPoolOverflow | kPoolNoWalk)) == 0u) {
                    const uint32_t nq = fl & 0xFFu;
#pragma unroll 1
                    for (uint32_t e = 0; e < nq; ++e) {
                        if (__uint_as_float(pu[(7 + e) * ps + i]) <= bu) {
                            const uint32_t idx = pu[(3 + e) * ps + i];
                            double t;
                            if (sphere_distance(la.spheres[idx], rx, &t)) hit_consider(h, t, la.sphere_ids[idx], 0, idx);
                            exact += 1;
                        }
                    }
                } else {                                  // no walk (origin out of range / NaN) or a dropped candidate: every sphere
                    for (uint32_t k = 0; k < sv.n_spheres; ++k) {
                        double t;
                        if (sphere_distance(la.spheres[k], rx, &t)) hit_consider(h, t, la.sphere_ids[k], 0, k);
                    }
                    exact += sv.n_spheres;
                }
                for (uint32_t k = 0; k < sv.n_planes; ++k) {
                    double t;
                    if (plane_distance(sv.planes[k], rx, &t)) hit_consider(h, t, sv.planes[k].id, 1, k);
                }
                for (uint32_t k = 0; k < sv.n_tri_filter; ++k) {       // the few triangles of a sphere scene (none of them in the tree)
                    const uint32_t tk = la.tri_fidx[k];
                    double t;
                    if (triangle_distance(la.tris[tk], rx, &t)) hit_consider(h, t, la.tris[tk].id, 2, tk);
                }
                exact += sv.n_planes + sv.n_tri_filter;
                bool done = true;
                if (h.id != kNone) {
                    advance_and_shade(sv, h, r);
                    done = (r.bounce >= bounce_limit) || light_is_zero(r);            // scene.rs:227-228
                }
                if (done) {
                    store_sample(samples, rv, ridx, r.result);
                    fl = kPoolFresh;                                    // the slot is free for the next survivor
                } else go = true;
            }
            // (b) a free slot: the next survivor of stage 1's queue, as it is after its first hit
            const unsigned long long fm = __ballot(have && (fl & kPoolFresh) != 0u);
            bool dead = false;
            if (fm != 0ull) {
                if (wave_next >= wave_end && !queue_empty) {
                    unsigned long long b = 0;
                    if (lane == 0) b = atomicAdd(work_counter, (unsigned long long)rv.grab);
                    b = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
                        __builtin_amdgcn_readfirstlane((uint32_t)b);
                    wave_next = b;
                    wave_end = b + rv.grab < n_rays ? b + rv.grab : n_rays;
                    if (b >= n_rays) { queue_empty = true; wave_next = wave_end = 0; }
                }
                if (have && (fl & kPoolFresh) != 0u) {
                    const unsigned long long rec = wave_next + bvh_mbcnt(fm);
                    if (rec < wave_end) {
                        const double4 *p = reinterpret_cast<const double4 *>(sq.rec + (sq.perm ? (unsigned long long)sq.perm[rec] : rec));
                        const double4 s0 = p[0], s1 = p[1];
                        ridx = (uint32_t)(unsigned long long)__double_as_longlong(s1.z);
                        if (ridx != kNone) {
                            uint32_t pl = 0, smp = 0;
                            if (rv.tiles_x != 0u) (void)ray_index_to_pixel_tiled(rv, ridx, pl, smp);
                            else ray_index_to_pixel(rv, ridx, pl, smp);
                            const uint32_t k = fastdiv(pl, rv.div_width), x = pl - k * rv.width;
                            const uint64_t pix = (uint64_t)image_row(rv, k) * rv.width + x;
                            r.key = rng_key(sv.seed, pix, rv.sample_begin + smp);
                            r.bounce = 1u;
                            r.draw = 8u;
                            r.pos = mk(s0.x, s0.y, s0.z);
                            r.dir = mk(s0.w, s1.x, s1.y);
                            // ray_hit's two folds of the first hit (scene.rs:276-277) from resulting_color = 0, light_color = 1 (ray.rs:18-19)
                            const MaterialX m = sv.materials[(uint32_t)((unsigned long long)__double_as_longlong(s1.z) >> 32)];
                            r.result = vadd(mk(0.0, 0.0, 0.0), vmulv(mk(1.0, 1.0, 1.0), m.emission_color));
                            r.light = vmulv(mk(1.0, 1.0, 1.0), m.base_color);
                            go = true;
                        }                                                // (a slot its wave reserved and did not use: asked again next time)
                    } else if (queue_empty) dead = true;                 // nothing left to take: the slot retires
                }
                const unsigned long long taken = (unsigned long long)__popcll(fm);
                wave_next = wave_next + taken < wave_end ? wave_next + taken : wave_end;
            }
            // (c) the segment's set-up: the walk's f32 parameters; the path state goes back to the slot
            bool nowalk = false;
            if (go) {
                const RayX rn = make_rayx(r.pos, r.dir);
                const float omax = fmaxf(fmaxf(__builtin_fabsf((float)r.pos.x), __builtin_fabsf((float)r.pos.y)),
                                         __builtin_fabsf((float)r.pos.z));
                const bool in32 = omax <= sv.bvh_origin_limit;                              // NaN origin -> no walk
                if (in32 || omax <= sv.bvh_origin_limit * kBvhRange64) {
                    SphereRay s2;
                    sphere_ray_from(sv, r.pos, r.dir, s2);
                    Ray32 q0;
                    make_ray32(r.pos, rn.dirn, (double)sv.bvh_inv_max, q0);
                    pu[12 * ps + i] = __float_as_uint(q0.ix); pu[13 * ps + i] = __float_as_uint(q0.iy); pu[14 * ps + i] = __float_as_uint(q0.iz);
                    pu[15 * ps + i] = __float_as_uint(q0.nx); pu[16 * ps + i] = __float_as_uint(q0.ny); pu[17 * ps + i] = __float_as_uint(q0.nz);
                    pu[18 * ps + i] = __float_as_uint(ray32_slack(q0.nx, q0.ny, q0.nz, in32));
                    pu[19 * ps + i] = __float_as_uint(s2.px); pu[20 * ps + i] = __float_as_uint(s2.py); pu[21 * ps + i] = __float_as_uint(s2.pz);
                    pu[22 * ps + i] = __float_as_uint(s2.dx); pu[23 * ps + i] = __float_as_uint(s2.dy); pu[24 * ps + i] = __float_as_uint(s2.dz);
                    pu[25 * ps + i] = __float_as_uint(s2.Kg); pu[26 * ps + i] = __float_as_uint(s2.c0); pu[27 * ps + i] = __float_as_uint(s2.K);
                } else nowalk = true;
                pf[0 * ps + i] = r.pos.x; pf[1 * ps + i] = r.pos.y; pf[2 * ps + i] = r.pos.z;
                pf[3 * ps + i] = r.dir.x; pf[4 * ps + i] = r.dir.y; pf[5 * ps + i] = r.dir.z;
                pf[6 * ps + i] = r.result.x; pf[7 * ps + i] = r.result.y; pf[8 * ps + i] = r.result.z;
                pf[9 * ps + i] = r.light.x; pf[10 * ps + i] = r.light.y; pf[11 * ps + i] = r.light.z;
                pf[12 * ps + i] = __longlong_as_double((long long)r.key);
                pu[0 * ps + i] = ridx;
                pu[1 * ps + i] = r.bounce;
                if (nowalk) { pu[2 * ps + i] = kPoolNoWalk; pu[11 * ps + i] = __float_as_uint(__builtin_inff()); }
            } else if (have && !dead) {
                pu[2 * ps + i] = kPoolFresh;                            // a free slot that got no survivor this time
            }
            // READY: set up and walkable.  DONE again: no walk possible (tested exhaustively next phase), or still free.
            const bool to_ready = go && !nowalk, to_done = have && !dead && !to_ready;
            const unsigned long long rm = __ballot(to_ready), dm = __ballot(to_done);
            if (to_ready) ready_list[n_ready + bvh_mbcnt(rm)] = (uint8_t)my;
            if (to_done) done_list[n_done + bvh_mbcnt(dm)] = (uint8_t)my;
            n_ready += (uint32_t)__popcll(rm);
            n_done += (uint32_t)__popcll(dm);
            // free slots while the queue still has chunks are asked again; when it is empty they retired above, so a phase that
            // only re-queued free slots cannot repeat for ever
            continue;
        }
        if (n_idle == 64u) break;                  // nothing walking, nothing READY, nothing DONE: every slot retired
        // ---- the walk: visits for every lane that has a node to open, until kPoolWait lanes have finished their walk (they
        //      are then served together, so that the serve code does not run for two or three lanes on every visit), a READY
        //      slot could be handed to an idle lane (nothing to hand out while the list is empty), or nobody walks any more.
        //      A tight loop of its own: what the f64 phase spills stays outside it.
        for (;;) {
            if (walking && node != kNone) {
                if constexpr (Q3)
                    sphere_step_q3<STACK, SPILL>(nodes, la.sphere_f32, la.sphere_prims, q, sr, node, sp, &lds_stack[0][0], lq, tid, spill,
                                                 spill_entries, spill_stride, glane, best_up, qcnt, overflow, nbox, nleaf);
                else
                    sphere_step<STACK, SPILL>(nodes, la.sphere_f32, la.sphere_prims, q, sr, node, sp, &lds_stack[0][0], lq, tid, spill,
                                              spill_entries, spill_stride, glane, best_up, qcnt, overflow, nbox, nleaf);
            }
            const uint32_t n_fin = (uint32_t)__popcll(__ballot(walking && node == kNone));
            if (n_fin >= kPoolWait || __ballot(walking && node != kNone) == 0ull) break;
        }
        const bool fin = walking && node == kNone;
        const unsigned long long fmask = __ballot(fin);
        if (fin) {                                                       // the candidates that can still be the winner go to the slot
            const size_t i = base + slot;
#pragma unroll
            for (int e = 0; e < kSphQueue; ++e) {
                if ((uint32_t)e < qcnt) {
                    pu[(3 + e) * ps + i] = lq[(size_t)e * kBvhThreads + tid];
                    pu[(7 + e) * ps + i] = lq[(size_t)(kSphQueue + e) * kBvhThreads + tid];
                }
            }
            pu[11 * ps + i] = __float_as_uint(best_up);
            pu[2 * ps + i] = qcnt | (overflow ? kPoolOverflow : 0u);
            done_list[n_done + bvh_mbcnt(fmask)] = (uint8_t)slot;
            box_tests += nbox; leaf_filters += nleaf;
            nbox = 0; nleaf = 0;
            walking = false;
        }
        n_done += (uint32_t)__popcll(fmask);
    }
    unsigned long long filt = box_tests + leaf_filters;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        segs += __shfl_xor(segs, off, 64);
        exact += __shfl_xor(exact, off, 64);
        filt += __shfl_xor(filt, off, 64);
        box_tests += __shfl_xor(box_tests, off, 64);
    }
    if (lane == 0) {
        const uint32_t shard = (blockIdx.x * (kBvhThreads >> 6) + (tid >> 6)) & (kCounterShards - 1);
        if (segs) atomicAdd(&ctr[shard].segments, segs);
        if (exact) atomicAdd(&ctr[shard].exact_tests, exact);
        if (filt) atomicAdd(&ctr[shard].filter_tests, filt);
        if (box_tests) atomicAdd(&ctr[2 + (shard % (kCounterShards - 2))].pad_, box_tests);
    }
}

// ---- stage 2 with two rays per lane ----------------------------------------------------------------------------------------
// The pool above pays a round trip to memory whenever a lane changes rays.  This form keeps the hand-over in REGISTERS: every
// lane owns two rays.  While it walks one, the other is either DONE (its walk ended: candidates in 10 registers, waiting for
// the f64 phase) or in the lane's POCKET (set up: the walk's 16 f32 parameters in registers, ready to go).  A lane whose walk
// ends takes its pocket ray on the spot -- no memory access, no waiting -- and the wave runs ONE f64 phase for all lanes that
// hold a DONE ray once kPairServe of them do (or nobody can walk): exact tests, ray_hit, the next segment's set-up into the
// pocket (or the sample store and a fresh survivor from stage 1's queue).  Only the f64 path state of the two rays lives in
// memory, lane-private and coalesced (slot s of lane l of wave w: field[(2 w + s) * 64 + l]), read and written once per segment
// by the f64 phase.  Same functions in the same order per ray: same bits.
#ifndef RTX_PAIR_SERVE
#define RTX_PAIR_SERVE 56
#endif
constexpr uint32_t kPairServe = RTX_PAIR_SERVE;       // lanes holding a DONE ray (or idle lanes) that trigger the f64 phase
#ifndef RTX_PAIR_WAIT
#define RTX_PAIR_WAIT 8
#endif
constexpr uint32_t kPairWait = RTX_PAIR_WAIT;
#ifndef RTX_PAIR_WAVES
#define RTX_PAIR_WAVES 3
#endif
constexpr int kPairWaves = RTX_PAIR_WAVES;         // workgroups per CU
constexpr int kPairF64 = 13;                      // pos, dir, result, light, rng key
constexpr int kPairU32 = 2;                       // ridx, bounce

size_t bvh_spheres_pair_bytes(int n_cus)
{
    const size_t lanes2 = (size_t)n_cus * kSphWavesPerSimd * kBvhThreads * 2;
    return lanes2 * (kPairF64 * sizeof(double) + kPairU32 * sizeof(uint32_t)) + 512;
}

struct SphPair { double *f; uint32_t *u; size_t stride; };

template <bool SPILL, bool Q3>
__global__ __launch_bounds__(kBvhThreads, kPairWaves) void trace_sph_pair_kernel(const SceneView *__restrict__ svp,
                                                                              const RowsView *__restrict__ rvp,
                                                                              double *__restrict__ samples, Counters *__restrict__ ctr,
                                                                              unsigned long long *__restrict__ work_counter,
                                                                              const float4 *__restrict__ nodes, const LeafArrays la,
                                                                              uint32_t *__restrict__ spill, uint32_t spill_entries,
                                                                              const SphQueue sq, const SphPair pp)
{
    constexpr int STACK = kSphStack;
    const SceneView &sv = *svp;
    const RowsView &rv = *rvp;
    __shared__ uint32_t lds_stack[STACK + 1][kBvhThreads];
    __shared__ uint32_t lds_q[2 * kSphQueue][kBvhThreads];
    __shared__ uint32_t lds_dbuf[2 * kSphQueue + 2][kBvhThreads];        // a DONE ray's candidates, flags and bound (per lane)
    uint32_t *const lq = &lds_q[0][0];
    uint32_t *const ld = &lds_dbuf[0][0];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const size_t spill_stride = (size_t)gridDim.x * kBvhThreads, glane = (size_t)blockIdx.x * kBvhThreads + tid;
    const size_t wave = (size_t)blockIdx.x * (kBvhThreads >> 6) + (tid >> 6);
    const uint32_t bounce_limit = sv.max_bounces >= 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)sv.max_bounces + 1u;
    const unsigned long long n_rays = *sq.count < sq.capacity ? *sq.count : sq.capacity;
    double *const pf = pp.f;
    uint32_t *const pu = pp.u;
    const size_t ps = pp.stride;

    unsigned long long wave_next = 0, wave_end = 0;
    bool queue_empty = false;
    // the ray being walked
    bool walking = false;
    uint32_t ws = 0, node = kNone, sp = 0, qcnt = 0, nbox = 0, nleaf = 0;
    bool overflow = false;
    float best_up = 0.f;
    Ray32S q;
    SphereRay sr;
    q.ix = q.iy = q.iz = 1.f; q.nx = q.ny = q.nz = q.e = 0.f;
    sr.px = sr.py = sr.pz = sr.dx = sr.dy = sr.dz = sr.Kg = sr.K = 0.f; sr.c0 = __builtin_inff();
    // the pocket: the lane's other ray, set up (slot 1 - ws while the lane walks)
    bool pocket = false, pocket_nowalk = false;
    uint32_t pslot = 0;
    Ray32S pq = q;
    SphereRay psr = sr;
    // DONE rays: a finished walk's candidates, flags and bound move to the lane's column of lds_dbuf (no memory access in the
    // hand-over); when the lane's other ray is DONE already they stay where they are, in the walk's LDS queue, and the lane
    // waits for the f64 phase
    bool have_done = false, lds_done = false;
    uint32_t dslot = 0, lslot = 0, lflags = 0;
    float lbu = 0.f;
    uint32_t free_slots = 3u;                         // bit s: slot s of this lane holds no ray
    uint32_t segs = 0, box_tests = 0, leaf_filters = 0, exact = 0;        // (per lane: far below 2^32)

    // the hand-over, inside the walk loop and outside it: a lane whose walk has ended parks its candidates (registers, or --
    // when its other ray is DONE already -- where they are, in its LDS queue), and a lane that does not walk takes its pocket
#define RTX_PAIR_HANDOVER()                                                                                          \
    {                                                                                                                \
        if (walking && node == kNone) {                                                                              \
            walking = false;                                                                                         \
            box_tests += nbox; leaf_filters += nleaf;                                                                \
            nbox = 0; nleaf = 0;                                                                                     \
            const uint32_t fl_ = qcnt | (overflow ? kPoolOverflow : 0u);                                             \
            if (!have_done) {                                                                                        \
                _Pragma("unroll") for (int e = 0; e < 2 * kSphQueue; ++e)                                            \
                    ld[(size_t)e * kBvhThreads + tid] = lq[(size_t)e * kBvhThreads + tid];                           \
                ld[(size_t)(2 * kSphQueue) * kBvhThreads + tid] = fl_;                                               \
                ld[(size_t)(2 * kSphQueue + 1) * kBvhThreads + tid] = __float_as_uint(best_up);                      \
                dslot = ws;                                                                                          \
                have_done = true;                                                                                    \
            } else {                                                                                                 \
                lslot = ws; lflags = fl_; lbu = best_up;                                                             \
                lds_done = true;                                                                                     \
            }                                                                                                        \
        }                                                                                                            \
        if (!walking && pocket && !lds_done) {                                                                       \
            pocket = false;                                                                                          \
            ws = pslot;                                                                                              \
            if (pocket_nowalk) { /* no f32 walk for this origin: DONE at once, every sphere gets the exact test */   \
                if (!have_done) {                                                                                    \
                    ld[(size_t)(2 * kSphQueue) * kBvhThreads + tid] = kPoolNoWalk;                                   \
                    ld[(size_t)(2 * kSphQueue + 1) * kBvhThreads + tid] = __float_as_uint(__builtin_inff());         \
                    dslot = ws; have_done = true;                                                                    \
                } else { lds_done = true; lslot = ws; lflags = kPoolNoWalk; lbu = __builtin_inff(); }                \
            } else {                                                                                                 \
                q = pq; sr = psr;                                                                                    \
                node = sv.bvh_root; sp = 0; qcnt = 0; overflow = false; best_up = __builtin_inff();                  \
                walking = true;                                                                                      \
            }                                                                                                        \
        }                                                                                                            \
    }

    for (;;) {
        RTX_PAIR_HANDOVER()
        uint32_t n_walk = (uint32_t)__popcll(__ballot(walking));
        uint32_t n_done = (uint32_t)__popcll(__ballot(have_done));
        const uint32_t n_fresh = (uint32_t)__popcll(__ballot(free_slots != 0u && !queue_empty));   // lanes that would take a survivor
        if (n_walk == 0u && n_done == 0u && n_fresh == 0u) break;       // nobody walks, nothing DONE, nothing to fetch, no pocket left
        // the f64 phase is due when kPairServe lanes hold a DONE ray, when that many lanes are idle and there is anything for
        // it to do, or when nobody walks
#define RTX_PAIR_DUE() (n_walk == 0u || n_done >= kPairServe || (64u - n_walk >= kPairServe && n_done + n_fresh != 0u))
        // ---- the walk: visits, with the hand-over in the loop (every kPairWait finished walks), until the f64 phase is due.
        //      What the f64 phase spills is moved once per phase, not once per hand-over.
        if (!RTX_PAIR_DUE()) {
            for (;;) {
                if (walking && node != kNone) {
                    if constexpr (Q3)
                        sphere_step_q3<STACK, SPILL>(nodes, la.sphere_f32, la.sphere_prims, q, sr, node, sp, &lds_stack[0][0], lq, tid, spill,
                                                     spill_entries, spill_stride, glane, best_up, qcnt, overflow, nbox, nleaf);
                    else
                        sphere_step<STACK, SPILL>(nodes, la.sphere_f32, la.sphere_prims, q, sr, node, sp, &lds_stack[0][0], lq, tid, spill,
                                                  spill_entries, spill_stride, glane, best_up, qcnt, overflow, nbox, nleaf);
                }
                const uint32_t n_fin = (uint32_t)__popcll(__ballot(walking && node == kNone));
                if (n_fin >= kPairWait || __ballot(walking && node != kNone) == 0ull) {
                    RTX_PAIR_HANDOVER()
                    n_walk = (uint32_t)__popcll(__ballot(walking));
                    n_done = (uint32_t)__popcll(__ballot(have_done));
                    if (RTX_PAIR_DUE()) break;
                }
            }
        }
#undef RTX_PAIR_DUE
        // ---- the f64 phase: for the lanes that hold a DONE ray (or an empty slot while the queue has survivors)
        {
            RayState r;
            uint32_t ridx = 0, slot = 0;
            bool go = false;
            if (have_done) {
                slot = dslot;
                have_done = false;
                const size_t i = (wave * 2 + slot) * 64 + lane;
                const uint32_t dflags = ld[(size_t)(2 * kSphQueue) * kBvhThreads + tid];
                const float dbu = __uint_as_float(ld[(size_t)(2 * kSphQueue + 1) * kBvhThreads + tid]);
                r.pos = mk(pf[0 * ps + i], pf[1 * ps + i], pf[2 * ps + i]);
                r.dir = mk(pf[3 * ps + i], pf[4 * ps + i], pf[5 * ps + i]);
                r.result = mk(pf[6 * ps + i], pf[7 * ps + i], pf[8 * ps + i]);
                r.light = mk(pf[9 * ps + i], pf[10 * ps + i], pf[11 * ps + i]);
                r.key = (uint64_t)__double_as_longlong(pf[12 * ps + i]);
                ridx = pu[0 * ps + i];
                r.bounce = pu[1 * ps + i];
                r.draw = 6u + 2u * r.bounce;
                const RayX rx = make_rayx(r.pos, r.dir);
                Hit h;
                hit_init(h);
                ++segs;
                if ((dflags & (kPoolOverflow | kPoolNoWalk)) == 0u) {
                    const uint32_t nq = dflags & 0xFFu;
#pragma unroll 1
                    for (uint32_t e = 0; e < nq; ++e) {
                        if (__uint_as_float(ld[(size_t)(kSphQueue + e) * kBvhThreads + tid]) <= dbu) {
                            const uint32_t idx = ld[(size_t)e * kBvhThreads + tid];
                            double t;
                            if (sphere_distance(la.spheres[idx], rx, &t)) hit_consider(h, t, la.sphere_ids[idx], 0, idx);
                            exact += 1;
                        }
                    }
                } else {                                  // no walk (origin out of range / NaN) or a dropped candidate: every sphere
                    for (uint32_t k = 0; k < sv.n_spheres; ++k) {
                        double t;
                        if (sphere_distance(la.spheres[k], rx, &t)) hit_consider(h, t, la.sphere_ids[k], 0, k);
                    }
                    exact += sv.n_spheres;
                }
                for (uint32_t k = 0; k < sv.n_planes; ++k) {
                    double t;
                    if (plane_distance(sv.planes[k], rx, &t)) hit_consider(h, t, sv.planes[k].id, 1, k);
                }
                for (uint32_t k = 0; k < sv.n_tri_filter; ++k) {
                    const uint32_t tk = la.tri_fidx[k];
                    double t;
                    if (triangle_distance(la.tris[tk], rx, &t)) hit_consider(h, t, la.tris[tk].id, 2, tk);
                }
                exact += sv.n_planes + sv.n_tri_filter;
                bool done = true;
                if (h.id != kNone) {
                    advance_and_shade(sv, h, r);
                    done = (r.bounce >= bounce_limit) || light_is_zero(r);            // scene.rs:227-228
                }
                if (done) {
                    store_sample(samples, rv, ridx, r.result);
                    free_slots |= 1u << slot;
                } else go = true;
            }
            // an empty slot takes the next survivor of stage 1's queue, as it is after its first hit
            const bool ask = !go && free_slots != 0u && !queue_empty;
            const unsigned long long fm = __ballot(ask);
            if (fm != 0ull) {
                if (wave_next >= wave_end && !queue_empty) {
                    unsigned long long b = 0;
                    if (lane == 0) b = atomicAdd(work_counter, (unsigned long long)rv.grab);
                    b = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
                        __builtin_amdgcn_readfirstlane((uint32_t)b);
                    wave_next = b;
                    wave_end = b + rv.grab < n_rays ? b + rv.grab : n_rays;
                    if (b >= n_rays) { queue_empty = true; wave_next = wave_end = 0; }
                }
                if (ask) {
                    const unsigned long long rec = wave_next + bvh_mbcnt(fm);
                    if (rec < wave_end) {
                        const double4 *p = reinterpret_cast<const double4 *>(sq.rec + (sq.perm ? (unsigned long long)sq.perm[rec] : rec));
                        const double4 s0 = p[0], s1 = p[1];
                        ridx = (uint32_t)(unsigned long long)__double_as_longlong(s1.z);
                        if (ridx != kNone) {
                            slot = (free_slots & 1u) ? 0u : 1u;
                            free_slots &= ~(1u << slot);
                            uint32_t pl = 0, smp = 0;
                            if (rv.tiles_x != 0u) (void)ray_index_to_pixel_tiled(rv, ridx, pl, smp);
                            else ray_index_to_pixel(rv, ridx, pl, smp);
                            const uint32_t k = fastdiv(pl, rv.div_width), x = pl - k * rv.width;
                            const uint64_t pix = (uint64_t)image_row(rv, k) * rv.width + x;
                            r.key = rng_key(sv.seed, pix, rv.sample_begin + smp);
                            r.bounce = 1u;
                            r.draw = 8u;
                            r.pos = mk(s0.x, s0.y, s0.z);
                            r.dir = mk(s0.w, s1.x, s1.y);
                            // ray_hit's two folds of the first hit (scene.rs:276-277) from resulting_color = 0, light_color = 1 (ray.rs:18-19)
                            const MaterialX m = sv.materials[(uint32_t)((unsigned long long)__double_as_longlong(s1.z) >> 32)];
                            r.result = vadd(mk(0.0, 0.0, 0.0), vmulv(mk(1.0, 1.0, 1.0), m.emission_color));
                            r.light = vmulv(mk(1.0, 1.0, 1.0), m.base_color);
                            go = true;
                        }
                    }
                }
                const unsigned long long taken = (unsigned long long)__popcll(fm);
                wave_next = wave_next + taken < wave_end ? wave_next + taken : wave_end;
            }
            // the segment's set-up: path state back to the slot, the walk's parameters into the pocket -- or straight into the walk
            if (go) {
                const size_t i = (wave * 2 + slot) * 64 + lane;
                pf[0 * ps + i] = r.pos.x; pf[1 * ps + i] = r.pos.y; pf[2 * ps + i] = r.pos.z;
                pf[3 * ps + i] = r.dir.x; pf[4 * ps + i] = r.dir.y; pf[5 * ps + i] = r.dir.z;
                pf[6 * ps + i] = r.result.x; pf[7 * ps + i] = r.result.y; pf[8 * ps + i] = r.result.z;
                pf[9 * ps + i] = r.light.x; pf[10 * ps + i] = r.light.y; pf[11 * ps + i] = r.light.z;
                pf[12 * ps + i] = __longlong_as_double((long long)r.key);
                pu[0 * ps + i] = ridx;
                pu[1 * ps + i] = r.bounce;
                const RayX rn = make_rayx(r.pos, r.dir);
                const float omax = fmaxf(fmaxf(__builtin_fabsf((float)r.pos.x), __builtin_fabsf((float)r.pos.y)),
                                         __builtin_fabsf((float)r.pos.z));
                const bool in32 = omax <= sv.bvh_origin_limit;                              // NaN origin -> no walk
                pocket_nowalk = !(in32 || omax <= sv.bvh_origin_limit * kBvhRange64);
                if (!pocket_nowalk) {
                    sphere_ray_from(sv, r.pos, r.dir, psr);
                    Ray32 q0;
                    make_ray32(r.pos, rn.dirn, (double)sv.bvh_inv_max, q0);
                    pq.ix = q0.ix; pq.iy = q0.iy; pq.iz = q0.iz; pq.nx = q0.nx; pq.ny = q0.ny; pq.nz = q0.nz;
                    pq.e = ray32_slack(q0.nx, q0.ny, q0.nz, in32);
                }
                pocket = true;
                pslot = slot;
            }
            // a second DONE ray that waited in the walk's LDS queue becomes the lane's DONE ray (and frees the queue)
            if (lds_done) {
#pragma unroll
                for (int e = 0; e < 2 * kSphQueue; ++e) ld[(size_t)e * kBvhThreads + tid] = lq[(size_t)e * kBvhThreads + tid];
                ld[(size_t)(2 * kSphQueue) * kBvhThreads + tid] = lflags;
                ld[(size_t)(2 * kSphQueue + 1) * kBvhThreads + tid] = __float_as_uint(lbu);
                dslot = lslot;
                have_done = true;
                lds_done = false;
            }
        }
    }
#undef RTX_PAIR_HANDOVER
    unsigned long long wsegs = segs, wexact = exact, wbox = box_tests, wfilt = (unsigned long long)box_tests + leaf_filters;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        wsegs += __shfl_xor(wsegs, off, 64);
        wexact += __shfl_xor(wexact, off, 64);
        wfilt += __shfl_xor(wfilt, off, 64);
        wbox += __shfl_xor(wbox, off, 64);
    }
    if (lane == 0) {
        const uint32_t shard = (blockIdx.x * (kBvhThreads >> 6) + (tid >> 6)) & (kCounterShards - 1);
        if (wsegs) atomicAdd(&ctr[shard].segments, wsegs);
        if (wexact) atomicAdd(&ctr[shard].exact_tests, wexact);
        if (wfilt) atomicAdd(&ctr[shard].filter_tests, wfilt);
        if (wbox) atomicAdd(&ctr[2 + (shard % (kCounterShards - 2))].pad_, wbox);
    }
}

// ---- ordering the survivors for stage 2 ----------------------------------------------------------------------------------
// A wave's round lasts as long as its longest walk, and how long a walk is depends mostly on how far the ray travels inside
// the cloud.  The survivors are therefore binned by the distance at which their ray leaves the scene's box (kSortT bins) and
// the octant of their direction, with a counting sort over the queue: histogram (keys kept), scan, scatter of the record
// indices.  Within a bin the queue's tile order is kept up to the arrival order of the workgroups.
constexpr int kSortT = 32, kSortBins = kSortT * 8;
struct SphSort {
    uint8_t *key;               // per queue slot
    uint32_t *perm;             // sorted position -> queue slot
    unsigned int *hist;         // [kSortBins] counts, then running offsets
    float lo[3], hi[3], inv_dt; // the scene's box (sphere centre -+ reach) and kSortT / its diagonal
};

__device__ __forceinline__ uint32_t sph_sort_key(const SphSort &so, const SphSurvivor &r)
{
    if (r.ridx == kNone) return (uint32_t)kSortBins - 1u;                      // a dead slot: to the very end
    const float px = (float)r.px, py = (float)r.py, pz = (float)r.pz, dx = (float)r.dx, dy = (float)r.dy, dz = (float)r.dz;
    const float tx = ((dx > 0.f ? so.hi[0] : so.lo[0]) - px) / dx, ty = ((dy > 0.f ? so.hi[1] : so.lo[1]) - py) / dy,
                tz = ((dz > 0.f ? so.hi[2] : so.lo[2]) - pz) / dz;
    float t = fminf(fminf(tx, ty), tz) * so.inv_dt;                               // (NaN / inf operands: any bin will do)
    t = t >= 0.f ? t : 0.f;
    const uint32_t tb = t < (float)(kSortT - 1) ? (uint32_t)t : (uint32_t)(kSortT - 1);
    const uint32_t oct = (dx < 0.f ? 1u : 0u) | (dy < 0.f ? 2u : 0u) | (dz < 0.f ? 4u : 0u);
#ifndef RTX_SORT_MODE
#define RTX_SORT_MODE 0
#endif
#if RTX_SORT_MODE == 1
    uint32_t k = oct;                                                             // direction octant alone (tile order within it)
    (void)tb;
#elif RTX_SORT_MODE == 2
    // the cell of the ray's origin (2 x 4 x 4 over the scene's box) and the octant
    const float fx = (px - so.lo[0]) / (so.hi[0] - so.lo[0]), fy = (py - so.lo[1]) / (so.hi[1] - so.lo[1]), fz = (pz - so.lo[2]) / (so.hi[2] - so.lo[2]);
    const uint32_t cx = fx > 0.5f ? 1u : 0u, cy = fy <= 0.f ? 0u : (fy >= 1.f ? 3u : (uint32_t)(fy * 4.f)), cz = fz <= 0.f ? 0u : (fz >= 1.f ? 3u : (uint32_t)(fz * 4.f));
    uint32_t k = ((cx * 4u + cy) * 4u + cz) * 8u + oct;
    (void)tb;
#else
    uint32_t k = tb * 8u + oct;
#endif
    return k < (uint32_t)kSortBins - 1u ? k : (uint32_t)kSortBins - 2u;
}

__global__ __launch_bounds__(256) void sph_sort_hist_kernel(const SphQueue sq, const SphSort so)
{
    __shared__ unsigned int h[kSortBins];
    h[threadIdx.x] = 0u;
    __syncthreads();
    const unsigned long long n = *sq.count < sq.capacity ? *sq.count : sq.capacity;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256u + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256u) {
        const uint32_t k = sph_sort_key(so, sq.rec[i]);
        so.key[i] = (uint8_t)k;
        atomicAdd(&h[k], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&so.hist[threadIdx.x], h[threadIdx.x]);
}

__global__ __launch_bounds__(256) void sph_sort_scan_kernel(const SphSort so)          // one workgroup: counts -> first positions
{
    __shared__ unsigned int h[kSortBins];
    h[threadIdx.x] = so.hist[threadIdx.x];
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int run = 0;
        for (int k = 0; k < kSortBins; ++k) { const unsigned int c = h[k]; h[k] = run; run += c; }
    }
    __syncthreads();
    so.hist[threadIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(256) void sph_sort_scatter_kernel(const SphQueue sq, const SphSort so)
{
    __shared__ unsigned int cnt[kSortBins], base[kSortBins];
    const unsigned long long n = *sq.count < sq.capacity ? *sq.count : sq.capacity;
    // a workgroup takes chunks of 4096 consecutive slots: one reservation per (chunk, bin)
    constexpr unsigned long long kChunk = 4096;
    for (unsigned long long c0 = (unsigned long long)blockIdx.x * kChunk; c0 < n; c0 += (unsigned long long)gridDim.x * kChunk) {
        cnt[threadIdx.x] = 0u;
        __syncthreads();
        uint32_t rank[kChunk / 256], key[kChunk / 256];
#pragma unroll
        for (int j = 0; j < (int)(kChunk / 256); ++j) {
            const unsigned long long i = c0 + (unsigned long long)j * 256u + threadIdx.x;
            key[j] = i < n ? (uint32_t)so.key[i] : 0xFFFFFFFFu;
            rank[j] = key[j] != 0xFFFFFFFFu ? atomicAdd(&cnt[key[j]], 1u) : 0u;
        }
        __syncthreads();
        base[threadIdx.x] = cnt[threadIdx.x] ? atomicAdd(&so.hist[threadIdx.x], cnt[threadIdx.x]) : 0u;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < (int)(kChunk / 256); ++j) {
            const unsigned long long i = c0 + (unsigned long long)j * 256u + threadIdx.x;
            if (key[j] != 0xFFFFFFFFu) so.perm[base[key[j]] + rank[j]] = (uint32_t)i;
        }
        __syncthreads();
    }
}


// ---- stage 2 over ray slots (round 4, LAB_NOTEBOOK R4.3: built, bit-identical, on par with the lock-step form -- not shipped) ---------
// The lock-step stage 2 above gives a lane ONE ray for the ray's whole life, so a round of its wave lasts as long as the round's
// longest walk: node visits run with 37 of 64 lanes, the f64 phase with 49 (LAB_NOTEBOOK R3.12).  Round 3 tried twice to hand a
// lane another ray the moment its walk ends (R3.4: a pool of slots in device memory, R3.7: two rays per lane); both cut the
// instructions by a third and lost the time again to memory latency -- every hand-over was a round trip to a 113 MB pool, or cost
// the registers that hide latency.  What was missing is LDS, and the LDS was there all along: the 30-entry stack holds at most 9
// entries on C2 (0 of 3.7e7 walks went deeper, 0.6 % beyond 7: R4.2), so 20 KB of every workgroup's 39 KB were never touched.
//
// Here a wave owns kSlotN ray slots.  A slot is
//   READY    set up: the 16 f32 walk parameters (Ray32S + SphereRay) are in the slot's LDS words, waiting for a lane
//   WALKING  a lane has taken the parameters into registers and walks; the slot's LDS words now hold its candidate queue
//   DONE     the walk has ended: candidates, best_up and flags in the slot's LDS words, waiting for the f64 phase
//   FREE     (a DONE slot flagged fresh) no ray: the next f64 phase puts a survivor of stage 1's queue into it
// * a lane whose walk ends leaves its slot DONE and takes a READY one -- LDS reads only, served for kSlotWait lanes together so
//   that the hand-over code is not issued for one lane at a time;
// * when 64 slots are DONE (or the walk starves) the WAVE runs one f64 phase over 64 DONE slots with all lanes -- lane i
//   serves slot done[i], not the ray it walks: exact tests, ray_hit, the next segment's set-up (-> READY), or the sample store;
// * the f64 path state of a slot (pos, dir, result, light, RNG key, indices: one 128-byte record) lives in device memory,
//   read and written once per segment by the f64 phase -- the lock-step kernel moved as much through its scratch spills.
// The walk is the phased walk of rtx_traverse.h (node visits and leaf visits apart) over the 64-byte nodes, per lane, with the
// same pruning: same candidates up to the order best_up tightens in, same exact tests, same bits.
#ifndef RTX_SLOT_N
#define RTX_SLOT_N 88
#endif
#ifndef RTX_SLOT_TOP
#define RTX_SLOT_TOP 85
#endif
#ifndef RTX_SLOT_STACK
#define RTX_SLOT_STACK 10
#endif
#ifndef RTX_SLOT_WAIT
#define RTX_SLOT_WAIT 12
#endif
#ifndef RTX_SLOT_SERVE
#define RTX_SLOT_SERVE 24
#endif
constexpr int kSlotN = RTX_SLOT_N;                    // ray slots per wave
constexpr int kSlotStack = RTX_SLOT_STACK;            // LDS stack entries per lane (+ the sink row); deeper walks use the HBM column
constexpr int kSlotWords = 16;                        // LDS words per slot
constexpr int kSlotTop = RTX_SLOT_TOP;                // nodes [0, kSlotTop) of the tree (its first levels) are copied into LDS, per workgroup
constexpr uint32_t kSlotWait = RTX_SLOT_WAIT;         // lanes whose walk has ended before they are served together
constexpr uint32_t kSlotServe = RTX_SLOT_SERVE;       // idle lanes (nothing READY) that trigger an f64 phase before 64 slots are DONE
constexpr uint32_t kSlotFresh = 1u << 10, kSlotNoWalk = 1u << 9, kSlotOverflow = 1u << 8;
static_assert(kSlotN >= 64 && kSlotN <= 255 && kSlotN % 4 == 0, "slot ids are bytes; the first f64 phase fills 64 slots");
struct SlotRec { double4 a, b, c, d; };               // {pos.xyz, dir.x} {dir.yz, result.xy} {result.z, light.xyz} {key, ridx | bounce << 32, -, -}
static_assert(sizeof(SlotRec) == 128, "one slot record = one 128-byte line");

size_t bvh_spheres_slots_bytes(int n_cus)
{
    return (size_t)n_cus * kSphWavesPerSimd * (kBvhThreads / 64) * kSlotN * sizeof(SlotRec) + 256;
}

uint32_t bvh_spheres_slots_spill_entries(const SceneView &sv)
{
    const uint32_t need = 3u * sv.bvh_depth + 2u;
    return need > (uint32_t)kSlotStack ? need - (uint32_t)kSlotStack : 0u;
}

template <bool SPILL>
__global__ __launch_bounds__(kBvhThreads, kSphWavesPerSimd) void trace_sph_slots_kernel(const SceneView *__restrict__ svp,
                                                                               const RowsView *__restrict__ rvp,
                                                                               double *__restrict__ samples, Counters *__restrict__ ctr,
                                                                               unsigned long long *__restrict__ work_counter,
                                                                               const float4 *__restrict__ qnodes, const LeafArrays la,
                                                                               uint32_t *__restrict__ spill, uint32_t spill_entries,
                                                                               const SphQueue sq, SlotRec *__restrict__ pool)
{
    constexpr int STACK = kSlotStack;
    constexpr int WAVES = kBvhThreads >> 6;
    const SceneView &sv = *svp;
    const RowsView &rv = *rvp;
    __shared__ uint32_t lds_stack[STACK + 1][kBvhThreads];                 // + the sink row of the branch-free pushes
    __shared__ uint32_t lds_slot[WAVES][kSlotWords][kSlotN];              // READY: 0-6 Ray32S, 7-15 SphereRay; WALKING / DONE: 0-3 candidate
                                                                           // index, 4-7 its t_lo, 8 best_up, 9 qcnt | flags
    __shared__ uint8_t lds_lists[WAVES][2][kSlotN];                       // per wave: [0] DONE slots (a queue), [1] READY slots (a stack)
    __shared__ float4 lds_top[kSlotTop > 0 ? 4 * kSlotTop : 4];           // the 64-byte nodes [0, top_n): the top of the tree
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t top_n = kSlotTop > 0 ? (sv.n_bvh_nodes < (uint32_t)kSlotTop ? sv.n_bvh_nodes : (uint32_t)kSlotTop) : 0u;
    for (uint32_t k = tid; k < 4u * top_n; k += kBvhThreads) lds_top[k] = qnodes[k];
    __syncthreads();
    uint32_t *const sl = &lds_slot[wv][0][0];                             // word w of slot k: sl[w * kSlotN + k]
    uint8_t *const done_list = &lds_lists[wv][0][0], *const ready_list = &lds_lists[wv][1][0];
    const size_t spill_stride = (size_t)gridDim.x * kBvhThreads, glane = (size_t)blockIdx.x * kBvhThreads + tid;
    SlotRec *const recs = pool + ((size_t)blockIdx.x * WAVES + wv) * kSlotN;      // this wave's records
    const uint32_t bounce_limit = sv.max_bounces >= 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)sv.max_bounces + 1u;
    const unsigned long long n_rays = *sq.count < sq.capacity ? *sq.count : sq.capacity;

    unsigned long long wave_next = 0, wave_end = 0;
    bool queue_empty = false;
    uint32_t n_done = kSlotN, n_ready = 0;                                 // wave-uniform
    for (uint32_t s = lane; s < (uint32_t)kSlotN; s += 64u) {             // every slot starts FREE: the first f64 phases fill them
        done_list[s] = (uint8_t)s;
        sl[9 * kSlotN + s] = kSlotFresh;
    }
    bool walking = false;
    uint32_t slot = 0, node = kNone, sp = 0, qcnt = 0;
    uint32_t nbox = 0, nleaf = 0, segs = 0;                               // per lane and launch: far below 2^32 (the host keeps a launch below 2^32 rays)
    bool overflow = false;
    float best_up = 0.f;
    Ray32S q;
    SphereRay sr;
    q.ix = q.iy = q.iz = 1.f; q.nx = q.ny = q.nz = q.e = 0.f;
    sr.px = sr.py = sr.pz = sr.dx = sr.dy = sr.dz = sr.Kg = sr.K = 0.f; sr.c0 = __builtin_inff();
    unsigned long long exact = 0;                                         // (an exhaustive fallback adds n_spheres at a time)

#ifdef RTX_SLOT_PROFILE                      // lab build: where a wave's time goes (cycles / 64, by lane 0), through exact_tests
    unsigned long long sp_prof = 0;
#define RTX_SLOT_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime();
#define RTX_SLOT_ACC(k, a, b) { if (RTX_SLOT_PROFILE == (k) && lane == 0) sp_prof += ((b) - (a)) >> 6; }
#else
#define RTX_SLOT_T(v)
#define RTX_SLOT_ACC(k, a, b)
#endif
    for (;;) {
        RTX_SLOT_T(t_s0)
        // ---- lanes whose walk has ended leave their slot DONE (its candidates are in its LDS words already)
        {
            const bool fin = walking && node == kNone;
            const unsigned long long fmask = __ballot(fin);
            if (fmask != 0ull) {
                if (fin) {
                    sl[8 * kSlotN + slot] = __float_as_uint(best_up);
                    sl[9 * kSlotN + slot] = qcnt | (overflow ? kSlotOverflow : 0u);
                    done_list[n_done + bvh_mbcnt(fmask)] = (uint8_t)slot;
                    walking = false;
                }
                n_done += (uint32_t)__popcll(fmask);
            }
        }
        // ---- idle lanes take READY slots (the list's tail): 16 LDS words into registers
        unsigned long long idle_mask = __ballot(!walking);
        uint32_t n_idle = (uint32_t)__popcll(idle_mask);
        if (n_idle != 0u && n_ready != 0u) {
            const uint32_t k = bvh_mbcnt(idle_mask);
            if (!walking && k < n_ready) {
                slot = ready_list[n_ready - 1u - k];
                q.ix = __uint_as_float(sl[0 * kSlotN + slot]); q.iy = __uint_as_float(sl[1 * kSlotN + slot]); q.iz = __uint_as_float(sl[2 * kSlotN + slot]);
                q.nx = __uint_as_float(sl[3 * kSlotN + slot]); q.ny = __uint_as_float(sl[4 * kSlotN + slot]); q.nz = __uint_as_float(sl[5 * kSlotN + slot]);
                q.e = __uint_as_float(sl[6 * kSlotN + slot]);
                sr.px = __uint_as_float(sl[7 * kSlotN + slot]); sr.py = __uint_as_float(sl[8 * kSlotN + slot]); sr.pz = __uint_as_float(sl[9 * kSlotN + slot]);
                sr.dx = __uint_as_float(sl[10 * kSlotN + slot]); sr.dy = __uint_as_float(sl[11 * kSlotN + slot]); sr.dz = __uint_as_float(sl[12 * kSlotN + slot]);
                sr.Kg = __uint_as_float(sl[13 * kSlotN + slot]); sr.c0 = __uint_as_float(sl[14 * kSlotN + slot]); sr.K = __uint_as_float(sl[15 * kSlotN + slot]);
                node = sv.bvh_root; sp = 0; qcnt = 0; overflow = false; best_up = __builtin_inff();
                walking = true;
            }
            n_ready -= n_idle < n_ready ? n_idle : n_ready;
            idle_mask = __ballot(!walking);
            n_idle = (uint32_t)__popcll(idle_mask);
        }
        RTX_SLOT_T(t_s1)
        RTX_SLOT_ACC(3, t_s0, t_s1)
        // ---- the f64 phase, for up to 64 DONE slots (the head of the queue), when it runs full -- or the walk is starving
        if (n_done >= 64u || (n_done != 0u && n_ready == 0u && (n_idle == 64u || n_idle >= kSlotServe))) {
            const uint32_t take = n_done < 64u ? n_done : 64u;
            const bool have = lane < take;
            const uint32_t my = have ? (uint32_t)done_list[lane] : 0u;
            {   // the rest of the queue moves up (at most kSlotN - 64 <= 64 entries: one per lane)
                const uint32_t rest = n_done - take;
                const uint32_t mv = lane < rest ? (uint32_t)done_list[take + lane] : 0u;
                if (lane < rest) done_list[lane] = (uint8_t)mv;
                n_done = rest;
            }
            SlotRec *const rec = recs + my;
            uint32_t fl = have ? sl[9 * kSlotN + my] : 0u;
            RayState r;
            uint32_t ridx = 0;
            bool go = false;                                            // this slot has a segment to set up
            // (a) a slot whose walk ended: closest_object's exact part + ray_hit
            if (have && (fl & kSlotFresh) == 0u) {
                const double4 ra = rec->a, rb = rec->b, rc = rec->c, rd = rec->d;
                r.pos = mk(ra.x, ra.y, ra.z);
                r.dir = mk(ra.w, rb.x, rb.y);
                r.result = mk(rb.z, rb.w, rc.x);
                r.light = mk(rc.y, rc.z, rc.w);
                r.key = (uint64_t)__double_as_longlong(rd.x);
                const unsigned long long ib = (unsigned long long)__double_as_longlong(rd.y);
                ridx = (uint32_t)ib;
                r.bounce = (uint32_t)(ib >> 32);
                r.draw = 6u + 2u * r.bounce;
                const float bu = __uint_as_float(sl[8 * kSlotN + my]);
                const RayX rx = make_rayx(r.pos, r.dir);
                Hit h;
                hit_init(h);
                ++segs;
                if ((fl & (kSlotOverflow | kSlotNoWalk)) == 0u) {
                    const uint32_t nq = fl & 0xFFu;
#pragma unroll 1
                    for (uint32_t e = 0; e < nq; ++e) {
                        if (__uint_as_float(sl[(size_t)(kSphQueue + e) * kSlotN + my]) <= bu) {
                            const uint32_t idx = sl[(size_t)e * kSlotN + my];
                            double t;
                            if (sphere_distance(la.spheres[idx], rx, &t)) hit_consider(h, t, la.sphere_ids[idx], 0, idx);
                            exact += 1;
                        }
                    }
                } else {                                  // no walk (origin out of range / NaN) or a dropped candidate: every sphere
                    for (uint32_t k = 0; k < sv.n_spheres; ++k) {
                        double t;
                        if (sphere_distance(la.spheres[k], rx, &t)) hit_consider(h, t, la.sphere_ids[k], 0, k);
                    }
                    exact += sv.n_spheres;
                }
                for (uint32_t k = 0; k < sv.n_planes; ++k) {
                    double t;
                    if (plane_distance(sv.planes[k], rx, &t)) hit_consider(h, t, sv.planes[k].id, 1, k);
                }
                for (uint32_t k = 0; k < sv.n_tri_filter; ++k) {       // the few triangles of a sphere scene (none of them in the tree)
                    const uint32_t tk = la.tri_fidx[k];
                    double t;
                    if (triangle_distance(la.tris[tk], rx, &t)) hit_consider(h, t, la.tris[tk].id, 2, tk);
                }
                exact += sv.n_planes + sv.n_tri_filter;
                bool done = true;
                if (h.id != kNone) {
                    advance_and_shade(sv, h, r);
                    done = (r.bounce >= bounce_limit) || light_is_zero(r);            // scene.rs:227-228
                }
                if (done) {
                    store_sample(samples, rv, ridx, r.result);
                    fl = kSlotFresh;                                    // the slot is free for the next survivor
                } else go = true;
            }
            // (b) a free slot: the next survivor of stage 1's queue, as it is after its first hit
            const unsigned long long fm = __ballot(have && (fl & kSlotFresh) != 0u);
            bool dead = false;
            if (fm != 0ull) {
                if (wave_next >= wave_end && !queue_empty) {
                    unsigned long long b = 0;
                    if (lane == 0) b = atomicAdd(work_counter, (unsigned long long)rv.grab);
                    b = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(b >> 32)) << 32) |
                        __builtin_amdgcn_readfirstlane((uint32_t)b);
                    wave_next = b;
                    wave_end = b + rv.grab < n_rays ? b + rv.grab : n_rays;
                    if (b >= n_rays) { queue_empty = true; wave_next = wave_end = 0; }
                }
                if (have && (fl & kSlotFresh) != 0u) {
                    const unsigned long long qi = wave_next + bvh_mbcnt(fm);
                    if (qi < wave_end) {
                        const double4 *p = reinterpret_cast<const double4 *>(sq.rec + (sq.perm ? (unsigned long long)sq.perm[qi] : qi));
                        const double4 s0 = p[0], s1 = p[1];
                        ridx = (uint32_t)(unsigned long long)__double_as_longlong(s1.z);
                        if (ridx != kNone) {
                            uint32_t pl = 0, smp = 0;
                            if (rv.tiles_x != 0u) (void)ray_index_to_pixel_tiled(rv, ridx, pl, smp);
                            else ray_index_to_pixel(rv, ridx, pl, smp);
                            const uint32_t k = fastdiv(pl, rv.div_width), x = pl - k * rv.width;
                            const uint64_t pix = (uint64_t)image_row(rv, k) * rv.width + x;
                            r.key = rng_key(sv.seed, pix, rv.sample_begin + smp);
                            r.bounce = 1u;
                            r.draw = 8u;
                            r.pos = mk(s0.x, s0.y, s0.z);
                            r.dir = mk(s0.w, s1.x, s1.y);
                            // ray_hit's two folds of the first hit (scene.rs:276-277) from resulting_color = 0, light_color = 1 (ray.rs:18-19)
                            const MaterialX m = sv.materials[(uint32_t)((unsigned long long)__double_as_longlong(s1.z) >> 32)];
                            r.result = vadd(mk(0.0, 0.0, 0.0), vmulv(mk(1.0, 1.0, 1.0), m.emission_color));
                            r.light = vmulv(mk(1.0, 1.0, 1.0), m.base_color);
                            go = true;
                        }                                                // (a slot its wave reserved and did not use: asked again next time)
                    } else if (queue_empty) dead = true;                 // nothing left to take: the slot retires
                }
                const unsigned long long taken = (unsigned long long)__popcll(fm);
                wave_next = wave_next + taken < wave_end ? wave_next + taken : wave_end;
            }
            // (c) the segment's set-up: the walk's f32 parameters into the slot's LDS words, the path state back to its record
            bool nowalk = false;
            if (go) {
                const RayX rn = make_rayx(r.pos, r.dir);
                const float omax = fmaxf(fmaxf(__builtin_fabsf((float)r.pos.x), __builtin_fabsf((float)r.pos.y)),
                                         __builtin_fabsf((float)r.pos.z));
                const bool in32 = omax <= sv.bvh_origin_limit;                              // NaN origin -> no walk
                if (in32 || omax <= sv.bvh_origin_limit * kBvhRange64) {
                    SphereRay s2;
                    sphere_ray_from(sv, r.pos, r.dir, s2);
                    Ray32 q0;
                    make_ray32(r.pos, rn.dirn, (double)sv.bvh_inv_max, q0);
                    sl[0 * kSlotN + my] = __float_as_uint(q0.ix); sl[1 * kSlotN + my] = __float_as_uint(q0.iy); sl[2 * kSlotN + my] = __float_as_uint(q0.iz);
                    sl[3 * kSlotN + my] = __float_as_uint(q0.nx); sl[4 * kSlotN + my] = __float_as_uint(q0.ny); sl[5 * kSlotN + my] = __float_as_uint(q0.nz);
                    sl[6 * kSlotN + my] = __float_as_uint(ray32_slack(q0.nx, q0.ny, q0.nz, in32));
                    sl[7 * kSlotN + my] = __float_as_uint(s2.px); sl[8 * kSlotN + my] = __float_as_uint(s2.py); sl[9 * kSlotN + my] = __float_as_uint(s2.pz);
                    sl[10 * kSlotN + my] = __float_as_uint(s2.dx); sl[11 * kSlotN + my] = __float_as_uint(s2.dy); sl[12 * kSlotN + my] = __float_as_uint(s2.dz);
                    sl[13 * kSlotN + my] = __float_as_uint(s2.Kg); sl[14 * kSlotN + my] = __float_as_uint(s2.c0); sl[15 * kSlotN + my] = __float_as_uint(s2.K);
                } else nowalk = true;
                rec->a = make_double4(r.pos.x, r.pos.y, r.pos.z, r.dir.x);
                rec->b = make_double4(r.dir.y, r.dir.z, r.result.x, r.result.y);
                rec->c = make_double4(r.result.z, r.light.x, r.light.y, r.light.z);
                rec->d = make_double4(__longlong_as_double((long long)r.key),
                                      __longlong_as_double((long long)(((unsigned long long)r.bounce << 32) | (unsigned long long)ridx)), 0.0, 0.0);
                if (nowalk) { sl[9 * kSlotN + my] = kSlotNoWalk; sl[8 * kSlotN + my] = __float_as_uint(__builtin_inff()); }
            } else if (have && !dead) {
                sl[9 * kSlotN + my] = kSlotFresh;                       // a free slot that got no survivor this time
            }
            // READY: set up and walkable.  DONE again: no walk possible (tested exhaustively next phase), or still free.
            const bool to_ready = go && !nowalk, to_done = have && !dead && !to_ready;
            const unsigned long long rm = __ballot(to_ready), dm = __ballot(to_done);
            if (to_ready) ready_list[n_ready + bvh_mbcnt(rm)] = (uint8_t)my;
            if (to_done) done_list[n_done + bvh_mbcnt(dm)] = (uint8_t)my;
            n_ready += (uint32_t)__popcll(rm);
            n_done += (uint32_t)__popcll(dm);
#ifndef RTX_SLOT_NOFENCE
            __threadfence_block();                                      // the records are read by other lanes of this wave, later
#endif
            { RTX_SLOT_T(t_s2) RTX_SLOT_ACC(2, t_s1, t_s2) }
            // free slots while the queue still has chunks are asked again; when it is empty they retired above, so a phase that
            // only re-queued free slots cannot repeat for ever
            continue;
        }
        if (n_idle == 64u) break;                  // nothing walking, nothing READY, nothing DONE: every slot retired
        // ---- the walk: node visits and leaf visits apart (sphere_walk_phased's iteration) for every lane that has an entry to
        //      open, until kSlotWait lanes have finished their walk (they are then served together), or nobody walks any more.
        //      A tight loop of its own: what the f64 phase spills stays outside it.
        for (;;) {
            const bool w = walking && node != kNone;
            const bool at_leaf = w && (node >> 29) != 0u;
            const unsigned long long lm = __ballot(at_leaf), am = __ballot(w);
            if (am == 0ull) break;
            if ((uint32_t)__popcll(lm) >= kSphLeafLanes || lm == am) {
                if (at_leaf)
                    sphere_leaf_step_at<STACK, SPILL, kSlotN>(la.sphere_f32, la.sphere_prims, sr, node, sp, &lds_stack[0][0], sl, tid, slot, spill,
                                                              spill_stride, glane, best_up, qcnt, overflow, nleaf);
            } else if (w && !at_leaf) {
#if defined(RTX_LAB) && defined(RTX_SLOT_COUNT_TOP)          // lab build: node visits served from the LDS copy, reported through exact_tests
                if (node < (uint32_t)(RTX_SLOT_COUNT_TOP)) exact += 1;
#endif
                sphere_node_step_q3<STACK, SPILL>(qnodes, q, node, sp, &lds_stack[0][0], tid, spill, spill_entries, spill_stride, glane, best_up,
                                                  overflow, nbox, kSlotTop > 0 ? (LdsF4Ptr)&lds_top[0] : (LdsF4Ptr) nullptr, top_n);
            }
            const uint32_t n_fin = (uint32_t)__popcll(__ballot(walking && node == kNone));
            if (n_fin >= kSlotWait) break;
        }
        { RTX_SLOT_T(t_s3) RTX_SLOT_ACC(1, t_s1, t_s3) }
    }
#ifdef RTX_SLOT_PROFILE
    exact = sp_prof;
#endif
    unsigned long long box_tests = nbox, filt = (unsigned long long)nbox + nleaf, segs64 = segs;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        segs64 += __shfl_xor(segs64, off, 64);
        exact += __shfl_xor(exact, off, 64);
        filt += __shfl_xor(filt, off, 64);
        box_tests += __shfl_xor(box_tests, off, 64);
    }
    if (lane == 0) {
        const uint32_t shard = (blockIdx.x * (kBvhThreads >> 6) + (tid >> 6)) & (kCounterShards - 1);
        if (segs64) atomicAdd(&ctr[shard].segments, segs64);
        if (exact) atomicAdd(&ctr[shard].exact_tests, exact);
        if (filt) atomicAdd(&ctr[shard].filter_tests, filt);
        if (box_tests) atomicAdd(&ctr[2 + (shard % (kCounterShards - 2))].pad_, box_tests);
    }
}


// ---- host: the lab forms of stage 2 (the caller has run stage 1 and carved the queue) ------------------------------------------------
// Returns hipErrorNotReady when the launch flags ask for none of them (the caller then launches the lock-step stage 2 itself).
// The survivors' sort, when asked for, fills sq.perm for whichever stage 2 follows.
static hipError_t launch_sph_lab_stage2(const SceneView *d_sv, const SceneView &sv, const RowsView *d_rv, const RowsView &rv, double *samples,
                                        Counters *counters, uint32_t *spill, uint32_t spill_entries, int n_cus, char *p, uint64_t capacity,
                                        unsigned long long *ctrs, SphQueue &sq, const LeafArrays &la, const float4 *nodes, uint32_t flags,
                                        void *pool_mem, hipStream_t stream)
{
    (void)rv;
    hipError_t e = hipSuccess;
    const bool deep = spill_entries != 0u;
    const uint64_t cap = (uint64_t)n_cus * kSphWavesPerSimd;
    if (flags & kSphSortSurvivors) {
        SphSort so{};
        char *q = p + 256 + capacity * sizeof(SphSurvivor);
        q = reinterpret_cast<char *>(((uintptr_t)q + 255) & ~(uintptr_t)255);
        so.hist = reinterpret_cast<unsigned int *>(q);
        so.perm = reinterpret_cast<uint32_t *>(q + 4 * 256);
        so.key = reinterpret_cast<uint8_t *>(so.perm + capacity);
        double diag2 = 0.0;
        for (int a = 0; a < 3; ++a) {
            so.lo[a] = (float)(sv.sphere_center[a] - sv.sphere_cmax); so.hi[a] = (float)(sv.sphere_center[a] + sv.sphere_cmax);
            diag2 += 4.0 * sv.sphere_cmax * sv.sphere_cmax;
        }
        so.inv_dt = diag2 > 0.0 ? (float)(kSortT / (0.6 * std::sqrt(diag2))) : 0.f;
        if ((e = hipMemsetAsync(so.hist, 0, kSortBins * sizeof(unsigned int), stream)) != hipSuccess) return e;
        const uint32_t sblocks = (uint32_t)n_cus * 8u;
        hipLaunchKernelGGL(sph_sort_hist_kernel, dim3(sblocks), dim3(256), 0, stream, sq, so);
        hipLaunchKernelGGL(sph_sort_scan_kernel, dim3(1), dim3(256), 0, stream, so);
        hipLaunchKernelGGL(sph_sort_scatter_kernel, dim3(sblocks), dim3(256), 0, stream, sq, so);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        sq.perm = so.perm;
    }
    const bool pq3 = (sv.bvh_flags & 16u) != 0u && sv.bvh_q3nodes != nullptr && (sv.tuning & RTX_TUNE_NO_QNODES) == 0u;
    if (pool_mem && (flags & kSphPair) != 0u) {
        // stage 2 with two rays per lane (trace_sph_pair_kernel)
        SphPair pp{};
        const size_t lanes2 = (size_t)n_cus * kSphWavesPerSimd * kBvhThreads * 2;
        pp.stride = lanes2;
        pp.f = reinterpret_cast<double *>(pool_mem);
        pp.u = reinterpret_cast<uint32_t *>(pp.f + (size_t)kPairF64 * lanes2);
        auto kp = pq3 ? (deep ? trace_sph_pair_kernel<true, true> : trace_sph_pair_kernel<false, true>)
                      : (deep ? trace_sph_pair_kernel<true, false> : trace_sph_pair_kernel<false, false>);
        hipLaunchKernelGGL(kp, dim3((uint32_t)n_cus * kPairWaves), dim3(kBvhThreads), 0, stream, d_sv, d_rv, samples, counters, ctrs + 1,
                           pq3 ? reinterpret_cast<const float4 *>(sv.bvh_q3nodes) : nodes, la, spill, spill_entries, sq, pp);
        return hipGetLastError();
    }
    if (pool_mem && (flags & kSphPool) != 0u) {
        // stage 2 as a wave-local pool (trace_sph_pool_kernel): the grid is the resident waves, each with its own slots
        SphPool pool{};
        const size_t slots = (size_t)n_cus * kSphWavesPerSimd * (kBvhThreads / 64) * kPoolSlots;
        pool.stride = slots;
        pool.f = reinterpret_cast<double *>(pool_mem);
        pool.u = reinterpret_cast<uint32_t *>(pool.f + (size_t)kPoolF64 * slots);
        const uint32_t pool_spill = kPoolStack < (int)(3u * sv.bvh_depth + 2u) && spill ? 3u * sv.bvh_depth + 2u - (uint32_t)kPoolStack : 0u;
        auto kp = pq3 ? (pool_spill ? trace_sph_pool_kernel<true, true> : trace_sph_pool_kernel<false, true>)
                      : (pool_spill ? trace_sph_pool_kernel<true, false> : trace_sph_pool_kernel<false, false>);
        hipLaunchKernelGGL(kp, dim3((uint32_t)cap), dim3(kBvhThreads), 0, stream, d_sv, d_rv, samples, counters, ctrs + 1,
                           pq3 ? reinterpret_cast<const float4 *>(sv.bvh_q3nodes) : nodes, la, spill, pool_spill, sq, pool);
        return hipGetLastError();
    }
    return hipErrorNotReady;
}
