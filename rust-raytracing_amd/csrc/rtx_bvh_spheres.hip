// rtx_bvh_spheres.hip -- RTX_KERNEL_BVH for trees that hold spheres only (C2, C4).
//
// Same rays, same tree, same exact tests and the same bits as trace_bvh_kernel<false, *> (rtx_bvh.hip); what differs is the
// traversal, which holds no f64 value: it prunes with the conservative f32 distance bounds of rtx_traverse.h instead of the
// exact winner's distance, and the candidates that can still be the winner get their exact f64 test (sphere.rs:19-30) once,
// after the walk.  What is in this file, in the order a big launch runs it (history in profiles/LAB_NOTEBOOK.md):
//   * trace_sph_packet_kernel      stage 1: the primary rays, ONE wave-uniform walk per 8x8 tile through the scalar cache
//   * trace_bvh_spheres_kernel     stage 2 (MODE 2, from the survivors' queue) and small launches (MODE 0): per-lane walks over
//                                  the 64-byte nodes, node visits and leaf visits apart (sphere_walk_phased), a wave leaving
//                                  its round when few lanes still walk (the walk is resumable); MODE 1 = stage 1 per lane (A/B)
// (Experiments behind RtxConfig.tuning bits -- trace_sph_pool_kernel, trace_sph_pair_kernel, sph_sort_*, MODE 1, the 128-byte-node and
// inline-leaf instances -- are compiled with -DRTX_LAB only: rtx_bvh_spheres_lab.h, librtx_hip_lab.so.)
// The traversal loops are free of scratch traffic (tools/isa_spills.py): what the allocator spills (the f64 path state) is touched
// only between walks.  4 workgroups per CU: the loops are VALU-issue bound, more waves only add spill reloads inside them
// (5 / 6 / 8 per SIMD measured in round 1), 3 without any spill are 3 % slower (round 3).
#include "rtx_launch.h"
#include "rtx_traverse.h"

#include <cmath>

namespace rtx {

#ifndef RTX_SPH_WAVES
#define RTX_SPH_WAVES 4
#endif
constexpr int kSphWavesPerSimd = RTX_SPH_WAVES;          // = workgroups per CU (4 waves each)
constexpr int kSphStack = 30;                // LDS stack entries per lane: (30 + 1 sink row + 2 * kSphQueue queue rows) KB per workgroup

// Two stages.  A wave's round lasts as long as its longest walk.  Primary rays (an 8x8 tile per wave: coherent, short
// walks) and bounced rays (incoherent, long walks) in one wave make the primaries wait for the bounces: measured on C2,
// the primary rays alone take 24.7 of the 89.5 ms although they are 47 % of the segments.  So the launch is split:
//   MODE 1   primary rays only: every lane takes a fresh ray each round; a ray that survives its first hit goes to a
//            queue (state structure-of-arrays by slot; a wave reserves 512 slots per atomic and marks what it leaves
//            unused as dead)
//   MODE 2   the same kernel fed from that queue: every lane carries a bounced ray, idle lanes take the next ones
// MODE 0 is the single launch (small launches, A/B runs: RTX_TUNE_ONE_STAGE).
// A survivor is what a ray is after its FIRST hit: position, new direction, its index in the launch's queue and the object
// it hit -- ray.resulting_color / light_color are then exactly that object's emission / base colour folded into (0,0,0) /
// (1,1,1) (scene.rs:276-277), which stage 2 recomputes with the same two operations.  One 64-byte record per survivor,
// written and read whole (four 16-byte accesses per lane, a wave's records contiguous).
struct SphSurvivor {
    double px, py, pz, dx, dy, dz;
    uint32_t ridx;                            // kNone: a slot its wave reserved and did not use
    uint32_t first_id;                        // scene index of the object of the first hit
    uint32_t pad_[2];
};
static_assert(sizeof(SphSurvivor) == 64, "SphSurvivor must be one 64-byte line");
struct SphQueue {
    SphSurvivor *rec;
    const uint32_t *perm;                     // stage 2 reads rec[perm[k]] (the survivors ordered by sph_sort_*), or null: queue order
    unsigned long long *count;                // slots reserved by stage 1 = the length stage 2 walks
    unsigned long long capacity;
    uint32_t cut_walkers;                     // sphere_walk_resumable's threshold for the lock-step kernels (0: walks are never cut)
};
constexpr uint32_t kSphQueueChunk = 512;
#ifndef RTX_SPH_CUT
#define RTX_SPH_CUT 24
#endif
#ifndef RTX_SPH_CUT_DONE
#define RTX_SPH_CUT_DONE 32
#endif
#ifndef RTX_SPH_LEAF_LANES
#define RTX_SPH_LEAF_LANES 8
#endif
constexpr uint32_t kSphLeafLanes = RTX_SPH_LEAF_LANES;   // a leaf visit runs when this many lanes of the wave hold a leaf (or all that walk do)
constexpr uint32_t kSphCutWalkers = RTX_SPH_CUT;     // a round's walk is left when fewer lanes than this still walk ...
constexpr uint32_t kSphCutDone = RTX_SPH_CUT_DONE;   // ... and at least this many of the wave's rays wait for their f64 phase

// Appends the wave's survivors (lanes with `go`) to the queue: consecutive slots of the wave's current reservation; when it
// runs out mid-way the rest continue in a fresh chunk (one atomic per 512 records), so no slot is wasted except the tail of
// a wave's LAST chunk, which the kernel marks dead before it leaves.  A slot beyond the capacity (the host sizes the queue
// so that it cannot happen: n_rays + one chunk per wave) raises the launch's watchdog word instead of being dropped silently.
__device__ __forceinline__ void sph_queue_append(const SphQueue &sq, Counters *__restrict__ ctr, bool go, const RayState &r,
                                                 uint32_t ridx, uint32_t first_id, uint32_t lane,
                                                 unsigned long long &out_next, unsigned long long &out_end)
{
    const unsigned long long m = __ballot(go);
    const uint32_t n = (uint32_t)__popcll(m);
    if (n == 0u) return;
    const unsigned long long rem = out_end - out_next;
    unsigned long long base2 = 0;
    if (rem < (unsigned long long)n) {
        if (lane == 0) base2 = atomicAdd(sq.count, (unsigned long long)kSphQueueChunk);
        base2 = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base2 >> 32)) << 32) |
                __builtin_amdgcn_readfirstlane((uint32_t)base2);
    }
    if (go) {
        const unsigned long long k = bvh_mbcnt(m);
        const unsigned long long slot = k < rem ? out_next + k : base2 + (k - rem);
        if (slot < sq.capacity) {
            double4 *p = reinterpret_cast<double4 *>(sq.rec + slot);
            p[0] = make_double4(r.pos.x, r.pos.y, r.pos.z, r.dir.x);
            uint2 tail;
            tail.x = ridx; tail.y = first_id;
            p[1] = make_double4(r.dir.y, r.dir.z, __longlong_as_double(((long long)tail.y << 32) | (long long)tail.x), 0.0);
        } else {
            atomicAdd(&ctr[1].pad_, 1ull);
        }
    }
    if (rem < (unsigned long long)n) { out_next = base2 + ((unsigned long long)n - rem); out_end = base2 + kSphQueueChunk; }
    else out_next += n;
}

__device__ __forceinline__ void sph_queue_close(const SphQueue &sq, uint32_t lane, unsigned long long out_next, unsigned long long out_end)
{
    for (unsigned long long s = out_next + lane; s < out_end; s += 64ull)
        if (s < sq.capacity) sq.rec[s].ridx = kNone;
}

template <bool SPILL, int MODE, int Q3 = 0>               // Q3: `nodes` are the 64-byte nodes (rtx_bvh.h BvhQ3Node), 4 float4 each; 2: leaf visits apart (sphere_walk_phased), 1: inside the node visit
__global__ __launch_bounds__(kBvhThreads, kSphWavesPerSimd) void trace_bvh_spheres_kernel(const SceneView *__restrict__ svp,
                                                                             const RowsView *__restrict__ rvp,
                                                                             double *__restrict__ samples, Counters *__restrict__ ctr,
                                                                             unsigned long long *__restrict__ work_counter,
                                                                             const float4 *__restrict__ nodes, const LeafArrays la,
                                                                             uint32_t *__restrict__ spill, uint32_t spill_entries,
                                                                             const SphQueue sq)
{
    constexpr int STACK = kSphStack;
    const SceneView &sv = *svp;
    const RowsView &rv = *rvp;
    __shared__ uint32_t lds_stack[STACK + 1][kBvhThreads];             // + the sink row of the branch-free pushes
    __shared__ uint32_t lds_q[2 * kSphQueue][kBvhThreads];             // candidate indices, then their t_lo
    uint32_t *const lq = &lds_q[0][0];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const size_t spill_stride = (size_t)gridDim.x * kBvhThreads, glane = (size_t)blockIdx.x * kBvhThreads + tid;
    const uint32_t bounce_limit = sv.max_bounces >= 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)sv.max_bounces + 1u;

    const unsigned long long n_rays = MODE == 2 ? (*sq.count < sq.capacity ? *sq.count : sq.capacity) : rv.n_rays;
    unsigned long long wave_next = 0, wave_end = 0;      // this wave's share of the ray queue (wave-uniform)
    unsigned long long out_next = 0, out_end = 0;        // MODE 1: this wave's reserved slots of the survivors' queue
    bool queue_empty = false;
    bool alive = false;
    RayState r;
    uint32_t ridx = 0;                       // the ray's index in the launch's queue = where its sample goes
    uint32_t first_id = 0;                   // MODE 1: the object of the first hit
    unsigned long long segs = 0, box_tests = 0, leaf_filters = 0, exact = 0;
    // the walk's state: lives across rounds for a lane whose walk was cut (sphere_walk_resumable)
    float best_up = __builtin_inff();
    uint32_t qcnt = 0, nbox = 0, nleaf = 0, w_node = kNone, w_sp = 0;
    bool overflow = false, walked = false, midwalk = false;
#if defined(RTX_LAB) && defined(RTX_SPH_PROFILE)
    unsigned long long rtx_prof = 0, t_prev = 0;
#define RTX_PROF_PASS , rtx_prof
#else
#define RTX_PROF_PASS
#endif
    const uint32_t cut_walkers = MODE == 1 ? 0u : sq.cut_walkers;       // stage 1 hands every lane a fresh ray each round: no cut there

    for (;;) {
        // ---- hand rays to idle lanes: ballot + prefix sum over the wave's local range, one atomic per rv.grab rays
        const unsigned long long idle_mask = __ballot(!alive);
        if (idle_mask != 0ull) {
            if (wave_next >= wave_end && !queue_empty) {
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(work_counter, (unsigned long long)rv.grab);
                base = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
                       __builtin_amdgcn_readfirstlane((uint32_t)base);
                wave_next = base;
                wave_end = base + rv.grab < n_rays ? base + rv.grab : n_rays;
                if (base >= n_rays) { queue_empty = true; wave_next = wave_end = 0; }
            }
            if (!alive && wave_next < wave_end) {
                const unsigned long long my = wave_next + bvh_mbcnt(idle_mask);
                bool valid = my < wave_end;
                uint32_t pl = 0, smp = 0;
                if constexpr (MODE == 2) {
                    double4 s0 = make_double4(0., 0., 0., 0.), s1 = s0;
                    if (valid) {                           // a ray in flight, as stage 1 left it after its first hit
                        const double4 *p = reinterpret_cast<const double4 *>(sq.rec + (sq.perm ? (unsigned long long)sq.perm[my] : my));
                        s0 = p[0]; s1 = p[1];
                        ridx = (uint32_t)(unsigned long long)__double_as_longlong(s1.z);
                        valid = ridx != kNone;
                    }
                    if (valid) {
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
                        alive = true;
                    }
                } else {
                    if (valid) {
                        if (rv.tiles_x != 0u) valid = ray_index_to_pixel_tiled(rv, my, pl, smp);
                        else ray_index_to_pixel(rv, my, pl, smp);
                    }
                    if (valid) {
                        gen_primary(sv, rv, pl, rv.sample_begin + smp, r);
                        ridx = (uint32_t)my;                                          // (the host keeps rv.n_rays below 2^32)
                        alive = true;
                    }
                }
            }
            const unsigned long long taken = (unsigned long long)__popcll(idle_mask);
            wave_next = wave_next + taken < wave_end ? wave_next + taken : wave_end;
        }
        if (__ballot(alive) == 0ull) {
            if (queue_empty) break;            // wave-uniform: nothing live, nothing left to take
            continue;
        }

        // ---- one segment: closest_object (scene.rs:243-251).  Phase 1, f32 only: walk the tree, collect candidates.
        // A lane whose walk was cut (midwalk) comes back with its walk state and goes on where it left
#if defined(RTX_LAB) && defined(RTX_SPH_PROFILE) && RTX_SPH_PROFILE >= 10
        const unsigned long long t_w0 = __builtin_amdgcn_s_memtime();       // lab build: where a wave's time goes (1/64 of the cycles, by lane 0)
        if (RTX_SPH_PROFILE == 11 && lane == 0 && t_prev != 0ull) rtx_prof += (t_w0 - t_prev) >> 6;
#endif
        const uint32_t n_alive = (uint32_t)__popcll(__ballot(alive));
        if (!midwalk) { best_up = __builtin_inff(); qcnt = 0; nbox = 0; nleaf = 0; overflow = false; walked = false; w_node = kNone; w_sp = 0; }
        RayX rx;
        if (alive) {
            rx = make_rayx(r.pos, r.dir);
            const float omax = fmaxf(fmaxf(__builtin_fabsf((float)r.pos.x), __builtin_fabsf((float)r.pos.y)),
                                     __builtin_fabsf((float)r.pos.z));
            const bool in32 = omax <= sv.bvh_origin_limit;                                  // NaN origin -> exhaustive branch
            const bool in64 = !in32 && omax <= sv.bvh_origin_limit * kBvhRange64;
            if (in32 || in64) {
                SphereRay sr;
                sphere_ray_from(sv, r.pos, r.dir, sr);
                const V3 dirn = rx.dirn;
                if (!midwalk) w_node = sv.bvh_root;
                if constexpr (Q3) {                   // (a far origin walks the same tree with Ray32S's slack instead of an f64 slab test)
                    Ray32 q0;
                    make_ray32(r.pos, dirn, (double)sv.bvh_inv_max, q0);
                    Ray32S q;
                    q.ix = q0.ix; q.iy = q0.iy; q.iz = q0.iz; q.nx = q0.nx; q.ny = q0.ny; q.nz = q0.nz;
                    q.e = ray32_slack(q0.nx, q0.ny, q0.nz, in32);
                    if constexpr (Q3 == 2)            // node visits and leaf visits apart (the default)
                        sphere_walk_phased<STACK, SPILL>(nodes, la.sphere_f32, la.sphere_prims, q, sr, w_node, w_sp, &lds_stack[0][0],
                                                         lq, tid, spill, spill_entries, spill_stride, glane, best_up, qcnt,
                                                         overflow, nbox, nleaf, cut_walkers, kSphCutDone, n_alive, kSphLeafLanes RTX_PROF_PASS);
                    else
                        sphere_walk_resumable<STACK, SPILL, true>(nodes, la.sphere_f32, la.sphere_prims, q, sr, w_node, w_sp, &lds_stack[0][0],
                                                                  lq, tid, spill, spill_entries, spill_stride, glane, best_up, qcnt,
                                                                  overflow, nbox, nleaf, cut_walkers, kSphCutDone, n_alive);
                } else if (in32) {
                    Ray32 q;
                    make_ray32(r.pos, dirn, (double)sv.bvh_inv_max, q);
                    sphere_walk_resumable<STACK, SPILL, false>(nodes, la.sphere_f32, la.sphere_prims, q, sr, w_node, w_sp, &lds_stack[0][0],
                                                               lq, tid, spill, spill_entries, spill_stride, glane, best_up, qcnt,
                                                               overflow, nbox, nleaf, cut_walkers, kSphCutDone, n_alive);
                } else {                              // origin far outside the scene: the same walk with an f64 slab test
                    Ray64 q;
                    make_ray64(r.pos, dirn, (double)sv.bvh_inv_max, q);
                    sphere_walk_resumable<STACK, SPILL, false>(nodes, la.sphere_f32, la.sphere_prims, q, sr, w_node, w_sp, &lds_stack[0][0],
                                                               lq, tid, spill, spill_entries, spill_stride, glane, best_up, qcnt,
                                                               overflow, nbox, nleaf, cut_walkers, kSphCutDone, n_alive);
                }
                walked = true;
                midwalk = w_node != kNone;
            }
        }
        // ---- phase 2, f64: exact tests of the candidates that can still be the winner, the shapes outside the tree, ray_hit
#if defined(RTX_LAB) && defined(RTX_SPH_PROFILE) && RTX_SPH_PROFILE >= 10
        {
            const unsigned long long t_w1 = __builtin_amdgcn_s_memtime();
            if (RTX_SPH_PROFILE == 10 && lane == 0) rtx_prof += (t_w1 - t_w0) >> 6;        // the walk
            t_prev = t_w1;                                                                   // 11: from here to the next walk (f64 phase + refill)
        }
#endif
#if defined(RTX_LAB) && defined(RTX_SPH_PROFILE)
        {   // lab build: 1 f64 phases (rounds), 2 the exact loop's wave iterations (8: its lane iterations = exact tests + skipped entries)
            const unsigned long long fm = __ballot(alive && !midwalk);
            if (RTX_SPH_PROFILE == 1 && fm != 0ull && (uint32_t)(__ffsll((long long)fm) - 1) == lane) rtx_prof += 1;
            if (RTX_SPH_PROFILE == 2) {
                uint32_t mx = alive && !midwalk && walked && !overflow ? qcnt : 0u;
                for (int off = 32; off > 0; off >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)mx, off, 64); mx = mx > o ? mx : o; }
                if (lane == 0) rtx_prof += mx;
            }
            if (RTX_SPH_PROFILE == 8 && alive && !midwalk && walked && !overflow) rtx_prof += qcnt;
        }
#endif
        if (alive && !midwalk) {
            Hit h;
            hit_init(h);
            ++segs;
            box_tests += nbox;
            leaf_filters += nleaf;
            if (walked && !overflow) {
#pragma unroll 1
                for (uint32_t e = 0; e < qcnt; ++e) {
                    if (__uint_as_float(lq[(size_t)(kSphQueue + e) * kBvhThreads + tid]) <= best_up) {
                        const uint32_t idx = lq[(size_t)e * kBvhThreads + tid];
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
            // the few triangles of a sphere scene (none of them in the tree), by filter record
            for (uint32_t k = 0; k < sv.n_tri_filter; ++k) {
                const uint32_t tk = la.tri_fidx[k];
                double t;
                if (triangle_distance(la.tris[tk], rx, &t)) hit_consider(h, t, la.tris[tk].id, 2, tk);
            }
            exact += sv.n_planes + sv.n_tri_filter;

            // ---- render_ray's match arm + ray_hit (scene.rs:232-239, 260-278)
            bool done = true;
            if (h.id != kNone) {
                advance_and_shade(sv, h, r);
                first_id = h.id;
                done = (r.bounce >= bounce_limit) || light_is_zero(r);            // scene.rs:227-228
            }
            if (done) {
                store_sample(samples, rv, ridx, r.result);
                alive = false;
            }
        }
        if constexpr (MODE == 1) {
            // ---- the survivors of this round move to the queue of rays in flight; every lane is free again
            sph_queue_append(sq, ctr, alive, r, ridx, first_id, lane, out_next, out_end);
            alive = false;
        }
    }
    if constexpr (MODE == 1) sph_queue_close(sq, lane, out_next, out_end);      // the unused rest of the wave's last reservation
#if defined(RTX_LAB) && defined(RTX_SPH_PROFILE)
    if (MODE == 2) exact = rtx_prof;                             // lab build: stage 2 reports its profile count through exact_tests
#endif
    // counters: segments, exact f64 shape tests, box tests + leaf filter tests (reported through filter_tests)
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
        if (box_tests) atomicAdd(&ctr[2 + (shard % (kCounterShards - 2))].pad_, box_tests);   // shards 0,1 carry debug flags
    }
}

// ---- stage 1 as packets -----------------------------------------------------------------------------------------------------
// The primary rays of an 8x8 pixel tile leave (almost) one point in directions a few pixels apart: at the far side of C2's
// cloud they are less than a unit apart, a fraction of the spacing of the spheres.  So the wave walks the tree ONCE for the
// 64 of them (the move that took C3's level 0 from 62.9 to 29.0 ms, rtx_wavefront.hip): a wave-uniform stack, the node and
// the leaf's {c - centre, r} record read at a wave-uniform address through the scalar cache, every lane testing ITS ray
// against the four boxes with ITS bound and bounding the leaf's sphere with sphere_step's closest-approach terms.  A child
// is opened when any lane's ray enters it, in the first entering lane's order.  What a lane collects is what its own walk
// collects for SOME visiting order -- every candidate with t_lo <= its best_up -- and the exact tests (sphere.rs:19-30)
// decide as always: bit-identical.  Around the walk the wave runs render_pixel's prologue, the exact tests and ray_hit
// with all 64 lanes (no lane waits for a longer walk), and the survivors go to stage 2's queue.
//
// When all rays of the tile point into the same octant (every tile but those on the image's centre row / column) the
// near and far plane of each slab are picked with scalar selects and the slab test is 6 FMAs + max3 / min3 instead of
// 6 FMAs + 12 min / max: same values (the FMA is monotone in the plane), fewer VALU instructions.
constexpr int kSpkStack = 63;                // wave-uniform stack entries: the lanes of ONE VGPR (the host checks 3 * depth + 2 against it)
#ifndef RTX_SPK_WAVES
#define RTX_SPK_WAVES 4
#endif
constexpr int kSpkWaves = RTX_SPK_WAVES;     // workgroups per CU
#ifdef RTX_LAB                               // launch flags of the lab forms (rtx_api.hip builds them from RtxConfig.tuning)
constexpr uint32_t kSphNoPackets = 1u;       // stage 1 per lane (A/B runs)
constexpr uint32_t kSphPair = 8u;            // stage 2 with two rays per lane (trace_sph_pair_kernel)
constexpr uint32_t kSphPool = 4u;            // stage 2 as the wave-local pool (an experiment: slower) instead of lock-step
constexpr uint32_t kSphSortSurvivors = 2u;   // stage 2 reads the survivors ordered by exit distance and octant
#endif

// One sphere of a leaf against one lane's ray, f32 only: the candidate test and bounds of sphere_step (rtx_traverse.h).  A sphere the
// exact test cannot be excluded for joins the lane's candidate queue (LDS: {local index, t_lo}); a certain hit tightens best_up.
__device__ __forceinline__ void sph_packet_leaf_test(const float4 rec, const uint32_t prim, const SphereRay &sr, uint32_t *lq, uint32_t tid,
                                 float &best_up, uint32_t &qcnt, bool &overflow)
{
    const float ox = rec.x - sr.px, oy = rec.y - sr.py, oz = rec.z - sr.pz;
    const float bq = __builtin_fmaf(ox, sr.dx, __builtin_fmaf(oy, sr.dy, oz * sr.dz));
    const float lx = __builtin_fmaf(-bq, sr.dx, ox), ly = __builtin_fmaf(-bq, sr.dy, oy), lz = __builtin_fmaf(-bq, sr.dz, oz);
    const float l2 = __builtin_fmaf(lx, lx, __builtin_fmaf(ly, ly, lz * lz));
    const float Dl = __builtin_fmaf(rec.w, rec.w, -l2);
    const float G = __builtin_fmaf(sr.Kg, rec.w, sr.c0);
    const float Dp = Dl + G;
    if (Dp >= 0.0f) {                                      // the exact test cannot be excluded (rtx_traverse.h, sphere_step)
        const float tlo = bq - __builtin_amdgcn_sqrtf(Dp) * (1.0f + 4.76837158e-7f) - sr.K;
        const float Dm = Dl - G;
        const float thi = Dm > 0.0f ? bq - __builtin_amdgcn_sqrtf(Dm) * (1.0f - 4.76837158e-7f) + sr.K : __builtin_inff();
        if (tlo <= best_up && !(thi < 0.0f)) {
            if (tlo > sr.K) best_up = fminf(best_up, thi);
            if (qcnt == (uint32_t)kSphQueue) {             // drop the entries a later certain hit has overtaken
                uint32_t w = 0;
#pragma unroll
                for (int e = 0; e < kSphQueue; ++e) {
                    const uint32_t ie = lq[(size_t)e * kBvhThreads + tid];
                    const uint32_t te = lq[(size_t)(kSphQueue + e) * kBvhThreads + tid];
                    if (__uint_as_float(te) <= best_up) {
                        lq[(size_t)w * kBvhThreads + tid] = ie;
                        lq[(size_t)(kSphQueue + w) * kBvhThreads + tid] = te;
                        w += 1;
                    }
                }
                qcnt = w;
            }
            if (qcnt == (uint32_t)kSphQueue) overflow = true;   // (the segment then tests every sphere exactly)
            else {
                lq[(size_t)qcnt * kBvhThreads + tid] = prim;
                lq[(size_t)(kSphQueue + qcnt) * kBvhThreads + tid] = __float_as_uint(tlo);
                qcnt += 1;
            }
        }
    }
}

// The walk of one tile.  SGN < 8: every ray of the tile points into octant SGN (bit a set: direction component a is
// negative), so the near / far plane of each slab is known at compile time; SGN == 8: mixed signs, min / max per slab.
//
// Where the time goes decides the shape of this loop.  The first version (profiles/r03_sph_packets_v1.txt) issued 44 %
// fewer VALU instructions than the per-lane walk and was only 16 % faster: it had become SCALAR-bound -- 9.6e9 SALU
// instructions per launch against 7.7e9 VALU, and a CU has one scalar pipe for its four SIMDs (74 % busy) -- on the
// uniform near / far selects (48 s_and + s_cselect per visit), the exec-masked lane-0 stores of the LDS stack, and
// branches around every child.  Now: the selects are resolved at compile time (one instance of the loop per octant), the
// stack lives in the lanes of one VGPR (v_writelane / v_readlane with a scalar index: no LDS, no exec masking, three
// unconditional pushes), the hit masks of the box tests are kept as they come out of v_cmp and reused for the ordering
// keys, and the leaf code exists once.
// (rtx_writelane: rtx_traverse.h)

template <int SGN>
__device__ __forceinline__ void sph_packet_walk(const PkConst4 cnodes, const PkConst4 csph, const PkConstU32 cprims, uint32_t root,
                                                const Ray32S &q, const SphereRay &sr, uint32_t *lq, uint32_t tid, uint32_t lane,
                                                float &best_up, uint32_t &qcnt, bool &overflow, uint32_t &nbox, uint32_t &nleaf)
{
    int stk = 0;                                                       // lane k = stack entry k
    uint32_t sp = 0, node = __builtin_amdgcn_readfirstlane(root);
    while (node != kNone) {
        // the node's 128 bytes at a wave-uniform address: 4 x {lo.xyz, link}, 4 x {hi.xyz, count}
        const PkConst4 np = cnodes + 8 * (size_t)node;
        float4 a[4], b[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { a[c] = np[c]; b[c] = np[4 + c]; }
        uint32_t lnk[4], cnt[4];
        float tc[4];
        unsigned long long hm[4];                                      // lanes whose ray enters child c
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            lnk[c] = pk_bits(a[c].w);
            cnt[c] = pk_bits(b[c].w);
            float tn, tf;
            if (SGN < 8) {
                const float nx_ = (SGN & 1) ? b[c].x : a[c].x, fx_ = (SGN & 1) ? a[c].x : b[c].x;
                const float ny_ = (SGN & 2) ? b[c].y : a[c].y, fy_ = (SGN & 2) ? a[c].y : b[c].y;
                const float nz_ = (SGN & 4) ? b[c].z : a[c].z, fz_ = (SGN & 4) ? a[c].z : b[c].z;
                tn = fmaxf(fmaxf(__builtin_fmaf(nx_, q.ix, q.nx), __builtin_fmaf(ny_, q.iy, q.ny)), fmaxf(__builtin_fmaf(nz_, q.iz, q.nz), 0.0f));
                tf = fminf(fminf(__builtin_fmaf(fx_, q.ix, q.nx), __builtin_fmaf(fy_, q.iy, q.ny)), __builtin_fmaf(fz_, q.iz, q.nz));
            } else {
                const float x0 = __builtin_fmaf(a[c].x, q.ix, q.nx), x1 = __builtin_fmaf(b[c].x, q.ix, q.nx);
                const float y0 = __builtin_fmaf(a[c].y, q.iy, q.ny), y1 = __builtin_fmaf(b[c].y, q.iy, q.ny);
                const float z0 = __builtin_fmaf(a[c].z, q.iz, q.nz), z1 = __builtin_fmaf(b[c].z, q.iz, q.nz);
                tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.0f));
                tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
            }
            // box_entry32's widening (rtx_traverse.h): (1 -+ 2^-21) and the slack of a far origin
            tc[c] = __builtin_fmaf(tn, 1.0f - 4.76837158e-7f, -q.e);
            const float tf_hi = __builtin_fmaf(tf, 1.0f + 4.76837158e-7f, q.e);
            hm[c] = __builtin_amdgcn_ballot_w64(tc[c] <= fminf(tf_hi, best_up));   // (one compare: the mask comes straight out of v_cmp;
                                                                                    //  an empty slot's box is inverted: never entered)
        }
        if (best_up >= 0.0f) nbox += 4;
        // leaves first: their certain hits tighten the bound the interior children are then held against
        if (((cnt[0] | cnt[1] | cnt[2] | cnt[3]) & 0xFFFFu) != 0u) {       // some child is a leaf (or an empty slot: never entered)
            uint32_t leafmask = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (cnt[c] - 1u < 0xFFFFu && hm[c] != 0ull) leafmask |= 1u << c;             // wave-uniform
            while (leafmask != 0u) {
                const uint32_t c = (uint32_t)__builtin_ctz(leafmask);
                leafmask &= leafmask - 1u;
                const uint32_t first = c == 0 ? lnk[0] : (c == 1 ? lnk[1] : (c == 2 ? lnk[2] : lnk[3]));
                const uint32_t n = (c == 0 ? cnt[0] : (c == 1 ? cnt[1] : (c == 2 ? cnt[2] : cnt[3]))) & 0xFFFFu;
                const unsigned long long m = c == 0 ? hm[0] : (c == 1 ? hm[1] : (c == 2 ? hm[2] : hm[3]));
                const bool in = ((m >> lane) & 1ull) != 0ull;
                if (in) nleaf += n;
                for (uint32_t j = 0; j < n; ++j) {
                    const float4 rec = csph[first + j];                    // {c - centre, r}
                    const uint32_t prim = cprims[first + j];
                    if (!in) continue;
                    sph_packet_leaf_test(rec, prim, sr, lq, tid, best_up, qcnt, overflow);
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) hm[c] &= __builtin_amdgcn_ballot_w64(tc[c] <= best_up);   // the bound may have tightened
        }
        // interior children any lane still enters, nearest first by the first entering lane's distance (wave-uniform
        // integer keys -- the bits of a non-negative float order like the float; a far origin's slack can make a bound
        // negative: as a signed integer it still sorts in front --: scalar code); the farther ones are pushed
#ifndef RTX_SPK_FAST1
#define RTX_SPK_FAST1 2
#endif
#if RTX_SPK_FAST1
        // no or one interior child entered (the common case below the top levels): nothing to order, nothing to push -- the ordering
        // below is ~65 scalar instructions, and the packet walk is bound by the CU's scalar pipe
        {
            uint32_t ib = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) ib |= (cnt[c] == 0u && hm[c] != 0ull) ? 1u << c : 0u;
            if ((ib & (ib - 1u)) == 0u) {
                node = ib == 0u ? kNone : (ib == 1u ? lnk[0] : (ib == 2u ? lnk[1] : (ib == 4u ? lnk[2] : lnk[3])));
                if (node == kNone && sp != 0u) {
                    sp -= 1;
                    node = (uint32_t)__builtin_amdgcn_readlane(stk, (int)sp);
                }
                continue;
            }
#if RTX_SPK_FAST1 >= 2
            // two (20 % of the visits; none or one: 73 %): one comparison of the two first-entering lanes' distances, one push --
            // the same order as the network below gives (the later child goes first only when strictly nearer)
            if (__builtin_popcount(ib) == 2) {
                uint32_t far_link = 0;
#define RTX_SPK_PAIR(A, B)                                                                                                        \
                {                                                                                                                   \
                    const int ka = (int)__builtin_amdgcn_readlane(__float_as_uint(tc[A]), (int)__builtin_ctzll(hm[A]));            \
                    const int kb = (int)__builtin_amdgcn_readlane(__float_as_uint(tc[B]), (int)__builtin_ctzll(hm[B]));            \
                    const bool sw = kb < ka;                                                                                        \
                    node = sw ? lnk[B] : lnk[A];                                                                                    \
                    far_link = sw ? lnk[A] : lnk[B];                                                                                \
                }
                switch (ib) {
                    case 3u: RTX_SPK_PAIR(0, 1) break;
                    case 5u: RTX_SPK_PAIR(0, 2) break;
                    case 9u: RTX_SPK_PAIR(0, 3) break;
                    case 6u: RTX_SPK_PAIR(1, 2) break;
                    case 10u: RTX_SPK_PAIR(1, 3) break;
                    default: RTX_SPK_PAIR(2, 3) break;
                }
#undef RTX_SPK_PAIR
                stk = rtx_writelane((int)far_link, (int)sp, stk);
                sp += 1u;
                continue;
            }
#endif
        }
#endif
        int key[4];
        uint32_t kl[4];
        constexpr int kFar = 0x7F800000;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned long long m = cnt[c] == 0u ? hm[c] : 0ull;
            const int k = (int)__builtin_amdgcn_readlane(__float_as_uint(tc[c]), (int)__builtin_ctzll(m | 0x8000000000000000ull));
            key[c] = m != 0ull ? k : kFar;
            kl[c] = lnk[c];
        }
#define RTX_CSWAP(i, j) { if (key[j] < key[i]) { int tk = key[i]; key[i] = key[j]; key[j] = tk; uint32_t tl = kl[i]; kl[i] = kl[j]; kl[j] = tl; } }
        RTX_CSWAP(0, 1) RTX_CSWAP(2, 3) RTX_CSWAP(0, 2) RTX_CSWAP(1, 3) RTX_CSWAP(1, 2)
#undef RTX_CSWAP
        // three unconditional pushes, farthest first; an invalid one lands on the slot the next push (or a later one)
        // overwrites (sp <= 3 * depth + 2 <= 63: the index stays a lane of the register)
        stk = rtx_writelane((int)kl[3], (int)sp, stk); sp += key[3] < kFar ? 1u : 0u;
        stk = rtx_writelane((int)kl[2], (int)sp, stk); sp += key[2] < kFar ? 1u : 0u;
        stk = rtx_writelane((int)kl[1], (int)sp, stk); sp += key[1] < kFar ? 1u : 0u;
        node = key[0] < kFar ? kl[0] : kNone;
        if (node == kNone && sp != 0u) {
            sp -= 1;
            node = (uint32_t)__builtin_amdgcn_readlane(stk, (int)sp);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Tile lists: what the primary rays of an 8x8 pixel tile can hit, found ONCE per tile instead of once per packet.
//
// Every ray of a tile, whatever its sample's jitter, starts in the box O = cam_pos + [0, non_focal_offset]^3 and goes through the box
// T = hull of the tile's focal points + [0, focal_offset]^3 (scene.rs:202-205): p(u) = o + u (t - o), u >= 0, distance along the ray
// = u |t - o|.  With interval arithmetic per axis, p_a(u) lies in [omin_a + u (tmin_a - omax_a), omax_a + u (tmax_a - omin_a)] for
// every u >= 0 -- a "ray" with an interval origin and an interval direction, and a slab test against a node's box is six linear
// inequalities in u.  One thread per tile walks the tree with that test and writes down every sphere whose leaf box the beam can
// enter: {the leaf's f32 record, its index, a lower bound of the distance at which any ray of the tile can enter the box}, sorted
// by that bound.  C2: ~10 spheres per tile where a packet's walk visits ~33 nodes -- for each of the tile's 64 samples.
// The packet kernel then runs the walk's leaf test over the list, and stops at the first entry whose bound lies beyond every
// lane's nearest certain hit.  The list is a superset of what any walk of the tile would reach (the beam contains every ray; the
// boxes are the walk's own), the candidates' exact tests decide as before: same bits.  A tile whose list would exceed kTileListCap
// entries (the camera inside a cluster, focal_length ~ 0, NaNs), or whose walk exceeds the builder's stack, keeps the packet walk.
// (kTileListCap, TileEntry, TileLists: rtx_launch.h; the builder is the wave-per-tile kernel of rtx_wavefront.hip, which also serves
//  the meshes' lists: launch_build_sphere_tile_lists)

__global__ __launch_bounds__(kBvhThreads, kSpkWaves) void trace_sph_packet_kernel(const SceneView *__restrict__ svp,
                                                                                  const RowsView *__restrict__ rvp,
                                                                                  double *__restrict__ samples, Counters *__restrict__ ctr,
                                                                                  unsigned long long *__restrict__ work_counter,
                                                                                  const float4 *__restrict__ nodes, const LeafArrays la,
                                                                                  const SphQueue sq, const TileLists tl)
{
    const SceneView &sv = *svp;
    const RowsView &rv = *rvp;
    __shared__ uint32_t lds_q[2 * kSphQueue][kBvhThreads];
    uint32_t *const lq = &lds_q[0][0];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t bounce_limit = sv.max_bounces >= 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)sv.max_bounces + 1u;
    const unsigned long long n_rays = rv.n_rays;                        // a multiple of 64: the padded tile grid
    const PkConst4 cnodes = pk_const(nodes), csph = pk_const(la.sphere_f32);
    const PkConstU32 cprims = pk_const(la.sphere_prims);
    unsigned long long wave_next = 0, wave_end = 0;                     // this wave's share of the ray queue, in rays
    unsigned long long out_next = 0, out_end = 0;                       // its reserved slots of the survivors' queue
    unsigned long long segs = 0, box_tests = 0, leaf_filters = 0, exact = 0;

    // a packet over its tile's list is ~10 us of work, so the grabs are twice the launch's size for them (the host sizes rv.grab for
    // stage 2's longer segments): rank 0's band of the C2 frame at N = 8: stage 1 1.83 -> 1.44 ms (x4: 1.42, x8: 1.52); the full frame: level
#ifndef RTX_SPK_GRAB_MUL
#define RTX_SPK_GRAB_MUL 2
#endif
    const unsigned long long pk_grab = (unsigned long long)rv.grab * RTX_SPK_GRAB_MUL;
    for (;;) {
        if (wave_next >= wave_end) {
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(work_counter, pk_grab);     // (a multiple of 64)
            base = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
                   __builtin_amdgcn_readfirstlane((uint32_t)base);
            if (base >= n_rays) break;
            wave_next = base;
            wave_end = base + pk_grab < n_rays ? base + pk_grab : n_rays;
        }
        const unsigned long long my = wave_next + lane;
        wave_next += 64ull;
        // ---- render_pixel's prologue for the tile's 64 rays (scene.rs:196-207)
        uint32_t pl = 0, smp = 0;
        bool alive = ray_index_to_pixel_tiled(rv, my, pl, smp);         // false: the padding of a partial tile
        RayState r;
        if (alive) gen_primary(sv, rv, pl, rv.sample_begin + smp, r);
        else { r.pos = mk(0., 0., 0.); r.dir = mk(1., 0., 0.); r.result = mk(0., 0., 0.); r.light = mk(1., 1., 1.); r.key = 0; r.draw = 6; r.bounce = 0; }
        const uint32_t ridx = (uint32_t)my;                             // (the host keeps rv.n_rays below 2^32)
        const RayX rx = make_rayx(r.pos, r.dir);

        // ---- closest_object (scene.rs:243-251), phase 1: one walk for the tile, f32 only
        float best_up = -__builtin_inff();                              // -inf: this lane enters nothing
        uint32_t qcnt = 0, nbox = 0, nleaf = 0;
        bool overflow = false, walked = false;
        Ray32S q;
        SphereRay sr;
        q.ix = q.iy = q.iz = 1.0f; q.nx = q.ny = q.nz = 0.0f; q.e = 0.0f;
        sr.px = sr.py = sr.pz = sr.dx = sr.dy = sr.dz = sr.Kg = sr.K = 0.f; sr.c0 = __builtin_inff();
        if (alive) {
            const float omax = fmaxf(fmaxf(__builtin_fabsf((float)r.pos.x), __builtin_fabsf((float)r.pos.y)),
                                     __builtin_fabsf((float)r.pos.z));
            const bool in32 = omax <= sv.bvh_origin_limit;                                  // NaN origin -> exhaustive branch
            if (in32 || omax <= sv.bvh_origin_limit * kBvhRange64) {
                sphere_ray_from(sv, r.pos, r.dir, sr);
                Ray32 q0;
                make_ray32(r.pos, rx.dirn, (double)sv.bvh_inv_max, q0);
                q.ix = q0.ix; q.iy = q0.iy; q.iz = q0.iz; q.nx = q0.nx; q.ny = q0.ny; q.nz = q0.nz;
                q.e = ray32_slack(q0.nx, q0.ny, q0.nz, in32);           // 0 inside origin_limit: the bits of Ray32
                best_up = __builtin_inff();
                walked = true;
            }
        }
        const unsigned long long wm = __ballot(walked);
        // the tile's list, if it has one (wave-uniform: scalar loads)
        uint32_t list_n = kTileListWalk;
        const uint32_t tile_of = __builtin_amdgcn_readfirstlane((uint32_t)(((uint32_t)my - smp * rv.div_per_sample.d) >> 6));
        if (tl.count != nullptr && wm != 0ull) list_n = pk_const(tl.count)[tile_of];
        if (wm != 0ull && list_n != kTileListWalk) {
            const PkConst4 ent = pk_const(reinterpret_cast<const float4 *>(tl.entries + (size_t)tile_of * kTileListCap));
            for (uint32_t k = 0; k < list_n; ++k) {
                const float4 rec = ent[2 * k], tail = ent[2 * k + 1];
                const float t_lb = tail.y;
                const bool in = walked && t_lb <= best_up;
                if (__ballot(in) == 0ull) break;          // sorted by t_lb: nothing further can beat any lane's certain hit
                if (in) {
                    nleaf += 1;
                    sph_packet_leaf_test(rec, __float_as_uint(tail.x), sr, lq, tid, best_up, qcnt, overflow);
                }
            }
        } else if (wm != 0ull) {
            const uint32_t my_sgn = (q.ix < 0.0f ? 1u : 0u) | (q.iy < 0.0f ? 2u : 0u) | (q.iz < 0.0f ? 4u : 0u);
            const uint32_t sgn = __builtin_amdgcn_readlane(my_sgn, (int)(__ffsll((long long)wm) - 1));
            const uint32_t oct = __ballot(walked && my_sgn != sgn) == 0ull ? sgn : 8u;      // 8: the tile straddles an axis
#define RTX_SPK_WALK(S) sph_packet_walk<S>(cnodes, csph, cprims, sv.bvh_root, q, sr, lq, tid, lane, best_up, qcnt, overflow, nbox, nleaf)
            switch (oct) {
                case 0: RTX_SPK_WALK(0); break;
                case 1: RTX_SPK_WALK(1); break;
                case 2: RTX_SPK_WALK(2); break;
                case 3: RTX_SPK_WALK(3); break;
                case 4: RTX_SPK_WALK(4); break;
                case 5: RTX_SPK_WALK(5); break;
                case 6: RTX_SPK_WALK(6); break;
                case 7: RTX_SPK_WALK(7); break;
                default: RTX_SPK_WALK(8); break;
            }
#undef RTX_SPK_WALK
        }
        // ---- phase 2, f64, all lanes together: the exact tests of the candidates that can still be the winner, the shapes
        //      outside the tree, ray_hit
        uint32_t first_id = 0;
        if (alive) {
            Hit h;
            hit_init(h);
            ++segs;
            box_tests += nbox;
            leaf_filters += nleaf;
            if (walked && !overflow) {
#pragma unroll 1
                for (uint32_t e = 0; e < qcnt; ++e) {
                    if (__uint_as_float(lq[(size_t)(kSphQueue + e) * kBvhThreads + tid]) <= best_up) {
                        const uint32_t idx = lq[(size_t)e * kBvhThreads + tid];
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

            // ---- render_ray's match arm + ray_hit (scene.rs:232-239, 260-278)
            bool done = true;
            if (h.id != kNone) {
                advance_and_shade(sv, h, r);
                first_id = h.id;
                done = (r.bounce >= bounce_limit) || light_is_zero(r);            // scene.rs:227-228
            }
            if (done) {
                store_sample(samples, rv, ridx, r.result);
                alive = false;
            }
        }
        sph_queue_append(sq, ctr, alive, r, ridx, first_id, lane, out_next, out_end);
    }
    sph_queue_close(sq, lane, out_next, out_end);
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

#ifdef RTX_LAB
#include "rtx_bvh_spheres_lab.h"      // trace_sph_pool_kernel, trace_sph_pair_kernel, sph_sort_* : librtx_hip_lab.so only
#endif

uint32_t bvh_spheres_spill_entries(const SceneView &sv)
{
    const uint32_t need = 3u * sv.bvh_depth + 2u;       // a 4-wide node pushes at most 3 entries per level
    return need > (uint32_t)kSphStack ? need - (uint32_t)kSphStack : 0u;
}

size_t bvh_spheres_spill_bytes(const SceneView &sv, int n_cus)
{
#ifdef RTX_LAB
    const int lds_rows = kSlotStack < kPoolStack ? kSlotStack : kPoolStack;      // (the shortest LDS stack of the kernels that share the columns)
#else
    const int lds_rows = kSphStack;
#endif
    const uint32_t need = 3u * sv.bvh_depth + 2u;
    const uint32_t entries = need > (uint32_t)lds_rows ? need - (uint32_t)lds_rows : 0u;
    return (size_t)entries * (size_t)n_cus * kSphWavesPerSimd * kBvhThreads * sizeof(uint32_t);
}

// the survivors' queue of the two-stage form: one 64-byte record per slot; capacity = the launch's rays (every ray can
// survive: a closed scene) + the tail of one chunk per resident wave of stage 1, two u64 counters in front
static uint64_t sph_queue_capacity(uint64_t n_rays, int n_cus)
{
    const int wpc = kSphWavesPerSimd > kSpkWaves ? kSphWavesPerSimd : kSpkWaves;
    return n_rays + (uint64_t)n_cus * wpc * (kBvhThreads / 64) * kSphQueueChunk + kSphQueueChunk;
}

// the tile lists of a launch: a count per tile (padded to 256 bytes) + kTileListCap entries of 32 bytes per tile
size_t bvh_spheres_tile_list_bytes(uint64_t rays_per_sample)
{
    const uint64_t n_tiles = rays_per_sample >> 6;
    return (size_t)(((n_tiles * sizeof(uint32_t) + 255) & ~(uint64_t)255) + n_tiles * kTileListCap * sizeof(TileEntry));
}

size_t bvh_spheres_queue_bytes(uint64_t n_rays, int n_cus)
{
    const uint64_t cap = sph_queue_capacity(n_rays, n_cus);
    size_t bytes = (size_t)(cap * sizeof(SphSurvivor) + 2 * 256);
#ifdef RTX_LAB
    bytes += (size_t)((cap * 5 + 255) & ~(uint64_t)255) + 4 * 256 + 256;      // + the lab sort's keys, permutation, histogram
#endif
    return bytes;
}

// may stage 1 walk as packets?  The ray queue in 8x8 tiles and a tree the wave-uniform stack holds.
static bool sph_packets_ok(const SceneView &sv, bool tiled)
{
    return tiled && 3u * sv.bvh_depth + 2u <= (uint32_t)kSpkStack;
}

bool bvh_spheres_two_stage_ok(const SceneView &sv, bool tiled)
{
#ifdef RTX_LAB
    (void)sv; (void)tiled;
    return true;                                         // (per-lane primary rays, MODE 1, when packets cannot run)
#else
    return sph_packets_ok(sv, tiled);
#endif
}

hipError_t launch_trace_bvh_spheres(const SceneView *d_sv, const SceneView &sv, const RowsView *d_rv, const RowsView &rv,
                                    double *samples, Counters *counters, unsigned long long *work_counter, uint32_t *spill, int n_cus,
                                    void *queue_mem, uint32_t flags, hipStream_t stream, Counters *stage1_snapshot, hipEvent_t stage1_done,
                                    void *pool_mem, void *slots_mem, void *tile_list_mem, bool build_tile_lists)
{
    const uint64_t want = (rv.n_rays + kBvhThreads - 1) / kBvhThreads;
    const uint64_t cap = (uint64_t)n_cus * kSphWavesPerSimd;
    const uint32_t blocks = (uint32_t)(want < cap ? want : cap);
    if (blocks == 0) return hipSuccess;
    LeafArrays la;
    la.sphere_f32 = sv.bvh_leaf_cr; la.sphere_prims = sv.bvh_prims; la.spheres = sv.spheres; la.sphere_ids = sv.sphere_id;
    la.tri_f32 = sv.tri_f32; la.tri_fidx = sv.tri_fidx; la.tris = sv.tris;
    const uint32_t spill_entries = spill ? bvh_spheres_spill_entries(sv) : 0u;
    const bool deep = spill_entries != 0u;
    const float4 *nodes = reinterpret_cast<const float4 *>(sv.bvh_nodes);
    SphQueue sq{};
    sq.cut_walkers = (sv.tuning & RTX_TUNE_NO_CUT) != 0u ? 0u : kSphCutWalkers;
    const float4 *q3nodes = reinterpret_cast<const float4 *>(sv.bvh_q3nodes);
#ifdef RTX_LAB
    const bool q3 = (sv.bvh_flags & 16u) != 0u && sv.bvh_q3nodes != nullptr && (sv.tuning & RTX_TUNE_NO_QNODES) == 0u;
    const bool inl = (sv.tuning & RTX_TUNE_INLINE_LEAVES) != 0u;
#else
    // the product holds the 64-byte-node instances only (the API sends a sphere tree without that form to the LDS sweep)
    if ((sv.bvh_flags & 16u) == 0u || sv.bvh_q3nodes == nullptr) return hipErrorInvalidValue;
    (void)pool_mem; (void)flags;
#endif
    const bool tiled = rv.tiles_x != 0u && (rv.n_rays & 63ull) == 0ull;
    if (!queue_mem || sv.max_bounces == 0) {     // one stage (small launches): per-lane walks over the 64-byte nodes as in stage 2
#ifdef RTX_LAB
        auto kernel = !q3 ? (deep ? trace_bvh_spheres_kernel<true, 0> : trace_bvh_spheres_kernel<false, 0>)
                          : inl ? (deep ? trace_bvh_spheres_kernel<true, 0, 1> : trace_bvh_spheres_kernel<false, 0, 1>)
                                : (deep ? trace_bvh_spheres_kernel<true, 0, 2> : trace_bvh_spheres_kernel<false, 0, 2>);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kBvhThreads), 0, stream, d_sv, d_rv, samples, counters, work_counter,
                           q3 ? q3nodes : nodes, la, spill, spill_entries, sq);
#else
        auto kernel = deep ? trace_bvh_spheres_kernel<true, 0, 2> : trace_bvh_spheres_kernel<false, 0, 2>;
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kBvhThreads), 0, stream, d_sv, d_rv, samples, counters, work_counter,
                           q3nodes, la, spill, spill_entries, sq);
#endif
        return hipGetLastError();
    }
    // ---- two stages: carve the queue (256-byte boundaries), zero its two counters
    const uint64_t capacity = sph_queue_capacity(rv.n_rays, n_cus);
    char *p = static_cast<char *>(queue_mem);
    unsigned long long *ctrs = reinterpret_cast<unsigned long long *>(p);
    sq.rec = reinterpret_cast<SphSurvivor *>(p + 256);
    sq.count = ctrs;
    sq.capacity = capacity;
    hipError_t e = hipMemsetAsync(ctrs, 0, 2 * sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
#ifdef RTX_LAB
    if ((flags & kSphNoPackets) != 0u || !sph_packets_ok(sv, tiled)) {
        auto k1 = deep ? trace_bvh_spheres_kernel<true, 1> : trace_bvh_spheres_kernel<false, 1>;
        hipLaunchKernelGGL(k1, dim3(blocks), dim3(kBvhThreads), 0, stream, d_sv, d_rv, samples, counters, work_counter, nodes, la, spill,
                           spill_entries, sq);
    } else
#else
    if (!sph_packets_ok(sv, tiled)) return hipErrorInvalidValue;      // (bvh_spheres_two_stage_ok() keeps the caller from asking)
#endif
    {
        const uint64_t pcap = (uint64_t)n_cus * kSpkWaves;
        const uint32_t pblocks = (uint32_t)(want < pcap ? want : pcap);
        TileLists tl{};
        if (tile_list_mem && tiled && (sv.tuning & RTX_TUNE_NO_TILE_LISTS) == 0u) {
            // what each tile's primary rays can hit, once per tile (build_tile_lists_kernel); the packets then run over the lists
            const uint32_t n_tiles = (uint32_t)((rv.n_rays / rv.n_samples) >> 6);
            tl.count = reinterpret_cast<uint32_t *>(tile_list_mem);
            tl.entries = reinterpret_cast<TileEntry *>(static_cast<char *>(tile_list_mem) + (((size_t)n_tiles * sizeof(uint32_t) + 255) & ~(size_t)255));
            if (build_tile_lists && (e = launch_build_sphere_tile_lists(d_sv, d_rv, sv, n_tiles, tl.count, tl.entries, stream)) != hipSuccess) return e;
        }
        hipLaunchKernelGGL(trace_sph_packet_kernel, dim3(pblocks), dim3(kBvhThreads), 0, stream, d_sv, d_rv, samples, counters, work_counter,
                           nodes, la, sq, tl);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (stage1_snapshot && (e = hipMemcpyAsync(stage1_snapshot, counters, sizeof(Counters) * kCounterShards, hipMemcpyDeviceToDevice, stream)) != hipSuccess) return e;
    if (stage1_done && (e = hipEventRecord(stage1_done, stream)) != hipSuccess) return e;
#ifdef RTX_LAB
    if (hipError_t le = launch_sph_lab_stage2(d_sv, sv, d_rv, rv, samples, counters, spill, spill_entries, n_cus, p, capacity, ctrs, sq, la, nodes,
                                              flags, pool_mem, stream); le != hipErrorNotReady)
        return le;                                       // a lab form of stage 2 ran (or failed); hipErrorNotReady: none was asked for
    if (!q3 || inl) {
        auto k2 = !q3 ? (deep ? trace_bvh_spheres_kernel<true, 2> : trace_bvh_spheres_kernel<false, 2>)
                      : (deep ? trace_bvh_spheres_kernel<true, 2, 1> : trace_bvh_spheres_kernel<false, 2, 1>);
        hipLaunchKernelGGL(k2, dim3(blocks), dim3(kBvhThreads), 0, stream, d_sv, d_rv, samples, counters, ctrs + 1, q3 ? q3nodes : nodes, la, spill,
                           spill_entries, sq);
        return hipGetLastError();
    }
#endif
#ifdef RTX_LAB
    if (slots_mem) {                                     // stage 2 over ray slots: the grid is the resident waves, each with its own slots
        if (!q3) return hipErrorInvalidValue;
        const uint32_t se = spill ? bvh_spheres_slots_spill_entries(sv) : 0u;
        if (bvh_spheres_slots_spill_entries(sv) != 0u && !spill) return hipErrorInvalidValue;
        auto ks = se ? trace_sph_slots_kernel<true> : trace_sph_slots_kernel<false>;
        hipLaunchKernelGGL(ks, dim3((uint32_t)cap), dim3(kBvhThreads), 0, stream, d_sv, d_rv, samples, counters, ctrs + 1, q3nodes, la, spill, se, sq,
                           reinterpret_cast<SlotRec *>(slots_mem));
        return hipGetLastError();
    }
#else
    (void)slots_mem;
#endif
    // the queue-fed stage walks per lane over the 64-byte nodes: half the L1 requests per visit
    auto k2 = deep ? trace_bvh_spheres_kernel<true, 2, 2> : trace_bvh_spheres_kernel<false, 2, 2>;
    hipLaunchKernelGGL(k2, dim3(blocks), dim3(kBvhThreads), 0, stream, d_sv, d_rv, samples, counters, ctrs + 1, q3nodes, la, spill, spill_entries, sq);
    return hipGetLastError();
}

}  // namespace rtx
