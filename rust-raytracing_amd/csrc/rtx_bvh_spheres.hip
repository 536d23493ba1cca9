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
//   * trace_sph_pool_kernel, trace_sph_pair_kernel, sph_sort_*   experiments behind RtxConfig.tuning bits (slower or on par)
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
#define RTX_SPH_CUT 16
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
#ifdef RTX_SPH_PROFILE
    unsigned long long rtx_prof = 0;
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
#ifdef RTX_SPH_PROFILE
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
#ifdef RTX_SPH_PROFILE
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
constexpr uint32_t kSphNoPackets = 1u;       // launch flag: stage 1 per lane (A/B runs)
constexpr uint32_t kSphPair = 8u;            // launch flag: stage 2 with two rays per lane (trace_sph_pair_kernel)
constexpr uint32_t kSphPool = 4u;            // launch flag: stage 2 as the wave-local pool (an experiment: slower) instead of lock-step
constexpr uint32_t kSphSortSurvivors = 2u;   // launch flag: stage 2 reads the survivors ordered by exit distance and octant

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
// v_writelane_b32 (this clang has no __builtin_amdgcn_writelane; the LLVM intrinsic is reached by its name)
extern "C" __device__ int rtx_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

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
#if defined(RTX_SPK_LAB) && RTX_SPK_LAB >= 10         // lab build: visits by the number of interior children entered (through nleaf of lane 0)
            if (lane == 0 && (uint32_t)__builtin_popcount(ib) == (uint32_t)(RTX_SPK_LAB - 10)) nleaf += 1000000u;
#endif
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

__global__ __launch_bounds__(kBvhThreads, kSpkWaves) void trace_sph_packet_kernel(const SceneView *__restrict__ svp,
                                                                                  const RowsView *__restrict__ rvp,
                                                                                  double *__restrict__ samples, Counters *__restrict__ ctr,
                                                                                  unsigned long long *__restrict__ work_counter,
                                                                                  const float4 *__restrict__ nodes, const LeafArrays la,
                                                                                  const SphQueue sq)
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

    for (;;) {
        if (wave_next >= wave_end) {
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(work_counter, (unsigned long long)rv.grab);     // (rv.grab is a multiple of 64)
            base = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
                   __builtin_amdgcn_readfirstlane((uint32_t)base);
            if (base >= n_rays) break;
            wave_next = base;
            wave_end = base + rv.grab < n_rays ? base + rv.grab : n_rays;
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
#if defined(RTX_SPK_LAB) && RTX_SPK_LAB == 1            // lab timing build: no walk (wrong images)
        const unsigned long long wm = 0ull;
#else
        const unsigned long long wm = __ballot(walked);
#endif
        if (wm != 0ull) {
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
#if defined(RTX_SPK_LAB) && RTX_SPK_LAB == 2            // lab timing build: no f64 phase after the walk (wrong images)
        if (alive && best_up == 12345.0f) {
#else
        if (alive) {
#endif
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

uint32_t bvh_spheres_spill_entries(const SceneView &sv)
{
    const uint32_t need = 3u * sv.bvh_depth + 2u;       // a 4-wide node pushes at most 3 entries per level
    return need > (uint32_t)kSphStack ? need - (uint32_t)kSphStack : 0u;
}

static uint32_t bvh_spheres_pool_spill_entries(const SceneView &sv)
{
    const uint32_t need = 3u * sv.bvh_depth + 2u;
    return need > (uint32_t)kPoolStack ? need - (uint32_t)kPoolStack : 0u;
}

size_t bvh_spheres_spill_bytes(const SceneView &sv, int n_cus)
{
    return (size_t)bvh_spheres_pool_spill_entries(sv) * (size_t)n_cus * kSphWavesPerSimd * kBvhThreads * sizeof(uint32_t);   // (the pool kernel's stack is the shorter one)
}

// the survivors' queue of the two-stage form: one 64-byte record per slot; capacity = the launch's rays (every ray can
// survive: a closed scene) + the tail of one chunk per resident wave of stage 1, two u64 counters in front
static uint64_t sph_queue_capacity(uint64_t n_rays, int n_cus)
{
    const int wpc = kSphWavesPerSimd > kSpkWaves ? kSphWavesPerSimd : kSpkWaves;
    return n_rays + (uint64_t)n_cus * wpc * (kBvhThreads / 64) * kSphQueueChunk + kSphQueueChunk;
}

size_t bvh_spheres_queue_bytes(uint64_t n_rays, int n_cus)
{
    const uint64_t cap = sph_queue_capacity(n_rays, n_cus);
    return (size_t)(cap * sizeof(SphSurvivor) + 2 * 256) + (size_t)((cap * 5 + 255) & ~(uint64_t)255) + 4 * 256 + 256;   // + the sort's keys, permutation, histogram
}

// may stage 1 walk as packets?  The ray queue in 8x8 tiles and a tree the wave-uniform stack holds.
static bool sph_packets_ok(const SceneView &sv, const RowsView &rv)
{
    return rv.tiles_x != 0u && (rv.n_rays & 63ull) == 0ull && 3u * sv.bvh_depth + 2u <= (uint32_t)kSpkStack;
}

hipError_t launch_trace_bvh_spheres(const SceneView *d_sv, const SceneView &sv, const RowsView *d_rv, const RowsView &rv,
                                    double *samples, Counters *counters, unsigned long long *work_counter, uint32_t *spill, int n_cus,
                                    void *queue_mem, uint32_t flags, hipStream_t stream, Counters *stage1_snapshot, hipEvent_t stage1_done,
                                    void *pool_mem)
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
    const bool q3 = (sv.bvh_flags & 16u) != 0u && sv.bvh_q3nodes != nullptr && (sv.tuning & RTX_TUNE_NO_QNODES) == 0u;
    const bool inl = (sv.tuning & RTX_TUNE_INLINE_LEAVES) != 0u;
    const float4 *q3nodes = reinterpret_cast<const float4 *>(sv.bvh_q3nodes);
    if (!queue_mem || sv.max_bounces == 0) {     // one stage (small launches): per-lane walks over the 64-byte nodes as in stage 2
        auto kernel = !q3 ? (deep ? trace_bvh_spheres_kernel<true, 0> : trace_bvh_spheres_kernel<false, 0>)
                          : inl ? (deep ? trace_bvh_spheres_kernel<true, 0, 1> : trace_bvh_spheres_kernel<false, 0, 1>)
                                : (deep ? trace_bvh_spheres_kernel<true, 0, 2> : trace_bvh_spheres_kernel<false, 0, 2>);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kBvhThreads), 0, stream, d_sv, d_rv, samples, counters, work_counter,
                           q3 ? q3nodes : nodes, la, spill, spill_entries, sq);
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
    if ((flags & kSphNoPackets) == 0u && sph_packets_ok(sv, rv)) {
        const uint64_t pcap = (uint64_t)n_cus * kSpkWaves;
        const uint32_t pblocks = (uint32_t)(want < pcap ? want : pcap);
        hipLaunchKernelGGL(trace_sph_packet_kernel, dim3(pblocks), dim3(kBvhThreads), 0, stream, d_sv, d_rv, samples, counters, work_counter,
                           nodes, la, sq);
    } else {
        auto k1 = deep ? trace_bvh_spheres_kernel<true, 1> : trace_bvh_spheres_kernel<false, 1>;
        hipLaunchKernelGGL(k1, dim3(blocks), dim3(kBvhThreads), 0, stream, d_sv, d_rv, samples, counters, work_counter, nodes, la, spill,
                           spill_entries, sq);
    }
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (stage1_snapshot && (e = hipMemcpyAsync(stage1_snapshot, counters, sizeof(Counters) * kCounterShards, hipMemcpyDeviceToDevice, stream)) != hipSuccess) return e;
    if (stage1_done && (e = hipEventRecord(stage1_done, stream)) != hipSuccess) return e;
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
    if (pool_mem && (flags & kSphPair) != 0u) {
        // stage 2 with two rays per lane (trace_sph_pair_kernel)
        SphPair pp{};
        const size_t lanes2 = (size_t)n_cus * kSphWavesPerSimd * kBvhThreads * 2;
        pp.stride = lanes2;
        pp.f = reinterpret_cast<double *>(pool_mem);
        pp.u = reinterpret_cast<uint32_t *>(pp.f + (size_t)kPairF64 * lanes2);
        const bool pq3 = (sv.bvh_flags & 16u) != 0u && sv.bvh_q3nodes != nullptr && (sv.tuning & RTX_TUNE_NO_QNODES) == 0u;
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
        const bool pq3 = (sv.bvh_flags & 16u) != 0u && sv.bvh_q3nodes != nullptr && (sv.tuning & RTX_TUNE_NO_QNODES) == 0u;
        auto kp = pq3 ? (pool_spill ? trace_sph_pool_kernel<true, true> : trace_sph_pool_kernel<false, true>)
                      : (pool_spill ? trace_sph_pool_kernel<true, false> : trace_sph_pool_kernel<false, false>);
        hipLaunchKernelGGL(kp, dim3((uint32_t)cap), dim3(kBvhThreads), 0, stream, d_sv, d_rv, samples, counters, ctrs + 1,
                           pq3 ? reinterpret_cast<const float4 *>(sv.bvh_q3nodes) : nodes, la, spill, pool_spill, sq, pool);
        return hipGetLastError();
    }
    if (q3) {       // the queue-fed stage walks per lane over the 64-byte nodes: half the L1 requests per visit
        auto k2 = inl ? (deep ? trace_bvh_spheres_kernel<true, 2, 1> : trace_bvh_spheres_kernel<false, 2, 1>)
                      : (deep ? trace_bvh_spheres_kernel<true, 2, 2> : trace_bvh_spheres_kernel<false, 2, 2>);
        hipLaunchKernelGGL(k2, dim3(blocks), dim3(kBvhThreads), 0, stream, d_sv, d_rv, samples, counters, ctrs + 1,
                           q3nodes, la, spill, spill_entries, sq);
        return hipGetLastError();
    }
    auto k2 = deep ? trace_bvh_spheres_kernel<true, 2> : trace_bvh_spheres_kernel<false, 2>;
    hipLaunchKernelGGL(k2, dim3(blocks), dim3(kBvhThreads), 0, stream, d_sv, d_rv, samples, counters, ctrs + 1, nodes, la, spill,
                       spill_entries, sq);
    return hipGetLastError();
}

}  // namespace rtx
