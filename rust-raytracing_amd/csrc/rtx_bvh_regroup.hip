// rtx_bvh_regroup.hip -- trace_bvh_regroup_kernel: the flat-BVH traversal of rtx_bvh.hip under a different schedule.
//
// Same rays, same traversal step (bvh_step, rtx_traverse.h), same exact tests, same bits.  trace_bvh_kernel walks
// a wave through its segments in lock-step: every lane waits for the wave's longest traversal before any lane
// shades.  That is the right trade for sphere scenes (C2: 47 wave-steps per 24 lane-steps per segment), but on large
// triangle meshes the traversal lengths are heavy-tailed -- most segments are the reference's self-hits after a bounce
// (a dozen steps), a few cross a thousand footprints -- and the lock-step wave spends 86 % of its lane-steps
// idle (C5: 811 wave-steps per 117 lane-steps per segment).  Here a lane is in one of three states and the wave
// alternates between two phases:
//     TRAV  walking the tree              -- phase B: one bvh_step for every TRAV lane, repeated until phase A is due
//     FIN   traversal complete            -- phase A: remaining exact tests, planes / shapes outside the tree, ray_hit,
//     IDLE  no ray                                     refill from the queue, set up the next segment
// Phase A runs only when at least `thresh` lanes are waiting for it (or nobody is traversing), so its f64 code runs
// with that many lanes active.  The price: more lanes per step = more leaf code per step and a longer wait for the
// slowest of more node fetches, and the shading phase runs more often; DESIGN.md 3.2 has the measurements that decide
// when AUTO takes this kernel.
#include "rtx_launch.h"
#include "rtx_traverse.h"

#include <cstdlib>

namespace rtx {

template <bool TRIS, bool SPILL>
__global__ __launch_bounds__(kBvhThreads, kBvhWavesPerSimd) void trace_bvh_regroup_kernel(const SceneView *__restrict__ svp,
                                                                   const RowsView *__restrict__ rvp,
                                                                   double *__restrict__ samples, Counters *__restrict__ ctr,
                                                                   unsigned long long *__restrict__ work_counter,
                                                                   const float4 *__restrict__ nodes, const LeafArrays la,
                                                                   uint32_t *__restrict__ spill, uint32_t spill_entries,
                                                                   uint32_t thresh)
{
    const SceneView &sv = *svp;
    const RowsView &rv = *rvp;
    __shared__ uint32_t lds_stack[kBvh4StackEntries + 1][kBvhThreads];      // + the sink row of bvh_step's pushes
    __shared__ uint32_t lds_q[kBvhQueue][kBvhThreads];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const size_t spill_stride = (size_t)gridDim.x * kBvhThreads, glane = (size_t)blockIdx.x * kBvhThreads + tid;
    const uint32_t bounce_limit = sv.max_bounces >= 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)sv.max_bounces + 1u;
    enum : uint32_t { S_IDLE = 0, S_TRAV = 1, S_FIN = 2, S_SETUP = 3 };

    unsigned long long wave_next = 0, wave_end = 0;      // this wave's share of the ray queue (wave-uniform)
    bool queue_empty = false;
    uint32_t state = S_IDLE;
    RayState r;
    RayX rx;
    Hit h;
    Ray32 q;
    FilterParams fpar;
    TriFilterParams tpar;
    float best_up = 0.f;
    uint32_t node = kNone, sp = 0, qcnt = 0, nbox = 0, nleaf = 0;
    bool overflow = false, tree_used = false;
    uint32_t ridx = 0, wave_step = 0;        // ridx: the ray's index in the launch's queue = where its sample goes
    unsigned long long segs = 0, box_tests = 0, leaf_filters = 0, exact = 0;
    hit_init(h);
    filter_idle(fpar);
    tri_filter_idle(tpar);
    q.ix = q.iy = q.iz = q.nx = q.ny = q.nz = 0.f;
    rx = make_rayx(mk(0.0, 0.0, 0.0), mk(1.0, 0.0, 0.0));
#ifdef RTX_BVH_STATS
    unsigned long long wave_steps = 0, cyc_trav = 0, cyc_other = 0, cyc_mark = __builtin_amdgcn_s_memtime();
#define RTX_MARK(acc) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - cyc_mark; cyc_mark = now_; }
#else
#define RTX_MARK(acc)
#endif

    for (;;) {
        // ================= phase A (entered when the inner loop below finds it due) =================
        if (__ballot(state != S_IDLE) == 0ull && queue_empty) break;       // wave-uniform: nothing live, nothing left to take
        if (state == S_FIN) {
            // ---- the rest of closest_object (scene.rs:243-251) for this segment
            flush_candidates<TRIS>(la, rx, &lds_q[0][0], tid, qcnt, h, best_up, exact);
            box_tests += nbox;
            leaf_filters += nleaf;
            nbox = 0; nleaf = 0;
            const bool covered = tree_used && !overflow;          // (a dropped subtree: every shape gets the exact test)
            if (!covered || (sv.bvh_flags & 1u) == 0u) {
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
            // triangles outside the tree, by filter record (triangles without a record can never be hit: rtx_api.hip)
            const uint32_t tri_sweep_from = (covered && (sv.bvh_flags & 2u)) ? sv.n_tri_tree : 0u;
            for (uint32_t k = tri_sweep_from; k < sv.n_tri_filter; ++k) {
                const uint32_t tk = la.tri_fidx[k];
                double t;
                if (triangle_distance(la.tris[tk], rx, &t)) hit_consider(h, t, la.tris[tk].id, 2, tk);
            }
            exact += sv.n_planes + (sv.n_tri_filter - tri_sweep_from);
            // ---- render_ray's match arm + ray_hit (scene.rs:232-239, 260-278)
            bool done = true;
            if (h.id != kNone) {
                advance_and_shade(sv, h, r);
                done = (r.bounce >= bounce_limit) || light_is_zero(r);            // scene.rs:227-228
            }
            if (done) {
                store_sample(samples, rv, ridx, r.result);
                state = S_IDLE;
            } else {
                state = S_SETUP;
            }
        }
        // ---- hand rays to idle lanes: ballot + prefix sum over the wave's local range, one atomic per rv.grab rays
        for (;;) {
            const unsigned long long idle_mask = __ballot(state == S_IDLE);
            if (idle_mask == 0ull || queue_empty) break;
            if (wave_next >= wave_end) {
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(work_counter, (unsigned long long)rv.grab);
                base = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
                       __builtin_amdgcn_readfirstlane((uint32_t)base);
                wave_next = base;
                wave_end = base + rv.grab < rv.n_rays ? base + rv.grab : rv.n_rays;
                if (base >= rv.n_rays) { queue_empty = true; wave_next = wave_end = 0; break; }
            }
            if (state == S_IDLE) {
                const unsigned long long my = wave_next + bvh_mbcnt(idle_mask);
                bool valid = my < wave_end;
                uint32_t pl = 0, smp = 0;
                if (valid) {
                    if (rv.tiles_x != 0u) valid = ray_index_to_pixel_tiled(rv, my, pl, smp);
                    else ray_index_to_pixel(rv, my, pl, smp);
                }
                if (valid) {
                    gen_primary(sv, rv, pl, rv.sample_begin + smp, r);
                    state = S_SETUP;
                    ridx = (uint32_t)my;                                          // (the host keeps rv.n_rays below 2^32)
                    if (sv.n_objects == 0) {                                      // scene.rs:224-226
                        store_sample(samples, rv, ridx, mk(0.0, 0.0, 0.0));
                        state = S_IDLE;
                    }
                }
            }
            const unsigned long long taken = (unsigned long long)__popcll(idle_mask);
            wave_next = wave_next + taken < wave_end ? wave_next + taken : wave_end;
        }
        // ---- set up the next segment
        if (state == S_SETUP) {
            rx = make_rayx(r.pos, r.dir);
            hit_init(h);
            ++segs;
            qcnt = 0; sp = 0; overflow = false;
            best_up = __builtin_inff();
            const float omax = fmaxf(fmaxf(__builtin_fabsf((float)r.pos.x), __builtin_fabsf((float)r.pos.y)),
                                     __builtin_fabsf((float)r.pos.z));
            tree_used = sv.n_bvh_nodes != 0 && omax <= sv.bvh_origin_limit;          // NaN origin -> exhaustive branch
            if (tree_used || (sv.n_bvh_nodes != 0 && omax <= sv.bvh_origin_limit * kBvhRange64)) {
                if (sv.bvh_flags & 1u) filter_from_ray(sv, r.pos, r.dir, fpar); else filter_idle(fpar);
                if (TRIS && (sv.bvh_flags & 2u)) tri_filter_from_ray(sv, r.pos, r.dir, tpar); else tri_filter_idle(tpar);
            }
            if (tree_used) {
                make_ray32(r.pos, rx.dirn, (double)sv.bvh_inv_max, q);
                node = sv.bvh_root;              // wide node 0 (flagged when it is a footprint node)
                state = S_TRAV;
            } else {
                if (sv.n_bvh_nodes != 0 && omax <= sv.bvh_origin_limit * kBvhRange64) {
                    // origin far outside the scene (rare): the whole walk right here, with the f64 slab test
                    Ray64 q64;
                    make_ray64(r.pos, rx.dirn, (double)sv.bvh_inv_max, q64);
                    unsigned long long unused_steps = 0;
                    bvh_traverse<TRIS, SPILL>(nodes, la, q64, fpar, tpar, rx, sv.bvh_root, overflow, h, &lds_stack[0][0], &lds_q[0][0], tid, spill,
                                              spill_entries, spill_stride, glane, box_tests, leaf_filters, exact, unused_steps);
                    tree_used = true;
                }
                node = kNone;
                state = S_FIN;
            }
        }
        // ================= phase B: traversal steps until phase A is due again =================
        // (an inner loop of its own, so that the ray state phase A works on stays out of the traversal's registers)
        RTX_MARK(cyc_other)
        for (;;) {
#ifdef RTX_BVH_STATS
            wave_steps += 1;
#endif
            if (state == S_TRAV) {
                bvh_step<TRIS, SPILL>(nodes, la, q, fpar, tpar, rx, node, sp, qcnt, overflow, h, best_up, &lds_stack[0][0], &lds_q[0][0],
                                      tid, spill, spill_entries, spill_stride, glane, nbox, nleaf, exact);
                if (node == kNone) state = S_FIN;
            }
            // the queued candidates' exact f64 tests, every 4th step for all lanes together (the pruning bound lags by
            // at most 4 steps, which only costs visits)
            wave_step += 1;
            if ((wave_step & 3u) == 0u && qcnt != 0u) flush_candidates<TRIS>(la, rx, &lds_q[0][0], tid, qcnt, h, best_up, exact);
            const unsigned long long m_trav = __ballot(state == S_TRAV);
            const uint32_t waiting = (uint32_t)__popcll(__ballot(state == S_FIN)) +
                                     (queue_empty ? 0u : (uint32_t)__popcll(__ballot(state == S_IDLE)));
            if (waiting >= thresh || m_trav == 0ull) break;
        }
        RTX_MARK(cyc_trav)
    }
    unsigned long long filt = box_tests + leaf_filters;
#ifdef RTX_BVH_STATS
    RTX_MARK(cyc_other)
    exact = lane == 0 ? wave_steps : 0ull;
    filt = lane == 0 ? cyc_trav : 0ull;
    box_tests = lane == 0 ? cyc_other : 0ull;
#endif
#undef RTX_MARK
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

#ifndef RTX_BVH_THRESH
#define RTX_BVH_THRESH 16
#endif

hipError_t launch_trace_bvh_regroup(const SceneView *d_sv, const SceneView &sv, const RowsView *d_rv, const RowsView &rv,
                                    double *samples, Counters *counters, unsigned long long *work_counter, uint32_t *spill,
                                    int n_cus, hipStream_t stream)
{
    const uint64_t want = (rv.n_rays + kBvhThreads - 1) / kBvhThreads;
    const uint64_t cap = (uint64_t)n_cus * kBvhWavesPerSimd;
    const uint32_t blocks = (uint32_t)(want < cap ? want : cap);
    if (blocks == 0) return hipSuccess;
    const uint32_t thresh = [&] {                        // tuning knob
        const uint32_t t = (sv.tuning >> RTX_TUNE_THRESH_SHIFT) & 127u;
        long v = t ? (long)t : RTX_BVH_THRESH;
        return (uint32_t)(v < 1 ? 1 : (v > 64 ? 64 : v));
    }();
    LeafArrays la;
    la.sphere_f32 = sv.bvh_leaf_f32; la.sphere_prims = sv.bvh_prims; la.spheres = sv.spheres; la.sphere_ids = sv.sphere_id;
    la.tri_f32 = sv.tri_f32; la.tri_fidx = sv.tri_fidx; la.tris = sv.tris;
    const uint32_t spill_entries = spill ? bvh_spill_entries(sv) : 0u;
    const bool tris = (sv.bvh_flags & 2u) != 0u, deep = spill_entries != 0u;
    auto kernel = tris ? (deep ? trace_bvh_regroup_kernel<true, true> : trace_bvh_regroup_kernel<true, false>)
                       : (deep ? trace_bvh_regroup_kernel<false, true> : trace_bvh_regroup_kernel<false, false>);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kBvhThreads), 0, stream, d_sv, d_rv, samples, counters, work_counter,
                       reinterpret_cast<const float4 *>(sv.bvh_nodes), la, spill, spill_entries, thresh);
    return hipGetLastError();
}

}  // namespace rtx
