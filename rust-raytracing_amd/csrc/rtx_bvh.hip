// rtx_bvh.hip -- trace_bvh_kernel: closest_object (scene.rs:243-251) by flat-BVH traversal over spheres and triangles.
//
// Persistent waves; each lane owns one ray for its whole life (ray state in registers).  A wave takes rays
// from the global queue 512 at a time (one atomic per rv.grab rays) and hands them to idle lanes with
// ballot + mbcnt, so lanes whose path ended are refilled at once.  Per segment each lane walks the BVH with its
// own LDS stack (depth-major layout: no bank conflicts).  Nodes are 4-wide (128 B: the boxes of up to four
// children inline), so a step is ONE dependent fetch; children are visited nearest first and boxes whose entry
// distance exceeds the best hit so far are pruned.
//
// Exactness.  The tree only decides WHICH shapes get the exact f64 test (sphere.rs:19-30, triangle.rs:108-127); it
// never decides a hit.  (1) Boxes: f32, rounded outward, inflated (rtx_bvh.h) -- 3-D for spheres, the (x, y)
// footprint with unbounded z for triangles, which is what keeps the reference's phantom hits (rtx_bvh.h); the f32
// slab test widens its interval by the relative rounding error, so it cannot reject a box that contains a
// reportable hit point.  (2) Leaves: each shape first passes the conservative f32 filter of the sweep kernel
// (rtx_device.h: sphere discriminant / triangle footprint at q), survivors get the exact test.  (3) Pruning keeps
// ties (entry <= best), and the winner is the lexicographic minimum of (t, scene index) -- the reference's
// first-minimal rule.  Planes (unbounded, plane.rs) and shapes the tree does not hold are tested for every segment.
// Rays whose origin is outside the range the f32 test was validated for, or whose stack overflowed, test every
// shape exactly.  Same bits as trace_exact_kernel.
//
// Bound: VALU issue of the traversal step; algorithmic bytes per segment = box_tests * 32 +
// leaf_filter_tests * 16|32 + exact_tests * 32|200 (BVH config), all counted by the kernel.
#include "rtx_launch.h"
#include "rtx_traverse.h"

namespace rtx {

// TRIS: the tree holds triangle leaves.  SPILL: the tree is deep enough that a stack may need more than the LDS
// entries.  (Both only remove code: the spheres-only shallow-tree variant is what C2 runs.)
template <bool TRIS, bool SPILL>
__global__ __launch_bounds__(kBvhThreads, kBvhWavesPerSimd) void trace_bvh_kernel(const SceneView *__restrict__ svp,
                                                                   const RowsView *__restrict__ rvp,
                                                                   double *__restrict__ samples, Counters *__restrict__ ctr,
                                                                   unsigned long long *__restrict__ work_counter,
                                                                   const float4 *__restrict__ nodes, const LeafArrays la,
                                                                   uint32_t *__restrict__ spill, uint32_t spill_entries)
{
    // the hot arrays come in as kernel arguments (= known global address space -> global_load); a pointer read
    // from the SceneView in memory would be `flat`, whose loads also tie up lgkmcnt together with the LDS stack
    const SceneView &sv = *svp;
    const RowsView &rv = *rvp;
    __shared__ uint32_t lds_stack[kBvh4StackEntries + 1][kBvhThreads];      // + the sink row of bvh_step's pushes
    __shared__ uint32_t lds_q[kBvhQueue][kBvhThreads];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const size_t spill_stride = (size_t)gridDim.x * kBvhThreads, glane = (size_t)blockIdx.x * kBvhThreads + tid;
    const uint32_t bounce_limit = sv.max_bounces >= 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)sv.max_bounces + 1u;

    unsigned long long wave_next = 0, wave_end = 0;      // this wave's share of the ray queue (wave-uniform)
    bool queue_empty = false;
    bool alive = false;
    RayState r;
    uint32_t ridx = 0;                       // the ray's index in the launch's queue = where its sample goes
    unsigned long long segs = 0, box_tests = 0, leaf_filters = 0, exact = 0;
    unsigned long long wave_steps = 0;      // (-DRTX_BVH_STATS: traversal-loop iterations of the wave, reported via exact_tests)
#ifdef RTX_BVH_STATS
    unsigned long long cyc_trav = 0, cyc_other = 0, cyc_mark = __builtin_amdgcn_s_memtime();   // (reported via filter_tests / box_tests)
#define RTX_MARK(acc) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - cyc_mark; cyc_mark = now_; }
#else
#define RTX_MARK(acc)
#endif

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
                wave_end = base + rv.grab < rv.n_rays ? base + rv.grab : rv.n_rays;
                if (base >= rv.n_rays) { queue_empty = true; wave_next = wave_end = 0; }
            }
            if (!alive && wave_next < wave_end) {
                const unsigned long long my = wave_next + bvh_mbcnt(idle_mask);
                bool valid = my < wave_end;
                uint32_t pl = 0, smp = 0;
                if (valid) {
                    if (rv.tiles_x != 0u) valid = ray_index_to_pixel_tiled(rv, my, pl, smp);
                    else ray_index_to_pixel(rv, my, pl, smp);
                }
                if (valid) {
                    gen_primary(sv, rv, pl, rv.sample_begin + smp, r);
                    alive = true;
                    ridx = (uint32_t)my;                                          // (the host keeps rv.n_rays below 2^32)
                    if (sv.n_objects == 0) {                                      // scene.rs:224-226
                        store_sample(samples, rv, ridx, mk(0.0, 0.0, 0.0));
                        alive = false;
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

        if (alive) {
            // ---- one segment: closest_object (scene.rs:243-251)
            const RayX rx = make_rayx(r.pos, r.dir);
            Hit h;
            hit_init(h);
            ++segs;
            const float omax = fmaxf(fmaxf(__builtin_fabsf((float)r.pos.x), __builtin_fabsf((float)r.pos.y)),
                                     __builtin_fabsf((float)r.pos.z));
            bool sweep_spheres = true;                   // shapes the tree did not cover get the exact test below
            uint32_t tri_sweep_from = 0;
            const bool in32 = sv.n_bvh_nodes != 0 && omax <= sv.bvh_origin_limit;          // NaN origin -> exhaustive branch
            const bool in64 = !in32 && sv.n_bvh_nodes != 0 && omax <= sv.bvh_origin_limit * kBvhRange64;
            if (in32 || in64) {
                FilterParams fpar;
                TriFilterParams tpar;
                if (sv.bvh_flags & 1u) filter_from_ray(sv, r.pos, r.dir, fpar); else filter_idle(fpar);
                if (TRIS && (sv.bvh_flags & 2u)) tri_filter_from_ray(sv, r.pos, r.dir, tpar); else tri_filter_idle(tpar);
                bool overflow = false;
                RTX_MARK(cyc_other)
                if (in32) {
                    Ray32 q;
                    make_ray32(r.pos, rx.dirn, (double)sv.bvh_inv_max, q);
                    bvh_traverse<TRIS, SPILL>(nodes, la, q, fpar, tpar, rx, sv.bvh_root, overflow, h, &lds_stack[0][0], &lds_q[0][0], tid, spill,
                                              spill_entries, spill_stride, glane, box_tests, leaf_filters, exact, wave_steps);
                } else {                              // origin far outside the scene: the same walk with an f64 slab test
                    Ray64 q;
                    make_ray64(r.pos, rx.dirn, (double)sv.bvh_inv_max, q);
                    bvh_traverse<TRIS, SPILL>(nodes, la, q, fpar, tpar, rx, sv.bvh_root, overflow, h, &lds_stack[0][0], &lds_q[0][0], tid, spill,
                                              spill_entries, spill_stride, glane, box_tests, leaf_filters, exact, wave_steps);
                }
                RTX_MARK(cyc_trav)
                if (!overflow) {                      // (a dropped subtree: every shape gets the exact test)
                    sweep_spheres = (sv.bvh_flags & 1u) == 0u;
                    tri_sweep_from = (sv.bvh_flags & 2u) ? sv.n_tri_tree : 0u;
                }
            }
            if (sweep_spheres) {
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
                alive = false;
            }
        }
    }
    // counters: segments, exact f64 shape tests, box tests + leaf filter tests (reported through filter_tests)
    unsigned long long filt = box_tests + leaf_filters;
#ifdef RTX_BVH_STATS
    RTX_MARK(cyc_other)
    exact = wave_steps;
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

// A 4-wide node pushes at most 3 entries per level, so 3 * depth bounds the stack.
uint32_t bvh_spill_entries(const SceneView &sv)
{
    const uint32_t need = 3u * sv.bvh_depth + 2u;
    return need > (uint32_t)kBvh4StackEntries ? need - (uint32_t)kBvh4StackEntries : 0u;
}

size_t bvh_spill_bytes(const SceneView &sv, int n_cus)
{
    return (size_t)bvh_spill_entries(sv) * (size_t)n_cus * kBvhWavesPerSimd * kBvhThreads * sizeof(uint32_t);
}

hipError_t launch_trace_bvh(const SceneView *d_sv, const SceneView &sv, const RowsView *d_rv, const RowsView &rv,
                            double *samples, Counters *counters, unsigned long long *work_counter, uint32_t *spill, int n_cus,
                            hipStream_t stream)
{
    const uint64_t want = (rv.n_rays + kBvhThreads - 1) / kBvhThreads;
    const uint64_t cap = (uint64_t)n_cus * kBvhWavesPerSimd;
    const uint32_t blocks = (uint32_t)(want < cap ? want : cap);
    if (blocks == 0) return hipSuccess;
    LeafArrays la;
    la.sphere_f32 = sv.bvh_leaf_f32; la.sphere_prims = sv.bvh_prims; la.spheres = sv.spheres; la.sphere_ids = sv.sphere_id;
    la.tri_f32 = sv.tri_f32; la.tri_fidx = sv.tri_fidx; la.tris = sv.tris;
    const uint32_t spill_entries = spill ? bvh_spill_entries(sv) : 0u;
    const bool tris = (sv.bvh_flags & 2u) != 0u, deep = spill_entries != 0u;
    auto kernel = tris ? (deep ? trace_bvh_kernel<true, true> : trace_bvh_kernel<true, false>)
                       : (deep ? trace_bvh_kernel<false, true> : trace_bvh_kernel<false, false>);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kBvhThreads), 0, stream, d_sv, d_rv, samples, counters, work_counter,
                       reinterpret_cast<const float4 *>(sv.bvh_nodes), la, spill, spill_entries);
    return hipGetLastError();
}

}  // namespace rtx
