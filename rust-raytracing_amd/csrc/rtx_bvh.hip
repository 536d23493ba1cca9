// rtx_bvh.hip -- trace_bvh_kernel: closest_object (scene.rs:243-251) by flat-BVH traversal over spheres and triangles.
//
// Persistent waves; each lane owns one ray for its whole life (ray state in registers).  A wave takes rays
// from the global queue 512 at a time (one atomic per 512 rays) and hands them to idle lanes with
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

namespace {

#ifndef RTX_BVH_WPE
#define RTX_BVH_WPE 4
#endif
constexpr int kBvhWavesPerSimd = RTX_BVH_WPE;     // = workgroups per CU (4 waves each)
constexpr uint32_t kGrab = 512;            // rays a wave takes from the global queue per atomic

__device__ __forceinline__ uint32_t bvh_mbcnt(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

}  // namespace

// One traversal step of one lane: open wide node `node`, filter its leaf children into the candidate queue, push
// the interior children still in reach (farthest first) and move to the nearest (or pop).  node == kNone afterwards
// means the traversal is complete.  lds_stack has one row more than kBvh4StackEntries: the sink of the branch-free
// pushes.
template <bool TRIS, bool SPILL>
__device__ __forceinline__ void bvh_step(const float4 *__restrict__ nodes, const LeafArrays &la, const Ray32 &q,
                                         const FilterParams &fpar, const TriFilterParams &tpar, const RayX &rx, uint32_t &node,
                                         uint32_t &sp, uint32_t &qcnt, bool &overflow, Hit &h, float &best_up,
                                         uint32_t *lds_stack, uint32_t *lds_q, uint32_t tid, uint32_t *__restrict__ spill,
                                         uint32_t spill_entries, size_t spill_stride, size_t glane,
                                         uint32_t &nbox, uint32_t &nleaf, unsigned long long &exact)
{
    // one 128-byte fetch: the boxes of up to four children (rtx_bvh.h Bvh4Node)
    const float4 *np = nodes + 8 * (size_t)node;
    float4 ca[4], cb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { ca[c] = np[c]; cb[c] = np[4 + c]; }
    float tc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) tc[c] = box_entry32(ca[c], cb[c], q, best_up);
    nbox += 4;
    // leaf children that the ray enters: f32 filter now, survivors are queued; the exact f64 tests run
    // every 4th step (and at the end) for all lanes together, so their cost is not paid per (step, child,
    // shape) under divergence.  The pruning bound lags by at most 4 steps, which only costs visits.
    uint32_t leafmask = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint32_t count = __float_as_uint(cb[c].w);
        if (tc[c] < __builtin_inff() && count - 1u < 0x1FFFFu) leafmask |= 1u << c;          // neither interior (0) nor empty (~0)
    }
    while (leafmask != 0u) {                  // one copy of the leaf code, however many of the four children are leaves
        const uint32_t c = (uint32_t)__builtin_ctz(leafmask);
        leafmask &= leafmask - 1u;
        const uint32_t first = __float_as_uint(c == 0 ? ca[0].w : (c == 1 ? ca[1].w : (c == 2 ? ca[2].w : ca[3].w)));
        const uint32_t count = __float_as_uint(c == 0 ? cb[0].w : (c == 1 ? cb[1].w : (c == 2 ? cb[2].w : cb[3].w)));
        const uint32_t n = count & 0xFFFFu;
        for (uint32_t k = 0; k < n; ++k) {
            bool cand;
            uint32_t entry = (first + k) | kQueueTri;
            if (TRIS && (count & kBvhTriLeaf)) {
                const float4 A = la.tri_f32[2 * (size_t)(first + k)], B = la.tri_f32[2 * (size_t)(first + k) + 1];
                cand = (int)tri_filter_sign(A, B, tpar) >= 0;                       // q may be above the footprint
            } else {
                const float4 rec = la.sphere_f32[first + k];
                cand = (int)__float_as_uint(filter_disc1(rec, fpar)) >= 0;          // D >= 0: cannot be excluded
                if (cand) entry = la.sphere_prims[first + k];
            }
            if (cand) {
                if (qcnt == (uint32_t)kBvhQueue) flush_candidates<TRIS>(la, rx, lds_q, tid, qcnt, h, best_up, exact);
                lds_q[(size_t)qcnt * kBvhThreads + tid] = entry;
                qcnt += 1;
            }
        }
        nleaf += n;
    }
    // interior children still in reach, nearest first: keys = entry distance (inf = not to be visited)
    float key[4];
    uint32_t lnk[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        key[c] = __float_as_uint(cb[c].w) == 0u ? tc[c] : __builtin_inff();     // (tc is already inf for a box out of reach)
        lnk[c] = __float_as_uint(ca[c].w);
    }
#define RTX_CSWAP(i, j) { if (key[j] < key[i]) { float tk = key[i]; key[i] = key[j]; key[j] = tk; uint32_t tl = lnk[i]; lnk[i] = lnk[j]; lnk[j] = tl; } }
    RTX_CSWAP(0, 1) RTX_CSWAP(2, 3) RTX_CSWAP(0, 2) RTX_CSWAP(1, 3) RTX_CSWAP(1, 2)
#undef RTX_CSWAP
    // push the farther ones (farthest first), descend into the nearest.  The keys are sorted, so the children to push
    // are lnk[1..npush]; lnk[i] goes to row sp + npush - i, the others to the sink row -- three unconditional stores.
    // (stack + queue = 39 words of LDS per lane, which is what 16 waves per CU leave; entries beyond the 30 in LDS go
    // to the lane's column of the HBM spill area, which the launcher sizes from the tree's depth so that it cannot run
    // out -- the exhaustive sweep after an overflow is only a guard)
    const uint32_t npush = (key[1] < __builtin_inff() ? 1u : 0u) + (key[2] < __builtin_inff() ? 1u : 0u) +
                           (key[3] < __builtin_inff() ? 1u : 0u);
    if (sp + 3u <= (uint32_t)kBvh4StackEntries) {
#pragma unroll
        for (uint32_t i = 1; i <= 3; ++i) {
            const uint32_t row = i <= npush ? sp + npush - i : (uint32_t)kBvh4StackEntries;
            lds_stack[(size_t)row * kBvhThreads + tid] = lnk[i];
        }
        sp += npush;
    } else {
#define RTX_PUSH(v)                                                                                      \
        {                                                                                                \
            if (sp < (uint32_t)kBvh4StackEntries) { lds_stack[(size_t)sp * kBvhThreads + tid] = (v); sp += 1; } \
            else if (SPILL && sp - (uint32_t)kBvh4StackEntries < spill_entries) {                        \
                spill[(size_t)(sp - (uint32_t)kBvh4StackEntries) * spill_stride + glane] = (v); sp += 1; \
            } else overflow = true;                                                                      \
        }
        if (key[3] < __builtin_inff()) RTX_PUSH(lnk[3])
        if (key[2] < __builtin_inff()) RTX_PUSH(lnk[2])
        if (key[1] < __builtin_inff()) RTX_PUSH(lnk[1])
#undef RTX_PUSH
    }
    node = key[0] < __builtin_inff() ? lnk[0] : kNone;
    if (node == kNone && sp != 0u) {
        sp -= 1;                        // its boxes are re-tested against the current bound when it is opened
        node = (!SPILL || sp < (uint32_t)kBvh4StackEntries) ? lds_stack[(size_t)sp * kBvhThreads + tid]
                                                : spill[(size_t)(sp - (uint32_t)kBvh4StackEntries) * spill_stride + glane];
    }
}

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
    uint32_t pl = 0, smp = 0;
    unsigned long long segs = 0, box_tests = 0, leaf_filters = 0, exact = 0;
#ifdef RTX_BVH_STATS
    unsigned long long wave_steps = 0;      // diagnostic: traversal-loop iterations of the wave (reported via exact_tests)
    unsigned long long cyc_trav = 0, cyc_other = 0, cyc_mark = __builtin_amdgcn_s_memtime();   // (reported via filter_tests / box_tests)
#define RTX_MARK(acc) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - cyc_mark; cyc_mark = now_; }
#else
#define RTX_MARK(acc)
#endif

    for (;;) {
        // ---- hand rays to idle lanes: ballot + prefix sum over the wave's local range, one atomic per 512 rays
        const unsigned long long idle_mask = __ballot(!alive);
        if (idle_mask != 0ull) {
            if (wave_next >= wave_end && !queue_empty) {
                unsigned long long base = 0;
                if (lane == 0) base = atomicAdd(work_counter, (unsigned long long)kGrab);
                base = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
                       __builtin_amdgcn_readfirstlane((uint32_t)base);
                wave_next = base;
                wave_end = base + kGrab < rv.n_rays ? base + kGrab : rv.n_rays;
                if (base >= rv.n_rays) { queue_empty = true; wave_next = wave_end = 0; }
            }
            if (!alive && wave_next < wave_end) {
                const unsigned long long my = wave_next + bvh_mbcnt(idle_mask);
                if (my < wave_end) {
                    ray_index_to_pixel(rv, my, pl, smp);
                    gen_primary(sv, rv, pl, rv.sample_begin + smp, r);
                    alive = true;
                    if (sv.n_objects == 0) {                                      // scene.rs:224-226
                        double *o = samples + ((uint64_t)smp * rv.npix + pl) * 3;
                        o[0] = 0.0; o[1] = 0.0; o[2] = 0.0;
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
            if (sv.n_bvh_nodes != 0 && omax <= sv.bvh_origin_limit) {          // NaN origin -> exhaustive branch
                Ray32 q;
                make_ray32(r.pos, rx.dirn, (double)sv.bvh_inv_max, q);
                FilterParams fpar;
                TriFilterParams tpar;
                if (sv.bvh_flags & 1u) filter_from_ray(sv, r.pos, r.dir, fpar); else filter_idle(fpar);
                if (TRIS && (sv.bvh_flags & 2u)) tri_filter_from_ray(sv, r.pos, r.dir, tpar); else tri_filter_idle(tpar);
                float best_up = __builtin_inff();
                uint32_t sp = 0, qcnt = 0, step = 0, nbox = 0, nleaf = 0;
                bool overflow = false;
                uint32_t node = 0;                       // wide node 0 is the root
                RTX_MARK(cyc_other)
                while (node != kNone) {
#ifdef RTX_BVH_STATS
                    { const unsigned long long am = __ballot(true); if (lane == (uint32_t)(__ffsll((long long)am) - 1)) wave_steps += 1; }
#endif
                    bvh_step<TRIS, SPILL>(nodes, la, q, fpar, tpar, rx, node, sp, qcnt, overflow, h, best_up, &lds_stack[0][0], &lds_q[0][0],
                                          tid, spill, spill_entries, spill_stride, glane, nbox, nleaf, exact);
                    step += 1;
                    if ((step & 3u) == 0u) flush_candidates<TRIS>(la, rx, &lds_q[0][0], tid, qcnt, h, best_up, exact);
                }
                RTX_MARK(cyc_trav)
                box_tests += nbox;
                leaf_filters += nleaf;
                flush_candidates<TRIS>(la, rx, &lds_q[0][0], tid, qcnt, h, best_up, exact);
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
                double *o = samples + ((uint64_t)smp * rv.npix + pl) * 3;
                o[0] = r.result.x; o[1] = r.result.y; o[2] = r.result.z;
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
