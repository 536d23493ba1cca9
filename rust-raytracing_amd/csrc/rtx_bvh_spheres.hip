// rtx_bvh_spheres.hip -- trace_bvh_spheres_kernel: RTX_KERNEL_BVH for trees that hold spheres only (C2, C4).
//
// Same rays, same tree, same exact tests and the same bits as trace_bvh_kernel<false, *> (rtx_bvh.hip); what differs is
// the traversal loop, which holds no f64 value: it prunes with the conservative f32 distance bounds of
// bvh_traverse_spheres (rtx_traverse.h) instead of the exact winner's distance, and the candidates that can still be the
// winner get their exact f64 test (sphere.rs:19-30) once, after the walk.  What that buys on C2 (one MI355X, 64 spp):
//   * 1381 -> 1471 Mrays/s at the same 4 waves per SIMD: no f64 code inside the loop (the old loop ran the exact tests
//     every 4th step for the whole wave), 1.08 instead of 1.22 exact tests and 93.4 instead of 96.0 box tests per segment
//     (the pruning bound no longer lags by up to 4 steps);
//   * the traversal loop is free of scratch traffic (tools/isa_spills.py): what the allocator spills (31 VGPRs) is
//     touched only between segments.
// Measured and dropped: more waves per SIMD.  With the f64 state out of the loop the kernel also builds for 5 / 6 / 8
// waves per SIMD (96 / 80 / 64 VGPRs, LDS stack 22 / 17 / 11 entries + the HBM column): 1420 / 1177 / 995 Mrays/s -- the
// loop is VALU-issue bound (its min/max/cmp/cndmask mix issues at ~4 cycles per instruction, SQ counters in profiles/),
// so extra waves add nothing and their spill reloads inside the loop (4 / 8 / 18 per step) cost.
#include "rtx_launch.h"
#include "rtx_traverse.h"

namespace rtx {

constexpr int kSphWavesPerSimd = 4;          // = workgroups per CU (4 waves each)
constexpr int kSphStack = 30;                // LDS stack entries per lane: (30 + 1 sink row + 2 * kSphQueue queue rows) KB per workgroup

// Two stages.  A wave's round lasts as long as its longest walk.  Primary rays (an 8x8 tile per wave: coherent, short
// walks) and bounced rays (incoherent, long walks) in one wave make the primaries wait for the bounces: measured on C2,
// the primary rays alone take 24.7 of the 89.5 ms although they are 47 % of the segments.  So the launch is split:
//   MODE 1   primary rays only: every lane takes a fresh ray each round; a ray that survives its first hit goes to a
//            queue (state structure-of-arrays by slot; a wave reserves 512 slots per atomic and marks what it leaves
//            unused as dead)
//   MODE 2   the same kernel fed from that queue: every lane carries a bounced ray, idle lanes take the next ones
// MODE 0 is the single launch (small launches, A/B runs: RTX_HIP_BVH_ONE_STAGE=1).
struct SphQueue {
    double *pos[3], *dir[3], *res[3], *lig[3];
    uint32_t *ridx;                           // kNone: a slot its wave reserved and did not use
    unsigned long long *count;                // slots reserved by stage 1 = the length stage 2 walks
    unsigned long long capacity;
};
constexpr uint32_t kSphQueueChunk = 512;

template <bool SPILL, int MODE>
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
    unsigned long long segs = 0, box_tests = 0, leaf_filters = 0, exact = 0;

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
                    if (valid) {                           // a ray in flight, as stage 1 left it after its first hit
                        ridx = sq.ridx[my];
                        valid = ridx != kNone;
                    }
                    if (valid) {
                        if (rv.tiles_x != 0u) (void)ray_index_to_pixel_tiled(rv, ridx, pl, smp);
                        else ray_index_to_pixel(rv, ridx, pl, smp);
                        const uint32_t k = pl / rv.width, x = pl - k * rv.width;
                        const uint64_t pix = (uint64_t)(rv.row_begin + k * rv.row_stride) * rv.width + x;
                        r.key = rng_key(sv.seed, pix, rv.sample_begin + smp);
                        r.bounce = 1u;
                        r.draw = 8u;
                        r.pos = mk(sq.pos[0][my], sq.pos[1][my], sq.pos[2][my]);
                        r.dir = mk(sq.dir[0][my], sq.dir[1][my], sq.dir[2][my]);
                        r.result = mk(sq.res[0][my], sq.res[1][my], sq.res[2][my]);
                        r.light = mk(sq.lig[0][my], sq.lig[1][my], sq.lig[2][my]);
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

        // ---- one segment: closest_object (scene.rs:243-251).  Phase 1, f32 only: walk the tree, collect candidates
        float best_up = __builtin_inff();
        uint32_t qcnt = 0, nbox = 0, nleaf = 0;
        bool overflow = false, walked = false;
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
                if (in32) {
                    Ray32 q;
                    make_ray32(r.pos, dirn, (double)sv.bvh_inv_max, q);
                    bvh_traverse_spheres<STACK, SPILL>(nodes, la.sphere_f32, la.sphere_prims, q, sr, sv.bvh_root, &lds_stack[0][0],
                                                       lq, tid, spill, spill_entries, spill_stride, glane, best_up, qcnt,
                                                       overflow, nbox, nleaf);
                } else {                              // origin far outside the scene: the same walk with an f64 slab test
                    Ray64 q;
                    make_ray64(r.pos, dirn, (double)sv.bvh_inv_max, q);
                    bvh_traverse_spheres<STACK, SPILL>(nodes, la.sphere_f32, la.sphere_prims, q, sr, sv.bvh_root, &lds_stack[0][0],
                                                       lq, tid, spill, spill_entries, spill_stride, glane, best_up, qcnt,
                                                       overflow, nbox, nleaf);
                }
                walked = true;
            }
        }
        // ---- phase 2, f64: exact tests of the candidates that can still be the winner, the shapes outside the tree, ray_hit
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
                done = (r.bounce >= bounce_limit) || light_is_zero(r);            // scene.rs:227-228
            }
            if (done) {
                store_sample(samples, rv, ridx, r.result);
                alive = false;
            }
        }
        if constexpr (MODE == 1) {
            // ---- the survivors of this round move to the queue of rays in flight; every lane is free again
            const bool go = alive;
            const unsigned long long m = __ballot(go);
            const uint32_t n = (uint32_t)__popcll(m);
            if (n != 0u) {
                if (out_end - out_next < (unsigned long long)n) {
                    // what is left of the wave's reservation does not take them: mark it unused, reserve the next chunk
                    if (out_next + lane < out_end) sq.ridx[out_next + lane] = kNone;       // (fewer than 64 slots are left)
                    unsigned long long base = 0;
                    if (lane == 0) base = atomicAdd(sq.count, (unsigned long long)kSphQueueChunk);
                    base = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
                           __builtin_amdgcn_readfirstlane((uint32_t)base);
                    out_next = base;
                    out_end = base + kSphQueueChunk;
                }
                if (go) {
                    const unsigned long long slot = out_next + bvh_mbcnt(m);
                    if (slot < sq.capacity) {                  // (the host sizes the queue so that this always holds)
                        sq.pos[0][slot] = r.pos.x; sq.pos[1][slot] = r.pos.y; sq.pos[2][slot] = r.pos.z;
                        sq.dir[0][slot] = r.dir.x; sq.dir[1][slot] = r.dir.y; sq.dir[2][slot] = r.dir.z;
                        sq.res[0][slot] = r.result.x; sq.res[1][slot] = r.result.y; sq.res[2][slot] = r.result.z;
                        sq.lig[0][slot] = r.light.x; sq.lig[1][slot] = r.light.y; sq.lig[2][slot] = r.light.z;
                        sq.ridx[slot] = ridx;
                    }
                }
                out_next += n;
            }
            alive = false;
        }
    }
    if constexpr (MODE == 1) {                    // the unused rest of the wave's last reservation
        for (unsigned long long s = out_next + lane; s < out_end; s += 64ull)
            if (s < sq.capacity) sq.ridx[s] = kNone;
    }
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

uint32_t bvh_spheres_spill_entries(const SceneView &sv)
{
    const uint32_t need = 3u * sv.bvh_depth + 2u;       // a 4-wide node pushes at most 3 entries per level
    return need > (uint32_t)kSphStack ? need - (uint32_t)kSphStack : 0u;
}

size_t bvh_spheres_spill_bytes(const SceneView &sv, int n_cus)
{
    return (size_t)bvh_spheres_spill_entries(sv) * (size_t)n_cus * kSphWavesPerSimd * kBvhThreads * sizeof(uint32_t);
}

// the survivors' queue of the two-stage form: 12 f64 + the ray index per slot, capacity = the launch's rays + what the waves'
// reservations can leave unused, two u64 counters
static uint64_t sph_queue_capacity(uint64_t n_rays, int n_cus)
{
    return n_rays + (uint64_t)n_cus * kSphWavesPerSimd * (kBvhThreads / 64) * (kSphQueueChunk + 64u) + kSphQueueChunk;
}

size_t bvh_spheres_queue_bytes(uint64_t n_rays, int n_cus)
{
    const uint64_t cap = sph_queue_capacity(n_rays, n_cus);
    return (size_t)(cap * (12 * sizeof(double) + sizeof(uint32_t)) + 16 * 256);
}

hipError_t launch_trace_bvh_spheres(const SceneView *d_sv, const SceneView &sv, const RowsView *d_rv, const RowsView &rv,
                                    double *samples, Counters *counters, unsigned long long *work_counter, uint32_t *spill, int n_cus,
                                    void *queue_mem, hipStream_t stream)
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
    if (!queue_mem || sv.max_bounces == 0) {     // one stage
        auto kernel = deep ? trace_bvh_spheres_kernel<true, 0> : trace_bvh_spheres_kernel<false, 0>;
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kBvhThreads), 0, stream, d_sv, d_rv, samples, counters, work_counter, nodes, la,
                           spill, spill_entries, sq);
        return hipGetLastError();
    }
    // ---- two stages: carve the queue (every array on a 256-byte boundary), zero its two counters
    const uint64_t capacity = sph_queue_capacity(rv.n_rays, n_cus);
    char *p = static_cast<char *>(queue_mem);
    auto take = [&](size_t bytes) { char *q = p; p += (bytes + 255) & ~(size_t)255; return q; };
    unsigned long long *ctrs = reinterpret_cast<unsigned long long *>(take(2 * sizeof(unsigned long long)));
    for (int k = 0; k < 3; ++k) sq.pos[k] = reinterpret_cast<double *>(take(capacity * sizeof(double)));
    for (int k = 0; k < 3; ++k) sq.dir[k] = reinterpret_cast<double *>(take(capacity * sizeof(double)));
    for (int k = 0; k < 3; ++k) sq.res[k] = reinterpret_cast<double *>(take(capacity * sizeof(double)));
    for (int k = 0; k < 3; ++k) sq.lig[k] = reinterpret_cast<double *>(take(capacity * sizeof(double)));
    sq.ridx = reinterpret_cast<uint32_t *>(take(capacity * sizeof(uint32_t)));
    sq.count = ctrs;
    sq.capacity = capacity;
    hipError_t e = hipMemsetAsync(ctrs, 0, 2 * sizeof(unsigned long long), stream);
    if (e != hipSuccess) return e;
    auto k1 = deep ? trace_bvh_spheres_kernel<true, 1> : trace_bvh_spheres_kernel<false, 1>;
    hipLaunchKernelGGL(k1, dim3(blocks), dim3(kBvhThreads), 0, stream, d_sv, d_rv, samples, counters, work_counter, nodes, la, spill,
                       spill_entries, sq);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    auto k2 = deep ? trace_bvh_spheres_kernel<true, 2> : trace_bvh_spheres_kernel<false, 2>;
    hipLaunchKernelGGL(k2, dim3(blocks), dim3(kBvhThreads), 0, stream, d_sv, d_rv, samples, counters, ctrs + 1, nodes, la, spill,
                       spill_entries, sq);
    return hipGetLastError();
}

}  // namespace rtx
