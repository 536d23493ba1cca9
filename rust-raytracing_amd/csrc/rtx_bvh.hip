// rtx_bvh.hip -- trace_bvh_kernel: closest_object (scene.rs:243-251) over the spheres by flat-BVH traversal.
//
// Persistent waves; each lane owns one ray for its whole life (ray state in registers).  A wave takes rays
// from the global queue 512 at a time (one atomic per 512 rays) and hands them to idle lanes with
// ballot + mbcnt, so lanes whose path ended are refilled at once.  Per segment each lane walks the BVH with its
// own LDS stack (depth-major layout: no bank conflicts), nearest child first, pruning boxes whose entry distance
// exceeds the best hit so far.  The slab test runs in f64 on the f32 boxes (rounded outward + inflated at build
// time), the leaves run the exact f64 sphere test of sphere.rs:19-30, planes and triangles are tested exactly and
// exhaustively (they cannot be culled: plane.rs is unbounded, triangle.rs has phantom hits), and the winner is the
// lexicographic minimum of (t, scene index) -- the reference's first-minimal rule.  Same bits as trace_exact_kernel.
//
// Bound: latency/L2-bandwidth of the node fetches (32 B per node, ~2 nodes per visit); algorithmic bytes per
// segment = nodes_visited * 32 + leaf_tests * 32 (BVH config), both counted by the kernel.
#include "rtx_launch.h"

namespace rtx {

namespace {

constexpr int kBvhThreads = 256;
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kGrab = 512;            // rays a wave takes from the global queue per atomic

__device__ __forceinline__ uint32_t bvh_mbcnt(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// f64 slab test of the ray against a node box; returns the entry distance or +inf on a miss.
// fmin/fmax return the non-NaN operand (0 * inf when an origin lies on a slab of an axis-parallel ray), which
// only widens the interval: conservative.
__device__ __forceinline__ double box_entry(const float4 lo, const float4 hi, const V3 o, const V3 inv, double best)
{
    const double x0 = ((double)lo.x - o.x) * inv.x, x1 = ((double)hi.x - o.x) * inv.x;
    const double y0 = ((double)lo.y - o.y) * inv.y, y1 = ((double)hi.y - o.y) * inv.y;
    const double z0 = ((double)lo.z - o.z) * inv.z, z1 = ((double)hi.z - o.z) * inv.z;
    const double tn = fmax(fmax(fmin(x0, x1), fmin(y0, y1)), fmin(z0, z1));
    const double tf = fmin(fmin(fmax(x0, x1), fmax(y0, y1)), fmax(z0, z1));
    // the boxes carry a 2^-20 relative inflation, the f64 slab arithmetic errs by ~1e-15 relative: no further slack
    // is needed, but ties (tn == best) must be kept for the first-wins rule
    const bool hit = (tn <= tf) && (tf >= 0.0) && (tn <= best);
    return hit ? tn : (double)INFINITY;
}

}  // namespace

__global__ __launch_bounds__(kBvhThreads, 4) void trace_bvh_kernel(const SceneView *__restrict__ svp,
                                                                   const RowsView *__restrict__ rvp,
                                                                   double *__restrict__ samples, Counters *__restrict__ ctr,
                                                                   unsigned long long *__restrict__ work_counter)
{
    const SceneView &sv = *svp;
    const RowsView &rv = *rvp;
    __shared__ uint32_t lds_stack[kBvhMaxDepth + 2][kBvhThreads];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const float4 *nodes = reinterpret_cast<const float4 *>(sv.bvh_nodes);
    const uint32_t bounce_limit = sv.max_bounces >= 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)sv.max_bounces + 1u;

    unsigned long long wave_next = 0, wave_end = 0;      // this wave's share of the ray queue (wave-uniform)
    bool queue_empty = false;
    bool alive = false;
    RayState r;
    uint32_t pl = 0, smp = 0;
    unsigned long long segs = 0, node_visits = 0, leaf_tests = 0;

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
            const V3 inv = mk(1.0 / rx.dirn.x, 1.0 / rx.dirn.y, 1.0 / rx.dirn.z);
            Hit h;
            hit_init(h);
            double best = (double)INFINITY;
            ++segs;
            if (sv.n_bvh_nodes != 0) {
                uint32_t sp = 0;
                uint32_t node = 0;
                {
                    const float4 lo = nodes[0], hi = nodes[1];
                    if (!(box_entry(lo, hi, r.pos, inv, best) < (double)INFINITY)) node = kNone;
                    ++node_visits;
                }
                while (node != kNone) {
                    const float4 lo = nodes[2 * (size_t)node], hi = nodes[2 * (size_t)node + 1];
                    const uint32_t link = __float_as_uint(lo.w), count = __float_as_uint(hi.w);
                    if (count != 0u) {
                        for (uint32_t k = 0; k < count; ++k) {
                            const uint32_t idx = sv.bvh_prims[link + k];
                            double t;
                            if (sphere_distance(sv.spheres[idx], rx, &t)) {
                                hit_consider(h, t, sv.sphere_id[idx], 0, idx);
                                if (h.id != kNone) best = h.t;
                            }
                        }
                        leaf_tests += count;
                        node = kNone;
                    } else {
                        const uint32_t left = node + 1u, right = link;
                        const float4 llo = nodes[2 * (size_t)left], lhi = nodes[2 * (size_t)left + 1];
                        const float4 rlo = nodes[2 * (size_t)right], rhi = nodes[2 * (size_t)right + 1];
                        const double tl = box_entry(llo, lhi, r.pos, inv, best);
                        const double tr = box_entry(rlo, rhi, r.pos, inv, best);
                        node_visits += 2;
                        const bool hl = tl < (double)INFINITY, hr = tr < (double)INFINITY;
                        if (hl && hr) {
                            const bool left_first = tl <= tr;
                            lds_stack[sp][tid] = left_first ? right : left;
                            sp += 1;
                            node = left_first ? left : right;
                        } else {
                            node = hl ? left : (hr ? right : kNone);
                        }
                    }
                    // pop until a node whose box can still hold a closer (or tied) hit
                    while (node == kNone && sp != 0u) {
                        sp -= 1;
                        const uint32_t cand = lds_stack[sp][tid];
                        const float4 clo = nodes[2 * (size_t)cand], chi = nodes[2 * (size_t)cand + 1];
                        ++node_visits;
                        if (box_entry(clo, chi, r.pos, inv, best) < (double)INFINITY) node = cand;
                    }
                }
            } else {
                for (uint32_t k = 0; k < sv.n_spheres; ++k) {
                    double t;
                    if (sphere_distance(sv.spheres[k], rx, &t)) hit_consider(h, t, sv.sphere_id[k], 0, k);
                }
                leaf_tests += sv.n_spheres;
            }
            for (uint32_t k = 0; k < sv.n_planes; ++k) {
                double t;
                if (plane_distance(sv.planes[k], rx, &t)) hit_consider(h, t, sv.planes[k].id, 1, k);
            }
            for (uint32_t k = 0; k < sv.n_tris; ++k) {
                double t;
                if (triangle_distance(sv.tris[k], rx, &t)) hit_consider(h, t, sv.tris[k].id, 2, k);
            }

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
    // counters: segments, exact f64 shape tests, BVH node visits (reported through filter_tests)
    unsigned long long exact = leaf_tests + segs * (sv.n_planes + sv.n_tris);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        segs += __shfl_xor(segs, off, 64);
        exact += __shfl_xor(exact, off, 64);
        node_visits += __shfl_xor(node_visits, off, 64);
    }
    if (lane == 0) {
        const uint32_t shard = (blockIdx.x * (kBvhThreads >> 6) + (tid >> 6)) & (kCounterShards - 1);
        if (segs) atomicAdd(&ctr[shard].segments, segs);
        if (exact) atomicAdd(&ctr[shard].exact_tests, exact);
        if (node_visits) atomicAdd(&ctr[shard].filter_tests, node_visits);
    }
}

hipError_t launch_trace_bvh(const SceneView *d_sv, const SceneView &sv, const RowsView *d_rv, const RowsView &rv,
                            double *samples, Counters *counters, unsigned long long *work_counter, int n_cus,
                            hipStream_t stream)
{
    (void)sv;
    const uint64_t want = (rv.n_rays + kBvhThreads - 1) / kBvhThreads;
    const uint64_t cap = (uint64_t)n_cus * 4;             // 16 waves per CU (4 per SIMD)
    const uint32_t blocks = (uint32_t)(want < cap ? want : cap);
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(trace_bvh_kernel, dim3(blocks), dim3(kBvhThreads), 0, stream, d_sv, d_rv, samples, counters,
                       work_counter);
    return hipGetLastError();
}

}  // namespace rtx
