// rtx_bvh_spheres_pool.hip -- trace_bvh_spheres_pool_kernel: RTX_KERNEL_BVH_REGROUP for trees that hold spheres only.
//
// trace_bvh_spheres_kernel (one ray per lane, a round = one segment of every lane) issues VALU instructions 78 % of the
// time but only 46 % of the lanes of an instruction work: the wave iterates its walk until the longest of its 64
// traversals ends.  Refilling lanes inside the walk needs a supply of segments that are ready to walk.  Here every lane
// owns K rays (its pool, in lane-private device memory, structure-of-arrays across the resident lanes so that every
// access is coalesced) and a round is
//     f64 phase   for each of the lane's K rays, all 64 lanes together: the exact tests of the candidates its last walk
//                 left, the shapes outside the tree, ray_hit (or a new primary ray when the path ended), then the f32
//                 parameters of the next segment
//     walk        sphere_step (rtx_traverse.h) over the lane's pending segments one after the other; a lane whose walk
//                 ended waits until kPoolService lanes do, then they store their candidates and start their next
//                 pending segment together -- the wave leaves the loop when every lane has walked all its segments
// so the lanes of a wave stay busy until the pools run dry at the end of the round, and the f64 phase runs with all
// lanes as before.  Same functions, same operation order, same bits as every other kernel.
#include "rtx_launch.h"
#include "rtx_traverse.h"

namespace rtx {

#ifndef RTX_POOL_K
#define RTX_POOL_K 4
#endif
#ifndef RTX_POOL_SERVICE
#define RTX_POOL_SERVICE 16
#endif
constexpr int kPoolK = RTX_POOL_K;                    // rays per lane
constexpr uint32_t kPoolService = RTX_POOL_SERVICE;   // lanes of a wave that wait before they are served together
constexpr int kPoolWaves = 4;                         // workgroups per CU
constexpr int kPoolStack = 30;                        // LDS stack entries per lane (+ sink row + 2 * kSphQueue queue rows)

// the pool: per ray kPoolD 8-byte fields and kPoolU 4-byte fields, field-major over the resident lanes
constexpr int kPoolD = 13;                            // pos, dir, result, light, RNG key
constexpr int kPoolU = 22;                            // ridx, bounce | Ray32 (6), SphereRay (9) | candidate count, 4 candidates
enum : int { PU_RIDX = 0, PU_BOUNCE = 1, PU_PAR = 2, PU_CNT = 17, PU_CAND = 18 };
constexpr uint32_t kPoolFallback = 0x80000000u;       // candidate count: test every sphere (no walk possible / queue overflow)

struct Pool {
    double *d;
    uint32_t *u;
    size_t lanes;                                     // resident lanes of the launch
};

__device__ __forceinline__ double &pool_d(const Pool &p, int k, int f, size_t glane) { return p.d[((size_t)(k * kPoolD + f)) * p.lanes + glane]; }
__device__ __forceinline__ uint32_t &pool_u(const Pool &p, int k, int f, size_t glane) { return p.u[((size_t)(k * kPoolU + f)) * p.lanes + glane]; }

template <bool SPILL>
__global__ __launch_bounds__(kBvhThreads, kPoolWaves) void trace_bvh_spheres_pool_kernel(const SceneView *__restrict__ svp,
                                                                                         const RowsView *__restrict__ rvp,
                                                                                         double *__restrict__ samples,
                                                                                         Counters *__restrict__ ctr,
                                                                                         unsigned long long *__restrict__ work_counter,
                                                                                         const float4 *__restrict__ nodes,
                                                                                         const LeafArrays la, const Pool pool,
                                                                                         uint32_t *__restrict__ spill,
                                                                                         uint32_t spill_entries)
{
    constexpr int STACK = kPoolStack;
    const SceneView &sv = *svp;
    const RowsView &rv = *rvp;
    __shared__ uint32_t lds_stack[STACK + 1][kBvhThreads];
    __shared__ uint32_t lds_q[2 * kSphQueue][kBvhThreads];
    uint32_t *const ls = &lds_stack[0][0];
    uint32_t *const lq = &lds_q[0][0];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const size_t spill_stride = (size_t)gridDim.x * kBvhThreads, glane = (size_t)blockIdx.x * kBvhThreads + tid;
    const uint32_t bounce_limit = sv.max_bounces >= 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)sv.max_bounces + 1u;

    unsigned long long wave_next = 0, wave_end = 0;      // this wave's share of the ray queue (wave-uniform)
    bool queue_empty = false;
    unsigned long long segs = 0, box_tests = 0, leaf_filters = 0, exact = 0;
#pragma unroll 1
    for (int k = 0; k < kPoolK; ++k) pool_u(pool, k, PU_RIDX, glane) = kNone;

    for (;;) {
        // ---- f64 phase: every ray of the pool, all lanes together ------------------------------------------------------
        uint32_t pending = 0;                            // bit k: ray k has a segment to walk in this round
        bool any_alive = false;
#pragma unroll 1
        for (int k = 0; k < kPoolK; ++k) {
            RayState r;
            uint32_t ridx = pool_u(pool, k, PU_RIDX, glane);
            bool alive = ridx != kNone;
            if (alive) {
                // closest_object's exact part for the segment walked in the last round (scene.rs:243-251)
                r.pos = mk(pool_d(pool, k, 0, glane), pool_d(pool, k, 1, glane), pool_d(pool, k, 2, glane));
                r.dir = mk(pool_d(pool, k, 3, glane), pool_d(pool, k, 4, glane), pool_d(pool, k, 5, glane));
                const RayX rx = make_rayx(r.pos, r.dir);
                Hit h;
                hit_init(h);
                ++segs;
                const uint32_t cnt = pool_u(pool, k, PU_CNT, glane);
                if (cnt & kPoolFallback) {
                    for (uint32_t j = 0; j < sv.n_spheres; ++j) {
                        double t;
                        if (sphere_distance(la.spheres[j], rx, &t)) hit_consider(h, t, la.sphere_ids[j], 0, j);
                    }
                    exact += sv.n_spheres;
                } else {
#pragma unroll 1
                    for (uint32_t e = 0; e < cnt; ++e) {
                        const uint32_t idx = pool_u(pool, k, PU_CAND + (int)e, glane);
                        double t;
                        if (sphere_distance(la.spheres[idx], rx, &t)) hit_consider(h, t, la.sphere_ids[idx], 0, idx);
                    }
                    exact += cnt;
                }
                for (uint32_t j = 0; j < sv.n_planes; ++j) {
                    double t;
                    if (plane_distance(sv.planes[j], rx, &t)) hit_consider(h, t, sv.planes[j].id, 1, j);
                }
                // the few triangles of a sphere scene (none of them in the tree), by filter record
                for (uint32_t j = 0; j < sv.n_tri_filter; ++j) {
                    const uint32_t tk = la.tri_fidx[j];
                    double t;
                    if (triangle_distance(la.tris[tk], rx, &t)) hit_consider(h, t, la.tris[tk].id, 2, tk);
                }
                exact += sv.n_planes + sv.n_tri_filter;

                // ---- render_ray's match arm + ray_hit (scene.rs:232-239, 260-278)
                r.result = mk(pool_d(pool, k, 6, glane), pool_d(pool, k, 7, glane), pool_d(pool, k, 8, glane));
                bool done = true;
                if (h.id != kNone) {
                    r.light = mk(pool_d(pool, k, 9, glane), pool_d(pool, k, 10, glane), pool_d(pool, k, 11, glane));
                    r.key = (uint64_t)__double_as_longlong(pool_d(pool, k, 12, glane));
                    r.bounce = pool_u(pool, k, PU_BOUNCE, glane);
                    r.draw = 6u + 2u * r.bounce;
                    advance_and_shade(sv, h, r);
                    done = (r.bounce >= bounce_limit) || light_is_zero(r);            // scene.rs:227-228
                }
                if (done) {
                    store_sample(samples, rv, ridx, r.result);
                    alive = false;
                }
            }
            // ---- a dead ray's slot takes the next ray of the wave's range of the queue (one atomic per rv.grab rays)
            const bool fresh_wanted = !alive;
            const unsigned long long idle_mask = __ballot(fresh_wanted);
            bool fresh = false;
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
                if (fresh_wanted && wave_next < wave_end) {
                    const unsigned long long my = wave_next + bvh_mbcnt(idle_mask);
                    bool valid = my < wave_end;
                    uint32_t pl = 0, smp = 0;
                    if (valid) {
                        if (rv.tiles_x != 0u) valid = ray_index_to_pixel_tiled(rv, my, pl, smp);
                        else ray_index_to_pixel(rv, my, pl, smp);
                    }
                    if (valid) {
                        gen_primary(sv, rv, pl, rv.sample_begin + smp, r);
                        ridx = (uint32_t)my;                                          // (the host keeps rv.n_rays below 2^32)
                        alive = true;
                        fresh = true;
                    }
                }
                const unsigned long long taken = (unsigned long long)__popcll(idle_mask);
                wave_next = wave_next + taken < wave_end ? wave_next + taken : wave_end;
            }
            // ---- the next segment of a live ray: its state goes back to the pool, its f32 parameters beside it
            if (alive) {
                pool_d(pool, k, 0, glane) = r.pos.x; pool_d(pool, k, 1, glane) = r.pos.y; pool_d(pool, k, 2, glane) = r.pos.z;
                pool_d(pool, k, 3, glane) = r.dir.x; pool_d(pool, k, 4, glane) = r.dir.y; pool_d(pool, k, 5, glane) = r.dir.z;
                pool_d(pool, k, 6, glane) = r.result.x; pool_d(pool, k, 7, glane) = r.result.y; pool_d(pool, k, 8, glane) = r.result.z;
                pool_d(pool, k, 9, glane) = r.light.x; pool_d(pool, k, 10, glane) = r.light.y; pool_d(pool, k, 11, glane) = r.light.z;
                if (fresh) pool_d(pool, k, 12, glane) = __longlong_as_double((long long)r.key);
                pool_u(pool, k, PU_BOUNCE, glane) = r.bounce;
                const float omax = fmaxf(fmaxf(__builtin_fabsf((float)r.pos.x), __builtin_fabsf((float)r.pos.y)),
                                         __builtin_fabsf((float)r.pos.z));
                const bool in32 = omax <= sv.bvh_origin_limit;                                  // NaN origin -> exhaustive
                const bool in64 = !in32 && omax <= sv.bvh_origin_limit * kBvhRange64;
                if (in32 || in64) {
                    SphereRay sr;
                    sphere_ray_from(sv, r.pos, r.dir, sr);
                    const V3 dirn = vnorm(r.dir);
                    if (in32) {
                        Ray32 q;
                        make_ray32(r.pos, dirn, (double)sv.bvh_inv_max, q);
                        pool_u(pool, k, PU_PAR + 0, glane) = __float_as_uint(q.ix); pool_u(pool, k, PU_PAR + 1, glane) = __float_as_uint(q.iy);
                        pool_u(pool, k, PU_PAR + 2, glane) = __float_as_uint(q.iz); pool_u(pool, k, PU_PAR + 3, glane) = __float_as_uint(q.nx);
                        pool_u(pool, k, PU_PAR + 4, glane) = __float_as_uint(q.ny); pool_u(pool, k, PU_PAR + 5, glane) = __float_as_uint(q.nz);
                        pool_u(pool, k, PU_PAR + 6, glane) = __float_as_uint(sr.px); pool_u(pool, k, PU_PAR + 7, glane) = __float_as_uint(sr.py);
                        pool_u(pool, k, PU_PAR + 8, glane) = __float_as_uint(sr.pz); pool_u(pool, k, PU_PAR + 9, glane) = __float_as_uint(sr.dx);
                        pool_u(pool, k, PU_PAR + 10, glane) = __float_as_uint(sr.dy); pool_u(pool, k, PU_PAR + 11, glane) = __float_as_uint(sr.dz);
                        pool_u(pool, k, PU_PAR + 12, glane) = __float_as_uint(sr.Kg); pool_u(pool, k, PU_PAR + 13, glane) = __float_as_uint(sr.c0);
                        pool_u(pool, k, PU_PAR + 14, glane) = __float_as_uint(sr.K);
                        pending |= 1u << k;
                    } else {
                        // an origin far outside the scene: the same walk with an f64 slab test, here and now (rare: a camera
                        // outside 4 x the scene's extent)
                        Ray64 q;
                        make_ray64(r.pos, dirn, (double)sv.bvh_inv_max, q);
                        float best_up = __builtin_inff();
                        uint32_t qcnt = 0, nbox = 0, nleaf = 0;
                        bool overflow = false;
                        bvh_traverse_spheres<STACK, SPILL>(nodes, la.sphere_f32, la.sphere_prims, q, sr, sv.bvh_root, ls, lq, tid, spill,
                                                           spill_entries, spill_stride, glane, best_up, qcnt, overflow, nbox, nleaf);
                        box_tests += nbox; leaf_filters += nleaf;
                        uint32_t n = 0;
                        for (uint32_t e = 0; e < qcnt; ++e) {
                            if (__uint_as_float(lq[(size_t)(kSphQueue + e) * kBvhThreads + tid]) <= best_up) {
                                pool_u(pool, k, PU_CAND + (int)n, glane) = lq[(size_t)e * kBvhThreads + tid];
                                n += 1;
                            }
                        }
                        pool_u(pool, k, PU_CNT, glane) = overflow ? kPoolFallback : n;
                    }
                } else {
                    pool_u(pool, k, PU_CNT, glane) = kPoolFallback;
                }
                any_alive = true;
            }
            pool_u(pool, k, PU_RIDX, glane) = alive ? ridx : kNone;
        }
        if (__ballot(any_alive) == 0ull) break;            // (a live ray exists only while the queue had rays: nothing left to take)

        // ---- the walk: the lane's pending segments one after the other, f32 only ----------------------------------------
        bool walking = false, have = false;
        int cur = 0;
        Ray32 q;
        SphereRay sr;
        float best_up = 0.f;
        uint32_t node = kNone, sp = 0, qcnt = 0, nbox = 0, nleaf = 0;
        bool overflow = false;
        q.ix = q.iy = q.iz = q.nx = q.ny = q.nz = 0.f;
        sr.px = sr.py = sr.pz = sr.dx = sr.dy = sr.dz = sr.Kg = sr.K = 0.f; sr.c0 = __builtin_inff();
        for (;;) {
            const unsigned long long wmask = __ballot(walking);
            const uint32_t n_walk = (uint32_t)__popcll(wmask);
            const uint32_t n_serv = (uint32_t)__popcll(__ballot(!walking && (have || pending != 0u)));
            // (towards the end of the round few lanes have segments left: do not let them wait for a count they cannot reach)
            const uint32_t want = (n_walk + n_serv) / 4u + 1u;
            const bool serve = wmask == 0ull || n_serv >= (want < kPoolService ? want : kPoolService);
            if (serve) {
                if (!walking && have) {                    // the candidates that can still be the winner
                    uint32_t n = 0;
#pragma unroll
                    for (int e = 0; e < kSphQueue; ++e) {
                        if ((uint32_t)e < qcnt && __uint_as_float(lq[(size_t)(kSphQueue + e) * kBvhThreads + tid]) <= best_up) {
                            pool_u(pool, cur, PU_CAND + (int)n, glane) = lq[(size_t)e * kBvhThreads + tid];
                            n += 1;
                        }
                    }
                    pool_u(pool, cur, PU_CNT, glane) = overflow ? kPoolFallback : n;
                    box_tests += nbox; leaf_filters += nleaf;
                    nbox = 0; nleaf = 0;
                    have = false;
                }
                if (!walking && pending != 0u) {
                    cur = __builtin_ctz(pending);
                    pending &= pending - 1u;
                    q.ix = __uint_as_float(pool_u(pool, cur, PU_PAR + 0, glane)); q.iy = __uint_as_float(pool_u(pool, cur, PU_PAR + 1, glane));
                    q.iz = __uint_as_float(pool_u(pool, cur, PU_PAR + 2, glane)); q.nx = __uint_as_float(pool_u(pool, cur, PU_PAR + 3, glane));
                    q.ny = __uint_as_float(pool_u(pool, cur, PU_PAR + 4, glane)); q.nz = __uint_as_float(pool_u(pool, cur, PU_PAR + 5, glane));
                    sr.px = __uint_as_float(pool_u(pool, cur, PU_PAR + 6, glane)); sr.py = __uint_as_float(pool_u(pool, cur, PU_PAR + 7, glane));
                    sr.pz = __uint_as_float(pool_u(pool, cur, PU_PAR + 8, glane)); sr.dx = __uint_as_float(pool_u(pool, cur, PU_PAR + 9, glane));
                    sr.dy = __uint_as_float(pool_u(pool, cur, PU_PAR + 10, glane)); sr.dz = __uint_as_float(pool_u(pool, cur, PU_PAR + 11, glane));
                    sr.Kg = __uint_as_float(pool_u(pool, cur, PU_PAR + 12, glane)); sr.c0 = __uint_as_float(pool_u(pool, cur, PU_PAR + 13, glane));
                    sr.K = __uint_as_float(pool_u(pool, cur, PU_PAR + 14, glane));
                    best_up = __builtin_inff();
                    qcnt = 0; sp = 0; overflow = false;
                    node = sv.bvh_root;
                    walking = true; have = true;
                }
                if (__ballot(walking) == 0ull) break;
            }
            if (walking) {
                sphere_step<STACK, SPILL>(nodes, la.sphere_f32, la.sphere_prims, q, sr, node, sp, ls, lq, tid, spill, spill_entries,
                                          spill_stride, glane, best_up, qcnt, overflow, nbox, nleaf);
                walking = node != kNone;
            }
        }
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

uint32_t bvh_spheres_pool_spill_entries(const SceneView &sv)
{
    const uint32_t need = 3u * sv.bvh_depth + 2u;       // a 4-wide node pushes at most 3 entries per level
    return need > (uint32_t)kPoolStack ? need - (uint32_t)kPoolStack : 0u;
}

static size_t pool_lanes(int n_cus) { return (size_t)n_cus * kPoolWaves * kBvhThreads; }
static size_t pool_d_bytes(int n_cus) { return ((pool_lanes(n_cus) * kPoolK * kPoolD * sizeof(double)) + 255) & ~(size_t)255; }
static size_t pool_u_bytes(int n_cus) { return ((pool_lanes(n_cus) * kPoolK * kPoolU * sizeof(uint32_t)) + 255) & ~(size_t)255; }

// device scratch of a launch: the pools, then the HBM stack columns
size_t bvh_spheres_pool_bytes(const SceneView &sv, int n_cus)
{
    return pool_d_bytes(n_cus) + pool_u_bytes(n_cus) + (size_t)bvh_spheres_pool_spill_entries(sv) * pool_lanes(n_cus) * sizeof(uint32_t);
}

hipError_t launch_trace_bvh_spheres_pool(const SceneView *d_sv, const SceneView &sv, const RowsView *d_rv, const RowsView &rv,
                                         double *samples, Counters *counters, unsigned long long *work_counter, void *scratch,
                                         int n_cus, hipStream_t stream)
{
    const uint64_t want = (rv.n_rays + (uint64_t)kBvhThreads * kPoolK - 1) / ((uint64_t)kBvhThreads * kPoolK);
    const uint64_t cap = (uint64_t)n_cus * kPoolWaves;
    const uint32_t blocks = (uint32_t)(want < cap ? want : cap);
    if (blocks == 0) return hipSuccess;
    LeafArrays la;
    la.sphere_f32 = sv.bvh_leaf_cr; la.sphere_prims = sv.bvh_prims; la.spheres = sv.spheres; la.sphere_ids = sv.sphere_id;
    la.tri_f32 = sv.tri_f32; la.tri_fidx = sv.tri_fidx; la.tris = sv.tris;
    Pool pool;
    char *p = static_cast<char *>(scratch);
    pool.d = reinterpret_cast<double *>(p);
    pool.u = reinterpret_cast<uint32_t *>(p + pool_d_bytes(n_cus));
    pool.lanes = (size_t)blocks * kBvhThreads;
    uint32_t *spill = reinterpret_cast<uint32_t *>(p + pool_d_bytes(n_cus) + pool_u_bytes(n_cus));
    const uint32_t spill_entries = bvh_spheres_pool_spill_entries(sv);
    auto kernel = spill_entries != 0u ? trace_bvh_spheres_pool_kernel<true> : trace_bvh_spheres_pool_kernel<false>;
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kBvhThreads), 0, stream, d_sv, d_rv, samples, counters, work_counter,
                       reinterpret_cast<const float4 *>(sv.bvh_nodes), la, pool, spill, spill_entries);
    return hipGetLastError();
}

}  // namespace rtx
