// rtx_pool.hip -- trace_pool_kernel: the flat-BVH traversal of rtx_bvh.hip with a wave-local ray pool.
//
// trace_bvh_kernel gives every lane one ray and iterates segment by segment, so a wave waits for its longest
// traversal (VALU lane utilisation 35 % on C2, profiles/r01_pmc_sq_bvh_vs_sweep.txt).  Here a wave owns a POOL of
// 64 * P rays whose state lives in HBM structure-of-arrays, and a round has three phases:
//   1. refill   every lane generates primary rays for its dead pool entries (P per lane, coalesced SoA writes);
//               rays come from the global queue 512 at a time, handed out with ballot + mbcnt.
//   2. traverse lanes PULL pool entries one after the other: a lane whose traversal ended takes the next entry
//               (wave-uniform cursor + ballot/mbcnt prefix sum, no atomics needed inside a wave), so lanes stay
//               busy until the pool is empty.  The finish/start block (exact f64 tests of queued candidates,
//               exhaustive planes/triangles, store the hit, load the next ray) is heavy, so it only runs when at
//               least kPoolThreshold lanes are waiting or nobody is traversing.
//   3. shade    every lane shades its own P entries (ray_hit, scene.rs:260-278) and writes finished samples.
// Traversal, filters, exact tests, winner rule: exactly those of trace_bvh_kernel (rtx_traverse.h) -> same bits.
#include "rtx_launch.h"
#include "rtx_traverse.h"

namespace rtx {

namespace {

#ifndef RTX_POOL_P
#define RTX_POOL_P 4
#endif
#ifndef RTX_POOL_T
#define RTX_POOL_T 16
#endif
constexpr int kPoolP = RTX_POOL_P;            // pool entries per lane
constexpr uint32_t kPoolSize = 64u * kPoolP;  // rays per wave
constexpr int kPoolThreshold = RTX_POOL_T;    // waiting lanes that trigger the finish/start block
constexpr uint32_t kPoolGrab = 512;
constexpr uint32_t kDead = 0xFFFFFFFFu;
constexpr size_t kPoolEntryBytes = 13 * sizeof(double) + 5 * sizeof(uint32_t);

__device__ __forceinline__ uint32_t pool_mbcnt(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

}  // namespace

__global__ __launch_bounds__(kBvhThreads, 4) void trace_pool_kernel(const SceneView *__restrict__ svp,
                                                                    const RowsView *__restrict__ rvp,
                                                                    double *__restrict__ samples, char *__restrict__ pool_mem,
                                                                    Counters *__restrict__ ctr,
                                                                    unsigned long long *__restrict__ work_counter,
                                                                    const float4 *__restrict__ nodes,
                                                                    const float4 *__restrict__ leaf_f32,
                                                                    const uint32_t *__restrict__ leaf_prims,
                                                                    const SphereX *__restrict__ spheres,
                                                                    const uint32_t *__restrict__ sphere_ids)
{
    const SceneView &sv = *svp;
    const RowsView &rv = *rvp;
    __shared__ uint32_t lds_stack[kBvh4StackEntries][kBvhThreads];
    __shared__ uint32_t lds_q[kBvhQueue][kBvhThreads];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const uint32_t bounce_limit = sv.max_bounces >= 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)sv.max_bounces + 1u;

    // pool memory: SoA over all waves' entries; entry (wave, e) -> E0 + e, e = k * 64 + lane for the owner lane
    const uint64_t n_entries = (uint64_t)gridDim.x * (kBvhThreads / 64) * kPoolSize;
    const uint64_t E0 = ((uint64_t)blockIdx.x * (kBvhThreads / 64) + (tid >> 6)) * kPoolSize;
    double *pf = reinterpret_cast<double *>(pool_mem);              // [13][n_entries]: pos, dir, result, light, hit_t
    uint32_t *pi = reinterpret_cast<uint32_t *>(pf + 13 * n_entries); // [5][n_entries]: pl, smp, bnc, hit_id, hit_kind_local

    unsigned long long wave_next = 0, wave_end = 0;      // this wave's share of the global ray queue (wave-uniform)
    bool queue_empty = false;
    uint32_t live = 0;                                   // bit k: my pool entry k holds a ray
    unsigned long long segs = 0, box_tests = 0, leaf_filters = 0, exact = 0;

    for (;;) {
        // ================= phase 1: refill =================
#pragma unroll 1
        for (int k = 0; k < kPoolP; ++k) {
            const uint64_t e = E0 + (uint64_t)k * 64u + lane;
            const bool dead = ((live >> k) & 1u) == 0u;
            const unsigned long long idle_mask = __ballot(dead);
            if (idle_mask != 0ull) {
                if (wave_next >= wave_end && !queue_empty) {
                    unsigned long long base = 0;
                    if (lane == 0) base = atomicAdd(work_counter, (unsigned long long)kPoolGrab);
                    base = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
                           __builtin_amdgcn_readfirstlane((uint32_t)base);
                    wave_next = base;
                    wave_end = base + kPoolGrab < rv.n_rays ? base + kPoolGrab : rv.n_rays;
                    if (base >= rv.n_rays) { queue_empty = true; wave_next = wave_end = 0; }
                }
                if (dead && wave_next < wave_end) {
                    const unsigned long long my = wave_next + pool_mbcnt(idle_mask);
                    if (my < wave_end) {
                        uint32_t pl, smp;
                        ray_index_to_pixel(rv, my, pl, smp);
                        RayState r;
                        gen_primary(sv, rv, pl, rv.sample_begin + smp, r);
                        if (sv.n_objects == 0) {                                  // scene.rs:224-226
                            double *o = samples + ((uint64_t)smp * rv.npix + pl) * 3;
                            o[0] = 0.0; o[1] = 0.0; o[2] = 0.0;
                        } else {
                            pf[0 * n_entries + e] = r.pos.x; pf[1 * n_entries + e] = r.pos.y; pf[2 * n_entries + e] = r.pos.z;
                            pf[3 * n_entries + e] = r.dir.x; pf[4 * n_entries + e] = r.dir.y; pf[5 * n_entries + e] = r.dir.z;
                            pf[6 * n_entries + e] = 0.0; pf[7 * n_entries + e] = 0.0; pf[8 * n_entries + e] = 0.0;
                            pf[9 * n_entries + e] = 1.0; pf[10 * n_entries + e] = 1.0; pf[11 * n_entries + e] = 1.0;
                            pi[0 * n_entries + e] = pl; pi[1 * n_entries + e] = smp; pi[2 * n_entries + e] = 0u;
                            live |= 1u << k;
                        }
                    }
                }
                const unsigned long long taken = (unsigned long long)__popcll(idle_mask);
                wave_next = wave_next + taken < wave_end ? wave_next + taken : wave_end;
            }
            if (((live >> k) & 1u) == 0u) pi[0 * n_entries + e] = kDead;            // the traversal phase skips it
        }
        if (__ballot(live != 0u) == 0ull) {
            if (queue_empty) break;            // wave-uniform: pool empty, nothing left to take
            continue;
        }

        // ================= phase 2: traverse, pulling entries from the pool =================
        {
            uint32_t pool_next = 0;            // wave-uniform cursor into this wave's pool
            bool has = false;                  // this lane is traversing entry `cur`
            bool fin = false;                  // this lane's traversal ended; the hit is not stored yet
            uint64_t cur = 0;
            RayX rx;
            rx.pos = rx.dir = rx.dirn = mk(0.0, 0.0, 0.0);
            rx.a = rx.a2 = rx.a4 = 0.0;
            Ray32 q = { 0.f, 0.f, 0.f, 0.f, 0.f, 0.f };
            FilterParams fpar;
            filter_idle(fpar);
            Hit h;
            hit_init(h);
            float best_up = __builtin_inff();
            uint32_t sp = 0, qcnt = 0, node = kNone;
            bool overflow = false;

            for (uint32_t it = 0;; ++it) {
                const unsigned long long want = __ballot(!has);
                const unsigned long long trav = ~want;
                const uint32_t n_fin = (uint32_t)__popcll(__ballot(fin));
                const bool pool_left = pool_next < kPoolSize;
                const bool run_block = want != 0ull && (trav == 0ull || (pool_left ? (uint32_t)__popcll(want) >= (uint32_t)kPoolThreshold
                                                                                     : n_fin >= (uint32_t)kPoolThreshold));
                if (run_block) {
                    // ---- finish: closest_object's result for the entries whose traversal ended
                    if (fin) {
                        flush_candidates(spheres, sphere_ids, rx, &lds_q[0][0], tid, qcnt, h, best_up, exact);
                        if (overflow) {                   // some subtree was dropped: every sphere gets the exact test
                            for (uint32_t k = 0; k < sv.n_spheres; ++k) {
                                double t;
                                if (sphere_distance(spheres[k], rx, &t)) hit_consider(h, t, sphere_ids[k], 0, k);
                            }
                            exact += sv.n_spheres;
                        }
                        for (uint32_t k = 0; k < sv.n_planes; ++k) {
                            double t;
                            if (plane_distance(sv.planes[k], rx, &t)) hit_consider(h, t, sv.planes[k].id, 1, k);
                        }
                        for (uint32_t k = 0; k < sv.n_tris; ++k) {
                            double t;
                            if (triangle_distance(sv.tris[k], rx, &t)) hit_consider(h, t, sv.tris[k].id, 2, k);
                        }
                        exact += sv.n_planes + sv.n_tris;
                        pf[12 * n_entries + cur] = h.t;
                        pi[3 * n_entries + cur] = h.id;
                        pi[4 * n_entries + cur] = (h.kind << 30) | (h.local & 0x3FFFFFFFu);
                        ++segs;
                        fin = false;
                    }
                    // ---- start: pull the next pool entries
                    if (pool_left) {
                        if (!has) {
                            const uint32_t my = pool_next + pool_mbcnt(want);
                            if (my < kPoolSize) {
                                const uint64_t e = E0 + my;
                                if (pi[0 * n_entries + e] != kDead) {
                                    cur = e;
                                    const V3 pos = mk(pf[0 * n_entries + e], pf[1 * n_entries + e], pf[2 * n_entries + e]);
                                    const V3 dir = mk(pf[3 * n_entries + e], pf[4 * n_entries + e], pf[5 * n_entries + e]);
                                    rx = make_rayx(pos, dir);
                                    hit_init(h);
                                    best_up = __builtin_inff();
                                    sp = 0; qcnt = 0; overflow = false;
                                    const float omax = fmaxf(fmaxf(__builtin_fabsf((float)pos.x), __builtin_fabsf((float)pos.y)),
                                                             __builtin_fabsf((float)pos.z));
                                    if (sv.n_bvh_nodes != 0 && omax <= sv.bvh_origin_limit) {   // NaN origin -> exhaustive branch
                                        q.ox = (float)pos.x; q.oy = (float)pos.y; q.oz = (float)pos.z;
                                        q.ix = (float)(1.0 / rx.dirn.x); q.iy = (float)(1.0 / rx.dirn.y); q.iz = (float)(1.0 / rx.dirn.z);
                                        filter_from_ray(sv, pos, dir, fpar);
                                        node = 0;
                                        has = true;
                                    } else {
                                        // no tree for this ray: exhaustive exact sweep now; it is finished at the next block run
                                        for (uint32_t k = 0; k < sv.n_spheres; ++k) {
                                            double t;
                                            if (sphere_distance(spheres[k], rx, &t)) hit_consider(h, t, sphere_ids[k], 0, k);
                                        }
                                        exact += sv.n_spheres;
                                        fin = true;
                                    }
                                }
                            }
                        }
                        const uint32_t taken = (uint32_t)__popcll(want);
                        pool_next = pool_next + taken < kPoolSize ? pool_next + taken : kPoolSize;
                    }
                    // leave when the pool is exhausted, nobody traverses and every result is stored
                    if (pool_next >= kPoolSize && __ballot(has) == 0ull && __ballot(fin) == 0ull) break;
                }

                if (has) {
                    // ---- one traversal step (identical to trace_bvh_kernel)
                    const float4 *np = nodes + 8 * (size_t)node;
                    float4 ca[4], cb[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) { ca[c] = np[c]; cb[c] = np[4 + c]; }
                    float tc[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) tc[c] = box_entry32(ca[c], cb[c], q, best_up);
                    box_tests += 4;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const uint32_t count = __float_as_uint(cb[c].w);
                        if (tc[c] < __builtin_inff() && count - 1u < (uint32_t)kBvhLeafSize) {
                            const uint32_t first = __float_as_uint(ca[c].w);
                            for (uint32_t k = 0; k < count; ++k) {
                                const float4 rec = leaf_f32[first + k];
                                if ((int)__float_as_uint(filter_disc1(rec, fpar)) >= 0) {   // D >= 0: cannot be excluded
                                    if (qcnt == (uint32_t)kBvhQueue)
                                        flush_candidates(spheres, sphere_ids, rx, &lds_q[0][0], tid, qcnt, h, best_up, exact);
                                    lds_q[qcnt][tid] = leaf_prims[first + k];
                                    qcnt += 1;
                                }
                            }
                            leaf_filters += count;
                        }
                    }
                    // exact tests of the queued candidates every 4th iteration (wave-uniform) for all traversing lanes together
                    if ((it & 3u) == 3u) flush_candidates(spheres, sphere_ids, rx, &lds_q[0][0], tid, qcnt, h, best_up, exact);
                    float key[4];
                    uint32_t lnk[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const bool go = __float_as_uint(cb[c].w) == 0u && tc[c] < __builtin_inff() && tc[c] <= best_up;
                        key[c] = go ? tc[c] : __builtin_inff();
                        lnk[c] = __float_as_uint(ca[c].w);
                    }
#define RTX_CSWAP(i, j) { if (key[j] < key[i]) { float tk = key[i]; key[i] = key[j]; key[j] = tk; uint32_t tl = lnk[i]; lnk[i] = lnk[j]; lnk[j] = tl; } }
                    RTX_CSWAP(0, 1) RTX_CSWAP(2, 3) RTX_CSWAP(0, 2) RTX_CSWAP(1, 3) RTX_CSWAP(1, 2)
#undef RTX_CSWAP
                    if (key[3] < __builtin_inff()) { if (sp < (uint32_t)kBvh4StackEntries) { lds_stack[sp][tid] = lnk[3]; sp += 1; } else overflow = true; }
                    if (key[2] < __builtin_inff()) { if (sp < (uint32_t)kBvh4StackEntries) { lds_stack[sp][tid] = lnk[2]; sp += 1; } else overflow = true; }
                    if (key[1] < __builtin_inff()) { if (sp < (uint32_t)kBvh4StackEntries) { lds_stack[sp][tid] = lnk[1]; sp += 1; } else overflow = true; }
                    node = key[0] < __builtin_inff() ? lnk[0] : kNone;
                    if (node == kNone && sp != 0u) {
                        sp -= 1;
                        node = lds_stack[sp][tid];
                    }
                    if (node == kNone) { has = false; fin = true; }       // traversal over: the finish block stores the hit
                }
            }
        }

        // ================= phase 3: shade my own entries (scene.rs:232-239, 260-278) =================
#pragma unroll 1
        for (int k = 0; k < kPoolP; ++k) {
            if (((live >> k) & 1u) == 0u) continue;
            const uint64_t e = E0 + (uint64_t)k * 64u + lane;
            Hit h;
            h.t = pf[12 * n_entries + e];
            h.id = pi[3 * n_entries + e];
            const uint32_t kl = pi[4 * n_entries + e];
            h.kind = kl >> 30;
            h.local = kl & 0x3FFFFFFFu;
            const uint32_t pl = pi[0 * n_entries + e], smp = pi[1 * n_entries + e], bnc = pi[2 * n_entries + e];
            RayState r;
            r.result = mk(pf[6 * n_entries + e], pf[7 * n_entries + e], pf[8 * n_entries + e]);
            bool done = true;
            if (h.id != kNone) {
                r.pos = mk(pf[0 * n_entries + e], pf[1 * n_entries + e], pf[2 * n_entries + e]);
                r.dir = mk(pf[3 * n_entries + e], pf[4 * n_entries + e], pf[5 * n_entries + e]);
                r.light = mk(pf[9 * n_entries + e], pf[10 * n_entries + e], pf[11 * n_entries + e]);
                const uint32_t row = pl / rv.width;
                const uint32_t x = pl - row * rv.width;
                const uint64_t pix = (uint64_t)(rv.row_begin + row * rv.row_stride) * rv.width + x;
                r.key = rng_key(sv.seed, pix, rv.sample_begin + smp);
                r.draw = 6u + 2u * bnc;
                r.bounce = bnc;
                advance_and_shade(sv, h, r);
                done = (r.bounce >= bounce_limit) || light_is_zero(r);            // scene.rs:227-228
                if (!done) {
                    pf[0 * n_entries + e] = r.pos.x; pf[1 * n_entries + e] = r.pos.y; pf[2 * n_entries + e] = r.pos.z;
                    pf[3 * n_entries + e] = r.dir.x; pf[4 * n_entries + e] = r.dir.y; pf[5 * n_entries + e] = r.dir.z;
                    pf[6 * n_entries + e] = r.result.x; pf[7 * n_entries + e] = r.result.y; pf[8 * n_entries + e] = r.result.z;
                    pf[9 * n_entries + e] = r.light.x; pf[10 * n_entries + e] = r.light.y; pf[11 * n_entries + e] = r.light.z;
                    pi[2 * n_entries + e] = r.bounce;
                }
            }
            if (done) {
                double *o = samples + ((uint64_t)smp * rv.npix + pl) * 3;
                o[0] = r.result.x; o[1] = r.result.y; o[2] = r.result.z;
                live &= ~(1u << k);
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

size_t pool_state_bytes(int n_cus)
{
    return (size_t)n_cus * 4 /*workgroups per CU*/ * (kBvhThreads / 64) * kPoolSize * kPoolEntryBytes;
}

hipError_t launch_trace_pool(const SceneView *d_sv, const SceneView &sv, const RowsView *d_rv, const RowsView &rv,
                             double *samples, char *pool_mem, Counters *counters, unsigned long long *work_counter, int n_cus,
                             hipStream_t stream)
{
    const uint64_t rays_per_block = (uint64_t)(kBvhThreads / 64) * kPoolSize;
    const uint64_t want = (rv.n_rays + rays_per_block - 1) / rays_per_block;
    const uint64_t cap = (uint64_t)n_cus * 4;
    const uint32_t blocks = (uint32_t)(want < cap ? want : cap);
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(trace_pool_kernel, dim3(blocks), dim3(kBvhThreads), 0, stream, d_sv, d_rv, samples, pool_mem, counters,
                       work_counter, reinterpret_cast<const float4 *>(sv.bvh_nodes), sv.bvh_leaf_f32, sv.bvh_prims, sv.spheres,
                       sv.sphere_id);
    return hipGetLastError();
}

}  // namespace rtx
