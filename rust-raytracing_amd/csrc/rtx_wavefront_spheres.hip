// rtx_wavefront_spheres.hip -- RTX_KERNEL_WAVEFRONT for trees that hold spheres only (C2, C4).
//
// trace_bvh_spheres_kernel (rtx_bvh_spheres.hip) is VALU-issue bound with 47 % of its lanes working: a wave iterates
// its walk until the longest of its 64 traversals ends (53 wave-steps for 24 lane-steps on C2).  Schedules that refill
// lanes inside the megakernel (regrouping, bounded rounds) lose what they gain to the f64 phase they run more often with
// fewer lanes.  Here the two phases are separate kernels per bounce level, the ray state in HBM (rtx_wavefront.h):
//     wf_trace_spheres_kernel   sphere_step (rtx_traverse.h) and nothing else.  A lane whose walk ended waits until
//                               kWfSphService lanes of its wave do; then they write their candidates and take the next
//                               records of the level's queue (wave-local chunks, one atomic per chunk), so the walk runs
//                               with most lanes busy
//     wf_shade_spheres_kernel   one thread per ray of the level, f64: the exact tests of its <= 4 candidates
//                               (sphere.rs:19-30), the shapes outside the tree, ray_hit, the next segment's record
// Same functions, same operation order, same bits as every other kernel.
#include "rtx_launch.h"
#include "rtx_wavefront.h"

#include <algorithm>
#include <cstdlib>

namespace rtx {

#ifndef RTX_WF_SPH_WAVES
#define RTX_WF_SPH_WAVES 4
#endif
constexpr int kWfSphWaves = RTX_WF_SPH_WAVES;                     // workgroups per CU of the walk kernel
constexpr int kWfSphStack = 160 / kWfSphWaves - 2 - 2 * kSphQueue;   // LDS stack entries per lane (+ sink row + 2 * kSphQueue queue rows)
#ifndef RTX_WF_SPH_SERVICE
#define RTX_WF_SPH_SERVICE 16
#endif
constexpr uint32_t kWfSphService = RTX_WF_SPH_SERVICE;            // lanes of a wave that wait before they are served together

struct WfSphRec {                     // 64 bytes: what the f32 walk needs of one segment (stored where WfRec is)
    float ix, iy, iz, nx, ny, nz;     // Ray32
    float px, py, pz, dx, dy, dz;     // SphereRay: origin - scene centre, direction
    float Kg, c0;                     // its error terms (K = 24uM is derived from Kg = 128uM); Kg = NaN: no walk for this origin
    float slack;                      // Ray32S::e: 0 for an origin inside origin_limit
    uint32_t ridx;                    // the ray (index in the launch's queue order)
};
static_assert(sizeof(WfSphRec) == sizeof(WfRec), "WfSphRec shares WfRec's slots");
// WfCand of this form: count (| kWfFallback), e[0..3] = local sphere indices, e[6] = ridx

__device__ __forceinline__ void wf_make_sph_rec(const SceneView &sv, const V3 &pos, const V3 &dir, const V3 &dirn, uint32_t ridx,
                                                WfSphRec &w)
{
    SphereRay sr;
    sphere_ray_from(sv, pos, dir, sr);
    Ray32 q;
    make_ray32(pos, dirn, (double)sv.bvh_inv_max, q);
    w.ix = q.ix; w.iy = q.iy; w.iz = q.iz; w.nx = q.nx; w.ny = q.ny; w.nz = q.nz;
    w.px = sr.px; w.py = sr.py; w.pz = sr.pz; w.dx = sr.dx; w.dy = sr.dy; w.dz = sr.dz;
    w.c0 = sr.c0;
    // an origin outside origin_limit walks with Ray32S's slack; beyond 2^27 times that, or NaN: no walk (every sphere is tested)
    const float omax = fmaxf(fmaxf(__builtin_fabsf((float)pos.x), __builtin_fabsf((float)pos.y)), __builtin_fabsf((float)pos.z));
    const bool in32 = omax <= sv.bvh_origin_limit;
    w.slack = ray32_slack(q.nx, q.ny, q.nz, in32);
    w.Kg = (in32 || omax <= sv.bvh_origin_limit * kBvhRange64) ? sr.Kg : __builtin_nanf("");
    w.ridx = ridx;
}

// ---- level 0: render_pixel's prologue for every ray of the launch (result = 0 and light = 1 are implied at level 0) -----
__global__ __launch_bounds__(256) void wf_generate_spheres_kernel(const SceneView *__restrict__ svp, const RowsView *__restrict__ rvp,
                                                                  const WfState st)
{
    const SceneView &sv = *svp;
    const RowsView &rv = *rvp;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool valid = i < rv.n_rays;
    uint32_t pl = 0, smp = 0;
    if (valid) {
        if (rv.tiles_x != 0u) valid = ray_index_to_pixel_tiled(rv, i, pl, smp);
        else ray_index_to_pixel(rv, i, pl, smp);
    }
    // level 0's queue is the launch's ray queue itself (slot i = ray i, no atomics); the padding of partial tiles is marked dead
    if (i == 0) st.count[0] = rv.n_rays;
    if (i >= rv.n_rays) return;
    WfSphRec w;
    if (valid) {
        RayState r;
        gen_primary(sv, rv, pl, rv.sample_begin + smp, r);
        const WfRays &rs = st.ray[0];
        rs.pos[0][i] = r.pos.x; rs.pos[1][i] = r.pos.y; rs.pos[2][i] = r.pos.z;
        rs.dir[0][i] = r.dir.x; rs.dir[1][i] = r.dir.y; rs.dir[2][i] = r.dir.z;
        wf_make_sph_rec(sv, r.pos, r.dir, vnorm(r.dir), (uint32_t)i, w);
    } else {
        w.ix = w.iy = w.iz = w.nx = w.ny = w.nz = w.px = w.py = w.pz = w.dx = w.dy = w.dz = w.Kg = w.c0 = w.slack = 0.f;
        w.ridx = kNone;
    }
    reinterpret_cast<WfSphRec *>(st.rec[0])[i] = w;
}

// ---- the walk ------------------------------------------------------------------------------------------------------------
template <bool SPILL>
__global__ __launch_bounds__(kBvhThreads, kWfSphWaves) void wf_trace_spheres_kernel(const WfState st, Counters *__restrict__ ctr,
                                                                                    const float4 *__restrict__ nodes,
                                                                                    const float4 *__restrict__ leaf_cr,
                                                                                    const uint32_t *__restrict__ leaf_prims,
                                                                                    uint32_t root, uint32_t *__restrict__ spill,
                                                                                    uint32_t spill_entries)
{
    constexpr int STACK = kWfSphStack;
    __shared__ uint32_t lds_stack[STACK + 1][kBvhThreads];
    __shared__ uint32_t lds_q[2 * kSphQueue][kBvhThreads];
    uint32_t *const ls = &lds_stack[0][0];
    uint32_t *const lq = &lds_q[0][0];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const size_t spill_stride = (size_t)gridDim.x * kBvhThreads, glane = (size_t)blockIdx.x * kBvhThreads + tid;
    const unsigned long long n_queue = st.count[0];
    const WfSphRec *__restrict__ recs = reinterpret_cast<const WfSphRec *>(st.rec[0]);
    unsigned long long *const head = &st.work[0];
    const unsigned long long grab = wf_grab_size(n_queue);
    WfChunk ch;
    ch.next = ch.end = 0; ch.drained = false;
    bool walking = false, have = false;           // have: the lane holds a segment (walking, or complete and not yet written)
    uint32_t pos = 0, ridx = 0;
    Ray32S q;
    SphereRay sr;
    float best_up = 0.f;
    uint32_t node = kNone, sp = 0, qcnt = 0, nbox = 0, nleaf = 0;
    bool overflow = false;
    unsigned long long box_tests = 0, leaf_filters = 0;
    q.ix = q.iy = q.iz = q.nx = q.ny = q.nz = q.e = 0.f;
    sr.px = sr.py = sr.pz = sr.dx = sr.dy = sr.dz = sr.Kg = sr.K = 0.f; sr.c0 = __builtin_inff();

    for (;;) {
        const unsigned long long wmask = __ballot(walking);
        const unsigned long long fin = __ballot(!walking && have);
        const uint32_t n_fin = (uint32_t)__popcll(fin);
        const bool serve = wmask == 0ull || (ch.drained ? n_fin >= kWfSphService : 64u - (uint32_t)__popcll(wmask) >= kWfSphService);
        if (serve) {
            // ---- the lanes whose walk is complete write the candidates that can still be the winner
            if (!walking && have) {
                WfCand c;
                uint32_t k = 0;
#pragma unroll
                for (int e = 0; e < 6; ++e) c.e[e] = 0u;
                c.e[6] = ridx;
#pragma unroll
                for (int e = 0; e < kSphQueue; ++e) {
                    if ((uint32_t)e < qcnt && __uint_as_float(lq[(size_t)(kSphQueue + e) * kBvhThreads + tid]) <= best_up) {
                        const uint32_t v = lq[(size_t)e * kBvhThreads + tid];
                        c.e[0] = k == 0u ? v : c.e[0]; c.e[1] = k == 1u ? v : c.e[1];
                        c.e[2] = k == 2u ? v : c.e[2]; c.e[3] = k == 3u ? v : c.e[3];
                        k += 1;
                    }
                }
                c.count = ridx == kNone ? kWfDead : (overflow ? kWfFallback : k);
                st.cand[pos] = c;
                box_tests += nbox; leaf_filters += nleaf;
                nbox = 0; nleaf = 0;
                have = false;
            }
            // ---- and every lane without a segment takes the next record of the wave's chunk
            if (!ch.drained) {
                unsigned long long my;
                if (wf_take(ch, head, grab, n_queue, !have, my)) {
                    const WfSphRec w = recs[my];
                    pos = (uint32_t)my;                          // (the host keeps a launch below 2^32 rays)
                    ridx = w.ridx;
                    q.ix = w.ix; q.iy = w.iy; q.iz = w.iz; q.nx = w.nx; q.ny = w.ny; q.nz = w.nz; q.e = w.slack;
                    sr.px = w.px; sr.py = w.py; sr.pz = w.pz; sr.dx = w.dx; sr.dy = w.dy; sr.dz = w.dz;
                    sr.Kg = w.Kg; sr.c0 = w.c0;
                    sr.K = w.Kg * (0.1875f * (1.0f + 4.76837158e-7f));       // 24uM from 128uM, rounded up
                    best_up = __builtin_inff();
                    qcnt = 0; sp = 0; overflow = false;
                    have = true;
                    if (w.ridx == kNone) node = kNone;                               // a dead slot: passed on as such
                    else if (w.Kg == w.Kg) { node = root; walking = true; }
                    else { node = kNone; overflow = true; }      // no f32 walk for this origin: the shade kernel takes it
                }
            }
            if (__ballot(walking) == 0ull) {
                if (ch.drained && __ballot(have) == 0ull) break;
                continue;
            }
        }
        if (walking) {
            sphere_step<STACK, SPILL>(nodes, leaf_cr, leaf_prims, q, sr, node, sp, ls, lq, tid, spill, spill_entries, spill_stride,
                                      glane, best_up, qcnt, overflow, nbox, nleaf);
            walking = node != kNone;
        }
    }
    unsigned long long filt = box_tests + leaf_filters;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        filt += __shfl_xor(filt, off, 64);
        box_tests += __shfl_xor(box_tests, off, 64);
    }
    if (lane == 0) {
        const uint32_t shard = (blockIdx.x * (kBvhThreads >> 6) + (tid >> 6)) & (kCounterShards - 1);
        if (filt) atomicAdd(&ctr[shard].filter_tests, filt);
        if (box_tests) atomicAdd(&ctr[2 + (shard % (kCounterShards - 2))].pad_, box_tests);
    }
}

// ---- closest_object's exact part, ray_hit, the next segment's record -----------------------------------------------------
__global__ __launch_bounds__(kBvhThreads, 4) void wf_shade_spheres_kernel(const SceneView *__restrict__ svp,
                                                                          const RowsView *__restrict__ rvp, const WfState st,
                                                                          uint32_t level, double *__restrict__ samples,
                                                                          Counters *__restrict__ ctr, const LeafArrays la)
{
    const SceneView &sv = *svp;
    const RowsView &rv = *rvp;
    __shared__ uint32_t lds_append[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t bounce_limit = sv.max_bounces >= 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)sv.max_bounces + 1u;
    const unsigned long long n_queue = st.count[0];
    WfSphRec *__restrict__ recs_out = reinterpret_cast<WfSphRec *>(st.rec[1]);
    const WfRays &rin = st.ray[0], &rout = st.ray[1];
    unsigned long long segs = 0, exact = 0, box_tests = 0, leaf_filters = 0;

    // grid-stride over the level's queue (the grid is sized for the launch's ray count; deep levels are short)
    // (block-uniform trip count: the append is one atomic per workgroup iteration)
    uint32_t it = 0;
    for (unsigned long long p0 = (unsigned long long)blockIdx.x * kBvhThreads; p0 < n_queue;
         p0 += (unsigned long long)gridDim.x * kBvhThreads, ++it) {
        const unsigned long long p = p0 + tid;
        WfCand c;
        c.count = kWfDead;
        if (p < n_queue) c = st.cand[p];
        const bool have = (c.count & kWfDead) == 0u;
        bool next = false;
        WfSphRec w;
        RayState r;
        if (have) {
            const uint32_t ridx = c.e[6];
            r.pos = mk(rin.pos[0][p], rin.pos[1][p], rin.pos[2][p]);
            r.dir = mk(rin.dir[0][p], rin.dir[1][p], rin.dir[2][p]);
            const RayX rx = make_rayx(r.pos, r.dir);
            Hit h;
            hit_init(h);
            ++segs;
            if (c.count & kWfFallback) {
                // the walk's queue overflowed (more than kSphQueue live candidates at once) or the origin is outside every
                // range a walk is valid for: every sphere
                for (uint32_t k = 0; k < sv.n_spheres; ++k) {
                    double t;
                    if (sphere_distance(la.spheres[k], rx, &t)) hit_consider(h, t, la.sphere_ids[k], 0, k);
                }
                exact += sv.n_spheres;
            } else {
                const uint32_t n = c.count < (uint32_t)kSphQueue ? c.count : (uint32_t)kSphQueue;
#pragma unroll 1
                for (uint32_t e = 0; e < n; ++e) {
                    const uint32_t idx = e == 0 ? c.e[0] : e == 1 ? c.e[1] : e == 2 ? c.e[2] : c.e[3];
                    double t;
                    if (sphere_distance(la.spheres[idx], rx, &t)) hit_consider(h, t, la.sphere_ids[idx], 0, idx);
                }
                exact += n;
            }
            // the shapes outside the tree: planes, the few triangles of a sphere scene (by filter record)
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

            // ---- render_ray's match arm + ray_hit (scene.rs:232-239, 260-278)
            r.result = level == 0u ? mk(0.0, 0.0, 0.0) : mk(rin.res[0][p], rin.res[1][p], rin.res[2][p]);
            bool done = true;
            if (h.id != kNone) {
                r.light = level == 0u ? mk(1.0, 1.0, 1.0) : mk(rin.lig[0][p], rin.lig[1][p], rin.lig[2][p]);
                uint32_t pl = 0, smp = 0;
                if (rv.tiles_x != 0u) (void)ray_index_to_pixel_tiled(rv, ridx, pl, smp);
                else ray_index_to_pixel(rv, ridx, pl, smp);
                const uint32_t k = fastdiv(pl, rv.div_width), x = pl - k * rv.width;
                const uint64_t pix = (uint64_t)image_row(rv, k) * rv.width + x;
                r.key = rng_key(sv.seed, pix, rv.sample_begin + smp);
                r.draw = 6u + 2u * level;
                r.bounce = level;
                advance_and_shade(sv, h, r);
                done = (r.bounce >= bounce_limit) || light_is_zero(r);            // scene.rs:227-228
            }
            if (done) {
                store_sample(samples, rv, ridx, r.result);
            } else {
                wf_make_sph_rec(sv, r.pos, r.dir, vnorm(r.dir), ridx, w);
                next = true;
            }
        }
        // the survivors move to the next level's queue: record and state at their new slot, unit stride across the block
        const unsigned long long slot = wf_append_block(&st.count[1], next, lds_append, it);
        if (next) {
            recs_out[slot] = w;
            rout.pos[0][slot] = r.pos.x; rout.pos[1][slot] = r.pos.y; rout.pos[2][slot] = r.pos.z;
            rout.dir[0][slot] = r.dir.x; rout.dir[1][slot] = r.dir.y; rout.dir[2][slot] = r.dir.z;
            rout.res[0][slot] = r.result.x; rout.res[1][slot] = r.result.y; rout.res[2][slot] = r.result.z;
            rout.lig[0][slot] = r.light.x; rout.lig[1][slot] = r.light.y; rout.lig[2][slot] = r.light.z;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        segs += __shfl_xor(segs, off, 64);
        exact += __shfl_xor(exact, off, 64);
        box_tests += __shfl_xor(box_tests, off, 64);
        leaf_filters += __shfl_xor(leaf_filters, off, 64);
    }
    if (lane == 0) {
        const uint32_t shard = (blockIdx.x * (kBvhThreads >> 6) + (tid >> 6)) & (kCounterShards - 1);
        if (segs) atomicAdd(&ctr[shard].segments, segs);
        if (exact) atomicAdd(&ctr[shard].exact_tests, exact);
        if (box_tests + leaf_filters) atomicAdd(&ctr[shard].filter_tests, box_tests + leaf_filters);
        if (box_tests) atomicAdd(&ctr[2 + (shard % (kCounterShards - 2))].pad_, box_tests);
    }
}

// ---- host ------------------------------------------------------------------------------------------------------------------
uint32_t wavefront_spheres_spill_entries(const SceneView &sv)
{
    const uint32_t need = 3u * sv.bvh_depth + 2u;
    return need > (uint32_t)kWfSphStack ? need - (uint32_t)kWfSphStack : 0u;
}

size_t wavefront_spheres_spill_bytes(const SceneView &sv, int n_cus)
{
    // one column per resident lane of whichever kernel walks (both are built for 4 workgroups per CU)
    return (size_t)wavefront_spheres_spill_entries(sv) * (size_t)n_cus * kWfSphWaves * kBvhThreads * sizeof(uint32_t);
}

hipError_t launch_trace_wavefront_spheres(const SceneView *d_sv, const SceneView &sv, const RowsView *d_rv, const RowsView &rv,
                                          double *samples, void *state_mem, Counters *counters, uint32_t *spill, int n_cus,
                                          hipStream_t stream)
{
    if (rv.n_rays == 0) return hipSuccess;
    const uint32_t levels = wavefront_levels(sv);
    const uint64_t n = rv.n_rays;
    WfState st;
    wf_carve(state_mem, n, st);

    LeafArrays la;
    la.sphere_f32 = sv.bvh_leaf_cr; la.sphere_prims = sv.bvh_prims; la.spheres = sv.spheres; la.sphere_ids = sv.sphere_id;
    la.tri_f32 = sv.tri_f32; la.tri_fidx = sv.tri_fidx; la.tris = sv.tris;
    const uint32_t spill_entries = spill ? wavefront_spheres_spill_entries(sv) : 0u;
    const bool deep = spill_entries != 0u;
    const float4 *nodes = reinterpret_cast<const float4 *>(sv.bvh_nodes);
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((n + kBvhThreads - 1) / kBvhThreads, (uint64_t)n_cus * kWfSphWaves);
    auto generate = [&](const WfState &s0) {
        hipLaunchKernelGGL(wf_generate_spheres_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, d_sv, d_rv, s0);
        return hipGetLastError();
    };
    auto level_fn = [&](const WfState &sk, uint32_t level) {
        if (deep) hipLaunchKernelGGL(wf_trace_spheres_kernel<true>, dim3(blocks), dim3(kBvhThreads), 0, stream, sk, counters, nodes, la.sphere_f32, la.sphere_prims, sv.bvh_root, spill, spill_entries);
        else hipLaunchKernelGGL(wf_trace_spheres_kernel<false>, dim3(blocks), dim3(kBvhThreads), 0, stream, sk, counters, nodes, la.sphere_f32, la.sphere_prims, sv.bvh_root, spill, spill_entries);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(wf_shade_spheres_kernel, dim3(blocks), dim3(kBvhThreads), 0, stream, d_sv, d_rv, sk, level, samples, counters, la);
        return hipGetLastError();
    };
    return wf_run_levels(st, levels, stream, generate, level_fn);
}

}  // namespace rtx
