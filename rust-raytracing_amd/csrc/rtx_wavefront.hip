// rtx_wavefront.hip -- RTX_KERNEL_WAVEFRONT: the path of a pure triangle mesh as three kernels per bounce level instead
// of one megakernel (C3, C5).
//
// trace_bvh_mesh_kernel (rtx_bvh_mesh.hip) keeps the f64 phase (exact tests, ray_hit, ray set-up) and the f32 walk in one
// kernel: the f64 phase owns the register file (246 VGPRs to hold everything, so 4 waves per SIMD with 120-230 spilled
// registers), runs with 16-32 of 64 lanes, and the walk -- which is latency-bound, 2/3 of its wave cycles waiting on
// dependent fetches -- gets 4 waves per SIMD and 37 % lane utilisation.  Here the ray state lives in HBM
// (structure-of-arrays, every access coalesced) and each bounce level is
//     wf_trace_kernel   the f32-only walk (mesh_step, rtx_mesh_step.h) and nothing else: few registers -> 6 waves per SIMD;
//                       a lane whose walk ends takes the next ray of the level's queue at once (a 64-byte record), so the
//                       lanes stay busy; output: the candidates that can still be the winner
//     wf_shade_kernel   one thread per ray of the level, f64: the exact tests of its candidates (sphere.rs / triangle.rs /
//                       plane.rs), closest_object's winner, ray_hit, then the set-up of the next segment (self-hit pre-test,
//                       slab-test and filter parameters) appended to the next level's queue with a wave-aggregated atomic
// after wf_generate_kernel wrote level 0 (render_pixel's prologue, scene.rs:196-207).  Same functions, same operation
// order, same bits as every other kernel.  What it costs: the state streams through HBM once per segment (96 B of f64
// state + 64 B record + 32 B candidates, read and written: ~400 B per segment) -- the "SoA rays in HBM + ballot / prefix
// sum compaction" of BASELINE.json's north star, and for the first time a visible share of the HBM roofline.
//
// A ray the walk cannot finish in f32 -- origin outside the tree's range, more live candidates than the queue holds --
// is flagged and the shade kernel walks it itself with round 1's step (bvh_traverse: exact tests interleaved), or tests
// every shape when even the f64 slab test is out of range.
#include "rtx_launch.h"
#include "rtx_mesh_step.h"

#include <algorithm>
#include <cstdlib>

namespace rtx {

constexpr int kWfTraceWaves = 6;                                  // workgroups per CU of the walk kernel
constexpr int kWfStack = 160 / kWfTraceWaves - 1 - 2 * kMeshQueue;   // 13 LDS stack entries per lane, the HBM column behind them
constexpr uint32_t kWfFallback = 0x80000000u;                     // cand.count flag: the shade kernel walks this ray itself

struct WfRec {                        // 64 bytes: what the f32 walk needs of one segment
    float px, py, pz;                 // origin - scene centre
    float dx, dy, dz;                 // direction
    float ix, iy, iz, nx, ny, nz;     // Ray32: inv = fl(1/d), noi = fl(-o * inv)
    float best_up;                    // the self-hit's distance rounded up, or +inf; NaN: no f32 walk (see kWfFallback)
    float A;                          // tri_filter_from_ray's slack 64uS
    uint32_t ridx;                    // the ray (index in the launch's queue order)
    uint32_t pad;
};
static_assert(sizeof(WfRec) == 64, "WfRec must be 64 bytes");

struct WfCand { uint32_t count; uint32_t e[7]; };                 // 32 bytes per queue position
static_assert(sizeof(WfCand) == 32, "WfCand must be 32 bytes");

struct WfState {                      // structure-of-arrays over the launch's rays (capacity n), all on the device
    double *pos[3], *dir[3], *res[3], *lig[3];
    double *hit_t;                    // the pre-tested self-hit's distance, 0.0 = none
    uint32_t *left;                   // the triangle the ray just left (index in tris[]), kNone = none
    WfRec *rec[2];                    // [0]: the records of the level being processed, [1]: where the next level's go
    WfCand *cand;
    unsigned long long *count;        // count[0]: this level's queue length, count[1]: the next level's (being appended to)
    unsigned long long *work;         // work[0]: the walk kernel's queue head for this level
    uint64_t n;
};

__device__ __forceinline__ void wf_make_rec(const SceneView &sv, const V3 &pos, const V3 &dir, const V3 &dirn, float best_up,
                                            uint32_t ridx, WfRec &w)
{
    TriFilterParams tp;
    tri_filter_from_ray(sv, pos, dir, tp);
    Ray32 q;
    make_ray32(pos, dirn, (double)sv.bvh_inv_max, q);
    w.px = -tp.npx; w.py = -tp.npy; w.pz = -tp.npz;
    w.dx = tp.dx; w.dy = tp.dy; w.dz = tp.dz;
    w.ix = q.ix; w.iy = q.iy; w.iz = q.iz; w.nx = q.nx; w.ny = q.ny; w.nz = q.nz;
    w.A = tp.A;
    const float omax = fmaxf(fmaxf(__builtin_fabsf((float)pos.x), __builtin_fabsf((float)pos.y)), __builtin_fabsf((float)pos.z));
    w.best_up = omax <= sv.bvh_origin_limit ? best_up : __builtin_nanf("");       // NaN origin -> no walk either
    w.ridx = ridx;
    w.pad = 0;
}

// wave-aggregated append: one atomic per wave, the lanes that append get consecutive slots
__device__ __forceinline__ unsigned long long wf_append_slot(unsigned long long *counter, bool want)
{
    const unsigned long long m = __ballot(want);
    if (m == 0ull) return 0ull;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t leader = (uint32_t)__ffsll((long long)m) - 1u;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(counter, (unsigned long long)__popcll(m));
    base = ((unsigned long long)__builtin_amdgcn_readlane((uint32_t)(base >> 32), leader) << 32) |
           __builtin_amdgcn_readlane((uint32_t)base, leader);
    return base + bvh_mbcnt(m);
}

// ---- level 0: render_pixel's prologue for every ray of the launch --------------------------------------------------------
__global__ __launch_bounds__(256) void wf_generate_kernel(const SceneView *__restrict__ svp, const RowsView *__restrict__ rvp,
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
    WfRec w;
    if (valid) {
        RayState r;
        gen_primary(sv, rv, pl, rv.sample_begin + smp, r);
        st.pos[0][i] = r.pos.x; st.pos[1][i] = r.pos.y; st.pos[2][i] = r.pos.z;
        st.dir[0][i] = r.dir.x; st.dir[1][i] = r.dir.y; st.dir[2][i] = r.dir.z;
        st.res[0][i] = 0.0; st.res[1][i] = 0.0; st.res[2][i] = 0.0;
        st.lig[0][i] = 1.0; st.lig[1][i] = 1.0; st.lig[2][i] = 1.0;
        st.hit_t[i] = 0.0;
        st.left[i] = kNone;
        wf_make_rec(sv, r.pos, r.dir, vnorm(r.dir), __builtin_inff(), (uint32_t)i, w);
    }
    const unsigned long long slot = wf_append_slot(&st.count[0], valid);
    if (valid) st.rec[0][slot] = w;
}

// ---- the walk ------------------------------------------------------------------------------------------------------------
template <bool SPILL, int PLAIN>
__global__ __launch_bounds__(kBvhThreads, kWfTraceWaves) void wf_trace_kernel(const SceneView *__restrict__ svp, const WfState st,
                                                                              uint32_t level, Counters *__restrict__ ctr,
                                                                              const float4 *__restrict__ nodes, const MeshArrays ma,
                                                                              uint32_t *__restrict__ spill, uint32_t spill_entries)
{
    const SceneView &sv = *svp;
    __shared__ uint32_t lds_stack[kWfStack + 1][kBvhThreads];
    __shared__ uint32_t lds_q[2 * kMeshQueue][kBvhThreads];
    uint32_t *const ls = &lds_stack[0][0];
    uint32_t *const lq = &lds_q[0][0];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const size_t spill_stride = (size_t)gridDim.x * kBvhThreads, glane = (size_t)blockIdx.x * kBvhThreads + tid;
    const unsigned long long n_queue = st.count[0];
    const WfRec *__restrict__ recs = st.rec[0];
    unsigned long long *const head = &st.work[0];
    (void)level;

    bool busy = false, drained = false;
    unsigned long long pos = 0;
    Ray32 q;
    SphereRay sr;                                 // (unused by the PLAIN step)
    TriFilterParams tp;
    float best_up = 0.f;
    uint32_t node = kNone, sp = 0, qcnt = 0, resume = 0, resume_node = 0, nbox = 0, nleaf = 0;
    bool overflow = false;
    unsigned long long box_tests = 0, leaf_filters = 0;
    sr.px = sr.py = sr.pz = sr.dx = sr.dy = sr.dz = sr.Kg = sr.K = 0.f; sr.c0 = __builtin_inff();
    q.ix = q.iy = q.iz = q.nx = q.ny = q.nz = 0.f;
    tri_filter_idle(tp);

    for (;;) {
        // ---- idle lanes take the next rays of the queue: one atomic per refill, consecutive records -> coalesced loads
        const unsigned long long idle = __ballot(!busy);
        if (idle != 0ull && !drained && ((uint32_t)__popcll(idle) >= 8u || idle == ~0ull)) {
            const unsigned long long slot = wf_append_slot(head, !busy);
            if (__ballot(!busy && slot < n_queue) == 0ull) drained = true;       // the head ran past the queue: nothing left
            if (!busy && slot < n_queue) {
                const WfRec w = recs[slot];
                pos = slot;
                q.ix = w.ix; q.iy = w.iy; q.iz = w.iz; q.nx = w.nx; q.ny = w.ny; q.nz = w.nz;
                tp.dx = w.dx; tp.dy = w.dy; tp.dz = w.dz; tp.npx = -w.px; tp.npy = -w.py; tp.npz = -w.pz; tp.A = w.A; tp.pad = 0.f;
                best_up = w.best_up;
                qcnt = 0; sp = 0; resume = 0; overflow = false;
                busy = true;
                if (best_up == best_up) node = sv.bvh_root;
                else {                             // no f32 walk for this origin: the shade kernel takes it
                    node = kNone;
                    WfCand c;
                    c.count = kWfFallback;
#pragma unroll
                    for (int e = 0; e < 7; ++e) c.e[e] = 0u;
                    st.cand[pos] = c;
                    busy = false;
                }
            }
        }
        if (__ballot(busy) == 0ull) {
            if (drained) break;
            continue;
        }
        if (busy) {
            float4 nd[MeshNode<PLAIN>::n];
            if (!mesh_step<SPILL, PLAIN, kWfStack>(nodes, ma, q, sr, tp, nd, node, sp, qcnt, overflow, best_up, resume, resume_node, ls, lq,
                                                  tid, spill, spill_entries, spill_stride, glane, nbox, nleaf)) {
                overflow = true;                   // the queue cannot take the next leaf: the shade kernel walks this ray itself
                node = kNone;
            }
            if (node == kNone) {
                WfCand c;
                uint32_t k = 0;
#pragma unroll
                for (int e = 0; e < 7; ++e) c.e[e] = 0u;
#pragma unroll
                for (int e = 0; e < kMeshQueue; ++e) {
                    if ((uint32_t)e < qcnt && __uint_as_float(lq[(size_t)(kMeshQueue + e) * kBvhThreads + tid]) <= best_up) {
                        const uint32_t v = lq[(size_t)e * kBvhThreads + tid];
                        c.e[0] = k == 0u ? v : c.e[0]; c.e[1] = k == 1u ? v : c.e[1]; c.e[2] = k == 2u ? v : c.e[2];
                        c.e[3] = k == 3u ? v : c.e[3]; c.e[4] = k == 4u ? v : c.e[4]; c.e[5] = k == 5u ? v : c.e[5];
                        k += 1;
                    }
                }
                c.count = overflow ? kWfFallback : k;
#ifdef RTX_WF_DIAG
                if (overflow) atomicAdd(&ctr[0].pad_, 1ull);
#endif
                st.cand[pos] = c;
                box_tests += nbox; leaf_filters += nleaf;
                nbox = 0; nleaf = 0;
                busy = false;
            }
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

// ---- closest_object's exact part, ray_hit, the next segment's set-up ---------------------------------------------------------
template <bool SPILL>
__global__ __launch_bounds__(kBvhThreads, 4) void wf_shade_kernel(const SceneView *__restrict__ svp, const RowsView *__restrict__ rvp,
                                                                  const WfState st, uint32_t level, double *__restrict__ samples,
                                                                  Counters *__restrict__ ctr, const float4 *__restrict__ nodes,
                                                                  const LeafArrays la, uint32_t *__restrict__ spill,
                                                                  uint32_t spill_entries)
{
    const SceneView &sv = *svp;
    const RowsView &rv = *rvp;
    __shared__ uint32_t lds_stack[kBvh4StackEntries + 1][kBvhThreads];       // the fallback walk's (round 1's bvh_traverse)
    __shared__ uint32_t lds_q[kBvhQueue][kBvhThreads];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const size_t spill_stride = (size_t)gridDim.x * kBvhThreads, glane = (size_t)blockIdx.x * kBvhThreads + tid;
    const uint32_t bounce_limit = sv.max_bounces >= 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)sv.max_bounces + 1u;
    const unsigned long long n_queue = st.count[0];
    const WfRec *__restrict__ recs = st.rec[0];
    WfRec *__restrict__ recs_out = st.rec[1];
    unsigned long long segs = 0, exact = 0, box_tests = 0, leaf_filters = 0;

    // grid-stride over the level's queue (the grid is sized for the launch's ray count; deep levels are short)
    for (unsigned long long p = (unsigned long long)blockIdx.x * kBvhThreads + tid; __ballot(p < n_queue) != 0ull;
         p += (unsigned long long)gridDim.x * kBvhThreads) {
        const bool have = p < n_queue;
        bool next = false;
        WfRec w;
        if (have) {
            const uint32_t ridx = recs[p].ridx;
            const float rec_best = recs[p].best_up;
            RayState r;
            r.pos = mk(st.pos[0][ridx], st.pos[1][ridx], st.pos[2][ridx]);
            r.dir = mk(st.dir[0][ridx], st.dir[1][ridx], st.dir[2][ridx]);
            const RayX rx = make_rayx(r.pos, r.dir);
            Hit h;
            hit_init(h);
            ++segs;
            const uint32_t left = st.left[ridx];
            const double ht = st.hit_t[ridx];
            if (ht != 0.0) { h.t = ht; h.id = la.tris[left].id; h.kind = 2; h.local = left; }     // the pre-tested self-hit
            const WfCand c = st.cand[p];
            bool covered = true;
            if (c.count & kWfFallback) {
                covered = false;
                const float omax = fmaxf(fmaxf(__builtin_fabsf((float)r.pos.x), __builtin_fabsf((float)r.pos.y)),
                                         __builtin_fabsf((float)r.pos.z));
                const bool in32 = rec_best == rec_best, in64 = omax <= sv.bvh_origin_limit * kBvhRange64;
                if (in32 || in64) {                // round 1's walk: exact tests interleaved, f32 or f64 slab test
                    FilterParams fpar;
                    TriFilterParams tpar;
                    filter_idle(fpar);
                    tri_filter_from_ray(sv, r.pos, r.dir, tpar);
                    bool ovf = false;
                    unsigned long long unused_steps = 0;
                    if (in32) {
                        Ray32 q;
                        make_ray32(r.pos, rx.dirn, (double)sv.bvh_inv_max, q);
                        bvh_traverse<true, SPILL>(nodes, la, q, fpar, tpar, rx, sv.bvh_root, ovf, h, &lds_stack[0][0], &lds_q[0][0], tid,
                                                  spill, spill_entries, spill_stride, glane, box_tests, leaf_filters, exact, unused_steps);
                    } else {
                        Ray64 q;
                        make_ray64(r.pos, rx.dirn, (double)sv.bvh_inv_max, q);
                        bvh_traverse<true, SPILL>(nodes, la, q, fpar, tpar, rx, sv.bvh_root, ovf, h, &lds_stack[0][0], &lds_q[0][0], tid,
                                                  spill, spill_entries, spill_stride, glane, box_tests, leaf_filters, exact, unused_steps);
                    }
                    covered = !ovf;
                }
            } else {
                const uint32_t n = c.count < 7u ? c.count : 7u;
#pragma unroll 1
                for (uint32_t e = 0; e < n; ++e) {
                    const uint32_t idx = e == 0 ? c.e[0] : e == 1 ? c.e[1] : e == 2 ? c.e[2] : e == 3 ? c.e[3] : e == 4 ? c.e[4] : e == 5 ? c.e[5] : c.e[6];
                    const uint32_t tk = la.tri_fidx[idx & ~kQueueTri];
                    double t;
                    if (triangle_distance(la.tris[tk], rx, &t)) hit_consider(h, t, la.tris[tk].id, 2, tk);
                }
                exact += n;
            }
            // the shapes outside the tree: every sphere (a pure mesh tree holds none), the planes, the triangles past n_tri_tree
            for (uint32_t k = 0; k < sv.n_spheres; ++k) {
                double t;
                if (sphere_distance(la.spheres[k], rx, &t)) hit_consider(h, t, la.sphere_ids[k], 0, k);
            }
            for (uint32_t k = 0; k < sv.n_planes; ++k) {
                double t;
                if (plane_distance(sv.planes[k], rx, &t)) hit_consider(h, t, sv.planes[k].id, 1, k);
            }
            const uint32_t tri_sweep_from = covered ? sv.n_tri_tree : 0u;
            for (uint32_t k = tri_sweep_from; k < sv.n_tri_filter; ++k) {
                const uint32_t tk = la.tri_fidx[k];
                double t;
                if (triangle_distance(la.tris[tk], rx, &t)) hit_consider(h, t, la.tris[tk].id, 2, tk);
            }
            exact += sv.n_spheres + sv.n_planes + (sv.n_tri_filter - tri_sweep_from);

            // ---- render_ray's match arm + ray_hit (scene.rs:232-239, 260-278)
            r.result = mk(st.res[0][ridx], st.res[1][ridx], st.res[2][ridx]);
            bool done = true;
            if (h.id != kNone) {
                r.light = mk(st.lig[0][ridx], st.lig[1][ridx], st.lig[2][ridx]);
                uint32_t pl = 0, smp = 0;
                if (rv.tiles_x != 0u) (void)ray_index_to_pixel_tiled(rv, ridx, pl, smp);
                else ray_index_to_pixel(rv, ridx, pl, smp);
                const uint32_t k = pl / rv.width, x = pl - k * rv.width;
                const uint64_t pix = (uint64_t)(rv.row_begin + k * rv.row_stride) * rv.width + x;
                r.key = rng_key(sv.seed, pix, rv.sample_begin + smp);
                r.draw = 6u + 2u * level;
                r.bounce = level;
                advance_and_shade(sv, h, r);
                done = (r.bounce >= bounce_limit) || light_is_zero(r);            // scene.rs:227-228
            }
            if (done) {
                store_sample(samples, rv, ridx, r.result);
            } else {
                st.pos[0][ridx] = r.pos.x; st.pos[1][ridx] = r.pos.y; st.pos[2][ridx] = r.pos.z;
                st.dir[0][ridx] = r.dir.x; st.dir[1][ridx] = r.dir.y; st.dir[2][ridx] = r.dir.z;
                st.res[0][ridx] = r.result.x; st.res[1][ridx] = r.result.y; st.res[2][ridx] = r.result.z;
                st.lig[0][ridx] = r.light.x; st.lig[1][ridx] = r.light.y; st.lig[2][ridx] = r.light.z;
                // the next segment: the reference's self-hit is tested here, exactly (rtx_bvh_mesh.hip), and bounds the walk
                const RayX rn = make_rayx(r.pos, r.dir);
                const uint32_t lt = h.kind == 2u ? h.local : kNone;
                double t0 = 0.0;
                float bu = __builtin_inff();
                if (lt != kNone) {
                    double t;
                    if (triangle_distance(la.tris[lt], rn, &t) && is_normal_positive(t)) { t0 = t; bu = round_up32(t); }
                    exact += 1;
                }
                st.left[ridx] = lt;
                st.hit_t[ridx] = t0;
                wf_make_rec(sv, r.pos, r.dir, rn.dirn, bu, ridx, w);
                next = true;
            }
        }
        const unsigned long long slot = wf_append_slot(&st.count[1], next);
        if (next) recs_out[slot] = w;
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
// The levels are enqueued without host round trips: every kernel reads its queue length from the device.  A level whose
// queue is empty costs two near-empty launches, so paths that may run for more than kWfLevelsPerSync levels are checked
// from the host every that many levels (max_bounces is 10 by default: one chunk).
constexpr uint32_t kWfLevelsPerSync = 16;

// Per ray of a launch: 12 f64 of state + the self-hit's distance + the triangle it left + two 64-byte records + 32 bytes of
// candidates; per level two u64 counters.
size_t wavefront_state_bytes(uint64_t n_rays, uint32_t levels)
{
    (void)levels;
    return (size_t)n_rays * (13 * sizeof(double) + sizeof(uint32_t) + 2 * sizeof(WfRec) + sizeof(WfCand)) +
           (size_t)(2 * kWfLevelsPerSync + 8) * sizeof(unsigned long long) + 32 * 256;      // (every array starts on a 256-byte boundary)
}

uint32_t wavefront_levels(const SceneView &sv)
{
    return sv.max_bounces >= 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)sv.max_bounces + 1u;
}

uint32_t wavefront_spill_entries(const SceneView &sv)
{
    const uint32_t need = 3u * sv.bvh_depth + 2u;
    return need > (uint32_t)kWfStack ? need - (uint32_t)kWfStack : 0u;
}

size_t wavefront_spill_bytes(const SceneView &sv, int n_cus)
{
    // one column per resident lane of whichever kernel walks: the walk kernel at kWfTraceWaves, the shade kernel's fallback at 4
    return (size_t)wavefront_spill_entries(sv) * (size_t)n_cus * kWfTraceWaves * kBvhThreads * sizeof(uint32_t);
}

hipError_t launch_trace_wavefront(const SceneView *d_sv, const SceneView &sv, const RowsView *d_rv, const RowsView &rv,
                                  double *samples, void *state_mem, Counters *counters, uint32_t *spill, int n_cus, hipStream_t stream)
{
    if (rv.n_rays == 0) return hipSuccess;
    const uint32_t levels = wavefront_levels(sv);
    const uint64_t n = rv.n_rays;
    // carve the state block
    char *p = static_cast<char *>(state_mem);
    WfState st;
    auto take = [&](size_t bytes) { char *q = p; p += (bytes + 255) & ~(size_t)255; return q; };
    for (int k = 0; k < 3; ++k) st.pos[k] = reinterpret_cast<double *>(take(n * sizeof(double)));
    for (int k = 0; k < 3; ++k) st.dir[k] = reinterpret_cast<double *>(take(n * sizeof(double)));
    for (int k = 0; k < 3; ++k) st.res[k] = reinterpret_cast<double *>(take(n * sizeof(double)));
    for (int k = 0; k < 3; ++k) st.lig[k] = reinterpret_cast<double *>(take(n * sizeof(double)));
    st.hit_t = reinterpret_cast<double *>(take(n * sizeof(double)));
    st.left = reinterpret_cast<uint32_t *>(take(n * sizeof(uint32_t)));
    st.rec[0] = reinterpret_cast<WfRec *>(take(n * sizeof(WfRec)));
    st.rec[1] = reinterpret_cast<WfRec *>(take(n * sizeof(WfRec)));
    st.cand = reinterpret_cast<WfCand *>(take(n * sizeof(WfCand)));
    // counters for one chunk of levels at a time (re-zeroed per chunk; the carried-over queue length is copied to slot 0)
    st.count = reinterpret_cast<unsigned long long *>(take((kWfLevelsPerSync + 2) * sizeof(unsigned long long)));
    st.work = reinterpret_cast<unsigned long long *>(take((kWfLevelsPerSync + 2) * sizeof(unsigned long long)));
    st.n = n;

    LeafArrays la;
    la.sphere_f32 = sv.bvh_leaf_f32; la.sphere_prims = sv.bvh_prims; la.spheres = sv.spheres; la.sphere_ids = sv.sphere_id;
    la.tri_f32 = sv.tri_f32; la.tri_fidx = sv.tri_fidx; la.tris = sv.tris;
    MeshArrays ma;
    ma.sphere_cr = sv.bvh_leaf_cr; ma.sphere_prims = sv.bvh_prims; ma.tri_f32 = sv.tri_f32; ma.tri_geo = sv.tri_geo;
    const uint32_t spill_entries = spill ? wavefront_spill_entries(sv) : 0u;
    const bool deep = spill_entries != 0u;
    const float4 *nodes = reinterpret_cast<const float4 *>(sv.bvh_nodes);
    const float4 *qnodes = reinterpret_cast<const float4 *>(sv.bvh_qnodes);
    const bool qn = (sv.bvh_flags & 8u) != 0u && !std::getenv("RTX_HIP_NO_QNODES");
    const uint32_t trace_blocks = (uint32_t)std::min<uint64_t>((n + kBvhThreads - 1) / kBvhThreads, (uint64_t)n_cus * kWfTraceWaves);
    // (grid-stride; 4 workgroups per CU are resident, and the fallback walk's HBM stack column is indexed by the resident lane)
    const uint32_t shade_blocks = (uint32_t)std::min<uint64_t>((n + kBvhThreads - 1) / kBvhThreads, (uint64_t)n_cus * 4u);
    const size_t counter_bytes = (kWfLevelsPerSync + 2) * sizeof(unsigned long long);

    hipError_t e = hipMemsetAsync(st.count, 0, counter_bytes, stream);
    if (e == hipSuccess) e = hipMemsetAsync(st.work, 0, counter_bytes, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(wf_generate_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, d_sv, d_rv, st);
    if ((e = hipGetLastError()) != hipSuccess) return e;

    uint32_t level = 0;                           // the path's level: draw / bounce indices
    while (level < levels) {
        const uint32_t chunk = levels - level < kWfLevelsPerSync ? levels - level : kWfLevelsPerSync;
        for (uint32_t k = 0; k < chunk; ++k) {
            // counters are indexed by the level's position in the chunk (k); the records by the parity of k as well
            WfState sk = st;
            sk.count = st.count + k; sk.work = st.work + k;
            sk.rec[0] = st.rec[k & 1u]; sk.rec[1] = st.rec[(k + 1u) & 1u];
            if (qn) {
                if (deep) hipLaunchKernelGGL((wf_trace_kernel<true, 2>), dim3(trace_blocks), dim3(kBvhThreads), 0, stream, d_sv, sk, level + k, counters, qnodes, ma, spill, spill_entries);
                else hipLaunchKernelGGL((wf_trace_kernel<false, 2>), dim3(trace_blocks), dim3(kBvhThreads), 0, stream, d_sv, sk, level + k, counters, qnodes, ma, spill, spill_entries);
            } else {
                if (deep) hipLaunchKernelGGL((wf_trace_kernel<true, 1>), dim3(trace_blocks), dim3(kBvhThreads), 0, stream, d_sv, sk, level + k, counters, nodes, ma, spill, spill_entries);
                else hipLaunchKernelGGL((wf_trace_kernel<false, 1>), dim3(trace_blocks), dim3(kBvhThreads), 0, stream, d_sv, sk, level + k, counters, nodes, ma, spill, spill_entries);
            }
            if ((e = hipGetLastError()) != hipSuccess) return e;
            if (deep) hipLaunchKernelGGL(wf_shade_kernel<true>, dim3(shade_blocks), dim3(kBvhThreads), 0, stream, d_sv, d_rv, sk, level + k, samples, counters, nodes, la, spill, spill_entries);
            else hipLaunchKernelGGL(wf_shade_kernel<false>, dim3(shade_blocks), dim3(kBvhThreads), 0, stream, d_sv, d_rv, sk, level + k, samples, counters, nodes, la, spill, spill_entries);
            if ((e = hipGetLastError()) != hipSuccess) return e;
        }
        level += chunk;
        if (level >= levels) break;
        // more levels allowed than one chunk: stop when the queue ran empty, else carry the queue length over
        unsigned long long left = 0;
        if ((e = hipMemcpyAsync(&left, st.count + chunk, sizeof left, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
        if (left == 0) break;
        if ((e = hipMemsetAsync(st.count, 0, counter_bytes, stream)) != hipSuccess) return e;
        if ((e = hipMemsetAsync(st.work, 0, counter_bytes, stream)) != hipSuccess) return e;
        if ((e = hipMemcpyAsync(st.count, &left, sizeof left, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;      // (`left` is a stack variable)
        if (chunk & 1u) std::swap(st.rec[0], st.rec[1]);                      // the next chunk's level 0 reads what this chunk's last level wrote
    }
    return hipSuccess;
}

}  // namespace rtx
