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
// A walk whose 6-entry candidate queue runs full moves its live entries to the level's overflow list (wf_flush_to_extra) and
// goes on; a far origin walks with Ray32S's slack.  What no f32 walk can finish -- an origin beyond 2^27 x origin_limit or
// NaN, a full overflow list -- is flagged kWfFallback and the shade kernel tests every shape exactly for that segment.
#include "rtx_launch.h"
#include "rtx_mesh_step.h"
#include "rtx_wavefront.h"

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdlib>

namespace rtx {

#ifndef RTX_WF_SERVICE
#define RTX_WF_SERVICE 8
#endif
[[maybe_unused]] constexpr uint32_t kWfService = RTX_WF_SERVICE;                   // lanes of a wave that wait before they are served together
#ifndef RTX_WF_TRACE_WAVES
#define RTX_WF_TRACE_WAVES 4
#endif
constexpr int kWfTraceWaves = RTX_WF_TRACE_WAVES;
#ifndef RTX_WF_SHADE_WAVES
#define RTX_WF_SHADE_WAVES 4
#endif
constexpr int kWfShadeWaves = RTX_WF_SHADE_WAVES;                 // workgroups per CU of the shade kernel                                  // workgroups per CU of the walk kernel
constexpr int kWfStack = 160 / kWfTraceWaves - 1 - 2 * kMeshQueue;   // 13 LDS stack entries per lane, the HBM column behind them

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
    // an origin outside origin_limit (a bounce off one of the reference's far phantom hits) walks with Ray32S's slack; beyond
    // 2^27 times that, or NaN: no walk (the shade kernel tests every shape)
    const float omax = fmaxf(fmaxf(__builtin_fabsf((float)pos.x), __builtin_fabsf((float)pos.y)), __builtin_fabsf((float)pos.z));
    const bool in32 = omax <= sv.bvh_origin_limit;
    w.slack = ray32_slack(q.nx, q.ny, q.nz, in32);
    w.best_up = (in32 || omax <= sv.bvh_origin_limit * kBvhRange64) ? best_up : __builtin_nanf("");
    w.ridx = ridx;
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
    // level 0's queue is the launch's ray queue itself (slot i = ray i, no atomics); the padding of partial tiles is marked dead
    if (i == 0) st.count[0] = rv.n_rays;
    if (i >= rv.n_rays) return;
    WfRec w;
    if (valid) {
        RayState r;
        gen_primary(sv, rv, pl, rv.sample_begin + smp, r);
        const WfRays &rs = st.ray[0];                        // (result = 0, light = 1, no self-hit are implied at level 0)
        rs.pos[0][i] = r.pos.x; rs.pos[1][i] = r.pos.y; rs.pos[2][i] = r.pos.z;
        rs.dir[0][i] = r.dir.x; rs.dir[1][i] = r.dir.y; rs.dir[2][i] = r.dir.z;
        wf_make_rec(sv, r.pos, r.dir, vnorm(r.dir), __builtin_inff(), (uint32_t)i, w);
    } else {
        w.px = w.py = w.pz = w.dx = w.dy = w.dz = w.ix = w.iy = w.iz = w.nx = w.ny = w.nz = w.best_up = w.A = 0.f;
        w.ridx = kNone; w.slack = 0.f;
    }
    st.rec[0][i] = w;
}

#ifdef RTX_LAB      // the per-lane walk kernel of the all-levels wavefront form (RTX_TUNE_WF_PURE, RTX_TUNE_NO_PACKETS): librtx_hip_lab.so only
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

    const unsigned long long grab = wf_grab_size(n_queue);
    WfChunk ch;
    ch.next = ch.end = 0; ch.drained = false;
    bool walking = false, have = false;           // have: the lane holds a segment (walking, or complete and not yet written)
    uint32_t pos = 0, ridx = 0;
    Ray32S q;
    SphereRay sr;                                 // (unused by the PLAIN step)
    TriFilterParams tp;
    float best_up = 0.f;
    uint32_t node = kNone, sp = 0, qcnt = 0, resume = 0, resume_node = 0, nbox = 0, nleaf = 0;
    bool overflow = false, extra = false;
    unsigned long long box_tests = 0, leaf_filters = 0;
    sr.px = sr.py = sr.pz = sr.dx = sr.dy = sr.dz = sr.Kg = sr.K = 0.f; sr.c0 = __builtin_inff();
    q.ix = q.iy = q.iz = q.nx = q.ny = q.nz = q.e = 0.f;
    tri_filter_idle(tp);

    for (;;) {
        // ---- lanes whose walk is complete wait until kWfService of the wave do; then they write their candidates and
        //      take the next records of the wave's chunk of the queue (consecutive records -> coalesced loads)
        const unsigned long long wmask = __ballot(walking);
        const uint32_t n_fin = (uint32_t)__popcll(__ballot(!walking && have));
        const bool serve = wmask == 0ull || (ch.drained ? n_fin >= kWfService : 64u - (uint32_t)__popcll(wmask) >= kWfService);
        if (serve) {
            if (!walking && have) {
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
                c.e[6] = ridx;
                c.count = ridx == kNone ? kWfDead : (overflow ? kWfFallback : (k | (extra ? kWfExtra : 0u)));
                st.cand[pos] = c;
                box_tests += nbox; leaf_filters += nleaf;
                nbox = 0; nleaf = 0;
                have = false;
            }
            if (!ch.drained) {
                unsigned long long my;
                if (wf_take(ch, head, grab, n_queue, !have, my)) {
                    const WfRec w = recs[my];
                    pos = (uint32_t)my;                          // (the host keeps a launch below 2^32 rays)
                    ridx = w.ridx;
                    q.ix = w.ix; q.iy = w.iy; q.iz = w.iz; q.nx = w.nx; q.ny = w.ny; q.nz = w.nz; q.e = w.slack;
                    tp.dx = w.dx; tp.dy = w.dy; tp.dz = w.dz; tp.npx = -w.px; tp.npy = -w.py; tp.npz = -w.pz; tp.A = w.A; tp.pad = 0.f;
                    best_up = w.best_up;
                    qcnt = 0; sp = 0; resume = 0; overflow = false; extra = false;
                    have = true;
                    if (w.ridx == kNone) node = kNone;                               // a dead slot: passed on as such
                    else if (best_up == best_up) { node = sv.bvh_root; walking = true; }
                    else { node = kNone; overflow = true; }      // no f32 walk for this origin: the shade kernel takes it
                }
            }
            if (__ballot(walking) == 0ull) {
                if (ch.drained && __ballot(have) == 0ull) break;
                continue;
            }
        }
        if (walking) {
            float4 nd[MeshNode<PLAIN>::n];
            if (!mesh_step<SPILL, PLAIN, kWfStack>(nodes, ma, q, sr, tp, nd, node, sp, qcnt, overflow, best_up, resume, resume_node, ls, lq,
                                                  tid, spill, spill_entries, spill_stride, glane, nbox, nleaf)) {
                // the queue cannot take the next leaf: its live entries move to the level's overflow list and the step
                // resumes with the leaf children it had not read (resume != 0); only a full list ends the walk here
                if (wf_flush_to_extra(st, pos, lq, kMeshQueue, tid, qcnt, best_up)) extra = true;
                else { overflow = true; node = kNone; resume = 0; }
            }
            walking = node != kNone || resume != 0u;
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

#endif  // RTX_LAB

// ---- the walk of level 0 as packets --------------------------------------------------------------------------------------
// Level 0's queue is the launch's ray queue: 64 consecutive records are the primary rays of one 8x8 pixel tile of one
// sample -- one origin (up to the aperture jitter), directions a few pixels apart.  They cross the same footprints, so
// the wave walks the tree ONCE for all of them: one wave-uniform stack, the node and the leaf records read at a
// wave-uniform address (one request instead of 64 address-divergent ones -- the gather rate is what bounds the per-lane
// walk), every lane testing its own ray against them with its own bound.  A child is opened when any
// lane's ray enters it; the order is the first entering lane's.  What a lane collects is what its own walk would have
// collected for SOME visiting order -- every candidate that can still be the winner (t_lo <= its best_up) -- and the
// exact tests decide as always.  Pure (x, y)-footprint trees; the 96-byte footprint nodes (at a uniform address the two
// extra requests are free and the rectangles need no decoding).
// ------------------------------------------------------------------------------------------------------------------------------
// Tile lists for level 0 of a pure (x, y)-footprint tree (the sphere packets' idea, rtx_bvh_spheres.hip, with the test a mesh needs).
// Every ray of an 8x8 tile starts in O = cam_pos + [0, non_focal_offset]^3 and goes through T = hull of the tile's focal points +
// [0, focal_offset]^3.  Which filter records can such a ray pass?  NOT "the triangles the beam touches": Triangle::distance places its
// hit point at |n.(v0 - p) / n.d| along the ray (triangle.rs:108-127), so a triangle behind the origin can report a phantom hit in
// front, and the footprint filter (tri_filter_sign, rtx_device.h) is the only geometry there is:
//     e_x = (p_x - c_x) |n.d| + d_x |n.(v0 - p)|,    candidate <=> |e_x| <= h_x |n.d| + A     (and the same in y).
// One thread per tile walks the footprint tree with the beam's (x, y) slabs (an interval origin and an interval direction: linear
// inequalities in the ray parameter, as build_tile_lists_kernel) and evaluates those two inequalities for each record it reaches in
// INTERVAL arithmetic over p in O and the unit directions d of the beam: a record some ray of the tile can pass has min |e_x| <=
// h_x max |n.d| + 2 A (twice the filter's own error allowance: whatever the f32 filter passes for a ray of the tile is in the list).
// C3: ~1500 records lie under a tile's strip, a few dozen survive.  Entry: {record index, a lower bound of |t| over the beam =
// min |n.(v0 - p)| / max |n.d|}, sorted by that bound; the packet kernel runs its leaf loop over the list and stops when the bound
// passes every lane's certain hit.  A tile whose list overflows kMeshTileCap, whose beam degenerates or whose walk exceeds the stack
// keeps the packet walk.
constexpr uint32_t kMeshTileCap = 512;
constexpr uint32_t kMeshTileWalk = 0xFFFFFFFFu;
constexpr uint32_t kMeshTileSphere = 0x80000000u;          // entry: an index into the sphere leaf arrays (a joint tree), else a filter record
struct MeshTileLists {
    uint32_t *count;                 // [tiles]; null: no lists
    uint2 *entries;                  // [tiles][kMeshTileCap]: {filter record, bits of t_lb}
    uint32_t tiles_per_sample;
};

struct Iv { double lo, hi; };
__device__ __forceinline__ Iv iv_scale(double k, Iv a) { return k >= 0.0 ? Iv{k * a.lo, k * a.hi} : Iv{k * a.hi, k * a.lo}; }
__device__ __forceinline__ Iv iv_add(Iv a, Iv b) { return Iv{a.lo + b.lo, a.hi + b.hi}; }
__device__ __forceinline__ Iv iv_abs(Iv a) { return (a.lo <= 0.0 && a.hi >= 0.0) ? Iv{0.0, fmax(-a.lo, a.hi)} : (a.lo > 0.0 ? a : Iv{-a.hi, -a.lo}); }
__device__ __forceinline__ Iv iv_mul(Iv a, Iv b)
{
    const double p0 = a.lo * b.lo, p1 = a.lo * b.hi, p2 = a.hi * b.lo, p3 = a.hi * b.hi;
    return Iv{fmin(fmin(p0, p1), fmin(p2, p3)), fmax(fmax(p0, p1), fmax(p2, p3))};
}
__device__ __forceinline__ Iv iv_widen(Iv a) { const double w = (fabs(a.lo) + fabs(a.hi)) * 1e-12 + 1e-300; return Iv{a.lo - w, a.hi + w}; }

#ifndef RTX_TILE_BUILD_WAVES
#define RTX_TILE_BUILD_WAVES 3
#endif
constexpr uint32_t kMeshTileStack = 512;                     // the builder's node stack per wave (LDS)
constexpr uint32_t kMeshTileBudget = 32768;                  // records + node children a tile's build may look at before it gives up (the tile
                                                             // then walks): a wide beam in a dense mesh would test everything under it only to
                                                             // overflow at the end -- C5's tiles at 4K look at ~12 000

__device__ __forceinline__ double wave_min_f64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ double wave_max_f64(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

// One WAVE per tile: lane j computes pixel j's focal point; the walk takes up to 16 nodes off the wave's stack per step, lane l
// tests child l & 3 of node l >> 2 against the beam's (x, y) slabs; a lane whose child is a leaf the beam can enter tests that
// leaf's records itself.  The filter's inequalities are evaluated at the 8 corners of the direction box D = T - O (for a fixed
// origin and fixed signs of n.D and n.(v0 - p) they are affine in each component of D, so their extremes over the box sit at its
// corners) with the origin as an interval: the dependency between a hit point and the direction that leads to it survives, which
// plain interval arithmetic over D loses (a record 100 units away would be accepted +- 2 units around its true footprint).
__global__ __launch_bounds__(256, RTX_TILE_BUILD_WAVES) void build_mesh_tile_lists_kernel(const SceneView *__restrict__ svp, const RowsView *__restrict__ rvp,
                                                                    const float4 *__restrict__ nodes, const float4 *__restrict__ tri_f32,
                                                                    MeshTileLists tl, uint32_t n_tiles, uint32_t root, uint32_t cap,
                                                                    TileEntry *__restrict__ sph_out, const float4 *__restrict__ sphere_f32,
                                                                    const uint32_t *__restrict__ sphere_prims)
{
    // cap <= kMeshTileCap: entries a tile's list may hold.  sph_out != null (a sphere tree: every entry is a sphere): the list is
    // written in the sphere packets' format -- TileEntry {record, index, bound} -- instead of {entry, bound} pairs
    __shared__ uint32_t s_stack[4][kMeshTileStack];
    __shared__ uint32_t s_idx[4][kMeshTileCap];
    __shared__ float s_tlb[4][kMeshTileCap];
    __shared__ uint32_t s_cnt[4];
    __shared__ uint32_t s_lstart[4][64], s_lfirst[4][64], s_lsph[4][64];
    const SceneView &sv = *svp;
    const RowsView &rv = *rvp;
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t tile = blockIdx.x * 4u + wv;
    if (tile >= n_tiles) return;                            // (wave-uniform)
    const double nfo = sv.non_focal_offset, fo = sv.focal_offset;
    const double cam[3] = { sv.cam_pos.x, sv.cam_pos.y, sv.cam_pos.z };
    double omin[3], omax[3], tmin[3], tmax[3];
    uint32_t pl, smp;
    const bool have = ray_index_to_pixel_tiled(rv, (uint64_t)tile * 64u + lane, pl, smp);
    V3 f = mk(0., 0., 0.);
    if (have) f = primary_focal_point(sv, rv, pl);
    const double fp[3] = { f.x, f.y, f.z };
    for (int a = 0; a < 3; ++a) {
        omin[a] = cam[a] + fmin(0.0, nfo); omax[a] = cam[a] + fmax(0.0, nfo);
        tmin[a] = wave_min_f64(have ? fp[a] + fmin(0.0, fo) : __builtin_inf());
        tmax[a] = wave_max_f64(have ? fp[a] + fmax(0.0, fo) : -__builtin_inf());
    }
    if (__ballot(have) == 0ull) { if (lane == 0) tl.count[tile] = 0u; return; }
    double dlo[3], dhi[3], gap2 = 0.0, far2 = 0.0;
    bool ok = true;
    for (int a = 0; a < 3; ++a) {
        const double lo = tmin[a] - omax[a], hi = tmax[a] - omin[a];
        const double w = (fabs(lo) + fabs(hi)) * 1e-12 + 1e-300;
        dlo[a] = lo - w; dhi[a] = hi + w;
        const double g = fmax(0.0, fmax(tmin[a] - omax[a], omin[a] - tmax[a]));
        gap2 += g * g;
        const double m = fmax(fabs(dlo[a]), fabs(dhi[a]));
        far2 += m * m;
        ok = ok && isfinite(dlo[a]) && isfinite(dhi[a]) && isfinite(omin[a]) && isfinite(omax[a]);
    }
    const double lmin = sqrt(gap2) * (1.0 - 1e-9), lmax = sqrt(far2) * (1.0 + 1e-9);
    ok = ok && lmin > 0.0 && isfinite(lmax);
    Iv pr[3];                                              // the origin relative to the filter's centre
    double S = sv.tri_extent + 1.0;
    for (int a = 0; a < 3; ++a) {
        pr[a] = iv_widen(Iv{omin[a] - sv.sphere_center[a], omax[a] - sv.sphere_center[a]});
        S += fmax(fabs(pr[a].lo), fabs(pr[a].hi));
    }
    ok = ok && S < 1.0e14;
    // the filter's allowance A = 64 u S for a UNIT direction; the inequalities below are written in D = t - o (both sides scale with |D|)
    const double A2 = 2.0 * S * (64.0 / 16777216.0) * (1.0 + 1e-6) * lmax;

    uint32_t *const stk = &s_stack[wv][0];
    if (lane == 0) { s_cnt[wv] = 0u; stk[0] = root; }                 // (links keep their kBvhFlatNode flag on the stack)
    __builtin_amdgcn_wave_barrier();
    uint32_t sp = 1, looked = 0;
    ok = __ballot(!ok) == 0ull;
    while (ok && sp != 0u) {
        const uint32_t take = sp < 16u ? sp : 16u;
        const uint32_t slot = lane >> 2, c = lane & 3u;
        const bool active = slot < take;
        const uint32_t node = active ? stk[sp - 1u - slot] : 0u;
        sp -= take;
        __builtin_amdgcn_wave_barrier();
        bool bad = false, interior = false, leaf = false;
        uint32_t link = 0, cnt = 0xFFFFFFFFu;
        double u_enter = 0.0;                              // where the beam can enter the child's box (the sorting bound of a sphere leaf)
        if (active) {
            const bool flat = (node & kBvhFlatNode) != 0u;
            const float4 *np = nodes + 8 * (size_t)(node & ~kBvhFlatNode);
            double lo[3], hi[3];
            if (flat) {                                    // a footprint node: 4 x {lo.x, lo.y, hi.x, hi.y}, the links, the counts
                const float4 r = np[c];
                link = reinterpret_cast<const uint32_t *>(np + 4)[c];
                cnt = reinterpret_cast<const uint32_t *>(np + 5)[c];
                lo[0] = (double)r.x; lo[1] = (double)r.y; hi[0] = (double)r.z; hi[1] = (double)r.w;
                lo[2] = -__builtin_inf(); hi[2] = __builtin_inf();
            } else {                                       // a 3-D node of a joint tree: 4 x {lo.xyz, link}, 4 x {hi.xyz, count}
                const float4 a = np[c], b = np[4 + c];
                link = __float_as_uint(a.w); cnt = __float_as_uint(b.w);
                lo[0] = (double)a.x; lo[1] = (double)a.y; lo[2] = (double)a.z; hi[0] = (double)b.x; hi[1] = (double)b.y; hi[2] = (double)b.z;
            }
            if (cnt != 0xFFFFFFFFu) {
                double u0 = 0.0, u1 = __builtin_inf();
                bool miss = false;
                for (int k = 0; k < 3; ++k) {
                    if (lo[k] == -__builtin_inf() && hi[k] == __builtin_inf()) continue;       // unbounded along this axis
                    if (dlo[k] > 0.0) u1 = fmin(u1, (hi[k] - omin[k]) / dlo[k]);
                    else if (dlo[k] < 0.0) u0 = fmax(u0, (hi[k] - omin[k]) / dlo[k]);
                    else if (omin[k] > hi[k]) miss = true;
                    if (dhi[k] > 0.0) u0 = fmax(u0, (lo[k] - omax[k]) / dhi[k]);
                    else if (dhi[k] < 0.0) u1 = fmin(u1, (lo[k] - omax[k]) / dhi[k]);
                    else if (omax[k] < lo[k]) miss = true;
                }
                if (!(u0 == u0) || !(u1 == u1)) bad = true;
                else if (!(miss || u0 * (1.0 - 1e-9) > u1 * (1.0 + 1e-9) + 1e-300)) {
                    interior = cnt == 0u;
                    leaf = cnt != 0u;
                    u_enter = u0;
                }
            }
        }
        // the interior children the beam enters go back to the stack
        const unsigned long long im = __ballot(interior);
        const uint32_t pos = sp + (uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)(im >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)im, 0u));
        if (interior) { if (pos < kMeshTileStack) stk[pos] = link; else bad = true; }
        sp += (uint32_t)__builtin_popcountll(im);
        // the records of the leaves the beam enters, flattened across the lanes (a leaf holds up to 6: one lane per RECORD, not per
        // leaf -- the loads of a step's ~100 records are in flight together instead of five deep)
        {
            const bool lf = leaf && !bad;
            const uint32_t m = lf ? (cnt & 0xFFFFu) : 0u;
            uint32_t incl = m;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = (uint32_t)__shfl_up((int)incl, off, 64);
                if ((int)lane >= off) incl += t;
            }
            const uint32_t total = (uint32_t)__shfl((int)incl, 63, 64);
            looked += total + 4u * take;
            s_lstart[wv][lane] = incl - m;
            s_lfirst[wv][lane] = link;
            // a sphere leaf's entries are bounded by their box: kind flag + the bits of the bound (u_enter |t - o|min), in place of a filter test
            s_lsph[wv][lane] = (lf && (cnt & kBvhTriLeaf) == 0u) ? (0x80000000u | (__float_as_uint(fmaxf(round_down_f32_dev(u_enter * lmin * (1.0 - 1e-6) - 1e-30), 0.0f)) >> 1)) : 0u;
            __builtin_amdgcn_wave_barrier();
            for (uint32_t r = lane; r < total; r += 64u) {
                uint32_t L = 0;                                   // the last lane whose first record is <= r (binary search: starts ascend)
#pragma unroll
                for (int step = 32; step > 0; step >>= 1)
                    if (s_lstart[wv][L + step] <= r) L += step;
                const uint32_t rec = s_lfirst[wv][L] + (r - s_lstart[wv][L]);
                const uint32_t sph = s_lsph[wv][L];
                if (sph != 0u) {                                // an entry of a sphere leaf: index into the leaf arrays, kMeshTileSphere set
                    const uint32_t k = atomicAdd(&s_cnt[wv], 1u);
                    if (k < kMeshTileCap) { s_idx[wv][k] = rec | kMeshTileSphere; s_tlb[wv][k] = __uint_as_float((sph & 0x7FFFFFFFu) << 1); }
                    continue;
                }
                const float4 A = tri_f32[2 * (size_t)rec], B = tri_f32[2 * (size_t)rec + 1];
                const bool planar = B.z < 1.0e29f;             // a record with an (x, y) rectangle; the other planes' records pass every ray
                Iv nv = iv_add(iv_add(iv_scale(-(double)A.x, pr[0]), iv_scale(-(double)A.y, pr[1])), iv_scale(-(double)A.z, pr[2]));
                nv.lo += (double)A.w; nv.hi += (double)A.w;
                const Iv anv = iv_abs(iv_widen(nv));
                // (planar records only) a cheap necessary condition first (most records under the strip fail it: their plane is elsewhere in z): the hit point
                // q = p + u D has its (x, y) in the record's rectangle only for u in [u0, u1] (the slab test again, on the rectangle, in
                // the filter's coordinates and with its allowance), and |n.(v0 - p)| = u |n.D| there -- in plain interval arithmetic
                if (planar) {
                    const double pad = A2 / lmin + 1e-9 * S;                                   // (the allowance as a length, generously)
                    double u0 = 0.0, u1 = __builtin_inf();
                    bool miss = false;
                    for (int k = 0; k < 2; ++k) {
                        const double rlo = (double)(k == 0 ? B.x : B.y) - (double)(k == 0 ? B.z : B.w) - pad;
                        const double rhi = (double)(k == 0 ? B.x : B.y) + (double)(k == 0 ? B.z : B.w) + pad;
                        if (dlo[k] > 0.0) u1 = fmin(u1, (rhi - pr[k].lo) / dlo[k]);
                        else if (dlo[k] < 0.0) u0 = fmax(u0, (rhi - pr[k].lo) / dlo[k]);
                        else if (pr[k].lo > rhi) miss = true;
                        if (dhi[k] > 0.0) u0 = fmax(u0, (rlo - pr[k].hi) / dhi[k]);
                        else if (dhi[k] < 0.0) u1 = fmin(u1, (rlo - pr[k].hi) / dhi[k]);
                        else if (pr[k].hi < rlo) miss = true;
                    }
                    if (u0 == u0 && u1 == u1) {                                                // (NaN: no shortcut, the corners decide)
                        if (miss || u0 * (1.0 - 1e-9) > u1 * (1.0 + 1e-9) + 1e-300) continue;
                        const Iv nDb = iv_abs(iv_widen(iv_add(iv_add(iv_scale((double)A.x, Iv{dlo[0], dhi[0]}), iv_scale((double)A.y, Iv{dlo[1], dhi[1]})),
                                                              iv_scale((double)A.z, Iv{dlo[2], dhi[2]}))));
                        const double slack = 1e-9 * (anv.hi + nDb.hi * (u1 < 1e300 ? u1 : 0.0)) + A2 * 4.0;
                        if (anv.lo > nDb.hi * u1 + slack || anv.hi + slack < nDb.lo * u0) continue;
                    }
                }
                const Iv axv = Iv{pr[0].lo - (double)B.x, pr[0].hi - (double)B.x}, ayv = Iv{pr[1].lo - (double)B.y, pr[1].hi - (double)B.y};
                double m1x = __builtin_inf(), m2x = -__builtin_inf(), m1y = __builtin_inf(), m2y = -__builtin_inf();
                double nd_min = __builtin_inf(), nd_max = -__builtin_inf(), and_max = 0.0;
                for (int v = 0; v < 8; ++v) {
                    const double Dx = (v & 1) ? dhi[0] : dlo[0], Dy = (v & 2) ? dhi[1] : dlo[1], Dz = (v & 4) ? dhi[2] : dlo[2];
                    const double nD = (double)A.x * Dx + (double)A.y * Dy + (double)A.z * Dz;
                    const double a = fabs(nD) * (1.0 + 1e-12);
                    nd_min = fmin(nd_min, nD); nd_max = fmax(nd_max, nD); and_max = fmax(and_max, a);
                    const Iv ex = iv_add(iv_scale(a, axv), iv_scale(Dx, anv)), ey = iv_add(iv_scale(a, ayv), iv_scale(Dy, anv));
                    const double Rx = ((double)B.z * a + A2) * (1.0 + 1e-9), Ry = ((double)B.w * a + A2) * (1.0 + 1e-9);
                    m1x = fmin(m1x, ex.lo - Rx); m2x = fmax(m2x, ex.hi + Rx);
                    m1y = fmin(m1y, ey.lo - Ry); m2y = fmax(m2y, ey.hi + Ry);
                }
                if (!(m1x == m1x) || !(m2x == m2x) || !(m1y == m1y) || !(m2y == m2y)) { bad = true; continue; }
                if (and_max == 0.0) continue;             // n.D = 0 at every corner: no ray of the tile has a finite distance
                const bool mixed = nd_min < 0.0 && nd_max > 0.0;            // n.D changes sign inside the box: not affine there -- a candidate
                const bool pass = mixed || !planar || (m1x <= 0.0 && m2x >= 0.0 && m1y <= 0.0 && m2y >= 0.0);
                if (!pass) continue;
                const float t_lb = round_down_f32_dev(anv.lo * lmin / and_max * (1.0 - 1e-6));
                const uint32_t k = atomicAdd(&s_cnt[wv], 1u);
                if (k < kMeshTileCap) { s_idx[wv][k] = rec; s_tlb[wv][k] = t_lb; }
            }
            __builtin_amdgcn_wave_barrier();
        }
        __builtin_amdgcn_wave_barrier();
        ok = __ballot(bad) == 0ull && s_cnt[wv] <= cap && looked <= kMeshTileBudget;
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t n = s_cnt[wv];
    if (!ok || n > cap) { if (lane == 0) tl.count[tile] = kMeshTileWalk; return; }
    // rank sort by t_lb (ties: the order of arrival) into the tile's list
    uint2 *const out = tl.entries + (size_t)tile * kMeshTileCap;
    TileEntry *const sout = sph_out ? sph_out + (size_t)tile * cap : nullptr;
    for (uint32_t e = lane; e < n; e += 64u) {
        const float te = s_tlb[wv][e];
        uint32_t rank = 0;
        for (uint32_t g = 0; g < n; ++g) {
            const float tg = s_tlb[wv][g];
            rank += (tg < te || (tg == te && g < e)) ? 1u : 0u;
        }
        if (sout) {
            const uint32_t li = s_idx[wv][e] & ~kMeshTileSphere;
            TileEntry t;
            t.rec = sphere_f32[li]; t.prim = sphere_prims[li]; t.t_lb = te; t.pad0 = t.pad1 = 0u;
            sout[rank] = t;
        } else {
            out[rank] = make_uint2(s_idx[wv][e], __float_as_uint(te));
        }
    }
    if (lane == 0) tl.count[tile] = n;
}

constexpr int kPkStack = 128;                                      // wave-uniform stack entries (the host checks 3 * depth + 2 against it)
#ifndef RTX_PK_WAVES
#define RTX_PK_WAVES 7
#endif
constexpr int kPkWaves = RTX_PK_WAVES;                             // workgroups per CU (a joint tree's instance: kPkWavesJoint)
#ifndef RTX_PK_WAVES_JOINT
#define RTX_PK_WAVES_JOINT 7
#endif
constexpr int kPkWavesJoint = RTX_PK_WAVES_JOINT;

// (PkConst4 / pk_const / pk_bits: wave-uniform reads through the scalar cache, rtx_traverse.h)

// (rtx_writelane: rtx_traverse.h)

// The wave-uniform stack of a packet walk in the lanes of ONE VGPR (v_writelane / v_readlane with a scalar index) when the
// tree's depth allows it (3 * depth + 2 <= kPkLaneStack), else in LDS (lane 0 stores, exec-masked).  The packet kernels are
// SCALAR-pipe bound (C3: 1.06e10 SALU instructions per launch against 0.95e10 VALU, one scalar pipe per CU ~80 % busy,
// profiles/r03_mesh_packets_pmc.txt), and an LDS push costs 6 scalar instructions of exec handling where the register push
// costs 2.  (Measured and dropped here: the sphere packets' full treatment -- one loop instance per direction quadrant, the
// leaf code once behind scalar selects -- issued MORE instructions on a mesh, whose walk is dominated by leaf records
// (705 per tile against 369 node visits on C3): 39.8 against 34.2 ms; RTX_TUNE_PK_LDS_STACK keeps the LDS form for A/B.)
constexpr int kPkLaneStack = 63;
#ifdef RTX_PK_OLD_BALLOT
#define RTX_PK_BALLOT(x) __ballot(x)
#else
#define RTX_PK_BALLOT(x) __builtin_amdgcn_ballot_w64(x)          // the mask straight out of v_cmp (HIP's __ballot re-materialises the bool: 2 more VALU)
#endif

// One filter record of a triangle leaf against the wave's rays (uniform address: scalar loads): the lanes that are IN filter it,
// bound it (tri_bounds) and queue it as a candidate; used by the walk's leaf loop and by the sweep over a tile's list.
#define RTX_PK_TRI_RECORD(IDX, IN)                                                                                                  \
                {                                                                                                                   \
                    const uint32_t rec_ = (IDX);                                                                                    \
                    const PkConst4 rp = ctri + 2 * (size_t)rec_;                                                                    \
                    const float4 A = rp[0], B = rp[1];                                                                              \
                    const bool pass = (IN) && (int)tri_filter_sign(A, B, tp) >= 0;                                                  \
                    if (RTX_PK_BALLOT(pass) != 0ull) {                                                                              \
                        const PkConst4 gp = cgeo + 2 * (size_t)rec_;                                                                \
                        const float4 g0 = gp[0], g1 = gp[1];                                                                        \
                        if (pass) {                                                                                                 \
                            float thi;                                                                                              \
                            const float tlo = tri_bounds(A, g0, g1, tp, thi);                                                       \
                            if (tlo <= best_up && tlo < __builtin_inff()) {                                                         \
                                best_up = fminf(best_up, thi);                                                                      \
                                if (!mesh_queue_room(lq, tid, qcnt, best_up, 1u)) {       /* more live candidates than the queue holds: */ \
                                    if (wf_flush_to_extra(st, (uint32_t)p, lq, kMeshQueue, tid, qcnt, best_up)) extra = true;   /* to the overflow list */ \
                                    else { overflow = true; best_up = -__builtin_inff(); }     /* (full: the shade kernel tests every shape for this ray) */ \
                                }                                                                                                   \
                                if (!overflow) {                                                                                    \
                                    lq[(size_t)qcnt * kBvhThreads + tid] = rec_ | kQueueTri;                                        \
                                    lq[(size_t)(kMeshQueue + qcnt) * kBvhThreads + tid] = __float_as_uint(tlo);                     \
                                    qcnt += 1;                                                                                      \
                                }                                                                                                   \
                            }                                                                                                       \
                        }                                                                                                           \
                    }                                                                                                               \
                }

// One record of a sphere leaf of a joint tree against the wave's rays (bvh_traverse_spheres' bounds, rtx_traverse.h): walk and list sweep.
#define RTX_PK_SPH_RECORD(IDX, IN)                                                                                                  \
                {                                                                                                                   \
                    const uint32_t e_ = (IDX);                                                                                      \
                    const float4 rec = csph[e_];                          /* {c - centre, r} */                                     \
                    const uint32_t prim = ma.sphere_prims[e_];                                                                      \
                    if (IN) {                                                                                                       \
                        const float ox = rec.x - sr.px, oy = rec.y - sr.py, oz = rec.z - sr.pz;                                     \
                        const float b = __builtin_fmaf(ox, sr.dx, __builtin_fmaf(oy, sr.dy, oz * sr.dz));                           \
                        const float lx = __builtin_fmaf(-b, sr.dx, ox), ly = __builtin_fmaf(-b, sr.dy, oy), lz = __builtin_fmaf(-b, sr.dz, oz);\
                        const float l2 = __builtin_fmaf(lx, lx, __builtin_fmaf(ly, ly, lz * lz));                                   \
                        const float Dl = __builtin_fmaf(rec.w, rec.w, -l2);                                                         \
                        const float G = __builtin_fmaf(sr.Kg, rec.w, sr.c0);                                                        \
                        const float Dp = Dl + G;                                                                                    \
                        if (Dp >= 0.0f) {                                                                                           \
                            const float tlo = b - __builtin_amdgcn_sqrtf(Dp) * (1.0f + 4.76837158e-7f) - sr.K;                      \
                            const float Dm = Dl - G;                                                                                \
                            const float thi = Dm > 0.0f ? b - __builtin_amdgcn_sqrtf(Dm) * (1.0f - 4.76837158e-7f) + sr.K : __builtin_inff();\
                            if (tlo <= best_up && !(thi < 0.0f)) {                                                                  \
                                if (tlo > sr.K) best_up = fminf(best_up, thi);                                                      \
                                if (!mesh_queue_room(lq, tid, qcnt, best_up, 1u)) {                                                 \
                                    if (wf_flush_to_extra(st, (uint32_t)p, lq, kMeshQueue, tid, qcnt, best_up)) extra = true;       \
                                    else { overflow = true; best_up = -__builtin_inff(); }                                          \
                                }                                                                                                   \
                                if (!overflow) {                                                                                    \
                                    lq[(size_t)qcnt * kBvhThreads + tid] = prim;                                                    \
                                    lq[(size_t)(kMeshQueue + qcnt) * kBvhThreads + tid] = __float_as_uint(tlo);                     \
                                    qcnt += 1;                                                                                      \
                                }                                                                                                   \
                            }                                                                                                       \
                        }                                                                                                           \
                    }                                                                                                               \
                }

// PLAIN 1: every node is a footprint node, every leaf a triangle leaf (C3, C5).  PLAIN 0: a joint tree -- 3-D nodes over
// sphere boxes and the footprints of the other planes, the (x, y) footprint sub-tree behind links with kBvhFlatNode, sphere
// leaves with bvh_traverse_spheres' bounds (cmax_ru: SceneView::sphere_cmax rounded up, for their error terms).
template <int PLAIN>
__global__ __launch_bounds__(kBvhThreads, PLAIN ? kPkWaves : kPkWavesJoint) void wf_trace_packet_kernel(const WfState st, Counters *__restrict__ ctr,
                                                                                const float4 *__restrict__ nodes, const MeshArrays ma,
                                                                                uint32_t root, float cmax_ru, uint32_t lane_stack, const MeshTileLists tl)
{
    __shared__ uint32_t pk_stack[kBvhThreads >> 6][kPkStack];
    __shared__ uint32_t lds_q[2 * kMeshQueue][kBvhThreads];
    uint32_t *const lq = &lds_q[0][0];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    uint32_t *const stk = &pk_stack[tid >> 6][0];
    const unsigned long long n_queue = st.count[0];
    const unsigned long long n_tiles = (n_queue + 63ull) >> 6;
    const WfRec *__restrict__ recs = st.rec[0];
    unsigned long long *const head = &st.work[0];                  // counts tiles here
    unsigned long long grab = n_tiles / ((unsigned long long)gridDim.x * (kBvhThreads >> 6) * 8ull);
    grab = grab > 8ull ? 8ull : (grab < 1ull ? 1ull : grab);
    unsigned long long t_next = 0, t_end = 0;
    unsigned long long box_tests = 0, leaf_filters = 0;
    const PkConst4 cnodes = pk_const(nodes), ctri = pk_const(ma.tri_f32), cgeo = pk_const(ma.tri_geo), csph = pk_const(ma.sphere_cr);

    for (;;) {
        if (t_next >= t_end) {
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(head, grab);
            base = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
                   __builtin_amdgcn_readfirstlane((uint32_t)base);
            if (base >= n_tiles) break;
            t_next = base;
            t_end = base + grab < n_tiles ? base + grab : n_tiles;
        }
        const unsigned long long p = (t_next << 6) + lane;
        t_next += 1;
        WfRec w;
        w.ridx = kNone; w.best_up = 0.f;
        w.px = w.py = w.pz = w.dx = w.dy = w.dz = w.ix = w.iy = w.iz = w.nx = w.ny = w.nz = w.A = w.slack = 0.f;
        if (p < n_queue) w = recs[p];
        const bool valid = w.ridx != kNone;
        const bool walk = valid && w.best_up == w.best_up;         // (NaN: no f32 walk for this origin)
        bool overflow = valid && !walk, extra = false;
        Ray32S q;
        TriFilterParams tp;
        q.ix = w.ix; q.iy = w.iy; q.iz = w.iz; q.nx = w.nx; q.ny = w.ny; q.nz = w.nz; q.e = w.slack;
        tp.dx = w.dx; tp.dy = w.dy; tp.dz = w.dz; tp.npx = -w.px; tp.npy = -w.py; tp.npz = -w.pz; tp.A = w.A; tp.pad = 0.f;
        float best_up = walk ? w.best_up : -__builtin_inff();       // -inf: this lane enters nothing
        uint32_t qcnt = 0, nbox = 0, nleaf = 0;
        uint32_t sp = 0;
        uint32_t node = __ballot(walk) != 0ull ? (PLAIN ? (root & ~kBvhFlatNode) : root) : kNone;
        SphereRay sr;                                               // (joint trees: sphere_ray_from's terms, from the record's f32 origin)
        sr.px = w.px; sr.py = w.py; sr.pz = w.pz; sr.dx = w.dx; sr.dy = w.dy; sr.dz = w.dz;
        sr.Kg = 0.0f; sr.c0 = __builtin_inff(); sr.K = 0.0f;
        if constexpr (!PLAIN) {
            const float pn = __builtin_amdgcn_sqrtf(__builtin_fmaf(w.px, w.px, __builtin_fmaf(w.py, w.py, w.pz * w.pz))) * (1.0f + 9.5367432e-7f);
            const float M = (cmax_ru + pn) * (1.0f + 2.3841858e-7f);              // >= sphere_cmax + |p|
            if (M < 1.0e14f && M > 1.0e-12f) {
                const float u = 5.9604645e-8f;
                sr.Kg = 128.0f * u * M * (1.0f + 4.76837158e-7f);
                sr.c0 = 8192.0f * u * u * M * M * (1.0f + 9.5367432e-7f);
                sr.K = 24.0f * u * M * (1.0f + 4.76837158e-7f);
            } else {                                  // outside the range the bounds were derived for: every sphere of a visited leaf is a candidate
                sr.px = sr.py = sr.pz = sr.dx = sr.dy = sr.dz = 0.0f;
            }
        }

        // the tile's list, if it has one (wave-uniform: scalar loads), instead of the walk
        {
            if (tl.count != nullptr && node != kNone) {
                const uint32_t tile_of = (uint32_t)((t_next - 1ull) % (unsigned long long)tl.tiles_per_sample);
                const uint32_t list_n = pk_const(tl.count)[tile_of];
                if (list_n != kMeshTileWalk) {
                    const PkConstU32 ent = pk_const(reinterpret_cast<const uint32_t *>(tl.entries + (size_t)tile_of * kMeshTileCap));
                    for (uint32_t k = 0; k < list_n; ++k) {
                        const uint32_t e = ent[2 * k];
                        const float t_lb = __uint_as_float(ent[2 * k + 1]);
                        const bool in = walk && t_lb <= best_up;
                        if (RTX_PK_BALLOT(in) == 0ull) break;      // sorted by t_lb: nothing further can beat any lane's certain hit
                        if (in) nleaf += 1;
                        if (!PLAIN && (e & kMeshTileSphere) != 0u) RTX_PK_SPH_RECORD(e & ~kMeshTileSphere, in)
                        else RTX_PK_TRI_RECORD(e, in)
                    }
                    node = kNone;
                }
            }
        }
        int stk_v = 0;                                              // the lane-stack form of the wave's stack
        while (node != kNone) {
            // the node's bytes at a wave-uniform address.  A footprint node: 4 x {lo.x, lo.y, hi.x, hi.y}, the links, the counts
            // (96 bytes); a 3-D node of a joint tree: 4 x {lo.xyz, link}, 4 x {hi.xyz, count}
            const PkConst4 np = cnodes + 8 * (size_t)(node & ~kBvhFlatNode);
            uint32_t lnk[4], cnt[4];
            float tc[4];
            if (PLAIN || (node & kBvhFlatNode)) {
                const float4 r0 = np[0], r1 = np[1], r2 = np[2], r3 = np[3], l4 = np[4], c4 = np[5];
                lnk[0] = pk_bits(l4.x); lnk[1] = pk_bits(l4.y); lnk[2] = pk_bits(l4.z); lnk[3] = pk_bits(l4.w);
                cnt[0] = pk_bits(c4.x); cnt[1] = pk_bits(c4.y); cnt[2] = pk_bits(c4.z); cnt[3] = pk_bits(c4.w);
                tc[0] = rect_entry32(r0, q, best_up); tc[1] = rect_entry32(r1, q, best_up);
                tc[2] = rect_entry32(r2, q, best_up); tc[3] = rect_entry32(r3, q, best_up);
            } else {
                const float4 a0 = np[0], a1 = np[1], a2 = np[2], a3 = np[3], b0 = np[4], b1 = np[5], b2 = np[6], b3 = np[7];
                lnk[0] = pk_bits(a0.w); lnk[1] = pk_bits(a1.w); lnk[2] = pk_bits(a2.w); lnk[3] = pk_bits(a3.w);
                cnt[0] = pk_bits(b0.w); cnt[1] = pk_bits(b1.w); cnt[2] = pk_bits(b2.w); cnt[3] = pk_bits(b3.w);
                tc[0] = box_entry32(a0, b0, q, best_up); tc[1] = box_entry32(a1, b1, q, best_up);
                tc[2] = box_entry32(a2, b2, q, best_up); tc[3] = box_entry32(a3, b3, q, best_up);
            }
            if (best_up >= 0.0f) nbox += 4;
            uint32_t key[4], kl[4];                                // keys: the bits of a non-negative float order like the float
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const bool in = tc[c] < __builtin_inff();
                const unsigned long long hm = RTX_PK_BALLOT(in);
                key[c] = 0x7F800000u;
                kl[c] = PLAIN ? (lnk[c] & ~kBvhFlatNode) : lnk[c];
                if (hm == 0ull || cnt[c] == 0xFFFFFFFFu) continue;
                if (cnt[c] == 0u) {                                // interior: opened for the wave, ordered by its first entering lane
                    const int k = (int)__builtin_amdgcn_readlane(__float_as_uint(tc[c]), (int)(__ffsll((long long)hm) - 1));
                    key[c] = (uint32_t)(k < 0 ? 0 : k);            // (a slack can make the bound negative)
                    continue;
                }
                // a leaf (uniform addresses): the lanes that enter it filter and bound its records
                const uint32_t first = lnk[c], n = cnt[c] & 0xFFFFu;
                if (in) nleaf += n;
                if (!PLAIN && (cnt[c] & kBvhTriLeaf) == 0u) {         // spheres: bvh_traverse_spheres' bounds (rtx_traverse.h)
                    for (uint32_t j = 0; j < n; ++j) RTX_PK_SPH_RECORD(first + j, in)
                    continue;
                }
                for (uint32_t j = 0; j < n; ++j) RTX_PK_TRI_RECORD(first + j, in)
            }
            // interior children any lane entered, nearest first (wave-uniform integer keys: scalar code); the farther ones go
            // to the wave's stack.  None or one entered (most visits below the top levels): nothing to order, nothing to push --
            // the network and the three pushes are ~45 scalar instructions, and this walk is bound by the CU's scalar pipe
#ifndef RTX_PK_FAST1
#define RTX_PK_FAST1 1
#endif
#if RTX_PK_FAST1
            {
                const uint32_t vb = (key[0] != 0x7F800000u ? 1u : 0u) | (key[1] != 0x7F800000u ? 2u : 0u) |
                                    (key[2] != 0x7F800000u ? 4u : 0u) | (key[3] != 0x7F800000u ? 8u : 0u);
                if ((vb & (vb - 1u)) == 0u) {
                    node = vb == 0u ? kNone : (vb == 1u ? kl[0] : (vb == 2u ? kl[1] : (vb == 4u ? kl[2] : kl[3])));
                    if (node == kNone && sp != 0u) {
                        sp -= 1;
                        node = lane_stack ? (uint32_t)__builtin_amdgcn_readlane(stk_v, (int)sp) : __builtin_amdgcn_readfirstlane(stk[sp]);
                    }
                    continue;
                }
            }
#endif
#define RTX_CSWAP(i, j) { if (key[j] < key[i]) { uint32_t tk = key[i]; key[i] = key[j]; key[j] = tk; uint32_t tl = kl[i]; kl[i] = kl[j]; kl[j] = tl; } }
            RTX_CSWAP(0, 1) RTX_CSWAP(2, 3) RTX_CSWAP(0, 2) RTX_CSWAP(1, 3) RTX_CSWAP(1, 2)
#undef RTX_CSWAP
            if (lane_stack) {
                // three unconditional pushes, farthest first; an invalid one lands on the slot the next push overwrites
                stk_v = rtx_writelane((int)kl[3], (int)sp, stk_v); sp += key[3] < 0x7F800000u ? 1u : 0u;
                stk_v = rtx_writelane((int)kl[2], (int)sp, stk_v); sp += key[2] < 0x7F800000u ? 1u : 0u;
                stk_v = rtx_writelane((int)kl[1], (int)sp, stk_v); sp += key[1] < 0x7F800000u ? 1u : 0u;
            } else {
                if (key[3] < 0x7F800000u && sp < (uint32_t)kPkStack) { if (lane == 0) stk[sp] = kl[3]; sp += 1; }
                if (key[2] < 0x7F800000u && sp < (uint32_t)kPkStack) { if (lane == 0) stk[sp] = kl[2]; sp += 1; }
                if (key[1] < 0x7F800000u && sp < (uint32_t)kPkStack) { if (lane == 0) stk[sp] = kl[1]; sp += 1; }
            }
            node = key[0] < 0x7F800000u ? kl[0] : kNone;
            if (node == kNone && sp != 0u) {
                sp -= 1;
                node = lane_stack ? (uint32_t)__builtin_amdgcn_readlane(stk_v, (int)sp) : __builtin_amdgcn_readfirstlane(stk[sp]);
            }
        }
        if (p < n_queue) {
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
            c.e[6] = w.ridx;
            c.count = !valid ? kWfDead : (overflow ? kWfFallback : (k | (extra ? kWfExtra : 0u)));
            st.cand[p] = c;
        }
        box_tests += nbox; leaf_filters += nleaf;
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

#ifdef RTX_LAB      // RTX_TUNE_BEAMS: librtx_hip_lab.so only
// ---- the walk of level 0 as beams ------------------------------------------------------------------------------------------
// The packet walk above opens ONE node per step for the whole wave: every lane tests its ray against the four rectangles, the
// ordering and the stack are scalar code -- 369 such steps per C3 tile, ~60 VALU and ~70 SALU instructions each, and the kernel
// is bound by the CU's one scalar pipe.  But WHICH nodes a tile opens hardly depends on the lane: its 64 rays leave (almost) one
// point a few pixels apart.  So here the lanes test different NODES against the tile's BEAM: a step takes up to 16 nodes off the
// wave's stack, lane l tests child l & 3 of node l >> 2 (its rectangle, link and count: three quad-coalesced loads) against
// interval bounds of all the tile's rays at once, the interior children that pass go back to the stack (ballot + mbcnt), and the
// leaves that pass are handled as in the packet walk: records at a wave-uniform address through the scalar cache, every lane
// filtering and bounding them for ITS ray behind ITS own test of the leaf's rectangle.  ~23 steps per C3 tile instead of 369.
//
// The beam test is rect_entry32 with every term replaced by a bound over the tile: a lane's slab distances are
// fl(b * i + n) with its i in [i_min, i_max] and n in [n_min, n_max] (wave reductions over the lanes that walk), so
// min(fl(b i_min + n_min), fl(b i_max + n_min)) <= each of them <= max(fl(b i_min + n_max), fl(b i_max + n_max)) -- b * i is
// linear in i and rounding is monotone --, for either plane of the slab and any mix of signs; entry / exit / widening / slack /
// best_up are then combined exactly as a lane does, with the largest slack and the largest best_up.  Whatever a lane's own test
// accepts the beam accepts: a lane sees every leaf its own walk would open, in some order, and collects every candidate with
// t_lo <= its best_up -- the exact tests decide as always.  What is given up is the order (no nearest-first: a tile of rays that
// all end early still sees the beam's whole length until every lane has a certain hit).
// MEASURED AND NOT SHIPPED (RTX_TUNE_BEAMS selects it; LAB_NOTEBOOK R3.8): bit-identical, and slower -- C3 39.0 against 36.2 ms,
// C5 band 80.3 against 52.2.  The premise was wrong: a mesh tile pays for its leaf records (the wave opens ~400 leaves = 2000
// records per C3 tile, every one filtered by all 64 lanes), not for its node visits, and the beam adds the leaves an ordered walk
// never reaches once its rays have ended (C5: most primary rays end early).
constexpr int kBeamStack = 512;                                    // wave-uniform stack entries in LDS; a step that could overflow it opens fewer nodes
constexpr int kBeamNodes = 16;                                     // nodes opened per step: 64 children, one per lane
#ifndef RTX_BEAM_WAVES
#define RTX_BEAM_WAVES 6
#endif
constexpr int kBeamWaves = RTX_BEAM_WAVES;                         // workgroups per CU

__device__ __forceinline__ float beam_wave_min(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float beam_wave_max(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

struct Beam { float ixa, ixb, iya, iyb, nxa, nxb, nya, nyb, e, best; };      // a: minimum, b: maximum over the tile's walking lanes

__device__ __forceinline__ bool beam_enters_rect(const float4 r4, const Beam &b)
{
    const float lx = fminf(fminf(__builtin_fmaf(r4.x, b.ixa, b.nxa), __builtin_fmaf(r4.x, b.ixb, b.nxa)),
                           fminf(__builtin_fmaf(r4.z, b.ixa, b.nxa), __builtin_fmaf(r4.z, b.ixb, b.nxa)));
    const float ux = fmaxf(fmaxf(__builtin_fmaf(r4.x, b.ixa, b.nxb), __builtin_fmaf(r4.x, b.ixb, b.nxb)),
                           fmaxf(__builtin_fmaf(r4.z, b.ixa, b.nxb), __builtin_fmaf(r4.z, b.ixb, b.nxb)));
    const float ly = fminf(fminf(__builtin_fmaf(r4.y, b.iya, b.nya), __builtin_fmaf(r4.y, b.iyb, b.nya)),
                           fminf(__builtin_fmaf(r4.w, b.iya, b.nya), __builtin_fmaf(r4.w, b.iyb, b.nya)));
    const float uy = fmaxf(fmaxf(__builtin_fmaf(r4.y, b.iya, b.nyb), __builtin_fmaf(r4.y, b.iyb, b.nyb)),
                           fmaxf(__builtin_fmaf(r4.w, b.iya, b.nyb), __builtin_fmaf(r4.w, b.iyb, b.nyb)));
    const float tn = fmaxf(fmaxf(lx, ly), 0.0f);
    const float tf = fminf(ux, uy);
    const float tn_lo = __builtin_fmaf(tn, 1.0f - 4.76837158e-7f, -b.e);
    const float tf_hi = __builtin_fmaf(tf, 1.0f + 4.76837158e-7f, b.e);
    return (tn_lo <= tf_hi) && (tn_lo <= b.best);
}

__global__ __launch_bounds__(kBvhThreads, kBeamWaves) void wf_trace_beam_kernel(const WfState st, Counters *__restrict__ ctr,
                                                                                const float4 *__restrict__ nodes, const MeshArrays ma,
                                                                                uint32_t root)
{
    __shared__ uint32_t bm_stack[kBvhThreads >> 6][kBeamStack];
    __shared__ uint32_t lds_q[2 * kMeshQueue][kBvhThreads];
    uint32_t *const lq = &lds_q[0][0];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    uint32_t *const stk = &bm_stack[tid >> 6][0];
    const unsigned long long n_queue = st.count[0];
    const unsigned long long n_tiles = (n_queue + 63ull) >> 6;
    const WfRec *__restrict__ recs = st.rec[0];
    unsigned long long *const head = &st.work[0];                  // counts tiles here
    unsigned long long grab = n_tiles / ((unsigned long long)gridDim.x * (kBvhThreads >> 6) * 8ull);
    grab = grab > 8ull ? 8ull : (grab < 1ull ? 1ull : grab);
    unsigned long long t_next = 0, t_end = 0;
    unsigned long long box_tests = 0, leaf_filters = 0;
    const PkConst4 ctri = pk_const(ma.tri_f32), cgeo = pk_const(ma.tri_geo);
    const float inf = __builtin_inff();

    for (;;) {
        if (t_next >= t_end) {
            unsigned long long base = 0;
            if (lane == 0) base = atomicAdd(head, grab);
            base = ((unsigned long long)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32)) << 32) |
                   __builtin_amdgcn_readfirstlane((uint32_t)base);
            if (base >= n_tiles) break;
            t_next = base;
            t_end = base + grab < n_tiles ? base + grab : n_tiles;
        }
        const unsigned long long p = (t_next << 6) + lane;
        t_next += 1;
        WfRec w;
        w.ridx = kNone; w.best_up = 0.f;
        w.px = w.py = w.pz = w.dx = w.dy = w.dz = w.ix = w.iy = w.iz = w.nx = w.ny = w.nz = w.A = w.slack = 0.f;
        if (p < n_queue) w = recs[p];
        const bool valid = w.ridx != kNone;
        const bool walk = valid && w.best_up == w.best_up;         // (NaN: no f32 walk for this origin)
        bool overflow = valid && !walk, extra = false;
        Ray32S q;
        TriFilterParams tp;
        q.ix = w.ix; q.iy = w.iy; q.iz = w.iz; q.nx = w.nx; q.ny = w.ny; q.nz = w.nz; q.e = w.slack;
        tp.dx = w.dx; tp.dy = w.dy; tp.dz = w.dz; tp.npx = -w.px; tp.npy = -w.py; tp.npz = -w.pz; tp.A = w.A; tp.pad = 0.f;
        float best_up = walk ? w.best_up : -inf;                    // -inf: this lane enters nothing
        uint32_t qcnt = 0, nbox = 0, nleaf = 0;
        uint32_t sp = 0;
        Beam bm;
        bm.ixa = bm.ixb = bm.iya = bm.iyb = bm.nxa = bm.nxb = bm.nya = bm.nyb = bm.e = 0.0f; bm.best = -inf;
        if (__ballot(walk) != 0ull) {
            bm.ixa = beam_wave_min(walk ? w.ix : inf); bm.ixb = beam_wave_max(walk ? w.ix : -inf);
            bm.iya = beam_wave_min(walk ? w.iy : inf); bm.iyb = beam_wave_max(walk ? w.iy : -inf);
            bm.nxa = beam_wave_min(walk ? w.nx : inf); bm.nxb = beam_wave_max(walk ? w.nx : -inf);
            bm.nya = beam_wave_min(walk ? w.ny : inf); bm.nyb = beam_wave_max(walk ? w.ny : -inf);
            bm.e = beam_wave_max(walk ? w.slack : 0.0f);
            bm.best = beam_wave_max(best_up);
            if (lane == 0) stk[0] = root & ~kBvhFlatNode;
            sp = 1;
        }
        while (sp != 0u) {
            // how many nodes this step opens: opening k leaves at most sp + 3 k entries
            uint32_t k = sp < (uint32_t)kBeamNodes ? sp : (uint32_t)kBeamNodes;
            const uint32_t room = ((uint32_t)kBeamStack - sp) / 3u;
            k = k < room ? k : room;
            if (k == 0u) {                                          // a tree deeper than the stack can follow: every shape for these rays
                if (walk) { overflow = true; best_up = -inf; }
                break;
            }
            const uint32_t slot = lane >> 2, c = lane & 3u;
            const bool act = slot < k;
            float4 rc = make_float4(0.f, 0.f, 0.f, 0.f);
            uint32_t lk = 0u, ct = 0xFFFFFFFFu;
            if (act) {
                const uint32_t nd = stk[sp - 1u - slot];
                const float4 *np = nodes + 8 * (size_t)nd;             // a footprint node: 4 x {lo.x, lo.y, hi.x, hi.y}, the links, the counts
                rc = np[c];
                lk = reinterpret_cast<const uint32_t *>(np + 4)[c];
                ct = reinterpret_cast<const uint32_t *>(np + 5)[c];
            }
            sp -= k;
            const bool in = act && ct != 0xFFFFFFFFu && beam_enters_rect(rc, bm);
            if (act) nbox += 1;
            const bool inner = in && ct == 0u;
            const unsigned long long im = RTX_PK_BALLOT(inner);
            if (inner) stk[sp + bvh_mbcnt(im)] = lk & ~kBvhFlatNode;
            sp += (uint32_t)__popcll(im);
            unsigned long long lm = RTX_PK_BALLOT(in && ct != 0u);
            bool bounded = false;
            while (lm != 0ull) {
                const int l = __ffsll((long long)lm) - 1;
                lm &= lm - 1ull;
                const uint32_t first = (uint32_t)__builtin_amdgcn_readlane((int)lk, l);
                const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)ct, l) & 0xFFFFu;
                float4 lr;
                lr.x = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(rc.x), l));
                lr.y = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(rc.y), l));
                lr.z = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(rc.z), l));
                lr.w = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(rc.w), l));
                const bool lin = rect_entry32(lr, q, best_up) < inf;           // the lane's own test of the leaf's rectangle
                if (best_up >= 0.0f) nbox += 1;
                if (RTX_PK_BALLOT(lin) == 0ull) continue;
                if (lin) nleaf += n;
                for (uint32_t j = 0; j < n; ++j) {
                    const PkConst4 rp = ctri + 2 * (size_t)(first + j);
                    const float4 A = rp[0], B = rp[1];
                    const bool pass = lin && (int)tri_filter_sign(A, B, tp) >= 0;
                    if (RTX_PK_BALLOT(pass) == 0ull) continue;
                    const PkConst4 gp = cgeo + 2 * (size_t)(first + j);
                    const float4 g0 = gp[0], g1 = gp[1];
                    if (pass) {
                        float thi;
                        const float tlo = tri_bounds(A, g0, g1, tp, thi);
                        if (tlo <= best_up && tlo < inf) {
                            bounded = bounded || thi < best_up;
                            best_up = fminf(best_up, thi);
                            if (!mesh_queue_room(lq, tid, qcnt, best_up, 1u)) {       // more live candidates than the queue holds:
                                if (wf_flush_to_extra(st, (uint32_t)p, lq, kMeshQueue, tid, qcnt, best_up)) extra = true;   // to the overflow list
                                else { overflow = true; best_up = -inf; }                  // (full: the shade kernel tests every shape for this ray)
                            }
                            if (!overflow) {
                                lq[(size_t)qcnt * kBvhThreads + tid] = (first + j) | kQueueTri;
                                lq[(size_t)(kMeshQueue + qcnt) * kBvhThreads + tid] = __float_as_uint(tlo);
                                qcnt += 1;
                            }
                        }
                    }
                }
            }
            if (RTX_PK_BALLOT(bounded) != 0ull) bm.best = beam_wave_max(best_up);      // the beam ends where its last ray ends
        }
        if (p < n_queue) {
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
            c.e[6] = w.ridx;
            c.count = !valid ? kWfDead : (overflow ? kWfFallback : (k | (extra ? kWfExtra : 0u)));
            st.cand[p] = c;
        }
        box_tests += nbox; leaf_filters += nleaf;
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

#endif  // RTX_LAB

// ---- closest_object's exact part, ray_hit, the next segment's set-up ---------------------------------------------------------
__global__ __launch_bounds__(kBvhThreads, kWfShadeWaves) void wf_shade_kernel(const SceneView *__restrict__ svp, const RowsView *__restrict__ rvp,
                                                                  const WfState st, uint32_t level, double *__restrict__ samples,
                                                                  Counters *__restrict__ ctr, const LeafArrays la, uint32_t queue_only)
{
    const SceneView &sv = *svp;
    const RowsView &rv = *rvp;
    __shared__ uint32_t lds_append[16];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t bounce_limit = sv.max_bounces >= 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)sv.max_bounces + 1u;
    const unsigned long long n_queue = st.count[0];
    WfRec *__restrict__ recs_out = st.rec[1];
    const WfRays &rin = st.ray[0], &rout = st.ray[1];
    unsigned long long segs = 0, exact = 0;

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
        WfRec w;
        RayState r;
        uint32_t lt = kNone;
        double t0 = 0.0;
        if (have) {
            const uint32_t ridx = c.e[6];
            r.pos = mk(rin.pos[0][p], rin.pos[1][p], rin.pos[2][p]);
            r.dir = mk(rin.dir[0][p], rin.dir[1][p], rin.dir[2][p]);
            const RayX rx = make_rayx(r.pos, r.dir);
            Hit h;
            hit_init(h);
            ++segs;
            if (level != 0u) {                     // the pre-tested self-hit (none at level 0)
                const double ht = rin.hit_t[p];
                if (ht != 0.0) { const uint32_t left = rin.left[p]; h.t = ht; h.id = la.tris[left].id; h.kind = 2; h.local = left; }
            }
            bool covered = true;
            if (c.count & kWfFallback) {
                // no walk was possible (an origin beyond 2^27 x origin_limit, NaN) or the level's overflow list was full:
                // every triangle, exactly
                covered = false;
            } else {
                const uint32_t nc = c.count & 0xFFFFu;
                const uint32_t n = nc < (uint32_t)kMeshQueue ? nc : (uint32_t)kMeshQueue;
                if (c.count & kWfExtra) {          // this walk's queue ran full: the rest of its candidates are in the level's overflow list
                    const unsigned long long nx = st.xcount[0] < (unsigned long long)kWfExtraCap ? st.xcount[0] : (unsigned long long)kWfExtraCap;
                    for (unsigned long long i = 0; i < nx; ++i) {
                        const uint2 x = st.extra[i];
                        if (x.x != (uint32_t)p) continue;
                        double t;
                        if (x.y & kQueueTri) {
                            const uint32_t tk = la.tri_fidx[x.y & ~kQueueTri];
                            if (triangle_distance(la.tris[tk], rx, &t)) hit_consider(h, t, la.tris[tk].id, 2, tk);
                        } else if (sphere_distance(la.spheres[x.y], rx, &t)) hit_consider(h, t, la.sphere_ids[x.y], 0, x.y);
                        exact += 1;
                    }
                }
#pragma unroll 1
                for (uint32_t e = 0; e < n; ++e) {
                    const uint32_t idx = e == 0 ? c.e[0] : e == 1 ? c.e[1] : e == 2 ? c.e[2] : e == 3 ? c.e[3] : e == 4 ? c.e[4] : c.e[5];
                    double t;
                    if (idx & kQueueTri) {
                        const uint32_t tk = la.tri_fidx[idx & ~kQueueTri];
                        if (triangle_distance(la.tris[tk], rx, &t)) hit_consider(h, t, la.tris[tk].id, 2, tk);
                    } else if (sphere_distance(la.spheres[idx], rx, &t)) hit_consider(h, t, la.sphere_ids[idx], 0, idx);
                }
                exact += n;
            }
            // the shapes outside the tree: the spheres unless the tree holds them, the planes, the triangles past n_tri_tree
            const bool sweep_spheres = !covered || (sv.bvh_flags & 1u) == 0u;
            if (sweep_spheres) {
                for (uint32_t k = 0; k < sv.n_spheres; ++k) {
                    double t;
                    if (sphere_distance(la.spheres[k], rx, &t)) hit_consider(h, t, la.sphere_ids[k], 0, k);
                }
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
            exact += (sweep_spheres ? sv.n_spheres : 0u) + sv.n_planes + (sv.n_tri_filter - tri_sweep_from);

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
                lt = h.kind == 2u ? h.local : kNone;
                if (queue_only) {                  // the megakernel takes the ray from here and sets its next segment up itself
                    w.ridx = ridx;
                } else {
                    // the next segment: the reference's self-hit is tested here, exactly (rtx_bvh_mesh.hip), and bounds the walk
                    const RayX rn = make_rayx(r.pos, r.dir);
                    float bu = __builtin_inff();
                    if (lt != kNone) {
                        double t;
                        if (triangle_distance(la.tris[lt], rn, &t) && is_normal_positive(t)) { t0 = t; bu = round_up32(t); }
                        exact += 1;
                    }
                    wf_make_rec(sv, r.pos, r.dir, rn.dirn, bu, ridx, w);
                }
                next = true;
            }
        }
        // the survivors move to the next level's queue: record and state at their new slot, unit stride across the block
        const unsigned long long slot = wf_append_block(&st.count[1], next, lds_append, it);
        if (next) {
            if (queue_only) recs_out[slot].ridx = w.ridx;
            else recs_out[slot] = w;
            rout.pos[0][slot] = r.pos.x; rout.pos[1][slot] = r.pos.y; rout.pos[2][slot] = r.pos.z;
            rout.dir[0][slot] = r.dir.x; rout.dir[1][slot] = r.dir.y; rout.dir[2][slot] = r.dir.z;
            rout.res[0][slot] = r.result.x; rout.res[1][slot] = r.result.y; rout.res[2][slot] = r.result.z;
            rout.lig[0][slot] = r.light.x; rout.lig[1][slot] = r.light.y; rout.lig[2][slot] = r.light.z;
            rout.left[slot] = lt;
            if (!queue_only) rout.hit_t[slot] = t0;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        segs += __shfl_xor(segs, off, 64);
        exact += __shfl_xor(exact, off, 64);
    }
    if (lane == 0) {
        const uint32_t shard = (blockIdx.x * (kBvhThreads >> 6) + (tid >> 6)) & (kCounterShards - 1);
        if (segs) atomicAdd(&ctr[shard].segments, segs);
        if (exact) atomicAdd(&ctr[shard].exact_tests, exact);
    }
}

// ---- host ------------------------------------------------------------------------------------------------------------------
// Per ray of a launch: two sets of {12 f64 of state, the self-hit's distance, the triangle it left}, two 64-byte records,
// 32 bytes of candidates; per level three u64 counters; the overflow list.
size_t wavefront_state_bytes(uint64_t n_rays, uint32_t levels)
{
    (void)levels;
    return (size_t)n_rays * (2 * (13 * sizeof(double) + sizeof(uint32_t)) + 2 * sizeof(WfRec) + sizeof(WfCand)) +
           (size_t)(3 * kWfLevelsPerSync + 12) * sizeof(unsigned long long) + (size_t)kWfExtraCap * sizeof(uint2) +
           64 * 256;                              // (every array starts on a 256-byte boundary)
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

// The wavefront form of a tree that holds triangles: any pure (x, y)-footprint tree; a joint tree (spheres, footprints of
// other planes) when its level 0 can walk as packets -- the ray queue in tiles, a depth the wave-uniform stack holds.
bool wavefront_mesh_supported(const SceneView &sv, bool tiled)
{
    if ((sv.bvh_flags & 2u) == 0u || sv.n_bvh_nodes == 0) return false;
#ifdef RTX_LAB
    if ((sv.bvh_flags & 4u) != 0u) return true;      // (the lab's per-lane walk kernel takes a pure footprint tree of any depth, tiled or not)
#endif
    return tiled && 3u * sv.bvh_depth + 2u <= (uint32_t)kPkStack;
}

size_t wavefront_spill_bytes(const SceneView &sv, int n_cus)
{
    // one column per resident lane of the walk kernel; the hybrid's megakernel stage uses the same buffer
    const size_t own = (size_t)wavefront_spill_entries(sv) * (size_t)n_cus * kWfTraceWaves * kBvhThreads * sizeof(uint32_t);
    return std::max(own, bvh_mesh_spill_bytes(sv, n_cus));
}

// the same builder for a sphere tree (its nodes are 3-D, its leaves sphere leaves): lists in the sphere packets' format
hipError_t launch_build_sphere_tile_lists(const SceneView *d_sv, const RowsView *d_rv, const SceneView &sv, uint32_t n_tiles, uint32_t *count,
                                          TileEntry *entries, hipStream_t stream)
{
    if (n_tiles == 0) return hipSuccess;
    MeshTileLists tl{};
    tl.count = count; tl.entries = nullptr; tl.tiles_per_sample = n_tiles;
    hipLaunchKernelGGL(build_mesh_tile_lists_kernel, dim3((n_tiles + 3u) / 4u), dim3(256), 0, stream, d_sv, d_rv,
                       reinterpret_cast<const float4 *>(sv.bvh_nodes), (const float4 *)nullptr, tl, n_tiles, sv.bvh_root, kTileListCap, entries,
                       sv.bvh_leaf_cr, sv.bvh_prims);
    return hipGetLastError();
}

size_t wavefront_tile_list_bytes(uint64_t rays_per_sample)
{
    const uint64_t n_tiles = rays_per_sample >> 6;
    return (size_t)(((n_tiles * sizeof(uint32_t) + 255) & ~(uint64_t)255) + n_tiles * kMeshTileCap * sizeof(uint2));
}

hipError_t launch_trace_wavefront(const SceneView *d_sv, const SceneView &sv, const RowsView *d_rv, const RowsView &rv,
                                  double *samples, void *state_mem, Counters *counters, uint32_t *spill, int n_cus, hipStream_t stream,
                                  void *tile_list_mem, bool build_tile_lists)
{
    if (rv.n_rays == 0) return hipSuccess;
    const uint32_t levels = wavefront_levels(sv);
    const uint64_t n = rv.n_rays;
    WfState st;
    wf_carve(state_mem, n, st);

    LeafArrays la;
    la.sphere_f32 = sv.bvh_leaf_f32; la.sphere_prims = sv.bvh_prims; la.spheres = sv.spheres; la.sphere_ids = sv.sphere_id;
    la.tri_f32 = sv.tri_f32; la.tri_fidx = sv.tri_fidx; la.tris = sv.tris;
    MeshArrays ma;
    ma.sphere_cr = sv.bvh_leaf_cr; ma.sphere_prims = sv.bvh_prims; ma.tri_f32 = sv.tri_f32; ma.tri_geo = sv.tri_geo;
    const uint32_t spill_entries = spill ? wavefront_spill_entries(sv) : 0u;
    const bool deep = spill_entries != 0u;
    const float4 *nodes = reinterpret_cast<const float4 *>(sv.bvh_nodes);
    const float4 *qnodes = reinterpret_cast<const float4 *>(sv.bvh_qnodes);
#ifdef RTX_LAB
    const bool qn = (sv.bvh_flags & 8u) != 0u && (sv.tuning & RTX_TUNE_NO_QNODES) == 0u;
#endif
#ifdef RTX_LAB
    const uint32_t trace_blocks = (uint32_t)std::min<uint64_t>((n + kBvhThreads - 1) / kBvhThreads, (uint64_t)n_cus * kWfTraceWaves);
#endif
    const uint32_t shade_blocks = (uint32_t)std::min<uint64_t>((n + kBvhThreads - 1) / kBvhThreads, (uint64_t)n_cus * kWfShadeWaves);   // (grid-stride)
    auto generate = [&](const WfState &s0) {
        hipLaunchKernelGGL(wf_generate_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, stream, d_sv, d_rv, s0);
        return hipGetLastError();
    };
    // The hybrid (the default): only level 0 in this form -- its packets are what pays (primary rays alone: C5 band 38 ms
    // against the megakernel's 62, C3 30 against 68) -- and everything after the first hit in the regrouping megakernel, fed
    // from level 1's queue: its per-lane walks run beside other waves' f64 phases, which the per-lane walk kernel here cannot
    // offer (levels 1-10 of the C5 band: 55 ms here, ~33 ms there), and 4 kernels per launch instead of 23 suit small frames
    // (C3 at 960x540x8: 15.2 ms against 21.1 all-wavefront and 25.4 megakernel; 480x270x8: 6.5 / 11.6 / 8.0).  With 4 triangles
    // per leaf and the queue-fed stage's own regrouping threshold it is level with the all-wavefront form even on C3 at
    // 1080p x 64 spp (322.6 against 322.9 ms), so that form runs only on request: RTX_TUNE_WF_PURE (tests, A/B runs).
    const bool joint = (sv.bvh_flags & 4u) == 0u;      // spheres and / or footprints of other planes in the tree: packets + megakernel only
    // level 0 as packets: the ray queue in 8x8 tiles, a tree the wave-uniform stack can hold
#ifdef RTX_LAB
    const bool hybrid = joint || (sv.tuning & RTX_TUNE_WF_PURE) == 0u;
    const bool packets = rv.tiles_x != 0u && 3u * sv.bvh_depth + 2u <= (uint32_t)kPkStack && (joint || (sv.tuning & RTX_TUNE_NO_PACKETS) == 0u);
    if (joint && !packets) return hipErrorInvalidValue;      // (wavefront_mesh_supported() keeps the caller from asking)
#else
    // the product: the hybrid with packets, nothing else (the per-lane walk kernel of the other forms is a lab kernel)
    const bool hybrid = true;
    const bool packets = rv.tiles_x != 0u && 3u * sv.bvh_depth + 2u <= (uint32_t)kPkStack;
    if (!packets) return hipErrorInvalidValue;               // (wavefront_mesh_supported() keeps the caller from asking)
    (void)qnodes; (void)deep;
#endif
    const float cmax_ru = std::nextafterf((float)sv.sphere_cmax, INFINITY);
    const uint32_t packet_blocks = (uint32_t)std::min<uint64_t>((n + kBvhThreads - 1) / kBvhThreads, (uint64_t)n_cus * (joint ? kPkWavesJoint : kPkWaves));
#ifdef RTX_LAB
    // on request, a pure footprint tree: the tile's rays as ONE beam, the lanes across nodes (wf_trace_beam_kernel); any depth
    const bool beams = rv.tiles_x != 0u && !joint && (sv.tuning & RTX_TUNE_NO_PACKETS) == 0u && (sv.tuning & RTX_TUNE_BEAMS) != 0u;
    const uint32_t beam_blocks = (uint32_t)std::min<uint64_t>((n + kBvhThreads - 1) / kBvhThreads, (uint64_t)n_cus * kBeamWaves);
#endif
    auto level_fn = [&](const WfState &sk, uint32_t level) {
#ifdef RTX_LAB
        if (level == 0u && beams) {
            hipLaunchKernelGGL(wf_trace_beam_kernel, dim3(beam_blocks), dim3(kBvhThreads), 0, stream, sk, counters, nodes, ma, sv.bvh_root);
        } else
#endif
        if (level == 0u && packets) {
            const uint32_t lane_stack = 3u * sv.bvh_depth + 2u <= (uint32_t)kPkLaneStack && (sv.tuning & RTX_TUNE_PK_LDS_STACK) == 0u ? 1u : 0u;
            MeshTileLists tl{};
            if (tile_list_mem && (sv.tuning & RTX_TUNE_NO_TILE_LISTS) == 0u && rv.n_samples != 0u) {
                // what each tile's primary rays can pass the filter of, once per tile (build_mesh_tile_lists_kernel)
                const uint32_t n_tiles = (uint32_t)((rv.n_rays / rv.n_samples) >> 6);
                tl.count = reinterpret_cast<uint32_t *>(tile_list_mem);
                tl.entries = reinterpret_cast<uint2 *>(static_cast<char *>(tile_list_mem) + (((size_t)n_tiles * sizeof(uint32_t) + 255) & ~(size_t)255));
                tl.tiles_per_sample = n_tiles;
                if (build_tile_lists)
                    hipLaunchKernelGGL(build_mesh_tile_lists_kernel, dim3((n_tiles + 3u) / 4u), dim3(256), 0, stream, d_sv, d_rv, nodes, ma.tri_f32, tl,
                                   n_tiles, sv.bvh_root, kMeshTileCap, (TileEntry *)nullptr, (const float4 *)nullptr, (const uint32_t *)nullptr);
            }
            if (joint) hipLaunchKernelGGL(wf_trace_packet_kernel<0>, dim3(packet_blocks), dim3(kBvhThreads), 0, stream, sk, counters, nodes, ma, sv.bvh_root, cmax_ru, lane_stack, tl);
            else hipLaunchKernelGGL(wf_trace_packet_kernel<1>, dim3(packet_blocks), dim3(kBvhThreads), 0, stream, sk, counters, nodes, ma, sv.bvh_root, cmax_ru, lane_stack, tl);
        }
#ifdef RTX_LAB
        else if (qn) {
            if (deep) hipLaunchKernelGGL((wf_trace_kernel<true, 2>), dim3(trace_blocks), dim3(kBvhThreads), 0, stream, d_sv, sk, level, counters, qnodes, ma, spill, spill_entries);
            else hipLaunchKernelGGL((wf_trace_kernel<false, 2>), dim3(trace_blocks), dim3(kBvhThreads), 0, stream, d_sv, sk, level, counters, qnodes, ma, spill, spill_entries);
        } else {
            if (deep) hipLaunchKernelGGL((wf_trace_kernel<true, 1>), dim3(trace_blocks), dim3(kBvhThreads), 0, stream, d_sv, sk, level, counters, nodes, ma, spill, spill_entries);
            else hipLaunchKernelGGL((wf_trace_kernel<false, 1>), dim3(trace_blocks), dim3(kBvhThreads), 0, stream, d_sv, sk, level, counters, nodes, ma, spill, spill_entries);
        }
#else
        else return hipErrorInvalidValue;                    // (the product runs level 0 only, as packets)
#endif
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(wf_shade_kernel, dim3(shade_blocks), dim3(kBvhThreads), 0, stream, d_sv, d_rv, sk, level, samples, counters, la,
                           hybrid ? 1u : 0u);
        return hipGetLastError();
    };
    if (!hybrid) return wf_run_levels(st, levels, stream, generate, level_fn);
    hipError_t e = wf_run_levels(st, 1u, stream, generate, level_fn);
    if (e != hipSuccess || levels < 2u) return e;
    MeshRaySource src;
    for (int k = 0; k < 3; ++k) { src.pos[k] = st.ray[1].pos[k]; src.dir[k] = st.ray[1].dir[k]; src.res[k] = st.ray[1].res[k]; src.lig[k] = st.ray[1].lig[k]; }
    src.left = st.ray[1].left;
    src.ridx = &reinterpret_cast<const uint32_t *>(st.rec[1])[offsetof(WfRec, ridx) / sizeof(uint32_t)];
    src.ridx_stride = sizeof(WfRec) / sizeof(uint32_t);
    src.count = st.count + 1;
    return launch_trace_bvh_mesh_from_queue(d_sv, sv, d_rv, rv, samples, counters, st.work + 1, src, spill, n_cus, stream);
}

}  // namespace rtx
