// rtx_bvh_mesh.hip -- trace_bvh_mesh_kernel: RTX_KERNEL_BVH_REGROUP for trees that hold triangles (C3, C5).
//
// The schedule of trace_bvh_regroup_kernel (rtx_bvh_regroup.hip: a lane is TRAV / FIN / IDLE, the wave alternates between
// traversal steps for the TRAV lanes and an f64 phase that runs when enough lanes wait for it), the same tree, the same
// exact tests, the same bits -- with a traversal step that holds no f64 value:
//
//   * Spheres in a joint tree: the closest-approach bounds of bvh_traverse_spheres (rtx_traverse.h).
//   * Triangles: a candidate that passes the footprint filter (rtx_device.h) gets its distance and its barycentric
//     coordinates evaluated in f32 with explicit error bounds (tri_bounds below).  When the cull test, the plane
//     distance and the inside test all hold with margin, the reference CERTAINLY reports the hit (triangle.rs:108-127)
//     and t_hi bounds the winner's distance; every candidate carries t_lo, and only candidates with t_lo <= best_up
//     reach the exact f64 test -- in the f64 phase, after the walk.
//   * The reference's self-hit (a bounced ray re-hits the triangle it left at |t| ~ 1e-16, SURVEY H2d: most segments of a
//     mesh) cannot be certified in f32 -- whether it is reported hangs on the last bit of an f64 rounding -- so the f64
//     phase that sets a segment up tests the triangle the ray just left exactly and starts the walk with that bound.
//   * A lane whose candidate queue cannot take a leaf's records asks the f64 phase to test what it holds (FLUSH) and then
//     resumes its walk at the same node; the exhaustive sweep remains only for a single node that yields more live
//     candidates than the queue has entries (coincident shapes).
//
// Measured (same box): C3 130 -> 144 Mrays/s, C5 band 34.5 -> 39.8.  The mesh walk is latency-bound (SQ_WAIT_ANY /
// SQ_WAVE_CYCLES 0.73 / 0.76 with round 1's step, 0.66 / 0.68 with this one), so two ways to put more fetches in flight
// were tried and dropped: requesting the NEXT node before reading the current node's leaf records
// (-DRTX_MESH_PIPE=1: needs 32 more registers than 4 waves per SIMD leave; at 3 or 2 waves per SIMD it only breaks even) and
// 5 or 6 waves per SIMD (-DRTX_MESH_WAVES: in-loop spill reloads).
#include "rtx_launch.h"
#include "rtx_mesh_step.h"

#include <cstdlib>

namespace rtx {

#ifndef RTX_MESH_WAVES
#define RTX_MESH_WAVES 4
#endif
constexpr uint32_t kMeshWaves = RTX_MESH_WAVES;   // waves per SIMD (= workgroups per CU)
#ifndef RTX_MESH_SPLIT
#define RTX_MESH_SPLIT 1
#endif

constexpr int kMeshStack = (RTX_MESH_WAVES <= 4 ? 39 : 160 / RTX_MESH_WAVES) - 1 - 2 * kMeshQueue;   // LDS stack entries per lane: (entries + 1 sink row + 2 * kMeshQueue) KB per workgroup
// The queue-fed instance (rays in flight from the wavefront form's level 0) is built for RTX_MESH_WAVES_Q workgroups per CU:
// with 3 it has 168 instead of 128 VGPRs (fewer spills around the f64 phase) and 37 LDS stack entries (C5's tree, 3 * 11 + 2
// = 35, needs no HBM column).  Measured, whole wavefront launch: C3 37.9 -> 36.1 ms at 3 (5: 43.8), C5 band level; a joint tree's
// instance stays at 4 (240k axis-aligned faces: 166 ms at 4, 169 at 3); generating its own rays the kernel wants 4 (C5 band
// alone: 72.3 ms at 4, 80.4 at 3, 89.6 at 5).
// Segments that start on the triangle they left skip the walk (rtx_mesh_step.h, mesh_point_step).  Round 4's experiment, in the lab
// library only: it removes 45 % of the exact tests, 7 % of the box tests and 15-20 % of the fabric traffic -- and no time, because the
// traversal loop's iterations are set by the rays that LEAVE their triangle (C5: 4 % of the segments, 630 iterations each), beside
// which the short self-hit walks ran in otherwise idle lanes (LAB_NOTEBOOK R4.5: C5 98.2 -> 101.5 ms, C3 35.8 -> 34.9 ms, same box).
#ifndef RTX_MESH_POINT
#ifdef RTX_LAB
#define RTX_MESH_POINT 1
#else
#define RTX_MESH_POINT 0
#endif
#endif
constexpr bool kMeshPoint = RTX_MESH_POINT != 0;
#ifndef RTX_MESH_POINT_LANES
#define RTX_MESH_POINT_LANES 1
#endif
constexpr uint32_t kMeshPointLanes = RTX_MESH_POINT_LANES;   // lanes that locate a point before their step runs (8 / 16 measured: slower)
#ifndef RTX_MESH_WAVES_Q
#define RTX_MESH_WAVES_Q 3
#endif
constexpr uint32_t kMeshWavesQ = RTX_MESH_WAVES_Q;
// which instances run as fewer, fatter workgroups (kMeshWavesQ per CU: 168 registers per lane instead of 128)
#ifndef RTX_MESH_WIDE
#define RTX_MESH_WIDE 0
#endif
constexpr bool mesh_wide(bool queue, int plain) { return RTX_MESH_WIDE == 2 ? true : (RTX_MESH_WIDE == 1 ? queue : (queue && plain != 0)); }
constexpr int kMeshStackQ = 152 / RTX_MESH_WAVES_Q - 1 - 2 * kMeshQueue;      // (37 at 3: 150 KB per CU; at 159 KB only 2 workgroups fit)
// QUEUE: the rays are not generated here but taken from `src`, a queue of rays in flight at path level 1 (the hybrid of
// rtx_wavefront.hip: the primary rays of a mesh whose tree exceeds the L2s walk as packets there, and everything after the
// first hit runs here, where the f64 phases of other waves fill the waits of the per-lane walks).
template <bool SPILL, int PLAIN, bool QUEUE>
__global__ __launch_bounds__(kBvhThreads, mesh_wide(QUEUE, PLAIN) ? kMeshWavesQ : kMeshWaves) void trace_bvh_mesh_kernel(const SceneView *__restrict__ svp,
                                                                                 const RowsView *__restrict__ rvp,
                                                                                 double *__restrict__ samples, Counters *__restrict__ ctr,
                                                                                 unsigned long long *__restrict__ work_counter,
                                                                                 const float4 *__restrict__ nodes, const LeafArrays la,
                                                                                 const MeshArrays ma, uint32_t *__restrict__ spill,
                                                                                 uint32_t spill_entries, uint32_t thresh,
                                                                                 const MeshRaySource src)
{
    const SceneView &sv = *svp;
    const RowsView &rv = *rvp;
    const unsigned long long n_rays = QUEUE ? *src.count : rv.n_rays;
    constexpr int STACKN = mesh_wide(QUEUE, PLAIN) ? kMeshStackQ : kMeshStack;
    __shared__ uint32_t lds_stack[STACKN + 1][kBvhThreads];            // + the sink row of the branch-free pushes
    __shared__ uint32_t lds_q[2 * kMeshQueue][kBvhThreads];            // candidate entries, then their t_lo
    uint32_t *const ls = &lds_stack[0][0];
    uint32_t *const lq = &lds_q[0][0];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const size_t spill_stride = (size_t)gridDim.x * kBvhThreads, glane = (size_t)blockIdx.x * kBvhThreads + tid;
    const uint32_t bounce_limit = sv.max_bounces >= 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)sv.max_bounces + 1u;
    enum : uint32_t { S_IDLE = 0, S_TRAV = 1, S_FIN = 2, S_SETUP = 3, S_FLUSH = 4, S_POINT = 5 };

    unsigned long long wave_next = 0, wave_end = 0;      // this wave's share of the ray queue (wave-uniform)
    bool queue_empty = false;
    uint32_t state = S_IDLE;
    RayState r;
    Hit h;                                   // the exact winner so far (self-hit pre-test, flushes); completed in FIN
    Ray32 q;
    SphereRay sr;
    TriFilterParams tpar;
    float best_up = 0.f;
    uint32_t node = kNone, sp = 0, qcnt = 0, nbox = 0, nleaf = 0, resume = 0, resume_node = 0;
    constexpr bool kSplit = RTX_MESH_SPLIT != 0 && PLAIN != 1;      // the step in two halves (rtx_mesh_step.h; not built for 96-byte-node trees)
    MeshPending pend;
    pend.p0 = pend.p1 = pend.p2 = pend.p3 = 0u; pend.n = 0u;
    bool overflow = false, tree_used = false;
    uint32_t ridx = 0;
    uint32_t left_tri = kNone;               // the triangle (index in tris[]) the ray has just bounced off, if any
    // the triangles near the point the ray sits on (rtx_mesh_step.h, mesh_point_query): the key -- the point as the filter sees
    // it and the triangle it is on --, their number (kPointCacheBad: too many, walk) and their filter records
    uint32_t ck_x = 0x7FC00000u, ck_y = 0u, ck_z = 0u, ck_tri = kNone, c_n = 0u, c0 = 0u, c1 = 0u, c2 = 0u;
    unsigned long long segs = 0, box_tests = 0, leaf_filters = 0, exact = 0;
    hit_init(h);
    tri_filter_idle(tpar);
    q.ix = q.iy = q.iz = q.nx = q.ny = q.nz = 0.f;
    sr.px = sr.py = sr.pz = sr.dx = sr.dy = sr.dz = sr.Kg = sr.K = 0.f; sr.c0 = __builtin_inff();

#if defined(RTX_LAB) && defined(RTX_MESH_PROFILE)      // lab build: one count per build, reported through exact_tests
    unsigned long long rtx_prof = 0;
#define RTX_MPROF(k, cond, inc) { if (RTX_MESH_PROFILE == (k) && (cond)) rtx_prof += (inc); }
#define RTX_MSTAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime();
#define RTX_MTIME(k, a, b) { if (RTX_MESH_PROFILE == (k) && lane == 0) rtx_prof += ((b) - (a)) >> 6; }   // cycles / 64 of lane 0's wave
#else
#define RTX_MPROF(k, cond, inc)
#define RTX_MSTAMP(v)
#define RTX_MTIME(k, a, b)
#endif
    for (;;) {
        // ================= the f64 phase (entered when the inner loop below finds it due) =================
        if (__ballot(state != S_IDLE) == 0ull && queue_empty) break;       // wave-uniform: nothing live, nothing left to take
        RTX_MPROF(1, lane == 0, 1)                                          // outer iterations (f64 phases) per wave
        RTX_MPROF(2, state == S_FIN || state == S_FLUSH, 1)                 // lanes served by them
        RTX_MSTAMP(t_m0)
        if (state == S_FIN || state == S_FLUSH) {
            // ---- exact tests (sphere.rs:19-30, triangle.rs:108-127) of the candidates that can still be the winner
            const RayX rx = make_rayx(r.pos, r.dir);
#pragma unroll 1
            for (uint32_t e = 0; e < qcnt; ++e) {
                if (__uint_as_float(lq[(size_t)(kMeshQueue + e) * kBvhThreads + tid]) <= best_up) {
                    const uint32_t idx = lq[(size_t)e * kBvhThreads + tid];
                    double t;
                    if (idx & kQueueTri) {
                        const uint32_t tk = la.tri_fidx[idx & ~kQueueTri];
                        if (triangle_distance(la.tris[tk], rx, &t)) hit_consider(h, t, la.tris[tk].id, 2, tk);
                    } else {
                        if (sphere_distance(la.spheres[idx], rx, &t)) hit_consider(h, t, la.sphere_ids[idx], 0, idx);
                    }
                    exact += 1;
                }
            }
            qcnt = 0;
            if (state == S_FLUSH) {                    // resume the walk with the exact bound
                if (h.id != kNone) best_up = fminf(best_up, round_up32(h.t));
                state = S_TRAV;
            } else {
                // ---- the rest of closest_object (scene.rs:243-251) for this segment
                box_tests += nbox;
                leaf_filters += nleaf;
                nbox = 0; nleaf = 0;
                const bool covered = tree_used && !overflow;          // (a dropped subtree / candidate: every shape gets the exact test)
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
                const uint32_t tri_sweep_from = covered ? sv.n_tri_tree : 0u;
                for (uint32_t k = tri_sweep_from; k < sv.n_tri_filter; ++k) {
                    const uint32_t tk = la.tri_fidx[k];
                    double t;
                    if (triangle_distance(la.tris[tk], rx, &t)) hit_consider(h, t, la.tris[tk].id, 2, tk);
                }
                exact += sv.n_planes + (sv.n_tri_filter - tri_sweep_from);
                // ---- render_ray's match arm + ray_hit (scene.rs:232-239, 260-278)
                bool done = true;
                left_tri = kNone;
                if (h.id != kNone) {
                    if (h.kind == 2u) left_tri = h.local;
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
        }
        RTX_MSTAMP(t_m1)
        RTX_MTIME(9, t_m0, t_m1)                                            // cycles: exact tests + the rest of closest_object + ray_hit
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
                wave_end = base + rv.grab < n_rays ? base + rv.grab : n_rays;
                if (base >= n_rays) { queue_empty = true; wave_next = wave_end = 0; break; }
            }
            if (state == S_IDLE) {
                const unsigned long long my = wave_next + bvh_mbcnt(idle_mask);
                bool valid = my < wave_end;
                uint32_t pl = 0, smp = 0;
                if constexpr (QUEUE) {
                    if (valid) {                       // a ray in flight: its state as wf_shade_kernel left it after the first hit
                        ridx = src.ridx[(size_t)my * src.ridx_stride];
                        if (rv.tiles_x != 0u) (void)ray_index_to_pixel_tiled(rv, ridx, pl, smp);
                        else ray_index_to_pixel(rv, ridx, pl, smp);
                        const uint32_t k = fastdiv(pl, rv.div_width), x = pl - k * rv.width;
                        const uint64_t pix = (uint64_t)image_row(rv, k) * rv.width + x;
                        r.key = rng_key(sv.seed, pix, rv.sample_begin + smp);
                        r.bounce = 1u;
                        r.draw = 8u;
                        r.pos = mk(src.pos[0][my], src.pos[1][my], src.pos[2][my]);
                        r.dir = mk(src.dir[0][my], src.dir[1][my], src.dir[2][my]);
                        r.result = mk(src.res[0][my], src.res[1][my], src.res[2][my]);
                        r.light = mk(src.lig[0][my], src.lig[1][my], src.lig[2][my]);
                        left_tri = src.left[my];
                        state = S_SETUP;
                    }
                } else {
                    if (valid) {
                        if (rv.tiles_x != 0u) valid = ray_index_to_pixel_tiled(rv, my, pl, smp);
                        else ray_index_to_pixel(rv, my, pl, smp);
                    }
                    if (valid) {
                        gen_primary(sv, rv, pl, rv.sample_begin + smp, r);
                        ridx = (uint32_t)my;                                          // (the host keeps rv.n_rays below 2^32)
                        left_tri = kNone;
                        state = S_SETUP;
                    }
                }
            }
            const unsigned long long taken = (unsigned long long)__popcll(idle_mask);
            wave_next = wave_next + taken < wave_end ? wave_next + taken : wave_end;
        }
        RTX_MSTAMP(t_m2)
        RTX_MTIME(10, t_m1, t_m2)                                           // cycles: refill
        // ---- set up the next segment
        RTX_MPROF(3, state == S_SETUP, 1)                                   // lanes set up
        if (state == S_SETUP) {
            const RayX rx = make_rayx(r.pos, r.dir);
            hit_init(h);
            ++segs;
            qcnt = 0; sp = 0; overflow = false; resume = 0;
            best_up = __builtin_inff();
            // the reference's self-hit: decided by the last bit of an f64 rounding, so it is tested here, exactly, and the
            // walk starts with its distance as the bound (the triangle stays in the tree: a second test changes nothing)
            if (left_tri != kNone) {
                double t;
                if (triangle_distance(la.tris[left_tri], rx, &t)) hit_consider(h, t, la.tris[left_tri].id, 2, left_tri);
                exact += 1;
                if (h.id != kNone) best_up = round_up32(h.t);
            }
            const float omax = fmaxf(fmaxf(__builtin_fabsf((float)r.pos.x), __builtin_fabsf((float)r.pos.y)),
                                     __builtin_fabsf((float)r.pos.z));
            tree_used = omax <= sv.bvh_origin_limit;                                         // NaN origin -> exhaustive branch
            if (tree_used) {
                if constexpr (!PLAIN) { if (sv.bvh_flags & 1u) sphere_ray_from(sv, r.pos, r.dir, sr); }
                tri_filter_from_ray(sv, r.pos, r.dir, tpar);
                bool walk = true;
                if constexpr (PLAIN == 2 && kSplit && kMeshPoint) {
                    // a self-hit close enough for the point's set to decide it (t_self <= tau = tpar.A / 4): exact tests only, no walk
                    // -- after ONE point location per point, which runs in the traversal loop below (S_POINT)
                    if (h.id != kNone && left_tri != kNone && tpar.A < 1.0e29f && h.t <= (double)(tpar.A * 0.25f)) {
                        const bool same = __float_as_uint(tpar.npx) == ck_x && __float_as_uint(tpar.npy) == ck_y &&
                                          __float_as_uint(tpar.npz) == ck_z && left_tri == ck_tri;
                        if (!same) {
                            ck_x = __float_as_uint(tpar.npx); ck_y = __float_as_uint(tpar.npy); ck_z = __float_as_uint(tpar.npz);
                            ck_tri = left_tri;
                            c_n = 0u;
                            make_ray32(r.pos, rx.dirn, (double)sv.bvh_inv_max, q);      // (should the set turn out unusable: the walk's ray)
                            node = sv.bvh_root & ~kBvhFlatNode;
                            if constexpr (kSplit) pend.n = 0u;
                            state = S_POINT;
                            walk = false;
                        } else if (c_n <= kPointCache) {
                            if (c_n > 0u) { lq[(size_t)0 * kBvhThreads + tid] = c0 | kQueueTri; lq[(size_t)(kMeshQueue + 0) * kBvhThreads + tid] = 0u; }
                            if (c_n > 1u) { lq[(size_t)1 * kBvhThreads + tid] = c1 | kQueueTri; lq[(size_t)(kMeshQueue + 1) * kBvhThreads + tid] = 0u; }
                            if (c_n > 2u) { lq[(size_t)2 * kBvhThreads + tid] = c2 | kQueueTri; lq[(size_t)(kMeshQueue + 2) * kBvhThreads + tid] = 0u; }
                            qcnt = c_n;
                            node = kNone;
                            if constexpr (kSplit) pend.n = 0u;
                            state = S_FIN;
                            walk = false;
                        }
                    }
                }
                RTX_MPROF(7, walk, 1)                                       // segments that walk
                RTX_MPROF(8, state == S_POINT, 1)                           // segments that locate their point
                if (walk) {
                    make_ray32(r.pos, rx.dirn, (double)sv.bvh_inv_max, q);
                    node = sv.bvh_root;              // wide node 0 (flagged when it is a footprint node)
                    if constexpr (kSplit) { if (PLAIN == 2) node &= ~kBvhFlatNode; pend.n = 0u; }
                    state = S_TRAV;
                }
            } else {
                if (omax <= sv.bvh_origin_limit * kBvhRange64) {
                    // origin far outside the scene (rare: a bounce off one of the reference's far phantom hits): the whole
                    // walk right here with the f64 slab test, flushing inline
                    if constexpr (!PLAIN) { if (sv.bvh_flags & 1u) sphere_ray_from(sv, r.pos, r.dir, sr); }
                    tri_filter_from_ray(sv, r.pos, r.dir, tpar);
                    Ray64 q64;
                    make_ray64(r.pos, rx.dirn, (double)sv.bvh_inv_max, q64);
                    node = sv.bvh_root;
                    float4 nd[MeshNode<PLAIN>::n];
                    if constexpr (kMeshPipe) mesh_load_node<PLAIN>(nodes, node, nd);
                    while (node != kNone || resume != 0u) {
                        if (mesh_step<SPILL, PLAIN, STACKN>(nodes, ma, q64, sr, tpar, nd, node, sp, qcnt, overflow, best_up, resume, resume_node, ls, lq,
                                             tid, spill, spill_entries, spill_stride, glane, nbox, nleaf))
                            continue;
#pragma unroll 1
                        for (uint32_t e = 0; e < qcnt; ++e) {
                            if (__uint_as_float(lq[(size_t)(kMeshQueue + e) * kBvhThreads + tid]) <= best_up) {
                                const uint32_t idx = lq[(size_t)e * kBvhThreads + tid];
                                double t;
                                if (idx & kQueueTri) {
                                    const uint32_t tk = la.tri_fidx[idx & ~kQueueTri];
                                    if (triangle_distance(la.tris[tk], rx, &t)) hit_consider(h, t, la.tris[tk].id, 2, tk);
                                } else {
                                    if (sphere_distance(la.spheres[idx], rx, &t)) hit_consider(h, t, la.sphere_ids[idx], 0, idx);
                                }
                                exact += 1;
                            }
                        }
                        qcnt = 0;
                        if (h.id != kNone) best_up = fminf(best_up, round_up32(h.t));
                        if constexpr (kMeshPipe) mesh_load_node<PLAIN>(nodes, resume_node, nd);  // the flush interrupted this node's leaves
                    }
                    tree_used = true;
                }
                node = kNone;
                state = S_FIN;
            }
        }
        RTX_MSTAMP(t_m3)
        RTX_MTIME(11, t_m2, t_m3)                                           // cycles: set-up (self-hit pre-test, filter and slab parameters)
        // ================= traversal steps (f32 only) until the f64 phase is due again =================
        {
            float4 nd[MeshNode<PLAIN>::n];    // the node each TRAV lane opens next (the prefetched copy does not outlive this loop)
            if constexpr (kMeshPipe) { if (state == S_TRAV) mesh_load_node<PLAIN>(nodes, resume != 0u ? resume_node : node, nd); }
            if constexpr (kSplit) {
                for (;;) {
                    RTX_MPROF(4, lane == 0, 1)                              // traversal-loop iterations per wave
                    RTX_MPROF(5, state == S_TRAV, 1)                        // lanes walking in them
                    RTX_MPROF(6, state == S_POINT, 1)                       // lanes locating their point in them
                    if constexpr (PLAIN == 2 && kMeshPoint) {
                        // the lanes that locate their point: one node each; a finished set sends its lane to its exact tests (or,
                        // unusable, into the ordinary walk of this segment).  Like the leaf reads, the block runs when enough lanes
                        // want it (or nobody walks): issued for one or two lanes in every iteration it cost more than the walks it saves
                        const bool pt = state == S_POINT;
                        const uint32_t n_pt = (uint32_t)__popcll(__ballot(pt));
                        if (n_pt >= kMeshPointLanes || (n_pt != 0u && __ballot(state == S_TRAV) == 0ull)) {
                            if (pt) {
                                mesh_point_step<STACKN>(nodes, ma, la.tri_fidx, (float)r.pos.x, (float)r.pos.y, tpar,
                                                        tpar.A + sv.bvh_origin_limit * 9.5367432e-7f, left_tri, node, sp, ls, tid, c_n, c0, c1, c2,
                                                        nbox, nleaf);
                                if (node == kNone || c_n == kPointCacheBad) {
                                    sp = 0;
                                    if (c_n <= kPointCache) {
                                        if (c_n > 0u) { lq[(size_t)0 * kBvhThreads + tid] = c0 | kQueueTri; lq[(size_t)(kMeshQueue + 0) * kBvhThreads + tid] = 0u; }
                                        if (c_n > 1u) { lq[(size_t)1 * kBvhThreads + tid] = c1 | kQueueTri; lq[(size_t)(kMeshQueue + 1) * kBvhThreads + tid] = 0u; }
                                        if (c_n > 2u) { lq[(size_t)2 * kBvhThreads + tid] = c2 | kQueueTri; lq[(size_t)(kMeshQueue + 2) * kBvhThreads + tid] = 0u; }
                                        qcnt = c_n;
                                        node = kNone;
                                        state = S_FIN;
                                    } else {
                                        qcnt = 0;
                                        node = sv.bvh_root & ~kBvhFlatNode;
                                        pend.n = 0u;
                                        state = S_TRAV;
                                    }
                                }
                            }
                        }
                    }
                    const bool trav = state == S_TRAV;
                    const bool has_leaf = trav && pend.n != 0u, can_open = trav && pend.n == 0u && node != kNone;
                    const uint32_t n_leaf = (uint32_t)__popcll(__ballot(has_leaf)), n_open = (uint32_t)__popcll(__ballot(can_open));
                    if (n_leaf != 0u && n_leaf >= n_open) {         // the lanes with noted leaves read one each (2:1 / 1:2 / 4:1 measured: no better)
                        if (has_leaf) {
                            bool ok;
                            if constexpr (PLAIN == 2) ok = qleaf_read(ma, tpar, pend, qcnt, overflow, best_up, lq, tid, nleaf);
                            else ok = jleaf_read(ma, sr, tpar, pend, qcnt, overflow, best_up, lq, tid, nleaf);
                            if (!ok) state = S_FLUSH;
                        }
                    } else if (n_open != 0u) {                      // the others open their next node
                        if (can_open) {
                            if constexpr (PLAIN == 2) qnode_open<SPILL, STACKN>(nodes, q, node, sp, overflow, best_up, pend, ls, tid, spill, spill_entries,
                                                                                    spill_stride, glane, nbox);
                            else jnode_open<SPILL, STACKN>(nodes, q, node, sp, overflow, best_up, pend, ls, tid, spill, spill_entries,
                                                               spill_stride, glane, nbox);
                        }
                    }
                    if (state == S_TRAV && pend.n == 0u && node == kNone) state = S_FIN;
                    const unsigned long long m_trav = __ballot(state == S_TRAV || state == S_POINT);
                    const uint32_t waiting = (uint32_t)__popcll(__ballot(state == S_FIN || state == S_FLUSH)) +
                                             (queue_empty ? 0u : (uint32_t)__popcll(__ballot(state == S_IDLE)));
                    if (waiting >= thresh || m_trav == 0ull) break;
                }
            } else
            for (;;) {
                if (state == S_TRAV) {
                    if (!mesh_step<SPILL, PLAIN, STACKN>(nodes, ma, q, sr, tpar, nd, node, sp, qcnt, overflow, best_up, resume, resume_node, ls, lq, tid,
                                          spill, spill_entries, spill_stride, glane, nbox, nleaf))
                        state = S_FLUSH;
                    else if (node == kNone) state = S_FIN;
                }
                const unsigned long long m_trav = __ballot(state == S_TRAV);
                const uint32_t waiting = (uint32_t)__popcll(__ballot(state == S_FIN || state == S_FLUSH)) +
                                         (queue_empty ? 0u : (uint32_t)__popcll(__ballot(state == S_IDLE)));
                if (waiting >= thresh || m_trav == 0ull) break;
            }
        }
        { RTX_MSTAMP(t_m4) RTX_MTIME(12, t_m3, t_m4) }                        // cycles: the traversal loop
    }
#if defined(RTX_LAB) && defined(RTX_MESH_PROFILE)
    exact = rtx_prof;
#endif
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

#ifndef RTX_BVH_MESH_THRESH
#define RTX_BVH_MESH_THRESH 16
#endif

uint32_t bvh_mesh_spill_entries(const SceneView &sv)
{
    const uint32_t need = 3u * sv.bvh_depth + 2u;       // a 4-wide node pushes at most 3 entries per level
    return need > (uint32_t)kMeshStack ? need - (uint32_t)kMeshStack : 0u;
}

size_t bvh_mesh_spill_bytes(const SceneView &sv, int n_cus)
{
    return (size_t)bvh_mesh_spill_entries(sv) * (size_t)n_cus * kMeshWaves * kBvhThreads * sizeof(uint32_t);
}

static hipError_t launch_mesh(const SceneView *d_sv, const SceneView &sv, const RowsView *d_rv, const RowsView &rv, double *samples,
                              Counters *counters, unsigned long long *work_counter, uint32_t *spill, int n_cus, hipStream_t stream,
                              const MeshRaySource *src)
{
    const uint64_t want = (rv.n_rays + kBvhThreads - 1) / kBvhThreads;
    const bool wide = mesh_wide(src != nullptr, (sv.bvh_flags & 4u) != 0u ? 2 : 0);      // fewer, fatter workgroups
    const uint64_t cap = (uint64_t)n_cus * (wide ? kMeshWavesQ : kMeshWaves);
    const uint32_t blocks = (uint32_t)(want < cap ? want : cap);
    if (blocks == 0) return hipSuccess;
    // lanes that wait for the f64 phase before it runs (tuning knob).  Fed from a queue of bounced rays the kernel holds no
    // primary rays, whose long walks made early phases pay; measured there (C3 / C5 band, the whole wavefront launch in ms):
    // 8: 52.5 / 70.0, 16: 46.9 / 66.7, 24: 44.5 / 66.0, 32: 43.2 / 64.7, 40: 43.1 / 65.2, 48: 43.8 / 69.4, 64: 59.0 / 135;
    // generating its own rays 16 stays (C5 band alone: 96 ms at 16, 108 at 32), and so it does for a joint tree (240k axis-aligned
    // faces from the queue: 222 ms at 16, 233 at 32)
    const uint32_t thresh = [&] {
        const uint32_t t = (sv.tuning >> RTX_TUNE_THRESH_SHIFT) & 127u;
        // (a joint tree from the queue, re-measured in round 4 with level 0 over tile lists: J1 16 / 24 / 32: 39.1 / 36.7 / 36.7 ms; 240k
        //  axis-aligned faces + 2k spheres 16 / 20 / 24 / 32: 181 / 178 / 178 / 182; the faces alone 16 / 24: 160 / 164 -> 24)
        long v = t ? (long)t : (src ? ((sv.bvh_flags & 4u) != 0u ? 2 * RTX_BVH_MESH_THRESH : (3 * RTX_BVH_MESH_THRESH) / 2) : RTX_BVH_MESH_THRESH);
        return (uint32_t)(v < 1 ? 1 : (v > 64 ? 64 : v));
    }();
    LeafArrays la;
    la.sphere_f32 = sv.bvh_leaf_f32; la.sphere_prims = sv.bvh_prims; la.spheres = sv.spheres; la.sphere_ids = sv.sphere_id;
    la.tri_f32 = sv.tri_f32; la.tri_fidx = sv.tri_fidx; la.tris = sv.tris;
    MeshArrays ma;
    ma.sphere_cr = sv.bvh_leaf_cr; ma.sphere_prims = sv.bvh_prims; ma.tri_f32 = sv.tri_f32; ma.tri_geo = sv.tri_geo;
    const uint32_t need_stack = 3u * sv.bvh_depth + 2u, lds_stack = (uint32_t)(wide ? kMeshStackQ : kMeshStack);
    const uint32_t spill_entries = spill && need_stack > lds_stack ? need_stack - lds_stack : 0u;
    // 0: a joint tree; 1: nothing but (x, y)-footprint triangles; 2: ... with the 64-byte nodes (RTX_TUNE_NO_QNODES: A/B runs)
#ifdef RTX_LAB
    const bool no_q = (sv.tuning & RTX_TUNE_NO_QNODES) != 0u;
    const int plain = (sv.bvh_flags & 4u) == 0u ? 0 : ((sv.bvh_flags & 8u) != 0u && !no_q ? 2 : 1);
#else
    // (the product holds instances 0 and 2: rtx_api.hip clears flag 4 of a pure footprint tree without 64-byte nodes)
    if ((sv.bvh_flags & 12u) == 4u) return hipErrorInvalidValue;
    const int plain = (sv.bvh_flags & 4u) == 0u ? 0 : 2;
#endif
    void (*kernel)(const SceneView *, const RowsView *, double *, Counters *, unsigned long long *, const float4 *, const LeafArrays,
                   const MeshArrays, uint32_t *, uint32_t, uint32_t, const MeshRaySource) = nullptr;
    const bool deep = spill_entries != 0u;
    if (src) {
        if (plain == 2) kernel = deep ? trace_bvh_mesh_kernel<true, 2, true> : trace_bvh_mesh_kernel<false, 2, true>;
#ifdef RTX_LAB
        else if (plain == 1) kernel = deep ? trace_bvh_mesh_kernel<true, 1, true> : trace_bvh_mesh_kernel<false, 1, true>;
#endif
        else kernel = deep ? trace_bvh_mesh_kernel<true, 0, true> : trace_bvh_mesh_kernel<false, 0, true>;
    } else if (plain == 2) kernel = deep ? trace_bvh_mesh_kernel<true, 2, false> : trace_bvh_mesh_kernel<false, 2, false>;
#ifdef RTX_LAB
    else if (plain == 1) kernel = deep ? trace_bvh_mesh_kernel<true, 1, false> : trace_bvh_mesh_kernel<false, 1, false>;
#endif
    else kernel = deep ? trace_bvh_mesh_kernel<true, 0, false> : trace_bvh_mesh_kernel<false, 0, false>;
    const float4 *nodes = plain == 2 ? reinterpret_cast<const float4 *>(sv.bvh_qnodes) : reinterpret_cast<const float4 *>(sv.bvh_nodes);
    MeshRaySource none{};
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kBvhThreads), 0, stream, d_sv, d_rv, samples, counters, work_counter,
                       nodes, la, ma, spill, spill_entries, thresh, src ? *src : none);
    return hipGetLastError();
}

hipError_t launch_trace_bvh_mesh(const SceneView *d_sv, const SceneView &sv, const RowsView *d_rv, const RowsView &rv,
                                 double *samples, Counters *counters, unsigned long long *work_counter, uint32_t *spill, int n_cus,
                                 hipStream_t stream)
{
    return launch_mesh(d_sv, sv, d_rv, rv, samples, counters, work_counter, spill, n_cus, stream, nullptr);
}

hipError_t launch_trace_bvh_mesh_from_queue(const SceneView *d_sv, const SceneView &sv, const RowsView *d_rv, const RowsView &rv,
                                            double *samples, Counters *counters, unsigned long long *head, const MeshRaySource &src,
                                            uint32_t *spill, int n_cus, hipStream_t stream)
{
    return launch_mesh(d_sv, sv, d_rv, rv, samples, counters, head, spill, n_cus, stream, &src);
}

}  // namespace rtx
