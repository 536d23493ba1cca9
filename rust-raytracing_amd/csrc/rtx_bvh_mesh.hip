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
#include "rtx_traverse.h"

#include <cstdlib>

namespace rtx {

constexpr int kMeshQueue = 6;                // live candidates per lane: {entry, t_lo}
#ifndef RTX_MESH_WAVES
#define RTX_MESH_WAVES 4
#endif
#ifndef RTX_MESH_PIPE
#define RTX_MESH_PIPE 0
#endif
constexpr uint32_t kMeshWaves = RTX_MESH_WAVES;   // waves per SIMD (= workgroups per CU)
constexpr bool kMeshPipe = RTX_MESH_PIPE != 0;    // request the next node before reading the current node's leaf records
constexpr int kMeshStack = (RTX_MESH_WAVES <= 4 ? 39 : 160 / RTX_MESH_WAVES) - 1 - 2 * kMeshQueue;   // LDS stack entries per lane: (entries + 1 sink row + 2 * kMeshQueue) KB per workgroup
// f32 bounds of Triangle::distance for one tree triangle (footprint in the (x, y) plane).
//   A  = {n.xyz, n.(v0 - centre)}                       (the filter record's first half)
//   g0 = {v0.x, v0.y (relative to the centre), m00, m01}     g1 = {m10, m11, n.v0 (absolute), -}
// where (a, b) = M (q - v0)_xy solves a r + b s = q - v0 in the rows Triangle::contains reads (triangle.rs:55-100).
// With u = 2^-24 and S >= every coordinate magnitude relative to the centre (tri_filter_from_ray's S):
//   dn = n.d        |dn^ - dn| <= 8u             nv = n.(v0 - p)      |nv^ - nv| <= 16uS
//   t = |nv / dn|  in  [ (|nv^| - 16uS)+ / (|dn^| + 8u),  (|nv^| + 16uS) / (|dn^| - 8u) ]        (the latter needs |dn^| > 8u)
//   q = p + d t    per coordinate within  |d_k| (t_hi - t_lo)/2 + 6u(S + t_hi)  of  p_k + d_k (t_lo + t_hi)/2
//   a, b           within  (|m_k0| e_w0 + |m_k1| e_w1)(1 + 4u) + 4u(|m_k0 w0| + |m_k1 w1|),   e_w = e_q + 2uS
// The reference's own f64 roundings (1e-16 times the conditioning of the projection, which the upload bounds by 1e6
// for a triangle in the tree) are far inside these margins.  Returns t_lo (a lower bound of the distance of ANY hit the
// reference reports for this triangle) and sets t_hi = +inf unless the hit is certain.
__device__ __forceinline__ float tri_bounds(const float4 A, const float4 g0, const float4 g1, const TriFilterParams &f, float &thi)
{
    const float u = 5.9604645e-8f;
    const float S = f.A * 262144.0f * (1.0f + 4.0f * u);                 // f.A = 64uS, rounded once
    const float dn = __builtin_fmaf(A.x, f.dx, __builtin_fmaf(A.y, f.dy, A.z * f.dz));
    const float nv = __builtin_fmaf(A.x, f.npx, __builtin_fmaf(A.y, f.npy, __builtin_fmaf(A.z, f.npz, A.w)));
    const float N = __builtin_fabsf(nv), D = __builtin_fabsf(dn);
    const float e_nv = 16.0f * u * S, e_dn = 8.0f * u;
    const float tlo = fmaxf(N - e_nv, 0.0f) / (D + e_dn) * (1.0f - 4.0f * u);
    thi = __builtin_inff();
    if (D > 4.0f * e_dn && tlo > 0.0f) {
        const float th = (N + e_nv) / (D - e_dn) * (1.0f + 4.0f * u);
        const bool cull_ok = (g1.z - dn) > 2.0f * u * __builtin_fabsf(g1.z) + 2.0f * e_dn;       // n.(v0 - dir) >= 0 for certain (triangle.rs:115)
        const float tm = 0.5f * (tlo + th), ht = 0.5f * (th - tlo) * (1.0f + 4.0f * u) + u * th;
        const float qx = __builtin_fmaf(f.dx, tm, -f.npx), qy = __builtin_fmaf(f.dy, tm, -f.npy);
        const float eq0 = 6.0f * u * (S + th) + 2.0f * u * S;
        const float ewx = __builtin_fmaf(__builtin_fabsf(f.dx), ht, eq0), ewy = __builtin_fmaf(__builtin_fabsf(f.dy), ht, eq0);
        const float wx = qx - g0.x, wy = qy - g0.y;
        const float a = __builtin_fmaf(g0.z, wx, g0.w * wy), b = __builtin_fmaf(g1.x, wx, g1.y * wy);
        const float ea = (__builtin_fabsf(g0.z) * ewx + __builtin_fabsf(g0.w) * ewy) * (1.0f + 4.0f * u) +
                         4.0f * u * (__builtin_fabsf(g0.z * wx) + __builtin_fabsf(g0.w * wy));
        const float eb = (__builtin_fabsf(g1.x) * ewx + __builtin_fabsf(g1.y) * ewy) * (1.0f + 4.0f * u) +
                         4.0f * u * (__builtin_fabsf(g1.x * wx) + __builtin_fabsf(g1.y * wy));
        const bool inside = (a - ea >= 0.0f) && (b - eb >= 0.0f) && (a + b + ea + eb + 4.0f * u <= 1.0f);
        if (cull_ok && inside && th < __builtin_inff()) thi = th;        // (a NaN anywhere fails the comparisons: not certain)
    }
    return tlo;
}

// Room for `need` more queue entries?  Drops the entries a later bound has overtaken first.
__device__ __forceinline__ bool mesh_queue_room(uint32_t *lds_q, uint32_t tid, uint32_t &qcnt, float best_up, uint32_t need)
{
    if (qcnt + need <= (uint32_t)kMeshQueue) return true;
    uint32_t w = 0;
#pragma unroll
    for (int e = 0; e < kMeshQueue; ++e) {
        const uint32_t ie = lds_q[(size_t)e * kBvhThreads + tid];
        const uint32_t te = lds_q[(size_t)(kMeshQueue + e) * kBvhThreads + tid];
        if ((uint32_t)e < qcnt && __uint_as_float(te) <= best_up) {
            lds_q[(size_t)w * kBvhThreads + tid] = ie;
            lds_q[(size_t)(kMeshQueue + w) * kBvhThreads + tid] = te;
            w += 1;
        }
    }
    qcnt = w;
    return qcnt + need <= (uint32_t)kMeshQueue;
}

struct MeshArrays {                          // kernel arguments (global address space)
    const float4 *sphere_cr;                 // per sphere leaf entry: {c - centre, |r|}
    const uint32_t *sphere_prims;
    const float4 *tri_f32;                   // two per triangle filter record
    const float4 *tri_geo;                   // two per tree record: tri_bounds' g0, g1
};

// PLAIN: the tree holds nothing but triangles with (x, y) footprints (C3, C5: no spheres, no faces solved in another
// plane), so every node is a footprint node: 96 of its 128 bytes, a two-slab test, triangle leaves only -- the step then
// needs neither the 3-D test nor the sphere bounds nor their registers.
template <bool PLAIN> struct MeshNode { static constexpr int n = PLAIN ? 6 : 8; };

// The 128 bytes of wide node `idx` (a footprint node uses the first 96).
template <bool PLAIN>
__device__ __forceinline__ void mesh_load_node(const float4 *__restrict__ nodes, uint32_t idx, float4 (&nd)[MeshNode<PLAIN>::n])
{
    const float4 *np = nodes + 8 * (size_t)(idx & ~kBvhFlatNode);
#pragma unroll
    for (int c = 0; c < 6; ++c) nd[c] = np[c];
    if constexpr (!PLAIN) {
        if (!(idx & kBvhFlatNode)) { nd[6] = np[6]; nd[7] = np[7]; }
    }
}

// One traversal step of one lane, f32 only (RAY = Ray64: the slab test in f64, for origins far outside the scene),
// software-pipelined: on entry `nd` holds the data of the node to open; the step tests its children, orders and pushes
// the interior ones, REQUESTS THE NEXT NODE, and only then reads the leaf records of the current one, so the two dependent
// fetches of a step are in flight together; on exit `node` / `nd` are the next node and its data.
// Returns false when the lane has to wait for the exact tests of what its queue holds (the queue cannot take the next
// leaf's records): `resume` then holds the leaf children still to be read and `resume_node` the node they belong to; the
// caller flushes, reloads nd for resume_node and calls again -- that call reads only those children and moves on to `node`.
template <bool SPILL, bool PLAIN, class RAY>
__device__ __forceinline__ bool mesh_step(const float4 *__restrict__ nodes, const MeshArrays &ma, const RAY &q, const SphereRay &sr,
                                          const TriFilterParams &tpar, float4 (&nd)[MeshNode<PLAIN>::n], uint32_t &node, uint32_t &sp, uint32_t &qcnt,
                                          bool &overflow, float &best_up, uint32_t &resume, uint32_t &resume_node, uint32_t *lds_stack,
                                          uint32_t *lds_q, uint32_t tid, uint32_t *__restrict__ spill, uint32_t spill_entries,
                                          size_t spill_stride, size_t glane, uint32_t &nbox, uint32_t &nleaf)
{
    const uint32_t cur = resume != 0u ? resume_node : node;
    if constexpr (!kMeshPipe) mesh_load_node<PLAIN>(nodes, cur, nd);
    uint32_t lnk[4], cnt[4];
    if (PLAIN || (cur & kBvhFlatNode)) {
        lnk[0] = __float_as_uint(nd[4].x); lnk[1] = __float_as_uint(nd[4].y); lnk[2] = __float_as_uint(nd[4].z); lnk[3] = __float_as_uint(nd[4].w);
        cnt[0] = __float_as_uint(nd[5].x); cnt[1] = __float_as_uint(nd[5].y); cnt[2] = __float_as_uint(nd[5].z); cnt[3] = __float_as_uint(nd[5].w);
    } else if constexpr (!PLAIN) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { lnk[c] = __float_as_uint(nd[c].w); cnt[c] = __float_as_uint(nd[4 + c].w); }
    }
    uint32_t leafmask = 0, next = node;
    if (resume == 0u) {
        float tc[4];
        if (PLAIN || (cur & kBvhFlatNode)) {
#pragma unroll
            for (int c = 0; c < 4; ++c) tc[c] = rect_entry32(nd[c], q, best_up);
        } else if constexpr (!PLAIN) {
#pragma unroll
            for (int c = 0; c < 4; ++c) tc[c] = box_entry32(nd[c], nd[4 + c], q, best_up);
        }
        nbox += 4;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (tc[c] < __builtin_inff() && cnt[c] - 1u < 0x1FFFFu) leafmask |= 1u << c;      // neither interior (0) nor empty (~0)
        // interior children still in reach, nearest first; the farther ones go to the stack
        float key[4];
        uint32_t kl[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { key[c] = cnt[c] == 0u ? tc[c] : __builtin_inff(); kl[c] = lnk[c]; }
#define RTX_CSWAP(i, j) { if (key[j] < key[i]) { float tk = key[i]; key[i] = key[j]; key[j] = tk; uint32_t tl = kl[i]; kl[i] = kl[j]; kl[j] = tl; } }
        RTX_CSWAP(0, 1) RTX_CSWAP(2, 3) RTX_CSWAP(0, 2) RTX_CSWAP(1, 3) RTX_CSWAP(1, 2)
#undef RTX_CSWAP
        const uint32_t npush = (key[1] < __builtin_inff() ? 1u : 0u) + (key[2] < __builtin_inff() ? 1u : 0u) +
                               (key[3] < __builtin_inff() ? 1u : 0u);
        if (sp + 3u <= (uint32_t)kMeshStack) {
#pragma unroll
            for (uint32_t i = 1; i <= 3; ++i) {
                const uint32_t row = i <= npush ? sp + npush - i : (uint32_t)kMeshStack;      // (row kMeshStack = the sink)
                lds_stack[(size_t)row * kBvhThreads + tid] = kl[i];
            }
            sp += npush;
        } else {
#define RTX_PUSH(v)                                                                                           \
            {                                                                                                 \
                if (sp < (uint32_t)kMeshStack) { lds_stack[(size_t)sp * kBvhThreads + tid] = (v); sp += 1; } \
                else if (SPILL && sp - (uint32_t)kMeshStack < spill_entries) {                                \
                    spill[(size_t)(sp - (uint32_t)kMeshStack) * spill_stride + glane] = (v); sp += 1;         \
                } else overflow = true;                                                                       \
            }
            if (key[3] < __builtin_inff()) RTX_PUSH(kl[3])
            if (key[2] < __builtin_inff()) RTX_PUSH(kl[2])
            if (key[1] < __builtin_inff()) RTX_PUSH(kl[1])
#undef RTX_PUSH
        }
        next = key[0] < __builtin_inff() ? kl[0] : kNone;
        if (next == kNone && sp != 0u) {
            sp -= 1;
            next = (!SPILL || sp < (uint32_t)kMeshStack) ? lds_stack[(size_t)sp * kBvhThreads + tid]
                                                         : spill[(size_t)(sp - (uint32_t)kMeshStack) * spill_stride + glane];
        }
    } else {                                   // after a flush: only the leaf children that were not read yet; `node` is already the next one
        leafmask = resume;
        resume = 0u;
    }
    // ---- request the next node; its fetch overlaps the leaf-record fetches below
    if constexpr (kMeshPipe) { if (next != kNone) mesh_load_node<PLAIN>(nodes, next, nd); }
    // ---- the leaf children of the current node
    bool flush = false;
    while (leafmask != 0u) {
        const uint32_t c = (uint32_t)__builtin_ctz(leafmask);
        const uint32_t first = c == 0 ? lnk[0] : (c == 1 ? lnk[1] : (c == 2 ? lnk[2] : lnk[3]));
        const uint32_t count = c == 0 ? cnt[0] : (c == 1 ? cnt[1] : (c == 2 ? cnt[2] : cnt[3]));
        const uint32_t n = count & 0xFFFFu;
        if (!mesh_queue_room(lds_q, tid, qcnt, best_up, n)) {
            if (qcnt != 0u) { resume = leafmask; resume_node = cur; flush = true; break; }     // exact tests of what the queue holds first
            overflow = true;                   // a single leaf with more records than the queue has entries (a tuning build)
        }
        leafmask &= leafmask - 1u;
        if (PLAIN || (count & kBvhTriLeaf)) {
            for (uint32_t k = 0; k < n; k += 2u) {
                const float4 *rp = ma.tri_f32 + 2 * (size_t)(first + k);
                const float4 A0 = rp[0], B0 = rp[1], A1 = rp[2], B1 = rp[3];      // (padded: the second pair may belong to the next leaf)
                uint32_t m = (int)tri_filter_sign(A0, B0, tpar) >= 0 ? 1u : 0u;
                if (k + 1u < n && (int)tri_filter_sign(A1, B1, tpar) >= 0) m |= 2u;
                while (m != 0u) {
                    const uint32_t j = (uint32_t)__builtin_ctz(m);
                    m &= m - 1u;
                    const float4 *gp = ma.tri_geo + 2 * (size_t)(first + k + j);
                    const float4 g0 = gp[0], g1 = gp[1];
                    float thi;
                    const float tlo = tri_bounds(j == 0u ? A0 : A1, g0, g1, tpar, thi);
                    if (tlo <= best_up) {
                        best_up = fminf(best_up, thi);
                        if (qcnt < (uint32_t)kMeshQueue) {
                            lds_q[(size_t)qcnt * kBvhThreads + tid] = (first + k + j) | kQueueTri;
                            lds_q[(size_t)(kMeshQueue + qcnt) * kBvhThreads + tid] = __float_as_uint(tlo);
                            qcnt += 1;
                        }
                    }
                }
            }
        } else if constexpr (!PLAIN) {
            for (uint32_t k = 0; k < n; ++k) {
                const float4 rec = ma.sphere_cr[first + k];                              // {c - centre, r}: bvh_traverse_spheres' bounds
                const float ox = rec.x - sr.px, oy = rec.y - sr.py, oz = rec.z - sr.pz;
                const float b = __builtin_fmaf(ox, sr.dx, __builtin_fmaf(oy, sr.dy, oz * sr.dz));
                const float lx = __builtin_fmaf(-b, sr.dx, ox), ly = __builtin_fmaf(-b, sr.dy, oy), lz = __builtin_fmaf(-b, sr.dz, oz);
                const float l2 = __builtin_fmaf(lx, lx, __builtin_fmaf(ly, ly, lz * lz));
                const float Dl = __builtin_fmaf(rec.w, rec.w, -l2);
                const float G = __builtin_fmaf(sr.Kg, rec.w, sr.c0);
                const float Dp = Dl + G;
                if (Dp >= 0.0f) {
                    const float tlo = b - __builtin_amdgcn_sqrtf(Dp) * (1.0f + 4.76837158e-7f) - sr.K;
                    const float Dm = Dl - G;
                    const float thi = Dm > 0.0f ? b - __builtin_amdgcn_sqrtf(Dm) * (1.0f - 4.76837158e-7f) + sr.K : __builtin_inff();
                    if (tlo <= best_up && !(thi < 0.0f)) {
                        if (tlo > sr.K) best_up = fminf(best_up, thi);
                        if (qcnt < (uint32_t)kMeshQueue) {
                            lds_q[(size_t)qcnt * kBvhThreads + tid] = ma.sphere_prims[first + k];
                            lds_q[(size_t)(kMeshQueue + qcnt) * kBvhThreads + tid] = __float_as_uint(tlo);
                            qcnt += 1;
                        }
                    }
                }
            }
        }
        nleaf += n;
    }
    node = next;
    return !flush;
}

template <bool SPILL, bool PLAIN>
__global__ __launch_bounds__(kBvhThreads, kMeshWaves) void trace_bvh_mesh_kernel(const SceneView *__restrict__ svp,
                                                                                 const RowsView *__restrict__ rvp,
                                                                                 double *__restrict__ samples, Counters *__restrict__ ctr,
                                                                                 unsigned long long *__restrict__ work_counter,
                                                                                 const float4 *__restrict__ nodes, const LeafArrays la,
                                                                                 const MeshArrays ma, uint32_t *__restrict__ spill,
                                                                                 uint32_t spill_entries, uint32_t thresh)
{
    const SceneView &sv = *svp;
    const RowsView &rv = *rvp;
    __shared__ uint32_t lds_stack[kMeshStack + 1][kBvhThreads];        // + the sink row of the branch-free pushes
    __shared__ uint32_t lds_q[2 * kMeshQueue][kBvhThreads];            // candidate entries, then their t_lo
    uint32_t *const ls = &lds_stack[0][0];
    uint32_t *const lq = &lds_q[0][0];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    const size_t spill_stride = (size_t)gridDim.x * kBvhThreads, glane = (size_t)blockIdx.x * kBvhThreads + tid;
    const uint32_t bounce_limit = sv.max_bounces >= 0xFFFFFFF0ull ? 0xFFFFFFF0u : (uint32_t)sv.max_bounces + 1u;
    enum : uint32_t { S_IDLE = 0, S_TRAV = 1, S_FIN = 2, S_SETUP = 3, S_FLUSH = 4 };

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
    bool overflow = false, tree_used = false;
    uint32_t ridx = 0;
    uint32_t left_tri = kNone;               // the triangle (index in tris[]) the ray has just bounced off, if any
    unsigned long long segs = 0, box_tests = 0, leaf_filters = 0, exact = 0;
    hit_init(h);
    tri_filter_idle(tpar);
    q.ix = q.iy = q.iz = q.nx = q.ny = q.nz = 0.f;
    sr.px = sr.py = sr.pz = sr.dx = sr.dy = sr.dz = sr.Kg = sr.K = 0.f; sr.c0 = __builtin_inff();

    for (;;) {
        // ================= the f64 phase (entered when the inner loop below finds it due) =================
        if (__ballot(state != S_IDLE) == 0ull && queue_empty) break;       // wave-uniform: nothing live, nothing left to take
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
                wave_end = base + rv.grab < rv.n_rays ? base + rv.grab : rv.n_rays;
                if (base >= rv.n_rays) { queue_empty = true; wave_next = wave_end = 0; break; }
            }
            if (state == S_IDLE) {
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
                    left_tri = kNone;
                    state = S_SETUP;
                }
            }
            const unsigned long long taken = (unsigned long long)__popcll(idle_mask);
            wave_next = wave_next + taken < wave_end ? wave_next + taken : wave_end;
        }
        // ---- set up the next segment
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
                make_ray32(r.pos, rx.dirn, (double)sv.bvh_inv_max, q);
                node = sv.bvh_root;              // wide node 0 (flagged when it is a footprint node)
                state = S_TRAV;
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
                        if (mesh_step<SPILL, PLAIN>(nodes, ma, q64, sr, tpar, nd, node, sp, qcnt, overflow, best_up, resume, resume_node, ls, lq,
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
        // ================= traversal steps (f32 only) until the f64 phase is due again =================
        {
            float4 nd[MeshNode<PLAIN>::n];    // the node each TRAV lane opens next (the prefetched copy does not outlive this loop)
            if constexpr (kMeshPipe) { if (state == S_TRAV) mesh_load_node<PLAIN>(nodes, resume != 0u ? resume_node : node, nd); }
            for (;;) {
                if (state == S_TRAV) {
                    if (!mesh_step<SPILL, PLAIN>(nodes, ma, q, sr, tpar, nd, node, sp, qcnt, overflow, best_up, resume, resume_node, ls, lq, tid,
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
    }
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

hipError_t launch_trace_bvh_mesh(const SceneView *d_sv, const SceneView &sv, const RowsView *d_rv, const RowsView &rv,
                                 double *samples, Counters *counters, unsigned long long *work_counter, uint32_t *spill, int n_cus,
                                 hipStream_t stream)
{
    const uint64_t want = (rv.n_rays + kBvhThreads - 1) / kBvhThreads;
    const uint64_t cap = (uint64_t)n_cus * kMeshWaves;
    const uint32_t blocks = (uint32_t)(want < cap ? want : cap);
    if (blocks == 0) return hipSuccess;
    static const uint32_t thresh = [] {                         // tuning knob
        const char *e = std::getenv("RTX_HIP_BVH_THRESH");
        long v = e && *e ? std::strtol(e, nullptr, 10) : RTX_BVH_MESH_THRESH;
        return (uint32_t)(v < 1 ? 1 : (v > 64 ? 64 : v));
    }();
    LeafArrays la;
    la.sphere_f32 = sv.bvh_leaf_f32; la.sphere_prims = sv.bvh_prims; la.spheres = sv.spheres; la.sphere_ids = sv.sphere_id;
    la.tri_f32 = sv.tri_f32; la.tri_fidx = sv.tri_fidx; la.tris = sv.tris;
    MeshArrays ma;
    ma.sphere_cr = sv.bvh_leaf_cr; ma.sphere_prims = sv.bvh_prims; ma.tri_f32 = sv.tri_f32; ma.tri_geo = sv.tri_geo;
    const uint32_t spill_entries = spill ? bvh_mesh_spill_entries(sv) : 0u;
    const bool plain = (sv.bvh_flags & 4u) != 0u;       // nothing but (x, y)-footprint triangles in the tree
    auto kernel = spill_entries != 0u ? (plain ? trace_bvh_mesh_kernel<true, true> : trace_bvh_mesh_kernel<true, false>)
                                      : (plain ? trace_bvh_mesh_kernel<false, true> : trace_bvh_mesh_kernel<false, false>);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(kBvhThreads), 0, stream, d_sv, d_rv, samples, counters, work_counter,
                       reinterpret_cast<const float4 *>(sv.bvh_nodes), la, ma, spill, spill_entries, thresh);
    return hipGetLastError();
}

}  // namespace rtx
