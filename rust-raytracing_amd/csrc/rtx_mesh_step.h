// rtx_mesh_step.h -- the f32-only traversal step of trees that hold triangles, shared by trace_bvh_mesh_kernel
// (rtx_bvh_mesh.hip: regrouping schedule inside one kernel) and the wavefront kernels (rtx_wavefront.hip: the walk as a
// kernel of its own).  See rtx_bvh_mesh.hip for the scheme; tri_bounds below for the error analysis.
#pragma once

#include "rtx_traverse.h"

namespace rtx {

constexpr int kMeshQueue = 6;                // live candidates per lane: {entry, t_lo}
#ifndef RTX_MESH_PIPE
#define RTX_MESH_PIPE 0
#endif
constexpr bool kMeshPipe = RTX_MESH_PIPE != 0;    // request the next node before reading the current node's leaf records
// f32 bounds of Triangle::distance for one tree triangle.
//   A  = {n.xyz, n.(v0 - centre)}                       (the filter record's first half)
//   g0 = {v0.u, v0.v (relative to the centre), m00, m01}     g1 = {m10, m11, n.v0 (absolute), plane}
// where (u, v) are the two rows Triangle::contains reads -- plane 0: (x, y), 1: (x, z), 2: (y, z) -- and
// (a, b) = M (q - v0)_uv solves a r + b s = q - v0 in those rows (triangle.rs:55-100).
// With u = 2^-24 and S >= every coordinate magnitude relative to the centre (tri_filter_from_ray's S):
//   dn = n.d        |dn^ - dn| <= 8u             nv = n.(v0 - p)      |nv^ - nv| <= 16uS
//   t = |nv / dn|  in  [ (|nv^| - 16uS)+ / (|dn^| + 8u),  (|nv^| + 16uS) / (|dn^| - 8u) ]        (the latter needs |dn^| > 8u)
//   q = p + d t    per coordinate within  |d_k| (t_hi - t_lo)/2 + 6u(S + t_hi)  of  p_k + d_k (t_lo + t_hi)/2
//   a, b           within  (|m_k0| e_w0 + |m_k1| e_w1)(1 + 4u) + 4u(|m_k0 w0| + |m_k1 w1|),   e_w = e_q + 2uS
// The reference's own f64 roundings (1e-16 times the conditioning of the projection, which the upload bounds by 1e6
// for a triangle in the tree) are far inside these margins.  Returns t_lo (a lower bound of the distance of ANY hit the
// reference reports for this triangle; +inf when it certainly reports none: the footprint filter passes every ray whose hit
// point lies in the triangle's bounding RECTANGLE, half of which is outside the triangle) and sets t_hi = +inf unless the
// hit is certain.
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
        // the two coordinates Triangle::contains reads: g1.w = 0: (x, y), 1: (x, z), 2: (y, z)
        const float du = g1.w == 2.0f ? f.dy : f.dx, pu = g1.w == 2.0f ? f.npy : f.npx;
        const float dv = g1.w == 0.0f ? f.dy : f.dz, pv = g1.w == 0.0f ? f.npy : f.npz;
        const float qx = __builtin_fmaf(du, tm, -pu), qy = __builtin_fmaf(dv, tm, -pv);
        const float eq0 = 6.0f * u * (S + th) + 2.0f * u * S;
        const float ewx = __builtin_fmaf(__builtin_fabsf(du), ht, eq0), ewy = __builtin_fmaf(__builtin_fabsf(dv), ht, eq0);
        const float wx = qx - g0.x, wy = qy - g0.y;
        const float a = __builtin_fmaf(g0.z, wx, g0.w * wy), b = __builtin_fmaf(g1.x, wx, g1.y * wy);
        const float ea = (__builtin_fabsf(g0.z) * ewx + __builtin_fabsf(g0.w) * ewy) * (1.0f + 4.0f * u) +
                         4.0f * u * (__builtin_fabsf(g0.z * wx) + __builtin_fabsf(g0.w * wy));
        const float eb = (__builtin_fabsf(g1.x) * ewx + __builtin_fabsf(g1.y) * ewy) * (1.0f + 4.0f * u) +
                         4.0f * u * (__builtin_fabsf(g1.x * wx) + __builtin_fabsf(g1.y * wy));
        const bool inside = (a - ea >= 0.0f) && (b - eb >= 0.0f) && (a + b + ea + eb + 4.0f * u <= 1.0f);
        if (cull_ok && inside && th < __builtin_inff()) thi = th;        // (a NaN anywhere fails the comparisons: not certain)
        // ... and the other way round: (a, b) outside [0, 1] x [0, 1], a + b <= 1 beyond the same margins, or the cull test failing
        // beyond its margin: Triangle::contains / the cull CERTAINLY reject, whatever t in [t_lo, t_hi] is -- not a candidate
        const bool outside = (a + ea < 0.0f) || (b + eb < 0.0f) || (a + b - ea - eb - 4.0f * u > 1.0f);
        const bool culled = (g1.z - dn) < -(2.0f * u * __builtin_fabsf(g1.z) + 2.0f * e_dn);
        if (outside || culled) return __builtin_inff();
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
// PLAIN == 2: ... and the tree has the 64-byte quantised nodes (rtx_bvh.h BvhQNode): 4 requests per node instead of 6.
template <int PLAIN> struct MeshNode { static constexpr int n = PLAIN == 2 ? 4 : PLAIN == 1 ? 6 : 8; };

// entry distance of a quantised child rectangle: plane = o + q * s, t = (plane - origin) * inv = q * (s * inv) + (o * inv + noi);
// s is a power of two, so A = s * inv is exact, B is one fma rounding of a quantity of the size of a distance inside the
// node (covered, like noi's rounding, by the boxes' absolute padding), t one more
// (e: Ray32S's slack, 0 for Ray32 -- it covers the rounding of noi inside B for origins the padding does not; the node's own
// o * inv term is of the size of a distance inside the scene and stays covered by the padding)
__device__ __forceinline__ float qrect_entry(uint32_t qx, uint32_t qy, float Ax, float Bx, float Ay, float By, float best_up, float e)
{
    const float x0 = __builtin_fmaf((float)(qx & 0xFFFFu), Ax, Bx), x1 = __builtin_fmaf((float)(qx >> 16), Ax, Bx);
    const float y0 = __builtin_fmaf((float)(qy & 0xFFFFu), Ay, By), y1 = __builtin_fmaf((float)(qy >> 16), Ay, By);
    const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), 0.0f);
    const float tf = fminf(fmaxf(x0, x1), fmaxf(y0, y1));
    const float tn_lo = __builtin_fmaf(tn, 1.0f - 4.76837158e-7f, -e);
    const float tf_hi = __builtin_fmaf(tf, 1.0f + 4.76837158e-7f, e);
    return (tn_lo <= tf_hi && tn_lo <= best_up) ? tn_lo : __builtin_inff();
}
__device__ __forceinline__ double ray_slack(const Ray64 &) { return 0.0; }
__device__ __forceinline__ float qnode_offset(float o, float inv, float noi) { return __builtin_fmaf(o, inv, noi); }
__device__ __forceinline__ double qnode_offset(float o, double inv, double noi) { return __builtin_fma((double)o, inv, noi); }
__device__ __forceinline__ float qrect_entry(uint32_t qx, uint32_t qy, double Ax, double Bx, double Ay, double By, float best_up, double)
{
    const double x0 = __builtin_fma((double)(qx & 0xFFFFu), Ax, Bx), x1 = __builtin_fma((double)(qx >> 16), Ax, Bx);
    const double y0 = __builtin_fma((double)(qy & 0xFFFFu), Ay, By), y1 = __builtin_fma((double)(qy >> 16), Ay, By);
    const double tn = fmax(fmax(fmin(x0, x1), fmin(y0, y1)), 0.0);
    const double tf = fmin(fmax(x0, x1), fmax(y0, y1));
    const double tn_lo = tn * (1.0 - 4.76837158e-7), tf_hi = tf * (1.0 + 4.76837158e-7);
    return (tn_lo <= tf_hi && tn_lo <= (double)best_up) ? __double2float_rd(tn_lo) : __builtin_inff();
}

// The 128 bytes of wide node `idx` (a footprint node uses the first 96).
template <int PLAIN>
__device__ __forceinline__ void mesh_load_node(const float4 *__restrict__ nodes, uint32_t idx, float4 (&nd)[MeshNode<PLAIN>::n])
{
    if constexpr (PLAIN == 2) {                   // `nodes` = the BvhQNode array
        const float4 *np = nodes + 4 * (size_t)(idx & ~kBvhFlatNode);
#pragma unroll
        for (int c = 0; c < 4; ++c) nd[c] = np[c];
    } else {
        const float4 *np = nodes + 8 * (size_t)(idx & ~kBvhFlatNode);
#pragma unroll
        for (int c = 0; c < 6; ++c) nd[c] = np[c];
        if constexpr (!PLAIN) {
            if (!(idx & kBvhFlatNode)) { nd[6] = np[6]; nd[7] = np[7]; }
        }
    }
}

// One traversal step of one lane, f32 only (RAY = Ray64: the slab test in f64, for origins far outside the scene),
// software-pipelined: on entry `nd` holds the data of the node to open; the step tests its children, orders and pushes
// the interior ones, REQUESTS THE NEXT NODE, and only then reads the leaf records of the current one, so the two dependent
// fetches of a step are in flight together; on exit `node` / `nd` are the next node and its data.
// Returns false when the lane has to wait for the exact tests of what its queue holds (the queue cannot take the next
// leaf's records): `resume` then holds the leaf children still to be read and `resume_node` the node they belong to; the
// caller flushes, reloads nd for resume_node and calls again -- that call reads only those children and moves on to `node`.
template <bool SPILL, int PLAIN, int STACK, class RAY>
__device__ __forceinline__ bool mesh_step(const float4 *__restrict__ nodes, const MeshArrays &ma, const RAY &q, const SphereRay &sr,
                                          const TriFilterParams &tpar, float4 (&nd)[MeshNode<PLAIN>::n], uint32_t &node, uint32_t &sp, uint32_t &qcnt,
                                          bool &overflow, float &best_up, uint32_t &resume, uint32_t &resume_node, uint32_t *lds_stack,
                                          uint32_t *lds_q, uint32_t tid, uint32_t *__restrict__ spill, uint32_t spill_entries,
                                          size_t spill_stride, size_t glane, uint32_t &nbox, uint32_t &nleaf)
{
    const uint32_t cur = resume != 0u ? resume_node : node;
    if constexpr (!kMeshPipe) mesh_load_node<PLAIN>(nodes, cur, nd);
    uint32_t lnk[4], cnt[4];
    if constexpr (PLAIN == 2) {
        const uint32_t w[4] = { __float_as_uint(nd[3].x), __float_as_uint(nd[3].y), __float_as_uint(nd[3].z), __float_as_uint(nd[3].w) };
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t type = w[c] >> kQNodeShift;
            lnk[c] = w[c] & kQNodeIndexMask;
            cnt[c] = type == 0u ? 0u : (type == kQNodeEmpty ? 0xFFFFFFFFu : (type | kBvhTriLeaf));
        }
    } else if (PLAIN || (cur & kBvhFlatNode)) {
        lnk[0] = __float_as_uint(nd[4].x); lnk[1] = __float_as_uint(nd[4].y); lnk[2] = __float_as_uint(nd[4].z); lnk[3] = __float_as_uint(nd[4].w);
        cnt[0] = __float_as_uint(nd[5].x); cnt[1] = __float_as_uint(nd[5].y); cnt[2] = __float_as_uint(nd[5].z); cnt[3] = __float_as_uint(nd[5].w);
    } else if constexpr (!PLAIN) {
#pragma unroll
        for (int c = 0; c < 4; ++c) { lnk[c] = __float_as_uint(nd[c].w); cnt[c] = __float_as_uint(nd[4 + c].w); }
    }
    uint32_t leafmask = 0, next = node;
    if (resume == 0u) {
        float tc[4];
        if constexpr (PLAIN == 2) {
            const auto Ax = nd[0].z * q.ix, Ay = nd[0].w * q.iy;                  // (float or double with RAY)
            const uint32_t qx[4] = { __float_as_uint(nd[1].x), __float_as_uint(nd[1].y), __float_as_uint(nd[1].z), __float_as_uint(nd[1].w) };
            const uint32_t qy[4] = { __float_as_uint(nd[2].x), __float_as_uint(nd[2].y), __float_as_uint(nd[2].z), __float_as_uint(nd[2].w) };
            const auto Bxf = qnode_offset(nd[0].x, q.ix, q.nx), Byf = qnode_offset(nd[0].y, q.iy, q.ny);
#pragma unroll
            for (int c = 0; c < 4; ++c) tc[c] = qrect_entry(qx[c], qy[c], Ax, Bxf, Ay, Byf, best_up, ray_slack(q));
        } else if (PLAIN || (cur & kBvhFlatNode)) {
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
        if (sp + 3u <= (uint32_t)STACK) {
#pragma unroll
            for (uint32_t i = 1; i <= 3; ++i) {
                const uint32_t row = i <= npush ? sp + npush - i : (uint32_t)STACK;      // (row STACK = the sink)
                lds_stack[(size_t)row * kBvhThreads + tid] = kl[i];
            }
            sp += npush;
        } else {
#define RTX_PUSH(v)                                                                                           \
            {                                                                                                 \
                if (sp < (uint32_t)STACK) { lds_stack[(size_t)sp * kBvhThreads + tid] = (v); sp += 1; } \
                else if (SPILL && sp - (uint32_t)STACK < spill_entries) {                                \
                    spill[(size_t)(sp - (uint32_t)STACK) * spill_stride + glane] = (v); sp += 1;         \
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
            next = (!SPILL || sp < (uint32_t)STACK) ? lds_stack[(size_t)sp * kBvhThreads + tid]
                                                         : spill[(size_t)(sp - (uint32_t)STACK) * spill_stride + glane];
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
                    if (tlo <= best_up && tlo < __builtin_inff()) {                 // (+inf: certainly no hit -- also while best_up is still +inf)
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

// ---- the step in two halves (64-byte nodes) ---------------------------------------------------------------------------------
// mesh_step opens a node and reads the records of every leaf child the ray enters before it returns: in a wave the leaf
// loop runs as often as the lane with the most leaf children needs, with most lanes idle.  Here a lane that opens a node
// only NOTES its leaf children (their link words, at most 4) and the wave decides per iteration whether the lanes with
// noted leaves read one each or the others open their next node -- whichever group is larger -- so either half runs with
// at least half of the walking lanes.  A lane opens its next node only when it has no leaf left, so 4 slots suffice; a
// noted leaf is read even if the bound has moved past its rectangle since (its records are then rejected by t_lo).
struct MeshPending { uint32_t p0, p1, p2, p3, n; };

template <bool SPILL, int STACK, class RAY>
__device__ __forceinline__ void qnode_open(const float4 *__restrict__ nodes, const RAY &q, uint32_t &node, uint32_t &sp, bool &overflow,
                                           float best_up, MeshPending &pend, uint32_t *lds_stack, uint32_t tid,
                                           uint32_t *__restrict__ spill, uint32_t spill_entries, size_t spill_stride, size_t glane,
                                           uint32_t &nbox)
{
    const float4 *np = nodes + 4 * (size_t)node;
    const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
    const uint32_t w[4] = { __float_as_uint(n3.x), __float_as_uint(n3.y), __float_as_uint(n3.z), __float_as_uint(n3.w) };
    const uint32_t qx[4] = { __float_as_uint(n1.x), __float_as_uint(n1.y), __float_as_uint(n1.z), __float_as_uint(n1.w) };
    const uint32_t qy[4] = { __float_as_uint(n2.x), __float_as_uint(n2.y), __float_as_uint(n2.z), __float_as_uint(n2.w) };
    const auto Ax = n0.z * q.ix, Ay = n0.w * q.iy;
    const auto Bx = qnode_offset(n0.x, q.ix, q.nx), By = qnode_offset(n0.y, q.iy, q.ny);
    float key[4];
    uint32_t kl[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float tc = qrect_entry(qx[c], qy[c], Ax, Bx, Ay, By, best_up, ray_slack(q));
        const uint32_t type = w[c] >> kQNodeShift;
        const bool in = tc < __builtin_inff();
        if (in && type != 0u && type != kQNodeEmpty) {          // a leaf the ray enters: noted, read later
            pend.p3 = pend.p2; pend.p2 = pend.p1; pend.p1 = pend.p0; pend.p0 = w[c];
            pend.n += 1;
        }
        key[c] = type == 0u ? tc : __builtin_inff();
        kl[c] = w[c] & kQNodeIndexMask;
    }
    nbox += 4;
#define RTX_CSWAP(i, j) { if (key[j] < key[i]) { float tk = key[i]; key[i] = key[j]; key[j] = tk; uint32_t tl = kl[i]; kl[i] = kl[j]; kl[j] = tl; } }
    RTX_CSWAP(0, 1) RTX_CSWAP(2, 3) RTX_CSWAP(0, 2) RTX_CSWAP(1, 3) RTX_CSWAP(1, 2)
#undef RTX_CSWAP
    const uint32_t npush = (key[1] < __builtin_inff() ? 1u : 0u) + (key[2] < __builtin_inff() ? 1u : 0u) +
                           (key[3] < __builtin_inff() ? 1u : 0u);
    if (sp + 3u <= (uint32_t)STACK) {
#pragma unroll
        for (uint32_t i = 1; i <= 3; ++i) {
            const uint32_t row = i <= npush ? sp + npush - i : (uint32_t)STACK;      // (row STACK = the sink)
            lds_stack[(size_t)row * kBvhThreads + tid] = kl[i];
        }
        sp += npush;
    } else {
#define RTX_PUSH(v)                                                                                           \
        {                                                                                                     \
            if (sp < (uint32_t)STACK) { lds_stack[(size_t)sp * kBvhThreads + tid] = (v); sp += 1; }           \
            else if (SPILL && sp - (uint32_t)STACK < spill_entries) {                                         \
                spill[(size_t)(sp - (uint32_t)STACK) * spill_stride + glane] = (v); sp += 1;                  \
            } else overflow = true;                                                                           \
        }
        if (key[3] < __builtin_inff()) RTX_PUSH(kl[3])
        if (key[2] < __builtin_inff()) RTX_PUSH(kl[2])
        if (key[1] < __builtin_inff()) RTX_PUSH(kl[1])
#undef RTX_PUSH
    }
    node = key[0] < __builtin_inff() ? kl[0] : kNone;
    if (node == kNone && sp != 0u) {
        sp -= 1;
        node = (!SPILL || sp < (uint32_t)STACK) ? lds_stack[(size_t)sp * kBvhThreads + tid]
                                                : spill[(size_t)(sp - (uint32_t)STACK) * spill_stride + glane];
    }
}

// Reads the noted leaf on top: its records are filtered and bounded as in mesh_step.  Returns false (the leaf stays noted)
// when the queue cannot take its records: the caller has the queue's candidates tested exactly first.
__device__ __forceinline__ bool qleaf_read(const MeshArrays &ma, const TriFilterParams &tpar, MeshPending &pend, uint32_t &qcnt,
                                           bool &overflow, float &best_up, uint32_t *lds_q, uint32_t tid, uint32_t &nleaf)
{
    const uint32_t link = pend.p0;
    const uint32_t n = link >> kQNodeShift, first = link & kQNodeIndexMask;
    if (!mesh_queue_room(lds_q, tid, qcnt, best_up, n)) {
        if (qcnt != 0u) return false;
        overflow = true;                       // (a leaf of more records than the queue has entries: the upload never builds one)
    }
    pend.p0 = pend.p1; pend.p1 = pend.p2; pend.p2 = pend.p3;
    pend.n -= 1;
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
            if (tlo <= best_up && tlo < __builtin_inff()) {
                best_up = fminf(best_up, thi);
                if (qcnt < (uint32_t)kMeshQueue) {
                    lds_q[(size_t)qcnt * kBvhThreads + tid] = (first + k + j) | kQueueTri;
                    lds_q[(size_t)(kMeshQueue + qcnt) * kBvhThreads + tid] = __float_as_uint(tlo);
                    qcnt += 1;
                }
            }
        }
    }
    nleaf += n;
    return true;
}

// ---- a segment that starts ON the triangle it has just left: no walk ----------------------------------------------------------
// A ray that bounces off a triangle of a mesh re-hits that triangle at |t| ~ 1e-16 (no epsilon anywhere in the
// reference), bounces again from the same point, re-hits again ... until the bounce limit: 6 of C5's 7.2 segments per ray and
// 2.8 of C3's 3.8 are such self-hits, and every one of them used to walk the tree -- a point location in a tree of 2-D
// footprints that overlap ~12-fold: ~85 node visits, ~60 filter records, 10 KB of memory traffic (C5: 137 GB per launch).
// The self-hit itself is found exactly in the set-up phase (the triangle the ray left is tested first).  What the walk adds is the
// proof that no OTHER triangle is at most as far: t_k <= t_self.  That proof does not depend on the direction:
//     a triangle k the reference reports at distance t_k has the point q = p + d t_k in its plane and its footprint (Triangle::
//     contains, triangle.rs:37-101: q's projection inside the projected triangle, hence inside its bounding rectangle); |d| = 1,
//     so with t_k <= tau:  dist(p, plane_k) <= tau  and  p.xy within tau of rectangle_k.
// So for an origin p the set  C(p) = { k : |n_k . (v0_k - p)| <= T, p.xy within T of rectangle_k },  T >= tau + roundings, holds
// every triangle that can beat or tie a self-hit with t_self <= tau from p -- whatever the direction.  It is computed ONCE per
// point (mesh_point_step: containment tests only, no slabs, no ordering, no bounds; the lanes that locate a point do so in the
// kernel's traversal loop, a node per iteration, beside the lanes that walk) and kept in three registers (it is {} for a random
// mesh: the triangle itself is excluded); the following segments from that point go straight to their exact tests.
// Roundings, with S >= every coordinate magnitude (tri_filter_from_ray), u = 2^-24: the f32 key identifies p to 2^-23 S, nv is
// evaluated to 16uS = 2^-20 S, rectangles decode to 2^-24 of their coordinates; tau = 2^-20 S = tpar.A / 4 and T = tpar.A = 2^-18 S
// cover them with room to spare (T >= tau + 2 * 2^-23 S * sqrt(3) + 2^-20 S).
constexpr uint32_t kPointCache = 3, kPointCacheBad = 0xFFu;

// One node of the point location: the children whose rectangle holds the point (within Tb) are pushed / entered, the records of
// such a leaf are tested on the spot.  node == kNone afterwards: the set is complete (c_n: its size, kPointCacheBad: no usable set).
template <int STACK>
__device__ __forceinline__ void mesh_point_step(const float4 *__restrict__ qnodes, const MeshArrays &ma, const uint32_t *__restrict__ tri_fidx,
                                                float px, float py, const TriFilterParams &tpar, float Tb, uint32_t self_tri,
                                                uint32_t &node, uint32_t &sp, uint32_t *lds_stack, uint32_t tid,
                                                uint32_t &c_n, uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &nbox, uint32_t &nleaf)
{
    const float T = tpar.A;
    const float rx = -tpar.npx, ry = -tpar.npy;                         // the point relative to the scene centre (the records' frame)
    const float4 *np = qnodes + 4 * (size_t)node;
    const float4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
    const uint32_t w[4] = { __float_as_uint(n3.x), __float_as_uint(n3.y), __float_as_uint(n3.z), __float_as_uint(n3.w) };
    const uint32_t qx[4] = { __float_as_uint(n1.x), __float_as_uint(n1.y), __float_as_uint(n1.z), __float_as_uint(n1.w) };
    const uint32_t qy[4] = { __float_as_uint(n2.x), __float_as_uint(n2.y), __float_as_uint(n2.z), __float_as_uint(n2.w) };
    uint32_t next = kNone;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint32_t type = w[c] >> kQNodeShift, idx = w[c] & kQNodeIndexMask;
        const float lox = __builtin_fmaf((float)(qx[c] & 0xFFFFu), n0.z, n0.x), hix = __builtin_fmaf((float)(qx[c] >> 16), n0.z, n0.x);
        const float loy = __builtin_fmaf((float)(qy[c] & 0xFFFFu), n0.w, n0.y), hiy = __builtin_fmaf((float)(qy[c] >> 16), n0.w, n0.y);
        const bool in = type != kQNodeEmpty && px >= lox - Tb && px <= hix + Tb && py >= loy - Tb && py <= hiy + Tb;
        if (!in) continue;
        if (type == 0u) {                                              // an interior child whose rectangle holds the point
            if (next == kNone) next = idx;
            else if (sp < (uint32_t)STACK) { lds_stack[(size_t)sp * kBvhThreads + tid] = idx; sp += 1; }
            else c_n = kPointCacheBad;                                 // (deeper than the LDS stack: no set for this point, its segments walk)
        } else {
            for (uint32_t k = 0; k < type; ++k) {
                const float4 A = ma.tri_f32[2 * (size_t)(idx + k)], B = ma.tri_f32[2 * (size_t)(idx + k) + 1];
                const float nv = __builtin_fmaf(A.x, tpar.npx, __builtin_fmaf(A.y, tpar.npy, __builtin_fmaf(A.z, tpar.npz, A.w)));
                const bool near = __builtin_fabsf(nv) <= T && __builtin_fabsf(rx - B.x) <= B.z + T && __builtin_fabsf(ry - B.y) <= B.w + T;
                if (near && tri_fidx[idx + k] != self_tri) {
                    if (c_n == 0u) c0 = idx + k; else if (c_n == 1u) c1 = idx + k; else if (c_n == 2u) c2 = idx + k;
                    c_n = c_n < kPointCache ? c_n + 1u : kPointCacheBad;
                }
            }
            nleaf += type;
        }
    }
    nbox += 4;
    node = next;
    if (node == kNone && sp != 0u) { sp -= 1; node = lds_stack[(size_t)sp * kBvhThreads + tid]; }
}

// ---- the same two halves for a joint tree (128-byte nodes: 3-D nodes and footprint nodes, sphere and triangle leaves) ---------
// A noted leaf is one word: bit 31 = triangle leaf, bits 28-30 = its records (1..6; a sphere leaf has 1), bits 0-27 = the first
// record / leaf entry (the upload keeps a tree below 2^28 shapes).
constexpr uint32_t kJNoteTri = 0x80000000u, kJNoteShift = 28, kJNoteIndexMask = (1u << 28) - 1u;

template <bool SPILL, int STACK, class RAY>
__device__ __forceinline__ void jnode_open(const float4 *__restrict__ nodes, const RAY &q, uint32_t &node, uint32_t &sp, bool &overflow,
                                           float best_up, MeshPending &pend, uint32_t *lds_stack, uint32_t tid,
                                           uint32_t *__restrict__ spill, uint32_t spill_entries, size_t spill_stride, size_t glane,
                                           uint32_t &nbox)
{
    const float4 *np = nodes + 8 * (size_t)(node & ~kBvhFlatNode);
    float4 nd[8];
#pragma unroll
    for (int c = 0; c < 6; ++c) nd[c] = np[c];
    uint32_t lnk[4], cnt[4];
    float tc[4];
    if (node & kBvhFlatNode) {
        lnk[0] = __float_as_uint(nd[4].x); lnk[1] = __float_as_uint(nd[4].y); lnk[2] = __float_as_uint(nd[4].z); lnk[3] = __float_as_uint(nd[4].w);
        cnt[0] = __float_as_uint(nd[5].x); cnt[1] = __float_as_uint(nd[5].y); cnt[2] = __float_as_uint(nd[5].z); cnt[3] = __float_as_uint(nd[5].w);
#pragma unroll
        for (int c = 0; c < 4; ++c) tc[c] = rect_entry32(nd[c], q, best_up);
    } else {
        nd[6] = np[6]; nd[7] = np[7];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            lnk[c] = __float_as_uint(nd[c].w); cnt[c] = __float_as_uint(nd[4 + c].w);
            tc[c] = box_entry32(nd[c], nd[4 + c], q, best_up);
        }
    }
    nbox += 4;
    float key[4];
    uint32_t kl[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (tc[c] < __builtin_inff() && cnt[c] - 1u < 0x1FFFFu) {                 // a leaf the ray enters: noted, read later
            const uint32_t note = ((cnt[c] & kBvhTriLeaf) ? kJNoteTri : 0u) | ((cnt[c] & 0xFFFFu) << kJNoteShift) | lnk[c];
            pend.p3 = pend.p2; pend.p2 = pend.p1; pend.p1 = pend.p0; pend.p0 = note;
            pend.n += 1;
        }
        key[c] = cnt[c] == 0u ? tc[c] : __builtin_inff();
        kl[c] = lnk[c];
    }
#define RTX_CSWAP(i, j) { if (key[j] < key[i]) { float tk = key[i]; key[i] = key[j]; key[j] = tk; uint32_t tl = kl[i]; kl[i] = kl[j]; kl[j] = tl; } }
    RTX_CSWAP(0, 1) RTX_CSWAP(2, 3) RTX_CSWAP(0, 2) RTX_CSWAP(1, 3) RTX_CSWAP(1, 2)
#undef RTX_CSWAP
    const uint32_t npush = (key[1] < __builtin_inff() ? 1u : 0u) + (key[2] < __builtin_inff() ? 1u : 0u) +
                           (key[3] < __builtin_inff() ? 1u : 0u);
    if (sp + 3u <= (uint32_t)STACK) {
#pragma unroll
        for (uint32_t i = 1; i <= 3; ++i) {
            const uint32_t row = i <= npush ? sp + npush - i : (uint32_t)STACK;      // (row STACK = the sink)
            lds_stack[(size_t)row * kBvhThreads + tid] = kl[i];
        }
        sp += npush;
    } else {
#define RTX_PUSH(v)                                                                                           \
        {                                                                                                     \
            if (sp < (uint32_t)STACK) { lds_stack[(size_t)sp * kBvhThreads + tid] = (v); sp += 1; }           \
            else if (SPILL && sp - (uint32_t)STACK < spill_entries) {                                         \
                spill[(size_t)(sp - (uint32_t)STACK) * spill_stride + glane] = (v); sp += 1;                  \
            } else overflow = true;                                                                           \
        }
        if (key[3] < __builtin_inff()) RTX_PUSH(kl[3])
        if (key[2] < __builtin_inff()) RTX_PUSH(kl[2])
        if (key[1] < __builtin_inff()) RTX_PUSH(kl[1])
#undef RTX_PUSH
    }
    node = key[0] < __builtin_inff() ? kl[0] : kNone;
    if (node == kNone && sp != 0u) {
        sp -= 1;
        node = (!SPILL || sp < (uint32_t)STACK) ? lds_stack[(size_t)sp * kBvhThreads + tid]
                                                : spill[(size_t)(sp - (uint32_t)STACK) * spill_stride + glane];
    }
}

__device__ __forceinline__ bool jleaf_read(const MeshArrays &ma, const SphereRay &sr, const TriFilterParams &tpar, MeshPending &pend,
                                           uint32_t &qcnt, bool &overflow, float &best_up, uint32_t *lds_q, uint32_t tid, uint32_t &nleaf)
{
    const uint32_t note = pend.p0;
    const uint32_t n = (note >> kJNoteShift) & 7u, first = note & kJNoteIndexMask;
    if (!mesh_queue_room(lds_q, tid, qcnt, best_up, n)) {
        if (qcnt != 0u) return false;
        overflow = true;
    }
    pend.p0 = pend.p1; pend.p1 = pend.p2; pend.p2 = pend.p3;
    pend.n -= 1;
    if (note & kJNoteTri) {
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
                if (tlo <= best_up && tlo < __builtin_inff()) {
                    best_up = fminf(best_up, thi);
                    if (qcnt < (uint32_t)kMeshQueue) {
                        lds_q[(size_t)qcnt * kBvhThreads + tid] = (first + k + j) | kQueueTri;
                        lds_q[(size_t)(kMeshQueue + qcnt) * kBvhThreads + tid] = __float_as_uint(tlo);
                        qcnt += 1;
                    }
                }
            }
        }
    } else {
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
    return true;
}

}  // namespace rtx
