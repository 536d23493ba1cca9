// rtx_traverse.h -- pieces of the flat-BVH traversal shared by trace_bvh_kernel and trace_pool_kernel.
#pragma once

#include "rtx_device.h"

namespace rtx {

constexpr int kBvhThreads = 256;
constexpr uint32_t kNone = 0xFFFFFFFFu;

// The ray as the slab test sees it: per axis inv = fl(1 / d) and noi = fl(-o * inv), so that the distance to the
// plane x = b is one FMA, t = fl(b * inv + noi).
struct Ray32 { float ix, iy, iz, nx, ny, nz; };

// |1/d| is clamped to inv_max (SceneView::bvh_inv_max, <= 1e30 and small enough that o * inv stays finite): an
// axis the ray is (almost) parallel to then gives two huge finite distances of the right signs instead of inf / NaN.
// Over the distances that matter (t <= 4 * origin_limit: both ends of a reportable hit lie in the tree's range) the
// ray moves by less than 1e-29 * origin_limit along such an axis, far inside the boxes' padding.
__device__ __forceinline__ void ray32_axis(double o, double dn, double inv_max, float &inv, float &noi)
{
    double i = 1.0 / dn;
    if (!(fabs(i) <= inv_max)) i = copysign(inv_max, dn);
    inv = (float)i;
    noi = (float)(-o * (double)inv);          // one rounding of the exact product (the f64 product's own error is 2^-53)
}

__device__ __forceinline__ void make_ray32(const V3 &pos, const V3 &dirn, double inv_max, Ray32 &r)
{
    ray32_axis(pos.x, dirn.x, inv_max, r.ix, r.nx);
    ray32_axis(pos.y, dirn.y, inv_max, r.iy, r.ny);
    ray32_axis(pos.z, dirn.z, inv_max, r.iz, r.nz);
}

// f32 slab test; returns a lower bound (>= 0) of the entry distance, or +inf on a certain miss.
//   t = fl(b * inv + noi) = (b - o) * inv up to: the rounding of noi, |o * inv| * 2^-24, which is the plane moved by
//   2^-24 |o| and is covered by the boxes' absolute padding (rtx_bvh.h); and two relative roundings (inv, the FMA),
//   < 2^-22 on every t, for which the interval is widened by 2^-21 |t|.
// A NaN can only come from 0 * huge (never: inv is finite) or from a z slab of +-inf bounds times a finite inv
// (never NaN either), so plain min/max are safe; an unbounded z slab gives -inf/+inf and drops out.
__device__ __forceinline__ float box_entry32(const float4 lo, const float4 hi, const Ray32 &r, float best_up)
{
    const float x0 = __builtin_fmaf(lo.x, r.ix, r.nx), x1 = __builtin_fmaf(hi.x, r.ix, r.nx);
    const float y0 = __builtin_fmaf(lo.y, r.iy, r.ny), y1 = __builtin_fmaf(hi.y, r.iy, r.ny);
    const float z0 = __builtin_fmaf(lo.z, r.iz, r.nz), z1 = __builtin_fmaf(hi.z, r.iz, r.nz);
    const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.0f));
    const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    const float tn_lo = tn * (1.0f - 4.76837158e-7f);                   // tn >= 0: (1 - 2^-21) tn is a lower bound
    const float tf_hi = tf * (1.0f + 4.76837158e-7f);                   // an upper bound when tf >= 0 (else a miss anyway)
    const bool hit = (tn_lo <= tf_hi) && (tn_lo <= best_up);
    return hit ? tn_lo : __builtin_inff();      // the widened (conservative) entry distance
}

// best (f64) rounded UP to f32 for the pruning comparison
__device__ __forceinline__ float round_up32(double best)
{
    float b = (float)best;
    if ((double)b < best) b = __uint_as_float(__float_as_uint(b) + (b >= 0.0f ? 1u : 0xFFFFFFFFu));
    return b;
}

constexpr int kBvhQueue = 8;                // candidate shapes a lane may hold between two exact passes
constexpr uint32_t kQueueTri = 0x80000000u; // queue entry: a triangle filter record (else a local sphere index)

struct LeafArrays {                         // the arrays the leaves index (kernel arguments: global address space)
    const float4 *sphere_f32;               // per sphere leaf entry
    const uint32_t *sphere_prims;
    const SphereX *spheres;
    const uint32_t *sphere_ids;
    const float4 *tri_f32;                  // two per triangle filter record
    const uint32_t *tri_fidx;
    const TriX *tris;
};

// Exact f64 tests (sphere.rs:19-30, triangle.rs:108-127) of the queued candidates; updates the winner and the
// pruning bound.
template <bool TRIS>
__device__ __forceinline__ void flush_candidates(const LeafArrays &la, const RayX &rx, const uint32_t *lds_q, uint32_t tid,
                                                 uint32_t &qcnt, Hit &h, float &best_up, unsigned long long &exact)
{
#pragma unroll 1
    for (uint32_t k = 0; k < qcnt; ++k) {         // not unrolled: 8 inlined copies of the f64 tests per call site bloat the traversal loop
        const uint32_t idx = lds_q[(size_t)k * kBvhThreads + tid];
        double t;
        if (TRIS && (idx & kQueueTri)) {
            const uint32_t tk = la.tri_fidx[idx & ~kQueueTri];
            if (triangle_distance(la.tris[tk], rx, &t)) hit_consider(h, t, la.tris[tk].id, 2, tk);
        } else {
            if (sphere_distance(la.spheres[idx], rx, &t)) hit_consider(h, t, la.sphere_ids[idx], 0, idx);
        }
    }
    exact += qcnt;
    qcnt = 0;
    if (h.id != kNone) best_up = round_up32(h.t);
}


}  // namespace rtx
