// rtx_traverse.h -- pieces of the flat-BVH traversal shared by trace_bvh_kernel and trace_pool_kernel.
#pragma once

#include "rtx_device.h"

namespace rtx {

constexpr int kBvhThreads = 256;
constexpr uint32_t kNone = 0xFFFFFFFFu;

struct Ray32 { float ox, oy, oz, ix, iy, iz; };

// f32 slab test; returns a lower bound of the entry distance, or +inf on a certain miss.
//   t = fl(fl(b - fl(o)) * fl(1/d)): the origin rounding is covered by the boxes' absolute padding (rtx_bvh.h);
//   the remaining roundings are a relative error < 2^-22 on every t, so the interval is widened by 2^-21 |t|.
// fminf/fmaxf return the non-NaN operand (0 * inf: origin on a slab of an axis-parallel ray), which only widens
// the interval; an infinite tn/tf of a ray that runs outside a slab turns the widened bound into NaN and the
// comparison into "miss", which is the right answer.
__device__ __forceinline__ float box_entry32(const float4 lo, const float4 hi, const Ray32 &r, float best_up)
{
    const float x0 = (lo.x - r.ox) * r.ix, x1 = (hi.x - r.ox) * r.ix;
    const float y0 = (lo.y - r.oy) * r.iy, y1 = (hi.y - r.oy) * r.iy;
    const float z0 = (lo.z - r.oz) * r.iz, z1 = (hi.z - r.oz) * r.iz;
    const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
    const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    const float eps = 4.76837158e-7f;                                   // 2^-21
    const float tn_lo = __builtin_fmaf(-eps, __builtin_fabsf(tn), tn);
    const float tf_hi = __builtin_fmaf(eps, __builtin_fabsf(tf), tf);
    const bool hit = (tn_lo <= tf_hi) && (tf_hi >= 0.0f) && (tn_lo <= best_up);
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
__device__ __forceinline__ void flush_candidates(const LeafArrays &la, const RayX &rx, const uint32_t *lds_q, uint32_t tid,
                                                 uint32_t &qcnt, Hit &h, float &best_up, unsigned long long &exact)
{
#pragma unroll 1
    for (uint32_t k = 0; k < qcnt; ++k) {         // not unrolled: 8 inlined copies of the f64 tests per call site bloat the traversal loop
        const uint32_t idx = lds_q[(size_t)k * kBvhThreads + tid];
        double t;
        if (idx & kQueueTri) {
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
