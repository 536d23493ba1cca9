// rtx_traverse.h -- the flat-BVH traversal shared by trace_bvh_kernel and trace_bvh_regroup_kernel.
#pragma once

#include "rtx_device.h"

namespace rtx {

constexpr int kBvhThreads = 256;
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr float kBvhRange64 = 134217728.0f;       // 2^27: origins up to this multiple of origin_limit walk the tree in f64

// The ray as the slab test sees it: per axis inv = fl(1 / d) and noi = fl(-o * inv), so that the distance to the
// plane x = b is one FMA, t = fl(b * inv + noi).
struct Ray32 { float ix, iy, iz, nx, ny, nz; };
// The same in f64, for rays whose origin is outside the range the f32 test is valid for (a bounce off one of the
// reference's far phantom hits, a camera far from the scene): rounding noi in f64 moves the planes by 2^-53 |o|, which
// the boxes' padding covers up to |o| <= 2^27 * origin_limit.  Without it such a ray would test every shape exactly --
// 0.3 s for one segment over 500k triangles, while its wave and the launch wait.
struct Ray64 { double ix, iy, iz, nx, ny, nz; };
// Ray32 with an absolute slack on every slab distance, for the same far origins where no f64 is wanted (the wavefront
// walk kernels): what the boxes' padding no longer covers is the rounding of noi = fl(-o * inv), at most 2^-24 |o * inv|
// per axis, so the interval [tn, tf] is widened by e = 2^-23 max_axis |noi| on both sides -- conservative for any finite
// origin, and e = 0 (the same bits as Ray32) for origins inside origin_limit.
struct Ray32S { float ix, iy, iz, nx, ny, nz, e; };
__device__ __forceinline__ float ray_slack(const Ray32 &) { return 0.0f; }
__device__ __forceinline__ float ray_slack(const Ray32S &r) { return r.e; }

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
template <class R32>
__device__ __forceinline__ float box_entry32(const float4 lo, const float4 hi, const R32 &r, float best_up)
{
    const float x0 = __builtin_fmaf(lo.x, r.ix, r.nx), x1 = __builtin_fmaf(hi.x, r.ix, r.nx);
    const float y0 = __builtin_fmaf(lo.y, r.iy, r.ny), y1 = __builtin_fmaf(hi.y, r.iy, r.ny);
    const float z0 = __builtin_fmaf(lo.z, r.iz, r.nz), z1 = __builtin_fmaf(hi.z, r.iz, r.nz);
    const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), 0.0f));
    const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
    const float e = ray_slack(r);
    const float tn_lo = __builtin_fmaf(tn, 1.0f - 4.76837158e-7f, -e);   // tn >= 0: (1 - 2^-21) tn is a lower bound
    const float tf_hi = __builtin_fmaf(tf, 1.0f + 4.76837158e-7f, e);    // an upper bound when tf >= 0 (else a miss anyway)
    const bool hit = (tn_lo <= tf_hi) && (tn_lo <= best_up);
    return hit ? tn_lo : __builtin_inff();      // the widened (conservative) entry distance
}

// origin_limit < |o|_inf: the slack of Ray32S (rounded up); 0 inside the range the padding covers
__device__ __forceinline__ float ray32_slack(float nx, float ny, float nz, bool in_range)
{
    const float m = fmaxf(fmaxf(__builtin_fabsf(nx), __builtin_fabsf(ny)), __builtin_fabsf(nz));
    return in_range ? 0.0f : m * (1.1920929e-7f * (1.0f + 9.5367432e-7f));
}

__device__ __forceinline__ void make_ray64(const V3 &pos, const V3 &dirn, double inv_max, Ray64 &r)
{
    double i;
    i = 1.0 / dirn.x; if (!(fabs(i) <= inv_max)) i = copysign(inv_max, dirn.x); r.ix = i; r.nx = -pos.x * i;
    i = 1.0 / dirn.y; if (!(fabs(i) <= inv_max)) i = copysign(inv_max, dirn.y); r.iy = i; r.ny = -pos.y * i;
    i = 1.0 / dirn.z; if (!(fabs(i) <= inv_max)) i = copysign(inv_max, dirn.z); r.iz = i; r.nz = -pos.z * i;
}

// f64 slab test on the same f32 boxes (same widening, far more than the f64 roundings need); the entry distance is
// returned as an f32 rounded DOWN, so the f32 ordering / pruning code of bvh_step is shared.
__device__ __forceinline__ float box_entry32(const float4 lo, const float4 hi, const Ray64 &r, float best_up)
{
    const double x0 = __builtin_fma((double)lo.x, r.ix, r.nx), x1 = __builtin_fma((double)hi.x, r.ix, r.nx);
    const double y0 = __builtin_fma((double)lo.y, r.iy, r.ny), y1 = __builtin_fma((double)hi.y, r.iy, r.ny);
    const double z0 = __builtin_fma((double)lo.z, r.iz, r.nz), z1 = __builtin_fma((double)hi.z, r.iz, r.nz);
    const double tn = fmax(fmax(fmin(x0, x1), fmin(y0, y1)), fmax(fmin(z0, z1), 0.0));
    const double tf = fmin(fmin(fmax(x0, x1), fmax(y0, y1)), fmax(z0, z1));
    const double tn_lo = tn * (1.0 - 4.76837158e-7);
    const double tf_hi = tf * (1.0 + 4.76837158e-7);
    const bool hit = (tn_lo <= tf_hi) && (tn_lo <= (double)best_up);
    return hit ? __double2float_rd(tn_lo) : __builtin_inff();
}

// The same tests for a footprint node's child {lo.x, lo.y, hi.x, hi.y} (rtx_bvh.h): two slabs.
template <class R32>
__device__ __forceinline__ float rect_entry32(const float4 r4, const R32 &r, float best_up)
{
    const float x0 = __builtin_fmaf(r4.x, r.ix, r.nx), x1 = __builtin_fmaf(r4.z, r.ix, r.nx);
    const float y0 = __builtin_fmaf(r4.y, r.iy, r.ny), y1 = __builtin_fmaf(r4.w, r.iy, r.ny);
    const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), 0.0f);
    const float tf = fminf(fmaxf(x0, x1), fmaxf(y0, y1));
    const float e = ray_slack(r);
    const float tn_lo = __builtin_fmaf(tn, 1.0f - 4.76837158e-7f, -e);
    const float tf_hi = __builtin_fmaf(tf, 1.0f + 4.76837158e-7f, e);
    const bool hit = (tn_lo <= tf_hi) && (tn_lo <= best_up);
    return hit ? tn_lo : __builtin_inff();
}

__device__ __forceinline__ float rect_entry32(const float4 r4, const Ray64 &r, float best_up)
{
    const double x0 = __builtin_fma((double)r4.x, r.ix, r.nx), x1 = __builtin_fma((double)r4.z, r.ix, r.nx);
    const double y0 = __builtin_fma((double)r4.y, r.iy, r.ny), y1 = __builtin_fma((double)r4.w, r.iy, r.ny);
    const double tn = fmax(fmax(fmin(x0, x1), fmin(y0, y1)), 0.0);
    const double tf = fmin(fmax(x0, x1), fmax(y0, y1));
    const double tn_lo = tn * (1.0 - 4.76837158e-7);
    const double tf_hi = tf * (1.0 + 4.76837158e-7);
    const bool hit = (tn_lo <= tf_hi) && (tn_lo <= (double)best_up);
    return hit ? __double2float_rd(tn_lo) : __builtin_inff();
}

// best (f64) rounded UP to f32 for the pruning comparison
__device__ __forceinline__ float round_up32(double best)
{
    float b = (float)best;
    if ((double)b < best) b = __uint_as_float(__float_as_uint(b) + (b >= 0.0f ? 1u : 0xFFFFFFFFu));
    return b;
}

// wave-uniform reads through the scalar cache: the constant address space makes the compiler select s_load for them
// (the packet walks of rtx_wavefront.hip and rtx_bvh_spheres.hip: one request per wave instead of 64 address-divergent ones)
typedef float PkF4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) PkF4 *LdsF4Ptr;      // a pointer into LDS, typed as such (ds_read, never flat)
struct PkConst4 {
    const __attribute__((address_space(4))) PkF4 *p;
    __device__ __forceinline__ PkConst4 operator+(size_t i) const { return PkConst4{p + i}; }
    __device__ __forceinline__ float4 operator[](size_t i) const { const PkF4 v = p[i]; return make_float4(v.x, v.y, v.z, v.w); }
};
__device__ __forceinline__ PkConst4 pk_const(const float4 *p) { return PkConst4{(const __attribute__((address_space(4))) PkF4 *)(uintptr_t)p}; }
__device__ __forceinline__ uint32_t pk_bits(float f) { return __builtin_amdgcn_readfirstlane(__float_as_uint(f)); }
struct PkConstU32 {
    const __attribute__((address_space(4))) uint32_t *p;
    __device__ __forceinline__ uint32_t operator[](size_t i) const { return p[i]; }
};
__device__ __forceinline__ PkConstU32 pk_const(const uint32_t *p) { return PkConstU32{(const __attribute__((address_space(4))) uint32_t *)(uintptr_t)p}; }

// v_writelane_b32: lane `lane` (wave-uniform) of `old` replaced by the wave-uniform `value` -- the packet walks keep their
// wave-uniform stack in the lanes of one VGPR.  This clang has no __builtin_amdgcn_writelane; the LLVM intrinsic is reached by
// its name.  That is a toolchain-version hazard, so it is guarded twice: the build refuses a clang major nobody has validated
// (re-run tests/test_gpu_parity.py::test_writelane_and_f64_minmax_kats on the new toolchain, then extend the list or pass
// -DRTX_WRITELANE_VALIDATED), and that test checks the instruction's semantics on the device through rtx_debug_math op 6.
#if !defined(RTX_WRITELANE_VALIDATED) && defined(__clang_major__)
static_assert(__clang_major__ == 22, "rtx_writelane binds llvm.amdgcn.writelane.i32 by name: validated with AMD clang 22 (ROCm 7.2) only");
#endif
extern "C" __device__ int rtx_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");

// The child sort's compare-exchange on (key, link) pairs packed as the high / low word of an f64 (sphere_node_step_q3).  Written as
// instructions: the pairs are not canonical doubles and fmin() would quiet them first.  A key of +0.0 (the ray starts inside the
// child's box: the common case) makes the pair an f64 DENORMAL whose whole payload is the link, so this relies on the kernels' f64
// denormal mode being "preserve" -- the default for gfx950, needed by the reference's arithmetic anyway, and checked on the device
// (rtx_debug_math ops 7 / 8, same test): a flushed pair would send the walk to node 0 for ever.
__device__ __forceinline__ void rtx_minmax_f64_bits(double a, double b, double &lo, double &hi)
{
    asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(a), "v"(b));
    asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(a), "v"(b));
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


// ---- shared by trace_bvh_kernel (rtx_bvh.hip) and trace_bvh_regroup_kernel (rtx_bvh_regroup.hip) ----------------------
#ifndef RTX_BVH_WPE
#define RTX_BVH_WPE 4
#endif
constexpr int kBvhWavesPerSimd = RTX_BVH_WPE;     // = workgroups per CU (4 waves each)

__device__ __forceinline__ uint32_t bvh_mbcnt(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// One traversal step of one lane: open wide node `node`, filter its leaf children into the candidate queue, push
// the interior children still in reach (farthest first) and move to the nearest (or pop).  node == kNone afterwards
// means the traversal is complete.  lds_stack has one row more than kBvh4StackEntries: the sink of the branch-free
// pushes.
template <bool TRIS, bool SPILL, class RAY>
__device__ __forceinline__ void bvh_step(const float4 *__restrict__ nodes, const LeafArrays &la, const RAY &q,
                                         const FilterParams &fpar, const TriFilterParams &tpar, const RayX &rx, uint32_t &node,
                                         uint32_t &sp, uint32_t &qcnt, bool &overflow, Hit &h, float &best_up,
                                         uint32_t *lds_stack, uint32_t *lds_q, uint32_t tid, uint32_t *__restrict__ spill,
                                         uint32_t spill_entries, size_t spill_stride, size_t glane,
                                         uint32_t &nbox, uint32_t &nleaf, unsigned long long &exact)
{
    // one 128-byte node: the boxes of up to four children (rtx_bvh.h Bvh4Node; 8 loads), or -- in the triangle
    // sub-tree -- their footprint rectangles, links and counts (6 loads)
    const float4 *np = nodes + 8 * (size_t)(node & ~kBvhFlatNode);
    float tc[4];
    uint32_t lnk[4], cnt[4];
    if (TRIS && (node & kBvhFlatNode)) {
        float4 rc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) rc[c] = np[c];
        const float4 lk = np[4], ct = np[5];
#pragma unroll
        for (int c = 0; c < 4; ++c) tc[c] = rect_entry32(rc[c], q, best_up);
        lnk[0] = __float_as_uint(lk.x); lnk[1] = __float_as_uint(lk.y); lnk[2] = __float_as_uint(lk.z); lnk[3] = __float_as_uint(lk.w);
        cnt[0] = __float_as_uint(ct.x); cnt[1] = __float_as_uint(ct.y); cnt[2] = __float_as_uint(ct.z); cnt[3] = __float_as_uint(ct.w);
    } else {
        float4 ca[4], cb[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) { ca[c] = np[c]; cb[c] = np[4 + c]; }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            tc[c] = box_entry32(ca[c], cb[c], q, best_up);
            lnk[c] = __float_as_uint(ca[c].w);
            cnt[c] = __float_as_uint(cb[c].w);
        }
    }
    nbox += 4;
    // leaf children that the ray enters: f32 filter now, survivors are queued; the exact f64 tests run
    // every 4th step (and at the end) for all lanes together, so their cost is not paid per (step, child,
    // shape) under divergence.  The pruning bound lags by at most 4 steps, which only costs visits.
    uint32_t leafmask = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (tc[c] < __builtin_inff() && cnt[c] - 1u < 0x1FFFFu) leafmask |= 1u << c;          // neither interior (0) nor empty (~0)
    while (leafmask != 0u) {                  // one copy of the leaf code, however many of the four children are leaves
        const uint32_t c = (uint32_t)__builtin_ctz(leafmask);
        leafmask &= leafmask - 1u;
        const uint32_t first = c == 0 ? lnk[0] : (c == 1 ? lnk[1] : (c == 2 ? lnk[2] : lnk[3]));
        const uint32_t count = c == 0 ? cnt[0] : (c == 1 ? cnt[1] : (c == 2 ? cnt[2] : cnt[3]));
        const uint32_t n = count & 0xFFFFu;
        if (TRIS && (count & kBvhTriLeaf)) {
            // two records per round, their four loads in flight together (the array is padded by one record, so the
            // second pair may be read even when it belongs to the next leaf)
            for (uint32_t k = 0; k < n; k += 2u) {
                const float4 *rp = la.tri_f32 + 2 * (size_t)(first + k);
                const float4 A0 = rp[0], B0 = rp[1], A1 = rp[2], B1 = rp[3];
                uint32_t m = (int)tri_filter_sign(A0, B0, tpar) >= 0 ? 1u : 0u;      // q may be above the footprint
                if (k + 1u < n && (int)tri_filter_sign(A1, B1, tpar) >= 0) m |= 2u;
                while (m != 0u) {
                    const uint32_t j = (uint32_t)__builtin_ctz(m);
                    m &= m - 1u;
                    if (qcnt == (uint32_t)kBvhQueue) flush_candidates<TRIS>(la, rx, lds_q, tid, qcnt, h, best_up, exact);
                    lds_q[(size_t)qcnt * kBvhThreads + tid] = (first + k + j) | kQueueTri;
                    qcnt += 1;
                }
            }
        } else {
            for (uint32_t k = 0; k < n; ++k) {
                const float4 rec = la.sphere_f32[first + k];
                if ((int)__float_as_uint(filter_disc1(rec, fpar)) >= 0) {            // D >= 0: cannot be excluded
                    if (qcnt == (uint32_t)kBvhQueue) flush_candidates<TRIS>(la, rx, lds_q, tid, qcnt, h, best_up, exact);
                    lds_q[(size_t)qcnt * kBvhThreads + tid] = la.sphere_prims[first + k];
                    qcnt += 1;
                }
            }
        }
        nleaf += n;
    }
    // interior children still in reach, nearest first: keys = entry distance (inf = not to be visited)
    float key[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) key[c] = cnt[c] == 0u ? tc[c] : __builtin_inff();     // (tc is already inf for a box out of reach)
#define RTX_CSWAP(i, j) { if (key[j] < key[i]) { float tk = key[i]; key[i] = key[j]; key[j] = tk; uint32_t tl = lnk[i]; lnk[i] = lnk[j]; lnk[j] = tl; } }
    RTX_CSWAP(0, 1) RTX_CSWAP(2, 3) RTX_CSWAP(0, 2) RTX_CSWAP(1, 3) RTX_CSWAP(1, 2)
#undef RTX_CSWAP
    // push the farther ones (farthest first), descend into the nearest.  The keys are sorted, so the children to push
    // are lnk[1..npush]; lnk[i] goes to row sp + npush - i, the others to the sink row -- three unconditional stores.
    // (stack + queue = 39 words of LDS per lane, which is what 16 waves per CU leave; entries beyond the 30 in LDS go
    // to the lane's column of the HBM spill area, which the launcher sizes from the tree's depth so that it cannot run
    // out -- the exhaustive sweep after an overflow is only a guard)
    const uint32_t npush = (key[1] < __builtin_inff() ? 1u : 0u) + (key[2] < __builtin_inff() ? 1u : 0u) +
                           (key[3] < __builtin_inff() ? 1u : 0u);
    if (sp + 3u <= (uint32_t)kBvh4StackEntries) {
#pragma unroll
        for (uint32_t i = 1; i <= 3; ++i) {
            const uint32_t row = i <= npush ? sp + npush - i : (uint32_t)kBvh4StackEntries;
            lds_stack[(size_t)row * kBvhThreads + tid] = lnk[i];
        }
        sp += npush;
    } else {
#define RTX_PUSH(v)                                                                                      \
        {                                                                                                \
            if (sp < (uint32_t)kBvh4StackEntries) { lds_stack[(size_t)sp * kBvhThreads + tid] = (v); sp += 1; } \
            else if (SPILL && sp - (uint32_t)kBvh4StackEntries < spill_entries) {                        \
                spill[(size_t)(sp - (uint32_t)kBvh4StackEntries) * spill_stride + glane] = (v); sp += 1; \
            } else overflow = true;                                                                      \
        }
        if (key[3] < __builtin_inff()) RTX_PUSH(lnk[3])
        if (key[2] < __builtin_inff()) RTX_PUSH(lnk[2])
        if (key[1] < __builtin_inff()) RTX_PUSH(lnk[1])
#undef RTX_PUSH
    }
    node = key[0] < __builtin_inff() ? lnk[0] : kNone;
    if (node == kNone && sp != 0u) {
        sp -= 1;                        // its boxes are re-tested against the current bound when it is opened
        node = (!SPILL || sp < (uint32_t)kBvh4StackEntries) ? lds_stack[(size_t)sp * kBvhThreads + tid]
                                                : spill[(size_t)(sp - (uint32_t)kBvh4StackEntries) * spill_stride + glane];
    }
}

// ---- spheres-only trees: a traversal loop without f64 -----------------------------------------------------------------
// bvh_step above interleaves the exact f64 tests with the walk (every 4th step) because its pruning bound is the exact
// winner's distance.  For spheres an f32 evaluation bounds the hit distance tightly from both sides when it is done
// relative to the RAY ORIGIN (the sweep filter's expanded form cancels |c|^2 ~ M^2 against itself: its error 24uM^2 is
// the size of r^2 for the small spheres of C2; here the cancellation is over |l| <= r):
//     o = c - p        beta = o.d        l = o - beta d        Delta = r^2 - |l|^2        t = beta - sqrt(Delta)
// (|d| = 1; sphere.rs:22-29 in closest-approach form: Delta is a quarter of the reference's discriminant.)  With
// u = 2^-24, every input rounded once to f32 (c and p relative to the scene centre, |c| + r + |p| <= M) and one
// rounding per operation:   |o^ - o| <= 3uM per component,   |beta^ - beta| <= 16uM,   |l^ - l| <= 24uM per component,
//     |Delta^ - Delta| <= 84uM|l| + 1728u^2M^2 + 4u(|l|^2 + r^2)  <=  G := 128uM r + 8192u^2M^2      whenever |l| <= r + 43uM,
// which holds for every sphere the reference reports (|l| < r) and whenever Delta^ > G.  Hence
//     * a sphere the reference can report has Delta^ + G >= 0            -> the leaf filter (conservative)
//     * Delta^ - G > 0 and t_lo > K  =>  the reference DOES report it     -> a certain hit
//     * t_lo = beta^ - sqrt(Delta^ + G) - K  <=  t  <=  beta^ - sqrt(Delta^ - G) + K = t_hi,   K = 24uM
// (K: beta's 16uM, the roundings of these few operations, the hardware sqrt's 1 ulp via the (1 +- 2^-21) factors, and
// the reference's own f64 roundings, ~1e-15 M).  Boxes are pruned against best_up = min t_hi over the certain hits --
// pure f32 -- and only candidates that can still be the winner (t_lo <= best_up) reach the exact f64 test, once, after
// the walk.  The winner W is never lost: its boxes have an entry distance <= t_W <= best_up, it passes the filter, and
// t_lo(W) <= t_W <= best_up.  The loop then holds no f64 value.
struct SphereRay { float px, py, pz, dx, dy, dz, Kg, c0, K; };     // p relative to the scene centre; Kg = 128uM, c0 = 8192u^2M^2
constexpr int kSphQueue = 4;                 // live candidates per lane: {local sphere index, t_lo}; more than 4 at once
                                             // (coincident spheres) -> the segment tests every sphere exactly

__device__ __forceinline__ void sphere_ray_from(const SceneView &sv, V3 pos, V3 dir, SphereRay &f)
{
    const double px = pos.x - sv.sphere_center[0], py = pos.y - sv.sphere_center[1], pz = pos.z - sv.sphere_center[2];
    const double M = sv.sphere_cmax + sqrt(px * px + py * py + pz * pz);
    f.px = (float)px; f.py = (float)py; f.pz = (float)pz;
    f.dx = (float)dir.x; f.dy = (float)dir.y; f.dz = (float)dir.z;
    if (!(M < 1.0e14) || !(M > 1.0e-12)) {             // outside the range the bounds were derived for: every sphere of a
        f.px = f.py = f.pz = f.dx = f.dy = f.dz = 0.0f;   // visited leaf is a candidate, none is certain, t_lo = -inf
        f.Kg = 0.0f; f.c0 = __builtin_inff(); f.K = 0.0f;
        return;
    }
    const double u = 1.0 / 16777216.0;
    f.Kg = __double2float_ru(128.0 * u * M);
    f.c0 = __double2float_ru(8192.0 * u * u * M * M);
    f.K = __double2float_ru(24.0 * u * M);
}

// One visit of the walk: open `node`, test its four boxes, bound the spheres of the leaves the ray enters, push the
// interior children still in reach (nearest on top) and pop the next node (kNone: the walk is complete).
template <int STACK, bool SPILL, class RAY>
__device__ __forceinline__ void sphere_step(const float4 *__restrict__ nodes, const float4 *__restrict__ leaf_f32,
                                            const uint32_t *__restrict__ leaf_prims, const RAY &q, const SphereRay &sr,
                                            uint32_t &node, uint32_t &sp, uint32_t *lds_stack, uint32_t *lds_q,
                                            uint32_t tid, uint32_t *__restrict__ spill, uint32_t spill_entries,
                                            size_t spill_stride, size_t glane, float &best_up, uint32_t &qcnt, bool &overflow,
                                            uint32_t &nbox, uint32_t &nleaf)
{
    const float4 *np = nodes + 8 * (size_t)node;
    float4 ca[4], cb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { ca[c] = np[c]; cb[c] = np[4 + c]; }
    float tc[4];
    uint32_t lnk[4], cnt[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        tc[c] = box_entry32(ca[c], cb[c], q, best_up);
        lnk[c] = __float_as_uint(ca[c].w);
        cnt[c] = __float_as_uint(cb[c].w);
    }
    nbox += 4;
    uint32_t leafmask = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (tc[c] < __builtin_inff() && cnt[c] - 1u < 0xFFFFu) leafmask |= 1u << c;              // a sphere leaf the ray enters
    while (leafmask != 0u) {
        const uint32_t c = (uint32_t)__builtin_ctz(leafmask);
        leafmask &= leafmask - 1u;
        const uint32_t first = c == 0 ? lnk[0] : (c == 1 ? lnk[1] : (c == 2 ? lnk[2] : lnk[3]));
        const uint32_t n = (c == 0 ? cnt[0] : (c == 1 ? cnt[1] : (c == 2 ? cnt[2] : cnt[3]))) & 0xFFFFu;
        for (uint32_t k = 0; k < n; ++k) {
            const float4 rec = leaf_f32[first + k];                              // {c - centre, r}
            const float ox = rec.x - sr.px, oy = rec.y - sr.py, oz = rec.z - sr.pz;
            const float b = __builtin_fmaf(ox, sr.dx, __builtin_fmaf(oy, sr.dy, oz * sr.dz));
            const float lx = __builtin_fmaf(-b, sr.dx, ox), ly = __builtin_fmaf(-b, sr.dy, oy), lz = __builtin_fmaf(-b, sr.dz, oz);
            const float l2 = __builtin_fmaf(lx, lx, __builtin_fmaf(ly, ly, lz * lz));
            const float Dl = __builtin_fmaf(rec.w, rec.w, -l2);
            const float G = __builtin_fmaf(sr.Kg, rec.w, sr.c0);
            const float Dp = Dl + G;
            if (Dp >= 0.0f) {                                                     // the exact test cannot be excluded
                const float tlo = b - __builtin_amdgcn_sqrtf(Dp) * (1.0f + 4.76837158e-7f) - sr.K;
                const float Dm = Dl - G;
                // Delta > 0 for certain: t <= t_hi.  t_hi < 0 (the origin is inside the sphere or past it): the reference's
                // near root is negative and closest_object drops it (scene.rs:249) -- not a candidate at all
                const float thi = Dm > 0.0f ? b - __builtin_amdgcn_sqrtf(Dm) * (1.0f - 4.76837158e-7f) + sr.K : __builtin_inff();
                if (tlo <= best_up && !(thi < 0.0f)) {                            // (else it cannot be the winner)
                    if (tlo > sr.K) best_up = fminf(best_up, thi);                // certainly reported: bounds the winner's distance
                    if (qcnt == (uint32_t)kSphQueue) {                            // drop the entries a later certain hit has overtaken
                        uint32_t w = 0;
#pragma unroll
                        for (int e = 0; e < kSphQueue; ++e) {
                            const uint32_t ie = lds_q[(size_t)e * kBvhThreads + tid];
                            const uint32_t te = lds_q[(size_t)(kSphQueue + e) * kBvhThreads + tid];
                            if (__uint_as_float(te) <= best_up) {
                                lds_q[(size_t)w * kBvhThreads + tid] = ie;
                                lds_q[(size_t)(kSphQueue + w) * kBvhThreads + tid] = te;
                                w += 1;
                            }
                        }
                        qcnt = w;
                    }
                    if (qcnt == (uint32_t)kSphQueue) overflow = true;             // (the segment then tests every sphere exactly)
                    else {
                        lds_q[(size_t)qcnt * kBvhThreads + tid] = leaf_prims[first + k];
                        lds_q[(size_t)(kSphQueue + qcnt) * kBvhThreads + tid] = __float_as_uint(tlo);
                        qcnt += 1;
                    }
                }
            }
        }
        nleaf += n;
    }
    // interior children still in reach, nearest first (as bvh_step)
    float key[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) key[c] = (cnt[c] == 0u && tc[c] <= best_up) ? tc[c] : __builtin_inff();
#define RTX_CSWAP(i, j) { if (key[j] < key[i]) { float tk = key[i]; key[i] = key[j]; key[j] = tk; uint32_t tl = lnk[i]; lnk[i] = lnk[j]; lnk[j] = tl; } }
    RTX_CSWAP(0, 1) RTX_CSWAP(2, 3) RTX_CSWAP(0, 2) RTX_CSWAP(1, 3) RTX_CSWAP(1, 2)
#undef RTX_CSWAP
    const uint32_t npush = (key[1] < __builtin_inff() ? 1u : 0u) + (key[2] < __builtin_inff() ? 1u : 0u) +
                           (key[3] < __builtin_inff() ? 1u : 0u);
    if (sp + 3u <= (uint32_t)STACK) {
#pragma unroll
        for (uint32_t i = 1; i <= 3; ++i) {
            const uint32_t row = i <= npush ? sp + npush - i : (uint32_t)STACK;       // (row STACK = the sink)
            lds_stack[(size_t)row * kBvhThreads + tid] = lnk[i];
        }
        sp += npush;
    } else {
#define RTX_PUSH(v)                                                                                      \
        {                                                                                            \
            if (sp < (uint32_t)STACK) { lds_stack[(size_t)sp * kBvhThreads + tid] = (v); sp += 1; } \
            else if (SPILL && sp - (uint32_t)STACK < spill_entries) {                                \
                spill[(size_t)(sp - (uint32_t)STACK) * spill_stride + glane] = (v); sp += 1;         \
            } else overflow = true;                                                                  \
        }
        if (key[3] < __builtin_inff()) RTX_PUSH(lnk[3])
        if (key[2] < __builtin_inff()) RTX_PUSH(lnk[2])
        if (key[1] < __builtin_inff()) RTX_PUSH(lnk[1])
#undef RTX_PUSH
    }
    node = key[0] < __builtin_inff() ? lnk[0] : kNone;
    if (node == kNone && sp != 0u) {
        sp -= 1;                        // its boxes are re-tested against the current bound when it is opened
        node = (!SPILL || sp < (uint32_t)STACK) ? lds_stack[(size_t)sp * kBvhThreads + tid]
                                                : spill[(size_t)(sp - (uint32_t)STACK) * spill_stride + glane];
    }
}

// The same visit over the 64-byte nodes (rtx_bvh.h BvhQ3Node): 4 address-divergent requests instead of 8 -- the per-lane walk
// is bound by the L1's request rate before it is by the VALUs.  plane = o + q * s is exact (s a power of two, o a multiple of
// s), so the slab distance is t = fl(q * S + O) with S = s * inv (exact: a power of two times inv) and O = fl(o * inv + noi):
// against box_entry32's fl(b * inv + noi) that is ONE more rounding, of O, at most 2^-24 |o * inv + noi| -- in position terms
// 2^-24 (|o| + |ray origin|), which the second abs_pad the 64-byte boxes were built with covers (rtx_bvh.h build_q3nodes) and,
// for the part that scales with a far origin, Ray32S's slack.  Everything after the four entry distances is sphere_step's.
#ifndef RTX_Q3_SORT64
#define RTX_Q3_SORT64 1
#endif
template <int STACK, bool SPILL, class RAY>
__device__ __forceinline__ void sphere_step_q3(const float4 *__restrict__ qnodes, const float4 *__restrict__ leaf_f32,
                                               const uint32_t *__restrict__ leaf_prims, const RAY &q, const SphereRay &sr,
                                               uint32_t &node, uint32_t &sp, uint32_t *lds_stack, uint32_t *lds_q,
                                               uint32_t tid, uint32_t *__restrict__ spill, uint32_t spill_entries,
                                               size_t spill_stride, size_t glane, float &best_up, uint32_t &qcnt, bool &overflow,
                                               uint32_t &nbox, uint32_t &nleaf)
{
    const float4 *np = qnodes + 4 * (size_t)node;
    const float4 h0 = np[0], h1 = np[1], h2 = np[2], h3 = np[3];
    const float Sx = h0.w * q.ix, Sy = h1.x * q.iy, Sz = h1.y * q.iz;
    const float Ox = __builtin_fmaf(h0.x, q.ix, q.nx), Oy = __builtin_fmaf(h0.y, q.iy, q.ny), Oz = __builtin_fmaf(h0.z, q.iz, q.nz);
    const uint32_t lox = __float_as_uint(h1.z), loy = __float_as_uint(h1.w), loz = __float_as_uint(h2.x);
    const uint32_t hix = __float_as_uint(h2.y), hiy = __float_as_uint(h2.z), hiz = __float_as_uint(h2.w);
    const uint32_t lk[4] = { __float_as_uint(h3.x), __float_as_uint(h3.y), __float_as_uint(h3.z), __float_as_uint(h3.w) };
    const float e = ray_slack(q);
    float tc[4];
    uint32_t lnk[4], typ[4];
    // near / far plane of each slab by the sign of the ray's (clamped, never zero or infinite) reciprocal direction, chosen ONCE per
    // visit on the words that hold the four children's bytes: lo <= hi and the FMA is monotone in the plane, so fl(near * S + O) is
    // min(x0, x1) of the two-sided form bit for bit, without the 24 v_min / v_max
    const bool gx = q.ix < 0.0f, gy = q.iy < 0.0f, gz = q.iz < 0.0f;
    const uint32_t nxw = gx ? hix : lox, fxw = gx ? lox : hix;
    const uint32_t nyw = gy ? hiy : loy, fyw = gy ? loy : hiy;
    const uint32_t nzw = gz ? hiz : loz, fzw = gz ? loz : hiz;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float x0 = __builtin_fmaf((float)((nxw >> (8 * c)) & 255u), Sx, Ox), x1 = __builtin_fmaf((float)((fxw >> (8 * c)) & 255u), Sx, Ox);
        const float y0 = __builtin_fmaf((float)((nyw >> (8 * c)) & 255u), Sy, Oy), y1 = __builtin_fmaf((float)((fyw >> (8 * c)) & 255u), Sy, Oy);
        const float z0 = __builtin_fmaf((float)((nzw >> (8 * c)) & 255u), Sz, Oz), z1 = __builtin_fmaf((float)((fzw >> (8 * c)) & 255u), Sz, Oz);
        const float tn = fmaxf(fmaxf(x0, y0), fmaxf(z0, 0.0f));
        const float tf = fminf(fminf(x1, y1), z1);
        const float tn_lo = __builtin_fmaf(tn, 1.0f - 4.76837158e-7f, -e);
        const float tf_hi = __builtin_fmaf(tf, 1.0f + 4.76837158e-7f, e);
        typ[c] = lk[c] >> 29;
        lnk[c] = lk[c] & 0x1FFFFFFFu;
        tc[c] = (tn_lo <= tf_hi && tn_lo <= best_up && typ[c] != 7u) ? tn_lo : __builtin_inff();
    }
    nbox += 4;
    uint32_t leafmask = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (tc[c] < __builtin_inff() && typ[c] != 0u) leafmask |= 1u << c;                        // a sphere leaf the ray enters
    while (leafmask != 0u) {
        const uint32_t c = (uint32_t)__builtin_ctz(leafmask);
        leafmask &= leafmask - 1u;
        const uint32_t first = c == 0 ? lnk[0] : (c == 1 ? lnk[1] : (c == 2 ? lnk[2] : lnk[3]));
        const uint32_t n = c == 0 ? typ[0] : (c == 1 ? typ[1] : (c == 2 ? typ[2] : typ[3]));
        for (uint32_t k = 0; k < n; ++k) {
            const float4 rec = leaf_f32[first + k];                              // {c - centre, r}
            const float ox = rec.x - sr.px, oy = rec.y - sr.py, oz = rec.z - sr.pz;
            const float b = __builtin_fmaf(ox, sr.dx, __builtin_fmaf(oy, sr.dy, oz * sr.dz));
            const float lx = __builtin_fmaf(-b, sr.dx, ox), ly = __builtin_fmaf(-b, sr.dy, oy), lz = __builtin_fmaf(-b, sr.dz, oz);
            const float l2 = __builtin_fmaf(lx, lx, __builtin_fmaf(ly, ly, lz * lz));
            const float Dl = __builtin_fmaf(rec.w, rec.w, -l2);
            const float G = __builtin_fmaf(sr.Kg, rec.w, sr.c0);
            const float Dp = Dl + G;
            if (Dp >= 0.0f) {                                                     // the exact test cannot be excluded (sphere_step)
                const float tlo = b - __builtin_amdgcn_sqrtf(Dp) * (1.0f + 4.76837158e-7f) - sr.K;
                const float Dm = Dl - G;
                const float thi = Dm > 0.0f ? b - __builtin_amdgcn_sqrtf(Dm) * (1.0f - 4.76837158e-7f) + sr.K : __builtin_inff();
                if (tlo <= best_up && !(thi < 0.0f)) {
                    if (tlo > sr.K) best_up = fminf(best_up, thi);
                    if (qcnt == (uint32_t)kSphQueue) {
                        uint32_t w = 0;
#pragma unroll
                        for (int e2 = 0; e2 < kSphQueue; ++e2) {
                            const uint32_t ie = lds_q[(size_t)e2 * kBvhThreads + tid];
                            const uint32_t te = lds_q[(size_t)(kSphQueue + e2) * kBvhThreads + tid];
                            if (__uint_as_float(te) <= best_up) {
                                lds_q[(size_t)w * kBvhThreads + tid] = ie;
                                lds_q[(size_t)(kSphQueue + w) * kBvhThreads + tid] = te;
                                w += 1;
                            }
                        }
                        qcnt = w;
                    }
                    if (qcnt == (uint32_t)kSphQueue) overflow = true;
                    else {
                        lds_q[(size_t)qcnt * kBvhThreads + tid] = leaf_prims[first + k];
                        lds_q[(size_t)(kSphQueue + qcnt) * kBvhThreads + tid] = __float_as_uint(tlo);
                        qcnt += 1;
                    }
                }
            }
        }
        nleaf += n;
    }
    float key[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) key[c] = (typ[c] == 0u && tc[c] <= best_up) ? tc[c] : __builtin_inff();
#if RTX_Q3_SORT64
    // the (key, link) pairs as the high and low word of an f64: same sign, and for these keys (no NaN; +inf is a finite f64 pattern)
    // the same order as the f32 key's, ties by link -- so a compare-exchange is v_min_f64 + v_max_f64 instead of v_cmp + 4 v_cndmask.
    // (Written as instructions: the pairs are not canonical f64 values and fmin() would first quiet them.)
    double kd[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) kd[c] = __hiloint2double((int)__float_as_uint(key[c]), (int)lnk[c]);
#define RTX_CSWAP(i, j) { double lo_, hi_; rtx_minmax_f64_bits(kd[i], kd[j], lo_, hi_); kd[i] = lo_; kd[j] = hi_; }
    RTX_CSWAP(0, 1) RTX_CSWAP(2, 3) RTX_CSWAP(0, 2) RTX_CSWAP(1, 3) RTX_CSWAP(1, 2)
#undef RTX_CSWAP
#pragma unroll
    for (int c = 0; c < 4; ++c) { key[c] = __uint_as_float((uint32_t)__double2hiint(kd[c])); lnk[c] = (uint32_t)__double2loint(kd[c]); }
#else
#define RTX_CSWAP(i, j) { if (key[j] < key[i]) { float tk = key[i]; key[i] = key[j]; key[j] = tk; uint32_t tl = lnk[i]; lnk[i] = lnk[j]; lnk[j] = tl; } }
    RTX_CSWAP(0, 1) RTX_CSWAP(2, 3) RTX_CSWAP(0, 2) RTX_CSWAP(1, 3) RTX_CSWAP(1, 2)
#undef RTX_CSWAP
#endif
    const uint32_t npush = (key[1] < __builtin_inff() ? 1u : 0u) + (key[2] < __builtin_inff() ? 1u : 0u) +
                           (key[3] < __builtin_inff() ? 1u : 0u);
    if (sp + 3u <= (uint32_t)STACK) {
#pragma unroll
        for (uint32_t i = 1; i <= 3; ++i) {
            const uint32_t row = i <= npush ? sp + npush - i : (uint32_t)STACK;
            lds_stack[(size_t)row * kBvhThreads + tid] = lnk[i];
        }
        sp += npush;
    } else {
#define RTX_PUSH(v)                                                                                  \
        {                                                                                            \
            if (sp < (uint32_t)STACK) { lds_stack[(size_t)sp * kBvhThreads + tid] = (v); sp += 1; } \
            else if (SPILL && sp - (uint32_t)STACK < spill_entries) {                                \
                spill[(size_t)(sp - (uint32_t)STACK) * spill_stride + glane] = (v); sp += 1;         \
            } else overflow = true;                                                                  \
        }
        if (key[3] < __builtin_inff()) RTX_PUSH(lnk[3])
        if (key[2] < __builtin_inff()) RTX_PUSH(lnk[2])
        if (key[1] < __builtin_inff()) RTX_PUSH(lnk[1])
#undef RTX_PUSH
    }
    node = key[0] < __builtin_inff() ? lnk[0] : kNone;
    if (node == kNone && sp != 0u) {
        sp -= 1;
        node = (!SPILL || sp < (uint32_t)STACK) ? lds_stack[(size_t)sp * kBvhThreads + tid]
                                                : spill[(size_t)(sp - (uint32_t)STACK) * spill_stride + glane];
    }
}

template <int STACK, bool SPILL, class RAY>
__device__ __forceinline__ void bvh_traverse_spheres(const float4 *__restrict__ nodes, const float4 *__restrict__ leaf_f32,
                                                     const uint32_t *__restrict__ leaf_prims, const RAY &q, const SphereRay &sr,
                                                     uint32_t root, uint32_t *lds_stack, uint32_t *lds_q,
                                                     uint32_t tid, uint32_t *__restrict__ spill, uint32_t spill_entries,
                                                     size_t spill_stride, size_t glane, float &best_up, uint32_t &qcnt, bool &overflow,
                                                     uint32_t &nbox, uint32_t &nleaf)
{
    uint32_t sp = 0;
    uint32_t node = root;
    while (node != kNone)
        sphere_step<STACK, SPILL>(nodes, leaf_f32, leaf_prims, q, sr, node, sp, lds_stack, lds_q, tid, spill, spill_entries,
                                  spill_stride, glane, best_up, qcnt, overflow, nbox, nleaf);
}

// ---- the walk in two phases: node visits and leaf visits apart -----------------------------------------------------------------
// sphere_step_q3 handles the leaves a node's children turn out to be INSIDE the visit, under the mask of the lanes that have one.
// A lane has such a leaf in 8 % of its visits -- and a wave with 41 walking lanes therefore in 97 % of its iterations: the ~65
// instructions of leaf code (bounds, candidate queue) are issued in nearly every iteration for two or three lanes (measured,
// LAB_NOTEBOOK R3.11).  Here a leaf child is pushed like any other child -- the stack holds the 32-bit links, type bits and all,
// in entry order -- and its sphere records are bounded when it is POPPED.  An iteration of the wave is then either a node visit (for
// the lanes whose current entry is a node; the lanes holding a leaf wait) or a leaf visit (the other way round), chosen for the
// wave: leaves when at least `leaf_lanes` lanes hold one, or nobody holds a node.  Both kinds of code run with the lanes that
// need them, the node visit has no leaf code in it, and the leaves come in distance order with the nodes.
// Every leaf a ray enters is still bounded with the lane's own best_up before the walk ends: same candidates for the exact tests.
// lds_top / top_n (optional): an LDS copy of nodes [0, top_n) -- the top of the tree, which every walk passes through
// (collapse_to_bvh4 numbers it first).  A per-lane walk fetches its node as four address-divergent 16-byte requests and the CU's L1
// serves about one such request per cycle: with the lanes kept busy (the slot kernel) that rate, not the VALUs, bounds the walk
// (LAB_NOTEBOOK R4.3), and the top levels are half of a walk's visits.
template <int STACK, bool SPILL, class RAY>
__device__ __forceinline__ void sphere_node_step_q3(const float4 *__restrict__ qnodes, const RAY &q, uint32_t &node, uint32_t &sp,
                                                    uint32_t *lds_stack, uint32_t tid, uint32_t *__restrict__ spill,
                                                    uint32_t spill_entries, size_t spill_stride, size_t glane, float best_up,
                                                    bool &overflow, uint32_t &nbox, LdsF4Ptr lds_top = nullptr, uint32_t top_n = 0u)
{
    float4 h0, h1, h2, h3;
    if (lds_top != nullptr && node < top_n) {             // (an LDS-typed pointer: ds_read_b128, not a flat load of a selected address)
        LdsF4Ptr lp = lds_top + 4 * node;
        const PkF4 v0 = lp[0], v1 = lp[1], v2 = lp[2], v3 = lp[3];
        h0 = make_float4(v0.x, v0.y, v0.z, v0.w); h1 = make_float4(v1.x, v1.y, v1.z, v1.w);
        h2 = make_float4(v2.x, v2.y, v2.z, v2.w); h3 = make_float4(v3.x, v3.y, v3.z, v3.w);
    } else {
        const float4 *np = qnodes + 4 * (size_t)node;
        h0 = np[0]; h1 = np[1]; h2 = np[2]; h3 = np[3];
    }
    const float Sx = h0.w * q.ix, Sy = h1.x * q.iy, Sz = h1.y * q.iz;
    const float Ox = __builtin_fmaf(h0.x, q.ix, q.nx), Oy = __builtin_fmaf(h0.y, q.iy, q.ny), Oz = __builtin_fmaf(h0.z, q.iz, q.nz);
    const uint32_t lox = __float_as_uint(h1.z), loy = __float_as_uint(h1.w), loz = __float_as_uint(h2.x);
    const uint32_t hix = __float_as_uint(h2.y), hiy = __float_as_uint(h2.z), hiz = __float_as_uint(h2.w);
    const uint32_t lk[4] = { __float_as_uint(h3.x), __float_as_uint(h3.y), __float_as_uint(h3.z), __float_as_uint(h3.w) };
    const float e = ray_slack(q);
    const bool gx = q.ix < 0.0f, gy = q.iy < 0.0f, gz = q.iz < 0.0f;                 // (sphere_step_q3: near / far plane by the sign)
    const uint32_t nxw = gx ? hix : lox, fxw = gx ? lox : hix;
    const uint32_t nyw = gy ? hiy : loy, fyw = gy ? loy : hiy;
    const uint32_t nzw = gz ? hiz : loz, fzw = gz ? loz : hiz;
    double kd[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const float x0 = __builtin_fmaf((float)((nxw >> (8 * c)) & 255u), Sx, Ox), x1 = __builtin_fmaf((float)((fxw >> (8 * c)) & 255u), Sx, Ox);
        const float y0 = __builtin_fmaf((float)((nyw >> (8 * c)) & 255u), Sy, Oy), y1 = __builtin_fmaf((float)((fyw >> (8 * c)) & 255u), Sy, Oy);
        const float z0 = __builtin_fmaf((float)((nzw >> (8 * c)) & 255u), Sz, Oz), z1 = __builtin_fmaf((float)((fzw >> (8 * c)) & 255u), Sz, Oz);
        const float tn = fmaxf(fmaxf(x0, y0), fmaxf(z0, 0.0f));
        const float tf = fminf(fminf(x1, y1), z1);
        const float tn_lo = __builtin_fmaf(tn, 1.0f - 4.76837158e-7f, -e);
        const float tf_hi = __builtin_fmaf(tf, 1.0f + 4.76837158e-7f, e);
        // (one compare against the smaller of the two limits; type 7 is an empty slot: one select for both conditions)
        const bool enter = (tn_lo <= fminf(tf_hi, best_up)) & (lk[c] < 0xE0000000u);
        const float key = enter ? tn_lo : __builtin_inff();
        kd[c] = __hiloint2double((int)__float_as_uint(key), (int)lk[c]);
    }
    nbox += 4;
    // (key, link) pairs ordered as f64 values (sphere_step_q3)
#define RTX_CSWAP(i, j) { double lo_, hi_; rtx_minmax_f64_bits(kd[i], kd[j], lo_, hi_); kd[i] = lo_; kd[j] = hi_; }
    RTX_CSWAP(0, 1) RTX_CSWAP(2, 3) RTX_CSWAP(0, 2) RTX_CSWAP(1, 3) RTX_CSWAP(1, 2)
#undef RTX_CSWAP
    float key[4];
    uint32_t lnk[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { key[c] = __uint_as_float((uint32_t)__double2hiint(kd[c])); lnk[c] = (uint32_t)__double2loint(kd[c]); }
    const uint32_t npush = (key[1] < __builtin_inff() ? 1u : 0u) + (key[2] < __builtin_inff() ? 1u : 0u) +
                           (key[3] < __builtin_inff() ? 1u : 0u);
    if (sp + 3u <= (uint32_t)STACK) {
#pragma unroll
        for (uint32_t i = 1; i <= 3; ++i) {
            const uint32_t row = i <= npush ? sp + npush - i : (uint32_t)STACK;
            lds_stack[(size_t)row * kBvhThreads + tid] = lnk[i];
        }
        sp += npush;
    } else {
#define RTX_PUSH(v)                                                                                  \
        {                                                                                            \
            if (sp < (uint32_t)STACK) { lds_stack[(size_t)sp * kBvhThreads + tid] = (v); sp += 1; } \
            else if (SPILL && sp - (uint32_t)STACK < spill_entries) {                                \
                spill[(size_t)(sp - (uint32_t)STACK) * spill_stride + glane] = (v); sp += 1;         \
            } else overflow = true;                                                                  \
        }
        if (key[3] < __builtin_inff()) RTX_PUSH(lnk[3])
        if (key[2] < __builtin_inff()) RTX_PUSH(lnk[2])
        if (key[1] < __builtin_inff()) RTX_PUSH(lnk[1])
#undef RTX_PUSH
    }
    node = key[0] < __builtin_inff() ? lnk[0] : kNone;
    if (node == kNone && sp != 0u) {
        sp -= 1;
        node = (!SPILL || sp < (uint32_t)STACK) ? lds_stack[(size_t)sp * kBvhThreads + tid]
                                                : spill[(size_t)(sp - (uint32_t)STACK) * spill_stride + glane];
    }
}

// The leaf `ref` (type << 29 | first record, type = its 1..6 spheres): sphere_step's bounds for each record, then the next entry.
// The candidate queue is column `qcol` of an LDS array of 2 * kSphQueue rows, QS words per row (the lock-step kernels: the lane's
// own column of lds_q[.][kBvhThreads]; the slot kernel: the column of the ray slot the lane walks).
template <int STACK, bool SPILL, int QS>
__device__ __forceinline__ void sphere_leaf_step_at(const float4 *__restrict__ leaf_f32, const uint32_t *__restrict__ leaf_prims,
                                                    const SphereRay &sr, uint32_t &node, uint32_t &sp, uint32_t *lds_stack, uint32_t *lds_q,
                                                    uint32_t tid, uint32_t qcol, uint32_t *__restrict__ spill, size_t spill_stride, size_t glane,
                                                    float &best_up, uint32_t &qcnt, bool &overflow, uint32_t &nleaf)
{
    const uint32_t first = node & 0x1FFFFFFFu, n = node >> 29;
    for (uint32_t k = 0; k < n; ++k) {
        const float4 rec = leaf_f32[first + k];                              // {c - centre, r}
        const float ox = rec.x - sr.px, oy = rec.y - sr.py, oz = rec.z - sr.pz;
        const float b = __builtin_fmaf(ox, sr.dx, __builtin_fmaf(oy, sr.dy, oz * sr.dz));
        const float lx = __builtin_fmaf(-b, sr.dx, ox), ly = __builtin_fmaf(-b, sr.dy, oy), lz = __builtin_fmaf(-b, sr.dz, oz);
        const float l2 = __builtin_fmaf(lx, lx, __builtin_fmaf(ly, ly, lz * lz));
        const float Dl = __builtin_fmaf(rec.w, rec.w, -l2);
        const float G = __builtin_fmaf(sr.Kg, rec.w, sr.c0);
        const float Dp = Dl + G;
        if (Dp >= 0.0f) {                                                     // the exact test cannot be excluded (sphere_step)
            const float tlo = b - __builtin_amdgcn_sqrtf(Dp) * (1.0f + 4.76837158e-7f) - sr.K;
            const float Dm = Dl - G;
            const float thi = Dm > 0.0f ? b - __builtin_amdgcn_sqrtf(Dm) * (1.0f - 4.76837158e-7f) + sr.K : __builtin_inff();
            if (tlo <= best_up && !(thi < 0.0f)) {
                if (tlo > sr.K) best_up = fminf(best_up, thi);
                if (qcnt == (uint32_t)kSphQueue) {
                    uint32_t w = 0;
#pragma unroll
                    for (int e2 = 0; e2 < kSphQueue; ++e2) {
                        const uint32_t ie = lds_q[(size_t)e2 * QS + qcol];
                        const uint32_t te = lds_q[(size_t)(kSphQueue + e2) * QS + qcol];
                        if (__uint_as_float(te) <= best_up) {
                            lds_q[(size_t)w * QS + qcol] = ie;
                            lds_q[(size_t)(kSphQueue + w) * QS + qcol] = te;
                            w += 1;
                        }
                    }
                    qcnt = w;
                }
                if (qcnt == (uint32_t)kSphQueue) overflow = true;
                else {
                    lds_q[(size_t)qcnt * QS + qcol] = leaf_prims[first + k];
                    lds_q[(size_t)(kSphQueue + qcnt) * QS + qcol] = __float_as_uint(tlo);
                    qcnt += 1;
                }
            }
        }
    }
    nleaf += n;
    node = kNone;
    if (sp != 0u) {
        sp -= 1;
        node = (!SPILL || sp < (uint32_t)STACK) ? lds_stack[(size_t)sp * kBvhThreads + tid]
                                                : spill[(size_t)(sp - (uint32_t)STACK) * spill_stride + glane];
    }
}

template <int STACK, bool SPILL>
__device__ __forceinline__ void sphere_leaf_step(const float4 *__restrict__ leaf_f32, const uint32_t *__restrict__ leaf_prims,
                                                 const SphereRay &sr, uint32_t &node, uint32_t &sp, uint32_t *lds_stack, uint32_t *lds_q,
                                                 uint32_t tid, uint32_t *__restrict__ spill, size_t spill_stride, size_t glane,
                                                 float &best_up, uint32_t &qcnt, bool &overflow, uint32_t &nleaf)
{
    sphere_leaf_step_at<STACK, SPILL, kBvhThreads>(leaf_f32, leaf_prims, sr, node, sp, lds_stack, lds_q, tid, tid, spill, spill_stride, glane,
                                                   best_up, qcnt, overflow, nleaf);
}

#if defined(RTX_LAB) && defined(RTX_SPH_PROFILE)          // lab build (-DRTX_SPH_PROFILE=k): one per-lane count per build, summed and reported through exact_tests
#define RTX_PROF_ARG , unsigned long long &rtx_prof
#else
#define RTX_PROF_ARG
#endif
// The resumable walk (below) in this form.  `node` is a link: type 0 = a node, 1..6 = a leaf; kNone = the walk has ended.
// (Measured and dropped, LAB_NOTEBOOK R3.11: a lane that pops a leaf putting it aside in a register and going on with the next
// entry instead of waiting for the wave's next leaf visit -- 54.8 against 54.6 ms, 1 % more box tests from the later best_up.)
template <int STACK, bool SPILL, class RAY>
__device__ __forceinline__ void sphere_walk_phased(const float4 *__restrict__ qnodes, const float4 *__restrict__ leaf_f32,
                                                   const uint32_t *__restrict__ leaf_prims, const RAY &q, const SphereRay &sr,
                                                   uint32_t &node, uint32_t &sp, uint32_t *lds_stack, uint32_t *lds_q,
                                                   uint32_t tid, uint32_t *__restrict__ spill, uint32_t spill_entries,
                                                   size_t spill_stride, size_t glane, float &best_up, uint32_t &qcnt, bool &overflow,
                                                   uint32_t &nbox, uint32_t &nleaf, uint32_t cut_walkers, uint32_t cut_done, uint32_t n_alive,
                                                   uint32_t leaf_lanes RTX_PROF_ARG)
{
    while (node != kNone) {
        const bool at_leaf = (node >> 29) != 0u;
        const unsigned long long lm = __ballot(at_leaf), am = __ballot(true);
#if defined(RTX_LAB) && defined(RTX_SPH_PROFILE)
        {   // lab build: 3 node-visit iterations, 4 lanes in them, 5 leaf-visit iterations, 6 lanes in them, 7 lanes in the loop
            const bool leafv = (uint32_t)__popcll(lm) >= leaf_lanes || lm == am;
            const bool leader = (uint32_t)(__ffsll((long long)am) - 1) == (tid & 63u);
            if (RTX_SPH_PROFILE == 3 && !leafv && leader) rtx_prof += 1;
            if (RTX_SPH_PROFILE == 4 && !leafv && !at_leaf) rtx_prof += 1;
            if (RTX_SPH_PROFILE == 5 && leafv && leader) rtx_prof += 1;
            if (RTX_SPH_PROFILE == 6 && leafv && at_leaf) rtx_prof += 1;
            if (RTX_SPH_PROFILE == 7) rtx_prof += 1;
        }
#endif
#if defined(RTX_LAB) && defined(RTX_SPH_PROFILE) && RTX_SPH_PROFILE == 9
        const uint32_t sp_before = sp;              // lab build: pushes that take the stack above RTX_SPH_PROFILE_SP entries (what a shorter LDS stack would spill)
#endif
        if ((uint32_t)__popcll(lm) >= leaf_lanes || lm == am) {
            if (at_leaf)
                sphere_leaf_step<STACK, SPILL>(leaf_f32, leaf_prims, sr, node, sp, lds_stack, lds_q, tid, spill, spill_stride, glane,
                                               best_up, qcnt, overflow, nleaf);
        } else if (!at_leaf) {
            sphere_node_step_q3<STACK, SPILL>(qnodes, q, node, sp, lds_stack, tid, spill, spill_entries, spill_stride, glane, best_up,
                                              overflow, nbox);
        }
#if defined(RTX_LAB) && defined(RTX_SPH_PROFILE) && RTX_SPH_PROFILE == 9
        if (sp_before <= (uint32_t)(RTX_SPH_PROFILE_SP) && sp > (uint32_t)(RTX_SPH_PROFILE_SP)) rtx_prof += 1;
#endif
        const uint32_t still = (uint32_t)__popcll(__ballot(node != kNone));
        if (still < cut_walkers && n_alive - still >= cut_done) break;
    }
}

// The same walks, resumable: `node` / `sp` (and the caller's best_up, qcnt, nbox, nleaf, the lane's LDS stack and candidate
// columns) are the whole state of a walk, so a wave may leave the loop while a few lanes are still in it and come back to it
// after the f64 phase of the others.  The loop is left when fewer than `cut_walkers` of its lanes still walk and at least
// `cut_done` of the wave's `n_alive` rays wait for their f64 phase (wave-uniform: the ballot is over the lanes still in the
// loop); cut_walkers = 0 never cuts.  A round of a lock-step wave lasts as long as its longest walk -- ~55 visits against 21
// on average (lane utilisation 0.39) --: cutting the tail lets the many finished lanes go on while the few long walks
// continue, a visit later, beside the next segments.
template <int STACK, bool SPILL, bool Q3, class RAY>
__device__ __forceinline__ void sphere_walk_resumable(const float4 *__restrict__ nodes, const float4 *__restrict__ leaf_f32,
                                                      const uint32_t *__restrict__ leaf_prims, const RAY &q, const SphereRay &sr,
                                                      uint32_t &node, uint32_t &sp, uint32_t *lds_stack, uint32_t *lds_q,
                                                      uint32_t tid, uint32_t *__restrict__ spill, uint32_t spill_entries,
                                                      size_t spill_stride, size_t glane, float &best_up, uint32_t &qcnt, bool &overflow,
                                                      uint32_t &nbox, uint32_t &nleaf, uint32_t cut_walkers, uint32_t cut_done, uint32_t n_alive)
{
    while (node != kNone) {
        if constexpr (Q3)
            sphere_step_q3<STACK, SPILL>(nodes, leaf_f32, leaf_prims, q, sr, node, sp, lds_stack, lds_q, tid, spill, spill_entries,
                                         spill_stride, glane, best_up, qcnt, overflow, nbox, nleaf);
        else
            sphere_step<STACK, SPILL>(nodes, leaf_f32, leaf_prims, q, sr, node, sp, lds_stack, lds_q, tid, spill, spill_entries,
                                      spill_stride, glane, best_up, qcnt, overflow, nbox, nleaf);
        const uint32_t still = (uint32_t)__popcll(__ballot(node != kNone));
        if (still < cut_walkers && n_alive - still >= cut_done) break;
    }
}

template <int STACK, bool SPILL, class RAY>
__device__ __forceinline__ void bvh_traverse_spheres_q3(const float4 *__restrict__ qnodes, const float4 *__restrict__ leaf_f32,
                                                        const uint32_t *__restrict__ leaf_prims, const RAY &q, const SphereRay &sr,
                                                        uint32_t root, uint32_t *lds_stack, uint32_t *lds_q,
                                                        uint32_t tid, uint32_t *__restrict__ spill, uint32_t spill_entries,
                                                        size_t spill_stride, size_t glane, float &best_up, uint32_t &qcnt, bool &overflow,
                                                        uint32_t &nbox, uint32_t &nleaf)
{
    uint32_t sp = 0;
    uint32_t node = root;
    while (node != kNone)
        sphere_step_q3<STACK, SPILL>(qnodes, leaf_f32, leaf_prims, q, sr, node, sp, lds_stack, lds_q, tid, spill, spill_entries,
                                     spill_stride, glane, best_up, qcnt, overflow, nbox, nleaf);
}

// A whole traversal of one lane's segment: steps until the stack is empty, the queued candidates' exact f64 tests
// every 4th step (and at the end) for all lanes together, so their cost is not paid per (step, child, shape) under
// divergence -- the pruning bound lags by at most 4 steps, which only costs visits.
template <bool TRIS, bool SPILL, class RAY>
__device__ __forceinline__ void bvh_traverse(const float4 *__restrict__ nodes, const LeafArrays &la, const RAY &q,
                                             const FilterParams &fpar, const TriFilterParams &tpar, const RayX &rx,
                                             uint32_t root, bool &overflow, Hit &h, uint32_t *lds_stack, uint32_t *lds_q, uint32_t tid,
                                             uint32_t *__restrict__ spill, uint32_t spill_entries, size_t spill_stride, size_t glane,
                                             unsigned long long &box_tests, unsigned long long &leaf_filters, unsigned long long &exact,
                                             unsigned long long &wave_steps)
{
    float best_up = __builtin_inff();
    uint32_t sp = 0, qcnt = 0, step = 0, nbox = 0, nleaf = 0;
    uint32_t node = root;                    // wide node 0 (with kBvhFlatNode when it is a footprint node)
    while (node != kNone) {
#ifdef RTX_BVH_STATS
        { const unsigned long long am = __ballot(true); if ((tid & 63u) == (uint32_t)(__ffsll((long long)am) - 1)) wave_steps += 1; }
#endif
        bvh_step<TRIS, SPILL>(nodes, la, q, fpar, tpar, rx, node, sp, qcnt, overflow, h, best_up, lds_stack, lds_q, tid, spill,
                              spill_entries, spill_stride, glane, nbox, nleaf, exact);
        step += 1;
        if ((step & 3u) == 0u) flush_candidates<TRIS>(la, rx, lds_q, tid, qcnt, h, best_up, exact);
    }
    box_tests += nbox;
    leaf_filters += nleaf;
    flush_candidates<TRIS>(la, rx, lds_q, tid, qcnt, h, best_up, exact);
}

}  // namespace rtx
