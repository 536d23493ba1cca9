// rtx_device.h -- kernel-side views of an uploaded scene and the per-ray path logic shared by the
// EXACT and MIXED trace kernels (primary-ray generation, shading, bounce direction).
#pragma once

#include "rtx_bvh.h"
#include "rtx_scene.h"

namespace rtx {

// Everything a trace kernel needs, passed by value as a kernel argument.
struct SceneView {
    // Config (scene.rs:16-28) + seed
    uint64_t rays_per_pixel;
    uint64_t max_bounces;
    double   focal_length, focal_offset, non_focal_offset;
    uint64_t seed;
    // Camera (camera.rs:7-15): what render() reads
    V3 cam_pos;
    V3 to_world_x, to_world_y, to_world_z;     // rows of to_world_space (mat.rs:11-18)
    // shapes, segregated by type; ids give the index in Scene.objects (tie-break, scene.rs:250)
    uint32_t n_objects, n_spheres, n_planes, n_tris;
    const SphereX  *spheres;
    const uint32_t *sphere_id;
    const PlaneX   *planes;
    const TriX     *tris;
    const MaterialX *materials;                // by scene index
    // f32 filter records for the MIXED kernel
    // f32 filter records of the triangles that can be hit at all (MIXED kernel's triangle sweep): 2 x float4 each,
    // {n.xyz, n.(v0 - centre)} {cx, cy, hx, hy} = unit normal, plane offset, centre/half-size of the triangle's box in
    // the (x, y) projection Triangle::contains solves in (triangle.rs:55-100); tri_fidx maps a record to tris[]
    const float4   *tri_f32;
    const uint32_t *tri_fidx;
    const float4   *tri_geo;                   // per TREE record (the first n_tri_tree), 2 x float4: {v0.uv - centre, m00, m01} {m10, m11, n.v0, plane}
                                               // with (a, b) = M (q - v0)_uv in the footprint's plane (0: xy, 1: xz, 2: yz): the f32
                                               // certain-hit bounds of rtx_mesh_step.h
    uint32_t        n_tri_filter;
    uint32_t        pad0_;
    double          tri_extent;                // max over filtered triangles of |vertex - centre|_inf
    const float4   *sphere_f32;                // pair-interleaved {x0,x1,y0,y1},{z0,z1,w0,w1}; c - centre, w = |c|^2 - r*r;
                                               // padded to a multiple of 4 spheres
    double          sphere_center[3];          // centre of the bounding box of sphere centres and triangle vertices
    double          sphere_cmax;               // max over spheres of |c - centre| + r
    // flat BVH over the sphere boxes and triangle footprints (rtx_bvh.h); null when the scene has none
    const Bvh4Node *bvh_nodes;                 // 4-wide nodes, 128 B each; node 0 is the root
    const BvhQNode *bvh_qnodes;                // the same nodes in the 64-byte quantised form (bvh_flags bit 3), else null
    const BvhQ3Node *bvh_q3nodes;              // a sphere tree's nodes in their 64-byte form (bvh_flags bit 4), else null
    const uint32_t *bvh_prims;                 // local sphere indices, leaf-contiguous
    const float4   *bvh_leaf_f32;              // per sphere leaf entry: the sphere's filter record {c - centre, |c - centre|^2 - r^2}
    const float4   *bvh_leaf_cr;               // the same entries as {c - centre, |r|} (rounded up): the spheres kernel's record
    uint32_t        n_bvh_nodes;
    uint32_t        bvh_root;                  // where a traversal starts: 0, or kBvhFlatNode when node 0 is a footprint node
    uint32_t        bvh_depth;
    uint32_t        tuning;                    // RtxConfig.tuning (RTX_TUNE_* bits, include/rtx_hip.h): read by the host launchers only
    float           bvh_origin_limit;          // the f32 slab test is valid for ray origins with |o|_inf <= this
    uint32_t        bvh_flags;                 // bit 3: bvh_qnodes is valid;  bit 2: the tree holds nothing but (x, y)-footprint triangles (every node is a footprint node);
                                               // bit 0: the spheres are in the tree, bit 1: the first n_tri_tree filter records are
    uint32_t        n_tri_tree;                // triangle filter records [0, n_tri_tree) are in leaf order (a triangle leaf's link
    float           bvh_inv_max;               // indexes them); [n_tri_tree, n_tri_filter) are outside the tree.  bvh_inv_max: rtx_traverse.h
};

// Which pixels/samples one launch covers.
// n / d for a divisor that is fixed for the launch: q = (t + ((n - t) >> s1)) >> s2 with t = mulhi(m, n) -- exact for every 32-bit n
// (Granlund & Montgomery's round-up multiplier; l = ceil(log2 d), m = floor(2^32 (2^l - d) / d) + 1, s1 = min(l, 1), s2 = max(l - 1, 0)).
// Five integer instructions instead of the ~30 of an emulated 32-bit division (~150 for the 64-bit form): a ray's index -> pixel
// arithmetic has four of them, once per primary ray and once per refill of the queue-fed kernels.
struct FastDiv { uint32_t m, s1, s2, d; };

inline FastDiv make_fastdiv(uint32_t d)
{
    FastDiv f;
    f.d = d ? d : 1u;
    uint32_t l = 0;
    while (l < 32u && (1ull << l) < (unsigned long long)f.d) ++l;
    f.m = (uint32_t)((((1ull << l) - (unsigned long long)f.d) << 32) / (unsigned long long)f.d + 1ull);
    f.s1 = l < 1u ? l : 1u;
    f.s2 = l > 1u ? l - 1u : 0u;
    return f;
}

RTX_HD uint32_t fastdiv(uint32_t n, const FastDiv &f)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t t = __umulhi(f.m, n);
#else
    const uint32_t t = (uint32_t)(((unsigned long long)f.m * (unsigned long long)n) >> 32);
#endif
    return (t + ((n - t) >> f.s1)) >> f.s2;
}

struct RowsView {
    uint32_t width, height;
    uint32_t row_begin, row_stride, n_rows;       // local row k is image row image_row(rv, k)
    uint32_t row_block;            // rows per block of the band (1: single interleaved rows; 8: whole 8x8 ray tiles)
    uint32_t npix;                 // n_rows * width  (local pixels)
    uint32_t sample_begin;         // first sample index of this batch
    uint32_t n_samples;            // samples in this batch
    uint64_t n_rays;               // npix * n_samples
    uint32_t grab;                 // rays a wave of the BVH kernels takes from the global queue per atomic: 512 for big
    uint32_t tiles_x;              // launches, less when that would leave fewer than ~8 grabs per wave (the last grabs
                                   // decide how long the slowest wave runs).  tiles_x != 0: the BVH kernels' ray index
                                   // runs over 8x8 pixel tiles (a wave's 64 rays = one tile), n_rays counts the padded grid
    // host-libm trig tables (scene.rs:214-220): sin/cos(fov*(x/w-0.5)) per column,
    // sin/cos(vfov*(y/h-0.5)) per LOCAL row
    const double *sin_x, *cos_x, *sin_y, *cos_y;
    FastDiv div_width, div_row_block, div_npix, div_tiles_x, div_per_sample;    // per_sample = tiles_x * tiles_y * 64 (the host keeps n_rays < 2^32)
};

// Local row k of a band -> image row.  The band is made of blocks of row_block consecutive image rows, row_stride rows
// apart (block b of the band starts at row_begin + b * row_stride); row_block == 1 is the single-row form
// row_begin + k * row_stride.  With row_block == 8 an 8x8 tile of the BVH kernels' ray queue (8 LOCAL rows) is 8x8
// neighbouring pixels whatever the number of bands -- with single interleaved rows it is 8 columns x 8n image rows.
RTX_HD uint32_t image_row(const RowsView &rv, uint32_t k)
{
    const uint32_t b = fastdiv(k, rv.div_row_block);
    return rv.row_begin + b * rv.row_stride + (k - b * rv.row_block);
}

// One segment of a path's transcript (rtx_debug_paths, lab library): the ray as closest_object saw it, the winning distance and the
// winner's index in Scene.objects (-1 and +inf: the ray left the scene).  Same layout as RtxPathStep (include/rtx_hip.h).
struct PathStep {
    double pos[3], dir[3], t;
    long long object;
};

// Sharded counters (one slot per wave-id hash) summed on the host.
struct Counters {
    unsigned long long segments;
    unsigned long long exact_tests;
    unsigned long long filter_tests;
    unsigned long long pad_;
};
constexpr int kCounterShards = 64;

struct RayState {                  // raytracing/ray.rs:4-21 + RNG cursor
    V3 pos, dir, result, light;
    uint64_t key;
    uint32_t draw;
    uint32_t bounce;
};

struct Hit {
    double   t;
    uint32_t id;                   // scene index of the winner; 0xFFFFFFFF = none
    uint32_t kind;                 // RTX_SPHERE/PLANE/TRIANGLE
    uint32_t local;                // index inside its type array
};

__device__ __forceinline__ void hit_init(Hit &h) { h.t = 0.0; h.id = 0xFFFFFFFFu; h.kind = 0; h.local = 0; }

// closest_object's filter + min_by (scene.rs:249-250): keep is_normal && positive; the FIRST
// minimal element in Scene.objects order wins -> lexicographic (t, id).
__device__ __forceinline__ void hit_consider(Hit &h, double t, uint32_t id, uint32_t kind, uint32_t local)
{
    if (!is_normal_positive(t)) return;
    if (h.id == 0xFFFFFFFFu || t < h.t || (t == h.t && id < h.id)) { h.t = t; h.id = id; h.kind = kind; h.local = local; }
}

// map a ray index of the launch to (local pixel, sample), pixel-fastest so that a wave's 64 rays
// are 64 neighbouring pixels of one sample (coalesced sample-plane writes).
__device__ __forceinline__ void ray_index_to_pixel(const RowsView &rv, uint64_t i, uint32_t &pl, uint32_t &s_local)
{
    s_local = fastdiv((uint32_t)i, rv.div_npix);              // (i < n_rays < 2^32)
    pl = (uint32_t)i - s_local * rv.npix;
}

// The same over 8x8 pixel tiles: 64 consecutive indices = one tile of one sample (more coherent primary rays than a
// 64x1 strip: shared nodes, similar traversal lengths).  Returns false for the padding of partial tiles.
__device__ __forceinline__ bool ray_index_to_pixel_tiled(const RowsView &rv, uint64_t i, uint32_t &pl, uint32_t &s_local)
{
    s_local = fastdiv((uint32_t)i, rv.div_per_sample);        // (i < n_rays < 2^32)
    const uint32_t r = (uint32_t)i - s_local * rv.div_per_sample.d;
    const uint32_t t = r >> 6, j = r & 63u;
    const uint32_t ty = fastdiv(t, rv.div_tiles_x), tx = t - ty * rv.tiles_x;
    const uint32_t x = tx * 8u + (j & 7u), k = ty * 8u + (j >> 3);
    pl = k * rv.width + x;
    return x < rv.width && k < rv.n_rows;
}

// Samples: one 32-byte record {r, g, b, 0} per ray, in RAY-QUEUE order (rv.n_rays of them, the padding of partial
// tiles included).  A ray finishes on its own lane at its own time, so its store is a lone lane's: the record is one
// whole 32-byte sector, which is what the memory side writes anyway.  Measured on C2 at 64 spp (WRITE_SIZE per launch
// for 3.19 GB of samples): [pixel][3] doubles, 24 B across two sectors: 9.4 GB; three 8-byte planes: 13.6 GB (one
// sector per 8-byte store -- L2 does not merge them before they leave); this layout: see profiles/.
__device__ __forceinline__ void store_sample(double *__restrict__ samples, const RowsView &rv, uint64_t ridx, V3 c)
{
    (void)rv;
    double4 *rec = reinterpret_cast<double4 *>(samples) + ridx;
    *rec = make_double4(c.x, c.y, c.z, 0.0);
}

// focal_point of a local pixel: get_ray_dir (scene.rs:213-222, with the host-computed trig values) and scene.rs:203
__device__ __forceinline__ V3 primary_focal_point(const SceneView &sv, const RowsView &rv, uint32_t pl)
{
    const uint32_t k = fastdiv(pl, rv.div_width);        // local row
    const uint32_t x = pl - k * rv.width;
    const V3 cam_space_dir = mk(rv.sin_x[x], rv.sin_y[k], rv.cos_x[x] * rv.cos_y[k]);                   // scene.rs:216-221
    const V3 ray_dir = mk(dot(cam_space_dir, sv.to_world_x), dot(cam_space_dir, sv.to_world_y),
                          dot(cam_space_dir, sv.to_world_z));                                            // mat/mul.rs:42-50
    return vadd(sv.cam_pos, vmuls(ray_dir, sv.focal_length));                                            // scene.rs:203
}

__device__ __forceinline__ float round_down_f32_dev(double x)
{
    float f = (float)x;
    if ((double)f > x) f = __uint_as_float(f > 0.0f ? __float_as_uint(f) - 1u : (f < 0.0f ? __float_as_uint(f) + 1u : 0x80000001u));
    return f;
}

// render_pixel's per-sample prologue (scene.rs:196-207) + get_ray_dir (scene.rs:213-222).
__device__ __forceinline__ void gen_primary(const SceneView &sv, const RowsView &rv, uint32_t pl, uint32_t sample,
                                            RayState &r)
{
    uint32_t k = fastdiv(pl, rv.div_width);        // local row
    uint32_t x = pl - k * rv.width;
    uint32_t y = image_row(rv, k);
    uint64_t pix = (uint64_t)y * rv.width + x;     // index in the FULL image keys the RNG
    r.key = rng_key(sv.seed, pix, sample);
    V3 rnd1;                                                             // vector.rs:29-35: x, y, z in order
    rnd1.x = rng_u01(r.key, 0); rnd1.y = rng_u01(r.key, 1); rnd1.z = rng_u01(r.key, 2);
    V3 ray_position = vadd(sv.cam_pos, vmuls(rnd1, sv.non_focal_offset));      // scene.rs:202
    V3 focal_point = primary_focal_point(sv, rv, pl);                          // scene.rs:203, 213-222
    V3 rnd2;
    rnd2.x = rng_u01(r.key, 3); rnd2.y = rng_u01(r.key, 4); rnd2.z = rng_u01(r.key, 5);
    V3 target_point = vadd(focal_point, vmuls(rnd2, sv.focal_offset));         // scene.rs:204
    V3 ray_direction = vsub(target_point, ray_position);                       // scene.rs:205
    r.pos = ray_position;
    r.dir = vnorm(ray_direction);                                              // scene.rs:207
    r.result = mk(0.0, 0.0, 0.0);                                              // ray.rs:18
    r.light = mk(1.0, 1.0, 1.0);                                               // ray.rs:19
    r.draw = 6;
    r.bounce = 0;
}

// Vector3::random_direction (vector.rs:36-45)
__device__ __forceinline__ V3 random_direction(double u_z, double u_theta)
{
    double z = u_z * 2.0 - 1.0;
    double theta = u_theta * 2.0 * 3.14159265358979323846;
    double r = sqrt(1.0 - z * z);
    double sn, cs;
    sincos_2pi(theta, &sn, &cs);                                   // (rtx_math.h: theta is in [0, 2 pi))
    return vnorm(mk(r * cs, r * sn, z));
}

// random_bounce_dir (scene.rs:279-292)
__device__ __forceinline__ V3 random_bounce_dir(V3 ray_dir, V3 surface_normal, double surface_roughness,
                                                double u_z, double u_theta)
{
    V3 random_dir = random_direction(u_z, u_theta);
    V3 reflection_dir = vsub(ray_dir, vmuls(vmuls(surface_normal, 2.0), dot(ray_dir, surface_normal)));
    V3 random_to_reflection_dir = vsub(reflection_dir, random_dir);
    double reflection_mult = 1.0 - surface_roughness;
    V3 final_direction = vadd(random_dir, vmuls(random_to_reflection_dir, reflection_mult));
    final_direction = vnorm(final_direction);
    if (dot(final_direction, surface_normal) > 0.0) return final_direction;
    return vneg(final_direction);
}

// Object::normal_at of the winning object (object.rs:37-39)
__device__ __forceinline__ V3 normal_at(const SceneView &sv, const Hit &h, V3 world_pos)
{
    if (h.kind == 0) return sphere_normal_at(sv.spheres[h.local], world_pos);
    if (h.kind == 1) return sv.planes[h.local].nhat;
    return sv.tris[h.local].nn;
}

// render_ray's Some((dst, obj)) arm (scene.rs:233-236) + ray_hit (scene.rs:260-278)
__device__ __forceinline__ void advance_and_shade(const SceneView &sv, const Hit &h, RayState &r)
{
    r.pos = vadd(r.pos, vmuls(r.dir, h.t));                                    // scene.rs:234
    const MaterialX m = sv.materials[h.id];
    double u_z = rng_u01(r.key, r.draw);                                       // vector.rs:37
    double u_theta = rng_u01(r.key, r.draw + 1);                               // vector.rs:38
    r.draw += 2;
    r.dir = random_bounce_dir(r.dir, normal_at(sv, h, r.pos), m.roughness, u_z, u_theta);  // scene.rs:275
    r.result = vadd(r.result, vmulv(r.light, m.emission_color));               // scene.rs:276
    r.light = vmulv(r.light, m.base_color);                                    // scene.rs:277
    r.bounce += 1;
}

__device__ __forceinline__ bool light_is_zero(const RayState &r)               // scene.rs:228
{
    return r.light.x == 0.0 && r.light.y == 0.0 && r.light.z == 0.0;
}

// ---- conservative f32 sphere filter (trace_mixed_kernel's sweep, trace_bvh_kernel's leaves); see rtx_kernels.hip
struct FilterParams { float dx, dy, dz, npd, p2x, p2y, p2z, nppE; };

__device__ __forceinline__ void filter_idle(FilterParams &f)      // D = -3e30 - w < 0: nothing passes
{
    f.dx = f.dy = f.dz = f.npd = f.p2x = f.p2y = f.p2z = 0.0f;
    f.nppE = -3.0e30f;
}

__device__ __forceinline__ void filter_pass_all(FilterParams &f)  // D = 1e30 - w >= 0: every record is a candidate
{
    f.dx = f.dy = f.dz = f.npd = f.p2x = f.p2y = f.p2z = 0.0f;
    f.nppE = 1.0e30f;
}

// Magnitudes outside 1e-12 < M < 1e14 could overflow/underflow the f32 products: such a ray gets the
// pass-all filter (its queue overflows and the slot takes the exact f64 sweep).  Inside the range every
// filter operation stays finite, so the sweep's sign-bit test is exact: D >= 0 <=> sign bit clear.
__device__ __forceinline__ void filter_from_ray(const SceneView &sv, V3 pos, V3 dir, FilterParams &f)
{
    // centre the origin (better conditioned f32 products); all in f64, then one rounding each
    double px = pos.x - sv.sphere_center[0];
    double py = pos.y - sv.sphere_center[1];
    double pz = pos.z - sv.sphere_center[2];
    double pp = px * px + py * py + pz * pz;
    double pd = px * dir.x + py * dir.y + pz * dir.z;
    // error bound: |D_f32 - D| <= 24 * 2^-24 * M^2,  M = max(|c| + r) + |p|  (DESIGN.md 3.1); E = 64 * 2^-24 * M^2
    double M = sv.sphere_cmax + sqrt(pp);
    if (!(M < 1.0e14) || !(M > 1.0e-12)) { filter_pass_all(f); return; }
    double E = M * M * (64.0 / 16777216.0);
    f.dx = (float)dir.x; f.dy = (float)dir.y; f.dz = (float)dir.z;
    f.npd = (float)(-pd);
    f.p2x = (float)(2.0 * px); f.p2y = (float)(2.0 * py); f.p2z = (float)(2.0 * pz);
    f.nppE = (float)(E - pp);
}

// ---- conservative f32 triangle filter -----------------------------------------------------------------------
// Triangle::distance (triangle.rs:108-127) reports a hit only if the point q = p + dir * |n.(v0-p) / n.d| has its
// (x, y) projection inside the projected triangle (rows x, y of the elimination; upload keeps other pivot rows and
// ill-conditioned projections as "always candidate").  Inside the triangle implies inside its projected bounding
// box |q_x - c_x| <= h_x, |q_y - c_y| <= h_y.  Multiplied by |n.d| the test needs no division:
//     e_x = (p_x - c_x) |n.d| + d_x |n.(v0 - p)|,      candidate  <=>  |e_x| <= h_x |n.d| + A   (same for y)
// With u = 2^-24 and S >= every coordinate magnitude (relative to the scene centre), the f32 evaluation of e_x is
// off by <= 31 u S and of the right-hand side by <= 13 u S; A = 64 u S.  The cull test (triangle.rs:115) is not
// evaluated here: triangles it always rejects were dropped at upload, the others are left to the exact test.
struct TriFilterParams { float dx, dy, dz, npx, npy, npz, A, pad; };

__device__ __forceinline__ void tri_filter_idle(TriFilterParams &f)
{
    f.dx = f.dy = f.dz = f.npx = f.npy = f.npz = f.pad = 0.0f;
    f.A = -3.0e30f;
}

__device__ __forceinline__ void tri_filter_pass_all(TriFilterParams &f)
{
    f.dx = f.dy = f.dz = f.npx = f.npy = f.npz = f.pad = 0.0f;
    f.A = 1.0e30f;
}

__device__ __forceinline__ void tri_filter_from_ray(const SceneView &sv, V3 pos, V3 dir, TriFilterParams &f)
{
    const double px = pos.x - sv.sphere_center[0], py = pos.y - sv.sphere_center[1], pz = pos.z - sv.sphere_center[2];
    const double S = sv.tri_extent + fabs(px) + fabs(py) + fabs(pz) + 1.0;
    if (!(S < 1.0e14)) { tri_filter_pass_all(f); return; }
    f.dx = (float)dir.x; f.dy = (float)dir.y; f.dz = (float)dir.z;
    f.npx = (float)(-px); f.npy = (float)(-py); f.npz = (float)(-pz);
    f.A = (float)(S * (64.0 / 16777216.0));
    f.pad = 0.0f;
}

// value whose sign bit is clear <=> the triangle record is a candidate for this ray
__device__ __forceinline__ uint32_t tri_filter_sign(const float4 A, const float4 B, const TriFilterParams &f)
{
    const float dn = __builtin_fmaf(A.x, f.dx, __builtin_fmaf(A.y, f.dy, A.z * f.dz));
    const float nv = __builtin_fmaf(A.x, f.npx, __builtin_fmaf(A.y, f.npy, __builtin_fmaf(A.z, f.npz, A.w)));
    const float adn = __builtin_fabsf(dn), anv = __builtin_fabsf(nv);
    const float ax = -f.npx - B.x, ay = -f.npy - B.y;
    const float ex = __builtin_fmaf(ax, adn, f.dx * anv);
    const float ey = __builtin_fmaf(ay, adn, f.dy * anv);
    const float sx = __builtin_fmaf(B.z, adn, f.A) - __builtin_fabsf(ex);
    const float sy = __builtin_fmaf(B.w, adn, f.A) - __builtin_fabsf(ey);
    return __float_as_uint(sx) | __float_as_uint(sy);
}

// D for one sphere record {c - centre (xyz), |c - centre|^2 - r^2}
__device__ __forceinline__ float filter_disc1(const float4 s, const FilterParams &f)
{
    float b = __builtin_fmaf(s.x, f.dx, __builtin_fmaf(s.y, f.dy, __builtin_fmaf(s.z, f.dz, f.npd)));
    float q = __builtin_fmaf(s.x, f.p2x, __builtin_fmaf(s.y, f.p2y, __builtin_fmaf(s.z, f.p2z, f.nppE)));
    return __builtin_fmaf(b, b, q - s.w);
}

}  // namespace rtx
