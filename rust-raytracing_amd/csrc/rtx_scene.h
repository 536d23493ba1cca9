// rtx_scene.h -- device-resident scene snapshot (immutable during a render) and the exact
// f64 shape tests, written once for host (precompute) and device (evaluation).
//
// Layout in HBM (one contiguous allocation per array, scene replicated per GPU):
//   spheres   SphereX[ns]   32 B  {c.xyz, r*r}        + sphere_id[ns]  u32 scene index
//   planes    PlaneX[np]    80 B  {pos, normal, nhat, id}
//   triangles TriX[nt]     112 B  {v0, n, nn, elimination constants, id}
//   materials MaterialX[n_objects] 56 B, indexed by SCENE index (object.rs:78-86)
//   f32 filter records (MIXED kernel): float4 per sphere {c.xyz, r*r} staged through LDS.
// The per-shape records hold everything that is ray-independent in the reference's
// distance()/normal() (the reference recomputes it per call; hoisting is bit-identical because
// it does not depend on the ray).
#pragma once

#include "rtx_math.h"

namespace rtx {

struct SphereX {                       // object/sphere.rs:9-12
    double cx, cy, cz;
    double rr;                         // radius * radius (sphere.rs:24)
};

struct PlaneX {                        // object/plane.rs:9-12
    V3 position;
    V3 normal;                         // as given
    V3 nhat;                           // normal.norm() (plane.rs:22; == Object::normal_at, object.rs:38)
    uint32_t id;                       // index in Scene.objects
    uint32_t pad_;
};

struct TriX {                          // object/triangle.rs:9-11
    V3 v0;                             // vertices[0]
    V3 n;                              // (v1-v0).cross(v2-v0).norm()            (triangle.rs:104-107)
    V3 nn;                             // n.norm(): Object::normal_at's second norm (object.rs:38)
    // Triangle::contains (triangle.rs:37-101) restricted to what does not depend on the point:
    // the Gauss-Jordan steps on the r,s columns.  The point column then needs
    //   z1 = p[i]/piv1; z2 = (p[j] - z1*f)/piv2; a = z1 - z2*g; b = z2
    double piv1, f, piv2, g;
    uint32_t i, j;                     // which of p.x/p.y/p.z feed rows 1 and 2 after the swaps
    uint32_t degenerate;               // "can't handle LGS" (triangle.rs:64,83): never contains
    uint32_t id;                       // index in Scene.objects
};

struct MaterialX {                     // object.rs:78-86
    V3 base_color;
    V3 emission_color;
    double roughness;
};

// ---- host/device precompute -------------------------------------------------------------

RTX_HD SphereX make_sphere(const double g[4])
{
    SphereX s;
    s.cx = g[0]; s.cy = g[1]; s.cz = g[2];
    s.rr = g[3] * g[3];
    return s;
}

RTX_HD PlaneX make_plane(const double g[6], uint32_t id)
{
    PlaneX p;
    p.position = mk(g[0], g[1], g[2]);
    p.normal = mk(g[3], g[4], g[5]);
    p.nhat = vnorm(p.normal);
    p.id = id; p.pad_ = 0;
    return p;
}

RTX_HD TriX make_triangle(const double g[9], uint32_t id)
{
    TriX t;
    t.id = id;
    t.v0 = mk(g[0], g[1], g[2]);
    V3 r = vsub(mk(g[3], g[4], g[5]), t.v0);          // triangle.rs:20-25 plane_vectors
    V3 s = vsub(mk(g[6], g[7], g[8]), t.v0);
    t.n = vnorm(cross(r, s));                         // triangle.rs:104-107
    t.nn = vnorm(t.n);                                // object.rs:38
    // triangle.rs:55-57: rows (r.c, s.c, p.c) for c = x, y, z; we track (r.c, s.c) and the row's c.
    double ax[3] = { r.x, r.y, r.z };                 // first column
    double ay[3] = { s.x, s.y, s.z };                 // second column
    uint32_t idx[3] = { 0, 1, 2 };
    t.degenerate = 0; t.piv1 = 1.0; t.f = 0.0; t.piv2 = 1.0; t.g = 0.0; t.i = 0; t.j = 1;
#define RTX_SWAP_ROWS(a, b) { double tx = ax[a]; ax[a] = ax[b]; ax[b] = tx; double ty = ay[a]; ay[a] = ay[b]; ay[b] = ty; \
                              uint32_t ti = idx[a]; idx[a] = idx[b]; idx[b] = ti; }
    if (ax[0] == 0.0) {                               // triangle.rs:60-71
        if (ax[1] == 0.0) {
            if (ax[2] == 0.0) { t.degenerate = 1; return t; }
            RTX_SWAP_ROWS(2, 0)
        } else {
            RTX_SWAP_ROWS(0, 1)
        }
    }
    t.piv1 = ax[0];
    double l1y = ay[0] / t.piv1;                      // triangle.rs:72  lgs1 /= lgs1.x
    // lgs1.x is now piv1/piv1 == 1.0, so (lgsK.x / lgs1.x) == lgsK.x exactly (triangle.rs:73-74)
    double f2 = ax[1], f3 = ax[2];
    double l2y = ay[1] - l1y * f2;                    // triangle.rs:73
    double l3y = ay[2] - l1y * f3;                    // triangle.rs:74
    uint32_t jrow = 1;
    double fj = f2;
    if (l2y == 0.0) {                                 // triangle.rs:81-87
        if (l3y == 0.0) { t.degenerate = 1; return t; }
        l2y = l3y; jrow = 2; fj = f3;
    }
    t.piv2 = l2y;                                     // triangle.rs:88  lgs2 /= lgs2.y
    t.f = fj;
    t.g = l1y;                                        // triangle.rs:89: lgs1.y / lgs2.y with lgs2.y == 1.0
    t.i = idx[0];
    t.j = idx[jrow];
#undef RTX_SWAP_ROWS
    return t;
}

// ---- exact shape tests: Some(dst) -> true -----------------------------------------------------
// `dirn` is ray_direction.norm(), which every shape recomputes per call in the reference
// (sphere.rs:21, plane.rs:23, triangle.rs:28); it depends on the ray only, so it is computed
// once per segment here -- same bits.

struct RayX {
    V3 pos, dir;       // ray.position, ray.direction
    V3 dirn;           // ray.direction.norm()
    double a, a2, a4;  // sphere.rs:22: a = dirn.dot(dirn); 2*a; 4*a
};

RTX_HD RayX make_rayx(V3 pos, V3 dir)
{
    RayX r;
    r.pos = pos; r.dir = dir;
    r.dirn = vnorm(dir);
    r.a = dot(r.dirn, r.dirn);
    r.a2 = 2.0 * r.a;
    r.a4 = 4.0 * r.a;
    return r;
}

RTX_HD bool sphere_distance(const SphereX &s, const RayX &ray, double *dst)      // sphere.rs:19-30
{
    V3 offset = mk(ray.pos.x - s.cx, ray.pos.y - s.cy, ray.pos.z - s.cz);
    double b = 2.0 * dot(offset, ray.dirn);
    double c = dot(offset, offset) - s.rr;
    double discriminant = b * b - ray.a4 * c;
    if (discriminant <= 1e-100) return false;
    *dst = (-b - sqrt(discriminant)) / ray.a2;
    return true;
}

RTX_HD V3 sphere_normal_at(const SphereX &s, V3 world_pos)            // sphere.rs:31-33 + object.rs:38
{
    return vnorm(vnorm(mk(world_pos.x - s.cx, world_pos.y - s.cy, world_pos.z - s.cz)));
}

RTX_HD bool plane_distance(const PlaneX &p, const RayX &ray, double *dst)        // plane.rs:20-31
{
    V3 offset = vsub(ray.pos, p.position);
    if (dot(ray.dirn, p.normal) >= 0. || dot(offset, p.normal) <= 0.) return false;
    double t = dot(offset, p.nhat) / dot(ray.dirn, p.nhat);
    V3 intersection_point = vadd(offset, vmuls(ray.dirn, t));
    *dst = vlen(vsub(offset, intersection_point));
    return true;
}

RTX_HD bool triangle_distance(const TriX &t, const RayX &ray, double *dst)       // triangle.rs:108-127
{
    if (dot(t.n, vsub(t.v0, ray.dir)) < 0.0) return false;           // triangle.rs:115 (uses the DIRECTION)
    double dn = dot(ray.dirn, t.n);                                   // triangle.rs:31
    if (dn == 0.0) return false;                                      // INFINITY -> None (triangle.rs:32,119)
    double distance = fabs(dot(t.n, vsub(t.v0, ray.pos)) / dn);       // triangle.rs:34,118
    if (distance == (double)INFINITY) return false;                  // triangle.rs:119
    V3 hit = vadd(ray.pos, vmuls(ray.dir, distance));                 // triangle.rs:122
    if (t.degenerate) return false;                                   // triangle.rs:64,83
    V3 p = vsub(hit, t.v0);                                           // triangle.rs:39
    double pi = t.i == 0 ? p.x : (t.i == 1 ? p.y : p.z);
    double pj = t.j == 0 ? p.x : (t.j == 1 ? p.y : p.z);
    double z1 = pi / t.piv1;                                          // triangle.rs:72
    double z2 = pj - z1 * t.f;                                        // triangle.rs:73/74
    z2 = z2 / t.piv2;                                                 // triangle.rs:88
    double a = z1 - z2 * t.g;                                         // triangle.rs:89
    double b = z2;                                                    // triangle.rs:96
    if (!(0. <= a && a <= 1. && 0. <= b && b <= 1. && (a + b) <= 1.)) return false;   // triangle.rs:100
    *dst = distance;
    return true;
}

}  // namespace rtx
