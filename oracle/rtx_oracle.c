/*
 * rtx_oracle.c -- CPU oracle (test infrastructure; see rtx_oracle.h for the rules and the
 * "parity unpinned" statement).  Plain C, IEEE f64, one statement per reference operation,
 * compiled with -ffp-contract=off so that no a*b+c is fused (Rust never fuses).
 *
 * Citations are `path:line` relative to the reference's src/ directory.
 */
#define _GNU_SOURCE                                   /* sincos() */
#include "rtx_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------- */
/* RNG: counter-based, replaces fastrand's thread-local WyRand (math/vector.rs:31-33,37-38).     */
/* Same output distribution as fastrand::f64(): 52 random mantissa bits, [1,2) - 1.0.            */
/* ------------------------------------------------------------------------------------------- */
static inline uint64_t mix64(uint64_t z)            /* SplitMix64 finalizer */
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
    z ^= z >> 27; z *= 0x94D049BB133111EBULL;
    z ^= z >> 31;
    return z;
}

uint64_t rtxo_rng_key(uint64_t seed, uint64_t pixel_index, uint64_t sample_index)
{
    uint64_t a = mix64(seed + 0x9E3779B97F4A7C15ULL * (pixel_index + 1));
    return mix64(a ^ (0xD1B54A32D192ED03ULL * (sample_index + 1)));
}

double rtxo_rng_u01(uint64_t key, uint64_t draw_index)
{
    uint64_t z = mix64(key + 0x9E3779B97F4A7C15ULL * (draw_index + 1));
    uint64_t bits = 0x3FF0000000000000ULL | (z >> 12);
    double d;
    memcpy(&d, &bits, sizeof d);
    return d - 1.0;
}

/* ------------------------------------------------------------------------------------------- */
/* Vector3 (math/vector.rs, math/vector/{add,sub,mul,div}.rs)                                    */
/* ------------------------------------------------------------------------------------------- */
static inline rtxo_vec3 v3(double x, double y, double z) { rtxo_vec3 r = { x, y, z }; return r; }
static inline rtxo_vec3 vadd(rtxo_vec3 a, rtxo_vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); } /* add.rs:16-24 */
static inline rtxo_vec3 vsub(rtxo_vec3 a, rtxo_vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); } /* sub.rs:16-24 */
static inline rtxo_vec3 vmuls(rtxo_vec3 a, double s)   { return v3(a.x * s, a.y * s, a.z * s); }       /* mul.rs:11-20 */
static inline rtxo_vec3 vmulv(rtxo_vec3 a, rtxo_vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); } /* mul.rs:22-30 element-wise */
static inline rtxo_vec3 vdivs(rtxo_vec3 a, double s)   { return v3(a.x / s, a.y / s, a.z / s); }       /* div.rs:11-20 */
static inline rtxo_vec3 vneg(rtxo_vec3 a)              { return v3(-a.x, -a.y, -a.z); }                /* vector.rs:115-121 */

double rtxo_dot(rtxo_vec3 a, rtxo_vec3 b)              /* vector.rs:85-87: left-to-right */
{
    double xx = a.x * b.x;
    double yy = a.y * b.y;
    double zz = a.z * b.z;
    double s = xx + yy;
    return s + zz;
}

rtxo_vec3 rtxo_cross(rtxo_vec3 a, rtxo_vec3 b)         /* vector.rs:89-95 */
{
    return v3(a.y * b.z - a.z * b.y,
              a.z * b.x - a.x * b.z,
              a.x * b.y - a.y * b.x);
}

double rtxo_len(rtxo_vec3 a)                           /* vector.rs:101-103: (self*self).sum().sqrt() */
{
    rtxo_vec3 sq = vmulv(a, a);
    double s = sq.x + sq.y;                            /* vector.rs:97-99 sum(): x + y + z */
    s = s + sq.z;
    return sqrt(s);
}

rtxo_vec3 rtxo_norm(rtxo_vec3 a)                       /* vector.rs:105-107: three true divisions */
{
    return vdivs(a, rtxo_len(a));
}

/* ------------------------------------------------------------------------------------------- */
/* Mat3x3 (math/mat.rs, math/mat/{specific_math,mul,div}.rs) -- only what Camera reaches          */
/* ------------------------------------------------------------------------------------------- */
rtxo_vec3 rtxo_mat_mul_vec(const rtxo_mat3 *m, rtxo_vec3 v)   /* mat/mul.rs:42-50: rhs.dot(row) */
{
    return v3(rtxo_dot(v, m->x), rtxo_dot(v, m->y), rtxo_dot(v, m->z));
}

rtxo_mat3 rtxo_mat_transpose(const rtxo_mat3 *m)       /* specific_math.rs:16-21 */
{
    rtxo_mat3 r;
    r.x = v3(m->x.x, m->y.x, m->z.x);
    r.y = v3(m->x.y, m->y.y, m->z.y);
    r.z = v3(m->x.z, m->y.z, m->z.z);
    return r;
}

static double mat_determinant(const rtxo_mat3 *m)      /* specific_math.rs:23-43 */
{
    double sum1 = m->x.x * m->y.y * m->z.z;
    double sum2 = m->x.y * m->y.z * m->z.x;
    double sum3 = m->x.z * m->y.x * m->z.y;
    double sub1 = m->z.x * m->y.y * m->x.z;
    double sub2 = m->z.y * m->y.z * m->x.x;
    double sub3 = m->z.z * m->y.x * m->x.y;
    return (sum1 + sum2 + sum3) - (sub1 + sub2 + sub3);
}

static rtxo_mat3 mat_adjugate(const rtxo_mat3 *m)      /* specific_math.rs:45-71 */
{
    double a = m->x.x, b = m->x.y, c = m->x.z;
    double d = m->y.x, e = m->y.y, f = m->y.z;
    double g = m->z.x, h = m->z.y, i = m->z.z;
    rtxo_mat3 r;
    r.x = v3(e * i - f * h, c * h - b * i, b * f - c * e);
    r.y = v3(f * g - d * i, a * i - c * g, c * d - a * f);
    r.z = v3(d * h - e * g, b * g - a * h, a * e - b * d);
    return r;
}

rtxo_mat3 rtxo_mat_inverse(const rtxo_mat3 *m)         /* specific_math.rs:10-14; mat/div.rs:10-20 */
{
    rtxo_mat3 adj = mat_adjugate(m);
    double det = mat_determinant(m);
    rtxo_mat3 r;
    r.x = vdivs(adj.x, det);
    r.y = vdivs(adj.y, det);
    r.z = vdivs(adj.z, det);
    return r;
}

/* ------------------------------------------------------------------------------------------- */
/* Camera (raytracing/camera.rs)                                                                 */
/* ------------------------------------------------------------------------------------------- */
static rtxo_mat3 derive_to_world_space_mat(rtxo_vec3 direction)   /* camera.rs:42-49 */
{
    rtxo_vec3 cam_forward = rtxo_norm(direction);
    rtxo_vec3 cam_right = rtxo_cross(cam_forward, v3(0., 0., -1.));
    rtxo_vec3 cam_up = rtxo_cross(cam_forward, cam_right);
    rtxo_mat3 rows = { cam_right, cam_up, cam_forward };
    return rtxo_mat_transpose(&rows);
}

void rtxo_camera_new(rtxo_camera *cam, rtxo_vec3 position, rtxo_vec3 direction, double fov) /* camera.rs:19-28 */
{
    rtxo_mat3 to_world = derive_to_world_space_mat(direction);
    cam->fov = fov;
    cam->position = position;
    cam->direction = direction;
    cam->to_cam_space = rtxo_mat_inverse(&to_world);
    cam->to_world_space = to_world;
}

void rtxo_camera_set_direction(rtxo_camera *cam, rtxo_vec3 direction)     /* camera.rs:35-40 */
{
    /* The reference derives the matrices from the OLD self.direction and only then stores the
     * new one, so the matrices lag one call behind.  Restated as-is. */
    rtxo_mat3 to_world = derive_to_world_space_mat(cam->direction);
    cam->to_cam_space = rtxo_mat_inverse(&to_world);
    cam->to_world_space = to_world;
    cam->direction = direction;
}

rtxo_vec3 rtxo_camera_to_cam_space(const rtxo_camera *cam, rtxo_vec3 v)   /* camera.rs:51-53 */
{
    return rtxo_mat_mul_vec(&cam->to_cam_space, vsub(v, cam->position));
}

rtxo_vec3 rtxo_camera_to_world_space(const rtxo_camera *cam, rtxo_vec3 v) /* camera.rs:55-57 */
{
    return vadd(rtxo_mat_mul_vec(&cam->to_world_space, v), cam->position);
}

rtxo_vec3 rtxo_camera_rotate_to_world_space(const rtxo_camera *cam, rtxo_vec3 v) /* camera.rs:65-67 */
{
    return rtxo_mat_mul_vec(&cam->to_world_space, v);
}

/* ------------------------------------------------------------------------------------------- */
/* Sphere (raytracing/object/sphere.rs)                                                          */
/* ------------------------------------------------------------------------------------------- */
int rtxo_sphere_distance(const double g[4], rtxo_vec3 ray_position, rtxo_vec3 ray_direction, double *dst)
{                                                                   /* sphere.rs:19-30 */
    rtxo_vec3 position = v3(g[0], g[1], g[2]);
    double radius = g[3];
    rtxo_vec3 offset = vsub(ray_position, position);
    rtxo_vec3 dir = rtxo_norm(ray_direction);
    double a = rtxo_dot(dir, dir);
    double b = 2.0 * rtxo_dot(offset, dir);
    double c = rtxo_dot(offset, offset) - radius * radius;
    double discriminant = b * b - 4.0 * a * c;
    if (discriminant <= 1e-100)
        return 0;
    *dst = (-b - sqrt(discriminant)) / (2.0 * a);
    return 1;
}

rtxo_vec3 rtxo_sphere_normal(const double g[4], rtxo_vec3 world_position)   /* sphere.rs:31-33 */
{
    return rtxo_norm(vsub(world_position, v3(g[0], g[1], g[2])));
}

/* ------------------------------------------------------------------------------------------- */
/* Plane (raytracing/object/plane.rs)                                                            */
/* ------------------------------------------------------------------------------------------- */
int rtxo_plane_distance(const double g[6], rtxo_vec3 ray_pos, rtxo_vec3 ray_dir, double *dst)
{                                                                   /* plane.rs:20-31 */
    rtxo_vec3 position = v3(g[0], g[1], g[2]);
    rtxo_vec3 normal = v3(g[3], g[4], g[5]);
    rtxo_vec3 offset = vsub(ray_pos, position);
    rtxo_vec3 norm = rtxo_norm(normal);
    rtxo_vec3 dir = rtxo_norm(ray_dir);
    if (rtxo_dot(dir, normal) >= 0. || rtxo_dot(offset, normal) <= 0.)
        return 0;
    double t = rtxo_dot(offset, norm) / rtxo_dot(dir, norm);
    rtxo_vec3 intersection_point = vadd(offset, vmuls(dir, t));
    *dst = rtxo_len(vsub(offset, intersection_point));
    return 1;
}

rtxo_vec3 rtxo_plane_normal(const double g[6], rtxo_vec3 relative_position)  /* plane.rs:33-35 */
{
    (void)relative_position;
    return v3(g[3], g[4], g[5]);
}

/* ------------------------------------------------------------------------------------------- */
/* Triangle (raytracing/object/triangle.rs)                                                      */
/* ------------------------------------------------------------------------------------------- */
static void tri_plane_vectors(const double g[9], rtxo_vec3 *basis, rtxo_vec3 *dir1, rtxo_vec3 *dir2)
{                                                                   /* triangle.rs:20-25 */
    *basis = v3(g[0], g[1], g[2]);
    *dir1 = vsub(v3(g[3], g[4], g[5]), *basis);
    *dir2 = vsub(v3(g[6], g[7], g[8]), *basis);
}

rtxo_vec3 rtxo_triangle_normal(const double g[9], rtxo_vec3 world_position)  /* triangle.rs:104-107 */
{
    (void)world_position;
    rtxo_vec3 basis, a, b;
    tri_plane_vectors(g, &basis, &a, &b);
    return rtxo_norm(rtxo_cross(a, b));
}

static double tri_plane_distance(const double g[9], rtxo_vec3 ray_pos, rtxo_vec3 dir)
{                                                                   /* triangle.rs:26-35 */
    dir = rtxo_norm(dir);
    rtxo_vec3 normal = rtxo_triangle_normal(g, v3(0, 0, 0));
    rtxo_vec3 self_pos = v3(g[0], g[1], g[2]);
    if (rtxo_dot(dir, normal) == 0.0)
        return INFINITY;
    return rtxo_dot(normal, vsub(self_pos, ray_pos)) / rtxo_dot(dir, normal);
}

int rtxo_triangle_contains(const double g[9], rtxo_vec3 point)      /* triangle.rs:37-101 */
{
    rtxo_vec3 pos, r, s;
    tri_plane_vectors(g, &pos, &r, &s);
    rtxo_vec3 p = vsub(point, pos);
    /* a * r + b * s = p  (triangle.rs:54-57) */
    rtxo_vec3 lgs1 = v3(r.x, s.x, p.x);
    rtxo_vec3 lgs2 = v3(r.y, s.y, p.y);
    rtxo_vec3 lgs3 = v3(r.z, s.z, p.z);
    rtxo_vec3 tmp;

    if (lgs1.x == 0.0) {                                            /* triangle.rs:60-71 */
        if (lgs2.x == 0.0) {
            if (lgs3.x == 0.0)
                return -1;                                          /* eprintln!("can't handle LGS"); false */
            tmp = lgs3; lgs3 = lgs1; lgs1 = tmp;                    /* (lgs3, lgs1) = (lgs1, lgs3) */
        } else {
            tmp = lgs1; lgs1 = lgs2; lgs2 = tmp;                    /* (lgs1, lgs2) = (lgs2, lgs1) */
        }
    }
    lgs1 = vdivs(lgs1, lgs1.x);                                     /* triangle.rs:72 */
    lgs2 = vsub(lgs2, vmuls(lgs1, lgs2.x / lgs1.x));                /* triangle.rs:73 */
    lgs3 = vsub(lgs3, vmuls(lgs1, lgs3.x / lgs1.x));                /* triangle.rs:74 */
    /* triangle.rs:76-78 assert_eq!s: hold for finite non-degenerate input */
    if (lgs2.y == 0.0) {                                            /* triangle.rs:81-87 */
        if (lgs3.y == 0.0)
            return -1;                                              /* "can't handle LGS" */
        tmp = lgs2; lgs2 = lgs3; lgs3 = tmp;
    }
    lgs2 = vdivs(lgs2, lgs2.y);                                     /* triangle.rs:88 */
    lgs1 = vsub(lgs1, vmuls(lgs2, lgs1.y / lgs2.y));                /* triangle.rs:89 */
    lgs3 = vsub(lgs3, vmuls(lgs2, lgs3.y / lgs2.y));                /* triangle.rs:90 */
    (void)lgs3;
    double a = lgs1.z, b = lgs2.z;                                  /* triangle.rs:96 */
    return (0. <= a && a <= 1. && 0. <= b && b <= 1. && (a + b) <= 1.) ? 1 : 0; /* triangle.rs:100 */
}

int rtxo_triangle_distance(const double g[9], rtxo_vec3 pos, rtxo_vec3 dir, double *dst)
{                                                                   /* triangle.rs:108-127 */
    rtxo_vec3 v0 = v3(g[0], g[1], g[2]);
    /* "behind triangle" test uses the ray DIRECTION, as the reference does (triangle.rs:115) */
    if (rtxo_dot(rtxo_triangle_normal(g, v3(0, 0, 0)), vsub(v0, dir)) < 0.0)
        return 0;
    double distance = fabs(tri_plane_distance(g, pos, dir));        /* triangle.rs:118 */
    if (distance == INFINITY)
        return 0;
    rtxo_vec3 hit_point = vadd(pos, vmuls(dir, distance));          /* triangle.rs:122 */
    if (rtxo_triangle_contains(g, hit_point) != 1)
        return 0;
    *dst = distance;
    return 1;
}

/* ------------------------------------------------------------------------------------------- */
/* Object (raytracing/object.rs): in faithful mode every call takes the object's mutex            */
/* ------------------------------------------------------------------------------------------- */
typedef int (*distance_fn)(const double *, rtxo_vec3, rtxo_vec3, double *);
typedef rtxo_vec3 (*normal_fn)(const double *, rtxo_vec3);
static const distance_fn k_distance[3] = {
    (distance_fn)rtxo_sphere_distance, (distance_fn)rtxo_plane_distance, (distance_fn)rtxo_triangle_distance };
static const normal_fn k_normal[3] = {
    (normal_fn)rtxo_sphere_normal, (normal_fn)rtxo_plane_normal, (normal_fn)rtxo_triangle_normal };

int rtxo_object_distance(const rtxo_object *o, rtxo_vec3 pos, rtxo_vec3 dir, double *dst)  /* object.rs:49-51 */
{
    if (o->kind > 2) return 0;
    return k_distance[o->kind](o->geom, pos, dir, dst);
}

rtxo_vec3 rtxo_object_normal_at(const rtxo_object *o, rtxo_vec3 world_pos)                 /* object.rs:37-39 */
{
    if (o->kind > 2) return v3(NAN, NAN, NAN);
    return rtxo_norm(k_normal[o->kind](o->geom, world_pos));      /* second normalisation, as the reference */
}

/* render context: scene + optional per-object locks (faithful mode) */
typedef struct {
    const rtxo_scene *scene;
    pthread_mutex_t  *locks;      /* NULL in clean mode */
    rtxo_path_step   *steps;      /* NULL, or where ctx_render_ray writes the current path's transcript (rtxo_trace_row) */
    uint32_t          max_steps, *n_steps;
} render_ctx;

static inline int ctx_distance(const render_ctx *c, uint64_t i, rtxo_vec3 pos, rtxo_vec3 dir, double *dst)
{
    const rtxo_object *o = &c->scene->objects[i];
    if (!c->locks)
        return rtxo_object_distance(o, pos, dir, dst);
    pthread_mutex_lock(&c->locks[i]);                             /* object.rs:50 lock().unwrap() */
    int r = rtxo_object_distance(o, pos, dir, dst);
    pthread_mutex_unlock(&c->locks[i]);
    return r;
}

static inline rtxo_vec3 ctx_normal_at(const render_ctx *c, uint64_t i, rtxo_vec3 world_pos)
{
    const rtxo_object *o = &c->scene->objects[i];
    if (!c->locks)
        return rtxo_object_normal_at(o, world_pos);
    pthread_mutex_lock(&c->locks[i]);                             /* object.rs:38 */
    rtxo_vec3 n = k_normal[o->kind](o->geom, world_pos);
    pthread_mutex_unlock(&c->locks[i]);
    return rtxo_norm(n);
}

/* ------------------------------------------------------------------------------------------- */
/* Shading + bounce loop (raytracing/scene.rs)                                                   */
/* ------------------------------------------------------------------------------------------- */
/* sin / cos of random_direction's angle.  Mode 0 (the default, and what the reference does): the platform libm, f64::cos / f64::sin
 * (vector.rs:40-41).  Mode 1 -- a test mode, not the reference: the routine the DEVICE uses for this angle (the product's
 * rtx_math.h sincos_2pi, restated here operation for operation: an exact reduction by pi/2, fdlibm's polynomials on [-pi/4, pi/4]
 * with the squares' rounding errors carried along, one final rounding).
 * No two libms agree on the last bit of sin / cos, so that is the one place the device may differ from this oracle (<= 1 ulp, images
 * within 1e-9); with mode 1 the difference is gone and the kernels must equal the oracle BIT FOR BIT, which turns every seeded
 * comparison of the test suite into an exact one (tests/test_gpu_parity.py). */
static int g_sincos_mode = 0;
void rtxo_set_sincos_mode(int mode) { g_sincos_mode = mode; }

void rtxo_device_sincos(double x, double *sn, double *cs)
{
    static const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00,
                        pio2_2 = 6.07710050630396597660e-11, pio2_2t = 2.02226624879595063154e-21;
    static const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                        S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    static const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                        C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double fn = rint(x * invpio2);                              /* n = 0..4 for x in [0, 2 pi] */
    const double r1 = x - fn * pio2_1, w2 = fn * pio2_2;              /* x - n pi/2: exact, exact, */
    const double r = r1 - w2;                                         /* rounded -- the error joins the third piece */
    const double w = fn * pio2_2t - ((r1 - r) - w2);
    const double y0 = r - w, y1 = (r - y0) - w;                       /* the reduced angle and its tail */
    const double z = y0 * y0, zl = fma(y0, y0, -z);
    const double v = z * y0, vl = fma(z, y0, -v) + zl * y0;
    const double rs = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    double sm = z * (v * rs - 0.5 * y1) + y1;
    sm = fma(vl, S1, sm);
    sm = fma(v, S1, sm);
    const double s = y0 + sm;
    const double rc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double h = 0.5 * z, w1 = 1.0 - h;
    double cm = fma(z, rc, -(y0 * y1));
    cm = fma(-0.5, zl, cm);
    cm = ((1.0 - w1) - h) + cm;
    const double c = w1 + cm;
    const int n = (int)fn & 3;
    *sn = n == 0 ? s : (n == 1 ? c : (n == 2 ? -s : -c));
    *cs = n == 0 ? c : (n == 1 ? -s : (n == 2 ? -c : s));
}

void rtxo_device_sincos_n(const double *x, size_t n, double *sn, double *cs)
{
    for (size_t i = 0; i < n; ++i) rtxo_device_sincos(x[i], sn + i, cs + i);
}

rtxo_vec3 rtxo_random_direction(double u_z, double u_theta)         /* vector.rs:36-45 */
{
    double z = u_z * 2.0 - 1.0;
    double theta = u_theta * 2.0 * 3.14159265358979323846;          /* std::f64::consts::PI */
    double r = sqrt(1.0 - z * z);
    if (g_sincos_mode == 1) {
        double sn, cs;
        rtxo_device_sincos(theta, &sn, &cs);
        return rtxo_norm(v3(r * cs, r * sn, z));
    }
    double sn, cs;
    sincos(theta, &sn, &cs);                                        /* (see rtxo_get_ray_dir: one glibc sincos call, by name) */
    return rtxo_norm(v3(r * cs, r * sn, z));
}

rtxo_vec3 rtxo_random_bounce_dir(rtxo_vec3 ray_dir, rtxo_vec3 surface_normal, double surface_roughness,
                                 double u_z, double u_theta)        /* scene.rs:279-292 */
{
    rtxo_vec3 random_dir = rtxo_random_direction(u_z, u_theta);
    /* ray_dir - surface_normal * 2 * ray_dir.dot(surface_normal): ((n*2)*dot) */
    rtxo_vec3 reflection_dir = vsub(ray_dir, vmuls(vmuls(surface_normal, 2.0), rtxo_dot(ray_dir, surface_normal)));
    rtxo_vec3 random_to_reflection_dir = vsub(reflection_dir, random_dir);
    double reflection_mult = 1.0 - surface_roughness;
    rtxo_vec3 final_direction = vadd(random_dir, vmuls(random_to_reflection_dir, reflection_mult));
    final_direction = rtxo_norm(final_direction);
    if (rtxo_dot(final_direction, surface_normal) > 0.0)
        return final_direction;
    return vneg(final_direction);
}

static inline int is_normal_positive(double d)                      /* scene.rs:249 */
{
    return isnormal(d) && !signbit(d);
}

static int64_t ctx_closest_object(const render_ctx *c, rtxo_vec3 pos, rtxo_vec3 dir, double *dst_out)
{                                                                   /* scene.rs:243-251 */
    int64_t best = -1;
    double best_dst = 0.0;
    for (uint64_t i = 0; i < c->scene->n_objects; ++i) {
        double dst;
        if (!ctx_distance(c, i, pos, dir, &dst))
            continue;
        if (!is_normal_positive(dst))
            continue;
        /* min_by(total_cmp): the FIRST minimal element wins (strict <) */
        if (best < 0 || dst < best_dst) {
            best = (int64_t)i;
            best_dst = dst;
        }
    }
    if (best >= 0) *dst_out = best_dst;
    return best;
}

int64_t rtxo_closest_object(const rtxo_scene *s, rtxo_vec3 pos, rtxo_vec3 dir, double *dst)
{
    render_ctx c = { s, NULL, NULL, 0, NULL };
    return ctx_closest_object(&c, pos, dir, dst);
}

rtxo_vec3 rtxo_get_ray_dir(const rtxo_scene *s, double x, double y, double vertical_fov) /* scene.rs:213-222 */
{
    double angle_x = s->camera.fov * (x - 0.5);
    double angle_y = vertical_fov * (y - 0.5);
    /* f64::sin and f64::cos of one angle (scene.rs:216-219): on a GNU target the compiler makes ONE glibc sincos() call of such a
     * pair (LLVM for the reference's rustc build, gcc for this file -- it did, before this was spelled out), and glibc's sincos
     * differs from its sin / cos in the last place on ~0.07 % of arguments (2.35).  Called by name here and in random_direction so
     * that the oracle does not depend on the optimiser's mood; the product's host tables call the same function. */
    double sx, cx, sy, cy;
    sincos(angle_x, &sx, &cx);
    sincos(angle_y, &sy, &cy);
    rtxo_vec3 cam_space_dir = v3(sx, sy, cx * cy);
    return rtxo_camera_rotate_to_world_space(&s->camera, cam_space_dir);
}

typedef struct {                                                    /* raytracing/ray.rs:4-21 */
    rtxo_vec3 position, direction, resulting_color, light_color;
} ray_t;

static rtxo_vec3 ctx_render_ray(const render_ctx *c, ray_t ray, uint64_t key, uint64_t *draw, uint64_t *segments)
{                                                                   /* scene.rs:223-242 */
    const rtxo_scene *s = c->scene;
    if (s->n_objects == 0)
        return ray.resulting_color;
    for (uint64_t bounce = 0; bounce < s->config.max_bounces + 1; ++bounce) {
        if (ray.light_color.x == 0.0 && ray.light_color.y == 0.0 && ray.light_color.z == 0.0)
            break;                                                  /* scene.rs:228 */
        double dst;
        if (segments) ++*segments;
        int64_t hit = ctx_closest_object(c, ray.position, ray.direction, &dst);
        if (c->steps) {
            if (*c->n_steps < c->max_steps) {
                rtxo_path_step *st = &c->steps[*c->n_steps];
                st->position[0] = ray.position.x; st->position[1] = ray.position.y; st->position[2] = ray.position.z;
                st->direction[0] = ray.direction.x; st->direction[1] = ray.direction.y; st->direction[2] = ray.direction.z;
                st->distance = hit < 0 ? INFINITY : dst;
                st->object = hit;
            }
            ++*c->n_steps;
        }
        if (hit < 0)
            break;
        ray.position = vadd(ray.position, vmuls(ray.direction, dst));  /* scene.rs:234 */
        /* ray_hit, scene.rs:260-278 */
        const rtxo_object *obj = &s->objects[hit];
        double u_z = rtxo_rng_u01(key, (*draw)++);                  /* vector.rs:37 */
        double u_theta = rtxo_rng_u01(key, (*draw)++);              /* vector.rs:38 */
        ray.direction = rtxo_random_bounce_dir(ray.direction, ctx_normal_at(c, (uint64_t)hit, ray.position),
                                               obj->roughness, u_z, u_theta);
        rtxo_vec3 em = v3(obj->emission_color[0], obj->emission_color[1], obj->emission_color[2]);
        rtxo_vec3 bc = v3(obj->base_color[0], obj->base_color[1], obj->base_color[2]);
        ray.resulting_color = vadd(ray.resulting_color, vmulv(ray.light_color, em)); /* scene.rs:276 */
        ray.light_color = vmulv(ray.light_color, bc);                               /* scene.rs:277 */
    }
    return ray.resulting_color;
}

static rtxo_vec3 ctx_render_pixel(const render_ctx *c, double u, double v, double vertical_fov,
                                  uint64_t pixel_index, uint64_t *segments)
{                                                                   /* scene.rs:194-212 */
    const rtxo_scene *s = c->scene;
    rtxo_vec3 ray_dir = rtxo_get_ray_dir(s, u, v, vertical_fov);
    rtxo_vec3 sum = v3(0.0, 0.0, 0.0);                              /* iter_ops.rs:4-8: fold from zeros */
    uint64_t n = s->config.rays_per_pixel;
    for (uint64_t sample = 0; sample < n; ++sample) {
        uint64_t key = rtxo_rng_key(s->config.seed, pixel_index, sample);
        uint64_t draw = 0;
        ray_t ray;
        ray.position = s->camera.position;                          /* ray.rs:14-21 */
        ray.direction = ray_dir;
        ray.resulting_color = v3(0.0, 0.0, 0.0);
        ray.light_color = v3(1.0, 1.0, 1.0);
        /* Vector3::random(): fields drawn in order x, y, z (vector.rs:29-35) */
        rtxo_vec3 rnd1;
        rnd1.x = rtxo_rng_u01(key, draw++); rnd1.y = rtxo_rng_u01(key, draw++); rnd1.z = rtxo_rng_u01(key, draw++);
        rtxo_vec3 ray_position = vadd(ray.position, vmuls(rnd1, s->config.non_focal_offset));   /* scene.rs:202 */
        rtxo_vec3 focal_point = vadd(ray.position, vmuls(ray.direction, s->config.focal_length)); /* scene.rs:203 */
        rtxo_vec3 rnd2;
        rnd2.x = rtxo_rng_u01(key, draw++); rnd2.y = rtxo_rng_u01(key, draw++); rnd2.z = rtxo_rng_u01(key, draw++);
        rtxo_vec3 target_point = vadd(focal_point, vmuls(rnd2, s->config.focal_offset));          /* scene.rs:204 */
        rtxo_vec3 ray_direction = vsub(target_point, ray_position);                                /* scene.rs:205 */
        ray.position = ray_position;
        ray.direction = rtxo_norm(ray_direction);                                                 /* scene.rs:207 */
        render_ctx cs = *c;
        uint32_t n_steps = 0;
        if (c->steps) {                                             /* rtxo_trace_row: this sample's slice of the row's transcript */
            cs.steps = c->steps + sample * c->max_steps;
            cs.n_steps = &n_steps;
        }
        rtxo_vec3 col = ctx_render_ray(&cs, ray, key, &draw, segments);
        if (c->steps) c->n_steps[sample] = n_steps;
        sum = vadd(sum, col);
    }
    return vdivs(sum, (double)n);                                   /* scene.rs:253-259: sum / len */
}

rtxo_vec3 rtxo_render_pixel(const rtxo_scene *s, double u, double v, double vertical_fov,
                            uint64_t pixel_index, uint64_t *segments)
{
    render_ctx c = { s, NULL, NULL, 0, NULL };
    return ctx_render_pixel(&c, u, v, vertical_fov, pixel_index, segments);
}

/* The transcript of every path of one image row (test aid; the lab library's rtx_debug_paths is the device's): per segment the ray
 * closest_object was asked about, the winning distance and object index (-1, +inf: none).  steps [width][rays_per_pixel][max_steps],
 * counts [width][rays_per_pixel]. */
int rtxo_trace_row(const rtxo_scene *s, uint32_t width, uint32_t height, uint32_t row, uint32_t max_steps,
                   rtxo_path_step *steps, uint32_t *counts)
{
    if (!s || !steps || !counts || row >= height || width == 0 || max_steps == 0) return -1;
    const double vertical_fov = (double)height / (double)width * s->camera.fov;     /* scene.rs:145 */
    const double y = (double)row / (double)height;
    const uint64_t spp = s->config.rays_per_pixel;
    for (uint32_t xi = 0; xi < width; ++xi) {
        render_ctx c = { s, NULL, steps + (uint64_t)xi * spp * max_steps, max_steps, counts + (uint64_t)xi * spp };
        (void)ctx_render_pixel(&c, (double)xi / (double)width, y, vertical_fov, (uint64_t)row * width + xi, NULL);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------- */
/* Scene::render (scene.rs:144-170)                                                              */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    render_ctx ctx;
    uint32_t   width, height;
    double     vertical_fov;
    double    *out;
    uint64_t  *segments;
    /* clean mode: shared row cursor over the selected rows */
    uint32_t   row_begin, row_stride;
    pthread_mutex_t *cursor_lock;
    uint32_t  *cursor;
    /* faithful mode: the one row this thread owns */
    uint32_t   row;
} row_job;

static void render_row(const row_job *j, uint32_t yi)
{
    double y = (double)yi / (double)j->height;                      /* scene.rs:153 */
    for (uint32_t xi = 0; xi < j->width; ++xi) {
        double x = (double)xi / (double)j->width;                   /* scene.rs:157 */
        uint64_t pix = (uint64_t)yi * j->width + xi;
        uint64_t seg = 0;
        rtxo_vec3 c = ctx_render_pixel(&j->ctx, x, y, j->vertical_fov, pix, j->segments ? &seg : NULL);
        double *o = j->out + 3 * pix;
        o[0] = c.x; o[1] = c.y; o[2] = c.z;
        if (j->segments) j->segments[pix] = seg;
    }
}

static void *faithful_thread(void *arg)
{
    const row_job *j = (const row_job *)arg;
    render_row(j, j->row);
    return NULL;
}

static void *clean_thread(void *arg)
{
    const row_job *j = (const row_job *)arg;
    for (;;) {
        pthread_mutex_lock(j->cursor_lock);
        uint32_t yi = *j->cursor;
        *j->cursor = yi + j->row_stride;
        pthread_mutex_unlock(j->cursor_lock);
        if (yi >= j->height)
            return NULL;
        render_row(j, yi);
    }
}

int rtxo_render(const rtxo_scene *s, uint32_t width, uint32_t height,
                uint32_t row_begin, uint32_t row_stride,
                double *out_rgb, uint64_t *segments, int n_threads, int mode)
{
    if (!s || !out_rgb || row_stride == 0) return 1;
    if (width == 0 || height == 0) return 0;
    double vertical_fov = (double)height / (double)width * s->camera.fov;   /* scene.rs:145 */

    row_job base;
    memset(&base, 0, sizeof base);
    base.ctx.scene = s;
    base.ctx.locks = NULL;
    base.width = width; base.height = height;
    base.vertical_fov = vertical_fov;
    base.out = out_rgb; base.segments = segments;
    base.row_begin = row_begin; base.row_stride = row_stride;

    if (mode == RTXO_MODE_FAITHFUL) {
        /* one mutex per object (object.rs:12 Arc<Mutex<dyn CustomShape>>), one thread per row (scene.rs:151) */
        pthread_mutex_t *locks = (pthread_mutex_t *)malloc(sizeof(pthread_mutex_t) * (s->n_objects ? s->n_objects : 1));
        if (!locks) return 2;
        for (uint64_t i = 0; i < s->n_objects; ++i) pthread_mutex_init(&locks[i], NULL);
        base.ctx.locks = locks;
        uint32_t n_rows = row_begin < height ? (height - row_begin + row_stride - 1) / row_stride : 0;
        pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (n_rows ? n_rows : 1));
        row_job *jobs = (row_job *)malloc(sizeof(row_job) * (n_rows ? n_rows : 1));
        int rc = 0;
        uint32_t started = 0;
        pthread_attr_t attr;
        pthread_attr_init(&attr);
        pthread_attr_setstacksize(&attr, 256 * 1024);
        for (uint32_t k = 0; k < n_rows; ++k) {
            jobs[k] = base;
            jobs[k].row = row_begin + k * row_stride;
            if (pthread_create(&th[k], &attr, faithful_thread, &jobs[k]) != 0) { rc = 3; break; }
            ++started;
        }
        for (uint32_t k = 0; k < started; ++k) pthread_join(th[k], NULL);   /* scene.rs:168 join in row order */
        pthread_attr_destroy(&attr);
        for (uint64_t i = 0; i < s->n_objects; ++i) pthread_mutex_destroy(&locks[i]);
        free(jobs); free(th); free(locks);
        return rc;
    }

    if (n_threads < 1) n_threads = 1;
    pthread_mutex_t cursor_lock = PTHREAD_MUTEX_INITIALIZER;
    uint32_t cursor = row_begin;
    base.cursor_lock = &cursor_lock;
    base.cursor = &cursor;
    if (n_threads == 1) {
        clean_thread(&base);
        return 0;
    }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    if (!th) return 2;
    int started = 0, rc = 0;
    for (int k = 0; k < n_threads; ++k) {
        if (pthread_create(&th[k], NULL, clean_thread, &base) != 0) { rc = 3; break; }
        ++started;
    }
    for (int k = 0; k < started; ++k) pthread_join(th[k], NULL);
    free(th);
    return rc;
}

/* ------------------------------------------------------------------------------------------- */
/* Selected pixels of Scene::render's width x height frame (scene.rs:144-170 restricted to a list) */
/* ------------------------------------------------------------------------------------------- */
typedef struct {
    render_ctx       ctx;
    uint32_t         width, height;
    double           vertical_fov;
    uint64_t         n;
    const uint32_t  *xs, *ys;
    double          *out;
    uint64_t        *segments;
    pthread_mutex_t *cursor_lock;
    uint64_t        *cursor;
} pixel_job;

static void *pixels_thread(void *arg)
{
    const pixel_job *j = (const pixel_job *)arg;
    for (;;) {
        pthread_mutex_lock(j->cursor_lock);
        uint64_t k = *j->cursor;
        *j->cursor = k + 1;
        pthread_mutex_unlock(j->cursor_lock);
        if (k >= j->n)
            return NULL;
        uint32_t xi = j->xs[k], yi = j->ys[k];
        double x = (double)xi / (double)j->width;                   /* scene.rs:157 */
        double y = (double)yi / (double)j->height;                  /* scene.rs:153 */
        uint64_t seg = 0;
        rtxo_vec3 c = ctx_render_pixel(&j->ctx, x, y, j->vertical_fov, (uint64_t)yi * j->width + xi, j->segments ? &seg : NULL);
        double *o = j->out + 3 * k;
        o[0] = c.x; o[1] = c.y; o[2] = c.z;
        if (j->segments) j->segments[k] = seg;
    }
}

int rtxo_render_pixels(const rtxo_scene *s, uint32_t width, uint32_t height, uint64_t n,
                       const uint32_t *xs, const uint32_t *ys, double *out_rgb, uint64_t *segments, int n_threads)
{
    if (!s || (n && (!xs || !ys || !out_rgb))) return 1;
    if (n == 0) return 0;
    if (width == 0 || height == 0) return 1;
    for (uint64_t k = 0; k < n; ++k)
        if (xs[k] >= width || ys[k] >= height) return 1;
    pixel_job job;
    memset(&job, 0, sizeof job);
    job.ctx.scene = s;
    job.ctx.locks = NULL;
    job.width = width; job.height = height;
    job.vertical_fov = (double)height / (double)width * s->camera.fov;      /* scene.rs:145 */
    job.n = n; job.xs = xs; job.ys = ys; job.out = out_rgb; job.segments = segments;
    pthread_mutex_t cursor_lock = PTHREAD_MUTEX_INITIALIZER;
    uint64_t cursor = 0;
    job.cursor_lock = &cursor_lock;
    job.cursor = &cursor;
    if (n_threads < 1) n_threads = 1;
    if ((uint64_t)n_threads > n) n_threads = (int)n;
    if (n_threads == 1) {
        pixels_thread(&job);
        return 0;
    }
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    if (!th) return 2;
    int started = 0, rc = 0;
    for (int k = 0; k < n_threads; ++k) {
        if (pthread_create(&th[k], NULL, pixels_thread, &job) != 0) { rc = 3; break; }
        ++started;
    }
    if (started == 0) pixels_thread(&job);
    for (int k = 0; k < started; ++k) pthread_join(th[k], NULL);
    free(th);
    return started ? 0 : rc;
}

/* ------------------------------------------------------------------------------------------- */
/* render_to_image (scene.rs:172-178)                                                            */
/* ------------------------------------------------------------------------------------------- */
static inline uint8_t rust_as_u8(double v)      /* Rust `f64 as u8`: saturating, NaN -> 0, truncation toward 0 */
{
    if (!(v == v)) return 0;
    if (v <= 0.0) return 0;
    if (v >= 255.0) return 255;
    return (uint8_t)v;
}

void rtxo_quantize_image(const double *rgb, uint32_t width, uint32_t height, uint8_t *out)
{
    for (uint32_t y = 0; y < height; ++y)
        for (uint32_t x = 0; x < width; ++x) {
            const double *c = rgb + 3 * ((uint64_t)(height - y - 1) * width + x);   /* img[height - y - 1][x] */
            uint8_t *o = out + 3 * ((uint64_t)y * width + x);
            o[0] = rust_as_u8(c[0] * 256.0);                                        /* `* 256` then `as u8` */
            o[1] = rust_as_u8(c[1] * 256.0);
            o[2] = rust_as_u8(c[2] * 256.0);
        }
}
