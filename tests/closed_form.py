"""Scenes whose rendered values follow from the TEXT of the reference alone -- no oracle, no RNG stream involved -- so
that the device path and the CPU oracle are each pinned against the reference independently of one another (VERDICT r02:
every RNG-dependent check went through oracle/rtx_oracle.c, written by the author of the kernels).

Each builder returns (objects, camera, config kwargs, width, height, expected) where `expected` is the exact f64 value of
every channel of every pixel, or a per-pixel array."""
import math

import numpy as np


def _obj(dtype, n):
    return np.zeros(n, dtype=dtype)


def closed_box(dtype, n_spheres=0, seed=5):
    """Six inward one-sided planes (plane.rs:25: a plane is hit only from the side its normal points to) around the camera,
    optionally with spheres inside; EVERY object: emission 1, base colour 1/2, roughness 1.  The ray can never leave, every
    iteration of render_ray's loop hits something, and ray_hit adds light * 1 BEFORE it halves light (scene.rs:276-277), so
    every sample is sum_{k=0}^{max_bounces} 2^-k exactly, whatever the seed, the directions drawn or the object hit; the
    loop runs max_bounces + 1 times (scene.rs:227).  A sphere is never re-hit from its own surface (near root negative,
    sphere.rs:29 + the is_sign_positive filter scene.rs:249), a plane never from a point on it (plane.rs:25)."""
    o = _obj(dtype, 6 + n_spheres)
    h = 9.0
    walls = [((h, 0, 0), (-1, 0, 0)), ((-h, 0, 0), (1, 0, 0)), ((0, h, 0), (0, -1, 0)), ((0, -h, 0), (0, 1, 0)),
             ((0, 0, h), (0, 0, -1)), ((0, 0, -h), (0, 0, 1))]
    for k, (p, n) in enumerate(walls):
        o[k]["kind"] = 1
        o[k]["geom"][:6] = (*p, *n)
    if n_spheres:
        rng = np.random.default_rng(seed)
        c = rng.uniform(-7.0, 7.0, (n_spheres, 3))
        c[:, 0] = np.abs(c[:, 0]) + 1.5                       # in front of the camera, which stays outside every sphere
        o["kind"][6:] = 0
        o["geom"][6:, :3] = c
        o["geom"][6:, 3] = rng.uniform(0.1, 0.6, n_spheres)
    o["emission_color"] = 1.0
    o["base_color"] = 0.5
    o["roughness"] = 1.0
    return o


def closed_box_value(max_bounces):
    return sum(2.0 ** -k for k in range(max_bounces + 1))       # 1.9990234375 for the default 10, 1.875 for 3


def inside_a_sphere(dtype):
    """The camera inside a sphere sees nothing of it: Sphere::distance returns the NEAR root (sphere.rs:29), negative from
    inside, and closest_object drops non-positive distances (scene.rs:249) -> render_ray returns resulting_color = 0 (black),
    however bright the sphere.  Five more spheres behind the camera (never in view) give the tree kernels a tree."""
    o = _obj(dtype, 6)
    o["kind"] = 0
    o[0]["geom"][:4] = (0.5, 0.2, -0.1, 3.0)
    for k in range(1, 6):
        o[k]["geom"][:4] = (-10.0 - 3.0 * k, 2.0 * k - 6.0, 1.0, 1.0)
    o["emission_color"] = 4.0
    o["base_color"] = 0.5
    o["roughness"] = 1.0
    return o


TRI = (5.0, -1.0, -1.0, 5.0, 1.0, -1.0, 5.0, 0.0, 1.0)           # the plane x = 5, seen from the origin
TRI_EMIT, SPH_EMIT = (0.25, 0.5, 0.75), (2.0, 0.125, 1.0)


def triangle_distance_bracket(dtype, direction, distance, delta):
    """Triangle::distance takes |t| of the plane distance and `contains` solves two rows only (triangle.rs:108-127, :37-101):
    from the origin the triangle TRI is reported at 5.0 along +x, at 5.0 along -x (a phantom hit BEHIND the ray) and at
    5.007244751357777 along norm(-1, 0.05, 0.02) (phantom).  The device path exposes no distances, so the value is pinned
    through closest_object's ordering (scene.rs:250): a sphere whose near root is `distance + delta` along the same ray wins
    for delta < 0 and loses for delta > 0.  One pixel, no jitter, a 1e-12 rad field of view: the pixel is the winner's
    emission exactly (0 + 1 * e, base colour 0 ends the path).  Filler shapes far off the ray give the tree kernels trees."""
    d = np.asarray(direction, dtype=np.float64)
    d = d / math.sqrt(float(d @ d))
    o = _obj(dtype, 2 + 6 + 6)
    o[0]["kind"] = 2
    o[0]["geom"] = TRI
    o[0]["emission_color"] = TRI_EMIT
    r = 1.0
    o[1]["kind"] = 0
    o[1]["geom"][:3] = d * (distance + delta + r)               # near root = |c| - r = distance + delta (to ~1e-15)
    o[1]["geom"][3] = r
    o[1]["emission_color"] = SPH_EMIT
    for k in range(6):                                           # filler spheres: beside the ray, never on it
        o[2 + k]["kind"] = 0
        o[2 + k]["geom"][:4] = (3.0 * k - 7.0, 40.0 + k, 25.0, 1.0)
        o[2 + k]["emission_color"] = (9.0, 9.0, 9.0)
    for k in range(6):                                           # filler triangles: tiny, far away in y (no phantom hit on this ray)
        o[8 + k]["kind"] = 2
        base = np.array([2.0 + k, 900.0 + 3.0 * k, 0.3 * k])
        o[8 + k]["geom"] = np.concatenate([base, base + (0.5, 0.1, 0.0), base + (0.1, 0.6, 0.2)])
        o[8 + k]["emission_color"] = (7.0, 7.0, 7.0)
    o["base_color"] = 0.0
    o["roughness"] = 1.0
    cam = ((0.0, 0.0, 0.0), tuple(d), 1e-12)
    cfg = dict(rays_per_pixel=3, focal_offset=0.0, non_focal_offset=0.0, seed=1)
    expected = SPH_EMIT if delta < 0 else TRI_EMIT
    return o, cam, cfg, expected


TRIANGLE_CASES = [((1.0, 0.0, 0.0), 5.0), ((-1.0, 0.0, 0.0), 5.0), ((-1.0, 0.05, 0.02), 5.007244751357777)]
