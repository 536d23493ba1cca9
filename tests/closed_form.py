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


# ---- the bounce direction (scene.rs:279-292, vector.rs:36-45), pinned without the oracle ------------------------------------
# The scenes above are direction-independent by construction.  These two are not: a wrong reflection formula, blend weight,
# hemisphere flip or direction distribution changes their pixels.
MIRROR_BASE, MIRROR_EMIT, LIGHT_EMIT = (0.5, 0.25, 0.75), (0.125, 0.0, 0.25), (4.0, 2.0, 1.0)


def _norm(v):
    v = np.asarray(v, dtype=np.float64)
    return v / math.sqrt(float(v @ v))


def _fillers(dtype, centre, below):
    """Six black spheres and six black triangles that no path of these scenes can reach before it ends -- `below` a floor, or far
    behind everything -- so that the tree kernels get their trees.  (A phantom triangle hit, triangle.rs:118, would add 0 like the
    miss it replaces: base colour and emission are 0.)"""
    o = _obj(dtype, 12)
    c = np.asarray(centre, dtype=np.float64)
    for k in range(6):
        o[k]["kind"] = 0
        o[k]["geom"][:4] = (*(c + np.asarray(below) * (40.0 + 3.0 * k) + (2.0 * k, 0.0, 0.0)), 1.0)
        o[6 + k]["kind"] = 2
        base = c + np.asarray(below) * 50.0 + (0.3 * k, 900.0 + 3.0 * k, 0.0)
        o[6 + k]["geom"] = np.concatenate([base, base + (0.5, 0.1, 0.0), base + (0.1, 0.6, 0.0)])
    o["roughness"] = 1.0
    return o


def mirror_cases(dtype):
    """ROUGHNESS 0: random_bounce_dir returns norm(w + (refl - w) * 1) = the mirror direction refl = d - (n * 2) * (d . n) up to
    rounding, whatever w was drawn (scene.rs:281-285), flipped to the normal's side when refl . n <= 0 (scene.rs:287-291).  One
    pixel, no jitter, a 1e-12 rad field of view: the primary ray leaves the origin along +x, hits the mirror (base MIRROR_BASE,
    emission MIRROR_EMIT) at P, and the bounced ray either meets a light sphere (emission LIGHT_EMIT, base 0, radius 1) centred
    10 units along the expected direction -- pixel = MIRROR_EMIT + MIRROR_BASE * LIGHT_EMIT exactly (dyadic values; the light's
    base colour 0 ends the path, scene.rs:228) -- or, with the light displaced by 2.5 > its radius, nothing: pixel = MIRROR_EMIT.
    Mirrors: an axis-aligned plane, a plane with an oblique un-normalised normal (object.rs:37-39 normalises it), a sphere
    (normal = norm(P - c)), and a triangle seen from BEHIND (its normal is never turned towards the ray, triangle.rs:29: refl
    points behind it and the flip sends the ray on through it -- along +x again for this incidence).
    Yields (name, objects, camera, config kwargs, expected rgb)."""
    d = np.array([1.0, 0.0, 0.0])
    cases = []
    # (name, mirror kind, geom, hit point P, unit normal as the reference computes it, direction the fillers hide in)
    cases.append(("plane x=4 facing the camera", 1, (4.0, 0.0, 0.0, -1.0, 0.0, 0.0), (4.0, 0.0, 0.0), (-1.0, 0.0, 0.0), None))
    cases.append(("oblique plane, normal (-3, 0, 1.5)", 1, (4.0, 0.0, 0.0, -3.0, 0.0, 1.5), (4.0, 0.0, 0.0), _norm((-3.0, 0.0, 1.5)), None))
    cases.append(("oblique plane, normal (-1, 2, 0.5) through (6, 1, 1)", 1, (6.0, 1.0, 1.0, -1.0, 2.0, 0.5), None, _norm((-1.0, 2.0, 0.5)), None))
    cases.append(("sphere (5, 0.6, 0) r 1", 0, (5.0, 0.6, 0.0, 1.0), (4.2, 0.0, 0.0), (-0.8, -0.6, 0.0), None))
    cases.append(("triangle in x = 5 seen from behind", 2, TRI, (5.0, 0.0, 0.0), (1.0, 0.0, 0.0), None))
    out = []
    for name, kind, geom, P, n, _ in cases:
        n = np.asarray(n, dtype=np.float64)
        if P is None:                                            # plane through `pos`: the point of the x-axis on it
            pos, nn = np.asarray(geom[:3]), np.asarray(geom[3:6])
            P = (float(pos @ nn) / nn[0], 0.0, 0.0)
        P = np.asarray(P, dtype=np.float64)
        refl = d - (n * 2.0) * float(d @ n)                      # scene.rs:281
        f = _norm(refl)
        if not float(f @ n) > 0.0:                               # scene.rs:287-291
            f = -f
        side = _norm(np.cross(f, (0.0, 0.3, 1.0)))               # a direction perpendicular to the bounced ray
        for displaced in (False, True):
            o = _obj(dtype, 2)
            o[0]["kind"] = kind
            o[0]["geom"][:len(geom)] = geom
            o[0]["base_color"], o[0]["emission_color"], o[0]["roughness"] = MIRROR_BASE, MIRROR_EMIT, 0.0
            o[1]["kind"] = 0
            o[1]["geom"][:3] = P + f * 10.0 + (side * 2.5 if displaced else 0.0)
            o[1]["geom"][3] = 1.0
            o[1]["base_color"], o[1]["emission_color"], o[1]["roughness"] = 0.0, LIGHT_EMIT, 1.0
            # fillers behind the camera, off the x-axis and away from every bounced ray of these cases
            fill = _fillers(dtype, (-60.0, 35.0, 20.0), (-1.0, 0.0, 0.0))
            objs = np.concatenate([o, fill])
            want = tuple(m + (0.0 if displaced else b * e) for m, b, e in zip(MIRROR_EMIT, MIRROR_BASE, LIGHT_EMIT))
            cfg = dict(rays_per_pixel=3, focal_offset=0.0, non_focal_offset=0.0, seed=5)
            out.append((name + (" (light displaced)" if displaced else ""), objs, ((0.0, 0.0, 0.0), (1.0, 0.0, 0.0), 1e-12), cfg, want))
    return out


HEMI_BASE, HEMI_EMIT, HEMI_R, HEMI_D = 0.5, 4.0, 3.0, 5.0


def hemisphere_light(dtype):
    """ROUGHNESS 1: random_bounce_dir returns norm(w + (refl - w) * 0) = w, the direction Vector3::random_direction drew --
    z = 2u - 1, theta = 2 pi u', (sqrt(1 - z^2) cos, sqrt(1 - z^2) sin, z): uniform on the sphere (vector.rs:36-45) -- flipped to the
    normal's side: uniform on the hemisphere.  A floor plane z = -1 (base HEMI_BASE, emission 0, roughness 1) under a light
    sphere (emission HEMI_EMIT, base 0) of radius HEMI_R whose centre is HEMI_D above the point P the primary rays hit; fillers
    hide below the floor.  A bounced ray meets the light iff it lies in the cone of half angle asin(r / d) about the normal:
    probability 1 - sqrt(1 - (r/d)^2) (the cap's share of the hemisphere's solid angle) = 0.2 for r/d = 0.6; each sample is
    HEMI_BASE * HEMI_EMIT = 2 with that probability and 0 otherwise (the light's base colour 0 ends the path; a ray that misses
    leaves the scene).  A cosine-weighted hemisphere would give (r/d)^2 = 0.36, a sphere without the flip 0.1.
    Returns (objects, camera, config kwargs without seed / rays_per_pixel, p, sample value)."""
    d = _norm((1.0, 0.0, -1.0))
    P = np.array([1.0, 0.0, -1.0])
    o = _obj(dtype, 2)
    o[0]["kind"] = 1
    o[0]["geom"][:6] = (0.0, 0.0, -1.0, 0.0, 0.0, 1.0)
    o[0]["base_color"], o[0]["roughness"] = HEMI_BASE, 1.0
    o[1]["kind"] = 0
    o[1]["geom"][:4] = (P[0], P[1], P[2] + HEMI_D, HEMI_R)
    o[1]["emission_color"], o[1]["roughness"] = HEMI_EMIT, 1.0
    fill = _fillers(dtype, (0.0, 0.0, -1.0), (0.0, 0.0, -1.0))
    p = 1.0 - math.sqrt(1.0 - (HEMI_R / HEMI_D) ** 2)
    cam = ((0.0, 0.0, 0.0), tuple(d), 1e-12)
    return np.concatenate([o, fill]), cam, dict(focal_offset=0.0, non_focal_offset=0.0), p, HEMI_BASE * HEMI_EMIT


def hemisphere_check(mean, n_samples, p, value, sigmas=5.0):
    """(|mean - mu|, bound): the grand mean of n_samples Bernoulli(p) * value samples lies within sigmas * sigma / sqrt(N) of
    mu = p * value (5 sigma: a false alarm once in 1.7e6 runs; the alternatives named above are > 100 sigma away)."""
    mu = p * value
    bound = sigmas * value * math.sqrt(p * (1.0 - p) / n_samples)
    return abs(mean - mu), bound
