"""Synthetic scenes of BASELINE.json's configs (the reference ships no scenes).

Generator PRNG: SplitMix64 in counter form -- draw n of a scene is
    mix64(seed + (n + 1) * 0x9E3779B97F4A7C15),  u = (z >> 11) * 2^-53
so every object consumes a FIXED number of draws and the arrays are produced vectorised.
All scenes: camera pos (0,0,0), dir (1,0,0), fov pi/2; Config::default() with rays_per_pixel
per config; render seed 42.
"""
import math

import numpy as np

from .abi import OBJECT_DTYPE, RTX_PLANE, RTX_SPHERE, RTX_TRIANGLE

CAMERA = ((0.0, 0.0, 0.0), (1.0, 0.0, 0.0), math.pi / 2)
RENDER_SEED = 42


def _mix64(z):
    z = z.copy()
    z ^= z >> np.uint64(30); z *= np.uint64(0xBF58476D1CE4E5B9)
    z ^= z >> np.uint64(27); z *= np.uint64(0x94D049BB133111EB)
    z ^= z >> np.uint64(31)
    return z


def splitmix_u01(seed, n):
    """First n SplitMix64 outputs of `seed` as doubles in [0,1) (53-bit)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = _mix64(np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15))
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _materials(out, u_kind, u_c, u_rough):
    """kind u<0.05 -> Material::light(rgb in [0.5,1)*4) else base rgb in [0.2,0.9), roughness in [0,1)."""
    light = u_kind < 0.05
    emission = (0.5 + 0.5 * u_c) * 4.0
    base = 0.2 + 0.7 * u_c
    out["emission_color"] = np.where(light[:, None], emission, 0.0)
    out["base_color"] = np.where(light[:, None], 0.0, base)
    out["roughness"] = np.where(light, 1.0, u_rough)          # Material::light has roughness 1 (object.rs:130-132)


def three_spheres():
    """C1: a white light above, a red diffuse and a glossy grey sphere."""
    o = np.zeros(3, dtype=OBJECT_DTYPE)
    o["kind"] = RTX_SPHERE
    o[0]["geom"][:4] = (6, 0, 8, 5);     o[0]["emission_color"] = (1, 1, 1); o[0]["roughness"] = 1.0       # Material::light
    o[1]["geom"][:4] = (6, -1.2, 0, 1);  o[1]["base_color"] = (0.8, 0.2, 0.2); o[1]["roughness"] = 1.0     # Material::colored
    o[2]["geom"][:4] = (6, 1.2, 0, 1);   o[2]["base_color"] = (0.9, 0.9, 0.9); o[2]["roughness"] = 0.1     # Material::new
    return o


def random_spheres(n=10000, seed=1, box=1.0):
    """C2/C4: n spheres, 9 draws each: cx in [10,110), cy,cz in [-50,50)*box, r in [0.2,1), kind, rgb, roughness."""
    u = splitmix_u01(seed, 9 * n).reshape(n, 9)
    o = np.zeros(n, dtype=OBJECT_DTYPE)
    o["kind"] = RTX_SPHERE
    g = o["geom"]
    g[:, 0] = 10.0 + 100.0 * u[:, 0]
    g[:, 1] = (-50.0 + 100.0 * u[:, 1]) * box
    g[:, 2] = (-50.0 + 100.0 * u[:, 2]) * box
    g[:, 3] = 0.2 + 0.8 * u[:, 3]
    _materials(o, u[:, 4], u[:, 5:8], u[:, 8])
    return o


def random_triangles(n=100000, seed=2, box=1.0):
    """C3/C5: n triangles, 17 draws each: centroid as the spheres, 3 vertex offsets in [-0.5,0.5)^3, material."""
    u = splitmix_u01(seed, 17 * n).reshape(n, 17)
    o = np.zeros(n, dtype=OBJECT_DTYPE)
    o["kind"] = RTX_TRIANGLE
    c = np.stack([10.0 + 100.0 * u[:, 0], (-50.0 + 100.0 * u[:, 1]) * box, (-50.0 + 100.0 * u[:, 2]) * box], axis=1)
    e = u[:, 3:12].reshape(n, 3, 3) - 0.5
    o["geom"] = (c[:, None, :] + e).reshape(n, 9)
    _materials(o, u[:, 12], u[:, 13:16], u[:, 16])
    return o


def axis_aligned_mesh(n_cubes=800, seed=5, span=6.0, x0=8.0):
    """Cubes of random size made of 12 triangles each: every face lies in a plane x, y or z = const, so Triangle::contains
    meets zero pivots and solves most faces in the (y, z) or (x, z) rows (triangle.rs:60-71,81-87) -- the mesh shape real
    models have and the random BASELINE meshes do not."""
    u = splitmix_u01(seed, 8 * n_cubes).reshape(n_cubes, 8)
    c = np.stack([x0 + span * u[:, 0], span * (u[:, 1] - 0.5), span * (u[:, 2] - 0.5)], axis=1)
    hs = 0.05 + 0.25 * u[:, 3]
    corners = np.array([[sx, sy, sz] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float64)
    # two triangles per face, as corner indices (bit 2 = x, bit 1 = y, bit 0 = z)
    faces = [(0, 1, 3), (0, 3, 2), (4, 6, 7), (4, 7, 5), (0, 4, 5), (0, 5, 1), (2, 3, 7), (2, 7, 6), (0, 2, 6), (0, 6, 4), (1, 5, 7), (1, 7, 3)]
    o = np.zeros(12 * n_cubes, dtype=OBJECT_DTYPE)
    o["kind"] = RTX_TRIANGLE
    vtx = c[:, None, :] + hs[:, None, None] * corners[None, :, :]                # (n, 8, 3)
    geom = np.stack([vtx[:, list(f), :].reshape(n_cubes, 9) for f in faces], axis=1)   # (n, 12, 9)
    o["geom"] = geom.reshape(-1, 9)
    uk = np.repeat(u[:, 4], 12); uc = np.repeat(u[:, 5:8], 12, axis=0)
    _materials(o, np.where(uk < 0.3, 0.0, 1.0), uc, np.repeat(u[:, 3], 12))      # 30 % of the cubes are lights
    return o


def compact(objs, k=0.08, x0=4.0):
    """Moves spheres/triangles of the C2/C3 recipes close to the camera (centres x -> x0 + (x-10)*k, y,z -> *k;
    sphere radii * 0.6, triangle shapes kept) so that small test scenes have most rays hitting something."""
    o = objs.copy()
    sph = o["kind"] == RTX_SPHERE
    tri = o["kind"] == RTX_TRIANGLE
    g = o["geom"]
    gs = g[sph]
    gs[:, 0] = x0 + (gs[:, 0] - 10.0) * k
    gs[:, 1] *= k
    gs[:, 2] *= k
    gs[:, 3] *= 0.6
    g[sph] = gs
    tc = g[tri].reshape(-1, 3, 3)
    cen = tc.mean(axis=1, keepdims=True)
    new_cen = cen.copy()
    new_cen[..., 0] = x0 + (cen[..., 0] - 10.0) * k
    new_cen[..., 1] = cen[..., 1] * k
    new_cen[..., 2] = cen[..., 2] * k
    g[tri] = (tc - cen + new_cen).reshape(-1, 9)
    o["geom"] = g
    return o


def light_every(objs, n=4, gain=2.0):
    """Turns every n-th object into a Material::light so that bounced paths pick up colour."""
    o = objs.copy()
    lit = np.arange(len(o)) % n == 0
    o["emission_color"][lit] = (0.5 + 0.5 * o["base_color"][lit]) * gain + o["emission_color"][lit]
    o["base_color"][lit] = 0.0
    o["roughness"][lit] = 1.0
    return o


def mixed_scene(n_spheres=40, n_tris=40, n_planes=1, seed=7):
    """Small scene with all three shape kinds interleaved in scene order (tie-break / ordering tests)."""
    s = light_every(compact(random_spheres(n_spheres, seed)))
    t = light_every(compact(random_triangles(n_tris, seed + 1)))
    p = np.zeros(n_planes, dtype=OBJECT_DTYPE)
    p["kind"] = RTX_PLANE
    for i in range(n_planes):
        p[i]["geom"][:6] = (0, 0, -4.5 - i, 0.05 * i, 0, 1)
        p[i]["base_color"] = (0.7, 0.7, 0.6)
        p[i]["roughness"] = 0.6
    out = np.zeros(n_spheres + n_tris + n_planes, dtype=OBJECT_DTYPE)
    lists = [list(s), list(t), list(p)]          # scene order = round-robin over the three kinds
    k = 0
    while any(lists):
        for lst in lists:
            if lst:
                out[k] = lst.pop(0)
                k += 1
    return out
