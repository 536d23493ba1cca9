"""Shared helpers of the test-suite: one scene description feeds both the oracle and the HIP library."""
import math

import numpy as np

DEFAULT_CAM = ((0.0, 0.0, 0.0), (1.0, 0.0, 0.0), math.pi / 2)


def oracle_render(oracle, objects, width, height, cam=DEFAULT_CAM, want_segments=False, n_threads=None, **cfg):
    sc = oracle.make_scene(objects, cam, **cfg)
    return oracle.render(sc, width, height, n_threads=n_threads, want_segments=want_segments)


def hip_scene(rtx, objects, cam=DEFAULT_CAM, kernel=None, **cfg):
    if kernel is not None:
        cfg["kernel"] = kernel
    config = rtx.Config(**cfg)
    camera = rtx.Camera(*cam)
    return rtx.Scene.from_packed(config, camera, objects)


def hip_render(rtx, objects, width, height, cam=DEFAULT_CAM, kernel=None, **cfg):
    return hip_scene(rtx, objects, cam, kernel, **cfg).render(width, height)


def max_abs_diff(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    d = np.abs(a - b)
    d[both_nan] = 0.0
    return float(np.nanmax(d)) if d.size else 0.0


def fuzz_scene(rtx, rng):
    """Random small scene mixing every shape class the upload treats differently (in or out of the tree, dropped, ...)."""
    ns, nt, npl = int(rng.integers(0, 40)), int(rng.integers(0, 70)), int(rng.integers(0, 3))
    o = np.zeros(ns + nt + npl, dtype=rtx.OBJECT_DTYPE)
    span = float(rng.choice([2.0, 8.0, 40.0]))
    for k in range(ns):
        o[k]["kind"] = 0
        o[k]["geom"][:4] = (*rng.uniform(-span, span, 3), float(rng.choice([0.0, -0.5, rng.uniform(0.05, 0.3 * span)])))
    for k in range(ns, ns + nt):
        o[k]["kind"] = 2
        c = rng.uniform(-span, span, 3)
        cls = int(rng.integers(0, 7))
        e = rng.uniform(-1, 1, (3, 3)) * float(rng.choice([0.05, 0.5, 3.0])) * span / 8
        if cls == 1:   e[:, int(rng.integers(0, 3))] = 0.0                  # axis-aligned plane: pivot-row swaps
        elif cls == 2: e[1] = e[0] * (1 + 1e-9)                             # needle / nearly collinear
        elif cls == 3: e[2] = e[1]                                          # degenerate: two equal vertices
        elif cls == 4: e[:, 2] *= 1e-7                                      # nearly horizontal: z footprint irrelevant
        elif cls == 5: e[:, :2] *= 1e-7                                     # nearly vertical: tiny (x, y) footprint
        o[k]["geom"] = (c[None, :] + e).reshape(9)
    for k in range(ns + nt, ns + nt + npl):
        o[k]["kind"] = 1
        n = rng.normal(size=3)
        o[k]["geom"][:6] = (*rng.uniform(-span, span, 3), *n)
    o["base_color"] = rng.uniform(0.2, 0.9, (len(o), 3))
    o["roughness"] = rng.uniform(0, 1, len(o))
    lit = rng.random(len(o)) < 0.4
    o["emission_color"][lit] = rng.uniform(0.5, 2.0, (int(lit.sum()), 3))
    rng.shuffle(o)                                                           # scene order decides ties
    pos = rng.uniform(-span, span, 3) * float(rng.choice([0.2, 1.0, 3.0]))
    d = rng.normal(size=3)
    if rng.random() < 0.7:
        d = -pos + rng.normal(size=3) * 0.3 * span                          # look (roughly) at the cloud
    if rng.random() < 0.2:
        d = np.eye(3)[int(rng.integers(0, 3))] * float(rng.choice([-1.0, 1.0]))     # axis-parallel view
    cam = (tuple(pos), tuple(d), float(rng.uniform(0.3, 2.5)))
    return o, cam


def sincos_args():
    """random_direction's angles: theta = u * 2 * pi for u in [0, 1) (vector.rs:38), plus the places a range reduction goes wrong:
    both ends, every multiple of pi / 4 and its neighbours (the quadrant boundaries; next to k * pi / 2 the reduced argument is the
    rounding error of pi itself), and tiny angles."""
    rng = np.random.default_rng(11)
    u = np.concatenate([rng.random(1 << 20), rng.integers(0, 1 << 53, 1 << 12) / 2.0 ** 53, [0.0, 2.0 ** -53, 1.0 - 2.0 ** -53, 0.5, 0.25, 0.125]])
    th = u * 2.0 * math.pi
    edges = np.arange(0, 9) * (math.pi / 4)
    edges = np.concatenate([edges, np.nextafter(edges, 10.0), np.nextafter(edges, -10.0)])
    near = np.concatenate([np.arange(0, 9) * (math.pi / 4) + d for d in (0.0, 1e-17, -1e-17, 1e-12, -1e-12, 1e-7, -1e-7, 3e-4, -3e-4)])
    near = near[(near >= 0) & (near <= 2 * math.pi)]
    return np.concatenate([th, edges[(edges >= 0) & (edges <= 2 * math.pi)], near, 10.0 ** rng.uniform(-300, 0, 4096)])
