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
