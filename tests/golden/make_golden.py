"""Generates tests/golden/*.npz and rng_vectors.json with the CPU oracle (oracle/rtx_oracle.c).

The reference itself cannot run here (Rust, no toolchain), and its own tests hold only the four
camera KATs (src/raytracing/camera.rs:82-109), which tests/test_oracle_kats.py states directly.
These fixtures therefore pin the ORACLE's behaviour (RNG draw order, shading order, tie-break,
image layout) against regressions; they are not outputs of the reference ("parity unpinned"
beyond the camera basis -- see oracle/rtx_oracle.h).

Run from the repository root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import rtx_oracle as oracle          # noqa: E402
from rust_raytracing_amd import scenes           # noqa: E402

CASES = {
    # name: (objects factory, width, height, config)
    "c1_three_spheres_32x32": (scenes.three_spheres, 32, 32, dict(rays_per_pixel=4, seed=42)),
    "spheres200_48x27": (lambda: scenes.random_spheres(200, 11, box=0.2), 48, 27, dict(rays_per_pixel=2, seed=7)),
    "mixed_40x24": (lambda: scenes.mixed_scene(), 40, 24, dict(rays_per_pixel=3, seed=5, max_bounces=4)),
    "tris300_32x18": (lambda: scenes.light_every(scenes.compact(scenes.random_triangles(300, 2)), 3), 32, 18,
                      dict(rays_per_pixel=2, seed=9)),
}


def main():
    for name, (factory, w, h, cfg) in CASES.items():
        objs = factory()
        img, seg = oracle.render(oracle.make_scene(objs, scenes.CAMERA, **cfg), w, h, want_segments=True, n_threads=4)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), image=img, segments=seg,
                            objects=objs.view(np.uint8).reshape(len(objs), -1),
                            config=json.dumps(cfg), width=w, height=h)
        print(name, img.shape, "mean", img.mean(), "segments", int(seg.sum()))
    vec = []
    for seed, pix, smp in [(0, 0, 0), (42, 0, 0), (42, 1, 0), (42, 0, 1), (2**63 + 5, 123456789, 63), (7, 2**32 + 1, 1023)]:
        key = oracle.lib().rtxo_rng_key(seed, pix, smp)
        vec.append(dict(seed=seed, pixel=pix, sample=smp, key=int(key),
                        u=[oracle.rng_u01(seed, pix, smp, k).hex() for k in range(8)]))
    with open(os.path.join(HERE, "rng_vectors.json"), "w") as f:
        json.dump(vec, f, indent=1)
    print("rng_vectors.json", len(vec))


if __name__ == "__main__":
    main()
