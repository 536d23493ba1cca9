#!/usr/bin/env python3
"""Where does a device path part from the oracle's?  Renders a scene row by row with the exhaustive f64 kernel, finds the rows whose
segment counts differ from the oracle's (the oracle on the device's sin / cos routine), takes both transcripts of those rows
(rtx_debug_paths of the lab library / rtxo_trace_row) and prints the first step at which a path differs, field by field in hex.

    python tools/transcript_diff.py [c2|mesh]          (a GPU box)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib

rtx = importlib.import_module("rust_raytracing_amd")
from oracle import rtx_oracle as oracle
from rust_raytracing_amd import scenes

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
if which == "c2":
    objs, w, h, cam, cfg = scenes.random_spheres(10000, 1), 160, 90, scenes.CAMERA, dict(rays_per_pixel=4, seed=42)
else:
    objs, w, h, cam, cfg = scenes.axis_aligned_mesh(), 96, 54, scenes.CAMERA, dict(rays_per_pixel=2, seed=42)
oracle.build()
oracle.set_device_sincos(True)
sc = oracle.make_scene(objs, cam, **cfg)
ref, seg = oracle.render(sc, w, h, want_segments=True)
hnd = rtx.Scene.from_packed(rtx.Config(kernel=rtx.RTX_KERNEL_EXACT, **cfg), rtx.Camera(*cam), objs).upload(0, lab=True)
max_steps = 12
bad_rows = 0
for row in range(h):
    steps, counts = hnd.debug_paths(w, h, row, max_steps)
    osteps, ocounts = oracle.trace_row(sc, w, h, row, max_steps)
    assert int(ocounts.sum()) == int(seg[row].sum())
    if np.array_equal(counts, ocounts) and steps.tobytes() == osteps.tobytes():
        continue
    bad_rows += 1
    for x in range(w):
        for s in range(cfg["rays_per_pixel"]):
            n = min(int(max(counts[x, s], ocounts[x, s])), max_steps)
            for b in range(n):
                a, o = steps[x, s, b], osteps[x, s, b]
                if a.tobytes() != o.tobytes():
                    print("row %d x %d sample %d: first difference at step %d (device %d steps, oracle %d)" % (row, x, s, b, counts[x, s], ocounts[x, s]))
                    for f in ("position", "direction", "distance", "object"):
                        av, ov = np.atleast_1d(a[f]), np.atleast_1d(o[f])
                        print("   %-9s device %s\n             oracle %s" % (f, " ".join(v.tobytes()[::-1].hex() for v in av), " ".join(v.tobytes()[::-1].hex() for v in ov)))
                        print("             values %s | %s" % (av, ov))
                    if b:
                        p = steps[x, s, b - 1]
                        print("   previous step: object %d (kind %d) distance %r position %s direction %s" % (p["object"], objs[int(p["object"])]["kind"], float(p["distance"]), p["position"], p["direction"]))
                        print("   its geometry: %s roughness %r" % (objs[int(p["object"])]["geom"], float(objs[int(p["object"])]["roughness"])))
                    break
print("%d of %d rows differ" % (bad_rows, h))
hnd.close()
