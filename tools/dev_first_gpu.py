import sys, time, math, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
os.environ["RTX_HIP_DEBUG"] = "1"
import numpy as np
import rust_raytracing_amd as rtx
from rust_raytracing_amd import scenes
from oracle import rtx_oracle as oracle
from helpers import *
print("devices", rtx.device_count(), flush=True)
cases = [("c1", scenes.three_spheres(), 64, 64, 4), ("sph500", scenes.random_spheres(500, 3), 64, 36, 2),
         ("mixed", scenes.mixed_scene(), 48, 32, 4)]
which = sys.argv[1:] or [c[0] for c in cases]
for name, objs, w, h, spp in cases:
    if name not in which: continue
    ref, seg = oracle_render(oracle, objs, w, h, want_segments=True, rays_per_pixel=spp)
    print(name, "oracle segments", int(seg.sum()), "mean", ref.mean(), flush=True)
    for kern in [int(k) for k in os.environ.get('KERNELS', '1,2,3').split(',')]:
        t0 = time.time()
        try:
            img = hip_render(rtx, objs, w, h, kernel=kern, rays_per_pixel=spp)
        except Exception as e:
            print(name, "kernel", kern, "FAILED", e, flush=True); continue
        dt = time.time() - t0
        print(name, "kernel", kern, "maxdiff", max_abs_diff(img, ref), "bit-identical pixels", int((img == ref).all(axis=2).sum()), "/", w*h, "mean", img.mean(), "nonzero px", int((img != 0).any(axis=2).sum()), "t=%.3f" % dt, flush=True)
