import sys, time, math, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, R + "/tests")
os.environ["RTX_HIP_DEBUG"] = "1"
import numpy as np
import rust_raytracing_amd as rtx
from rust_raytracing_amd import scenes
from helpers import *
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 4
kern = int(sys.argv[2]) if len(sys.argv) > 2 else 2
w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
objs = scenes.random_spheres(10000, 1)
for it in range(2):
    t0 = time.time()
    img = hip_render(rtx, objs, w, h, kernel=kern, rays_per_pixel=spp)
    dt = time.time() - t0
    print("kernel", kern, "spp", spp, "wall %.3f s" % dt, "Mrays/s (wall, incl upload+copy) %.1f" % (w*h*spp/dt/1e6), "mean", img.mean(), flush=True)
