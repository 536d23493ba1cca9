#!/usr/bin/env python3
"""Soak: the tree kernels (BVH lock-step, BVH regroup, wavefront form) and the LDS sweep against the exhaustive f64 kernel, bit for bit, on
the full-size C2 / C3 scenes from several cameras (outside, inside the cloud, looking along each axis).  Prints one line
per (scene, camera) with the number of segments compared; exits non-zero on the first difference."""
import sys, os, math
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np
import torch
import rust_raytracing_amd as rtx
from rust_raytracing_amd import scenes

CAMS = [((0.0, 0.0, 0.0), (1.0, 0.0, 0.0), math.pi / 2),            # the benchmark camera
        ((60.0, 3.0, -2.0), (0.3, 1.0, 0.2), 1.2),                  # inside the cloud
        ((60.0, 0.0, 120.0), (0.0, 0.0, -1.0), 0.9),                # above, looking straight down (along the footprints' unbounded axis)
        ((-400.0, 10.0, 5.0), (1.0, 0.0, 0.0), 0.25),               # far outside the tree's f32 range: f64 slab walk
        ((60.0, -80.0, 0.0), (0.0, 1.0, 0.0), 1.0)]                 # looking along +y


SEED_BASE = int(os.environ.get("SEED_BASE", 1000))     # SPP / SEED_BASE in the environment: another sample of the same scenes
SPP = int(os.environ.get("SPP", 1))


def run(name, objs, w, h, spp, kernels):
    spp = SPP
    total = 0
    for ci, cam in enumerate(CAMS):
        out = {}
        for kern in [rtx.RTX_KERNEL_EXACT] + kernels:
            hnd = rtx.Scene.from_packed(rtx.Config(rays_per_pixel=spp, seed=SEED_BASE + ci, kernel=kern), rtx.Camera(*cam), objs).upload(0)
            buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
            st = hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
            out[repr(kern)] = (buf.cpu().numpy(), st.segments)
            hnd.close()
        ref = out[repr(rtx.RTX_KERNEL_EXACT)]
        for kern in kernels:
            got = out[repr(kern)]
            same = np.array_equal(ref[0], got[0], equal_nan=True) and ref[1] == got[1]
            if not same:
                print("MISMATCH", name, "camera", ci, "kernel", kern, "max |d|", float(np.nanmax(np.abs(ref[0] - got[0]))))
                sys.exit(1)
        total += ref[1]
        print(name, "camera", ci, "segments", ref[1], "mean %.6f" % float(np.nanmean(ref[0])), "identical:", kernels, flush=True)
    return total


if __name__ == "__main__":
    scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
    # the product library's kernels under each tree id and the LDS sweep; LAB=1 adds the lab library's kernel family of each id
    L = rtx.LabKernel
    LABK = [L(rtx.RTX_KERNEL_BVH), L(rtx.RTX_KERNEL_BVH_REGROUP), L(rtx.RTX_KERNEL_WAVEFRONT)] if os.environ.get("LAB") else []
    K = [rtx.RTX_KERNEL_BVH, rtx.RTX_KERNEL_BVH_REGROUP, rtx.RTX_KERNEL_WAVEFRONT, rtx.RTX_KERNEL_MIXED] + LABK
    n = run("C2 10k spheres", scenes.random_spheres(10000, 1), int(1920 * scale), int(1080 * scale), 1, K)
    n += run("C3 100k triangles", scenes.random_triangles(100000, 2), int(960 * scale), int(540 * scale), 1, K)
    n += run("mixed 3k spheres + 30k triangles + 2 planes",
             np.concatenate([scenes.random_spheres(3000, 11), scenes.random_triangles(30000, 12), scenes.mixed_scene(0, 0, 2)]),
             int(960 * scale), int(540 * scale), 1, K)
    aam = scenes.axis_aligned_mesh(4000, seed=9, span=80.0, x0=20.0)          # 48k cube faces: footprints in all three planes
    n += run("axis-aligned mesh 48k faces + 2k spheres",
             np.concatenate([aam, scenes.random_spheres(2000, 13)]), int(960 * scale), int(540 * scale), 1, K)
    if "c5" in sys.argv[2:]:                      # 1M triangles: the exhaustive kernel needs ~10 s per camera at this size
        n += run("C5 1M triangles", scenes.random_triangles(1000000, 3, box=2.0), int(480 * scale), int(270 * scale), 1,
                 [rtx.RTX_KERNEL_BVH, rtx.RTX_KERNEL_BVH_REGROUP, rtx.RTX_KERNEL_WAVEFRONT] + LABK)
    print("soak ok:", n, "segments compared bit for bit")
