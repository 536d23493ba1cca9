#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configs on ONE MI355X (they are parity-test cases, not bench.py lines):
C3 = 100k-triangle mesh, 1920x1080 (full frame, reduced spp), C5 = 1M-triangle mesh, 3840x2160 (the band one rank of 8
owns, reduced spp), C4 = 10k spheres 3840x2160 (band of one rank of 8, reduced spp).  Mrays/s is a rate; the spp used is
stated.  Writes one JSON object per config to stdout."""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import rust_raytracing_amd as rtx
from rust_raytracing_amd import scenes, tiles


def run(name, objs, w, h, spp, world, rank, kernel=rtx.RTX_KERNEL_AUTO, reps=2):
    hnd = rtx.Scene.from_packed(rtx.Config(rays_per_pixel=spp, seed=42, kernel=kernel), rtx.Camera(*scenes.CAMERA), objs).upload(0)
    rb, rs, n = tiles.rows_for_rank(h, rank, world)
    buf = torch.zeros((n, w, 3), dtype=torch.float64, device="cuda:0")
    best = None
    for _ in range(reps):
        st = hnd.render_rows(w, h, rb, rs, n, buf.data_ptr())
        if best is None or st.trace_ms < best.trace_ms:
            best = st
    hnd.close()
    st = best
    rays = n * w * spp
    out = {"config": name, "kernel": int(st.kernel), "width": w, "height": h, "rows_traced": n, "rays_per_pixel": spp,
           "primary_rays": rays, "trace_ms": st.trace_ms, "Mrays_per_s": rays / st.trace_ms / 1e3,
           "Msegments_per_s": st.segments / st.trace_ms / 1e3, "segments_per_ray": st.segments / rays,
           "exact_tests_per_segment": st.exact_tests / max(st.segments, 1), "box_tests_per_segment": st.box_tests / max(st.segments, 1),
           "image_mean": float(buf.mean())}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["c3", "c3_sweep", "c4", "c5"]
    if "c3" in which:
        run("C3: 100k triangles (seed 2), 1920x1080, 16 spp (named config: 64), AUTO", scenes.random_triangles(100000, 2), 1920, 1080, 16, 1, 0)
    if "c3_sweep" in which:
        run("C3: 100k triangles (seed 2), 1920x1080, 2 spp, LDS sweep kernel", scenes.random_triangles(100000, 2), 1920, 1080, 2, 1, 0,
            kernel=rtx.RTX_KERNEL_MIXED, reps=1)
    if "c4" in which:
        run("C4: 10k spheres (seed 1), 3840x2160, band of rank 0 of 8, 64 spp (named config: 1024), AUTO", scenes.random_spheres(10000, 1),
            3840, 2160, 64, 8, 0)
    if "c5" in which:
        run("C5: 1M triangles (seed 3, box x2), 3840x2160, band of rank 0 of 8, 4 spp (named config: 256), AUTO",
            scenes.random_triangles(1000000, 3, box=2.0), 3840, 2160, 4, 8, 0)
