"""Variants against each other on a BASELINE config (same box, same process): time, rates, counters and bit equality of the frames.
    tools/ab_kernels.py CONFIG SPP SPEC [SPEC ...] [band]
        SPEC = [L]KERNEL[:TUNING]   KERNEL = RTX_KERNEL_* id (L4: the lab library's family of id 4), TUNING = RtxConfig.tuning bits
                                    (decimal or 0x..; a RTX_TUNE_LAB_MASK bit renders through librtx_hip_lab.so)
        e.g.  C2 64 4 4:8 4:32      (two stages with packets | stage 1 per lane | one stage)
              C3 8 5 6   |   C5 4 5 6 band
    W / H override the frame size, MAXB max_bounces, SCENE=axis:N[:mixed] the scene, BLOCK the row block of the band;
    RTX_HIP_LIB selects another build of the library (tools/build_variant.sh)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import rust_raytracing_amd as rtx
from rust_raytracing_amd import scenes, tiles

args = [a for a in sys.argv[1:] if a != "band"]
name, spp, specs = args[0], int(args[1]), args[2:]
world = 8 if "band" in sys.argv else 1
cfg = bench.CONFIGS[name]
objs = bench.make_objects(cfg)
if os.environ.get("SCENE", "").startswith("axis"):                    # SCENE=axis:20000 -> 240k axis-aligned cube faces (+ SCENE=axis:20000:mixed: 2k spheres)
    import numpy as np
    parts = os.environ["SCENE"].split(":")
    objs = scenes.axis_aligned_mesh(int(parts[1]), seed=9, span=80.0, x0=20.0)
    if len(parts) > 2:
        objs = np.concatenate([objs, scenes.random_spheres(2000, 13)])
w, h = int(os.environ.get("W", cfg["w"])), int(os.environ.get("H", cfg["h"]))
block = int(os.environ.get("BLOCK", tiles.ROW_BLOCK))
part = tiles.Partition(h, 0, world, block)
dev = torch.device("cuda", 0)
out = {}
for spec in specs:
    k, _, t = spec.partition(":")
    k, t = (rtx.LabKernel(int(k[1:])) if k.startswith("L") else int(k)), int(t, 0) if t else 0
    hnd = rtx.Scene.from_packed(rtx.Config(rays_per_pixel=spp, seed=scenes.RENDER_SEED, kernel=k, tuning=t, max_bounces=int(os.environ.get("MAXB", 10)), non_focal_offset=float(os.environ.get("NFO", 0.1))), rtx.Camera(*scenes.CAMERA), objs).upload(0)
    band = part.alloc_band(w, dev)
    part.render(hnd, w, band)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        st = part.render(hnd, w, band, want_stats=True)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    out[spec] = band[:part.n_rows].clone()
    rays = part.n_rows * w * spp
    print("%-8s (ran %d): %.2f ms  trace %.2f ms%s  %.1f Mrays/s  seg %d exact/seg %.3f box/seg %.1f launches %d" % (
        spec, st.kernel, min(ts) * 1e3, st.trace_ms, "  (stage 1 %.2f ms)" % st.stage1_ms if getattr(st, "stage1_ms", 0.0) else "",
        rays / min(ts) / 1e6, st.segments, st.exact_tests / max(st.segments, 1),
        st.box_tests / max(st.segments, 1), st.trace_launches), flush=True)
    if os.environ.get("LAB"):
        print("    filter_tests - box_tests = %d" % (st.filter_tests - st.box_tests), flush=True)
    hnd.close()
ref = out[specs[0]].view(torch.int64)
same = all(torch.equal(ref, out[s].view(torch.int64)) for s in specs[1:])
print("bit-identical:", same, flush=True)
sys.exit(0 if same else 1)
