"""Kernel A against kernel B on a BASELINE config (same box, same process): time, rates, counters and bit equality of the frames.
    tools/ab_kernels.py CONFIG SPP KERNEL_A KERNEL_B [band]      e.g.  C3 8 5 6   |   C5 4 5 6 band   (W / H override the frame size, MAXB max_bounces, SCENE=axis:N[:mixed] the scene;
    RTX_HIP_LIB selects another build of the library, see tools/build_variant.sh)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import rust_raytracing_amd as rtx
from rust_raytracing_amd import scenes, tiles

name, spp, ka, kb = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
world = 8 if len(sys.argv) > 5 and sys.argv[5] == "band" else 1
cfg = bench.CONFIGS[name]
objs = bench.make_objects(cfg)
if os.environ.get("SCENE", "").startswith("axis"):                    # SCENE=axis:20000 -> 240k axis-aligned cube faces (+ SCENE=axis:20000:mixed: 2k spheres)
    import numpy as np
    parts = os.environ["SCENE"].split(":")
    objs = scenes.axis_aligned_mesh(int(parts[1]), seed=9, span=80.0, x0=20.0)
    if len(parts) > 2:
        objs = np.concatenate([objs, scenes.random_spheres(2000, 13)])
w, h = int(os.environ.get("W", cfg["w"])), int(os.environ.get("H", cfg["h"]))
rb, rs, n_rows = tiles.rows_for_rank(h, 0, world)
dev = torch.device("cuda", 0)
out = {}
for k in (ka, kb):
    hnd = rtx.Scene.from_packed(rtx.Config(rays_per_pixel=spp, seed=scenes.RENDER_SEED, kernel=k, max_bounces=int(os.environ.get("MAXB", 10))), rtx.Camera(*scenes.CAMERA), objs).upload(0)
    band = tiles.alloc_band(h, w, world, dev)
    hnd.render_rows(w, h, rb, rs, n_rows, band.data_ptr())
    torch.cuda.synchronize()
    ts = []
    for _ in range(2):
        t0 = time.perf_counter()
        st = hnd.render_rows(w, h, rb, rs, n_rows, band.data_ptr())
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    out[k] = band[:n_rows].clone()
    rays = n_rows * w * spp
    print("kernel %d (ran %d): %.2f ms  trace %.2f ms  %.1f Mrays/s  seg %d exact/seg %.3f box/seg %.1f launches %d" % (
        k, st.kernel, min(ts) * 1e3, st.trace_ms, rays / min(ts) / 1e6, st.segments, st.exact_tests / max(st.segments, 1),
        st.box_tests / max(st.segments, 1), st.trace_launches), flush=True)
    hnd.close()
same = torch.equal(out[ka].view(torch.int64), out[kb].view(torch.int64))
print("bit-identical:", same, flush=True)
sys.exit(0 if same else 1)
