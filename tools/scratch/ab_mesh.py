import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import rust_raytracing_amd as rtx
from rust_raytracing_amd import scenes, tiles
for name in ["C3", "C5"]:
    if name == "C3": objs, w, h, spp, world = scenes.random_triangles(100000, 2), 1920, 1080, 8, 1
    else: objs, w, h, spp, world = scenes.random_triangles(1000000, 3, box=2.0), 3840, 2160, 4, 8
    rb, rs, n = tiles.rows_for_rank(h, 0, world)
    for kern in (5, 6):
        hnd = rtx.Scene.from_packed(rtx.Config(rays_per_pixel=spp, seed=42, kernel=kern), rtx.Camera(*scenes.CAMERA), objs).upload(0)
        buf = tiles.alloc_band(h, w, world, torch.device("cuda", 0))
        for it in range(3): st = hnd.render_rows(w, h, rb, rs, n, buf.data_ptr())
        print(name, "kernel", kern, "%.1f Mrays/s %.2f ms" % (n*w*spp/st.trace_ms/1e3, st.trace_ms), "box/seg %.1f exact/seg %.2f mean %.9f" % (st.box_tests/st.segments, st.exact_tests/st.segments, float(buf[:n].mean())), flush=True)
        hnd.close()
