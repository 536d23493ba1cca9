import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import rust_raytracing_amd as rtx
from rust_raytracing_amd import scenes, tiles
which = sys.argv[1:] or ["C3", "C5"]
for name in which:
    if name == "C3":
        objs, w, h, spp, world = scenes.random_triangles(100000, 2), 1920, 1080, 8, 1
    else:
        objs, w, h, spp, world = scenes.random_triangles(1000000, 3, box=2.0), 3840, 2160, 4, 8
    rb, rs, n = tiles.rows_for_rank(h, 0, world)
    hnd = rtx.Scene.from_packed(rtx.Config(rays_per_pixel=spp, seed=42), rtx.Camera(*scenes.CAMERA), objs).upload(0)
    buf = tiles.alloc_band(h, w, world, torch.device("cuda", 0))
    for it in range(3):
        st = hnd.render_rows(w, h, rb, rs, n, buf.data_ptr())
    print(name, os.environ.get("RTX_HIP_BVH_CLASSIC", "new"), "kernel", st.kernel, "%.1f Mrays/s" % (n * w * spp / st.trace_ms / 1e3), "%.2f ms" % st.trace_ms,
          "seg/ray %.3f" % (st.segments / (n * w * spp)), "box/seg %.1f" % (st.box_tests / st.segments),
          "leaf/seg %.1f" % ((st.filter_tests - st.box_tests) / st.segments), "exact/seg %.2f" % (st.exact_tests / st.segments), "mean %.9f" % float(buf[:n].mean()), flush=True)
    hnd.close()
