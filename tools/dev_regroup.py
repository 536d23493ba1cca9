"""Dev: lock-step BVH kernel (4) vs regroup schedule (5) on triangle meshes / C2 / the C5 band."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import rust_raytracing_amd as rtx
from rust_raytracing_amd import scenes, tiles
which = sys.argv[1] if len(sys.argv) > 1 else "c5"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 2
w, h, world = 1920, 1080, 1
if which == "c2":
    objs = scenes.random_spheres(10000, 1)
elif which == "c5":
    objs = scenes.random_triangles(1000000, 3, box=2.0); w, h, world = 3840, 2160, 8
else:
    objs = scenes.random_triangles(int(which), 2)
rb, rs, n = tiles.rows_for_rank(h, 0, world)
for kern in (4, 5):
    hnd = rtx.Scene.from_packed(rtx.Config(rays_per_pixel=spp, kernel=kern), rtx.Camera(*scenes.CAMERA), objs).upload(0)
    buf = torch.zeros((n, w, 3), dtype=torch.float64, device="cuda:0")
    for it in range(2):
        st = hnd.render_rows(w, h, rb, rs, n, buf.data_ptr())
    print(which, "spp", spp, "kernel", st.kernel, "thresh", os.environ.get("RTX_HIP_BVH_THRESH", "default"), "trace %.2f ms" % st.trace_ms,
          "Mrays/s %.2f" % (w*n*spp/st.trace_ms/1e3), "Mseg/s %.1f" % (st.segments/st.trace_ms/1e3), "mean %.6f" % float(buf.mean()), flush=True)
    hnd.close()
