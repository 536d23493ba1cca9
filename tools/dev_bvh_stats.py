import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import rust_raytracing_amd as rtx
from rust_raytracing_amd import scenes
objs = scenes.random_spheres(10000, 1)
w, h, spp = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 8
for kern in (rtx.RTX_KERNEL_BVH, rtx.RTX_KERNEL_MIXED):
    hnd = rtx.Scene.from_packed(rtx.Config(rays_per_pixel=spp, kernel=kern), rtx.Camera(*scenes.CAMERA), objs).upload(0)
    buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
    for it in range(2):
        st = hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
    print("kernel", kern, "trace %.2f ms" % st.trace_ms, "Mrays/s %.1f" % (w*h*spp/st.trace_ms/1e3), "segments", st.segments,
          "exact/seg %.2f" % (st.exact_tests/st.segments), "filter/seg %.1f" % (st.filter_tests/st.segments), "box/seg %.1f" % (st.box_tests/st.segments))
    hnd.close()
