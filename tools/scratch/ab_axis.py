import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, numpy as np
import rust_raytracing_amd as rtx
from rust_raytracing_amd import scenes
objs = scenes.light_every(scenes.axis_aligned_mesh(20000, seed=9, span=80.0, x0=20.0), n=6)
w, h, spp = 1920, 1080, 2
hnd = rtx.Scene.from_packed(rtx.Config(rays_per_pixel=spp, seed=42, kernel=5), rtx.Camera(*scenes.CAMERA), objs).upload(0)
buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
for it in range(3):
    st = hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
print(os.environ.get("RTX_HIP_BVH_CLASSIC", "new"), "axis mesh 240k faces: %.1f Mrays/s %.2f ms" % (w*h*spp/st.trace_ms/1e3, st.trace_ms), "seg/ray %.2f box/seg %.1f exact/seg %.2f mean %.6f" % (st.segments/(w*h*spp), st.box_tests/st.segments, st.exact_tests/st.segments, float(buf.mean())))
