"""Dev: where the lock-step BVH kernel's wave cycles go (needs a -DRTX_BVH_STATS build via RTX_HIP_LIB + RTX_HIP_BVH_LOCKSTEP=1).
STATS build reports: exact_tests = wave traversal steps, filter_tests = wave cycles in the traversal loop, box_tests = other wave cycles."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import rust_raytracing_amd as rtx
from rust_raytracing_amd import scenes
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
objs = scenes.random_spheres(10000, 1) if which == "c2" else scenes.random_triangles(int(which), 2)
w, h = 1920, 1080
hnd = rtx.Scene.from_packed(rtx.Config(rays_per_pixel=spp, kernel=rtx.RTX_KERNEL_BVH), rtx.Camera(*scenes.CAMERA), objs).upload(0)
buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
for it in range(2):
    st = hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
tot = st.filter_tests + st.box_tests
print(which, "spp", spp, "trace %.2f ms" % st.trace_ms, "Mrays/s %.1f" % (w*h*spp/st.trace_ms/1e3), "segments", st.segments,
      "wave-steps/seg*64 %.1f" % (st.exact_tests * 64 / st.segments),
      "cycles: traversal %.1f%% other %.1f%%" % (100 * st.filter_tests / tot, 100 * st.box_tests / tot),
      "cycles/wave-step %.0f" % (st.filter_tests / max(st.exact_tests, 1)), "other cycles per 64 segments %.0f" % (st.box_tests * 64 / st.segments))
