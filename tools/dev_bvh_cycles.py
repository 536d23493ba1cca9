"""Dev: where the BVH kernel's wave cycles go (needs a -DRTX_BVH_STATS build selected with RTX_HIP_LIB).
STATS build reports: exact_tests = wave traversal steps, filter_tests = wave cycles in the traversal loop, box_tests = other wave cycles.
usage: dev_bvh_cycles.py c2|c5|<n_triangles> [spp]"""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
import rust_raytracing_amd as rtx
from rust_raytracing_amd import scenes, tiles
which = sys.argv[1] if len(sys.argv) > 1 else "c2"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
w, h, world = 1920, 1080, 1
if which == "c2":
    objs = scenes.random_spheres(10000, 1)
elif which == "c5":
    objs = scenes.random_triangles(1000000, 3, box=2.0); w, h, world = 3840, 2160, 8
else:
    objs = scenes.random_triangles(int(which), 2)
t0 = time.time()
hnd = rtx.Scene.from_packed(rtx.Config(rays_per_pixel=spp, kernel=int(os.environ.get("KERN", "4"))), rtx.Camera(*scenes.CAMERA), objs).upload(0)
t_upload = time.time() - t0
rb, rs, n = tiles.rows_for_rank(h, 0, world)
buf = torch.zeros((n, w, 3), dtype=torch.float64, device="cuda:0")
for it in range(2):
    st = hnd.render_rows(w, h, rb, rs, n, buf.data_ptr())
tot = st.filter_tests + st.box_tests
print(which, "spp", spp, "upload %.0f ms" % (t_upload * 1e3), "trace %.2f ms" % st.trace_ms, "Mrays/s %.1f" % (w*n*spp/st.trace_ms/1e3), "segments", st.segments,
      "wave-steps/seg*64 %.1f" % (st.exact_tests * 64 / st.segments),
      "cycles: traversal %.1f%% other %.1f%%" % (100 * st.filter_tests / tot, 100 * st.box_tests / tot),
      "cycles/wave-step %.0f" % (st.filter_tests / max(st.exact_tests, 1)), "other cycles per 64 segments %.0f" % (st.box_tests * 64 / st.segments))
