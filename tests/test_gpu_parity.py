"""Parity of the HIP path with the CPU oracle, through the C ABI (python -m pytest tests -m gpu).

Tolerance (stated once, used everywhere): the arithmetic is f64 in the reference's operation order, / and
sqrt are correctly rounded on gfx950 (test_device_div_sqrt_are_correctly_rounded), and the column/row
sin/cos of get_ray_dir come from host libm, so the only divergence from the oracle is the device's
sin/cos in random_direction (vector.rs:38-42; rtx_math.h sincos_2pi), which differ from glibc's by <= 1 ulp on
~15 % of arguments -- as two libms do.  A 1-ulp difference moves a pixel by ~1e-15 unless it flips a discrete hit/miss decision.
    ATOL = 1e-9 absolute per channel, on every pixel (no outlier allowance needed in these cases).
Integer outputs (segment counts, u8 images, camera matrices, RNG) are compared bit-exactly.
That the sin / cos IS the only divergence is itself tested: the oracle can be switched to the device's routine
(oracle.device_sincos(): the same operations on the CPU), and then every image must equal the oracle's BIT FOR BIT
(test_hip_equals_the_oracle_bit_for_bit_with_the_device_sincos) -- also on the axis-aligned meshes where a last-place
difference in a bounce direction flips a path.
"""
import json
import math
import os
import subprocess

import numpy as np
import pytest

from helpers import DEFAULT_CAM, fuzz_scene, hip_render, hip_scene, max_abs_diff, oracle_render, sincos_args

pytestmark = pytest.mark.gpu
ATOL = 1e-9
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def gpu(rtx):
    if rtx.device_count() < 1:
        pytest.fail("no gfx950 device: the gpu tests must run on an MI355X (there is no CPU fallback to test)")
    return rtx


def two_sided(oracle, img, objs, w, h, cam=DEFAULT_CAM, flips=1, **cfg):
    """The tolerance statement where a path DOES flip on the last bit of a sin / cos: equal, bit for bit, to the oracle that uses the
    device's routine; within ATOL of the libm oracle on all but `flips` pixels (whose paths took another turn)."""
    with oracle.device_sincos():
        assert np.array_equal(img, oracle_render(oracle, objs, w, h, cam=cam, **cfg), equal_nan=True)
    off = int((np.abs(img - oracle_render(oracle, objs, w, h, cam=cam, **cfg)).max(axis=2) > ATOL).sum())
    assert off <= flips, off
    return off


def _kernels(rtx):
    """Every kernel id through the product library (librtx_hip.so: one tree-kernel family per kind of tree, whichever tree id is
    asked for) and the three tree ids through the lab library (librtx_hip_lab.so: one family per id -- round 1's lock-step and
    regrouping kernels, the pool kernel, the all-levels wavefront forms).  Same bits everywhere."""
    L = rtx.LabKernel
    return [rtx.RTX_KERNEL_EXACT, rtx.RTX_KERNEL_MIXED, rtx.RTX_KERNEL_BVH, rtx.RTX_KERNEL_BVH_REGROUP, rtx.RTX_KERNEL_WAVEFRONT,
            L(rtx.RTX_KERNEL_BVH), L(rtx.RTX_KERNEL_BVH_REGROUP), L(rtx.RTX_KERNEL_WAVEFRONT)]


# ---- device arithmetic ---------------------------------------------------------------------------
def test_device_div_sqrt_are_correctly_rounded(gpu):
    rng = np.random.default_rng(0)
    a = np.concatenate([rng.uniform(1e-3, 1e3, 1 << 18), 10.0 ** rng.uniform(-200, 200, 1 << 18), [1e-310, 4.0, 2.0]])
    b = np.concatenate([rng.uniform(1e-3, 1e3, 1 << 18), 10.0 ** rng.uniform(-100, 100, 1 << 18), [3.0, 1e-300, 7.0]])
    assert np.array_equal(gpu.debug_math(0, a, b), a / b)
    assert np.array_equal(gpu.debug_math(1, a), np.sqrt(a))


def test_device_sin_cos_within_one_ulp_of_libm(gpu, oracle):
    """Ops 9 / 10: the sin / cos the path uses for random_direction's angle (rtx_math.h sincos_2pi) over that angle's range and
    the hard places of a range reduction -- within one ulp of this libm's, and bit for bit what the oracle's restatement of the
    routine gives on the CPU.  Ops 2 / 3 (the device library's general sin / cos, not on the path) for comparison."""
    th = sincos_args()
    s, c = oracle.device_sincos_values(th)
    for op, ref, same in ((9, np.sin(th), s), (10, np.cos(th), c)):
        got = gpu.debug_math(op, th)
        assert np.all(np.abs(got - ref) <= np.spacing(np.abs(ref)))
        assert np.array_equal(got, same)
    th = np.random.default_rng(1).uniform(0, 2 * math.pi, 1 << 18)
    for op, ref in ((2, np.sin(th)), (3, np.cos(th))):
        got = gpu.debug_math(op, th)
        assert np.all(np.abs(got - ref) <= np.spacing(np.abs(ref)))


def test_writelane_and_f64_minmax_kats(gpu):
    """Two toolchain / mode assumptions of the walks, checked on the device (rtx_debug_math ops 6-8, rtx_traverse.h):
    * rtx_writelane binds llvm.amdgcn.writelane.i32 by name (this clang has no builtin): lane L of every wave takes the value,
      every other lane keeps its own;
    * the child sort orders (key, link) pairs with v_min_f64 / v_max_f64 on raw bit patterns: a key of +0.0 makes the pair an
      f64 DENORMAL whose payload is the link -- it must come back bit for bit (denormals preserved), as must patterns with a
      +inf key; a flushed pair would send a walk to node 0 for ever."""
    n = 1024
    a = np.arange(n, dtype=np.float64) + 7.0
    for lane in (0, 1, 31, 32, 63):
        b = np.zeros(n)
        b[0], b[1] = -123456.0, float(lane)
        got = gpu.debug_math(6, a, b)
        want = a.copy()
        want[lane::64] = -123456.0
        assert np.array_equal(got, want), lane
    rng = np.random.default_rng(3)
    keys = np.concatenate([np.zeros(300, np.float32), rng.uniform(0, 1e3, 300).astype(np.float32), np.full(100, np.inf, np.float32),
                           rng.uniform(0, 1e-38, 100).astype(np.float32), -rng.uniform(0, 1e-3, 224).astype(np.float32)])
    keys[-1] = -0.0
    links = rng.integers(0, 1 << 32, size=(2, n), dtype=np.uint64)
    pa = (keys[rng.permutation(n)].view(np.uint32).astype(np.uint64) << 32) | links[0]
    pb = (keys[rng.permutation(n)].view(np.uint32).astype(np.uint64) << 32) | links[1]
    lo = gpu.debug_math(7, pa.view(np.float64), pb.view(np.float64)).view(np.uint64)
    hi = gpu.debug_math(8, pa.view(np.float64), pb.view(np.float64)).view(np.uint64)
    # as f64 values (no NaN among them): sign-magnitude order of the 64-bit patterns
    def order(u):
        s = u.astype(np.int64)
        return np.where(s < 0, ~s ^ np.int64(-(2 ** 63)), s)
    oa, ob = order(pa), order(pb)
    want_lo, want_hi = np.where(oa <= ob, pa, pb), np.where(oa <= ob, pb, pa)
    eq = oa == ob
    assert np.array_equal(lo[~eq], want_lo[~eq]) and np.array_equal(hi[~eq], want_hi[~eq])
    assert np.all((lo[eq] == pa[eq]) | (lo[eq] == pb[eq]))
    assert int((pa >> 32 == 0).sum()) > 100                     # the denormal pairs (key +0.0) were among them


# ---- golden fixtures and seeded scenes: HIP == oracle ------------------------------------------------
@pytest.mark.parametrize("name", ["c1_three_spheres_32x32", "spheres200_48x27", "mixed_40x24", "tris300_32x18"])
def test_hip_matches_golden(gpu, oracle, name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    objs = np.frombuffer(z["objects"].tobytes(), dtype=gpu.OBJECT_DTYPE)
    cfg = json.loads(str(z["config"]))
    w, h = int(z["width"]), int(z["height"])
    for kern in _kernels(gpu):
        img = hip_render(gpu, objs, w, h, kernel=kern, **cfg)
        assert max_abs_diff(img, z["image"]) <= ATOL, (name, kern)


@pytest.mark.parametrize("case", ["c1", "spheres2k", "mixed", "tris", "tris20k"])
def test_hip_matches_oracle_seeded(gpu, oracle, case):
    from rust_raytracing_amd import scenes
    objs, w, h, cfg = {
        "c1": (scenes.three_spheres(), 256, 256, dict(rays_per_pixel=1, seed=42)),           # BASELINE configs[0], full size
        "spheres2k": (scenes.compact(scenes.random_spheres(2000, 1), k=0.3), 96, 54, dict(rays_per_pixel=4, seed=42)),
        "mixed": (scenes.mixed_scene(60, 50, 2, seed=21), 64, 40, dict(rays_per_pixel=4, seed=3)),
        "tris": (scenes.light_every(scenes.compact(scenes.random_triangles(400, 5)), 3), 48, 32, dict(rays_per_pixel=3, seed=8)),
        "tris20k": (scenes.random_triangles(20000, 6, box=0.3), 64, 36, dict(rays_per_pixel=2, seed=4)),   # C3 recipe, denser
    }[case]
    ref, seg = oracle_render(oracle, objs, w, h, want_segments=True, **cfg)
    for kern in (gpu.RTX_KERNEL_EXACT, gpu.RTX_KERNEL_MIXED, gpu.RTX_KERNEL_MIXED_VERIFY, gpu.RTX_KERNEL_BVH, gpu.RTX_KERNEL_BVH_REGROUP):
        scene = hip_scene(gpu, objs, kernel=kern, **cfg)
        img = scene.render(w, h)
        assert max_abs_diff(img, ref) <= ATOL, (case, kern)
        assert np.isfinite(img).all()


def test_hip_equals_the_oracle_bit_for_bit_with_the_device_sincos(gpu, oracle):
    """The stated tolerance has ONE source: random_direction's sin / cos (libm on the CPU, sincos_2pi on the device).  With the
    oracle switched to the device's routine (the same operations, on the CPU) nothing is left to differ, and every image and
    segment count must be EQUAL: the golden scenes, the seeded scenes, 60 fuzz scenes, and the axis-aligned mesh with bouncing
    materials -- the case where the libm oracle and the kernels part on a handful of pixels (a hit point on a face x = const lands
    on the face or one ulp off; a last-place difference in a bounce direction flips that for a later hit)."""
    import torch
    from rust_raytracing_amd import scenes
    kerns = (gpu.RTX_KERNEL_EXACT, gpu.RTX_KERNEL_AUTO, gpu.RTX_KERNEL_BVH, gpu.RTX_KERNEL_WAVEFRONT)
    cases = []
    for name in ("c1_three_spheres_32x32", "spheres200_48x27", "mixed_40x24", "tris300_32x18"):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        cases.append((name, np.frombuffer(z["objects"].tobytes(), dtype=gpu.OBJECT_DTYPE), int(z["width"]), int(z["height"]), DEFAULT_CAM,
                      json.loads(str(z["config"]))))
    cases += [("c1", scenes.three_spheres(), 256, 256, DEFAULT_CAM, dict(rays_per_pixel=1, seed=42)),
              ("spheres2k", scenes.compact(scenes.random_spheres(2000, 1), k=0.3), 96, 54, DEFAULT_CAM, dict(rays_per_pixel=4, seed=42)),
              ("mixed", scenes.mixed_scene(60, 50, 2, seed=21), 64, 40, DEFAULT_CAM, dict(rays_per_pixel=4, seed=3)),
              ("tris20k", scenes.random_triangles(20000, 6, box=0.3), 64, 36, DEFAULT_CAM, dict(rays_per_pixel=2, seed=4)),
              ("c2 recipe", scenes.random_spheres(10000, 1), 160, 90, scenes.CAMERA, dict(rays_per_pixel=4, seed=42))]
    mesh = scenes.axis_aligned_mesh()
    for cam in (scenes.CAMERA, ((11.0, 0.2, 0.1), (0.3, 1.0, 0.2), 1.4), ((11.0, 0.0, 6.0), (0.0, 0.0, -1.0), 1.2)):
        cases.append(("axis-aligned mesh", mesh, 96, 54, cam, dict(rays_per_pixel=2, seed=42)))
    rng = np.random.default_rng(77)
    for it in range(60):
        objs, cam = fuzz_scene(gpu, rng)
        cases.append(("fuzz %d" % it, objs, 20, 12, cam, dict(rays_per_pixel=2, seed=it, max_bounces=int(rng.choice([3, 10])))))
    differ = 0
    for name, objs, w, h, cam, cfg in cases:
        libm = oracle_render(oracle, objs, w, h, cam=cam, **cfg)
        with oracle.device_sincos():
            ref, seg = oracle_render(oracle, objs, w, h, cam=cam, want_segments=True, **cfg)
        differ += not np.array_equal(libm, ref, equal_nan=True)
        for kern in kerns:
            hnd = hip_scene(gpu, objs, cam=cam, kernel=kern, **cfg).upload(0)
            buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
            st = hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
            hnd.close()
            assert np.array_equal(buf.cpu().numpy(), ref, equal_nan=True), (name, kern)
            assert st.segments == int(seg.sum()), (name, kern)
    assert differ >= 2                     # (images are functions of hit sequences: a last-place change shows in few of them -- but
                                           #  it does: the two forms of the oracle are not the same function, the switch was live)


def test_path_transcripts_equal_the_oracles(gpu, oracle):
    """An image compares hit SEQUENCES (a colour is a product of materials and a light); a transcript compares the arithmetic: per
    segment the ray closest_object was asked about, the winning distance, the winner (rtx_debug_paths of the lab library -- the
    exhaustive f64 kernel's loop, writing down -- against rtxo_trace_row).  Every path of every row of four scenes (spheres; spheres,
    planes and triangles; a random mesh; the axis-aligned mesh):
      * against the libm oracle the PRIMARY ray (step 0: scene.rs:194-207, the host's trig tables, the lens draws) and its hit are
        equal to the last bit -- this caught glibc's sincos() differing from its sin() / cos() when one side's compiler had merged
        the pair and the other's had not;
      * against the oracle on the device's sin / cos routine EVERY step is equal to the last bit: positions, bounce directions,
        distances, objects, lengths."""
    from rust_raytracing_amd import scenes
    cases = [("c2 recipe", scenes.random_spheres(10000, 1), 160, 90, scenes.CAMERA, dict(rays_per_pixel=4, seed=42)),
             ("mixed", scenes.mixed_scene(60, 50, 2, seed=21), 64, 40, DEFAULT_CAM, dict(rays_per_pixel=4, seed=3)),
             ("tris", scenes.light_every(scenes.compact(scenes.random_triangles(400, 5)), 3), 48, 32, DEFAULT_CAM, dict(rays_per_pixel=3, seed=8)),
             ("axis-aligned mesh", scenes.axis_aligned_mesh(), 96, 54, ((11.0, 0.2, 0.1), (0.3, 1.0, 0.2), 1.4), dict(rays_per_pixel=2, seed=42))]
    max_steps, bounced = 12, 0
    for name, objs, w, h, cam, cfg in cases:
        sc = oracle.make_scene(objs, cam, **cfg)
        hnd = hip_scene(gpu, objs, cam=cam, kernel=gpu.RTX_KERNEL_EXACT, **cfg).upload(0, lab=True)
        for row in range(h):
            steps, counts = hnd.debug_paths(w, h, row, max_steps)
            libm_steps, _ = oracle.trace_row(sc, w, h, row, max_steps)
            assert steps[:, :, 0].tobytes() == libm_steps[:, :, 0].tobytes(), (name, row)
            with oracle.device_sincos():
                want, want_counts = oracle.trace_row(sc, w, h, row, max_steps)
            assert np.array_equal(counts, want_counts), (name, row)
            assert steps.tobytes() == want.tobytes(), (name, row)
            bounced += int((counts > 1).sum())
        hnd.close()
    assert bounced > 5000
    # the product library has no such hook
    hnd = hip_scene(gpu, scenes.three_spheres(), kernel=gpu.RTX_KERNEL_EXACT, rays_per_pixel=1).upload(0)
    with pytest.raises(gpu.RtxError) as e:
        hnd.debug_paths(8, 8, 0, 4)
    assert e.value.status == gpu.abi.RTX_ERR_UNSUPPORTED
    hnd.close()


def test_mixed_kernel_is_bit_identical_to_exact_kernel(gpu):
    """The f32 filter may only discard spheres the f64 test would reject: same bits, same segment count,
    zero filter mismatches -- at a size the oracle would need minutes for."""
    import torch
    from rust_raytracing_amd import scenes
    objs = scenes.random_spheres(10000, 1)
    w, h, spp = 320, 180, 4
    cam = gpu.Camera(*scenes.CAMERA)
    out = {}
    for kern in (gpu.RTX_KERNEL_EXACT, gpu.RTX_KERNEL_MIXED_VERIFY, gpu.RTX_KERNEL_BVH):
        hnd = gpu.Scene.from_packed(gpu.Config(rays_per_pixel=spp, kernel=kern), cam, objs).upload(0)
        buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
        st = hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
        out[kern] = (buf.cpu().numpy(), st.segments, st.filter_mismatches, st.exact_tests, st.filter_tests)
        hnd.close()
    a, b, c = out[gpu.RTX_KERNEL_EXACT], out[gpu.RTX_KERNEL_MIXED_VERIFY], out[gpu.RTX_KERNEL_BVH]
    assert np.array_equal(a[0], b[0])
    assert a[1] == b[1] and b[2] == 0
    assert a[0].mean() > 0.01
    # the BVH kernel visits a tiny fraction of the 10^4 spheres per segment and still produces the same bits
    assert np.array_equal(a[0], c[0]) and a[1] == c[1]
    assert c[3] < a[3] / 50 and c[4] > 0


# ---- values that follow from the reference's text alone: no oracle involved -------------------------------------------
def test_closed_form_values_pin_the_device_path(gpu):
    """tests/closed_form.py: scenes whose pixels the reference's TEXT determines exactly -- a closed box where every sample is
    sum 2^-k (scene.rs:227, :276-277, plane.rs:25), the camera inside a sphere (sphere.rs:29, scene.rs:249: black), and
    SURVEY 8c's triangle distances 5.0 / 5.0 (phantom) / 5.007244751357777 (phantom) bracketed to 1e-9 through
    closest_object's ordering -- for every kernel, with no oracle in the loop."""
    import closed_form as cf
    dt = gpu.OBJECT_DTYPE
    cam = ((0.3, -0.2, 0.1), (1.0, 0.1, -0.05), 1.5)
    for n_sph in (0, 40):
        box = cf.closed_box(dt, n_sph)
        for kern in _kernels(gpu):
            for mb, seed, spp in ((10, 1, 1), (10, 77, 5), (3, 4, 3), (0, 9, 2)):
                img = hip_render(gpu, box, 24, 16, cam=cam, kernel=kern, rays_per_pixel=spp, seed=seed, max_bounces=mb)
                assert np.all(img == cf.closed_box_value(mb)), (n_sph, kern, mb)
    for kern in _kernels(gpu):
        assert not hip_render(gpu, cf.inside_a_sphere(dt), 24, 16, kernel=kern, rays_per_pixel=2).any(), kern
        for direction, dist in cf.TRIANGLE_CASES:
            for delta in (-1e-9, 1e-9):
                objs, tcam, cfg, want = cf.triangle_distance_bracket(dt, direction, dist, delta)
                img = hip_render(gpu, objs, 1, 1, cam=tcam, kernel=kern, **cfg)
                assert tuple(img.ravel()) == tuple(want), (kern, direction, delta)


def test_bounce_direction_is_pinned_without_the_oracle(gpu):
    """random_bounce_dir / random_direction (scene.rs:279-292, vector.rs:36-45) on the device against values the reference's TEXT
    determines (tests/closed_form.py), no oracle in the loop, for every kernel: (i) roughness 0 -- the bounced ray is the mirror
    direction d - (n * 2) * (d . n), flipped to the normal's side: a light centred on it is met (pixel = mirror emission + mirror
    base * light emission, exactly), a light displaced by more than its radius is not; planes, a sphere, a triangle seen from
    behind.  (ii) roughness 1 -- uniform on the hemisphere: of 2^22 bounced rays the share that meets a light subtending a cone
    of half angle asin(0.6) is 1 - sqrt(1 - 0.36) = 0.2 within 5 sigma (cosine-weighted would be 0.36, no flip 0.1)."""
    import closed_form as cf
    dt = gpu.OBJECT_DTYPE
    for name, objs, cam, cfg, want in cf.mirror_cases(dt):
        for kern in _kernels(gpu):
            img = hip_render(gpu, objs, 1, 1, cam=cam, kernel=kern, **cfg)
            assert tuple(img.ravel()) == tuple(want), (name, kern)
    objs, cam, cfg, p, value = cf.hemisphere_light(dt)
    for kern in _kernels(gpu):
        for tune in ((0, gpu.RTX_TUNE_TWO_STAGE) if int(kern) == gpu.RTX_KERNEL_BVH else (0,)):
            means = [hip_render(gpu, objs, 64, 64, cam=cam, kernel=kern, rays_per_pixel=256, seed=seed, tuning=tune, **cfg).mean(axis=(0, 1))
                     for seed in (21, 22, 23, 24)]
            err, bound = cf.hemisphere_check(float(np.mean(means)), 4 * 64 * 64 * 256, p, value)
            assert err <= bound, (kern, tune, err, bound)


def test_closed_box_at_two_stage_size_every_ray_survives(gpu):
    """The same closed box with 300 spheres at 1024 x 1024 (1.05e6 rays: AUTO = the sphere kernel in two stages, stage 1 as
    packets): EVERY primary ray survives its first hit, so the survivors' queue takes as many records as the launch has
    rays -- the case its capacity and chunk reservation are sized for; a record it could not take raises the launch's
    watchdog word (an error from render_rows / close) instead of vanishing.  Every pixel is 1.9990234375 exactly."""
    import torch
    import closed_form as cf
    box = cf.closed_box(gpu.OBJECT_DTYPE, 300, seed=8)
    w = h = 1024
    for tune in (0, gpu.RTX_TUNE_STAGE2_SLOTS, gpu.RTX_TUNE_NO_PACKETS, gpu.RTX_TUNE_NO_CUT, gpu.RTX_TUNE_INLINE_LEAVES, gpu.RTX_TUNE_STAGE2_PAIR,
                 gpu.RTX_TUNE_STAGE2_POOL):
        hnd = hip_scene(gpu, box, cam=((0.3, -0.2, 0.1), (1.0, 0.1, -0.05), 1.5), rays_per_pixel=1, seed=3, tuning=tune).upload(0)
        buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
        st = hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
        assert st.kernel == gpu.RTX_KERNEL_BVH and st.segments == 11 * w * h and st.stage1_ms > 0
        assert bool((buf == cf.closed_box_value(10)).all())
        buf.zero_()
        hnd.render_rows(w, h, 0, 1, h, buf.data_ptr(), want_stats=False)          # asynchronous: a raised watchdog word surfaces in close()
        hnd.close()
        assert bool((buf == cf.closed_box_value(10)).all())


# ---- properties that hold at any size --------------------------------------------------------------------
def test_emission_only_scene_is_seed_independent_and_exact(gpu, oracle):
    from rust_raytracing_amd import scenes
    objs = scenes.compact(scenes.random_spheres(300, 5))
    objs["base_color"] = 0.0
    objs["emission_color"] = np.round(np.random.default_rng(5).uniform(0, 4, size=(300, 3)) * 8) / 8
    cfg = dict(rays_per_pixel=4, focal_offset=0.0, non_focal_offset=0.0)
    ref = oracle_render(oracle, objs, 64, 36, seed=1, **cfg)
    for kern in _kernels(gpu):
        a = hip_render(gpu, objs, 64, 36, kernel=kern, seed=1, **cfg)
        b = hip_render(gpu, objs, 64, 36, kernel=kern, seed=77, **cfg)
        assert np.array_equal(a, b) and np.array_equal(a, ref)          # no RNG-dependent arithmetic: bit-exact


def test_tie_break_first_object_wins(gpu):
    from rust_raytracing_amd import scenes
    o = scenes.three_spheres()[:2].copy()
    o[0]["geom"][:4] = (6, 0, 0, 2); o[1]["geom"][:4] = (6, 0, 0, 2)
    o["base_color"] = 0.0
    o[0]["emission_color"] = (1, 0, 0); o[1]["emission_color"] = (0, 1, 0)
    cfg = dict(rays_per_pixel=1, focal_offset=0.0, non_focal_offset=0.0)
    for kern in _kernels(gpu):
        assert hip_render(gpu, o, 16, 16, kernel=kern, **cfg)[8, 8].tolist() == [1.0, 0.0, 0.0]
        assert hip_render(gpu, o[::-1].copy(), 16, 16, kernel=kern, **cfg)[8, 8].tolist() == [0.0, 1.0, 0.0]


def test_triangle_filter_classes_and_no_mismatch(gpu, oracle):
    """The f32 triangle filter of the sweep kernel must never lose a hit the exact test reports -- including the
    reference's phantom hits.  Scene: random triangles plus the special classes the upload treats separately:
    axis-aligned triangles (pivot-row swaps -> "always candidate"), needle triangles (ill-conditioned projection),
    degenerate ones ("can't handle LGS"), triangles behind / around the camera (undecided cull, |n.v0| <= 1), and a
    triangle that the direction-based cull (triangle.rs:115) always rejects."""
    import torch
    from rust_raytracing_amd import scenes
    base = scenes.light_every(scenes.compact(scenes.random_triangles(600, 4)), 3)
    extra = np.zeros(9, dtype=gpu.OBJECT_DTYPE)
    extra["kind"] = 2
    geoms = [
        (6, -1, -1, 6, 1, -1, 6, 0, 1),                 # in the plane x = 6: r.x = s.x = 0 -> row swaps
        (5, -2, 0.5, 7, -2, 0.5, 6, 2, 0.5),            # in the plane z = 0.5
        (4, 0, 0, 9, 1e-7, 0, 6.5, 5e-8, 3),            # needle in the xy projection (ill-conditioned)
        (5, 0, 0, 6, 0, 0, 7, 0, 0),                    # collinear: degenerate
        (0.3, -0.2, -0.2, 0.3, 0.2, -0.2, 0.3, 0, 0.3),  # just in front of the camera: |n.v0| < 1
        (-3, -1, -1, -3, 1, -1, -3, 0, 1),              # entirely behind the camera: phantom hits (abs of the distance)
        (7, 2, 2, 7.5, 2, 2.2, 7.2, 2.6, 2.1),
        (3, 1, -1, 3, -1, -1, 3, 0, 1),                 # reversed winding of a facing triangle: always culled
        (8, -3, 1, 8.5, -3.5, 1.5, 8.2, -2.6, 1.1),
    ]
    for k, g in enumerate(geoms):
        extra[k]["geom"] = g
        extra[k]["emission_color"] = (0.25 * (k + 1), 0.5, 1.0)
        extra[k]["roughness"] = 1.0
    objs = np.concatenate([base[:300], extra, base[300:]])
    cfg = dict(rays_per_pixel=3, seed=13)
    w, h = 72, 48
    ref, seg = oracle_render(oracle, objs, w, h, want_segments=True, **cfg)
    assert ref.mean() > 0.01
    for kern in (gpu.RTX_KERNEL_EXACT, gpu.RTX_KERNEL_MIXED, gpu.RTX_KERNEL_BVH, gpu.RTX_KERNEL_BVH_REGROUP, gpu.RTX_KERNEL_AUTO):
        assert max_abs_diff(hip_render(gpu, objs, w, h, kernel=kern, **cfg), ref) <= ATOL
    hnd = hip_scene(gpu, objs, kernel=gpu.RTX_KERNEL_MIXED_VERIFY, **cfg).upload(0)
    buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
    st = hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
    hnd.close()
    assert st.filter_mismatches == 0 and st.segments == int(seg.sum())
    assert max_abs_diff(buf.cpu().numpy(), ref) <= ATOL
    # the filter did real work: far fewer exact tests than segments x triangles
    assert st.exact_tests < 0.2 * st.segments * len(objs)


@pytest.mark.parametrize("n_tris,w,h,spp", [(100000, 240, 135, 2), (600000, 96, 54, 1)])
def test_bvh_triangle_footprint_tree_is_bit_identical_to_exact_kernel(gpu, n_tris, w, h, spp):
    """Triangles are in the BVH with their (x, y) footprint and unbounded z (rtx_bvh.h): the reference's phantom hits
    (|t| of the plane distance, 2-row containment) survive the culling.  C3-recipe meshes at sizes the oracle would
    need minutes for, against the exhaustive f64 kernel: same bits, same segment count, ~1e4 x fewer exact tests.
    The 600k mesh is deep enough for the traversal stack to spill from LDS to HBM (SPILL variant)."""
    import torch
    from rust_raytracing_amd import scenes
    objs = scenes.random_triangles(n_tris, 2)
    cam = gpu.Camera(*scenes.CAMERA)
    out = {}
    for kern in (gpu.RTX_KERNEL_EXACT, gpu.RTX_KERNEL_BVH, gpu.RTX_KERNEL_BVH_REGROUP, gpu.RTX_KERNEL_AUTO, gpu.LabKernel(gpu.RTX_KERNEL_BVH)):
        hnd = gpu.Scene.from_packed(gpu.Config(rays_per_pixel=spp, kernel=kern), cam, objs).upload(0)
        buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
        st = hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
        out[repr(kern) if getattr(kern, "lab", False) else kern] = (buf.cpu().numpy(), st.segments, st.exact_tests, st.kernel)
        hnd.close()
    a, b, c = out[gpu.RTX_KERNEL_EXACT], out[gpu.RTX_KERNEL_BVH], out[gpu.RTX_KERNEL_AUTO]
    lock = out["lab:%d" % gpu.RTX_KERNEL_BVH]                          # round 1's lock-step kernel (librtx_hip_lab.so)
    assert lock[3] == gpu.RTX_KERNEL_BVH and np.array_equal(a[0], lock[0]) and a[1] == lock[1]
    assert a[0].mean() > 0.01 and np.isfinite(a[0]).all()
    assert np.array_equal(a[0], b[0]) and a[1] == b[1]
    g = out[gpu.RTX_KERNEL_BVH_REGROUP]
    assert np.array_equal(a[0], g[0]) and a[1] == g[1]
    assert b[2] < a[2] / 1000
    assert c[3] in (gpu.RTX_KERNEL_BVH, gpu.RTX_KERNEL_BVH_REGROUP) and np.array_equal(a[0], c[0])       # AUTO picks the tree for triangle meshes


def test_axis_aligned_mesh_stays_inside_the_tree(gpu, oracle):
    """A mesh of axis-aligned cubes: most faces are solved by Triangle::contains in the (y, z) or (x, z) rows
    (zero-pivot swaps, triangle.rs:60-71,81-87).  They sit in the tree with footprints in those planes (none is left to
    the per-segment sweep, so AUTO keeps the tree kernels).  Checks, from the benchmark camera, from inside the mesh and
    looking straight along each axis:
      * every kernel's frame and segment count equal the exhaustive f64 kernel's bit for bit;
      * with every face a light (paths end at the first hit: no bounce, no sin/cos) every pixel equals the oracle's;
      * with bouncing materials all but a handful of pixels equal the oracle's.  The handful is the stated sin/cos
        divergence meeting this geometry: a hit point on a face x = const is p.x = o.x + d.x * ((v0.x - o.x) / d.x),
        which lands on v0.x exactly or one ulp off; the bounced ray's self-test then returns distance 0 (filtered by
        is_normal, scene.rs:249: the ray leaves) or 1e-16 (a self-hit: it bounces again).  A 1-ulp difference in a bounce
        direction (device sincos vs glibc, <= 1 ulp on ~3 % of arguments) flips that coin for a later hit, and the path
        changes -- between two libm versions of the reference itself just as well."""
    import torch
    from rust_raytracing_amd import scenes
    objs = scenes.axis_aligned_mesh()
    st = gpu.debug_host_scene(gpu.Scene.from_packed(gpu.Config(), gpu.Camera(*scenes.CAMERA), objs))
    assert st["tri_in_tree"] == st["tri_filter_records"] and st["tri_other_footprints"] > 3000
    lights = objs.copy()
    lights["emission_color"] = 0.25 + 0.75 * np.abs(np.sin(np.arange(len(objs))[:, None] * np.array([0.37, 0.61, 0.83])))
    lights["base_color"] = 0.0
    cams = [scenes.CAMERA, ((11.0, 0.2, 0.1), (0.3, 1.0, 0.2), 1.4), ((11.0, 0.0, 6.0), (0.0, 0.0, -1.0), 1.2),
            ((11.0, -6.0, 0.0), (0.0, 1.0, 0.0), 1.2), ((2.0, 0.0, 0.0), (1.0, 0.0, 0.0), 0.8)]
    w, h, spp = 96, 54, 2

    def run(ob, cam, kern):
        hnd = hip_scene(gpu, ob, cam=cam, kernel=kern, rays_per_pixel=spp, seed=42).upload(0)
        buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
        stt = hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
        hnd.close()
        return buf.cpu().numpy(), stt.segments, stt.kernel

    lit = flipped = 0
    for cam in cams:
        ref_l = oracle_render(oracle, lights, w, h, cam=cam, rays_per_pixel=spp, seed=42)
        ref_b = oracle_render(oracle, objs, w, h, cam=cam, rays_per_pixel=spp, seed=42)
        lit += ref_l.mean() > 0.01
        for ob, ref, exact_everywhere in ((lights, ref_l, True), (objs, ref_b, False)):
            imgs = {repr(kern): run(ob, cam, kern) for kern in [gpu.RTX_KERNEL_AUTO] + _kernels(gpu)}
            assert imgs[repr(gpu.RTX_KERNEL_AUTO)][2] == gpu.RTX_KERNEL_BVH_REGROUP   # the tree holds the mesh: no fallback to the sweep
            ex = imgs[repr(gpu.RTX_KERNEL_EXACT)]
            for kern in imgs:
                assert np.array_equal(imgs[kern][0], ex[0]) and imgs[kern][1] == ex[1], (cam, kern)
            bad = int((np.abs(ex[0] - ref).max(axis=2) > ATOL).sum())
            if exact_everywhere:
                assert bad == 0, cam
            else:
                assert bad <= 0.004 * w * h, (cam, bad)
                flipped += bad
    assert lit >= 3
    # the same mesh next to spheres and a random mesh: four sub-trees under the joint nodes
    mix = np.concatenate([scenes.compact(scenes.random_spheres(400, 3), k=0.06, x0=9.0), lights[::3],
                          scenes.light_every(scenes.compact(scenes.random_triangles(900, 4), k=0.06, x0=9.0))])
    stm = gpu.debug_host_scene(gpu.Scene.from_packed(gpu.Config(), gpu.Camera(*scenes.CAMERA), mix))
    assert stm["flags"] == 3 and stm["tri_other_footprints"] > 900 and stm["tri_xy_footprints"] > 300
    ex = hip_render(gpu, mix, w, h, kernel=gpu.RTX_KERNEL_EXACT, rays_per_pixel=spp, seed=7)
    for kern in _kernels(gpu):
        assert np.array_equal(hip_render(gpu, mix, w, h, kernel=kern, rays_per_pixel=spp, seed=7), ex), kern
    ref = oracle_render(oracle, mix, w, h, rays_per_pixel=spp, seed=7)
    assert int((np.abs(ex - ref).max(axis=2) > ATOL).sum()) <= 0.004 * w * h and ref.mean() > 0.01


def test_spheres_kernel_paths(gpu, oracle):
    """trace_bvh_spheres_kernel (RTX_KERNEL_BVH on a tree without triangle leaves: the f32-only traversal loop with
    conservative distance bounds, exact tests after the walk) against the exhaustive f64 kernel bit for bit, on the paths
    C2 does not reach: a tree deep enough for the HBM stack column, origins far outside the scene (f64 slab walk) and
    beyond any walk (every sphere tested), axis-parallel rays, more coincident candidates than the 4-entry queue holds,
    nested shells (every shell a certain hit, the nearest wins), and spheres whose radius is at the f32 resolution of
    their centre (no certain hit exists: the bounds only ever add candidates).  The same step under its two other
    schedules runs beside it: RTX_KERNEL_BVH_REGROUP (trace_bvh_spheres_pool_kernel: a pool of rays per lane, waiting lanes
    served together) and RTX_KERNEL_WAVEFRONT (walk / shade kernels per bounce level, far origins with the f32 slack)."""
    import torch
    from rust_raytracing_amd import scenes

    def both(objs, cam, w=64, h=36, spp=2, scratch=0, **cfg):
        # RTX_KERNEL_BVH three times: one stage (what a frame this small takes), and forced into the two-stage form of big
        # frames -- stage 1 as one packet walk per 8x8 tile (trace_sph_packet_kernel) and per lane --, so that every edge
        # case below also crosses the packets, the survivors' queue and the queue-fed stage 2
        out = []
        # (a lab tuning bit, or a LabKernel id, renders through librtx_hip_lab.so; in the product library the three tree ids all
        #  run the sphere kernels on this tree)
        L, BVH = gpu.LabKernel, gpu.RTX_KERNEL_BVH
        for kern, tune, ran in ((BVH, 0, BVH), (BVH, gpu.RTX_TUNE_TWO_STAGE, BVH),
                                (BVH, gpu.RTX_TUNE_TWO_STAGE | gpu.RTX_TUNE_NO_CUT, BVH),
                                (BVH, gpu.RTX_TUNE_TWO_STAGE | gpu.RTX_TUNE_STAGE2_SLOTS, BVH),
                                (gpu.RTX_KERNEL_BVH_REGROUP, 0, BVH), (gpu.RTX_KERNEL_WAVEFRONT, gpu.RTX_TUNE_TWO_STAGE, BVH),
                                (L(BVH), 0, BVH), (L(BVH), gpu.RTX_TUNE_TWO_STAGE, BVH),
                                (BVH, gpu.RTX_TUNE_TWO_STAGE | gpu.RTX_TUNE_NO_PACKETS, BVH),
                                (BVH, gpu.RTX_TUNE_TWO_STAGE | gpu.RTX_TUNE_STAGE2_PAIR, BVH),
                                (BVH, gpu.RTX_TUNE_TWO_STAGE | gpu.RTX_TUNE_INLINE_LEAVES, BVH),
                                (BVH, gpu.RTX_TUNE_TWO_STAGE | gpu.RTX_TUNE_STAGE2_POOL | gpu.RTX_TUNE_NO_QNODES, BVH),
                                (L(gpu.RTX_KERNEL_BVH_REGROUP), 0, gpu.RTX_KERNEL_BVH_REGROUP),
                                (L(gpu.RTX_KERNEL_WAVEFRONT), 0, gpu.RTX_KERNEL_WAVEFRONT), (gpu.RTX_KERNEL_EXACT, 0, gpu.RTX_KERNEL_EXACT)):
            hnd = hip_scene(gpu, objs, cam=cam, kernel=kern, rays_per_pixel=spp, seed=42, tuning=tune, **cfg).upload(0)
            hnd.set_scratch_limit(scratch)
            buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
            st = hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
            hnd.close()
            assert st.kernel == ran, (kern, tune, st.kernel)
            out.append((buf.cpu().numpy(), st.segments, st.exact_tests))
        for o in out[:-1]:
            assert np.array_equal(o[0], out[-1][0]) and o[1] == out[-1][1]
        return out[0]

    deep = scenes.random_spheres(300000, 11)
    st = gpu.debug_host_scene(gpu.Scene.from_packed(gpu.Config(), gpu.Camera(*scenes.CAMERA), deep))
    assert st["flags"] == 1 + 16 and st["stack_bound"] > 30                      # 3 * depth + 2 exceeds the LDS stack
    img, segs, exact = both(deep, scenes.CAMERA)
    assert img.mean() > 0.01 and exact < 8 * segs, (exact, segs)              # (no exhaustive fallback: about one exact test per segment)
    c2 = scenes.random_spheres(10000, 1)
    for cam in (((-400.0, 30.0, 10.0), (1.0, -0.05, 0.0), 0.5),             # outside origin_limit: the f64 slab walk
                ((-3.0e12, 0.0, 0.0), (1.0, 0.0, 0.0), 1e-10),              # beyond it: every sphere, exactly
                ((60.0, 0.0, 0.0), (0.0, 0.0, 1.0), 1.2),                   # inside the cloud
                ((60.0, -80.0, 0.0), (0.0, 1.0, 0.0), 1e-9)):               # (almost) axis-parallel rays
        img, segs, exact = both(c2, cam, focal_offset=0.0, non_focal_offset=0.0)
    # 9 coincident spheres in front of everything: 9 live candidates > kSphQueue -> the segment's exhaustive fallback;
    # first in scene order wins (scene.rs:250)
    co = scenes.light_every(scenes.compact(scenes.random_spheres(300, 5)))
    co["geom"][20:29] = (3.0, 0.1, 0.05, 0.8, 0, 0, 0, 0, 0)
    co["emission_color"][20:29] = np.linspace(0.1, 0.9, 9)[:, None]
    co["base_color"][20:29] = 0.0
    img, segs, exact = both(co, scenes.CAMERA)
    ref = oracle_render(oracle, co, 64, 36, rays_per_pixel=2, seed=42)
    assert max_abs_diff(img, ref) <= ATOL and exact > 50 * segs / 10         # the fallback did run
    shells = scenes.three_spheres()[[0] * 12].copy()
    for k in range(12):
        shells[k]["geom"][:4] = (8.0, 0.0, 0.0, 0.5 + 0.25 * k)
        shells[k]["emission_color"] = (0.1 * k, 1.0 - 0.08 * k, 0.5); shells[k]["base_color"] = 0.0
    img, segs, exact = both(shells[::-1].copy(), scenes.CAMERA)
    assert max_abs_diff(img, oracle_render(oracle, shells[::-1].copy(), 64, 36, rays_per_pixel=2, seed=42)) <= ATOL
    tiny = scenes.random_spheres(2000, 17)
    tiny["geom"][:, 3] *= 2e-5                                               # r ~ 1e-5 at |c| ~ 100: r is 1-2 ulp(f32) of the centre
    tiny["geom"][:, 0] = 10.0 + (tiny["geom"][:, 0] - 10.0) * 1e-3            # a thin slab the rays must cross
    both(tiny, scenes.CAMERA, w=128, h=72)
    both(c2, scenes.CAMERA, w=67, h=35, spp=3)                               # partial tiles (dead queue slots), odd sample count
    # from 2^20 rays per launch on RTX_KERNEL_BVH runs in two stages (primary rays; then the rays that survived their first
    # hit, from a queue whose unused reserved slots are marked dead): against one stage and the exhaustive kernel, with
    # partial tiles, and with a scratch cap that cuts the frame into several launches
    big = {}
    for name, tune, limit in (("two", 0, 0), ("two_lanes", gpu.RTX_TUNE_NO_PACKETS, 0), ("one", gpu.RTX_TUNE_ONE_STAGE, 0),
                              ("two_batched", 0, 150 << 20), ("two_walks", gpu.RTX_TUNE_NO_TILE_LISTS, 0)):
        hnd = hip_scene(gpu, c2, cam=scenes.CAMERA, kernel=gpu.RTX_KERNEL_BVH, rays_per_pixel=2, seed=42, tuning=tune).upload(0)
        hnd.set_scratch_limit(limit)
        buf = torch.zeros((997, 1203, 3), dtype=torch.float64, device="cuda:0")
        st = hnd.render_rows(1203, 997, 0, 1, 997, buf.data_ptr())
        hnd.close()
        big[name] = (buf.cpu().numpy(), st.segments, st.trace_launches, st.box_tests)
    assert big["two"][2] == 1 and big["one"][2] == 1 and big["two_batched"][2] == 2
    for name in ("two_lanes", "one", "two_batched", "two_walks"):
        assert np.array_equal(big["two"][0], big[name][0]) and big["two"][1] == big[name][1], name
    assert big["two_walks"][3] > big["two_lanes"][3]                             # (a packet tests the union of its rays' nodes)
    assert big["two"][3] < big["two_lanes"][3]                                   # (... unless it runs over its tile's list: no box tests)
    hnd = hip_scene(gpu, c2, cam=scenes.CAMERA, kernel=gpu.RTX_KERNEL_EXACT, rays_per_pixel=2, seed=42).upload(0)
    buf = torch.zeros((997, 1203, 3), dtype=torch.float64, device="cuda:0")
    ste = hnd.render_rows(1203, 997, 0, 1, 997, buf.data_ptr())
    hnd.close()
    assert np.array_equal(big["two"][0], buf.cpu().numpy()) and ste.segments == big["two"][1]
    bouncy = co.copy()
    bouncy["base_color"] = 0.97; bouncy["emission_color"] *= 0.05              # paths survive: more levels than one host-side chunk
    both(bouncy, scenes.CAMERA, max_bounces=40)
    both(bouncy, scenes.CAMERA, max_bounces=0)
    both(co, scenes.CAMERA, spp=3, scratch=300000)                             # 2560 ray slots x 96 B: one sample per batch


def test_mesh_kernel_paths(gpu, oracle):
    """trace_bvh_mesh_kernel (RTX_KERNEL_BVH_REGROUP on a tree with triangle leaves: f32-only traversal step with certain-hit
    bounds, exact tests in the f64 phase, self-hit pre-test) against the exhaustive f64 kernel bit for bit on the paths
    the benchmark meshes rarely take: more live candidates than the 6-entry queue holds (coincident triangles: the lane
    asks for a flush and resumes at the same node; first in scene order wins, scene.rs:250), origins far outside the
    scene (f64 slab walk, inline flushes) and beyond any walk, a pure (x, y)-footprint tree (the PLAIN variant) and a
    joint tree with spheres and faces solved in other planes, needle / edge-on triangles the bounds can never certify.
    RTX_KERNEL_WAVEFRONT (the same step as a kernel of its own per bounce level, exact tests + ray_hit in a second kernel,
    ray state in HBM) runs beside it in both its modes -- every level in that form (incl. more bounce levels than one
    host-side chunk of 16, and sample batches), and level 0 only with this kernel continuing from the level-1 queue; on
    the joint scenes it resolves to the regrouping kernel."""
    import torch
    from rust_raytracing_amd import scenes

    def both(objs, cam, w=64, h=36, spp=2, scratch=0, **cfg):
        out = []
        # RTX_KERNEL_WAVEFRONT twice: every level in the wavefront form / the megakernel from level 1 on (the default)
        # and with level 0 as beams (the lanes across nodes) instead of packets (one node per step)
        # (WF_PURE and BEAMS are lab bits: those two render through librtx_hip_lab.so, as does LabKernel(BVH) = round 1's lock-step kernel;
        #  RTX_KERNEL_BVH in the product library runs the mesh kernel on a tree that holds triangles)
        for kern, tune in ((gpu.RTX_KERNEL_BVH_REGROUP, 0), (gpu.RTX_KERNEL_BVH, 0), (gpu.LabKernel(gpu.RTX_KERNEL_BVH), 0),
                           (gpu.RTX_KERNEL_WAVEFRONT, gpu.RTX_TUNE_WF_PURE),
                           (gpu.RTX_KERNEL_WAVEFRONT, gpu.RTX_TUNE_BEAMS), (gpu.RTX_KERNEL_WAVEFRONT, 0), (gpu.RTX_KERNEL_EXACT, 0)):
            hnd = hip_scene(gpu, objs, cam=cam, kernel=kern, rays_per_pixel=spp, seed=42, tuning=tune, **cfg).upload(0)
            hnd.set_scratch_limit(scratch)
            buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
            st = hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
            hnd.close()
            out.append((buf.cpu().numpy(), st.segments, st.exact_tests))
        for o in out[:-1]:
            assert np.array_equal(o[0], out[-1][0]) and o[1] == out[-1][1]
        return out[0]

    mesh = scenes.light_every(scenes.compact(scenes.random_triangles(4000, 6), k=0.05, x0=5.0))
    st = gpu.debug_host_scene(gpu.Scene.from_packed(gpu.Config(), gpu.Camera(*scenes.CAMERA), mesh))
    assert st["flags"] == 2 + 4 + 8 and st["tri_other_footprints"] == 0 and st["flat_nodes"] == st["wide_nodes"]   # the PLAIN variant
    co = mesh.copy()
    co["geom"][100:114] = (3.0, -0.6, -0.5, 3.0, 0.7, -0.4, 3.1, 0.0, 0.8)        # 14 coincident triangles in front of the mesh
    co["emission_color"][100:114] = np.linspace(0.1, 0.9, 14)[:, None]
    co["base_color"][100:114] = 0.0
    img, segs, exact = both(co, scenes.CAMERA)
    two_sided(oracle, img, co, 64, 36, cam=scenes.CAMERA, rays_per_pixel=2, seed=42)    # (one path of this frame turns on a last bit)
    assert exact < 40 * segs and img.mean() > 0.01                            # flushes, not the exhaustive fallback (4000 per segment)
    for cam in (((-300.0, 2.0, 1.0), (1.0, 0.0, 0.0), 0.05),                  # outside origin_limit: the f64 slab walk
                ((-3.0e12, 0.0, 0.0), (1.0, 0.0, 0.0), 1e-11),                # beyond it: every shape, exactly
                ((5.5, 0.0, 9.0), (0.0, 0.0, -1.0), 1.0),                     # straight down the footprints' unbounded axis
                ((5.5, 0.1, 0.0), (0.2, 1.0, 0.1), 1.4)):                     # from inside the mesh
        both(co, cam, focal_offset=0.0, non_focal_offset=0.0)
    far = both(co, ((-300.0, 2.0, 1.0), (1.0, 0.0, 0.0), 0.05))               # with the default jitter as well
    bouncy = co.copy()
    bouncy["base_color"] = 0.97; bouncy["emission_color"] *= 0.05              # paths survive: up to 41 / 1 segments per ray
    both(bouncy, scenes.CAMERA, max_bounces=40)
    both(bouncy, scenes.CAMERA, max_bounces=0)
    both(co, scenes.CAMERA, spp=3, scratch=1 << 20)                            # 64 x 36 x 300 B per sample: one sample per batch
    joint = np.concatenate([scenes.compact(scenes.random_spheres(300, 3), k=0.06, x0=5.0), co[:1500], scenes.axis_aligned_mesh(60, x0=4.0, span=2.0)])
    stj = gpu.debug_host_scene(gpu.Scene.from_packed(gpu.Config(), gpu.Camera(*scenes.CAMERA), joint))
    assert stj["flags"] == 3 and stj["tri_other_footprints"] > 100
    img, segs, exact = both(joint, scenes.CAMERA)
    both(joint, ((-300.0, 2.0, 1.0), (1.0, 0.0, 0.0), 0.05))
    needles = mesh.copy()
    needles["geom"][:, 6:9] = needles["geom"][:, 0:3] + (needles["geom"][:, 3:6] - needles["geom"][:, 0:3]) * 0.5 + 1e-5   # area ~ 1e-5
    needles["geom"][::2, 2] = needles["geom"][::2, 5] = needles["geom"][::2, 8]                                             # every other one edge-on in z
    both(needles, scenes.CAMERA, w=96, h=54)


def test_wavefront_overflow_list_saturates_cleanly(gpu):
    """The wavefront form's per-level overflow list (65 536 entries) written far beyond its capacity: 14 coincident
    triangles fill the view, so every primary ray of a 512x512x8 frame (2.1e6 rays: packets) holds more live candidates
    than its 6-entry queue and flushes; the first ~8k flushes fill the list, the reservation that straddles the cap must
    leave no unwritten or stale entry behind (they are marked as nobody's), every later ray takes the exhaustive fallback.
    Twice on the same handle (the list is not cleared between launches), against the exhaustive f64 kernel bit for bit."""
    import torch
    from rust_raytracing_amd import scenes
    mesh = scenes.light_every(scenes.compact(scenes.random_triangles(1500, 6), k=0.05, x0=5.0))
    mesh["geom"][100:114] = (3.0, -9.0, -9.0, 3.3, 9.0, -9.0, 3.6, 0.0, 12.0)      # a stack of 14 that covers the 90-degree view
    mesh["emission_color"][100:114] = np.linspace(0.1, 0.9, 14)[:, None]
    mesh["base_color"][100:114] = 0.5
    w, h, spp = 512, 512, 8
    out = {}
    for kern in (gpu.RTX_KERNEL_WAVEFRONT, gpu.RTX_KERNEL_EXACT):
        hnd = hip_scene(gpu, mesh, kernel=kern, rays_per_pixel=spp, seed=7).upload(0)
        buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
        for _ in range(2 if kern == gpu.RTX_KERNEL_WAVEFRONT else 1):
            buf.zero_()
            st = hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
            assert st.kernel == kern
            out.setdefault(kern, []).append((buf.cpu().numpy(), st.segments, st.exact_tests))
        hnd.close()
    ref = out[gpu.RTX_KERNEL_EXACT][0]
    for img, segs, exact in out[gpu.RTX_KERNEL_WAVEFRONT]:
        assert np.array_equal(img, ref[0]) and segs == ref[1]
    assert out[gpu.RTX_KERNEL_WAVEFRONT][0][2] > 100 * ref[1] // 4            # the fallback did run for most segments
    assert ref[0].mean() > 0.01


def test_auto_takes_the_wavefront_form_for_a_resident_mesh(gpu, oracle):
    """AUTO on a pure (x, y)-footprint mesh with >= 2^20 rays: RTX_KERNEL_WAVEFRONT, whose level 0 walks each 8x8 tile of
    primary rays as one packet (wf_trace_packet_kernel); the regrouping megakernel continues from the level-1 queue (the
    default below 2^24 rays or for a tree beyond the L2s) or every level stays in the wavefront form (RTX_TUNE_WF_PURE).  Same
    bits as the megakernel alone, as the wavefront form with per-lane walks at level 0 (RTX_TUNE_NO_PACKETS), and the
    oracle on scattered pixels; below the ray count AUTO stays with the megakernel.  A frame that does not divide into
    tiles and a row band are included."""
    import torch
    from rust_raytracing_amd import scenes
    objs = scenes.light_every(scenes.compact(scenes.random_triangles(30000, 8), k=0.06, x0=5.0))
    w, h, spp = 517, 509, 16                                                  # 4.21e6 rays, partial tiles on both edges

    def render(kernel, rb=0, rs=1, rows=h, tuning=0):
        hnd = hip_scene(gpu, objs, kernel=kernel, rays_per_pixel=spp, seed=42, tuning=tuning).upload(0)       # (objs, spp: the enclosing scope's current values)
        buf = torch.zeros((rows, w, 3), dtype=torch.float64, device="cuda:0")
        st = hnd.render_rows(w, h, rb, rs, rows, buf.data_ptr())
        hnd.close()
        return buf.cpu().numpy(), st

    auto, st = render(gpu.RTX_KERNEL_AUTO)
    assert st.kernel == gpu.RTX_KERNEL_WAVEFRONT
    mega, stm = render(gpu.RTX_KERNEL_BVH_REGROUP)
    assert np.array_equal(auto, mega) and st.segments == stm.segments
    pure, stp = render(gpu.RTX_KERNEL_WAVEFRONT, tuning=gpu.RTX_TUNE_WF_PURE)
    assert np.array_equal(auto, pure) and stp.segments == st.segments
    bm, stb = render(gpu.RTX_KERNEL_WAVEFRONT, tuning=gpu.RTX_TUNE_BEAMS)       # level 0 as beams (wf_trace_beam_kernel)
    assert np.array_equal(auto, bm) and stb.segments == st.segments and stb.box_tests != st.box_tests
    for mode in (gpu.RTX_TUNE_WF_PURE, 0):                                     # every level in the wavefront form / the hybrid
        lanes, stl = render(gpu.RTX_KERNEL_WAVEFRONT, tuning=gpu.RTX_TUNE_NO_PACKETS | mode)
        assert np.array_equal(auto, lanes) and stl.segments == st.segments
        if mode == 0:                                                          # (same deeper levels: a packet tests the union of its rays' nodes)
            assert stl.box_tests < st.box_tests
    xs, ys = _scattered_pixels(auto, 150, 40, seed=3)
    ref = oracle.render_pixels(oracle.make_scene(objs, DEFAULT_CAM, rays_per_pixel=spp, seed=42), w, h, xs, ys)
    assert max_abs_diff(auto[ys, xs], ref) <= ATOL
    for mode in (gpu.RTX_TUNE_WF_PURE, 0):
        band, stb = render(gpu.RTX_KERNEL_WAVEFRONT, rb=3, rs=8, rows=len(range(3, h, 8)), tuning=mode)
        assert np.array_equal(band, auto[3::8])
    small, sts = render(gpu.RTX_KERNEL_AUTO, rows=h // 8)                       # 5.2e5 rays: the megakernel
    assert sts.kernel == gpu.RTX_KERNEL_BVH_REGROUP and np.array_equal(small, auto[:h // 8])
    # a joint tree (spheres + faces solved in the other planes + (x, y) footprints under one root) takes the same form: its
    # packets walk 3-D nodes, footprint nodes and sphere leaves; against the megakernel alone and the exhaustive kernel
    objs = np.concatenate([scenes.compact(scenes.random_spheres(400, 3), k=0.06, x0=5.0), objs[:6000], scenes.axis_aligned_mesh(300, x0=4.0, span=2.0)])
    stj = gpu.debug_host_scene(gpu.Scene.from_packed(gpu.Config(), gpu.Camera(*DEFAULT_CAM), objs))
    assert stj["flags"] == 3 and stj["tri_other_footprints"] > 1000
    spp = 8
    autoj, st = render(gpu.RTX_KERNEL_AUTO, rows=h // 2)                        # 1.05e6 rays
    assert st.kernel == gpu.RTX_KERNEL_WAVEFRONT
    for kern in (gpu.RTX_KERNEL_BVH_REGROUP, gpu.RTX_KERNEL_EXACT):
        other, sto = render(kern, rows=h // 2)
        assert np.array_equal(autoj, other) and sto.segments == st.segments


def test_ab_knobs_keep_the_bits(gpu):
    """Every A/B switch of RtxConfig.tuning selects another schedule, node format or tree build -- never other bits (and the
    library reads no environment variable for any of them).  Each one, on a pure mesh, a sphere scene and a joint scene, for
    the tree kernels, against the exhaustive f64 kernel: RTX_TUNE_BVH_CLASSIC (round 1's kernels), NO_QNODES (96-byte instead
    of 64-byte nodes), NO_TILES (ray queue in rows instead of 8x8 tiles: also no packets), TRI_LEAF (1 / 2 / 6 triangles per
    leaf instead of the default; larger values are clamped to 6), THRESH (regrouping threshold), BVH_MEDIAN (median splits
    instead of SAH), TWO_STAGE / ONE_STAGE / NO_PACKETS (the sphere kernel's stages)."""
    import torch
    from rust_raytracing_amd import scenes
    mesh = scenes.light_every(scenes.compact(scenes.random_triangles(3000, 16), k=0.05, x0=5.0))
    balls = scenes.light_every(scenes.compact(scenes.random_spheres(400, 15)))
    joint = np.concatenate([balls[:200], mesh[:800], scenes.axis_aligned_mesh(40, x0=4.0, span=2.0)])
    w, h, spp = 72, 40, 2

    def render(objs, kern, tuning=0):
        hnd = hip_scene(gpu, objs, cam=scenes.CAMERA, kernel=kern, rays_per_pixel=spp, seed=9, tuning=tuning).upload(0)
        buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
        st = hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
        hnd.close()
        return buf.cpu().numpy(), st.segments

    ref = {name: render(o, gpu.RTX_KERNEL_EXACT) for name, o in (("mesh", mesh), ("balls", balls), ("joint", joint))}
    leaf, thresh = gpu.RTX_TUNE_TRI_LEAF_SHIFT, gpu.RTX_TUNE_THRESH_SHIFT
    knobs = [0, gpu.RTX_TUNE_BVH_CLASSIC, gpu.RTX_TUNE_NO_QNODES, gpu.RTX_TUNE_NO_TILES, 1 << leaf, 2 << leaf, 6 << leaf, 8 << leaf,
             4 << thresh, 48 << thresh, gpu.RTX_TUNE_BVH_MEDIAN, gpu.RTX_TUNE_NO_QNODES | gpu.RTX_TUNE_WF_PURE, 6 << leaf,
             gpu.RTX_TUNE_TWO_STAGE, gpu.RTX_TUNE_TWO_STAGE | gpu.RTX_TUNE_NO_PACKETS, gpu.RTX_TUNE_TWO_STAGE | gpu.RTX_TUNE_NO_TILES,
             gpu.RTX_TUNE_ONE_STAGE | gpu.RTX_TUNE_TWO_STAGE, gpu.RTX_TUNE_TWO_STAGE | gpu.RTX_TUNE_STAGE2_PAIR,
             gpu.RTX_TUNE_TWO_STAGE | gpu.RTX_TUNE_STAGE2_POOL, gpu.RTX_TUNE_TWO_STAGE | gpu.RTX_TUNE_SORT_SURVIVORS,
             gpu.RTX_TUNE_TWO_STAGE | gpu.RTX_TUNE_STAGE2_PAIR | gpu.RTX_TUNE_NO_QNODES, gpu.RTX_TUNE_PK_LDS_STACK, gpu.RTX_TUNE_BEAMS, gpu.RTX_TUNE_BEAMS | gpu.RTX_TUNE_PK_LDS_STACK, gpu.RTX_TUNE_TWO_STAGE | gpu.RTX_TUNE_NO_CUT, gpu.RTX_TUNE_ONE_STAGE | gpu.RTX_TUNE_NO_CUT,
             gpu.RTX_TUNE_TWO_STAGE | gpu.RTX_TUNE_INLINE_LEAVES, gpu.RTX_TUNE_INLINE_LEAVES | gpu.RTX_TUNE_NO_CUT,
             gpu.RTX_TUNE_TWO_STAGE | gpu.RTX_TUNE_STAGE2_SLOTS, gpu.RTX_TUNE_STAGE2_SLOTS,
             gpu.RTX_TUNE_TWO_STAGE | gpu.RTX_TUNE_HALVES, gpu.RTX_TUNE_TWO_STAGE | gpu.RTX_TUNE_NO_HALVES,
             gpu.RTX_TUNE_TWO_STAGE | gpu.RTX_TUNE_HALVES | gpu.RTX_TUNE_NO_PACKETS, gpu.RTX_TUNE_HALVES,
             gpu.RTX_TUNE_TWO_STAGE | gpu.RTX_TUNE_NO_TILE_LISTS, gpu.RTX_TUNE_NO_TILE_LISTS]
    # a knob with a RTX_TUNE_LAB_MASK bit renders through librtx_hip_lab.so (the product library refuses it, below); the others
    # through the product library and, as LabKernel ids, through the lab library's kernel family of each id
    L = gpu.LabKernel
    for tune in knobs:
        for name, o in (("mesh", mesh), ("balls", balls), ("joint", joint)):
            for kern in (gpu.RTX_KERNEL_BVH, gpu.RTX_KERNEL_BVH_REGROUP, gpu.RTX_KERNEL_WAVEFRONT, L(gpu.RTX_KERNEL_BVH),
                         L(gpu.RTX_KERNEL_BVH_REGROUP), L(gpu.RTX_KERNEL_WAVEFRONT)):
                if (tune & gpu.abi.RTX_TUNE_LAB_MASK) and getattr(kern, "lab", False):
                    continue                                   # (the plain id already went to the lab library)
                img, segs = render(o, kern, tune)
                assert np.array_equal(img, ref[name][0]) and segs == ref[name][1], (tune, name, kern)
    # the product library refuses what it does not hold: every lab bit, and bits the header does not name
    from rust_raytracing_amd import abi
    import ctypes as C
    prod = abi.load_library(False)
    assert prod.rtx_lab_build() == 0 and abi.load_library(True).rtx_lab_build() == 1
    hnd = hip_scene(gpu, balls, cam=scenes.CAMERA, rays_per_pixel=1).upload(0)
    assert not hnd.lab
    for bit in range(32):
        tune = 1 << bit
        cfgc = gpu.Config(rays_per_pixel=1, tuning=tune).to_c()
        rc = prod.rtx_scene_set_config(hnd._h, C.byref(cfgc))
        want = abi.RTX_ERR_UNSUPPORTED if (tune & abi.RTX_TUNE_LAB_MASK) or not (tune & abi.RTX_TUNE_KNOWN_MASK) else abi.RTX_OK
        assert rc == want, (bit, rc)
        if rc:
            assert b"tuning" in prod.rtx_last_error()
    hnd.close()


def test_two_halves_in_flight_keep_the_bits(gpu, oracle):
    """RTX_TUNE_HALVES: a two-stage sphere launch as two halves of the samples on two streams with their own queues, counters and
    stack columns (render_band; an experiment kept behind its bit).  Same image and counters as the single launch (RTX_TUNE_NO_HALVES)
    and as the exhaustive kernel: even and odd sample counts, a band, a tree deep enough for the HBM stack columns, renders back to
    back on one handle and on two caller streams; the statistics count it as ONE launch and add the halves' counters up."""
    import torch
    from rust_raytracing_amd import scenes
    objs = scenes.random_spheres(10000, 1)
    deep = scenes.random_spheres(3000, 2).copy()
    deep["geom"][:, :3] *= np.logspace(0, 1.5, len(deep))[:, None]                 # sizes over 1.5 decades: a deeper tree
    w, h = 512, 288
    for name, o, spp in (("c2", objs, 8), ("c2", objs, 5), ("c2", objs, 2), ("deep", deep, 8)):
        res = {}
        for tag, tune in (("halves", gpu.RTX_TUNE_HALVES), ("one", gpu.RTX_TUNE_NO_HALVES | gpu.RTX_TUNE_NO_TILE_LISTS), ("default", 0)):
            tune |= gpu.RTX_TUNE_TWO_STAGE                                         # (two stages whatever the ray count; the halves walk: no tile lists)
            hnd = hip_scene(gpu, o, cam=scenes.CAMERA, kernel=gpu.RTX_KERNEL_AUTO, rays_per_pixel=spp, seed=42, tuning=tune).upload(0)
            buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
            st = hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
            again = torch.zeros_like(buf)
            st2 = hnd.render_rows(w, h, 0, 1, h, again.data_ptr())                     # the second set is reused, not re-created
            assert torch.equal(buf, again) and st2.segments == st.segments
            res[tag] = (buf.cpu().numpy(), st)
            hnd.close()
        a, sa = res["halves"]
        b, sb = res["one"]
        assert sa.kernel == gpu.RTX_KERNEL_BVH and sa.trace_launches == 1 and sb.trace_launches == 1 and sa.stage1_ms > 0
        assert np.array_equal(a, b) and np.array_equal(a, res["default"][0]), (name, spp)
        for f in ("segments", "exact_tests", "filter_tests", "box_tests", "stage1_box_tests", "stage1_exact_tests", "primary_rays"):
            assert getattr(sa, f) == getattr(sb, f), (name, spp, f)
        if spp == 2:
            ex = hip_render(gpu, o, w, h, cam=scenes.CAMERA, kernel=gpu.RTX_KERNEL_EXACT, rays_per_pixel=spp, seed=42)
            assert np.array_equal(a, ex)
    # a band of blocks (what one of 8 ranks renders), on a caller's stream, against the same rows of the full frame
    full = res["halves"][0]                                                          # ("deep", 8 spp)
    hnd = hip_scene(gpu, deep, cam=scenes.CAMERA, kernel=gpu.RTX_KERNEL_AUTO, rays_per_pixel=8, seed=42,
                    tuning=gpu.RTX_TUNE_HALVES | gpu.RTX_TUNE_TWO_STAGE).upload(0)
    n = int(gpu.abi.load_library(False).rtx_blocks_row_count(h, 8, 3, 8))
    band = torch.zeros((n, w, 3), dtype=torch.float64, device="cuda:0")
    s1 = torch.cuda.Stream()
    hnd.render_blocks(w, h, 8, 3, 8, band.data_ptr(), stream=s1.cuda_stream)
    s1.synchronize()
    rows = np.concatenate([np.arange(r, min(r + 8, h)) for r in range(3 * 8, h, 8 * 8)])
    assert len(rows) == n
    assert np.array_equal(band.cpu().numpy(), full[rows])
    hnd.close()


def test_tile_lists_of_the_primary_rays_keep_the_bits(gpu, oracle):
    """Stage 1 of the sphere path: what an 8x8 tile's primary rays can hit is found once per tile (build_tile_lists_kernel: a walk
    with an interval origin and an interval direction) and every packet of the tile runs the walk's leaf test over that list
    instead of walking (rtx_bvh_spheres.hip).  Same image and segment count as the packets' own walks (RTX_TUNE_NO_TILE_LISTS) and
    as the exhaustive kernel, where the list is short, where it overflows (a dense cluster: those tiles walk), where both happen in
    one frame, with the camera inside the cloud and far outside the tree's range, for partial tiles, a band of blocks, lens
    settings that degenerate the beam (focal length 0 and negative, no jitter, negative and huge offsets), planes and loose
    triangles next to the tree.  The lists must also have been USED: fewer box tests than the walks."""
    import torch
    from rust_raytracing_amd import scenes
    T = gpu.RTX_TUNE_TWO_STAGE
    rng = np.random.default_rng(5)

    def cluster(n, centre, spread, rmin, rmax, seed):
        o = scenes.random_spheres(n, seed).copy()
        u = np.random.default_rng(seed).random((n, 4))
        o["geom"][:, :3] = np.asarray(centre) + (u[:, :3] - 0.5) * spread
        o["geom"][:, 3] = rmin + (rmax - rmin) * u[:, 3]
        return o

    sparse = scenes.random_spheres(10000, 1)
    dense = cluster(3000, (14.0, 0.0, 0.0), (6.0, 16.0, 9.0), 0.4, 1.5, 8)              # ~200 spheres behind every tile: the lists overflow
    half = np.concatenate([cluster(2500, (14.0, -6.0, 0.0), (6.0, 8.0, 9.0), 0.4, 1.5, 9), sparse[:4000]])
    extras = np.concatenate([sparse[:3000], scenes.mixed_scene(1, 6, 2, seed=3)])
    cam_in = ((60.0, 3.0, -2.0), (0.7, 0.6, 0.2), 1.3)
    cam_far = ((-3.0e6, 10.0, 5.0), (1.0, 0.0, 0.0), 0.0003)
    cases = [("sparse", sparse, 512, 288, scenes.CAMERA, {}), ("partial tiles", sparse, 501, 283, scenes.CAMERA, {}),
             ("dense", dense, 256, 144, scenes.CAMERA, {}), ("half dense", half, 384, 216, scenes.CAMERA, {}),
             ("inside", sparse, 384, 216, cam_in, {}), ("far outside", sparse, 256, 144, cam_far, {}),
             ("planes and loose triangles", extras, 384, 216, scenes.CAMERA, {}),
             ("focal length 0", sparse, 256, 144, scenes.CAMERA, dict(focal_length=0.0)),
             ("focal length < 0", sparse, 256, 144, scenes.CAMERA, dict(focal_length=-4.0)),
             ("pinhole", sparse, 256, 144, scenes.CAMERA, dict(focal_offset=0.0, non_focal_offset=0.0)),
             ("negative offsets", sparse, 256, 144, scenes.CAMERA, dict(focal_offset=-0.3, non_focal_offset=-0.2)),
             ("wide aperture", sparse, 256, 144, scenes.CAMERA, dict(non_focal_offset=5.0, focal_length=30.0))]
    used = 0
    for name, objs, w, h, cam, lens in cases:
        out = {}
        for tag, kern, tune in (("lists", gpu.RTX_KERNEL_AUTO, T), ("walks", gpu.RTX_KERNEL_AUTO, T | gpu.RTX_TUNE_NO_TILE_LISTS),
                                ("exact", gpu.RTX_KERNEL_EXACT, 0)):
            spp = 2
            hnd = hip_scene(gpu, objs, cam=cam, kernel=kern, rays_per_pixel=spp, seed=11, tuning=tune, **lens).upload(0)
            buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
            st = hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
            hnd.close()
            out[tag] = (buf.cpu().numpy(), st)
        a, sa = out["lists"]
        assert sa.kernel == gpu.RTX_KERNEL_BVH and sa.stage1_ms > 0, name
        for other in ("walks", "exact"):
            b, sb = out[other]
            assert np.array_equal(a, b, equal_nan=True) and sa.segments == sb.segments, (name, other)
        assert sa.stage1_box_tests <= out["walks"][1].stage1_box_tests, name
        used += sa.stage1_box_tests < 0.5 * out["walks"][1].stage1_box_tests
        if name == "dense":
            assert sa.stage1_box_tests > 0.5 * out["walks"][1].stage1_box_tests          # (its tiles walk: the lists overflowed)
    assert used >= 6
    # several sample batches of one frame (a scratch cap): the lists are built with the first and reused by the others
    out = {}
    for tag, tune in (("lists", T), ("walks", T | gpu.RTX_TUNE_NO_TILE_LISTS)):
        hnd = hip_scene(gpu, sparse, cam=scenes.CAMERA, kernel=gpu.RTX_KERNEL_AUTO, rays_per_pixel=24, seed=5, tuning=tune).upload(0)
        hnd.set_scratch_limit(300 << 20)
        buf = torch.zeros((288, 512, 3), dtype=torch.float64, device="cuda:0")
        st = hnd.render_rows(512, 288, 0, 1, 288, buf.data_ptr())
        hnd.close()
        out[tag] = (buf.cpu().numpy(), st)
    assert out["lists"][1].trace_launches >= 2 and out["walks"][1].trace_launches >= 2          # (the lists come off the cap: 3 and 2)
    assert np.array_equal(out["lists"][0], out["walks"][0]) and out["lists"][1].segments == out["walks"][1].segments
    assert out["lists"][1].stage1_box_tests < 0.5 * out["walks"][1].stage1_box_tests
    # a band of blocks on a caller's stream
    hnd = hip_scene(gpu, sparse, cam=scenes.CAMERA, kernel=gpu.RTX_KERNEL_AUTO, rays_per_pixel=2, seed=11, tuning=T).upload(0)
    w, h = 512, 288
    full = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
    hnd.render_rows(w, h, 0, 1, h, full.data_ptr())
    n = int(gpu.abi.load_library(False).rtx_blocks_row_count(h, 8, 5, 8))
    band = torch.zeros((n, w, 3), dtype=torch.float64, device="cuda:0")
    hnd.render_blocks(w, h, 8, 5, 8, band.data_ptr())
    rows = np.concatenate([np.arange(r, min(r + 8, h)) for r in range(5 * 8, h, 8 * 8)])
    assert np.array_equal(band.cpu().numpy(), full.cpu().numpy()[rows])
    hnd.close()


def test_tile_lists_of_a_mesh_keep_the_bits(gpu):
    """Level 0 of a tree that holds triangles (RTX_KERNEL_WAVEFRONT's packets; pure (x, y)-footprint trees and joint trees with spheres and
    faces solved in the other planes): which filter records (and spheres) a tile's primary rays can pass
    is found once per tile (build_mesh_tile_lists_kernel: the footprint tree walked with the beam's (x, y) slabs, tri_filter_sign's
    inequalities at the corners of the direction box) and every packet of the tile runs its leaf code over that list.  Same image and
    segment count as the packets' own walks (RTX_TUNE_NO_TILE_LISTS) and as the exhaustive kernel -- in particular where
    Triangle::distance's |t| matters: the camera INSIDE the mesh (triangles behind the origin report phantom hits in front,
    triangle.rs:108-127), looking along each axis and down the footprints' unbounded axis, far outside, with a dense mesh whose
    lists overflow, partial tiles, a band of blocks, and lens settings that degenerate the beam."""
    import torch
    from rust_raytracing_amd import scenes
    mesh = scenes.light_every(scenes.random_triangles(60000, 2), 7)
    dense = scenes.light_every(scenes.compact(scenes.random_triangles(60000, 3), k=0.12, x0=6.0), 7)       # ~1000 records behind a tile
    # a joint tree: spheres, (x, y) footprints and faces solved in the other planes -- the list holds sphere entries and filter records
    joint = np.concatenate([scenes.random_spheres(3000, 4), scenes.light_every(scenes.random_triangles(30000, 5), 7),
                            scenes.axis_aligned_mesh(400, seed=9, span=60.0, x0=20.0)])
    cases = [("c3 recipe", mesh, 384, 216, scenes.CAMERA, {}), ("partial tiles", mesh, 381, 211, scenes.CAMERA, {}),
             ("joint", joint, 384, 216, scenes.CAMERA, {}), ("joint, inside", joint, 256, 144, ((60.0, 3.0, -2.0), (0.7, 0.6, 0.2), 1.3), {}),
             ("joint, inside, down z", joint, 256, 144, ((60.0, 0.0, 30.0), (0.0, 0.0, -1.0), 1.2), {}),
             ("joint, wide aperture", joint, 256, 144, scenes.CAMERA, dict(non_focal_offset=5.0, focal_length=30.0)),
             ("inside", mesh, 256, 144, ((60.0, 3.0, -2.0), (0.7, 0.6, 0.2), 1.3), {}),
             ("inside, along -x", mesh, 256, 144, ((60.0, 0.0, 0.0), (-1.0, 0.0, 0.0), 1.2), {}),
             ("inside, along +y", mesh, 256, 144, ((60.0, 0.0, 0.0), (0.0, 1.0, 0.0), 1.2), {}),
             ("inside, down z", mesh, 256, 144, ((60.0, 0.0, 30.0), (0.0, 0.0, -1.0), 1.2), {}),
             ("behind the mesh, looking away", mesh, 256, 144, ((130.0, 5.0, 5.0), (1.0, 0.1, 0.0), 1.0), {}),
             ("far outside", mesh, 256, 144, ((-3.0e6, 10.0, 5.0), (1.0, 0.0, 0.0), 0.0003), {}),
             ("dense", dense, 256, 144, scenes.CAMERA, {}),
             ("focal length 0", mesh, 256, 144, scenes.CAMERA, dict(focal_length=0.0)),
             ("focal length < 0", mesh, 256, 144, scenes.CAMERA, dict(focal_length=-4.0)),
             ("pinhole", mesh, 256, 144, scenes.CAMERA, dict(focal_offset=0.0, non_focal_offset=0.0)),
             ("negative offsets", mesh, 256, 144, scenes.CAMERA, dict(focal_offset=-0.3, non_focal_offset=-0.2)),
             ("wide aperture", mesh, 256, 144, scenes.CAMERA, dict(non_focal_offset=5.0, focal_length=30.0))]
    used = 0
    for name, objs, w, h, cam, lens in cases:
        out = {}
        for tag, kern, tune in (("lists", gpu.RTX_KERNEL_WAVEFRONT, 0), ("walks", gpu.RTX_KERNEL_WAVEFRONT, gpu.RTX_TUNE_NO_TILE_LISTS),
                                ("exact", gpu.RTX_KERNEL_EXACT, 0)):
            hnd = hip_scene(gpu, objs, cam=cam, kernel=kern, rays_per_pixel=2, seed=11, tuning=tune, **lens).upload(0)
            buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
            st = hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
            hnd.close()
            out[tag] = (buf.cpu().numpy(), st)
        a, sa = out["lists"]
        assert sa.kernel == gpu.RTX_KERNEL_WAVEFRONT, name
        for other in ("walks", "exact"):
            b, sb = out[other]
            assert np.array_equal(a, b, equal_nan=True) and sa.segments == sb.segments, (name, other)
        assert sa.box_tests <= out["walks"][1].box_tests, name
        used += sa.box_tests < 0.8 * out["walks"][1].box_tests
    assert used >= 9
    # several sample batches of one frame (a scratch cap): the lists are built with the first and reused by the others
    out = {}
    for tag, tune in (("lists", 0), ("walks", gpu.RTX_TUNE_NO_TILE_LISTS)):
        hnd = hip_scene(gpu, mesh, cam=scenes.CAMERA, kernel=gpu.RTX_KERNEL_WAVEFRONT, rays_per_pixel=6, seed=5, tuning=tune).upload(0)
        hnd.set_scratch_limit(96 << 20)
        buf = torch.zeros((216, 384, 3), dtype=torch.float64, device="cuda:0")
        st = hnd.render_rows(384, 216, 0, 1, 216, buf.data_ptr())
        hnd.close()
        out[tag] = (buf.cpu().numpy(), st)
    assert out["lists"][1].trace_launches >= 2 and out["walks"][1].trace_launches >= 2
    assert np.array_equal(out["lists"][0], out["walks"][0]) and out["lists"][1].segments == out["walks"][1].segments
    assert out["lists"][1].box_tests < 0.8 * out["walks"][1].box_tests
    # a band of blocks
    hnd = hip_scene(gpu, mesh, cam=scenes.CAMERA, kernel=gpu.RTX_KERNEL_WAVEFRONT, rays_per_pixel=2, seed=11).upload(0)
    w, h = 384, 216
    full = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
    hnd.render_rows(w, h, 0, 1, h, full.data_ptr())
    n = int(gpu.abi.load_library(False).rtx_blocks_row_count(h, 8, 5, 8))
    band = torch.zeros((n, w, 3), dtype=torch.float64, device="cuda:0")
    hnd.render_blocks(w, h, 8, 5, 8, band.data_ptr())
    rows = np.concatenate([np.arange(r, min(r + 8, h)) for r in range(5 * 8, h, 8 * 8)])
    assert np.array_equal(band.cpu().numpy(), full.cpu().numpy()[rows])
    hnd.close()


def test_fuzz_tile_lists_against_the_exhaustive_kernel(gpu):
    """Random scenes, cameras and lenses through the paths that use tile lists: 200 sphere clouds (two stages forced), 200 triangle
    meshes and 100 joint scenes (RTX_KERNEL_WAVEFRONT) -- sizes, densities and radii over two decades, the camera anywhere from deep
    inside to far outside looking anywhere, focal lengths of both signs, apertures from none to wider than the scene's spacing --
    against the exhaustive f64 kernel: image and segment count bit for bit."""
    import torch
    from rust_raytracing_amd import scenes
    rng = np.random.default_rng(20251005)

    def render(objs, cam, kern, tune, lens, w, h):
        hnd = hip_scene(gpu, objs, cam=cam, kernel=kern, rays_per_pixel=2, tuning=tune, **lens).upload(0)
        buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
        st = hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
        hnd.close()
        return buf.cpu().numpy(), st

    used = 0
    for it in range(500):
        kind = "spheres" if it < 200 else ("mesh" if it < 400 else "joint")
        n = int(10 ** rng.uniform(1.7, 3.6))
        k = float(10 ** rng.uniform(-1.3, 0.0))                       # how far the recipe's cloud is pulled together
        if kind == "spheres":
            objs = scenes.random_spheres(n, 100 + it)
            objs["geom"][:, 3] *= float(10 ** rng.uniform(-0.7, 0.7))
        elif kind == "mesh":
            objs = scenes.light_every(scenes.random_triangles(n, 100 + it), 5)
        else:
            objs = np.concatenate([scenes.random_spheres(max(n // 8, 6), 100 + it), scenes.light_every(scenes.random_triangles(n, 300 + it), 5),
                                   scenes.axis_aligned_mesh(max(n // 60, 2), seed=it, span=60.0, x0=20.0)])
        objs = scenes.compact(objs, k=k, x0=float(rng.uniform(2.0, 12.0)))
        centre = np.array([objs["geom"][:, 0].mean(), 0.0, 0.0])
        where = rng.random()
        pos = centre + rng.normal(size=3) * (3.0 * k * 40 if where < 0.5 else (60.0 * k * 40 if where < 0.8 else 0.3))
        if where >= 0.8 and rng.random() < 0.5:
            pos = np.zeros(3)
        look = (centre - pos) if rng.random() < 0.6 else rng.normal(size=3)
        if not np.any(look):
            look = np.array([1.0, 0.0, 0.0])
        cam = (tuple(pos), tuple(look), float(rng.uniform(0.3, 2.2)))
        lens = dict(seed=it, focal_length=float(rng.choice([10.0, 3.0, 40.0, -5.0])), focal_offset=float(rng.choice([1e-4, 0.0, 0.05])),
                    non_focal_offset=float(rng.choice([0.1, 0.0, 1.5, -0.3])), max_bounces=int(rng.choice([1, 4, 10])))
        w, h = int(rng.choice([96, 101, 160])), int(rng.choice([64, 59, 88]))
        kern, tune = (gpu.RTX_KERNEL_AUTO, gpu.RTX_TUNE_TWO_STAGE) if kind == "spheres" else (gpu.RTX_KERNEL_WAVEFRONT, 0)
        a, sa = render(objs, cam, kern, tune, lens, w, h)
        e, se = render(objs, cam, gpu.RTX_KERNEL_EXACT, 0, lens, w, h)
        assert np.array_equal(a, e, equal_nan=True) and sa.segments == se.segments, (it, kind, n, cam, lens)
        b, sb = render(objs, cam, kern, tune | gpu.RTX_TUNE_NO_TILE_LISTS, lens, w, h)
        assert np.array_equal(a, b, equal_nan=True)
        used += sa.box_tests < 0.9 * sb.box_tests
    assert used >= 200


def test_product_fallback_for_a_sphere_tree_without_64_byte_nodes(gpu):
    """The product library holds the sphere kernels in their 64-byte-node instances only.  A sphere tree whose nodes have no such
    form (coordinates beyond the quantisation's exact range: |origin / step| + 256 >= 2^24) renders with the LDS sweep under every
    kernel id, with the exhaustive kernel's bits; the lab library walks its 128-byte nodes.  (A pure footprint tree always has its
    64-byte form when the tree is built at all -- rtx_api.hip would let it walk as a joint tree otherwise.)"""
    import torch
    from rust_raytracing_amd import scenes
    balls = scenes.light_every(scenes.compact(scenes.random_spheres(300, 5)))
    balls["geom"][:, :3] += (3.0e8, 0.0, 0.0)
    cam = ((3.0e8 - 5.0, 0.0, 0.0), (1.0, 0.0, 0.0), 1.2)
    st = gpu.debug_host_scene(gpu.Scene.from_packed(gpu.Config(), gpu.Camera(*cam), balls))
    assert st["flags"] & 1 and not st["flags"] & 16 and st["wide_nodes"] > 50
    ref = hip_render(gpu, balls, 48, 32, cam=cam, kernel=gpu.RTX_KERNEL_EXACT, rays_per_pixel=2, seed=4)
    assert ref.mean() > 0.01
    L = gpu.LabKernel
    for kern, ran in ((gpu.RTX_KERNEL_AUTO, gpu.RTX_KERNEL_MIXED), (gpu.RTX_KERNEL_BVH, gpu.RTX_KERNEL_MIXED),
                      (gpu.RTX_KERNEL_BVH_REGROUP, gpu.RTX_KERNEL_MIXED), (gpu.RTX_KERNEL_WAVEFRONT, gpu.RTX_KERNEL_MIXED),
                      (L(gpu.RTX_KERNEL_BVH), gpu.RTX_KERNEL_BVH), (L(gpu.RTX_KERNEL_AUTO), gpu.RTX_KERNEL_BVH)):
        hnd = hip_scene(gpu, balls, cam=cam, kernel=kern, rays_per_pixel=2, seed=4, tuning=gpu.RTX_TUNE_TWO_STAGE).upload(0)
        buf = torch.zeros((32, 48, 3), dtype=torch.float64, device="cuda:0")
        stt = hnd.render_rows(48, 32, 0, 1, 32, buf.data_ptr())
        hnd.close()
        assert stt.kernel == ran and np.array_equal(buf.cpu().numpy(), ref), (kern, stt.kernel)


def test_bvh_joint_tree_with_out_of_range_and_axis_parallel_rays(gpu, oracle):
    """Spheres + triangles + a plane under one root; camera variants: inside the cloud looking along -z (rays nearly
    parallel to the footprints' unbounded axis), exactly axis-parallel directions (0 * inf in the slab test), and far
    outside the tree's validated origin range (those rays test every shape exactly)."""
    from rust_raytracing_amd import scenes
    objs = scenes.mixed_scene(80, 120, 1, seed=33)
    cfg = dict(rays_per_pixel=2, seed=5)
    for cam in (((8.0, 0.5, 6.0), (0.0, 0.0, -1.0), 1.2), ((0.0, 0.0, 0.0), (1.0, 0.0, 0.0), 1e-9),
                ((8.0, 0.0, 0.0), (0.0, 1.0, 0.0), 1e-9), ((-5000.0, 30.0, 10.0), (1.0, 0.0, 0.0), 0.02)):
        ref = oracle_render(oracle, objs, 40, 24, cam=cam, **cfg)
        for kern in (gpu.RTX_KERNEL_BVH, gpu.RTX_KERNEL_BVH_REGROUP, gpu.RTX_KERNEL_MIXED):
            assert max_abs_diff(hip_render(gpu, objs, 40, 24, cam=cam, kernel=kern, **cfg), ref) <= ATOL, (cam, kern)


def test_fuzz_random_scenes_all_kernels_match_oracle(gpu, oracle):
    """150 random scenes (0-40 spheres incl. zero/negative radius, 0-70 triangles of seven classes, 0-2 planes, random
    order, camera anywhere incl. inside the cloud, far outside and axis-parallel): every kernel against the oracle."""
    rng = np.random.default_rng(20251004)
    nonblack = 0
    for it in range(150):
        objs, cam = fuzz_scene(gpu, rng)
        cfg = dict(rays_per_pixel=2, seed=it, max_bounces=int(rng.choice([0, 3, 10])))
        ref = oracle_render(oracle, objs, 20, 12, cam=cam, **cfg)
        nonblack += int(np.nanmax(ref) > 0) if ref.size else 0
        for kern in (gpu.RTX_KERNEL_EXACT, gpu.RTX_KERNEL_MIXED, gpu.RTX_KERNEL_BVH, gpu.RTX_KERNEL_BVH_REGROUP, gpu.RTX_KERNEL_WAVEFRONT):
            got = hip_render(gpu, objs, 20, 12, cam=cam, kernel=kern, **cfg)
            assert max_abs_diff(got, ref) <= ATOL, (it, kern, len(objs))
    assert nonblack >= 75


def test_many_identical_spheres_first_wins(gpu, oracle):
    """24 copies of one sphere (all in one BVH leaf region, zero-extent centroid bounds) + 40 others: the copy that
    comes first in Scene.objects must win every tie (scene.rs:250), whatever order the BVH visits them in."""
    from rust_raytracing_amd import scenes
    o = scenes.compact(scenes.random_spheres(64, 9))
    o["base_color"] = 0.0
    o["emission_color"] = np.round(np.random.default_rng(2).uniform(0.1, 2, size=(64, 3)) * 16) / 16
    dup = np.arange(5, 64, 2)[:24]
    o["geom"][dup] = (5.0, 0.0, 0.0, 1.5, 0, 0, 0, 0, 0)
    cfg = dict(rays_per_pixel=2, focal_offset=0.0, non_focal_offset=0.0)
    ref = oracle_render(oracle, o, 48, 32, **cfg)
    assert ref[16, 24].tolist() == o[dup[0]]["emission_color"].tolist()
    for kern in _kernels(gpu):
        assert np.array_equal(hip_render(gpu, o, 48, 32, kernel=kern, **cfg), ref)
        assert np.array_equal(hip_render(gpu, o[::-1].copy(), 48, 32, kernel=kern, **cfg),
                              oracle_render(oracle, o[::-1].copy(), 48, 32, **cfg))


def test_candidate_queue_overflow_falls_back_to_exact_sweep(gpu, oracle):
    """More spheres along one line of sight than the per-ray LDS queue holds (8): 40 concentric-ish shells."""
    from rust_raytracing_amd import scenes
    n = 40
    o = np.zeros(n, dtype=gpu.OBJECT_DTYPE)
    o["kind"] = 0
    for i in range(n):
        o[i]["geom"][:4] = (5 + 3 * i, 0, 0, 1.0)
        o[i]["base_color"] = (0.5, 0.6, 0.7)
        o[i]["roughness"] = 0.8
    o[n - 1]["emission_color"] = (2, 2, 2)
    o[3]["emission_color"] = (1, 0.5, 0.25)
    cfg = dict(rays_per_pixel=4, seed=11)
    ref = oracle_render(oracle, o, 48, 32, **cfg)
    for kern in (gpu.RTX_KERNEL_MIXED, gpu.RTX_KERNEL_MIXED_VERIFY, gpu.RTX_KERNEL_BVH, gpu.RTX_KERNEL_BVH_REGROUP):
        assert max_abs_diff(hip_render(gpu, o, 48, 32, kernel=kern, **cfg), ref) <= ATOL


def test_partition_by_rows_is_invisible(gpu):
    """Bands rendered separately (as the ranks of a multi-GPU job do) reassemble to the one-shot image bit for bit: single
    interleaved rows (rtx_render_rows) and blocks of 8 / 5 rows (rtx_render_blocks; 8 = whole ray tiles, what the multi-GPU
    paths use), a height that does not divide, more parts than blocks; the per-part segment counts add up to the frame's."""
    import torch
    from rust_raytracing_amd import scenes, tiles
    objs = scenes.compact(scenes.random_spheres(500, 2), k=0.3)
    w, h = 80, 45
    sc = gpu.Scene.from_packed(gpu.Config(rays_per_pixel=3), gpu.Camera(*scenes.CAMERA), objs)
    full = sc.render(w, h)
    hnd = sc.upload(0)
    whole = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
    total = hnd.render_rows(w, h, 0, 1, h, whole.data_ptr()).segments
    assert np.array_equal(whole.cpu().numpy(), full)
    for world, block in ((4, 1), (4, 8), (3, 8), (8, 8), (2, 5)):
        parts, segs = [], 0
        for r in range(world):
            part = tiles.Partition(h, r, world, block)
            band = part.alloc_band(w, "cuda:0")
            if block == 1:
                rb, rs, n = tiles.rows_for_rank(h, r, world)
                assert n == part.n_rows
                st = hnd.render_rows(w, h, rb, rs, n, band.data_ptr())
            else:
                st = part.render(hnd, w, band, want_stats=True)
            assert st.primary_rays == part.n_rows * w * 3
            segs += st.segments
            parts.append(band)
        got = tiles.deinterleave(parts, h, w, block).cpu().numpy()
        assert np.array_equal(got, full), (world, block)
        assert segs == total
    hnd.close()


def test_resident_scene_config_and_camera_updates(gpu, oracle):
    """Split form of the C ABI: one upload, then rtx_scene_set_config / rtx_scene_set_camera between renders
    (change camera / settings without re-uploading the objects)."""
    import torch
    from rust_raytracing_amd import scenes
    objs = scenes.light_every(scenes.compact(scenes.random_spheres(200, 31)), 4)
    w, h = 56, 40
    cfg0 = gpu.Config(rays_per_pixel=2, seed=1)
    cam0 = gpu.Camera(*scenes.CAMERA)
    hnd = gpu.Scene.from_packed(cfg0, cam0, objs).upload(0)
    buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
    steps = [
        (cfg0, scenes.CAMERA),
        (cfg0.with_seed(9).with_rays_per_pixel(3), scenes.CAMERA),
        (cfg0.with_seed(9).with_rays_per_pixel(3), ((0.5, -1.0, 0.3), (1.0, 0.2, -0.1), 1.0)),          # moved camera, new fov
        (cfg0.with_max_bounces(1).with_kernel(gpu.RTX_KERNEL_MIXED), ((0.5, -1.0, 0.3), (1.0, 0.2, -0.1), 1.0)),
        (cfg0, scenes.CAMERA),                                                                           # and back
    ]
    for cfg, cam in steps:
        hnd.set_config(cfg)
        hnd.set_camera(gpu.Camera(*cam))
        hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
        ref = oracle_render(oracle, objs, w, h, cam=cam, rays_per_pixel=cfg.rays_per_pixel, seed=cfg.seed,
                            max_bounces=cfg.max_bounces)
        assert max_abs_diff(buf.cpu().numpy(), ref) <= ATOL
    hnd.close()


def test_resident_scene_append_objects(gpu, oracle):
    """rtx_scene_append_objects = Scene::add_object on a resident scene (row N4): start with 30 spheres (a tree), append
    triangles and a plane, then more spheres; after every step the frame equals the oracle's for the same object list.  A
    failing append (unknown kind) leaves the scene as it was."""
    import torch
    from rust_raytracing_amd import scenes
    parts = [scenes.compact(scenes.random_spheres(30, 4)), scenes.light_every(scenes.compact(scenes.random_triangles(40, 6)), 3),
             scenes.mixed_scene(0, 0, 1), scenes.compact(scenes.random_spheres(25, 8))]
    cfg = dict(rays_per_pixel=3, seed=17)
    w, h = 48, 32
    hnd = hip_scene(gpu, parts[0], **cfg).upload(0)
    buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
    have = parts[0]
    for k in range(len(parts)):
        if k:
            hnd.append_objects(parts[k])
            have = np.concatenate([have, parts[k]])
        hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
        assert max_abs_diff(buf.cpu().numpy(), oracle_render(oracle, have, w, h, **cfg)) <= ATOL, k
    bad = parts[0][:1].copy()
    bad["kind"] = 9
    with pytest.raises(gpu.RtxError):
        hnd.append_objects(bad)
    hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
    assert max_abs_diff(buf.cpu().numpy(), oracle_render(oracle, have, w, h, **cfg)) <= ATOL
    hnd.close()


def test_sample_batching_keeps_the_left_fold(gpu, oracle):
    """With scratch capped, samples are traced in several batches; the fold order (iter_ops.rs:4-8) must not change."""
    import torch
    from rust_raytracing_amd import scenes
    objs = scenes.three_spheres()
    cfg = dict(rays_per_pixel=7, seed=5)
    a = hip_render(gpu, objs, 300, 200, **cfg)
    buf = torch.zeros((200, 300, 3), dtype=torch.float64, device="cuda:0")
    # (the exhaustive kernel has no per-launch state beside the sample records: the limit is what sizes its batches; AUTO is the
    #  LDS sweep here, whose slot memory -- fixed, whatever the batch -- comes off the limit first: one sample per launch, the floor)
    for kern, launches in ((gpu.RTX_KERNEL_EXACT, 4), (gpu.RTX_KERNEL_AUTO, 7)):
        hnd = hip_scene(gpu, objs, kernel=kern, **cfg).upload(0)
        hnd.set_scratch_limit(4 << 20)                      # 300*200*32 B = 1.92 MB per sample -> batches of 2
        buf.zero_()
        st = hnd.render_rows(300, 200, 0, 1, 200, buf.data_ptr())
        hnd.close()
        assert st.trace_launches == launches, (kern, st.trace_launches)
        assert np.array_equal(a, buf.cpu().numpy())
    hnd = hip_scene(gpu, objs, **cfg).upload(0)
    hnd.set_scratch_limit(1 << 50)                          # far above the device's memory: clamped to 3/4 of what is free, not an error
    buf.zero_()
    st = hnd.render_rows(300, 200, 0, 1, 200, buf.data_ptr())
    hnd.close()
    assert st.trace_launches == 1 and np.array_equal(a, buf.cpu().numpy())
    ref = oracle_render(oracle, objs, 300, 200, **cfg)
    assert max_abs_diff(a, ref) <= ATOL


# ---- edge cases the reference's semantics define ----------------------------------------------------------
def test_edge_cases(gpu, oracle):
    from rust_raytracing_amd import scenes
    empty = np.zeros(0, dtype=gpu.OBJECT_DTYPE)
    for kern in _kernels(gpu):
        img = hip_render(gpu, empty, 9, 5, kernel=kern, rays_per_pixel=2)
        assert img.shape == (5, 9, 3) and not img.any()                            # scene.rs:224-226
        assert np.isnan(hip_render(gpu, scenes.three_spheres(), 9, 5, kernel=kern, rays_per_pixel=0)).all()  # 0/0
        one = hip_render(gpu, scenes.three_spheres(), 1, 1, kernel=kern, rays_per_pixel=2)
        assert max_abs_diff(one, oracle_render(oracle, scenes.three_spheres(), 1, 1, rays_per_pixel=2)) <= ATOL
        assert hip_render(gpu, scenes.three_spheres(), 0, 4, kernel=kern).shape == (4, 0, 3)
        # max_bounces = 0: exactly one segment per ray (scene.rs:227)
        cfg = dict(rays_per_pixel=2, max_bounces=0)
        assert max_abs_diff(hip_render(gpu, scenes.three_spheres(), 33, 17, kernel=kern, **cfg),
                            oracle_render(oracle, scenes.three_spheres(), 33, 17, **cfg)) <= ATOL
    # camera inside a sphere: near root negative -> invisible from inside (sphere.rs:29)
    o = scenes.three_spheres()[:1].copy()
    o[0]["geom"][:4] = (0, 0, 0, 3)
    assert not hip_render(gpu, o, 8, 8, rays_per_pixel=1).any()
    # non-default camera + planes-only scene
    cam = ((1.0, -2.0, 0.5), (0.3, 0.8, -0.2), 1.2)
    p = np.zeros(2, dtype=gpu.OBJECT_DTYPE)
    p["kind"] = 1
    p[0]["geom"][:6] = (0, 0, -2, 0, 0.1, 1); p[0]["base_color"] = (.8, .8, .8); p[0]["roughness"] = 0.5
    p[1]["geom"][:6] = (0, 9, 0, 0, -1, 0); p[1]["emission_color"] = (1, 1, 1); p[1]["roughness"] = 1.0
    ref = oracle_render(oracle, p, 40, 30, cam=cam, rays_per_pixel=4)
    for kern in _kernels(gpu):
        assert max_abs_diff(hip_render(gpu, p, 40, 30, cam=cam, kernel=kern, rays_per_pixel=4), ref) <= ATOL
    assert ref.mean() > 0.01


def _edge_scene(name):
    """Scenes that push the conservative f32 machinery (filters, BVH boxes) to its guards; the f64 semantics of the
    reference must survive every one of them."""
    from rust_raytracing_amd import scenes
    base = scenes.light_every(scenes.compact(scenes.random_spheres(48, 17)), 3)
    cam = DEFAULT_CAM
    cfg = dict(rays_per_pixel=2, seed=3)
    if name == "negative_and_zero_radius":           # radius enters only as r*r (sphere.rs:24): -r behaves like r, 0 never hits
        base["geom"][::5, 3] *= -1.0
        base["geom"][1::7, 3] = 0.0
    elif name == "huge_coordinates":                 # |c| ~ 1e15: f32 filter disabled (pass-all), BVH origin limit exceeded
        base["geom"][:, :3] *= 1e15 / 8.0
        base["geom"][:, 3] *= 1e15 / 8.0
        cfg.update(focal_length=1e15, non_focal_offset=1e13, focal_offset=1e11)
    elif name == "tiny_scale":                       # everything ~1e-13: filter error bound would underflow -> pass-all
        base["geom"][:, :4] *= 1e-13
        cfg.update(focal_length=1e-12, non_focal_offset=1e-15, focal_offset=1e-17)
    elif name == "nan_and_inf_shapes":               # NaN centre: never hit; inf radius: discriminant inf -> t = -inf/NaN -> filtered
        base["geom"][3, 0] = np.nan
        base["geom"][9, 3] = np.inf
        base["geom"][12, 1] = -np.inf
    elif name == "unclamped_materials":              # negative / >1 base colours and roughness outside [0,1] are used as given
        base["base_color"][::2] *= -1.5
        base["base_color"][1::4] *= 3.0
        base["roughness"][::3] = 1.7
        base["roughness"][1::3] = 0.0
    elif name == "fov_90_radians":                   # the reference's own default: Camera::new(.., 90f64) (scene.rs:90)
        cam = (DEFAULT_CAM[0], DEFAULT_CAM[1], 90.0)
    elif name == "camera_along_z":                   # direction parallel to z: right = fwd x (0,0,-1) = 0 -> NaN/zero basis (camera.rs:44)
        cam = ((0.0, 0.0, 0.0), (0.0, 0.0, 1.0), 1.2)
    elif name == "pinhole":                          # no jitter at all
        cfg.update(focal_offset=0.0, non_focal_offset=0.0)
    elif name == "zero_focal_length":
        cfg.update(focal_length=0.0)
    elif name == "deep_paths":
        base["base_color"] = 0.95
        base["emission_color"] *= 0.05
        cfg.update(max_bounces=40, rays_per_pixel=1)
    elif name == "camera_inside_a_big_sphere_and_far_plane":
        extra = np.zeros(2, dtype=base.dtype)
        extra[0]["kind"] = 0; extra[0]["geom"][:4] = (0, 0, 0, 50.0); extra[0]["emission_color"] = (1, 1, 1); extra[0]["roughness"] = 1
        extra[1]["kind"] = 1; extra[1]["geom"][:6] = (0, 0, -3.0, 0, 0, 1); extra[1]["base_color"] = (.8, .8, .8); extra[1]["roughness"] = .4
        base = np.concatenate([base, extra])
    else:
        raise KeyError(name)
    return base, cam, cfg


@pytest.mark.parametrize("name", ["negative_and_zero_radius", "huge_coordinates", "tiny_scale", "nan_and_inf_shapes",
                                  "unclamped_materials", "fov_90_radians", "camera_along_z", "pinhole",
                                  "zero_focal_length", "deep_paths", "camera_inside_a_big_sphere_and_far_plane"])
def test_edge_scenes_match_oracle(gpu, oracle, name):
    objs, cam, cfg = _edge_scene(name)
    w, h = 40, 28
    ref, seg = oracle_render(oracle, objs, w, h, cam=cam, want_segments=True, **cfg)
    for kern in (gpu.RTX_KERNEL_EXACT, gpu.RTX_KERNEL_MIXED, gpu.RTX_KERNEL_MIXED_VERIFY, gpu.RTX_KERNEL_BVH, gpu.RTX_KERNEL_BVH_REGROUP, gpu.RTX_KERNEL_AUTO):
        img = hip_render(gpu, objs, w, h, cam=cam, kernel=kern, **cfg)
        assert np.array_equal(np.isnan(img), np.isnan(ref)), (name, kern)
        scale = max(1.0, float(np.nanmax(np.abs(ref))) if np.isfinite(np.nanmax(np.abs(ref))) else 1.0)
        assert max_abs_diff(img, ref) <= ATOL * scale, (name, kern)


def test_render_is_reentrant_across_threads(gpu, oracle):
    """Scene::render(&self) may be called from several threads at once in the reference (scene.rs:144; shapes shared behind
    Arc<Mutex>, object.rs:9-15).  Six threads render different scenes / kernels through rtx_render concurrently (ctypes drops
    the GIL for the call); every frame must equal its oracle frame."""
    import threading
    from rust_raytracing_amd import scenes
    jobs = [(scenes.three_spheres(), gpu.RTX_KERNEL_AUTO), (scenes.mixed_scene(60, 50, 2, seed=21), gpu.RTX_KERNEL_BVH),
            (scenes.compact(scenes.random_spheres(500, 3), k=0.3), gpu.RTX_KERNEL_MIXED),
            (scenes.light_every(scenes.compact(scenes.random_triangles(400, 5)), 3), gpu.RTX_KERNEL_BVH_REGROUP),
            (scenes.mixed_scene(30, 80, 1, seed=5), gpu.RTX_KERNEL_EXACT), (scenes.compact(scenes.random_spheres(900, 8), k=0.3), gpu.RTX_KERNEL_AUTO)]
    refs = [oracle_render(oracle, o, 48, 30, rays_per_pixel=3, seed=9) for o, _ in jobs]
    out, errs = [None] * len(jobs), []

    def work(i):
        try:
            for _ in range(3):
                out[i] = hip_render(gpu, jobs[i][0], 48, 30, kernel=jobs[i][1], rays_per_pixel=3, seed=9)
        except Exception as e:          # noqa: BLE001
            errs.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for i, ref in enumerate(refs):
        assert max_abs_diff(out[i], ref) <= ATOL, i


def test_unsupported_and_invalid_arguments(gpu):
    from rust_raytracing_amd import scenes
    bad = scenes.three_spheres().copy()
    bad[1]["kind"] = 7
    with pytest.raises(gpu.RtxError) as e:
        hip_render(gpu, bad, 4, 4)
    assert e.value.status == gpu.abi.RTX_ERR_UNSUPPORTED
    with pytest.raises(gpu.RtxError) as e:
        hip_render(gpu, scenes.three_spheres(), 4, 4, kernel=9)
    assert e.value.status == gpu.abi.RTX_ERR_INVALID_ARGUMENT
    hnd = hip_scene(gpu, scenes.three_spheres()).upload(0)
    with pytest.raises(gpu.RtxError):
        hnd.render_rows(8, 8, 0, 1, 9, 1)              # rows beyond the image
    hnd.close()


# ---- render_to_image (scene.rs:172-178): integer output, bit-exact ---------------------------------------
def test_render_to_image_matches_oracle_quantisation(gpu, oracle):
    from rust_raytracing_amd import scenes
    objs = scenes.three_spheres()
    sc = hip_scene(gpu, objs, rays_per_pixel=4)
    f = sc.render(64, 48)
    q = sc.render_to_image(64, 48)
    assert np.array_equal(q, oracle.quantize_image(f))
    assert q.max() == 255 and q.min() == 0


# ---- the C++ host API (include/rtx.hpp) ------------------------------------------------------------------
def test_cpp_host_api_example(gpu, oracle, tmp_path):
    from rust_raytracing_amd import scenes
    exe = os.path.join(ROOT, "examples", "render_c1")
    if not os.path.exists(exe):
        pytest.fail("examples/render_c1 not built; run __graft_entry__.build()")
    f64, rgb8 = str(tmp_path / "o.f64"), str(tmp_path / "o.rgb8")
    subprocess.check_call([exe, "40", "30", "3", f64, rgb8])
    img = np.fromfile(f64, dtype=np.float64).reshape(30, 40, 3)
    ref = oracle_render(oracle, scenes.three_spheres(), 40, 30, rays_per_pixel=3, seed=42)
    assert max_abs_diff(img, ref) <= ATOL
    assert np.array_equal(np.fromfile(rgb8, dtype=np.uint8).reshape(30, 40, 3), oracle.quantize_image(img))


def test_cpp_resident_scene_camera_updates(gpu, oracle, tmp_path):
    """examples/orbit_camera.cpp: rtx::Scene::upload -> Resident::set_camera / render into device memory, four frames of
    one resident scene (row N4), each against the oracle with the same camera."""
    from rust_raytracing_amd import scenes
    exe = os.path.join(ROOT, "examples", "orbit_camera")
    if not os.path.exists(exe):
        pytest.fail("examples/orbit_camera not built; run __graft_entry__.build()")
    out = str(tmp_path / "frames.f64")
    w, h, spp, frames = 40, 30, 2, 4
    subprocess.check_call([exe, str(w), str(h), str(spp), str(frames), out])
    got = np.fromfile(out, dtype=np.float64).reshape(frames, h, w, 3)
    objs = np.zeros(4, dtype=gpu.OBJECT_DTYPE)
    objs[:3] = scenes.three_spheres()
    objs[3]["kind"] = 2
    objs[3]["geom"] = (8, -3, -1, 8, 3, -1, 8, 0, 2.5)
    objs[3]["base_color"] = (0.2, 0.6, 0.9)
    objs[3]["roughness"] = 1.0
    for k in range(frames):
        pos = (0.25 * k, -1.5 + 0.75 * k, 0.1 * k)
        cam = (pos, (6.0 - pos[0], 0.0 - pos[1], 0.5 - pos[2]), math.pi / 2)
        ref = oracle_render(oracle, objs, w, h, cam=cam, rays_per_pixel=spp, seed=42)
        assert max_abs_diff(got[k], ref) <= ATOL, k
    assert got.mean() > 0.01 and not np.array_equal(got[0], got[1])


# ---- BASELINE.json's full-size workload through size-independent properties ----------------------------------
def test_c4_shaped_band_of_one_rank(gpu, oracle):
    """BASELINE.json configs[3] is 3840x2160 over 8 GPUs: render the band rank 3 of 8 would own (270 interleaved rows) AS
    BENCHED -- the sphere kernel's two stages (packets for the primary rays, the survivors' queue, the queue-fed stage) and
    sample batching together: 4 spp with the scratch capped so that a batch holds 2 samples (2.09e6 ray slots >= 2^20) and
    two batches run -- and check three of its rows against the oracle; the one-stage form must give the same bits."""
    import torch
    from rust_raytracing_amd import scenes, tiles
    objs = scenes.random_spheres(10000, 1)
    w, h, world, rank = 3840, 2160, 8, 3
    part = tiles.Partition(h, rank, world)             # blocks of 8 rows 3, 11, 19, ...: 272 rows
    n = part.n_rows
    res = {}
    for name, tune in (("two", 0), ("one", gpu.RTX_TUNE_ONE_STAGE)):
        hnd = gpu.Scene.from_packed(gpu.Config(rays_per_pixel=4, seed=42, tuning=tune), gpu.Camera(*scenes.CAMERA), objs).upload(0)
        hnd.set_scratch_limit(400 << 20)               # 134 MB of the limit are the queue's fixed part (a chunk per resident wave of 256
                                                       # CUs); 480*34*64 slots * (32 + 64 + 5) B = 105.5 MB per sample -> 2 samples per batch
        band = part.alloc_band(w, "cuda:0")
        st = part.render(hnd, w, band, want_stats=True)
        hnd.close()
        res[name] = (band[:n].cpu().numpy(), st)
    got, st = res["two"]
    assert st.kernel == gpu.RTX_KERNEL_BVH and st.trace_launches == 2 and st.primary_rays == n * w * 4
    assert np.array_equal(got, res["one"][0]) and st.segments == res["one"][1].segments
    assert st.stage1_ms > 0 and st.box_tests < res["one"][1].box_tests      # stage 1 ran as packets over their tiles' lists: no box tests there
    osc = oracle.make_scene(objs, scenes.CAMERA, rays_per_pixel=4, seed=42)
    assert n == 272 and part.rows[0] == 24 and part.rows[8] == 88
    for k in (0, 133, 271):
        y = int(part.rows[k])
        ref = oracle.render(osc, w, h, row_begin=y, row_stride=h)       # exactly one row
        assert max_abs_diff(got[k], ref[y]) <= ATOL
    assert got.mean() > 0.01


def test_c5_shaped_rows_of_one_rank(gpu):
    """BASELINE.json configs[4] is 1M triangles at 3840x2160 over 8 GPUs.  With the footprint tree the BVH kernel keeps the
    reference's triangle semantics (no "geometric semantics" needed): rows of the band rank 5 of 8 would own,
    1 spp, AUTO (= BVH, deep tree -> HBM stack spill variant) against the exhaustive f64 kernel bit for bit (the oracle
    needs ~20 minutes per row at this triangle count; it pins the same kernels on the smaller meshes above)."""
    import torch
    from rust_raytracing_amd import scenes, tiles
    objs = scenes.random_triangles(1000000, 3, box=2.0)
    w, h, world, rank = 3840, 2160, 8, 5
    y0 = int(tiles.Partition(h, rank, world).rows[100])                        # row 100 of the band: 3 consecutive image rows of one block
    out = {}
    for kern in (gpu.RTX_KERNEL_AUTO, gpu.RTX_KERNEL_EXACT):
        hnd = gpu.Scene.from_packed(gpu.Config(rays_per_pixel=1, seed=42, kernel=kern), gpu.Camera(*scenes.CAMERA), objs).upload(0)
        buf = torch.zeros((3, w, 3), dtype=torch.float64, device="cuda:0")
        st = hnd.render_rows(w, h, y0, 1, 3, buf.data_ptr())
        out[kern] = (buf.cpu().numpy(), st.segments, st.kernel)
        hnd.close()
    a, e = out[gpu.RTX_KERNEL_AUTO], out[gpu.RTX_KERNEL_EXACT]
    assert a[2] in (gpu.RTX_KERNEL_BVH, gpu.RTX_KERNEL_BVH_REGROUP)
    assert np.array_equal(a[0], e[0]) and a[1] == e[1]
    assert a[0].mean() > 0.01


def _scattered_pixels(img, n_random, n_bright, seed):
    """Pixels to hand to the oracle: the four corners, the last column / row (the padding of partial 8x8 tiles starts
    right behind them), the brightest pixels of the frame (paths that found a light) and uniformly random ones."""
    h, w, _ = img.shape
    rng = np.random.default_rng(seed)
    xs = [0, w - 1, 0, w - 1, w - 1, w - 1, w // 2, 3]
    ys = [0, 0, h - 1, h - 1, h // 2, 5, h - 1, h - 1]
    lum = img.sum(axis=2).ravel()
    bright = np.argsort(lum)[::-1][:4 * n_bright]
    bright = bright[lum[bright] > 0][::4][:n_bright]                # every 4th of the brightest: not all from one blob
    xs += list(bright % w); ys += list(bright // w)
    xs += list(rng.integers(0, w, n_random)); ys += list(rng.integers(0, h, n_random))
    return np.asarray(xs, dtype=np.uint32), np.asarray(ys, dtype=np.uint32)


def test_c3_full_size_against_the_oracle(gpu, oracle):
    """BASELINE.json configs[2] on its own scene at its own size: 100k random triangles (scene seed 2), 1920x1080, AUTO
    (= the wavefront form: packets at level 0, the regrouping kernel from the level-1 queue).  (1) two renders are bit-identical, (2) segments within [rays, 11 rays], (3) 300+
    scattered pixels -- corners, edge columns, the brightest pixels, random ones -- equal the oracle's
    (rtxo_render_pixels: every triangle tested per segment, triangle.rs:108-127, scene.rs:243-251) within ATOL,
    (4) the lock-step BVH kernel, the regrouping kernel alone and the wavefront form at every level produce the same frame
    bit for bit."""
    import torch
    from rust_raytracing_amd import scenes
    objs = scenes.random_triangles(100000, 2)
    w, h, spp = 1920, 1080, 2
    sc = gpu.Scene.from_packed(gpu.Config(rays_per_pixel=spp, seed=42), gpu.Camera(*scenes.CAMERA), objs)
    hnd = sc.upload(0)
    buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
    st = hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
    a = buf.cpu().numpy()
    buf.zero_()
    hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
    assert np.array_equal(a, buf.cpu().numpy())
    assert st.kernel == gpu.RTX_KERNEL_WAVEFRONT
    assert st.primary_rays == w * h * spp and w * h * spp <= st.segments <= 11 * w * h * spp
    # the product library: both other tree ids run the mesh kernel alone on this tree
    for kern in (gpu.RTX_KERNEL_BVH, gpu.RTX_KERNEL_BVH_REGROUP):
        hnd.set_config(gpu.Config(rays_per_pixel=spp, seed=42, kernel=kern))
        buf.zero_()
        st2 = hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
        assert st2.kernel == gpu.RTX_KERNEL_BVH_REGROUP and np.array_equal(a, buf.cpu().numpy()) and st2.segments == st.segments
    hnd.close()
    # the lab library: round 1's lock-step kernel, and every level in the wavefront form
    hnd = sc.upload(0, lab=True)
    for kern, tune in ((gpu.RTX_KERNEL_BVH, 0), (gpu.RTX_KERNEL_WAVEFRONT, gpu.RTX_TUNE_WF_PURE)):
        hnd.set_config(gpu.Config(rays_per_pixel=spp, seed=42, kernel=kern, tuning=tune))
        buf.zero_()
        st3 = hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
        assert st3.kernel == kern and np.array_equal(a, buf.cpu().numpy()) and st3.segments == st.segments
    hnd.close()
    xs, ys = _scattered_pixels(a, 200, 100, seed=11)
    assert len(xs) >= 256
    osc = oracle.make_scene(objs, scenes.CAMERA, rays_per_pixel=spp, seed=42)
    ref = oracle.render_pixels(osc, w, h, xs, ys)
    got = a[ys, xs]
    assert max_abs_diff(got, ref) <= ATOL
    assert (ref.sum(axis=1) > 0).sum() >= 50 and a.mean() > 1e-4


def test_c5_band_against_the_oracle(gpu, oracle):
    """BASELINE.json configs[4] on its own scene: 1M random triangles (scene seed 3, box x2), 3840x2160, the band rank 5
    of 8 owns (272 rows in blocks of 8), AUTO (at 1 spp just under 2^20 rays: the regrouping BVH kernel, deep tree -> the HBM
    stack-spill variant) and the wavefront form AUTO takes from 2^20 rays on (packets at level 0 over a tree that exceeds
    the L2s, the regrouping kernel from the level-1 queue), bit for bit.  100+ scattered pixels of the band against the
    oracle, which tests all 10^6 triangles per segment."""
    import torch
    from rust_raytracing_amd import scenes, tiles
    objs = scenes.random_triangles(1000000, 3, box=2.0)
    w, h, world, rank, spp = 3840, 2160, 8, 5, 1
    part = tiles.Partition(h, rank, world)                      # blocks of 8 rows 5, 13, 21, ...
    n = part.n_rows
    hnd = gpu.Scene.from_packed(gpu.Config(rays_per_pixel=spp, seed=42), gpu.Camera(*scenes.CAMERA), objs).upload(0)
    band = part.alloc_band(w, "cuda:0")
    st = part.render(hnd, w, band, want_stats=True)
    assert st.kernel == gpu.RTX_KERNEL_BVH_REGROUP and st.primary_rays == n * w * spp
    assert n * w * spp <= st.segments <= 11 * n * w * spp
    got = band[:n].cpu().numpy()
    hnd.set_config(gpu.Config(rays_per_pixel=spp, seed=42, kernel=gpu.RTX_KERNEL_WAVEFRONT))
    band.zero_()
    stw = part.render(hnd, w, band, want_stats=True)
    hnd.close()
    assert stw.kernel == gpu.RTX_KERNEL_WAVEFRONT and stw.segments == st.segments and np.array_equal(got, band[:n].cpu().numpy())
    xs, ks = _scattered_pixels(got, 64, 40, seed=12)            # ks = row index inside the band
    ys = part.rows[ks.astype(np.int64)].astype(np.uint32)
    osc = oracle.make_scene(objs, scenes.CAMERA, rays_per_pixel=spp, seed=42)
    ref = oracle.render_pixels(osc, w, h, xs, ys)
    assert max_abs_diff(got[ks, xs], ref) <= ATOL
    assert (ref.sum(axis=1) > 0).sum() >= 20


def test_empty_scene_with_more_rays_than_resident_slots(gpu):
    """Scene::default().render(w, h): every pixel is 0 (scene.rs:224-226) -- at a size where the rays outnumber what the
    persistent kernels hold at once (the sweep kernel left early before the host short-circuit existed), on a handle
    that has already rendered something else into its scratch, for AUTO and every explicit kernel."""
    import torch
    from rust_raytracing_amd import scenes
    w, h = 1024, 600
    buf = torch.empty((h, w, 3), dtype=torch.float64, device="cuda:0")
    for kern in [gpu.RTX_KERNEL_AUTO] + _kernels(gpu):
        hnd = gpu.Scene.from_packed(gpu.Config(rays_per_pixel=1, seed=1, kernel=kern), gpu.Camera(*scenes.CAMERA),
                                    scenes.three_spheres()).upload(0)
        hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
        assert float(buf.abs().sum()) > 0
        hnd.close()
        empty = gpu.Scene.from_packed(gpu.Config(rays_per_pixel=1, seed=1, kernel=kern), gpu.Camera(*scenes.CAMERA),
                                      np.zeros(0, dtype=gpu.OBJECT_DTYPE)).upload(0)
        buf.fill_(7.0)
        st = empty.render_rows(w, h, 0, 1, h, buf.data_ptr())
        assert st.primary_rays == w * h and st.segments == 0
        assert not bool(buf.any()), kern
        buf.fill_(7.0)
        empty.render_rows(w, h, 0, 1, h, buf.data_ptr(), want_stats=False)      # the asynchronous path
        torch.cuda.synchronize()
        assert not bool(buf.any()), kern
        empty.close()
    img = gpu.Scene.from_packed(gpu.Config(rays_per_pixel=3), gpu.Camera(*scenes.CAMERA), np.zeros(0, dtype=gpu.OBJECT_DTYPE)).render(w, h)
    assert img.shape == (h, w, 3) and not img.any()


def test_render_devices_partition_is_invisible(gpu, oracle):
    """rtx_render_devices (multi-GPU behind the C ABI): the device list [0, 0] and [0, 0, 0] put two / three bands --
    separate handles, host threads and streams -- on this box's one GPU; the gathered, de-interleaved frame equals
    rtx_render's bit for bit, for f64 and for the u8 epilogue, heights that do not divide evenly, a height smaller
    than the device count, and bands large enough for the kernels AUTO takes on large renders."""
    from rust_raytracing_amd import scenes
    objs = scenes.mixed_scene(60, 50, 2, seed=21)
    sc = gpu.Scene.from_packed(gpu.Config(rays_per_pixel=3, seed=42), gpu.Camera(*scenes.CAMERA), objs)
    for (w, h) in ((67, 41), (16, 2)):
        one = sc.render(w, h)
        assert max_abs_diff(one, oracle_render(oracle, objs, w, h, rays_per_pixel=3, seed=42)) <= ATOL
        for devs in ([0], [0, 0], [0, 0, 0]):
            assert np.array_equal(sc.render(w, h, devices=devs), one), (w, h, devs)
        assert np.array_equal(sc.render_to_image(w, h, devices=[0, 0]), sc.render_to_image(w, h))
    big = gpu.Scene.from_packed(gpu.Config(rays_per_pixel=2, seed=42), gpu.Camera(*scenes.CAMERA), scenes.random_spheres(10000, 1))
    assert np.array_equal(big.render(480, 270, devices=[0, 0]), big.render(480, 270))
    # bands big enough for what AUTO takes on large renders: the two-stage sphere kernel (>= 2^20 rays per band) ...
    assert np.array_equal(big.render(1203, 997, devices=[0, 0]), big.render(1203, 997))
    # ... and the wavefront form of a mesh (packets over a band of interleaved rows, the regrouping kernel from their queue)
    mesh = gpu.Scene.from_packed(gpu.Config(rays_per_pixel=2, seed=42), gpu.Camera(*scenes.CAMERA),
                                 scenes.light_every(scenes.compact(scenes.random_triangles(30000, 8), k=0.06, x0=5.0)))
    assert np.array_equal(mesh.render(1203, 997, devices=[0, 0]), mesh.render(1203, 997))
    with pytest.raises(gpu.RtxError):
        sc.render(8, 8, devices=[])
    with pytest.raises(gpu.RtxError):
        sc.render(8, 8, devices=[0, 99])


def test_handle_orders_its_work_across_streams(gpu):
    """A handle's descriptors, tables and scratch are shared by its renders: switching streams between calls (band A on
    s1, then a camera change + band B on s2, nothing synchronised by the caller) must give the frames of two
    separate synchronous renders."""
    import torch
    from rust_raytracing_amd import scenes
    objs = scenes.random_spheres(10000, 1)
    w, h = 640, 360
    cam2 = ((0.0, 0.5, 0.2), (1.0, 0.1, 0.0), 1.3)
    ref = []
    for cam in (scenes.CAMERA, cam2):
        hnd = gpu.Scene.from_packed(gpu.Config(rays_per_pixel=2, seed=42), gpu.Camera(*cam), objs).upload(0)
        b = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
        hnd.render_rows(w, h, 0, 1, h, b.data_ptr())
        ref.append(b.cpu().numpy())
        hnd.close()
    hnd = gpu.Scene.from_packed(gpu.Config(rays_per_pixel=2, seed=42), gpu.Camera(*scenes.CAMERA), objs).upload(0)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    a = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
    b = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    hnd.render_rows(w, h, 0, 1, h, a.data_ptr(), stream=s1.cuda_stream, want_stats=False)
    hnd.set_camera(gpu.Camera(*cam2))
    hnd.render_rows(w, h, 0, 1, h, b.data_ptr(), stream=s2.cuda_stream, want_stats=False)
    torch.cuda.synchronize()
    hnd.close()
    assert np.array_equal(a.cpu().numpy(), ref[0]) and np.array_equal(b.cpu().numpy(), ref[1])


def test_c2_full_size_properties(gpu, oracle):
    """10k spheres at 1920x1080 (1 spp here; the oracle cannot do this size in seconds):
    (1) a band of rows agrees with the oracle, (2) two renders are bit-identical (determinism),
    (3) segment accounting: segments >= primary rays and <= primary * (max_bounces + 1)."""
    import torch
    from rust_raytracing_amd import scenes
    objs = scenes.random_spheres(10000, 1)
    w, h = 1920, 1080
    sc = gpu.Scene.from_packed(gpu.Config(rays_per_pixel=1, seed=42), gpu.Camera(*scenes.CAMERA), objs)
    hnd = sc.upload(0)
    buf = torch.zeros((h, w, 3), dtype=torch.float64, device="cuda:0")
    st = hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
    a = buf.cpu().numpy()
    buf.zero_()
    hnd.render_rows(w, h, 0, 1, h, buf.data_ptr())
    assert np.array_equal(a, buf.cpu().numpy())
    hnd.close()
    assert st.primary_rays == w * h and w * h <= st.segments <= 11 * w * h
    # oracle on every 216th row (5 rows x 1920 px x 10k spheres)
    osc = oracle.make_scene(objs, scenes.CAMERA, rays_per_pixel=1, seed=42)
    ref = oracle.render(osc, w, h, row_begin=100, row_stride=216)
    assert max_abs_diff(a[100::216], ref[100::216]) <= ATOL
    assert a.mean() > 0.01


def _bench(args, timeout=900, launcher=None, env=None):
    import sys
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, "bench.py")] + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and out.stdout.rstrip().splitlines()[-1] == lines[0], out.stdout[-2000:]    # ONE JSON line, and it is the last line
    # what a record that keeps only a tail of stdout can still parse (r03: a 53 KB line went unparsed)
    assert len(lines[0]) < 2000 and json.loads(out.stdout[-2000:].splitlines()[-1]) == json.loads(lines[0])
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "detail"):
        assert key in line, key
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in line["roofline"], key
    assert "workload" in line["config"]
    d = json.load(open(os.path.join(ROOT, line["detail"]) if not os.path.isabs(line["detail"]) else line["detail"]))
    # every figure of the line is a rounded copy of the full record's
    assert abs(line["value"] - d["value"]) <= 1e-5 * d["value"] and abs(line["roofline"]["frac"] - d["roofline"]["frac"]) <= 1e-3 * d["roofline"]["frac"]
    d["_line"] = line
    return d


def _check_roofline(r):
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "hbm_frac", "counters_source", "algorithmic", "avg_launch_ms"):
        assert key in r, key
    assert r["bound"] == "valu" and abs(r["peak"] - 78.6432) < 1e-3 and r["unit"] == "Tlane-op/s"
    assert 0.0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9        # a fraction of a real ceiling
    assert 0.0 < r["algorithmic"]["frac_of_valu_peak"] <= 1.0 and r["frac"] == r["algorithmic"]["frac_of_valu_peak"]   # frac = the USEFUL fraction
    if r.get("issued"):                                   # the issued fraction on the same scale: never below the useful one
        assert r["frac"] <= r["issued"]["frac"] <= 1.0
        ident = r["valu_busy"] * r["lane_utilisation"] * 2.0 / r["valu_cycles_per_instruction"]
        assert abs(ident - r["issued"]["frac"]) < 1e-9
        for k in r.get("kernels", []):
            assert k["avg_ms_per_launch"] > 0 and (k.get("issued_frac") is None or 0 < k["issued_frac"] <= 1.0)
    if r["traffic"] is not None:
        assert 0.0 < r["hbm_frac"] <= 1.0 and r["traffic"] >= r["traffic_raw"] > 0
        assert abs(r["hbm_gbs"] - r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["hbm_gbs"]


def test_bench_line_contract(gpu):
    """bench.py at a reduced sample count (counters off: they are the next test): one JSON line with the contract's keys,
    strong scaling by default, roofline objects that are fractions of a real ceiling for the value kernel, the LDS sweep
    and the three other configs, a cpu_baseline on the benchmark's own view, and the two kernels' frames bit-identical."""
    d = _bench(["--steps", "1", "--warmup", "0", "--spp", "2", "--cpu-seconds", "1", "--other-spp", "C3=1,C3band=8,C4=2,C5=1,C5band=1,J1=1", "--no-pmc",
                "--detail-out", "gpurun_out/test_bench_detail.json"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "other_configs", "lds_sweep"):
        assert key in d, key
    assert d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["steps"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["scaling"] == "strong" and "10k-sphere 1080p" in d["metric"]
    assert "workload" in d["config"] and d["config"]["rays_per_pixel"] == 2 and d["value"] > 0
    _check_roofline(d["roofline"])
    assert d["roofline"]["traffic"] is None or "profiles/pmc_counters.json" in d["roofline"]["counters_source"]
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample", "Msegments_s", "faithful_Mrays_s", "single_thread_Mrays_s", "threads_started",
                "parallel_efficiency"):
        assert key in c, key
    assert c["kind"] == "port" and c["value"] > 0 and c["single_thread_Mrays_s"] > 0
    # `cores` is the threads' worth the pool DELIVERED (pool rate / single-thread rate), not what the scheduler advertises
    assert 0.5 <= c["cores"] <= c["threads_started"] * 1.5 and abs(c["cores"] - c["value"] / c["single_thread_Mrays_s"]) < 0.06
    # the CPU leg samples the GPU's own view: same segments per primary ray up to sampling noise
    assert abs(c["segments_per_primary_ray"] - d["segments_per_primary_ray"]) < 0.35
    assert d["speedup_vs_cpu"]["primary_rays"] > 100 and d["speedup_vs_cpu"]["segments"] > 100
    assert d["lds_sweep"]["image_identical_to_value_kernel"] is True
    _check_roofline(d["lds_sweep"]["roofline"])
    assert d["speedup_vs_cpu"]["like_for_like_linear_scan"] > 10          # the LDS sweep scans the list as the CPU does
    assert [o["config"] for o in d["other_configs"]] == ["C3", "C3", "C4", "C5", "C5", "J1"]
    assert ["band" in o["workload"] for o in d["other_configs"]] == [False, True, True, False, True, False]
    # the stdout line: the contract's keys + roofline and cpu_baseline as objects, the other legs as rows of numbers
    ln = d["_line"]
    assert ln["detail"] == "gpurun_out/test_bench_detail.json"
    for key in ("value", "unit", "cores", "kind", "sample", "threads_started", "single_thread_Mrays_s"):
        assert key in ln["cpu_baseline"], key
    assert ln["cpu_baseline"]["kind"] == "port" and len(ln["roofline"]["stages"]) == 2
    assert list(ln["other_configs"]) == ["C3", "C3band", "C4band", "C5", "C5band", "J1"]
    assert all(len(row) == len(ln["other_cols"]) for row in ln["other_configs"].values())
    assert abs(ln["other_configs"]["C5"][0] - d["other_configs"][3]["value"]) <= 1e-3 * d["other_configs"][3]["value"]
    bal = d["partition_balance"]
    assert len(bal["segments_per_band"]) == 8 and 1.0 <= bal["max_over_mean"] < 1.1        # blocks of 8 rows round-robin balance the frame
    assert d["other_configs"][1]["band_rate_over_full_frame_rate"] > 0
    for o in d["other_configs"]:
        assert o["value"] > 0 and o["Msegments_per_s"] > 0 and 1.0 <= o["segments_per_primary_ray"] <= 11.0
        _check_roofline(o["roofline"])
    # (at 2 spp the frame is 4.1e6 rays: the sphere kernel's two stages) stage 1 and stage 2 reported separately and adding up
    st = d["roofline"]["stages"]
    assert len(st) == 2 and abs(st[0]["ms"] + st[1]["ms"] - d["roofline"]["avg_launch_ms"]) < 1e-6
    assert st[0]["segments"] == 1920 * 1080 * 2 and all(0 < s_["frac"] <= 1 for s_ in st)


def test_bench_collects_counters_in_the_run(gpu):
    """With rocprofv3 on the PATH the roofline's counters come from this very run (child passes under rocprofv3 --pmc):
    lane-ops issued vs the VALU ceiling, HBM-side bytes vs 8 TB/s -- both fractions below 1."""
    import shutil
    if shutil.which("rocprofv3") is None and not os.path.exists("/opt/rocm/bin/rocprofv3"):
        pytest.skip("rocprofv3 not installed")
    d = _bench(["--steps", "1", "--warmup", "0", "--spp", "2", "--no-cpu-baseline", "--no-lds-sweep", "--no-other-configs"])
    r = d["roofline"]
    assert "of this run" in r["counters_source"], (r["counters_source"], d.get("log"))
    _check_roofline(r)
    assert r["issued"]["source"].startswith("SQ_THREAD_CYCLES_VALU") and 0.0 < r["lane_utilisation"] <= 1.0
    syms = [k["kernel"] for k in r["kernels"]]
    assert "trace_sph_packet_kernel" in syms and any(k.startswith("trace_bvh_spheres_kernel<false, 2") for k in syms)      # per kernel symbol of the launch
    assert all("issued_frac" in s_ and "lane_utilisation" in s_ for s_ in r["stages"])
    assert r["traffic"] > 1920 * 1080 * 2 * 24            # at least the sample planes were written
    assert d["hbm_gbs"] == r["hbm_gbs"]


def test_bench_two_rank_rehearsal_on_one_gpu(gpu):
    """The N = 2 path of bench.py with both ranks on this box's one GPU (gloo gather through host copies, because RCCL
    refuses two ranks on one device): launch line, STRONG-scaling row partition of the same 4-spp frame, per-rank render,
    gather inside the timed region, de-interleave, max-over-ranks timing -- the assembled frame is the single-rank frame
    (same pixels, same samples), and so is the segment count."""
    import sys
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    common = ["--steps", "1", "--warmup", "0", "--spp", "4", "--no-cpu-baseline", "--no-lds-sweep", "--no-other-configs", "--no-pmc"]
    d2 = _bench(["--gpus", "2", "--rehearse-on-one-gpu"] + common, env=env,
                launcher=[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29653"])
    assert d2["n_gpus"] == 2 and d2["config"]["rays_per_pixel"] == 4 and "rehearsal" in d2 and d2["scaling"] == "strong"
    d1 = _bench(common)
    assert d1["config"]["rays_per_pixel"] == 4 and d1["metric"] == d2["metric"]
    # (the means are reduced on different devices -- host copy vs GPU tensor -- so they may differ in the last bits)
    assert abs(d1["image_mean"] - d2["image_mean"]) <= 1e-12 and d1["segments_per_primary_ray"] == d2["segments_per_primary_ray"]
    dw = _bench(["--gpus", "2", "--rehearse-on-one-gpu", "--weak"] + common[:4] + ["--spp", "2"] + common[6:], env=env,
                launcher=[sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29654"])
    assert dw["scaling"] == "weak" and dw["config"]["rays_per_pixel"] == 4 and "weak" in dw["metric"]
    assert abs(dw["image_mean"] - d1["image_mean"]) <= 1e-12
