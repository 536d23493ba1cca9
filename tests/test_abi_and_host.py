"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/rtx_hip.h
declares, its structs have the layout the bindings assume, the host logic mirrors the reference's constructors,
and the product fails loudly without a GPU (no fallback).  No compute calls are made here."""
import ctypes as C
import math
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rtx_hip.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rtx_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(rtx):
    lib = rtx.load_library()
    declared = _declared_functions()
    assert len(declared) >= 12
    bound = {name for name, _, _ in rtx.abi.SYMBOLS}
    assert set(declared) == bound                    # the Python binding covers the whole header
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.rtx_version() == 100 and lib.rtx_lab_build() == 0
    lab = rtx.load_library(lab=True)                 # the lab library: the same ABI from the same sources with -DRTX_LAB
    for name in declared:
        assert getattr(lab, name) is not None
    assert lab.rtx_version() == 100 and lab.rtx_lab_build() == 1


def test_product_library_holds_only_what_auto_can_reach():
    """include/rtx_hip.h, "Product and lab": librtx_hip.so = the kernels RTX_KERNEL_AUTO can reach + RTX_KERNEL_EXACT / MIXED and the
    epilogues -- at most 25 kernel instances (round 3's cut; rtx_debug_math's kernel went to the lab library, the tile-list builder of round 4 -- one kernel for sphere, mesh and joint trees -- is on AUTO's path),
    none of the experiments (tools/kernel_instances.py reads the code objects); the lab library holds them all.  A config check needs no GPU: the product refuses lab tuning bits, the lab library takes them."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_ki", os.path.join(ROOT, "tools", "kernel_instances.py"))
    ki = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ki)
    prod = ki.kernel_names(os.path.join(ROOT, "rust-raytracing_amd", "librtx_hip.so"))
    lab = ki.kernel_names(os.path.join(ROOT, "rust-raytracing_amd", "librtx_hip_lab.so"))
    assert 10 <= len(prod) <= 25, prod
    assert set(prod) <= set(lab) and len(lab) > len(prod) + 20
    for experiment in ("trace_sph_pool_kernel", "trace_sph_pair_kernel", "sph_sort_", "trace_bvh_spheres_pool_kernel", "trace_bvh_kernel",
                       "trace_bvh_regroup_kernel", "wf_trace_beam_kernel", "wf_trace_kernel", "wf_trace_spheres_kernel"):
        assert not any(k.startswith(experiment) for k in prod), experiment
        assert any(k.startswith(experiment) for k in lab), experiment
    for shipped in ("build_mesh_tile_lists_kernel", "trace_sph_packet_kernel", "trace_bvh_spheres_kernel<false, 2, 2>", "trace_bvh_spheres_kernel<false, 0, 2>",
                    "wf_trace_packet_kernel<1>", "trace_bvh_mesh_kernel<false, 2, true>", "trace_exact_kernel", "resolve_kernel"):
        assert shipped in prod, shipped
    # no "wrong images" timing branches in the product sources any more
    for f in os.listdir(os.path.join(ROOT, "rust-raytracing_amd", "csrc")):
        assert "RTX_SPK_LAB" not in open(os.path.join(ROOT, "rust-raytracing_amd", "csrc", f)).read(), f


def test_struct_layouts_match_the_header(rtx, tmp_path):
    prog = tmp_path / "layout.c"
    prog.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "rtx_hip.h"\n'
                    "int main(void){printf(\"%zu %zu %zu %zu %zu %zu %zu %zu\\n\", sizeof(RtxObject), sizeof(RtxConfig),"
                    " sizeof(RtxCamera), sizeof(RtxScene), sizeof(RtxStats), offsetof(RtxObject, base_color),"
                    " offsetof(RtxScene, objects), offsetof(RtxCamera, to_world_space));return 0;}\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-I" + os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    a = rtx.abi
    want = [a.OBJECT_DTYPE.itemsize, C.sizeof(a.RtxConfig), C.sizeof(a.RtxCamera), C.sizeof(a.RtxScene),
            C.sizeof(a.RtxStats), a.OBJECT_DTYPE.fields["base_color"][1], a.RtxScene.objects.offset,
            a.RtxCamera.to_world_space.offset]
    assert got == want
    assert got[0] == 136


def _c_struct_layouts(tmp_path):
    """{struct: (size, [(field, offset, size), ...])} as gcc lays out include/rtx_hip.h (field lists parsed from the header)."""
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    structs = {}
    for body, name in re.findall(r"typedef\s+struct\s+\w+\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = [re.sub(r"\[.*", "", f.strip().split()[-1]).lstrip("*") for f in body.split(";") if f.strip()]
        structs[name] = fields
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "rtx_hip.h"', 'int main(void){']
    for name, fields in structs.items():
        lines.append('printf("S %s %%zu\\n", sizeof(%s));' % (name, name))
        for f in fields:
            lines.append('printf("F %s %s %%zu %%zu\\n", offsetof(%s, %s), sizeof(((%s *)0)->%s));' % (name, f, name, f, name, f))
    lines.append('return 0;}')
    prog = tmp_path / "fields.c"
    prog.write_text("\n".join(lines) + "\n")
    exe = tmp_path / "fields"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-I" + os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    out = {}
    for ln in subprocess.check_output([str(exe)]).decode().splitlines():
        t = ln.split()
        if t[0] == "S":
            out[t[1]] = (int(t[2]), [])
        else:
            out[t[1]][1].append((t[2], int(t[3]), int(t[4])))
    return out


_RUST_SCALARS = {"u8": 1, "u32": 4, "i32": 4, "u64": 8, "i64": 8, "f64": 8, "f32": 4}


def _rust_repr_c_layouts(path):
    """#[repr(C)] structs of a Rust source file, laid out by the C rules: {struct: (size, [(field, offset, size)])}."""
    src = re.sub(r"//[^\n]*", "", open(path).read())
    out = {}

    def size_align(ty):
        ty = ty.strip()
        m = re.fullmatch(r"\[\s*(\w+)\s*;\s*(\d+)\s*\]", ty)
        if m:
            s, a = size_align(m.group(1))
            return s * int(m.group(2)), a
        if ty.startswith("*const") or ty.startswith("*mut"):
            return 8, 8
        if ty in _RUST_SCALARS:
            return _RUST_SCALARS[ty], _RUST_SCALARS[ty]
        size, fields = out[ty]                                       # a struct declared earlier in the file
        return size, max(size_align_of_fields(ty))

    aligns = {}

    def size_align_of_fields(name):
        return aligns[name]

    for name, body in re.findall(r"#\[repr\(C\)\]\s*(?:#\[[^\]]*\]\s*)*pub\s+struct\s+(\w+)\s*\{(.*?)\}", src, flags=re.S):
        off, fields, al = 0, [], [1]
        for fname, fty in re.findall(r"(?:pub\s+)?(\w+)\s*:\s*([^,]+),", body):
            s, a = size_align(fty)
            off = (off + a - 1) // a * a
            fields.append((fname, off, s))
            off += s
            al.append(a)
        amax = max(al)
        aligns[name] = al
        out[name] = ((off + amax - 1) // amax * amax, fields)
    return out


def test_enum_constants_match_the_header(rtx, tmp_path):
    """Every RTX_KERNEL_* / RTX_TUNE_* / RTX_ERR_* / RTX_OK enumerator of include/rtx_hip.h, as gcc evaluates it, against the
    Python mirror (rust-raytracing_amd/abi.py): an A/B bit that drifted would silently select another kernel in the tests."""
    import re
    import subprocess
    hdr = open(os.path.join(ROOT, "include", "rtx_hip.h")).read()
    names = sorted(set(re.findall(r"\b(RTX_(?:KERNEL|TUNE|ERR)_[A-Z0-9_]+|RTX_OK)\b\s*=", hdr)))
    assert len(names) > 25
    names += ["RTX_TUNE_LAB_MASK", "RTX_TUNE_KNOWN_MASK"]           # (macros over the enumerators)
    src = tmp_path / "enums.c"
    src.write_text('#include <stdio.h>\n#include "rtx_hip.h"\nint main(void){\n' +
                   "".join('printf("%s %%lld\\n", (long long)%s);\n' % (n, n) for n in names) + "return 0;}\n")
    exe = tmp_path / "enums"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    vals = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    from rust_raytracing_amd import abi
    missing = [n for n in names if not hasattr(abi, n)]
    assert not missing, missing
    for n in names:
        assert int(vals[n]) == int(getattr(abi, n)), n


def test_rust_shim_matches_the_header(tmp_path):
    """rust/src/raytracing/hip.rs cannot be compiled here (no rustc): its #[repr(C)] structs are parsed and laid out by
    the C rules, and must have the header's field names, order, offsets and sizes; its extern "C" block must declare
    functions of the header with the header's argument counts."""
    c = _c_struct_layouts(tmp_path)
    r = _rust_repr_c_layouts(os.path.join(ROOT, "rust", "src", "raytracing", "hip.rs"))
    for name in ("RtxObject", "RtxConfig", "RtxCamera", "RtxScene", "RtxStats"):
        assert name in r, name
        assert r[name] == c[name], name
    assert c["RtxObject"][0] == 136 and c["RtxConfig"][0] == 56 and c["RtxCamera"][0] == 200 and c["RtxScene"][0] == 272
    hdr = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    c_fns = {n: (0 if a.strip() in ("", "void") else a.count(",") + 1)
             for n, a in re.findall(r"\b(rtx_[a-z_0-9]+)\s*\(([^)]*)\)\s*;", hdr)}
    rs = re.sub(r"//[^\n]*", "", open(os.path.join(ROOT, "rust", "src", "raytracing", "hip.rs")).read())
    ext = re.search(r'extern\s+"C"\s*\{(.*?)\n\}', rs, flags=re.S).group(1)
    r_fns = {n: (0 if not a.strip() else a.count(":")) for n, a in re.findall(r"pub\s+fn\s+(rtx_\w+)\s*\(([^)]*)\)", ext)}
    assert len(r_fns) >= 14
    for n, k in r_fns.items():
        assert n in c_fns and c_fns[n] == k, n
    for needed in ("rtx_render", "rtx_render_to_image", "rtx_render_devices", "rtx_render_to_image_devices", "rtx_last_error"):
        assert needed in r_fns
    # the shim's Scene::render body goes through these and nothing else
    body = open(os.path.join(ROOT, "rust", "patches", "scene_render.rs")).read()
    assert set(re.findall(r"hip::(rtx_\w+)", body)) <= set(r_fns)
    # the shim adds no field to the reference's pub structs (users build Config / Scene with struct literals, scene.rs:16-28,
    # :78-85) and no builder that consumes self (every reference builder takes &self, scene.rs:38-53)
    assert not os.path.exists(os.path.join(ROOT, "rust", "patches", "config_seed.rs"))
    assert "self.config.seed" not in body and "self.config.devices" not in body
    for f in os.listdir(os.path.join(ROOT, "rust", "patches")):
        assert not re.search(r"pub\s+fn\s+with_\w+\s*\(\s*self\b", open(os.path.join(ROOT, "rust", "patches", f)).read()), f
    # ... and no process-wide render state: Scene::render(&self) is re-entrant across threads in the reference, so seed
    # and device list travel per call (SceneHipExt::render_with / render_seeded / render_on); the plain render() reads THREAD-local
    # defaults.  No `static mut`, no global Mutex / atomic / OnceLock / lazy_static in the shim.
    for path in [os.path.join(ROOT, "rust", "src", "raytracing", "hip.rs")] + [os.path.join(ROOT, "rust", "patches", f)
                                                                               for f in os.listdir(os.path.join(ROOT, "rust", "patches"))]:
        code = re.sub(r"//[^\n]*", "", open(path).read())
        statics = re.findall(r"^\s*(?:pub\s+)?static\s+(?:mut\s+)?\w+\s*:\s*[^=;]+", code, flags=re.M)
        outside_tls = [st for st in statics if not re.search(r"thread_local!\s*\{[^}]*" + re.escape(st.strip()), code, flags=re.S)]
        assert not outside_tls, (path, outside_tls)
        assert not re.search(r"\bstatic\s+mut\b|lazy_static!|OnceLock|OnceCell<", code), path
    assert "thread_local!" in rs and "pub trait SceneHipExt" in rs
    for fn in ("render_with", "render_seeded", "render_on"):
        assert re.search(r"fn\s+%s\s*\(\s*&self\b" % fn, rs), fn
    assert "impl SceneHipExt for Scene" in body and "hip::render_defaults()" in body and "set_render_seed" not in rs + body


def test_cpp_host_header_compiles(tmp_path):
    # include/rtx.hpp (the C++ mirror of the crate's lib.rs surface) must compile on its own
    src = tmp_path / "t.cpp"
    src.write_text('#include "rtx.hpp"\nint main(){ rtx::Scene s; s.add_object(rtx::object::Object('
                   "rtx::object::sphere::Sphere(rtx::Vector3(1,2,3), 1.0), rtx::object::Material::mirror()));"
                   " return (int)s.pack().size() - 1; }\n")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only",
                           "-I" + os.path.join(ROOT, "include"), str(src)])


def test_camera_new_matches_oracle_bitwise(rtx, oracle):
    rng = np.random.default_rng(1)
    dirs = [(1, 0, 0), (0, 1, 0), (-1, 0, 0), (0.3, -0.4, 0.5)] + [tuple(rng.normal(size=3)) for _ in range(50)]
    for d in dirs:
        pos = tuple(rng.normal(size=3))
        cam = rtx.Camera(pos, d, 1.1)
        ref = oracle.camera_new(pos, d, 1.1)
        tw = [ref.to_world_space.x, ref.to_world_space.y, ref.to_world_space.z]
        tc = [ref.to_cam_space.x, ref.to_cam_space.y, ref.to_cam_space.z]
        assert cam._to_world == [c for r in tw for c in r.tuple()]
        got, want = np.array(cam._to_cam), np.array([c for r in tc for c in r.tuple()])
        assert np.array_equal(got, want, equal_nan=True)


def test_camera_kats_through_the_host_api(rtx):                # camera.rs:82-109 via the product's Camera
    V = rtx.Vector3
    cam = rtx.Camera(V.zeros(), V(1, 0, 0), math.radians(90.0))
    assert cam.to_cam_space(V(1, 0, 0)) == V(0, 0, 1)
    assert cam.to_cam_space(V(0, 1, 0)) == V(1, 0, 0)
    assert cam.to_cam_space(V(0, 0, 1)) == V(0, 1, 0)
    assert cam.to_world_space(V(1, 0, 0)) == V(0, 1, 0)
    assert cam.to_world_space(V(0, 1, 0)) == V(0, 0, 1)
    assert cam.to_world_space(V(0, 0, 1)) == V(1, 0, 0)
    cam = rtx.Camera(V.zeros(), V(0, 1, 0), math.radians(90.0))
    assert cam.to_world_space(V(1, 0, 0)) == -V(1, 0, 0)
    assert cam.to_world_space(V(0, 1, 0)) == V(0, 0, 1)
    assert cam.to_world_space(V(0, 0, 1)) == V(0, 1, 0)
    assert cam.to_cam_space(V(1, 0, 0)) == -V(1, 0, 0)
    assert cam.to_cam_space(V(0, 1, 0)) == V(0, 0, 1)
    assert cam.to_cam_space(V(0, 0, 1)) == V(0, 1, 0)
    # set_direction keeps the reference's one-call lag (camera.rs:35-40)
    cam = rtx.Camera(V.zeros(), V(1, 0, 0), 1.0)
    before = cam.rotate_to_world_space(V(0, 0, 1))
    cam.set_direction(V(0, 1, 0))
    assert cam.get_direction() == V(0, 1, 0) and cam.rotate_to_world_space(V(0, 0, 1)) == before
    cam.set_direction(V(0, 1, 0))
    assert cam.rotate_to_world_space(V(0, 0, 1)) == V(0, 1, 0)


def test_config_material_and_packing(rtx):
    c = rtx.Config.default()
    assert (c.rays_per_pixel, c.max_bounces, c.focal_length, c.focal_offset, c.non_focal_offset) == (16, 10, 10.0, 1e-4, 1e-1)
    c2 = c.with_rays_per_pixel(32).with_max_bounces(3).with_focal_length(2.0).with_focal_offset(0.5).with_non_focal_offset(0.25)
    assert (c2.rays_per_pixel, c2.max_bounces, c2.focal_length, c2.focal_offset, c2.non_focal_offset) == (32, 3, 2.0, 0.5, 0.25)
    assert c.rays_per_pixel == 16                                        # builders copy (scene.rs:29-37)
    M, V = rtx.Material, rtx.Vector3
    assert M.colored((1, 1, 0)).roughness == 1.0 and M.colored((1, 1, 0)).emission_color == V.zeros()
    assert M.light((1, .8, .5)).base_color == V.zeros() and M.light((1, .8, .5)).roughness == 1.0
    assert M.mirror().roughness == 1.0 and M.mirror().base_color == V.ones()     # object.rs:133-135: 1.0 on the CPU path
    sc = rtx.Scene(rtx.Config(), rtx.Camera((-1, 0, 0), (1, 0, 0), 90.0))     # the doc example, scene.rs:110
    sc.add_object(rtx.Object(rtx.Sphere((1, 2, 3), 4), M.colored((.1, .2, .3))))
    sc.add_object(rtx.Object(rtx.Triangle([(0, 0, 0), (1, 0, 0), (0, 1, 0)]), M.light((1, 1, 1))))
    sc.add_object(rtx.Object(rtx.Plane((0, 0, -1), (0, 0, 1)), M.mirror()))
    p = sc.packed()
    assert p["kind"].tolist() == [0, 2, 1]
    assert p[0]["geom"][:4].tolist() == [1, 2, 3, 4] and p[0]["base_color"].tolist() == [.1, .2, .3]
    assert p[1]["geom"].tolist() == [0, 0, 0, 1, 0, 0, 0, 1, 0] and p[1]["emission_color"].tolist() == [1, 1, 1]
    assert p[2]["geom"][:6].tolist() == [0, 0, -1, 0, 0, 1]

    class Custom:            # a user CustomShape (object.rs:53-76): no device primitive, no fallback
        pass
    with pytest.raises(rtx.RtxError):
        rtx.Object(Custom(), M.mirror())


def test_render_fails_loudly_without_a_gpu(rtx):
    if rtx.device_count() > 0:
        pytest.skip("a gfx950 device is present")
    from rust_raytracing_amd import scenes
    sc = rtx.Scene.from_packed(rtx.Config(rays_per_pixel=1), rtx.Camera(*scenes.CAMERA), scenes.three_spheres())
    with pytest.raises(rtx.RtxError) as e:
        sc.render(4, 4)
    assert e.value.status == rtx.abi.RTX_ERR_NO_DEVICE
    with pytest.raises(rtx.RtxError):
        sc.upload(0)


def test_product_never_touches_the_oracle():
    # the product path must not import, link or execute anything under oracle/
    pkg = os.path.join(ROOT, "rust-raytracing_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp", ".hpp")):
                text = open(os.path.join(base, f), errors="replace").read()
                assert "rtx_oracle" not in text and "rtxo_" not in text, f
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
    for f in ("include/rtx_hip.h", "include/rtx.hpp", "rust_raytracing_amd.py"):
        assert "rtxo_" not in open(os.path.join(ROOT, f)).read()
    out = subprocess.check_output(["ldd", os.path.join(pkg, "librtx_hip.so")]).decode()
    assert "oracle" not in out


def test_scene_generators_are_deterministic():
    import hashlib
    from rust_raytracing_amd import scenes
    a, b = scenes.random_spheres(10000, 1), scenes.random_spheres(10000, 1)
    assert a.tobytes() == b.tobytes()
    g = a["geom"]
    assert g[:, 0].min() >= 10 and g[:, 0].max() < 110 and np.abs(g[:, 1:3]).max() <= 50
    assert g[:, 3].min() >= 0.2 and g[:, 3].max() < 1.0
    lights = (a["emission_color"] > 0).any(axis=1)
    assert 0.03 < lights.mean() < 0.07 and not a["base_color"][lights].any()
    t = scenes.random_triangles(1000, 2)
    assert t.tobytes() == scenes.random_triangles(1000, 2).tobytes()
    # the arrays BASELINE.json's configs are built from, pinned by hash
    assert hashlib.sha256(a.tobytes()).hexdigest()[:16] == open(os.path.join(ROOT, "tests", "golden", "scene_hashes.txt")).read().split()[1]


# ---- host half of the upload: scene packing + SAH build of the flat BVH, checked without a GPU ----
def _host_scene(rtx, objs):
    from rust_raytracing_amd import scenes
    return rtx.debug_host_scene(rtx.Scene.from_packed(rtx.Config(), rtx.Camera(*scenes.CAMERA), objs))


def test_bvh_invariants_on_the_benchmark_scenes(rtx):
    """rtx_debug_host_scene builds what rtx_scene_upload builds and walks the tree: child boxes inside their parents', every
    sphere / footprint inside its leaf's box, every shape in exactly one leaf, links, layout flags and depth consistent."""
    from rust_raytracing_amd import scenes
    st = _host_scene(rtx, scenes.random_spheres(10000, 1))                       # C2
    assert st["flags"] == 1 + 16 and st["quantised_nodes"] == st["wide_nodes"] and st["sphere_leaf_entries"] == 10000 and st["largest_leaf"] == 1 and st["flat_nodes"] == 0
    assert st["depth"] <= 9 and st["stack_bound"] <= 30                          # C2 runs the LDS-stack-only kernel variant
    st = _host_scene(rtx, scenes.random_triangles(100000, 2))                    # C3
    assert st["flags"] == 2 + 4 + 8 and st["quantised_nodes"] == st["wide_nodes"] and st["tri_in_tree"] == st["tri_leaf_entries"] == st["tri_filter_records"]
    assert 45000 < st["tri_in_tree"] < 56000                                     # about half are culled for every direction
    assert st["flat_nodes"] == st["wide_nodes"] and st["largest_leaf"] == 5                     # (the 64-byte nodes hold leaves of up to 6 records)
    st = _host_scene(rtx, scenes.mixed_scene(60, 50, 2, seed=21))                # joint root: spheres + triangles
    assert st["flags"] == 3 and st["sphere_leaf_entries"] == 60 and 0 < st["flat_nodes"] < st["wide_nodes"]
    st = _host_scene(rtx, scenes.three_spheres())                                # C1: too small for a tree
    assert st["wide_nodes"] == 0 and st["flags"] == 0
    # faces in planes x / y / z = const: Triangle::contains swaps pivot rows (triangle.rs:60-71,81-87) and solves them in
    # (y, z) / (x, z); they enter the tree with the footprint in THAT plane instead of being tested for every segment
    st = _host_scene(rtx, scenes.axis_aligned_mesh())
    assert st["tri_in_tree"] == st["tri_filter_records"] == st["tri_leaf_entries"] > 5000
    assert st["tri_xy_footprints"] > 1500 and st["tri_other_footprints"] > 3000
    assert st["tri_xy_footprints"] + st["tri_other_footprints"] == st["tri_in_tree"] and 0 < st["flat_nodes"] < st["wide_nodes"]
    st = _host_scene(rtx, np.concatenate([scenes.random_spheres(3000, 1), scenes.axis_aligned_mesh(), scenes.random_triangles(5000, 2)]))
    assert st["flags"] == 3 and st["sphere_leaf_entries"] == 3000 and st["tri_in_tree"] == st["tri_filter_records"]
    assert st["tri_other_footprints"] > 3000


def test_bvh_invariants_on_awkward_and_random_scenes(rtx):
    from helpers import fuzz_scene
    from rust_raytracing_amd import scenes
    o = scenes.compact(scenes.random_spheres(64, 9))
    o["geom"][5:60:2] = (5.0, 0.0, 0.0, 1.5, 0, 0, 0, 0, 0)                      # many identical spheres: zero-extent centroid bounds
    assert _host_scene(rtx, o)["sphere_leaf_entries"] == 64
    o = scenes.random_spheres(100, 3)
    o[7]["geom"][0] = float("nan")                                                # a non-finite sphere: no sphere sub-tree
    assert _host_scene(rtx, o)["flags"] == 0
    o = scenes.random_triangles(300, 4)
    o["geom"] *= 1.0e29                                                           # beyond the f32 slab test's range: no tree at all
    assert _host_scene(rtx, o)["wide_nodes"] == 0
    o = scenes.random_triangles(64, 5)
    o["geom"][:, 3:6] = o["geom"][:, 0:3]                                         # all degenerate: nothing can be hit, nothing in the tree
    assert _host_scene(rtx, o)["tri_filter_records"] == 0
    rng = np.random.default_rng(7)
    built = 0
    for _ in range(200):
        objs, _cam = fuzz_scene(rtx, rng)
        st = _host_scene(rtx, objs)                                               # raises on any violated invariant
        built += st["wide_nodes"] > 0
        assert st["tri_leaf_entries"] == st["tri_in_tree"] or not (st["flags"] & 2)
    assert built > 100
    deep = _host_scene(rtx, scenes.random_triangles(600000, 2))                   # deep enough for the HBM stack spill variant
    assert deep["stack_bound"] > 30 and deep["tri_leaf_entries"] == deep["tri_in_tree"]



def test_bench_roofline_object_is_a_fraction_of_a_real_ceiling(tmp_path, monkeypatch):
    """bench.py's roofline arithmetic on synthetic counts (no GPU): ONE scale (lane-instructions against the VALU peak) with
    frac = the USEFUL fraction (the tests the kernel counted x LANE_OPS) <= issued.frac (SQ_THREAD_CYCLES_VALU), the identity
    issued.frac = valu_busy x lane_utilisation x 2 / valu_cycles_per_instruction, hbm_frac = (2 x FETCH + WRITE) / launch
    time / 8 TB/s, the two stages of the sphere kernel reported separately and adding up; the LDS sweep's packed filter
    priced so that useful <= issued there too; a counter file is refused when it was measured on other kernel sources."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("_bench", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    acc = b.Acc()

    class St:                                   # round 2's C2 launch, split as round 3's stats report it
        trace_ms = 81.3; segments = 280452564; filter_tests = int(280452564 * 95.05); exact_tests = int(280452564 * 0.566)
        box_tests = int(280452564 * 93.45); trace_launches = 1; kernel = 4; primary_rays = 1920 * 1080 * 64
        stage1_ms = 25.3; stage1_box_tests = 132710400 * 100; stage1_filter_tests = 132710400 * 101; stage1_exact_tests = 132710400 // 2
    acc.add(St)
    counters = {"SQ_THREAD_CYCLES_VALU": 1.3347e12, "SQ_ACTIVE_INST_VALU": 4.0188e10, "SQ_INSTS_VALU": 3.9665e10,
                "SQ_INSTS_SALU": 1.0e10, "SQ_WAVE_CYCLES": 1.9577e11, "SQ_WAIT_ANY": 1.0707e11,
                "FETCH_SIZE": 16316652.5, "WRITE_SIZE": 14350844.6}
    per = {"trace_sph_packet_kernel": {"SQ_THREAD_CYCLES_VALU": 4.5e11, "SQ_ACTIVE_INST_VALU": 8.5e9, "SQ_INSTS_VALU": 8.0e9, "_ms": 25.0, "_calls": 1.0},
           "trace_bvh_spheres_kernel<false, 2>": {"SQ_THREAD_CYCLES_VALU": 6.3e11, "SQ_ACTIVE_INST_VALU": 2.566e10, "SQ_INSTS_VALU": 2.535e10, "_ms": 56.0, "_calls": 1.0}}
    r = b.roofline_of(acc, b.CONFIGS["C2"], counters, "test", per)
    assert r["bound"] == "valu" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["frac"] == r["algorithmic"]["frac_of_valu_peak"] and 0.05 < r["frac"] < r["issued"]["frac"] < 0.3     # useful <= issued
    assert abs(r["lane_utilisation"] - 0.526) < 0.01 and 0.5 < r["valu_busy"] < 1.0
    ident = r["valu_busy"] * r["lane_utilisation"] * 2.0 / r["valu_cycles_per_instruction"]
    assert abs(ident - r["issued"]["frac"]) < 1e-9                                  # the stated identity holds
    assert abs(r["traffic"] - (2 * 16316652.5 + 14350844.6) * 1024) < 1 and 0.05 < r["hbm_frac"] < 0.1
    assert abs(r["algorithmic"]["bytes_per_segment"] - (93.45 * 32 + 1.6 * 16 + 0.566 * 32)) < 1.0
    st = r["stages"]
    assert len(st) == 2 and abs(st[0]["ms"] + st[1]["ms"] - 81.3) < 1e-9 and abs(st[0]["segments"] + st[1]["segments"] - 280452564) < 1
    assert st[0]["kernel"] == "trace_sph_packet_kernel" and st[1]["kernel"].endswith("<false, 2>")
    assert all(0 < s_["frac"] < s_["issued_frac"] for s_ in st)
    assert [k["kernel"] for k in r["kernels"]] == ["trace_bvh_spheres_kernel<false, 2>", "trace_sph_packet_kernel"]
    r0 = b.roofline_of(acc, b.CONFIGS["C2"], None, "none")
    assert r0["traffic"] is None and r0["hbm_frac"] is None and r0["issued"] is None and 0 < r0["frac"] <= 1 and "useful" in r0["frac_source"]
    # the LDS sweep: 10^4 packed filter evaluations per segment priced at 4 lane-instructions -> useful below issued
    sweep = b.Acc()

    class Sw:
        trace_ms = 440.0; segments = 280452564; filter_tests = 280452564 * 10124; exact_tests = int(280452564 * 1.9)
        box_tests = 0; trace_launches = 1; kernel = 2; primary_rays = 1920 * 1080 * 64
        stage1_ms = 0.0; stage1_box_tests = 0; stage1_filter_tests = 0; stage1_exact_tests = 0
    sweep.add(Sw)
    rs = b.roofline_of(sweep, b.CONFIGS["C2"], {"SQ_THREAD_CYCLES_VALU": 0.383 * 78.6432e12 * 0.44, "SQ_ACTIVE_INST_VALU": 0.383 * 78.6432e12 * 0.44 / 62.0,
                                                "SQ_INSTS_VALU": 0.383 * 78.6432e12 * 0.44 / 62.0}, "test")
    assert 0.2 < rs["frac"] < rs["issued"]["frac"] < 0.5
    # stored counters: refused for other sources, accepted for this tree
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    os.makedirs(tmp_path / "profiles")
    os.makedirs(tmp_path / "rust-raytracing_amd" / "csrc")
    (tmp_path / "rust-raytracing_amd" / "csrc" / "k.hip").write_text("kernel v1")
    json.dump({"kernel_source_hash": b.kernel_source_hash(), "collected": "t", "legs": {"C2:64:full:0": counters},
               "legs_per_kernel": {"C2:64:full:0": per}},
              open(tmp_path / "profiles" / "pmc_counters.json", "w"))
    got, gper, src = b.stored_counters("C2:64:full:0")
    assert got == counters and gper == per and "pmc_counters.json" in src
    (tmp_path / "rust-raytracing_amd" / "csrc" / "k.hip").write_text("kernel v2")
    got, gper, src = b.stored_counters("C2:64:full:0")
    assert got is None and "other kernel sources" in src
    # the CPU baseline states the parallelism it may use: min(affinity, cgroup quota)
    n, how, quota = b.cpu_threads_available()
    assert n >= 1 and "sched_getaffinity" in how
    # the ONE stdout line: compact (< 2000 characters whatever the record holds), every contract key, roofline + cpu_baseline objects
    other = [{"config": c, "leg": leg, "rays_per_pixel": 64, "value": 456.789, "ms_per_step": 36.3, "roofline": r, "workload": "x" * 500,
              "band_rate_over_full_frame_rate": 0.99} for c, leg in (("C3", "C3"), ("C3", "C3band"), ("C4", "C4band"), ("C5", "C5"), ("C5", "C5band"),
                                                                        ("J1", "J1"))]
    full = {"metric": "Mrays/s (primary rays, whole node), 10k-sphere 1080p 64spp", "value": 2614.3456789, "unit": "Mrays/s", "n_gpus": 1, "steps": 20,
            "warmup": 5, "ms_per_step": 50.7623456, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic", "config": {"workload": "w" * 300, "workload_short": "C2: 10k random spheres (scene seed 1), 1920x1080, 64 spp",
                                            "partition_short": "single GPU", "width": 1920, "height": 1080, "rays_per_pixel": 64, "n_objects": 10000},
            "segments_per_primary_ray": 2.11326741, "image_mean": 0.118618046039477, "roofline": r, "other_configs": other,
            "lds_sweep": {"value": 304.6}, "partition_balance": {"max_over_mean": 1.0134},
            "cpu_baseline": {"value": 0.0669438, "unit": "Mrays/s", "cores": 15.4, "kind": "port", "sample": "s" * 400,
                             "sample_short": "every 2th row+column of the C2 1080p frame (518400 px, 1 spp), C port of the CPU path, 16 threads, 27 s",
                             "threads_started": 16, "single_thread_Mrays_s": 0.00435632, "faithful_Mrays_s": 0.0068229, "seconds": 26.9},
            "speedup_vs_cpu": {"primary_rays": 39052.1, "like_for_like_linear_scan": 4550.2}, "log": ["l" * 1000] * 20}
    text = b.compact_line(full, "gpurun_out/bench_detail_n1.json")
    ln = json.loads(text)
    assert len(text) < 2000 and "\n" not in text
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline", "other_configs", "detail"):
        assert key in ln, key
    assert ln["roofline"]["bound"] == "valu" and abs(ln["roofline"]["frac"] - r["frac"]) < 1e-3 * r["frac"] and len(ln["roofline"]["stages"]) == 2
    assert abs(ln["roofline"]["traffic"] - r["traffic"]) < 1e-3 * r["traffic"] and ln["cpu_baseline"]["kind"] == "port"
    assert list(ln["other_configs"]) == ["C3", "C3band", "C4band", "C5", "C5band", "J1"] and ln["value"] == 2614.35
    # a record that would not fit loses its extras, never the contract keys
    full["cpu_baseline"]["sample_short"] = "s" * 700
    ln2 = json.loads(b.compact_line(full, "d.json"))
    assert "other_configs" not in ln2 and "roofline" in ln2 and "cpu_baseline" in ln2
    # the full record goes to a file
    path = b.write_detail(full, str(tmp_path / "sub" / "detail.json"), 1)
    assert path == os.path.join("sub", "detail.json")               # relative to the repo root (monkeypatched above) when inside it
    assert json.load(open(tmp_path / path))["log"] == full["log"]
