"""In-tree build of librtx_hip.so -- the product -- and librtx_hip_lab.so (hipcc, gfx950 only).

hipcc cross-compiles without a GPU.  The .so files are git-ignored but travel to the GPU box
with the gpurun snapshot, so nothing is JIT-compiled there.

Two libraries from the same sources (include/rtx_hip.h, "Product and lab"):
  librtx_hip.so       what RTX_KERNEL_AUTO can reach + RTX_KERNEL_EXACT / MIXED: the kernels that ship
  librtx_hip_lab.so   -DRTX_LAB: + every experiment behind a RTX_TUNE_LAB_MASK bit and the older kernel families
                      (rtx_bvh.hip, rtx_bvh_regroup.hip, rtx_bvh_spheres_pool.hip, rtx_wavefront_spheres.hip, rtx_bvh_spheres_lab.h);
                      loaded by the lab tests and the A/B tools, never by the product path

Every .hip file is its own translation unit (no cross-file device calls), so the objects are
compiled in parallel and cached by content hash of (source, headers, flags); the link is one
`hipcc -shared`, redone whenever the list of object digests differs from the one the library was linked from.
"""
import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librtx_hip.so")
LAB_LIB = os.path.join(HERE, "librtx_hip_lab.so")
OBJ_DIR = os.path.join(HERE, "build")
SOURCES = ["rtx_kernels.hip", "rtx_bvh_spheres.hip", "rtx_bvh_mesh.hip", "rtx_wavefront.hip", "rtx_api.hip"]
LAB_SOURCES = SOURCES + ["rtx_bvh.hip", "rtx_bvh_spheres_pool.hip", "rtx_bvh_regroup.hip", "rtx_wavefront_spheres.hip"]
HEADERS = ["rtx_math.h", "rtx_scene.h", "rtx_bvh.h", "rtx_device.h", "rtx_traverse.h", "rtx_mesh_step.h", "rtx_wavefront.h",
           "rtx_launch.h", "rtx_bvh_spheres_lab.h"]
# -ffp-contract=off: the exact path must round like the reference (Rust never fuses a*b+c);
# the f32 filter asks for FMAs explicitly.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]
LAB_FLAGS = ["-DRTX_LAB"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: librtx_hip.so cannot be built (there is no CPU fallback)")
    return exe


def _input_files():
    files = [os.path.join(CSRC, f) for f in HEADERS]
    files.append(os.path.join(HERE, "..", "include", "rtx_hip.h"))
    return files


def _digest(source, flags):
    h = hashlib.sha256()
    h.update(" ".join(flags).encode())
    for f in [os.path.join(CSRC, source)] + _input_files():
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:20]


def build(force=False, extra_flags=(), verbose=False, lib=None, jobs=None, lab=False):
    """Compile rust-raytracing_amd/librtx_hip.so (lab=True: librtx_hip_lab.so; or `lib`); returns its path."""
    if lib is None:
        lib = LAB_LIB if lab else LIB
    flags = FLAGS + (LAB_FLAGS if lab else []) + list(extra_flags)
    sources = LAB_SOURCES if lab else SOURCES
    os.makedirs(OBJ_DIR, exist_ok=True)
    exe = hipcc()
    objs, todo = [], []
    for s in sources:
        obj = os.path.join(OBJ_DIR, "%s.%s.o" % (s[:-4], _digest(s, flags)))
        objs.append(obj)
        if force or not os.path.exists(obj):
            todo.append((s, obj))

    def compile_one(job):
        s, obj = job
        cmd = [exe] + flags + ["-c", os.path.join(CSRC, s), "-o", obj + ".tmp"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        os.replace(obj + ".tmp", obj)

    if todo:
        with ThreadPoolExecutor(max_workers=jobs or min(len(todo), max(1, (os.cpu_count() or 2) - 1))) as ex:
            list(ex.map(compile_one, todo))
    # the link: whenever the library is absent or was linked from another set of objects (an interrupted link, a variant
    # name reused after its flags changed) -- the ordered digests are kept next to it
    stamp = lib + ".objs"
    want = "\n".join(os.path.basename(o) for o in objs) + "\n"
    have = open(stamp).read() if os.path.exists(stamp) else None
    if todo or not os.path.exists(lib) or have != want:
        if os.path.exists(stamp):
            os.remove(stamp)
        cmd = [exe, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic"] + objs + ["-o", lib + ".tmp"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        os.replace(lib + ".tmp", lib)
        with open(stamp, "w") as fh:
            fh.write(want)
    # stale objects: whatever no library present in this directory was linked from
    if not extra_flags:
        keep = set()
        for st in os.listdir(HERE):                      # librtx_hip.so.objs, librtx_hip_lab.so.objs, lib_variant_*.so.objs
            if st.endswith(".so.objs") and os.path.exists(os.path.join(HERE, st[:-5])):
                keep.update(open(os.path.join(HERE, st)).read().split())
        for f in os.listdir(OBJ_DIR):
            if f.endswith(".o") and f not in keep:
                os.remove(os.path.join(OBJ_DIR, f))
    return lib


def build_all(force=False, verbose=False):
    """The product and the lab library."""
    return build(force=force, verbose=verbose), build(force=force, verbose=verbose, lab=True)


if __name__ == "__main__":
    import sys
    for path in build_all(force="--force" in sys.argv, verbose=True):
        print(path)
