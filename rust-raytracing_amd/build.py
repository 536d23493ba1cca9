"""In-tree build of librtx_hip.so (hipcc, gfx950 only).

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box
with the gpurun snapshot, so nothing is JIT-compiled there.

Every .hip file is its own translation unit (no cross-file device calls), so the objects are
compiled in parallel and cached by content hash of (source, headers, flags); the link is one
`hipcc -shared`.
"""
import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librtx_hip.so")
OBJ_DIR = os.path.join(HERE, "build")
SOURCES = ["rtx_kernels.hip", "rtx_bvh.hip", "rtx_bvh_spheres.hip", "rtx_bvh_spheres_pool.hip", "rtx_bvh_regroup.hip",
           "rtx_bvh_mesh.hip", "rtx_wavefront.hip", "rtx_wavefront_spheres.hip", "rtx_api.hip"]
HEADERS = ["rtx_math.h", "rtx_scene.h", "rtx_bvh.h", "rtx_device.h", "rtx_traverse.h", "rtx_mesh_step.h", "rtx_wavefront.h",
           "rtx_launch.h"]
# -ffp-contract=off: the exact path must round like the reference (Rust never fuses a*b+c);
# the f32 filter asks for FMAs explicitly.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


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


def build(force=False, extra_flags=(), verbose=False, lib=LIB, jobs=None):
    """Compile rust-raytracing_amd/librtx_hip.so (or `lib`); returns its path."""
    flags = FLAGS + list(extra_flags)
    os.makedirs(OBJ_DIR, exist_ok=True)
    exe = hipcc()
    objs, todo = [], []
    for s in SOURCES:
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
        # stale objects of earlier digests
        keep = set(objs)
        for f in os.listdir(OBJ_DIR):
            p = os.path.join(OBJ_DIR, f)
            if f.endswith(".o") and p not in keep and not extra_flags:
                os.remove(p)
    if todo or not os.path.exists(lib):
        cmd = [exe, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib + ".tmp"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        os.replace(lib + ".tmp", lib)
    return lib


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
