"""In-tree build of librtx_hip.so (hipcc, gfx950 only).

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels to the GPU box
with the gpurun snapshot, so nothing is JIT-compiled there.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librtx_hip.so")
SOURCES = ["rtx_kernels.hip", "rtx_bvh.hip", "rtx_bvh_spheres.hip", "rtx_bvh_spheres_pool.hip", "rtx_bvh_regroup.hip", "rtx_bvh_mesh.hip", "rtx_wavefront.hip", "rtx_wavefront_spheres.hip", "rtx_api.hip"]
HEADERS = ["rtx_math.h", "rtx_scene.h", "rtx_bvh.h", "rtx_device.h", "rtx_traverse.h", "rtx_mesh_step.h", "rtx_wavefront.h", "rtx_launch.h"]
# -ffp-contract=off: the exact path must round like the reference (Rust never fuses a*b+c);
# the f32 filter asks for FMAs explicitly.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: librtx_hip.so cannot be built (there is no CPU fallback)")
    return exe


def _newest_input():
    files = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    files.append(os.path.join(HERE, "..", "include", "rtx_hip.h"))
    files.append(os.path.abspath(__file__))
    return max(os.path.getmtime(f) for f in files)


def build(force=False, extra_flags=(), verbose=False):
    """Compile rust-raytracing_amd/librtx_hip.so; returns its path."""
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest_input():
        return LIB
    cmd = [hipcc()] + FLAGS + list(extra_flags) + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
