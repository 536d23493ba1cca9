"""ctypes view of include/rtx_hip.h (the C ABI of librtx_hip.so).

The library is the product: if it is missing or cannot be loaded this module raises --
there is no Python/NumPy/CPU fallback for the render path.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTX_HIP_LIB") or os.path.join(HERE, "librtx_hip.so")   # RTX_HIP_LIB: A/B builds
# the lab library: same sources with -DRTX_LAB, same ABI, + the experiments behind RTX_TUNE_LAB_MASK bits and the older kernel
# families (include/rtx_hip.h, "Product and lab").  Loaded only for a Config that asks for it (Config.lab / a lab tuning bit).
LAB_LIB_PATH = os.environ.get("RTX_HIP_LAB_LIB") or os.path.join(HERE, "librtx_hip_lab.so")

RTX_SPHERE, RTX_PLANE, RTX_TRIANGLE = 0, 1, 2
RTX_KERNEL_AUTO, RTX_KERNEL_EXACT, RTX_KERNEL_MIXED, RTX_KERNEL_MIXED_VERIFY, RTX_KERNEL_BVH, RTX_KERNEL_BVH_REGROUP = 0, 1, 2, 3, 4, 5
RTX_KERNEL_WAVEFRONT = 6
# RtxConfig.tuning bits (include/rtx_hip.h): A/B switches, every combination renders the same bits
RTX_TUNE_NO_TILES, RTX_TUNE_BVH_CLASSIC, RTX_TUNE_NO_QNODES, RTX_TUNE_NO_PACKETS = 1, 2, 4, 8
RTX_TUNE_WF_PURE, RTX_TUNE_ONE_STAGE, RTX_TUNE_TWO_STAGE, RTX_TUNE_BVH_MEDIAN = 16, 32, 64, 128
RTX_TUNE_TRI_LEAF_SHIFT, RTX_TUNE_THRESH_SHIFT = 8, 12
RTX_TUNE_SORT_SURVIVORS = 1 << 19
RTX_TUNE_PK_LDS_STACK = 1 << 20
RTX_TUNE_STAGE2_POOL = 1 << 21
RTX_TUNE_STAGE2_PAIR = 1 << 22
RTX_TUNE_NO_CUT = 1 << 23
RTX_TUNE_HALVES = 1 << 27
RTX_TUNE_NO_HALVES = 1 << 28
RTX_TUNE_NO_TILE_LISTS = 1 << 29
RTX_TUNE_BEAMS = 1 << 24
RTX_TUNE_INLINE_LEAVES = 1 << 25
RTX_TUNE_STAGE2_SLOTS = 1 << 26
RTX_TUNE_LAB_MASK = (RTX_TUNE_BVH_CLASSIC | RTX_TUNE_NO_QNODES | RTX_TUNE_NO_PACKETS | RTX_TUNE_WF_PURE | RTX_TUNE_PK_LDS_STACK |
                     RTX_TUNE_STAGE2_POOL | RTX_TUNE_STAGE2_PAIR | RTX_TUNE_BEAMS | RTX_TUNE_INLINE_LEAVES | RTX_TUNE_SORT_SURVIVORS |
                     RTX_TUNE_STAGE2_SLOTS)
RTX_TUNE_KNOWN_MASK = (RTX_TUNE_LAB_MASK | RTX_TUNE_NO_TILES | RTX_TUNE_ONE_STAGE | RTX_TUNE_TWO_STAGE | RTX_TUNE_BVH_MEDIAN |
                       (15 << RTX_TUNE_TRI_LEAF_SHIFT) | (127 << RTX_TUNE_THRESH_SHIFT) | RTX_TUNE_NO_CUT | RTX_TUNE_HALVES | RTX_TUNE_NO_HALVES | RTX_TUNE_NO_TILE_LISTS)
RTX_OK, RTX_ERR_INVALID_ARGUMENT, RTX_ERR_NO_DEVICE, RTX_ERR_HIP, RTX_ERR_UNSUPPORTED, RTX_ERR_OUT_OF_MEMORY = range(6)

# RtxObject, 136 bytes: one entry of Scene.objects (scene.rs:80; object.rs:9-15,78-86)
OBJECT_DTYPE = np.dtype([
    ("kind", "<u4"), ("reserved", "<u4"), ("geom", "<f8", (9,)),
    ("base_color", "<f8", (3,)), ("emission_color", "<f8", (3,)), ("roughness", "<f8"),
])
assert OBJECT_DTYPE.itemsize == 136


class RtxConfig(C.Structure):
    _fields_ = [("rays_per_pixel", C.c_uint64), ("max_bounces", C.c_uint64),
                ("focal_length", C.c_double), ("focal_offset", C.c_double), ("non_focal_offset", C.c_double),
                ("seed", C.c_uint64), ("kernel", C.c_uint32), ("tuning", C.c_uint32)]


class RtxCamera(C.Structure):
    _fields_ = [("fov", C.c_double), ("position", C.c_double * 3), ("direction", C.c_double * 3),
                ("to_cam_space", C.c_double * 9), ("to_world_space", C.c_double * 9)]


class RtxScene(C.Structure):
    _fields_ = [("config", RtxConfig), ("camera", RtxCamera), ("n_objects", C.c_uint64), ("objects", C.c_void_p)]


class RtxStats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("segments", C.c_uint64), ("exact_tests", C.c_uint64),
                ("filter_tests", C.c_uint64), ("trace_ms", C.c_double), ("resolve_ms", C.c_double),
                ("filter_mismatches", C.c_uint64), ("box_tests", C.c_uint64),
                ("trace_launches", C.c_uint32), ("kernel", C.c_uint32),
                ("stage1_ms", C.c_double), ("stage1_box_tests", C.c_uint64), ("stage1_filter_tests", C.c_uint64),
                ("stage1_exact_tests", C.c_uint64)]


# every symbol include/rtx_hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("rtx_version", C.c_int32, []),
    ("rtx_last_error", C.c_char_p, []),
    ("rtx_device_count", C.c_int32, []),
    ("rtx_lab_build", C.c_int32, []),
    ("rtx_camera_new", C.c_int32, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_double, C.POINTER(RtxCamera)]),
    ("rtx_render", C.c_int32, [C.POINTER(RtxScene), C.c_uint32, C.c_uint32, C.c_void_p]),
    ("rtx_render_to_image", C.c_int32, [C.POINTER(RtxScene), C.c_uint32, C.c_uint32, C.c_void_p]),
    ("rtx_render_devices", C.c_int32, [C.POINTER(RtxScene), C.c_uint32, C.c_uint32, C.POINTER(C.c_int32), C.c_uint32, C.c_void_p]),
    ("rtx_render_to_image_devices", C.c_int32, [C.POINTER(RtxScene), C.c_uint32, C.c_uint32, C.POINTER(C.c_int32), C.c_uint32,
                                                C.c_void_p]),
    ("rtx_scene_upload", C.c_int32, [C.POINTER(RtxScene), C.c_int32, C.POINTER(C.c_void_p)]),
    ("rtx_scene_free", C.c_int32, [C.c_void_p]),
    ("rtx_scene_set_config", C.c_int32, [C.c_void_p, C.POINTER(RtxConfig)]),
    ("rtx_scene_set_scratch_limit", C.c_int32, [C.c_void_p, C.c_uint64]),
    ("rtx_scene_set_camera", C.c_int32, [C.c_void_p, C.POINTER(RtxCamera)]),
    ("rtx_scene_append_objects", C.c_int32, [C.c_void_p, C.c_void_p, C.c_uint64]),
    ("rtx_render_rows", C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.c_void_p, C.c_void_p, C.POINTER(RtxStats)]),
    ("rtx_blocks_row_count", C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    ("rtx_render_blocks", C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                      C.c_void_p, C.c_void_p, C.POINTER(RtxStats)]),
    ("rtx_quantize_image_device", C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int32, C.c_void_p]),
    ("rtx_debug_math", C.c_int32, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]),
    ("rtx_debug_paths", C.c_int32, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]),
    ("rtx_debug_host_scene", C.c_int32, [C.POINTER(RtxScene), C.POINTER(C.c_uint64)]),
]


class RtxError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("librtx_hip status %d: %s" % (status, message))
        self.status = status


_libs = {}


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME
    libamdhip64.so.7, the same SONAME as /opt/rocm's).  If it is mapped before librtx_hip.so, the dynamic
    linker resolves our NEEDED libamdhip64.so.7 to it, and torch tensors' device pointers, streams and our
    launches all live in one runtime.  torch itself is not imported here."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library(lab=False):
    """dlopen librtx_hip.so (lab=True: librtx_hip_lab.so) and bind every entry point.  Raises if the library is absent."""
    lab = bool(lab)
    if lab in _libs:
        return _libs[lab]
    path = LAB_LIB_PATH if lab else LIB_PATH
    if not os.path.exists(path):
        raise RuntimeError(
            "%s is not built (%s). Build it with `python rust-raytracing_amd/build.py` "
            "(hipcc, gfx950). There is no CPU fallback for the render path." % (os.path.basename(path), path))
    _share_hip_runtime_with_torch()
    lib = C.CDLL(path)                   # RTLD_LOCAL: the two libraries export the same names and stay apart
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = restype
        fn.argtypes = argtypes
    if "RTX_HIP_LIB" not in os.environ and "RTX_HIP_LAB_LIB" not in os.environ and int(lib.rtx_lab_build()) != int(lab):
        raise RuntimeError("%s reports rtx_lab_build() = %d" % (path, int(lib.rtx_lab_build())))
    _libs[lab] = lib
    return lib


def check(status, lib=None):
    if status != 0:
        msg = (lib or load_library()).rtx_last_error()
        raise RtxError(status, msg.decode("utf-8", "replace") if msg else "")
