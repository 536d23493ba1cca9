"""ctypes binding of the CPU oracle (oracle/rtx_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- as the checker, never by the product package.  See rtx_oracle.h for
the parity-pinning statement ("parity unpinned" beyond the four camera KATs of
src/raytracing/camera.rs:82-109).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librtx_oracle.so")

SPHERE, PLANE, TRIANGLE = 0, 1, 2
MODE_CLEAN, MODE_FAITHFUL = 0, 1

# identical in layout to rtxo_object (rtx_oracle.h) and RtxObject (include/rtx_hip.h): 136 bytes
OBJECT_DTYPE = np.dtype([
    ("kind", "<u4"), ("pad", "<u4"), ("geom", "<f8", (9,)),
    ("base_color", "<f8", (3,)), ("emission_color", "<f8", (3,)), ("roughness", "<f8"),
])
assert OBJECT_DTYPE.itemsize == 136


class Vec3(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]

    def tuple(self):
        return (self.x, self.y, self.z)


class Mat3(C.Structure):
    _fields_ = [("x", Vec3), ("y", Vec3), ("z", Vec3)]


class Camera(C.Structure):
    _fields_ = [("fov", C.c_double), ("position", Vec3), ("direction", Vec3),
                ("to_cam_space", Mat3), ("to_world_space", Mat3)]


class Config(C.Structure):
    _fields_ = [("rays_per_pixel", C.c_uint64), ("max_bounces", C.c_uint64),
                ("focal_length", C.c_double), ("focal_offset", C.c_double),
                ("non_focal_offset", C.c_double), ("seed", C.c_uint64)]


class Scene(C.Structure):
    _fields_ = [("config", Config), ("camera", Camera),
                ("n_objects", C.c_uint64), ("objects", C.c_void_p)]


def build(force=False):
    """Compile oracle/librtx_oracle.so with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "rtx_oracle.c")
    hdrs = [os.path.join(_HERE, "rtx_oracle.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(f) for f in [src] + hdrs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "librtx_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    V = Vec3
    dp = C.POINTER(C.c_double)
    L.rtxo_rng_key.restype = C.c_uint64
    L.rtxo_rng_key.argtypes = [C.c_uint64] * 3
    L.rtxo_rng_u01.restype = C.c_double
    L.rtxo_rng_u01.argtypes = [C.c_uint64, C.c_uint64]
    L.rtxo_dot.restype = C.c_double
    L.rtxo_dot.argtypes = [V, V]
    for name in ("rtxo_cross",):
        getattr(L, name).restype = V
        getattr(L, name).argtypes = [V, V]
    L.rtxo_len.restype = C.c_double
    L.rtxo_len.argtypes = [V]
    L.rtxo_norm.restype = V
    L.rtxo_norm.argtypes = [V]
    L.rtxo_camera_new.restype = None
    L.rtxo_camera_new.argtypes = [C.POINTER(Camera), V, V, C.c_double]
    L.rtxo_camera_set_direction.restype = None
    L.rtxo_camera_set_direction.argtypes = [C.POINTER(Camera), V]
    for name in ("rtxo_camera_to_cam_space", "rtxo_camera_to_world_space", "rtxo_camera_rotate_to_world_space"):
        getattr(L, name).restype = V
        getattr(L, name).argtypes = [C.POINTER(Camera), V]
    for name in ("rtxo_sphere_distance", "rtxo_plane_distance", "rtxo_triangle_distance"):
        getattr(L, name).restype = C.c_int
        getattr(L, name).argtypes = [dp, V, V, dp]
    for name in ("rtxo_sphere_normal", "rtxo_plane_normal", "rtxo_triangle_normal"):
        getattr(L, name).restype = V
        getattr(L, name).argtypes = [dp, V]
    L.rtxo_triangle_contains.restype = C.c_int
    L.rtxo_triangle_contains.argtypes = [dp, V]
    L.rtxo_set_sincos_mode.restype = None
    L.rtxo_set_sincos_mode.argtypes = [C.c_int]
    L.rtxo_device_sincos.restype = None
    L.rtxo_device_sincos.argtypes = [C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.rtxo_trace_row.restype = C.c_int
    L.rtxo_trace_row.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    L.rtxo_device_sincos_n.restype = None
    L.rtxo_device_sincos_n.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.rtxo_random_direction.restype = V
    L.rtxo_random_direction.argtypes = [C.c_double, C.c_double]
    L.rtxo_random_bounce_dir.restype = V
    L.rtxo_random_bounce_dir.argtypes = [V, V, C.c_double, C.c_double, C.c_double]
    L.rtxo_closest_object.restype = C.c_int64
    L.rtxo_closest_object.argtypes = [C.POINTER(Scene), V, V, dp]
    L.rtxo_get_ray_dir.restype = V
    L.rtxo_get_ray_dir.argtypes = [C.POINTER(Scene), C.c_double, C.c_double, C.c_double]
    L.rtxo_render.restype = C.c_int
    L.rtxo_render.argtypes = [C.POINTER(Scene), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                              C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    L.rtxo_render_pixels.restype = C.c_int
    L.rtxo_render_pixels.argtypes = [C.POINTER(Scene), C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_int]
    L.rtxo_quantize_image.restype = None
    L.rtxo_quantize_image.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    _lib = L
    return L


def vec(v):
    return Vec3(float(v[0]), float(v[1]), float(v[2]))


def _geom(values, n):
    arr = (C.c_double * n)(*[float(x) for x in values])
    return arr


def rng_u01(seed, pixel, sample, draw):
    L = lib()
    return L.rtxo_rng_u01(L.rtxo_rng_key(seed, pixel, sample), draw)


def camera_new(position, direction, fov):
    cam = Camera()
    lib().rtxo_camera_new(C.byref(cam), vec(position), vec(direction), float(fov))
    return cam


def sphere_distance(center, radius, pos, direction):
    g = _geom(list(center) + [radius], 4)
    d = C.c_double()
    ok = lib().rtxo_sphere_distance(g, vec(pos), vec(direction), C.byref(d))
    return d.value if ok else None


def plane_distance(position, normal, pos, direction):
    g = _geom(list(position) + list(normal), 6)
    d = C.c_double()
    ok = lib().rtxo_plane_distance(g, vec(pos), vec(direction), C.byref(d))
    return d.value if ok else None


def triangle_distance(v0, v1, v2, pos, direction):
    g = _geom(list(v0) + list(v1) + list(v2), 9)
    d = C.c_double()
    ok = lib().rtxo_triangle_distance(g, vec(pos), vec(direction), C.byref(d))
    return d.value if ok else None


def triangle_contains(v0, v1, v2, point):
    g = _geom(list(v0) + list(v1) + list(v2), 9)
    return lib().rtxo_triangle_contains(g, vec(point))


def triangle_normal(v0, v1, v2):
    g = _geom(list(v0) + list(v1) + list(v2), 9)
    return lib().rtxo_triangle_normal(g, Vec3(0, 0, 0)).tuple()


def closest_object(scene, pos, direction):
    """(index, distance) of Scene::closest_object (scene.rs:243-251); (-1, None) when nothing is hit."""
    d = C.c_double()
    i = lib().rtxo_closest_object(C.byref(scene), vec(pos), vec(direction), C.byref(d))
    return (int(i), d.value) if i >= 0 else (-1, None)


def random_direction(u_z, u_theta):
    return lib().rtxo_random_direction(float(u_z), float(u_theta)).tuple()


def random_bounce_dir(ray_dir, normal, roughness, u_z, u_theta):
    return lib().rtxo_random_bounce_dir(vec(ray_dir), vec(normal), float(roughness),
                                        float(u_z), float(u_theta)).tuple()


def make_scene(objects, camera, rays_per_pixel=16, max_bounces=10, focal_length=10.0,
               focal_offset=1e-4, non_focal_offset=1e-1, seed=42):
    """objects: numpy array of OBJECT_DTYPE in scene order; camera: (position, direction, fov)."""
    objects = np.ascontiguousarray(objects, dtype=OBJECT_DTYPE)
    s = Scene()
    s.config = Config(int(rays_per_pixel), int(max_bounces), float(focal_length), float(focal_offset),
                      float(non_focal_offset), int(seed))
    s.camera = camera_new(*camera)
    s.n_objects = len(objects)
    s.objects = objects.ctypes.data if len(objects) else None
    s._keepalive = objects
    return s


def render(scene, width, height, n_threads=None, mode=MODE_CLEAN, row_begin=0, row_stride=1,
           want_segments=False):
    """Scene::render restatement.  Returns img[h][w][3] float64 (and segments[h][w] uint64)."""
    if n_threads is None:
        n_threads = os.cpu_count() or 1
    out = np.zeros((height, width, 3), dtype=np.float64)
    seg = np.zeros((height, width), dtype=np.uint64) if want_segments else None
    rc = lib().rtxo_render(C.byref(scene), width, height, row_begin, row_stride,
                           out.ctypes.data, seg.ctypes.data if want_segments else None,
                           int(n_threads), int(mode))
    if rc != 0:
        raise RuntimeError("rtxo_render failed: %d" % rc)
    return (out, seg) if want_segments else out


PATH_STEP_DTYPE = np.dtype([("position", "<f8", (3,)), ("direction", "<f8", (3,)), ("distance", "<f8"), ("object", "<i8")])


def trace_row(scene, width, height, row, max_steps):
    """(steps, counts): the transcript of every path of image row `row` -- per segment the ray closest_object was asked about
    (scene.rs:232), the winning distance and object index (-1, inf: none).  steps [width][rays_per_pixel][max_steps], counts
    [width][rays_per_pixel]."""
    spp = int(scene.config.rays_per_pixel)
    steps = np.zeros((int(width), spp, int(max_steps)), dtype=PATH_STEP_DTYPE)
    counts = np.zeros((int(width), spp), dtype=np.uint32)
    rc = lib().rtxo_trace_row(C.byref(scene), int(width), int(height), int(row), int(max_steps), steps.ctypes.data, counts.ctypes.data)
    if rc != 0:
        raise RuntimeError("rtxo_trace_row failed: %d" % rc)
    return steps, counts


def render_pixels(scene, width, height, xs, ys, n_threads=None, want_segments=False):
    """img[ys[k]][xs[k]] of Scene::render's width x height frame for a list of pixels -> (n, 3) float64
    (and (n,) uint64 closest_object counts)."""
    if n_threads is None:
        n_threads = os.cpu_count() or 1
    xs = np.ascontiguousarray(xs, dtype=np.uint32)
    ys = np.ascontiguousarray(ys, dtype=np.uint32)
    assert xs.shape == ys.shape and xs.ndim == 1
    out = np.zeros((len(xs), 3), dtype=np.float64)
    seg = np.zeros(len(xs), dtype=np.uint64) if want_segments else None
    rc = lib().rtxo_render_pixels(C.byref(scene), int(width), int(height), len(xs), xs.ctypes.data, ys.ctypes.data,
                                  out.ctypes.data, seg.ctypes.data if want_segments else None, int(n_threads))
    if rc != 0:
        raise RuntimeError("rtxo_render_pixels failed: %d" % rc)
    return (out, seg) if want_segments else out


def quantize_image(rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float64)
    h, w, _ = rgb.shape
    out = np.zeros((h, w, 3), dtype=np.uint8)
    lib().rtxo_quantize_image(rgb.ctypes.data, w, h, out.ctypes.data)
    return out


def set_device_sincos(on):
    """random_direction's sin / cos: False = libm (the reference, default), True = the device's routine (the kernels then equal the
    oracle bit for bit).  Process-wide."""
    lib().rtxo_set_sincos_mode(1 if on else 0)


class device_sincos:
    """with oracle.device_sincos(): ...   -- the oracle computes random_direction's sin / cos as the device does, inside the block."""

    def __enter__(self):
        set_device_sincos(True)
        return self

    def __exit__(self, *exc):
        set_device_sincos(False)
        return False


def device_sincos_values(x):
    """(sin, cos) arrays of the device's routine evaluated on the CPU (bit-identical arithmetic)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    s, c = np.zeros_like(x), np.zeros_like(x)
    lib().rtxo_device_sincos_n(x.ctypes.data, x.size, s.ctypes.data, c.ctypes.data)
    return s, c
