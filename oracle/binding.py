"""ctypes binding of oracle/liboracle.so (the f64 CPU restatement of the reference path).

TEST INFRASTRUCTURE ONLY: the product package must not import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")

FAITHFUL, TIGHT = 0, 1

# struct oracle_object: the same 88-byte layout as the product's rthip_object.
OBJECT_DTYPE = np.dtype(
    [("kind", "<i4"), ("material", "<i4"), ("center", "<f8", (3,)), ("extent", "<f8", (3,)),
     ("albedo", "<f8", (3,)), ("param", "<f8")], align=True)


class CameraDesc(C.Structure):
    _fields_ = [("origin", C.c_double * 3), ("direction", C.c_double * 3), ("fov", C.c_double),
                ("focus_dist", C.c_double), ("defocus_angle", C.c_double), ("width", C.c_int32), ("height", C.c_int32)]


class RenderParams(C.Structure):
    _fields_ = [("spp_begin", C.c_uint32), ("spp_count", C.c_uint32), ("max_depth", C.c_int32), ("seed", C.c_uint32),
                ("sky_mode", C.c_int32), ("mode", C.c_int32), ("threads", C.c_int32), ("x0", C.c_int32), ("y0", C.c_int32),
                ("x1", C.c_int32), ("y1", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("rays", "box_tests", "sphere_tests", "cube_tests", "paths", "scatter_lambertian",
                                          "scatter_metal", "scatter_dielectric", "sky_misses", "rng_draws")]


_lib = None
_D3 = C.c_double * 3


def build():
    subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.oracle_scene_create.restype = C.c_void_p
        L.oracle_scene_create.argtypes = [C.c_void_p, C.c_int64]
        L.oracle_scene_create_ordered.restype = C.c_void_p
        L.oracle_scene_create_ordered.argtypes = [C.c_void_p, C.c_int64, C.c_int32]
        L.oracle_scene_destroy.argtypes = [C.c_void_p]
        L.oracle_bvh_depth.restype = C.c_int32
        L.oracle_bvh_depth.argtypes = [C.c_void_p]
        L.oracle_bvh_leaf_order.restype = C.c_int64
        L.oracle_bvh_leaf_order.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]
        L.oracle_camera_fields.argtypes = [C.POINTER(CameraDesc), C.POINTER(C.c_double)]
        L.oracle_camera_rotation.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double)]
        L.oracle_trace_primary.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.c_int32, C.c_int32, C.c_uint32, C.c_uint32,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_render.restype = C.c_double
        L.oracle_render.argtypes = [C.c_void_p, C.POINTER(CameraDesc), C.POINTER(RenderParams), C.c_void_p, C.c_void_p,
                                    C.POINTER(Counters)]
        L.oracle_quantize_rgb8.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
        L.oracle_object_hit.restype = C.c_int32
        L.oracle_object_hit.argtypes = [C.c_int32, _D3, _D3, _D3, _D3, C.POINTER(C.c_double)]
        L.oracle_object_normal.argtypes = [C.c_int32, _D3, _D3, _D3, _D3]
        L.oracle_object_aabb.argtypes = [C.c_int32, _D3, _D3, _D3, _D3]
        L.oracle_aabb_hit.restype = C.c_int32
        L.oracle_aabb_hit.argtypes = [_D3, _D3, _D3, _D3]
        L.oracle_slab_hit.restype = C.c_int32
        L.oracle_slab_hit.argtypes = [_D3, _D3, _D3, _D3, C.c_double, C.POINTER(C.c_double)]
        L.oracle_sky.argtypes = [_D3, C.c_int32, _D3]
        L.oracle_reflectance.restype = C.c_double
        L.oracle_reflectance.argtypes = [C.c_double, C.c_double, C.c_int32]
        L.oracle_reflect.argtypes = [_D3, _D3, _D3]
        L.oracle_refract.argtypes = [_D3, _D3, C.c_double, _D3]
        L.oracle_scatter.restype = C.c_int32
        L.oracle_scatter.argtypes = [C.c_int32, _D3, C.c_double, _D3, _D3, _D3, C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32,
                                     C.c_int32, _D3, _D3, C.POINTER(C.c_int32)]
        L.oracle_emission.argtypes = [C.c_int32, _D3, C.c_double, _D3]
        L.oracle_rng_words.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_void_p]
        L.oracle_camera_ray.argtypes = [C.POINTER(CameraDesc), C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, C.c_int32, _D3, _D3]
        L.oracle_camera_rays.argtypes = [C.POINTER(CameraDesc), C.c_uint32, C.c_uint32, C.c_int32, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def camera_rays(cam, seed, sample, jitter):
    """(origins, dirs), each [H, W, 3]: the primary rays of sample `sample`."""
    o = np.zeros((cam.height, cam.width, 3), np.float64)
    d = np.zeros((cam.height, cam.width, 3), np.float64)
    lib().oracle_camera_rays(C.byref(cam), seed, sample, int(jitter), o.ctypes.data, d.ctypes.data)
    return o, d


def d3(v):
    return _D3(*[float(x) for x in v])


def camera_desc(origin, direction, fov, focus_dist, defocus_angle, width, height):
    return CameraDesc(d3(origin), d3(direction), float(fov), float(focus_dist), float(defocus_angle), int(width), int(height))


class OracleScene:
    """A scene with its reference-shaped BVH built (oracle_scene_create)."""

    def __init__(self, objects, tie_reverse=False):
        """tie_reverse: objects with equal sort keys in BVH::new end up in reversed instead of kept
        order (the reference's sort_unstable_by, src/bvh.rs:49, leaves their order open)."""
        self.objects = np.ascontiguousarray(objects, dtype=OBJECT_DTYPE)
        self._h = lib().oracle_scene_create_ordered(self.objects.ctypes.data, len(self.objects), int(bool(tie_reverse)))
        if not self._h:
            raise ValueError("empty scene: BVH::new returns None")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_scene_destroy(self._h)
            self._h = None

    def depth(self):
        return lib().oracle_bvh_depth(self._h)

    def leaf_order(self):
        n = len(self.objects)
        ids, sizes, nl = np.zeros(n, np.int32), np.zeros(n, np.int32), C.c_int64(0)
        lib().oracle_bvh_leaf_order(self._h, ids.ctypes.data, sizes.ctypes.data, C.byref(nl))
        return ids, sizes[:nl.value].copy()

    def trace_primary(self, cam, mode=TIGHT, jitter=False, seed=0, sample=0):
        h, w = cam.height, cam.width
        ids, t = np.zeros((h, w), np.int32), np.zeros((h, w), np.float64)
        nrm, front = np.zeros((h, w, 3), np.float64), np.zeros((h, w), np.uint8)
        lib().oracle_trace_primary(self._h, C.byref(cam), mode, int(jitter), seed, sample, ids.ctypes.data, t.ctypes.data,
                                   nrm.ctypes.data, front.ctypes.data)
        return ids, t, nrm, front

    def render(self, cam, spp, max_depth, seed, mode=TIGHT, sky_mode=0, spp_begin=0, threads=0, window=None, samples=False):
        """Returns (accum [H,W,3] = sum_s sqrt(sample), samples [spp,H,W,3] or None, counters dict, seconds)."""
        h, w = cam.height, cam.width
        p = RenderParams(spp_begin, spp, max_depth, seed, sky_mode, mode, threads, 0, 0, 0, 0)
        if window is not None:
            p.x0, p.y0, p.x1, p.y1 = window
        accum = np.zeros((h, w, 3), np.float64)
        smp = np.zeros((spp, h, w, 3), np.float64) if samples else None
        c = Counters()
        secs = lib().oracle_render(self._h, C.byref(cam), C.byref(p), accum.ctypes.data, smp.ctypes.data if samples else None,
                                   C.byref(c))
        return accum, smp, {n: getattr(c, n) for n, _ in Counters._fields_}, secs


def quantize_rgb8(mean):
    mean = np.ascontiguousarray(mean, dtype=np.float64)
    out = np.zeros(mean.shape, np.uint8)
    lib().oracle_quantize_rgb8(mean.ctypes.data, mean.size, out.ctypes.data)
    return out


# Algorithmic work per unit, from counters of a TIGHT-mode run: one box test = 3 x (1 div,
# 2 sub, 2 mul) = 15 flop, sphere test 30, cube test 25, hit record 12, scatter 40 / 55 / 70
# (Lambertian / Metal / Dielectric), sky miss 12, camera ray 45 per path, sqrt-accumulate 6.
def algorithmic_flops(c):
    hits = c["rays"] - c["sky_misses"]
    return (15 * c["box_tests"] + 30 * c["sphere_tests"] + 25 * c["cube_tests"] + 12 * hits + 40 * c["scatter_lambertian"] +
            55 * c["scatter_metal"] + 70 * c["scatter_dielectric"] + 12 * c["sky_misses"] + (45 + 6) * c["paths"])


def algorithmic_bytes(c):
    hits = c["rays"] - c["sky_misses"]
    return 32 * c["box_tests"] + 16 * c["sphere_tests"] + 24 * c["cube_tests"] + 20 * hits
