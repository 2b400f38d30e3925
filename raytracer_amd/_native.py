"""ctypes binding of librthip.so (the C ABI declared in include/rthip.h).

The library is built in-tree by `raytracer_amd/csrc/Makefile` (see `__graft_entry__.build`).
There is no fallback: if the shared library is missing, importing this module raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTHIP_LIBRARY") or os.path.join(_HERE, "librthip.so")  # override: A/B builds of the same ABI

OK, ERR_INVALID, ERR_EMPTY, ERR_STATE, ERR_HIP, ERR_LIMIT, ERR_INTERNAL = 0, 1, 2, 3, 4, 5, 6
SPHERE, CUBE = 0, 1
LAMBERTIAN, METAL, DIELECTRIC = 0, 1, 2
F32, F64, F32_RAW = 0, 1, 2
SKY_REFERENCE, SKY_WALLPAPER = 0, 1
MEGAKERNEL, WAVEFRONT = 0, 1

# struct rthip_object (88 bytes)
OBJECT_DTYPE = np.dtype(
    [("kind", "<i4"), ("material", "<i4"), ("center", "<f8", (3,)), ("extent", "<f8", (3,)),
     ("albedo", "<f8", (3,)), ("param", "<f8")], align=True)
assert OBJECT_DTYPE.itemsize == 88


class CameraDesc(C.Structure):
    _fields_ = [("origin", C.c_double * 3), ("direction", C.c_double * 3), ("fov", C.c_double),
                ("focus_dist", C.c_double), ("defocus_angle", C.c_double), ("width", C.c_int32), ("height", C.c_int32)]


class RenderParams(C.Structure):
    _fields_ = [("spp_begin", C.c_uint32), ("spp_count", C.c_uint32), ("max_depth", C.c_int32), ("seed", C.c_uint32),
                ("sky_mode", C.c_int32), ("precision", C.c_int32), ("variant", C.c_int32), ("part_index", C.c_int32),
                ("part_count", C.c_int32), ("reserved", C.c_int32 * 3)]


class Stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays", C.c_uint64), ("box_tests", C.c_uint64), ("sphere_tests", C.c_uint64),
                ("cube_tests", C.c_uint64), ("kernel_ms", C.c_double), ("n_nodes", C.c_int32), ("n_leaves", C.c_int32),
                ("n_prims", C.c_int32), ("tree_depth", C.c_int32), ("scene_in_lds", C.c_int32), ("n_gates", C.c_int32),
                ("launch_threads", C.c_int32), ("launch_grid", C.c_int32), ("launch_lds_bytes", C.c_int32), ("waves_per_cu", C.c_int32)]


# Every symbol include/rthip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "rthip_create": (C.c_int32, [C.c_int32, C.POINTER(_P)]),
    "rthip_destroy": (None, [_P]),
    "rthip_last_error": (C.c_char_p, [_P]),
    "rthip_set_stream": (C.c_int32, [_P, _P]),
    "rthip_set_option": (C.c_int32, [_P, C.c_char_p, C.c_double]),
    "rthip_scene_upload": (C.c_int32, [_P, _P, C.c_int64]),
    "rthip_scene_refit": (C.c_int32, [_P, _P, C.c_int64]),
    "rthip_camera_set": (C.c_int32, [_P, C.POINTER(CameraDesc)]),
    "rthip_camera_fields": (C.c_int32, [C.POINTER(CameraDesc), C.POINTER(C.c_double)]),
    "rthip_render_accumulate": (C.c_int32, [_P, C.POINTER(RenderParams)]),
    "rthip_render_accumulate_async": (C.c_int32, [_P, C.POINTER(RenderParams)]),
    "rthip_render_accumulate_noalloc": (C.c_int32, [_P, C.POINTER(RenderParams)]),
    "rthip_synchronize": (C.c_int32, [_P]),
    "rthip_reset_accumulation": (C.c_int32, [_P]),
    "rthip_accum_buffer": (C.c_int32, [_P, C.POINTER(_P), C.POINTER(C.c_uint64)]),
    "rthip_set_accum_buffer": (C.c_int32, [_P, _P, C.c_uint64]),
    "rthip_comm_unique_id": (C.c_int32, [_P]),
    "rthip_comm_init": (C.c_int32, [_P, _P, C.c_int32, C.c_int32]),
    "rthip_merge_reduce": (C.c_int32, [_P, C.c_int32]),
    "rthip_merge_wait": (C.c_int32, [_P, _P]),
    "rthip_merge_local": (C.c_int32, [C.POINTER(_P), C.c_int32, C.c_int32]),
    "rthip_accum_download": (C.c_int32, [_P, _P, C.c_uint64]),
    "rthip_accum_upload": (C.c_int32, [_P, _P, C.c_uint64]),
    "rthip_read_accum": (C.c_int32, [_P, _P]),
    "rthip_read_rgb8": (C.c_int32, [_P, C.c_uint32, _P]),
    "rthip_read_xrgb32": (C.c_int32, [_P, C.c_uint32, _P]),
    "rthip_trace_primary": (C.c_int32, [_P, C.c_int32, C.c_int32, C.c_uint32, C.c_uint32, _P, _P, _P, _P]),
    "rthip_render_samples": (C.c_int32, [_P, C.POINTER(RenderParams), _P]),
    "rthip_get_stats": (C.c_int32, [_P, C.POINTER(Stats)]),
    "rthip_bvh_leaf_order": (C.c_int32, [_P, C.c_int64, _P, _P, C.POINTER(C.c_int64), C.POINTER(C.c_int32)]),
    "rthip_bvh_gate_count": (C.c_int32, [_P, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "rthip_scene_validate": (C.c_int32, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "rthip_bvh_validate": (C.c_int32, [_P, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]),
    "rthip_scene_generate": (C.c_int32, [C.c_char_p, C.c_uint32, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "rthip_camera_rotation": (None, [C.c_double, C.c_double, C.POINTER(C.c_double)]),
}

_lib = None


def _share_hip_runtime_with_torch():
    """PyTorch wheels bundle their own HIP and HSA runtimes (torch/lib), with the same sonames as
    the system's.  A process must run ONE of them: if librthip.so pulled in the system runtime
    first, a later `import torch` finds no GPU.  So when PyTorch is installed, its runtime is
    loaded before librthip.so, whose dependency then resolves to it by soname.  (Nothing of
    PyTorch itself is imported.)"""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if not spec or not spec.origin:
        return
    libdir = os.path.join(os.path.dirname(spec.origin), "lib")
    for name in ("libhsa-runtime64.so", "libamdhip64.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            C.CDLL(path, mode=C.RTLD_GLOBAL)


def lib():
    """The loaded library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C raytracer_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
        _share_hip_runtime_with_torch()
        handle = C.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SYMBOLS.items():
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib


class RthipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"rthip status {code}: {message}")
        self.code = code


def check(code, ctx=None):
    if code != OK:
        msg = lib().rthip_last_error(ctx)
        raise RthipError(code, msg.decode() if msg else "")
