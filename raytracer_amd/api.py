"""Host-side mirror of the reference's Scene / Camera / BVH API over librthip.

Names and argument meaning follow the Rust crate (file:line cited per item); what differs is
stated: scene factories take a seed (the reference draws from an OS-seeded generator), and
the per-sample `render()` (src/renderer.rs:29-40) is replaced by `Renderer.accumulate`, one
call per pass over the image, because the sample loop itself runs on the GPU.
"""
import ctypes as C

import numpy as np

from . import _native as N


class Material:
    """src/materials.rs:4-8."""

    def __init__(self, kind, albedo=(0.0, 0.0, 0.0), param=0.0):
        self.kind, self.albedo, self.param = kind, tuple(float(a) for a in albedo), float(param)

    @staticmethod
    def lambertian(albedo, emission=0.0):
        return Material(N.LAMBERTIAN, albedo, emission)

    @staticmethod
    def metal(albedo, fuzz):
        return Material(N.METAL, albedo, fuzz)

    @staticmethod
    def dielectric(ir):
        return Material(N.DIELECTRIC, (0.0, 0.0, 0.0), ir)


class Scene:
    """src/scene.rs:80-98: an ordered list of objects; the position in the list is the object id."""

    def __init__(self, objects=None):
        self.objects = np.zeros(0, dtype=N.OBJECT_DTYPE) if objects is None else np.ascontiguousarray(objects, dtype=N.OBJECT_DTYPE)

    def __len__(self):
        return len(self.objects)

    def _push(self, kind, center, extent, material):
        o = np.zeros(1, dtype=N.OBJECT_DTYPE)
        o["kind"], o["material"] = kind, material.kind
        o["center"], o["extent"], o["albedo"], o["param"] = center, extent, material.albedo, material.param
        self.objects = np.concatenate([self.objects, o])

    def add_sphere(self, center, radius, material):  # src/scene.rs:90-93
        self._push(N.SPHERE, center, (radius, 0.0, 0.0), material)

    def add_cube(self, center, size, material):  # src/scene.rs:95-98 (size = half extents)
        self._push(N.CUBE, center, size, material)

    @staticmethod
    def generate(name, seed=0x5EED0001):
        """Seeded scene factory by name (see rthip_scene_generate in include/rthip.h)."""
        lib = N.lib()
        n = C.c_int64(0)
        N.check(lib.rthip_scene_generate(name.encode(), seed, None, 0, C.byref(n)))
        objs = np.zeros(n.value, dtype=N.OBJECT_DTYPE)
        N.check(lib.rthip_scene_generate(name.encode(), seed, objs.ctypes.data, n.value, C.byref(n)))
        return Scene(objs)

    @staticmethod
    def create_sample_scene(seed=0x5EED0001):  # src/scene.rs:122-205
        return Scene.generate("default", seed)


def combine_spheres_and_cubes(spheres, cubes):  # src/scenes.rs:5-14
    return Scene(np.concatenate([np.asarray(spheres, dtype=N.OBJECT_DTYPE), np.asarray(cubes, dtype=N.OBJECT_DTYPE)]))


def _split(scene):
    o = scene.objects
    return o[o["kind"] == N.SPHERE], o[o["kind"] == N.CUBE]


def create_simple_scene():  # src/scenes.rs:17-75 -> (spheres, cubes)
    return _split(Scene.generate("simple"))


def create_glass_scene():  # src/scenes.rs:78-121
    return _split(Scene.generate("glass"))


def create_10_metallic_scene():  # src/scenes.rs:124-170
    return _split(Scene.generate("metal"))


def create_wallpaper_scene():  # src/scenes.rs:173-198
    return _split(Scene.generate("wallpaper"))


def create_sample_scene(seed=0x5EED0001):  # src/scenes.rs:201-300 (many spheres and cubes)
    return _split(Scene.generate("sample", seed))


def create_synthetic_scene(n, seed=0x5EED0001):
    """n random spheres and cubes (no counterpart in the reference)."""
    return Scene.generate(f"synthetic:{int(n)}", seed)


def get_camera_rotation(yaw, pitch):  # src/camera.rs:95-103
    out = (C.c_double * 3)()
    N.lib().rthip_camera_rotation(float(yaw), float(pitch), out)
    return tuple(out)


class Camera:
    """src/camera.rs:25-33 Camera::new(origin, direction, fov, focus_dist, defocus_angle, width, height)."""

    DEFAULT_FOV, DEFAULT_FOCUS_DIST, DEFAULT_DEFOCUS_ANGLE = 90.0, 10.0, 0.6  # src/camera.rs:21-23

    def __init__(self, origin, direction, fov, focus_dist, defocus_angle, width, height):
        self.desc = N.CameraDesc((C.c_double * 3)(*origin), (C.c_double * 3)(*direction), float(fov), float(focus_dist),
                                 float(defocus_angle), int(width), int(height))

    @property
    def width(self):
        return self.desc.width

    @property
    def height(self):
        return self.desc.height

    def fields(self):
        """The derived private fields of src/camera.rs:5-11 as a dict of numpy arrays."""
        out = (C.c_double * 19)()
        N.check(N.lib().rthip_camera_fields(C.byref(self.desc), out))
        v = np.array(out[:18]).reshape(6, 3)
        names = ["origin", "pixel00_loc", "pixel_delta_x", "pixel_delta_y", "defocus_disk_x", "defocus_disk_y"]
        d = {n: v[i] for i, n in enumerate(names)}
        d["defocus_angle"] = out[18]
        return d


def merge_local(renderers, root=0):
    """Sums the accumulation buffers of several Renderers of ONE process into renderers[root]'s
    (rthip_merge_local: peer copies and an add kernel, no RCCL)."""
    arr = (C.c_void_p * len(renderers))(*[r._ctx for r in renderers])
    code = N.lib().rthip_merge_local(arr, len(renderers), int(root))
    if code != N.OK:
        renderers[root]._raise(code)


def bvh_gate_count(scene):
    """(gate boxes, gated spheres) of the scene's build (rthip_bvh_gate_count; host only)."""
    g, s = C.c_int64(0), C.c_int64(0)
    N.check(N.lib().rthip_bvh_gate_count(scene.objects.ctypes.data, len(scene), C.byref(g), C.byref(s)))
    return g.value, s.value


class BVH:
    """src/bvh.rs:17: BVH::new(scene) -> Option<BVH>.  Host-side view of the tree shape; the
    device copy is made by Renderer."""

    def __init__(self, scene):
        if len(scene) == 0:
            raise ValueError("BVH::new(empty scene) is None")  # src/bvh.rs:51
        self.scene = scene
        n = len(scene)
        ids = np.zeros(n, dtype=np.int32)
        sizes = np.zeros(n, dtype=np.int32)
        nl, depth = C.c_int64(0), C.c_int32(0)
        N.check(N.lib().rthip_bvh_leaf_order(scene.objects.ctypes.data, n, ids.ctypes.data, sizes.ctypes.data, C.byref(nl),
                                             C.byref(depth)))
        self.leaf_ids, self.leaf_sizes, self.depth = ids, sizes[:nl.value].copy(), depth.value


class Renderer:
    """Owns one GPU context: the frame loop of raytracer_headless/src/main.rs:34-57 and the
    progressive loop of raytracer_runtime/src/main.rs:69-94 (update_camera / reset_accumulation
    / frame counter follow the shape of raytracer_gpu/src/renderer.rs:484-587)."""

    def __init__(self, scene, camera, device=0, max_depth=10, seed=0x5EED0001, sky_mode=N.SKY_REFERENCE, precision=N.F32,
                 variant=N.MEGAKERNEL, options=None):
        self._lib = N.lib()
        self._ctx = C.c_void_p()
        self.device = int(device)
        N.check(self._lib.rthip_create(int(device), C.byref(self._ctx)))
        for name, value in (options or {}).items():   # build knobs must be set before the scene is uploaded
            self.set_option(name, value)
        self.max_depth, self.seed, self.sky_mode, self.precision = max_depth, seed, sky_mode, precision
        self.variant = variant  # MEGAKERNEL (persistent threads) or WAVEFRONT (generate / extend / shade per bounce)
        self.spp_done = 0
        self._upload(scene)
        self.update_camera(camera)

    def _upload(self, scene):
        objs = scene.objects if isinstance(scene, Scene) else scene.scene.objects
        self._objs = np.ascontiguousarray(objs, dtype=N.OBJECT_DTYPE)
        code = self._lib.rthip_scene_upload(self._ctx, self._objs.ctypes.data, len(self._objs))
        if code != N.OK:
            self._raise(code)

    def set_scene(self, scene):
        """Another scene (or the same one under other build options) into this context: tree rebuilt,
        accumulation cleared, camera kept."""
        self._upload(scene)
        self.reset_accumulation()

    def refit_scene(self, scene):
        """The same objects at new places (rthip_scene_refit): the device tree keeps its shape, its
        boxes are fitted again on the GPU; results equal those of a fresh upload.  Accumulation
        cleared, camera kept.  Raises when the objects are not the ones the tree was built over."""
        objs = scene.objects if isinstance(scene, Scene) else scene.scene.objects
        objs = np.ascontiguousarray(objs, dtype=N.OBJECT_DTYPE)
        self._check(self._lib.rthip_scene_refit(self._ctx, objs.ctypes.data, len(objs)))
        self._objs = objs
        self.reset_accumulation()

    def refit_device_objects(self, device_ptr, n):
        """refit_scene for objects that live in device memory: `n` records of the object layout
        (88 bytes each) at `device_ptr` (a torch tensor's data_ptr(), say).  Scenes of 2 048
        objects and more are refitted without a copy to or from the host."""
        self._check(self._lib.rthip_scene_refit(self._ctx, C.c_void_p(int(device_ptr)), int(n)))
        self._objs = None
        self.reset_accumulation()

    def _raise(self, code):
        msg = self._lib.rthip_last_error(self._ctx)
        raise N.RthipError(code, msg.decode() if msg else "")

    def _check(self, code):
        if code != N.OK:
            self._raise(code)

    def close(self):
        if self._ctx:
            self._lib.rthip_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name, value):
        """Tuning / build knob by name (rthip_set_option in include/rthip.h)."""
        self._check(self._lib.rthip_set_option(self._ctx, name.encode(), float(value)))

    def set_stream(self, stream_handle):
        """Launch on a caller-owned hipStream_t; 0 = the context's own stream (which is NOT ordered
        against the default stream: see set_torch_stream)."""
        self._check(self._lib.rthip_set_stream(self._ctx, C.c_void_p(stream_handle)))

    def set_torch_stream(self, stream=None):
        """Launch on a torch stream (default: the current one), so that torch operations on that
        stream (zero_() of a caller-owned accumulation buffer, reads of it) and renders are ordered.
        torch's default stream has the handle 0, which rthip_set_stream reads as "own stream":
        it is addressed as hipStreamLegacy (the handle 1) instead."""
        import torch
        if stream is None:
            stream = torch.cuda.current_stream(self.device)
        self.set_stream(stream.cuda_stream or 1)

    def update_camera(self, camera):
        """Camera change: new pose, accumulation cleared, sample counter back to 0
        (raytracer_runtime/src/main.rs:101-103,140-149)."""
        self.camera = camera
        self._check(self._lib.rthip_camera_set(self._ctx, C.byref(camera.desc)))
        self.reset_accumulation()

    def reset_accumulation(self):
        self._check(self._lib.rthip_reset_accumulation(self._ctx))
        self.spp_done = 0

    def _params(self, spp_begin, spp_count, part=(0, 1), precision=None):
        p = N.RenderParams()
        p.spp_begin, p.spp_count, p.max_depth, p.seed = spp_begin, spp_count, self.max_depth, self.seed
        p.sky_mode, p.precision, p.variant = self.sky_mode, self.precision if precision is None else precision, self.variant
        p.part_index, p.part_count = part
        return p

    def accumulate(self, spp, part=(0, 1), spp_begin=None, sync=True, noalloc=False):
        """Adds `spp` samples per pixel (samples spp_begin .. spp_begin+spp of the stream).
        noalloc: through rthip_render_accumulate_noalloc, the entry point that allocates nothing."""
        begin = self.spp_done if spp_begin is None else spp_begin
        p = self._params(begin, spp, part)
        if noalloc:
            self._check(self._lib.rthip_render_accumulate_noalloc(self._ctx, C.byref(p)))
            if sync:
                self.synchronize()
        else:
            fn = self._lib.rthip_render_accumulate if sync else self._lib.rthip_render_accumulate_async
            self._check(fn(self._ctx, C.byref(p)))
        if spp_begin is None:
            self.spp_done += spp

    def synchronize(self):
        self._check(self._lib.rthip_synchronize(self._ctx))

    def validate_tree(self):
        """Downloads the tree the device holds and checks it (rthip_scene_validate): every box
        contains what is below it, every primitive is in exactly one leaf, depth as reported.
        Returns (n_nodes, depth); raises with the first violation otherwise."""
        bad, nodes, depth = C.c_int64(0), C.c_int32(0), C.c_int32(0)
        self._check(self._lib.rthip_scene_validate(self._ctx, C.byref(bad), C.byref(nodes), C.byref(depth)))
        return nodes.value, depth.value

    def stats(self):
        s = N.Stats()
        self._check(self._lib.rthip_get_stats(self._ctx, C.byref(s)))
        return {k: getattr(s, k) for k, _ in N.Stats._fields_ if k != "reserved"}

    def accum_buffer(self):
        ptr, n = C.c_void_p(), C.c_uint64()
        self._check(self._lib.rthip_accum_buffer(self._ctx, C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    def set_accum_buffer(self, device_ptr, n_words):
        self._check(self._lib.rthip_set_accum_buffer(self._ctx, C.c_void_p(device_ptr), n_words))

    # ---- multi-GPU merge behind the C ABI (rthip_comm_* / rthip_merge_* in include/rthip.h) ----
    @staticmethod
    def comm_unique_id():
        """128 bytes that rank 0 makes and hands to every other rank."""
        buf = (C.c_uint8 * 128)()
        N.check(N.lib().rthip_comm_unique_id(buf))
        return bytes(buf)

    def comm_init(self, unique_id, rank, world):
        """Collective over all ranks: this context becomes rank `rank` of `world` GPUs (RCCL)."""
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self._lib.rthip_comm_init(self._ctx, buf, int(rank), int(world)))

    def merge_reduce(self, root=0):
        """One ncclReduce (root >= 0) or ncclAllReduce (root = -1) of the accumulation buffer, in
        place, enqueued behind the renders on this context's stream."""
        self._check(self._lib.rthip_merge_reduce(self._ctx, int(root)))

    def merge_wait(self, device_ptr=0):
        """Orders this context's stream behind the reduces in flight on that buffer (0: any)."""
        self._check(self._lib.rthip_merge_wait(self._ctx, C.c_void_p(device_ptr)))

    def save_checkpoint(self, path):
        """Writes the accumulation words and the sample count to an .npz file; load_checkpoint on a
        renderer of the same scene, camera and seed resumes the render where it stopped (sums are
        integers, so the resumed image is bit-identical to an uninterrupted one)."""
        _, n = self.accum_buffer()
        words = np.zeros(n, dtype=np.uint64)
        self._check(self._lib.rthip_accum_download(self._ctx, words.ctypes.data, n))
        np.savez_compressed(path, words=words, spp_done=np.int64(self.spp_done), seed=np.int64(self.seed),
                            width=np.int64(self.camera.width), height=np.int64(self.camera.height))

    def load_checkpoint(self, path):
        with np.load(path) as z:
            if int(z["width"]) != self.camera.width or int(z["height"]) != self.camera.height:
                raise ValueError("checkpoint is for a %dx%d frame" % (int(z["width"]), int(z["height"])))
            if int(z["seed"]) != self.seed:
                raise ValueError("checkpoint was rendered with another seed: its samples would repeat or be skipped")
            words = np.ascontiguousarray(z["words"], dtype=np.uint64)
            self._check(self._lib.rthip_accum_upload(self._ctx, words.ctypes.data, len(words)))
            self.spp_done = int(z["spp_done"])

    def read_accum(self):
        """sum_s sqrt(sample), float64 [H, W, 3]."""
        out = np.zeros((self.camera.height, self.camera.width, 3), dtype=np.float64)
        self._check(self._lib.rthip_read_accum(self._ctx, out.ctypes.data))
        return out

    def read_rgb8(self, spp_total=None):
        out = np.zeros((self.camera.height, self.camera.width, 3), dtype=np.uint8)
        self._check(self._lib.rthip_read_rgb8(self._ctx, self.spp_done if spp_total is None else spp_total, out.ctypes.data))
        return out

    def read_xrgb32(self, spp_total=None):
        out = np.zeros((self.camera.height, self.camera.width), dtype=np.uint32)
        self._check(self._lib.rthip_read_xrgb32(self._ctx, self.spp_done if spp_total is None else spp_total, out.ctypes.data))
        return out

    def trace_primary(self, precision=None, jitter=False, sample=0):
        """Parity probe for BVH::trace: (id int32 [H,W], t [H,W], normal [H,W,3], front_face [H,W])."""
        h, w = self.camera.height, self.camera.width
        ids = np.zeros((h, w), dtype=np.int32)
        t = np.zeros((h, w), dtype=np.float64)
        nrm = np.zeros((h, w, 3), dtype=np.float64)
        front = np.zeros((h, w), dtype=np.uint8)
        self._check(self._lib.rthip_trace_primary(self._ctx, self.precision if precision is None else precision, int(jitter),
                                                  self.seed, sample, ids.ctypes.data, t.ctypes.data, nrm.ctypes.data,
                                                  front.ctypes.data))
        return ids, t, nrm, front

    def render_samples(self, spp, spp_begin=0, precision=None):
        """Linear radiance of every sample, float64 [spp, H, W, 3]; nothing is accumulated."""
        out = np.zeros((spp, self.camera.height, self.camera.width, 3), dtype=np.float64)
        p = self._params(spp_begin, spp, precision=precision)
        self._check(self._lib.rthip_render_samples(self._ctx, C.byref(p), out.ctypes.data))
        return out
