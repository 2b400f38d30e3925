"""The C ABI under misuse: every mistake a host can make comes back as a status code with a
message, never as a crash (the reference panics; a library under a foreign host must not:
include/rthip.h).  Called through ctypes with raw pointers on purpose."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_misuse_returns_status_codes(rt):
    from raytracer_amd import _native as N
    lib = N.lib()
    null = C.c_void_p()
    p = N.RenderParams(0, 1, 10, 1, 0, 0, 0, 0, 1)
    cam = rt.Camera((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6, 64, 36)
    st = N.Stats()
    # a NULL context
    for call in (lambda: lib.rthip_render_accumulate(null, C.byref(p)), lambda: lib.rthip_render_accumulate_async(null, C.byref(p)),
                 lambda: lib.rthip_render_accumulate_noalloc(null, C.byref(p)), lambda: lib.rthip_synchronize(null),
                 lambda: lib.rthip_camera_set(null, C.byref(cam.desc)), lambda: lib.rthip_scene_upload(null, None, 0),
                 lambda: lib.rthip_reset_accumulation(null), lambda: lib.rthip_get_stats(null, C.byref(st)),
                 lambda: lib.rthip_set_stream(null, None), lambda: lib.rthip_set_option(null, b"chunk", 4.0),
                 lambda: lib.rthip_merge_reduce(null, 0), lambda: lib.rthip_merge_wait(null, None),
                 lambda: lib.rthip_scene_validate(null, None, None, None), lambda: lib.rthip_read_accum(null, None)):
        assert call() == N.ERR_INVALID
    lib.rthip_destroy(null)   # a no-op
    ctx = C.c_void_p()
    assert lib.rthip_create(0, None) == N.ERR_INVALID
    N.check(lib.rthip_create(0, C.byref(ctx)))
    # before a scene or a camera exists
    ptr, words = C.c_void_p(), C.c_uint64()
    assert lib.rthip_accum_buffer(ctx, C.byref(ptr), C.byref(words)) == N.ERR_STATE
    assert lib.rthip_reset_accumulation(ctx) == N.ERR_STATE
    assert lib.rthip_scene_validate(ctx, None, None, None) == N.ERR_STATE
    assert lib.rthip_merge_reduce(ctx, 0) == N.ERR_STATE and b"communicator" in lib.rthip_last_error(ctx)
    assert lib.rthip_render_accumulate_noalloc(ctx, C.byref(p)) == N.ERR_STATE
    # scenes that are not scenes
    objs = rt.Scene.generate("simple").objects.copy()
    assert lib.rthip_scene_upload(ctx, objs.ctypes.data, -1) == N.ERR_INVALID
    assert lib.rthip_scene_upload(ctx, None, 3) == N.ERR_INVALID
    for field, value in (("kind", 5), ("material", 9), ("param", np.nan), ("center", np.inf)):
        bad = objs.copy()
        bad[field][1] = value
        assert lib.rthip_scene_upload(ctx, bad.ctypes.data, len(bad)) == N.ERR_INVALID
        assert b"object 1" in lib.rthip_last_error(ctx)
    N.check(lib.rthip_scene_upload(ctx, objs.ctypes.data, len(objs)), ctx)
    # cameras that are not cameras
    assert lib.rthip_camera_set(ctx, None) == N.ERR_INVALID
    for w, h, fov in ((0, 10, 20.0), (10, -1, 20.0), (1 << 15, 1 << 15, 20.0), (16, 16, float("nan"))):
        d = N.CameraDesc((C.c_double * 3)(0, 0, 0), (C.c_double * 3)(0, 0, -1), fov, 10.0, 0.0, w, h)
        assert lib.rthip_camera_set(ctx, C.byref(d)) == N.ERR_INVALID
    N.check(lib.rthip_camera_set(ctx, C.byref(cam.desc)), ctx)
    # parameters out of range
    assert lib.rthip_render_accumulate(ctx, None) == N.ERR_INVALID
    for bad in (N.RenderParams(0, 1, 10, 1, 2, 0, 0, 0, 1), N.RenderParams(0, 1, 10, 1, 0, 2, 0, 0, 1), N.RenderParams(0, 1, 10, 1, 0, 0, 2, 0, 1),
                N.RenderParams(0, 1, 10, 1, 0, 0, 0, -1, 1), N.RenderParams(0, 1, 10, 1, 0, 0, 0, 0, 0), N.RenderParams(0xFFFFFFFF, 2, 10, 1, 0, 0, 0, 0, 1)):
        assert lib.rthip_render_accumulate(ctx, C.byref(bad)) == N.ERR_INVALID
    reserved = N.RenderParams(0, 1, 10, 1, 0, 0, 0, 0, 1)
    reserved.reserved[1] = 1
    assert lib.rthip_render_accumulate(ctx, C.byref(reserved)) == N.ERR_INVALID
    wavefront = N.RenderParams(0, 1, 10, 1, 0, 0, N.WAVEFRONT, 0, 1)
    N.check(lib.rthip_accum_buffer(ctx, C.byref(ptr), C.byref(words)), ctx)
    assert lib.rthip_render_accumulate_noalloc(ctx, C.byref(wavefront)) == N.ERR_LIMIT
    deep = N.RenderParams(0, 1, 100000, 1, 0, 0, N.WAVEFRONT, 0, 1)
    assert lib.rthip_render_accumulate(ctx, C.byref(deep)) == N.ERR_LIMIT
    # buffers of the wrong shape
    assert lib.rthip_set_accum_buffer(ctx, ptr, words.value - 1) == N.ERR_INVALID
    assert lib.rthip_set_accum_buffer(ctx, C.c_void_p(ptr.value + 4), words.value) == N.ERR_INVALID
    host = np.zeros(words.value, dtype=np.uint64)
    assert lib.rthip_accum_download(ctx, host.ctypes.data, words.value + 1) == N.ERR_INVALID
    assert lib.rthip_accum_download(ctx, None, words.value) == N.ERR_INVALID
    rgb = np.zeros(64 * 36 * 3, dtype=np.uint8)
    assert lib.rthip_read_rgb8(ctx, 0, rgb.ctypes.data) == N.ERR_INVALID
    assert lib.rthip_read_rgb8(ctx, 1, None) == N.ERR_INVALID
    # options
    assert lib.rthip_set_option(ctx, None, 1.0) == N.ERR_INVALID
    assert lib.rthip_set_option(ctx, b"chunk", float("nan")) == N.ERR_INVALID
    assert lib.rthip_set_option(ctx, b"no_such_option", 1.0) == N.ERR_INVALID
    # communicator calls with bad arguments
    uid = (C.c_uint8 * 128)()
    assert lib.rthip_comm_init(ctx, None, 0, 1) == N.ERR_INVALID
    assert lib.rthip_comm_init(ctx, uid, 2, 2) == N.ERR_INVALID
    ctxs = (C.c_void_p * 2)(ctx, None)
    assert lib.rthip_merge_local(ctxs, 2, 0) != N.OK and lib.rthip_merge_local(None, 1, 0) == N.ERR_INVALID
    assert lib.rthip_merge_local(ctxs, 2, 5) == N.ERR_INVALID
    # and after all that the context still renders
    N.check(lib.rthip_render_accumulate(ctx, C.byref(p)), ctx)
    N.check(lib.rthip_get_stats(ctx, C.byref(st)), ctx)
    assert st.paths == 64 * 36
    lib.rthip_destroy(ctx)


def test_contexts_on_several_host_threads(rt):
    """A context belongs to one thread at a time, but contexts are independent: four host threads
    with a context each (different scenes, the ctypes calls run without the interpreter lock)
    produce the words a lone context produces."""
    import threading
    cam = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6, 160, 90)
    scenes = ["default", "sample", "glass", "synthetic:3000"]

    def words(name, out, k, rounds):
        r = rt.Renderer(rt.Scene.generate(name), rt.Camera(*cam), seed=4, max_depth=8)
        for _ in range(rounds):
            r.reset_accumulation()
            r.accumulate(3, spp_begin=0)
            r.accumulate(2, spp_begin=3, noalloc=(k % 2 == 0 and rounds == 1))   # the static scratch is one per GPU: only unshared use
        _, n = r.accum_buffer()
        w = np.zeros(n, dtype=np.uint64)
        r._check(r._lib.rthip_accum_download(r._ctx, w.ctypes.data, n))
        out[k] = w
        r.close()

    alone = [None] * len(scenes)
    for k, name in enumerate(scenes):
        words(name, alone, k, 1)
    together = [None] * len(scenes)
    threads = [threading.Thread(target=words, args=(name, together, k, 5)) for k, name in enumerate(scenes)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for a, b in zip(alone, together):
        assert b is not None and np.array_equal(a, b)
