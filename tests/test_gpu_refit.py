"""rthip_scene_refit: the same objects at new places, the device tree keeps its shape and gets its
boxes fitted again on the GPU.  The bars are those of a fresh upload: the downloaded tree is
sound, f64 hits and whole paths equal the oracle's on the MOVED scene bit for bit (tie ranks and
gates are recomputed for it), f32 hit ids as far as a fresh upload finds them."""
import ctypes as C

import numpy as np
import pytest

from util import SCENES, camera_args

pytestmark = pytest.mark.gpu


def moved(rt, scene, seed, spread, resize=0.2):
    """Every ordinary object somewhere else (each on its own: shells leave their cores), a bit larger
    or smaller, a fifth of them with another material; oversized ones (ground, sun) stay."""
    from raytracer_amd import _native as N
    g = np.random.default_rng(seed)
    o = scene.objects.copy()
    small = np.abs(o["extent"]).max(axis=1) < 10.0
    n = int(small.sum())
    o["center"][small] += g.normal(0.0, spread, (n, 3)) * [1.0, 0.15, 1.0]
    o["extent"][small] *= g.uniform(1.0 - resize, 1.0 + resize, (n, 1))
    swap = small & (g.random(len(o)) < 0.2)
    o["material"][swap] = N.METAL
    o["param"][swap] = 0.1
    return rt.Scene(o)


def words(r, spp=2):
    r.reset_accumulation()
    r.accumulate(spp, spp_begin=0)
    _, n = r.accum_buffer()
    w = np.zeros(n, dtype=np.uint64)
    r._check(r._lib.rthip_accum_download(r._ctx, w.ctypes.data, n))
    return w


def check_against_oracle(rt, oracle, r, sc, cam, seed, depth, f32_bound=2e-4):
    osc = oracle.OracleScene(sc.objects)
    desc = oracle.camera_desc(*cam)
    ref = osc.trace_primary(desc, oracle.TIGHT, jitter=True, seed=seed, sample=1)
    for a, b in zip(ref, r.trace_primary(rt.F64, jitter=True, sample=1)):
        assert np.array_equal(a, b)
    _, smp, cnt, _ = osc.render(desc, 2, depth, seed, oracle.TIGHT, samples=True)
    assert np.array_equal(r.render_samples(2), smp) and r.stats()["rays"] == cnt["rays"]
    assert (r.trace_primary(rt.F32, jitter=True, sample=1)[0] != ref[0]).mean() <= f32_bound


@pytest.mark.parametrize("ref_gpu", [0, 1])
@pytest.mark.parametrize("scene", SCENES + ["one", "nothing_to_hit"])
def test_refit_where_the_objects_are_changes_nothing(rt, scene, ref_gpu):
    from raytracer_amd import _native as N
    if scene == "one":
        sc = rt.Scene(rt.Scene.generate("simple").objects[:1].copy())
    elif scene == "nothing_to_hit":      # a cube with a negative half extent: in the reference's tree, never hit
        sc = rt.Scene(rt.Scene.generate("simple").objects[:1].copy())
        sc.objects["kind"][0] = N.CUBE
        sc.objects["extent"][0] = (0.5, 0.5, -0.2)
    else:
        sc = rt.Scene.generate(scene)
    cam = camera_args(scene if scene in SCENES else "simple", 160, 90, rt)
    r = rt.Renderer(sc, rt.Camera(*cam), seed=3, max_depth=10, options={"ref_gpu": ref_gpu})
    r.set_option("count", 1)
    before = words(r)
    tests_before = r.stats()["box_tests"]
    shape = r.validate_tree()
    r.refit_scene(sc)
    assert r.validate_tree() == shape
    # the same boxes: the same image and the same number of box tests on the way to it
    assert np.array_equal(words(r), before) and r.stats()["box_tests"] == tests_before
    r.close()


@pytest.mark.parametrize("tree", ["host", "gpu", "ref", "host-hostpass"])
@pytest.mark.parametrize("scene", ["default", "sample", "glass"])
def test_refit_moved_objects_f64_bit_exact(rt, oracle, scene, tree):
    """(ref_gpu: ranks, gates and primitive records made on the device, as scenes of 2048 objects
    and more have them by default, or by the host's pass over the reference's tree.)"""
    sc = rt.Scene.generate(scene)
    cam = camera_args(scene, 160, 90, rt)
    options = {"host": {"ref_gpu": 1}, "gpu": {"tree_gpu": 1, "ref_gpu": 1}, "ref": {"tree_ref": 1, "ref_gpu": 1}, "host-hostpass": {"ref_gpu": 0}}[tree]
    r = rt.Renderer(sc, rt.Camera(*cam), seed=17, precision=rt.F64, max_depth=12, options=options)
    shape = r.validate_tree()
    for step, spread in enumerate((0.05, 0.6)):   # a nudge, then a shuffle
        sc = moved(rt, sc, 100 + step, spread)
        r.refit_scene(sc)
        assert r.validate_tree() == shape
        check_against_oracle(rt, oracle, r, sc, cam, 17, 12)
    r.close()


@pytest.mark.parametrize("tree", ["host", "gpu", "host-hostpass"])
def test_refit_large_scene_in_global_memory(rt, oracle, tree):
    """200 000 objects (quantised nodes).  One object leaves the old bounds by far, so the grid of
    the quantised nodes has to be laid out again."""
    sc = rt.Scene.generate("synthetic:200000", 0x5EED0001)
    cam = ((0.0, 60.0, -550.0), (0.0, -60.0, 550.0), 40.0, 10.0, 0.0, 128, 72)
    options = {"host": {}, "gpu": {"tree_gpu": 1}, "host-hostpass": {"ref_gpu": 0}}[tree]
    r = rt.Renderer(sc, rt.Camera(*cam), seed=5, precision=rt.F64, max_depth=20, options=options)
    shape = r.validate_tree()
    assert r.stats()["scene_in_lds"] == 0
    new = moved(rt, sc, 7, 0.5)
    new.objects["center"][1234] += [2500.0, 0.0, -1800.0]
    r.refit_scene(new)
    assert r.validate_tree() == shape
    check_against_oracle(rt, oracle, r, new, cam, 5, 20, f32_bound=2e-3)
    # and back: the first image again
    first = rt.Renderer(sc, rt.Camera(*cam), seed=5, max_depth=20, options=options)
    r.refit_scene(sc)
    r.precision = rt.F32
    assert np.array_equal(words(r), words(first))
    r.close(); first.close()


def test_refit_with_another_number_of_gate_boxes(rt, oracle):
    """Hollow shells whose cores wander off: the gate array grows, so everything behind the nodes
    is laid out again on the device."""
    from test_gpu_configs import hollow_shells
    sc = hollow_shells(rt, 3000)
    cam = ((13.0, 4.0, 9.0), (-13.0, -3.0, -9.0), 25.0, 10.0, 0.1, 160, 90)
    r = rt.Renderer(sc, rt.Camera(*cam), seed=21, precision=rt.F64, max_depth=8, options={"ref_gpu": 1})
    gates = r.stats()["n_gates"]
    new = moved(rt, sc, 9, 0.3, resize=0.0)
    r.refit_scene(new)
    r.validate_tree()
    print(f"[refit] gate boxes {gates} -> {r.stats()['n_gates']}")
    assert r.stats()["n_gates"] != gates
    check_against_oracle(rt, oracle, r, new, cam, 21, 8, f32_bound=2e-3)
    r.close()


@pytest.mark.parametrize("ref_gpu", [0, 1])
def test_refit_turns_down_other_objects(rt, ref_gpu):
    from raytracer_amd import _native as N
    sc = rt.Scene.generate("default")
    cam = camera_args("default", 96, 54, rt)
    r = rt.Renderer(sc, rt.Camera(*cam), seed=1, options={"ref_gpu": ref_gpu})
    before = words(r)
    cube = int(np.flatnonzero(sc.objects["kind"] == N.CUBE)[1])
    sphere = int(np.flatnonzero(sc.objects["kind"] == N.SPHERE)[5])
    fewer = rt.Scene(sc.objects[:-1].copy())
    swapped = rt.Scene(sc.objects.copy())
    swapped.objects["kind"][sphere] = N.CUBE
    unhittable = rt.Scene(sc.objects.copy())
    unhittable.objects["extent"][cube, 1] *= -1.0
    not_finite = rt.Scene(sc.objects.copy())
    not_finite.objects["center"][3, 0] = np.nan
    for bad in (fewer, swapped, unhittable, not_finite):
        with pytest.raises(N.RthipError) as e:
            r.refit_scene(bad)
        assert e.value.code == N.ERR_INVALID
        assert np.array_equal(words(r), before)   # the scene on the device is as it was
    r.close()
    # nothing to refit
    lib = N.lib()
    ctx = C.c_void_p()
    N.check(lib.rthip_create(0, C.byref(ctx)))
    objs = np.ascontiguousarray(sc.objects, dtype=N.OBJECT_DTYPE)
    assert lib.rthip_scene_refit(ctx, objs.ctypes.data, len(objs)) == N.ERR_STATE
    assert lib.rthip_scene_refit(None, objs.ctypes.data, len(objs)) == N.ERR_INVALID
    lib.rthip_destroy(ctx)


@pytest.mark.parametrize("scene", ["default", "synthetic:30000"])
def test_refit_and_upload_from_device_memory(rt, scene):
    """Objects handed over in device memory (a torch tensor): the same images as from host memory,
    for a scene the device's pass handles without touching the host and for one small enough to go
    through the host's pass; invalid objects are found on the device too."""
    import torch
    from raytracer_amd import _native as N
    sc = rt.Scene.generate(scene, 0x5EED0001)
    cam = camera_args("default", 160, 90, rt) if scene == "default" else ((0.0, 60.0, -550.0), (0.0, -60.0, 550.0), 40.0, 10.0, 0.0, 160, 90)
    new = moved(rt, sc, 3, 0.2)

    def on_device(objs):
        raw = np.frombuffer(np.ascontiguousarray(objs, dtype=N.OBJECT_DTYPE).tobytes(), dtype=np.uint8)
        return torch.from_numpy(raw.copy()).cuda()

    host = rt.Renderer(sc, rt.Camera(*cam), seed=4, max_depth=8)
    host.refit_scene(new)
    want = words(host)
    dev = rt.Renderer(sc, rt.Camera(*cam), seed=4, max_depth=8)
    t = on_device(new.objects)
    dev.refit_device_objects(t.data_ptr(), len(new))
    dev.validate_tree()
    assert np.array_equal(words(dev), want)
    # an upload from device memory
    lib = N.lib()
    N.check(lib.rthip_scene_upload(dev._ctx, C.c_void_p(t.data_ptr()), len(new)), dev._ctx)
    fresh = rt.Renderer(new, rt.Camera(*cam), seed=4, max_depth=8)
    assert np.array_equal(words(dev), words(fresh))
    # a NaN among the objects on the device
    bad = new.objects.copy()
    bad["center"][len(bad) // 2, 1] = np.nan
    tb = on_device(bad)
    before = words(dev)
    with pytest.raises(N.RthipError) as e:
        dev.refit_device_objects(tb.data_ptr(), len(bad))
    assert e.value.code == N.ERR_INVALID and str(len(bad) // 2) in str(e.value)
    assert np.array_equal(words(dev), before)
    host.close(); dev.close(); fresh.close()
