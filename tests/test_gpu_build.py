"""The device tree built ON the GPU (build option tree_gpu: Morton codes, radix sort, linear-BVH
hierarchy, bottom-up boxes) against the same bars as the host-built trees: structure checked on the
downloaded tree, f64 hits and whole paths bit for bit against the oracle (the closest hit does not
depend on the tree that finds it), f32 hit ids as the host-built tree finds them."""
import numpy as np
import pytest

from util import SCENES, camera_args

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scene", SCENES)
def test_gpu_built_tree_f64_bit_exact(rt, oracle, scene):
    w, h = 160, 90
    sc = rt.Scene.generate(scene)
    cam = camera_args(scene, w, h, rt)
    r = rt.Renderer(sc, rt.Camera(*cam), seed=11, precision=rt.F64, max_depth=12, options={"tree_gpu": 1, "gpu_leaf": 2, "ref_gpu": 1})
    host = rt.Renderer(sc, rt.Camera(*cam), seed=11, precision=rt.F64, max_depth=12)
    s = r.stats()
    if len(sc) <= 2:       # a scene of one leaf is built on the host whatever the option says
        assert s["n_nodes"] == host.stats()["n_nodes"]
    n_nodes, depth = r.validate_tree()
    assert n_nodes == s["n_nodes"] and depth == s["tree_depth"]
    osc = oracle.OracleScene(sc.objects)
    desc = oracle.camera_desc(*cam)
    ref = osc.trace_primary(desc, oracle.TIGHT, jitter=True, seed=11, sample=2)
    got = r.trace_primary(rt.F64, jitter=True, sample=2)
    for a, b in zip(ref, got):
        assert np.array_equal(a, b)
    _, smp, cnt, _ = osc.render(desc, 2, 12, 11, oracle.TIGHT, samples=True)
    assert np.array_equal(r.render_samples(2), smp) and r.stats()["rays"] == cnt["rays"]
    # f32: the same primary-hit ids as the host-built tree finds
    assert np.array_equal(r.trace_primary(rt.F32, jitter=True, sample=2)[0], host.trace_primary(rt.F32, jitter=True, sample=2)[0])
    r.close(); host.close()


@pytest.mark.parametrize("leaf", [1, 4, 7])
def test_gpu_built_tree_large_scene(rt, oracle, leaf):
    """200 000 objects (quantised nodes, scene in global memory): structure, f64 paths against the
    oracle, and the f32 accumulation words of the GPU-built and the host-built tree."""
    sc = rt.Scene.generate("synthetic:200000", 0x5EED0001)
    cam = ((0.0, 60.0, -550.0), (0.0, -60.0, 550.0), 40.0, 10.0, 0.0, 128, 72)
    r = rt.Renderer(sc, rt.Camera(*cam), seed=5, precision=rt.F64, max_depth=20, options={"tree_gpu": 1, "gpu_leaf": leaf})
    s = r.stats()
    assert s["n_nodes"] >= 200000 // leaf - 8 and s["scene_in_lds"] == 0
    r.validate_tree()
    osc = oracle.OracleScene(sc.objects)
    desc = oracle.camera_desc(*cam)
    ref = osc.trace_primary(desc, oracle.TIGHT, jitter=True, seed=5, sample=0)
    for a, b in zip(ref, r.trace_primary(rt.F64, jitter=True, sample=0)):
        assert np.array_equal(a, b)
    _, smp, cnt, _ = osc.render(desc, 2, 20, 5, oracle.TIGHT, samples=True)
    assert np.array_equal(r.render_samples(2), smp) and r.stats()["rays"] == cnt["rays"]
    g32 = r.trace_primary(rt.F32, jitter=True, sample=0)
    assert (g32[0] != ref[0]).mean() <= 2e-3
    print(f"[gpu build] leaf {leaf}: {s['n_nodes']} nodes, depth {s['tree_depth']}")
    r.close()


def test_gpu_build_is_repeatable_and_rebuilds(rt):
    """Two builds of one scene give the same words; a context can go from one scene to another and
    from the device build back to the host build."""
    cam = ((0.0, 60.0, -550.0), (0.0, -60.0, 550.0), 40.0, 10.0, 0.0, 256, 144)
    a = rt.Scene.generate("synthetic:50000", 1)
    b = rt.Scene.generate("synthetic:30000", 2)

    def words(r):
        r.reset_accumulation()
        r.accumulate(2, spp_begin=0)
        _, n = r.accum_buffer()
        w = np.zeros(n, dtype=np.uint64)
        r._check(r._lib.rthip_accum_download(r._ctx, w.ctypes.data, n))
        return w

    r = rt.Renderer(a, rt.Camera(*cam), seed=9, max_depth=10, options={"tree_gpu": 1})
    first = words(r)
    r.set_scene(b)
    other = words(r)
    r.set_scene(a)
    assert np.array_equal(words(r), first) and not np.array_equal(other, first)
    r.set_option("tree_gpu", 0)
    r.set_scene(a)
    r.validate_tree()
    host_words = words(r)
    # f32 paths may differ between two trees only where a hit lies within rounding of a box face: nearly all words agree
    assert (host_words != first).mean() < 1e-3
    r.close()


def _objects(rt, centres, radius, kinds=None):
    from raytracer_amd import _native as N
    n = len(centres)
    o = np.zeros(n, dtype=N.OBJECT_DTYPE)
    o["kind"] = N.SPHERE if kinds is None else kinds
    o["material"] = N.LAMBERTIAN
    o["center"] = centres
    o["extent"][:, 0] = radius
    o["extent"][:, 1] = radius
    o["extent"][:, 2] = radius
    o["albedo"] = 0.5
    return rt.Scene(o)


@pytest.mark.parametrize("case", ["same_centre", "on_a_line", "two_clusters", "hollow_shells"])
def test_gpu_build_degenerate_layouts(rt, oracle, case):
    """Layouts that give a Morton-code builder nothing to split on (equal codes fall back on the
    primitive's index), or everything on one axis, or two far clusters; and many gated spheres."""
    from raytracer_amd import _native as N
    g = np.random.default_rng(3)
    n = 700
    if case == "same_centre":       # concentric spheres of growing radius, seen from outside and (via glass) inside
        sc = _objects(rt, np.tile([0.0, 1.0, 0.0], (n, 1)), np.linspace(0.05, 3.0, n))
        sc.objects["material"][::3] = N.DIELECTRIC
        sc.objects["param"][::3] = 1.5
    elif case == "on_a_line":
        c = np.zeros((n, 3)); c[:, 0] = np.linspace(-40, 40, n); c[:, 1] = 1.0
        sc = _objects(rt, c, g.uniform(0.05, 0.3, n), kinds=np.where(np.arange(n) % 2 == 0, N.SPHERE, N.CUBE))
    elif case == "two_clusters":
        c = np.concatenate([g.normal(0, 0.5, (n // 2, 3)) + [-2000, 1, 0], g.normal(0, 0.5, (n - n // 2, 3)) + [3, 1, 0]])
        sc = _objects(rt, c, g.uniform(0.02, 0.1, n))
    else:
        from test_gpu_configs import hollow_shells
        sc = hollow_shells(rt, 4000)
    cam = ((13.0, 4.0, 9.0), (-13.0, -3.0, -9.0), 25.0, 10.0, 0.1, 160, 90)
    r = rt.Renderer(sc, rt.Camera(*cam), seed=21, precision=rt.F64, max_depth=8, options={"tree_gpu": 1, "ref_gpu": 1})
    r.validate_tree()
    osc = oracle.OracleScene(sc.objects)
    desc = oracle.camera_desc(*cam)
    ref = osc.trace_primary(desc, oracle.TIGHT, jitter=True, seed=21, sample=0)
    for a, b in zip(ref, r.trace_primary(rt.F64, jitter=True, sample=0)):
        assert np.array_equal(a, b)
    _, smp, cnt, _ = osc.render(desc, 2, 8, 21, oracle.TIGHT, samples=True)
    assert np.array_equal(r.render_samples(2), smp) and r.stats()["rays"] == cnt["rays"]
    r.close()


@pytest.mark.parametrize("scene", ["default", "glass", "hollow", "same_centre", "synthetic:70001", "five"])
def test_device_reference_pass_equals_host_pass(rt, scene):
    """Tie ranks and gates computed on the device (level-by-level segmented sorts, the default with
    tree_gpu) against the host's pass over the reference's tree (ref_gpu = 0): the same gate count, the
    same image word for word in both precisions, after an upload and after a refit."""
    from raytracer_amd import _native as N
    if scene == "hollow":
        from test_gpu_configs import hollow_shells
        sc = hollow_shells(rt, 2500)
    elif scene == "same_centre":     # every sort key equal: the order is what a stable sort leaves
        sc = _objects(rt, np.tile([0.0, 1.0, 0.0], (300, 1)), np.linspace(0.05, 3.0, 300))
        sc.objects["extent"][1::2, 0] *= -1.0
        sc.objects["material"][:] = N.DIELECTRIC
        sc.objects["param"][:] = 1.5
    elif scene == "five":            # one leaf: no tree for the device to build, ranks by one sort
        sc = rt.Scene(rt.Scene.generate("default").objects[1:6].copy())
    else:
        sc = rt.Scene.generate(scene, 0x5EED0001)
    cam = ((13.0, 4.0, 9.0), (-13.0, -3.0, -9.0), 25.0, 10.0, 0.1, 160, 90)
    g = np.random.default_rng(4)
    new = rt.Scene(sc.objects.copy())
    small = np.abs(new.objects["extent"]).max(axis=1) < 10.0
    new.objects["center"][small] += g.normal(0.0, 0.2, (int(small.sum()), 3))

    def images(r):
        out = []
        for precision in (rt.F32, rt.F64):
            r.precision = precision
            r.reset_accumulation()
            r.accumulate(2, spp_begin=0)
            _, n = r.accum_buffer()
            w = np.zeros(n, dtype=np.uint64)
            r._check(r._lib.rthip_accum_download(r._ctx, w.ctypes.data, n))
            out.append(w)
        return out

    dev = rt.Renderer(sc, rt.Camera(*cam), seed=2, max_depth=8, options={"tree_gpu": 1, "ref_gpu": 1})
    host = rt.Renderer(sc, rt.Camera(*cam), seed=2, max_depth=8, options={"tree_gpu": 1, "ref_gpu": 0})
    for step in range(2):
        a, b = dev.stats(), host.stats()
        assert (a["n_gates"], a["n_leaves"], a["n_prims"], a["n_nodes"]) == (b["n_gates"], b["n_leaves"], b["n_prims"], b["n_nodes"])
        dev.validate_tree()
        for x, y in zip(images(dev), images(host)):
            assert np.array_equal(x, y)
        if step == 0:
            dev.refit_scene(new)
            host.refit_scene(new)
    print(f"[device reference pass] {scene}: {a['n_gates']} gate boxes, {a['n_leaves']} reference leaves")
    dev.close(); host.close()


@pytest.mark.parametrize("spheres", [0, 1, 3])
@pytest.mark.parametrize("tree_gpu", [0, 1])
def test_large_scene_of_cubes_that_cannot_be_hit(rt, oracle, spheres, tree_gpu):
    """3 000 cubes with a negative half extent (they shape the reference's tree and are in no device
    tree) and at most three spheres: large enough for the device's reference pass, too few
    primitives for a device-built tree (or for any tree at all)."""
    from raytracer_amd import _native as N
    g = np.random.default_rng(8)
    n = 3000
    o = np.zeros(n + spheres, dtype=N.OBJECT_DTYPE)
    o["kind"][:n], o["material"][:n] = N.CUBE, N.LAMBERTIAN
    o["center"][:n] = g.uniform(-5, 5, (n, 3))
    o["extent"][:n] = g.uniform(0.05, 0.3, (n, 3)) * [1.0, -1.0, 1.0]
    o["albedo"][:] = 0.6
    o["kind"][n:], o["material"][n:] = N.SPHERE, N.METAL
    o["center"][n:] = np.array([(0.0, 1.0, 0.0), (1.5, 0.5, 0.5), (-1.0, 0.3, 1.0)])[:spheres]
    o["extent"][n:, 0] = np.array([1.0, 0.5, 0.3])[:spheres]
    sc = rt.Scene(o)
    cam = ((9.0, 4.0, 11.0), (-9.0, -3.5, -11.0), 40.0, 10.0, 0.2, 96, 54)
    r = rt.Renderer(sc, rt.Camera(*cam), seed=5, precision=rt.F64, max_depth=8, options={"tree_gpu": tree_gpu})
    assert r.stats()["n_prims"] == spheres
    r.validate_tree()
    osc = oracle.OracleScene(sc.objects)
    desc = oracle.camera_desc(*cam)
    _, smp, cnt, _ = osc.render(desc, 2, 8, 5, oracle.TIGHT, samples=True)
    assert np.array_equal(r.render_samples(2), smp) and r.stats()["rays"] == cnt["rays"]
    r.refit_scene(sc)
    assert np.array_equal(r.render_samples(2), smp)
    r.close()
