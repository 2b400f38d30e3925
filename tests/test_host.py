"""CPU tests of the product's host side: the C ABI loads and exports what include/rthip.h
declares, and the scene factories / Camera::new / BVH build agree with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from util import SCENES, camera_args

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(rt):
    from raytracer_amd import _native as N
    header = open(os.path.join(ROOT, "include", "rthip.h")).read()
    declared = set(re.findall(r"\b(rthip_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(N.SYMBOLS), (declared ^ set(N.SYMBOLS))
    lib = N.lib()
    for name in declared:
        assert getattr(lib, name) is not None


def test_struct_layouts(rt):
    from raytracer_amd import _native as N
    assert N.OBJECT_DTYPE.itemsize == 88 and N.OBJECT_DTYPE.fields["param"][1] == 80
    assert C.sizeof(N.CameraDesc) == 80 and C.sizeof(N.RenderParams) == 48 and C.sizeof(N.Stats) == 88


def test_product_does_not_import_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "raytracer_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".inc", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower(), f"{f} mentions the oracle"


def test_default_scene_structure(rt):  # src/scene.rs:122-205
    s = rt.Scene.create_sample_scene(seed=1).objects
    assert s[0]["kind"] == 1 and tuple(s[0]["center"]) == (0.0, -1000.0, 0.0) and tuple(s[0]["extent"]) == (1000.0,) * 3
    assert [int(k) for k in s["kind"][:7]] == [1, 0, 0, 0, 0, 0, 1]
    assert s[2]["extent"][0] == -0.98 and s[5]["extent"][0] == 7500.0 and s[5]["param"] == 60.0
    grid = s[7:]
    assert 380 <= len(grid) <= 484 + 80
    assert (grid["center"][:, 1] == 0.2).all()
    # the hole around (4, 0.2, 0): src/scene.rs:165
    d = np.linalg.norm(grid["center"] - np.array([4.0, 0.2, 0.0]), axis=1)
    assert (d > 0.9).all()
    lam, met, die = (grid["material"] == m for m in (0, 1, 2))
    assert (grid["param"][lam] < 0.35).all() and (grid["param"][lam] >= 0).all()      # emission = selector
    assert (grid["param"][met] < 0.3).all() and (grid["albedo"][met] >= 0.5).all()
    assert (grid["param"][die] == 1.5).all()
    # glass comes in pairs: outer size 0.2, inner -0.19, same centre, same kind
    idx = np.flatnonzero(die)
    assert len(idx) % 2 == 0
    for a, b in zip(idx[0::2], idx[1::2]):
        assert b == a + 1 and grid["extent"][a][0] == 0.2 and grid["extent"][b][0] == -0.19
        assert (grid["center"][a] == grid["center"][b]).all() and grid["kind"][a] == grid["kind"][b]
    frac = [lam.sum() / len(grid), met.sum() / len(grid)]
    assert 0.25 < frac[0] < 0.45 and 0.35 < frac[1] < 0.6


def test_scene_factories_are_seeded_and_deterministic(rt):
    a, b, c = (rt.Scene.generate("default", seed).objects for seed in (5, 5, 6))
    assert a.tobytes() == b.tobytes() and a.tobytes() != c.tobytes()


def test_sample_scene_is_spheres_then_cubes(rt):  # src/scenes.rs:5-14,201-300
    s = rt.Scene.generate("sample").objects
    k = s["kind"]
    first_cube = int(np.argmax(k == 1))
    assert (k[:first_cube] == 0).all() and (k[first_cube:] == 1).all()
    assert tuple(s[first_cube]["center"]) == (0.0, -100000.0, 0.0) and s[4]["param"] == 15.0
    assert set(np.unique(np.abs(s["extent"][7:first_cube, 0]))) <= {0.4, 0.38}
    spheres, cubes = rt.create_sample_scene()
    assert len(spheres) == first_cube and len(spheres) + len(cubes) == len(s)
    assert rt.combine_spheres_and_cubes(spheres, cubes).objects.tobytes() == s.tobytes()


def test_fixed_scenes(rt):  # src/scenes.rs:17-198
    for fn, n_s, n_c in ((rt.create_simple_scene, 5, 1), (rt.create_glass_scene, 11, 6), (rt.create_10_metallic_scene, 12, 1),
                         (rt.create_wallpaper_scene, 2, 1)):
        s, c = fn()
        assert (len(s), len(c)) == (n_s, n_c)
    s, c = rt.create_10_metallic_scene()
    assert [round(float(x), 1) for x in s["param"][2:]] == [i / 10 for i in range(10)]
    s, c = rt.create_wallpaper_scene()
    assert tuple(c[0]["extent"]) == (0.0, 0.0, 0.0) and s[1]["extent"][0] == -0.98


def test_synthetic_scene(rt):
    s = rt.create_synthetic_scene(5000, seed=3).objects
    assert len(s) == 5000 and 0.4 < (s["kind"] == 0).mean() < 0.6
    body = s[2:]
    assert (np.abs(body["center"][:, 0]) <= 500).all() and (body["center"][:, 1] >= 0.2).all()
    with pytest.raises(rt.RthipError):
        rt.Scene.generate("no-such-scene")


@pytest.mark.parametrize("scene", SCENES + ["synthetic:3000"])
def test_bvh_shape_matches_oracle(rt, oracle, scene):  # src/bvh.rs:17-81
    sc = rt.Scene.generate(scene)
    bvh = rt.BVH(sc)
    osc = oracle.OracleScene(sc.objects)
    ids, sizes = osc.leaf_order()
    assert np.array_equal(ids, bvh.leaf_ids) and np.array_equal(sizes, bvh.leaf_sizes)
    assert sorted(bvh.leaf_ids.tolist()) == list(range(len(sc)))
    assert bvh.leaf_sizes.min() >= 1 and bvh.leaf_sizes.max() <= 5                      # bvh.rs:52
    if len(sc) >= 6:
        assert bvh.leaf_sizes.min() >= 3                                                # 6 -> 3+3 ... 11 -> 5+(3+3)
    assert bvh.depth == max(1, osc.depth() - 1)


def test_empty_scene_is_an_error(rt):  # bvh.rs:51
    from raytracer_amd import _native as N
    with pytest.raises(ValueError):
        rt.BVH(rt.Scene())
    code = N.lib().rthip_bvh_leaf_order(None, 0, None, None, None, None)
    assert code == N.ERR_EMPTY


def test_invalid_objects_are_rejected(rt):
    from raytracer_amd import _native as N
    s = rt.Scene.generate("simple").objects.copy()
    s[1]["center"][0] = np.nan
    assert N.lib().rthip_bvh_leaf_order(s.ctypes.data, len(s), None, None, None, None) == N.ERR_INVALID
    s = rt.Scene.generate("simple").objects.copy()
    s[0]["material"] = 7
    assert N.lib().rthip_bvh_leaf_order(s.ctypes.data, len(s), None, None, None, None) == N.ERR_INVALID


@pytest.mark.parametrize("scene", SCENES)
def test_camera_fields_match_oracle_bit_for_bit(rt, oracle, scene):  # camera.rs:25-69
    for (w, h) in ((1920, 1080), (2560, 1440), (37, 23)):
        args = camera_args(scene, w, h, rt)
        f = rt.Camera(*args).fields()
        out = (C.c_double * 19)()
        cd = oracle.camera_desc(*args)
        oracle.lib().oracle_camera_fields(C.byref(cd), out)
        mine = np.concatenate([f[k] for k in ("origin", "pixel00_loc", "pixel_delta_x", "pixel_delta_y", "defocus_disk_x",
                                             "defocus_disk_y")] + [[f["defocus_angle"]]])
        assert np.array_equal(mine, np.array(out[:]))


def test_camera_rotation_matches_oracle(rt, oracle):  # camera.rs:95-103
    for yaw, pitch in ((0, 0), (-90, 0), (33.3, -20), (180, 89)):
        out = oracle.d3((0, 0, 0))
        oracle.lib().oracle_camera_rotation(yaw, pitch, out)
        assert rt.get_camera_rotation(yaw, pitch) == tuple(out)


def test_accum_word_packing_roundtrip():
    from util import pack_accum_words, unpack_accum_words
    rng = np.random.default_rng(0)
    img = rng.uniform(0, 50, (23, 37, 3))
    words = pack_accum_words(img, 37, 23)
    assert words.size == 5 * 3 * 64 * 3
    assert np.abs(unpack_accum_words(words, 37, 23) - img).max() <= 2.0 ** -37


@pytest.mark.parametrize("scene", SCENES + ["synthetic:3000", "synthetic:40000"])
def test_device_tree_is_conservative(rt, scene):
    """Every f32 child box (centre and half extent) and every quantised box contains the f32
    bounds of all primitives below it, every object that can be hit is in exactly one leaf:
    the invariant all culling rests on, checked on the host for each tree variant."""
    import ctypes as C
    from raytracer_amd import _native as N
    sc = rt.Scene.generate(scene)
    objs = sc.objects
    for env in ({}, {"RTHIP_TREE": "ref"}, {"RTHIP_NO_BIG_LEAF": "1"}, {"RTHIP_SAH_CPRIM": "1.0", "RTHIP_SAH_LEAF": "1"}):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            bad, nodes, depth = C.c_int64(-1), C.c_int32(0), C.c_int32(0)
            rc = N.lib().rthip_bvh_validate(objs.ctypes.data, len(objs), C.byref(bad), C.byref(nodes), C.byref(depth), None)
            assert rc == 0 and bad.value == 0, (scene, env, N.lib().rthip_last_error(None))
            assert nodes.value >= 1 and 1 <= depth.value <= 32
        finally:
            for k, v in old.items():
                if v is None: os.environ.pop(k, None)
                else: os.environ[k] = v


def test_threaded_build_equals_single_threaded(rt):
    """Ranges of 32 k objects and more are built on several threads (reference tree and surface-area
    tree alike); every array handed to the device is the same as from one thread."""
    import ctypes as C
    from raytracer_amd import _native as N
    objs = rt.create_synthetic_scene(150000, seed=7).objects
    sums = []
    for threads in ("1", "0"):
        os.environ["RTHIP_BUILD_THREADS"] = threads
        try:
            bad, h = C.c_int64(-1), C.c_uint64(0)
            rc = N.lib().rthip_bvh_validate(objs.ctypes.data, len(objs), C.byref(bad), None, None, C.byref(h))
            assert rc == 0 and bad.value == 0, N.lib().rthip_last_error(None)
            sums.append(h.value)
        finally:
            os.environ.pop("RTHIP_BUILD_THREADS", None)
    assert sums[0] == sums[1] and sums[0] != 0
