"""Known-answer tests that pin the CPU oracle to the reference's arithmetic (closed forms
derived from the cited lines of /root/reference; the reference itself ships no tests)."""
import ctypes as C
import math

import numpy as np
import pytest

from util import SCENES, camera_args


def hit(ob, kind, c, e, o, d):
    t = C.c_double(0)
    h = ob.lib().oracle_object_hit(kind, ob.d3(c), ob.d3(e), ob.d3(o), ob.d3(d), C.byref(t))
    return bool(h), t.value


def normal(ob, kind, c, e, p):
    n = ob.d3((0, 0, 0))
    ob.lib().oracle_object_normal(kind, ob.d3(c), ob.d3(e), ob.d3(p), n)
    return tuple(n)


def test_sphere_head_on(oracle):  # sphere.rs:26-48: origin (0,0,-5), dir +z, unit sphere -> t = 4
    h, t = hit(oracle, 0, (0, 0, 0), (1, 0, 0), (0, 0, -5), (0, 0, 1))
    assert h and t == 4.0
    assert normal(oracle, 0, (0, 0, 0), (1, 0, 0), (0, 0, -1)) == (0.0, 0.0, -1.0)


def test_sphere_unnormalised_direction(oracle):  # dir of length 2 halves t (a = d.d is kept)
    h, t = hit(oracle, 0, (0, 0, 0), (1, 0, 0), (0, 0, -5), (0, 0, 2))
    assert h and t == 2.0


def test_sphere_tangent_is_a_hit(oracle):  # discriminant == 0 is not < 0 (sphere.rs:42)
    h, t = hit(oracle, 0, (0, 0, 0), (1, 0, 0), (1, 0, -5), (0, 0, 1))
    assert h and t == 5.0


def test_sphere_miss(oracle):
    h, _ = hit(oracle, 0, (0, 0, 0), (1, 0, 0), (1.5, 0, -5), (0, 0, 1))
    assert not h


def test_sphere_from_inside_returns_negative_near_root(oracle):
    # near root only (sphere.rs:47): from the centre t = -1, which scene.rs:105 (t > 0.001) rejects
    h, t = hit(oracle, 0, (0, 0, 0), (1, 0, 0), (0, 0, 0), (0, 0, 1))
    assert h and t == -1.0


def test_negative_radius_same_t_inverted_normal(oracle):  # sphere.rs:61: (p - c) / r with r < 0
    h, t = hit(oracle, 0, (0, 0, 0), (-1, 0, 0), (0, 0, -5), (0, 0, 1))
    assert h and t == 4.0
    assert normal(oracle, 0, (0, 0, 0), (-1, 0, 0), (0, 0, -1)) == (-0.0, -0.0, 1.0)


def test_negative_radius_aabb_is_inverted(oracle):  # sphere.rs:13-19
    mn, mx = oracle.d3((0, 0, 0)), oracle.d3((0, 0, 0))
    oracle.lib().oracle_object_aabb(0, oracle.d3((1, 2, 3)), oracle.d3((-0.5, 0, 0)), mn, mx)
    assert tuple(mn) == (1.5, 2.5, 3.5) and tuple(mx) == (0.5, 1.5, 2.5)


def test_cube_from_outside(oracle):  # cube.rs:24-42
    h, t = hit(oracle, 1, (0, 0, 0), (1, 1, 1), (0, 0, -5), (0, 0, 1))
    assert h and t == 4.0
    h, t = hit(oracle, 1, (0, 0, 0), (1, 2, 3), (0.25, 0.5, -5), (0, 0, 0.5))
    assert h and t == 4.0


def test_cube_from_inside_is_a_miss(oracle):  # cube.rs:37: t_n <= 0 -> None
    h, _ = hit(oracle, 1, (0, 0, 0), (1, 1, 1), (0.1, 0.2, 0.3), (0.3, 0.2, 1))
    assert not h


def test_cube_behind_is_a_miss(oracle):  # t_f <= 0
    h, _ = hit(oracle, 1, (0, 0, 0), (1, 1, 1), (0, 0, 5), (0.01, 0.02, 1))
    assert not h


def test_negative_half_extent_never_hits(oracle):  # cube.rs:29-37: t1 > t2 on every axis
    rng = np.random.default_rng(1)
    for _ in range(200):
        o = rng.uniform(-3, 3, 3)
        d = rng.uniform(-1, 1, 3)
        h, _ = hit(oracle, 1, (0, 0, 0), (-0.5, -0.5, -0.5), o, d)
        assert not h


def test_cube_normal_axis_order_and_signum(oracle):  # cube.rs:54-64
    assert normal(oracle, 1, (0, 0, 0), (1, 1, 1), (1, 0.2, 0.3)) == (1.0, 0.0, 0.0)
    assert normal(oracle, 1, (0, 0, 0), (1, 1, 1), (0.2, -1, 0.3)) == (0.0, -1.0, 0.0)
    assert normal(oracle, 1, (0, 0, 0), (1, 1, 1), (0.2, 0.3, -1)) == (0.0, 0.0, -1.0)
    # equal |x| and |y|: x is tested first
    assert normal(oracle, 1, (0, 0, 0), (1, 1, 1), (-1, 1, 0)) == (-1.0, 0.0, 0.0)
    # Rust signum(+0.0) = +1, signum(-0.0) = -1; the axis rule ignores the half extents
    assert normal(oracle, 1, (0, 0, 0), (1, 1, 1), (0.0, 0.0, 0.0)) == (1.0, 0.0, 0.0)
    assert normal(oracle, 1, (0, 0, 0), (1, 5, 1), (0.9, 1.0, 0.1)) == (0.0, 1.0, 0.0)


def aabb(ob, mn, mx, o, d):
    return bool(ob.lib().oracle_aabb_hit(ob.d3(mn), ob.d3(mx), ob.d3(o), ob.d3(d)))


def test_reference_aabb_is_per_axis_only(oracle):  # math.rs:356-371
    box = ((-1, -1, -1), (1, 1, 1))
    assert aabb(oracle, *box, (0, 0, -5), (0.1, 0.1, 1))
    # a ray that passes far beside the box still "hits": the interval is not carried across axes
    assert aabb(oracle, *box, (-5, -5, -5), (1, 2, 3))
    assert not oracle.lib().oracle_slab_hit(oracle.d3(box[0]), oracle.d3(box[1]), oracle.d3((-5, -5, -5)), oracle.d3((1, 2, 3)),
                                            1e300, C.byref(C.c_double()))
    # wholly behind the origin on one axis -> rejected
    assert not aabb(oracle, *box, (0, 0, 5), (0.1, 0.1, 1))
    # far plane exactly 0.001 ahead: max <= min -> rejected; slightly more -> accepted (eps of :364)
    assert not aabb(oracle, (-1, -1, -1), (1, 1, 1), (0, 0, 0), (1, 1, 1000))
    assert aabb(oracle, (-1, -1, -1), (1, 1, 1), (0, 0, 0), (1, 1, 999))
    # an inverted box is rejected on every axis where the ray could enter it
    assert not aabb(oracle, (1, 1, 1), (-1, -1, -1), (0, 0, -5), (0.1, 0.1, 1))


def test_schlick(oracle):  # materials.rs:80-85
    L = oracle.lib()
    assert L.oracle_reflectance(1.0, 1.5, 1) == pytest.approx(0.04, abs=1e-15)
    assert L.oracle_reflectance(0.0, 1.5, 1) == 1.0
    for c in np.linspace(0, 1, 33):
        a, b = L.oracle_reflectance(float(c), 1.0 / 1.5, 1), L.oracle_reflectance(float(c), 1.0 / 1.5, 0)
        assert abs(a - b) <= 4 * np.spacing(a)  # powf(5.) vs the multiply chain the kernels use


def test_reflect_refract(oracle):  # math.rs:97-107
    L = oracle.lib()
    out = oracle.d3((0, 0, 0))
    L.oracle_reflect(oracle.d3((1, -1, 0)), oracle.d3((0, 1, 0)), out)
    assert tuple(out) == (1.0, 1.0, 0.0)
    # straight through at eta = 1
    v = np.array([0.6, -0.8, 0.0])
    L.oracle_refract(oracle.d3(v), oracle.d3((0, 1, 0)), 1.0, out)
    assert np.allclose(tuple(out), v, atol=1e-15)
    # Snell: sin(t) = eta sin(i)
    L.oracle_refract(oracle.d3(v), oracle.d3((0, 1, 0)), 1 / 1.5, out)
    o = np.array(tuple(out))
    assert abs(np.linalg.norm(o) - 1) < 1e-15 and abs(o[0] - 0.6 / 1.5) < 1e-15 and o[1] < 0


def test_sky(oracle):  # scene.rs:116-120
    out = oracle.d3((0, 0, 0))
    oracle.lib().oracle_sky(oracle.d3((0, 1, 0)), 0, out)
    assert tuple(out) == (0.5, 0.7, 1.0)
    oracle.lib().oracle_sky(oracle.d3((0, -2, 0)), 0, out)
    assert tuple(out) == (1.0, 1.0, 1.0)
    oracle.lib().oracle_sky(oracle.d3((3, 0, 0)), 0, out)
    assert np.allclose(tuple(out), (0.75, 0.85, 1.0), atol=1e-15)


def test_emission(oracle):  # materials.rs:71-77
    out = oracle.d3((0, 0, 0))
    oracle.lib().oracle_emission(0, oracle.d3((0.8, 0.4, 0.2)), 60.0, out)
    assert tuple(out) == (0.8 * 60.0, 0.4 * 60.0, 0.2 * 60.0)
    for m in (1, 2):
        oracle.lib().oracle_emission(m, oracle.d3((0.8, 0.4, 0.2)), 60.0, out)
        assert tuple(out) == (0.0, 0.0, 0.0)


def scatter(ob, mat, albedo, param, din, n, front, seed=1, pixel=2, sample=3):
    att, dout, draws = ob.d3((0, 0, 0)), ob.d3((0, 0, 0)), C.c_int32(0)
    ok = ob.lib().oracle_scatter(mat, ob.d3(albedo), param, ob.d3(din), ob.d3((0, 0, 0)), ob.d3(n), int(front), seed, pixel,
                                 sample, 1, att, dout, C.byref(draws))
    return bool(ok), np.array(tuple(att)), np.array(tuple(dout)), draws.value


def test_scatter_draw_counts_and_rules(oracle):  # materials.rs:17-68
    ok, att, d, n = scatter(oracle, 0, (0.1, 0.2, 0.3), 5.0, (0, -1, 0), (0, 1, 0), True)
    assert ok and n == 3 and tuple(att) == (0.1, 0.2, 0.3)
    assert abs(np.linalg.norm(d - np.array([0, 1, 0])) - 1) < 1e-15  # normal + unit vector, not normalised
    ok, att, d, n = scatter(oracle, 1, (0.7, 0.6, 0.5), 0.0, (1, -1, 0), (0, 1, 0), True)
    assert ok and n == 3  # three draws even with fuzz 0
    assert np.allclose(d, np.array([1, 1, 0]) / math.sqrt(2), atol=1e-15)
    ok, _, _, n = scatter(oracle, 1, (0.7, 0.6, 0.5), 0.0, (1, 1, 0), (0, 1, 0), True)
    assert not ok and n == 3  # reflected below the surface -> None
    # dielectric from inside at a grazing angle: total internal reflection, no draw
    ok, att, d, n = scatter(oracle, 2, (0, 0, 0), 1.5, (1, -0.1, 0), (0, 1, 0), False)
    assert ok and n == 0 and tuple(att) == (1.0, 1.0, 1.0) and d[1] > 0
    # head on: can refract, one draw
    ok, _, d, n = scatter(oracle, 2, (0, 0, 0), 1.5, (0, -1, 0), (0, 1, 0), True)
    assert ok and n == 1


def test_unit_vector_is_cube_sample_normalised(oracle):  # math.rs:156-168
    w = np.zeros(3, dtype=np.uint32)
    oracle.lib().oracle_rng_words(1, 2, 3, 3, w.ctypes.data)
    p = -1.0 + 2.0 * (w.astype(np.float64) / 2.0 ** 32)
    _, _, d, _ = scatter(oracle, 0, (1, 1, 1), 0.0, (0, -1, 0), (0, 1, 0), True)
    expect = np.array([0, 1, 0]) + p / math.sqrt((p * p).sum())
    assert np.allclose(d, expect, atol=1e-15)


def test_rng_known_words(oracle):
    """The stream specification (PCG RXS-M-XS 32 keyed by hashed seed/pixel/sample), restated in Python."""
    def mix32(x):
        x ^= x >> 16; x = (x * 0x7feb352d) & 0xFFFFFFFF; x ^= x >> 15; x = (x * 0x846ca68b) & 0xFFFFFFFF; x ^= x >> 16
        return x

    def words(seed, pixel, sample, n):
        a = mix32((seed + 0x9E3779B9 * (pixel + 1)) & 0xFFFFFFFF)
        b = mix32(a ^ ((0x85EBCA6B * (sample + 1)) & 0xFFFFFFFF))
        state, inc = b, ((mix32((b + a + 0xC2B2AE35) & 0xFFFFFFFF) << 1) | 1) & 0xFFFFFFFF
        out = []
        for _ in range(n):
            old = state
            state = (old * 747796405 + inc) & 0xFFFFFFFF
            word = (((old >> ((old >> 28) + 4)) ^ old) * 277803737) & 0xFFFFFFFF
            out.append((word >> 22) ^ word)
        return out

    for key in [(0, 0, 0), (0x5EED0001, 12345, 7), (0xFFFFFFFF, 0xFFFFFFFF, 0xFFFFFFFF)]:
        w = np.zeros(16, dtype=np.uint32)
        oracle.lib().oracle_rng_words(*key, 16, w.ctypes.data)
        assert list(w) == words(*key, 16)
    # uniformity smoke: mean of many first draws
    first = np.zeros(1, dtype=np.uint32)
    acc = []
    for p in range(2000):
        oracle.lib().oracle_rng_words(9, p, 0, 1, first.ctypes.data)
        acc.append(first[0] / 2.0 ** 32)
    assert abs(np.mean(acc) - 0.5) < 0.03


def test_camera_fields_closed_form(oracle):  # camera.rs:25-69
    cd = oracle.camera_desc((0, 0, 0), (0, 0, -1), 90, 1, 0, 4, 2)
    out = (C.c_double * 19)()
    oracle.lib().oracle_camera_fields(C.byref(cd), out)
    v = np.array(out[:18]).reshape(6, 3)
    # fov 90, focus 1: viewport 2 high, 4 wide; u = +x, v = +y; pixel (0,0) centre at (-1.5, 0.5, -1)
    assert np.allclose(v[2], (1.0, 0, 0), atol=1e-15) and np.allclose(v[3], (0, -1.0, 0), atol=1e-15)
    assert np.allclose(v[1], (-1.5, 0.5, -1.0), atol=1e-15)
    assert np.allclose(v[4], 0) and np.allclose(v[5], 0) and out[18] == 0.0


def test_camera_ray_draws_and_defocus_product(oracle):  # camera.rs:72-92
    cd = oracle.camera_desc((13, 1.5, 3), (-13, -1.5, -3), 20, 10, 0.6, 64, 36)
    o, d = oracle.d3((0, 0, 0)), oracle.d3((0, 0, 0))
    oracle.lib().oracle_camera_ray(C.byref(cd), 5, 6, 1, 2, 1, o, d)
    w = np.zeros(4, dtype=np.uint32)
    oracle.lib().oracle_rng_words(1, 6 * 64 + 5, 2, 4, w.ctypes.data)
    u = w.astype(np.float64) / 2.0 ** 32
    f = (C.c_double * 19)()
    oracle.lib().oracle_camera_fields(C.byref(cd), f)
    v = np.array(f[:18]).reshape(6, 3)
    p = np.array([-1 + 2 * u[2], -1 + 2 * u[3], 0.0])
    p = p / math.sqrt((p * p).sum())
    origin = v[0] + (p[0] * v[4]) * (p[1] * v[5])  # component-wise PRODUCT, as camera.rs:80
    assert np.allclose(tuple(o), origin, atol=1e-15)
    target = v[1] + v[2] * 5 + v[3] * 6 + ((-0.5 + u[0]) * v[2] + (-0.5 + u[1]) * v[3])
    dd = target - origin
    assert np.allclose(tuple(d), dd / np.linalg.norm(dd), atol=1e-14)
    # defocus off: origin is the camera position and only two draws are used
    cd0 = oracle.camera_desc((13, 1.5, 3), (-13, -1.5, -3), 20, 10, 0.0, 64, 36)
    oracle.lib().oracle_camera_ray(C.byref(cd0), 5, 6, 1, 2, 1, o, d)
    assert tuple(o) == (13.0, 1.5, 3.0)


def test_camera_rotation(oracle):  # camera.rs:95-103
    out = oracle.d3((0, 0, 0))
    oracle.lib().oracle_camera_rotation(-90.0, 0.0, out)
    assert abs(out[0]) < 1e-15 and out[1] == 0.0 and out[2] == -1.0


def test_quantize_is_rust_as_u8(oracle):  # raytracer_headless/src/main.rs:6-12
    v = np.array([0.0, 0.5, 1.0, 1.2, -0.3, np.nan, 0.999999, 1.0 / 255.0, np.inf, 254.9999 / 255.0])
    q = oracle.quantize_rgb8(v)
    assert list(q) == [0, 127, 255, 255, 0, 0, 254, 1, 255, 254]


@pytest.mark.parametrize("scene", SCENES)
def test_faithful_equals_tight(oracle, rt, scene):
    """The reference's own traversal and the culled, ordered traversal the GPU implements
    return the same hits and the same radiance (to summation order) on every scene."""
    sc = rt.Scene.generate(scene)
    osc = oracle.OracleScene(sc.objects)
    cam = oracle.camera_desc(*camera_args(scene, 96, 54, rt))
    for jitter in (False, True):
        a = osc.trace_primary(cam, oracle.FAITHFUL, jitter=jitter, seed=3, sample=1)
        b = osc.trace_primary(cam, oracle.TIGHT, jitter=jitter, seed=3, sample=1)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    depth = 10 if scene != "sample" else 50
    _, sf, cf, _ = osc.render(cam, 2, depth, 11, oracle.FAITHFUL, samples=True)
    _, st, ct, _ = osc.render(cam, 2, depth, 11, oracle.TIGHT, samples=True)
    assert cf["rays"] == ct["rays"] and cf["rng_draws"] == ct["rng_draws"]
    assert np.allclose(sf, st, rtol=1e-12, atol=1e-300)
    assert ct["box_tests"] < cf["box_tests"] or len(sc) < 20


def test_ray_color_quirks(oracle, rt):
    """renderer.rs:3-26: the sky term carries the SQUARED throughput; depth 0 is black;
    an absorbed metal ray returns black."""
    # one grey Lambertian ground slab under the sky: a path that hits the slab once and then
    # the sky returns albedo * (sky * albedo)
    s = rt.Scene()
    s.add_cube((0, -1, 0), (1000, 1, 1000), rt.Material.lambertian((0.5, 0.5, 0.5), 0.0))
    osc = oracle.OracleScene(s.objects)
    cam = oracle.camera_desc((0, 5, 0), (0, -1, 1e-3), 40, 5, 0.0, 8, 8)
    _, smp, cnt, _ = osc.render(cam, 8, 2, 5, oracle.FAITHFUL, samples=True)
    # with depth 2 every path is slab -> sky (or slab -> slab is impossible from above)
    assert cnt["rays"] == 2 * cnt["paths"]
    lo = 0.5 * 0.5 * np.array([0.5, 0.7, 1.0])   # darkest sky (straight up) x albedo^2
    hi = 0.5 * 0.5 * np.array([1.0, 1.0, 1.0])
    assert (smp >= lo - 1e-12).all() and (smp <= hi + 1e-12).all()
    _, smp0, cnt0, _ = osc.render(cam, 2, 0, 5, oracle.FAITHFUL, samples=True)
    assert (smp0 == 0).all() and cnt0["rays"] == 0
    _, smp1, _, _ = osc.render(cam, 2, 1, 5, oracle.FAITHFUL, samples=True)
    assert (smp1 == 0).all()  # one bounce allowed: hit, scatter, then depth 0 -> black, no emission here
