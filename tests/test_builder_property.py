"""Property tests of the host builder on degenerate inputs (hypothesis): whatever finite objects a
host hands over - zero sizes, negative sizes, duplicates, huge and tiny coordinates side by side,
one object or a few thousand - the build either succeeds with a sound tree (every box contains what
is below it, every primitive that can be hit in exactly one leaf, depth as reported) or says why
not; the same input gives the same bytes; and the oracle's two traversals agree on the scene."""
import ctypes as C

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from raytracer_amd import _native as N

SIZES = st.sampled_from([0.0, 1e-30, 1e-3, 0.2, 1.0, 50.0, 1e5, -0.2, -1.0])
COORDS = st.sampled_from([0.0, -0.0, 1e-9, 0.5, -3.0, 7.25, 1e4, -1e6])


@st.composite
def object_arrays(draw, max_objects=40):
    n = draw(st.integers(min_value=1, max_value=max_objects))
    o = np.zeros(n, dtype=N.OBJECT_DTYPE)
    for i in range(n):
        if i and draw(st.integers(0, 4)) == 0:
            o[i] = o[draw(st.integers(0, i - 1))]          # an exact duplicate
            continue
        o["kind"][i] = draw(st.sampled_from([N.SPHERE, N.CUBE]))
        o["material"][i] = draw(st.sampled_from([N.LAMBERTIAN, N.METAL, N.DIELECTRIC]))
        o["center"][i] = [draw(COORDS), draw(COORDS), draw(COORDS)]
        o["extent"][i] = [draw(SIZES), draw(SIZES), draw(SIZES)]
        o["albedo"][i] = [0.5, 0.6, 0.7]
        o["param"][i] = draw(st.sampled_from([0.0, 0.3, 1.5]))
    return o


def validate(objs):
    bad, nodes, depth, ck = C.c_int64(-1), C.c_int32(), C.c_int32(), C.c_uint64()
    code = N.lib().rthip_bvh_validate(objs.ctypes.data, len(objs), C.byref(bad), C.byref(nodes), C.byref(depth), C.byref(ck))
    return code, bad.value, nodes.value, depth.value, ck.value


@settings(max_examples=150, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow])
@given(object_arrays())
def test_any_finite_scene_builds_a_sound_tree(objs):
    code, bad, nodes, depth, ck = validate(objs)
    assert code == N.OK, N.lib().rthip_last_error(None)
    assert bad == 0 and nodes >= 1 and 1 <= depth <= 48
    assert validate(objs.copy())[4] == ck                  # the same input, the same bytes


@settings(max_examples=40, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(object_arrays())
def test_oracle_traversals_agree_on_degenerate_scenes(oracle, objs):
    osc = oracle.OracleScene(objs)
    desc = oracle.camera_desc((9.0, 4.0, 11.0), (-9.0, -3.5, -11.0), 40.0, 10.0, 0.2, 24, 16)
    a = osc.trace_primary(desc, oracle.FAITHFUL, jitter=True, seed=5, sample=1)
    b = osc.trace_primary(desc, oracle.TIGHT, jitter=True, seed=5, sample=1)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_scale_thresholds_of_the_builder(rt):
    """Object counts around the builder's switches (one leaf, the exact sweep, threads, quantised nodes)."""
    for n in (1, 2, 4, 5, 6, 31, 32, 33, 47, 48, 49, 2047, 2049, 4100):
        sc = rt.Scene.generate(f"synthetic:{n}", 99)
        code, bad, nodes, depth, _ = validate(sc.objects)
        assert code == N.OK and bad == 0, (n, N.lib().rthip_last_error(None))
