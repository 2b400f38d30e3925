"""The degenerate scenes of tests/test_builder_property.py on the GPU: f64 hits and paths bit for
bit against the oracle, on the host-built tree and on the GPU-built one, with tie ranks and gates
from the host's pass and from the device's, after an upload and after a refit."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings

from test_builder_property import object_arrays

pytestmark = pytest.mark.gpu


def _check(rt, oracle, objs):
    cam = ((9.0, 4.0, 11.0), (-9.0, -3.5, -11.0), 40.0, 10.0, 0.2, 32, 24)
    osc = oracle.OracleScene(objs)
    desc = oracle.camera_desc(*cam)
    ref = osc.trace_primary(desc, oracle.TIGHT, jitter=True, seed=5, sample=1)
    _, smp, cnt, _ = osc.render(desc, 2, 8, 5, oracle.TIGHT, samples=True)
    # (ref_gpu = 1: tie ranks and gates from the device's level-by-level sorts instead of the host's
    # pass over the reference's tree, which is what scenes this small get by default)
    for options in ({}, {"tree_gpu": 1}, {"tree_gpu": 1, "ref_gpu": 1}, {"ref_gpu": 1}):
        r = rt.Renderer(rt.Scene(objs), rt.Camera(*cam), seed=5, precision=rt.F64, max_depth=8, options=options)
        for refitted in (False, True):
            if refitted:
                if "tree_gpu" in options:
                    break
                r.refit_scene(rt.Scene(objs))     # the objects where they are: ranks, gates and boxes made again
            r.validate_tree()
            for a, b in zip(ref, r.trace_primary(rt.F64, jitter=True, sample=1)):
                assert np.array_equal(a, b)
            assert np.array_equal(r.render_samples(2), smp, equal_nan=True) and r.stats()["rays"] == cnt["rays"]
        r.precision = rt.F32                      # production arithmetic: must not hang or fault on such input
        r.accumulate(2, spp_begin=0)
        assert r.stats()["paths"] == 32 * 24 * 2
        r.close()


@settings(max_examples=30, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture])
@given(object_arrays())
def test_degenerate_scenes_f64_bit_exact(rt, oracle, objs):
    _check(rt, oracle, objs)


@settings(max_examples=12, deadline=None, derandomize=True, suppress_health_check=[HealthCheck.too_slow, HealthCheck.function_scoped_fixture, HealthCheck.data_too_large, HealthCheck.large_base_example])
@given(object_arrays(max_objects=400))
def test_larger_degenerate_scenes_f64_bit_exact(rt, oracle, objs):
    _check(rt, oracle, objs)
