"""Differential test on random adversarial scenes: the reference's own traversal (FAITHFUL)
and the ordered, culled, rank-and-gate traversal the kernels implement (TIGHT) must agree on
every hit and every path."""
import numpy as np
import pytest

from util import random_scene


@pytest.mark.parametrize("seed", range(40))
def test_faithful_equals_tight_on_random_scenes(oracle, rt, seed):
    sc, cam = random_scene(rt, seed)
    osc = oracle.OracleScene(sc.objects)
    desc = oracle.camera_desc(*cam, 64, 48)
    for jitter in (False, True):
        a = osc.trace_primary(desc, oracle.FAITHFUL, jitter=jitter, seed=seed, sample=2)
        b = osc.trace_primary(desc, oracle.TIGHT, jitter=jitter, seed=seed, sample=2)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    _, sf, cf, _ = osc.render(desc, 2, 12, seed, oracle.FAITHFUL, samples=True)
    _, st, ct, _ = osc.render(desc, 2, 12, seed, oracle.TIGHT, samples=True)
    assert cf["rays"] == ct["rays"] and cf["rng_draws"] == ct["rng_draws"]
    assert np.allclose(sf, st, rtol=1e-12, atol=1e-300, equal_nan=True)


def test_faithful_equals_tight_with_thousands_of_gates(oracle, rt):
    """3 000 hollow glass shells (src/scene.rs:193-194's idiom): about a fifth of the inverted inner
    spheres are parted from their outer shell by a split and carry gate boxes.  Paths refract
    through the shells, so the gates decide hits on nearly every path."""
    from test_gpu_configs import hollow_shells
    sc = hollow_shells(rt, 3000, seed=11)
    sc.objects["center"][:-1] *= 0.1                     # a denser field: most rays go through several shells
    sc.objects["extent"][:-1, 0] *= 0.6
    gates, gated = rt.bvh_gate_count(sc)
    assert gates >= 1000 and gated >= 400
    osc = oracle.OracleScene(sc.objects)
    desc = oracle.camera_desc((0.0, 6.0, -45.0), (0.0, -6.0, 45.0), 40.0, 10.0, 0.0, 96, 54)
    for jitter in (False, True):
        a = osc.trace_primary(desc, oracle.FAITHFUL, jitter=jitter, seed=1, sample=2)
        b = osc.trace_primary(desc, oracle.TIGHT, jitter=jitter, seed=1, sample=2)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    _, sf, cf, _ = osc.render(desc, 2, 12, 1, oracle.FAITHFUL, samples=True)
    _, st, ct, _ = osc.render(desc, 2, 12, 1, oracle.TIGHT, samples=True)
    assert cf["rays"] == ct["rays"] and cf["rng_draws"] == ct["rng_draws"] and cf["scatter_dielectric"] > 5 * cf["paths"] // 10
    assert np.allclose(sf, st, rtol=1e-12, atol=1e-300, equal_nan=True)
