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
