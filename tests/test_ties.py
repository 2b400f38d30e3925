"""How much the one open end of BVH::new matters: `sort_unstable_by` (src/bvh.rs:45,49) leaves the
order of objects with EQUAL sort keys unspecified, and concentric shells (outer radius r, inner
radius -0.98 r: min + max = 2c for both) have equal keys on every axis.  The oracle, the goldens
and the product keep such objects in their current order (a stable sort).  Here the tree is also
built with equal keys in REVERSED order, the other extreme, and the two are compared with the
reference's own traversal (FAITHFUL): which half an inner shell lands in decides which boxes lie
above it, and so whether the reference reaches it for a given ray."""
import os

import numpy as np
import pytest

from util import SCENES, camera_args


@pytest.mark.parametrize("scene", SCENES)
def test_tie_order_exposure(rt, oracle, scene):
    sc = rt.Scene.generate(scene)
    kept, reversed_ = oracle.OracleScene(sc.objects), oracle.OracleScene(sc.objects, tie_reverse=True)
    cam = oracle.camera_desc(*camera_args(scene, 160, 90, rt))
    # primary rays come from outside every shell: the closest hit is the same object either way
    for jit in (False, True):
        a = kept.trace_primary(cam, oracle.FAITHFUL, jitter=jit, seed=5, sample=9)
        b = reversed_.trace_primary(cam, oracle.FAITHFUL, jitter=jit, seed=5, sample=9)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    # whole paths: rays refracted into a shell meet the inner sphere or not depending on the order
    _, sa, ca, _ = kept.render(cam, 4, 10, 5, oracle.FAITHFUL, samples=True)
    _, sb, cb, _ = reversed_.render(cam, 4, 10, 5, oracle.FAITHFUL, samples=True)
    differing = int((sa != sb).any(axis=-1).sum())
    total = sa.shape[0] * sa.shape[1] * sa.shape[2]
    print(f"[ties] {scene}: {differing} of {total} samples differ between kept and reversed tie order "
          f"({ca['rays']} vs {cb['rays']} rays)")
    has_shells = bool((sc.objects["extent"][sc.objects["kind"] == 0, 0] < 0).any())
    if not has_shells:
        assert differing == 0
    assert differing <= 0.02 * total          # measured: 0.1 % (default), 0.2 % (sample), 0.8 % (glass)
    mean_a, mean_b = np.sqrt(np.clip(sa, 0, None)).mean(), np.sqrt(np.clip(sb, 0, None)).mean()
    assert abs(mean_a - mean_b) <= 2e-3 * mean_a


@pytest.mark.parametrize("scene", ["default", "sample", "glass"])
def test_product_builds_the_same_reversed_tree(rt, oracle, scene, monkeypatch):
    sc = rt.Scene.generate(scene)
    ids_kept = rt.BVH(sc).leaf_ids
    monkeypatch.setenv("RTHIP_TIE_REVERSE", "1")
    bvh = rt.BVH(sc)
    ids, sizes = oracle.OracleScene(sc.objects, tie_reverse=True).leaf_order()
    assert np.array_equal(bvh.leaf_ids, ids) and np.array_equal(bvh.leaf_sizes, sizes)
    assert not np.array_equal(ids, ids_kept)
