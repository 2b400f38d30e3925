"""Every combination of kernel variant (megakernel, wavefront) and tree builder (surface-area on
the host, the reference's shape, linear BVH on the device) traces the same paths: f64 samples
bit-identical across all six combinations, f32 accumulation words identical as well (the closest
hit does not depend on the tree or on the order rays are worked off in)."""
import itertools

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CAM = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6, 96, 54)
CAM5 = ((0.0, 60.0, -550.0), (0.0, -60.0, 550.0), 40.0, 10.0, 0.0, 96, 54)


@pytest.mark.parametrize("scene,cam", [("default", CAM), ("sample", CAM), ("synthetic:5000", CAM5), ("synthetic:120000", CAM5)])
def test_variants_and_trees_agree(rt, scene, cam):
    sc = rt.Scene.generate(scene)
    ref_samples = ref_words = None
    for tree, variant in itertools.product(({}, {"tree_ref": 1}, {"tree_gpu": 1}), (rt.MEGAKERNEL, rt.WAVEFRONT)):
        r = rt.Renderer(sc, rt.Camera(*cam), seed=3, max_depth=8, precision=rt.F64, variant=variant, options=tree)
        smp = r.render_samples(2)
        r.precision = rt.F32
        r.accumulate(3, spp_begin=0)
        _, n = r.accum_buffer()
        words = np.zeros(n, dtype=np.uint64)
        r._check(r._lib.rthip_accum_download(r._ctx, words.ctypes.data, n))
        r.close()
        if ref_samples is None:
            ref_samples, ref_words = smp, words
            assert np.isfinite(smp).all() and words.any()
            continue
        assert np.array_equal(smp, ref_samples), (tree, variant)
        # f32: a hit within rounding of a box face may be culled by one tree and not by another
        assert (words != ref_words).mean() < 1e-3, (tree, variant)


@pytest.mark.parametrize("scene,cam,sky", [("default", CAM, 0), ("sample", CAM, 0), ("glass", CAM, 0), ("wallpaper", CAM, 1), ("synthetic:5000", CAM5, 0),
                                           ("synthetic:120000", CAM5, 0)])
def test_launch_options_change_no_word(rt, scene, cam, sky):
    """The launch-side experiments and shortcuts (rays regrouped by shading arm through LDS queues, with
    small and with large queues; traversal stacks in global memory; both at once; a one-leaf scene walked
    from its root instead of tested at ray start; other workgroup sizes; for scenes in global memory the
    camera rays walked one by one instead of 64 at a time as a packet, on full and on quantised nodes, the packets made of one sample
    of 64 pixels or of all samples of a few pixels) render the same accumulation
    words as the defaults: paths are deterministic and accumulation is integer, so which lane, wave or
    workgroup carries a ray, and where its stack lives, cannot show in the image."""
    sc = rt.Scene.generate(scene)
    w, h = 640, 360
    ref = None
    for opts in ({}, {"xchg": 1}, {"xchg": 1, "xchg_cap": 64, "xchg_pop_min": 8}, {"xchg": 1, "xchg_push_max": 6, "xchg_pop_min": 40},
                 {"stack_mem": 1}, {"xchg": 1, "stack_mem": 1}, {"root_leaf": 0}, {"threads": 256}, {"threads": 512}, {"chunk": 3}, {"packet": 0}, {"packet": 0, "qnodes": 0}, {"qnodes": 0},
                 {"scene_lds": 0}, {"scene_lds": 0, "packet": 0}, {"scene_lds": 0, "pixel_major": 1}, {"scene_lds": 0, "pixel_major": 1, "chunk": 5},
                 {"pixel_major": 0}, {"pixel_major": 1, "chunk": 16}):
        r = rt.Renderer(sc, rt.Camera(*cam[:5], w, h), seed=3, max_depth=12, sky_mode=sky, options=opts)
        r.accumulate(16, spp_begin=0)       # 3.7 M paths: long enough for the queues to be used (short launches leave them out) and served
        _, n = r.accum_buffer()
        words = np.zeros(n, dtype=np.uint64)
        r._check(r._lib.rthip_accum_download(r._ctx, words.ctypes.data, n))
        s = r.stats()
        r.close()
        if ref is None:
            ref, ref_rays = words, s["rays"]
            assert words.any()
            continue
        assert s["rays"] == ref_rays, opts
        assert np.array_equal(words, ref), opts


@pytest.mark.parametrize("scene,cam", [("default", CAM), ("synthetic:5000", CAM5)])
def test_depth_limits_with_and_without_packets(rt, scene, cam):
    """Depth limits 0, 1 and 2 (src/renderer.rs:10-12: depth <= 0 is black): a path then ends at or right
    after its first ray, which is the part that batches of camera rays handle on their own when they walk
    the tree as packets.  Same words with packets and without; path and ray counts as the limits say; the
    f32 kernels that write per-sample radiance (the instrumented ones) agree with the accumulated image."""
    sc = rt.Scene.generate(scene)
    w, h, spp = 320, 184, 4
    for depth in (0, 1, 2):
        got = {}
        for packet in (1, 0):
            r = rt.Renderer(sc, rt.Camera(*cam[:5], w, h), seed=5, max_depth=depth, options={"packet": packet})
            r.accumulate(spp, spp_begin=0)
            words = r.read_accum()
            s = r.stats()
            assert s["paths"] == w * h * spp, (depth, packet)
            assert s["rays"] == (0 if depth == 0 else s["paths"] if depth == 1 else s["rays"]), (depth, packet)
            if depth == 2:
                assert s["paths"] < s["rays"] <= 2 * s["paths"]
            smp = r.render_samples(spp)                      # [spp][H][W][3] linear radiance, f32
            r.close()
            got[packet] = (words, s["rays"], smp)
        assert np.array_equal(got[1][0], got[0][0]), depth
        assert got[1][1] == got[0][1], depth
        assert np.array_equal(got[1][2], got[0][2]), depth
        if depth == 0:
            assert not got[1][0].any() and not got[1][2].any()
        else:
            assert got[1][0].any()
        # the image is the sum over samples of sqrt(radiance) (f32 square root: an ulp; fixed point: 2^-36 per sample)
        assert np.allclose(got[1][0], np.sqrt(np.clip(got[1][2], 0.0, 65536.0)).sum(axis=0), rtol=2e-6, atol=1e-6), depth


def test_packets_on_adversarial_scenes(rt):
    """Random adversarial scenes (hollow shells and stand-alone inverted spheres with their gates, coplanar
    and degenerate cubes, objects around and behind the camera) pushed through the kernels for scenes in
    global memory: camera rays walked as packets (on full nodes, with quantised nodes beside them or not)
    render the same words as camera rays walked one by one, and as the LDS kernels."""
    from util import random_scene
    w, h = 96, 64
    for seed in range(16):
        sc, cam = random_scene(rt, 100 + seed)
        ref = None
        for opts in ({}, {"scene_lds": 0}, {"scene_lds": 0, "packet": 0}, {"scene_lds": 0, "qnodes": 0}, {"scene_lds": 0, "threads": 512},
                     {"scene_lds": 0, "pixel_major": 1}, {"scene_lds": 0, "pixel_major": 1, "chunk": 3}):
            r = rt.Renderer(sc, rt.Camera(*cam, w, h), seed=seed, max_depth=6, options=opts)
            r.accumulate(6, spp_begin=0)
            words, rays = r.read_accum(), r.stats()["rays"]
            r.close()
            if ref is None:
                ref, ref_rays = words, rays
                continue
            assert rays == ref_rays, (seed, opts)
            assert np.array_equal(words, ref), (seed, opts)


def test_shares_add_up_on_a_large_scene(rt):
    """A scene of more than 65 536 primitives (global memory, quantised nodes, packets made of all samples of
    a few pixels, work items taken in blocks of tiles): tile shares and sample shares rendered one after the
    other into one buffer give the words of the whole render, whatever the chunk."""
    sc = rt.Scene.generate("synthetic:70000")
    w, h, spp = 200, 136, 9           # a frame whose last tile column and row are partial
    for opts in ({}, {"chunk": 4}, {"pixel_major": 0}):
        r = rt.Renderer(sc, rt.Camera(*CAM5[:5], w, h), seed=9, max_depth=10, options=opts)
        r.accumulate(spp, spp_begin=0)
        whole, rays = r.read_accum(), r.stats()["rays"]
        r.reset_accumulation()
        total = 0
        for part in range(3):
            r.accumulate(spp, part=(part, 3), spp_begin=0)
            total += r.stats()["rays"]
        assert np.array_equal(r.read_accum(), whole), opts
        assert total == rays
        r.reset_accumulation()
        for begin, count in ((0, 2), (2, 6), (8, 1)):
            r.accumulate(count, spp_begin=begin)
        assert np.array_equal(r.read_accum(), whole), opts
        r.close()


@pytest.mark.parametrize("scene,cam", [("default", CAM), ("synthetic:5000", CAM5)])
def test_short_launches_that_overlap_add_up(rt, scene, cam):
    """A viewer's loop: 1-spp frames queued without waits on the context's own stream, which the library
    spreads over three streams so that their tails overlap.  The sum is the words of one launch of all the
    samples (integer adds), a reset or a read-back in between is ordered behind every launch in flight, the
    statistics are the last launch's, and with the option off the same words come out."""
    sc = rt.Scene.generate(scene)
    w, h, frames = 320, 184, 37
    want = None
    for opts in ({"overlap": 0}, {}):
        r = rt.Renderer(sc, rt.Camera(*cam[:5], w, h), seed=12, max_depth=10, options=opts)
        r.accumulate(frames, spp_begin=0)
        whole = r.read_accum()
        if want is None:
            want = whole
        assert np.array_equal(whole, want)
        r.reset_accumulation()
        for k in range(frames):
            r.accumulate(1, spp_begin=k, sync=False)
        assert np.array_equal(r.read_accum(), want), opts          # a read-back right behind the queue of launches
        assert r.stats()["paths"] == w * h
        for k in range(5):
            r.accumulate(1, spp_begin=k, sync=False)
        r.reset_accumulation()                                     # a clear while launches are in flight comes after them
        for k in range(frames):
            r.accumulate(1, spp_begin=k, sync=False)
            if k == 20:
                assert r.stats()["rays"] >= w * h                  # a look at the statistics in the middle joins the streams and waits
        r.synchronize()
        assert np.array_equal(r.read_accum(), want), opts
        r.close()
