"""rthip_render_accumulate_noalloc: the render call that allocates nothing (its control block lives
in a static device array instead of memory from hipMalloc) against rthip_render_accumulate, bit for
bit, on two calls in a row, the second on a smaller input."""
import ctypes as C

import numpy as np
import pytest

from util import camera_args

pytestmark = pytest.mark.gpu


def _words(r):
    _, n = r.accum_buffer()
    w = np.zeros(n, dtype=np.uint64)
    r._check(r._lib.rthip_accum_download(r._ctx, w.ctypes.data, n))
    return w


def test_noalloc_entry_point_matches_the_allocating_one(rt):
    from raytracer_amd import _native as N
    sc = rt.Scene.generate("default")
    r = rt.Renderer(sc, rt.Camera(*camera_args("default", 640, 360, rt)), seed=41)
    # two calls in a row through the static scratch; the second on a smaller frame with fewer samples
    for (w, h, spp, prec) in ((640, 360, 6, rt.F32), (200, 120, 3, rt.F32), (96, 54, 2, rt.F64)):
        r.update_camera(rt.Camera(*camera_args("default", w, h, rt)))
        r.precision = prec
        r.accumulate(spp, spp_begin=0)
        want, want_stats = _words(r), r.stats()
        r.reset_accumulation()
        r.accumulate(spp, spp_begin=0, noalloc=True)
        got, got_stats = _words(r), r.stats()
        assert np.array_equal(got, want), (w, h, spp)
        assert got_stats["paths"] == want_stats["paths"] == w * h * spp and got_stats["rays"] == want_stats["rays"]
    # back to back without a host wait in between, then the allocating call again: the blocks do not mix
    r.precision = rt.F32
    r.reset_accumulation()
    r.accumulate(2, spp_begin=0, noalloc=True, sync=False)
    r.accumulate(1, spp_begin=2, noalloc=True, sync=False)
    r.synchronize()
    assert r.stats()["paths"] == 96 * 54
    three = _words(r)
    r.reset_accumulation(); r.accumulate(3, spp_begin=0)
    assert np.array_equal(_words(r), three)
    # what it cannot do without allocating is refused with a message, not done by allocating
    p = r._params(0, 1)
    p.variant = rt.WAVEFRONT
    assert r._lib.rthip_render_accumulate_noalloc(r._ctx, C.byref(p)) == N.ERR_LIMIT
    assert b"wavefront" in r._lib.rthip_last_error(r._ctx)
    r.close()
    # a caller-owned accumulation buffer, a second context on the same device: same scratch, same result
    import torch
    from raytracer_amd import distributed as D
    r2 = rt.Renderer(sc, rt.Camera(*camera_args("default", 96, 54, rt)), seed=41)
    buf = D.new_accum_tensor(96, 54, "cuda:0")
    torch.cuda.synchronize()
    r2.set_accum_buffer(buf.data_ptr(), buf.numel())
    r2.accumulate(3, spp_begin=0, noalloc=True)
    assert np.array_equal(buf.cpu().numpy().view(np.uint64), three)
    r2.close()


def test_noalloc_overlap_on_two_streams_is_refused(rt):
    """Two contexts on one GPU share the GPU's static control block: while a launch of one is still
    running on its stream, a call from the other (another stream) returns RTHIP_ERR_STATE and launches
    nothing; once the first has finished the same call goes through, and calls that share a stream
    follow one another without a refusal."""
    from raytracer_amd import _native as N
    sc = rt.Scene.generate("default")
    a = rt.Renderer(sc, rt.Camera(*camera_args("default", 960, 540, rt)), seed=5)
    b = rt.Renderer(sc, rt.Camera(*camera_args("default", 96, 54, rt)), seed=5)
    b.accumulate(2, spp_begin=0)
    want = _words(b)
    b.reset_accumulation()
    b.synchronize()
    a.accumulate(256, spp_begin=0, noalloc=True, sync=False)      # tens of milliseconds of work
    p = b._params(0, 2)
    rc = b._lib.rthip_render_accumulate_noalloc(b._ctx, C.byref(p))
    assert rc == N.ERR_STATE, rc
    assert b"static control block" in b._lib.rthip_last_error(b._ctx)
    a.synchronize()
    assert a.stats()["paths"] == 960 * 540 * 256                   # the refused call left the running launch alone
    b.accumulate(2, spp_begin=0, noalloc=True)
    assert np.array_equal(_words(b), want)
    # one stream for both contexts: the launches are ordered, nothing to refuse
    import torch
    s = torch.cuda.Stream()
    a.set_torch_stream(s); b.set_torch_stream(s)
    a.reset_accumulation(); b.reset_accumulation()
    a.accumulate(64, spp_begin=0, noalloc=True, sync=False)
    b.accumulate(2, spp_begin=0, noalloc=True, sync=False)
    a.synchronize(); b.synchronize()
    assert np.array_equal(_words(b), want)
    a.close(); b.close()


def test_noalloc_after_a_reduce_and_under_graph_capture(rt):
    """A render through the allocation-free entry point queued right behind an RCCL reduce of the same
    buffer waits for it (one-rank communicator: the words come back as rendered, twice the samples after
    the second render); and the call can be captured into a HIP graph (torch.cuda.graph) whose replays
    render the same words as the direct call."""
    import torch
    from raytracer_amd import distributed as D
    sc = rt.Scene.generate("default")
    r = rt.Renderer(sc, rt.Camera(*camera_args("default", 320, 180, rt)), seed=9)
    r.accumulate(3, spp_begin=0); r.accumulate(3, spp_begin=3)
    six = _words(r)
    r.reset_accumulation()
    r.comm_init(rt.Renderer.comm_unique_id(), 0, 1)
    r.accumulate(3, spp_begin=0, noalloc=True, sync=False)
    r.merge_reduce(0)
    r.accumulate(3, spp_begin=3, noalloc=True, sync=False)         # behind the reduce, on the render stream
    r.synchronize()
    assert np.array_equal(_words(r), six)
    r.close()
    # graph capture: a caller-owned buffer, the context on the capturing stream
    g = rt.Renderer(sc, rt.Camera(*camera_args("default", 320, 180, rt)), seed=9)
    buf = D.new_accum_tensor(320, 180, "cuda:0")
    g.set_accum_buffer(buf.data_ptr(), buf.numel())
    s = torch.cuda.Stream()
    g.set_torch_stream(s)
    with torch.cuda.stream(s):
        g.accumulate(3, spp_begin=0, noalloc=True)                  # warm-up outside the capture (occupancy queries, attributes)
        buf.zero_()
    s.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=s):
        g.accumulate(3, spp_begin=0, noalloc=True, sync=False)
        g.accumulate(3, spp_begin=3, noalloc=True, sync=False)
    buf.zero_()
    torch.cuda.synchronize()
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(buf.cpu().numpy().view(np.uint64), six)
    buf.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert np.array_equal(buf.cpu().numpy().view(np.uint64), six)
    del graph
    g.close()


def test_noalloc_on_a_scene_in_global_memory(rt):
    """The same comparison on a scene whose tree stays in global memory (quantised nodes, camera rays walked
    as packets): those kernels take nothing from the allocator either, and two calls in a row, the second
    smaller, render the words of the allocating entry point."""
    sc = rt.Scene.generate("synthetic:5000")
    cam5 = ((0.0, 60.0, -550.0), (0.0, -60.0, 550.0), 40.0, 10.0, 0.0)
    r = rt.Renderer(sc, rt.Camera(*cam5, 480, 270), seed=43, max_depth=8)
    for (w, h, spp) in ((480, 270, 5), (160, 96, 2)):
        r.update_camera(rt.Camera(*cam5, w, h))
        r.accumulate(spp, spp_begin=0)
        want, want_stats = _words(r), r.stats()
        assert not want_stats["scene_in_lds"]
        r.reset_accumulation()
        r.accumulate(spp, spp_begin=0, noalloc=True)
        got, got_stats = _words(r), r.stats()
        assert np.array_equal(got, want), (w, h, spp)
        assert got_stats["paths"] == want_stats["paths"] == w * h * spp and got_stats["rays"] == want_stats["rays"]
    r.close()
