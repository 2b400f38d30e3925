"""The multi-GPU merge behind the C ABI, rehearsed on one GPU: N contexts stand for N ranks
("virtual ranks"), each renders its share of the wallpaper frame (BASELINE configs[3]: tile- or
spp-partitioned), and the merged accumulation words must equal a single context's, bit for bit.
The RCCL path is exercised with a communicator of one rank (a box with one GPU cannot hold
more: RCCL refuses two ranks on one device)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

W, H, SPP, SEED = 480, 270, 8, 0x5EED0001


def _wallpaper(rt):
    cam = ((0.0, 0.0, 0.0), rt.get_camera_rotation(-90.0, 0.0), 90.0, 10.0, 0.6)
    return rt.Scene.generate("wallpaper"), rt.Camera(*cam, W, H)


def _words(r):
    _, n = r.accum_buffer()
    w = np.zeros(n, dtype=np.uint64)
    r._check(r._lib.rthip_accum_download(r._ctx, w.ctypes.data, n))
    return w


@pytest.mark.parametrize("split", ["tile", "spp"])
@pytest.mark.parametrize("ranks", [2, 4, 7])
def test_virtual_ranks_merge_local(rt, split, ranks):
    from raytracer_amd import distributed as D
    sc, cam = _wallpaper(rt)
    kw = dict(seed=SEED, max_depth=10, sky_mode=rt.SKY_WALLPAPER)
    single = rt.Renderer(sc, cam, **kw)
    single.accumulate(SPP, spp_begin=0)
    want = _words(single)
    rs = [rt.Renderer(sc, cam, **kw) for _ in range(ranks)]
    for k, r in enumerate(rs):
        if split == "tile":
            r.accumulate(SPP, spp_begin=0, part=D.tile_part(k, ranks), sync=False)
        else:
            begin, count = D.spp_slice(SPP, k, ranks)
            r.accumulate(count, spp_begin=begin, sync=False)
    rt.merge_local(rs, root=ranks - 1)
    assert np.array_equal(_words(rs[ranks - 1]), want)
    # the other ranks' buffers are untouched: merging them once more doubles nothing on their side
    if ranks > 1 and split == "tile":
        total = sum(int(_words(r).astype(np.float64).sum() > 0) for r in rs[:-1])
        assert total == ranks - 1
    for r in rs + [single]:
        r.close()


def test_virtual_ranks_through_caller_owned_buffers(rt):
    import torch
    from raytracer_amd import distributed as D
    sc, cam = _wallpaper(rt)
    kw = dict(seed=SEED, max_depth=10, sky_mode=rt.SKY_WALLPAPER)
    ranks = 3
    rs = [rt.Renderer(sc, cam, **kw) for _ in range(ranks)]
    bufs = [D.new_accum_tensor(W, H, "cuda:0") for _ in range(ranks)]
    torch.cuda.synchronize()
    for k, r in enumerate(rs):
        r.set_accum_buffer(bufs[k].data_ptr(), bufs[k].numel())
        begin, count = D.spp_slice(SPP, k, ranks)
        r.accumulate(count, spp_begin=begin, part=(0, 1), sync=False)
    rt.merge_local(rs, root=0)
    single = rt.Renderer(sc, cam, **kw)
    single.accumulate(SPP, spp_begin=0)
    assert np.array_equal(bufs[0].cpu().numpy().view(np.uint64), _words(single))
    # errors: frames of different sizes, a list with a hole
    other = rt.Renderer(sc, rt.Camera((0.0, 0.0, 0.0), (0.0, 0.0, -1.0), 90.0, 10.0, 0.0, 64, 64), **kw)
    with pytest.raises(rt.RthipError):
        rt.merge_local([rs[0], other], root=0)
    for r in rs + [single, other]:
        r.close()


def test_rccl_reduce_of_one_rank(rt):
    """Binds librccl at run time, makes a communicator of one rank and runs the in-place reduce and
    all-reduce of the accumulation buffer through it: the words must come back unchanged."""
    sc, cam = _wallpaper(rt)
    r = rt.Renderer(sc, cam, seed=SEED, max_depth=10, sky_mode=rt.SKY_WALLPAPER)
    with pytest.raises(rt.RthipError):
        r.merge_reduce(0)                      # no communicator yet
    uid = rt.Renderer.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    r.comm_init(uid, 0, 1)
    r.accumulate(4, spp_begin=0)
    before = _words(r)
    r.merge_reduce(0)
    r.merge_reduce(-1)
    r.synchronize()
    assert np.array_equal(_words(r), before)
    with pytest.raises(rt.RthipError):
        r.merge_reduce(3)                      # not a rank of this communicator
    r.close()


def test_reserved_slots_and_torch_default_stream(rt):
    """A launch that leaves workgroup slots empty for the kernels of other streams (what a context
    with a multi-rank communicator does) renders the same words; and a caller-owned buffer zeroed by
    torch on its DEFAULT stream (handle 0) is ordered against renders through set_torch_stream."""
    import torch
    from raytracer_amd import distributed as D
    sc = rt.Scene.generate("default")
    cam = rt.Camera((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6, 640, 360)
    ref = rt.Renderer(sc, cam, seed=SEED, max_depth=10)
    ref.accumulate(SPP, spp_begin=0)
    want = _words(ref)
    ref.close()
    r = rt.Renderer(sc, cam, seed=SEED, max_depth=10, options={"grid_reserve": 48})
    r.set_torch_stream()                       # torch's current stream here is the default one
    buf = D.new_accum_tensor(640, 360, "cuda:0")
    r.set_accum_buffer(buf.data_ptr(), buf.numel())
    for _ in range(3):                         # zero-fill and render back to back, no host waits in between
        buf.zero_()
        r.accumulate(SPP, spp_begin=0, sync=False)
    torch.cuda.current_stream().synchronize()  # the renders are on that very stream
    assert np.array_equal(buf.cpu().numpy().view(np.uint64), want)
    r.set_accum_buffer(0, 0)
    r.close()
