"""world_size-2 gloo test of the multi-GPU merge path: each rank fills an accumulation buffer
for its share (spp slice or tile set; the oracle stands in for the renderer on CPU), one
reduce merges them, and the result must equal the single-process render."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, SPP, DEPTH, SEED = 40, 24, 4, 6, 9


def _worker(rank, world, port, mode, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import raytracer_amd as rt
    from raytracer_amd import distributed as D
    from oracle import binding as ob
    from util import camera_args, pack_accum_words
    sc = rt.Scene.generate("default")
    osc = ob.OracleScene(sc.objects)
    cam = ob.camera_desc(*camera_args("default", W, H, rt))
    if mode == "spp":
        begin, count = D.spp_slice(SPP, rank, world)
        img, _, _, _ = osc.render(cam, count, DEPTH, SEED, ob.TIGHT, spp_begin=begin, threads=2)
    else:
        full, _, _, _ = osc.render(cam, SPP, DEPTH, SEED, ob.TIGHT, threads=2)
        part, parts = D.tile_part(rank, world)
        ty, tx = np.mgrid[0:H, 0:W]
        tile = (ty // 8) * ((W + 7) // 8) + (tx // 8)
        img = np.where((tile % parts == part)[..., None], full, 0.0)
    acc = D.new_accum_tensor(W, H, "cpu")
    acc.copy_(torch.from_numpy(pack_accum_words(img, W, H).view(np.int64)))
    D.merge_accum(acc, dst=0)
    if rank == 0:
        np.save(out_path, D.accum_to_image(acc, W, H))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["spp", "tile"])
def test_two_rank_merge_equals_single_render(tmp_path, mode, rt, oracle):
    from util import camera_args
    out = str(tmp_path / "merged.npy")
    port = 29500 + (os.getpid() % 2000) + (0 if mode == "spp" else 1)
    mp.spawn(_worker, args=(2, port, mode, out), nprocs=2, join=True)
    merged = np.load(out)
    osc = oracle.OracleScene(rt.Scene.generate("default").objects)
    cam = oracle.camera_desc(*camera_args("default", W, H, rt))
    full, _, _, _ = osc.render(cam, SPP, DEPTH, SEED, oracle.TIGHT, threads=2)
    # each rank rounds its sum to 2^-36 once
    assert np.abs(merged - full).max() <= 2 * 2.0 ** -36 + 1e-12


def test_partitions_cover_everything_once():
    sys.path.insert(0, ROOT)
    from raytracer_amd import distributed as D
    for world in (1, 2, 3, 4, 8):
        for spp in (1, 7, 100, 2000):
            got = []
            for r in range(world):
                b, c = D.spp_slice(spp, r, world)
                got += list(range(b, b + c))
            assert got == list(range(spp))
        tiles = np.arange(240 * 135)
        owners = sum(((tiles % world) == D.tile_part(r, world)[0]).astype(int) for r in range(world))
        assert (owners == 1).all()
    assert D.accum_words(1920, 1080) == 240 * 135 * 192


def _pipeline_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from raytracer_amd import distributed as D
    n = D.accum_words(W, H)
    pipe = D.MergePipeline(lambda: torch.zeros(n, dtype=torch.int64), dst=0)
    seen = []
    frames = 5
    for k in range(frames):
        buf = pipe.acquire()
        if k >= 2 and rank == 0:
            seen.append(buf.clone())          # frame k - 2, merged
        buf.zero_()
        buf += (rank + 1) * (k + 1) + torch.arange(n, dtype=torch.int64) % 7   # this rank's "render" of frame k
        pipe.release()
    pipe.drain()
    if rank == 0:
        seen.append(pipe.bufs[(frames - 2) % 2].clone())
        seen.append(pipe.bufs[(frames - 1) % 2].clone())
        torch.save(torch.stack(seen), out_path)
    dist.barrier()
    dist.destroy_process_group()


def test_double_buffered_merge_pipeline(tmp_path):
    """Frames stream through two buffers; every frame's reduce completes before its buffer is
    reused, and rank 0 ends up with each frame's sum over ranks."""
    sys.path.insert(0, ROOT)
    from raytracer_amd import distributed as D
    out = str(tmp_path / "frames.pt")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_pipeline_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out, weights_only=True)
    n = D.accum_words(W, H)
    base = torch.arange(n, dtype=torch.int64) % 7
    assert got.shape[0] == 5
    for k in range(5):
        assert torch.equal(got[k], (1 + 2) * (k + 1) + 2 * base), k
