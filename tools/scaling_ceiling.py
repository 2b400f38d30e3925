"""Strong-scaling ceiling from one GPU: the launches an N-way split of a bench step produces, timed
back to back on one MI355X the way bench.py runs its steps (zero-fill of the buffer, one render per
step, no host wait in between), for N = 1, 2, 4, 8 and both splits:
  spp split  - rank r renders samples [r*spp/N, (r+1)*spp/N) of the full frame (the largest share is timed);
  tile split - rank r renders all samples of the tiles r, r+N, r+2N, ... (every r is timed, the slowest counts).
Best-case efficiency per N = t(1) / (N * t_share(N)): what N GPUs could reach if the merge were free
(it runs on a communication stream under the next step's tracing).  The cost of merging two buffers on
one GPU (rthip_merge_local: an add kernel over the frame's words) is printed beside it as the scale of
the merge.  Writes gpurun_out/scaling_ceiling.json.
  python tools/scaling_ceiling.py [cfg2,cfg3,cfg4]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import distributed as D
import bench

names = (sys.argv[1] if len(sys.argv) > 1 else "cfg2,cfg3,cfg4").split(",")
dev = torch.device("cuda", 0)
torch.cuda.set_stream(torch.cuda.Stream(dev))
out = {}
for name in names:
    wl = bench.WORKLOADS[name]
    sc = rt.Scene.generate(wl["scene"], bench.SCENE_SEED)
    W, H, spp = wl["width"], wl["height"], wl["spp"]
    r = rt.Renderer(sc, rt.Camera(*wl["camera"](), W, H), max_depth=wl["depth"], seed=bench.RENDER_SEED, sky_mode=wl["sky"])
    r.set_torch_stream()
    bufs = [D.new_accum_tensor(W, H, dev) for _ in range(2)]

    def ms_per_step(begin, count, part, steps):
        def step(k):
            b = bufs[k & 1]
            b.zero_()
            r.set_accum_buffer(b.data_ptr(), b.numel())
            r.accumulate(count, part=part, spp_begin=begin, sync=False)
        for k in range(2): step(k)
        torch.cuda.synchronize(dev)
        best = 1e30
        for _ in range(3):
            t0 = time.perf_counter()
            for k in range(steps): step(k)
            torch.cuda.synchronize(dev)
            best = min(best, (time.perf_counter() - t0) / steps * 1e3)
        return best

    rows = {}
    t1 = ms_per_step(0, spp, (0, 1), 8)
    for n in (1, 2, 4, 8):
        shares = [D.spp_slice(spp, k, n) for k in range(n)]
        big = max(shares, key=lambda s: s[1])
        t_spp = t1 if n == 1 else ms_per_step(big[0], big[1], (0, 1), 8 * n)
        t_tile = t1 if n == 1 else max(ms_per_step(0, spp, D.tile_part(k, n), 8 * n) for k in range(n))
        rows[n] = dict(spp_share=big[1], ms_spp_split=t_spp, eff_spp_split=t1 / (n * t_spp), ms_tile_split=t_tile, eff_tile_split=t1 / (n * t_tile))
    # the merge's scale: two contexts on this GPU, one add of the frame's words
    r2 = rt.Renderer(sc, rt.Camera(*wl["camera"](), W, H), max_depth=wl["depth"], seed=bench.RENDER_SEED, sky_mode=wl["sky"])
    r.set_accum_buffer(0, 0); r.reset_accumulation(); r2.reset_accumulation()
    r.accumulate(1, spp_begin=0); r2.accumulate(1, spp_begin=1)
    rt.merge_local([r, r2], root=0)
    t0 = time.perf_counter()
    for _ in range(5): rt.merge_local([r, r2], root=0)
    merge_ms = (time.perf_counter() - t0) / 5 * 1e3
    r2.close(); r.close()
    words = bufs[0].numel()
    out[name] = dict(workload=wl["desc"], ms_one_gpu=t1, by_gpus=rows, merge_local_ms=merge_ms, buffer_mb=words * 8 / 1e6)
    print(f"{name}: one GPU {t1:.3f} ms per step of {spp} spp; buffer {words * 8 / 1e6:.0f} MB, merge of two buffers on one GPU {merge_ms:.3f} ms")
    for n, v in rows.items():
        print(f"  N={n}: spp split {v['spp_share']:4d} spp {v['ms_spp_split']:8.3f} ms -> best-case efficiency {v['eff_spp_split']:.3f};"
              f"  tile split {v['ms_tile_split']:8.3f} ms -> {v['eff_tile_split']:.3f}", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/scaling_ceiling.json", "w"), indent=1)
