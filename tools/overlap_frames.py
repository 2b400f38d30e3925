"""How much consecutive short launches gain from overlapping their tails: 1-spp frames added into one
accumulation buffer by one context (launches follow one another on its stream) against two and three
contexts on streams of their own taking turns (their launches overlap; integer adds commute, so the sum
is the same).  Prints ms per frame."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import raytracer_amd as rt
from raytracer_amd import distributed as D
CAM = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6)
sc = rt.Scene.generate("default")
for (w, h, spp) in ((1920, 1080, 1), (1920, 1080, 2), (1920, 1080, 4), (1920, 1080, 12), (3840, 2160, 1), (480, 270, 1)):
    ref = None
    for n_ctx in (1, 2, 3):
        buf = D.new_accum_tensor(w, h, "cuda:0")
        torch.cuda.synchronize()
        rs = [rt.Renderer(sc, rt.Camera(*CAM, w, h), max_depth=10, seed=1) for _ in range(n_ctx)]
        for r in rs:
            r.set_accum_buffer(buf.data_ptr(), buf.numel())
        frames = 60
        for k in range(6):
            rs[k % n_ctx].accumulate(spp, spp_begin=k * spp, sync=False)
        for r in rs: r.synchronize()
        buf.zero_(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(frames):
            rs[k % n_ctx].accumulate(spp, spp_begin=k * spp, sync=False)
        for r in rs: r.synchronize()
        dt = (time.perf_counter() - t0) / frames * 1e3
        words = buf.cpu().numpy()
        if ref is None: ref = words
        print(f"{w}x{h} {spp} spp per frame, {n_ctx} context(s): {dt:.3f} ms per frame, same words as one context: {bool(np.array_equal(words, ref))}", flush=True)
        for r in rs: r.close()
