"""Soak: many launches of random shape under random option sets on a handful of scenes; every launch is made
twice and must render the same words both times, option sets that must not change the image are held
against the defaults' words, and so is the same render made as a queue of one-sample launches.  Prints a line per hundred launches.
  python tools/soak.py [N] [seed]"""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as rt
N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
g = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
CAM = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6)
CAM5 = ((0.0, 60.0, -550.0), (0.0, -60.0, 550.0), 40.0, 10.0, 0.0)
SCENES = [("default", CAM), ("sample", CAM), ("glass", CAM), ("synthetic:3000", CAM5), ("synthetic:70000", CAM5)]
OPTS = [{}, {"packet": 0}, {"pixel_major": 1}, {"pixel_major": 0}, {"threads": 256}, {"threads": 512}, {"chunk": 3}, {"chunk": 7}, {"scene_lds": 0},
        {"scene_lds": 0, "pixel_major": 1}, {"qnodes": 0}, {"stack_mem": 1}, {"xchg": 1}, {"regen_min": 4}, {"trav_threshold": 30}, {"inner_exit": 2}]
scenes = [(name, rt.Scene.generate(name), cam) for name, cam in SCENES]
t0 = time.time()
bad = 0
for k in range(N):
    name, sc, cam = scenes[int(g.integers(0, len(scenes)))]
    w, h = int(g.integers(8, 400)), int(g.integers(8, 300))
    spp, depth, seed = int(g.integers(1, 20)), int(g.integers(0, 12)), int(g.integers(0, 1 << 30))
    begin = int(g.integers(0, 1000))
    words = []
    for opts in ({}, OPTS[int(g.integers(0, len(OPTS)))]):
        for rep in range(2):
            r = rt.Renderer(sc, rt.Camera(*cam, w, h), seed=seed, max_depth=depth, options=opts)
            r.accumulate(spp, spp_begin=begin)
            words.append((zlib.crc32(r.read_accum().tobytes()), r.stats()["rays"], opts))
            r.close()
    # the same samples as a queue of one-sample launches without waits (short ones overlap on the library's side streams)
    r = rt.Renderer(sc, rt.Camera(*cam, w, h), seed=seed, max_depth=depth)
    total = 0
    for k in range(spp):
        r.accumulate(1, spp_begin=begin + k, sync=False)
        if g.random() < 0.1:
            total += 0 * r.stats()["rays"]          # a look at the statistics now and then (joins the streams)
    words.append((zlib.crc32(r.read_accum().tobytes()), words[0][1], "queue of 1-spp launches"))
    r.close()
    if len({(c, n) for c, n, _ in words}) != 1:
        bad += 1
        print("MISMATCH", name, w, h, spp, depth, seed, begin, words, flush=True)
    if (k + 1) % 100 == 0:
        print(f"{k + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print("done:", N, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
