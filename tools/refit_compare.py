"""Refit (rthip_scene_refit: the tree keeps its shape, boxes fitted again on the GPU) against a
fresh upload, on scenes whose ordinary objects have moved by a normal step of the given spread:
wall time of each call and the rate of the tree that comes out.
  python tools/refit_compare.py [scene ...]        (default: default synthetic:200000 synthetic:1000000)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as rt
HEADLESS = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6)
CAM5 = ((0.0, 60.0, -550.0), (0.0, -60.0, 550.0), 40.0, 10.0, 0.0)


def moved(scene, spread, seed=1):
    g = np.random.default_rng(seed)
    o = scene.objects.copy()
    small = np.abs(o["extent"]).max(axis=1) < 10.0
    o["center"][small] += g.normal(0.0, spread, (int(small.sum()), 3)) * [1.0, 0.0, 1.0]
    return rt.Scene(o)


def rate(r, spp):
    best = 0
    for _ in range(3):
        r.reset_accumulation(); r.accumulate(spp, spp_begin=0)
        s = r.stats(); best = max(best, s["paths"] / s["kernel_ms"] / 1e3)
    return best


def timed(fn):
    t = []
    for _ in range(3):
        t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
    return min(t) * 1e3


for name in sys.argv[1:] or ["default", "synthetic:200000", "synthetic:1000000"]:
    sc = rt.Scene.generate(name, 0x5EED0001)
    small = not name.startswith("synthetic")
    cam = rt.Camera(*(HEADLESS if small else CAM5), 1920, 1080)
    spp, depth = (32, 10) if small else (8, 50)
    for tree, opts in [("host surface-area tree", {}), ("device-built tree", {"tree_gpu": 1})]:
        r = rt.Renderer(sc, cam, max_depth=depth, seed=1, options=opts)
        base = rate(r, spp)
        up = timed(lambda: (r.set_scene(sc), r.synchronize()))
        print(f"{name}, {tree}: upload {up:8.1f} ms, {base:7.0f} Mpaths/s", flush=True)
        for spread in (0.05, 0.2, 1.0, 5.0):
            new = moved(sc, spread)
            r.set_scene(sc)
            ms = timed(lambda: (r.refit_scene(new), r.synchronize()))
            refitted = rate(r, spp)
            r.set_scene(new)
            fresh = rate(r, spp)
            print(f"    moved by {spread:4.2f}: refit {ms:8.1f} ms, {refitted:7.0f} Mpaths/s refitted, {fresh:7.0f} built anew ({refitted / fresh:.2f})", flush=True)
        r.close()
