"""Host surface-area build against the device-side linear-BVH build (option tree_gpu): wall time of
rthip_scene_upload and the rate of the tree that comes out, on the synthetic scenes.
  python tools/build_compare.py [objects ...]        (default 1000000)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
CAM5 = ((0.0, 60.0, -550.0), (0.0, -60.0, 550.0), 40.0, 10.0, 0.0)
for n in [int(a) for a in sys.argv[1:]] or [1000000]:
    sc = rt.Scene.generate(f"synthetic:{n}", 0x5EED0001)
    for name, opts in [("host surface-area build", {}), ("device build, 1 per leaf", {"tree_gpu": 1}), ("device build, 2 per leaf", {"tree_gpu": 1, "gpu_leaf": 2}),
                       ("device build, 4 per leaf", {"tree_gpu": 1, "gpu_leaf": 4})]:
        r = rt.Renderer(sc, rt.Camera(*CAM5, 1920, 1080), max_depth=50, seed=1, options=opts)
        t = []
        for _ in range(3):
            t0 = time.perf_counter(); r.set_scene(sc); r.synchronize(); t.append(time.perf_counter() - t0)
        best = 0
        for _ in range(3):
            r.reset_accumulation(); r.accumulate(8, spp_begin=0)
            s = r.stats(); best = max(best, s["paths"] / s["kernel_ms"] / 1e3)
        print(f"{n} objects, {name:26s}: upload {min(t) * 1e3:8.1f} ms, {best:7.0f} Mpaths/s, {s['n_nodes']} nodes, depth {s['tree_depth']}", flush=True)
        r.close()
