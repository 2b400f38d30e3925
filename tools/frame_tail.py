"""Where a short launch's time goes: the 1-spp frame at several depths (depth 1 = one ray per path:
no long paths, hence no tail of stragglers) and frame sizes (the intercept is what a launch costs
whatever its size)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
CAM = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6)
sc = rt.Scene.generate("default")
SIZES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(1920, 1080), (960, 540), (480, 270), (3840, 2160)]
for w, h in SIZES:
    for depth in (1, 2, 3, 5, 10):
        r = rt.Renderer(sc, rt.Camera(*CAM, w, h), max_depth=depth, seed=1)
        for _ in range(3):
            r.accumulate(1)
        ks = []
        for _ in range(10):
            r.accumulate(1); s = r.stats(); ks.append(s["kernel_ms"])
        print(f"{w}x{h} depth {depth:2d}: kernel {min(ks):.3f} ms, {s['rays'] / s['paths']:.2f} rays/path, {s['rays'] / min(ks) / 1e6:.1f} Grays/s", flush=True)
        r.close()
