"""Kernel time of a 1-spp launch against image size: the fixed part of a launch (start-up, scene
copy to LDS, tail) shows as the intercept."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
CAM = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6)
sc = rt.Scene.generate("default")
for (w, h) in ((64, 64), (240, 135), (480, 270), (960, 540), (1920, 1080), (3840, 2160)):
    r = rt.Renderer(sc, rt.Camera(*CAM, w, h), max_depth=10, seed=1)
    for depth in (1, 10):
        r.max_depth = depth
        ks = []
        for _ in range(8):
            r.accumulate(1); ks.append(r.stats()["kernel_ms"])
        print(f"{w}x{h} depth {depth}: kernel {min(ks)*1e3:.0f} us  ({w*h/min(ks)/1e3:.0f} Mpaths/s)", flush=True)
    r.close()
