"""Throughput of the f64 parity mode (the reference's own arithmetic on the GPU) on both bench scenes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
CAM = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6)
for which, scene, depth in (("cfg2", "default", 10), ("cfg3", "sample", 50)):
    r = rt.Renderer(rt.Scene.generate(scene), rt.Camera(*CAM, 1920, 1080), max_depth=depth, seed=1, precision=rt.F64)
    best = 0
    for _ in range(3):
        r.reset_accumulation(); r.accumulate(16, spp_begin=0); s = r.stats(); best = max(best, s["paths"]/s["kernel_ms"]/1e3)
    print(which, "f64", round(best), "Mpaths/s in_lds", s["scene_in_lds"])
    r.close()
