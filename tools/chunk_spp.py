"""Work-item size (samples of a tile per item) against samples per launch, both bench scenes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
CAM = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6)
for which, scene, depth in (("cfg2", "default", 10), ("cfg3", "sample", 50)):
    r = rt.Renderer(rt.Scene.generate(scene), rt.Camera(*CAM, 1920, 1080), max_depth=depth, seed=1)
    for spp in (16, 64, 128, 256, 500):
        row = []
        for chunk in (2, 4, 8, 16, 32):
            os.environ["RTHIP_CHUNK"] = str(chunk)
            best = 0.0
            for _ in range(2 if spp > 100 else 3):
                r.reset_accumulation(); r.accumulate(spp, spp_begin=0); s = r.stats()
                best = max(best, s["paths"] / s["kernel_ms"] / 1e3)
            row.append(f"chunk {chunk}: {best:6.0f}")
        print(which, f"spp {spp:3d} |", "  ".join(row), flush=True)
    r.close()
