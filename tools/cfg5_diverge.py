import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as rt
sc = rt.create_synthetic_scene(1000000, seed=0x5EED0001)
w, h = 960, 540
cam = rt.Camera((0.0, 60.0, -550.0), (0.0, -60.0, 550.0), 40.0, 10.0, 0.0, w, h)
r = rt.Renderer(sc, cam, max_depth=50, seed=1)
ids32 = r.trace_primary(rt.F32, jitter=True, sample=0); ids64 = r.trace_primary(rt.F64, jitter=True, sample=0)
print("primary id mismatch", float((ids32[0] != ids64[0]).mean()), "hit fraction", float((ids64[0] >= 0).mean()))
for depth in (1, 2, 3, 4, 6, 10, 50):
    r.max_depth = depth
    a = r.render_samples(2, precision=rt.F32).astype(np.float64); b = r.render_samples(2, precision=rt.F64)
    d = np.abs(a - b).max(axis=-1)
    print("depth", depth, "frac |diff|>1e-3:", float((d > 1e-3).mean()), "frac >1e-2:", float((d > 1e-2).mean()), "mean|b|", float(np.abs(b).mean()), flush=True)
r.close()
