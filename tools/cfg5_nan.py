import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as rt
sc = rt.create_synthetic_scene(1000000, seed=0x5EED0001)
w, h = 1920, 1080
cam = rt.Camera((0.0, 60.0, -550.0), (0.0, -60.0, 550.0), 40.0, 10.0, 0.0, w, h)
r = rt.Renderer(sc, cam, max_depth=50, seed=1)
for prec, name in ((rt.F32, "f32"), (rt.F64, "f64")):
    for depth in (1, 2, 3, 5, 50):
        r.max_depth = depth
        t0 = time.perf_counter(); s = r.render_samples(4, precision=prec); dt = time.perf_counter() - t0
        bad = ~np.isfinite(s).all(axis=-1)
        print(name, "depth", depth, "shape", s.shape, "nonfinite samples", int(bad.sum()), "where", np.argwhere(bad)[:5].tolist(), f"kernel {r.stats()['kernel_ms']:.1f} ms", flush=True)
r.close()
