"""Converged images at the bench's full size: the f32 production path against the f64 path of the
same library (which is bit-exact against the oracle) on the same random stream, 1920x1080 at
1024 spp, both bench scenes.  Differences are f32 arithmetic only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import raytracer_amd as rt
from util import camera_args

W, H, SPP = 1920, 1080, 1024
for scene, depth in (("default", 10), ("sample", 50)):
    sc = rt.Scene.generate(scene)
    cam = camera_args(scene, W, H, rt)
    imgs = {}
    for name, prec in (("f32", rt.F32), ("f64", rt.F64)):
        r = rt.Renderer(sc, rt.Camera(*cam), max_depth=depth, seed=7, precision=prec)
        t0 = time.time()
        for s0 in range(0, SPP, 128):
            r.accumulate(128, spp_begin=s0)
            print(f"  {scene} {name} {s0 + 128}/{SPP} spp", flush=True)
        imgs[name] = (np.clip(r.read_accum() / SPP, 0, 1), r.read_rgb8(SPP), time.time() - t0)
        r.close()
    a, b = imgs["f32"][0], imgs["f64"][0]
    rmse = float(np.sqrt(((a - b) ** 2).mean()))
    differ = (imgs["f32"][1] != imgs["f64"][1])
    print(f"{scene:8s} {W}x{H} {SPP} spp depth {depth}: f32 vs f64 RMSE {rmse:.2e} (PSNR {10*np.log10(1/rmse**2):.1f} dB), mean diff {float((a-b).mean()):+.1e}, "
          f"max abs diff {float(np.abs(a-b).max()):.2e}; 8-bit values that differ: {int(differ.sum())} of {differ.size} ({differ.mean()*100:.3f} %), none by more than {int(np.abs(imgs['f32'][1].astype(int)-imgs['f64'][1].astype(int)).max())}; "
          f"f32 {imgs['f32'][2]:.1f} s, f64 {imgs['f64'][2]:.1f} s", flush=True)
