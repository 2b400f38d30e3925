"""Converged-image check (the third correctness bar): the f32 GPU render against the CPU oracle
at high spp, (a) on the same random stream - differences are f32 arithmetic only - and (b) with
an independent seed - differences are Monte-Carlo noise, which bounds any bias."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import raytracer_amd as rt
from oracle import binding as ob
from util import camera_args

W, H = 320, 180
for scene, depth, spp in (("default", 10, 1024), ("sample", 50, 1024), ("glass", 10, 1024), ("metal", 10, 1024)):
    sc = rt.Scene.generate(scene)
    cam = camera_args(scene, W, H, rt)
    r = rt.Renderer(sc, rt.Camera(*cam), max_depth=depth, seed=7)
    r.accumulate(spp); gpu = np.clip(r.read_accum() / spp, 0, 1)
    r.seed = 8; r.reset_accumulation(); r.accumulate(spp); gpu2 = np.clip(r.read_accum() / spp, 0, 1)
    osc = ob.OracleScene(sc.objects)
    t0 = time.time()
    ref, _, _, _ = osc.render(ob.camera_desc(*cam), spp, depth, 7, ob.FAITHFUL, threads=16)
    ref = np.clip(ref / spp, 0, 1)
    def stats(a, b):
        rmse = float(np.sqrt(((a - b) ** 2).mean()))
        return rmse, (10 * np.log10(1.0 / rmse ** 2) if rmse > 0 else float("inf")), float((a - b).mean())
    same, indep = stats(gpu, ref), stats(gpu2, ref)
    q, qr = r.read_rgb8(), ob.quantize_rgb8(ref)
    print(f"{scene:8s} {W}x{H} {spp} spp depth {depth}: same stream RMSE {same[0]:.2e} (PSNR {same[1]:.1f} dB, mean diff {same[2]:+.1e}); "
          f"independent seed RMSE {indep[0]:.2e} (PSNR {indep[1]:.1f} dB, mean diff {indep[2]:+.1e}); CPU reference {time.time()-t0:.1f} s", flush=True)
    r.close()
