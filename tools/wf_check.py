"""Wavefront variant vs megakernel: same paths, same sums."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as rt
CAM = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6)
for scene, depth in (("default", 10), ("sample", 50), ("glass", 10)):
    sc = rt.Scene.generate(scene)
    r = rt.Renderer(sc, rt.Camera(*CAM, 96, 54), max_depth=depth, seed=3)
    for prec in (rt.F64, rt.F32):
        r.variant = rt.MEGAKERNEL
        a = r.render_samples(3, precision=prec); sa = r.stats()
        r.variant = rt.WAVEFRONT
        b = r.render_samples(3, precision=prec); sb = r.stats()
        print(scene, "f64" if prec == rt.F64 else "f32", "samples equal:", bool(np.array_equal(a, b)), "max abs", float(np.abs(a - b).max()),
              "rays", sa["rays"], sb["rays"], "paths", sa["paths"], sb["paths"])
        r.precision = prec
        r.variant = rt.MEGAKERNEL; r.reset_accumulation(); r.accumulate(5); x = r.read_accum()
        r.variant = rt.WAVEFRONT; r.reset_accumulation(); r.accumulate(5); y = r.read_accum()
        print("   accum equal:", bool(np.array_equal(x, y)), float(np.abs(x - y).max()))
    r.close()
for which, scene, depth in (("cfg2", "default", 10), ("cfg3", "sample", 50)):
    r = rt.Renderer(rt.Scene.generate(scene), rt.Camera(*CAM, 1920, 1080), max_depth=depth, seed=1)
    for variant, name in ((rt.MEGAKERNEL, "mega"), (rt.WAVEFRONT, "wavefront")):
        r.variant = variant
        best = None
        for _ in range(3):
            r.reset_accumulation()
            t0 = time.perf_counter(); r.accumulate(32, spp_begin=0); dt = time.perf_counter() - t0
            s = r.stats()
            best = s if best is None or s["kernel_ms"] < best["kernel_ms"] else best
        print(which, name, f"{best['paths']/best['kernel_ms']/1e3:.0f} Mpaths/s  kernel {best['kernel_ms']:.2f} ms  wall {dt*1e3:.2f} ms rays {best['rays']}")
    r.close()
