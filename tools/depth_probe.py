"""Render time of a 32-spp 1080p launch of the two bench scenes against the depth limit (1, 2, 3, 4 and the
workload's own), megakernel and wavefront variant: what the first segment of a path costs beside the later ones
(a depth-1 render is camera rays only).  python tools/depth_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
CAM = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6)
for scene, full in (("default", 10), ("sample", 50)):
    sc = rt.Scene.generate(scene)
    for depth in (1, 2, 3, 4, full):
        for variant in (rt.MEGAKERNEL, rt.WAVEFRONT):
            r = rt.Renderer(sc, rt.Camera(*CAM, 1920, 1080), max_depth=depth, seed=1)
            r.variant = variant
            best = 1e9
            for _ in range(3):
                r.reset_accumulation(); r.accumulate(32, spp_begin=0)
                s = r.stats(); best = min(best, s["kernel_ms"])
            print(f"{scene} depth {depth} {'mega' if variant == rt.MEGAKERNEL else 'wave'}: {best:.3f} ms, {s['rays']/s['paths']:.3f} rays/path, {s['rays']/best/1e6:.2f} Grays/s", flush=True)
            r.close()
