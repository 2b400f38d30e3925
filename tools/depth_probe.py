import os, sys
sys.path.insert(0, "/root/repo")
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
