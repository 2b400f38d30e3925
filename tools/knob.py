"""One tuning knob (an environment variable the library reads at every launch) swept on both
bench workloads: python tools/knob.py NAME v1 v2 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
CAM = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6)
name, vals = sys.argv[1], sys.argv[2:]
for which, scene, depth in (("cfg2", "default", 10), ("cfg3", "sample", 50)):
    r = None
    for v in vals:
        os.environ[name] = v
        if r is None or name.startswith(("RTHIP_SAH", "RTHIP_TREE", "RTHIP_BIG")):  # read at scene upload
            if r is not None:
                r.close()
            r = rt.Renderer(rt.Scene.generate(scene), rt.Camera(*CAM, 1920, 1080), max_depth=depth, seed=1)
        best = 0.0
        for _ in range(4):
            r.reset_accumulation(); r.accumulate(32, spp_begin=0); s = r.stats()
            best = max(best, s["paths"] / s["kernel_ms"] / 1e3)
        print(f"{which} {name}={v}: {best:.0f} Mpaths/s", flush=True)
    r.close()
