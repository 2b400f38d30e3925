import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
CAM = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6)
which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
scene, depth = ("default", 10) if which == "cfg2" else ("sample", 50)
r = rt.Renderer(rt.Scene.generate(scene), rt.Camera(*CAM, 1920, 1080), max_depth=depth, seed=1, variant=rt.WAVEFRONT)
for i in range(2):
    r.reset_accumulation(); r.accumulate(16, spp_begin=0)
    s = r.stats(); print(which, f"{s['paths']/s['kernel_ms']/1e3:.0f} Mpaths/s {s['kernel_ms']:.2f} ms")
r.close()
