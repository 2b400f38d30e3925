"""Instrumented run: SIMD lane utilisation per code region of the render kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
CAM = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6)
os.environ["RTHIP_COUNT"] = "1"   # read when the context is created
for which, scene, depth in (("cfg2", "default", 10), ("cfg3", "sample", 50)):
    path = f"gpurun_out/regions_{which}.txt"
    if os.path.exists(path): os.remove(path)
    os.environ["RTHIP_REGION_DUMP"] = path
    r = rt.Renderer(rt.Scene.generate(scene), rt.Camera(*CAM, 1920, 1080), max_depth=depth, seed=1)
    r.accumulate(16)
    s = r.stats()
    print(which, s)
    print(open(path).read())
    r.close()
