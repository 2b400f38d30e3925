"""Device-side builds of the million-object scene (for a rocprofv3 kernel trace): four uploads
(reference pass, primitive records, tree) and three refits."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
sc = rt.Scene.generate("synthetic:1000000", 0x5EED0001)
r = rt.Renderer(sc, rt.Camera((0.0, 60.0, -550.0), (0.0, -60.0, 550.0), 40.0, 10.0, 0.0, 640, 360), max_depth=4, seed=1, options={"tree_gpu": 1})
for _ in range(3):
    r.set_scene(sc)
for _ in range(3):
    r.refit_scene(sc)
r.synchronize()
print(r.stats())
r.close()
