"""Option sets must not change a single word of the image: renders the two bench scenes at a small
size under each set and compares the accumulation buffers with the defaults' (paths are
deterministic, accumulation is integer).
  python tools/ab_equal.py [--big] "stack_mem=1" "stack_mem=1,threads=512" ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as rt
CAM = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6)
sets = sys.argv[1:]
bad = 0
CAM5 = ((0.0, 60.0, -550.0), (0.0, -60.0, 550.0), 40.0, 10.0, 0.0)
scenes = [("default", 10), ("sample", 50), ("glass", 10), ("wallpaper", 10)]
if sets and sets[0] == "--big":   # scenes that stay in global memory as well (quantised nodes)
    sets = sets[1:]
    scenes += [("synthetic:20000", 50), ("synthetic:1000000", 50)]
for scene, depth in scenes:
    sc = rt.Scene.generate(scene)
    if scene.startswith("synthetic"): CAM = CAM5
    ref = None
    for spec in ["-"] + sets:
        opts = {} if spec == "-" else {k: float(v) for k, v in (kv.split("=") for kv in spec.split(","))}
        r = rt.Renderer(sc, rt.Camera(*CAM, 640, 360), max_depth=depth, seed=3, options=opts)
        r.accumulate(24)
        img = r.read_accum()
        s = r.stats()
        r.close()
        if ref is None:
            ref = img
            continue
        same = np.array_equal(img, ref)
        bad += 0 if same else 1
        print(f"{scene:8s} {spec:40s} {'identical' if same else 'DIFFERENT: %d values' % int((img != ref).sum())}  rays {s['rays']}", flush=True)
sys.exit(1 if bad else 0)
