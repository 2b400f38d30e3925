"""One tuning knob swept on the million-object scene at 1080p: python tools/cfg5_knob.py NAME v1 v2 ..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
name, vals = sys.argv[1], sys.argv[2:]
sc = rt.create_synthetic_scene(int(os.environ.get("CFG5_OBJECTS", "1000000")), seed=0x5EED0001)
cam = rt.Camera((0.0, 60.0, -550.0), (0.0, -60.0, 550.0), 40.0, 10.0, 0.0, 1920, 1080)
r = None
for v in vals:
    os.environ[name] = v
    if r is None or name.startswith(("RTHIP_SAH", "RTHIP_TREE")):
        if r is not None:
            r.close()
        r = rt.Renderer(sc, cam, max_depth=50, seed=1)
    best = 0.0
    for _ in range(3):
        r.reset_accumulation(); r.accumulate(16, spp_begin=0); s = r.stats()
        best = max(best, s["paths"] / s["kernel_ms"] / 1e3)
    print(f"cfg5 {name}={v}: {best:.0f} Mpaths/s  ({s['rays']/s['paths']:.2f} rays/path, nodes {s['n_nodes']}, depth {s['tree_depth']})", flush=True)
r.close()
