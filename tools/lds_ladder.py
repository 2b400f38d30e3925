"""Scene-size sweep across the LDS residency cliff: the default build (tree fineness chosen so
that the CU stays full) against a fixed leaf cost: python tools/lds_ladder.py [sizes...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
sizes = [int(v) for v in sys.argv[1:]] or [300, 500, 600, 700, 800, 1000, 1300]
cam = rt.Camera((0.0, 60.0, -550.0), (0.0, -60.0, 550.0), 40.0, 10.0, 0.0, 1920, 1080)
for n in sizes:
    sc = rt.create_synthetic_scene(n, seed=0x5EED0001)
    row = []
    for fixed in (None, "0.4", "0.7"):
        if fixed is None: os.environ.pop("RTHIP_SAH_CPRIM", None)
        else: os.environ["RTHIP_SAH_CPRIM"] = fixed
        r = rt.Renderer(sc, cam, max_depth=50, seed=1)
        best = 0.0
        for _ in range(3):
            r.reset_accumulation(); r.accumulate(32, spp_begin=0); s = r.stats()
            best = max(best, s["paths"] / s["kernel_ms"] / 1e3)
        row.append(f"{fixed or 'auto'}: {best:6.0f} Mpaths/s ({s['n_nodes']} nodes, depth {s['tree_depth']})")
        r.close()
    print(f"{n:5d} objects | " + " | ".join(row), flush=True)
