"""Global-memory traversal against scene size: throughput and time per node visit as the tree
outgrows L2 (4 MB per XCD) and the Infinity Cache (256 MB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
cam = rt.Camera((0.0, 60.0, -550.0), (0.0, -60.0, 550.0), 40.0, 10.0, 0.0, 1920, 1080)
for n in (3000, 30000, 100000, 300000, 1000000, 3000000):
    sc = rt.create_synthetic_scene(n, seed=0x5EED0001)
    r = rt.Renderer(sc, cam, max_depth=50, seed=1, options={"scene_lds": 0})
    r.set_option("count", 1)
    r.accumulate(4, spp_begin=0); c = r.stats()
    r.set_option("count", 0)
    best = None
    for _ in range(2):
        r.reset_accumulation(); r.accumulate(8, spp_begin=0); s = r.stats()
        best = s if best is None or s["kernel_ms"] < best["kernel_ms"] else best
    visits = c["box_tests"] / 2 / c["rays"]
    prims = (c["sphere_tests"] + c["cube_tests"]) / c["rays"]
    lane_visits_per_s = best["rays"] * visits / (best["kernel_ms"] * 1e-3)
    print(f"{n:8d} objects: nodes {c['n_nodes']:8d} ({c['n_nodes']*64/1e6:7.1f} MB) depth {c['tree_depth']:2d}  {best['paths']/best['kernel_ms']/1e3:7.0f} Mpaths/s "
          f"{best['rays']/best['kernel_ms']/1e3:7.0f} Mrays/s  visits/ray {visits:5.1f} prims/ray {prims:4.1f}  {lane_visits_per_s/1e9:6.1f} G node visits/s "
          f"= {lane_visits_per_s*64/1e12:5.2f} TB/s of node records", flush=True)
    r.close()
