"""The progressive viewer's case: one sample per pixel per launch at 1080p (cfg2 scene).
Kernel time (HIP events) and wall time per frame, for a few launch shapes (option sets)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
CAM = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6)
r = rt.Renderer(rt.Scene.generate("default"), rt.Camera(*CAM, 1920, 1080), max_depth=10, seed=1)
DEFAULTS = {"threads": 0, "blocks_per_cu": 0, "regen_min": 12, "trav_threshold": -1, "mats_lds": 1, "chunk": 0}
sets = [{}] + [dict(kv.split("=") for kv in a.split(",")) for a in sys.argv[1:]]
for env in sets:
    for k, v in {**DEFAULTS, **env}.items():
        r.set_option(k, float(v))
    for spp in (1, 2, 4):
        r.reset_accumulation()
        for _ in range(3):
            r.accumulate(spp)
        ks = []
        for _ in range(10):
            r.accumulate(spp); ks.append(r.stats()["kernel_ms"])
        t0 = time.perf_counter()
        for _ in range(20):
            r.accumulate(spp, sync=False)
        r.synchronize()
        wall = (time.perf_counter() - t0) / 20 * 1e3
        print(f"{env or 'default'} spp {spp}: kernel {min(ks):.3f} ms (median {sorted(ks)[5]:.3f}), wall {wall:.3f} ms per frame, {1920*1080*spp/min(ks)/1e3:.0f} Mpaths/s", flush=True)
r.close()
