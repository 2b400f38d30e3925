import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
CAM = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6)
variants = [dict(RTHIP_SCENE_LDS=l, RTHIP_THREADS=t, RTHIP_SAH_CPRIM=c) for l in ("1", "0") for t in ("256", "512") for c in ("0.4", "0.7")]
for which, scene, depth in (("cfg2", "default", 10), ("cfg3", "sample", 50)):
    for v in variants:
        os.environ.update(v)
        r = rt.Renderer(rt.Scene.generate(scene), rt.Camera(*CAM, 1920, 1080), max_depth=depth, seed=1)
        best = 0
        for _ in range(3):
            r.reset_accumulation(); r.accumulate(32, spp_begin=0)
            s = r.stats(); best = max(best, s["paths"] / s["kernel_ms"] / 1e3)
        print(which, v, f"{best:.0f} Mpaths/s lds={s['scene_in_lds']} nodes {s['n_nodes']} depth {s['tree_depth']}")
        for k in v: os.environ.pop(k)
        r.close()
