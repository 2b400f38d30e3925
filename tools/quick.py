"""Quick A/B of launch knobs on both bench workloads (interleaved rounds, one process)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
CAM = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6)
variants = [dict(RTHIP_THREADS=t, RTHIP_TRAV_THRESHOLD=th) for t in ("256", "512") for th in ("16", "24")]
if len(sys.argv) > 1:
    variants = [dict(kv.split("=") for kv in a.split(",")) for a in sys.argv[1:]]
for which, scene, depth in (("cfg2", "default", 10), ("cfg3", "sample", 50)):
    r = rt.Renderer(rt.Scene.generate(scene), rt.Camera(*CAM, 1920, 1080), max_depth=depth, seed=1)
    res = {i: [] for i in range(len(variants))}
    for rnd in range(4):
        for i, v in enumerate(variants):
            os.environ.update(v)
            r.reset_accumulation(); r.accumulate(32, spp_begin=0)
            s = r.stats()
            if rnd: res[i].append(s["paths"] / s["kernel_ms"] / 1e3)
            for k in v: os.environ.pop(k)
    for i, v in enumerate(variants):
        print(which, v, "Mpaths/s median %.1f max %.1f" % (sorted(res[i])[len(res[i]) // 2], max(res[i])))
    r.close()
