"""Device-tree A/B: reference-shaped vs SAH (and SAH leaf sizes) on both bench workloads."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
CAM = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6)
variants = [dict(tree_ref=1)] + [dict(tree_ref=0, sah_leaf=k, sah_cprim=c) for k in (1, 2, 4, 6) for c in (0.7,)]
variants += [dict(tree_ref=0, sah_leaf=4, sah_cprim=c) for c in (0.4, 1.0, 1.5)]
for which, scene, depth in (("cfg2", "default", 10), ("cfg3", "sample", 50)):
    for v in variants:
        r = rt.Renderer(rt.Scene.generate(scene), rt.Camera(*CAM, 1920, 1080), max_depth=depth, seed=1, options=v)
        best = 0
        for _ in range(3):
            r.reset_accumulation(); r.accumulate(32, spp_begin=0)
            s = r.stats(); best = max(best, s["paths"] / s["kernel_ms"] / 1e3)
        r.set_option("count", 1)
        r.reset_accumulation(); r.accumulate(4, spp_begin=0)
        c = r.stats()
        print(which, v, f"{best:.0f} Mpaths/s | nodes {c['n_nodes']} depth {c['tree_depth']} prims {c['n_prims']} | per ray: box {c['box_tests']/c['rays']:.1f} sphere {c['sphere_tests']/c['rays']:.2f} cube {c['cube_tests']/c['rays']:.2f}")
        r.close()
