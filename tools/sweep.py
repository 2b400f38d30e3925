"""Tuning sweep on one GPU: throughput of the render kernel under the library's tuning knobs
(environment variables read at every launch)."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as rt

CAM = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6)

def run(r, spp, reps=3):
    best = None
    for _ in range(reps):
        r.reset_accumulation()
        r.accumulate(spp, spp_begin=0)
        s = r.stats()
        best = s if best is None or s["kernel_ms"] < best["kernel_ms"] else best
    return best

def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
    scene, depth, spp = ("default", 10, 32) if which == "cfg2" else ("sample", 50, 32)
    sc = rt.Scene.generate(scene)
    r = rt.Renderer(sc, rt.Camera(*CAM, 1920, 1080), max_depth=depth, seed=1)
    knobs = {
        "RTHIP_TRAV_THRESHOLD": ["0", "8", "16", "24", "32", "40", "48", "56"],
        "RTHIP_SCENE_LDS": ["1", "0"],
        "RTHIP_CHUNK": ["2", "4", "8", "16", "32"],
        "RTHIP_BLOCKS_PER_CU": ["1", "2", "3", "4", "6", "8"],
    }
    base = {"RTHIP_TRAV_THRESHOLD": "24", "RTHIP_SCENE_LDS": "1", "RTHIP_CHUNK": "8", "RTHIP_BLOCKS_PER_CU": "0"}
    for k, v in base.items():
        os.environ[k] = v
    s = run(r, spp)
    print(f"{which} base {base}: {s['paths']/s['kernel_ms']/1e3:8.1f} Mpaths/s  {s['rays']/s['kernel_ms']/1e3:8.1f} Mrays/s  {s['kernel_ms']:.2f} ms")
    for k, vals in knobs.items():
        for v in vals:
            os.environ[k] = v
            s = run(r, spp)
            print(f"  {k}={v:3s}: {s['paths']/s['kernel_ms']/1e3:8.1f} Mpaths/s  {s['kernel_ms']:.2f} ms  lds={s['scene_in_lds']}")
        os.environ[k] = base[k]
    r.close()

main()
