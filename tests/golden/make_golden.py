"""Generates tests/golden/*.npz from the CPU oracle in FAITHFUL mode (the reference's own
algorithm restated).  The reference has no golden vectors and cannot be built here, so these
fixtures pin the restatement (and through it the HIP kernels) against regressions; they are
data only: inputs (scene name, seeds, camera) and expected outputs.

Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import raytracer_amd as rt  # noqa: E402  (scene factories and camera presets only; no GPU)
from oracle import binding as ob  # noqa: E402
from util import SCENES, SCENE_SEED, camera_args  # noqa: E402

W, H, SPP, SEED, SAMPLE = 48, 27, 3, 7, 3


def main():
    for name in SCENES:
        sc = rt.Scene.generate(name, SCENE_SEED)
        osc = ob.OracleScene(sc.objects)
        cam = ob.camera_desc(*camera_args(name, W, H, rt))
        depth = 50 if name == "sample" else 10
        sky = 1 if name == "wallpaper" else 0
        out = {"objects": sc.objects, "width": W, "height": H, "spp": SPP, "seed": SEED, "probe_sample": SAMPLE,
               "max_depth": depth, "sky_mode": sky, "camera": np.array(camera_args(name, W, H, rt)[:5], dtype=object)}
        for jit in (0, 1):
            ids, t, n, f = osc.trace_primary(cam, ob.FAITHFUL, jitter=bool(jit), seed=SEED, sample=SAMPLE)
            out.update({f"id{jit}": ids, f"t{jit}": t, f"n{jit}": n, f"front{jit}": f})
        accum, smp, cnt, _ = osc.render(cam, SPP, depth, SEED, ob.FAITHFUL, sky_mode=sky, samples=True)
        out["samples"] = smp
        out["accum"] = accum
        out["rays"] = cnt["rays"]
        del out["camera"]
        np.savez_compressed(os.path.join(HERE, f"{name}.npz"), **out)
        print(name, len(sc.objects), "objects,", cnt["rays"], "rays")


if __name__ == "__main__":
    main()
