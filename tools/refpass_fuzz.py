"""Differential run of the device's reference pass (tie ranks, gates: refpass_gpu.hip) against the
host's pass over the reference's tree, on random scenes made to produce ties and gates: object
counts around the lengths where the range kernels and the level count change, duplicated centres,
keys equal on the sort axis, hollow shells parted from their cores.  Compares gate counts and the
f64 image words of both after an upload and after a refit.
  python tools/refpass_fuzz.py [scenes [seed]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as rt
from raytracer_amd import _native as N

CAM = ((9.0, 4.0, 11.0), (-9.0, -3.5, -11.0), 40.0, 10.0, 0.2, 64, 36)


def scene(g):
    n = int(g.choice([1, 2, 5, 6, 7, 11, 12, 23, 24, 47, 96, 191, 192, 193, 385, 386, 770, 1537, 3000, 6145, 20000]) + g.integers(0, 3))
    o = np.zeros(n, dtype=N.OBJECT_DTYPE)
    o["kind"] = np.where(g.random(n) < 0.6, N.SPHERE, N.CUBE)
    o["material"] = g.integers(0, 3, n)
    o["param"] = np.where(o["material"] == N.DIELECTRIC, 1.5, 0.2)
    o["albedo"] = g.uniform(0.2, 0.9, (n, 3))
    mode = g.integers(0, 4)
    c = g.uniform(-6, 6, (n, 3)) * [1.0, 0.3, 1.0] + [0, 1, 0]
    if mode == 1:      # few distinct centres
        c = c[g.integers(0, max(1, n // 8), n)]
    elif mode == 2:    # a lattice: equal keys on every axis
        c = np.round(c)
    elif mode == 3:    # pairs share a centre (shell and core)
        c[1::2] = c[0:n - 1:2][: len(c[1::2])]
    o["center"] = c
    e = g.uniform(0.05, 0.6, (n, 3))
    if mode == 2:
        e = np.round(e * 4) / 4 + 0.25
    o["extent"] = e
    neg = (o["kind"] == N.SPHERE) & (g.random(n) < 0.3)
    o["extent"][neg, 0] *= -0.9
    bad_cube = (o["kind"] == N.CUBE) & (g.random(n) < 0.05)
    o["extent"][bad_cube, 1] *= -1.0
    return rt.Scene(o)


def words(r):
    r.reset_accumulation()
    r.accumulate(1, spp_begin=0)
    _, n = r.accum_buffer()
    w = np.zeros(n, dtype=np.uint64)
    r._check(r._lib.rthip_accum_download(r._ctx, w.ctypes.data, n))
    return w


count = int(sys.argv[1]) if len(sys.argv) > 1 else 60
g = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for k in range(count):
    sc = scene(g)
    moved = rt.Scene(sc.objects.copy())
    moved.objects["center"] += g.normal(0, 0.3, moved.objects["center"].shape) * (g.random((len(sc), 1)) < 0.5)
    tree = {"tree_gpu": 1} if k % 2 else {}
    dev = rt.Renderer(sc, rt.Camera(*CAM), seed=3, precision=rt.F64, max_depth=6, options=dict(tree, ref_gpu=1))
    host = rt.Renderer(sc, rt.Camera(*CAM), seed=3, precision=rt.F64, max_depth=6, options=dict(tree, ref_gpu=0))
    for step in range(2):
        same = dev.stats()["n_gates"] == host.stats()["n_gates"] and np.array_equal(words(dev), words(host))
        if not same:
            bad += 1
            print(f"scene {k} ({len(sc)} objects, step {step}): gates {dev.stats()['n_gates']} / {host.stats()['n_gates']}: DIFFERENT", flush=True)
        if step == 0:
            dev.refit_scene(moved); host.refit_scene(moved)
    print(f"scene {k}: {len(sc)} objects, {host.stats()['n_gates']} gate boxes, {'ok' if same else 'DIFFERENT'}", flush=True)
    dev.close(); host.close()
print(f"{count} scenes, {bad} differences")
sys.exit(1 if bad else 0)
