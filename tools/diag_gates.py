"""Diagnosis: primary rays from INSIDE the glass layer of hollow shells (between the outer sphere
and the inverted inner one), where the closest hit is the inner sphere and its gates decide.
Compares the f64 probe with the oracle ray by ray and prints the first mismatches."""
import sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import raytracer_amd as rt
from oracle import binding as ob
from test_gpu_configs import hollow_shells

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
sc = hollow_shells(rt, n)
osc = ob.OracleScene(sc.objects)
objs = sc.objects
bad_total = 0
printed = 0
r = None
for k in range(0, 2 * n, max(2, (2 * n // 300) // 2 * 2)):
    c, rad = objs["center"][k], objs["extent"][k, 0]
    d = np.array([0.3, -0.5, 0.8]); d /= np.linalg.norm(d)
    origin = tuple(c - d * 0.975 * rad)
    cam = (origin, tuple(d), 60.0, 10.0, 0.0)
    w, h = 24, 24
    camera = rt.Camera(*cam, w, h)
    if r is None:
        r = rt.Renderer(sc, camera, seed=3, precision=rt.F64, max_depth=12)
    else:
        r.update_camera(camera)
    desc = ob.camera_desc(*cam, w, h)
    ref = osc.trace_primary(desc, ob.TIGHT, jitter=False, seed=3, sample=0)
    fa = osc.trace_primary(desc, ob.FAITHFUL, jitter=False, seed=3, sample=0)
    got = r.trace_primary(rt.F64, jitter=False, sample=0)
    assert np.array_equal(ref[0], fa[0])
    bad = np.argwhere(got[0] != ref[0])
    bad_total += len(bad)
    if len(bad) and printed < 25:
        printed += 1
        y, x = bad[0]
        print(f"shell {k}: {len(bad)} of {w*h} ids differ; first at ({x},{y}): gpu id {got[0][y,x]} t {got[1][y,x]:.6f}  oracle id {ref[0][y,x]} t {ref[1][y,x]:.6f}; inner id {k+1}", flush=True)
print("total mismatching rays", bad_total, "gates", r.stats()["n_gates"])
