"""Ray and test counts of several builds of librthip on both bench workloads (one subprocess per build)."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, json
sys.path.insert(0, %r)
import raytracer_amd as rt
CAM = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6)
os.environ["RTHIP_COUNT"] = "1"
for which, scene, depth in (("cfg2", "default", 10), ("cfg3", "sample", 50)):
    r = rt.Renderer(rt.Scene.generate(scene), rt.Camera(*CAM, 1920, 1080), max_depth=depth, seed=1)
    r.accumulate(8); s = r.stats()
    print(which, {k: s[k] for k in ("rays", "box_tests", "sphere_tests", "cube_tests")})
    r.close()
''' % ROOT
for l in sys.argv[1:]:
    env = dict(os.environ, RTHIP_LIBRARY=os.path.join(ROOT, "raytracer_amd", l))
    o = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    print(l, o.stdout.strip(), o.stderr[-300:] if o.returncode else "")
