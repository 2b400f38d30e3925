"""Instrumented run: SIMD lane utilisation per code region of the render kernel and the phase clocks
of its main loop, per option set.
  python tools/regions.py [--wl cfg2,cfg3] [--spp N] "-" "xchg=1" ...
(RTHIP_LIBRARY picks a variant build, e.g. one compiled with -DRT_CLOCK_ONLY for undistorted phase shares.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
CAM = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6)
WL = {"cfg2": ("default", 10), "cfg3": ("sample", 50), "cfg4": ("wallpaper", 10), "cfg5": ("synthetic:1000000", 50)}
CAM5 = ((0.0, 60.0, -550.0), (0.0, -60.0, 550.0), 40.0, 10.0, 0.0)
CAM4 = ((0.0, 0.0, 0.0), rt.get_camera_rotation(-90.0, 0.0), 90.0, 10.0, 0.6)
args = sys.argv[1:]
which, spp = ["cfg2", "cfg3"], 16
while args and args[0] in ("--wl", "--spp"):
    if args[0] == "--wl": which = args[1].split(",")
    else: spp = int(args[1])
    args = args[2:]
sets = args or ["-"]
os.environ["RTHIP_COUNT"] = "1"   # read when the context is created
for name in which:
    scene, depth = WL[name]
    sc = rt.Scene.generate(scene)
    for spec in sets:
        tag = spec.replace("=", "").replace(",", "_").replace("-", "base")
        path = f"gpurun_out/regions_{name}_{tag}.txt"
        if os.path.exists(path): os.remove(path)
        os.environ["RTHIP_REGION_DUMP"] = path
        opts = {} if spec == "-" else {k: float(v) for k, v in (kv.split("=") for kv in spec.split(","))}
        cam = rt.Camera(*CAM4, 3840, 2160) if name == "cfg4" else rt.Camera(*CAM5, 3840, 2160) if name == "cfg5" else rt.Camera(*CAM, 1920, 1080)
        r = rt.Renderer(sc, cam, max_depth=depth, seed=1, options=opts, sky_mode=1 if name == "cfg4" else 0)
        r.accumulate(spp)
        s = r.stats()
        print(f"== {name} {spec}: {s['paths'] / s['kernel_ms'] / 1e3:.0f} Mpaths/s (instrumented) threads {s['launch_threads']} waves/CU {s['waves_per_cu']}")
        print(open(path).read())
        r.close()
