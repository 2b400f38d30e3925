"""A/B of option sets (rthip_set_option names; the old per-variable sweeps in one tool) on bench
workloads, one process:
  python tools/ab_opts.py [--wl cfg2,cfg3,cfg5] [--spp N] [--size WxH] "sah_cprim=0.4,mats_lds=0" "threads=768" ...   ('-' = defaults)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
CAM = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6)
CAM5 = ((0.0, 60.0, -550.0), (0.0, -60.0, 550.0), 40.0, 10.0, 0.0)
# any other name is taken as a scene of rt.Scene.generate (e.g. synthetic:1500) under the cfg5 camera
CAM4 = ((0.0, 0.0, 0.0), rt.get_camera_rotation(-90.0, 0.0), 90.0, 10.0, 0.6)   # cfg4: the wallpaper view (pass --size 3840x2160 for its own frame)
WL = {"cfg4": ("wallpaper", 10, CAM4, 32), "cfg2": ("default", 10, CAM, 32), "cfg3": ("sample", 50, CAM, 32), "cfg5": ("synthetic:1000000", 50, CAM5, 8)}
args = sys.argv[1:]
which = ["cfg2", "cfg3"]
if args and args[0] == "--wl":
    which, args = args[1].split(","), args[2:]
spp_override = None
if args and args[0] == "--spp":
    spp_override, args = int(args[1]), args[2:]
size = (1920, 1080)
if args and args[0] == "--size":
    size, args = tuple(int(v) for v in args[1].split("x")), args[2:]
sets = args or ["-"]
for name in which:
    scene, depth, cam, spp = WL.get(name, (name, 50, CAM5, 8))
    spp = spp_override or spp
    sc = rt.Scene.generate(scene)
    for spec in sets:
        opts = {} if spec == "-" else {k: float(v) for k, v in (kv.split("=") for kv in spec.split(","))}
        r = rt.Renderer(sc, rt.Camera(*cam, *size), max_depth=depth, seed=1, options=opts, sky_mode=1 if name == "cfg4" else 0)
        best = 0
        for _ in range(4):
            r.reset_accumulation(); r.accumulate(spp, spp_begin=0)
            s = r.stats(); best = max(best, s["paths"] / s["kernel_ms"] / 1e3)
        print(f"{name} spp {spp} {spec:40s} {best:8.0f} Mpaths/s  threads {s['launch_threads']} waves/CU {s['waves_per_cu']} lds {s['launch_lds_bytes']} nodes {s['n_nodes']} prims {s['n_prims']} depth {s['tree_depth']} in_lds {s['scene_in_lds']}", flush=True)
        r.close()
