"""Short fixed workload for rocprofv3: a few launches of the render kernel on the bench
workloads (cfg2 by default; pass cfg3, cfg4 or cfg5 as the first argument):
  python tools/profile_target.py [workload [spp [launches [width height]]]]      (1920x1080 x 32 spp x 3 launches by default)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt

CAM = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6)
which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
launches = int(sys.argv[3]) if len(sys.argv) > 3 else 3
width, height = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (1920, 1080)
sky = 0
if which == "cfg5":   # the million-object scene (global-memory traversal), its own camera, 1080p
    scene, depth = "synthetic:1000000", 50
    CAM = ((0.0, 60.0, -550.0), (0.0, -60.0, 550.0), 40.0, 10.0, 0.0)
elif which == "cfg4":  # the three-object wallpaper scene, its own camera and sky, 1080p
    scene, depth, sky = "wallpaper", 10, 1
    CAM = ((0.0, 0.0, 0.0), rt.get_camera_rotation(-90.0, 0.0), 90.0, 10.0, 0.6)
else:
    scene, depth = ("default", 10) if which == "cfg2" else ("sample", 50)
r = rt.Renderer(rt.Scene.generate(scene), rt.Camera(*CAM, width, height), max_depth=depth, seed=1, sky_mode=sky)
for i in range(launches):
    r.reset_accumulation()
    r.accumulate(spp, spp_begin=0)
    s = r.stats()
    print(f"{which} launch {i}: {s['paths']/s['kernel_ms']/1e3:.1f} Mpaths/s, {s['kernel_ms']:.3f} ms, {s['rays']} rays")
r.close()
