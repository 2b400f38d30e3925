"""The two larger measurement configurations: the three-object 4K scene and the synthetic
million-object 8K scene (global-memory traversal).  Prints upload time and throughput."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt

which = sys.argv[1:] or ["cfg4", "cfg5"]
if "cfg4" in which:
    for defocus in (0.0, 0.6):
        cam = rt.Camera((0.0, 0.0, 0.0), rt.get_camera_rotation(-90.0, 0.0), 90.0, 10.0, defocus, 3840, 2160)
        r = rt.Renderer(rt.Scene.generate("wallpaper"), cam, max_depth=10, seed=1, sky_mode=1)
        r.accumulate(8)
        best = None
        for _ in range(3):
            r.reset_accumulation(); r.accumulate(64, spp_begin=0); s = r.stats()
            best = s if best is None or s["kernel_ms"] < best["kernel_ms"] else best
        print(f"cfg4 defocus {defocus}: {best['paths']/best['kernel_ms']/1e3:.0f} Mpaths/s  {best['rays']/best['kernel_ms']/1e3:.0f} Mrays/s  "
              f"kernel {best['kernel_ms']:.2f} ms (64 spp at 3840x2160)", flush=True)
        r.close()
if "cfg5" in which:
    n = int(os.environ.get("CFG5_OBJECTS", "1000000"))
    t0 = time.perf_counter(); sc = rt.create_synthetic_scene(n, seed=0x5EED0001); t1 = time.perf_counter()
    cam = rt.Camera((0.0, 60.0, -550.0), (0.0, -60.0, 550.0), 40.0, 10.0, 0.0, 7680, 4320)
    r = rt.Renderer(sc, cam, max_depth=50, seed=1); t2 = time.perf_counter()
    print(f"cfg5: generate {t1-t0:.2f} s, upload+build {t2-t1:.2f} s, objects {len(sc.objects)}", flush=True)
    for spp in (1, 4, 16):
        r.reset_accumulation(); r.accumulate(spp, spp_begin=0); s = r.stats()
        print(f"cfg5 spp {spp}: {s['paths']/s['kernel_ms']/1e3:.0f} Mpaths/s  {s['rays']/s['kernel_ms']/1e3:.0f} Mrays/s  kernel {s['kernel_ms']:.1f} ms  "
              f"rays/path {s['rays']/s['paths']:.2f} in_lds {s['scene_in_lds']}", flush=True)
    r.close()
