"""Work-item size (samples of a tile per item) against samples per launch, both bench scenes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
CAM = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6)
WALL = ((0.0, 0.0, 0.0), rt.get_camera_rotation(-90.0, 0.0), 90.0, 10.0, 0.0)
for which, scene, depth, cam, size, sky in (("cfg2", "default", 10, CAM, (1920, 1080), 0), ("cfg3", "sample", 50, CAM, (1920, 1080), 0),
                                            ("cfg4", "wallpaper", 10, WALL, (3840, 2160), 1), ("cfg4@1080p", "wallpaper", 10, WALL, (1920, 1080), 1)):
    r = rt.Renderer(rt.Scene.generate(scene), rt.Camera(*cam, *size), max_depth=depth, seed=1, sky_mode=sky)
    for spp in (4, 16, 32, 100, 256, 500):
        row = []
        for chunk in (2, 4, 8, 16, 32):
            r.set_option("chunk", chunk)
            best = 0.0
            for _ in range(2 if spp > 100 else 3):
                r.reset_accumulation(); r.accumulate(spp, spp_begin=0); s = r.stats()
                best = max(best, s["paths"] / s["kernel_ms"] / 1e3)
            row.append(f"chunk {chunk}: {best:6.0f}")
        print(which, f"spp {spp:3d} |", "  ".join(row), flush=True)
    r.close()
