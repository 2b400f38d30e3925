"""Host-to-host cost of one frame through the boundary: scene upload (host tree build + copy),
render, and the read-back of the 8-bit image into host memory, on the default bench workload."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
CAM = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6)
sc = rt.Scene.generate("default")
t0 = time.perf_counter()
r = rt.Renderer(sc, rt.Camera(*CAM, 1920, 1080), max_depth=10, seed=1)
t1 = time.perf_counter()
r.accumulate(100); img = r.read_rgb8()  # warm-up
for spp in (100, 1):
    best = None
    for _ in range(5):
        r.reset_accumulation()
        a = time.perf_counter(); r.accumulate(spp); b = time.perf_counter(); img = r.read_rgb8(); c = time.perf_counter()
        if best is None or c - a < best[0]: best = (c - a, b - a, c - b, r.stats()["kernel_ms"])
    paths = 1920 * 1080 * spp
    print(f"{spp:3d} spp: render call {best[1]*1e3:.2f} ms (kernel {best[3]:.2f} ms), read_rgb8 to host {best[2]*1e3:.2f} ms, "
          f"host-to-host {best[0]*1e3:.2f} ms = {paths/best[0]/1e6:.0f} Mpaths/s against {paths/best[3]/1e3:.0f} kernel-only", flush=True)
print(f"scene upload (host tree build + copies, {len(sc)} objects): {(t1-t0)*1e3:.1f} ms")
r.close()
