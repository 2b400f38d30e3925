"""Where the time goes on the million-object scene: resolution / spp sweeps, repeats, test counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import raytracer_amd as rt
n = int(os.environ.get("CFG5_OBJECTS", "1000000"))
sc = rt.create_synthetic_scene(n, seed=0x5EED0001)
for (w, h) in ((1920, 1080), (7680, 4320)):
    cam = rt.Camera((0.0, 60.0, -550.0), (0.0, -60.0, 550.0), 40.0, 10.0, 0.0, w, h)
    r = rt.Renderer(sc, cam, max_depth=int(os.environ.get("DEPTH", "50")), seed=1)
    s = r.stats(); print("nodes", s["n_nodes"], "leaves", s["n_leaves"], "prims", s["n_prims"], "depth", s["tree_depth"])
    for spp in (1, 1, 4, 4, 16):
        r.reset_accumulation(); t0 = time.perf_counter(); r.accumulate(spp, spp_begin=0); dt = time.perf_counter() - t0; s = r.stats()
        print(f"{w}x{h} spp {spp}: {s['paths']/s['kernel_ms']/1e3:.0f} Mpaths/s {s['rays']/s['kernel_ms']/1e3:.0f} Mrays/s kernel {s['kernel_ms']:.1f} ms wall {dt*1e3:.1f} "
              f"rays/path {s['rays']/s['paths']:.2f} box/ray {s['box_tests']/max(1,s['rays']):.1f} sph/ray {s['sphere_tests']/max(1,s['rays']):.1f} cube/ray {s['cube_tests']/max(1,s['rays']):.1f}", flush=True)
    r.close()
