"""A scene whose objects move every frame, rendered progressively the way raytracer_runtime's loop
renders a static one (raytracer_runtime/src/main.rs:69-94): per frame the objects drift, the tree is
refitted (rthip_scene_refit) or rebuilt on the GPU (option tree_gpu), and `spp` samples per pixel
are traced at 1080p.  Prints the time of each part and the frame rate.
  python tools/animate.py [objects [frames [spp]]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as rt

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 1
sc = rt.Scene.generate(f"synthetic:{n}", 0x5EED0001)
cam = rt.Camera((0.0, 60.0, -550.0), (0.0, -60.0, 550.0), 40.0, 10.0, 0.0, 1920, 1080)
g = np.random.default_rng(1)
small = np.abs(sc.objects["extent"]).max(axis=1) < 10.0
velocity = g.normal(0.0, 0.02, (int(small.sum()), 3)) * [1.0, 0.0, 1.0]
def on_the_device():
    """The same animation with the objects in device memory, moved there (torch), refitted from there."""
    import torch
    from raytracer_amd import _native as N
    dev = torch.device("cuda", 0)
    r = rt.Renderer(sc, cam, max_depth=50, seed=1)
    r.set_torch_stream()
    raw = np.frombuffer(np.ascontiguousarray(sc.objects, dtype=N.OBJECT_DTYPE).tobytes(), dtype=np.uint8).copy()
    objs = torch.from_numpy(raw).to(dev).view(torch.float64).view(len(sc), 11)   # kind | material, centre, extent, albedo, parameter
    rows = torch.from_numpy(np.flatnonzero(small)).to(dev)
    v = torch.from_numpy(velocity).to(dev)
    torch.cuda.synchronize()
    t_tree = t_trace = 0.0
    t_start = time.perf_counter()
    for f in range(frames):
        t1 = time.perf_counter()
        objs[rows, 1:4] += v
        r.refit_device_objects(objs.data_ptr(), len(sc))
        r.synchronize()
        t2 = time.perf_counter()
        r.accumulate(spp)
        t3 = time.perf_counter()
        t_tree += t2 - t1; t_trace += t3 - t2
    total = time.perf_counter() - t_start
    print(f"{n} objects, {spp} spp at 1080p, objects moved and refitted in device memory: {frames / total:6.1f} frames/s "
          f"(per frame: move and refit {t_tree / frames * 1e3:.1f} ms, trace {t_trace / frames * 1e3:.1f} ms)", flush=True)
    r.close()


for name, opts, rebuild_every in (("refit every frame (surface-area tree from the host, rebuilt never)", {}, 0),
                                  ("refit every frame, device-built tree rebuilt every 16th", {"tree_gpu": 1}, 16),
                                  ("device-built tree rebuilt every frame", {"tree_gpu": 1}, 1)):
    r = rt.Renderer(sc, cam, max_depth=50, seed=1, options=opts)
    cur = rt.Scene(sc.objects.copy())
    t_move = t_tree = t_trace = 0.0
    t_start = time.perf_counter()
    for f in range(frames):
        t0 = time.perf_counter()
        cur.objects["center"][small] += velocity
        t1 = time.perf_counter()
        if rebuild_every and f % rebuild_every == 0:
            r.set_scene(cur)
        else:
            r.refit_scene(cur)
        r.synchronize()
        t2 = time.perf_counter()
        r.accumulate(spp)
        t3 = time.perf_counter()
        t_move += t1 - t0; t_tree += t2 - t1; t_trace += t3 - t2
    total = time.perf_counter() - t_start
    print(f"{n} objects, {spp} spp at 1080p, {name}: {frames / total:6.1f} frames/s "
          f"(per frame: host moves objects {t_move / frames * 1e3:.1f} ms, tree {t_tree / frames * 1e3:.1f} ms, trace {t_trace / frames * 1e3:.1f} ms)", flush=True)
    r.close()
on_the_device()
