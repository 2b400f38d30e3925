"""Ad-hoc first-light check on a GPU box: HIP path vs the CPU oracle."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import raytracer_amd as rt
from oracle import binding as ob

def cams(w, h):
    return {"default": ((13, 1.5, 3), (-13, -1.5, -3), 20, 10, 0.6, w, h),
            "sample": ((13, 1.5, 3), (-13, -1.5, -3), 20, 10, 0.6, w, h),
            "glass": ((1, 1.5, 3), rt.get_camera_rotation(-90, -20), 90, 10, 0.6, w, h),
            "wallpaper": ((0, 0, 0), rt.get_camera_rotation(-90, 0), 90, 10, 0.0, w, h)}

W, H = 160, 90
for name, ca in cams(W, H).items():
    sc = rt.Scene.generate(name)
    osc = ob.OracleScene(sc.objects)
    ocam = ob.camera_desc(*ca)
    r = rt.Renderer(sc, rt.Camera(*ca), seed=7)
    for jit in (False, True):
        ref = osc.trace_primary(ocam, ob.TIGHT, jitter=jit, seed=7, sample=3)
        g64 = r.trace_primary(rt.F64, jitter=jit, sample=3)
        g32 = r.trace_primary(rt.F32, jitter=jit, sample=3)
        hit = ref[0] >= 0
        print(name, "jit", jit, "f64 exact:", [bool((a == b).all()) for a, b in zip(ref, g64)],
              "| f32 id mismatch", int((ref[0] != g32[0]).sum()), "of", ref[0].size,
              "max rel t", float((np.abs(g32[1] - ref[1])[hit & (g32[0] == ref[0])] / ref[1][hit & (g32[0] == ref[0])]).max()),
              "max |dn|", float(np.abs(g32[2] - ref[2])[hit & (g32[0] == ref[0])].max()))
    spp, depth = 4, 10
    acc, smp, cnt, secs = osc.render(ocam, spp, depth, 7, ob.TIGHT, samples=True)
    g = r.render_samples(spp, precision=rt.F64)
    print(name, "f64 samples bit-exact:", bool((g == smp).all()), "max abs", float(np.abs(g - smp).max()), "stats", r.stats())
    g = r.render_samples(spp, precision=rt.F32)
    d = np.abs(g - smp)
    print(name, "f32 samples: frac within 1e-3 rel", float((d <= 1e-3 * np.maximum(np.abs(smp), 1e-3)).mean()), "mean abs", float(d.mean()))
    r.precision = rt.F64
    r.reset_accumulation(); r.accumulate(spp)
    a64 = r.read_accum()
    print(name, "f64 accum max abs diff", float(np.abs(a64 - acc).max()))
    r.precision = rt.F32
    r.reset_accumulation(); r.accumulate(spp)
    a32 = r.read_accum()
    print(name, "f32 accum mean abs diff", float(np.abs(a32 - acc).mean()), "rmse(mean img)", float(np.sqrt((((a32 - acc) / spp) ** 2).mean())))
    # partition invariance
    r.reset_accumulation()
    for part in range(3):
        r.accumulate(spp, part=(part, 3), spp_begin=0)
    a32p = r.read_accum()
    print(name, "tile partition bit-identical:", bool((a32p == a32).all()))
    r.reset_accumulation(); r.accumulate(2, spp_begin=0); r.accumulate(2, spp_begin=2)
    print(name, "spp split bit-identical:", bool((r.read_accum() == a32).all()))
    r.close()

# throughput first light
sc = rt.Scene.generate("default")
ca = cams(1920, 1080)["default"]
r = rt.Renderer(sc, rt.Camera(*ca), seed=7)
for spp in (4, 16, 16):
    r.reset_accumulation()
    t0 = time.time(); r.accumulate(spp); t1 = time.time()
    s = r.stats()
    print("1080p spp", spp, "wall", round(t1 - t0, 4), "kernel ms", round(s["kernel_ms"], 3), "Mpaths/s", round(s["paths"] / s["kernel_ms"] / 1e3, 1),
          "Mrays/s", round(s["rays"] / s["kernel_ms"] / 1e3, 1), "lds", s["scene_in_lds"])
