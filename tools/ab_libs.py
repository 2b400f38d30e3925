"""A/B of several builds of librthip (same ABI) on both bench workloads: one subprocess per
build (a process can load only one), two rounds each, interleaved; the checksums of a 4-spp image of
either scene show whether the builds render the same words."""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:]
code = r'''
import os, sys, json
sys.path.insert(0, %r)
import raytracer_amd as rt
CAM = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6)
out = {}
for which, scene, depth in (("cfg2", "default", 10), ("cfg3", "sample", 50)):
    r = rt.Renderer(rt.Scene.generate(scene), rt.Camera(*CAM, 1920, 1080), max_depth=depth, seed=1)
    best = 0
    for _ in range(4):
        r.reset_accumulation(); r.accumulate(32, spp_begin=0)
        s = r.stats(); best = max(best, s["paths"] / s["kernel_ms"] / 1e3)
    out[which] = best
    import zlib
    r.reset_accumulation(); r.accumulate(4, spp_begin=0)
    out[which + "_crc"] = zlib.crc32(r.read_accum().tobytes())   # equal images across builds show as equal checksums
    r.close()
for name, w, h in (("frame1080_us", 1920, 1080), ("frame270_us", 480, 270)):   # the 1-spp frame, kernel time in microseconds
    r = rt.Renderer(rt.Scene.generate("default"), rt.Camera(*CAM, w, h), max_depth=10, seed=1)
    for _ in range(3): r.accumulate(1)
    ks = []
    for _ in range(12):
        r.accumulate(1); ks.append(r.stats()["kernel_ms"])
    out[name] = min(ks) * 1e3
    r.close()
print(json.dumps(out))
''' % ROOT
res = {l: [] for l in libs}
for rnd in range(2):
    for l in libs:
        env = dict(os.environ, RTHIP_LIBRARY=os.path.join(ROOT, "raytracer_amd", l))
        o = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=200)
        if o.returncode != 0:
            print(l, "FAILED", o.stderr[-400:]); continue
        res[l].append(json.loads(o.stdout.strip().splitlines()[-1]))
for l in libs:
    print(l, " ".join(f"{k}: " + "/".join(f"{r[k]:.0f}" for r in res[l]) for k in ("cfg2", "cfg3", "frame1080_us", "frame270_us")),
          "crc", " ".join("%08x" % r[k] for r in res[l][:1] for k in ("cfg2_crc", "cfg3_crc")))
