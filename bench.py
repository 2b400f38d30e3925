"""Benchmark of the path-tracing hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched under torch.distributed.run)

A step = one pass of the hot path over the workload: render the workload's `spp` samples of every
pixel of its frame into an accumulation buffer resident in HBM.  With N > 1 GPUs every GPU renders the
whole workload, rank r the samples [r*spp, (r+1)*spp) of the same frame, and ONE RCCL reduce of 64-bit
integer words per step merges the ranks' buffers into a frame of N x spp samples per pixel (weak
scaling, the default: more GPUs buy more samples, and every GPU's launch stays the one BASELINE.json's
config names).  The reduce runs on a communication stream and overlaps the next step's tracing (two
buffers in turn).  The same line carries `strong_scaling`, measured in the same run: the step's fixed
budget cut into N slices (rank r renders samples [r*spp/N, (r+1)*spp/N) of the full frame; `--split
tile` cuts the 8x8 tiles instead), i.e. what N GPUs do for the time of ONE frame; `--scaling strong`
makes that the headline.  Rank 0 prints ONE JSON line.  The CPU legs (rank 0, N = 1) use the oracle as a
measuring stick only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import raytracer_amd as rt  # noqa: E402
from raytracer_amd import distributed as D  # noqa: E402

SCENE_SEED = 0x5EED0001
RENDER_SEED = 0x5EED0001
HEADLESS_CAMERA = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6)  # raytracer_headless/src/main.rs:20-28
PEAK_FP32_VALU_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def _wallpaper_camera():
    return ((0.0, 0.0, 0.0), rt.get_camera_rotation(-90.0, 0.0), 90.0, 10.0, 0.6)   # raytracer_gpu/src/main.rs:70


WORKLOADS = {
    # BASELINE.json configs[1]: the headless default scene at 1080p, 100 spp, depth 10
    "cfg2": dict(scene="default", width=1920, height=1080, spp=100, depth=10, sky=0, camera=lambda: HEADLESS_CAMERA, bound="valu_fp32",
                 desc="configs[1]: Scene::create_sample_scene (seed 0x5EED0001), headless camera, 1920x1080, 100 spp, depth 10"),
    # BASELINE.json configs[2]: many spheres and cubes, 1080p, 500 spp, depth 50
    "cfg3": dict(scene="sample", width=1920, height=1080, spp=500, depth=50, sky=0, camera=lambda: HEADLESS_CAMERA, bound="valu_fp32",
                 desc="configs[2]: scenes::create_sample_scene (many spheres and cubes, seed 0x5EED0001), headless camera, 1920x1080, 500 spp, depth 50"),
    # BASELINE.json configs[3]: wallpaper scene at 4K (2000 spp in the config: here 100 spp per step, throughput is spp-independent)
    "cfg4": dict(scene="wallpaper", width=3840, height=2160, spp=100, depth=10, sky=1, camera=_wallpaper_camera, bound="valu_fp32",
                 desc="configs[3]: scenes::create_wallpaper_scene, camera origin 0 / yaw -90 / fov 90, wallpaper sky, 3840x2160, 100 spp per step (of 2000), depth 10"),
    # BASELINE.json configs[4]: one million random spheres and cubes at 8K (tree and primitives in HBM / L2)
    "cfg5": dict(scene="synthetic:1000000", width=7680, height=4320, spp=16, depth=50, sky=0, bound="hbm",
                 camera=lambda: ((0.0, 60.0, -550.0), (0.0, -60.0, 550.0), 40.0, 10.0, 0.0),
                 desc="configs[4]: synthetic 1,000,000 spheres + cubes (seed 0x5EED0001), camera (0,60,-550) looking at the origin, fov 40, 7680x4320, 16 spp, depth 50"),
}


def host_threads():
    for var in ("OMP_NUM_THREADS",):
        v = os.environ.get(var)
        if v and v.isdigit() and int(v) > 0:
            return int(v)
    return os.cpu_count() or 1


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def cpu_side(scene, wl, run_baseline, name):
    """Oracle legs (rank 0, N = 1 only): algorithmic work per path from a TIGHT-mode count on a
    reduced frame of the same view; the reference algorithm (FAITHFUL) timed on a bounded sample of
    the workload; and the minimal standard algorithm (TIGHT: true slab test, ordered, culled) timed on
    the same sample, so that a speed-up splits into "better traversal" x "MI355X"."""
    from oracle import binding as ob  # measurement only: never on the product path
    log(f"{name}: oracle builds the reference tree and counts algorithmic work")
    osc = ob.OracleScene(scene.objects)
    camera = wl["camera"]()
    shrink = 8 if len(scene) < 100000 else 32   # the million-object scene: count work on a 240x135 frame
    cam_small = ob.camera_desc(*camera, wl["width"] // shrink, wl["height"] // shrink)
    _, _, cnt, _ = osc.render(cam_small, 4 if shrink == 8 else 1, wl["depth"], RENDER_SEED, ob.TIGHT, sky_mode=wl["sky"], threads=host_threads())
    trav_flop = (15 * cnt["box_tests"] + 30 * cnt["sphere_tests"] + 25 * cnt["cube_tests"]) / cnt["paths"]
    work = dict(flop_per_path=ob.algorithmic_flops(cnt) / cnt["paths"], bytes_per_path=ob.algorithmic_bytes(cnt) / cnt["paths"],
                traversal_flop_per_path=trav_flop, rays_per_path=cnt["rays"] / cnt["paths"], box_tests_per_ray=cnt["box_tests"] / cnt["rays"],
                prim_tests_per_ray=(cnt["sphere_tests"] + cnt["cube_tests"]) / cnt["rays"],
                hits_per_path=(cnt["rays"] - cnt["sky_misses"]) / cnt["paths"])
    baseline = tight = None
    if run_baseline:
        threads = host_threads()
        W, Hh = wl["width"], wl["height"]
        cam_full = ob.camera_desc(*camera, W, Hh)
        log(f"{name}: cpu baseline probes the reference algorithm on {threads} threads")
        if shrink == 8:
            # size the sample to ~12 s of CPU work from a quick probe: whole frames when they are
            # cheap enough (spp >= 1), else a centred band of rows at 1 spp
            _, _, c0, s0 = osc.render(cam_small, 2, wl["depth"], RENDER_SEED, ob.FAITHFUL, sky_mode=wl["sky"], threads=threads)
            budget_paths = 12.0 * c0["paths"] / s0
            frame = W * Hh
            if budget_paths >= frame:
                spp, win = int(min(wl["spp"], budget_paths // frame)), (0, 0, W, Hh)
            else:
                rows = int(max(8, budget_paths // W))
                spp, win = 1, (0, (Hh - rows) // 2, W, (Hh - rows) // 2 + rows)
        else:
            # The reference's walk tests nearly every primitive of a million-object scene per ray
            # (~20 ms per ray): time a centred crop of the full-resolution frame, one row per thread
            # at least, sized from a probe to ~15 s.
            x0, y0 = W // 2 - 8, Hh // 2 - threads // 2
            _, _, c0, s0 = osc.render(cam_full, 1, wl["depth"], RENDER_SEED, ob.FAITHFUL, sky_mode=wl["sky"], threads=threads,
                                      window=(x0, y0, x0 + 16, y0 + threads))
            per_path = s0 * threads / c0["paths"]                     # thread-seconds per path
            cols = int(max(16, min(128, 15.0 * threads / per_path / 72)))
            rows = int(max(threads, min(72, 15.0 * threads / per_path / cols)))
            x0, y0 = (W - cols) // 2, (Hh - rows) // 2
            spp, win = 1, (x0, y0, x0 + cols, y0 + rows)
        log(f"{name}: cpu baseline times window {win}, {spp} spp")
        _, _, c1, s1 = osc.render(cam_full, spp, wl["depth"], RENDER_SEED, ob.FAITHFUL, sky_mode=wl["sky"], threads=threads, window=win)
        sample = f"pixels x {win[0]}..{win[2]}, y {win[1]}..{win[3]} of the {W}x{Hh} frame, {spp} spp, depth {wl['depth']}"
        baseline = dict(value=c1["paths"] / s1 / 1e6, unit="Mpaths/s", cores=threads, kind="port",
                        sample=f"{sample}: {c1['paths']} paths in {s1:.1f} s; f64 restatement of the reference algorithm "
                               f"(recursive both-children BVH walk, per-axis AABB test), row-parallel on {threads} threads",
                        mrays_per_s=c1["rays"] / s1 / 1e6,
                        # what the reference's walk costs per ray (its per-axis box test rejects almost nothing): beside the
                        # minimal algorithm's count in `roofline`, this is the algorithmic part of any speed-up
                        box_tests_per_ray=c1["box_tests"] / max(1, c1["rays"]),
                        prim_tests_per_ray=(c1["sphere_tests"] + c1["cube_tests"]) / max(1, c1["rays"]))
        # the algorithm-matched figure on the same window; repeated until it has run for a second or so
        reps, secs, paths = 0, 0.0, 0
        while secs < 1.0 and reps < 64:
            _, _, c2, s2 = osc.render(cam_full, spp, wl["depth"], RENDER_SEED, ob.TIGHT, sky_mode=wl["sky"], threads=threads, window=win)
            reps, secs, paths = reps + 1, secs + s2, paths + c2["paths"]
        tight = dict(value=paths / secs / 1e6, unit="Mpaths/s", cores=threads, kind="port",
                     box_tests_per_ray=c2["box_tests"] / max(1, c2["rays"]), prim_tests_per_ray=(c2["sphere_tests"] + c2["cube_tests"]) / max(1, c2["rays"]),
                     sample=f"{sample}, {reps} pass(es): {paths} paths in {secs:.2f} s; the same paths by the minimal standard algorithm "
                            f"(true slab test, near child first, culled by the best hit) on the reference-shaped tree")
    return work, baseline, tight


def measured_traffic(name):
    """HBM bytes per path from rocprofv3 counters (FETCH_SIZE + WRITE_SIZE, corrected as the
    microarchitecture guide prescribes), measured per round by tools/profile.sh."""
    for rel in ("profiles/r03/traffic.json", "profiles/r02/traffic.json", "profiles/traffic.json"):
        path = os.path.join(ROOT, rel)
        if os.path.exists(path):
            t = json.load(open(path)).get(name)
            if t:
                return t["hbm_bytes_per_path"], f"{rel} ({t.get('launch', '1920x1080 x 32 spp')})"
    return None, None


def predicted_scaling(name):
    """Best-case strong-scaling efficiency per GPU count from ONE GPU (tools/scaling_ceiling.py: the launches
    an N-way split produces, timed back to back; the merge taken as hidden under the next step)."""
    path = os.path.join(ROOT, "profiles/r03/scaling_ceiling.json")
    if not os.path.exists(path):
        return None
    d = json.load(open(path)).get(name)
    if not d:
        return None
    return {"source": "profiles/r03/scaling_ceiling.json (one MI355X, tools/scaling_ceiling.py)", "ms_one_gpu": d["ms_one_gpu"],
            "strong_spp_split_efficiency": {n: v["eff_spp_split"] for n, v in d["by_gpus"].items()},
            "strong_tile_split_efficiency": {n: v["eff_tile_split"] for n, v in d["by_gpus"].items()},
            "weak_efficiency": "1.0 less the merge that does not hide under the next step's tracing (every GPU runs the one-GPU launch)"}


def measured_counters(name):
    """Occupancy and divergence of the render kernel on this workload from the rocprofv3 counter passes
    kept under profiles/ (tools/profile.sh, one 32-spp 1080p launch per pass): share of VALU lanes active
    per issued instruction, resident waves per CU, share of wave cycles parked in s_waitcnt."""
    for rel in (f"profiles/r03/{name}_pmc_summary.json", f"profiles/r02/{name}_pmc_summary.json"):
        path = os.path.join(ROOT, rel)
        if os.path.exists(path):
            d = json.load(open(path))["derived"]
            return {"valu_lane_utilisation": d.get("valu_lane_utilisation"), "resident_waves_per_cu": d.get("resident_waves_per_cu"),
                    "wave_cycles_waiting": d.get("wave_cycles_waiting_s_waitcnt"), "wave_cycles_issuing": d.get("wave_cycles_issuing"),
                    "salu_per_valu_instruction": d.get("salu_per_valu_instruction"), "source": rel}
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override the workload's samples per pixel per step")
    ap.add_argument("--scaling", default="weak", choices=["strong", "weak"],
                    help="N > 1: weak = every GPU renders the workload's whole budget, its own samples of the frame (default); "
                         "strong = the step's fixed budget is split over the GPUs")
    ap.add_argument("--split", default="spp", choices=["spp", "tile"],
                    help="N > 1, strong scaling: spp = every GPU renders its slice of the samples of the full frame; "
                         "tile = every GPU renders all samples of its own 8x8 tiles")
    ap.add_argument("--merge", default="abi", choices=["abi", "torch"],
                    help="N > 1: abi = rthip_merge_reduce (ncclReduce through the C ABI), falling back to torch.distributed.reduce")
    ap.add_argument("--rehearse-dist", action="store_true",
                    help="run the N > 1 code path (process group, communicator, merge pipeline, self-check) with the ranks there are, even one")
    ap.add_argument("--one-device", action="store_true",
                    help="rehearsal of the N > 1 path on a box with one GPU: every rank uses GPU 0 and the process group runs over gloo "
                         "(RCCL refuses two ranks on one GPU, so the merge falls back from the C ABI to torch.distributed); not a measurement")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline legs")
    ap.add_argument("--no-extra", action="store_true", help="skip the other configs, the f64 rate and the 1-spp frame")
    args = ap.parse_args()

    # Libraries under this process write banners to stdout (RCCL prints its version block at the first
    # communicator): everything but the one JSON line goes to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if args.gpus != 1 and world == 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    dev_index = 0 if args.one_device else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    # Everything of this process (buffer zero-fills, renders, timing events) goes onto one stream of its
    # own: torch's default stream has the handle 0, which the C ABI reads as "the context's own stream",
    # and that one is not ordered against torch's operations.
    torch.cuda.set_stream(torch.cuda.Stream(device))
    multi = world > 1 or args.rehearse_dist
    if args.one_device:
        import faulthandler
        faulthandler.dump_traceback_later(90, exit=False)   # a rehearsal that hangs says where
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        if args.one_device:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    weak = args.scaling == "weak" and multi
    tile_split = args.split == "tile" and multi and not weak

    def share(spp, mode=None):
        """(spp_begin, spp_count, tile part) of this rank for a step of `spp` samples; mode "weak",
        "spp" or "tile" (default: what the command line asked for)."""
        if not multi:
            return 0, spp, (0, 1)
        if mode is None:
            mode = "weak" if weak else "tile" if tile_split else "spp"
        if mode == "weak":
            return rank * spp, spp, (0, 1)
        if mode == "tile":
            return 0, spp, D.tile_part(rank, world)
        begin, count = D.spp_slice(spp, rank, world)
        return begin, count, (0, 1)

    merge_kind = {"kind": "none"}

    def make_pipeline(r, width, height):
        make = lambda: D.new_accum_tensor(width, height, device)  # noqa: E731
        if multi and args.merge == "abi" and D.init_abi_comm(r, rank, world, device):
            merge_kind["kind"] = "rthip_merge_reduce: ncclReduce(sum, uint64) through the C ABI, on a communication stream"
            return D.AbiMergePipeline(r, make, root=0)
        if multi:
            merge_kind["kind"] = ("all_reduce(SUM, int64) over gloo, staged through host memory (rehearsal)" if args.one_device else
                                  "torch.distributed.reduce(SUM, int64) over RCCL")
            if world > 1:
                r.set_option("grid_reserve", 48)   # room for the reduce kernel under the next step's render (the ABI merge reserves it by itself)
        return D.MergePipeline(make, dst=0)

    def run_workload(name, steps, warmup, spp_override=0, precision=rt.F32, mode=None):
        wl = dict(WORKLOADS[name])
        if spp_override:
            wl["spp"] = spp_override
        scene = rt.Scene.generate(wl["scene"], SCENE_SEED)
        cam = rt.Camera(*wl["camera"](), wl["width"], wl["height"])
        r = rt.Renderer(scene, cam, device=dev_index, max_depth=wl["depth"], seed=RENDER_SEED, sky_mode=wl["sky"], precision=precision)
        r.set_torch_stream()
        pipe = make_pipeline(r, wl["width"], wl["height"])
        begin, count, part = share(wl["spp"], mode)

        def step():
            accum = pipe.acquire()   # the other buffer may still be in its reduce
            accum.zero_()
            r.set_accum_buffer(accum.data_ptr(), accum.numel())
            r.accumulate(count, part=part, spp_begin=begin, sync=False)
            pipe.release()           # one RCCL reduce per step, overlapped with the next step's tracing

        for _ in range(warmup):
            step()
        pipe.drain()
        if multi and rank == 0:
            log(f"{name}: {mode or ('weak' if weak else 'split')} run, warm-up done")
        kernel_ms = []
        rays = paths = 0
        if multi:
            dist.barrier()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
            if rank == 0 and not multi:
                s = r.stats()  # waits for this step's kernel (HIP events on the launch stream)
                kernel_ms.append(s["kernel_ms"]); rays, paths = s["rays"], s["paths"]
        pipe.drain()
        torch.cuda.synchronize(device)
        if multi:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        stats = r.stats()
        if multi:
            coll = torch.device("cpu") if args.one_device else device   # gloo: host tensors
            t = torch.tensor([elapsed], dtype=torch.float64, device=coll)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            c = torch.tensor([stats["rays"], stats["paths"]], dtype=torch.int64, device=coll)   # of every rank's last step
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            rays, paths = int(c[0].item()), int(c[1].item())
        out = dict(wl=wl, scene=scene, elapsed=elapsed, kernel_ms=float(np.mean(kernel_ms)) if kernel_ms else stats["kernel_ms"],
                   rays=rays, paths=paths, stats=stats)
        if rank == 0 and not multi and precision == rt.F32 and not args.no_extra:
            # executed work: box and primitive tests the kernel really ran, on the tree it really walks
            r.set_option("count", 1)
            r.set_accum_buffer(0, 0)
            r.reset_accumulation()
            r.accumulate(min(4, wl["spp"]), spp_begin=0)
            s = r.stats()
            out["executed"] = dict(box_tests_per_ray=s["box_tests"] / max(1, s["rays"]),
                                   prim_tests_per_ray=(s["sphere_tests"] + s["cube_tests"]) / max(1, s["rays"]),
                                   traversal_flop_per_path=(15 * s["box_tests"] + 30 * s["sphere_tests"] + 25 * s["cube_tests"]) / max(1, s["paths"]),
                                   # a visit of a node reads it whole (two child boxes: 64 bytes, quantised 32), a primitive test 32 bytes, a hit its 16-byte material
                                   node_bytes=32 if (not s["scene_in_lds"] and s["n_nodes"] > 2048) else 64,
                                   node_visits_per_path=s["box_tests"] / 2 / max(1, s["paths"]),
                                   prim_tests_per_path=(s["sphere_tests"] + s["cube_tests"]) / max(1, s["paths"]))
            r.set_option("count", 0)
        # the 1-spp progressive frame (raytracer_runtime-style caller): ms per frame of a queue of frames on the
        # context's own stream, where the library lets the tail of one short launch run under the start of the
        # next (option overlap), and with one launch after the other
        if rank == 0 and name == "cfg2" and not args.no_extra and not multi and precision == rt.F32:
            r.set_accum_buffer(0, 0)
            r.set_stream(0)
            for key, overlap in (("frame_ms_1spp", 1), ("frame_ms_1spp_serial", 0)):
                r.set_option("overlap", overlap)
                r.reset_accumulation()
                for _ in range(6):
                    r.accumulate(1, sync=False)
                r.synchronize()
                t1 = time.perf_counter()
                for _ in range(60):
                    r.accumulate(1, sync=False)
                r.synchronize()
                out[key] = (time.perf_counter() - t1) / 60 * 1e3
            r.set_option("overlap", 1)
        r.close()
        return out

    def self_check():
        """N > 1: a small frame rendered in shares and merged must equal rank 0's own render of all
        of it, bit for bit."""
        wl = WORKLOADS[args.workload]
        w, h, spp = 256, 144, 8
        scene = rt.Scene.generate(wl["scene"] if len(wl["scene"]) < 12 else "default", SCENE_SEED)
        cam = rt.Camera(*wl["camera"](), w, h)
        r = rt.Renderer(scene, cam, device=dev_index, max_depth=wl["depth"], seed=RENDER_SEED, sky_mode=wl["sky"])
        r.set_torch_stream()
        pipe = make_pipeline(r, w, h)
        begin, count, part = (rank * spp, spp, (0, 1)) if weak else share(spp)
        buf = pipe.acquire()
        buf.zero_()
        r.set_accum_buffer(buf.data_ptr(), buf.numel())
        r.accumulate(count, part=part, spp_begin=begin, sync=False)
        pipe.release()
        pipe.drain()
        torch.cuda.synchronize(device)
        ok = True
        if rank == 0:
            merged = buf.clone()
            own = D.new_accum_tensor(w, h, device)
            r.set_accum_buffer(own.data_ptr(), own.numel())
            r.accumulate(spp * world if weak else spp, spp_begin=0)
            torch.cuda.synchronize(device)
            ok = bool(torch.equal(own, merged))
        r.close()
        return ok

    check = None
    if multi:
        check = self_check()
    res = run_workload(args.workload, args.steps, args.warmup, args.spp)
    # weak scaling is the headline; the fixed-budget split of the same workload is measured beside it
    strong = run_workload(args.workload, args.steps, 1, args.spp, mode="tile" if args.split == "tile" else "spp") if weak else None
    wl = res["wl"]
    paths_per_step_all = wl["width"] * wl["height"] * wl["spp"] * (world if weak else 1)
    value = paths_per_step_all * args.steps / res["elapsed"] / 1e6

    if rank == 0:
        split_text = ("whole budget per GPU (weak scaling: N GPUs render N x spp samples)" if weak else
                      "tile (each GPU renders all samples of its own 8x8 tiles)" if tile_split else
                      "spp (each GPU renders its slice of the step's samples of the full frame)")
        line = {
            "metric": "Mpaths/s (path-tracing throughput, 1080p)", "value": value, "unit": "Mpaths/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["elapsed"] / args.steps * 1e3,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["desc"], "spp_per_step": wl["spp"], "split": split_text, "merge": merge_kind["kind"],
                       "scene_objects": int(len(res["scene"])), "tree_depth": res["stats"]["tree_depth"], "scene_in_lds": res["stats"]["scene_in_lds"]},
            # ray segments of the last step (summed over the ranks) per second of step time
            "mrays_per_s": res["rays"] / (res["elapsed"] / args.steps) / 1e6,
            "rays_per_path": res["rays"] / max(1, res["paths"]),
        }
        if args.one_device:
            line["rehearsal"] = "--one-device: every rank on GPU 0, process group over gloo; the figures of this line are not measurements"
        if check is not None:
            line["merge_self_check"] = "bit-identical to one GPU (256x144, 8 spp)" if check else "MISMATCH"
        if strong is not None:
            ws = strong["wl"]
            line["strong_scaling"] = {
                "value": ws["width"] * ws["height"] * ws["spp"] * args.steps / strong["elapsed"] / 1e6, "unit": "Mpaths/s",
                "ms_per_step": strong["elapsed"] / args.steps * 1e3, "steps": args.steps,
                "split": "tile" if args.split == "tile" else "spp",
                "note": "the step's fixed budget of %d spp cut into %d shares, one reduce per step inside the timed region: the frame time N GPUs reach" % (ws["spp"], world)}
        if multi:
            pred = predicted_scaling(args.workload)
            if pred:
                line["predicted_scaling"] = pred
        if "frame_ms_1spp" in res:
            line["frame_ms_1080p_1spp"] = res["frame_ms_1spp"]
            line["frame_ms_1080p_1spp_one_after_the_other"] = res["frame_ms_1spp_serial"]

        def rooflines(name, rs, work):
            """The roofline objects of one workload from its run `rs` and the oracle's count `work`."""
            wlx = rs["wl"]
            secs = rs["kernel_ms"] * 1e-3
            achieved = work["flop_per_path"] * rs["paths"] / secs / 1e12
            per_path, src = measured_traffic(f"{name}_8k" if name == "cfg5" else name)   # cfg5: counters of the bench's own 8K launch
            if per_path is None:
                per_path, src = measured_traffic(name)
                if per_path is not None and name == "cfg5":
                    src += " -- a 1080p launch's bytes per path applied to the 8K launch: an estimate"
            traffic = per_path * rs["paths"] if per_path else None
            valu = {"bound": "valu_fp32", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_FP32_VALU_TFLOPS, "traffic": traffic, "traffic_source": src, "kernel": "render_kernel<float>",
                    "kernel_ms": rs["kernel_ms"], "algorithmic_flop_per_path": work["flop_per_path"],
                    "algorithmic_bytes_per_path": work["bytes_per_path"], "box_tests_per_ray": work["box_tests_per_ray"],
                    "prim_tests_per_ray": work["prim_tests_per_ray"]}
            ex = rs.get("executed")
            if ex:
                # the same yardstick with the traversal part replaced by what the kernel executed on its own tree
                flop_exec = work["flop_per_path"] - work["traversal_flop_per_path"] + ex["traversal_flop_per_path"]
                valu.update(frac_executed=flop_exec * rs["paths"] / secs / 1e12 / PEAK_FP32_VALU_TFLOPS, executed_flop_per_path=flop_exec,
                            executed_box_tests_per_ray=ex["box_tests_per_ray"], executed_prim_tests_per_ray=ex["prim_tests_per_ray"])
            st = rs["stats"]
            valu["launch"] = {"threads_per_workgroup": st["launch_threads"], "workgroups": st["launch_grid"], "lds_bytes_per_workgroup": st["launch_lds_bytes"],
                              "waves_per_cu_planned": st["waves_per_cu"]}
            counters = measured_counters(name)
            if counters:
                valu["counters"] = counters
            out = {"roofline": valu}
            hbm_alg = wlx["width"] * wlx["height"] * 3 * 8 * 2   # one read-modify-write of every accumulation word
            if wlx["bound"] == "hbm" and ex:
                # Scene in global memory: the bytes the walk asks of the memory system, counted on the tree
                # the kernel walks (device tallies), against the HBM roof; `traffic` is what missed L2.
                bytes_per_path = ex["node_visits_per_path"] * ex["node_bytes"] + ex["prim_tests_per_path"] * 32 + work["hits_per_path"] * 16
                gbs = bytes_per_path * rs["paths"] / secs / 1e9
                # What reached HBM (L2 misses, measured with rocprofv3 counters on this workload) is the figure
                # to hold against the HBM roof; the requested bytes are mostly served by L1/L2 and may exceed it.
                measured = (traffic / secs / 1e9) if traffic else None
                out["roofline_hbm"] = {"bound": "hbm", "achieved": measured, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                       "frac": (measured / PEAK_HBM_GBS) if measured else None,
                                       "traffic": traffic, "traffic_source": src,
                                       "requested_bytes_per_path": bytes_per_path, "requested_gbs": gbs,
                                       "kernel": "render_kernel<float> (scene in global memory, quantised nodes)",
                                       "note": "achieved = measured L2-miss bytes (FETCH_SIZE + WRITE_SIZE per path x paths) over kernel time; "
                                               "requested = node visits x node size + primitive tests x 32 + hits x 16 on the tree the kernel walks, "
                                               "(camera rays walk it as wave packets whose node reads are one scalar load per wave: they are counted per lane here), "
                                               "served mostly by L1/L2 (the kernel waits on dependent loads and issues VALU work, it is not bound by HBM)"}
            else:
                gbs = hbm_alg / secs / 1e9
                out["roofline_hbm"] = {"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "traffic": traffic,
                                       "note": "not the binding roof: scene and ray state stay on chip, the kernel is VALU-issue bound"}
            return out

        if not multi:
            work, baseline, tight = cpu_side(res["scene"], wl, not args.no_cpu, args.workload)
            line.update(rooflines(args.workload, res, work))
            if baseline:
                line["cpu_baseline"] = baseline
                line["cpu_baseline_tight"] = tight
                line["speedup_vs_cpu"] = value / baseline["value"]
                line["speedup_vs_cpu_tight"] = value / tight["value"]
        if not multi and not args.no_extra and args.workload == "cfg2":
            # the reference's own arithmetic on the GPU (bit-exact parity mode), same workload at a tenth of the samples
            r64 = run_workload("cfg2", 2, 1, 10, precision=rt.F64)
            line["f64_mode"] = {"value": r64["wl"]["width"] * r64["wl"]["height"] * 10 * 2 / r64["elapsed"] / 1e6, "unit": "Mpaths/s",
                                "kernel_ms": r64["kernel_ms"], "spp_per_step": 10, "dtype": "f64"}
            # the other single-GPU configurations, measured in the same run (bounded: few steps each)
            for name, steps in (("cfg3", 5), ("cfg4", 3), ("cfg5", 3)):
                log(f"extra workload {name}")
                rx = run_workload(name, steps, 1)
                wx = rx["wl"]
                vx = wx["width"] * wx["height"] * wx["spp"] * steps / rx["elapsed"] / 1e6
                extra = {"workload": wx["desc"], "value": vx, "unit": "Mpaths/s", "steps": steps, "ms_per_step": rx["elapsed"] / steps * 1e3,
                         "mrays_per_s": rx["rays"] / (rx["elapsed"] / steps) / 1e6, "rays_per_path": rx["rays"] / max(1, rx["paths"]),
                         "kernel_ms": rx["kernel_ms"], "scene_in_lds": rx["stats"]["scene_in_lds"]}
                workx, basex, tightx = cpu_side(rx["scene"], wx, not args.no_cpu, name)
                extra.update(rooflines(name, rx, workx))
                if basex:
                    extra.update(cpu_baseline=basex, cpu_baseline_tight=tightx, speedup_vs_cpu=vx / basex["value"],
                                 speedup_vs_cpu_tight=vx / tightx["value"])
                line[name] = extra
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + "\n").encode())
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
