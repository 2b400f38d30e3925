"""Benchmark of the path-tracing hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched under torch.distributed.run)

A step = one pass of the hot path: render `spp` samples of every pixel of the workload's frame
into an accumulation buffer resident in HBM (and, for N > 1, merge the ranks' buffers with one
RCCL reduce per frame, double-buffered so that it overlaps the next frame's tracing).  Scaling is weak: every GPU renders its own `spp` samples of the full frame, so
N GPUs deliver an image of N x spp samples in the same time.  Rank 0 prints ONE JSON line.
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


def cpu_side(scene, wl, run_baseline):
    """Oracle legs (rank 0, N = 1 only): algorithmic work per path from a TIGHT-mode count on a
    reduced frame of the same view, and the reference algorithm (FAITHFUL) timed on a bounded
    sample of the workload."""
    from oracle import binding as ob  # measurement only: never on the product path
    print("[bench] oracle: building the reference tree and counting algorithmic work", file=sys.stderr, flush=True)
    osc = ob.OracleScene(scene.objects)
    camera = wl["camera"]()
    shrink = 8 if len(scene) < 100000 else 32   # the million-object scene: count work on a 240x135 frame
    cam_small = ob.camera_desc(*camera, wl["width"] // shrink, wl["height"] // shrink)
    _, _, cnt, _ = osc.render(cam_small, 4 if shrink == 8 else 1, wl["depth"], RENDER_SEED, ob.TIGHT, sky_mode=wl["sky"], threads=host_threads())
    work = dict(flop_per_path=ob.algorithmic_flops(cnt) / cnt["paths"], bytes_per_path=ob.algorithmic_bytes(cnt) / cnt["paths"],
                rays_per_path=cnt["rays"] / cnt["paths"], box_tests_per_ray=cnt["box_tests"] / cnt["rays"],
                prim_tests_per_ray=(cnt["sphere_tests"] + cnt["cube_tests"]) / cnt["rays"])
    baseline = None
    if run_baseline:
        threads = host_threads()
        W, Hh = wl["width"], wl["height"]
        cam_full = ob.camera_desc(*camera, W, Hh)
        print(f"[bench] cpu baseline: probing the reference algorithm on {threads} threads", file=sys.stderr, flush=True)
        if shrink == 8:
            # size the sample to ~15 s of CPU work from a quick probe: whole frames when they are
            # cheap enough (spp >= 1), else a centred band of rows at 1 spp
            _, _, c0, s0 = osc.render(cam_small, 2, wl["depth"], RENDER_SEED, ob.FAITHFUL, sky_mode=wl["sky"], threads=threads)
            budget_paths = 15.0 * c0["paths"] / s0
            frame = W * Hh
            if budget_paths >= frame:
                spp, win = int(min(wl["spp"], budget_paths // frame)), (0, 0, W, Hh)
            else:
                rows = int(max(8, budget_paths // W))
                spp, win = 1, (0, (Hh - rows) // 2, W, (Hh - rows) // 2 + rows)
        else:
            # The reference's walk tests nearly every primitive of a million-object scene per ray
            # (~20 ms per ray): time a centred crop of the full-resolution frame, one row per thread
            # at least, sized from a probe to ~20 s.
            x0, y0 = W // 2 - 8, Hh // 2 - threads // 2
            _, _, c0, s0 = osc.render(cam_full, 1, wl["depth"], RENDER_SEED, ob.FAITHFUL, sky_mode=wl["sky"], threads=threads,
                                      window=(x0, y0, x0 + 16, y0 + threads))
            per_path = s0 * threads / c0["paths"]                     # thread-seconds per path
            cols = int(max(16, min(128, 20.0 * threads / per_path / 72)))
            rows = int(max(threads, min(72, 20.0 * threads / per_path / cols)))
            x0, y0 = (W - cols) // 2, (Hh - rows) // 2
            spp, win = 1, (x0, y0, x0 + cols, y0 + rows)
        print(f"[bench] cpu baseline: timing window {win}, {spp} spp", file=sys.stderr, flush=True)
        _, _, c1, s1 = osc.render(cam_full, spp, wl["depth"], RENDER_SEED, ob.FAITHFUL, sky_mode=wl["sky"], threads=threads, window=win)
        baseline = dict(value=c1["paths"] / s1 / 1e6, unit="Mpaths/s", cores=threads, kind="port",
                        sample=f"pixels x {win[0]}..{win[2]}, y {win[1]}..{win[3]} of the {W}x{Hh} frame, {spp} spp, depth {wl['depth']}: "
                               f"{c1['paths']} paths in {s1:.1f} s; f64 restatement of the reference algorithm "
                               f"(recursive both-children BVH walk, per-axis AABB test), row-parallel on {threads} threads",
                        mrays_per_s=c1["rays"] / s1 / 1e6)
    return work, baseline


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override samples per pixel per GPU per step")
    ap.add_argument("--split", default="spp", choices=["spp", "tile"],
                    help="N > 1: spp = every GPU renders its own samples of the full frame (weak scaling, the default); "
                         "tile = every GPU renders all samples of its own 8x8 tiles (strong scaling)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the cfg3 / 1-spp frame extras")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if args.gpus != 1 and world == 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    def run_workload(name, steps, warmup, spp_override=0):
        wl = dict(WORKLOADS[name])
        if spp_override:
            wl["spp"] = spp_override
        scene = rt.Scene.generate(wl["scene"], SCENE_SEED)
        cam = rt.Camera(*wl["camera"](), wl["width"], wl["height"])
        r = rt.Renderer(scene, cam, device=local_rank, max_depth=wl["depth"], seed=RENDER_SEED, sky_mode=wl["sky"])
        stream = torch.cuda.current_stream(device)
        r.set_stream(stream.cuda_stream)
        pipe = D.MergePipeline(lambda: D.new_accum_tensor(wl["width"], wl["height"], device), dst=0)
        tile_split = args.split == "tile" and world > 1
        begin = 0 if tile_split else rank * wl["spp"]  # spp split: every rank renders its own samples of the full frame
        part = D.tile_part(rank, world) if tile_split else (0, 1)

        def step():
            accum = pipe.acquire()   # the other buffer may still be in its reduce
            accum.zero_()
            r.set_accum_buffer(accum.data_ptr(), accum.numel())
            r.accumulate(wl["spp"], part=part, spp_begin=begin, sync=False)
            pipe.release()           # one RCCL reduce per frame, overlapped with the next frame's tracing

        for _ in range(warmup):
            step()
        pipe.drain()
        kernel_ms = []
        rays = paths = 0
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
            if rank == 0:
                s = r.stats()  # waits for this step's kernel (HIP events on the launch stream)
                kernel_ms.append(s["kernel_ms"]); rays, paths = s["rays"], s["paths"]
        pipe.drain()
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        out = dict(wl=wl, scene=scene, elapsed=elapsed, kernel_ms=float(np.mean(kernel_ms)) if kernel_ms else 0.0,
                   rays=rays, paths=paths, stats=r.stats() if rank == 0 else None)
        # the 1-spp progressive frame (raytracer_runtime-style caller): ms per frame
        if rank == 0 and name == "cfg2" and not args.no_extra and world == 1:
            r.reset_accumulation()
            for _ in range(3):
                r.accumulate(1)
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            for _ in range(20):
                r.accumulate(1, sync=False)
            r.synchronize()
            out["frame_ms_1spp"] = (time.perf_counter() - t1) / 20 * 1e3
        r.close()
        return out

    res = run_workload(args.workload, args.steps, args.warmup, args.spp)
    wl = res["wl"]
    tile_split = args.split == "tile" and world > 1
    paths_per_step_all = wl["width"] * wl["height"] * wl["spp"] * (1 if tile_split else world)
    value = paths_per_step_all * args.steps / res["elapsed"] / 1e6

    if rank == 0:
        line = {
            "metric": "Mpaths/s (path-tracing throughput, 1080p)", "value": value, "unit": "Mpaths/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["elapsed"] / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if tile_split else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl["desc"], "spp_per_gpu_per_step": wl["spp"], "split": ("tile (each GPU renders all samples of its own 8x8 tiles; one RCCL reduce merges)" if tile_split else
                                 "spp (each GPU renders its own samples of the full frame; one RCCL reduce merges)"),
                       "scene_objects": int(len(res["scene"])), "tree_depth": res["stats"]["tree_depth"], "scene_in_lds": res["stats"]["scene_in_lds"]},
            "mrays_per_s": res["rays"] * world * args.steps / res["elapsed"] / 1e6 if res["kernel_ms"] else None,  # rank 0's rays x ranks
            "rays_per_path": res["rays"] / max(1, res["paths"]),
        }
        if "frame_ms_1spp" in res:
            line["frame_ms_1080p_1spp"] = res["frame_ms_1spp"]
        work, baseline = (None, None)
        if world == 1:
            work, baseline = cpu_side(res["scene"], wl, not args.no_cpu)
        if work:
            flops_per_launch = work["flop_per_path"] * res["paths"]
            achieved = flops_per_launch / (res["kernel_ms"] * 1e-3) / 1e12
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):  # measured with rocprofv3 counters (bytes per path), scaled to this launch
                t = json.load(open(tpath)).get(args.workload)
                traffic = t["hbm_bytes_per_path"] * res["paths"] if t else None
            valu = {"bound": "valu_fp32", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                    "frac": achieved / PEAK_FP32_VALU_TFLOPS, "traffic": traffic, "kernel": "render_kernel<float>",
                    "kernel_ms": res["kernel_ms"], "algorithmic_flop_per_path": work["flop_per_path"],
                    "algorithmic_bytes_per_path": work["bytes_per_path"], "box_tests_per_ray": work["box_tests_per_ray"],
                    "prim_tests_per_ray": work["prim_tests_per_ray"]}
            line["roofline"] = valu
            if wl["bound"] == "hbm":
                # Scene in global memory: every node and primitive record a ray reads comes through L1/L2.
                # The algorithmic bytes (oracle TIGHT counts on the reference-shaped tree) over time
                # exceed the HBM roof whenever caches serve most of them; `traffic` is what actually
                # missed L2 (rocprofv3 FETCH_SIZE + WRITE_SIZE), the figure to hold against 8 TB/s.
                gbs = work["bytes_per_path"] * res["paths"] / (res["kernel_ms"] * 1e-3) / 1e9
                line["roofline_hbm"] = {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0, "traffic": traffic,
                                        "measured_l2_miss_gbs": (traffic / (res["kernel_ms"] * 1e-3) / 1e9) if traffic else None,
                                        "kernel": "render_kernel<float> (scene in global memory)",
                                        "note": "latency bound: 70 % of wave cycles wait on memory (profiles/r01/cfg5_pmc_summary.json)"}
            # the same launch against the memory roof: what the algorithm must move through HBM is one
            # read-modify-write of the accumulation words of every pixel (scene and ray state stay on chip)
            hbm_alg = wl["width"] * wl["height"] * 3 * 8 * 2
            gbs = hbm_alg / (res["kernel_ms"] * 1e-3) / 1e9
            if wl["bound"] != "hbm":
                line["roofline_hbm"] = {"bound": "hbm", "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0, "traffic": traffic,
                                        "note": "not the binding roof: the kernel is VALU-issue bound"}
        if baseline:
            line["cpu_baseline"] = baseline
            line["speedup_vs_cpu"] = value / baseline["value"]
        if world == 1 and not args.no_extra and args.workload == "cfg2":
            # the many-spheres-and-cubes target workload, measured in the same run
            r3 = run_workload("cfg3", 2, 1)
            w3 = r3["wl"]
            v3 = w3["width"] * w3["height"] * w3["spp"] * 2 / r3["elapsed"] / 1e6
            extra = {"workload": w3["desc"], "value": v3, "unit": "Mpaths/s", "ms_per_step": r3["elapsed"] / 2 * 1e3,
                     "mrays_per_s": r3["rays"] * 2 / r3["elapsed"] / 1e6, "kernel_ms": r3["kernel_ms"]}
            if not args.no_cpu:
                work3, base3 = cpu_side(r3["scene"], w3, True)
                ach3 = work3["flop_per_path"] * r3["paths"] / (r3["kernel_ms"] * 1e-3) / 1e12
                extra.update(cpu_baseline=base3, speedup_vs_cpu=v3 / base3["value"],
                             roofline={"bound": "valu_fp32", "achieved": ach3, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                                       "frac": ach3 / PEAK_FP32_VALU_TFLOPS, "algorithmic_flop_per_path": work3["flop_per_path"]})
            line["cfg3"] = extra
        print(json.dumps(line))
    elif world == 1:
        pass
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
