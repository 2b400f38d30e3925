"""Does a kernel of another stream run WHILE the persistent render kernel runs, or after it?
The merge of frame k (an RCCL reduce on the communication stream) is meant to hide behind the
tracing of frame k + 1; a render launch that fills every workgroup slot of the chip would make it
wait for the render's tail instead.  One GPU: a 50 MB int64 add on a side stream stands in for
the reduce kernel.
  python tools/overlap_probe.py [grid_reserve ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import raytracer_amd as rt
from raytracer_amd import distributed as D

CAM = ((13.0, 1.5, 3.0), (-13.0, -1.5, -3.0), 20.0, 10.0, 0.6)
W, H, SPP = 1920, 1080, 12
dev = torch.device("cuda", 0)
main, side = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
torch.cuda.set_stream(main)
x = D.new_accum_tensor(W, H, dev); y = D.new_accum_tensor(W, H, dev)
for reserve in [int(a) for a in sys.argv[1:]] or [0]:
    r = rt.Renderer(rt.Scene.generate("default"), rt.Camera(*CAM, W, H), max_depth=10, seed=1, options={"grid_reserve": reserve})
    r.set_torch_stream(main)
    buf = D.new_accum_tensor(W, H, dev)
    r.set_accum_buffer(buf.data_ptr(), buf.numel())
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    def render_only():
        ev[0].record(main); r.accumulate(SPP, spp_begin=0, sync=False); ev[1].record(main)
    def add_only():
        with torch.cuda.stream(side):
            ev[2].record(side); x.add_(y); ev[3].record(side)
    for _ in range(3): render_only(); add_only()
    torch.cuda.synchronize()
    best = {}
    for _ in range(5):
        render_only(); torch.cuda.synchronize(); t_r = ev[0].elapsed_time(ev[1])
        add_only(); torch.cuda.synchronize(); t_a = ev[2].elapsed_time(ev[3])
        # both: the add is queued 0.2 ms after the render has started
        ev[0].record(main); r.accumulate(SPP, spp_begin=0, sync=False); ev[1].record(main)
        time.sleep(0.0002)
        add_only(); torch.cuda.synchronize()
        both_r, add_end = ev[0].elapsed_time(ev[1]), ev[0].elapsed_time(ev[3])
        rec = dict(render_ms=t_r, add_ms=t_a, render_with_add_ms=both_r, add_done_after_render_start_ms=add_end)
        if not best or rec["render_with_add_ms"] < best["render_with_add_ms"]: best = rec
    print(f"grid_reserve {reserve:4d}: " + "  ".join(f"{k} {v:.3f}" for k, v in best.items()), flush=True)
    r.close()
