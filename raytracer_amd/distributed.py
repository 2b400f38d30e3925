"""Multi-GPU partitioning and merge for the path tracer: one process per GPU.

Every (pixel, sample) is independent (the random stream is keyed by seed, pixel and sample)
and the scene is replicated, so a render shards two ways with no exchange during tracing:

* spp split: rank r renders samples [r * spp / N, (r + 1) * spp / N) of every pixel;
* tile split: rank r renders the 8x8 tiles t with t % N == r.

Either way each rank ends with a full-frame accumulation buffer of 64-bit integers (zeros
where it owns nothing) and the image is their element-wise sum: ONE reduce over RCCL/xGMI.
Integer sums commute, so the merged image is bit-identical to a single-GPU render.
There is no counterpart in the reference, which is a single-process renderer.
"""
import numpy as np
import torch
import torch.distributed as dist


def spp_slice(spp_total, rank, world):
    """Contiguous sample range of `rank`: (begin, count)."""
    lo = spp_total * rank // world
    hi = spp_total * (rank + 1) // world
    return lo, hi - lo


def tile_part(rank, world):
    """(part_index, part_count) for rthip_render_params."""
    return rank, world


def accum_words(width, height):
    return ((width + 7) // 8) * ((height + 7) // 8) * 64 * 3


def new_accum_tensor(width, height, device):
    """A zeroed accumulation buffer as an int64 tensor (uint64 words of 2^-36; sums stay far
    below 2^63, so int64 addition is exact)."""
    return torch.zeros(accum_words(width, height), dtype=torch.int64, device=device)


def _reduce(accum, dst, group, async_op=False):
    """dist.reduce; under gloo a device buffer is summed through host memory with a blocking all_reduce
    (rehearsals of the multi-rank path on one GPU: gloo's own handling of device tensors did not
    finish there for frame-sized buffers); rank `dst` ends with the same sums either way."""
    if accum.is_cuda and dist.get_backend(group) == "gloo":
        host = accum.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        accum.copy_(host)
        return None
    return dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def merge_accum(accum, dst=0, group=None):
    """Sum every rank's accumulation buffer into rank `dst` (a single reduce)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        _reduce(accum, dst, group)
    return accum


class MergePipeline:
    """Double-buffered merge for a stream of frames: while frame k's accumulation buffer is being
    reduced (RCCL runs on its own stream), frame k + 1 renders into the other buffer, so the
    collective hides behind tracing instead of adding to every frame.

        buf = pipe.acquire()      # the buffer of this frame; waits for the reduce that last used it
        ... zero it, render into it ...
        pipe.release()            # starts the (asynchronous) reduce of this frame
        pipe.drain()              # after the last frame

    On rank `dst`, the buffer handed out two frames ago holds that frame's merged sums (read it
    after acquire() returns it again, or after drain())."""

    def __init__(self, make_buffer, dst=0, group=None):
        self.bufs = [make_buffer(), make_buffer()]
        self.pending = [None, None]
        self.dst, self.group, self.k = dst, group, 0
        self.active = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1

    def acquire(self):
        i = self.k % 2
        if self.pending[i] is not None:
            self.pending[i].wait()   # NCCL: makes the current stream wait, the host does not block
            self.pending[i] = None
        return self.bufs[i]

    def release(self):
        i = self.k % 2
        if self.active:
            self.pending[i] = _reduce(self.bufs[i], self.dst, self.group, async_op=True)
        self.k += 1
        return self.bufs[i]

    def drain(self):
        for i in (0, 1):
            if self.pending[i] is not None:
                self.pending[i].wait()
                self.pending[i] = None


def init_abi_comm(renderer, rank, world, device):
    """Makes `renderer`'s context rank `rank` of an RCCL communicator of `world` GPUs through the C
    ABI (rthip_comm_unique_id / rthip_comm_init).  The 128-byte id travels from rank 0 over the
    torch.distributed group that is already up.  Returns True when every rank succeeded; on False
    the caller merges through torch.distributed instead."""
    ok = 1
    raw = bytes(128)
    if dist.get_backend() == "gloo":
        device = torch.device("cpu")   # the few bytes exchanged here need no device buffer
    if rank == 0:
        try:
            raw = renderer.comm_unique_id()
        except Exception as e:  # no librccl: an all-zero id tells the others
            print(f"[distributed] rank 0: C-ABI communicator not available ({e})", flush=True)
            raw = bytes(128)
    # every rank takes part in every collective below, whatever failed before it
    ident = torch.tensor(list(raw), dtype=torch.uint8, device=device)
    dist.broadcast(ident, src=0)
    raw = bytes(ident.cpu().tolist())
    if any(raw):
        try:
            renderer.comm_init(raw, rank, world)
        except Exception as e:  # RCCL refused: every rank learns of it below
            print(f"[distributed] rank {rank}: C-ABI communicator not available ({e})", flush=True)
            ok = 0
    else:
        ok = 0
    flag = torch.tensor([ok], dtype=torch.int32, device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item())


class AbiMergePipeline:
    """MergePipeline with the reduce behind the C ABI (rthip_merge_reduce: one ncclReduce of uint64
    words on the context's communication stream).  The renderer's stream must be the stream the
    caller zeroes the buffers on (Renderer.set_stream), so that acquire() orders both behind the
    reduce that last used the buffer."""

    def __init__(self, renderer, make_buffer, root=0):
        self.r, self.root, self.k = renderer, root, 0
        self.bufs = [make_buffer(), make_buffer()]

    def acquire(self):
        buf = self.bufs[self.k % 2]
        self.r.merge_wait(buf.data_ptr())
        return buf

    def release(self):
        buf = self.bufs[self.k % 2]
        self.r.merge_reduce(self.root)     # of the renderer's active buffer: the one acquire() handed out
        self.k += 1
        return buf

    def drain(self):
        self.r.merge_wait(0)


def accum_to_image(accum, width, height):
    """Decode a (merged) accumulation buffer to sum_s sqrt(sample), float64 [H, W, 3], on the host."""
    words = accum.detach().cpu().numpy().view(np.uint64)
    tiles_x, tiles_y = (width + 7) // 8, (height + 7) // 8
    a = words.reshape(tiles_y, tiles_x, 3, 8, 8)  # [tile row][tile col][channel][y in tile][x in tile]
    img = a.transpose(0, 3, 1, 4, 2).reshape(tiles_y * 8, tiles_x * 8, 3)[:height, :width]
    return img.astype(np.float64) / 2.0 ** 36
