"""Multi-GPU plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on
ROCm, "gloo" for CPU rehearsals).  The BVH is replicated (every rank builds it); only the framebuffer
is exchanged, once per frame:

  tiles    every rank renders the 64x64 tiles t with t % world == rank, other pixels stay 0; the shares
           are disjoint, so every rank sends only its own tiles' pixels (1/world of the frame, one gather)
           and the destination places them: bit-identical to a single-GPU render;
  samples  every rank renders the sample indices s with s % world == rank of every pixel and leaves
           the plain sum; SUM over ranks then one multiply by 1/samples^2 gives the frame up to
           rounding (the sum is taken in another order than a single GPU's sample loop: equal to a few
           ulp, not bit for bit -- tiles are the bit-exact mode);
  frames   every rank renders the whole frame with its own seed; SUM / world averages them
           (world times the samples in the same wall time: the weak-scaling mode of bench.py).
"""
from __future__ import annotations

import numpy as np

from . import abi

MODES = {"none": 0, "tiles": 1, "samples": 2, "frames": 0}


def shard_record(mode: str, rank: int, world: int) -> np.ndarray:
    """The rthip_shard a rank passes to rthip_cast_device."""
    sh = np.zeros(1, dtype=abi.SHARD)
    sh[0] = (MODES[mode] if world > 1 else 0, rank if MODES[mode] else 0, world if MODES[mode] else 1)
    return sh


def combine_framebuffer(fb, mode: str, samples: int, world: int, dst: int = 0, tracer=None):
    """Sums the ranks' framebuffers onto `dst` (in place, one collective) and applies the mode's scale
    there.  `fb` is a torch tensor on the backend's device.  Returns fb (meaningful on dst only)."""
    import torch.distributed as dist

    if world > 1 and mode == "tiles" and fb.dim() == 3:
        return gather_tiles(fb, fb.shape[1], fb.shape[0], world, dst, tracer=tracer)  # disjoint shares: moved, not added
    if world > 1:
        dist.reduce(fb, dst=dst, op=dist.ReduceOp.SUM)
        if dist.get_rank() == dst:
            if mode == "samples":
                fb.mul_(1.0 / float(samples * samples))
            elif mode == "frames":
                fb.mul_(1.0 / float(world))
    return fb


_TILE = 64  # rthip_shard's tile edge
_tile_index_cache = {}


def tile_pixels(width: int, height: int, rank: int, world: int, device="cpu"):
    """Flat pixel indices (row-major, torch int64 on `device`) of the pixels a rank renders in tile mode: its 64x64 tiles
    t % world == rank in tile order, each tile's pixels row by row (tiles at the right and bottom edge are clipped)."""
    import torch

    key = (width, height, rank, world, str(device))
    if key not in _tile_index_cache:
        tx, ty = -(-width // _TILE), -(-height // _TILE)
        parts = []
        for t in range(rank, tx * ty, world):
            x0, y0 = (t % tx) * _TILE, (t // tx) * _TILE
            ys = np.arange(y0, min(y0 + _TILE, height))[:, None]
            xs = np.arange(x0, min(x0 + _TILE, width))[None, :]
            parts.append((ys * width + xs).reshape(-1))
        idx = np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64)
        _tile_index_cache[key] = torch.from_numpy(idx.astype(np.int64)).to(device)
    return _tile_index_cache[key]


_TILE_FLOATS = _TILE * _TILE * 3


def gather_tiles(fb, width: int, height: int, world: int, dst: int = 0, tracer=None):
    """Tile mode: every rank sends only the pixels of its own tiles (1/world of the frame) and `dst` places them; the
    frame on `dst` is then complete, bit-identical to a single-GPU render (the shares are disjoint, nothing is added).
    `fb` is this rank's framebuffer [height, width, 3] on the backend's device.  One collective (gather).

    With `tracer` (an RT_Tracer on fb's GPU) the tiles are packed and placed by the library's own kernels
    (rthip_tiles_pack_device / rthip_tiles_unpack_device on the current stream: whole 64x64 tiles, 48 KiB each); without
    one -- the CPU rehearsals over gloo -- by index tensors."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return fb
    rank = dist.get_rank()
    if tracer is not None and fb.is_cuda:
        lib = tracer.lib
        stream = torch.cuda.current_stream().cuda_stream
        slots = [int(lib.rthip_tiles_of_rank(width, height, r, world)) for r in range(world)]
        send = torch.empty(max(slots) * _TILE_FLOATS, dtype=fb.dtype, device=fb.device)
        tracer.check(lib.rthip_tiles_pack_device(tracer.h, fb.data_ptr(), width, height, rank, world, send.data_ptr(), stream),
                     "rthip_tiles_pack_device")
        recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
        if dist.get_backend() == "gloo":  # rehearsal: gloo moves host memory only
            host = send.cpu()
            hrecv = [torch.empty_like(host) for _ in range(world)] if rank == dst else None
            dist.gather(host, hrecv, dst=dst)
            if rank == dst:
                for r in range(world):
                    recv[r].copy_(hrecv[r])
        else:
            dist.gather(send, recv, dst=dst)
        if rank == dst:
            for r in range(world):
                if r != rank and slots[r]:
                    tracer.check(lib.rthip_tiles_unpack_device(tracer.h, recv[r].data_ptr(), width, height, r, world, fb.data_ptr(),
                                                               stream), "rthip_tiles_unpack_device")
            for t in recv:  # the unpack kernels read them on this stream after this function has returned
                t.record_stream(torch.cuda.current_stream())
        return fb
    flat = fb.view(-1, 3)
    counts = [int(tile_pixels(width, height, r, world, "cpu").numel()) for r in range(world)]
    longest = max(counts)
    send = torch.zeros((longest, 3), dtype=fb.dtype, device=fb.device)
    mine = tile_pixels(width, height, rank, world, fb.device)
    send[: counts[rank]] = flat[mine]
    recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, recv, dst=dst)
    if rank == dst:
        for r in range(world):
            if r != rank and counts[r]:
                flat[tile_pixels(width, height, r, world, fb.device)] = recv[r][: counts[r]]
    return fb


class _Done:
    """A finished collective (the synchronous rehearsal path)."""

    def wait(self):
        return True


class FramePipeline:
    """Frames in flight: the collective of frame k runs while frame k + 1 is rendered.

    `buffers` are framebuffer tensors of one shape (two are enough).  Per frame the caller does

        fb = pipe.acquire()      # a buffer whose previous reduce has finished (stream-ordered on a GPU)
        ... render into fb ...
        pipe.submit(fb)          # starts the asynchronous SUM onto dst

    and `drain()` at the end, which returns the last combined frame (meaningful on dst).  On the
    "nccl" (RCCL) backend `wait()` only makes the current stream wait for the collective, so the host
    keeps enqueueing; on "gloo" it blocks, which is all a CPU rehearsal needs.
    """

    def __init__(self, buffers, mode: str, samples: int, world: int, dst: int = 0, tracer=None):
        assert len(buffers) >= 1
        self.tracer = tracer  # an RT_Tracer on the buffers' GPU: tile shares are packed / placed by the library's kernels
        self.buffers = list(buffers)
        self.pending = [None] * len(self.buffers)
        self.mode, self.samples, self.world, self.dst = mode, samples, world, dst
        self.next = 0
        self.last = None

    @staticmethod
    def _gloo() -> bool:
        import torch.distributed as dist

        return dist.get_backend() == "gloo"

    def _finish(self, k: int) -> None:
        work = self.pending[k]
        if work is None:
            return
        import torch.distributed as dist

        work.wait()
        self.pending[k] = None
        if dist.get_rank() == self.dst:
            if self.mode == "samples":
                self.buffers[k].mul_(1.0 / float(self.samples * self.samples))
            elif self.mode == "frames":
                self.buffers[k].mul_(1.0 / float(self.world))

    def acquire(self):
        k = self.next
        self._finish(k)
        return self.buffers[k]

    def submit(self, fb) -> None:
        k = self.next
        assert fb is self.buffers[k], "submit the buffer acquire() returned"
        if self.world > 1 and self.mode == "tiles" and (self.tracer is not None or not (fb.is_cuda and self._gloo())):
            # disjoint shares: one gather of 1/world of the frame per rank; stream-ordered on RCCL, blocking on gloo
            gather_tiles(fb, fb.shape[1], fb.shape[0], self.world, self.dst, tracer=self.tracer)
            self.pending[k] = None
        elif self.world > 1:
            import torch.distributed as dist

            if fb.is_cuda and dist.get_backend() == "gloo":
                # rehearsal on a box without RCCL peers: gloo has no GPU reduce, so stage through the host
                host = fb.cpu()
                dist.reduce(host, dst=self.dst, op=dist.ReduceOp.SUM)
                if dist.get_rank() == self.dst:
                    fb.copy_(host)
                self.pending[k] = _Done()
            else:
                self.pending[k] = dist.reduce(fb, dst=self.dst, op=dist.ReduceOp.SUM, async_op=True)
        self.last = k
        self.next = (k + 1) % len(self.buffers)

    def drain(self):
        for k in range(len(self.buffers)):
            self._finish(k)
        return None if self.last is None else self.buffers[self.last]
