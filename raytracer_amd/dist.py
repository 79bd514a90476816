"""Multi-GPU plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on
ROCm, "gloo" for CPU rehearsals).  The BVH is replicated (every rank builds it); only the framebuffer
is exchanged, once per frame:

  tiles    every rank renders the 64x64 tiles t with t % world == rank, other pixels stay 0, so the
           element-wise SUM of the shares is the frame (bit-identical to a single-GPU render);
  samples  every rank renders the sample indices s with s % world == rank of every pixel and leaves
           the plain sum; SUM over ranks then one multiply by 1/samples^2 gives the frame;
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


def combine_framebuffer(fb, mode: str, samples: int, world: int, dst: int = 0):
    """Sums the ranks' framebuffers onto `dst` (in place, one collective) and applies the mode's scale
    there.  `fb` is a torch tensor on the backend's device.  Returns fb (meaningful on dst only)."""
    import torch.distributed as dist

    if world > 1:
        dist.reduce(fb, dst=dst, op=dist.ReduceOp.SUM)
        if dist.get_rank() == dst:
            if mode == "samples":
                fb.mul_(1.0 / float(samples * samples))
            elif mode == "frames":
                fb.mul_(1.0 / float(world))
    return fb
