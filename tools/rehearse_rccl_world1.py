"""RCCL on a one-GPU box: a process group of one rank runs the collectives bench.py and raytracer_amd/dist.py issue
(reduce, asynchronous reduce followed by a stream-side wait, gather, all_reduce, barrier) on device tensors.  It shows
that the backend initialises and accepts these calls here; it says nothing about bandwidth or about N > 1."""
import os
import sys

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29517")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
fb = torch.full((1080, 1920, 3), 0.5, device="cuda")
dist.reduce(fb, dst=0, op=dist.ReduceOp.SUM)
work = dist.reduce(fb, dst=0, op=dist.ReduceOp.SUM, async_op=True)
work.wait()
fb.mul_(2.0)
send = torch.arange(12, dtype=torch.float32, device="cuda").view(4, 3)
recv = [torch.empty_like(send)]
dist.gather(send, recv, dst=0)
t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
ok = float(fb[0, 0, 0]) == 1.0 and torch.equal(recv[0], send) and float(t) == 1.25
print("rccl world-1 rehearsal:", "ok" if ok else "WRONG", "| backend", dist.get_backend(), "| torch", torch.__version__)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
