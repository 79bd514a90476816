"""RCCL on a one-GPU box: a process group of ONE rank runs the collectives bench.py and raytracer_amd/dist.py issue, on
tensors of the shapes they issue them on -- the framebuffer sum of the sample / frame modes (reduce, and the asynchronous
reduce with a stream-side wait of the frame pipeline), the gather of packed 64x64 tiles of the tile mode (packed and
placed by the library's own kernels), the MAX / SUM of the timing figures, barrier.  It shows that the backend
initialises and accepts these calls here; it says nothing about bandwidth or about N > 1."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29517")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
import torch
import torch.distributed as dist

from raytracer_amd import dist as rdist, tracer as T

torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
W, H = 1920, 1080
fb = torch.full((H, W, 3), 0.5, device="cuda")
dist.reduce(fb, dst=0, op=dist.ReduceOp.SUM)
work = dist.reduce(fb, dst=0, op=dist.ReduceOp.SUM, async_op=True)
work.wait()
fb.mul_(2.0)
ok = float(fb[0, 0, 0]) == 1.0

# the frame pipeline of --combine every, as bench.py drives it
pipe = rdist.FramePipeline([torch.zeros((H, W, 3), device="cuda") for _ in range(2)], "samples", 2, 1, dst=0)
for k in range(3):
    b = pipe.acquire()
    b.fill_(float(k))
    pipe.submit(b)
ok = ok and float(pipe.drain()[0, 0, 0]) == 2.0

# tile mode: this rank's tiles packed by the library, gathered, placed again
tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=1, sky=True), device_id=0)
frame = torch.rand((H, W, 3), device="cuda")
slots = int(tr.lib.rthip_tiles_of_rank(W, H, 0, 1))
ok = ok and slots == ((W + 63) // 64) * ((H + 63) // 64)
send = torch.empty(slots * 64 * 64 * 3, dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
tr.check(tr.lib.rthip_tiles_pack_device(tr.h, frame.data_ptr(), W, H, 0, 1, send.data_ptr(), stream), "pack")
recv = [torch.empty_like(send)]
dist.gather(send, recv, dst=0)
back = torch.zeros_like(frame)
tr.check(tr.lib.rthip_tiles_unpack_device(tr.h, recv[0].data_ptr(), W, H, 0, 1, back.data_ptr(), stream), "unpack")
torch.cuda.synchronize()
ok = ok and torch.equal(back, frame)
# ranks of a larger world: every tile is placed by exactly one rank, the shares are disjoint and complete
whole = torch.zeros_like(frame)
for r in range(3):
    n = int(tr.lib.rthip_tiles_of_rank(W, H, r, 3))
    part = torch.empty(max(n, 1) * 64 * 64 * 3, dtype=torch.float32, device="cuda")
    tr.check(tr.lib.rthip_tiles_pack_device(tr.h, frame.data_ptr(), W, H, r, 3, part.data_ptr(), stream), "pack")
    tr.check(tr.lib.rthip_tiles_unpack_device(tr.h, part.data_ptr(), W, H, r, 3, whole.data_ptr(), stream), "unpack")
torch.cuda.synchronize()
ok = ok and torch.equal(whole, frame)
T.rt_tracer_cleanup(tr)

t = torch.tensor([1.25], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.all_reduce(t, op=dist.ReduceOp.SUM)
dist.barrier()
torch.cuda.synchronize()
ok = ok and float(t) == 1.25
print("rccl world-1 rehearsal:", "ok" if ok else "WRONG", "| backend", dist.get_backend(), "| torch", torch.__version__)
dist.destroy_process_group()
sys.exit(0 if ok else 1)
