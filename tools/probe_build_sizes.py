"""dev probe: LBVH build time of n random boxes for a list of sizes in one process (best of 5 after a warm-up, HIP events).
python3 tools/probe_build_sizes.py [n ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from raytracer_amd import tracer as T

sizes = [int(a) for a in sys.argv[1:]] or [600, 1500, 3000, 10_000, 30_000, 81_920, 131_072]
tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=1, sky=True))
s = torch.cuda.current_stream()
row = []
for n in sizes:
    g = torch.Generator(device="cuda")
    g.manual_seed(n)
    cen = torch.rand((n, 3), device="cuda", generator=g)
    boxes = torch.cat([cen - 0.001, cen + 0.001], dim=1).contiguous()
    nodes = torch.empty((2 * n - 1, 8), dtype=torch.float32, device="cuda")
    best = 1e9
    for it in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        tr.check(tr.lib.rthip_lbvh_build_device(tr.h, boxes.data_ptr(), n, nodes.data_ptr(), s.cuda_stream), "build")
        e1.record(s)
        torch.cuda.synchronize()
        if it:
            best = min(best, e0.elapsed_time(e1))
    row.append(f"{n}: {best * 1000:.1f} us")
print("  ".join(row))
