"""dev probe: incoherent ray streams traced as given, and after sorting the rays by the Morton code of their origin."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import abi, tracer as T
import bench

def morton3(q):  # q: int64 [n,3] with 10 bits each
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)

for name in sys.argv[1:] or ["soup-rays", "instanced-rays"]:
    wl = bench.make_workload(name)
    tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=wl["bounces"], sky=wl["sky"]))
    T.rt_tracer_build_blas(tr, wl["world"]); T.rt_tracer_build_tlas(tr, wl["world"])
    T.set_traversal(tr, os.environ.get("TRAVERSAL", "exact"))  # TRAVERSAL=fast: the ordered walk
    n = wl["rays"]
    g = torch.Generator(device="cuda"); g.manual_seed(2)
    lo, hi = wl.get("ray_box", ((0.0, 0.0, 0.0), (1.0, 1.0, 1.0)))
    lo, hi = torch.tensor(lo, device="cuda"), torch.tensor(hi, device="cuda")
    o = (lo + torch.rand((n, 3), device="cuda", generator=g) * (hi - lo)).contiguous()
    v = torch.randn((n, 3), device="cuda", generator=g)
    d = (v / v.norm(dim=1, keepdim=True)).contiguous()
    hits = torch.zeros((n, 4), dtype=torch.int32, device="cuda")
    stream = torch.cuda.current_stream()
    def run(oo, dd, label):
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            tr.check(tr.lib.rthip_trace_rays_device(tr.h, oo.data_ptr(), dd.data_ptr(), n, hits.data_ptr(), stream.cuda_stream), "trace")
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print(f"{name:15s} {label:34s}: {best*1e3:8.2f} ms  {n/best/1e6:8.1f} Mrays/s", flush=True)
        return hits.clone()
    h0 = run(o, d, "as given")
    for bits in (4, 6, 8, 10):
        t0 = time.perf_counter()
        q = ((o - lo) / (hi - lo) * (1 << bits)).long().clamp_(0, (1 << bits) - 1)
        key = morton3(q)
        octant = ((d[:, 0] < 0).long() | ((d[:, 1] < 0).long() << 1) | ((d[:, 2] < 0).long() << 2))
        for with_dir in (False, True):
            k = (key << 3) | octant if with_dir else key
            perm = torch.argsort(k)
            os_, ds_ = o[perm].contiguous(), d[perm].contiguous()
            torch.cuda.synchronize()
            h = run(os_, ds_, f"origin Morton {bits} bits/axis" + (" + octant" if with_dir else ""))
            back = torch.empty_like(h); back[perm] = h
            assert torch.equal(back, h0)
    T.rt_tracer_cleanup(tr)
