"""dev probe: the LBVH build launched directly and replayed from a captured HIP graph (launch-bound sizes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import tracer as T
tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=1, sky=True))
for n in [int(a) for a in sys.argv[1:]] or [2000, 81920, 1_000_000, 10_000_000]:
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    cen = torch.rand((n, 1, 3), device="cuda", generator=g)
    off = (torch.rand((n, 3, 3), device="cuda", generator=g) * 2 - 1) * 0.001
    p = (cen + off).reshape(3 * n, 3).float()
    px, py, pz = (p[:, k].contiguous() for k in range(3))
    nodes = torch.empty((2 * n - 1, 8), dtype=torch.float32, device="cuda")
    def build(stream):
        tr.check(tr.lib.rthip_mesh_build_device(tr.h, px.data_ptr(), py.data_ptr(), pz.data_ptr(), 3 * n, None, n, nodes.data_ptr(), stream.cuda_stream), "build")
    s = torch.cuda.current_stream()
    def timed(fn, reps=20):
        best = 1e9
        for it in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream()); fn(); e1.record(torch.cuda.current_stream()); torch.cuda.synchronize()
            if it: best = min(best, e0.elapsed_time(e1))
        return best
    direct = timed(lambda: build(torch.cuda.current_stream()))
    want = nodes.clone()
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=side):
        build(torch.cuda.current_stream())
    nodes.zero_()
    replay = timed(graph.replay)
    assert torch.equal(nodes.view(torch.int32), want.view(torch.int32))
    print(f"n={n}: direct {direct:.3f} ms, graph replay {replay:.3f} ms", flush=True)
