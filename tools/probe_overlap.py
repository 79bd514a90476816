"""dev probe: frames of one scene rendered back to back on 1, 2 or 4 HIP streams (independent framebuffers)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from raytracer_amd import abi, tracer as T
import bench

for name in sys.argv[1:] or ["bunny", "cornell", "instanced"]:
    wl = bench.make_workload(name)
    tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=wl["bounces"], sky=wl["sky"]))
    T.rt_tracer_build_blas(tr, wl["world"]); T.rt_tracer_build_tlas(tr, wl["world"])
    W, H = wl["width"], wl["height"]
    cast = wl["cam"](W, H, wl["samples"])
    T.set_counting(tr, True); T.get_stats(tr)
    fb0 = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    tr.check(tr.lib.rthip_cast_device(tr.h, abi.ptr(cast), 1, fb0.data_ptr(), W, H, None, torch.cuda.current_stream().cuda_stream), "cast")
    torch.cuda.synchronize()
    rays = T.get_stats(tr)["rays"]; T.set_counting(tr, False)
    for n_streams, extra in ((1, ""), (4, ""), (4, "accumulate"), (4, "events"), (4, "accumulate events"), (8, "accumulate events")):
        streams = [torch.cuda.Stream() for _ in range(n_streams)]
        fbs = [torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(n_streams)]
        acc = [torch.zeros((H, W, 3), dtype=torch.float32, device="cuda") for _ in range(n_streams)]
        K = int(os.environ.get('PROBE_K', 40)) if rays < 5e7 else 8
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(K):
                s = streams[k % n_streams]
                if "events" in extra: ev[k][0].record(s)
                tr.check(tr.lib.rthip_cast_device(tr.h, abi.ptr(cast), 1 + k, fbs[k % n_streams].data_ptr(), W, H, None, s.cuda_stream), "cast")
                if "events" in extra: ev[k][1].record(s)
                if "accumulate" in extra:
                    with torch.cuda.stream(s):
                        acc[k % n_streams].add_(fbs[k % n_streams])
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"{name:10s} {n_streams} stream(s) {extra:18s}: {dt / K * 1e3:.3f} ms per frame (host enqueue {t_host / K * 1e3:.3f}), {rays * K / dt / 1e6:.0f} Mrays/s", flush=True)
    T.rt_tracer_cleanup(tr)
