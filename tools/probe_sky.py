"""dev probe: the default frame with the camera turned away from the mesh (every ray one box test and the sky):
what a pixel costs before any traversal.  Run under rocprofv3 --pmc SQ_INSTS_VALU / --kernel-trace --stats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import abi, scenes, tracer as T
import bench

wl = bench.make_workload("bunny")
tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=wl["bounces"], sky=wl["sky"]))
T.rt_tracer_build_blas(tr, wl["world"]); T.rt_tracer_build_tlas(tr, wl["world"])
W, H = 1920, 1080
away = sys.argv[1] != "mesh" if len(sys.argv) > 1 else True
cast = scenes.get_rt_cast_settings(W, H, 1, (-0.3, 1.2, 3.5), (-0.3, 1.2, 10.0 if away else 0.0), vfov=scenes.degrees_to_rad(45))
fb = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream()
for _ in range(5):
    tr.check(tr.lib.rthip_cast_device(tr.h, abi.ptr(cast), 1, fb.data_ptr(), W, H, None, s.cuda_stream), "cast")
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(s)
for _ in range(20):
    tr.check(tr.lib.rthip_cast_device(tr.h, abi.ptr(cast), 1, fb.data_ptr(), W, H, None, s.cuda_stream), "cast")
e1.record(s); torch.cuda.synchronize()
print("away" if away else "mesh", "frame ms", e0.elapsed_time(e1) / 20, "mean", float(fb.mean()))
