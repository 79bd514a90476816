"""dev probe: time the LBVH build of n random triangles (no validation), for kernel-trace runs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracer_amd import tracer as T
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=1, sky=True))
kind = sys.argv[2] if len(sys.argv) > 2 else "soup"
idx = None
if kind == "heightfield":  # a connected surface of about n triangles (scenes.heightfield_world), indexed
    import math
    import numpy as np
    from raytracer_amd import scenes
    side = int(math.sqrt(n / 2))
    mesh = scenes.heightfield_world(side)[0].meshes[0]
    verts = np.asarray(mesh.vertices, dtype=np.float32).reshape(-1, 4)
    p = torch.from_numpy(verts[:, :3].copy()).cuda()
    idx = torch.from_numpy(np.asarray(mesh.indices, dtype=np.uint32).astype(np.int32)).cuda()
    n = idx.numel() // 3
else:
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    cen = torch.rand((n, 1, 3), device="cuda", generator=g)
    off = (torch.rand((n, 3, 3), device="cuda", generator=g) * 2 - 1) * 0.001
    p = (cen + off).reshape(3 * n, 3).float()
px, py, pz = (p[:, k].contiguous() for k in range(3))
nodes = torch.empty((2 * n - 1, 8), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream()
best = 1e9
for it in range(4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    tr.check(tr.lib.rthip_mesh_build_device(tr.h, px.data_ptr(), py.data_ptr(), pz.data_ptr(), px.numel(), idx.data_ptr() if idx is not None else None, n, nodes.data_ptr(), s.cuda_stream), "build")
    e1.record(s); torch.cuda.synchronize()
    if it: best = min(best, e0.elapsed_time(e1))
print(f"n={n} {kind} build {best:.3f} ms")
