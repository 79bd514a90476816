import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
from raytracer_amd import scenes, tracer as T
tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=2, sky=True))
verts, indices = scenes.bunny_class_mesh()
w, cam, sky = scenes.bunny(verts, indices)
T.rt_tracer_build_blas(tr, w); T.rt_tracer_build_tlas(tr, w)
for (W, H) in ((1920, 1080), (3840, 2160)):
    out = np.zeros((H, W, 3), np.float32)
    T.rt_tracer_cast(tr, cam(W, H, 1), out, W, H, seed=1)
    best = 1e9
    for _ in range(8):
        data = T.hdr_encode(tr, None, W, H)
        best = min(best, T.get_stats(tr)["last_encode_ms"])
    print(W, H, f"kernels {best:.4f} ms", len(data), flush=True)
