import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import scenes, tracer as T
verts, idx = scenes.bunny_class_mesh()
w, cam, sky = scenes.bunny(verts, idx)
tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=2, sky=sky))
T.rt_tracer_build_blas(tr, w); T.rt_tracer_build_tlas(tr, w)
W,H=1920,1080
c = cam(W,H,1)
img = np.zeros((H,W,3), dtype=np.float32)
for i in range(3): T.rt_tracer_cast(tr, c, img, W, H, seed=1)
t0=time.perf_counter()
for i in range(10): T.rt_tracer_cast(tr, c, img, W, H, seed=1)
dt=(time.perf_counter()-t0)/10
print(f"rthip_cast (host framebuffer) 1080p: {dt*1e3:.3f} ms per call, kernel {T.get_stats(tr)['last_cast_ms']:.3f} ms")
