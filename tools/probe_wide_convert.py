"""The ordered walk's records (rthip_set_traversal FAST) for the 10^7-triangle soup: wall time of the call that builds
them from the finished tree.  Under `rocprofv3 --kernel-trace --stats` the wide_* and scan kernels give the device time."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import scenes, tracer as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
w, cam, sky = scenes.soup_world(n, seed=1)
for rep in range(2):
    tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=4, sky=sky))
    T.rt_tracer_build_blas(tr, w)
    T.rt_tracer_build_tlas(tr, w)
    t0 = time.perf_counter()
    T.set_traversal(tr, "fast")
    ms = (time.perf_counter() - t0) * 1e3
    print(f"{n} triangles: rthip_set_traversal(FAST) {ms:.2f} ms (wall: record arrays allocated, five kernels and two scans, one wait)", flush=True)
    T.rt_tracer_cleanup(tr)
