"""rthip_cast (the call behind rt_tracer_cast: a frame into the caller's host buffer) on the default 1080p frame: into
pageable memory and into page-locked memory from rthip_host_alloc, each in 1, 2, 4 and 8 bands of rows (RTHIP_CAST_BANDS;
1 band = one launch and one copy, into pageable memory the runtime's own staged copy).
Run on the GPU box: python3 tools/probe_host_cast.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from raytracer_amd import scenes, tracer as T

verts, idx = scenes.bunny_class_mesh()
w, cam, sky = scenes.bunny(verts, idx)
W, H = 1920, 1080
c = cam(W, H, 1)


def run(bands, pinned, traversal="exact", reps=30):
    os.environ["RTHIP_CAST_BANDS"] = str(bands)
    tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=2, sky=sky))
    T.rt_tracer_build_blas(tr, w)
    T.rt_tracer_build_tlas(tr, w)
    T.set_traversal(tr, traversal)
    img = T.host_alloc(tr, (H, W, 3)) if pinned else np.zeros((H, W, 3), dtype=np.float32)
    for _ in range(5):
        T.rt_tracer_cast(tr, c, img, W, H, seed=1)
    rays = None
    t0 = time.perf_counter()
    for _ in range(reps):
        T.rt_tracer_cast(tr, c, img, W, H, seed=1)
    dt = (time.perf_counter() - t0) / reps
    ms = T.get_stats(tr)["last_cast_ms"]
    checksum = float(img.sum(dtype=np.float64))
    T.rt_tracer_cleanup(tr)
    return dt * 1e3, ms, checksum


for traversal in ("exact", "fast"):
    for pinned in (False, True):
        for bands in (1, 2, 4, 8):
            ms, k, chk = run(bands, pinned, traversal)
            kind = "page-locked" if pinned else "pageable   "
            print(f"{traversal}: {kind}, {bands} band(s) {ms:.3f} ms per call (render {k:.3f} ms)  checksum {chk:.6f}", flush=True)
