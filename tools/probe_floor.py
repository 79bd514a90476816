"""dev probe: how much of the default frame is per-pixel fixed cost?  Renders the bunny frame, the same frame with the
camera turned away from the mesh (every pixel is sky), and both at other workgroup-count-equivalent sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import scenes, tracer as T
import bench

wl = bench.make_workload("bunny")
tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=wl["bounces"], sky=wl["sky"]))
T.rt_tracer_build_blas(tr, wl["world"]); T.rt_tracer_build_tlas(tr, wl["world"])
for kern in ("static", "persistent"):
    T.set_cast_kernel(tr, kern)
    for (W, H) in ((1920, 1080), (3840, 2160)):
        out = np.zeros((H, W, 3), np.float32)
        for label, cast in (("bunny", wl["cam"](W, H, 1)),
                            ("sky only", scenes.get_rt_cast_settings(W, H, 1, (-0.3, 1.2, 3.5), (-0.3, 1.2, 7.0),
                                                                       vfov=scenes.degrees_to_rad(45)))):
            best = 1e9
            for _ in range(6):
                T.rt_tracer_cast(tr, cast, out, W, H, seed=1)
                best = min(best, T.get_stats(tr)["last_cast_ms"])
            print(f"{kern:10s} {W}x{H} {label:8s}: kernel {best:.3f} ms  ({W*H/best/1e3:.0f} Mpixels/s)", flush=True)
T.rt_tracer_cleanup(tr)
