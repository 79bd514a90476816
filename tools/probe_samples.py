"""dev probe: pixel-per-thread against sample-per-lane static kernel (and the persistent one) over frame sizes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import scenes, tracer as T
import bench

for name in sys.argv[1:] or ["cornell", "cornell-mesh", "instanced", "bunny"]:
    wl = bench.make_workload(name)
    tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=wl["bounces"], sky=wl["sky"]))
    T.rt_tracer_build_blas(tr, wl["world"]); T.rt_tracer_build_tlas(tr, wl["world"])
    for (W, H) in ((256, 256), (512, 512), (1024, 1024), (1920, 1080), (2048, 2048)):
        for samples in (2, 4):
            out = np.zeros((H, W, 3), np.float32)
            cast = wl["cam"](W, H, samples)
            res = {}
            for kern in ("static", "static-samples", "persistent"):
                T.set_cast_kernel(tr, kern)
                best = 1e9
                for _ in range(3):
                    T.rt_tracer_cast(tr, cast, out, W, H, seed=1)
                    best = min(best, T.get_stats(tr)["last_cast_ms"])
                res[kern] = best
            print(f"{name:12s} {W}x{H} s{samples}: static {res['static']:.3f}  samples {res['static-samples']:.3f}  persistent {res['persistent']:.3f} ms   "
                  f"pixel waves {((W+7)//8)*((H+7)//8)}", flush=True)
    T.rt_tracer_cleanup(tr)
