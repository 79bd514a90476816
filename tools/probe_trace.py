"""dev probe: node throughput of the pure traversal kernels (primary rays / random rays)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import scenes, tracer as T, world
import bench
for name in sys.argv[1:] or ["bunny", "instanced", "soup"]:
    wl = bench.make_workload(name)
    tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=wl["bounces"], sky=wl["sky"]))
    T.rt_tracer_build_blas(tr, wl["world"]); T.rt_tracer_build_tlas(tr, wl["world"])
    W, H = wl["width"], wl["height"]
    cast = wl["cam"](W, H, 1)
    T.set_counting(tr, True); T.get_stats(tr)
    T.primary_hits(tr, cast, W, H)
    st = T.get_stats(tr); T.set_counting(tr, False)
    best = 1e9
    for _ in range(5):
        T.primary_hits(tr, cast, W, H); best = min(best, T.get_stats(tr)["last_cast_ms"])
    print(f"{name}: primary rays {st['rays']} nodes/ray {st['nodes_tested']/st['rays']:.1f} lane_util {st['nodes_tested']/(64*st['wave_steps']):.3f} "
          f"kernel {best:.3f} ms -> {st['rays']/best/1e3:.0f} Mrays/s, {st['nodes_tested']/best/1e6:.1f} G nodes/s")
    T.rt_tracer_cleanup(tr)
