"""dev probe: batched instance entries (RTHIP_SAMPLES_INST_BATCH) on scenes of small meshes other than Cornell: a grid of
n x n instances of a quad (3 nodes), of a box (23 nodes) and of an icosahedron (39 nodes), at two frame sizes; every
kernel, batch size 0 and 64 (one process per batch size: the variable is read when the context is made).
Usage: RTHIP_SAMPLES_INST_BATCH=<k> probe_entries.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import scenes, tracer as T

quad_v = np.array([[-1, 0, -1, 0, 0, 1, 0, 0], [1, 0, -1, 0, 0, 1, 0, 0], [1, 0, 1, 0, 0, 1, 0, 0], [-1, 0, 1, 0, 0, 1, 0, 0]], np.float32)
quad_i = np.array([0, 1, 2, 0, 2, 3], np.uint32)
box_v = np.array([[x, y, z, 0, x, y, z, 0] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float32)
box_i = np.array([0, 1, 3, 0, 3, 2, 4, 6, 7, 4, 7, 5, 0, 4, 5, 0, 5, 1, 2, 3, 7, 2, 7, 6, 0, 2, 6, 0, 6, 4, 1, 5, 7, 1, 7, 3], np.uint32)
ico_v, ico_i = scenes.bunny_class_mesh(0)
print("RTHIP_SAMPLES_INST_BATCH =", os.environ.get("RTHIP_SAMPLES_INST_BATCH", "(rule)"))
for name, (v, i) in (("cornell", (None, None)), ("quad", (quad_v, quad_i)), ("box", (box_v, box_i)), ("icosahedron", (ico_v, ico_i))):
    for n_side in ((0,) if name == "cornell" else (4, 16, 100)):
        w, cam, sky = scenes.cornell() if name == "cornell" else scenes.instanced_grid(n_side, seed=1, verts=v, indices=i)
        tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=8, sky=sky))
        T.rt_tracer_build_blas(tr, w); T.rt_tracer_build_tlas(tr, w)
        for (W, H, samples) in ((512, 512, 2), (1920, 1080, 1)):
            out = np.zeros((H, W, 3), np.float32)
            cast = cam(W, H, samples)
            T.set_cast_kernel(tr, "static")
            T.set_counting(tr, True); T.get_stats(tr)
            T.rt_tracer_cast(tr, cast, out, W, H, seed=1)
            st = T.get_stats(tr); T.set_counting(tr, False)
            ratio = st["inst_entered"] / max(st["nodes_tested"], 1)
            res = {}
            for kern in ("static", "static-samples", "persistent", "auto"):
                T.set_cast_kernel(tr, kern)
                best = 1e9
                for _ in range(3):
                    T.rt_tracer_cast(tr, cast, out, W, H, seed=1)
                    best = min(best, T.get_stats(tr)["last_cast_ms"])
                res[kern] = best
            print(f"{name:12s} {n_side:3d}^2 instances {W}x{H} s{samples}: " + "  ".join(f"{k} {t:.3f}" for k, t in res.items()) + f" ms   entries/box test {ratio:.3f}, boxes/ray {st['nodes_tested'] / max(st['rays'], 1):.1f}", flush=True)
        T.rt_tracer_cleanup(tr)
