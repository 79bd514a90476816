"""The ordered walk against the exact one on seeded random worlds (scenes.random_world: every material and instance
type, rotated / mirrored / non-uniformly scaled instances, spheres, degenerate and coincident triangles) with rays aimed
at the geometry: rays at another distance, rays with another primitive at the same distance, hit-versus-miss rays and
occlusion answers that differ.  Every fourth world also asks the ordered walk's resident kernels (lanes parked at
instances and entered in batches; the plain kernels enter at once): the same rays as one long stream, and a frame from
the persistent against the pixel-per-lane kernel -- those must agree bit for bit (`resident_differs`).
python3 tools/fuzz_fast.py [first_seed] [count]   (GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from raytracer_amd import scenes, tracer as T

first = int(sys.argv[1]) if len(sys.argv) > 1 else 9000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 600
tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=4, sky=True))
tot = {"worlds": 0, "rays": 0, "t_differs": 0, "tie_other_primitive": 0, "hit_vs_miss": 0, "occlusion_differs": 0,
       "resident_worlds": 0, "resident_differs": 0}
worst = []
for seed in range(first, first + count):
    w, cam, sky, bounces = scenes.random_world(seed)
    tr.check(tr.lib.rthip_set_tracer(tr.h, bounces, 1 if sky else 0), "set_tracer")
    T.rt_tracer_build_blas(tr, w)
    T.rt_tracer_build_tlas(tr, w)
    o, d = scenes.aimed_rays(3000, seed)
    T.set_traversal(tr, "exact")
    a = T.trace_rays(tr, o, d)
    occ_a = T.occluded(tr, o, d, 4.0)
    T.set_traversal(tr, "fast")
    b = T.trace_rays(tr, o, d)
    occ_b = T.occluded(tr, o, d, 4.0)
    if seed % 4 == 0:
        reps = -(-(1 << 18) // o.shape[0])
        long_b = T.trace_rays(tr, np.tile(o, (reps, 1)), np.tile(d, (reps, 1)))
        same = all((x.view(np.uint32) == np.tile(y, (reps,) + (1,) * (y.ndim - 1)).view(np.uint32)).all() for x, y in zip(long_b, b))
        frames = []
        for kernel in ("static", "persistent"):
            T.set_cast_kernel(tr, kernel)
            img = np.zeros((48, 64, 3), dtype=np.float32)
            T.rt_tracer_cast(tr, cam(64, 48, 2), img, 64, 48, seed=seed)
            frames.append(img)
        T.set_cast_kernel(tr, "auto")
        same = same and (frames[0].view(np.uint32) == frames[1].view(np.uint32)).all()
        tot["resident_worlds"] += 1
        tot["resident_differs"] += 0 if same else 1
        if not same:
            print("resident kernels differ from the plain ones: seed", seed, flush=True)
    T.set_traversal(tr, "exact")
    t_diff = a[2].view(np.uint32) != b[2].view(np.uint32)
    prim = (a[0] != b[0]) | (a[1] != b[1])
    hm = (a[0] != 0) != (b[0] != 0)
    tot["worlds"] += 1
    tot["rays"] += o.shape[0]
    tot["t_differs"] += int(t_diff.sum())
    tot["tie_other_primitive"] += int((prim & ~t_diff).sum())
    tot["hit_vs_miss"] += int(hm.sum())
    tot["occlusion_differs"] += int((occ_a != occ_b).sum())
    if t_diff.any() or hm.any() or (occ_a != occ_b).any():
        both = t_diff & ~hm
        rel = np.abs(a[2][both].astype(np.float64) - b[2][both]) / np.maximum(np.abs(a[2][both].astype(np.float64)), 1e-30)
        worst.append((seed, int(t_diff.sum()), int(hm.sum()), int((occ_a != occ_b).sum()),
                      f"max relative distance difference {float(rel.max()) if rel.size else 0.0:.2e}, fast closer on {int((b[2][both] < a[2][both]).sum())}"))
print(tot)
print("worlds with a ray at another distance / hit-vs-miss / occlusion difference (seed, counts):", worst[:40])
