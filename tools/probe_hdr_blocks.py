"""dev probe: which of hdr_rows_kernel's three block paths the 256-byte blocks of a frame take (numpy classification of
the RGBE planes; close to, not bit for bit, the encoder's own conversion).  GPU box: renders Cornell and the bunny-class
frame at 3840x2160 and 1920x1080."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import scenes, tracer as T


def planes(rgb):
    m = rgb.max(axis=2)
    mant, ex = np.frexp(m)
    scale = np.where(m < 1e-32, 0.0, mant * 256.0 / np.maximum(m, 1e-38)).astype(np.float32)
    out = np.zeros(rgb.shape[:2] + (4,), dtype=np.uint8)
    out[..., :3] = np.clip((rgb * scale[..., None]).astype(np.int64), 0, 255)
    out[..., 3] = np.where(m < 1e-32, 0, ex + 128)
    return out


def classify(p):  # p: (H, W) bytes of one component
    H, W = p.shape
    e = np.zeros((H, W), dtype=bool)
    e[:, :-1] = p[:, :-1] == p[:, 1:]
    tri = np.zeros((H, W), dtype=bool)
    tri[:, :-2] = e[:, :-2] & e[:, 1:-1]
    run = tri.copy()
    run[:, 1:] |= tri[:, :-1]
    run[:, 2:] |= tri[:, :-2]
    nb = (W + 255) // 256
    counts = np.zeros(3, dtype=np.int64)
    for b in range(nb):
        lo, hi = b * 256, min(W, b * 256 + 256)
        r = run[:, lo:hi]
        before = run[:, lo - 1] if lo > 0 else np.zeros(H, dtype=bool)
        none = ~r.any(axis=1) & ~before
        heads = r & ~np.concatenate([(e[:, lo - 1:lo] if lo > 0 else np.zeros((H, 1), dtype=bool)), e[:, lo:hi - 1]], axis=1)
        inside = r.all(axis=1) & ~heads.any(axis=1)
        counts[0] += int(none.sum())
        counts[1] += int((inside & ~none).sum())
        counts[2] += int((~none & ~inside).sum())
    return counts


tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=4, sky=True))
verts, indices = scenes.bunny_class_mesh()
for name, (w, cam, sky) in (("cornell", scenes.cornell()), ("bunny", scenes.bunny(verts, indices))):
    tr.check(tr.lib.rthip_set_tracer(tr.h, 4, 1 if sky else 0), "set_tracer")
    T.rt_tracer_build_blas(tr, w)
    T.rt_tracer_build_tlas(tr, w)
    for (W, H) in ((1920, 1080), (3840, 2160)):
        out = np.zeros((H, W, 3), np.float32)
        T.rt_tracer_cast(tr, cam(W, H, 1), out, W, H, seed=1)
        pl = planes(out)
        tot = np.zeros(3, dtype=np.int64)
        per = []
        for c in range(4):
            k = classify(pl[..., c])
            per.append(k)
            tot += k
        print(name, W, H, "blocks: no run / inside one run / general =", tot.tolist(), [f"{x / tot.sum():.2f}" for x in tot],
              "per component", [k.tolist() for k in per], flush=True)
T.rt_tracer_cleanup(tr)
