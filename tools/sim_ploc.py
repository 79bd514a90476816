"""dev study (CPU only, numpy): what a PLOC tree (parallel locally-ordered clustering over the Morton-sorted leaves, the
cheapest of the GPU tree-quality passes) would buy the ordered walk on the benchmark's mesh.  Builds the LBVH (the
oracle's) and PLOC trees of the bunny-class mesh, prints their surface-area costs and the box tests per ray of a 4-wide
nearest-first walk over both (camera rays of the bunny frame, random rays through the mesh's box).
Measured (subdivision level 5, 20 480 triangles): SAH cost 45.5 -> 39.6 (radius 16), box tests -6 % on the camera rays
and -14 % on random rays: far from the 20 % a restructuring pass was to bring, so none was built.
Usage: sim_ploc.py [subdivision level, default 5]"""
import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from oracle import oracle as O
from raytracer_amd import scenes, world

def area(b):  # b[...,6]
    d = np.maximum(b[..., 3:] - b[..., :3], 0)
    return 2 * (d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0])

def ploc(leaf_boxes, R=8):
    n = len(leaf_boxes)
    boxes = leaf_boxes.astype(np.float64).copy()
    refs = np.arange(n)  # node ids: leaves 0..n-1, inner n..
    node_box = list(leaf_boxes.astype(np.float64))
    left = {}; right = {}
    nb = [None] * (2 * n)
    L = np.full(2 * n, -1); Rr = np.full(2 * n, -1)
    B = np.zeros((2 * n, 6)); B[:n] = boxes
    nxt = n
    it = 0
    while len(refs) > 1:
        m = len(refs)
        best = np.full(m, np.inf); nn = np.full(m, -1)
        for d in range(1, R + 1):
            if d >= m: break
            u = np.concatenate([np.minimum(boxes[:-d, :3], boxes[d:, :3]), np.maximum(boxes[:-d, 3:], boxes[d:, 3:])], axis=1)
            a = area(u)
            # pair (i, i+d)
            idx = np.arange(m - d)
            better = a < best[idx]
            best[idx] = np.where(better, a, best[idx]); nn[idx] = np.where(better, idx + d, nn[idx])
            idx2 = idx + d
            better2 = a < best[idx2]
            best[idx2] = np.where(better2, a, best[idx2]); nn[idx2] = np.where(better2, idx, nn[idx2])
        i = np.arange(m)
        mutual = (nn[nn] == i)
        lead = mutual & (i < nn)
        drop = mutual & (i > nn)
        k = np.nonzero(lead)[0]
        ids = nxt + np.arange(len(k))
        L[ids] = refs[k]; Rr[ids] = refs[nn[k]]
        u = np.concatenate([np.minimum(boxes[k, :3], boxes[nn[k], :3]), np.maximum(boxes[k, 3:], boxes[nn[k], 3:])], axis=1)
        B[ids] = u
        boxes[k] = u; refs = refs.copy(); refs[k] = ids
        keep = ~drop
        boxes = boxes[keep]; refs = refs[keep]
        nxt += len(k); it += 1
    return refs[0], L, Rr, B, it

def lbvh_tree(leaf_boxes_sorted_order_nodes):
    pass

def sah(root, L, R, B, n):
    # sum of areas of inner nodes / root area  + leaves
    inner = [root]; tot = 0.0; leaves = 0.0
    stack = [root]
    ra = area(B[root])
    while stack:
        k = stack.pop()
        if L[k] < 0:
            leaves += area(B[k]); continue
        tot += area(B[k])
        stack.append(L[k]); stack.append(R[k])
    return (tot + leaves) / ra

def slab(b, o, inv, tmax):
    t0 = (b[:3] - o) * inv; t1 = (b[3:] - o) * inv
    near = np.minimum(t0, t1); far = np.maximum(t0, t1)
    tn = max(near.max(), 1e-7); tf = min(far.min(), tmax)
    return tf >= tn, tn

def tri_hit(o, d, tri, tmax):
    a, b, c = tri
    e1 = b - a; e2 = c - a
    p = np.cross(d, e2); det = e1 @ p
    if det == 0: return None
    inv = 1.0 / det; s = o - a; u = inv * (s @ p)
    if u < 0 or u > 1: return None
    q = np.cross(s, e1); v = inv * (d @ q)
    if v < 0 or u + v > 1: return None
    t = inv * (e2 @ q)
    if t < 1e-7 or t > tmax: return None
    return t

def slots_fixed(k, L, R, B):
    slots = []
    for c in (L[k], R[k]):
        if L[c] < 0: slots.append(c)
        else: slots.extend([L[c], R[c]])
    return slots

_AREA = {}
def slots_greedy(k, L, R, B, width=4):
    """the node's two children, then the inner slot with the largest box is replaced by its two children until there are
    `width` slots (or none is inner): the usual surface-area-guided collapse of a binary tree into a wide one"""
    slots = [L[k], R[k]]
    while len(slots) < width:
        best = -1; ba = -1.0
        for j, c in enumerate(slots):
            if L[c] >= 0:
                a = area(B[c])
                if a > ba: ba = a; best = j
        if best < 0: break
        c = slots[best]
        slots[best:best + 1] = [L[c], R[c]]
    return slots

SLOTS = slots_fixed
WIDE_STEPS = [0]

def traverse_wide(root, L, R, B, tris, o, d):
    """4-wide (two levels collapsed), nearest first, stack; returns box tests, tri tests"""
    inv = 1.0 / d
    tmax = np.inf; nb = 0; nt = 0
    stack = [(0.0, root)]
    if L[root] < 0:
        return 1, 1
    while stack:
        tn, k = stack.pop()
        if tn > tmax: continue
        if L[k] < 0:
            nt += 1
            t = tri_hit(o, d, tris[k], tmax)
            if t is not None: tmax = t
            continue
        slots = SLOTS(k, L, R, B)
        WIDE_STEPS[0] += 1
        hits = []
        for sidx in slots:
            nb += 1
            h, tn2 = slab(B[sidx], o, inv, tmax)
            if h: hits.append((tn2, sidx))
        hits.sort(key=lambda x: -x[0])
        stack.extend(hits)
    return nb, nt

if __name__ == "__main__":
    level = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    verts, indices = scenes.bunny_class_mesh(level)
    P = verts.reshape(-1, verts.shape[-1])[:, :3].astype(np.float64) if verts.ndim > 1 else None
    V = np.asarray(verts, dtype=np.float32).reshape(len(verts) if verts.ndim>1 else -1, -1)[:, :3].astype(np.float64)
    tri_idx = np.asarray(indices).reshape(-1, 3)
    tv = V[tri_idx]  # [T,3,3]
    tb = np.concatenate([tv.min(axis=1), tv.max(axis=1)], axis=1).astype(np.float32)
    T = len(tb)
    codes, cs, ids, nodes = O.lbvh_build(tb)
    order = ids.astype(np.int64) - 1
    print("tris", T)
    # LBVH pointer tree from pre-order nodes
    nn = len(nodes)
    Ll = np.full(2 * T, -1); Rl = np.full(2 * T, -1); Bl = np.zeros((2 * T, 6))
    # map: leaves -> index = position in morton order (0..T-1); inner -> T + running
    pos_of_prim = np.empty(T, dtype=np.int64); pos_of_prim[order] = np.arange(T)
    node_ref = np.empty(nn, dtype=np.int64); nxt = T
    for i in range(nn):
        if nodes["id"][i]: node_ref[i] = pos_of_prim[nodes["id"][i] - 1]
        else: node_ref[i] = nxt; nxt += 1
    for i in range(nn):
        r = node_ref[i]
        Bl[r, :3] = nodes["aabb_min"][i]; Bl[r, 3:] = nodes["aabb_max"][i]
        if nodes["id"][i] == 0:
            Ll[r] = node_ref[i + 1]; Rl[r] = node_ref[nodes["skip"][i + 1]]
    root_l = node_ref[0]
    leaf_boxes = tb[order]
    tris_by_pos = tv[order]
    print("LBVH SAH", sah(root_l, Ll, Rl, Bl, T))
    for Rr in (8, 16, 32):
        t0 = time.time()
        root_p, Lp, Rp, Bp, it = ploc(leaf_boxes, Rr)
        print(f"PLOC R={Rr}: iterations {it}, SAH {sah(root_p, Lp, Rp, Bp, T):.2f}, {time.time()-t0:.1f}s")
    root_p, Lp, Rp, Bp, it = ploc(leaf_boxes, 16)
    # rays: camera rays of the bunny frame (subset) and random rays through the box
    rng = np.random.default_rng(1)
    w, cam, sky = scenes.bunny(verts, indices)
    cast = cam(96, 54, 1)
    o, d = O.primary_rays(cast, 96, 54)
    sel = rng.choice(len(o), 700, replace=False)
    lo, hi = tb[:, :3].min(0), tb[:, 3:].max(0)
    ro = rng.uniform(lo, hi, (500, 3)); rd = rng.normal(size=(500, 3)); rd /= np.linalg.norm(rd, axis=1)[:, None]
    for name, oo, dd in (("primary", o[sel].astype(np.float64), d[sel].astype(np.float64)), ("random", ro, rd)):
        res = {}
        for tname, (rt, L_, R_, B_) in (("lbvh", (root_l, Ll, Rl, Bl)), ("ploc16", (root_p, Lp, Rp, Bp))):
            nb = nt = 0
            for k in range(len(oo)):
                dk = np.where(dd[k] == 0, 1e-30, dd[k])
                a, b = traverse_wide(rt, L_, R_, B_, tris_by_pos, oo[k], dk)
                nb += a; nt += b
            res[tname] = (nb / len(oo), nt / len(oo))
        print(name, res, "ratio boxes", res["ploc16"][0] / res["lbvh"][0])
