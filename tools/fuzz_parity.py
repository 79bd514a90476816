"""dev probe: long fuzz runs of the GPU path against the CPU oracle, bit for bit.
  worlds: seeded random worlds (scenes.random_world) -- trees, closest hits and occlusion of aimed rays, whole frames
          from every cast kernel (the oracle draws the same counter-based random numbers)
  boxes:  random box sets (scenes.random_boxes) -- Morton codes, sorted order, every tree node, in both forms of the build
  bigboxes: the same with up to 150 000 boxes (the sizes of the sort in tiles and of the range form of the climb, and above)
  hugeboxes: the same with up to 600 000 boxes (the sort by the upper key bytes with its two finishing kernels, the climb
          with device-wide hand-offs)
  obj:    random OBJ texts (scenes.random_obj_text) -- vertices and indices
Usage: fuzz_parity.py [worlds|boxes|bigboxes|hugeboxes|obj] [first_seed [count]]"""
import sys, os, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
from raytracer_amd import scenes, tracer as T
from oracle import oracle
from helpers import assert_same_tree, bits, check_random_world, obj_written_columns

oracle.build_oracle()
kind = sys.argv[1] if len(sys.argv) > 1 else "worlds"
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
count = int(sys.argv[3]) if len(sys.argv) > 3 else 200
tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=8, sky=True))
tmp = tempfile.mkdtemp()


def one(seed):
    if kind == "worlds":
        check_random_world(oracle, tr, seed)
    elif kind in ("boxes", "bigboxes", "hugeboxes"):
        boxes = scenes.random_boxes(seed, {"boxes": 20_000, "bigboxes": 150_000, "hugeboxes": 600_000}[kind])
        codes, cs, ids, nodes, _ = T.lbvh_build(tr, boxes)
        ocodes, ocs, oids, onodes = oracle.lbvh_build(boxes)
        assert (codes == ocodes).all() and (cs == ocs).all() and (ids == oids).all(), f"codes / order of {boxes.shape[0]} boxes"
        assert_same_tree(nodes, onodes, f"{boxes.shape[0]} boxes")
        _, cs, ids, nodes, _ = T.lbvh_build(tr, boxes, want_codes=False)  # as a scene builds: the sort makes the codes
        assert (cs == ocs).all() and (ids == oids).all(), f"order of {boxes.shape[0]} boxes, codes made by the sort"
        assert_same_tree(nodes, onodes, f"{boxes.shape[0]} boxes, codes made by the sort")
    else:
        text, attrs = scenes.random_obj_text(seed)
        path = os.path.join(tmp, "fuzz.obj")
        with open(path, "wb") as f:
            f.write(text.encode())
        verts, idx, _ = T.ms_load_obj(tr, path, attrs)
        want_verts, want_idx = oracle.load_obj(path, attrs)
        assert verts.shape == want_verts.shape and (idx == want_idx).all(), "indices"
        cols = obj_written_columns(attrs, verts.shape[1])
        assert (bits(verts[:, cols]) == bits(want_verts[:, cols])).all(), "vertices"


bad = 0
for seed in range(first, first + count):
    try:
        one(seed)
    except AssertionError as e:
        bad += 1
        print(f"{kind} seed {seed}: {e}", flush=True)
    if (seed - first) % 250 == 249:
        print(f"... {kind}: {seed - first + 1} done, {bad} with differences", flush=True)
print(f"{kind}: {count} from seed {first}: {bad} with differences", flush=True)
sys.exit(1 if bad else 0)
