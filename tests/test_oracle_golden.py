"""The CPU oracle against outputs of the reference itself (tests/golden, written by
oracle/make_golden.py from oracle/_ref).  Bit-exact everywhere the reference is deterministic."""
import glob
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden
from helpers import assert_same_tree, assert_same_tree_modulo_runs, bits, nan_free, same_or_both_nan
from raytracer_amd import scenes, world

LBVH_SETS = sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN, "lbvh_*.npz")) if "big" not in p)


@pytest.mark.parametrize("name", LBVH_SETS)
def test_lbvh_matches_reference(oracle, name):
    g = golden("lbvh_" + name)
    codes, codes_sorted, ids, nodes = oracle.lbvh_build(g["aabbs"])
    assert (codes == g["ref_codes"]).all(), "Morton codes differ from the reference"
    assert (np.sort(g["ref_codes"]) == codes_sorted).all()
    want = oracle.parse_dump(g["ref_dump"].tobytes())
    if codes.size <= 28 or np.unique(codes).size == codes.size:
        # no duplicate codes (or a set the reference sorts stably): byte-identical dump
        assert oracle.dump_bytes(nodes) == g["ref_dump"].tobytes()
        assert (ids == g["ref_ids_tree_order"]).all()
        assert_same_tree(nodes, want, name)
    else:
        assert_same_tree_modulo_runs(nodes, want, codes_sorted, name)


def test_lbvh_big_digests(oracle):
    g = golden("lbvh_big_digests")
    rng = np.random.default_rng(7)
    for n in (100_000, 1_000_000):
        cen = rng.random((n, 3)).astype(np.float32)
        half = (rng.random((n, 3)) * 0.001).astype(np.float32)
        boxes = np.concatenate([cen - half, cen + half], axis=1).astype(np.float32)
        assert hashlib.sha256(boxes.tobytes()).hexdigest() == str(g[f"n{n}_aabbs"]), "input generator drifted"
        if n > 100_000:
            continue  # the 1M case is covered on the GPU side; keep the CPU suite short
        codes, cs, ids, nodes = oracle.lbvh_build(boxes)
        assert int(str(g[f"n{n}_unique_codes"])) == n, "fixture was meant to be duplicate-free"
        assert hashlib.sha256(codes.tobytes()).hexdigest() == str(g[f"n{n}_codes"])
        assert hashlib.sha256(ids.tobytes()).hexdigest() == str(g[f"n{n}_ids"])
        assert hashlib.sha256(oracle.dump_bytes(nodes)).hexdigest() == str(g[f"n{n}_dump"])


def test_lbvh_debug_flavour_matches_reference(oracle):
    """A BUILD_DEBUG build of the reference normalises centroids by the bounds of the boxes (lbvh.c:101-104): other
    codes, other trees.  The oracle's box-bounds mode against that build's answers on the same sets."""
    g = golden("debug_flavour")
    oracle.set_morton_bounds(True)
    try:
        differs = 0
        for name in g["names"].tolist():
            boxes = golden("lbvh_" + name)["aabbs"]
            codes, codes_sorted, ids, nodes = oracle.lbvh_build(boxes)
            assert (codes == g[name + "_ref_codes"]).all(), name
            assert_same_tree_modulo_runs(nodes, oracle.parse_dump(g[name + "_ref_dump"].tobytes()), codes_sorted, name)
            differs += int((codes != golden("lbvh_" + name)["ref_codes"]).any())
        assert differs >= 8, "the two flavours should disagree on most sets"
        # the out.bvh of the reference's Cornell demo, debug build
        w, cam, sky = scenes.cornell()
        sc = oracle.Scene(world.flatten_world(w), 8, sky)
        assert oracle.dump_bytes(sc.tree(-1)) == g["cornell_demo_tlas_dump"].tobytes()
        assert oracle.dump_bytes(sc.tree(-1)) != golden("demo_cornell")["ref_tlas_dump"].tobytes()
    finally:
        oracle.set_morton_bounds(False)


def test_lbvh_random_sets_against_reference_when_built(oracle):
    """Fuzzed box sets (scenes.random_boxes: clusters, lattices, planes, lines, 40 decades of magnitude, denormals,
    identical boxes, negative zeros) against lbvh_make of the reference compiled here."""
    if not oracle.have_ref("core"):
        pytest.skip("oracle/_ref is not built here")
    for seed in range(1, 121):
        boxes = scenes.random_boxes(seed)
        for debug in (False, True):  # release and BUILD_DEBUG flavour of lbvh_make
            oracle.set_morton_bounds(debug)
            try:
                codes, codes_sorted, ids, nodes = oracle.lbvh_build(boxes)
            finally:
                oracle.set_morton_bounds(False)
            ref_codes, _, ref_dump, _ = oracle.ref_lbvh_build(boxes, flavour="core_debug" if debug else "core")
            assert (codes == ref_codes).all(), f"set {seed}: Morton codes differ from the reference"
            assert_same_tree_modulo_runs(nodes, oracle.parse_dump(ref_dump), codes_sorted, f"set {seed}")


DEMOS = {"cornell": scenes.cornell, "tri": scenes.tri, "spheres": scenes.spheres}


@pytest.mark.parametrize("name", sorted(DEMOS))
def test_demo_scene_matches_reference(oracle, name):
    g = golden("demo_" + name)
    w, cam, sky = DEMOS[name]()
    flat = world.flatten_world(w)
    cast = cam(64, 64, 1)
    assert cast.tobytes() == g["ref_cast"].tobytes(), "camera record differs from get_rt_cast_settings"
    sc = oracle.Scene(flat, 8, sky)
    assert oracle.dump_bytes(sc.tree(-1)) == g["ref_tlas_dump"].tobytes(), "TLAS differs from the reference's out.bvh"
    # camera rays: the oracle's generator vs the reference's own ray dump (debug build)
    c16 = cam(16, 16, 1)
    if name == "spheres":
        pass  # defocus draws random numbers; covered by the radiance comparison below
    else:
        o, d = oracle.primary_rays(c16, 16, 16)
        assert (bits(o) == bits(g["ref_debug_rays_16_o"])).all() and (bits(d) == bits(g["ref_debug_rays_16_d"])).all()
    inst, tri, t, uv, nrm, _ = sc.trace(g["rays_o"], g["rays_d"])
    assert (inst == g["ref_inst"]).all() and (tri == g["ref_tri"]).all() and (bits(t) == bits(g["ref_t"])).all()
    img, _ = sc.cast(cast, 1, 64, 64, rng_mode=oracle.RNG_GLIBC)
    assert img.tobytes() == g["ref_rgb_s1_seed1"].tobytes(), "radiance (samples=1, seed 1) is not bit-exact"
    img, _ = sc.cast(cam(32, 32, 2), 3, 32, 32, rng_mode=oracle.RNG_GLIBC)
    assert img.tobytes() == g["ref_rgb_32_s2_seed3"].tobytes(), "radiance (samples=2, seed 3) is not bit-exact"


def test_cornell_512_primary_hits(oracle):
    g = golden("demo_cornell")
    w, cam, sky = scenes.cornell()
    sc = oracle.Scene(world.flatten_world(w), 8, sky)
    o, d = oracle.primary_rays(cam(512, 512, 1), 512, 512)
    inst, tri, t, _, _, _ = sc.trace(o, d, threads=4, want_normal=False)
    assert int((inst > 0).sum()) == int(g["ref_512_hits"]) == 251920
    assert hashlib.sha256(inst.tobytes()).hexdigest() == str(g["ref_512_inst_sha"])
    assert hashlib.sha256(tri.tobytes()).hexdigest() == str(g["ref_512_tri_sha"])
    assert hashlib.sha256(t.tobytes()).hexdigest() == str(g["ref_512_t_sha"])


def test_bunny_class_matches_reference(oracle):
    g = golden("demo_bunny_class")
    verts, indices = scenes.bunny_class_mesh()
    assert hashlib.sha256(verts.tobytes()).hexdigest() == str(g["mesh_verts_sha"]), "stand-in mesh is not reproducible"
    assert hashlib.sha256(indices.tobytes()).hexdigest() == str(g["mesh_indices_sha"])
    w, cam, sky = scenes.bunny(verts, indices)
    sc = oracle.Scene(world.flatten_world(w), 2, sky)
    assert hashlib.sha256(oracle.dump_bytes(sc.tree(0))).hexdigest() == str(g["ref_blas_dump_sha"])
    cast = cam(160, 90, 1)
    assert cast.tobytes() == g["cast_160x90"].tobytes()
    o, d = oracle.primary_rays(cast, 160, 90)
    inst, tri, t, uv, nrm, _ = sc.trace(o, d)
    assert (inst == g["ref_inst"]).all() and (tri == g["ref_tri"]).all() and (bits(t) == bits(g["ref_t"])).all()
    assert (bits(uv) == bits(g["ref_uv"])).all() and (bits(nrm) == bits(g["ref_normal"])).all()
    img, _ = sc.cast(cam(96, 54, 2), 5, 96, 54, rng_mode=oracle.RNG_GLIBC)
    assert img.tobytes() == g["ref_rgb_96x54_s2_b2_seed5"].tobytes()


def test_glibc_rand_clone(oracle):
    import ctypes

    libc = ctypes.CDLL("libc.so.6")
    for seed in (0, 1, 2, 12345, 2**31 - 1):
        libc.srand(seed)
        oracle.lib().ora_srand(seed)
        for _ in range(1000):
            assert libc.rand() == oracle.lib().ora_rand()


def test_counter_modes_agree_statistically(oracle):
    """The counter-RNG flavour (what the GPU computes) against the reference's own seed-to-seed noise."""
    g = golden("demo_cornell")
    w, cam, sky = scenes.cornell()
    sc = oracle.Scene(world.flatten_world(w), 8, sky)
    a, b = g["ref_rgb_64_s8_seed1"], g["ref_rgb_64_s8_seed2"]
    img, _ = sc.cast(cam(64, 64, 8), 1, 64, 64, rng_mode=oracle.RNG_COUNTER, threads=8)
    rmse = lambda x, y: float(np.sqrt(np.mean((x.astype(np.float64) - y) ** 2)))
    assert rmse(img, a) <= 1.25 * rmse(a, b)
    lum = lambda x: float(x.astype(np.float64).mean())
    assert abs(lum(img) - lum(a)) <= 0.01 * lum(a) + 2.0 * abs(lum(a) - lum(b))


def test_counter_streams_of_neighbouring_seeds_are_unrelated(oracle):
    """Frames of consecutive seeds are averaged (accumulation, weak-scaling ranks), so a pixel's streams under seed s
    must not be the streams of seed s ^ 1 handed to other strata, and the upper seed word must not merely XOR the
    draws with a constant."""
    L = oracle.lib()
    draws = lambda seed, pixel, sample: tuple(L.ora_counter_rand(seed, pixel, sample, d) for d in range(4))
    for spp in (4, 16, 64):
        for pixel in (0, 1, 777, 1920 * 1080 - 1):
            sets = [{draws(seed, pixel, s) for s in range(spp)} for seed in range(8)]
            for i in range(8):
                assert len(sets[i]) == spp
                for j in range(i + 1, 8):
                    assert not (sets[i] & sets[j]), f"seeds {i} and {j} share streams at pixel {pixel}, {spp} spp"
    # the first draws alone must differ too (a shared key word with another XOR mask would pass the test above)
    firsts = [{L.ora_counter_rand(seed, 5, s, 0) ^ L.ora_counter_rand(seed, 5, s, 1) for s in range(16)} for seed in range(4)]
    for i in range(4):
        for j in range(i + 1, 4):
            assert not (firsts[i] & firsts[j])
    hi = 1 << 32
    x = {L.ora_counter_rand(1, 9, 0, d) ^ L.ora_counter_rand(1 | hi, 9, 0, d) for d in range(16)}
    assert len(x) == 16, "the upper seed word only XORs the stream with a constant"
    # two (pixel, sample) pairs never share a whole key: 64 bits of it are independent
    keys = {(L.ora_counter_rand(3, p, s, 0), L.ora_counter_rand(3, p, s, 1)) for p in range(4096) for s in range(4)}
    assert len(keys) == 4096 * 4


def _occlusion_scenes():
    verts, indices = scenes.bunny_class_mesh()
    return {"cornell": (scenes.cornell(), 8), "bunny": (scenes.bunny(verts, indices), 2)}


def test_occlusion_matches_reference(oracle):
    """The interval-limited TLAS query (shadow rays) against the reference's own answers."""
    g = golden("occlusion")
    for name, ((w, cam, sky), bounces) in _occlusion_scenes().items():
        sc = oracle.Scene(world.flatten_world(w), bounces, sky)
        o, d = g[f"{name}_o"], g[f"{name}_d"]
        for k in range(3):
            got = sc.occluded(o, d, float(g[f"{name}_tmax{k}"]))
            assert (got == g[f"{name}_ref_occluded{k}"]).all(), f"{name} tmax {k}"
        # with the full interval, occlusion is "the closest-hit query found something"
        inst = sc.trace(o, d, want_normal=False)[0]
        assert ((inst > 0).astype(np.uint8) == g[f"{name}_ref_occluded2"]).all()


def test_edge_case_rays_match_reference(oracle):
    """Axis-parallel directions (1/0 and 0 * inf in the slab test), origins on box planes, edges and vertices,
    degenerate / coincident / sliver triangles: the reference's own answers."""
    g = golden("edge_rays")
    for name, w in (("edge", scenes.edge_scene()), ("cornell", scenes.cornell()[0])):
        sc = oracle.Scene(world.flatten_world(w), 4, True)
        inst, tri, t, uv, nrm, _ = sc.trace(g[f"{name}_o"], g[f"{name}_d"])
        assert (inst == g[f"{name}_ref_inst"]).all() and (tri == g[f"{name}_ref_tri"]).all(), name
        assert (bits(t) == bits(g[f"{name}_ref_t"])).all() and (bits(uv) == bits(g[f"{name}_ref_uv"])).all(), name
        assert (bits(nrm) == bits(g[f"{name}_ref_normal"])).all(), name


def test_random_worlds_match_reference(oracle):
    """Seeded random worlds (every material type, mirrored and non-uniformly scaled instances, spheres, degenerate
    triangles, the three camera kinds): closest hits of aimed rays and a path-traced frame drawn with glibc's
    rand(), against the reference's own answers."""
    g = golden("random_worlds")
    for seed in g["seeds"].tolist():
        w, sky, bounces, o, d, cast = scenes.random_world_case(seed)
        sc = oracle.Scene(world.flatten_world(w), bounces, sky)
        inst, tri, t, uv, nrm, _ = sc.trace(o, d)
        assert (inst == g[f"w{seed}_ref_inst"]).all() and (tri == g[f"w{seed}_ref_tri"]).all(), seed
        assert (bits(t) == bits(g[f"w{seed}_ref_t"])).all(), seed
        assert hashlib.sha256(uv.tobytes()).hexdigest() == str(g[f"w{seed}_ref_uv_sha"]), seed
        assert hashlib.sha256(nan_free(nrm).tobytes()).hexdigest() == str(g[f"w{seed}_ref_normal_sha"]), seed
        rgb, _ = sc.cast(cast, seed, 24, 16, rng_mode=oracle.RNG_GLIBC)
        assert same_or_both_nan(rgb, g[f"w{seed}_ref_rgb"]).all(), f"world {seed}: frame differs from the reference's"
        sc.close()


def test_random_worlds_against_reference_when_built(oracle):
    """Fresh random worlds against the reference compiled here (oracle/_ref): hits, occlusion and whole frames."""
    if not oracle.have_ref("core"):
        pytest.skip("oracle/_ref is not built here")
    for seed in range(7000, 7150):
        w, cam, sky, bounces = scenes.random_world(seed, coincident=False)
        flat = world.flatten_world(w)
        sc, rs = oracle.Scene(flat, bounces, sky), oracle.RefScene(flat, bounces, sky)
        o, d = scenes.aimed_rays(2000, seed)
        got, want = sc.trace(o, d, threads=4), rs.trace(o, d)
        assert (got[0] == want[0]).all() and (got[1] == want[1]).all() and (bits(got[2]) == bits(want[2])).all(), seed
        assert (bits(got[3]) == bits(want[3])).all() and same_or_both_nan(got[4], want[4]).all(), seed
        assert (sc.occluded(o, d, 4.0) == rs.occluded(o, d, 4.0)).all(), seed
        wd, ht, smp = 8 + seed % 50, 8 + seed % 31, 1 + seed % 4
        a, _ = sc.cast(cam(wd, ht, smp), seed, wd, ht, rng_mode=oracle.RNG_GLIBC)
        b, _ = rs.cast(cam(wd, ht, smp), seed, wd, ht)
        assert same_or_both_nan(a, b).all(), f"world {seed}: frame differs from the reference's"
        sc.close()
        rs.close()


def _hdr_names():
    g = golden("hdr")
    return sorted(k[:-4] for k in g.files if k.endswith("_hdr"))


def test_hdr_encoder_matches_reference_writer(oracle):
    """The restated .hdr writer against the bytes the reference's stbi_write_hdr produced (tests/golden/hdr.npz):
    every run-length branch (widths 1..300, runs of 1..256 equal bytes) and every special pixel value."""
    g = golden("hdr")
    names = _hdr_names()
    assert len(names) >= 20
    for name in names:
        got = oracle.hdr_encode(g[name + "_rgb"])
        assert got == g[name + "_hdr"].tobytes(), name


def test_hdr_test_images_are_reproducible():
    g = golden("hdr")
    for name, img in scenes.hdr_test_images().items():
        assert bits(img).tobytes() == bits(g[name + "_rgb"]).tobytes(), f"generator of {name} drifted"


def test_hdr_encoder_against_reference_when_built(oracle):
    """Random images through the reference itself (only where oracle/_ref has been built)."""
    if not oracle.have_ref("hdr"):
        pytest.skip("oracle/_ref/libref_hdr.so is not built here")
    rng = np.random.default_rng(11)
    for w, h in ((64, 5), (63, 4), (65, 4), (500, 7), (4097, 2), (7, 9), (32768, 1)):
        img = (rng.random((h, w, 3), dtype=np.float32) * np.float32(4.0)).astype(np.float32)
        img[:, w // 3: w // 2] = img[:, w // 3: w // 3 + 1]  # a run in every scanline
        assert oracle.hdr_encode(img) == oracle.ref_hdr_encode(img), (w, h)


def test_obj_loader_matches_reference(oracle, tmp_path):
    """The restated ms_load_obj against the reference's results (tests/golden/obj.npz), including the last line that
    the reference's read loop takes twice when the file ends in a newline."""
    from helpers import obj_written_columns

    g = golden("obj")
    files = scenes.obj_test_files()
    assert len(files) >= 6
    for name, (text, attrs) in files.items():
        assert hashlib.sha256(text.encode()).hexdigest() == str(g[name + "_sha"]), f"generator of {name} drifted"
        path = tmp_path / (name + ".obj")
        path.write_bytes(text.encode())
        verts, idx = oracle.load_obj(str(path), attrs)
        written = np.ascontiguousarray(verts[:, obj_written_columns(attrs, verts.shape[1])])
        if name + "_ref_indices" in g.files:
            assert (idx == g[name + "_ref_indices"]).all(), name
            assert bits(written).tobytes() == bits(g[name + "_ref_vertices"]).tobytes(), name
        else:
            assert (verts.shape[0], idx.size) == tuple(g[name + "_ref_counts"]), name
            assert hashlib.sha256(idx.tobytes()).hexdigest() == str(g[name + "_ref_indices_sha"]), name
            assert hashlib.sha256(written.tobytes()).hexdigest() == str(g[name + "_ref_vertices_sha"]), name
    # the trailing face of a file that ends in a newline is a duplicated triangle, without the newline it is not
    assert g["pnt_cube_ref_indices"].size == 39 and g["pt_strip_ref_indices"].size == 9


RAY_RECORD_CASES = [(n, c) for n in ("cornell", "tri", "spheres") for c in ((16, 16, 1, 1, 1), (12, 10, 2, 4, 3))]


def same_ray_records(mine, g, key):
    """abi.RAY_RECORD array against the reference's out.rays fields stored under `key`: float fields bit for bit,
    materials as the same grouping (the reference writes handles, i.e. addresses)."""
    assert len(mine) == len(g[key + "_t"]), key
    for a, b in (("o", "origin"), ("d", "direction"), ("p", "p"), ("n", "n"), ("t", "t")):
        assert bits(np.ascontiguousarray(mine[b])).tobytes() == bits(g[key + "_" + a]).tobytes(), (key, a)
    pairs = set(zip(g[key + "_m"].tolist(), mine["material"].tolist()))
    assert len(pairs) == len(set(g[key + "_m"].tolist())) == len(set(mine["material"].tolist())), key
    assert ((g[key + "_m"] == 0) == (mine["material"] < 0)).all(), key  # misses carry a zero handle


def test_obj_random_texts_against_reference_when_built(oracle, tmp_path):
    """Fuzzed OBJ texts (scenes.random_obj_text) through the reference's ms_load_obj compiled here."""
    from helpers import obj_written_columns

    if not oracle.have_ref("core"):
        pytest.skip("oracle/_ref is not built here")
    path = tmp_path / "fuzz.obj"
    for seed in range(1, 151):
        text, attrs = scenes.random_obj_text(seed)
        path.write_bytes(text.encode())
        verts, idx = oracle.load_obj(str(path), attrs)
        ref_verts, ref_idx = oracle.ref_load_obj(str(path), attrs, len(text) + 16, len(text) + 16)
        assert verts.shape == ref_verts.shape and (idx == ref_idx).all(), f"text {seed}"
        cols = obj_written_columns(attrs, verts.shape[1])
        assert (bits(verts[:, cols]) == bits(ref_verts[:, cols])).all(), f"text {seed}"


@pytest.mark.parametrize("name,case", RAY_RECORD_CASES)
def test_ray_records_match_the_reference_debug_dump(oracle, name, case):
    """Every ray of a frame (origin, direction, hit point, normal, distance, material) as the reference's BUILD_DEBUG
    build dumps it to out.rays, multi-bounce paths included: the restatement with the glibc rand() order."""
    W, H, smp, b, seed = case
    g = golden("ray_records")
    key = f"{name}_{W}x{H}_s{smp}_b{b}_seed{seed}"
    w, cam, sky = {"cornell": scenes.cornell, "tri": scenes.tri, "spheres": scenes.spheres}[name]()
    sc = oracle.Scene(world.flatten_world(w), b, sky)
    same_ray_records(sc.cast_records(g[key + "_cast"], seed, W, H, rng_mode=oracle.RNG_GLIBC), g, key)
