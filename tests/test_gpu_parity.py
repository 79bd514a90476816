"""Parity of the HIP path (through the C ABI of librthip.so) with the CPU oracle and with the
reference's own outputs (tests/golden).  Integer results and float bit patterns must be equal."""
import glob
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden
from helpers import assert_same_tree, bits, check_random_world, nan_free
from raytracer_amd import abi, scenes, tracer as T, world

pytestmark = pytest.mark.gpu

LBVH_SETS = sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN, "lbvh_*.npz")) if "big" not in p)


def check_lbvh(oracle, gpu_tracer, boxes, what):
    codes, cs, ids, nodes, ms = T.lbvh_build(gpu_tracer, boxes)
    ocodes, ocs, oids, onodes = oracle.lbvh_build(boxes)
    assert (codes == ocodes).all(), f"{what}: Morton codes"
    assert (cs == ocs).all(), f"{what}: sorted codes"
    assert (ids == oids).all(), f"{what}: sorted ids"
    assert_same_tree(nodes, onodes, what)
    # once more as a scene's build runs: nobody asks for the unsorted codes, so the sort's first kernel makes them
    _, cs2, ids2, nodes2, _ = T.lbvh_build(gpu_tracer, boxes, want_codes=False)
    assert (cs2 == ocs).all() and (ids2 == oids).all(), f"{what}: sorted codes / ids of the build without the codes' output"
    assert_same_tree(nodes2, onodes, what + " (build without the codes' output)")
    return codes, nodes


@pytest.mark.parametrize("name", LBVH_SETS)
def test_lbvh_golden_sets(oracle, gpu_tracer, name):
    g = golden("lbvh_" + name)
    codes, nodes = check_lbvh(oracle, gpu_tracer, g["aabbs"], name)
    assert (codes == g["ref_codes"]).all(), "Morton codes differ from the reference"
    if codes.size <= 28 or np.unique(codes).size == codes.size:
        assert oracle.dump_bytes(nodes) == g["ref_dump"].tobytes(), "tree dump differs from the reference's"


@pytest.mark.parametrize("n", [1, 2, 3, 5, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1024, 4095, 4096, 4097, 8193, 50_000])
def test_lbvh_sizes(oracle, gpu_tracer, n):
    rng = np.random.default_rng(n)
    cen = rng.random((n, 3)).astype(np.float32)
    half = (rng.random((n, 3)) * 0.01).astype(np.float32)
    check_lbvh(oracle, gpu_tracer, np.concatenate([cen - half, cen + half], axis=1), f"n={n}")


def test_lbvh_random_sets(oracle, gpu_tracer):
    """Fuzzed box sets (clusters, lattices, planes, lines, 40 decades of magnitude, denormals, identical boxes,
    negative zeros; 1 ... 20 000 boxes): codes, order and every node against the oracle."""
    for seed in range(1, 201):
        check_lbvh(oracle, gpu_tracer, scenes.random_boxes(seed), f"random set {seed}")


def test_lbvh_box_bounds_flavour(oracle, gpu_tracer):
    """RTHIP_BOUNDS_BOXES: the trees of the reference's BUILD_DEBUG builds (centroids normalised by the bounds of the
    boxes, lbvh.c:101-104) -- against that build's own codes, the oracle in the same mode, and its Cornell out.bvh."""
    g = golden("debug_flavour")
    T.set_morton_bounds(gpu_tracer, True)
    oracle.set_morton_bounds(True)
    try:
        for name in g["names"].tolist():
            codes, _ = check_lbvh(oracle, gpu_tracer, golden("lbvh_" + name)["aabbs"], name + " (box bounds)")
            assert (codes == g[name + "_ref_codes"]).all(), name
        for seed in range(300, 360):  # through the one-launch path, the fused sort and the general one
            check_lbvh(oracle, gpu_tracer, scenes.random_boxes(seed), f"random set {seed} (box bounds)")
        w, cam, sky = scenes.cornell()
        build(gpu_tracer, w, 8, sky)
        assert oracle.dump_bytes(T.get_tree(gpu_tracer, -1)) == g["cornell_demo_tlas_dump"].tobytes()
        sc = oracle.Scene(world.flatten_world(w), 8, sky)
        for k in range(2):
            assert_same_tree(T.get_tree(gpu_tracer, k), sc.tree(k), f"Cornell BLAS {k} (box bounds)")
        verts, indices = scenes.bunny_class_mesh(4)  # 5120 triangles: bounds from the triangle pass
        w, cam, sky = scenes.bunny(verts, indices)
        build(gpu_tracer, w, 2, sky)
        assert_same_tree(T.get_tree(gpu_tracer, 0), oracle.Scene(world.flatten_world(w), 2, sky).tree(0), "mesh (box bounds)")
    finally:
        T.set_morton_bounds(gpu_tracer, False)
        oracle.set_morton_bounds(False)
    with pytest.raises(T.RTHipError):
        gpu_tracer.check(gpu_tracer.lib.rthip_set_morton_bounds(gpu_tracer.h, 2), "rthip_set_morton_bounds")


def test_lbvh_duplicate_heavy_large(oracle, gpu_tracer):
    rng = np.random.default_rng(3)
    n = 200_000
    cen = rng.integers(0, 12, (n, 3)).astype(np.float32)  # <= 1728 distinct centroids: long runs
    half = (rng.random((n, 3)) * 0.3).astype(np.float32)
    check_lbvh(oracle, gpu_tracer, np.concatenate([cen - half, cen + half], axis=1), "duplicate-heavy")
    same = np.tile(np.array([[1, 2, 3, 4, 5, 6]], dtype=np.float32), (5000, 1))
    check_lbvh(oracle, gpu_tracer, same, "all-identical")


def test_lbvh_one_million_digest(oracle, gpu_tracer):
    g = golden("lbvh_big_digests")
    rng = np.random.default_rng(7)
    for n in (100_000, 1_000_000):
        cen = rng.random((n, 3)).astype(np.float32)
        half = (rng.random((n, 3)) * 0.001).astype(np.float32)
        boxes = np.concatenate([cen - half, cen + half], axis=1).astype(np.float32)
        nodes2 = T.lbvh_build(gpu_tracer, boxes, want_codes=False)[3]
        codes, cs, ids, nodes, ms = T.lbvh_build(gpu_tracer, boxes)
        assert nodes2.tobytes() == nodes.tobytes(), "the build whose sort makes the codes gives another tree"
        assert hashlib.sha256(codes.tobytes()).hexdigest() == str(g[f"n{n}_codes"])
        assert hashlib.sha256(ids.tobytes()).hexdigest() == str(g[f"n{n}_ids"])
        assert hashlib.sha256(oracle.dump_bytes(nodes)).hexdigest() == str(g[f"n{n}_dump"]), "tree differs from the reference's"


def test_lbvh_is_deterministic(gpu_tracer):
    rng = np.random.default_rng(11)
    n = 300_000
    cen = rng.random((n, 3)).astype(np.float32)
    boxes = np.concatenate([cen - 0.001, cen + 0.001], axis=1).astype(np.float32)
    a = T.lbvh_build(gpu_tracer, boxes)[3]
    for _ in range(3):
        b = T.lbvh_build(gpu_tracer, boxes)[3]
        assert a.tobytes() == b.tobytes()


def test_hierarchy_hand_off_modes_build_the_same_tree(oracle):
    """The upper hierarchy kernels hand records over in one of three ways (RTHIP_REFIT_FENCE: 2 relaxed atomics ordered by
    draining the stores, the default; 0 release / acquire through the counter; 1 __threadfence): the same tree from each."""
    rng = np.random.default_rng(21)
    n = 400_000
    cen = rng.random((n, 3)).astype(np.float32)
    cen[: n // 4] = np.floor(cen[: n // 4] * 6) / 6  # a quarter of them on a lattice: long runs of equal codes
    half = (rng.random((n, 3)) * 0.002).astype(np.float32)
    boxes = np.concatenate([cen - half, cen + half], axis=1)
    want = oracle.lbvh_build(boxes)[3]
    old = os.environ.get("RTHIP_REFIT_FENCE")
    try:
        for mode in ("0", "1", "2"):
            os.environ["RTHIP_REFIT_FENCE"] = mode
            t = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=1, sky=True), device_id=0)
            try:
                for want_codes in (True, False):
                    assert_same_tree(T.lbvh_build(t, boxes, want_codes=want_codes)[3], want, f"hand-off mode {mode}")
            finally:
                T.rt_tracer_cleanup(t)
    finally:
        if old is None:
            os.environ.pop("RTHIP_REFIT_FENCE", None)
        else:
            os.environ["RTHIP_REFIT_FENCE"] = old


def test_fast_hand_off_is_taken_only_after_its_self_check(capfd):
    """A context checks the fast form of the device-wide hand-off on its device before its first build that needs it
    (one tree of 400 000 boxes built both ways), once; smaller builds never ask for it."""
    saved = {k: os.environ.pop(k, None) for k in ("RTHIP_REFIT_FENCE",)}
    os.environ["RTHIP_TUNE_LOG"] = "1"
    t = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=1, sky=True), device_id=0)
    try:
        rng = np.random.default_rng(5)
        small = rng.random((50_000, 3)).astype(np.float32)
        capfd.readouterr()
        T.lbvh_build(t, np.concatenate([small, small + 1e-3], axis=1), want_codes=False)
        assert "self-check" not in capfd.readouterr().err
        big = rng.random((200_000, 3)).astype(np.float32)
        a = T.lbvh_build(t, np.concatenate([big, big + 1e-3], axis=1), want_codes=False)[3]
        assert "hand-off self-check passed" in capfd.readouterr().err
        b = T.lbvh_build(t, np.concatenate([big, big + 1e-3], axis=1), want_codes=False)[3]
        assert "self-check" not in capfd.readouterr().err and a.tobytes() == b.tobytes()
    finally:
        T.rt_tracer_cleanup(t)
        os.environ.pop("RTHIP_TUNE_LOG", None)
        for k, v in saved.items():
            if v is not None:
                os.environ[k] = v


@pytest.mark.parametrize("n", [513, 1023, 1024, 1025, 2048, 2049, 4097, 16_384, 16_385, 65_537, 131_072, 131_073, 262_144, 262_145])
def test_lbvh_sizes_at_the_seams_of_the_builder(oracle, n):
    """Sizes just below, at and above the points where the builder changes its path: one workgroup (512), tiles of the
    tile sort (1024 keys up to 16 384 boxes, 2048 above), tile sort + merge against radix passes (131 072), small against large radix tiles (262 144), the
    upper box climb in one workgroup (131 072).  Half of the boxes sit on a lattice (runs of equal codes, pairs among
    them, as the two triangles of a quad give).  Every size against the oracle, and the same tree when the sort goes
    pass by pass, when the upper climb runs device-wide, and when builds of up to 131 072 leaves climb by hand-offs in one
    workgroup instead of taking the nodes above the chunks from leaf ranges."""
    rng = np.random.default_rng(n)
    cen = rng.random((n, 3)).astype(np.float32)
    cen[: n // 2] = np.floor(cen[: n // 2] * 40) / 40
    cen[1 : n // 4 : 2] = cen[0 : n // 4 - 1 : 2][: len(cen[1 : n // 4 : 2])]  # pairs with the same centre
    half = np.full((n, 3), 0.001, dtype=np.float32)
    boxes = np.concatenate([cen - half, cen + half], axis=1)
    want = oracle.lbvh_build(boxes)[3]
    saved = {k: os.environ.get(k) for k in ("RTHIP_SORT_BY_PASSES", "RTHIP_UPPER_DEVICE_WIDE", "RTHIP_SORT_SMALL_TILE_MAX", "RTHIP_UPPER_RANGES")}
    try:
        for env in ({}, {"RTHIP_SORT_BY_PASSES": "1", "RTHIP_UPPER_DEVICE_WIDE": "1"}, {"RTHIP_SORT_BY_PASSES": "1", "RTHIP_SORT_SMALL_TILE_MAX": "0"},
                    {"RTHIP_UPPER_RANGES": "0"}, {"RTHIP_UPPER_RANGES": "1"}):
            for k in saved:
                os.environ.pop(k, None)
            os.environ.update(env)
            t = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=1, sky=True), device_id=0)
            try:
                assert_same_tree(T.lbvh_build(t, boxes)[3], want, f"n = {n}, {env}")
                assert_same_tree(T.lbvh_build(t, boxes, want_codes=False)[3], want, f"n = {n}, {env}, codes made by the sort")
            finally:
                T.rt_tracer_cleanup(t)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("case", ["uniform", "clusters", "dense_cloud", "identical", "stretch_128", "stretch_129", "stretch_2048",
                                  "stretch_2049"])
def test_lbvh_sort_by_upper_bytes(oracle, capfd, case):
    """Builds of more than 131 072 boxes sort by the upper four key bytes and finish the sort inside the stretches of keys
    that share them, on the device: up to 128 keys by the finishing kernel, longer stretches that are out of order by a
    kernel of their own (up to 2048 keys by ranking in LDS, beyond that by four counting passes of one workgroup).  No
    build waits for an answer; word of a long stretch reaches the host behind the build and the NEXT build of that many
    boxes sorts by all eight bytes (it says so with RTHIP_TUNE_LOG).  Every case against the oracle: uniform boxes
    (stretches are rare pairs), clusters of 60 boxes within 1e-5 (every stretch needs work, many straddle the finishing
    kernel's tiles), a cloud of 300 000 boxes inside 1e-4 of the scene (one long stretch, out of order), 300 000
    identical boxes (one long stretch, in order: nothing to do), and stretches of exactly 128, 129, 2048 and 2049 keys
    in descending order."""
    n = 300_000
    rng = np.random.default_rng(11)
    cen = rng.random((n, 3)).astype(np.float32)
    long_stretch = False
    if case == "clusters":
        centres = rng.random((5000, 3)).astype(np.float32)
        cen = centres[rng.integers(0, 5000, n)] + (rng.random((n, 3)).astype(np.float32) - 0.5) * 1e-5
    elif case == "dense_cloud":
        cen = 0.5 + (rng.random((n, 3)).astype(np.float32) - 0.5) * 1e-4
        long_stretch = True
    elif case == "identical":
        cen[:] = 0.375
    elif case.startswith("stretch_"):
        # y = z = 0.25: a key's upper 31 bits hold the upper 10 bits of x, so the keys of one 2^-10 wide cell of x form a
        # stretch; the cells' keys are handed over with descending x, i.e. every stretch is in reverse order
        length = int(case.split("_")[1])
        cells = min(700, 200_000 // length)
        j = np.arange(length, dtype=np.float64)[::-1]
        xs = (np.arange(cells, dtype=np.float64)[:, None] + 17) / 1024.0 + (j[None, :] + 1) * 2.0 ** -22
        m = cells * length
        cen[:m, 0] = xs.reshape(-1).astype(np.float32)
        cen[:m, 1:] = 0.25
        cen[m:, 0] = 0.9 + 0.1 * cen[m:, 0]  # the rest lies beyond the cells
        long_stretch = length > 128
    cen[-2], cen[-1] = (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)  # fix the bounds
    half = np.full((n, 3), 1e-6, dtype=np.float32)
    boxes = np.concatenate([cen - half, cen + half], axis=1).astype(np.float32)
    want = oracle.lbvh_build(boxes)[3]
    saved = os.environ.get("RTHIP_TUNE_LOG")
    os.environ["RTHIP_TUNE_LOG"] = "1"
    t = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=1, sky=True), device_id=0)
    try:
        capfd.readouterr()
        assert_same_tree(T.lbvh_build(t, boxes, want_codes=False)[3], want, case)  # finished on the device, whatever it took
        assert "all eight passes" not in capfd.readouterr().err
        assert_same_tree(T.lbvh_build(t, boxes)[3], want, case + ", second build")
        said = "all eight passes" in capfd.readouterr().err  # the first build's word has arrived by now
        assert said == long_stretch, f"{case}: second build went by all eight passes: {said}, expected {long_stretch}"
        assert_same_tree(T.lbvh_build(t, boxes, want_codes=False)[3], want, case + ", third build")
        assert "all eight passes" not in capfd.readouterr().err
    finally:
        T.rt_tracer_cleanup(t)
        if saved is None:
            os.environ.pop("RTHIP_TUNE_LOG", None)
        else:
            os.environ["RTHIP_TUNE_LOG"] = saved


@pytest.mark.parametrize("case", ["uniform", "clusters", "dense_cloud", "identical", "stretch_31", "stretch_32", "stretch_200"])
def test_lbvh_tile_sort_by_upper_bytes(oracle, case):
    """Builds of up to 131 072 boxes sort tiles of keys in LDS by the upper four key bytes, finish the stretches of keys
    that share them by ranking inside the tile (up to 31 keys), start the tile again with all eight digits when a
    stretch is longer, and merge the tiles by rank; in the default form of the climb the tile sort makes the Morton
    codes itself.  Every case against the oracle -- uniform boxes, clusters of 20 boxes within 1e-5 (every stretch needs
    work, many straddle tiles), a cloud inside 1e-4 of the scene and identical boxes (long stretches: all digits),
    stretches of exactly 31, 32 and 200 keys in descending order -- with each tile size, with all eight digits from the
    start, and with the climb by hand-offs (codes from their own kernel)."""
    n = 60_000
    rng = np.random.default_rng(17)
    cen = rng.random((n, 3)).astype(np.float32)
    if case == "clusters":
        centres = rng.random((3000, 3)).astype(np.float32)
        cen = centres[rng.integers(0, 3000, n)] + (rng.random((n, 3)).astype(np.float32) - 0.5) * 1e-5
    elif case == "dense_cloud":
        cen[: n // 2] = 0.5 + (rng.random((n // 2, 3)).astype(np.float32) - 0.5) * 1e-4
    elif case == "identical":
        cen[: n // 2] = 0.375
    elif case.startswith("stretch_"):
        length = int(case.split("_")[1])
        cells = 20_000 // length
        j = np.arange(length, dtype=np.float64)[::-1]
        xs = (np.arange(cells, dtype=np.float64)[:, None] + 17) / 1024.0 + (j[None, :] + 1) * 2.0 ** -20
        m = cells * length
        cen[:m, 0] = xs.reshape(-1).astype(np.float32)  # one stretch per 2^-10 wide cell of x, handed over in reverse
        cen[:m, 1:] = 0.25
        cen[m:, 0] = 0.9 + 0.1 * cen[m:, 0]  # the rest lies beyond the cells
    cen[-2], cen[-1] = (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)  # fix the bounds
    half = np.full((n, 3), 1e-6, dtype=np.float32)
    boxes = np.concatenate([cen - half, cen + half], axis=1).astype(np.float32)
    want = oracle.lbvh_build(boxes)[3]
    names = ("RTHIP_SORT_MERGE_TILE", "RTHIP_SORT_TOP_BYTES", "RTHIP_UPPER_RANGES")
    saved = {k: os.environ.get(k) for k in names}
    try:
        for env in ({}, {"RTHIP_SORT_MERGE_TILE": "1024"}, {"RTHIP_SORT_MERGE_TILE": "4096"}, {"RTHIP_SORT_TOP_BYTES": "8"},
                    {"RTHIP_UPPER_RANGES": "0"}, {"RTHIP_UPPER_RANGES": "1"}):
            for k in names:
                os.environ.pop(k, None)
            os.environ.update(env)
            t = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=1, sky=True), device_id=0)
            try:
                assert_same_tree(T.lbvh_build(t, boxes)[3], want, f"{case}, {env}")
                assert_same_tree(T.lbvh_build(t, boxes, want_codes=False)[3], want, f"{case}, {env}, codes made by the sort")
            finally:
                T.rt_tracer_cleanup(t)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_large_build_can_be_captured_in_a_hip_graph(gpu_tracer):
    """A build of more than 131 072 boxes only enqueues kernels (its sort by the upper key bytes is finished on the device,
    long stretches included), so it can be captured as it is.  The captured build, replayed on other boxes -- with short
    stretches, equal keys, and one stretch of 5 000 keys in reverse order -- gives the tree of the build outside the
    capture."""
    import torch

    lib, h = gpu_tracer.lib, gpu_tracer.h
    n = 200_000
    gpu_tracer.check(lib.rthip_lbvh_reserve(h, n), "rthip_lbvh_reserve")

    def boxes(seed):
        g = torch.Generator(device="cuda")
        g.manual_seed(seed)
        cen = torch.rand((n, 3), device="cuda", generator=g)
        cen[: n // 2] = torch.floor(cen[: n // 2] * 300) / 300  # stretches of keys with equal upper bytes, equal keys among them
        if seed == 2:  # a long stretch, out of order: 5 000 centres inside one 2^-10 cell of x, descending
            cen[-5000:, 0] = 0.5 + 2.0 ** -11 - torch.arange(5000, device="cuda") * 2.0 ** -24
            cen[-5000:, 1:] = 0.25
        return torch.cat([cen - 0.001, cen + 0.001], dim=1).contiguous()

    buf = boxes(1)
    nodes = torch.zeros((2 * n - 1, 8), dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=s):
        gpu_tracer.check(lib.rthip_lbvh_build_device(h, buf.data_ptr(), n, nodes.data_ptr(), s.cuda_stream), "build (capture)")
    for seed in (1, 2):
        buf.copy_(boxes(seed))
        nodes.fill_(-1)
        graph.replay()
        torch.cuda.synchronize()
        ref = torch.zeros((2 * n - 1, 8), dtype=torch.int32, device="cuda")
        gpu_tracer.check(lib.rthip_lbvh_build_device(h, buf.data_ptr(), n, ref.data_ptr(), torch.cuda.current_stream().cuda_stream), "build")
        torch.cuda.synchronize()
        assert torch.equal(nodes, ref), f"replay with seed {seed}"


def test_lbvh_tile_merge_with_lopsided_tiles(oracle, gpu_tracer):
    """The merge of the tile-sorted keys copies, for 64 neighbouring keys of one tile, the stretch of every other tile
    that lies between them; a stretch of 128 keys or more is searched in global memory instead.  Here the tiles of 1024
    keys are made to differ: spread over the whole range, packed into a thousandth of it, all equal, and spread again in
    descending order, so that both paths and the equal-key rule (earlier tile first) are taken."""
    rng = np.random.default_rng(5)
    x = np.concatenate([
        np.sort(rng.random(1024)),                      # spread
        0.5 + np.sort(rng.random(1024)) * 1e-3,         # packed: 1024 keys between two neighbours of the first tile
        np.full(1024, 0.5005),                          # one code 1024 times, inside the packed range
        np.sort(rng.random(1024))[::-1],                # spread, descending
        0.5 + rng.random(700) * 1e-3,                   # a last, partial tile, packed as well
    ]).astype(np.float32)
    cen = np.stack([x, np.full_like(x, 0.25), np.full_like(x, 0.75)], axis=1)
    cen = np.concatenate([cen, np.array([[0, 0, 0], [1, 1, 1]], dtype=np.float32)])  # fix the bounds
    half = np.full(cen.shape, 1e-4, dtype=np.float32)
    boxes = np.concatenate([cen - half, cen + half], axis=1)
    assert_same_tree(T.lbvh_build(gpu_tracer, boxes)[3], oracle.lbvh_build(boxes)[3], "lopsided tiles")


def test_lbvh_ten_million_invariants(gpu_tracer):
    """The benchmark's largest build (10^7 triangles), checked on the GPU through size-independent properties:
    a well-formed pre-order tree, every primitive in exactly one leaf with its own box, every inner box the
    exact merge of its two children, and twice the same bits."""
    from helpers import build_and_check_invariants

    build_and_check_invariants(gpu_tracer, 10_000_000, seed=3)


DEMOS = {"cornell": scenes.cornell, "tri": scenes.tri, "spheres": scenes.spheres}


def build(gpu_tracer, w, bounces, sky):
    gpu_tracer.check(gpu_tracer.lib.rthip_set_tracer(gpu_tracer.h, bounces, 1 if sky else 0), "set_tracer")
    T.rt_tracer_build_blas(gpu_tracer, w)
    T.rt_tracer_build_tlas(gpu_tracer, w)


@pytest.mark.parametrize("name", sorted(DEMOS))
def test_demo_scene(oracle, gpu_tracer, name):
    g = golden("demo_" + name)
    w, cam, sky = DEMOS[name]()
    build(gpu_tracer, w, 8, sky)
    flat = world.flatten_world(w)
    sc = oracle.Scene(flat, 8, sky)
    assert_same_tree(T.get_tree(gpu_tracer, -1), sc.tree(-1), name + " TLAS")
    assert oracle.dump_bytes(T.get_tree(gpu_tracer, -1)) == g["ref_tlas_dump"].tobytes()
    for k in range(len(flat.meshes)):
        assert_same_tree(T.get_tree(gpu_tracer, k), sc.tree(k), f"{name} BLAS {k}")
    # the reference's own answers for the pixel-centre rays
    cast = cam(64, 64, 1)
    inst, tri, t, uv, nrm = T.trace_rays(gpu_tracer, g["rays_o"], g["rays_d"])
    assert (inst == g["ref_inst"]).all() and (tri == g["ref_tri"]).all() and (bits(t) == bits(g["ref_t"])).all()
    oi, ot, ott, ouv, onrm, _ = sc.trace(g["rays_o"], g["rays_d"])
    assert (bits(uv) == bits(ouv)).all() and (bits(nrm) == bits(onrm)).all()
    if name != "spheres":  # the spheres camera has defocus, which primary_hits leaves out like the fixture does
        pi, pt, ptt = T.primary_hits(gpu_tracer, cast, 64, 64)
        assert (pi == g["ref_inst"]).all() and (pt == g["ref_tri"]).all() and (bits(ptt) == bits(g["ref_t"])).all()
    # full path tracing: bit-exact against the oracle in counter-RNG mode
    for (wd, ht, smp, seed) in ((64, 64, 1, 1), (48, 40, 2, 9), (33, 17, 3, 2)):
        c = cam(wd, ht, smp)
        img = np.zeros((ht, wd, 3), dtype=np.float32)
        T.rt_tracer_cast(gpu_tracer, c, img, wd, ht, seed=seed)
        want, _ = sc.cast(c, seed, wd, ht, rng_mode=oracle.RNG_COUNTER, threads=8)
        bad = np.nonzero((bits(img) != bits(want)).any(axis=2))
        assert bad[0].size == 0, f"{name} {wd}x{ht} s{smp}: {bad[0].size} pixels differ, first {bad[0][:4]},{bad[1][:4]}"


def test_cornell_512_primary_hits_match_reference(gpu_tracer):
    g = golden("demo_cornell")
    w, cam, sky = scenes.cornell()
    build(gpu_tracer, w, 8, sky)
    inst, tri, t = T.primary_hits(gpu_tracer, cam(512, 512, 1), 512, 512)
    assert int((inst > 0).sum()) == int(g["ref_512_hits"])
    assert hashlib.sha256(inst.tobytes()).hexdigest() == str(g["ref_512_inst_sha"])
    assert hashlib.sha256(tri.tobytes()).hexdigest() == str(g["ref_512_tri_sha"])
    assert hashlib.sha256(t.tobytes()).hexdigest() == str(g["ref_512_t_sha"])


def test_cornell_radiance_within_reference_noise(gpu_tracer):
    g = golden("demo_cornell")
    w, cam, sky = scenes.cornell()
    build(gpu_tracer, w, 8, sky)
    img = np.zeros((64, 64, 3), dtype=np.float32)
    T.rt_tracer_cast(gpu_tracer, cam(64, 64, 8), img, 64, 64, seed=1)
    a, b = g["ref_rgb_64_s8_seed1"], g["ref_rgb_64_s8_seed2"]
    rmse = lambda x, y: float(np.sqrt(np.mean((x.astype(np.float64) - y) ** 2)))
    # tolerance: the reference's own seed-to-seed RMSE at equal spp, times 1.25; mean luminance within 1 %
    # (plus the reference's own seed-to-seed luminance spread)
    assert rmse(img, a) <= 1.25 * rmse(a, b)
    lum = lambda x: float(x.astype(np.float64).mean())
    assert abs(lum(img) - lum(a)) <= 0.01 * lum(a) + 2.0 * abs(lum(a) - lum(b))


def test_cornell_converges_to_the_reference(gpu_tracer):
    """1024 spp on the GPU against the reference's 4096-spp image: the relative MSE may be at most twice the
    reference's own 1024-spp image's (a different noise realisation of the same estimator), and the means agree."""
    g = golden("cornell_converged")
    w, cam, sky = scenes.cornell()
    build(gpu_tracer, w, 8, sky)
    img = np.zeros((32, 32, 3), dtype=np.float32)
    T.rt_tracer_cast(gpu_tracer, cam(32, 32, 32), img, 32, 32, seed=5)
    truth, ref1024 = g["ref_rgb_32_s64_seed1"].astype(np.float64), g["ref_rgb_32_s32_seed2"].astype(np.float64)
    rel = lambda x: float(np.mean((x.astype(np.float64) - truth) ** 2 / (truth ** 2 + 1e-3)))
    assert rel(img) <= 2.0 * rel(ref1024), f"relMSE {rel(img):.4f} vs the reference's own {rel(ref1024):.4f}"
    assert abs(float(img.mean()) - float(truth.mean())) <= 0.01 * float(truth.mean())


def test_bunny_class_scene(oracle, gpu_tracer):
    g = golden("demo_bunny_class")
    verts, indices = scenes.bunny_class_mesh()
    w, cam, sky = scenes.bunny(verts, indices)
    build(gpu_tracer, w, 2, sky)
    assert hashlib.sha256(oracle.dump_bytes(T.get_tree(gpu_tracer, 0))).hexdigest() == str(g["ref_blas_dump_sha"])
    cast = cam(160, 90, 1)
    o, d = oracle.primary_rays(cast, 160, 90)
    inst, tri, t, uv, nrm = T.trace_rays(gpu_tracer, o, d)
    assert (inst == g["ref_inst"]).all() and (tri == g["ref_tri"]).all() and (bits(t) == bits(g["ref_t"])).all()
    assert (bits(uv) == bits(g["ref_uv"])).all() and (bits(nrm) == bits(g["ref_normal"])).all()
    inst, tri, t = T.primary_hits(gpu_tracer, cam(1920, 1080, 1), 1920, 1080)
    assert int((inst > 0).sum()) == int(g["ref_1080_hits"])
    assert hashlib.sha256(inst.tobytes()).hexdigest() == str(g["ref_1080_inst_sha"])
    assert hashlib.sha256(tri.tobytes()).hexdigest() == str(g["ref_1080_tri_sha"])
    assert hashlib.sha256(t.tobytes()).hexdigest() == str(g["ref_1080_t_sha"])
    sc = oracle.Scene(world.flatten_world(w), 2, sky)
    c = cam(96, 54, 2)
    img = np.zeros((54, 96, 3), dtype=np.float32)
    T.rt_tracer_cast(gpu_tracer, c, img, 96, 54, seed=5)
    want, _ = sc.cast(c, 5, 96, 54, rng_mode=oracle.RNG_COUNTER, threads=8)
    assert (bits(img) == bits(want)).all()


def test_instanced_scene_and_counters(oracle, gpu_tracer):
    verts, indices = scenes.bunny_class_mesh(3)  # 1280 triangles
    w, cam, sky = scenes.instanced_grid(12, seed=1, verts=verts, indices=indices)
    build(gpu_tracer, w, 4, sky)
    flat = world.flatten_world(w)
    sc = oracle.Scene(flat, 4, sky)
    assert_same_tree(T.get_tree(gpu_tracer, -1), sc.tree(-1), "instanced TLAS")
    o, d = scenes.random_rays(20_000, seed=2, lo=(-15, -2, -15), hi=(15, 3, 15))
    T.set_counting(gpu_tracer, True)
    T.get_stats(gpu_tracer)
    inst, tri, t, uv, nrm = T.trace_rays(gpu_tracer, o, d)
    st = T.get_stats(gpu_tracer)
    T.set_counting(gpu_tracer, False)
    oi, ot, ott, ouv, onrm, cnt = sc.trace(o, d, threads=8)
    assert (inst == oi).all() and (tri == ot).all() and (bits(t) == bits(ott)).all()
    assert (bits(uv) == bits(ouv)).all() and (bits(nrm) == bits(onrm)).all()
    assert int((inst > 0).sum()) > 1000
    for k in ("rays", "nodes_tested", "tris_tested", "inst_entered"):
        assert st[k] == cnt[k], f"traversal counter {k}: {st[k]} vs {cnt[k]}"
    c = cam(80, 45, 2)
    img = np.zeros((45, 80, 3), dtype=np.float32)
    T.rt_tracer_cast(gpu_tracer, c, img, 80, 45, seed=4)
    want, _ = sc.cast(c, 4, 80, 45, rng_mode=oracle.RNG_COUNTER, threads=8)
    assert (bits(img) == bits(want)).all()


@pytest.mark.parametrize("kernel", ["persistent", "persistent-batched", "static", "wavefront"])
def test_every_cast_kernel_gives_the_same_bits(oracle, gpu_tracer, kernel):
    """The three renderer implementations against the oracle: full frames, partial tiles, both shard
    modes, a scene with refraction/metal (spheres) and one with instances."""
    import torch

    T.set_cast_kernel(gpu_tracer, kernel.split("-")[0])
    if kernel == "persistent-batched":
        T.set_persistent_params(gpu_tracer, refill=24, inst_batch=6)
    try:
        verts, indices = scenes.bunny_class_mesh(3)
        cases = [(scenes.cornell(), 8, (70, 50, 2, 6)), (scenes.spheres(), 8, (64, 40, 2, 3)),
                 (scenes.instanced_grid(10, seed=2, verts=verts, indices=indices), 4, (90, 60, 1, 8))]
        for (w, cam, sky), bounces, (wd, ht, smp, seed) in cases:
            build(gpu_tracer, w, bounces, sky)
            sc = oracle.Scene(world.flatten_world(w), bounces, sky)
            c = cam(wd, ht, smp)
            img = np.zeros((ht, wd, 3), dtype=np.float32)
            T.rt_tracer_cast(gpu_tracer, c, img, wd, ht, seed=seed)
            want, _ = sc.cast(c, seed, wd, ht, rng_mode=oracle.RNG_COUNTER, threads=8)
            assert (bits(img) == bits(want)).all(), f"{kernel}: frame differs"
            for shard in ((1, 1, 2), (2, 2, 3)):
                buf = torch.full((ht, wd, 3), 7.0, dtype=torch.float32, device="cuda:0")
                sh = np.zeros(1, dtype=abi.SHARD)
                sh[0] = shard
                rc = gpu_tracer.lib.rthip_cast_device(gpu_tracer.h, abi.ptr(c), seed, buf.data_ptr(), wd, ht, abi.ptr(sh),
                                                      torch.cuda.current_stream().cuda_stream)
                gpu_tracer.check(rc, "rthip_cast_device")
                torch.cuda.synchronize()
                want, _ = sc.cast(c, seed, wd, ht, rng_mode=oracle.RNG_COUNTER, shard=shard, threads=8)
                assert (bits(buf.cpu().numpy()) == bits(want)).all(), f"{kernel}: shard {shard} differs"
        # shard mode 3, bands of whole rows (150 rows in 2 bands of 128): a band's launch leaves the other rows alone, so
        # the bands of one frame can be rendered into one buffer in any order
        w, cam, sky = scenes.cornell()
        build(gpu_tracer, w, 8, sky)
        wd, ht = 70, 150
        c = cam(wd, ht, 2)
        want, _ = oracle.Scene(world.flatten_world(w), 8, sky).cast(c, 9, wd, ht, rng_mode=oracle.RNG_COUNTER, threads=8)
        buf = torch.full((ht, wd, 3), 7.0, dtype=torch.float32, device="cuda:0")
        for rank in (1, 0):
            sh = np.zeros(1, dtype=abi.SHARD)
            sh[0] = (3, rank, 2)
            rc = gpu_tracer.lib.rthip_cast_device(gpu_tracer.h, abi.ptr(c), 9, buf.data_ptr(), wd, ht, abi.ptr(sh),
                                                  torch.cuda.current_stream().cuda_stream)
            gpu_tracer.check(rc, "rthip_cast_device")
            torch.cuda.synchronize()
            got = buf.cpu().numpy()
            assert (bits(got[128:]) == bits(want[128:])).all(), f"{kernel}: band 1 differs"
            if rank == 1:
                assert (got[:128] == 7.0).all(), f"{kernel}: band 1 wrote into band 0's rows"
            else:
                assert (bits(got) == bits(want)).all(), f"{kernel}: band 0 differs"
        # max_bounces 0: black frame (raytracer_cpu.c:247-249)
        w, cam, sky = scenes.cornell()
        build(gpu_tracer, w, 0, sky)
        img = np.ones((16, 16, 3), dtype=np.float32)
        T.rt_tracer_cast(gpu_tracer, cam(16, 16, 1), img, 16, 16, seed=1)
        assert not img.any()
    finally:
        T.set_cast_kernel(gpu_tracer, "auto")
        T.set_persistent_params(gpu_tracer)


@pytest.mark.parametrize("samples,wd,ht", [(2, 70, 50), (4, 37, 23), (8, 19, 13), (3, 33, 20)])
def test_sample_per_lane_kernel_gives_the_same_bits(oracle, gpu_tracer, samples, wd, ht):
    """RTHIP_KERNEL_STATIC_SAMPLES: the samples of a pixel in neighbouring lanes, added in sample order.  Frames
    with ragged edges, both shard modes; samples = 3 has no lane layout and must run the pixel-per-thread kernel."""
    import torch

    T.set_cast_kernel(gpu_tracer, "static-samples")
    try:
        verts, indices = scenes.bunny_class_mesh(3)
        for (w, cam, sky), bounces in ((scenes.cornell(), 8), (scenes.spheres(), 6),
                                        (scenes.instanced_grid(6, seed=2, verts=verts, indices=indices), 3)):
            build(gpu_tracer, w, bounces, sky)
            sc = oracle.Scene(world.flatten_world(w), bounces, sky)
            c = cam(wd, ht, samples)
            img = np.zeros((ht, wd, 3), dtype=np.float32)
            T.rt_tracer_cast(gpu_tracer, c, img, wd, ht, seed=5)
            assert T.get_stats(gpu_tracer)["last_cast_kernel"] == (4 if samples in (2, 4, 8) else 1)
            want, _ = sc.cast(c, 5, wd, ht, rng_mode=oracle.RNG_COUNTER, threads=8)
            assert (bits(img) == bits(want)).all(), "frame differs"
            for shard in ((1, 1, 2), (2, 2, 3), (2, 0, 5)):
                buf = torch.full((ht, wd, 3), 7.0, dtype=torch.float32, device="cuda:0")
                sh = np.zeros(1, dtype=abi.SHARD)
                sh[0] = shard
                rc = gpu_tracer.lib.rthip_cast_device(gpu_tracer.h, abi.ptr(c), 5, buf.data_ptr(), wd, ht, abi.ptr(sh),
                                                      torch.cuda.current_stream().cuda_stream)
                gpu_tracer.check(rc, "rthip_cast_device")
                torch.cuda.synchronize()
                want, _ = sc.cast(c, 5, wd, ht, rng_mode=oracle.RNG_COUNTER, shard=shard, threads=8)
                assert (bits(buf.cpu().numpy()) == bits(want)).all(), f"shard {shard} differs"
        # max_bounces 0: black frame
        w, cam, sky = scenes.cornell()
        build(gpu_tracer, w, 0, sky)
        img = np.ones((16, 16, 3), dtype=np.float32)
        T.rt_tracer_cast(gpu_tracer, cam(16, 16, 2), img, 16, 16, seed=1)
        assert not img.any()
        # AUTO takes the same kernel for a small frame of a shallow scene with 4 samples per pixel
        T.set_cast_kernel(gpu_tracer, "auto")
        build(gpu_tracer, w, 4, sky)
        img = np.zeros((40, 40, 3), dtype=np.float32)
        T.rt_tracer_cast(gpu_tracer, cam(40, 40, 2), img, 40, 40, seed=2)
        assert T.get_stats(gpu_tracer)["last_cast_kernel"] == 4
        T.set_cast_kernel(gpu_tracer, "static")
        ref = np.zeros((40, 40, 3), dtype=np.float32)
        T.rt_tracer_cast(gpu_tracer, cam(40, 40, 2), ref, 40, 40, seed=2)
        assert (bits(img) == bits(ref)).all()
    finally:
        T.set_cast_kernel(gpu_tracer, "auto")


@pytest.mark.parametrize("kernel", ["persistent", "static", "static-samples"])
def test_casts_on_several_streams_overlap_without_changing_a_bit(gpu_tracer, kernel):
    """Frames enqueued on different streams of one context run concurrently; each must come out as it does alone."""
    import torch

    verts, indices = scenes.bunny_class_mesh(3)
    w, cam, sky = scenes.instanced_grid(8, seed=4, verts=verts, indices=indices)
    build(gpu_tracer, w, 4, sky)
    wd, ht = 320, 200
    c = cam(wd, ht, 2)
    T.set_cast_kernel(gpu_tracer, kernel)
    try:
        def cast(seed, buf, stream):
            gpu_tracer.check(gpu_tracer.lib.rthip_cast_device(gpu_tracer.h, abi.ptr(c), seed, buf.data_ptr(), wd, ht, None,
                                                              stream.cuda_stream), "rthip_cast_device")

        n = 24
        alone = []
        for k in range(n):
            buf = torch.zeros((ht, wd, 3), dtype=torch.float32, device="cuda:0")
            cast(100 + k, buf, torch.cuda.current_stream())
            torch.cuda.synchronize()
            alone.append(buf)
        streams = [torch.cuda.Stream() for _ in range(3)]
        bufs = [torch.full((ht, wd, 3), -1.0, dtype=torch.float32, device="cuda:0") for _ in range(n)]
        torch.cuda.synchronize()
        for k in range(n):
            cast(100 + k, bufs[k], streams[k % 3])
        torch.cuda.synchronize()
        for k in range(n):
            assert torch.equal(bufs[k].view(torch.int32), alone[k].view(torch.int32)), f"frame {k}"
    finally:
        T.set_cast_kernel(gpu_tracer, "auto")


@pytest.mark.parametrize("kernel", ["persistent", "static", "static-samples", "wavefront", "auto"])
def test_accumulating_cast_adds_the_frame(gpu_tracer, kernel):
    """rthip_cast_accumulate_device: buffer + frame, pixel by pixel in binary32, for every kernel; under a tile shard the
    other ranks' pixels keep their value; with the probe of the automatic choice (a frame of >= 2^20 paths)."""
    import torch

    verts, indices = scenes.bunny_class_mesh(3)
    w, cam, sky = scenes.cornell(extra_mesh=(verts, indices))
    build(gpu_tracer, w, 4, sky)
    stream = torch.cuda.current_stream().cuda_stream
    T.set_cast_kernel(gpu_tracer, kernel)
    try:
        for (wd, ht, smp, shard) in ((90, 70, 2, None), (130, 64, 1, (1, 1, 2)), (40, 40, 2, (2, 0, 3))) + \
                (((1100, 1000, 1, None),) if kernel == "auto" else ()):
            c = cam(wd, ht, smp)
            sh = None
            if shard:
                sh = np.zeros(1, dtype=abi.SHARD)
                sh[0] = shard
            base = torch.rand((ht, wd, 3), dtype=torch.float32, device="cuda:0") * 3.0
            want = base.clone()
            acc = base.clone()
            for seed in (5, 6, 7):
                frame = torch.zeros((ht, wd, 3), dtype=torch.float32, device="cuda:0")
                gpu_tracer.check(gpu_tracer.lib.rthip_cast_device(gpu_tracer.h, abi.ptr(c), seed, frame.data_ptr(), wd, ht,
                                                                  abi.ptr(sh) if shard else None, stream), "rthip_cast_device")
                want += frame
                gpu_tracer.check(gpu_tracer.lib.rthip_cast_accumulate_device(gpu_tracer.h, abi.ptr(c), seed, acc.data_ptr(), wd, ht,
                                                                             abi.ptr(sh) if shard else None, stream),
                                 "rthip_cast_accumulate_device")
            torch.cuda.synchronize()
            assert torch.equal(acc.view(torch.int32), want.view(torch.int32)), (kernel, wd, ht, shard)
            if shard and shard[0] == 1:  # pixels of the other rank's 64x64 tiles are untouched
                assert torch.equal(acc[:, 0:64].view(torch.int32), base[:, 0:64].view(torch.int32))
    finally:
        T.set_cast_kernel(gpu_tracer, "auto")


@pytest.mark.parametrize("kernel", ["static", "persistent", "auto"])
def test_casts_can_be_captured_in_a_hip_graph(gpu_tracer, kernel):
    """rthip_cast_device only enqueues work, so a stream capture can record it (the automatic kernel choice skips its
    probe while capturing); the replayed graph must write the same frames."""
    import torch

    verts, indices = scenes.bunny_class_mesh(3)
    w, cam, sky = scenes.cornell(extra_mesh=(verts, indices))
    build(gpu_tracer, w, 4, sky)
    wd, ht = 1100, 1000  # >= 2^20 paths: a frame for which the automatic choice would probe outside a capture
    c = cam(wd, ht, 1)
    T.set_cast_kernel(gpu_tracer, kernel)
    try:
        bufs = [torch.zeros((ht, wd, 3), dtype=torch.float32, device="cuda:0") for _ in range(3)]
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            for k, b in enumerate(bufs):
                gpu_tracer.check(gpu_tracer.lib.rthip_cast_device(gpu_tracer.h, abi.ptr(c), 40 + k, b.data_ptr(), wd, ht, None,
                                                                  torch.cuda.current_stream().cuda_stream), "rthip_cast_device")
        for b in bufs:
            b.fill_(-1.0)
        graph.replay()
        torch.cuda.synchronize()
        first = [b.clone() for b in bufs]
        for b in bufs:
            b.fill_(-2.0)
        graph.replay()
        torch.cuda.synchronize()
        T.set_cast_kernel(gpu_tracer, "static")
        for k, b in enumerate(bufs):
            want = torch.zeros((ht, wd, 3), dtype=torch.float32, device="cuda:0")
            gpu_tracer.check(gpu_tracer.lib.rthip_cast_device(gpu_tracer.h, abi.ptr(c), 40 + k, want.data_ptr(), wd, ht, None,
                                                              torch.cuda.current_stream().cuda_stream), "rthip_cast_device")
            torch.cuda.synchronize()
            assert torch.equal(b.view(torch.int32), want.view(torch.int32)), f"replay 2, frame {k}"
            assert torch.equal(first[k].view(torch.int32), want.view(torch.int32)), f"replay 1, frame {k}"
    finally:
        T.set_cast_kernel(gpu_tracer, "auto")


def test_persistent_launches_never_share_work_counters(gpu_tracer):
    """Every launch of the persistent kernel owns a slot of work counters until it has finished.  One long frame on one
    stream while another stream runs through more short launches than there are slots, a captured graph replayed beside
    eager casts, and more unfinished launches than slots (an error, not a silent collision)."""
    import torch

    verts, indices = scenes.bunny_class_mesh(4)
    w, cam, sky = scenes.instanced_grid(24, seed=1, verts=verts, indices=indices)
    build(gpu_tracer, w, 6, sky)
    lib, h = gpu_tracer.lib, gpu_tracer.h
    big_c, small_c = cam(1600, 1200, 2), cam(40, 24, 1)

    def cast(c, wd, ht, seed, buf, stream):
        gpu_tracer.check(lib.rthip_cast_device(h, abi.ptr(c), seed, buf.data_ptr(), wd, ht, None, stream.cuda_stream), "rthip_cast_device")

    T.set_cast_kernel(gpu_tracer, "static")
    main = torch.cuda.current_stream()
    want_big = torch.zeros((1200, 1600, 3), dtype=torch.float32, device="cuda")
    cast(big_c, 1600, 1200, 5, want_big, main)
    want_small = [torch.zeros((24, 40, 3), dtype=torch.float32, device="cuda") for _ in range(150)]
    for k, b in enumerate(want_small):
        cast(small_c, 40, 24, 100 + k, b, main)
    torch.cuda.synchronize()
    T.set_cast_kernel(gpu_tracer, "persistent")
    try:
        a, b_stream, g_stream = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
        # a graph of two persistent launches, captured first
        gbufs = [torch.zeros((24, 40, 3), dtype=torch.float32, device="cuda") for _ in range(2)]
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=g_stream):
            for k, gb in enumerate(gbufs):
                cast(small_c, 40, 24, 100 + k, gb, torch.cuda.current_stream())
        got_big = torch.full((1200, 1600, 3), -1.0, dtype=torch.float32, device="cuda")
        got_small = [torch.full((24, 40, 3), -1.0, dtype=torch.float32, device="cuda") for _ in range(150)]
        cast(big_c, 1600, 1200, 5, got_big, a)  # long: still running while the short ones go round the ring twice
        for k, sb in enumerate(got_small):
            cast(small_c, 40, 24, 100 + k, sb, b_stream)
            if k % 10 == 0:
                with torch.cuda.stream(g_stream):
                    graph.replay()
            if k % 50 == 49:
                b_stream.synchronize()  # short launches finish and free their slots; the long one keeps its own
        torch.cuda.synchronize()
        assert torch.equal(got_big.view(torch.int32), want_big.view(torch.int32)), "the long frame lost pixels"
        for k in range(150):
            assert torch.equal(got_small[k].view(torch.int32), want_small[k].view(torch.int32)), f"short frame {k}"
        for k in range(2):
            assert torch.equal(gbufs[k].view(torch.int32), want_small[k].view(torch.int32)), f"graph frame {k}"
        # more unfinished launches than slots: the call says so
        blocker = torch.cuda.Stream()
        with torch.cuda.stream(blocker):
            torch.cuda._sleep(300_000_000)  # keeps everything behind it on this stream unfinished for a while
        with pytest.raises(T.RTHipError, match="unfinished"):
            for k in range(70):
                cast(small_c, 40, 24, 100 + k, got_small[k], blocker)
        torch.cuda.synchronize()
        cast(small_c, 40, 24, 100, got_small[0], main)  # and the context works on afterwards
        torch.cuda.synchronize()
        assert torch.equal(got_small[0].view(torch.int32), want_small[0].view(torch.int32))
    finally:
        T.set_cast_kernel(gpu_tracer, "auto")


def test_cornell_with_mesh_scene(oracle, gpu_tracer):
    """The multi-GPU configuration's scene (Cornell box + a metal PN mesh): trees, radiance and a tile shard."""
    verts, indices = scenes.bunny_class_mesh(3)
    w, cam, sky = scenes.cornell(extra_mesh=(verts, indices))
    build(gpu_tracer, w, 8, sky)
    flat = world.flatten_world(w)
    sc = oracle.Scene(flat, 8, sky)
    assert_same_tree(T.get_tree(gpu_tracer, -1), sc.tree(-1), "TLAS")
    for k in range(len(flat.meshes)):
        assert_same_tree(T.get_tree(gpu_tracer, k), sc.tree(k), f"BLAS {k}")
    wd, ht, smp = 96, 54, 2
    c = cam(wd, ht, smp)
    img = np.zeros((ht, wd, 3), dtype=np.float32)
    T.rt_tracer_cast(gpu_tracer, c, img, wd, ht, seed=11)
    want, _ = sc.cast(c, 11, wd, ht, rng_mode=oracle.RNG_COUNTER, threads=8)
    assert (bits(img) == bits(want)).all()
    assert float(img.mean()) > 0.01  # the light is seen
    # the configuration's 1024 samples per pixel (samples = 32) on a handful of pixels, whole and as a sample shard
    wd, ht, smp = 8, 6, 32
    c = cam(wd, ht, smp)
    want, _ = sc.cast(c, 12, wd, ht, rng_mode=oracle.RNG_COUNTER, threads=8)
    try:
        for kernel in ("auto", "static", "persistent"):
            T.set_cast_kernel(gpu_tracer, kernel)
            img = np.zeros((ht, wd, 3), dtype=np.float32)
            T.rt_tracer_cast(gpu_tracer, c, img, wd, ht, seed=12)
            assert (bits(img) == bits(want)).all(), kernel
    finally:
        T.set_cast_kernel(gpu_tracer, "auto")
    import torch
    part, _ = sc.cast(c, 12, wd, ht, rng_mode=oracle.RNG_COUNTER, shard=(2, 5, 8), threads=8)
    buf = torch.zeros((ht, wd, 3), dtype=torch.float32, device="cuda:0")
    sh = np.zeros(1, dtype=abi.SHARD)
    sh[0] = (2, 5, 8)
    gpu_tracer.check(gpu_tracer.lib.rthip_cast_device(gpu_tracer.h, abi.ptr(c), 12, buf.data_ptr(), wd, ht, abi.ptr(sh),
                                                      torch.cuda.current_stream().cuda_stream), "rthip_cast_device")
    torch.cuda.synchronize()
    assert (bits(buf.cpu().numpy()) == bits(part)).all(), "sample shard 5 of 8 at 1024 spp"


def test_auto_kernel_choice_changes_no_bit(gpu_tracer):
    """A big frame makes RTHIP_KERNEL_AUTO render a counted probe on a sparse sample of tiles first
    (into the same framebuffer); the frame must come out exactly as from a fixed kernel, also from the
    cached choice, and also under a tile shard."""
    import torch

    verts, indices = scenes.bunny_class_mesh(3)
    w, cam, sky = scenes.cornell(extra_mesh=(verts, indices))
    build(gpu_tracer, w, 8, sky)
    wd = ht = 2048
    c = cam(wd, ht, 4)
    stream = torch.cuda.current_stream().cuda_stream

    def render(shard=None):
        buf = torch.full((ht, wd, 3), 3.0, dtype=torch.float32, device="cuda:0")
        sh = None
        if shard:
            sh = np.zeros(1, dtype=abi.SHARD)
            sh[0] = shard
        gpu_tracer.check(gpu_tracer.lib.rthip_cast_device(gpu_tracer.h, abi.ptr(c), 21, buf.data_ptr(), wd, ht,
                                                          abi.ptr(sh) if shard else None, stream), "rthip_cast_device")
        torch.cuda.synchronize()
        return buf

    try:
        T.set_cast_kernel(gpu_tracer, "static")
        want = render()
        want_shard = render((1, 1, 3))
        T.set_cast_kernel(gpu_tracer, "auto")
        first = render()
        st = T.get_stats(gpu_tracer)
        assert st["last_cast_kernel"] in (0, 1)
        assert torch.equal(first.view(torch.int32), want.view(torch.int32)), "frame after the probe differs"
        assert torch.equal(render().view(torch.int32), want.view(torch.int32)), "frame from the cached choice differs"
        assert torch.equal(render((1, 1, 3)).view(torch.int32), want_shard.view(torch.int32)), "tile shard differs"
    finally:
        T.set_cast_kernel(gpu_tracer, "auto")


def test_edge_case_rays(oracle, gpu_tracer):
    """Axis-parallel rays (1/0, 0 * inf), origins on box planes / edges / vertices, degenerate, coincident and
    sliver triangles, an empty TLAS: the reference's answers, bit for bit."""
    g = golden("edge_rays")
    for name, w in (("edge", scenes.edge_scene()), ("cornell", scenes.cornell()[0])):
        build(gpu_tracer, w, 4, True)
        inst, tri, t, uv, nrm = T.trace_rays(gpu_tracer, g[f"{name}_o"], g[f"{name}_d"])
        assert (inst == g[f"{name}_ref_inst"]).all() and (tri == g[f"{name}_ref_tri"]).all(), name
        assert (bits(t) == bits(g[f"{name}_ref_t"])).all() and (bits(uv) == bits(g[f"{name}_ref_uv"])).all(), name
        assert (bits(nrm) == bits(g[f"{name}_ref_normal"])).all(), name
        occ = T.occluded(gpu_tracer, g[f"{name}_o"], g[f"{name}_d"])
        assert (occ == (g[f"{name}_ref_inst"] > 0)).all(), name
    # a world with meshes but no instance: everything misses, the frame is the sky
    w = scenes.edge_scene()
    w.instances.clear()
    build(gpu_tracer, w, 4, True)
    inst, tri, t, uv, nrm = T.trace_rays(gpu_tracer, g["edge_o"], g["edge_d"])
    assert not inst.any() and (t == -1.0).all()
    c = scenes.get_rt_cast_settings(32, 24, 2, (0, 0, 5), (0, 0, 0), vfov=scenes.degrees_to_rad(40))
    img = np.zeros((24, 32, 3), dtype=np.float32)
    T.rt_tracer_cast(gpu_tracer, c, img, 32, 24, seed=1)
    want, _ = oracle.Scene(world.flatten_world(w), 4, True).cast(c, 1, 32, 24, rng_mode=oracle.RNG_COUNTER)
    assert (bits(img) == bits(want)).all() and float(img.min()) > 0.3


def test_full_size_frame_agrees_across_kernels_and_shards(oracle, gpu_tracer):
    """The instanced benchmark workload at full size (10 000 instances, 1920x1080, 4 spp, 8 bounces): the whole frame is
    too big for the CPU oracle, so the three renderer kernels must agree with each other bit for bit, three tile shards
    must add up to the frame, and four of its 64x64 tiles are compared with the oracle."""
    import torch

    verts, indices = scenes.bunny_class_mesh()
    w, cam, sky = scenes.instanced_grid(100, seed=1, verts=verts, indices=indices)
    build(gpu_tracer, w, 8, sky)
    wd, ht = 1920, 1080
    c = cam(wd, ht, 2)
    stream = torch.cuda.current_stream().cuda_stream

    def render(shard=None):
        buf = torch.full((ht, wd, 3), -1.0, dtype=torch.float32, device="cuda:0")
        sh = None
        if shard:
            sh = np.zeros(1, dtype=abi.SHARD)
            sh[0] = shard
        gpu_tracer.check(gpu_tracer.lib.rthip_cast_device(gpu_tracer.h, abi.ptr(c), 7, buf.data_ptr(), wd, ht,
                                                          abi.ptr(sh) if shard else None, stream), "rthip_cast_device")
        torch.cuda.synchronize()
        return buf

    try:
        frames = {}
        for kernel in ("static", "persistent", "wavefront"):
            T.set_cast_kernel(gpu_tracer, kernel)
            frames[kernel] = render()
        assert float(frames["static"].min()) >= 0.0 and float(frames["static"].mean()) > 0.05
        for kernel in ("persistent", "wavefront"):
            assert torch.equal(frames[kernel].view(torch.int32), frames["static"].view(torch.int32)), kernel
        T.set_cast_kernel(gpu_tracer, "auto")
        total = sum(render((1, r, 3)) for r in range(3))
        assert torch.equal((total + 0.0).view(torch.int32), (frames["static"] + 0.0).view(torch.int32)), "tile shards"
    finally:
        T.set_cast_kernel(gpu_tracer, "auto")
    # ... and the frame itself is pinned to the oracle tile by tile: the CPU renders single 64x64 tiles of the SAME
    # full-size frame (a tile shard of as many ranks as there are tiles), one among the instances, one across the
    # horizon, one of sky, one in the ragged bottom row
    sc = oracle.Scene(world.flatten_world(w), 8, sky)
    frame = frames["static"].cpu().numpy()
    tiles_x, tiles_y = (wd + 63) // 64, (ht + 63) // 64
    for name, tile in (("instances", (tiles_y * 2 // 3) * tiles_x + tiles_x // 2), ("horizon", (tiles_y // 2 - 1) * tiles_x + tiles_x // 3),
                       ("sky", tiles_x + 3), ("ragged bottom row", (tiles_y - 1) * tiles_x + tiles_x - 1)):
        assert_tile_matches_oracle(oracle, sc, c, 7, frame, tile, name)


def assert_tile_matches_oracle(oracle, sc, cast, seed, frame, tile, name):
    """One 64x64 tile of a full-size frame against the oracle's render of exactly that tile (bit for bit)."""
    ht, wd = frame.shape[0], frame.shape[1]
    tiles_x, tiles_y = (wd + 63) // 64, (ht + 63) // 64
    want, cnt = sc.cast(cast, seed, wd, ht, rng_mode=oracle.RNG_COUNTER, shard=(1, tile, tiles_x * tiles_y), threads=16)
    x0, y0 = (tile % tiles_x) * 64, (tile // tiles_x) * 64
    got_tile, want_tile = frame[y0:y0 + 64, x0:x0 + 64], want[y0:y0 + 64, x0:x0 + 64]
    assert got_tile.size > 0 and cnt["rays"] > 0, name
    bad = int((bits(got_tile) != bits(want_tile)).any(axis=2).sum())
    assert bad == 0, f"tile {tile} ({name}): {bad} of {got_tile.shape[0] * got_tile.shape[1]} pixels differ from the oracle"
    return float(want_tile.mean())


def test_config5_frame_is_pinned_to_the_oracle_tile_by_tile(oracle, gpu_tracer):
    """BASELINE config 5's frame -- Cornell box + the full bunny-class mesh at 3840x2160, 8 bounces -- whole on the GPU
    (64 spp), four of its 64x64 tiles against the oracle; and at the configuration's own 1024 spp one tile, rendered by
    the GPU as the tile shard a rank of a large group would get."""
    import torch

    verts, indices = scenes.bunny_class_mesh()
    w, cam, sky = scenes.cornell(extra_mesh=(verts, indices))
    build(gpu_tracer, w, 8, sky)
    sc = oracle.Scene(world.flatten_world(w), 8, sky)
    wd, ht = 3840, 2160
    tiles_x, tiles_y = (wd + 63) // 64, (ht + 63) // 64
    stream = torch.cuda.current_stream().cuda_stream
    c = cam(wd, ht, 8)
    buf = torch.full((ht, wd, 3), -1.0, dtype=torch.float32, device="cuda:0")
    gpu_tracer.check(gpu_tracer.lib.rthip_cast_device(gpu_tracer.h, abi.ptr(c), 3, buf.data_ptr(), wd, ht, None, stream), "rthip_cast_device")
    torch.cuda.synchronize()
    frame = buf.cpu().numpy()
    assert float(frame.min()) >= 0.0
    mid = (tiles_y // 2) * tiles_x + tiles_x // 2
    lit = 0.0
    for name, tile in (("centre: mesh and back wall", mid), ("mesh seen from outside, reflecting the opening", mid + tiles_x + 3), ("wall seam", (tiles_y - 8) * tiles_x + tiles_x // 2 + 4),
                       ("light", (tiles_y // 2 - 9) * tiles_x + tiles_x // 2), ("outside the box", 5 * tiles_x + 3)):
        lit += assert_tile_matches_oracle(oracle, sc, c, 3, frame, tile, name)
    assert lit > 0.01  # the chosen tiles do see the scene
    # 1024 spp: the tile alone, as a shard
    c = cam(wd, ht, 32)
    tile = (tiles_y - 8) * tiles_x + tiles_x // 2 + 4
    sh = np.zeros(1, dtype=abi.SHARD)
    sh[0] = (1, tile, tiles_x * tiles_y)
    buf.fill_(-1.0)
    gpu_tracer.check(gpu_tracer.lib.rthip_cast_device(gpu_tracer.h, abi.ptr(c), 1, buf.data_ptr(), wd, ht, abi.ptr(sh), stream), "rthip_cast_device")
    torch.cuda.synchronize()
    frame = buf.cpu().numpy()
    assert_tile_matches_oracle(oracle, sc, c, 1, frame, tile, "wall seam at 1024 spp")
    x0, y0 = (tile % tiles_x) * 64, (tile // tiles_x) * 64
    frame[y0:y0 + 64, x0:x0 + 64] = 0.0
    assert not frame.any(), "a tile shard leaves the other ranks' pixels at zero"


def test_long_ray_stream(oracle, gpu_tracer):
    """Ray streams of 2^18 rays and more go through the resident-wave kernel with lane refill: same
    answers and the same traversal counters as the oracle, with a ragged tail."""
    verts, indices = scenes.bunny_class_mesh(3)
    w, cam, sky = scenes.instanced_grid(12, seed=1, verts=verts, indices=indices)
    build(gpu_tracer, w, 4, sky)
    sc = oracle.Scene(world.flatten_world(w), 4, sky)
    n = (1 << 18) + 1234
    o, d = scenes.random_rays(n, seed=9, lo=(-15, -2, -15), hi=(15, 3, 15))
    T.set_counting(gpu_tracer, True)
    T.get_stats(gpu_tracer)
    inst, tri, t, uv, nrm = T.trace_rays(gpu_tracer, o, d)
    st = T.get_stats(gpu_tracer)
    T.set_counting(gpu_tracer, False)
    oi, ot, ott, ouv, onrm, cnt = sc.trace(o, d, threads=8)
    assert (inst == oi).all() and (tri == ot).all() and (bits(t) == bits(ott)).all()
    assert (bits(uv) == bits(ouv)).all() and (bits(nrm) == bits(onrm)).all()
    for k in ("rays", "nodes_tested", "tris_tested", "inst_entered"):
        assert st[k] == cnt[k], f"traversal counter {k}: {st[k]} vs {cnt[k]}"
    inst2 = T.trace_rays(gpu_tracer, o, d)[0]  # counting off: the other template instance
    assert (inst2 == oi).all()


def test_soup_scene_rays_and_frames(oracle, gpu_tracer):
    """A 200 000-triangle soup (one deep BLAS, the benchmark's 10 M-triangle scene in small): incoherent rays with
    their traversal counters and a path-traced frame from every kernel against the oracle."""
    w, cam, sky = scenes.soup_world(200_000, seed=1)
    build(gpu_tracer, w, 4, sky)
    flat = world.flatten_world(w)
    sc = oracle.Scene(flat, 4, sky)
    assert_same_tree(T.get_tree(gpu_tracer, 0), sc.tree(0), "soup BLAS")
    o, d = scenes.random_rays(40_000, seed=2)
    T.set_counting(gpu_tracer, True)
    T.get_stats(gpu_tracer)
    inst, tri, t, uv, nrm = T.trace_rays(gpu_tracer, o, d)
    st = T.get_stats(gpu_tracer)
    T.set_counting(gpu_tracer, False)
    oi, ot, ott, ouv, onrm, cnt = sc.trace(o, d, threads=8)
    assert (inst == oi).all() and (tri == ot).all() and (bits(t) == bits(ott)).all()
    assert (bits(uv) == bits(ouv)).all() and (bits(nrm) == bits(onrm)).all()
    assert 100 < int((inst > 0).sum()) < 40_000
    for k in ("rays", "nodes_tested", "tris_tested", "inst_entered"):
        assert st[k] == cnt[k], f"traversal counter {k}: {st[k]} vs {cnt[k]}"
    assert (T.occluded(gpu_tracer, o, d, 0.25) == sc.occluded(o, d, 0.25)).all()
    c = cam(96, 54, 2)
    want, _ = sc.cast(c, 6, 96, 54, rng_mode=oracle.RNG_COUNTER, threads=8)
    try:
        for kernel in ("static", "persistent", "static-samples", "wavefront", "auto"):
            T.set_cast_kernel(gpu_tracer, kernel)
            img = np.zeros((54, 96, 3), dtype=np.float32)
            T.rt_tracer_cast(gpu_tracer, c, img, 96, 54, seed=6)
            assert (bits(img) == bits(want)).all(), kernel
    finally:
        T.set_cast_kernel(gpu_tracer, "auto")


def test_heightfield_scene(oracle, gpu_tracer):
    """The coherent variant of the 10 M-triangle configuration in small (a 150 x 150-cell displaced height field, indexed):
    tree, rays with counters, frames from every kernel, and the ordered walk, against the oracle."""
    w, cam, sky = scenes.heightfield_world(150, seed=1)
    build(gpu_tracer, w, 4, sky)
    flat = world.flatten_world(w)
    sc = oracle.Scene(flat, 4, sky)
    assert_same_tree(T.get_tree(gpu_tracer, 0), sc.tree(0), "height field BLAS")
    o, d = scenes.random_rays(30_000, seed=4, lo=(0.0, 0.35, 0.0), hi=(1.0, 0.6, 1.0))
    d[:, 1] = -np.abs(d[:, 1])  # downwards: most of them reach the surface
    T.set_counting(gpu_tracer, True)
    T.get_stats(gpu_tracer)
    inst, tri, t, uv, nrm = T.trace_rays(gpu_tracer, o, d)
    st = T.get_stats(gpu_tracer)
    T.set_counting(gpu_tracer, False)
    oi, ot, ott, ouv, onrm, cnt = sc.trace(o, d, threads=8)
    assert (inst == oi).all() and (tri == ot).all() and (bits(t) == bits(ott)).all()
    assert (bits(uv) == bits(ouv)).all() and (bits(nrm) == bits(onrm)).all() and int((inst > 0).sum()) > 10_000
    for k in ("rays", "nodes_tested", "tris_tested", "inst_entered"):
        assert st[k] == cnt[k], f"traversal counter {k}: {st[k]} vs {cnt[k]}"
    t_diff, prim_diff, hit_miss, _, _ = fast_vs_exact_rays(gpu_tracer, o, d)
    assert t_diff == 0 and hit_miss == 0 and prim_diff <= 30, (t_diff, prim_diff, hit_miss)  # shared edges: ties between neighbours
    c = cam(96, 54, 2)
    want, _ = sc.cast(c, 8, 96, 54, rng_mode=oracle.RNG_COUNTER, threads=8)
    try:
        for kernel in ("static", "persistent", "static-samples", "auto"):
            T.set_cast_kernel(gpu_tracer, kernel)
            img = np.zeros((54, 96, 3), dtype=np.float32)
            T.rt_tracer_cast(gpu_tracer, c, img, 96, 54, seed=8)
            assert (bits(img) == bits(want)).all(), kernel
    finally:
        T.set_cast_kernel(gpu_tracer, "auto")


def test_ten_million_triangle_soup_incoherent_rays(oracle, gpu_tracer):
    """The benchmark's 10 M-triangle scene and its ray stream (2^24 incoherent rays, seed 2) at full size: the plain
    ray kernel (the one such a scene takes by default under the exact walk) and the resident-wave kernel must agree on every ray bit for bit,
    and the first 2^16 rays -- hits, distances and traversal counters -- must be the oracle's on the same scene."""
    import torch

    w, cam, sky = scenes.soup_world(10_000_000, seed=1)
    build(gpu_tracer, w, 1, sky)
    flat = world.flatten_world(w)
    n = 1 << 24
    o, d = scenes.incoherent_rays_device(torch, n, 2)
    stream = torch.cuda.current_stream().cuda_stream

    def run(count, counting=False):
        hits = torch.full((count, 4), -1, dtype=torch.int32, device="cuda")
        if counting:
            T.set_counting(gpu_tracer, True)
            T.get_stats(gpu_tracer)
        gpu_tracer.check(gpu_tracer.lib.rthip_trace_rays_device(gpu_tracer.h, o.data_ptr(), d.data_ptr(), count, hits.data_ptr(),
                                                                stream), "rthip_trace_rays_device")
        torch.cuda.synchronize()
        st = None
        if counting:
            st = T.get_stats(gpu_tracer)
            T.set_counting(gpu_tracer, False)
        return hits, st

    try:
        T.set_cast_kernel(gpu_tracer, "auto")  # a scene of this size takes trace_rays_kernel
        plain, _ = run(n)
        T.set_cast_kernel(gpu_tracer, "persistent")  # forces trace_rays_persistent_kernel
        resident, _ = run(n)
        assert torch.equal(plain, resident), "plain and resident-wave ray kernels differ"
        assert bool((plain[:, 3] == 0).all())
        hit_share = float((plain[:, 0] != 0).float().mean())
        assert 0.5 < hit_share < 1.0, hit_share
        T.set_cast_kernel(gpu_tracer, "auto")
        m = 1 << 16
        head, st = run(m, counting=True)
        assert torch.equal(head, plain[:m])
        # the opt-in ordered walk on the same stream: the same distance bit for bit on every ray; another primitive only
        # where two lie at exactly that distance (the reference reports the last one it visits); fewer box tests
        T.set_traversal(gpu_tracer, "fast")
        fast, fst = run(n, counting=True)
        assert torch.equal(fast[:, 2], plain[:, 2]), f"{int((fast[:, 2] != plain[:, 2]).sum())} rays at another distance"
        other = int(((fast[:, 0] != plain[:, 0]) | (fast[:, 1] != plain[:, 1])).sum())
        assert other <= 8, f"{other} rays report another primitive"
        assert fst["nodes_tested"] < 0.85 * 651.87 * n
        # the ordered walk takes the resident-wave kernel at any scene size; the plain kernel gives the same answers
        T.set_cast_kernel(gpu_tracer, "static")
        fast_plain, _ = run(n)
        assert torch.equal(fast_plain, fast), "plain and resident-wave ray kernels differ under the ordered walk"
    finally:
        T.set_traversal(gpu_tracer, "exact")
        T.set_cast_kernel(gpu_tracer, "auto")
    sc = oracle.Scene(flat, 1, sky)
    oi, ot, ott, _, _, cnt = sc.trace(o[:m].cpu().numpy(), d[:m].cpu().numpy(), threads=8, want_normal=False)
    got = head.cpu().numpy().view(np.uint32)
    assert (got[:, 0] == oi).all() and (got[:, 1] == ot).all(), f"{int((got[:, 0] != oi).sum())} rays hit another instance"
    assert (got[:, 2] == bits(ott)).all(), "t bits differ"
    for k in ("rays", "nodes_tested", "tris_tested", "inst_entered"):
        assert st[k] == cnt[k], f"traversal counter {k}: {st[k]} vs {cnt[k]}"
    # the camera frame of the same scene (bench workload `soup` in small): static and persistent kernels against the oracle
    gpu_tracer.check(gpu_tracer.lib.rthip_set_tracer(gpu_tracer.h, 4, 1), "set_tracer")
    sc4 = oracle.Scene(flat, 4, sky)
    c = cam(64, 36, 1)
    want, _ = sc4.cast(c, 3, 64, 36, rng_mode=oracle.RNG_COUNTER, threads=8)
    try:
        for kernel in ("static", "persistent"):
            T.set_cast_kernel(gpu_tracer, kernel)
            img = np.zeros((36, 64, 3), dtype=np.float32)
            T.rt_tracer_cast(gpu_tracer, c, img, 64, 36, seed=3)
            assert (bits(img) == bits(want)).all(), kernel
    finally:
        T.set_cast_kernel(gpu_tracer, "auto")


def fast_vs_exact_rays(gpu_tracer, o, d):
    """(rays at another distance, rays with another primitive at the same distance, hit-versus-miss rays) between the
    two walks, and both result sets."""
    T.set_traversal(gpu_tracer, "exact")
    a = T.trace_rays(gpu_tracer, o, d)
    T.set_traversal(gpu_tracer, "fast")
    try:
        b = T.trace_rays(gpu_tracer, o, d)
    finally:
        T.set_traversal(gpu_tracer, "exact")
    t_differs = bits(a[2]) != bits(b[2])
    prim_differs = (a[0] != b[0]) | (a[1] != b[1])
    return int(t_differs.sum()), int((prim_differs & ~t_differs).sum()), int(((a[0] != 0) != (b[0] != 0)).sum()), a, b


def test_fast_traversal_against_the_exact_walk(oracle, gpu_tracer):
    """RTHIP_TRAV_FAST (4-wide records, nearest box first, stack in LDS) against the reference's walk.  Pinned: on the reference's own
    scenes every primary ray gets the same primitive at the same distance; on incoherent rays through deep scenes the
    distance is equal bit for bit and at most a handful of rays report another primitive at that same distance."""
    g = golden("demo_cornell")
    w, cam, sky = scenes.cornell()
    build(gpu_tracer, w, 8, sky)
    o, d = oracle.primary_rays(cam(512, 512, 1), 512, 512)
    assert fast_vs_exact_rays(gpu_tracer, o, d)[:3] == (0, 0, 0)
    T.set_traversal(gpu_tracer, "fast")
    try:
        inst, tri, t = T.primary_hits(gpu_tracer, cam(512, 512, 1), 512, 512)  # the reference's 512^2 answers, seam rays included
        assert hashlib.sha256(inst.tobytes()).hexdigest() == str(g["ref_512_inst_sha"])
        assert hashlib.sha256(tri.tobytes()).hexdigest() == str(g["ref_512_tri_sha"])
        assert hashlib.sha256(t.tobytes()).hexdigest() == str(g["ref_512_t_sha"])
        # whole frames: every kernel renders the fast walk's image, and on Cornell that is the exact walk's image
        c = cam(96, 80, 2)
        want, _ = oracle.Scene(world.flatten_world(w), 8, sky).cast(c, 4, 96, 80, rng_mode=oracle.RNG_COUNTER, threads=8)
        for kernel in ("static", "static-samples", "persistent", "wavefront", "auto"):
            T.set_cast_kernel(gpu_tracer, kernel)
            img = np.zeros((80, 96, 3), dtype=np.float32)
            T.rt_tracer_cast(gpu_tracer, c, img, 96, 80, seed=4)
            assert (bits(img) == bits(want)).all(), kernel
    finally:
        T.set_cast_kernel(gpu_tracer, "auto")
        T.set_traversal(gpu_tracer, "exact")
    for name in ("tri", "spheres"):
        w, cam, sky = DEMOS[name]()
        build(gpu_tracer, w, 8, sky)
        gd = golden("demo_" + name)
        assert fast_vs_exact_rays(gpu_tracer, gd["rays_o"], gd["rays_d"])[:3] == (0, 0, 0), name
    # bunny-class mesh, 1920x1080 primary rays: the reference's answers
    gb = golden("demo_bunny_class")
    verts, indices = scenes.bunny_class_mesh()
    w, cam, sky = scenes.bunny(verts, indices)
    build(gpu_tracer, w, 2, sky)
    T.set_traversal(gpu_tracer, "fast")
    try:
        inst, tri, t = T.primary_hits(gpu_tracer, cam(1920, 1080, 1), 1920, 1080)
        assert hashlib.sha256(inst.tobytes()).hexdigest() == str(gb["ref_1080_inst_sha"])
        assert hashlib.sha256(tri.tobytes()).hexdigest() == str(gb["ref_1080_tri_sha"])
        assert hashlib.sha256(t.tobytes()).hexdigest() == str(gb["ref_1080_t_sha"])
    finally:
        T.set_traversal(gpu_tracer, "exact")
    # instances (rotated, scaled) and a soup, incoherent rays, occlusion, animated TLAS rebuilds while in fast mode
    verts, indices = scenes.bunny_class_mesh(3)
    w, cam, sky = scenes.instanced_grid(12, seed=1, verts=verts, indices=indices)
    build(gpu_tracer, w, 4, sky)
    o, d = scenes.random_rays(200_000, seed=3, lo=(-15, -2, -15), hi=(15, 3, 15))
    t_diff, prim_diff, hit_miss, a, b = fast_vs_exact_rays(gpu_tracer, o, d)
    assert t_diff == 0 and hit_miss == 0 and prim_diff <= 2, (t_diff, prim_diff, hit_miss)
    same = (a[0] == b[0]) & (a[1] == b[1])
    assert (bits(a[3]) == bits(b[3]))[same].all() and (bits(a[4]) == bits(b[4]))[same].all()  # uv and normals
    occ = T.occluded(gpu_tracer, o, d, 3.0)
    T.set_traversal(gpu_tracer, "fast")
    try:
        assert int((T.occluded(gpu_tracer, o, d, 3.0) != occ).sum()) == 0
        T.rt_tracer_build_tlas(gpu_tracer, w)  # a rebuild in fast mode makes its records too
        assert (T.trace_rays(gpu_tracer, o, d)[0] == b[0]).all()
    finally:
        T.set_traversal(gpu_tracer, "exact")
    w, cam, sky = scenes.soup_world(200_000, seed=1)
    build(gpu_tracer, w, 4, sky)
    o, d = scenes.random_rays(100_000, seed=2)
    t_diff, prim_diff, hit_miss, a, b = fast_vs_exact_rays(gpu_tracer, o, d)
    assert t_diff == 0 and hit_miss == 0 and prim_diff <= 2, (t_diff, prim_diff, hit_miss)
    with pytest.raises(T.RTHipError):
        gpu_tracer.check(gpu_tracer.lib.rthip_set_traversal(gpu_tracer.h, 7), "rthip_set_traversal")


def deep_tree_triangles():
    """Sixty large triangles, each with a centroid coordinate half the previous one's: their radix tree is about sixty
    levels deep, and a ray along z pierces all of its boxes at once."""
    tris = []
    for j in range(1, 21):
        dlt = 0.5 ** j
        for axis in range(3):
            xmax = 3 + 2 * dlt if axis == 0 else 3.0
            ymax = 3 + 2 * dlt if axis == 1 else 3.0
            z = dlt if axis == 2 else 0.0
            tris.append([[-3, -3, z], [xmax, -3, z], [-3, ymax, z]])
    return np.array(tris, dtype=np.float32)


def tree_depth(nodes):
    depth, open_until = 0, []
    for i in range(nodes.shape[0]):
        while open_until and open_until[-1] <= i:
            open_until.pop()
        depth = max(depth, len(open_until))
        open_until.append(int(nodes["skip"][i]))
    return depth


def test_resident_kernels_enter_instances_in_batches_with_the_same_answers(oracle, monkeypatch):
    """The resident kernels park lanes at instances and enter them several at a time (the ordered walk:
    RTHIP_FAST_INST_BATCH, default 8; the exact walk's ray streams: RTHIP_RAYS_INST_BATCH, default 8 from 64 instances).
    Parking delays a lane's entry and changes nothing else about its walk: frames and ray streams are equal bit for bit
    whatever the batch, also with a batch no wave ever fills (64); the exact walk's stream equals the oracle's answers."""
    verts, indices = scenes.bunny_class_mesh(3)
    w, cam, sky = scenes.instanced_grid(24, seed=3, verts=verts, indices=indices)
    c = cam(320, 200, 2)
    rng = np.random.default_rng(5)
    n = 1 << 18  # long enough for the resident ray-stream kernel
    o = rng.uniform(-30.0, 30.0, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    got, exact = {}, {}
    for batch in ("0", "8", "3", "64"):
        monkeypatch.setenv("RTHIP_FAST_INST_BATCH", batch)
        monkeypatch.setenv("RTHIP_RAYS_INST_BATCH", batch)
        tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=4, sky=sky), device_id=0)
        try:
            T.rt_tracer_build_blas(tr, w)
            T.rt_tracer_build_tlas(tr, w)
            exact[batch] = T.trace_rays(tr, o, d)
            T.set_traversal(tr, "fast")
            T.set_cast_kernel(tr, "persistent")
            img = np.zeros((200, 320, 3), dtype=np.float32)
            T.rt_tracer_cast(tr, c, img, 320, 200, seed=9)
            got[batch] = (img, T.trace_rays(tr, o, d))
        finally:
            T.rt_tracer_cleanup(tr)
    assert int((got["0"][1][0] != 0).sum()) > n // 20, "the rays miss the grid: the test tests nothing"
    for batch in ("8", "3", "64"):
        assert (bits(got[batch][0]) == bits(got["0"][0])).all(), batch
        for a, b in zip(got[batch][1], got["0"][1]):
            assert (bits(a) == bits(b)).all() if a.dtype == np.float32 else (a == b).all(), batch
        for a, b in zip(exact[batch], exact["0"]):
            assert (bits(a) == bits(b)).all() if a.dtype == np.float32 else (a == b).all(), ("exact walk", batch)
    sc = oracle.Scene(world.flatten_world(w), 4, sky)
    k = 4096
    oi, otri, ot, ouv, onrm, _ = sc.trace(o[:k], d[:k], threads=8)
    gi, gtri, gt, guv, gnrm = exact["8"]
    assert (gi[:k] == oi).all() and (gtri[:k] == otri).all() and (bits(gt[:k]) == bits(ot)).all()
    assert (bits(guv[:k]) == bits(ouv)).all() and (bits(gnrm[:k]) == bits(onrm)).all()


@pytest.mark.parametrize("level", ["mesh", "tlas"])
def test_fast_traversal_on_a_tree_deeper_than_its_stack(gpu_tracer, level):
    """A tree about sixty levels deep whose boxes a ray pierces all at once wants more stack than the ordered walk keeps
    (16 entries).  As one mesh: the boxes that do not fit are gone through in the binary tree's stackless order inside
    the step.  As sixty one-triangle instances (the same tree as the TLAS): such a ray is answered by the stackless walk.
    Most of the triangles lie in one plane, so which of them is reported is a tie; the distance is not."""
    tris = deep_tree_triangles()
    w = world.rt_make_world()
    mat = world.rt_world_add_material(w)
    mat.type, mat.albedo = abi.MAT_LAMBERTIAN, scenes.v3(0.5, 0.5, 0.5)
    groups = [tris.reshape(180, 3)] if level == "mesh" else [t for t in tris]
    for g in groups:
        v = np.zeros((g.shape[0], 4), dtype=np.float32)
        v[:, :3] = g
        m = world.rt_world_add_mesh(w)
        scenes._set_mesh(m, v, None, scenes.GEO_P)
        it = world.rt_world_add_instance(w)
        it.type, it.material, it.mesh = abi.INST_MESH, mat, m
        it.translation, it.rotation, it.scale = scenes.v3(0, 0, 0), scenes.IDENTITY_QUAT, scenes.v3(1, 1, 1)
    build(gpu_tracer, w, 2, True)
    depth = tree_depth(T.get_tree(gpu_tracer, 0 if level == "mesh" else -1))
    assert depth >= 40, depth
    rng = np.random.default_rng(1)
    n = 4096
    o = np.concatenate([rng.uniform(-2.5, -1.5, (n, 2)), np.full((n, 1), 3.0)], axis=1).astype(np.float32)
    d = np.tile(np.array([[0.0, 0.0, -1.0]], dtype=np.float32), (n, 1))
    d[n // 2:, 2] = 1.0  # half of them from behind
    o[n // 2:, 2] = -3.0
    t_diff, prim_diff, hit_miss, a, b = fast_vs_exact_rays(gpu_tracer, o, d)
    assert (t_diff, hit_miss) == (0, 0) and int((a[0] != 0).sum()) == n
    assert np.allclose(a[2][: n // 2], 2.5, rtol=1e-6) and np.allclose(a[2][n // 2:], 3.0, rtol=1e-6)
    occ = T.occluded(gpu_tracer, o, d, 2.75)
    T.set_traversal(gpu_tracer, "fast")
    try:
        assert (T.occluded(gpu_tracer, o, d, 2.75) == occ).all() and int(occ.sum()) == n // 2
    finally:
        T.set_traversal(gpu_tracer, "exact")


def test_fast_traversal_with_a_long_bounce_stack(oracle, gpu_tracer):
    """40 bounces: the bounce stack (40 KB) and the ordered walk's stacks (32 KB) together pass the 64 KB of dynamic LDS
    a launch gets without asking; every kernel must still start, and the Cornell frame is the exact walk's."""
    w, cam, sky = scenes.cornell()
    build(gpu_tracer, w, 40, sky)
    c = cam(48, 40, 2)
    want, _ = oracle.Scene(world.flatten_world(w), 40, sky).cast(c, 6, 48, 40, rng_mode=oracle.RNG_COUNTER, threads=8)
    T.set_traversal(gpu_tracer, "fast")
    try:
        for kernel in ("static", "static-samples", "persistent"):
            T.set_cast_kernel(gpu_tracer, kernel)
            img = np.zeros((40, 48, 3), dtype=np.float32)
            T.rt_tracer_cast(gpu_tracer, c, img, 48, 40, seed=6)
            assert (bits(img) == bits(want)).all(), kernel
    finally:
        T.set_cast_kernel(gpu_tracer, "auto")
        T.set_traversal(gpu_tracer, "exact")


def test_occlusion_rays(oracle, gpu_tracer):
    """Shadow rays (any-hit walk) against the reference's interval-limited query, and against the
    oracle on an instanced scene with random rays and several tmax."""
    g = golden("occlusion")
    verts, indices = scenes.bunny_class_mesh()
    for name, (w, cam, sky), bounces in (("cornell", scenes.cornell(), 8), ("bunny", scenes.bunny(verts, indices), 2)):
        build(gpu_tracer, w, bounces, sky)
        for k in range(3):
            got = T.occluded(gpu_tracer, g[f"{name}_o"], g[f"{name}_d"], float(g[f"{name}_tmax{k}"]))
            assert (got == g[f"{name}_ref_occluded{k}"]).all(), f"{name} tmax {k}"
    verts, indices = scenes.bunny_class_mesh(3)
    w, cam, sky = scenes.instanced_grid(12, seed=1, verts=verts, indices=indices)
    build(gpu_tracer, w, 4, sky)
    sc = oracle.Scene(world.flatten_world(w), 4, sky)
    o, d = scenes.random_rays(30_000, seed=5, lo=(-15, -2, -15), hi=(15, 3, 15))
    for tmax in (0.5, 3.0, 40.0, 3.402823466e+38):
        got = T.occluded(gpu_tracer, o, d, tmax)
        want = sc.occluded(o, d, tmax)
        assert (got == want).all(), f"instanced tmax {tmax}: {(got != want).sum()} rays differ"
    assert 0 < int(T.occluded(gpu_tracer, o, d, 3.0).sum()) < int(T.occluded(gpu_tracer, o, d, 40.0).sum())
    assert T.occluded(gpu_tracer, np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32)).size == 0


def test_cast_into_page_locked_host_memory(gpu_tracer):
    """A framebuffer from rthip_host_alloc changes how the frame reaches the host, not a bit of it."""
    w, cam, sky = scenes.cornell()
    build(gpu_tracer, w, 8, sky)
    c = cam(200, 120, 2)
    plain = np.zeros((120, 200, 3), dtype=np.float32)
    T.rt_tracer_cast(gpu_tracer, c, plain, 200, 120, seed=3)
    big = T.host_alloc(gpu_tracer, (4, 120, 200, 3))
    big[:] = -1.0
    for k in (2, 0):
        T.rt_tracer_cast(gpu_tracer, c, big[k], 200, 120, seed=3)
        assert (bits(big[k]) == bits(plain)).all()
    assert float(big[1].max()) == -1.0 and float(big[3].max()) == -1.0
    T.host_free(gpu_tracer, big)
    with pytest.raises(T.RTHipError):
        gpu_tracer.check(gpu_tracer.lib.rthip_host_free(gpu_tracer.h, plain.ctypes.data), "rthip_host_free")
    # Frames of 2^19 pixels and more travel in bands of rows (rthip_shard mode 3), every band rendered on a stream of
    # its own with its copy behind it: every kernel, a ragged last band, and a height that leaves the last band empty.
    verts, indices = scenes.bunny_class_mesh(3)
    w, cam, sky = scenes.bunny(verts, indices)
    build(gpu_tracer, w, 2, sky)
    try:
        for wd, ht, samples in ((1024, 600, 1), (2100, 260, 1), (1024, 520, 2)):
            c = cam(wd, ht, samples)
            plain = np.zeros((ht, wd, 3), dtype=np.float32)
            T.set_cast_kernel(gpu_tracer, "auto")
            T.rt_tracer_cast(gpu_tracer, c, plain, wd, ht, seed=5)
            assert float(plain.max()) > 0.0
            pinned = T.host_alloc(gpu_tracer, (ht, wd, 3))
            for kernel in ("auto", "static", "persistent", "wavefront") + (("static-samples",) if samples == 2 else ()):
                T.set_cast_kernel(gpu_tracer, kernel)
                pinned[:] = -1.0
                T.rt_tracer_cast(gpu_tracer, c, pinned, wd, ht, seed=5)
                assert (bits(pinned) == bits(plain)).all(), (wd, ht, kernel)
            T.host_free(gpu_tracer, pinned)
    finally:
        T.set_cast_kernel(gpu_tracer, "auto")


def test_host_buffer_casts_of_changing_sizes_and_contexts(gpu_tracer):
    """rthip_cast into pageable memory goes through page-locked staging and a pool of host threads that the context
    keeps: frames that grow and shrink (the staging is regrown), small frames in between (one launch, one copy), pageable
    and page-locked buffers in turn, and contexts that are destroyed with their pool; every frame against the same
    frame rendered into device memory."""
    import torch

    verts, indices = scenes.bunny_class_mesh(3)
    w, cam, sky = scenes.bunny(verts, indices)
    for _ in range(2):  # two contexts, one after the other: the first one's threads and staging go with it
        tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=2, sky=sky))
        try:
            T.rt_tracer_build_blas(tr, w)
            T.rt_tracer_build_tlas(tr, w)
            for wd, ht in ((1024, 512), (300, 200), (1920, 1080), (1024, 600), (64, 64), (2048, 1100)):
                c = cam(wd, ht, 1)
                dev = torch.zeros((ht, wd, 3), dtype=torch.float32, device="cuda:0")
                tr.check(tr.lib.rthip_cast_device(tr.h, abi.ptr(c), 21, dev.data_ptr(), wd, ht, None,
                                                  torch.cuda.current_stream().cuda_stream), "rthip_cast_device")
                torch.cuda.synchronize()
                want = dev.cpu().numpy()
                img = np.full((ht, wd, 3), -1.0, dtype=np.float32)
                T.rt_tracer_cast(tr, c, img, wd, ht, seed=21)
                assert (bits(img) == bits(want)).all(), ("pageable", wd, ht)
                pinned = T.host_alloc(tr, (ht, wd, 3))
                pinned[:] = -1.0
                T.rt_tracer_cast(tr, c, pinned, wd, ht, seed=21)
                assert (bits(pinned) == bits(want)).all(), ("page-locked", wd, ht)
                T.host_free(tr, pinned)
        finally:
            T.rt_tracer_cleanup(tr)


def test_group_of_contexts_renders_the_single_context_frame(gpu_tracer):
    """rthip_group_*: several contexts (here on one GPU) behind one handle.  Tiles: each context renders and ships only its
    own 64x64 tiles and the frame is the single context's bit for bit, into host memory and into device memory; samples:
    the partial sums are added in rank order.  Ragged frame sizes, more ranks than tiles, a one-context group."""
    import torch

    verts, indices = scenes.bunny_class_mesh(3)
    w, cam, sky = scenes.cornell(extra_mesh=(verts, indices))
    build(gpu_tracer, w, 6, sky)
    settings = T.RT_TracerSettings(max_bounces=6, sky=sky)
    for ids, sizes in (((0, 0), ((200, 150), (64, 64), (65, 129))), ((0, 0, 0, 0, 0), ((130, 70),)), ((0,), ((100, 90),))):
        for shard in ("tiles", "samples"):
            g = T.RT_TracerGroup(ids, settings, shard)
            try:
                g.build_blas(w)
                g.build_tlas(w)
                assert g.lib.rthip_group_size(g.g) == len(ids)
                for wd, ht in sizes:
                    c = cam(wd, ht, 2)
                    want = np.zeros((ht, wd, 3), dtype=np.float32)
                    T.rt_tracer_cast(gpu_tracer, c, want, wd, ht, seed=11)
                    got = np.full((ht, wd, 3), -1.0, dtype=np.float32)
                    g.cast(c, got, wd, ht, seed=11)
                    dev = torch.full((ht, wd, 3), -1.0, dtype=torch.float32, device="cuda:0")
                    g.check(g.lib.rthip_group_cast_device(g.g, abi.ptr(np.ascontiguousarray(c, dtype=abi.CAST).reshape(1)), 11,
                                                          dev.data_ptr(), wd, ht, g.shard_mode), "rthip_group_cast_device")
                    torch.cuda.synchronize()
                    if shard == "tiles" or len(ids) == 1:
                        assert (bits(got) == bits(want)).all(), (ids, shard, wd, ht)
                        assert (bits(dev.cpu().numpy()) == bits(want)).all(), (ids, shard, wd, ht, "device")
                    else:
                        assert np.allclose(got, want, rtol=2e-6, atol=1e-7) and np.allclose(dev.cpu().numpy(), want, rtol=2e-6, atol=1e-7)
                        assert (bits(dev.cpu().numpy()) == bits(got)).all()  # both paths add in rank order
                # the same tree on every context, and per-context taps work through the group
                t0, t1 = g.ctx(0), g.ctx(len(ids) - 1)
                assert T.get_tree(t0, -1).tobytes() == T.get_tree(t1, -1).tobytes() == T.get_tree(gpu_tracer, -1).tobytes()
            finally:
                g.close()
    with pytest.raises(T.RTHipError):
        T.RT_TracerGroup((0, 99), settings)


def test_shards_sum_to_the_frame(oracle, gpu_tracer):
    import ctypes as C

    import torch

    w, cam, sky = scenes.cornell()
    build(gpu_tracer, w, 8, sky)
    wd, ht, smp = 160, 130, 2
    c = cam(wd, ht, smp)
    full = np.zeros((ht, wd, 3), dtype=np.float32)
    T.rt_tracer_cast(gpu_tracer, c, full, wd, ht, seed=3)
    for mode, world_size in ((1, 4), (2, 3)):
        parts = []
        for rank in range(world_size):
            buf = torch.zeros((ht, wd, 3), dtype=torch.float32, device="cuda:0")
            sh = np.zeros(1, dtype=abi.SHARD)
            sh[0] = (mode, rank, world_size)
            rc = gpu_tracer.lib.rthip_cast_device(gpu_tracer.h, abi.ptr(c), 3, buf.data_ptr(), wd, ht, abi.ptr(sh),
                                                  torch.cuda.current_stream().cuda_stream)
            gpu_tracer.check(rc, "rthip_cast_device")
            torch.cuda.synchronize()
            parts.append(buf.cpu().numpy())
            want, _ = oracle.Scene(world.flatten_world(w), 8, sky).cast(c, 3, wd, ht, rng_mode=oracle.RNG_COUNTER,
                                                                       shard=(mode, rank, world_size), threads=8)
            assert (bits(parts[-1]) == bits(want)).all(), f"shard mode {mode} rank {rank}"
        total = np.sum(np.stack(parts), axis=0, dtype=np.float32)
        if mode == 1:
            assert (bits(total + np.float32(0)) == bits(full + np.float32(0))).all(), "tiles do not add up to the frame"
        else:
            total = total * np.float32(1.0 / (smp * smp))
            assert np.allclose(total, full, rtol=1e-5, atol=1e-6)


def test_random_worlds_match_reference(gpu_tracer):
    """The reference's own closest hits on seeded random worlds (every material type, mirrored and non-uniformly
    scaled instances, spheres, degenerate triangles)."""
    g = golden("random_worlds")
    for seed in g["seeds"].tolist():
        w, sky, bounces, o, d, cast = scenes.random_world_case(seed)
        build(gpu_tracer, w, bounces, sky)
        inst, tri, t, uv, nrm = T.trace_rays(gpu_tracer, o, d)
        assert (inst == g[f"w{seed}_ref_inst"]).all() and (tri == g[f"w{seed}_ref_tri"]).all(), seed
        assert (bits(t) == bits(g[f"w{seed}_ref_t"])).all(), seed
        assert hashlib.sha256(uv.tobytes()).hexdigest() == str(g[f"w{seed}_ref_uv_sha"]), seed
        assert hashlib.sha256(nan_free(nrm).tobytes()).hexdigest() == str(g[f"w{seed}_ref_normal_sha"]), seed


def test_random_worlds_every_kernel_against_the_oracle(oracle, gpu_tracer):
    """Random worlds, now also with coincident triangles (ties): trees, hits, occlusion and whole frames from every
    cast kernel, bit for bit against the CPU oracle drawing the same counter-based random numbers."""
    for seed in range(9000, 9120):
        check_random_world(oracle, gpu_tracer, seed)


def test_batched_instance_entries_change_no_bit(oracle):
    """The pixel- and sample-per-lane kernels with a wave's instance entries run in batches (forced on and off through
    RTHIP_SAMPLES_INST_BATCH, which a context reads when it is made): the Cornell box and a grid of small meshes."""
    verts, indices = scenes.bunny_class_mesh(0)  # 20 triangles
    cases = [(scenes.cornell(), 8, (96, 80)), (scenes.instanced_grid(5, seed=3, verts=verts, indices=indices), 6, (80, 45))]
    wanted = {}
    for batch in ("0", "7", "64"):
        os.environ["RTHIP_SAMPLES_INST_BATCH"] = batch
        try:
            tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=8, sky=True))
        finally:
            del os.environ["RTHIP_SAMPLES_INST_BATCH"]
        try:
            for k, ((w, cam, sky), bounces, (wd, ht)) in enumerate(cases):
                build(tr, w, bounces, sky)
                for kernel, smp in (("static", 1), ("static", 3), ("static-samples", 2), ("static-samples", 4)):
                    T.set_cast_kernel(tr, kernel)
                    img = np.zeros((ht, wd, 3), dtype=np.float32)
                    T.rt_tracer_cast(tr, cam(wd, ht, smp), img, wd, ht, seed=11)
                    key = (k, smp)
                    if key not in wanted:
                        sc = oracle.Scene(world.flatten_world(w), bounces, sky)
                        wanted[key], _ = sc.cast(cam(wd, ht, smp), 11, wd, ht, rng_mode=oracle.RNG_COUNTER, threads=8)
                    assert (bits(img) == bits(wanted[key])).all(), f"batch {batch}, scene {k}, {kernel}, samples {smp}"
        finally:
            T.rt_tracer_cleanup(tr)


def test_errors_are_reported(gpu_tracer):
    lib = gpu_tracer.lib
    assert lib.rthip_set_tracer(gpu_tracer.h, 64, 1) == 4  # RTHIP_E_LIMIT, max_bounces < 64
    assert b"max_bounces" in lib.rthip_last_error(gpu_tracer.h)
    assert lib.rthip_blas_begin(gpu_tracer.h, 1) == 0
    v = np.zeros((3, 4), dtype=np.float32)
    bad_idx = np.array([0, 1, 7], dtype=np.uint32)
    assert lib.rthip_blas_add_mesh(gpu_tracer.h, 0, abi.ptr(v), 3, 16, 0, -1, abi.ptr(bad_idx), 3) == 1
    assert lib.rthip_blas_add_mesh(gpu_tracer.h, 5, abi.ptr(v), 3, 16, 0, -1, None, 0) == 1
    assert lib.rthip_blas_commit(gpu_tracer.h) == 1  # mesh 0 never added
    assert lib.rthip_tlas_build(gpu_tracer.h, None, 0, None, 0) == 1  # before a committed BLAS set
    img = np.zeros((4, 4, 3), dtype=np.float32)
    c = scenes.get_rt_cast_settings(4, 4, 1, (0, 0, 1), (0, 0, 0))
    assert lib.rthip_cast(gpu_tracer.h, abi.ptr(c), 0, abi.ptr(img), 4, 4) == 1  # no TLAS


def test_animated_instances_rebuild_the_tlas_every_frame(oracle, gpu_tracer):
    """The reference animates by calling rt_tracer_build_tlas again (raytracer_cpu.c:32-37).  Four frames: instances
    move, turn, shrink, disappear and come back; the BLAS is built once.  Every frame's TLAS and hits against the
    oracle built from scratch for that frame."""
    verts, indices = scenes.bunny_class_mesh(3)
    w, cam, sky = scenes.instanced_grid(9, seed=3, verts=verts, indices=indices)
    gpu_tracer.check(gpu_tracer.lib.rthip_set_tracer(gpu_tracer.h, 3, 1 if sky else 0), "set_tracer")
    T.rt_tracer_build_blas(gpu_tracer, w)
    blas0 = T.get_tree(gpu_tracer, 0).copy()
    o, d = scenes.random_rays(6000, seed=8, lo=(-12, -2, -12), hi=(12, 3, 12))
    removed = []
    for frame in range(4):
        for k, it in enumerate(w.instances):
            a = 0.37 * frame + 0.11 * k
            it.translation = (np.float32(it.translation[0] + 0.4 * np.sin(a)), np.float32(it.translation[1] + 0.05 * frame),
                              np.float32(it.translation[2] + 0.4 * np.cos(a)))
            it.rotation = (np.float32(0), np.float32(np.sin(a / 2)), np.float32(0), np.float32(np.cos(a / 2)))
            s = np.float32(0.5 + 0.4 * abs(np.sin(a)))
            it.scale = (s, s, s)
        if frame == 1:  # 30 instances leave ...
            removed = [w.instances.pop() for _ in range(30)]
        if frame == 3:  # ... and return
            w.instances.extend(removed)
        T.rt_tracer_build_tlas(gpu_tracer, w)
        flat = world.flatten_world(w)
        sc = oracle.Scene(flat, 3, sky)
        assert_same_tree(T.get_tree(gpu_tracer, -1), sc.tree(-1), f"frame {frame} TLAS")
        inst, tri, t, uv, nrm = T.trace_rays(gpu_tracer, o, d)
        oi, ot, ott, ouv, onrm, _ = sc.trace(o, d, threads=8)
        assert (inst == oi).all() and (tri == ot).all() and (bits(t) == bits(ott)).all(), f"frame {frame}"
        assert (bits(uv) == bits(ouv)).all() and (bits(nrm) == bits(onrm)).all()
        assert int((inst > 0).sum()) > 200
        c = cam(64, 36, 1)
        img = np.zeros((36, 64, 3), dtype=np.float32)
        T.rt_tracer_cast(gpu_tracer, c, img, 64, 36, seed=frame)
        want, _ = sc.cast(c, frame, 64, 36, rng_mode=oracle.RNG_COUNTER, threads=8)
        assert (bits(img) == bits(want)).all(), f"frame {frame} radiance"
    assert T.get_tree(gpu_tracer, 0).tobytes() == blas0.tobytes(), "the BLAS must not change"


# ---- ray records ------------------------------------------------------------------------------


@pytest.mark.parametrize("name", ["cornell", "tri", "spheres"])
def test_ray_records_primary_rays_match_the_reference_dump(gpu_tracer, name):
    """rthip_cast_ray_records on a one-bounce, one-sample frame: every record (camera ray, hit point, normal, distance,
    material grouping) as in the out.rays file the reference's BUILD_DEBUG build wrote (no random numbers involved;
    the spheres camera has defocus, so only its hit / miss pattern is not compared there)."""
    from test_oracle_golden import same_ray_records

    g = golden("ray_records")
    key = f"{name}_16x16_s1_b1_seed1"
    w, cam, sky = {"cornell": scenes.cornell, "tri": scenes.tri, "spheres": scenes.spheres}[name]()
    build(gpu_tracer, w, 1, sky)
    cast = g[key + "_cast"].copy()
    if name == "spheres":
        cast["defocus"] = 0  # lens samples come from the generator, which differs from the reference's
    rec = T.cast_ray_records(gpu_tracer, cast, 16, 16, seed=1)
    assert rec.size == 256
    if name != "spheres":
        same_ray_records(rec, g, key)


def test_ray_records_match_the_oracle_over_whole_paths(oracle, gpu_tracer):
    """Multi-bounce frames: the records of every ray of every path, in the reference's order, bit for bit against the
    CPU restatement with the same counter-based generator; and the frame they describe is the frame rthip_cast renders."""
    verts, indices = scenes.bunny_class_mesh(3)
    for (w, cam, sky), bounces, (wd, ht, smp, seed) in (
            (scenes.cornell(), 6, (40, 30, 2, 4)), (scenes.spheres(), 8, (33, 21, 2, 9)),
            (scenes.instanced_grid(6, seed=2, verts=verts, indices=indices), 4, (48, 27, 1, 2))):
        build(gpu_tracer, w, bounces, sky)
        sc = oracle.Scene(world.flatten_world(w), bounces, sky)
        c = cam(wd, ht, smp)
        rec = T.cast_ray_records(gpu_tracer, c, wd, ht, seed=seed)
        want = sc.cast_records(c, seed, wd, ht, rng_mode=oracle.RNG_COUNTER)
        assert rec.size == want.size and rec.size > wd * ht * smp * smp
        assert rec.tobytes() == want.tobytes()
        T.set_counting(gpu_tracer, True)
        T.get_stats(gpu_tracer)
        img = np.zeros((ht, wd, 3), dtype=np.float32)
        T.rt_tracer_cast(gpu_tracer, c, img, wd, ht, seed=seed)
        assert T.get_stats(gpu_tracer)["rays"] == rec.size
        T.set_counting(gpu_tracer, False)
    # capacity too small: the count is still reported
    import ctypes as C
    n = C.c_uint64(0)
    small = np.zeros(8, dtype=abi.RAY_RECORD)
    cc = np.ascontiguousarray(c, dtype=abi.CAST).reshape(1)
    rc = gpu_tracer.lib.rthip_cast_ray_records(gpu_tracer.h, abi.ptr(cc), seed, wd, ht, abi.ptr(small), 8, C.addressof(n))
    assert rc == 4 and n.value == rec.size


# ---- OBJ files --------------------------------------------------------------------------------


def test_obj_loader_matches_reference(gpu_tracer, tmp_path):
    """rthip_obj_load (host threads parse, the GPU deduplicates) against the reference's ms_load_obj results."""
    from helpers import obj_written_columns

    g = golden("obj")
    for name, (text, attrs) in scenes.obj_test_files().items():
        path = tmp_path / (name + ".obj")
        path.write_bytes(text.encode())
        verts, idx, info = T.ms_load_obj(gpu_tracer, str(path), attrs)
        assert info["vertex_count"] == verts.shape[0] and info["index_count"] == idx.size
        written = np.ascontiguousarray(verts[:, obj_written_columns(attrs, verts.shape[1])])
        if name + "_ref_indices" in g.files:
            assert (idx == g[name + "_ref_indices"]).all(), name
            assert bits(written).tobytes() == bits(g[name + "_ref_vertices"]).tobytes(), name
        else:
            assert info["host_threads"] > 1, "the big file must be split over threads"
            assert (verts.shape[0], idx.size) == tuple(g[name + "_ref_counts"]), name
            assert hashlib.sha256(idx.tobytes()).hexdigest() == str(g[name + "_ref_indices_sha"]), name
            assert hashlib.sha256(written.tobytes()).hexdigest() == str(g[name + "_ref_vertices_sha"]), name


def test_obj_loader_random_texts(oracle, gpu_tracer, tmp_path):
    """Fuzzed OBJ texts: every face syntax, mixed number notations, blanks and tabs, foreign lines, CRLF, faces with more
    than three corners, with and without a final line end."""
    from helpers import obj_written_columns

    path = tmp_path / "fuzz.obj"
    for seed in range(1, 201):
        text, attrs = scenes.random_obj_text(seed)
        path.write_bytes(text.encode())
        verts, idx, info = T.ms_load_obj(gpu_tracer, str(path), attrs)
        want_verts, want_idx = oracle.load_obj(str(path), attrs)
        assert verts.shape == want_verts.shape and (idx == want_idx).all(), f"text {seed}"
        cols = obj_written_columns(attrs, verts.shape[1])
        assert (bits(verts[:, cols]) == bits(want_verts[:, cols])).all(), f"text {seed}"


def test_obj_loader_against_oracle_on_a_large_file(oracle, gpu_tracer, tmp_path):
    """300 k faces with heavily repeated index triples, all four syntaxes."""
    rng = np.random.default_rng(12)
    n_v, n_n, n_t, n_f = 50_000, 700, 900, 300_000
    head = ["v %.9g %.9g %.9g" % tuple(r) for r in rng.random((n_v, 3)).astype(np.float32)]
    head += ["vn %.9g %.9g %.9g" % tuple(r) for r in rng.standard_normal((n_n, 3)).astype(np.float32)]
    head += ["vt %.9g %.9g" % tuple(r) for r in rng.random((n_t, 2)).astype(np.float32)]
    fp = rng.integers(1, n_v + 1, (n_f, 3))
    fn = rng.integers(1, 5, (n_f, 3))
    ft = rng.integers(1, 4, (n_f, 3))
    fmt = {1: lambda p, n, t: "%d" % p, 3: lambda p, n, t: "%d//%d" % (p, n), 5: lambda p, n, t: "%d/%d" % (p, t),
           7: lambda p, n, t: "%d/%d/%d" % (p, t, n)}
    for attrs in (1, 3, 5, 7):
        faces = ["f " + " ".join(fmt[attrs](fp[k, c], fn[k, c], ft[k, c]) for c in range(3)) for k in range(n_f)]
        path = tmp_path / f"big_{attrs}.obj"
        path.write_text("\n".join(head + faces) + "\n")
        verts, idx, info = T.ms_load_obj(gpu_tracer, str(path), attrs)
        overts, oidx = oracle.load_obj(str(path), attrs)
        assert (idx == oidx).all() and bits(verts).tobytes() == bits(overts).tobytes(), attrs
        assert idx.size == 3 * (n_f + 1) and verts.shape[0] < idx.size


def test_loaded_obj_goes_to_the_blas_on_the_device(oracle, gpu_tracer, tmp_path):
    """rthip_blas_add_loaded_obj: the mesh built from the device-resident loader result equals the one built from the
    host arrays the loader returns (and hits equal the oracle's)."""
    text, attrs = scenes.obj_test_files()["pn_mesh"]
    path = tmp_path / "mesh.obj"
    path.write_bytes(text.encode())
    verts, idx, info = T.ms_load_obj(gpu_tracer, str(path), attrs)
    gpu_tracer.check(gpu_tracer.lib.rthip_set_tracer(gpu_tracer.h, 3, 1), "set_tracer")
    T.blas_add_loaded_obj(gpu_tracer)
    w = world.rt_make_world()
    m = world.rt_world_add_mesh(w)
    m.vertices, m.vertices_count, m.indices, m.indices_count, m.attrs = verts, verts.shape[0], idx, idx.size, attrs
    mat = world.rt_world_add_material(w)
    mat.type, mat.albedo = abi.MAT_LAMBERTIAN, (np.float32(0.7), np.float32(0.7), np.float32(0.7))
    it = world.rt_world_add_instance(w)
    it.type, it.material, it.mesh = abi.INST_MESH, mat, m
    it.translation, it.rotation, it.scale = (0, 0, 0), scenes.IDENTITY_QUAT, (1, 1, 1)
    T.rt_tracer_build_tlas(gpu_tracer, w)
    tree_device = T.get_tree(gpu_tracer, 0).copy()
    o, d = scenes.random_rays(5000, seed=3, lo=(-2, -1, -2), hi=(2, 3, 2))
    hits_device = T.trace_rays(gpu_tracer, o, d)
    T.rt_tracer_build_blas(gpu_tracer, w)  # the same mesh through the host arrays
    T.rt_tracer_build_tlas(gpu_tracer, w)
    assert T.get_tree(gpu_tracer, 0).tobytes() == tree_device.tobytes()
    hits_host = T.trace_rays(gpu_tracer, o, d)
    for a, b in zip(hits_device, hits_host):
        assert bits(a).tobytes() == bits(b).tobytes() if a.dtype == np.float32 else (a == b).all()
    sc = oracle.Scene(world.flatten_world(w), 3, True)
    oi, ot, ott, _, _, _ = sc.trace(o, d, threads=8)
    assert (hits_device[0] == oi).all() and (hits_device[1] == ot).all() and (bits(hits_device[2]) == bits(ott)).all()
    assert int((oi > 0).sum()) > 100


def test_obj_loader_errors(gpu_tracer, tmp_path):
    with pytest.raises(T.RTHipError, match="Failed to open"):
        T.ms_load_obj(gpu_tracer, str(tmp_path / "missing.obj"), 3)
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
    with pytest.raises(T.RTHipError, match="bad.obj:4"):
        T.ms_load_obj(gpu_tracer, str(bad), 3)  # PN asked for, the face has positions only
    far = tmp_path / "far.obj"
    far.write_text("v 0 0 0\nf 1 2 3")
    with pytest.raises(T.RTHipError, match="does not hold"):
        T.ms_load_obj(gpu_tracer, str(far), 1)
    empty = tmp_path / "empty.obj"
    empty.write_text("")
    verts, idx, info = T.ms_load_obj(gpu_tracer, str(empty), 1)
    assert verts.shape[0] == 0 and idx.size == 0


# ---- .hdr output ------------------------------------------------------------------------------


def test_hdr_encode_matches_reference_bytes(gpu_tracer):
    """rthip_hdr_encode against the bytes the reference's stbi_write_hdr wrote for the same images."""
    g = golden("hdr")
    names = sorted(k[:-4] for k in g.files if k.endswith("_hdr"))
    assert len(names) >= 20
    for name in names:
        img = g[name + "_rgb"]
        got = T.hdr_encode(gpu_tracer, img, img.shape[1], img.shape[0])
        assert got == g[name + "_hdr"].tobytes(), name


@pytest.mark.parametrize("w,h", [(8, 1), (9, 2), (10, 2), (11, 2), (63, 3), (64, 3), (65, 3), (127, 2), (128, 2), (129, 2), (191, 5),
                                 (192, 5), (253, 3), (254, 3), (255, 3), (256, 3), (257, 3), (258, 3), (259, 3), (511, 2), (512, 2),
                                 (513, 2), (767, 2), (770, 2), (1000, 11), (4097, 3), (16384, 2), (32767, 2), (32768, 2), (5, 7)])
def test_hdr_encode_shapes(oracle, gpu_tracer, w, h):
    rng = np.random.default_rng(w * 131 + h)
    img = (rng.random((h, w, 3), dtype=np.float32) * np.float32(3.0)).astype(np.float32)
    # runs that start and end at every offset relative to the 256-byte blocks a wave codes and the 4 bytes of a lane
    x = 0
    while x < w:
        run = int(rng.integers(1, 200))
        if rng.random() < 0.5:
            img[h // 2, x:x + run] = img[h // 2, x]
        x += run
    img[0, :, 2] = 0.25
    assert T.hdr_encode(gpu_tracer, img, w, h) == oracle.hdr_encode(img)


@pytest.mark.parametrize("seed", range(6))
def test_hdr_encode_short_runs_at_every_alignment(oracle, gpu_tracer, seed):
    """Images of a few grey levels in runs of 1 to 6 pixels (and, in other rows, of up to 400): runs of two (literal),
    three and four equal bytes begin and end at every position of a lane's four bytes and across the 256-byte blocks,
    and the 127 / 128-byte packet limits fall everywhere."""
    rng = np.random.default_rng(100 + seed)
    w, h = (int(rng.integers(8, 1500)), 24)
    img = np.zeros((h, w, 3), dtype=np.float32)
    levels = np.array([0.1, 0.2, 0.4, 0.8, 1.6, 0.0], dtype=np.float32)
    for y in range(h):
        longest = 6 if y % 3 else 400
        x = 0
        while x < w:
            run = int(rng.integers(1, longest + 1))
            img[y, x:x + run] = levels[rng.integers(0, len(levels), 3)] if y % 2 else levels[rng.integers(0, len(levels))]
            x += run
    assert T.hdr_encode(gpu_tracer, img, w, h) == oracle.hdr_encode(img)


def test_hdr_encode_long_runs_at_every_phase(oracle, gpu_tracer):
    """Runs long enough to hold whole 256-byte blocks (which the coder recognises from their words alone), beginning at
    every kind of offset and of lengths around the multiples of 127 and 256: packet ends on a block's first and last
    bytes, runs that end with a block, with a packet, with both, with the row."""
    rng = np.random.default_rng(9)
    starts = list(range(0, 300, 7)) + [253, 254, 255, 256, 257, 258]
    lengths = [257, 258, 381, 382, 383, 508, 509, 510, 511, 512, 513, 635, 636, 637, 762, 763, 1016, 1017, 1270, 1271]
    w = 300 + 1271 + 40
    rows = []
    for s0 in starts:
        for n in lengths:
            row = (rng.random((w, 3), dtype=np.float32) * np.float32(4.0)).astype(np.float32)
            row[s0:s0 + n] = np.float32(0.37)
            rows.append(row)
    for n in (w, w - 1, w - 3):  # a run to the row's end, from its start or behind a few bytes
        row = (rng.random((w, 3), dtype=np.float32) * np.float32(4.0)).astype(np.float32)
        row[w - n:] = np.float32(1.5)
        rows.append(row)
    img = np.stack(rows).astype(np.float32)
    got = T.hdr_encode(gpu_tracer, img, w, img.shape[0])
    want = oracle.hdr_encode(img)
    assert len(got) == len(want) and got == want
    # the same with rows a whole number of blocks wide (every block of the row can be such a block)
    img2 = np.ascontiguousarray(img[:200, :1536])
    img2[::3] = np.float32(0.0)
    img2[1::3, 5:] = np.float32(7.25)
    assert T.hdr_encode(gpu_tracer, img2, 1536, 200) == oracle.hdr_encode(img2)


def test_hdr_encode_full_frame(oracle, gpu_tracer):
    """A 3840 x 2160 frame with flat regions, gradients and noise."""
    w, h = 3840, 2160
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([xx / w, yy / h, np.float32(0.2) + 0 * xx], -1).astype(np.float32)
    img[500:900, 1000:2500] = np.float32(0.73)                                  # flat
    img[1200:1500] = rng.random((300, w, 3), dtype=np.float32) * np.float32(50)  # noise
    img[1800:, :, :] = 0.0                                                      # black
    got = T.hdr_encode(gpu_tracer, img, w, h)
    want = oracle.hdr_encode(img)
    assert len(got) == len(want)
    assert hashlib.sha256(got).hexdigest() == hashlib.sha256(want).hexdigest()


def test_hdr_encode_of_the_resident_frame(oracle, gpu_tracer):
    """rgb = NULL encodes the image the last rthip_cast left in device memory."""
    w, cam, sky = scenes.cornell()
    tr = T.rt_make_tracer(T.RT_TracerSettings(max_bounces=4, sky=sky))
    try:
        T.rt_tracer_build_blas(tr, w)
        T.rt_tracer_build_tlas(tr, w)
        W, H = 200, 120
        out = np.zeros((H, W, 3), np.float32)
        T.rt_tracer_cast(tr, cam(W, H, 2), out, W, H, seed=3)
        assert T.hdr_encode(tr, None, W, H) == oracle.hdr_encode(out)
        with pytest.raises(T.RTHipError):
            T.hdr_encode(tr, None, W + 1, H)
    finally:
        T.rt_tracer_cleanup(tr)


def test_hdr_encode_rejects_a_short_buffer(gpu_tracer):
    import ctypes as C
    img = np.random.default_rng(1).random((4, 64, 3), dtype=np.float32)
    out = np.zeros(64, np.uint8)
    size = C.c_uint64(0)
    rc = gpu_tracer.lib.rthip_hdr_encode(gpu_tracer.h, abi.ptr(img), 64, 4, abi.ptr(out), out.size, C.addressof(size))
    assert rc == 4 and size.value > 64  # RTHIP_E_LIMIT, and the length is still reported
