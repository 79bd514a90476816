"""Shared checks for the parity tests."""
import numpy as np


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def assert_same_tree(got, want, what=""):
    """Pre-order node arrays must agree in every field, floats compared as bit patterns."""
    assert got.shape == want.shape, f"{what}: node count {got.shape} vs {want.shape}"
    for f in ("id", "skip"):
        bad = np.nonzero(got[f] != want[f])[0]
        assert bad.size == 0, f"{what}: field {f} differs at {bad[:8]} (of {bad.size})"
    for f in ("aabb_min", "aabb_max"):
        bad = np.nonzero((bits(got[f]) != bits(want[f])).any(axis=1))[0]
        assert bad.size == 0, f"{what}: field {f} differs at {bad[:8]} (of {bad.size})"


def leaf_ranges(nodes):
    """(first, last) sorted-leaf position covered by every pre-order node."""
    n = nodes.shape[0]
    is_leaf = nodes["id"] > 0
    leaf_pos = np.cumsum(is_leaf) - 1  # position of the most recent leaf at or before each node
    first = np.where(is_leaf, leaf_pos, leaf_pos + 1)
    last = leaf_pos[nodes["skip"].astype(np.int64) - 1]
    return first, last


def assert_same_tree_modulo_runs(got, want, sorted_codes, what=""):
    """Shape must be identical; leaves may be permuted inside runs of equal Morton codes (the
    reference's sort is unstable there), so boxes are compared only for nodes whose leaf range
    starts and ends on run boundaries, and leaf ids as multisets per run."""
    assert got.shape == want.shape
    assert ((got["id"] > 0) == (want["id"] > 0)).all(), f"{what}: leaf/inner pattern differs"
    assert (got["skip"] == want["skip"]).all(), f"{what}: skip links differ"
    n = sorted_codes.shape[0]
    run_start = np.ones(n, dtype=bool)
    run_start[1:] = sorted_codes[1:] != sorted_codes[:-1]
    run_end = np.ones(n, dtype=bool)
    run_end[:-1] = run_start[1:]
    run_id = np.cumsum(run_start) - 1
    gl, wl = got[got["id"] > 0], want[want["id"] > 0]
    for r in np.unique(run_id):
        sel = run_id == r
        assert sorted(gl["id"][sel].tolist()) == sorted(wl["id"][sel].tolist()), f"{what}: run {r} holds other leaves"
    first, last = leaf_ranges(want)
    whole = run_start[first] & run_end[last]
    for f in ("aabb_min", "aabb_max"):
        same = (bits(got[f]) == bits(want[f])).all(axis=1)
        # signed zeros may legitimately differ with leaf order; compare values there
        close = (got[f] == want[f]).all(axis=1)
        assert (same | close)[whole].all(), f"{what}: {f} differs on run-aligned nodes"


def obj_written_columns(attrs: int, width: int):
    """Columns of an ms_load_obj vertex row that an attribute writes (3 of every 16-byte slot, 2 for the uv slot)."""
    attrs = attrs or 7
    slots = width // 4
    cols, slot = [], 0
    for bit, n in ((1, 3), (2, 3), (4, 2)):
        if attrs & bit:
            cols += [4 * slot + k for k in range(n)]
            slot += 1
    assert slot == slots
    return np.array(cols)


def build_and_check_invariants(gpu_tracer, n, seed=3, builds=2):
    """Builds the BLAS tree of n seeded random triangles on the GPU (vertices generated there) and checks what holds
    at any size: a well-formed pre-order tree, every primitive in exactly one leaf with its own box, every inner box
    the exact merge of its two children, and the same bits from every build.  Returns the last build's milliseconds."""
    import torch

    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    cen = torch.rand((n, 1, 3), device="cuda", generator=g)
    off = (torch.rand((n, 3, 3), device="cuda", generator=g) * 2 - 1) * 0.001
    p = (cen + off).reshape(3 * n, 3).float()
    del cen, off
    px, py, pz = (p[:, k].contiguous() for k in range(3))
    stream = torch.cuda.current_stream().cuda_stream
    N = 2 * n - 1
    first = None
    for _ in range(builds):
        nodes = torch.empty((N, 8), dtype=torch.float32, device="cuda")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gpu_tracer.check(gpu_tracer.lib.rthip_mesh_build_device(gpu_tracer.h, px.data_ptr(), py.data_ptr(), pz.data_ptr(),
                                                                3 * n, None, n, nodes.data_ptr(), stream), "build")
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        if first is None:
            first = nodes
        else:
            assert torch.equal(first.view(torch.int32), nodes.view(torch.int32)), "two builds differ"
            del nodes
    nodes = first
    ints = nodes.view(torch.int32)
    skip, ids = ints[:, 3].long(), ints[:, 7].long()
    idx = torch.arange(N, device="cuda")
    assert int(skip[0]) == N and bool(((skip > idx) & (skip <= N)).all())
    leaf = ids != 0
    assert int(leaf.sum()) == n and bool((skip[leaf] == idx[leaf] + 1).all())
    assert bool((torch.sort(ids[leaf]).values == torch.arange(1, n + 1, device="cuda")).all()), "ids are not a permutation"
    # leaf boxes are the triangles' own boxes
    tri = p.reshape(n, 3, 3)
    prim = ids[leaf] - 1
    assert torch.equal(nodes[leaf][:, 0:3], tri.min(dim=1).values[prim]) and torch.equal(nodes[leaf][:, 4:7], tri.max(dim=1).values[prim])
    # inner node i: children i + 1 and skip[i + 1]; box = exact merge
    inner = idx[~leaf]
    left = inner + 1
    right = skip[left]
    assert bool((right < N).all()) and bool((skip[right] == skip[inner]).all()), "right child does not end where the parent ends"
    assert torch.equal(nodes[inner][:, 0:3], torch.minimum(nodes[left][:, 0:3], nodes[right][:, 0:3]))
    assert torch.equal(nodes[inner][:, 4:7], torch.maximum(nodes[left][:, 4:7], nodes[right][:, 4:7]))
    return ms


def same_or_both_nan(a, b):
    """Element-wise: the same bits, or a NaN on both sides (the sign and payload of a NaN are the hardware's)."""
    a, b = np.asarray(a), np.asarray(b)
    return (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))


def check_random_world(oracle, gpu_tracer, seed, kernels=("auto", "static", "persistent", "wavefront", "static-samples")):
    """One seeded random world (scenes.random_world): trees, closest hits and occlusion of random rays, and whole
    frames from every cast kernel, against the CPU restatement drawing the same counter-based random numbers."""
    from raytracer_amd import scenes, world, tracer as T

    w, cam, sky, bounces = scenes.random_world(seed)
    gpu_tracer.check(gpu_tracer.lib.rthip_set_tracer(gpu_tracer.h, bounces, 1 if sky else 0), "set_tracer")
    T.rt_tracer_build_blas(gpu_tracer, w)
    T.rt_tracer_build_tlas(gpu_tracer, w)
    flat = world.flatten_world(w)
    sc = oracle.Scene(flat, bounces, sky)
    try:
        assert_same_tree(T.get_tree(gpu_tracer, -1), sc.tree(-1), f"seed {seed} TLAS")
        for k in range(len(flat.meshes)):
            assert_same_tree(T.get_tree(gpu_tracer, k), sc.tree(k), f"seed {seed} BLAS {k}")
        o, d = scenes.aimed_rays(3000, seed)
        inst, tri, t, uv, nrm = T.trace_rays(gpu_tracer, o, d)
        oi, ot, ott, ouv, onrm, _ = sc.trace(o, d, threads=4)
        assert (inst == oi).all() and (tri == ot).all() and (bits(t) == bits(ott)).all(), f"closest hits differ ({int((inst != oi).sum())} ids)"
        assert (bits(uv) == bits(ouv)).all(), "uv differ"
        assert same_or_both_nan(nrm, onrm).all(), "normals differ"
        occ = T.occluded(gpu_tracer, o, d, 4.0)
        assert (occ == sc.occluded(o, d, 4.0)).all(), "occlusion differs"
        # the ordered walk on the same rays (they are aimed at vertices and edges, so seams are common): never hit
        # against miss, the same occlusion answers, the distance equal up to a seam's 1-2 ulp
        T.set_traversal(gpu_tracer, "fast")
        try:
            finst, _, ft, _, _ = T.trace_rays(gpu_tracer, o, d)
            focc = T.occluded(gpu_tracer, o, d, 4.0)
        finally:
            T.set_traversal(gpu_tracer, "exact")
        assert ((finst != 0) == (inst != 0)).all(), "ordered walk: hit against miss"
        assert (focc == occ).all(), "ordered walk: occlusion differs"
        hit = inst != 0
        assert np.allclose(ft[hit], t[hit], rtol=1e-6, atol=0.0), "ordered walk: another distance"
        rng = np.random.default_rng(seed)
        wd, ht, smp = int(rng.integers(8, 70)), int(rng.integers(8, 50)), int(rng.choice([1, 2, 2, 3, 4]))
        c = cam(wd, ht, smp)
        want, _ = sc.cast(c, seed, wd, ht, rng_mode=oracle.RNG_COUNTER, threads=8)
        for kernel in kernels:
            T.set_cast_kernel(gpu_tracer, kernel)
            img = np.full((ht, wd, 3), -1.0, dtype=np.float32)
            T.rt_tracer_cast(gpu_tracer, c, img, wd, ht, seed=seed)
            bad = np.nonzero(~same_or_both_nan(img, want).all(axis=2))
            assert bad[0].size == 0, (f"{kernel} {wd}x{ht} samples {smp} bounces {bounces}: {bad[0].size} pixels differ, "
                                      f"first (y, x) = {list(zip(bad[0][:3].tolist(), bad[1][:3].tolist()))}")
        # the records of every ray of every path, and one shard of the frame (tiles or sample indices)
        T.set_cast_kernel(gpu_tracer, "auto")
        rec = T.cast_ray_records(gpu_tracer, c, wd, ht, seed=seed)
        want_rec = sc.cast_records(c, seed, wd, ht, rng_mode=oracle.RNG_COUNTER)
        assert rec.size == want_rec.size, f"{rec.size} ray records, the oracle has {want_rec.size}"
        for f in rec.dtype.names:
            same = same_or_both_nan(rec[f], want_rec[f]) if rec[f].dtype == np.float32 else rec[f] == want_rec[f]
            assert same.all(), f"ray records differ in {f}"
        from raytracer_amd import abi
        mode, world_size = int(rng.integers(1, 3)), int(rng.integers(2, 5))
        rank = int(rng.integers(0, world_size))
        sh = np.zeros(1, dtype=abi.SHARD)
        sh[0] = (mode, rank, world_size)
        import torch
        buf = torch.zeros((ht, wd, 3), dtype=torch.float32, device="cuda:0")
        gpu_tracer.check(gpu_tracer.lib.rthip_cast_device(gpu_tracer.h, abi.ptr(c), seed, buf.data_ptr(), wd, ht, abi.ptr(sh),
                                                          torch.cuda.current_stream().cuda_stream), "rthip_cast_device")
        torch.cuda.synchronize()
        want_part, _ = sc.cast(c, seed, wd, ht, rng_mode=oracle.RNG_COUNTER, shard=(mode, rank, world_size), threads=8)
        assert same_or_both_nan(buf.cpu().numpy(), want_part).all(), f"shard mode {mode} rank {rank} of {world_size}"
    finally:
        T.set_cast_kernel(gpu_tracer, "auto")
        sc.close()


def nan_free(a):
    """NaNs replaced by one fixed pattern before hashing (their sign and payload belong to the hardware)."""
    a = np.array(a, dtype=np.float32, copy=True)
    a[np.isnan(a)] = np.float32(12345.0)
    return a
