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
