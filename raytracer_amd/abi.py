"""ctypes view of include/rthip.h: record layouts and the loader for librthip.so.

The records are numpy structured dtypes whose field offsets equal the C structs
of include/rthip.h, so arrays of them can be handed to the C ABI as plain
pointers.  Nothing here computes anything: the library is the product, and
`load()` raises when it is missing instead of falling back to anything else.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTHIP_LIBRARY", os.path.join(HERE, "librthip.so"))  # override: a development build

# rthip_material (RT_Material, src/raytracer/raytracer_core.h:74-84)
MATERIAL = np.dtype(
    [("type", "<u4"), ("albedo", "<f4", 3), ("emissive", "<f4", 3), ("roughness", "<f4"), ("ior", "<f4"),
     ("billboard", "<u4")],
    align=False,
)
# rthip_instance (RT_Instance, src/raytracer/raytracer_core.h:17-45)
INSTANCE = np.dtype(
    [("type", "<u4"), ("material", "<i4"), ("blas_id", "<u4"), ("sphere_center", "<f4", 3), ("sphere_radius", "<f4"),
     ("translation", "<f4", 3), ("rotation", "<f4", 4), ("scale", "<f4", 3)],
    align=False,
)
# rthip_cast_settings (RT_CastSettings, src/raytracer/raytracer_core.h:163-178)
CAST = np.dtype(
    [("eye", "<f4", 3), ("up", "<f4", 3), ("forward", "<f4", 3), ("right", "<f4", 3), ("viewport", "<f4", 3),
     ("samples", "<u4"), ("ior", "<f4"), ("defocus", "<u4"), ("defocus_disk", "<f4", 2), ("orthographic", "<u4")],
    align=False,
)
# rthip_node
NODE = np.dtype([("aabb_min", "<f4", 3), ("skip", "<u4"), ("aabb_max", "<f4", 3), ("id", "<u4")], align=False)
# rthip_shard
SHARD = np.dtype([("mode", "<u4"), ("rank", "<u4"), ("world", "<u4")], align=False)
# rthip_stats
STATS = np.dtype(
    [("rays", "<u8"), ("nodes_tested", "<u8"), ("tris_tested", "<u8"), ("inst_entered", "<u8"),
     ("wave_steps", "<u8"), ("last_cast_ms", "<f4"), ("last_build_ms", "<f4"),
     ("last_cast_kernel", "<u4"), ("last_cast_refill", "<u4"),
     ("last_encode_ms", "<f4"), ("reserved", "<u4"), ("lane_steps", "<u8")],
    align=False,
)
# rthip_ray_record
RAY_RECORD = np.dtype([("origin", "<f4", 3), ("direction", "<f4", 3), ("p", "<f4", 3), ("n", "<f4", 3), ("t", "<f4"),
                       ("material", "<i4")], align=False)
assert RAY_RECORD.itemsize == 56
# rthip_obj_info
OBJ_INFO = np.dtype([("vertex_count", "<u4"), ("index_count", "<u4"), ("attrs", "<u4"), ("vertex_bytes", "<u4"),
                     ("parse_ms", "<f4"), ("dedup_ms", "<f4"), ("total_ms", "<f4"), ("host_threads", "<u4")], align=False)
assert OBJ_INFO.itemsize == 32
assert MATERIAL.itemsize == 40 and INSTANCE.itemsize == 68 and CAST.itemsize == 84
assert NODE.itemsize == 32 and SHARD.itemsize == 12 and STATS.itemsize == 72

MAT_LAMBERTIAN, MAT_DIELECTRIC, MAT_METAL, MAT_NORMAL, MAT_LIGHT = range(5)
INST_SPHERE, INST_MESH = 0, 1

# every symbol include/rthip.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("rthip_create", C.c_int, [C.POINTER(_P), C.c_int]),
    ("rthip_destroy", None, [_P]),
    ("rthip_last_error", C.c_char_p, [_P]),
    ("rthip_abi_version", C.c_int, []),
    ("rthip_set_tracer", C.c_int, [_P, C.c_uint32, C.c_int]),
    ("rthip_set_morton_bounds", C.c_int, [_P, C.c_int]),
    ("rthip_blas_begin", C.c_int, [_P, C.c_uint32]),
    ("rthip_blas_add_mesh", C.c_int,
     [_P, C.c_uint32, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, _P, C.c_uint32]),
    ("rthip_blas_commit", C.c_int, [_P]),
    ("rthip_tlas_build", C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32]),
    ("rthip_cast", C.c_int, [_P, _P, C.c_uint64, _P, C.c_int, C.c_int]),
    ("rthip_host_alloc", C.c_int, [_P, C.c_uint64, C.POINTER(_P)]),
    ("rthip_host_free", C.c_int, [_P, _P]),
    ("rthip_cast_device", C.c_int, [_P, _P, C.c_uint64, _P, C.c_int, C.c_int, _P, _P]),
    ("rthip_cast_accumulate_device", C.c_int, [_P, _P, C.c_uint64, _P, C.c_int, C.c_int, _P, _P]),
    ("rthip_group_create", C.c_int, [C.POINTER(_P), C.c_int, _P]),
    ("rthip_group_destroy", None, [_P]),
    ("rthip_group_size", C.c_int, [_P]),
    ("rthip_group_ctx", _P, [_P, C.c_int]),
    ("rthip_group_last_error", C.c_char_p, [_P]),
    ("rthip_group_transport", C.c_char_p, [_P]),
    ("rthip_group_set_tracer", C.c_int, [_P, C.c_uint32, C.c_int]),
    ("rthip_group_set_morton_bounds", C.c_int, [_P, C.c_int]),
    ("rthip_group_set_traversal", C.c_int, [_P, C.c_int]),
    ("rthip_group_blas_begin", C.c_int, [_P, C.c_uint32]),
    ("rthip_group_blas_add_mesh", C.c_int,
     [_P, C.c_uint32, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, _P, C.c_uint32]),
    ("rthip_group_blas_commit", C.c_int, [_P]),
    ("rthip_group_tlas_build", C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint32]),
    ("rthip_group_cast", C.c_int, [_P, _P, C.c_uint64, _P, C.c_int, C.c_int, C.c_int]),
    ("rthip_group_cast_device", C.c_int, [_P, _P, C.c_uint64, _P, C.c_int, C.c_int, C.c_int]),
    ("rthip_tiles_of_rank", C.c_uint32, [C.c_int, C.c_int, C.c_uint32, C.c_uint32]),
    ("rthip_tiles_pack_device", C.c_int, [_P, _P, C.c_int, C.c_int, C.c_uint32, C.c_uint32, _P, _P]),
    ("rthip_tiles_unpack_device", C.c_int, [_P, _P, C.c_int, C.c_int, C.c_uint32, C.c_uint32, _P, _P]),
    ("rthip_lbvh_build", C.c_int, [_P, _P, C.c_uint64, _P, _P, _P, _P, _P]),
    ("rthip_lbvh_build_device", C.c_int, [_P, _P, C.c_uint64, _P, _P]),
    ("rthip_lbvh_reserve", C.c_int, [_P, C.c_uint64]),
    ("rthip_lbvh_build_device_noalloc", C.c_int, [_P, _P, C.c_uint64, _P, _P]),
    ("rthip_mesh_build_device", C.c_int, [_P, _P, _P, _P, C.c_uint32, _P, C.c_uint64, _P, _P]),
    ("rthip_primary_hits", C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int]),
    ("rthip_trace_rays", C.c_int, [_P, _P, _P, C.c_uint64, _P, _P, _P, _P, _P]),
    ("rthip_trace_rays_device", C.c_int, [_P, _P, _P, C.c_uint64, _P, _P]),
    ("rthip_occluded", C.c_int, [_P, _P, _P, C.c_uint64, C.c_float, _P]),
    ("rthip_occluded_device", C.c_int, [_P, _P, _P, C.c_uint64, C.c_float, _P, _P]),
    ("rthip_get_stats", C.c_int, [_P, _P]),
    ("rthip_set_counting", C.c_int, [_P, C.c_int]),
    ("rthip_set_traversal", C.c_int, [_P, C.c_int]),
    ("rthip_set_cast_kernel", C.c_int, [_P, C.c_int]),
    ("rthip_set_persistent_params", C.c_int, [_P, C.c_uint32, C.c_uint32]),
    ("rthip_get_tree", C.c_int, [_P, C.c_int, _P, _P]),
    ("rthip_cast_ray_records", C.c_int, [_P, _P, C.c_uint64, C.c_int, C.c_int, _P, C.c_uint64, _P]),
    ("rthip_obj_load", C.c_int, [_P, C.c_char_p, C.c_uint32, _P]),
    ("rthip_obj_get", C.c_int, [_P, _P, _P]),
    ("rthip_blas_add_loaded_obj", C.c_int, [_P, C.c_uint32]),
    ("rthip_hdr_bound", C.c_uint64, [C.c_int, C.c_int]),
    ("rthip_hdr_encode", C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_uint64, _P]),
    ("rthip_hdr_encode_device", C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_uint64, _P, _P]),
]

_lib = None


def load() -> C.CDLL:
    """Loads raytracer_amd/librthip.so (built by __graft_entry__.build()); no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
            )
        # One process must hold ONE HIP runtime.  PyTorch ships its own copy (same soname as the
        # system one); when torch is installed it is loaded first, so that librthip.so binds to the
        # runtime torch uses and device pointers / streams can be shared between the two.
        try:
            import torch  # noqa: F401
        except Exception:  # torch is plumbing for the Python harnesses only
            pass
        lib = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def ptr(a) -> int | None:
    """Address of a numpy array's buffer (None for None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
    return a.ctypes.data
