"""The five tracer hooks of the reference (src/raytracer/raytracer_core.h:180-184)
on top of the C ABI of librthip.so, plus thin wrappers of the parity taps.

Everything is computed by the HIP library; a missing library or a failing call
raises RTHipError -- there is no CPU path behind these functions.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from . import abi
from .world import RT_World, flatten_meshes, flatten_world


class RTHipError(RuntimeError):
    pass


@dataclass
class RT_TracerSettings:  # raytracer_core.h:155-159
    max_bounces: int = 8
    sky: bool = True


class RT_Tracer:
    """Owner of one rthip_ctx (one GPU)."""

    def __init__(self, device_id: int = 0):
        self.lib = abi.load()
        h = C.c_void_p()
        rc = self.lib.rthip_create(C.byref(h), device_id)
        if rc != 0 or not h.value:
            raise RTHipError(f"rthip_create failed with code {rc} (no usable gfx950 device {device_id}?)")
        self.h = h

    def check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.rthip_last_error(self.h)
            raise RTHipError(f"{what}: code {rc}: {msg.decode() if msg else ''}")

    def close(self):
        if getattr(self, "h", None) is not None and self.h.value:
            self.lib.rthip_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rt_make_tracer(settings: RT_TracerSettings, device_id: int = 0) -> RT_Tracer:
    """rt_make_tracer, src/raytracer/cpu/raytracer_cpu.c:14-25."""
    assert settings.max_bounces < 64  # RT_MAX_MAX_BOUNCES, raytracer_core.h:161
    t = RT_Tracer(device_id)
    t.check(t.lib.rthip_set_tracer(t.h, int(settings.max_bounces), 1 if settings.sky else 0), "rthip_set_tracer")
    return t


def rt_tracer_build_blas(tracer: RT_Tracer, world: RT_World) -> None:
    """rt_tracer_build_blas, raytracer_cpu.c:26-31."""
    meshes = flatten_meshes(world)
    upload_meshes(tracer, meshes)


def upload_meshes(tracer: RT_Tracer, meshes: list) -> None:
    lib = tracer.lib
    tracer.check(lib.rthip_blas_begin(tracer.h, len(meshes)), "rthip_blas_begin")
    for blas_id, m in enumerate(meshes):
        idx = m["indices"]
        tracer.check(
            lib.rthip_blas_add_mesh(
                tracer.h, blas_id, abi.ptr(m["verts"]), m["vertex_count"], m["stride"], m["pos_off"], m["nrm_off"],
                abi.ptr(idx) if idx is not None else None, 0 if idx is None else idx.size,
            ),
            "rthip_blas_add_mesh",
        )
    tracer.check(lib.rthip_blas_commit(tracer.h), "rthip_blas_commit")


def rt_tracer_build_tlas(tracer: RT_Tracer, world: RT_World) -> None:
    """rt_tracer_build_tlas, raytracer_cpu.c:32-37."""
    flat = flatten_world(world)
    build_tlas_flat(tracer, flat.instances, flat.materials)


def build_tlas_flat(tracer: RT_Tracer, instances: np.ndarray, materials: np.ndarray) -> None:
    instances = np.ascontiguousarray(instances, dtype=abi.INSTANCE)
    materials = np.ascontiguousarray(materials, dtype=abi.MATERIAL)
    tracer.check(
        tracer.lib.rthip_tlas_build(tracer.h, abi.ptr(instances), instances.size, abi.ptr(materials), materials.size),
        "rthip_tlas_build",
    )


def rt_tracer_cast(tracer: RT_Tracer, settings: np.ndarray, out_radiance: np.ndarray, width: int, height: int,
                   seed: int = 0) -> None:
    """rt_tracer_cast, raytracer_cpu.c:45-49.  out_radiance: float32 [height, width, 3]."""
    cast = np.ascontiguousarray(settings, dtype=abi.CAST).reshape(1)
    assert out_radiance.dtype == np.float32 and out_radiance.size == width * height * 3
    tracer.check(
        tracer.lib.rthip_cast(tracer.h, abi.ptr(cast), int(seed) & (2**64 - 1), abi.ptr(out_radiance), width, height),
        "rthip_cast",
    )


def rt_tracer_cleanup(tracer: RT_Tracer) -> None:
    """rt_tracer_cleanup, raytracer_cpu.c:38-43."""
    tracer.close()


# ---- parity / measurement taps ---------------------------------------------


def lbvh_build(tracer: RT_Tracer, aabbs: np.ndarray, want_codes: bool = True):
    """lbvh_make (src/lbvh/lbvh.c:92-125) on the GPU.  Returns (codes, codes_sorted, ids_sorted, nodes, ms).
    want_codes=False does not ask for the unsorted codes (None is returned for them): the build then runs as it does
    for a scene, where the sort's first kernel makes the codes instead of a kernel of their own."""
    aabbs = np.ascontiguousarray(aabbs, dtype=np.float32).reshape(-1, 6)
    n = aabbs.shape[0]
    codes = np.zeros(n, dtype=np.uint64) if want_codes else None
    codes_sorted = np.zeros(n, dtype=np.uint64)
    ids_sorted = np.zeros(n, dtype=np.uint32)
    nodes = np.zeros(max(2 * n - 1, 0), dtype=abi.NODE)
    ms = C.c_float(0)
    tracer.check(
        tracer.lib.rthip_lbvh_build(tracer.h, abi.ptr(aabbs), n, abi.ptr(codes), abi.ptr(codes_sorted),
                                    abi.ptr(ids_sorted), abi.ptr(nodes), C.addressof(ms)),
        "rthip_lbvh_build",
    )
    return codes, codes_sorted, ids_sorted, nodes, ms.value


def primary_hits(tracer: RT_Tracer, settings: np.ndarray, width: int, height: int):
    cast = np.ascontiguousarray(settings, dtype=abi.CAST).reshape(1)
    inst = np.zeros(width * height, dtype=np.uint32)
    tri = np.zeros(width * height, dtype=np.uint32)
    t = np.zeros(width * height, dtype=np.float32)
    tracer.check(
        tracer.lib.rthip_primary_hits(tracer.h, abi.ptr(cast), abi.ptr(inst), abi.ptr(tri), abi.ptr(t), width, height),
        "rthip_primary_hits",
    )
    return inst, tri, t


def trace_rays(tracer: RT_Tracer, origins: np.ndarray, directions: np.ndarray, want_uv=True, want_normal=True):
    o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
    n = o.shape[0]
    inst = np.zeros(n, dtype=np.uint32)
    tri = np.zeros(n, dtype=np.uint32)
    t = np.zeros(n, dtype=np.float32)
    uv = np.zeros((n, 2), dtype=np.float32) if want_uv else None
    nrm = np.zeros((n, 3), dtype=np.float32) if want_normal else None
    tracer.check(
        tracer.lib.rthip_trace_rays(tracer.h, abi.ptr(o), abi.ptr(d), n, abi.ptr(inst), abi.ptr(tri), abi.ptr(t),
                                    abi.ptr(uv), abi.ptr(nrm)),
        "rthip_trace_rays",
    )
    return inst, tri, t, uv, nrm


def get_tree(tracer: RT_Tracer, which: int) -> np.ndarray:
    cnt = C.c_uint64(0)
    tracer.check(tracer.lib.rthip_get_tree(tracer.h, which, None, C.addressof(cnt)), "rthip_get_tree")
    nodes = np.zeros(cnt.value, dtype=abi.NODE)
    tracer.check(tracer.lib.rthip_get_tree(tracer.h, which, abi.ptr(nodes), C.addressof(cnt)), "rthip_get_tree")
    return nodes


def get_stats(tracer: RT_Tracer) -> dict:
    st = np.zeros(1, dtype=abi.STATS)
    tracer.check(tracer.lib.rthip_get_stats(tracer.h, abi.ptr(st)), "rthip_get_stats")
    return {k: st[0][k].item() for k in abi.STATS.names}


def occluded(tracer: RT_Tracer, origins, directions, tmax: float = 3.402823466e+38) -> np.ndarray:
    """1 per ray whose TLAS query over [EPSILON_F32, tmax] reports a hit (any-hit walk), else 0."""
    o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
    d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
    out = np.zeros(o.shape[0], dtype=np.uint8)
    tracer.check(tracer.lib.rthip_occluded(tracer.h, abi.ptr(o), abi.ptr(d), o.shape[0], float(tmax), abi.ptr(out)),
                 "rthip_occluded")
    return out


CAST_KERNELS = {"persistent": 0, "static": 1, "wavefront": 2, "auto": 3, "static-samples": 4}


def cast_ray_records(tracer: RT_Tracer, settings: np.ndarray, width: int, height: int, seed: int = 0) -> np.ndarray:
    """The out.rays records of the frame rt_tracer_cast renders with the same seed (abi.RAY_RECORD; the dump of the
    reference's BUILD_DEBUG build, src/raytracer/cpu/extra/dump.c:11-32), in the reference's order."""
    cast = np.ascontiguousarray(settings, dtype=abi.CAST).reshape(1)
    n = C.c_uint64(0)
    tracer.check(tracer.lib.rthip_cast_ray_records(tracer.h, abi.ptr(cast), int(seed) & (2**64 - 1), width, height, None, 0,
                                                   C.addressof(n)), "rthip_cast_ray_records")
    out = np.zeros(n.value, dtype=abi.RAY_RECORD)
    if n.value:
        tracer.check(tracer.lib.rthip_cast_ray_records(tracer.h, abi.ptr(cast), int(seed) & (2**64 - 1), width, height,
                                                       abi.ptr(out), n.value, C.addressof(n)), "rthip_cast_ray_records")
    return out


def ms_load_obj(tracer: RT_Tracer, path: str, attrs: int = 0, read_back: bool = True):
    """ms_load_obj, src/mesh/mesh.c:73-219: (vertices float32 [vertex_count, 4 * attributes], indices uint32, info).
    attrs: GEO_VertexAttributes bits (P = 1, N = 2, T = 4; 0 = all).  The mesh stays in device memory for
    blas_add_loaded_obj; read_back=False skips the copy to the host and returns (None, None, info)."""
    info = np.zeros(1, dtype=abi.OBJ_INFO)
    tracer.check(tracer.lib.rthip_obj_load(tracer.h, os.fsencode(path), attrs, abi.ptr(info)), "rthip_obj_load")
    info = {k: info[0][k].item() for k in abi.OBJ_INFO.names}
    if not read_back:
        return None, None, info
    verts = np.zeros((info["vertex_count"], info["vertex_bytes"] // 4), dtype=np.float32)
    idx = np.zeros(info["index_count"], dtype=np.uint32)
    tracer.check(tracer.lib.rthip_obj_get(tracer.h, abi.ptr(verts), abi.ptr(idx)), "rthip_obj_get")
    return verts, idx, info


def blas_add_loaded_obj(tracer: RT_Tracer, n_meshes: int = 1, blas_id: int = 0, commit: bool = True) -> None:
    """The loaded mesh as mesh `blas_id` of a BLAS set of n_meshes, device to device."""
    if blas_id == 0:
        tracer.check(tracer.lib.rthip_blas_begin(tracer.h, n_meshes), "rthip_blas_begin")
    tracer.check(tracer.lib.rthip_blas_add_loaded_obj(tracer.h, blas_id), "rthip_blas_add_loaded_obj")
    if commit:
        tracer.check(tracer.lib.rthip_blas_commit(tracer.h), "rthip_blas_commit")


def hdr_encode(tracer: RT_Tracer, rgb, width: int, height: int) -> bytes:
    """The bytes of the .hdr file the reference writes with stbi_write_hdr(path, width, height, 3, rgb)
    (stb_image_write.h:794; src/demos/cornell/main.c:222), encoded on the GPU.  rgb: float32 [height, width, 3], or
    None for the image of the last rt_tracer_cast (still in device memory)."""
    if rgb is not None:
        rgb = np.ascontiguousarray(rgb, dtype=np.float32)
        assert rgb.size == width * height * 3
    cap = int(tracer.lib.rthip_hdr_bound(width, height))
    out = np.empty(cap, dtype=np.uint8)
    size = C.c_uint64(0)
    tracer.check(tracer.lib.rthip_hdr_encode(tracer.h, abi.ptr(rgb), width, height, abi.ptr(out), cap, C.addressof(size)),
                 "rthip_hdr_encode")
    return out[: size.value].tobytes()


def write_hdr(tracer: RT_Tracer, path: str, rgb, width: int, height: int) -> None:
    """stbi_write_hdr's file, encoded on the GPU (see hdr_encode)."""
    with open(path, "wb") as f:
        f.write(hdr_encode(tracer, rgb, width, height))


def set_cast_kernel(tracer: RT_Tracer, kernel: str) -> None:
    """Selects the renderer implementation (all produce the same bits); see include/rthip.h."""
    tracer.check(tracer.lib.rthip_set_cast_kernel(tracer.h, CAST_KERNELS[kernel]), "rthip_set_cast_kernel")


def set_persistent_params(tracer: RT_Tracer, refill: int = 16, inst_batch: int = 0) -> None:
    """Scheduling of the persistent kernel when it is selected by name; see include/rthip.h."""
    tracer.check(tracer.lib.rthip_set_persistent_params(tracer.h, refill, inst_batch), "rthip_set_persistent_params")


def set_morton_bounds(tracer: RT_Tracer, boxes: bool) -> None:
    """Trees built from now on normalise centroids by the centroids' bounds (release builds of the reference, the
    default) or by the boxes' bounds (its BUILD_DEBUG builds), lbvh.c:97-104."""
    tracer.check(tracer.lib.rthip_set_morton_bounds(tracer.h, 1 if boxes else 0), "rthip_set_morton_bounds")


def set_counting(tracer: RT_Tracer, enabled: bool) -> None:
    tracer.check(tracer.lib.rthip_set_counting(tracer.h, 1 if enabled else 0), "rthip_set_counting")


def host_alloc(tracer: RT_Tracer, shape, dtype=np.float32) -> np.ndarray:
    """A numpy array in page-locked host memory from rthip_host_alloc: rt_tracer_cast copies frames into it by DMA
    (see include/rthip.h).  The memory lives until host_free or rt_tracer_cleanup."""
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    p = C.c_void_p()
    tracer.check(tracer.lib.rthip_host_alloc(tracer.h, n, C.byref(p)), "rthip_host_alloc")
    buf = (C.c_char * n).from_address(p.value)
    return np.frombuffer(buf, dtype=dtype).reshape(shape)


def host_free(tracer: RT_Tracer, a: np.ndarray) -> None:
    tracer.check(tracer.lib.rthip_host_free(tracer.h, a.ctypes.data), "rthip_host_free")


TRAVERSALS = {"exact": 0, "fast": 1}


def set_traversal(tracer: RT_Tracer, mode: str) -> None:
    """"exact": the reference's fixed left-before-right walk (default, bit-identical results); "fast": 4-wide records,
    nearest box first, a per-lane stack (see include/rthip.h at rthip_set_traversal)."""
    tracer.check(tracer.lib.rthip_set_traversal(tracer.h, TRAVERSALS[mode]), "rthip_set_traversal")


class RT_TracerGroup:
    """Several GPUs of one node behind the five hooks (rthip_group_*): one context per device id, the scene replicated,
    every cast split by tiles (bit-identical to one GPU) or by samples."""

    def __init__(self, device_ids, settings: RT_TracerSettings, shard: str = "tiles"):
        self.lib = abi.load()
        ids = (C.c_int * len(device_ids))(*device_ids)
        g = C.c_void_p()
        rc = self.lib.rthip_group_create(C.byref(g), len(device_ids), ids)
        if rc != 0 or not g.value:
            raise RTHipError(f"rthip_group_create failed with code {rc}")
        self.g = g
        self.shard_mode = {"tiles": 1, "samples": 2}[shard]
        self.check(self.lib.rthip_group_set_tracer(self.g, int(settings.max_bounces), 1 if settings.sky else 0), "rthip_group_set_tracer")

    def check(self, rc: int, what: str):
        if rc != 0:
            msg = self.lib.rthip_group_last_error(self.g)
            raise RTHipError(f"{what}: code {rc}: {msg.decode() if msg else ''}")

    def ctx(self, rank: int) -> RT_Tracer:
        """A borrowed view of one device's context (for taps and per-context settings); do not close it."""
        t = RT_Tracer.__new__(RT_Tracer)
        t.lib = self.lib
        t.h = C.c_void_p(self.lib.rthip_group_ctx(self.g, rank))
        t.close = lambda: None
        return t

    def build_blas(self, world: RT_World) -> None:
        meshes = flatten_meshes(world)
        self.check(self.lib.rthip_group_blas_begin(self.g, len(meshes)), "rthip_group_blas_begin")
        for blas_id, m in enumerate(meshes):
            idx = m["indices"]
            self.check(self.lib.rthip_group_blas_add_mesh(self.g, blas_id, abi.ptr(m["verts"]), m["vertex_count"], m["stride"], m["pos_off"],
                                                          m["nrm_off"], abi.ptr(idx) if idx is not None else None,
                                                          0 if idx is None else idx.size), "rthip_group_blas_add_mesh")
        self.check(self.lib.rthip_group_blas_commit(self.g), "rthip_group_blas_commit")

    def build_tlas(self, world: RT_World) -> None:
        flat = flatten_world(world)
        inst = np.ascontiguousarray(flat.instances, dtype=abi.INSTANCE)
        mats = np.ascontiguousarray(flat.materials, dtype=abi.MATERIAL)
        self.check(self.lib.rthip_group_tlas_build(self.g, abi.ptr(inst), inst.size, abi.ptr(mats), mats.size), "rthip_group_tlas_build")

    def cast(self, settings: np.ndarray, out_radiance: np.ndarray, width: int, height: int, seed: int = 0) -> None:
        cast = np.ascontiguousarray(settings, dtype=abi.CAST).reshape(1)
        assert out_radiance.dtype == np.float32 and out_radiance.size == width * height * 3
        self.check(self.lib.rthip_group_cast(self.g, abi.ptr(cast), int(seed) & (2**64 - 1), abi.ptr(out_radiance), width, height,
                                             self.shard_mode), "rthip_group_cast")

    def cast_device(self, settings: np.ndarray, d_out_ptr: int, width: int, height: int, seed: int = 0, shard: str = None) -> None:
        """rthip_group_cast_device: the frame is left in memory of the group's first device (`d_out_ptr`, W*H*3 floats)."""
        cast = np.ascontiguousarray(settings, dtype=abi.CAST).reshape(1)
        mode = self.shard_mode if shard is None else {"tiles": 1, "samples": 2}[shard]
        self.check(self.lib.rthip_group_cast_device(self.g, abi.ptr(cast), int(seed) & (2**64 - 1), C.c_void_p(d_out_ptr), width, height,
                                                    mode), "rthip_group_cast_device")

    def transport(self) -> str:
        return self.lib.rthip_group_transport(self.g).decode()

    def close(self):
        if getattr(self, "g", None) is not None and self.g.value:
            self.lib.rthip_group_destroy(self.g)
            self.g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
