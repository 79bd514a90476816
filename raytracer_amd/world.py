"""Host-side mirror of the reference's scene API (src/raytracer/raytracer_core.{h,c}).

Same names and the same list semantics as the reference: `rt_world_add_*` push to
the FRONT of the world's lists (raytracer_core.c:21,37,53) and return a handle;
`rt_world_resolve_*` gives the record to fill in.  `flatten_world` turns a world
into the plain arrays the C ABI takes, in list-iteration order (first -> next),
which is the order the reference's backends walk them in
(src/raytracer/cpu/raytracer_cpu.c:108,159).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from . import abi

# GEO_VertexAttributes bits, src/geo/geo.h:4-9
GEO_P, GEO_N, GEO_T, GEO_C = 1, 2, 4, 8


def geo_vertex_size(attrs: int) -> int:
    """geo_vertex_size, src/geo/geo.c:2-4: 16 bytes per present attribute."""
    return bin(attrs).count("1") * 16


def geo_vertex_offset(attrs: int, attr: int) -> int:
    """geo_vertex_offset, src/geo/geo.c:8-11."""
    assert attrs & attr
    return bin(attrs & (attr - 1)).count("1") * 16


@dataclass
class RT_Mesh:  # raytracer_core.h:104-119
    vertices: Optional[np.ndarray] = None  # float32 [vertices_count, 4*popcount(attrs)]
    vertices_count: int = 0
    indices: Optional[np.ndarray] = None  # uint32 [indices_count] or None (auto-index)
    indices_count: int = 0
    attrs: int = GEO_P
    blas_id: int = -1


@dataclass
class RT_Material:  # raytracer_core.h:74-84 (zero-initialised like push_array)
    type: int = abi.MAT_LAMBERTIAN
    albedo: tuple = (0.0, 0.0, 0.0)
    emissive: tuple = (0.0, 0.0, 0.0)
    roughness: float = 0.0
    ior: float = 0.0
    billboard: bool = False


@dataclass
class RT_Instance:  # raytracer_core.h:17-45
    type: int = abi.INST_SPHERE
    material: Optional[RT_Material] = None  # None == zero handle
    sphere_center: tuple = (0.0, 0.0, 0.0)
    sphere_radius: float = 0.0
    mesh: Optional[RT_Mesh] = None
    translation: tuple = (0.0, 0.0, 0.0)
    rotation: tuple = (0.0, 0.0, 0.0, 0.0)
    scale: tuple = (0.0, 0.0, 0.0)


@dataclass
class RT_World:  # raytracer_core.h:139-144; index 0 of each list is `first`
    instances: list = field(default_factory=list)
    materials: list = field(default_factory=list)
    meshes: list = field(default_factory=list)


def rt_make_world() -> RT_World:
    return RT_World()


def rt_world_add_mesh(world: RT_World) -> RT_Mesh:
    m = RT_Mesh()
    world.meshes.insert(0, m)
    return m


def rt_world_add_material(world: RT_World) -> RT_Material:
    m = RT_Material()
    world.materials.insert(0, m)
    return m


def rt_world_add_instance(world: RT_World) -> RT_Instance:
    i = RT_Instance()
    world.instances.insert(0, i)
    return i


def rt_world_resolve_mesh(world, handle):
    return handle


def rt_world_resolve_material(world, handle):
    return handle


def rt_world_resolve_instance(world, handle):
    return handle


@dataclass
class FlatWorld:
    """A world as plain arrays, in list-iteration order."""

    meshes: list  # of dict(verts=float32 2-D, vertex_count, stride, pos_off, nrm_off, indices=uint32|None, attrs)
    instances: np.ndarray  # abi.INSTANCE
    materials: np.ndarray  # abi.MATERIAL


def flatten_meshes(world: RT_World) -> list:
    out = []
    for blas_id, m in enumerate(world.meshes):  # rt_cpu_build_blas, raytracer_cpu.c:107-115
        m.blas_id = blas_id
        verts = np.ascontiguousarray(m.vertices, dtype=np.float32).reshape(m.vertices_count, -1)
        stride = geo_vertex_size(m.attrs)
        assert verts.shape[1] * 4 == stride, "vertex rows must be 16 bytes per attribute"
        idx = None
        if m.indices is not None and m.indices_count:
            idx = np.ascontiguousarray(m.indices, dtype=np.uint32).reshape(-1)[: m.indices_count]
        out.append(
            dict(
                verts=verts,
                vertex_count=int(m.vertices_count),
                stride=stride,
                pos_off=geo_vertex_offset(m.attrs, GEO_P),
                nrm_off=geo_vertex_offset(m.attrs, GEO_N) if (m.attrs & GEO_N) else -1,
                indices=idx,
                attrs=int(m.attrs),
            )
        )
    return out


def flatten_world(world: RT_World) -> FlatWorld:
    meshes = flatten_meshes(world)
    mats = np.zeros(len(world.materials), dtype=abi.MATERIAL)
    mat_index = {}
    for k, m in enumerate(world.materials):
        mat_index[id(m)] = k
        mats[k]["type"] = m.type
        mats[k]["albedo"] = np.asarray(m.albedo, dtype=np.float32)
        mats[k]["emissive"] = np.asarray(m.emissive, dtype=np.float32)
        mats[k]["roughness"] = np.float32(m.roughness)
        mats[k]["ior"] = np.float32(m.ior)
        mats[k]["billboard"] = 1 if m.billboard else 0
    insts = np.zeros(len(world.instances), dtype=abi.INSTANCE)
    for k, it in enumerate(world.instances):  # rt_cpu_build_tlas, raytracer_cpu.c:158-166
        insts[k]["type"] = it.type
        insts[k]["material"] = mat_index[id(it.material)] if it.material is not None else -1
        if it.type == abi.INST_SPHERE:
            insts[k]["sphere_center"] = np.asarray(it.sphere_center, dtype=np.float32)
            insts[k]["sphere_radius"] = np.float32(it.sphere_radius)
        else:
            assert it.mesh is not None and it.mesh.blas_id >= 0, "build_blas must run before build_tlas"
            insts[k]["blas_id"] = it.mesh.blas_id
            insts[k]["translation"] = np.asarray(it.translation, dtype=np.float32)
            insts[k]["rotation"] = np.asarray(it.rotation, dtype=np.float32)
            insts[k]["scale"] = np.asarray(it.scale, dtype=np.float32)
    return FlatWorld(meshes=meshes, instances=insts, materials=mats)
