"""Scene data of the reference's demos and of the benchmark configurations, built
through the host-side mirror of the reference's scene API (world.py).

All arithmetic that the reference performs in binary32 on the host (quad
rotations, box centres, camera derivation) is done here in numpy float32 scalars,
one IEEE operation at a time and in the reference's order, so the instance and
camera records are bit-identical to the ones the reference's demos build.
tanf comes from the host libm, as it does for the reference.
"""
from __future__ import annotations

import ctypes
import ctypes.util
import math

import numpy as np

from . import abi
from .world import (GEO_N, GEO_P, RT_World, rt_make_world, rt_world_add_instance, rt_world_add_material,
                    rt_world_add_mesh)

f32 = np.float32
PI_F32 = f32(3.1415926535897932384626433832795028841971693993751058209749445923078164062)
EPSILON_F32 = f32(1.1920929e-7)

_libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
_libm.tanf.restype = ctypes.c_float
_libm.tanf.argtypes = [ctypes.c_float]


def tanf(x) -> np.float32:
    return f32(_libm.tanf(ctypes.c_float(float(x))))


# ---- binary32 vector helpers in the reference's operation order (src/common/common_math.c) ----
def v3(x, y, z):
    return (f32(x), f32(y), f32(z))


def add3(a, b):
    return (a[0] + b[0], a[1] + b[1], a[2] + b[2])


def sub3(a, b):
    return (a[0] - b[0], a[1] - b[1], a[2] - b[2])


def mul3(a, s):
    return (a[0] * s, a[1] * s, a[2] * s)


def elmul3(a, b):
    return (a[0] * b[0], a[1] * b[1], a[2] * b[2])


def dot3(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def length3(a):
    return np.sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2])


def normalize3(a):
    l = length3(a)
    return (a[0] / l, a[1] / l, a[2] / l)


def cross3(a, b):  # common_math.c:29
    return (a[1] * b[2] - a[2] * b[1], -(a[0] * b[2] - a[2] * b[0]), a[0] * b[1] - a[1] * b[0])


def min3(a, b):
    return tuple(x if x < y else y for x, y in zip(a, b))


def max3(a, b):
    return tuple(x if x > y else y for x, y in zip(a, b))


def orthogonal3(x):  # common_math.c:37-46
    a = tuple(abs(c) for c in x)
    mx = f32(1.0 if (a[0] <= a[1] and a[0] <= a[2]) else 0.0)
    my = f32(1.0 if (a[1] < a[0] and a[1] <= a[2]) else 0.0)
    mz = f32(1.0) - mx - my
    return cross3(x, (mx, my, mz))


def make_a_to_b_quat(a, b):  # common_math.c:92-100
    k = np.sqrt(dot3(a, a) * dot3(b, b))
    k_cos_theta = dot3(a, b)
    with np.errstate(all="ignore"):
        if k_cos_theta / k < f32(-1.0) + EPSILON_F32:
            o = normalize3(orthogonal3(a))
            return (o[0], o[1], o[2], f32(0.0))
    c = cross3(a, b)
    w = k_cos_theta + k
    l = np.sqrt(c[0] * c[0] + c[1] * c[1] + c[2] * c[2] + w * w)
    return (c[0] / l, c[1] / l, c[2] / l, w / l)


IDENTITY_QUAT = (f32(0), f32(0), f32(0), f32(1))


def degrees_to_rad(v: int) -> np.float32:
    """DegreesToRad, common_math.h:289: (v * (PI_F32 / 180.f)) with an int v."""
    return f32(v) * (PI_F32 / f32(180.0))


def get_rt_cast_settings(width, height, samples, eye, subject, defocus_angle=0.0, orthographic=False, vfov=0.0):
    """get_rt_cast_settings, src/demos/demos_inc.c:104-130 -> one abi.CAST record."""
    eye = v3(*eye)
    subject = v3(*subject)
    d = sub3(subject, eye)
    up = v3(0, 1, 0)
    forward = normalize3(d)
    right = cross3(forward, up)
    aspect = f32(width) / f32(height)
    vfov = f32(vfov) if f32(vfov) > 0 else degrees_to_rad(45)
    focus = length3(d)
    h = tanf(vfov / f32(2.0))
    plane_h = f32(2) * h * focus
    defocus_angle = f32(defocus_angle)
    defocus_radius = focus * tanf(defocus_angle / f32(2.0))
    c = np.zeros(1, dtype=abi.CAST)
    c[0]["eye"] = eye
    c[0]["up"] = up
    c[0]["forward"] = forward
    c[0]["right"] = right
    c[0]["viewport"] = (plane_h * aspect, plane_h, focus)
    c[0]["samples"] = samples
    c[0]["ior"] = 1.0
    c[0]["defocus"] = 1 if defocus_angle > 0 else 0
    c[0]["defocus_disk"] = (defocus_radius, defocus_radius)
    c[0]["orthographic"] = 1 if orthographic else 0
    return c


# ---- demo scenes ---------------------------------------------------------------


def _set_mesh(m, verts, indices, attrs):
    m.vertices = np.ascontiguousarray(verts, dtype=np.float32)
    m.vertices_count = m.vertices.shape[0]
    if indices is not None:
        m.indices = np.ascontiguousarray(indices, dtype=np.uint32).reshape(-1)
        m.indices_count = m.indices.size
    m.attrs = attrs


def cornell(extra_mesh=None):
    """src/demos/cornell/main.c:79-226.  Returns (world, camera(width, height, samples), sky).

    extra_mesh = (verts, indices) of a PN mesh centred near (-0.3, 1.2, 0) with radius about 1 (the
    bunny-class mesh) adds one instance of it, scaled by 80, standing on the short box: the 4K
    multi-GPU configuration.  Without it the scene is exactly the demo's."""
    world = rt_make_world()
    box_v = np.array([[-1, -1, -1, 0], [1, -1, -1, 0], [1, 1, -1, 0], [1, -1, 1, 0], [1, 1, 1, 0], [-1, 1, -1, 0],
                      [-1, 1, 1, 0], [-1, -1, 1, 0]], dtype=np.float32)
    box_i = np.array([0, 1, 2, 0, 2, 5, 7, 3, 4, 7, 4, 6, 1, 3, 4, 1, 4, 2, 0, 5, 6, 0, 6, 7, 5, 2, 4, 5, 4, 6, 0, 7,
                      3, 0, 3, 1], dtype=np.uint32)
    quad_v = np.array([[-1, 0, -1, 0], [1, 0, -1, 0], [1, 0, 1, 0], [-1, 0, 1, 0]], dtype=np.float32)
    quad_i = np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32)
    box = rt_world_add_mesh(world)
    _set_mesh(box, box_v, box_i, GEO_P)
    quad = rt_world_add_mesh(world)
    _set_mesh(quad, quad_v, quad_i, GEO_P)

    def lambertian(albedo):
        m = rt_world_add_material(world)
        m.type, m.albedo, m.billboard = abi.MAT_LAMBERTIAN, albedo, True
        return m

    ext = f32(277.5)
    white = lambertian(v3(0.73, 0.73, 0.73))
    red = lambertian(v3(0.65, 0.05, 0.05))
    green = lambertian(v3(0.12, 0.45, 0.15))
    light = rt_world_add_material(world)
    light.type, light.emissive = abi.MAT_LIGHT, v3(16, 16, 16)
    metal = rt_world_add_material(world)  # created by the demo, used by nothing
    metal.type, metal.roughness, metal.billboard = abi.MAT_METAL, f32(0.01), True

    def add_quad(mat, c, u, v):  # cornell/main.c:33-48
        n = cross3(v, u)
        it = rt_world_add_instance(world)
        it.type, it.material, it.mesh = abi.INST_MESH, mat, quad
        it.translation = c
        it.rotation = make_a_to_b_quat(v3(0, 1, 0), n)
        it.scale = (length3(u), f32(1), length3(v))

    def add_box(mat, a, b):  # cornell/main.c:50-67
        mn, mx = min3(a, b), max3(a, b)
        it = rt_world_add_instance(world)
        it.type, it.material, it.mesh = abi.INST_MESH, mat, box
        it.translation = mul3(add3(mn, mx), f32(0.5))
        it.rotation = IDENTITY_QUAT
        it.scale = mul3(sub3(mx, mn), f32(0.5))

    z = f32(0)
    add_quad(white, (z, -ext, z), (ext, z, z), (z, z, ext))  # floor
    add_quad(white, (z, ext, z), (ext, z, z), (z, z, ext))  # ceiling
    add_quad(white, (z, z, -ext), (ext, z, z), (z, ext, z))  # back wall
    add_quad(green, (-ext, z, z), (z, z, ext), (z, ext, z))  # left wall
    add_quad(red, (ext, z, z), (z, z, ext), (z, ext, z))  # right wall
    add_quad(light, (z, ext - f32(1), z), (ext / f32(4), z, z), (z, z, ext / f32(4)))  # light
    flip = v3(-1, 1, -1)
    e3 = (ext, ext, ext)
    add_box(white, elmul3(sub3(v3(130, 0, 65), e3), flip), elmul3(sub3(v3(295, 165, 230), e3), flip))
    add_box(white, elmul3(sub3(v3(265, 0, 295), e3), flip), elmul3(sub3(v3(430, 330, 460), e3), flip))

    if extra_mesh is not None:
        verts, indices = extra_mesh
        mesh = rt_world_add_mesh(world)
        _set_mesh(mesh, verts, indices, GEO_P | GEO_N)
        it = rt_world_add_instance(world)
        it.type, it.material, it.mesh = abi.INST_MESH, metal, mesh
        s80 = f32(80)
        # the mesh's centre (-0.3, 1.2, 0) goes to (65, -112.5 + 85, 130): above the middle of the short box's top face
        it.translation = (f32(65) + f32(0.3) * s80, f32(-27.5) - f32(1.2) * s80, f32(130))
        it.rotation = IDENTITY_QUAT
        it.scale = (s80, s80, s80)

    def camera(width, height, samples):
        return get_rt_cast_settings(width, height, samples, (z, z, f32(2.4) * ext), (0, 0, 0), vfov=degrees_to_rad(40))

    return world, camera, False


def tri():
    """src/demos/tri/main.c:4-62 (auto-indexed single triangle, orthographic camera, sky on)."""
    world = rt_make_world()
    s = np.sqrt(f32(3.0)) / f32(4.0)
    verts = np.array([[-0.5, -s, 0, 0], [0.5, -s, 0, 0], [0.0, s, 0, 0]], dtype=np.float32)
    m = rt_world_add_mesh(world)
    _set_mesh(m, verts, None, GEO_P)
    mat = rt_world_add_material(world)
    mat.type, mat.albedo, mat.billboard = abi.MAT_LAMBERTIAN, v3(0.5, 0.5, 0.5), True
    it = rt_world_add_instance(world)
    it.type, it.material, it.mesh = abi.INST_MESH, mat, m
    it.translation, it.rotation, it.scale = v3(0, 0, 0), IDENTITY_QUAT, v3(1, 1, 1)

    def camera(width, height, samples):
        return get_rt_cast_settings(width, height, samples, (0, 0, 1), (0, 0, 0), orthographic=True)

    return world, camera, True


def spheres(defocus=True):
    """src/demos/spheres/main.c:40-72 (five spheres, no meshes, defocus camera, sky on)."""
    world = rt_make_world()

    def add_sphere(center, radius):
        mat = rt_world_add_material(world)
        it = rt_world_add_instance(world)
        it.type, it.material = abi.INST_SPHERE, mat
        it.sphere_center, it.sphere_radius = v3(*center), f32(radius)
        return mat

    m = add_sphere((0, 0, 0), 1.0)
    m.type, m.ior = abi.MAT_DIELECTRIC, f32(1.52)
    m = add_sphere((0, 0, 0), 0.8)
    m.type, m.albedo = abi.MAT_LAMBERTIAN, v3(0.8, 0.5, 0.5)
    m = add_sphere((-1.6, 0, 0), 0.5)
    m.type, m.roughness = abi.MAT_METAL, f32(0.1)
    m = add_sphere((1.6, 0, 0), 0.5)
    m.type, m.roughness = abi.MAT_METAL, f32(0.9)
    m = add_sphere((0, -101, 0), 100.0)
    m.type, m.albedo = abi.MAT_LAMBERTIAN, v3(0.5, 0.6, 0.5)

    def camera(width, height, samples):
        return get_rt_cast_settings(width, height, samples, (0, 0, -2.5), (0, 0, 0),
                                    defocus_angle=degrees_to_rad(5) if defocus else 0.0)

    return world, camera, True


# ---- stand-in for data/bunny.obj (absent from the reference checkout) ----------


def _icosphere(subdiv: int):
    t = (1.0 + math.sqrt(5.0)) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t), (t, 0, -1),
         (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7),
         (9, 8, 1)]
    verts = np.array(v, dtype=np.float64)
    verts /= np.sqrt((verts * verts).sum(axis=1))[:, None]
    faces = np.array(f, dtype=np.int64)
    for _ in range(subdiv):
        a, b, c = faces[:, 0], faces[:, 1], faces[:, 2]
        edges = np.concatenate([np.stack([a, b], 1), np.stack([b, c], 1), np.stack([c, a], 1)], axis=0)
        key = np.minimum(edges[:, 0], edges[:, 1]) * (1 << 32) + np.maximum(edges[:, 0], edges[:, 1])
        uniq, inverse = np.unique(key, return_inverse=True)
        lo, hi = uniq >> 32, uniq & ((1 << 32) - 1)
        mid = verts[lo] + verts[hi]
        mid /= np.sqrt((mid * mid).sum(axis=1))[:, None]
        base = verts.shape[0]
        verts = np.concatenate([verts, mid], axis=0)
        nf = faces.shape[0]
        ab, bc, ca = base + inverse[:nf], base + inverse[nf:2 * nf], base + inverse[2 * nf:]
        faces = np.concatenate([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1), np.stack([c, ca, bc], 1),
                                np.stack([ab, bc, ca], 1)], axis=0)
    return verts, faces


def bunny_class_mesh(subdiv: int = 6):
    """Deterministic 'bunny-class' mesh: an icosphere (subdiv 6 -> 40 962 vertices, 81 920
    triangles) with a smooth radial displacement, centred at (-0.3, 1.2, 0), radius ~1, with
    area-weighted vertex normals.  Only +,-,*,/ and sqrt are used, so the float32 result is
    the same on every machine.  Returns (verts float32 [V, 8] = P,pad,N,pad ; indices uint32)."""
    p, faces = _icosphere(subdiv)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    bump = (0.10 * (x * y + y * z * z - x * z) + 0.05 * (4 * x * x * x - 3 * x) * (2 * y * y - 1)
            + 0.03 * (8 * z * z * z * z - 8 * z * z + 1) * (2 * x * y))
    p = p * (1.0 + bump)[:, None] + np.array([-0.3, 1.2, 0.0])
    p32 = p.astype(np.float32)
    pf = p32.astype(np.float64)
    e1 = pf[faces[:, 1]] - pf[faces[:, 0]]
    e2 = pf[faces[:, 2]] - pf[faces[:, 0]]
    fn = np.stack([e1[:, 1] * e2[:, 2] - e1[:, 2] * e2[:, 1], e1[:, 2] * e2[:, 0] - e1[:, 0] * e2[:, 2],
                   e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0]], axis=1)
    vn = np.zeros_like(pf)
    for k in range(3):
        for c in range(3):
            vn[:, c] += np.bincount(faces[:, k], weights=fn[:, c], minlength=pf.shape[0])
    vn /= np.sqrt((vn * vn).sum(axis=1))[:, None]
    verts = np.zeros((p32.shape[0], 8), dtype=np.float32)
    verts[:, 0:3] = p32
    verts[:, 4:7] = vn.astype(np.float32)
    return verts, faces.astype(np.uint32).reshape(-1)


def write_obj(path: str, verts: np.ndarray, indices: np.ndarray) -> None:
    """Writes the `v` / `vn` / `f a//a` flavour that the reference's PN loader reads (src/mesh/mesh.c:162-166)."""
    with open(path, "w") as f:
        for r in verts:
            f.write("v %.9g %.9g %.9g\n" % (r[0], r[1], r[2]))
        for r in verts:
            f.write("vn %.9g %.9g %.9g\n" % (r[4], r[5], r[6]))
        tri = indices.reshape(-1, 3) + 1
        for a, b, c in tri:
            f.write("f %d//%d %d//%d %d//%d\n" % (a, a, b, b, c, c))


def bunny(verts=None, indices=None):
    """src/demos/bunny/main.c:39-78 with the stand-in mesh (one PN mesh, one instance, sky on)."""
    if verts is None:
        verts, indices = bunny_class_mesh()
    world = rt_make_world()
    m = rt_world_add_mesh(world)
    _set_mesh(m, verts, indices, GEO_P | GEO_N)
    mat = rt_world_add_material(world)
    mat.type, mat.albedo, mat.billboard = abi.MAT_LAMBERTIAN, v3(0.73, 0.73, 0.73), True
    it = rt_world_add_instance(world)
    it.type, it.material, it.mesh = abi.INST_MESH, mat, m
    it.translation, it.rotation, it.scale = v3(0, 0, 0), IDENTITY_QUAT, v3(1, 1, 1)

    def camera(width, height, samples):
        return get_rt_cast_settings(width, height, samples, (-0.3, 1.2, 3.5), (-0.3, 1.2, 0), vfov=degrees_to_rad(45))

    return world, camera, True


# ---- counter-based host randomness for the synthetic configurations -------------


def _mix64(z: np.ndarray) -> np.ndarray:
    z = z.astype(np.uint64)
    with np.errstate(over="ignore"):
        z ^= z >> np.uint64(30)
        z *= np.uint64(0xBF58476D1CE4E5B9)
        z ^= z >> np.uint64(27)
        z *= np.uint64(0x94D049BB133111EB)
        z ^= z >> np.uint64(31)
    return z


def hash_unit(seed: int, stream: int, n: int, start: int = 0) -> np.ndarray:
    """n float64 values in [0,1) from a counter hash (24 random bits each: exact in float32)."""
    with np.errstate(over="ignore"):
        idx = np.arange(start, start + n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)
        key = _mix64(np.uint64(seed) * np.uint64(0xD1342543DE82EF95) + np.uint64(stream))
        bits = _mix64(idx + key) >> np.uint64(40)
    return bits.astype(np.float64) / float(1 << 24)


def instanced_grid(n_side: int = 100, seed: int = 1, verts=None, indices=None):
    """Configuration 3: the stand-in mesh instanced n_side x n_side times on an XZ grid (spacing
    2.5), per-instance yaw and uniform scale in [0.5, 1], three materials cycling."""
    if verts is None:
        verts, indices = bunny_class_mesh()
    world = rt_make_world()
    m = rt_world_add_mesh(world)
    _set_mesh(m, verts, indices, GEO_P | GEO_N)
    mats = []
    for kind in range(3):
        mat = rt_world_add_material(world)
        if kind == 0:
            mat.type, mat.albedo, mat.billboard = abi.MAT_LAMBERTIAN, v3(0.73, 0.73, 0.73), True
        elif kind == 1:
            mat.type, mat.albedo, mat.billboard = abi.MAT_LAMBERTIAN, v3(0.65, 0.05, 0.05), True
        else:
            mat.type, mat.roughness, mat.billboard = abi.MAT_METAL, f32(0.1), True
        mats.append(mat)
    n = n_side * n_side
    yaw_c = hash_unit(seed, 1, n) * 2.0 - 1.0  # cos-like parameter in [-1,1)
    scl = 0.5 + 0.5 * hash_unit(seed, 2, n)
    half = (n_side - 1) * 2.5 / 2.0
    for k in range(n):
        gx, gz = k % n_side, k // n_side
        # yaw quaternion (0, sin(a/2), 0, cos(a/2)) from c = cos(a/2) in [-1,1): s = sqrt(1-c^2)
        c = f32(yaw_c[k])
        s = np.sqrt(f32(1.0) - c * c)
        it = rt_world_add_instance(world)
        it.type, it.material, it.mesh = abi.INST_MESH, mats[k % 3], m
        sc = f32(scl[k])
        it.translation = v3(gx * 2.5 - half + 0.3 * float(sc), -1.2 * float(sc), gz * 2.5 - half)
        it.rotation = (f32(0), s, f32(0), c)
        it.scale = (sc, sc, sc)

    def camera(width, height, samples):
        return get_rt_cast_settings(width, height, samples, (0.0, half * 0.35, half * 1.15), (0, 0, 0),
                                    vfov=degrees_to_rad(45))

    return world, camera, True


def triangle_soup(n_tris: int, seed: int = 1, chunk: int = 1 << 20):
    """Configuration 4: n_tris small triangles, centroid uniform in [0,1)^3, vertices = centroid
    + U(-0.001, 0.001)^3, auto-indexed.  Returns float32 [3*n_tris, 4] (P + pad)."""
    out = np.zeros((3 * n_tris, 4), dtype=np.float32)
    for start in range(0, n_tris, chunk):
        n = min(chunk, n_tris - start)
        cen = np.stack([hash_unit(seed, 10 + a, n, start) for a in range(3)], axis=1)
        for vtx in range(3):
            off = np.stack([hash_unit(seed, 20 + 3 * vtx + a, n, start) for a in range(3)], axis=1)
            out[3 * start + vtx:3 * (start + n):3, 0:3] = (cen + (off * 2.0 - 1.0) * 0.001).astype(np.float32)
    return out


def soup_world(n_tris: int, seed: int = 1):
    world = rt_make_world()
    m = rt_world_add_mesh(world)
    _set_mesh(m, triangle_soup(n_tris, seed), None, GEO_P)
    mat = rt_world_add_material(world)
    mat.type, mat.albedo, mat.billboard = abi.MAT_LAMBERTIAN, v3(0.73, 0.73, 0.73), True
    it = rt_world_add_instance(world)
    it.type, it.material, it.mesh = abi.INST_MESH, mat, m
    it.translation, it.rotation, it.scale = v3(0, 0, 0), IDENTITY_QUAT, v3(1, 1, 1)

    def camera(width, height, samples):
        return get_rt_cast_settings(width, height, samples, (0.5, 0.5, 2.6), (0.5, 0.5, 0.5), vfov=degrees_to_rad(45))

    return world, camera, True


def heightfield_world(n_side: int = 2236, seed: int = 1):
    """Configuration 4's coherent variant: a displaced height field over [0,1]^2, n_side x n_side cells of two triangles
    each (2236^2 x 2 = 10.0 M), one indexed mesh without normals.  Heights are a sum of five seeded sine octaves
    (amplitude halving, frequency doubling), about +-0.12 around y = 0.2: a connected surface whose Morton order follows
    its triangle order, unlike the soup."""
    k = np.arange(n_side + 1, dtype=np.float64) / n_side
    x, z = np.meshgrid(k, k, indexing="xy")
    y = np.full_like(x, 0.2)
    ph = hash_unit(seed, 70, 20) * 2.0 * math.pi
    for o in range(5):
        f = 2.0 * math.pi * (1.5 * 2 ** o)
        y += (0.06 / 2 ** o) * np.sin(f * x + ph[4 * o]) * np.sin(f * z + ph[4 * o + 1])
        y += (0.03 / 2 ** o) * np.sin(0.7 * f * (x + z) + ph[4 * o + 2]) * np.cos(1.3 * f * (x - z) + ph[4 * o + 3])
    verts = np.zeros(((n_side + 1) ** 2, 4), dtype=np.float32)
    verts[:, 0], verts[:, 1], verts[:, 2] = x.reshape(-1), y.reshape(-1), z.reshape(-1)
    del x, y, z
    row = np.arange(n_side, dtype=np.uint32)
    v00 = (row[:, None] * np.uint32(n_side + 1) + row[None, :]).reshape(-1)  # cell (row r, column c): vertex r * (n + 1) + c
    idx = np.empty((n_side * n_side, 6), dtype=np.uint32)
    idx[:, 0], idx[:, 1], idx[:, 2] = v00, v00 + np.uint32(n_side + 1), v00 + np.uint32(1)
    idx[:, 3], idx[:, 4], idx[:, 5] = v00 + np.uint32(1), v00 + np.uint32(n_side + 1), v00 + np.uint32(n_side + 2)
    world = rt_make_world()
    m = rt_world_add_mesh(world)
    _set_mesh(m, verts, idx.reshape(-1), GEO_P)
    mat = rt_world_add_material(world)
    mat.type, mat.albedo, mat.billboard = abi.MAT_LAMBERTIAN, v3(0.73, 0.73, 0.73), True
    it = rt_world_add_instance(world)
    it.type, it.material, it.mesh = abi.INST_MESH, mat, m
    it.translation, it.rotation, it.scale = v3(0, 0, 0), IDENTITY_QUAT, v3(1, 1, 1)

    def camera(width, height, samples):
        return get_rt_cast_settings(width, height, samples, (0.5, 0.75, 0.95), (0.5, 0.2, 0.5), vfov=degrees_to_rad(45))

    return world, camera, True


def random_rays(n: int, seed: int, lo=(0.0, 0.0, 0.0), hi=(1.0, 1.0, 1.0)):
    """'Incoherent' rays: origin uniform in the box [lo,hi], direction uniform on the sphere
    (rejection-free: z uniform, azimuth by a rational parametrisation), float32, unit length."""
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    o = np.stack([hash_unit(seed, 40 + a, n) for a in range(3)], axis=1) * (hi - lo) + lo
    zc = hash_unit(seed, 50, n) * 2.0 - 1.0
    # azimuth: point on the unit circle from t in [-1,1) via (1-t^2, 2t)/(1+t^2), quadrant by two sign bits
    t = hash_unit(seed, 51, n) * 2.0 - 1.0
    cx, cy = (1 - t * t) / (1 + t * t), (2 * t) / (1 + t * t)
    flip = hash_unit(seed, 52, n) < 0.5
    cx = np.where(flip, -cx, cx)
    r = np.sqrt(np.maximum(0.0, 1.0 - zc * zc))
    d = np.stack([r * cx, r * cy, zc], axis=1).astype(np.float32)
    d64 = d.astype(np.float64)
    d = (d64 / np.sqrt((d64 * d64).sum(axis=1))[:, None]).astype(np.float32)
    return o.astype(np.float32), d


def incoherent_rays_device(torch, n: int, seed: int, lo=(0.0, 0.0, 0.0), hi=(1.0, 1.0, 1.0)):
    """The benchmark's incoherent ray stream, generated on the GPU (torch's generator): origins uniform in the box
    [lo, hi], directions uniform on the sphere (normalised Gaussians).  Returns two contiguous float32 [n, 3] tensors."""
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    lo_t, hi_t = torch.tensor(lo, device="cuda"), torch.tensor(hi, device="cuda")
    o = (lo_t + torch.rand((n, 3), device="cuda", generator=g) * (hi_t - lo_t)).contiguous()
    v = torch.randn((n, 3), device="cuda", generator=g)
    d = (v / v.norm(dim=1, keepdim=True)).contiguous()
    return o, d


# ---- awkward inputs for the parity tests -----------------------------------------------------


def edge_scene():
    """A small world with awkward geometry: a degenerate (zero-area) triangle, two coincident triangles,
    a sliver, a triangle in an axis plane, auto-indexed and indexed meshes, a scaled + rotated instance and
    a sphere."""
    w = rt_make_world()
    v = np.array([[0, 0, 0, 0], [1, 0, 0, 0], [0, 1, 0, 0],      # plain triangle in the z = 0 plane
                  [0, 0, 0, 0], [1, 0, 0, 0], [0, 1, 0, 0],      # the same triangle again (coincident)
                  [2, 2, 2, 0], [2, 2, 2, 0], [2, 2, 2, 0],      # a point (zero area)
                  [0, 0, 1, 0], [3, 0, 1, 0], [1.5, 1e-6, 1, 0],  # a sliver
                  [0, 0, -1, 0], [0, 2, -1, 0], [0, 0, -3, 0]],   # in the x = 0 plane
                 dtype=np.float32)
    m0 = rt_world_add_mesh(w)
    _set_mesh(m0, v, None, GEO_P)  # auto-indexed
    quad_v = np.array([[-1, 0, -1, 0], [1, 0, -1, 0], [1, 0, 1, 0], [-1, 0, 1, 0]], dtype=np.float32)
    m1 = rt_world_add_mesh(w)
    _set_mesh(m1, quad_v, np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32), GEO_P)
    mat = rt_world_add_material(w)
    mat.type, mat.albedo = abi.MAT_LAMBERTIAN, v3(0.5, 0.5, 0.5)
    it = rt_world_add_instance(w)
    it.type, it.material, it.mesh = abi.INST_MESH, mat, m0
    it.translation, it.rotation, it.scale = v3(0, 0, 0), IDENTITY_QUAT, v3(1, 1, 1)
    it = rt_world_add_instance(w)
    it.type, it.material, it.mesh = abi.INST_MESH, mat, m1
    it.translation = v3(0.5, -0.5, 0.25)
    it.rotation = make_a_to_b_quat(v3(0, 1, 0), v3(1, 1, 0))
    it.scale = v3(2, 1, 0.5)
    it = rt_world_add_instance(w)
    it.type, it.material = abi.INST_SPHERE, mat
    it.sphere_center, it.sphere_radius = v3(-2, 0, 0), np.float32(0.75)
    return w


def edge_rays():
    """Rays with exact zeros in their directions (1/0 = inf, 0 * inf = NaN in the slab test), origins exactly
    on box planes and on triangle edges / vertices, and a few generic ones."""
    o, d = [], []
    axes = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1), (1, 1, 0), (0, 1, 1), (1, 0, -1)]
    pts = [(0, 0, 0), (0.25, 0.25, 5), (0.25, 0.25, -5), (1, 0, 5), (0, 1, 5), (0.5, 0.5, 5), (0.5, 0, 5), (5, 0.25, 0.25),
           (-5, 0, 0), (-2, 5, 0), (-2, 0.75, 5), (0, 5, -2), (2, 2, 5), (1.5, 5e-7, 5), (0.5, -0.5, 5), (0.5, 5, 0.25),
           (3, 0, 1), (-1.25, 0, 0), (0, 0, 1), (0.3, 0.3, 0.3)]
    for p in pts:
        for a in axes:
            o.append(p)
            d.append(a)
    rng = np.random.default_rng(5)
    o += list(rng.uniform(-4, 4, (400, 3)))
    dd = rng.normal(size=(400, 3))
    d += list(dd / np.linalg.norm(dd, axis=1, keepdims=True))
    return np.asarray(o, dtype=np.float32), np.asarray(d, dtype=np.float32)


def hdr_test_images(seed: int = 7):
    """Small float images that exercise every branch of the .hdr writer (stb_image_write.h:639-757): the widths at which
    run-length coding switches on, runs and literal stretches around the packet limits (127 / 128), and the pixel values
    at which the RGBE conversion changes behaviour (the 1e-32 cut-off, negative, huge, infinite and NaN components).
    Returns {name: float32[h, w, 3]}."""
    rng = np.random.default_rng(seed)
    out = {}
    for w in (1, 7, 8, 9, 127, 128, 129, 130, 257, 300):
        out[f"rand_{w}"] = (rng.random((3, w, 3), dtype=np.float32) * np.float32(rng.choice([1e-3, 1.0, 100.0, 1e5]))).astype(np.float32)
        out[f"steps_{w}"] = ((rng.integers(0, 3, (3, w, 1)) * 0.25).astype(np.float32) * np.ones((1, 1, 3), np.float32)).astype(np.float32)
    img = np.zeros((2, 1500, 3), np.float32)
    x = 0
    for run in (1, 2, 3, 4, 126, 127, 128, 129, 254, 255, 256):
        img[:, x:x + run] = np.float32(rng.random() * 5)
        x += run
    out["runs"] = img
    img = rng.random((4, 700, 3), dtype=np.float32)
    img[1, 100:500] = img[1, 100]      # one long run inside noise
    img[2, :, 1] = 0.5                 # one constant component
    img[3] = 0.0                       # a black scanline
    out["mixed"] = img
    sp = np.array([0, 1e-33, 1e-32, 1.0000001e-32, 1e-38, 1e-45, -1.0, -0.25, 1.0, 0.5, 255.0, 256.0, 1e10, 3.4e38,
                   np.inf, -np.inf, np.nan, 65504.0, 0.99999994, 2.0 ** -100], np.float32)
    out["special"] = np.stack(np.meshgrid(sp, sp, sp, indexing="ij"), -1).reshape(20, 400, 3).astype(np.float32)
    return out


def obj_test_files(seed: int = 3):
    """OBJ texts for the loader tests -> {name: (text, attrs)}: the four face syntaxes the reference reads
    (src/mesh/mesh.c:151-172), shared and repeated index triples, a quad (its first three corners count), comments,
    blank and foreign lines, CRLF line ends, an indented record (not a record for the reference), exponent notation, a
    position with a missing coordinate; and one file of more than a megabyte, which the loader splits over threads."""
    rng = np.random.default_rng(seed)
    out = {}
    verts, idx = bunny_class_mesh(2)
    lines = ["# stand-in mesh", "o mesh"]
    lines += ["v %.9g %.9g %.9g" % (r[0], r[1], r[2]) for r in verts]
    lines += ["vn %.9g %.9g %.9g" % (r[4], r[5], r[6]) for r in verts]
    lines += ["f %d//%d %d//%d %d//%d" % (a, a, b, b, c, c) for a, b, c in idx.reshape(-1, 3) + 1]
    out["pn_mesh"] = ("\n".join(lines) + "\n", 3)
    cube_v = [(x, y, z) for x in (0, 1) for y in (0, 1) for z in (0, 1)]
    cube_n = [(1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)]
    cube_t = [(0, 0), (1, 0), (1, 1), (0, 1)]
    quads = [((4, 5, 7, 6), 0), ((0, 2, 3, 1), 1), ((2, 6, 7, 3), 2), ((0, 1, 5, 4), 3), ((1, 3, 7, 5), 4), ((0, 4, 6, 2), 5)]
    lines = ["mtllib none.mtl", "g cube", ""]
    lines += ["v %g %g %g" % v for v in cube_v] + ["vt %g %g" % t for t in cube_t] + ["vn %g %g %g" % n for n in cube_n]
    lines += ["usemtl m", "s off"]
    for q, n in quads:
        lines.append("f %d/1/%d %d/2/%d %d/3/%d" % (q[0] + 1, n + 1, q[1] + 1, n + 1, q[2] + 1, n + 1))
        lines.append("f %d/1/%d %d/3/%d %d/4/%d" % (q[0] + 1, n + 1, q[2] + 1, n + 1, q[3] + 1, n + 1))
    out["pnt_cube"] = ("\n".join(lines) + "\n", 7)
    out["pnt_cube_default_attrs"] = (out["pnt_cube"][0], 0)
    lines = ["v 0 0 0", "v 1e0 0 0", "v 1 1.5e-1 -0", " v 9 9 9", "v 0.25 1", "# a comment", "", "v -1.25E+1 3 4", "vt 0.5 0.5",
             "f 1 2 3", "f 2 3 4 5", "f 5 4 1", "f 1 2 3", "g tail", "f 3 2 1"]
    out["p_mixed_crlf"] = ("\r\n".join(lines) + "\r\n", 1)
    lines = ["v 0 0 0", "v 1 0 0", "v 0 1 0", "v 1 1 0", "vt 0 0", "vt 1 0", "vt 0 1", "f 1/1 2/2 3/3", "f 2/2 4/1 3/3", "f 1/2 2/2 3/3"]
    out["pt_strip"] = ("\n".join(lines), 5)  # no newline at the end of the file
    forms = ["+1.5", "-.5", "5.", "1e3", "1E-2", "0x10", "inf", "-inf", "1e-46", "1e-40", "3.4028236e38", "1e39", "0.1",
             "16777217", "0.3333333333333333333333333", "-0", "0x1.8p1", "1e+2", "00012", "4.9406564584124654e-324",
             "1.17549435e-38", "7.00649232e-46", "9007199254740993", "1.00000005960464477539062500001"]
    lines = ["v %s %s %s" % (forms[k], forms[(k + 5) % len(forms)], forms[(k + 11) % len(forms)]) for k in range(len(forms))]
    lines += ["f %d %d %d" % (k + 1, (k + 1) % len(forms) + 1, (k + 2) % len(forms) + 1) for k in range(len(forms))]
    lines += ["f +1 002 3", "f 4294967297 2 3"]  # a sign, leading zeros, a value that wraps to 1 in 32 bits
    out["p_number_forms"] = ("\n".join(lines) + "\n", 1)
    n_v, n_f = 9000, 42000
    pos = rng.random((n_v, 3)).astype(np.float32)
    nrm = rng.standard_normal((500, 3)).astype(np.float32)
    lines = ["v %.9g %.9g %.9g" % tuple(r) for r in pos] + ["vn %.9g %.9g %.9g" % tuple(r) for r in nrm]
    fp = rng.integers(1, n_v + 1, (n_f, 3))
    fn = rng.integers(1, 4, (n_f, 3))  # few normals per position: many repeated pairs
    lines += ["f %d//%d %d//%d %d//%d" % (a, x, b, y, c, z) for (a, b, c), (x, y, z) in zip(fp, fn)]
    out["pn_random_big"] = ("\n".join(lines) + "\n", 3)
    return out


def random_world(seed: int, coincident: bool = True):
    """A seeded random world for fuzzing the parity tests: 1-3 small meshes (indexed or auto-indexed, with or
    without normals, some triangles degenerate and, with `coincident`, some duplicated or sharing their box --
    equal Morton codes, whose order the reference's unstable sort leaves open), 2-12 instances (meshes under random rotations,
    translations and non-uniform scales, and spheres, some overlapping), every material type with random
    parameters (and a few instances with no material), and a random camera (perspective, orthographic, or with
    defocus).  Returns (world, cast_settings_fn(width, height, samples), sky, max_bounces)."""
    rng = np.random.default_rng(seed)
    w = rt_make_world()
    meshes = []
    for _ in range(int(rng.integers(1, 4))):
        n_verts = int(rng.integers(3, 40))
        with_normals = bool(rng.integers(0, 2))
        attrs = GEO_P | (GEO_N if with_normals else 0)
        pos = rng.uniform(-1, 1, (n_verts, 3)).astype(np.float32)
        if rng.integers(0, 3) == 0 and coincident:  # quantised positions: coincident vertices, axis-aligned and zero-area triangles
            pos = (np.round(pos * 2) / 2).astype(np.float32)
        cols = [np.concatenate([pos, np.zeros((n_verts, 1), np.float32)], axis=1)]
        if with_normals:
            nr = rng.normal(size=(n_verts, 3)).astype(np.float32)
            cols.append(np.concatenate([nr, np.zeros((n_verts, 1), np.float32)], axis=1))
        verts = np.concatenate(cols, axis=1).astype(np.float32)
        m = rt_world_add_mesh(w)
        if rng.integers(0, 3) == 0 and n_verts >= 6:  # auto-indexed: every three vertices are a triangle
            keep = n_verts - n_verts % 3
            _set_mesh(m, np.ascontiguousarray(verts[:keep]), None, attrs)
        else:
            n_tris = int(rng.integers(1, 60))
            idx = rng.integers(0, n_verts, (n_tris, 3)).astype(np.uint32)
            if n_tris > 2 and rng.integers(0, 2) == 0 and coincident:
                idx[-1] = idx[0]  # a coincident pair: the tie goes to the later leaf
            if not coincident:  # keep one triangle per bounding box, so that no two share a Morton code by construction
                corners = pos[idx]
                boxes = np.concatenate([corners.min(axis=1), corners.max(axis=1)], axis=1)
                _, keep = np.unique(boxes, axis=0, return_index=True)
                idx = idx[np.sort(keep)]
            _set_mesh(m, verts, idx.reshape(-1), attrs)
        meshes.append(m)
    mats = []
    for _ in range(int(rng.integers(2, 7))):
        m = rt_world_add_material(w)
        m.type = int(rng.integers(0, 5))
        m.albedo = v3(*rng.uniform(0, 1, 3))
        m.emissive = v3(*(rng.uniform(0, 4, 3) if (m.type == abi.MAT_LIGHT or rng.integers(0, 4) == 0) else (0, 0, 0)))
        m.roughness = float(np.float32(rng.uniform(0, 1))) if rng.integers(0, 3) else 0.0
        m.ior = float(np.float32(rng.uniform(1.0, 2.5)))
        m.billboard = bool(rng.integers(0, 4) == 0)
        mats.append(m)
    for _ in range(int(rng.integers(2, 13))):
        it = rt_world_add_instance(w)
        it.material = None if rng.integers(0, 8) == 0 else mats[int(rng.integers(0, len(mats)))]
        if rng.integers(0, 3) == 0:
            it.type = abi.INST_SPHERE
            it.sphere_center = v3(*rng.uniform(-3, 3, 3))
            it.sphere_radius = np.float32(rng.uniform(0.2, 1.5))
        else:
            it.type = abi.INST_MESH
            it.mesh = meshes[int(rng.integers(0, len(meshes)))]
            it.translation = v3(*rng.uniform(-3, 3, 3))
            if rng.integers(0, 4) == 0:
                it.rotation = IDENTITY_QUAT
            else:
                a, b = rng.normal(size=3), rng.normal(size=3)
                it.rotation = make_a_to_b_quat(normalize3(v3(*a)), normalize3(v3(*b)))
            s = rng.uniform(0.3, 3.0, 3) if rng.integers(0, 2) else np.full(3, rng.uniform(0.3, 3.0))
            if rng.integers(0, 6) == 0:
                s[int(rng.integers(0, 3))] *= -1  # a mirrored instance
            it.scale = v3(*s)
    eye = mul3(normalize3(v3(*rng.normal(size=3))), np.float32(rng.uniform(5, 9)))
    subject = v3(*rng.uniform(-1, 1, 3))
    mode = int(rng.integers(0, 3))
    sky = bool(rng.integers(0, 4) != 0)
    bounces = int(rng.integers(1, 13))

    vfov = float(np.float32(rng.uniform(0.4, 1.4)))  # radians
    defocus = float(np.float32(rng.uniform(0.005, 0.05)))

    def camera(width, height, samples):
        if mode == 0:
            return get_rt_cast_settings(width, height, samples, eye, subject, vfov=vfov)
        if mode == 1:
            return get_rt_cast_settings(width, height, samples, eye, subject, orthographic=True, vfov=vfov)
        return get_rt_cast_settings(width, height, samples, eye, subject, defocus_angle=defocus, vfov=vfov)

    return w, camera, sky, bounces


def aimed_rays(n: int, seed: int, reach: float = 6.0, spread: float = 3.0):
    """Seeded rays from points in [-reach, reach]^3 towards points in [-spread, spread]^3 (where random_world puts
    its instances), so that a good share of them hits something.  Returns float32 (origins[n, 3], directions[n, 3])."""
    rng = np.random.default_rng(seed)
    o = rng.uniform(-reach, reach, (n, 3)).astype(np.float32)
    to = rng.uniform(-spread, spread, (n, 3)).astype(np.float32)
    d = (to - o).astype(np.float32)
    ln = np.sqrt((d * d).sum(axis=1, dtype=np.float32)).astype(np.float32)
    ln[ln == 0] = np.float32(1)
    return o, (d / ln[:, None]).astype(np.float32)


def random_world_case(seed: int):
    """The inputs of one random-world fixture (tests/golden/random_worlds.npz): a world without coincident
    triangles, 600 aimed rays and the camera of a 24x16 frame.  Returns (world, sky, max_bounces, o, d, cast)."""
    w, cam, sky, bounces = random_world(seed, coincident=False)
    o, d = aimed_rays(600, seed)
    return w, sky, bounces, o, d, cam(24, 16, 1 + seed % 3)


def random_boxes(seed: int, max_n: int = 20_000):
    """A seeded box set for fuzzing the LBVH build: a random size (1 ... max_n) and one of several awkward
    distributions -- uniform, tight clusters, points on a coarse lattice (many equal Morton codes), a plane or a line
    (zero extent on one or two axes), wildly different magnitudes, tiny (denormal) coordinates, all boxes equal,
    negative zeros.  Returns float32[n, 6] (min xyz, max xyz)."""
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 2, 3, int(rng.integers(4, 64)), int(rng.integers(64, 2000)), int(rng.integers(2000, 20001))]))
    if max_n > 20_000 and seed % 6 != 0:  # the large sets: mostly large
        n = int(rng.integers(16_385, max_n + 1))
    kind = int(rng.integers(0, 9))
    cen = rng.uniform(-1, 1, (n, 3))
    half = rng.uniform(0, 0.05, (n, 3))
    if kind == 1:
        cen = rng.integers(0, 4, (n, 1)) * 10.0 + rng.normal(scale=1e-3, size=(n, 3))
    elif kind == 2:
        cen = rng.integers(0, 5, (n, 3)) * 0.25
        half = rng.integers(0, 2, (n, 3)) * 0.125
    elif kind == 3:
        cen[:, int(rng.integers(0, 3))] = 0.75
        half[:] = 0
    elif kind == 4:
        cen[:, 1:] = (3.0, -2.0)
        half[:, 1:] = 0
    elif kind == 5:
        cen = cen * 10.0 ** rng.integers(-20, 20, (n, 1))
    elif kind == 6:
        cen, half = cen * 1e-41, half * 1e-42
    elif kind == 7:
        cen[:] = cen[0]
        half[:] = half[0]
    elif kind == 8:
        cen = np.where(rng.integers(0, 2, (n, 3)) == 0, -0.0, cen)
        half[:] = 0
    cen, half = cen.astype(np.float32), half.astype(np.float32)
    return np.concatenate([cen - half, cen + half], axis=1).astype(np.float32)


def random_obj_text(seed: int):
    """A seeded random OBJ text within what the reference's loader reads (src/mesh/mesh.c:73-219) -> (text, attrs):
    one of the four face syntaxes, numbers in mixed notations, runs of blanks and tabs between fields, trailing
    blanks, comments, blank and foreign lines in between, faces with more than three corners, many repeated index
    triples, LF or CRLF line ends, with or without a line end after the last record."""
    rng = np.random.default_rng(seed)
    attrs = int(rng.choice([1, 3, 5, 7]))
    n_v, n_n, n_t = int(rng.integers(3, 60)), int(rng.integers(1, 20)), int(rng.integers(1, 20))
    n_f = int(rng.integers(1, 200))
    fmts = ["%.9g", "%g", "%.3f", "%e", "%.1f", "%+.4g"]

    def num(x):
        return fmts[int(rng.integers(0, len(fmts)))] % x

    def gap():
        return str(rng.choice([" ", " ", " ", "  ", "\t", " \t "]))

    def tail():
        return str(rng.choice(["", "", "", " ", "\t", "  "]))

    lines = []
    records = [("v", 3)] * n_v + ([("vn", 3)] * n_n if attrs & 2 else []) + ([("vt", 2)] * n_t if attrs & 4 else [])
    for tag, k in records:
        vals = rng.normal(scale=float(rng.choice([1e-3, 1.0, 100.0])), size=k)
        lines.append(tag + " " + gap().join(num(v) for v in vals) + tail())
    if rng.integers(0, 2):
        order = rng.permutation(len(lines))
        lines = [lines[k] for k in order]  # positions, normals and texture coordinates interleaved
    faces = []
    for _ in range(n_f):
        corners = int(rng.choice([3, 3, 3, 4, 5]))
        parts = []
        for _ in range(corners):
            p, n, t = int(rng.integers(1, n_v + 1)), int(rng.integers(1, min(n_n, 3) + 1)), int(rng.integers(1, min(n_t, 3) + 1))
            parts.append({1: "%d" % p, 3: "%d//%d" % (p, n), 5: "%d/%d" % (p, t), 7: "%d/%d/%d" % (p, t, n)}[attrs])
        faces.append("f " + gap().join(parts) + tail())
    lines += faces
    extras = ["# note", "", "g part", "s 1", "usemtl none", "o thing", "#", "   "]
    for _ in range(int(rng.integers(0, 12))):
        lines.insert(int(rng.integers(0, len(lines) + 1)), extras[int(rng.integers(0, len(extras)))])
    end = "\r\n" if rng.integers(0, 4) == 0 else "\n"
    text = end.join(lines) + (end if rng.integers(0, 3) else "")
    return text, (0 if attrs == 7 and rng.integers(0, 2) else attrs)
