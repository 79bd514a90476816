// Device-side building blocks shared by the rendering kernels (trace.hip, wavefront.hip):
// binary32 vector helpers in the reference's operation order, the slab / triangle / sphere
// tests, the resumable two-level BVH walk, hit epilogue, random numbers, camera rays and the
// per-bounce shading step.
//
// Traversal is the reference's: depth-first, always left before right, a leaf's primitive
// tested after its box, the interval shrinking at every accepted hit (src/lbvh/lbvh.c:127-161).
// On the pre-order node array that recursion is a linear walk: next = index + 1 to descend,
// next = skip when a box is missed or a leaf is done.  TLAS and BLAS walks run in ONE loop per
// lane (a lane switches arrays when it enters or leaves an instance), so lanes of a wave that
// are in different levels still advance together.
//
// Arithmetic is binary32 in the reference's operation order; every file that includes this
// header must be compiled with -ffp-contract=off (no fused multiply-add) and without fast-math.
#pragma once

#include "rthip_internal.h"

namespace {

constexpr int kBlock = 256;
constexpr float kEps = 1.1920929E-7f;       // EPSILON_F32, src/common/common_math.h:288
constexpr float kMaxF = 3.402823466e+38f;   // MAX_F32
constexpr float kPi = 3.1415926535897932384626433832795028841971693993751058209749445923078164062f;
constexpr float kSurfaceOffset = 0.001f;    // RT_CPU_SURFACE_OFFSET, raytracer_cpu.c:257

struct V3 {
    float x, y, z;
};
__device__ __forceinline__ V3 mk(float x, float y, float z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 add(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 sub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 scale(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ V3 mul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 divv(V3 a, V3 b) { return mk(a.x / b.x, a.y / b.y, a.z / b.z); }
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 cross(V3 a, V3 b) {  // common_math.c:29
    return mk(a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ V3 normalize(V3 a) {  // common_math.c:27-28: three divisions by the length
    float l = sqrtf(a.x * a.x + a.y * a.y + a.z * a.z);
    return mk(a.x / l, a.y / l, a.z / l);
}
__device__ __forceinline__ V3 rot_quat(V3 p, V3 qv, float qw) {  // common_math.c:91
    V3 a = scale(p, qw * qw - dot(qv, qv));
    V3 b = scale(qv, 2.f * dot(qv, p));
    V3 c = scale(cross(qv, p), 2.f * qw);
    return add(add(a, b), c);
}
__device__ __forceinline__ V3 reflect(V3 i, V3 n) { return sub(i, scale(n, 2.f * dot(i, n))); }  // common_math.c:30
__device__ __forceinline__ V3 orthogonal(V3 x) {  // common_math.c:37-46
    float ax = fabsf(x.x), ay = fabsf(x.y), az = fabsf(x.z);
    float mx = ((ax <= ay) & (ax <= az)) ? 1.f : 0.f;
    float my = ((ay < ax) & (ay <= az)) ? 1.f : 0.f;
    float mz = 1.f - mx - my;
    return cross(x, mk(mx, my, mz));
}
__device__ __forceinline__ V3 refract(V3 i, V3 n, float eta) {  // common_math.c:47-53
    float ndoti = dot(n, i);
    float k = 1.f - eta * eta * (1.f - ndoti * ndoti);
    if (k < 0.f) return mk(0.f, 0.f, 0.f);
    return sub(scale(i, eta), scale(n, eta * ndoti + sqrtf(k)));
}

// Whether a pixel belongs to this context's share of the frame (rthip_shard): mode 1 = 64x64 tiles dealt round-robin,
// mode 3 = bands of whole rows, `band_rows` each (shard_band_rows).  Modes 0 and 2 own every pixel.
__device__ __forceinline__ uint32_t shard_band_rows(const rthip_shard& shard, int height) {
    if (shard.mode != 3) return 1u;
    const uint32_t rows = ((uint32_t)height + shard.world - 1u) / shard.world;
    return (rows + 63u) & ~63u;
}
__device__ __forceinline__ bool shard_owns(const rthip_shard& shard, uint32_t px, uint32_t py, int width, uint32_t band_rows) {
    if (shard.mode == 1) return (((py / 64u) * (((uint32_t)width + 63u) / 64u) + (px / 64u)) % shard.world) == shard.rank;
    if (shard.mode == 3) return py / band_rows == shard.rank;
    return true;
}

struct Hit {
    uint32_t inst;  // 1-based TLAS id, 0 = miss
    uint32_t tri;   // index-buffer offset of the triangle ((id-1)*3), meshes only
    float u, v;
    float t;        // interval.max after the query
};

struct Counters {
    uint32_t nodes, tris, insts;
    uint32_t wave_steps;  // node steps issued by this lane's wave (counted on one lane per step)
    uint32_t lane_steps;  // node steps this lane took part in: sum over lanes / (64 * wave_steps) = share of the lanes at work
};

// lbvh_aabb_query_ray, lbvh.c:127-149, with 1/d hoisted (the same correctly rounded
// quotient every time, so hoisting does not change a bit).  Works on a copy of the
// interval, as the reference does.  Written with selects instead of the reference's nested
// ifs: `t0 < t1 ? (t0, t1) : (t1, t0)` picks the same candidates, the two strict comparisons
// update the interval exactly when the reference's do (a NaN candidate, from 0 * inf, fails them
// and changes nothing), and because the lower end only rises and the upper end only falls, one
// `max < min` test after the third axis rejects exactly the boxes the reference rejects after
// the first axis that empties the interval.
__device__ __forceinline__ void slab_axis(float lo, float hi, float o, float inv, float& tmin, float& tmax) {
    const float t0 = (lo - o) * inv, t1 = (hi - o) * inv;
    const bool ordered = t0 < t1;
    const float near = ordered ? t0 : t1, far = ordered ? t1 : t0;
    // the interval ends are never NaN, so maxNum / minNum (a NaN candidate is ignored) are the strict comparisons
    tmin = fmaxf(tmin, near);
    tmax = fminf(tmax, far);
}
__device__ __forceinline__ bool slab(float4 lo, float4 hi, V3 o, V3 inv, float tmin, float tmax) {
    slab_axis(lo.x, hi.x, o.x, inv.x, tmin, tmax);
    slab_axis(lo.y, hi.y, o.y, inv.y, tmin, tmax);
    slab_axis(lo.z, hi.z, o.z, inv.z, tmin, tmax);
    return !(tmax < tmin);
}

// geo_intersect_tri, src/geo/geo.c:72-109
__device__ __forceinline__ bool tri_test(V3 o, V3 d, V3 a, V3 b, V3 c, float tmin, float& tmax, float& u_out,
                                         float& v_out) {
    V3 edge1 = sub(b, a);
    V3 edge2 = sub(c, a);
    V3 rxe2 = cross(d, edge2);
    float det = dot(edge1, rxe2);
    if (det == 0.f) return false;
    float inv_det = 1.f / det;
    V3 s = sub(o, a);
    float u = inv_det * dot(s, rxe2);
    if ((u < 0.f && fabsf(u) > kEps) || (u > 1.f && fabsf(u - 1.f) > kEps)) return false;
    V3 sxe1 = cross(s, edge1);
    float v = inv_det * dot(d, sxe1);
    if ((v < 0.f && fabsf(v) > kEps) || (u + v > 1.f && fabsf(u + v - 1.f) > kEps)) return false;
    float t = inv_det * dot(edge2, sxe1);
    if (!(t >= tmin && t <= tmax)) return false;  // geo_in_interval, geo.c:32-34
    tmax = t;
    u_out = u;
    v_out = v;
    return true;
}

// geo_intersect_sphere, geo.c:36-69
__device__ __forceinline__ bool sphere_test(V3 o, V3 d, V3 center, float radius, float tmin, float& tmax) {
    V3 omc = sub(o, center);
    float a = dot(d, d);
    float half_b = dot(d, omc);
    float c = dot(omc, omc) - radius * radius;
    float qtr_d = half_b * half_b - a * c;
    if (qtr_d < 0.f) return false;
    float half_sqrt_d = sqrtf(qtr_d);
    float t = (-half_b - half_sqrt_d) / a;
    if (!(t >= tmin && t <= tmax)) {
        t = (-half_b + half_sqrt_d) / a;
        if (!(t >= tmin && t <= tmax)) return false;
    }
    tmax = t;
    return true;
}

// rt_cpu_intersect's query, raytracer_cpu.c:402-407: lbvh_query_ray over the TLAS with
// rt_cpu_tlas_hit -> rt_cpu_intersect_tlas_node (:494-531) -> BLAS query with
// rt_cpu_blas_node_hit (:459-492).  The walk is kept in a struct so that a lane can be
// suspended between two steps (the persistent kernel refills idle lanes in between).
struct Trav {
    V3 wo, wd, winv;    // world-space ray and 1/d
    V3 o, d, inv;       // ray of the level being walked (model space inside an instance)
    float tmax;         // interval.max; interval.min stays EPSILON_F32 (geo_make_pos_interval, geo.c:28-30)
    uint32_t idx, end;  // next node and node count of the array being walked
    const float4* nodes;  // the array being walked (TLAS, or one mesh's part of the BLAS array)
    uint32_t resume;    // TLAS index to continue from when the current BLAS walk ends
    uint32_t cur_inst;  // TLAS id of the instance being walked (0: in the TLAS)
    uint32_t tri_base;
    uint32_t pend_id;   // TLAS leaf whose instance is still to be entered (0: none); see trav_enter
    uint32_t pend_skip; // where the TLAS walk continues after that instance
    Hit h;
};

__device__ __forceinline__ void trav_begin(Trav& T, const DevScene& S, V3 o, V3 d) {
    T.wo = o;
    T.wd = d;
    T.winv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
    T.o = o;
    T.d = d;
    T.inv = T.winv;
    T.tmax = kMaxF;
    T.idx = 0;
    T.end = S.n_tlas;
    T.nodes = S.tlas;
    T.resume = 0;
    T.cur_inst = 0;
    T.tri_base = 0;
    T.pend_id = 0;
    T.pend_skip = 0;
    T.h.inst = 0;
    T.h.tri = 0;
    T.h.u = 0.f;
    T.h.v = 0.f;
    T.h.t = kMaxF;
}

// A TLAS leaf whose box the ray hits (rt_cpu_intersect_tlas_node, raytracer_cpu.c:494-531): a sphere is tested,
// a mesh instance is entered (the ray goes to model space and the BLAS walk starts).
template <bool COUNT>
__device__ __forceinline__ void trav_enter_leaf(Trav& T, const DevScene& S, Counters& cnt, uint32_t id, uint32_t skip) {
    const DevInstance* it = &S.inst[id - 1];
    if (COUNT) cnt.insts++;
    if (it->type == RTHIP_INST_SPHERE) {
        const float4 cr = *(const float4*)it->sphere_center;
        if (sphere_test(T.o, T.d, mk(cr.x, cr.y, cr.z), it->sphere_radius, kEps, T.tmax)) {
            T.h.inst = id;
            T.h.tri = 0;
            T.h.u = 0.f;
            T.h.v = 0.f;
        }
        T.idx = skip;
        return;
    }
    // rt_cpu_inv_transform_ray, raytracer_cpu.c:572-580,607-612; the model-space direction is
    // not renormalised, so t is shared between the two spaces
    const float4 tr = *(const float4*)it->translation;
    const float4 q = *(const float4*)it->rotation;
    const float4 sc = *(const float4*)it->scale;
    const V3 qv = mk(-q.x, -q.y, -q.z);  // inv_quat, common_math.c:90
    const V3 s3 = mk(sc.x, sc.y, sc.z);
    T.o = divv(rot_quat(sub(T.wo, mk(tr.x, tr.y, tr.z)), qv, q.w), s3);
    T.d = divv(rot_quat(T.wd, qv, q.w), s3);
    T.inv = mk(1.f / T.d.x, 1.f / T.d.y, 1.f / T.d.z);
    T.resume = skip;
    T.cur_inst = id;
    T.tri_base = it->tri_base;
    T.nodes = S.blas + 2 * (size_t)it->node_base;
    T.idx = 0;
    T.end = it->n_nodes;
}

// Enters the instance a lane was parked at by trav_step<COUNT, true>.
template <bool COUNT>
__device__ __forceinline__ void trav_enter(Trav& T, const DevScene& S, Counters& cnt) {
    const uint32_t id = T.pend_id, skip = T.pend_skip;
    T.pend_id = 0;
    trav_enter_leaf<COUNT>(T, S, cnt, id, skip);
}

// One step of the walk.  Returns true when the query has ended (T.h is final).
template <bool COUNT, bool DEFER = false>
__device__ __forceinline__ bool trav_step(Trav& T, const DevScene& S, Counters& cnt) {
    if (T.idx >= T.end) {
        if (T.cur_inst == 0) {
            T.h.t = T.tmax;
            return true;
        }
        // leave the instance: back to the TLAS walk with the world-space ray
        T.cur_inst = 0;
        T.nodes = S.tlas;
        T.idx = T.resume;
        T.end = S.n_tlas;
        T.o = T.wo;
        T.d = T.wd;
        T.inv = T.winv;
        return false;
    }
    // the context's trees hold (min.x, max.x, min.y, max.y | min.z, max.z, skip, id): the two bounds of an axis arrive in
    // neighbouring registers, as the packed subtract / multiply of the slab test take them (put_node, lbvh_build.hip)
    const float4 ra = T.nodes[2 * (size_t)T.idx], rb = T.nodes[2 * (size_t)T.idx + 1];
    const float4 lo = make_float4(ra.x, ra.z, rb.x, 0.f), hi = make_float4(ra.y, ra.w, rb.y, 0.f);
    if (COUNT) cnt.nodes++;
    const uint32_t id = __float_as_uint(rb.w);
    uint32_t skip = __float_as_uint(rb.z);
    if (skip <= T.idx) skip = T.end;  // a well-formed tree never needs this; it bounds the walk on a corrupt one
    if (!slab(lo, hi, T.o, T.inv, kEps, T.tmax)) {
        T.idx = skip;
        return false;
    }
    if (id == 0) {  // inner node: left child is the next record
        T.idx = T.idx + 1;
        return false;
    }
    if (T.cur_inst != 0) {
        // BLAS leaf: triangle id (1-based), vertices pre-gathered per triangle
        const float4* tv = S.tri_v + 3 * (size_t)(T.tri_base + (id - 1));
        const float4 a = tv[0], b = tv[1], c = tv[2];
        if (COUNT) cnt.tris++;
        float u, v;
        if (tri_test(T.o, T.d, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), kEps, T.tmax, u, v)) {
            T.h.inst = T.cur_inst;
            T.h.tri = (id - 1) * 3;  // idx = (id - 1)*3, raytracer_cpu.c:467,479
            T.h.u = u;
            T.h.v = v;
        }
        T.idx = skip;
        return false;
    }
    // TLAS leaf
    if (DEFER) {  // the caller enters the instance later, for several lanes at once (trav_enter)
        T.pend_id = id;
        T.pend_skip = skip;
        return false;
    }
    trav_enter_leaf<COUNT>(T, S, cnt, id, skip);
    return false;
}

template <bool COUNT>
__device__ __forceinline__ Hit closest_hit_query(const DevScene& S, V3 wo, V3 wd, Counters& cnt) {
    Trav T;
    trav_begin(T, S, wo, wd);
    for (;;) {
        if (COUNT) {
            const uint64_t walking = __ballot(1);
            if ((int)(threadIdx.x & 63) == __ffsll((long long)walking) - 1) cnt.wave_steps++;
            cnt.lane_steps++;
        }
        if (trav_step<COUNT>(T, S, cnt)) break;
    }
    return T.h;
}

// The same query with the instance entries of a wave run in batches: a lane that reaches an instance parks there
// (trav_step<COUNT, true>) and the wave runs the entry (two quaternion rotations, nine divisions: the longest branch of
// the walk) once `batch` lanes wait or no lane is left to step.  Every lane does the same tests in the same order; only
// how often the wave executes the entry code changes.  Pays where the meshes are a handful of triangles and rays
// enter instances every few steps (Cornell: +6 %); with real meshes the parked lanes cost more than the entries saved.
template <bool COUNT>
__device__ __forceinline__ Hit closest_hit_query_batched(const DevScene& S, V3 wo, V3 wd, Counters& cnt, uint32_t batch) {
    Trav T;
    trav_begin(T, S, wo, wd);
    for (;;) {
        const uint64_t walking = __ballot(1);  // the lanes still in this loop
        if (COUNT && (int)(threadIdx.x & 63) == __ffsll((long long)walking) - 1) cnt.wave_steps++;
        if (COUNT) cnt.lane_steps++;
        const uint64_t parked = __ballot(T.pend_id != 0);
        const bool enter_now = parked != 0 && ((uint32_t)__popcll(parked) >= batch || parked == walking);
        if (T.pend_id != 0) {
            if (enter_now) trav_enter<COUNT>(T, S, cnt);
        } else if (trav_step<COUNT, true>(T, S, cnt)) {
            break;
        }
    }
    return T.h;
}

template <bool COUNT>
__device__ __forceinline__ bool any_hit_query(const DevScene& S, V3 wo, V3 wd, float tmax, Counters& cnt);

// ---- ordered traversal (RTHIP_TRAV_FAST) ------------------------------------------------------------------------
// The reference's query visits the two children of a node always left first (lbvh.c:157-158), whatever the ray, so a
// ray that hits something close by still walks every subtree whose box it pierces in front of AND behind the hit until
// the interval has shrunk.  The opt-in fast mode walks the same tree with the same box, triangle and sphere tests, but
// (1) from 4-wide records collapsed from the binary tree (wide_emit_kernel, lbvh_build.hip: one 128-byte line holds
// the boxes of a node's grandchildren, so a step is one aligned fetch and the chain of dependent fetches of a ray is
// half as long; leaves have no record of their own, a reference is either a record or a primitive id), (2) nearest
// box first, the others pushed on a per-lane stack in LDS together with the distance at which the ray enters them, so
// that a close hit shrinks the interval before the far subtrees are looked at and a subtree that has fallen behind the
// hit by the time it is popped costs one comparison instead of a fetch.
// Same acceptance rule for every primitive (min <= t <= max, ties accepted), so the result is the closest hit too; it
// can differ from the reference's only where the reference's own answer depends on its visiting order: among hits at
// exactly equal t the last visited wins, and a primitive whose box test (no slack) fails against the already shrunk
// interval is never tested although its triangle test (with slack) would have accepted it.
constexpr uint32_t kLeafBit = 0x80000000u;  // reference: kLeafBit | 1-based primitive id, else index of a record
constexpr uint32_t kPop = 0xffffffffu;      // "take the next reference from the stack"; also the reference of an empty slot
// Stack entries per lane, in LDS (entry e of a lane at lds[e * kBlock]: 8 bytes, no bank conflicts).  The stack holds
// the boxes hit but not yet entered along the current path: up to three per level of the 4-wide tree, in practice one
// or two for a few levels, and LDS per workgroup decides how many waves a CU holds (measured: 16 entries against 24
// is +60 % on frames through 10 000 instances; 12 entries lose 7 % on rays through that TLAS).  When the boxes of a step do not fit, the farthest ones are gone
// through at once in the binary tree's stackless order (inside a mesh; see trav_fast_step), so a short stack costs
// some ordering, not the answer.  Only a ray that runs out of stack among the TLAS's own boxes sets `overflow` and
// is answered by the stackless fixed-order walk instead, which needs no stack at all.
#ifndef RT_FAST_STACK
#define RT_FAST_STACK 16
#endif
constexpr int kFastLds = RT_FAST_STACK;

struct FastStack {
    uint2* lds;  // this lane's column: (key, reference) per entry
};
__device__ __forceinline__ uint2* fast_stack_column(uint32_t* workgroup_base) { return (uint2*)workgroup_base + threadIdx.x; }

// The slab test of `slab` that also reports where the ray enters the box (the lower end of the clipped interval).
__device__ __forceinline__ bool slab_near(float4 xy, float zlo, float zhi, V3 o, V3 inv, float tmax, float& tnear) {
    float tmin = kEps;
    slab_axis(xy.x, xy.y, o.x, inv.x, tmin, tmax);
    slab_axis(xy.z, xy.w, o.y, inv.y, tmin, tmax);
    slab_axis(zlo, zhi, o.z, inv.z, tmin, tmax);
    tnear = tmin;
    return !(tmax < tmin);
}

// Sort key of a box the ray hits: the entry distance (>= EPSILON_F32 > 0, so its bits order like the number) with its
// two lowest bits replaced by the slot, so that boxes entered at (nearly) the same distance go left first like the
// reference and a key tells which slot it came from.  `key & ~3` as a float is <= the true entry distance: a subtree
// is dropped at pop time only if even that lies behind the interval's end -- exactly when its box test would fail.
__device__ __forceinline__ uint32_t fast_key(bool hit, float tnear, uint32_t slot) {
    return hit ? ((__float_as_uint(tnear) & ~3u) | slot) : 0xffffffffu;
}
__device__ __forceinline__ void fast_order(uint32_t& ka, uint32_t& ra, uint32_t& kb, uint32_t& rb) {
    const bool swap = kb < ka;
    const uint32_t k0 = swap ? kb : ka, k1 = swap ? ka : kb, r0 = swap ? rb : ra, r1 = swap ? ra : rb;
    ka = k0;
    kb = k1;
    ra = r0;
    rb = r1;
}

struct TravFast {
    V3 wo, wd, winv;  // world-space ray and 1/d
    V3 o, d, inv;     // ray of the level being walked
    float tmax;
    uint32_t cur;       // the reference to look at next (kPop: take one from the stack)
    uint32_t sp, base;  // stack height, and the height at which the current instance was entered
    uint32_t cur_inst;  // TLAS id of the instance being walked (0: in the TLAS)
    uint32_t tri_base;
    uint32_t overflow;  // the stack was too short for this ray: T.h is not valid, ask the stackless walk
    uint32_t pend_id;   // TLAS leaf whose instance is still to be entered (0: none); see trav_fast_enter
    const float4* nodes;  // the records being walked: 8 x float4 each
    Hit h;
};

__device__ __forceinline__ void trav_fast_begin(TravFast& T, const DevScene& S, V3 o, V3 d) {
    T.wo = o;
    T.wd = d;
    T.winv = mk(1.f / d.x, 1.f / d.y, 1.f / d.z);
    T.o = o;
    T.d = d;
    T.inv = T.winv;
    T.tmax = kMaxF;
    T.cur = S.n_inst == 0 ? kPop : (S.n_inst == 1 ? (kLeafBit | 1u) : 0u);
    T.sp = 0;
    T.base = 0;
    T.cur_inst = 0;
    T.tri_base = 0;
    T.overflow = 0;
    T.pend_id = 0;
    T.nodes = S.tlas_wide;
    T.h.inst = 0;
    T.h.tri = 0;
    T.h.u = 0.f;
    T.h.v = 0.f;
    T.h.t = kMaxF;
}

// A TLAS leaf whose turn has come (rt_cpu_intersect_tlas_node, raytracer_cpu.c:494-531): a sphere is tested, a mesh
// instance is entered -- the ray goes to model space and the instance's records are walked from the stack height at
// which it was entered.
template <bool COUNT>
__device__ __forceinline__ void trav_fast_enter_leaf(TravFast& T, const DevScene& S, Counters& cnt, uint32_t id) {
    const DevInstance* it = &S.inst[id - 1];
    if (COUNT) cnt.insts++;
    if (it->type == RTHIP_INST_SPHERE) {
        const float4 cr = *(const float4*)it->sphere_center;
        if (sphere_test(T.o, T.d, mk(cr.x, cr.y, cr.z), it->sphere_radius, kEps, T.tmax)) {
            T.h.inst = id;
            T.h.tri = 0;
            T.h.u = 0.f;
            T.h.v = 0.f;
        }
        return;
    }
    const float4 tr = *(const float4*)it->translation;
    const float4 q = *(const float4*)it->rotation;
    const float4 sc = *(const float4*)it->scale;
    const V3 qv = mk(-q.x, -q.y, -q.z);
    const V3 s3 = mk(sc.x, sc.y, sc.z);
    T.o = divv(rot_quat(sub(T.wo, mk(tr.x, tr.y, tr.z)), qv, q.w), s3);
    T.d = divv(rot_quat(T.wd, qv, q.w), s3);
    T.inv = mk(1.f / T.d.x, 1.f / T.d.y, 1.f / T.d.z);
    T.cur_inst = id;
    T.tri_base = it->tri_base;
    T.nodes = S.blas_wide + 8 * (size_t)it->wide_base;
    T.base = T.sp;
    T.cur = it->n_nodes == 1 ? (kLeafBit | 1u) : 0u;  // a one-triangle mesh has no record
}

// Enters the instance a lane was parked at by trav_fast_step<COUNT, true>.
template <bool COUNT>
__device__ __forceinline__ void trav_fast_enter(TravFast& T, const DevScene& S, Counters& cnt) {
    const uint32_t id = T.pend_id;
    T.pend_id = 0;
    trav_fast_enter_leaf<COUNT>(T, S, cnt, id);
}

// One step.  Returns true when the query has ended (T.h is final).  With DEFER a lane that reaches an instance parks
// there (T.pend_id) and the caller enters it later with trav_fast_enter, for several lanes of the wave at once: the entry
// is the longest branch of a step (two quaternion rotations, nine divisions) and in a scene of many instances nearly
// every step of a wave has some lane that needs it.
template <bool COUNT, bool DEFER = false>
__device__ __forceinline__ bool trav_fast_step(TravFast& T, const DevScene& S, Counters& cnt, FastStack& st) {
    uint32_t cur = T.cur;
    if (cur == kPop) {
        // the nearest waiting box that still reaches into the interval; the ones behind its end are dropped unseen
        while (T.sp != T.base) {
            T.sp--;
            const uint2 e = st.lds[T.sp * kBlock];
            if (!(__uint_as_float(e.x & ~3u) > T.tmax)) {
                cur = e.y;
                break;
            }
        }
        if (cur == kPop) {
            if (T.cur_inst == 0) {
                T.h.t = T.tmax;
                return true;
            }
            // the instance's subtree is done: back to the TLAS walk with the world-space ray
            T.cur_inst = 0;
            T.nodes = S.tlas_wide;
            T.o = T.wo;
            T.d = T.wd;
            T.inv = T.winv;
            T.base = 0;
            return false;
        }
    }
    if (cur & kLeafBit) {
        const uint32_t id = cur & ~kLeafBit;
        T.cur = kPop;
        if (T.cur_inst != 0) {  // a triangle
            const float4* tv = S.tri_v + 3 * (size_t)(T.tri_base + (id - 1));
            const float4 a = tv[0], b = tv[1], c = tv[2];
            if (COUNT) cnt.tris++;
            float u, v;
            if (tri_test(T.o, T.d, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), kEps, T.tmax, u, v)) {
                T.h.inst = T.cur_inst;
                T.h.tri = (id - 1) * 3;
                T.h.u = u;
                T.h.v = v;
            }
            return false;
        }
        if (S.n_inst == 1u) {
            // The TLAS is this one leaf and has no record, so no step has tested its box: the reference does
            // (lbvh.c:151-156), and a ray that misses it is spared the change to model space.
            const float4 ra = S.tlas[0], rb = S.tlas[1];
            if (COUNT) cnt.nodes++;
            if (!slab(make_float4(ra.x, ra.z, rb.x, 0.f), make_float4(ra.y, ra.w, rb.y, 0.f), T.o, T.inv, kEps, T.tmax)) return false;
        }
        if (DEFER) {  // the caller enters the instance later (trav_fast_enter)
            T.pend_id = id;
            return false;
        }
        trav_fast_enter_leaf<COUNT>(T, S, cnt, id);
        return false;
    }
    const float4* rec = T.nodes + 8 * (size_t)cur;
    const float4 b0 = rec[0], b1 = rec[1], b2 = rec[2], b3 = rec[3], z01 = rec[4], z23 = rec[5];
    const float4 refs = rec[6];
    uint32_t r0 = __float_as_uint(refs.x), r1 = __float_as_uint(refs.y), r2 = __float_as_uint(refs.z), r3 = __float_as_uint(refs.w);
    float t0, t1, t2, t3;
    const bool h0 = slab_near(b0, z01.x, z01.y, T.o, T.inv, T.tmax, t0);
    const bool h1 = slab_near(b1, z01.z, z01.w, T.o, T.inv, T.tmax, t1);
    const bool h2 = slab_near(b2, z23.x, z23.y, T.o, T.inv, T.tmax, t2) && r2 != kPop;
    const bool h3 = slab_near(b3, z23.z, z23.w, T.o, T.inv, T.tmax, t3) && r3 != kPop;
    if (COUNT) cnt.nodes += 2u + (r2 != kPop ? 1u : 0u) + (r3 != kPop ? 1u : 0u);
    uint32_t k0 = fast_key(h0, t0, 0u), k1 = fast_key(h1, t1, 1u), k2 = fast_key(h2, t2, 2u), k3 = fast_key(h3, t3, 3u);
    // sorting network of four: (0,1) (2,3) (0,2) (1,3) (1,2); boxes that were missed carry the largest key and end up last
    fast_order(k0, r0, k1, r1);
    fast_order(k2, r2, k3, r3);
    fast_order(k0, r0, k2, r2);
    fast_order(k1, r1, k3, r3);
    fast_order(k1, r1, k2, r2);
    uint32_t hits = (h0 ? 1u : 0u) + (h1 ? 1u : 0u) + (h2 ? 1u : 0u) + (h3 ? 1u : 0u);
    if (hits > 1u) {
        if (T.sp + hits - 1u > (uint32_t)kFastLds) {
            // No room for all of them.  Inside a mesh the farthest boxes that do not fit are gone through here and now,
            // in the binary tree's own stackless order (trav_step's walk over that slot's subtree): the same tests, only
            // not nearest first.  At the TLAS level the walk gives up and the caller asks the stackless walk.
            if (T.cur_inst == 0) {
                T.overflow = 1;
                T.h.t = T.tmax;
                return true;
            }
            const float4 bins = rec[7];
            const float4* bn = S.blas + 2 * (size_t)S.inst[T.cur_inst - 1].node_base;
            while (hits > 1u && T.sp + hits - 1u > (uint32_t)kFastLds) {
                const uint32_t key = hits > 3u ? k3 : (hits > 2u ? k2 : k1);  // the farthest one left; its low bits name the slot
                const uint32_t slot = key & 3u;
                uint32_t idx = __float_as_uint(slot == 0u ? bins.x : (slot == 1u ? bins.y : (slot == 2u ? bins.z : bins.w)));
                uint32_t end = 0xffffffffu;
                while (idx < end) {
                    const float4 ra = bn[2 * (size_t)idx], rb = bn[2 * (size_t)idx + 1];
                    uint32_t skip = __float_as_uint(rb.z);
                    const uint32_t id = __float_as_uint(rb.w);
                    if (skip <= idx) skip = idx + 1u;  // never on a well-formed tree; bounds the walk on a corrupt one
                    if (end == 0xffffffffu) end = skip;  // the first node is the slot's: its subtree ends at its skip
                    if (COUNT) cnt.nodes++;
                    if (!slab(make_float4(ra.x, ra.z, rb.x, 0.f), make_float4(ra.y, ra.w, rb.y, 0.f), T.o, T.inv, kEps, T.tmax)) {
                        idx = skip;
                        continue;
                    }
                    if (id == 0u) {
                        idx++;
                        continue;
                    }
                    const float4* tv = S.tri_v + 3 * (size_t)(T.tri_base + (id - 1));
                    const float4 a = tv[0], b = tv[1], c = tv[2];
                    if (COUNT) cnt.tris++;
                    float u, v;
                    if (tri_test(T.o, T.d, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), kEps, T.tmax, u, v)) {
                        T.h.inst = T.cur_inst;
                        T.h.tri = (id - 1) * 3;
                        T.h.u = u;
                        T.h.v = v;
                    }
                    idx = skip;
                }
                hits--;
            }
        }
        // farthest first, so that the nearest of them is popped first
        if (hits > 3u) st.lds[(T.sp++) * kBlock] = make_uint2(k3, r3);
        if (hits > 2u) st.lds[(T.sp++) * kBlock] = make_uint2(k2, r2);
        if (hits > 1u) st.lds[(T.sp++) * kBlock] = make_uint2(k1, r1);
    }
    T.cur = hits ? r0 : kPop;
    return false;
}

template <bool COUNT>
__device__ __forceinline__ Hit closest_hit_query_fast(const DevScene& S, V3 wo, V3 wd, Counters& cnt, uint32_t* lds_base) {
    TravFast T;
    FastStack st;
    st.lds = fast_stack_column(lds_base);
    trav_fast_begin(T, S, wo, wd);
    for (;;) {
        if (COUNT) {
            const uint64_t walking = __ballot(1);
            if ((int)(threadIdx.x & 63) == __ffsll((long long)walking) - 1) cnt.wave_steps++;
            cnt.lane_steps++;
        }
        if (trav_fast_step<COUNT>(T, S, cnt, st)) break;
    }
    if (T.overflow) return closest_hit_query<COUNT>(S, wo, wd, cnt);
    return T.h;
}

template <bool COUNT>
__device__ __forceinline__ bool any_hit_query_fast(const DevScene& S, V3 wo, V3 wd, float tmax, Counters& cnt, uint32_t* lds_base) {
    TravFast T;
    FastStack st;
    st.lds = fast_stack_column(lds_base);
    trav_fast_begin(T, S, wo, wd);
    T.tmax = tmax;
    for (;;) {
        if (COUNT) {
            const uint64_t walking = __ballot(1);
            if ((int)(threadIdx.x & 63) == __ffsll((long long)walking) - 1) cnt.wave_steps++;
            cnt.lane_steps++;
        }
        if (trav_fast_step<COUNT>(T, S, cnt, st) || T.h.inst != 0) break;
    }
    if (T.overflow) return any_hit_query<COUNT>(S, wo, wd, tmax, cnt);
    return T.h.inst != 0;
}

// Occlusion: whether the same query over the interval [EPSILON_F32, tmax] would report a hit.  The
// walk up to the first accepted primitive does not depend on anything after it, so it stops there.
template <bool COUNT>
__device__ __forceinline__ bool any_hit_query(const DevScene& S, V3 wo, V3 wd, float tmax, Counters& cnt) {
    Trav T;
    trav_begin(T, S, wo, wd);
    T.tmax = tmax;
    for (;;) {
        if (COUNT) {
            const uint64_t walking = __ballot(1);
            if ((int)(threadIdx.x & 63) == __ffsll((long long)walking) - 1) cnt.wave_steps++;
            cnt.lane_steps++;
        }
        if (trav_step<COUNT>(T, S, cnt) || T.h.inst != 0) break;
    }
    return T.h.inst != 0;
}

// epilogue of rt_cpu_intersect, raytracer_cpu.c:411-454: position and shading normal
__device__ __forceinline__ void finish_hit(const DevScene& S, V3 o, V3 d, const Hit& h, V3& p, V3& n, int& material) {
    const DevInstance* it = &S.inst[h.inst - 1];
    p = add(o, scale(d, h.t));
    material = it->material;
    if (it->type == RTHIP_INST_SPHERE) {
        const float4 cr = *(const float4*)it->sphere_center;
        n = scale(sub(p, mk(cr.x, cr.y, cr.z)), 1.f / it->sphere_radius);
        return;
    }
    const size_t t3 = 3 * (size_t)(it->tri_base + h.tri / 3);
    V3 tri_n;
    if (it->has_normals) {
        // U weights vertex 0 and V vertex 1, as the reference writes it (:441-445)
        const float4 n0 = S.tri_n[t3], n1 = S.tri_n[t3 + 1], n2 = S.tri_n[t3 + 2];
        tri_n = add(add(scale(mk(n0.x, n0.y, n0.z), h.u), scale(mk(n1.x, n1.y, n1.z), h.v)),
                    scale(mk(n2.x, n2.y, n2.z), 1.f - h.u - h.v));
    } else {
        const float4 a = S.tri_v[t3], b = S.tri_v[t3 + 1], c = S.tri_v[t3 + 2];
        tri_n = cross(sub(mk(b.x, b.y, b.z), mk(a.x, a.y, a.z)), sub(mk(c.x, c.y, c.z), mk(a.x, a.y, a.z)));
    }
    const float4 q = *(const float4*)it->rotation;
    const float4 sc = *(const float4*)it->scale;
    n = normalize(rot_quat(mul(tri_n, mk(sc.x, sc.y, sc.z)), mk(q.x, q.y, q.z), q.w));  // rt_cpu_transform_dir, :575-577
}

// ---- random numbers -------------------------------------------------------------------
// The reference draws from one global rand() stream (src/common/common_math.h:310), which
// ties the image to the serial pixel order.  Here every draw is a pure function of
// (seed, pixel, sample, draw index), 31 bits out, converted exactly like (f32)rand()/(f32)RAND_MAX (RAND_MAX rounds
// to 2^31).  All of it is 32-bit arithmetic: a 32-bit integer multiply runs at a quarter of the VALU rate here and a
// 64-bit one is four of them, and a generator built on two 64-bit multiplies per draw (the splitmix64 finaliser, which
// this one replaced) was half of what a pixel costs before its ray meets anything.  hash32 is a two-multiply integer
// hash with low avalanche bias; a path's key is two independently hashed words (a, b) of (seed, pixel, sample),
// draw d is hash32(a + d * 0x9E3779B9) ^ b.
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
struct Rng {
    uint64_t key;   // a | b << 32
    uint32_t draw;  // d * 0x9E3779B9, kept as a running sum
};
__device__ __forceinline__ Rng rng_make(uint64_t seed, uint32_t pixel, uint32_t sample) {
    Rng g;
    // the seed is hashed before it meets the sample index (seeds that differ in their low bits must not hand the
    // same streams to other strata), and the second word is a hash of (seed, pixel, sample) of its own, not of the
    // first: the key has 64 independent bits.  Both seed words are wave-uniform, so they cost scalar instructions.
    const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    const uint32_t s1 = hash32(lo ^ hash32(hi + 0x6a09e667u));
    const uint32_t s2 = hash32(hi ^ hash32(lo + 0xbb67ae85u));
    const uint32_t a = hash32(pixel + hash32(sample + s1));
    const uint32_t b = hash32(sample + hash32(pixel + s2));
    g.key = (uint64_t)a | ((uint64_t)b << 32);
    g.draw = 0;
    return g;
}
__device__ __forceinline__ float rand_unit(Rng& g) {
    const uint32_t v = (hash32((uint32_t)g.key + g.draw) ^ (uint32_t)(g.key >> 32)) >> 1;
    g.draw += 0x9E3779B9u;
    return (float)(int)v / 2147483648.0f;
}

// sin and cos for 0 <= x <= 2*pi in binary32 without fused operations: Cody-Waite
// reduction by pi/2 and the classic single-precision minimax kernels.  The host-side
// checker evaluates the same expressions, so both produce the same bits.
__device__ __forceinline__ void poly_sincos(float x, float& s_out, float& c_out) {
    int k = (int)(x * 0.636619772367581343f + 0.5f);
    float kf = (float)k;
    float r = ((x - kf * 1.5703125f) - kf * 4.837512969970703125e-4f) - kf * 7.54978995489188216e-8f;
    float r2 = r * r;
    float s = r + r * r2 * (-1.6666667163e-01f + r2 * (8.3333337680e-03f + r2 * (-1.9841270114e-04f + r2 * 2.7557314297e-06f)));
    float c = (1.f - 0.5f * r2) + r2 * r2 * (4.1666667908e-02f + r2 * (-1.3888889225e-03f + r2 * (2.4801587642e-05f + r2 * -2.7557314297e-07f)));
    switch (k & 3) {
        case 0: s_out = s; c_out = c; break;
        case 1: s_out = c; c_out = -s; break;
        case 2: s_out = -s; c_out = -c; break;
        default: s_out = -c; c_out = s; break;
    }
}

// rt_cpu_cosine_sample_hemisphere + rt_cpu_transform_uvw_to_hemisphere, raytracer_cpu.c:536-559
__device__ __forceinline__ V3 cosine_sample(V3 n, Rng& g) {
    float r1 = rand_unit(g);
    float r2 = rand_unit(g);
    float phi = 2.f * kPi * r1;
    float sn, cs;
    poly_sincos(phi, sn, cs);
    float sq = sqrtf(r2);
    float u = cs * sq;
    float v = sn * sq;
    float w = sqrtf(1.f - r2);
    V3 ub = orthogonal(n);
    V3 vb = cross(ub, n);
    return normalize(add(add(scale(ub, u), scale(vb, v)), scale(n, w)));
}

// camera ray of rt_cpu_raygen, raytracer_cpu.c:195-231.  `randomise` selects the
// samples > 1 jitter and the defocus offset (both draw random numbers).
__device__ __forceinline__ void camera_ray(const rthip_cast_settings& c, int width, int height, int x, int y, int xs,
                                           int ys, bool randomise, Rng& g, V3& origin, V3& dir) {
    const int samples = (int)c.samples;
    const float x_size = 1.f / (float)(width * samples);
    const float y_size = 1.f / (float)(height * samples);
    float x_norm = ((float)x / (float)width) + ((float)xs / (float)samples) * x_size;
    float y_norm = ((float)y / (float)height) + ((float)ys / (float)samples) * y_size;
    if (randomise && samples > 1) {
        x_norm += rand_unit(g) * x_size;
        y_norm += rand_unit(g) * y_size;
    } else {
        x_norm += 0.5f * x_size;
        y_norm += 0.5f * y_size;
    }
    const V3 ndc = mk(2.f * x_norm - 1.f, 1.f - 2.f * y_norm, 1.f);
    const V3 view = mul(ndc, mk(c.viewport[0], c.viewport[1], c.viewport[2]));
    const V3 eye = mk(c.eye[0], c.eye[1], c.eye[2]);
    const V3 right = mk(c.right[0], c.right[1], c.right[2]);
    const V3 up = mk(c.up[0], c.up[1], c.up[2]);
    const V3 fwd = mk(c.forward[0], c.forward[1], c.forward[2]);
    const V3 sample = add(add(add(eye, scale(right, view.x)), scale(up, view.y)), scale(fwd, view.z));
    origin = (!c.orthographic) ? eye : sub(sample, fwd);
    if (randomise && c.defocus) {
        float dx, dy;
        for (;;) {  // rand_unit_sphere_2f32, common_math.c:10-16: positive quadrant only
            dx = rand_unit(g);
            dy = rand_unit(g);
            if (dx * dx + dy * dy <= 1.f) break;
        }
        dx = dx * c.defocus_disk[0];
        dy = dy * c.defocus_disk[1];
        origin = add(add(origin, scale(right, dx)), scale(up, dy));
    }
    dir = normalize(sub(sample, origin));
}

// rt_cpu_miss, raytracer_cpu.c:354-371 (pow(y, 0.5) evaluated as a correctly rounded sqrt)
__device__ __forceinline__ V3 miss_radiance(const DevScene& S, V3 d) {
    if (!S.sky) return mk(0.f, 0.f, 0.f);
    float y = (d.y < 0.f) ? 0.f : ((d.y > 1.f) ? 1.f : d.y);
    float t = sqrtf(y);
    V3 sky = mk((1.f - t) * 1.f + t * 0.5f, (1.f - t) * 1.f + t * 0.7f, (1.f - t) * 1.f + t * 1.f);
    return scale(sky, 0.7f + 0.25f * y);
}

// One bounce of rt_cpu_closest_hit / rt_cpu_miss, raytracer_cpu.c:259-371, for a ray whose query
// has ended.  Either starts the next ray of the path (returns false; o, d hold the new ray) or ends
// the path (returns true; L holds the terminal radiance before it is folded back through the
// stack).  The reference returns albedo * L_in + emissive from the inside out; to keep that
// order of operations the materials of the scattering bounces are recorded on a per-lane stack
// in LDS and the radiance is folded back through it once the path has ended (path_fold).
struct PathState {
    uint32_t depth;   // bounces left (rt_cpu_trace_ray's depth argument)
    uint32_t sp;      // entries on the material stack
    float ior_top, ior_below;  // RT_CPU_TraceContext (raytracer_cpu.h:52-56): only the top two entries
                               // of its ior stack are ever read, and pops happen after the recursion
    Rng g;
};

__device__ __forceinline__ bool path_bounce(const DevScene& S, PathState& P, const Hit& h, V3& o, V3& d, V3& L,
                                            uint32_t* stack, uint32_t stride) {
    if (h.inst == 0) {
        L = miss_radiance(S, d);
        return true;
    }
    V3 p, n;
    int material;
    finish_hit(S, o, d, h, p, n, material);
    if (material < 0) {  // zero handle, raytracer_cpu.c:264-266
        L = mk(1.f, 1.f, 1.f);
        return true;
    }
    const rthip_material* mat = &S.mats[material];
    const uint32_t type = mat->type;
    if (mat->billboard && dot(n, d) > 0.f) n = scale(n, -1.f);
    if (type == RTHIP_MAT_LAMBERTIAN) {
        o = add(p, scale(n, kSurfaceOffset));
        d = cosine_sample(n, P.g);
        stack[P.sp * stride] = (uint32_t)material;
        P.sp++;
    } else if (type == RTHIP_MAT_DIELECTRIC) {
        float idotn = -dot(n, d);
        bool backface = idotn < 0.f;
        V3 n_corr = scale(n, backface ? -1.f : 1.f);
        float eta_i = P.ior_top;
        float eta_t = backface ? mat->ior : P.ior_below;
        float eta = eta_i / eta_t;
        bool tir = sqrtf(1.f - idotn * idotn) * eta > 1.f;
        bool refl = tir;
        if (!tir) {  // rt_cpu_fresnel_schlick, :560-564, pow(x, 5) by multiplication
            float sr0 = (eta_i - eta_t) / (eta_i + eta_t);
            float r0 = sr0 * sr0;
            float x = 1.f - fabsf(idotn);
            float x2 = x * x;
            float f = r0 + (1.f - r0) * (x2 * x2 * x);
            refl = f > rand_unit(P.g);
        }
        if (refl) {
            o = add(p, scale(n_corr, kSurfaceOffset));
            d = reflect(d, n_corr);
        } else {
            o = add(p, scale(n_corr, -kSurfaceOffset));
            d = refract(d, n_corr, eta);
            P.ior_below = P.ior_top;
            P.ior_top = eta_t;
        }
        stack[P.sp * stride] = (uint32_t)material;
        P.sp++;
    } else if (type == RTHIP_MAT_METAL) {
        V3 i = reflect(d, n);
        V3 r;
        for (;;) {  // rand_unit_sphere_3f32, common_math.c:54-60: positive octant only
            r.x = rand_unit(P.g);
            r.y = rand_unit(P.g);
            r.z = rand_unit(P.g);
            if (dot(r, r) <= 1.f) break;
        }
        i = add(i, scale(r, mat->roughness));
        o = add(p, scale(n, kSurfaceOffset));
        d = normalize(i);
    } else if (type == RTHIP_MAT_NORMAL) {  // rt_cpu_normal_to_radiance, :565-567
        L = scale(add(n, mk(1.f, 1.f, 1.f)), 0.5f);
        return true;
    } else {  // RTHIP_MAT_LIGHT
        L = mk(mat->emissive[0], mat->emissive[1], mat->emissive[2]);
        return true;
    }
    P.depth--;
    if (P.depth == 0) {  // rt_cpu_trace_ray with depth 0 returns black (:247-249)
        L = mk(0.f, 0.f, 0.f);
        return true;
    }
    return false;
}

__device__ __forceinline__ V3 path_fold(const DevScene& S, PathState& P, V3 L, const uint32_t* stack, uint32_t stride) {
    while (P.sp > 0) {
        P.sp--;
        const rthip_material* mat = &S.mats[stack[P.sp * stride]];
        const V3 e = mk(mat->emissive[0], mat->emissive[1], mat->emissive[2]);
        if (mat->type == RTHIP_MAT_LAMBERTIAN)
            L = add(mul(mk(mat->albedo[0], mat->albedo[1], mat->albedo[2]), L), e);  // :286-289
        else
            L = add(L, e);  // dielectric, :327
    }
    return L;
}

__device__ __forceinline__ void flush_counters(uint64_t* counters, const Counters& c, uint32_t rays) {
    // one atomic per wave and counter
    uint32_t v[6] = {rays, c.nodes, c.tris, c.insts, c.wave_steps, c.lane_steps};
#pragma unroll
    for (int k = 0; k < 6; k++) {
        uint32_t x = v[k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd((unsigned long long*)&counters[k], (unsigned long long)x);
    }
}

}  // namespace
