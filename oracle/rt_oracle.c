/*
 * TEST INFRASTRUCTURE -- CPU restatement ("oracle") of the hot path of
 * patricklbell/raytracer: LBVH build, TLAS/BLAS traversal, intersection,
 * shading.  Written from the behaviour of the reference, function by function;
 * every function names the reference lines it follows (paths relative to the
 * reference repository).  Not used by the product; see rt_oracle.h.
 *
 * Floating point: plain IEEE binary32 in the order the reference writes it.
 * Build with -ffp-contract=off and without -ffast-math (oracle/Makefile).
 */
#include "rt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* src/common/common_math.h:286-288 */
#define ORA_PI 3.1415926535897932384626433832795028841971693993751058209749445923078164062f
#define ORA_EPS (1.1920929E-7f)
#define ORA_MAXF 3.402823466e+38f
/* src/raytracer/cpu/raytracer_cpu.c:257 */
#define ORA_SURFACE_OFFSET 0.001f
/* src/raytracer/raytracer_core.h:161 */
#define ORA_MAX_BOUNCES 64

typedef struct { float x, y, z; } v3;
typedef struct { float x, y, z, w; } v4;

/* ---- vector helpers, src/common/common_math.c:18-35,91 ------------------ */
static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vscale(v3 a, float b) { return V3(a.x * b, a.y * b, a.z * b); }
static inline v3 vmul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vdiv(v3 a, v3 b) { return V3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float vlen(v3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
static inline v3 vnormalize(v3 a) { float l = vlen(a); return V3(a.x / l, a.y / l, a.z / l); }
/* common_math.c:29 -- note the negated middle component */
static inline v3 vcross(v3 a, v3 b) {
    return V3(a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x);
}
/* Min/Max are the ternaries of common_core.h:113-114: on a tie (or NaN) the
 * SECOND operand is returned. */
static inline float fmin_ref(float a, float b) { return (a < b) ? a : b; }
static inline float fmax_ref(float a, float b) { return (a > b) ? a : b; }
static inline v3 vmin(v3 a, v3 b) { return V3(fmin_ref(a.x, b.x), fmin_ref(a.y, b.y), fmin_ref(a.z, b.z)); }
static inline v3 vmax(v3 a, v3 b) { return V3(fmax_ref(a.x, b.x), fmax_ref(a.y, b.y), fmax_ref(a.z, b.z)); }
/* common_math.c:30 */
static inline v3 vreflect(v3 i, v3 n) { return vsub(i, vscale(n, 2.f * vdot(i, n))); }
/* common_math.c:91 */
static inline v3 rot_quat(v3 p, v4 q) {
    v3 qv = V3(q.x, q.y, q.z);
    v3 a = vscale(p, q.w * q.w - vdot(qv, qv));
    v3 b = vscale(qv, 2.f * vdot(qv, p));
    v3 c = vscale(vcross(qv, p), 2.f * q.w);
    return vadd(vadd(a, b), c);
}
static inline v4 inv_quat(v4 q) { v4 r = {-q.x, -q.y, -q.z, q.w}; return r; }
/* common_math.c:37-46 */
static inline v3 vorthogonal(v3 x) {
    v3 a = V3(fabsf(x.x), fabsf(x.y), fabsf(x.z));
    float mx = (float)((a.x <= a.y) & (a.x <= a.z));
    float my = (float)((a.y < a.x) & (a.y <= a.z));
    float mz = 1.f - mx - my;
    return vcross(x, V3(mx, my, mz));
}
/* common_math.c:47-53 */
static inline v3 vrefract(v3 i, v3 n, float eta) {
    float ndoti = vdot(n, i);
    float k = 1.f - eta * eta * (1.f - ndoti * ndoti);
    if (k < 0.f) return V3(0.f, 0.f, 0.f);
    return vsub(vscale(i, eta), vscale(n, eta * ndoti + sqrtf(k)));
}

/* ======================================================================== */
/* LBVH build, src/lbvh/lbvh.c                                               */
/* ======================================================================== */

/* lbvh_c_to_morton_code, lbvh.c:74-90.  norm is fp32; the scale by 2^21 happens
 * in double and is exact; the cast to u32 of a NaN (0/0 on a zero-extent axis) is
 * what x86-64 produces for it: 0.  Only bits 0..20 of each axis are used, so
 * norm == 1.0 (x == 2^21) contributes 0. */
static uint32_t morton_axis(float num, float ext) {
    float norm = num / ext;
    double scaled = (double)norm * (double)(1 << 21);
    if (!(scaled == scaled)) return 0u;       /* NaN */
    if (scaled < 0.0) return 0u;              /* cannot happen for centroids inside their bounds */
    if (scaled >= 4294967296.0) return 0u;    /* +inf: cvttsd2si gives 0x8000000000000000, low word 0 */
    return (uint32_t)scaled;
}
static uint64_t morton_from(v3 c, v3 mn, v3 ext) {
    uint32_t x = morton_axis(c.x - mn.x, ext.x);
    uint32_t y = morton_axis(c.y - mn.y, ext.y);
    uint32_t z = morton_axis(c.z - mn.z, ext.z);
    uint64_t code = 0;
    for (uint32_t i = 0; i < 21; i++) {
        code |= (uint64_t)((x >> i) & 1u) << (i * 3 + 0);
        code |= (uint64_t)((y >> i) & 1u) << (i * 3 + 1);
        code |= (uint64_t)((z >> i) & 1u) << (i * 3 + 2);
    }
    return code;
}
uint64_t ora_morton(const float* c, const float* mn, const float* ext) {
    return morton_from(V3(c[0], c[1], c[2]), V3(mn[0], mn[1], mn[2]), V3(ext[0], ext[1], ext[2]));
}

typedef struct { uint64_t code; uint32_t id; } key_id; /* NodeIndex, lbvh.c:5-8 */

/* The reference sorts with an unstable comparison sort (radsort.h:426-606; for
 * n <= 28 it is a stable insertion sort, :341-369).  The order inside a run of
 * equal codes is therefore an artefact; the canonical order here (and in the HIP
 * library) is ascending id inside a run: an LSD radix sort over the code. */
static void sort_keys(key_id* a, key_id* tmp, uint64_t n) {
    for (int pass = 0; pass < 8; pass++) {
        uint64_t hist[257] = {0};
        int shift = pass * 8;
        for (uint64_t i = 0; i < n; i++) hist[((a[i].code >> shift) & 255u) + 1]++;
        if (hist[((a[0].code >> shift) & 255u) + 1] == n) continue; /* all equal in this digit */
        for (int d = 0; d < 256; d++) hist[d + 1] += hist[d];
        for (uint64_t i = 0; i < n; i++) tmp[hist[(a[i].code >> shift) & 255u]++] = a[i];
        memcpy(a, tmp, n * sizeof(key_id));
    }
}

/* lbvh_get_split_idx, lbvh.c:15-42 */
static uint64_t split_index(const key_id* a, uint64_t count) {
    uint64_t f = a[0].code, l = a[count - 1].code;
    if (f == l) return count / 2;
    uint64_t bit = 1ull << (63 - __builtin_clzll(f ^ l));
    uint64_t lo = 0, hi = count;
    while (lo < hi) {
        uint64_t mid = (lo + hi) / 2;
        if (a[mid].code & bit) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}

/* lbvh_create_subtree, lbvh.c:44-72, emitting pre-order records instead of
 * arena nodes.  Iterative with an explicit stack so that 10^7 primitives do not
 * depend on the C stack; the visiting order is the reference's: node, left, right. */
typedef struct { uint64_t start, count; uint64_t node; int state; } build_frame;

static void build_tree(const key_id* a, const float* aabb6, uint64_t n, rthip_node* out) {
    uint64_t next = 0;
    uint64_t cap = 256, top = 0;
    build_frame* st = (build_frame*)malloc(cap * sizeof(build_frame));
    st[top++] = (build_frame){0, n, 0, 0};
    while (top) {
        build_frame* f = &st[top - 1];
        if (f->state == 0) {
            f->node = next++;
            if (f->count == 1) {
                uint32_t id = a[f->start].id;
                const float* b = aabb6 + 6 * (uint64_t)(id - 1);
                rthip_node* nd = &out[f->node];
                nd->aabb_min[0] = b[0]; nd->aabb_min[1] = b[1]; nd->aabb_min[2] = b[2];
                nd->aabb_max[0] = b[3]; nd->aabb_max[1] = b[4]; nd->aabb_max[2] = b[5];
                nd->id = id;
                nd->skip = (uint32_t)next;
                top--;
                continue;
            }
            uint64_t m = split_index(a + f->start, f->count);
            f->state = 1;
            if (top + 2 > cap) {
                cap *= 2;
                st = (build_frame*)realloc(st, cap * sizeof(build_frame));
                f = &st[top - 1];
            }
            /* right is pushed first so that left is processed first */
            build_frame right = {f->start + m, f->count - m, 0, 0};
            build_frame left = {f->start, m, 0, 0};
            st[top++] = right;
            st[top++] = left;
        } else {
            /* both children done: left child = node+1, right child = left.skip.
             * merge_rng3_f32(left, right), common_math.c:411 */
            rthip_node* nd = &out[f->node];
            const rthip_node* l = &out[f->node + 1];
            const rthip_node* r = &out[l->skip];
            for (int k = 0; k < 3; k++) {
                nd->aabb_min[k] = fmin_ref(l->aabb_min[k], r->aabb_min[k]);
                nd->aabb_max[k] = fmax_ref(l->aabb_max[k], r->aabb_max[k]);
            }
            nd->id = 0;
            nd->skip = (uint32_t)next;
            top--;
        }
    }
    free(st);
}

/* Which bounds lbvh_make normalises the centroids by: 0 = the centroids' own (release builds, lbvh.c:97-100),
 * 1 = the boxes' (BUILD_DEBUG builds, lbvh.c:101-104).  Process-wide; every tree built afterwards follows it. */
static int g_box_bounds = 0;
void ora_set_morton_bounds(int boxes) { g_box_bounds = boxes ? 1 : 0; }

/* lbvh_make, lbvh.c:92-125 */
int ora_lbvh_build(const float* aabb6, uint64_t n, uint64_t* out_codes, uint64_t* out_codes_sorted,
                   uint32_t* out_ids_sorted, rthip_node* out_nodes) {
    if (n == 0) return 1;
    v3 mn = V3(ORA_MAXF, ORA_MAXF, ORA_MAXF), mx = V3(-ORA_MAXF, -ORA_MAXF, -ORA_MAXF);
    for (uint64_t i = 0; i < n; i++) {
        const float* b = aabb6 + 6 * i;
        if (g_box_bounds) {
            mn = vmin(mn, V3(b[0], b[1], b[2]));
            mx = vmax(mx, V3(b[3], b[4], b[5]));
        } else {
            v3 c = vscale(vadd(V3(b[0], b[1], b[2]), V3(b[3], b[4], b[5])), 0.5f);
            mn = vmin(mn, c);
            mx = vmax(mx, c);
        }
    }
    v3 ext = vsub(mx, mn);
    key_id* keys = (key_id*)malloc(n * sizeof(key_id));
    key_id* tmp = (key_id*)malloc(n * sizeof(key_id));
    if (!keys || !tmp) { free(keys); free(tmp); return 2; }
    for (uint64_t i = 0; i < n; i++) {
        const float* b = aabb6 + 6 * i;
        v3 c = vscale(vadd(V3(b[0], b[1], b[2]), V3(b[3], b[4], b[5])), 0.5f);
        keys[i].code = morton_from(c, mn, ext);
        keys[i].id = (uint32_t)(i + 1);
        if (out_codes) out_codes[i] = keys[i].code;
    }
    sort_keys(keys, tmp, n);
    for (uint64_t i = 0; i < n; i++) {
        if (out_codes_sorted) out_codes_sorted[i] = keys[i].code;
        if (out_ids_sorted) out_ids_sorted[i] = keys[i].id;
    }
    if (out_nodes) build_tree(keys, aabb6, n, out_nodes);
    free(keys);
    free(tmp);
    return 0;
}

/* lvbh_dump_node, src/lbvh/extra/dump.c:1-13: u8 present, u64 id, f32 min[3],
 * f32 max[3], then left, right; a null child is a single 0 byte.  Returns the
 * number of bytes the dump needs; writes only if it fits. */
uint64_t ora_lbvh_dump(const rthip_node* nodes, uint64_t count, uint8_t* out, uint64_t cap) {
    /* 33 bytes per node + two null bytes per leaf; leaves = (count+1)/2 */
    uint64_t need = count * 33 + (count + 1) / 2 * 2;
    if (!out || need > cap) return need;
    uint64_t w = 0;
    for (uint64_t i = 0; i < count; i++) {
        out[w++] = 1;
        uint64_t id = nodes[i].id;
        memcpy(out + w, &id, 8); w += 8;
        memcpy(out + w, nodes[i].aabb_min, 12); w += 12;
        memcpy(out + w, nodes[i].aabb_max, 12); w += 12;
        if (nodes[i].id != 0) { out[w++] = 0; out[w++] = 0; }
    }
    return w;
}

/* ======================================================================== */
/* intersection                                                              */
/* ======================================================================== */

typedef struct { v3 o, d; } ray_t;

/* lbvh_aabb_query_ray, lbvh.c:127-149: works on a copy of the interval */
static inline int slab_test(const float* bmin, const float* bmax, const ray_t* r, float tmin, float tmax) {
    const float o[3] = {r->o.x, r->o.y, r->o.z};
    const float d[3] = {r->d.x, r->d.y, r->d.z};
    for (int axis = 0; axis < 3; axis++) {
        const float adinv = 1.f / d[axis];
        float t0 = (bmin[axis] - o[axis]) * adinv;
        float t1 = (bmax[axis] - o[axis]) * adinv;
        if (t0 < t1) {
            if (t0 > tmin) tmin = t0;
            if (t1 < tmax) tmax = t1;
        } else {
            if (t1 > tmin) tmin = t1;
            if (t0 < tmax) tmax = t0;
        }
        if (tmax < tmin) return 0;
    }
    return 1;
}
int ora_slab(const float* aabb6, const float* ray6, const float* interval2) {
    ray_t r = {V3(ray6[0], ray6[1], ray6[2]), V3(ray6[3], ray6[4], ray6[5])};
    return slab_test(aabb6, aabb6 + 3, &r, interval2[0], interval2[1]);
}

/* geo_in_interval, src/geo/geo.c:32-34 */
static inline int in_interval(float x, float lo, float hi) { return x >= lo && x <= hi; }

/* geo_intersect_tri, geo.c:72-109 */
static inline int tri_test(const ray_t* r, v3 a, v3 b, v3 c, float tmin, float* tmax, float* u_out, float* v_out) {
    v3 edge1 = vsub(b, a);
    v3 edge2 = vsub(c, a);
    v3 ray_cross_e2 = vcross(r->d, edge2);
    float det = vdot(edge1, ray_cross_e2);
    if (det == 0.f) return 0;
    float inv_det = 1.f / det;
    v3 s = vsub(r->o, a);
    float u = inv_det * vdot(s, ray_cross_e2);
    if ((u < 0 && fabsf(u) > ORA_EPS) || (u > 1 && fabsf(u - 1) > ORA_EPS)) return 0;
    v3 s_cross_e1 = vcross(s, edge1);
    float v = inv_det * vdot(r->d, s_cross_e1);
    if ((v < 0 && fabsf(v) > ORA_EPS) || (u + v > 1 && fabsf(u + v - 1) > ORA_EPS)) return 0;
    float t = inv_det * vdot(edge2, s_cross_e1);
    if (!in_interval(t, tmin, *tmax)) return 0;
    *tmax = t;
    *u_out = u;
    *v_out = v;
    return 1;
}
int ora_intersect_tri(const float* ray6, const float* a, const float* b, const float* c, float* interval2,
                      float* uv2) {
    ray_t r = {V3(ray6[0], ray6[1], ray6[2]), V3(ray6[3], ray6[4], ray6[5])};
    return tri_test(&r, V3(a[0], a[1], a[2]), V3(b[0], b[1], b[2]), V3(c[0], c[1], c[2]), interval2[0],
                    &interval2[1], &uv2[0], &uv2[1]);
}

/* geo_intersect_sphere, geo.c:36-69 */
static inline int sphere_test(const ray_t* r, v3 center, float radius, float tmin, float* tmax) {
    v3 o_minus_c = vsub(r->o, center);
    float a = vdot(r->d, r->d);
    float half_b = vdot(r->d, o_minus_c);
    float c = vdot(o_minus_c, o_minus_c) - radius * radius;
    float qtr_d = half_b * half_b - a * c;
    if (qtr_d < 0.f) return 0;
    float half_sqrt_d = sqrtf(qtr_d);
    float t = (-half_b - half_sqrt_d) / a;
    if (!in_interval(t, tmin, *tmax)) {
        t = (-half_b + half_sqrt_d) / a;
        if (!in_interval(t, tmin, *tmax)) return 0;
    }
    *tmax = t;
    return 1;
}
int ora_intersect_sphere(const float* ray6, const float* center, float radius, float* interval2) {
    ray_t r = {V3(ray6[0], ray6[1], ray6[2]), V3(ray6[3], ray6[4], ray6[5])};
    return sphere_test(&r, V3(center[0], center[1], center[2]), radius, interval2[0], &interval2[1]);
}

/* rt_cpu_transform_point / _aabb, raytracer_cpu.c:569-571,583-606 */
static inline v3 transform_point(v3 p, v3 t, v4 q, v3 s) { return vadd(rot_quat(vmul(p, s), q), t); }
static void transform_aabb(const float* b, v3 t, v4 q, v3 s, float* out6) {
    v3 mn = V3(ORA_MAXF, ORA_MAXF, ORA_MAXF), mx = V3(-ORA_MAXF, -ORA_MAXF, -ORA_MAXF);
    for (int i = 0; i < 8; i++) {
        /* corner order of raytracer_cpu.c:584-593: x is the slowest bit, z the fastest */
        v3 corner = V3((i & 4) ? b[3] : b[0], (i & 2) ? b[4] : b[1], (i & 1) ? b[5] : b[2]);
        v3 p = transform_point(corner, t, q, s);
        mn = vmin(mn, p);
        mx = vmax(mx, p);
    }
    out6[0] = mn.x; out6[1] = mn.y; out6[2] = mn.z;
    out6[3] = mx.x; out6[4] = mx.y; out6[5] = mx.z;
}
void ora_transform_aabb(const float* aabb6, const float* t3, const float* q4, const float* s3, float* out6) {
    v4 q = {q4[0], q4[1], q4[2], q4[3]};
    transform_aabb(aabb6, V3(t3[0], t3[1], t3[2]), q, V3(s3[0], s3[1], s3[2]), out6);
}

/* ======================================================================== */
/* scene                                                                     */
/* ======================================================================== */

typedef struct {
    uint8_t* verts; /* private copy of the interleaved buffer */
    uint32_t vertex_count, stride, pos_off;
    int32_t nrm_off;
    uint32_t* indices; /* NULL: auto-index */
    uint32_t index_count;
    uint64_t n_tris;
    rthip_node* nodes;
    uint64_t n_nodes;
} ora_mesh;

struct ora_scene {
    uint32_t max_bounces;
    int sky;
    ora_mesh* meshes;
    uint32_t n_meshes;
    rthip_instance* insts;
    uint32_t n_insts;
    rthip_material* mats;
    uint32_t n_mats;
    rthip_node* tlas;
    uint64_t n_tlas;
};

static inline v3 mesh_pos(const ora_mesh* m, uint32_t vi) {
    const float* p = (const float*)(m->verts + (size_t)m->pos_off + (size_t)vi * m->stride);
    return V3(p[0], p[1], p[2]);
}
static inline v3 mesh_nrm(const ora_mesh* m, uint32_t vi) {
    const float* p = (const float*)(m->verts + (size_t)m->nrm_off + (size_t)vi * m->stride);
    return V3(p[0], p[1], p[2]);
}
/* rt_cpu_get_tri index rule, raytracer_cpu.c:391-399 */
static inline void mesh_tri_indices(const ora_mesh* m, uint32_t idx, uint32_t* i0, uint32_t* i1, uint32_t* i2) {
    if (m->indices) { *i0 = m->indices[idx]; *i1 = m->indices[idx + 1]; *i2 = m->indices[idx + 2]; }
    else { *i0 = idx; *i1 = idx + 1; *i2 = idx + 2; }
}

/* triangle boxes, rt_cpu_aabb_from_tri + loops, raytracer_cpu.c:54-87 */
static void tri_aabbs(const ora_mesh* m, float* out) {
    for (uint64_t t = 0; t < m->n_tris; t++) {
        uint32_t i0, i1, i2;
        mesh_tri_indices(m, (uint32_t)(t * 3), &i0, &i1, &i2);
        v3 a = mesh_pos(m, i0), b = mesh_pos(m, i1), c = mesh_pos(m, i2);
        v3 mn = vmin(vmin(a, b), c), mx = vmax(vmax(a, b), c);
        float* o = out + 6 * t;
        o[0] = mn.x; o[1] = mn.y; o[2] = mn.z; o[3] = mx.x; o[4] = mx.y; o[5] = mx.z;
    }
}
void ora_tri_aabbs(const ora_mesh_desc* d, float* out_aabb6) {
    ora_mesh m;
    memset(&m, 0, sizeof m);
    m.verts = (uint8_t*)d->verts;
    m.vertex_count = d->vertex_count;
    m.stride = d->stride_bytes;
    m.pos_off = d->pos_off;
    m.nrm_off = d->nrm_off;
    m.indices = (d->indices && d->index_count) ? (uint32_t*)d->indices : NULL;
    m.index_count = d->index_count;
    m.n_tris = m.indices ? d->index_count / 3 : d->vertex_count / 3;
    tri_aabbs(&m, out_aabb6);
}

/* rt_cpu_build_blas + rt_cpu_build_tlas, raytracer_cpu.c:101-174 */
ora_scene* ora_scene_create(uint32_t max_bounces, int sky, const ora_mesh_desc* meshes, uint32_t n_meshes,
                            const rthip_instance* insts, uint32_t n_insts, const rthip_material* mats,
                            uint32_t n_mats) {
    ora_scene* s = (ora_scene*)calloc(1, sizeof *s);
    s->max_bounces = max_bounces;
    s->sky = sky;
    s->n_meshes = n_meshes;
    s->meshes = (ora_mesh*)calloc(n_meshes ? n_meshes : 1, sizeof(ora_mesh));
    for (uint32_t k = 0; k < n_meshes; k++) {
        const ora_mesh_desc* d = &meshes[k];
        ora_mesh* m = &s->meshes[k];
        size_t vbytes = (size_t)d->stride_bytes * d->vertex_count;
        m->verts = (uint8_t*)malloc(vbytes ? vbytes : 1);
        memcpy(m->verts, d->verts, vbytes);
        m->vertex_count = d->vertex_count;
        m->stride = d->stride_bytes;
        m->pos_off = d->pos_off;
        m->nrm_off = d->nrm_off;
        if (d->indices && d->index_count) {
            m->indices = (uint32_t*)malloc((size_t)d->index_count * 4);
            memcpy(m->indices, d->indices, (size_t)d->index_count * 4);
            m->index_count = d->index_count;
            m->n_tris = d->index_count / 3;
        } else {
            m->n_tris = d->vertex_count / 3;
        }
        float* boxes = (float*)malloc((size_t)m->n_tris * 24 + 24);
        tri_aabbs(m, boxes);
        m->n_nodes = 2 * m->n_tris - 1;
        m->nodes = (rthip_node*)malloc((size_t)m->n_nodes * sizeof(rthip_node));
        ora_lbvh_build(boxes, m->n_tris, NULL, NULL, NULL, m->nodes);
        free(boxes);
    }
    s->n_mats = n_mats;
    s->mats = (rthip_material*)malloc((n_mats ? n_mats : 1) * sizeof(rthip_material));
    if (n_mats) memcpy(s->mats, mats, n_mats * sizeof(rthip_material));
    s->n_insts = n_insts;
    s->insts = (rthip_instance*)malloc((n_insts ? n_insts : 1) * sizeof(rthip_instance));
    if (n_insts) memcpy(s->insts, insts, n_insts * sizeof(rthip_instance));

    /* rt_cpu_tlas_node_to_aabb, raytracer_cpu.c:131-147 */
    float* boxes = (float*)malloc((size_t)n_insts * 24 + 24);
    for (uint32_t k = 0; k < n_insts; k++) {
        const rthip_instance* it = &insts[k];
        float* o = boxes + 6 * (size_t)k;
        if (it->type == RTHIP_INST_SPHERE) {
            float r = it->sphere_radius;
            o[0] = it->sphere_center[0] - r; o[1] = it->sphere_center[1] - r; o[2] = it->sphere_center[2] - r;
            o[3] = it->sphere_center[0] + r; o[4] = it->sphere_center[1] + r; o[5] = it->sphere_center[2] + r;
        } else {
            const rthip_node* root = &s->meshes[it->blas_id].nodes[0];
            float model[6] = {root->aabb_min[0], root->aabb_min[1], root->aabb_min[2],
                              root->aabb_max[0], root->aabb_max[1], root->aabb_max[2]};
            v4 q = {it->rotation[0], it->rotation[1], it->rotation[2], it->rotation[3]};
            transform_aabb(model, V3(it->translation[0], it->translation[1], it->translation[2]), q,
                           V3(it->scale[0], it->scale[1], it->scale[2]), o);
        }
    }
    if (n_insts) {
        s->n_tlas = 2 * (uint64_t)n_insts - 1;
        s->tlas = (rthip_node*)malloc((size_t)s->n_tlas * sizeof(rthip_node));
        ora_lbvh_build(boxes, n_insts, NULL, NULL, NULL, s->tlas);
    }
    free(boxes);
    return s;
}

void ora_scene_destroy(ora_scene* s) {
    if (!s) return;
    for (uint32_t k = 0; k < s->n_meshes; k++) {
        free(s->meshes[k].verts);
        free(s->meshes[k].indices);
        free(s->meshes[k].nodes);
    }
    free(s->meshes);
    free(s->mats);
    free(s->insts);
    free(s->tlas);
    free(s);
}

int ora_scene_get_tree(const ora_scene* s, int which, rthip_node* out_nodes, uint64_t* out_count) {
    const rthip_node* src;
    uint64_t n;
    if (which < 0) { src = s->tlas; n = s->n_tlas; }
    else if ((uint32_t)which < s->n_meshes) { src = s->meshes[which].nodes; n = s->meshes[which].n_nodes; }
    else return 1;
    if (out_count) *out_count = n;
    if (out_nodes) memcpy(out_nodes, src, (size_t)n * sizeof(rthip_node));
    return 0;
}

/* ---- closest hit query: rt_cpu_intersect, raytracer_cpu.c:402-457 -------- */

typedef struct {
    int hit;
    uint32_t inst; /* 1-based TLAS id */
    uint32_t tri_idx;
    float u, v;
    float t;
} tlas_hit;

/* lbvh_node_query_ray, lbvh.c:151-161, on the pre-order array: the recursion
 * "node, then left, then right" is a linear walk that jumps to `skip` whenever a
 * box is missed or a leaf is done. */
static int blas_query(const ora_mesh* m, const ray_t* r, float tmin, float* tmax, uint32_t* tri_idx, float* u,
                      float* v, ora_counters* cnt) {
    int any = 0;
    uint64_t i = 0, end = m->n_nodes;
    while (i < end) {
        const rthip_node* nd = &m->nodes[i];
        cnt->nodes_tested++;
        if (!slab_test(nd->aabb_min, nd->aabb_max, r, tmin, *tmax)) { i = nd->skip; continue; }
        if (nd->id) {
            /* rt_cpu_blas_node_hit, raytracer_cpu.c:459-492 */
            uint32_t idx = (nd->id - 1) * 3, i0, i1, i2;
            mesh_tri_indices(m, idx, &i0, &i1, &i2);
            cnt->tris_tested++;
            if (tri_test(r, mesh_pos(m, i0), mesh_pos(m, i1), mesh_pos(m, i2), tmin, tmax, u, v)) {
                any = 1;
                *tri_idx = idx;
            }
            i = nd->skip;
        } else {
            i = i + 1;
        }
    }
    return any;
}

static void tlas_query_interval(const ora_scene* s, const ray_t* r, float tmax, tlas_hit* h, ora_counters* cnt) {
    float tmin = ORA_EPS; /* geo_make_pos_interval, geo.c:28-30, with a caller-chosen upper end */
    memset(h, 0, sizeof *h);
    cnt->rays++;
    uint64_t i = 0, end = s->n_tlas;
    while (i < end) {
        const rthip_node* nd = &s->tlas[i];
        cnt->nodes_tested++;
        if (!slab_test(nd->aabb_min, nd->aabb_max, r, tmin, tmax)) { i = nd->skip; continue; }
        if (nd->id) {
            /* rt_cpu_intersect_tlas_node, raytracer_cpu.c:494-531 */
            const rthip_instance* it = &s->insts[nd->id - 1];
            cnt->inst_entered++;
            if (it->type == RTHIP_INST_SPHERE) {
                if (sphere_test(r, V3(it->sphere_center[0], it->sphere_center[1], it->sphere_center[2]),
                                it->sphere_radius, tmin, &tmax)) {
                    h->hit = 1;
                    h->inst = nd->id;
                    h->tri_idx = 0; /* a sphere hit has no triangle or barycentrics */
                    h->u = 0.f;
                    h->v = 0.f;
                }
            } else {
                /* rt_cpu_inv_transform_ray, raytracer_cpu.c:572-580,607-612; the local
                 * direction is not renormalised, so t is shared between the spaces */
                v4 q = {it->rotation[0], it->rotation[1], it->rotation[2], it->rotation[3]};
                v4 qi = inv_quat(q);
                v3 tr = V3(it->translation[0], it->translation[1], it->translation[2]);
                v3 sc = V3(it->scale[0], it->scale[1], it->scale[2]);
                ray_t local;
                local.o = vdiv(rot_quat(vsub(r->o, tr), qi), sc);
                local.d = vdiv(rot_quat(r->d, qi), sc);
                uint32_t tri = 0;
                float u = 0.f, v = 0.f;
                if (blas_query(&s->meshes[it->blas_id], &local, tmin, &tmax, &tri, &u, &v, cnt)) {
                    h->hit = 1;
                    h->inst = nd->id;
                    h->tri_idx = tri;
                    h->u = u;
                    h->v = v;
                }
            }
            i = nd->skip;
        } else {
            i = i + 1;
        }
    }
    h->t = tmax;
}

static void tlas_query(const ora_scene* s, const ray_t* r, tlas_hit* h, ora_counters* cnt) {
    tlas_query_interval(s, r, ORA_MAXF, h, cnt);
}

typedef struct { v3 p, n; float t; int32_t material; } hit_record;

/* epilogue of rt_cpu_intersect, raytracer_cpu.c:411-454 */
static void finish_hit(const ora_scene* s, const ray_t* r, const tlas_hit* h, hit_record* rec) {
    const rthip_instance* it = &s->insts[h->inst - 1];
    rec->t = h->t;
    rec->p = vadd(r->o, vscale(r->d, rec->t));
    rec->material = it->material;
    if (it->type == RTHIP_INST_SPHERE) {
        v3 c = V3(it->sphere_center[0], it->sphere_center[1], it->sphere_center[2]);
        rec->n = vscale(vsub(rec->p, c), 1.f / it->sphere_radius);
    } else {
        const ora_mesh* m = &s->meshes[it->blas_id];
        uint32_t i0, i1, i2;
        mesh_tri_indices(m, h->tri_idx, &i0, &i1, &i2);
        v3 tri_n;
        if (m->nrm_off >= 0) {
            /* U weights vertex 0, V vertex 1 -- as written in the reference (:441-445) */
            v3 n0 = mesh_nrm(m, i0), n1 = mesh_nrm(m, i1), n2 = mesh_nrm(m, i2);
            tri_n = vadd(vadd(vscale(n0, h->u), vscale(n1, h->v)), vscale(n2, 1 - h->u - h->v));
        } else {
            v3 v0 = mesh_pos(m, i0), v1 = mesh_pos(m, i1), v2 = mesh_pos(m, i2);
            tri_n = vcross(vsub(v1, v0), vsub(v2, v0));
        }
        v4 q = {it->rotation[0], it->rotation[1], it->rotation[2], it->rotation[3]};
        v3 sc = V3(it->scale[0], it->scale[1], it->scale[2]);
        rec->n = vnormalize(rot_quat(vmul(tri_n, sc), q)); /* rt_cpu_transform_dir, :575-577 */
    }
}

typedef struct {
    const ora_scene* s;
    const float* o3;
    const float* d3;
    uint64_t begin, end;
    uint32_t* out_inst;
    uint32_t* out_tri;
    float* out_t;
    float* out_uv;
    float* out_normal;
    ora_counters cnt;
} trace_job;

static void* trace_worker(void* arg) {
    trace_job* j = (trace_job*)arg;
    for (uint64_t i = j->begin; i < j->end; i++) {
        ray_t r = {V3(j->o3[3 * i], j->o3[3 * i + 1], j->o3[3 * i + 2]),
                   V3(j->d3[3 * i], j->d3[3 * i + 1], j->d3[3 * i + 2])};
        tlas_hit h;
        tlas_query(j->s, &r, &h, &j->cnt);
        if (j->out_inst) j->out_inst[i] = h.hit ? h.inst : 0;
        if (j->out_tri) j->out_tri[i] = h.hit ? h.tri_idx : 0;
        if (j->out_t) j->out_t[i] = h.hit ? h.t : -1.0f;
        if (j->out_uv) { j->out_uv[2 * i] = h.hit ? h.u : 0.f; j->out_uv[2 * i + 1] = h.hit ? h.v : 0.f; }
        if (j->out_normal) {
            v3 n = V3(0, 0, 0);
            if (h.hit) { hit_record rec; finish_hit(j->s, &r, &h, &rec); n = rec.n; }
            j->out_normal[3 * i] = n.x; j->out_normal[3 * i + 1] = n.y; j->out_normal[3 * i + 2] = n.z;
        }
    }
    return NULL;
}

void ora_scene_trace(const ora_scene* s, const float* o3, const float* d3, uint64_t n, uint32_t* out_inst,
                     uint32_t* out_tri, float* out_t, float* out_uv, float* out_normal, ora_counters* counters,
                     int threads) {
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    trace_job jobs[256];
    pthread_t tids[256];
    uint64_t per = (n + (uint64_t)threads - 1) / (uint64_t)threads;
    for (int k = 0; k < threads; k++) {
        trace_job* j = &jobs[k];
        memset(j, 0, sizeof *j);
        j->s = s; j->o3 = o3; j->d3 = d3;
        j->begin = per * (uint64_t)k < n ? per * (uint64_t)k : n;
        j->end = j->begin + per < n ? j->begin + per : n;
        j->out_inst = out_inst; j->out_tri = out_tri; j->out_t = out_t; j->out_uv = out_uv;
        j->out_normal = out_normal;
        if (threads == 1) trace_worker(j);
        else pthread_create(&tids[k], NULL, trace_worker, j);
    }
    ora_counters total = {0, 0, 0, 0};
    for (int k = 0; k < threads; k++) {
        if (threads > 1) pthread_join(tids[k], NULL);
        total.rays += jobs[k].cnt.rays;
        total.nodes_tested += jobs[k].cnt.nodes_tested;
        total.tris_tested += jobs[k].cnt.tris_tested;
        total.inst_entered += jobs[k].cnt.inst_entered;
    }
    if (counters) *counters = total;
}

/* Occlusion: does lbvh_query_ray (lbvh.c:163-165) over the TLAS with the interval
 * [EPSILON_F32, tmax] report a hit?  The walk up to the first accepted primitive does not
 * depend on what comes after it, so "some primitive is accepted" is what an any-hit walk
 * that stops there computes. */
void ora_scene_occluded(const ora_scene* s, const float* o3, const float* d3, uint64_t n, float tmax, uint8_t* out) {
    ora_counters cnt = {0, 0, 0, 0};
    for (uint64_t i = 0; i < n; i++) {
        ray_t r = {V3(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]), V3(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2])};
        tlas_hit h;
        tlas_query_interval(s, &r, tmax, &h, &cnt);
        out[i] = h.hit ? 1 : 0;
    }
}

/* ======================================================================== */
/* random numbers and the two libm flavours                                  */
/* ======================================================================== */

/* glibc random_r TYPE_3 (x^31 + x^3 + 1), the generator behind rand()/srand() on
 * the reference's platform (call sites: src/common/common_math.h:310,
 * common_math.c:426-428).  glibc 2.35, stdlib/random_r.c: seed 0 is replaced by
 * 1, state filled by the Lehmer step 16807*x mod (2^31-1), 310 outputs dropped. */
typedef struct { int32_t r[31]; int f, b; } glibc_rng;
static void glibc_seed(glibc_rng* g, uint32_t seed) {
    if (seed == 0) seed = 1;
    g->r[0] = (int32_t)seed;
    int32_t word = (int32_t)seed;
    for (int i = 1; i < 31; i++) {
        long hi = word / 127773, lo = word % 127773;
        long w = 16807 * lo - 2836 * hi;
        if (w < 0) w += 2147483647;
        word = (int32_t)w;
        g->r[i] = word;
    }
    g->f = 3;
    g->b = 0;
    for (int i = 0; i < 310; i++) {
        g->r[g->f] = (int32_t)((uint32_t)g->r[g->f] + (uint32_t)g->r[g->b]);
        if (++g->f >= 31) g->f = 0;
        if (++g->b >= 31) g->b = 0;
    }
}
static inline int glibc_next(glibc_rng* g) {
    uint32_t v = (uint32_t)g->r[g->f] + (uint32_t)g->r[g->b];
    g->r[g->f] = (int32_t)v;
    if (++g->f >= 31) g->f = 0;
    if (++g->b >= 31) g->b = 0;
    return (int)(v >> 1);
}
static glibc_rng g_test_rng;
void ora_srand(uint32_t seed) { glibc_seed(&g_test_rng, seed); }
int ora_rand(void) { return glibc_next(&g_test_rng); }

/* Counter-based generator of the HIP kernels: 31 random bits per draw, keyed by (seed, pixel, sample, draw), in 32-bit
 * arithmetic throughout.  hash32 is a two-multiply integer hash; a path's key is two hashed words (a, b), draw d is
 * hash32(a + d * 0x9E3779B9) ^ b. */
static inline uint32_t hash32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
static inline uint64_t counter_key(uint64_t seed, uint32_t pixel, uint32_t sample) {
    /* the seed is hashed before it meets the sample index, and b is a hash of (seed, pixel, sample) of its own */
    const uint32_t lo = (uint32_t)seed, hi = (uint32_t)(seed >> 32);
    const uint32_t s1 = hash32(lo ^ hash32(hi + 0x6a09e667u));
    const uint32_t s2 = hash32(hi ^ hash32(lo + 0xbb67ae85u));
    const uint32_t a = hash32(pixel + hash32(sample + s1));
    const uint32_t b = hash32(sample + hash32(pixel + s2));
    return (uint64_t)a | ((uint64_t)b << 32);
}
static inline uint32_t counter_draw(uint64_t key, uint32_t draw) {
    return (hash32((uint32_t)key + draw * 0x9E3779B9u) ^ (uint32_t)(key >> 32)) >> 1;
}
uint32_t ora_counter_rand(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t draw) {
    return counter_draw(counter_key(seed, pixel, sample), draw);
}

/* sin / cos of the counter flavour: Cody-Waite reduction by pi/2 and the classic
 * single-precision minimax kernels, evaluated in binary32 with no fused
 * operations, so that a GPU and a CPU produce the same bits.  Valid for the
 * range the path tracer needs (0 <= x <= 2*pi). */
static inline void poly_sincos(float x, float* s_out, float* c_out) {
    int k = (int)(x * 0.636619772367581343f + 0.5f);
    float kf = (float)k;
    float r = ((x - kf * 1.5703125f) - kf * 4.837512969970703125e-4f) - kf * 7.54978995489188216e-8f;
    float r2 = r * r;
    float s = r + r * r2 * (-1.6666667163e-01f + r2 * (8.3333337680e-03f + r2 * (-1.9841270114e-04f + r2 * 2.7557314297e-06f)));
    float c = (1.f - 0.5f * r2) + r2 * r2 * (4.1666667908e-02f + r2 * (-1.3888889225e-03f + r2 * (2.4801587642e-05f + r2 * -2.7557314297e-07f)));
    switch (k & 3) {
        case 0: *s_out = s; *c_out = c; break;
        case 1: *s_out = c; *c_out = -s; break;
        case 2: *s_out = -s; *c_out = -c; break;
        default: *s_out = -c; *c_out = s; break;
    }
}
float ora_poly_sin(float x) { float s, c; poly_sincos(x, &s, &c); return s; }
float ora_poly_cos(float x) { float s, c; poly_sincos(x, &s, &c); return c; }

typedef struct {
    int mode;
    glibc_rng* serial;  /* ORA_RNG_GLIBC */
    uint64_t key;       /* ORA_RNG_COUNTER */
    uint32_t draw;
} rng_t;

/* rand_unit_f32, common_math.h:310: (f32)rand()/(f32)RAND_MAX, RAND_MAX rounds to 2^31 */
static inline float rand_unit(rng_t* g) {
    int v = (g->mode == ORA_RNG_GLIBC) ? glibc_next(g->serial) : (int)counter_draw(g->key, g->draw++);
    return (float)v / (float)2147483647;
}
static inline void env_sincos(const rng_t* g, float x, float* s, float* c) {
    if (g->mode == ORA_RNG_GLIBC) { *c = cosf(x); *s = sinf(x); }
    else poly_sincos(x, s, c);
}
static inline float env_sqrt_y(const rng_t* g, float y) { /* pow_f32(y, 0.5f), raytracer_cpu.c:364 */
    return (g->mode == ORA_RNG_GLIBC) ? powf(y, 0.5f) : sqrtf(y);
}
static inline float env_pow5(const rng_t* g, float x) { /* pow_f32(x, 5.f), raytracer_cpu.c:563 */
    if (g->mode == ORA_RNG_GLIBC) return powf(x, 5.f);
    float x2 = x * x;
    return x2 * x2 * x;
}

/* ======================================================================== */
/* path tracing: rt_cpu_raygen / trace_ray / closest_hit / miss              */
/* ======================================================================== */

typedef struct {
    rthip_ray_record* out; /* NULL: count only */
    uint64_t cap, n;
} record_sink;
typedef struct { uint8_t ior_count; float ior[ORA_MAX_BOUNCES]; record_sink* sink; } trace_ctx; /* raytracer_cpu.h:52-56 */

static v3 trace_ray(const ora_scene* s, trace_ctx* ctx, const ray_t* r, int depth, rng_t* g, ora_counters* cnt);

/* rand_unit_sphere_3f32 / _2f32, common_math.c:10-16,54-60: rejection from the
 * positive octant / quadrant only */
static v3 rand_sphere3(rng_t* g) {
    for (;;) {
        v3 p;
        p.x = rand_unit(g); p.y = rand_unit(g); p.z = rand_unit(g);
        if (vdot(p, p) <= 1.f) return p;
    }
}

/* rt_cpu_cosine_sample_hemisphere + rt_cpu_transform_uvw_to_hemisphere, raytracer_cpu.c:536-559 */
static v3 cosine_sample(v3 n, rng_t* g) {
    float r1 = rand_unit(g);
    float r2 = rand_unit(g);
    float phi = 2 * ORA_PI * r1;
    float sn, cs;
    env_sincos(g, phi, &sn, &cs);
    float u = cs * sqrtf(r2);
    float v = sn * sqrtf(r2);
    float w = sqrtf(1 - r2);
    v3 ub = vorthogonal(n);
    v3 vb = vcross(ub, n);
    v3 d = vadd(vadd(vscale(ub, u), vscale(vb, v)), vscale(n, w));
    return vnormalize(d);
}

/* rt_cpu_fresnel_schlick, raytracer_cpu.c:560-564 */
static float schlick(const rng_t* g, float eta_i, float eta_t, float cos_theta) {
    float sqrt_r0 = (eta_i - eta_t) / (eta_i + eta_t);
    float r0 = sqrt_r0 * sqrt_r0;
    return r0 + (1.f - r0) * env_pow5(g, 1.f - cos_theta);
}

/* rt_cpu_miss, raytracer_cpu.c:354-371 */
static v3 miss(const ora_scene* s, const ray_t* r, const rng_t* g) {
    if (!s->sky) return V3(0.f, 0.f, 0.f);
    float y = (r->d.y < 0.f) ? 0.f : ((r->d.y > 1.f) ? 1.f : r->d.y);
    float t = env_sqrt_y(g, y);
    v3 sky = V3((1.f - t) * 1.f + t * 0.5f, (1.f - t) * 1.f + t * 0.7f, (1.f - t) * 1.f + t * 1.f);
    return vscale(sky, 0.7f + 0.25f * y);
}

/* rt_cpu_closest_hit, raytracer_cpu.c:259-352 */
static v3 closest_hit(const ora_scene* s, trace_ctx* ctx, const ray_t* r, int depth, hit_record* rec, rng_t* g,
                      ora_counters* cnt) {
    if (rec->material < 0) return V3(1.f, 1.f, 1.f);
    const rthip_material* mat = &s->mats[rec->material];
    if (mat->billboard && vdot(rec->n, r->d) > 0.f) rec->n = vscale(rec->n, -1.f);
    v3 albedo = V3(mat->albedo[0], mat->albedo[1], mat->albedo[2]);
    v3 emissive = V3(mat->emissive[0], mat->emissive[1], mat->emissive[2]);
    switch (mat->type) {
        case RTHIP_MAT_LAMBERTIAN: {
            ray_t nr;
            nr.o = vadd(rec->p, vscale(rec->n, ORA_SURFACE_OFFSET));
            nr.d = cosine_sample(rec->n, g);
            v3 li = trace_ray(s, ctx, &nr, depth - 1, g, cnt);
            return vadd(vmul(albedo, li), emissive);
        }
        case RTHIP_MAT_DIELECTRIC: {
            float idotn = -vdot(rec->n, r->d);
            int backface = idotn < 0.f;
            v3 n_corr = vscale(rec->n, backface ? -1.f : 1.f);
            float eta_i = ctx->ior[ctx->ior_count];
            int prev = (int)ctx->ior_count - 1;
            float eta_t = backface ? mat->ior : ctx->ior[prev > 0 ? prev : 0];
            float eta = eta_i / eta_t;
            int tir = sqrtf(1 - idotn * idotn) * eta > 1.f;
            int reflect = tir || schlick(g, eta_i, eta_t, fabsf(idotn)) > rand_unit(g);
            ray_t nr;
            if (reflect) {
                nr.o = vadd(rec->p, vscale(n_corr, ORA_SURFACE_OFFSET));
                nr.d = vreflect(r->d, n_corr);
            } else {
                nr.o = vadd(rec->p, vscale(n_corr, -ORA_SURFACE_OFFSET));
                nr.d = vrefract(r->d, n_corr, eta);
                ctx->ior_count++;
                ctx->ior[ctx->ior_count] = eta_t;
            }
            v3 li = trace_ray(s, ctx, &nr, depth - 1, g, cnt);
            if (!reflect) ctx->ior_count--;
            return vadd(li, emissive);
        }
        case RTHIP_MAT_METAL: {
            v3 i = vreflect(r->d, rec->n);
            i = vadd(i, vscale(rand_sphere3(g), mat->roughness));
            ray_t nr;
            nr.o = vadd(rec->p, vscale(rec->n, ORA_SURFACE_OFFSET));
            nr.d = vnormalize(i);
            return trace_ray(s, ctx, &nr, depth - 1, g, cnt);
        }
        case RTHIP_MAT_NORMAL: /* rt_cpu_normal_to_radiance, :565-567 */
            return vscale(vadd(rec->n, V3(1.f, 1.f, 1.f)), 0.5f);
        case RTHIP_MAT_LIGHT:
            return emissive;
        default:
            return V3(0.f, 0.f, 0.f);
    }
}

/* rt_cpu_dump_add_ray_hit_record / _miss_record, src/raytracer/cpu/extra/dump.c:11-32 */
static void sink_add(record_sink* k, const ray_t* r, const hit_record* rec) {
    if (k->out && k->n < k->cap) {
        rthip_ray_record* o = &k->out[k->n];
        memset(o, 0, sizeof *o);
        o->origin[0] = r->o.x, o->origin[1] = r->o.y, o->origin[2] = r->o.z;
        o->direction[0] = r->d.x, o->direction[1] = r->d.y, o->direction[2] = r->d.z;
        if (rec) {
            o->p[0] = rec->p.x, o->p[1] = rec->p.y, o->p[2] = rec->p.z;
            o->n[0] = rec->n.x, o->n[1] = rec->n.y, o->n[2] = rec->n.z;
            o->t = rec->t;
            o->material = rec->material;
        } else {
            o->t = -1.0f;
            o->material = -1;
        }
    }
    k->n++;
}

/* rt_cpu_trace_ray, raytracer_cpu.c:244-255 */
static v3 trace_ray(const ora_scene* s, trace_ctx* ctx, const ray_t* r, int depth, rng_t* g, ora_counters* cnt) {
    if (depth == 0) return V3(0.f, 0.f, 0.f);
    tlas_hit h;
    tlas_query(s, r, &h, cnt);
    if (h.hit) {
        hit_record rec;
        finish_hit(s, r, &h, &rec);
        if (ctx->sink) sink_add(ctx->sink, r, &rec); /* first statement of rt_cpu_closest_hit, :260-262 */
        return closest_hit(s, ctx, r, depth, &rec, g, cnt);
    }
    if (ctx->sink) sink_add(ctx->sink, r, NULL); /* first statement of rt_cpu_miss, :355-357 */
    return miss(s, r, g);
}

/* one camera sample of rt_cpu_raygen, raytracer_cpu.c:195-236 */
static v3 camera_sample(const ora_scene* s, const rthip_cast_settings* c, int width, int height, int x, int y,
                        int xs, int ys, rng_t* g, ora_counters* cnt, ray_t* out_ray, record_sink* sink) {
    int samples = (int)c->samples;
    float x_size = 1.f / (float)(width * samples);
    float y_size = 1.f / (float)(height * samples);
    float x_norm = ((float)x / width) + ((float)xs / samples) * x_size;
    float y_norm = ((float)y / height) + ((float)ys / samples) * y_size;
    if (samples > 1) {
        x_norm += rand_unit(g) * x_size;
        y_norm += rand_unit(g) * y_size;
    } else {
        x_norm += 0.5f * x_size;
        y_norm += 0.5f * y_size;
    }
    v3 ndc = V3(2.f * x_norm - 1.f, 1.f - 2.f * y_norm, 1.f);
    v3 view = vmul(ndc, V3(c->viewport[0], c->viewport[1], c->viewport[2]));
    v3 eye = V3(c->eye[0], c->eye[1], c->eye[2]);
    v3 right = V3(c->right[0], c->right[1], c->right[2]);
    v3 up = V3(c->up[0], c->up[1], c->up[2]);
    v3 fwd = V3(c->forward[0], c->forward[1], c->forward[2]);
    v3 sample = vadd(vadd(vadd(eye, vscale(right, view.x)), vscale(up, view.y)), vscale(fwd, view.z));
    v3 origin = (!c->orthographic) ? eye : vsub(sample, fwd);
    if (c->defocus) {
        float dx, dy;
        for (;;) { /* rand_unit_sphere_2f32 */
            dx = rand_unit(g);
            dy = rand_unit(g);
            if (dx * dx + dy * dy <= 1.f) break;
        }
        dx = dx * c->defocus_disk[0];
        dy = dy * c->defocus_disk[1];
        origin = vadd(vadd(origin, vscale(right, dx)), vscale(up, dy));
    }
    ray_t r;
    r.o = origin;
    r.d = vnormalize(vsub(sample, origin));
    if (out_ray) { *out_ray = r; return V3(0, 0, 0); }
    trace_ctx ctx;
    memset(&ctx, 0, sizeof ctx);
    ctx.ior[0] = c->ior;
    ctx.sink = sink;
    return trace_ray(s, &ctx, &r, (int)s->max_bounces, g, cnt);
}

void ora_primary_rays(const rthip_cast_settings* cast, int width, int height, float* out_o3, float* out_d3) {
    rthip_cast_settings c = *cast;
    c.samples = 1;
    c.defocus = 0;
    rng_t g = {ORA_RNG_COUNTER, NULL, 0, 0};
    for (int y = 0; y < height; y++)
        for (int x = 0; x < width; x++) {
            ray_t r;
            camera_sample(NULL, &c, width, height, x, y, 0, 0, &g, NULL, &r, NULL);
            size_t i = (size_t)y * width + x;
            out_o3[3 * i] = r.o.x; out_o3[3 * i + 1] = r.o.y; out_o3[3 * i + 2] = r.o.z;
            out_d3[3 * i] = r.d.x; out_d3[3 * i + 1] = r.d.y; out_d3[3 * i + 2] = r.d.z;
        }
}

typedef struct {
    const ora_scene* s;
    const rthip_cast_settings* c;
    uint64_t seed;
    int mode, width, height, y0, y1;
    float* out;
    rthip_shard shard;
    glibc_rng* serial;
    ora_counters cnt;
    record_sink* sink;
} cast_job;

static void* cast_worker(void* arg) {
    cast_job* j = (cast_job*)arg;
    int samples = (int)j->c->samples;
    float inv_sample_count = 1.f / ((float)samples * samples);
    int tiles_x = (j->width + 63) / 64;
    for (int y = j->y0; y < j->y1; y++) {
        for (int x = 0; x < j->width; x++) {
            float* px = j->out + 3 * ((size_t)y * j->width + x);
            v3 c = V3(0.f, 0.f, 0.f);
            if (j->shard.mode == 1) {
                uint32_t tile = (uint32_t)((y / 64) * tiles_x + (x / 64));
                if (tile % j->shard.world != j->shard.rank) { px[0] = px[1] = px[2] = 0.f; continue; }
            }
            for (int ys = 0; ys < samples; ys++)
                for (int xs = 0; xs < samples; xs++) {
                    uint32_t sidx = (uint32_t)(ys * samples + xs);
                    if (j->shard.mode == 2 && sidx % j->shard.world != j->shard.rank) continue;
                    rng_t g;
                    g.mode = j->mode;
                    g.serial = j->serial;
                    g.key = counter_key(j->seed, (uint32_t)(y * j->width + x), sidx);
                    g.draw = 0;
                    c = vadd(c, camera_sample(j->s, j->c, j->width, j->height, x, y, xs, ys, &g, &j->cnt, NULL, j->sink));
                }
            if (j->shard.mode != 2) c = vscale(c, inv_sample_count);
            px[0] = c.x; px[1] = c.y; px[2] = c.z;
        }
    }
    return NULL;
}

/* rt_cpu_raygen, raytracer_cpu.c:179-242.  ORA_RNG_GLIBC runs serially (the
 * reference's random stream is one global sequence); ORA_RNG_COUNTER may use
 * `threads` row bands. */
void ora_scene_cast(const ora_scene* s, const rthip_cast_settings* cast, uint64_t seed, int rng_mode,
                    float* out_rgb, int width, int height, const rthip_shard* shard, int threads,
                    ora_counters* counters) {
    glibc_rng serial;
    if (rng_mode == ORA_RNG_GLIBC) { threads = 1; glibc_seed(&serial, (uint32_t)seed); }
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    if (threads > height) threads = height;
    cast_job jobs[256];
    pthread_t tids[256];
    int per = (height + threads - 1) / threads;
    for (int k = 0; k < threads; k++) {
        cast_job* j = &jobs[k];
        memset(j, 0, sizeof *j);
        j->s = s; j->c = cast; j->seed = seed; j->mode = rng_mode;
        j->width = width; j->height = height;
        j->y0 = per * k < height ? per * k : height;
        j->y1 = j->y0 + per < height ? j->y0 + per : height;
        j->out = out_rgb;
        if (shard) j->shard = *shard;
        if (j->shard.mode != 0 && j->shard.world == 0) j->shard.mode = 0;
        j->serial = &serial;
        if (threads == 1) cast_worker(j);
        else pthread_create(&tids[k], NULL, cast_worker, j);
    }
    ora_counters total = {0, 0, 0, 0};
    for (int k = 0; k < threads; k++) {
        if (threads > 1) pthread_join(tids[k], NULL);
        total.rays += jobs[k].cnt.rays;
        total.nodes_tested += jobs[k].cnt.nodes_tested;
        total.tris_tested += jobs[k].cnt.tris_tested;
        total.inst_entered += jobs[k].cnt.inst_entered;
    }
    if (counters) *counters = total;
}

uint64_t ora_scene_cast_records(const ora_scene* s, const rthip_cast_settings* cast, uint64_t seed, int rng_mode,
                                int width, int height, rthip_ray_record* out, uint64_t cap) {
    glibc_rng serial;
    if (rng_mode == ORA_RNG_GLIBC) glibc_seed(&serial, (uint32_t)seed);
    record_sink sink = {out, cap, 0};
    float* rgb = (float*)malloc((size_t)width * height * 3 * sizeof(float));
    cast_job j;
    memset(&j, 0, sizeof j);
    j.s = s; j.c = cast; j.seed = seed; j.mode = rng_mode;
    j.width = width; j.height = height; j.y0 = 0; j.y1 = height;
    j.out = rgb;
    j.serial = &serial;
    j.sink = &sink;
    cast_worker(&j);
    free(rgb);
    return sink.n;
}

/* get_rt_cast_settings, src/demos/demos_inc.c:104-130 */
void ora_cast_settings(int width, int height, uint32_t samples, const float* eye_, const float* subject_,
                       float defocus_angle, int orthographic, float vfov_in, rthip_cast_settings* out) {
    v3 eye = V3(eye_[0], eye_[1], eye_[2]), subject = V3(subject_[0], subject_[1], subject_[2]);
    v3 d = vsub(subject, eye);
    v3 up = V3(0, 1, 0);
    v3 forward = vnormalize(d);
    v3 right = vcross(forward, up);
    float aspect = (float)width / height;
    float vfov = (vfov_in > 0.f) ? vfov_in : (45 * (ORA_PI / 180.f));
    float focus = vlen(d);
    float h = tanf(vfov / 2.f);
    float plane_h = 2 * h * focus;
    float defocus_radius = focus * tanf(defocus_angle / 2.f);
    memset(out, 0, sizeof *out);
    out->eye[0] = eye.x; out->eye[1] = eye.y; out->eye[2] = eye.z;
    out->up[0] = up.x; out->up[1] = up.y; out->up[2] = up.z;
    out->forward[0] = forward.x; out->forward[1] = forward.y; out->forward[2] = forward.z;
    out->right[0] = right.x; out->right[1] = right.y; out->right[2] = right.z;
    out->viewport[0] = plane_h * aspect; out->viewport[1] = plane_h; out->viewport[2] = focus;
    out->samples = samples;
    out->ior = 1.f;
    out->defocus = defocus_angle > 0.f;
    out->defocus_disk[0] = defocus_radius; out->defocus_disk[1] = defocus_radius;
    out->orthographic = orthographic != 0;
}

/* make_a_to_b_quat, common_math.c:92-100 (used by the cornell demo's add_quad,
 * src/demos/cornell/main.c:33-38, with an unnormalised target vector) */
void ora_a_to_b_quat(const float* a_, const float* b_, float* out4) {
    v3 a = V3(a_[0], a_[1], a_[2]), b = V3(b_[0], b_[1], b_[2]);
    float k = sqrtf(vdot(a, a) * vdot(b, b));
    float k_cos_theta = vdot(a, b);
    float x, y, z, w;
    if (k_cos_theta / k < -1.f + ORA_EPS) {
        v3 o = vnormalize(vorthogonal(a));
        x = o.x; y = o.y; z = o.z; w = 0.f;
    } else {
        v3 cr = vcross(a, b);
        x = cr.x; y = cr.y; z = cr.z; w = k_cos_theta + k;
        float l = sqrtf(x * x + y * y + z * z + w * w);
        x = x / l; y = y / l; z = z / l; w = w / l;
        out4[0] = x; out4[1] = y; out4[2] = z; out4[3] = w;
        return;
    }
    out4[0] = x; out4[1] = y; out4[2] = z; out4[3] = w;
}

/* ---- .hdr output (the reference writes its image with stbi_write_hdr, e.g. src/demos/cornell/main.c:222) ---------- */

/* (unsigned char)(float) as the reference's x86-64 build performs it: truncate to a 32-bit integer (out of range and
 * NaN give INT_MIN), keep the low byte.  The C standard leaves negative and too-large values undefined; the image
 * never holds them, the edge vectors do. */
static uint8_t byte_of_float(float x) {
    int32_t v = (x >= -2147483648.f && x < 2147483648.f) ? (int32_t)x : INT32_MIN;
    return (uint8_t)((uint32_t)v & 0xffu);
}

void ora_rgbe(const float* l, uint8_t* out4) { /* stbiw__linear_to_rgbe, stb_image_write.h:639-654 */
    const float m12 = l[1] > l[2] ? l[1] : l[2];
    const float maxcomp = l[0] > m12 ? l[0] : m12;
    if (maxcomp < 1e-32f) {
        out4[0] = out4[1] = out4[2] = out4[3] = 0;
        return;
    }
    int exponent = 0;
    const float mant = (float)frexp((double)maxcomp, &exponent);
    const float normalize = mant * 256.0f / maxcomp;
    out4[0] = byte_of_float(l[0] * normalize);
    out4[1] = byte_of_float(l[1] * normalize);
    out4[2] = byte_of_float(l[2] * normalize);
    out4[3] = (uint8_t)((uint32_t)(exponent + 128) & 0xffu);
}

typedef struct {
    uint8_t* out;
    uint64_t cap, len;
} hdr_sink;

static void hdr_put(hdr_sink* s, const uint8_t* data, uint64_t n) {
    for (uint64_t i = 0; i < n; i++)
        if (s->len + i < s->cap) s->out[s->len + i] = data[i];
    s->len += n;
}

/* one component of one scanline, stb_image_write.h:722-757: stretches without three equal bytes in a row are copied
 * in packets of at most 128, runs of at least three equal bytes become (128 + length, byte) packets of at most 127 */
static void hdr_rle_component(hdr_sink* s, const uint8_t* comp, int width) {
    int x = 0;
    while (x < width) {
        int r = x;
        while (r + 2 < width && !(comp[r] == comp[r + 1] && comp[r] == comp[r + 2])) r++;
        const int found = r + 2 < width;
        if (!found) r = width;
        while (x < r) {
            const int len = r - x > 128 ? 128 : r - x;
            const uint8_t lb = (uint8_t)len;
            hdr_put(s, &lb, 1);
            hdr_put(s, comp + x, (uint64_t)len);
            x += len;
        }
        if (found) {
            while (r < width && comp[r] == comp[x]) r++;
            while (x < r) {
                const int len = r - x > 127 ? 127 : r - x;
                const uint8_t pk[2] = {(uint8_t)(len + 128), comp[x]};
                hdr_put(s, pk, 2);
                x += len;
            }
        }
    }
}

uint64_t ora_hdr_encode(const float* rgb, int width, int height, uint8_t* out, uint64_t cap) { /* :761-785 */
    if (width <= 0 || height <= 0 || !rgb) return 0;
    hdr_sink s = {out, cap, 0};
    char head[192];
    const int hl = snprintf(head, sizeof head,
                            "#?RADIANCE\n# Written by stb_image_write.h\nFORMAT=32-bit_rle_rgbe\n"
                            "EXPOSURE=          1.0000000000000\n\n-Y %d +X %d\n", height, width);
    hdr_put(&s, (const uint8_t*)head, (uint64_t)hl);
    uint8_t* planes = (uint8_t*)malloc((size_t)width * 4);
    for (int y = 0; y < height; y++) {
        const float* row = rgb + (size_t)y * width * 3;
        if (width < 8 || width >= 32768) { /* :683-697: no run-length coding */
            for (int x = 0; x < width; x++) {
                uint8_t px[4];
                ora_rgbe(row + 3 * x, px);
                hdr_put(&s, px, 4);
            }
            continue;
        }
        for (int x = 0; x < width; x++) {
            uint8_t px[4];
            ora_rgbe(row + 3 * x, px);
            for (int c = 0; c < 4; c++) planes[x + (size_t)width * c] = px[c];
        }
        const uint8_t marker[4] = {2, 2, (uint8_t)((width & 0xff00) >> 8), (uint8_t)(width & 0xff)};
        hdr_put(&s, marker, 4);
        for (int c = 0; c < 4; c++) hdr_rle_component(&s, planes + (size_t)width * c, width);
    }
    free(planes);
    return s.len;
}

/* ---- OBJ loading (ms_load_obj, src/mesh/mesh.c:73-219) ----------------------------------------------------------- */

typedef struct {
    uint32_t p, n, t, index;
    int used;
} obj_slot;

static int obj_scan_uint(const char** s, uint32_t* out) { /* " %u" */
    const char* q = *s;
    while (*q == ' ' || *q == '\t' || *q == '\r' || *q == '\v' || *q == '\f') q++;
    if (!((*q >= '0' && *q <= '9') || *q == '+' || *q == '-')) return 0;
    char* after = NULL;
    unsigned long v = strtoul(q, &after, 10);
    if (after == q) return 0;
    *out = (uint32_t)v;
    *s = after;
    return 1;
}

static void obj_scan_floats(const char* s, float* out, int n) { /* "%f %f %f": stops at the first failure */
    for (int k = 0; k < n; k++) {
        char* after = NULL;
        float v = strtof(s, &after);
        if (after == s) return;
        out[k] = v;
        s = after;
    }
}

int ora_load_obj(const char* path, uint32_t attrs, float* out_vertices, uint32_t* out_vertex_count,
                 uint32_t* out_indices, uint32_t* out_index_count) {
    if (attrs == 0) attrs = 7; /* mesh.c:85-87 */
    FILE* f = fopen(path, "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    char* text = (char*)malloc((size_t)size + 1);
    if (fread(text, 1, (size_t)size, f) != (size_t)size) {
        fclose(f);
        free(text);
        return 2;
    }
    fclose(f);
    text[size] = 0;
    /* The reference reads with `while (!feof) { fgets; ... }` (mesh.c:108-109, 140-141; os_platform_linux+wasm.c:27-41):
     * after the last line of a file that ends in a newline, end-of-file has not been seen yet, the next fgets fails and
     * leaves the buffer as it was, and the loop body runs once more on the last line.  So such a file's last line counts
     * twice, in both passes (a trailing face is a duplicated triangle). */
    const int repeat_last = size > 0 && text[size - 1] == '\n';
    /* first pass: counts (mesh.c:107-120) */
    uint64_t nv = 0, nvn = 0, nvt = 0, nf = 0;
    char* last_line = text;
    for (char* s = text; s < text + size;) {
        char* nl = memchr(s, '\n', (size_t)(text + size - s));
        last_line = s;
        for (int twice = 0; twice < ((nl == text + size - 1 && repeat_last) ? 2 : 1); twice++) {
            if (strncmp(s, "v ", 2) == 0) nv++;
            else if (strncmp(s, "vn ", 3) == 0) nvn++;
            else if (strncmp(s, "vt ", 3) == 0) nvt++;
            else if (strncmp(s, "f ", 2) == 0) nf++;
        }
        s = nl ? nl + 1 : text + size;
    }
    float* pos = (float*)calloc(3 * nv + 1, 4);
    float* nrm = (float*)calloc(3 * nvn + 1, 4);
    float* uv = (float*)calloc(2 * nvt + 1, 4);
    uint32_t* corners = (uint32_t*)calloc(9 * nf + 1, 4); /* p, n, t per corner */
    uint64_t iv = 0, ivn = 0, ivt = 0, ic = 0;
    int rc = 0;
    /* second pass (mesh.c:140-212) */
    for (char* s = text; s < text + size && rc == 0;) {
        char* nl = memchr(s, '\n', (size_t)(text + size - s));
        if (nl) *nl = 0;
        for (int twice = 0; twice < ((s == last_line && repeat_last) ? 2 : 1) && rc == 0; twice++) {
            if (strncmp(s, "v ", 2) == 0) obj_scan_floats(s + 1, pos + 3 * iv++, 3);
            else if (strncmp(s, "vn ", 3) == 0) obj_scan_floats(s + 2, nrm + 3 * ivn++, 3);
            else if (strncmp(s, "vt ", 3) == 0) obj_scan_floats(s + 2, uv + 2 * ivt++, 2);
            else if (strncmp(s, "f ", 2) == 0) {
                const char* q = s + 1;
                for (int k = 0; k < 3 && rc == 0; k++) {
                    uint32_t p = 0, n = 0, t = 0;
                    int ok = obj_scan_uint(&q, &p);
                    if (ok && attrs == 5) ok = *q == '/' && (q++, obj_scan_uint(&q, &t));
                    if (ok && attrs == 3) ok = q[0] == '/' && q[1] == '/' && (q += 2, obj_scan_uint(&q, &n));
                    if (ok && attrs == 7) ok = *q == '/' && (q++, obj_scan_uint(&q, &t)) && *q == '/' && (q++, obj_scan_uint(&q, &n));
                    if (!ok || p > nv || n > nvn || t > nvt) rc = 3;
                    corners[3 * ic] = p;
                    corners[3 * ic + 1] = n;
                    corners[3 * ic + 2] = t;
                    ic++;
                }
            }
        }
        s = nl ? nl + 1 : text + size;
    }
    free(text);
    /* vertex = distinct (p, n, t), numbered by first appearance (mesh.c:19-37, 183-190) */
    uint64_t cap = 16;
    while (cap < 2 * (ic + 1)) cap <<= 1;
    obj_slot* table = (obj_slot*)calloc(cap, sizeof(obj_slot));
    uint32_t n_vertices = 0;
    const uint32_t slots = (attrs & 1u) + ((attrs >> 1) & 1u) + ((attrs >> 2) & 1u);
    for (uint64_t c = 0; c < ic && rc == 0; c++) {
        const uint32_t p = corners[3 * c], n = corners[3 * c + 1], t = corners[3 * c + 2];
        uint64_t h = (p * 0x9E3779B97F4A7C15ull) ^ (n * 0xC2B2AE3D27D4EB4Full) ^ (t * 0x165667B19E3779F9ull);
        h ^= h >> 29;
        uint64_t at = h & (cap - 1);
        while (table[at].used && !(table[at].p == p && table[at].n == n && table[at].t == t)) at = (at + 1) & (cap - 1);
        if (!table[at].used) {
            table[at].used = 1;
            table[at].p = p, table[at].n = n, table[at].t = t;
            table[at].index = n_vertices;
            if (out_vertices) { /* mesh.c:39-70: one 16-byte slot per attribute, 1-based indices, 0 = absent */
                float* v = out_vertices + (size_t)n_vertices * slots * 4;
                uint32_t slot = 0;
                memset(v, 0, (size_t)slots * 16);
                if (attrs & 1u) {
                    if (p) memcpy(v + 4 * slot, pos + 3 * (size_t)(p - 1), 12);
                    slot++;
                }
                if (attrs & 2u) {
                    if (n) memcpy(v + 4 * slot, nrm + 3 * (size_t)(n - 1), 12);
                    slot++;
                }
                if (attrs & 4u) {
                    if (t) memcpy(v + 4 * slot, uv + 2 * (size_t)(t - 1), 8);
                    slot++;
                }
            }
            n_vertices++;
        }
        if (out_indices) out_indices[c] = table[at].index;
    }
    *out_vertex_count = n_vertices;
    *out_index_count = (uint32_t)ic;
    free(table);
    free(corners);
    free(pos);
    free(nrm);
    free(uv);
    return rc;
}
