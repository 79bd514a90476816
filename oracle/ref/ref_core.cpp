// TEST INFRASTRUCTURE -- not part of the product.
//
// Thin extern "C" harness around the UNMODIFIED reference sources.  This file
// contains none of the reference's code: it #includes the reference's unity
// translation unit from where it lies (-I /root/reference/src, see
// oracle/Makefile) and exposes a handful of its static functions to ctypes so
// that tests can compare the CPU restatement (oracle/rt_oracle.c) and the HIP
// library with the real thing on arbitrary inputs.  Output: oracle/_ref/*.so.
//
// Build flags follow the reference's release recipe (build.sh:5,87):
//   g++ -x c++ -I src -O3 -DBUILD_DEBUG=0 ... -lm
// REF_DEBUG_BUILD=1 builds the BUILD_DEBUG=1 flavour (build.sh:89), whose only
// use here is the reference's own ray dump (raytracer/cpu/extra/dump.c).

#include "demos/demos_inc.h"
#define main ref_unused_demo_main
#include "demos/demos_inc.c"
#undef main
static void render(const DEMO_Settings*) {}

#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <vector>

#include "rthip.h"

static ThreadCtx g_ref_ctx;
static bool g_ref_ready = false;
static void ref_ready() {
    if (!g_ref_ready) {
        thread_equip(&g_ref_ctx);
        g_ref_ready = true;
        (void)&ref_unused_demo_main;
        (void)&render;
    }
}

// pre-order serialisation in the byte format of lbvh/extra/dump.c
static void ref_walk(const LBVH_Node* node, std::vector<uint8_t>& out) {
    uint8_t present = node ? 1 : 0;
    out.push_back(present);
    if (!node) return;
    const uint8_t* p = (const uint8_t*)&node->id;
    out.insert(out.end(), p, p + 8);
    p = (const uint8_t*)&node->aabb.min;
    out.insert(out.end(), p, p + 12);
    p = (const uint8_t*)&node->aabb.max;
    out.insert(out.end(), p, p + 12);
    ref_walk(node->left, out);
    ref_walk(node->right, out);
}
static void ref_inorder_ids(const LBVH_Node* node, uint64_t* ids, uint64_t* k) {
    if (!node) return;
    if (node->id > 0) ids[(*k)++] = node->id;
    ref_inorder_ids(node->left, ids, k);
    ref_inorder_ids(node->right, ids, k);
}

extern "C" {

int ref_is_debug_build(void) { return BUILD_DEBUG; }

uint64_t ref_morton(const float* c, const float* mn, const float* ext) {
    ref_ready();
    return lbvh_c_to_morton_code(make_3f32(c[0], c[1], c[2]), make_3f32(mn[0], mn[1], mn[2]),
                                 make_3f32(ext[0], ext[1], ext[2]));
}

// lbvh_make on n boxes.  out_codes: code per input box, computed with the
// reference's own helpers from the bounds its release build uses.  out_sorted_ids:
// leaf ids in tree (= sorted) order.  The dump goes to out_dump (capacity cap).
int ref_lbvh_build(const float* aabb6, uint64_t n, uint64_t* out_codes, uint64_t* out_sorted_ids,
                   uint8_t* out_dump, uint64_t cap, uint64_t* out_len, double* out_seconds) {
    ref_ready();
    rng3_f32* boxes = (rng3_f32*)malloc(sizeof(rng3_f32) * n);
    memcpy(boxes, aabb6, sizeof(rng3_f32) * n);
    Arena* arena = arena_alloc();
    double t0 = os_now_seconds();
    LBVH_Tree tree = lbvh_make(arena, boxes, n);
    double t1 = os_now_seconds();
    if (out_seconds) *out_seconds = t1 - t0;
    if (out_codes) {
        vec3_f32 mn = make_scale_3f32(MAX_F32), mx = make_scale_3f32(-MAX_F32);
        for (uint64_t i = 0; i < n; i++) {
#if !BUILD_DEBUG
            vec3_f32 c = mul_3f32(add_3f32(boxes[i].min, boxes[i].max), 0.5f);
            mn = min_3f32(mn, c);
            mx = max_3f32(mx, c);
#else
            mn = min_3f32(mn, boxes[i].min);
            mx = max_3f32(mx, boxes[i].max);
#endif
        }
        vec3_f32 ext = sub_3f32(mx, mn);
        for (uint64_t i = 0; i < n; i++) {
            vec3_f32 c = mul_3f32(add_3f32(boxes[i].min, boxes[i].max), 0.5f);
            out_codes[i] = lbvh_c_to_morton_code(c, mn, ext);
        }
    }
    if (out_sorted_ids) {
        uint64_t k = 0;
        ref_inorder_ids(tree.root, out_sorted_ids, &k);
    }
    int rc = 0;
    if (out_dump || out_len) {
        std::vector<uint8_t> bytes;
        bytes.reserve(n * 70);
        ref_walk(tree.root, bytes);
        if (out_len) *out_len = bytes.size();
        if (out_dump) {
            if (bytes.size() > cap) rc = 1;
            else memcpy(out_dump, bytes.data(), bytes.size());
        }
    }
    arena_release(arena);
    free(boxes);
    return rc;
}

// the reference's own dump routine, for pinning the byte format of ref_walk
int ref_lbvh_dump_file(const float* aabb6, uint64_t n, const char* path) {
    ref_ready();
    rng3_f32* boxes = (rng3_f32*)malloc(sizeof(rng3_f32) * n);
    memcpy(boxes, aabb6, sizeof(rng3_f32) * n);
    Arena* arena = arena_alloc();
    LBVH_Tree tree = lbvh_make(arena, boxes, n);
    lbvh_dump_tree(&tree, path);
    arena_release(arena);
    free(boxes);
    return 0;
}

int ref_slab(const float* aabb6, const float* ray6, const float* interval2) {
    rng3_f32 box, ray;
    memcpy(&box, aabb6, 24);
    memcpy(&ray, ray6, 24);
    rng_f32 iv = make_rng_f32(interval2[0], interval2[1]);
    return lbvh_aabb_query_ray(box, &ray, &iv) ? 1 : 0;
}

int ref_intersect_tri(const float* ray6, const float* a, const float* b, const float* c, float* interval2,
                      float* uv2) {
    rng3_f32 ray;
    memcpy(&ray, ray6, 24);
    rng_f32 iv = make_rng_f32(interval2[0], interval2[1]);
    vec2_f32 uv = {};
    bool hit = geo_intersect_tri(&ray, make_3f32(a[0], a[1], a[2]), make_3f32(b[0], b[1], b[2]),
                                 make_3f32(c[0], c[1], c[2]), &iv, &uv);
    interval2[0] = iv.min;
    interval2[1] = iv.max;
    if (hit) {
        uv2[0] = uv.U;
        uv2[1] = uv.V;
    }
    return hit ? 1 : 0;
}

int ref_intersect_sphere(const float* ray6, const float* center, float radius, float* interval2) {
    rng3_f32 ray;
    memcpy(&ray, ray6, 24);
    rng_f32 iv = make_rng_f32(interval2[0], interval2[1]);
    bool hit = geo_intersect_sphere(&ray, make_3f32(center[0], center[1], center[2]), radius, &iv);
    interval2[0] = iv.min;
    interval2[1] = iv.max;
    return hit ? 1 : 0;
}

void ref_transform_aabb(const float* aabb6, const float* t3, const float* q4, const float* s3, float* out6) {
    rng3_f32 box;
    memcpy(&box, aabb6, 24);
    vec4_f32 q = make_4f32(q4[0], q4[1], q4[2], q4[3]);
    rng3_f32 r = rt_cpu_transform_aabb(box, make_3f32(t3[0], t3[1], t3[2]), q, make_3f32(s3[0], s3[1], s3[2]));
    memcpy(out6, &r, 24);
}

// get_rt_cast_settings, demos_inc.c:104-130, run by the reference itself
void ref_cast_settings(int width, int height, uint32_t samples, const float* eye, const float* subject,
                       float defocus_angle, int orthographic, float vfov, rthip_cast_settings* out) {
    DEMO_Settings s = {};
    s.width = width;
    s.height = height;
    s.samples = (u8)samples;
    DEMO_ExtraCastSettings extra = {};
    extra.defocus_angle = defocus_angle;
    extra.orthographic = orthographic != 0;
    extra.vfov = vfov;
    RT_CastSettings c = get_rt_cast_settings(&s, make_3f32(eye[0], eye[1], eye[2]),
                                             make_3f32(subject[0], subject[1], subject[2]), extra);
    memcpy(out->eye, &c.eye, 12);
    memcpy(out->up, &c.up, 12);
    memcpy(out->forward, &c.forward, 12);
    memcpy(out->right, &c.right, 12);
    memcpy(out->viewport, &c.viewport, 12);
    out->samples = c.samples;
    out->ior = c.ior;
    out->defocus = c.defocus;
    memcpy(out->defocus_disk, &c.defocus_disk, 8);
    out->orthographic = c.orthographic;
}

// make_a_to_b_quat, common_math.c:92-100 (the cornell demo derives its quad rotations with it)
void ref_a_to_b_quat(const float* a, const float* b, float* out4) {
    vec4_f32 q = make_a_to_b_quat(make_3f32(a[0], a[1], a[2]), make_3f32(b[0], b[1], b[2]));
    memcpy(out4, &q, 16);
}

// ---- a world built through the reference's own scene API -----------------

struct RefScene {
    RT_World* world;
    RT_Handle tracer;
    std::vector<RT_Handle> meshes;     // by blas_id
    std::vector<RT_Handle> materials;  // by material index
    std::vector<std::vector<uint8_t>> vertex_copies;
    std::vector<std::vector<uint32_t>> index_copies;
};

struct ref_mesh_desc {
    const void* verts;
    uint32_t vertex_count;
    const uint32_t* indices;
    uint32_t index_count;
    uint32_t attrs;  // GEO_VertexAttributes bits
};

// meshes / instances / materials arrive in list-iteration order (element 0 = added
// last), so they are added to the world back to front.
void* ref_scene_create(uint32_t max_bounces, int sky, const ref_mesh_desc* meshes, uint32_t n_meshes,
                       const rthip_instance* insts, uint32_t n_insts, const rthip_material* mats,
                       uint32_t n_mats) {
    ref_ready();
    RefScene* s = new RefScene();
    RT_WorldSettings ws = {};
    s->world = rt_make_world(ws);
    RT_TracerSettings ts = {};
    ts.max_bounces = (u8)max_bounces;
    ts.sky = sky != 0;
    s->tracer = rt_make_tracer(ts);

    s->meshes.resize(n_meshes);
    s->vertex_copies.resize(n_meshes);
    s->index_copies.resize(n_meshes);
    for (uint32_t k = n_meshes; k-- > 0;) {
        const ref_mesh_desc* d = &meshes[k];
        GEO_VertexAttributes attrs = (GEO_VertexAttributes)d->attrs;
        size_t vbytes = (size_t)geo_vertex_size(attrs) * d->vertex_count;
        s->vertex_copies[k].assign((const uint8_t*)d->verts, (const uint8_t*)d->verts + vbytes);
        if (d->indices && d->index_count)
            s->index_copies[k].assign(d->indices, d->indices + d->index_count);
        RT_Handle h = rt_world_add_mesh(s->world);
        RT_Mesh* m = rt_world_resolve_mesh(s->world, h);
        m->vertices = s->vertex_copies[k].data();
        m->vertices_count = d->vertex_count;
        m->indices = s->index_copies[k].empty() ? NULL : s->index_copies[k].data();
        m->indices_count = (u32)s->index_copies[k].size();
        m->primitive = GEO_Primitive_TRI_LIST;
        m->attrs = attrs;
        s->meshes[k] = h;
    }
    rt_tracer_build_blas(s->tracer, s->world);

    s->materials.resize(n_mats);
    for (uint32_t k = n_mats; k-- > 0;) {
        RT_Handle h = rt_world_add_material(s->world);
        RT_Material* m = rt_world_resolve_material(s->world, h);
        m->type = (RT_MaterialType)mats[k].type;
        m->albedo = make_3f32(mats[k].albedo[0], mats[k].albedo[1], mats[k].albedo[2]);
        m->emissive = make_3f32(mats[k].emissive[0], mats[k].emissive[1], mats[k].emissive[2]);
        m->roughness = mats[k].roughness;
        m->ior = mats[k].ior;
        m->billboard = mats[k].billboard != 0;
        s->materials[k] = h;
    }
    for (uint32_t k = n_insts; k-- > 0;) {
        const rthip_instance* d = &insts[k];
        RT_Handle h = rt_world_add_instance(s->world);
        RT_Instance* it = rt_world_resolve_instance(s->world, h);
        it->type = (RT_InstanceType)d->type;
        if (d->material >= 0) it->material = s->materials[d->material];
        if (d->type == RT_InstanceType_Sphere) {
            it->sphere.center = make_3f32(d->sphere_center[0], d->sphere_center[1], d->sphere_center[2]);
            it->sphere.radius = d->sphere_radius;
        } else {
            it->mesh.handle = s->meshes[d->blas_id];
            it->mesh.translation = make_3f32(d->translation[0], d->translation[1], d->translation[2]);
            it->mesh.rotation = make_4f32(d->rotation[0], d->rotation[1], d->rotation[2], d->rotation[3]);
            it->mesh.scale = make_3f32(d->scale[0], d->scale[1], d->scale[2]);
        }
    }
    // rt_cpu_build_tlas always writes "out.bvh" into the working directory
    // (raytracer_cpu.c:171-173); keep that file out of the caller's tree.
    char cwd[4096];
    bool moved = getcwd(cwd, sizeof cwd) && chdir("/tmp") == 0;
    rt_tracer_build_tlas(s->tracer, s->world);
    if (moved && chdir(cwd) != 0) {}
    return s;
}

void ref_scene_destroy(void* scene) {
    RefScene* s = (RefScene*)scene;
    rt_tracer_cleanup(s->tracer);
    rt_world_cleanup(s->world);
    delete s;
}

static void cast_from_pod(const rthip_cast_settings* c, RT_CastSettings* o) {
    memset(o, 0, sizeof *o);
    memcpy(&o->eye, c->eye, 12);
    memcpy(&o->up, c->up, 12);
    memcpy(&o->forward, c->forward, 12);
    memcpy(&o->right, c->right, 12);
    memcpy(&o->viewport, c->viewport, 12);
    o->samples = (u8)c->samples;
    o->ior = c->ior;
    o->defocus = c->defocus != 0;
    memcpy(&o->defocus_disk, c->defocus_disk, 8);
    o->orthographic = c->orthographic != 0;
}

// rt_tracer_cast after srand(seed), as src/demos/demos_inc.c:97 does.  In the
// BUILD_DEBUG=1 flavour the reference appends every ray to ./out.rays, so the
// caller chooses the working directory through `workdir` (may be NULL).
double ref_scene_cast(void* scene, const rthip_cast_settings* cast, int seed, float* out_rgb, int width, int height,
                      const char* workdir) {
    RefScene* s = (RefScene*)scene;
    RT_CastSettings cs;
    cast_from_pod(cast, &cs);
    char cwd[4096];
    bool moved = workdir && getcwd(cwd, sizeof cwd) && chdir(workdir) == 0;
    rand_seed((u64)seed);
    double t0 = os_now_seconds();
    rt_tracer_cast(s->tracer, cs, (vec3_f32*)out_rgb, width, height);
    double t1 = os_now_seconds();
    if (moved && chdir(cwd) != 0) {}
    return t1 - t0;
}

// rt_cpu_intersect for caller-supplied rays, plus the TLAS-level record
// (instance id, tri_idx, uv) that rt_cpu_intersect consumes internally.
// lbvh_query_ray over the reference-built TLAS with the interval [EPSILON_F32, tmax]: hit or not
void ref_scene_occluded(void* scene, const float* o3, const float* d3, uint64_t n, float tmax, uint8_t* out) {
    RefScene* s = (RefScene*)scene;
    RT_CPU_Tracer* tr = rt_cpu_handle_to_tracer(s->tracer);
    for (uint64_t i = 0; i < n; i++) {
        rng3_f32 ray;
        ray.origin = make_3f32(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]);
        ray.direction = make_3f32(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]);
        rng_f32 iv = geo_make_pos_interval();
        iv.max = tmax;
        RT_CPU_TLASData data = {};
        data.tlas = &tr->tlas;
        out[i] = lbvh_query_ray(&tr->tlas.lbvh, &ray, &iv, &rt_cpu_tlas_hit, (void*)&data) != 0 ? 1 : 0;
    }
}

double ref_scene_trace(void* scene, const float* o3, const float* d3, uint64_t n, uint32_t* out_inst,
                       uint32_t* out_tri, float* out_t, float* out_uv, float* out_normal) {
    RefScene* s = (RefScene*)scene;
    RT_CPU_Tracer* tr = rt_cpu_handle_to_tracer(s->tracer);
    double t0 = os_now_seconds();
    for (uint64_t i = 0; i < n; i++) {
        rng3_f32 ray;
        ray.origin = make_3f32(o3[3 * i], o3[3 * i + 1], o3[3 * i + 2]);
        ray.direction = make_3f32(d3[3 * i], d3[3 * i + 1], d3[3 * i + 2]);
        rng_f32 iv = geo_make_pos_interval();
        RT_CPU_TLASData data = {};
        data.tlas = &tr->tlas;
        bool hit = lbvh_query_ray(&tr->tlas.lbvh, &ray, &iv, &rt_cpu_tlas_hit, (void*)&data) != 0;
        uint32_t inst = 0, tri = 0;
        float t = -1.0f, u = 0.f, v = 0.f;
        if (hit) {
            inst = (uint32_t)(data.hit_record.tlas_node - tr->tlas.nodes) + 1;
            t = iv.max;
            if (data.hit_record.tlas_node->instance->type == RT_InstanceType_Mesh) {
                tri = data.hit_record.tri_idx;
                u = data.hit_record.uv.U;
                v = data.hit_record.uv.V;
            }
        }
        if (out_inst) out_inst[i] = inst;
        if (out_tri) out_tri[i] = tri;
        if (out_t) out_t[i] = t;
        if (out_uv) {
            out_uv[2 * i] = u;
            out_uv[2 * i + 1] = v;
        }
        if (out_normal) {
            RT_CPU_HitRecord rec = {};
            bool hit2 = rt_cpu_intersect(tr, &ray, geo_make_pos_interval(), &rec);
            out_normal[3 * i + 0] = hit2 ? rec.n.x : 0.f;
            out_normal[3 * i + 1] = hit2 ? rec.n.y : 0.f;
            out_normal[3 * i + 2] = hit2 ? rec.n.z : 0.f;
        }
    }
    return os_now_seconds() - t0;
}

// tree dumps: which = -1 TLAS, else blas id
int ref_scene_dump_tree(void* scene, int which, uint8_t* out, uint64_t cap, uint64_t* out_len) {
    RefScene* s = (RefScene*)scene;
    RT_CPU_Tracer* tr = rt_cpu_handle_to_tracer(s->tracer);
    const LBVH_Node* root = NULL;
    if (which < 0) root = tr->tlas.lbvh.root;
    else if ((uint64_t)which < tr->blas.node_count) root = tr->blas.nodes[which].lbvh.root;
    else return 2;
    std::vector<uint8_t> bytes;
    ref_walk(root, bytes);
    if (out_len) *out_len = bytes.size();
    if (out) {
        if (bytes.size() > cap) return 1;
        memcpy(out, bytes.data(), bytes.size());
    }
    return 0;
}

// ms_load_obj, src/mesh/mesh.c:73-219, with attrs PN or P; copies the result out.
int ref_load_obj(const char* path, uint32_t attrs, float* out_verts, uint64_t vert_cap_floats,
                 uint32_t* out_vertex_count, uint32_t* out_indices, uint64_t index_cap,
                 uint32_t* out_index_count) {
    ref_ready();
    Arena* arena = arena_alloc();
    MS_LoadSettings ls = {};
    ls.primitive = GEO_Primitive_TRI_LIST;
    ls.attrs = (GEO_VertexAttributes)attrs;
    NTString8 p = make_ntstr8((char*)path, strlen(path));
    MS_LoadResult r = ms_load_obj(arena, p, ls);
    int rc = 0;
    if (r.error.length != 0) rc = 2;
    else {
        uint64_t floats = (uint64_t)geo_vertex_size(r.v.attrs) / 4 * r.v.vertices_count;
        *out_vertex_count = r.v.vertices_count;
        *out_index_count = r.v.indices_count;
        if (out_verts) {
            if (floats > vert_cap_floats) rc = 1;
            else memcpy(out_verts, r.v.vertices, floats * 4);
        }
        if (out_indices) {
            if (r.v.indices_count > index_cap) rc = 1;
            else memcpy(out_indices, r.v.indices, (size_t)r.v.indices_count * 4);
        }
    }
    arena_release(arena);
    return rc;
}

}  // extern "C"
