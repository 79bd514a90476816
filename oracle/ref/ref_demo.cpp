// TEST INFRASTRUCTURE -- not part of the product.
//
// Runs one of the reference's demo programs (src/demos/<name>/main.c, chosen with
// -DREF_DEMO_MAIN=...) unmodified inside a shared library and taps the state it
// holds at the moment it would write its .hdr file: the reference-built tracer,
// the camera it derived and the radiance buffer.  Contains no reference code; the
// sources are #included from /root/reference/src (see oracle/Makefile).
// Output: oracle/_ref/libref_<demo>.so.

#include "demos/demos_inc.h"

#include <stdint.h>
#include <string.h>
#include <unistd.h>
#include <vector>

#include "rthip.h"

struct RefTapRequest {
    float* out_rgb;
    rthip_cast_settings* out_cast;
    const float* rays_o;
    const float* rays_d;
    uint64_t n_rays;
    uint32_t* out_inst;
    uint32_t* out_tri;
    float* out_t;
    uint8_t* tlas_dump;
    uint64_t tlas_cap;
    uint64_t* tlas_len;
    uint32_t* out_n_instances;
};
static RefTapRequest g_req;

static int ref_tap(RT_Handle tracer, RT_CastSettings cs, const char* path, int w, int h, const float* data);
#define stbi_write_hdr(path, w, h, comp, data) ref_tap(tracer, csettings, path, w, h, data)
#define main ref_demo_main
#include REF_DEMO_MAIN
#undef main
#undef stbi_write_hdr

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

static int ref_tap(RT_Handle tracer, RT_CastSettings cs, const char* path, int w, int h, const float* data) {
    (void)path;
    RT_CPU_Tracer* tr = rt_cpu_handle_to_tracer(tracer);
    if (g_req.out_rgb) memcpy(g_req.out_rgb, data, sizeof(float) * 3 * (size_t)w * h);
    if (g_req.out_cast) {
        rthip_cast_settings* o = g_req.out_cast;
        memcpy(o->eye, &cs.eye, 12);
        memcpy(o->up, &cs.up, 12);
        memcpy(o->forward, &cs.forward, 12);
        memcpy(o->right, &cs.right, 12);
        memcpy(o->viewport, &cs.viewport, 12);
        o->samples = cs.samples;
        o->ior = cs.ior;
        o->defocus = cs.defocus;
        memcpy(o->defocus_disk, &cs.defocus_disk, 8);
        o->orthographic = cs.orthographic;
    }
    if (g_req.out_n_instances) *g_req.out_n_instances = (uint32_t)tr->tlas.node_count;
    if (g_req.tlas_len || g_req.tlas_dump) {
        std::vector<uint8_t> bytes;
        ref_walk(tr->tlas.lbvh.root, bytes);
        if (g_req.tlas_len) *g_req.tlas_len = bytes.size();
        if (g_req.tlas_dump && bytes.size() <= g_req.tlas_cap) memcpy(g_req.tlas_dump, bytes.data(), bytes.size());
    }
    for (uint64_t i = 0; i < g_req.n_rays; i++) {
        rng3_f32 ray;
        ray.origin = make_3f32(g_req.rays_o[3 * i], g_req.rays_o[3 * i + 1], g_req.rays_o[3 * i + 2]);
        ray.direction = make_3f32(g_req.rays_d[3 * i], g_req.rays_d[3 * i + 1], g_req.rays_d[3 * i + 2]);
        rng_f32 iv = geo_make_pos_interval();
        RT_CPU_TLASData td = {};
        td.tlas = &tr->tlas;
        bool hit = lbvh_query_ray(&tr->tlas.lbvh, &ray, &iv, &rt_cpu_tlas_hit, (void*)&td) != 0;
        uint32_t inst = 0, tri = 0;
        float t = -1.0f;
        if (hit) {
            inst = (uint32_t)(td.hit_record.tlas_node - tr->tlas.nodes) + 1;
            t = iv.max;
            if (td.hit_record.tlas_node->instance->type == RT_InstanceType_Mesh) tri = td.hit_record.tri_idx;
        }
        if (g_req.out_inst) g_req.out_inst[i] = inst;
        if (g_req.out_tri) g_req.out_tri[i] = tri;
        if (g_req.out_t) g_req.out_t[i] = t;
    }
    return 1;
}

extern "C" {

int ref_demo_is_debug_build(void) { return BUILD_DEBUG; }

// Runs the demo exactly as its command line would:
//   <demo> OUT --width=W --height=H --samples=S --bounces=B --seed=SEED
// from inside `workdir` (where the reference writes out.bvh / out.rays and looks
// for data/bunny.obj).  Every output pointer may be NULL.
int ref_demo_run(int width, int height, int samples, int bounces, int seed, const char* workdir,
                 float* out_rgb, rthip_cast_settings* out_cast, const float* rays_o, const float* rays_d,
                 uint64_t n_rays, uint32_t* out_inst, uint32_t* out_tri, float* out_t, uint8_t* tlas_dump,
                 uint64_t tlas_cap, uint64_t* tlas_len, uint32_t* out_n_instances) {
    g_req.out_rgb = out_rgb;
    g_req.out_cast = out_cast;
    g_req.rays_o = rays_o;
    g_req.rays_d = rays_d;
    g_req.n_rays = n_rays;
    g_req.out_inst = out_inst;
    g_req.out_tri = out_tri;
    g_req.out_t = out_t;
    g_req.tlas_dump = tlas_dump;
    g_req.tlas_cap = tlas_cap;
    g_req.tlas_len = tlas_len;
    g_req.out_n_instances = out_n_instances;

    char a_w[32], a_h[32], a_s[32], a_b[32], a_seed[32];
    snprintf(a_w, sizeof a_w, "--width=%d", width);
    snprintf(a_h, sizeof a_h, "--height=%d", height);
    snprintf(a_s, sizeof a_s, "--samples=%d", samples);
    snprintf(a_b, sizeof a_b, "--bounces=%d", bounces);
    snprintf(a_seed, sizeof a_seed, "--seed=%d", seed);
    char prog[] = "demo", out[] = "unused.hdr";
    char* argv[] = {prog, out, a_w, a_h, a_s, a_b, a_seed, NULL};

    char cwd[4096];
    bool moved = workdir && getcwd(cwd, sizeof cwd) && chdir(workdir) == 0;
    int rc = ref_demo_main(7, argv);
    if (moved && chdir(cwd) != 0) {}
    return rc;
}

}  // extern "C"
