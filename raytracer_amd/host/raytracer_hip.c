// HIP backend of the tracer hooks; see raytracer_hip.h.
//
// What stays on the host: walking the world's linked lists and copying their records into the
// arrays rthip_* takes.  Everything else (BLAS/TLAS construction, traversal, shading) runs on
// the GPU.  The hooks keep the reference's void/handle signatures; a failing rthip_* call is
// fatal (AssertAlways), because the reference's callers have no error path either.

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

internal RT_HIP_Tracer* rt_hip_handle_to_tracer(RT_Handle handle) {
    return (RT_HIP_Tracer*)handle.v64[0];
}
internal RT_Handle rt_hip_tracer_to_handle(RT_HIP_Tracer* tracer) {
    RT_Handle handle = zero_struct;
    handle.v64[0] = (u64)tracer;
    return handle;
}

static void rt_hip_check(RT_HIP_Tracer* tracer, int rc, const char* what) {
    if (rc != RTHIP_OK) {
        const char* msg = tracer->ctx ? rthip_last_error(tracer->ctx) : "no context";
        if (tracer->group && rthip_group_last_error(tracer->group)[0]) msg = rthip_group_last_error(tracer->group);
        fprintf(stderr, "rthip: %s failed (%d): %s\n", what, rc, msg);
    }
    AssertAlways(rc == RTHIP_OK);
}

// The reference's tree dump (lbvh_dump_tree, src/lbvh/extra/dump.c:1-23, read by extra/blender.py): the
// pointer tree in depth-first order, per node a 1-byte sentinel 1, the u64 id, min, max, and a sentinel 0
// for every NULL child.  The GPU tree is already stored in that order (left child = next record), and a
// node is a leaf exactly when its id is not 0, so the file is one pass over the node array.
static void rt_hip_dump_tree(RT_HIP_Tracer* tracer, int which, const char* path) {
    u64 count = 0;
    rt_hip_check(tracer, rthip_get_tree(tracer->ctx, which, NULL, &count), "rthip_get_tree");
    FILE* f = fopen(path, "w");
    if (!f) return;
    {DeferResource(Temp scratch = scratch_begin(NULL, 0), scratch_end(scratch)) {
        rthip_node* nodes = push_array(scratch.arena, rthip_node, count + 1);
        if (count) rt_hip_check(tracer, rthip_get_tree(tracer->ctx, which, nodes, &count), "rthip_get_tree");
        if (count == 0) {
            u8 none = 0;
            fwrite(&none, 1, 1, f);
        }
        for (u64 i = 0; i < count; i++) {
            u8 one = 1, none[2] = {0, 0};
            u64 id = nodes[i].id;
            fwrite(&one, 1, 1, f);
            fwrite(&id, sizeof(id), 1, f);
            fwrite(nodes[i].aabb_min, sizeof(f32), 3, f);
            fwrite(nodes[i].aabb_max, sizeof(f32), 3, f);
            if (id != 0) fwrite(none, 1, 2, f);
        }
    }}
    fclose(f);
}

// The reference's ray dump (src/raytracer/cpu/extra/dump.c:1-32): per traced ray origin, direction, hit position,
// normal, t and the material handle (60 bytes), a miss as zeros with t = -1; rays in the order of the CPU loop.
static void rt_hip_dump_rays(RT_HIP_Tracer* tracer, const rthip_cast_settings* c, u64 seed, int width, int height, const char* path) {
    FILE* f = fopen(path, "wb");
    if (!f) return;
    u64 count = 0;
    rt_hip_check(tracer, rthip_cast_ray_records(tracer->ctx, c, seed, width, height, NULL, 0, &count), "rthip_cast_ray_records");
    rthip_ray_record* rec = (rthip_ray_record*)malloc(sizeof(rthip_ray_record)*(count + 1));
    if (count) rt_hip_check(tracer, rthip_cast_ray_records(tracer->ctx, c, seed, width, height, rec, count, &count), "rthip_cast_ray_records");
    for (u64 i = 0; i < count; i++) {
        u64 material = 0;
        if (rec[i].material >= 0 && (u32)rec[i].material < tracer->material_count) material = tracer->material_handles[rec[i].material];
        fwrite(rec[i].origin, sizeof(f32), 3, f);
        fwrite(rec[i].direction, sizeof(f32), 3, f);
        fwrite(rec[i].p, sizeof(f32), 3, f);
        fwrite(rec[i].n, sizeof(f32), 3, f);
        fwrite(&rec[i].t, sizeof(f32), 1, f);
        fwrite(&material, sizeof(u64), 1, f);
    }
    free(rec);
    fclose(f);
}

// rt_make_tracer: one GPU context per tracer.  The device is GPU 0 unless RT_HIP_DEVICE says otherwise.
//
// Random numbers: the CPU backend consumes the process-wide rand() stream that the demo seeded with
// srand(--seed) (src/demos/demos_inc.c:97).  A GPU cannot share that stream, so this backend takes ONE
// value from it -- here, before the GPU runtime is initialised, because that initialisation draws from
// rand() itself -- and uses it as the seed of the GPU's counter-based generator: the image depends on
// --seed and on nothing else.
rt_hook RT_Handle rt_make_tracer(RT_TracerSettings settings) {
    Assert(settings.max_bounces < RT_MAX_MAX_BOUNCES);
    u64 seed = (u64)rand();

    Arena* arena = arena_alloc();
    RT_HIP_Tracer* tracer = push_array(arena, RT_HIP_Tracer, 1);
    tracer->arena = arena;
    tracer->max_bounces = settings.max_bounces;
    tracer->sky = settings.sky;
    tracer->seed = seed;

    // RT_HIP_DEVICES=0,1,2,...: the whole node behind the same five hooks (one context per listed GPU, the scene
    // replicated, every frame split by 64x64 tiles, bit-identical to one GPU).  Otherwise one GPU: RT_HIP_DEVICE or 0.
    const char* devs = getenv("RT_HIP_DEVICES");
    int rc = RTHIP_OK;
    if (devs && devs[0]) {
        int ids[64];
        int n = 0;
        for (const char* p = devs; *p && n < 64;) {
            ids[n++] = atoi(p);
            while (*p && *p != ',') p++;
            if (*p == ',') p++;
        }
        rc = rthip_group_create(&tracer->group, n, ids);
        if (rc == RTHIP_OK) tracer->ctx = rthip_group_ctx(tracer->group, 0);
    } else {
        const char* dev = getenv("RT_HIP_DEVICE");
        rc = rthip_create(&tracer->ctx, dev ? atoi(dev) : 0);
    }
    if (rc != RTHIP_OK) fprintf(stderr, "rthip: no usable MI355X (gfx950) device; this backend has no CPU path\n");
    AssertAlways(rc == RTHIP_OK);
    if (tracer->group) rt_hip_check(tracer, rthip_group_set_tracer(tracer->group, settings.max_bounces, settings.sky ? 1 : 0), "rthip_group_set_tracer");
    else rt_hip_check(tracer, rthip_set_tracer(tracer->ctx, settings.max_bounces, settings.sky ? 1 : 0), "rthip_set_tracer");
    // RT_HIP_TRAVERSAL=fast: the opt-in ordered walk (rthip_set_traversal: 4-wide records, nearest box first).  The frame
    // equals the default walk's except where the reference's own answer hangs on its visiting order (hits at exactly
    // equal distance, rays along a seam between two triangles); anything else, or unset: the reference's fixed walk.
    {
        const char* trav = getenv("RT_HIP_TRAVERSAL");
        if (trav && strcmp(trav, "fast") == 0) {
            if (tracer->group) rt_hip_check(tracer, rthip_group_set_traversal(tracer->group, RTHIP_TRAV_FAST), "rthip_group_set_traversal");
            else rt_hip_check(tracer, rthip_set_traversal(tracer->ctx, RTHIP_TRAV_FAST), "rthip_set_traversal");
        } else if (trav && trav[0] && strcmp(trav, "exact") != 0) {
            fprintf(stderr, "rthip: RT_HIP_TRAVERSAL=%s is neither \"exact\" nor \"fast\"; using the exact walk\n", trav);
        }
    }
    // lbvh_make normalises centroids by the boxes' bounds in a BUILD_DEBUG build and by the centroids' own in a
    // release build (lbvh.c:97-104): the trees differ, so the backend follows the flavour it is compiled into.
    #if BUILD_DEBUG
    if (tracer->group) rt_hip_check(tracer, rthip_group_set_morton_bounds(tracer->group, RTHIP_BOUNDS_BOXES), "rthip_group_set_morton_bounds");
    else rt_hip_check(tracer, rthip_set_morton_bounds(tracer->ctx, RTHIP_BOUNDS_BOXES), "rthip_set_morton_bounds");
    #endif
    return rt_hip_tracer_to_handle(tracer);
}

// rt_tracer_build_blas: meshes in list order; the position in that order becomes RT_Mesh.blas_id,
// exactly as the CPU backend assigns it.  Vertex and index memory is copied to the GPU during the
// call and not referenced afterwards.
rt_hook void rt_tracer_build_blas(RT_Handle handle, RT_World* world) {
    RT_HIP_Tracer* tracer = rt_hip_handle_to_tracer(handle);
    RT_MeshList* meshes = &world->meshes;

    u32 count = 0;
    for EachList(node, RT_MeshNode, meshes->first) count++;
    if (tracer->group) rt_hip_check(tracer, rthip_group_blas_begin(tracer->group, count), "rthip_group_blas_begin");
    else rt_hip_check(tracer, rthip_blas_begin(tracer->ctx, count), "rthip_blas_begin");

    u32 idx = 0;
    for EachList(node, RT_MeshNode, meshes->first) {
        RT_Mesh* mesh = &node->v;
        Assert(mesh->primitive == GEO_Primitive_TRI_LIST);
        u32 stride = (u32)geo_vertex_stride(mesh->attrs, GEO_VertexAttributes_P);
        u32 pos_off = (u32)geo_vertex_offset(mesh->attrs, GEO_VertexAttributes_P);
        s32 nrm_off = (mesh->attrs & GEO_VertexAttributes_N) ? (s32)geo_vertex_offset(mesh->attrs, GEO_VertexAttributes_N) : -1;
        if (tracer->group)
            rt_hip_check(tracer,
                         rthip_group_blas_add_mesh(tracer->group, idx, mesh->vertices, mesh->vertices_count, stride, pos_off, nrm_off,
                                                   mesh->indices, mesh->indices_count),
                         "rthip_group_blas_add_mesh");
        else
            rt_hip_check(tracer,
                         rthip_blas_add_mesh(tracer->ctx, idx, mesh->vertices, mesh->vertices_count, stride, pos_off, nrm_off,
                                             mesh->indices, mesh->indices_count),
                         "rthip_blas_add_mesh");
        mesh->blas_id = idx;
        idx++;
    }
    if (tracer->group) rt_hip_check(tracer, rthip_group_blas_commit(tracer->group), "rthip_group_blas_commit");
    else rt_hip_check(tracer, rthip_blas_commit(tracer->ctx), "rthip_blas_commit");

#if BUILD_DEBUG
    // "<mesh name>.bvh" per named mesh, as the CPU backend writes them (raytracer_cpu.c:92-97)
    {DeferResource(Temp scratch = scratch_begin(NULL, 0), scratch_end(scratch)) {
        for EachList(node, RT_MeshNode, meshes->first) {
            RT_Mesh* mesh = &node->v;
            if (mesh->name.length > 0) {
                NTString8 path = ntstr8_concatenate(scratch.arena, mesh->name, ntstr8_lit(".bvh"));
                rt_hip_dump_tree(tracer, (int)mesh->blas_id, path.cstr);
            }
        }
    }}
#endif
}

// rt_tracer_build_tlas: instances and materials in list order.  Instances refer to materials by the
// address of the material's list node; those addresses are turned into array indices here.
rt_hook void rt_tracer_build_tlas(RT_Handle handle, RT_World* world) {
    RT_HIP_Tracer* tracer = rt_hip_handle_to_tracer(handle);

    {DeferResource(Temp scratch = scratch_begin(NULL, 0), scratch_end(scratch)) {
        u32 mat_count = 0, inst_count = 0;
        for EachList(node, RT_MaterialNode, world->materials.first) mat_count++;
        for EachList(node, RT_InstanceNode, world->instances.first) inst_count++;

        rthip_material* mats = push_array(scratch.arena, rthip_material, mat_count + 1);
        // open-addressing table: material node address -> index
        u32 slots = 16;
        while (slots < 2*(mat_count + 1)) slots *= 2;
        u64* keys = push_array(scratch.arena, u64, slots);
        u32* vals = push_array(scratch.arena, u32, slots);

        free(tracer->material_handles);
        tracer->material_handles = (u64*)malloc(sizeof(u64)*(mat_count + 1));
        tracer->material_count = mat_count;

        u32 mi = 0;
        for EachList(node, RT_MaterialNode, world->materials.first) {
            const RT_Material* m = &node->v;
            tracer->material_handles[mi] = (u64)node;
            rthip_material* o = &mats[mi];
            o->type = (u32)m->type;
            o->albedo[0] = m->albedo.x;     o->albedo[1] = m->albedo.y;     o->albedo[2] = m->albedo.z;
            o->emissive[0] = m->emissive.x; o->emissive[1] = m->emissive.y; o->emissive[2] = m->emissive.z;
            o->roughness = m->roughness;
            o->ior = m->ior;
            o->billboard = m->billboard ? 1u : 0u;

            u64 key = (u64)node;
            u32 slot = (u32)((key >> 4) * 0x9E3779B97F4A7C15ull >> 32) & (slots - 1);
            while (keys[slot] != 0) slot = (slot + 1) & (slots - 1);
            keys[slot] = key;
            vals[slot] = mi;
            mi++;
        }

        rthip_instance* insts = push_array(scratch.arena, rthip_instance, inst_count + 1);
        u32 ii = 0;
        for EachList(node, RT_InstanceNode, world->instances.first) {
            const RT_Instance* in = &node->v;
            rthip_instance* o = &insts[ii];
            o->type = (u32)in->type;
            o->material = -1;
            if (!rt_is_zero_handle(in->material)) {
                u64 key = in->material.v64[0];
                u32 slot = (u32)((key >> 4) * 0x9E3779B97F4A7C15ull >> 32) & (slots - 1);
                while (keys[slot] != 0 && keys[slot] != key) slot = (slot + 1) & (slots - 1);
                AssertAlways(keys[slot] == key); // a material that is not in this world's list
                o->material = (s32)vals[slot];
            }
            switch (in->type) {
                case RT_InstanceType_Sphere:{
                    o->sphere_center[0] = in->sphere.center.x;
                    o->sphere_center[1] = in->sphere.center.y;
                    o->sphere_center[2] = in->sphere.center.z;
                    o->sphere_radius = in->sphere.radius;
                }break;
                case RT_InstanceType_Mesh:{
                    RT_Mesh* mesh = rt_world_resolve_mesh(world, in->mesh.handle);
                    o->blas_id = (u32)mesh->blas_id;
                    o->translation[0] = in->mesh.translation.x;
                    o->translation[1] = in->mesh.translation.y;
                    o->translation[2] = in->mesh.translation.z;
                    o->rotation[0] = in->mesh.rotation.x;
                    o->rotation[1] = in->mesh.rotation.y;
                    o->rotation[2] = in->mesh.rotation.z;
                    o->rotation[3] = in->mesh.rotation.w;
                    o->scale[0] = in->mesh.scale.x;
                    o->scale[1] = in->mesh.scale.y;
                    o->scale[2] = in->mesh.scale.z;
                }break;
                default: break;
            }
            ii++;
        }
        if (tracer->group) rt_hip_check(tracer, rthip_group_tlas_build(tracer->group, insts, inst_count, mats, mat_count), "rthip_group_tlas_build");
        else rt_hip_check(tracer, rthip_tlas_build(tracer->ctx, insts, inst_count, mats, mat_count), "rthip_tlas_build");
    }}

    // The CPU backend writes the TLAS to "out.bvh" in every build flavour (its guard is `#ifdef BUILD_DEBUG`,
    // and the macro is always defined, raytracer_cpu.c:171-173).  Same here; RT_HIP_NO_DUMP=1 turns it off.
    #ifdef BUILD_DEBUG
    {
        const char* off = getenv("RT_HIP_NO_DUMP");
        if (!(off && off[0] == '1')) rt_hip_dump_tree(tracer, -1, "out.bvh");
    }
    #endif
}

rt_hook void rt_tracer_cleanup(RT_Handle handle) {
    RT_HIP_Tracer* tracer = rt_hip_handle_to_tracer(handle);
    free(tracer->material_handles);
    if (tracer->group) rthip_group_destroy(tracer->group);
    else rthip_destroy(tracer->ctx);
    arena_release(tracer->arena);
}

// rt_tracer_cast.  Every cast of a tracer uses the seed taken in rt_make_tracer plus the number of casts
// made so far.  RT_HIP_DUMP_F32=<path> additionally writes the raw float framebuffer (width*height*3 floats).
rt_hook void rt_tracer_cast(RT_Handle handle, RT_CastSettings settings, vec3_f32* out_radiance, int width, int height) {
    RT_HIP_Tracer* tracer = rt_hip_handle_to_tracer(handle);

    rthip_cast_settings c = zero_struct;
    c.eye[0] = settings.eye.x;           c.eye[1] = settings.eye.y;           c.eye[2] = settings.eye.z;
    c.up[0] = settings.up.x;             c.up[1] = settings.up.y;             c.up[2] = settings.up.z;
    c.forward[0] = settings.forward.x;   c.forward[1] = settings.forward.y;   c.forward[2] = settings.forward.z;
    c.right[0] = settings.right.x;       c.right[1] = settings.right.y;       c.right[2] = settings.right.z;
    c.viewport[0] = settings.viewport.x; c.viewport[1] = settings.viewport.y; c.viewport[2] = settings.viewport.z;
    c.samples = settings.samples;
    c.ior = settings.ior;
    c.defocus = settings.defocus ? 1u : 0u;
    c.defocus_disk[0] = settings.defocus_disk.x;
    c.defocus_disk[1] = settings.defocus_disk.y;
    c.orthographic = settings.orthographic ? 1u : 0u;

    u64 seed = tracer->seed + tracer->casts;
    tracer->casts++;
    if (tracer->group) rt_hip_check(tracer, rthip_group_cast(tracer->group, &c, seed, &out_radiance[0].v[0], width, height, 1), "rthip_group_cast");
    else rt_hip_check(tracer, rthip_cast(tracer->ctx, &c, seed, &out_radiance[0].v[0], width, height), "rthip_cast");
    tracer->last_frame = &out_radiance[0].v[0];

    const char* dump = getenv("RT_HIP_DUMP_F32");
    if (dump) {
        FILE* f = fopen(dump, "wb");
        if (f) {
            fwrite(out_radiance, sizeof(vec3_f32), (size_t)width*height, f);
            fclose(f);
        }
    }
    const char* hdr = getenv("RT_HIP_HDR");
    if (hdr) rt_hip_write_hdr(handle, hdr, width, height);

    // A BUILD_DEBUG build of the CPU backend leaves every ray of the cast in ./out.rays (raytracer_cpu.c:180-182,
    // 260-262, 355-357; read by extra/blender.py).  Same file here, from the records of the same frame (same seed).
    #if BUILD_DEBUG
    rt_hip_dump_rays(tracer, &c, seed, width, height, "out.rays");
    #endif
}

// The image of the last rt_tracer_cast as a Radiance .hdr file, byte for byte the file
// stbi_write_hdr(path, width, height, 3, radiance) writes (stb_image_write.h:794; the call every demo makes
// after its cast, e.g. src/demos/cornell/main.c:222), encoded on the GPU from the frame that is still in
// device memory: the host only writes the bytes out.  Returns 1 on success like stbi_write_hdr, 0 otherwise.
// RT_HIP_HDR=<path> makes rt_tracer_cast call it after every cast.
internal int rt_hip_write_hdr(RT_Handle handle, const char* path, int width, int height) {
    RT_HIP_Tracer* tracer = rt_hip_handle_to_tracer(handle);
    int ok = 0;
    {DeferResource(Temp scratch = scratch_begin(NULL, 0), scratch_end(scratch)) {
        u64 capacity = rthip_hdr_bound(width, height), size = 0;
        u8* bytes = push_array(scratch.arena, u8, capacity + 1);
        // one GPU: the frame of the last cast is still in its memory; several: it was assembled in the caller's buffer
        int rc = rthip_hdr_encode(tracer->ctx, tracer->group ? tracer->last_frame : NULL, width, height, bytes, capacity, &size);
        if (rc != RTHIP_OK) {
            fprintf(stderr, "rthip: rthip_hdr_encode failed (%d): %s\n", rc, rthip_last_error(tracer->ctx));
        } else {
            FILE* f = fopen(path, "wb");
            if (f) {
                ok = fwrite(bytes, 1, (size_t)size, f) == (size_t)size;
                fclose(f);
            }
        }
    }}
    return ok;
}


// ms_load_obj (src/mesh/mesh.c:73-219) with the text parsed on the host's cores and the vertex deduplication on the
// GPU (rthip_obj_load): same MS_Mesh -- same vertices in the same order, same indices -- in the caller's arena, about 50x
// faster on a 1 M-triangle file.  The demos load their meshes before they make a tracer (src/demos/bunny/main.c:41-52),
// so the loader keeps a GPU context of its own, made on first use.  Making it starts the GPU runtime, which draws from
// rand(); the stream is restarted from its own next value afterwards, so that what rt_make_tracer takes as its seed
// still depends on the demo's srand(--seed) alone.
static rthip_ctx* rt_hip_loader_ctx = NULL;
static void rt_hip_loader_release(void) {
    if (rt_hip_loader_ctx) rthip_destroy(rt_hip_loader_ctx);
    rt_hip_loader_ctx = NULL;
}
internal MS_LoadResult rt_hip_load_obj(Arena* arena, NTString8 path, MS_LoadSettings settings) {
    if (!rt_hip_loader_ctx) {
        atexit(rt_hip_loader_release);  // the loader's context lives until the program ends
        unsigned keep = (unsigned)rand();
        const char* dev = getenv("RT_HIP_DEVICE");
        int rc = rthip_create(&rt_hip_loader_ctx, dev ? atoi(dev) : 0);
        srand(keep);
        if (rc != RTHIP_OK) fprintf(stderr, "rthip: no usable MI355X (gfx950) device; this backend has no CPU path\n");
        AssertAlways(rc == RTHIP_OK);
    }
    MS_LoadResult result = zero_struct;
    rthip_obj_info info = zero_struct;
    if (rthip_obj_load(rt_hip_loader_ctx, path.cstr, (uint32_t)settings.attrs, &info) != RTHIP_OK) {
        fprintf(stderr, "rthip: %s\n", rthip_last_error(rt_hip_loader_ctx));
        result.error = (NTString8)ntstr8_lit_init("Failed to open file");  // the reference's only error, mesh.c:76-78
        return result;
    }
    result.v.attrs = (GEO_VertexAttributes)info.attrs;
    result.v.primitive = GEO_Primitive_TRI_LIST;
    result.v.vertices_count = info.vertex_count;
    result.v.indices_count = info.index_count;
    result.v.vertices = arena_push(arena, (u64)info.vertex_bytes*info.vertex_count + 16, 16);
    result.v.indices = push_array(arena, u32, info.index_count + 1);
    int rc = rthip_obj_get(rt_hip_loader_ctx, (float*)result.v.vertices, result.v.indices);
    if (rc != RTHIP_OK) fprintf(stderr, "rthip: %s\n", rthip_last_error(rt_hip_loader_ctx));
    AssertAlways(rc == RTHIP_OK);
    result.error = (NTString8)ntstr8_lit_init("");
    return result;
}
