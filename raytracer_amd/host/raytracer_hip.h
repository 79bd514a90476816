// HIP (MI355X / gfx950) backend of the raytracer's tracer hooks.
//
// This header/source pair is written for the reference's unity build: it is meant to be
// #included by src/raytracer/raytracer_inc.h / raytracer_inc.c next to the CPU backend
// (src/raytracer/cpu/raytracer_cpu.{h,c}) and implements the same five `rt_hook` functions
// (src/raytracer/raytracer_core.h:180-184) by flattening the RT_World lists into plain arrays
// and calling the C ABI of librthip.so (include/rthip.h).  It relies on the reference's own
// types and macros (RT_World, Arena, EachList, ...) and therefore only compiles inside the
// reference's translation unit.
#pragma once

#include "rthip.h"

typedef struct RT_HIP_Tracer RT_HIP_Tracer;
struct RT_HIP_Tracer {
    Arena* arena;
    rthip_ctx* ctx;         // the GPU context (with several GPUs: the first device's, for dumps and the .hdr encoder)
    rthip_group* group;     // RT_HIP_DEVICES=0,1,...: one context per listed GPU, every cast split by tiles; else NULL
    const float* last_frame;  // the caller's framebuffer of the last cast (the group's frame lives on the host only)
    u8 max_bounces;
    bool sky;
    u64 seed;   // first rand() after the caller's srand(): seeds the GPU generator
    u64 casts;  // casts made with this tracer
    u64* material_handles;  // handle (node address) of material i of the last rt_tracer_build_tlas, for out.rays
    u32 material_count;
};

internal RT_HIP_Tracer* rt_hip_handle_to_tracer(RT_Handle handle);
internal RT_Handle      rt_hip_tracer_to_handle(RT_HIP_Tracer* tracer);

// not one of the five hooks: the GPU-side replacement of the stbi_write_hdr call that follows rt_tracer_cast
internal int rt_hip_write_hdr(RT_Handle handle, const char* path, int width, int height);
// not a hook either: ms_load_obj's result from rthip_obj_load (parallel parse, GPU deduplication)
internal MS_LoadResult rt_hip_load_obj(Arena* arena, NTString8 path, MS_LoadSettings settings);
