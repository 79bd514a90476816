// The C ABI of include/rthip.h: context management, uploads, and the entry points that
// sequence the kernels of lbvh_build.hip and trace.hip.  No computation happens on the
// host: a missing or non-gfx950 GPU makes rthip_create fail.

#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rthip_internal.h"

int rthip_fail(rthip_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

int rthip_reserve(rthip_ctx* ctx, DevBuf& b, size_t bytes) {
    if (bytes == 0) bytes = 256;
    if (b.cap >= bytes) return RTHIP_OK;
    if (b.fixed)
        return rthip_fail(ctx, RTHIP_E_LIMIT, "a fixed workspace slice of %zu bytes cannot hold %zu bytes", b.cap, bytes);
    if (b.p) {
        (void)hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = (bytes + 255) & ~(size_t)255;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return rthip_fail(ctx, RTHIP_E_NOMEM, "hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
    }
    b.cap = want;
    return RTHIP_OK;
}

void rthip_release(DevBuf& b) {
    if (b.p && !b.fixed) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

namespace {

void release_mesh(HostMesh& m) {
    rthip_release(m.interleaved);
    rthip_release(m.indices);
    rthip_release(m.px);
    rthip_release(m.py);
    rthip_release(m.pz);
    rthip_release(m.nx);
    rthip_release(m.ny);
    rthip_release(m.nz);
    m = HostMesh();
}

DevScene make_scene(const rthip_ctx* ctx) {
    DevScene s;
    s.tlas = (const float4*)ctx->tlas_nodes.p;
    s.n_tlas = ctx->n_inst ? 2 * ctx->n_inst - 1 : 0;
    s.blas = (const float4*)ctx->blas_nodes.p;
    s.tlas_wide = (const float4*)ctx->tlas_wide.p;
    s.blas_wide = (const float4*)ctx->blas_wide.p;
    s.tri_v = (const float4*)ctx->tri_v.p;
    s.tri_n = (const float4*)ctx->tri_n.p;
    s.inst = (const DevInstance*)ctx->inst.p;
    s.mats = (const rthip_material*)ctx->mats.p;
    s.n_inst = ctx->n_inst;
    s.n_mats = ctx->n_mats;
    s.max_bounces = ctx->max_bounces;
    s.sky = ctx->sky ? 1u : 0u;
    return s;
}

int check_ready(rthip_ctx* ctx, const char* what) {
    if (!ctx) return RTHIP_E_INVALID;
    if (!ctx->tlas_built) return rthip_fail(ctx, RTHIP_E_INVALID, "%s: rthip_tlas_build has not run", what);
    return RTHIP_OK;
}

int check_cast(rthip_ctx* ctx, const rthip_cast_settings* cast, int width, int height) {
    if (!cast) return rthip_fail(ctx, RTHIP_E_INVALID, "cast settings are NULL");
    if (width <= 0 || height <= 0) return rthip_fail(ctx, RTHIP_E_INVALID, "image size %dx%d is not positive", width, height);
    if ((uint64_t)width * (uint64_t)height > (1ull << 31)) return rthip_fail(ctx, RTHIP_E_LIMIT, "image has more than 2^31 pixels");
    if (cast->samples == 0 || cast->samples > 255)
        return rthip_fail(ctx, RTHIP_E_INVALID, "samples must be in 1..255 (RT_CastSettings.samples is a u8), got %u", cast->samples);
    return RTHIP_OK;
}

uint64_t* counters_or_null(rthip_ctx* ctx) { return ctx->counting ? (uint64_t*)ctx->counters.p : nullptr; }

// RTHIP_TRAV_FAST: the records of the ordered walk for every committed mesh / for the TLAS (enqueued on ctx->stream)
int convert_blas_wide(rthip_ctx* ctx) {
    RT_TRY(rthip_reserve(ctx, ctx->blas_wide, (ctx->total_tris ? ctx->total_tris : 1) * 128));
    for (HostMesh& m : ctx->meshes)
        RT_TRY(wide_convert_device(ctx, (const float4*)ctx->blas_nodes.p + 2 * (size_t)m.dev.node_base, m.dev.n_nodes,
                                   (float4*)ctx->blas_wide.p + 8 * (size_t)m.dev.wide_base, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->blas_wide_ready = true;
    return RTHIP_OK;
}
int convert_tlas_wide(rthip_ctx* ctx) {
    RT_TRY(rthip_reserve(ctx, ctx->tlas_wide, (ctx->n_inst ? (size_t)ctx->n_inst : 1) * 128));
    if (ctx->n_inst)
        RT_TRY(wide_convert_device(ctx, (const float4*)ctx->tlas_nodes.p, 2 * ctx->n_inst - 1, (float4*)ctx->tlas_wide.p, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->tlas_wide_ready = true;
    return RTHIP_OK;
}

}  // namespace

void HostPool::start(uint32_t n, int device) {
    for (uint32_t i = 0; i < n; i++)
        threads.emplace_back([this, device] {
            (void)hipSetDevice(device);
            uint64_t seen = 0;
            for (;;) {
                std::unique_lock<std::mutex> lock(m);
                wake.wait(lock, [&] { return stop || generation != seen; });
                if (stop) return;
                seen = generation;
                while (next < pieces) {
                    const uint32_t mine = next++;
                    lock.unlock();
                    job(mine);
                    lock.lock();
                    if (++finished == pieces) done.notify_all();
                }
            }
        });
}

void HostPool::run(uint32_t n_pieces, std::function<void(uint32_t)> fn) {
    std::unique_lock<std::mutex> lock(m);
    job = std::move(fn);
    pieces = n_pieces;
    next = 0;
    finished = 0;
    generation++;
    wake.notify_all();
    done.wait(lock, [&] { return finished == pieces; });
}

HostPool::~HostPool() {
    {
        std::lock_guard<std::mutex> lock(m);
        stop = true;
    }
    wake.notify_all();
    for (std::thread& t : threads) t.join();
}

extern "C" {

int rthip_abi_version(void) { return RTHIP_ABI_VERSION; }

const char* rthip_last_error(const rthip_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int rthip_create(rthip_ctx** out_ctx, int device_id) {
    if (!out_ctx) return RTHIP_E_INVALID;
    *out_ctx = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device_id < 0 || device_id >= count) return RTHIP_E_HIP;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return RTHIP_E_HIP;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return RTHIP_E_HIP;  // the kernels exist for gfx950 only
    if (hipSetDevice(device_id) != hipSuccess) return RTHIP_E_HIP;
    rthip_ctx* ctx = new rthip_ctx();
    ctx->device = device_id;
    const char* fence = getenv("RTHIP_REFIT_FENCE");
    ctx->refit_fence = (fence && fence[0] >= '0' && fence[0] <= '2') ? fence[0] - '0' : 2;
    ctx->refit_checked = fence && fence[0] >= '0' && fence[0] <= '2';  // an explicit choice is taken as it is
    const char* by_passes = getenv("RTHIP_SORT_BY_PASSES");
    ctx->sort_by_passes = (by_passes && by_passes[0] == '1') ? 1 : 0;
    const char* top_bytes = getenv("RTHIP_SORT_TOP_BYTES");
    if (top_bytes && (top_bytes[0] == '4' || top_bytes[0] == '8')) ctx->sort_top_bytes = top_bytes[0] - '0';
    const char* small_tile = getenv("RTHIP_SORT_SMALL_TILE_MAX");
    if (small_tile && small_tile[0]) ctx->sort_small_tile_max = strtoull(small_tile, nullptr, 10);
    const char* merge_tile = getenv("RTHIP_SORT_MERGE_TILE");
    if (merge_tile && (atoi(merge_tile) == 1024 || atoi(merge_tile) == 2048 || atoi(merge_tile) == 4096)) ctx->sort_merge_tile = (uint32_t)atoi(merge_tile);
    const char* merge_max = getenv("RTHIP_SORT_MERGE_MAX");
    if (merge_max && merge_max[0]) ctx->sort_merge_max = strtoull(merge_max, nullptr, 10);
    if (ctx->sort_merge_max > 128ull * (ctx->sort_merge_tile ? ctx->sort_merge_tile : 2048u))  // tile_merge_kernel: 32 tiles per wave
        ctx->sort_merge_max = 128ull * (ctx->sort_merge_tile ? ctx->sort_merge_tile : 2048u);
    const char* ranges = getenv("RTHIP_UPPER_RANGES");
    if (ranges && ranges[0] >= '0' && ranges[0] <= '2') ctx->upper_ranges = ranges[0] - '0';
    const char* wide = getenv("RTHIP_UPPER_DEVICE_WIDE");
    ctx->upper_device_wide = (wide && wide[0] == '1') ? 1 : 0;
    const char* mode = getenv("RTHIP_CAST");
    if (mode && strcmp(mode, "static") == 0) ctx->cast_mode = 1;
    if (mode && strcmp(mode, "static-samples") == 0) ctx->cast_mode = 4;
    if (mode && strcmp(mode, "wavefront") == 0) ctx->cast_mode = 2;
    if (mode && strcmp(mode, "persistent") == 0) ctx->cast_mode = 0;
    if (mode && strcmp(mode, "auto") == 0) ctx->cast_mode = 3;
    ctx->persistent_blocks = (uint32_t)prop.multiProcessorCount * 5u;
    ctx->cus = (uint32_t)prop.multiProcessorCount;
    const char* per_cu = getenv("RTHIP_PERSISTENT_PER_CU");  // resident workgroups per CU the persistent kernels are launched for
    if (per_cu && atoi(per_cu) >= 1 && atoi(per_cu) <= 8) {
        ctx->persistent_blocks = (uint32_t)prop.multiProcessorCount * (uint32_t)atoi(per_cu);
        ctx->persistent_blocks_forced = true;
    }
    const char* cast_bands = getenv("RTHIP_CAST_BANDS");
    if (cast_bands && atoi(cast_bands) >= 1 && atoi(cast_bands) <= rthip_ctx::kMaxBands) ctx->cast_bands = atoi(cast_bands);
    const char* ib = getenv("RTHIP_INST_BATCH");
    if (ib && atoi(ib) >= 0 && atoi(ib) <= 64) ctx->inst_batch = (uint32_t)atoi(ib);
    const char* refill = getenv("RTHIP_REFILL");
    const char* sib = getenv("RTHIP_SAMPLES_INST_BATCH");
    if (sib && atoi(sib) >= 0 && atoi(sib) <= 64) ctx->samples_inst_batch_env = atoi(sib);
    const char* rib = getenv("RTHIP_RAYS_INST_BATCH");
    if (rib && atoi(rib) >= 0 && atoi(rib) <= 64) ctx->rays_inst_batch = (uint32_t)atoi(rib);
    const char* fib = getenv("RTHIP_FAST_INST_BATCH");
    if (fib && atoi(fib) >= 0 && atoi(fib) <= 64) ctx->fast_inst_batch = (uint32_t)atoi(fib);
    if (refill && atoi(refill) >= 1 && atoi(refill) <= 64) ctx->refill_lanes = (uint32_t)atoi(refill);
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess ||
        rthip_reserve(ctx, ctx->counters, 64) != RTHIP_OK || hipMemset(ctx->counters.p, 0, 64) != hipSuccess ||
        rthip_reserve(ctx, ctx->work_ring, kWorkSlots * 1024) != RTHIP_OK ||
        hipMemset(ctx->work_ring.p, 0, kWorkSlots * 1024) != hipSuccess) {
        rthip_destroy(ctx);
        return RTHIP_E_HIP;
    }
    // the address of the static build workspace on this device (used by rthip_lbvh_build_device_noalloc); a failure
    // here is reported by that entry point, which tries again
    (void)lbvh_static_workspace_init(ctx);
    // where large builds leave word of how their sort went (read by the next build; lbvh_build_device): without it
    // builds still work, they only never learn to skip the attempt
    if (hipHostMalloc((void**)&ctx->sort_flag_host, 64, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&ctx->sort_flag_ev, hipEventDisableTiming) != hipSuccess) {
        (void)hipGetLastError();
        if (ctx->sort_flag_host) (void)hipHostFree(ctx->sort_flag_host);
        ctx->sort_flag_host = nullptr;
    } else {
        *ctx->sort_flag_host = 0;
    }
    ctx->err.clear();
    *out_ctx = ctx;
    return RTHIP_OK;
}

void rthip_destroy(rthip_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (const auto& r : ctx->pinned) (void)hipHostFree(r.p);
    for (auto& m : ctx->meshes) release_mesh(m);
    LbvhWorkspace& w = ctx->ws;
    DevBuf* all[] = {&w.aabb, &w.keys[0], &w.keys[1], &w.vals[0], &w.vals[1], &w.block_hist, &w.scan_aux, &w.prio,
                     &w.child, &w.vis, &w.flags, &w.starters, &w.lefts, &w.bounds, &ctx->blas_nodes, &ctx->tri_v, &ctx->tri_n,
                     &ctx->blas_wide, &ctx->tlas_wide, &ctx->wide_rank, &ctx->tlas_nodes, &ctx->inst, &ctx->mats, &ctx->inst_aabb, &ctx->counters, &ctx->work_ring, &ctx->scratch_a,
                     &ctx->scratch_b, &ctx->scratch_c, &ctx->scratch_d, &ctx->scratch_e, &ctx->frame, &ctx->hdr_stage,
                     &ctx->hdr_lengths, &ctx->hdr_offsets, &ctx->hdr_out, &ctx->hdr_size, &ctx->obj.cp, &ctx->obj.cn, &ctx->obj.ct,
                     &ctx->obj.pos, &ctx->obj.nrm, &ctx->obj.uv, &ctx->obj.head, &ctx->obj.first, &ctx->obj.run_first,
                     &ctx->obj.indices, &ctx->obj.px, &ctx->obj.py, &ctx->obj.pz, &ctx->obj.nx, &ctx->obj.ny, &ctx->obj.nz,
                     &ctx->obj.interleaved};
    for (DevBuf* b : all) rthip_release(*b);
    for (DevBuf& b : ctx->wf) rthip_release(b);
    for (hipEvent_t e : ctx->slot_done)
        if (e) (void)hipEventDestroy(e);
    delete ctx->movers;
    ctx->movers = nullptr;
    if (ctx->host_stage) (void)hipHostFree(ctx->host_stage);
    if (ctx->inst_stage) (void)hipHostFree(ctx->inst_stage);
    for (hipStream_t b : ctx->band_stream)
        if (b) (void)hipStreamDestroy(b);
    for (hipStream_t b : ctx->band_copy_stream)
        if (b) (void)hipStreamDestroy(b);
    for (hipEvent_t e : ctx->band_done)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->chunk_done)
        if (e) (void)hipEventDestroy(e);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->sort_flag_ev) (void)hipEventDestroy(ctx->sort_flag_ev);
    if (ctx->sort_flag_host) (void)hipHostFree(ctx->sort_flag_host);
    rthip_release(ctx->tune_counters);
    rthip_release(ctx->tune_frame);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int rthip_set_tracer(rthip_ctx* ctx, uint32_t max_bounces, int sky) {
    if (!ctx) return RTHIP_E_INVALID;
    if (max_bounces >= 64)  // Assert(settings.max_bounces < RT_MAX_MAX_BOUNCES), raytracer_cpu.c:15
        return rthip_fail(ctx, RTHIP_E_LIMIT, "max_bounces must be < 64, got %u", max_bounces);
    ctx->max_bounces = max_bounces;
    ctx->sky = sky ? 1 : 0;
    ctx->scene_generation++;
    return RTHIP_OK;
}

int rthip_set_morton_bounds(rthip_ctx* ctx, int mode) {
    if (!ctx) return RTHIP_E_INVALID;
    if (mode != RTHIP_BOUNDS_CENTROIDS && mode != RTHIP_BOUNDS_BOXES)
        return rthip_fail(ctx, RTHIP_E_INVALID, "morton bounds mode must be 0 (centroids) or 1 (boxes), got %d", mode);
    ctx->box_bounds = (uint32_t)mode;
    ctx->bounds_of = nullptr;  // bounds left by a prepare pass were taken in the other mode
    return RTHIP_OK;
}

int rthip_set_traversal(rthip_ctx* ctx, int mode) {
    if (!ctx) return RTHIP_E_INVALID;
    if (mode != RTHIP_TRAV_EXACT && mode != RTHIP_TRAV_FAST)
        return rthip_fail(ctx, RTHIP_E_INVALID, "traversal mode must be RTHIP_TRAV_EXACT (0) or RTHIP_TRAV_FAST (1), got %d", mode);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (mode == RTHIP_TRAV_FAST) {  // trees built so far get their records now, later ones when they are built
        RT_HIP(ctx, hipDeviceSynchronize());
        if (ctx->blas_committed && !ctx->blas_wide_ready) RT_TRY(convert_blas_wide(ctx));
        if (ctx->tlas_built && !ctx->tlas_wide_ready) RT_TRY(convert_tlas_wide(ctx));
    }
    ctx->traversal = mode;
    ctx->scene_generation++;  // measured kernel choices belong to the other walk
    return RTHIP_OK;
}

int rthip_set_counting(rthip_ctx* ctx, int enabled) {
    if (!ctx) return RTHIP_E_INVALID;
    ctx->counting = enabled ? 1 : 0;
    return RTHIP_OK;
}

int rthip_set_cast_kernel(rthip_ctx* ctx, int kernel) {
    if (!ctx) return RTHIP_E_INVALID;
    if (kernel < 0 || kernel > 4) return rthip_fail(ctx, RTHIP_E_INVALID, "unknown cast kernel %d", kernel);
    ctx->cast_mode = kernel;
    return RTHIP_OK;
}

int rthip_set_persistent_params(rthip_ctx* ctx, uint32_t refill, uint32_t inst_batch) {
    if (!ctx) return RTHIP_E_INVALID;
    if (refill < 1 || refill > 64 || inst_batch > 64)
        return rthip_fail(ctx, RTHIP_E_INVALID, "refill must be in 1..64 and inst_batch in 0..64, got %u and %u", refill, inst_batch);
    ctx->refill_lanes = refill;
    ctx->inst_batch = inst_batch;
    return RTHIP_OK;
}

// ---- BLAS ----------------------------------------------------------------------------

int rthip_blas_begin(rthip_ctx* ctx, uint32_t n_meshes) {
    if (!ctx) return RTHIP_E_INVALID;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipDeviceSynchronize());
    for (auto& m : ctx->meshes) release_mesh(m);
    ctx->meshes.clear();
    ctx->meshes.resize(n_meshes);
    ctx->blas_committed = false;
    ctx->tlas_built = false;
    ctx->blas_wide_ready = ctx->tlas_wide_ready = false;
    ctx->total_blas_nodes = 0;
    ctx->total_tris = 0;
    return RTHIP_OK;
}

int rthip_blas_add_mesh(rthip_ctx* ctx, uint32_t blas_id, const void* verts, uint32_t vertex_count,
                        uint32_t stride_bytes, uint32_t pos_off, int32_t nrm_off, const uint32_t* idx,
                        uint32_t index_count) {
    if (!ctx) return RTHIP_E_INVALID;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (blas_id >= ctx->meshes.size()) return rthip_fail(ctx, RTHIP_E_INVALID, "blas_id %u out of range (%zu meshes)", blas_id, ctx->meshes.size());
    if (ctx->blas_committed) return rthip_fail(ctx, RTHIP_E_INVALID, "rthip_blas_add_mesh after rthip_blas_commit");
    if (!verts || vertex_count == 0) return rthip_fail(ctx, RTHIP_E_INVALID, "mesh %u has no vertices", blas_id);
    if (stride_bytes < 12 || (stride_bytes & 3) || (pos_off & 3) || pos_off + 12 > stride_bytes)
        return rthip_fail(ctx, RTHIP_E_INVALID, "mesh %u: bad stride/offset (%u, %u)", blas_id, stride_bytes, pos_off);
    if (nrm_off >= 0 && (((uint32_t)nrm_off & 3) || (uint32_t)nrm_off + 12 > stride_bytes))
        return rthip_fail(ctx, RTHIP_E_INVALID, "mesh %u: bad normal offset %d", blas_id, nrm_off);
    const bool indexed = idx != nullptr && index_count != 0;
    const uint32_t n_tris = indexed ? index_count / 3 : vertex_count / 3;  // raytracer_cpu.c:63
    if (n_tris == 0) return rthip_fail(ctx, RTHIP_E_INVALID, "mesh %u has no triangles", blas_id);
    if (indexed) {
        for (uint32_t k = 0; k < n_tris * 3; k++)
            if (idx[k] >= vertex_count)
                return rthip_fail(ctx, RTHIP_E_INVALID, "mesh %u: index %u at position %u exceeds vertex count %u", blas_id, idx[k], k, vertex_count);
    }
    HostMesh& m = ctx->meshes[blas_id];
    release_mesh(m);
    m.vertex_count = vertex_count;
    m.stride = stride_bytes;
    m.pos_off = pos_off;
    m.nrm_off = nrm_off;
    m.index_count = indexed ? n_tris * 3 : 0;
    m.n_tris = n_tris;
    const size_t vbytes = (size_t)vertex_count * stride_bytes;
    RT_TRY(rthip_reserve(ctx, m.interleaved, vbytes));
    RT_HIP(ctx, hipMemcpyAsync(m.interleaved.p, verts, vbytes, hipMemcpyHostToDevice, ctx->stream));
    if (indexed) {
        RT_TRY(rthip_reserve(ctx, m.indices, (size_t)m.index_count * 4));
        RT_HIP(ctx, hipMemcpyAsync(m.indices.p, idx, (size_t)m.index_count * 4, hipMemcpyHostToDevice, ctx->stream));
    }
    const size_t fbytes = (size_t)vertex_count * 4;
    RT_TRY(rthip_reserve(ctx, m.px, fbytes));
    RT_TRY(rthip_reserve(ctx, m.py, fbytes));
    RT_TRY(rthip_reserve(ctx, m.pz, fbytes));
    RT_TRY(deinterleave_device(ctx, m.interleaved.p, vertex_count, stride_bytes, pos_off, (float*)m.px.p,
                               (float*)m.py.p, (float*)m.pz.p, ctx->stream));
    if (nrm_off >= 0) {
        RT_TRY(rthip_reserve(ctx, m.nx, fbytes));
        RT_TRY(rthip_reserve(ctx, m.ny, fbytes));
        RT_TRY(rthip_reserve(ctx, m.nz, fbytes));
        RT_TRY(deinterleave_device(ctx, m.interleaved.p, vertex_count, stride_bytes, (uint32_t)nrm_off,
                                   (float*)m.nx.p, (float*)m.ny.p, (float*)m.nz.p, ctx->stream));
    }
    // the caller's memory must not be referenced after this call returns
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    m.present = true;
    return RTHIP_OK;
}

// ---- mesh files -----------------------------------------------------------------------

int rthip_obj_load(rthip_ctx* ctx, const char* path, uint32_t attrs, rthip_obj_info* out_info) {
    if (!ctx) return RTHIP_E_INVALID;
    if (!path) return rthip_fail(ctx, RTHIP_E_INVALID, "path is NULL");
    return obj_load(ctx, path, attrs, out_info);
}

int rthip_obj_get(rthip_ctx* ctx, float* out_vertices, uint32_t* out_indices) {
    if (!ctx) return RTHIP_E_INVALID;
    return obj_read_back(ctx, out_vertices, out_indices);
}

int rthip_blas_add_loaded_obj(rthip_ctx* ctx, uint32_t blas_id) {
    if (!ctx) return RTHIP_E_INVALID;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const ObjState& o = ctx->obj;
    if (!o.loaded) return rthip_fail(ctx, RTHIP_E_INVALID, "no OBJ file has been loaded");
    if (blas_id >= ctx->meshes.size()) return rthip_fail(ctx, RTHIP_E_INVALID, "blas_id %u out of range (%zu meshes)", blas_id, ctx->meshes.size());
    if (ctx->blas_committed) return rthip_fail(ctx, RTHIP_E_INVALID, "rthip_blas_add_loaded_obj after rthip_blas_commit");
    if (o.index_count < 3 || o.vertex_count == 0) return rthip_fail(ctx, RTHIP_E_INVALID, "the loaded file has no triangles");
    HostMesh& m = ctx->meshes[blas_id];
    release_mesh(m);
    const bool normals = (o.attrs & 2u) != 0;
    m.vertex_count = o.vertex_count;
    m.stride = 16u * ((o.attrs & 1u) + ((o.attrs >> 1) & 1u) + ((o.attrs >> 2) & 1u));
    m.pos_off = 0;
    m.nrm_off = normals ? 16 : -1;
    m.n_tris = o.index_count / 3;
    m.index_count = m.n_tris * 3;
    const size_t fbytes = (size_t)o.vertex_count * 4;
    RT_TRY(rthip_reserve(ctx, m.indices, (size_t)m.index_count * 4));
    RT_HIP(ctx, hipMemcpyAsync(m.indices.p, o.indices.p, (size_t)m.index_count * 4, hipMemcpyDeviceToDevice, ctx->stream));
    DevBuf* dst[6] = {&m.px, &m.py, &m.pz, &m.nx, &m.ny, &m.nz};
    const DevBuf* src[6] = {&o.px, &o.py, &o.pz, &o.nx, &o.ny, &o.nz};
    for (int k = 0; k < (normals ? 6 : 3); k++) {
        RT_TRY(rthip_reserve(ctx, *dst[k], fbytes));
        RT_HIP(ctx, hipMemcpyAsync(dst[k]->p, src[k]->p, fbytes, hipMemcpyDeviceToDevice, ctx->stream));
    }
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    m.present = true;
    return RTHIP_OK;
}

int rthip_blas_commit(rthip_ctx* ctx) {
    if (!ctx) return RTHIP_E_INVALID;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->blas_committed) return rthip_fail(ctx, RTHIP_E_INVALID, "rthip_blas_commit: this mesh set is already committed (start another with rthip_blas_begin)");
    uint64_t nodes = 0, tris = 0;
    bool any_normals = false;
    for (size_t k = 0; k < ctx->meshes.size(); k++) {
        HostMesh& m = ctx->meshes[k];
        if (!m.present) return rthip_fail(ctx, RTHIP_E_INVALID, "mesh %zu was never added", k);
        m.dev.node_base = (uint32_t)nodes;
        m.dev.wide_base = (uint32_t)(tris - k);  // room for T - 1 records per mesh of T triangles (it has T - 1 inner nodes; about half of them get a record)
        m.dev.n_nodes = 2 * m.n_tris - 1;
        m.dev.tri_base = (uint32_t)tris;
        m.dev.n_tris = m.n_tris;
        m.dev.has_normals = m.nrm_off >= 0 ? 1u : 0u;
        any_normals |= m.nrm_off >= 0;
        nodes += m.dev.n_nodes;
        tris += m.n_tris;
        if (nodes > 0xffffffffull) return rthip_fail(ctx, RTHIP_E_LIMIT, "more than 2^32 BLAS nodes in total");
    }
    ctx->total_blas_nodes = nodes;
    ctx->total_tris = tris;
    RT_TRY(rthip_reserve(ctx, ctx->blas_nodes, nodes * 32));
    RT_TRY(rthip_reserve(ctx, ctx->tri_v, tris * 48));
    if (any_normals) RT_TRY(rthip_reserve(ctx, ctx->tri_n, tris * 48));
    // all meshes are enqueued on the context's stream (the builds share the workspace, so they run one after the other
    // there anyway) and the host waits once; the workspace is sized for the largest mesh first so that no build
    // reallocates under an earlier one
    uint32_t largest = 0;
    for (const HostMesh& m : ctx->meshes) largest = m.n_tris > largest ? m.n_tris : largest;
    RT_TRY(rthip_reserve(ctx, ctx->ws.aabb, (size_t)largest * 24));
    RT_TRY(lbvh_reserve_workspace(ctx, largest));
    RT_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    for (HostMesh& m : ctx->meshes) {
        RT_TRY(mesh_prepare_device(ctx, (const float*)m.px.p, (const float*)m.py.p, (const float*)m.pz.p,
                                   (const float*)m.nx.p, (const float*)m.ny.p, (const float*)m.nz.p,
                                   m.index_count ? (const uint32_t*)m.indices.p : nullptr, m.n_tris,
                                   (float*)ctx->ws.aabb.p, (float4*)ctx->tri_v.p + 3 * (size_t)m.dev.tri_base,
                                   m.nrm_off >= 0 ? (float4*)ctx->tri_n.p + 3 * (size_t)m.dev.tri_base : nullptr,
                                   ctx->stream));
        RT_TRY(lbvh_build_device(ctx, (const float*)ctx->ws.aabb.p, m.n_tris,
                                 (float4*)ctx->blas_nodes.p + 2 * (size_t)m.dev.node_base, ctx->stream, nullptr,
                                 nullptr, nullptr, true));
    }
    RT_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    float total_ms = 0.f;
    RT_HIP(ctx, hipEventElapsedTime(&total_ms, ctx->ev0, ctx->ev1));
    // the staging copies (the caller's interleaved buffer, SoA attributes, indices) are not read after this: the walk
    // uses the per-triangle records and the nodes only
    for (HostMesh& m : ctx->meshes) {
        DevBuf* staging[] = {&m.interleaved, &m.indices, &m.px, &m.py, &m.pz, &m.nx, &m.ny, &m.nz};
        for (DevBuf* b : staging) rthip_release(*b);
    }
    ctx->last_build_ms = total_ms;
    ctx->blas_wide_ready = false;
    if (ctx->traversal == RTHIP_TRAV_FAST) RT_TRY(convert_blas_wide(ctx));
    ctx->blas_committed = true;
    ctx->scene_generation++;
    ctx->capture_next = 0;  // graphs captured over the previous scene are void
    return RTHIP_OK;
}

// ---- TLAS ----------------------------------------------------------------------------

int rthip_tlas_build(rthip_ctx* ctx, const rthip_instance* instances, uint32_t n_instances,
                     const rthip_material* materials, uint32_t n_materials) {
    if (!ctx) return RTHIP_E_INVALID;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->blas_committed) return rthip_fail(ctx, RTHIP_E_INVALID, "rthip_tlas_build before rthip_blas_commit");
    if (n_instances && !instances) return rthip_fail(ctx, RTHIP_E_INVALID, "instances is NULL");
    if (n_materials && !materials) return rthip_fail(ctx, RTHIP_E_INVALID, "materials is NULL");
    if (n_instances > (1u << 30)) return rthip_fail(ctx, RTHIP_E_LIMIT, "more than 2^30 instances");
    // The device's records are written straight into page-locked memory of the context (grown on demand), from where the
    // copy is one DMA; out of a pageable vector the runtime stages it itself, which took longer than the TLAS build.
    std::vector<DevInstance> pageable;
    const size_t stage_bytes = (size_t)n_instances * sizeof(DevInstance);
    if (ctx->inst_stage_bytes < stage_bytes) {
        if (ctx->inst_stage) (void)hipHostFree(ctx->inst_stage);
        ctx->inst_stage = nullptr;
        ctx->inst_stage_bytes = 0;
        const size_t want = stage_bytes + stage_bytes / 4 + 4096;
        if (hipHostMalloc(&ctx->inst_stage, want, hipHostMallocDefault) == hipSuccess && ctx->inst_stage) ctx->inst_stage_bytes = want;
        else {
            (void)hipGetLastError();
            ctx->inst_stage = nullptr;
        }
    }
    if (!ctx->inst_stage) pageable.resize(n_instances);
    DevInstance* dev = ctx->inst_stage ? (DevInstance*)ctx->inst_stage : pageable.data();
    for (uint32_t k = 0; k < n_instances; k++) {
        const rthip_instance& s = instances[k];
        DevInstance& d = dev[k];
        memset(&d, 0, sizeof d);
        if (s.material >= (int32_t)n_materials)
            return rthip_fail(ctx, RTHIP_E_INVALID, "instance %u: material %d out of range (%u materials)", k, s.material, n_materials);
        d.type = s.type;
        d.material = s.material < 0 ? -1 : s.material;
        if (s.type == RTHIP_INST_SPHERE) {
            memcpy(d.sphere_center, s.sphere_center, 12);
            d.sphere_radius = s.sphere_radius;
        } else if (s.type == RTHIP_INST_MESH) {
            if (s.blas_id >= ctx->meshes.size())
                return rthip_fail(ctx, RTHIP_E_INVALID, "instance %u: blas_id %u out of range (%zu meshes)", k, s.blas_id, ctx->meshes.size());
            const DevMesh& m = ctx->meshes[s.blas_id].dev;
            d.node_base = m.node_base;
            d.wide_base = m.wide_base;
            d.n_nodes = m.n_nodes;
            d.tri_base = m.tri_base;
            d.has_normals = m.has_normals;
            memcpy(d.translation, s.translation, 12);
            memcpy(d.rotation, s.rotation, 16);
            memcpy(d.scale, s.scale, 12);
        } else {
            return rthip_fail(ctx, RTHIP_E_INVALID, "instance %u: unknown type %u", k, s.type);
        }
    }
    for (uint32_t k = 0; k < n_materials; k++)
        if (materials[k].type > RTHIP_MAT_LIGHT)
            return rthip_fail(ctx, RTHIP_E_INVALID, "material %u: unknown type %u", k, materials[k].type);

    {   // mesh instances of a few triangles each (at most 32 nodes on average): rays enter an instance every few steps
        uint64_t mesh_instances = 0, nodes = 0;
        for (uint32_t k = 0; k < n_instances; k++)
            if (dev[k].type == RTHIP_INST_MESH) {
                mesh_instances++;
                nodes += dev[k].n_nodes;
            }
        ctx->small_meshes = mesh_instances != 0 && nodes <= 32 * mesh_instances;
    }
    RT_HIP(ctx, hipDeviceSynchronize());  // the previous structure may still be in use by an enqueued cast
    ctx->tlas_built = false;
    ctx->n_inst = n_instances;
    ctx->scene_generation++;
    ctx->capture_next = 0;  // graphs captured over the previous scene are void
    ctx->n_mats = n_materials;
    RT_TRY(rthip_reserve(ctx, ctx->inst, (size_t)n_instances * sizeof(DevInstance)));
    RT_TRY(rthip_reserve(ctx, ctx->mats, (size_t)n_materials * sizeof(rthip_material)));
    RT_TRY(rthip_reserve(ctx, ctx->inst_aabb, (size_t)n_instances * 24));
    RT_TRY(rthip_reserve(ctx, ctx->tlas_nodes, n_instances ? (2 * (size_t)n_instances - 1) * 32 : 32));
    if (n_materials)
        RT_HIP(ctx, hipMemcpyAsync(ctx->mats.p, materials, (size_t)n_materials * sizeof(rthip_material), hipMemcpyHostToDevice, ctx->stream));
    if (n_instances) {
        RT_HIP(ctx, hipMemcpyAsync(ctx->inst.p, dev, (size_t)n_instances * sizeof(DevInstance), hipMemcpyHostToDevice, ctx->stream));
        RT_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
        RT_TRY(instance_aabbs_device(ctx, (const DevInstance*)ctx->inst.p, n_instances, (const float4*)ctx->blas_nodes.p,
                                     (float*)ctx->inst_aabb.p, ctx->stream));
        RT_TRY(lbvh_build_device(ctx, (const float*)ctx->inst_aabb.p, n_instances, (float4*)ctx->tlas_nodes.p, ctx->stream,
                                 nullptr, nullptr, nullptr, true));
        RT_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    }
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));  // dev[] must outlive the copy
    if (n_instances) RT_HIP(ctx, hipEventElapsedTime(&ctx->last_build_ms, ctx->ev0, ctx->ev1));
    ctx->tlas_wide_ready = false;
    if (ctx->traversal == RTHIP_TRAV_FAST) RT_TRY(convert_tlas_wide(ctx));
    ctx->tlas_built = true;
    return RTHIP_OK;
}

// ---- rendering ---------------------------------------------------------------------

int rthip_cast_device(rthip_ctx* ctx, const rthip_cast_settings* cast, uint64_t seed, float* d_out_rgb, int width,
                      int height, const rthip_shard* shard, void* stream) {
    RT_TRY(check_ready(ctx, "rthip_cast_device"));
    RT_TRY(check_cast(ctx, cast, width, height));
    if (!d_out_rgb) return rthip_fail(ctx, RTHIP_E_INVALID, "framebuffer is NULL");
    rthip_shard sh = {0, 0, 1};
    if (shard) {
        sh = *shard;
        if (sh.mode > 3 || (sh.mode != 0 && (sh.world == 0 || sh.rank >= sh.world)))
            return rthip_fail(ctx, RTHIP_E_INVALID, "bad shard {mode %u, rank %u, world %u}", sh.mode, sh.rank, sh.world);
    }
    RT_HIP(ctx, hipSetDevice(ctx->device));
    return cast_launch(ctx, make_scene(ctx), *cast, seed, d_out_rgb, width, height, sh, counters_or_null(ctx),
                       (hipStream_t)stream);
}

int rthip_cast_accumulate_device(rthip_ctx* ctx, const rthip_cast_settings* cast, uint64_t seed, float* d_accum_rgb, int width,
                                 int height, const rthip_shard* shard, void* stream) {
    if (!ctx) return RTHIP_E_INVALID;
    ctx->cast_accumulate = 1;
    const int rc = rthip_cast_device(ctx, cast, seed, d_accum_rgb, width, height, shard, stream);
    ctx->cast_accumulate = 0;
    return rc;
}

int rthip_host_alloc(rthip_ctx* ctx, uint64_t bytes, void** out_ptr) {
    if (!ctx) return RTHIP_E_INVALID;
    if (!out_ptr || bytes == 0) return rthip_fail(ctx, RTHIP_E_INVALID, "rthip_host_alloc: empty request");
    *out_ptr = nullptr;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    void* p = nullptr;
    const hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
    if (e != hipSuccess || !p) return rthip_fail(ctx, RTHIP_E_NOMEM, "hipHostMalloc(%llu bytes) failed: %s", (unsigned long long)bytes, hipGetErrorString(e));
    ctx->pinned.push_back({p, (size_t)bytes});
    *out_ptr = p;
    return RTHIP_OK;
}

int rthip_host_free(rthip_ctx* ctx, void* ptr) {
    if (!ctx) return RTHIP_E_INVALID;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    for (size_t k = 0; k < ctx->pinned.size(); k++)
        if (ctx->pinned[k].p == ptr) {
            RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            RT_HIP(ctx, hipHostFree(ptr));
            ctx->pinned.erase(ctx->pinned.begin() + (long)k);
            return RTHIP_OK;
        }
    return rthip_fail(ctx, RTHIP_E_INVALID, "rthip_host_free: %p does not come from rthip_host_alloc of this context", ptr);
}

// Whether a host pointer is page-locked memory the device can copy into without staging (rthip_host_alloc, hipHostMalloc).
static bool host_pointer_is_pinned(const void* p) {
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();  // an ordinary malloc'ed pointer is reported as an error by some runtimes
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

int rthip_cast(rthip_ctx* ctx, const rthip_cast_settings* cast, uint64_t seed, float* out_rgb, int width, int height) {
    RT_TRY(check_ready(ctx, "rthip_cast"));
    RT_TRY(check_cast(ctx, cast, width, height));
    if (!out_rgb) return rthip_fail(ctx, RTHIP_E_INVALID, "out_rgb is NULL");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)width * height * 3 * sizeof(float);
    ctx->frame_w = ctx->frame_h = 0;
    RT_TRY(rthip_reserve(ctx, ctx->frame, bytes));
    // A large frame goes to the host in bands of rows (rthip_shard mode 3): every band is rendered on a stream of its
    // own with its copy on a copy stream behind it, so the rows that are finished early (sky) are on their way while
    // the bands that hold the scene still render, and the bands overlap like frames on several streams.  Into
    // page-locked memory the copies are DMA straight into the caller's buffer.  Into ordinary (pageable) memory the
    // runtime's own staged copy holds the calling thread for the whole frame (1.4 ms for 1920x1080 floats); here the
    // bands are copied by DMA into page-locked staging of the context and from there into the caller's buffer by one
    // host thread per band, each as soon as its band has arrived.
    uint32_t bands = (uint32_t)ctx->cast_bands;
    // (the staged renderer keeps its path state in one set of arrays per context: one launch at a time)
    if ((uint64_t)width * height < (1ull << 19) || ctx->counting || ctx->cast_mode == 2 || (uint32_t)height < 64u * bands) bands = 1;
    const bool pinned = bands > 1 && host_pointer_is_pinned(out_rgb);
    if (bands > 1 && !pinned && ctx->host_stage_bytes < bytes) {
        if (ctx->host_stage) (void)hipHostFree(ctx->host_stage);
        ctx->host_stage = nullptr;
        ctx->host_stage_bytes = 0;
        if (hipHostMalloc(&ctx->host_stage, bytes, hipHostMallocDefault) == hipSuccess && ctx->host_stage) ctx->host_stage_bytes = bytes;
        else {
            (void)hipGetLastError();
            ctx->host_stage = nullptr;
            bands = 1;  // no page-locked memory to be had: the runtime's staged copy
        }
    }
    RT_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    // (a lambda, so that a failure half-way leaves through one place: below, behind a drain of everything enqueued)
    const auto render_and_copy = [&]() -> int {
    if (bands <= 1) {
        RT_TRY(rthip_cast_device(ctx, cast, seed, (float*)ctx->frame.p, width, height, nullptr, ctx->stream));
        RT_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
        RT_HIP(ctx, hipMemcpyAsync(out_rgb, ctx->frame.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    } else {
        const uint32_t rows = (((uint32_t)height + bands - 1u) / bands + 63u) & ~63u;  // shard_band_rows
        for (uint32_t k = 0; k < bands; k++) {
            if (!ctx->band_stream[k]) RT_HIP(ctx, hipStreamCreateWithFlags(&ctx->band_stream[k], hipStreamNonBlocking));
            if (!ctx->band_copy_stream[k]) RT_HIP(ctx, hipStreamCreateWithFlags(&ctx->band_copy_stream[k], hipStreamNonBlocking));
            if (!ctx->band_done[k]) RT_HIP(ctx, hipEventCreateWithFlags(&ctx->band_done[k], hipEventDisableTiming));
        }
        // all the launches first, then the copies on streams of their own: a copy waits for its band only, and no launch
        // queues up behind a copy
        uint32_t used = 0;
        for (uint32_t k = 0; k < bands && k * rows < (uint32_t)height; k++, used++) {
            const rthip_shard band = {3u, k, bands};
            hipStream_t bs = ctx->band_stream[k];
            RT_HIP(ctx, hipStreamWaitEvent(bs, ctx->ev0, 0));  // behind whatever the context's stream held before this call
            RT_TRY(rthip_cast_device(ctx, cast, seed, (float*)ctx->frame.p, width, height, &band, bs));
            RT_HIP(ctx, hipEventRecord(ctx->band_done[k], bs));
            RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->band_done[k], 0));
        }
        float* landing = pinned ? out_rgb : (float*)ctx->host_stage;
        constexpr uint32_t kChunks = rthip_ctx::kBandChunks;
        const uint32_t chunks = pinned ? 1u : kChunks;  // into staging a band travels in pieces, so that the host threads can start on it
        size_t off[rthip_ctx::kMaxBands], count[rthip_ctx::kMaxBands], piece[rthip_ctx::kMaxBands];
        for (uint32_t k = 0; k < used; k++) {
            const uint32_t y0 = k * rows, y1 = (k + 1u) * rows < (uint32_t)height ? (k + 1u) * rows : (uint32_t)height;
            off[k] = (size_t)y0 * width * 3;
            count[k] = (size_t)(y1 - y0) * width * 3;
            piece[k] = ((count[k] + chunks - 1) / chunks + 1023) & ~(size_t)1023;
            RT_HIP(ctx, hipStreamWaitEvent(ctx->band_copy_stream[k], ctx->band_done[k], 0));
            for (uint32_t c = 0; c < chunks; c++) {
                const size_t first = (size_t)c * piece[k];
                const size_t n = first >= count[k] ? 0 : (count[k] - first < piece[k] ? count[k] - first : piece[k]);
                if (n) RT_HIP(ctx, hipMemcpyAsync(landing + off[k] + first, (const float*)ctx->frame.p + off[k] + first, n * sizeof(float), hipMemcpyDeviceToHost, ctx->band_copy_stream[k]));
                if (!pinned) {
                    hipEvent_t& e = ctx->chunk_done[k * kChunks + c];
                    if (!e) RT_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
                    RT_HIP(ctx, hipEventRecord(e, ctx->band_copy_stream[k]));
                }
            }
        }
        RT_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));  // every band rendered: the frame is whole in device memory
        if (pinned) {
            for (uint32_t k = 0; k < used; k++) RT_HIP(ctx, hipStreamSynchronize(ctx->band_copy_stream[k]));
        } else {
            // two host threads per band take its pieces in turn: wait for the piece's DMA, then move it into the caller's
            // buffer (measured at 1920x1080 with four bands, whole bands per thread: 1.02 ms per call with one thread per
            // band, 0.95 with two, 0.93 with four)
            constexpr uint32_t per_band = 2u;
            if (!ctx->movers) {
                ctx->movers = new HostPool;
                ctx->movers->start((uint32_t)ctx->cast_bands * per_band, ctx->device);
            }
            hipError_t status[rthip_ctx::kMaxBands * per_band];
            for (uint32_t t = 0; t < used * per_band; t++) status[t] = hipSuccess;
            ctx->movers->run(used * per_band, [&](uint32_t t) {
                const uint32_t k = t / per_band;
                for (uint32_t c = t % per_band; c < kChunks && status[t] == hipSuccess; c += per_band) {
                    const size_t first = (size_t)c * piece[k];
                    const size_t n = first >= count[k] ? 0 : (count[k] - first < piece[k] ? count[k] - first : piece[k]);
                    status[t] = hipEventSynchronize(ctx->chunk_done[k * kChunks + c]);
                    if (status[t] == hipSuccess && n) memcpy(out_rgb + off[k] + first, landing + off[k] + first, n * sizeof(float));
                }
            });
            for (uint32_t t = 0; t < used * per_band; t++) RT_HIP(ctx, status[t]);
        }
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return RTHIP_OK;
    };
    const int rc = render_and_copy();
    if (rc != RTHIP_OK) {
        // Band launches and copies into the caller's buffer or the staging may still be in flight: the caller is free to
        // release out_rgb once this call has returned, so nothing may be writing to it any more.  (The message of the
        // failure is already in the context; what the drains return is not reported on top of it.)
        for (uint32_t k = 0; k < rthip_ctx::kMaxBands; k++) {
            if (ctx->band_stream[k]) (void)hipStreamSynchronize(ctx->band_stream[k]);
            if (ctx->band_copy_stream[k]) (void)hipStreamSynchronize(ctx->band_copy_stream[k]);
        }
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipGetLastError();
        return rc;
    }
    RT_HIP(ctx, hipEventElapsedTime(&ctx->last_cast_ms, ctx->ev0, ctx->ev1));
    ctx->frame_w = width;  // the image stays in device memory for rthip_hdr_encode(rgb = NULL)
    ctx->frame_h = height;
    return RTHIP_OK;
}

// ---- ray records -----------------------------------------------------------------------

int rthip_cast_ray_records(rthip_ctx* ctx, const rthip_cast_settings* cast, uint64_t seed, int width, int height,
                           rthip_ray_record* out_records, uint64_t capacity, uint64_t* out_count) {
    RT_TRY(check_ready(ctx, "rthip_cast_ray_records"));
    RT_TRY(check_cast(ctx, cast, width, height));
    if (!out_count) return rthip_fail(ctx, RTHIP_E_INVALID, "out_count is NULL");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t n_paths = (uint64_t)width * height * cast->samples * cast->samples;
    const uint64_t n_slots = n_paths * (ctx->max_bounces ? ctx->max_bounces : 1);
    if (n_paths >= (1ull << 31) || n_slots * sizeof(rthip_ray_record) > (8ull << 30))
        return rthip_fail(ctx, RTHIP_E_LIMIT, "ray records of %llu paths x %u bounces exceed 8 GiB: use a smaller frame",
                          (unsigned long long)n_paths, ctx->max_bounces);
    *out_count = 0;
    if (ctx->max_bounces == 0) return RTHIP_OK;
    RT_TRY(rthip_reserve(ctx, ctx->scratch_a, n_slots * sizeof(rthip_ray_record)));
    RT_TRY(rthip_reserve(ctx, ctx->scratch_b, n_paths * 4 + 16));
    const DevScene scene = make_scene(ctx);
    uint64_t total = 0;
    // first without an output buffer: the number of records decides the size of the compacted array
    RT_TRY(cast_records_launch(ctx, scene, *cast, seed, width, height, (rthip_ray_record*)ctx->scratch_a.p,
                               (uint32_t*)ctx->scratch_b.p, nullptr, &total, ctx->stream));
    *out_count = total;
    if (!out_records) return RTHIP_OK;
    if (total > capacity)
        return rthip_fail(ctx, RTHIP_E_LIMIT, "the frame has %llu rays, the buffer holds %llu records",
                          (unsigned long long)total, (unsigned long long)capacity);
    if (total == 0) return RTHIP_OK;
    RT_TRY(rthip_reserve(ctx, ctx->scratch_c, total * sizeof(rthip_ray_record)));
    RT_TRY(cast_records_launch(ctx, scene, *cast, seed, width, height, (rthip_ray_record*)ctx->scratch_a.p,
                               (uint32_t*)ctx->scratch_b.p, (rthip_ray_record*)ctx->scratch_c.p, &total, ctx->stream));
    RT_HIP(ctx, hipMemcpyAsync(out_records, ctx->scratch_c.p, total * sizeof(rthip_ray_record), hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RTHIP_OK;
}

// ---- .hdr output -------------------------------------------------------------------

uint64_t rthip_hdr_bound(int width, int height) { return hdr_bound(width, height); }

int rthip_hdr_encode_device(rthip_ctx* ctx, const float* d_rgb, int width, int height, uint8_t* d_out, uint64_t capacity,
                            uint64_t* d_size, void* stream) {
    if (!ctx) return RTHIP_E_INVALID;
    if (!d_rgb || !d_out || !d_size) return rthip_fail(ctx, RTHIP_E_INVALID, "NULL device pointer");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    return hdr_encode_device(ctx, d_rgb, width, height, d_out, capacity, d_size, (hipStream_t)stream);
}

int rthip_hdr_encode(rthip_ctx* ctx, const float* rgb, int width, int height, uint8_t* out, uint64_t capacity,
                     uint64_t* out_size) {
    if (!ctx) return RTHIP_E_INVALID;
    if (!out || !out_size) return rthip_fail(ctx, RTHIP_E_INVALID, "out / out_size is NULL");
    if (width <= 0 || height <= 0) return rthip_fail(ctx, RTHIP_E_INVALID, "image %d x %d", width, height);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const float* d_rgb = nullptr;
    if (rgb) {
        const size_t bytes = (size_t)width * height * 3 * sizeof(float);
        RT_TRY(rthip_reserve(ctx, ctx->scratch_a, bytes));
        RT_HIP(ctx, hipMemcpyAsync(ctx->scratch_a.p, rgb, bytes, hipMemcpyHostToDevice, ctx->stream));
        d_rgb = (const float*)ctx->scratch_a.p;
    } else {
        if (ctx->frame_w != width || ctx->frame_h != height)
            return rthip_fail(ctx, RTHIP_E_INVALID, "rgb is NULL but the last rthip_cast produced %d x %d, not %d x %d",
                              ctx->frame_w, ctx->frame_h, width, height);
        d_rgb = (const float*)ctx->frame.p;
    }
    const uint64_t bound = hdr_bound(width, height);
    RT_TRY(rthip_reserve(ctx, ctx->hdr_out, bound));
    RT_TRY(rthip_reserve(ctx, ctx->hdr_size, 16));
    RT_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    RT_TRY(hdr_encode_device(ctx, d_rgb, width, height, (uint8_t*)ctx->hdr_out.p, bound, (uint64_t*)ctx->hdr_size.p,
                             ctx->stream));
    RT_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    uint64_t size = 0;
    RT_HIP(ctx, hipMemcpyAsync(&size, ctx->hdr_size.p, sizeof size, hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    RT_HIP(ctx, hipEventElapsedTime(&ctx->last_encode_ms, ctx->ev0, ctx->ev1));
    *out_size = size;
    if (size > capacity)
        return rthip_fail(ctx, RTHIP_E_LIMIT, "the encoded image has %llu bytes, the buffer %llu",
                          (unsigned long long)size, (unsigned long long)capacity);
    RT_HIP(ctx, hipMemcpy(out, ctx->hdr_out.p, size, hipMemcpyDeviceToHost));
    return RTHIP_OK;
}

// ---- taps ------------------------------------------------------------------------------

int rthip_lbvh_build_device(rthip_ctx* ctx, const float* d_aabb6, uint64_t n, rthip_node* d_nodes, void* stream) {
    if (!ctx) return RTHIP_E_INVALID;
    if (!d_aabb6 || !d_nodes) return rthip_fail(ctx, RTHIP_E_INVALID, "NULL device pointer");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    return lbvh_build_device(ctx, d_aabb6, n, (float4*)d_nodes, (hipStream_t)stream, nullptr, nullptr, nullptr);
}

int rthip_lbvh_reserve(rthip_ctx* ctx, uint64_t n) {
    if (!ctx) return RTHIP_E_INVALID;
    if (n == 0 || n > (1ull << 30)) return rthip_fail(ctx, RTHIP_E_LIMIT, "box count %llu outside 1..2^30", (unsigned long long)n);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_TRY(rthip_reserve(ctx, ctx->ws.aabb, n * 24));
    return lbvh_reserve_workspace(ctx, n);
}

int rthip_lbvh_build_device_noalloc(rthip_ctx* ctx, const float* d_aabb6, uint64_t n, rthip_node* d_nodes, void* stream) {
    if (!ctx) return RTHIP_E_INVALID;
    if (!d_aabb6 || !d_nodes) return rthip_fail(ctx, RTHIP_E_INVALID, "NULL device pointer");
    if (n == 0) return rthip_fail(ctx, RTHIP_E_INVALID, "lbvh build needs at least one box");
    if (n > kStaticBuildMaxBoxes)
        return rthip_fail(ctx, RTHIP_E_LIMIT, "rthip_lbvh_build_device_noalloc builds at most %llu boxes (got %llu): its workspace is a fixed array",
                          (unsigned long long)kStaticBuildMaxBoxes, (unsigned long long)n);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_TRY(lbvh_static_workspace_init(ctx));
    return lbvh_build_device(ctx, d_aabb6, n, (float4*)d_nodes, (hipStream_t)stream, nullptr, nullptr, nullptr, false,
                             &ctx->ws_static);
}

int rthip_mesh_build_device(rthip_ctx* ctx, const float* d_px, const float* d_py, const float* d_pz,
                            uint32_t vertex_count, const uint32_t* d_indices, uint64_t n_tris, rthip_node* d_nodes,
                            void* stream) {
    if (!ctx) return RTHIP_E_INVALID;
    if (!d_px || !d_py || !d_pz || !d_nodes) return rthip_fail(ctx, RTHIP_E_INVALID, "NULL device pointer");
    if (n_tris == 0 || n_tris > (1ull << 30)) return rthip_fail(ctx, RTHIP_E_LIMIT, "triangle count %llu outside 1..2^30", (unsigned long long)n_tris);
    if (!d_indices && 3 * n_tris > vertex_count) return rthip_fail(ctx, RTHIP_E_INVALID, "auto-indexed mesh needs 3 vertices per triangle");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_TRY(rthip_reserve(ctx, ctx->ws.aabb, n_tris * 24));
    RT_TRY(mesh_prepare_device(ctx, d_px, d_py, d_pz, nullptr, nullptr, nullptr, d_indices, n_tris, (float*)ctx->ws.aabb.p,
                               nullptr, nullptr, (hipStream_t)stream));
    return lbvh_build_device(ctx, (const float*)ctx->ws.aabb.p, n_tris, (float4*)d_nodes, (hipStream_t)stream, nullptr,
                             nullptr, nullptr);
}

int rthip_lbvh_build(rthip_ctx* ctx, const float* aabb6, uint64_t n, uint64_t* out_codes, uint64_t* out_codes_sorted,
                     uint32_t* out_ids_sorted, rthip_node* out_nodes, float* out_ms) {
    if (!ctx) return RTHIP_E_INVALID;
    if (!aabb6 || n == 0) return rthip_fail(ctx, RTHIP_E_INVALID, "rthip_lbvh_build needs at least one box");
    if (n > (1ull << 30)) return rthip_fail(ctx, RTHIP_E_LIMIT, "lbvh build is limited to 2^30 boxes");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_TRY(rthip_reserve(ctx, ctx->scratch_a, n * 24));
    RT_TRY(rthip_reserve(ctx, ctx->scratch_b, (2 * n - 1) * 32));
    RT_TRY(rthip_reserve(ctx, ctx->scratch_c, n * 8));
    RT_TRY(rthip_reserve(ctx, ctx->scratch_d, n * 8));
    RT_TRY(rthip_reserve(ctx, ctx->scratch_e, n * 4));
    hipStream_t s = ctx->stream;
    RT_HIP(ctx, hipMemcpyAsync(ctx->scratch_a.p, aabb6, n * 24, hipMemcpyHostToDevice, s));
    RT_HIP(ctx, hipEventRecord(ctx->ev0, s));
    RT_TRY(lbvh_build_device(ctx, (const float*)ctx->scratch_a.p, n, (float4*)ctx->scratch_b.p, s,
                             out_codes ? (uint64_t*)ctx->scratch_c.p : nullptr,
                             out_codes_sorted ? (uint64_t*)ctx->scratch_d.p : nullptr,
                             out_ids_sorted ? (uint32_t*)ctx->scratch_e.p : nullptr));
    RT_HIP(ctx, hipEventRecord(ctx->ev1, s));
    if (out_codes) RT_HIP(ctx, hipMemcpyAsync(out_codes, ctx->scratch_c.p, n * 8, hipMemcpyDeviceToHost, s));
    if (out_codes_sorted) RT_HIP(ctx, hipMemcpyAsync(out_codes_sorted, ctx->scratch_d.p, n * 8, hipMemcpyDeviceToHost, s));
    if (out_ids_sorted) RT_HIP(ctx, hipMemcpyAsync(out_ids_sorted, ctx->scratch_e.p, n * 4, hipMemcpyDeviceToHost, s));
    if (out_nodes) RT_HIP(ctx, hipMemcpyAsync(out_nodes, ctx->scratch_b.p, (2 * n - 1) * 32, hipMemcpyDeviceToHost, s));
    RT_HIP(ctx, hipStreamSynchronize(s));
    RT_HIP(ctx, hipEventElapsedTime(&ctx->last_build_ms, ctx->ev0, ctx->ev1));
    if (out_ms) *out_ms = ctx->last_build_ms;
    if (out_ids_sorted)
        for (uint64_t i = 0; i < n; i++) out_ids_sorted[i] += 1;  // NodeIndex.id = idx + 1, lbvh.c:116
    return RTHIP_OK;
}

static int unpack_hits(rthip_ctx* ctx, const void* d_hits4, uint64_t n, uint32_t* out_inst, uint32_t* out_tri, float* out_t,
                       hipStream_t s) {
    std::vector<uint32_t> h(4 * n);
    RT_HIP(ctx, hipMemcpyAsync(h.data(), d_hits4, 16 * n, hipMemcpyDeviceToHost, s));
    RT_HIP(ctx, hipStreamSynchronize(s));
    for (uint64_t i = 0; i < n; i++) {
        if (out_inst) out_inst[i] = h[4 * i];
        if (out_tri) out_tri[i] = h[4 * i + 1];
        if (out_t) memcpy(&out_t[i], &h[4 * i + 2], 4);
    }
    return RTHIP_OK;
}

int rthip_primary_hits(rthip_ctx* ctx, const rthip_cast_settings* cast, uint32_t* out_instance_id, uint32_t* out_tri_idx,
                       float* out_t, int width, int height) {
    RT_TRY(check_ready(ctx, "rthip_primary_hits"));
    RT_TRY(check_cast(ctx, cast, width, height));
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t n = (uint64_t)width * height;
    RT_TRY(rthip_reserve(ctx, ctx->scratch_a, n * 16));
    RT_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    RT_TRY(primary_hits_launch(ctx, make_scene(ctx), *cast, width, height, (uint32_t*)ctx->scratch_a.p,
                               counters_or_null(ctx), ctx->stream));
    RT_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    RT_TRY(unpack_hits(ctx, ctx->scratch_a.p, n, out_instance_id, out_tri_idx, out_t, ctx->stream));
    RT_HIP(ctx, hipEventElapsedTime(&ctx->last_cast_ms, ctx->ev0, ctx->ev1));
    return RTHIP_OK;
}

int rthip_trace_rays_device(rthip_ctx* ctx, const float* d_origins, const float* d_directions, uint64_t n,
                            uint32_t* d_hits, void* stream) {
    RT_TRY(check_ready(ctx, "rthip_trace_rays_device"));
    if (n && (!d_origins || !d_directions || !d_hits)) return rthip_fail(ctx, RTHIP_E_INVALID, "NULL device pointer");
    if (n >= (1ull << 32)) return rthip_fail(ctx, RTHIP_E_LIMIT, "at most 2^32 - 1 rays in one call (got %llu)", (unsigned long long)n);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    return trace_rays_launch(ctx, make_scene(ctx), d_origins, d_directions, n, d_hits, nullptr, nullptr,
                             counters_or_null(ctx), (hipStream_t)stream);
}

int rthip_trace_rays(rthip_ctx* ctx, const float* origins, const float* directions, uint64_t n, uint32_t* out_instance_id,
                     uint32_t* out_tri_idx, float* out_t, float* out_uv, float* out_normal) {
    RT_TRY(check_ready(ctx, "rthip_trace_rays"));
    if (n == 0) return RTHIP_OK;
    if (!origins || !directions) return rthip_fail(ctx, RTHIP_E_INVALID, "ray arrays are NULL");
    if (n >= (1ull << 32)) return rthip_fail(ctx, RTHIP_E_LIMIT, "at most 2^32 - 1 rays in one call (got %llu)", (unsigned long long)n);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    RT_TRY(rthip_reserve(ctx, ctx->scratch_a, n * 16));
    RT_TRY(rthip_reserve(ctx, ctx->scratch_b, n * 12));
    RT_TRY(rthip_reserve(ctx, ctx->scratch_c, n * 12));
    if (out_uv) RT_TRY(rthip_reserve(ctx, ctx->scratch_d, n * 8));
    if (out_normal) RT_TRY(rthip_reserve(ctx, ctx->scratch_e, n * 12));
    RT_HIP(ctx, hipMemcpyAsync(ctx->scratch_b.p, origins, n * 12, hipMemcpyHostToDevice, s));
    RT_HIP(ctx, hipMemcpyAsync(ctx->scratch_c.p, directions, n * 12, hipMemcpyHostToDevice, s));
    RT_HIP(ctx, hipEventRecord(ctx->ev0, s));
    RT_TRY(trace_rays_launch(ctx, make_scene(ctx), (const float*)ctx->scratch_b.p, (const float*)ctx->scratch_c.p, n,
                             (uint32_t*)ctx->scratch_a.p, out_uv ? (float*)ctx->scratch_d.p : nullptr,
                             out_normal ? (float*)ctx->scratch_e.p : nullptr, counters_or_null(ctx), s));
    RT_HIP(ctx, hipEventRecord(ctx->ev1, s));
    if (out_uv) RT_HIP(ctx, hipMemcpyAsync(out_uv, ctx->scratch_d.p, n * 8, hipMemcpyDeviceToHost, s));
    if (out_normal) RT_HIP(ctx, hipMemcpyAsync(out_normal, ctx->scratch_e.p, n * 12, hipMemcpyDeviceToHost, s));
    RT_TRY(unpack_hits(ctx, ctx->scratch_a.p, n, out_instance_id, out_tri_idx, out_t, s));
    RT_HIP(ctx, hipEventElapsedTime(&ctx->last_cast_ms, ctx->ev0, ctx->ev1));
    return RTHIP_OK;
}

int rthip_occluded_device(rthip_ctx* ctx, const float* d_origins, const float* d_directions, uint64_t n, float tmax,
                          uint8_t* d_out, void* stream) {
    RT_TRY(check_ready(ctx, "rthip_occluded_device"));
    if (n && (!d_origins || !d_directions || !d_out)) return rthip_fail(ctx, RTHIP_E_INVALID, "NULL device pointer");
    if (n >= (1ull << 32)) return rthip_fail(ctx, RTHIP_E_LIMIT, "at most 2^32 - 1 rays in one call (got %llu)", (unsigned long long)n);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    return occlusion_launch(ctx, make_scene(ctx), d_origins, d_directions, n, tmax, d_out, counters_or_null(ctx),
                            (hipStream_t)stream);
}

int rthip_occluded(rthip_ctx* ctx, const float* origins, const float* directions, uint64_t n, float tmax, uint8_t* out) {
    RT_TRY(check_ready(ctx, "rthip_occluded"));
    if (n == 0) return RTHIP_OK;
    if (!origins || !directions || !out) return rthip_fail(ctx, RTHIP_E_INVALID, "ray or output arrays are NULL");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    RT_TRY(rthip_reserve(ctx, ctx->scratch_a, n));
    RT_TRY(rthip_reserve(ctx, ctx->scratch_b, n * 12));
    RT_TRY(rthip_reserve(ctx, ctx->scratch_c, n * 12));
    RT_HIP(ctx, hipMemcpyAsync(ctx->scratch_b.p, origins, n * 12, hipMemcpyHostToDevice, s));
    RT_HIP(ctx, hipMemcpyAsync(ctx->scratch_c.p, directions, n * 12, hipMemcpyHostToDevice, s));
    RT_HIP(ctx, hipEventRecord(ctx->ev0, s));
    RT_TRY(rthip_occluded_device(ctx, (const float*)ctx->scratch_b.p, (const float*)ctx->scratch_c.p, n, tmax,
                                 (uint8_t*)ctx->scratch_a.p, s));
    RT_HIP(ctx, hipEventRecord(ctx->ev1, s));
    RT_HIP(ctx, hipMemcpyAsync(out, ctx->scratch_a.p, n, hipMemcpyDeviceToHost, s));
    RT_HIP(ctx, hipStreamSynchronize(s));
    RT_HIP(ctx, hipEventElapsedTime(&ctx->last_cast_ms, ctx->ev0, ctx->ev1));
    return RTHIP_OK;
}

int rthip_get_stats(rthip_ctx* ctx, rthip_stats* out) {
    if (!ctx || !out) return RTHIP_E_INVALID;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipDeviceSynchronize());
    uint64_t c[8];
    RT_HIP(ctx, hipMemcpy(c, ctx->counters.p, 64, hipMemcpyDeviceToHost));
    RT_HIP(ctx, hipMemset(ctx->counters.p, 0, 64));
    out->rays = c[0];
    out->nodes_tested = c[1];
    out->tris_tested = c[2];
    out->inst_entered = c[3];
    out->wave_steps = c[4];
    out->lane_steps = c[5];
    out->last_cast_ms = ctx->last_cast_ms;
    out->last_build_ms = ctx->last_build_ms;
    out->last_cast_kernel = ctx->last_cast_kernel;
    out->last_cast_refill = ctx->last_cast_refill;
    out->last_encode_ms = ctx->last_encode_ms;
    out->reserved = 0;
    return RTHIP_OK;
}

int rthip_get_tree(rthip_ctx* ctx, int which, rthip_node* out_nodes, uint64_t* out_count) {
    if (!ctx) return RTHIP_E_INVALID;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const void* src = nullptr;
    uint64_t n = 0;
    if (which < 0) {
        if (!ctx->tlas_built) return rthip_fail(ctx, RTHIP_E_INVALID, "no TLAS has been built");
        src = ctx->tlas_nodes.p;
        n = ctx->n_inst ? 2 * (uint64_t)ctx->n_inst - 1 : 0;
    } else {
        if (!ctx->blas_committed || (size_t)which >= ctx->meshes.size())
            return rthip_fail(ctx, RTHIP_E_INVALID, "blas_id %d is not built", which);
        const DevMesh& m = ctx->meshes[which].dev;
        src = (const float4*)ctx->blas_nodes.p + 2 * (size_t)m.node_base;
        n = m.n_nodes;
    }
    if (out_count) *out_count = n;
    if (out_nodes && n) {
        RT_HIP(ctx, hipDeviceSynchronize());
        RT_HIP(ctx, hipMemcpy(out_nodes, src, n * 32, hipMemcpyDeviceToHost));
        // the context keeps (min.x, max.x, min.y, max.y | min.z, max.z, skip, id); callers get rthip_node
        for (uint64_t k = 0; k < n; k++) {
            uint32_t w[8];
            memcpy(w, &out_nodes[k], 32);
            const uint32_t r[8] = {w[0], w[2], w[4], w[6], w[1], w[3], w[5], w[7]};  // min.xyz, skip, max.xyz, id
            memcpy(&out_nodes[k], r, 32);
        }
    }
    return RTHIP_OK;
}

}  // extern "C"
