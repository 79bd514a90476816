// Several GPUs of one node behind one handle: the scene is replicated (every device builds its own BVHs from the same
// host data), a frame is split by 64x64 tiles or by sample index (rthip_shard), and the shares are brought together with
// peer copies over xGMI: in tile mode every device ships only its own tiles (1/N of the frame), packed; in sample mode
// the devices' partial sums are added on the first device in rank order, so the result does not depend on timing.
// One host thread per device drives its context; nothing here computes on the host except placing tile rows into the
// caller's buffer.

#include <string.h>

#include <string>
#include <thread>
#include <vector>

#include "rthip_internal.h"

namespace {

constexpr int kBlock = 256;
constexpr uint32_t kTile = 64;                       // rthip_shard's tile edge
constexpr uint32_t kTileFloats = kTile * kTile * 3;  // one packed tile: 48 KiB

__device__ __forceinline__ bool tile_pixel(int w, int h, uint32_t rank, uint32_t world, uint32_t& slot_pixel, uint32_t& pixel) {
    const uint32_t tiles_x = ((uint32_t)w + kTile - 1) / kTile;
    const uint32_t slot = blockIdx.x / (kTile * kTile / kBlock);
    const uint32_t p = (blockIdx.x % (kTile * kTile / kBlock)) * kBlock + threadIdx.x;
    const uint32_t tile = rank + slot * world;
    const uint32_t x = (tile % tiles_x) * kTile + (p % kTile), y = (tile / tiles_x) * kTile + (p / kTile);
    slot_pixel = slot * kTile * kTile + p;
    pixel = y * (uint32_t)w + x;
    return x < (uint32_t)w && y < (uint32_t)h;
}

// this rank's tiles (tile % world == rank, row-major tile numbering) one after the other, 64 x 64 x 3 floats each
__global__ __launch_bounds__(kBlock) void tiles_pack_kernel(const float* __restrict__ rgb, int w, int h, uint32_t rank, uint32_t world,
                                                            float* __restrict__ packed) {
    uint32_t sp, px;
    const bool inside = tile_pixel(w, h, rank, world, sp, px);
    float* o = packed + 3 * (size_t)sp;
    const float* s = rgb + 3 * (size_t)px;
    o[0] = inside ? s[0] : 0.f;
    o[1] = inside ? s[1] : 0.f;
    o[2] = inside ? s[2] : 0.f;
}

__global__ __launch_bounds__(kBlock) void tiles_unpack_kernel(const float* __restrict__ packed, int w, int h, uint32_t rank,
                                                              uint32_t world, float* __restrict__ rgb) {
    uint32_t sp, px;
    if (!tile_pixel(w, h, rank, world, sp, px)) return;
    const float* s = packed + 3 * (size_t)sp;
    float* o = rgb + 3 * (size_t)px;
    o[0] = s[0];
    o[1] = s[1];
    o[2] = s[2];
}

// out = (part[0] + part[1] + ... + part[n-1]) * scale, added in that order
__global__ __launch_bounds__(kBlock) void parts_sum_kernel(const float* __restrict__ parts, uint32_t n_parts, uint64_t floats,
                                                           float scale, float* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= floats) return;
    float acc = parts[i];
    for (uint32_t k = 1; k < n_parts; k++) acc = acc + parts[(uint64_t)k * floats + i];
    out[i] = acc * scale;
}

uint32_t tiles_of_rank(int w, int h, uint32_t rank, uint32_t world) {
    const uint32_t tiles = (((uint32_t)w + kTile - 1) / kTile) * (((uint32_t)h + kTile - 1) / kTile);
    return tiles > rank ? (tiles - rank + world - 1) / world : 0u;
}

}  // namespace

struct rthip_group {
    std::vector<rthip_ctx*> ctx;
    std::vector<DevBuf> frame;      // per device: its share of the frame (full size)
    std::vector<DevBuf> packed;     // per device: its tiles, packed
    std::vector<void*> host_stage;  // per device: page-locked staging for rthip_group_cast
    std::vector<size_t> host_stage_bytes;
    DevBuf gather;                  // on device 0: every rank's packed tiles / partial frames
    std::string err;
};

namespace {

int group_fail(rthip_group* g, int code, const std::string& msg) {
    g->err = msg;
    return code;
}

// fn(k) on one host thread per device; the first failure is reported with the device's message
template <class F>
int for_each_device(rthip_group* g, F fn) {
    const size_t n = g->ctx.size();
    std::vector<int> rc(n, RTHIP_OK);
    std::vector<std::thread> th;
    for (size_t k = 1; k < n; k++) th.emplace_back([&, k] { rc[k] = fn((int)k); });
    rc[0] = fn(0);
    for (auto& t : th) t.join();
    for (size_t k = 0; k < n; k++)
        if (rc[k] != RTHIP_OK) return group_fail(g, rc[k], "device " + std::to_string(g->ctx[k]->device) + " (rank " + std::to_string(k) + "): " + g->ctx[k]->err);
    return RTHIP_OK;
}

int check_group_cast(rthip_group* g, const rthip_cast_settings* cast, int w, int h, int mode) {
    if (!cast) return group_fail(g, RTHIP_E_INVALID, "cast settings are NULL");
    if (w <= 0 || h <= 0) return group_fail(g, RTHIP_E_INVALID, "image size is not positive");
    if (mode != 1 && mode != 2) return group_fail(g, RTHIP_E_INVALID, "shard mode must be 1 (tiles) or 2 (samples)");
    return RTHIP_OK;
}

// rank k renders its share into its own frame buffer and, in tile mode, packs its tiles; enqueued on the context's stream
int render_share(rthip_group* g, int k, const rthip_cast_settings* cast, uint64_t seed, int w, int h, int mode) {
    rthip_ctx* c = g->ctx[k];
    const uint32_t world = (uint32_t)g->ctx.size();
    RT_HIP(c, hipSetDevice(c->device));
    RT_TRY(rthip_reserve(c, g->frame[k], (size_t)w * h * 12));
    const rthip_shard sh = {(uint32_t)mode, (uint32_t)k, world};
    RT_TRY(rthip_cast_device(c, cast, seed, (float*)g->frame[k].p, w, h, world > 1 ? &sh : nullptr, c->stream));
    if (mode == 1) {
        const uint32_t slots = tiles_of_rank(w, h, (uint32_t)k, world);
        RT_TRY(rthip_reserve(c, g->packed[k], (size_t)(slots ? slots : 1) * kTileFloats * 4));
        if (slots) {
            tiles_pack_kernel<<<slots * (kTile * kTile / kBlock), kBlock, 0, c->stream>>>((const float*)g->frame[k].p, w, h, (uint32_t)k,
                                                                                        world, (float*)g->packed[k].p);
            RT_HIP(c, hipGetLastError());
        }
    }
    return RTHIP_OK;
}

}  // namespace

extern "C" {

int rthip_group_create(rthip_group** out, int n, const int* device_ids) {
    if (!out) return RTHIP_E_INVALID;
    *out = nullptr;
    if (n <= 0 || n > 64 || !device_ids) return RTHIP_E_INVALID;
    rthip_group* g = new rthip_group();
    for (int k = 0; k < n; k++) {
        rthip_ctx* c = nullptr;
        const int rc = rthip_create(&c, device_ids[k]);
        if (rc != RTHIP_OK) {
            for (rthip_ctx* made : g->ctx) rthip_destroy(made);
            delete g;
            return rc;
        }
        g->ctx.push_back(c);
    }
    // peer access from every device to the first one's memory (where frames are gathered); a refusal only means the
    // copies are staged by the runtime
    for (int k = 1; k < n; k++)
        if (g->ctx[k]->device != g->ctx[0]->device && hipSetDevice(g->ctx[k]->device) == hipSuccess) {
            (void)hipDeviceEnablePeerAccess(g->ctx[0]->device, 0);
            (void)hipGetLastError();
        }
    g->frame.resize(n);
    g->packed.resize(n);
    g->host_stage.assign(n, nullptr);
    g->host_stage_bytes.assign(n, 0);
    *out = g;
    return RTHIP_OK;
}

void rthip_group_destroy(rthip_group* g) {
    if (!g) return;
    for (size_t k = 0; k < g->ctx.size(); k++) {
        (void)hipSetDevice(g->ctx[k]->device);
        (void)hipDeviceSynchronize();
        rthip_release(g->frame[k]);
        rthip_release(g->packed[k]);
        if (g->host_stage[k]) (void)hipHostFree(g->host_stage[k]);
    }
    if (!g->ctx.empty()) {
        (void)hipSetDevice(g->ctx[0]->device);
        rthip_release(g->gather);
    }
    for (rthip_ctx* c : g->ctx) rthip_destroy(c);
    delete g;
}

int rthip_group_size(const rthip_group* g) { return g ? (int)g->ctx.size() : 0; }
rthip_ctx* rthip_group_ctx(rthip_group* g, int rank) { return (g && rank >= 0 && (size_t)rank < g->ctx.size()) ? g->ctx[rank] : nullptr; }
const char* rthip_group_last_error(const rthip_group* g) { return g ? g->err.c_str() : "null group"; }

int rthip_group_set_tracer(rthip_group* g, uint32_t max_bounces, int sky) {
    if (!g) return RTHIP_E_INVALID;
    return for_each_device(g, [&](int k) { return rthip_set_tracer(g->ctx[k], max_bounces, sky); });
}
int rthip_group_set_morton_bounds(rthip_group* g, int mode) {
    if (!g) return RTHIP_E_INVALID;
    return for_each_device(g, [&](int k) { return rthip_set_morton_bounds(g->ctx[k], mode); });
}
int rthip_group_set_traversal(rthip_group* g, int mode) {
    if (!g) return RTHIP_E_INVALID;
    return for_each_device(g, [&](int k) { return rthip_set_traversal(g->ctx[k], mode); });
}
int rthip_group_blas_begin(rthip_group* g, uint32_t n_meshes) {
    if (!g) return RTHIP_E_INVALID;
    return for_each_device(g, [&](int k) { return rthip_blas_begin(g->ctx[k], n_meshes); });
}
int rthip_group_blas_add_mesh(rthip_group* g, uint32_t blas_id, const void* verts, uint32_t vertex_count, uint32_t stride_bytes,
                              uint32_t pos_off, int32_t nrm_off, const uint32_t* idx, uint32_t index_count) {
    if (!g) return RTHIP_E_INVALID;
    return for_each_device(g, [&](int k) {
        return rthip_blas_add_mesh(g->ctx[k], blas_id, verts, vertex_count, stride_bytes, pos_off, nrm_off, idx, index_count);
    });
}
int rthip_group_blas_commit(rthip_group* g) {
    if (!g) return RTHIP_E_INVALID;
    return for_each_device(g, [&](int k) { return rthip_blas_commit(g->ctx[k]); });
}
int rthip_group_tlas_build(rthip_group* g, const rthip_instance* instances, uint32_t n_instances, const rthip_material* materials,
                           uint32_t n_materials) {
    if (!g) return RTHIP_E_INVALID;
    return for_each_device(g, [&](int k) { return rthip_tlas_build(g->ctx[k], instances, n_instances, materials, n_materials); });
}

int rthip_group_cast(rthip_group* g, const rthip_cast_settings* cast, uint64_t seed, float* out_rgb, int width, int height, int shard_mode) {
    if (!g) return RTHIP_E_INVALID;
    RT_TRY(check_group_cast(g, cast, width, height, shard_mode));
    if (!out_rgb) return group_fail(g, RTHIP_E_INVALID, "out_rgb is NULL");
    const uint32_t world = (uint32_t)g->ctx.size();
    const size_t frame_bytes = (size_t)width * height * 12;
    if (shard_mode == 2 && world > 1) {  // partial sums are added on the first device, then one copy to the host
        rthip_ctx* c0 = g->ctx[0];
        RT_HIP(c0, hipSetDevice(c0->device));
        RT_TRY(rthip_reserve(c0, c0->frame, frame_bytes));
        c0->frame_w = c0->frame_h = 0;
        const int rc = rthip_group_cast_device(g, cast, seed, (float*)c0->frame.p, width, height, 2);
        if (rc != RTHIP_OK) return rc;
        RT_HIP(c0, hipSetDevice(c0->device));
        RT_HIP(c0, hipMemcpy(out_rgb, c0->frame.p, frame_bytes, hipMemcpyDeviceToHost));
        c0->frame_w = width;
        c0->frame_h = height;
        return RTHIP_OK;
    }
    // tiles: every device copies its own packed tiles to page-locked host memory and its thread places their rows
    return for_each_device(g, [&](int k) -> int {
        rthip_ctx* c = g->ctx[k];
        RT_TRY(render_share(g, k, cast, seed, width, height, 1));
        const uint32_t slots = tiles_of_rank(width, height, (uint32_t)k, world);
        if (!slots) return RTHIP_OK;
        const size_t bytes = (size_t)slots * kTileFloats * 4;
        if (g->host_stage_bytes[k] < bytes) {
            if (g->host_stage[k]) (void)hipHostFree(g->host_stage[k]);
            g->host_stage[k] = nullptr;
            g->host_stage_bytes[k] = 0;
            RT_HIP(c, hipHostMalloc(&g->host_stage[k], bytes, hipHostMallocDefault));
            g->host_stage_bytes[k] = bytes;
        }
        RT_HIP(c, hipMemcpyAsync(g->host_stage[k], g->packed[k].p, bytes, hipMemcpyDeviceToHost, c->stream));
        RT_HIP(c, hipStreamSynchronize(c->stream));
        const uint32_t tiles_x = ((uint32_t)width + kTile - 1) / kTile;
        const float* src = (const float*)g->host_stage[k];
        for (uint32_t s = 0; s < slots; s++) {
            const uint32_t tile = (uint32_t)k + s * world;
            const uint32_t x0 = (tile % tiles_x) * kTile, y0 = (tile / tiles_x) * kTile;
            const uint32_t cols = x0 + kTile <= (uint32_t)width ? kTile : (uint32_t)width - x0;
            for (uint32_t r = 0; r < kTile && y0 + r < (uint32_t)height; r++)
                memcpy(out_rgb + 3 * ((size_t)(y0 + r) * width + x0), src + (size_t)s * kTileFloats + 3 * (size_t)r * kTile, (size_t)cols * 12);
        }
        return RTHIP_OK;
    });
}

int rthip_group_cast_device(rthip_group* g, const rthip_cast_settings* cast, uint64_t seed, float* d_out_rgb, int width, int height,
                            int shard_mode) {
    if (!g) return RTHIP_E_INVALID;
    RT_TRY(check_group_cast(g, cast, width, height, shard_mode));
    if (!d_out_rgb) return group_fail(g, RTHIP_E_INVALID, "framebuffer is NULL");
    const uint32_t world = (uint32_t)g->ctx.size();
    rthip_ctx* c0 = g->ctx[0];
    const uint64_t floats = (uint64_t)width * height * 3;
    // where every rank's share lands on the first device
    std::vector<size_t> offset(world + 1, 0);
    for (uint32_t k = 0; k < world; k++)
        offset[k + 1] = offset[k] + (shard_mode == 1 ? (size_t)tiles_of_rank(width, height, k, world) * kTileFloats * 4 : (size_t)floats * 4);
    RT_HIP(c0, hipSetDevice(c0->device));
    RT_TRY(rthip_reserve(c0, g->gather, offset[world] ? offset[world] : 256));
    {
        const int rc = for_each_device(g, [&](int k) -> int {
            rthip_ctx* c = g->ctx[k];
            RT_TRY(render_share(g, k, cast, seed, width, height, shard_mode));
            const size_t bytes = offset[k + 1] - offset[k];
            if (bytes) {
                const void* src = shard_mode == 1 ? g->packed[k].p : g->frame[k].p;
                RT_HIP(c, hipMemcpyPeerAsync((char*)g->gather.p + offset[k], c0->device, src, c->device, bytes, c->stream));
            }
            RT_HIP(c, hipStreamSynchronize(c->stream));
            return RTHIP_OK;
        });
        if (rc != RTHIP_OK) return rc;
    }
    RT_HIP(c0, hipSetDevice(c0->device));
    if (shard_mode == 1) {
        for (uint32_t k = 0; k < world; k++) {
            const uint32_t slots = tiles_of_rank(width, height, k, world);
            if (!slots) continue;
            tiles_unpack_kernel<<<slots * (kTile * kTile / kBlock), kBlock, 0, c0->stream>>>(
                (const float*)((const char*)g->gather.p + offset[k]), width, height, k, world, d_out_rgb);
            RT_HIP(c0, hipGetLastError());
        }
    } else {
        const float scale = 1.f / ((float)cast->samples * (float)cast->samples);
        parts_sum_kernel<<<(uint32_t)((floats + kBlock - 1) / kBlock), kBlock, 0, c0->stream>>>((const float*)g->gather.p, world, floats,
                                                                                                world > 1 ? scale : 1.f, d_out_rgb);
        RT_HIP(c0, hipGetLastError());
    }
    RT_HIP(c0, hipStreamSynchronize(c0->stream));
    return RTHIP_OK;
}

}  // extern "C"
