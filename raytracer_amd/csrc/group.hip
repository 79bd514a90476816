// Several GPUs of one node behind one handle: the scene is replicated (every device builds its own BVHs from the same
// host data), a frame is split by 64x64 tiles or by sample index (rthip_shard), and the shares are brought together with
// peer copies over xGMI: in tile mode every device ships only its own tiles (1/N of the frame), packed; in sample mode
// the devices' partial sums are added on the first device.  Transport between devices: RCCL (one communicator per device
// from ncclCommInitAll, the tiles gathered with grouped ncclSend / ncclRecv, the partial sums with ncclReduce) when the
// RCCL library can be loaded and the group's devices are distinct; otherwise hipMemcpyPeerAsync into the first device and
// a sum there in rank order.  One persistent host thread per device drives its context (started with the group, ended
// with it); nothing here computes on the host except placing tile rows into the caller's buffer.

#include <dlfcn.h>
#include <string.h>

#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <rccl/rccl.h>  // types and prototypes only: the library is loaded at run time (rccl_load), not linked

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

// The RCCL entry points the group uses, looked up in the library at run time: a process that never makes a group of
// several devices never loads it, and a box without it still gets the peer-copy transport.
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    bool ok = false;
};

// One host thread per device beyond the first, alive for as long as the group: a call hands every thread its piece and
// runs the first device's piece itself.
struct DeviceWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = false, stop = false;
    int rc = RTHIP_OK;
    void start() {
        th = std::thread([this] {
            for (;;) {
                std::unique_lock<std::mutex> lock(m);
                cv.wait(lock, [&] { return stop || has_job; });
                if (stop) return;
                std::function<int()> fn = std::move(job);
                has_job = false;
                lock.unlock();
                const int r = fn();
                lock.lock();
                rc = r;
                done = true;
                cv.notify_all();
            }
        });
    }
    void post(std::function<int()> fn) {
        std::lock_guard<std::mutex> lock(m);
        job = std::move(fn);
        has_job = true;
        done = false;
        cv.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lock(m);
        cv.wait(lock, [&] { return done; });
        return rc;
    }
    void end() {
        {
            std::lock_guard<std::mutex> lock(m);
            stop = true;
        }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};

}  // namespace

struct rthip_group {
    std::vector<rthip_ctx*> ctx;
    std::vector<DevBuf> frame;      // per device: its share of the frame (full size)
    std::vector<DevBuf> packed;     // per device: its tiles, packed
    std::vector<void*> host_stage;  // per device: page-locked staging for rthip_group_cast
    std::vector<size_t> host_stage_bytes;
    std::vector<DeviceWorker*> workers;  // workers[k - 1] drives device k
    DevBuf gather;                  // on device 0: every rank's packed tiles / partial frames (RCCL: the reduced frame)
    RcclApi rccl;
    std::vector<ncclComm_t> comms;  // one per device when the transport is RCCL
    bool use_rccl = false;
    std::string transport = "peer copies (hipMemcpyPeerAsync)";
    std::string err;
};

namespace {

int group_fail(rthip_group* g, int code, const std::string& msg) {
    g->err = msg;
    return code;
}

// fn(k) for every device: device 0 on the calling thread, the others on their own threads; the first failure is
// reported with the device's message
template <class F>
int for_each_device(rthip_group* g, F fn) {
    const size_t n = g->ctx.size();
    for (size_t k = 1; k < n; k++) g->workers[k - 1]->post([&fn, k] { return fn((int)k); });
    std::vector<int> rc(n, RTHIP_OK);
    rc[0] = fn(0);
    for (size_t k = 1; k < n; k++) rc[k] = g->workers[k - 1]->wait();
    for (size_t k = 0; k < n; k++)
        if (rc[k] != RTHIP_OK) return group_fail(g, rc[k], "device " + std::to_string(g->ctx[k]->device) + " (rank " + std::to_string(k) + "): " + g->ctx[k]->err);
    return RTHIP_OK;
}

bool rccl_load(RcclApi& a) {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* name : names)  // a copy the process already holds first (a host program that links RCCL, torch's own)
        if (!a.lib) a.lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    for (const char* name : names)
        if (!a.lib) a.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (!a.lib) return false;
#define RT_RCCL_SYM(field, symbol) a.field = (decltype(a.field))dlsym(a.lib, symbol)
    RT_RCCL_SYM(CommInitAll, "ncclCommInitAll");
    RT_RCCL_SYM(CommDestroy, "ncclCommDestroy");
    RT_RCCL_SYM(GetErrorString, "ncclGetErrorString");
    RT_RCCL_SYM(GroupStart, "ncclGroupStart");
    RT_RCCL_SYM(GroupEnd, "ncclGroupEnd");
    RT_RCCL_SYM(Reduce, "ncclReduce");
    RT_RCCL_SYM(Send, "ncclSend");
    RT_RCCL_SYM(Recv, "ncclRecv");
#undef RT_RCCL_SYM
    a.ok = a.CommInitAll && a.CommDestroy && a.GetErrorString && a.GroupStart && a.GroupEnd && a.Reduce && a.Send && a.Recv;
    return a.ok;
}

// RTHIP_GROUP_TRANSPORT: "peer" never loads RCCL; "rccl" insists on it (group creation fails without it, and a group of
// ONE device runs its exchange through it too: the rehearsal on a one-GPU box); unset / "auto": RCCL for several distinct
// devices when it can be had, peer copies otherwise.
int transport_setup(rthip_group* g) {
    const char* want = getenv("RTHIP_GROUP_TRANSPORT");
    const bool force_rccl = want && strcmp(want, "rccl") == 0, force_peer = want && strcmp(want, "peer") == 0;
    const size_t n = g->ctx.size();
    bool distinct = true;
    for (size_t a = 0; a < n; a++)
        for (size_t b = a + 1; b < n; b++) distinct = distinct && g->ctx[a]->device != g->ctx[b]->device;
    if (force_peer || (!force_rccl && (n < 2 || !distinct))) return RTHIP_OK;
    if (!distinct) return group_fail(g, RTHIP_E_INVALID, "RTHIP_GROUP_TRANSPORT=rccl needs distinct devices (one communicator per GPU)");
    if (!rccl_load(g->rccl)) {
        if (force_rccl) return group_fail(g, RTHIP_E_HIP, std::string("RTHIP_GROUP_TRANSPORT=rccl but the RCCL library cannot be loaded: ") + (dlerror() ? dlerror() : "symbols missing"));
        return RTHIP_OK;
    }
    std::vector<int> devs(n);
    for (size_t k = 0; k < n; k++) devs[k] = g->ctx[k]->device;
    g->comms.assign(n, nullptr);
    const ncclResult_t r = g->rccl.CommInitAll(g->comms.data(), (int)n, devs.data());
    if (r != ncclSuccess) {
        g->comms.clear();
        (void)hipGetLastError();
        if (force_rccl) return group_fail(g, RTHIP_E_HIP, std::string("ncclCommInitAll failed: ") + g->rccl.GetErrorString(r));
        return RTHIP_OK;  // the peer copies remain
    }
    g->use_rccl = true;
    g->transport = "RCCL (ncclCommInitAll; tiles by grouped ncclSend / ncclRecv, sample sums by ncclReduce)";
    return RTHIP_OK;
}

#define RT_RCCL(g, call)                                                                                                  \
    do {                                                                                                                  \
        const ncclResult_t r_ = (call);                                                                                   \
        if (r_ != ncclSuccess) return group_fail(g, RTHIP_E_HIP, std::string(#call) + ": " + (g)->rccl.GetErrorString(r_)); \
    } while (0)

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
    for (int k = 1; k < n; k++) {
        g->workers.push_back(new DeviceWorker());
        g->workers.back()->start();
    }
    const int rc = transport_setup(g);
    if (rc != RTHIP_OK) {
        fprintf(stderr, "rthip_group_create: %s\n", g->err.c_str());
        rthip_group_destroy(g);
        return rc;
    }
    *out = g;
    return RTHIP_OK;
}

void rthip_group_destroy(rthip_group* g) {
    if (!g) return;
    for (DeviceWorker* w : g->workers) {
        w->end();
        delete w;
    }
    g->workers.clear();
    for (size_t k = 0; k < g->ctx.size(); k++) {
        (void)hipSetDevice(g->ctx[k]->device);
        (void)hipDeviceSynchronize();
    }
    for (ncclComm_t c : g->comms)
        if (c) (void)g->rccl.CommDestroy(c);
    g->comms.clear();
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
const char* rthip_group_transport(const rthip_group* g) { return g ? g->transport.c_str() : "null group"; }

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
    const bool rccl = g->use_rccl;
    // where every rank's share lands on the first device (RCCL sums the partial frames of sample mode itself: one frame)
    std::vector<size_t> offset(world + 1, 0);
    for (uint32_t k = 0; k < world; k++)
        offset[k + 1] = offset[k] + (shard_mode == 1 ? (size_t)tiles_of_rank(width, height, k, world) * kTileFloats * 4
                                                     : ((rccl && k > 0) ? 0 : (size_t)floats * 4));
    RT_HIP(c0, hipSetDevice(c0->device));
    RT_TRY(rthip_reserve(c0, g->gather, offset[world] ? offset[world] : 256));
    {
        const int rc = for_each_device(g, [&](int k) -> int {
            rthip_ctx* c = g->ctx[k];
            RT_TRY(render_share(g, k, cast, seed, width, height, shard_mode));
            if (rccl) return RTHIP_OK;  // the exchange is enqueued for all devices together below
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
    if (rccl) {
        // one grouped call over every device's communicator and stream, behind the renders already enqueued there
        RT_RCCL(g, g->rccl.GroupStart());
        for (uint32_t k = 0; k < world; k++) {
            rthip_ctx* c = g->ctx[k];
            if (shard_mode == 1) {
                const size_t n = (offset[k + 1] - offset[k]) / 4;
                if (!n) continue;
                RT_RCCL(g, g->rccl.Send(g->packed[k].p, n, ncclFloat32, 0, g->comms[k], c->stream));
                RT_RCCL(g, g->rccl.Recv((char*)g->gather.p + offset[k], n, ncclFloat32, (int)k, g->comms[0], c0->stream));
            } else {
                RT_RCCL(g, g->rccl.Reduce(g->frame[k].p, g->gather.p, floats, ncclFloat32, ncclSum, 0, g->comms[k], c->stream));
            }
        }
        RT_RCCL(g, g->rccl.GroupEnd());
        for (uint32_t k = 1; k < world; k++) {
            RT_HIP(g->ctx[k], hipSetDevice(g->ctx[k]->device));
            RT_HIP(g->ctx[k], hipStreamSynchronize(g->ctx[k]->stream));
        }
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
        parts_sum_kernel<<<(uint32_t)((floats + kBlock - 1) / kBlock), kBlock, 0, c0->stream>>>((const float*)g->gather.p, rccl ? 1u : world, floats,
                                                                                                world > 1 ? scale : 1.f, d_out_rgb);
        RT_HIP(c0, hipGetLastError());
    }
    RT_HIP(c0, hipStreamSynchronize(c0->stream));
    return RTHIP_OK;
}

/* ---- the tile exchange's two kernels for callers that move the tiles themselves (one process per GPU over RCCL) ---- */

uint32_t rthip_tiles_of_rank(int width, int height, uint32_t rank, uint32_t world) {
    return (width > 0 && height > 0 && world > 0 && rank < world) ? tiles_of_rank(width, height, rank, world) : 0u;
}

int rthip_tiles_pack_device(rthip_ctx* ctx, const float* d_rgb, int width, int height, uint32_t rank, uint32_t world, float* d_packed,
                            void* stream) {
    if (!ctx) return RTHIP_E_INVALID;
    if (!d_rgb || !d_packed || width <= 0 || height <= 0 || world == 0 || rank >= world)
        return rthip_fail(ctx, RTHIP_E_INVALID, "rthip_tiles_pack_device: bad arguments");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t slots = tiles_of_rank(width, height, rank, world);
    if (!slots) return RTHIP_OK;
    tiles_pack_kernel<<<slots * (kTile * kTile / kBlock), kBlock, 0, (hipStream_t)stream>>>(d_rgb, width, height, rank, world, d_packed);
    RT_HIP(ctx, hipGetLastError());
    return RTHIP_OK;
}

int rthip_tiles_unpack_device(rthip_ctx* ctx, const float* d_packed, int width, int height, uint32_t rank, uint32_t world, float* d_rgb,
                              void* stream) {
    if (!ctx) return RTHIP_E_INVALID;
    if (!d_rgb || !d_packed || width <= 0 || height <= 0 || world == 0 || rank >= world)
        return rthip_fail(ctx, RTHIP_E_INVALID, "rthip_tiles_unpack_device: bad arguments");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t slots = tiles_of_rank(width, height, rank, world);
    if (!slots) return RTHIP_OK;
    tiles_unpack_kernel<<<slots * (kTile * kTile / kBlock), kBlock, 0, (hipStream_t)stream>>>(d_packed, width, height, rank, world, d_rgb);
    RT_HIP(ctx, hipGetLastError());
    return RTHIP_OK;
}

}  // extern "C"
