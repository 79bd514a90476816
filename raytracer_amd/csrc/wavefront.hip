// Wavefront path tracer: rt_cpu_raygen / rt_cpu_trace_ray / rt_cpu_closest_hit
// (src/raytracer/cpu/raytracer_cpu.c:179-371) split into stages that each run with full waves.
//
//   wf_raygen   one thread per path (pixel, sample): camera ray + path state          -> TRACE
//   wf_trace    persistent waves walk the BVHs; a lane that finishes its ray takes the next
//               waiting ray of its wave's chunk (ballot + LDS compaction), so waves stay
//               full without a global queue                                            -> SHADE
//   wf_shade    one thread per path: hit epilogue, material, next ray                 -> TRACE | DONE
//   wf_resolve  one thread per pixel: sums the pixel's samples in the reference's order
//
// Path state lives in HBM as structure-of-arrays, one slot per path of the batch; a path never
// moves, its status byte says which stage owns it.  Because every random draw is a function of
// (seed, pixel, sample, draw index) and every pixel is summed in sample order, the image is
// bit-identical to the one-lane-per-path kernels of trace.hip whatever the scheduling.
//
// The trace kernel is small on purpose (no shading code, no camera, no material data): fewer
// registers -> more resident waves -> more node fetches in flight, which is what bounds a
// dependent-load walk over a tree that does not fit the caches.

#include "rt_device.h"

namespace {

enum : uint8_t { PS_DONE = 0, PS_TRACE = 1, PS_SHADE = 2 };

constexpr uint32_t kChunkPaths = 256;  // paths per chunk; chunks are dealt to waves round-robin

struct WfArrays {
    float4* ray_o;       // xyz origin
    float4* ray_d;       // xyz direction
    uint4* hit;          // inst, tri, bits(u), bits(v)
    float* hit_t;        // interval.max after the query
    uint64_t* rng_key;
    uint32_t* rng_draw;
    uint32_t* depth_sp;  // depth | sp << 8
    float2* ior;         // top, below
    uint32_t* mstack;    // [max_bounces][cap] materials of the scattering bounces
    float4* radiance;    // final radiance of the path
    uint8_t* status;
    uint32_t cap;        // slots per array
};

struct WfBatch {
    uint32_t first;    // first position (tile order) of the batch
    uint32_t n_paths;  // paths in this batch = positions * spp
    uint32_t spp;
    uint32_t tiles_x;
};

__device__ __forceinline__ void position_to_pixel(uint32_t pos, uint32_t tiles_x, uint32_t& px, uint32_t& py) {
    const uint32_t tile = pos >> 6, in = pos & 63u;
    px = (tile % tiles_x) * 8u + (in & 7u);
    py = (tile / tiles_x) * 8u + (in >> 3);
}

__device__ __forceinline__ bool pixel_is_mine(const rthip_shard& shard, uint32_t px, uint32_t py, int width, int height) {
    return shard_owns(shard, px, py, width, shard_band_rows(shard, height));
}

__global__ __launch_bounds__(kBlock) void wf_raygen_kernel(DevScene S, rthip_cast_settings c, uint64_t seed, int width,
                                                           int height, rthip_shard shard, WfArrays A, WfBatch B) {
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= B.n_paths) return;
    const uint32_t pos = B.first + p / B.spp, sidx = p % B.spp;
    uint32_t px, py;
    position_to_pixel(pos, B.tiles_x, px, py);
    bool live = px < (uint32_t)width && py < (uint32_t)height && pixel_is_mine(shard, px, py, width, height);
    if (shard.mode == 2 && (sidx % shard.world) != shard.rank) live = false;
    if (S.max_bounces == 0) live = false;  // rt_cpu_trace_ray with depth 0 is black (:247-249)
    A.radiance[p] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!live) {
        A.status[p] = PS_DONE;
        return;
    }
    const uint32_t pixel = py * (uint32_t)width + px;
    const int samples = (int)c.samples;
    Rng g = rng_make(seed, pixel, sidx);
    V3 o, d;
    camera_ray(c, width, height, (int)px, (int)py, (int)(sidx % (uint32_t)samples), (int)(sidx / (uint32_t)samples), true,
               g, o, d);
    A.ray_o[p] = make_float4(o.x, o.y, o.z, 0.f);
    A.ray_d[p] = make_float4(d.x, d.y, d.z, 0.f);
    A.rng_key[p] = g.key;
    A.rng_draw[p] = g.draw;
    A.depth_sp[p] = S.max_bounces;
    A.ior[p] = make_float2(c.ior, c.ior);
    A.status[p] = PS_TRACE;
}

// Persistent traversal.  Wave w owns chunks w, w + W, w + 2W ... of kChunkPaths consecutive paths.
template <bool COUNT>
__global__ __launch_bounds__(kBlock) void wf_trace_kernel(DevScene S, WfArrays A, uint32_t n_paths, uint32_t refill,
                                                          uint64_t* counters) {
    __shared__ uint32_t slots[kBlock / 64][64];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const uint32_t n_waves = gridDim.x * (kBlock / 64);
    const uint32_t wave_id = blockIdx.x * (kBlock / 64) + wave;
    volatile uint32_t* my_slots = slots[wave];

    Counters cnt = {0, 0, 0, 0};
    uint32_t rays = 0;
    bool active = false;
    uint32_t pid = 0;
    Trav T;
    trav_begin(T, S, mk(0.f, 0.f, 0.f), mk(0.f, 0.f, 1.f));

    // wave-uniform scan position
    uint64_t chunk = wave_id;
    uint32_t cursor = 0, chunk_end = 0;
    bool exhausted = (uint64_t)wave_id * kChunkPaths >= n_paths;
    if (!exhausted) {
        cursor = wave_id * kChunkPaths;
        chunk_end = cursor + kChunkPaths < n_paths ? cursor + kChunkPaths : n_paths;
    }

    for (;;) {
        uint64_t idle = __ballot(!active);
        const bool none_active = idle == ~0ull;
        if (!exhausted && ((uint32_t)__popcll(idle) >= refill || none_active)) {
            // hand waiting rays of this wave's chunks to the idle lanes
            while (idle != 0 && !exhausted) {
                if (cursor >= chunk_end) {
                    chunk += n_waves;
                    if (chunk * kChunkPaths >= n_paths) {
                        exhausted = true;
                        break;
                    }
                    cursor = (uint32_t)(chunk * kChunkPaths);
                    chunk_end = cursor + kChunkPaths < n_paths ? cursor + kChunkPaths : n_paths;
                }
                const uint32_t e = cursor + lane;
                const bool waiting = e < chunk_end && A.status[e] == PS_TRACE;
                const uint64_t wmask = __ballot(waiting);
                const uint32_t n_wait = (uint32_t)__popcll(wmask);
                if (n_wait == 0) {
                    cursor += 64;
                    continue;
                }
                if (waiting) my_slots[__popcll(wmask & lt_mask)] = e;  // compact the waiting paths
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint32_t n_idle = (uint32_t)__popcll(idle);
                const uint32_t take = n_wait < n_idle ? n_wait : n_idle;
                const uint32_t my_rank = (uint32_t)__popcll(idle & lt_mask);
                if (!active && my_rank < take) {
                    pid = my_slots[my_rank];
                    const float4 o = A.ray_o[pid], d = A.ray_d[pid];
                    trav_begin(T, S, mk(o.x, o.y, o.z), mk(d.x, d.y, d.z));
                    active = true;
                    rays++;
                }
                cursor = (n_wait <= n_idle) ? cursor + 64 : my_slots[take - 1] + 1;
                __builtin_amdgcn_wave_barrier();
                idle = __ballot(!active);
            }
        }
        const uint64_t walking = __ballot(active);
        if (walking == 0) {
            if (exhausted) break;
            continue;
        }
        if (COUNT && (int)lane == __ffsll((long long)walking) - 1) cnt.wave_steps++;
        if (COUNT && active) cnt.lane_steps++;
        if (active) {
            if (trav_step<COUNT>(T, S, cnt)) {
                A.hit[pid] = make_uint4(T.h.inst, T.h.tri, __float_as_uint(T.h.u), __float_as_uint(T.h.v));
                A.hit_t[pid] = T.h.t;
                A.status[pid] = PS_SHADE;
                active = false;
            }
        }
    }
    if (COUNT) flush_counters(counters, cnt, rays);
}

__global__ __launch_bounds__(kBlock) void wf_shade_kernel(DevScene S, WfArrays A, uint32_t n_paths) {
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n_paths || A.status[p] != PS_SHADE) return;
    const float4 ro = A.ray_o[p], rd = A.ray_d[p];
    V3 o = mk(ro.x, ro.y, ro.z), d = mk(rd.x, rd.y, rd.z);
    const uint4 hr = A.hit[p];
    Hit h;
    h.inst = hr.x;
    h.tri = hr.y;
    h.u = __uint_as_float(hr.z);
    h.v = __uint_as_float(hr.w);
    h.t = A.hit_t[p];
    PathState P;
    const uint32_t ds = A.depth_sp[p];
    P.depth = ds & 0xffu;
    P.sp = ds >> 8;
    const float2 ior = A.ior[p];
    P.ior_top = ior.x;
    P.ior_below = ior.y;
    P.g.key = A.rng_key[p];
    P.g.draw = A.rng_draw[p];
    V3 L;
    uint32_t* stack = A.mstack + p;
    if (path_bounce(S, P, h, o, d, L, stack, A.cap)) {
        L = path_fold(S, P, L, stack, A.cap);
        A.radiance[p] = make_float4(L.x, L.y, L.z, 0.f);
        A.status[p] = PS_DONE;
        return;
    }
    A.ray_o[p] = make_float4(o.x, o.y, o.z, 0.f);
    A.ray_d[p] = make_float4(d.x, d.y, d.z, 0.f);
    A.depth_sp[p] = P.depth | (P.sp << 8);
    A.ior[p] = make_float2(P.ior_top, P.ior_below);
    A.rng_draw[p] = P.g.draw;
    A.status[p] = PS_TRACE;
}

// `*c = add(*c, sample)` over the samples in order, then `*c = mul(*c, inv_sample_count)`,
// raytracer_cpu.c:190-239
__global__ __launch_bounds__(kBlock) void wf_resolve_kernel(rthip_cast_settings c, int width, int height, rthip_shard shard,
                                                            WfArrays A, WfBatch B, float* __restrict__ out,
                                                            uint32_t accumulate) {
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= B.n_paths / B.spp) return;
    uint32_t px, py;
    position_to_pixel(B.first + k, B.tiles_x, px, py);
    if (px >= (uint32_t)width || py >= (uint32_t)height) return;
    V3 acc = mk(0.f, 0.f, 0.f);
    const bool mine = pixel_is_mine(shard, px, py, width, height);
    if (!mine && shard.mode == 3) return;  // rows of another band are left alone
    if (mine) {
        for (uint32_t s = 0; s < B.spp; s++) {
            if (shard.mode == 2 && (s % shard.world) != shard.rank) continue;
            const float4 L = A.radiance[(size_t)k * B.spp + s];
            acc = add(acc, mk(L.x, L.y, L.z));
        }
        const int samples = (int)c.samples;
        if (shard.mode != 2) acc = scale(acc, 1.f / ((float)samples * (float)samples));
    }
    float* o3 = out + 3 * ((size_t)py * width + px);
    if (accumulate) {  // rthip_cast_accumulate_device: the frame is added to what the buffer holds
        o3[0] = o3[0] + acc.x;
        o3[1] = o3[1] + acc.y;
        o3[2] = o3[2] + acc.z;
    } else {
        o3[0] = acc.x;
        o3[1] = acc.y;
        o3[2] = acc.z;
    }
}

inline uint32_t div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

}  // namespace

int cast_wavefront_launch(rthip_ctx* ctx, const DevScene& scene, const rthip_cast_settings& cast, uint64_t seed,
                          float* d_rgb, int width, int height, rthip_shard shard, uint64_t* d_counters,
                          hipStream_t stream) {
    const uint32_t spp = cast.samples * cast.samples;
    const uint32_t tiles_x = div_up(width, 8), tiles_y = div_up(height, 8);
    const uint64_t positions = (uint64_t)tiles_x * tiles_y * 64ull;
    // batch: as many whole 64-position tiles as fit the path capacity
    uint64_t cap_paths = ctx->wf_capacity;
    uint64_t pos_per_batch = (cap_paths / spp) & ~63ull;
    if (pos_per_batch < 64) pos_per_batch = 64;
    if (pos_per_batch > positions) pos_per_batch = positions;
    const uint64_t cap = pos_per_batch * spp;
    if (cap > 0xffffffffull) return rthip_fail(ctx, RTHIP_E_LIMIT, "samples too large for the wavefront batch");
    const uint32_t depth = scene.max_bounces ? scene.max_bounces : 1;

    RT_TRY(rthip_reserve(ctx, ctx->wf[0], cap * 16));
    RT_TRY(rthip_reserve(ctx, ctx->wf[1], cap * 16));
    RT_TRY(rthip_reserve(ctx, ctx->wf[2], cap * 16));
    RT_TRY(rthip_reserve(ctx, ctx->wf[3], cap * 4));
    RT_TRY(rthip_reserve(ctx, ctx->wf[4], cap * 8));
    RT_TRY(rthip_reserve(ctx, ctx->wf[5], cap * 4));
    RT_TRY(rthip_reserve(ctx, ctx->wf[6], cap * 4));
    RT_TRY(rthip_reserve(ctx, ctx->wf[7], cap * 8));
    RT_TRY(rthip_reserve(ctx, ctx->wf[8], cap * 4 * depth));
    RT_TRY(rthip_reserve(ctx, ctx->wf[9], cap * 16));
    RT_TRY(rthip_reserve(ctx, ctx->wf[10], cap));
    WfArrays A;
    A.ray_o = (float4*)ctx->wf[0].p;
    A.ray_d = (float4*)ctx->wf[1].p;
    A.hit = (uint4*)ctx->wf[2].p;
    A.hit_t = (float*)ctx->wf[3].p;
    A.rng_key = (uint64_t*)ctx->wf[4].p;
    A.rng_draw = (uint32_t*)ctx->wf[5].p;
    A.depth_sp = (uint32_t*)ctx->wf[6].p;
    A.ior = (float2*)ctx->wf[7].p;
    A.mstack = (uint32_t*)ctx->wf[8].p;
    A.radiance = (float4*)ctx->wf[9].p;
    A.status = (uint8_t*)ctx->wf[10].p;
    A.cap = (uint32_t)cap;

    for (uint64_t first = 0; first < positions; first += pos_per_batch) {
        const uint64_t npos = positions - first < pos_per_batch ? positions - first : pos_per_batch;
        WfBatch B;
        B.first = (uint32_t)first;
        B.n_paths = (uint32_t)(npos * spp);
        B.spp = spp;
        B.tiles_x = tiles_x;
        const uint32_t pblocks = div_up(B.n_paths, kBlock);
        wf_raygen_kernel<<<pblocks, kBlock, 0, stream>>>(scene, cast, seed, width, height, shard, A, B);
        const uint32_t wf_refill = ctx->refill_lanes;
        uint32_t tblocks = div_up(div_up(B.n_paths, kChunkPaths), kBlock / 64);
        if (tblocks > 256u * 8u) tblocks = 256u * 8u;
        for (uint32_t bounce = 0; bounce < scene.max_bounces; bounce++) {
            if (d_counters)
                wf_trace_kernel<true><<<tblocks, kBlock, 0, stream>>>(scene, A, B.n_paths, wf_refill, d_counters);
            else
                wf_trace_kernel<false><<<tblocks, kBlock, 0, stream>>>(scene, A, B.n_paths, wf_refill, nullptr);
            wf_shade_kernel<<<pblocks, kBlock, 0, stream>>>(scene, A, B.n_paths);
        }
        wf_resolve_kernel<<<div_up(npos, kBlock), kBlock, 0, stream>>>(cast, width, height, shard, A, B, d_rgb, ctx->cast_accumulate);
        RT_HIP(ctx, hipGetLastError());
    }
    return RTHIP_OK;
}
