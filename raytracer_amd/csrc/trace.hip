// Rendering kernels that keep a whole camera path inside one lane: the persistent kernel with
// per-lane refill, the fixed pixel-per-thread kernel, and the parity taps (primary hits, caller
// supplied rays).  Device-side building blocks live in rt_device.h.

#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "rt_device.h"

namespace {

// rt_cpu_raygen, raytracer_cpu.c:179-242, as a persistent kernel.
//
// Work item = one pixel (all of its samples, in the reference's order: y_sample outer, x_sample
// inner, summed in that order, then scaled by 1/samples^2).  8x8-pixel tiles are dealt to the
// waves round-robin and a lane that has finished its pixel takes the next pixel of its wave's
// tile sequence, so a wave that starts together starts on one tile.  Every lane is a small
// state machine:
//
//   NEED_PIXEL -> NEED_SAMPLE -> TRAVERSE -> SHADE -> TRAVERSE ... -> NEED_SAMPLE ... -> NEED_PIXEL
//
// and the wave loops over three phases: fetch/raygen for the lanes that need it, a traversal
// loop in which every TRAVERSE lane advances one node per iteration, and shading for the lanes
// whose query has ended.  The traversal loop is left as soon as `refill` lanes have something else
// to do (or nobody traverses), so finished lanes get their next ray instead of idling until the
// longest ray of the wave is done.  Refill slots are handed out with one ballot + mbcnt-style
// prefix; there is no global queue and no lane ever waits for another wave.
enum : uint32_t { ST_NEED_PIXEL = 0, ST_NEED_SAMPLE = 1, ST_TRAVERSE = 2, ST_SHADE = 3, ST_DONE = 4 };

// A pixel's result goes to the framebuffer, or is added to it (progressive accumulation: rthip_cast_accumulate_device).
__device__ __forceinline__ void put_pixel(float* __restrict__ out, uint32_t pixel, V3 v, uint32_t accumulate) {
    float* o3 = out + 3 * (size_t)pixel;
    if (accumulate) {
        o3[0] = o3[0] + v.x;
        o3[1] = o3[1] + v.y;
        o3[2] = o3[2] + v.z;
    } else {
        o3[0] = v.x;
        o3[1] = v.y;
        o3[2] = v.z;
    }
}

// The state of one lane's query in either traversal mode behind one interface, so that every renderer kernel exists
// for both: the reference's fixed-order stackless walk (Trav) and the ordered walk with a stack (TravFast).
template <bool FAST>
struct Walk;
template <>
struct Walk<false> {
    Trav T;
    __device__ __forceinline__ void init(uint32_t*) {}
    __device__ __forceinline__ void begin(const DevScene& S, V3 o, V3 d) { trav_begin(T, S, o, d); }
    template <bool COUNT, bool DEFER>
    __device__ __forceinline__ bool step(const DevScene& S, Counters& cnt) { return trav_step<COUNT, DEFER>(T, S, cnt); }
    template <bool COUNT>
    __device__ __forceinline__ void enter(const DevScene& S, Counters& cnt) { trav_enter<COUNT>(T, S, cnt); }
};
template <>
struct Walk<true> {
    TravFast T;
    FastStack st;
    __device__ __forceinline__ void init(uint32_t* lds_base) { st.lds = fast_stack_column(lds_base); }
    __device__ __forceinline__ void begin(const DevScene& S, V3 o, V3 d) { trav_fast_begin(T, S, o, d); }
    template <bool COUNT, bool DEFER>
    __device__ __forceinline__ bool step(const DevScene& S, Counters& cnt) {
        if (!trav_fast_step<COUNT, DEFER>(T, S, cnt, st)) return false;
        if (T.overflow) T.h = closest_hit_query<COUNT>(S, T.wo, T.wd, cnt);  // see kFastLds
        return true;
    }
    template <bool COUNT>
    __device__ __forceinline__ void enter(const DevScene& S, Counters& cnt) { trav_fast_enter<COUNT>(T, S, cnt); }
};

// The workgroup's block of the ordered walk's stacks: it lies behind the bounce stack in the dynamic LDS block.
__device__ __forceinline__ uint32_t* fast_column(uint32_t* bounce_stack_base, uint32_t max_bounces) {
    return bounce_stack_base + (size_t)(max_bounces ? max_bounces : 1u) * kBlock;
}

// One closest-hit query in the chosen mode.
template <bool COUNT, bool FAST>
__device__ __forceinline__ Hit query(const DevScene& S, V3 o, V3 d, Counters& cnt, uint32_t* fast_col) {
    if constexpr (FAST) return closest_hit_query_fast<COUNT>(S, o, d, cnt, fast_col);
    else return closest_hit_query<COUNT>(S, o, d, cnt);
}

// End of a persistent kernel: the slot of work counters is left zeroed for its next user by the last wave to get here
// (every wave counts itself in word 16 when it has stopped taking work; no barrier, a finished wave leaves at once).
// No memset between launches, so a launch is one graph node and replays of a captured graph start from zeroed counters.
__device__ __forceinline__ void work_slot_release(uint32_t* work) {
    if ((threadIdx.x & 63u) == 0) {
        const uint32_t arrived = atomicAdd(&work[16], 1u);
        if (arrived == gridDim.x * (kBlock / 64) - 1u) {
            for (uint32_t k = 0; k < 8u; k++) atomicExch(&work[k * 32u], 0u);
            atomicExch(&work[16], 0u);
        }
    }
}

#ifndef RT_PERSIST_WAVES
#define RT_PERSIST_WAVES 5
#endif
#ifndef RT_FAST_WAVES
#define RT_FAST_WAVES 4  // waves per SIMD of the renderers under the ordered walk (their stacks in LDS allow four workgroups per CU)
#endif
template <bool COUNT, bool DEFER, bool FAST = false>
__global__ __launch_bounds__(kBlock, FAST ? RT_FAST_WAVES : RT_PERSIST_WAVES) void cast_kernel(DevScene S, rthip_cast_settings c, uint64_t seed,
                                                      float* __restrict__ out, int width, int height,
                                                      rthip_shard shard, uint32_t* work_counters, uint32_t refill, uint32_t inst_batch,
                                                      uint32_t accumulate, uint64_t* counters) {
    extern __shared__ uint32_t bounce_stack[];
    uint32_t* stack = bounce_stack + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const int samples = (int)c.samples;
    const uint32_t spp = (uint32_t)(samples * samples);
    const uint32_t tiles_x = (uint32_t)(width + 7) / 8, tiles_y = (uint32_t)(height + 7) / 8;
    const uint64_t n_tiles = (uint64_t)tiles_x * tiles_y;
    const uint32_t wave_id = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    // wave-uniform bookkeeping of the tile the wave is handing out
    uint64_t tile_id = 0;
    uint32_t tile_pos = 64;  // positions of tile_id already handed out (64: none left)
    uint32_t qslot = wave_id & 7u, qtried = 0;  // which of the 8 dynamic counters, and how many ran dry
    const uint32_t band_rows = shard_band_rows(shard, height);

    Counters cnt = {0, 0, 0, 0};
    uint32_t rays = 0;
    uint32_t state = ST_NEED_PIXEL;
    bool exhausted = false;  // wave-uniform: the work counter has run past the last pixel
    uint32_t pixel = 0, px = 0, py = 0, sidx = 0;
    V3 acc = mk(0.f, 0.f, 0.f);
    PathState P;
    P.depth = 0;
    P.sp = 0;
    P.ior_top = P.ior_below = c.ior;
    P.g.key = 0;
    P.g.draw = 0;
    Walk<FAST> Wk;
    Wk.init(fast_column(bounce_stack, S.max_bounces));
    Wk.begin(S, mk(0.f, 0.f, 0.f), mk(0.f, 0.f, 1.f));

    for (;;) {
        // ---- phase 1: a pixel whose samples are all done is written out ---------------------------
        if (state == ST_NEED_SAMPLE) {
            if (shard.mode == 2)
                while (sidx < spp && (sidx % shard.world) != shard.rank) sidx++;
            if (sidx >= spp) {
                if (shard.mode != 2) acc = scale(acc, 1.f / ((float)samples * (float)samples));
                put_pixel(out, pixel, acc, accumulate);
                state = ST_NEED_PIXEL;
            }
        }

        // ---- phase 2: pixels for the lanes that have none ---------------------------------------------
        // A wave owns one 8x8 tile at a time and hands its 64 positions to idle lanes with a ballot and
        // an mbcnt-style prefix.  A new tile costs ONE atomic per wave.  Tiles come from 8 interleaved
        // counters (counter s serves tiles 8k + s), so that waves with cheap tiles take over work from
        // waves with expensive ones and no single word has to serve every wave.
        uint64_t need = __ballot(state == ST_NEED_PIXEL);
        while (need != 0 && !exhausted) {
            if (tile_pos >= 64u) {  // the wave's tile is used up: next tile (wave-uniform)
                bool got = false;
                while (qtried < 8u) {
                    uint32_t k = 0;
                    if (lane == 0) k = atomicAdd(&work_counters[qslot * 32u], 1u);
                    k = __shfl(k, 0);
                    const uint64_t t = (uint64_t)k * 8u + qslot;
                    if (t < n_tiles) {
                        tile_id = t;
                        tile_pos = 0;
                        got = true;
                        break;
                    }
                    qslot = (qslot + 1u) & 7u;
                    qtried++;
                }
                if (!got) {
                    exhausted = true;
                    break;
                }
            }
            const uint32_t room = 64u - tile_pos;
            const uint32_t rank_in_need = (uint32_t)__popcll(need & lt_mask);
            const bool served = (state == ST_NEED_PIXEL) && ((need >> lane) & 1ull) && rank_in_need < room;
            if (served) {
                const uint32_t in = tile_pos + rank_in_need;
                const uint32_t tile = (uint32_t)tile_id;
                px = (tile % tiles_x) * 8u + (in & 7u);
                py = (tile / tiles_x) * 8u + (in >> 3);
                if (px < (uint32_t)width && py < (uint32_t)height) {
                    pixel = py * (uint32_t)width + px;
                    const bool mine = shard_owns(shard, px, py, width, band_rows);
                    if (mine) {
                        acc = mk(0.f, 0.f, 0.f);
                        sidx = 0;
                        if (shard.mode == 2)
                            while (sidx < spp && (sidx % shard.world) != shard.rank) sidx++;
                        state = ST_NEED_SAMPLE;
                    } else if (!accumulate && shard.mode != 3) {  // another rank's tile: this rank's share of the pixel is 0 (rows of another band are left alone)
                        put_pixel(out, pixel, mk(0.f, 0.f, 0.f), 0u);
                    }
                }
                // positions outside the image (partial tiles) and foreign pixels: the lane asks again
            }
            const uint32_t n = (uint32_t)__popcll(need);
            tile_pos += (n < room) ? n : room;
            need = __ballot(state == ST_NEED_PIXEL && !served);
        }
        if (exhausted && state == ST_NEED_PIXEL) state = ST_DONE;
        if (__ballot(state != ST_DONE) == 0) break;

        // ---- phase 3: camera ray of the next sample ------------------------------------------------
        if (state == ST_NEED_SAMPLE && sidx < spp) {
            P.g = rng_make(seed, pixel, sidx);
            V3 o, d;
            camera_ray(c, width, height, (int)px, (int)py, (int)(sidx % (uint32_t)samples),
                       (int)(sidx / (uint32_t)samples), true, P.g, o, d);
            P.depth = S.max_bounces;
            P.sp = 0;
            P.ior_top = P.ior_below = c.ior;
            if (P.depth == 0) {  // max_bounces == 0: every path is black (raytracer_cpu.c:247-249)
                acc = add(acc, mk(0.f, 0.f, 0.f));
                sidx++;
            } else {
                Wk.begin(S, o, d);
                rays++;
                state = ST_TRAVERSE;
            }
        }

        // ---- phase 4: traversal, until enough lanes have something else to do ------------------------
        for (;;) {
            const uint64_t walking = __ballot(state == ST_TRAVERSE);
            if (walking == 0) break;
            const uint64_t waiting = __ballot(state != ST_TRAVERSE && state != ST_DONE);
            if ((uint32_t)__popcll(waiting) >= refill) break;
            // Entering an instance (two quaternion rotations, nine divisions) is by far the longest branch of
            // a step and only a few lanes need it in any one iteration.  With DEFER a lane that reaches an
            // instance parks there, and the wave runs the entry once `inst_batch` lanes are parked (or nobody
            // else can step): the same work per lane, in fewer executions of the long branch.  It pays where
            // rays enter instances often (many small instances); elsewhere the plain step is faster.
            if (COUNT && (int)lane == __ffsll((long long)walking) - 1) cnt.wave_steps++;
            if (COUNT && state == ST_TRAVERSE) cnt.lane_steps++;
            if constexpr (DEFER) {
                const uint64_t parked = __ballot(state == ST_TRAVERSE && Wk.T.pend_id != 0);
                const bool enter_now =
                    parked != 0 && ((uint32_t)__popcll(parked) >= inst_batch || (walking & ~parked) == 0);
                if (state == ST_TRAVERSE) {
                    if (Wk.T.pend_id != 0) {
                        if (enter_now) Wk.template enter<COUNT>(S, cnt);
                    } else if (Wk.template step<COUNT, true>(S, cnt)) {
                        state = ST_SHADE;
                    }
                }
            } else if (state == ST_TRAVERSE) {
                if (Wk.template step<COUNT, false>(S, cnt)) state = ST_SHADE;
            }
        }

        // ---- phase 5: shading / bounce generation ----------------------------------------------------
        if (state == ST_SHADE) {
            V3 o = Wk.T.wo, d = Wk.T.wd;
            V3 L;
            if (path_bounce(S, P, Wk.T.h, o, d, L, stack, kBlock)) {
                L = path_fold(S, P, L, stack, kBlock);
                acc = add(acc, L);
                sidx++;
                state = ST_NEED_SAMPLE;
            } else {
                Wk.begin(S, o, d);
                rays++;
                state = ST_TRAVERSE;
            }
        }
    }
    work_slot_release(work_counters);
    if (COUNT) flush_counters(counters, cnt, rays);
}

#ifdef RT_STAMP
__device__ uint64_t g_stamps[1u << 18];  // dev build only: start / end wall clock of every wave of the static kernel
#endif

// A wave covers an 8x8 pixel tile, a workgroup 16x16, so that the rays of a wave stay close.
__device__ __forceinline__ bool pixel_of_thread(int width, int height, int& x, int& y) {
    const int tiles_x = (width + 15) / 16;
    const int bx = blockIdx.x % tiles_x, by = blockIdx.x / tiles_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    x = bx * 16 + (wave & 1) * 8 + (lane & 7);
    y = by * 16 + (wave >> 1) * 8 + (lane >> 3);
    return x < width && y < height;
}

// The same renderer with a fixed pixel per thread (wave = 8x8 tile) and no refill: less
// bookkeeping per path, but a wave is as slow as its longest pixel.  Faster than the persistent
// kernel when paths are short and neighbouring pixels behave alike.
#ifndef RT_STATIC_WAVES
#define RT_STATIC_WAVES 5
#endif
template <bool COUNT, bool BATCH = false, bool FAST = false>
__global__ __launch_bounds__(kBlock, FAST ? RT_FAST_WAVES : RT_STATIC_WAVES) void cast_static_kernel(DevScene S, rthip_cast_settings c, uint64_t seed,
                                                             float* __restrict__ out, int width, int height,
                                                             rthip_shard shard, uint32_t accumulate, uint64_t* counters,
                                                             uint32_t inst_batch = 0) {
    extern __shared__ uint32_t bounce_stack[];
    uint32_t* stack = bounce_stack + threadIdx.x;
    int x, y;
    const bool inside = pixel_of_thread(width, height, x, y);
    Counters cnt = {0, 0, 0, 0};
    uint32_t rays = 0;
#ifdef RT_STAMP
    const uint64_t stamp0 = wall_clock64();
#endif
    if (inside) {
        const int samples = (int)c.samples;
        const uint32_t pixel = (uint32_t)(y * width + x);
        const bool mine = shard_owns(shard, (uint32_t)x, (uint32_t)y, width, shard_band_rows(shard, height));
        V3 acc = mk(0.f, 0.f, 0.f);
        if (mine) {
            for (int ys = 0; ys < samples; ys++)
                for (int xs = 0; xs < samples; xs++) {
                    const uint32_t sidx = (uint32_t)(ys * samples + xs);
                    if (shard.mode == 2 && (sidx % shard.world) != shard.rank) continue;
                    PathState P;
                    P.g = rng_make(seed, pixel, sidx);
                    V3 o, d;
                    camera_ray(c, width, height, x, y, xs, ys, true, P.g, o, d);
                    P.depth = S.max_bounces;
                    P.sp = 0;
                    P.ior_top = P.ior_below = c.ior;
                    V3 L = mk(0.f, 0.f, 0.f);
                    if (P.depth != 0) {
                        for (;;) {
                            rays++;
                            const Hit h = BATCH ? closest_hit_query_batched<COUNT>(S, o, d, cnt, inst_batch)
                                                : query<COUNT, FAST>(S, o, d, cnt, fast_column(bounce_stack, S.max_bounces));
                            if (path_bounce(S, P, h, o, d, L, stack, kBlock)) break;
                        }
                        L = path_fold(S, P, L, stack, kBlock);
                    }
                    acc = add(acc, L);
                }
            if (shard.mode != 2) acc = scale(acc, 1.f / ((float)samples * (float)samples));
        }
        if (mine || (!accumulate && shard.mode != 3)) put_pixel(out, pixel, acc, accumulate);
    }
#ifdef RT_STAMP
    {
        uint64_t steps = 0, nodes = 0;
        for (int l = 0; l < 64; l++) {
            steps += __shfl((unsigned)cnt.wave_steps, l);
            nodes += __shfl((unsigned)cnt.nodes, l);
        }
        if ((threadIdx.x & 63) == 0) {
            const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
            if (w < (1u << 16)) {
                g_stamps[4 * w] = stamp0;
                g_stamps[4 * w + 1] = wall_clock64();
                g_stamps[4 * w + 2] = steps;
                g_stamps[4 * w + 3] = nodes;
            }
        }
    }
#endif
    if (COUNT) flush_counters(counters, cnt, rays);
}

// The pixel-per-thread renderer with one lane per SAMPLE instead: 2^log2_group lanes (4, 16 or 64 = samples^2)
// share a pixel, a wave covers 4x4, 2x2 or 1 pixel.  A frame with few pixels and many samples per pixel becomes
// samples^2 times as many waves of one path per lane, which the dispatcher spreads where the pixel-per-thread
// kernel would hold the frame in a few long-running waves (Cornell 512^2 at 4 spp: 4096 waves, all resident from the
// start, the slowest 1.5 times the median).  The samples of a pixel are added in the order of the sample loop
// (raytracer_cpu.c:196-236) by the pixel's first lane, so the sum has the same bits.
#ifndef RT_SAMPLES_WAVES
#define RT_SAMPLES_WAVES 7  // (Cornell 512^2 at 4 spp, Mrays/s at 5 / 6 / 7 waves per SIMD: 7139 / 7710 / 7856; the pixel-per-lane kernel loses from 6 up)
#endif
template <bool COUNT, bool BATCH, bool FAST = false>
__global__ __launch_bounds__(kBlock, FAST ? RT_FAST_WAVES : RT_SAMPLES_WAVES) void cast_samples_kernel(DevScene S, rthip_cast_settings c, uint64_t seed,
                                                              float* __restrict__ out, int width, int height,
                                                              rthip_shard shard, uint32_t log2_group, uint32_t accumulate,
                                                              uint64_t* counters, uint32_t inst_batch) {
    extern __shared__ uint32_t bounce_stack[];
    uint32_t* stack = bounce_stack + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t group = 1u << log2_group;
    const uint32_t sidx = lane & (group - 1u);
    const uint32_t pix = lane >> log2_group;                                   // pixel of the wave
    const uint32_t side = log2_group == 2 ? 4u : (log2_group == 4 ? 2u : 1u);  // the wave covers side x side pixels
    const uint32_t tiles_x = ((uint32_t)width + 2u * side - 1u) / (2u * side);
    const int x = (int)((blockIdx.x % tiles_x) * 2u * side + (wave & 1u) * side + pix % side);
    const int y = (int)((blockIdx.x / tiles_x) * 2u * side + (wave >> 1) * side + pix / side);
    const bool inside = x < width && y < height;
    const int samples = (int)c.samples;
    const uint32_t pixel = inside ? (uint32_t)(y * width + x) : 0u;
    Counters cnt = {0, 0, 0, 0};
    uint32_t rays = 0;
    const bool mine = inside && shard_owns(shard, (uint32_t)x, (uint32_t)y, width, shard_band_rows(shard, height));
    const bool have = mine && !(shard.mode == 2 && (sidx % shard.world) != shard.rank);
    V3 L = mk(0.f, 0.f, 0.f);
    if (have) {
        PathState P;
        P.g = rng_make(seed, pixel, sidx);
        V3 o, d;
        camera_ray(c, width, height, x, y, (int)(sidx % (uint32_t)samples), (int)(sidx / (uint32_t)samples), true, P.g, o, d);
        P.depth = S.max_bounces;
        P.sp = 0;
        P.ior_top = P.ior_below = c.ior;
        if (P.depth != 0) {
            for (;;) {
                rays++;
                const Hit h = BATCH ? closest_hit_query_batched<COUNT>(S, o, d, cnt, inst_batch)
                                    : query<COUNT, FAST>(S, o, d, cnt, fast_column(bounce_stack, S.max_bounces));
                if (path_bounce(S, P, h, o, d, L, stack, kBlock)) break;
            }
            L = path_fold(S, P, L, stack, kBlock);
        }
    }
    // every lane of a pixel walks through the pixel's samples in order; the first lane keeps the sum
    V3 acc = mk(0.f, 0.f, 0.f);
    const uint32_t first = lane & ~(group - 1u);
    for (uint32_t k = 0; k < group; k++) {
        const float lx = __shfl(L.x, (int)(first + k)), ly = __shfl(L.y, (int)(first + k)), lz = __shfl(L.z, (int)(first + k));
        const int has = __shfl((int)have, (int)(first + k));
        if (has) acc = add(acc, mk(lx, ly, lz));
    }
    if (inside && sidx == 0) {
        if (mine && shard.mode != 2) acc = scale(acc, 1.f / ((float)samples * (float)samples));
        if (mine || (!accumulate && shard.mode != 3)) put_pixel(out, pixel, acc, accumulate);
    }
    if (COUNT) flush_counters(counters, cnt, rays);
}

// Debugging aid: the renderer with one thread per path (pixel-major, then sample: the reference's loop order) that
// also stores what a BUILD_DEBUG build of the reference dumps for every ray (src/raytracer/cpu/extra/dump.c:11-32):
// path i writes its k-th ray to slots[i * max_bounces + k] and the number of its rays to counts[i].
template <bool FAST>
__global__ __launch_bounds__(kBlock) void cast_records_kernel(DevScene S, rthip_cast_settings c, uint64_t seed, int width,
                                                              int height, rthip_ray_record* __restrict__ slots,
                                                              uint32_t* __restrict__ counts) {
    extern __shared__ uint32_t bounce_stack[];
    uint32_t* stack = bounce_stack + threadIdx.x;
    const uint32_t samples = c.samples, spp = samples * samples;
    const uint64_t path = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (path >= (uint64_t)width * height * spp) return;
    const uint32_t pixel = (uint32_t)(path / spp), sidx = (uint32_t)(path % spp);
    const int x = (int)(pixel % (uint32_t)width), y = (int)(pixel / (uint32_t)width);
    Counters cnt = {0, 0, 0, 0};
    PathState P;
    P.g = rng_make(seed, pixel, sidx);
    V3 o, d;
    camera_ray(c, width, height, x, y, (int)(sidx % samples), (int)(sidx / samples), true, P.g, o, d);
    P.depth = S.max_bounces;
    P.sp = 0;
    P.ior_top = P.ior_below = c.ior;
    uint32_t n = 0;
    rthip_ray_record* mine = slots + path * S.max_bounces;
    if (P.depth != 0) {
        V3 L = mk(0.f, 0.f, 0.f);
        for (;;) {
            const Hit h = query<false, FAST>(S, o, d, cnt, fast_column(bounce_stack, S.max_bounces));
            rthip_ray_record r;
            r.origin[0] = o.x, r.origin[1] = o.y, r.origin[2] = o.z;
            r.direction[0] = d.x, r.direction[1] = d.y, r.direction[2] = d.z;
            r.p[0] = r.p[1] = r.p[2] = 0.f;
            r.n[0] = r.n[1] = r.n[2] = 0.f;
            r.t = -1.0f;
            r.material = -1;
            if (h.inst != 0) {
                V3 p, nn;
                int material;
                finish_hit(S, o, d, h, p, nn, material);
                r.p[0] = p.x, r.p[1] = p.y, r.p[2] = p.z;
                r.n[0] = nn.x, r.n[1] = nn.y, r.n[2] = nn.z;
                r.t = h.t;
                r.material = material;
            }
            mine[n++] = r;
            if (path_bounce(S, P, h, o, d, L, stack, kBlock)) break;
        }
    }
    counts[path] = n;
}

// counts_scan: inclusive prefix sum of the counts; path i's records go to out[counts_scan[i] - count ...]
__global__ __launch_bounds__(kBlock) void records_compact_kernel(const rthip_ray_record* __restrict__ slots,
                                                                 const uint32_t* __restrict__ counts_scan, uint64_t n_paths,
                                                                 uint32_t max_bounces, rthip_ray_record* __restrict__ out) {
    const uint64_t path = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (path >= n_paths) return;
    const uint32_t end = counts_scan[path], begin = path ? counts_scan[path - 1] : 0u;
    for (uint32_t k = 0; k < end - begin; k++) out[begin + k] = slots[path * max_bounces + k];
}

template <bool COUNT, bool FAST = false>
__global__ __launch_bounds__(kBlock) void primary_hits_kernel(DevScene S, rthip_cast_settings c, int width, int height,
                                                              uint32_t* __restrict__ hits4, uint64_t* counters) {
    extern __shared__ uint32_t fast_lds[];  // FAST: kFastLds entries per lane
    int x, y;
    const bool inside = pixel_of_thread(width, height, x, y);
    Counters cnt = {0, 0, 0, 0};
    uint32_t rays = 0;
    if (inside) {
        rthip_cast_settings c1 = c;
        c1.samples = 1;
        Rng g = {0, 0};
        V3 o, d;
        camera_ray(c1, width, height, x, y, 0, 0, false, g, o, d);
        rays = 1;
        const Hit h = query<COUNT, FAST>(S, o, d, cnt, fast_lds);
        uint32_t* r = hits4 + 4 * ((size_t)y * width + x);
        r[0] = h.inst;
        r[1] = h.inst ? h.tri : 0u;
        r[2] = __float_as_uint(h.inst ? h.t : -1.0f);
        r[3] = 0;
    }
    if (COUNT) flush_counters(counters, cnt, rays);
}

// Result record of one explicit ray (the taps' output format).
__device__ __forceinline__ void write_ray_result(const DevScene& S, uint64_t i, const Hit& h, V3 o, V3 d,
                                                 uint32_t* __restrict__ hits4, float* __restrict__ uv,
                                                 float* __restrict__ normal) {
    uint32_t* r = hits4 + 4 * i;
    r[0] = h.inst;
    r[1] = h.inst ? h.tri : 0u;
    r[2] = __float_as_uint(h.inst ? h.t : -1.0f);
    r[3] = 0;
    if (uv) {
        uv[2 * i] = h.inst ? h.u : 0.f;
        uv[2 * i + 1] = h.inst ? h.v : 0.f;
    }
    if (normal) {
        V3 p, nn = mk(0.f, 0.f, 0.f);
        int material;
        if (h.inst) finish_hit(S, o, d, h, p, nn, material);
        normal[3 * i] = nn.x;
        normal[3 * i + 1] = nn.y;
        normal[3 * i + 2] = nn.z;
    }
}

template <bool COUNT, bool FAST = false>
__global__ __launch_bounds__(kBlock) void trace_rays_kernel(DevScene S, const float* __restrict__ ro,
                                                            const float* __restrict__ rd, uint64_t n,
                                                            uint32_t* __restrict__ hits4, float* __restrict__ uv,
                                                            float* __restrict__ normal, uint64_t* counters) {
    extern __shared__ uint32_t fast_lds[];  // FAST: kFastLds entries per lane
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    Counters cnt = {0, 0, 0, 0};
    uint32_t rays = 0;
    if (i < n) {
        const V3 o = mk(ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]);
        const V3 d = mk(rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]);
        rays = 1;
        const Hit h = query<COUNT, FAST>(S, o, d, cnt, fast_lds);
        write_ray_result(S, i, h, o, d, hits4, uv, normal);
    }
    if (COUNT) flush_counters(counters, cnt, rays);
}

// The same queries for long ray streams: resident waves, and a lane whose ray has ended takes the
// next ray of the stream instead of idling until the longest ray of its wave is done.  A wave
// reserves kRayGrab consecutive rays with one atomic and deals them to its idle lanes by ballot
// and prefix count; it goes back for rays whenever `refill` lanes are idle.
// Waves per SIMD (the register budget the compiler is given), measured in Mrays/s on rays through 10 000 instances: the
// exact walk 1183 / 1253 / 1156 at 5 (what the compiler takes unasked) / 6 / 7; the ordered walk 1504 / 1747 / 1231 at
// 3 (unasked: 147 registers) / 4 / 5, and on the 10 M-triangle soup 490 / 534 / 383.
constexpr uint32_t kRayGrab = 1024;

template <bool COUNT, bool FAST = false, bool DEFER = false>
__global__ __launch_bounds__(kBlock, FAST ? 4 : 6) void trace_rays_persistent_kernel(DevScene S, const float* __restrict__ ro,
                                                                       const float* __restrict__ rd, uint64_t n,
                                                                       uint32_t* __restrict__ hits4,
                                                                       float* __restrict__ uv, float* __restrict__ normal,
                                                                       uint32_t* work_counter, uint32_t refill,
                                                                       uint64_t* counters, uint32_t inst_batch = 0) {
    extern __shared__ uint32_t fast_lds[];  // FAST: kFastLds entries per lane
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    Counters cnt = {0, 0, 0, 0};
    uint32_t rays = 0;
    uint64_t cur = 0, cur_end = 0;  // wave-uniform: the reserved rays not yet dealt
    bool exhausted = false;         // wave-uniform: the stream has no rays left to reserve
    bool active = false;
    uint64_t ray = 0;
    Walk<FAST> Wk;
    Wk.init(fast_lds);
    Wk.begin(S, mk(0.f, 0.f, 0.f), mk(0.f, 0.f, 1.f));

    for (;;) {
        uint64_t idle = __ballot(!active);
        while (idle != 0 && !exhausted) {
            if (cur >= cur_end) {
                uint32_t k = 0;
                if (lane == 0) k = atomicAdd(work_counter, 1u);
                k = __shfl(k, 0);
                cur = (uint64_t)k * kRayGrab;
                if (cur >= n) {
                    exhausted = true;
                    break;
                }
                cur_end = cur + kRayGrab < n ? cur + kRayGrab : n;
            }
            const uint64_t room = cur_end - cur;
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            const uint32_t take = room < (uint64_t)n_idle ? (uint32_t)room : n_idle;
            const uint32_t my_rank = (uint32_t)__popcll(idle & lt_mask);
            if (!active && my_rank < take) {
                ray = cur + my_rank;
                Wk.begin(S, mk(ro[3 * ray], ro[3 * ray + 1], ro[3 * ray + 2]),
                         mk(rd[3 * ray], rd[3 * ray + 1], rd[3 * ray + 2]));
                active = true;
                rays++;
            }
            cur += take;
            idle = __ballot(!active);
        }
        if (__ballot(active) == 0) break;  // nothing in flight and nothing left to reserve
        for (;;) {
            if (COUNT) {
                const uint64_t walking = __ballot(active);
                if ((int)lane == __ffsll((long long)walking) - 1) cnt.wave_steps++;
                if (active) cnt.lane_steps++;
            }
            if constexpr (DEFER) {  // lanes park at instances and enter them together (see cast_kernel)
                const uint64_t walking = __ballot(active);
                const uint64_t parked = __ballot(active && Wk.T.pend_id != 0);
                const bool enter_now = parked != 0 && ((uint32_t)__popcll(parked) >= inst_batch || (walking & ~parked) == 0);
                if (active) {
                    if (Wk.T.pend_id != 0) {
                        if (enter_now) Wk.template enter<COUNT>(S, cnt);
                    } else if (Wk.template step<COUNT, true>(S, cnt)) {
                        write_ray_result(S, ray, Wk.T.h, Wk.T.wo, Wk.T.wd, hits4, uv, normal);
                        active = false;
                    }
                }
            } else if (active && Wk.template step<COUNT, false>(S, cnt)) {
                write_ray_result(S, ray, Wk.T.h, Wk.T.wo, Wk.T.wd, hits4, uv, normal);
                active = false;
            }
            const uint32_t n_idle = (uint32_t)__popcll(__ballot(!active));
            if (n_idle == 64u || (!exhausted && n_idle >= refill)) break;
        }
    }
    work_slot_release(work_counter);
    if (COUNT) flush_counters(counters, cnt, rays);
}

template <bool COUNT, bool FAST = false>
__global__ __launch_bounds__(kBlock) void occlusion_kernel(DevScene S, const float* __restrict__ ro,
                                                           const float* __restrict__ rd, uint64_t n, float tmax,
                                                           uint8_t* __restrict__ out, uint64_t* counters) {
    extern __shared__ uint32_t fast_lds[];  // FAST: kFastLds entries per lane
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    Counters cnt = {0, 0, 0, 0};
    uint32_t rays = 0;
    if (i < n) {
        const V3 o = mk(ro[3 * i], ro[3 * i + 1], ro[3 * i + 2]);
        const V3 d = mk(rd[3 * i], rd[3 * i + 1], rd[3 * i + 2]);
        rays = 1;
        bool hit;
        if constexpr (FAST) hit = any_hit_query_fast<COUNT>(S, o, d, tmax, cnt, fast_lds);
        else hit = any_hit_query<COUNT>(S, o, d, tmax, cnt);
        out[i] = hit ? 1 : 0;
    }
    if (COUNT) flush_counters(counters, cnt, rays);
}

inline uint32_t div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

constexpr size_t kFastLdsBytes = (size_t)kFastLds * kBlock * sizeof(uint2);  // the ordered walk's LDS stack per workgroup

// A launch that asks for more than 64 KB of dynamic LDS (the ordered walk's stacks behind a bounce stack of more than
// 32 bounces) has to say so first.
template <class K>
int allow_lds(rthip_ctx* ctx, K kernel, size_t bytes) {
    if (bytes > (64u << 10)) RT_HIP(ctx, hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return RTHIP_OK;
}

}  // namespace

// How many workgroups of a persistent kernel the GPU holds at once (registers and LDS decide; asked of the runtime once
// per kernel and LDS size): a persistent launch is that many workgroups, never more than the work.
template <class K>
uint32_t resident_blocks(rthip_ctx* ctx, K kernel, size_t lds) {
    if (ctx->persistent_blocks_forced) return ctx->persistent_blocks;
    for (const rthip_ctx::Resident& r : ctx->resident)
        if (r.kernel == (const void*)kernel && r.lds == lds) return r.blocks;
    int per_cu = 0;
    uint32_t blocks = ctx->persistent_blocks;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kBlock, lds) == hipSuccess && per_cu >= 1) blocks = ctx->cus * (uint32_t)per_cu;
    else (void)hipGetLastError();
    ctx->resident.push_back({(const void*)kernel, lds, blocks});
    return blocks;
}

int work_slot_take(rthip_ctx* ctx, hipStream_t stream, uint32_t** out_work, int* out_slot) {
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (stream) (void)hipStreamIsCapturing(stream, &capturing);
    if (capturing != hipStreamCaptureStatusNone) {
        if (ctx->capture_next >= kCaptureSlots)
            return rthip_fail(ctx, RTHIP_E_LIMIT, "more than %u launches of a persistent kernel captured into HIP graphs since the last scene build", kCaptureSlots);
        *out_slot = -1;
        *out_work = (uint32_t*)ctx->work_ring.p + (size_t)(kEagerSlots + ctx->capture_next++) * 256u;
        return RTHIP_OK;
    }
    for (uint32_t tried = 0; tried < kEagerSlots; tried++) {
        const uint32_t k = (ctx->work_next + tried) % kEagerSlots;
        if (ctx->slot_used[k]) {
            const hipError_t e = hipEventQuery(ctx->slot_done[k]);
            if (e == hipErrorNotReady) continue;  // its launch has not finished: the slot's counters are live
            if (e != hipSuccess) return rthip_fail(ctx, RTHIP_E_HIP, "hipEventQuery failed: %s", hipGetErrorString(e));
            ctx->slot_used[k] = false;
        }
        ctx->work_next = (k + 1) % kEagerSlots;
        *out_slot = (int)k;
        *out_work = (uint32_t*)ctx->work_ring.p + (size_t)k * 256u;
        return RTHIP_OK;
    }
    return rthip_fail(ctx, RTHIP_E_LIMIT, "%u launches of a persistent kernel are unfinished: wait for some of them before enqueueing more", kEagerSlots);
}

int work_slot_launched(rthip_ctx* ctx, int slot, hipStream_t stream) {
    if (slot < 0) return RTHIP_OK;  // a captured launch keeps its slot
    if (!ctx->slot_done[slot]) RT_HIP(ctx, hipEventCreateWithFlags(&ctx->slot_done[slot], hipEventDisableTiming));
    RT_HIP(ctx, hipEventRecord(ctx->slot_done[slot], stream));
    ctx->slot_used[slot] = true;
    return RTHIP_OK;
}

int trace_rays_launch(rthip_ctx* ctx, const DevScene& scene, const float* d_o, const float* d_d, uint64_t n,
                      uint32_t* d_hits4, float* d_uv, float* d_normal, uint64_t* d_counters, hipStream_t stream) {
    if (n == 0) return RTHIP_OK;
    const bool fast = ctx->traversal == RTHIP_TRAV_FAST;
    // Long streams over a scene that the caches can hold: resident waves with lane refill.  Once the
    // scene is larger than the Infinity Cache the exact walk is bound by HBM line fetches (a 128-byte line per
    // 32-byte record: 0.8 of the peak), not by idle lanes, and the plain kernel (fewer registers, more waves) is the
    // faster one: 160 against 122 Mrays/s on the 10 M-triangle soup.  The ordered walk uses the whole line it fetches
    // and is short of requests in flight instead (27 % of the lanes at work in the plain kernel): resident with refill
    // 490 against 298 Mrays/s on the same rays.
    const uint64_t scene_bytes = ctx->total_blas_nodes * 32ull + ctx->total_tris * 48ull;
    const bool resident = ctx->cast_mode == 0 || (ctx->cast_mode != 1 && ctx->cast_mode != 4 && (fast || scene_bytes <= (256ull << 20)));
    if (n >= (1ull << 18) && n < (1ull << 32) * kRayGrab && resident) {
        uint32_t blocks = div_up(n, kBlock);
        // Lanes park at instances and enter them in batches (see cast_kernel).  Measured on rays through 10 000 instances:
        // exact walk 1106 / 1173 / 1183 / 1166 Mrays/s at batch 0 / 4 / 8 / 16, ordered walk 1326 / 1463 / 1504 / 1481.
        const uint32_t batch = fast ? (ctx->n_inst >= 8u ? ctx->fast_inst_batch : 0u) : (ctx->n_inst >= 64u ? ctx->rays_inst_batch : 0u);
        uint32_t held;
#define RT_HELD_RAYS_P(FAST, DEFER, LDS)                                                                                 \
    (d_counters ? resident_blocks(ctx, trace_rays_persistent_kernel<true, FAST, DEFER>, LDS)                            \
                : resident_blocks(ctx, trace_rays_persistent_kernel<false, FAST, DEFER>, LDS))
        if (fast) held = batch ? RT_HELD_RAYS_P(true, true, kFastLdsBytes) : RT_HELD_RAYS_P(true, false, kFastLdsBytes);
        else held = batch ? RT_HELD_RAYS_P(false, true, 0) : RT_HELD_RAYS_P(false, false, 0);
#undef RT_HELD_RAYS_P
        if (blocks > held) blocks = held;
        uint32_t* work = nullptr;  // a slot of zeroed counters: see work_slot_release
        int slot = -1;
        RT_TRY(work_slot_take(ctx, stream, &work, &slot));
#define RT_LAUNCH_RAYS_P(COUNT, FAST, DEFER, CNT)                                                                       \
    trace_rays_persistent_kernel<COUNT, FAST, DEFER><<<blocks, kBlock, (FAST) ? kFastLdsBytes : 0, stream>>>(           \
        scene, d_o, d_d, n, d_hits4, d_uv, d_normal, work, ctx->refill_lanes, CNT, batch)
        if (fast && batch) {
            if (d_counters) RT_LAUNCH_RAYS_P(true, true, true, d_counters);
            else RT_LAUNCH_RAYS_P(false, true, true, nullptr);
        } else if (fast) {
            if (d_counters) RT_LAUNCH_RAYS_P(true, true, false, d_counters);
            else RT_LAUNCH_RAYS_P(false, true, false, nullptr);
        } else if (batch) {
            if (d_counters) RT_LAUNCH_RAYS_P(true, false, true, d_counters);
            else RT_LAUNCH_RAYS_P(false, false, true, nullptr);
        } else {
            if (d_counters) RT_LAUNCH_RAYS_P(true, false, false, d_counters);
            else RT_LAUNCH_RAYS_P(false, false, false, nullptr);
        }
#undef RT_LAUNCH_RAYS_P
        RT_HIP(ctx, hipGetLastError());
        return work_slot_launched(ctx, slot, stream);
    }
    const uint32_t blocks = div_up(n, kBlock);
#define RT_LAUNCH_RAYS(COUNT, FAST, CNT)                                                                                \
    trace_rays_kernel<COUNT, FAST><<<blocks, kBlock, (FAST) ? kFastLdsBytes : 0, stream>>>(scene, d_o, d_d, n, d_hits4, d_uv, \
                                                                                          d_normal, CNT)
    if (fast) {
        if (d_counters) RT_LAUNCH_RAYS(true, true, d_counters);
        else RT_LAUNCH_RAYS(false, true, nullptr);
    } else {
        if (d_counters) RT_LAUNCH_RAYS(true, false, d_counters);
        else RT_LAUNCH_RAYS(false, false, nullptr);
    }
#undef RT_LAUNCH_RAYS
    RT_HIP(ctx, hipGetLastError());
    return RTHIP_OK;
}

int occlusion_launch(rthip_ctx* ctx, const DevScene& scene, const float* d_o, const float* d_d, uint64_t n, float tmax,
                     uint8_t* d_out, uint64_t* d_counters, hipStream_t stream) {
    if (n == 0) return RTHIP_OK;
    const uint32_t blocks = div_up(n, kBlock);
    if (ctx->traversal == RTHIP_TRAV_FAST) {
        if (d_counters)
            occlusion_kernel<true, true><<<blocks, kBlock, kFastLdsBytes, stream>>>(scene, d_o, d_d, n, tmax, d_out, d_counters);
        else
            occlusion_kernel<false, true><<<blocks, kBlock, kFastLdsBytes, stream>>>(scene, d_o, d_d, n, tmax, d_out, nullptr);
    } else if (d_counters)
        occlusion_kernel<true><<<blocks, kBlock, 0, stream>>>(scene, d_o, d_d, n, tmax, d_out, d_counters);
    else
        occlusion_kernel<false><<<blocks, kBlock, 0, stream>>>(scene, d_o, d_d, n, tmax, d_out, nullptr);
    RT_HIP(ctx, hipGetLastError());
    return RTHIP_OK;
}

int primary_hits_launch(rthip_ctx* ctx, const DevScene& scene, const rthip_cast_settings& cast, int width, int height,
                        uint32_t* d_hits4, uint64_t* d_counters, hipStream_t stream) {
    const uint32_t blocks = div_up(width, 16) * div_up(height, 16);
    if (ctx->traversal == RTHIP_TRAV_FAST) {
        if (d_counters)
            primary_hits_kernel<true, true><<<blocks, kBlock, kFastLdsBytes, stream>>>(scene, cast, width, height, d_hits4, d_counters);
        else
            primary_hits_kernel<false, true><<<blocks, kBlock, kFastLdsBytes, stream>>>(scene, cast, width, height, d_hits4, nullptr);
    } else if (d_counters)
        primary_hits_kernel<true><<<blocks, kBlock, 0, stream>>>(scene, cast, width, height, d_hits4, d_counters);
    else
        primary_hits_kernel<false><<<blocks, kBlock, 0, stream>>>(scene, cast, width, height, d_hits4, nullptr);
    RT_HIP(ctx, hipGetLastError());
    return RTHIP_OK;
}

int cast_records_launch(rthip_ctx* ctx, const DevScene& scene, const rthip_cast_settings& cast, uint64_t seed, int width,
                        int height, rthip_ray_record* d_slots, uint32_t* d_counts, rthip_ray_record* d_out,
                        uint64_t* out_total, hipStream_t stream) {
    const uint64_t n_paths = (uint64_t)width * height * cast.samples * cast.samples;
    const size_t lds = (size_t)(scene.max_bounces ? scene.max_bounces : 1) * kBlock * sizeof(uint32_t);
    const uint32_t blocks = div_up(n_paths, kBlock);
    if (ctx->traversal == RTHIP_TRAV_FAST) RT_TRY(allow_lds(ctx, cast_records_kernel<true>, lds + kFastLdsBytes));
    if (ctx->traversal == RTHIP_TRAV_FAST)
        cast_records_kernel<true><<<blocks, kBlock, lds + kFastLdsBytes, stream>>>(scene, cast, seed, width, height, d_slots, d_counts);
    else
        cast_records_kernel<false><<<blocks, kBlock, lds, stream>>>(scene, cast, seed, width, height, d_slots, d_counts);
    RT_HIP(ctx, hipGetLastError());
    RT_TRY(scan_inclusive_u32_device(ctx, d_counts, n_paths, stream));
    uint32_t total = 0;
    RT_HIP(ctx, hipMemcpyAsync(&total, d_counts + (n_paths - 1), 4, hipMemcpyDeviceToHost, stream));
    RT_HIP(ctx, hipStreamSynchronize(stream));
    *out_total = total;
    if (d_out) {
        records_compact_kernel<<<blocks, kBlock, 0, stream>>>(d_slots, d_counts, n_paths, scene.max_bounces, d_out);
        RT_HIP(ctx, hipGetLastError());
    }
    return RTHIP_OK;
}

namespace {

int launch_static(rthip_ctx* ctx, const DevScene& scene, const rthip_cast_settings& cast, uint64_t seed, float* d_rgb,
                  int width, int height, rthip_shard shard, uint64_t* d_counters, hipStream_t stream, uint32_t ib = 0) {
    const size_t lds = (size_t)(scene.max_bounces ? scene.max_bounces : 1) * kBlock * sizeof(uint32_t);
    const uint32_t blocks = div_up(width, 16) * div_up(height, 16);
    if (ctx->traversal == RTHIP_TRAV_FAST) {
        if (d_counters) RT_TRY(allow_lds(ctx, cast_static_kernel<true, false, true>, lds + kFastLdsBytes));
        else RT_TRY(allow_lds(ctx, cast_static_kernel<false, false, true>, lds + kFastLdsBytes));
        if (d_counters)
            cast_static_kernel<true, false, true><<<blocks, kBlock, lds + kFastLdsBytes, stream>>>(scene, cast, seed, d_rgb, width, height, shard, ctx->cast_accumulate, d_counters);
        else
            cast_static_kernel<false, false, true><<<blocks, kBlock, lds + kFastLdsBytes, stream>>>(scene, cast, seed, d_rgb, width, height, shard, ctx->cast_accumulate, nullptr);
        RT_HIP(ctx, hipGetLastError());
        return RTHIP_OK;
    }
    // ib != 0: the instance entries of a wave run in batches (closest_hit_query_batched; entry_batch decides)
    if (ib) {
        if (d_counters)
            cast_static_kernel<true, true><<<blocks, kBlock, lds, stream>>>(scene, cast, seed, d_rgb, width, height, shard, ctx->cast_accumulate, d_counters, ib);
        else
            cast_static_kernel<false, true><<<blocks, kBlock, lds, stream>>>(scene, cast, seed, d_rgb, width, height, shard, ctx->cast_accumulate, nullptr, ib);
    } else if (d_counters)
        cast_static_kernel<true><<<blocks, kBlock, lds, stream>>>(scene, cast, seed, d_rgb, width, height, shard, ctx->cast_accumulate, d_counters);
    else
        cast_static_kernel<false><<<blocks, kBlock, lds, stream>>>(scene, cast, seed, d_rgb, width, height, shard, ctx->cast_accumulate, nullptr);
    RT_HIP(ctx, hipGetLastError());
    return RTHIP_OK;
}

// samples = 2, 4 or 8 only (4, 16 or 64 lanes per pixel)
int launch_samples(rthip_ctx* ctx, const DevScene& scene, const rthip_cast_settings& cast, uint64_t seed, float* d_rgb,
                   int width, int height, rthip_shard shard, uint64_t* d_counters, hipStream_t stream, uint32_t ib = 0) {
    const size_t lds = (size_t)(scene.max_bounces ? scene.max_bounces : 1) * kBlock * sizeof(uint32_t);
    const uint32_t log2_group = cast.samples == 2 ? 2u : (cast.samples == 4 ? 4u : 6u);
    const uint32_t tile = log2_group == 2 ? 8u : (log2_group == 4 ? 4u : 2u);  // pixels a workgroup covers, per side
    const uint32_t blocks = div_up(width, tile) * div_up(height, tile);
    // ib != 0: the instance entries of a wave run in batches (closest_hit_query_batched; entry_batch decides)
#define RT_LAUNCH_SAMPLES(COUNT, BATCH, CNT)                                                                               \
    cast_samples_kernel<COUNT, BATCH><<<blocks, kBlock, lds, stream>>>(scene, cast, seed, d_rgb, width, height, shard, log2_group, \
                                                                       ctx->cast_accumulate, CNT, ib)
    if (ctx->traversal == RTHIP_TRAV_FAST) {
        if (d_counters) RT_TRY(allow_lds(ctx, cast_samples_kernel<true, false, true>, lds + kFastLdsBytes));
        else RT_TRY(allow_lds(ctx, cast_samples_kernel<false, false, true>, lds + kFastLdsBytes));
        if (d_counters)
            cast_samples_kernel<true, false, true><<<blocks, kBlock, lds + kFastLdsBytes, stream>>>(scene, cast, seed, d_rgb, width, height, shard, log2_group, ctx->cast_accumulate, d_counters, 0u);
        else
            cast_samples_kernel<false, false, true><<<blocks, kBlock, lds + kFastLdsBytes, stream>>>(scene, cast, seed, d_rgb, width, height, shard, log2_group, ctx->cast_accumulate, nullptr, 0u);
    } else if (d_counters) {
        if (ib) RT_LAUNCH_SAMPLES(true, true, d_counters);
        else RT_LAUNCH_SAMPLES(true, false, d_counters);
    } else {
        if (ib) RT_LAUNCH_SAMPLES(false, true, nullptr);
        else RT_LAUNCH_SAMPLES(false, false, nullptr);
    }
#undef RT_LAUNCH_SAMPLES
    RT_HIP(ctx, hipGetLastError());
    return RTHIP_OK;
}

int launch_persistent(rthip_ctx* ctx, const DevScene& scene, const rthip_cast_settings& cast, uint64_t seed, float* d_rgb,
                      int width, int height, rthip_shard shard, uint32_t refill, uint32_t inst_batch, uint64_t* d_counters,
                      hipStream_t stream) {
    // persistent grid: as many workgroups as are resident at once (resident_blocks), never more than the work
    const uint64_t total = (uint64_t)div_up(width, 8) * div_up(height, 8) * 64ull;
    uint32_t blocks = div_up(total, kBlock);
    const size_t lds = (size_t)(scene.max_bounces ? scene.max_bounces : 1) * kBlock * sizeof(uint32_t);
    const bool fast_walk = ctx->traversal == RTHIP_TRAV_FAST, batched = inst_batch != 0;
    uint32_t held;
    if (fast_walk && batched) held = d_counters ? resident_blocks(ctx, cast_kernel<true, true, true>, lds + kFastLdsBytes) : resident_blocks(ctx, cast_kernel<false, true, true>, lds + kFastLdsBytes);
    else if (fast_walk) held = d_counters ? resident_blocks(ctx, cast_kernel<true, false, true>, lds + kFastLdsBytes) : resident_blocks(ctx, cast_kernel<false, false, true>, lds + kFastLdsBytes);
    else if (batched) held = d_counters ? resident_blocks(ctx, cast_kernel<true, true>, lds) : resident_blocks(ctx, cast_kernel<false, true>, lds);
    else held = d_counters ? resident_blocks(ctx, cast_kernel<true, false>, lds) : resident_blocks(ctx, cast_kernel<false, false>, lds);
    if (blocks > held) blocks = held;
    // every launch gets its own set of 8 work counters (one 128-byte line each), so that casts enqueued back to back or
    // on several streams do not share them; a slot is taken only once its previous launch has finished, and is zero
    // when a launch starts (the ring is zeroed once, and every launch leaves its slot zeroed: work_slot_release)
    uint32_t* work = nullptr;
    int slot = -1;
    RT_TRY(work_slot_take(ctx, stream, &work, &slot));
    const bool defer = inst_batch != 0;
#define RT_LAUNCH_CAST(COUNT, DEFER, CNT)                                                                              \
    cast_kernel<COUNT, DEFER><<<blocks, kBlock, lds, stream>>>(scene, cast, seed, d_rgb, width, height, shard, work, \
                                                               refill, inst_batch, ctx->cast_accumulate, CNT)
#define RT_LAUNCH_CAST_FAST(COUNT, DEFER, CNT)                                                                         \
    do {                                                                                                               \
        RT_TRY(allow_lds(ctx, cast_kernel<COUNT, DEFER, true>, lds + kFastLdsBytes));                                  \
        cast_kernel<COUNT, DEFER, true><<<blocks, kBlock, lds + kFastLdsBytes, stream>>>(                              \
            scene, cast, seed, d_rgb, width, height, shard, work, refill, inst_batch, ctx->cast_accumulate, CNT);      \
    } while (0)
    if (ctx->traversal == RTHIP_TRAV_FAST) {
        if (d_counters) {
            if (defer) RT_LAUNCH_CAST_FAST(true, true, d_counters);
            else RT_LAUNCH_CAST_FAST(true, false, d_counters);
        } else {
            if (defer) RT_LAUNCH_CAST_FAST(false, true, nullptr);
            else RT_LAUNCH_CAST_FAST(false, false, nullptr);
        }
    } else if (d_counters) {
        if (defer) RT_LAUNCH_CAST(true, true, d_counters);
        else RT_LAUNCH_CAST(true, false, d_counters);
    } else {
        if (defer) RT_LAUNCH_CAST(false, true, nullptr);
        else RT_LAUNCH_CAST(false, false, nullptr);
    }
#undef RT_LAUNCH_CAST
#undef RT_LAUNCH_CAST_FAST
    RT_HIP(ctx, hipGetLastError());
    return work_slot_launched(ctx, slot, stream);
}

struct Candidate {
    int kernel;           // RTHIP_KERNEL_STATIC or RTHIP_KERNEL_PERSISTENT
    uint32_t refill;      // persistent only
    uint32_t inst_batch;  // persistent only: 0 = enter instances at once, else park and enter in batches
};

// Which renderer suits a scene and camera depends on how long and how unequal the paths of one wave
// are, which only running some of them tells.  So before the first frame of a scene (per frame
// geometry) a probe renders a sparse sample of 64x64 tiles (1 in `stride`, one sample per pixel) with
// the pixel-per-thread kernel and its traversal counters on, into the caller's framebuffer, which the
// real render overwrites afterwards.  Two numbers decide:
//   lane utilisation = box tests / (64 x wave steps): below 0.28 most lanes of a wave wait for its
//     longest path, and the persistent kernel's refill pays for its bookkeeping;
//   box tests per ray: short walks want more lanes gathered before shading (refill 32), long ones
//     want idle lanes back in the walk quickly (refill 16);
//   instance entries per box test: above 2 % the long entry branch runs in most iterations for two or
//     three lanes, and parking lanes to enter in batches of 8 pays (+10 % on Cornell + mesh; -5 % where
//     entries are rare).
// The thresholds sit between the measured workloads (Cornell 0.33 / 18 keeps the static kernel; 10M-triangle
// soup 0.30 / 145, 10k instances 0.24 / 146, Cornell + mesh at 4K 0.16 / 22 are 25-45 % faster on the
// persistent one).  The counters are deterministic, so the choice is the same on every run.  The probe
// waits on the stream, so it is skipped while the stream is being captured.
// The counted probe: a sparse sample of 64x64 tiles, one sample per pixel, pixel-per-thread kernel, plain walk.
// h = {rays, box tests, triangle tests, instance entries, wave steps}.  Renders into d_rgb (or, for an accumulating
// cast, into a buffer of its own) and waits on the stream.
int probe_counters(rthip_ctx* ctx, const DevScene& scene, const rthip_cast_settings& cast, uint64_t seed, float* d_rgb, int width,
                   int height, rthip_shard shard, hipStream_t stream, uint64_t h[8], uint32_t* stride_out) {
    float* probe_rgb = d_rgb;
    const uint32_t accumulate = ctx->cast_accumulate;
    if (accumulate) {
        RT_TRY(rthip_reserve(ctx, ctx->tune_frame, (size_t)width * height * 3 * sizeof(float)));
        probe_rgb = (float*)ctx->tune_frame.p;
    }
    rthip_cast_settings pc = cast;
    pc.samples = 1;
    const uint32_t tiles64 = div_up(width, 64) * div_up(height, 64);
    uint32_t stride = tiles64 / 32u;  // about 32 tiles = 131k paths
    if (stride < 1) stride = 1;
    rthip_shard probe = {1u, 0u, stride};
    if (shard.mode == 1) {  // stay inside this rank's own tiles
        probe.world = shard.world * stride;
        probe.rank = shard.rank;
    }
    RT_TRY(rthip_reserve(ctx, ctx->tune_counters, 64));
    uint64_t* d_cnt = (uint64_t*)ctx->tune_counters.p;
    for (int k = 0; k < 8; k++) h[k] = 0;
    RT_HIP(ctx, hipMemsetAsync(d_cnt, 0, 64, stream));
    ctx->cast_accumulate = 0;
    const int rc = launch_static(ctx, scene, pc, seed, probe_rgb, width, height, probe, d_cnt, stream);
    ctx->cast_accumulate = accumulate;
    RT_TRY(rc);
    RT_HIP(ctx, hipMemcpyAsync(h, d_cnt, 64, hipMemcpyDeviceToHost, stream));
    RT_HIP(ctx, hipStreamSynchronize(stream));
    if (stride_out) *stride_out = stride;
    return RTHIP_OK;
}

int tune_cast(rthip_ctx* ctx, const DevScene& scene, const rthip_cast_settings& cast, uint64_t seed, float* d_rgb, int width,
              int height, rthip_shard shard, hipStream_t stream, Candidate* best) {
    uint64_t h[8];
    uint32_t stride = 1;
    RT_TRY(probe_counters(ctx, scene, cast, seed, d_rgb, width, height, shard, stream, h, &stride));
    ctx->probed_entry_ratio = h[1] ? (double)h[3] / (double)h[1] : 0.0;
    const double rays = (double)h[0], nodes = (double)h[1], entries = (double)h[3], steps = (double)h[4];
    const double util = steps > 0 ? nodes / (64.0 * steps) : 1.0;
    const double nodes_per_ray = rays > 0 ? nodes / rays : 0.0;
    // deep scenes sit further from the border: their walks are long-tailed even where the sample looks even
    const bool deep = ctx->n_inst >= 64u || ctx->total_tris >= (1ull << 20);
    if (util < (deep ? 0.45 : 0.28)) {
        best->kernel = 0;
        // (Cornell + mesh at 3840x2160, 20 box tests per ray: 4553 / 4986 / 5134 / 5108 / 4465 Mrays/s at 24 / 32 / 40 / 48 / 56)
        best->refill = nodes_per_ray < 64.0 ? 40u : 16u;
        // (10 000 instances, entries 0.8 % of the box tests: 1146 / 1183 / 1164 / 1130 Mrays/s at batch 0 / 4 / 8 / 16)
        best->inst_batch = (nodes > 0 && entries / nodes > 0.02) ? 8u : ((nodes > 0 && entries / nodes > 0.005) ? 4u : 0u);
    } else {
        best->kernel = 1;
        best->refill = 0;
        best->inst_batch = 0;
    }
    if (getenv("RTHIP_TUNE_LOG"))
        fprintf(stderr, "rthip tune: 1 tile in %u, %.0f rays, %.1f box tests per ray, lane utilisation %.3f -> kernel %d refill %u instance batch %u\n",
                stride, rays, nodes_per_ray, util, best->kernel, best->refill, best->inst_batch);
    return RTHIP_OK;
}

// Whether the pixel- and sample-per-lane kernels run a wave's instance entries in batches (closest_hit_query_batched).
// It pays where rays enter an instance every few box tests: measured over grids of 16 ... 10 000 instances of quads, boxes
// and icosahedra and the Cornell box (tools/probe_entries.py), +10 % where the entries are 4.1 % and 4.8 % of the box
// tests (16 boxes, Cornell) and -6 ... -40 % where they are 3.2 % or less.  Only scenes whose meshes are a handful of
// triangles can get there (rthip_tlas_build marks them), and for those the counted probe measures the share once per
// scene, frame size and sharding.  RTHIP_SAMPLES_INST_BATCH overrides.  Skipped (no batches) while the stream is captured.
constexpr double kEntryBatchRatio = 0.04;

int entry_batch(rthip_ctx* ctx, const DevScene& scene, const rthip_cast_settings& cast, uint64_t seed, float* d_rgb, int width,
                int height, rthip_shard shard, hipStream_t stream, uint32_t* batch) {
    *batch = 0;
    if (ctx->samples_inst_batch_env >= 0) {
        *batch = (uint32_t)ctx->samples_inst_batch_env;
        return RTHIP_OK;
    }
    // a kernel selected by name runs as named: only RTHIP_KERNEL_AUTO probes (the probe renders and waits on the stream)
    if (!ctx->small_meshes || ctx->cast_mode != 3) return RTHIP_OK;
    const uint64_t key[4] = {ctx->scene_generation, ((uint64_t)(uint32_t)width << 32) | (uint32_t)height,
                             (uint64_t)cast.samples * cast.samples, ((uint64_t)shard.mode << 32) | shard.world};
    if (!(ctx->batch_known && memcmp(key, ctx->batch_key, sizeof key) == 0)) {
        hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
        if (stream) (void)hipStreamIsCapturing(stream, &capturing);
        if (capturing != hipStreamCaptureStatusNone) return RTHIP_OK;
        uint64_t h[8];
        RT_TRY(probe_counters(ctx, scene, cast, seed, d_rgb, width, height, shard, stream, h, nullptr));
        const double ratio = h[1] ? (double)h[3] / (double)h[1] : 0.0;
        ctx->batch_known = true;
        memcpy(ctx->batch_key, key, sizeof key);
        ctx->batch_value = ratio >= kEntryBatchRatio ? 64u : 0u;
        if (getenv("RTHIP_TUNE_LOG"))
            fprintf(stderr, "rthip tune: instance entries are %.1f %% of the box tests -> entry batch %u\n", 100.0 * ratio, ctx->batch_value);
    }
    *batch = ctx->batch_value;
    return RTHIP_OK;
}

}  // namespace

int cast_launch(rthip_ctx* ctx, const DevScene& scene, const rthip_cast_settings& cast, uint64_t seed, float* d_rgb,
                int width, int height, rthip_shard shard, uint64_t* d_counters, hipStream_t stream) {
    ctx->last_cast_kernel = 2;
    const bool fast = ctx->traversal == RTHIP_TRAV_FAST;
    if (ctx->cast_mode == 2 && !fast)
        return cast_wavefront_launch(ctx, scene, cast, seed, d_rgb, width, height, shard, d_counters, stream);
    // the wavefront renderer has no ordered walk: in that mode its name selects the persistent kernel
    Candidate choice = {ctx->cast_mode == 2 ? 0 : ctx->cast_mode, ctx->refill_lanes, ctx->inst_batch};
    // One lane per sample needs samples^2 = 4, 16 or 64.  Measured against the other two kernels over frame sizes
    // (tools/probe_samples.py): never slower than the pixel-per-thread kernel (its waves are shorter and the samples of
    // a pixel walk the same part of the tree), 1.2x (Cornell 512^2) to 7x (256^2 frames of deep scenes) faster on
    // small frames, and faster than the persistent kernel up to about 1024^2 pixels on deep scenes and 1920x1080 on
    // shallow ones; beyond that the persistent kernel's lane refill wins where the probe says so.
    const bool by_samples_possible = cast.samples == 2 || cast.samples == 4 || cast.samples == 8;
    const bool deep = ctx->n_inst >= 64u || ctx->total_tris >= (1ull << 20);
    const uint64_t pixel_waves = (uint64_t)div_up(width, 8) * div_up(height, 8);
    if (choice.kernel == 4 && !by_samples_possible) choice.kernel = 1;
    if (ctx->cast_mode == 3 && fast) {
        // No probe for the ordered walk.  Measured at 1920x1080 (Mrays/s, four streams; persistent with refill 16 against
        // the pixel- / sample-per-lane kernels): 10 000 instances 1329 vs 1031, 10 M-triangle soup 1509 vs 1374, height
        // field 1340 vs 1249 -- deep scenes keep their long-tailed walks in this mode too -- but bunny-class 6308 vs 7498
        // (one stream): short paths that are alike want the least bookkeeping per ray.
        const uint64_t paths = (uint64_t)width * (uint64_t)height * cast.samples * cast.samples;
        if (deep && paths >= (1ull << 20)) {
            // (refill threshold, Mrays/s at 8 / 16 / 24 / 32 / 48 idle lanes: 10 000 instances 1354 / 1456 / 1490 / 1445 / 1496,
            // height field 1165 / 1383 / 1479 / 1514 / 1560, soup 1581 / 1628 / 1613 / 1581 / 1489: a step of this walk is long,
            // so the ray generation and shading between two runs of steps weigh less per ray when more lanes share them)
            choice.kernel = 0;
            choice.refill = 24;
            choice.inst_batch = 0;
        } else {
            choice.kernel = by_samples_possible ? 4 : 1;
        }
    } else if (ctx->cast_mode == 3 && by_samples_possible && pixel_waves < (deep ? 20480ull : 40000ull)) {
        choice.kernel = 4;
    } else if (ctx->cast_mode == 3) {
        const uint64_t spp = (uint64_t)cast.samples * cast.samples;
        const uint64_t paths = (uint64_t)width * (uint64_t)height * spp;
        const uint64_t key[4] = {ctx->scene_generation, ((uint64_t)(uint32_t)width << 32) | (uint32_t)height, spp,
                                 ((uint64_t)shard.mode << 32) | shard.world};
        hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
        if (stream) (void)hipStreamIsCapturing(stream, &capturing);
        if (ctx->tuned && memcmp(key, ctx->tuned_key, sizeof key) == 0) {
            choice = {ctx->tuned_kernel, ctx->tuned_refill, ctx->tuned_inst_batch};
        } else if (paths >= (1ull << 20) && capturing == hipStreamCaptureStatusNone) {
            RT_TRY(tune_cast(ctx, scene, cast, seed, d_rgb, width, height, shard, stream, &choice));
            ctx->batch_known = true;  // the probe has counted the instance entries as well
            memcpy(ctx->batch_key, key, sizeof key);
            ctx->batch_value = ctx->probed_entry_ratio >= kEntryBatchRatio ? 64u : 0u;
            ctx->tuned = true;
            memcpy(ctx->tuned_key, key, sizeof key);
            ctx->tuned_kernel = choice.kernel;
            ctx->tuned_refill = choice.refill;
            ctx->tuned_inst_batch = choice.inst_batch;
        } else {
            // Small frames are not worth a probe.  Paths through a small scene are short and alike, and the
            // pixel-per-thread kernel has the least bookkeeping per ray; deep scenes (many instances, large
            // meshes) have long-tailed walks that pay for the persistent kernel's lane refill.
            choice.kernel = deep ? 0 : 1;
            choice.refill = ctx->refill_lanes;
            choice.inst_batch = 0;
        }
        if (choice.kernel == 1 && by_samples_possible) choice.kernel = 4;
    }
    // The ordered walk parks lanes at instances and enters them in batches wherever a scene has instances to speak of: a
    // step of that walk is long already (four box tests, a sort, the stack), and with a few per cent of the lane steps
    // ending in an instance nearly every step of a wave pays for the entry as well (RTHIP_FAST_INST_BATCH; 0 = never).
    if (fast) choice.inst_batch = ctx->n_inst >= 8u ? ctx->fast_inst_batch : 0u;
    ctx->last_cast_kernel = (uint32_t)choice.kernel;
    ctx->last_cast_refill = choice.kernel == 0 ? choice.refill : 0u;
    uint32_t ib = 0;
    if (!fast && (choice.kernel == 4 || choice.kernel == 1)) RT_TRY(entry_batch(ctx, scene, cast, seed, d_rgb, width, height, shard, stream, &ib));
    if (choice.kernel == 4) return launch_samples(ctx, scene, cast, seed, d_rgb, width, height, shard, d_counters, stream, ib);
    if (choice.kernel == 1) return launch_static(ctx, scene, cast, seed, d_rgb, width, height, shard, d_counters, stream, ib);
    return launch_persistent(ctx, scene, cast, seed, d_rgb, width, height, shard, choice.refill, choice.inst_batch, d_counters,
                             stream);
}

#ifdef RT_STAMP
extern "C" int rthip_debug_stamps(uint64_t* host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), n * sizeof(uint64_t));
}
#endif
