// LBVH construction on gfx950: Morton codes, LDS-staged wave64 radix sort, bottom-up
// hierarchy emission with in-flight box refit, pre-order layout.
//
// The tree produced here is, node for node, the tree that the reference's top-down
// builder produces (src/lbvh/lbvh.c:15-125): a radix tree over the distinct Morton
// codes in which every run of equal codes is a subtree split at count/2
// (lbvh_get_split_idx, lbvh.c:22-24).  How that is reached without recursion:
//
//   * every inner node of a binary tree over a sorted leaf sequence owns exactly one
//     "gap" g (between sorted leaves g and g+1): the place where it splits its range;
//   * give every gap a priority: the number of leading bits its two codes share when
//     they differ (lbvh.c:25), or 64 + (depth inside the midpoint-split tree of its
//     run) when they are equal.  The reference tree is the Cartesian tree of those
//     priorities: a node covers the widest range around its gap in which every other
//     gap has a larger priority;
//   * so each subtree can find its parent locally: of the two gaps that bound its leaf
//     range, the one with the larger priority (bottom-up walk, one thread per leaf, the
//     second child to arrive at a node carries on);
//   * pre-order index of a node = 2 * (its first leaf) + (the number of its ancestors that
//     hold it in their left subtree); for a leaf that number comes from a scan over the gap
//     priorities, an inner node sits directly in front of its left child; skip = index +
//     subtree size.
//
// Floating point: compiled with -ffp-contract=off; min/max are the reference's
// ternaries (common_core.h:113-114) so that ties and signed zeros resolve the same way.

#include "rthip_internal.h"

namespace {

constexpr int kBlock = 256;
constexpr int kSortBlock = 1024;                    // threads per workgroup in the radix kernels
constexpr int kSortItems = 4;                       // keys per thread
constexpr int kSortTile = kSortBlock * kSortItems;  // 4096 keys per workgroup
constexpr int kScanItems = 8;
constexpr int kScanTile = kBlock * kScanItems;  // 2048 values per workgroup

struct alignas(16) ChildRec {  // == ChildInfo: what a child leaves at its parent
    float box[6];
    uint32_t far_end;  // the end of the child's leaf range that is not the parent's gap
    uint32_t lefts;    // low half: how many of the child's ancestors hold it in their LEFT subtree (its pre-order index
                       // is 2 * first leaf + that); upper half, in records that go through HBM: a gap priority (ClimbArgs)
};
static_assert(sizeof(ChildRec) == 32, "ChildRec must be 32 bytes");

__device__ __forceinline__ float ref_min(float a, float b) { return (a < b) ? a : b; }  // Min(A,B)
__device__ __forceinline__ float ref_max(float a, float b) { return (a > b) ? a : b; }  // Max(A,B)

// order-preserving map float -> uint32 for atomicMin / atomicMax
__device__ __forceinline__ uint32_t f2ord(float f) {
    uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
    uint32_t b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return __uint_as_float(b);
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Centroid bounds are gathered in kBoundSlots copies, 4 KB + 128 bytes apart (whatever the address bits that pick an L2
// channel, neighbouring slots differ in them), and folded by their reader
// (bounds_fold): atomics on one address are served one after the other by one L2 channel, a few nanoseconds each, and
// with every workgroup of a grid sending its six to the same six words that queue was most of the triangle pass at
// small and middle sizes (12 us of it at 81 920 triangles, 75 us at 10^6).
constexpr int kBoundSlots = 16;
constexpr int kBoundStride = 1056;  // uint32 from slot to slot
constexpr size_t kBoundBytes = (size_t)kBoundSlots * kBoundStride * 4 + 256;  // + a flag word of the sort (lbvh_build_device)

__device__ __forceinline__ void bounds_fold(const uint32_t* __restrict__ bounds, uint32_t* sb /* LDS, 6 words */) {
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        uint32_t v = bounds[k];
        for (int slot = 1; slot < kBoundSlots; slot++) {
            const uint32_t o = bounds[slot * kBoundStride + k];
            v = (k < 3) ? (o < v ? o : v) : (o > v ? o : v);
        }
        sb[k] = v;
    }
    __syncthreads();
}

// order-preserving reduction of per-thread centroid bounds into the workgroup's slot (6 atomics per workgroup)
__device__ __forceinline__ void block_bounds_reduce(const float* mn, const float* mx, uint32_t* __restrict__ bounds) {
    __shared__ float red[6][kBlock / 64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float a = wave_min(mn[k]), b = wave_max(mx[k]);
        if (lane == 0) {
            red[k][wave] = a;
            red[k + 3][wave] = b;
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        float v = red[k][0];
        for (int w = 1; w < kBlock / 64; w++) v = (k < 3) ? fminf(v, red[k][w]) : fmaxf(v, red[k][w]);
        uint32_t* slot = bounds + (blockIdx.x % kBoundSlots) * kBoundStride;
        if (k < 3) atomicMin(&slot[k], f2ord(v));
        else atomicMax(&slot[k], f2ord(v));
    }
}

// ---- K1: triangle boxes + per-triangle vertex records ----------------------------
// rt_cpu_aabb_from_tri and its loops, src/raytracer/cpu/raytracer_cpu.c:54-87, fused with the
// centroid bounds of lbvh_make's first loop (lbvh.c:95-105) so the boxes are not read again
// for them.  Grid-stride, so that the bounds cost 6 atomics per workgroup of a bounded grid (see kBoundSlots).
__global__ __launch_bounds__(kBlock) void tri_prepare_kernel(const float* __restrict__ px, const float* __restrict__ py,
                                                             const float* __restrict__ pz, const float* __restrict__ nx,
                                                             const float* __restrict__ ny, const float* __restrict__ nz,
                                                             const uint32_t* __restrict__ indices, uint64_t n_tris,
                                                             float* __restrict__ aabb6, float4* __restrict__ tri_v,
                                                             float4* __restrict__ tri_n, uint32_t* __restrict__ bounds,
                                                             uint32_t box_bounds) {
    float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
    float mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    for (uint64_t t = (uint64_t)blockIdx.x * kBlock + threadIdx.x; t < n_tris; t += (uint64_t)gridDim.x * kBlock) {
        uint32_t i0, i1, i2;
        if (indices) {
            i0 = indices[3 * t];
            i1 = indices[3 * t + 1];
            i2 = indices[3 * t + 2];
        } else {
            i0 = (uint32_t)(3 * t);
            i1 = i0 + 1;
            i2 = i0 + 2;
        }
        float ax = px[i0], ay = py[i0], az = pz[i0];
        float bx = px[i1], by = py[i1], bz = pz[i1];
        float cx = px[i2], cy = py[i2], cz = pz[i2];
        float b[6];
        b[0] = ref_min(ref_min(ax, bx), cx);
        b[1] = ref_min(ref_min(ay, by), cy);
        b[2] = ref_min(ref_min(az, bz), cz);
        b[3] = ref_max(ref_max(ax, bx), cx);
        b[4] = ref_max(ref_max(ay, by), cy);
        b[5] = ref_max(ref_max(az, bz), cz);
        float* o = aabb6 + 6 * t;
#pragma unroll
        for (int k = 0; k < 6; k++) o[k] = b[k];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float c = (b[k] + b[k + 3]) * 0.5f;  // mul_3f32(add_3f32(min, max), 0.5f), lbvh.c:98
            mn[k] = ref_min(mn[k], box_bounds ? b[k] : c);      // a BUILD_DEBUG build bounds the boxes, lbvh.c:101-104
            mx[k] = ref_max(mx[k], box_bounds ? b[k + 3] : c);
        }
        if (tri_v) {
            tri_v[3 * t + 0] = make_float4(ax, ay, az, 0.f);
            tri_v[3 * t + 1] = make_float4(bx, by, bz, 0.f);
            tri_v[3 * t + 2] = make_float4(cx, cy, cz, 0.f);
        }
        if (tri_n && nx) {
            tri_n[3 * t + 0] = make_float4(nx[i0], ny[i0], nz[i0], 0.f);
            tri_n[3 * t + 1] = make_float4(nx[i1], ny[i1], nz[i1], 0.f);
            tri_n[3 * t + 2] = make_float4(nx[i2], ny[i2], nz[i2], 0.f);
        }
    }
    block_bounds_reduce(mn, mx, bounds);
}

__global__ __launch_bounds__(kBlock) void deinterleave_kernel(const uint8_t* __restrict__ src, uint32_t count,
                                                              uint32_t stride, uint32_t off, float* __restrict__ x,
                                                              float* __restrict__ y, float* __restrict__ z) {
    uint32_t v = blockIdx.x * kBlock + threadIdx.x;
    if (v >= count) return;
    const float* p = (const float*)(src + (size_t)v * stride + off);
    x[v] = p[0];
    y[v] = p[1];
    z[v] = p[2];
}

// ---- centroid bounds: first loop of lbvh_make, src/lbvh/lbvh.c:95-105 (release build) ----
__global__ void bounds_init_kernel(uint32_t* bounds) {  // 8 threads per slot
    const int slot = threadIdx.x / 8, i = threadIdx.x % 8;
    if (i < 3) bounds[slot * kBoundStride + i] = f2ord(3.402823466e+38f);
    else if (i < 6) bounds[slot * kBoundStride + i] = f2ord(-3.402823466e+38f);
}

__global__ __launch_bounds__(kBlock) void bounds_kernel(const float* __restrict__ aabb6, uint64_t n,
                                                        uint32_t* __restrict__ bounds, uint32_t box_bounds) {
    float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
    float mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
        const float* b = aabb6 + 6 * i;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float c = (b[k] + b[k + 3]) * 0.5f;  // mul_3f32(add_3f32(min, max), 0.5f), lbvh.c:98
            mn[k] = ref_min(mn[k], box_bounds ? b[k] : c);      // a BUILD_DEBUG build bounds the boxes, lbvh.c:101-104
            mx[k] = ref_max(mx[k], box_bounds ? b[k + 3] : c);
        }
    }
    block_bounds_reduce(mn, mx, bounds);
}

// ---- K2: lbvh_c_to_morton_code, lbvh.c:74-90 ----------------------------------------
__device__ __forceinline__ uint64_t spread21(uint32_t v) {  // bit i of v -> bit 3i
    uint64_t x = v & 0x1fffffu;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
// norm is binary32 (subtraction then IEEE division); the reference widens it to double
// and scales by 2^21, which is exact, so the same integer comes out of a binary32
// scale.  0/0 on a zero-extent axis is NaN, which the reference's x86-64 cast turns
// into 0.  Only bits 0..20 are used, so norm == 1.0 contributes 0 (no clamp).
__device__ __forceinline__ uint32_t morton_axis(float c, float mn, float ext) {
    float norm = (c - mn) / ext;
    float scaled = norm * 2097152.0f;
    if (!(scaled == scaled)) return 0u;
    if (!(scaled < 4294967296.0f) || scaled < 0.f) return 0u;
    return (uint32_t)scaled;
}

// the code of box i: centroid (lbvh.c:114) normalised by the centroid bounds, 21 bits per axis, x lowest
__device__ __forceinline__ uint64_t morton_of_box(const float* __restrict__ aabb6, uint64_t i, const float* mn, const float* ext) {
    const float* b = aabb6 + 6 * i;
    uint64_t code = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float c = (b[k] + b[k + 3]) * 0.5f;  // lbvh.c:114
        code |= spread21(morton_axis(c, mn[k], ext[k])) << k;
    }
    return code;
}

__global__ __launch_bounds__(kBlock) void morton_kernel(const float* __restrict__ aabb6, uint64_t n,
                                                        const uint32_t* __restrict__ bounds,
                                                        uint64_t* __restrict__ keys, uint32_t* __restrict__ vals,
                                                        uint32_t* __restrict__ zero, uint32_t* __restrict__ zero_word = nullptr) {
    __shared__ uint32_t sb[6];
    if (zero_word && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0u;  // the sort's flag, cleared on the way as well
    bounds_fold(bounds, sb);  // once per workgroup of a bounded grid
    float mn[3], ext[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        mn[k] = ord2f(sb[k]);
        ext[k] = ord2f(sb[k + 3]) - mn[k];  // extents = max - min, lbvh.c:106
    }
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
        if (zero) zero[i] = 0u;  // the climb's arrival counters, cleared on the way (one launch fewer)
        keys[i] = morton_of_box(aabb6, i, mn, ext);
        if (vals) vals[i] = (uint32_t)i;  // the sort's first pass takes positions as values by itself
    }
}

// ---- K3: LSD radix sort, 8 bits per pass, stable ---------------------------------------
// Replaces radsort(...) at lbvh.c:120.  Equal codes keep ascending primitive index.

// MORTON (a build's first pass): the keys are made here, as morton_kernel would have made them -- the Morton codes of
// the boxes go to `keys`, the climb's counters and the sort's flag are cleared -- and counted on the way.
template <int BLOCK, bool MORTON = false>  // threads per workgroup; a tile is BLOCK * kSortItems keys
__global__ __launch_bounds__(BLOCK) void radix_hist_kernel(uint64_t* __restrict__ keys, uint64_t n, int shift,
                                                           uint32_t* __restrict__ block_hist, uint32_t tiles, MortonSource src = MortonSource{}) {
    __shared__ uint32_t hist[256];
    float mn[3] = {0.f, 0.f, 0.f}, ext[3] = {0.f, 0.f, 0.f};
    if (threadIdx.x < 256) hist[threadIdx.x] = 0;
    if constexpr (MORTON) {
        __shared__ uint32_t sb[6];
        bounds_fold(src.bounds, sb);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            mn[k] = ord2f(sb[k]);
            ext[k] = ord2f(sb[k + 3]) - mn[k];  // extents = max - min, lbvh.c:106
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            if (src.zero_word) *src.zero_word = 0u;
            if (src.zero_word2) *src.zero_word2 = 0u;
        }
    }
    __syncthreads();
    uint64_t base = (uint64_t)blockIdx.x * (BLOCK * kSortItems);
#pragma unroll
    for (int j = 0; j < kSortItems; j++) {
        uint64_t i = base + (uint64_t)j * BLOCK + threadIdx.x;
        if (i < n) {
            uint64_t key;
            if constexpr (MORTON) {
                key = morton_of_box(src.aabb6, i, mn, ext);
                keys[i] = key;
                if (src.zero_all) src.zero_all[i] = 0u;
            } else {
                key = keys[i];
            }
            atomicAdd(&hist[(uint32_t)(key >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < 256) block_hist[(uint64_t)threadIdx.x * tiles + blockIdx.x] = hist[threadIdx.x];
}

// Workgroup d turns row d of the digit-major tile histogram into exclusive prefix sums over the tiles
// and leaves the row's total behind the table; the scatter kernel adds the digit bases itself.
__global__ __launch_bounds__(kBlock) void radix_offsets_kernel(uint32_t* __restrict__ block_hist, uint32_t tiles) {
    __shared__ uint32_t wave_tot[kBlock / 64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    uint32_t* row = block_hist + (uint64_t)blockIdx.x * tiles;
    uint32_t carry = 0;
    for (uint32_t base = 0; base < tiles; base += kScanTile) {
        const uint32_t first = base + (uint32_t)tid * kScanItems;
        uint32_t v[kScanItems];
        uint32_t run = 0;
#pragma unroll
        for (int j = 0; j < kScanItems; j++) {
            v[j] = (first + j < tiles) ? row[first + j] : 0u;
            run += v[j];
        }
        uint32_t incl = run;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t up = __shfl_up(incl, o);
            if (lane >= o) incl += up;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        uint32_t wsum = 0, total = 0;
        for (int w = 0; w < kBlock / 64; w++) {
            if (w < wave) wsum += wave_tot[w];
            total += wave_tot[w];
        }
        uint32_t acc = carry + wsum + incl - run;
#pragma unroll
        for (int j = 0; j < kScanItems; j++) {
            if (first + j < tiles) row[first + j] = acc;
            acc += v[j];
        }
        carry += total;
        __syncthreads();  // wave_tot is reused by the next round
    }
    if (tid == 0) block_hist[256ull * tiles + blockIdx.x] = carry;
}

// Each workgroup ranks the keys of its tile by the current digit (wave-level match-any on
// ballots, wave-private LDS counters), sorts the tile in LDS and writes every digit
// bucket to its place in the output with consecutive lanes on consecutive addresses.
// Tiles of 4096 keys (1024 threads); of 1024 keys (256 threads) for inputs so small that the large tiles would leave
// most CUs idle (see radix_sort_workspace_device).
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void radix_scatter_kernel(const uint64_t* __restrict__ keys_in,
                                                              const uint32_t* __restrict__ vals_in,
                                                              uint64_t* __restrict__ keys_out,
                                                              uint32_t* __restrict__ vals_out, uint64_t n, int shift,
                                                              const uint32_t* __restrict__ gbase, uint32_t tiles) {
    // gbase[d * tiles + t]: keys with digit d in the tiles before t; gbase[256 * tiles + d]: keys with digit d
    constexpr int kSortBlock = BLOCK, kSortTile = BLOCK * kSortItems;  // shadow the file-scope values of the large tile
    constexpr int kWaves = kSortBlock / 64;
    __shared__ uint32_t wave_cnt[kWaves][256];
    __shared__ uint32_t bin_start[256];
    __shared__ uint32_t out_base[256];
    __shared__ uint32_t wave_tot[4], dig_tot[4];  // the 256 digits live on the first four waves
    __shared__ uint64_t skeys[kSortTile];
    __shared__ uint32_t svals[kSortTile];

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint64_t base = (uint64_t)blockIdx.x * kSortTile;
    const uint64_t wbase = base + (uint64_t)wave * (64 * kSortItems);

    for (int k = tid; k < kWaves * 256; k += kSortBlock) (&wave_cnt[0][0])[k] = 0;

    // everything the tile needs from global memory is asked for here, in one round trip: keys, values, and the two
    // offsets of "its" digit that each of the first 256 threads will use after the ranking
    uint64_t key[kSortItems];
    uint32_t val[kSortItems];
#pragma unroll
    for (int j = 0; j < kSortItems; j++) {
        uint64_t i = wbase + (uint64_t)j * 64 + lane;
        key[j] = (i < n) ? keys_in[i] : ~0ull;
        val[j] = (i < n) ? (vals_in ? vals_in[i] : (uint32_t)i) : 0u;
    }
    uint32_t total = 0, before_tile = 0;
    if (tid < 256) {
        total = gbase[256ull * tiles + tid];
        before_tile = gbase[(uint64_t)tid * tiles + blockIdx.x];
    }
    __syncthreads();

    uint32_t rank[kSortItems];
    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int j = 0; j < kSortItems; j++) {
        uint32_t d = (uint32_t)(key[j] >> shift) & 255u;
        uint64_t same = ~0ull;
#pragma unroll
        for (int b = 0; b < 8; b++) {
            bool bit = (d >> b) & 1u;
            uint64_t bal = __ballot(bit);
            same &= bit ? bal : ~bal;
        }
        uint32_t below = __popcll(same & lt_mask);
        uint32_t old = 0;
        if (below == 0) {  // lowest lane of each digit group updates the wave's counter
            old = wave_cnt[wave][d];
            wave_cnt[wave][d] = old + __popcll(same);
        }
        old = __shfl(old, __ffsll((long long)same) - 1);
        rank[j] = old + below;
    }
    __syncthreads();

    {   // thread d < 256: offsets of each wave inside digit d, digit totals, exclusive scans over the digits
        const bool owner = tid < 256;
        uint32_t run = 0;
        if (owner) {
#pragma unroll
            for (int w = 0; w < kWaves; w++) {
                uint32_t c = wave_cnt[w][tid];
                wave_cnt[w][tid] = run;
                run += c;
            }
        }
        uint32_t incl = run, dincl = total;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t up = __shfl_up(incl, o), dup = __shfl_up(dincl, o);
            if (lane >= o) {
                incl += up;
                dincl += dup;
            }
        }
        if (owner && lane == 63) {
            wave_tot[wave] = incl;
            dig_tot[wave] = dincl;
        }
        __syncthreads();
        if (owner) {
            uint32_t wsum = 0, dsum = 0;
            for (int w = 0; w < wave; w++) {
                wsum += wave_tot[w];
                dsum += dig_tot[w];
            }
            const uint32_t start = wsum + incl - run;
            bin_start[tid] = start;
            // first output position of digit d = keys with a smaller digit + keys with digit d in earlier tiles
            out_base[tid] = (dsum + dincl - total) + before_tile - start;
        }
    }
    __syncthreads();

#pragma unroll
    for (int j = 0; j < kSortItems; j++) {
        uint32_t d = (uint32_t)(key[j] >> shift) & 255u;
        uint32_t pos = bin_start[d] + wave_cnt[wave][d] + rank[j];
        skeys[pos] = key[j];
        svals[pos] = val[j];
    }
    __syncthreads();

    uint64_t left = n - base;
    uint32_t valid = left < (uint64_t)kSortTile ? (uint32_t)left : (uint32_t)kSortTile;
#pragma unroll
    for (int j = 0; j < kSortItems; j++) {
        uint32_t p = (uint32_t)j * kSortBlock + tid;
        if (p < valid) {
            uint64_t k = skeys[p];
            uint32_t d = (uint32_t)(k >> shift) & 255u;
            uint32_t dst = out_base[d] + p;
            keys_out[dst] = k;
            vals_out[dst] = svals[p];
        }
    }
}

// ---- sorts of at most 128 tiles: every tile sorted completely in LDS, then merged by rank ----------------
// Up to 2^17 keys the pass-by-pass sort is a chain of 24 small launches (its keys fit the L2 many times over), so
// these sizes take two.  tile_sort_kernel sorts each tile of 1024 or 2048 keys (merge_tile_keys) without leaving LDS
// (the same wave-ballot ranking as the scatter kernel, the tile ping-ponging between registers and LDS; tiles this
// small, so that a sort of 80 000 keys is spread over 40 - 80 CUs).  tile_merge_kernel gives every
// key its final position: its position in its own tile plus, for every other tile, the number of keys there that must
// come before it (keys <= in earlier tiles, < in later ones, which keeps equal keys in their original order).
constexpr int kSmallSortBlock = 256;
constexpr int kSmallTile = kSmallSortBlock * kSortItems;  // 1024 keys
constexpr int kMergeBatch = 4;   // tiles a wave ranks against side by side
constexpr int kMergeSeg = 128;   // keys of a tile's matching stretch held in LDS
constexpr uint32_t kTileFixLong = 32;  // tile_sort_kernel finishes stretches of equal upper bytes up to this long by ranking

// A tile of BLOCK * ITEMS keys, sorted stably through the digits top_first .. n_bytes - 1 in LDS.  With top_first > 0
// (Morton codes: the upper four bytes decide the order of all but a few neighbours) the keys that share those digits
// then lie in short stretches in their original order, and each is finished by ranking inside its stretch (smaller
// lower digits first, equal ones by position); a stretch longer than kTileFixLong makes the workgroup start again
// with all n_bytes digits.  MORTON: the keys are the Morton codes of the boxes aabb6 (keys_in is not read: what
// morton_kernel would have written there is made here, one launch fewer; the values are the positions), and *zero_word
// is cleared for the kernels that follow.
template <int BLOCK, int ITEMS, bool MORTON = false>
__global__ __launch_bounds__(BLOCK) void tile_sort_kernel(const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                          uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                          uint64_t n, int n_bytes, int top_first, const float* __restrict__ aabb6 = nullptr,
                                                          const uint32_t* __restrict__ bounds = nullptr, uint32_t* __restrict__ zero_word = nullptr) {
    constexpr int kWaves = BLOCK / 64, kTile = BLOCK * ITEMS;
    static_assert(BLOCK >= 256, "one thread per digit");
    __shared__ uint32_t wave_cnt[kWaves][256];
    __shared__ uint32_t bin_start[256];
    __shared__ uint32_t wave_tot[4];
    __shared__ uint64_t skeys[kTile];
    __shared__ uint32_t svals[kTile];
    __shared__ uint32_t long_stretch;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const uint64_t base = (uint64_t)blockIdx.x * kTile;
    const uint32_t wpos = (uint32_t)wave * (64 * ITEMS);  // this wave's first position in the tile
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const uint32_t valid = n - base < (uint64_t)kTile ? (uint32_t)(n - base) : (uint32_t)kTile;
    const bool owner = tid < 256;  // thread d < 256 scans digit d

    if (tid == 0) long_stretch = 0;  // set only behind the barriers of the passes, read behind one more
    uint64_t code[MORTON ? ITEMS : 1];
    if constexpr (MORTON) {
        __shared__ uint32_t sb[6];
        float cen[ITEMS][3];  // asked for before the bounds: the two round trips to memory overlap
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
            const uint64_t i = base + wpos + (uint64_t)j * 64 + lane;
            const float* b = aabb6 + 6 * (i < n ? i : 0);
#pragma unroll
            for (int k = 0; k < 3; k++) cen[j][k] = (b[k] + b[k + 3]) * 0.5f;  // lbvh.c:114
        }
        bounds_fold(bounds, sb);
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
            code[j] = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float mn = ord2f(sb[k]), ext = ord2f(sb[k + 3]) - mn;  // extents = max - min, lbvh.c:106
                code[j] |= spread21(morton_axis(cen[j][k], mn, ext)) << k;
            }
        }
        if (blockIdx.x == 0 && tid == 0) *zero_word = 0u;
    }
    uint64_t key[ITEMS];
    uint32_t val[ITEMS];
    for (int first = top_first;; first = 0) {
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
            const uint64_t i = base + wpos + (uint64_t)j * 64 + lane;
            if constexpr (MORTON) {
                key[j] = (i < n) ? code[j] : ~0ull;
                val[j] = (i < n) ? (uint32_t)i : 0u;
            } else {
                key[j] = (i < n) ? keys_in[i] : ~0ull;  // padding sorts behind every real key and keeps its place there
                val[j] = (i < n) ? (vals_in ? vals_in[i] : (uint32_t)i) : 0u;
            }
        }
        for (int pass = first; pass < n_bytes; pass++) {
            const int shift = pass * 8;
            // a wave's counters are its own until the barrier below (the previous pass's readers of them are behind that
            // pass's later barriers), so clearing them needs no barrier
#pragma unroll
            for (int q = 0; q < 4; q++) wave_cnt[wave][q * 64 + lane] = 0;
            // the first lane of every group of equal digits adds the group's size to the wave's counter; the items' adds
            // are issued one behind the other (LDS keeps a wave's order, so item j counts before item j + 1) and waited
            // for together
            uint32_t rank[ITEMS], old[ITEMS];
            int leader[ITEMS];
#pragma unroll
            for (int j = 0; j < ITEMS; j++) {
                const uint32_t d = (uint32_t)(key[j] >> shift) & 255u;
                uint64_t same = ~0ull;
#pragma unroll
                for (int b = 0; b < 8; b++) {
                    const bool bit = (d >> b) & 1u;
                    const uint64_t bal = __ballot(bit);
                    same &= bit ? bal : ~bal;
                }
                rank[j] = __popcll(same & lt_mask);
                leader[j] = __ffsll((long long)same) - 1;
                old[j] = 0;
                if (rank[j] == 0) old[j] = atomicAdd(&wave_cnt[wave][d], (uint32_t)__popcll(same));
            }
#pragma unroll
            for (int j = 0; j < ITEMS; j++) rank[j] += __shfl(old[j], leader[j]);
            __syncthreads();
            {   // thread d: offsets of each wave inside digit d, then the exclusive scan over the digits
                uint32_t run = 0;
                if (owner) {
#pragma unroll
                    for (int w = 0; w < kWaves; w++) {
                        const uint32_t c = wave_cnt[w][tid];
                        wave_cnt[w][tid] = run;
                        run += c;
                    }
                }
                uint32_t incl = run;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const uint32_t up = __shfl_up(incl, o);
                    if (lane >= o) incl += up;
                }
                if (owner) {
                    if (lane == 63) wave_tot[wave] = incl;
                    bin_start[tid] = incl - run;  // among the 64 digits of this thread's wave; the waves before it: below
                }
            }
            __syncthreads();
            const uint32_t g1 = wave_tot[0], g2 = g1 + wave_tot[1], g3 = g2 + wave_tot[2];
#pragma unroll
            for (int j = 0; j < ITEMS; j++) {
                const uint32_t d = (uint32_t)(key[j] >> shift) & 255u;
                const uint32_t group = d < 64u ? 0u : (d < 128u ? g1 : (d < 192u ? g2 : g3));  // keys with digits of earlier groups of 64
                const uint32_t pos = group + bin_start[d] + wave_cnt[wave][d] + rank[j];
                skeys[pos] = key[j];
                svals[pos] = val[j];
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < ITEMS; j++) {
                key[j] = skeys[wpos + j * 64 + lane];
                val[j] = svals[wpos + j * 64 + lane];
            }
            // no barrier: the next pass writes the tile only after its own three barriers
        }
        if (first == 0) break;

        // the tile is in order by its upper digits and nobody writes it before the barrier below
        const uint32_t* words = (const uint32_t*)skeys;  // key q: low word 2q, high word 2q + 1
        uint32_t dst[ITEMS];
        bool gave_up = false;
#pragma unroll
        for (int j = 0; j < ITEMS; j++) {
            const uint32_t p = wpos + (uint32_t)j * 64 + (uint32_t)lane;
            dst[j] = p;
            if (p < valid) {
                const uint32_t top = (uint32_t)(key[j] >> 32), low = (uint32_t)key[j];
                uint32_t a = p, b = p + 1;
                while (a > 0 && p - a < kTileFixLong && words[2 * (a - 1) + 1] == top) a--;
                while (b < valid && b - p < kTileFixLong && words[2 * b + 1] == top) b++;
                if (p - a >= kTileFixLong || b - p >= kTileFixLong) {
                    gave_up = true;
                } else if (b - a > 1) {
                    uint32_t r = a;
                    for (uint32_t q = a; q < b; q++) {
                        const uint32_t lq = words[2 * q];
                        r += (lq < low || (lq == low && q < p)) ? 1u : 0u;
                    }
                    dst[j] = r;
                }
            }
        }
        if (gave_up) long_stretch = 1;
        __syncthreads();
        if (!long_stretch) {
#pragma unroll
            for (int j = 0; j < ITEMS; j++) {
                const uint32_t p = wpos + (uint32_t)j * 64 + (uint32_t)lane;
                if (p < valid) {
                    keys_out[base + dst[j]] = key[j];
                    vals_out[base + dst[j]] = val[j];
                }
            }
            return;
        }
    }
#pragma unroll
    for (int j = 0; j < ITEMS; j++) {
        const uint64_t i = base + wpos + (uint64_t)j * 64 + lane;
        if (i < n) {
            keys_out[i] = key[j];
            vals_out[i] = val[j];
        }
    }
}

// keys of a sorted stretch that come before k: `<=` when the stretch's tile lies before k's tile, `<` when behind
__device__ __forceinline__ uint32_t rank_in_global(const uint64_t* __restrict__ keys, uint32_t count, uint64_t k, bool or_equal, uint64_t mask) {
    uint32_t lo = 0, hi = count;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        const uint64_t m = keys[mid] & mask;
        if (or_equal ? m <= k : m < k) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

// A workgroup ranks 64 neighbouring keys of one sorted tile against all the other tiles, a quarter of the tiles per
// wave.  Neighbouring keys have neighbouring ranks: a wave first finds where its first and its last key fall in each of
// its tiles (one lane per tile and end, binary searches in global memory), which bounds the stretch of the tile that
// can lie between its keys -- as long as the 64 keys themselves on average.  Then, kMergeBatch tiles at a time, it
// copies those stretches to LDS with coalesced loads and every lane ranks its key there.  Against one binary search of
// the whole tile per key and tile (12 dependent, fully divergent loads each; 41 us for 81 920 keys) this leaves two
// searches per 64 keys and tile.  A stretch of kMergeSeg keys or more (tiles with very different key ranges) is
// searched in global memory.
__global__ __launch_bounds__(256) void tile_merge_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                         uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out,
                                                         uint32_t n, uint32_t tiles, uint32_t tile_keys, uint64_t mask) {
    __shared__ uint32_t ends[4][2][32];
    __shared__ uint64_t seg[4][kMergeBatch][kMergeSeg];
    __shared__ uint32_t partial[4][64];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t i0 = blockIdx.x * 64u, i = i0 + lane;
    const bool live = i < n;
    const uint32_t last_lane = (n - i0 < 64u ? n - i0 : 64u) - 1u;
    const uint64_t full = keys[live ? i : i0], k = full & mask;
    const uint64_t k_first = __shfl(k, 0), k_last = __shfl(k, (int)last_lane);
    const uint32_t mine = i0 / tile_keys;
    const uint32_t per_wave = (tiles + 3u) / 4u, t0 = wave * per_wave;  // this wave's tiles: t0 .. t0 + per_wave - 1

    {   // where the first (lanes 0-31) and the last key (lanes 32-63) fall in tile t0 + (lane & 31)
        const uint32_t tt = lane & 31u, t = t0 + tt, e = lane >> 5;
        uint32_t r = 0;
        if (tt < per_wave && t < tiles && t != mine) {
            const uint32_t first = t * tile_keys;
            const uint32_t count = n - first < tile_keys ? n - first : tile_keys;
            r = rank_in_global(keys + first, count, e ? k_last : k_first, t < mine, mask);
        }
        ends[wave][e][tt] = r;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");

    uint32_t before = 0;
    for (uint32_t tb = 0; tb < per_wave; tb += kMergeBatch) {
        // A stretch of up to kMergeSeg - 1 keys goes to LDS, padded with all-ones words; a key k is ranked by the
        // longest prefix of words below kk = k (+ 1 for tiles before this one, where equal keys come first too): the
        // padding is below no kk, so the search needs no length.
        uint32_t lo[kMergeBatch], len[kMergeBatch];
        uint64_t a[kMergeBatch], b[kMergeBatch], kk[kMergeBatch];
        bool every[kMergeBatch];
#pragma unroll
        for (int u = 0; u < kMergeBatch; u++) {
            const uint32_t tt = tb + u, t = t0 + tt;
            const bool valid = tt < per_wave && t < tiles && t != mine;
            lo[u] = valid ? ends[wave][0][tt & 31u] : 0u;
            len[u] = valid ? ends[wave][1][tt & 31u] - lo[u] : 0u;
            const bool or_equal = t < mine;
            kk[u] = k + (or_equal ? 1ull : 0ull);
            every[u] = or_equal && k == ~0ull;  // k + 1 wrapped: every key of an earlier tile comes first
            const uint64_t* src = keys + (uint64_t)t * tile_keys + lo[u];
            a[u] = b[u] = ~0ull;
            if (len[u] < (uint32_t)kMergeSeg) {
                if (lane < len[u]) a[u] = src[lane] & mask;
                if (lane + 64u < len[u]) b[u] = src[lane + 64u] & mask;
            }
        }
#pragma unroll
        for (int u = 0; u < kMergeBatch; u++) {
            seg[wave][u][lane] = a[u];
            seg[wave][u][lane + 64u] = b[u];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        uint32_t pos[kMergeBatch];
#pragma unroll
        for (int u = 0; u < kMergeBatch; u++) pos[u] = 0;
#pragma unroll
        for (uint32_t step = kMergeSeg / 2; step >= 1u; step >>= 1) {
#pragma unroll
            for (int u = 0; u < kMergeBatch; u++) {
                const uint32_t p = pos[u] + step;  // <= kMergeSeg - 1
                if (seg[wave][u][p - 1u] < kk[u]) pos[u] = p;
            }
        }
#pragma unroll
        for (int u = 0; u < kMergeBatch; u++) {
            if (len[u] >= (uint32_t)kMergeSeg) {  // uniform over the wave
                const uint32_t t = t0 + tb + u;
                pos[u] = rank_in_global(keys + (uint64_t)t * tile_keys + lo[u], len[u], k, t < mine, mask);
            } else if (every[u]) {
                pos[u] = len[u];
            }
            before += lo[u] + pos[u];
        }
        __builtin_amdgcn_wave_barrier();  // the next batch overwrites the stretches
    }
    partial[wave][lane] = before;
    __syncthreads();
    if (wave == 0 && live) {
        const uint32_t rank = (i - mine * tile_keys) + partial[0][lane] + partial[1][lane] + partial[2][lane] + partial[3][lane];
        keys_out[rank] = full;
        vals_out[rank] = vals[i];
    }
}

// ---- a sort by the upper digits only, finished locally --------------------------------------------------------
// Morton codes of a real scene differ in their upper bits: after stable passes over the upper kTopDigits bytes only,
// the keys that share all of those bits form short stretches (random points: 23 000 pairs in 10^7 keys; a surface: half
// a dozen keys), already in their original order.  sort_fix_kernel finishes the sort inside every stretch of up to
// kFixLong keys by ranking (a key's place is the stretch's first place plus the number of keys of the stretch that must
// come before it: smaller low bits, or equal ones and an earlier position), in place: a stretch belongs to the
// workgroup whose tile holds its first key, which reads all of it before anything is written; other workgroups look
// only at its upper bits, which no write changes.  A longer stretch that is out of order (a dense object in a scene
// thousands of times its size) is left alone and reported in *gave_up: sort_long_kernel, launched next, finishes those.
constexpr int kFixTile = 2048;  // positions per workgroup
constexpr int kFixLong = 128;   // stretches up to this long are finished here; also the halo read on either side of the tile
constexpr int kFixWin = kFixTile + 2 * kFixLong;
constexpr int kFixPer = (kFixTile + kFixLong + 255) / 256;  // positions of tile + right halo per thread

__global__ __launch_bounds__(256) void sort_fix_kernel(uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, uint32_t n,
                                                       uint32_t* __restrict__ gave_up) {
    __shared__ uint32_t top[kFixWin], low[kFixWin];
    __shared__ uint32_t any_out_of_order;
    const int tid = threadIdx.x;
    const int64_t w0 = (int64_t)blockIdx.x * kFixTile - kFixLong;  // window entry k is position w0 + k
    if (tid == 0) any_out_of_order = 0;
    for (int k = tid; k < kFixWin; k += 256) {
        const int64_t p = w0 + k;
        const bool inside = p >= 0 && p < (int64_t)n;
        const uint64_t key = inside ? keys[p] : 0ull;
        top[k] = inside ? (uint32_t)(key >> 32) : 0xffffffffu;  // no key has bit 63: equal to no real neighbour
        low[k] = (uint32_t)key;
    }
    __syncthreads();
    {
        bool found = false;
        for (int k = kFixLong + tid; k < kFixWin; k += 256) found |= top[k] == top[k - 1] && low[k - 1] > low[k];
        if (found) any_out_of_order = 1;
    }
    __syncthreads();
    if (!any_out_of_order) return;

    // (Building the tile's new arrangement in LDS and writing it back with consecutive lanes was slower than these
    // scattered stores: 52 against 33 us at 10^7 random keys, 126 against 98 us on a surface, for the LDS it takes.)
    uint32_t to[kFixPer], val[kFixPer];
#pragma unroll
    for (int j = 0; j < kFixPer; j++) {
        to[j] = 0xffffffffu;
        const int k = kFixLong + tid + 256 * j;
        const int64_t p = w0 + k;
        if (k >= kFixWin || p >= (int64_t)n) continue;
        const uint32_t t = top[k], mine_low = low[k];
        if (top[k - 1] != t && (k + 1 >= kFixWin || top[k + 1] != t)) continue;  // a stretch of one key (most are)
        // keys of the same stretch before and behind this one, counted up to the length that makes the stretch long
        int before = 0, behind = 0;
        while (before <= kFixLong && k - before - 1 >= 0 && top[k - before - 1] == t) before++;
        while (before + behind <= kFixLong && k + behind + 1 < kFixWin && top[k + behind + 1] == t) behind++;
        // the window's edges are at least kFixLong away from every key that a stretch of this tile can hold, so a scan
        // that ends at an edge has seen a long stretch as well
        const bool is_long = before + behind + 1 > kFixLong || k - before - 1 < 0 || k + behind + 1 >= kFixWin;
        const bool in_tile = k < kFixLong + kFixTile;
        if (is_long) {
            if (in_tile && before > 0 && low[k - 1] > mine_low) atomicOr(gave_up, 1u);
            continue;
        }
        const int first = k - before;
        if (first < kFixLong || first >= kFixLong + kFixTile) continue;  // another workgroup's stretch
        int rank = 0;
        for (int q = first; q <= k + behind; q++) rank += (low[q] < mine_low || (low[q] == mine_low && q < k)) ? 1 : 0;
        if (rank != before) {
            to[j] = (uint32_t)(w0 + first + rank);
            val[j] = vals[p];
        }
    }
    __syncthreads();  // every value that moves has been read
#pragma unroll
    for (int j = 0; j < kFixPer; j++) {
        if (to[j] != 0xffffffffu) {
            const int k = kFixLong + tid + 256 * j;
            keys[to[j]] = ((uint64_t)top[k] << 32) | low[k];
            vals[to[j]] = val[j];
        }
    }
}

// The long stretches sort_fix_kernel left alone, finished on the device as well, so that no build has to tell the host
// how its sort went and wait for the answer.  Launched behind sort_fix_kernel in every build; when that kernel reported
// nothing (nearly always) every workgroup leaves at its first instruction.  Otherwise each workgroup looks for the heads
// of stretches in its share of the positions and sorts the long ones that are out of order by their lower four bytes,
// stably: up to kLongLds keys by ranking in LDS, longer ones by four radix passes of its own through the sort's other
// buffer (one workgroup per stretch: bounded, not fast -- a build of that many boxes goes by all eight passes from the
// next time on, see lbvh_build_device).
constexpr int kLongLds = 2048;

__global__ __launch_bounds__(256) void sort_long_kernel(uint64_t* __restrict__ keys, uint32_t* __restrict__ vals, uint64_t* __restrict__ tmp_keys,
                                                        uint32_t* __restrict__ tmp_vals, uint32_t n, const uint32_t* __restrict__ gave_up) {
    if (*gave_up == 0u) return;
    __shared__ uint32_t s_low[kLongLds], s_val[kLongLds];
    __shared__ uint32_t hist[256], cnt[4][256];
    __shared__ uint32_t heads[256];
    __shared__ uint32_t n_heads, s_end, s_unordered;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const uint32_t share = (n + gridDim.x - 1u) / gridDim.x;
    const uint32_t first = blockIdx.x * share, last = first + share < n ? first + share : n;  // heads in [first, last) are this workgroup's
    for (uint32_t p0 = first; p0 < last; p0 += 256u) {
        if (tid == 0) n_heads = 0;
        __syncthreads();
        {
            const uint32_t p = p0 + tid;
            if (p < last) {
                const uint32_t t = (uint32_t)(keys[p] >> 32);
                const bool head = p == 0u || (uint32_t)(keys[p - 1u] >> 32) != t;
                // sorted by the upper bytes: the stretch is longer than kFixLong exactly if the key kFixLong further on shares them
                if (head && p + (uint32_t)kFixLong < n && (uint32_t)(keys[p + (uint32_t)kFixLong] >> 32) == t) heads[atomicAdd(&n_heads, 1u)] = p;
            }
        }
        __syncthreads();
        const uint32_t count = n_heads;
        for (uint32_t e = 0; e < count; e++) {
            const uint32_t h = heads[e];
            const uint32_t t = (uint32_t)(keys[h] >> 32);
            if (tid == 0) {  // the stretch's end: gallop, then bisect (the upper bytes are in order)
                uint32_t lo = h + (uint32_t)kFixLong, step = (uint32_t)kFixLong;  // lo is inside the stretch
                while (lo + step < n && (uint32_t)(keys[lo + step] >> 32) == t) {
                    lo += step;
                    if (step < (1u << 30)) step <<= 1;
                }
                uint32_t hi = lo + step < n ? lo + step : n;  // hi is outside (or n)
                while (hi - lo > 1u) {
                    const uint32_t mid = lo + (hi - lo) / 2u;
                    if ((uint32_t)(keys[mid] >> 32) == t) lo = mid;
                    else hi = mid;
                }
                s_end = hi;
                s_unordered = 0;
            }
            __syncthreads();
            const uint32_t len = s_end - h;
            {
                bool found = false;
                for (uint32_t q = 1u + tid; q < len; q += 256u) found |= (uint32_t)keys[h + q - 1u] > (uint32_t)keys[h + q];
                if (found) s_unordered = 1;
            }
            __syncthreads();
            if (!s_unordered) continue;  // (uniform: read after the barrier, next written after the next one)
            if (len <= (uint32_t)kLongLds) {
                for (uint32_t q = tid; q < len; q += 256u) {
                    s_low[q] = (uint32_t)keys[h + q];
                    s_val[q] = vals[h + q];
                }
                __syncthreads();
                for (uint32_t q = tid; q < len; q += 256u) {
                    const uint32_t mine = s_low[q];
                    uint32_t rank = 0;
                    for (uint32_t r = 0; r < len; r++) rank += (s_low[r] < mine || (s_low[r] == mine && r < q)) ? 1u : 0u;
                    keys[h + rank] = ((uint64_t)t << 32) | mine;
                    vals[h + rank] = s_val[q];
                }
                __syncthreads();
                continue;
            }
            // four stable counting passes over the lower bytes, there and back between the two buffers
            for (int pass = 0; pass < 4; pass++) {
                const uint64_t* src_k = (pass & 1) ? tmp_keys + h : keys + h;
                const uint32_t* src_v = (pass & 1) ? tmp_vals + h : vals + h;
                uint64_t* dst_k = (pass & 1) ? keys + h : tmp_keys + h;
                uint32_t* dst_v = (pass & 1) ? vals + h : tmp_vals + h;
                const int shift = 8 * pass;
                hist[tid] = 0;
                cnt[0][tid] = cnt[1][tid] = cnt[2][tid] = cnt[3][tid] = 0;
                __syncthreads();
                for (uint32_t q = tid; q < len; q += 256u) atomicAdd(&hist[(uint32_t)(src_k[q] >> shift) & 255u], 1u);
                __syncthreads();
                if (tid == 0) {  // hist becomes the first place of every digit
                    uint32_t run = 0;
                    for (int d = 0; d < 256; d++) {
                        const uint32_t c = hist[d];
                        hist[d] = run;
                        run += c;
                    }
                }
                __syncthreads();
                for (uint32_t c0 = 0; c0 < len; c0 += 256u) {
                    const uint32_t q = c0 + tid;
                    const bool live = q < len;
                    const uint64_t key = live ? src_k[q] : 0ull;
                    const uint32_t val = live ? src_v[q] : 0u;
                    const uint32_t d = (uint32_t)(key >> shift) & 255u;
                    uint64_t same = __ballot(live);
#pragma unroll
                    for (int b = 0; b < 8; b++) {
                        const uint64_t ones = __ballot(live && ((d >> b) & 1u));
                        same &= ((d >> b) & 1u) ? ones : ~ones;
                    }
                    const uint32_t rank_in_wave = (uint32_t)__popcll(same & lt_mask);
                    if (live && rank_in_wave == 0u) cnt[wave][d] = (uint32_t)__popcll(same);
                    __syncthreads();
                    if (live) {
                        uint32_t to = hist[d] + rank_in_wave;
                        for (uint32_t w = 0; w < wave; w++) to += cnt[w][d];
                        dst_k[to] = key;
                        dst_v[to] = val;
                    }
                    __syncthreads();
                    hist[tid] += cnt[0][tid] + cnt[1][tid] + cnt[2][tid] + cnt[3][tid];
                    cnt[0][tid] = cnt[1][tid] = cnt[2][tid] = cnt[3][tid] = 0;
                    __syncthreads();
                }
            }
        }
        __syncthreads();  // heads[] is rewritten by the next round
    }
}

// ---- device-wide prefix sums of uint32 ------------------------------------------------
template <bool INCLUSIVE>
__global__ __launch_bounds__(kBlock) void scan_tile_kernel(uint32_t* __restrict__ data, uint64_t n,
                                                           uint32_t* __restrict__ sums) {
    __shared__ uint32_t wave_tot[kBlock / 64];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    static_assert(kScanItems == 8, "a thread's items are two 16-byte vectors");
    uint64_t base = (uint64_t)blockIdx.x * kScanTile + (uint64_t)tid * kScanItems;
    // a thread's eight values as two 16-byte accesses wherever the array allows it: with 4-byte accesses every
    // instruction of a wave touches all 16 of its cache lines (10^7 values: 43 us against 16)
    const bool vec = base + kScanItems <= n && ((uintptr_t)data & 15u) == 0;
    uint32_t v[kScanItems];
    if (vec) {
        const uint4 a = ((const uint4*)(data + base))[0], b = ((const uint4*)(data + base))[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
        v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int j = 0; j < kScanItems; j++) v[j] = (base + j < n) ? data[base + j] : 0u;
    }
    uint32_t run = 0;
#pragma unroll
    for (int j = 0; j < kScanItems; j++) run += v[j];
    uint32_t incl = run;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t up = __shfl_up(incl, o);
        if (lane >= o) incl += up;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t wsum = 0, total = 0;
    for (int w = 0; w < kBlock / 64; w++) {
        if (w < wave) wsum += wave_tot[w];
        total += wave_tot[w];
    }
    uint32_t acc = wsum + incl - run;  // exclusive prefix of this thread's first item
    uint32_t o[kScanItems];
#pragma unroll
    for (int j = 0; j < kScanItems; j++) {
        uint32_t x = v[j];
        if (INCLUSIVE) acc += x;
        o[j] = acc;
        if (!INCLUSIVE) acc += x;
    }
    if (vec) {
        ((uint4*)(data + base))[0] = make_uint4(o[0], o[1], o[2], o[3]);
        ((uint4*)(data + base))[1] = make_uint4(o[4], o[5], o[6], o[7]);
    } else {
#pragma unroll
        for (int j = 0; j < kScanItems; j++)
            if (base + j < n) data[base + j] = o[j];
    }
    if (sums && tid == 0) sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(kBlock) void scan_add_kernel(uint32_t* __restrict__ data, uint64_t n,
                                                          const uint32_t* __restrict__ sums) {
    uint32_t add = sums[blockIdx.x];
    uint64_t base = (uint64_t)blockIdx.x * kScanTile;
#pragma unroll
    for (int j = 0; j < kScanItems; j++) {
        uint64_t i = base + (uint64_t)j * kBlock + threadIdx.x;
        if (i < n) data[i] += add;
    }
}

// ---- K4: priority of every gap -----------------------------------------------------------
// lbvh_get_split_idx, lbvh.c:15-42, seen from the gap: differing codes -> shared leading
// bits; equal codes -> 64 + depth of the gap in the count/2 tree of its run.
__device__ __forceinline__ uint8_t gap_priority_of(const uint64_t* keys, uint64_t n, uint64_t g) {
    uint64_t k0 = keys[g], k1 = keys[g + 1];
    if (k0 != k1) return (uint8_t)__clzll((long long)(k0 ^ k1));
    // Run of equal codes [a, b): a = first index with key == k0, b = first index with key > k0.  Both ends are looked
    // for outwards from the gap in doubling steps, then by bisection inside the last step: the runs that occur in
    // practice are short (the two triangles of an axis-aligned quad share their box, hence their code), and a pair
    // costs two loads next to the gap instead of two bisections of the whole array.
    uint64_t in = g, step = 1;  // keys[in] == k0
    while (in >= step && keys[in - step] == k0) {
        in -= step;
        step <<= 1;
    }
    uint64_t lo = in >= step ? in - step + 1 : 0, hi = in;  // the first key of the run is in [lo, hi]
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if (keys[mid] < k0) lo = mid + 1;
        else hi = mid;
    }
    const uint64_t a = lo;
    in = g + 1;
    step = 1;
    while (in + step < n && keys[in + step] == k0) {
        in += step;
        step <<= 1;
    }
    lo = in + 1;  // the first key behind the run is in [lo, hi]
    hi = in + step < n ? in + step : n;
    while (lo < hi) {
        uint64_t mid = (lo + hi) >> 1;
        if (keys[mid] > k0) hi = mid;
        else lo = mid + 1;
    }
    const uint64_t b = lo;
    uint64_t s = a, c = b - a;
    uint32_t depth = 0;
    for (;;) {
        uint64_t m = c >> 1;          // count/2, lbvh.c:23
        uint64_t gap = s + m - 1;     // the gap this node splits at
        if (g == gap) break;
        if (g < gap) c = m;
        else {
            s += m;
            c -= m;
        }
        depth++;
    }
    return (uint8_t)(64u + depth);
}

// ---- K5 + K6: hierarchy emission with box refit ------------------------------------------
// lbvh_create_subtree, lbvh.c:44-72, without the recursion.
//
// Shape.  The node that splits at gap g (between sorted leaves g and g + 1) covers the leaves from behind the nearest
// gap of smaller priority on its left to the nearest gap of smaller priority on its right; its parent is the one of
// those two gaps with the larger priority (gap_priority_of: two gaps of equal priority always have one of smaller
// priority between them, so there are no ties to break).  In pre-order, a node with leaf range [l, r] sits at
//     2 l + (number of its ancestors that have it in their LEFT subtree):
// the l leaves before it, and of the l gaps before it those that are not its ancestors lie entirely to its left,
// while those that are count as ancestors either way.  For leaf i that number is the number of gaps g >= i whose node
// reaches down to leaf i, i.e. whose priority is smaller than that of every gap in [i, g): the records of a running
// minimum that starts at gap i.  Priorities are below 128 (63 bit levels, 64 + the depth in an equal-code run), so the
// set of records seen from gap i is a 128-bit mask M_i, and
//     M_i = (M_{i+1} & bits below p_i) | bit p_i                                         (VisF below)
// is a suffix scan with a small associative operator.  Three small kernels give every leaf popcount(M_i), one byte, its
// pre-order index being 2 i + popcount(M_i).  An inner node sits directly in front of its left child, so the box climb
// carries the count along (a parent's is its left child's minus one) and every node it finishes knows its place.  No
// climb, no hand-offs, no atomics for the shape.
//
// Boxes (merge_rng3_f32(left, right), lbvh.c:58) do need the children first: one bottom-up climb, one thread per sorted
// leaf.  A node is identified by its leaf range; a child leaves a record in its parent's slot and bumps the parent's
// arrival counter; the second child to arrive reads its sibling's record, becomes the parent, writes the parent's
// final 32-byte record straight to its pre-order place and walks on.  Nobody ever waits for anybody.
//
// The climb is two kernels.  climb_local: a workgroup owns kChunk consecutive sorted leaves.  A gap whose node lies
// entirely inside the chunk (it is bounded on both sides, inside the chunk, by gaps of smaller priority) is "local":
// both of its children are produced by this workgroup, so their hand-off goes through LDS.  A thread that reaches a
// parent that is not local leaves its record in HBM, counts itself there and is done; the second child to be counted
// puts the parent on a list.  climb_upper finishes the nodes that span chunks (a few per cent of all): one thread per
// node on the list, hand-offs through HBM.
//
// Hand-offs through HBM.  Every access to a record and to the counter is a device-scope atomic, so what the hand-off
// needs beyond per-location coherence is ORDER: record before count, count before the sibling's record.  MODE 2 (the
// default): the wave drains its record stores (s_waitcnt vmcnt(0): device-scope stores are acknowledged once they are
// visible device-wide on gfx950) before a relaxed counter add, and reads the sibling after the add has returned; the
// asm statement is a compiler barrier as well.  That is the hardware's ordering, not one the HIP memory model promises
// for relaxed atomics.  MODE 0 (RTHIP_REFIT_FENCE=0) is the model's own form: a device-scope release read-modify-write
// on the counter, and an acquire fence before the second child to arrive reads the sibling's record (the first reads
// nothing).  Same trees bit for bit (tests run all three); its release writes the L2 back and its acquire invalidates
// it at every level of a latency-bound chain (+8 % at 10^7 triangles).  MODE 1 (RTHIP_REFIT_FENCE=1, diagnostic):
// plain accesses bracketed by __threadfence().  Because MODE 2 leans on the chip and not on the model, a context takes
// it only after it has checked it on the device it runs on (handoff_self_check: one tree built both ways before the
// first build that needs the device-wide climb; any difference, or a build that cannot wait for the check, means MODE 0).
constexpr int kChunk = 512;  // sorted leaves per workgroup, one per thread
constexpr uint64_t kTopSortMin = 1ull << 17;  // builds of more boxes sort by the upper bytes first (lbvh_build_device)

// ---- the records of a running minimum, as a scan ------------------------------------------------------------------
// f(M) = (M & bits below t) | B: what a stretch of gaps does to the mask of the records seen from behind its right end.
// The mask is two 64-bit words: priorities below 64 (bit levels), and 64 and above (depths inside equal-code runs).
struct VisF {
    uint32_t t;     // 128: keeps everything (the empty stretch)
    uint64_t lo, hi;
};
__device__ __forceinline__ uint64_t below_lo(uint32_t t) { return t >= 64u ? ~0ull : ((1ull << t) - 1ull); }
__device__ __forceinline__ uint64_t below_hi(uint32_t t) { return t <= 64u ? 0ull : (t >= 128u ? ~0ull : ((1ull << (t - 64u)) - 1ull)); }
__device__ __forceinline__ VisF vis_identity() { return VisF{128u, 0ull, 0ull}; }
// the stretch `left` followed on its right by the stretch `right` (right acts on the mask first)
__device__ __forceinline__ VisF vis_join(const VisF& left, const VisF& right) {
    VisF r;
    r.t = left.t < right.t ? left.t : right.t;
    r.lo = (right.lo & below_lo(left.t)) | left.lo;
    r.hi = (right.hi & below_hi(left.t)) | left.hi;
    return r;
}
// one gap of priority p (1 .. 127) in front of the stretch F
__device__ __forceinline__ VisF vis_push(uint32_t p, const VisF& F) {
    VisF r;
    r.t = p < F.t ? p : F.t;
    r.lo = (F.lo & below_lo(p)) | (p < 64u ? 1ull << p : 0ull);
    r.hi = (F.hi & below_hi(p)) | (p >= 64u ? 1ull << (p - 64u) : 0ull);
    return r;
}
__device__ __forceinline__ uint32_t vis_count(const VisF& F) { return (uint32_t)(__popcll(F.lo) + __popcll(F.hi)); }
__device__ __forceinline__ uint64_t shfl_down_u64(uint64_t v, int o) {
    const uint32_t a = __shfl_down((uint32_t)v, o), b = __shfl_down((uint32_t)(v >> 32), o);
    return (uint64_t)a | ((uint64_t)b << 32);
}
__device__ __forceinline__ VisF vis_shfl_down(const VisF& F, int o) {
    VisF r;
    r.t = __shfl_down(F.t, o);
    r.lo = shfl_down_u64(F.lo, o);
    r.hi = shfl_down_u64(F.hi, o);
    return r;
}
// Every thread of the workgroup owns a stretch, thread 0 the leftmost.  Returns what lies to the RIGHT of the thread's
// stretch inside the workgroup; *total: the whole workgroup's stretch (valid in every thread).  THREADS / 64 <= 16.
template <int THREADS>
__device__ __forceinline__ VisF vis_block_suffix(const VisF& mine, VisF* total) {
    constexpr int kWaves = THREADS / 64;
    __shared__ VisF wave_f[kWaves];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    VisF incl = mine;  // lanes [lane, 63] of this wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const VisF other = vis_shfl_down(incl, o);
        if (lane + o < 64) incl = vis_join(incl, other);
    }
    VisF right = vis_shfl_down(incl, 1);
    if (lane == 63) right = vis_identity();
    __syncthreads();  // wave_f may still be read from an earlier call
    if (lane == 0) wave_f[wave] = incl;
    __syncthreads();
    VisF later = vis_identity(), all = vis_identity();  // the waves behind this one; all waves
    for (int w = kWaves - 1; w >= 0; w--) {
        if (w > wave) later = vis_join(wave_f[w], later);
        all = vis_join(wave_f[w], all);
    }
    *total = all;
    return vis_join(right, later);
}

constexpr int kVisItems = 8;
constexpr int kVisTile = kBlock * kVisItems;  // 2048 gaps per workgroup
constexpr uint32_t kVisFusedTiles = 2 * kBlock;  // up to this many tiles (10^6 gaps), pass 3 finds the mask behind its tile by itself

// Pass 1: priorities of a tile's gaps (written for the box climb) and the tile's stretch; a tile is kBlock * ITEMS gaps.
template <int ITEMS>
__global__ __launch_bounds__(kBlock) void vis_tile_kernel(const uint64_t* __restrict__ keys, uint32_t n, uint8_t* __restrict__ prio,
                                                          VisF* __restrict__ tile_f) {
    constexpr int kVisItems = ITEMS, kVisTile = kBlock * ITEMS;  // inside this kernel: its own tile
    __shared__ uint8_t sp[kVisTile];
    const uint32_t base = blockIdx.x * (uint32_t)kVisTile;
#pragma unroll
    for (int j = 0; j < kVisItems; j++) {  // consecutive threads on consecutive gaps
        const uint32_t g = base + (uint32_t)j * kBlock + threadIdx.x;
        uint8_t p = 0;
        if (g + 1 < n) {
            p = gap_priority_of(keys, n, g);
            prio[g] = p;
        }
        sp[j * kBlock + threadIdx.x] = p;
    }
    __syncthreads();
    VisF mine = vis_identity();
#pragma unroll
    for (int j = kVisItems - 1; j >= 0; j--) {
        const uint32_t g = base + threadIdx.x * kVisItems + (uint32_t)j;
        if (g + 1 < n) mine = vis_push(sp[threadIdx.x * kVisItems + j], mine);
    }
    VisF total;
    (void)vis_block_suffix<kBlock>(mine, &total);
    if (threadIdx.x == 0) tile_f[blockIdx.x] = total;
}

// Pass 2 (one workgroup; only for more than kVisFusedTiles tiles): the mask of the records seen from behind every
// tile's right end.  A thread takes up to kVisPerThread neighbouring tiles, loaded at once.
constexpr int kVisTilesBlock = 1024;
constexpr uint32_t kVisPerThread = 8;
__global__ __launch_bounds__(kVisTilesBlock) void vis_tiles_kernel(const VisF* __restrict__ tile_f, uint32_t tiles, ulonglong2* __restrict__ tile_in) {
    const uint32_t per = (tiles + kVisTilesBlock - 1) / kVisTilesBlock;
    const uint32_t first = threadIdx.x * per;
    VisF mine = vis_identity();
    if (per <= kVisPerThread) {
        VisF f[kVisPerThread];
#pragma unroll
        for (uint32_t k = 0; k < kVisPerThread; k++) f[k] = (k < per && first + k < tiles) ? tile_f[first + k] : vis_identity();
#pragma unroll
        for (int k = (int)kVisPerThread - 1; k >= 0; k--) mine = vis_join(f[k], mine);
        VisF total;
        VisF right = vis_block_suffix<kVisTilesBlock>(mine, &total);
#pragma unroll
        for (int k = (int)kVisPerThread - 1; k >= 0; k--) {  // right: what lies behind tile first + k
            if ((uint32_t)k < per && first + k < tiles) tile_in[first + k] = make_ulonglong2(right.lo, right.hi);  // applied to the empty mask
            right = vis_join(f[k], right);
        }
        return;
    }
    for (uint32_t k = per; k-- > 0;)
        if (first + k < tiles) mine = vis_join(tile_f[first + k], mine);
    VisF total;
    VisF right = vis_block_suffix<kVisTilesBlock>(mine, &total);
    for (uint32_t k = per; k-- > 0;) {
        if (first + k >= tiles) continue;
        tile_in[first + k] = make_ulonglong2(right.lo, right.hi);
        right = vis_join(tile_f[first + k], right);
    }
}

// Pass 3: the number of records seen from gap i, for the leaves of a tile (the last leaf has no gap: 0).
// tile_in == nullptr: at most kVisFusedTiles tiles, and the workgroup joins the tiles behind its own itself.
__global__ __launch_bounds__(kBlock) void vis_leaf_kernel(const uint8_t* __restrict__ prio, uint32_t n, const VisF* __restrict__ tile_f,
                                                          uint32_t tiles, const ulonglong2* __restrict__ tile_in, uint8_t* __restrict__ lefts) {
    const uint32_t g0 = blockIdx.x * (uint32_t)kVisTile + threadIdx.x * kVisItems;
    uint32_t p[kVisItems];
    if (g0 + kVisItems < n) {  // all eight gaps exist: one 8-byte load (prio is 256-byte aligned, g0 a multiple of 8)
        const uint2 v = *(const uint2*)(prio + g0);
#pragma unroll
        for (int j = 0; j < 4; j++) {
            p[j] = (v.x >> (8 * j)) & 0xffu;
            p[j + 4] = (v.y >> (8 * j)) & 0xffu;
        }
    } else {
#pragma unroll
        for (int j = 0; j < kVisItems; j++) p[j] = (g0 + j + 1 < n) ? prio[g0 + j] : 0u;
    }
    VisF behind;  // a function that ignores its argument: the mask behind this tile
    if (tile_in) {
        const ulonglong2 in = tile_in[blockIdx.x];
        behind = VisF{0u, in.x, in.y};
    } else {
        const uint32_t t = blockIdx.x + 1u + 2u * threadIdx.x;  // two neighbouring tiles per thread
        VisF total;
        (void)vis_block_suffix<kBlock>(vis_join(t < tiles ? tile_f[t] : vis_identity(), t + 1u < tiles ? tile_f[t + 1u] : vis_identity()), &total);
        behind = VisF{0u, total.lo, total.hi};
    }
    VisF mine = vis_identity();
#pragma unroll
    for (int j = kVisItems - 1; j >= 0; j--)
        if (g0 + j + 1 < n) mine = vis_push(p[j], mine);
    VisF total;
    VisF right = vis_join(vis_block_suffix<kBlock>(mine, &total), behind);
    uint32_t out[kVisItems];
#pragma unroll
    for (int j = kVisItems - 1; j >= 0; j--) {
        const uint32_t i = g0 + (uint32_t)j;
        if (i + 1 < n) right = vis_push(p[j], right);
        out[j] = i + 1 < n ? vis_count(right) : 0u;  // at most 127
    }
    if (g0 + kVisItems <= n) {  // one 8-byte store (lefts is 256-byte aligned, g0 a multiple of 8)
        *(uint2*)(lefts + g0) = make_uint2(out[0] | (out[1] << 8) | (out[2] << 16) | (out[3] << 24),
                                           out[4] | (out[5] << 8) | (out[6] << 16) | (out[7] << 24));
    } else {
#pragma unroll
        for (int j = 0; j < kVisItems; j++)
            if (g0 + j < n) lefts[g0 + j] = (uint8_t)out[j];
    }
}

template <bool FENCE>
__device__ __forceinline__ void child_store(ChildRec* slot, const float* box, uint32_t far_end, uint32_t lefts) {
    uint64_t w0 = (uint64_t)__float_as_uint(box[0]) | ((uint64_t)__float_as_uint(box[1]) << 32);
    uint64_t w1 = (uint64_t)__float_as_uint(box[2]) | ((uint64_t)__float_as_uint(box[3]) << 32);
    uint64_t w2 = (uint64_t)__float_as_uint(box[4]) | ((uint64_t)__float_as_uint(box[5]) << 32);
    uint64_t w3 = (uint64_t)far_end | ((uint64_t)lefts << 32);
    uint64_t* p = (uint64_t*)slot;
    if (FENCE) {
        p[0] = w0; p[1] = w1; p[2] = w2; p[3] = w3;
    } else {
        __hip_atomic_store(p + 0, w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p + 1, w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p + 2, w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p + 3, w3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
template <bool FENCE>
__device__ __forceinline__ void child_load(const ChildRec* slot, float* box, uint32_t* far_end, uint32_t* lefts) {
    uint64_t* p = (uint64_t*)slot;
    uint64_t w0, w1, w2, w3;
    if (FENCE) {
        w0 = ((volatile uint64_t*)p)[0]; w1 = ((volatile uint64_t*)p)[1];
        w2 = ((volatile uint64_t*)p)[2]; w3 = ((volatile uint64_t*)p)[3];
    } else {
        w0 = __hip_atomic_load(p + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        w1 = __hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        w2 = __hip_atomic_load(p + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        w3 = __hip_atomic_load(p + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    box[0] = __uint_as_float((uint32_t)w0); box[1] = __uint_as_float((uint32_t)(w0 >> 32));
    box[2] = __uint_as_float((uint32_t)w1); box[3] = __uint_as_float((uint32_t)(w1 >> 32));
    box[4] = __uint_as_float((uint32_t)w2); box[5] = __uint_as_float((uint32_t)(w2 >> 32));
    *far_end = (uint32_t)w3;
    *lefts = (uint32_t)(w3 >> 32);
}

// merge with a sibling: the LEFT child's box is the first argument of every Min / Max (lbvh.c:58)
__device__ __forceinline__ void merge_sibling(float* box, const float* sbox, bool sibling_is_right) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
        box[k] = sibling_is_right ? ref_min(box[k], sbox[k]) : ref_min(sbox[k], box[k]);
        box[k + 3] = sibling_is_right ? ref_max(box[k + 3], sbox[k + 3]) : ref_max(sbox[k + 3], box[k + 3]);
    }
}

// ---- K7: pre-order records ------------------------------------------------------------------
// A record in one of two forms: the documented rthip_node (min.xyz, skip | max.xyz, id), or the walk's own
// (min.x, max.x, min.y, max.y | min.z, max.z, skip, id), in which the two bounds of an axis sit in neighbouring
// registers after the load -- what the packed subtract / multiply of the slab test want (trav_step in rt_device.h
// reads it).
template <bool PAIRED>
__device__ __forceinline__ void put_node(float4* __restrict__ nodes, uint32_t pre, const float* box, uint32_t skip, uint32_t id) {
    if (PAIRED) {
        nodes[2 * (uint64_t)pre] = make_float4(box[0], box[3], box[1], box[4]);
        nodes[2 * (uint64_t)pre + 1] = make_float4(box[2], box[5], __uint_as_float(skip), __uint_as_float(id));
    } else {
        nodes[2 * (uint64_t)pre] = make_float4(box[0], box[1], box[2], __uint_as_float(skip));
        nodes[2 * (uint64_t)pre + 1] = make_float4(box[3], box[4], box[5], __uint_as_float(id));
    }
}

// the finished inner node with leaf range [l, r] that `lefts` of its ancestors hold in their left subtree
template <bool PAIRED>
__device__ __forceinline__ void put_inner(float4* __restrict__ nodes, const float* box, uint32_t l, uint32_t r, uint32_t lefts) {
    const uint32_t pre = 2u * l + lefts;
    put_node<PAIRED>(nodes, pre, box, pre + 2u * (r - l + 1u) - 1u, 0u);
}

struct ClimbArgs {
    uint32_t n;
    const uint8_t* prio;       // gap priorities
    const uint32_t* sorted_idx;
    const float* aabb6;
    ChildRec* child;           // records for parents that span chunks; the `lefts` word also carries, in its upper half, the
                               // priority of the gap beyond the child's far end (which the parent needs to find ITS parent)
    uint32_t* flags;           // arrival counters, zero before the climb
    const uint8_t* lefts;      // of every sorted leaf: how many of its ancestors hold it in their left subtree
    uint8_t* lefts_w;          // the same array for the range form's chunk kernel, which fills it in itself ...
    const VisF* chunk_f;       // ... from the stretch of every chunk's gaps (vis_tile_kernel with chunk-sized tiles)
    uint32_t* starters;        // gaps of the nodes that span chunks and got both children from the local kernel
    uint32_t* start_count;
    float4* nodes;             // the pre-order records
    // the climb above the chunks by leaf ranges (range_upper_kernel; builds of up to kOneGroupUpper leaves): no child
    // records, counters or start list, their buffers hold these instead
    float* pre_box;            // float[6n]: per leaf, the box of the leaves from its chunk's first leaf up to it
    float* suf_box;            // float[6n]: per leaf, the box of the leaves from it to its chunk's last leaf
    uint16_t* tab;             // per chunk 256 entries: [q] the rightmost window entry in 0 .. kChunk-1 with a priority below q,
                               // [128 + q] the leftmost window entry in 1 .. kChunk with a priority below q (0xffff: none)
    uint8_t* cmin;             // per chunk 2 bytes: the smallest priority among window entries 0 .. kChunk-1, and among 1 .. kChunk
    uint32_t ranges;           // 1: this form above the chunks; 2: inside the chunks as well (chunk_local_ranges)
};

#ifdef RT_CLIMB_STAMP
__device__ uint64_t g_climb_stamps[8 * 32768];  // dev build only: wall clock at the phases of the first workgroups
#define CLIMB_STAMP(k) if (threadIdx.x == 0 && blockIdx.x < 32768u) g_climb_stamps[8 * blockIdx.x + (k)] = wall_clock64()
#else
#define CLIMB_STAMP(k)
#endif

// The chunk-local part of the climb.  What a thread reads for its leaf comes in two steps: LeafIn, then the box through
// LeafIn::prim.
struct ClimbLds {
    ChildRec lrec[2 * kChunk];         // per local gap: what its left and its right child left there
    uint32_t lflag[kChunk];            // per local gap: children arrived
    uint16_t queue[3][kChunk / 2];     // the gaps (relative to c0) whose second child arrived in the generation before
    uint16_t pend[kChunk];             // gap - c0 + 1 of the parents that span chunks at which this chunk left a record
    uint16_t stage[kChunk];            // of those, the ones that this chunk found complete: for the upper kernel
    uint32_t qn[3], n_pend, n_start, range_cmin0;
    uint32_t wmin_l[kChunk / 64], wmin_r[kChunk / 64];
    // window of gap priorities: entry k is gap c0 - 1 + k, k = 0 .. kChunk; bit 7 marks a local gap.
    // Gaps that do not exist (-1, >= n-1) read as 0, below every real priority (>= 1).
    uint8_t win[kChunk + 8];
};

struct LeafIn {
    uint32_t prim;      // the sorted leaf's primitive
    uint32_t lefts;     // how many of its ancestors hold it in their left subtree
    uint32_t prio;      // of gap c0 - 1 + thread (0: there is none)
    uint32_t prio_end;  // thread 0: of gap c0 - 1 + kChunk
};

__device__ __forceinline__ LeafIn leaf_in(uint32_t c0, const ClimbArgs& A, bool shape_known = true) {
    LeafIn in = {0u, 0u, 0u, 0u};
    const uint32_t n = A.n, i = c0 + threadIdx.x;
    const int64_t g = (int64_t)c0 - 1 + threadIdx.x;
    if (g >= 0 && g + 1 < (int64_t)n) in.prio = A.prio[g];
    if (threadIdx.x == 0) {
        const int64_t ge = (int64_t)c0 - 1 + kChunk;
        if (ge + 1 < (int64_t)n) in.prio_end = A.prio[ge];
    }
    if (i < n) {
        in.prim = A.sorted_idx[i];
        if (shape_known) in.lefts = A.lefts[i];
    }
    return in;
}

__device__ __forceinline__ void leaf_box(uint32_t c0, const ClimbArgs& A, const LeafIn& in, float* box) {
    if (c0 + threadIdx.x < A.n) {
        const float* b = A.aabb6 + 6 * (uint64_t)in.prim;  // node->aabb = in_aabbs[id - 1], lbvh.c:53
#pragma unroll
        for (int k = 0; k < 6; k++) box[k] = b[k];
    }
}

// Which gaps of the chunk are local.  Ends with a barrier.
__device__ __forceinline__ void chunk_window(ClimbLds& L, const LeafIn& in, uint32_t* min_before = nullptr, uint32_t* min_after = nullptr) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0) {
        L.qn[0] = L.qn[1] = L.qn[2] = 0;
        L.n_pend = 0;
        L.n_start = 0;
        L.win[kChunk] = (uint8_t)in.prio_end;
    }
    L.lflag[tid] = 0;
    // local[k] <=> win[k] > min(win[0..k-1]) and win[k] > min(win[k+1..kChunk]), k = 1 .. kChunk-1; thread t owns entry t
    const uint32_t e = in.prio;
    uint32_t pre = e, suf = e;  // inclusive over the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t a = __shfl_up(pre, o), b = __shfl_down(suf, o);
        if (lane >= o) pre = a < pre ? a : pre;
        if (lane + o < 64) suf = b < suf ? b : suf;
    }
    if (lane == 63) L.wmin_l[wave] = pre;
    if (lane == 0) L.wmin_r[wave] = suf;
    __syncthreads();
    uint32_t before = 0xffu, after = L.win[kChunk];  // minima over the entries before / after this one
    for (int w = 0; w < kChunk / 64; w++) {
        if (w < wave) before = L.wmin_l[w] < before ? L.wmin_l[w] : before;
        if (w > wave) after = L.wmin_r[w] < after ? L.wmin_r[w] : after;
    }
    const uint32_t pl = __shfl_up(pre, 1), sr = __shfl_down(suf, 1);
    if (lane > 0) before = pl < before ? pl : before;
    if (lane < 63) after = sr < after ? sr : after;
    const bool loc = tid >= 1 && e > before && e > after;
    if (min_before) *min_before = before;  // over the window's entries 0 .. tid-1 (0xff: none)
    if (min_after) *min_after = after;     // over the entries tid+1 .. kChunk
    L.win[tid] = (uint8_t)(e | (loc ? 0x80u : 0u));
    __syncthreads();
}

// A workgroup barrier that orders LDS accesses only.  __syncthreads() also waits for the wave's outstanding global
// stores (vmcnt(0)): inside the generation loop that is a round trip to HBM per generation for stores nobody reads here.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Appends `value` of the lanes that `want` to a list in LDS: one atomic per wave.  Call from wave-uniform control flow.
__device__ __forceinline__ void wave_push(bool want, uint32_t value, uint32_t* count, uint16_t* list) {
    const uint64_t mask = __ballot(want);
    if (mask == 0) return;
    const uint32_t lane = threadIdx.x & 63u;
    const int leader = __ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if ((int)lane == leader) base = atomicAdd(count, (uint32_t)__popcll(mask));
    base = __shfl(base, leader);
    if (want) list[base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)value;
}

// A finished node [l, r] goes to its parent: the one of the two gaps that bound it with the larger priority.  Inside
// the chunk it leaves its record in the parent's slot, and the second child to arrive puts the parent on the next
// generation's queue; at a parent that spans chunks it leaves the record in HBM and notes the parent for the end of
// the chunk.  Every lane of the wave calls this (the lists are appended to per wave); `active` says which have a node.
// pr_l, pr_r: the window's entries for the gaps in front of leaf l and behind leaf r (0: there is none); a record left
// inside the chunk carries the entry of the gap beyond its far end, so that only the leaves read the window.
__device__ __forceinline__ void arrive(ClimbLds& L, uint32_t c0, const ClimbArgs& A, bool active, uint32_t l, uint32_t r,
                                       uint32_t lefts, uint32_t pr_l, uint32_t pr_r, const float* box, uint32_t next_q) {
    const uint32_t n = A.n;
    bool queued = false, noted = false;
    uint32_t where = 0;
    if (active && !(l == 0 && r == n - 1)) {  // not the root
        const bool parent_right = (l == 0) || (r != n - 1 && (pr_r & 0x7fu) > (pr_l & 0x7fu));
        const uint32_t parent = parent_right ? r : l - 1;
        const uint32_t side = parent_right ? 0u : 1u;  // 0: this node is the LEFT child
        const bool local = ((parent_right ? pr_r : pr_l) & 0x80u) != 0;
        ChildRec mine;
#pragma unroll
        for (int k = 0; k < 6; k++) mine.box[k] = box[k];
        mine.far_end = parent_right ? l : r;
        if (local) {
            where = parent - c0;
            mine.lefts = lefts | ((parent_right ? pr_l : pr_r) << 16);
            L.lrec[2 * where + side] = mine;
            queued = atomicAdd(&L.lflag[where], 1u) == 1u;  // the records are read after the next barrier
        } else if (!A.ranges) {
            const uint32_t far_prio = (parent_right ? pr_l : pr_r) & 0x7fu;  // of the gap beyond the far end
            mine.lefts = lefts | (far_prio << 16);
            A.child[2 * (uint64_t)parent + side] = mine;
            where = parent + 1u - c0;
            noted = true;
        }  // (with ranges the nodes above the chunks take their boxes from the leaves' prefix / suffix boxes)
    }
    wave_push(queued, where, &L.qn[next_q], L.queue[next_q]);
    wave_push(noted, where, &L.n_pend, L.pend);
}

// The climb of one chunk from its leaves' boxes, generation by generation: the nodes whose second child arrived in one
// generation are finished in the next by neighbouring lanes of the first waves, the other waves wait at the barrier.
// (A thread that climbs from its leaf for as long as it is the second to arrive leaves all eight waves running with one
// or two lanes each for most of the chunk's height: 13-17 us of a chunk's 20-27 us instead of 10-15.)  A chunk of
// random triangles has 12 generations on average, 184, 115, 76, 50, ... nodes each, and a generation takes a microsecond
// whatever its size: a couple of hundred dependent instructions and half a dozen LDS round trips of one wave.
template <bool PAIRED>
__device__ __forceinline__ void chunk_climb(ClimbLds& L, uint32_t c0, const ClimbArgs& A, const LeafIn& in, float* box) {
    const uint32_t n = A.n;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    {
        const uint32_t i = c0 + tid;
        if (i < n) {
            const uint32_t pre = 2u * i + in.lefts;
            put_node<PAIRED>(A.nodes, pre, box, pre + 1, in.prim + 1);  // id = idx + 1, lbvh.c:116
        }
        // while a node is inside the chunk both gaps that bound it are inside the window
        const uint32_t pr_l = (i < n && i != 0) ? (uint32_t)L.win[tid] : 0u;
        const uint32_t pr_r = (i + 1 < n) ? (uint32_t)L.win[tid + 1] : 0u;
        arrive(L, c0, A, i < n, i, i, in.lefts, pr_l, pr_r, box, 0u);
    }
#ifdef RT_CLIMB_STAMP
    uint64_t gens = 0, first = 0;
#endif
    for (uint32_t cur = 0, nxt = 1, clr = 2;;) {
        lds_barrier();
#ifdef RT_CLIMB_STAMP
        if (gens == 0) first = wall_clock64();
        gens++;
        if (threadIdx.x == 0 && blockIdx.x < 32768u) { g_climb_stamps[8 * blockIdx.x + 6] = gens; g_climb_stamps[8 * blockIdx.x + 7] = first; }
#endif
        const uint32_t count = L.qn[cur];  // at most kChunk / 2: the nodes of a generation are disjoint
        const uint32_t slot = L.queue[cur][tid & (kChunk / 2 - 1)];  // asked for together with the count
        if (count == 0) break;
        if (tid == 0) L.qn[clr] = 0;  // last read in the generation before this one
        uint32_t l = 0, r = 0, lefts = 0, pr_l = 0, pr_r = 0;
        if (tid < count) {
            const ChildRec a = L.lrec[2 * slot], b = L.lrec[2 * slot + 1];  // left, right
#pragma unroll
            for (int k = 0; k < 6; k++) box[k] = a.box[k];
            merge_sibling(box, b.box, true);
            l = a.far_end;
            r = b.far_end;
            lefts = (a.lefts & 0xffffu) - 1u;  // the parent sits directly in front of its left child
            pr_l = a.lefts >> 16;
            pr_r = b.lefts >> 16;
            put_inner<PAIRED>(A.nodes, box, l, r, lefts);
        }
        arrive(L, c0, A, tid < count, l, r, lefts, pr_l, pr_r, box, nxt);
        const uint32_t t = cur;
        cur = nxt;
        nxt = clr;
        clr = t;
    }
    CLIMB_STAMP(3);
    if (A.ranges) return;
    // The parents that span chunks: count this chunk's records there (all at once: a counter in HBM takes microseconds
    // to answer); where it was the second to arrive the parent is complete and goes on the upper kernel's list, with
    // one device-wide atomic per chunk.
    {
        const uint32_t pending = L.n_pend;  // at most kChunk
        bool complete = false;
        uint32_t where = 0;
        if (tid < pending) {
            where = L.pend[tid];
            complete = atomicAdd(&A.flags[c0 + where - 1u], 1u) == 1u;
        }
        wave_push(complete, where, &L.n_start, L.stage);
    }
    lds_barrier();
    const uint32_t total = L.n_start;
    if (tid < 64u && total != 0) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(A.start_count, total);
        base = __shfl(base, 0);
        for (uint32_t k = lane; k < total; k += 64u) A.starters[base + k] = c0 + L.stage[k] - 1u;
    }
}

// The chunk's part of the list of gaps above the chunks goes to memory: one device-wide atomic per chunk, whose answer
// takes a microsecond or two -- only wave 0 waits for it.  After a barrier behind chunk_ranges' last wave_push.
// chunk_ranges_claim asks for the chunk's place in the list (thread 0; the answer is first used in chunk_ranges_flush, so
// whatever lies between the two calls runs while it is on its way).
__device__ __forceinline__ uint32_t chunk_ranges_claim(ClimbLds& L, uint32_t c0, const ClimbArgs& A) {
    const uint64_t end_gap = (uint64_t)c0 - 1u + kChunk;
    uint32_t base = 0;
    if (threadIdx.x == 0u) base = atomicAdd(A.start_count, L.n_start + (end_gap + 1u < A.n ? 1u : 0u));
    return base;
}
__device__ __forceinline__ void chunk_ranges_flush(ClimbLds& L, uint32_t c0, const ClimbArgs& A, uint32_t base) {
    const uint32_t n = A.n, tid = threadIdx.x, lane = tid & 63u;
    const uint32_t e_end = (uint32_t)L.win[kChunk] & 0x7fu;
    const uint32_t cmin0 = L.range_cmin0;
    const uint32_t total = L.n_start;
    const uint64_t end_gap = (uint64_t)c0 - 1u + kChunk;
    const bool end_exists = end_gap + 1u < n;
    if (tid < 64u) {
        base = __shfl(base, 0);
        for (uint32_t k = lane; k < total; k += 64u) {
            const uint32_t v = L.stage[k], entry = v & 0x3fffu;
            A.starters[base + k] = (c0 - 1u + entry) | (((uint32_t)L.win[entry] & 0x7fu) << 17) | ((v & 0xc000u) << 16);
        }
        if (lane == 0u && end_exists) A.starters[base + total] = (uint32_t)end_gap | (e_end << 17) | (cmin0 < e_end ? 0x40000000u : 0u);
    }
}

// What the range form of the climb above the chunks needs from a chunk (see range_upper_kernel): every leaf's prefix
// and suffix box inside the chunk, the chunk's two tables of "nearest gap with a priority below q", and the list of
// the chunk's gaps whose nodes reach beyond it.  Window entry k is gap c0 - 1 + k (chunk_window); a gap beyond the
// array's ends has priority 0, below every real one, and so ends every search.  Call after chunk_window, by all threads.
__device__ __forceinline__ void merge_after(float* acc, const float* later) { merge_sibling(acc, later, true); }
__device__ __forceinline__ void chunk_ranges(ClimbLds& L, uint32_t c0, const ClimbArgs& A, const float* box, uint32_t min_before,
                                             uint32_t min_after, bool flush_now) {
    const uint32_t n = A.n, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t chunk = c0 / (uint32_t)kChunk;
    const float inf = __uint_as_float(0x7f800000u);
    __shared__ float wtot[2][kChunk / 64][6];  // per wave: box of its leaves, for the prefix and for the suffix scan
    __shared__ uint16_t tab[256];
    __shared__ uint32_t bestA[128], bestB[128], cmins[2];
    // ---- boxes: inclusive scans in leaf order, the earlier leaf always the first operand
    float mine[6], pre[6], suf[6];
#pragma unroll
    for (int k = 0; k < 6; k++) mine[k] = (c0 + tid < n) ? box[k] : (k < 3 ? inf : -inf);
#pragma unroll
    for (int k = 0; k < 6; k++) pre[k] = suf[k] = mine[k];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float up[6], down[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            up[k] = __shfl_up(pre[k], o);
            down[k] = __shfl_down(suf[k], o);
        }
        if (lane >= (uint32_t)o) {  // pre = (leaves before) then (pre)
            merge_after(up, pre);
#pragma unroll
            for (int k = 0; k < 6; k++) pre[k] = up[k];
        }
        if (lane + (uint32_t)o < 64u) merge_after(suf, down);
    }
    if (lane == 63u)
#pragma unroll
        for (int k = 0; k < 6; k++) wtot[0][wave][k] = pre[k];
    if (lane == 0u)
#pragma unroll
        for (int k = 0; k < 6; k++) wtot[1][wave][k] = suf[k];
    if (tid < 128u) {
        bestA[tid] = 0u;
        bestB[tid] = 0xffffu;
    }
    if (tid < 2u) cmins[tid] = 0xffu;
    __syncthreads();
    {
        float acc[6];
#pragma unroll
        for (int k = 0; k < 6; k++) acc[k] = k < 3 ? inf : -inf;
        for (uint32_t w = 0; w < wave; w++) merge_after(acc, wtot[0][w]);
        merge_after(acc, pre);
#pragma unroll
        for (int k = 0; k < 6; k++) pre[k] = acc[k];
        for (uint32_t w = wave + 1; w < (uint32_t)kChunk / 64u; w++) merge_after(suf, wtot[1][w]);
    }
    if (c0 + tid < n) {
#pragma unroll
        for (int k = 0; k < 6; k++) {
            A.pre_box[6 * (uint64_t)(c0 + tid) + k] = pre[k];
            A.suf_box[6 * (uint64_t)(c0 + tid) + k] = suf[k];
        }
    }
    // ---- tables.  Thread t looks at window entry t (0 .. kChunk-1); entry kChunk has no thread, thread 0 sees to it.
    // rightmost entry below q = the largest position among the entries with priority + 1 <= q: every entry marks
    // best[priority + 1] with its position, a prefix maximum over the 128 slots does the rest (leftmost: minimum).
    const uint32_t e = (uint32_t)L.win[tid] & 0x7fu;          // entry tid
    const uint32_t e_end = (uint32_t)L.win[kChunk] & 0x7fu;  // entry kChunk
    if (e < 127u) {
        atomicMax(&bestA[e + 1u], tid + 1u);                       // among entries 0 .. kChunk-1; 0 = none
        if (tid >= 1u) atomicMin(&bestB[e + 1u], tid);             // among entries 1 .. kChunk; 0xffff = none
    }
    if (tid == 0u && e_end < 127u) atomicMin(&bestB[e_end + 1u], (uint32_t)kChunk);
    __syncthreads();
    if (wave < 2u) {  // wave 0: prefix maximum of bestA, wave 1: prefix minimum of bestB, two slots per lane
        uint32_t* best = wave == 0u ? bestA : bestB;
        uint32_t a = best[2u * lane], b = best[2u * lane + 1u];
        b = wave == 0u ? (a > b ? a : b) : (a < b ? a : b);
        uint32_t run = b;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(run, o);
            if (lane >= (uint32_t)o) run = wave == 0u ? (u > run ? u : run) : (u < run ? u : run);
        }
        const uint32_t prev = __shfl_up(run, 1);
        if (lane > 0u) a = wave == 0u ? (prev > a ? prev : a) : (prev < a ? prev : a);
        if (wave == 0u) {
            tab[2u * lane] = a ? (uint16_t)(a - 1u) : (uint16_t)0xffffu;
            tab[2u * lane + 1u] = run ? (uint16_t)(run - 1u) : (uint16_t)0xffffu;
        } else {
            tab[128u + 2u * lane] = (uint16_t)a;
            tab[128u + 2u * lane + 1u] = (uint16_t)run;
        }
    }
    // the smallest priorities: among the entries 0 .. kChunk-1, and among 1 .. kChunk
    uint32_t mA = e, mB = tid == 0u ? e_end : e;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const uint32_t x = __shfl_xor(mA, o), y = __shfl_xor(mB, o);
        mA = x < mA ? x : mA;
        mB = y < mB ? y : mB;
    }
    if (lane == 0u) {
        atomicMin(&cmins[0], mA);
        atomicMin(&cmins[1], mB);
    }
    __syncthreads();
    if (tid < 256u) A.tab[256u * chunk + tid] = tab[tid];
    if (tid < 2u) A.cmin[2u * chunk + tid] = (uint8_t)cmins[tid];
    if (tid == 0u) L.range_cmin0 = cmins[0];  // for chunk_ranges_flush
    // ---- the gaps whose nodes reach beyond the chunk: entries 1 .. kChunk-1 that are not local, and entry kChunk.
    // List word: the gap, its priority in bits 17-23, bit 30: a smaller priority among the entries before it in the
    // window, bit 31: among those after it.
    {
        const uint64_t gap = (uint64_t)c0 - 1u + tid;
        const bool exists = tid >= 1u && gap + 1u < n;
        const bool upper = exists && (L.win[tid] & 0x80u) == 0u;
        wave_push(upper, tid | (min_before < e ? 0x4000u : 0u) | (min_after < e ? 0x8000u : 0u), &L.n_start, L.stage);
    }
    if (flush_now) {  // otherwise the caller flushes, behind a barrier of its own
        lds_barrier();
        chunk_ranges_flush(L, c0, A, chunk_ranges_claim(L, c0, A));
        lds_barrier();
        if (tid == 0u) L.n_start = 0;
    }
}

// The nodes INSIDE a chunk by leaf ranges as well (instead of chunk_climb's dozen generations of hand-offs): one thread
// per gap of the window.  A local gap's node covers the leaves between the nearest entries of smaller priority on either
// side; its box is the reduction of those leaves' boxes in leaf order, which has the bits of the nested merges however it
// is bracketed (merge_after keeps the later operand on ties, as the reference's Min / Max do); its place follows from
// the shape byte of its last leaf.
//   * nearest smaller entry: inside the wave from ballots, one round per distinct priority among the wave's local gaps;
//     beyond the wave from per-wave tables "rightmost / leftmost entry with a priority below q" (marked with LDS atomics,
//     finished by a prefix maximum / minimum over q), looked up wave by wave;
//   * boxes: per leaf the box from its block of 8's first leaf up to it and from it to the block's last leaf (segmented
//     scans by shuffles), a sparse table over the 64 blocks' boxes (built by wave 0 with shuffles); a range is one
//     suffix, at most two table entries and one prefix, or up to 8 leaves of one block.
// The leaves' boxes, the prefixes and the waves' tables lie in ClimbLds::lrec, which this form does not use.
struct LocalRangeLds {
    float suf8[kChunk][6];       // leaf i .. the last leaf of its block
    float sp[5][kChunk / 8][6];  // sp[k - 1][b]: blocks b .. b + 2^k - 1
    float bpre[kChunk / 8][6];   // blocks 0 .. b
    float bsuf[kChunk / 8][6];   // blocks b .. 63
    uint8_t lefts[kChunk];
    uint32_t cmin1;              // the smallest priority among the window's entries 1 .. kChunk
};
static_assert(sizeof(ChildRec) * 2 * kChunk >= 2 * kChunk * 6 * sizeof(float) + 2 * (kChunk / 64) * 128 * sizeof(uint32_t), "lrec holds the tables");

template <bool PAIRED>
__device__ __forceinline__ void chunk_local_ranges(ClimbLds& L, LocalRangeLds& R, uint32_t c0, const ClimbArgs& A, const LeafIn& in,
                                                   const float* box, uint32_t min_before, uint32_t min_after) {
    const uint32_t n = A.n, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t chunk = c0 / (uint32_t)kChunk;
    float(*lb)[6] = reinterpret_cast<float(*)[6]>(L.lrec);
    float(*pre8)[6] = lb + kChunk;  // the block's first leaf .. leaf i
    uint32_t(*near_r)[128] = reinterpret_cast<uint32_t(*)[128]>(pre8 + kChunk);  // [wave][q]: lane + 1 of the rightmost entry below q (0: none)
    uint32_t(*near_l)[128] = near_r + kChunk / 64;                                // [wave][q]: lane of the leftmost (0xff: none)
    const float inf = __uint_as_float(0x7f800000u);
    const uint32_t i = c0 + tid;
    if (i < n) {
        const uint32_t pre = 2u * i + in.lefts;
        put_node<PAIRED>(A.nodes, pre, box, pre + 1, in.prim + 1);  // id = idx + 1, lbvh.c:116
    }
    {
        float p8[6], s8[6];
#pragma unroll
        for (int k = 0; k < 6; k++) p8[k] = s8[k] = (i < n) ? box[k] : (k < 3 ? inf : -inf);
#pragma unroll
        for (int k = 0; k < 6; k++) lb[tid][k] = p8[k];
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            float up[6], down[6];
#pragma unroll
            for (int k = 0; k < 6; k++) {
                up[k] = __shfl_up(p8[k], o);
                down[k] = __shfl_down(s8[k], o);
            }
            if ((lane & 7u) >= (uint32_t)o) {
                merge_after(up, p8);
#pragma unroll
                for (int k = 0; k < 6; k++) p8[k] = up[k];
            }
            if ((lane & 7u) + (uint32_t)o < 8u) merge_after(s8, down);
        }
#pragma unroll
        for (int k = 0; k < 6; k++) {
            pre8[tid][k] = p8[k];
            R.suf8[tid][k] = s8[k];
        }
        R.lefts[tid] = (uint8_t)in.lefts;
    }
    // ---- the wave's tables
    const uint32_t e = (uint32_t)L.win[tid] & 0x7fu;  // entry tid: gap c0 - 1 + tid (0: there is none)
    const bool local = (L.win[tid] & 0x80u) != 0u;
    near_r[wave][lane] = near_r[wave][lane + 64u] = 0u;
    near_l[wave][lane] = near_l[wave][lane + 64u] = 0xffu;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    if (e < 127u) {
        atomicMax(&near_r[wave][e + 1u], lane + 1u);
        if (tid != 0u) atomicMin(&near_l[wave][e + 1u], lane);  // entry 0 is nobody's bound on the right (and not in the chunk's table)
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    {   // prefix maximum / minimum over q, two slots per lane
        uint32_t a = near_r[wave][2u * lane], b = near_r[wave][2u * lane + 1u];
        uint32_t c = near_l[wave][2u * lane], d = near_l[wave][2u * lane + 1u];
        b = a > b ? a : b;
        d = c < d ? c : d;
        uint32_t run_r = b, run_l = d;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(run_r, o), v = __shfl_up(run_l, o);
            if (lane >= (uint32_t)o) {
                run_r = u > run_r ? u : run_r;
                run_l = v < run_l ? v : run_l;
            }
        }
        const uint32_t prev_r = __shfl_up(run_r, 1), prev_l = __shfl_up(run_l, 1);
        if (lane > 0u) {
            a = prev_r > a ? prev_r : a;
            c = prev_l < c ? prev_l : c;
        }
        near_r[wave][2u * lane] = a;
        near_r[wave][2u * lane + 1u] = run_r;
        near_l[wave][2u * lane] = c;
        near_l[wave][2u * lane + 1u] = run_l;
    }
    // ---- nearest smaller entries inside the wave
    uint32_t jl = 0xffffu, jr = 0xffffu;
    {
        const uint64_t lt = (1ull << lane) - 1ull, gt = ~lt & ~(1ull << lane);
        uint64_t todo = __ballot(local);
        while (todo != 0ull) {
            const uint32_t q = (uint32_t)__builtin_amdgcn_readlane((int)e, __ffsll((long long)todo) - 1);  // todo is uniform
            const uint64_t smaller = __ballot(e < q), same = __ballot(e == q);
            if (local && e == q) {
                const uint64_t lo = smaller & lt, hi = smaller & gt;
                if (lo != 0ull) jl = wave * 64u + 63u - (uint32_t)__clzll((long long)lo);
                if (hi != 0ull) jr = wave * 64u + (uint32_t)__ffsll((long long)hi) - 1u;
            }
            todo &= ~same;
        }
    }
    {   // the gaps whose nodes reach beyond the chunk go on the list for range_upper_kernel (see chunk_ranges)
        const uint64_t gap = (uint64_t)c0 - 1u + tid;
        const bool upper = tid >= 1u && gap + 1u < n && !local;
        wave_push(upper, tid | (min_before < e ? 0x4000u : 0u) | (min_after < e ? 0x8000u : 0u), &L.n_start, L.stage);
        if (tid == 0u) R.cmin1 = min_after;
    }
    CLIMB_STAMP(6);
    lds_barrier();
    const uint32_t list_base = chunk_ranges_claim(L, c0, A);  // the list is complete behind this barrier
    // What range_upper_kernel needs from the chunk, from the same tables: waves 1 and 2 scan the blocks' boxes from the
    // left and from the right (a leaf's box up to / from the chunk's end is its block's part and one of these), wave 3
    // joins the waves' nearest-entry tables into the chunk's two and notes the chunk's smallest priorities.
    if (wave == 1u || wave == 2u) {
        float cur[6];
#pragma unroll
        for (int k = 0; k < 6; k++) cur[k] = pre8[8u * lane + 7u][k];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            float other[6];
#pragma unroll
            for (int k = 0; k < 6; k++) other[k] = wave == 1u ? __shfl_up(cur[k], o) : __shfl_down(cur[k], o);
            if (wave == 1u) {
                if (lane >= (uint32_t)o) {
                    merge_after(other, cur);
#pragma unroll
                    for (int k = 0; k < 6; k++) cur[k] = other[k];
                }
            } else if (lane + (uint32_t)o < 64u) {
                merge_after(cur, other);
            }
        }
#pragma unroll
        for (int k = 0; k < 6; k++) (wave == 1u ? R.bpre : R.bsuf)[lane][k] = cur[k];
    }
    if (wave == 3u) {
        const uint32_t e_end = (uint32_t)L.win[kChunk] & 0x7fu;
#pragma unroll
        for (uint32_t half = 0; half < 2u; half++) {
            const uint32_t q = lane + 64u * half;
            uint32_t right = 0xffffu, left = 0xffffu;
            for (int w = kChunk / 64 - 1; w >= 0; w--) {
                const uint32_t v = near_r[w][q];
                if (v != 0u && right == 0xffffu) right = (uint32_t)w * 64u + v - 1u;
            }
            for (uint32_t w = 0; w < (uint32_t)kChunk / 64u; w++) {
                const uint32_t v = near_l[w][q];
                if (v != 0xffu && left == 0xffffu) left = w * 64u + v;
            }
            if (left == 0xffffu && e_end < q) left = (uint32_t)kChunk;
            A.tab[256u * chunk + q] = (uint16_t)right;         // among the entries 0 .. kChunk - 1
            A.tab[256u * chunk + 128u + q] = (uint16_t)left;   // among the entries 1 .. kChunk
        }
        if (lane == 0u) {
            uint32_t m = 0xffu;
            for (uint32_t w = 0; w < (uint32_t)kChunk / 64u; w++) m = L.wmin_l[w] < m ? L.wmin_l[w] : m;
            L.range_cmin0 = m;
            A.cmin[2u * chunk] = (uint8_t)m;
            A.cmin[2u * chunk + 1u] = (uint8_t)R.cmin1;
        }
    }
    if (wave == 0u) {  // the sparse table over the blocks' boxes: lane b holds blocks b .. b + 2^k - 1
        float cur[6];
#pragma unroll
        for (int k = 0; k < 6; k++) cur[k] = pre8[8u * lane + 7u][k];
#pragma unroll
        for (int k = 1; k <= 5; k++) {
            float other[6];
#pragma unroll
            for (int c = 0; c < 6; c++) other[c] = __shfl_down(cur[c], 1 << (k - 1));
            if (lane + (1u << k) <= 64u) {
                merge_after(cur, other);
#pragma unroll
                for (int c = 0; c < 6; c++) R.sp[k - 1][lane][c] = cur[c];
            }
        }
    }
    if (local) {  // beyond the wave; a local gap has both of its bounds inside the window
        if (jl == 0xffffu) {
            for (int w = (int)wave - 1; w >= 0; w--) {
                const uint32_t v = near_r[w][e];
                if (v != 0u) {
                    jl = (uint32_t)w * 64u + v - 1u;
                    break;
                }
            }
        }
        if (jr == 0xffffu) {
            jr = (uint32_t)kChunk;
            for (uint32_t w = wave + 1u; w < (uint32_t)kChunk / 64u; w++) {
                const uint32_t v = near_l[w][e];
                if (v != 0xffu) {
                    jr = w * 64u + v;
                    break;
                }
            }
        }
    }
    lds_barrier();
    CLIMB_STAMP(7);
    if (i < n) {  // for range_upper_kernel: the box of the leaves from the chunk's first up to this one, and from this one to the last
        const uint32_t b = tid >> 3;
        float pre[6], suf[6];
#pragma unroll
        for (int k = 0; k < 6; k++) {
            pre[k] = pre8[tid][k];
            suf[k] = R.suf8[tid][k];
        }
        if (b > 0u) {
            float acc[6];
#pragma unroll
            for (int k = 0; k < 6; k++) acc[k] = R.bpre[b - 1u][k];
            merge_after(acc, pre);
#pragma unroll
            for (int k = 0; k < 6; k++) pre[k] = acc[k];
        }
        if (b + 1u < (uint32_t)kChunk / 8u) merge_after(suf, R.bsuf[b + 1u]);
#pragma unroll
        for (int k = 0; k < 6; k++) {
            A.pre_box[6 * (uint64_t)i + k] = pre[k];
            A.suf_box[6 * (uint64_t)i + k] = suf[k];
        }
    }
    if (local) {
        const uint32_t l = jl, r = jr - 1u;  // the leaf behind gap jl .. the leaf in front of gap jr, relative to c0
        const uint32_t bl = l >> 3, br = r >> 3;
        float b[6];
        if (bl == br) {
#pragma unroll
            for (int k = 0; k < 6; k++) b[k] = lb[l][k];
            for (uint32_t t = l + 1u; t <= r; t++) merge_after(b, lb[t]);
        } else {
#pragma unroll
            for (int k = 0; k < 6; k++) b[k] = R.suf8[l][k];
            if (br > bl + 1u) {  // whole blocks bl + 1 .. br - 1
                const uint32_t count = br - bl - 1u, k = 31u - (uint32_t)__clz(count);
                const float* first = k == 0u ? pre8[8u * (bl + 1u) + 7u] : R.sp[k - 1u][bl + 1u];
                const float* second = k == 0u ? pre8[8u * (br - 1u) + 7u] : R.sp[k - 1u][br - (1u << k)];
                merge_after(b, first);
                merge_after(b, second);
            }
            merge_after(b, pre8[r]);
        }
        put_inner<PAIRED>(A.nodes, b, c0 + l, c0 + r, (uint32_t)R.lefts[r]);
    }
    chunk_ranges_flush(L, c0, A, list_base);
}

template <bool PAIRED, bool RANGES = false>
__global__ __launch_bounds__(kChunk) void climb_local_kernel(ClimbArgs A) {
    __shared__ ClimbLds L;
    const uint32_t c0 = blockIdx.x * (uint32_t)kChunk;
    CLIMB_STAMP(0);
    LeafIn in = leaf_in(c0, A, !RANGES);
    float box[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if constexpr (RANGES) {
        leaf_box(c0, A, in, box);  // asked for before the scan's barriers, used after them
        // The last step of the shape scan for this chunk's leaves (vis_leaf_kernel's, with chunks for tiles): the mask
        // seen from behind the chunk is the join of the stretches of all later chunks, at most kChunk - 1 of them.
        const uint32_t i = c0 + threadIdx.x, n = A.n;
        const bool has_gap = i + 1u < n;
        const uint32_t p = has_gap ? (uint32_t)A.prio[i] : 0u;
        // Two scans share their barriers: over the later chunks' stretches (only their join is needed; waves whose
        // chunks do not exist skip it) and over this chunk's gaps.
        __shared__ VisF wave_own[kChunk / 64], wave_later[kChunk / 64];
        const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        const uint32_t t = blockIdx.x + 1u + threadIdx.x;
        VisF own = has_gap ? vis_push(p, vis_identity()) : vis_identity();  // lanes [lane, 63] of this wave
        VisF later = vis_identity();
        const bool any_later = blockIdx.x + 1u + 64u * wave < gridDim.x;    // uniform over the wave
        if (any_later && t < gridDim.x) later = A.chunk_f[t];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const VisF other = vis_shfl_down(own, o);
            if (lane + (uint32_t)o < 64u) own = vis_join(own, other);
            if (any_later) {
                const VisF other_later = vis_shfl_down(later, o);
                if (lane + (uint32_t)o < 64u) later = vis_join(later, other_later);
            }
        }
        VisF right = vis_shfl_down(own, 1);
        if (lane == 63u) right = vis_identity();
        if (lane == 0u) {
            wave_own[wave] = own;
            wave_later[wave] = later;
        }
        __syncthreads();
        VisF behind = vis_identity();  // everything behind this wave's gaps: the later waves, then the later chunks
        for (int w = kChunk / 64 - 1; w >= 0; w--) behind = vis_join(wave_later[w], behind);
        behind = VisF{0u, behind.lo, behind.hi};  // applied to the empty mask
        for (int w = kChunk / 64 - 1; w > (int)wave; w--) behind = vis_join(wave_own[w], behind);
        right = vis_join(right, behind);
        if (has_gap) right = vis_push(p, right);
        in.lefts = has_gap ? vis_count(right) : 0u;
        if (i < n) A.lefts_w[i] = (uint8_t)in.lefts;
    }
    uint32_t min_before = 0xffu, min_after = 0xffu;
    chunk_window(L, in, &min_before, &min_after);
    CLIMB_STAMP(1);
    if constexpr (!RANGES) leaf_box(c0, A, in, box);
#ifdef RT_CLIMB_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CLIMB_STAMP(2);
#endif
    if constexpr (RANGES) {  // their LDS exists in this instance only
        __shared__ LocalRangeLds R;
        CLIMB_STAMP(3);
        if (A.ranges == 2u) {  // everything from leaf ranges: the chunk's own nodes, and what range_upper_kernel needs from it
            chunk_local_ranges<PAIRED>(L, R, c0, A, in, box, min_before, min_after);
            CLIMB_STAMP(4);
#ifdef RT_CLIMB_STAMP
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            CLIMB_STAMP(5);
#endif
            return;
        }
        chunk_ranges(L, c0, A, box, min_before, min_after, true);
    }
    chunk_climb<PAIRED>(L, c0, A, in, box);
    CLIMB_STAMP(4);
#ifdef RT_CLIMB_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CLIMB_STAMP(5);
#endif
}

#ifdef RT_CLIMB_STAMP
}  // namespace
extern "C" int rthip_debug_climb_stamps(uint64_t* host, size_t n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_climb_stamps), n * sizeof(uint64_t));
}
namespace {
#endif

// MODE 3: the whole upper climb in ONE workgroup, for builds of up to kOneGroupUpper leaves, where the upper tree is a
// couple of thousand nodes and the chain of hand-offs is all there is: all threads sit on one CU, so the hand-off is a
// workgroup-scope release / acquire pair through the counter, which costs no cache maintenance.  Measured: 81 920
// leaves 35 us instead of 41 us device-wide; at 10^6 leaves (20 000 upper nodes on 1024 threads) 301 us instead of
// 60 us, hence the limit.
constexpr uint64_t kOneGroupUpper = 1ull << 17;
constexpr int kOneGroupThreads = 1024;

// The climb above the chunks: one thread per complete node on the list.
template <int MODE, bool PAIRED>
__global__ __launch_bounds__(MODE == 3 ? kOneGroupThreads : kBlock) void climb_upper_kernel(ClimbArgs A) {
    const uint32_t n = A.n;
    const uint32_t count = *A.start_count;
    for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < count; t += gridDim.x * blockDim.x) {
        uint32_t parent = A.starters[t];
        const ChildRec a = A.child[2 * (uint64_t)parent], b = A.child[2 * (uint64_t)parent + 1];  // left, right
        uint32_t l = a.far_end, r = b.far_end, lefts = (a.lefts & 0xffffu) - 1u;
        uint32_t pl = a.lefts >> 16, pr = b.lefts >> 16;  // priorities of gaps l - 1 and r (0: there is none)
        float box[6];
#pragma unroll
        for (int k = 0; k < 6; k++) box[k] = a.box[k];
        merge_sibling(box, b.box, true);
        for (;;) {
            put_inner<PAIRED>(A.nodes, box, l, r, lefts);
            if (l == 0 && r == n - 1) break;
            const bool parent_right = (l == 0) || (r != n - 1 && pr > pl);
            parent = parent_right ? r : l - 1;
            const uint32_t side = parent_right ? 0u : 1u;
            child_store<MODE == 1 || MODE == 3>(&A.child[2 * (uint64_t)parent + side], box, parent_right ? l : r,
                                                lefts | ((parent_right ? pl : pr) << 16));
            uint32_t old;
            if (MODE == 3) {
                old = __hip_atomic_fetch_add(&A.flags[parent], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else if (MODE == 0) {  // release: the record is visible device-wide before the count
                old = __hip_atomic_fetch_add(&A.flags[parent], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (MODE == 1) __threadfence();
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                old = atomicAdd(&A.flags[parent], 1u);
            }
            if (old == 0u) break;  // first to arrive: the sibling carries on from here
            if (MODE == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // the sibling's record, released before its count
            if (MODE == 1) __threadfence();
            uint32_t s_far, s_lefts;
            float sbox[6];
            child_load<MODE == 1 || MODE == 3>(&A.child[2 * (uint64_t)parent + (1u - side)], sbox, &s_far, &s_lefts);
            merge_sibling(box, sbox, side == 0);
            if (side == 0) {
                r = s_far;
                lefts = lefts - 1;
                pr = s_lefts >> 16;
            } else {
                l = s_far;
                lefts = (s_lefts & 0xffffu) - 1;
                pl = s_lefts >> 16;
            }
        }
    }
}

// The climb above the chunks WITHOUT a chain of hand-offs, for builds of up to kOneGroupUpper leaves (one workgroup).
// A node's box is the box of its leaf range, and the reference's Min / Max (common_core.h:113-114: the right operand
// on ties) return the rightmost of the equal extremes of a range however the reduction is bracketed, so any reduction
// that keeps the leaves in order gives the bits of the reference's nested merges (lbvh.c:58).  The node that splits at
// gap g reaches from the nearest gap of smaller priority on its left to the nearest one on its right: found through a
// sparse table of the chunks' minimum priorities and the chunks' own tables (chunk_ranges); its box is (suffix box of
// its first leaf) + (whole chunks between, from a sparse table of chunk boxes) + (prefix box of its last leaf); its place
// is 2 l + S(r), S(r) = how many ancestors hold leaf r in their left subtree (the node's own ancestors of that kind are
// exactly those: the running-minimum records seen from gap r).  One thread per node, nothing waits for anything.
constexpr int kRangeLevels = 9;  // 2^8 = 256 chunks of 512 leaves = kOneGroupUpper
template <bool PAIRED>
__global__ __launch_bounds__(kOneGroupThreads) void range_upper_kernel(ClimbArgs A) {
    __shared__ uint8_t minA[kRangeLevels][256], minB[kRangeLevels][256];
    __shared__ float cbox[kRangeLevels - 1][256][6];
    const uint32_t n = A.n, tid = threadIdx.x;
    const uint32_t chunks = (n + (uint32_t)kChunk - 1u) / (uint32_t)kChunk;  // <= 256
    const uint32_t first = blockIdx.x * (uint32_t)kOneGroupThreads + tid;
    // asked for before the tables are made, and together with the count (the list has room for n + 1 words; what lies
    // behind the count is not used)
    uint32_t word = A.starters[first < n ? first : n];
    const uint32_t count = *A.start_count;
    if (tid < chunks) {
        minA[0][tid] = A.cmin[2u * tid];
        minB[0][tid] = A.cmin[2u * tid + 1u];
        const uint32_t last = ((tid + 1u) * (uint32_t)kChunk < n ? (tid + 1u) * (uint32_t)kChunk : n) - 1u;
#pragma unroll
        for (int k = 0; k < 6; k++) cbox[0][tid][k] = A.pre_box[6 * (uint64_t)last + k];
    }
    __syncthreads();
    for (int lv = 1; lv < kRangeLevels && (1u << lv) <= chunks; lv++) {
        const uint32_t half = 1u << (lv - 1);
        if (tid < chunks && tid + 2u * half <= chunks) {
            const uint8_t a = minA[lv - 1][tid], b = minA[lv - 1][tid + half], c = minB[lv - 1][tid], d = minB[lv - 1][tid + half];
            minA[lv][tid] = a < b ? a : b;
            minB[lv][tid] = c < d ? c : d;
            if (lv < kRangeLevels - 1) {
                float box[6];
#pragma unroll
                for (int k = 0; k < 6; k++) box[k] = cbox[lv - 1][tid][k];
                merge_sibling(box, cbox[lv - 1][tid + half], true);
#pragma unroll
                for (int k = 0; k < 6; k++) cbox[lv][tid][k] = box[k];
            }
        }
        __syncthreads();
    }
    for (uint32_t t = first; t < count; t += gridDim.x * (uint32_t)kOneGroupThreads) {
        if (t != first) word = A.starters[t];
        const uint32_t g = word & 0x1ffffu;
        const bool left_in = (word & 0x40000000u) != 0u, right_in = (word & 0x80000000u) != 0u;
        if (g + 1u >= n) continue;
        const uint32_t q = (word >> 17) & 0x7fu;
        const uint32_t c = g / (uint32_t)kChunk;  // the chunk in whose window g is one of the entries 1 .. kChunk
        // left end: the nearest gap before g with a priority below q (gap -1 counts, with priority 0)
        uint32_t cl = c;
        if (!left_in) {  // none in this chunk's window: the nearest chunk before it whose entries 0 .. kChunk-1 hold one
            uint32_t pos = c;
            for (int lv = kRangeLevels - 1; lv >= 0; lv--) {
                const uint32_t step = 1u << lv;
                if (step <= chunks && pos >= step && minA[lv][pos - step] >= q) pos -= step;
            }
            if (pos == 0u) continue;  // cannot happen: chunk 0 holds gap -1
            cl = pos - 1u;
        }
        const uint32_t le = A.tab[256u * cl + q];
        // right end: the nearest gap behind g with a priority below q (gap n-1 counts, with priority 0)
        uint32_t cr = c;
        if (!right_in) {
            uint32_t pos = c + 1u;
            for (int lv = kRangeLevels - 1; lv >= 0; lv--) {
                const uint32_t step = 1u << lv;
                if (pos + step <= chunks && minB[lv][pos] >= q) pos += step;
            }
            if (pos >= chunks) continue;  // cannot happen: the last chunk holds gap n-1
            cr = pos;
        }
        const uint32_t re = A.tab[256u * cr + 128u + q];
        if (le == 0xffffu || re == 0xffffu) continue;  // cannot happen
        const uint32_t l = cl * (uint32_t)kChunk + le;               // gap cl*kChunk - 1 + le, plus one
        uint32_t r = cr * (uint32_t)kChunk - 1u + re;                // the gap itself: the node's last leaf
        if (r > n - 1u) r = n - 1u;
        if (!(l <= g && g < r)) continue;  // cannot happen
        const uint32_t first_chunk = l / (uint32_t)kChunk, last_chunk = r / (uint32_t)kChunk;
        if (last_chunk <= first_chunk) continue;  // cannot happen: such a node lies inside a chunk
        float box[6];
#pragma unroll
        for (int k = 0; k < 6; k++) box[k] = A.suf_box[6 * (uint64_t)l + k];
        if (last_chunk > first_chunk + 1u) {  // whole chunks between: two overlapping power-of-two stretches
            const uint32_t a = first_chunk + 1u, b = last_chunk - 1u;
            const int lv = 31 - __clz((int)(b - a + 1u));
            merge_sibling(box, cbox[lv][a], true);
            merge_sibling(box, cbox[lv][b + 1u - (1u << lv)], true);
        }
        if (last_chunk > first_chunk) {
            float tail[6];
#pragma unroll
            for (int k = 0; k < 6; k++) tail[k] = A.pre_box[6 * (uint64_t)r + k];
            merge_sibling(box, tail, true);
        }
        const uint32_t lefts = r + 1u < n ? (uint32_t)A.lefts[r] : 0u;
        put_inner<PAIRED>(A.nodes, box, l, r, lefts);
    }
}

// ---- the whole build in one workgroup, for n <= kChunk ---------------------------------------
// Small trees (the demos' meshes and TLASes have 1 .. 12 primitives) are all launch overhead in the
// pipeline above.  Here one 512-thread workgroup, one primitive per thread, runs the same steps with
// the same device functions: centroid bounds, Morton codes, a stable sort in LDS (bitonic network on
// (code, index) pairs), gap priorities, every leaf's S (the scan of the running-minimum records, here one gap per
// thread) and the box climb that writes the pre-order records.
template <bool PAIRED>
__global__ __launch_bounds__(kChunk) void lbvh_small_kernel(uint32_t n, const float* __restrict__ aabb6,
                                                            uint64_t* __restrict__ codes_unsorted,
                                                            uint64_t* __restrict__ keys_sorted,
                                                            uint32_t* __restrict__ vals_sorted, ClimbArgs A,
                                                            uint8_t* __restrict__ prio_w, uint8_t* __restrict__ lefts_w,
                                                            uint32_t box_bounds) {
    __shared__ uint64_t skey[kChunk];
    __shared__ uint32_t sidx[kChunk];
    __shared__ uint32_t sbounds[6];
    const uint32_t t = threadIdx.x;

    if (t < 3) sbounds[t] = f2ord(3.402823466e+38f);
    else if (t < 6) sbounds[t] = f2ord(-3.402823466e+38f);
    __syncthreads();
    float c[3] = {0.f, 0.f, 0.f};
    if (t < n) {
        const float* b = aabb6 + 6 * (uint64_t)t;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            c[k] = (b[k] + b[k + 3]) * 0.5f;  // lbvh.c:98
            atomicMin(&sbounds[k], f2ord(box_bounds ? b[k] : c[k]));  // a BUILD_DEBUG build bounds the boxes, lbvh.c:101-104
            atomicMax(&sbounds[k + 3], f2ord(box_bounds ? b[k + 3] : c[k]));
        }
    }
    __syncthreads();
    uint64_t code = ~0ull;
    if (t < n) {
        code = 0;
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float mn = ord2f(sbounds[k]);
            const float ext = ord2f(sbounds[k + 3]) - mn;  // extents = max - min, lbvh.c:106
            code |= spread21(morton_axis(c[k], mn, ext)) << k;
        }
        codes_unsorted[t] = code;
    }
    skey[t] = code;
    sidx[t] = (t < n) ? t : 0xffffffffu;
    __syncthreads();
    // ascending by (code, index): equal codes keep ascending primitive index, like the stable radix passes
    for (uint32_t k = 2; k <= (uint32_t)kChunk; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            const uint32_t other = t ^ j;
            if (other > t) {
                const uint64_t ka = skey[t], kb = skey[other];
                const uint32_t ia = sidx[t], ib = sidx[other];
                const bool a_after_b = ka > kb || (ka == kb && ia > ib);
                const bool ascending = (t & k) == 0;
                if (a_after_b == ascending) {
                    skey[t] = kb;
                    skey[other] = ka;
                    sidx[t] = ib;
                    sidx[other] = ia;
                }
            }
            __syncthreads();
        }
    }
    if (t < n) {
        keys_sorted[t] = skey[t];
        vals_sorted[t] = sidx[t];
    }
    {   // S(t) = t + the records of the running minimum of the priorities that starts at gap t
        uint32_t p = 0;
        VisF mine = vis_identity();
        if (t + 1 < n) {
            p = gap_priority_of(skey, n, t);
            prio_w[t] = (uint8_t)p;
            mine = vis_push(p, mine);
        }
        VisF total;
        const VisF right = vis_block_suffix<kChunk>(mine, &total);
        if (t + 1 < n) mine = vis_push(p, right);
        if (t < n) lefts_w[t] = (uint8_t)(t + 1 < n ? vis_count(mine) : 0u);
    }
    __syncthreads();  // the climb reads prio, lefts and vals_sorted back from memory
    __shared__ ClimbLds L;
    const LeafIn in = leaf_in(0u, A);
    float box[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    leaf_box(0u, A, in, box);
    chunk_window(L, in);
    chunk_climb<PAIRED>(L, 0u, A, in, box);
}

// ---- K8: world box of every instance, rt_cpu_tlas_node_to_aabb, raytracer_cpu.c:131-147 ------
__device__ __forceinline__ void rot_quat(const float p[3], const float q[4], float out[3]) {  // common_math.c:91
    float qq = q[0] * q[0] + q[1] * q[1] + q[2] * q[2];
    float s0 = q[3] * q[3] - qq;
    float qp = q[0] * p[0] + q[1] * p[1] + q[2] * p[2];
    float s1 = 2.f * qp;
    float cx = q[1] * p[2] - q[2] * p[1];
    float cy = -(q[0] * p[2] - q[2] * p[0]);
    float cz = q[0] * p[1] - q[1] * p[0];
    float s2 = 2.f * q[3];
    out[0] = (p[0] * s0 + q[0] * s1) + cx * s2;
    out[1] = (p[1] * s0 + q[1] * s1) + cy * s2;
    out[2] = (p[2] * s0 + q[2] * s1) + cz * s2;
}

__global__ __launch_bounds__(kBlock) void instance_aabb_kernel(const DevInstance* __restrict__ inst, uint32_t n,
                                                               const float4* __restrict__ blas,
                                                               float* __restrict__ aabb6) {
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const DevInstance it = inst[i];
    float* o = aabb6 + 6 * (uint64_t)i;
    if (it.type == RTHIP_INST_SPHERE) {
        float rr = it.sphere_radius;
        for (int k = 0; k < 3; k++) {
            o[k] = it.sphere_center[k] - rr;
            o[k + 3] = it.sphere_center[k] + rr;
        }
        return;
    }
    const float4 ra = blas[2 * (uint64_t)it.node_base], rb = blas[2 * (uint64_t)it.node_base + 1];  // root record, the walk's form
    float bmin[3] = {ra.x, ra.z, rb.x}, bmax[3] = {ra.y, ra.w, rb.y};
    float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
    float mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    for (int c = 0; c < 8; c++) {  // corner order of rt_cpu_transform_aabb, raytracer_cpu.c:584-593
        float p[3] = {(c & 4) ? bmax[0] : bmin[0], (c & 2) ? bmax[1] : bmin[1], (c & 1) ? bmax[2] : bmin[2]};
        float ps[3] = {p[0] * it.scale[0], p[1] * it.scale[1], p[2] * it.scale[2]};
        float w[3];
        rot_quat(ps, it.rotation, w);
        for (int k = 0; k < 3; k++) {
            float v = w[k] + it.translation[k];
            mn[k] = ref_min(mn[k], v);
            mx[k] = ref_max(mx[k], v);
        }
    }
    for (int k = 0; k < 3; k++) {
        o[k] = mn[k];
        o[k + 3] = mx[k];
    }
}

// ---- records of the ordered (fast) traversal -------------------------------------------------------------------
// The ordered walk reads a 4-wide tree collapsed from the finished pre-order binary tree (the walk's record form):
// an inner node at EVEN depth becomes a record that holds the boxes of its grandchildren (of a child that is a leaf,
// the child itself), 2 to 4 slots in pre-order; inner nodes at odd depth have no record.  Every box is a box of the
// binary tree, bit for bit, and a grandchild's box lies inside its parent's, so the walk reaches exactly the primitives
// the binary tree would lead it to.  Records are numbered in pre-order among themselves (root = 0).
//
// One record = 8 float4 = one 128-byte line: slot k's {min.x, max.x, min.y, max.y} in [k], the z bounds
// {min.z0, max.z0, min.z1, max.z1} {min.z2, max.z2, min.z3, max.z3} in [4] [5], the references in [6] (record index,
// 0x80000000 | id for a primitive, 0xffffffff for an empty slot, whose box is a point at +infinity), and in [7] each
// slot's node index in the binary tree (read only by a walk that has run out of stack and goes through a slot's
// subtree in the binary tree's own order).
//
// Depth without a top-down pass: in pre-order, depth(i) = (inner nodes before i) - (inner nodes whose subtree has
// ended before or at i), so every inner node j adds 1 at j + 1 and takes 1 away at skip(j), and a prefix sum gives the depths.
__global__ __launch_bounds__(kBlock) void wide_depth_marks_kernel(const float4* __restrict__ nodes, uint32_t n_nodes,
                                                                  uint32_t* __restrict__ marks) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_nodes) return;
    const float4 mine = nodes[2 * (uint64_t)i + 1];
    if (__float_as_uint(mine.w) != 0u) return;
    atomicAdd(&marks[i + 1], 1u);
    atomicSub(&marks[__float_as_uint(mine.z)], 1u);  // skip(i) <= n_nodes: marks has n_nodes + 1 words
}

// depth -> "has a record" (inner and at even depth), in place; the word behind the last node stays 0
__global__ __launch_bounds__(kBlock) void wide_flags_kernel(const float4* __restrict__ nodes, uint32_t n_nodes,
                                                            uint32_t* __restrict__ depth_to_flag) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i > n_nodes) return;
    uint32_t f = 0u;
    if (i < n_nodes) f = (__float_as_uint(nodes[2 * (uint64_t)i + 1].w) == 0u && (depth_to_flag[i] & 1u) == 0u) ? 1u : 0u;
    depth_to_flag[i] = f;
}

__global__ __launch_bounds__(kBlock) void wide_emit_kernel(const float4* __restrict__ nodes, uint32_t n_nodes,
                                                           const uint32_t* __restrict__ rank, float4* __restrict__ wide) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_nodes) return;
    const uint32_t my_rank = rank[i];
    if (rank[i + 1] == my_rank) return;  // no record: a leaf, or an inner node at odd depth
    // The slots' nodes, left to right: a child that is a leaf stands for itself, an inner child for its two children.
    // (Written with selects and fixed slots: an array filled at a running position would live in scratch memory.)
    constexpr uint32_t kNone = 0xffffffffu;
    const uint32_t c0 = i + 1;
    const float4 c0b = nodes[2 * (uint64_t)c0 + 1];
    const uint32_t c1 = __float_as_uint(c0b.z);  // the right child starts where the left child's subtree ends
    const float4 c1b = nodes[2 * (uint64_t)c1 + 1];
    const bool in0 = __float_as_uint(c0b.w) == 0u, in1 = __float_as_uint(c1b.w) == 0u;
    const uint32_t l0 = c0 + 1, l1 = c1 + 1;  // left grandchildren (read only under an inner child)
    const uint32_t r0 = in0 ? __float_as_uint(nodes[2 * (uint64_t)l0 + 1].z) : kNone;
    const uint32_t r1 = in1 ? __float_as_uint(nodes[2 * (uint64_t)l1 + 1].z) : kNone;
    const uint32_t first1 = in1 ? l1 : c1;  // the first slot of the right child
    uint32_t node[4];
    node[0] = in0 ? l0 : c0;
    node[1] = in0 ? r0 : first1;
    node[2] = in0 ? first1 : (in1 ? r1 : kNone);
    node[3] = (in0 && in1) ? r1 : kNone;
    const float inf = __uint_as_float(0x7f800000u);
    float4 xy[4];
    float zlo[4], zhi[4];
    uint32_t ref[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        xy[k] = make_float4(inf, inf, inf, inf);
        zlo[k] = zhi[k] = inf;
        ref[k] = kNone;
        if (node[k] != kNone) {
            const float4 a = nodes[2 * (uint64_t)node[k]], b = nodes[2 * (uint64_t)node[k] + 1];
            const uint32_t id = __float_as_uint(b.w);
            xy[k] = a;
            zlo[k] = b.x;
            zhi[k] = b.y;
            ref[k] = id ? (0x80000000u | id) : rank[node[k]];
        } else {
            node[k] = 0u;
        }
    }
    float4* out = wide + 8 * (uint64_t)my_rank;
    out[0] = xy[0];
    out[1] = xy[1];
    out[2] = xy[2];
    out[3] = xy[3];
    out[4] = make_float4(zlo[0], zhi[0], zlo[1], zhi[1]);
    out[5] = make_float4(zlo[2], zhi[2], zlo[3], zhi[3]);
    out[6] = make_float4(__uint_as_float(ref[0]), __uint_as_float(ref[1]), __uint_as_float(ref[2]), __uint_as_float(ref[3]));
    out[7] = make_float4(__uint_as_float(node[0]), __uint_as_float(node[1]), __uint_as_float(node[2]), __uint_as_float(node[3]));
}

constexpr uint32_t div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// exclusive / inclusive prefix sum of n uint32 in place; tmp holds the block sums of every level
template <bool INCLUSIVE>
int scan_u32(rthip_ctx* ctx, uint32_t* d, uint64_t n, uint32_t* tmp, hipStream_t s) {
    uint32_t blocks = div_up(n, kScanTile);
    if (blocks <= 1) {
        scan_tile_kernel<INCLUSIVE><<<1, kBlock, 0, s>>>(d, n, nullptr);
        RT_HIP(ctx, hipGetLastError());
        return RTHIP_OK;
    }
    scan_tile_kernel<INCLUSIVE><<<blocks, kBlock, 0, s>>>(d, n, tmp);
    RT_HIP(ctx, hipGetLastError());
    RT_TRY((scan_u32<false>(ctx, tmp, blocks, tmp + ((blocks + 63) & ~63u), s)));
    scan_add_kernel<<<blocks, kBlock, 0, s>>>(d, n, tmp);
    RT_HIP(ctx, hipGetLastError());
    return RTHIP_OK;
}

// the shape scan's records: a mask and a stretch per tile of kVisTile gaps, or a stretch per chunk (range form)
constexpr size_t vis_bytes(uint64_t n) {
    const size_t tiles = (size_t)div_up(n, kVisTile) * (sizeof(ulonglong2) + sizeof(VisF)), chunks = (size_t)div_up(n, kChunk) * sizeof(VisF);
    return tiles > chunks ? tiles : chunks;
}

size_t scan_tmp_elems(uint64_t n) {
    size_t total = 0;
    while (n > (uint64_t)kScanTile) {
        uint64_t blocks = (n + kScanTile - 1) / kScanTile;
        total += (blocks + 63) & ~63ull;
        n = blocks;
    }
    return total + 64;
}

// ---- the self-check of the fast hand-off (handoff_self_check) ------------------------------------------------------------
// boxes of a scene that makes the upper climb deep and uneven: clusters of different densities plus a uniform rest
__global__ __launch_bounds__(kBlock) void check_boxes_kernel(float* __restrict__ aabb6, uint32_t n) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    auto mix = [](uint32_t x) {
        x ^= x >> 16; x *= 0x7feb352du;
        x ^= x >> 15; x *= 0x846ca68bu;
        x ^= x >> 16;
        return x;
    };
    const uint32_t a = mix(3u * i + 1u), b = mix(3u * i + 2u), c = mix(3u * i + 3u);
    float x = (float)(a >> 8) * (1.f / 16777216.f), y = (float)(b >> 8) * (1.f / 16777216.f), z = (float)(c >> 8) * (1.f / 16777216.f);
    const uint32_t cluster = mix(i / 97u) & 15u;
    if (cluster < 9u) {  // most boxes lie in small clusters: long runs of neighbouring codes, subtrees of very different heights
        const uint32_t k = mix(i / 97u + 77u);
        const float s = cluster < 3u ? 1e-4f : 1e-2f;
        x = (float)(k & 1023u) * (1.f / 1024.f) + x * s;
        y = (float)((k >> 10) & 1023u) * (1.f / 1024.f) + y * s;
        z = (float)((k >> 20) & 1023u) * (1.f / 1024.f) + z * s;
    }
    float* o = aabb6 + 6 * (size_t)i;
    o[0] = x - 1e-4f; o[1] = y - 1e-4f; o[2] = z - 1e-4f;
    o[3] = x + 1e-4f; o[4] = y + 1e-4f; o[5] = z + 1e-4f;
}

__global__ __launch_bounds__(kBlock) void check_compare_kernel(const uint4* __restrict__ a, const uint4* __restrict__ b, uint64_t n16,
                                                               uint32_t* __restrict__ differ) {
    const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n16) return;
    const uint4 x = a[i], y = b[i];
    if (x.x != y.x || x.y != y.y || x.z != y.z || x.w != y.w) atomicOr(differ, 1u);
}

}  // namespace

// Keys per tile of the sort in tiles.  Measured, whole build of random boxes, tiles of 1024 / 2048 / 4096 keys: 3 000 boxes
// 55 / 60 / 60 us, 10 000 59 / 61 / 71, 30 000 77 / 66 / 76, 81 920 82 / 77 / 83, 131 072 114 / 93 / 92: small tiles
// spread the tile sort over more CUs, large ones leave the merge fewer tiles to rank against.
static uint32_t merge_tile_keys(const rthip_ctx* ctx, uint64_t n) {
    return ctx->sort_merge_tile ? ctx->sort_merge_tile : (n <= 16384 ? 1024u : 2048u);
}

// whether a sort of n eight-byte keys can make its keys itself, as the Morton codes of boxes (radix_sort_workspace_device's codes_of)
static bool sort_makes_codes(const rthip_ctx* ctx, uint64_t n) {
    return n <= ctx->sort_merge_max && !ctx->sort_by_passes && merge_tile_keys(ctx, n) <= 2048;
}

int radix_sort_workspace_device(rthip_ctx* ctx, uint64_t n, int n_bytes, hipStream_t s, int* cur_io, LbvhWorkspace* ws,
                                bool identity_vals, int first_byte, const MortonSource* codes_of) {
    LbvhWorkspace& w = ws ? *ws : ctx->ws;
    // Sizes just above the merge path sort in tiles of 1024 keys, which spreads a pass over more CUs.  Measured (whole
    // build, 1024- against 4096-key tiles): 200 000 keys 0.224 / 0.241 ms, 500 000 0.274 / 0.276, 10^6 0.369 / 0.334,
    // 4 * 10^6 1.02 / 0.83 -- the four times larger tile histogram soon costs more than the spread gains.
    const bool small_tile = n <= ctx->sort_small_tile_max;
    const uint32_t tiles = small_tile ? div_up(n, kSmallTile) : div_up(n, kSortTile);
    RT_TRY(rthip_reserve(ctx, w.block_hist, (256ull * tiles + 256) * 4));
    uint64_t* keys[2] = {(uint64_t*)w.keys[0].p, (uint64_t*)w.keys[1].p};
    uint32_t* vals[2] = {(uint32_t*)w.vals[0].p, (uint32_t*)w.vals[1].p};
    int cur = *cur_io;
    if (n <= ctx->sort_merge_max && !ctx->sort_by_passes && first_byte == 0) {
        // tiles sorted in LDS, then merged by rank; eight-byte keys by their upper four bytes first (tile_sort_kernel)
        const uint32_t tile_keys = merge_tile_keys(ctx, n);
        const uint32_t merge_tiles = div_up(n, tile_keys);
        const int top_first = (n_bytes == 8 && ctx->sort_top_bytes == 4) ? 4 : 0;
        const uint32_t* v_in = identity_vals ? nullptr : vals[cur];
        if (codes_of) {
            if (!(sort_makes_codes(ctx, n) && n_bytes == 8 && identity_vals)) return rthip_fail(ctx, RTHIP_E_INVALID, "sort: codes of boxes asked for where the sort cannot make them");
            if (tile_keys == 1024)
                tile_sort_kernel<256, 4, true><<<merge_tiles, 256, 0, s>>>(nullptr, nullptr, keys[cur ^ 1], vals[cur ^ 1], n, n_bytes, top_first,
                                                                           codes_of->aabb6, codes_of->bounds, codes_of->zero_word);
            else
                tile_sort_kernel<512, 4, true><<<merge_tiles, 512, 0, s>>>(nullptr, nullptr, keys[cur ^ 1], vals[cur ^ 1], n, n_bytes, top_first,
                                                                           codes_of->aabb6, codes_of->bounds, codes_of->zero_word);
        } else if (tile_keys == 1024) tile_sort_kernel<256, 4><<<merge_tiles, 256, 0, s>>>(keys[cur], v_in, keys[cur ^ 1], vals[cur ^ 1], n, n_bytes, top_first);
        else if (tile_keys == 2048) tile_sort_kernel<512, 4><<<merge_tiles, 512, 0, s>>>(keys[cur], v_in, keys[cur ^ 1], vals[cur ^ 1], n, n_bytes, top_first);
        else tile_sort_kernel<512, 8><<<merge_tiles, 512, 0, s>>>(keys[cur], v_in, keys[cur ^ 1], vals[cur ^ 1], n, n_bytes, top_first);
        RT_HIP(ctx, hipGetLastError());
        cur ^= 1;
        if (merge_tiles > 1) {
            const uint64_t mask = n_bytes >= 8 ? ~0ull : ((1ull << (8 * n_bytes)) - 1ull);
            tile_merge_kernel<<<div_up(n, 64), 256, 0, s>>>(keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], (uint32_t)n, merge_tiles, tile_keys, mask);
            RT_HIP(ctx, hipGetLastError());
            cur ^= 1;
        }
        *cur_io = cur;
        return RTHIP_OK;
    }
    if (codes_of && !(n_bytes == 8 && identity_vals)) return rthip_fail(ctx, RTHIP_E_INVALID, "sort: codes of boxes asked for where the sort cannot make them");
    // A one-sweep form (digit totals of all passes from one pass over the keys, tile offsets by decoupled look-back over
    // per-tile status words instead of the histogram and offsets kernels) was built and measured at 10^7 keys: 94 us
    // per pass against 61 + 18 + 6 here -- with ~600 tiles in flight a tile's predecessors have published their own
    // counts but not yet their inclusive ones, so every look-back walks far, one device-scope load latency per step.
    for (int pass = first_byte; pass < n_bytes; pass++) {
        const int shift = pass * 8;
        uint32_t* hist = (uint32_t*)w.block_hist.p;
        const uint32_t* v_in = (pass == first_byte && identity_vals) ? nullptr : vals[cur];
        if (pass == first_byte && codes_of) {  // the keys are made on the way
            if (small_tile) radix_hist_kernel<kSmallSortBlock, true><<<tiles, kSmallSortBlock, 0, s>>>(keys[cur], n, shift, hist, tiles, *codes_of);
            else radix_hist_kernel<kSortBlock, true><<<tiles, kSortBlock, 0, s>>>(keys[cur], n, shift, hist, tiles, *codes_of);
        } else if (small_tile) radix_hist_kernel<kSmallSortBlock><<<tiles, kSmallSortBlock, 0, s>>>(keys[cur], n, shift, hist, tiles);
        else radix_hist_kernel<kSortBlock><<<tiles, kSortBlock, 0, s>>>(keys[cur], n, shift, hist, tiles);
        RT_HIP(ctx, hipGetLastError());
        radix_offsets_kernel<<<256, kBlock, 0, s>>>(hist, tiles);
        if (small_tile)
            radix_scatter_kernel<kSmallSortBlock><<<tiles, kSmallSortBlock, 0, s>>>(keys[cur], v_in, keys[cur ^ 1], vals[cur ^ 1], n, shift, hist, tiles);
        else
            radix_scatter_kernel<kSortBlock><<<tiles, kSortBlock, 0, s>>>(keys[cur], v_in, keys[cur ^ 1], vals[cur ^ 1], n, shift, hist, tiles);
        RT_HIP(ctx, hipGetLastError());
        cur ^= 1;
    }
    *cur_io = cur;
    return RTHIP_OK;
}

int wide_convert_device(rthip_ctx* ctx, const float4* d_nodes, uint32_t n_nodes, float4* d_wide, hipStream_t s) {
    if (n_nodes < 3) return RTHIP_OK;  // a single leaf has no record
    RT_TRY(rthip_reserve(ctx, ctx->wide_rank, ((size_t)n_nodes + 1) * 4));
    RT_TRY(rthip_reserve(ctx, ctx->ws.scan_aux, scan_tmp_elems((uint64_t)n_nodes + 1) * 4));
    uint32_t* rank = (uint32_t*)ctx->wide_rank.p;
    const uint32_t blocks = div_up((uint64_t)n_nodes + 1, kBlock);
    RT_HIP(ctx, hipMemsetAsync(rank, 0, ((size_t)n_nodes + 1) * 4, s));
    wide_depth_marks_kernel<<<blocks, kBlock, 0, s>>>(d_nodes, n_nodes, rank);
    RT_HIP(ctx, hipGetLastError());
    RT_TRY((scan_u32<true>(ctx, rank, (uint64_t)n_nodes + 1, (uint32_t*)ctx->ws.scan_aux.p, s)));
    wide_flags_kernel<<<blocks, kBlock, 0, s>>>(d_nodes, n_nodes, rank);
    RT_HIP(ctx, hipGetLastError());
    RT_TRY((scan_u32<false>(ctx, rank, (uint64_t)n_nodes + 1, (uint32_t*)ctx->ws.scan_aux.p, s)));
    wide_emit_kernel<<<blocks, kBlock, 0, s>>>(d_nodes, n_nodes, rank, d_wide);
    RT_HIP(ctx, hipGetLastError());
    return RTHIP_OK;
}

int scan_inclusive_u32_device(rthip_ctx* ctx, uint32_t* d, uint64_t n, hipStream_t s) {
    RT_TRY(rthip_reserve(ctx, ctx->ws.scan_aux, scan_tmp_elems(n) * 4));
    return scan_u32<true>(ctx, d, n, (uint32_t*)ctx->ws.scan_aux.p, s);
}

int deinterleave_device(rthip_ctx* ctx, const void* d_interleaved, uint32_t vertex_count, uint32_t stride,
                        uint32_t off, float* x, float* y, float* z, hipStream_t stream) {
    if (vertex_count == 0) return RTHIP_OK;
    deinterleave_kernel<<<div_up(vertex_count, kBlock), kBlock, 0, stream>>>((const uint8_t*)d_interleaved,
                                                                              vertex_count, stride, off, x, y, z);
    RT_HIP(ctx, hipGetLastError());
    return RTHIP_OK;
}

int mesh_prepare_device(rthip_ctx* ctx, const float* px, const float* py, const float* pz, const float* nx,
                        const float* ny, const float* nz, const uint32_t* indices, uint64_t n_tris,
                        float* d_aabb6, float4* d_tri_v, float4* d_tri_n, hipStream_t stream) {
    if (n_tris == 0) return RTHIP_OK;
    RT_TRY(rthip_reserve(ctx, ctx->ws.bounds, kBoundBytes));
    uint32_t* bounds = (uint32_t*)ctx->ws.bounds.p;
    bounds_init_kernel<<<1, kBoundSlots * 8, 0, stream>>>(bounds);
    uint32_t blocks = div_up(n_tris, kBlock);
    if (blocks > 4096) blocks = 4096;
    tri_prepare_kernel<<<blocks, kBlock, 0, stream>>>(px, py, pz, nx, ny, nz, indices, n_tris, d_aabb6, d_tri_v, d_tri_n,
                                                      bounds, ctx->box_bounds);
    RT_HIP(ctx, hipGetLastError());
    // the centroid bounds of exactly these boxes are now in the workspace: the build that follows on this
    // stream for this array can skip its own pass over them
    ctx->bounds_of = d_aabb6;
    ctx->bounds_n = n_tris;
    ctx->bounds_stream = stream;
    return RTHIP_OK;
}

int instance_aabbs_device(rthip_ctx* ctx, const DevInstance* d_inst, uint32_t n, const float4* d_blas,
                          float* d_aabb6, hipStream_t stream) {
    if (n == 0) return RTHIP_OK;
    instance_aabb_kernel<<<div_up(n, kBlock), kBlock, 0, stream>>>(d_inst, n, d_blas, d_aabb6);
    RT_HIP(ctx, hipGetLastError());
    return RTHIP_OK;
}

int lbvh_reserve_workspace(rthip_ctx* ctx, uint64_t n, LbvhWorkspace* ws) {
    LbvhWorkspace& w = ws ? *ws : ctx->ws;
    const uint32_t tiles = n <= ctx->sort_small_tile_max ? div_up(n, kSmallTile) : div_up(n, kSortTile);
    const uint64_t hist_elems = 256ull * tiles;
    RT_TRY(rthip_reserve(ctx, w.keys[0], n * 8));

    RT_TRY(rthip_reserve(ctx, w.keys[1], n * 8));
    RT_TRY(rthip_reserve(ctx, w.vals[0], n * 4));
    RT_TRY(rthip_reserve(ctx, w.vals[1], n * 4));
    RT_TRY(rthip_reserve(ctx, w.block_hist, (hist_elems + 256) * 4));
    RT_TRY(rthip_reserve(ctx, w.prio, n + 8));
    RT_TRY(rthip_reserve(ctx, w.child, 2 * n * sizeof(ChildRec)));
    RT_TRY(rthip_reserve(ctx, w.vis, vis_bytes(n)));
    RT_TRY(rthip_reserve(ctx, w.flags, n * 4));
    RT_TRY(rthip_reserve(ctx, w.starters, (n + 1) * 4));  // the start list of the upper kernel (at most n / 2 nodes), or every gap above the chunks (range form)
    RT_TRY(rthip_reserve(ctx, w.lefts, n + 32));
    RT_TRY(rthip_reserve(ctx, w.bounds, kBoundBytes));
    return RTHIP_OK;
}

// The default hand-off of the climb above the chunks (MODE 2 of climb_upper_kernel) is ordered by what gfx950 does, not by
// what the HIP memory model promises.  Before a context uses it for the first time it builds one tree of 400 000 boxes
// both ways, on its own stream and in a workspace of its own (about 90 MB, released again), and compares every record:
// equal -> MODE 2 stays; different, or anything fails on the way -> the model's own release / acquire form from then on.
// Once per context; the calling thread waits for it (a few milliseconds).
static int handoff_self_check(rthip_ctx* ctx) {
    ctx->refit_checked = true;  // whatever comes of it: not again (and the two builds below must not come back here)
    constexpr uint32_t n = 400000;
    const uint64_t node_bytes = (2ull * n - 1ull) * 32ull;
    DevBuf boxes, nodes_a, nodes_b, flag;
    LbvhWorkspace ws;
    int rc = RTHIP_OK;
    uint32_t differ = 1;
    const auto run = [&]() -> int {
        RT_TRY(rthip_reserve(ctx, boxes, (size_t)n * 24));
        RT_TRY(rthip_reserve(ctx, nodes_a, node_bytes));
        RT_TRY(rthip_reserve(ctx, nodes_b, node_bytes));
        RT_TRY(rthip_reserve(ctx, flag, 256));
        hipStream_t s = ctx->stream;
        RT_HIP(ctx, hipMemsetAsync(flag.p, 0, 256, s));
        check_boxes_kernel<<<div_up(n, kBlock), kBlock, 0, s>>>((float*)boxes.p, n);
        RT_HIP(ctx, hipGetLastError());
        ctx->refit_fence = 2;
        RT_TRY(lbvh_build_device(ctx, (const float*)boxes.p, n, (float4*)nodes_a.p, s, nullptr, nullptr, nullptr, true, &ws));
        ctx->refit_fence = 0;
        RT_TRY(lbvh_build_device(ctx, (const float*)boxes.p, n, (float4*)nodes_b.p, s, nullptr, nullptr, nullptr, true, &ws));
        check_compare_kernel<<<div_up(node_bytes / 16, kBlock), kBlock, 0, s>>>((const uint4*)nodes_a.p, (const uint4*)nodes_b.p, node_bytes / 16,
                                                                                (uint32_t*)flag.p);
        RT_HIP(ctx, hipGetLastError());
        RT_HIP(ctx, hipMemcpyAsync(&differ, flag.p, 4, hipMemcpyDeviceToHost, s));
        RT_HIP(ctx, hipStreamSynchronize(s));
        return RTHIP_OK;
    };
    rc = run();
    for (DevBuf* b : {&boxes, &nodes_a, &nodes_b, &flag, &ws.aabb, &ws.keys[0], &ws.keys[1], &ws.vals[0], &ws.vals[1], &ws.block_hist, &ws.scan_aux,
                      &ws.prio, &ws.child, &ws.vis, &ws.flags, &ws.starters, &ws.lefts, &ws.bounds})
        rthip_release(*b);
    ctx->refit_fence = (rc == RTHIP_OK && differ == 0u) ? 2 : 0;
    if (ctx->refit_fence == 0)
        fprintf(stderr, "rthip: the fast hand-off of the LBVH builder did not pass its self-check on device %d (%s): release / acquire hand-offs from now on\n",
                ctx->device, rc == RTHIP_OK ? "trees differ" : ctx->err.c_str());
    else if (getenv("RTHIP_TUNE_LOG"))
        fprintf(stderr, "rthip: hand-off self-check passed on device %d\n", ctx->device);
    ctx->err.clear();
    return RTHIP_OK;
}

int lbvh_build_device(rthip_ctx* ctx, const float* d_aabb6, uint64_t n, float4* d_nodes, hipStream_t s,
                      uint64_t* d_codes_unsorted, uint64_t* d_keys_sorted, uint32_t* d_ids_sorted, bool paired,
                      LbvhWorkspace* ws) {
    if (n == 0) return rthip_fail(ctx, RTHIP_E_INVALID, "lbvh build needs at least one box");
    if (n > (1ull << 30)) return rthip_fail(ctx, RTHIP_E_LIMIT, "lbvh build is limited to 2^30 boxes (got %llu)",
                                            (unsigned long long)n);
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (s) (void)hipStreamIsCapturing(s, &capturing);
    // The device-wide climb takes its fast hand-off only after this context has checked it on its device, before the
    // first build that needs it; a build that cannot wait for the check (its stream is being captured) goes without.
    if (ctx->refit_fence == 2 && !ctx->refit_checked && (n > kOneGroupUpper || ctx->upper_device_wide) && n > (uint64_t)kChunk &&
        capturing == hipStreamCaptureStatusNone)
        RT_TRY(handoff_self_check(ctx));
    LbvhWorkspace& w = ws ? *ws : ctx->ws;
    if (ws) ctx->bounds_of = nullptr;  // bounds left by a triangle pass are in the context's own workspace
    RT_TRY(lbvh_reserve_workspace(ctx, n, &w));

    uint32_t* bounds = (uint32_t*)w.bounds.p;
    uint64_t* keys[2] = {(uint64_t*)w.keys[0].p, (uint64_t*)w.keys[1].p};
    uint32_t* vals[2] = {(uint32_t*)w.vals[0].p, (uint32_t*)w.vals[1].p};
    uint8_t* prio = (uint8_t*)w.prio.p;
    uint8_t* lefts = (uint8_t*)w.lefts.p;

    ClimbArgs A;
    A.n = (uint32_t)n;
    A.prio = prio;
    A.aabb6 = d_aabb6;
    A.child = (ChildRec*)w.child.p;
    A.flags = (uint32_t*)w.flags.p;
    A.lefts = lefts;
    A.lefts_w = lefts;
    A.chunk_f = nullptr;
    A.starters = (uint32_t*)w.starters.p;
    A.start_count = (uint32_t*)w.flags.p + (n - 1);
    A.nodes = d_nodes;
    // up to kOneGroupUpper leaves the nodes above the chunks can come from leaf ranges instead of a chain of hand-offs
    // (range_upper_kernel); the child records' and the counters' buffers then hold the range form's tables
    const uint32_t n_chunks = div_up(n, kChunk);
    A.ranges = (n > (uint64_t)kChunk && n <= kOneGroupUpper && ctx->upper_ranges && !ctx->upper_device_wide) ? (uint32_t)ctx->upper_ranges : 0u;
    A.pre_box = (float*)w.child.p;
    A.suf_box = A.pre_box + 6 * n;
    A.tab = (uint16_t*)w.flags.p;
    A.cmin = (uint8_t*)w.flags.p + 512ull * n_chunks;

    if (n <= (uint64_t)kChunk) {  // one workgroup does everything
        ctx->bounds_of = nullptr;
        A.sorted_idx = vals[1];
        A.start_count = (uint32_t*)w.flags.p;  // never counted: no gap spans chunks
        if (paired) lbvh_small_kernel<true><<<1, kChunk, 0, s>>>((uint32_t)n, d_aabb6, keys[0], keys[1], vals[1], A, prio, lefts, ctx->box_bounds);
        else lbvh_small_kernel<false><<<1, kChunk, 0, s>>>((uint32_t)n, d_aabb6, keys[0], keys[1], vals[1], A, prio, lefts, ctx->box_bounds);
        RT_HIP(ctx, hipGetLastError());
        if (d_codes_unsorted) RT_HIP(ctx, hipMemcpyAsync(d_codes_unsorted, keys[0], n * 8, hipMemcpyDeviceToDevice, s));
        if (d_keys_sorted) RT_HIP(ctx, hipMemcpyAsync(d_keys_sorted, keys[1], n * 8, hipMemcpyDeviceToDevice, s));
        if (d_ids_sorted) RT_HIP(ctx, hipMemcpyAsync(d_ids_sorted, vals[1], n * 4, hipMemcpyDeviceToDevice, s));
        return RTHIP_OK;
    }
    if (!(ctx->bounds_of == d_aabb6 && ctx->bounds_n == n && ctx->bounds_stream == s)) {
        bounds_init_kernel<<<1, kBoundSlots * 8, 0, s>>>(bounds);
        uint32_t bblocks = div_up(n, kBlock);
        if (bblocks > 2048) bblocks = 2048;
        bounds_kernel<<<bblocks, kBlock, 0, s>>>(d_aabb6, n, bounds, ctx->box_bounds);
    }
    ctx->bounds_of = nullptr;
    // values: see identity_vals; the arrival counters of the n - 1 gaps + the starter count are cleared on the way
    // The range form has no arrival counters, only its list's length to clear: when nobody asks for the unsorted codes
    // and the sort is one in tiles, the tile sort makes the codes itself and clears that word (one launch fewer).
    // Larger builds: the first histogram pass of the sort makes the codes and clears the counters and the sort's flag.
    const bool tiles_sort = n <= ctx->sort_merge_max && !ctx->sort_by_passes;  // radix_sort_workspace_device's choice
    const bool sort_codes = !d_codes_unsorted && (!tiles_sort || (A.ranges != 0u && sort_makes_codes(ctx, n)));
    MortonSource codes_of;
    codes_of.aabb6 = d_aabb6;
    codes_of.bounds = bounds;
    codes_of.zero_word = A.start_count;
    // what the previous build by the upper bytes reported, if it has arrived: a build of that many boxes whose long
    // stretches had to be finished by sort_long_kernel goes by all eight passes from now on
    // (not looked at while a stream is being captured: an event query is not allowed then)
    if (ctx->sort_flag_pending && capturing == hipStreamCaptureStatusNone && hipEventQuery(ctx->sort_flag_ev) == hipSuccess) {
        ctx->sort_flag_pending = false;
        if (*ctx->sort_flag_host != 0u) {
            ctx->sort_top_gave_up_n = ctx->sort_flag_n;
            if (getenv("RTHIP_TUNE_LOG")) fprintf(stderr, "rthip sort: a long stretch of keys with equal upper bytes was out of order at %llu boxes: all eight passes from now on\n", (unsigned long long)ctx->sort_flag_n);
        }
    }
    const bool top_first = n > ctx->sort_merge_max && n > kTopSortMin && ctx->sort_top_bytes < 8 && ctx->sort_top_gave_up_n != n;
    uint32_t* gave_up = (uint32_t*)w.bounds.p + kBoundSlots * kBoundStride;  // a word behind the bounds' copies
    if (!tiles_sort) {
        codes_of.zero_all = (uint32_t*)w.flags.p;  // n words, the list's length among them
        codes_of.zero_word = nullptr;
        codes_of.zero_word2 = top_first ? gave_up : nullptr;
    }
    if (!sort_codes) {
        morton_kernel<<<div_up(n, kBlock) < 4096u ? div_up(n, kBlock) : 4096u, kBlock, 0, s>>>(d_aabb6, n, bounds, keys[0], nullptr, (uint32_t*)w.flags.p,
                                                                                               top_first ? gave_up : nullptr);
        RT_HIP(ctx, hipGetLastError());
    }
    if (d_codes_unsorted)
        RT_HIP(ctx, hipMemcpyAsync(d_codes_unsorted, keys[0], n * 8, hipMemcpyDeviceToDevice, s));

    // Above kTopSortMin keys: passes over the upper four bytes, the rest finished inside the stretches of keys that share
    // them -- the short ones (nearly all) by sort_fix_kernel, long ones that are out of order by sort_long_kernel, which
    // costs a launch and nothing else when there are none.  Nothing is read back in between: the build never waits for
    // its stream, and a captured build takes the same kernels.  The flag travels to the host behind the build (not
    // while capturing) and is looked at by the NEXT build.
    int cur = 0;
    if (top_first) {  // (the flag was cleared by the kernel that made the codes)
        RT_TRY(radix_sort_workspace_device(ctx, n, 8, s, &cur, &w, true, 8 - ctx->sort_top_bytes, sort_codes ? &codes_of : nullptr));
        sort_fix_kernel<<<div_up(n, kFixTile), 256, 0, s>>>(keys[cur], vals[cur], (uint32_t)n, gave_up);
        RT_HIP(ctx, hipGetLastError());
        sort_long_kernel<<<256, 256, 0, s>>>(keys[cur], vals[cur], keys[cur ^ 1], vals[cur ^ 1], (uint32_t)n, gave_up);
        RT_HIP(ctx, hipGetLastError());
        if (capturing == hipStreamCaptureStatusNone && ctx->sort_flag_host && !ctx->sort_flag_pending) {
            RT_HIP(ctx, hipMemcpyAsync(ctx->sort_flag_host, gave_up, 4, hipMemcpyDeviceToHost, s));
            RT_HIP(ctx, hipEventRecord(ctx->sort_flag_ev, s));
            ctx->sort_flag_pending = true;
            ctx->sort_flag_n = n;
        }
    } else {
        RT_TRY(radix_sort_workspace_device(ctx, n, 8, s, &cur, &w, true, 0, sort_codes ? &codes_of : nullptr));
    }
    if (d_keys_sorted) RT_HIP(ctx, hipMemcpyAsync(d_keys_sorted, keys[cur], n * 8, hipMemcpyDeviceToDevice, s));
    if (d_ids_sorted) RT_HIP(ctx, hipMemcpyAsync(d_ids_sorted, vals[cur], n * 4, hipMemcpyDeviceToDevice, s));

    A.sorted_idx = vals[cur];
    const uint32_t chunks = div_up(n, kChunk);
    uint32_t ublocks = div_up(n - 1, kBlock);
    if (ublocks > 2048) ublocks = 2048;
    // shape: gap priorities and every leaf's S, by a suffix scan of the running-minimum records
    const uint32_t vtiles = div_up(n, kVisTile);
    ulonglong2* tile_in = (ulonglong2*)w.vis.p;
    VisF* tile_f = (VisF*)(tile_in + vtiles);
    if (A.ranges) {
        // the range form's chunk kernel finishes the scan for its own leaves: one stretch per chunk is all it needs
        A.chunk_f = (const VisF*)w.vis.p;
        vis_tile_kernel<kChunk / kBlock><<<chunks, kBlock, 0, s>>>(keys[cur], (uint32_t)n, prio, (VisF*)w.vis.p);
    } else {
        vis_tile_kernel<kVisItems><<<vtiles, kBlock, 0, s>>>(keys[cur], (uint32_t)n, prio, tile_f);
        RT_HIP(ctx, hipGetLastError());
        if (vtiles > kVisFusedTiles) vis_tiles_kernel<<<1, kVisTilesBlock, 0, s>>>(tile_f, vtiles, tile_in);
        vis_leaf_kernel<<<vtiles, kBlock, 0, s>>>(prio, (uint32_t)n, tile_f, vtiles, vtiles > kVisFusedTiles ? tile_in : nullptr, lefts);
    }
    RT_HIP(ctx, hipGetLastError());
    // boxes: every node's record goes to its pre-order place
    if (A.ranges) {
        if (paired) climb_local_kernel<true, true><<<chunks, kChunk, 0, s>>>(A);
        else climb_local_kernel<false, true><<<chunks, kChunk, 0, s>>>(A);
    } else if (paired) climb_local_kernel<true><<<chunks, kChunk, 0, s>>>(A);
    else climb_local_kernel<false><<<chunks, kChunk, 0, s>>>(A);
    RT_HIP(ctx, hipGetLastError());
    if (A.ranges) {
        const uint32_t groups = n_chunks >= 64u ? 8u : 2u;  // every workgroup builds the tables for itself (a microsecond)
        if (paired) range_upper_kernel<true><<<groups, kOneGroupThreads, 0, s>>>(A);
        else range_upper_kernel<false><<<groups, kOneGroupThreads, 0, s>>>(A);
        RT_HIP(ctx, hipGetLastError());
        return RTHIP_OK;
    }
    int mode = (n <= kOneGroupUpper && !ctx->upper_device_wide) ? 3 : (int)ctx->refit_fence;
    if (mode == 2 && !ctx->refit_checked) mode = 0;  // unchecked (this build could not wait for the check): the memory model's own form
#define RT_UPPER(MODE, GRID, BLOCK)                                               \
    if (paired) climb_upper_kernel<MODE, true><<<GRID, BLOCK, 0, s>>>(A);         \
    else climb_upper_kernel<MODE, false><<<GRID, BLOCK, 0, s>>>(A)
    if (mode == 3) { RT_UPPER(3, 1, kOneGroupThreads); }
    else if (mode == 1) { RT_UPPER(1, ublocks, kBlock); }
    else if (mode == 2) { RT_UPPER(2, ublocks, kBlock); }
    else { RT_UPPER(0, ublocks, kBlock); }
#undef RT_UPPER
    RT_HIP(ctx, hipGetLastError());
    return RTHIP_OK;
}

// ---- a workspace that is not allocated ------------------------------------------------------------
// Callers that cannot allocate (code under HIP graph capture, serving loops, harnesses that forbid hipMalloc) build in
// this file-scope device array instead.  Every GPU the library runs on holds its own copy at its own address until the
// process ends (about 13 MB); the array's name is only a host-side stand-in, so its device address is looked up once
// per device with hipGetSymbolAddress and only that pointer is ever handed to kernels and memsets.
namespace {

// What a build of kStaticBuildMaxBoxes boxes carves out of the array, in carve order (every piece rounded up to 256
// bytes): about 98 bytes per box -- key / value ping-pong 24, priorities 1, child slots 64, counters 4, list 4, one
// byte of shape -- plus the tile histograms of the finer tiling, the shape scan's records and the bounds.
constexpr size_t carve_round(size_t bytes) { return (bytes + 255) & ~(size_t)255; }
constexpr uint64_t kStaticN = kStaticBuildMaxBoxes;
constexpr size_t kStaticHistBytes = (256ull * div_up(kStaticN, kSmallTile) + 256) * 4;
constexpr size_t kStaticCarves[] = {
    kStaticN * 8, kStaticN * 8, kStaticN * 4, kStaticN * 4,      // keys[0], keys[1], vals[0], vals[1]
    kStaticHistBytes,                                            // block_hist
    kStaticN + 8, 2 * kStaticN * sizeof(ChildRec),               // prio, child
    vis_bytes(kStaticN), kStaticN * 4,                           // vis, flags
    (kStaticN + 1) * 4, kStaticN + 32, kBoundBytes,              // starters, lefts, bounds
};
constexpr size_t carve_total() {
    size_t t = 0;
    for (size_t b : kStaticCarves) t += carve_round(b);
    return t;
}
constexpr size_t kStaticBytes = carve_total();
static_assert(kStaticBytes <= kStaticBuildMaxBoxes * 128u, "the static workspace has outgrown 128 bytes per box: check the carve list");
static __device__ __align__(256) unsigned char scratch[kStaticBytes];

constexpr int kMaxDevices = 64;
void* g_scratch_ptr[kMaxDevices];  // device address of `scratch` on device d, once looked up; guarded by g_scratch_mutex
std::mutex g_scratch_mutex;        // contexts are created from any thread

bool carve(DevBuf& b, unsigned char* base, size_t& off, size_t bytes) {
    bytes = carve_round(bytes);
    if (off + bytes > kStaticBytes) return false;
    b.p = base + off;
    b.cap = bytes;
    b.fixed = true;
    off += bytes;
    return true;
}

}  // namespace

int lbvh_static_workspace_init(rthip_ctx* ctx) {
    if (ctx->ws_static_ready) return RTHIP_OK;
    if (ctx->device < 0 || ctx->device >= kMaxDevices)
        return rthip_fail(ctx, RTHIP_E_LIMIT, "device %d is beyond the %d devices the static workspace is kept for", ctx->device, kMaxDevices);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    unsigned char* base = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[ctx->device]) {
            void* ptr = nullptr;
            RT_HIP(ctx, hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
            if (!ptr) return rthip_fail(ctx, RTHIP_E_HIP, "hipGetSymbolAddress returned a null address for the static workspace");
            g_scratch_ptr[ctx->device] = ptr;
        }
        base = (unsigned char*)g_scratch_ptr[ctx->device];
    }
    // the same list as kStaticCarves, which sized the array (a piece that no longer fits fails the build of the library's
    // own sizes at compile time, not a caller's at run time)
    LbvhWorkspace& w = ctx->ws_static;
    DevBuf* const pieces[] = {&w.keys[0], &w.keys[1], &w.vals[0], &w.vals[1], &w.block_hist, &w.prio, &w.child,
                              &w.vis, &w.flags, &w.starters, &w.lefts, &w.bounds};
    static_assert(sizeof(pieces) / sizeof(pieces[0]) == sizeof(kStaticCarves) / sizeof(kStaticCarves[0]), "one size per piece");
    size_t off = 0;
    bool ok = true;
    for (size_t k = 0; k < sizeof(pieces) / sizeof(pieces[0]); k++) ok = ok && carve(*pieces[k], base, off, kStaticCarves[k]);
    if (!ok) return rthip_fail(ctx, RTHIP_E_LIMIT, "the static workspace of %zu bytes is too small for %llu boxes", kStaticBytes, (unsigned long long)kStaticN);
    ctx->ws_static_ready = true;
    return RTHIP_OK;
}
