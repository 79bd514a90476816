// Ceiling for BVH traversal on MI355X: how many dependent random 32-byte record fetches per second
// the chip sustains.  Every lane walks its own pseudo-random chain through a table of 32-byte
// records (two 16-byte loads per record, the next index depends on the loaded data), like one ray
// walking a BVH whose nodes are not in cache order.
//
//   hipcc --offload-arch=gfx950 -O3 -o gather_rate gather_rate.hip && ./gather_rate
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return 1;                                                             \
        }                                                                         \
    } while (0)

template <int LOADS>
__global__ __launch_bounds__(256) void chase(const uint4* __restrict__ table, uint32_t mask, uint32_t steps,
                                             uint32_t active_lanes, uint32_t* __restrict__ out) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t me = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 0x9E3779B9u;  // keeps the chains apart
    uint32_t idx = me;
    uint32_t acc = 0;
    if (lane < active_lanes) {
        for (uint32_t s = 0; s < steps; s++) {
            idx &= mask;
            const uint4 a = table[2 * (size_t)idx];
            uint32_t v = a.x;
            if (LOADS == 2) {
                const uint4 b = table[2 * (size_t)idx + 1];
                v ^= b.w;
            }
            acc += v;
            idx = ((v ^ me) + s) * 2654435761u;  // next record depends on the data
            idx ^= idx >> 15;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}


// The same chase through a copy of the table in LDS (16 KB = 512 records per workgroup).
__global__ __launch_bounds__(256) void chase_lds(const uint4* __restrict__ table, uint32_t steps, uint32_t active_lanes,
                                                 uint32_t* __restrict__ out) {
    __shared__ uint4 local[1024];
    for (uint32_t k = threadIdx.x; k < 1024; k += 256) local[k] = table[k];
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t me = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 0x9E3779B9u;
    uint32_t idx = me;
    uint32_t acc = 0;
    if (lane < active_lanes) {
        for (uint32_t s = 0; s < steps; s++) {
            idx &= 511u;
            const uint4 a = local[2 * idx];
            const uint4 b = local[2 * idx + 1];
            const uint32_t v = a.x ^ b.w;
            acc += v;
            idx = ((v ^ me) + s) * 2654435761u;
            idx ^= idx >> 15;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t* d_out;
    CHECK(hipMalloc(&d_out, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("records_MB loads waves/SIMD lanes  G_records/s  GB/s(32B)\n");
    for (uint64_t records : {1ull << 8, 1ull << 9, 1ull << 11, 1ull << 13, 1ull << 15, 1ull << 17, 1ull << 20, 1ull << 22, 1ull << 24}) {  // 8 KB ... 512 MB
        std::vector<uint32_t> h(records * 8);
        uint32_t x = 12345;
        for (auto& w : h) {
            x = x * 1664525u + 1013904223u;
            w = x;
        }
        uint4* d_table;
        CHECK(hipMalloc(&d_table, records * 32));
        CHECK(hipMemcpy(d_table, h.data(), records * 32, hipMemcpyHostToDevice));
        for (int loads : {2, 1})
            for (int waves_per_simd : {2, 5, 8})
                for (uint32_t lanes : {64u, 32u, 16u}) {
                    const uint32_t blocks = (uint32_t)cus * waves_per_simd;  // 4 waves per block = waves_per_simd per SIMD
                    const uint32_t steps = 2000;
                    float best = 1e30f;
                    for (int rep = 0; rep < 3; rep++) {
                        CHECK(hipEventRecord(e0));
                        if (loads == 2)
                            chase<2><<<blocks, 256>>>(d_table, (uint32_t)records - 1, steps, lanes, d_out);
                        else
                            chase<1><<<blocks, 256>>>(d_table, (uint32_t)records - 1, steps, lanes, d_out);
                        CHECK(hipEventRecord(e1));
                        CHECK(hipEventSynchronize(e1));
                        float ms;
                        CHECK(hipEventElapsedTime(&ms, e0, e1));
                        if (ms < best) best = ms;
                    }
                    const double n = (double)blocks * 4 * lanes * steps;
                    printf("%9.3f %5d %10d %5u %12.1f %10.0f\n", records * 32 / 1e6, loads, waves_per_simd, lanes,
                           n / best / 1e6, n * 32 / best / 1e6);
                }
        if (records == (1ull << 9))
            for (int waves_per_simd : {2, 5})
                for (uint32_t lanes : {64u, 16u}) {
                    const uint32_t blocks = (uint32_t)cus * waves_per_simd;
                    const uint32_t steps = 2000;
                    float best = 1e30f;
                    for (int rep = 0; rep < 3; rep++) {
                        CHECK(hipEventRecord(e0));
                        chase_lds<<<blocks, 256>>>(d_table, steps, lanes, d_out);
                        CHECK(hipEventRecord(e1));
                        CHECK(hipEventSynchronize(e1));
                        float ms;
                        CHECK(hipEventElapsedTime(&ms, e0, e1));
                        if (ms < best) best = ms;
                    }
                    const double n = (double)blocks * 4 * lanes * steps;
                    printf("      LDS %5d %10d %5u %12.1f %10.0f\n", 2, waves_per_simd, lanes, n / best / 1e6, n * 32 / best / 1e6);
                }
        CHECK(hipFree(d_table));
    }
    return 0;
}
