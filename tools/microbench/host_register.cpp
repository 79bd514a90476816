// Which page-locked host memory the box offers for the frame's device-to-host copy: hipHostRegister of malloc'ed memory
// (aligned or not, default or portable|mapped flags) and hipHostMalloc, each with the copy time of one 1920x1080 float frame.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
static double ms_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}
int main() {
    const size_t bytes = 24883200;
    char* raw = (char*)malloc(bytes + 8192);
    memset(raw, 1, bytes + 8192);
    void* dev = nullptr;
    if (hipMalloc(&dev, bytes) != hipSuccess) return 1;
    for (int variant = 0; variant < 4; variant++) {
        char* p = (variant & 2) ? (char*)(((uintptr_t)raw + 4095) & ~(uintptr_t)4095) : raw + 16;
        const unsigned flags = (variant & 1) ? (hipHostRegisterPortable | hipHostRegisterMapped) : hipHostRegisterDefault;
        auto t0 = std::chrono::steady_clock::now();
        hipError_t e = hipHostRegister(p, bytes, flags);
        printf("hipHostRegister(%s, flags %u) -> %s (%.3f ms)\n", (variant & 2) ? "page-aligned" : "unaligned", flags, hipGetErrorString(e), ms_since(t0));
        if (e == hipSuccess) {
            for (int k = 0; k < 3; k++) {
                auto a = std::chrono::steady_clock::now();
                (void)hipMemcpy(p, dev, bytes, hipMemcpyDeviceToHost);
                printf("   registered D2H %.3f ms\n", ms_since(a));
            }
            (void)hipHostUnregister(p);
        } else (void)hipGetLastError();
    }
    void* pinned = nullptr;
    auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipHostMalloc(&pinned, bytes, hipHostMallocDefault);
    printf("hipHostMalloc -> %s (%.3f ms)\n", hipGetErrorString(e), ms_since(t0));
    if (e == hipSuccess)
        for (int k = 0; k < 4; k++) {
            auto a = std::chrono::steady_clock::now();
            (void)hipMemcpy(pinned, dev, bytes, hipMemcpyDeviceToHost);
            printf("   hipHostMalloc D2H %.3f ms\n", ms_since(a));
        }
    for (int k = 0; k < 3; k++) {
        auto a = std::chrono::steady_clock::now();
        (void)hipMemcpy(raw, dev, bytes, hipMemcpyDeviceToHost);
        printf("pageable D2H %.3f ms\n", ms_since(a));
    }
    return 0;
}
