// tools/store_policy_probe.hip — measurement aid (profiles/r05_arena.md section 5): does the cache policy of the stores change what
// the two kinds of memory do?  784 MiB blocks of 14 x 56 MiB hipMemCreate chunks, linear fill with every combination of the
// gfx950 store modifiers sc0 / sc1 / nt, TB/s per block.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/store_policy_probe.hip -o tools/store_policy_probe; run: tools/store_policy_probe [blocks]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

constexpr size_t kMiB = size_t(1) << 20;
constexpr int kRun = 12 * 1024;
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int POLICY>
__device__ __forceinline__ void store16(u4* p, u4 v) {
    if constexpr (POLICY == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if constexpr (POLICY == 1) asm volatile("global_store_dwordx4 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if constexpr (POLICY == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if constexpr (POLICY == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    if constexpr (POLICY == 4) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    if constexpr (POLICY == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" ::"v"(p), "v"(v) : "memory");
    if constexpr (POLICY == 6) asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
    if constexpr (POLICY == 7) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
}

template <int POLICY>
__global__ __launch_bounds__(256) void fill(char* base, size_t runs) {
    const int lane = threadIdx.x & 63;
    const size_t waves = (size_t)gridDim.x * 4, w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const u4 zero = {0u, 0u, 0u, 0u};
    for (size_t r = w; r < runs; r += waves) {
        u4* p = (u4*)(base + r * kRun) + lane;
#pragma unroll
        for (int j = 0; j < kRun / 1024; ++j) store16<POLICY>(p + j * 64, zero);
    }
}

template <int POLICY>
float rate(char* va, size_t bytes) {
    const size_t runs = bytes / kRun;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    fill<POLICY><<<2048, 256>>>(va, runs);
    float best = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        fill<POLICY><<<2048, 256>>>(va, runs);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::max(best, (float)(runs * kRun / 1e12 / (ms * 1e-3)));
    }
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return best;
}

int main(int argc, char** argv) {
    const int nblocks = argc > 1 ? atoi(argv[1]) : 12;
    const size_t chunk = 56 * kMiB;
    const int per = 14;
    const size_t bytes = per * chunk;
    CK(hipSetDevice(0));
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<char*> vas(nblocks);
    for (int b = 0; b < nblocks; ++b) {
        void* va = nullptr;
        CK(hipMemAddressReserve(&va, bytes, 0, nullptr, 0));
        for (int i = 0; i < per; ++i) {
            hipMemGenericAllocationHandle_t h;
            CK(hipMemCreate(&h, chunk, &prop, 0));
            CK(hipMemMap((char*)va + (size_t)i * chunk, chunk, 0, h, 0));
        }
        CK(hipMemSetAccess(va, bytes, &acc, 1));
        vas[b] = (char*)va;
    }
    printf("linear fill TB/s per block; columns: default, sc0, sc1, sc0 sc1, nt, sc0 nt, sc1 nt, sc0 sc1 nt\n");
    for (int b = 0; b < nblocks; ++b) {
        printf("block %2d: %.2f %.2f %.2f %.2f %.2f %.2f %.2f %.2f\n", b, rate<0>(vas[b], bytes), rate<1>(vas[b], bytes), rate<2>(vas[b], bytes),
               rate<3>(vas[b], bytes), rate<4>(vas[b], bytes), rate<5>(vas[b], bytes), rate<6>(vas[b], bytes), rate<7>(vas[b], bytes));
        fflush(stdout);
    }
    return 0;
}
