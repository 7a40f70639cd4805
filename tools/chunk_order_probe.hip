// tools/chunk_order_probe.hip — measurement aid (profiles/r05_arena.md section 5): is the write rate of a 784 MiB block a property of
// the physical chunks it is made of, or of the ORDER in which they are mapped?  Creates blocks of 14 x 56 MiB hipMemCreate chunks,
// times a plane-strided fill (the arena's probe pattern) and a linear fill into each, then maps the SAME chunks into fresh address
// ranges in other orders and times again.  Address ranges are never reused (ROCm 7.2 stale translations, csrc/cnsn_arena.hip).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/chunk_order_probe.hip -o tools/chunk_order_probe; run: tools/chunk_order_probe [blocks] [random orders] [chunk MiB]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
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

__global__ __launch_bounds__(256) void strided_fill(char* base, size_t runs) {
    const int lane = threadIdx.x & 63;
    const size_t waves = (size_t)gridDim.x * 4, w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t cols = runs / 256, full = cols * 256;
    const uint4 zero = {0u, 0u, 0u, 0u};
    for (size_t r = w; r < runs; r += waves) {
        const size_t loc = r < full ? (r % 256) * cols + r / 256 : r;
        uint4* p = (uint4*)(base + loc * kRun) + lane;
#pragma unroll
        for (int j = 0; j < kRun / 1024; ++j) p[j * 64] = zero;
    }
}
__global__ __launch_bounds__(256) void linear_fill(char* base, size_t runs) {
    const int lane = threadIdx.x & 63;
    const size_t waves = (size_t)gridDim.x * 4, w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const uint4 zero = {0u, 0u, 0u, 0u};
    for (size_t r = w; r < runs; r += waves) {
        uint4* p = (uint4*)(base + r * kRun) + lane;
#pragma unroll
        for (int j = 0; j < kRun / 1024; ++j) p[j * 64] = zero;
    }
}

__global__ __launch_bounds__(256) void linear_read(char* base, size_t runs) {  // reads everything, writes (almost) nothing
    const int lane = threadIdx.x & 63;
    const size_t waves = (size_t)gridDim.x * 4, w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned acc = 0;
    for (size_t r = w; r < runs; r += waves) {
        const uint4* p = (const uint4*)(base + r * kRun) + lane;
#pragma unroll
        for (int j = 0; j < kRun / 1024; ++j) {
            const uint4 v = p[j * 64];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) ((unsigned*)base)[0] = acc;
}
__global__ __launch_bounds__(256) void linear_copy(char* dst, const char* src, size_t runs) {
    const int lane = threadIdx.x & 63;
    const size_t waves = (size_t)gridDim.x * 4, w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (size_t r = w; r < runs; r += waves) {
        const uint4* p = (const uint4*)(src + r * kRun) + lane;
        uint4* q = (uint4*)(dst + r * kRun) + lane;
#pragma unroll
        for (int j = 0; j < kRun / 1024; ++j) q[j * 64] = p[j * 64];
    }
}
float copy_rate(char* dst, const char* src, size_t bytes) {  // TB/s of bytes moved (read + written)
    const size_t runs = bytes / kRun;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    linear_copy<<<2048, 256>>>(dst, src, runs);
    float best = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        linear_copy<<<2048, 256>>>(dst, src, runs);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::max(best, (float)(2.0 * runs * kRun / 1e9 / (ms * 1e-3)));
    }
    return best;
}

template <typename K>
float rate(K kernel, char* va, size_t bytes) {
    const size_t runs = bytes / kRun;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    kernel<<<2048, 256>>>(va, runs);
    float best = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        kernel<<<2048, 256>>>(va, runs);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::max(best, (float)(runs * kRun / 1e9 / (ms * 1e-3)));
    }
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    return best;
}

int main(int argc, char** argv) {
    const int nblocks = argc > 1 ? atoi(argv[1]) : 10, nperm = argc > 2 ? atoi(argv[2]) : 5;
    const size_t chunk = (argc > 3 ? atoi(argv[3]) : 56) * kMiB;
    const int per = (int)((784 * kMiB + chunk - 1) / chunk);
    const size_t bytes = per * chunk;
    CK(hipSetDevice(0));
    hipMemAllocationProp prop{};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    std::vector<std::vector<hipMemGenericAllocationHandle_t>> blocks(nblocks);
    for (auto& b : blocks) {  // all physical memory first, like the arena's candidates
        b.resize(per);
        for (auto& h : b) CK(hipMemCreate(&h, chunk, &prop, 0));
    }
    std::mt19937 rng(12345);
    printf("%d blocks of %d x %zu MiB; columns: order 0 = creation order, then %d random orders, then reversed; strided / linear fill TB/s\n",
           nblocks, per, chunk / kMiB, nperm);
    for (int bi = 0; bi < nblocks; ++bi) {
        printf("block %2d:", bi);
        std::vector<int> order(per);
        for (int i = 0; i < per; ++i) order[i] = i;
        for (int p = 0; p < nperm + 2; ++p) {
            if (p > 0 && p <= nperm) std::shuffle(order.begin(), order.end(), rng);
            if (p == nperm + 1)
                for (int i = 0; i < per; ++i) order[i] = per - 1 - i;
            void* va = nullptr;
            CK(hipMemAddressReserve(&va, bytes, 0, nullptr, 0));
            for (int i = 0; i < per; ++i) CK(hipMemMap((char*)va + (size_t)i * chunk, chunk, 0, blocks[bi][order[i]], 0));
            CK(hipMemSetAccess(va, bytes, &acc, 1));
            const float s = rate(strided_fill, (char*)va, bytes), l = rate(linear_fill, (char*)va, bytes);
            printf("  %.2f/%.2f", s / 1000, l / 1000);
            CK(hipDeviceSynchronize());
            for (int i = 0; i < per; ++i) CK(hipMemUnmap((char*)va + (size_t)i * chunk, chunk));  // (the range stays reserved)
        }
        printf("\n");
        fflush(stdout);
    }
    // blocks assembled ACROSS the candidates: chunk i of block (i + k) % nblocks
    printf("mixed blocks (chunk i from block (i + k) %% n):\n");
    for (int k = 0; k < std::min(nblocks, 6); ++k) {
        void* va = nullptr;
        CK(hipMemAddressReserve(&va, bytes, 0, nullptr, 0));
        for (int i = 0; i < per; ++i) CK(hipMemMap((char*)va + (size_t)i * chunk, chunk, 0, blocks[(i + k) % nblocks][i], 0));
        CK(hipMemSetAccess(va, bytes, &acc, 1));
        const float s = rate(strided_fill, (char*)va, bytes), l = rate(linear_fill, (char*)va, bytes);
        printf("  mixed %d: %.2f/%.2f\n", k, s / 1000, l / 1000);
        CK(hipDeviceSynchronize());
        for (int i = 0; i < per; ++i) CK(hipMemUnmap((char*)va + (size_t)i * chunk, chunk));
    }
    // reads: every block mapped once more in creation order, all at the same time
    std::vector<char*> vas(nblocks);
    std::vector<float> wr(nblocks), rd(nblocks);
    for (int bi = 0; bi < nblocks; ++bi) {
        void* va = nullptr;
        CK(hipMemAddressReserve(&va, bytes, 0, nullptr, 0));
        for (int i = 0; i < per; ++i) CK(hipMemMap((char*)va + (size_t)i * chunk, chunk, 0, blocks[bi][i], 0));
        CK(hipMemSetAccess(va, bytes, &acc, 1));
        vas[bi] = (char*)va;
    }
    printf("write / read TB/s per block:\n");
    for (int bi = 0; bi < nblocks; ++bi) {
        wr[bi] = rate(linear_fill, vas[bi], bytes) / 1000;
        rd[bi] = rate(linear_read, vas[bi], bytes) / 1000;
        printf("  block %2d: write %.2f read %.2f\n", bi, wr[bi], rd[bi]);
    }
    int fast = -1, fast2 = -1, slow = -1, slow2 = -1;
    std::vector<int> idx(nblocks);
    for (int i = 0; i < nblocks; ++i) idx[i] = i;
    std::sort(idx.begin(), idx.end(), [&](int x, int y) { return wr[x] > wr[y]; });
    fast = idx[0], fast2 = idx[1], slow = idx[nblocks - 1], slow2 = idx[nblocks - 2];
    printf("copies (TB/s of bytes moved): fast %d,%d slow %d,%d\n", fast, fast2, slow, slow2);
    printf("  fast -> fast %.2f\n", copy_rate(vas[fast2], vas[fast], bytes) / 1000);
    printf("  slow -> fast %.2f\n", copy_rate(vas[fast], vas[slow], bytes) / 1000);
    printf("  fast -> slow %.2f\n", copy_rate(vas[slow], vas[fast], bytes) / 1000);
    printf("  slow -> slow %.2f\n", copy_rate(vas[slow2], vas[slow], bytes) / 1000);
    return 0;
}
