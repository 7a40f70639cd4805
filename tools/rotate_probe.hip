// tools/rotate_probe.hip — where in the device memory is a plane-strided write fast, and why not everywhere?  (measurement aid,
// not part of the library; build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/rotate_probe.hip -o tools/rotate_probe;
// results: profiles/r04_placement_sensitivity.md, profiles/r04_memory_map.md.)
// The kernel is tools/pattern_bench.hip's plane copy ("column" order: wave w of workgroup-item (c, k) copies plane n = 4k + w of
// channel c, 13 x 1 KB loads and stores, software-pipelined through LDS like the cluster kernels; or linear order).  Modes:
//   (none)            six 784 MiB buffers, eight (x, y) pairs both ways round; slot order of every plane rotated by r
//                     (register slot j holds memory slot (j + r) mod 13; r = 0, n % 13, 5n % 13, (n/4) % 13) and linear order
//   vmm [MiB]         the same on hipMemCreate + hipMemMap buffers (the alignment request is not honoured)
//   carve [GiB] [MiB] ONE allocation, write targets carved out of it every so many MiB
//   aux               the eight cache policies of the stores on six targets
//   split             a plane per wave vs a plane per workgroup (split over its waves)
//   map [n] [src]     n buffers (default 300 = 235 GB) as write targets of buffer `src`: the map of profiles/r04_memory_map.md
//   mix [pool] [MiB]  write targets MAPPED from a pool of smaller physical allocations (adjacent chunks / chunks spread out)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes > 0 ? bytes : 0, 0x00020000);
}
template <int NV, int SAUX = 2>
__global__ __launch_bounds__(256, 3) void plane_copy(const float* __restrict__ x, float* __restrict__ y, int N, int C, int M, int K, int items,
                                                     int mode) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int voff = lane * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v4i* mypark = (v4i*)smem + (size_t)wave * NV * 64 + lane;
    v4i d[NV];
    auto plane_of = [&](int item, int& rot) -> long {
        if (mode == 4) { rot = 0; return (long)item * 4 + wave; }
        const int c = item / K, k = item - c * K;
        const int n = k * 4 + wave;
        rot = mode == 1 ? n % NV : mode == 2 ? (5 * n) % NV : mode == 3 ? (n / 4) % NV : 0;
        return n < N ? (long)n * C + c : -1;
    };
    auto slot = [&](int j, int rot) { const int t = j + rot; return t >= NV ? t - NV : t; };
    auto load = [&](long pl, int rot) {
        const float* pb = x + (pl < 0 ? 0 : pl) * M;
        const int bytes = pl < 0 ? 0 : M * 4;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int jm = slot(j, rot);
            d[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc(pb + jm * 256, bytes - jm * 1024), voff, 0, 2);
        }
    };
    int item = blockIdx.x;
    if (item >= items) return;
    int rot, nrot = 0;
    long pl = plane_of(item, rot);
    load(pl, rot);
    for (;;) {
#pragma unroll
        for (int j = 0; j < NV; ++j) mypark[j * 64] = d[j];
        const int next = item + gridDim.x;
        const bool more = next < items;
        const long npl = more ? plane_of(next, nrot) : -1;
        float* yb = y + (pl < 0 ? 0 : pl) * M;
        const int bytes = pl < 0 ? 0 : M * 4;
        const float* pb = x + (npl < 0 ? 0 : npl) * M;
        const int nbytes = npl < 0 ? 0 : M * 4;
#pragma unroll
        for (int j = 0; j < NV; ++j) {  // slot by slot as the pipelined kernels: store item t's slot, load item t+1's
            const int jm = slot(j, rot), jn = slot(j, nrot);
            v4i v = mypark[j * 64];
            __builtin_amdgcn_raw_buffer_store_b128(v, rsrc(yb + jm * 256, bytes - jm * 1024), voff, 0, SAUX);
            if (more) d[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc(pb + jn * 256, nbytes - jn * 1024), voff, 0, 2);
        }
        if (!more) break;
        item = next;
        pl = npl;
        rot = nrot;
    }
}
// column order, no software pipeline: SPLIT = 0: wave w copies plane 4k + w (13 slots); SPLIT = 1: the four waves of a workgroup
// share EVERY plane (wave w takes slots w, w + 4, w + 8, w + 12 of each of the workgroup's four planes, one plane after the
// other): a quarter as many planes are being written at any moment
template <int SPLIT>
__global__ __launch_bounds__(256, 3) void plane_copy_np(const float* __restrict__ x, float* __restrict__ y, int N, int C, int M, int K, int items) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int voff = lane * 16;
    v4i d[16];
    for (int item = blockIdx.x; item < items; item += gridDim.x) {
        const int c = item / K, k = item - c * K;
        if (!SPLIT) {
            const long pl = (long)(k * 4 + wave) * C + c;
            const float* pb = x + pl * M;
            float* yb = y + pl * M;
#pragma unroll
            for (int j = 0; j < 13; ++j) d[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc(pb + j * 256, M * 4 - j * 1024), voff, 0, 2);
#pragma unroll
            for (int j = 0; j < 13; ++j) __builtin_amdgcn_raw_buffer_store_b128(d[j], rsrc(yb + j * 256, M * 4 - j * 1024), voff, 0, 2);
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float* pb = x + ((long)(k * 4 + p) * C + c) * M;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int j = wave + 4 * i;
                    d[p * 4 + i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc(pb + j * 256, j < 13 ? M * 4 - j * 1024 : 0), voff, 0, 2);
                }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float* yb = y + ((long)(k * 4 + p) * C + c) * M;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int j = wave + 4 * i;
                    __builtin_amdgcn_raw_buffer_store_b128(d[p * 4 + i], rsrc(yb + j * 256, j < 13 ? M * 4 - j * 1024 : 0), voff, 0, 2);
                }
            }
        }
    }
}
template <typename F>
float time_ms(F&& f, int reps = 20) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}
int main(int argc, char** argv) {
    const int N = 256, C = 256, M = 56 * 56, K = N / 4, items = C * K;
    const size_t E = (size_t)N * C * M;
    const int NB = 6;
    float* buf[NB];
    // argv[1] = "vmm": every buffer is a hipMemCreate'd physical allocation mapped at a virtual address aligned to argv[2] MiB
    // (default 1024) — if what makes a placement good is the page-table FRAGMENT size (the largest power of two to which the
    // virtual and the physical address are both aligned and contiguous), these are good every time
    if (argc > 1 && !strcmp(argv[1], "carve")) {
        // ONE allocation of argv[2] GiB (default 24), write targets carved out of it every argv[3] MiB (default 256): is "a good
        // write target" a property of WHERE in the allocation (i.e. of physical regions) the plane copy's output lies?
        const size_t total = (size_t)(argc > 2 ? atoi(argv[2]) : 24) << 30, step = (size_t)(argc > 3 ? atoi(argv[3]) : 256) << 20;
        char* big;
        float* src;
        CK(hipMalloc(&src, E * 4)); CK(hipMemset(src, 0, E * 4));
        CK(hipMalloc(&big, total)); CK(hipMemset(big, 0, total));
        const double gb2 = 2.0 * E * 4 / 1e9;
        auto k2 = plane_copy<13>;
        printf("src %p big %p\n| offset MiB | column GB/s | linear GB/s | read-from-here column GB/s |\n|---|---|---|---|\n", (void*)src, (void*)big);
        for (size_t off = 0; off + E * 4 <= total; off += step) {
            float* dst = (float*)(big + off);
            float m0 = time_ms([&] { k2<<<768, 256, (size_t)4 * 64 * 13 * 16>>>(src, dst, N, C, M, K, items, 0); }, 10);
            float m4 = time_ms([&] { k2<<<768, 256, (size_t)4 * 64 * 13 * 16>>>(src, dst, N, C, M, K, items, 4); }, 10);
            float mr = time_ms([&] { k2<<<768, 256, (size_t)4 * 64 * 13 * 16>>>(dst, src, N, C, M, K, items, 0); }, 10);
            printf("| %zu | %.0f | %.0f | %.0f |\n", off >> 20, gb2 / m0 * 1e3, gb2 / m4 * 1e3, gb2 / mr * 1e3);
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "aux")) {
        // the cache policy of the STORES (aux operand of buffer_store: 0 default, 1 sc0, 2 nt, 3 sc0 nt, 16 sc1, 17 sc0 sc1, 18 sc1 nt,
        // 19 sc0 sc1 nt) on six buffers written in turn from one source: does a policy make every buffer a good write target?
        float* b6[7];
        for (int i = 0; i < 7; ++i) { CK(hipMalloc(&b6[i], E * 4)); CK(hipMemset(b6[i], 0, E * 4)); }
        const double gb2 = 2.0 * E * 4 / 1e9;
        const size_t l2 = (size_t)4 * 64 * 13 * 16;
        printf("| write target | aux 0 | 1 | 2 (the kernels) | 3 | 16 | 17 | 18 | 19 |\n|---|---|---|---|---|---|---|---|---|\n");
        for (int i = 1; i < 7; ++i) {
            printf("| %p |", (void*)b6[i]);
            auto go = [&](auto kern) {
                float ms = time_ms([&] { kern<<<768, 256, l2>>>(b6[0], b6[i], N, C, M, K, items, 0); }, 10);
                printf(" %.0f |", gb2 / ms * 1e3);
            };
            go(plane_copy<13, 0>); go(plane_copy<13, 1>); go(plane_copy<13, 2>); go(plane_copy<13, 3>);
            go(plane_copy<13, 16>); go(plane_copy<13, 17>); go(plane_copy<13, 18>); go(plane_copy<13, 19>);
            printf("\n");
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "map")) {
        // argv[2] buffers of 784 MiB (default 300 = 235 GB of the 288): every one as the WRITE target of the plane copy from buffer
        // 0, in the kernels' order and in linear order — where in the device memory are the good targets?
        const int nb = argc > 2 ? atoi(argv[2]) : 300;
        const int src_i = argc > 3 ? atoi(argv[3]) : 0;  // which buffer is read
        std::vector<float*> bs(nb, nullptr);
        int got = 0;
        for (int i = 0; i < nb; ++i) {
            if (hipMalloc(&bs[i], E * 4) != hipSuccess) { (void)hipGetLastError(); break; }
            CK(hipMemsetAsync(bs[i], 0, E * 4));
            ++got;
        }
        CK(hipDeviceSynchronize());
        const double gb2 = 2.0 * E * 4 / 1e9;
        const size_t l2 = (size_t)4 * 64 * 13 * 16;
        printf("%d buffers\n| # | address | column GB/s | linear GB/s | read-from-it column GB/s |\n|---|---|---|---|---|\n", got);
        for (int i = 0; i < got; ++i) {
            if (i == src_i) continue;
            float m0 = time_ms([&] { plane_copy<13><<<768, 256, l2>>>(bs[src_i], bs[i], N, C, M, K, items, 0); }, 6);
            float m4 = time_ms([&] { plane_copy<13><<<768, 256, l2>>>(bs[src_i], bs[i], N, C, M, K, items, 4); }, 6);
            float mr = time_ms([&] { plane_copy<13><<<768, 256, l2>>>(bs[i], bs[src_i], N, C, M, K, items, 0); }, 6);
            printf("| %d | %p | %.0f | %.0f | %.0f |\n", i, (void*)bs[i], gb2 / m0 * 1e3, gb2 / m4 * 1e3, gb2 / mr * 1e3);
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "mix")) {
        // Are the fast write targets the buffers whose pages come from TWO regions?  A pool of argv[2] (default 128) physical
        // allocations of argv[3] MiB (default 392 = half a tensor; must divide 784), and write targets MAPPED from them
        // (hipMemAddressReserve + hipMemMap): chunks that were allocated next to each other, or far apart.
        const int pool = argc > 2 ? atoi(argv[2]) : 128;
        const size_t chunk = (size_t)(argc > 3 ? atoi(argv[3]) : 392) << 20;
        const int per = (int)((E * 4) / chunk);
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = 0;
        std::vector<hipMemGenericAllocationHandle_t> hs(pool);
        for (int i = 0; i < pool; ++i) CK(hipMemCreate(&hs[i], chunk, &prop, 0));
        float* src;
        CK(hipMalloc(&src, E * 4)); CK(hipMemset(src, 0, E * 4));
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        const double gb2 = 2.0 * E * 4 / 1e9;
        const size_t l2 = (size_t)4 * 64 * 13 * 16;
        printf("pool %d chunks of %zu MiB, %d per tensor\n| chunks mapped | column GB/s | linear GB/s |\n|---|---|---|\n", pool, chunk >> 20, per);
        auto run = [&](const std::vector<int>& idx) {
            void* va = nullptr;
            CK(hipMemAddressReserve(&va, E * 4, 0, nullptr, 0));
            for (int q = 0; q < per; ++q) CK(hipMemMap((char*)va + q * chunk, chunk, 0, hs[idx[q]], 0));
            CK(hipMemSetAccess(va, E * 4, &acc, 1));
            float* dst = (float*)va;
            float m0 = time_ms([&] { plane_copy<13><<<768, 256, l2>>>(src, dst, N, C, M, K, items, 0); }, 6);
            float m4 = time_ms([&] { plane_copy<13><<<768, 256, l2>>>(src, dst, N, C, M, K, items, 4); }, 6);
            printf("|");
            for (int q = 0; q < per; ++q) printf(" %d", idx[q]);
            printf(" | %.0f | %.0f |\n", gb2 / m0 * 1e3, gb2 / m4 * 1e3);
            CK(hipDeviceSynchronize());
            CK(hipMemUnmap(va, E * 4));
            CK(hipMemAddressFree(va, E * 4));
        };
        for (int base = 0; base + per <= pool; base += pool / 8) {  // adjacent chunks
            std::vector<int> idx;
            for (int q = 0; q < per; ++q) idx.push_back(base + q);
            run(idx);
        }
        for (int base = 0; base < pool / per; base += std::max(1, pool / per / 8)) {  // chunks spread over the whole pool
            std::vector<int> idx;
            for (int q = 0; q < per; ++q) idx.push_back(base + q * (pool / per));
            run(idx);
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "split")) {
        float* b6[7];
        for (int i = 0; i < 7; ++i) { CK(hipMalloc(&b6[i], E * 4)); CK(hipMemset(b6[i], 0, E * 4)); }
        const double gb2 = 2.0 * E * 4 / 1e9;
        printf("| write target | a plane per wave | a plane per workgroup (split over its waves) | pipelined, plane per wave (the kernels) |\n|---|---|---|---|\n");
        for (int i = 1; i < 7; ++i) {
            float m0 = time_ms([&] { plane_copy_np<0><<<768, 256>>>(b6[0], b6[i], N, C, M, K, items); }, 10);
            float m1 = time_ms([&] { plane_copy_np<1><<<768, 256>>>(b6[0], b6[i], N, C, M, K, items); }, 10);
            float m2 = time_ms([&] { plane_copy<13><<<768, 256, (size_t)4 * 64 * 13 * 16>>>(b6[0], b6[i], N, C, M, K, items, 0); }, 10);
            printf("| %p | %.0f | %.0f | %.0f |\n", (void*)b6[i], gb2 / m0 * 1e3, gb2 / m1 * 1e3, gb2 / m2 * 1e3);
        }
        return 0;
    }
    const bool vmm = argc > 1 && !strcmp(argv[1], "vmm");
    const size_t align = (size_t)(argc > 2 ? atoi(argv[2]) : 1024) << 20;
    for (int i = 0; i < NB; ++i) {
        if (!vmm) {
            CK(hipMalloc(&buf[i], E * 4));
        } else {
            hipMemAllocationProp prop = {};
            prop.type = hipMemAllocationTypePinned;
            prop.location.type = hipMemLocationTypeDevice;
            prop.location.id = 0;
            size_t gran = 0;
            CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
            const size_t sz = ((E * 4 + gran - 1) / gran) * gran;
            hipMemGenericAllocationHandle_t h;
            CK(hipMemCreate(&h, sz, &prop, 0));
            void* va = nullptr;
            CK(hipMemAddressReserve(&va, sz, align, nullptr, 0));
            CK(hipMemMap(va, sz, 0, h, 0));
            hipMemAccessDesc acc = {};
            acc.location = prop.location;
            acc.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(va, sz, &acc, 1));
            buf[i] = (float*)va;
            if (i == 0) printf("vmm: granularity %zu, size %zu, alignment %zu MiB\n", gran, sz, align >> 20);
        }
        CK(hipMemset(buf[i], 0, E * 4));
    }
    const double gb = 2.0 * E * 4 / 1e9;
    const size_t lds = (size_t)4 * 64 * 13 * 16;
    auto kern = plane_copy<13>;
    printf("| x | y | column r=0 | r = n %% 13 | r = 5n %% 13 | r = (n/4) %% 13 | linear |\n|---|---|---|---|---|---|---|\n");
    const int pairs[][2] = {{0, 1}, {1, 0}, {2, 3}, {3, 2}, {4, 5}, {5, 4}, {0, 3}, {2, 5}};
    for (auto& p : pairs) {
        printf("| %p | %p |", (void*)buf[p[0]], (void*)buf[p[1]]);
        for (int mode = 0; mode < 5; ++mode) {
            float ms = time_ms([&] { kern<<<768, 256, lds>>>(buf[p[0]], buf[p[1]], N, C, M, K, items, mode); });
            printf(" %.4f ms %.0f GB/s |", ms, gb / ms * 1e3);
        }
        printf("\n");
    }
    return 0;
}
