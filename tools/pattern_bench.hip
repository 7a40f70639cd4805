// tools/pattern_bench.hip — what does the single-touch ACCESS PATTERN itself cost on this box?  (measurement aid; not part
// of the library.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/pattern_bench.hip -o pattern_bench; run:
// ./pattern_bench [N C]; results: profiles/r02_access_pattern.md.)
// Persistent workgroups, each wave copies whole 56x56 fp32 planes (13 x 1 KB loads, then 13 x 1 KB stores) with NO
// exchange and NO arithmetic, in the item order of the cluster-resident kernels ("column": the 256 planes of a channel
// are 3.2 MB apart) or in linear order; optionally through LDS like the pipelined kernels; `plane_triad` is the backward's
// shape (two planes in, one out).  The column-order numbers are the ceiling the resident kernels are measured against.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* base, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes > 0 ? bytes : 0, 0x00020000);
}
__device__ __forceinline__ float wsum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// ORDER 0: column (resident kernels), 1: linear, 2: column but the cluster's channels advance by 1 (c = item / K as 0) with XCD-aware swizzle (same as 0 here)
template <int NV, int OCC, int AUX>
__global__ __launch_bounds__(256, OCC) void plane_copy(const float* __restrict__ x, float* __restrict__ y, int N, int C, int M, int K,
                                                       int items, int order, int dep, int pipe) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int voff = lane * 16;
    v4i d[NV];
    auto plane_of = [&](int item) -> long {
        if (order == 1) return (long)item * 4 + wave;
        if (order == 2) {  // 4 adjacent channels of ONE instance per workgroup; consecutive workgroups: consecutive instances
            const int c4 = item / N, n = item - c4 * N;
            return (long)n * C + c4 * 4 + wave;
        }
        if (order == 3) {  // 4 adjacent channels of one instance; consecutive workgroups: the next 4 channels (linear within an instance, 16 channels wide), then the next instance
            const int per = 4;  // channel quads side by side
            const int q = item % per, r = item / per;
            const int n = r % N, cq = (r / N) * per + q;
            return (long)n * C + cq * 4 + wave;
        }
        const int c = item / K, k = item - c * K;
        const int n = k * 4 + wave;
        return n < N ? (long)n * C + c : -1;
    };
    auto load = [&](long pl) {
        const float* pb = x + (pl < 0 ? 0 : pl) * M;
        const int bytes = pl < 0 ? 0 : M * 4;
#pragma unroll
        for (int j = 0; j < NV; ++j) d[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc(pb + j * 256, bytes - j * 1024), voff, 0, AUX);
    };
    if (!pipe) {
        for (int item = blockIdx.x; item < items; item += gridDim.x) {
            const long pl = plane_of(item);
            load(pl);
            int scale = 0;
            if (dep) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < NV; ++j) s += __int_as_float(d[j][0]) + __int_as_float(d[j][1]) + __int_as_float(d[j][2]) + __int_as_float(d[j][3]);
                s = wsum(s);
                scale = (s == 12345.678f) ? 1 : 0;
            }
            float* yb = y + (pl < 0 ? 0 : pl) * M;
            const int bytes = pl < 0 ? 0 : M * 4;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                v4i v = d[j];
                v[0] += scale;
                __builtin_amdgcn_raw_buffer_store_b128(v, rsrc(yb + j * 256, bytes - j * 1024), voff, 0, AUX);
            }
        }
    } else {
        // software pipeline through LDS: park item t, load t+1, then store t from LDS (the pipelined kernel's traffic shape)
        extern __shared__ __attribute__((aligned(16))) char smem[];
        v4i* mypark = (v4i*)smem + (size_t)wave * NV * 64 + lane;
        int item = blockIdx.x;
        if (item >= items) return;
        long pl = plane_of(item);
        load(pl);
        for (;;) {
            int scale = 0;
            if (dep) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < NV; ++j) s += __int_as_float(d[j][0]) + __int_as_float(d[j][1]) + __int_as_float(d[j][2]) + __int_as_float(d[j][3]);
                s = wsum(s);
                scale = (s == 12345.678f) ? 1 : 0;
            }
#pragma unroll
            for (int j = 0; j < NV; ++j) mypark[j * 64] = d[j];
            const int next = item + gridDim.x;
            const bool more = next < items;
            const long npl = more ? plane_of(next) : -1;
            if (more) load(npl);
            float* yb = y + (pl < 0 ? 0 : pl) * M;
            const int bytes = pl < 0 ? 0 : M * 4;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                v4i v = mypark[j * 64];
                v[0] += scale;
                __builtin_amdgcn_raw_buffer_store_b128(v, rsrc(yb + j * 256, bytes - j * 1024), voff, 0, AUX);
            }
            if (!more) break;
            item = next;
            pl = npl;
        }
    }
}

// backward shape: two planes in (g, x), one out; PIPE: next item's loads issued before the current item is stored
template <int NV, int OCC, int PIPE>
__global__ __launch_bounds__(256, OCC) void plane_triad(const float* __restrict__ g, const float* __restrict__ x, float* __restrict__ y,
                                                        int N, int C, int M, int K, int items, int order) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int voff = lane * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    v4i* mypark = (v4i*)smem + (size_t)wave * 2 * NV * 64 + lane;
    v4i dg[NV], dx[NV];
    auto plane_of = [&](int item) -> long {
        if (order == 1) return (long)item * 4 + wave;
        const int c = item / K, k = item - c * K;
        const int n = k * 4 + wave;
        return n < N ? (long)n * C + c : -1;
    };
    auto load = [&](long pl) {
        const float* gb = g + (pl < 0 ? 0 : pl) * M;
        const float* xb = x + (pl < 0 ? 0 : pl) * M;
        const int bytes = pl < 0 ? 0 : M * 4;
#pragma unroll
        for (int j = 0; j < NV; ++j) dg[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc(gb + j * 256, bytes - j * 1024), voff, 0, 2);
#pragma unroll
        for (int j = 0; j < NV; ++j) dx[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc(xb + j * 256, bytes - j * 1024), voff, 0, 2);
    };
    auto dep = [&]() {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) s += __int_as_float(dg[j][0]) * __int_as_float(dx[j][0]) + __int_as_float(dg[j][3]) + __int_as_float(dx[j][2]);
        s = wsum(s);
        return (s == 12345.678f) ? 1 : 0;
    };
    int item = blockIdx.x;
    if (item >= items) return;
    long pl = plane_of(item);
    load(pl);
    for (;;) {
        const int scale = dep();
        const int next = item + gridDim.x;
        const bool more = next < items;
        const long npl = more ? plane_of(next) : -1;
        float* yb = y + (pl < 0 ? 0 : pl) * M;
        const int bytes = pl < 0 ? 0 : M * 4;
        if (PIPE) {
#pragma unroll
            for (int j = 0; j < NV; ++j) { mypark[j * 64] = dg[j]; mypark[(NV + j) * 64] = dx[j]; }
            if (more) load(npl);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                v4i v = mypark[j * 64], w = mypark[(NV + j) * 64];
                v[0] += scale + w[1];
                __builtin_amdgcn_raw_buffer_store_b128(v, rsrc(yb + j * 256, bytes - j * 1024), voff, 0, 2);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                v4i v = dg[j];
                v[0] += scale + dx[j][1];
                __builtin_amdgcn_raw_buffer_store_b128(v, rsrc(yb + j * 256, bytes - j * 1024), voff, 0, 2);
            }
            if (more) load(npl);
        }
        if (!more) break;
        item = next;
        pl = npl;
    }
}

__global__ void stream_triad(const v4i* __restrict__ a, const v4i* __restrict__ b, v4i* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        v4i u = __builtin_nontemporal_load(a + i), w = __builtin_nontemporal_load(b + i);
        u[0] += w[1];
        __builtin_nontemporal_store(u, y + i);
    }
}

__global__ void stream_copy(const v4i* __restrict__ x, v4i* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(__builtin_nontemporal_load(x + i), y + i);
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
    const int N = argc > 1 ? atoi(argv[1]) : 256, C = argc > 2 ? atoi(argv[2]) : 256, H = 56, W = 56, M = H * W;
    const bool brief = argc > 3;  // ./pattern_bench N C brief: the two numbers bench.py rides along, as JSON
    if (!brief) printf("N %d C %d\n", N, C);
    const size_t E = (size_t)N * C * M;
    float *x, *y;
    CK(hipMalloc(&x, E * 4)); CK(hipMalloc(&y, E * 4));
    CK(hipMemset(x, 0, E * 4)); CK(hipMemset(y, 0, E * 4));
    const double gb = 2.0 * E * 4 / 1e9;
    for (int g : {2048, 4096, 8192, 16384}) {
        if (brief) break;
        float ms = time_ms([&] { stream_copy<<<g, 256>>>((const v4i*)x, (v4i*)y, E / 4); });
        printf("stream copy grid %5d: %.4f ms  %.0f GB/s\n", g, ms, gb / ms * 1e3);
    }
    const int K = N / 4, items = C * K;
    auto run = [&](auto kern, int occ, int order, int dep, int pipe, const char* tag) {
        const size_t lds = pipe ? (size_t)4 * 64 * 13 * 16 : 0;
        int o = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, kern, 256, lds));
        if (o > occ) o = occ;
        int grid = o * 256 / 64 * 64;
        float ms = time_ms([&] { kern<<<grid, 256, lds>>>(x, y, N, C, M, K, items, order, dep, pipe); });
        const char* on[] = {"column", "linear", "chan4", "chan4x4"};
        printf("%-10s occ %d(%d) order %-8s dep %d pipe %d: %.4f ms  %.0f GB/s\n", tag, occ, o, on[order], dep, pipe, ms, gb / ms * 1e3);
    };
    if (brief) {
        float* gq;
        CK(hipMalloc(&gq, E * 4)); CK(hipMemset(gq, 0, E * 4));
        auto kc = plane_copy<13, 4, 2>;
        auto kt = plane_triad<13, 2, 0>;
        const float mc = time_ms([&] { kc<<<4 * 256 / 64 * 64, 256>>>(x, y, N, C, M, K, items, 0, 1, 0); }, 10);
        const float mt = time_ms([&] { kt<<<2 * 256 / 64 * 64, 256>>>(gq, x, y, N, C, M, K, items, 0); }, 10);
        printf("{\"column_copy_ms\": %.4f, \"column_copy_GBps\": %.1f, \"column_triad_ms\": %.4f, \"column_triad_GBps\": %.1f}\n", mc,
               gb / mc * 1e3, mt, 3.0 * E * 4 / 1e9 / mt * 1e3);
        return 0;
    }
    for (int order = 0; order < 4; ++order) {
        run(plane_copy<13, 4, 2>, 4, order, 1, 0, "nt");
        run(plane_copy<13, 8, 2>, 8, order, 1, 0, "nt");
        run(plane_copy<13, 3, 2>, 3, order, 1, 1, "nt");
    }
    float* g;
    CK(hipMalloc(&g, E * 4)); CK(hipMemset(g, 0, E * 4));
    const double gb3 = 3.0 * E * 4 / 1e9;
    for (int gr : {4096, 16384}) {
        float ms = time_ms([&] { stream_triad<<<gr, 256>>>((const v4i*)g, (const v4i*)x, (v4i*)y, E / 4); });
        printf("stream triad grid %5d: %.4f ms  %.0f GB/s\n", gr, ms, gb3 / ms * 1e3);
    }
    auto runt = [&](auto kern, int occ, int order, int pipe) {
        const size_t lds = pipe ? (size_t)4 * 64 * 26 * 16 : 0;
        if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int o = 0;
        CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, kern, 256, lds));
        if (o > occ) o = occ;
        int grid = o * 256 / 64 * 64;
        float ms = time_ms([&] { kern<<<grid, 256, lds>>>(g, x, y, N, C, M, K, items, order); });
        printf("triad occ %d(%d) order %-6s pipe %d: %.4f ms  %.0f GB/s\n", occ, o, order ? "linear" : "column", pipe, ms, gb3 / ms * 1e3);
    };
    for (int order = 0; order < 2; ++order) {
        runt(plane_triad<13, 2, 0>, 2, order, 0);
        runt(plane_triad<13, 3, 0>, 3, order, 0);
        runt(plane_triad<13, 4, 0>, 4, order, 0);
        runt(plane_triad<13, 1, 1>, 1, order, 1);
    }
    return 0;
}
