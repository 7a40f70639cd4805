// does the range check of a raw buffer access on gfx950 look at voffset + inst_offset only (soffset excluded)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const float* src, float* dst, int plane_bytes, int soff) {
    const int lane = threadIdx.x;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, plane_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t w = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, plane_bytes, 0x00020000);
    const int so = __builtin_amdgcn_readfirstlane(soff);
    float v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane * 4, so, 0));
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v + 1000.f), w, lane * 4, so, 0);
}
int main() {
    const int n = 4096;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *s, *d;
    hipMalloc(&s, n * 4); hipMalloc(&d, n * 4);
    hipMemcpy(s, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(d, 0, n * 4);
    probe<<<1, 64>>>(s, d, 40 * 4, 1024 * 4);   // plane of 40 floats at element 1024
    hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int i = 0; i < n; ++i) {
        const float want = (i >= 1024 && i < 1064) ? (float)i + 1000.f : 0.f;
        if (h[i] != want) { ok = 0; printf("mismatch at %d: %g (want %g)\n", i, h[i], want); if (i > 1100) break; }
    }
    printf("soffset excluded from the range check: %s\n", ok ? "YES" : "NO");
    return 0;
}
