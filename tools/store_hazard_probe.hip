// Does a VALU write right behind a 16-byte buffer store clobber the store's data on gfx950?
// Hand-written sequences on explicit registers (nothing for the compiler to pad or to move):
//   form A: buffer_store_dwordx4 v[20:23], voff, srsrc, SGPR soffset ; v_mov_b32 v20, poison ; v_mov_b32 v21, poison
//   form B: the same with soffset = 0 (the offset added into voffset)
//   form C / D: form A with s_nop 0 / s_nop 1 between the store and the writes
// Counts the elements of the output that differ from the value computed before the store.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef int v4i __attribute__((ext_vector_type(4)));

template <int FORM>
__global__ void probe(const int* __restrict__ src, int* __restrict__ dst, int bytes, int rounds) {
    const int lane = threadIdx.x & 63;
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t w = __builtin_amdgcn_make_buffer_rsrc((void*)dst, 0, bytes, 0x00020000);
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    for (int it = 0; it < rounds; ++it) {
        const int so = __builtin_amdgcn_readfirstlane((wave * rounds + it) * 1024);
        v4i v = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16, so, 0);
        v.x += 1; v.y += 1; v.z += 1; v.w += 1;
        const int vo = FORM == 1 ? lane * 16 + so : lane * 16;
        // explicit registers: v[20:23] = the data, the store, then a VALU write of v20 with 0 / 1 / 2 wait states between
#define PROBE_HEAD "v_mov_b32 v20, %0\n\tv_mov_b32 v21, %1\n\tv_mov_b32 v22, %2\n\tv_mov_b32 v23, %3\n\ts_nop 4\n\t"
        if (FORM == 0)
            asm volatile(PROBE_HEAD "buffer_store_dwordx4 v[20:23], %4, %5, %6 offen\n\tv_mov_b32 v20, 0xdead\n\tv_mov_b32 v21, 0xdead"
                         :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "v"(vo), "s"(w), "s"(so) : "v20", "v21", "v22", "v23", "memory");
        else if (FORM == 1)
            asm volatile(PROBE_HEAD "buffer_store_dwordx4 v[20:23], %4, %5, 0 offen\n\tv_mov_b32 v20, 0xdead\n\tv_mov_b32 v21, 0xdead"
                         :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "v"(vo), "s"(w) : "v20", "v21", "v22", "v23", "memory");
        else if (FORM == 2)
            asm volatile(PROBE_HEAD "buffer_store_dwordx4 v[20:23], %4, %5, %6 offen\n\ts_nop 0\n\tv_mov_b32 v20, 0xdead\n\tv_mov_b32 v21, 0xdead"
                         :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "v"(vo), "s"(w), "s"(so) : "v20", "v21", "v22", "v23", "memory");
        else
            asm volatile(PROBE_HEAD "buffer_store_dwordx4 v[20:23], %4, %5, %6 offen\n\ts_nop 1\n\tv_mov_b32 v20, 0xdead\n\tv_mov_b32 v21, 0xdead"
                         :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w), "v"(vo), "s"(w), "s"(so) : "v20", "v21", "v22", "v23", "memory");
    }
}

int main() {
    const int waves = 256 * 8 * 4, rounds = 16;
    const size_t n = (size_t)waves * rounds * 256;  // ints
    std::vector<int> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (int)(i & 0xffff);
    int *s, *d;
    CHECK(hipMalloc(&s, n * 4)); CHECK(hipMalloc(&d, n * 4));
    CHECK(hipMemcpy(s, h.data(), n * 4, hipMemcpyHostToDevice));
    const char* names[4] = {"A: SGPR soffset, VALU write of the data right behind", "B: soffset 0 (offset in voffset), VALU write right behind",
                            "C: SGPR soffset, s_nop 0 (one wait state), VALU write", "D: SGPR soffset, s_nop 1 (two wait states), VALU write"};
    for (int form = 0; form < 4; ++form) {
        size_t bad = 0;
        for (int rep = 0; rep < 5; ++rep) {
            CHECK(hipMemset(d, 0, n * 4));
            if (form == 0) probe<0><<<waves / 4, 256>>>(s, d, (int)(n * 4), rounds);
            if (form == 1) probe<1><<<waves / 4, 256>>>(s, d, (int)(n * 4), rounds);
            if (form == 2) probe<2><<<waves / 4, 256>>>(s, d, (int)(n * 4), rounds);
            if (form == 3) probe<3><<<waves / 4, 256>>>(s, d, (int)(n * 4), rounds);
            CHECK(hipDeviceSynchronize());
            std::vector<int> o(n);
            CHECK(hipMemcpy(o.data(), d, n * 4, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < n; ++i) bad += o[i] != h[i] + 1;
        }
        printf("%-60s wrong elements in 5 launches of %zu: %zu\n", names[form], n, bad);
    }
    return 0;
}
