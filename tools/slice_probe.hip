// tools/slice_probe.hip — what does a channels-last tensor cost when it is read and written one COLUMN SLICE at a time?
// (measurement aid; not part of the library.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/slice_probe.hip -o
// tools/slice_probe; run: tools/slice_probe; result: profiles/r06_nhwc.md section 4.)
// A single-TOUCH channels-last SelfNorm would have to hold all N instances of a channel subset on chip (BatchNorm1d over N,
// models/cnsn.py:121,138): ~96 MB of registers + LDS admit 64-128 bytes of every pixel's row at 56x56 / 28x28, so the tensor
// would stream through the chip in ROWB / SL whole-grid passes, pass p touching bytes [p*SL, (p+1)*SL) of every row.  This
// probe copies a [rows][ROWB] tensor that way — no arithmetic, no exchange, no barriers between the passes (the real kernel
// would add two grid barriers per pass) — and prints the rate over all passes against a whole-row copy of the same tensor.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

// lanes_per_row = SL / 16; a wave takes 64 / lanes_per_row consecutive rows per step, U steps in flight
template <int U>
__global__ __launch_bounds__(256) void slice_copy(const char* __restrict__ x, char* __restrict__ y, long rows, int rowb, int sl, int pass) {
    const int lpr = sl / 16, rpw = 64 / lpr;                       // lanes per row, rows per wave and step
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long wid = (long)blockIdx.x * 4 + wave, nw = (long)gridDim.x * 4;
    const int col = pass * sl + (lane % lpr) * 16, rsub = lane / lpr;
    for (long r0 = wid * rpw * U; r0 < rows; r0 += nw * rpw * U) {
        v4i d[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long r = r0 + (long)u * rpw + rsub;
            if (r < rows) d[u] = __builtin_nontemporal_load((const v4i*)(x + r * rowb + col));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long r = r0 + (long)u * rpw + rsub;
            if (r < rows) __builtin_nontemporal_store(d[u], (v4i*)(y + r * rowb + col));
        }
    }
}

int main() {
    struct Case { const char* name; long rows; int rowb; } cases[] = {
        {"(256,256,56,56) bf16: 802 816 rows of 512 B", 256l * 56 * 56, 512},
        {"(256,512,28,28) bf16: 200 704 rows of 1 KB", 256l * 28 * 28, 1024},
        {"(256,1024,14,14) bf16: 50 176 rows of 2 KB", 256l * 14 * 14, 2048}};
    for (auto& c : cases) {
        const size_t bytes = (size_t)c.rows * c.rowb;
        char *x, *y;
        CK(hipMalloc(&x, bytes));
        CK(hipMalloc(&y, bytes));
        CK(hipMemset(x, 1, bytes));
        CK(hipMemset(y, 0, bytes));
        printf("%s = %.0f MB\n", c.name, bytes / 1e6);
        for (int sl : {c.rowb > 1024 ? 1024 : c.rowb, 256, 128, 64}) {  // (a wave takes at most 1 KB of a row per load)
            if (sl > c.rowb) continue;
            const int passes = c.rowb / sl;
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0));
            CK(hipEventCreate(&e1));
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipEventRecord(e0));
                for (int p = 0; p < passes; ++p) slice_copy<4><<<2048, 256>>>(x, y, c.rows, c.rowb, sl, p);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf("  slices of %4d B, %2d pass(es): %.3f ms = %.2f TB/s (read + write)\n", sl, passes, best, 2.0 * bytes / best / 1e9);
            fflush(stdout);
        }
        CK(hipFree(x));
        CK(hipFree(y));
    }
    return 0;
}
