// tools/coop_probe.hip — what would hipLaunchCooperativeKernel cost the cluster kernels?  (measurement aid, not part of the
// library; build: hipcc --offload-arch=gfx950 -O2 -std=c++17 tools/coop_probe.hip -o tools/coop_probe; result:
// profiles/r05_exchange_hardening.md.)
// The cluster-resident kernels need every workgroup of their persistent grid on the device at once.  The library sizes the grid
// from an occupancy query and relies on in-order dispatch; a cooperative launch would turn "sized to fit" into a
// guarantee checked by the runtime.  This probe times back-to-back launches of one persistent-grid-shaped kernel (512 x 256
// threads, 48 KB of dynamic LDS, a few microseconds of work) both ways, on one stream, and a pair of DEPENDENT launches —
// the shape of a training step's forward + backward — with a normal kernel of another stream in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)

__global__ __launch_bounds__(256) void grid_kernel(float* out, int spin) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    float v = lds[(threadIdx.x + 1) & 255];
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
    if (v == 12345.f) out[blockIdx.x] = v;
}

__global__ void filler(float* out, int spin) {
    float v = (float)threadIdx.x;
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
    if (v == 12345.f) out[blockIdx.x] = v;
}

int main() {
    float* out;
    CK(hipMalloc(&out, 1 << 20));
    hipStream_t s, other;
    CK(hipStreamCreate(&s));
    CK(hipStreamCreate(&other));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int grid = 512, lds = 48 * 1024, reps = 200;
    int spin = 2000;
    int coop = 0;
    CK(hipDeviceGetAttribute(&coop, hipDeviceAttributeCooperativeLaunch, 0));
    printf("cooperative launch supported: %d\n", coop);
    void* args[] = {(void*)&out, (void*)&spin};
    auto run = [&](bool cooperative, bool with_other) {
        float best = 1e9f;
        for (int round = 0; round < 5; ++round) {
            CK(hipDeviceSynchronize());
            if (with_other) filler<<<64, 256, 0, other>>>(out + 4096, 4000000);   // ~tens of ms on a few CUs
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < reps; ++i) {
                if (cooperative)
                    CK(hipLaunchCooperativeKernel((const void*)grid_kernel, dim3(grid), dim3(256), args, lds, s));
                else
                    grid_kernel<<<grid, 256, lds, s>>>(out, spin);
            }
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        CK(hipDeviceSynchronize());
        return best / reps * 1e3f;
    };
    printf("per launch, back to back on one stream (us):  normal %.2f   cooperative %.2f\n", run(false, false), coop ? run(true, false) : -1.f);
    printf("... with a long kernel of ANOTHER stream in flight:  normal %.2f   cooperative %.2f\n", run(false, true), coop ? run(true, true) : -1.f);
    return 0;
}
