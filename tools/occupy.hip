// tools/liboccupy.so — a FOREIGN persistent kernel for tests (not part of libcnsn_hip.so): `workgroups` workgroups that each
// take `lds_bytes` of LDS (160 KiB = a whole CU's: nothing else that needs LDS fits beside it) and spin on the chip-wide
// 100 MHz clock for `milliseconds`.  Stands in for what RCCL's channel kernels do to a training process: they sit on
// compute units for the whole length of a collective while the op's persistent cluster grids are launched next to them
// (tests/test_gpu_foreign_kernel.py).
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void occupy_kernel(long long ticks, unsigned* sink) {
    extern __shared__ char smem[];
    smem[threadIdx.x] = 1;
    const long long t0 = (long long)wall_clock64();
    while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
    if (smem[(threadIdx.x + 1) & 255] == 2 && sink) *sink = 1;  // (keeps the LDS allocation alive)
}

extern "C" int occupy_launch(int workgroups, int lds_bytes, int milliseconds, void* stream) {
    if (workgroups <= 0 || lds_bytes < 256 || milliseconds < 0) return -1;
    hipError_t e = hipFuncSetAttribute((const void*)occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return (int)e;
    occupy_kernel<<<workgroups, 256, lds_bytes, (hipStream_t)stream>>>((long long)milliseconds * 100000ll, nullptr);
    return (int)hipGetLastError();
}
