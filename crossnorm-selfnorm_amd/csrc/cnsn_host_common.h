// Host-side helpers shared by the translation units of libcnsn_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>
#include <unordered_map>

#include "../../include/cnsn_hip.h"

namespace cnsn {

inline int elem_bytes(int dtype) { return dtype == CNSN_F32 ? 4 : 2; }

// widest power-of-two vector (<= 16 B) that divides `span` elements
inline int pick_vec(int dtype, int span) {
    int v = 16 / elem_bytes(dtype);
    while (v > 1 && (span % v) != 0) v >>= 1;
    return v;
}

template <typename T>
struct TypeTag {
    using type = T;
};
template <int V>
struct IntTag {
    static constexpr int value = V;
};

// A launch with more than 64 KiB of dynamic LDS needs hipFuncAttributeMaxDynamicSharedMemorySize raised first — per
// kernel AND per device.  Remembered here so that the runtime is asked once per (kernel, device), not per launch.
template <typename Kern>
inline bool allow_dynamic_lds(Kern kern, size_t lds) {
    if (lds <= 64 * 1024) return true;
    static std::mutex mu;
    static std::unordered_map<uintptr_t, size_t> allowed;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = allowed[(uintptr_t)(const void*)kern * 31u + (uintptr_t)dev];
    if (have >= lds) return true;
    if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    have = lds;
    return true;
}

}  // namespace cnsn
