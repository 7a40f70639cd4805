// Host-side helpers shared by the translation units of libcnsn_hip.so.
#pragma once
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

}  // namespace cnsn
