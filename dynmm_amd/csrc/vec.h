// Width-templated global memory access: V=4 issues 16-byte (dwordx4) loads/stores — the coalescing
// sweet spot for the HBM-bound passes — and V=1 is the fallback for planes whose size or alignment
// does not permit it (e.g. the 27x37 gate map, 5x5 PPM bins).
#pragma once
#include <initializer_list>
#include "common.h"

namespace dynmm {

template <int V>
__device__ __forceinline__ void vload(const float* __restrict__ p, float (&v)[V]);
template <>
__device__ __forceinline__ void vload<4>(const float* __restrict__ p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <>
__device__ __forceinline__ void vload<1>(const float* __restrict__ p, float (&v)[1]) { v[0] = *p; }

template <int V>
__device__ __forceinline__ void vstore(float* __restrict__ p, const float (&v)[V]);
template <>
__device__ __forceinline__ void vstore<4>(float* __restrict__ p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <>
__device__ __forceinline__ void vstore<1>(float* __restrict__ p, const float (&v)[1]) { *p = v[0]; }

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// V=4 is legal when the plane length is a multiple of 4 and every (non-null) pointer is 16B aligned.
static inline bool can_vec4(int plane, std::initializer_list<const void*> ptrs) {
    if (plane % 4 != 0) return false;
    for (const void* p : ptrs)
        if (p && !aligned16(p)) return false;
    return true;
}

}  // namespace dynmm
