// Shared device/host helpers for the gfx950 kernels of libdynmm_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "dynmm_hip.h"

// These kernels are written for one target: 160 KB of LDS per workgroup (mha_bwd_kernel<32,false> alone declares 66 KB),
// global_load_lds_dwordx4, the gfx950 MFMA set.  Fail at compile time, not at the first launch, on anything else.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libdynmm_hip is gfx950 (MI355X) code: build with --offload-arch=gfx950"
#endif

#define DYNMM_LAUNCH_CHECK()                                   \
    do {                                                       \
        hipError_t e__ = hipGetLastError();                    \
        if (e__ != hipSuccess) return -(1000 + (int)e__);      \
    } while (0)

#define DYNMM_HIP_TRY(expr)                                    \
    do {                                                       \
        hipError_t e__ = (expr);                               \
        if (e__ != hipSuccess) return -(1000 + (int)e__);      \
    } while (0)

namespace dynmm {

constexpr int kWave = 64;      // CDNA wavefront
constexpr int kNumXCD = 8;     // MI355X: 8 XCDs, each with a private L2

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t ceil_div_sz(size_t a, size_t b) { return (a + b - 1) / b; }

// Workgroups are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8).  Remap so that
// consecutive *logical* tile ids land on the same XCD and share its L2 (bijective for any nblk).
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int xcd = bid % kNumXCD;
    const int local = bid / kNumXCD;
    const int q = nblk / kNumXCD, r = nblk % kNumXCD;
    const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return start + local;
}

template <typename T>
__device__ __forceinline__ T wave_reduce_sum(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Sum over a 256-thread block; result valid in thread 0.  `smem` must hold >= 4 T's.
template <typename T>
__device__ __forceinline__ T block_reduce_sum_256(T v, T* smem) {
    v = wave_reduce_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) smem[wave] = v;
    __syncthreads();
    T r = T(0);
    if (threadIdx.x == 0) r = smem[0] + smem[1] + smem[2] + smem[3];
    return r;
}

__device__ __forceinline__ float act_fwd(float v, int act) {
    if (act == DYNMM_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == DYNMM_ACT_TANH) return tanhf(v);
    return v;
}

// derivative of the activation expressed through its OUTPUT y (ReLU: y>0, tanh: 1-y^2), so the
// backward never needs the pre-activation tensor (SURVEY.md §7 "shared in-place ReLU").
__device__ __forceinline__ float act_bwd(float g, float y, int act) {
    if (act == DYNMM_ACT_RELU) return y > 0.f ? g : 0.f;
    if (act == DYNMM_ACT_TANH) return g * (1.f - y * y);
    return g;
}

// out[i] = sum_s slabs[s][i] (+ an optional second region) in a fixed order: the deterministic tail of every
// split reduction in the library (conv_igemm.hip).
void launch_reduce_slabs(const float* slabs, float* out, int n, int nslabs, hipStream_t st,
                         const float* slabs2 = nullptr, float* out2 = nullptr, int n2 = 0);

}  // namespace dynmm
