// Argument block shared by the implicit-GEMM convolution kernels (conv_igemm.hip: register-staged tiles;
// conv_igemm_v5.hip: direct global->LDS operand ring).
#pragma once
#include <hip/hip_runtime.h>

namespace dynmm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct IgemmArgs {
    const float* x;        // gemm input  [N, Ci, H, W]  (first c_in_split channels)
    const float* x2;       // remaining input channels or nullptr
    const float* wp;       // packed weights [K][CoP]  (CoP = Co rounded up to 4)
    const float* scale;    // [Co] or nullptr
    const float* shift;    // [Co] or nullptr
    const float* residual; // like y or nullptr
    const float* mask;     // like y or nullptr : y *= (mask > 0)
    float* y;              // gemm output [N, Co(first c_out_split), Ho, Wo]
    float* y2;             // remaining output channels or nullptr
    int N, Ci, H, W;
    int Co, Ho, Wo;
    int KH, KW, SH, SW, PH, PW;
    int c_in_split, c_out_split;
    int act;
    int M, K, CoP;
    int CiR;               // weight rows per filter tap: Ci, or Ci rounded up to 16 (zero rows) when 8 <= Ci, Ci % 16 != 0
    int n_co_tiles, n_pix_tiles;
    int subpix;            // DGRAD with stride > 1 and Ho % SH == Wo % SW == 0: output pixels are enumerated
                           // parity class by parity class (see pix_decode), so a tile is (mostly) class-pure
};

// ---- weight gradient --------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* x;
    const float* x2;
    const float* dy;
    float* out;            // slabs [splits][Co*K] in the layout of w: [Co][Ci][KH][KW]
    float* out_bias;       // optional bias-gradient slabs [splits][Co] (sum over pixels of dy), or nullptr
    int N, Ci, H, W;
    int Co, Ho, Wo;
    int KH, KW, SH, SW, PH, PW;
    int c_split;
    int M, K;
    int n_co_tiles, n_k_tiles;
    int steps_per_split;   // 32-pixel steps handled by one workgroup
    unsigned magic_wo;     // floor(2^32 / Wo) + 1 when Ho*Wo*Wo < 2^32 (exact rem / Wo by mulhi), else 0
    int k_major_out;       // v4 slabs: out[co][k] with k = tap*Ci + ci (128-byte store runs); permuted by the slab reduction
};

// Up to 8 weight gradients of identical geometry in ONE launch (dynmm_conv2d_wgrad_group): the workgroups of problem p are
// [p*per, (p+1)*per).  A launch holds one residency round whatever the number of problems, so each workgroup walks a
// nprob-times longer pixel range of its problem: the fixed costs of a launch (cold prologue, slab burst, tail) and the slab
// traffic are paid once per group instead of once per convolution.
constexpr int kWgradGroupMax = 8;
struct WgradGroup {
    const float* x[kWgradGroupMax];
    const float* dy[kWgradGroupMax];
    float* out[kWgradGroupMax];
    float* out_bias[kWgradGroupMax];
    int nprob, per;
};

// ---- direct global -> LDS loads, counted by hand (conv_igemm_v5.hip, conv_wgrad_v6.hip) --------------------------------
// s_waitcnt vmcnt(N) lgkmcnt(0)   (gfx9 encoding: vmcnt = simm16[15:14]:[3:0], expcnt [6:4] = 7 (no wait), lgkmcnt [11:8])
template <int N>
__device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | (7 << 4));
}

// One wave instruction: lane l copies the 16 bytes at sbase + voff[l] to LDS byte address lds + 16*l.
__device__ __forceinline__ void dma16(const float* sbase, unsigned voff_bytes, unsigned lds_addr) {
    // (an SALU write of M0 needs one wait state before an LDS-DMA instruction reads it — ISA "manually inserted wait states";
    // the compiler pads its own code, not inline assembly)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :
                 : "s"(lds_addr), "v"(voff_bytes), "s"(sbase)
                 : "memory", "m0");
}

// Per-workgroup phase timestamps for kernel-structure experiments (scratch/trace/): compiled in only
// with -DDYNMM_TRACE, never in the shipped library.
#ifdef DYNMM_TRACE
// (trace builds are ONE translation unit: conv_igemm.hip includes conv_igemm_v5.hip)
__device__ unsigned long long* g_trace = nullptr;   // [gridDim.x][6]: wall clock at start, after prologue, after loop, after epilogue; shader clock at entry, exit
#define DYNMM_TRACE_MARK(slot)                                                                  \
    do {                                                                                        \
        if (g_trace && threadIdx.x == 0) g_trace[(size_t)blockIdx.x * 6 + (slot)] = wall_clock64(); \
    } while (0)
#else
#define DYNMM_TRACE_MARK(slot) do {} while (0)
#endif

// The operand-ring kernels (conv_igemm_v5.hip).  `eligible` is a pure function of the geometry / pointers, so callers
// may use it to predict which kernel a launch gets; `launch` returns false when the shape is not eligible.
bool igemm_v5_eligible(const IgemmArgs& a, bool dgrad);
bool launch_igemm_v5(IgemmArgs& a, bool dgrad, hipStream_t st);

}  // namespace dynmm
