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
    float* ws;             // operand-ring kernels, K split over `ksplit` workgroups per tile: partial accumulator tiles
    unsigned* flags;       //   [tiles][256 threads][accumulators] and one arrival counter per tile (zeroed by the launcher)
    int ksplit;            // 1: no split
    int subpix;            // DGRAD with stride > 1 and Ho % SH == Wo % SW == 0: output pixels are enumerated
                           // parity class by parity class (see pix_decode), so a tile is (mostly) class-pure
};

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
bool launch_igemm_v5(IgemmArgs& a, bool dgrad, hipStream_t st, void* workspace = nullptr, size_t workspace_bytes = 0);
// K-split the launcher would use for this geometry given a workspace (1 = none) and the bytes it needs
int igemm_v5_ksplit(const IgemmArgs& a, bool dgrad);
size_t igemm_v5_workspace_bytes(const IgemmArgs& a, bool dgrad);

}  // namespace dynmm
