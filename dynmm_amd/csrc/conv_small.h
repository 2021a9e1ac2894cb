// Direct small-channel convolutions (csrc/conv_small.hip); dispatched from the entry points in conv_igemm.hip.
#pragma once
#include <hip/hip_runtime.h>

namespace dynmm {

struct SmallConvArgs {
    const float* x;
    const float* x2;        // channels [c_split, Ci) when c_split < Ci
    const float* wp;        // packed forward weight: wp[(tap * CiR + ci) * CoP + co]
    const float* scale;
    const float* shift;
    float* y;
    int N, Ci, H, W, Co, Ho, Wo, KH, KW, SH, SW, PH, PW, c_split, CiR, CoP, act;
};

bool small_conv_fwd_eligible(const SmallConvArgs& a, const float* residual);
int launch_small_conv_fwd(const SmallConvArgs& a, hipStream_t st);
bool stem_conv_fwd_eligible(const SmallConvArgs& a, const float* residual);
int launch_stem_conv_fwd(const SmallConvArgs& a, hipStream_t st);

}  // namespace dynmm
