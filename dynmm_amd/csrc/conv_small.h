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
    double* stats = nullptr;   // stem forward only: [2][Co] per-channel sums of y and y^2 ADDED here (BatchNorm batch statistics)
};

struct StemWgradArgs {
    const float* x;
    const float* dy;
    float* slabs;           // [workgroups][64][k-tiles * 32] partial weight gradients
    int N, H, W, Ho, Wo;
};

bool stem_conv_wgrad_eligible(int Ci, int Co, int KH, int KW, int SH, int SW, int PH, int PW, bool has_x2, bool has_bias);
size_t stem_conv_wgrad_workspace_bytes(int N, int Ci, int Ho, int Wo);
int launch_stem_conv_wgrad(const float* x, const float* dy, float* dw, float* workspace, int N, int Ci, int H, int W,
                           int Ho, int Wo, hipStream_t st);

// gate conv (5x5, stride 2, <= 8 output channels) weight + bias gradient on the vector ALUs
bool co8_wgrad_eligible(int Ci, int Co, int H, int W, int Ho, int Wo, int KH, int KW, int SH, int SW, int PH, int PW, int c_split);
size_t co8_wgrad_workspace_bytes(int N, int Ci, int Co);
int launch_co8_wgrad(const float* x, const float* x2, const float* dy, float* dw, float* dbias, float* workspace, int N, int Ci,
                     int H, int W, int Co, int Ho, int Wo, int c_split, hipStream_t st);

bool small_conv_fwd_eligible(const SmallConvArgs& a, const float* residual);
int launch_small_conv_fwd(const SmallConvArgs& a, hipStream_t st);
bool stem_conv_fwd_eligible(const SmallConvArgs& a, const float* residual);
int launch_stem_conv_fwd(const SmallConvArgs& a, hipStream_t st);

}  // namespace dynmm
