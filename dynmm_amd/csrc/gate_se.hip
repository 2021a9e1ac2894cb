// The two "tiny" pieces of the hot path that sit between the big feature-map passes:
//   * SE excitation MLPs + gate blend coefficients  (one workgroup per sample, C <= 1024)
//   * the global-gate head: 1x1 fc, DiffSoftmax (temperature softmax + straight-through arg-max),
//     cumulative stage weights and the FLOP regulariser (single workgroup, latency-bound).
// They are latency-bound, so each is ONE launch forward and ONE backward.
#include "common.h"

namespace dynmm {

constexpr int kMaxC = 2048;     // ResNet-50 stage 4: 2048 channels
constexpr int kMaxHid = 128;

struct SeParams { const float* p[8]; };   // W1r b1r W2r b2r W1d b1d W2d b2d

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

// one modality: s[C] (LDS) -> h[Hd] (LDS, also saved) -> g[C] (saved)
__device__ void se_mlp_fwd(const float* s, const float* W1, const float* b1, const float* W2,
                           const float* b2, float* h_lds, float* h_out, float* g_out, int C, int Hd) {
    for (int j = threadIdx.x; j < Hd; j += blockDim.x) {
        float acc = b1[j];
        for (int c = 0; c < C; ++c) acc += W1[j * C + c] * s[c];
        acc = acc > 0.f ? acc : 0.f;
        h_lds[j] = acc;
        h_out[j] = acc;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = b2[c];
        for (int j = 0; j < Hd; ++j) acc += W2[c * Hd + j] * h_lds[j];
        g_out[c] = sigmoidf_(acc);
    }
    __syncthreads();
}

__global__ void __launch_bounds__(256) se_coeff_fwd_kernel(
    const float* __restrict__ sr, const float* __restrict__ sd, SeParams P,
    const float* __restrict__ wc, int wc_stride, float* __restrict__ a, float* __restrict__ b,
    float* __restrict__ hr, float* __restrict__ hd, float* __restrict__ gr, float* __restrict__ gd,
    int C, int use_se) {
    __shared__ float s_lds[kMaxC];
    __shared__ float h_lds[kMaxHid];
    const int n = blockIdx.x;
    const int Hd = C / 16;
    const float w = wc ? wc[(size_t)n * wc_stride] : 0.f;
    if (use_se) {
        for (int c = threadIdx.x; c < C; c += blockDim.x) s_lds[c] = sr[(size_t)n * C + c];
        __syncthreads();
        se_mlp_fwd(s_lds, P.p[0], P.p[1], P.p[2], P.p[3], h_lds, hr + (size_t)n * Hd, gr + (size_t)n * C, C, Hd);
        for (int c = threadIdx.x; c < C; c += blockDim.x) s_lds[c] = sd[(size_t)n * C + c];
        __syncthreads();
        se_mlp_fwd(s_lds, P.p[4], P.p[5], P.p[6], P.p[7], h_lds, hd + (size_t)n * Hd, gd + (size_t)n * C, C, Hd);
    }
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float g_r = use_se ? gr[(size_t)n * C + c] : 1.f;
        const float g_d = use_se ? gd[(size_t)n * C + c] : 1.f;
        // out = w*rgb + (1-w)*(rgb*g_r + depth*g_d)
        a[(size_t)n * C + c] = w + (1.f - w) * g_r;
        b[(size_t)n * C + c] = (1.f - w) * g_d;
    }
}

// backward of one modality's MLP for one sample.  dg[C] in LDS (grad wrt g), produces ds[C] and this
// sample's pre-activation gradients dz2[C] (second layer) and dh[Hd] (first layer) in global scratch.  The
// parameter gradients are sums over the samples of outer products of those with the saved activations; they
// are formed by mlp_param_grad_kernel in a fixed sample order (no float atomics => bit-reproducible).
__device__ void se_mlp_bwd(const float* dg, const float* h, const float* g, const float* W1, const float* W2,
                           float* dz2_lds, float* dh_lds, float* dz2_out, float* dh_out, float* ds_out,
                           int C, int Hd) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float gg = g[c];
        const float dz = dg[c] * gg * (1.f - gg);
        dz2_lds[c] = dz;
        dz2_out[c] = dz;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < Hd; j += blockDim.x) {
        float acc = 0.f;
        for (int c = 0; c < C; ++c) acc += W2[c * Hd + j] * dz2_lds[c];
        acc = h[j] > 0.f ? acc : 0.f;
        dh_lds[j] = acc;
        dh_out[j] = acc;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
        for (int j = 0; j < Hd; ++j) acc += W1[j * C + c] * dh_lds[j];
        ds_out[c] = acc;
    }
    __syncthreads();
}

// Parameter gradients of up to two excitation MLPs (blockIdx.y = which) from the per-sample scratch:
//   dW1[j][c] = sum_n dh[n][j] s[n][c]   db1[j] = sum_n dh[n][j]
//   dW2[c][j] = sum_n dz[n][c] h[n][j]   db2[c] = sum_n dz[n][c]          (n ascending: deterministic)
struct MlpGradJob {
    const float* s;    // [N][C]  layer-1 input (pooled features)
    const float* h;    // [N][Hd] layer-1 output (post-ReLU)
    const float* dz;   // [N][C]  scratch
    const float* dh;   // [N][Hd] scratch
    float* dW1; float* db1; float* dW2; float* db2;
    int s2_split;      // > 0: s is the concatenation [s (first s2_split cols) ; s2] (reweigh gate)
    const float* s2;
};
struct MlpGradJobs { MlpGradJob j[2]; };

__global__ void __launch_bounds__(256) mlp_param_grad_kernel(MlpGradJobs jobs, int N, int C, int Hd) {
    const MlpGradJob J = jobs.j[blockIdx.y];
    const int e0 = blockIdx.x * 256 + threadIdx.x;
    const int nW = Hd * C;
    if (e0 < nW) {                                   // dW1[j][c]
        const int j = e0 / C, c = e0 - j * C;
        float acc = 0.f;
        if (J.s2_split > 0) {
            const int c1 = J.s2_split, c2 = C - c1;
            for (int n = 0; n < N; ++n) {
                const float sv = c < c1 ? J.s[(size_t)n * c1 + c] : J.s2[(size_t)n * c2 + (c - c1)];
                acc += J.dh[(size_t)n * Hd + j] * sv;
            }
        } else {
            for (int n = 0; n < N; ++n) acc += J.dh[(size_t)n * Hd + j] * J.s[(size_t)n * C + c];
        }
        J.dW1[e0] = acc;
        return;
    }
    int e = e0 - nW;
    if (e < Hd) {                                    // db1[j]
        float acc = 0.f;
        for (int n = 0; n < N; ++n) acc += J.dh[(size_t)n * Hd + e];
        J.db1[e] = acc;
        return;
    }
    e -= Hd;
    if (e < nW) {                                    // dW2[c][j]
        const int c = e / Hd, j = e - c * Hd;
        float acc = 0.f;
        for (int n = 0; n < N; ++n) acc += J.dz[(size_t)n * C + c] * J.h[(size_t)n * Hd + j];
        J.dW2[e] = acc;
        return;
    }
    e -= nW;
    if (e < C) {                                     // db2[c]
        float acc = 0.f;
        for (int n = 0; n < N; ++n) acc += J.dz[(size_t)n * C + e];
        J.db2[e] = acc;
    }
}

static int launch_mlp_param_grad(const MlpGradJobs& jobs, int njobs, int N, int C, int Hd, hipStream_t st) {
    const int total = 2 * Hd * C + Hd + C;
    hipLaunchKernelGGL(mlp_param_grad_kernel, dim3(ceil_div(total, 256), njobs), dim3(256), 0, st, jobs, N, C, Hd);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

__global__ void __launch_bounds__(256) se_coeff_bwd_kernel(
    const float* __restrict__ da, const float* __restrict__ db, const float* __restrict__ sr,
    const float* __restrict__ sd, SeParams P, const float* __restrict__ wc, int wc_stride,
    const float* __restrict__ hr, const float* __restrict__ hd, const float* __restrict__ gr,
    const float* __restrict__ gd, float* __restrict__ ws, float* __restrict__ dsr, float* __restrict__ dsd,
    float* __restrict__ dwc, int dwc_stride, int N, int C, int use_se) {
    __shared__ float dg_lds[kMaxC];
    __shared__ float dz_lds[kMaxC];
    __shared__ float dh_lds[kMaxHid];
    __shared__ float red[4];
    const int n = blockIdx.x;
    const int Hd = C / 16;
    const float w = wc ? wc[(size_t)n * wc_stride] : 0.f;
    const float* dan = da + (size_t)n * C;
    const float* dbn = db + (size_t)n * C;
    // dw = sum_c da*(1-g_r) - db*g_d
    if (dwc) {
        float acc = 0.f;
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            const float g_r = use_se ? gr[(size_t)n * C + c] : 1.f;
            const float g_d = use_se ? gd[(size_t)n * C + c] : 1.f;
            acc += dan[c] * (1.f - g_r) - dbn[c] * g_d;
        }
        const float t = block_reduce_sum_256<float>(acc, red);
        if (threadIdx.x == 0) dwc[(size_t)n * dwc_stride] = t;
    }
    if (!use_se) return;
    for (int c = threadIdx.x; c < C; c += blockDim.x) dg_lds[c] = dan[c] * (1.f - w);
    __syncthreads();
    // scratch layout: dz_r [N][C] | dh_r [N][Hd] | dz_d [N][C] | dh_d [N][Hd]
    float* const dz_r = ws, * const dh_r = dz_r + (size_t)N * C;
    float* const dz_d = dh_r + (size_t)N * Hd, * const dh_d = dz_d + (size_t)N * C;
    se_mlp_bwd(dg_lds, hr + (size_t)n * Hd, gr + (size_t)n * C, P.p[0], P.p[2], dz_lds, dh_lds,
               dz_r + (size_t)n * C, dh_r + (size_t)n * Hd, dsr + (size_t)n * C, C, Hd);
    for (int c = threadIdx.x; c < C; c += blockDim.x) dg_lds[c] = dbn[c] * (1.f - w);
    __syncthreads();
    se_mlp_bwd(dg_lds, hd + (size_t)n * Hd, gd + (size_t)n * C, P.p[4], P.p[6], dz_lds, dh_lds,
               dz_d + (size_t)n * C, dh_d + (size_t)n * Hd, dsd + (size_t)n * C, C, Hd);
}

// ------------------------------------------------------------------------------------------------
// gate head (single workgroup; N samples looped by the threads)
// ------------------------------------------------------------------------------------------------
constexpr int kBranches = 5;

__global__ void __launch_bounds__(256) gate_head_fwd_kernel(
    const float* __restrict__ pooled, const float* __restrict__ fc, float* __restrict__ weight,
    float* __restrict__ wcum, float* __restrict__ soft, float* __restrict__ flop_loss,
    const float* __restrict__ flop_table, const int* __restrict__ force_branch, int N, int J, float temp, int hard,
    int mode) {
    __shared__ float red[4];
    float col[kBranches] = {0.f, 0.f, 0.f, 0.f, 0.f};   // this thread's partial column sums of weight
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float w[kBranches];
        if (mode == 0) {
            float z[kBranches];
            float zmax = -INFINITY;
#pragma unroll
            for (int k = 0; k < kBranches; ++k) {
                float acc = 0.f;
                for (int j = 0; j < J; ++j) acc += fc[k * J + j] * pooled[(size_t)n * J + j];
                z[k] = acc / temp;
                zmax = fmaxf(zmax, z[k]);
            }
            float den = 0.f;
#pragma unroll
            for (int k = 0; k < kBranches; ++k) { z[k] = expf(z[k] - zmax); den += z[k]; }
            int arg = 0;
            float best = -1.f;
#pragma unroll
            for (int k = 0; k < kBranches; ++k) {
                z[k] = z[k] / den;
                soft[(size_t)n * kBranches + k] = z[k];
                if (z[k] > best) { best = z[k]; arg = k; }   // first maximum, as torch.max
            }
            // benchmark / test knob: a FIXED branch per sample replaces the arg-max of the hard decision (the
            // gate network, its soft output and its straight-through gradient are evaluated as usual)
            if (force_branch) arg = min(max(force_branch[n], 0), kBranches - 1);
#pragma unroll
            for (int k = 0; k < kBranches; ++k) {
                // straight-through value: (y_hard - y_soft) + y_soft, evaluated in that order
                w[k] = hard ? ((k == arg ? 1.f : 0.f) - z[k]) + z[k] : z[k];
            }
        } else {
#pragma unroll
            for (int k = 0; k < kBranches; ++k) {
                w[k] = weight[(size_t)n * kBranches + k];
                soft[(size_t)n * kBranches + k] = w[k];
            }
        }
#pragma unroll
        for (int k = 0; k < kBranches; ++k) {
            if (mode == 0) weight[(size_t)n * kBranches + k] = w[k];
            col[k] += w[k];
        }
        const float c1 = w[0], c2 = w[0] + w[1], c3 = c2 + w[2];
        wcum[(size_t)n * 4 + 0] = c1;
        wcum[(size_t)n * 4 + 1] = c2;
        wcum[(size_t)n * 4 + 2] = c3;
        wcum[(size_t)n * 4 + 3] = 1.f - w[4];
    }
    float loss = 0.f;
#pragma unroll
    for (int k = 0; k < kBranches; ++k) {
        const float t = block_reduce_sum_256<float>(col[k], red);
        if (threadIdx.x == 0) loss += (t / (float)N) * flop_table[k];
    }
    if (threadIdx.x == 0) flop_loss[0] = loss / (float)kBranches;
}

constexpr int kMaxGateN = 2048;

__global__ void __launch_bounds__(256) gate_head_bwd_kernel(
    const float* __restrict__ d_weight, const float* __restrict__ d_wcum,
    const float* __restrict__ d_loss, const float* __restrict__ pooled, const float* __restrict__ fc,
    const float* __restrict__ soft, const float* __restrict__ flop_table,
    float* __restrict__ d_pooled, float* __restrict__ d_fc, int N, int J, float temp) {
    __shared__ float dz_lds[kMaxGateN * kBranches];
    const float dl = d_loss ? d_loss[0] : 0.f;
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float dw[kBranches];
#pragma unroll
        for (int k = 0; k < kBranches; ++k) {
            dw[k] = (d_weight ? d_weight[(size_t)n * kBranches + k] : 0.f) +
                    dl * flop_table[k] / ((float)kBranches * (float)N);
        }
        if (d_wcum) {
            const float c1 = d_wcum[(size_t)n * 4 + 0], c2 = d_wcum[(size_t)n * 4 + 1];
            const float c3 = d_wcum[(size_t)n * 4 + 2], c4 = d_wcum[(size_t)n * 4 + 3];
            dw[0] += c1 + c2 + c3;
            dw[1] += c2 + c3;
            dw[2] += c3;
            dw[4] -= c4;
        }
        // through (y_hard - sg(y_soft)) + y_soft : d y_soft = dw ; softmax backward, /temp
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < kBranches; ++k) dot += soft[(size_t)n * kBranches + k] * dw[k];
        float dz[kBranches];
#pragma unroll
        for (int k = 0; k < kBranches; ++k) {
            dz[k] = soft[(size_t)n * kBranches + k] * (dw[k] - dot) / temp;
            dz_lds[n * kBranches + k] = dz[k];
        }
        for (int j = 0; j < J; ++j) {
            float acc = 0.f;
#pragma unroll
            for (int k = 0; k < kBranches; ++k) acc += dz[k] * fc[k * J + j];
            d_pooled[(size_t)n * J + j] = acc;
        }
    }
    __syncthreads();
    // d_fc[k][j] = sum_n dz[n][k] * pooled[n][j], samples in ascending order (deterministic, no atomics)
    for (int e = threadIdx.x; e < kBranches * J; e += blockDim.x) {
        const int k = e / J, j = e - k * J;
        float acc = 0.f;
        for (int n = 0; n < N; ++n) acc += dz_lds[n * kBranches + k] * pooled[(size_t)n * J + j];
        d_fc[e] = acc;
    }
}


// K16 — gate-decision compaction (new capability; the reference never skips compute, …globalgate.py:276-310).
// From the one-hot gate weights [N,5]:  branch[n] = arg-max (first maximum);  order = the samples sorted by
// branch, DESCENDING and stable;  inv = its inverse;  counts[j-1] = #{n : branch[n] >= j}, j = 1..4.
// In the sorted batch the samples that still need depth stage j are the PREFIX of length counts[j-1], so every
// depth-encoder stage runs on a contiguous prefix view — no per-stage gather / scatter, one small device-to-host
// read (4 ints) per forward.  Single workgroup: N <= kMaxGateN.
__global__ void __launch_bounds__(256) gate_decide_kernel(const float* __restrict__ weight, int* __restrict__ branch,
                                                          int* __restrict__ order, int* __restrict__ inv,
                                                          int* __restrict__ counts, int N) {
    __shared__ unsigned char key[kMaxGateN];        // 4 - branch: ascending key == descending branch
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        int arg = 0;
        float best = weight[(size_t)n * kBranches];
#pragma unroll
        for (int k = 1; k < kBranches; ++k) {
            const float v = weight[(size_t)n * kBranches + k];
            if (v > best) { best = v; arg = k; }
        }
        branch[n] = arg;
        key[n] = (unsigned char)(kBranches - 1 - arg);
    }
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const int kn = key[n];
        int pos = 0;
        for (int m = 0; m < N; ++m) {
            const int km = key[m];
            pos += (km < kn || (km == kn && m < n)) ? 1 : 0;
        }
        order[pos] = n;
        inv[n] = pos;
    }
    if (threadIdx.x < kBranches - 1) {
        const int j = threadIdx.x + 1;                // stage j runs for branch >= j  <=>  key <= 4 - j
        int c = 0;
        for (int m = 0; m < N; ++m) c += (key[m] <= kBranches - 1 - j) ? 1 : 0;
        counts[threadIdx.x] = c;
    }
}


// ------------------------------------------------------------------------------------------------
// SkipESANet per-stage gate (rgb_depth_fusion.py:29-65, model_utils.py:54-70, model_skip_mod.py:248-311)
//   p      = [GAP(rgb); GAP(depth)]                       (2C values, from gap2)
//   g      = sigmoid(W2 relu(W1 p + b1) + b2)             (SE excitation over the concatenation)
//   s      = mean_{c,h,w}(x * g) = sum_c g[c] p[c] / 2C   (the feature map is never re-read)
//   w      = sigmoid(s);  logits = [w, 1-w] / temp
//   y      = gumbel_softmax(logits, tau=1, hard)          (G = -log E, E ~ Exp(1): given or Philox)
//   prev  -> b1 = y1 * prev ; b0 = 1 - b1                 (chained stage weights)
// and, in the same launch, the blend coefficients of THIS stage's fusion
//   out = w0*rgb + w1*(rgb + depth) = (w0 + w1)*rgb + w1*depth.
// One workgroup per sample; latency-bound.
// ------------------------------------------------------------------------------------------------
constexpr int kAux = 6;   // aux[n] = {w, ysoft0, ysoft1, y1 (forward value), E0, E1}

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float exp1_from_bits(uint32_t x) {
    const float u = ((float)(x >> 8) + 0.5f) * (1.f / 16777216.f);   // (0,1)
    return -logf(u);
}

struct MlpParams { const float* p[4]; };   // W1[2C/16,2C] b1 W2[2C,2C/16] b2

__global__ void __launch_bounds__(256) reweigh_fwd_kernel(
    const float* __restrict__ sr, const float* __restrict__ sd, MlpParams P,
    const float* __restrict__ wblend, int blend_mode, const float* __restrict__ prev, int prev_stride,
    const float* __restrict__ noise, unsigned long long seed, unsigned long long offset, float temp,
    int hard, float* __restrict__ a, float* __restrict__ b, float* __restrict__ wnext,
    float* __restrict__ h, float* __restrict__ g, float* __restrict__ aux, int C) {
    __shared__ float p_lds[kMaxC];
    __shared__ float h_lds[kMaxHid];
    __shared__ float red[4];
    const int n = blockIdx.x;
    if (a) {
        float fa = 1.f, fb = 1.f;                       // mode 1: rgb + depth
        if (blend_mode == 0) fb = 0.f;                  // rgb only
        else if (blend_mode == 2) { fb = wblend[2 * n + 1]; fa = wblend[2 * n] + fb; }
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            a[(size_t)n * C + c] = fa;
            b[(size_t)n * C + c] = fb;
        }
    }
    if (!wnext) return;
    const int C2 = 2 * C, Hd = C2 / 16;
    for (int c = threadIdx.x; c < C2; c += blockDim.x)
        p_lds[c] = c < C ? sr[(size_t)n * C + c] : sd[(size_t)n * C + (c - C)];
    __syncthreads();
    float* gn = g + (size_t)n * C2;
    se_mlp_fwd(p_lds, P.p[0], P.p[1], P.p[2], P.p[3], h_lds, h + (size_t)n * Hd, gn, C2, Hd);
    float acc = 0.f;
    for (int c = threadIdx.x; c < C2; c += blockDim.x) acc += gn[c] * p_lds[c];   // same thread wrote gn[c]
    const float tot = block_reduce_sum_256<float>(acc, red);
    if (threadIdx.x == 0) {
        const float w = sigmoidf_(tot / (float)C2);
        float e0, e1;
        if (noise) { e0 = noise[2 * n]; e1 = noise[2 * n + 1]; }
        else {
            uint32_t r[4];
            philox4x32_10((uint32_t)n, 0u, (uint32_t)offset, (uint32_t)(offset >> 32), (uint32_t)seed,
                          (uint32_t)(seed >> 32), r);
            e0 = exp1_from_bits(r[0]); e1 = exp1_from_bits(r[1]);
        }
        const float z0 = w / temp - logf(e0), z1 = (1.f - w) / temp - logf(e1);
        const float zm = fmaxf(z0, z1);
        const float x0 = expf(z0 - zm), x1 = expf(z1 - zm);
        const float ys0 = x0 / (x0 + x1), ys1 = x1 / (x0 + x1);
        float y0 = ys0, y1 = ys1;
        if (hard) {                                      // (y_hard - sg(y_soft)) + y_soft, first maximum
            const int arg = ys1 > ys0 ? 1 : 0;
            y0 = ((arg == 0 ? 1.f : 0.f) - ys0) + ys0;
            y1 = ((arg == 1 ? 1.f : 0.f) - ys1) + ys1;
        }
        float* ax = aux + (size_t)n * kAux;
        ax[0] = w; ax[1] = ys0; ax[2] = ys1; ax[3] = y1; ax[4] = e0; ax[5] = e1;
        if (prev) {
            const float b1 = y1 * prev[(size_t)n * prev_stride];
            y0 = 1.f - b1; y1 = b1;
        }
        wnext[2 * n] = y0;
        wnext[2 * n + 1] = y1;
    }
}

__global__ void __launch_bounds__(256) reweigh_bwd_kernel(
    const float* __restrict__ d_wnext, const float* __restrict__ da, const float* __restrict__ db,
    const float* __restrict__ sr, const float* __restrict__ sd, MlpParams P,
    const float* __restrict__ prev, int prev_stride, const float* __restrict__ h,
    const float* __restrict__ g, const float* __restrict__ aux, float* __restrict__ ws, float* __restrict__ dsr,
    float* __restrict__ dsd, float* __restrict__ d_wblend, float* __restrict__ d_prev, float temp, int N, int C) {
    __shared__ float p_lds[kMaxC];
    __shared__ float dg_lds[kMaxC];
    __shared__ float dz_lds[kMaxC];
    __shared__ float ds_lds[kMaxC];
    __shared__ float dh_lds[kMaxHid];
    __shared__ float red[4];
    __shared__ float ds_sh;
    const int n = blockIdx.x;
    if (d_wblend) {        // w0*rgb + w1*(rgb+depth): d w0 = <g,rgb> ; d w1 = <g,rgb> + <g,depth>
        float s0 = 0.f, s1 = 0.f;
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            s0 += da[(size_t)n * C + c];
            s1 += db[(size_t)n * C + c];
        }
        const float t0 = block_reduce_sum_256<float>(s0, red);
        const float t1 = block_reduce_sum_256<float>(s1, red);
        if (threadIdx.x == 0) { d_wblend[2 * n] = t0; d_wblend[2 * n + 1] = t0 + t1; }
    }
    if (!d_wnext) return;
    const int C2 = 2 * C, Hd = C2 / 16;
    if (threadIdx.x == 0) {
        const float* ax = aux + (size_t)n * kAux;
        const float w = ax[0], ys0 = ax[1], ys1 = ax[2], y1 = ax[3];
        float dy0 = d_wnext[2 * n], dy1 = d_wnext[2 * n + 1];
        if (prev) {
            const float db1 = dy1 - dy0;                 // b0 = 1 - b1
            if (d_prev) d_prev[n] = db1 * y1;
            dy1 = db1 * prev[(size_t)n * prev_stride];
            dy0 = 0.f;
        }
        const float dot = ys0 * dy0 + ys1 * dy1;
        const float dz0 = ys0 * (dy0 - dot), dz1 = ys1 * (dy1 - dot);
        ds_sh = (dz0 - dz1) / temp * w * (1.f - w);
    }
    for (int c = threadIdx.x; c < C2; c += blockDim.x)
        p_lds[c] = c < C ? sr[(size_t)n * C + c] : sd[(size_t)n * C + (c - C)];
    __syncthreads();
    const float dsn = ds_sh / (float)C2;
    for (int c = threadIdx.x; c < C2; c += blockDim.x) dg_lds[c] = dsn * p_lds[c];
    __syncthreads();
    const float* gn = g + (size_t)n * C2;
    float* const dz_g = ws, * const dh_g = ws + (size_t)N * C2;        // scratch: dz [N][2C] | dh [N][2C/16]
    se_mlp_bwd(dg_lds, h + (size_t)n * Hd, gn, P.p[0], P.p[2], dz_lds, dh_lds, dz_g + (size_t)n * C2,
               dh_g + (size_t)n * Hd, ds_lds, C2, Hd);
    for (int c = threadIdx.x; c < C2; c += blockDim.x) {
        const float v = ds_lds[c] + dsn * gn[c];
        if (c < C) dsr[(size_t)n * C + c] = v;
        else dsd[(size_t)n * C + (c - C)] = v;
    }
}

}  // namespace dynmm

using namespace dynmm;

extern "C" int dynmm_se_coeff_fwd(const float* sr, const float* sd, const float* const* params,
                                  const float* wc, int wc_stride, float* a, float* b, float* hr,
                                  float* hd, float* gr, float* gd, int N, int C, int use_se,
                                  void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!a || !b || N <= 0 || C <= 0 || C > kMaxC) return DYNMM_EINVAL;
    SeParams P{};
    if (use_se) {
        if (!sr || !sd || !params || !hr || !hd || !gr || !gd) return DYNMM_EINVAL;
        if (C % 16 != 0 || C / 16 > kMaxHid) return DYNMM_EUNSUPPORTED;
        for (int i = 0; i < 8; ++i) {
            if (!params[i]) return DYNMM_EINVAL;
            P.p[i] = params[i];
        }
    }
    hipLaunchKernelGGL(se_coeff_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, sr, sd, P, wc,
                       wc_stride, a, b, hr, hd, gr, gd, C, use_se);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" size_t dynmm_se_coeff_bwd_workspace_bytes(int N, int C) {
    return sizeof(float) * 2 * (size_t)N * (size_t)(C + C / 16);
}

extern "C" int dynmm_se_coeff_bwd(const float* da, const float* db, const float* sr, const float* sd,
                                  const float* const* params, const float* wc, int wc_stride,
                                  const float* hr, const float* hd, const float* gr, const float* gd,
                                  float* const* dparams, float* dsr, float* dsd, float* dwc,
                                  int dwc_stride, float* workspace, int N, int C, int use_se, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!da || !db || N <= 0 || C <= 0 || C > kMaxC) return DYNMM_EINVAL;
    SeParams P{};
    hipStream_t st = (hipStream_t)stream;
    if (use_se) {
        if (!sr || !sd || !params || !dparams || !hr || !hd || !gr || !gd || !dsr || !dsd)
            return DYNMM_EINVAL;
        if (!workspace) return DYNMM_EWORKSPACE;
        if (C % 16 != 0 || C / 16 > kMaxHid) return DYNMM_EUNSUPPORTED;
        for (int i = 0; i < 8; ++i) {
            if (!params[i] || !dparams[i]) return DYNMM_EINVAL;
            P.p[i] = params[i];
        }
    }
    hipLaunchKernelGGL(se_coeff_bwd_kernel, dim3(N), dim3(256), 0, st, da, db, sr, sd, P, wc,
                       wc_stride, hr, hd, gr, gd, workspace, dsr, dsd, dwc, dwc_stride, N, C, use_se);
    DYNMM_LAUNCH_CHECK();
    if (use_se) {
        const int Hd = C / 16;
        float* dz_r = workspace, *dh_r = dz_r + (size_t)N * C, *dz_d = dh_r + (size_t)N * Hd, *dh_d = dz_d + (size_t)N * C;
        MlpGradJobs jobs{};
        jobs.j[0] = MlpGradJob{sr, hr, dz_r, dh_r, dparams[0], dparams[1], dparams[2], dparams[3], 0, nullptr};
        jobs.j[1] = MlpGradJob{sd, hd, dz_d, dh_d, dparams[4], dparams[5], dparams[6], dparams[7], 0, nullptr};
        return launch_mlp_param_grad(jobs, 2, N, C, Hd, st);
    }
    return DYNMM_OK;
}

extern "C" int dynmm_gate_head_fwd(const float* pooled, const float* fc, float* weight, float* wcum,
                                   float* soft, float* flop_loss, const float* flop_table,
                                   const int* force_branch, int N, int J, float temp, int hard, int mode,
                                   void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!weight || !wcum || !soft || !flop_loss || !flop_table || N <= 0) return DYNMM_EINVAL;
    if (mode == 0 && (!pooled || !fc || J <= 0 || !(temp > 0.f))) return DYNMM_EINVAL;
    hipLaunchKernelGGL(gate_head_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pooled, fc,
                       weight, wcum, soft, flop_loss, flop_table, force_branch, N, J, temp, hard, mode);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_gate_decide(const float* weight, int* branch, int* order, int* inv, int* counts, int N,
                                 void* stream) {
    (void)hipGetLastError();
    if (!weight || !branch || !order || !inv || !counts || N <= 0) return DYNMM_EINVAL;
    if (N > kMaxGateN) return DYNMM_EUNSUPPORTED;
    hipLaunchKernelGGL(gate_decide_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, weight, branch, order, inv,
                       counts, N);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_gate_head_bwd(const float* d_weight, const float* d_wcum, const float* d_loss,
                                   const float* pooled, const float* fc, const float* soft,
                                   const float* flop_table, float* d_pooled, float* d_fc, int N,
                                   int J, float temp, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!pooled || !fc || !soft || !flop_table || !d_pooled || !d_fc || N <= 0 || J <= 0)
        return DYNMM_EINVAL;
    if (N > kMaxGateN) return DYNMM_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gate_head_bwd_kernel, dim3(1), dim3(256), 0, st, d_weight, d_wcum, d_loss,
                       pooled, fc, soft, flop_table, d_pooled, d_fc, N, J, temp);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_reweigh_fwd(const float* sr, const float* sd, const float* const* params,
                                 const float* wblend, int blend_mode, const float* prev,
                                 int prev_stride, const float* noise, unsigned long long seed,
                                 unsigned long long offset, float temp, int hard, float* a, float* b,
                                 float* wnext, float* h, float* g, float* aux, int N, int C,
                                 void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (N <= 0 || C <= 0 || (!a && !wnext) || (a && !b)) return DYNMM_EINVAL;
    if (blend_mode < 0 || blend_mode > 2 || (a && blend_mode == 2 && !wblend)) return DYNMM_EINVAL;
    MlpParams P{};
    if (wnext) {
        if (!sr || !sd || !params || !h || !g || !aux || !(temp > 0.f)) return DYNMM_EINVAL;
        if ((2 * C) % 16 != 0 || 2 * C > kMaxC || (2 * C) / 16 > kMaxHid) return DYNMM_EUNSUPPORTED;
        for (int i = 0; i < 4; ++i) {
            if (!params[i]) return DYNMM_EINVAL;
            P.p[i] = params[i];
        }
    }
    hipLaunchKernelGGL(reweigh_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, sr, sd, P, wblend,
                       blend_mode, prev, prev_stride, noise, seed, offset, temp, hard, a, b, wnext, h, g,
                       aux, C);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" size_t dynmm_reweigh_bwd_workspace_bytes(int N, int C) {
    return sizeof(float) * (size_t)N * (size_t)(2 * C + (2 * C) / 16);
}

extern "C" int dynmm_reweigh_bwd(const float* d_wnext, const float* da, const float* db,
                                 const float* sr, const float* sd, const float* const* params,
                                 const float* prev, int prev_stride, const float* h, const float* g,
                                 const float* aux, float* const* dparams, float* dsr, float* dsd,
                                 float* d_wblend, float* d_prev, float* workspace, float temp, int N, int C,
                                 void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (N <= 0 || C <= 0 || (!d_wnext && !d_wblend)) return DYNMM_EINVAL;
    if (d_wblend && (!da || !db)) return DYNMM_EINVAL;
    MlpParams P{};
    hipStream_t st = (hipStream_t)stream;
    if (d_wnext) {
        if (!sr || !sd || !params || !dparams || !h || !g || !aux || !dsr || !dsd || !(temp > 0.f))
            return DYNMM_EINVAL;
        if (!workspace) return DYNMM_EWORKSPACE;
        if ((2 * C) % 16 != 0 || 2 * C > kMaxC || (2 * C) / 16 > kMaxHid) return DYNMM_EUNSUPPORTED;
        for (int i = 0; i < 4; ++i) {
            if (!params[i] || !dparams[i]) return DYNMM_EINVAL;
            P.p[i] = params[i];
        }
    }
    hipLaunchKernelGGL(reweigh_bwd_kernel, dim3(N), dim3(256), 0, st, d_wnext, da, db, sr, sd, P, prev,
                       prev_stride, h, g, aux, workspace, dsr, dsd, d_wblend, d_prev, temp, N, C);
    DYNMM_LAUNCH_CHECK();
    if (d_wnext) {
        const int C2 = 2 * C, Hd = C2 / 16;
        MlpGradJobs jobs{};
        jobs.j[0] = MlpGradJob{sr, h, workspace, workspace + (size_t)N * C2, dparams[0], dparams[1], dparams[2],
                               dparams[3], C, sd};
        return launch_mlp_param_grad(jobs, 1, N, C2, Hd, st);
    }
    return DYNMM_OK;
}
