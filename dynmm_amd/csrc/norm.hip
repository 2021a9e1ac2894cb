// BatchNorm2d (train + eval, forward + backward) and the activation/bias backward pass.
// All of these are HBM-bound streaming passes over NCHW planes: one workgroup owns whole (n,c)
// planes (so the channel is uniform per workgroup), lanes walk the contiguous HW axis with 16-byte
// accesses, per-channel reductions are wave-shuffle -> LDS -> one fp64 atomic per workgroup.
#include "common.h"
#include "vec.h"

namespace dynmm {

// ReLU decisions of a BatchNorm + residual + ReLU (resnet.py:136-147: bn2 + identity) as ONE BIT per element, written by the
// forward's normalise pass and read by the two backward passes in place of the output tensor (4 bytes per element, twice).
// Layout: the V = 4 kernels walk a plane in wave groups of 256 consecutive elements (lane l: elements 4 l .. 4 l + 3); group k
// of plane p owns words [(p * groups + k) * 4, + 4): word j = the wave's ballot of "element 4 l + j is positive".  A bit is
// addressed by (plane, element) alone, so the reduce pass (whole planes per workgroup) and the apply pass (8192-element
// chunks) read the same words the forward wrote.
__device__ __forceinline__ size_t mask_word(int plane, int groups, int i) { return ((size_t)plane * groups + (i >> 8)) * 4; }

static inline int reduce_splits(int N, int C) {
    int s = 2048 / (C > 0 ? C : 1);
    if (s < 1) s = 1;
    if (s > N) s = N;
    return s;
}

// (A "last workgroup adds the partials" reduction without the memset was built and measured: its two
// __threadfence()s per workgroup are L2 write-backs on gfx950 and cost 12 ms per step — fp64 atomics into a
// zeroed buffer stay.)
// sums[c] = sum x ; sums[C+c] = sum x^2
template <int V>
__global__ void __launch_bounds__(256) bn_stats_kernel(const float* __restrict__ x,
                                                       double* __restrict__ sums,
                                                       int N, int C, int HW) {
    __shared__ float red[4];
    const int c = blockIdx.x, S = gridDim.y;
    float s1 = 0.f, s2 = 0.f;
    for (int n = blockIdx.y; n < N; n += S) {
        const float* p = x + ((size_t)n * C + c) * HW;
        float a1 = 0.f, a2 = 0.f;
        for (int i = threadIdx.x * V; i < HW; i += 256 * V) {
            float v[V];
            vload<V>(p + i, v);
#pragma unroll
            for (int j = 0; j < V; ++j) { a1 += v[j]; a2 += v[j] * v[j]; }
        }
        s1 += a1; s2 += a2;
    }
    const float t1 = block_reduce_sum_256<float>(s1, red);
    const float t2 = block_reduce_sum_256<float>(s2, red);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[c], (double)t1);
        atomicAdd(&sums[C + c], (double)t2);
    }
}

template <int V>
__global__ void __launch_bounds__(256) bn_apply_kernel(
    const float* __restrict__ x, const double* __restrict__ sums, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ running_mean, float* __restrict__ running_var,
    float* __restrict__ save_mean, float* __restrict__ save_invstd, const float* __restrict__ residual,
    float* __restrict__ y, long long* __restrict__ num_batches_tracked, int N, int C, int HW, float eps,
    float momentum, int training, int act, int chunk, unsigned long long* __restrict__ mask_bits) {
    const int plane = blockIdx.x;
    const int c = plane % C;
    float mean, invstd;
    if (training && num_batches_tracked && plane == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        *num_batches_tracked += 1;          // nn.BatchNorm2d's step counter, without a launch of its own
    if (training) {
        // training = number of slabs of `sums` ([training][2][C]: a producer that spreads its atomics over several slabs —
        // conv_wino.hip STATS — or 1), added in slab order
        double t1 = 0.0, t2 = 0.0;
        for (int k = 0; k < training; ++k) {
            t1 += sums[(size_t)k * 2 * C + c];
            t2 += sums[(size_t)k * 2 * C + C + c];
        }
        const double M = (double)N * HW;
        const double mu = t1 / M;
        double var = t2 / M - mu * mu;
        if (var < 0.0) var = 0.0;
        mean = (float)mu;
        invstd = (float)(1.0 / sqrt(var + (double)eps));
        if (plane < C && blockIdx.y == 0 && threadIdx.x == 0) {
            if (save_mean) save_mean[c] = mean;
            if (save_invstd) save_invstd[c] = invstd;
            if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
            if (running_var) {
                const double unb = var * (M / (M - 1.0));
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
            }
        }
    } else {
        mean = running_mean[c];
        invstd = 1.f / sqrtf(running_var[c] + eps);
        if (plane < C && blockIdx.y == 0 && threadIdx.x == 0) {
            if (save_mean) save_mean[c] = mean;
            if (save_invstd) save_invstd[c] = invstd;
        }
    }
    const float sc = gamma[c] * invstd;
    const float sh = fmaf(-mean, sc, beta[c]);      // explicit: bn_bwd_* re-evaluate exactly these two lines
    const size_t base = (size_t)plane * HW;
    const int beg = blockIdx.y * chunk;
    const int end = min(HW, beg + chunk);
    const int groups = (HW + 255) >> 8;
    // (mask_bits: V = 4 only; the loop bound is rounded up so that every lane of a wave takes part in the ballots)
    const int end_w = (V == 4 && mask_bits) ? beg + ((end - beg + 255) & ~255) : end;
    for (int i = beg + threadIdx.x * V; i < end_w; i += 256 * V) {
        float v[V], r[V];
        const bool in = i < end;
        if (in) {
            vload<V>(x + base + i, v);
            if (residual) vload<V>(residual + base + i, r);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float o = fmaf(v[j], sc, sh);       // (the backward re-evaluates exactly this expression)
                if (residual) o += r[j];
                v[j] = act_fwd(o, act);
            }
            vstore<V>(y + base + i, v);
        }
        if constexpr (V == 4) {
            if (mask_bits) {
                unsigned long long w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) w[j] = __ballot(in && v[j] > 0.f);
                if ((threadIdx.x & 63) == 0) {
                    unsigned long long* dst = mask_bits + mask_word(plane, groups, i);
#pragma unroll
                    for (int j = 0; j < 4; ++j) dst[j] = w[j];
                }
            }
        }
    }
}

// The per-channel part of bn_apply_kernel on its own (training): batch statistics -> mean / invstd, running
// statistics, step counter, and the affine pair scale = gamma*invstd, shift = beta - mean*scale that consumers apply on
// load when the normalised tensor is never written (pointwise.hip: gap2 / axpby_pool kernels).
__global__ void __launch_bounds__(256) bn_finalize_kernel(const double* __restrict__ sums, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, float* __restrict__ save_mean,
                                                          float* __restrict__ save_invstd,
                                                          long long* __restrict__ num_batches_tracked,
                                                          float* __restrict__ scale, float* __restrict__ shift, int N, int C,
                                                          int HW, float eps, float momentum) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c == 0 && num_batches_tracked) *num_batches_tracked += 1;
    if (c >= C) return;
    const double M = (double)N * HW;
    const double mu = sums[c] / M;
    double var = sums[C + c] / M - mu * mu;
    if (var < 0.0) var = 0.0;
    const float mean = (float)mu;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    save_mean[c] = mean;
    save_invstd[c] = invstd;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean;
    if (running_var) {
        const double unb = var * (M / (M - 1.0));
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
    const float sc = gamma[c] * invstd;
    scale[c] = sc;
    shift[c] = fmaf(-mean, sc, beta[c]);
}

// sums[c] += sum g_eff ; sums[C+c] += sum g_eff * xhat
template <int V>
__global__ void __launch_bounds__(256) bn_bwd_reduce_kernel(
    const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
    const float* __restrict__ beta, double* __restrict__ sums, int N, int C, int HW, int act,
    const unsigned long long* __restrict__ mask_bits) {
    __shared__ float red[4];
    const int c = blockIdx.x, S = gridDim.y;
    const float mu = mean[c], is = invstd[c];
    const int groups = (HW + 255) >> 8;
    const int lane = threadIdx.x & 63;
    // y == nullptr (ReLU, no residual): the mask [y > 0] is re-derived from x with the forward's own
    // arithmetic, fma(x, sc, sh) > 0 — one tensor read less in each backward pass
    const bool remask = act != DYNMM_ACT_NONE && y == nullptr && mask_bits == nullptr;
    const float sc = remask ? gamma[c] * is : 0.f;
    const float sh = remask ? fmaf(-mu, sc, beta[c]) : 0.f;
    float s1 = 0.f, s2 = 0.f;
    for (int n = blockIdx.y; n < N; n += S) {
        const size_t base = ((size_t)n * C + c) * HW;
        float a1 = 0.f, a2 = 0.f;
        for (int i = threadIdx.x * V; i < HW; i += 256 * V) {
            float gv[V], yv[V], xv[V];
            vload<V>(g + base + i, gv);
            vload<V>(x + base + i, xv);
            const bool bits = V == 4 && mask_bits != nullptr;
            if (bits) {                                 // the forward's ReLU decisions, one bit per element (ReLU only)
                const unsigned long long* wsrc = mask_bits + mask_word(n * C + c, groups, i);
#pragma unroll
                for (int j = 0; j < V; ++j) yv[j] = ((wsrc[j & 3] >> lane) & 1ull) ? 1.f : 0.f;
            } else if (act != DYNMM_ACT_NONE && !remask) {
                vload<V>(y + base + i, yv);
            }
#pragma unroll
            for (int j = 0; j < V; ++j) {
                if (remask && !bits) yv[j] = fmaf(xv[j], sc, sh);
                const float ge = (act != DYNMM_ACT_NONE) ? act_bwd(gv[j], yv[j], act) : gv[j];
                a1 += ge;
                a2 += ge * (xv[j] - mu) * is;
            }
        }
        s1 += a1; s2 += a2;
    }
    const float t1 = block_reduce_sum_256<float>(s1, red);
    const float t2 = block_reduce_sum_256<float>(s2, red);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[c], (double)t1);
        atomicAdd(&sums[C + c], (double)t2);
    }
}

template <int V>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(
    const float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ invstd, const float* __restrict__ gamma,
    const float* __restrict__ beta, const double* __restrict__ sums, float* __restrict__ dx,
    float* __restrict__ dres, float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int C, int HW,
    int training, int act, int chunk, const unsigned long long* __restrict__ mask_bits) {
    const int plane = blockIdx.x;
    const int c = plane % C;
    const float mu = mean[c], is = invstd[c];
    const int groups = (HW + 255) >> 8;
    const int lane = threadIdx.x & 63;
    const bool remask = act != DYNMM_ACT_NONE && y == nullptr && mask_bits == nullptr;
    const float sc = remask ? gamma[c] * is : 0.f;
    const float sh = remask ? fmaf(-mu, sc, beta[c]) : 0.f;
    // (training = the number of slabs [n][2][C] the two sums arrive in: 1 from bn_bwd_reduce, the slots of a convolution's
    // input-gradient epilogue otherwise — conv_wino.hip BNRED; summed here in slab order by every workgroup alike)
    double sg_d = sums[c], sgx_d = sums[C + c];
    for (int s = 1; s < training; ++s) {
        sg_d += sums[(size_t)(2 * s) * C + c];
        sgx_d += sums[(size_t)(2 * s + 1) * C + c];
    }
    const float sg = (float)sg_d, sgx = (float)sgx_d;
    if (plane < C && blockIdx.y == 0 && threadIdx.x == 0) {
        if (dgamma) dgamma[c] = sgx;
        if (dbeta) dbeta[c] = sg;
    }
    const float invM = 1.f / ((float)N * (float)HW);
    const float k0 = gamma[c] * is;
    const float m1 = training ? sg * invM : 0.f;
    const float m2 = training ? sgx * invM : 0.f;
    const size_t base = (size_t)plane * HW;
    const int beg = blockIdx.y * chunk;
    const int end = min(HW, beg + chunk);
    for (int i = beg + threadIdx.x * V; i < end; i += 256 * V) {
        float gv[V], yv[V], xv[V], o[V];
        vload<V>(g + base + i, gv);
        vload<V>(x + base + i, xv);
        const bool bits = V == 4 && mask_bits != nullptr;
        if (bits) {
            const unsigned long long* wsrc = mask_bits + mask_word(plane, groups, i);
#pragma unroll
            for (int j = 0; j < V; ++j) yv[j] = ((wsrc[j & 3] >> lane) & 1ull) ? 1.f : 0.f;
        } else if (act != DYNMM_ACT_NONE && !remask) {
            vload<V>(y + base + i, yv);
        }
#pragma unroll
        for (int j = 0; j < V; ++j) {
            if (remask && !bits) yv[j] = fmaf(xv[j], sc, sh);
            const float ge = (act != DYNMM_ACT_NONE) ? act_bwd(gv[j], yv[j], act) : gv[j];
            gv[j] = ge;
            o[j] = k0 * (ge - m1 - (xv[j] - mu) * is * m2);
        }
        vstore<V>(dx + base + i, o);
        if (dres) vstore<V>(dres + base + i, gv);
    }
}

__global__ void __launch_bounds__(256) bn_fold_kernel(
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ rm,
    const float* __restrict__ rv, const float* __restrict__ cbias, float* __restrict__ scale,
    float* __restrict__ shift, int C, float eps) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    const float b = cbias ? cbias[c] : 0.f;
    scale[c] = sc;
    shift[c] = beta[c] + (b - rm[c]) * sc;
}

// g_out = act'(y) * g ; dbias[split][c] = sum g_out over the split's samples (the splits are then summed in a
// fixed order by launch_reduce_slabs: no float atomics)
template <int V>
__global__ void __launch_bounds__(256) act_bwd_bias_kernel(
    const float* __restrict__ g, const float* __restrict__ y, float* __restrict__ gout,
    float* __restrict__ dbias, int N, int C, int HW, int act) {
    __shared__ float red[4];
    const int c = blockIdx.x, S = gridDim.y;
    float s1 = 0.f;
    for (int n = blockIdx.y; n < N; n += S) {
        const size_t base = ((size_t)n * C + c) * HW;
        float a1 = 0.f;
        for (int i = threadIdx.x * V; i < HW; i += 256 * V) {
            float gv[V], yv[V];
            vload<V>(g + base + i, gv);
            if (act != DYNMM_ACT_NONE) {
                vload<V>(y + base + i, yv);
#pragma unroll
                for (int j = 0; j < V; ++j) gv[j] = act_bwd(gv[j], yv[j], act);
            }
#pragma unroll
            for (int j = 0; j < V; ++j) a1 += gv[j];
            if (gout) vstore<V>(gout + base + i, gv);
        }
        s1 += a1;
    }
    if (dbias) {
        const float t1 = block_reduce_sum_256<float>(s1, red);
        if (threadIdx.x == 0) dbias[(size_t)blockIdx.y * C + c] = t1;
    }
}

static inline int plane_chunk(int HW, int* nchunks) {
    const int chunk = 8192;   // floats per workgroup pass: 8 x dwordx4 per lane
    *nchunks = (HW + chunk - 1) / chunk;
    return chunk;
}

}  // namespace dynmm

using namespace dynmm;

extern "C" int dynmm_bn_stats(const float* x, double* sums, int N, int C, int HW, int sums_are_zero,
                              void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!x || !sums || N <= 0 || C <= 0 || HW <= 0) return DYNMM_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (!sums_are_zero) DYNMM_HIP_TRY(hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, st));
    dim3 grid(C, reduce_splits(N, C));
    if (can_vec4(HW, {x}))
        hipLaunchKernelGGL(bn_stats_kernel<4>, grid, dim3(256), 0, st, x, sums, N, C, HW);
    else
        hipLaunchKernelGGL(bn_stats_kernel<1>, grid, dim3(256), 0, st, x, sums, N, C, HW);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_bn_apply(const float* x, const double* sums, const float* gamma,
                              const float* beta, float* running_mean, float* running_var,
                              float* save_mean, float* save_invstd, const float* residual, float* y,
                              long long* num_batches_tracked, int N, int C, int HW, float eps, float momentum,
                              int training, int act, unsigned long long* relu_bits, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!x || !gamma || !beta || !y || N <= 0 || C <= 0 || HW <= 0) return DYNMM_EINVAL;
    if (relu_bits && (act != DYNMM_ACT_RELU || !can_vec4(HW, {x, residual, y}) || (reinterpret_cast<uintptr_t>(relu_bits) & 7u)))
        return DYNMM_EUNSUPPORTED;
    if (training && (!sums || (long long)N * HW <= 1)) return DYNMM_EINVAL;
    if (!training && (!running_mean || !running_var)) return DYNMM_EINVAL;
    int nchunks;
    const int chunk = plane_chunk(HW, &nchunks);
    dim3 grid(N * C, nchunks);
    hipStream_t st = (hipStream_t)stream;
    if (can_vec4(HW, {x, residual, y}))
        hipLaunchKernelGGL(bn_apply_kernel<4>, grid, dim3(256), 0, st, x, sums, gamma, beta,
                           running_mean, running_var, save_mean, save_invstd, residual, y,
                           num_batches_tracked, N, C, HW, eps, momentum, training, act, chunk, relu_bits);
    else
        hipLaunchKernelGGL(bn_apply_kernel<1>, grid, dim3(256), 0, st, x, sums, gamma, beta,
                           running_mean, running_var, save_mean, save_invstd, residual, y,
                           num_batches_tracked, N, C, HW, eps, momentum, training, act, chunk, (unsigned long long*)nullptr);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" size_t dynmm_bn_relu_bits_words(int N, int C, int HW) {
    if (N <= 0 || C <= 0 || HW <= 0 || HW % 4 != 0) return 0;
    return (size_t)N * C * ((HW + 255) / 256) * 4;
}

extern "C" int dynmm_bn_finalize(const double* sums, const float* gamma, const float* beta, float* running_mean,
                                 float* running_var, float* save_mean, float* save_invstd,
                                 long long* num_batches_tracked, float* scale, float* shift, int N, int C, int HW,
                                 float eps, float momentum, void* stream) {
    (void)hipGetLastError();
    if (!sums || !gamma || !beta || !save_mean || !save_invstd || !scale || !shift || N <= 0 || C <= 0 || HW <= 0 ||
        (long long)N * HW <= 1)
        return DYNMM_EINVAL;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, gamma, beta,
                       running_mean, running_var, save_mean, save_invstd, num_batches_tracked, scale, shift, N, C, HW, eps,
                       momentum);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_bn_bwd_reduce(const float* g, const float* y, const float* x, const float* mean,
                                   const float* invstd, const float* gamma, const float* beta,
                                   double* sums, int N, int C, int HW, int act, int sums_are_zero,
                                   const unsigned long long* relu_bits, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!g || !x || !mean || !invstd || !sums || N <= 0 || C <= 0 || HW <= 0) return DYNMM_EINVAL;
    if (act != DYNMM_ACT_NONE && !y && !relu_bits && (act != DYNMM_ACT_RELU || !gamma || !beta)) return DYNMM_EINVAL;
    if (relu_bits && (act != DYNMM_ACT_RELU || !can_vec4(HW, {g, x}) || (reinterpret_cast<uintptr_t>(relu_bits) & 7u)))
        return DYNMM_EUNSUPPORTED;
    if (relu_bits) y = nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (!sums_are_zero) DYNMM_HIP_TRY(hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, st));
    dim3 grid(C, reduce_splits(N, C));
    if (can_vec4(HW, {g, y, x}))
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<4>, grid, dim3(256), 0, st, g, y, x, mean, invstd, gamma, beta,
                           sums, N, C, HW, act, relu_bits);
    else
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<1>, grid, dim3(256), 0, st, g, y, x, mean, invstd, gamma, beta,
                           sums, N, C, HW, act, (const unsigned long long*)nullptr);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_bn_bwd_apply(const float* g, const float* y, const float* x, const float* mean,
                                  const float* invstd, const float* gamma, const float* beta,
                                  const double* sums, float* dx, float* d_residual, float* dgamma,
                                  float* dbeta, int N, int C, int HW, int training, int act,
                                  const unsigned long long* relu_bits, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!g || !x || !mean || !invstd || !gamma || !sums || !dx || N <= 0 || C <= 0 || HW <= 0)
        return DYNMM_EINVAL;
    if (act != DYNMM_ACT_NONE && !y && !relu_bits && (act != DYNMM_ACT_RELU || !beta || d_residual)) return DYNMM_EINVAL;
    if (relu_bits && (act != DYNMM_ACT_RELU || !can_vec4(HW, {g, x, dx, d_residual}) || (reinterpret_cast<uintptr_t>(relu_bits) & 7u)))
        return DYNMM_EUNSUPPORTED;
    if (relu_bits) y = nullptr;
    int nchunks;
    const int chunk = plane_chunk(HW, &nchunks);
    dim3 grid(N * C, nchunks);
    hipStream_t st = (hipStream_t)stream;
    if (can_vec4(HW, {g, y, x, dx, d_residual}))
        hipLaunchKernelGGL(bn_bwd_apply_kernel<4>, grid, dim3(256), 0, st, g, y, x, mean, invstd,
                           gamma, beta, sums, dx, d_residual, dgamma, dbeta, N, C, HW, training, act, chunk, relu_bits);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, grid, dim3(256), 0, st, g, y, x, mean, invstd,
                           gamma, beta, sums, dx, d_residual, dgamma, dbeta, N, C, HW, training, act, chunk,
                           (const unsigned long long*)nullptr);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_bn_fold(const float* gamma, const float* beta, const float* running_mean,
                             const float* running_var, const float* conv_bias, float* scale,
                             float* shift, int C, float eps, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!gamma || !beta || !running_mean || !running_var || !scale || !shift || C <= 0)
        return DYNMM_EINVAL;
    hipLaunchKernelGGL(bn_fold_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, (hipStream_t)stream,
                       gamma, beta, running_mean, running_var, conv_bias, scale, shift, C, eps);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" size_t dynmm_act_bwd_bias_workspace_bytes(int N, int C) {
    if (N <= 0 || C <= 0) return 0;
    const int S = reduce_splits(N, C);
    return S > 1 ? sizeof(float) * (size_t)S * C : 0;
}

extern "C" int dynmm_act_bwd_bias(const float* g, const float* y, float* g_out, float* dbias, float* workspace,
                                  int N, int C, int HW, int act, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!g || N <= 0 || C <= 0 || HW <= 0) return DYNMM_EINVAL;
    if (act != DYNMM_ACT_NONE && !y) return DYNMM_EINVAL;
    if (!g_out && !dbias) return DYNMM_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const int S = reduce_splits(N, C);
    if (dbias && S > 1 && !workspace) return DYNMM_EWORKSPACE;
    float* part = (dbias && S > 1) ? workspace : dbias;
    dim3 grid(C, S);
    if (can_vec4(HW, {g, y, g_out}))
        hipLaunchKernelGGL(act_bwd_bias_kernel<4>, grid, dim3(256), 0, st, g, y, g_out, part, N, C, HW, act);
    else
        hipLaunchKernelGGL(act_bwd_bias_kernel<1>, grid, dim3(256), 0, st, g, y, g_out, part, N, C, HW, act);
    DYNMM_LAUNCH_CHECK();
    if (dbias && S > 1) {
        launch_reduce_slabs(part, dbias, C, S, st);
        DYNMM_LAUNCH_CHECK();
    }
    return DYNMM_OK;
}
