// Callers immediately either side of the hot path: the class-weighted 2-D cross
// entropy that reads the 49 MB/img logits (one fused log-softmax + NLL pass, fp64 accumulation of the
// two scalars) and a flat fused SGD-Nesterov update.  Both are pure HBM streaming.
#include "common.h"
#include "vec.h"

namespace dynmm {

constexpr int kMaxClasses = 64;

// One pixel per thread, its C logits held in registers (CMAX compile-time bound): every logit is read
// from memory exactly once and all C loads of a pixel are in flight together.  (The first version re-read
// the channel column once per pass — max, sum, output — from L2: 850 / 1625 us fwd / bwd on the 40x480x640
// logits of a batch of 32, against ~350 / ~700 us at the HBM roofline.)
template <int CMAX>
__device__ __forceinline__ void load_logits(const float* __restrict__ xn, int C, int HW, int p, float (&v)[CMAX]) {
#pragma unroll
    for (int c = 0; c < CMAX; ++c) v[c] = c < C ? xn[(size_t)c * HW + p] : -INFINITY;
}

template <int CMAX>
__global__ void __launch_bounds__(256) ce2d_fwd_kernel(const float* __restrict__ x,
                                                       const unsigned char* __restrict__ target,
                                                       const float* __restrict__ cw,
                                                       double* __restrict__ out2, int C, int HW, int dual) {
    __shared__ float red[4];
    const int n = blockIdx.y;
    const float* xn = x + (size_t)n * C * HW;
    const unsigned char* tn = target + (size_t)n * HW;
    float ls = 0.f, ws = 0.f, us = 0.f, cnt = 0.f;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
        const int t = (int)tn[p] - 1;
        if (t < 0 || t >= C) continue;   // void (ignore_index = -1 after the shift)
        float v[CMAX];
        load_logits<CMAX>(xn, C, HW, p, v);
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) mx = fmaxf(mx, v[c]);
        float den = 0.f, xt = 0.f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            den += expf(v[c] - mx);            // exp(-inf) = 0 for the padding slots
            xt = c == t ? v[c] : xt;
        }
        const float lse = logf(den) + mx;
        const float w = cw[t];
        ls += w * (lse - xt);
        ws += w;
        us += lse - xt;
        cnt += 1.f;
    }
    const float tl = block_reduce_sum_256<float>(ls, red);
    const float tw = block_reduce_sum_256<float>(ws, red);
    if (threadIdx.x == 0) {
        atomicAdd(&out2[0], (double)tl);
        atomicAdd(&out2[1], (double)tw);
    }
    if (dual) {   // validation: the unweighted sum and the non-void pixel count from the same pass (src/utils.py:77-97)
        const float tu = block_reduce_sum_256<float>(us, red);
        const float tc = block_reduce_sum_256<float>(cnt, red);   // <= 256 * ceil(HW / (256 * gridDim.x)): exact in fp32
        if (threadIdx.x == 0) {
            atomicAdd(&out2[2], (double)tu);
            atomicAdd(&out2[3], (double)tc);
        }
    }
}

template <int CMAX>
__global__ void __launch_bounds__(256) ce2d_bwd_kernel(const float* __restrict__ x,
                                                       const unsigned char* __restrict__ target,
                                                       const float* __restrict__ cw,
                                                       const float* __restrict__ gscale,
                                                       float* __restrict__ dx, int C, int HW) {
    const int n = blockIdx.y;
    const float* xn = x + (size_t)n * C * HW;
    float* dn = dx + (size_t)n * C * HW;
    const unsigned char* tn = target + (size_t)n * HW;
    const float gs = gscale[0];
    for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
        const int t = (int)tn[p] - 1;
        if (t < 0 || t >= C) {
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (c < C) dn[(size_t)c * HW + p] = 0.f;
            continue;
        }
        float v[CMAX];
        load_logits<CMAX>(xn, C, HW, p, v);
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) mx = fmaxf(mx, v[c]);
        float den = 0.f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            v[c] = expf(v[c] - mx);
            den += v[c];
        }
        const float k = cw[t] * gs;
        const float inv = 1.f / den;
#pragma unroll
        for (int c = 0; c < CMAX; ++c)
            if (c < C) dn[(size_t)c * HW + p] = k * (v[c] * inv - (c == t ? 1.f : 0.f));
    }
}

// train.py:313-321 in one tiny launch: the S per-scale losses loss_s = sum_s / wsum_s (from the fp64 accumulators
// of ce2d_fwd), total = sum_s loss_s + ratio * max(0, flop_loss - budget), and the seeds of the backward pass:
// gscale_s = d total / d (sum_s) = 1 / wsum_s for ce2d_bwd, d_flop = ratio * [flop_loss > budget].
__global__ void loss_head_kernel(const double* __restrict__ acc, int S, const float* __restrict__ flop_loss,
                                 float ratio, float budget, float* __restrict__ losses, float* __restrict__ total,
                                 float* __restrict__ gscale, float* __restrict__ d_flop) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float tot = 0.f;
    for (int s_ = 0; s_ < S; ++s_) {
        const float l = (float)(acc[2 * s_] / acc[2 * s_ + 1]);
        losses[s_] = l;
        gscale[s_] = (float)(1.0 / acc[2 * s_ + 1]);
        tot += l;                                  // left to right, like sum(losses) on the host
    }
    float df = 0.f;
    if (flop_loss && ratio > 0.f) {
        const float ex = flop_loss[0] - budget;
        if (ex > 0.f) { tot += ratio * ex; df = ratio; }
    }
    total[0] = tot;
    if (d_flop) d_flop[0] = df;
}

// Flat fused optimizer updates (train.py:554-579).  One kernel covers an element range [lo, hi) of the flat
// parameter / gradient / state buffers (16-byte aligned bases): aligned groups of 4 go through dwordx4
// accesses, the ragged edges of the range element by element.  Hyper-parameters that the training driver
// changes between steps (learning rate, momentum / beta1 under OneCycleLR's cycle_momentum) live in a small
// DEVICE array, so a step captured in a hipGraph sees the new values without re-capture.
// NaN guard (train.py:334-335 checks the loss on the host every step): when `loss` is given and not finite
// the update is skipped and *nan_flag receives 1 + the device step counter (first offender wins).
struct OptRange {
    size_t lo, hi;
};

template <typename F>
__device__ __forceinline__ void for_range4(size_t lo, size_t hi, F&& f) {
    // f(i, count): count == 4 for an aligned full group, else 1
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
    const size_t lo4 = (lo + 3) & ~(size_t)3, hi4 = hi & ~(size_t)3;
    if (lo4 >= hi4) {
        for (size_t k = lo + tid; k < hi; k += nth) f(k, 1);
        return;
    }
    for (size_t i = lo4 + tid * 4; i < hi4; i += nth * 4) f(i, 4);
    if (tid < lo4 - lo) f(lo + tid, 1);
    if (tid < hi - hi4) f(hi4 + tid, 1);
}

__device__ __forceinline__ bool loss_bad(const float* loss, int* nan_flag, const int* step) {
    if (!loss) return false;
    const float l = loss[0];
    const bool bad = !(fabsf(l) <= 3.0e38f);        // NaN or +-inf
    if (bad && nan_flag && blockIdx.x == 0 && threadIdx.x == 0) atomicCAS(nan_flag, 0, 1 + (step ? step[0] : 0));
    return bad;
}

__global__ void __launch_bounds__(256) sgd_nesterov_kernel(float* __restrict__ p,
                                                           const float* __restrict__ g,
                                                           float* __restrict__ buf, OptRange r,
                                                           const float* __restrict__ hyper,
                                                           float wd, float gscale,
                                                           const float* __restrict__ loss, int* nan_flag,
                                                           const int* __restrict__ step) {
    if (loss_bad(loss, nan_flag, step)) return;
    const float l = hyper[0], momentum = hyper[1];
    for_range4(r.lo, r.hi, [&](size_t i, int cnt) {
        if (cnt == 4) {
            float pv[4], gv[4], bv[4];
            vload<4>(p + i, pv);
            vload<4>(g + i, gv);
            vload<4>(buf + i, bv);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = gv[j] * gscale + wd * pv[j];
                bv[j] = momentum * bv[j] + d;
                pv[j] -= l * (d + momentum * bv[j]);
            }
            vstore<4>(p + i, pv);
            vstore<4>(buf + i, bv);
        } else {
            const float d = g[i] * gscale + wd * p[i];
            buf[i] = momentum * buf[i] + d;
            p[i] -= l * (d + momentum * buf[i]);
        }
    });
}

// torch.optim.Adam (L2 weight decay folded into the gradient, amsgrad off), train.py:564-570:
//   g += wd*p;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;
//   p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps),   t = step[0] (already incremented by dynmm_opt_tick)
// decoupled != 0: torch.optim.AdamW (Supervised_Learning.py:91 via affect_dyn.py:216): p *= 1 - lr*wd first, the
// gradient is used as is.  gscale_dev (optional): device scalar multiplied into the gradient (the clip coefficient
// of dynmm_clip_grad_norm).
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, OptRange r,
                                                   const float* __restrict__ hyper, const int* __restrict__ step,
                                                   float wd, float gscale, const float* __restrict__ loss,
                                                   int* nan_flag, int decoupled, const float* __restrict__ gscale_dev) {
    if (loss_bad(loss, nan_flag, step)) return;
    if (gscale_dev) gscale *= gscale_dev[0];
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3];
    const double t = (double)step[0];
    const float bc2s = (float)sqrt(1.0 - pow((double)b2, t));
    const float step_size = (float)((double)lr / (1.0 - pow((double)b1, t)));
    const float decay = decoupled ? 1.f - lr * wd : 1.f;
    const float l2 = decoupled ? 0.f : wd;
    auto upd = [&](float& pv, float gv, float& mv, float& vv) {
        pv *= decay;
        const float d = gv * gscale + l2 * pv;
        mv = b1 * mv + (1.f - b1) * d;
        vv = b2 * vv + (1.f - b2) * d * d;
        pv -= step_size * (mv / (sqrtf(vv) / bc2s + eps));       // torch: denom = sqrt(v)/sqrt(bc2) + eps
    };
    for_range4(r.lo, r.hi, [&](size_t i, int cnt) {
        if (cnt == 4) {
            float pv[4], gv[4], mv[4], vv[4];
            vload<4>(p + i, pv);
            vload<4>(g + i, gv);
            vload<4>(m + i, mv);
            vload<4>(v + i, vv);
#pragma unroll
            for (int j = 0; j < 4; ++j) upd(pv[j], gv[j], mv[j], vv[j]);
            vstore<4>(p + i, pv);
            vstore<4>(m + i, mv);
            vstore<4>(v + i, vv);
        } else {
            upd(p[i], g[i], m[i], v[i]);
        }
    });
}

__global__ void opt_tick_kernel(int* step) { step[0] += 1; }

// eval.py:117-141 as ONE pass: bilinear resize (align_corners=False) of the 40-channel logits to the
// label resolution, arg-max over classes, void mask (label 0), confusion-matrix increment
// cm[(label-1)*C + pred] += 1 (int64).  The resized logits and the arg-max map are never materialised.
__global__ void __launch_bounds__(256) eval_confusion_kernel(const float* __restrict__ x,
                                                             const unsigned char* __restrict__ label,
                                                             unsigned long long* __restrict__ cm,
                                                             int C, int H, int W, int Ho, int Wo) {
    const int n = blockIdx.y;
    const float* xn = x + (size_t)n * C * H * W;
    const unsigned char* ln = label + (size_t)n * Ho * Wo;
    const float sh = (float)H / (float)Ho, sw = (float)W / (float)Wo;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < Ho * Wo; p += gridDim.x * 256) {
        const int lab = (int)ln[p] - 1;
        if (lab < 0 || lab >= C) continue;
        const int oh = p / Wo, ow = p - oh * Wo;
        float fy = sh * ((float)oh + 0.5f) - 0.5f, fx = sw * ((float)ow + 0.5f) - 0.5f;
        fy = fy < 0.f ? 0.f : fy;
        fx = fx < 0.f ? 0.f : fx;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
        const float ly1 = fy - (float)y0, lx1 = fx - (float)x0;
        const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
        float best = -INFINITY;
        int arg = 0;
        for (int c = 0; c < C; ++c) {
            const float* pc = xn + (size_t)c * H * W;
            const float v = ly0 * (lx0 * pc[y0 * W + x0] + lx1 * pc[y0 * W + x1]) +
                            ly1 * (lx0 * pc[y1 * W + x0] + lx1 * pc[y1 * W + x1]);
            if (v > best) { best = v; arg = c; }       // first maximum, as torch.argmax
        }
        atomicAdd(&cm[(size_t)lab * C + arg], 1ull);
    }
}

}  // namespace dynmm

using namespace dynmm;

extern "C" int dynmm_loss_head(const double* acc, int S, const float* flop_loss, float ratio, float budget,
                               float* losses, float* total, float* gscale, float* d_flop, void* stream) {
    (void)hipGetLastError();
    if (!acc || S <= 0 || !losses || !total || !gscale) return DYNMM_EINVAL;
    hipLaunchKernelGGL(loss_head_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, acc, S, flop_loss, ratio, budget,
                       losses, total, gscale, d_flop);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_ce2d_fwd(const float* x, const unsigned char* target, const float* cw,
                              double* loss_sum_wsum, int N, int C, int HW, int acc_is_zero, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!x || !target || !cw || !loss_sum_wsum || N <= 0 || C <= 0 || C > kMaxClasses || HW <= 0)
        return DYNMM_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (!acc_is_zero) DYNMM_HIP_TRY(hipMemsetAsync(loss_sum_wsum, 0, 2 * sizeof(double), st));
    int bx = ceil_div(HW, 256);
    if (bx > 256) bx = 256;
    if (C <= 40)
        hipLaunchKernelGGL(ce2d_fwd_kernel<40>, dim3(bx, N), dim3(256), 0, st, x, target, cw, loss_sum_wsum, C, HW, 0);
    else
        hipLaunchKernelGGL(ce2d_fwd_kernel<kMaxClasses>, dim3(bx, N), dim3(256), 0, st, x, target, cw, loss_sum_wsum, C, HW, 0);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

// validate()'s two losses (train.py:104-115, src/utils.py:53-97) from ONE pass over the logits: acc4 +=
// (sum w[t]*CE, sum w[t], sum CE, #non-void pixels).  Always accumulates (the caller zeroes acc4 per validation run).
extern "C" int dynmm_ce2d_valid(const float* x, const unsigned char* target, const float* cw,
                                double* acc4, int N, int C, int HW, void* stream) {
    (void)hipGetLastError();
    if (!x || !target || !cw || !acc4 || N <= 0 || C <= 0 || C > kMaxClasses || HW <= 0)
        return DYNMM_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    int bx = ceil_div(HW, 256);
    if (bx > 256) bx = 256;
    if (C <= 40)
        hipLaunchKernelGGL(ce2d_fwd_kernel<40>, dim3(bx, N), dim3(256), 0, st, x, target, cw, acc4, C, HW, 1);
    else
        hipLaunchKernelGGL(ce2d_fwd_kernel<kMaxClasses>, dim3(bx, N), dim3(256), 0, st, x, target, cw, acc4, C, HW, 1);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_ce2d_bwd(const float* x, const unsigned char* target, const float* cw,
                              const float* gscale, float* dx, int N, int C, int HW, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!x || !target || !cw || !gscale || !dx || N <= 0 || C <= 0 || C > kMaxClasses || HW <= 0)
        return DYNMM_EINVAL;
    int bx = ceil_div(HW, 256);
    if (bx > 256) bx = 256;
    if (C <= 40)
        hipLaunchKernelGGL(ce2d_bwd_kernel<40>, dim3(bx, N), dim3(256), 0, (hipStream_t)stream, x, target, cw,
                           gscale, dx, C, HW);
    else
        hipLaunchKernelGGL(ce2d_bwd_kernel<kMaxClasses>, dim3(bx, N), dim3(256), 0, (hipStream_t)stream, x, target,
                           cw, gscale, dx, C, HW);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

static unsigned opt_blocks(size_t n) {
    size_t blocks = ceil_div_sz(n, 1024);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    return (unsigned)blocks;
}

extern "C" int dynmm_opt_tick(int* step, void* stream) {
    (void)hipGetLastError();
    if (!step) return DYNMM_EINVAL;
    hipLaunchKernelGGL(opt_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_sgd_nesterov(float* p, const float* g, float* buf, size_t lo, size_t hi, const float* hyper,
                                  float weight_decay, float grad_scale, const float* loss, int* nan_flag,
                                  const int* step, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!p || !g || !buf || !hyper || hi <= lo) return DYNMM_EINVAL;
    if (!aligned16(p) || !aligned16(g) || !aligned16(buf)) return DYNMM_EUNSUPPORTED;
    hipLaunchKernelGGL(sgd_nesterov_kernel, dim3(opt_blocks(hi - lo)), dim3(256), 0, (hipStream_t)stream,
                       p, g, buf, OptRange{lo, hi}, hyper, weight_decay, grad_scale, loss, nan_flag, step);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_adam(float* p, const float* g, float* m, float* v, size_t lo, size_t hi, const float* hyper,
                          const int* step, float weight_decay, float grad_scale, const float* loss, int* nan_flag,
                          int decoupled, const float* grad_scale_dev, void* stream) {
    (void)hipGetLastError();
    if (!p || !g || !m || !v || !hyper || !step || hi <= lo) return DYNMM_EINVAL;
    if (!aligned16(p) || !aligned16(g) || !aligned16(m) || !aligned16(v)) return DYNMM_EUNSUPPORTED;
    hipLaunchKernelGGL(adam_kernel, dim3(opt_blocks(hi - lo)), dim3(256), 0, (hipStream_t)stream,
                       p, g, m, v, OptRange{lo, hi}, hyper, step, weight_decay, grad_scale, loss, nan_flag, decoupled,
                       grad_scale_dev);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_eval_confusion(const float* logits, const unsigned char* label, long long* cm, int N,
                                    int C, int H, int W, int Ho, int Wo, void* stream) {
    (void)hipGetLastError();
    if (!logits || !label || !cm || N <= 0 || C <= 0 || C > kMaxClasses || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0)
        return DYNMM_EINVAL;
    int bx = ceil_div(Ho * Wo, 256);
    if (bx > 512) bx = 512;
    hipLaunchKernelGGL(eval_confusion_kernel, dim3(bx, N), dim3(256), 0, (hipStream_t)stream, logits, label,
                       reinterpret_cast<unsigned long long*>(cm), C, H, W, Ho, Wo);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}
