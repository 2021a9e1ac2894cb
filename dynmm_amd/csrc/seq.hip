// Modality-level DynMM (ModalityDynMM/affect/affect_dyn.py): the pieces of the sequence experts that are not
// GEMMs.  Every Linear / Conv1d(k=1) of those experts is a 1x1 convolution over tokens and runs on the
// implicit-GEMM MFMA kernels of conv_igemm.hip with activations laid out [B, D, T] (= NCHW with H = 1, W = T,
// exactly the layout the reference produces with x.permute([0, 2, 1]) in front of its Conv1d).  This file adds
//   * LayerNorm over the channel axis D of [B, D, T] (post-norm TransformerEncoderLayer), forward + backward,
//   * multi-head self-attention for short sequences (T <= 64): one wave per (sample, head), Q K V and the
//     T x T probabilities stay in LDS, forward + backward,
//   * the mixture head: DiffSoftmax gate over K experts (affect_dyn.py:18-28), blend, L1 loss, gate regulariser
//     and the backward seeds (affect_dyn.py:152-165, Supervised_Learning.py:135-136), one launch,
//   * global gradient-norm clipping (Supervised_Learning.py:143) as a deterministic two-stage reduction.
// All of it is latency-bound bookkeeping around ~0.3 GMAC/sample of feed-forward GEMMs.
#include "common.h"

namespace dynmm {

constexpr int kSeqMaxT = 64;
constexpr int kSeqMaxDh = 32;

// ---------------------------------------------------------------------------------------------------------------
// Dropout (nn.TransformerEncoderLayer trains with p = 0.1 at four places: the attention probabilities, the attention
// block's output, the feed-forward hidden layer and the feed-forward output).  An element survives with probability
// 1 - p and is scaled by 1/(1 - p).  The decision for element `idx` of a site is a pure function of
// (seed, offset + *step, idx) through Philox-4x32-10, so the backward pass regenerates it instead of storing masks and a
// captured hipGraph draws new masks at every replay (`step` is a device counter the training step advances).
// `mask` (tests): explicit keep flags, one byte per element, instead of the generator.
// ---------------------------------------------------------------------------------------------------------------
struct DropSpec {
    const unsigned char* mask;
    const unsigned long long* step;
    unsigned long long seed, offset;
    float p;
};

__device__ __forceinline__ void seq_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                  uint32_t k1, uint32_t out[4]) {
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

// 0 (dropped) or 1/(1-p) (kept); 1 when the site has no dropout
struct DropState {
    const unsigned char* mask;
    unsigned long long off;
    uint32_t k0, k1;
    float p, inv;
    __device__ __forceinline__ explicit DropState(const DropSpec& d)
        : mask(d.mask), off(d.offset + (d.step ? *d.step : 0ull)), k0((uint32_t)d.seed), k1((uint32_t)(d.seed >> 32)),
          p(d.p), inv(d.p > 0.f ? 1.f / (1.f - d.p) : 1.f) {}
    __device__ __forceinline__ float operator()(size_t idx) const {
        if (!(p > 0.f)) return 1.f;
        if (mask) return mask[idx] ? inv : 0.f;
        uint32_t r[4];
        seq_philox4x32_10((uint32_t)idx, (uint32_t)((unsigned long long)idx >> 32), (uint32_t)off, (uint32_t)(off >> 32), k0, k1, r);
        const float u = (float)(r[0] >> 8) * (1.f / 16777216.f);        // [0, 1)
        return u >= p ? inv : 0.f;
    }
};

__global__ void __launch_bounds__(256) dropout_kernel(const float* __restrict__ x, float* __restrict__ y, size_t n,
                                                      const DropSpec spec) {
    const DropState ds(spec);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = x[i] * ds(i);
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm over D for every token (b, t) of x[B, D, T]; lanes walk consecutive t (coalesced), D is strided.
// ---------------------------------------------------------------------------------------------------------------
// `res` (optional): the layer normalises x + res (the residual connection of the encoder layer) without a
// separate add pass.  Workgroup = 16 tokens x 16 channel groups: lanes of a 16-lane row read consecutive t, the 16
// groups split the channel loop (D = 10 ... 120: a thread touches <= 8 channels per pass) and meet in LDS.
constexpr int kLnTok = 16, kLnGrp = 16;

__device__ __forceinline__ float ln_group_sum(float v, float (*sh)[kLnTok], int tl, int grp) {
    sh[grp][tl] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < kLnGrp; ++k) s += sh[k][tl];       // fixed order: deterministic
    __syncthreads();
    return s;
}

// `nparts` > 1 / `xbias`: x arrives as partial sums (the hidden-unit split of ffn_kernel) plus a per-channel bias; the sum is
// formed once, in slab order, written to `xsum` (the backward's `x`) and used from there.
__global__ void __launch_bounds__(256) ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int B, int D,
                                                     int T, float eps, const DropSpec spec, int nparts, size_t pstride,
                                                     const float* __restrict__ xbias, float* __restrict__ xsum) {
    __shared__ float sh[kLnGrp][kLnTok];
    const DropState ds(spec);
    const int tl = threadIdx.x & (kLnTok - 1), grp = threadIdx.x / kLnTok;
    const int tok = blockIdx.x * kLnTok + tl;
    const bool ok = tok < B * T;
    const int b = ok ? tok / T : 0, t = ok ? tok - b * T : 0;
    const size_t base = (size_t)b * D * T + t;
    const float* xs = xsum ? xsum : x;
    float s = 0.f;
    if (ok)
        for (int c = grp; c < D; c += kLnGrp) {
            const size_t i = base + (size_t)c * T;
            float xv = x[i];
            if (xsum) {
                for (int k = 1; k < nparts; ++k) xv += x[(size_t)k * pstride + i];
                if (xbias) xv += xbias[c];
                xsum[i] = xv;
            }
            s += xv * ds(i) + (res ? res[i] : 0.f);
        }
    const float mu = ln_group_sum(s, sh, tl, grp) / (float)D;
    float v = 0.f;
    if (ok)
        for (int c = grp; c < D; c += kLnGrp) {
            const size_t i = base + (size_t)c * T;
            const float d = xs[i] * ds(i) + (res ? res[i] : 0.f) - mu;
            v += d * d;
        }
    const float rs = rsqrtf(ln_group_sum(v, sh, tl, grp) / (float)D + eps);
    if (!ok) return;
    if (grp == 0) {
        if (mean) mean[tok] = mu;
        if (rstd) rstd[tok] = rs;
    }
    for (int c = grp; c < D; c += kLnGrp) {
        const size_t i = base + (size_t)c * T;
        const float sv = xs[i] * ds(i) + (res ? res[i] : 0.f);
        y[i] = (sv - mu) * rs * gamma[c] + beta[c];
    }
}

// dx = rstd * (g*gamma - mean_c(g*gamma) - xhat * mean_c(g*gamma*xhat))
__global__ void __launch_bounds__(256) ln_bwd_dx_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                        const float* __restrict__ res,
                                                        const float* __restrict__ gamma, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, float* __restrict__ dx,
                                                        float* __restrict__ dres, int B, int D, int T,
                                                        const DropSpec spec) {
    __shared__ float sh[kLnGrp][kLnTok];
    const DropState ds(spec);
    const int tl = threadIdx.x & (kLnTok - 1), grp = threadIdx.x / kLnTok;
    const int tok = blockIdx.x * kLnTok + tl;
    const bool ok = tok < B * T;
    const int b = ok ? tok / T : 0, t = ok ? tok - b * T : 0;
    const size_t base = (size_t)b * D * T + t;
    const float mu = ok ? mean[tok] : 0.f, rs = ok ? rstd[tok] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    if (ok)
        for (int c = grp; c < D; c += kLnGrp) {
            const size_t i = base + (size_t)c * T;
            const float gg = g[i] * gamma[c];
            const float xh = (x[i] * ds(i) + (res ? res[i] : 0.f) - mu) * rs;
            s1 += gg;
            s2 += gg * xh;
        }
    s1 = ln_group_sum(s1, sh, tl, grp) / (float)D;
    s2 = ln_group_sum(s2, sh, tl, grp) / (float)D;
    if (!ok) return;
    // d(x*keep + res): the residual branch receives it as is (dres), x through its keep factor (dx)
    for (int c = grp; c < D; c += kLnGrp) {
        const size_t i = base + (size_t)c * T;
        const float gg = g[i] * gamma[c];
        const float k = ds(i);
        const float xh = (x[i] * k + (res ? res[i] : 0.f) - mu) * rs;
        const float dv = rs * (gg - s1 - xh * s2);
        if (dres) dres[i] = dv;
        if (dx) dx[i] = dv * k;
    }
}

// dgamma[c] = sum_tok g * xhat ; dbeta[c] = sum_tok g : one workgroup per channel, fixed summation order
__global__ void __launch_bounds__(256) ln_bwd_param_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                           const float* __restrict__ res,
                                                           const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, float* __restrict__ dgamma,
                                                           float* __restrict__ dbeta, int B, int D, int T,
                                                           const DropSpec spec) {
    __shared__ float red[4];
    const DropState ds(spec);
    const int c = blockIdx.x;
    float a = 0.f, bsum = 0.f;
    for (int tok = threadIdx.x; tok < B * T; tok += 256) {
        const int b = tok / T, t = tok - b * T;
        const size_t i = ((size_t)b * D + c) * T + t;
        const float gv = g[i];
        a += gv * (x[i] * ds(i) + (res ? res[i] : 0.f) - mean[tok]) * rstd[tok];
        bsum += gv;
    }
    const float ta = block_reduce_sum_256<float>(a, red);
    const float tb = block_reduce_sum_256<float>(bsum, red);
    if (threadIdx.x == 0) {
        dgamma[c] = ta;
        dbeta[c] = tb;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Multi-head self-attention, qkv [B, 3D, T] (q | k | v along channels, as nn.MultiheadAttention's in_proj), heads H,
// dh = D / H.  One 64-lane wave per (b, h); lane i owns query i (forward) / query i and key i (backward).
//   P = softmax_j( (q_i . k_j) / sqrt(dh) ),  out[c][i] = sum_j P[i][j] v[c][j]
// ---------------------------------------------------------------------------------------------------------------
// Lane i keeps its query (forward) / its query-side and key-side accumulators (backward) in REGISTERS (DH = compile-time
// bound on the head dimension, dh <= DH), K / V / Q / dOut tiles sit in LDS and are read as wave-wide broadcasts
// (every lane the same address) or conflict-free rows/columns of the [T][T+1] probability tiles: one LDS read per
// FMA instead of two, and no dependent-latency chain per lane (the first version, with everything in LDS at ~2.5
// waves per CU, was bound by LDS latency: 36 / 88 us forward / backward per launch for B = 128).
// EXACT: dh == DH (no per-channel predicates inside the unrolled loops — with them every unrolled iteration is a
// scalar branch and the kernel runs 3x slower); the non-exact instantiation serves any dh <= DH.
template <int DH, bool EXACT>
__global__ void __launch_bounds__(64) mha_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                     float* __restrict__ probs, int D, int T, int H, const DropSpec spec) {
    __shared__ float ks[DH][kSeqMaxT], vs[DH][kSeqMaxT];
    __shared__ float ps[kSeqMaxT][kSeqMaxT + 1];
    const DropState drop(spec);
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int dh = D / H;
    const int i = threadIdx.x;
    const bool act = i < T;
    const float* base = qkv + (size_t)b * 3 * D * T;
    const float scale = rsqrtf((float)dh);
    float q[DH], o[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) {
        q[c] = 0.f;
        o[c] = 0.f;
        if ((EXACT || c < dh) && act) {
            q[c] = base[(size_t)(h * dh + c) * T + i] * scale;          // torch scales q before q @ k^T
            ks[c][i] = base[(size_t)(D + h * dh + c) * T + i];
            vs[c][i] = base[(size_t)(2 * D + h * dh + c) * T + i];
        }
    }
    __syncthreads();
    if (!act) return;
    float mx = -INFINITY;
    for (int j = 0; j < T; ++j) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < DH; ++c)
            if (EXACT || c < dh) s += q[c] * ks[c][j];
        ps[i][j] = s;
        mx = fmaxf(mx, s);
    }
    float den = 0.f;
    const size_t prow = ((size_t)blockIdx.x * T + i) * T;      // probs (and their keep flags) are [B*H][T][T]
    for (int j = 0; j < T; ++j) {
        const float e = expf(ps[i][j] - mx);
        ps[i][j] = e;
        den += e;
        const float ek = e * drop(prow + j);                     // dropout acts on the normalised probabilities: linear in e
#pragma unroll
        for (int c = 0; c < DH; ++c)
            if (EXACT || c < dh) o[c] += ek * vs[c][j];
    }
    const float inv = 1.f / den;
    float* pg = probs + prow;
    for (int j = 0; j < T; ++j) pg[j] = ps[i][j] * inv;        // saved for the backward: the probabilities BEFORE dropout
    float* ob = out + (size_t)b * D * T;
#pragma unroll
    for (int c = 0; c < DH; ++c)
        if (EXACT || c < dh) ob[(size_t)(h * dh + c) * T + i] = o[c] * inv;
}

template <int DH, bool EXACT>
__global__ void __launch_bounds__(64) mha_bwd_kernel(const float* __restrict__ g, const float* __restrict__ qkv,
                                                     const float* __restrict__ probs, float* __restrict__ dqkv, int D,
                                                     int T, int H, const DropSpec spec) {
    const DropState drop(spec);
    __shared__ float qs[DH][kSeqMaxT], ks[DH][kSeqMaxT], vs[DH][kSeqMaxT], gs[DH][kSeqMaxT];
    __shared__ float ps[kSeqMaxT][kSeqMaxT + 1], ds[kSeqMaxT][kSeqMaxT + 1];
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int dh = D / H;
    const int i = threadIdx.x;
    const bool act = i < T;
    const float* base = qkv + (size_t)b * 3 * D * T;
    const float* gb = g + (size_t)b * D * T;
    float gq[DH];                                   // dOut column of query i
#pragma unroll
    for (int c = 0; c < DH; ++c) {
        gq[c] = 0.f;
        if ((EXACT || c < dh) && act) {
            qs[c][i] = base[(size_t)(h * dh + c) * T + i];
            ks[c][i] = base[(size_t)(D + h * dh + c) * T + i];
            vs[c][i] = base[(size_t)(2 * D + h * dh + c) * T + i];
            gq[c] = gb[(size_t)(h * dh + c) * T + i];
            gs[c][i] = gq[c];
        }
    }
    if (act) {
        const float* pg = probs + ((size_t)blockIdx.x * T + i) * T;
        for (int j = 0; j < T; ++j) ps[i][j] = pg[j];
    }
    __syncthreads();
    if (act) {
        // P' = P * keep (what multiplied V);  dP'[i][j] = sum_c g[c][i] v[c][j];  dP = dP' * keep;
        // dS = P * (dP - sum_j P dP).  Row i then keeps P' in ps (dV needs it), P is not used again.
        const size_t prow = ((size_t)blockIdx.x * T + i) * T;
        float dot = 0.f;
        for (int j = 0; j < T; ++j) {
            float sacc = 0.f;
#pragma unroll
            for (int c = 0; c < DH; ++c)
                if (EXACT || c < dh) sacc += gq[c] * vs[c][j];
            sacc *= drop(prow + j);
            ds[i][j] = sacc;
            dot += ps[i][j] * sacc;
        }
        for (int j = 0; j < T; ++j) {
            ds[i][j] = ps[i][j] * (ds[i][j] - dot);
            ps[i][j] *= drop(prow + j);
        }
    }
    __syncthreads();
    if (!act) return;
    const float scale = rsqrtf((float)dh);
    float dq[DH], dk[DH], dv[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) dq[c] = dk[c] = dv[c] = 0.f;
    for (int j = 0; j < T; ++j) {
        const float dsr = ds[i][j], dsc = ds[j][i], pc = ps[j][i];     // row (query i), columns (key i)
#pragma unroll
        for (int c = 0; c < DH; ++c)
            if (EXACT || c < dh) {
                dq[c] += dsr * ks[c][j];
                dk[c] += dsc * qs[c][j];
                dv[c] += pc * gs[c][j];
            }
    }
    float* db = dqkv + (size_t)b * 3 * D * T;
#pragma unroll
    for (int c = 0; c < DH; ++c)
        if (EXACT || c < dh) {
            db[(size_t)(h * dh + c) * T + i] = dq[c] * scale;
            db[(size_t)(D + h * dh + c) * T + i] = dk[c] * scale;
            db[(size_t)(2 * D + h * dh + c) * T + i] = dv[c];
        }
}

// ---------------------------------------------------------------------------------------------------------------
// Mixture head + loss + backward seeds (one workgroup; B <= a few thousand).
//   w = DiffSoftmax(logits / temp, hard)                                   affect_dyn.py:18-28,153
//   out[b] = sum_k w[b,k] * pred_k[b]          (infer_mode 0)              affect_dyn.py:164
//   aux = mean_b w[b, K-1]                                                  affect_dyn.py:165
//   loss1 = mean_b |out[b] - y[b]| ; total = loss1 + reg * aux              Supervised_Learning.py:135-136
// Seeds of d total: d_pred_k[b] = w[b,k] * sgn/B ; d_logits through the soft path (straight-through).
// ---------------------------------------------------------------------------------------------------------------
struct MoePreds { const float* p[4]; };
struct MoeGrads { float* p[4]; };

__global__ void __launch_bounds__(256) moe_head_kernel(const float* __restrict__ logits, MoePreds P, int K,
                                                       const float* __restrict__ target, float temp, int hard, float reg,
                                                       float* __restrict__ out, float* __restrict__ weight,
                                                       float* __restrict__ scalars /* loss1, aux, total */,
                                                       MoeGrads dP, float* __restrict__ d_logits, int B) {
    __shared__ float red[4];
    float l1 = 0.f, aux = 0.f;
    const float invB = 1.f / (float)B;
    for (int b = threadIdx.x; b < B; b += 256) {
        float z[4], w[4];
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) { z[k] = logits[(size_t)b * K + k] / temp; mx = fmaxf(mx, z[k]); }
        float den = 0.f;
        for (int k = 0; k < K; ++k) { z[k] = expf(z[k] - mx); den += z[k]; }
        int arg = 0;
        float best = -1.f;
        for (int k = 0; k < K; ++k) {
            z[k] /= den;
            if (z[k] > best) { best = z[k]; arg = k; }
        }
        float o = 0.f;
        for (int k = 0; k < K; ++k) {
            w[k] = hard ? ((k == arg ? 1.f : 0.f) - z[k]) + z[k] : z[k];
            weight[(size_t)b * K + k] = w[k];
            o += w[k] * P.p[k][b];
        }
        out[b] = o;
        aux += w[K - 1];
        if (target) {
            const float diff = o - target[b];
            l1 += fabsf(diff);
            const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
            const float go = sgn * invB;
            float dw[4], dot = 0.f;
            for (int k = 0; k < K; ++k) {
                if (dP.p[k]) dP.p[k][b] = w[k] * go;
                dw[k] = P.p[k][b] * go + (k == K - 1 ? reg * invB : 0.f);
                dot += z[k] * dw[k];
            }
            if (d_logits)
                for (int k = 0; k < K; ++k) d_logits[(size_t)b * K + k] = z[k] * (dw[k] - dot) / temp;
        }
    }
    const float t1 = block_reduce_sum_256<float>(l1, red);
    const float ta = block_reduce_sum_256<float>(aux, red);
    if (threadIdx.x == 0) {
        scalars[0] = t1 * invB;
        scalars[1] = ta * invB;
        scalars[2] = t1 * invB + reg * ta * invB;
    }
}

// backward of the blend for arbitrary upstream gradients (model used under plain autograd):
//   d_pred_k = w_k * d_out ; d_w_k = pred_k * d_out + [k == K-1] * d_aux / B ; d_logits through the soft path
__global__ void __launch_bounds__(256) moe_blend_bwd_kernel(const float* __restrict__ d_out, const float* __restrict__ d_aux,
                                                            const float* __restrict__ logits, MoePreds P, int K,
                                                            const float* __restrict__ weight, float temp, MoeGrads dP,
                                                            float* __restrict__ d_logits, int B) {
    const float da = d_aux ? d_aux[0] / (float)B : 0.f;
    for (int b = blockIdx.x * 256 + threadIdx.x; b < B; b += gridDim.x * 256) {
        float z[4];
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) { z[k] = logits[(size_t)b * K + k] / temp; mx = fmaxf(mx, z[k]); }
        float den = 0.f;
        for (int k = 0; k < K; ++k) { z[k] = expf(z[k] - mx); den += z[k]; }
        const float go = d_out ? d_out[b] : 0.f;
        float dw[4], dot = 0.f;
        for (int k = 0; k < K; ++k) {
            z[k] /= den;
            if (dP.p[k]) dP.p[k][b] = weight[(size_t)b * K + k] * go;
            dw[k] = P.p[k][b] * go + (k == K - 1 ? da : 0.f);
            dot += z[k] * dw[k];
        }
        for (int k = 0; k < K; ++k) d_logits[(size_t)b * K + k] = z[k] * (dw[k] - dot) / temp;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// sum of squares of a flat buffer -> clip coefficient min(1, max_norm / (norm + 1e-6)) (torch.nn.utils.clip_grad_norm_)
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) sumsq_partial_kernel(const float* __restrict__ x, size_t n,
                                                            double* __restrict__ part) {
    __shared__ double red[4];
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double v = (double)x[i];
        s += v * v;
    }
    const double t = block_reduce_sum_256<double>(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}

__global__ void clip_coef_kernel(const double* __restrict__ part, int nparts, float max_norm, float* __restrict__ out2) {
    if (threadIdx.x != 0) return;
    double s = 0.0;
    for (int i = 0; i < nparts; ++i) s += part[i];
    const double norm = sqrt(s);
    out2[0] = (float)norm;
    const double c = (double)max_norm / (norm + 1e-6);
    out2[1] = c < 1.0 ? (float)c : 1.f;
}

}  // namespace dynmm

using namespace dynmm;

#define ST ((hipStream_t)stream)

static DropSpec drop_spec(const dynmm_dropout* d) {
    DropSpec s{};
    if (d && d->p > 0.f) {
        s.mask = d->mask; s.step = d->step; s.seed = d->seed; s.offset = d->offset; s.p = d->p;
    }
    return s;
}
static bool drop_ok(const dynmm_dropout* d) { return !d || (d->p >= 0.f && d->p < 1.f); }

extern "C" int dynmm_dropout_apply(const float* x, float* y, size_t n, const dynmm_dropout* drop, void* stream) {
    (void)hipGetLastError();
    if (!x || !y || n == 0 || !drop_ok(drop)) return DYNMM_EINVAL;
    const size_t nb = (n + 255) / 256;
    hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, ST, x, y, n, drop_spec(drop));
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_layernorm_drop_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                                        float* mean, float* rstd, int B, int D, int T, float eps,
                                        const dynmm_dropout* drop, void* stream) {
    (void)hipGetLastError();
    if (!x || !gamma || !beta || !y || B <= 0 || D <= 0 || T <= 0 || !drop_ok(drop)) return DYNMM_EINVAL;
    hipLaunchKernelGGL(ln_fwd_kernel, dim3(ceil_div(B * T, kLnTok)), dim3(256), 0, ST, x, res, gamma, beta, y, mean, rstd,
                       B, D, T, eps, drop_spec(drop), 1, (size_t)0, (const float*)nullptr, (float*)nullptr);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_layernorm_parts_fwd(const float* parts, int nparts, const float* xbias, float* xsum, const float* res,
                                         const float* gamma, const float* beta, float* y, float* mean, float* rstd, int B,
                                         int D, int T, float eps, const dynmm_dropout* drop, void* stream) {
    (void)hipGetLastError();
    if (!parts || nparts <= 0 || !xsum || !gamma || !beta || !y || B <= 0 || D <= 0 || T <= 0 || !drop_ok(drop)) return DYNMM_EINVAL;
    hipLaunchKernelGGL(ln_fwd_kernel, dim3(ceil_div(B * T, kLnTok)), dim3(256), 0, ST, parts, res, gamma, beta, y, mean, rstd,
                       B, D, T, eps, drop_spec(drop), nparts, (size_t)B * D * T, xbias, xsum);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                                   float* mean, float* rstd, int B, int D, int T, float eps, void* stream) {
    return dynmm_layernorm_drop_fwd(x, res, gamma, beta, y, mean, rstd, B, D, T, eps, nullptr, stream);
}

extern "C" int dynmm_layernorm_drop_bwd(const float* g, const float* x, const float* res, const float* gamma,
                                        const float* mean, const float* rstd, float* dx, float* dres, float* dgamma,
                                        float* dbeta, int B, int D, int T, const dynmm_dropout* drop, void* stream) {
    (void)hipGetLastError();
    if (!g || !x || !gamma || !mean || !rstd || B <= 0 || D <= 0 || T <= 0 || !drop_ok(drop)) return DYNMM_EINVAL;
    if (dx || dres) {
        hipLaunchKernelGGL(ln_bwd_dx_kernel, dim3(ceil_div(B * T, kLnTok)), dim3(256), 0, ST, g, x, res, gamma, mean, rstd,
                           dx, dres, B, D, T, drop_spec(drop));
        DYNMM_LAUNCH_CHECK();
    }
    if (dgamma && dbeta) {
        hipLaunchKernelGGL(ln_bwd_param_kernel, dim3(D), dim3(256), 0, ST, g, x, res, mean, rstd, dgamma, dbeta, B, D, T,
                           drop_spec(drop));
        DYNMM_LAUNCH_CHECK();
    }
    return DYNMM_OK;
}

extern "C" int dynmm_layernorm_bwd(const float* g, const float* x, const float* res, const float* gamma,
                                   const float* mean, const float* rstd, float* dx, float* dgamma, float* dbeta, int B,
                                   int D, int T, void* stream) {
    return dynmm_layernorm_drop_bwd(g, x, res, gamma, mean, rstd, dx, nullptr, dgamma, dbeta, B, D, T, nullptr, stream);
}

extern "C" int dynmm_mha_drop_fwd(const float* qkv, float* out, float* probs, int B, int D, int T, int heads,
                                  const dynmm_dropout* drop, void* stream) {
    (void)hipGetLastError();
    if (!qkv || !out || !probs || B <= 0 || D <= 0 || T <= 0 || heads <= 0 || D % heads != 0 || !drop_ok(drop)) return DYNMM_EINVAL;
    const DropSpec spec = drop_spec(drop);
    if (T > kSeqMaxT || D / heads > kSeqMaxDh) return DYNMM_EUNSUPPORTED;
    const int dh = D / heads;
#define DYNMM_MHA_F(DH, EX) hipLaunchKernelGGL((mha_fwd_kernel<DH, EX>), dim3(B * heads), dim3(64), 0, ST, qkv, out, probs, D, T, heads, spec)
    if (dh == 24) DYNMM_MHA_F(24, true);
    else if (dh == 12) DYNMM_MHA_F(12, true);
    else if (dh == 2) DYNMM_MHA_F(2, true);
    else DYNMM_MHA_F(32, false);
#undef DYNMM_MHA_F
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_mha_fwd(const float* qkv, float* out, float* probs, int B, int D, int T, int heads, void* stream) {
    return dynmm_mha_drop_fwd(qkv, out, probs, B, D, T, heads, nullptr, stream);
}

extern "C" int dynmm_mha_drop_bwd(const float* g, const float* qkv, const float* probs, float* dqkv, int B, int D, int T,
                                  int heads, const dynmm_dropout* drop, void* stream) {
    (void)hipGetLastError();
    if (!g || !qkv || !probs || !dqkv || B <= 0 || D <= 0 || T <= 0 || heads <= 0 || D % heads != 0 || !drop_ok(drop)) return DYNMM_EINVAL;
    const DropSpec spec = drop_spec(drop);
    if (T > kSeqMaxT || D / heads > kSeqMaxDh) return DYNMM_EUNSUPPORTED;
    const int dh = D / heads;
#define DYNMM_MHA_B(DH, EX) hipLaunchKernelGGL((mha_bwd_kernel<DH, EX>), dim3(B * heads), dim3(64), 0, ST, g, qkv, probs, dqkv, D, T, heads, spec)
    if (dh == 24) DYNMM_MHA_B(24, true);
    else if (dh == 12) DYNMM_MHA_B(12, true);
    else if (dh == 2) DYNMM_MHA_B(2, true);
    else DYNMM_MHA_B(32, false);
#undef DYNMM_MHA_B
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_mha_bwd(const float* g, const float* qkv, const float* probs, float* dqkv, int B, int D, int T,
                             int heads, void* stream) {
    return dynmm_mha_drop_bwd(g, qkv, probs, dqkv, B, D, T, heads, nullptr, stream);
}

extern "C" int dynmm_moe_head(const float* logits, const float* const* preds, int K, const float* target, float temp,
                              int hard, float reg, float* out, float* weight, float* scalars, float* const* d_preds,
                              float* d_logits, int B, void* stream) {
    (void)hipGetLastError();
    if (!logits || !preds || K < 1 || K > 4 || !out || !weight || !scalars || B <= 0 || !(temp > 0.f)) return DYNMM_EINVAL;
    MoePreds P{};
    MoeGrads G{};
    for (int k = 0; k < K; ++k) {
        if (!preds[k]) return DYNMM_EINVAL;
        P.p[k] = preds[k];
        G.p[k] = d_preds ? d_preds[k] : nullptr;
    }
    hipLaunchKernelGGL(moe_head_kernel, dim3(1), dim3(256), 0, ST, logits, P, K, target, temp, hard, reg, out, weight,
                       scalars, G, d_logits, B);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_moe_blend_bwd(const float* d_out, const float* d_aux, const float* logits, const float* const* preds,
                                   int K, const float* weight, float temp, float* const* d_preds, float* d_logits, int B,
                                   void* stream) {
    (void)hipGetLastError();
    if (!logits || !preds || !weight || !d_logits || K < 1 || K > 4 || B <= 0 || !(temp > 0.f)) return DYNMM_EINVAL;
    MoePreds P{};
    MoeGrads G{};
    for (int k = 0; k < K; ++k) {
        if (!preds[k]) return DYNMM_EINVAL;
        P.p[k] = preds[k];
        G.p[k] = d_preds ? d_preds[k] : nullptr;
    }
    hipLaunchKernelGGL(moe_blend_bwd_kernel, dim3(ceil_div(B, 256)), dim3(256), 0, ST, d_out, d_aux, logits, P, K, weight,
                       temp, G, d_logits, B);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" size_t dynmm_clip_grad_norm_workspace_bytes(void) { return 1024 * sizeof(double); }

extern "C" int dynmm_clip_grad_norm(const float* flat_grad, size_t n, float max_norm, double* workspace,
                                    float* norm_and_coef, void* stream) {
    (void)hipGetLastError();
    if (!flat_grad || n == 0 || !workspace || !norm_and_coef) return DYNMM_EINVAL;
    size_t blocks = (n + 4095) / 4096;
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3((unsigned)blocks), dim3(256), 0, ST, flat_grad, n, workspace);
    DYNMM_LAUNCH_CHECK();
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(64), 0, ST, workspace, (int)blocks, max_norm, norm_and_coef);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}
