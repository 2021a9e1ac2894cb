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

// 0 (dropped) or 1/(1-p) (kept); 1 when the site has no dropout.
// A Philox call costs a wave ~900 issue cycles (forty quarter-rate 32-bit multiplies): the shaped accessors below draw
// EIGHT decisions from one call (16 random bits each, keep iff u16 >= round(65536 p), as ffn_kernel does) for the eight
// elements a lane owns — eight consecutive channels of a token (LayerNorm sites) or eight consecutive keys of a query row
// (attention probabilities).  A site is read through ONE accessor by its forward and backward kernels; which element a
// counter serves is therefore a property of the site's kernel family, not of the flat index.
struct DropState {
    const unsigned char* mask;
    unsigned long long off;
    uint32_t k0, k1, thr16;
    float p, inv;
    __device__ __forceinline__ explicit DropState(const DropSpec& d)
        : mask(d.mask), off(d.offset + (d.step ? *d.step : 0ull)), k0((uint32_t)d.seed), k1((uint32_t)(d.seed >> 32)),
          thr16((uint32_t)(d.p * 65536.f + 0.5f)), p(d.p), inv(d.p > 0.f ? 1.f / (1.f - d.p) : 1.f) {}
    // flat sites (dropout_kernel): element idx = counter idx, 24 random bits
    __device__ __forceinline__ float operator()(size_t idx) const {
        if (!(p > 0.f)) return 1.f;
        if (mask) return mask[idx] ? inv : 0.f;
        uint32_t r[4];
        seq_philox4x32_10((uint32_t)idx, (uint32_t)((unsigned long long)idx >> 32), (uint32_t)off, (uint32_t)(off >> 32), k0, k1, r);
        const float u = (float)(r[0] >> 8) * (1.f / 16777216.f);        // [0, 1)
        return u >= p ? inv : 0.f;
    }
    __device__ __forceinline__ void philox8(unsigned long long ctr, float k[8]) const {
        uint32_t r[4];
        seq_philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), (uint32_t)off, (uint32_t)(off >> 32), k0, k1, r);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            k[2 * q] = (r[q] & 0xffffu) >= thr16 ? inv : 0.f;
            k[2 * q + 1] = (r[q] >> 16) >= thr16 ? inv : 0.f;
        }
    }
    // [B, D, T] sites: channels 8 c8 ... 8 c8 + 7 of token (b, t).  Injected flags are indexed by the tensor's flat layout.
    __device__ __forceinline__ void keep8(int b, int c8, int t, int D, int T, float k[8]) const {
        if (!(p > 0.f)) {
#pragma unroll
            for (int e = 0; e < 8; ++e) k[e] = 1.f;
        } else if (mask) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = c8 * 8 + e;
                k[e] = (c < D && mask[((size_t)b * D + c) * T + t]) ? inv : 0.f;
            }
        } else {
            philox8(((unsigned long long)b * ((D + 7) / 8) + c8) * T + t, k);
        }
    }
    __device__ __forceinline__ float keep1(int b, int c, int t, int D, int T) const {
        if (!(p > 0.f)) return 1.f;
        if (mask) return mask[((size_t)b * D + c) * T + t] ? inv : 0.f;
        float k[8];
        philox8(((unsigned long long)b * ((D + 7) / 8) + (c >> 3)) * T + t, k);
        float r = k[0];
#pragma unroll
        for (int e = 1; e < 8; ++e) r = (c & 7) == e ? k[e] : r;
        return r;
    }
    // [R, T] sites (attention probabilities, R = B * heads * T query rows): keys 8 j8 ... 8 j8 + 7 of row r
    __device__ __forceinline__ void row8(size_t r, int j8, int T, float k[8]) const {
        if (!(p > 0.f)) {
#pragma unroll
            for (int e = 0; e < 8; ++e) k[e] = 1.f;
        } else if (mask) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = j8 * 8 + e;
                k[e] = (j < T && mask[r * T + j]) ? inv : 0.f;
            }
        } else {
            philox8((unsigned long long)r * ((T + 7) / 8) + j8, k);
        }
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
// separate add pass.  Workgroup = 16 tokens x 16 channel groups: lanes of a 16-lane row read consecutive t, group k of
// pass p owns the EIGHT consecutive channels 8 (16 p + k) ... + 7 — the eight elements of one Philox call of the site
// (DropState::keep8) — and keeps dropout(x) + res of them in registers: every element is read once and costs an eighth
// of a generator call (round 4: three reads and three calls per element, 25 us per launch at B = 128, D = 120 against
// 4 us of HBM time — a Philox call is ~900 issue cycles of a wave, quarter-rate integer multiplies).
// NP = passes = ceil(D / 128), compile-time so that the cache stays in registers.
constexpr int kLnTok = 16, kLnGrp = 16, kLnCh = 8;

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
template <int NP>
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
    float v[NP][kLnCh], gm[NP][kLnCh], bt[NP][kLnCh];
    float s = 0.f;
    // (loads are unconditional on clamped addresses and the values selected afterwards: a predicate per element is a divergent
    //  branch with its own wait, i.e. a chain of dependent round trips instead of loads in flight together)
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
        const int c8 = pp * kLnGrp + grp;
        float k[kLnCh], xv[kLnCh], rv[kLnCh];
        ds.keep8(b, min(c8, (D - 1) / kLnCh), t, D, T, k);
#pragma unroll
        for (int e = 0; e < kLnCh; ++e) {
            const int cc = min(c8 * kLnCh + e, D - 1);
            const size_t i = base + (size_t)cc * T;
            xv[e] = x[i];
            rv[e] = res ? res[i] : 0.f;
            gm[pp][e] = gamma[cc];                              // (used after the two reductions: requested here)
            bt[pp][e] = beta[cc];
        }
        if (xsum) {
            // slab-major: the eight loads of a slab are in flight together (channel-major, the slab loop of each channel was a
            // chain of dependent round trips — 30 us per launch at eight slabs)
            float bv[kLnCh];
            const float* bp = xbias ? xbias : gamma;
#pragma unroll
            for (int e = 0; e < kLnCh; ++e) bv[e] = bp[min(c8 * kLnCh + e, D - 1)];
            for (int q = 1; q < nparts; ++q) {
                const float* xq = x + (size_t)q * pstride;
#pragma unroll
                for (int e = 0; e < kLnCh; ++e) xv[e] += xq[base + (size_t)min(c8 * kLnCh + e, D - 1) * T];
            }
#pragma unroll
            for (int e = 0; e < kLnCh; ++e) {
                const int c = c8 * kLnCh + e;
                xv[e] += xbias ? bv[e] : 0.f;
                if (ok && c < D) xsum[base + (size_t)c * T] = xv[e];
            }
        }
#pragma unroll
        for (int e = 0; e < kLnCh; ++e) {
            const bool valid = ok && c8 * kLnCh + e < D;
            v[pp][e] = valid ? xv[e] * k[e] + rv[e] : 0.f;
            s += v[pp][e];
        }
    }
    const float mu = ln_group_sum(s, sh, tl, grp) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int pp = 0; pp < NP; ++pp)
#pragma unroll
        for (int e = 0; e < kLnCh; ++e)
            if (ok && (pp * kLnGrp + grp) * kLnCh + e < D) {
                const float d = v[pp][e] - mu;
                q += d * d;
            }
    const float rs = rsqrtf(ln_group_sum(q, sh, tl, grp) / (float)D + eps);
    if (!ok) return;
    if (grp == 0) {
        if (mean) mean[tok] = mu;
        if (rstd) rstd[tok] = rs;
    }
#pragma unroll
    for (int pp = 0; pp < NP; ++pp)
#pragma unroll
        for (int e = 0; e < kLnCh; ++e) {
            const int c = (pp * kLnGrp + grp) * kLnCh + e;
            if (c < D) y[base + (size_t)c * T] = (v[pp][e] - mu) * rs * gm[pp][e] + bt[pp][e];
        }
}

// dx = rstd * (g*gamma - mean_c(g*gamma) - xhat * mean_c(g*gamma*xhat)); the same thread layout and register cache as the
// forward.  `part` (optional, [gridDim.x][2][D]): this workgroup's 16-token sums of g * xhat and g per channel — the
// parameter gradients without a second pass over g, x and res (ln_param_reduce_kernel adds the workgroups' rows in order).
template <int NP>
__global__ void __launch_bounds__(256) ln_bwd_dx_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                        const float* __restrict__ res,
                                                        const float* __restrict__ gamma, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, float* __restrict__ dx,
                                                        float* __restrict__ dres, float* __restrict__ part, int B, int D,
                                                        int T, const DropSpec spec) {
    __shared__ float sh[kLnGrp][kLnTok];
    const DropState ds(spec);
    const int tl = threadIdx.x & (kLnTok - 1), grp = threadIdx.x / kLnTok;
    const int tok = blockIdx.x * kLnTok + tl;
    const bool ok = tok < B * T;
    const int b = ok ? tok / T : 0, t = ok ? tok - b * T : 0;
    const size_t base = (size_t)b * D * T + t;
    const float mu = ok ? mean[tok] : 0.f, rs = ok ? rstd[tok] : 0.f;
    float gv[NP][kLnCh], xh[NP][kLnCh], kk[NP][kLnCh], gm[NP][kLnCh];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int pp = 0; pp < NP; ++pp) {
        const int c8 = pp * kLnGrp + grp;
        ds.keep8(b, min(c8, (D - 1) / kLnCh), t, D, T, kk[pp]);
        float xv[kLnCh], rv[kLnCh];
#pragma unroll
        for (int e = 0; e < kLnCh; ++e) {                        // unconditional loads on clamped addresses (see the forward)
            const int cc = min(c8 * kLnCh + e, D - 1);
            const size_t i = base + (size_t)cc * T;
            gv[pp][e] = g[i];
            xv[e] = x[i];
            rv[e] = res ? res[i] : 0.f;
            gm[pp][e] = gamma[cc];
        }
#pragma unroll
        for (int e = 0; e < kLnCh; ++e) {
            const bool valid = ok && c8 * kLnCh + e < D;
            gv[pp][e] = valid ? gv[pp][e] : 0.f;
            xh[pp][e] = valid ? (xv[e] * kk[pp][e] + rv[e] - mu) * rs : 0.f;
            const float gg = gv[pp][e] * gm[pp][e];
            s1 += gg;
            s2 += gg * xh[pp][e];
        }
    }
    s1 = ln_group_sum(s1, sh, tl, grp) / (float)D;
    s2 = ln_group_sum(s2, sh, tl, grp) / (float)D;
    // d(x*keep + res): the residual branch receives it as is (dres), x through its keep factor (dx)
#pragma unroll
    for (int pp = 0; pp < NP; ++pp)
#pragma unroll
        for (int e = 0; e < kLnCh; ++e) {
            const int c = (pp * kLnGrp + grp) * kLnCh + e;
            if (ok && c < D) {
                const size_t i = base + (size_t)c * T;
                const float dv = rs * (gv[pp][e] * gm[pp][e] - s1 - xh[pp][e] * s2);
                if (dres) dres[i] = dv;
                if (dx) dx[i] = dv * kk[pp][e];
            }
        }
    if (part) {
        // the 16 tokens of a channel sit in the 16 lanes of one row of the wave: xor-butterfly inside the row
        float* pr = part + (size_t)blockIdx.x * 2 * D;
#pragma unroll
        for (int pp = 0; pp < NP; ++pp)
#pragma unroll
            for (int e = 0; e < kLnCh; ++e) {
                float a = gv[pp][e] * xh[pp][e], bs = gv[pp][e];
#pragma unroll
                for (int o = 8; o > 0; o >>= 1) {
                    a += __shfl_xor(a, o, 16);
                    bs += __shfl_xor(bs, o, 16);
                }
                const int c = (pp * kLnGrp + grp) * kLnCh + e;
                if (tl == 0 && c < D) {
                    pr[c] = a;
                    pr[D + c] = bs;
                }
            }
    }
}

// dgamma[c], dbeta[c] = sum over the workgroups' rows of ln_bwd_dx_kernel's `part`, in a fixed order (deterministic):
// workgroup = 4 of the 2 D columns x a 64-way split of the rows (lane = split: at 400 rows a thread adds 7 values), the 64
// splits of a column meet in a wave butterfly.
__global__ void __launch_bounds__(256) ln_param_reduce_kernel(const float* __restrict__ part, int nrows, int D,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta) {
    const int sp = threadIdx.x & 63;
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6);           // 0 .. 2D-1, one wave per column
    if (col >= 2 * D) return;
    const int per = (nrows + 63) / 64;
    const int r1 = min(nrows, (sp + 1) * per);
    float s = 0.f;
    for (int r = sp * per; r < r1; ++r) s += part[(size_t)r * 2 * D + col];
    s = wave_reduce_sum<float>(s);
    if (sp == 0) {
        if (col < D) dgamma[col] = s;
        else dbeta[col - D] = s;
    }
}

// dgamma[c] = sum_tok g * xhat ; dbeta[c] = sum_tok g : one workgroup per channel, fixed summation order (the entry
// without a workspace; one generator call per element, an eighth of it used)
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
        a += gv * (x[i] * ds.keep1(b, c, t, D, T) + (res ? res[i] : 0.f) - mean[tok]) * rstd[tok];
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
// dh = D / H.  One workgroup of NW waves per (b, h); lane i of every wave owns query i (and, in the backward, key i).
//   P = softmax_j( (q_i . k_j) / sqrt(dh) ),  out[c][i] = sum_j P[i][j] v[c][j]
// ---------------------------------------------------------------------------------------------------------------
// DH = compile-time bound on the head dimension (dh <= DH; channels dh .. DH-1 are zero-filled, so the unrolled loops carry
// no predicates).  The K / V / Q / dOut tiles sit in LDS TOKEN-major ([T][DH]): the row of key j is read by every lane at
// the same address (a broadcast) with 16-byte reads — round 4 held them channel-major and paid one ds_read_b32 per FMA.
// The [T][T|1] probability tiles are read as conflict-free rows / columns.  LDS is sized for the launch's T (dynamic).
// Round 6: a wave here is bound by the instructions it issues, and B * H = 640 single-wave workgroups left 3/8 of the SIMDs
// empty and the others with one wave (nothing to hide an LDS wait behind).  The NW waves of a workgroup now split
//   * the KEYS in the phases that are sums over channels (scores; exp + dropout; dP, dS): wave w takes 8-key chunks
//     w * CPW .. — one generator call per chunk (DropState::row8), each decision drawn once per workgroup;
//   * the CHANNELS in the phases that are sums over keys (P'V; dQ, dK, dV): wave w owns channels w * CS .. + CS - 1 of every
//     query — no partial sums to combine.
// Row maxima / denominators / dot products meet in a [NW][64] LDS array.
template <int N>
__device__ __forceinline__ void mha_ld(const float* __restrict__ p, float r[N]) {      // p: aligned to the widest unit used
    if constexpr (N % 4 == 0) {
#pragma unroll
        for (int q = 0; q < N / 4; ++q) {
            const float4 t = reinterpret_cast<const float4*>(p)[q];
            r[4 * q] = t.x; r[4 * q + 1] = t.y; r[4 * q + 2] = t.z; r[4 * q + 3] = t.w;
        }
    } else if constexpr (N % 2 == 0) {
#pragma unroll
        for (int q = 0; q < N / 2; ++q) {
            const float2 t = reinterpret_cast<const float2*>(p)[q];
            r[2 * q] = t.x; r[2 * q + 1] = t.y;
        }
    } else {
#pragma unroll
        for (int q = 0; q < N; ++q) r[q] = p[q];
    }
}

// copy between a [T][T] global tile and the [T][ld] LDS tile, NT consecutive elements per step (coalesced)
template <bool TO_LDS, int NT>
__device__ __forceinline__ void mha_tile_copy(float* __restrict__ gl, float* lds, int T, int ld, const float* rowscale) {
    const int a = NT / T, b0 = NT - a * T;
    int r = threadIdx.x / T, j = threadIdx.x - r * T;
    for (int idx = threadIdx.x; idx < T * T; idx += NT) {
        if (TO_LDS) lds[r * ld + j] = gl[idx];
        else gl[idx] = lds[r * ld + j] * rowscale[r];
        r += a; j += b0;
        if (j >= T) { j -= T; ++r; }
    }
}

template <int DH, int NW, bool EXACT>
__global__ void __launch_bounds__(64 * NW) mha_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                          float* __restrict__ probs, int D, int T, int H,
                                                          const DropSpec spec) {
    constexpr int CS = DH / NW;                                  // channels per wave
    static_assert(CS * NW == DH, "head dimension bound must split evenly over the waves");
    extern __shared__ __align__(16) float smem[];
    const int ld = T | 1;
    float* ks = smem;                    // [T][DH]
    float* vs = ks + T * DH;             // [T][DH]
    float* ps = vs + T * DH;             // [T][ld]  scores -> exp
    float* pk = ps + T * ld;             // [T][ld]  exp * keep / (1 - p)
    float* red = pk + T * ld;            // [NW][64]
    float* invs = red + NW * 64;         // [64]
    const DropState drop(spec);
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int dh = D / H;
    const int i = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool act = i < T;
    const float* base = qkv + (size_t)b * 3 * D * T;
    const float scale = rsqrtf((float)dh);
    // (unconditional loads on clamped indices, values selected afterwards: a predicate per element would make every load a
    //  branch with its own wait — 24 dependent round trips instead of 24 loads in flight)
    const int ic = min(i, T - 1);
    float q[DH];
#pragma unroll
    for (int c = 0; c < DH; ++c) {
        const float t = base[(size_t)(h * dh + (EXACT ? c : min(c, dh - 1))) * T + ic] * scale;      // torch scales q before q @ k^T
        q[c] = ((EXACT || c < dh) && act) ? t : 0.f;
    }
#pragma unroll
    for (int e = 0; e < CS; ++e) {
        const int c = w * CS + e, cc = EXACT ? c : min(c, dh - 1);
        const float kv = base[(size_t)(D + h * dh + cc) * T + ic];
        const float vv = base[(size_t)(2 * D + h * dh + cc) * T + ic];
        if (act) {
            ks[i * DH + c] = (EXACT || c < dh) ? kv : 0.f;
            vs[i * DH + c] = (EXACT || c < dh) ? vv : 0.f;
        }
    }
    __syncthreads();
    const int nchunk = (T + 7) >> 3, cpw = (nchunk + NW - 1) / NW;
    const int j_lo = min(T, w * cpw * 8), j_hi = min(T, (w + 1) * cpw * 8);
    const size_t row = (size_t)blockIdx.x * T + i;             // probs (and their keep flags) are [B*H][T][T]
    float* pr = ps + i * ld;
    float* pkr = pk + i * ld;
    float mx = -INFINITY;
    if (act)
        for (int j = j_lo; j < j_hi; ++j) {
            float kr[DH];
            mha_ld<DH>(ks + j * DH, kr);
            float sc = 0.f;
#pragma unroll
            for (int c = 0; c < DH; ++c) sc += q[c] * kr[c];
            pr[j] = sc;
            mx = fmaxf(mx, sc);
        }
    red[w * 64 + i] = mx;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NW; ++k) mx = fmaxf(mx, red[k * 64 + i]);
    __syncthreads();                                            // (red is re-used for the denominators)
    float den = 0.f;
    if (act)
        for (int j0 = j_lo; j0 < j_hi; j0 += 8) {
            float kp[8];
            drop.row8(row, j0 >> 3, T, kp);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = j0 + e;
                if (j < T) {
                    const float ex = expf(pr[j] - mx);
                    pr[j] = ex;
                    den += ex;
                    pkr[j] = ex * kp[e];                         // dropout acts on the normalised probabilities: linear in e
                }
            }
        }
    red[w * 64 + i] = den;
    __syncthreads();
    den = 0.f;
#pragma unroll
    for (int k = 0; k < NW; ++k) den += red[k * 64 + i];
    const float inv = 1.f / den;
    if (w == 0) invs[i] = inv;
    if (act) {
        float o[CS];
#pragma unroll
        for (int e = 0; e < CS; ++e) o[e] = 0.f;
        for (int j = 0; j < T; ++j) {
            float vr[CS];
            mha_ld<CS>(vs + j * DH + w * CS, vr);
            const float pj = pkr[j];
#pragma unroll
            for (int e = 0; e < CS; ++e) o[e] += pj * vr[e];
        }
        float* ob = out + (size_t)b * D * T;
#pragma unroll
        for (int e = 0; e < CS; ++e) {
            const int c = w * CS + e;
            if (EXACT || c < dh) ob[(size_t)(h * dh + c) * T + i] = o[e] * inv;
        }
    }
    __syncthreads();
    // saved for the backward: the probabilities BEFORE dropout
    mha_tile_copy<false, 64 * NW>(probs + (size_t)blockIdx.x * T * T, ps, T, ld, invs);
}

template <int DH, int NW, bool EXACT>
__global__ void __launch_bounds__(64 * NW) mha_bwd_kernel(const float* __restrict__ g, const float* __restrict__ qkv,
                                                          const float* __restrict__ probs, float* __restrict__ dqkv, int D,
                                                          int T, int H, const DropSpec spec) {
    constexpr int CS = DH / NW;
    static_assert(CS * NW == DH, "head dimension bound must split evenly over the waves");
    extern __shared__ __align__(16) float smem[];
    const int ld = T | 1;
    float* qs = smem;                    // [T][DH] each
    float* ks = qs + T * DH;
    float* vs = ks + T * DH;
    float* gs = vs + T * DH;
    float* ps = gs + T * DH;             // [T][ld]  P -> P' = P * keep / (1 - p)
    float* ds = ps + T * ld;             // [T][ld]  dP -> dS
    float* red = ds + T * ld;            // [NW][64]
    const DropState drop(spec);
    const int b = blockIdx.x / H, h = blockIdx.x - b * H;
    const int dh = D / H;
    const int i = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool act = i < T;
    const float* base = qkv + (size_t)b * 3 * D * T;
    const float* gb = g + (size_t)b * D * T;
    const int ic = min(i, T - 1);                                // unconditional loads on clamped indices (see the forward)
#pragma unroll
    for (int e = 0; e < CS; ++e) {
        const int c = w * CS + e, cc = EXACT ? c : min(c, dh - 1);
        const float qv = base[(size_t)(h * dh + cc) * T + ic];
        const float kv = base[(size_t)(D + h * dh + cc) * T + ic];
        const float vv = base[(size_t)(2 * D + h * dh + cc) * T + ic];
        const float gv = gb[(size_t)(h * dh + cc) * T + ic];
        if (act) {
            const bool in = EXACT || c < dh;
            qs[i * DH + c] = in ? qv : 0.f;
            ks[i * DH + c] = in ? kv : 0.f;
            vs[i * DH + c] = in ? vv : 0.f;
            gs[i * DH + c] = in ? gv : 0.f;
        }
    }
    mha_tile_copy<true, 64 * NW>(const_cast<float*>(probs) + (size_t)blockIdx.x * T * T, ps, T, ld, nullptr);
    __syncthreads();
    // P' = P * keep (what multiplied V);  dP'[i][j] = sum_c g[c][i] v[c][j];  dP = dP' * keep;
    // dS = P * (dP - sum_j P dP).  Row i then keeps P' in ps (dV needs it), P is not used again.
    const int nchunk = (T + 7) >> 3, cpw = (nchunk + NW - 1) / NW;
    const int j_lo = min(T, w * cpw * 8), j_hi = min(T, (w + 1) * cpw * 8);
    const size_t row = (size_t)blockIdx.x * T + i;
    float* pr = ps + i * ld;
    float* dr = ds + i * ld;
    float dot = 0.f;
    unsigned long long kept = 0ull;                              // this wave's keep decisions of row i (bit j), drawn once
    if (act) {
        float gq[DH];                                            // dOut column of query i
        mha_ld<DH>(gs + i * DH, gq);
        for (int j0 = j_lo; j0 < j_hi; j0 += 8) {
            float kp[8];
            drop.row8(row, j0 >> 3, T, kp);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int j = j0 + e;
                if (j < T) {
                    float vr[DH];
                    mha_ld<DH>(vs + j * DH, vr);
                    float sacc = 0.f;
#pragma unroll
                    for (int c = 0; c < DH; ++c) sacc += gq[c] * vr[c];
                    sacc *= kp[e];
                    if (kp[e] != 0.f) kept |= 1ull << j;
                    dr[j] = sacc;
                    dot += pr[j] * sacc;
                }
            }
        }
    }
    red[w * 64 + i] = dot;
    __syncthreads();
    dot = 0.f;
#pragma unroll
    for (int k = 0; k < NW; ++k) dot += red[k * 64 + i];
    if (act)
        for (int j = j_lo; j < j_hi; ++j) {
            const float pv = pr[j];
            dr[j] = pv * (dr[j] - dot);
            pr[j] = ((kept >> j) & 1ull) ? pv * drop.inv : 0.f;
        }
    __syncthreads();
    if (!act) return;
    const float scale = rsqrtf((float)dh);
    float dq[CS], dk[CS], dv[CS];
#pragma unroll
    for (int e = 0; e < CS; ++e) dq[e] = dk[e] = dv[e] = 0.f;
    for (int j = 0; j < T; ++j) {
        const float dsr = ds[i * ld + j], dsc = ds[j * ld + i], pc = ps[j * ld + i];     // row (query i), columns (key i)
        float kr[CS], qr[CS], gr[CS];
        mha_ld<CS>(ks + j * DH + w * CS, kr);
        mha_ld<CS>(qs + j * DH + w * CS, qr);
        mha_ld<CS>(gs + j * DH + w * CS, gr);
#pragma unroll
        for (int e = 0; e < CS; ++e) {
            dq[e] += dsr * kr[e];
            dk[e] += dsc * qr[e];
            dv[e] += pc * gr[e];
        }
    }
    float* db = dqkv + (size_t)b * 3 * D * T;
#pragma unroll
    for (int e = 0; e < CS; ++e) {
        const int c = w * CS + e;
        if (EXACT || c < dh) {
            db[(size_t)(h * dh + c) * T + i] = dq[e] * scale;
            db[(size_t)(D + h * dh + c) * T + i] = dk[e] * scale;
            db[(size_t)(2 * D + h * dh + c) * T + i] = dv[e];
        }
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

// one wave: lane l adds parts l, l + 64, ... in order, the 64 lane sums meet in a fixed butterfly (deterministic)
__global__ void __launch_bounds__(64) clip_coef_kernel(const double* __restrict__ part, int nparts, float max_norm,
                                                       float* __restrict__ out2) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 64) s += part[i];
    s = wave_reduce_sum<double>(s);
    if (threadIdx.x != 0) return;
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

// NP = ceil(D / 128) passes of the 16 x 8-channel thread layout, rounded up to an instantiated count
#define DYNMM_LN_DISPATCH(D, CALL)                 \
    do {                                           \
        if ((D) <= 128) { CALL(1); }               \
        else if ((D) <= 256) { CALL(2); }          \
        else if ((D) <= 512) { CALL(4); }          \
        else { CALL(8); }                          \
    } while (0)
constexpr int kLnMaxD = 1024;

static int launch_ln_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* y, float* mean,
                         float* rstd, int B, int D, int T, float eps, const DropSpec& spec, int nparts, size_t pstride,
                         const float* xbias, float* xsum, hipStream_t st) {
#define DYNMM_LN_F(NP) hipLaunchKernelGGL((ln_fwd_kernel<NP>), dim3(ceil_div(B * T, kLnTok)), dim3(256), 0, st, x, res, gamma, beta, \
                                          y, mean, rstd, B, D, T, eps, spec, nparts, pstride, xbias, xsum)
    DYNMM_LN_DISPATCH(D, DYNMM_LN_F);
#undef DYNMM_LN_F
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_layernorm_drop_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                                        float* mean, float* rstd, int B, int D, int T, float eps,
                                        const dynmm_dropout* drop, void* stream) {
    (void)hipGetLastError();
    if (!x || !gamma || !beta || !y || B <= 0 || D <= 0 || T <= 0 || !drop_ok(drop)) return DYNMM_EINVAL;
    if (D > kLnMaxD) return DYNMM_EUNSUPPORTED;
    return launch_ln_fwd(x, res, gamma, beta, y, mean, rstd, B, D, T, eps, drop_spec(drop), 1, (size_t)0, nullptr, nullptr, ST);
}

extern "C" int dynmm_layernorm_parts_fwd(const float* parts, int nparts, const float* xbias, float* xsum, const float* res,
                                         const float* gamma, const float* beta, float* y, float* mean, float* rstd, int B,
                                         int D, int T, float eps, const dynmm_dropout* drop, void* stream) {
    (void)hipGetLastError();
    if (!parts || nparts <= 0 || !xsum || !gamma || !beta || !y || B <= 0 || D <= 0 || T <= 0 || !drop_ok(drop)) return DYNMM_EINVAL;
    if (D > kLnMaxD) return DYNMM_EUNSUPPORTED;
    return launch_ln_fwd(parts, res, gamma, beta, y, mean, rstd, B, D, T, eps, drop_spec(drop), nparts, (size_t)B * D * T,
                         xbias, xsum, ST);
}

extern "C" int dynmm_layernorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                                   float* mean, float* rstd, int B, int D, int T, float eps, void* stream) {
    return dynmm_layernorm_drop_fwd(x, res, gamma, beta, y, mean, rstd, B, D, T, eps, nullptr, stream);
}

extern "C" size_t dynmm_layernorm_bwd_workspace_bytes(int B, int D, int T) {
    if (B <= 0 || D <= 0 || T <= 0) return 0;
    return (size_t)ceil_div(B * T, kLnTok) * 2 * D * sizeof(float);
}

// workspace (optional, dynmm_layernorm_bwd_workspace_bytes): the parameter gradients come out of the input-gradient pass
// (per-workgroup sums + one ordered reduction) instead of a second pass over g, x and res.
extern "C" int dynmm_layernorm_drop_bwd_ws(const float* g, const float* x, const float* res, const float* gamma,
                                           const float* mean, const float* rstd, float* dx, float* dres, float* dgamma,
                                           float* dbeta, int B, int D, int T, const dynmm_dropout* drop, float* workspace,
                                           size_t workspace_bytes, void* stream) {
    (void)hipGetLastError();
    if (!g || !x || !gamma || !mean || !rstd || B <= 0 || D <= 0 || T <= 0 || !drop_ok(drop)) return DYNMM_EINVAL;
    if (D > kLnMaxD) return DYNMM_EUNSUPPORTED;
    const bool params = dgamma && dbeta;
    const bool fused = params && workspace && workspace_bytes >= dynmm_layernorm_bwd_workspace_bytes(B, D, T);
    if (workspace && params && !fused) return DYNMM_EINVAL;
    const DropSpec spec = drop_spec(drop);
    const int nblk = ceil_div(B * T, kLnTok);
    if (dx || dres || fused) {
        float* part = fused ? workspace : nullptr;
#define DYNMM_LN_B(NP) hipLaunchKernelGGL((ln_bwd_dx_kernel<NP>), dim3(nblk), dim3(256), 0, ST, g, x, res, gamma, mean, rstd, dx, dres, \
                                          part, B, D, T, spec)
        DYNMM_LN_DISPATCH(D, DYNMM_LN_B);
#undef DYNMM_LN_B
        DYNMM_LAUNCH_CHECK();
    }
    if (fused) {
        hipLaunchKernelGGL(ln_param_reduce_kernel, dim3(ceil_div(2 * D, 4)), dim3(256), 0, ST, workspace, nblk, D, dgamma, dbeta);
        DYNMM_LAUNCH_CHECK();
    } else if (params) {
        hipLaunchKernelGGL(ln_bwd_param_kernel, dim3(D), dim3(256), 0, ST, g, x, res, mean, rstd, dgamma, dbeta, B, D, T, spec);
        DYNMM_LAUNCH_CHECK();
    }
    return DYNMM_OK;
}

extern "C" int dynmm_layernorm_drop_bwd(const float* g, const float* x, const float* res, const float* gamma,
                                        const float* mean, const float* rstd, float* dx, float* dres, float* dgamma,
                                        float* dbeta, int B, int D, int T, const dynmm_dropout* drop, void* stream) {
    return dynmm_layernorm_drop_bwd_ws(g, x, res, gamma, mean, rstd, dx, dres, dgamma, dbeta, B, D, T, drop, nullptr, 0, stream);
}

extern "C" int dynmm_layernorm_bwd(const float* g, const float* x, const float* res, const float* gamma,
                                   const float* mean, const float* rstd, float* dx, float* dgamma, float* dbeta, int B,
                                   int D, int T, void* stream) {
    return dynmm_layernorm_drop_bwd(g, x, res, gamma, mean, rstd, dx, nullptr, dgamma, dbeta, B, D, T, nullptr, stream);
}

template <int DH, int NW, bool EXACT, bool BWD>
static int launch_mha(const float* g, const float* qkv, float* out_or_dqkv, float* probs, int B, int D, int T, int heads,
                      const DropSpec& spec, hipStream_t st) {
    const int ld = T | 1;
    const size_t lds = (BWD ? (size_t)4 * T * DH + (size_t)2 * T * ld + NW * 64
                            : (size_t)2 * T * DH + (size_t)2 * T * ld + NW * 64 + 64) * sizeof(float);
    static size_t attr_done = 64 * 1024;                      // the runtime's default bound on dynamic LDS
    if (lds > attr_done) {
        const void* fn = BWD ? reinterpret_cast<const void*>(&mha_bwd_kernel<DH, NW, EXACT>)
                             : reinterpret_cast<const void*>(&mha_fwd_kernel<DH, NW, EXACT>);
        DYNMM_HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = lds;
    }
    if (BWD)
        hipLaunchKernelGGL((mha_bwd_kernel<DH, NW, EXACT>), dim3(B * heads), dim3(64 * NW), lds, st, g, qkv,
                           (const float*)probs, out_or_dqkv, D, T, heads, spec);
    else
        hipLaunchKernelGGL((mha_fwd_kernel<DH, NW, EXACT>), dim3(B * heads), dim3(64 * NW), lds, st, qkv, out_or_dqkv, probs,
                           D, T, heads, spec);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

template <bool BWD>
static int dispatch_mha(const float* g, const float* qkv, float* out_or_dqkv, float* probs, int B, int D, int T, int heads,
                        const DropSpec& spec, hipStream_t st) {
    const int dh = D / heads;
    if (dh == 24) return launch_mha<24, 4, true, BWD>(g, qkv, out_or_dqkv, probs, B, D, T, heads, spec, st);
    if (dh == 12) return launch_mha<12, 4, true, BWD>(g, qkv, out_or_dqkv, probs, B, D, T, heads, spec, st);
    if (dh == 2) return launch_mha<2, 2, true, BWD>(g, qkv, out_or_dqkv, probs, B, D, T, heads, spec, st);
    return launch_mha<32, 4, false, BWD>(g, qkv, out_or_dqkv, probs, B, D, T, heads, spec, st);
}

extern "C" int dynmm_mha_drop_fwd(const float* qkv, float* out, float* probs, int B, int D, int T, int heads,
                                  const dynmm_dropout* drop, void* stream) {
    (void)hipGetLastError();
    if (!qkv || !out || !probs || B <= 0 || D <= 0 || T <= 0 || heads <= 0 || D % heads != 0 || !drop_ok(drop)) return DYNMM_EINVAL;
    if (T > kSeqMaxT || D / heads > kSeqMaxDh) return DYNMM_EUNSUPPORTED;
    return dispatch_mha<false>(nullptr, qkv, out, probs, B, D, T, heads, drop_spec(drop), ST);
}

extern "C" int dynmm_mha_fwd(const float* qkv, float* out, float* probs, int B, int D, int T, int heads, void* stream) {
    return dynmm_mha_drop_fwd(qkv, out, probs, B, D, T, heads, nullptr, stream);
}

extern "C" int dynmm_mha_drop_bwd(const float* g, const float* qkv, const float* probs, float* dqkv, int B, int D, int T,
                                  int heads, const dynmm_dropout* drop, void* stream) {
    (void)hipGetLastError();
    if (!g || !qkv || !probs || !dqkv || B <= 0 || D <= 0 || T <= 0 || heads <= 0 || D % heads != 0 || !drop_ok(drop)) return DYNMM_EINVAL;
    if (T > kSeqMaxT || D / heads > kSeqMaxDh) return DYNMM_EUNSUPPORTED;
    return dispatch_mha<true>(g, qkv, dqkv, const_cast<float*>(probs), B, D, T, heads, drop_spec(drop), ST);
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
