// The training tail of the decoder as ONE forward and ONE backward kernel: the last learned 2x up-sampling
// (model.py:404-410, 40 channels, 240x320 -> 480x640) fused with the full-resolution weighted cross entropy
// (src/utils.py:34-50).  In training the 49 MB/img logits are consumed by the loss only, so they are never
// materialised: the forward keeps a pixel's 40 logits in flight (online log-sum-exp) and stores one float per
// output pixel (its log-sum-exp); the backward re-derives every logit it needs from the 3x3 input neighbourhood
// it has in registers anyway.  Unfused, this tail moved ~9.4 GB per batch-32 step through HBM (logits written
// once and read four times, their gradient written once and read twice: 2.9 ms of kernel time at the HBM
// roofline); fused it reads the 12 MB/img input twice and writes 2 MB/img of log-sum-exps.
//
// Stencil (pointwise.hip::upsample_fwd_kernel): nearest-2x followed by a zero-padded 3x3 collapses to a 2x2 window
// of the INPUT with pre-summed taps: output row 2i reads input rows {i-1: w0, i: w1+w2}, row 2i+1 reads
// {i: w0+w1, i+1: w2} (same along columns; out-of-range input pixels contribute 0 = the zero padding).
#include "common.h"

namespace dynmm {

constexpr int kTailMaxC = 64;
constexpr int kCoefLd = 20;       // 16 pre-summed taps + bias, rows padded to 16 bytes (ds_read_b128 broadcasts)

// coef[c][tail_ci(ar, wr, ac, wc)] (16 per channel) + bias: built once per workgroup in LDS.  The two column parities of a tap
// sit next to each other, ac = 1 first: frame outputs 2k (ac = 1) and 2k + 1 (ac = 0) read the SAME 2x2 input window, so their
// logits and their contributions to dx are ONE packed operation each (v_pk_fma_f32: two fp32 FMAs per lane and issue slot) with
// an 8-byte LDS read of the coefficient pair (round 6; the backward is bound by its vector issue slots).
typedef float tail_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ constexpr int tail_ci(int ar, int wr, int ac, int wc) { return ((ar * 2 + wr) * 2 + wc) * 2 + (1 - ac); }
__device__ __forceinline__ void tail_build_coef(const float* __restrict__ wgt, const float* __restrict__ bias,
                                                float (*coef)[kCoefLd], int C) {
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float k[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) k[j] = wgt[c * 9 + j];
#pragma unroll
        for (int ar = 0; ar < 2; ++ar)
#pragma unroll
            for (int wr = 0; wr < 2; ++wr)
#pragma unroll
                for (int ac = 0; ac < 2; ++ac)
#pragma unroll
                    for (int wc = 0; wc < 2; ++wc) {
                        float s = 0.f;
#pragma unroll
                        for (int r = 0; r < 3; ++r) {
                            const bool rin = ar == 0 ? (wr == 0 ? r == 0 : r >= 1) : (wr == 0 ? r <= 1 : r == 2);
                            if (!rin) continue;
#pragma unroll
                            for (int q = 0; q < 3; ++q) {
                                const bool qin = ac == 0 ? (wc == 0 ? q == 0 : q >= 1) : (wc == 0 ? q <= 1 : q == 2);
                                if (qin) s += k[r * 3 + q];
                            }
                        }
                        coef[c][tail_ci(ar, wr, ac, wc)] = s;
                    }
        coef[c][16] = bias ? bias[c] : 0.f;
    }
    __syncthreads();
}

// Work decomposition.  A workgroup owns a tile of kTR x kTC = 16 x 64 INPUT pixels of one image; lane (by, bx) =
// (tid >> 5, tid & 31) owns the 2x2 block at tile rows 2by, 2by+1 / cols 2bx, 2bx+1.  The lane's 4x4 patch covers
// input rows i0-1..i0+2 / cols j0-1..j0+2; the 6x6 "frame" of output pixels (a, b) <-> output row 2*i0 + a - 1, col
// 2*j0 + b - 1 holds every output whose 2x2 input window touches the block; the lane's OWN outputs are a, b in 1..4.
// Output (a, b): row parity ar = (a+1)&1, window = patch rows (a>>1, (a>>1)+1); same along columns.
//
// Staging.  The kernels are a long dependent chain per channel, and a lane-private register prefetch left 80 % of
// the wave cycles waiting on HBM (2 waves / SIMD at ~180 VGPRs).  So the (16+2) x (64+2) halo tile of kCH = 4
// channels at a time goes through LDS, double buffered: the global loads of chunk k+1 are issued before the
// arithmetic of chunk k and land in LDS after it (one barrier per chunk), which keeps ~20 KB per workgroup in
// flight for the whole duration of a chunk's arithmetic.  Pixels outside the image read as the conv's zero padding.
constexpr int kTR = 16, kTC = 64, kCH = 4;
constexpr int kPC = kTC + 2;                    // tile row length in LDS (even: the 2-float reads stay 8-byte aligned)
constexpr int kTile = (kTR + 2) * kPC;          // 1188 floats per channel
constexpr int kTileLoads = (kTile + 255) / 256;

struct TailGeom {
    int n, ti0, tj0;        // image, tile origin (input pixels)
    int i0, j0;             // the lane's block origin
    int lofs;               // LDS offset of the lane's patch inside a channel tile
    int gofs[kTileLoads];   // per staging load: offset inside a channel plane (clamped)
    bool gok[kTileLoads];   // ... inside the image
    bool gin[kTileLoads];   // ... inside the tile (the last load is partial)
};

__device__ __forceinline__ TailGeom tail_geom(int N, int H, int W) {
    TailGeom g;
    const int tw = (W + kTC - 1) / kTC, th = (H + kTR - 1) / kTR;
    int b = blockIdx.x;
    g.n = b / (th * tw);
    b -= g.n * th * tw;
    g.ti0 = (b / tw) * kTR;
    g.tj0 = (b - (b / tw) * tw) * kTC;
    const int by = threadIdx.x >> 5, bx = threadIdx.x & 31;
    g.i0 = g.ti0 + 2 * by;
    g.j0 = g.tj0 + 2 * bx;
    g.lofs = 2 * by * kPC + 2 * bx;
#pragma unroll
    for (int k = 0; k < kTileLoads; ++k) {
        const int e = k * 256 + threadIdx.x;
        const int r = e / kPC, q = e - r * kPC;
        const int ii = g.ti0 - 1 + r, jj = g.tj0 - 1 + q;
        g.gin[k] = e < kTile;
        g.gok[k] = g.gin[k] && ii >= 0 && ii < H && jj >= 0 && jj < W;
        g.gofs[k] = min(max(ii, 0), H - 1) * W + min(max(jj, 0), W - 1);
    }
    return g;
}

// issue the loads of channels [c0, c0 + kCH) of image plane stack xn (unconditional, clamped addresses)
__device__ __forceinline__ void tail_stage_load(const float* __restrict__ xn, int HW, int c0, int C, const TailGeom& g,
                                                float (&st)[kCH][kTileLoads]) {
#pragma unroll
    for (int s = 0; s < kCH; ++s) {
        const float* xc = xn + (size_t)min(c0 + s, C - 1) * HW;
#pragma unroll
        for (int k = 0; k < kTileLoads; ++k) st[s][k] = xc[g.gofs[k]];
    }
}

__device__ __forceinline__ void tail_stage_store(float* __restrict__ buf, const TailGeom& g,
                                                 const float (&st)[kCH][kTileLoads]) {
#pragma unroll
    for (int s = 0; s < kCH; ++s)
#pragma unroll
        for (int k = 0; k < kTileLoads; ++k)
            if (g.gin[k]) buf[s * kTile + k * 256 + threadIdx.x] = g.gok[k] ? st[s][k] : 0.f;
}

__device__ __forceinline__ void tail_read_patch(const float* __restrict__ tile, const TailGeom& g, float (&v)[4][4]) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float2 lo = *reinterpret_cast<const float2*>(tile + g.lofs + r * kPC);
        const float2 hi = *reinterpret_cast<const float2*>(tile + g.lofs + r * kPC + 2);
        v[r][0] = lo.x;
        v[r][1] = lo.y;
        v[r][2] = hi.x;
        v[r][3] = hi.y;
    }
}

// logit of frame output (a, b); a, b are compile-time after unrolling
__device__ __forceinline__ float tail_logit(const float* __restrict__ cf, const float (&v)[4][4], int a, int b) {
    const int ar = (a + 1) & 1, ac = (b + 1) & 1, r0 = a >> 1, c0 = b >> 1;
    return cf[16] + cf[tail_ci(ar, 0, ac, 0)] * v[r0][c0] + cf[tail_ci(ar, 0, ac, 1)] * v[r0][c0 + 1] +
           cf[tail_ci(ar, 1, ac, 0)] * v[r0 + 1][c0] + cf[tail_ci(ar, 1, ac, 1)] * v[r0 + 1][c0 + 1];
}

// sum over the 64 lanes in DPP adds (no LDS traffic); the total lands in lane 63
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
#define DYNMM_DPP(x, ctrl, rmask, bmask) \
    __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, rmask, bmask, true))
    float t = v + DYNMM_DPP(v, 0x111, 0xf, 0xf);     // row_shr:1
    t += DYNMM_DPP(v, 0x112, 0xf, 0xf);              // row_shr:2
    t += DYNMM_DPP(v, 0x113, 0xf, 0xf);              // row_shr:3   -> sums of 4
    t += DYNMM_DPP(t, 0x114, 0xf, 0xe);              // row_shr:4, banks 1..3 -> sums of 8
    t += DYNMM_DPP(t, 0x118, 0xf, 0xc);              // row_shr:8, banks 2..3 -> lane 15 of a row = the row's sum
    t += DYNMM_DPP(t, 0x142, 0xa, 0xf);              // row_bcast:15 into rows 1, 3
    t += DYNMM_DPP(t, 0x143, 0xc, 0xf);              // row_bcast:31 into rows 2, 3 -> lane 63 = total
#undef DYNMM_DPP
    return t;
}

// ---- forward: all C channels of the lane's 4x4 own outputs, online log-sum-exp --------------------------------
__global__ void __launch_bounds__(256) up2ce_fwd_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                                        const float* __restrict__ bias,
                                                        const unsigned char* __restrict__ target,
                                                        const float* __restrict__ cw, float* __restrict__ lse,
                                                        double* __restrict__ acc2, int N, int C, int H, int W) {
    __shared__ __attribute__((aligned(16))) float coef[kTailMaxC][kCoefLd];
    __shared__ __attribute__((aligned(16))) float tiles[2][kCH * kTile];
    __shared__ float red[4];
    tail_build_coef(wgt, bias, coef, C);
    const int HW = H * W, H2 = 2 * H, W2 = 2 * W;
    const TailGeom g = tail_geom(N, H, W);
    const unsigned char* tn = target + (size_t)g.n * 4 * HW;
    int tt[16];
    bool ok[16];
    float m[16], s[16], xt[16];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int oh = 2 * g.i0 + a, ow = 2 * g.j0 + b, o = a * 4 + b;
            ok[o] = oh < H2 && ow < W2;
            // (unconditional load on a clamped address: sixteen loads in flight instead of sixteen predicated round trips)
            const int traw = (int)tn[(size_t)min(oh, H2 - 1) * W2 + min(ow, W2 - 1)];
            tt[o] = ok[o] ? traw - 1 : -1;
            m[o] = -INFINITY;
            s[o] = 0.f;
            xt[o] = 0.f;
        }
    const float* xn = x + (size_t)g.n * C * HW;
    float st[kCH][kTileLoads];
    tail_stage_load(xn, HW, 0, C, g, st);
    tail_stage_store(tiles[0], g, st);
    __syncthreads();
    for (int c0 = 0, kb = 0; c0 < C; c0 += kCH, kb ^= 1) {
        const bool more = c0 + kCH < C;
        if (more) tail_stage_load(xn, HW, c0 + kCH, C, g, st);          // in flight during this chunk's arithmetic
        for (int sl = 0; sl < kCH && c0 + sl < C; ++sl) {
            const int c = c0 + sl;
            float cur[4][4];
            tail_read_patch(tiles[kb] + sl * kTile, g, cur);
            const float* cf = coef[c];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int o = a * 4 + b;
                    const float l = tail_logit(cf, cur, a + 1, b + 1);
                    // online log-sum-exp with ONE exponential: e = exp(-|l - m|) rescales whichever side is smaller
                    const float d = l - m[o];
                    const float e = __expf(-fabsf(d));
                    s[o] = d > 0.f ? fmaf(s[o], e, 1.f) : s[o] + e;
                    m[o] = fmaxf(m[o], l);
                    xt[o] = (tt[o] == c) ? l : xt[o];
                }
        }
        if (more) tail_stage_store(tiles[kb ^ 1], g, st);
        __syncthreads();
    }
    float ls = 0.f, ws = 0.f;
    float* ln = lse + (size_t)g.n * 4 * HW;
    float wq[16];
#pragma unroll
    for (int o = 0; o < 16; ++o) wq[o] = cw[min(max(tt[o], 0), C - 1)];      // class weights requested together
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int o = a * 4 + b;
            if (ok[o]) {
                const float e = m[o] + logf(s[o]);
                ln[(size_t)(2 * g.i0 + a) * W2 + 2 * g.j0 + b] = e;
                if (tt[o] >= 0 && tt[o] < C) {
                    ls += wq[o] * (e - xt[o]);
                    ws += wq[o];
                }
            }
        }
    const float tl = block_reduce_sum_256<float>(ls, red);
    const float tw = block_reduce_sum_256<float>(ws, red);
    if (threadIdx.x == 0) {
        atomicAdd(&acc2[0], (double)tl);
        atomicAdd(&acc2[1], (double)tw);
    }
}

// ---- backward ---------------------------------------------------------------------------------------------------
// The lane owns dx of its 2x2 input pixels, which collect from the 6x6 frame; every frame output's window lies in
// the lane's 4x4 patch, so each logit is recomputed locally (2.25 logits per output pixel instead of a 49 MB/img
// round trip through HBM):  dlogit[c] = k * (exp(logit[c] - lse) - [c == t]),  k = cw[t] * gscale.
// k is folded into the exponent (lse' = lse - log k; 1e30 for void / outside pixels, whose exp() is then exactly 0),
// which leaves 6 VALU operations per recomputed output and channel after the 4 FMAs of the logit.
// dw / db of the depthwise conv come from the lane's OWN 4x4 outputs (each output counted once), reduced over the
// wave in DPP adds, over the workgroup in LDS (fixed order) and written as one partial row per workgroup.
__global__ void __launch_bounds__(256) up2ce_bwd_kernel(const float* __restrict__ x, const float* __restrict__ wgt,
                                                        const float* __restrict__ bias,
                                                        const unsigned char* __restrict__ target,
                                                        const float* __restrict__ cw, const float* __restrict__ lse,
                                                        const float* __restrict__ gscale, float* __restrict__ dx,
                                                        float* __restrict__ part, int N, int C, int H, int W) {
    __shared__ __attribute__((aligned(16))) float coef[kTailMaxC][kCoefLd];
    __shared__ __attribute__((aligned(16))) float tiles[2][kCH * kTile];
    __shared__ float wpart[4][kTailMaxC][10];
    tail_build_coef(wgt, bias, coef, C);
    const int HW = H * W, H2 = 2 * H, W2 = 2 * W;
    const TailGeom g = tail_geom(N, H, W);
    const float gs = gscale[0];
    float le[36];
    unsigned tcp[9];        // target class + 1 per frame output, one byte each (0 = no gradient)
    const unsigned char* tn = target + (size_t)g.n * 4 * HW;
    const float* ln = lse + (size_t)g.n * 4 * HW;
#pragma unroll
    for (int w4 = 0; w4 < 9; ++w4) tcp[w4] = 0u;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            const int oh = 2 * g.i0 + a - 1, ow = 2 * g.j0 + b - 1, o = a * 6 + b;
            const bool in = oh >= 0 && oh < H2 && ow >= 0 && ow < W2;
            // unconditional loads on clamped addresses, values selected afterwards: the frame's 36 targets, 36 log-sum-exps and
            // 36 class weights are three batches of loads in flight instead of 108 predicated, dependent round trips
            const size_t at = (size_t)min(max(oh, 0), H2 - 1) * W2 + min(max(ow, 0), W2 - 1);
            // (masks and arithmetic on the loaded values rather than selects around them: the compiler sinks a load whose only
            //  use sits in one arm of a select back under that arm's branch)
            const int t = (int)tn[at] & -(int)in;
            const float lv = ln[at];
            const float cv = cw[min(max(t - 1, 0), C - 1)];
            const float kk = cv * gs * ((t >= 1 && t <= C) ? 1.f : 0.f);
            const bool has = kk > 0.f;
            const float cand = lv - logf(has ? kk : 1.f);
            le[o] = has ? cand : 1e30f;
            tcp[o >> 2] |= has ? (unsigned)t << (8 * (o & 3)) : 0u;
        }
    const float* xn = x + (size_t)g.n * C * HW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool pair_store = (W & 1) == 0;
    float st[kCH][kTileLoads];
    tail_stage_load(xn, HW, 0, C, g, st);
    tail_stage_store(tiles[0], g, st);
    __syncthreads();
    for (int c0 = 0, kb = 0; c0 < C; c0 += kCH, kb ^= 1) {
        const bool more = c0 + kCH < C;
        if (more) tail_stage_load(xn, HW, c0 + kCH, C, g, st);
        for (int sl = 0; sl < kCH && c0 + sl < C; ++sl) {
            const int c = c0 + sl;
            float cur[4][4];
            tail_read_patch(tiles[kb] + sl * kTile, g, cur);
            const float* cf = coef[c];
            const float kc = cw[c] * gs;          // k of the outputs whose target is this channel
            const unsigned cc = (unsigned)(c + 1);
            // frame columns in pairs (2k, 2k + 1): one window, the coefficient pairs P[(ar, wr, wc)] = (ac = 1, ac = 0)
            const tail_f2* P = reinterpret_cast<const tail_f2*>(cf);
            const float bias = cf[16];
            tail_f2 dxa2[2][2];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int q = 0; q < 2; ++q) dxa2[p][q] = tail_f2{0.f, 0.f};
            float acc[10];
#pragma unroll
            for (int q = 0; q < 10; ++q) acc[q] = 0.f;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                const int ar = (a + 1) & 1, r0 = a >> 1;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int o0 = a * 6 + 2 * k, o1 = o0 + 1;
                    // the same chain as tail_logit, two outputs at a time
                    tail_f2 l2 = tail_f2{bias, bias};
                    l2 += P[(ar * 2 + 0) * 2 + 0] * cur[r0][k];
                    l2 += P[(ar * 2 + 0) * 2 + 1] * cur[r0][k + 1];
                    l2 += P[(ar * 2 + 1) * 2 + 0] * cur[r0 + 1][k];
                    l2 += P[(ar * 2 + 1) * 2 + 1] * cur[r0 + 1][k + 1];
                    const float hot0 = ((tcp[o0 >> 2] >> (8 * (o0 & 3))) & 0xffu) == cc ? kc : 0.f;
                    const float hot1 = ((tcp[o1 >> 2] >> (8 * (o1 & 3))) & 0xffu) == cc ? kc : 0.f;
                    const tail_f2 ex = l2 - tail_f2{le[o0], le[o1]};
                    const tail_f2 dl2 = tail_f2{__expf(ex.x), __expf(ex.y)} - tail_f2{hot0, hot1};
                    // the window's pixels that belong to the block: patch (r0 + wr, k + wc) = block pixel (.. - 1)
#pragma unroll
                    for (int wr = 0; wr < 2; ++wr)
#pragma unroll
                        for (int wc = 0; wc < 2; ++wc) {
                            const int p = r0 + wr - 1, q = k + wc - 1;
                            if (p >= 0 && p < 2 && q >= 0 && q < 2) dxa2[p][q] += dl2 * P[(ar * 2 + wr) * 2 + wc];
                        }
#pragma unroll
                    for (int h = 0; h < 2; ++h) {                // the lane's own outputs
                        const int b = 2 * k + h;
                        const float dl = h == 0 ? dl2.x : dl2.y;
                        if (a >= 1 && a <= 4 && b >= 1 && b <= 4) {
                            acc[9] += dl;
                            // tap (r, q) of output (a, b) reads up-sampled row 2*i0 + a + r - 2 = patch row (a + r) >> 1
                            // (these 144 FMAs stay scalar: their operand pairs are not register-aligned — packing them was
                            //  built and bought nothing, the pairs cost as many moves as they save issue slots)
#pragma unroll
                            for (int r = 0; r < 3; ++r)
#pragma unroll
                                for (int q = 0; q < 3; ++q)
                                    acc[r * 3 + q] = fmaf(dl, cur[(a + r) >> 1][(b + q) >> 1], acc[r * 3 + q]);
                        }
                    }
                }
            }
            float dxa[2][2];
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int q = 0; q < 2; ++q) dxa[p][q] = dxa2[p][q].x + dxa2[p][q].y;
            float* dc = dx + ((size_t)g.n * C + c) * HW;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                if (g.i0 + p < H && g.j0 < W) {
                    float* row = dc + (size_t)(g.i0 + p) * W + g.j0;
                    if (pair_store) *reinterpret_cast<float2*>(row) = make_float2(dxa[p][0], dxa[p][1]);
                    else {
                        row[0] = dxa[p][0];
                        if (g.j0 + 1 < W) row[1] = dxa[p][1];
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < 10; ++q) {
                const float t = wave_sum_to_lane63(acc[q]);
                if (lane == 63) wpart[wave][c][q] = t;
            }
        }
        if (more) tail_stage_store(tiles[kb ^ 1], g, st);
        __syncthreads();
    }
    for (int e = threadIdx.x; e < C * 10; e += 256) {
        const int c = e / 10, q = e - c * 10;
        part[(size_t)blockIdx.x * C * 10 + e] = ((wpart[0][c][q] + wpart[1][c][q]) + wpart[2][c][q]) + wpart[3][c][q];
    }
}

// out[g][col] = sum of rows [g*rpb, min(rows, (g+1)*rpb)) of in[rows][cols], rows in ascending order (deterministic).
// final != 0: cols = C*10 and the single output row is split into dw[c*9 + q] (q < 9) and db[c] (q == 9).
__global__ void __launch_bounds__(256) tail_colsum_kernel(const float* __restrict__ in, int rows, int cols, int rpb,
                                                          float* __restrict__ out, float* __restrict__ dw,
                                                          float* __restrict__ db, int final) {
    const int r0 = blockIdx.x * rpb, r1 = min(rows, r0 + rpb);
    for (int col = threadIdx.x; col < cols; col += 256) {
        float s = 0.f;
        for (int r = r0; r < r1; ++r) s += in[(size_t)r * cols + col];
        if (!final) out[(size_t)blockIdx.x * cols + col] = s;
        else {
            const int c = col / 10, q = col - c * 10;
            if (q < 9) dw[c * 9 + q] = s;
            else db[c] = s;
        }
    }
}

}  // namespace dynmm

using namespace dynmm;

constexpr int kTailRowsPerBlock = 64;

// one workgroup per 16 x 64 tile of input pixels
static size_t tail_blocks(int N, int H, int W) {
    return (size_t)N * ((H + kTR - 1) / kTR) * ((W + kTC - 1) / kTC);
}

extern "C" int dynmm_up2ce_fwd(const float* x, const float* w, const float* b, const unsigned char* target,
                               const float* cw, float* lse, double* loss_sum_wsum, int N, int C, int H, int W,
                               int acc_is_zero, void* stream) {
    (void)hipGetLastError();
    if (!x || !w || !target || !cw || !lse || !loss_sum_wsum || N <= 0 || C <= 0 || C > kTailMaxC || H <= 0 || W <= 0)
        return DYNMM_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (!acc_is_zero) DYNMM_HIP_TRY(hipMemsetAsync(loss_sum_wsum, 0, 2 * sizeof(double), st));
    const size_t blocks = tail_blocks(N, H, W);
    if (blocks > 0x7fffffffu) return DYNMM_EUNSUPPORTED;
    hipLaunchKernelGGL(up2ce_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, w, b, target, cw, lse,
                       loss_sum_wsum, N, C, H, W);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}


extern "C" size_t dynmm_up2ce_bwd_workspace_bytes(int N, int C, int H, int W) {
    if (N <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    const size_t blocks = tail_blocks(N, H, W);
    const size_t groups = (blocks + kTailRowsPerBlock - 1) / kTailRowsPerBlock;
    return sizeof(float) * (blocks + groups) * (size_t)C * 10;
}

extern "C" int dynmm_up2ce_bwd(const float* x, const float* w, const float* b, const unsigned char* target,
                               const float* cw, const float* lse, const float* gscale, float* dx, float* dw, float* db,
                               float* workspace, int N, int C, int H, int W, void* stream) {
    (void)hipGetLastError();
    if (!x || !w || !target || !cw || !lse || !gscale || !dx || !dw || !db || !workspace || N <= 0 || C <= 0 ||
        C > kTailMaxC || H <= 0 || W <= 0)
        return DYNMM_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const size_t blocks = tail_blocks(N, H, W);
    if (blocks > 0x7fffffffu) return DYNMM_EUNSUPPORTED;
    const int cols = C * 10;
    const int groups = (int)((blocks + kTailRowsPerBlock - 1) / kTailRowsPerBlock);
    float* part = workspace;
    float* part2 = workspace + blocks * (size_t)cols;
    hipLaunchKernelGGL(up2ce_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, w, b, target, cw, lse, gscale, dx,
                       part, N, C, H, W);
    DYNMM_LAUNCH_CHECK();
    // two-level ordered column sums of the per-workgroup partial rows (no atomics => bit-reproducible dw / db)
    hipLaunchKernelGGL(tail_colsum_kernel, dim3(groups), dim3(256), 0, st, part, (int)blocks, cols, kTailRowsPerBlock,
                       part2, (float*)nullptr, (float*)nullptr, 0);
    DYNMM_LAUNCH_CHECK();
    hipLaunchKernelGGL(tail_colsum_kernel, dim3(1), dim3(256), 0, st, part2, groups, cols, groups, (float*)nullptr, dw,
                       db, 1);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}
