// HBM-bound streaming kernels of the hot path: max-pool, adaptive average pool, nearest resize into
// a concat buffer (PPM), learned 2x upsample (nearest + depthwise 3x3 [+ skip add]), global average
// pool, and the per-(n,c)-coefficient blend  out = a*rgb + b*depth  that implements SE fusion and
// the gate blend in ONE pass over the feature maps (instead of the reference's 6 ATen launches per
// site).  One workgroup owns (a chunk of) one NCHW plane; lanes walk the
// contiguous HW axis with 16-byte accesses wherever size/alignment allow.
#include "common.h"
#include <initializer_list>
#include "vec.h"

namespace dynmm {

// v where keep, +0.0 otherwise — as a bit mask on the loaded value.  (`keep ? v : 0.f` lets the compiler sink the LOAD of v under
// the select's arm again: a predicated load with its own wait, which is what the unconditional clamped address was for.)
__device__ __forceinline__ float keep_if(float v, bool keep) { return __int_as_float(__float_as_int(v) & -(int)keep); }


static inline int plane_chunks(int HW, int chunk) { return (HW + chunk - 1) / chunk; }
constexpr int kChunk = 8192;

// ------------------------------------------------------------------------------------------------
// F.max_pool2d(kernel 3, stride 2, pad 1): first maximum in row-major window order wins (strict >),
// NaN propagates — the tie rule matters because the inputs are post-ReLU (many exact zeros).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) maxpool_fwd_kernel(const float* __restrict__ x,
                                                          float* __restrict__ y,
                                                          signed char* __restrict__ idx, int H,
                                                          int W, int Ho, int Wo) {
    const size_t plane = blockIdx.x;
    const float* xp = x + plane * H * W;
    const int HoWo = Ho * Wo;
    const int beg = blockIdx.y * kChunk, end = min(HoWo, beg + kChunk);
    for (int i = beg + threadIdx.x; i < end; i += 256) {
        const int oh = i / Wo, ow = i - oh * Wo;
        float best = -INFINITY;
        int bi = -1;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int ih = 2 * oh - 1 + r;
            if (ih < 0 || ih >= H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int iw = 2 * ow - 1 + s;
                if (iw < 0 || iw >= W) continue;
                const float v = xp[ih * W + iw];
                if (bi < 0 || v > best || v != v) { best = v; bi = r * 3 + s; }
            }
        }
        y[plane * HoWo + i] = best;
        if (idx) idx[plane * HoWo + i] = (signed char)bi;
    }
}

__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const float* __restrict__ g,
                                                          const signed char* __restrict__ idx,
                                                          float* __restrict__ dx, int H, int W,
                                                          int Ho, int Wo) {
    const size_t plane = blockIdx.x;
    const float* gp = g + plane * Ho * Wo;
    const signed char* ip = idx + plane * Ho * Wo;
    const int HW = H * W;
    const int beg = blockIdx.y * kChunk, end = min(HW, beg + kChunk);
    for (int i = beg + threadIdx.x; i < end; i += 256) {
        const int ih = i / W, iw = i - ih * W;
        float acc = 0.f;
        const int oh_lo = ih / 2, oh_hi = min(Ho - 1, (ih + 1) / 2);
        const int ow_lo = iw / 2, ow_hi = min(Wo - 1, (iw + 1) / 2);
        for (int oh = oh_lo; oh <= oh_hi; ++oh)
            for (int ow = ow_lo; ow <= ow_hi; ++ow) {
                const int r = ih - (2 * oh - 1), s = iw - (2 * ow - 1);
                if (ip[oh * Wo + ow] == r * 3 + s) acc += gp[oh * Wo + ow];
            }
        dx[plane * HW + i] = acc;
    }
}

// W % 4 == 0: four consecutive input pixels per thread (one 16-byte store); their windows are the
// 2 pooled rows x 3 pooled columns around them, fetched once (the scalar version above did a divide,
// up to 4 byte-gathers and a 4-byte store per pixel: 738 us on 32x64x240x320 vs ~190 us of HBM time).
__global__ void __launch_bounds__(256) maxpool_bwd4_kernel(const float* __restrict__ g,
                                                           const signed char* __restrict__ idx,
                                                           float* __restrict__ dx, int H, int W,
                                                           int Ho, int Wo) {
    const size_t plane = blockIdx.x;
    const float* gp = g + plane * Ho * Wo;
    const signed char* ip = idx + plane * Ho * Wo;
    float* dp = dx + plane * (size_t)H * W;
    const int Wq = W / 4, nq = H * Wq;
    const int beg = blockIdx.y * (kChunk / 4), end = min(nq, beg + kChunk / 4);
    for (int q = beg + threadIdx.x; q < end; q += 256) {
        const int ih = q / Wq, iw0 = (q - ih * Wq) * 4;
        const int oh0 = ih / 2, oh1 = min(Ho - 1, (ih + 1) / 2);      // the (at most) two pooled rows
        const int ow0 = iw0 / 2;                                      // pooled columns ow0 .. ow0+2
        float out[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int oh = rr ? oh1 : oh0;
            if (rr && oh1 == oh0) break;
            const int r = ih - (2 * oh - 1);                          // tap row of this input row in window oh
            if (r < 0 || r > 2) continue;
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) {
                const int ow = ow0 + cc;
                if (ow >= Wo) continue;
                const int code = ip[oh * Wo + ow];
                const float gv = gp[oh * Wo + ow];
                // window ow covers input columns 2*ow-1 .. 2*ow+1, i.e. local pixels 2*cc-1 .. 2*cc+1
                const int s = code - r * 3;                           // tap column the max came from, if in row r
                const int local = 2 * cc - 1 + s;
                if (s >= 0 && s <= 2 && local >= 0 && local < 4) {
                    if (local == 0) out[0] += gv;
                    else if (local == 1) out[1] += gv;
                    else if (local == 2) out[2] += gv;
                    else out[3] += gv;
                }
            }
        }
        *reinterpret_cast<float4*>(dp + (size_t)ih * W + iw0) = make_float4(out[0], out[1], out[2], out[3]);
    }
}

// Even H, W % 8 == 0 (the stems' 240x320 maps): wide versions.  Round 1's kernels issued 9 scalar loads per pooled
// pixel (forward: 2.4 TB/s) / 12 scalar + byte gathers per 16-byte store (backward: 1.9 TB/s).
// Forward: a thread owns 4 pooled pixels of one row = input columns 8t-1 .. 8t+7: one scalar + two 16-byte loads per
// input row, one 16-byte store + one 4-byte index store.  Same tie rule, tap by tap in row-major window order.
__global__ void __launch_bounds__(256) maxpool_fwd4_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                           signed char* __restrict__ idx, int H, int W, int Ho,
                                                           int Wo) {
    const size_t plane = blockIdx.x;
    const float* xp = x + plane * (size_t)H * W;
    const int Wq = Wo / 4, nq = Ho * Wq;
    const int beg = blockIdx.y * (kChunk / 4), end = min(nq, beg + kChunk / 4);
    for (int q = beg + threadIdx.x; q < end; q += 256) {
        const int oh = q / Wq, t = q - oh * Wq;
        float best[4];
        int bi[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { best[j] = -INFINITY; bi[j] = -1; }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int ih = 2 * oh - 1 + r;                     // < H always (H even); -1 for the first pooled row
            if (ih < 0) continue;
            const float* row = xp + (size_t)ih * W + 8 * t;
            const float4 a = *reinterpret_cast<const float4*>(row);
            const float4 b = *reinterpret_cast<const float4*>(row + 4);
            const float left = t > 0 ? row[-1] : 0.f;
            const float v[9] = {left, a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) {
                    if (j == 0 && sx == 0 && t == 0) continue;          // input column -1
                    const float val = v[2 * j + sx];
                    if (bi[j] < 0 || val > best[j] || val != val) { best[j] = val; bi[j] = r * 3 + sx; }
                }
        }
        const size_t o = plane * (size_t)Ho * Wo + (size_t)oh * Wo + 4 * t;
        *reinterpret_cast<float4*>(y + o) = make_float4(best[0], best[1], best[2], best[3]);
        if (idx) {
            const unsigned pk = (unsigned)(bi[0] & 0xff) | ((unsigned)(bi[1] & 0xff) << 8) |
                                ((unsigned)(bi[2] & 0xff) << 16) | ((unsigned)(bi[3] & 0xff) << 24);
            *reinterpret_cast<unsigned*>(idx + o) = pk;
        }
    }
}

// Backward: a thread owns input rows 2a, 2a+1 x columns 8t .. 8t+7 = four 2x2 blocks; the block at pooled position
// (a, m) collects from the windows (a, m), (a, m+1), (a+1, m), (a+1, m+1) by their arg-max code:
//   (2a, 2m): code 4 of w(a,m)            (2a,   2m+1): 5 of w(a,m), 3 of w(a,m+1)
//   (2a+1, 2m): 7 of w(a,m), 1 of w(a+1,m)    (2a+1, 2m+1): 8 of w(a,m), 6 of w(a,m+1), 2 of w(a+1,m), 0 of w(a+1,m+1)
// i.e. 2 pooled rows x 5 pooled columns of (gradient, code), fetched as one 16-byte + one scalar load each.
// gradient of the 2 x 8 input pixels (rows 2a, 2a+1; columns 8t .. 8t+7) of one plane from the pooled gradient gp and
// the arg-max codes ip: top[] = row 2a, bot[] = row 2a+1
__device__ __forceinline__ void pool_bwd_block(const float* __restrict__ gp, const signed char* __restrict__ ip, int a,
                                               int t, int Ho, int Wo, float (&top)[8], float (&bot)[8]) {
    float gv[2][5];
    int cd[2][5];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const bool rok = a + rr < Ho;
        const size_t o = (size_t)(rok ? a + rr : a) * Wo + 4 * t;
        const float4 g4 = *reinterpret_cast<const float4*>(gp + o);
        const unsigned c4 = *reinterpret_cast<const unsigned*>(ip + o);
        const bool cok = 4 * t + 4 < Wo;
        const float g5 = cok ? gp[o + 4] : 0.f;
        const int c5 = cok ? (int)ip[o + 4] : -1;
        gv[rr][0] = g4.x; gv[rr][1] = g4.y; gv[rr][2] = g4.z; gv[rr][3] = g4.w; gv[rr][4] = g5;
#pragma unroll
        for (int j = 0; j < 4; ++j) cd[rr][j] = rok ? (int)(signed char)((c4 >> (8 * j)) & 0xffu) : -1;
        cd[rr][4] = rok ? c5 : -1;
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const float g00 = gv[0][m], g01 = gv[0][m + 1], g10 = gv[1][m], g11 = gv[1][m + 1];
        const int c00 = cd[0][m], c01 = cd[0][m + 1], c10 = cd[1][m], c11 = cd[1][m + 1];
        top[2 * m] = c00 == 4 ? g00 : 0.f;
        top[2 * m + 1] = (c00 == 5 ? g00 : 0.f) + (c01 == 3 ? g01 : 0.f);
        bot[2 * m] = (c00 == 7 ? g00 : 0.f) + (c10 == 1 ? g10 : 0.f);
        bot[2 * m + 1] = ((c00 == 8 ? g00 : 0.f) + (c01 == 6 ? g01 : 0.f)) + ((c10 == 2 ? g10 : 0.f) + (c11 == 0 ? g11 : 0.f));
    }
}

__global__ void __launch_bounds__(256) maxpool_bwd8_kernel(const float* __restrict__ g,
                                                           const signed char* __restrict__ idx,
                                                           float* __restrict__ dx, int H, int W, int Ho, int Wo) {
    const size_t plane = blockIdx.x;
    const float* gp = g + plane * (size_t)Ho * Wo;
    const signed char* ip = idx + plane * (size_t)Ho * Wo;
    float* dp = dx + plane * (size_t)H * W;
    const int Wq = Wo / 4, nq = Ho * Wq;
    const int beg = blockIdx.y * (kChunk / 4), end = min(nq, beg + kChunk / 4);
    for (int q = beg + threadIdx.x; q < end; q += 256) {
        const int a = q / Wq, t = q - a * Wq;
        float top[8], bot[8];
        pool_bwd_block(gp, ip, a, t, Ho, Wo, top, bot);
        float* r0 = dp + (size_t)(2 * a) * W + 8 * t;
        *reinterpret_cast<float4*>(r0) = make_float4(top[0], top[1], top[2], top[3]);
        *reinterpret_cast<float4*>(r0 + 4) = make_float4(top[4], top[5], top[6], top[7]);
        *reinterpret_cast<float4*>(r0 + W) = make_float4(bot[0], bot[1], bot[2], bot[3]);
        *reinterpret_cast<float4*>(r0 + W + 4) = make_float4(bot[4], bot[5], bot[6], bot[7]);
    }
}

// ------------------------------------------------------------------------------------------------
// Stem fusion + both max-pools as one forward pass and two backward passes (…globalgate.py:258-261: fuse =
// se_layer0(rgb, depth); rgb = max_pool(fuse); depth = max_pool(depth)).  The full-resolution fused map has no other
// consumer, so it is never written: forward reads rgb / depth once and writes the two pooled maps (+ arg-max codes);
// backward re-derives d(fuse) and d(pooled depth) block by block from the pooled gradients and the codes.  Unfused,
// the same work is 7 passes over 629 MB tensors forward and 12 backward (batch 32).  Even H, W % 8 == 0 only.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void pool_update(float& best, int& bi, float val, int code) {
    if (bi < 0 || val > best || val != val) { best = val; bi = code; }
}

__global__ void __launch_bounds__(256) axpby_pool_fwd_kernel(const float* __restrict__ xr, const float* __restrict__ xd,
                                                             const float* __restrict__ ca, const float* __restrict__ cb,
                                                             float* __restrict__ yo, signed char* __restrict__ io,
                                                             float* __restrict__ yd, signed char* __restrict__ id, int H,
                                                             int W, int Ho, int Wo, const float* __restrict__ tr, int C) {
    const size_t plane = blockIdx.x;
    const float* rp = xr + plane * (size_t)H * W;
    const float* dp = xd + plane * (size_t)H * W;
    const float fa = ca[plane], fb = cb[plane];
    // tr = [4][C] {scale_r, shift_r, scale_d, shift_d}: xr / xd are the un-normalised stem conv outputs, BatchNorm +
    // ReLU applied on load (gap2_kernel)
    const int ch = tr ? (int)(plane % C) : 0;
    const float scr = tr ? tr[ch] : 1.f, shr = tr ? tr[C + ch] : 0.f;
    const float scd = tr ? tr[2 * C + ch] : 1.f, shd = tr ? tr[3 * C + ch] : 0.f;
    const int Wq = Wo / 4, nq = Ho * Wq;
    const int beg = blockIdx.y * (kChunk / 4), end = min(nq, beg + kChunk / 4);
    for (int q = beg + threadIdx.x; q < end; q += 256) {
        const int oh = q / Wq, t = q - oh * Wq;
        float bo[4], bd[4];
        int co[4], cd[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { bo[j] = bd[j] = -INFINITY; co[j] = cd[j] = -1; }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int ih = 2 * oh - 1 + r;
            if (ih < 0) continue;
            const size_t o = (size_t)ih * W + 8 * t;
            const float4 r0 = *reinterpret_cast<const float4*>(rp + o), r1 = *reinterpret_cast<const float4*>(rp + o + 4);
            const float4 d0 = *reinterpret_cast<const float4*>(dp + o), d1 = *reinterpret_cast<const float4*>(dp + o + 4);
            const float rl = t > 0 ? rp[o - 1] : 0.f, dl = t > 0 ? dp[o - 1] : 0.f;
            float rv[9] = {rl, r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
            float dv[9] = {dl, d0.x, d0.y, d0.z, d0.w, d1.x, d1.y, d1.z, d1.w};
            if (tr) {
#pragma unroll
                for (int k = 0; k < 9; ++k) {
                    rv[k] = fmaxf(fmaf(rv[k], scr, shr), 0.f);
                    dv[k] = fmaxf(fmaf(dv[k], scd, shd), 0.f);
                }
            }
            float fv[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) fv[k] = fmaf(fa, rv[k], fb * dv[k]);     // axpby_fwd_kernel's expression
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int sx = 0; sx < 3; ++sx) {
                    if (j == 0 && sx == 0 && t == 0) continue;                  // input column -1
                    pool_update(bo[j], co[j], fv[2 * j + sx], r * 3 + sx);
                    pool_update(bd[j], cd[j], dv[2 * j + sx], r * 3 + sx);
                }
        }
        const size_t o = plane * (size_t)Ho * Wo + (size_t)oh * Wo + 4 * t;
        *reinterpret_cast<float4*>(yo + o) = make_float4(bo[0], bo[1], bo[2], bo[3]);
        *reinterpret_cast<float4*>(yd + o) = make_float4(bd[0], bd[1], bd[2], bd[3]);
        *reinterpret_cast<unsigned*>(io + o) = (unsigned)(co[0] & 0xff) | ((unsigned)(co[1] & 0xff) << 8) |
                                               ((unsigned)(co[2] & 0xff) << 16) | ((unsigned)(co[3] & 0xff) << 24);
        *reinterpret_cast<unsigned*>(id + o) = (unsigned)(cd[0] & 0xff) | ((unsigned)(cd[1] & 0xff) << 8) |
                                               ((unsigned)(cd[2] & 0xff) << 16) | ((unsigned)(cd[3] & 0xff) << 24);
    }
}

// da[plane] = sum d(fuse) * rgb ; db[plane] = sum d(fuse) * depth   (one workgroup per plane: fixed summation order)
__global__ void __launch_bounds__(256) axpby_pool_bwd_reduce_kernel(const float* __restrict__ go,
                                                                    const signed char* __restrict__ io,
                                                                    const float* __restrict__ xr,
                                                                    const float* __restrict__ xd, float* __restrict__ da,
                                                                    float* __restrict__ db, int H, int W, int Ho, int Wo,
                                                                    const float* __restrict__ tr, int C) {
    __shared__ float red[4];
    const size_t plane = blockIdx.x;
    const int ch = tr ? (int)(plane % C) : 0;
    const float scr = tr ? tr[ch] : 1.f, shr = tr ? tr[C + ch] : 0.f;
    const float scd = tr ? tr[2 * C + ch] : 1.f, shd = tr ? tr[3 * C + ch] : 0.f;
    const float* gp = go + plane * (size_t)Ho * Wo;
    const signed char* ip = io + plane * (size_t)Ho * Wo;
    const float* rp = xr + plane * (size_t)H * W;
    const float* dp = xd + plane * (size_t)H * W;
    const int Wq = Wo / 4, nq = Ho * Wq;
    float sa = 0.f, sb = 0.f;
    for (int q = threadIdx.x; q < nq; q += 256) {
        const int a = q / Wq, t = q - a * Wq;
        float top[8], bot[8];
        pool_bwd_block(gp, ip, a, t, Ho, Wo, top, bot);
        const size_t o = (size_t)(2 * a) * W + 8 * t;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float4 r0 = *reinterpret_cast<const float4*>(rp + o + 4 * h), r1 = *reinterpret_cast<const float4*>(rp + o + W + 4 * h);
            float4 d0 = *reinterpret_cast<const float4*>(dp + o + 4 * h), d1 = *reinterpret_cast<const float4*>(dp + o + W + 4 * h);
            if (tr) {
#define DYNMM_BNRELU4(v, sc, sh) \
    v.x = fmaxf(fmaf(v.x, sc, sh), 0.f); v.y = fmaxf(fmaf(v.y, sc, sh), 0.f); \
    v.z = fmaxf(fmaf(v.z, sc, sh), 0.f); v.w = fmaxf(fmaf(v.w, sc, sh), 0.f)
                DYNMM_BNRELU4(r0, scr, shr); DYNMM_BNRELU4(r1, scr, shr);
                DYNMM_BNRELU4(d0, scd, shd); DYNMM_BNRELU4(d1, scd, shd);
#undef DYNMM_BNRELU4
            }
            sa += top[4 * h] * r0.x + top[4 * h + 1] * r0.y + top[4 * h + 2] * r0.z + top[4 * h + 3] * r0.w;
            sa += bot[4 * h] * r1.x + bot[4 * h + 1] * r1.y + bot[4 * h + 2] * r1.z + bot[4 * h + 3] * r1.w;
            sb += top[4 * h] * d0.x + top[4 * h + 1] * d0.y + top[4 * h + 2] * d0.z + top[4 * h + 3] * d0.w;
            sb += bot[4 * h] * d1.x + bot[4 * h + 1] * d1.y + bot[4 * h + 2] * d1.z + bot[4 * h + 3] * d1.w;
        }
    }
    const float ta = block_reduce_sum_256<float>(sa, red);
    const float tb = block_reduce_sum_256<float>(sb, red);
    if (threadIdx.x == 0) { da[plane] = ta; db[plane] = tb; }
}

// d rgb = a * d(fuse) + ca*cscale ; d depth = b * d(fuse) + cb*cscale + d(pooled depth)
__global__ void __launch_bounds__(256) axpby_pool_bwd_apply_kernel(
    const float* __restrict__ go, const signed char* __restrict__ io, const float* __restrict__ gd,
    const signed char* __restrict__ id, const float* __restrict__ a, const float* __restrict__ b,
    const float* __restrict__ ca, const float* __restrict__ cb, float cscale, float* __restrict__ dxr,
    float* __restrict__ dxd, int H, int W, int Ho, int Wo) {
    const size_t plane = blockIdx.x;
    const size_t po = plane * (size_t)Ho * Wo;
    const float fa = a[plane], fb = b[plane];
    const float oa = ca ? ca[plane] * cscale : 0.f, ob = cb ? cb[plane] * cscale : 0.f;
    float* rp = dxr + plane * (size_t)H * W;
    float* dp = dxd + plane * (size_t)H * W;
    const int Wq = Wo / 4, nq = Ho * Wq;
    const int beg = blockIdx.y * (kChunk / 4), end = min(nq, beg + kChunk / 4);
    for (int q = beg + threadIdx.x; q < end; q += 256) {
        const int ar = q / Wq, t = q - ar * Wq;
        float ft[8], fbm[8], dt[8], dbm[8];
        pool_bwd_block(go + po, io + po, ar, t, Ho, Wo, ft, fbm);
        pool_bwd_block(gd + po, id + po, ar, t, Ho, Wo, dt, dbm);
        const size_t o = (size_t)(2 * ar) * W + 8 * t;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // (axpby_bwd_apply_kernel's expressions; the pooled-depth gradient is added last, as add_n did)
            *reinterpret_cast<float4*>(rp + o + 4 * h) = make_float4(fmaf(fa, ft[4 * h], oa), fmaf(fa, ft[4 * h + 1], oa),
                                                                      fmaf(fa, ft[4 * h + 2], oa), fmaf(fa, ft[4 * h + 3], oa));
            *reinterpret_cast<float4*>(rp + o + W + 4 * h) = make_float4(fmaf(fa, fbm[4 * h], oa), fmaf(fa, fbm[4 * h + 1], oa),
                                                                          fmaf(fa, fbm[4 * h + 2], oa), fmaf(fa, fbm[4 * h + 3], oa));
            *reinterpret_cast<float4*>(dp + o + 4 * h) =
                make_float4(fmaf(fb, ft[4 * h], ob) + dt[4 * h], fmaf(fb, ft[4 * h + 1], ob) + dt[4 * h + 1],
                            fmaf(fb, ft[4 * h + 2], ob) + dt[4 * h + 2], fmaf(fb, ft[4 * h + 3], ob) + dt[4 * h + 3]);
            *reinterpret_cast<float4*>(dp + o + W + 4 * h) =
                make_float4(fmaf(fb, fbm[4 * h], ob) + dbm[4 * h], fmaf(fb, fbm[4 * h + 1], ob) + dbm[4 * h + 1],
                            fmaf(fb, fbm[4 * h + 2], ob) + dbm[4 * h + 2], fmaf(fb, fbm[4 * h + 3], ob) + dbm[4 * h + 3]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// adaptive average pool (windows [floor(o*I/O), ceil((o+1)*I/O)) ) — tiny maps only (PPM).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int ap_start(int o, int I, int O) { return (o * I) / O; }
__device__ __forceinline__ int ap_end(int o, int I, int O) { return ((o + 1) * I + O - 1) / O; }

__global__ void __launch_bounds__(256) adaptive_avgpool_fwd_kernel(const float* __restrict__ x,
                                                                   float* __restrict__ y, int NC,
                                                                   int H, int W, int OH, int OW) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= NC * OH * OW) return;
    const int ow = i % OW, oh = (i / OW) % OH, p = i / (OW * OH);
    const int hs = ap_start(oh, H, OH), he = ap_end(oh, H, OH);
    const int ws = ap_start(ow, W, OW), we = ap_end(ow, W, OW);
    const float* xp = x + (size_t)p * H * W;
    float s = 0.f;
    for (int h = hs; h < he; ++h)
        for (int w = ws; w < we; ++w) s += xp[h * W + w];
    y[i] = s / (float)((he - hs) * (we - ws));
}

// The window bounds of every output row / column are tabulated once per workgroup (they cost two integer divisions each, and
// the per-element loops below used to evaluate them OH * OW times per input element: 56 us per launch on the 15 x 20 map of the
// pyramid pooling module — round 6); same visiting order, same arithmetic per term: bit-identical.
constexpr int kApMaxBins = 64;

__global__ void __launch_bounds__(256) adaptive_avgpool_bwd_kernel(const float* __restrict__ g,
                                                                   float* __restrict__ dx, int NC,
                                                                   int H, int W, int OH, int OW) {
    __shared__ int hs_[kApMaxBins], he_[kApMaxBins], ws_[kApMaxBins], we_[kApMaxBins];
    for (int o = threadIdx.x; o < OH; o += 256) { hs_[o] = ap_start(o, H, OH); he_[o] = ap_end(o, H, OH); }
    for (int o = threadIdx.x; o < OW; o += 256) { ws_[o] = ap_start(o, W, OW); we_[o] = ap_end(o, W, OW); }
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= NC * H * W) return;
    const int w = i % W, h = (i / W) % H, p = i / (W * H);
    const float* gp = g + (size_t)p * OH * OW;
    float s = 0.f;
    for (int oh = 0; oh < OH; ++oh) {
        const int hs = hs_[oh], he = he_[oh];
        if (h < hs || h >= he) continue;
        for (int ow = 0; ow < OW; ++ow) {
            const int ws = ws_[ow], we = we_[ow];
            if (w < ws || w >= we) continue;
            s += gp[oh * OW + ow] / (float)((he - hs) * (we - ws));
        }
    }
    dx[i] = s;
}

// ------------------------------------------------------------------------------------------------
// nearest resize of y[N,C,h,w] into channels [c_off, c_off+C) of out[N,Ctot,H,W]  (PPM cat).
// src index = min(floor(dst * (float)in/out), in-1), exactly ATen's nearest rule.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int nearest_src(int dst, int in, int out) {
    const float scale = (float)in / (float)out;
    const int s = (int)floorf((float)dst * scale);
    return s < in - 1 ? s : in - 1;
}

__global__ void __launch_bounds__(256) nearest_into_fwd_kernel(const float* __restrict__ y,
                                                               float* __restrict__ out, int N, int C,
                                                               int h, int w, int Ctot, int c_off,
                                                               int H, int W) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N * C * H * W) return;
    const int x = i % W, yy = (i / W) % H, c = (i / (W * H)) % C, n = i / (W * H * C);
    const int sy = nearest_src(yy, h, H), sx = nearest_src(x, w, W);
    out[(((size_t)n * Ctot + c_off + c) * H + yy) * W + x] = y[(((size_t)n * C + c) * h + sy) * w + sx];
}

// Backward: dy[p][q] = sum of g over the destination pixels whose source is (p, q) — a contiguous rectangle (the source index is
// monotone in the destination index).  L lanes share one output element: they find the rectangle (H + W evaluations of the
// source rule instead of H * W), walk it L elements at a time (coalesced) and meet in a fixed xor-butterfly.  Round 6: one lane
// per output looped over the whole H x W plane with the rule evaluated per pixel — 69 / 74 us per launch for the two pooled
// branches of the pyramid pooling module, exposed at the head of the encoder's backward.
template <int L>
__global__ void __launch_bounds__(256) nearest_into_bwd_kernel(const float* __restrict__ g,
                                                               float* __restrict__ dy, int N, int C,
                                                               int h, int w, int Ctot, int c_off,
                                                               int H, int W) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int i = t / L, l = t % L;
    const bool live = i < N * C * h * w;
    const int ii = live ? i : 0;
    const int q = ii % w, p = (ii / w) % h, c = (ii / (w * h)) % C, n = ii / (w * h * C);
    const float* gp = g + ((size_t)n * Ctot + c_off + c) * H * W;
    if (h == H && w == W) {   // identity resize: plain strided copy (the `x` branch of the PPM cat); launched with L = 1
        if (live) dy[i] = gp[p * W + q];
        return;
    }
    int y0 = H, y1 = 0, x0 = W, x1 = 0;
    for (int yy = 0; yy < H; ++yy)
        if (nearest_src(yy, h, H) == p) { y0 = min(y0, yy); y1 = yy + 1; }
    for (int x = 0; x < W; ++x)
        if (nearest_src(x, w, W) == q) { x0 = min(x0, x); x1 = x + 1; }
    const int rw = max(x1 - x0, 0), cnt = rw * max(y1 - y0, 0);
    float s = 0.f;
    for (int e = l; e < cnt; e += L) {
        const int r = e / rw;
        s += gp[(y0 + r) * W + x0 + (e - r * rw)];
    }
#pragma unroll
    for (int o = L / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, L);
    if (live && l == 0) dy[i] = s;
}

// ------------------------------------------------------------------------------------------------
// learned 2x upsample: y = dwconv3x3(nearest2x(x)) + bias (+ skip)   — write-bandwidth bound.
//
// nearest-2x followed by a zero-padded 3x3 collapses to a 2x2 stencil on the INPUT with pre-summed
// taps: output row 2i   reads input rows {i-1: w[0],      i: w[1]+w[2]},
//       output row 2i+1 reads input rows {i:   w[0]+w[1], i+1: w[2]}      (same along columns;
// out-of-range input rows/cols contribute 0, which is exactly the zero padding of the upsampled
// map).  One lane owns 2 input pixels = a 2x4 output patch: 12 input loads, 32 FMAs, two 16-byte
// stores, instead of 9 loads + index tests per output.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) upsample_fwd_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ wgt,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ skip,
                                                           float* __restrict__ y, int C, int H,
                                                           int W) {
    const size_t plane = blockIdx.x;
    const int c = (int)(plane % C);
    const int W2 = 2 * W;
    const float* xp = x + plane * H * W;
    float* yp = y + plane * 4 * H * W;
    const float* sp = skip ? skip + plane * 4 * H * W : nullptr;
    float k[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) k[j] = wgt[c * 9 + j];
    const float b = bias ? bias[c] : 0.f;
    // row/col tap groups: [a][0] multiplies the "previous/this" input, [a][1] the "this/next" one
    // a = 0 (even output): {w0, w1+w2};  a = 1 (odd output): {w0+w1, w2}
    float rc[2][2][2][2];   // [a_row][which_row][a_col][which_col]
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
                    rc[ar][wr][ac][wc] = s;
                }
    const int Wh = (W + 1) / 2;                 // lane-owned column pairs per input row
    const int items = H * Wh;
    const int beg = blockIdx.y * (kChunk / 8), end = min(items, beg + kChunk / 8);
    for (int it = beg + threadIdx.x; it < end; it += 256) {
        const int i = it / Wh, j0 = (it - i * Wh) * 2;
        // 3 x 4 input neighbourhood: rows i-1..i+1, cols j0-1..j0+2 (zero outside)
        // (the unconditional-clamped-load form of this gather was measured: 28.6 / 78.9 / 155.2 / 171.3 -> 28.9 / 81.0 / 156.6 / 170.5 us
        //  at the four decoder shapes — neutral; the launch is bound by its 16-byte stores and the skip read)
        float v[3][4];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int ii = i + r - 1;
            const bool rok = ii >= 0 && ii < H;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int jj = j0 + q - 1;
                v[r][q] = (rok && jj >= 0 && jj < W) ? xp[ii * W + jj] : 0.f;
            }
        }
        const bool two = j0 + 1 < W;
#pragma unroll
        for (int ar = 0; ar < 2; ++ar) {
            float o[4];
#pragma unroll
            for (int pc = 0; pc < 2; ++pc)          // which of the lane's 2 input columns
#pragma unroll
                for (int ac = 0; ac < 2; ++ac) {
                    // input rows: ar==0 -> (i-1, i) = v[0], v[1]; ar==1 -> (i, i+1) = v[1], v[2]
                    // input cols: ac==0 -> (j-1, j);            ac==1 -> (j, j+1)
                    const int r0 = ar, c0 = pc + ac;
                    o[pc * 2 + ac] = b + rc[ar][0][ac][0] * v[r0][c0] + rc[ar][0][ac][1] * v[r0][c0 + 1] +
                                     rc[ar][1][ac][0] * v[r0 + 1][c0] + rc[ar][1][ac][1] * v[r0 + 1][c0 + 1];
                }
            const size_t off = (size_t)(2 * i + ar) * W2 + 2 * j0;
            if (two && (W2 % 4 == 0)) {
                if (sp) {
                    const float4 sk = *reinterpret_cast<const float4*>(sp + off);
                    o[0] += sk.x; o[1] += sk.y; o[2] += sk.z; o[3] += sk.w;
                }
                *reinterpret_cast<float4*>(yp + off) = make_float4(o[0], o[1], o[2], o[3]);
            } else {
                const int nout = two ? 4 : 2;
                for (int q = 0; q < nout; ++q) yp[off + q] = o[q] + (sp ? sp[off + q] : 0.f);
            }
        }
    }
}

// dx[i][j] = sum over the 4x4 output patch rows 2i-1..2i+2, cols 2j-1..2j+2 with the transposed
// pre-summed taps (same grouping as above, seen from the input pixel).
__global__ void __launch_bounds__(256) upsample_bwd_dx_kernel(const float* __restrict__ g,
                                                              const float* __restrict__ wgt,
                                                              float* __restrict__ dx, int C, int H,
                                                              int W, int pairs) {
    const size_t plane = blockIdx.x;
    const int c = (int)(plane % C);
    const int H2 = 2 * H, W2 = 2 * W;
    const float* gp = g + plane * H2 * W2;
    float k[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) k[j] = wgt[c * 9 + j];
    // effective 4x4 stencil on g: taps a,b in {-1,0,1,2}; R(-1)={2}, R(0)={1,2}, R(1)={0,1}, R(2)={0}
    float e[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const bool rin = (a == 0) ? (r == 2) : (a == 1) ? (r >= 1) : (a == 2) ? (r <= 1) : (r == 0);
                if (!rin) continue;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const bool qin = (bb == 0) ? (q == 2) : (bb == 1) ? (q >= 1) : (bb == 2) ? (q <= 1) : (q == 0);
                    if (qin) s += k[r * 3 + q];
                }
            }
            e[a][bb] = s;
        }
    const int HW = H * W;
    if (pairs) {
        // two adjacent input pixels per lane: their 4x6 patch of g is one 16-byte load + two edge scalars per
        // row, and the result is one 8-byte store (half the load instructions of the one-pixel form below)
        const int Wh = W / 2, items = H * Wh;
        const int beg = blockIdx.y * (kChunk / 2), end = min(items, beg + kChunk / 2);
        for (int it = beg + threadIdx.x; it < end; it += 256) {
            const int ih = it / Wh, iw = (it - ih * Wh) * 2;
            float s0 = 0.f, s1 = 0.f;
            // (the 4 x 3 loads of the patch are unconditional on clamped coordinates — in flight together — and a row / edge
            //  column outside the map is zeroed afterwards; same products in the same order)
            float4 m[4];
            float lft[4], rgt[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int oh = 2 * ih + a - 1;
                const float* row = gp + (size_t)min(max(oh, 0), H2 - 1) * W2;
                m[a] = *reinterpret_cast<const float4*>(row + 2 * iw);                 // cols 2iw .. 2iw+3
                lft[a] = row[max(2 * iw - 1, 0)];
                rgt[a] = row[min(2 * iw + 4, W2 - 1)];
            }
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int oh = 2 * ih + a - 1;
                const bool rok = oh >= 0 && oh < H2;                                   // a row outside the map adds exact zeros
                const float mx = keep_if(m[a].x, rok), my = keep_if(m[a].y, rok), mz = keep_if(m[a].z, rok), mw = keep_if(m[a].w, rok);
                const float l = keep_if(lft[a], rok && iw > 0), r = keep_if(rgt[a], rok && iw + 2 < W);
                s0 += e[a][0] * l + e[a][1] * mx + e[a][2] * my + e[a][3] * mz;
                s1 += e[a][0] * my + e[a][1] * mz + e[a][2] * mw + e[a][3] * r;
            }
            *reinterpret_cast<float2*>(dx + plane * HW + (size_t)ih * W + iw) = make_float2(s0, s1);
        }
        return;
    }
    const int beg = blockIdx.y * kChunk, end = min(HW, beg + kChunk);
    for (int i = beg + threadIdx.x; i < end; i += 256) {
        const int ih = i / W, iw = i - ih * W;
        float s = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int oh = 2 * ih + a - 1;
            if (oh < 0 || oh >= H2) continue;
            const float* row = gp + (size_t)oh * W2;
            // the two centre columns 2iw, 2iw+1 are always in range and 8-byte aligned
            const float2 mid = *reinterpret_cast<const float2*>(row + 2 * iw);
            s += e[a][1] * mid.x + e[a][2] * mid.y;
            if (iw > 0) s += e[a][0] * row[2 * iw - 1];
            if (iw < W - 1) s += e[a][3] * row[2 * iw + 2];
        }
        dx[plane * HW + i] = s;
    }
}

// part_w[split][c][r][s] = sum g*U ; part_b[split][c] = sum g over this split's samples.  grid (C, splits
// over n); the splits are summed in a fixed order by launch_reduce_slabs (no float atomics).  One lane owns
// 4 consecutive outputs of a row (16-byte load of g) and the 3x4 input patch they see.
__global__ void __launch_bounds__(256) upsample_bwd_w_kernel(const float* __restrict__ g,
                                                             const float* __restrict__ x,
                                                             float* __restrict__ dw,
                                                             float* __restrict__ db, int N, int C,
                                                             int H, int W) {
    __shared__ float red[4];
    const int c = blockIdx.x, S = gridDim.y;
    const int H2 = 2 * H, W2 = 2 * W;
    const bool vec = (W2 % 4 == 0);
    float acc[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) acc[j] = 0.f;
    for (int n = blockIdx.y; n < N; n += S) {
        const float* gp = g + ((size_t)n * C + c) * H2 * W2;
        const float* xp = x + ((size_t)n * C + c) * H * W;
        if (vec) {
            const int Wq = W2 / 4, items = H2 * Wq;
            for (int it = threadIdx.x; it < items; it += 256) {
                const int oh = it / Wq, ow0 = (it - oh * Wq) * 4;
                const float4 gq = *reinterpret_cast<const float4*>(gp + (size_t)oh * W2 + ow0);
                const float gv[4] = {gq.x, gq.y, gq.z, gq.w};
                acc[9] += (gv[0] + gv[1]) + (gv[2] + gv[3]);
                const int jb = ow0 / 2;     // outputs ow0..ow0+3 see input cols jb-1 .. jb+2
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int uy = oh + r - 1;
                    if (uy < 0 || uy >= H2) continue;
                    const float* row = xp + (uy >> 1) * W;
                    const float xm = jb > 0 ? row[jb - 1] : 0.f;
                    const float x0 = row[jb], x1 = row[jb + 1];
                    const float xpv = jb + 2 < W ? row[jb + 2] : 0.f;
                    // U(ow+s-1): ow0+q+s-1 >> 1 ; q = 0..3, s = 0..2
                    acc[r * 3 + 0] += gv[0] * xm + gv[1] * x0 + gv[2] * x0 + gv[3] * x1;   // s=0: cols ow-1
                    acc[r * 3 + 1] += gv[0] * x0 + gv[1] * x0 + gv[2] * x1 + gv[3] * x1;   // s=1: cols ow
                    acc[r * 3 + 2] += gv[0] * x0 + gv[1] * x1 + gv[2] * x1 + gv[3] * xpv;  // s=2: cols ow+1
                }
            }
        } else {
            for (int i = threadIdx.x; i < H2 * W2; i += 256) {
                const int oh = i / W2, ow = i - oh * W2;
                const float gv = gp[i];
                acc[9] += gv;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int uy = oh + r - 1;
                    if (uy < 0 || uy >= H2) continue;
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        const int ux = ow + s - 1;
                        if (ux >= 0 && ux < W2) acc[r * 3 + s] += gv * xp[(uy >> 1) * W + (ux >> 1)];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        const float t = block_reduce_sum_256<float>(acc[j], red);
        if (threadIdx.x == 0) {
            if (j < 9) dw[((size_t)blockIdx.y * C + c) * 9 + j] = t;
            else if (db) db[(size_t)blockIdx.y * C + c] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// global average pool of two tensors; blend out = a*xr + b*xd and its backward passes
// ------------------------------------------------------------------------------------------------
// tr (optional) = [4][C] {scale_r, shift_r, scale_d, shift_d}: the inputs are BatchNorm+ReLU outputs that were never
// written — relu(fma(x, scale, shift)) is applied on load (norm.hip: bn_apply_kernel's expression)
// U chunks of 256 * V elements per tensor are requested before any is added.  Measured in isolation (scratch/r6/gap2_time.py,
// batch 32): the large planes ran at the fabric's rate before and after (47 us for 315 MB at 64 x 120x160 = 6.6 TB/s), the short
// ones gain a microsecond (15.2 -> 14.1 us at 256 x 30x40, 14.4 -> 13.1 at 512 x 15x20).  The summation order of a lane is
// unchanged (chunks in ascending order): bit-identical results.
template <int V, bool DUAL, bool TR>
__global__ void __launch_bounds__(256) gap2_kernel(const float* __restrict__ xr,
                                                   const float* __restrict__ xd,
                                                   float* __restrict__ sr, float* __restrict__ sd,
                                                   int HW, const float* __restrict__ tr, int C) {
    constexpr int U = 4;
    __shared__ float red[4];
    const size_t base = (size_t)blockIdx.x * HW;
    const int ch = TR ? (int)(blockIdx.x % C) : 0;
    const float scr = TR ? tr[ch] : 1.f, shr = TR ? tr[C + ch] : 0.f;
    const float scd = TR ? tr[2 * C + ch] : 1.f, shd = TR ? tr[3 * C + ch] : 0.f;
    const float* pr = xr + base;
    const float* pd = DUAL ? xd + base : pr;
    float a = 0.f, b = 0.f;
    int i = threadIdx.x * V;
    for (; i + (U - 1) * 256 * V < HW; i += U * 256 * V) {        // U full chunks for this lane
        float v[U][V], w[U][V];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            vload<V>(pr + i + u * 256 * V, v[u]);
            if (DUAL) vload<V>(pd + i + u * 256 * V, w[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < V; ++j) {
                a += TR ? fmaxf(fmaf(v[u][j], scr, shr), 0.f) : v[u][j];
                if (DUAL) b += TR ? fmaxf(fmaf(w[u][j], scd, shd), 0.f) : w[u][j];
            }
    }
    for (; i < HW; i += 256 * V) {
        float v[V];
        vload<V>(pr + i, v);
#pragma unroll
        for (int j = 0; j < V; ++j) a += TR ? fmaxf(fmaf(v[j], scr, shr), 0.f) : v[j];
        if (DUAL) {
            vload<V>(pd + i, v);
#pragma unroll
            for (int j = 0; j < V; ++j) b += TR ? fmaxf(fmaf(v[j], scd, shd), 0.f) : v[j];
        }
    }
    const float ta = block_reduce_sum_256<float>(a, red);
    const float tb = block_reduce_sum_256<float>(b, red);
    if (threadIdx.x == 0) {
        sr[blockIdx.x] = ta / (float)HW;
        if (DUAL) sd[blockIdx.x] = tb / (float)HW;
    }
}

template <int V>
__global__ void __launch_bounds__(256) axpby_fwd_kernel(const float* __restrict__ xr,
                                                        const float* __restrict__ xd,
                                                        const float* __restrict__ a,
                                                        const float* __restrict__ b,
                                                        float* __restrict__ out, int HW) {
    const size_t base = (size_t)blockIdx.x * HW;
    const float ca = a[blockIdx.x], cb = b[blockIdx.x];
    const int beg = blockIdx.y * kChunk, end = min(HW, beg + kChunk);
    for (int i = beg + threadIdx.x * V; i < end; i += 256 * V) {
        float r[V], d[V];
        vload<V>(xr + base + i, r);
        vload<V>(xd + base + i, d);
#pragma unroll
        for (int j = 0; j < V; ++j) r[j] = fmaf(ca, r[j], cb * d[j]);      // explicit: axpby_pool_fwd_kernel repeats it
        vstore<V>(out + base + i, r);
    }
}

template <int V>
__global__ void __launch_bounds__(256) axpby_bwd_reduce_kernel(const float* __restrict__ g,
                                                               const float* __restrict__ xr,
                                                               const float* __restrict__ xd,
                                                               float* __restrict__ da,
                                                               float* __restrict__ db, int HW) {
    __shared__ float red[4];
    const size_t base = (size_t)blockIdx.x * HW;
    float sa = 0.f, sb = 0.f;
    for (int i = threadIdx.x * V; i < HW; i += 256 * V) {
        float gv[V], r[V], d[V];
        vload<V>(g + base + i, gv);
        vload<V>(xr + base + i, r);
        vload<V>(xd + base + i, d);
#pragma unroll
        for (int j = 0; j < V; ++j) { sa += gv[j] * r[j]; sb += gv[j] * d[j]; }
    }
    const float ta = block_reduce_sum_256<float>(sa, red);
    const float tb = block_reduce_sum_256<float>(sb, red);
    if (threadIdx.x == 0) { da[blockIdx.x] = ta; db[blockIdx.x] = tb; }
}

template <int V>
__global__ void __launch_bounds__(256) axpby_bwd_apply_kernel(
    const float* __restrict__ g, const float* __restrict__ a, const float* __restrict__ b,
    const float* __restrict__ ca, const float* __restrict__ cb, float cscale,
    float* __restrict__ dxr, float* __restrict__ dxd, int HW) {
    const size_t base = (size_t)blockIdx.x * HW;
    const float fa = a[blockIdx.x], fb = b[blockIdx.x];
    const float oa = ca ? ca[blockIdx.x] * cscale : 0.f, ob = cb ? cb[blockIdx.x] * cscale : 0.f;
    const int beg = blockIdx.y * kChunk, end = min(HW, beg + kChunk);
    for (int i = beg + threadIdx.x * V; i < end; i += 256 * V) {
        float gv[V], r[V], d[V];
        vload<V>(g + base + i, gv);
#pragma unroll
        for (int j = 0; j < V; ++j) { r[j] = fa * gv[j] + oa; d[j] = fb * gv[j] + ob; }
        vstore<V>(dxr + base + i, r);
        vstore<V>(dxd + base + i, d);
    }
}

// ------------------------------------------------------------------------------------------------
// gate-decision stream compaction: batch-row gather / merge.
//   gather: dst[i]   = src[idx[i]]                      (i < n_out)
//   merge : out[n]   = map[n] >= 0 ? sub[map[n]] : base[n]   (n < N)
// Rows are whole samples (C*H*W floats), so every access is a long contiguous run.
// ------------------------------------------------------------------------------------------------
template <int V>
__global__ void __launch_bounds__(256) batch_gather_kernel(const float* __restrict__ src,
                                                           const int* __restrict__ idx,
                                                           float* __restrict__ dst, size_t row) {
    const size_t s = (size_t)(idx ? idx[blockIdx.y] : (int)blockIdx.y) * row, d = (size_t)blockIdx.y * row;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * V; i < row; i += (size_t)gridDim.x * 256 * V) {
        float v[V];
        vload<V>(src + s + i, v);
        vstore<V>(dst + d + i, v);
    }
}

template <int V>
__global__ void __launch_bounds__(256) batch_merge_kernel(const float* __restrict__ base,
                                                          const float* __restrict__ sub,
                                                          const int* __restrict__ map,
                                                          float* __restrict__ out, size_t row) {
    const int m = map[blockIdx.y];
    const float* src = m >= 0 ? sub + (size_t)m * row : base + (size_t)blockIdx.y * row;
    const size_t d = (size_t)blockIdx.y * row;
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * V; i < row; i += (size_t)gridDim.x * 256 * V) {
        float v[V];
        vload<V>(src + i, v);
        vstore<V>(out + d + i, v);
    }
}

// out = src[0] + src[1] (+ src[2] (+ src[3])), summed left to right: the gradient of a tensor that fans out to
// several consumers (one pass over n inputs instead of autograd's n-1 pairwise accumulation passes).
struct AddSrcs { const float* p[4]; };
template <int V>
__global__ void __launch_bounds__(256) add_n_kernel(AddSrcs S, int n, float* __restrict__ out, size_t numel) {
    for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * V; i < numel; i += (size_t)gridDim.x * 256 * V) {
        float acc[V], v[V];
        vload<V>(S.p[0] + i, acc);
        vload<V>(S.p[1] + i, v);
#pragma unroll
        for (int j = 0; j < V; ++j) acc[j] += v[j];
        if (n > 2) {
            vload<V>(S.p[2] + i, v);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += v[j];
        }
        if (n > 3) {
            vload<V>(S.p[3] + i, v);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += v[j];
        }
        vstore<V>(out + i, acc);
    }
}

}  // namespace dynmm

using namespace dynmm;

#define ST ((hipStream_t)stream)

extern "C" int dynmm_add_n(const float* const* srcs, int n, float* out, size_t numel, void* stream) {
    (void)hipGetLastError();
    if (!srcs || !out || n < 2 || n > 4 || numel == 0) return DYNMM_EINVAL;
    AddSrcs S{};
    bool v4 = (numel % 4 == 0) && aligned16(out);
    for (int i = 0; i < n; ++i) {
        if (!srcs[i]) return DYNMM_EINVAL;
        S.p[i] = srcs[i];
        v4 = v4 && aligned16(srcs[i]);
    }
    size_t blocks = (numel / (v4 ? 4 : 1) + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    if (v4)
        hipLaunchKernelGGL(add_n_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, ST, S, n, out, numel);
    else
        hipLaunchKernelGGL(add_n_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, ST, S, n, out, numel);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_batch_gather(const float* src, const int* idx, float* dst, int n_out, size_t row,
                                  void* stream) {
    (void)hipGetLastError();
    if (!src || !dst || n_out <= 0 || row == 0) return DYNMM_EINVAL;
    unsigned bx = (unsigned)((row / 4 + 255) / 256);
    if (bx > 64) bx = 64;
    if (bx < 1) bx = 1;
    if (row % 4 == 0 && aligned16(src) && aligned16(dst))
        hipLaunchKernelGGL(batch_gather_kernel<4>, dim3(bx, n_out), dim3(256), 0, ST, src, idx, dst, row);
    else
        hipLaunchKernelGGL(batch_gather_kernel<1>, dim3(bx, n_out), dim3(256), 0, ST, src, idx, dst, row);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_batch_merge(const float* base, const float* sub, const int* map, float* out, int N,
                                 size_t row, void* stream) {
    (void)hipGetLastError();
    if (!base || !sub || !map || !out || N <= 0 || row == 0) return DYNMM_EINVAL;
    unsigned bx = (unsigned)((row / 4 + 255) / 256);
    if (bx > 64) bx = 64;
    if (bx < 1) bx = 1;
    if (row % 4 == 0 && aligned16(base) && aligned16(sub) && aligned16(out))
        hipLaunchKernelGGL(batch_merge_kernel<4>, dim3(bx, N), dim3(256), 0, ST, base, sub, map, out, row);
    else
        hipLaunchKernelGGL(batch_merge_kernel<1>, dim3(bx, N), dim3(256), 0, ST, base, sub, map, out, row);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_maxpool3x3s2_fwd(const float* x, float* y, signed char* idx, int N, int C, int H,
                                      int W, int Ho, int Wo, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0) return DYNMM_EINVAL;
    if (Ho != (H + 2 - 3) / 2 + 1 || Wo != (W + 2 - 3) / 2 + 1) return DYNMM_EINVAL;
    dim3 grid(N * C, plane_chunks(Ho * Wo, kChunk));
    const bool wide = H % 2 == 0 && W % 8 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15u) == 0 &&
                      (reinterpret_cast<uintptr_t>(idx) & 3u) == 0;
    if (wide)
        hipLaunchKernelGGL(maxpool_fwd4_kernel, grid, dim3(256), 0, ST, x, y, idx, H, W, Ho, Wo);
    else
        hipLaunchKernelGGL(maxpool_fwd_kernel, grid, dim3(256), 0, ST, x, y, idx, H, W, Ho, Wo);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_maxpool3x3s2_bwd(const float* g, const signed char* idx, float* dx, int N, int C,
                                      int H, int W, int Ho, int Wo, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!g || !idx || !dx || N <= 0 || C <= 0 || H <= 0 || W <= 0) return DYNMM_EINVAL;
    dim3 grid(N * C, plane_chunks(H * W, kChunk));
    if (H % 2 == 0 && W % 8 == 0 && ((reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(g)) & 15u) == 0 &&
        (reinterpret_cast<uintptr_t>(idx) & 3u) == 0) {
        // one thread per 2 x 8 input pixels = per 4 pooled pixels: the forward's chunking
        hipLaunchKernelGGL(maxpool_bwd8_kernel, dim3(N * C, plane_chunks(Ho * Wo, kChunk)), dim3(256), 0, ST, g, idx, dx,
                           H, W, Ho, Wo);
    } else if (W % 4 == 0 && (reinterpret_cast<uintptr_t>(dx) & 15u) == 0)
        hipLaunchKernelGGL(maxpool_bwd4_kernel, grid, dim3(256), 0, ST, g, idx, dx, H, W, Ho, Wo);
    else
        hipLaunchKernelGGL(maxpool_bwd_kernel, grid, dim3(256), 0, ST, g, idx, dx, H, W, Ho, Wo);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_adaptive_avgpool_fwd(const float* x, float* y, int NC, int H, int W, int OH,
                                          int OW, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!x || !y || NC <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return DYNMM_EINVAL;
    hipLaunchKernelGGL(adaptive_avgpool_fwd_kernel, dim3(ceil_div(NC * OH * OW, 256)), dim3(256), 0,
                       ST, x, y, NC, H, W, OH, OW);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_adaptive_avgpool_bwd(const float* g, float* dx, int NC, int H, int W, int OH,
                                          int OW, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!g || !dx || NC <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0) return DYNMM_EINVAL;
    if (OH > kApMaxBins || OW > kApMaxBins) return DYNMM_EUNSUPPORTED;
    hipLaunchKernelGGL(adaptive_avgpool_bwd_kernel, dim3(ceil_div(NC * H * W, 256)), dim3(256), 0,
                       ST, g, dx, NC, H, W, OH, OW);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_nearest_into_fwd(const float* y, float* out, int N, int C, int h, int w,
                                      int Ctot, int c_off, int H, int W, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!y || !out || N <= 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return DYNMM_EINVAL;
    if (c_off < 0 || c_off + C > Ctot) return DYNMM_EINVAL;
    hipLaunchKernelGGL(nearest_into_fwd_kernel, dim3(ceil_div(N * C * H * W, 256)), dim3(256), 0, ST,
                       y, out, N, C, h, w, Ctot, c_off, H, W);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_nearest_into_bwd(const float* g_out, float* dy, int N, int C, int h, int w,
                                      int Ctot, int c_off, int H, int W, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!g_out || !dy || N <= 0 || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return DYNMM_EINVAL;
    if (c_off < 0 || c_off + C > Ctot) return DYNMM_EINVAL;
    // lanes per output element: a power of two near a quarter of the rectangle a source pixel collects from
    const int area = ceil_div(H, h) * ceil_div(W, w);
    const int nout = N * C * h * w;
#define DYNMM_NIB(L) hipLaunchKernelGGL(nearest_into_bwd_kernel<L>, dim3(ceil_div_sz((size_t)nout * L, 256)), dim3(256), 0, ST, \
                                        g_out, dy, N, C, h, w, Ctot, c_off, H, W)
    if ((h == H && w == W) || area < 8) DYNMM_NIB(1);
    else if (area < 32) DYNMM_NIB(4);
    else if (area < 128) DYNMM_NIB(16);
    else DYNMM_NIB(64);
#undef DYNMM_NIB
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_upsample2x_dw3x3_fwd(const float* x, const float* w, const float* b,
                                          const float* skip, float* y, int N, int C, int H, int W,
                                          void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!x || !w || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0) return DYNMM_EINVAL;
    if (!aligned16(y) || (skip && !aligned16(skip))) return DYNMM_EUNSUPPORTED;
    dim3 grid(N * C, plane_chunks(H * ((W + 1) / 2), kChunk / 8));
    hipLaunchKernelGGL(upsample_fwd_kernel, grid, dim3(256), 0, ST, x, w, b, skip, y, C, H, W);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

static int upsample_w_splits(int N, int C) {
    int S = 2048 / C;
    if (S < 1) S = 1;
    if (S > N) S = N;
    return S;
}

extern "C" size_t dynmm_upsample2x_dw3x3_bwd_workspace_bytes(int N, int C) {
    if (N <= 0 || C <= 0) return 0;
    const int S = upsample_w_splits(N, C);
    return S > 1 ? sizeof(float) * (size_t)S * C * 10 : 0;
}

extern "C" int dynmm_upsample2x_dw3x3_bwd(const float* g, const float* x, const float* w, float* dx,
                                          float* dw, float* db, float* workspace, int N, int C, int H, int W,
                                          void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!g || !w || N <= 0 || C <= 0 || H <= 0 || W <= 0) return DYNMM_EINVAL;
    if (dx) {
        dim3 grid(N * C, plane_chunks(H * W, kChunk));
        const int pairs = (W % 2 == 0) && aligned16(g) && ((reinterpret_cast<uintptr_t>(dx) & 7u) == 0);
        hipLaunchKernelGGL(upsample_bwd_dx_kernel, grid, dim3(256), 0, ST, g, w, dx, C, H, W, pairs);
        DYNMM_LAUNCH_CHECK();
    }
    if (dw) {
        if (!x) return DYNMM_EINVAL;
        const int S = upsample_w_splits(N, C);
        if (S == 1) {
            hipLaunchKernelGGL(upsample_bwd_w_kernel, dim3(C, 1), dim3(256), 0, ST, g, x, dw, db, N, C, H, W);
            DYNMM_LAUNCH_CHECK();
        } else {
            if (!workspace) return DYNMM_EWORKSPACE;
            float* pw = workspace, *pb = workspace + (size_t)S * C * 9;
            hipLaunchKernelGGL(upsample_bwd_w_kernel, dim3(C, S), dim3(256), 0, ST, g, x, pw, db ? pb : nullptr, N, C, H, W);
            DYNMM_LAUNCH_CHECK();
            launch_reduce_slabs(pw, dw, C * 9, S, ST, db ? pb : nullptr, db, db ? C : 0);
            DYNMM_LAUNCH_CHECK();
        }
    }
    return DYNMM_OK;
}

static int gap2_launch(const float* xr, const float* xd, float* sr, float* sd, int NC, int HW, const float* tr, int C,
                       void* stream);

extern "C" int dynmm_gap2_fwd(const float* xr, const float* xd, float* sr, float* sd, int NC, int HW,
                              void* stream) {
    return gap2_launch(xr, xd, sr, sd, NC, HW, nullptr, 1, stream);
}

extern "C" int dynmm_gap2_bnrelu_fwd(const float* xr, const float* xd, const float* bn_tr, int C, float* sr, float* sd,
                                     int NC, int HW, void* stream) {
    if (!bn_tr || C <= 0 || NC % C != 0) return DYNMM_EINVAL;
    return gap2_launch(xr, xd, sr, sd, NC, HW, bn_tr, C, stream);
}

static int gap2_launch(const float* xr, const float* xd, float* sr, float* sd, int NC, int HW, const float* tr, int C,
                       void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!xr || !sr || NC <= 0 || HW <= 0) return DYNMM_EINVAL;
    if (xd && !sd) return DYNMM_EINVAL;
#define DYNMM_GAP2(V, DUAL, TR) hipLaunchKernelGGL((gap2_kernel<V, DUAL, TR>), dim3(NC), dim3(256), 0, ST, xr, xd, sr, sd, HW, tr, C)
#define DYNMM_GAP2_V(V)                                    \
    do {                                                   \
        if (xd && tr) DYNMM_GAP2(V, true, true);           \
        else if (xd) DYNMM_GAP2(V, true, false);           \
        else if (tr) DYNMM_GAP2(V, false, true);           \
        else DYNMM_GAP2(V, false, false);                  \
    } while (0)
    if (can_vec4(HW, {xr, xd}))
        DYNMM_GAP2_V(4);
    else
        DYNMM_GAP2_V(1);
#undef DYNMM_GAP2_V
#undef DYNMM_GAP2
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_axpby_fwd(const float* xr, const float* xd, const float* a, const float* b,
                               float* out, int NC, int HW, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!xr || !xd || !a || !b || !out || NC <= 0 || HW <= 0) return DYNMM_EINVAL;
    dim3 grid(NC, plane_chunks(HW, kChunk));
    if (can_vec4(HW, {xr, xd, out}))
        hipLaunchKernelGGL(axpby_fwd_kernel<4>, grid, dim3(256), 0, ST, xr, xd, a, b, out, HW);
    else
        hipLaunchKernelGGL(axpby_fwd_kernel<1>, grid, dim3(256), 0, ST, xr, xd, a, b, out, HW);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_axpby_bwd_reduce(const float* g, const float* xr, const float* xd, float* da,
                                      float* db, int NC, int HW, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!g || !xr || !xd || !da || !db || NC <= 0 || HW <= 0) return DYNMM_EINVAL;
    if (can_vec4(HW, {g, xr, xd}))
        hipLaunchKernelGGL(axpby_bwd_reduce_kernel<4>, dim3(NC), dim3(256), 0, ST, g, xr, xd, da, db, HW);
    else
        hipLaunchKernelGGL(axpby_bwd_reduce_kernel<1>, dim3(NC), dim3(256), 0, ST, g, xr, xd, da, db, HW);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_axpby_bwd_apply(const float* g, const float* a, const float* b, const float* ca,
                                     const float* cb, float cscale, float* dxr, float* dxd, int NC,
                                     int HW, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!g || !a || !b || !dxr || !dxd || NC <= 0 || HW <= 0) return DYNMM_EINVAL;
    dim3 grid(NC, plane_chunks(HW, kChunk));
    if (can_vec4(HW, {g, dxr, dxd}))
        hipLaunchKernelGGL(axpby_bwd_apply_kernel<4>, grid, dim3(256), 0, ST, g, a, b, ca, cb, cscale, dxr, dxd, HW);
    else
        hipLaunchKernelGGL(axpby_bwd_apply_kernel<1>, grid, dim3(256), 0, ST, g, a, b, ca, cb, cscale, dxr, dxd, HW);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

namespace dynmm {

// ------------------------------------------------------------------------------------------------
// BatchNorm backward of a stem whose output gradient is never written either (ops._StemBNFusePool):
//   gy = coef[plane] * pool_bwd(go, io) + off[plane]*cscale (+ pool_bwd(gd, id))      (axpby_pool_bwd_apply's rows)
//   g_eff = gy * [fma(x, sc, sh) > 0]                                                  (the stem's ReLU, mask from x)
//   reduce: sums[c] += sum g_eff, sums[C+c] += sum g_eff * xhat        apply: dx = gamma*invstd*(g_eff - m1 - xhat*m2)
// i.e. bn_bwd_reduce / bn_bwd_apply (norm.hip) with their gradient operand derived block by block from the pooled
// gradients: 3 passes over a 629 MB tensor per stem instead of 6 (+1 shared).
// ------------------------------------------------------------------------------------------------
struct StemBnArgs {
    const float* go; const signed char* io;        // pooled gradient / codes of the fused map
    const float* gd; const signed char* id;        // ... of the pooled depth map (NULL for the RGB stem)
    const float* coef; const float* off; float cscale;
    const float* x; const float* mean; const float* invstd; const float* gamma; const float* beta;
    int N, C, H, W, Ho, Wo;
};

__device__ __forceinline__ void stem_gy_block(const StemBnArgs& a, size_t plane, int ar, int t, float fc, float fo,
                                              float (&top)[8], float (&bot)[8]) {
    const size_t po = plane * (size_t)a.Ho * a.Wo;
    pool_bwd_block(a.go + po, a.io + po, ar, t, a.Ho, a.Wo, top, bot);
#pragma unroll
    for (int k = 0; k < 8; ++k) { top[k] = fmaf(fc, top[k], fo); bot[k] = fmaf(fc, bot[k], fo); }
    if (a.gd) {
        float dt[8], db[8];
        pool_bwd_block(a.gd + po, a.id + po, ar, t, a.Ho, a.Wo, dt, db);
#pragma unroll
        for (int k = 0; k < 8; ++k) { top[k] += dt[k]; bot[k] += db[k]; }
    }
}

__global__ void __launch_bounds__(256) stem_bn_bwd_reduce_kernel(const StemBnArgs a, double* __restrict__ sums) {
    __shared__ float red[4];
    const int c = blockIdx.x, S = gridDim.y;
    const float mu = a.mean[c], is = a.invstd[c];
    const float sc = a.gamma[c] * is;
    const float sh = fmaf(-mu, sc, a.beta[c]);
    const int Wq = a.Wo / 4, nq = a.Ho * Wq;
    float s1 = 0.f, s2 = 0.f;
    for (int n = blockIdx.y; n < a.N; n += S) {
        const size_t plane = (size_t)n * a.C + c;
        const float fc = a.coef[plane], fo = a.off ? a.off[plane] * a.cscale : 0.f;
        const float* xp = a.x + plane * (size_t)a.H * a.W;
        float a1 = 0.f, a2 = 0.f;
        for (int q = threadIdx.x; q < nq; q += 256) {
            const int ar = q / Wq, t = q - ar * Wq;
            float top[8], bot[8];
            stem_gy_block(a, plane, ar, t, fc, fo, top, bot);
            const size_t o = (size_t)(2 * ar) * a.W + 8 * t;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 x0 = *reinterpret_cast<const float4*>(xp + o + 4 * h);
                const float4 x1 = *reinterpret_cast<const float4*>(xp + o + a.W + 4 * h);
                const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float gy = k < 4 ? top[4 * h + k] : bot[4 * h + k - 4];
                    const float ge = fmaf(xv[k], sc, sh) > 0.f ? gy : 0.f;
                    a1 += ge;
                    a2 += ge * (xv[k] - mu) * is;
                }
            }
        }
        s1 += a1; s2 += a2;
    }
    const float t1 = block_reduce_sum_256<float>(s1, red);
    const float t2 = block_reduce_sum_256<float>(s2, red);
    if (threadIdx.x == 0) {
        atomicAdd(&sums[c], (double)t1);
        atomicAdd(&sums[a.C + c], (double)t2);
    }
}

__global__ void __launch_bounds__(256) stem_bn_bwd_apply_kernel(const StemBnArgs a, const double* __restrict__ sums,
                                                                float* __restrict__ dx, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta) {
    const size_t plane = blockIdx.x;
    const int c = (int)(plane % a.C);
    const float mu = a.mean[c], is = a.invstd[c];
    const float sc = a.gamma[c] * is;
    const float sh = fmaf(-mu, sc, a.beta[c]);
    const float sg = (float)sums[c], sgx = (float)sums[a.C + c];
    if (plane < (size_t)a.C && blockIdx.y == 0 && threadIdx.x == 0) {
        if (dgamma) dgamma[c] = sgx;
        if (dbeta) dbeta[c] = sg;
    }
    const float invM = 1.f / ((float)a.N * (float)(a.H * a.W));
    const float k0 = sc, m1 = sg * invM, m2 = sgx * invM;
    const float fc = a.coef[plane], fo = a.off ? a.off[plane] * a.cscale : 0.f;
    const float* xp = a.x + plane * (size_t)a.H * a.W;
    float* dp = dx + plane * (size_t)a.H * a.W;
    const int Wq = a.Wo / 4, nq = a.Ho * Wq;
    const int beg = blockIdx.y * (kChunk / 4), end = min(nq, beg + kChunk / 4);
    for (int q = beg + threadIdx.x; q < end; q += 256) {
        const int ar = q / Wq, t = q - ar * Wq;
        float top[8], bot[8];
        stem_gy_block(a, plane, ar, t, fc, fo, top, bot);
        const size_t o = (size_t)(2 * ar) * a.W + 8 * t;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 x0 = *reinterpret_cast<const float4*>(xp + o + 4 * h);
            const float4 x1 = *reinterpret_cast<const float4*>(xp + o + a.W + 4 * h);
            const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            float ov[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float gy = k < 4 ? top[4 * h + k] : bot[4 * h + k - 4];
                const float ge = fmaf(xv[k], sc, sh) > 0.f ? gy : 0.f;
                ov[k] = k0 * (ge - m1 - (xv[k] - mu) * is * m2);
            }
            *reinterpret_cast<float4*>(dp + o + 4 * h) = make_float4(ov[0], ov[1], ov[2], ov[3]);
            *reinterpret_cast<float4*>(dp + o + a.W + 4 * h) = make_float4(ov[4], ov[5], ov[6], ov[7]);
        }
    }
}

}  // namespace dynmm

static bool pool_fusable(int H, int W, int Ho, int Wo, std::initializer_list<const void*> p16,
                         std::initializer_list<const void*> p4) {
    if (H % 2 != 0 || W % 8 != 0 || Ho != H / 2 || Wo != W / 2) return false;
    for (const void* p : p16) if (reinterpret_cast<uintptr_t>(p) & 15u) return false;
    for (const void* p : p4) if (reinterpret_cast<uintptr_t>(p) & 3u) return false;
    return true;
}

extern "C" int dynmm_axpby_pool_supported(int H, int W) { return (H > 0 && W > 0 && H % 2 == 0 && W % 8 == 0) ? 1 : 0; }

extern "C" int dynmm_axpby_pool_fwd(const float* xr, const float* xd, const float* a, const float* b, float* y_out,
                                    signed char* idx_out, float* y_depth, signed char* idx_depth, const float* bn_tr,
                                    int C, int NC, int H, int W, void* stream) {
    (void)hipGetLastError();
    if (!xr || !xd || !a || !b || !y_out || !idx_out || !y_depth || !idx_depth || NC <= 0) return DYNMM_EINVAL;
    if (bn_tr && (C <= 0 || NC % C != 0)) return DYNMM_EINVAL;
    const int Ho = H / 2, Wo = W / 2;
    if (!pool_fusable(H, W, Ho, Wo, {xr, xd, y_out, y_depth}, {idx_out, idx_depth})) return DYNMM_EUNSUPPORTED;
    hipLaunchKernelGGL(axpby_pool_fwd_kernel, dim3(NC, plane_chunks(Ho * Wo, kChunk)), dim3(256), 0, ST, xr, xd, a, b,
                       y_out, idx_out, y_depth, idx_depth, H, W, Ho, Wo, bn_tr, C);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_axpby_pool_bwd_reduce(const float* g_out, const signed char* idx_out, const float* xr,
                                           const float* xd, float* da, float* db, const float* bn_tr, int C, int NC,
                                           int H, int W, void* stream) {
    (void)hipGetLastError();
    if (!g_out || !idx_out || !xr || !xd || !da || !db || NC <= 0) return DYNMM_EINVAL;
    if (bn_tr && (C <= 0 || NC % C != 0)) return DYNMM_EINVAL;
    const int Ho = H / 2, Wo = W / 2;
    if (!pool_fusable(H, W, Ho, Wo, {g_out, xr, xd}, {idx_out})) return DYNMM_EUNSUPPORTED;
    hipLaunchKernelGGL(axpby_pool_bwd_reduce_kernel, dim3(NC), dim3(256), 0, ST, g_out, idx_out, xr, xd, da, db, H, W,
                       Ho, Wo, bn_tr, C);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_axpby_pool_bwd_apply(const float* g_out, const signed char* idx_out, const float* g_depth,
                                          const signed char* idx_depth, const float* a, const float* b, const float* ca,
                                          const float* cb, float cscale, float* dxr, float* dxd, int NC, int H, int W,
                                          void* stream) {
    (void)hipGetLastError();
    if (!g_out || !idx_out || !g_depth || !idx_depth || !a || !b || !dxr || !dxd || NC <= 0) return DYNMM_EINVAL;
    const int Ho = H / 2, Wo = W / 2;
    if (!pool_fusable(H, W, Ho, Wo, {g_out, g_depth, dxr, dxd}, {idx_out, idx_depth})) return DYNMM_EUNSUPPORTED;
    hipLaunchKernelGGL(axpby_pool_bwd_apply_kernel, dim3(NC, plane_chunks(Ho * Wo, kChunk)), dim3(256), 0, ST, g_out,
                       idx_out, g_depth, idx_depth, a, b, ca, cb, cscale, dxr, dxd, H, W, Ho, Wo);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

static int stem_bn_args(StemBnArgs& a, const float* g_out, const signed char* idx_out, const float* g_depth,
                        const signed char* idx_depth, const float* coef, const float* off, float cscale, const float* x,
                        const float* mean, const float* invstd, const float* gamma, const float* beta, int N, int C, int H,
                        int W) {
    if (!g_out || !idx_out || !coef || !x || !mean || !invstd || !gamma || !beta || N <= 0 || C <= 0) return DYNMM_EINVAL;
    if ((g_depth != nullptr) != (idx_depth != nullptr)) return DYNMM_EINVAL;
    const int Ho = H / 2, Wo = W / 2;
    if (!pool_fusable(H, W, Ho, Wo, {g_out, g_depth, x}, {idx_out, idx_depth})) return DYNMM_EUNSUPPORTED;
    a = StemBnArgs{g_out, idx_out, g_depth, idx_depth, coef, off, cscale, x, mean, invstd, gamma, beta, N, C, H, W, Ho, Wo};
    return DYNMM_OK;
}

extern "C" int dynmm_stem_bn_bwd_reduce(const float* g_out, const signed char* idx_out, const float* g_depth,
                                        const signed char* idx_depth, const float* coef, const float* off, float cscale,
                                        const float* x, const float* mean, const float* invstd, const float* gamma,
                                        const float* beta, double* sums, int N, int C, int H, int W, int sums_are_zero,
                                        void* stream) {
    (void)hipGetLastError();
    StemBnArgs a;
    const int rc = stem_bn_args(a, g_out, idx_out, g_depth, idx_depth, coef, off, cscale, x, mean, invstd, gamma, beta, N, C,
                                H, W);
    if (rc != DYNMM_OK || !sums) return rc != DYNMM_OK ? rc : DYNMM_EINVAL;
    if (!sums_are_zero) DYNMM_HIP_TRY(hipMemsetAsync(sums, 0, sizeof(double) * 2 * C, ST));
    int S = 2048 / C;
    S = S < 1 ? 1 : (S > N ? N : S);
    hipLaunchKernelGGL(stem_bn_bwd_reduce_kernel, dim3(C, S), dim3(256), 0, ST, a, sums);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_stem_bn_bwd_apply(const float* g_out, const signed char* idx_out, const float* g_depth,
                                       const signed char* idx_depth, const float* coef, const float* off, float cscale,
                                       const float* x, const float* mean, const float* invstd, const float* gamma,
                                       const float* beta, const double* sums, float* dx, float* dgamma, float* dbeta,
                                       int N, int C, int H, int W, void* stream) {
    (void)hipGetLastError();
    StemBnArgs a;
    const int rc = stem_bn_args(a, g_out, idx_out, g_depth, idx_depth, coef, off, cscale, x, mean, invstd, gamma, beta, N, C,
                                H, W);
    if (rc != DYNMM_OK || !sums || !dx) return rc != DYNMM_OK ? rc : DYNMM_EINVAL;
    if (reinterpret_cast<uintptr_t>(dx) & 15u) return DYNMM_EUNSUPPORTED;
    hipLaunchKernelGGL(stem_bn_bwd_apply_kernel, dim3(N * C, plane_chunks(a.Ho * a.Wo, kChunk)), dim3(256), 0, ST, a, sums,
                       dx, dgamma, dbeta);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}
