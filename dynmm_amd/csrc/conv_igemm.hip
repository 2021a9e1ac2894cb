// Implicit-GEMM convolution for gfx950 on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// fp32 MFMA (exact f32, 157 TFLOP/s peak) is used because the parity bar is 1e-3 relative on the
// logits and bf16/fp16 inputs miss it by 4-16x.
//
// GEMM view (all tensors NCHW, fp32):
//     D[co][pix] = sum_k  Wp[k][co] * Xcol[k][pix],   k = (tap, ci), pix = (n, oh, ow)
// The MFMA "A" operand is the weight tile (i = co) and the "B" operand is the gathered input
// tile (j = pixel), so in the accumulator lane&31 indexes consecutive PIXELS: every global store
// (and residual / mask load) of the epilogue is a 128-byte contiguous run per half-wave in NCHW,
// and the input gather is contiguous along W for stride-1 convs.
//
// One kernel template serves the forward conv and the data gradient (DGRAD = transposed
// addressing on pre-transposed weights); a second one computes the weight gradient with the
// pixel axis as the reduction dimension, split over workgroups into slabs that are reduced
// deterministically (no float atomics).
#include <stdlib.h>

#include "common.h"
#include "conv_igemm.h"
#include "conv_small.h"
#include "vec.h"

namespace dynmm {


// uniform (SGPR) base + 32-bit per-lane byte offset: lets the compiler pick the
// `global_load_dword v, v_off, s[base:base+1]` form — one VALU-free address per load instead of a
// 64-bit multiply-add chain.  All tensors on this path are < 4 GiB (checked in geom_ok).
__device__ __forceinline__ float ldg_f32(const float* sbase, unsigned voff_bytes) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(sbase) + voff_bytes);
}
__device__ __forceinline__ float4 ldg_f32x4(const float* sbase, unsigned voff_bytes) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sbase) + voff_bytes);
}


// ------------------------------------------------------------------------------------------------
// forward / dgrad
// ------------------------------------------------------------------------------------------------
// MODE 0: fast path (reduction channels % 16 == 0, one filter tap per K-step); MODE 1: the same with the channel
// count rounded up to 16 by zero weight rows (8 <= C, C % 16 != 0: dgrad of the 40-class convs, the 8-channel gate
// conv) — kept out of MODE 0 because even two extra VALU per staged element cost the hot kernels ~10 %;
// MODE 2: generic element-wise loader (stems C = 1 / 3).
template <int TCO, int TPIX, int WCO, int WPIX, bool DGRAD, int MODE>
__global__ void __launch_bounds__(256, (TCO * TPIX > 8192 ? 2 : (TPIX > 128 ? 4 : 6))) conv_igemm_kernel(const IgemmArgs a) {
    constexpr bool GENERIC = MODE == 2;
    constexpr bool padded = MODE == 1;
    constexpr int BK = 16;
    constexpr int MCO = WCO / 32, MPIX = WPIX / 32;
    constexpr int WAVES_PIX = TPIX / WPIX;
    static_assert((TCO / WCO) * WAVES_PIX == 4, "4 waves per workgroup");
    // weight tile: float4 along co.  AQ float4 per k-row.
    constexpr int AQ = TCO / 4, A_ROWSTEP = 256 / AQ;
    constexpr int A_PER = (BK * AQ + 255) / 256;
    constexpr int B_PER = BK * TPIX / 256, B_ROWSTEP = 256 / TPIX;

    __shared__ __attribute__((aligned(16))) float As[2][BK][TCO];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][TPIX];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wave_co = wave / WAVES_PIX, wave_pix = wave % WAVES_PIX;
    DYNMM_TRACE_MARK(0);
#ifdef DYNMM_TRACE
    if (g_trace && threadIdx.x == 0) g_trace[(size_t)blockIdx.x * 6 + 4] = clock64();   // shader clock at entry
#endif

    // XCD-aware tile order: consecutive logical ids (same XCD) walk the co-tiles of one pixel tile
    // first, so the gathered input tile is re-used out of that XCD's L2.
    const int nblk = a.n_co_tiles * a.n_pix_tiles;
    const int lin = xcd_remap(blockIdx.x, nblk);
    const int co0 = (lin % a.n_co_tiles) * TCO;
    const int pix0 = (lin / a.n_co_tiles) * TPIX;

    const int HW = a.H * a.W;
    const int HoWo = a.Ho * a.Wo;

    // ---- per-thread loader coordinates (fixed for the whole K loop) ----
    const int a_q = t % AQ, a_row0 = t / AQ;
    const bool a_active = a_row0 < BK;                 // TCO=32: only half the threads carry weights
    int q_glob = co0 / 4 + a_q;
    if (q_glob > a.CoP / 4 - 1) q_glob = a.CoP / 4 - 1;  // rows past Co: finite junk, discarded below
    const unsigned a_voff = (unsigned)((a_active ? a_row0 : 0) * a.CoP + q_glob * 4) * 4u;

    // GEMM pixel index m -> (n, oh, ow) of the output tensor.  Strided dgrad (a.subpix): within an image the
    // SH*SW parity classes (oh % SH, ow % SW) are enumerated one after the other, because a filter tap only
    // reaches the output pixels of ONE class — a tile inside a class can skip the other taps outright
    // instead of multiplying zeros (sub-pixel decomposition of the transposed convolution).
    auto pix_decode = [&](int m, int& n, int& oh, int& ow, int& cls) {
        n = m / HoWo;
        const int q = m - n * HoWo;
        if (DGRAD && a.subpix) {
            const int Wc = a.Wo / a.SW;
            const int csz = HoWo / (a.SH * a.SW);
            cls = q / csz;
            const int q2 = q - cls * csz;
            const int ohc = q2 / Wc;
            const int ph = cls / a.SW;
            oh = ohc * a.SH + ph;
            ow = (q2 - ohc * Wc) * a.SW + (cls - ph * a.SW);
        } else {
            cls = 0;
            oh = q / a.Wo;
            ow = q - oh * a.Wo;
        }
    };
    const int b_col = t % TPIX, b_row0 = t / TPIX;
    const int m_b = pix0 + b_col;
    const bool m_ok = m_b < a.M;
    int n_b = 0, oh_b = 0, ow_b = 0, cls_b = 0;
    if (m_ok) pix_decode(m_b, n_b, oh_b, ow_b, cls_b);
    (void)cls_b;
    const int c2 = a.Ci - a.c_in_split;

    auto in_coord = [&](int r, int s, int& ih, int& iw) -> bool {
        if (!DGRAD) {
            ih = oh_b * a.SH - a.PH + r;
            iw = ow_b * a.SW - a.PW + s;
            return m_ok && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
        } else {
            const int th = oh_b + a.PH - r, tw = ow_b + a.PW - s;
            if (!m_ok || th < 0 || tw < 0) return false;
            ih = th / a.SH;
            iw = tw / a.SW;
            return (ih * a.SH == th) && (iw * a.SW == tw) && ih < a.H && iw < a.W;
        }
    };

    float4 ra[A_PER];
    float rb[B_PER];
    f32x16 acc[MCO][MPIX];
#pragma unroll
    for (int mi = 0; mi < MCO; ++mi)
#pragma unroll
        for (int ni = 0; ni < MPIX; ++ni)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[mi][ni][j] = 0.f;

    const int khalf = lane >> 5, l31 = lane & 31;

    // Fragments for the whole K-step are fetched from LDS first (BK/2 * (MCO+MPIX) VGPRs), then the
    // MFMAs issue back to back: one LDS round trip per K-step per wave instead of one per k-pair
    // (+2 % measured).  Tried and rejected on MI355X (all cost occupancy or code quality, -5..-40 %):
    // BK=32, 2-deep register prefetch, a persistent multi-tile loop per workgroup.
    auto mfma_tile = [&](int buf) {
        float af[BK / 2][MCO], bf[BK / 2][MPIX];
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
#pragma unroll
            for (int mi = 0; mi < MCO; ++mi)
                af[kk][mi] = As[buf][2 * kk + khalf][wave_co * WCO + mi * 32 + l31];
#pragma unroll
            for (int ni = 0; ni < MPIX; ++ni)
                bf[kk][ni] = Bs[buf][2 * kk + khalf][wave_pix * WPIX + ni * 32 + l31];
        }
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk)
#pragma unroll
            for (int mi = 0; mi < MCO; ++mi)
#pragma unroll
                for (int ni = 0; ni < MPIX; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][mi], bf[kk][ni], acc[mi][ni], 0, 0, 0);
    };

    if constexpr (!GENERIC) {
        // ---------------- fast path: Ci % 16 == 0, one filter tap per K-step ----------------
        const int cpt = a.CiR / BK;              // MODE 1: the last K-step of a tap is partly zero rows
        // taps that can reach this tile: all of them, unless the tile lies inside one parity class
        int live_ph = -1, live_pw = -1;
        if (DGRAD && a.subpix) {
            int n0, oh0, ow0, c0, n1, oh1, ow1, c1;
            pix_decode(pix0, n0, oh0, ow0, c0);
            pix_decode(min(pix0 + TPIX, a.M) - 1, n1, oh1, ow1, c1);
            if (n0 == n1 && c0 == c1) { live_ph = c0 / a.SW; live_pw = c0 - live_ph * a.SW; }
        }
        auto tap_live = [&](int r, int s_) -> bool {
            if (!(DGRAD && a.subpix) || live_ph < 0) return true;
            return ((live_ph + a.PH - r) % a.SH == 0) && ((live_pw + a.PW - s_) % a.SW == 0);
        };
        int n_live = 0;
        for (int r = 0; r < a.KH; ++r)
            for (int s_ = 0; s_ < a.KW; ++s_) n_live += tap_live(r, s_) ? 1 : 0;
        const int nk = n_live * cpt;
        // K-iteration state (wave-uniform) and the per-lane tap geometry
        int ci0 = 0, tr = 0, ts = 0, kbase = 0;
        bool tap_ok;
        unsigned voff1, voff2;          // byte offsets into x / x2 for the current tap
        auto set_tap = [&]() {
            int ih = 0, iw = 0;
            tap_ok = in_coord(tr, ts, ih, iw);
            const unsigned pix = tap_ok ? (unsigned)(ih * a.W + iw) : 0u;
            voff1 = ((unsigned)(n_b * a.c_in_split + b_row0) * (unsigned)HW + pix) * 4u;
            voff2 = ((unsigned)(n_b * c2 + b_row0) * (unsigned)HW + pix) * 4u;
        };
        auto next_tap = [&]() {          // advance (tr, ts, kbase) to the next live tap, if any
            do {
                if (++ts == a.KW) { ts = 0; ++tr; }
                if (tr >= a.KH) return;
                if (tap_live(tr, ts)) break;
                kbase += a.CiR;          // skipped tap: its weight rows are never touched
            } while (true);
        };
        bool ld_ok = false;             // validity of the tile currently held in rb[]
        unsigned ld_cmask = 0xffffffffu; // bit i: channel of rb[i] exists (only ever cleared when `padded`)
        auto load_tile = [&]() {
            // weights: rows kbase + a_row0 + i*A_ROWSTEP, 4 consecutive co per lane
            const float* wbase = a.wp + (size_t)kbase * a.CoP;
#pragma unroll
            for (int i = 0; i < A_PER; ++i)
                ra[i] = ldg_f32x4(wbase + (size_t)(i * A_ROWSTEP) * a.CoP, a_voff);
            // activations: channels ci0 + b_row0 + i*B_ROWSTEP at this lane's (tap-shifted) pixel
            const bool first = ci0 < a.c_in_split;
            const float* xbase = first ? a.x + (size_t)ci0 * HW : a.x2 + (size_t)(ci0 - a.c_in_split) * HW;
            const unsigned voff = first ? voff1 : voff2;
            if constexpr (!padded) {
#pragma unroll
                for (int i = 0; i < B_PER; ++i)
                    rb[i] = ldg_f32(xbase + (size_t)(i * B_ROWSTEP) * HW, voff);
            } else {
                // channels past Ci: read channel ci0 (always present) instead and zero the value at the LDS store
                const unsigned voff_c0 = voff - (unsigned)b_row0 * (unsigned)HW * 4u;
                unsigned cm = 0;
#pragma unroll
                for (int i = 0; i < B_PER; ++i) {
                    const bool cv = ci0 + b_row0 + i * B_ROWSTEP < a.Ci;
                    rb[i] = cv ? ldg_f32(xbase + (size_t)(i * B_ROWSTEP) * HW, voff) : ldg_f32(xbase, voff_c0);
                    cm |= cv ? (1u << i) : 0u;
                }
                ld_cmask = cm;
            }
            ld_ok = tap_ok;
            // advance the K iterator
            ci0 += BK;
            kbase += BK;
            if (ci0 >= a.CiR) {
                ci0 = 0;
                next_tap();
                if (tr < a.KH) set_tap();
            }
        };
        auto store_tile = [&](int buf) {
            if (a_active) {
#pragma unroll
                for (int i = 0; i < A_PER; ++i)
                    *reinterpret_cast<float4*>(&As[buf][a_row0 + i * A_ROWSTEP][a_q * 4]) = ra[i];
            }
#pragma unroll
            for (int i = 0; i < B_PER; ++i)
                Bs[buf][b_row0 + i * B_ROWSTEP][b_col] = (ld_ok && (!padded || ((ld_cmask >> i) & 1u))) ? rb[i] : 0.f;
        };

        if (!tap_live(0, 0)) {           // first live tap
            kbase += a.CiR;
            next_tap();
        }
        if (nk > 0) {
            set_tap();
            load_tile();
            store_tile(0);
        }
        __syncthreads();
        DYNMM_TRACE_MARK(1);
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) load_tile();        // global loads fly under the MFMAs below
            mfma_tile(buf);
            if (kt + 1 < nk) store_tile(buf ^ 1);
            __syncthreads();
        }
    } else {
        // ---------------- generic path: any Ci (stems Ci=1/3, gate Ci=8, dgrad of Co=40) ----------------
        constexpr int AS_PER = BK * TCO / 256, AS_ROWSTEP = 256 / TCO;
        const int a_col = t % TCO, as_row0 = t / TCO;
        const int co_a = co0 + a_col;
        const bool a_ok = co_a < a.Co;
        const float* xn = a.x + (size_t)n_b * a.c_in_split * HW;
        const float* x2n = a.x2 ? a.x2 + (size_t)n_b * c2 * HW : nullptr;
        float rs[AS_PER];
        const int nk = (a.K + BK - 1) / BK;
        auto load_tile = [&](int kt) {
            const int kb = kt * BK;
#pragma unroll
            for (int i = 0; i < AS_PER; ++i) {
                const int k = kb + as_row0 + i * AS_ROWSTEP;
                // packed rows are (tap * CiR + ci); CiR == Ci whenever this path runs on un-rounded channel counts
                rs[i] = (a_ok && k < a.K) ? a.wp[(size_t)((k / a.Ci) * a.CiR + (k % a.Ci)) * a.CoP + co_a] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < B_PER; ++i) {
                const int k = kb + b_row0 + i * B_ROWSTEP;
                float v = 0.f;
                if (k < a.K) {
                    const int tap = k / a.Ci;
                    const int ci = k - tap * a.Ci;
                    const int r = tap / a.KW, s = tap - r * a.KW;
                    int ih = 0, iw = 0;
                    if (in_coord(r, s, ih, iw)) {
                        const float* p = (ci < a.c_in_split) ? xn + (size_t)ci * HW
                                                             : x2n + (size_t)(ci - a.c_in_split) * HW;
                        v = p[ih * a.W + iw];
                    }
                }
                rb[i] = v;
            }
        };
        auto store_tile = [&](int buf) {
#pragma unroll
            for (int i = 0; i < AS_PER; ++i) As[buf][as_row0 + i * AS_ROWSTEP][a_col] = rs[i];
#pragma unroll
            for (int i = 0; i < B_PER; ++i) Bs[buf][b_row0 + i * B_ROWSTEP][b_col] = rb[i];
        };
        load_tile(0);
        store_tile(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            const int buf = kt & 1;
            if (kt + 1 < nk) load_tile(kt + 1);
            mfma_tile(buf);
            if (kt + 1 < nk) store_tile(buf ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue: scale/shift (bias or folded BN), residual, activation, ReLU-mask; NCHW store ----
    // Written so that NO memory wait sits between two stores: per-channel scale/shift are staged in LDS
    // (the operand tiles are dead), residual/mask values of a whole 32x32 accumulator tile are loaded
    // as one batch, then its 16 stores issue back to back.  (The first version interleaved
    // load -> wait -> store per element; on gfx9 vmcnt also counts stores, so every element paid a
    // full store round trip: 22 us per workgroup, measured with -DDYNMM_TRACE, ~20 % of the kernel.)
    DYNMM_TRACE_MARK(2);
    float* const sc_lds = &As[0][0][0];          // [TCO] scale, [TCO] shift  (BK*TCO >= 2*TCO)
    float* const sh_lds = sc_lds + TCO;
    {
        const float* __restrict__ scale = a.scale;
        const float* __restrict__ shift = a.shift;
        for (int i = t; i < TCO; i += 256) {
            const int co = co0 + i;
            const bool ok = co < a.Co;
            sc_lds[i] = (scale && ok) ? scale[co] : 1.f;
            sh_lds[i] = (shift && ok) ? shift[co] : 0.f;
        }
    }
    __syncthreads();
    const float* __restrict__ res_p = a.residual;
    const float* __restrict__ mask_p = a.mask;
    float* __restrict__ y1_p = a.y;
    float* __restrict__ y2_p = a.y2;
    const bool has_res = res_p != nullptr, has_mask = mask_p != nullptr;
    const unsigned co_split = (unsigned)a.c_out_split;
    const unsigned co_rest = (unsigned)a.Co - co_split;
    const bool single_out = co_split >= (unsigned)a.Co;
    const int act = a.act;
    const unsigned row_bytes = (unsigned)HoWo * 4u;          // byte stride between channels

    // half of a 32x32 accumulator tile (8 values per lane) at a time: v = acc*scale + shift, then
    // mask/residual/activation, then 8 stores.  Half tiles + scheduling barriers keep the epilogue's
    // register footprint under the K loop's, so the kernel stays at 6 waves/SIMD.
    // FAST: the tile lies inside [0,Co) of a single output tensor -> uniform base + 32-bit offsets,
    // no per-element predicates.
    auto finish = [&](float (&v)[8], const float (&rv)[8], const float (&mv)[8]) {
        if (DGRAD) {
            // dx = (dgrad term) * [mask > 0] + accum : ReLU backward of the producer of x and the
            // residual-branch gradient, both fused into the store
            if (has_mask) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = mv[j] > 0.f ? v[j] : 0.f;
            }
            if (has_res) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += rv[j];
            }
        } else {
            if (has_res) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += rv[j];
            }
            if (act == DYNMM_ACT_RELU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
            } else if (act == DYNMM_ACT_TANH) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = tanhf(v[j]);
            }
        }
    };
    auto scaled = [&](float (&v)[8], int mi, int ni, int cl0, int h) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int j4 = 2 * h + q;
            const float4 sc = *reinterpret_cast<const float4*>(&sc_lds[cl0 + 8 * j4]);
            const float4 sh = *reinterpret_cast<const float4*>(&sh_lds[cl0 + 8 * j4]);
            v[4 * q + 0] = acc[mi][ni][4 * j4 + 0] * sc.x + sh.x;
            v[4 * q + 1] = acc[mi][ni][4 * j4 + 1] * sc.y + sh.y;
            v[4 * q + 2] = acc[mi][ni][4 * j4 + 2] * sc.z + sh.z;
            v[4 * q + 3] = acc[mi][ni][4 * j4 + 3] * sc.w + sh.w;
        }
    };

    // batched path: one output tensor, and channel validity decidable per group of 4 consecutive channels
    const bool full_co = co0 + TCO <= a.Co;
    const bool fast_out = single_out && (full_co || (a.Co & 3) == 0);
#pragma unroll
    for (int ni = 0; ni < MPIX; ++ni) {
        const int m = pix0 + wave_pix * WPIX + ni * 32 + l31;
        if (m >= a.M) continue;
        int n_e, oh_e, ow_e, cls_e;
        pix_decode(m, n_e, oh_e, ow_e, cls_e);
        const unsigned n = (unsigned)n_e;
        const unsigned rem = (unsigned)(oh_e * a.Wo + ow_e);
#pragma unroll
        for (int mi = 0; mi < MCO; ++mi) {
            const int cl0 = wave_co * WCO + mi * 32 + 4 * khalf;          // tile-local channel of j = 0
            if (fast_out) {
                const unsigned off0 = ((n * (unsigned)a.Co + (unsigned)(co0 + cl0)) * (unsigned)HoWo + rem) * 4u;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float v[8], rv[8], mv[8];
                    // element e of this half: j = 8h + e  ->  channel offset (e & 3) + 8 * (2h + (e >> 2))
                    // the half's two groups of 4 consecutive channels; in a partial channel tile a group
                    // is either entirely inside [0,Co) or entirely outside (Co % 4 == 0)
                    const bool g0 = full_co || co0 + cl0 + 8 * (2 * h) < a.Co;
                    const bool g1 = full_co || co0 + cl0 + 8 * (2 * h + 1) < a.Co;
#pragma unroll
                    for (int e = 0; e < 8; ++e) { mv[e] = 1.f; rv[e] = 0.f; }
                    if (has_mask) {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if ((e >> 2) ? g1 : g0)
                                mv[e] = ldg_f32(mask_p, off0 + (unsigned)((e & 3) + 8 * (2 * h + (e >> 2))) * row_bytes);
                    }
                    if (has_res) {
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if ((e >> 2) ? g1 : g0)
                                rv[e] = ldg_f32(res_p, off0 + (unsigned)((e & 3) + 8 * (2 * h + (e >> 2))) * row_bytes);
                    }
                    scaled(v, mi, ni, cl0, h);
                    finish(v, rv, mv);
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if ((e >> 2) ? g1 : g0)
                            *reinterpret_cast<float*>(reinterpret_cast<char*>(y1_p) +
                                                      (off0 + (unsigned)((e & 3) + 8 * (2 * h + (e >> 2))) * row_bytes)) = v[e];
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                // channel tail (Co % TCO != 0) and/or two output tensors (dgrad of a dual-input conv):
                // rare shapes; element-at-a-time keeps the register budget of the kernel at the fast path's
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const unsigned co = (unsigned)(co0 + cl0 + (j & 3) + 8 * (j >> 2));
                    if (co >= (unsigned)a.Co) continue;
                    size_t idx;
                    float* out;
                    if (single_out || co < co_split) {
                        idx = ((size_t)n * co_split + co) * HoWo + rem;
                        out = y1_p;
                    } else {
                        idx = ((size_t)n * co_rest + (co - co_split)) * HoWo + rem;
                        out = y2_p;
                    }
                    const int cl = cl0 + (j & 3) + 8 * (j >> 2);
                    float x = acc[mi][ni][j] * sc_lds[cl] + sh_lds[cl];
                    if (DGRAD) {
                        if (has_mask) x = mask_p[idx] > 0.f ? x : 0.f;
                        if (has_res) x += res_p[idx];
                    } else {
                        if (has_res) x += res_p[idx];
                        x = act_fwd(x, act);
                    }
                    out[idx] = x;
                }
            }
        }
    }
#ifdef DYNMM_TRACE
    __builtin_amdgcn_s_waitcnt(0);          // stores acknowledged
    __syncthreads();
    DYNMM_TRACE_MARK(3);
    if (g_trace && threadIdx.x == 0) g_trace[(size_t)blockIdx.x * 6 + 5] = clock64();   // shader clock at exit
#endif
}

#ifdef DYNMM_TRACE
}  // namespace dynmm
extern "C" int dynmm_debug_set_trace(void* p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(dynmm::g_trace), &p, sizeof(p));
}
namespace dynmm {
#endif

// rows per filter tap of a packed weight / K-steps of the fast path (see pack_weight_kernel)
static inline int round_k(int c) { return (c >= 8 && c % 16 != 0) ? ((c + 15) & ~15) : c; }

template <bool DGRAD>
static int launch_igemm(IgemmArgs& a, hipStream_t st) {
    const bool dual_in = a.x2 != nullptr;
    a.CiR = round_k(a.Ci);
    const bool generic = (a.CiR % 16 != 0) || (dual_in && (a.c_in_split % 16 != 0)) ||
                         ((reinterpret_cast<uintptr_t>(a.wp) & 15u) != 0);
    a.M = a.N * a.Ho * a.Wo;
    a.K = a.KH * a.KW * a.Ci;
    a.CoP = (a.Co + 3) & ~3;
    a.subpix = (DGRAD && !generic && a.SH * a.SW > 1 && a.Ho % a.SH == 0 && a.Wo % a.SW == 0) ? 1 : 0;
    if (!generic && launch_igemm_v5(a, DGRAD, st)) {      // stride-1 same-padded 1x1 / 3x1 / 1x3 / 3x3: the operand-ring kernels
        DYNMM_LAUNCH_CHECK();
        return DYNMM_OK;
    }
#define DYNMM_IGEMM_LAUNCH(TCO, TPIX, WCO, WPIX)                                               \
    do {                                                                                       \
        a.n_co_tiles = ceil_div(a.Co, TCO);                                                    \
        a.n_pix_tiles = ceil_div(a.M, TPIX);                                                   \
        dim3 grid((unsigned)(a.n_co_tiles * a.n_pix_tiles));                                   \
        if (generic)                                                                           \
            hipLaunchKernelGGL((conv_igemm_kernel<TCO, TPIX, WCO, WPIX, DGRAD, 2>), grid,      \
                               dim3(256), 0, st, a);                                           \
        else if (a.CiR != a.Ci)                                                                \
            hipLaunchKernelGGL((conv_igemm_kernel<TCO, TPIX, WCO, WPIX, DGRAD, 1>), grid,      \
                               dim3(256), 0, st, a);                                           \
        else                                                                                   \
            hipLaunchKernelGGL((conv_igemm_kernel<TCO, TPIX, WCO, WPIX, DGRAD, 0>), grid,      \
                               dim3(256), 0, st, a);                                           \
    } while (0)
    if (a.Co > 64) {
        // Measured on MI355X (batch 32): the 128co x 64pix tile (32 accumulator VGPRs, ~5 resident
        // workgroups/CU) beats 128x128 (64 acc, 3/CU) on every stage — 93 vs 83 TFLOP/s at C=128,
        // 98 vs 73 at C=256, 80 vs 56 at C=512: this kernel is limited by latency hiding / residency
        // rounds, not by MFMA issue, so more, smaller workgroups win.  (BK=32 and a 2-deep register
        // prefetch both lost for the same reason: they cost occupancy.)
        if (ceil_div(a.Co, 128) * ceil_div(a.M, 64) < 3 * 256)
            // fewer than 3 tiles per CU (C=512 @ 15x20: 600 tiles): halve the tile so the work spreads
            // evenly — 188.6 -> 174.6 us; at >= 4.7 tiles/CU the 128x64 tile's lower L2 traffic wins.
            DYNMM_IGEMM_LAUNCH(128, 32, 32, 32);
        else
            DYNMM_IGEMM_LAUNCH(128, 64, 64, 32);
    } else if (a.Co > 32) {
        DYNMM_IGEMM_LAUNCH(64, 128, 32, 64);         // 77/84 TFLOP/s (fwd/dgrad) at C=64 against 67/77 for a 64x256 tile
    } else {
        DYNMM_IGEMM_LAUNCH(32, 256, 32, 64);
    }
#undef DYNMM_IGEMM_LAUNCH
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

// ------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------
template <int TCO, int TK, int WCO, int WK, bool DUAL, bool FAST>
__global__ void __launch_bounds__(256, 2) conv_wgrad_kernel(const WgradArgs a_in, const WgradGroup grp) {
    WgradArgs a = a_in;
    static_assert(!(DUAL && FAST), "FAST is the single-input, Ci % 64 == 0 specialisation");
    constexpr int BP = 32, LDP = BP + 1;     // +1 pad: column reads of the [row][pixel] tiles
    constexpr int MCO = WCO / 32, MK = WK / 32;
    constexpr int WAVES_K = TK / WK;
    static_assert((TCO / WCO) * WAVES_K == 4, "4 waves per workgroup");
    constexpr int G_PER = TCO / 8, X_PER = TK / 8;

    __shared__ float Gs[TCO][LDP];
    __shared__ float Xs[TK][LDP];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wave_co = wave / WAVES_K, wave_k = wave % WAVES_K;
    const int khalf = lane >> 5, l31 = lane & 31;

    // XCD-aware order (1-D grid): consecutive logical ids stay on one XCD and walk the (co, k) tiles of ONE pixel
    // split first, so the dy / x tiles of that split are fetched from HBM once per XCD and re-used out of its L2 by
    // the other tiles (PMC, round 2: with blockIdx.x = tile the k-tiles of a split sat on different XCDs and every one
    // of them re-fetched dy: 2 x FETCH_SIZE = 2.2 x the algorithmic bytes).
    const int n_tiles = a.n_co_tiles * a.n_k_tiles;
    int lin = xcd_remap(blockIdx.x, gridDim.x);
    if (grp.nprob > 1) {                                     // grouped launch: problem p owns workgroups [p*per, (p+1)*per)
        const int p = lin / grp.per;
        lin -= p * grp.per;
        a.x = grp.x[p];
        a.dy = grp.dy[p];
        a.out = grp.out[p];
        a.out_bias = grp.out_bias[p];
    }
    const int tile = lin % n_tiles;
    const int co0 = (tile % a.n_co_tiles) * TCO;
    const int k0 = (tile / a.n_co_tiles) * TK;
    const int split = lin / n_tiles;
#ifdef DYNMM_TRACE
    const size_t trace_row = (size_t)blockIdx.x * 6;
    if (g_trace && threadIdx.x == 0) { g_trace[trace_row] = wall_clock64(); g_trace[trace_row + 1] = g_trace[trace_row]; g_trace[trace_row + 4] = clock64(); }
#endif

    const int HW = a.H * a.W, HoWo = a.Ho * a.Wo;
    const int p = t & 31, rg = t >> 5;
    const int c2 = a.Ci - a.c_split;

    // per-thread row descriptors, fixed for the whole pixel loop:
    //   dy rows: byte offset of channel min(co, Co-1) (rows past Co are junk and never stored)
    //   x rows : element offset ci*HW + r*W + s and packed (r, s, second-input flag, k<K flag)
    unsigned goff[G_PER];
#pragma unroll
    for (int i = 0; i < G_PER; ++i) {
        int co = co0 + rg + 8 * i;
        if (co > a.Co - 1) co = a.Co - 1;
        goff[i] = (unsigned)co * (unsigned)HoWo * 4u;
    }
    int xoff[X_PER], xrs[X_PER];
#pragma unroll
    for (int i = 0; i < X_PER; ++i) {
        const int k = k0 + rg + 8 * i;
        if (k < a.K) {
            const int tap = k / a.Ci;
            int ci = k - tap * a.Ci;
            const int r = tap / a.KW, s = tap - r * a.KW;
            int second = 0;
            if (DUAL && ci >= a.c_split) { ci -= a.c_split; second = 1; }
            xoff[i] = ci * HW + r * a.W + s;
            xrs[i] = r | (s << 8) | (second << 16) | (1 << 17);
        } else {
            xoff[i] = 0;
            xrs[i] = 0;
        }
    }

    // FAST (Ci % 64 == 0, one input tensor): this thread's rows [0,8) lie in one filter tap and rows
    // [8,16) in one tap (TK rows = TK/64 groups of 64, taps change at multiples of Ci), so padding validity and
    // the tap shift are evaluated twice per step instead of 16 times, and an invalid group simply reads
    // the un-shifted (always mapped) position: no per-row select.  ~100 VALU per step instead of ~400 —
    // with only 2 waves/SIMD that address arithmetic was not hidden behind the other wave's MFMAs.
    constexpr int NG = X_PER / 8;          // groups of 8 rows (= 64 k values) sharing one filter tap
    int f_r[NG], f_s[NG];
    bool f_ok[NG];
    unsigned xoffb[X_PER];
#pragma unroll
    for (int gidx = 0; gidx < NG; ++gidx) { f_r[gidx] = 0; f_s[gidx] = 0; f_ok[gidx] = false; }
    if (FAST) {
#pragma unroll
        for (int gidx = 0; gidx < NG; ++gidx) {
            const int k = k0 + rg + 64 * gidx;
            if (k < a.K) {
                const int tap = k / a.Ci;
                f_r[gidx] = tap / a.KW;
                f_s[gidx] = tap - f_r[gidx] * a.KW;
                f_ok[gidx] = true;
            }
        }
#pragma unroll
        for (int i = 0; i < X_PER; ++i) xoffb[i] = (xrs[i] >> 17) ? (unsigned)xoff[i] * 4u : 0u;
    }
    // running pixel position of this lane (FAST): m = n * HoWo + rem
    int f_m = split * a.steps_per_split * BP + p;
    int f_n = f_m / HoWo;
    int f_rem = f_m - f_n * HoWo;

    // One register stage of global prefetch (a second stage was measured: no gain — the kernel is not
    // bound by global-load latency but by LDS-fragment latency inside the MFMA block, see below).
    struct Stage {
        float g[G_PER], x[X_PER];
        unsigned vmask;      // validity of the held tile: bit i = x row i, bit 31 = pixel < M
    };
    Stage r0;
    r0.vmask = 0;
    const int step_begin = split * a.steps_per_split;
    const int total_steps = (a.M + BP - 1) / BP;
    const int step_end = min(total_steps, step_begin + a.steps_per_split);

    // Loads are issued unconditionally from clamped (always mapped) addresses and the zero-fill of
    // padding / out-of-range lanes is deferred to store_step(): nothing consumes a loaded value
    // before the MFMA block, so the whole global-load latency hides under the MFMAs.
    auto load_step = [&](int st, Stage& r) {
        if (FAST) {
            const bool ok = f_m < a.M;
            const int n = ok ? f_n : 0, rem = ok ? f_rem : 0;
            const int oh = a.magic_wo ? (int)__umulhi((unsigned)rem, a.magic_wo) : rem / a.Wo;
            const int ow = rem - oh * a.Wo;
            const unsigned gv = ((unsigned)(n * a.Co) * (unsigned)HoWo + (unsigned)rem) * 4u;
#pragma unroll
            for (int i = 0; i < G_PER; ++i) r.g[i] = ldg_f32(a.dy, gv + goff[i]);
            const int ihb = oh * a.SH - a.PH, iwb = ow * a.SW - a.PW;
            const int img = n * a.Ci * HW;
            const int shift = ihb * a.W + iwb;
            unsigned base[NG];
            unsigned vm = ok ? 0x80000000u : 0u;
#pragma unroll
            for (int gidx = 0; gidx < NG; ++gidx) {
                const bool v = ok && f_ok[gidx] && (unsigned)(ihb + f_r[gidx]) < (unsigned)a.H &&
                               (unsigned)(iwb + f_s[gidx]) < (unsigned)a.W;
                base[gidx] = (unsigned)(img + (v ? shift : 0)) * 4u;     // invalid tap: un-shifted, always mapped
                vm |= v ? (1u << gidx) : 0u;
            }
#pragma unroll
            for (int i = 0; i < X_PER; ++i) r.x[i] = ldg_f32(a.x, base[i >> 3] + xoffb[i]);
            r.vmask = vm;
            f_m += BP;
            f_rem += BP;
            while (f_rem >= HoWo) { f_rem -= HoWo; ++f_n; }
            return;
        }
        const int m = st * BP + p;
        const bool ok = m < a.M;
        int n = 0, oh = 0, ow = 0, rem = 0;
        if (ok) {
            n = m / HoWo;
            rem = m - n * HoWo;
            oh = rem / a.Wo;
            ow = rem - oh * a.Wo;
        }
        const unsigned gv = ((unsigned)(n * a.Co) * (unsigned)HoWo + (unsigned)rem) * 4u;
#pragma unroll
        for (int i = 0; i < G_PER; ++i) r.g[i] = ldg_f32(a.dy, gv + goff[i]);
        const int ihb = oh * a.SH - a.PH, iwb = ow * a.SW - a.PW;
        const int xb1 = n * a.c_split * HW + ihb * a.W + iwb;     // may be negative at the borders
        const int xb2 = DUAL ? n * c2 * HW + ihb * a.W + iwb : 0;
        unsigned vm = ok ? 0x80000000u : 0u;
#pragma unroll
        for (int i = 0; i < X_PER; ++i) {
            const int d = xrs[i];
            const int r_ = d & 0xff, s_ = (d >> 8) & 0xff;
            const bool valid = ok && (d >> 17) && (unsigned)(ihb + r_) < (unsigned)a.H &&
                               (unsigned)(iwb + s_) < (unsigned)a.W;
            const bool second = DUAL && ((d >> 16) & 1);
            const int e = (second ? xb2 : xb1) + xoff[i];
            const unsigned vo = valid ? (unsigned)e * 4u : 0u;
            r.x[i] = ldg_f32(second ? a.x2 : a.x, vo);
            vm |= valid ? (1u << i) : 0u;
        }
        r.vmask = vm;
    };
    auto store_step = [&](const Stage& r) {
        const bool ok = (r.vmask >> 31) != 0;
#pragma unroll
        for (int i = 0; i < G_PER; ++i) Gs[rg + 8 * i][p] = ok ? r.g[i] : 0.f;
        if (FAST) {
#pragma unroll
            for (int i = 0; i < X_PER; ++i) Xs[rg + 8 * i][p] = ((r.vmask >> (i >> 3)) & 1u) ? r.x[i] : 0.f;
        } else {
#pragma unroll
            for (int i = 0; i < X_PER; ++i) Xs[rg + 8 * i][p] = ((r.vmask >> i) & 1u) ? r.x[i] : 0.f;
        }
    };

    f32x16 acc[MCO][MK];
#pragma unroll
    for (int mi = 0; mi < MCO; ++mi)
#pragma unroll
        for (int ni = 0; ni < MK; ++ni)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[mi][ni][j] = 0.f;

    // MFMA block of one step: 16 k-pairs x (MCO x MK) MFMAs.  The A/B fragments of k-pair q+1 are read
    // from LDS BEFORE the MFMAs of k-pair q issue (two fragment sets).
    // Measured and rejected on MI355X (-DDYNMM_TRACE timelines, scratch/trace/): a second register stage
    // of global prefetch, double-buffered LDS with one barrier per step and all loads / LDS writes
    // interleaved between the MFMAs (with hand-counted vmcnt via asm loads, because hipcc drains vmcnt
    // to 0 before every new batch of loads in a loop), fragments two k-pairs ahead: all within +-3 % of
    // this simple loop.  The shader clock sits at ~1.93 GHz in this kernel (2.38 GHz in a pure-MFMA
    // loop, scratch/mfma/peak.hip), i.e. the sustainable fp32 MFMA rate here is ~126 TFLOP/s, not 157.
    auto frags = [&](int pp, float (&af)[MCO], float (&bf)[MK]) {
#pragma unroll
        for (int mi = 0; mi < MCO; ++mi) af[mi] = Gs[wave_co * WCO + mi * 32 + l31][2 * pp + khalf];
#pragma unroll
        for (int ni = 0; ni < MK; ++ni) bf[ni] = Xs[wave_k * WK + ni * 32 + l31][2 * pp + khalf];
    };
    auto mfmas = [&](const float (&af)[MCO], const float (&bf)[MK]) {
#pragma unroll
        for (int mi = 0; mi < MCO; ++mi)
#pragma unroll
            for (int ni = 0; ni < MK; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi], bf[ni], acc[mi][ni], 0, 0, 0);
    };
    if (step_begin < step_end) {
        load_step(step_begin, r0);
        store_step(r0);
    }
    __syncthreads();
    // bias gradient (optional): the k-tile-0 workgroups also sum their dy tile over the pixels — the tile
    // is in LDS anyway, so the separate full pass over dy that the bias gradient used to cost disappears
    const bool do_bias = a.out_bias != nullptr && (tile / a.n_co_tiles) == 0;
    float bsum = 0.f;
    for (int st = step_begin; st < step_end; ++st) {
        if (st + 1 < step_end) load_step(st + 1, r0);
        if (do_bias && t < TCO) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int q = 0; q < BP; q += 2) { s0 += Gs[t][q]; s1 += Gs[t][q + 1]; }
            bsum += s0 + s1;
        }
        float af0[MCO], bf0[MK], af1[MCO], bf1[MK];
        frags(0, af0, bf0);
#pragma unroll
        for (int pp = 0; pp < BP / 2; pp += 2) {
            frags(pp + 1, af1, bf1);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(af0, bf0);
            __builtin_amdgcn_sched_barrier(0);
            if (pp + 2 < BP / 2) frags(pp + 2, af0, bf0);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(af1, bf1);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        if (st + 1 < step_end) store_step(r0);
        __syncthreads();
    }

#ifdef DYNMM_TRACE
    if (g_trace && threadIdx.x == 0) g_trace[trace_row + 2] = wall_clock64();
#endif
    if (do_bias && t < TCO && co0 + t < a.Co) a.out_bias[(size_t)split * a.Co + co0 + t] = bsum;
    const int KHKW = a.KH * a.KW;
    float* out = a.out + (size_t)split * a.Co * a.K;
#pragma unroll
    for (int ni = 0; ni < MK; ++ni) {
        const int k = k0 + wave_k * WK + ni * 32 + l31;
        if (k >= a.K) continue;
        const int tap = k / a.Ci;
        const int ci = k - tap * a.Ci;
#pragma unroll
        for (int mi = 0; mi < MCO; ++mi)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int co = co0 + wave_co * WCO + mi * 32 + (j & 3) + 8 * (j >> 2) + 4 * khalf;
                if (co < a.Co) out[((size_t)co * a.Ci + ci) * KHKW + tap] = acc[mi][ni][j];
            }
    }
#ifdef DYNMM_TRACE
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    if (g_trace && threadIdx.x == 0) { g_trace[trace_row + 3] = wall_clock64(); g_trace[trace_row + 5] = clock64(); }
#endif
}

// ------------------------------------------------------------------------------------------------
// weight gradient, vectorised operand delivery ("v4"): the specialisation that carries the C >= 128 stages
// ------------------------------------------------------------------------------------------------
// Same GEMM as conv_wgrad_kernel (dW[co][k] = sum_pix dY[co][pix] * Xcol[k][pix], 32 pixels per step), rebuilt
// around 16-byte accesses end to end:
//   * global: one lane fetches 4 consecutive pixels of a row (dwordx4): 4 + 4(+4) loads per lane and step
//     instead of 16 + 16 dword gathers (the old loader's vector-memory issue was ~1/2 of the MFMA time);
//     a filter tap shifted by +-1 along W is the aligned quad plus one scalar neighbour, so every access
//     stays aligned and inside the tensor (taps shifted along H move whole rows);
//   * LDS: [row][36] tiles (144-byte rows: 16-byte aligned, and 9*r mod 16 is a permutation, so both the
//     ds_write_b128 of the loader and the ds_read_b128 of the fragments are conflict-free);
//   * MFMA k-pairing: instruction pp of a step contracts pixels {pp, pp+16} instead of {2pp, 2pp+1}; a
//     lane's 16 operands of a row are then CONTIGUOUS — 4 ds_read_b128 instead of 16 ds_read_b32;
//   * 8 waves (2 co x 4 k, wave tile 64 x 32, 32 accumulator VGPRs) on a double-buffered tile, one barrier
//     per step, <= 128 VGPRs: 2 workgroups = 4 waves per SIMD instead of 2 (nothing hid a wave's non-MFMA
//     instructions before).
// Eligible: one input tensor, Ci % 64 == 0, Co > 64, stride 1 along W with 'same' padding and KW in {1, 3},
// W % 4 == 0, (Ho*Wo) % 4 == 0, 16-byte aligned x / dy.  Everything else stays on conv_wgrad_kernel.
template <int TCO, int TK, int NBUF>
__global__ void __launch_bounds__(512, 4) conv_wgrad_v4_kernel(const WgradArgs a_in, const WgradGroup grp) {
    WgradArgs a = a_in;
    // NBUF = 2: double-buffered tiles, one barrier per step (72 KB of LDS per workgroup: fastest when the kernel has
    // the GPU to itself).  NBUF = 1: 36 KB, two barriers per step — leaves LDS for the workgroups of the other
    // streams' kernels (the training step runs the weight gradients beside the dgrad / BatchNorm chain).
    constexpr int BP = 32, LD = 36;
    constexpr int WAVES_K = TK / 32;
    static_assert(TCO == 128 && TK == 128, "8 waves: 2 (co) x 4 (k)");
    constexpr int MCO = 2;
    __shared__ __attribute__((aligned(16))) float Gs[NBUF][TCO][LD];
    __shared__ __attribute__((aligned(16))) float Xs[NBUF][TK][LD];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
#ifdef DYNMM_TRACE
    const size_t trace_row = (size_t)blockIdx.x * 6;
    if (g_trace && threadIdx.x == 0) { g_trace[trace_row] = wall_clock64(); g_trace[trace_row + 1] = g_trace[trace_row]; g_trace[trace_row + 4] = clock64(); }
#endif
    const int wave_co = wave / WAVES_K, wave_k = wave % WAVES_K;
    const int khalf = lane >> 5, l31 = lane & 31;
    const int n_tiles = a.n_co_tiles * a.n_k_tiles;          // XCD-aware order, see conv_wgrad_kernel
    int lin = xcd_remap(blockIdx.x, gridDim.x);
    if (grp.nprob > 1) {
        const int p = lin / grp.per;
        lin -= p * grp.per;
        a.x = grp.x[p];
        a.dy = grp.dy[p];
        a.out = grp.out[p];
        a.out_bias = grp.out_bias[p];
    }
    const int tile = lin % n_tiles;
    const int co0 = (tile % a.n_co_tiles) * TCO;
    const int k0 = (tile / a.n_co_tiles) * TK;
    const int split = lin / n_tiles;
    const int HW = a.H * a.W, HoWo = a.Ho * a.Wo;

    // loader coordinates: lane q owns pixels [4q, 4q+4) of a step; rows rr and rr + 64
    const int q = t & 7, rr = t >> 3;
    unsigned goff[2];                    // byte offset of the dy row (channel clamped: rows past Co are never stored)
    unsigned xoff[2];                    // byte offset ci*HW of the x row
    int f_r[2], f_dx[2];                 // filter row of the group's tap; its shift along W (-1, 0, +1)
    bool f_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int co = co0 + rr + 64 * i;
        if (co > a.Co - 1) co = a.Co - 1;
        goff[i] = (unsigned)co * (unsigned)HoWo * 4u;
        const int k = k0 + rr + 64 * i;
        f_ok[i] = k < a.K;
        const int kk = f_ok[i] ? k : 0;
        const int tap = kk / a.Ci;
        const int ci = kk - tap * a.Ci;
        f_r[i] = tap / a.KW;
        f_dx[i] = (tap - f_r[i] * a.KW) - a.PW;
        xoff[i] = (unsigned)ci * (unsigned)HW * 4u;
    }
    const int step_begin = split * a.steps_per_split;
    const int total_steps = (a.M + BP - 1) / BP;
    const int step_end = min(total_steps, step_begin + a.steps_per_split);
    int f_m = step_begin * BP + 4 * q;               // first pixel of this lane's quad
    int f_n = f_m / HoWo;
    int f_rem = f_m - f_n * HoWo;

    // staging registers of the next step's tile (named scalars, not arrays: they must stay in VGPRs)
    float4 rg0, rg1, rx0, rx1;
    float rs0 = 0.f, rs1 = 0.f;
    unsigned vmask = 0;       // bit 0/1: x group valid; bit 2/3: neighbour valid; bit 31: quad < M
    auto load_x = [&](int i, bool ok, unsigned img, int ihb, int ow, float4& rx, float& rs) __attribute__((always_inline)) -> unsigned {
        const int ih = ihb + f_r[i];
        const bool v = ok && f_ok[i] && (unsigned)ih < (unsigned)a.H;
        const unsigned row = img + xoff[i] + (unsigned)((v ? ih : 0) * a.W + ow) * 4u;   // invalid: row 0, mapped
        rx = ldg_f32x4(a.x, row);
        // neighbour along W for a shifted tap (wave-uniform branch: a wave's rows lie in one 64-group)
        bool nv = false;
        float s = 0.f;
        if (f_dx[i] < 0) {
            nv = v && ow > 0;
            s = ldg_f32(a.x, row - (nv ? 4u : 0u));
        } else if (f_dx[i] > 0) {
            nv = v && ow + 4 < a.W;
            s = ldg_f32(a.x, row + (nv ? 16u : 0u));
        }
        rs = s;
        return (v ? (1u << i) : 0u) | (nv ? (4u << i) : 0u);
    };
    auto load_step = [&]() __attribute__((always_inline)) {
        const bool ok = f_m < a.M;
        const int n = ok ? f_n : 0, rem = ok ? f_rem : 0;
        const int oh = a.magic_wo ? (int)__umulhi((unsigned)rem, a.magic_wo) : rem / a.Wo;
        const int ow = rem - oh * a.Wo;
        const unsigned gv = ((unsigned)(n * a.Co) * (unsigned)HoWo + (unsigned)rem) * 4u;
        rg0 = ldg_f32x4(a.dy, gv + goff[0]);
        rg1 = ldg_f32x4(a.dy, gv + goff[1]);
        const int ihb = oh * a.SH - a.PH;
        const unsigned img = (unsigned)(n * a.Ci) * (unsigned)HW * 4u;
        unsigned vm = ok ? 0x80000000u : 0u;
        vm |= load_x(0, ok, img, ihb, ow, rx0, rs0);
        vm |= load_x(1, ok, img, ihb, ow, rx1, rs1);
        vmask = vm;
        f_m += BP;
        f_rem += BP;
        while (f_rem >= HoWo) { f_rem -= HoWo; ++f_n; }
    };
    auto put_x = [&](int buf, int i, float4 v, float sraw) __attribute__((always_inline)) {
        const float s = ((vmask >> (2 + i)) & 1u) ? sraw : 0.f;
        float e0 = v.x, e1 = v.y, e2 = v.z, e3 = v.w;
        if (f_dx[i] < 0) { e3 = e2; e2 = e1; e1 = e0; e0 = s; }
        else if (f_dx[i] > 0) { e0 = e1; e1 = e2; e2 = e3; e3 = s; }
        const bool ok = ((vmask >> i) & 1u) != 0;      // component-wise selects: a ternary on float4 goes through scratch
        *reinterpret_cast<float4*>(&Xs[buf][rr + 64 * i][4 * q]) =
            make_float4(ok ? e0 : 0.f, ok ? e1 : 0.f, ok ? e2 : 0.f, ok ? e3 : 0.f);
    };
    auto store_step = [&](int buf) __attribute__((always_inline)) {
        const bool ok = (vmask >> 31) != 0;
        *reinterpret_cast<float4*>(&Gs[buf][rr][4 * q]) =
            make_float4(ok ? rg0.x : 0.f, ok ? rg0.y : 0.f, ok ? rg0.z : 0.f, ok ? rg0.w : 0.f);
        *reinterpret_cast<float4*>(&Gs[buf][rr + 64][4 * q]) =
            make_float4(ok ? rg1.x : 0.f, ok ? rg1.y : 0.f, ok ? rg1.z : 0.f, ok ? rg1.w : 0.f);
        put_x(buf, 0, rx0, rs0);
        put_x(buf, 1, rx1, rs1);
    };

    f32x16 acc[MCO];
#pragma unroll
    for (int mi = 0; mi < MCO; ++mi)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[mi][j] = 0.f;

    const bool do_bias = a.out_bias != nullptr && (tile / a.n_co_tiles) == 0;
    float bsum = 0.f;

    if (step_begin < step_end) {
        load_step();
        store_step(0);
    }
    __syncthreads();
    for (int st = step_begin; st < step_end; ++st) {
        const int buf = NBUF == 2 ? ((st - step_begin) & 1) : 0;
        const bool more = st + 1 < step_end;
        if (more) load_step();                         // global loads fly under the MFMAs below
        if (do_bias && t < TCO) {
            float s0 = 0.f;
#pragma unroll
            for (int c4 = 0; c4 < BP; c4 += 4) {
                const float4 v = *reinterpret_cast<const float4*>(&Gs[buf][t][c4]);
                s0 += (v.x + v.y) + (v.z + v.w);
            }
            bsum += s0;
        }
        // this lane's 16 pixels of the step: [16*khalf, 16*khalf + 16)
        const float* ga = &Gs[buf][wave_co * 64 + l31][16 * khalf];
        const float* xb = &Xs[buf][wave_k * 32 + l31][16 * khalf];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float af[MCO][8], bf[8];
#pragma unroll
            for (int mi = 0; mi < MCO; ++mi) {
                const float4 u0 = *reinterpret_cast<const float4*>(ga + mi * 32 * LD + 8 * h);
                const float4 u1 = *reinterpret_cast<const float4*>(ga + mi * 32 * LD + 8 * h + 4);
                af[mi][0] = u0.x; af[mi][1] = u0.y; af[mi][2] = u0.z; af[mi][3] = u0.w;
                af[mi][4] = u1.x; af[mi][5] = u1.y; af[mi][6] = u1.z; af[mi][7] = u1.w;
            }
            {
                const float4 u0 = *reinterpret_cast<const float4*>(xb + 8 * h);
                const float4 u1 = *reinterpret_cast<const float4*>(xb + 8 * h + 4);
                bf[0] = u0.x; bf[1] = u0.y; bf[2] = u0.z; bf[3] = u0.w;
                bf[4] = u1.x; bf[5] = u1.y; bf[6] = u1.z; bf[7] = u1.w;
            }
#pragma unroll
            for (int pp = 0; pp < 8; ++pp)
#pragma unroll
                for (int mi = 0; mi < MCO; ++mi)
                    acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi][pp], bf[pp], acc[mi], 0, 0, 0);
        }
        if (NBUF == 1) __syncthreads();                // every wave is done reading the tile
        if (more) store_step(NBUF == 2 ? (buf ^ 1) : 0);
        __syncthreads();
    }

#ifdef DYNMM_TRACE
    if (g_trace && threadIdx.x == 0) g_trace[trace_row + 2] = wall_clock64();
#endif
    if (do_bias && t < TCO && co0 + t < a.Co) a.out_bias[(size_t)split * a.Co + co0 + t] = bsum;
    const int KHKW = a.KH * a.KW;
    float* out = a.out + (size_t)split * a.Co * a.K;
    const int k = k0 + wave_k * 32 + l31;
    if (k < a.K) {
        const int tap = k / a.Ci;
        const int ci = k - tap * a.Ci;
        // slabs: [co][k] (a half-wave stores 128 contiguous bytes; the reduction kernel applies the [co][ci][tap]
        // permutation once); without a split the final OIHW layout is written directly
        const size_t col = a.k_major_out ? (size_t)k : (size_t)ci * KHKW + tap;
        const size_t rowlen = a.k_major_out ? (size_t)a.K : (size_t)a.Ci * KHKW;
#pragma unroll
        for (int mi = 0; mi < MCO; ++mi)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int co = co0 + wave_co * 64 + mi * 32 + (j & 3) + 8 * (j >> 2) + 4 * khalf;
                if (co < a.Co) out[(size_t)co * rowlen + col] = acc[mi][j];
            }
    }
#ifdef DYNMM_TRACE
    __builtin_amdgcn_s_waitcnt(0);
    if (g_trace && threadIdx.x == 0) { g_trace[trace_row + 3] = wall_clock64(); g_trace[trace_row + 5] = clock64(); }
#endif
}

// out[(co*Ci + ci)*KHKW + tap] = sum_s slabs[s][co*K + tap*Ci + ci], fixed order (4 slab groups in flight per
// column, combined through LDS like reduce_slabs_kernel); each lane owns 4 consecutive ci (Ci % 4 == 0).
// Workgroups >= nb1 reduce the optional bias-gradient slabs (plain layout).
// grp (grouped launches, dynmm_conv2d_wgrad_group): blockIdx.y selects the problem's slabs / destinations.
struct ReduceGroup {
    const float* slabs[8];
    float* out[8];
    const float* slabs2[8];
    float* out2[8];
    int nprob;
};

__global__ void __launch_bounds__(256) reduce_slabs_perm_kernel(const float* __restrict__ slabs, float* __restrict__ out,
                                                                int n, int nslabs, int Ci, int KHKW, int K,
                                                                const float* __restrict__ slabs2,
                                                                float* __restrict__ out2, int n2, int nb1,
                                                                const ReduceGroup grp) {
    __shared__ float part[4][64][4];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if (grp.nprob > 0) {
        slabs = grp.slabs[blockIdx.y];
        out = grp.out[blockIdx.y];
        slabs2 = grp.slabs2[blockIdx.y];
        out2 = grp.out2[blockIdx.y];
    }
    int bx = blockIdx.x;
    const bool second = bx >= nb1;
    if (second) { bx -= nb1; slabs = slabs2; out = out2; n = n2; }
    const int i = (bx * 64 + tx) * 4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (i < n) {
#pragma unroll 4
        for (int k = ty; k < nslabs; k += 4) {
            float v[4];
            vload<4>(slabs + (size_t)k * n + i, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] += v[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) part[ty][tx][j] = acc[j];
    __syncthreads();
    if (ty == 0 && i < n) {
        float r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = ((part[0][tx][j] + part[1][tx][j]) + part[2][tx][j]) + part[3][tx][j];
        if (second) {
            vstore<4>(out + i, r);
        } else {
            const int co = i / K, kk = i - co * K;
            const int tap = kk / Ci, ci = kk - tap * Ci;
#pragma unroll
            for (int j = 0; j < 4; ++j) out[((size_t)co * Ci + ci + j) * KHKW + tap] = r[j];
        }
    }
}

struct WgradPlan {
    int tco, tk, n_co_tiles, n_k_tiles, splits, steps_per_split;
    int v6;        // conv_wgrad_v6.hip: tile tco x (3 taps x 64 ci), 16-pixel steps
    int bp;        // pixels per step
    int target;    // workgroups of one residency round
};

// (conv_wgrad_v6.hip)
bool wgrad_v6_shape_ok(const dynmm_conv_geom* g);
int wgrad_v6_tco(const dynmm_conv_geom* g);
int wgrad_v6_occupancy(const dynmm_conv_geom* g);
void launch_wgrad_v6(const WgradArgs& a, const WgradGroup& grp, dim3 grid, int occ, hipStream_t st);
// (conv_wgrad_wino_vt.hip) vertical taps in the Winograd form: the reduction runs over pair positions, 8 per step
bool wgrad_wino_vt_on(const dynmm_conv_geom* g);
int wgrad_wino_vt_units(const dynmm_conv_geom* g);
int wgrad_wino_vt_bp();

static void plan_splits(WgradPlan& p, const dynmm_conv_geom* g, int nprob) {
    const int units = (p.v6 && wgrad_wino_vt_on(g)) ? wgrad_wino_vt_units(g) : g->N * g->Ho * g->Wo;
    const int total_steps = ceil_div(units, p.bp);
    const int tiles = p.n_co_tiles * p.n_k_tiles * nprob;
    int splits = p.target / tiles;
    if (splits < 1) splits = 1;
    const int max_splits = ceil_div(total_steps, 256 / p.bp);   // >= 256 pixels per workgroup
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    p.steps_per_split = ceil_div(total_steps, splits);
    p.splits = ceil_div(total_steps, p.steps_per_split);
}

static WgradPlan plan_wgrad(const dynmm_conv_geom* g, bool allow_v6 = true) {
    WgradPlan p;
    p.v6 = 0;
    p.bp = 32;
    if (allow_v6 && wgrad_v6_shape_ok(g)) {
        constexpr int target6 = 0;
        p.v6 = 1;
        p.bp = wgrad_wino_vt_on(g) ? wgrad_wino_vt_bp() : 16;
        p.tco = wgrad_v6_tco(g);
        p.tk = 192;
        p.n_co_tiles = ceil_div(g->Co, p.tco);
        p.n_k_tiles = (g->KH == 3 && g->KW == 3 ? 3 : 1) * (g->Ci / 64);       // 3x3: one vertical tap per workgroup
#ifndef DYNMM_WGRAD_ROUNDS
#define DYNMM_WGRAD_ROUNDS 1
#endif
        p.target = target6 ? target6 : 256 * wgrad_v6_occupancy(g) * DYNMM_WGRAD_ROUNDS;
        plan_splits(p, g, 1);
        return p;
    }
    p.tco = g->Co > 64 ? 128 : (g->Co > 32 ? 64 : 32);
    const int K = g->KH * g->KW * g->Ci;
    // k-tile width for the 64-row configuration: 192 for the C=64 three-tap convs (K = 192: one tile instead
    // of 128 + a half-empty 128) and for 128 < K <= 192 (RGB stem, K = 147); 64 for K <= 64 (depth stem, K = 49)
    p.tk = 128;
    if (p.tco == 64) {
        if ((K % 192 == 0 && g->Ci % 64 == 0 && g->c_split == g->Ci) || (K > 128 && K <= 192)) p.tk = 192;
        else if (K <= 64) p.tk = 64;
    }
    p.n_co_tiles = ceil_div(g->Co, p.tco);
    p.n_k_tiles = ceil_div(K, p.tk);
    constexpr int target_env = 0;
    p.target = target_env ? target_env : 512;          // ONE residency round: 256 CUs x 2 workgroups (180 VGPRs)
    plan_splits(p, g, 1);
    return p;
}

// out[i] = sum_s slabs[s][i], deterministic.  64 columns (V floats each) x 4 slab-groups per
// workgroup: four independent load streams per column are in flight, partials meet in LDS and are
// added in a fixed order (no atomics => bit-reproducible weight gradients).
// Workgroups >= nb1 reduce an optional second region (the bias-gradient slabs) in the same launch.
template <int V>
__device__ __forceinline__ void reduce_slabs_body(const float* __restrict__ slabs, float* __restrict__ out, int n, int nslabs,
                                                  const float* __restrict__ slabs2, float* __restrict__ out2, int n2, int nb1) {
    __shared__ float part[4][64][V];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    int bx = blockIdx.x;
    if (bx >= nb1) { bx -= nb1; slabs = slabs2; out = out2; n = n2; }
    const int i = (bx * 64 + tx) * V;
    float acc[V];
#pragma unroll
    for (int j = 0; j < V; ++j) acc[j] = 0.f;
    if (i < n) {
#pragma unroll 4
        for (int k = ty; k < nslabs; k += 4) {
            float v[V];
            vload<V>(slabs + (size_t)k * n + i, v);
#pragma unroll
            for (int j = 0; j < V; ++j) acc[j] += v[j];
        }
    }
#pragma unroll
    for (int j = 0; j < V; ++j) part[ty][tx][j] = acc[j];
    __syncthreads();
    if (ty == 0 && i < n) {
        float r[V];
#pragma unroll
        for (int j = 0; j < V; ++j) r[j] = ((part[0][tx][j] + part[1][tx][j]) + part[2][tx][j]) + part[3][tx][j];
        vstore<V>(out + i, r);
    }
}

template <int V>
__global__ void __launch_bounds__(256) reduce_slabs_kernel(const float* __restrict__ slabs,
                                                           float* __restrict__ out, int n, int nslabs,
                                                           const float* __restrict__ slabs2,
                                                           float* __restrict__ out2, int n2, int nb1) {
    reduce_slabs_body<V>(slabs, out, n, nslabs, slabs2, out2, n2, nb1);
}

// the same for the problems of a grouped weight-gradient launch (equal shapes): blockIdx.y selects the problem — one launch
// instead of one per problem (the ModalityDynMM step: 89 of its 114 reduction launches)
template <int V>
__global__ void __launch_bounds__(256) reduce_slabs_group_kernel(ReduceGroup rg, int n, int nslabs, int n2, int nb1) {
    const int q = blockIdx.y;
    reduce_slabs_body<V>(rg.slabs[q], rg.out[q], n, nslabs, rg.slabs2[q], rg.out2[q], n2, nb1);
}

static void launch_reduce_slabs_group(const ReduceGroup& rg, int n, int nslabs, int n2, hipStream_t st) {
    bool v4 = (n % 4 == 0) && (n2 % 4 == 0);
    for (int i = 0; i < rg.nprob; ++i) {
        v4 = v4 && ((reinterpret_cast<uintptr_t>(rg.slabs[i]) | reinterpret_cast<uintptr_t>(rg.out[i])) & 15u) == 0;
        if (n2) v4 = v4 && ((reinterpret_cast<uintptr_t>(rg.slabs2[i]) | reinterpret_cast<uintptr_t>(rg.out2[i])) & 15u) == 0;
    }
    if (v4) {
        const int nb1 = ceil_div(n / 4, 64), nb2 = n2 ? ceil_div(n2 / 4, 64) : 0;
        hipLaunchKernelGGL(reduce_slabs_group_kernel<4>, dim3(nb1 + nb2, rg.nprob), dim3(256), 0, st, rg, n, nslabs, n2, nb1);
    } else {
        const int nb1 = ceil_div(n, 64), nb2 = n2 ? ceil_div(n2, 64) : 0;
        hipLaunchKernelGGL(reduce_slabs_group_kernel<1>, dim3(nb1 + nb2, rg.nprob), dim3(256), 0, st, rg, n, nslabs, n2, nb1);
    }
}

void launch_reduce_slabs(const float* slabs, float* out, int n, int nslabs, hipStream_t st,
                         const float* slabs2, float* out2, int n2) {
    const bool v4 = (n % 4 == 0) && ((reinterpret_cast<uintptr_t>(slabs) & 15u) == 0) &&
                    ((reinterpret_cast<uintptr_t>(out) & 15u) == 0) &&
                    (n2 == 0 || ((n2 % 4 == 0) && ((reinterpret_cast<uintptr_t>(slabs2) & 15u) == 0) &&
                                 ((reinterpret_cast<uintptr_t>(out2) & 15u) == 0)));
    if (v4) {
        const int nb1 = ceil_div(n / 4, 64), nb2 = n2 ? ceil_div(n2 / 4, 64) : 0;
        hipLaunchKernelGGL(reduce_slabs_kernel<4>, dim3(nb1 + nb2), dim3(256), 0, st, slabs, out, n, nslabs, slabs2,
                           out2, n2, nb1);
    } else {
        const int nb1 = ceil_div(n, 64), nb2 = n2 ? ceil_div(n2, 64) : 0;
        hipLaunchKernelGGL(reduce_slabs_kernel<1>, dim3(nb1 + nb2), dim3(256), 0, st, slabs, out, n, nslabs, slabs2,
                           out2, n2, nb1);
    }
}

// wf[(tap*CiR+ci)*CoP + co], wd[(tap*CoR+co)*CiP + ci]: CoP / CiP = row length rounded up to 4 floats, CiR / CoR =
// rows per tap (round_k: rounded up to 16 for 8 <= C, C % 16 != 0, so those convs run the one-tap-per-K-step fast
// path with a few zero rows instead of the element-wise generic loader).  Every padding element is written as 0.
__global__ void __launch_bounds__(256) pack_weight_kernel(const float* __restrict__ w,
                                                          float* __restrict__ wf,
                                                          float* __restrict__ wd,
                                                          int Co, int Ci, int KHKW, int CoP, int CiP, int CiR, int CoR) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int nf = wf ? KHKW * CiR * CoP : 0;
    const int nd = wd ? KHKW * CoR * CiP : 0;
    if (i < nf) {
        const int co = i % CoP, k = i / CoP;
        const int ci = k % CiR, tap = k / CiR;
        wf[i] = (co < Co && ci < Ci) ? w[((size_t)co * Ci + ci) * KHKW + tap] : 0.f;
    } else if (i - nf < nd) {
        const int j = i - nf;
        const int ci = j % CiP, k = j / CiP;
        const int co = k % CoR, tap = k / CoR;
        wd[j] = (co < Co && ci < Ci) ? w[((size_t)co * Ci + ci) * KHKW + tap] : 0.f;
    }
}

// All conv weights of a model in ONE launch (engine.TrainStep: 186 pack launches per step otherwise).  desc[d] =
// {src, dst_fwd, dst_dgrad (or -1): float offsets from the two base pointers; Co, Ci, KHKW; first workgroup}.
struct PackDesc {
    long long src, dstf, dstd;
    int Co, Ci, KHKW, blk0;
};

constexpr int kPackTapsMax = 9;      // tiled transpose up to 3x3 filters (37 KB of LDS); larger filters are few and small

static inline int pack_multi_blocks(int Co, int Ci, int KHKW, int dgrad) {
    const int CoP = (Co + 3) & ~3, CiP = (Ci + 3) & ~3, CiR = round_k(Ci), CoR = round_k(Co);
    if (KHKW <= kPackTapsMax) {
        const int cox = dgrad ? (CoP > CoR ? CoP : CoR) : CoP, cix = dgrad ? (CiR > CiP ? CiR : CiP) : CiR;
        return ((cox + 31) / 32) * ((cix + 31) / 32);
    }
    const int n = KHKW * CiR * CoP + (dgrad ? KHKW * CoR * CiP : 0);
    return (n + 255) / 256;
}

// A workgroup transposes a 32 co x 32 ci x taps tile through LDS: the source [co][ci][tap] is read in contiguous
// runs of 32*taps floats per output channel, both operand layouts are written in 128-byte runs (the element-wise
// version gathered 4 bytes per lane at a stride of Ci*taps floats: 1.2 TB/s over the 390 MB it moves per step).
__global__ void __launch_bounds__(256) pack_weight_multi_kernel(const float* __restrict__ src_base,
                                                                float* __restrict__ dst_base,
                                                                const PackDesc* __restrict__ desc, int ndesc) {
    __shared__ float tile[32][32 * kPackTapsMax + 1];
    int lo = 0, hi = ndesc - 1;                    // last descriptor whose first workgroup is <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (desc[mid].blk0 <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const PackDesc d = desc[lo];
    const int CoP = (d.Co + 3) & ~3, CiP = (d.Ci + 3) & ~3;
    const int CiR = (d.Ci >= 8 && d.Ci % 16 != 0) ? ((d.Ci + 15) & ~15) : d.Ci;
    const int CoR = (d.Co >= 8 && d.Co % 16 != 0) ? ((d.Co + 15) & ~15) : d.Co;
    const float* __restrict__ w = src_base + d.src;
    const int KK = d.KHKW;
    const int blk = (int)blockIdx.x - d.blk0;
    if (KK > kPackTapsMax) {                       // element-wise path (stems, gate convs)
        const int i = blk * 256 + threadIdx.x;
        const int nf = KK * CiR * CoP;
        const int nd = d.dstd >= 0 ? KK * CoR * CiP : 0;
        if (i < nf) {
            const int co = i % CoP, k = i / CoP;
            const int ci = k % CiR, tap = k / CiR;
            dst_base[d.dstf + i] = (co < d.Co && ci < d.Ci) ? w[((size_t)co * d.Ci + ci) * KK + tap] : 0.f;
        } else if (i - nf < nd) {
            const int j = i - nf;
            const int ci = j % CiP, k = j / CiP;
            const int co = k % CoR, tap = k / CoR;
            dst_base[d.dstd + j] = (co < d.Co && ci < d.Ci) ? w[((size_t)co * d.Ci + ci) * KK + tap] : 0.f;
        }
        return;
    }
    const bool dg = d.dstd >= 0;
    const int cix = dg ? (CiR > CiP ? CiR : CiP) : CiR;
    const int cit = (cix + 31) / 32;
    const int co0 = (blk / cit) * 32, ci0 = (blk % cit) * 32;
    const int run = 32 * KK;                       // floats of one output channel's [ci0, ci0+32) x taps
    {
        const int co = threadIdx.x >> 3, l8 = threadIdx.x & 7;
        const bool cok = co0 + co < d.Co;
        const float* row = w + ((size_t)(co0 + co) * d.Ci + ci0) * KK;
        const int valid = (d.Ci - ci0 < 32 ? (d.Ci - ci0 > 0 ? d.Ci - ci0 : 0) : 32) * KK;    // floats inside the tensor
        for (int e = l8; e < run; e += 8) tile[co][e] = (cok && e < valid) ? row[e] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 31, grp = threadIdx.x >> 5;      // 8 rows of 32 per pass
    // forward operand: wf[(tap*CiR + ci)*CoP + co], 32 consecutive co per row
    if (co0 + lane < CoP) {
        for (int tap = 0; tap < KK; ++tap)
            for (int r = grp; r < 32; r += 8) {
                const int ci = ci0 + r;
                if (ci < CiR) dst_base[d.dstf + ((size_t)tap * CiR + ci) * CoP + co0 + lane] = tile[lane][r * KK + tap];
            }
    }
    // input-gradient operand: wd[(tap*CoR + co)*CiP + ci], 32 consecutive ci per row
    if (dg && ci0 + lane < CiP) {
        for (int tap = 0; tap < KK; ++tap)
            for (int r = grp; r < 32; r += 8) {
                const int co = co0 + r;
                if (co < CoR) dst_base[d.dstd + ((size_t)tap * CoR + co) * CiP + ci0 + lane] = tile[r][lane * KK + tap];
            }
    }
}

static bool geom_ok(const dynmm_conv_geom* g) {
    if (!g) return false;
    if (g->N <= 0 || g->Ci <= 0 || g->Co <= 0 || g->H <= 0 || g->W <= 0) return false;
    if (g->KH <= 0 || g->KW <= 0 || g->SH <= 0 || g->SW <= 0 || g->PH < 0 || g->PW < 0) return false;
    if (g->KH > 127 || g->KW > 127 || g->Ci > 65535) return false;
    if (g->Ho != (g->H + 2 * g->PH - g->KH) / g->SH + 1) return false;
    if (g->Wo != (g->W + 2 * g->PW - g->KW) / g->SW + 1) return false;
    if (g->c_split <= 0 || g->c_split > g->Ci) return false;
    if ((double)g->N * g->Ci * g->H * g->W >= 1073741824.0) return false;   // 32-bit byte offsets
    if ((double)g->N * g->Co * g->Ho * g->Wo >= 1073741824.0) return false;
    return true;
}

}  // namespace dynmm

using namespace dynmm;

extern "C" size_t dynmm_packed_weight_floats(int Co, int Ci, int KH, int KW, int dgrad) {
    if (Co <= 0 || Ci <= 0 || KH <= 0 || KW <= 0) return 0;
    return dgrad ? (size_t)KH * KW * round_k(Co) * ((Ci + 3) & ~3) : (size_t)KH * KW * round_k(Ci) * ((Co + 3) & ~3);
}

extern "C" int dynmm_pack_weight(const float* w, float* wp_fwd, float* wp_dgrad, int Co, int Ci,
                                 int KH, int KW, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!w || Co <= 0 || Ci <= 0 || KH <= 0 || KW <= 0) return DYNMM_EINVAL;
    const int CoP = (Co + 3) & ~3, CiP = (Ci + 3) & ~3, CiR = round_k(Ci), CoR = round_k(Co);
    const int total = (wp_fwd ? KH * KW * CiR * CoP : 0) + (wp_dgrad ? KH * KW * CoR * CiP : 0);
    if (total == 0) return DYNMM_OK;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(ceil_div(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, w, wp_fwd, wp_dgrad, Co, Ci, KH * KW, CoP, CiP, CiR, CoR);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_pack_weight_multi_blocks(int Co, int Ci, int KH, int KW, int dgrad) {
    if (Co <= 0 || Ci <= 0 || KH <= 0 || KW <= 0) return 0;
    return pack_multi_blocks(Co, Ci, KH * KW, dgrad);
}

extern "C" int dynmm_pack_weight_multi(const float* src_base, float* dst_base, const void* desc, int ndesc,
                                       int total_blocks, void* stream) {
    (void)hipGetLastError();
    if (!src_base || !dst_base || !desc || ndesc <= 0 || total_blocks <= 0) return DYNMM_EINVAL;
    static_assert(sizeof(PackDesc) == 40, "descriptor layout is part of the ABI (5 x int64 words)");
    hipLaunchKernelGGL(pack_weight_multi_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, src_base,
                       dst_base, (const PackDesc*)desc, ndesc);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_conv2d_fwd(const float* x, const float* x2, const float* wp_fwd,
                                const float* scale, const float* shift, const float* residual,
                                float* y, const dynmm_conv_geom* g, int act, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!x || !wp_fwd || !y || !geom_ok(g)) return DYNMM_EINVAL;
    if ((g->c_split < g->Ci) != (x2 != nullptr)) return DYNMM_EINVAL;
    IgemmArgs a{};
    a.x = x; a.x2 = x2; a.wp = wp_fwd; a.scale = scale; a.shift = shift; a.residual = residual;
    a.mask = nullptr; a.y = y; a.y2 = nullptr;
    a.N = g->N; a.Ci = g->Ci; a.H = g->H; a.W = g->W;
    a.Co = g->Co; a.Ho = g->Ho; a.Wo = g->Wo;
    a.KH = g->KH; a.KW = g->KW; a.SH = g->SH; a.SW = g->SW; a.PH = g->PH; a.PW = g->PW;
    a.c_in_split = g->c_split; a.c_out_split = g->Co; a.act = act;
    {
        SmallConvArgs s{x, x2, wp_fwd, scale, shift, y, g->N, g->Ci, g->H, g->W, g->Co, g->Ho, g->Wo, g->KH, g->KW,
                        g->SH, g->SW, g->PH, g->PW, g->c_split, round_k(g->Ci), (g->Co + 3) & ~3, act};
        if (small_conv_fwd_eligible(s, residual)) return launch_small_conv_fwd(s, (hipStream_t)stream);
        if (stem_conv_fwd_eligible(s, residual)) return launch_stem_conv_fwd(s, (hipStream_t)stream);
    }
    return launch_igemm<false>(a, (hipStream_t)stream);
}

// The 7x7 / stride-2 stem convolution (resnet.py:216-217) with the batch statistics of the BatchNorm that follows it
// (resnet.py:229) from the kernel's own epilogue: conv_small.hip, conv_stem_fwd_kernel<CI, STATS>.
extern "C" int dynmm_conv2d_stem_fwd_stats_supported(const dynmm_conv_geom* g) {
    if (!geom_ok(g) || g->c_split < g->Ci) return 0;
    SmallConvArgs s{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, g->N, g->Ci, g->H, g->W, g->Co, g->Ho, g->Wo, g->KH, g->KW,
                    g->SH, g->SW, g->PH, g->PW, g->c_split, round_k(g->Ci), (g->Co + 3) & ~3, DYNMM_ACT_NONE};
    return stem_conv_fwd_eligible(s, nullptr) ? 1 : 0;
}

extern "C" int dynmm_conv2d_stem_fwd_stats(const float* x, const float* wp_fwd, const float* bias, float* y, double* stats,
                                           const dynmm_conv_geom* g, void* stream) {
    (void)hipGetLastError();
    if (!x || !wp_fwd || !y || !stats || !geom_ok(g)) return DYNMM_EINVAL;
    if (!dynmm_conv2d_stem_fwd_stats_supported(g)) return DYNMM_EUNSUPPORTED;
    SmallConvArgs s{x, nullptr, wp_fwd, nullptr, bias, y, g->N, g->Ci, g->H, g->W, g->Co, g->Ho, g->Wo, g->KH, g->KW,
                    g->SH, g->SW, g->PH, g->PW, g->c_split, round_k(g->Ci), (g->Co + 3) & ~3, DYNMM_ACT_NONE};
    s.stats = stats;
    return launch_stem_conv_fwd(s, (hipStream_t)stream);
}

extern "C" int dynmm_conv2d_dgrad(const float* dy, const float* wp_dgrad, const float* mask,
                                  const float* accum, float* dx, float* dx2,
                                  const dynmm_conv_geom* g, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!dy || !wp_dgrad || !dx || !geom_ok(g)) return DYNMM_EINVAL;
    if ((g->c_split < g->Ci) != (dx2 != nullptr)) return DYNMM_EINVAL;
    if ((mask || accum) && dx2) return DYNMM_EUNSUPPORTED;
    IgemmArgs a{};
    a.x = dy; a.x2 = nullptr; a.wp = wp_dgrad; a.mask = mask; a.residual = accum; a.y = dx; a.y2 = dx2;
    // the GEMM's input is dy [N,Co,Ho,Wo], its output dx [N,Ci,H,W]
    a.N = g->N; a.Ci = g->Co; a.H = g->Ho; a.W = g->Wo;
    a.Co = g->Ci; a.Ho = g->H; a.Wo = g->W;
    a.KH = g->KH; a.KW = g->KW; a.SH = g->SH; a.SW = g->SW; a.PH = g->PH; a.PW = g->PW;
    a.c_in_split = g->Co; a.c_out_split = g->c_split; a.act = DYNMM_ACT_NONE;
    return launch_igemm<true>(a, (hipStream_t)stream);
}

// conv_wgrad_kernel<..., FAST>: the loader treats a thread's x rows in groups of eight (64 consecutive k) that share one filter
// tap and one validity bit.  True when taps change at multiples of 64 (Ci % 64 == 0) — and for EVERY Ci when there is a single
// tap (1x1): rows k >= K of a partly filled group then read a mapped address and feed accumulator columns that are never stored
// (a GEMM column depends on its own operand row only).  Round 6: the Linear layers of ModalityDynMM (Ci = 120 / 60 / 10) ran the
// per-row loader at a third of the FAST rate.
static bool wgrad_fast_rows_ok(const dynmm_conv_geom* g) {
    return (g->Ci % 64 == 0) || (g->KH == 1 && g->KW == 1);
}

static void launch_wgrad_generic(const WgradArgs& a, const WgradGroup& grp, const WgradPlan& p, dim3 grid, bool dual,
                                 bool fast, hipStream_t st) {
#define DYNMM_WGRAD_LAUNCH(TCO, TK, WCO, WK)                                                          \
    do {                                                                                              \
        if (dual)                                                                                     \
            hipLaunchKernelGGL((conv_wgrad_kernel<TCO, TK, WCO, WK, true, false>), grid, dim3(256), 0, st, a, grp);  \
        else if (fast)                                                                                \
            hipLaunchKernelGGL((conv_wgrad_kernel<TCO, TK, WCO, WK, false, true>), grid, dim3(256), 0, st, a, grp);  \
        else                                                                                          \
            hipLaunchKernelGGL((conv_wgrad_kernel<TCO, TK, WCO, WK, false, false>), grid, dim3(256), 0, st, a, grp); \
    } while (0)
    if (p.tco == 128)
        DYNMM_WGRAD_LAUNCH(128, 128, 64, 64);
    else if (p.tco == 64 && p.tk == 192)
        DYNMM_WGRAD_LAUNCH(64, 192, 32, 96);
    else if (p.tco == 64 && p.tk == 64)
        DYNMM_WGRAD_LAUNCH(64, 64, 32, 32);
    else if (p.tco == 64)
        DYNMM_WGRAD_LAUNCH(64, 128, 64, 32);
    else
        DYNMM_WGRAD_LAUNCH(32, 128, 32, 32);
#undef DYNMM_WGRAD_LAUNCH
}

static size_t generic_wgrad_workspace_bytes(const dynmm_conv_geom* g);

extern "C" size_t dynmm_conv2d_wgrad_workspace_bytes(const dynmm_conv_geom* g) {
    if (!geom_ok(g)) return 0;
    // (the stem kernel is chosen at call time only when no bias gradient is asked for: size for either path)
    const size_t stem = stem_conv_wgrad_eligible(g->Ci, g->Co, g->KH, g->KW, g->SH, g->SW, g->PH, g->PW, g->c_split < g->Ci, false)
                            ? stem_conv_wgrad_workspace_bytes(g->N, g->Ci, g->Ho, g->Wo) : 0;
    const size_t gen = generic_wgrad_workspace_bytes(g);
    const size_t co8 = co8_wgrad_eligible(g->Ci, g->Co, g->H, g->W, g->Ho, g->Wo, g->KH, g->KW, g->SH, g->SW, g->PH, g->PW, g->c_split)
                           ? co8_wgrad_workspace_bytes(g->N, g->Ci, g->Co) : 0;
    const size_t m = stem > gen ? stem : gen;
    return co8 > m ? co8 : m;
}

static size_t plan_workspace_bytes(const dynmm_conv_geom* g, const WgradPlan& p) {
    if (p.splits <= 1) return 0;
    // [splits][Co*K] weight-gradient slabs, then [splits][Co] bias-gradient slabs (16-byte aligned start)
    const size_t wslab = ((size_t)p.splits * g->Co * g->Ci * g->KH * g->KW + 3) & ~(size_t)3;
    return (wslab + (size_t)p.splits * g->Co) * sizeof(float);
}

// (the three-tap kernel needs 16-byte aligned tensors, known only at the call: room for either plan)
static size_t generic_wgrad_workspace_bytes(const dynmm_conv_geom* g) {
    const size_t a = plan_workspace_bytes(g, plan_wgrad(g)), b = plan_workspace_bytes(g, plan_wgrad(g, false));
    return a > b ? a : b;
}

extern "C" int dynmm_conv2d_wgrad(const float* x, const float* x2, const float* dy, float* dw, float* dbias,
                                  void* workspace, size_t workspace_bytes,
                                  const dynmm_conv_geom* g, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!x || !dy || !dw || !geom_ok(g)) return DYNMM_EINVAL;
    if ((g->c_split < g->Ci) != (x2 != nullptr)) return DYNMM_EINVAL;
    if (stem_conv_wgrad_eligible(g->Ci, g->Co, g->KH, g->KW, g->SH, g->SW, g->PH, g->PW, x2 != nullptr, dbias != nullptr) &&
        ((reinterpret_cast<uintptr_t>(dy) & 15u) == 0)) {
        const size_t need_s = stem_conv_wgrad_workspace_bytes(g->N, g->Ci, g->Ho, g->Wo);
        if (!workspace || workspace_bytes < need_s) return DYNMM_EWORKSPACE;
        return launch_stem_conv_wgrad(x, dy, dw, (float*)workspace, g->N, g->Ci, g->H, g->W, g->Ho, g->Wo, (hipStream_t)stream);
    }
    const bool aligned16 = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(dy)) & 15u) == 0;
    if (co8_wgrad_eligible(g->Ci, g->Co, g->H, g->W, g->Ho, g->Wo, g->KH, g->KW, g->SH, g->SW, g->PH, g->PW, g->c_split) &&
        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(x2)) & 15u) == 0) {
        // the gate head's first convolution: direct vector-ALU kernel (conv_small.hip), weight and bias gradient in one launch
        const size_t need_c = co8_wgrad_workspace_bytes(g->N, g->Ci, g->Co);
        if (!workspace || workspace_bytes < need_c || (reinterpret_cast<uintptr_t>(workspace) & 15u)) return DYNMM_EWORKSPACE;
        return launch_co8_wgrad(x, x2, dy, dw, dbias, (float*)workspace, g->N, g->Ci, g->H, g->W, g->Co, g->Ho, g->Wo,
                                g->c_split, (hipStream_t)stream);
    }
    const WgradPlan p = plan_wgrad(g, aligned16 && !x2);
    const size_t need = generic_wgrad_workspace_bytes(g);
    if (need > 0 && (!workspace || workspace_bytes < need)) return DYNMM_EWORKSPACE;
    WgradArgs a{};
    a.x = x; a.x2 = x2; a.dy = dy;
    a.out = p.splits > 1 ? (float*)workspace : dw;
    const size_t wslab = ((size_t)p.splits * g->Co * g->Ci * g->KH * g->KW + 3) & ~(size_t)3;
    float* bias_slabs = p.splits > 1 ? (float*)workspace + wslab : dbias;
    a.out_bias = dbias ? bias_slabs : nullptr;
    a.N = g->N; a.Ci = g->Ci; a.H = g->H; a.W = g->W; a.Co = g->Co; a.Ho = g->Ho; a.Wo = g->Wo;
    a.KH = g->KH; a.KW = g->KW; a.SH = g->SH; a.SW = g->SW; a.PH = g->PH; a.PW = g->PW;
    a.c_split = g->c_split;
    a.M = g->N * g->Ho * g->Wo;
    a.K = g->KH * g->KW * g->Ci;
    a.n_co_tiles = p.n_co_tiles; a.n_k_tiles = p.n_k_tiles; a.steps_per_split = p.steps_per_split;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)(p.n_co_tiles * p.n_k_tiles * p.splits));
    const bool dual = x2 != nullptr;
    const bool fast = !dual && wgrad_fast_rows_ok(g) && g->H >= g->KH && g->W >= g->KW;
    a.magic_wo = (g->Wo >= 2 && (unsigned long long)g->Ho * g->Wo * g->Wo < (1ull << 32)) ? (unsigned)((1ull << 32) / (unsigned)g->Wo) + 1u : 0u;
    const bool v4 = !p.v6 && !dual && p.tco == 128 && p.tk == 128 && (g->Ci % 64 == 0) && g->SW == 1 &&
                    (g->KW == 1 || g->KW == 3) && g->PW == g->KW / 2 && g->W == g->Wo && (g->W % 4 == 0) &&
                    ((g->Ho * g->Wo) % 4 == 0) && g->H >= g->KH &&
                    ((reinterpret_cast<uintptr_t>(x) & 15u) == 0) && ((reinterpret_cast<uintptr_t>(dy) & 15u) == 0);
    const bool bias_ok4 = !dbias || ((g->Co % 4 == 0) && ((reinterpret_cast<uintptr_t>(bias_slabs) & 15u) == 0) &&
                                     ((reinterpret_cast<uintptr_t>(dbias) & 15u) == 0));
    const bool perm = (v4 || p.v6) && p.splits > 1 && ((reinterpret_cast<uintptr_t>(workspace) & 15u) == 0) && bias_ok4;
    a.k_major_out = perm ? 1 : 0;
    if (p.v6 || v4) {
        if (p.v6)
            launch_wgrad_v6(a, WgradGroup{}, grid, wgrad_v6_occupancy(g), st);
        else
            hipLaunchKernelGGL((conv_wgrad_v4_kernel<128, 128, 1>), grid, dim3(512), 0, st, a, WgradGroup{});
        DYNMM_LAUNCH_CHECK();
        if (perm) {
            const int n = g->Co * a.K, nb1 = ceil_div(n / 4, 64), nb2 = dbias ? ceil_div(g->Co / 4, 64) : 0;
            hipLaunchKernelGGL(reduce_slabs_perm_kernel, dim3(nb1 + nb2), dim3(256), 0, st, (const float*)workspace, dw,
                               n, p.splits, g->Ci, g->KH * g->KW, a.K, dbias ? bias_slabs : nullptr, dbias,
                               dbias ? g->Co : 0, nb1, ReduceGroup{});
            DYNMM_LAUNCH_CHECK();
            return DYNMM_OK;
        }
    } else
        launch_wgrad_generic(a, WgradGroup{}, p, grid, dual, fast, st);
    DYNMM_LAUNCH_CHECK();
    if (p.splits > 1) {
        launch_reduce_slabs((const float*)workspace, dw, g->Co * a.K, p.splits, st, dbias ? bias_slabs : nullptr,
                            dbias, dbias ? g->Co : 0);
        DYNMM_LAUNCH_CHECK();
    }
    return DYNMM_OK;
}

// ---- grouped weight gradients --------------------------------------------------------------------------------------
static bool wgrad_v4_shape_ok(const dynmm_conv_geom* g, const WgradPlan& p) {
    if (p.v6) return (g->Co % 4 == 0);
    return g->c_split == g->Ci && p.tco == 128 && p.tk == 128 && (g->Ci % 64 == 0) && g->SW == 1 &&
           (g->KW == 1 || g->KW == 3) && g->PW == g->KW / 2 && g->W == g->Wo && (g->W % 4 == 0) &&
           ((g->Ho * g->Wo) % 4 == 0) && g->H >= g->KH && (g->Co % 4 == 0);
}

// the plan of n same-shape problems sharing one residency round
static WgradPlan plan_wgrad_group(const dynmm_conv_geom* g, int n) {
    WgradPlan p = plan_wgrad(g);
    if (!p.v6) p.target = 512;
    plan_splits(p, g, n);
    return p;
}

static size_t group_problem_floats(const dynmm_conv_geom* g, const WgradPlan& p) {
    const size_t wslab = ((size_t)p.splits * g->Co * g->Ci * g->KH * g->KW + 3) & ~(size_t)3;
    return (wslab + (size_t)p.splits * g->Co + 3) & ~(size_t)3;
}

// 2: the vectorised 128x128 kernel; 1: the generic tiles (single input, split pixel range); 0: not groupable
extern "C" int dynmm_conv2d_wgrad_groupable(const dynmm_conv_geom* g) {
    if (!geom_ok(g) || g->c_split != g->Ci) return 0;
    if (stem_conv_wgrad_eligible(g->Ci, g->Co, g->KH, g->KW, g->SH, g->SW, g->PH, g->PW, false, false)) return 0;
    if (co8_wgrad_eligible(g->Ci, g->Co, g->H, g->W, g->Ho, g->Wo, g->KH, g->KW, g->SH, g->SW, g->PH, g->PW, g->c_split)) return 0;
    const WgradPlan p = plan_wgrad(g);
    if (p.splits <= 1) return 0;
    return wgrad_v4_shape_ok(g, p) ? 2 : 1;
}

extern "C" int dynmm_conv2d_wgrad_variant(const dynmm_conv_geom* g) {
    if (!geom_ok(g)) return 0;
    if (stem_conv_wgrad_eligible(g->Ci, g->Co, g->KH, g->KW, g->SH, g->SW, g->PH, g->PW, g->c_split < g->Ci, false)) return 0;
    if (co8_wgrad_eligible(g->Ci, g->Co, g->H, g->W, g->Ho, g->Wo, g->KH, g->KW, g->SH, g->SW, g->PH, g->PW, g->c_split)) return 8;
    const WgradPlan p = plan_wgrad(g);
    if (p.v6) return 6;
    return wgrad_v4_shape_ok(g, p) ? 4 : 0;
}

extern "C" size_t dynmm_conv2d_wgrad_group_workspace_bytes(const dynmm_conv_geom* g, int n) {
    if (!geom_ok(g) || n < 1 || n > kWgradGroupMax) return 0;
    if (n == 1 || !dynmm_conv2d_wgrad_groupable(g)) return dynmm_conv2d_wgrad_workspace_bytes(g);
    const WgradPlan p = plan_wgrad_group(g, n);
    return sizeof(float) * group_problem_floats(g, p) * (size_t)n;
}

extern "C" int dynmm_conv2d_wgrad_group(int n, const float* const* xs, const float* const* dys, float* const* dws,
                                        float* const* dbiases, void* workspace, size_t workspace_bytes,
                                        const dynmm_conv_geom* g, void* stream) {
    (void)hipGetLastError();
    if (n < 1 || n > kWgradGroupMax || !xs || !dys || !dws || !geom_ok(g)) return DYNMM_EINVAL;
    bool aligned = (reinterpret_cast<uintptr_t>(workspace) & 15u) == 0;
    for (int i = 0; i < n; ++i) {
        if (!xs[i] || !dys[i] || !dws[i]) return DYNMM_EINVAL;
        aligned = aligned && ((reinterpret_cast<uintptr_t>(xs[i]) | reinterpret_cast<uintptr_t>(dys[i])) & 15u) == 0 &&
                  (!dbiases || !dbiases[i] || (reinterpret_cast<uintptr_t>(dbiases[i]) & 15u) == 0);
        if (dbiases && ((dbiases[i] != nullptr) != (dbiases[0] != nullptr))) return DYNMM_EINVAL;      // all or none
    }
    const int kind = dynmm_conv2d_wgrad_groupable(g);
    if (n == 1 || kind == 0 || (kind == 2 && !aligned)) {               // one ordinary launch per problem
        for (int i = 0; i < n; ++i) {
            const int rc = dynmm_conv2d_wgrad(xs[i], nullptr, dys[i], dws[i], dbiases ? dbiases[i] : nullptr, workspace,
                                              workspace_bytes, g, stream);
            if (rc != DYNMM_OK) return rc;
        }
        return DYNMM_OK;
    }
    const WgradPlan p = plan_wgrad_group(g, n);
    const size_t per_floats = group_problem_floats(g, p);
    if (!workspace || workspace_bytes < sizeof(float) * per_floats * (size_t)n) return DYNMM_EWORKSPACE;
    const bool has_bias = dbiases && dbiases[0];
    const size_t wslab = ((size_t)p.splits * g->Co * g->Ci * g->KH * g->KW + 3) & ~(size_t)3;
    WgradArgs a{};
    WgradGroup grp{};
    grp.nprob = n;
    grp.per = p.n_co_tiles * p.n_k_tiles * p.splits;
    for (int i = 0; i < n; ++i) {
        float* base = (float*)workspace + per_floats * (size_t)i;
        grp.x[i] = xs[i];
        grp.dy[i] = dys[i];
        grp.out[i] = base;
        grp.out_bias[i] = has_bias ? base + wslab : nullptr;
    }
    a.x = xs[0]; a.x2 = nullptr; a.dy = dys[0]; a.out = grp.out[0]; a.out_bias = grp.out_bias[0];
    a.N = g->N; a.Ci = g->Ci; a.H = g->H; a.W = g->W; a.Co = g->Co; a.Ho = g->Ho; a.Wo = g->Wo;
    a.KH = g->KH; a.KW = g->KW; a.SH = g->SH; a.SW = g->SW; a.PH = g->PH; a.PW = g->PW;
    a.c_split = g->c_split;
    a.M = g->N * g->Ho * g->Wo;
    a.K = g->KH * g->KW * g->Ci;
    a.n_co_tiles = p.n_co_tiles; a.n_k_tiles = p.n_k_tiles; a.steps_per_split = p.steps_per_split;
    a.magic_wo = (g->Wo >= 2 && (unsigned long long)g->Ho * g->Wo * g->Wo < (1ull << 32)) ? (unsigned)((1ull << 32) / (unsigned)g->Wo) + 1u : 0u;
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)(grp.per * n));
    if (kind == 1) {
        // generic tiles: plain [Co][Ci][KH][KW] slabs, summed by reduce_slabs_kernel (one launch per problem)
        a.k_major_out = 0;
        const bool fast = wgrad_fast_rows_ok(g) && g->H >= g->KH && g->W >= g->KW;
        launch_wgrad_generic(a, grp, p, grid, false, fast, st);
        DYNMM_LAUNCH_CHECK();
        ReduceGroup rgp{};
        rgp.nprob = n;
        for (int i = 0; i < n; ++i) {
            rgp.slabs[i] = grp.out[i];
            rgp.out[i] = dws[i];
            rgp.slabs2[i] = has_bias ? grp.out_bias[i] : nullptr;
            rgp.out2[i] = has_bias ? dbiases[i] : nullptr;
        }
        launch_reduce_slabs_group(rgp, g->Co * a.K, p.splits, has_bias ? g->Co : 0, st);
        DYNMM_LAUNCH_CHECK();
        return DYNMM_OK;
    }
    a.k_major_out = 1;
    if (p.v6)
        launch_wgrad_v6(a, grp, grid, wgrad_v6_occupancy(g), st);
    else
        hipLaunchKernelGGL((conv_wgrad_v4_kernel<128, 128, 1>), grid, dim3(512), 0, st, a, grp);
    DYNMM_LAUNCH_CHECK();
    const int nel = g->Co * a.K, nb1 = ceil_div(nel / 4, 64), nb2 = has_bias ? ceil_div(g->Co / 4, 64) : 0;
    ReduceGroup rg{};
    rg.nprob = n;
    for (int i = 0; i < n; ++i) {
        rg.slabs[i] = grp.out[i];
        rg.out[i] = dws[i];
        rg.slabs2[i] = has_bias ? grp.out_bias[i] : nullptr;
        rg.out2[i] = has_bias ? dbiases[i] : nullptr;
    }
    hipLaunchKernelGGL(reduce_slabs_perm_kernel, dim3(nb1 + nb2, n), dim3(256), 0, st, (const float*)nullptr, (float*)nullptr,
                       nel, p.splits, g->Ci, g->KH * g->KW, a.K, (const float*)nullptr, (float*)nullptr,
                       has_bias ? g->Co : 0, nb1, rg);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_reduce_slabs(const float* slabs, float* out, int n, int nslabs, void* stream) {
    (void)hipGetLastError();   // drop stale errors left by other users of the runtime
    if (!slabs || !out || n <= 0 || nslabs <= 0) return DYNMM_EINVAL;
    launch_reduce_slabs(slabs, out, n, nslabs, (hipStream_t)stream);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

#ifdef DYNMM_TRACE
#include "conv_igemm_v5.hip"      // trace build: one translation unit, so both kernel families see g_trace
#endif
