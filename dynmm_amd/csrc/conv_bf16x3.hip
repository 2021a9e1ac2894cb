// Split-precision implicit-GEMM convolution: fp32 tensors in HBM, bf16 matrix cores, fp32 accumulate.
//
// Every fp32 operand is split on the fly into NS bf16 pieces (v = p0 + p1 [+ p2], each piece the bf16
// rounding of the remaining residual) and each k-slice is accumulated on v_mfma_f32_32x32x16_bf16:
//   NS = 2 ("bf16x3"): p0*q0 + p0*q1 + p1*q0              16 mantissa bits, ~2^-17 per product;
//                      logits within 1.1e-5 of fp32 in eval, but train-mode batch-stat BN at tiny
//                      batches amplifies it to ~2e-3 — outside the 1e-3 bar, so opt-in only;
//   NS = 3 ("bf16x6"): + p0*q2 + p2*q0 + p1*q1            24 mantissa bits = fp32's own significand:
//                      dropped terms are <= 2^-24 relative, i.e. fp32-rounding class.
// 6 bf16 MFMAs cost 6/16 of one fp32-MFMA k-slice, so matrix-core time drops 2.7x (5.3x for NS=2) and
// the kernel becomes bound by operand delivery (L2/LDS, latency) instead of MFMA issue.
//
// Same GEMM orientation and epilogue as conv_igemm.hip: D[co][pix], weights = MFMA A (i = co),
// gathered activations = MFMA B (j = pixel), so lanes own consecutive pixels in every global access.
// Differences that the bf16 fragment shape (8 consecutive k per lane) imposes:
//   * weights are pre-split per step by pack_weight_bf16x3 into bf16 hi/lo matrices [Co][K]
//     (k contiguous): a weight-tile row is fetched as 16-byte chunks and dropped into LDS unchanged;
//   * LDS tiles are [k/8][row][8 bf16]: a fragment is ONE conflict-free ds_read_b128 per lane;
//   * activations are gathered as fp32 (one pixel, 8..16 consecutive channels per lane), split and
//     packed in registers after the MFMA block, and stored with 16-byte LDS writes.
// Eligibility (host side): Ci % 32 == 0, Co >= 33, single input tensor.  Everything else (stems, gate,
// 40-class heads' dgrad) stays on the fp32 kernels.
#include "common.h"

namespace dynmm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16v __attribute__((ext_vector_type(16)));

struct BfArgs {
    const float* x;             // gemm input  [N, Ci, H, W]
    const unsigned short* wsp;  // bf16 weight pieces [NS][Co][K]
    const float* scale;
    const float* shift;
    const float* residual;
    const float* mask;
    float* y;                   // gemm output [N, Co, Ho, Wo]
    int N, Ci, H, W;
    int Co, Ho, Wo;
    int KH, KW, SH, SW, PH, PW;
    int act;
    int M, K;
    int n_co_tiles, n_pix_tiles;
};

__device__ __forceinline__ float ldg_f32_bf(const float* sbase, unsigned voff_bytes) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(sbase) + voff_bytes);
}

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

template <int TCO, int TPIX, int WCO, int WPIX, int BK, int NS, bool DGRAD>
__global__ void __launch_bounds__(256) conv_igemm_bf16x3_kernel(const BfArgs a) {
    constexpr int KC = BK / 8;                          // chunks of 8 k per K-step
    constexpr int MCO = WCO / 32, MPIX = WPIX / 32;
    constexpr int WAVES_PIX = TPIX / WPIX;
    static_assert((TCO / WCO) * WAVES_PIX == 4, "4 waves per workgroup");
    constexpr int A_TOTAL = TCO * KC;                   // 16-byte chunks per array per K-step
    constexpr int A_CHUNKS = (A_TOTAL + 255) / 256;     // ... per thread
    constexpr int B_PER = BK * TPIX / 256;              // fp32 activations per thread per K-step (4, 8 or 16)
    static_assert(A_CHUNKS >= 1 && (B_PER == 4 || B_PER == 8 || B_PER == 16), "unsupported tile");

    // [stage][piece][k chunk][row][8 bf16]
    __shared__ __attribute__((aligned(16))) unsigned short As[2][NS][KC][TCO][8];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][NS][KC][TPIX][8];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wave_co = wave / WAVES_PIX, wave_pix = wave % WAVES_PIX;
    const int khalf = lane >> 5, l31 = lane & 31;

    const int nblk = a.n_co_tiles * a.n_pix_tiles;
    const int lin = xcd_remap(blockIdx.x, nblk);
    const int co0 = (lin % a.n_co_tiles) * TCO;
    const int pix0 = (lin / a.n_co_tiles) * TPIX;
    const int HW = a.H * a.W, HoWo = a.Ho * a.Wo;

    // ---- weight-tile loader: chunk q = t + 256*i -> (row = q / KC, c = q % KC) ----
    unsigned a_voff[A_CHUNKS];
#pragma unroll
    for (int i = 0; i < A_CHUNKS; ++i) {
        const int q = (t + 256 * i) % A_TOTAL;          // (threads past the tile re-read a valid chunk)
        int co = co0 + q / KC;
        if (co > a.Co - 1) co = a.Co - 1;               // rows past Co: finite junk, discarded below
        a_voff[i] = ((unsigned)co * (unsigned)a.K + (unsigned)(q % KC) * 8u) * 2u;
    }

    // ---- activation loader: one pixel, B_PER consecutive channels per lane ----
    const int b_pix = t % TPIX, b_kg = t / TPIX;
    const int m_b = pix0 + b_pix;
    const bool m_ok = m_b < a.M;
    int n_b = 0, oh_b = 0, ow_b = 0;
    if (m_ok) {
        n_b = m_b / HoWo;
        const int rem = m_b - n_b * HoWo;
        oh_b = rem / a.Wo;
        ow_b = rem - oh_b * a.Wo;
    }
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

    f32x16v acc[MCO][MPIX];
#pragma unroll
    for (int mi = 0; mi < MCO; ++mi)
#pragma unroll
        for (int ni = 0; ni < MPIX; ++ni)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[mi][ni][j] = 0.f;

    const int cpt = a.Ci / BK;
    const int nk = a.KH * a.KW * cpt;
    int ci0 = 0, tr = 0, ts = 0, kbase = 0;
    bool tap_ok;
    unsigned voff;
    auto set_tap = [&]() {
        int ih = 0, iw = 0;
        tap_ok = in_coord(tr, ts, ih, iw);
        const unsigned pix = tap_ok ? (unsigned)(ih * a.W + iw) : 0u;
        voff = ((unsigned)(n_b * a.Ci + b_kg * B_PER) * (unsigned)HW + pix) * 4u;
    };

    uint4 raw[NS][A_CHUNKS];
    float rb[B_PER];
    bool ld_ok = false;
    const size_t plane = (size_t)a.Co * a.K * 2;         // bytes between weight pieces
    auto load_tile = [&]() {
        const char* w0 = reinterpret_cast<const char*>(a.wsp) + (size_t)kbase * 2;
#pragma unroll
        for (int p = 0; p < NS; ++p)
#pragma unroll
            for (int i = 0; i < A_CHUNKS; ++i)
                raw[p][i] = *reinterpret_cast<const uint4*>(w0 + p * plane + a_voff[i]);
        const float* xbase = a.x + (size_t)ci0 * HW;
#pragma unroll
        for (int i = 0; i < B_PER; ++i) rb[i] = ldg_f32_bf(xbase + (size_t)i * HW, voff);
        ld_ok = tap_ok;
        ci0 += BK;
        kbase += BK;
        if (ci0 >= a.Ci) {
            ci0 = 0;
            if (++ts == a.KW) { ts = 0; ++tr; }
            set_tap();
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_CHUNKS; ++i) {
            const int q = t + 256 * i;
            if (A_TOTAL % 256 == 0 || q < A_TOTAL) {
#pragma unroll
                for (int p = 0; p < NS; ++p) *reinterpret_cast<uint4*>(&As[buf][p][q % KC][q / KC][0]) = raw[p][i];
            }
        }
        if constexpr (B_PER >= 8) {
#pragma unroll
            for (int c = 0; c < B_PER / 8; ++c) {
                bf16x8 pc[NS];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float r = ld_ok ? rb[c * 8 + j] : 0.f;
#pragma unroll
                    for (int p = 0; p < NS; ++p) {
                        const __bf16 h = (__bf16)r;
                        pc[p][j] = h;
                        r -= (float)h;                   // exact in fp32
                    }
                }
#pragma unroll
                for (int p = 0; p < NS; ++p)
                    *reinterpret_cast<bf16x8*>(&Bs[buf][p][b_kg * (B_PER / 8) + c][b_pix][0]) = pc[p];
            }
        } else {      // 4 channels per lane: half of an 8-k chunk, 8-byte LDS stores
            bf16x4 pc[NS];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float r = ld_ok ? rb[j] : 0.f;
#pragma unroll
                for (int p = 0; p < NS; ++p) {
                    const __bf16 h = (__bf16)r;
                    pc[p][j] = h;
                    r -= (float)h;
                }
            }
#pragma unroll
            for (int p = 0; p < NS; ++p)
                *reinterpret_cast<bf16x4*>(&Bs[buf][p][b_kg / 2][b_pix][(b_kg & 1) * 4]) = pc[p];
        }
    };
    auto mfma_tile = [&](int buf) {
#pragma unroll
        for (int kq = 0; kq < BK / 16; ++kq) {
            const int c = 2 * kq + khalf;                 // this lane's 8-k chunk of the K16 slice
            bf16x8 af[NS][MCO], bfr[NS][MPIX];
#pragma unroll
            for (int p = 0; p < NS; ++p) {
#pragma unroll
                for (int mi = 0; mi < MCO; ++mi)
                    af[p][mi] = *reinterpret_cast<const bf16x8*>(&As[buf][p][c][wave_co * WCO + mi * 32 + l31][0]);
#pragma unroll
                for (int ni = 0; ni < MPIX; ++ni)
                    bfr[p][ni] = *reinterpret_cast<const bf16x8*>(&Bs[buf][p][c][wave_pix * WPIX + ni * 32 + l31][0]);
            }
#pragma unroll
            for (int mi = 0; mi < MCO; ++mi)
#pragma unroll
                for (int ni = 0; ni < MPIX; ++ni) {
                    // every piece pair (p, q) with p + q < NS, smallest magnitude first, p0*q0 last
#pragma unroll
                    for (int sum = NS - 1; sum >= 0; --sum)
#pragma unroll
                        for (int p = 0; p <= sum; ++p)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[p][mi], bfr[sum - p][ni],
                                                                                  acc[mi][ni], 0, 0, 0);
                }
        }
    };

    set_tap();
    load_tile();
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile();
        mfma_tile(buf);
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue (identical to the fp32 kernel) ----
#pragma unroll
    for (int ni = 0; ni < MPIX; ++ni) {
        const int m = pix0 + wave_pix * WPIX + ni * 32 + l31;
        if (m >= a.M) continue;
        const int n = m / HoWo;
        const int rem = m - n * HoWo;
#pragma unroll
        for (int mi = 0; mi < MCO; ++mi) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int co = co0 + wave_co * WCO + mi * 32 + (j & 3) + 8 * (j >> 2) + 4 * khalf;
                if (co >= a.Co) continue;
                float v = acc[mi][ni][j];
                if (a.scale) v *= a.scale[co];
                if (a.shift) v += a.shift[co];
                const size_t idx = ((size_t)n * a.Co + co) * HoWo + rem;
                if (DGRAD) {
                    if (a.mask) v = a.mask[idx] > 0.f ? v : 0.f;
                    if (a.residual) v += a.residual[idx];
                } else {
                    if (a.residual) v += a.residual[idx];
                    v = act_fwd(v, a.act);
                }
                a.y[idx] = v;
            }
        }
    }
}

// w[Co][Ci][KH][KW] fp32 -> bf16 hi/lo, k-contiguous:
//   fwd  : fh/fl[co][tap*Ci + ci]        dgrad: dh/dl[ci][tap*Co + co]
__global__ void __launch_bounds__(256) pack_weight_bf16x3_kernel(const float* __restrict__ w,
                                                                 unsigned short* __restrict__ fw,
                                                                 unsigned short* __restrict__ dg,
                                                                 int Co, int Ci, int KHKW, int ns) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)Co * Ci * KHKW;
    if (i >= (int)total) return;
    const int tap = i % KHKW;
    const int ci = (i / KHKW) % Ci;
    const int co = i / (KHKW * Ci);
    float r = w[i];
    const size_t K = (size_t)KHKW * Ci, Kd = (size_t)KHKW * Co;
    for (int p = 0; p < ns; ++p) {
        const __bf16 h = (__bf16)r;
        r -= (float)h;
        const unsigned short hb = __builtin_bit_cast(unsigned short, h);
        if (fw) fw[p * total + (size_t)co * K + (size_t)tap * Ci + ci] = hb;
        if (dg) dg[p * total + (size_t)ci * Kd + (size_t)tap * Co + co] = hb;
    }
}

static bool bf_geom_ok(const dynmm_conv_geom* g) {
    if (!g) return false;
    if (g->N <= 0 || g->Ci <= 0 || g->Co <= 0 || g->H <= 0 || g->W <= 0) return false;
    if (g->KH <= 0 || g->KW <= 0 || g->SH <= 0 || g->SW <= 0 || g->PH < 0 || g->PW < 0) return false;
    if (g->Ho != (g->H + 2 * g->PH - g->KH) / g->SH + 1) return false;
    if (g->Wo != (g->W + 2 * g->PW - g->KW) / g->SW + 1) return false;
    if (g->c_split != g->Ci) return false;
    if ((double)g->N * g->Ci * g->H * g->W >= 1073741824.0) return false;
    if ((double)g->N * g->Co * g->Ho * g->Wo >= 1073741824.0) return false;
    return true;
}

template <bool DGRAD>
static int launch_bf16x3(BfArgs& a, int ns, hipStream_t st) {
    a.M = a.N * a.Ho * a.Wo;
    a.K = a.KH * a.KW * a.Ci;
#define DYNMM_BF_LAUNCH(TCO, TPIX, WCO, WPIX, BK)                                                   \
    do {                                                                                            \
        a.n_co_tiles = ceil_div(a.Co, TCO);                                                         \
        a.n_pix_tiles = ceil_div(a.M, TPIX);                                                        \
        if (ns == 3)                                                                                \
            hipLaunchKernelGGL((conv_igemm_bf16x3_kernel<TCO, TPIX, WCO, WPIX, BK, 3, DGRAD>),      \
                               dim3((unsigned)(a.n_co_tiles * a.n_pix_tiles)), dim3(256), 0, st, a);\
        else                                                                                        \
            hipLaunchKernelGGL((conv_igemm_bf16x3_kernel<TCO, TPIX, WCO, WPIX, BK, 2, DGRAD>),      \
                               dim3((unsigned)(a.n_co_tiles * a.n_pix_tiles)), dim3(256), 0, st, a);\
    } while (0)
    static const char* tile_env = getenv("DYNMM_BF16_TPIX");
    static const char* bk_env = getenv("DYNMM_BF16_BK");
    const int force = tile_env ? atoi(tile_env) : 0;
    const int bk = bk_env ? atoi(bk_env) : 16;    // measured: BK=16 (32 KB LDS, 3 waves/SIMD) 95-101 us vs BK=32 144-147 us at C=128/256
    if (a.Co > 64) {
        if (force == 64 && bk == 16) DYNMM_BF_LAUNCH(128, 64, 64, 32, 16);
        else if (force == 64) DYNMM_BF_LAUNCH(128, 64, 64, 32, 32);
        else if (bk == 16) DYNMM_BF_LAUNCH(128, 128, 64, 64, 16);
        else DYNMM_BF_LAUNCH(128, 128, 64, 64, 32);
    } else {
        if (bk == 16) DYNMM_BF_LAUNCH(64, 128, 32, 64, 16);
        else DYNMM_BF_LAUNCH(64, 128, 32, 64, 32);
    }
#undef DYNMM_BF_LAUNCH
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

}  // namespace dynmm

using namespace dynmm;

extern "C" int dynmm_conv_bf16x3_eligible(const dynmm_conv_geom* g, int dgrad) {
    if (!bf_geom_ok(g)) return 0;
    const int gemm_ci = dgrad ? g->Co : g->Ci;
    const int gemm_co = dgrad ? g->Ci : g->Co;
    return (gemm_ci % 32 == 0 && gemm_co > 32) ? 1 : 0;
}

extern "C" int dynmm_pack_weight_bf16(const float* w, unsigned short* fwd, unsigned short* dgrad, int Co, int Ci,
                                     int KH, int KW, int nsplit, void* stream) {
    (void)hipGetLastError();
    if (!w || Co <= 0 || Ci <= 0 || KH <= 0 || KW <= 0 || (nsplit != 2 && nsplit != 3) || (!fwd && !dgrad))
        return DYNMM_EINVAL;
    const int total = Co * Ci * KH * KW;
    hipLaunchKernelGGL(pack_weight_bf16x3_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream,
                       w, fwd, dgrad, Co, Ci, KH * KW, nsplit);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_conv2d_fwd_bf16(const float* x, const unsigned short* w_split, int nsplit, const float* scale,
                                     const float* shift, const float* residual, float* y,
                                     const dynmm_conv_geom* g, int act, void* stream) {
    (void)hipGetLastError();
    if (!x || !w_split || !y || !bf_geom_ok(g) || (nsplit != 2 && nsplit != 3)) return DYNMM_EINVAL;
    if (!dynmm_conv_bf16x3_eligible(g, 0)) return DYNMM_EUNSUPPORTED;
    BfArgs a{};
    a.x = x; a.wsp = w_split; a.scale = scale; a.shift = shift; a.residual = residual; a.mask = nullptr; a.y = y;
    a.N = g->N; a.Ci = g->Ci; a.H = g->H; a.W = g->W; a.Co = g->Co; a.Ho = g->Ho; a.Wo = g->Wo;
    a.KH = g->KH; a.KW = g->KW; a.SH = g->SH; a.SW = g->SW; a.PH = g->PH; a.PW = g->PW; a.act = act;
    return launch_bf16x3<false>(a, nsplit, (hipStream_t)stream);
}

extern "C" int dynmm_conv2d_dgrad_bf16(const float* dy, const unsigned short* wd_split, int nsplit, const float* mask,
                                       const float* accum, float* dx, const dynmm_conv_geom* g, void* stream) {
    (void)hipGetLastError();
    if (!dy || !wd_split || !dx || !bf_geom_ok(g) || (nsplit != 2 && nsplit != 3)) return DYNMM_EINVAL;
    if (!dynmm_conv_bf16x3_eligible(g, 1)) return DYNMM_EUNSUPPORTED;
    BfArgs a{};
    a.x = dy; a.wsp = wd_split; a.mask = mask; a.residual = accum; a.y = dx;
    a.N = g->N; a.Ci = g->Co; a.H = g->Ho; a.W = g->Wo; a.Co = g->Ci; a.Ho = g->H; a.Wo = g->W;
    a.KH = g->KH; a.KW = g->KW; a.SH = g->SH; a.SW = g->SW; a.PH = g->PH; a.PW = g->PW; a.act = DYNMM_ACT_NONE;
    return launch_bf16x3<true>(a, nsplit, (hipStream_t)stream);
}
