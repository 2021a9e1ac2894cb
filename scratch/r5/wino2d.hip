// EXPERIMENT (round 5): 2-D Winograd F(2x2, 3x3) forward on v_mfma_f32_16x16x4_f32 — 16 channel contractions per 2x2 output tile
// (4 per output) instead of 9 (direct) or 6 (the product's horizontal F(2,3) with the vertical taps looped).
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A,  d = the 4x4 input patch of the tile
// Workgroup: 64 co x 32 tiles (consecutive in the flattened (image, tile row, tile column) order); wave = 32 co x 16 tiles x 16
// transforms = 32 accumulator blocks of 16x16 (128 registers).  Stage = 4 input channels = one MFMA k-step = 32 MFMAs per wave.
// Filter operand U [ci][tq 4][co][4] (t = 4 tq + j = 4 i + c: vertical index i, horizontal index c), one ds_read_b128 per
// (co block, tq); raw input tile [4 ch][4 rows][72 positions] (64 + a 4-position halo either side, positions = 2 x tile index in
// the flattened order), a lane reads d[r][0], (d[r][1], d[r][2]), d[r][3] per row.
#include <stdlib.h>

#include "common.h"
#include "conv_igemm.h"

namespace dynmm {

typedef float f32x4v __attribute__((ext_vector_type(4)));

struct Wino2dArgs {
    const float* x;         // [N, Ci, H, W]
    const float* ut;        // [Ci][4][CoS][4]
    const float* shift;     // [Co] or nullptr
    const float* residual;  // like y or nullptr (added before the activation)
    float* y;               // [N, Co, H, W]
    int N, Ci, Co, H, W, CoS, act;
    int TH, TW, MT;         // tile rows per image, tile columns, tiles
    int n_co_tiles, n_t_tiles;
};

__global__ void __launch_bounds__(256, 2) conv_wino2d_kernel(const Wino2dArgs a) {
    constexpr int BK = 4, S = 3, TCO = 64, TT = 32;
    constexpr int A_STAGE = BK * 4 * TCO * 4;                  // floats (16 KB)
    constexpr int PW = 2 * TT + 8, BROW = PW, BCH = 4 * BROW;  // 72 positions per row, 4 rows per channel
    constexpr int B_STAGE = BK * BCH;                          // 1152 floats
    constexpr int QB = B_STAGE / 4, QPW = QB / 4;              // 288 quads per stage, 72 per wave
    constexpr int NIB = (QPW + 63) / 64, NIA = 4, NI = NIA + NIB;
    __shared__ __attribute__((aligned(16))) float As[S * A_STAGE];
    __shared__ __attribute__((aligned(16))) float Bs[S * B_STAGE];
    __shared__ __attribute__((aligned(16))) float Zs[PW + 8];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wave_co = wave >> 1, wave_t = wave & 1;
    const int l15 = lane & 15, kq = lane >> 4;

    const int nblk = a.n_co_tiles * a.n_t_tiles;
    const int lin = xcd_remap((int)blockIdx.x, nblk);
    const int co0 = (lin % a.n_co_tiles) * TCO;
    const int t0 = (lin / a.n_co_tiles) * TT;
    const int HW = a.H * a.W;
    const int nst = a.Ci / BK;

    for (int i = t; i < PW + 8; i += 256) Zs[i] = 0.f;

    // ---------------------------------------------------------------- loader
    unsigned b_off[NIB];
    bool b_act[NIB];
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
        const int ql = i * 64 + lane;
        b_act[i] = ql < QPW;
        const int q = wave * QPW + (b_act[i] ? ql : 0);
        const int k = q / (4 * 18), r = (q / 18) % 4, j = q % 18;
        int Q = 2 * t0 - 4 + 4 * j;                            // flattened (image, tile row, x) position of the quad
        const int Qmax = 2 * a.MT - 4;
        Q = Q < 0 ? 0 : (Q > Qmax ? Qmax : Q);
        const int rowid = Q / a.W, xx = Q - rowid * a.W;
        const int n = rowid / a.TH, th = rowid - n * a.TH;
        int h = 2 * th - 1 + r;
        h = h < 0 ? 0 : (h > a.H - 1 ? a.H - 1 : h);           // (a row outside the image: mapped, the reader takes the zero row)
        b_off[i] = ((unsigned)(n * a.Ci + k) * (unsigned)HW + (unsigned)(h * a.W + xx)) * 4u;
    }
    const unsigned a_voff = (unsigned)lane * 16u;
    const unsigned lds_a = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)As);
    const unsigned lds_b = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)Bs);
    // wave w stages channel w of the stage: 4 (tq) pieces of 64 co x 4 floats = 1 KB each
    const float* a_ptr = a.ut + ((size_t)(wave * 4) * a.CoS + co0) * 4;
    const size_t a_step = (size_t)BK * 4 * a.CoS * 4, a_tq = (size_t)a.CoS * 4;
    const float* b_ptr = a.x;
    const size_t b_step = (size_t)BK * HW;
    unsigned l_adst = lds_a + (unsigned)(wave * 4 * TCO * 4 * 4), l_bdst = lds_b + (unsigned)(wave * QPW * 4 * 4);
    const unsigned l_adst_end = l_adst + (unsigned)(S * A_STAGE * 4);
    int l_left = nst;
    auto issue = [&]() {
        if (l_left > 0) {
#pragma unroll
            for (int i = 0; i < NIA; ++i) dma16(a_ptr + (size_t)i * a_tq, a_voff, l_adst + (unsigned)i * 1024u);
#pragma unroll
            for (int i = 0; i < NIB; ++i)
                if (b_act[i]) dma16(b_ptr, b_off[i], l_bdst + (unsigned)i * 1024u);
            --l_left;
            a_ptr += a_step;
            b_ptr += b_step;
            l_adst += (unsigned)(A_STAGE * 4);
            l_bdst += (unsigned)(B_STAGE * 4);
            if (l_adst == l_adst_end) {
                l_adst -= (unsigned)(S * A_STAGE * 4);
                l_bdst -= (unsigned)(S * B_STAGE * 4);
            }
        }
    };

    // ---------------------------------------------------------------- consumer state
    const int tl = wave_t * 16 + l15;                           // tile of this lane inside the workgroup's 32
    const int tg = t0 + tl;
    const bool tvalid = tg < a.MT;
    int pn, pth, ptw;
    {
        const int tc = tvalid ? tg : 0;
        const int rowid = tc / a.TW;
        ptw = tc - rowid * a.TW;
        pn = rowid / a.TH;
        pth = rowid - pn * a.TH;
    }
    const bool c0ok = ptw > 0, c3ok = ptw < a.TW - 1;
    const bool r0ok = pth > 0, r2ok = 2 * pth + 1 < a.H, r3ok = 2 * pth + 2 < a.H;
    // fragment addresses (floats) inside a stage slot
    const int a_frag = ((kq * 4) * TCO + wave_co * 32 + l15) * 4;        // + (tq * TCO + cb * 16) * 4
    const int b_frag = kq * BCH + 2 * tl + 3;                            // + r * BROW
    // per-row base: the raw rows, or the zero row where the input row lies outside the image
    const int zoff = 2 * tl + 3;

    f32x4v acc[16][2];
#pragma unroll
    for (int tt = 0; tt < 16; ++tt)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[tt][cb][j] = 0.f;

    float4 fa[2][2][4];                            // [set][co block][tq]
    float fd[4][4];                                // raw patch [row][col] of the NEXT stage
    float fv[2][16];                               // transformed patch [set][4 i + c]
    auto read_frags = [&](int set, const float* Ap, const float* Bp) {
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int tq = 0; tq < 4; ++tq)
                fa[set][cb][tq] = *reinterpret_cast<const float4*>(Ap + a_frag + (tq * TCO + cb * 16) * 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool rok = r == 0 ? r0ok : (r == 1 ? true : (r == 2 ? r2ok : r3ok));
            const float* b = rok ? Bp + b_frag + r * BROW : Zs + zoff;
            const float2 u = *reinterpret_cast<const float2*>(b + 1);
            fd[r][0] = b[0];
            fd[r][1] = u.x;
            fd[r][2] = u.y;
            fd[r][3] = b[3];
        }
    };
    auto transform = [&](int set) {
        float tr[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float d0 = c0ok ? fd[r][0] : 0.f, d1 = fd[r][1], d2 = fd[r][2], d3 = c3ok ? fd[r][3] : 0.f;
            tr[r][0] = d0 - d2;
            tr[r][1] = d1 + d2;
            tr[r][2] = d2 - d1;
            tr[r][3] = d1 - d3;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            fv[set][0 + c] = tr[0][c] - tr[2][c];
            fv[set][4 + c] = tr[1][c] + tr[2][c];
            fv[set][8 + c] = tr[2][c] - tr[1][c];
            fv[set][12 + c] = tr[1][c] - tr[3][c];
        }
    };
    auto mfma_set = [&](int set) {
#pragma unroll
        for (int tq = 0; tq < 4; ++tq)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const float av[4] = {fa[set][cb][tq].x, fa[set][cb][tq].y, fa[set][cb][tq].z, fa[set][cb][tq].w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[4 * tq + j][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], fv[set][4 * tq + j], acc[4 * tq + j][cb], 0, 0, 0);
            }
    };

    // ---------------------------------------------------------------- K loop: the fragments of stage s + 1 are read under the MFMAs of stage s
    issue();
    issue();
    issue();
    if (nst >= 3) wait_vm<2 * NI>(); else if (nst == 2) wait_vm<NI>(); else wait_vm<0>();
    __syncthreads();
    int c_a = 0, c_b = 0;
    read_frags(0, As, Bs);
    transform(0);
    auto stage = [&](int set, int s) __attribute__((always_inline)) {
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < nst) {
            if (s + 2 < nst) wait_vm<NI>(); else wait_vm<0>();
            __syncthreads();
            issue();
            c_a += A_STAGE;
            c_b += B_STAGE;
            if (c_a == S * A_STAGE) { c_a = 0; c_b = 0; }
            read_frags(set ^ 1, As + c_a, Bs + c_b);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_set(set);
        __builtin_amdgcn_sched_barrier(0);
        transform(set ^ 1);
    };
    for (int s = 0; s < nst; s += 2) {
        stage(0, s);
        if (s + 1 < nst) stage(1, s + 1);
    }

    // ---------------------------------------------------------------- epilogue
    if (!tvalid) return;
    const bool y1ok = r2ok;                                     // the tile's second output row exists
    const size_t pix = (size_t)(2 * pth) * a.W + 2 * ptw;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = co0 + wave_co * 32 + cb * 16 + 4 * kq + i;
            float z[2][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float m0 = acc[c][cb][i], m1 = acc[4 + c][cb][i], m2 = acc[8 + c][cb][i], m3 = acc[12 + c][cb][i];
                z[0][c] = (m0 + m1) + m2;
                z[1][c] = (m1 - m2) - m3;
            }
            const float sh = a.shift ? a.shift[co] : 0.f;
            float y00 = (z[0][0] + z[0][1]) + z[0][2] + sh, y01 = (z[0][1] - z[0][2]) - z[0][3] + sh;
            float y10 = (z[1][0] + z[1][1]) + z[1][2] + sh, y11 = (z[1][1] - z[1][2]) - z[1][3] + sh;
            const size_t off = ((size_t)pn * a.Co + co) * HW + pix;
            if (a.residual) {
                const float2 r0 = *reinterpret_cast<const float2*>(a.residual + off);
                y00 += r0.x; y01 += r0.y;
                if (y1ok) {
                    const float2 r1 = *reinterpret_cast<const float2*>(a.residual + off + a.W);
                    y10 += r1.x; y11 += r1.y;
                }
            }
            if (a.act == DYNMM_ACT_RELU) {
                y00 = y00 > 0.f ? y00 : 0.f; y01 = y01 > 0.f ? y01 : 0.f;
                y10 = y10 > 0.f ? y10 : 0.f; y11 = y11 > 0.f ? y11 : 0.f;
            }
            *reinterpret_cast<float2*>(a.y + off) = make_float2(y00, y01);
            if (y1ok) *reinterpret_cast<float2*>(a.y + off + a.W) = make_float2(y10, y11);
        }
}

// w [Co][Ci][3][3] -> ut [Ci][4 tq][CoS][4]: U = G g G^T, t = 4 i + c
__global__ void __launch_bounds__(256) wino2d_pack_kernel(const float* __restrict__ w, float4* __restrict__ ut, int Co, int Ci, int CoS) {
    const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)Ci * 4 * CoS;
    if (o >= total) return;
    const int co = (int)(o % CoS);
    const int tq = (int)((o / CoS) % 4);
    const int ci = (int)(o / ((size_t)CoS * 4));
    if (co >= Co) { ut[o] = make_float4(0.f, 0.f, 0.f, 0.f); return; }
    const float* g = w + ((size_t)co * Ci + ci) * 9;
    // rows of G g: [g0; (g0+g1+g2)/2; (g0-g1+g2)/2; g2] over the vertical index
    float gr[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const float g0 = g[b], g1 = g[3 + b], g2 = g[6 + b];
        gr[b] = tq == 0 ? g0 : (tq == 1 ? (g0 + g1 + g2) * 0.5f : (tq == 2 ? (g0 - g1 + g2) * 0.5f : g2));
    }
    ut[o] = make_float4(gr[0], (gr[0] + gr[1] + gr[2]) * 0.5f, (gr[0] - gr[1] + gr[2]) * 0.5f, gr[2]);
}

}  // namespace dynmm

using namespace dynmm;

extern "C" size_t exp_wino2d_packed_floats(int Co, int Ci) { return (size_t)Ci * 4 * ((Co + 63) & ~63) * 4; }

extern "C" int exp_wino2d_pack(const float* w, float* ut, int Co, int Ci, void* stream) {
    const int CoS = (Co + 63) & ~63;
    const size_t total = (size_t)Ci * 4 * CoS;
    hipLaunchKernelGGL(wino2d_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                       reinterpret_cast<float4*>(ut), Co, Ci, CoS);
    return 0;
}

extern "C" int exp_conv2d_wino2d_fwd(const float* x, const float* ut, const float* bias, const float* residual, float* y, int N, int Ci,
                                     int H, int W, int Co, int act, void* stream) {
    if (Ci % 4 != 0 || Ci < 12 || Co % 64 != 0 || W % 4 != 0 || H < 2) return -2;
    Wino2dArgs a{};
    a.x = x; a.ut = ut; a.shift = bias; a.residual = residual; a.y = y;
    a.N = N; a.Ci = Ci; a.Co = Co; a.H = H; a.W = W; a.CoS = (Co + 63) & ~63; a.act = act;
    a.TH = (H + 1) / 2; a.TW = W / 2; a.MT = N * a.TH * a.TW;
    a.n_co_tiles = Co / 64; a.n_t_tiles = (a.MT + 31) / 32;
    hipLaunchKernelGGL(conv_wino2d_kernel, dim3((unsigned)(a.n_co_tiles * a.n_t_tiles)), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? 0 : -1000;
}
