// Weight gradient of the 3x1 convolutions (vertical taps, stride 1, 'same' padding: resnet.py:104-117) in the Winograd form
// on the fp32 matrix cores — the vertical-tap counterpart of conv_wgrad_v6_kernel<.., WINO = true> (conv_wgrad_v6.hip).
//
// The reduction runs over PAIR POSITIONS (n, r2, w): output rows 2 r2 and 2 r2 + 1 at column w.  With e0, e1 the two dY
// values of a position and d0..d3 the X values of rows 2 r2 - 1 .. 2 r2 + 2 at that column,
//     m1 = e0 (d0 - d2)    m2 = (e0 + e1)(d1 + d2) / 2    m3 = (e0 - e1)(d2 - d1) / 2    m4 = e1 (d1 - d3)
//     dW[row -1] += m1 + m2 + m3      dW[row 0] += m2 - m3      dW[row +1] += m2 + m3 - m4
// i.e. four contractions per position (= per two output pixels) instead of six: 2/3 of the direct kernel's matrix work
// AND 2/3 of its X traffic (four input rows per two output rows, where one row set per tap moves six).
//
// Structure = conv_wgrad_v6.hip's: one workgroup owns a (64 | 128) co x 64 ci tile and a range of steps of the reduction;
// a wave holds (32 | 64) co x 32 ci x 4 contractions = 4 | 8 accumulator blocks; operand rows (2 dY rows, 4 X rows per
// channel, 8 positions per step) arrive by `global_load_lds_dwordx4` into a ring (hand-counted vmcnt), one barrier per step
// (16 | 32 MFMAs per wave), the fragments of step s + 1 are read under the MFMAs of step s.  Ring depth (DYNMM_VT_NST): TWO slots
// since round 6 — the DMA of step s + 2 is issued right behind the barrier of step s and has one step to land; 49 KB of LDS per
// workgroup at Co % 128 == 0 instead of 74 KB (three slots, requested two steps ahead): the step 0.13 ms faster on 14 of 15
// alternating pairs (profiles/r06_ab_runs.md).
// LDS rows are 3 quads long (8 positions + a padding quad the loader masks off): 48-byte strides keep `ds_read_b128`
// conflict-free (3 r mod 16 is a permutation).  Rows outside the image (row -1 of the first pair, rows H / H + 1 of the last
// one when H is odd or even) and positions past the end of the tensor read an all-zero quad; their loads fetch a mapped
// row whose values are never used.  Output: the same k-major slabs [split][co][tap * Ci + ci] and bias-gradient slabs as
// v6 (the output transform runs on the accumulators; the halvings are exact).  Bit-reproducible.
//
// Round 5 built the rewrite VERDICT r4 #5 asked for — 16-position stages walking DOWN 16-column strips, X rows in row pairs
// reused by the next stage (1.0 input row per output row instead of 2.0), 64-byte row pieces in an unpadded XOR-swizzled layout,
// 64 MFMAs per barrier (scratch/r5/conv_wgrad_wino_vt_strips.hip; all tests green) — and measured it: C = 64 launches 1.69 ->
// 1.51 ms (107 -> 120 TFLOP/s), the Co % 128 == 0 group 6.21 -> 5.99 ms (131 -> 136), but the STEP 0.3 ms slower on alternating
// runs (65.96 against 65.68 ms): its two workgroups take all 160 KB of a CU's LDS, and a stage that ends a strip is followed by a
// halo item and the next stage, which the reader skips to in one move — the two-item look-ahead becomes one at every strip
// boundary, every 8 / 15 stages at C = 512 / 256.  Not kept; the fetch granularity was not what holds this kernel at 0.58.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "conv_igemm.h"

namespace dynmm {

template <int I>
using icv = std::integral_constant<int, I>;

template <int MCO>
__global__ void __launch_bounds__(256, 2) conv_wgrad_wino_vt_kernel(const WgradArgs a_in, const WgradGroup grp) {
    WgradArgs a = a_in;
#ifndef DYNMM_VT_NST
#define DYNMM_VT_NST 2
#endif
    constexpr int TCO = 64 * MCO, BP = 8, LD = 12, NST = DYNMM_VT_NST, RPI = 21;
    static_assert(NST == 2 || NST == 3, "ring depth");
    constexpr int GW = TCO / 4;                                  // dY channels requested by one wave (2 rows each)
    constexpr int G_ROWS = 2 * TCO, X_ROWS = 4 * 64;
    constexpr int G_STAGE = G_ROWS * LD, X_STAGE = X_ROWS * LD;  // floats per ring slot
    constexpr int NJG = (2 * GW + RPI - 1) / RPI;                // wave instructions per stage: dY (21 rows each)
    constexpr int NJX = (64 + RPI - 1) / RPI;                    //   X: 4 input rows x 16 channels per wave
    constexpr int J = NJG + NJX;
    static_assert(NST * J < 64, "vmcnt is a 6-bit counter");

    __shared__ __attribute__((aligned(16))) float Gs[NST * G_STAGE];
    __shared__ __attribute__((aligned(16))) float Xs[NST * X_STAGE];
    __shared__ __attribute__((aligned(16))) float Zs[4];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wave_co = wave >> 1, wave_k = wave & 1;
    const int khalf = lane >> 5, l31 = lane & 31;

    const int n_tiles = a.n_co_tiles * a.n_k_tiles;
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
    const int ci0 = (tile / a.n_co_tiles) * 64;
    const int split = lin / n_tiles;
    const int W = a.W, H = a.H, HW = H * W;
    const int H2 = (H + 1) / 2;
    const int P = a.N * H2 * W;                                   // pair positions

    const int total_steps = (P + BP - 1) / BP;
    const int step_begin = split * a.steps_per_split;
    const int step_end = min(total_steps, step_begin + a.steps_per_split);
    const int nsteps = step_end - step_begin;

    if (t < 4) Zs[t] = 0.f;

    // position -> (image, row pair, column); a quad of 4 positions never leaves its row pair (W % 4 == 0)
    struct Pos { int p, n, r2, w; };
    auto make_pos = [&](int p) {
        Pos q;
        q.p = p;
        const int per = H2 * W;
        q.n = p / per;
        const int rr = p - q.n * per;
        q.r2 = rr / W;
        q.w = rr - q.r2 * W;
        return q;
    };
    auto advance = [&](Pos& q) {                 // one step = 8 positions on (W >= 16: at most one row-pair wrap)
        q.p += BP; q.w += BP;
        if (q.w >= W) { q.w -= W; ++q.r2; }
        if (q.r2 >= H2) { q.r2 = 0; ++q.n; }
    };

    // ---------------------------------------------------------------- loader state
    // lane -> (row r3 of the instruction, quad q3; q3 == 2 is the padding quad, r3 == 21 does not exist)
    const int q3 = lane % 3, r3 = lane / 3;
    const bool l_act = q3 < 2 && r3 < RPI;
    Pos lp = make_pos(step_begin * BP + 4 * (q3 < 2 ? q3 : 1));
    // per instruction: channel byte offset and which row of the pair / of the four input rows the lane's LDS row holds
    unsigned g_c[NJG], x_c[NJX];
    int g_jd[NJG], x_j[NJX];
    bool g_on[NJG], x_on[NJX];
#pragma unroll
    for (int i = 0; i < NJG; ++i) {
        const int lr = RPI * i + r3;                               // wave-local LDS row: [jd][GW channels]
        g_on[i] = l_act && lr < 2 * GW;
        const int lrc = lr < 2 * GW ? lr : 0;
        g_jd[i] = lrc / GW;
        g_c[i] = (unsigned)((co0 + wave * GW + lrc % GW) * HW) * 4u;
    }
#pragma unroll
    for (int i = 0; i < NJX; ++i) {
        const int lr = RPI * i + r3;                               // wave-local LDS row: [j][16 channels]
        x_on[i] = l_act && lr < 64;
        const int lrc = lr < 64 ? lr : 0;
        x_j[i] = lrc / 16;
        x_c[i] = (unsigned)((ci0 + wave * 16 + lrc % 16) * HW) * 4u;
    }
    const unsigned lds_g = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)Gs);
    const unsigned lds_x = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)Xs);

    auto issue = [&](int slot) __attribute__((always_inline)) {
        const bool in = lp.p < P;
        const int n = in ? lp.n : 0, r2 = in ? lp.r2 : 0, w = in ? lp.w : 0;
        const unsigned gbase = (unsigned)(n * a.Co * HW + w) * 4u, xbase = (unsigned)(n * a.Ci * HW + w) * 4u;
        {
            const unsigned dst = lds_g + (unsigned)((slot * G_STAGE + wave * 2 * GW * LD) * 4);
#pragma unroll
            for (int i = 0; i < NJG; ++i) {
                int row = 2 * r2 + g_jd[i];
                row = row < H ? row : 2 * r2;                      // (row H of an odd image: the pair's first row, never read)
                if (g_on[i]) dma16(a.dy, gbase + g_c[i] + (unsigned)(row * W) * 4u, dst + (unsigned)(RPI * i * LD * 4));
            }
        }
        {
            const unsigned dst = lds_x + (unsigned)((slot * X_STAGE + wave * 64 * LD) * 4);
#pragma unroll
            for (int i = 0; i < NJX; ++i) {
                int row = 2 * r2 - 1 + x_j[i];
                row = (row >= 0 && row < H) ? row : 2 * r2;
                if (x_on[i]) dma16(a.x, xbase + x_c[i] + (unsigned)(row * W) * 4u, dst + (unsigned)(RPI * i * LD * 4));
            }
        }
        advance(lp);
    };

    // ---------------------------------------------------------------- reader state
    // lane (l31, khalf): channel l31 of its wave's blocks, positions [4 khalf, 4 khalf + 4) of the step = one quad per row
    Pos rp = make_pos(step_begin * BP + 4 * khalf);
    auto g_row = [&](int ch, int jd) { return ((ch / GW) * 2 + jd) * GW + ch % GW; };       // tile-local channel -> LDS row
    int rd_g[MCO][2];
#pragma unroll
    for (int mi = 0; mi < MCO; ++mi)
#pragma unroll
        for (int jd = 0; jd < 2; ++jd) rd_g[mi][jd] = g_row(wave_co * 32 * MCO + mi * 32 + l31, jd) * LD + 4 * khalf;
    int rd_x[4];
    {
        const int cl = wave_k * 32 + l31;
#pragma unroll
        for (int j = 0; j < 4; ++j) rd_x[j] = (((cl / 16) * 4 + j) * 16 + cl % 16) * LD + 4 * khalf;
    }

    f32x16 acc[MCO][4];
#pragma unroll
    for (int mi = 0; mi < MCO; ++mi)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[mi][s][j] = 0.f;

    const bool do_bias = a.out_bias != nullptr && (tile / a.n_co_tiles) == 0;
    float bsum = 0.f;
    Pos bp0 = make_pos(step_begin * BP);          // bias thread: the step's first quad (the second is 4 columns on)
    const int b_row0 = t < TCO ? g_row(t, 0) * LD : 0, b_row1 = t < TCO ? g_row(t, 1) * LD : 0;

    float4 ev[2][MCO][2], dv[2][4];               // [register set]: dY quads of both rows per block, X quads of the four rows
    auto read_frags = [&](auto SET, int slot) __attribute__((always_inline)) {
        constexpr int S = decltype(SET)::value;
        const float* gs = Gs + slot * G_STAGE;
        const float* xs = Xs + slot * X_STAGE;
        const bool in = rp.p < P;
        const bool row1 = in && 2 * rp.r2 + 1 < H;
#pragma unroll
        for (int mi = 0; mi < MCO; ++mi) {
            ev[S][mi][0] = *reinterpret_cast<const float4*>(in ? gs + rd_g[mi][0] : Zs);
            ev[S][mi][1] = *reinterpret_cast<const float4*>(row1 ? gs + rd_g[mi][1] : Zs);
        }
        dv[S][0] = *reinterpret_cast<const float4*>((in && rp.r2 > 0) ? xs + rd_x[0] : Zs);
        dv[S][1] = *reinterpret_cast<const float4*>(in ? xs + rd_x[1] : Zs);
        dv[S][2] = *reinterpret_cast<const float4*>(row1 ? xs + rd_x[2] : Zs);
        dv[S][3] = *reinterpret_cast<const float4*>((in && 2 * rp.r2 + 2 < H) ? xs + rd_x[3] : Zs);
        if (do_bias && t < TCO) {
            float s0 = 0.f;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                Pos q = bp0;
                if (c == 1) { q.p += 4; q.w += 4; if (q.w >= W) { q.w -= W; ++q.r2; } if (q.r2 >= H2) { q.r2 = 0; ++q.n; } }
                if (q.p < P) {
                    const float4 u = *reinterpret_cast<const float4*>(gs + b_row0 + 4 * c);
                    s0 += (u.x + u.y) + (u.z + u.w);
                    if (2 * q.r2 + 1 < H) {
                        const float4 v = *reinterpret_cast<const float4*>(gs + b_row1 + 4 * c);
                        s0 += (v.x + v.y) + (v.z + v.w);
                    }
                }
            }
            bsum += s0;
        }
        advance(rp);
        advance(bp0);
    };
    auto mfmas = [&](auto SET) __attribute__((always_inline)) {
        constexpr int S = decltype(SET)::value;
        const float d0[4] = {dv[S][0].x, dv[S][0].y, dv[S][0].z, dv[S][0].w}, d1[4] = {dv[S][1].x, dv[S][1].y, dv[S][1].z, dv[S][1].w};
        const float d2[4] = {dv[S][2].x, dv[S][2].y, dv[S][2].z, dv[S][2].w}, d3[4] = {dv[S][3].x, dv[S][3].y, dv[S][3].z, dv[S][3].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v0 = d0[j] - d2[j], v1 = d1[j] + d2[j], v2 = d2[j] - d1[j], v3 = d1[j] - d3[j];
#pragma unroll
            for (int mi = 0; mi < MCO; ++mi) {
                const float e0a[4] = {ev[S][mi][0].x, ev[S][mi][0].y, ev[S][mi][0].z, ev[S][mi][0].w};
                const float e1a[4] = {ev[S][mi][1].x, ev[S][mi][1].y, ev[S][mi][1].z, ev[S][mi][1].w};
                const float e0 = e0a[j], e1 = e1a[j];
                acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(e0, v0, acc[mi][0], 0, 0, 0);
                acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(e0 + e1, v1, acc[mi][1], 0, 0, 0);
                acc[mi][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(e0 - e1, v2, acc[mi][2], 0, 0, 0);
                acc[mi][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(e1, v3, acc[mi][3], 0, 0, 0);
            }
        }
    };

    // ---------------------------------------------------------------- prologue: request stages 0 .. 2, read the set of step 0
#pragma unroll
    for (int s = 0; s < NST; ++s)
        if (s < nsteps) issue(s);
    if (nsteps > 0) {
        if (NST == 3 && nsteps >= 3) wait_vm<2 * J>(); else if (nsteps >= 2) wait_vm<J>(); else wait_vm<0>();
        __syncthreads();
        read_frags(icv<0>{}, 0);
    }
    int slot = 0;
    auto step = [&](auto SET, int s) __attribute__((always_inline)) {
        constexpr int S = decltype(SET)::value;
        const int next = slot == NST - 1 ? 0 : slot + 1;
        if (s + 1 < nsteps) {
            if (NST == 3 && s + 2 < nsteps) wait_vm<J>(); else wait_vm<0>();
            __syncthreads();
            if (s + NST < nsteps) issue(slot);
            read_frags(icv<1 - S>{}, next);
        }
        mfmas(SET);
        slot = next;
    };
    for (int s = 0; s < nsteps; s += 2) {
        step(icv<0>{}, s);
        if (s + 1 < nsteps) step(icv<1>{}, s + 1);
    }

    if (do_bias && t < TCO) a.out_bias[(size_t)split * a.Co + co0 + t] = bsum;
    const int KHKW = 3;
    float* out = a.out + (size_t)split * a.Co * a.K;
    const int ci = ci0 + wave_k * 32 + l31;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const size_t col = a.k_major_out ? (size_t)(s * a.Ci + ci) : (size_t)ci * KHKW + s;
        const size_t rowlen = a.k_major_out ? (size_t)a.K : (size_t)a.Ci * KHKW;
#pragma unroll
        for (int mi = 0; mi < MCO; ++mi)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int co = co0 + wave_co * 32 * MCO + mi * 32 + (j & 3) + 8 * (j >> 2) + 4 * khalf;
                const float hs = 0.5f * (acc[mi][1][j] + acc[mi][2][j]);      // (m2, m3 were accumulated without their 1/2)
                const float v = s == 0 ? acc[mi][0][j] + hs : (s == 1 ? 0.5f * (acc[mi][1][j] - acc[mi][2][j]) : hs - acc[mi][3][j]);
                out[(size_t)co * rowlen + col] = v;
            }
    }
}

// does conv_wgrad_v6's launcher hand this (v6-eligible) geometry to the kernel above?  (the plan needs to know: its
// reduction runs over pair positions in steps of 8, not over pixels in steps of 16)
bool wgrad_wino_vt_on(const dynmm_conv_geom* g) {
    return g->KH == 3 && g->KW == 1 && g->SH == 1;          // (stride 2: conv_wgrad_s2.hip, 16-pixel steps)
}

int wgrad_wino_vt_units(const dynmm_conv_geom* g) { return g->N * ((g->H + 1) / 2) * g->W; }

int wgrad_wino_vt_bp() { return 8; }                        // plan units per step: 8 pair positions

void launch_wgrad_wino_vt(const WgradArgs& a, const WgradGroup& grp, dim3 grid, hipStream_t st) {
    if (a.Co % 128 == 0) hipLaunchKernelGGL((conv_wgrad_wino_vt_kernel<2>), grid, dim3(256), 0, st, a, grp);
    else hipLaunchKernelGGL((conv_wgrad_wino_vt_kernel<1>), grid, dim3(256), 0, st, a, grp);
}

}  // namespace dynmm
