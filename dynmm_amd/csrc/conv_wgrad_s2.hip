// Weight gradient of the STRIDE-2 three-tap convolutions that open encoder stages 2-4 (FusionDynMM/src/models/resnet.py:104-107:
// conv3x1_1 with stride (2,1), conv1x3_1 with stride (1,2) of the first NonBottleneck1D of a stage) on the fp32 matrix cores —
// the operand pipeline of conv_wgrad_v6.hip (direct global -> LDS loads into a 3-slot ring, hand-counted vmcnt, one barrier
// per 16-pixel step, ONE workgroup owning all three taps of its 64 input channels, k-major slabs) in the DIRECT form: with
// stride 2 along the tap axis the three taps of an output pixel read x at 2j - 1, 2j, 2j + 1 and neighbouring outputs share
// no product, so there is no Winograd saving to take (the polyphase split of the same sum is 5 contractions per pair
// against 6) — what these six launches per step lacked was the pipeline: they ran on the round-2 register-staged tiles
// (conv_wgrad_kernel / conv_wgrad_v4_kernel, 76-98 TFLOP/s).
//
// dW[co][tap][ci] = sum over dY pixels m = (n, oh, ow) of dY[co][m] * X[ci][n, SH oh + tap_h - PH, SW ow + tap_w - PW].
//   * vertical taps (3x1, stride (2,1), padding (1,0), H = 2 Ho): tap r reads image row 2 oh + r - 1 at the same column: three
//     X row sets per stage exactly like the stride-1 direct form, the loader skips every other row; only tap 0 of output row
//     0 falls outside the image (a zero quad at the fragment read).
//   * horizontal taps (1x3, stride (1,2), padding (0,1), W = 2 Wo): in the flattened pixel stream of a channel plane the x
//     pixel under tap t of dY pixel m is 2 m + t - 1 (rows are contiguous and W = 2 Wo), so a 16-pixel step stages the 40
//     x pixels [2 m0 - 4, 2 m0 + 36) of each input channel ONCE (10 quads + a padding quad: 176-byte rows, 11 r mod 16 is a
//     permutation, so ds_read_b128 stays conflict-free) and the three taps are register choices out of the 20 values a lane
//     reads: pixel i of the lane, tap t -> element 2 i + t + 3.  Only tap 0 of a row's first output is padding (W even).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "conv_igemm.h"

namespace dynmm {

template <int I>
using ic2 = std::integral_constant<int, I>;

template <int MCO, bool VT, int OCC>
__global__ void __launch_bounds__(256, OCC) conv_wgrad_s2_kernel(const WgradArgs a_in, const WgradGroup grp) {
#ifndef DYNMM_S2_NST
#define DYNMM_S2_NST 2      // two-slot operand ring since round 6 (three before): the step 0.14 ms faster on 15 of 20 alternating
                           // runs, 17 / 25 KB of LDS less per workgroup (profiles/r06_ab_runs.md); the DMA of step s + 2 is issued right
                           // behind the barrier of step s and has one step (32 - 48 MFMAs per wave) to land
#endif
    constexpr int NST = DYNMM_S2_NST;
    WgradArgs a = a_in;
    constexpr int TCO = 64 * MCO, BP = 16;
    constexpr int LDG = 20, LDX = VT ? 20 : 44;                 // row strides in floats
    constexpr int XROWS = VT ? 192 : 64;
    constexpr int G_STAGE = TCO * LDG, X_STAGE = XROWS * LDX;   // floats per ring slot
    constexpr int GW = TCO / 4;                                 // dY rows requested by one wave
    constexpr int NJG = (GW + 11) / 12;                         // wave instructions per stage: dY (12 rows of 5 quads each)
    constexpr int NJXV = 2;                                     //   vertical: 16 X rows per wave and tap, 12 per instruction
    constexpr int NJXH = 4;                                     //   horizontal: 16 X rows per wave, 5 rows of 11 quads per instruction
    constexpr int J = NJG + (VT ? 3 * NJXV : NJXH);             // loads in flight per wave and stage
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
    const int Wd = a.Wo, Hd = a.Ho, HWd = Hd * Wd;              // dY plane
    const int Wx = a.W, HWx = a.H * a.W;                        // X plane
    const int M = a.M;

    const int total_steps = (M + BP - 1) / BP;
    const int step_begin = split * a.steps_per_split;
    const int step_end = min(total_steps, step_begin + a.steps_per_split);
    const int nsteps = step_end - step_begin;

    if (t < 4) Zs[t] = 0.f;

    // ---------------------------------------------------------------- loader state (one quad per lane and instruction)
    // dY and the vertical-tap X rows: lane -> (row r5 of the instruction, quad q5; q5 == 4 is the padding quad)
    const int q5 = lane % 5, r5 = lane / 5;
    int l_m = step_begin * BP + 4 * (q5 < 4 ? q5 : 3);          // first dY pixel of the quad
    int l_rem, l_ow = 0, l_oh = 0;
    unsigned l_goff, l_xoff = 0;
    {
        const int n = l_m / HWd;
        l_rem = l_m - n * HWd;
        l_goff = ((unsigned)(n * a.Co + co0 + wave * GW + r5) * (unsigned)HWd + (unsigned)l_rem) * 4u;
        if (VT) {
            l_oh = l_rem / Wd;
            l_ow = l_rem - l_oh * Wd;
            // the CENTRE tap's quad: image row 2 oh, same column
            l_xoff = ((unsigned)(n * a.Ci + ci0 + wave * 16 + r5) * (unsigned)HWx + (unsigned)(2 * l_oh * Wx + l_ow)) * 4u;
        }
    }
    // horizontal-tap X rows: lane -> (row r11, quad q11 of 10: x pixels [2 m0 - 4, 2 m0 + 36); q11 == 10 is the padding quad)
    const int q11 = lane % 11, r11 = lane / 11;
    int h_m = 0, h_rem = 0;
    unsigned h_xoff = 0;
    if (!VT) {
        h_m = 2 * step_begin * BP - 4 + 4 * (q11 < 10 ? q11 : 9);
        const int n = h_m < 0 ? -1 : h_m / HWx;
        h_rem = h_m - n * HWx;
        h_xoff = (unsigned)(((n * a.Ci + ci0 + wave * 16 + r11) * HWx + h_rem) * 4);
    }
    const int Mx = 2 * M;                                       // x pixels per channel stream (horizontal taps: N H W = 2 M)
    const unsigned lds_g = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)Gs);
    const unsigned lds_x = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)Xs);

    auto issue = [&](int slot) __attribute__((always_inline)) {
        {
            const unsigned v = l_m < M ? l_goff : 0u;
            const unsigned dst = lds_g + (unsigned)((slot * G_STAGE + wave * GW * LDG) * 4);
#pragma unroll
            for (int i = 0; i < NJG; ++i) {
                const int rows = GW - 12 * i < 12 ? GW - 12 * i : 12;
                if (q5 < 4 && r5 < rows) dma16(a.dy + (size_t)(12 * i) * HWd, v, dst + (unsigned)(12 * i * LDG * 4));
            }
        }
        if (VT) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                // (tap 0 of output row 0 lies above the image: the centre row instead, mapped and never read)
                const bool ok = r > 0 || l_oh > 0;
                const unsigned v = l_m < M ? (ok ? l_xoff + (unsigned)((r - 1) * Wx * 4) : l_xoff) : 0u;
                const unsigned dst = lds_x + (unsigned)((slot * X_STAGE + (r * 64 + wave * 16) * LDX) * 4);
#pragma unroll
                for (int i = 0; i < NJXV; ++i) {
                    const int rows = 16 - 12 * i < 12 ? 16 - 12 * i : 12;
                    if (q5 < 4 && r5 < rows) dma16(a.x + (size_t)(12 * i) * HWx, v, dst + (unsigned)(12 * i * LDX * 4));
                }
            }
        } else {
            const unsigned v = (h_m >= 0 && h_m < Mx) ? h_xoff : 0u;
            const unsigned dst = lds_x + (unsigned)((slot * X_STAGE + wave * 16 * LDX) * 4);
#pragma unroll
            for (int i = 0; i < NJXH; ++i) {
                const int rows = 16 - 5 * i < 5 ? 16 - 5 * i : 5;
                if (q11 < 10 && r11 < rows) dma16(a.x + (size_t)(5 * i) * HWx, v, dst + (unsigned)(5 * i * LDX * 4));
            }
        }
        // advance the quad by one step
        l_m += BP; l_rem += BP; l_goff += BP * 4;
        if (VT) {
            l_xoff += BP * 4; l_ow += BP;
            if (l_ow >= Wd) { l_ow -= Wd; ++l_oh; l_xoff += (unsigned)(Wx * 4); }         // the next output row: two image rows on
        }
        if (l_rem >= HWd) {
            l_rem -= HWd;
            l_goff += (unsigned)((a.Co - 1) * HWd) * 4u;
            if (VT) { l_oh -= Hd; l_xoff += (unsigned)((a.Ci - 1) * HWx) * 4u; }
        }
        if (!VT) {
            h_m += 2 * BP; h_rem += 2 * BP; h_xoff += 2 * BP * 4;
            if (h_rem >= HWx) { h_rem -= HWx; h_xoff += (unsigned)((a.Ci - 1) * HWx) * 4u; }
        }
    };

    // ---------------------------------------------------------------- reader state
    // lane (l31, khalf): rows l31 of its wave's blocks, dY pixels [8*khalf, 8*khalf + 8) of the step = quads j = 0, 1
    int r_m = step_begin * BP + 8 * khalf;
    int r_ow, r_oh;
    {
        const int n = r_m / HWd, rem = r_m - n * HWd;
        r_oh = rem / Wd;
        r_ow = rem - r_oh * Wd;
    }
    const int rd_g = (wave_co * 32 * MCO + l31) * LDG + 8 * khalf;
    const int rd_x = (wave_k * 32 + l31) * LDX + (VT ? 8 : 16) * khalf;

    f32x16 acc[MCO][3];
#pragma unroll
    for (int mi = 0; mi < MCO; ++mi)
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[mi][s][j] = 0.f;

    const bool do_bias = a.out_bias != nullptr && (tile / a.n_co_tiles) == 0;
    float bsum = 0.f;

    // ---------------------------------------------------------------- fragments: two register sets
    // set S holds the operands of one step: dY [mi][8 pixels]; X: [tap][8 pixels] (vertical taps) or the 20 consecutive x
    // pixels under the lane's 8 dY pixels + the two row-start variants (horizontal taps).  The set of step s + 1 is read from
    // LDS under the MFMAs of step s.
    float av[2][MCO][8];
    float bx[2][VT ? 24 : 22];
    int r_step = step_begin;
    auto read_frags = [&](auto SET, int slot) __attribute__((always_inline)) {
        constexpr int S = decltype(SET)::value;
        const float* gs = Gs + slot * G_STAGE;
        const float* xs = Xs + slot * X_STAGE;
        // second quad of the lane: 4 pixels on, possibly in the next row
        int ow1 = r_ow + 4, oh1 = r_oh;
        if (ow1 >= Wd) { ow1 -= Wd; ++oh1; }
        if (oh1 >= Hd) oh1 -= Hd;
        const bool in0 = r_m < M, in1 = r_m + 4 < M;
#pragma unroll
        for (int mi = 0; mi < MCO; ++mi) {
            const float* p0 = in0 ? gs + rd_g + mi * 32 * LDG : Zs;
            const float* p1 = in1 ? gs + rd_g + mi * 32 * LDG + 4 : Zs;
            const float4 u0 = *reinterpret_cast<const float4*>(p0);
            const float4 u1 = *reinterpret_cast<const float4*>(p1);
            av[S][mi][0] = u0.x; av[S][mi][1] = u0.y; av[S][mi][2] = u0.z; av[S][mi][3] = u0.w;
            av[S][mi][4] = u1.x; av[S][mi][5] = u1.y; av[S][mi][6] = u1.z; av[S][mi][7] = u1.w;
        }
        if constexpr (VT) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const bool ok0 = r > 0 || r_oh > 0;
                const bool ok1 = r > 0 || oh1 > 0;
                const float* p0 = ok0 ? xs + rd_x + r * 64 * LDX : Zs;
                const float* p1 = ok1 ? xs + rd_x + r * 64 * LDX + 4 : Zs;
                const float4 u0 = *reinterpret_cast<const float4*>(p0);
                const float4 u1 = *reinterpret_cast<const float4*>(p1);
                bx[S][8 * r + 0] = u0.x; bx[S][8 * r + 1] = u0.y; bx[S][8 * r + 2] = u0.z; bx[S][8 * r + 3] = u0.w;
                bx[S][8 * r + 4] = u1.x; bx[S][8 * r + 5] = u1.y; bx[S][8 * r + 6] = u1.z; bx[S][8 * r + 7] = u1.w;
            }
        } else {
            // bx[i] = x pixel 2 (m0 + 8 khalf) - 4 + i of the channel stream, i < 20
#pragma unroll
            for (int qd = 0; qd < 5; ++qd) {
                const float4 u = *reinterpret_cast<const float4*>(xs + rd_x + 4 * qd);
                bx[S][4 * qd] = u.x; bx[S][4 * qd + 1] = u.y; bx[S][4 * qd + 2] = u.z; bx[S][4 * qd + 3] = u.w;
            }
            // tap 0 of a row's first output reads the left padding
            bx[S][20] = r_ow == 0 ? 0.f : bx[S][3];
            bx[S][21] = ow1 == 0 ? 0.f : bx[S][11];
        }
        if (do_bias && t < TCO) {
            const int mq = r_step * BP;
            float4 v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = *reinterpret_cast<const float4*>(gs + t * LDG + 4 * c);
            float s0;
            if (mq + BP <= M) {
                s0 = (((v[0].x + v[0].y) + (v[0].z + v[0].w)) + ((v[1].x + v[1].y) + (v[1].z + v[1].w))) +
                     (((v[2].x + v[2].y) + (v[2].z + v[2].w)) + ((v[3].x + v[3].y) + (v[3].z + v[3].w)));
            } else {                                   // last step of the tensor: quads past the end hold mapped junk
                s0 = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) s0 += mq + 4 * c < M ? (v[c].x + v[c].y) + (v[c].z + v[c].w) : 0.f;
            }
            bsum += s0;
        }
        // advance the reader by one step
        ++r_step;
        r_m += BP; r_ow += BP;
        if (r_ow >= Wd) { r_ow -= Wd; ++r_oh; }
        if (r_oh >= Hd) r_oh -= Hd;
    };
    auto mfmas = [&](auto SET) __attribute__((always_inline)) {
        constexpr int S = decltype(SET)::value;
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
            float b0, b1, b2;
            if constexpr (VT) {
                b0 = bx[S][pp]; b1 = bx[S][8 + pp]; b2 = bx[S][16 + pp];
            } else {
                b0 = pp == 0 ? bx[S][20] : (pp == 4 ? bx[S][21] : bx[S][2 * pp + 3]);
                b1 = bx[S][2 * pp + 4];
                b2 = bx[S][2 * pp + 5];
            }
#pragma unroll
            for (int mi = 0; mi < MCO; ++mi) acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[S][mi][pp], b0, acc[mi][0], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < MCO; ++mi) acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[S][mi][pp], b1, acc[mi][1], 0, 0, 0);
#pragma unroll
            for (int mi = 0; mi < MCO; ++mi) acc[mi][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[S][mi][pp], b2, acc[mi][2], 0, 0, 0);
        }
    };

    // ---------------------------------------------------------------- prologue: request stages 0 .. 2, read the set of step 0
#pragma unroll
    for (int s = 0; s < NST; ++s)
        if (s < nsteps) issue(s);
    if (nsteps > 0) {
        if (NST == 3 && nsteps >= 3) wait_vm<2 * J>(); else if (nsteps >= 2) wait_vm<J>(); else wait_vm<0>();
        __syncthreads();
        read_frags(ic2<0>{}, 0);
    }
    int slot = 0;                                   // slot of stage s
    // one step: the set of step s is in registers.  This wave's requests for stage s + 1 have landed; after the barrier so
    // have every wave's, and every wave has read the set of step s — the slot of stage s is free for stage s + 3.
    auto step = [&](auto SET, int s) __attribute__((always_inline)) {
        constexpr int S = decltype(SET)::value;
        const int next = slot == NST - 1 ? 0 : slot + 1;
        if (s + 1 < nsteps) {
            if (NST == 3 && s + 2 < nsteps) wait_vm<J>(); else wait_vm<0>();
            __syncthreads();
            if (s + NST < nsteps) issue(slot);
            read_frags(ic2<1 - S>{}, next);
        }
        mfmas(SET);
        slot = next;
    };
    for (int s = 0; s < nsteps; s += 2) {
        step(ic2<0>{}, s);
        if (s + 1 < nsteps) step(ic2<1>{}, s + 1);
    }

    if (do_bias && t < TCO) a.out_bias[(size_t)split * a.Co + co0 + t] = bsum;
    float* out = a.out + (size_t)split * a.Co * a.K;
    const int ci = ci0 + wave_k * 32 + l31;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const size_t col = a.k_major_out ? (size_t)s * a.Ci + ci : (size_t)ci * 3 + s;
        const size_t rowlen = a.k_major_out ? (size_t)a.K : (size_t)a.Ci * 3;
#pragma unroll
        for (int mi = 0; mi < MCO; ++mi)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int co = co0 + wave_co * 32 * MCO + mi * 32 + (j & 3) + 8 * (j >> 2) + 4 * khalf;
                out[(size_t)co * rowlen + col] = acc[mi][s][j];
            }
    }
}

// geometry only (pointer alignment is the launcher's business)
bool wgrad_s2_shape_ok(const dynmm_conv_geom* g) {
    const bool h_taps = g->KH == 1 && g->KW == 3 && g->SH == 1 && g->SW == 2 && g->PH == 0 && g->PW == 1 &&
                        g->W % 8 == 0 && g->Wo * 2 == g->W && g->Ho == g->H;
    const bool v_taps = g->KH == 3 && g->KW == 1 && g->SH == 2 && g->SW == 1 && g->PH == 1 && g->PW == 0 &&
                        g->H % 2 == 0 && g->Ho * 2 == g->H && g->Wo == g->W && g->W % 4 == 0;
    if (!h_taps && !v_taps) return false;
    if (g->c_split != g->Ci || g->Ci % 64 != 0 || g->Co % 64 != 0 || g->Wo < 16 || g->Ho * g->Wo < 32) return false;
    // 32-bit byte offsets inside one tensor, signed pixel counters
    const unsigned long long xs = (unsigned long long)g->N * g->Ci * g->H * g->W, ys = (unsigned long long)g->N * g->Co * g->Ho * g->Wo;
    if ((xs > ys ? xs : ys) * 4ull >= (1ull << 31)) return false;
    return true;
}

void launch_wgrad_s2(const WgradArgs& a, const WgradGroup& grp, dim3 grid, hipStream_t st) {
    const bool vt = a.KH == 3;
    const bool two = a.Co % 128 == 0;
    if (vt) {
        if (two) hipLaunchKernelGGL((conv_wgrad_s2_kernel<2, true, 2>), grid, dim3(256), 0, st, a, grp);
        else hipLaunchKernelGGL((conv_wgrad_s2_kernel<1, true, 2>), grid, dim3(256), 0, st, a, grp);
    } else {
        if (two) hipLaunchKernelGGL((conv_wgrad_s2_kernel<2, false, 2>), grid, dim3(256), 0, st, a, grp);
        else hipLaunchKernelGGL((conv_wgrad_s2_kernel<1, false, 3>), grid, dim3(256), 0, st, a, grp);
    }
}

}  // namespace dynmm
