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
// Structure = conv_wgrad_v6.hip's tile: one workgroup owns a (64 | 128) co x 64 ci tile and a range of STAGES of the reduction;
// a wave holds (32 | 64) co x 32 ci x 4 contractions = 4 | 8 accumulator blocks; output: the same k-major slabs
// [split][co][tap * Ci + ci] and bias-gradient slabs as v6 (the output transform runs on the accumulators; the halvings are
// exact).  Bit-reproducible.
//
// COLUMN STRIPS (built in round 5, in the product since round 6).  A stage is 16 positions — four quads of four columns,
// consecutive in the flattened (image, column quad) order — of ONE row pair, and consecutive stages walk DOWN the image inside
// that 16-column strip: the X rows come in row PAIRS k = (2k - 1, 2k), stage r2 uses pairs r2 and r2 + 1 and only pair r2 + 1 is
// new (1.0 input row per output row; the round-4 kernel walked row pair by row pair, 8 columns per step, and staged all four X
// rows of every step: 2.0).  Operand rows are 64-byte pieces, unpadded in LDS: the loader permutes the SOURCE quads of a row
// (lane -> quad q ^ ((row >> 2) & 3)), which makes the readers' ds_read_b128 over 32 consecutive rows conflict-free without a
// padding quad (a DMA lane's LDS address is fixed, its global address is not).  One barrier per stage = 64 | 32 MFMAs per wave.
// Zero padding (row -1, rows >= H, quads past the tensor) is a component-wise register select after the read (a select between
// float4 OBJECTS goes through private memory, which the compiler then parks in 12 KB of LDS).
//
// Round 5 measured its first form — separate "halo" items for the first pair of a strip, uniform items, TWO items in flight, a
// 3-slot dY ring and a 4-slot X ring = 80 KB per workgroup — faster in isolation and 0.3 ms SLOWER in the step, and shelved it.
// Round 6 (the weight-gradient launches run beside the backward's dependent chain, and what they keep in flight costs that
// chain: the 2-slot rings of conv_wgrad_v6.hip / conv_wgrad_s2.hip): ONE stage in flight — every wait is vmcnt(0) —, a stage that
// starts a strip (or this workgroup's range) simply requests both of its pairs, 2 dY slots + 4 X-pair slots = 64 | 48 KB.
// Measured (profiles/r06_ab_runs.md): the 3x1 weight gradients 428 -> 389 us per grouped launch at C = 256 (141 -> 155 TF/s
// algorithmic), 444 -> 393 at C = 128, 576 -> 467 at C = 64 (105 -> 129), 455 -> 423 at C = 512 — the horizontal kernel's rates —
// and the step 0.21 ms faster on 8 of 10 alternating pairs (round 5's two-in-flight form, same day, same box: +0.45 ms).
// What did NOT improve is the fabric-side byte count: FETCH_SIZE 329 -> 444 MiB per launch (raw) — a 64-byte piece is half a
// 128-byte line and the other half belongs to the neighbouring strip, H2 stages away, so the lines are fetched twice, where the
// pair kernel's 32-byte pieces shared their line with the next three steps.  The over-fetch the round-5 review asked to close
// (1.83x the algorithmic bytes) moved from the X rows (each fetched for two row pairs) to the line granularity (everything
// fetched twice) and is not closed; a 32-column strip would close it and needs 128 KB of LDS per workgroup.
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
    constexpr int TCO = 64 * MCO, ROWF = 16;
    constexpr int G_ROWS = 2 * TCO, X_ROWS = 2 * 64;
    constexpr int G_SLOT = G_ROWS * ROWF, X_SLOT = X_ROWS * ROWF;   // floats
    constexpr int NGS = 2, NXS = 4;                                 // round 6: ONE stage ahead (2 dY slots; X pairs: 2 live + up to 2 new)
    constexpr int JG = G_ROWS / 64, JX = X_ROWS / 64;               // wave instructions per row set (16 rows x 64 bytes each)
    constexpr int J = JG + JX;
    static_assert(J + JX < 64, "vmcnt is a 6-bit counter");

    __shared__ __attribute__((aligned(16))) float Gs[NGS * G_SLOT];
    __shared__ __attribute__((aligned(16))) float Xs[NXS * X_SLOT];

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
    const int WQ = W / 4;
    const int QC = a.N * WQ;                                       // column quads of the tensor
    const int groups = (QC + 3) / 4;
    const int total_stages = groups * H2;
    const int sb = split * a.steps_per_split;
    const int se = min(total_stages, sb + a.steps_per_split);

    // ---------------------------------------------------------------- the stage sequence (round 6)
    // Stage s = (strip g, row pair r2).  Its X operand is the row pairs k = r2 and r2 + 1 (rows 2k - 1, 2k); pair r2 is the
    // previous stage's pair r2 + 1 and already in LDS — except for the FIRST stage of a strip or of this workgroup's range, whose
    // request also brings pair r2 (round 5 had a separate "halo" item for it, uniform items and two items in flight; now one
    // stage is in flight, every wait is vmcnt(0), and a stage that starts a strip is simply a longer request).
    struct Seq { int s, g, r2; };
    auto seq_init = [&](Seq& q) { q.s = sb; q.g = sb / H2; q.r2 = sb - q.g * H2; };
    auto seq_next = [&](Seq& q) { ++q.s; if (++q.r2 == H2) { q.r2 = 0; ++q.g; } };
    auto seq_first = [&](const Seq& q) { return q.s == sb || q.r2 == 0; };

    // ---------------------------------------------------------------- loader
    // instruction i of a wave covers LDS rows [16 (wave * J? + i) ...): dY rows first (JG per wave), then X rows (JX per wave);
    // lane L: row L / 4 of the instruction, destination quad L % 4, SOURCE quad (L % 4) ^ ((row >> 2) & 3)
    int l_row[J], l_ch[J], l_sq[J];
#pragma unroll
    for (int i = 0; i < J; ++i) {
        const bool isg = i < JG;
        const int row = (isg ? (wave * JG + i) : (wave * JX + (i - JG))) * 16 + (lane >> 2);     // row of the slot
        l_row[i] = isg ? row / TCO : row / 64;                      // 0 | 1: which row of the pair
        l_ch[i] = isg ? co0 + row % TCO : ci0 + row % 64;
        l_sq[i] = (lane & 3) ^ ((row >> 2) & 3);
    }
    const unsigned lds_g = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)Gs);
    const unsigned lds_x = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)Xs);
    Seq lq;
    seq_init(lq);
    int l_g = -1;
    unsigned l_col[J];                                             // byte offset of (image, channel plane, column) per instruction
    int l_stage = 0, l_xw = 0;                                     // stages requested; X pairs requested (ring position)
    auto issue = [&]() __attribute__((always_inline)) {
        if (lq.s >= se) return;
        if (lq.g != l_g) {                                          // a new strip: this lane's column quads
            l_g = lq.g;
#pragma unroll
            for (int i = 0; i < J; ++i) {
                int qc = 4 * l_g + l_sq[i];
                qc = qc < QC ? qc : QC - 1;                         // (a quad past the tensor: mapped, never used)
                const int n = qc / WQ, wq = qc - n * WQ;
                const int C = i < JG ? a.Co : a.Ci;
                l_col[i] = ((unsigned)(n * C + l_ch[i]) * (unsigned)HW + (unsigned)(4 * wq)) * 4u;
            }
        }
        const unsigned gdst = lds_g + (unsigned)(((l_stage % NGS) * G_SLOT + wave * JG * 16 * ROWF) * 4);
#pragma unroll
        for (int i = 0; i < JG; ++i) {
            int row = 2 * lq.r2 + l_row[i];
            row = row > H - 1 ? H - 1 : row;                        // (outside the image: a mapped row, zeroed at the read)
            dma16(a.dy, l_col[i] + (unsigned)(row * W) * 4u, gdst + (unsigned)(i * 1024));
        }
        auto load_pair = [&](int kx) __attribute__((always_inline)) {     // X rows 2 kx - 1, 2 kx of this lane's channels
            const unsigned xdst = lds_x + (unsigned)(((l_xw % NXS) * X_SLOT + wave * JX * 16 * ROWF) * 4);
#pragma unroll
            for (int i = JG; i < J; ++i) {
                int row = 2 * kx - 1 + l_row[i];
                row = row < 0 ? 0 : (row > H - 1 ? H - 1 : row);
                dma16(a.x, l_col[i] + (unsigned)(row * W) * 4u, xdst + (unsigned)((i - JG) * 1024));
            }
            ++l_xw;
        };
        if (seq_first(lq)) load_pair(lq.r2);                        // (first stage of a strip / of this workgroup's range only)
        load_pair(lq.r2 + 1);
        ++l_stage;
        seq_next(lq);
    };

    // ---------------------------------------------------------------- reader
    // lane (l31, khalf), half-step hs: source quad q = 2 hs + khalf of row R sits at float offset R * 16 + 4 * (q ^ ((R >> 2) & 3))
    int rd_g[MCO][2], rd_x[2];                                     // row bases (floats) of this lane's rows; swizzle key per row
    int sw_g[MCO][2], sw_x[2];
#pragma unroll
    for (int mi = 0; mi < MCO; ++mi)
#pragma unroll
        for (int jd = 0; jd < 2; ++jd) {
            const int R = jd * TCO + wave_co * 32 * MCO + mi * 32 + l31;
            rd_g[mi][jd] = R * ROWF;
            sw_g[mi][jd] = (R >> 2) & 3;
        }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int R = j * 64 + wave_k * 32 + l31;
        rd_x[j] = R * ROWF;
        sw_x[j] = (R >> 2) & 3;
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
    const int b_R0 = t < TCO ? t : 0, b_R1 = t < TCO ? TCO + t : 0;

    float4 ev[2][MCO][2], dv[2][4];               // [register set]: dY quads of both rows per block, X quads of the four rows
    // (component-wise: a select between float4 OBJECTS goes through private memory, which the compiler then parks in LDS)
    auto sel4 = [](bool c, const float4& v) __attribute__((always_inline)) { return make_float4(c ? v.x : 0.f, c ? v.y : 0.f, c ? v.z : 0.f, c ? v.w : 0.f); };
    Seq rq;                                       // the reader's stage
    seq_init(rq);
    int r_stage = 0, r_xw = 1;                    // its index in this workgroup's range; ring position of its pair r2 + 1
    auto read_frags = [&](auto SET, int hs) __attribute__((always_inline)) {
        constexpr int S = decltype(SET)::value;
        const float* gs = Gs + (r_stage % NGS) * G_SLOT;
        const float* x1 = Xs + (r_xw % NXS) * X_SLOT;                         // pair r2 + 1
        const float* x0 = Xs + ((r_xw + NXS - 1) % NXS) * X_SLOT;             // pair r2 (requested just before it)
        const int q = 2 * hs + khalf;
        const bool in = 4 * rq.g + q < QC;
        const bool row1 = in && 2 * rq.r2 + 1 < H;
#pragma unroll
        for (int mi = 0; mi < MCO; ++mi) {
            const float4 e0 = *reinterpret_cast<const float4*>(gs + rd_g[mi][0] + 4 * (q ^ sw_g[mi][0]));
            const float4 e1 = *reinterpret_cast<const float4*>(gs + rd_g[mi][1] + 4 * (q ^ sw_g[mi][1]));
            ev[S][mi][0] = sel4(in, e0);
            ev[S][mi][1] = sel4(row1, e1);
        }
        const float4 d0 = *reinterpret_cast<const float4*>(x0 + rd_x[0] + 4 * (q ^ sw_x[0]));
        const float4 d1 = *reinterpret_cast<const float4*>(x0 + rd_x[1] + 4 * (q ^ sw_x[1]));
        const float4 d2 = *reinterpret_cast<const float4*>(x1 + rd_x[0] + 4 * (q ^ sw_x[0]));
        const float4 d3 = *reinterpret_cast<const float4*>(x1 + rd_x[1] + 4 * (q ^ sw_x[1]));
        dv[S][0] = sel4(in && rq.r2 > 0, d0);
        dv[S][1] = sel4(in, d1);
        dv[S][2] = sel4(row1, d2);
        dv[S][3] = sel4(in && 2 * rq.r2 + 2 < H, d3);
        if (do_bias && t < TCO && hs == 0) {
            float s0 = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (4 * rq.g + c < QC) {
                    const float4 u = *reinterpret_cast<const float4*>(gs + b_R0 * ROWF + 4 * (c ^ ((b_R0 >> 2) & 3)));
                    s0 += (u.x + u.y) + (u.z + u.w);
                    if (2 * rq.r2 + 1 < H) {
                        const float4 v = *reinterpret_cast<const float4*>(gs + b_R1 * ROWF + 4 * (c ^ ((b_R1 >> 2) & 3)));
                        s0 += (v.x + v.y) + (v.z + v.w);
                    }
                }
            }
            bsum += s0;
        }
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
    // the reader moves to the next stage: waits for it (the only request in flight), passes the barrier — every wave has the
    // fragments of the stage left behind in registers, its dY slot and its older X pair are free — and requests the stage after
    auto advance = [&]() __attribute__((always_inline)) {
        seq_next(rq);
        ++r_stage;
        r_xw += seq_first(rq) ? 2 : 1;
        wait_vm<0>();
        __syncthreads();
        issue();
    };

    // ---------------------------------------------------------------- pipeline
    const int n_stages = se - sb;
    if (n_stages > 0) {
        issue();                                    // stage sb (a first stage: both pairs)
        issue();                                    // stage sb + 1 stays in flight under stage sb
        // wait for stage sb only: what is in flight behind it is stage sb + 1 = J instructions, J + JX when it starts a strip
        {
            Seq nx;
            seq_init(nx);
            seq_next(nx);
            if (n_stages < 2) wait_vm<0>();
            else if (nx.r2 == 0) wait_vm<J + JX>();
            else wait_vm<J>();
        }
        __syncthreads();
        read_frags(icv<0>{}, 0);
        for (int s = 0; s < n_stages; ++s) {
            // half-step 0 of stage s is in set 0
            read_frags(icv<1>{}, 1);
            mfmas(icv<0>{});
            if (s + 1 < n_stages) {
                advance();
                read_frags(icv<0>{}, 0);
            }
            mfmas(icv<1>{});
        }
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

int wgrad_wino_vt_bp() { return 16; }                       // plan units per step: one stage = 16 positions

// reduction units for the plan: 16 positions per stage, stages = column-quad groups x row pairs
int wgrad_wino_vt_units(const dynmm_conv_geom* g) { return ((g->N * (g->W / 4) + 3) / 4) * ((g->H + 1) / 2) * 16; }

void launch_wgrad_wino_vt(const WgradArgs& a, const WgradGroup& grp, dim3 grid, hipStream_t st) {
    if (a.Co % 128 == 0) hipLaunchKernelGGL((conv_wgrad_wino_vt_kernel<2>), grid, dim3(256), 0, st, a, grp);
    else hipLaunchKernelGGL((conv_wgrad_wino_vt_kernel<1>), grid, dim3(256), 0, st, a, grp);
}

}  // namespace dynmm
