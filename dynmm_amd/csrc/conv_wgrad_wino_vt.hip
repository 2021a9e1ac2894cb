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
// consecutive in the flattened (image, column quad) order — of ONE row pair, and consecutive stages of a strip walk DOWN the
// image: the X rows come in row PAIRS k = (2k - 1, 2k), stage r2 uses pairs r2 and r2 + 1 and only pair r2 + 1 is new (1.0 input
// row per output row; the round-4 kernel walked row pair by row pair, 8 columns per step, and staged all four X rows of every
// step: 2.0).  TWO neighbouring strips are walked alternately — (r2, strip 0), (r2, strip 1), (r2 + 1, strip 0), ... — because a
// 64-byte row piece is half a 128-byte line (see "the stage sequence" below).  Operand rows are 64-byte pieces, unpadded in LDS:
// the loader permutes the SOURCE quads of a row (lane -> quad q ^ ((row >> 2) & 3)), which makes the readers' ds_read_b128 over
// 32 consecutive rows conflict-free without a padding quad (a DMA lane's LDS address is fixed, its global address is not).  One
// barrier per stage = 64 | 32 MFMAs per wave.  Zero padding (row -1, rows >= H, quads past the tensor) is a component-wise
// register select after the read (a select between float4 OBJECTS goes through private memory, which the compiler then parks
// in 12 KB of LDS).
//
// History of the form.  Round 5 measured a first version — separate "halo" items for the first pair of a strip, uniform items,
// TWO items in flight, 80 KB per workgroup — faster in isolation and 0.3 ms SLOWER in the step, and shelved it.  Round 6 (what the
// weight-gradient launches keep in flight costs the backward's dependent chain beside them: the 2-slot rings of
// conv_wgrad_v6.hip / conv_wgrad_s2.hip): ONE stage in flight — every wait is vmcnt(0) —, a stage that starts a strip pair (or
// this workgroup's range) simply requests the three pairs it and its successor need.  Measured (profiles/r06_ab_runs.md):
//   * the 3x1 weight gradients 428 -> 395 us per grouped launch at C = 256 (141 -> 153 TF/s algorithmic), 444 -> 394 at C = 128,
//     576 -> 459 at C = 64 (105 -> 132), 455 -> 426 at C = 512 — the horizontal kernel's rates;
//   * the step -0.21 ms (8 of 10 alternating pairs) for the one-in-flight strips against the pair kernel, and another -0.28 ms
//     (9 of 10) for walking two strips alternately; round 5's two-in-flight form, same day, same box: +0.45 ms;
//   * FETCH_SIZE per launch (raw KiB, `rocprofv3 --pmc FETCH_SIZE`): pair kernel 328 927 (Co % 128 == 0) / 1 150 075 (C = 64);
//     one strip after the other 444 490 / 1 230 932 — MORE: the other half of every 128-byte line belongs to the neighbouring
//     strip, H2 stages away, and was fetched again; two strips alternately **296 550 / 620 141**.  With the x2 correction of
//     16-byte-per-lane streams that is 1 295 MB per launch against ~1 258 MB algorithmic at C = 64 (1.03x; round 5: 1.85x) and
//     652 MB against ~360 MB at Co % 128 == 0 (1.81x).  What is left there is ALIGNMENT, per shape (one convolution per launch:
//     scratch/r6/wgrad_pmc_shapes.sh): 1.03x at W = 160 (640-byte rows: every strip pair is one 128-byte line), 1.50x at W = 80
//     (320-byte rows: odd rows start 64 bytes into a line, their strip pairs straddle two lines whose other halves belong to
//     the neighbouring pairs, H2 x 2 stages away: (1 + 2) / 2), 2.03x at W = 40 (160-byte rows: three of four row alignments
//     straddle), 1.35x at W = 20 (the tensors nearly fit the L2s).  Ruled out by experiment: the bias-gradient work of the
//     ci-tile-0 workgroups (without a bias 1.48 / 2.03), workgroups of a split running in lockstep (started 4 / 8 us apart: 1.49 /
//     1.57) — the horizontal kernel, which streams whole rows, is at 1.06 - 1.14 on every shape with the same tile decode.
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
    // ONE stage ahead.  dY: 2 slots.  X pairs: the stage being read holds two (two requests apart), the stage in flight brings
    // one — or three when it starts a strip pair: ring positions q - 1, q + 1 live and q + 2 .. q + 4 arriving must be distinct
    // modulo the ring size, which 6 is the smallest to satisfy (4 and 5 collide).
    constexpr int NGS = 2, NXS = 6;
    constexpr int JG = G_ROWS / 64, JX = X_ROWS / 64;               // wave instructions per row set (16 rows x 64 bytes each)
    constexpr int J = JG + JX;
    static_assert(J + 2 * JX < 64, "vmcnt is a 6-bit counter");

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
    const int groups = (QC + 3) / 4;                               // 16-column strips
    const int gpairs = (groups + 1) / 2;                           // strips walked two at a time (see the stage sequence)
    const int total_stages = gpairs * H2 * 2;
    const int sb = split * a.steps_per_split;
    const int se = min(total_stages, sb + a.steps_per_split);

    // ---------------------------------------------------------------- the stage sequence (round 6)
    // Stage s = (strip pair g2, row pair r2, half h): strip g = 2 g2 + h at row pair r2; the two strips of a pair are walked
    // ALTERNATELY down the image — (r2, 0), (r2, 1), (r2 + 1, 0), ... — because a 64-byte row piece is half a 128-byte line whose
    // other half belongs to the neighbouring strip: taken a stage apart the line is fetched once (one strip after the other,
    // H2 stages apart, the fabric moved every line twice: FETCH_SIZE 444 MiB per launch against 329 for the pair kernel).
    // X operand of a stage: row pairs k = r2 and r2 + 1 (rows 2k - 1, 2k) of ITS strip; pair r2 is what the same strip's
    // previous stage requested as its pair r2 + 1, TWO requests back in the X ring.  The first stage of a strip pair or of this
    // workgroup's range requests three pairs in this order — its own pair r2, the pair the NEXT stage (the other strip) will need
    // as its older one, its own pair r2 + 1 — so that "older pair = two requests back" holds from the first stage on.
    // One stage is in flight; every wait is vmcnt(0).
    struct Seq { int s, g2, r2, h; };
    auto seq_init = [&](Seq& q) {
        q.s = sb;
        q.g2 = sb / (2 * H2);
        const int rem = sb - q.g2 * 2 * H2;
        q.r2 = rem >> 1;
        q.h = rem & 1;
    };
    auto seq_next = [&](Seq& q) {
        ++q.s;
        if (q.h == 0) { q.h = 1; return; }
        q.h = 0;
        if (++q.r2 == H2) { q.r2 = 0; ++q.g2; }
    };
    auto seq_first = [&](const Seq& q) { return q.s == sb || (q.r2 == 0 && q.h == 0); };

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
    int l_g2 = -1;
    unsigned l_col[2][J];                                          // byte offset of (image, channel plane, column) per half and instruction
    int l_stage = 0, l_xw = 0;                                     // stages requested; X pairs requested (ring position)
    auto issue = [&]() __attribute__((always_inline)) {
        if (lq.s >= se) return;
        if (lq.g2 != l_g2) {                                        // a new strip pair: this lane's column quads in both strips
            l_g2 = lq.g2;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int i = 0; i < J; ++i) {
                    int qc = 4 * (2 * l_g2 + hh) + l_sq[i];
                    qc = qc < QC ? qc : QC - 1;                     // (a quad past the tensor: mapped, never used)
                    const int n = qc / WQ, wq = qc - n * WQ;
                    const int C = i < JG ? a.Co : a.Ci;
                    l_col[hh][i] = ((unsigned)(n * C + l_ch[i]) * (unsigned)HW + (unsigned)(4 * wq)) * 4u;
                }
        }
        const bool h1 = lq.h != 0;
        const unsigned gdst = lds_g + (unsigned)(((l_stage % NGS) * G_SLOT + wave * JG * 16 * ROWF) * 4);
#pragma unroll
        for (int i = 0; i < JG; ++i) {
            int row = 2 * lq.r2 + l_row[i];
            row = row > H - 1 ? H - 1 : row;                        // (outside the image: a mapped row, zeroed at the read)
            dma16(a.dy, (h1 ? l_col[1][i] : l_col[0][i]) + (unsigned)(row * W) * 4u, gdst + (unsigned)(i * 1024));
        }
        auto load_pair = [&](int kx, bool half1) __attribute__((always_inline)) {     // X rows 2 kx - 1, 2 kx of strip 2 g2 + half
            const unsigned xdst = lds_x + (unsigned)(((l_xw % NXS) * X_SLOT + wave * JX * 16 * ROWF) * 4);
#pragma unroll
            for (int i = JG; i < J; ++i) {
                int row = 2 * kx - 1 + l_row[i];
                row = row < 0 ? 0 : (row > H - 1 ? H - 1 : row);
                dma16(a.x, (half1 ? l_col[1][i] : l_col[0][i]) + (unsigned)(row * W) * 4u, xdst + (unsigned)((i - JG) * 1024));
            }
            ++l_xw;
        };
        if (seq_first(lq)) {
            load_pair(lq.r2, h1);                                   // its own older pair
            load_pair(h1 ? lq.r2 + 1 : lq.r2, !h1);                 // the next stage's (other strip's) older pair
        }
        load_pair(lq.r2 + 1, h1);
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
    int r_stage = 0, r_xw = 2;                    // its index in this workgroup's range; ring position of its pair r2 + 1
    auto read_frags = [&](auto SET, int hs) __attribute__((always_inline)) {
        constexpr int S = decltype(SET)::value;
        const float* gs = Gs + (r_stage % NGS) * G_SLOT;
        const float* x1 = Xs + (r_xw % NXS) * X_SLOT;                         // pair r2 + 1
        const float* x0 = Xs + ((r_xw + NXS - 2) % NXS) * X_SLOT;             // pair r2: two requests back
        const int q = 2 * hs + khalf;
        const bool in = 4 * (2 * rq.g2 + rq.h) + q < QC;
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
                if (4 * (2 * rq.g2 + rq.h) + c < QC) {
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
        r_xw += seq_first(rq) ? 3 : 1;
        wait_vm<0>();
        __syncthreads();
        issue();
    };

    // ---------------------------------------------------------------- pipeline
    const int n_stages = se - sb;
    if (n_stages > 0) {
        issue();                                    // stage sb (a first stage: both pairs)
        issue();                                    // stage sb + 1 stays in flight under stage sb
        // wait for stage sb only: what is in flight behind it is stage sb + 1 = J instructions, J + 2 JX when it starts a strip pair
        {
            Seq nx;
            seq_init(nx);
            seq_next(nx);
            if (n_stages < 2) wait_vm<0>();
            else if (nx.r2 == 0 && nx.h == 0) wait_vm<J + 2 * JX>();
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
int wgrad_wino_vt_units(const dynmm_conv_geom* g) { return ((((g->N * (g->W / 4) + 3) / 4) + 1) / 2) * 2 * ((g->H + 1) / 2) * 16; }

void launch_wgrad_wino_vt(const WgradArgs& a, const WgradGroup& grp, dim3 grid, hipStream_t st) {
    if (a.Co % 128 == 0) hipLaunchKernelGGL((conv_wgrad_wino_vt_kernel<2>), grid, dim3(256), 0, st, a, grp);
    else hipLaunchKernelGGL((conv_wgrad_wino_vt_kernel<1>), grid, dim3(256), 0, st, a, grp);
}

}  // namespace dynmm
