// Three-tap convolutions by the 1-D Winograd minimal filtering algorithm F(2,3) on the fp32 matrix cores (round 4).
//
// The factorised 3x1 / 1x3 convolutions of every NonBottleneck1D block (FusionDynMM/src/models/resnet.py:124-147) are 81 % of
// the path's MACs (the decoder's 3x3 convolutions, 12 %, run on the 2-D form in conv_wino2d.hip).  Along
// the tap axis two neighbouring outputs (y0, y1) of a 3-tap filter g over inputs d0..d3 are
//     m1 = (d0 - d2) g0            m2 = (d1 + d2) (g0 + g1 + g2) / 2
//     m4 = (d1 - d3) g2            m3 = (d2 - d1) (g0 - g1 + g2) / 2            y0 = m1 + m2 + m3,   y1 = m2 - m3 - m4
// i.e. FOUR channel contractions per output PAIR instead of six: 2/3 of the direct convolution's matrix-core work, in
// plain fp32 arithmetic (measured against fp64, scratch/r4/wino_numerics.py: rms error 2.0e-7 .. 3.8e-7 for C = 64 .. 512
// against 1.3e-7 .. 1.6e-7 for the direct fp32 sum).  The four contractions are four GEMMs
//     M_i[co][pair] = sum_ci U_i[co][ci] * V_i[ci][pair]
// with the filter transforms U_i precomputed per step (wino_pack_kernel) and the data transforms V_i formed in registers
// from the raw input tile at the fragment read (one add per MFMA operand); the output transform is lane-local because a
// lane of the 32x32 accumulator layout holds one pair's 16 channels for all four M_i.
//
// Used for the forward (training and inference) and the INPUT GRADIENT of those convolutions (the input gradient of a stride-1
// three-tap convolution is the three-tap convolution of dy with the flipped filter).
//
// Kernel structure: tile 64 co x 64 pairs; a wave owns 32 output channels x 32 pairs x 4 transforms = 4 accumulator blocks (64
// registers; 3 workgroups per CU for the vertical kernels, 4 for the horizontal three-tap ones: 37.6 KB of LDS each); operands go
// global -> LDS by `global_load_lds_dwordx4` into a 3-slot ring requested a stage ahead (hand-counted vmcnt), one barrier per
// 8-channel stage = 16 MFMAs per wave, fragments of k-pair q+1 are read under the 4 MFMAs of k-pair q.  (Round 4 also carried
// 128 x 64 / 64 x 128 tiles with 8 blocks per wave: slower at batch 32, removed in round 5.  What bounds the kernel:
// profiles/r05_wino_bound.md.)
//   * filter operand [tap row][ci][co][4 transforms]: a lane's four A values of a k-pair are ONE ds_read_b128;
//   * horizontal taps (1x3): the raw tile is [8 channels][2*pairs + 8] pixels (16-byte aligned quads, a 4-pixel halo
//     either side); a lane reads (., d0) (d1, d2) (d3, .) as three conflict-free ds_read_b64;
//   * vertical taps (3x1): the raw tile is [8 channels][4 input rows][pairs]: the four rows an output row pair needs
//     (2x the output bytes through L2, where one gather per tap moves 3x);
//   * zero padding: a value outside the image is replaced by 0 with a lane-constant select after the read (the load itself
//     is never predicated: an out-of-image row is replaced by a mapped one).
#include <stdlib.h>

#include "common.h"
#include "conv_igemm.h"

namespace dynmm {

struct WinoArgs {
    const float* x;         // input [N, Ci, H, W]  (input gradient: dy, Ci = the convolution's Co)
    const float* ut;        // transformed filters [Ci][CoS][4]
    const float* shift;     // [Co] or nullptr (forward: bias)
    const float* residual;  // like y or nullptr.  forward: added before the activation; input gradient: added after the mask
    const float* mask;      // like y or nullptr (input gradient): y = mask > 0 ? y : 0
    float* y;               // [N, Co, H, W]
    double* stats;          // STATS: [nslots][2][Co] running sums of y and y^2 over (N, H, W) (BatchNorm batch statistics), or nullptr
    const float *bn_mean, *bn_invstd, *bn_gamma, *bn_beta;   // BNRED: the BatchNorm whose output gradient this launch produces
    const unsigned long long* bits;   // BNRED == 2: that BatchNorm's ReLU decisions, one bit per element (norm.hip mask_word layout)
    int nslots;             //        pixel tile p adds into slab p % nslots (4800 tiles on one address cost a C = 64 launch 18 %)
    int N, Ci, Co, H, W;
    int CoS;                // row stride of `ut` (Co rounded up to the 64-row tile: the pack writes zero rows)
    int act;
    int MP;                 // output pairs
    int H2;                 // vertical taps: row pairs per image, (H + 1) / 2
    int n_co_tiles, n_p_tiles;
    int Hin, Win;           // input image (== H, W except for the stride-2 input gradients below)
};

__device__ __forceinline__ float quad_sum(float v) {            // sum over the lane's quad, in every lane of it (DPP quad_perm)
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));     // lanes ^ 1
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));     // lanes ^ 2
    return v;
}

// S2 (input gradient only): the same pair machinery for the STRIDE-2 three-tap convolutions that open stages 2-4 (resnet.py:
// 104-107 with stride (2,1) / (1,2)).  Along the strided axis an output pair of dx is fed by two neighbouring dy values e0, e1:
//     dx[2j] = W1^T e0          dx[2j+1] = W2^T e0 + W0^T e1
// — three contractions per pair, none of them on a zero (the polyphase form; not a Winograd saving, the direct count).  A pair
// = one dy position, the raw tile holds dy itself (horizontal: [8 ch][pairs + 8], vertical: 2 rows), the filter operand
// carries (W1, W2, W0, 0), M1 and M2 share an accumulator.  It replaces the round-2 tile kernel's parity-class enumeration
// (4-byte stores at stride 8, 71 TFLOP/s) with the pair kernel's 8-byte stores and operand ring.
// TAIL (forward only): Co is not a multiple of the 64-row tile (the 40-class conv_out, model.py:286): the filter operand is packed
// with zero rows up to the tile, the epilogue skips the channels past Co.
// STATS (forward, horizontal taps, small tile): the convolution feeds a training-mode BatchNorm (resnet.py:110,118; model_utils.py:
// 22): per-channel sums of y and y^2 of the tile are formed in the epilogue (lane quads by DPP, the rest through LDS in a fixed
// order) and added to a.stats with one fp64 atomic per channel and statistic — what bn_stats_kernel would produce with a launch
// and a pass over y of its own (the same fp64 atomics finish its sums).
// BNRED (input gradient, vertical taps, small tile): the launch produces the gradient g of z = relu(BN(c)) (resnet.py:131-135:
// conv3x1_2 consumes relu(bn1(.))).  `mask` carries c, the BatchNorm's INPUT: the epilogue re-derives [BN(c) > 0] with the
// forward's own fma, and leaves the BatchNorm backward's two reductions, sum g.[z > 0] and sum g.[z > 0].xhat, in a.stats
// (the `sums` of dynmm_bn_bwd_apply) — bn_bwd_reduce_kernel's launch and its pass over g and c are not needed.
// BNRED == 2 (round 5): the same for a BatchNorm + identity + ReLU (bn2 of a block, resnet.py:136-147) whose output feeds the
// NEXT block and nothing else: this launch is that block's first convolution (3x1), `residual` the gradient of its identity
// branch, so the epilogue holds the complete gradient g of out = relu(bn2(c) + identity).  The ReLU decision is not derivable
// from c: `bits` carries the one-bit-per-element record the forward's normalise pass left (norm.hip).  The launch writes
// g.[out > 0] (what bn_bwd_apply and the identity branch both consume: no second masked copy) and leaves the two reductions.
// (3x3 filters ran here in round 4 — horizontal taps in the pair form, the vertical taps looped as part of the reduction, 2/3 of
// the direct work; since round 5 they run on the 2-D form F(2x2,3x3) of conv_wino2d.hip at 4/9, and this kernel carries none
// of the tap bookkeeping: no zero slot in LDS, no row bits, a loader whose addresses advance by constant strides.)
template <int TCO, int MCO, bool VERT, bool DGRAD, bool S2 = false, bool TAIL = false, bool STATS = false, int BNRED = 0>
__global__ void __launch_bounds__(256, MCO == 1 ? ((VERT && !DGRAD) ? 4 : 3) : 2) conv_wino_kernel(const WinoArgs a) {
    static_assert(!BNRED || (DGRAD && VERT && !S2 && MCO == 1 && TCO == 64 && !TAIL && !STATS), "BatchNorm reductions: vertical dgrad");
    static_assert(!S2 || DGRAD, "the stride-2 form is an input gradient");
    static_assert(!STATS || (!DGRAD && !VERT && MCO == 1 && TCO == 64 && !TAIL), "statistics: the forward's small horizontal tile");
    static_assert(!TAIL || (!DGRAD && MCO == 1), "channel tails exist in the forward's small tile only");
    // MCO: 32-channel blocks per wave.  2: a wave owns 64 co x 32 pairs x 4 transforms (128 accumulator registers, two
    // workgroups per CU); 1: 32 co x 32 pairs x 4 (64 registers, three workgroups per CU: smaller tiles for the grids a
    // 8192-accumulator tile quantises badly, and a third neighbour to cover a workgroup's prologue / epilogue)
    // Ring depth.  The vertical kernels stage 4 input rows per pair (8 KB per stage beside the filters' 8 KB): three slots are
    // 48 KB = three workgroups per CU.  The vertical FORWARD (115 VGPRs) runs on a two-slot ring — 32 KB, FOUR workgroups per CU,
    // the same single barrier per stage, the DMA of stage s + 2 issued right behind that barrier and given one stage (instead
    // of two) to land: round 6, scratch/r6/wino_time.py in steady state (the first shape of a run is 10 % slow whatever runs
    // it): 97.9 -> 94.7 us at C = 128 (+ residual 102.7 -> 99.0), equal at C = 64 / 256 / 512; the step is the same within
    // 0.1 ms.  A small gain, kept because it also frees 16 KB of LDS per workgroup.  The vertical input gradients need 142 - 144
    // VGPRs (mask / accum / BatchNorm-reduction epilogues): at a 128-register budget they spill 17 - 26 and run slower; with
    // single-buffered epilogue operands (120 - 122 VGPRs, no spill) they are equal to the three-workgroup form within the
    // measurement's own order bias — they stay at three workgroups and three slots.  The horizontal kernels fit four with three slots.
    constexpr int S = (VERT && !DGRAD && MCO == 1) ? 2 : 3;
    constexpr int BK = 8;
    constexpr int WCO = 32 * MCO;
    constexpr int WAVES_CO = TCO / WCO, WAVES_P = 4 / WAVES_CO, TP = 32 * WAVES_P;
    static_assert(WAVES_CO * WAVES_P == 4 && (MCO == 1 || MCO == 2), "4 waves per workgroup");
    constexpr int A_STAGE = BK * TCO * 4;                                   // floats
    constexpr int PIXW = (S2 ? TP : 2 * TP) + 8;                            // horizontal: pixels per staged row
    constexpr int NP = S2 ? 2 : 4;                                          // vertical: input rows per pair
    constexpr int B_STAGE = VERT ? BK * NP * TP : BK * PIXW;
    constexpr int QPR = VERT ? TP / 4 : PIXW / 4;                           // quads per (channel[, input row])
    constexpr int QB = B_STAGE / 4, QPW = QB / 4;                           // quads per stage / per wave
    constexpr int NIB = (QPW + 63) / 64;
    constexpr int IPR = TCO / 64;                                           // instructions per filter row (1 KB each)
    constexpr int NIA = 2 * IPR;                                            // a wave loads 2 of the 8 rows
    constexpr int NI = NIA + NIB;
    static_assert(NI < 64 && QB % 4 == 0, "vmcnt is a 6-bit counter; the four waves split a stage evenly");

    __shared__ __attribute__((aligned(16))) float As[S * A_STAGE];
    __shared__ __attribute__((aligned(16))) float Bs[S * B_STAGE];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wave_co = wave / WAVES_P, wave_p = wave % WAVES_P;
    const int khalf = lane >> 5, l31 = lane & 31;

    const int nblk = a.n_co_tiles * a.n_p_tiles;
    const int lin = xcd_remap((int)blockIdx.x, nblk);
    const int co0 = (lin % a.n_co_tiles) * TCO;
    const int p0 = (lin / a.n_co_tiles) * TP;
    const int HW = a.H * a.W;
    const int HWin = a.Hin * a.Win;
    const int NC = a.Ci / BK;
    const int nst = NC;

    // ---------------------------------------------------------------- loader state
    unsigned b_off[NIB];
    bool b_act[NIB];
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
        const int ql = i * 64 + lane;
        b_act[i] = ql < QPW;
        const int q = wave * QPW + (b_act[i] ? ql : 0);
        if constexpr (VERT) {
            const int k = q / (NP * QPR), j = (q / QPR) % NP, gq = q % QPR;
            int p = p0 + 4 * gq;
            p = p > a.MP - 4 ? a.MP - 4 : p;       // quads past the tensor: any mapped address (never used)
            const int per = a.H2 * a.W;
            const int n = p / per, rr = p - n * per;
            const int r2 = rr / a.W, w = rr - r2 * a.W;
            const int row = S2 ? r2 + j : 2 * r2 - 1 + j;
            const int rowc = (row >= 0 && row < a.Hin) ? row : (S2 ? r2 : 2 * r2);
            b_off[i] = ((unsigned)(n * a.Ci + k) * (unsigned)HWin + (unsigned)(rowc * a.Win + w)) * 4u;
        } else {
            const int k = q / QPR, quad = q - k * QPR;
            const int M = S2 ? a.MP : 2 * a.MP;    // input pixels
            int m = (S2 ? p0 : 2 * p0) - 4 + 4 * quad;
            m = m < 0 ? 0 : (m > M - 4 ? M - 4 : m);
            const int n = m / HWin, rem = m - n * HWin;
            b_off[i] = ((unsigned)(n * a.Ci + k) * (unsigned)HWin + (unsigned)rem) * 4u;
        }
    }
    const unsigned a_voff = (unsigned)lane * 16u;
    const unsigned lds_a = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)As);
    const unsigned lds_b = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)Bs);
    // Next stage to request.  Everything the request needs advances by a constant per stage (the filter operand's row
    // 8 * stage + 2 * wave of [tap row][ci] x CoS, the input's channel chunk, the ring slot): running pointers, no
    // multiplications in the loop — the first version recomputed them from the stage index, ~70 scalar instructions per wave and
    // stage beside 16 MFMAs (profiles/r05_wino_bound.md).
    const float* a_ptr = a.ut + ((size_t)(2 * wave) * a.CoS + co0) * 4;
    const size_t a_step = (size_t)BK * a.CoS * 4, a_row = (size_t)a.CoS * 4;
    const float* b_ptr = a.x;
    const size_t b_step = (size_t)BK * HWin;
    unsigned l_adst = lds_a + (unsigned)(2 * wave * TCO * 4 * 4), l_bdst = lds_b + (unsigned)(wave * QPW * 4 * 4);
    const unsigned l_adst_end = l_adst + (unsigned)(S * A_STAGE * 4);
    int l_left = nst;                             // stages not yet requested
    auto issue = [&]() {
        if (l_left > 0) {
#pragma unroll
            for (int i = 0; i < NIA; ++i)
                dma16(a_ptr + (size_t)(i / IPR) * a_row + 64 * (i % IPR) * 4, a_voff, l_adst + (unsigned)i * 1024u);
#pragma unroll
            for (int i = 0; i < NIB; ++i) {
                if (b_act[i]) dma16(b_ptr, b_off[i], l_bdst + (unsigned)i * 1024u);
            }
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
    const int lp = wave_p * 32 + l31;             // pair of this lane inside the tile
    const int p = p0 + lp;
    const bool pvalid = p < a.MP;
    int pn, prem;                                 // image and pixel offset (inside the image) of the pair's first output
    bool m0, m2, m3;                              // d0 / d2 / d3 lie inside the image
    {
        const int pc = pvalid ? p : 0;
        if constexpr (VERT) {
            const int per = a.H2 * a.W;
            pn = pc / per;
            const int rr = pc - pn * per;
            const int r2 = rr / a.W, w = rr - r2 * a.W;
            prem = 2 * r2 * a.W + w;
            m0 = r2 > 0;
            m2 = 2 * r2 + 1 < a.H;
            m3 = S2 ? r2 + 1 < a.Hin : 2 * r2 + 2 < a.H;          // (S2: e1, the next dy row, exists)
        } else {
            const int m = 2 * pc;
            pn = m / HW;
            prem = m - pn * HW;
            const int h = prem / a.W, w = prem - h * a.W;
            m0 = w > 0;
            m2 = true;
            m3 = S2 ? (w >> 1) + 1 < a.Win : w + 2 < a.W;         // (S2: e1, the next dy column, exists)
        }
    }
    const int a_frag = (khalf * TCO + wave_co * WCO + l31) * 4;                          // + (2q * TCO + mi * 32) * 4
    const int b_frag = VERT ? khalf * NP * TP + lp : (S2 ? khalf * PIXW + 4 + lp : khalf * PIXW + 2 * lp + 2);   // + 2q * (NP TP | PIXW)

    f32x16 acc[4][MCO];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int mi = 0; mi < MCO; ++mi)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][mi][j] = 0.f;

    float4 fa[2][MCO];                            // [register set][mi]: U_0..U_3 of (k, co)
    float fd[2][4];                               // [register set]: raw d0..d3 of (k, pair)
    float fv[2][4];                               // [register set][transform]: V_i of (k, pair)
    // The fragment traffic of k-pair q + 1 is split in two so that no LDS wait sits between the MFMAs of k-pair q (measured
    // with the compiler's own interleaving: MFMA-busy 0.47): read_raw issues the LDS reads BEFORE the 8 MFMAs of k-pair q,
    // transform consumes them AFTER those MFMAs have been issued (its handful of VALU instructions runs while the last MFMA
    // executes); scheduling barriers pin the three phases.
    auto read_raw = [&](int set, int q, const float* Ap, const float* Bp) {
#pragma unroll
        for (int mi = 0; mi < MCO; ++mi)
            fa[set][mi] = *reinterpret_cast<const float4*>(Ap + a_frag + (2 * q * TCO + mi * 32) * 4);
        if constexpr (S2) {
            const float* b = Bp + b_frag + 2 * q * (VERT ? NP * TP : PIXW);
            fd[set][0] = b[0];
            fd[set][1] = b[VERT ? TP : 1];
            fd[set][2] = fd[set][3] = 0.f;
        } else if constexpr (VERT) {
            const float* b = Bp + b_frag + 2 * q * 4 * TP;
            fd[set][0] = b[0];
            fd[set][1] = b[TP];
            fd[set][2] = b[2 * TP];
            fd[set][3] = b[3 * TP];
        } else {
            const float* b = Bp + b_frag + 2 * q * PIXW;
            const float2 u1 = *reinterpret_cast<const float2*>(b + 2);
            fd[set][0] = b[1];
            fd[set][1] = u1.x;
            fd[set][2] = u1.y;
            fd[set][3] = b[4];
        }
    };
    auto transform = [&](int set) {
        if constexpr (S2) {                           // V = (e0, e0, e1): no arithmetic, the zero past the last row / column
            fv[set][0] = fv[set][1] = fd[set][0];
            fv[set][2] = m3 ? fd[set][1] : 0.f;
            fv[set][3] = 0.f;
            return;
        }
        const float d0 = m0 ? fd[set][0] : 0.f, d1 = fd[set][1];
        const float d2 = (VERT && !m2) ? 0.f : fd[set][2], d3 = m3 ? fd[set][3] : 0.f;
        fv[set][0] = d0 - d2;
        fv[set][1] = d1 + d2;
        fv[set][2] = d2 - d1;
        fv[set][3] = d1 - d3;
    };
    auto mfma_set = [&](int set) {
#pragma unroll
        for (int mi = 0; mi < MCO; ++mi) {
            const float av[4] = {fa[set][mi].x, fa[set][mi].y, fa[set][mi].z, fa[set][mi].w};
            if constexpr (S2) {                       // M0 -> y0; M1 and M2 -> y1 (one accumulator)
                acc[0][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], fv[set][0], acc[0][mi], 0, 0, 0);
                acc[1][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], fv[set][1], acc[1][mi], 0, 0, 0);
                acc[1][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], fv[set][2], acc[1][mi], 0, 0, 0);
                continue;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                acc[i][mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], fv[set][i], acc[i][mi], 0, 0, 0);
        }
    };
#define DYNMM_WINO_PHASE() __builtin_amdgcn_sched_barrier(0)

    // ---------------------------------------------------------------- K loop
    // Stage s = 4 k-pairs; the fragments of k-pair q + 1 are read under the 8 MFMAs of k-pair q — across the stage boundary
    // too: after the reads of the stage's last k-pair the wave waits for stage s + 1 (stage s + 2 may still be in flight),
    // passes the barrier (every wave has the last fragments of stage s in registers: its slot is free), requests stage s + 3
    // into that slot and reads the first fragments of stage s + 1, all under the last 8 MFMAs of stage s.
    static_assert(BK == 8, "four k-pairs per stage");
#pragma unroll
    for (int i = 0; i < S; ++i) issue();
    wait_vm<(S - 1) * NI>();                      // (nst >= 3: the launcher requires >= 24 reduction channels)
    __syncthreads();
    int c_a = 0, c_b = 0;                         // ring offsets (floats) of the stage being consumed
    const float* Ap = As;
    const float* Bp = Bs;
    read_raw(0, 0, Ap, Bp);
    transform(0);
    for (int s = 0; s < nst; ++s) {
        DYNMM_WINO_PHASE();
        read_raw(1, 1, Ap, Bp);
        DYNMM_WINO_PHASE();
        mfma_set(0);
        DYNMM_WINO_PHASE();
        transform(1);
        DYNMM_WINO_PHASE();
        read_raw(0, 2, Ap, Bp);
        DYNMM_WINO_PHASE();
        mfma_set(1);
        DYNMM_WINO_PHASE();
        transform(0);
        DYNMM_WINO_PHASE();
        read_raw(1, 3, Ap, Bp);
        DYNMM_WINO_PHASE();
        mfma_set(0);
        DYNMM_WINO_PHASE();
        transform(1);
        DYNMM_WINO_PHASE();
        if (s + 1 < nst) {
            if (S > 2 && s + 2 < nst) wait_vm<(S - 2) * NI>();
            else wait_vm<0>();
            __syncthreads();
            issue();
            c_a += A_STAGE;
            c_b += B_STAGE;
            if (c_a == S * A_STAGE) { c_a = 0; c_b = 0; }
            Ap = As + c_a;
            Bp = Bs + c_b;
            read_raw(0, 0, Ap, Bp);
        }
        DYNMM_WINO_PHASE();
        mfma_set(1);
        DYNMM_WINO_PHASE();
        transform(0);                             // (after the last stage: stale registers, no consumer)
    }
#undef DYNMM_WINO_PHASE
    // ---------------------------------------------------------------- epilogue
    // output transform (lane-local), bias / residual / activation (forward) or ReLU mask / accumulated gradient (input
    // gradient), NCHW stores: horizontal pairs as 8-byte stores (256-byte runs per half wave), vertical pairs as two rows.
    // The epilogue operands of batch b+1 (8 channels x 2 outputs of one accumulator block half) are requested before batch b
    // is transformed and stored, the first batch before the rings are released: with two workgroups per CU nothing else
    // hides their latency (a dgrad launch with a ReLU mask ran at 95 TFLOP/s against 121 without, before this).
    const float* __restrict__ res_p = a.residual;
    const float* __restrict__ mask_p = a.mask;
    float* __restrict__ y_p = a.y;
    const bool has_res = res_p != nullptr, has_mask = mask_p != nullptr;
    const int act = a.act;
    const unsigned row_bytes = (unsigned)HW * 4u;
    const unsigned second = VERT ? (unsigned)a.W * 4u : 4u;              // byte distance of the pair's second output
    const bool y1_ok = VERT ? m2 : true;
    const unsigned off_base = ((unsigned)(pn * a.Co + co0 + wave_co * WCO + 4 * khalf) * (unsigned)HW + (unsigned)prem) * 4u;
    auto off_of = [&](int b, int e) {            // batch b = 2 * mi + h, element e: channel mi * 32 + (e & 3) + 8 * (2 h + (e >> 2))
        return off_base + (unsigned)((b >> 1) * 32 + (e & 3) + 8 * (2 * (b & 1) + (e >> 2))) * row_bytes;
    };
    const int co_lim = a.Co - (co0 + wave_co * WCO + 4 * khalf);          // TAIL: channels of this lane below this are real
    auto live = [&](int b, int e) { return !TAIL || (b >> 1) * 32 + (e & 3) + 8 * (2 * (b & 1) + (e >> 2)) < co_lim; };
    // (BNRED == 2 carries three operands per output — c, the identity gradient, the decision word: one register set, loaded at
    // the top of its batch; with two the kernel spills at 3 workgroups per CU, whose other waves cover the latency instead)
    constexpr int NSET = BNRED == 2 ? 1 : 2;
    float k0[NSET][8], k1[NSET][8], r0[NSET][8], r1[NSET][8];
    unsigned w0[NSET][8], w1[NSET][8];            // BNRED == 2: the 32-bit halves that hold the decisions
    // bit of element i of a plane: word (i >> 8) * 4 + (i & 3) of the plane's groups, bit (i & 255) >> 2 (norm.hip)
    const unsigned bit_groups = (unsigned)((HW + 255) >> 8);
    auto bit_byte = [&](int i) { return (unsigned)(((i >> 8) * 4 + (i & 3)) * 8 + ((((i & 255) >> 2) >> 5) * 4)); };
    const unsigned bit_off0 = bit_byte(prem), bit_off1 = bit_byte(prem + a.W);
    const unsigned bit_sh0 = (unsigned)(((prem & 255) >> 2) & 31), bit_sh1 = (unsigned)((((prem + a.W) & 255) >> 2) & 31);
    const unsigned bit_plane0 = (unsigned)(pn * a.Co + co0 + wave_co * WCO + 4 * khalf) * bit_groups * 32u;
#pragma unroll
    for (int e = 0; e < 8; ++e)                   // (launches without epilogue operands never load: neutral values)
#pragma unroll
        for (int q = 0; q < NSET; ++q) {
            r0[q][e] = r1[q][e] = 0.f;
            k0[q][e] = k1[q][e] = 1.f;
        }
    auto load_batch = [&](int set, int b) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const unsigned off = off_of(b, e);
            r0[set][e] = r1[set][e] = 0.f;
            k0[set][e] = k1[set][e] = 1.f;
            if constexpr (BNRED == 2) w0[set][e] = w1[set][e] = 0u;
            if (!pvalid || !live(b, e)) continue;
            if constexpr (BNRED == 2) {
                const char* wp_ = reinterpret_cast<const char*>(a.bits) + bit_plane0 +
                                  (unsigned)((b >> 1) * 32 + (e & 3) + 8 * (2 * (b & 1) + (e >> 2))) * bit_groups * 32u;
                w0[set][e] = *reinterpret_cast<const unsigned*>(wp_ + bit_off0);
                w1[set][e] = y1_ok ? *reinterpret_cast<const unsigned*>(wp_ + bit_off1) : 0u;
            }
            if constexpr (VERT) {
                if (has_mask) {
                    k0[set][e] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(mask_p) + off);
                    if (y1_ok) k1[set][e] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(mask_p) + off + second);
                }
                if (has_res) {
                    r0[set][e] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(res_p) + off);
                    if (y1_ok) r1[set][e] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(res_p) + off + second);
                }
            } else {
                if (has_mask) {
                    const float2 kk = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(mask_p) + off);
                    k0[set][e] = kk.x;
                    k1[set][e] = kk.y;
                }
                if (has_res) {
                    const float2 rr = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(res_p) + off);
                    r0[set][e] = rr.x;
                    r1[set][e] = rr.y;
                }
            }
        }
    };
    if (has_mask || has_res) load_batch(0, 0);
    __syncthreads();                               // every wave is done with the rings: As is reused below
    float* const sh_lds = As;
    for (int i = t; i < TCO; i += 256) sh_lds[i] = (a.shift && (!TAIL || co0 + i < a.Co)) ? a.shift[co0 + i] : 0.f;
    __syncthreads();
    float* const cst_lds = As + TCO;              // BNRED: per channel {gamma.invstd, beta - mean.gamma.invstd, invstd, -mean.invstd}
    if constexpr (BNRED) {
        if (t < TCO) {
            const int c = co0 + t;
            const float is = a.bn_invstd[c], mu = a.bn_mean[c], sc = a.bn_gamma[c] * is;
            cst_lds[4 * t + 0] = sc;
            cst_lds[4 * t + 1] = fmaf(-mu, sc, a.bn_beta[c]);      // (bn_apply_kernel's own two lines)
            cst_lds[4 * t + 2] = is;
            cst_lds[4 * t + 3] = -mu * is;
        }
        __syncthreads();
    }
    if constexpr (!STATS && !BNRED) {
        if (!pvalid) return;
    }
    float* const st_lds = As + 5 * TCO;           // STATS / BNRED: [batch 2][wave 4][row 32 = (khalf, e, statistic)][lane quad 8]
#pragma unroll
    for (int b = 0; b < 2 * MCO; ++b) {
        const int mi = b >> 1, h = b & 1, set = NSET == 2 ? (b & 1) : 0;
        if constexpr (NSET == 2) {
            if (b + 1 < 2 * MCO && (has_mask || has_res)) load_batch(set ^ 1, b + 1);
        } else {
            if (b > 0) load_batch(0, b);
        }
        float v0[8], v1[8];
        float q1[8], q2[8];                       // BNRED: this lane's contributions to the two reductions
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int j = 8 * h + e;
            const float ma = acc[0][mi][j], mb = acc[1][mi][j], mc = acc[2][mi][j], md = acc[3][mi][j];
            const int cl = wave_co * WCO + mi * 32 + 4 * khalf + (e & 3) + 8 * (2 * h + (e >> 2));
            const float sh = sh_lds[cl];
            float y0 = S2 ? ma + sh : (ma + mb) + mc + sh;
            float y1 = S2 ? mb + sh : (mb - mc) - md + sh;
            if constexpr (BNRED != 0) {
                const float4 cs = *reinterpret_cast<const float4*>(cst_lds + 4 * cl);
                const float c0 = k0[set][e], c1 = k1[set][e];
                if constexpr (BNRED == 2) {
                    y0 = (pvalid && ((w0[set][e] >> bit_sh0) & 1u)) ? y0 + r0[set][e] : 0.f;
                    y1 = (pvalid && y1_ok && ((w1[set][e] >> bit_sh1) & 1u)) ? y1 + r1[set][e] : 0.f;
                } else {
                    y0 = (pvalid && fmaf(c0, cs.x, cs.y) > 0.f) ? y0 : 0.f;
                    y1 = (pvalid && y1_ok && fmaf(c1, cs.x, cs.y) > 0.f) ? y1 : 0.f;
                }
                q1[e] = y0 + y1;
                q2[e] = fmaf(y0, fmaf(c0, cs.z, cs.w), y1 * fmaf(c1, cs.z, cs.w));
            } else if (DGRAD) {
                if (has_mask) {
                    y0 = k0[set][e] > 0.f ? y0 : 0.f;
                    y1 = k1[set][e] > 0.f ? y1 : 0.f;
                }
                y0 += r0[set][e];
                y1 += r1[set][e];
            } else {
                y0 += r0[set][e];
                y1 += r1[set][e];
                if (act == DYNMM_ACT_RELU) {
                    y0 = y0 > 0.f ? y0 : 0.f;
                    y1 = y1 > 0.f ? y1 : 0.f;
                } else if (act == DYNMM_ACT_TANH) {
                    y0 = tanhf(y0);
                    y1 = tanhf(y1);
                }
            }
            v0[e] = y0;
            v1[e] = y1;
        }
        if constexpr (STATS || BNRED) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float s1 = quad_sum(BNRED ? q1[e] : (pvalid ? v0[e] + v1[e] : 0.f));
                const float s2 = quad_sum(BNRED ? q2[e] : (pvalid ? fmaf(v0[e], v0[e], v1[e] * v1[e]) : 0.f));
                if ((l31 & 3) == 0) {
                    float* dst = st_lds + ((b * 4 + wave) * 32 + (khalf * 8 + e) * 2) * 8 + (l31 >> 2);
                    dst[0] = s1;
                    dst[8] = s2;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const unsigned off = off_of(b, e);
            if (!live(b, e) || ((STATS || BNRED) && !pvalid)) continue;
            if constexpr (VERT) {
                *reinterpret_cast<float*>(reinterpret_cast<char*>(y_p) + off) = v0[e];
                if (y1_ok) *reinterpret_cast<float*>(reinterpret_cast<char*>(y_p) + off + second) = v1[e];
            } else {
                *reinterpret_cast<float2*>(reinterpret_cast<char*>(y_p) + off) = make_float2(v0[e], v1[e]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (STATS || BNRED) {
        __syncthreads();
        // thread t: batch t >> 7, (channel, statistic) row (t >> 1) & 63 = (wave_co, khalf, e, statistic), pixel half t & 1
        const int sb = t >> 7, r = (t >> 1) & 63, wp = t & 1;
        const int wco = r >> 5, row = r & 31;
        const float* src = st_lds + ((sb * 4 + wco * WAVES_P + wp) * 32 + row) * 8;
        float acc_s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc_s += src[j];
        acc_s += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(acc_s), 0xB1, 0xF, 0xF, true));   // + the other pixel half
        if (wp == 0) {
            const int kh = row >> 4, e = (row >> 1) & 7, stat = row & 1;
            const int c = co0 + wco * WCO + 4 * kh + (e & 3) + 8 * (2 * sb + (e >> 2));
            atomicAdd(a.stats + ((size_t)((lin / a.n_co_tiles) % a.nslots) * 2 + stat) * a.Co + c, (double)acc_s);
        }
    }
}

// Filter transforms.  w [Co][Ci][3 taps] -> ut [K][C][4] with (K, C) = (Ci, Co) for the forward operand and (Co, Ci)
// for the input gradient's (whose taps run the other way along the Winograd axis: g0 <-> g2).
__global__ void __launch_bounds__(256) wino_pack_kernel(const float* __restrict__ w, float4* __restrict__ ut,
                                                        const float* __restrict__ scale, int Co, int Ci, int dgrad) {
    const int K = dgrad ? Co : Ci, Cr = dgrad ? Ci : Co, Cc = (Cr + 63) & ~63;      // rows padded to the 64-row tile (zeros)
    const size_t total = (size_t)K * Cc;
    const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= total) return;
    const int c = (int)(o % Cc);
    const int k = (int)(o / Cc);
    if (c >= Cr) {
        ut[o] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const int co = dgrad ? k : c, ci = dgrad ? c : k;
    const float* g = w + ((size_t)co * Ci + ci) * 3;
    float g0 = g[0], g1 = g[1], g2 = g[2];
    if (dgrad == 2) {                             // stride-2 input gradient (polyphase): (W1, W2, W0, 0), no transform
        ut[o] = make_float4(g1, g2, g0, 0.f);
        return;
    }
    if (dgrad) { const float tmp = g0; g0 = g2; g2 = tmp; }
    if (scale) {                                  // inference: an eval-mode BatchNorm's per-channel factor folded into the filter
        const float sc = scale[co];
        g0 *= sc; g1 *= sc; g2 *= sc;
    }
    ut[o] = make_float4(g0, (g0 + g1 + g2) * 0.5f, (g0 - g1 + g2) * 0.5f, g2);
}

// Many filters in one launch (ops.PackedWeights: once per training step).  desc[d] = {src, dst: float offsets from the
// two bases; Co | Ci << 32; KH | KW << 8 | dgrad << 16 | first workgroup << 32} (KH, KW: 3x1 or 1x3 — three taps either way).
struct WinoPackDesc {
    long long src, dst;
    int Co, Ci, kk, blk0;
};

__global__ void __launch_bounds__(256) wino_pack_multi_kernel(const float* __restrict__ src_base, float* __restrict__ dst_base,
                                                              const WinoPackDesc* __restrict__ desc, int ndesc) {
    int lo = 0, hi = ndesc - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (desc[mid].blk0 <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const WinoPackDesc d = desc[lo];
    const int dgrad = (d.kk >> 16) & 3;
    const int K = dgrad ? d.Co : d.Ci, Cr = dgrad ? d.Ci : d.Co, Cc = (Cr + 63) & ~63;
    const size_t total = (size_t)K * Cc;
    const size_t o = (size_t)((int)blockIdx.x - d.blk0) * 256 + threadIdx.x;
    if (o >= total) return;
    const int c = (int)(o % Cc);
    const int k = (int)(o / Cc);
    if (c >= Cr) {
        reinterpret_cast<float4*>(dst_base + d.dst)[o] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const int co = dgrad ? k : c, ci = dgrad ? c : k;
    const float* g = src_base + d.src + ((size_t)co * d.Ci + ci) * 3;
    float g0 = g[0], g1 = g[1], g2 = g[2];
    if (dgrad == 2) {
        reinterpret_cast<float4*>(dst_base + d.dst)[o] = make_float4(g1, g2, g0, 0.f);
        return;
    }
    if (dgrad) { const float tmp = g0; g0 = g2; g2 = tmp; }
    reinterpret_cast<float4*>(dst_base + d.dst)[o] = make_float4(g0, (g0 + g1 + g2) * 0.5f, (g0 - g1 + g2) * 0.5f, g2);
}

// rows = output channels of the GEMM (a multiple of the 64-row tile), red = its reduction channels (8 per stage, >= 3 stages):
// forward (Co, Ci), input gradient (Ci, Co) — e.g. the input gradient of the 40-class conv_out (model.py:295-308) qualifies; the
// FORWARD also takes row counts that are not multiples of the tile (>= 24, % 8 == 0: conv_out's 40 — TAIL instantiations)
static bool wino_geom_ok(const dynmm_conv_geom* g, bool dgrad) {
    if (!g || g->c_split != g->Ci) return false;
    if (g->SH != 1 || g->SW != 1) return false;
    const bool k13 = g->KH == 1 && g->KW == 3, k31 = g->KH == 3 && g->KW == 1;      // (3x3: conv_wino2d.hip)
    if (!(k13 || k31)) return false;
    if (g->PH != g->KH / 2 || g->PW != g->KW / 2 || g->H != g->Ho || g->W != g->Wo) return false;
    if (g->W % 4 != 0 || g->W < 4 || g->H < 2) return false;
    const int rows = dgrad ? g->Ci : g->Co, red = dgrad ? g->Co : g->Ci;
    if ((dgrad ? rows % 64 != 0 : (rows % 8 != 0 || rows < 24)) || red % 8 != 0 || red < 24) return false;
    if ((long long)g->N * g->H * g->W < 256) return false;
    if ((double)g->N * (g->Ci > g->Co ? g->Ci : g->Co) * g->H * g->W >= 1073741824.0) return false;   // 32-bit byte offsets
    return true;
}

// the stride-2 three-tap convolutions whose INPUT gradient takes the polyphase pair form (S2): 3x1 stride (2,1) pad (1,0) or
// 1x3 stride (1,2) pad (0,1) on even extents
static bool wino_s2_geom_ok(const dynmm_conv_geom* g) {
    if (!g || g->c_split != g->Ci) return false;
    const bool v = g->KH == 3 && g->KW == 1 && g->SH == 2 && g->SW == 1 && g->PH == 1 && g->PW == 0;
    const bool h = g->KH == 1 && g->KW == 3 && g->SH == 1 && g->SW == 2 && g->PH == 0 && g->PW == 1;
    if (!v && !h) return false;
    if (v && (g->H % 2 != 0 || g->Ho * 2 != g->H || g->Wo != g->W)) return false;
    if (h && (g->W % 2 != 0 || g->Wo * 2 != g->W || g->Ho != g->H)) return false;
    if (g->W % 4 != 0 || g->Wo % 4 != 0 || g->Wo < 4 || g->Ho < 1) return false;
    if (g->Ci % 64 != 0 || g->Co % 8 != 0 || g->Co < 24) return false;
    if ((long long)g->N * g->Ho * g->Wo < 256) return false;
    if ((double)g->N * (g->Ci > g->Co ? g->Ci : g->Co) * g->H * g->W >= 1073741824.0) return false;
    return true;
}

static int launch_wino(WinoArgs& a, bool vert, bool dgrad, hipStream_t st, bool s2 = false) {
    if (s2) {                                      // pairs = dy positions; a.Hin / a.Win were set by the caller
        a.H2 = a.Hin;
        a.MP = a.N * a.Hin * a.Win;
        a.n_co_tiles = a.Co / 64;
        a.n_p_tiles = ceil_div(a.MP, 64);
        dim3 grid2((unsigned)(a.n_co_tiles * a.n_p_tiles));
        if (vert) hipLaunchKernelGGL((conv_wino_kernel<64, 1, true, true, true>), grid2, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv_wino_kernel<64, 1, false, true, true>), grid2, dim3(256), 0, st, a);
        DYNMM_LAUNCH_CHECK();
        return DYNMM_OK;
    }
    a.Hin = a.H;
    a.Win = a.W;
    a.H2 = (a.H + 1) / 2;
    a.MP = vert ? a.N * a.H2 * a.W : a.N * a.H * a.W / 2;
    // tile: 64 co x 64 pairs, 4 accumulator blocks per wave, 3 workgroups per CU.  (Round 4 also kept 128 x 64 / 64 x 128 tiles
    // with 8 blocks per wave behind DYNMM_WINO_TILE=1: slower on every encoder shape but one at batch 32 — C = 256 / 512: 143-148
    // / 125-134 against 115-122 / 108-110 TFLOP/s algorithmic — and removed in round 5 with their six instantiations.)
    const bool tail = a.Co % 64 != 0;
    a.n_co_tiles = ceil_div(a.Co, 64);
    a.n_p_tiles = ceil_div(a.MP, 64);
    dim3 grid((unsigned)(a.n_co_tiles * a.n_p_tiles));
#define DYNMM_WINO_LAUNCH(...) hipLaunchKernelGGL((conv_wino_kernel<64, 1, __VA_ARGS__>), grid, dim3(256), 0, st, a)
    if (a.stats && dgrad) {                       // BatchNorm backward reductions from the vertical input gradient
        if (a.bits) DYNMM_WINO_LAUNCH(true, true, false, false, false, 2);
        else DYNMM_WINO_LAUNCH(true, true, false, false, false, 1);
    } else if (a.stats) {                         // (the entry point admitted only what these instantiations serve)
        DYNMM_WINO_LAUNCH(false, false, false, false, true);
    } else if (tail) {
        if (vert) DYNMM_WINO_LAUNCH(true, false, false, true);
        else DYNMM_WINO_LAUNCH(false, false, false, true);
    } else if (vert) {
        if (dgrad) DYNMM_WINO_LAUNCH(true, true);
        else DYNMM_WINO_LAUNCH(true, false);
    } else {
        if (dgrad) DYNMM_WINO_LAUNCH(false, true);
        else DYNMM_WINO_LAUNCH(false, false);
    }
#undef DYNMM_WINO_LAUNCH
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

}  // namespace dynmm

using namespace dynmm;

extern "C" int dynmm_conv2d_wino_supported(const dynmm_conv_geom* g, int dgrad) {
    if (wino_geom_ok(g, dgrad != 0)) return 1;
    return (dgrad && wino_s2_geom_ok(g)) ? 2 : 0;
}

extern "C" size_t dynmm_wino_packed_floats(int Co, int Ci, int KH, int KW) {
    if (Co <= 0 || Ci <= 0 || KH <= 0 || KW <= 0) return 0;
    if (!((KH == 1 && KW == 3) || (KH == 3 && KW == 1))) return 0;
    // either operand: [K][rows rounded up to 64][4]
    const size_t fwd = (size_t)Ci * ((Co + 63) & ~63), dg = (size_t)Co * ((Ci + 63) & ~63);
    return (fwd > dg ? fwd : dg) * 4;
}

extern "C" int dynmm_wino_pack(const float* w, float* ut, const float* scale, int Co, int Ci, int KH, int KW, int dgrad,
                               void* stream) {
    (void)hipGetLastError();
    if (!w || !ut || Co <= 0 || Ci <= 0 || (scale && dgrad) || dgrad < 0 || dgrad > 2) return DYNMM_EINVAL;
    if (!((KH == 1 && KW == 3) || (KH == 3 && KW == 1))) return DYNMM_EUNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(ut) & 15u) return DYNMM_EINVAL;
    const size_t total = dynmm_wino_packed_floats(Co, Ci, KH, KW) / 4;
    hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)ceil_div_sz(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       reinterpret_cast<float4*>(ut), scale, Co, Ci, dgrad);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_wino_pack_multi_blocks(int Co, int Ci, int KH, int KW) {
    return (int)ceil_div_sz(dynmm_wino_packed_floats(Co, Ci, KH, KW) / 4, 256);
}

extern "C" int dynmm_wino_pack_multi(const float* src_base, float* dst_base, const void* desc, int ndesc, int total_blocks,
                                     void* stream) {
    (void)hipGetLastError();
    if (!src_base || !dst_base || !desc || ndesc <= 0 || total_blocks <= 0) return DYNMM_EINVAL;
    if (reinterpret_cast<uintptr_t>(dst_base) & 15u) return DYNMM_EINVAL;
    static_assert(sizeof(WinoPackDesc) == 32, "descriptor layout is part of the ABI (4 x int64 words)");
    hipLaunchKernelGGL(wino_pack_multi_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, src_base, dst_base,
                       (const WinoPackDesc*)desc, ndesc);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_conv2d_wino_fwd(const float* x, const float* ut, const float* bias, const float* residual, float* y,
                                     const dynmm_conv_geom* g, int act, void* stream) {
    (void)hipGetLastError();
    if (!x || !ut || !y || !g) return DYNMM_EINVAL;
    if (!wino_geom_ok(g, false)) return DYNMM_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(ut)) & 15u) return DYNMM_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual)) & 7u) return DYNMM_EUNSUPPORTED;
    WinoArgs a{};
    a.x = x; a.ut = ut; a.shift = bias; a.residual = residual; a.mask = nullptr; a.y = y;
    a.N = g->N; a.Ci = g->Ci; a.Co = g->Co; a.H = g->H; a.W = g->W;
    a.CoS = (g->Co + 63) & ~63;
    a.act = act;
    return launch_wino(a, g->KW == 1, false, (hipStream_t)stream);
}

extern "C" int dynmm_conv2d_wino_dgrad_bnred_supported(const dynmm_conv_geom* g) {
    // 3x1 stride-1 input gradient on the pair kernel
    return (wino_geom_ok(g, true) && g->KH == 3 && g->KW == 1 && g->Ci % 64 == 0) ? 1 : 0;
}

extern "C" int dynmm_conv2d_wino_dgrad_bnred_slots(const dynmm_conv_geom* g) {
    // thousands of pixel tiles adding to one address serialise (measured on the forward's statistics: 4800 tiles on one
    // address cost a C = 64 launch 18 %): pixel tile p adds into slab p % slots, <= 600 tiles per address up to 8 slabs
    if (!dynmm_conv2d_wino_dgrad_bnred_supported(g)) return 0;
    const int s = ceil_div(g->N * ((g->H + 1) / 2) * g->W, 64) / 600;
    return s < 1 ? 1 : (s > 8 ? 8 : s);
}

extern "C" int dynmm_conv2d_wino_dgrad_bnred(const float* dy, const float* ut, const float* bn_x, const float* bn_mean,
                                             const float* bn_invstd, const float* bn_gamma, const float* bn_beta,
                                             double* sums, float* dx, const dynmm_conv_geom* g, void* stream) {
    (void)hipGetLastError();
    if (!dy || !ut || !bn_x || !bn_mean || !bn_invstd || !bn_gamma || !bn_beta || !sums || !dx || !g) return DYNMM_EINVAL;
    if (!dynmm_conv2d_wino_dgrad_bnred_supported(g)) return DYNMM_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(ut)) & 15u) return DYNMM_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(bn_x) | reinterpret_cast<uintptr_t>(sums)) & 7u)
        return DYNMM_EUNSUPPORTED;
    WinoArgs a{};
    a.x = dy; a.ut = ut; a.shift = nullptr; a.residual = nullptr; a.mask = bn_x; a.y = dx;
    a.stats = sums; a.nslots = dynmm_conv2d_wino_dgrad_bnred_slots(g);
    a.bn_mean = bn_mean; a.bn_invstd = bn_invstd; a.bn_gamma = bn_gamma; a.bn_beta = bn_beta;
    a.N = g->N; a.Ci = g->Co; a.Co = g->Ci; a.H = g->H; a.W = g->W;
    a.CoS = (a.Co + 63) & ~63;
    a.act = DYNMM_ACT_NONE;
    a.Hin = g->Ho; a.Win = g->Wo;
    return launch_wino(a, true, true, (hipStream_t)stream);
}

extern "C" int dynmm_conv2d_wino_dgrad_bnred2(const float* dy, const float* ut, const float* accum, const float* bn_x,
                                              const unsigned long long* relu_bits, const float* bn_mean, const float* bn_invstd,
                                              double* sums, float* dx, const dynmm_conv_geom* g, void* stream) {
    (void)hipGetLastError();
    if (!dy || !ut || !bn_x || !relu_bits || !bn_mean || !bn_invstd || !sums || !dx || !g) return DYNMM_EINVAL;
    if (!dynmm_conv2d_wino_dgrad_bnred_supported(g) || (g->H * g->W) % 4 != 0) return DYNMM_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(ut)) & 15u) return DYNMM_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(bn_x) | reinterpret_cast<uintptr_t>(accum) |
         reinterpret_cast<uintptr_t>(sums) | reinterpret_cast<uintptr_t>(relu_bits)) & 7u)
        return DYNMM_EUNSUPPORTED;
    WinoArgs a{};
    a.x = dy; a.ut = ut; a.shift = nullptr; a.residual = accum; a.mask = bn_x; a.y = dx;
    a.stats = sums; a.nslots = dynmm_conv2d_wino_dgrad_bnred_slots(g); a.bits = relu_bits;
    a.bn_mean = bn_mean; a.bn_invstd = bn_invstd; a.bn_gamma = bn_invstd; a.bn_beta = bn_invstd;      // (gamma / beta: BNRED == 1 only)
    a.N = g->N; a.Ci = g->Co; a.Co = g->Ci; a.H = g->H; a.W = g->W;
    a.CoS = (a.Co + 63) & ~63;
    a.act = DYNMM_ACT_NONE;
    a.Hin = g->Ho; a.Win = g->Wo;
    return launch_wino(a, true, true, (hipStream_t)stream);
}

extern "C" int dynmm_conv2d_wino_fwd_stats_supported(const dynmm_conv_geom* g) {
    return (wino_geom_ok(g, false) && g->KW == 3 && g->Co % 64 == 0) ? 1 : 0;
}

extern "C" int dynmm_conv2d_wino_fwd_stats_slots(const dynmm_conv_geom* g) {
    if (!dynmm_conv2d_wino_fwd_stats_supported(g)) return 0;
    const int tiles = ceil_div(g->N * g->H * g->W / 2, 64);           // 64-pair tiles: each adds once per channel and statistic
    const int s = tiles / 600;
    return s < 1 ? 1 : (s > 8 ? 8 : s);
}

extern "C" int dynmm_conv2d_wino_fwd_stats(const float* x, const float* ut, const float* bias, float* y, double* stats,
                                           int nslots, const dynmm_conv_geom* g, void* stream) {
    (void)hipGetLastError();
    if (!x || !ut || !y || !stats || !g || nslots < 1 || nslots > 64) return DYNMM_EINVAL;
    if (!dynmm_conv2d_wino_fwd_stats_supported(g)) return DYNMM_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(ut)) & 15u) return DYNMM_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(stats)) & 7u) return DYNMM_EUNSUPPORTED;
    WinoArgs a{};
    a.x = x; a.ut = ut; a.shift = bias; a.residual = nullptr; a.mask = nullptr; a.y = y; a.stats = stats; a.nslots = nslots;
    a.N = g->N; a.Ci = g->Ci; a.Co = g->Co; a.H = g->H; a.W = g->W;
    a.CoS = g->Co;
    a.act = DYNMM_ACT_NONE;
    return launch_wino(a, false, false, (hipStream_t)stream);
}

extern "C" int dynmm_conv2d_wino_dgrad(const float* dy, const float* ut, const float* mask, const float* accum, float* dx,
                                       const dynmm_conv_geom* g, void* stream) {
    (void)hipGetLastError();
    if (!dy || !ut || !dx || !g) return DYNMM_EINVAL;
    const bool s2 = !wino_geom_ok(g, true) && wino_s2_geom_ok(g);
    if (!s2 && !wino_geom_ok(g, true)) return DYNMM_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(ut)) & 15u) return DYNMM_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(mask) | reinterpret_cast<uintptr_t>(accum)) & 7u)
        return DYNMM_EUNSUPPORTED;
    WinoArgs a{};
    a.x = dy; a.ut = ut; a.shift = nullptr; a.residual = accum; a.mask = mask; a.y = dx;
    a.N = g->N; a.Ci = g->Co; a.Co = g->Ci; a.H = g->H; a.W = g->W;         // the roles of the channel counts swap
    a.CoS = (a.Co + 63) & ~63;
    a.act = DYNMM_ACT_NONE;
    a.Hin = g->Ho; a.Win = g->Wo;                                             // (stride 2: dy is half as high / wide as dx)
    return launch_wino(a, g->KW == 1, true, (hipStream_t)stream, s2);
}
