// Weight gradient of the horizontal three-tap convolutions (1x3, stride 1, 'same' padding: resnet.py:104-117) and, one vertical
// tap per workgroup, of the 3x3 convolutions (resnet.py:66-84, model.py:343-357) on the fp32 matrix cores in the Winograd pair
// form, operand tiles by direct global -> LDS loads ("v6").  The 3x1 convolutions take conv_wgrad_wino_vt.hip, the stride-2
// three-tap ones conv_wgrad_s2.hip (the same pipeline, direct form).  (Rounds 3-4 also
// carried the direct form — three contractions per pixel, horizontal and vertical — behind DYNMM_WGRAD_WINO=0; removed in round 5.)
//
// dW[co][tap][ci] = sum_pix dY[co][pix] * X[ci][pix + tap shift]: M = co, N = (tap, ci), reduction over pixels.
// What changed against conv_wgrad_v4_kernel (conv_igemm.hip), and why (docs/DESIGN_history_r1-r4.md §4 "round 3"):
//   * ONE workgroup owns all three taps of its 64 input channels: tile (64 or 128) co x (3 taps x 64 ci), a wave holds
//     (32 or 64) co x (3 x 32 ci) = 3 or 6 accumulator blocks.  The dY tile is staged once for three taps (v4: once per
//     128-wide k-tile = per tap) and, for the horizontal taps, so is the X tile: the three taps are the same LDS rows read
//     one pixel apart — as REGISTER choices (a lane reads 16 consecutive pixels, tap -1 / 0 / +1 use elements 3..10 /
//     4..11 / 5..12), so every LDS access stays a 16-byte aligned ds_read_b128.  L2 -> LDS bytes per FLOP: 1/3.7 (1x3)
//     and 1/2.4 (3x1) of v4's.
//   * tiles arrive by `global_load_lds_dwordx4` (no staging registers, no ds_write, no address arithmetic on the data path)
//     into a ring of NST stages, requested NST-1 steps ahead and counted by hand (`s_waitcnt vmcnt`); one barrier per
//     16-pixel step (48 or 24 MFMAs per wave).  Rows are 5 (7 with the halo) quads long, one of them padding the loader
//     masks off: 80- / 112-byte strides keep ds_read_b128 conflict-free (5r, 7r mod 16 are permutations).
//   * padding is applied where it is cheapest: a vertical tap outside the image, or a pixel past the end of the tensor,
//     reads an all-zero LDS quad instead (address select); the two row ends of a horizontal tap are four register selects
//     per step (W % 4 == 0: a row can only start / end on a quad boundary); loads are never predicated on the data's
//     validity — an address that would leave the tensor is replaced by a mapped one whose value is never used.
// Output: k-major slabs [split][co][tap*Ci + ci] like v4 (summed and permuted by reduce_slabs_perm_kernel), bias
// gradient slabs from the dY tile.  Bit-reproducible: fixed pixel ranges, fixed order.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "conv_igemm.h"

namespace dynmm {

template <int I>
using ic = std::integral_constant<int, I>;

// WINO (horizontal taps): the three taps of a pixel PAIR come from FOUR contractions instead of six — the weight-gradient
// form of the 1-D Winograd algorithm the input-gradient kernels use (conv_wino.hip): with e0, e1 the pair's dY values and
// d0..d3 the four X values under them,
//     m1 = e0 (d0 - d2)    m2 = (e0 + e1)(d1 + d2) / 2    m3 = (e0 - e1)(d2 - d1) / 2    m4 = e1 (d1 - d3)
//     dW[tap -1] += m1 + m2 + m3      dW[tap 0] += m2 - m3      dW[tap +1] += m2 + m3 - m4
// summed over the pairs: 16 instead of 24 MFMAs per wave, step and 32-row block (2/3 of the matrix work), same loads, same
// slabs (the output transform runs on the accumulators, the halvings are exact), error vs fp64 of the class of the direct
// fp32 sum.  Both operand transforms are one add per MFMA operand in registers.
// K33 (round 5): a 3x3 filter (BasicBlock, resnet.py:66-84; the decoder's conv3x3, model.py:343-357) is three 1x3 filters on
// input rows shifted by -1 / 0 / +1: one workgroup owns ONE vertical tap of its 64 input channels (k-tiles = 3 x Ci / 64), stages
// the X rows `dr` image rows away (a quad whose row leaves the image reads the zero quad instead — the vertical padding) and
// writes taps 3 (dr + 1) .. 3 (dr + 1) + 2 of the slab.  Same Winograd pairs, same loads, same slabs as the 1x3 launch.
#ifndef DYNMM_V6_NST
#define DYNMM_V6_NST 2      // two-slot operand ring since round 6 (three before): the step 0.14 ms faster on 15 of 20 alternating
                           // runs, 17 / 25 KB of LDS less per workgroup (profiles/r06_ab_runs.md); the DMA of step s + 2 is issued right
                           // behind the barrier of step s and has one step (32 - 48 MFMAs per wave) to land
#endif
template <int MCO, int NST, int OCC, bool K33 = false>
__global__ void __launch_bounds__(256, OCC) conv_wgrad_v6_kernel(const WgradArgs a_in, const WgradGroup grp) {
    constexpr int NACC = 4;
    WgradArgs a = a_in;
    constexpr int TCO = 64 * MCO, BP = 16;
    constexpr int LDG = 20, LDX = 28;                           // row strides in floats
    constexpr int XROWS = 64;
    constexpr int G_STAGE = TCO * LDG, X_STAGE = XROWS * LDX;   // floats per ring slot
    constexpr int GW = TCO / 4;                                 // dY rows requested by one wave
    constexpr int NJG = (GW + 11) / 12;                         // wave instructions per stage: dY (12 rows each)
    constexpr int RJX = 9;                                      //   X rows per instruction
    constexpr int NJX1 = (16 + RJX - 1) / RJX;                  //   X (16 rows per wave and tap)
    constexpr int J = NJG + NJX1;                               // loads in flight per wave and stage
    static_assert(NST == 2 || NST == 3, "ring depth");
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
    const int ktile = tile / a.n_co_tiles;
    const int ci0 = (K33 ? ktile % (a.Ci / 64) : ktile) * 64;
    const int dr = K33 ? ktile / (a.Ci / 64) - 1 : 0;          // 3x3: this workgroup's vertical tap reads rows h + dr
    const int split = lin / n_tiles;
    const int HW = a.H * a.W, W = a.W, H = a.H, M = a.M;

    const int total_steps = (M + BP - 1) / BP;
    const int step_begin = split * a.steps_per_split;
    const int step_end = min(total_steps, step_begin + a.steps_per_split);
    const int nsteps = step_end - step_begin;

    if (t < 4) Zs[t] = 0.f;

    // ---------------------------------------------------------------- loader state (one quad per lane and instruction)
    // dY and the vertical-tap X rows: lane -> (row r5 of the instruction, quad q5; q5 == 4 is the padding quad)
    const int q5 = lane % 5, r5 = lane / 5;
    int l_m = step_begin * BP + 4 * (q5 < 4 ? q5 : 3);          // first pixel of the quad
    int l_rem;
    unsigned l_goff;
    {
        const int n = l_m / HW;
        l_rem = l_m - n * HW;
        l_goff = ((unsigned)(n * a.Co + co0 + wave * GW + r5) * (unsigned)HW + (unsigned)l_rem) * 4u;
    }
    // Co is not a multiple of the 64-row tile (the 40-class conv_out, model.py:286): a dY row past Co is requested from row
    // Co - 1 instead (every instruction is still issued: the hand-counted vmcnt relies on it) and read as the zero quad
    unsigned g_adj[NJG];
#pragma unroll
    for (int i = 0; i < NJG; ++i) {
        const int over = co0 + wave * GW + r5 + 12 * i - (a.Co - 1);
        g_adj[i] = over > 0 ? (unsigned)(over * HW) * 4u : 0u;
    }
    // horizontal-tap X rows: lane -> (row r7, quad q7 of 6: pixels [p0 - 4, p0 + 20); q7 == 6 is the padding quad)
    const int q7 = lane % 7, r7 = lane / 7;
    int h_m = 0, h_rem = 0;
    unsigned h_xoff = 0;
    h_m = step_begin * BP + 4 * ((q7 < 6 ? q7 : 5) - 1);
    const int n = h_m < 0 ? -1 : h_m / HW;
    h_rem = h_m - n * HW;
    h_xoff = (unsigned)(((n * a.Ci + ci0 + wave * 16 + r7) * HW + h_rem + dr * W) * 4);
    const unsigned lds_g = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)Gs);
    const unsigned lds_x = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)Xs);

    auto issue = [&](int slot) __attribute__((always_inline)) {
        {
            const unsigned v = l_m < M ? l_goff : 0u;
            const unsigned dst = lds_g + (unsigned)((slot * G_STAGE + wave * GW * LDG) * 4);
#pragma unroll
            for (int i = 0; i < NJG; ++i) {
                constexpr int dummy = 0;
                (void)dummy;
                const int rows = GW - 12 * i < 12 ? GW - 12 * i : 12;
                if (q5 < 4 && r5 < rows) dma16(a.dy + (size_t)(12 * i) * HW, l_m < M ? v - g_adj[i] : 0u, dst + (unsigned)(12 * i * LDG * 4));
            }
        }
        // (3x3: a quad whose shifted row leaves the image is never used — any mapped address)
        const unsigned v = (h_m >= 0 && h_m < M && (!K33 || (unsigned)(h_rem + dr * W) < (unsigned)HW)) ? h_xoff : 0u;
        const unsigned dst = lds_x + (unsigned)((slot * X_STAGE + wave * 16 * LDX) * 4);
#pragma unroll
        for (int i = 0; i < NJX1; ++i) {
            const int rows = 16 - 9 * i < 9 ? 16 - 9 * i : 9;
            if (q7 < 6 && r7 < rows) dma16(a.x + (size_t)(9 * i) * HW, v, dst + (unsigned)(9 * i * LDX * 4));
        }
    
        // advance the quad by one step
        l_m += BP; l_rem += BP; l_goff += BP * 4;
        if (l_rem >= HW) {
            l_rem -= HW;
            l_goff += (unsigned)((a.Co - 1) * HW) * 4u;
        }
        h_m += BP; h_rem += BP; h_xoff += BP * 4;
        if (h_rem >= HW) { h_rem -= HW; h_xoff += (unsigned)((a.Ci - 1) * HW) * 4u; }
    
    };

    // ---------------------------------------------------------------- reader state
    // lane (l31, khalf): rows l31 of its wave's blocks, pixels [8*khalf, 8*khalf + 8) of the step = quads j = 0, 1
    int r_m = step_begin * BP + 8 * khalf;
    int r_ow, r_oh;
    {
        const int n = r_m / HW, rem = r_m - n * HW;
        r_oh = rem / W;
        r_ow = rem - r_oh * W;
    }
    const int rd_g = (wave_co * 32 * MCO + l31) * LDG + 8 * khalf;
    const int rd_x = (wave_k * 32 + l31) * LDX + 8 * khalf;

    f32x16 acc[MCO][NACC];
#pragma unroll
    for (int mi = 0; mi < MCO; ++mi)
#pragma unroll
        for (int s = 0; s < NACC; ++s)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[mi][s][j] = 0.f;

    const bool do_bias = a.out_bias != nullptr && (tile / a.n_co_tiles) == 0;
    float bsum = 0.f;

    // ---------------------------------------------------------------- fragments: two register sets
    // set S holds the operands of one step: dY [mi][8 pixels]; X: 16 row elements + the 4 row-end variants (horizontal
    // taps) or [tap][8 pixels] (vertical taps).  The set of step s + 1 is read from LDS under the MFMAs of step s.
    float av[2][MCO][8];
    float bx[2][20];
    int r_step = step_begin;
    auto read_frags = [&](auto SET, int slot) __attribute__((always_inline)) {
        constexpr int S = decltype(SET)::value;
        const float* gs = Gs + slot * G_STAGE;
        const float* xs = Xs + slot * X_STAGE;
        // second quad of the lane: 4 pixels on, possibly in the next row
        int ow1 = r_ow + 4, oh1 = r_oh;
        if (ow1 >= W) { ow1 -= W; ++oh1; }
        if (oh1 >= H) oh1 -= H;
        const bool in0 = r_m < M, in1 = r_m + 4 < M;
#pragma unroll
        for (int mi = 0; mi < MCO; ++mi) {
            const bool rv = co0 + wave_co * 32 * MCO + mi * 32 + l31 < a.Co;          // (a row past Co: zeros)
            const float* p0 = (in0 && rv) ? gs + rd_g + mi * 32 * LDG : Zs;
            const float* p1 = (in1 && rv) ? gs + rd_g + mi * 32 * LDG + 4 : Zs;
            const float4 u0 = *reinterpret_cast<const float4*>(p0);
            const float4 u1 = *reinterpret_cast<const float4*>(p1);
            av[S][mi][0] = u0.x; av[S][mi][1] = u0.y; av[S][mi][2] = u0.z; av[S][mi][3] = u0.w;
            av[S][mi][4] = u1.x; av[S][mi][5] = u1.y; av[S][mi][6] = u1.z; av[S][mi][7] = u1.w;
        }
        // bx[i] = pixel 8*khalf + i - 4 of the step (row position 8*khalf + i), i < 16
        // (3x3: quads 0, 1 lie in the first quad's image row wherever their values are used, quads 2, 3 in the second's)
        bool okA = true, okB = true;
        if constexpr (K33) {
            okA = (unsigned)(r_oh + dr) < (unsigned)H;
            okB = (unsigned)(oh1 + dr) < (unsigned)H;
        }
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const float* px = (qd < 2 ? okA : okB) ? xs + rd_x + 4 * qd : Zs;
            const float4 u = *reinterpret_cast<const float4*>(px);
            bx[S][4 * qd] = u.x; bx[S][4 * qd + 1] = u.y; bx[S][4 * qd + 2] = u.z; bx[S][4 * qd + 3] = u.w;
        }
        // left neighbours of the quads' first pixels / right neighbours of their last pixels: zero at the row ends
        bx[S][16] = r_ow == 0 ? 0.f : bx[S][3];
        bx[S][17] = ow1 == 0 ? 0.f : bx[S][7];
        bx[S][18] = r_ow == W - 4 ? 0.f : bx[S][8];
        bx[S][19] = ow1 == W - 4 ? 0.f : bx[S][12];
    
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
        if (r_ow >= W) { r_ow -= W; ++r_oh; }
        if (r_oh >= H) r_oh -= H;
    };
    auto mfmas = [&](auto SET) __attribute__((always_inline)) {
        constexpr int S = decltype(SET)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) {           // the lane's four pixel pairs: pixels (2j, 2j + 1) = bx[4 + 2j], bx[5 + 2j]
            const float d0 = j == 0 ? bx[S][16] : (j == 2 ? bx[S][17] : bx[S][3 + 2 * j]);
            const float d1 = bx[S][4 + 2 * j], d2 = bx[S][5 + 2 * j];
            const float d3 = j == 1 ? bx[S][18] : (j == 3 ? bx[S][19] : bx[S][6 + 2 * j]);
            const float v0 = d0 - d2, v1 = d1 + d2, v2 = d2 - d1, v3 = d1 - d3;
#pragma unroll
            for (int mi = 0; mi < MCO; ++mi) {
                const float e0 = av[S][mi][2 * j], e1 = av[S][mi][2 * j + 1];
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
        read_frags(ic<0>{}, 0);
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
            read_frags(ic<1 - S>{}, next);
        }
        mfmas(SET);
        slot = next;
    };
    for (int s = 0; s < nsteps; s += 2) {
        step(ic<0>{}, s);
        if (s + 1 < nsteps) step(ic<1>{}, s + 1);
    }

    if (do_bias && t < TCO && co0 + t < a.Co) a.out_bias[(size_t)split * a.Co + co0 + t] = bsum;
    const int KHKW = K33 ? 9 : 3;
    float* out = a.out + (size_t)split * a.Co * a.K;
    const int ci = ci0 + wave_k * 32 + l31;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int tap = K33 ? 3 * (dr + 1) + s : s;
        const size_t col = a.k_major_out ? (size_t)tap * a.Ci + ci : (size_t)ci * KHKW + tap;
        const size_t rowlen = a.k_major_out ? (size_t)a.K : (size_t)a.Ci * KHKW;
#pragma unroll
        for (int mi = 0; mi < MCO; ++mi)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int co = co0 + wave_co * 32 * MCO + mi * 32 + (j & 3) + 8 * (j >> 2) + 4 * khalf;
                if (co >= a.Co) continue;
                // output transform (m2, m3 were accumulated without their 1/2)
                const float hs = 0.5f * (acc[mi][1][j] + acc[mi][2][j]);
                const float v = s == 0 ? acc[mi][0][j] + hs : (s == 1 ? 0.5f * (acc[mi][1][j] - acc[mi][2][j]) : hs - acc[mi][3][j]);
                out[(size_t)co * rowlen + col] = v;
            }
    }
}

// (conv_wgrad_s2.hip) the stride-2 three-tap convolutions of the first block of a stage, direct form on the same pipeline
bool wgrad_s2_shape_ok(const dynmm_conv_geom* g);
void launch_wgrad_s2(const WgradArgs& a, const WgradGroup& grp, dim3 grid, hipStream_t st);

// geometry only (pointer alignment is the launcher's business)
bool wgrad_v6_shape_ok(const dynmm_conv_geom* g) {
    if (wgrad_s2_shape_ok(g)) return true;
    const bool h_taps = g->KH == 1 && g->KW == 3 && g->PH == 0 && g->PW == 1;
    const bool v_taps = g->KH == 3 && g->KW == 1 && g->PH == 1 && g->PW == 0;
    const bool k33 = g->KH == 3 && g->KW == 3 && g->PH == 1 && g->PW == 1 && g->H >= 2;
    if (!h_taps && !v_taps && !k33) return false;
    if (g->SH != 1 || g->SW != 1 || g->H != g->Ho || g->W != g->Wo || g->c_split != g->Ci) return false;
    if (g->W % 4 != 0 || g->W < 16 || g->Ci % 64 != 0) return false;
    // rows: a multiple of the 64-row tile, or — horizontal taps and 3x3 only — any multiple of 4 from 24 up (conv_out's 40 classes)
    if (g->Co % 64 != 0 && (v_taps || g->Co % 4 != 0 || g->Co < 24)) return false;
    // 32-bit byte offsets inside one tensor, signed pixel counters
    const unsigned long long cmax = (unsigned long long)(g->Ci > g->Co ? g->Ci : g->Co);
    if ((unsigned long long)g->N * cmax * g->H * g->W * 4ull >= (1ull << 31)) return false;
    return true;
}

int wgrad_v6_tco(const dynmm_conv_geom* g) { return g->Co % 128 == 0 ? 128 : 64; }

// workgroups per CU the launcher compiles the kernel for; the plan sizes one residency round with it.  Vertical taps
// (conv_wgrad_wino_vt.hip): 2.  Horizontal taps: 128-row tiles hold 128 accumulators (2 waves per SIMD), 64-row tiles half of that (3).
int wgrad_v6_occupancy(const dynmm_conv_geom* g) {
    if (g->KH == 3 && g->KW == 1) return 2;         // (stride 1: the Winograd pair positions; stride 2: three X row sets per stage)
    return g->Co % 128 == 0 ? 2 : 3;
}

void launch_wgrad_wino_vt(const WgradArgs& a, const WgradGroup& grp, dim3 grid, hipStream_t st);

// Every eligible geometry runs the Winograd form since round 4; the direct instantiations (three contractions per pixel, the
// DYNMM_WGRAD_WINO=0 / _V6_OCC / _V6_MIN_CO / _NO_V6 switches that kept them reachable) were removed in round 5.
void launch_wgrad_v6(const WgradArgs& a, const WgradGroup& grp, dim3 grid, int occ, hipStream_t st) {
    (void)occ;
    const bool two = a.Co % 128 == 0;
    if (a.SH == 2 || a.SW == 2) {                   // stride-2 three-tap convolutions: direct form (conv_wgrad_s2.hip)
        launch_wgrad_s2(a, grp, grid, st);
    } else if (a.KH == 3 && a.KW == 1) {                   // vertical taps: pair positions (conv_wgrad_wino_vt.hip)
        launch_wgrad_wino_vt(a, grp, grid, st);
    } else if (a.KH == 3 && a.KW == 3) {            // one vertical tap per workgroup, horizontal Winograd pairs
        if (two) hipLaunchKernelGGL((conv_wgrad_v6_kernel<2, DYNMM_V6_NST, 2, true>), grid, dim3(256), 0, st, a, grp);
        else hipLaunchKernelGGL((conv_wgrad_v6_kernel<1, DYNMM_V6_NST, 3, true>), grid, dim3(256), 0, st, a, grp);
    } else {                                        // horizontal taps
        if (two) hipLaunchKernelGGL((conv_wgrad_v6_kernel<2, DYNMM_V6_NST, 2, false>), grid, dim3(256), 0, st, a, grp);
        else hipLaunchKernelGGL((conv_wgrad_v6_kernel<1, DYNMM_V6_NST, 3, false>), grid, dim3(256), 0, st, a, grp);
    }
}

}  // namespace dynmm
