// Input gradients of the stride-1 1x3 convolutions by the 1-D Winograd algorithm F(4,3) on the fp32 matrix cores:
// FOUR neighbouring outputs of a three-tap filter from SIX multiplications (direct: twelve; F(2,3), conv_wino.hip: eight), i.e.
// half of the direct convolution's matrix work.  With d0..d5 the six inputs under an output quad and g the (flipped) filter:
//     V = B^T d :  4d0-5d2+d4 | (d4-4d2)+(d3-4d1) | (d4-4d2)-(d3-4d1) | (d4-d2)+2(d3-d1) | (d4-d2)-2(d3-d1) | 4d1-5d3+d5
//     U = G g   :  g0/4 | -(g0+g1+g2)/6 | -(g0-g1+g2)/6 | g0/24+g1/12+g2/6 | g0/24-g1/12+g2/6 | g2
//     y = A^T m :  m0+m1+m2+m3+m4 | (m1-m2)+2(m3-m4) | (m1+m2)+4(m3+m4) | (m1-m2)+8(m3-m4)+m5          (m_i = U_i V_i)
// over the channels: six GEMMs per output QUAD.  The price is arithmetic error: the transforms carry factors up to 8 and 1/24;
// measured against fp64 (scratch/r4/wino43_numerics.py; tests/test_hip_ops.py on the GPU) the result is 1.7e-6 .. 2.8e-6 from
// the truth in max-norm where a direct fp32 sum is 2e-7 .. 3e-7.  That is why this form serves the BACKWARD only — an input
// gradient is compared with its fp64 value at 2e-4 (GTOL) and enters a gradient whose fp32 conditioning noise is 1e-2 (DESIGN.md
// section 1); forward passes (F(2,3), conv_wino.hip) never see it.  Horizontal taps only: the vertical form of round 4 — six
// input rows per four output rows — measured slower than F(2,3) and was removed in round 5; 3x3 filters (round 4: this form with
// the vertical taps looped, 1/2 of the direct work) run on the 2-D F(2x2,3x3) of conv_wino2d.hip (4/9) since round 5.
//
// Structure = conv_wino.hip's small tile: 64 co x 64 quads per workgroup, a wave owns 32 co x 32 quads x 6 transforms = 6
// accumulator blocks (96 registers); operands by direct global -> LDS loads into a ring (8 channels per stage), fragments of
// k-pair q + 1 read under the 6 MFMAs of k-pair q (across stage boundaries too), phases pinned.
// Round 6: THREE workgroups per CU (DYNMM_W43_WG = 3; round 5 ran two: 206 VGPRs, a 3-slot ring of 62 KB).  The ring has two
// slots — the DMA of stage s + 2 is issued right behind the barrier that ends stage s and has one stage to land; one barrier per
// stage as before — 41.5 KB; the epilogue's operands are single-buffered — 142 VGPRs.  A workgroup's epilogue is an HBM burst
// with no matrix work (mask and residual-gradient reads, stores); with two neighbours to cover it instead of one a launch with
// both operands takes 91.6 us instead of 126.3 at C = 128 (plain 78.1 / 81.9), 88.0 / 104.7 at C = 256, 135.7 / 183.6 at C = 64,
// and the step 0.49 ms less (profiles/r06_ab_runs.md).
//   * filter operand: two planes [ci][co][4] (U0..U3) and [ci][co][2] (U4, U5): one ds_read_b128 + one
//     ds_read_b64 per k-pair;
//   * horizontal taps: raw tile [8 channels][4 * quads + 8] pixels (16-byte quads, 4-pixel halo either side), a lane reads
//     d0 | (d1..d4) | d5;
//   * zero padding by lane-constant selects on the raw values (a column outside the image);
//   * epilogue: output transform, ReLU mask of the producer, accumulated residual gradient; horizontal quads are 16-byte
//     stores; the epilogue operands of batch b + 1 (4 channels x 4 outputs) are requested before batch b is stored.
#include <stdlib.h>

#include "common.h"
#include "conv_igemm.h"

namespace dynmm {

struct Wino43Args {
    const float* x;         // input [N, Ci, H, W] (the convolution's dy: Ci = its Co)
    const float* ut4;       // filter transforms U0..U3  [Ci][Co][4]
    const float* ut2;       //                   U4, U5  [Ci][Co][2]
    const float* residual;  // like y or nullptr: added after the mask
    const float* mask;      // like y or nullptr: y = mask > 0 ? y : 0
    float* y;               // [N, Co, H, W]
    int N, Ci, Co, H, W;
    int MQ;                 // output quads
    int n_co_tiles, n_q_tiles;
};

#ifndef DYNMM_W43_WG
#define DYNMM_W43_WG 3
#endif
#ifndef DYNMM_W43_PREFETCH
#define DYNMM_W43_PREFETCH 1
#endif
__global__ void __launch_bounds__(256, DYNMM_W43_WG) conv_wino43_kernel(const Wino43Args a) {
    constexpr int WG = DYNMM_W43_WG;
    constexpr int BK = 8, S = WG == 3 ? 2 : 3, TCO = 64, TQ = 64, NT = 6;
    constexpr int A4_STAGE = BK * TCO * 4, A2_STAGE = BK * TCO * 2;         // floats
    constexpr int PIXW = 4 * TQ + 8;
    constexpr int B_STAGE = BK * PIXW;
    constexpr int QPR = PIXW / 4;
    constexpr int QB = B_STAGE / 4, QPW = QB / 4;
    constexpr int NIB = (QPW + 63) / 64;
    constexpr int NI = 3 + NIB;                                             // per wave and stage: 2 rows of U0..3, 2 rows of U4,5, the tile
    static_assert(QB % 4 == 0 && NI < 32, "tile shape");

    __shared__ __attribute__((aligned(16))) float A4s[S * A4_STAGE];
    __shared__ __attribute__((aligned(16))) float A2s[S * A2_STAGE];
    __shared__ __attribute__((aligned(16))) float Bs[S * B_STAGE];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wave_co = wave >> 1, wave_q = wave & 1;
    const int khalf = lane >> 5, l31 = lane & 31;

    const int nblk = a.n_co_tiles * a.n_q_tiles;
    const int lin = xcd_remap((int)blockIdx.x, nblk);
    const int co0 = (lin % a.n_co_tiles) * TCO;
    const int q0 = (lin / a.n_co_tiles) * TQ;
    const int HW = a.H * a.W;
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
        const int k = q / QPR, quad = q - k * QPR;
        const int M = 4 * a.MQ;
        int m = 4 * q0 - 4 + 4 * quad;
        m = m < 0 ? 0 : (m > M - 4 ? M - 4 : m);
        const int n = m / HW, rem = m - n * HW;
        b_off[i] = ((unsigned)(n * a.Ci + k) * (unsigned)HW + (unsigned)rem) * 4u;
    }
    const unsigned a4_voff = (unsigned)lane * 16u;
    const unsigned a2_voff = (unsigned)((lane >> 5) * a.Co * 8 + (lane & 31) * 16);      // two 512-byte rows per instruction
    const unsigned lds_a4 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)A4s);
    const unsigned lds_a2 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)A2s);
    const unsigned lds_b = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)Bs);
    // running request state (conv_wino.hip's: constant strides per stage, no multiplications in the loop)
    const float* a4_ptr = a.ut4 + ((size_t)(2 * wave) * a.Co + co0) * 4;
    const float* a2_ptr = a.ut2 + ((size_t)(2 * wave) * a.Co + co0) * 2;
    const size_t a4_step = (size_t)BK * a.Co * 4, a2_step = (size_t)BK * a.Co * 2, a4_row = (size_t)a.Co * 4;
    const float* b_ptr = a.x;
    const size_t b_step = (size_t)BK * HW;
    unsigned l_a4dst = lds_a4 + (unsigned)(2 * wave * TCO * 4 * 4), l_a2dst = lds_a2 + (unsigned)(2 * wave * TCO * 2 * 4);
    unsigned l_bdst = lds_b + (unsigned)(wave * QPW * 4 * 4);
    const unsigned l_a4dst_end = l_a4dst + (unsigned)(S * A4_STAGE * 4);
    int l_left = nst;
    auto issue = [&]() {
        if (l_left > 0) {
            dma16(a4_ptr, a4_voff, l_a4dst);
            dma16(a4_ptr + a4_row, a4_voff, l_a4dst + 1024u);
            dma16(a2_ptr, a2_voff, l_a2dst);
#pragma unroll
            for (int i = 0; i < NIB; ++i) {
                if (b_act[i]) dma16(b_ptr, b_off[i], l_bdst + (unsigned)i * 1024u);
            }
            --l_left;
            a4_ptr += a4_step;
            a2_ptr += a2_step;
            b_ptr += b_step;
            l_a4dst += (unsigned)(A4_STAGE * 4);
            l_a2dst += (unsigned)(A2_STAGE * 4);
            l_bdst += (unsigned)(B_STAGE * 4);
            if (l_a4dst == l_a4dst_end) {
                l_a4dst -= (unsigned)(S * A4_STAGE * 4);
                l_a2dst -= (unsigned)(S * A2_STAGE * 4);
                l_bdst -= (unsigned)(S * B_STAGE * 4);
            }
        }
    };

    // ---------------------------------------------------------------- consumer state
    const int lq = wave_q * 32 + l31;             // quad of this lane inside the tile
    const int qg = q0 + lq;
    const bool qvalid = qg < a.MQ;
    int pn, prem;                                 // image and pixel offset of the quad's first output
    bool dv[6];                                   // d_j lies inside the image
    {
        const int pc = qvalid ? qg : 0;
        const int m = 4 * pc;
        pn = m / HW;
        prem = m - pn * HW;
        const int h = prem / a.W, w = prem - h * a.W;
#pragma unroll
        for (int j = 0; j < 6; ++j) dv[j] = true;
        dv[0] = w > 0;
        dv[5] = w + 4 < a.W;
    }
    const int a4_frag = (khalf * TCO + wave_co * 32 + l31) * 4;            // + 2q * TCO * 4
    const int a2_frag = (khalf * TCO + wave_co * 32 + l31) * 2;            // + 2q * TCO * 2
    const int b_frag = khalf * PIXW + 4 * lq + 3;

    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;

#ifndef DYNMM_W43_DPP
#define DYNMM_W43_DPP 0
#endif
    // The halo values d0 / d5 of a quad are element 3 / element 0 of the NEIGHBOURING lanes' quads.  As two ds_read_b32 at a
    // 16-byte lane stride they are 4-way bank conflicts (bank = (a / 4) mod 32: 8 banks for 32 lanes; round 5 counted 0.60 of this
    // kernel's LDS-active cycles as conflict cycles).  DYNMM_W43_DPP = 1 fetches them by a register exchange instead (v_mov_b32
    // dpp wave_shr:1 / wave_shl:1; the first / last lane of a wave's 32 quads, whose neighbour is the other wave's quad or the
    // tile halo, by ONE ds_read_b32 in which every other lane reads lane 0's address): 2 LDS cycles for 16, bit-identical results —
    // and 2 .. 9 % SLOWER at every large shape (round 6, scratch/r6/w43_time.py: 88.4 -> 90.0 us at C = 128, 92.7 -> 97.0 at 256,
    // with mask + accum 124.9 -> 133.9 / 104.6 -> 115.0; the same exchange in the F(2,3) pair kernel: +3 .. +8 %): the LDS is 15 -
    // 25 % busy either way, while the two DPP moves, two selects and their wait states land in the transform phase, the one
    // stretch of a k-pair in which the wave has no MFMA queued.  What the conflicts cost is LDS cycles nobody was waiting for.
    // Left in as a compile-time option, off.
    const bool edge_lo = l31 == 0, edge_hi = l31 == 31;
    const int e_frag = khalf * PIXW + (edge_hi ? 4 * lq + 8 : 4 * (wave_q * 32) + 3);
    float4 fa4[2];
    float2 fa2[2];
    float fd[2][6], fv[2][6];
    auto read_raw = [&](int set, int q, const float* A4p, const float* A2p, const float* Bp) {
        fa4[set] = *reinterpret_cast<const float4*>(A4p + a4_frag + 2 * q * TCO * 4);
        fa2[set] = *reinterpret_cast<const float2*>(A2p + a2_frag + 2 * q * TCO * 2);
        const float* b = Bp + b_frag + 2 * q * PIXW;
        const float4 u = *reinterpret_cast<const float4*>(b + 1);
        fd[set][1] = u.x; fd[set][2] = u.y; fd[set][3] = u.z; fd[set][4] = u.w;
        if constexpr (DYNMM_W43_DPP) {
            fd[set][0] = Bp[e_frag + 2 * q * PIXW];      // the edge lanes' halo value (d0 of lane 0, d5 of lane 31)
        } else {
            fd[set][0] = b[0];
            fd[set][5] = b[5];
        }
    };
    auto transform = [&](int set) {
        float d[6];
        if constexpr (DYNMM_W43_DPP) {
            const float edge = fd[set][0];
            const float lo = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, fd[set][4]), 0x138, 0xf, 0xf, false));   // wave_shr:1
            const float hi = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, fd[set][1]), 0x130, 0xf, 0xf, false));   // wave_shl:1
            fd[set][0] = edge_lo ? edge : lo;
            fd[set][5] = edge_hi ? edge : hi;
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) d[j] = dv[j] ? fd[set][j] : 0.f;
        const float p = fmaf(-4.f, d[2], d[4]), q_ = fmaf(-4.f, d[1], d[3]);
        const float c = d[4] - d[2], e = d[3] - d[1];
        fv[set][0] = fmaf(-5.f, d[2], fmaf(4.f, d[0], d[4]));
        fv[set][1] = p + q_;
        fv[set][2] = p - q_;
        fv[set][3] = fmaf(2.f, e, c);
        fv[set][4] = fmaf(-2.f, e, c);
        fv[set][5] = fmaf(-5.f, d[3], fmaf(4.f, d[1], d[5]));
    };
    auto mfma_set = [&](int set) {
        const float av[6] = {fa4[set].x, fa4[set].y, fa4[set].z, fa4[set].w, fa2[set].x, fa2[set].y};
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], fv[set][i], acc[i], 0, 0, 0);
    };
#define DYNMM_W43_PHASE() __builtin_amdgcn_sched_barrier(0)

    // ---------------------------------------------------------------- epilogue state (defined here: DYNMM_W43_PREFETCH)
    const float* __restrict__ res_p = a.residual;
    const float* __restrict__ mask_p = a.mask;
    float* __restrict__ y_p = a.y;
    const bool has_res = res_p != nullptr, has_mask = mask_p != nullptr;
    const unsigned row_bytes = (unsigned)HW * 4u;
    const unsigned off_base = ((unsigned)(pn * a.Co + co0 + wave_co * 32 + 4 * khalf) * (unsigned)HW + (unsigned)prem) * 4u;
    auto off_of = [&](int b, int e) {            // batch b: channels 8 b + 4 khalf + e, e = 0..3
        return off_base + (unsigned)(8 * b + e) * row_bytes;
    };
    constexpr int NS = WG == 3 ? 1 : 2;           // epilogue operand sets: double-buffered at two workgroups per CU; at three the
                                                  // register budget (168) has room for one and the third workgroup covers the loads
    float kk[NS][4][4], rr[NS][4][4];             // [set][channel e][output j]
#pragma unroll
    for (int z = 0; z < NS; ++z)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                kk[z][e][j] = 1.f;
                rr[z][e][j] = 0.f;
            }
    auto load_batch = [&](int set, int b) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned off = off_of(b, e);
            if (!qvalid) continue;
            if (has_mask) {
                const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(mask_p) + off);
                kk[set][e][0] = v.x; kk[set][e][1] = v.y; kk[set][e][2] = v.z; kk[set][e][3] = v.w;
            }
            if (has_res) {
                const float4 v = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(res_p) + off);
                rr[set][e][0] = v.x; rr[set][e][1] = v.y; rr[set][e][2] = v.z; rr[set][e][3] = v.w;
            }
        }
    };
    // ---------------------------------------------------------------- K loop (conv_wino.hip's)
#pragma unroll
    for (int i = 0; i < S; ++i) issue();
    wait_vm<(S - 1) * NI>();
    __syncthreads();
    int c_a4 = 0, c_a2 = 0, c_b = 0;              // ring offsets (floats) of the stage being consumed
    const float* A4p = A4s;
    const float* A2p = A2s;
    const float* Bp = Bs;
    read_raw(0, 0, A4p, A2p, Bp);
    transform(0);
    for (int s = 0; s < nst; ++s) {
#if DYNMM_W43_PREFETCH
        // the first batch of epilogue operands is requested at the top of the LAST stage (no operand DMA is outstanding any more:
        // the wait in front of this stage was vmcnt(0)), so that it arrives under that stage's 24 MFMAs
        if (s == nst - 1 && (has_mask || has_res)) load_batch(0, 0);
#endif
        DYNMM_W43_PHASE();
        read_raw(1, 1, A4p, A2p, Bp);
        DYNMM_W43_PHASE();
        mfma_set(0);
        DYNMM_W43_PHASE();
        transform(1);
        DYNMM_W43_PHASE();
        read_raw(0, 2, A4p, A2p, Bp);
        DYNMM_W43_PHASE();
        mfma_set(1);
        DYNMM_W43_PHASE();
        transform(0);
        DYNMM_W43_PHASE();
        read_raw(1, 3, A4p, A2p, Bp);
        DYNMM_W43_PHASE();
        mfma_set(0);
        DYNMM_W43_PHASE();
        transform(1);
        DYNMM_W43_PHASE();
        if (s + 1 < nst) {
            if (S > 2 && s + 2 < nst) wait_vm<(S - 2) * NI>();
            else wait_vm<0>();
            __syncthreads();
            issue();
            c_a4 += A4_STAGE;
            c_a2 += A2_STAGE;
            c_b += B_STAGE;
            if (c_a4 == S * A4_STAGE) { c_a4 = 0; c_a2 = 0; c_b = 0; }
            A4p = A4s + c_a4;
            A2p = A2s + c_a2;
            Bp = Bs + c_b;
            read_raw(0, 0, A4p, A2p, Bp);
        }
        DYNMM_W43_PHASE();
        mfma_set(1);
        DYNMM_W43_PHASE();
        transform(0);
    }
#undef DYNMM_W43_PHASE

    // ---------------------------------------------------------------- epilogue
#if !DYNMM_W43_PREFETCH
    if (has_mask || has_res) load_batch(0, 0);
#endif
    if (!qvalid) return;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int set = NS == 2 ? (b & 1) : 0;
        if (NS == 2 && b + 1 < 4 && (has_mask || has_res)) load_batch(set ^ 1, b + 1);
        if (NS == 1 && b > 0 && (has_mask || has_res)) load_batch(0, b);
        float yo[4][4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int j16 = 4 * b + e;
            const float m0 = acc[0][j16], m1 = acc[1][j16], m2 = acc[2][j16], m3 = acc[3][j16], m4 = acc[4][j16], m5 = acc[5][j16];
            const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
            float y[4];
            y[0] = (m0 + s12) + s34;
            y[1] = fmaf(2.f, d34, d12);
            y[2] = fmaf(4.f, s34, s12);
            y[3] = fmaf(8.f, d34, d12) + m5;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = y[j];
                if (has_mask) v = kk[set][e][j] > 0.f ? v : 0.f;
                yo[e][j] = v + rr[set][e][j];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const unsigned off = off_of(b, e);
            *reinterpret_cast<float4*>(reinterpret_cast<char*>(y_p) + off) = make_float4(yo[e][0], yo[e][1], yo[e][2], yo[e][3]);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Filter transforms of the input gradient: w [Co][Ci][1][3] -> ut4 [Co][Ci][4], ut2 [Co][Ci][2] (reduction = the
// convolution's Co, rows = its Ci; the taps run the other way along the Winograd axis).
__device__ __forceinline__ void wino43_u(float g0, float g1, float g2, float4& u4, float2& u2) {
    const float s = g0 + g2;
    u4 = make_float4(g0 * 0.25f, -(s + g1) * (1.f / 6.f), -(s - g1) * (1.f / 6.f),
                     fmaf(g1, 1.f / 12.f, fmaf(g0, 1.f / 24.f, g2 * (1.f / 6.f))));
    u2 = make_float2(fmaf(-g1, 1.f / 12.f, fmaf(g0, 1.f / 24.f, g2 * (1.f / 6.f))), g2);
}

struct Wino43PackDesc {
    long long src, dst;      // float offsets from the two bases (dst: the record's ut4; its ut2 follows the ut4 block)
    int Co, Ci, kk, blk0;    // kk = KH | KW << 8
};

__global__ void __launch_bounds__(256) wino43_pack_multi_kernel(const float* __restrict__ src_base, float* __restrict__ dst_base,
                                                                const Wino43PackDesc* __restrict__ desc, int ndesc) {
    int lo = 0, hi = ndesc - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (desc[mid].blk0 <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const Wino43PackDesc d = desc[lo];
    const size_t total = (size_t)d.Co * d.Ci;
    const size_t o = (size_t)((int)blockIdx.x - d.blk0) * 256 + threadIdx.x;
    if (o >= total) return;
    const float* g = src_base + d.src + o * 3;    // o = co * Ci + ci
    float4 u4;
    float2 u2;
    wino43_u(g[2], g[1], g[0], u4, u2);           // flipped taps
    reinterpret_cast<float4*>(dst_base + d.dst)[o] = u4;
    reinterpret_cast<float2*>(dst_base + d.dst + total * 4)[o] = u2;
}

__global__ void __launch_bounds__(256) wino43_pack_kernel(const float* __restrict__ w, float* __restrict__ ut, int Co, int Ci) {
    const size_t total = (size_t)Co * Ci;
    const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= total) return;
    const float* g = w + o * 3;                   // o = co * Ci + ci
    float4 u4;
    float2 u2;
    wino43_u(g[2], g[1], g[0], u4, u2);
    reinterpret_cast<float4*>(ut)[o] = u4;
    reinterpret_cast<float2*>(ut + total * 4)[o] = u2;
}

static bool wino43_geom_ok(const dynmm_conv_geom* g) {
    if (!g || g->c_split != g->Ci) return false;
    if (g->SH != 1 || g->SW != 1) return false;
    if (!(g->KH == 1 && g->KW == 3)) return false;      // horizontal taps (vertical: measured slower; 3x3: conv_wino2d.hip)
    if (g->PH != g->KH / 2 || g->PW != g->KW / 2 || g->H != g->Ho || g->W != g->Wo) return false;
    if (g->W % 4 != 0 || g->W < 4 || g->H < 2) return false;
    if (g->Ci % 64 != 0 || g->Co % 8 != 0 || g->Co < 24) return false;       // rows = Ci (64-row tile), reduction = Co
    if ((long long)g->N * g->H * g->W < 256) return false;
    if ((double)g->N * (g->Ci > g->Co ? g->Ci : g->Co) * ((g->H + 3) / 4 * 4) * g->W >= 1073741824.0) return false;
    return true;
}

}  // namespace dynmm

using namespace dynmm;

extern "C" int dynmm_conv2d_wino43_supported(const dynmm_conv_geom* g) { return wino43_geom_ok(g) ? 1 : 0; }

extern "C" size_t dynmm_wino43_packed_floats(int Co, int Ci, int KH, int KW) {
    if (Co <= 0 || Ci <= 0 || KH != 1 || KW != 3) return 0;
    return (size_t)Co * Ci * 6;
}

extern "C" int dynmm_wino43_pack(const float* w, float* ut, int Co, int Ci, int KH, int KW, void* stream) {
    (void)hipGetLastError();
    if (!w || !ut || Co <= 0 || Ci <= 0) return DYNMM_EINVAL;
    if (!(KH == 1 && KW == 3)) return DYNMM_EUNSUPPORTED;
    if (reinterpret_cast<uintptr_t>(ut) & 15u) return DYNMM_EINVAL;
    const size_t total = dynmm_wino43_packed_floats(Co, Ci, KH, KW) / 6;
    hipLaunchKernelGGL(wino43_pack_kernel, dim3((unsigned)ceil_div_sz(total, 256)), dim3(256), 0, (hipStream_t)stream, w, ut, Co, Ci);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_wino43_pack_multi_blocks(int Co, int Ci, int KH, int KW) {
    return (int)ceil_div_sz(dynmm_wino43_packed_floats(Co, Ci, KH, KW) / 6, 256);
}

extern "C" int dynmm_wino43_pack_multi(const float* src_base, float* dst_base, const void* desc, int ndesc, int total_blocks,
                                       void* stream) {
    (void)hipGetLastError();
    if (!src_base || !dst_base || !desc || ndesc <= 0 || total_blocks <= 0) return DYNMM_EINVAL;
    if (reinterpret_cast<uintptr_t>(dst_base) & 15u) return DYNMM_EINVAL;
    static_assert(sizeof(Wino43PackDesc) == 32, "descriptor layout is part of the ABI (4 x int64 words)");
    hipLaunchKernelGGL(wino43_pack_multi_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, src_base, dst_base,
                       (const Wino43PackDesc*)desc, ndesc);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_conv2d_wino43_dgrad(const float* dy, const float* ut, const float* mask, const float* accum, float* dx,
                                         const dynmm_conv_geom* g, void* stream) {
    (void)hipGetLastError();
    if (!dy || !ut || !dx || !g) return DYNMM_EINVAL;
    if (!wino43_geom_ok(g)) return DYNMM_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(ut) | reinterpret_cast<uintptr_t>(dx) |
         reinterpret_cast<uintptr_t>(mask) | reinterpret_cast<uintptr_t>(accum)) & 15u)
        return DYNMM_EUNSUPPORTED;
    Wino43Args a{};
    a.x = dy; a.ut4 = ut; a.ut2 = ut + (size_t)g->Co * g->Ci * 4; a.residual = accum; a.mask = mask; a.y = dx;
    a.N = g->N; a.Ci = g->Co; a.Co = g->Ci; a.H = g->H; a.W = g->W;          // the roles of the channel counts swap
    a.MQ = a.N * a.H * a.W / 4;
    a.n_co_tiles = a.Co / 64;
    a.n_q_tiles = ceil_div(a.MQ, 64);
    dim3 grid((unsigned)(a.n_co_tiles * a.n_q_tiles));
    hipLaunchKernelGGL(conv_wino43_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}
