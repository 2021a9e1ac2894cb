// 3x3 convolutions by the 2-D Winograd minimal filtering algorithm F(2x2, 3x3) on the fp32 matrix cores (round 5).
//
// The BasicBlock encoders (FusionDynMM/src/models/resnet.py:42-84: two 3x3 convolutions per block), the decoder's conv3x3
// (src/models/model.py:343-357) and conv_out (:286).  For a 2x2 output tile Y with its 4x4 input patch d and the filter g
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A
// with the 1-D F(2,3) matrices of conv_wino.hip applied along both axes: SIXTEEN channel contractions per four outputs — 4 per
// output where the direct convolution does 9 and the horizontal F(2,3) form with the vertical taps looped (rounds 4-5 for these
// filters) 6: 4/9 of the direct matrix work in plain fp32.  Measured against fp64 on the test shapes: 1.3e-7 .. 5.9e-7 in
// max-norm (the 1-D form: 1.6e-7 .. 8.8e-7, a direct fp32 sum: 3e-7 .. 1.5e-6) — fewer additions per output again.
//
// Sixteen accumulator blocks per tile do not fit the 32x32 MFMA (256 registers for one 32 x 32 block set), so this kernel runs on
// `v_mfma_f32_16x16x4_f32` (4 accumulator registers per 16 x 16 block, 32 cycles): a wave owns 32 output channels x 16 tiles x 16
// transforms = 32 blocks (128 registers), a workgroup 64 channels x 32 tiles (consecutive in the flattened (image, tile row, tile
// column) order); two workgroups per CU.  A stage is 4 reduction channels = ONE k-step = 32 MFMAs per wave:
//   * filter operand U = G g G^T packed once per step as [ci][tq 4][co][4] (transform t = 4 tq + j = 4 i + c: vertical index i,
//     horizontal index c): a lane's four A values of a (co block, tq) are one ds_read_b128, conflict-free across the wave;
//   * the raw input tile is [4 channels][4 input rows][64 + 8 positions] (position = 2 x tile index in the flattened order, a
//     4-position halo either side; an input row outside the image is requested from a mapped row and read from a zero row);
//     a lane reads d[r][0] | (d[r][1], d[r][2]) | d[r][3] of its tile per row and forms B^T d B in registers (32 adds);
//   * operands by `global_load_lds_dwordx4` into a 3-slot ring requested a stage ahead (hand-counted vmcnt), one barrier per
//     stage, the fragments of stage s + 1 read under the 32 MFMAs of stage s;
//   * epilogue: A^T M A per (channel, tile) from the lane's 16 accumulators (lane = tile column, register = channel row), then
//     bias / residual / activation (forward), ReLU mask / accumulated residual gradient (input gradient), 8-byte stores of the two
//     output rows; STATS: the BatchNorm batch statistics of the tile (16-lane sums by DPP, the two tile halves through LDS, fp64
//     atomics into the slabs conv_wino.hip's STATS uses); TAIL: row counts that are not a multiple of 64 (conv_out's 40 classes).
// The input gradient of a stride-1 3x3 convolution is the 3x3 convolution of dy with the filter flipped along both axes and
// the channel roles swapped: the same kernel on a different filter pack.
#include <stdlib.h>

#include "common.h"
#include "conv_igemm.h"

namespace dynmm {

typedef float f32x4v __attribute__((ext_vector_type(4)));

struct Wino2dArgs {
    const float* x;         // [N, Ci, H, W] (input gradient: dy, Ci = the convolution's Co)
    const float* ut;        // [Ci][4][CoS][4]
    const float* shift;     // [Co] or nullptr (forward: bias)
    const float* residual;  // like y or nullptr.  forward: added before the activation; input gradient: added after the mask
    const float* mask;      // like y or nullptr (input gradient): y = mask > 0 ? y : 0
    float* y;               // [N, Co, H, W]
    double* stats;          // STATS: [nslots][2][Co]
    int nslots;
    int N, Ci, Co, H, W, CoS, act;
    int TH, TW, MT;         // tile rows per image, tile columns, tiles
    int n_co_tiles, n_t_tiles;
};

__device__ __forceinline__ float row16_sum2d(float v) {          // sum over the lane's row of 16 lanes, in every lane of it
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));     // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));     // row_mirror
    return v;
}

template <bool DGRAD, bool TAIL, bool STATS>
__global__ void __launch_bounds__(256, 2) conv_wino2d_kernel(const Wino2dArgs a) {
    static_assert(!STATS || (!DGRAD && !TAIL), "statistics: the forward on full 64-row tiles");
    static_assert(!TAIL || !DGRAD, "row tails exist in the forward only");
    constexpr int BK = 4, S = 3, TCO = 64, TT = 32;
    constexpr int A_STAGE = BK * 4 * TCO * 4;                  // floats (16 KB)
    constexpr int PW = 2 * TT + 8, BROW = PW, BCH = 4 * BROW;  // 72 positions per row, 4 rows per channel
    constexpr int B_STAGE = BK * BCH;                          // 1152 floats
    constexpr int QB = B_STAGE / 4, QPW = QB / 4;              // 288 quads per stage, 72 per wave
    constexpr int NIB = (QPW + 63) / 64, NIA = 4, NI = NIA + NIB;
    static_assert(3 * NI < 64, "vmcnt is a 6-bit counter");
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
        Q = Q < 0 ? 0 : (Q > Qmax ? Qmax : Q);                 // (quads past the tensor: any mapped address, never used)
        const int rowid = Q / a.W, xx = Q - rowid * a.W;
        const int n = rowid / a.TH, th = rowid - n * a.TH;
        int h = 2 * th - 1 + r;
        h = h < 0 ? 0 : (h > a.H - 1 ? a.H - 1 : h);           // (a row outside the image: mapped; the reader takes the zero row)
        b_off[i] = ((unsigned)(n * a.Ci + k) * (unsigned)HW + (unsigned)(h * a.W + xx)) * 4u;
    }
    const unsigned a_voff = (unsigned)lane * 16u;
    const unsigned lds_a = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)As);
    const unsigned lds_b = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)Bs);
    // wave w stages channel w of the stage: 4 (tq) pieces of 64 co x 4 floats = 1 KB each; running pointers, constant strides
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
    const int a_frag = ((kq * 4) * TCO + wave_co * 32 + l15) * 4;        // + (tq * TCO + cb * 16) * 4
    const int b_frag = kq * BCH + 2 * tl + 3;                            // + r * BROW
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
    wait_vm<2 * NI>();                            // (nst >= 3: the launcher requires >= 12 reduction channels)
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
    if constexpr (!STATS) {
        if (!tvalid) return;
    }
    const bool y1ok = r2ok;                                     // the tile's second output row exists
    const size_t pix = (size_t)(2 * pth) * a.W + 2 * ptw;
    float* const st_lds = As;                                   // STATS: [wave_t 2][64 co][2] partial sums (after the barrier below)
    if constexpr (STATS) __syncthreads();                       // every wave is done with the rings
    const float* __restrict__ mask_p = a.mask;
    const float* __restrict__ res_p = a.residual;
    float* __restrict__ y_p = a.y;
    const bool has_mask = DGRAD && mask_p != nullptr, has_res = res_p != nullptr;
    auto co_of = [&](int cb, int i) { return wave_co * 32 + cb * 16 + 4 * kq + i; };
    // the epilogue's operands (two tensors as large as the output) are requested for all 8 channels of the lane before the
    // first output transform: with two workgroups per CU nothing else hides their latency (conv_wino.hip's lesson: the first
    // version loaded them channel by channel — C = 64 input gradient with mask + accumulated gradient 384 us against 248 plain)
    float2 km[2][4][2], rs[2][4][2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int co = co0 + co_of(cb, i);
            const bool live = tvalid && (!TAIL || co < a.Co);
            const size_t off = ((size_t)pn * a.Co + co) * HW + pix;
            km[cb][i][0] = km[cb][i][1] = make_float2(1.f, 1.f);
            rs[cb][i][0] = rs[cb][i][1] = make_float2(0.f, 0.f);
            if (has_mask && live) {
                km[cb][i][0] = *reinterpret_cast<const float2*>(mask_p + off);
                if (y1ok) km[cb][i][1] = *reinterpret_cast<const float2*>(mask_p + off + a.W);
            }
            if (has_res && live) {
                rs[cb][i][0] = *reinterpret_cast<const float2*>(res_p + off);
                if (y1ok) rs[cb][i][1] = *reinterpret_cast<const float2*>(res_p + off + a.W);
            }
        }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cl = co_of(cb, i);
            const int co = co0 + cl;
            const bool clive = !TAIL || co < a.Co;
            float z[2][4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float m0 = acc[c][cb][i], m1 = acc[4 + c][cb][i], m2 = acc[8 + c][cb][i], m3 = acc[12 + c][cb][i];
                z[0][c] = (m0 + m1) + m2;
                z[1][c] = (m1 - m2) - m3;
            }
            const float sh = (a.shift && clive) ? a.shift[co] : 0.f;
            float y00 = (z[0][0] + z[0][1]) + z[0][2] + sh, y01 = (z[0][1] - z[0][2]) - z[0][3] + sh;
            float y10 = (z[1][0] + z[1][1]) + z[1][2] + sh, y11 = (z[1][1] - z[1][2]) - z[1][3] + sh;
            const size_t off = ((size_t)pn * a.Co + co) * HW + pix;
            const bool live = tvalid && clive;
            if constexpr (DGRAD) {
                y00 = km[cb][i][0].x > 0.f ? y00 : 0.f; y01 = km[cb][i][0].y > 0.f ? y01 : 0.f;
                y10 = km[cb][i][1].x > 0.f ? y10 : 0.f; y11 = km[cb][i][1].y > 0.f ? y11 : 0.f;
            }
            y00 += rs[cb][i][0].x; y01 += rs[cb][i][0].y;
            y10 += rs[cb][i][1].x; y11 += rs[cb][i][1].y;
            if constexpr (!DGRAD) {
                if (a.act == DYNMM_ACT_RELU) {
                    y00 = y00 > 0.f ? y00 : 0.f; y01 = y01 > 0.f ? y01 : 0.f;
                    y10 = y10 > 0.f ? y10 : 0.f; y11 = y11 > 0.f ? y11 : 0.f;
                } else if (a.act == DYNMM_ACT_TANH) {
                    y00 = tanhf(y00); y01 = tanhf(y01); y10 = tanhf(y10); y11 = tanhf(y11);
                }
            }
            if (live) {
                *reinterpret_cast<float2*>(y_p + off) = make_float2(y00, y01);
                if (y1ok) *reinterpret_cast<float2*>(y_p + off + a.W) = make_float2(y10, y11);
            }
            if constexpr (STATS) {
                const bool l1 = tvalid && y1ok;
                float s1 = tvalid ? y00 + y01 : 0.f, s2 = tvalid ? fmaf(y00, y00, y01 * y01) : 0.f;
                s1 += l1 ? y10 + y11 : 0.f;
                s2 += l1 ? fmaf(y10, y10, y11 * y11) : 0.f;
                s1 = row16_sum2d(s1);
                s2 = row16_sum2d(s2);
                if (l15 == 0) {
                    st_lds[(wave_t * 64 + cl) * 2 + 0] = s1;
                    st_lds[(wave_t * 64 + cl) * 2 + 1] = s2;
                }
            }
        }
    if constexpr (STATS) {
        __syncthreads();
        if (t < 128) {                                          // (channel t >> 1, statistic t & 1): the two tile halves, one atomic
            const int cl = t >> 1, stat = t & 1;
            const float v = st_lds[cl * 2 + stat] + st_lds[(64 + cl) * 2 + stat];
            atomicAdd(a.stats + ((size_t)((lin / a.n_co_tiles) % a.nslots) * 2 + stat) * a.Co + co0 + cl, (double)v);
        }
    }
}

// Filter transforms U = G g G^T.  w [Co][Ci][3][3] -> ut [K][4 tq][Cs][4] with (K, C) = (Ci, Co) for the forward operand and
// (Co, Ci) for the input gradient's (filter flipped along both axes); rows padded to the 64-row tile with zeros.
__device__ __forceinline__ float4 wino2d_u(const float* __restrict__ g, int tq, bool flip, float sc) {
    float gr[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const int bb = flip ? 2 - b : b;
        float g0 = g[(flip ? 6 : 0) + bb], g1 = g[3 + bb], g2 = g[(flip ? 0 : 6) + bb];
        g0 *= sc; g1 *= sc; g2 *= sc;
        gr[b] = tq == 0 ? g0 : (tq == 1 ? (g0 + g1 + g2) * 0.5f : (tq == 2 ? (g0 - g1 + g2) * 0.5f : g2));
    }
    return make_float4(gr[0], (gr[0] + gr[1] + gr[2]) * 0.5f, (gr[0] - gr[1] + gr[2]) * 0.5f, gr[2]);
}

__global__ void __launch_bounds__(256) wino2d_pack_kernel(const float* __restrict__ w, float4* __restrict__ ut,
                                                          const float* __restrict__ scale, int Co, int Ci, int dgrad) {
    const int K = dgrad ? Co : Ci, Cr = dgrad ? Ci : Co, Cs = (Cr + 63) & ~63;
    const size_t total = (size_t)K * 4 * Cs;
    const size_t o = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (o >= total) return;
    const int c = (int)(o % Cs);
    const int tq = (int)((o / Cs) % 4);
    const int k = (int)(o / ((size_t)Cs * 4));
    if (c >= Cr) { ut[o] = make_float4(0.f, 0.f, 0.f, 0.f); return; }
    const int co = dgrad ? k : c, ci = dgrad ? c : k;
    ut[o] = wino2d_u(w + ((size_t)co * Ci + ci) * 9, tq, dgrad != 0, scale ? scale[co] : 1.f);
}

// Many filters in one launch (ops.PackedWeights): the descriptor layout of conv_wino.hip's wino_pack_multi (32 bytes).
struct Wino2dPackDesc {
    long long src, dst;
    int Co, Ci, kk, blk0;      // kk: dgrad << 16
};

__global__ void __launch_bounds__(256) wino2d_pack_multi_kernel(const float* __restrict__ src_base, float* __restrict__ dst_base,
                                                                const Wino2dPackDesc* __restrict__ desc, int ndesc) {
    int lo = 0, hi = ndesc - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (desc[mid].blk0 <= (int)blockIdx.x) lo = mid;
        else hi = mid - 1;
    }
    const Wino2dPackDesc d = desc[lo];
    const int dgrad = (d.kk >> 16) & 1;
    const int K = dgrad ? d.Co : d.Ci, Cr = dgrad ? d.Ci : d.Co, Cs = (Cr + 63) & ~63;
    const size_t total = (size_t)K * 4 * Cs;
    const size_t o = (size_t)((int)blockIdx.x - d.blk0) * 256 + threadIdx.x;
    if (o >= total) return;
    const int c = (int)(o % Cs);
    const int tq = (int)((o / Cs) % 4);
    const int k = (int)(o / ((size_t)Cs * 4));
    float4* ut = reinterpret_cast<float4*>(dst_base + d.dst);
    if (c >= Cr) { ut[o] = make_float4(0.f, 0.f, 0.f, 0.f); return; }
    const int co = dgrad ? k : c, ci = dgrad ? c : k;
    ut[o] = wino2d_u(src_base + d.src + ((size_t)co * d.Ci + ci) * 9, tq, dgrad != 0, 1.f);
}

// rows = output channels of the GEMM, red = its reduction channels (4 per stage, >= 3 stages): forward (Co, Ci), input
// gradient (Ci, Co).  Forward rows: any multiple of 8 from 24 up (TAIL instantiation); input gradient rows: multiples of 64.
static bool wino2d_geom_ok(const dynmm_conv_geom* g, bool dgrad) {
    if (!g || g->c_split != g->Ci) return false;
    if (g->KH != 3 || g->KW != 3 || g->SH != 1 || g->SW != 1 || g->PH != 1 || g->PW != 1) return false;
    if (g->H != g->Ho || g->W != g->Wo || g->W % 4 != 0 || g->W < 4 || g->H < 2) return false;
    const int rows = dgrad ? g->Ci : g->Co, red = dgrad ? g->Co : g->Ci;
    if ((dgrad ? rows % 64 != 0 : (rows % 8 != 0 || rows < 24)) || red % 4 != 0 || red < 12) return false;
    if ((long long)g->N * g->H * g->W < 256) return false;
    if ((double)g->N * (g->Ci > g->Co ? g->Ci : g->Co) * g->H * g->W >= 1073741824.0) return false;   // 32-bit byte offsets
    return true;
}

static int launch_wino2d(Wino2dArgs& a, bool dgrad, hipStream_t st) {
    a.TH = (a.H + 1) / 2;
    a.TW = a.W / 2;
    a.MT = a.N * a.TH * a.TW;
    a.CoS = (a.Co + 63) & ~63;
    a.n_co_tiles = a.CoS / 64;
    a.n_t_tiles = ceil_div(a.MT, 32);
    dim3 grid((unsigned)(a.n_co_tiles * a.n_t_tiles));
    if (dgrad) hipLaunchKernelGGL((conv_wino2d_kernel<true, false, false>), grid, dim3(256), 0, st, a);
    else if (a.stats) hipLaunchKernelGGL((conv_wino2d_kernel<false, false, true>), grid, dim3(256), 0, st, a);
    else if (a.Co % 64 != 0) hipLaunchKernelGGL((conv_wino2d_kernel<false, true, false>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_wino2d_kernel<false, false, false>), grid, dim3(256), 0, st, a);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

}  // namespace dynmm

using namespace dynmm;

extern "C" int dynmm_conv2d_wino2d_supported(const dynmm_conv_geom* g, int dgrad) { return wino2d_geom_ok(g, dgrad != 0) ? 1 : 0; }

extern "C" size_t dynmm_wino2d_packed_floats(int Co, int Ci) {
    if (Co <= 0 || Ci <= 0) return 0;
    const size_t fwd = (size_t)Ci * ((Co + 63) & ~63), dg = (size_t)Co * ((Ci + 63) & ~63);
    return (fwd > dg ? fwd : dg) * 16;
}

extern "C" int dynmm_wino2d_pack(const float* w, float* ut, const float* scale, int Co, int Ci, int dgrad, void* stream) {
    (void)hipGetLastError();
    if (!w || !ut || Co <= 0 || Ci <= 0 || (scale && dgrad) || dgrad < 0 || dgrad > 1) return DYNMM_EINVAL;
    if (reinterpret_cast<uintptr_t>(ut) & 15u) return DYNMM_EINVAL;
    const int K = dgrad ? Co : Ci, Cs = ((dgrad ? Ci : Co) + 63) & ~63;
    const size_t total = (size_t)K * 4 * Cs;
    hipLaunchKernelGGL(wino2d_pack_kernel, dim3((unsigned)ceil_div_sz(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       reinterpret_cast<float4*>(ut), scale, Co, Ci, dgrad);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_wino2d_pack_multi_blocks(int Co, int Ci, int dgrad) {
    const int K = dgrad ? Co : Ci, Cs = ((dgrad ? Ci : Co) + 63) & ~63;
    return (int)ceil_div_sz((size_t)K * 4 * Cs, 256);
}

extern "C" int dynmm_wino2d_pack_multi(const float* src_base, float* dst_base, const void* desc, int ndesc, int total_blocks,
                                       void* stream) {
    (void)hipGetLastError();
    if (!src_base || !dst_base || !desc || ndesc <= 0 || total_blocks <= 0) return DYNMM_EINVAL;
    if (reinterpret_cast<uintptr_t>(dst_base) & 15u) return DYNMM_EINVAL;
    static_assert(sizeof(Wino2dPackDesc) == 32, "descriptor layout is part of the ABI (4 x int64 words)");
    hipLaunchKernelGGL(wino2d_pack_multi_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, src_base, dst_base,
                       (const Wino2dPackDesc*)desc, ndesc);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

extern "C" int dynmm_conv2d_wino2d_fwd(const float* x, const float* ut, const float* bias, const float* residual, float* y,
                                       double* stats, int nslots, const dynmm_conv_geom* g, int act, void* stream) {
    (void)hipGetLastError();
    if (!x || !ut || !y || !g) return DYNMM_EINVAL;
    if (!wino2d_geom_ok(g, false)) return DYNMM_EUNSUPPORTED;
    if (stats && (g->Co % 64 != 0 || residual || act != DYNMM_ACT_NONE || nslots < 1 || nslots > 64)) return DYNMM_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(ut)) & 15u) return DYNMM_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(residual) | reinterpret_cast<uintptr_t>(stats)) & 7u)
        return DYNMM_EUNSUPPORTED;
    Wino2dArgs a{};
    a.x = x; a.ut = ut; a.shift = bias; a.residual = residual; a.mask = nullptr; a.y = y; a.stats = stats; a.nslots = nslots;
    a.N = g->N; a.Ci = g->Ci; a.Co = g->Co; a.H = g->H; a.W = g->W; a.act = act;
    return launch_wino2d(a, false, (hipStream_t)stream);
}

extern "C" int dynmm_conv2d_wino2d_stats_slots(const dynmm_conv_geom* g) {
    if (!wino2d_geom_ok(g, false) || g->Co % 64 != 0) return 0;
    const int tiles = ceil_div(g->N * ((g->H + 1) / 2) * (g->W / 2), 32);       // each adds once per channel and statistic
    const int s = tiles / 600;
    return s < 1 ? 1 : (s > 8 ? 8 : s);
}

extern "C" int dynmm_conv2d_wino2d_dgrad(const float* dy, const float* ut, const float* mask, const float* accum, float* dx,
                                         const dynmm_conv_geom* g, void* stream) {
    (void)hipGetLastError();
    if (!dy || !ut || !dx || !g) return DYNMM_EINVAL;
    if (!wino2d_geom_ok(g, true)) return DYNMM_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(dy) | reinterpret_cast<uintptr_t>(ut)) & 15u) return DYNMM_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(mask) | reinterpret_cast<uintptr_t>(accum)) & 7u)
        return DYNMM_EUNSUPPORTED;
    Wino2dArgs a{};
    a.x = dy; a.ut = ut; a.shift = nullptr; a.residual = accum; a.mask = mask; a.y = dx; a.stats = nullptr; a.nslots = 1;
    a.N = g->N; a.Ci = g->Co; a.Co = g->Ci; a.H = g->H; a.W = g->W; a.act = DYNMM_ACT_NONE;       // the channel roles swap
    return launch_wino2d(a, true, (hipStream_t)stream);
}
