// Direct (no MFMA) convolutions for the gate head's first conv (…globalgate.py:378-386: 128 -> 8 channels, 5x5,
// stride 2, no padding, on the 120x160 stage-1 maps).  With 8 output channels an MFMA tile is 3/4 padding (the
// implicit-GEMM kernel ran it at 11 TFLOP/s useful); on the vector ALUs — same fp32 FMA peak as the fp32 MFMA on
// CDNA4 — nothing is padded: the filter taps are wave-uniform and live in SGPRs (s_load from the packed weight
// buffer, one FMA operand straight from the scalar file), the input rows come from a wave-private LDS tile.
//
//   gate conv forward : a workgroup owns 4 x 80 output pixels of one image; its 4 waves split the INPUT channels
//             (each wave runs Ci/4 of them, wave-synchronously: no barrier in the channel loop), a lane keeps
//             8 channels x 5 pixels of accumulators; partial sums meet in LDS in a fixed order (bit-reproducible).
//
// The two stem convolutions (resnet.py:229: 3 / 1 -> 64 channels, 7x7, stride 2, pad 3, on the 480x640 inputs) get
// an fp32-MFMA kernel of their own further down: K = Ci * 49 is too short and too ragged for the implicit-GEMM
// loader (it ran at 42 TFLOP/s through the element-wise generic path).
#include "common.h"
#include "conv_small.h"
#include "conv_igemm.h"

namespace dynmm {

constexpr int kSP = 5;                // output pixels per lane (consecutive along W)
constexpr int kSRows = 4;             // output rows per workgroup
constexpr int kSColGroups = 16;       // lanes along W
constexpr int kSCols = kSColGroups * kSP;      // 80 output columns per workgroup

__device__ __forceinline__ float small_act(float v, int act) {
    if (act == DYNMM_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == DYNMM_ACT_TANH) return tanhf(v);
    return v;
}

template <int KS, int S>
__global__ void __launch_bounds__(256) conv_co8_fwd_kernel(const SmallConvArgs a) {
    constexpr int IR = (kSRows - 1) * S + KS;          // input rows of a tile (11)
    constexpr int IC = (kSCols - 1) * S + KS;          // input cols (163)
    constexpr int TILE = IR * IC;
    constexpr int NLD = (TILE + 63) / 64;              // staging loads per lane and channel
    constexpr int XW = (kSP - 1) * S + KS;             // input columns one lane touches per row (13)
    constexpr int RED = 4 * 8 * kSRows * kSCols;       // cross-wave reduction buffer (floats)
    constexpr int LDSF = (4 * 2 * TILE > RED) ? 4 * 2 * TILE : RED;
    __shared__ float lds[LDSF];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tw = (a.Wo + kSCols - 1) / kSCols, th = (a.Ho + kSRows - 1) / kSRows;
    int b = blockIdx.x;
    const int n = b / (th * tw);
    b -= n * th * tw;
    const int oh0 = (b / tw) * kSRows, ow0 = (b - (b / tw) * tw) * kSCols;
    const int ih0 = oh0 * S, iw0 = ow0 * S;
    const int r = lane >> 4, cg = lane & 15;           // the lane's output row / column group inside the tile

    // staging geometry: element e = k*64 + lane of the flat [IR][IC] tile
    int goff[NLD];
    unsigned okmask = 0u;
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int e = k * 64 + lane;
        const int rr = e / IC, cc = e - rr * IC;
        const bool ok = e < TILE && ih0 + rr < a.H && iw0 + cc < a.W;
        goff[k] = ok ? (ih0 + rr) * a.W + iw0 + cc : 0;
        okmask |= ok ? (1u << k) : 0u;
    }
    const int cpw = a.Ci / 4;                           // input channels per wave
    const int c_begin = wave * cpw;
    const int HW = a.H * a.W;
    float* const tile0 = lds + wave * 2 * TILE;

    auto plane = [&](int ci) -> const float* {
        return ci < a.c_split ? a.x + ((size_t)n * a.c_split + ci) * HW
                              : a.x2 + ((size_t)n * (a.Ci - a.c_split) + (ci - a.c_split)) * HW;
    };
    float st[NLD];
    auto stage_load = [&](int ci) {
        const float* p = plane(ci);
#pragma unroll
        for (int k = 0; k < NLD; ++k) st[k] = p[goff[k]];
    };
    auto stage_store = [&](float* t) {
#pragma unroll
        for (int k = 0; k < NLD; ++k)
            if (k * 64 + lane < TILE) t[k * 64 + lane] = ((okmask >> k) & 1u) ? st[k] : 0.f;
    };

    float acc[8][kSP];
#pragma unroll
    for (int co = 0; co < 8; ++co)
#pragma unroll
        for (int p = 0; p < kSP; ++p) acc[co][p] = 0.f;

    // wave-private tiles, wave-synchronous use: LDS executes one wave's operations in order, the fences only keep
    // the compiler from moving a lane's reads across the (other lanes') writes
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    stage_load(c_begin);
    stage_store(tile0);
    wave_sync();
    for (int i = 0; i < cpw; ++i) {
        const int ci = c_begin + i;
        const float* t = tile0 + (i & 1) * TILE;
        if (i + 1 < cpw) stage_load(ci + 1);                       // in flight during this channel's arithmetic
        const float* wci = a.wp + (size_t)ci * 8;                  // wp[(tap * CiR + ci) * 8 + co]
#pragma unroll
        for (int kh = 0; kh < KS; ++kh) {
            const float* row = t + (S * r + kh) * IC + S * kSP * cg;
            float xv[XW];
#pragma unroll
            for (int j = 0; j < XW; ++j) xv[j] = row[j];
#pragma unroll
            for (int kw = 0; kw < KS; ++kw) {
                const float* wt = wci + (size_t)(kh * KS + kw) * a.CiR * 8;     // wave-uniform: 8 floats from SGPRs
#pragma unroll
                for (int co = 0; co < 8; ++co) {
                    const float wv = wt[co];
#pragma unroll
                    for (int p = 0; p < kSP; ++p) acc[co][p] = fmaf(wv, xv[S * p + kw], acc[co][p]);
                }
            }
        }
        if (i + 1 < cpw) stage_store(tile0 + ((i + 1) & 1) * TILE);
        wave_sync();
    }
    __syncthreads();                                                // staging tiles are dead: reuse as red[wave][co][r][col]
#pragma unroll
    for (int co = 0; co < 8; ++co)
#pragma unroll
        for (int p = 0; p < kSP; ++p)
            lds[((wave * 8 + co) * kSRows + r) * kSCols + cg * kSP + p] = acc[co][p];
    __syncthreads();
    constexpr int PER = 8 * kSRows * kSCols;
    for (int e = threadIdx.x; e < PER; e += 256) {
        const int co = e / (kSRows * kSCols), rem = e - co * (kSRows * kSCols);
        const int rr = rem / kSCols, cc = rem - rr * kSCols;
        const int oh = oh0 + rr, ow = ow0 + cc;
        if (co < a.Co && oh < a.Ho && ow < a.Wo) {
            float v = ((lds[e] + lds[PER + e]) + lds[2 * PER + e]) + lds[3 * PER + e];
            v = v * (a.scale ? a.scale[co] : 1.f) + (a.shift ? a.shift[co] : 0.f);
            a.y[(((size_t)n * a.Co + co) * a.Ho + oh) * a.Wo + ow] = small_act(v, a.act);
        }
    }
}

// ---- gate conv weight gradient (round 5) ------------------------------------------------------------------------------
// dW[co][ci][kh][kw] = sum over (n, oh, ow) of dy[n][co][oh][ow] * x[n][ci][2 oh + kh][2 ow + kw]   (…globalgate.py:378-386,
// the 128 -> 8 channel 5x5 stride-2 convolution of the gate head on the 120x160 stage-1 maps of BOTH encoders).  With 8 output
// channels the implicit-GEMM tile is 3/4 padding (10.8 TFLOP/s, 0.69 ms per step); the same observation as for the forward
// above: on the vector ALUs nothing is padded.
//   A workgroup owns G input channels x NI images and ALL 25 x 8 weights of those channels.  A lane keeps the 200
//   accumulators of one channel (acc[kh][kw][co]) for its 5 consecutive output pixels of one output row; the four waves
//   cover 16 output rows per pass.  Per (image, pass): the 35 input rows the pass touches are ONE contiguous piece of the
//   channel's plane — staged by `global_load_lds_dwordx4` into a double-buffered LDS tile (the next piece is in flight under
//   this one's arithmetic); the lane's 8 x 5 dy values come straight from L2 (dy is 4.6 MB in all).  1000 FMAs per lane per
//   staged piece against 65 LDS reads.  After the channel's last piece: 16-lane sums by DPP, the 16 row sums of the workgroup
//   through LDS in a fixed order, one slab row per image group; the slabs (and the bias-gradient slabs of the first channel
//   group's workgroups) are summed by the library's ordered slab reduction — bit-reproducible like every other weight gradient.
struct Co8WgradArgs {
    const float* x;
    const float* x2;
    const float* dy;
    float* slabs;          // [image groups][Co * Ci * 25]
    float* bias_slabs;     // [image groups][Co] or nullptr
    int N, Ci, H, W, Co, Ho, Wo, c_split, NI, G;
};

__device__ __forceinline__ float row16_sum(float v) {            // sum over the lane's row of 16 lanes, in every lane of it
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));      // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));      // quad_perm [2,3,0,1]
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x141, 0xF, 0xF, true));     // row_half_mirror
    v += __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x140, 0xF, 0xF, true));     // row_mirror
    return v;
}

template <int KS, int S>
__global__ void __launch_bounds__(256, 1) conv_co8_wgrad_kernel(const Co8WgradArgs a) {
    constexpr int PR = 16;                             // output rows per pass (4 waves x 4 lane rows)
    constexpr int IR = (PR - 1) * S + KS;              // input rows a pass touches (35)
    constexpr int XW = (kSP - 1) * S + KS;             // input columns a lane touches per row (13)
    constexpr int NACC = KS * KS * 8;
    extern __shared__ __attribute__((aligned(16))) float dyn_lds[];
    const int tile_floats = IR * a.W + 16;             // (+ slack: the last lanes' windows run a few floats past the piece)
    float* const tiles = dyn_lds;                      // [2][tile_floats]
    float* const red = dyn_lds + 2 * tile_floats;      // [16 lane rows of the workgroup][NACC + 8]

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int cg = lane & 15, rl = lane >> 4;
    const int n_cgroups = a.Ci / a.G;
    const int cgrp = (int)blockIdx.x % n_cgroups, igrp = (int)blockIdx.x / n_cgroups;
    const int n0 = igrp * a.NI, n1 = min(a.N, n0 + a.NI);
    const int passes = (a.Ho + PR - 1) / PR;
    const int HW = a.H * a.W;
    const unsigned lds_t = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)tiles);

    auto plane = [&](int n, int ci) -> const float* {
        return ci < a.c_split ? a.x + ((size_t)n * a.c_split + ci) * HW
                              : a.x2 + ((size_t)n * (a.Ci - a.c_split) + (ci - a.c_split)) * HW;
    };
    // piece q = ((ci_local * (n1 - n0)) + n_local) * passes + pass of this workgroup's sequence
    const int per_ci = (n1 - n0) * passes;
    const int total = a.G * per_ci;
    auto request = [&](int q) {                         // DMA of piece q into tile q & 1: 16 bytes per lane and instruction
        const int cl = q / per_ci, rem = q - cl * per_ci;
        const int nl = rem / passes, ps = rem - nl * passes;
        const int row0 = ps * PR * S;
        const int rows = min(IR, a.H - row0);
        const int quads = rows * a.W / 4;
        const float* src = plane(n0 + nl, cgrp * a.G + cl) + (size_t)row0 * a.W;
        const unsigned dst = lds_t + (unsigned)((q & 1) * tile_floats * 4);
        for (int base = wave * 64; base < quads; base += 256) {
            const int qd = base + lane;
            if (qd < quads) dma16(src + (size_t)base * 4, (unsigned)lane * 16u, dst + (unsigned)base * 16u);
        }
    };

    float acc[KS][KS][8];
    float bacc[8];
#pragma unroll
    for (int co = 0; co < 8; ++co) bacc[co] = 0.f;
    const bool do_bias = a.bias_slabs != nullptr && cgrp == 0;

    // Lanes without a live output pixel (rows past Ho in the last pass, the column group past Wo) read tile positions no piece
    // of this pass wrote, against dy = 0: the tiles start as zeros, so such a position holds zeros or an earlier piece's finite
    // values — never an uninitialised bit pattern that could be a NaN.  The windows of the dead column groups reach up to
    // 146 - W floats PAST the second tile, i.e. into `red` (W = 16: 130 floats, W = 80: 66): `red` starts as zeros too (round 6 — a
    // NaN pattern left in LDS by an earlier kernel made 0 * NaN a NaN in a dead lane's accumulator and, through the 16-lane row
    // sums, in the result: one failure of test_gate_conv_weight_gradient_on_the_vector_alus[case2] in the first seconds of a
    // fresh box, never reproduced in 25 repetitions); later it holds the previous channel's finite sums.
    for (int i = t; i < 2 * tile_floats + 16 * (NACC + 8); i += 256) dyn_lds[i] = 0.f;
    __syncthreads();
    request(0);
    for (int q = 0; q < total; ++q) {
        const int cl = q / per_ci, rem = q - cl * per_ci;
        const int nl = rem / passes, ps = rem - nl * passes;
        if (rem == 0) {
#pragma unroll
            for (int kh = 0; kh < KS; ++kh)
#pragma unroll
                for (int kw = 0; kw < KS; ++kw)
#pragma unroll
                    for (int co = 0; co < 8; ++co) acc[kh][kw][co] = 0.f;
        }
        // this lane's dy values of the pass (zero outside the output map / past Co)
        const int orow = ps * PR + wave * 4 + rl;
        float dyv[8][kSP];
#pragma unroll
        for (int co = 0; co < 8; ++co)
#pragma unroll
            for (int p = 0; p < kSP; ++p) {
                const int ow = cg * kSP + p;
                const bool ok = co < a.Co && orow < a.Ho && ow < a.Wo;
                dyv[co][p] = ok ? a.dy[(((size_t)(n0 + nl) * a.Co + co) * a.Ho + orow) * a.Wo + ow] : 0.f;
            }
        // every wave is done with the other tile (read during piece q - 1); this wave's requests for piece q have landed
        wait_vm<0>();                                   // (the dy loads above as well: they are consumed right away)
        __syncthreads();
        if (q + 1 < total) request(q + 1);
        const float* tile = tiles + (q & 1) * tile_floats;
#pragma unroll
        for (int kh = 0; kh < KS; ++kh) {
            const float* row = tile + (S * (wave * 4 + rl) + kh) * a.W + S * kSP * cg;
            float xv[XW];
#pragma unroll
            for (int j = 0; j < XW; ++j) xv[j] = row[j];
#pragma unroll
            for (int kw = 0; kw < KS; ++kw)
#pragma unroll
                for (int co = 0; co < 8; ++co)
#pragma unroll
                    for (int p = 0; p < kSP; ++p) acc[kh][kw][co] = fmaf(dyv[co][p], xv[S * p + kw], acc[kh][kw][co]);
        }
        if (do_bias && cl == 0) {
#pragma unroll
            for (int co = 0; co < 8; ++co) bacc[co] += ((dyv[co][0] + dyv[co][1]) + (dyv[co][2] + dyv[co][3])) + dyv[co][4];
        }
        if (rem == per_ci - 1) {                        // the channel is complete: reduce its 200 sums over the workgroup
            __syncthreads();                            // (red is read by the previous channel's writers until here)
            float* dst = red + (wave * 4 + rl) * (NACC + 8);
#pragma unroll
            for (int kh = 0; kh < KS; ++kh)
#pragma unroll
                for (int kw = 0; kw < KS; ++kw)
#pragma unroll
                    for (int co = 0; co < 8; ++co) {
                        const float sres = row16_sum(acc[kh][kw][co]);
                        if (cg == 0) dst[(co * KS + kh) * KS + kw] = sres;
                    }
            const bool last_bias = do_bias && cl == 0;
            if (last_bias) {
#pragma unroll
                for (int co = 0; co < 8; ++co) {
                    const float sres = row16_sum(bacc[co]);
                    if (cg == 0) dst[NACC + co] = sres;
                }
            }
            __syncthreads();
            const int ci = cgrp * a.G + cl;
            if (t < NACC + (last_bias ? 8 : 0)) {
                float v = 0.f;
#pragma unroll
                for (int r16 = 0; r16 < 16; ++r16) v += red[r16 * (NACC + 8) + t];
                if (t < NACC) {
                    const int co = t / (KS * KS), tap = t - co * (KS * KS);
                    if (co < a.Co) a.slabs[(size_t)igrp * a.Co * a.Ci * (KS * KS) + ((size_t)co * a.Ci + ci) * (KS * KS) + tap] = v;
                } else if (t - NACC < a.Co) {
                    a.bias_slabs[(size_t)igrp * a.Co + (t - NACC)] = v;
                }
            }
        }
    }
}

static void co8_wgrad_plan(int N, int Ci, int* NI, int* G) {
    *G = 4;                                            // channels per workgroup
    int ni = (int)(((long)N * (Ci / 4)) / 256);        // one workgroup per CU (200 accumulators per lane): ~256 workgroups
    *NI = ni < 1 ? 1 : (ni > 8 ? 8 : ni);
}

bool co8_wgrad_eligible(int Ci, int Co, int H, int W, int Ho, int Wo, int KH, int KW, int SH, int SW, int PH, int PW, int c_split) {
    if (KH != 5 || KW != 5 || SH != 2 || SW != 2 || PH != 0 || PW != 0) return false;
    if (Co < 1 || Co > 8 || Ci < 16 || Ci % 4 != 0 || W % 4 != 0) return false;
    if (c_split != Ci && c_split % 4 != 0) return false;               // a channel group never straddles the two inputs
    if (Wo > kSCols || Ho < 1) return false;                           // one column group set covers the output row
    return (size_t)(2 * (35 * W + 16) + 16 * 208) * sizeof(float) <= 160 * 1024;
}

size_t co8_wgrad_workspace_bytes(int N, int Ci, int Co) {
    int NI, G;
    co8_wgrad_plan(N, Ci, &NI, &G);
    const size_t groups = (size_t)ceil_div(N, NI);
    return sizeof(float) * (((groups * Co * Ci * 25 + 3) & ~(size_t)3) + groups * Co);
}

int launch_co8_wgrad(const float* x, const float* x2, const float* dy, float* dw, float* dbias, float* workspace, int N, int Ci,
                     int H, int W, int Co, int Ho, int Wo, int c_split, hipStream_t st) {
    Co8WgradArgs a{};
    co8_wgrad_plan(N, Ci, &a.NI, &a.G);
    const int groups = ceil_div(N, a.NI);
    const size_t wfloats = ((size_t)groups * Co * Ci * 25 + 3) & ~(size_t)3;
    a.x = x; a.x2 = x2; a.dy = dy;
    a.slabs = groups > 1 ? workspace : dw;
    a.bias_slabs = dbias ? (groups > 1 ? workspace + wfloats : dbias) : nullptr;
    a.N = N; a.Ci = Ci; a.H = H; a.W = W; a.Co = Co; a.Ho = Ho; a.Wo = Wo; a.c_split = c_split;
    const size_t lds = (size_t)(2 * (35 * W + 16) + 16 * 208) * sizeof(float);
    static bool attr_set = false;
    if (!attr_set) {
        DYNMM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_co8_wgrad_kernel<5, 2>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_co8_wgrad_kernel<5, 2>), dim3((unsigned)(groups * (Ci / a.G))), dim3(256), lds, st, a);
    DYNMM_LAUNCH_CHECK();
    if (groups > 1)
        launch_reduce_slabs(workspace, dw, Co * Ci * 25, groups, st, dbias ? workspace + wfloats : nullptr, dbias, dbias ? Co : 0);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

// ---- stem: 1 or 3 input channels, 64 output channels, 7x7 stride 2 pad 3 — fp32 MFMA from an LDS patch ---------
// GEMM view: out[co][pix] = sum_k W[co][k] * X[k][pix], k = (ci, kh, kw) with kw padded 7 -> 8 (a zero weight
// column), so that the two k's of one v_mfma_f32_32x32x2_f32 step are always horizontal neighbours of the same
// input row: lane (pixel l31, k-parity khalf) reads patch[base + khalf + compile-time offset] — stride-2 dwords over
// the half-wave (even banks) and the other half on the odd banks: conflict-free without any im2col buffer.
// A workgroup owns 64 channels x (4 rows x 64 cols) of one image: wave w = channel tile w & 1, output rows
// 2 * (w >> 1) + {0, 1}, i.e. 1 channel tile x 4 pixel tiles of accumulators; its weight operand (84 / 28 values per
// lane) lives in registers for the lifetime of the persistent workgroup.
typedef float f32x16 __attribute__((ext_vector_type(16)));

// STATS (round 5): the stem convolution feeds a training-mode BatchNorm (resnet.py:229-231): a lane keeps running sums of y and
// y^2 of the values it stores (8 channels: fp32 over its ~20 tiles x 2 rows x 4 columns), summed over the 16 lanes that share a
// channel at the end of the persistent loop and added to a.stats with one fp64 atomic per wave, channel and statistic — what
// bn_stats_kernel would produce with a launch and a 629 MB pass of its own.
template <int CI, bool STATS = false>
__global__ void __launch_bounds__(256, CI == 1 ? 3 : 2) conv_stem_fwd_kernel(const SmallConvArgs a) {
    constexpr int KS = 7, S = 2, PAD = 3, KW8 = 8;
    constexpr int TRW = 4, TCW = 64;                   // output tile
    constexpr int IR = (TRW - 1) * S + KS;             // 13 input rows
    constexpr int ICP = (TCW - 1) * S + KW8 + 1;       // 135: 133 real columns + the padded tap's column (+1 spare)
    constexpr int NSTEP = CI * KS * KW8 / 2;           // 84 / 28 MFMA steps (2 k's each)
    constexpr int OP = TCW + 8;                        // output-chunk row pitch: rows cl and cl + 4 (the two half-waves) on disjoint banks
    constexpr int OCH = 16 * OP;                       // floats of one output chunk (16 channels x 64 columns)
    constexpr int NPATCH = CI * IR * ICP;
    constexpr int PATCH = ((NPATCH > 4 * OCH ? NPATCH : 4 * OCH) + 63) & ~63;
    constexpr int NIT = (NPATCH + 255) / 256;          // staging loads per lane and tile
    // Two patch buffers: the next tile's patch is fetched by direct global->LDS loads (global_load_lds_dword: no VGPR
    // staging, nothing waits on them until the end of the iteration) while this tile's MFMAs and stores run.
    __shared__ __attribute__((aligned(16))) float patch2[2][PATCH];

    const int lane = threadIdx.x & 63, l31 = lane & 31, khalf = lane >> 5;
    const int wave = threadIdx.x >> 6;
    const int mi = wave & 1, rp = wave >> 1;           // the wave's 32-channel tile and its pair of output rows
    const int tw = (a.Wo + TCW - 1) / TCW, th = (a.Ho + TRW - 1) / TRW;
    const int HW = a.H * a.W;
    const int tiles = a.N * th * tw;

    // persistent workgroups: the wave's weight operand (channel mi*32 + l31, k = 2*step + khalf) stays in registers
    float areg[NSTEP];
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
        const int ci = st / (KS * KW8 / 2), rem = st - ci * (KS * KW8 / 2);
        const int kh = rem / (KW8 / 2), kp = rem - kh * (KW8 / 2);
        const int kw = 2 * kp + khalf;
        areg[st] = kw < KS ? a.wp[((size_t)(kh * KS + kw) * a.CiR + ci) * a.CoP + mi * 32 + l31] : 0.f;
    }
    const int pb_off = (S * 2 * rp) * ICP + S * l31 + khalf;            // row 0, pixel tile 0 of the wave
    const bool vec_out = (a.Wo & 3) == 0;

    // element e = it*256 + tid of the flat [CI][IR][ICP] patch: a wave instruction fills 64 consecutive floats of LDS.
    // Lanes outside the image (the conv's zero padding) or past the patch issue no load: their slot gets a plain
    // ds_write of 0 instead (a masked lane of global_load_lds leaves its LDS slot untouched).
    auto fetch_patch = [&](int tile, float* dst) __attribute__((always_inline)) {
        int b = tile;
        const int n = b / (th * tw);
        b -= n * th * tw;
        const int ih0 = (b / tw) * TRW * S - PAD, iw0 = (b - (b / tw) * tw) * TCW * S - PAD;
        const float* xn = a.x + (size_t)n * CI * HW;
#pragma unroll 1
        for (int it = 0; it < NIT; ++it) {          // (rolled: the addresses are cheap, hoisting all of them spills)
            const int e = it * 256 + threadIdx.x;
            const int ci = min(e / (IR * ICP), CI - 1), rem = e - ci * (IR * ICP);
            const int rr = rem / ICP, cc = rem - rr * ICP;
            const int ih = ih0 + rr, iw = iw0 + cc;
            const bool inside = e < NPATCH;
            const bool ok = inside && ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
            float* slot = dst + it * 256 + (threadIdx.x & ~63);         // wave-uniform LDS base of this instruction
            if (ok)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(xn + (size_t)ci * HW + (size_t)ih * a.W + iw),
                                                 (__attribute__((address_space(3))) void*)slot, 4, 0, 0);
            else if (inside)
                dst[e] = 0.f;
        }
    };

    float st1[2][4], st2[2][4];                     // STATS: [q][ps] sums of channel mi*32 + 16q + 4ps + (lane >> 4)
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) st1[q][ps] = st2[q][ps] = 0.f;

    int buf = 0;
    if ((int)blockIdx.x < tiles) fetch_patch(blockIdx.x, patch2[0]);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x, buf ^= 1) {
        int b = tile;
        const int n = b / (th * tw);
        b -= n * th * tw;
        const int oh0 = (b / tw) * TRW, ow0 = (b - (b / tw) * tw) * TCW;
        const float* patch = patch2[buf];
        const float* pb = patch + pb_off;
        float* const ostage = patch2[buf] + wave * OCH;                 // the wave's output chunk (this patch is dead then)
        const int next = tile + (int)gridDim.x;
        if (next < tiles) fetch_patch(next, patch2[buf ^ 1]);           // in flight under the MFMAs and the stores below

        f32x16 acc[2][2];                                               // [row of the pair][pixel tile]
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
        for (int ci = 0; ci < CI; ++ci)
#pragma unroll
            for (int kh = 0; kh < KS; ++kh)
#pragma unroll
                for (int kp = 0; kp < KW8 / 2; ++kp) {
                    const int boff = (ci * IR + kh) * ICP + 2 * kp;
                    const float av = areg[(ci * KS + kh) * (KW8 / 2) + kp];
                    const float b00 = pb[boff], b01 = pb[boff + S * 32];
                    const float b10 = pb[boff + S * ICP], b11 = pb[boff + S * ICP + S * 32];
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b00, acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b01, acc[0][1], 0, 0, 0);
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b10, acc[1][0], 0, 0, 0);
                    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b11, acc[1][1], 0, 0, 0);
                }
        __syncthreads();                                                // every wave is done reading this patch

        // Epilogue through LDS: the accumulator layout gives 128-byte runs per half-wave (dword stores, ~2 TB/s
        // measured on this 629 MB output); transposed, a lane stores 16 bytes and a wave 4 x 256 contiguous bytes.
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                // chunk = channels mi*32 + 16q + [0,16) of output row rr: accumulator elements j in [8q, 8q+8),
                // channel-in-chunk = (j & 3) + 8 * ((j >> 2) & 1) + 4 * khalf
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                    const int j = 8 * q + jj;
                    const int cl = (jj & 3) + 8 * (jj >> 2) + 4 * khalf;
                    const int c = mi * 32 + 16 * q + cl;
                    const float sc = a.scale ? a.scale[c] : 1.f, sh = a.shift ? a.shift[c] : 0.f;
                    ostage[cl * OP + l31] = small_act(fmaf(acc[rr][0][j], sc, sh), a.act);
                    ostage[cl * OP + 32 + l31] = small_act(fmaf(acc[rr][1][j], sc, sh), a.act);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const int oh = oh0 + 2 * rp + rr;
                if (oh < a.Ho) {
#pragma unroll
                    for (int ps = 0; ps < 4; ++ps) {
                        const int cl = ps * 4 + (lane >> 4), col = (lane & 15) * 4;
                        const int c = mi * 32 + 16 * q + cl;
                        const float4 v = *reinterpret_cast<const float4*>(ostage + cl * OP + col);
                        float* yr = a.y + (((size_t)n * a.Co + c) * a.Ho + oh) * a.Wo + ow0 + col;
                        if (vec_out && ow0 + col + 3 < a.Wo) {
                            *reinterpret_cast<float4*>(yr) = v;
                            if constexpr (STATS) {
                                st1[q][ps] += (v.x + v.y) + (v.z + v.w);
                                st2[q][ps] += fmaf(v.x, v.x, v.y * v.y) + fmaf(v.z, v.z, v.w * v.w);
                            }
                        } else {
                            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (ow0 + col + k < a.Wo) {
                                    yr[k] = e[k];
                                    if constexpr (STATS) {
                                        st1[q][ps] += e[k];
                                        st2[q][ps] = fmaf(e[k], e[k], st2[q][ps]);
                                    }
                                }
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        __builtin_amdgcn_s_waitcnt(0);                                  // the next patch has landed (this wave's part)
        __syncthreads();                                                // ... everyone's; this patch / output staging is free
    }
    if constexpr (STATS) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                float s1 = st1[q][ps], s2 = st2[q][ps];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {                      // the 16 lanes of a row share the channel
                    s1 += __shfl_xor(s1, o, 64);
                    s2 += __shfl_xor(s2, o, 64);
                }
                if ((lane & 15) == 0) {
                    const int c = mi * 32 + 16 * q + 4 * ps + (lane >> 4);
                    atomicAdd(a.stats + c, (double)s1);
                    atomicAdd(a.stats + a.Co + c, (double)s2);
                }
            }
    }
}

// ---- stem weight gradient: the forward's LDS-patch scheme with the pixel axis as the MFMA reduction -------------
// dW[co][k] = sum_pix dY[co][pix] * X[k][pix], k = (ci, kh, kw8).  A = dY (lane: channel l31, pixel parity khalf) from
// an LDS tile [co][pixel] (odd pitch), B = X (lane: k-row l31, pixel parity khalf) = patch[koff(k) + pixel offset]: each
// lane keeps the patch offset of its k-rows, the pixel part is a compile-time immediate.  A workgroup owns 2 output
// rows x 64 columns of one image per step and is persistent; wave w accumulates channel tile w & 1 x its share of the
// k-tiles (3 of 6 for Ci = 3, 1 of 2 for Ci = 1) over ALL of its tiles in registers and writes one partial slab;
// stem_wgrad_finish sums the slabs in a fixed order and drops the padded tap column.
template <int CI>
__global__ void __launch_bounds__(256, CI == 1 ? 4 : 2) conv_stem_wgrad_kernel(const StemWgradArgs a) {
    constexpr int KS = 7, S = 2, PAD = 3, KW8 = 8;
    constexpr int TRW = 2, TCW = 64, NPX = TRW * TCW;         // pixel tile
    constexpr int IR = (TRW - 1) * S + KS;                    // 9 input rows
    constexpr int ICP = (TCW - 1) * S + KW8 + 1;              // 135
    constexpr int KT = CI * KS * KW8;                         // 168 / 56 real k-rows
    constexpr int NKT = (KT + 31) / 32;                       // 6 / 2 k-tiles
    constexpr int KPW = NKT / 2;                              // k-tiles per wave (3 / 1)
    constexpr int PP = NPX + 1;                               // dY tile pitch (odd: conflict-free channel-strided reads)
    __shared__ float patch[CI * IR * ICP];
    __shared__ float dyt[64 * PP];

    const int lane = threadIdx.x & 63, l31 = lane & 31, khalf = lane >> 5;
    const int wave = threadIdx.x >> 6;
    const int cot = wave & 1, kt0 = (wave >> 1) * KPW;
    const int tw = (a.Wo + TCW - 1) / TCW, th = (a.Ho + TRW - 1) / TRW;
    const int HW = a.H * a.W, HoWo = a.Ho * a.Wo;
    const int tiles = a.N * th * tw;

    int boff[KPW];                                            // patch offset of the lane's k-row in each of its k-tiles
#pragma unroll
    for (int j = 0; j < KPW; ++j) {
        int k = (kt0 + j) * 32 + l31;
        if (k >= KT) k = 0;                                   // padding rows of the last k-tile: any mapped address
        const int ci = k / (KS * KW8), rem = k - ci * (KS * KW8);
        boff[j] = (ci * IR + (rem >> 3)) * ICP + (rem & 7) + S * khalf;
    }
    const float* ab = dyt + (cot * 32 + l31) * PP + khalf;
    f32x16 acc[KPW];
#pragma unroll
    for (int j = 0; j < KPW; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
        int b = tile;
        const int n = b / (th * tw);
        b -= n * th * tw;
        const int oh0 = (b / tw) * TRW, ow0 = (b - (b / tw) * tw) * TCW;
        const int ih0 = oh0 * S - PAD, iw0 = ow0 * S - PAD;
        __syncthreads();                                      // previous tile's MFMA phase is done with both buffers
        {
            constexpr int NIT = (CI * IR * ICP + 255) / 256;
            float sv[NIT];
            const float* xn = a.x + (size_t)n * CI * HW;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int e = it * 256 + threadIdx.x;
                const int ci = min(e / (IR * ICP), CI - 1), rem = e - ci * (IR * ICP);
                const int rr = rem / ICP, cc = rem - rr * ICP;
                const int ih = ih0 + rr, iw = iw0 + cc;
                const bool ok = ih >= 0 && ih < a.H && iw >= 0 && iw < a.W;
                const float v = xn[(size_t)ci * HW + (size_t)min(max(ih, 0), a.H - 1) * a.W + min(max(iw, 0), a.W - 1)];
                sv[it] = ok ? v : 0.f;
            }
            // dY tile: 64 channels x (2 rows x 64 cols), 16-byte loads where the row allows it
            float4 dv[8];
            const float* dn = a.dy + (size_t)n * 64 * HoWo;
            const bool vec = (a.Wo & 3) == 0;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int i4 = it * 256 + threadIdx.x;
                const int co = i4 >> 5, q = i4 & 31, r = q >> 4, c4 = (q & 15) * 4;
                const int oh = oh0 + r, ow = ow0 + c4;
                const float* src = dn + (size_t)co * HoWo + (size_t)min(oh, a.Ho - 1) * a.Wo;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (oh < a.Ho) {
                    if (vec && ow + 3 < a.Wo) v = *reinterpret_cast<const float4*>(src + ow);
                    else {
                        if (ow < a.Wo) v.x = src[ow];
                        if (ow + 1 < a.Wo) v.y = src[ow + 1];
                        if (ow + 2 < a.Wo) v.z = src[ow + 2];
                        if (ow + 3 < a.Wo) v.w = src[ow + 3];
                    }
                }
                dv[it] = v;
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int e = it * 256 + threadIdx.x;
                if (e < CI * IR * ICP) patch[e] = sv[it];
            }
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int i4 = it * 256 + threadIdx.x;
                const int co = i4 >> 5, q = i4 & 31;
                float* dst = dyt + co * PP + q * 4;
                dst[0] = dv[it].x; dst[1] = dv[it].y; dst[2] = dv[it].z; dst[3] = dv[it].w;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ps = 0; ps < NPX / 2; ++ps) {                // pixel pair (2ps, 2ps+1): same output row
            const int row = (2 * ps) / TCW, col = (2 * ps) % TCW;
            const int poff = S * row * ICP + S * col;
            const float av = ab[2 * ps];
#pragma unroll
            for (int j = 0; j < KPW; ++j) {
                const float bv = patch[boff[j] + poff];
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[j], 0, 0, 0);
            }
        }
    }
    // partial slab of this workgroup: slab[co][k], k < NKT*32
    float* slab = a.slabs + (size_t)blockIdx.x * 64 * (NKT * 32);
#pragma unroll
    for (int j = 0; j < KPW; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int co = cot * 32 + (e & 3) + 8 * (e >> 2) + 4 * khalf;
            slab[(size_t)co * (NKT * 32) + (kt0 + j) * 32 + l31] = acc[j][e];
        }
}

// dw[co][ci][kh][kw] = sum over slabs (ascending) of slab[co][(ci*7 + kh)*8 + kw]
__global__ void __launch_bounds__(256) stem_wgrad_finish_kernel(const float* __restrict__ slabs, int nslabs, int kpad,
                                                                int CI, float* __restrict__ dw) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int per_co = CI * 49;
    if (i >= 64 * per_co) return;
    const int co = i / per_co, rem = i - co * per_co;
    const int ci = rem / 49, t = rem - ci * 49;
    const int k = (ci * 7 + t / 7) * 8 + t % 7;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const size_t stride = (size_t)64 * kpad;
    const float* p = slabs + (size_t)co * kpad + k;
    int sl = 0;
    for (; sl + 3 < nslabs; sl += 4) {
        s0 += p[(size_t)sl * stride];
        s1 += p[(size_t)(sl + 1) * stride];
        s2 += p[(size_t)(sl + 2) * stride];
        s3 += p[(size_t)(sl + 3) * stride];
    }
    for (; sl < nslabs; ++sl) s0 += p[(size_t)sl * stride];
    dw[i] = (s0 + s1) + (s2 + s3);
}

static int stem_wgrad_grid(long tiles, int Ci) {
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    const long cap = (long)cus * (Ci == 1 ? 4 : 2);        // resident workgroups per CU (registers: 112 / 210 VGPRs)
    return (int)(tiles < cap ? tiles : cap);
}

bool stem_conv_wgrad_eligible(int Ci, int Co, int KH, int KW, int SH, int SW, int PH, int PW, bool has_x2, bool has_bias) {
    if (has_x2 || has_bias) return false;
    if (KH != 7 || KW != 7 || SH != 2 || SW != 2 || PH != 3 || PW != 3) return false;
    return (Ci == 1 || Ci == 3) && Co == 64;
}

size_t stem_conv_wgrad_workspace_bytes(int N, int Ci, int Ho, int Wo) {
    const long tiles = (long)N * ceil_div(Ho, 2) * ceil_div(Wo, 64);
    const int kpad = ((Ci * 56 + 31) / 32) * 32;
    return sizeof(float) * (size_t)stem_wgrad_grid(tiles, Ci) * 64 * kpad;
}

int launch_stem_conv_wgrad(const float* x, const float* dy, float* dw, float* workspace, int N, int Ci, int H, int W,
                           int Ho, int Wo, hipStream_t st) {
    const long tiles = (long)N * ceil_div(Ho, 2) * ceil_div(Wo, 64);
    if (tiles > 0x7fffffffL) return DYNMM_EUNSUPPORTED;
    const int grid = stem_wgrad_grid(tiles, Ci);
    const int kpad = ((Ci * 56 + 31) / 32) * 32;
    StemWgradArgs a{x, dy, workspace, N, H, W, Ho, Wo};
    if (Ci == 3)
        hipLaunchKernelGGL((conv_stem_wgrad_kernel<3>), dim3(grid), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((conv_stem_wgrad_kernel<1>), dim3(grid), dim3(256), 0, st, a);
    DYNMM_LAUNCH_CHECK();
    hipLaunchKernelGGL(stem_wgrad_finish_kernel, dim3(ceil_div(64 * Ci * 49, 256)), dim3(256), 0, st, workspace, grid, kpad,
                       Ci, dw);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

bool stem_conv_fwd_eligible(const SmallConvArgs& a, const float* residual) {
    if (residual || a.x2) return false;
    if (a.KH != 7 || a.KW != 7 || a.SH != 2 || a.SW != 2 || a.PH != 3 || a.PW != 3) return false;
    return (a.Ci == 1 || a.Ci == 3) && a.Co == 64;
}

int launch_stem_conv_fwd(const SmallConvArgs& a_in, hipStream_t st) {
    const long tiles = (long)a_in.N * ceil_div(a_in.Ho, 4) * ceil_div(a_in.Wo, 64);
    if (tiles > 0x7fffffffL) return DYNMM_EUNSUPPORTED;
    const SmallConvArgs& a = a_in;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    const int resident = a.Ci == 1 ? 3 : 2;             // workgroups per CU = waves per SIMD (register-bound)
    const long cap = (long)cus * resident;
    const unsigned grid = (unsigned)(tiles < cap ? tiles : cap);
    if (a.stats) {
        if (a.act != DYNMM_ACT_NONE) return DYNMM_EUNSUPPORTED;          // statistics of the convolution's own output
        if (a.Ci == 3)
            hipLaunchKernelGGL((conv_stem_fwd_kernel<3, true>), dim3(grid), dim3(256), 0, st, a);
        else
            hipLaunchKernelGGL((conv_stem_fwd_kernel<1, true>), dim3(grid), dim3(256), 0, st, a);
    } else if (a.Ci == 3)
        hipLaunchKernelGGL((conv_stem_fwd_kernel<3>), dim3(grid), dim3(256), 0, st, a);
    else
        hipLaunchKernelGGL((conv_stem_fwd_kernel<1>), dim3(grid), dim3(256), 0, st, a);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

bool small_conv_fwd_eligible(const SmallConvArgs& a, const float* residual) {
    if (residual) return false;
    if (a.Co < 5 || a.Co > 8) return false;                       // packed rows of exactly 8 floats
    if (a.KH != 5 || a.KW != 5 || a.SH != 2 || a.SW != 2 || a.PH != 0 || a.PW != 0) return false;
    if (a.Ci % 4 != 0 || a.Ci < 16) return false;
    if (a.c_split < a.Ci && a.c_split % (a.Ci / 4) != 0) return false;      // a wave's channels come from one tensor
    return true;
}

int launch_small_conv_fwd(const SmallConvArgs& a, hipStream_t st) {
    const long tiles = (long)a.N * ceil_div(a.Ho, kSRows) * ceil_div(a.Wo, kSCols);
    if (tiles > 0x7fffffffL) return DYNMM_EUNSUPPORTED;
    hipLaunchKernelGGL((conv_co8_fwd_kernel<5, 2>), dim3((unsigned)tiles), dim3(256), 0, st, a);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

}  // namespace dynmm
