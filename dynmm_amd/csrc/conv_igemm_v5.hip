// Implicit-GEMM convolution, round-3 operand pipeline ("v5"): stride-1, same-padded 1x1 / 3x1 / 1x3 / 3x3 convolutions
// (forward and input gradient) on the fp32 matrix cores, every workgroup self-sufficient in latency hiding.
//
// Why (measured, scratch/trace + scratch/mfma/peak_random.hip, docs/DESIGN_history_r1-r4.md §4 "round 3"): with 6 thin workgroups per CU the
// register-staged kernel of conv_igemm.hip saturates the MFMA pipe only while all 6 are resident; the workgroups of a CU
// finish one after the other (the oldest wave wins the pipe), and whatever runs at reduced occupancy — the staggered tail
// of every launch, i.e. 25-45 % of its duration — exposes the full load -> LDS -> fragment latency chain of each K-step.
// Here each wave hides its own latencies instead of relying on five neighbours:
//   * operand tiles go global -> LDS by `global_load_lds_dwordx4` (no VGPR staging, no ds_write, no vmcnt stall in the
//     loop): a 4-stage ring for the weight tiles, a 2- or 4-stage ring for the activation tiles, filled three K-steps
//     ahead of their use, counted by hand (`s_waitcnt vmcnt(N)`; the loads are inline assembly because hipcc drains
//     vmcnt to 0 in front of every LDS read that follows a direct-to-LDS load it knows about);
//   * the MFMA fragments of step t+1 are read from LDS into a second register set under the 16 MFMAs of step t;
//   * one barrier per K-step; 16-byte, always aligned activation loads: for the 1x3 / 3x3 taps the tile is staged ONCE
//     per vertical tap and channel chunk with a 4-pixel halo on both sides and the three horizontal taps read it at
//     offsets -1 / 0 / +1 (3x fewer activation bytes through L2 than one gather per tap);
//   * zero padding is applied at the fragment READ: a lane whose pixel has no input under the current tap (per-lane
//     tap-validity bits) points its eight reads of the step at an all-zero LDS slot instead — one address select per
//     step, no arithmetic on the data, and the loads never need a predicate: an out-of-image row is replaced by the
//     un-shifted (always mapped) row, its values are never read.
// Accumulator / epilogue layout is the one of conv_igemm.hip (lane&31 = pixel: 128-byte store runs in NCHW).
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "conv_igemm.h"

namespace dynmm {

template <int I>
using ic = std::integral_constant<int, I>;

template <int TCO, int TPIX, int WCO, int WPIX, int KW, bool DGRAD, int SA, int SB>
__global__ void __launch_bounds__(256, 3) conv_igemm_v5_kernel(const IgemmArgs a) {
    // ring depths: SA weight stages; SB activation stages (KW = 3: two stages of three K-steps each; KW = 1: one per step)
    constexpr int BK = 16, HALO = (KW == 3 ? 4 : 0);
    static_assert((SA == 3 || SA == 4) && (KW == 3 ? SB == 2 : (SB == 3 || SB == 4)), "ring depths");
    constexpr int PIXW = TPIX + 2 * HALO;
    constexpr int MCO = WCO / 32, MPIX = WPIX / 32, WAVES_PIX = TPIX / WPIX;
    static_assert((TCO / WCO) * WAVES_PIX == 4, "4 waves per workgroup");
    constexpr int A_STAGE = BK * TCO, B_STAGE = BK * PIXW;          // floats per ring slot
    constexpr int AQ = TCO / 4;                                     // 16-byte quads per weight row of the tile
    constexpr int RPI = 64 / AQ;                                    // weight rows per wave instruction
    constexpr int NIA = BK / RPI / 4;                               // instructions per wave and weight stage
    constexpr int QPR = PIXW / 4, QB = BK * QPR, QPW = QB / 4;      // activation quads: per row / stage / wave
    constexpr int NIB = (QPW + 63) / 64;                            // instructions per wave and activation stage
    static_assert(NIA >= 1 && BK % (RPI * 4) == 0 && QB % 4 == 0, "tile shape");
    static_assert(NIB * 3 <= 30, "row-validity bits of the loader fit one register");

    __shared__ __attribute__((aligned(16))) float As[SA * A_STAGE];
    __shared__ __attribute__((aligned(16))) float Bs[SB * B_STAGE];
    __shared__ __attribute__((aligned(16))) float Zs[B_STAGE];      // zeros: what a padded tap reads

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    DYNMM_TRACE_MARK(0);
#ifdef DYNMM_TRACE
    if (g_trace && threadIdx.x == 0) g_trace[(size_t)blockIdx.x * 6 + 4] = clock64();
#endif
    const int wave_co = wave / WAVES_PIX, wave_pix = wave % WAVES_PIX;
    const int khalf = lane >> 5, l31 = lane & 31;

    const int nblk = a.n_co_tiles * a.n_pix_tiles;
    const int lin = xcd_remap((int)blockIdx.x, nblk);
    const int co0 = (lin % a.n_co_tiles) * TCO;
    const int pix0 = (lin / a.n_co_tiles) * TPIX;
    const int HW = a.H * a.W;
    const int NC = a.Ci / BK;                    // channel chunks
    const int nB = a.KH * NC;                    // activation stages (vertical tap, chunk)
    const int nsteps = nB * KW;                  // K-steps (one weight stage each)

    // vertical / horizontal offset of tap index r / s (the input gradient of a stride-1 convolution is the
    // convolution of dy with the flipped filter)
    auto dh_of = [&](int r) { return DGRAD ? a.PH - r : r - a.PH; };
    constexpr int DW0 = (KW == 3) ? (DGRAD ? 1 : -1) : 0;           // tap s: dw = DW0 + s * DWS
    constexpr int DWS = (KW == 3) ? (DGRAD ? -1 : 1) : 0;

    // ---------------------------------------------------------------- loader state
    const unsigned a_voff = (unsigned)(((lane / AQ) * a.CoP + co0 + 4 * (lane % AQ)) * 4);
    unsigned b_center[NIB];
    unsigned b_rows = 0;                          // bit 3*i + r: the row of this lane's quad shifted by tap r is inside the image
    bool b_act[NIB];
#pragma unroll
    for (int i = 0; i < NIB; ++i) {
        const int ql = i * 64 + lane;
        b_act[i] = ql < QPW;
        const int q = wave * QPW + (b_act[i] ? ql : 0);
        const int k = q / QPR, quad = q - k * QPR;
        int m = pix0 - HALO + 4 * quad;
        m = m < 0 ? 0 : (m > a.M - 4 ? a.M - 4 : m);      // quads outside the tensor: any mapped address (never used)
        const int n = m / HW, rem = m - n * HW;
        const int h = rem / a.W;
        b_center[i] = ((unsigned)(n * a.Ci + k) * (unsigned)HW + (unsigned)rem) * 4u;
        for (int r = 0; r < a.KH; ++r) {
            const int hh = h + dh_of(r);
            b_rows |= (hh >= 0 && hh < a.H) ? (1u << (3 * i + r)) : 0u;
        }
    }
    const unsigned lds_a = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)As);
    const unsigned lds_b = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)Bs);
    int la_t = 0, la_r = 0, la_c = 0, la_s = 0;   // next weight stage to request: step, tap row, chunk, tap column
    int lb_t = 0, lb_r = 0, lb_c = 0;             // next activation stage to request
    auto issue_a = [&]() {
        if (la_t < nsteps) {
            const float* base = a.wp + (size_t)((la_r * KW + la_s) * a.CiR + la_c * BK + wave * NIA * RPI) * a.CoP;
            const unsigned dst = lds_a + (unsigned)(((la_t % SA) * A_STAGE + wave * NIA * 256) * 4);
#pragma unroll
            for (int i = 0; i < NIA; ++i) dma16(base + (size_t)(i * RPI) * a.CoP, a_voff, dst + i * 1024);
            ++la_t;
            if (++la_s == KW) {
                la_s = 0;
                if (++la_c == NC) { la_c = 0; ++la_r; }
            }
        }
    };
    auto issue_b = [&]() {
        if (lb_t < nB) {
            const float* base = a.x + (size_t)(lb_c * BK) * HW;
            const unsigned dst = lds_b + (unsigned)(((lb_t % SB) * B_STAGE + wave * QPW * 4) * 4);
            const int shift = dh_of(lb_r) * a.W * 4;
#pragma unroll
            for (int i = 0; i < NIB; ++i) {
                const unsigned voff = b_center[i] + (((b_rows >> (3 * i + lb_r)) & 1u) ? (unsigned)shift : 0u);
                if (b_act[i]) dma16(base, voff, dst + i * 1024);
            }
            ++lb_t;
            if (++lb_c == NC) { lb_c = 0; ++lb_r; }
        }
    };

    // ---------------------------------------------------------------- consumer state
    const int a_frag = khalf * TCO + wave_co * WCO + l31;                 // + 2q*TCO + mi*32
    const int b_frag = khalf * PIXW + HALO + wave_pix * WPIX + l31;       // + 2q*PIXW + ni*32 + dw
    unsigned pmask[MPIX];                        // bit r*KW + s: tap (r, s) of this lane's output pixel reads inside the image
#pragma unroll
    for (int ni = 0; ni < MPIX; ++ni) {
        const int m = pix0 + wave_pix * WPIX + ni * 32 + l31;
        unsigned bits = 0;
        if (m < a.M) {
            const int rem = m % HW;
            const int h = rem / a.W, w = rem - h * a.W;
            for (int r = 0; r < a.KH; ++r) {
                const int hh = h + dh_of(r);
                for (int s = 0; s < KW; ++s) {
                    const int ww = w + DW0 + s * DWS;
                    bits |= (hh >= 0 && hh < a.H && ww >= 0 && ww < a.W) ? (1u << (r * KW + s)) : 0u;
                }
            }
        }
        pmask[ni] = bits;
    }

    f32x16 acc[MCO][MPIX];
#pragma unroll
    for (int mi = 0; mi < MCO; ++mi)
#pragma unroll
        for (int ni = 0; ni < MPIX; ++ni)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[mi][ni][j] = 0.f;

    float fa[2][BK / 2][MCO], fb[2][BK / 2][MPIX];

    // fragments of step `tn` (tap row rn, tap column S) -> register set SET
    auto read_frags = [&](auto set_c, auto s_c, int tn, int btn, int rn) {
        constexpr int SET = decltype(set_c)::value, S = decltype(s_c)::value;
        const float* Ap = As + (tn % SA) * A_STAGE + a_frag;
        const float* Bn = Bs + (btn % SB) * B_STAGE + b_frag + (DW0 + S * DWS);
        const int bit = rn * KW + S;
        const float* Bp[MPIX];
#pragma unroll
        for (int ni = 0; ni < MPIX; ++ni) Bp[ni] = ((pmask[ni] >> bit) & 1u) ? Bn + ni * 32 : Zs + b_frag;
#pragma unroll
        for (int q = 0; q < BK / 2; ++q) {
#pragma unroll
            for (int mi = 0; mi < MCO; ++mi) fa[SET][q][mi] = Ap[2 * q * TCO + mi * 32];
#pragma unroll
            for (int ni = 0; ni < MPIX; ++ni) fb[SET][q][ni] = Bp[ni][2 * q * PIXW];
        }
    };
    auto mfma_set = [&](auto set_c) {
        constexpr int SET = decltype(set_c)::value;
#pragma unroll
        for (int q = 0; q < BK / 2; ++q)
#pragma unroll
            for (int mi = 0; mi < MCO; ++mi)
#pragma unroll
                for (int ni = 0; ni < MPIX; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[SET][q][mi], fb[SET][q][ni], acc[mi][ni], 0, 0, 0);
    };

    for (int i = t; i < B_STAGE; i += 256) Zs[i] = 0.f;

    // ---------------------------------------------------------------- prologue: fill the rings
    // Request order = the steady state's (weight stage t at "step" t - SA, activation stage b at the last step of stage
    // b - SB, after that step's weight request), so the loop's wait counts hold from its first iteration on.
    // (weight stage j belongs to "step" j - SA, activation stage j to step j - SB (KW = 1) or 3j - 4 (KW = 3); within a
    // step the weight request comes first)
    constexpr int NEWER_A = (SA - 2) * NIA;
    constexpr int NEWER_B1 = SB > SA ? (SA - 1) * NIB : ((SA < SB ? SA : SB) - 2) * NIB;     // KW = 1
    constexpr int NEWER_A1 = SB < SA ? (SB - 2) * NIA : NEWER_A;                               // KW = 1
    if constexpr (KW == 3) {
        if constexpr (SA == 4) { issue_a(); issue_b(); issue_a(); issue_a(); issue_a(); issue_b(); }     // A0 B0 | A1 | A2 | A3 B1
        else { issue_b(); issue_a(); issue_a(); issue_a(); issue_b(); }                                // B0 | A0 | A1 | A2 B1
        wait_vm<(SA - 1) * NIA + NIB>();
    } else if constexpr (SA == SB) {
        for (int j = 0; j < SA; ++j) { issue_a(); issue_b(); }                                         // A0 B0 | A1 B1 | ...
        wait_vm<(SA - 1) * (NIA + NIB)>();
    } else if constexpr (SB > SA) {
        issue_b(); issue_a(); issue_b(); issue_a(); issue_b(); issue_a(); issue_b();                   // B0 | A0 B1 | A1 B2 | A2 B3
        wait_vm<2 * NIA + 3 * NIB>();
    } else {
        issue_a(); issue_a(); issue_b(); issue_a(); issue_b(); issue_a(); issue_b();                   // A0 | A1 B0 | A2 B1 | A3 B2
        wait_vm<2 * (NIA + NIB)>();
    }
    __syncthreads();
    read_frags(ic<0>{}, ic<0>{}, 0, 0, 0);
    DYNMM_TRACE_MARK(1);

    // ---------------------------------------------------------------- K loop
    // step tcur: [wait: stage tcur+1 has landed] barrier [request the stages that reuse the slots just released]
    //            [read fragments of step tcur+1 into the other register set] || [MFMAs of step tcur]
    int cr = 0, cc = 0;                           // tap row / chunk of the activation stage being consumed
    int tcur = 0;
    auto step = [&](auto set_c, auto s_c, int bt) {
        constexpr int SET = decltype(set_c)::value, S = decltype(s_c)::value;
        // loads newer than the ones step tcur+1 needs: weight stages tcur+2, tcur+3 and the activation stage(s)
        // requested together with them
        if (tcur + SA >= nsteps) {
            wait_vm<0>();
        } else if constexpr (KW == 3) {
            if (tcur + 8 >= nsteps) wait_vm<NEWER_A>();     // the activation requests stop two stages before the weight requests do
            else wait_vm<NEWER_A + (S == 2 ? 0 : NIB)>();
        } else {
            wait_vm<NEWER_A1 + NEWER_B1>();
        }
        __syncthreads();
        issue_a();
        if constexpr (KW == 1 || S == KW - 1) issue_b();
        // (unconditional: after the last step this reads a dead slot — branch-free, so the reads and the MFMAs below
        // stay in one scheduling region)
        if constexpr (S == KW - 1) {
            const int rn = (cc + 1 == NC) ? cr + 1 : cr;
            read_frags(ic<SET ^ 1>{}, ic<0>{}, tcur + 1, bt + 1, rn);
        } else {
            read_frags(ic<SET ^ 1>{}, ic<S + 1>{}, tcur + 1, bt, cr);
        }
        mfma_set(set_c);
#pragma unroll
        for (int q = 0; q < BK / 2; ++q) {      // spread the LDS reads / selects of the next step between the MFMAs
            __builtin_amdgcn_sched_group_barrier(0x008, MCO * MPIX, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
        ++tcur;
        if constexpr (S == KW - 1) {
            if (++cc == NC) { cc = 0; ++cr; }
        }
    };
    for (int bt = 0; bt < nB; bt += 2) {          // two activation stages per iteration: the register sets alternate per step
        if constexpr (KW == 3) {
            step(ic<0>{}, ic<0>{}, bt);
            step(ic<1>{}, ic<1>{}, bt);
            step(ic<0>{}, ic<2>{}, bt);
            step(ic<1>{}, ic<0>{}, bt + 1);
            step(ic<0>{}, ic<1>{}, bt + 1);
            step(ic<1>{}, ic<2>{}, bt + 1);
        } else {
            step(ic<0>{}, ic<0>{}, bt);
            step(ic<1>{}, ic<0>{}, bt + 1);
        }
    }
    __syncthreads();                               // every wave is done with the operand rings: As is reused below
    DYNMM_TRACE_MARK(2);

    // ---------------------------------------------------------------- epilogue (as conv_igemm.hip)
    // scale/shift (bias or folded BN), residual, activation, ReLU-mask; NCHW store.  No memory wait sits between two
    // stores: per-channel scale/shift come from LDS, residual / mask values of half an accumulator tile are loaded as one
    // batch, then its 8 stores issue back to back.
    const int HoWo = HW;
    float* const sc_lds = As;
    float* const sh_lds = As + TCO;
    {
        const float* __restrict__ scale = a.scale;
        const float* __restrict__ shift = a.shift;
        for (int i = t; i < TCO; i += 256) {
            const int co = co0 + i;
            sc_lds[i] = scale ? scale[co] : 1.f;
            sh_lds[i] = shift ? shift[co] : 0.f;
        }
    }
    __syncthreads();
    const float* __restrict__ res_p = a.residual;
    const float* __restrict__ mask_p = a.mask;
    float* __restrict__ y1_p = a.y;
    const bool has_res = res_p != nullptr, has_mask = mask_p != nullptr;
    const int act = a.act;
    const unsigned row_bytes = (unsigned)HoWo * 4u;

    auto finish = [&](float (&v)[8], const float (&rv)[8], const float (&mv)[8]) {
        if (DGRAD) {
            if (has_mask) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = mv[j] > 0.f ? v[j] : 0.f;
            }
            if (has_res) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += rv[j];
            }
        } else {
            if (has_res) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] += rv[j];
            }
            if (act == DYNMM_ACT_RELU) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
            } else if (act == DYNMM_ACT_TANH) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = tanhf(v[j]);
            }
        }
    };
    auto scaled = [&](float (&v)[8], int mi, int ni, int cl0, int h) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int j4 = 2 * h + q;
            const float4 sc = *reinterpret_cast<const float4*>(&sc_lds[cl0 + 8 * j4]);
            const float4 sh = *reinterpret_cast<const float4*>(&sh_lds[cl0 + 8 * j4]);
            v[4 * q + 0] = acc[mi][ni][4 * j4 + 0] * sc.x + sh.x;
            v[4 * q + 1] = acc[mi][ni][4 * j4 + 1] * sc.y + sh.y;
            v[4 * q + 2] = acc[mi][ni][4 * j4 + 2] * sc.z + sh.z;
            v[4 * q + 3] = acc[mi][ni][4 * j4 + 3] * sc.w + sh.w;
        }
    };
#pragma unroll
    for (int ni = 0; ni < MPIX; ++ni) {
        const int m = pix0 + wave_pix * WPIX + ni * 32 + l31;
        const bool valid = m < a.M;
        if (!valid) continue;
        const unsigned n = (unsigned)(m / HoWo);
        const unsigned rem = (unsigned)m - n * (unsigned)HoWo;
#pragma unroll
        for (int mi = 0; mi < MCO; ++mi) {
            const int cl0 = wave_co * WCO + mi * 32 + 4 * khalf;          // tile-local channel of j = 0
            const unsigned off0 = ((n * (unsigned)a.Co + (unsigned)(co0 + cl0)) * (unsigned)HoWo + rem) * 4u;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float v[8], rv[8], mv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { mv[e] = 1.f; rv[e] = 0.f; v[e] = 0.f; }
                if (has_mask) {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        mv[e] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(mask_p) +
                                                                (off0 + (unsigned)((e & 3) + 8 * (2 * h + (e >> 2))) * row_bytes));
                }
                if (has_res) {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        rv[e] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(res_p) +
                                                                (off0 + (unsigned)((e & 3) + 8 * (2 * h + (e >> 2))) * row_bytes));
                }
                scaled(v, mi, ni, cl0, h);
                finish(v, rv, mv);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    *reinterpret_cast<float*>(reinterpret_cast<char*>(y1_p) +
                                              (off0 + (unsigned)((e & 3) + 8 * (2 * h + (e >> 2))) * row_bytes)) = v[e];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#ifdef DYNMM_TRACE
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    DYNMM_TRACE_MARK(3);
    if (g_trace && threadIdx.x == 0) g_trace[(size_t)blockIdx.x * 6 + 5] = clock64();
#endif
}

// -1 (default): on; 0 / 1: forced by dynmm_debug_set_igemm_v5 — the test hook that lets tests/test_skip_esanet.py run one pass under
// both implicit-GEMM generations (two correct fp32 summation orders) inside one process
static int g_v5_override = -1;

bool igemm_v5_eligible(const IgemmArgs& a, bool dgrad) {
    if (g_v5_override == 0) return false;
    (void)dgrad;
    if (a.x2 || a.y2) return false;                                           // one input, one output tensor
    if (a.SH != 1 || a.SW != 1) return false;
    if (!((a.KH == 1 || a.KH == 3) && (a.KW == 1 || a.KW == 3))) return false;
    if (a.PH != a.KH / 2 || a.PW != a.KW / 2 || a.H != a.Ho || a.W != a.Wo) return false;
    if (a.W % 4 != 0 || a.H < a.KH || a.W < 4) return false;
    if (a.Ci % 32 != 0 || a.Ci < 64) return false;                           // an even number (>= 4) of 16-channel chunks
    if (a.Co % 64 != 0) return false;
    if (a.c_out_split < a.Co) return false;
    if (a.N * a.H * a.W < 64) return false;
    const uintptr_t al = reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.wp);
    if (al & 15u) return false;
    if (a.residual && (reinterpret_cast<uintptr_t>(a.residual) & 3u)) return false;
    return true;
}

bool launch_igemm_v5(IgemmArgs& a, bool dgrad, hipStream_t st) {
    if (!igemm_v5_eligible(a, dgrad)) return false;
    a.CiR = a.Ci;
    a.M = a.N * a.Ho * a.Wo;
    a.K = a.KH * a.KW * a.Ci;
    a.CoP = a.Co;
    a.subpix = 0;
    // ring depths: 3 filter slots, 3 pixel slots for the 1-tap-wide kernels and 2 for the 3-tap-wide ones (the round-3 sweep
    // over DYNMM_V5_SA / _SB settled on these; the other instantiations went with the switches in round 5)
#define DYNMM_V5_GO(TCO, TPIX, WCO, WPIX, KW_, DG_) \
    hipLaunchKernelGGL((conv_igemm_v5_kernel<TCO, TPIX, WCO, WPIX, KW_, DG_, 3, KW_ == 3 ? 2 : 3>), grid, dim3(256), 0, st, a)
#define DYNMM_V5_LAUNCH(TCO, TPIX, WCO, WPIX)                                                                          \
    do {                                                                                                               \
        a.n_co_tiles = a.Co / TCO;                                                                                     \
        a.n_pix_tiles = ceil_div(a.M, TPIX);                                                                           \
        dim3 grid((unsigned)(a.n_co_tiles * a.n_pix_tiles));                                                \
        if (a.KW == 3) {                                                                                               \
            if (dgrad) DYNMM_V5_GO(TCO, TPIX, WCO, WPIX, 3, true);                                                     \
            else DYNMM_V5_GO(TCO, TPIX, WCO, WPIX, 3, false);                                                          \
        } else {                                                                                                       \
            if (dgrad) DYNMM_V5_GO(TCO, TPIX, WCO, WPIX, 1, true);                                                     \
            else DYNMM_V5_GO(TCO, TPIX, WCO, WPIX, 1, false);                                                          \
        }                                                                                                              \
    } while (0)
    if (a.Co % 128 == 0)
        DYNMM_V5_LAUNCH(128, 64, 64, 32);
    else
        DYNMM_V5_LAUNCH(64, 128, 32, 64);
#undef DYNMM_V5_GO
#undef DYNMM_V5_LAUNCH
    return true;
}

}  // namespace dynmm

extern "C" int dynmm_debug_set_igemm_v5(int mode) {
    dynmm::g_v5_override = mode;
    return 0;
}

// Geometry-only form of igemm_v5_eligible (pointer alignment aside) for profiling tools that label launches:
// 1 if dynmm_conv2d_fwd (dgrad = 0) / dynmm_conv2d_dgrad (dgrad = 1) serve this convolution with the operand-ring kernels.
extern "C" int dynmm_conv2d_uses_operand_ring(const dynmm_conv_geom* g, int dgrad) {
    if (!g || g->c_split != g->Ci) return 0;
    dynmm::IgemmArgs a{};
    a.N = g->N; a.KH = g->KH; a.KW = g->KW; a.SH = g->SH; a.SW = g->SW; a.PH = g->PH; a.PW = g->PW;
    if (dgrad) { a.Ci = g->Co; a.H = g->Ho; a.W = g->Wo; a.Co = g->Ci; a.Ho = g->H; a.Wo = g->W; }
    else { a.Ci = g->Ci; a.H = g->H; a.W = g->W; a.Co = g->Co; a.Ho = g->Ho; a.Wo = g->Wo; }
    a.c_out_split = a.Co;
    return dynmm::igemm_v5_eligible(a, dgrad != 0) ? 1 : 0;
}
