// Forward of the stride-1 same-padded 1x1 / 3x1 / 1x3 / 3x3 convolutions, round-4 structure ("v8"): the operand-ring kernel's
// arithmetic — the SAME sequence of v_mfma_f32_32x32x2_f32 per accumulator, K order (vertical tap, 16-channel chunk,
// horizontal tap, k-pair), hence bit-identical results to conv_igemm_v5.hip — with fewer, fatter synchronisation steps.
//
// Why (profiles/r04_pmc_per_kernel.md and the kernels built this round): with one barrier per 16 MFMAs the operand-ring kernel
// keeps the matrix pipe 0.63-0.67 busy whatever else is tuned; the kernels of this library order themselves by MFMAs per
// barrier — 16: 0.54-0.65, 32: 0.58-0.69, 48: 0.73-0.77.  The training forward is the one three-tap pass that cannot take
// the Winograd form (DESIGN.md section 4 "Round 4"), runs alone on the machine (no weight gradients exist yet to fill its
// gaps) and is the largest item of the step, so its direct kernel gets the structure instead:
//   * a wave owns 64 | 32 output channels x 64 pixels (4 | 2 accumulator blocks), a workgroup 128 | 64 co x 128 pixels;
//   * one ring stage = ALL the K-steps that share an activation tile: horizontal taps (1x3, 3x3): the three taps of a 16-channel
//     chunk (weights 3 x 16 rows, one halo tile) = 96 | 48 MFMAs per wave and barrier; vertical taps / 1x1: two 16-channel
//     chunks (32 weight rows, 32 activation rows) = 64 | 32;
//   * 2-slot ring (direct global -> LDS loads, hand-counted vmcnt): a stage is requested one whole stage (3-6 us of matrix work)
//     before it is consumed; the fragments of sub-step u + 1 are read under the MFMAs of sub-step u, across the stage boundary
//     too (the wait + barrier + next request sit under the last sub-step's MFMAs);
//   * zero padding by a lane-constant select on the fragment values (no all-zero LDS tile: the LDS holds two stages only).
// The packed weight operand, the epilogue (scale / shift, residual, activation) and the tile -> XCD order are the operand-ring
// kernel's; tests/test_hip_ops.py checks torch.equal against it.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "conv_igemm.h"

namespace dynmm {

template <int TCO, int MCO, int KW>
__global__ void __launch_bounds__(256, MCO == 1 ? 3 : 2) conv_igemm_v8_kernel(const IgemmArgs a) {
    constexpr int TPIX = 128, MPIX = 2, WPIX = 64, WCO = 32 * MCO;
    constexpr int WAVES_PIX = TPIX / WPIX, WAVES_CO = TCO / WCO;
    static_assert(WAVES_PIX * WAVES_CO == 4, "4 waves per workgroup");
    constexpr int NSUB = KW == 3 ? 3 : 2;                            // sub-steps (8 k-pairs each) per stage
    constexpr int HALO = KW == 3 ? 4 : 0;
    constexpr int PIXW = TPIX + 2 * HALO;
    constexpr int BROWS = KW == 3 ? 16 : 32;                         // activation rows (channels) per stage
    constexpr int A_STAGE = NSUB * 16 * TCO, B_STAGE = BROWS * PIXW; // floats
    constexpr int AQ = TCO / 4, RPI = 64 / AQ;                       // quads per weight row, rows per wave instruction
    constexpr int NIA = NSUB * 16 / RPI / 4;                         // instructions per wave and stage: weights
    constexpr int QPR = PIXW / 4, QB = BROWS * QPR, QPW = QB / 4;
    constexpr int NIB = (QPW + 63) / 64;                             //   activations
    constexpr int NI = NIA + NIB;
    static_assert((NSUB * 16) % (RPI * 4) == 0 && NI < 32, "tile shape");

    __shared__ __attribute__((aligned(16))) float As[2 * A_STAGE];
    __shared__ __attribute__((aligned(16))) float Bs[2 * B_STAGE];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wave_co = wave / WAVES_PIX, wave_pix = wave % WAVES_PIX;
    const int khalf = lane >> 5, l31 = lane & 31;

    const int nblk = a.n_co_tiles * a.n_pix_tiles;
    const int lin = xcd_remap((int)blockIdx.x, nblk);
    const int co0 = (lin % a.n_co_tiles) * TCO;
    const int pix0 = (lin / a.n_co_tiles) * TPIX;
    const int HW = a.H * a.W;
    const int NC = KW == 3 ? a.Ci / 16 : a.Ci / 32;                  // stages per vertical tap
    const int nst = a.KH * NC;
    auto dh_of = [&](int r) { return r - a.PH; };

    // ---------------------------------------------------------------- loader state
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
        m = m < 0 ? 0 : (m > a.M - 4 ? a.M - 4 : m);
        const int n = m / HW, rem = m - n * HW;
        const int h = rem / a.W;
        b_center[i] = ((unsigned)(n * a.Ci + k) * (unsigned)HW + (unsigned)rem) * 4u;
        for (int r = 0; r < a.KH; ++r) {
            const int hh = h + dh_of(r);
            b_rows |= (hh >= 0 && hh < a.H) ? (1u << (3 * i + r)) : 0u;
        }
    }
    const unsigned a_voff = (unsigned)(((lane / AQ) * a.CoP + co0 + 4 * (lane % AQ)) * 4);
    const unsigned lds_a = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)As);
    const unsigned lds_b = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) float*)Bs);
    int l_t = 0, l_r = 0, l_c = 0;                // next stage to request: index, vertical tap, chunk (pair)
    auto issue = [&]() {
        if (l_t < nst) {
            const int slot = l_t & 1;
            const unsigned adst = lds_a + (unsigned)((slot * A_STAGE) * 4);
#pragma unroll
            for (int i = 0; i < NIA; ++i) {
                const int row = (wave * NIA + i) * RPI;                                   // first row of this instruction in the stage
                const int u = row / 16, k = row % 16;
                // horizontal: sub-step u = tap column u of the 16-channel chunk; vertical: u = which half of the 32 channels
                const size_t grow = KW == 3 ? (size_t)((l_r * 3 + u) * a.CiR + l_c * 16 + k)
                                            : (size_t)(l_r * a.CiR + l_c * 32 + u * 16 + k);
                dma16(a.wp + grow * a.CoP, a_voff, adst + (unsigned)(row * TCO * 4));
            }
            const float* bbase = a.x + (size_t)(l_c * BROWS) * HW;
            const unsigned bdst = lds_b + (unsigned)((slot * B_STAGE + wave * QPW * 4) * 4);
            const int shift = dh_of(l_r) * a.W * 4;
#pragma unroll
            for (int i = 0; i < NIB; ++i) {
                const unsigned voff = b_center[i] + (((b_rows >> (3 * i + l_r)) & 1u) ? (unsigned)shift : 0u);
                if (b_act[i]) dma16(bbase, voff, bdst + (unsigned)i * 1024u);
            }
            ++l_t;
            if (++l_c == NC) { l_c = 0; ++l_r; }
        }
    };

    // ---------------------------------------------------------------- consumer state
    const int a_frag = khalf * TCO + wave_co * WCO + l31;                 // + (u * 16 + 2q) * TCO + mi * 32
    const int b_frag = khalf * PIXW + HALO + wave_pix * WPIX + l31;       // + (2q [+ 16 u]) * PIXW + ni * 32 [+ u - 1]
    unsigned pmask[MPIX];                        // bit r*3 + s: tap (r, s) of this lane's output pixel reads inside the image
#pragma unroll
    for (int ni = 0; ni < MPIX; ++ni) {
        const int m = pix0 + wave_pix * WPIX + ni * 32 + l31;
        unsigned bits = 0;
        if (m < a.M) {
            const int rem = m % HW;
            const int h = rem / a.W, w = rem - h * a.W;
            for (int r = 0; r < a.KH; ++r) {
                const int hh = h + dh_of(r);
                for (int s = 0; s < 3; ++s) {
                    const int ww = KW == 3 ? w + s - 1 : w;
                    bits |= (hh >= 0 && hh < a.H && ww >= 0 && ww < a.W) ? (1u << (r * 3 + s)) : 0u;
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

    float fa[2][8][MCO], fb[2][8][MPIX];
    // fragments of sub-step u of the stage in `slot` (vertical tap r) -> register set
    auto read_frags = [&](int set, int u, int slot, int r) {
        const float* Ap = As + slot * A_STAGE + a_frag + u * 16 * TCO;
        const float* Bp = Bs + slot * B_STAGE + b_frag + (KW == 3 ? (u - 1) : u * 16 * PIXW);
        bool ok[MPIX];
#pragma unroll
        for (int ni = 0; ni < MPIX; ++ni) ok[ni] = (pmask[ni] >> (r * 3 + (KW == 3 ? u : 0))) & 1u;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
#pragma unroll
            for (int mi = 0; mi < MCO; ++mi) fa[set][q][mi] = Ap[2 * q * TCO + mi * 32];
#pragma unroll
            for (int ni = 0; ni < MPIX; ++ni) {
                const float v = Bp[2 * q * PIXW + ni * 32];
                fb[set][q][ni] = ok[ni] ? v : 0.f;
            }
        }
    };
    auto mfma_set = [&](int set) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int mi = 0; mi < MCO; ++mi)
#pragma unroll
                for (int ni = 0; ni < MPIX; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][q][mi], fb[set][q][ni], acc[mi][ni], 0, 0, 0);
    };
    auto spread = [&]() {                         // spread the LDS reads / selects of the next sub-step between the MFMAs
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, MCO * MPIX, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, MCO + MPIX, 0);
        }
    };

    // ---------------------------------------------------------------- K loop
    issue();
    issue();
    wait_vm<NI>();                                // stage 0 has landed (stage 1 may be in flight)
    __syncthreads();
    int cr = 0, cc = 0;                           // vertical tap / chunk of the stage being consumed
    read_frags(0, 0, 0, 0);
    // one stage; P = register set that holds its first sub-step.  The fragments of sub-step u + 1 are read under the MFMAs of
    // sub-step u; before the LAST sub-step's MFMAs the wave waits for stage s + 1, passes the barrier (every wave holds the
    // last fragments of stage s: its slot is free), requests stage s + 2 into that slot and reads stage s + 1's first fragments.
    auto stage = [&](auto par_c, int s) {
        constexpr int P = decltype(par_c)::value;
        constexpr int LAST = NSUB == 3 ? P : (P ^ 1);
        const int slot = s & 1;
        read_frags(P ^ 1, 1, slot, cr);
        mfma_set(P);
        spread();
        if constexpr (NSUB == 3) {
            read_frags(P, 2, slot, cr);
            mfma_set(P ^ 1);
            spread();
        }
        if (s + 1 < nst) {
            wait_vm<0>();                         // stage s + 1 has landed
            __syncthreads();
            issue();                              // stage s + 2 -> the slot of stage s
            if (++cc == NC) { cc = 0; ++cr; }
            read_frags(LAST ^ 1, 0, slot ^ 1, cr);
        }
        mfma_set(LAST);
        spread();
    };
    if constexpr (NSUB == 3) {                    // the first sub-step's register set alternates from stage to stage
        for (int s = 0; s < nst; s += 2) {
            stage(std::integral_constant<int, 0>{}, s);
            if (s + 1 < nst) stage(std::integral_constant<int, 1>{}, s + 1);
        }
    } else {
        for (int s = 0; s < nst; ++s) stage(std::integral_constant<int, 0>{}, s);
    }
    __syncthreads();                               // every wave is done with the operand rings: As is reused below

    // ---------------------------------------------------------------- epilogue (the operand-ring kernel's)
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
    float* __restrict__ y1_p = a.y;
    const bool has_res = res_p != nullptr;
    const int act = a.act;
    const unsigned row_bytes = (unsigned)HoWo * 4u;
#pragma unroll
    for (int ni = 0; ni < MPIX; ++ni) {
        const int m = pix0 + wave_pix * WPIX + ni * 32 + l31;
        if (m >= a.M) continue;
        const unsigned n = (unsigned)(m / HoWo);
        const unsigned rem = (unsigned)m - n * (unsigned)HoWo;
#pragma unroll
        for (int mi = 0; mi < MCO; ++mi) {
            const int cl0 = wave_co * WCO + mi * 32 + 4 * khalf;          // tile-local channel of j = 0
            const unsigned off0 = ((n * (unsigned)a.Co + (unsigned)(co0 + cl0)) * (unsigned)HoWo + rem) * 4u;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float v[8], rv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) rv[e] = 0.f;
                if (has_res) {
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        rv[e] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(res_p) +
                                                                (off0 + (unsigned)((e & 3) + 8 * (2 * h + (e >> 2))) * row_bytes));
                }
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
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    *reinterpret_cast<float*>(reinterpret_cast<char*>(y1_p) +
                                              (off0 + (unsigned)((e & 3) + 8 * (2 * h + (e >> 2))) * row_bytes)) = v[e];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

static int env_int_v8(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}

// -1: follow DYNMM_IGEMM_V8 (default on); 0 / 1: forced by dynmm_debug_set_igemm_v8 (bit-identity tests flip it in-process)
static int g_v8_override = -1;

bool igemm_v8_eligible(const IgemmArgs& a, bool dgrad) {
    static const int on = env_int_v8("DYNMM_IGEMM_V8", 1);
    if (!(g_v8_override >= 0 ? g_v8_override : on)) return false;
    if (dgrad || a.mask) return false;                                       // forward only (input gradients: conv_wino / v5)
    if (!igemm_v5_eligible(a, false)) return false;                          // same family of convolutions, same switch ...
    if (a.KW == 1 && a.Ci % 32 != 0) return false;                           // ... vertical / 1x1: two 16-channel chunks per stage
    if (a.KH * (a.KW == 3 ? a.Ci / 16 : a.Ci / 32) < 2) return false;
    return true;
}

bool launch_igemm_v8(IgemmArgs& a, bool dgrad, hipStream_t st) {
    if (!igemm_v8_eligible(a, dgrad)) return false;
    a.CiR = a.Ci;
    a.M = a.N * a.Ho * a.Wo;
    a.K = a.KH * a.KW * a.Ci;
    a.CoP = a.Co;
    a.subpix = 0;
    const int tco = (a.Co % 128 == 0) ? 128 : 64;
    a.n_co_tiles = a.Co / tco;
    a.n_pix_tiles = ceil_div(a.M, 128);
    dim3 grid((unsigned)(a.n_co_tiles * a.n_pix_tiles));
    if (tco == 128) {
        if (a.KW == 3) hipLaunchKernelGGL((conv_igemm_v8_kernel<128, 2, 3>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv_igemm_v8_kernel<128, 2, 1>), grid, dim3(256), 0, st, a);
    } else {
        if (a.KW == 3) hipLaunchKernelGGL((conv_igemm_v8_kernel<64, 1, 3>), grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv_igemm_v8_kernel<64, 1, 1>), grid, dim3(256), 0, st, a);
    }
    return true;
}

}  // namespace dynmm

extern "C" int dynmm_debug_set_igemm_v8(int mode) {
    dynmm::g_v8_override = mode;
    return 0;
}
