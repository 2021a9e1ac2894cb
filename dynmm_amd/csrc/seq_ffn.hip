// The feed-forward block of nn.TransformerEncoderLayer on the [B, D, T] layout of the ModalityDynMM experts
// (ModalityDynMM/affect/affect_dyn.py:107-175 builds them from nn.TransformerEncoderLayer(d_model <= 120, dim_feedforward 2048)):
//     out = W2 . dropout(relu(W1 . x + b1)) + b2
// as ONE launch forward and ONE launch for the two data gradients of the backward.  88 % of a layer's arithmetic is here, and as
// two separate 1x1 convolutions + dropout + activation-backward passes it ran at 25-38 TF/s with the hidden activation
// ([B, 2048, T]: 52 MB at batch 128) crossing HBM six times per layer and step.
//
// Decomposition: a workgroup owns 128 tokens (a wave: 32 = the column block of v_mfma_f32_32x32x2_f32) and F / nsplit hidden
// units, walked in blocks of 32.  Per block and wave
//   forward   H  [32 f x 32 tok]  = W1blk[32 x D] . X[D x 32]            (X lives in registers for the whole kernel)
//             Hd = dropout(relu(H + b1))  -> stored once (the backward needs it), and fed STRAIGHT from the accumulator
//             registers into the second product: a lane of the 32x32 accumulator holds rows {8i + 4h + j} of its token
//             column, which is exactly a B operand of the next MFMA if the A operand (W2) is read in that k order
//             out[D x 32 tok] += W2blk[D x 32 f] . Hd
//   backward  dHd = W2blk^T . dOut (dOut in registers), dH = dHd * (Hd > 0) / (1 - p) -> stored (weight gradients read it),
//             dX[D x 32 tok] += W1blk^T . dH     (same chaining through the accumulator registers)
// The two weight blocks of a step ([32 x D] rows of W1: contiguous; [D x 32] columns of W2) are staged global -> registers ->
// LDS one block ahead (double buffer, one barrier per block) in their NATURAL layout: no packed copies of weights that change
// every step.  The F split leaves `nsplit` partial sums of the output; the consumer (LayerNorm forward / the gradient sum of
// the residual branch) adds them in a fixed order — no atomics, results are bit-reproducible.
// The weight gradients (contractions over the tokens) stay on the grouped weight-gradient kernels with X / Hd / dH / dOut as
// 1x1-convolution operands.
#include "common.h"

namespace dynmm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct FfnDrop {
    const unsigned char* mask;
    const unsigned long long* step;
    unsigned long long seed, offset;
    float p;
};

struct FfnArgs {
    const float* x;        // forward: layer input [B, D, T]; backward: gradient of the block's output [B, D, T]
    const float* w1;       // [F, D]
    const float* b1;       // [F] (forward)
    const float* w2;       // [D, F]
    const float* hid_in;   // backward: Hd [B, F, T]
    float* hid_out;        // forward: Hd; backward: dH
    float* parts;          // [nsplit][B, D, T]
    int B, D, T, F, nsplit, nfb, ntok;
    float scale;           // backward: 1 / (1 - p)
    FfnDrop drop;
};

__device__ __forceinline__ void ffn_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                           uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

constexpr int kFfnLd2 = 36;                      // W2 block row: 32 hidden units + 4 (rows 144 B apart: conflict-free b128)

// DT: compile-time bound on D (D <= DT, DT % 4 == 0); the k order of the D contraction is k(s, h) = 4 (s/2) + 2 h + s%2 for
// MFMA step s and lane half h, so a lane reads its A values of two consecutive steps as one ds_read_b64.
// VEC: D % 4 == 0 (rows of W1 are 16-byte aligned).  Nothing in the kernel is predicated: tokens past the end are clamped to the
// last one (their lanes recompute and re-store its values), operands outside [0, D) are loaded from a clamped address and
// replaced by zero — guarded loads compile to one branch and one wait per load.
template <int DT, bool BWD, bool VEC>
__global__ void __launch_bounds__(256, 2) ffn_kernel(const FfnArgs a) {
    constexpr int NDB = (DT + 31) / 32;          // 32-row blocks of the D-sized output
    constexpr int KS = DT / 2;                   // MFMA steps of the contraction over D
    constexpr int LD1 = DT + 4;                  // W1 block row (16-byte aligned rows)
    constexpr int T1 = 32 * LD1, T2 = NDB * 32 * kFfnLd2;
    constexpr int NV1 = 32 * (DT / 4), NV2 = DT * 8, NV = NV1 + NV2;      // float4s of one block's two tiles
    constexpr int NST = (NV + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [2][T1 + T2] | b1 slice of this workgroup [nfb * 32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, h = lane >> 5;
    const int split = blockIdx.x % a.nsplit, tile = blockIdx.x / a.nsplit;
    const int D = a.D, T = a.T, F = a.F;
    const int tok = min(tile * 128 + wave * 32 + l32, a.ntok - 1);
    const int b = tok / T, t = tok - b * T;
    const size_t xbase = (size_t)b * D * T + t, hbase = (size_t)b * F * T + t;
    const int fb0 = split * a.nfb;               // first 32-block of hidden units of this workgroup

    // ---- the token operand: X (forward) / dOut (backward), resident in registers ----
    float xr[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        const int k = 4 * (s >> 1) + 2 * h + (s & 1);
        const float v = a.x[xbase + (size_t)min(k, D - 1) * T];
        xr[s] = k < D ? v : 0.f;
    }

    // zero both buffers once: columns / rows beyond D are read as operands when DT > D
    for (int i = tid; i < 2 * (T1 + T2); i += 256) lds[i] = 0.f;
    float* b1s = lds + 2 * (T1 + T2);
    if constexpr (!BWD)
        for (int i = tid; i < a.nfb * 32; i += 256) b1s[i] = a.b1[fb0 * 32 + i];
    __syncthreads();

    // ---- staging of one block's tiles: float4 v of this thread <- global, later -> LDS ----
    float4 stg[NST];
    auto stage_load = [&](int fb) {
        const int f0 = fb * 32;
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int v = min(tid + i * 256, NV - 1);
            const bool first = v < NV1;                       // W1 rows f0 .. f0+31, DT/4 quads each | W2 rows d, 8 quads each
            const int w = first ? v : v - NV1;
            const int row = first ? w / (DT / 4) : (w >> 3), c4 = first ? w - row * (DT / 4) : (w & 7);
            const bool live = first ? (c4 * 4 < D) : (row < D);
            const float* p1 = a.w1 + (size_t)(f0 + row) * D;
            const float* p2 = a.w2 + (size_t)(live ? row : 0) * F + f0 + c4 * 4;
            float4 val;
            if constexpr (VEC) {
                val = *reinterpret_cast<const float4*>(first ? p1 + (live ? c4 * 4 : 0) : p2);
            } else {                                          // W1 rows are not 16-byte aligned (the gate's d_model = 10)
                const float4 v2 = *reinterpret_cast<const float4*>(first ? a.w2 : p2);
                const float* q = first ? p1 : a.w1;
                const int c = first ? c4 * 4 : 0;
                float4 v1;
                v1.x = q[min(c + 0, D - 1)];
                v1.y = q[min(c + 1, D - 1)];
                v1.z = q[min(c + 2, D - 1)];
                v1.w = q[min(c + 3, D - 1)];
                if (c + 1 >= D) v1.y = 0.f;
                if (c + 2 >= D) v1.z = 0.f;
                if (c + 3 >= D) v1.w = 0.f;
                val = first ? v1 : v2;
            }
            stg[i] = live ? val : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stage_store = [&](int buf) {
        float* t1 = lds + buf * (T1 + T2);
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int v = tid + i * 256;
            const bool first = v < NV1;
            const int w = first ? v : v - NV1;
            const int row = first ? w / (DT / 4) : (w >> 3), c4 = first ? w - row * (DT / 4) : (w & 7);
            float* dst = first ? t1 + row * LD1 + c4 * 4 : t1 + T1 + row * kFfnLd2 + c4 * 4;
            if (i < NST - 1 || NV % 256 == 0 || v < NV) *reinterpret_cast<float4*>(dst) = stg[i];
        }
    };

    stage_load(fb0);
    stage_store(0);
    __syncthreads();

    // dropout state (forward)
    const unsigned long long doff = a.drop.offset + (a.drop.step ? *a.drop.step : 0ull);
    const uint32_t dk0 = (uint32_t)a.drop.seed, dk1 = (uint32_t)(a.drop.seed >> 32);
    const float dp = a.drop.p, dinv = dp > 0.f ? 1.f / (1.f - dp) : 1.f;
    const uint32_t thr16 = (uint32_t)(dp * 65536.f + 0.5f);      // keep iff a 16-bit uniform integer >= round(65536 p)

    f32x16 oacc[NDB];
#pragma unroll
    for (int j = 0; j < NDB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[j][r] = 0.f;

    for (int i = 0; i < a.nfb; ++i) {
        const int fb = fb0 + i, f0 = fb * 32;
        const float* t1 = lds + (i & 1) * (T1 + T2);
        const float* t2 = t1 + T1;
        if (i + 1 < a.nfb) stage_load(fb + 1);

        float hv[16];
        if constexpr (BWD) {
            // Hd of this block, requested before the product that it masks
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = f0 + 8 * (r >> 2) + 4 * h + (r & 3);
                hv[r] = a.hid_in[hbase + (size_t)f * T];
            }
        }

        // ---- first product: [32 f x 32 tok] over D ----
        f32x16 hacc;
#pragma unroll
        for (int r = 0; r < 16; ++r) hacc[r] = 0.f;
        // (A fragments are read one group of MFMAs ahead: left to itself the compiler waits for each read right before its use)
        if constexpr (!BWD) {
            const float* ap = t1 + l32 * LD1 + 2 * h;
            float2 nx[2] = {*reinterpret_cast<const float2*>(ap), *reinterpret_cast<const float2*>(ap + 4)};
#pragma unroll
            for (int u = 0; u < KS / 2; u += 2) {
                const float2 a0 = nx[0], a1 = nx[1];
                if (u + 2 < KS / 2) nx[0] = *reinterpret_cast<const float2*>(ap + 4 * (u + 2));
                if (u + 3 < KS / 2) nx[1] = *reinterpret_cast<const float2*>(ap + 4 * (u + 3));
                hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, xr[2 * u], hacc, 0, 0, 0);
                hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, xr[2 * u + 1], hacc, 0, 0, 0);
                if (u + 1 < KS / 2) {
                    hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, xr[2 * u + 2], hacc, 0, 0, 0);
                    hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, xr[2 * u + 3], hacc, 0, 0, 0);
                }
            }
        } else {
            const float* ap = t2 + 2 * h * kFfnLd2 + l32;                     // W2[d' = k(s, h)][f0 + l32]
            float nx[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) nx[e] = ap[(4 * (e >> 1) + (e & 1)) * kFfnLd2];
#pragma unroll
            for (int s = 0; s < KS; s += 4) {
                float av[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) av[e] = nx[e];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (s + 4 + e < KS) nx[e] = ap[(4 * ((s + 4 + e) >> 1) + (e & 1)) * kFfnLd2];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (s + e < KS) hacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], xr[s + e], hacc, 0, 0, 0);
            }
        }

        // the next block's tiles have had a whole product to arrive; writing them here (their buffer was last read before the
        // previous barrier) keeps the wait for them clear of the hidden-activation stores below
        if (i + 1 < a.nfb) stage_store((i + 1) & 1);

        // ---- the element-wise middle, on the accumulator registers ----
        if constexpr (!BWD) {
            // keep flags of this lane's 16 hidden units: one Philox call per EIGHT units (16 random bits each — a call is ~100
            // VALU instructions, 40 of them quarter-rate multiplies, and four calls per block cost 0.44 of the block's MFMA time)
            uint32_t rnd[2][4];
            if (dp > 0.f && !a.drop.mask) {
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const int fq = f0 + 16 * c + 4 * h;       // first of the two quads (q = 2c, 2c + 1) this call serves
                    const unsigned long long qi = ((unsigned long long)b * (F >> 2) + (fq >> 2)) * T + t;
                    ffn_philox((uint32_t)qi, (uint32_t)(qi >> 32), (uint32_t)doff, (uint32_t)(doff >> 32), dk0, dk1, rnd[c]);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int fq = f0 + 8 * q + 4 * h;                            // this lane's four consecutive hidden units
                float keep[4] = {dinv, dinv, dinv, dinv};
                if (dp > 0.f) {
                    if (a.drop.mask) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) keep[e] = a.drop.mask[hbase + (size_t)(fq + e) * T] ? dinv : 0.f;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const uint32_t word = rnd[q >> 1][2 * (q & 1) + (e >> 1)];
                            keep[e] = ((e & 1) ? (word >> 16) : (word & 0xffffu)) >= thr16 ? dinv : 0.f;
                        }
                    }
                }
                const float4 bq = *reinterpret_cast<const float4*>(b1s + (fq - fb0 * 32));
                const float bb[4] = {bq.x, bq.y, bq.z, bq.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = fmaxf(hacc[4 * q + e] + bb[e], 0.f) * keep[e];
                    hv[4 * q + e] = v;
                    a.hid_out[hbase + (size_t)(fq + e) * T] = v;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = f0 + 8 * (r >> 2) + 4 * h + (r & 3);
                const float v = hv[r] > 0.f ? hacc[r] * a.scale : 0.f;
                hv[r] = v;
                a.hid_out[hbase + (size_t)f * T] = v;
            }
        }

        // ---- second product: [D x 32 tok] += over the 32 hidden units, B operand = the registers just made ----
        if constexpr (!BWD) {
            const float* ap = t2 + l32 * kFfnLd2 + 4 * h;
            float4 nx = *reinterpret_cast<const float4*>(ap);
#pragma unroll
            for (int jq = 0; jq < NDB * 4; ++jq) {
                    const int j = jq >> 2, q = jq & 3;
                    const float4 av = nx;
                    if (jq + 1 < NDB * 4)
                        nx = *reinterpret_cast<const float4*>(ap + 32 * ((jq + 1) >> 2) * kFfnLd2 + 8 * ((jq + 1) & 3));
                    oacc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, hv[4 * q + 0], oacc[j], 0, 0, 0);
                    oacc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, hv[4 * q + 1], oacc[j], 0, 0, 0);
                    oacc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, hv[4 * q + 2], oacc[j], 0, 0, 0);
                    oacc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, hv[4 * q + 3], oacc[j], 0, 0, 0);
                }
        } else {
            const float* ap = t1 + 4 * h * LD1 + l32;                         // W1[f0 + fl(r, h)][d = 32 j + l32]
            float nx[NDB];
#pragma unroll
            for (int j = 0; j < NDB; ++j) nx[j] = ap[32 * j];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float av[NDB];
#pragma unroll
                for (int j = 0; j < NDB; ++j) av[j] = nx[j];
                if (r + 1 < 16) {
                    const int fl = 8 * ((r + 1) >> 2) + ((r + 1) & 3);
#pragma unroll
                    for (int j = 0; j < NDB; ++j) nx[j] = ap[fl * LD1 + 32 * j];
                }
#pragma unroll
                for (int j = 0; j < NDB; ++j) oacc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], hv[r], oacc[j], 0, 0, 0);
            }
        }

        __syncthreads();
    }

    // ---- this workgroup's partial sum over its hidden units ----
    {
        float* dst = a.parts + (size_t)split * a.B * D * T + xbase;
#pragma unroll
        for (int j = 0; j < NDB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int d = 32 * j + 8 * (r >> 2) + 4 * h + (r & 3);
                if (d < D) dst[(size_t)d * T] = oacc[j][r];
            }
    }
}

template <int DT>
static constexpr size_t ffn_lds_bytes() {
    return 2 * (size_t)(32 * (DT + 4) + ((DT + 31) / 32) * 32 * kFfnLd2) * sizeof(float);
}

template <int DT, bool BWD, bool VEC>
static int launch_ffn_t(const FfnArgs& a, hipStream_t st) {
    const size_t lds = ffn_lds_bytes<DT>() + (size_t)a.nfb * 32 * sizeof(float);
    static size_t attr_done = 0;
    if (lds > attr_done) {
        DYNMM_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&ffn_kernel<DT, BWD, VEC>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = lds;
    }
    const int tiles = ceil_div(a.ntok, 128);
    hipLaunchKernelGGL((ffn_kernel<DT, BWD, VEC>), dim3(tiles * a.nsplit), dim3(256), lds, st, a);
    DYNMM_LAUNCH_CHECK();
    return DYNMM_OK;
}

template <bool BWD>
static int launch_ffn(const FfnArgs& a, hipStream_t st) {
    if (a.D % 4 != 0) {                          // generic: scalar staging of W1
        if (a.D <= 12) return launch_ffn_t<12, BWD, false>(a, st);
        if (a.D <= 64) return launch_ffn_t<64, BWD, false>(a, st);
        return launch_ffn_t<128, BWD, false>(a, st);
    }
    if (a.D <= 32) return launch_ffn_t<32, BWD, true>(a, st);
    if (a.D <= 60) return launch_ffn_t<60, BWD, true>(a, st);
    if (a.D <= 64) return launch_ffn_t<64, BWD, true>(a, st);
    if (a.D <= 120) return launch_ffn_t<120, BWD, true>(a, st);
    return launch_ffn_t<128, BWD, true>(a, st);
}

static bool ffn_geom_ok(int B, int D, int T, int F) {
    return B > 0 && T > 0 && D >= 1 && D <= 128 && F >= 32 && F % 32 == 0 && (long long)B * T < (1ll << 30);
}

// how many ways the hidden units are split: enough workgroups for 2 x 256 slots, a divisor of F / 32, at most 16
static int ffn_pick_split(int B, int T, int F) {
    const int tiles = ceil_div(B * T, 128), nb = F / 32;
    int best = 1;
    for (int s = 1; s <= 16 && s <= nb; ++s)
        if (nb % s == 0) {
            best = s;
            if (tiles * s >= 384) break;
        }
    return best;
}

}  // namespace dynmm

using namespace dynmm;

extern "C" int dynmm_ffn_supported(int B, int D, int T, int F) { return ffn_geom_ok(B, D, T, F) ? 1 : 0; }

extern "C" int dynmm_ffn_nsplit(int B, int D, int T, int F) { return ffn_geom_ok(B, D, T, F) ? ffn_pick_split(B, T, F) : 0; }

extern "C" int dynmm_ffn_fwd(const float* x, const float* w1, const float* b1, const float* w2, float* hidden,
                             float* out_parts, int B, int D, int T, int F, int nsplit, const dynmm_dropout* drop,
                             void* stream) {
    (void)hipGetLastError();
    if (!x || !w1 || !b1 || !w2 || !hidden || !out_parts) return DYNMM_EINVAL;
    if (drop && !(drop->p >= 0.f && drop->p < 1.f)) return DYNMM_EINVAL;
    if (!ffn_geom_ok(B, D, T, F)) return DYNMM_EUNSUPPORTED;
    if (nsplit <= 0 || (F / 32) % nsplit != 0) return DYNMM_EINVAL;
    if ((((uintptr_t)w1 | (uintptr_t)w2 | (uintptr_t)b1) & 15) != 0) return DYNMM_EUNSUPPORTED;
    FfnArgs a{};
    a.x = x; a.w1 = w1; a.b1 = b1; a.w2 = w2; a.hid_out = hidden; a.parts = out_parts;
    a.B = B; a.D = D; a.T = T; a.F = F; a.nsplit = nsplit; a.nfb = F / 32 / nsplit; a.ntok = B * T; a.scale = 1.f;
    if (drop && drop->p > 0.f) {
        a.drop.mask = drop->mask; a.drop.step = drop->step; a.drop.seed = drop->seed; a.drop.offset = drop->offset;
        a.drop.p = drop->p;
    }
    return launch_ffn<false>(a, (hipStream_t)stream);
}

extern "C" int dynmm_ffn_bwd_data(const float* dout, const float* hidden, const float* w1, const float* w2, float* dhidden,
                                  float* dx_parts, int B, int D, int T, int F, int nsplit, float p, void* stream) {
    (void)hipGetLastError();
    if (!dout || !hidden || !w1 || !w2 || !dhidden || !dx_parts || !(p >= 0.f && p < 1.f)) return DYNMM_EINVAL;
    if (!ffn_geom_ok(B, D, T, F)) return DYNMM_EUNSUPPORTED;
    if (nsplit <= 0 || (F / 32) % nsplit != 0) return DYNMM_EINVAL;
    if ((((uintptr_t)w1 | (uintptr_t)w2) & 15) != 0) return DYNMM_EUNSUPPORTED;
    FfnArgs a{};
    a.x = dout; a.w1 = w1; a.w2 = w2; a.hid_in = hidden; a.hid_out = dhidden; a.parts = dx_parts;
    a.B = B; a.D = D; a.T = T; a.F = F; a.nsplit = nsplit; a.nfb = F / 32 / nsplit; a.ntok = B * T;
    a.scale = p > 0.f ? 1.f / (1.f - p) : 1.f;
    return launch_ffn<true>(a, (hipStream_t)stream);
}
