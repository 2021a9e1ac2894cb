// fp32 MFMA ceiling on RANDOM, CHANGING operands (scratch/mfma/peak.hip feeds the same two values to every MFMA, which
// toggles almost nothing and runs at 2.38 GHz; a convolution feeds fresh data to every instruction).
//   mode 0: constant operands (peak.hip's figure, for the same box)
//   mode 1: 16 random A + 16 random B registers per lane, a different pair per MFMA
//   mode 2: mode 1 + the LDS traffic of the implicit GEMM (24 ds_read_b32 per 16 MFMAs, operands taken from LDS)
//   mode 3: mode 2 + one 16-byte global load per lane per 16 MFMAs (L2-resident buffer), written back to LDS
//   mode 4: mode 3 with THREE 16-byte loads per lane and step (the 12 KB per workgroup and K-step of the convolution) from an
//           L2-resident buffer; mode 5: the same from a 1 GiB buffer walked once (every load an HBM miss)
// Reports TFLOP/s (HIP events) and the shader clock (s_memtime ticks per s_memrealtime tick x 100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void __launch_bounds__(256) k(float* out, const float* rnd, int iters, const float4* big, unsigned bigmask) {
    __shared__ __attribute__((aligned(16))) float lds[2][16][192];
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    float a[16], b[16];
    for (int i = 0; i < 16; ++i) {
        a[i] = MODE == 0 ? 1.f + t : rnd[(i * 256 + t) & 65535];
        b[i] = MODE == 0 ? 2.f : rnd[(4096 + i * 256 + t) & 65535];
    }
    for (int i = t; i < 2 * 16 * 192; i += 256) (&lds[0][0][0])[i] = rnd[i & 65535];
    __syncthreads();
    const float4* g4 = reinterpret_cast<const float4*>(rnd);
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 2) {
            const int buf = it & 1;
            float4 gl;
            if (MODE == 3) gl = g4[((it * 256 + t) * 7) & 16383];
            if (MODE >= 4) {
                const unsigned base = ((blockIdx.x * (unsigned)iters + it) * 768u + t) & bigmask;
                const float4 g0 = big[base], g1 = big[(base + 256) & bigmask], g2 = big[(base + 512) & bigmask];
                gl = make_float4(g0.x + g1.y, g0.y + g2.z, g1.z + g2.w, g0.w + g1.x);
            }
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                a[2 * kk] = lds[buf][2 * kk + (lane >> 5)][wave * 32 + (lane & 31)];
                a[2 * kk + 1] = lds[buf][2 * kk + (lane >> 5)][64 + (lane & 31)];
                b[kk] = lds[buf][2 * kk + (lane >> 5)][128 + (wave & 1) * 32 + (lane & 31)];
            }
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * kk], b[kk], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * kk + 1], b[kk], acc[1], 0, 0, 0);
            }
            if (MODE >= 3) {
                *reinterpret_cast<float4*>(&lds[buf ^ 1][t >> 4][(t & 15) * 4]) = gl;
                __syncthreads();
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * kk], b[(kk * 5) & 15], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2 * kk + 1], b[(kk * 5 + 3) & 15], acc[1], 0, 0, 0);
            }
        }
    }
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[2 + blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (float)(c1 - c0); out[1] = (float)(w1 - w0); }
}

template <int MODE>
void run(int blocks, int iters, const float* rnd, const float4* big = nullptr, unsigned bigmask = 0) {
    float* d; hipMalloc(&d, (2 + blocks * 256) * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, rnd, iters, big, bigmask);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, rnd, iters, big, bigmask);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    double flops = (double)blocks * 4 * iters * 16 * 4096.0;
    printf("mode %d blocks=%d (%.0f/CU): %.3f ms  %.1f TFLOP/s  clock %.0f MHz\n", MODE, blocks, blocks / 256.0, ms,
           flops / ms / 1e9, h[0] / (h[1] / 100.0));
    hipFree(d);
}

int main() {
    std::vector<float> h(65536);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float* rnd; hipMalloc(&rnd, h.size() * 4);
    hipMemcpy(rnd, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(1536, 3000, rnd);
        run<1>(1536, 3000, rnd);
        run<2>(1536, 3000, rnd);
        run<3>(1536, 3000, rnd);
    }
    run<1>(512, 6000, rnd);
    run<2>(512, 6000, rnd);
    // how far does ONE workgroup per CU (one wave per SIMD) get, with and without the barrier / load of the real loop?
    run<1>(256, 8000, rnd);
    run<2>(256, 8000, rnd);
    run<3>(256, 8000, rnd);
    run<3>(512, 6000, rnd);
    run<3>(768, 4000, rnd);
    float4* big; hipMalloc(&big, (size_t)1 << 30); hipMemset(big, 0x3c, (size_t)1 << 30);
    run<4>(768, 4000, rnd, big, (1u << 14) - 1);          // 256 KB window: L2 hits
    run<5>(768, 4000, rnd, big, (1u << 26) - 1);          // 1 GiB: HBM
    run<4>(1536, 2000, rnd, big, (1u << 14) - 1);
    run<5>(1536, 2000, rnd, big, (1u << 26) - 1);
    return 0;
}
