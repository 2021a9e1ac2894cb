// fp32 MFMA ceiling on this part: back-to-back v_mfma_f32_32x32x2_f32, NACC independent accumulators
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (float)(c1 - c0); out[1] = (float)(w1 - w0); }
}
template <int NACC>
void run(int blocks, int iters) {
    float* d; hipMalloc(&d, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    double flops = (double)blocks * 4 * iters * 4 * NACC * 4096.0;
    printf("NACC=%d blocks=%d (%.1f/CU): %.3f ms  %.1f TFLOP/s  clock %.0f MHz\n", NACC, blocks, blocks / 256.0, ms, flops / ms / 1e9, h[0] / (h[1] / 100.0));
    hipFree(d);
}
int main() {
    run<4>(256, 4000); run<4>(512, 4000); run<4>(1024, 2000); run<2>(512, 4000); run<1>(512, 4000); run<4>(256 * 6, 1000);
    return 0;
}
