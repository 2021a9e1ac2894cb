// Probe for the next round's operand pipeline: does hipcc for gfx950 accept the direct global->LDS load builtin, and
// what does a wave's dword / dwordx4 variant deliver?  (build: hipcc --offload-arch=gfx950 -O3 probe.hip -o probe)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int BYTES>
__global__ void __launch_bounds__(256) copy_via_lds(const float* __restrict__ src, float* __restrict__ dst, int n_per_wg) {
    extern __shared__ float tile[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int PER = BYTES / 4;                       // floats per lane and instruction
    const float* g = src + (size_t)blockIdx.x * n_per_wg;
    // each wave instruction moves 64 * BYTES contiguous bytes of global memory into 64 * BYTES contiguous bytes of LDS
    for (int i = wave * 64 * PER; i < n_per_wg; i += 4 * 64 * PER) {
        if constexpr (BYTES == 4)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + i + lane * PER),
                                             (__attribute__((address_space(3))) void*)(tile + i), 4, 0, 0);
        else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + i + lane * PER),
                                             (__attribute__((address_space(3))) void*)(tile + i), 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0);                       // vmcnt(0): the LDS writes of this wave have landed
    __syncthreads();
    for (int i = threadIdx.x; i < n_per_wg; i += 256) dst[(size_t)blockIdx.x * n_per_wg + i] = tile[i] * 2.f;
}

int main() {
    const int wgs = 2048, per = 8192;                    // 32 KB of LDS per workgroup
    const size_t n = (size_t)wgs * per;
    std::vector<float> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = (float)(i % 1000) * 0.5f;
    float *a, *b;
    (void)hipMalloc(&a, n * 4); (void)hipMalloc(&b, n * 4);
    (void)hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int variant = 0; variant < 2; ++variant) {
        (void)hipMemset(b, 0, n * 4);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            if (variant == 0) hipLaunchKernelGGL(copy_via_lds<4>, dim3(wgs), dim3(256), per * 4, 0, a, b, per);
            else hipLaunchKernelGGL(copy_via_lds<16>, dim3(wgs), dim3(256), per * 4, 0, a, b, per);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        }
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<float> out(n);
        (void)hipMemcpy(out.data(), b, n * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < n; ++i) bad += out[i] != h[i] * 2.f;
        printf("global_load_lds b%d: %s (%zu mismatches), %.1f us, %.2f TB/s read+write\n", variant ? 128 : 32,
               bad ? "WRONG" : "ok", bad, ms * 1e3, 2.0 * n * 4 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
