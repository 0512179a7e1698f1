// Microbenchmark: sustained issue rate of fp32 MFMA shapes on gfx950 (independent accumulators, no memory).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC> __global__ void __launch_bounds__(256) k16(float* out, int iters, float a, float b) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> __global__ void __launch_bounds__(256) k32(float* out, int iters, float a, float b) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename K> double run(K k, int wgs, int iters, double flops_per_mfma, int nacc, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, d, iters, 1.0f, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, 0, d, iters, 1.0f, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double total = (double)wgs * 4 * iters * 4 * nacc * flops_per_mfma;
    return total / (ms * 1e-3) / 1e12;
}
int main() {
    float* d; hipMalloc(&d, 256 * 8192 * 4);
    const int it = 20000;
    for (int wpc = 1; wpc <= 4; wpc *= 2) {   // workgroups (of 4 waves) per CU -> waves per SIMD
        int wgs = 256 * wpc;
        printf("waves/SIMD=%d  16x16x4: nacc4 %.1f TF  nacc8 %.1f TF | 32x32x2: nacc2 %.1f TF nacc4 %.1f TF\n", wpc,
               run(k16<4>, wgs, it, 2048, 4, d), run(k16<8>, wgs, it, 2048, 8, d), run(k32<2>, wgs, it, 4096, 2, d), run(k32<4>, wgs, it, 4096, 4, d));
    }
    return 0;
}
