// Does a back-to-back DEPENDENT chain of v_mfma_f32_16x16x4_f32 (same accumulator 4 times in a row, the order
// k_conv_n16's plain path issues them in) run as fast as the same MFMAs interleaved over 4 / 8 accumulators?
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_chain.hip -o /tmp/c && /tmp/c
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int CHAIN> __global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if (CHAIN) {
#pragma unroll
            for (int i = 0; i < NACC; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename K> double run(K kern, int wgs, int iters, int nacc, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, d, iters, 1.0f, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, d, iters, 1.0f, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return (double)wgs * 4 * iters * 4 * nacc * 2048.0 / (ms * 1e-3) / 1e12;
}
int main() {
    float* d; hipMalloc(&d, 256 * 8192 * 4);
    const int it = 20000;
    for (int wpc = 1; wpc <= 2; ++wpc) {
        const int wgs = 256 * wpc;
        printf("waves/SIMD=%d  16x16x4 interleaved: nacc4 %.1f nacc8 %.1f | 4-deep dependent chains: nacc4 %.1f nacc8 %.1f TFLOP/s\n", wpc,
               run(k<4, 0>, wgs, it, 4, d), run(k<8, 0>, wgs, it, 8, d), run(k<4, 1>, wgs, it, 4, d), run(k<8, 1>, wgs, it, 8, d));
    }
    return 0;
}
