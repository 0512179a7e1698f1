// How much VALU work fits in the shadow of v_mfma_f32_32x32x16_bf16?  Per iteration: 6 MFMAs (192 matrix-pipe cycles) on
// one accumulator and NV independent VALU instructions of the bf16x3 split (v_cvt_pk_bf16_f32, shifts / masks, v_pk_add_f32) on
// loop-carried registers, interleaved one MFMA : NV / 6 VALU by sched_group_barrier; 1 or 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/mfma_valu_coissue.hip -o /tmp/cv && /tmp/cv
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned pk(v2f x) { return __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2)); }
__device__ __forceinline__ v2f unpk(unsigned w) { return (v2f){__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xffff0000u)}; }
// NP = pairs split per iteration (9 VALU each); MF = MFMAs per iteration
template <int NP, int MF, int PIN> __global__ void __launch_bounds__(256) k(float* out, int iters, float a0) {
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    v2f x[NP > 0 ? NP : 1];
    for (int i = 0; i < NP; ++i) x[i] = (v2f){a0 + i + threadIdx.x, a0 * 3 + i};
    u32x4 a = {threadIdx.x, 2, 3, 4}, b = {5, 6, 7, threadIdx.x};
    unsigned sink = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < MF; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const unsigned h = pk(x[i]);
            v2f r1, r2;
            unsigned mm;
            if (PIN & 2) {            // packed subtraction spelled out
                const v2f a = unpk(h);
                asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r1) : "v"(x[i]), "v"(a));
                mm = pk(r1);
                const v2f b = unpk(mm);
                asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r2) : "v"(r1), "v"(b));
            } else if (PIN & 4) {     // x - bf16 half through v_dot2_f32_bf16: x + h.lo * (-1) + h.hi * 0
                asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r1.x) : "v"(h), "s"(0x0000bf80u), "v"(x[i].x));
                asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r1.y) : "v"(h), "s"(0xbf800000u), "v"(x[i].y));
                mm = pk(r1);
                asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r2.x) : "v"(mm), "s"(0x0000bf80u), "v"(r1.x));
                asm("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r2.y) : "v"(mm), "s"(0xbf800000u), "v"(r1.y));
            } else {
                r1 = x[i] - unpk(h);
                mm = pk(r1);
                r2 = r1 - unpk(mm);
            }
            sink ^= h ^ mm ^ pk(r2);
            x[i] = x[i] + (v2f){1.0f, 0.5f};
        }
        if (PIN & 1) {
#pragma unroll
            for (int m = 0; m < MF; ++m) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, (NP * 11 + (MF > 0 ? MF : 1) - 1) / (MF > 0 ? MF : 1), 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = __builtin_bit_cast(float, sink);
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename K> double run(K kern, int wgs, int iters, float* d) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, d, iters, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), 0, 0, d, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / iters;      // ns per iteration
}
int main() {
    float* d; hipMalloc(&d, 256 * 8192 * 4);
    const int it = 20000;
    for (int wpc = 1; wpc <= 2; ++wpc) {
        const int wgs = 256 * wpc;
        printf("waves/SIMD=%d, ns per iteration (6 MFMAs = 192 pipe cycles = 80 ns at 2.4 GHz per wave)\n", wpc);
        printf("  MFMA only            %.1f\n", run(k<0, 6, 0>, wgs, it, d));
        printf("  VALU only: 2 pairs %.1f  4 pairs %.1f  6 pairs %.1f\n", run(k<2, 0, 0>, wgs, it, d), run(k<4, 0, 0>, wgs, it, d), run(k<6, 0, 0>, wgs, it, d));
        printf("  both, compiler order: 2 pairs %.1f  4 pairs %.1f  6 pairs %.1f\n", run(k<2, 6, 0>, wgs, it, d), run(k<4, 6, 0>, wgs, it, d), run(k<6, 6, 0>, wgs, it, d));
        printf("  both, pinned 1 MFMA : n VALU: 2 pairs %.1f  4 pairs %.1f  6 pairs %.1f\n", run(k<2, 6, 1>, wgs, it, d), run(k<4, 6, 1>, wgs, it, d), run(k<6, 6, 1>, wgs, it, d));
        printf("  VALU only, asm v_pk_add_f32: 2 pairs %.1f  4 pairs %.1f  6 pairs %.1f | both: %.1f %.1f %.1f\n", run(k<2, 0, 2>, wgs, it, d), run(k<4, 0, 2>, wgs, it, d), run(k<6, 0, 2>, wgs, it, d),
               run(k<2, 6, 2>, wgs, it, d), run(k<4, 6, 2>, wgs, it, d), run(k<6, 6, 2>, wgs, it, d));
        printf("  VALU only, v_dot2_f32_bf16:  2 pairs %.1f  4 pairs %.1f  6 pairs %.1f | both: %.1f %.1f %.1f\n", run(k<2, 0, 4>, wgs, it, d), run(k<4, 0, 4>, wgs, it, d), run(k<6, 0, 4>, wgs, it, d),
               run(k<2, 6, 4>, wgs, it, d), run(k<4, 6, 4>, wgs, it, d), run(k<6, 6, 4>, wgs, it, d));
    }
    return 0;
}
