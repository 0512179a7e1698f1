// Microbenchmark of the conv_mfma stage structure on gfx950: 32 MFMAs (4x2 tiles x 4 k-substeps) per stage,
// optional A reads from LDS (4 ds_read_b128) and B loads from L2 (2 global_load_dwordx4) with ping-pong registers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int LDSR, int GLB, int ORDER, int IMM = 0>
__global__ void __launch_bounds__(256, 2) k(float* out, const float4* __restrict__ w, int stages, int ldsz) {
    extern __shared__ float4 sm[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < ldsz; i += 256) sm[i] = make_float4(i, 1.f, 2.f, 3.f);
    __syncthreads();
    f32x16 acc[4][2];
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    int aidx[4];
    for (int a = 0; a < 4; ++a) aidx[a] = (lane * 5 + a * 37) % (ldsz - 4096);
    float4 av[2][4], bv[2][2];
    for (int a = 0; a < 4; ++a) av[0][a] = av[1][a] = sm[aidx[a]];
    for (int b = 0; b < 2; ++b) bv[0][b] = bv[1][b] = w[lane + 64 * b];
    const float4* wp = w + lane;
    auto stage = [&](int s, auto CUR) {
        constexpr int cur = decltype(CUR)::value, nxt = cur ^ 1;
        // IMM: what a fully unrolled tap loop with compile-time geometry would give — offsets are instruction immediates
        int off = IMM ? ((cur * 29 + 7) & 1023) : (s * 29) & 4095;
        if (!IMM) asm volatile("" : "+s"(off));
        if (LDSR) {
#pragma unroll
            for (int a = 0; a < 4; ++a) av[nxt][a] = sm[aidx[a] + off];
        }
        if (GLB) {
#pragma unroll
            for (int b = 0; b < 2; ++b) bv[nxt][b] = IMM ? wp[(cur * 2 + b) * 64] : wp[(size_t)((s * 2 + b) & 1023) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        auto mf = [&](int q, int a, int b) {
            const float x = q == 0 ? av[cur][a].x : q == 1 ? av[cur][a].y : q == 2 ? av[cur][a].z : av[cur][a].w;
            const float y = q == 0 ? bv[cur][b].x : q == 1 ? bv[cur][b].y : q == 2 ? bv[cur][b].z : bv[cur][b].w;
            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a][b], 0, 0, 0);
        };
        if (ORDER == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) mf(q, a, b);
        } else if (ORDER == 1) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) mf(q, a, b);
                    __builtin_amdgcn_sched_barrier(0);
                }
        } else {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int b = 0; b < 2; ++b) mf(q, a, b);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int s = 0; s < stages; s += 2) {
        stage(s, std::integral_constant<int, 0>{});
        stage(s + 1, std::integral_constant<int, 1>{});
    }
    float sum = 0;
    for (int a = 0; a < 4; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) sum += acc[a][b][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}
template <typename K> double run(K kern, int wgs, int stages, float* d, float4* w, int ldsz) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), ldsz * 16, 0, d, w, stages, ldsz);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(256), ldsz * 16, 0, d, w, stages, ldsz);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return (double)wgs * 4 * stages * 32 * 4096.0 / (ms * 1e-3) / 1e12;
}
int main() {
    float* d; hipMalloc(&d, 256 * 4096 * 4);
    float4* w; hipMalloc(&w, 1024 * 64 * 16 * 2); hipMemset(w, 0, 1024 * 64 * 16 * 2);
    const int ldsz = 58 * 1024 / 16;   // 58 KB like the real kernel: two workgroups per CU
    const int st = 20000;
    for (int wgs : {512, 2048}) {
        printf("wgs=%d  q-major: mfma only %.1f  +lds+glb %.1f | tile-major(4 dependent): %.1f  %.1f | tile-major pairs (distance 2): %.1f  %.1f TF\n", wgs,
               run(k<0, 0, 0>, wgs, st, d, w, ldsz), run(k<1, 1, 0>, wgs, st, d, w, ldsz), run(k<0, 0, 1>, wgs, st, d, w, ldsz),
               run(k<1, 1, 1>, wgs, st, d, w, ldsz), run(k<0, 0, 2>, wgs, st, d, w, ldsz), run(k<1, 1, 2>, wgs, st, d, w, ldsz));
    }
    printf("wgs=512 q-major, +lds+glb with immediate offsets (no address arithmetic): %.1f TF\n", run(k<1, 1, 0, 1>, 512, st, d, w, ldsz));
    printf("wgs=512 q-major, +lds only: runtime offsets %.1f, immediates %.1f TF\n", run(k<1, 0, 0, 0>, 512, st, d, w, ldsz), run(k<1, 0, 0, 1>, 512, st, d, w, ldsz));
    return 0;
}
