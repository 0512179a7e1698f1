// Microbenchmark of k_conv_n16's tap loop on gfx950 (no staging, no epilogue): per tap TM x 4 v_mfma_f32_16x16x4_f32,
// TM ds_read_b128 of next-tap A fragments from a CS = 20 image, one global_load_dwordx4 refilling a 9-deep weight
// ring; 27 taps per chunk, optional barrier per chunk.  Which ingredient costs how much of the matrix pipe?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o n16_tap_bench n16_tap_bench.hip && ./n16_tap_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDSR: 0 no A reads after tap 0, 1 real pattern (16 consecutive voxels x 80 B), 2 conflict-free dummy pattern
// GLB:  0 no ring refills, 1 refills
// BAR:  barriers per chunk (0, 1, 2)
template <int WAVES, int TM, int LDSR, int GLB, int BAR, int CS4 = 5, int WRAP = 10, int IMM = 0>
__global__ void __launch_bounds__(WAVES * 64, WAVES == 4 ? 2 : 1) k(float* out, const float4* __restrict__ w, int chunks, int nvox) {
    constexpr int NT = WAVES * 64, BR = 9, NTAPS = 27;
    extern __shared__ __attribute__((aligned(16))) float4 sm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i16 = lane & 15, q = lane >> 4;
    for (int i = tid; i < nvox * CS4; i += NT) sm[i] = make_float4(i & 7, 1.f, 2.f, 3.f);
    __syncthreads();
    const f32x4* A = reinterpret_cast<const f32x4*>(sm);
    const int Wp = 12, Hp = 12;
    int aidx[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int row = ((wave * TM + tm) * 16 + i16) % (5 * WRAP * WRAP);          // output voxel of a WRAP x WRAP x 5 slab
        const int z = row / (WRAP * WRAP), y = (row % (WRAP * WRAP)) / WRAP, x = row % WRAP;
        aidx[tm] = ((z * Hp + y) * Wp + x) * CS4 + q;
    }
    f32x4 acc[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) acc[tm] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float4* wp = w + lane;
    float4 breg[BR];
#pragma unroll
    for (int t = 0; t < BR; ++t) breg[t] = wp[t * 64];
    constexpr bool PING = TM <= 4;
    constexpr int HT = PING ? TM : TM / 2;
    f32x4 av[PING ? 2 : 1][TM];
    for (int ch = 0; ch < chunks; ++ch) {
        if (BAR >= 1) __syncthreads();
        if (BAR >= 2) __syncthreads();
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) av[0][tm] = A[aidx[tm]];
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
            const int nt = t + 1;
            int noff = (((nt / 9) * Hp + (nt / 3) % 3) * Wp + nt % 3) * CS4;
            if (!IMM) asm volatile("" : "+s"(noff));   // IMM: tap offsets are compile-time constants -> ds_read immediates
            auto rd = [&](int tm) { return LDSR == 2 ? A[tid + tm * NT + (nt & 1) * 64] : A[aidx[tm] + noff]; };
            if (PING && nt < NTAPS && LDSR) {
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) av[nt & 1][tm] = rd(tm);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tm = 0; tm < HT; ++tm) {
                const f32x4 aq = av[PING ? (t & 1) : 0][tm];
                acc[tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq.x, breg[t % BR].x, acc[tm], 0, 0, 0);
                acc[tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq.y, breg[t % BR].y, acc[tm], 0, 0, 0);
                acc[tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq.z, breg[t % BR].z, acc[tm], 0, 0, 0);
                acc[tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq.w, breg[t % BR].w, acc[tm], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (!PING) {
                if (nt < NTAPS && LDSR) {
#pragma unroll
                    for (int tm = 0; tm < HT; ++tm) av[0][tm] = rd(tm);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int tm = HT; tm < TM; ++tm) {
                    const f32x4 aq = av[0][tm];
                    acc[tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq.x, breg[t % BR].x, acc[tm], 0, 0, 0);
                    acc[tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq.y, breg[t % BR].y, acc[tm], 0, 0, 0);
                    acc[tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq.z, breg[t % BR].z, acc[tm], 0, 0, 0);
                    acc[tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq.w, breg[t % BR].w, acc[tm], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (GLB) {
                int widx = (ch * NTAPS + t + BR) & 1023;
                if (GLB == 3) breg[t % BR] = wp[(t & 3) * 64];   // no address arithmetic at all: immediate offsets only
                else if (GLB == 2) {   // uniform base + per-lane 32-bit offset: the saddr form, no 64-bit VALU add
                    const float4* wu = w + (size_t)widx * 64;
                    breg[t % BR] = wu[lane];
                } else breg[t % BR] = wp[(size_t)widx * 64];
            }
            if (!PING && nt < NTAPS && LDSR) {
#pragma unroll
                for (int tm = HT; tm < TM; ++tm) av[0][tm] = rd(tm);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) s += acc[tm][0] + acc[tm][1] + acc[tm][2] + acc[tm][3];
    if (s == 12345.678f) out[0] = s;
}

template <int WAVES, int TM, int LDSR, int GLB, int BAR, int CS4 = 5, int WRAP = 10, int IMM = 0>
void run(const char* what, float* out, float4* w) {
    const int chunks = 400;
    const int nvox = 7 * 12 * 12;                       // one 5-plane slab with halo: 80 640 B
    const size_t lds = (size_t)nvox * CS4 * 16;
    const int grid = 256 * (WAVES == 4 ? 2 : 1);
    auto kern = k<WAVES, TM, LDSR, GLB, BAR, CS4, WRAP, IMM>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, 0, out, w, 20, nvox);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds, 0, out, w, chunks, nvox);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * WAVES * chunks * 27.0 * TM * 4 * 2048.0;
    printf("%-64s cs%d wrap%d w%d tm%d: %7.1f TFLOP/s (%5.1f %% of 157.3)\n", what, CS4 * 4, WRAP, WAVES, TM, flops / (ms * 1e-3) / 1e12, flops / (ms * 1e-3) / 1e12 / 1.573);
}

int main(int argc, char** argv) {
    float* out; float4* w;
    hipMalloc(&out, 4096);
    hipMalloc(&w, 1100 * 64 * sizeof(float4));
    hipMemset(w, 0, 1100 * 64 * sizeof(float4));
    if (argc > 2) {   // address arithmetic: immediates for the tap offsets, saddr form for the ring refills
        run<8, 8, 1, 0, 0, 5, 10, 0>("A reads, runtime tap offsets", out, w);
        run<8, 8, 1, 0, 0, 5, 10, 1>("A reads, immediate tap offsets", out, w);
        run<8, 8, 1, 1, 0, 5, 10, 0>("A reads + ring (64-bit vaddr)", out, w);
        run<8, 8, 1, 2, 0, 5, 10, 0>("A reads + ring (saddr)", out, w);
        run<8, 8, 1, 2, 0, 5, 10, 1>("A reads imm + ring (saddr)", out, w);
        run<8, 8, 1, 2, 2, 5, 10, 1>("A reads imm + ring (saddr) + 2 barriers", out, w);
        run<4, 4, 1, 0, 0, 5, 5, 0>("A reads, runtime tap offsets", out, w);
        run<4, 4, 1, 0, 0, 5, 5, 1>("A reads, immediate tap offsets", out, w);
        run<4, 4, 1, 1, 0, 5, 5, 0>("A reads + ring (64-bit vaddr)", out, w);
        run<4, 4, 1, 2, 0, 5, 5, 0>("A reads + ring (saddr)", out, w);
        run<4, 4, 1, 2, 0, 5, 5, 1>("A reads imm + ring (saddr)", out, w);
        run<4, 4, 1, 2, 2, 5, 5, 1>("A reads imm + ring (saddr) + 2 barriers", out, w);
        run<4, 4, 0, 2, 0, 5, 5, 1>("ring (saddr) only", out, w);
        run<4, 4, 0, 3, 0, 5, 5, 1>("ring, immediate addresses only", out, w);
        run<4, 4, 1, 3, 0, 5, 5, 1>("A reads imm + ring, immediate addresses", out, w);
        run<8, 8, 1, 3, 0, 5, 10, 1>("A reads imm + ring, immediate addresses", out, w);
        return 0;
    }
    if (argc > 1) {   // layout sweep: voxel stride and row wrap of the A image
        run<8, 8, 1, 0, 0, 4, 10>("A reads", out, w);
        run<8, 8, 1, 0, 0, 5, 10>("A reads", out, w);
        run<8, 8, 1, 0, 0, 6, 10>("A reads", out, w);
        run<8, 8, 1, 0, 0, 7, 10>("A reads", out, w);
        run<8, 8, 1, 0, 0, 9, 10>("A reads", out, w);
        run<8, 8, 1, 0, 0, 4, 12>("A reads, rows never wrap inside a tile", out, w);
        run<8, 8, 1, 0, 0, 5, 12>("A reads, rows never wrap inside a tile", out, w);
        run<8, 8, 1, 0, 0, 7, 12>("A reads, rows never wrap inside a tile", out, w);
        run<8, 8, 1, 0, 0, 9, 12>("A reads, rows never wrap inside a tile", out, w);
        run<4, 4, 1, 0, 0, 5, 5>("A reads, 5^3 rows", out, w);
        run<4, 4, 1, 0, 0, 7, 5>("A reads, 5^3 rows", out, w);
        run<4, 4, 1, 0, 0, 9, 5>("A reads, 5^3 rows", out, w);
        return 0;
    }
    run<4, 4, 0, 0, 0>("MFMAs only", out, w);
    run<4, 4, 1, 0, 0>("+ A reads (real pattern)", out, w);
    run<4, 4, 2, 0, 0>("+ A reads (conflict-free)", out, w);
    run<4, 4, 0, 1, 0>("+ ring refills", out, w);
    run<4, 4, 1, 1, 0>("+ A reads + ring refills", out, w);
    run<4, 4, 1, 1, 1>("+ A reads + ring refills + 1 barrier / chunk", out, w);
    run<4, 4, 1, 1, 2>("+ A reads + ring refills + 2 barriers / chunk", out, w);
    run<4, 8, 0, 0, 0>("MFMAs only", out, w);
    run<4, 8, 1, 0, 0>("+ A reads (real pattern)", out, w);
    run<4, 8, 1, 1, 0>("+ A reads + ring refills", out, w);
    run<4, 8, 1, 1, 2>("+ A reads + ring refills + 2 barriers / chunk", out, w);
    run<8, 8, 0, 0, 0>("MFMAs only", out, w);
    run<8, 8, 1, 0, 0>("+ A reads (real pattern)", out, w);
    run<8, 8, 2, 0, 0>("+ A reads (conflict-free)", out, w);
    run<8, 8, 1, 1, 0>("+ A reads + ring refills", out, w);
    run<8, 8, 1, 1, 2>("+ A reads + ring refills + 2 barriers / chunk", out, w);
    return 0;
}
