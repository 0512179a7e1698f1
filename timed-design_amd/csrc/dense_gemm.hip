// Dense over a batch of frames as one GEMM on the fp32 matrix pipe: out[n][O] = post(x[n][F] w[F][O] + bias) — the classifier behind
// Flatten (ProDCoNN's 1728 -> 96: SURVEY.md §8(a) P2a, the call served is reference predict.py:142).  k_dense (one thread per
// output, a serial fmaf chain over F) ran it at 0.015 of the pipe: 0.57 ms per 4096 frames, 5.7 % of ProDCoNN-synth.
//
// fp32 operands and fp32 accumulation like k_dense; only the order of the F additions differs (four contiguous quarters of F, each
// an MFMA chain, added in a fixed order).  Workgroup = 16 frames x all outputs (<= 128: up to 8 tiles of 16 columns), four waves
// each on a quarter of F with v_mfma_f32_16x16x4_f32; a lane reads its frame's features as float4 (the four k of a 16-k block it
// feeds the MFMA steps with) and the weight rows from L2 (w is F x O x 4 bytes, 0.66 MB for ProDCoNN; every workgroup reads all of it).
#include "common.h"
#include "device_math.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

struct DenseGemmArgs {
    int64_t n;
    const float* x; int64_t xfs; int F;
    const float* w; const float* bias; PostOps post;
    float* out; int64_t ofs; int ocoff; int O;
};

template <int NT>
__global__ void __launch_bounds__(256) k_dense_gemm(const DenseGemmArgs a) {
    __shared__ f32x4 red[3][NT][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    const int64_t f0 = (int64_t)blockIdx.x * 16;
    const int64_t fr = f0 + i16 < a.n ? f0 + i16 : a.n - 1;
    const float* const xr = a.x + fr * a.xfs;
    const int nb = (a.F + 15) >> 4;                                     // blocks of 16 features
    const int b0 = (nb * wave) >> 2, b1 = (nb * (wave + 1)) >> 2;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int col[NT];
    bool cok[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        cok[t] = 16 * t + i16 < a.O;
        col[t] = cok[t] ? 16 * t + i16 : 0;
    }
    // a ring of four 16-feature blocks: the loads of block b + 3 are issued before the MFMAs of block b (one wave per SIMD: without
    // the lookahead every block is a round trip to L2 / HBM in front of 4 NT MFMAs)
    struct Blk { float xs[4]; float bv[4][NT]; };
    Blk R[4];
    auto load = [&](int b, Blk& B) __attribute__((always_inline)) {
        const int k0 = 16 * b + 4 * kq;                                 // F is a multiple of 4: a float4 is whole or beyond the end
        const bool kok = k0 < a.F;
        const float4 xv = kok ? *reinterpret_cast<const float4*>(xr + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
        B.xs[0] = xv.x; B.xs[1] = xv.y; B.xs[2] = xv.z; B.xs[3] = xv.w;
        const float* const wr = a.w + (int64_t)(kok ? k0 : 0) * a.O;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < NT; ++t) B.bv[s][t] = (kok && cok[t]) ? wr[s * a.O + col[t]] : 0.f;
    };
    auto mma = [&](const Blk& B) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(B.xs[s], B.bv[s][t], acc[t], 0, 0, 0);
    };
#pragma unroll
    for (int j = 0; j < 3; ++j)
        if (b0 + j < b1) load(b0 + j, R[j]);
    for (int b = b0; b < b1; b += 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (b + j + 3 < b1) load(b + j + 3, R[(j + 3) & 3]);
            if (b + j < b1) mma(R[j]);
        }
    }
    if (wave) {
#pragma unroll
        for (int t = 0; t < NT; ++t) red[wave - 1][t][lane] = acc[t];
    }
    __syncthreads();
    if (wave) return;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        f32x4 v = acc[t];
#pragma unroll
        for (int q = 0; q < 3; ++q) v += red[q][t][lane];
        if (!cok[t]) continue;
        const float bv = a.bias ? a.bias[col[t]] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t f = f0 + 4 * kq + r;                          // accumulator row = frame
            if (f < a.n) a.out[f * a.ofs + a.ocoff + col[t]] = th_post(v[r] + bv, col[t], a.post);
        }
    }
}

}  // namespace

// plan time: F features (contiguous per frame, frames xfs floats apart), O outputs
// (F >= 512: below that k_dense costs microseconds, and the one-launch tails of kernels_generic.hip promise k_dense's bits)
bool dense_gemm_ok(int F, int O, int64_t xfs) { return F >= 512 && F % 4 == 0 && xfs % 4 == 0 && O >= 8 && O <= 128; }

int launch_dense_gemm(hipStream_t s, int64_t n, TView in, TView out, const float* w, const float* bias, PostOps post) {
    if (n <= 0) return TH_OK;
    if (!dense_gemm_ok(in.C, out.C, in.fs) || ((uintptr_t)(in.p + in.coff) % 16)) TH_FAIL(TH_EINVAL, "dense_gemm: %d features at stride %lld, %d outputs", in.C, (long long)in.fs, out.C);
    DenseGemmArgs a;
    a.n = n; a.x = in.p + in.coff; a.xfs = in.fs; a.F = in.C;
    a.w = w; a.bias = bias; a.post = post;
    a.out = out.p; a.ofs = out.fs; a.ocoff = out.coff; a.O = out.C;
    const int nt = (out.C + 15) / 16;
    const dim3 grid((unsigned)((n + 15) / 16)), block(256);
    switch (nt) {
        case 1: hipLaunchKernelGGL(k_dense_gemm<1>, grid, block, 0, s, a); break;
        case 2: hipLaunchKernelGGL(k_dense_gemm<2>, grid, block, 0, s, a); break;
        case 3: hipLaunchKernelGGL(k_dense_gemm<3>, grid, block, 0, s, a); break;
        case 4: hipLaunchKernelGGL(k_dense_gemm<4>, grid, block, 0, s, a); break;
        case 5: case 6: hipLaunchKernelGGL(k_dense_gemm<6>, grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL(k_dense_gemm<8>, grid, block, 0, s, a); break;
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) TH_FAIL(TH_EHIP, "k_dense_gemm launch failed: %s", hipGetErrorString(e));
    return TH_OK;
}
