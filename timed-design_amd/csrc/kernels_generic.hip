// Generic (shape-agnostic) gfx950 kernels for every op of the closed Keras inference set
// (SURVEY.md Appendix A).  These are the always-correct paths: the planner prefers the fused
// MFMA convolution (conv_mfma.hip) and uses these for everything it cannot fuse
// (strided / dilated convolutions, 'same' pooling, Dense, global pooling, softmax, copies).
// All tensors are channels-last fp32 views (common.h TView).
#include "common.h"
#include "device_math.h"

#include <hip/hip_fp16.h>

namespace {

constexpr int kThreads = 256;

inline unsigned grid_for(int64_t total, int per_block = kThreads, int64_t cap = 256 * 16) {
    int64_t b = (total + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

// ---- input conversion: what Keras' predict() does first (cast to float32) --------------------
__global__ void k_convert_frames(const void* __restrict__ src, int dtype, int64_t nvox, int C, TView dst) {
    // one thread per (voxel, dst channel); dst.cs may exceed C (zero padded)
    const int64_t total = nvox * dst.cs;
    const int V = dst.D * dst.H * dst.W;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t gv = i / dst.cs;
        const int c = (int)(i - gv * dst.cs);
        const int64_t f = gv / V;
        const int v = (int)(gv - f * V);
        float val = 0.f;
        if (c >= dst.coff && c < dst.coff + C) {
            const int64_t si = gv * C + (c - dst.coff);
            switch (dtype) {
                case TH_F32: val = ((const float*)src)[si]; break;
                case TH_F64: val = (float)((const double*)src)[si]; break;
                case TH_U8: val = (float)((const unsigned char*)src)[si]; break;
                case TH_BOOL: val = ((const unsigned char*)src)[si] ? 1.f : 0.f; break;
                case TH_F16: val = __half2float(((const __half*)src)[si]); break;
            }
        }
        dst.p[f * dst.fs + (int64_t)v * dst.cs + c] = val;
    }
}

// ---- direct convolution: one thread per output element, co fastest ---------------------------
__global__ void k_conv3d_direct(int64_t n, TView in, TView out, ConvGeom g, const float* __restrict__ w,
                                const float* __restrict__ bias, PreOp pre, PostOps post) {
    const int Cout = out.C, Cin = in.C;
    const int64_t total = n * out.D * out.H * out.W * Cout;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int co = (int)(t % Cout); t /= Cout;
        const int x = (int)(t % out.W); t /= out.W;
        const int y = (int)(t % out.H); t /= out.H;
        const int z = (int)(t % out.D); t /= out.D;
        const int64_t f = t;
        float acc = bias ? bias[co] : 0.f;
        const float* inf = in.p + f * in.fs + in.coff;
        for (int a = 0; a < g.kd; ++a) {
            const int zi = z * g.sd + a * g.dd - g.pz;
            if (zi < 0 || zi >= in.D) continue;
            for (int b = 0; b < g.kh; ++b) {
                const int yi = y * g.sh + b * g.dh - g.py;
                if (yi < 0 || yi >= in.H) continue;
                for (int c = 0; c < g.kw; ++c) {
                    const int xi = x * g.sw + c * g.dw - g.px;
                    if (xi < 0 || xi >= in.W) continue;
                    const float* ip = inf + ((int64_t)(zi * in.H + yi) * in.W + xi) * in.cs;
                    const float* wp = w + ((int64_t)((a * g.kh + b) * g.kw + c) * Cin) * Cout + co;
                    for (int ci = 0; ci < Cin; ++ci) {
                        float v = ip[ci];
                        if (pre.scale) v = fmaf(v, pre.scale[ci], pre.shift[ci]);
                        v = th_act(v, pre.act, pre.alpha);
                        acc = fmaf(v, wp[(int64_t)ci * Cout], acc);
                    }
                }
            }
        }
        acc = th_post(acc, co, post);
        out.p[f * out.fs + ((int64_t)(z * out.H + y) * out.W + x) * out.cs + out.coff + co] = acc;
    }
}

// ---- pooling ------------------------------------------------------------------------------
__global__ void k_pool3d(int64_t n, TView in, TView out, ConvGeom g, int is_max) {
    const int C = out.C;
    const int64_t total = n * out.D * out.H * out.W * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int c = (int)(t % C); t /= C;
        const int x = (int)(t % out.W); t /= out.W;
        const int y = (int)(t % out.H); t /= out.H;
        const int z = (int)(t % out.D); t /= out.D;
        const int64_t f = t;
        const float* inf = in.p + f * in.fs + in.coff + c;
        float m = -INFINITY, s = 0.f;
        int cnt = 0;
        for (int a = 0; a < g.kd; ++a) {
            const int zi = z * g.sd + a - g.pz;
            if (zi < 0 || zi >= in.D) continue;
            for (int b = 0; b < g.kh; ++b) {
                const int yi = y * g.sh + b - g.py;
                if (yi < 0 || yi >= in.H) continue;
                for (int d = 0; d < g.kw; ++d) {
                    const int xi = x * g.sw + d - g.px;
                    if (xi < 0 || xi >= in.W) continue;
                    const float v = inf[((int64_t)(zi * in.H + yi) * in.W + xi) * in.cs];
                    m = fmaxf(m, v);
                    s += v;
                    ++cnt;
                }
            }
        }
        // Keras: average pooling with 'same' padding divides by the number of in-bounds cells
        out.p[f * out.fs + ((int64_t)(z * out.H + y) * out.W + x) * out.cs + out.coff + c] =
            is_max ? m : s / (float)(cnt > 0 ? cnt : 1);
    }
}

__global__ void k_eltwise(int64_t n, TView in, TView out, PostOps ops) {
    const int C = out.C, V = out.D * out.H * out.W;
    const int64_t total = n * V * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int c = (int)(t % C); t /= C;
        const int v = (int)(t % V); t /= V;
        const float x = in.p[t * in.fs + (int64_t)v * in.cs + in.coff + c];
        out.p[t * out.fs + (int64_t)v * out.cs + out.coff + c] = th_post(x, c, ops);
    }
}

__global__ void k_copy(int64_t n, TView in, TView out) {
    const int C = out.C, V = out.D * out.H * out.W;
    const int64_t total = n * V * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int c = (int)(t % C); t /= C;
        const int v = (int)(t % V); t /= V;
        out.p[t * out.fs + (int64_t)v * out.cs + out.coff + c] = in.p[t * in.fs + (int64_t)v * in.cs + in.coff + c];
    }
}

__global__ void k_add(int64_t n, TView a, TView b, TView out) {
    const int C = out.C, V = out.D * out.H * out.W;
    const int64_t total = n * V * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = i;
        const int c = (int)(t % C); t /= C;
        const int v = (int)(t % V); t /= V;
        out.p[t * out.fs + (int64_t)v * out.cs + out.coff + c] =
            a.p[t * a.fs + (int64_t)v * a.cs + a.coff + c] + b.p[t * b.fs + (int64_t)v * b.cs + b.coff + c];
    }
}

// ---- global average / max pooling: one wave per (frame, 64-channel group) ---------------------
__global__ void k_global_pool(int64_t n, TView in, TView out, int is_max) {
    const int C = in.C, V = in.D * in.H * in.W;
    const int64_t total = n * C;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t f = i / C;
        const float* p = in.p + f * in.fs + in.coff + c;
        float m = -INFINITY, s = 0.f;
        for (int v = 0; v < V; ++v) {
            const float x = p[(int64_t)v * in.cs];
            m = fmaxf(m, x);
            s += x;
        }
        out.p[f * out.fs + out.coff + c] = is_max ? m : s / (float)V;
    }
}

// ---- dense: one thread per (frame, out) -------------------------------------------------------
__global__ void k_dense(int64_t n, TView in, TView out, const float* __restrict__ w, const float* __restrict__ bias,
                        PostOps post) {
    const int F = in.C, O = out.C;
    const int64_t total = n * O;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(i % O);
        const int64_t f = i / O;
        const float* x = in.p + f * in.fs + in.coff;
        float acc = bias ? bias[o] : 0.f;
        for (int k = 0; k < F; ++k) acc = fmaf(x[k], w[(int64_t)k * O + o], acc);
        out.p[f * out.fs + out.coff + o] = th_post(acc, o, post);
    }
}

// ---- softmax over the last axis: one wavefront (64 lanes) per row -----------------------------
__global__ void k_softmax(int64_t rows, TView in, TView out) {
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    const int C = in.C, V = in.D * in.H * in.W;
    for (int64_t row = (int64_t)blockIdx.x * wpb + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * wpb) {
        const int64_t f = row / V;
        const int v = (int)(row - f * V);
        const float* x = in.p + f * in.fs + (int64_t)v * in.cs + in.coff;
        float m = -INFINITY;
        for (int c = lane; c < C; c += 64) m = fmaxf(m, x[c]);
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += expf(x[c] - m);
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        float* y = out.p + f * out.fs + (int64_t)v * out.cs + out.coff;
        for (int c = lane; c < C; c += 64) y[c] = expf(x[c] - m) / s;
    }
}

// ---- GlobalAveragePooling3D -> Softmax tail in ONE launch: one wavefront per frame -------------------------------------------
// (TIMED's head: reference README.md:252-258 "GlobalAveragePooling3D instead of Dense; softmax".)  Lane l owns channels l, l + 64, ...:
// it walks its channels' voxels in k_global_pool's order and the row maximum / sum of exponentials are reduced with the same
// lane-partial + xor-shuffle scheme as k_softmax, so logits AND probabilities are bit-identical to the two-kernel path; the logits
// are written too (TH_PREDICT_LOGITS reads them).
__global__ void __launch_bounds__(256) k_gap_softmax(int64_t n, TView in, TView logits, TView probs) {
    const int lane = threadIdx.x & 63;
    const int C = in.C, V = in.D * in.H * in.W;
    const int64_t f = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (f >= n) return;
    float x[8];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = lane + 64 * k;
        x[k] = -INFINITY;
        if (c < C) {
            const float* p = in.p + f * in.fs + in.coff + c;
            float s = 0.f;
            for (int v = 0; v < V; ++v) s += p[(int64_t)v * in.cs];
            x[k] = s / (float)V;
            logits.p[f * logits.fs + logits.coff + c] = x[k];
            m = fmaxf(m, x[k]);
        }
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (lane + 64 * k < C) s += expf(x[k] - m);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = lane + 64 * k;
        if (c < C) probs.p[f * probs.fs + probs.coff + c] = expf(x[k] - m) / s;
    }
}

// ---- [BN / activation]* -> GlobalAveragePooling3D -> Dense -> Softmax tail in ONE launch: one wavefront per frame ------------
// (DenseCPD's head: BatchNormalization -> ReLU -> GlobalAveragePooling3D -> Dense(20) -> Softmax; SURVEY Appendix A.)  Every stage
// keeps the arithmetic AND the order of the kernel it replaces — th_post per element as k_eltwise, the voxel-order sum / V of
// k_global_pool, k_dense's fmaf chain over the features, k_softmax's lane-partial + xor-shuffle reductions — so pooled values,
// logits and probabilities are bit-identical to the five-launch path (tested).  The pooled vector lives in the wave's LDS row.
__global__ void __launch_bounds__(256) k_tail_dense(int64_t n, TView in, PostOps pre, TView pooled, TView logits, TView probs,
                                                     const float* __restrict__ w, const float* __restrict__ bias, PostOps post,
                                                     int softmax) {
    extern __shared__ float tail_lds[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int F = in.C, V = in.D * in.H * in.W, O = logits.C;
    const int64_t f = (int64_t)blockIdx.x * (blockDim.x >> 6) + wv;
    if (f >= n) return;                                  // (no workgroup barrier below: a wave works alone on its frame)
    float* pw = tail_lds + (size_t)wv * F;
    for (int c = lane; c < F; c += 64) {
        const float* p = in.p + f * in.fs + in.coff + c;
        float s = 0.f;
        for (int v = 0; v < V; ++v) s += th_post(p[(int64_t)v * in.cs], c, pre);
        s = s / (float)V;
        pw[c] = s;
        pooled.p[f * pooled.fs + pooled.coff + c] = s;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float x[8];
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int o = lane + 64 * k;
        x[k] = -INFINITY;
        if (o < O) {
            float acc = bias ? bias[o] : 0.f;
            for (int j = 0; j < F; ++j) acc = fmaf(pw[j], w[(int64_t)j * O + o], acc);
            x[k] = th_post(acc, o, post);
            logits.p[f * logits.fs + logits.coff + o] = x[k];
            m = fmaxf(m, x[k]);
        }
    }
    if (!softmax) return;
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (lane + 64 * k < O) s += expf(x[k] - m);
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int o = lane + 64 * k;
        if (o < O) probs.p[f * probs.fs + probs.coff + o] = expf(x[k] - m) / s;
    }
}

// ---- synthetic frames generated on the device (bench: keeps 22 GB of input off PCIe) ----------
__device__ inline uint32_t mix32(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return (uint32_t)x;
}
__global__ void k_synth_frames(float* d, int64_t n, int side, int C, int atoms, uint64_t seed) {
    // one thread per (frame, atom): atomically max-free splat (values only accumulate, then clipped
    // by a second pass) — overlap handling: atomicAdd then clamp in k_clip.
    const int64_t total = n * atoms;
    const float g1 = expf(-0.5f / (0.6f * 0.6f));
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t f = i / atoms;
        const uint32_t h0 = mix32(seed * 0x9E3779B97F4A7C15ULL + (uint64_t)i * 4 + 0);
        const uint32_t h1 = mix32(seed * 0x9E3779B97F4A7C15ULL + (uint64_t)i * 4 + 1);
        const int z = h0 % side, y = (h0 >> 10) % side, x = (h0 >> 20) % side, c = h1 % C;
        float* fr = d + f * (int64_t)side * side * side * C;
        for (int a = -1; a <= 1; ++a)
            for (int b = -1; b <= 1; ++b)
                for (int e = -1; e <= 1; ++e) {
                    const int zz = z + a, yy = y + b, xx = x + e;
                    if (zz < 0 || zz >= side || yy < 0 || yy >= side || xx < 0 || xx >= side) continue;
                    float v = 1.f;
                    if (a) v *= g1;
                    if (b) v *= g1;
                    if (e) v *= g1;
                    atomicAdd(fr + ((int64_t)(zz * side + yy) * side + xx) * C + c, v);
                }
    }
}
__global__ void k_clip01(float* d, int64_t total) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x)
        d[i] = fminf(fmaxf(d[i], 0.f), 1.f);
}

}  // namespace

#define LAUNCH_CHECK()                                                                         \
    do {                                                                                       \
        hipError_t _e = hipGetLastError();                                                     \
        if (_e != hipSuccess) {                                                                \
            th_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
            return TH_EHIP;                                                                    \
        }                                                                                      \
    } while (0)

int launch_convert_frames(hipStream_t s, const void* src, int dtype, int64_t n, int V, int C, TView dst) {
    if (dtype < TH_F32 || dtype > TH_F16) TH_FAIL(TH_EINVAL, "unknown frame dtype %d", dtype);
    const int64_t nvox = n * V;
    hipLaunchKernelGGL(k_convert_frames, dim3(grid_for(nvox * dst.cs)), dim3(kThreads), 0, s, src, dtype, nvox, C, dst);
    LAUNCH_CHECK();
    return TH_OK;
}
int launch_conv3d_direct(hipStream_t s, int64_t n, TView in, TView out, ConvGeom g, const float* w, const float* bias,
                         PreOp pre, PostOps post) {
    hipLaunchKernelGGL(k_conv3d_direct, dim3(grid_for(n * out.V() * out.C, kThreads, 1 << 20)), dim3(kThreads), 0, s, n, in,
                       out, g, w, bias, pre, post);
    LAUNCH_CHECK();
    return TH_OK;
}
int launch_pool3d(hipStream_t s, int64_t n, TView in, TView out, ConvGeom g, int is_max) {
    hipLaunchKernelGGL(k_pool3d, dim3(grid_for(n * out.V() * out.C)), dim3(kThreads), 0, s, n, in, out, g, is_max);
    LAUNCH_CHECK();
    return TH_OK;
}
int launch_eltwise(hipStream_t s, int64_t n, TView in, TView out, PostOps ops) {
    hipLaunchKernelGGL(k_eltwise, dim3(grid_for(n * out.V() * out.C)), dim3(kThreads), 0, s, n, in, out, ops);
    LAUNCH_CHECK();
    return TH_OK;
}
int launch_global_pool(hipStream_t s, int64_t n, TView in, TView out, int is_max) {
    hipLaunchKernelGGL(k_global_pool, dim3(grid_for(n * in.C)), dim3(kThreads), 0, s, n, in, out, is_max);
    LAUNCH_CHECK();
    return TH_OK;
}
int launch_dense(hipStream_t s, int64_t n, TView in, TView out, const float* w, const float* bias, PostOps post) {
    hipLaunchKernelGGL(k_dense, dim3(grid_for(n * out.C)), dim3(kThreads), 0, s, n, in, out, w, bias, post);
    LAUNCH_CHECK();
    return TH_OK;
}
int launch_softmax(hipStream_t s, int64_t n, TView in, TView out) {
    hipLaunchKernelGGL(k_softmax, dim3(grid_for(n * in.V(), kThreads / 64)), dim3(kThreads), 0, s, n * in.V(), in, out);
    LAUNCH_CHECK();
    return TH_OK;
}
int launch_gap_softmax(hipStream_t s, int64_t n, TView in, TView logits, TView probs) {
    if (n <= 0) return TH_OK;
    hipLaunchKernelGGL(k_gap_softmax, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, n, in, logits, probs);
    LAUNCH_CHECK();
    return TH_OK;
}
int launch_tail_dense(hipStream_t s, int64_t n, TView in, PostOps pre, TView pooled, TView logits, TView probs, const float* w,
                      const float* bias, PostOps post, int softmax) {
    if (n <= 0) return TH_OK;
    const size_t lds = (size_t)4 * in.C * sizeof(float);
    hipLaunchKernelGGL(k_tail_dense, dim3((unsigned)((n + 3) / 4)), dim3(256), lds, s, n, in, pre, pooled, logits, probs, w, bias, post,
                       softmax);
    LAUNCH_CHECK();
    return TH_OK;
}
int launch_copy(hipStream_t s, int64_t n, TView in, TView out) {
    hipLaunchKernelGGL(k_copy, dim3(grid_for(n * out.V() * out.C)), dim3(kThreads), 0, s, n, in, out);
    LAUNCH_CHECK();
    return TH_OK;
}
int launch_add(hipStream_t s, int64_t n, TView a, TView b, TView out) {
    hipLaunchKernelGGL(k_add, dim3(grid_for(n * out.V() * out.C)), dim3(kThreads), 0, s, n, a, b, out);
    LAUNCH_CHECK();
    return TH_OK;
}
int launch_synth_frames(hipStream_t s, float* d, int64_t n, int side, int channels, int atoms, uint64_t seed) {
    const int64_t total = n * side * side * side * channels;
    HIP_TRY(hipMemsetAsync(d, 0, total * sizeof(float), s));
    hipLaunchKernelGGL(k_synth_frames, dim3(grid_for(n * atoms)), dim3(kThreads), 0, s, d, n, side, channels, atoms, seed);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_clip01, dim3(grid_for(total)), dim3(kThreads), 0, s, d, total);
    LAUNCH_CHECK();
    return TH_OK;
}
