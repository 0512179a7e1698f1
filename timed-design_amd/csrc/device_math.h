// Elementwise device helpers shared by every kernel: Keras activations and the fused epilogue.
#pragma once
#include "common.h"

// Keras semantics: ELU x>0 ? x : alpha*(exp(x)-1); LeakyReLU x>0 ? x : alpha*x.
__device__ __forceinline__ float th_act(float x, int act, float alpha) {
    switch (act) {
        case ACT_RELU: return fmaxf(x, 0.f);
        // exp via the hardware exp2 path (~2 ulp): |error| <= 1.2e-7 absolute on the negative branch
        case ACT_ELU: return x > 0.f ? x : alpha * (__expf(x) - 1.f);
        case ACT_LEAKY: return x > 0.f ? x : alpha * x;
        case ACT_SIGMOID: return 1.f / (1.f + expf(-x));
        case ACT_TANH: return tanhf(x);
        default: return x;
    }
}

// The same activation over N register values with the op decoded ONCE (op-outer, element-inner).  The scalar th_act above
// inlines its whole switch — exp, the sigmoid division, tanhf's libdevice body — at every call site: in a kernel that applies
// it to 48 prefetched values per tile (k_conv_pw2's BN -> ReLU prologue) that was 6 000 VALU + 3 500 SALU instructions per 96
// MFMAs, ~160 KB of code thrashing the instruction cache, and the kernel ran at 44 % of the matrix pipe WITH ALL LOADS AND
// STORES KNOCKED OUT (round 3, TH_PW_DBG=3).  Here only the taken case's N arithmetic instructions execute.
template <int N>
__device__ __forceinline__ void th_act_vec(float (&x)[N], int act, float alpha) {
    switch (act) {
        case ACT_RELU:
#pragma unroll
            for (int k = 0; k < N; ++k) x[k] = fmaxf(x[k], 0.f);
            break;
        case ACT_ELU:
#pragma unroll
            for (int k = 0; k < N; ++k) x[k] = x[k] > 0.f ? x[k] : alpha * (__expf(x[k]) - 1.f);
            break;
        case ACT_LEAKY:
#pragma unroll
            for (int k = 0; k < N; ++k) x[k] = x[k] > 0.f ? x[k] : alpha * x[k];
            break;
        case ACT_SIGMOID:
#pragma unroll
            for (int k = 0; k < N; ++k) x[k] = 1.f / (1.f + expf(-x[k]));
            break;
        case ACT_TANH:
#pragma unroll
            for (int k = 0; k < N; ++k) x[k] = tanhf(x[k]);
            break;
        default:
            break;
    }
}

// epilogue: ordered list of activation / per-channel affine (folded BatchNormalization:
// scale = gamma*rsqrt(var+eps), shift = beta - mean*scale — the form tf.nn.batch_normalization uses)
__device__ __forceinline__ float th_post(float x, int c, const PostOps& ops) {
#pragma unroll
    for (int i = 0; i < TH_MAX_POST; ++i) {
        if (i < ops.n) {
            if (ops.type[i] == POP_ACT) x = th_act(x, ops.act[i], ops.alpha[i]);
            else x = fmaf(x, ops.scale[i][c], ops.shift[i][c]);
        }
    }
    return x;
}

// Two values of one channel (the pooled outputs a lane owns after a pool-first epilogue): op list decoded once.
__device__ __forceinline__ void th_post2(float& x0, float& x1, int c, const PostOps& ops) {
    float x[2] = {x0, x1};
    for (int i = 0; i < ops.n; ++i) {
        if (ops.type[i] == POP_AFFINE) {
            const float sc = ops.scale[i][c], sh = ops.shift[i][c];
            x[0] = fmaf(x[0], sc, sh);
            x[1] = fmaf(x[1], sc, sh);
        } else {
            th_act_vec<2>(x, ops.act[i], ops.alpha[i]);
        }
    }
    x0 = x[0]; x1 = x[1];
}

// Same epilogue for the 16 accumulator values a lane holds for ONE output channel c: the op list is
// decoded once (op-outer, element-inner), so the per-element cost is the bare arithmetic
// (ELU: exp2-path exp + select; affine: one fma) instead of a switch per element.
__device__ __forceinline__ void th_post16(float (&x)[16], int c, const PostOps& ops) {
    for (int i = 0; i < ops.n; ++i) {
        if (ops.type[i] == POP_AFFINE) {
            const float sc = ops.scale[i][c], sh = ops.shift[i][c];
#pragma unroll
            for (int k = 0; k < 16; ++k) x[k] = fmaf(x[k], sc, sh);
        } else {
            th_act_vec<16>(x, ops.act[i], ops.alpha[i]);
        }
    }
}
