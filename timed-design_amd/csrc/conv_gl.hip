// Small-volume convolutions as an implicit GEMM whose operands come straight from L2 — no LDS staging.  For the layers the LDS-tiled
// kernels serve badly: strided ones (ProDCoNN's 3x3x3 stride-2 on 10^3 x 32 ran at 0.13 of the fp32 pipe on k_conv_mfma) and
// 'valid' ones with a few dozen outputs per frame.  SURVEY.md §8(a) P2a; the call served is reference predict.py:142.
//
// GEMM rows are output voxels numbered across the whole batch (row = frame x Vo + voxel: tiles do not stop at a frame's end), a wave
// owns 32 rows (two 16-row tiles; one on small layers) x all output channels (<= 8 tiles of 16 columns) and walks K = taps x 16-channel blocks with
// v_mfma_f32_16x16x4_f32.  Per block a lane reads ONE float4 per row tile — channels 4 kq .. 4 kq + 3 of its row's input voxel
// under the tap (a row's 16 channels are 64 contiguous bytes, padding reads as zero by predication; an input
// prologue BN -> activation is applied to the loaded values in registers) — and one float4 per column
// tile from the prepacked weights (lane-contiguous, 1 KB per wave load); element s of both feeds MFMA step s, so the k order inside
// a block is (kq, s) on both sides.  A frame's input (128 KB at most here) is read 27/8 times by the four waves that share it and
// the weights by every wave: both live in L2 / L1.  Three blocks are in flight per wave and several waves per SIMD cover the rest.
// fp32 products, fp32 accumulation: the arithmetic of k_conv_mfma, another order of the K additions.
#include "common.h"
#include "device_math.h"

#include <cstring>

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

struct ConvGlArgs {
    const float* in; int64_t in_fs; int in_cs; int D, H, W;
    int Do, Ho, Wo;
    ConvGeom g;
    int ncb;                          // 16-channel blocks of Cin
    const float4* wpk;                // [tap][block][column tile][lane] x 4
    const float* bias; PostOps post;
    PreOp pre;                        // BN -> activation in front of the convolution (DenseCPD), applied to real voxels only
    float* out; int64_t out_fs; int out_cs, out_coff, Cout;
    int64_t rows;                     // frames x Do Ho Wo
};

template <int NT, bool PRE, int MT>
__global__ void __launch_bounds__(256) k_conv_gl(const ConvGlArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    const int64_t tile0 = ((int64_t)blockIdx.x * 4 + wave) * MT;
    const int Vo = a.Do * a.Ho * a.Wo, HWo = a.Ho * a.Wo;
    if (tile0 * 16 >= a.rows) return;
    const float* pb[MT];
    int iz0[MT], iy0[MT], ix0[MT];
    bool rok[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int64_t r = (tile0 + m) * 16 + i16;
        rok[m] = r < a.rows;
        const int64_t rr = rok[m] ? r : a.rows - 1;
        const int64_t f = rr / Vo;
        const int v = (int)(rr - f * Vo);
        const int oz = v / HWo, oy = (v - oz * HWo) / a.Wo, ox = v - oz * HWo - oy * a.Wo;
        iz0[m] = oz * a.g.sd - a.g.pz; iy0[m] = oy * a.g.sh - a.g.py; ix0[m] = ox * a.g.sw - a.g.px;
        pb[m] = a.in + f * a.in_fs + 4 * kq;
    }
    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[m][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    struct Blk { float4 x[MT]; float4 w[NT]; float4 sc, sh; bool ok[MT]; };
    Blk R[3];
    int dz = 0, dy = 0, dx = 0, cb = 0;                      // the block the next load() fetches (wave-uniform)
    const float4* wp = a.wpk + lane;
    auto load = [&](Blk& B) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int iz = iz0[m] + dz * a.g.dd, iy = iy0[m] + dy * a.g.dh, ix = ix0[m] + dx * a.g.dw;
            const bool ok = rok[m] && iz >= 0 && iz < a.D && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            B.x[m] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) B.x[m] = *reinterpret_cast<const float4*>(pb[m] + (int64_t)((iz * a.H + iy) * a.W + ix) * a.in_cs + 16 * cb);
            if (PRE) B.ok[m] = ok;
        }
        if (PRE && a.pre.scale) {
            B.sc = *reinterpret_cast<const float4*>(a.pre.scale + 16 * cb + 4 * kq);
            B.sh = *reinterpret_cast<const float4*>(a.pre.shift + 16 * cb + 4 * kq);
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) B.w[t] = wp[t * 64];
        wp += NT * 64;
        if (++cb == a.ncb) {
            cb = 0;
            if (++dx == a.g.kw) {
                dx = 0;
                if (++dy == a.g.kh) { dy = 0; ++dz; }
            }
        }
    };
    auto mma = [&](const Blk& B) __attribute__((always_inline)) {
        float xs[MT][4];
#pragma unroll
        for (int m = 0; m < MT; ++m) { xs[m][0] = B.x[m].x; xs[m][1] = B.x[m].y; xs[m][2] = B.x[m].z; xs[m][3] = B.x[m].w; }
        if (PRE) {                                           // the padding of the convolution is a padding of the ACTIVATED tensor: zeros stay zeros
            const float sc[4] = {B.sc.x, B.sc.y, B.sc.z, B.sc.w}, sh[4] = {B.sh.x, B.sh.y, B.sh.z, B.sh.w};
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (a.pre.scale) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) xs[m][k] = fmaf(xs[m][k], sc[k], sh[k]);
                }
                th_act_vec<4>(xs[m], a.pre.act, a.pre.alpha);
#pragma unroll
                for (int k = 0; k < 4; ++k) xs[m][k] = B.ok[m] ? xs[m][k] : 0.f;
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float ws = s == 0 ? B.w[t].x : s == 1 ? B.w[t].y : s == 2 ? B.w[t].z : B.w[t].w;
#pragma unroll
                for (int m = 0; m < MT; ++m) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xs[m][s], ws, acc[m][t], 0, 0, 0);
            }
    };
    const int nblk = a.g.kd * a.g.kh * a.g.kw * a.ncb;
#pragma unroll
    for (int j = 0; j < 2; ++j)
        if (j < nblk) load(R[j]);
    for (int b = 0; b < nblk; b += 3) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (b + j + 2 < nblk) load(R[(j + 2) % 3]);
            if (b + j < nblk) mma(R[j]);
        }
    }
    // accumulator: column = i16, rows 4 kq + r of the tile
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int co = 16 * t + i16;
        if (co >= a.Cout) continue;
        const float bv = a.bias ? a.bias[co] : 0.f;
        float y[4 * MT];                                     // the lane's values of channel co: the epilogue chain decoded once for all
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) y[4 * m + r] = acc[m][t][r] + bv;
        for (int i = 0; i < a.post.n; ++i) {
            if (a.post.type[i] == POP_AFFINE) {
                const float sc = a.post.scale[i][co], sh = a.post.shift[i][co];
#pragma unroll
                for (int k = 0; k < 4 * MT; ++k) y[k] = fmaf(y[k], sc, sh);
            } else {
                th_act_vec<4 * MT>(y, a.post.act[i], a.post.alpha[i]);
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = (tile0 + m) * 16 + 4 * kq + r;
                if (row >= a.rows) continue;
                const int64_t f = row / Vo;
                const int v = (int)(row - f * Vo);
                a.out[f * a.out_fs + (int64_t)v * a.out_cs + a.out_coff + co] = y[4 * m + r];
            }
    }
}

int nt_of(int Cout) {
    const int nt = (Cout + 15) / 16;
    return nt <= 4 ? nt : nt <= 6 ? 6 : 8;
}

}  // namespace

// plan time (channels-last input of Cin channels at voxel stride in_cs, first channel in_coff, frames in_fs floats apart)
bool conv_gl_ok(int Cin, int Cout, int in_cs, int in_coff, int64_t in_fs) {
    return Cin >= 16 && Cin % 16 == 0 && Cout >= 1 && Cout <= 128 && in_cs % 4 == 0 && in_coff % 4 == 0 && in_fs % 4 == 0;
}
size_t conv_gl_wpk_floats(const ConvGeom& g, int Cin, int Cout) { return (size_t)g.kd * g.kh * g.kw * (Cin / 16) * nt_of(Cout) * 64 * 4; }
// MFMA FLOPs issued per frame (column tiles rounded up; row tiles counted over the batch, i.e. exact per frame)
double conv_gl_exec_flops(const ConvGeom& g, int Cin, int Cout, int Vo) { return 2.0 * Vo * g.kd * g.kh * g.kw * Cin * 16.0 * nt_of(Cout); }
std::string conv_gl_label(int Cout) {
    char buf[200];
    snprintf(buf, sizeof buf, "conv_gl<%d column tiles> rows across the batch, operands from L2 (no LDS), three 16-channel blocks in flight (16x16x4 fp32 MFMA) [k_conv_gl]", nt_of(Cout));
    return buf;
}
// Keras [kd][kh][kw][Cin][Cout] -> [tap][block][column tile][lane = 16 kq + col][s]: channel 16 block + 4 kq + s, output 16 tile + col
void conv_gl_pack_weights(const ConvGeom& g, int Cin, int Cout, const float* w, float* dst) {
    const int taps = g.kd * g.kh * g.kw, ncb = Cin / 16, nt = nt_of(Cout);
    std::memset(dst, 0, conv_gl_wpk_floats(g, Cin, Cout) * sizeof(float));
    for (int tap = 0; tap < taps; ++tap)
        for (int cb = 0; cb < ncb; ++cb)
            for (int t = 0; t < nt; ++t)
                for (int lane = 0; lane < 64; ++lane)
                    for (int s = 0; s < 4; ++s) {
                        const int c = 16 * cb + 4 * (lane >> 4) + s, co = 16 * t + (lane & 15);
                        if (co < Cout) dst[((((size_t)tap * ncb + cb) * nt + t) * 64 + lane) * 4 + s] = w[((size_t)tap * Cin + c) * Cout + co];
                    }
}

int launch_conv_gl(hipStream_t s, int64_t n, TView in, TView out, ConvGeom g, int Cin, int Cout, const float* wpk, const float* bias,
                   PreOp pre, PostOps post) {
    if (n <= 0) return TH_OK;
    if (in.blk || out.blk || !conv_gl_ok(Cin, Cout, in.cs, in.coff, in.fs) || ((uintptr_t)in.p % 16))
        TH_FAIL(TH_EINVAL, "conv_gl: %d -> %d channels, input stride %d offset %d", Cin, Cout, in.cs, in.coff);
    ConvGlArgs a;
    std::memset(&a, 0, sizeof a);
    a.in = in.p + in.coff; a.in_fs = in.fs; a.in_cs = in.cs; a.D = in.D; a.H = in.H; a.W = in.W;
    a.Do = out.D; a.Ho = out.H; a.Wo = out.W;
    a.g = g; a.ncb = Cin / 16;
    a.wpk = reinterpret_cast<const float4*>(wpk);
    a.bias = bias; a.post = post; a.pre = pre;
    a.out = out.p; a.out_fs = out.fs; a.out_cs = out.cs; a.out_coff = out.coff; a.Cout = Cout;
    a.rows = n * (int64_t)out.V();
    // row tiles per wave: two share every weight fragment; ONE when the layer is so small that two leave SIMDs short of waves
    // (ProDCoNN's 'valid' layer, 3 456 such waves: 0.262 -> 0.247 ms with one; its strided layer, 16 000: 0.528 -> 0.586 ms with one)
    const int mt = (a.rows + 31) / 32 < 8 * 1024 ? 1 : 2;
    const int64_t waves = (a.rows + 16 * mt - 1) / (16 * mt);
    const dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    const bool has_pre = pre.scale || pre.act != ACT_LINEAR;
#define GL_LAUNCH(NT_) do { if (has_pre) { if (mt == 1) hipLaunchKernelGGL((k_conv_gl<NT_, true, 1>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_conv_gl<NT_, true, 2>), grid, block, 0, s, a); } \
                            else { if (mt == 1) hipLaunchKernelGGL((k_conv_gl<NT_, false, 1>), grid, block, 0, s, a); else hipLaunchKernelGGL((k_conv_gl<NT_, false, 2>), grid, block, 0, s, a); } } while (0)
    switch (nt_of(Cout)) {
        case 1: GL_LAUNCH(1); break;
        case 2: GL_LAUNCH(2); break;
        case 3: GL_LAUNCH(3); break;
        case 4: GL_LAUNCH(4); break;
        case 6: GL_LAUNCH(6); break;
        default: GL_LAUNCH(8); break;
    }
#undef GL_LAUNCH
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) TH_FAIL(TH_EHIP, "k_conv_gl launch failed: %s", hipGetErrorString(e));
    return TH_OK;
}
