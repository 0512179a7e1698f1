// Pointwise (1x1x1, stride 1) fused Conv3D for gfx950 — the DenseCPD/DenseNet bottleneck and
// transition convolutions (BN -> ReLU -> Conv1^3 [-> AvgPool 2]; SURVEY.md §8(d) DenseCPD-synth,
// reference README.md:242-258 for the model family, predict.py:142 for the call that runs it).
//
// A 1x1x1 convolution is a plain GEMM  Y[M, Cout] = post(pre(X[M, Cin]) . W + b)  over M = frames x
// voxels rows with arithmetic intensity ~ Cin*Cout/(2(Cin+Cout)) FLOP/B ~ 18 for 80 -> 64: it sits at
// the ridge of the fp32-MFMA / HBM roofline, so it is built as a STREAMING kernel, not as the
// haloed-brick implicit GEMM of conv_mfma.hip:
//   * no LDS staging of activations and no barrier in the loop: lane (j = l&31, h = l>>5) reads the
//     4 channels its MFMA k-slot contracts straight from global memory (16 B per lane, all Cin/8
//     loads of a 32-row tile issued back to back so every 128 B line is fetched once);
//   * the whole weight matrix sits in LDS in MFMA fragment order (one conflict-free ds_read_b128 per
//     4 MFMAs), loaded once per workgroup; workgroups are persistent and stride over the row tiles;
//   * the BN->ReLU prologue is applied to the loaded registers, bias + {act | BN-affine}* to the
//     accumulators, and a 2x2x2 max/avg pool is reduced in registers: GEMM rows are grouped 8
//     pool-mates per output voxel, which the 32x32 accumulator layout puts in 4 registers of lane l
//     and 4 of lane l^32 -> one cross-half exchange;
//   * stores are 128 B per (row, 32-channel tile): lanes 0..31 hold consecutive channels of one row.
// v_mfma_f32_32x32x2_f32, exact fp32.  Four 4-wave workgroups per CU hide the load latency.
#include "common.h"
#include "device_math.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "conv_pointwise_k.h"

// k_conv_pw2 instantiations compiled in conv_pointwise_b.hip: chunk-blocked output (blk), "BN-affine -> ReLU" epilogue (epi = 2)
const void* conv_pw2_extra(int blk, int epi, int kmi, int nti, int slot12);

namespace {

// [kmax index: 4, 8, 16][nt index: 1, 2, 4][pool]
const PwKernel kPw[3][3][3] = {
    {{k_conv_pw<4, 1, 0>, k_conv_pw<4, 1, 1>, k_conv_pw<4, 1, 2>},
     {k_conv_pw<4, 2, 0>, k_conv_pw<4, 2, 1>, k_conv_pw<4, 2, 2>},
     {k_conv_pw<4, 4, 0>, k_conv_pw<4, 4, 1>, k_conv_pw<4, 4, 2>}},
    {{k_conv_pw<8, 1, 0>, k_conv_pw<8, 1, 1>, k_conv_pw<8, 1, 2>},
     {k_conv_pw<8, 2, 0>, k_conv_pw<8, 2, 1>, k_conv_pw<8, 2, 2>},
     {k_conv_pw<8, 4, 0>, k_conv_pw<8, 4, 1>, k_conv_pw<8, 4, 2>}},
    {{k_conv_pw<16, 1, 0>, k_conv_pw<16, 1, 1>, k_conv_pw<16, 1, 2>},
     {k_conv_pw<16, 2, 0>, k_conv_pw<16, 2, 1>, k_conv_pw<16, 2, 2>},
     {k_conv_pw<16, 4, 0>, k_conv_pw<16, 4, 1>, k_conv_pw<16, 4, 2>}},
};
#define PW2_ROW(K, N) {k_conv_pw2<K, N, 0>, k_conv_pw2<K, N, 1>, k_conv_pw2<K, N, 2>}
const PwKernel kPw2[3][3][3] = {
    {PW2_ROW(4, 1), PW2_ROW(4, 2), PW2_ROW(4, 4)},
    {PW2_ROW(8, 1), PW2_ROW(8, 2), PW2_ROW(8, 4)},
    {PW2_ROW(16, 1), PW2_ROW(16, 2), {nullptr, nullptr, nullptr}},   // 16 x 4: no room for the second register set
};
// 12 slots for 72..96 input channels (most of DenseCPD's bottleneck layers): two register sets of 12 float4 leave room for
// three workgroups per CU where 16 slots allow two
const PwKernel kPw2_12[2][3] = {PW2_ROW(12, 1), PW2_ROW(12, 2)};
const int kPwKmax[3] = {4, 8, 16};
const int kPwNt[3] = {1, 2, 4};
constexpr size_t kPwLdsLimit = 64 * 1024;

}  // namespace

// plan encoding (ConvMfmaPlan): cfg = 300 + 3*kmax_index + nt_index, CI = 8*K8 (padded Cin), BN = 32*NT, bres = 4
bool conv_pw_plan(const TView& in, const TView& oc, const ConvGeom& g, int Cin, int Cout, int pool, ConvMfmaPlan* p) {
    const ThKnobs& kn = th_knobs_planning();
    if (kn.conv_nopw) return false;                             // A/B comparisons against conv_mfma
    p->knobs = &kn;
    if (g.kd != 1 || g.kh != 1 || g.kw != 1) return false;
    if (g.sd != 1 || g.sh != 1 || g.sw != 1 || g.dd != 1 || g.dh != 1 || g.dw != 1) return false;
    if (pool && (oc.D < 2 || oc.H < 2 || oc.W < 2)) return false;
    if (Cout > 128 || Cin < 1) return false;
    const int nti = Cout <= 32 ? 0 : (Cout <= 64 ? 1 : 2);
    const int NT = kPwNt[nti];
    const int K8 = (Cin + 7) / 8;
    const int kmi = K8 <= 4 ? 0 : (K8 <= 8 ? 1 : 2);
    const size_t lds = (size_t)K8 * NT * 64 * 16 + (size_t)K8 * 8 * 2 * 4;
    if (lds > kPwLdsLimit) return false;
    p->cfg = 300 + 3 * kmi + nti;
    p->CI = K8 * 8; p->CS = 0; p->BN = 32 * NT; p->nnb = 1; p->nchunks = 1; p->pool = pool; p->bres = 4;
    p->Dc = pool ? (oc.D / 2) * 2 : oc.D;
    p->Hc = pool ? (oc.H / 2) * 2 : oc.H;
    p->Wc = pool ? (oc.W / 2) * 2 : oc.W;
    p->FB = 1; p->ZB = p->Dc; p->nzb = 1; p->Zp = p->Dc; p->Hp = p->Hc; p->Wp = p->Wc;
    p->rows_pf = p->Dc * p->Hc * p->Wc;
    p->lds_bytes = lds;
    p->tab_off = 0;
    p->wpk_floats = (size_t)K8 * NT * 256;
    p->exec_flops = 2.0 * (double)p->rows_pf * (double)(32 * NT) * (double)(K8 * 8);
    char buf[224];
    // the pipelined kernel when the layer allows it (launch_conv_pw falls back to k_conv_pw for unaligned or > 4 GiB views)
    const bool pipe = !kn.pw_nopipe && kPw2[kmi][nti][pool] && Cin % 8 == 0 && K8 <= kPwKmax[kmi];
    const int kmax = (pipe && K8 > 8 && K8 <= 12 && nti <= 1) ? 12 : kPwKmax[kmi];
    snprintf(buf, sizeof buf, "conv_pw<k%d,nt%d,pool%d> K8=%d lds%zuK (streaming 1x1x1, weights in LDS%s) [k_conv_pw%s<%d,%d,%d%s>]",
             kmax, NT, pool, K8, lds / 1024, pipe ? ", next tile prefetched, buffer addressing" : "", pipe ? "2" : "",
             kmax, NT, pool, pipe ? ",0,0" : "");
    p->label = buf;
    (void)in;
    return true;
}

// Keras [1,1,1,Cin,Cout] -> fragment order [kk][ntile][h][j][q] with ci = kk*8 + 4h + q, co = ntile*32 + j
void conv_pw_pack_weights(const ConvMfmaPlan& p, int Cin, int Cout, const float* w, float* dst) {
    std::memset(dst, 0, p.wpk_floats * sizeof(float));
    const int K8 = p.CI / 8, NT = p.BN / 32;
    for (int kk = 0; kk < K8; ++kk)
        for (int nt = 0; nt < NT; ++nt) {
            float* frag = dst + ((size_t)kk * NT + nt) * 256;
            for (int h = 0; h < 2; ++h)
                for (int j = 0; j < 32; ++j) {
                    const int co = nt * 32 + j;
                    if (co >= Cout) continue;
                    for (int q = 0; q < 4; ++q) {
                        const int ci = kk * 8 + 4 * h + q;
                        if (ci < Cin) frag[(h * 32 + j) * 4 + q] = w[(size_t)ci * Cout + co];
                    }
                }
        }
}

// the "BN-affine -> ReLU" instantiation exists for this plan and chain (launch_conv_pw and the step label agree through this)
static bool pw_relu_epi(const ConvMfmaPlan& p, int K8, const PostOps& post) {
    const int idx = p.cfg - 300, kmi = idx / 3, nti = idx % 3;
    const bool relu_chain = post.n == 2 && post.type[0] == POP_AFFINE && post.type[1] == POP_ACT && post.act[1] == ACT_RELU;
    if (p.pool != 0 || !relu_chain || th_knobs_of(p.knobs).pw_noepi) return false;
    return conv_pw2_extra(0, 2, kmi, nti, (K8 > 8 && K8 <= 12 && nti <= 1) ? 1 : 0) != nullptr;
}
// the plan's label with the template arguments of the instantiation that runs (chunk-blocked output, epilogue chain)
std::string conv_pw_label(const ConvMfmaPlan& p, bool out_blk, const PostOps& post) {
    std::string l = p.label;
    const size_t k = l.rfind(",0,0>]");
    if (k == std::string::npos) return l;
    l[k + 1] = out_blk ? '1' : '0';
    l[k + 3] = pw_relu_epi(p, p.CI / 8, post) ? '2' : '0';
    return l;
}

int launch_conv_pw(hipStream_t s, int64_t n, const ConvMfmaPlan& p, TView in, TView out, int Cin, int Cout, const float* wpk,
                   const float* bias, PreOp pre, PostOps post) {
    const int idx = p.cfg - 300;
    if (idx < 0 || idx >= 9 || p.bres != 4) TH_FAIL(TH_EINVAL, "conv_pw: bad plan");
    const int kmi = idx / 3, nti = idx % 3;
    if (n <= 0) return TH_OK;
    ConvPwArgs a;
    std::memset(&a, 0, sizeof a);
    a.in = in.p; a.in_fs = in.fs; a.in_cs = in.cs; a.in_coff = in.coff; a.Cin = Cin; a.K8 = p.CI / 8;
    a.vec_ok = (in.cs % 4 == 0 && in.coff % 4 == 0 && in.fs % 4 == 0 && ((uintptr_t)in.p % 16) == 0) ? 1 : 0;
    a.V = in.D * in.H * in.W; a.H = in.H; a.W = in.W;
    a.in_dense = (in.fs == (int64_t)a.V * in.cs) ? 1 : 0;
    a.Vo = out.D * out.H * out.W; a.Ho = out.H; a.Wo = out.W;
    a.wpk = wpk; a.Cout = Cout; a.bias = bias; a.pre = pre; a.post = post;
    a.out = out.p; a.out_fs = out.fs; a.out_cs = out.cs; a.out_coff = out.coff;
    a.out_blk = out.blk;
    if (out.blk && (p.pool || out.blk != 4 || out.coff || out.cs != Cout || Cout % 4)) TH_FAIL(TH_EINVAL, "conv_pw: bad chunk-blocked output view");
    const int64_t out_v = p.pool ? a.Vo : a.V;
    a.out_dense = (out.fs == out_v * out.cs) ? 1 : 0;
    const int64_t nrows = p.pool ? n * a.Vo * 8 : n * a.V;
    if (nrows >= 0x7fffffffLL) TH_FAIL(TH_EINVAL, "conv_pw: %lld rows per launch exceed the 32-bit row index; lower the chunk size", (long long)nrows);
    a.nrows = (unsigned)nrows;
    a.ntiles = (unsigned)((nrows + 31) / 32);
    a.dbg = th_knobs_of(p.knobs).pw_dbg;
    const unsigned want = (a.ntiles + 3) / 4;
    const unsigned grid = std::min(want, 256u * 8u);   // persistent: up to 8 workgroups per CU queued, waves stride over tiles
    PwKernel k = kPw[kmi][nti][p.pool];
    // the pipelined buffer-addressing kernel when the views allow it (see k_conv_pw2)
    const int64_t in_span = ((n - 1) * in.fs + (int64_t)(a.V - 1) * in.cs + in.coff + Cin) * 4;
    const int64_t out_span = ((n - 1) * out.fs + (out_v - 1) * out.cs + out.coff + Cout) * 4;
    const bool no_pw2 = th_knobs_of(p.knobs).pw_nopipe != 0;   // A/B comparisons
    if (!no_pw2 && kPw2[kmi][nti][p.pool] && a.vec_ok && Cin % 8 == 0 && a.K8 <= kPwKmax[kmi] && in_span < 0xfffffff0LL &&
        out_span < 0xfffffff0LL) {
        a.in_bytes = (unsigned)in_span; a.out_bytes = (unsigned)out_span;
        k = (a.K8 > 8 && a.K8 <= 12 && nti <= 1) ? kPw2_12[nti][p.pool] : kPw2[kmi][nti][p.pool];
        const int slot12 = (a.K8 > 8 && a.K8 <= 12 && nti <= 1) ? 1 : 0;
        if (out.blk) k = (PwKernel)conv_pw2_extra(1, 0, kmi, nti, slot12);
        if (pw_relu_epi(p, a.K8, post)) {
            const int bk = out.blk ? 1 : 0;
            PwKernel kr = (PwKernel)conv_pw2_extra(bk, 2, kmi, nti, slot12);
            if (kr) k = kr;
        }
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), p.lds_bytes + (out.blk ? 4 * 32 * 36 * 4 + 16 : 0), s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) TH_FAIL(TH_EHIP, "conv_pw launch failed: %s (%s)", hipGetErrorString(e), p.label.c_str());
    return TH_OK;
}
