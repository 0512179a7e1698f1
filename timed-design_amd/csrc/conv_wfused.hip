// Fused Winograd convolution for 3x3x3 'same' stride-1 layers on volumes with EVEN in-plane extent — DenseCPD's growth
// convolutions (64 -> 16 at 10^3) and TIMED's conv3d_1 (32 -> 64 at 10^3, + MaxPool) — SURVEY.md §8(a) P2a; the call it
// serves is reference predict.py:142, the layer semantics SURVEY Appendix A.
//
// F(2,3) x F(2,3) in the two in-plane axes, the three z taps summed directly: a 2 x 2 output tile needs 16 products per
// (z tap, ci, co) instead of 36 — 2.25x fewer multiply-adds than the direct form, all in fp32 (BT and AT hold 0 / +-1 only,
// the halves of G are folded into the weights on the host in double precision).  Unlike conv_wino.hip (5^3 volumes, wide
// layers, V and M through HBM) NOTHING of the transform domain leaves the CU here:
//
//   R  [4 channels][z y rows + a zero row][x + 1]      the raw 4-channel slice of the frame (BN -> act prologue applied)
//   B  [a 4][dz 3][lane 64] x float4(b)               the stage's transformed weights, shared by the 8 waves
//   V  [ci 4][a 4][rec = z * NT + tile] x float4(b)   V = BT d BT^T of every 4 x 4 patch, double-buffered (2 x 64 KB)
//   M  accumulators: 16 positions x 2 row tiles of 16 (z, tile) rows x 16 output channels per wave (128 AGPRs)
//
// One workgroup = 8 waves = one frame x one block of 16 output channels, persistent over (frame, block) units.  Per 4-channel
// chunk ("stage") a wave issues 96 v_mfma_f32_16x16x4_f32: lane (i, q) of the A operand reads ONE ds_read_b128 = positions
// (a, 0..3) of channel q of row i, i.e. the operands of 4 MFMAs into 4 different accumulators; the V planes are laid out
// [q][a][rec] so that the 16-byte slot index mod 16 of a lane is rec mod 16 — conflict-free for any 16 consecutive rows
// whatever the hardware's lane grouping (each b128 lane group holds every i once).
//
// The stages are software-pipelined across the whole persistent loop (also across frames): while stage s multiplies out of
// V[s & 1], the raw slice of stage s + 1 (global loads issued during stage s - 1) is written to R, transformed and written to
// V[(s + 1) & 1], and the global loads of stage s + 2 are issued.  Two barriers per stage, no serial staging phase.
// The inverse transform (AT M AT^T: 16 accumulators of a lane -> its 2 x 2 outputs), bias, epilogue chain, optional 2^3
// pooling (the 2 x 2 outputs ARE the in-plane pool window; the z mate is the neighbouring accumulator register) and the
// stores happen in registers once per unit.
#include "common.h"
#include "device_math.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

namespace {

constexpr size_t kWfLdsLimit = 160 * 1024;

struct ConvWfArgs {
    const float* in; int64_t in_fs; int in_cs, in_coff;
    int in_blk;                       // chunk-blocked input (TView::blk): the slice of chunk c is the contiguous run [c][voxel][4]
    int Cin, nchunks;
    const float* wpk;                 // [cb][chunk][a][dz][lane = 16 (ci & 3) + (co & 15)][b]
    int Cout, ncb;
    const float* bias;
    PreOp pre;
    PostOps post;
    float* out; int64_t out_fs; int out_cs, out_coff;
    int64_t nframes;
    unsigned nslots;                  // ceil(nframes / 8) * 8 * ncb logical units (see unit_of)
};

// BT of F(2,3) along one axis: (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
__device__ __forceinline__ void wf_bt(float d0, float d1, float d2, float d3, float& o0, float& o1, float& o2, float& o3) {
    o0 = d0 - d2; o1 = d1 + d2; o2 = d2 - d1; o3 = d1 - d3;
}

// unit done: inverse transform (AT M AT^T on a lane's 16 accumulators), bias, epilogue chain, (pool,) store, accumulators cleared.
// C layout of v_mfma_f32_16x16x4_f32: col = i16, row = 4 q + r.
template <int D, int H, int W, int POOL>
__device__ __forceinline__ void wf_epilogue(const ConvWfArgs& a, f32x4 (&acc)[2][16], int64_t f, int cb, bool uok, int wave, int i16, int q) {
    constexpr int TY = H / 2, TX = W / 2, NT = TY * TX, NR = D * NT;
    (void)TY;
    const int co = cb * 16 + i16;
    const bool cok = uok && co < a.Cout;
    const int cc = co < a.Cout ? co : 0;
    const float bv = a.bias ? a.bias[cc] : 0.f;
    float* const outb = a.out + f * a.out_fs + a.out_coff + cc;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        float y[16];                             // [r][o][p]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float sm[4][2];
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) {
                const float m0 = acc[rt][4 * aa][r], m1 = acc[rt][4 * aa + 1][r], m2 = acc[rt][4 * aa + 2][r], m3 = acc[rt][4 * aa + 3][r];
                sm[aa][0] = (m0 + m1) + m2;
                sm[aa][1] = (m1 - m2) - m3;
            }
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                y[4 * r + p] = ((sm[0][p] + sm[1][p]) + sm[2][p]) + bv;
                y[4 * r + 2 + p] = ((sm[1][p] - sm[2][p]) - sm[3][p]) + bv;
            }
        }
#pragma unroll
        for (int p = 0; p < 16; ++p) acc[rt][p] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const int row0 = 32 * wave + 16 * rt + 4 * q;
        if (POOL == 0) {
            th_post16(y, cc, a.post);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + r;
                if (!cok || row >= NR) continue;
                const int z = row / NT, tile = row % NT, ty = tile / TX, tx = tile % TX;
                float* o = outb + (int64_t)((z * H + 2 * ty) * W + 2 * tx) * a.out_cs;
                o[0] = y[4 * r];
                o[a.out_cs] = y[4 * r + 1];
                o[(int64_t)W * a.out_cs] = y[4 * r + 2];
                o[(int64_t)(W + 1) * a.out_cs] = y[4 * r + 3];
            }
        } else {
            // rows (r = 0, 1) and (2, 3) are the z mates of one pooled voxel; its 2 x 2 in-plane window is the tile
            if (!(POOL == 1 && a.post.monotone)) th_post16(y, cc, a.post);
            float pv[2];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const float* yy = y + 8 * h2;
                if (POOL == 1) pv[h2] = fmaxf(fmaxf(fmaxf(yy[0], yy[1]), fmaxf(yy[2], yy[3])), fmaxf(fmaxf(yy[4], yy[5]), fmaxf(yy[6], yy[7])));
                else pv[h2] = (((yy[0] + yy[1]) + (yy[2] + yy[3])) + ((yy[4] + yy[5]) + (yy[6] + yy[7]))) * 0.125f;
            }
            if (POOL == 1 && a.post.monotone) th_post2(pv[0], pv[1], cc, a.post);
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const int row = row0 + 2 * h2;
                if (!cok || row >= NR) continue;
                const int pr = row >> 1, zp = pr / NT, tile = pr % NT;
                outb[(int64_t)(zp * NT + tile) * a.out_cs] = pv[h2];
            }
        }
    }
}

// POOL: 0 none, 1 max 2x2x2, 2 average 2x2x2.  PRE: prologue on the staged input — 0 none, 1 BN-affine -> ReLU (DenseNet-style
// pre-activation layers), 2 generic (optional affine, any activation: op decoded per stage)
// DBG: timing knock-outs (TH_WF_DBG, results are WRONG): 1 no transform, 2 no slice loads / R writes, 4 no weight traffic, 8 no barriers;
// 64 no V writes, 128 R reads replaced by loop invariants (the compiler then hoists the whole transform), 256 every slice from frame 0; 32 (results right): no MFMA / other interleave request inside the slots
template <int D, int H, int W, int POOL, int PRE, int DBG = 0>
__global__ void __launch_bounds__(512, 1) k_conv_wf(const ConvWfArgs a) {
    constexpr int TY = H / 2, TX = W / 2, NT = TY * TX, NR = D * NT, NZ = NR;
    // R: one plane per channel of D H real rows + ONE zero row, each W + 2 floats (x = -1 .. W; the two halo columns are never
    // written); patch rows above / below the frame's y range are pointed at the zero row.  4 dump floats behind every plane.
    constexpr int RX = W + 2, RROWS = D * H + 1, RPL = RROWS * RX + 4, NV = D * H * W;
    constexpr int kVB = 16 * 256;                      // float4 per V buffer
    static_assert(H % 2 == 0 && W % 2 == 0, "in-plane tiles are 2 x 2");
    static_assert(NR <= 250, "256 rows per workgroup: the zero record and the dump records live above the real ones");
    static_assert(NV <= 1024, "two voxels per thread");
    static_assert(POOL == 0 || D % 2 == 0, "z pooling pairs");
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    float4* const V4 = smem;                                            // [2][16 planes][256 records]
    float4* const B4 = smem + 2 * kVB;                                  // [a 4][dz 3][lane 64] + 1 dump: this stage's weight fragments
    float* const R1 = reinterpret_cast<float*>(smem + 2 * kVB + 769);   // [4 channels][RPL]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, q = lane >> 4;

    // ---- units: (frame, column block).  Frames are dealt to the 8 XCDs round-robin and the column blocks of a frame occupy
    // consecutive slots of ONE XCD (workgroup b runs on XCD b & 7): its L2 serves the frame to all of them.
    const unsigned G = gridDim.x;                                       // multiple of 8
    const unsigned b = blockIdx.x;
    if (b >= a.nslots) return;
    const int my_units = (int)((a.nslots - b + G - 1) / G);
    auto unit_of = [&](int k, int64_t& f, int& cb, bool& ok) {           // k-th unit of this workgroup
        const unsigned u = b + (unsigned)k * G;
        const unsigned xcd = u & 7u, j = u >> 3;
        cb = (int)(j % (unsigned)a.ncb);
        const int64_t ff = (int64_t)(j / (unsigned)a.ncb) * 8 + xcd;
        ok = ff < a.nframes;
        f = ok ? ff : a.nframes - 1;
    };
    const int S = my_units * a.nchunks;

    // ---- zero the whole LDS image once: the halo of R, the zero record of every V plane
    {
        constexpr int kTot = 2 * kVB + 769 + (4 * RPL + 3) / 4;
        for (int k = tid; k < kTot; k += 512) smem[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    // ---- per-thread constants -------------------------------------------------------------------------------------
    // load side: voxels tid and tid + 512 of the frame
    int goff[2], rdst[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int v = tid + 512 * j;
        const bool ok = v < NV;
        const int vv = ok ? v : 0;
        const int z = vv / (H * W), y = (vv / W) % H, x = vv % W;
        goff[j] = vv * (a.in_blk ? 4 : a.in_cs);
        rdst[j] = ok ? (z * H + y) * RX + x + 1 : RROWS * RX;
    }
    // transform side: record tid & 255, channels 2 (tid >> 8) and 2 (tid >> 8) + 1
    const int trec = tid & 255, th = tid >> 8;
    int roff[4], wdst;                  // the four patch rows (floats from the channel plane's start), the V record
    // record numbering: z * NT + tile; with a pool the two planes of a z pair interleave — (z >> 1) * 2 NT + 2 tile + (z & 1) — so
    // that the 16 rows of an A tile (8 tiles x the z pair, the order the pooled epilogue wants) are 16 CONSECUTIVE records
    // (SQ_LDS_BANK_CONFLICT: 38 % of the LDS-active cycles with the plain numbering, 27 % with this one, 25 % in the unpooled
    // instantiation; LDS busy 41 -> 35 % of the kernel — and the same 2.23 ms: the kernel is not LDS-bound)
    auto rec_of = [&](int z, int tile) { return POOL ? (z >> 1) * (2 * NT) + 2 * tile + (z & 1) : z * NT + tile; };
    {
        const bool ok = trec < NR;
        const int rr = ok ? trec : 0;
        const int z = POOL ? 2 * (rr / (2 * NT)) + (rr & 1) : rr / NT, tile = POOL ? (rr % (2 * NT)) >> 1 : rr % NT, ty = tile / TX, tx = tile % TX;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int y = 2 * ty - 1 + i;
            roff[i] = 2 * th * RPL + ((y >= 0 && y < H) ? z * H + y : D * H) * RX + 2 * tx;
        }
        wdst = th * 8 * 256 + (ok ? trec : 255);
    }
    // A side: rows 32 wave + 16 rt + i16 (POOL: rows are ordered (z pair, tile, z low) so that a lane's registers r = 0,1 and
    // 2,3 are z mates); offsets in float4 units inside a V buffer, minus / centre / plus z tap
    int abase[2][3];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int row = 32 * wave + 16 * rt + i16;
        const bool ok = row < NR;
        int z, tile;
        if (POOL) { const int pr = row >> 1; z = 2 * (pr / NT) + (row & 1); tile = pr % NT; }
        else { z = row / NT; tile = row % NT; }
        abase[rt][1] = q * 1024 + (ok ? rec_of(z, tile) : NZ);
        abase[rt][0] = q * 1024 + ((ok && z > 0) ? rec_of(z - 1, tile) : NZ);
        abase[rt][2] = q * 1024 + ((ok && z < D - 1) ? rec_of(z + 1, tile) : NZ);
    }

    f32x4 acc[2][16];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int p = 0; p < 16; ++p) acc[rt][p] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- pipeline state (wave-uniform) ----------------------------------------------------------------------------
    // stage s = (unit s / nchunks, chunk s % nchunks); "L" = the stage whose global loads are issued next, "R" = the stage
    // whose raw slice is written / transformed next
    const float* in0 = a.in + a.in_coff;
    auto slice_ptr = [&](int kunit, int chunk) -> const float* {
        int64_t f; int cb; bool ok;
        unit_of(kunit < my_units ? kunit : my_units - 1, f, cb, ok);
        if (DBG & 256) f = 0;                           // (timing: every workgroup reads frame 0 — the slices come from L2)
        return in0 + f * a.in_fs + (a.in_blk ? NV * 4 : 4) * chunk;
    };
    int kL = 0, cL = 0, cR = 0;
    float4 pre[2];
    const float* pslice = slice_ptr(0, 0);              // the slice whose loads are issued next
    auto advance_slice = [&]() {
        if (++cL == a.nchunks) { cL = 0; ++kL; }
        pslice = slice_ptr(kL, cL);
    };
    auto issue_loads = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) pre[j] = *reinterpret_cast<const float4*>(pslice + goff[j]);
        advance_slice();
    };
    // PRE == 1: scale / shift of the 4 channels the prologue handles next, requested a stage ahead like everything else
    float4 psc = make_float4(1.f, 1.f, 1.f, 1.f), psh = make_float4(0.f, 0.f, 0.f, 0.f);
    auto load_pre = [&]() {
        if (PRE == 1) {
            psc = *reinterpret_cast<const float4*>(a.pre.scale + 4 * cR);
            psh = *reinterpret_cast<const float4*>(a.pre.shift + 4 * cR);
        }
    };
    // BN -> activation prologue on voxel j's four channels (x[4 j .. 4 j + 3]) of chunk cR
    auto prologue4 = [&](float (&x)[8], int j) {
        if (PRE == 1) {
            x[4 * j + 0] = fmaxf(fmaf(x[4 * j + 0], psc.x, psh.x), 0.f);
            x[4 * j + 1] = fmaxf(fmaf(x[4 * j + 1], psc.y, psh.y), 0.f);
            x[4 * j + 2] = fmaxf(fmaf(x[4 * j + 2], psc.z, psh.z), 0.f);
            x[4 * j + 3] = fmaxf(fmaf(x[4 * j + 3], psc.w, psh.w), 0.f);
        } else if (PRE == 2) {
            float y[4] = {x[4 * j], x[4 * j + 1], x[4 * j + 2], x[4 * j + 3]};
            if (a.pre.scale) {
#pragma unroll
                for (int k = 0; k < 4; ++k) y[k] = fmaf(y[k], a.pre.scale[4 * cR + k], a.pre.shift[4 * cR + k]);
            }
            th_act_vec<4>(y, a.pre.act, a.pre.alpha);
#pragma unroll
            for (int k = 0; k < 4; ++k) x[4 * j + k] = y[k];
        }
    };
    auto store_R = [&](const float (&x)[8], int j) {
#pragma unroll
        for (int k = 0; k < 4; ++k) R1[k * RPL + rdst[j]] = x[4 * j + k];
    };
    auto next_pre = [&]() {
        if (++cR == a.nchunks) cR = 0;
        load_pre();
    };
    auto write_R = [&]() {
        float x[8] = {pre[0].x, pre[0].y, pre[0].z, pre[0].w, pre[1].x, pre[1].y, pre[1].z, pre[1].w};
        prologue4(x, 0);
        prologue4(x, 1);
        store_R(x, 0);
        store_R(x, 1);
        next_pre();
    };
    // V = BT d BT^T of this thread's patch, channel 2 th + k, into Vn.  The arithmetic runs on x PAIRS (the two floats of a
    // ds_read_b64; v_pk_add_f32 with sign / half selects), so that the four b values of a row come out in consecutive registers —
    // what ds_write_b128 wants: 8 ds_read_b64, ~16 packed adds, 4 ds_write_b128 per channel and nothing to shuffle.
    auto read_patch = [&](v2f (&d)[4][2], int k) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float* r = R1 + k * RPL + roff[i];
            if (DBG & 128) { d[i][0] = d[i][1] = (v2f){__builtin_bit_cast(float, roff[i] + k), 1.f}; continue; }
            d[i][0] = *reinterpret_cast<const v2f*>(r);
            d[i][1] = *reinterpret_cast<const v2f*>(r + 2);
        }
    };
    auto cols = [&](const v2f (&d)[4][2], v2f (&t)[4][2]) {
#pragma unroll
        for (int hx = 0; hx < 2; ++hx) {
            t[0][hx] = d[0][hx] - d[2][hx];
            t[1][hx] = d[1][hx] + d[2][hx];
            t[2][hx] = d[2][hx] - d[1][hx];
            t[3][hx] = d[1][hx] - d[3][hx];
        }
    };
    auto rows = [&](const v2f (&t)[4][2], float4* Vn, int k, int a0, int a1) {
#pragma unroll
        for (int aa = a0; aa < a1; ++aa) {
            const v2f t01 = t[aa][0], t23 = t[aa][1];
            v2f v01, v23;
            // (t0 - t2, t1 + t2) and (t2 - t1, t1 - t3) as ONE packed add each: half selects and sign bits are free operand
            // modifiers (written out: hipcc builds these from scalar adds, sign flips and moves)
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(v01) : "v"(t01), "v"(t23));
            asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(v23) : "v"(t23), "v"(t01));
            const float4 v = make_float4(v01.x, v01.y, v23.x, v23.y);
            if (DBG & 64) { if (v.x + v.y + v.z + v.w == 1.2345e-30f) Vn[wdst + (k * 4 + aa) * 256] = v; }
            else Vn[wdst + (k * 4 + aa) * 256] = v;
        }
    };
    auto transform_all = [&](float4* Vn) {
        v2f d[4][2], t[4][2];
#pragma unroll
        for (int k = 0; k < 2; ++k) { read_patch(d, k); cols(d, t); rows(t, Vn, k, 0, 4); }
    };
    // B fragments: the 768 float4 of a stage ([a][dz][lane]) live in LDS, read by every wave.  They arrive in two halves
    // (a-steps 0,1 / 2,3) through one register each of threads 0..383: loaded from L2 a stage ahead — BEFORE the slice loads of
    // the same stage top, so that waiting for them (vmcnt counts in order) never waits for the slice, which has a whole stage —
    // and written to LDS right after the barrier behind which nobody reads the half any more.
    float4 bn0, bn1;
    const int bsrc = tid < 384 ? tid : 0, bdst = tid < 384 ? tid : 768;
    const float4* const wq = reinterpret_cast<const float4*>(a.wpk) + bsrc;
    auto wbase = [&](int kunit, int chunk) -> unsigned {                 // float4 index of (unit's cb, chunk, a = 0, dz = 0)
        int64_t f; int cb; bool ok;
        unit_of(kunit < my_units ? kunit : my_units - 1, f, cb, ok);
        return (unsigned)((cb * a.nchunks + chunk) * 4) * 192u;
    };

    // ---- prologue of the pipeline: slice 0 -> R -> V[0]; slice 1 in flight ----------------------------------------
    load_pre();
    issue_loads();
    __syncthreads();                                    // LDS zeroed
    write_R();
    issue_loads();
    __syncthreads();
    transform_all(V4);
    {
        const unsigned w0 = wbase(0, 0);
        B4[bdst] = wq[w0];
        bn1 = wq[w0 + 384];
    }
    unsigned wnext = a.nchunks > 1 ? wbase(0, 1) : wbase(1, 0);     // the weights of the stage after the current one

    int ku = 0, c = 0;                                  // current stage's unit ordinal and chunk
    for (int s = 0; s < S; ++s) {
        if (!(DBG & 8)) __syncthreads();                // (A) V[s & 1] complete; nobody reads V[(s + 1) & 1] or R any more
        const float4* const Vc = V4 + (s & 1) * kVB;
        float4* const Vn = V4 + ((s + 1) & 1) * kVB;
        // The stage as 12 MFMA steps n = 3 g + dz (8 MFMAs each: positions (g, 0..3) of both row tiles for z tap dz) with
        // everything else placed in the slots between them; the operands of step n + 1 are requested before the MFMAs of
        // step n.  The slots are pinned (sched_barrier): all eight waves leave a barrier together, so whatever a wave does
        // between two MFMAs nobody else on its SIMD covers.
        float4 ob[2], oa0[2], oa1[2];
#define WF_FETCH_A(n) { oa0[(n) & 1] = Vc[abase[0][(n) % 3] + ((n) / 3) * 256]; oa1[(n) & 1] = Vc[abase[1][(n) % 3] + ((n) / 3) * 256]; }
#define WF_FETCH_B(n) { ob[(n) & 1] = B4[(n) * 64 + lane]; }
#define WF_MMA4(rt, A, n)                                                                                                                \
        acc[rt][4 * ((n) / 3) + 0] = __builtin_amdgcn_mfma_f32_16x16x4f32(A.x, ob[(n) & 1].x, acc[rt][4 * ((n) / 3) + 0], 0, 0, 0);     \
        acc[rt][4 * ((n) / 3) + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(A.y, ob[(n) & 1].y, acc[rt][4 * ((n) / 3) + 1], 0, 0, 0);     \
        acc[rt][4 * ((n) / 3) + 2] = __builtin_amdgcn_mfma_f32_16x16x4f32(A.z, ob[(n) & 1].z, acc[rt][4 * ((n) / 3) + 2], 0, 0, 0);     \
        acc[rt][4 * ((n) / 3) + 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(A.w, ob[(n) & 1].w, acc[rt][4 * ((n) / 3) + 3], 0, 0, 0);
#define WF_MMA(n) { WF_MMA4(0, oa0[(n) & 1], n) WF_MMA4(1, oa1[(n) & 1], n) }
// end of a slot: inside it the scheduler is asked for "one MFMA, then up to 5 other instructions (VALU | SALU | VMEM | DS)", eight
// times — a wave's own VALU / LDS work issues in the shadow of its own MFMAs instead of after them (all waves leave a barrier
// together: what one wave does after its MFMAs, the other wave of the SIMD does at the same time, and the matrix pipe idles)
#define WF_PIPE1 __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x096, 5, 0);
#define WF_SLOT { if (!(DBG & 32)) { WF_PIPE1 WF_PIPE1 WF_PIPE1 WF_PIPE1 WF_PIPE1 WF_PIPE1 WF_PIPE1 WF_PIPE1 } __builtin_amdgcn_sched_barrier(0); }
        if (!(DBG & 4)) B4[tid < 384 ? 384 + tid : 768] = bn1;          // a-steps 2,3 of this stage
        float x[8] = {pre[0].x, pre[0].y, pre[0].z, pre[0].w, pre[1].x, pre[1].y, pre[1].z, pre[1].w};   // slice of stage s + 1
        if (!(DBG & 4)) {
            bn0 = wq[wnext];                            // stage s + 1
            bn1 = wq[wnext + 384];
        }
        if (!(DBG & 2)) {                               // slice of stage s + 2
#pragma unroll
            for (int j = 0; j < 2; ++j) pre[j] = *reinterpret_cast<const float4*>(pslice + goff[j]);
        }
        // (tried: waves 0..3 and 4..7 — they share the SIMDs pairwise — on DIFFERENT slot schedules behind a wave-uniform branch, so that
        // one wave of a SIMD multiplies while the other stages: the branches cost 100+ spilled registers, 5.9 ms instead of 1.2)
        auto half1 = [&]() __attribute__((always_inline)) {
            constexpr int L = 0;                                    // first slot of the prologue work
            WF_FETCH_A(0) WF_FETCH_B(0)
            WF_SLOT
#define WF_P1(k)                                                                           \
            if (!(DBG & 2)) {                                                              \
                if ((k) == L) prologue4(x, 0);                                             \
                if ((k) == L + 1) { prologue4(x, 1); store_R(x, 0); }                      \
                if ((k) == L + 2) { store_R(x, 1); next_pre(); }                           \
            }
            WF_FETCH_A(1) WF_FETCH_B(1) WF_MMA(0) WF_P1(0)
            WF_SLOT
            WF_FETCH_A(2) WF_FETCH_B(2) WF_MMA(1) WF_P1(1)
            WF_SLOT
            WF_FETCH_A(3) WF_FETCH_B(3) WF_MMA(2) WF_P1(2)
            {   // where the next stage's weights and the slice after next live (scalar arithmetic only)
                const int cn = c + 1 == a.nchunks ? 0 : c + 1, kn = c + 1 == a.nchunks ? ku + 1 : ku;
                const int cn2 = cn + 1 == a.nchunks ? 0 : cn + 1, kn2 = cn + 1 == a.nchunks ? kn + 1 : kn;
                wnext = wbase(kn2, cn2);
                advance_slice();
            }
            WF_SLOT
            WF_FETCH_A(4) WF_FETCH_B(4) WF_MMA(3) WF_P1(3)
            WF_SLOT
            WF_FETCH_A(5) WF_FETCH_B(5) WF_MMA(4) WF_P1(4)
            WF_SLOT
            WF_FETCH_A(6) WF_MMA(5) WF_P1(5)
            WF_SLOT
#undef WF_P1
        };
        // the transform of slice s + 1 (two channels per thread) in six pieces
        auto half2 = [&]() __attribute__((always_inline)) {
            constexpr int L = 0;                                    // first slot of the transform
            v2f d[4][2], t[4][2];
#define WF_P2(k)                                                                           \
            if (!(DBG & 1)) {                                                              \
                if ((k) == L) read_patch(d, 0);                                            \
                if ((k) == L + 1) { cols(d, t); read_patch(d, 1); }                        \
                if ((k) == L + 2) { rows(t, Vn, 0, 0, 4); cols(d, t); }                    \
                if ((k) == L + 3) { rows(t, Vn, 1, 0, 2); }                                \
                if ((k) == L + 4) { rows(t, Vn, 1, 2, 4); }                                \
            }
            WF_FETCH_B(6) WF_P2(0)
            WF_SLOT
            WF_FETCH_A(7) WF_FETCH_B(7) WF_MMA(6) WF_P2(1)
            WF_SLOT
            WF_FETCH_A(8) WF_FETCH_B(8) WF_MMA(7) WF_P2(2)
            WF_SLOT
            WF_FETCH_A(9) WF_FETCH_B(9) WF_MMA(8) WF_P2(3)
            WF_SLOT
            WF_FETCH_A(10) WF_FETCH_B(10) WF_MMA(9) WF_P2(4)
            WF_SLOT
            WF_FETCH_A(11) WF_FETCH_B(11) WF_MMA(10) WF_P2(5)
            WF_SLOT
            WF_MMA(11)
#undef WF_P2
        };
        half1();
        if (!(DBG & 8)) __syncthreads();                // (B) R complete; nobody reads the first half of B any more
        if (!(DBG & 4)) B4[bdst] = bn0;                 // a-steps 0,1 of stage s + 1
        half2();
#undef WF_FETCH_A
#undef WF_FETCH_B
#undef WF_MMA4
#undef WF_MMA
#undef WF_SLOT
#undef WF_PIPE1

        if (c + 1 == a.nchunks) {
            // ---- unit done: inverse transform, bias, epilogue chain, (pool,) store; C layout col = i16, row = 4 q + r ----
            int64_t f; int cb; bool uok;
            unit_of(ku, f, cb, uok);
            wf_epilogue<D, H, W, POOL>(a, acc, f, cb, uok, wave, i16, q);
            c = 0; ++ku;
        } else {
            ++c;
        }
    }
}


typedef void (*WfKernel)(const ConvWfArgs);
struct WfGeo { int D, H, W; WfKernel k[3][3]; };     // [pool][pre]
#define WF_INST(D, H, W) {D, H, W, {{k_conv_wf<D, H, W, 0, 0>, k_conv_wf<D, H, W, 0, 1>, k_conv_wf<D, H, W, 0, 2>},  \
                                    {k_conv_wf<D, H, W, 1, 0>, k_conv_wf<D, H, W, 1, 1>, k_conv_wf<D, H, W, 1, 2>},  \
                                    {k_conv_wf<D, H, W, 2, 0>, k_conv_wf<D, H, W, 2, 1>, k_conv_wf<D, H, W, 2, 2>}}}
const WfGeo kWfGeo[] = {WF_INST(10, 10, 10)};
#undef WF_INST
struct WfDbg { int code; WfKernel k; };
#define WF_DBG(c) {c, k_conv_wf<10, 10, 10, 0, 0, c>}
const WfDbg kWfDbg[] = {WF_DBG(1), WF_DBG(2), WF_DBG(3), WF_DBG(4), WF_DBG(7), WF_DBG(8), WF_DBG(15), WF_DBG(32), WF_DBG(64), WF_DBG(256)};
#undef WF_DBG

}  // namespace

// ---- host side -------------------------------------------------------------------------------------------------------------
bool conv_wf_plan(const TView& in, const TView& out, const ConvGeom& g, int Cin, int Cout, int pool, ConvWfPlan* p) {
    if (g.kd != 3 || g.kh != 3 || g.kw != 3 || g.sd != 1 || g.sh != 1 || g.sw != 1 || g.dd != 1 || g.dh != 1 || g.dw != 1) return false;
    if (g.pz != 1 || g.py != 1 || g.px != 1) return false;                              // 'same'
    if (in.D != out.D || in.H != out.H || in.W != out.W) return false;
    if (Cin < 16 || Cin % 4 != 0 || Cout < 1) return false;
    if (pool < 0 || pool > 2) return false;
    int geo = -1;
    for (size_t k = 0; k < sizeof kWfGeo / sizeof kWfGeo[0]; ++k)
        if (kWfGeo[k].D == in.D && kWfGeo[k].H == in.H && kWfGeo[k].W == in.W) geo = (int)k;
    if (geo < 0) return false;
    const int NT = (in.H / 2) * (in.W / 2);
    p->geo = geo; p->pool = pool;
    p->knobs = &th_knobs_planning();
    p->Cin = Cin; p->Cout = Cout;
    p->ncb = (Cout + 15) / 16;
    p->nchunks = Cin / 4;
    p->wpk_floats = (size_t)p->ncb * p->nchunks * 3072;
    // the algorithm's own multiply-adds: 16 positions x 3 z taps per 2 x 2 tile (the z taps of the two border planes that
    // fall outside the volume are not counted; the kernel executes them on a zero record), and what the MFMAs issue
    p->own_flops = 2.0 * 16 * NT * (3.0 * in.D - 2) * Cin * (double)Cout;
    p->exec_flops = 2.0 * 16 * 256 * 3 * Cin * 16.0 * p->ncb;
    p->lds_bytes = (size_t)2 * 65536 + 769 * 16 + (size_t)4 * ((in.D * in.H + 1) * (in.W + 2) + 4) * 4 + 16;
    if (p->lds_bytes > kWfLdsLimit) return false;
    char buf[200];
    snprintf(buf, sizeof buf, "conv_wf<F(2,3)^2 in-plane fused in LDS, z direct; pool%d> 16c x %d, K%d, lds%zuK (16x16x4 MFMA) [k_conv_wf<%d,%d,%d,%d>]",
             pool, p->ncb, Cin, p->lds_bytes / 1024, in.D, in.H, in.W, pool);
    p->label = buf;
    return true;
}

// Keras [3][3][3][Cin][Cout] -> U[a][b][dz][ci][co] = sum_jk G[a][j] G[b][k] W[dz][j][k][ci][co] (double), G of F(2,3),
// laid out [cb][chunk][a][dz][lane = 16 (ci & 3) + (co & 15)][b]; columns past Cout are zero
void conv_wf_pack_weights(const ConvWfPlan& p, const float* w, float* dst) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int Cin = p.Cin, Cout = p.Cout;
    std::memset(dst, 0, p.wpk_floats * sizeof(float));
    for (int ci = 0; ci < Cin; ++ci)
        for (int co = 0; co < Cout; ++co) {
            const int cb = co / 16, chunk = ci / 4, ln = 16 * (ci & 3) + (co & 15);
            for (int dz = 0; dz < 3; ++dz) {
                double wk[3][3];
                for (int j = 0; j < 3; ++j)
                    for (int k = 0; k < 3; ++k) wk[j][k] = (double)w[((((size_t)dz * 3 + j) * 3 + k) * Cin + ci) * Cout + co];
                for (int aa = 0; aa < 4; ++aa) {
                    double t[3];
                    for (int k = 0; k < 3; ++k) t[k] = G[aa][0] * wk[0][k] + G[aa][1] * wk[1][k] + G[aa][2] * wk[2][k];
                    for (int bb = 0; bb < 4; ++bb) {
                        const double u = G[bb][0] * t[0] + G[bb][1] * t[1] + G[bb][2] * t[2];
                        dst[(((((size_t)cb * p.nchunks + chunk) * 4 + aa) * 3 + dz) * 64 + ln) * 4 + bb] = (float)u;
                    }
                }
            }
        }
}

bool conv_wf_view_ok(const TView& in) { return in.cs % 4 == 0 && in.coff % 4 == 0; }

// which prologue instantiation a layer runs on: 0 none, 1 BN-affine -> ReLU from 16-byte aligned vectors, 2 generic
int conv_wf_pre_kind(const PreOp& pre) {
    const bool relu_affine = pre.scale && pre.act == ACT_RELU && ((uintptr_t)pre.scale % 16) == 0 && ((uintptr_t)pre.shift % 16) == 0;
    return (!pre.scale && pre.act == ACT_LINEAR) ? 0 : relu_affine ? 1 : 2;
}
std::string conv_wf_label(const ConvWfPlan& p, const PreOp& pre) {
    const size_t k = p.label.rfind(">]");
    return k == std::string::npos ? p.label : p.label.substr(0, k) + "," + std::to_string(conv_wf_pre_kind(pre)) + ",0>]";
}

int launch_conv_wf(hipStream_t s, int64_t n, const ConvWfPlan& p, TView in, TView out, const float* wpk, const float* bias, PreOp pre,
                   PostOps post) {
    if (n <= 0) return TH_OK;
    if (p.geo < 0 || p.geo >= (int)(sizeof kWfGeo / sizeof kWfGeo[0])) TH_FAIL(TH_EINVAL, "conv_wf: bad plan");
    if (in.cs % 4 || in.coff % 4 || in.fs % 4 || ((uintptr_t)in.p % 16)) TH_FAIL(TH_EINVAL, "conv_wf: the input view is not 16-byte aligned");
    ConvWfArgs a;
    std::memset(&a, 0, sizeof a);
    a.in = in.p; a.in_fs = in.fs; a.in_cs = in.cs; a.in_coff = in.coff;
    a.in_blk = in.blk;
    if (in.blk && (in.blk != 4 || in.coff || in.cs != p.Cin)) TH_FAIL(TH_EINVAL, "conv_wf: bad chunk-blocked input view");
    a.Cin = p.Cin; a.nchunks = p.nchunks; a.wpk = wpk; a.Cout = p.Cout; a.ncb = p.ncb; a.bias = bias; a.pre = pre; a.post = post;
    a.out = out.p; a.out_fs = out.fs; a.out_cs = out.cs; a.out_coff = out.coff; a.nframes = n;
    const int64_t nslots = (n + 7) / 8 * 8 * p.ncb;
    if (nslots > 0x7fffffffLL) TH_FAIL(TH_EINVAL, "conv_wf: too many frames per launch");
    a.nslots = (unsigned)nslots;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    int64_t resident = ncu;                              // one 8-wave workgroup per CU (154 KB of LDS)
    const ThKnobs& kn = th_knobs_of(p.knobs);
    if (kn.wf_resident) resident = std::max(1, kn.wf_resident);       // tests: force multi-trip workgroups
    const int64_t trips = (nslots + resident - 1) / resident;
    int64_t grid = (nslots + trips - 1) / trips;
    grid = (grid + 7) / 8 * 8;
    const int pre_kind = conv_wf_pre_kind(pre);
    WfKernel k = kWfGeo[p.geo].k[p.pool][pre_kind];
    const size_t lds = p.lds_bytes;
    {   // timing experiments only (tools/bench_layer.py): knock-out instantiations of the plain 10^3 kernel
        const int dbg = kn.wf_dbg;
        if (dbg > 0 && p.geo == 0 && p.pool == 0 && pre_kind == 0)
            for (const WfDbg& d : kWfDbg) if (d.code == dbg) k = d.k;
    }
    HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWfLdsLimit));
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(512), lds, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) TH_FAIL(TH_EHIP, "conv_wf launch failed: %s (%s)", hipGetErrorString(e), p.label.c_str());
    return TH_OK;
}
