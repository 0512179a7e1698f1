// First layer on the bf16 matrix pipe with exactly split operands — the aposteriori case only: 21^3 x (5..6)-channel frames,
// Conv3D 3x3x3 'same' to <= 32 filters, 2^3 max-pool taken before a monotone epilogue chain (TIMED, the rotamer model,
// DenseCPD and ProDCoNN all open this way: SURVEY.md §8(a) P2a, Appendix A; the call served is reference predict.py:142).
// Every other first layer stays on k_conv_first_w / k_conv_first (conv_first.hip).
//
// Same algorithm as k_conv_first_w — F(2,3) along x, a GEMM row is an x PAIR, four transform points V0 = d0 - d2, V1 = d1 + d2,
// V2 = d2 - d1, V3 = d1 - d3 per (dz, dy) tap and channel, out(2 tx) = M0 + M1 + M2, out(2 tx + 1) = M1 - M2 - M3 — but the
// products run as v_mfma_f32_32x32x16_bf16 on operands split exactly into three bf16 pieces (x = h + m + l, round to nearest
// even at every step; six of the nine piece products, fp32 accumulation: the scheme of conv_wino.hip's k_wino_gemm_b3, error of
// one fp32 rounding).  A lane half holds the three channels 2 s + h: taps 0..7 fill 3 k-steps of 16 without a padding slot (72
// MFMAs of 32 cycles per tile of 32 pairs), tap 8 runs as 3 v_mfma_f32_32x32x2_f32 per point (table below): 3072 matrix-pipe
// cycles per tile where the fp32 form needs 6912 (108 MFMAs of 64).
//
// What changes around it:
//   * the data points are split IN REGISTERS right after the transform (9 VALU instructions per pair of values: three
//     v_cvt_pk_bf16_f32, four shifts / masks, two subtractions) — 36 + 4 per (k-step, point) unit and lane against its 6 MFMAs.
//     THE VALU IS WHAT BOUNDS THIS KERNEL (~855 instructions per tile and lane at ~4.1 cycles with two waves per SIMD, every
//     MFMA issued costs ~10 cycles of the same port: tools/microbench/mfma_valu_coissue.hip); every unit is
//     a window in which the wave multiplies one unit and prepares the next;
//   * the split weights (4 points x 3 k-steps x 3 pieces x 1 KB = 36 KB) do not fit the register file: they live in LDS, read
//     as one ds_read_b128 per fragment (3 per unit);
//   * that leaves room for ONE workgroup per CU, so the bricks of k_conv_first_w (whose ragged last round and staging the
//     second workgroup used to cover) are replaced by a persistent workgroup of 8 waves that streams whole frames through a RING
//     of 8 input planes (22 x 22 voxels x 6 floats, zero halo included; bank-spread layout below): a frame is 125 tiles = 16
//     rounds of 8 (the last one of 5), round r needs planes 2 pz - 1 .. 2 pz + 2 of at most two pooled planes pz (<= 6 planes),
//     the next two planes are requested from HBM when the round starts and written to their ring slots when it ends — one
//     barrier per round, no halo plane is staged twice, frame boundaries included (the next frame's first planes arrive during
//     the last two rounds).
#include "common.h"
#include "device_math.h"

#include <hip/hip_fp16.h>

#include <algorithm>
#include <cstring>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int kFD = 21;                       // frame extent (aposteriori frames)
constexpr int kPW = 22;                       // staged plane: 22 x 22 voxels (x_in, y_in = -1 .. 20)
constexpr int kPlaneVox = kPW * kPW;          // 484
// A staged plane is TWO images, laid out so that the 32 lanes of a half wave (8 consecutive pooled voxels x their 4 (dz, dy)
// mates, one lane half = one channel parity) spread over all 32 LDS banks:
//   pairs   [row 22][x 23][c0 c2 | c1 c3]   4 words per voxel: a lane half reads (s0, s1) as one ds_read_b64; rows of 23 voxels
//           (92 words = 28 mod 32) and 2 words behind the plane (2026 = 10 mod 32): the lanes' word addresses are
//           8 px + 28 my + 10 mz (+ 2 h): every even residue twice per half wave = the two cycles a b64 half wave needs anyway
//   singles [row 22][x 23][c4 | c5]         2 words per voxel (s2): 4 px + 14 my + 21 mz — all 32 residues different
// (With one 24-byte record per voxel — [c0 c2 | c1 c3 | c4 | c5] — the same reads hit 16 banks: SQ_LDS_BANK_CONFLICT was 61 %
// of the LDS-active cycles and the LDS was busy 49 % of the kernel.)
constexpr int kRowVox = 23;
constexpr int kPairWords = kPW * kRowVox * 4 + 2;     // 2026
constexpr int kSingleWords = kPW * kRowVox * 2 + 1;   // 1013
constexpr int kPlaneFloats = kPairWords + kSingleWords;
constexpr int kRing = 8;
constexpr int kSingleBase = kRing * kPairWords;       // the singles images start behind the 8 pair images
constexpr int kPlanesPerFrame = 22;           // c = 0: the zero plane z_in = -1; c = 1 .. 21: z_in = c - 1
constexpr int kTiles = 125;                   // 1000 pooled voxels x 4 (dz, dy) mates / 32 rows
constexpr int kRounds = 16;
constexpr int kB3WFrag = 4 * 3 * 3 * 64;      // uint4 fragments of the split weights: [point][k-step][piece][lane]
constexpr int kB3W8 = 4 * 3 * 64;             // floats behind them: tap 8 for the fp32 form, [point][step][lane]
constexpr size_t kB3Lds = (size_t)kRing * kPlaneFloats * 4 + (size_t)kB3WFrag * 16;

struct ConvFirstB3Args {
    const void* in; int dtype; int Cin; int vec8;
    const uint4* wpk;                 // [point 4][k-step 3][piece 3][lane 64] x 8 bf16, then tap 8 as fp32 [point 4][step 3][lane 64]
    int Cout;
    const float* bias;
    PostOps post;
    float* out; int64_t out_fs; int out_cs, out_coff, Ho, Wo;
    int out_blk_stride;
    int64_t nframes;
};

__device__ __forceinline__ float b3_load_elem(const void* base, int dtype, int64_t i) {
    switch (dtype) {
        case TH_F32: return ((const float*)base)[i];
        case TH_F64: return (float)((const double*)base)[i];
        case TH_U8: return (float)((const unsigned char*)base)[i];
        case TH_BOOL: return ((const unsigned char*)base)[i] ? 1.f : 0.f;
        default: return __half2float(((const __half*)base)[i]);
    }
}

__device__ __forceinline__ unsigned b3_pk(v2f x) {          // v_cvt_pk_bf16_f32: round to nearest even
    return __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2));
}
__device__ __forceinline__ v2f b3_unpk(unsigned w) {
    return (v2f){__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xffff0000u)};
}

// which (tap t9 = 3 dz + dy, channel slot s) sits in k-slot pair q of k-step ks.  Pairs 'P' are the two floats of a voxel's
// ds_read_b64 (s = 0, 1), pairs 'S' join the s = 2 singles of two taps:
//   ks 0: P(t0) P(t1) P(t2) S(t0, t1)      ks 1: P(t3) P(t4) S(t3, t4) S(t2, t5)      ks 2: P(t5) P(t6) P(t7) S(t6, t7)
// 8 taps x 3 channels = 24 slots per lane half = 3 k-steps without a padding slot; tap 8 (3 more values per half) would cost a
// fourth, mostly empty k-step (24 MFMAs, 768 cycles) — it runs as 3 v_mfma_f32_32x32x2_f32 per point instead: the same 768
// cycles, but no split arithmetic and its 12 weights per lane stay in registers.
// (host side: kB3Slot below is the same table per k-slot)
struct B3Slot { signed char t, s; };
constexpr B3Slot kB3Slot[3][8] = {
    {{0, 0}, {0, 1}, {1, 0}, {1, 1}, {2, 0}, {2, 1}, {0, 2}, {1, 2}},
    {{3, 0}, {3, 1}, {4, 0}, {4, 1}, {3, 2}, {4, 2}, {2, 2}, {5, 2}},
    {{5, 0}, {5, 1}, {6, 0}, {6, 1}, {7, 0}, {7, 1}, {6, 2}, {7, 2}},
};

// DBG (TH_FIRST_DBG with the split kernel, results WRONG unless noted): 1 no split arithmetic (the pieces are raw bit patterns),
// 2 no MFMAs, 4 no staging (the ring keeps the first planes), 8 no epilogue chain / stores, 64 (results right) the compiler's own
// instruction order inside the windows, 128 no weight-fragment reads, 256 no raw-voxel reads
// INT: the frames hold integers 0 .. 255 (uint8 / bool datasets: what the reference builds for voxels_as_gaussian=False,
// design_utils/utils.py:518-521).  Every voxel and the three difference points d0 - d2, d2 - d1, d1 - d3 are then exact in ONE bf16
// piece (|x| <= 255), the sum point d1 + d2 (<= 510) in two: 3 products per multiply-add instead of 6 (5 at the sum point) and no
// split arithmetic except one subtraction at the sum point.  The products that remain run in the order the general form runs
// them, and the ones dropped are exact zeros there: the results are bit-identical to the fp32-frame run of the same values.
template <int DBG, int INT = 0>
__global__ void __launch_bounds__(512, 1) k_conv_first_b3(const ConvFirstB3Args a) {
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    float* const A = reinterpret_cast<float*>(smem);
    uint4* const B4 = reinterpret_cast<uint4*>(A + kRing * kPlaneFloats);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int64_t G = gridDim.x;
    const int nk = (int)((a.nframes - blockIdx.x + G - 1) / G);          // frames of this workgroup: blockIdx.x + k G
    if (nk <= 0) return;
    const int total = nk * kPlanesPerFrame;

    // ---- staging: planes [S, S2) of the workgroup's plane sequence (22 per frame), two voxels per thread -------------------
    const int64_t frame_elems = (int64_t)kFD * kFD * kFD * a.Cin;
    const bool fast6 = a.vec8 != 0;
    const bool fastu8 = INT && a.vec8 == 0 && a.Cin == 6 && (a.dtype == TH_U8 || a.dtype == TH_BOOL) && ((uintptr_t)a.in % 2) == 0;
    float e[2][6];
    int sdst[2];
    // per-thread constants of its two voxels v = tid, tid + 512 of a PAIR of planes (2 x 484): which plane of the pair, the
    // record's place inside the plane, the element offset inside a frame's z plane (< 0: halo, stays zero)
    int spl[2], sloc[2], sgo[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int v = tid + 512 * u;
        spl[u] = v >= kPlaneVox ? 1 : 0;
        const int vox = v - spl[u] * kPlaneVox;
        const int y = vox / kPW, x = vox - y * kPW;
        const int yi = y - 1, xi = x - 1;
        sloc[u] = v < 2 * kPlaneVox ? y * kRowVox + x : -1;          // voxel index inside the padded plane
        sgo[u] = (v < 2 * kPlaneVox && yi >= 0 && yi < kFD && xi >= 0 && xi < kFD) ? (yi * kFD + xi) * a.Cin : -1;
    }
    const int plane_elems = kFD * kFD * a.Cin;
    auto issue = [&](int S, int S2) {
        // (wave-uniform: frame and z plane of the pair's two planes)
        int64_t pbase[2];
        int pslot[2];
        bool preal[2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const int gp = S + pl;
            const int k = gp / kPlanesPerFrame, c = gp - k * kPlanesPerFrame;
            preal[pl] = gp < S2 && c >= 1 && !(DBG & 4);
            pslot[pl] = gp < S2 ? (gp & (kRing - 1)) : -1;
            pbase[pl] = (blockIdx.x + (int64_t)k * G) * frame_elems + (int64_t)(c - 1) * plane_elems;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int c = 0; c < 6; ++c) e[u][c] = 0.f;
            const int slot = spl[u] ? pslot[1] : pslot[0];
            sdst[u] = (slot >= 0 && sloc[u] >= 0) ? slot * (1 << 16) + sloc[u] : -1;      // (ring slot, voxel)
            if ((spl[u] ? preal[1] : preal[0]) && sgo[u] >= 0) {
                const int64_t base = (spl[u] ? pbase[1] : pbase[0]) + sgo[u];
                if (fast6) {
                    const float2* p2 = reinterpret_cast<const float2*>((const float*)a.in + base);
                    const float2 u0 = p2[0], u1 = p2[1], u2 = p2[2];
                    e[u][0] = u0.x; e[u][1] = u0.y; e[u][2] = u1.x; e[u][3] = u1.y; e[u][4] = u2.x; e[u][5] = u2.y;
                } else if (fastu8) {          // six bytes of a voxel as three 16-bit loads (a voxel starts at an even byte)
                    const unsigned short* p16 = reinterpret_cast<const unsigned short*>((const unsigned char*)a.in + base);
                    const unsigned w0 = p16[0], w1 = p16[1], w2 = p16[2];
                    const unsigned b[6] = {w0 & 255u, w0 >> 8, w1 & 255u, w1 >> 8, w2 & 255u, w2 >> 8};
#pragma unroll
                    for (int c2 = 0; c2 < 6; ++c2) e[u][c2] = a.dtype == TH_BOOL ? (b[c2] ? 1.f : 0.f) : (float)b[c2];
                } else {
#pragma unroll
                    for (int c2 = 0; c2 < 6; ++c2)
                        if (c2 < a.Cin) e[u][c2] = b3_load_elem(a.in, a.dtype, base + c2);
                }
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (sdst[u] < 0) continue;
            const int slot = sdst[u] >> 16, vox = sdst[u] & 0xffff;
            float* rec = A + slot * kPairWords + vox * 4;
            *reinterpret_cast<float2*>(rec) = make_float2(e[u][0], e[u][2]);
            *reinterpret_cast<float2*>(rec + 2) = make_float2(e[u][1], e[u][3]);
            float* sg = A + kSingleBase + slot * kSingleWords + vox * 2;
            sg[0] = e[u][4];
            sg[1] = e[u][5];
        }
    };

    // ---- prologue: the first four planes, the split weights -------------------------------------------------------------
    int S = 0;
    issue(0, 2); commit();
    issue(2, 4); commit();
    S = 4;
    for (int i = tid; i < kB3WFrag; i += 512) B4[i] = a.wpk[i];
    float w8[4][3];                                                    // tap 8, fp32: [point][step], channel 2 step + h of column j
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int st = 0; st < 3; ++st) w8[p][st] = reinterpret_cast<const float*>(a.wpk + kB3WFrag)[(p * 3 + st) * 64 + lane];
    __syncthreads();

    const int co = j;
    const int cofs = a.out_blk_stride ? (co >> 2) * a.out_blk_stride + (co & 3) : co;
    const bool cok = co < a.Cout;
    const int cc = cok ? co : 0;
    const float bv = a.bias ? a.bias[cc] : 0.f;
    // the two chains TIMED / DenseCPD blocks use get straight-line code (see k_conv_first_w): 1 = ELU -> BN-affine, 2 = BN-affine -> ReLU
    int epi = 0;
    float esc = 1.f, esh = 0.f, ealpha = 1.f;
    if (a.post.n == 2) {
        if (a.post.type[0] == POP_ACT && a.post.act[0] == ACT_ELU && a.post.type[1] == POP_AFFINE) {
            epi = 1; ealpha = a.post.alpha[0]; esc = a.post.scale[1][cc]; esh = a.post.shift[1][cc];
        } else if (a.post.type[0] == POP_AFFINE && a.post.type[1] == POP_ACT && a.post.act[1] == ACT_RELU) {
            epi = 2; esc = a.post.scale[0][cc]; esh = a.post.shift[0][cc];
        }
    }
    epi = __builtin_amdgcn_readfirstlane(epi);

    // a finished tile leaves 4 pooled sums per lane; their chain and stores are deferred into the next tile (as k_conv_first_w)
    float mprev[4] = {0.f, 0.f, 0.f, 0.f};
    int tprev = -1;
    float* oprev = nullptr;
    auto finish_prev = [&]() {
        if (tprev < 0) return;
        if (DBG & 8) { if (mprev[0] + mprev[1] + mprev[2] + mprev[3] == 1.2345e-30f) oprev[0] = mprev[0]; tprev = -1; return; }
        if (epi == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float x = mprev[q];
                mprev[q] = fmaf(x > 0.f ? x : ealpha * (__expf(x) - 1.f), esc, esh);
            }
        } else if (epi == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) mprev[q] = fmaxf(fmaf(mprev[q], esc, esh), 0.f);
        } else {
            th_post2(mprev[0], mprev[1], cc, a.post);
            th_post2(mprev[2], mprev[3], cc, a.post);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pq = 8 * tprev + 2 * q + h;                    // rows 8 q + 4 h .. + 3 of the tile: the mates of this pooled voxel
            const int pz = pq / 100, rem = pq - 100 * pz, py = rem / 10, px = rem - 10 * py;
            if (cok) oprev[(int64_t)((pz * a.Ho + py) * a.Wo + px) * a.out_cs + cofs] = mprev[q];
        }
        tprev = -1;
    };

    for (int k = 0; k < nk; ++k) {
        const int64_t f = blockIdx.x + (int64_t)k * G;
        float* const outb = a.out + f * a.out_fs + a.out_coff;
        for (int r = 0; r < kRounds; ++r) {
            // planes that may be written when this round ends: none of the (<= 6) planes the round reads shares their slots
            const int lowG = kPlanesPerFrame * k + 2 * ((256 * r) / 400);
            const int S2 = min(min(lowG + kRing, S + 2), total);
            issue(S, S2);
            const int t = 8 * r + wave;
            if (t < kTiles) {
                // ---- this lane's row: pooled voxel 8 t + (j >> 2), mate j & 3 = (dz, dy) of the pool window; the pair is its x extent
                const int pq = 8 * t + (j >> 2), mate = j & 3;
                const int pz = pq / 100, rem = pq - 100 * pz, py = rem / 10, px = rem - 10 * py;
                const int z = 2 * pz + (mate >> 1), y = 2 * py + (mate & 1);
                int pa[3], sa[3];                                      // word index of d0 at (dz, dy = 0): this half's pair / single
#pragma unroll
                for (int dz = 0; dz < 3; ++dz) {
                    const int slot = (kPlanesPerFrame * k + z + dz) & (kRing - 1);
                    pa[dz] = slot * kPairWords + (y * kRowVox + 2 * px) * 4 + 2 * h;
                    sa[dz] = kSingleBase + slot * kSingleWords + (y * kRowVox + 2 * px) * 2 + h;
                }
                auto rdP = [&](int t9, v2f (&d)[4]) {
                    const int dz = t9 / 3, dy = t9 % 3;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (DBG & 256) { d[i] = (v2f){__builtin_bit_cast(float, pa[dz] + t9), (float)i}; continue; }
                        d[i] = *reinterpret_cast<const v2f*>(A + pa[dz] + (dy * kRowVox + i) * 4);
                    }
                };
                auto rdS = [&](int t9, float (&d)[4]) {
                    const int dz = t9 / 3, dy = t9 % 3;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (DBG & 256) { d[i] = __builtin_bit_cast(float, pa[dz] + t9 + i); continue; }
                        d[i] = A[sa[dz] + (dy * kRowVox + i) * 2];
                    }
                };
                auto join = [&](const float (&sa)[4], const float (&sb)[4], v2f (&d)[4]) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) d[i] = (v2f){sa[i], sb[i]};
                };
                // the raw voxels of k-step ks as four pairs of k-slots [q][voxel i] (the table in front of the kernel)
                auto load_raw = [&](int ks, v2f (&d)[4][4]) {
                    float sa[4], sb[4];
                    if (ks == 0) {
                        rdP(0, d[0]); rdP(1, d[1]); rdP(2, d[2]);
                        rdS(0, sa); rdS(1, sb); join(sa, sb, d[3]);
                    } else if (ks == 1) {
                        rdP(3, d[0]); rdP(4, d[1]);
                        rdS(3, sa); rdS(4, sb); join(sa, sb, d[2]);
                        rdS(2, sa); rdS(5, sb); join(sa, sb, d[3]);
                    } else {
                        rdP(5, d[0]); rdP(6, d[1]); rdP(7, d[2]);
                        rdS(6, sa); rdS(7, sb); join(sa, sb, d[3]);
                    }
                };
                auto point = [&](int p, const v2f (&d)[4]) -> v2f {
                    return p == 0 ? d[0] - d[2] : p == 1 ? d[1] + d[2] : p == 2 ? d[2] - d[1] : d[1] - d[3];
                };
                f32x16 acc[4];                                         // (zeroed by the first MFMA of each: its C operand is the constant 0)
                if (DBG & 2) {
#pragma unroll
                    for (int p = 0; p < 4; ++p)
#pragma unroll
                        for (int i = 0; i < 16; ++i) acc[p][i] = 0.f;
                }

                // Units: F p = tap 8 of point p on the fp32 pipe form (3 MFMAs of 64 cycles, nothing to split); U k = (k-step k / 4,
                // point k % 4): 6 MFMAs of 32 cycles on acc[point], back to back (a dependent chain costs nothing: measured).  Every
                // unit is a WINDOW of 192 matrix-pipe cycles in which the wave also prepares the NEXT unit — the point's four pairs
                // (1 packed add each), their split (9 VALU each), the three weight fragments (ds_read_b128) — with the instruction
                // order pinned: one MFMA, then what fits in its shadow.  All eight waves leave the round's barrier together, so
                // nothing else covers a wave's VALU phase: left to themselves the phases add up (measured: split 0.48 ms + MFMA
                // 0.66 ms + everything else 0.9 ms = the 2.2 ms the unpipelined kernel took).
                unsigned PA[2][3][4];                                  // [buffer][piece][pair]
                bf16x8 PB[2][3];
                v2f D[2][4][4];
                auto prep = [&](int p, const v2f (&d)[4][4], int buf) {
                    // (two-float subtractions: hipcc emits v_pk_add_f32 for a third of them and two scalar adds for the rest; forcing the
                    // packed form everywhere — inline asm — is SLOWER, 1.91 against 1.79 ms: a packed fp32 add is evidently not cheaper
                    // than two scalar ones here.)
                    // (x - piece through v_dot2c_f32_bf16 — x + piece.lo * (-1) + piece.hi * 0, one instruction per value instead of
                    // shift / mask / subtract — is exact too and 25 % fewer VALU instructions, but measured SLOWER next to the MFMAs
                    // (2.10 against 1.96 ms): the DOT unit is not independent of the matrix pipe.  Two traps on the way, kept here:
                    // hipcc encodes the operand (bf16 -1, 0) as the inline constant -1.0, which the hardware expands differently
                    // (garbage results), and a DOT result needs 3 wait states before a VALU reads it, which inline asm hides from
                    // the hazard recognizer.)
                    v2f x[4], r1[4], r2[4];
                    unsigned hh[4], mm[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) x[q] = point(p, d[q]);
                    if (DBG & 1) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            PA[buf][0][q] = __builtin_bit_cast(unsigned, x[q].x); PA[buf][1][q] = __builtin_bit_cast(unsigned, x[q].y); PA[buf][2][q] = PA[buf][0][q];
                        }
                        return;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) hh[q] = b3_pk(x[q]);
                    if (INT && p != 1) {           // exact in one piece
#pragma unroll
                        for (int q = 0; q < 4; ++q) { PA[buf][0][q] = hh[q]; PA[buf][1][q] = 0u; PA[buf][2][q] = 0u; }
                        return;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) r1[q] = x[q] - b3_unpk(hh[q]);
                    if (INT) {                     // the sum point d1 + d2 <= 510: exact in two
#pragma unroll
                        for (int q = 0; q < 4; ++q) { PA[buf][0][q] = hh[q]; PA[buf][1][q] = b3_pk(r1[q]); PA[buf][2][q] = 0u; }
                        return;
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) mm[q] = b3_pk(r1[q]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) r2[q] = r1[q] - b3_unpk(mm[q]);
#pragma unroll
                    for (int q = 0; q < 4; ++q) { PA[buf][0][q] = hh[q]; PA[buf][1][q] = mm[q]; PA[buf][2][q] = b3_pk(r2[q]); }
                };
                auto loadB = [&](int ks, int p, int buf) {
#pragma unroll
                    for (int piece = 0; piece < 3; ++piece) {
                        if (DBG & 128) { PB[buf][piece] = __builtin_bit_cast(bf16x8, (u32x4){(unsigned)lane, (unsigned)(ks + p), (unsigned)piece, 0u}); continue; }
                        PB[buf][piece] = __builtin_bit_cast(bf16x8, B4[((p * 3 + ks) * 3 + piece) * 64 + lane]);
                    }
                };
                auto mma = [&](int p, int buf) {
                    if (DBG & 2) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[p][q] += __builtin_bit_cast(float, PA[buf][0][q] ^ PA[buf][1][q] ^ PA[buf][2][q]);
                        return;
                    }
                    auto A8 = [&](int piece) { return __builtin_bit_cast(bf16x8, (u32x4){PA[buf][piece][0], PA[buf][piece][1], PA[buf][piece][2], PA[buf][piece][3]}); };
                    if (!INT) acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A8(2), PB[buf][0], acc[p], 0, 0, 0);      // l H
                    if (!INT || p == 1) acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A8(1), PB[buf][0], acc[p], 0, 0, 0);      // m H
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A8(0), PB[buf][0], acc[p], 0, 0, 0);      // h H
                    if (!INT || p == 1) acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A8(1), PB[buf][1], acc[p], 0, 0, 0);      // m M
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A8(0), PB[buf][1], acc[p], 0, 0, 0);      // h M
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A8(0), PB[buf][2], acc[p], 0, 0, 0);      // h L
                };
// a window's instruction order: N times "one MFMA, then up to V VALU and L LDS instructions"
#define B3_PIPE(V, L) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, V, 0); __builtin_amdgcn_sched_group_barrier(0x100, L, 0);
#define B3_WIN6 { if (!(DBG & 64)) { if (INT) { B3_PIPE(10, 6) B3_PIPE(10, 6) B3_PIPE(10, 6) B3_PIPE(10, 6) B3_PIPE(10, 6) } \
                                      else { B3_PIPE(7, 4) B3_PIPE(7, 4) B3_PIPE(7, 4) B3_PIPE(7, 4) B3_PIPE(7, 4) B3_PIPE(7, 4) } } __builtin_amdgcn_sched_barrier(0); }
#define B3_WIN3 { if (!(DBG & 64)) { B3_PIPE(15, 8) B3_PIPE(15, 8) B3_PIPE(15, 8) } __builtin_amdgcn_sched_barrier(0); }
                // ---- prologue: tap 8 (fp32), the raw voxels of k-step 0, the fragments of unit 0
                v2f d8p[4];
                float d8s[4];
                rdP(8, d8p);
                rdS(8, d8s);
                load_raw(0, D[0]);
                loadB(0, 0, 0);
                float v8[4][3];
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const v2f x = point(p, d8p);
                    v8[p][0] = x.x; v8[p][1] = x.y;
                    v8[p][2] = p == 0 ? d8s[0] - d8s[2] : p == 1 ? d8s[1] + d8s[2] : p == 2 ? d8s[2] - d8s[1] : d8s[1] - d8s[3];
                }
                __builtin_amdgcn_sched_barrier(0);
                auto fwin = [&](int p) {
                    if (DBG & 2) { acc[p][0] += (v8[p][0] + v8[p][1]) + v8[p][2]; return; }
                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(v8[p][0], w8[p][0], zero, 0, 0, 0);
#pragma unroll
                    for (int st = 1; st < 3; ++st) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(v8[p][st], w8[p][st], acc[p], 0, 0, 0);
                };
                fwin(0); prep(0, D[0], 0);
                B3_WIN3
                fwin(1); finish_prev();
                B3_WIN3
                fwin(2);
                B3_WIN3
                fwin(3);
                B3_WIN3
#pragma unroll
                for (int u = 0; u < 12; ++u) {
                    const int ks = u >> 2, p = u & 3;
                    mma(p, u & 1);
                    if (u + 1 < 12) {
                        const int kn = (u + 1) >> 2, pn = (u + 1) & 3;
                        loadB(kn, pn, (u + 1) & 1);
                        prep(pn, D[kn & 1], (u + 1) & 1);
                    }
                    if (p == 1 && ks < 2) load_raw(ks + 1, D[(ks + 1) & 1]);
                    B3_WIN6
                }
#undef B3_PIPE
#undef B3_WIN6
#undef B3_WIN3
                // ---- the pair's two outputs, the 2^3 pool (4 mates x the pair), bias; the chain waits for the next tile
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float m = 0.f;
#pragma unroll
                    for (int i = 4 * q; i < 4 * q + 4; ++i) {
                        const float x0 = (acc[0][i] + acc[1][i]) + acc[2][i];
                        const float x1 = (acc[1][i] - acc[2][i]) - acc[3][i];
                        const float mx = fmaxf(x0, x1);
                        m = i == 4 * q ? mx : fmaxf(m, mx);
                    }
                    mprev[q] = m + bv;
                }
                tprev = t;
                oprev = outb;
            }
            commit();
            S = S2;
            __syncthreads();
        }
    }
    finish_prev();
}

typedef void (*FirstB3Kernel)(const ConvFirstB3Args);
struct FirstB3Dbg { int code; FirstB3Kernel k; };
const FirstB3Dbg kFirstB3Dbg[] = {{1, k_conv_first_b3<1>}, {2, k_conv_first_b3<2>}, {3, k_conv_first_b3<3>}, {4, k_conv_first_b3<4>},
                                  {8, k_conv_first_b3<8>}, {64, k_conv_first_b3<64>}, {131, k_conv_first_b3<131>}, {259, k_conv_first_b3<259>},
                                  {399, k_conv_first_b3<399>}};

inline uint16_t b3_bf16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline double b3_bf16_val(uint16_t b) {
    const uint32_t u = (uint32_t)b << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return (double)f;
}

}  // namespace

// does the split kernel serve this first layer (the planner asks before it packs the weights)?
bool conv_first_b3_ok(const ConvMfmaPlan& p, int Din, int Hin, int Win, int Cin, int Cout, const ConvGeom& g, const PostOps& post) {
    const ThKnobs& kn = th_knobs_of(p.knobs);
    if (!kn.first_split || !p.first_wino || kn.no_pool_first) return false;
    if (Din != kFD || Hin != kFD || Win != kFD || (Cin != 5 && Cin != 6) || Cout < 1 || Cout > 32) return false;
    if (g.kd != 3 || g.kh != 3 || g.kw != 3 || g.pz != 1 || g.py != 1 || g.px != 1) return false;
    if (p.pool != 1 || !post.monotone || p.Dc != 20 || p.Hc != 20 || p.Wc != 20) return false;
    return true;
}

size_t conv_first_b3_wpk_floats() { return (size_t)kB3WFrag * 4 + kB3W8; }

// MFMA FLOPs issued per frame: 125 tiles x (72 instructions of 32 x 32 x 16 on the bf16 pipe + 12 of 32 x 32 x 2 fp32)
double conv_first_b3_exec_flops() { return 2.0 * kTiles * (72.0 * 32 * 32 * 16 + 12.0 * 32 * 32 * 2); }

std::string conv_first_b3_label(int nnb) {
    (void)nnb;
    char buf[256];
    snprintf(buf, sizeof buf, "conv_first_b3<F(2,3) along x; pool before the chain> persistent, ring of %d planes, lds%zuK; bf16x3 split operands, 6 products, "
             "fp32 accumulate; uint8 / bool frames: one-piece data, 3 products (weights in LDS, direct input) [k_conv_first_b3]", kRing, kB3Lds / 1024);
    return buf;
}

// Keras [3,3,3,Cin,Cout] -> U_p = sum_k G[p][k] W[dz][dy][k] (double).  Taps 0..7 split into three bf16 pieces,
// [p][ks][piece][lane = 32 h + co][e]: k-slot e of k-step ks is (tap, s) = kB3Slot[ks][e], channel 2 s + h; behind them tap 8
// as fp32, [p][step][lane]: channel 2 step + h
void conv_first_b3_pack_weights(int Cin, int Cout, const float* w, float* dst_f) {
    static const double Gm[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    uint16_t* dst = reinterpret_cast<uint16_t*>(dst_f);
    std::memset(dst_f, 0, conv_first_b3_wpk_floats() * sizeof(float));
    auto U = [&](int p, int t9, int c, int co) {
        double u = 0;
        for (int k = 0; k < 3; ++k) u += Gm[p][k] * (double)w[((size_t)(t9 * 3 + k) * Cin + c) * Cout + co];
        return u;
    };
    for (int p = 0; p < 4; ++p)
        for (int ln = 0; ln < 64; ++ln) {
            const int hh = ln >> 5, co = ln & 31;
            if (co >= Cout) continue;
            for (int ks = 0; ks < 3; ++ks)
                for (int e = 0; e < 8; ++e) {
                    const int t9 = kB3Slot[ks][e].t, c = 2 * kB3Slot[ks][e].s + hh;
                    if (c >= Cin) continue;
                    const double u = U(p, t9, c, co);
                    uint16_t pc[3];
                    pc[0] = b3_bf16_rne((float)u);
                    const double r1 = u - b3_bf16_val(pc[0]);
                    pc[1] = b3_bf16_rne((float)r1);
                    const double r2 = r1 - b3_bf16_val(pc[1]);
                    pc[2] = b3_bf16_rne((float)r2);
                    for (int piece = 0; piece < 3; ++piece)
                        dst[((((size_t)p * 3 + ks) * 3 + piece) * 64 + ln) * 8 + e] = pc[piece];
                }
            for (int st = 0; st < 3; ++st) {
                const int c = 2 * st + hh;
                if (c < Cin) dst_f[(size_t)kB3WFrag * 4 + (p * 3 + st) * 64 + ln] = (float)U(p, 8, c, co);
            }
        }
}

int launch_conv_first_b3(hipStream_t s, int64_t n, const ConvMfmaPlan& p, const void* frames, int dtype, int Cin, TView out, int Cout,
                         const float* wpk, const float* bias, PostOps post) {
    if (n <= 0) return TH_OK;
    ConvFirstB3Args a;
    std::memset(&a, 0, sizeof a);
    a.in = frames; a.dtype = dtype; a.Cin = Cin;
    a.vec8 = (dtype == TH_F32 && Cin == 6 && ((uintptr_t)frames % 8) == 0) ? 1 : 0;
    a.wpk = reinterpret_cast<const uint4*>(wpk);
    a.Cout = Cout; a.bias = bias; a.post = post;
    a.out = out.p; a.out_fs = out.fs; a.out_cs = out.cs; a.out_coff = out.coff; a.Ho = out.H; a.Wo = out.W;
    if (out.blk) {
        if (out.blk != 4 || out.coff || out.cs != Cout || Cout % 4) TH_FAIL(TH_EINVAL, "conv_first_b3: bad chunk-blocked output view");
        a.out_cs = 4;
        a.out_blk_stride = out.D * out.H * out.W * 4;
    }
    a.nframes = n;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    const ThKnobs& kn = th_knobs_of(p.knobs);
    int64_t resident = ncu;                              // one 8-wave workgroup per CU (139 KB of LDS)
    if (kn.wf_resident) resident = std::max(1, kn.wf_resident);       // tests: force multi-frame workgroups on small batches
    // equal trips: every workgroup streams ceil(n / grid) or one frame fewer
    const int64_t trips = (n + resident - 1) / resident;
    const int64_t grid = (n + trips - 1) / trips;
    // integer frames (uint8 / bool): one-piece data, 3 products (TH_FIRST_INT=0: the general kernel, same bits)
    const bool ints = (dtype == TH_U8 || dtype == TH_BOOL) && kn.first_int;
    FirstB3Kernel k = ints ? k_conv_first_b3<0, 1> : k_conv_first_b3<0>;
    if (kn.first_dbg > 0)
        for (const FirstB3Dbg& d : kFirstB3Dbg) if (d.code == kn.first_dbg) k = d.k;
    HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(512), kB3Lds, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) TH_FAIL(TH_EHIP, "conv_first_b3 launch failed: %s", hipGetErrorString(e));
    return TH_OK;
}
