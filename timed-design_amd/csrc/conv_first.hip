// First-layer fused Conv3D for gfx950: Conv3D(3x3x3, stride 1, Cin <= 8, Cout <= 32) + bias +
// {act | BN-affine}* + optional 2x2x2 max/avg pool, reading the CALLER'S frames directly
// (fp32 / fp64 / uint8 / bool / fp16 channels-last, exactly what load_batch builds — reference
// design_utils/utils.py:518-527 — and what Keras' predict() casts to fp32 first).
//
// Why a separate kernel: the first TIMED block (6 -> 32 channels on 21^3 voxels) is 15 % of the
// FLOPs but has a tiny K (27 taps x 6 channels), so the generic kernel's costs dominate it.  Here:
//   * the whole weight tensor (27 x 32 x 8 floats) lives in REGISTERS: lane (j = l&31, h = l>>5)
//     keeps, per tap, the 4 weights of output channel j that multiply the 4 channels the lane's MFMA
//     k-slot covers -> no weight traffic and no barrier inside the K loop;
//   * channel c sits in LDS slot (h = c&1, t = c>>1) of an 8-float voxel record, so MFMA step t
//     contracts channels (2t, 2t+1): Cin = 5..6 needs 3 steps per tap, not 4 (no zero-padded MFMA);
//   * no conversion pass: the staging loop reads the user's dtype and writes fp32 into LDS;
//   * bricks of ZB output planes keep the LDS image under 80 KiB -> 2 workgroups per CU, whose
//     staging / epilogue / ragged last round overlap each other's MFMA stream.
// GEMM view per workgroup: M = ZB x Hc x Wc output voxels (rows grouped 8 pool-mates at a time when
// pooled), N = 32, K = 27 x 2*NST.  v_mfma_f32_32x32x2_f32, exact fp32.
#include "common.h"
#include "device_math.h"

#include <hip/hip_fp16.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int kTaps = 27;

struct ConvFirstArgs {
    const void* in; int dtype; int Din, Hin, Win, Cin;
    int pz, py, px;
    int Dc, Hc, Wc;
    int ZB, nzb, Zp, Hp, Wp, rows, n_mtiles, tab_off, vec8;
    const float* wpk;
    int Cout;
    const float* bias;
    PostOps post;
    float* out; int64_t out_fs; int out_cs, out_coff, Ho, Wo;
    int out_blk_stride;   // > 0: chunk-blocked output (TView::blk): out_cs is 4 and channel co lives (co >> 2) * out_blk_stride + (co & 3) floats in
    int64_t nframes;
};

__device__ __forceinline__ float load_elem(const void* base, int dtype, int64_t i) {
    switch (dtype) {
        case TH_F32: return ((const float*)base)[i];
        case TH_F64: return (float)((const double*)base)[i];
        case TH_U8: return (float)((const unsigned char*)base)[i];
        case TH_BOOL: return ((const unsigned char*)base)[i] ? 1.f : 0.f;
        default: return __half2float(((const __half*)base)[i]);
    }
}

// PMODE: 0 no pool, 1 max 2x2x2, 2 avg 2x2x2, 3 max 2x2x2 taken BEFORE the epilogue chain (planner proved the chain
// monotone non-decreasing: PostOps::monotone)
// GEO > 0: Hp = Wp = GEO, Hc = Wc = GEO - 2 at compile time (3x3x3 'same' on cubic frames).  A workgroup lives for only
// ~13 us of MFMAs per wave, and building its row table and decoding its 1936 staged voxels costs ~29 integer divisions
// per thread: by run-time divisors that is ~1000 VALU instructions, by constants a tenth of that.
template <int WAVES, int NST, int PMODE, int GEO = 0>
__global__ void __launch_bounds__(WAVES * 64, 3) k_conv_first(const ConvFirstArgs a) {
    constexpr int POOL = PMODE == 3 ? 1 : PMODE;
    constexpr bool POOL_FIRST = PMODE == 3;
    constexpr int NTHREADS = WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;

    const int zb = blockIdx.x % a.nzb;
    const int64_t f = blockIdx.x / a.nzb;
    const int z0 = zb * a.ZB;
    const int gHp = GEO ? GEO : a.Hp, gWp = GEO ? GEO : a.Wp, gHc = GEO ? GEO - 2 : a.Hc, gWc = GEO ? GEO - 2 : a.Wc;
    const int nvox = a.Zp * gHp * gWp;
    // voxel record of REC = 2*NST floats; MFMA step t contracts channels (2t, 2t+1), lane half h supplies 2t+h.
    //   NST=4: [c0 c2 c4 c6 | c1 c3 c5 c7]  one ds_read_b128 at 4h
    //   NST=3: [c0 c2 | c1 c3 | c4 | c5]    ds_read_b64 at 2h + ds_read_b32 at 4+h   (24-byte records)
    //   NST=2: [c0 c2 | c1 c3]              ds_read_b64 at 2h
    //   NST=1: [c0 | c1]                    ds_read_b32 at h
    constexpr int REC = 2 * NST;
    float* A = reinterpret_cast<float*>(smem);
    int* rowvox = (int*)(reinterpret_cast<char*>(smem) + a.tab_off);
    int* rowout = rowvox + a.rows;

    // ---- weights -> registers ------------------------------------------------------------------------
    float4 breg[kTaps];
    {
        const float4* w4 = reinterpret_cast<const float4*>(a.wpk);
#pragma unroll
        for (int t = 0; t < kTaps; ++t) breg[t] = w4[(t * 2 + h) * 32 + j];
    }

    // ---- row tables ------------------------------------------------------------------------------------
    {
        const int ZBv = min(a.ZB, a.Dc - z0);
        for (int r = tid; r < a.rows; r += NTHREADS) {
            int vox = 0, oo = -1;
            if (POOL == 0) {
                const int hw = gHc * gWc;
                if (r < ZBv * hw) {
                    const int zl = r / hw, rem = r - zl * hw, y = rem / gWc, x = rem - y * gWc;
                    vox = (zl * gHp + y) * gWp + x;
                    oo = (((z0 + zl) * a.Ho + y) * a.Wo + x) * a.out_cs;
                }
                rowout[r] = oo;
            } else {
                const int pq = r >> 3, mate = r & 7;
                const int PH = gHc >> 1, PW = gWc >> 1;
                if (pq < (ZBv >> 1) * PH * PW) {
                    const int pzz = pq / (PH * PW), rem = pq - pzz * (PH * PW), pyy = rem / PW, pxx = rem - pyy * PW;
                    const int zl = 2 * pzz + (mate >> 2), y = 2 * pyy + ((mate >> 1) & 1), x = 2 * pxx + (mate & 1);
                    vox = (zl * gHp + y) * gWp + x;
                    oo = ((((z0 >> 1) + pzz) * a.Ho + pyy) * a.Wo + pxx) * a.out_cs;
                }
                if (mate == 0) rowout[r >> 3] = oo;
            }
            rowvox[r] = vox;
        }
    }

    // ---- stage the haloed input brick straight from the caller's frames ------------------------------
    {
        const int64_t fbase = f * (int64_t)a.Din * a.Hin * a.Win * a.Cin;
        const bool fast6 = (a.vec8 && NST == 3);  // 24-byte voxels: three 8-byte loads
        constexpr int U = 4;  // voxels in flight per thread: their global loads overlap
        for (int vb = tid; vb < nvox; vb += NTHREADS * U) {
            float e[U][8];
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int c = 0; c < 8; ++c) e[u][c] = 0.f;
                const int v = vb + u * NTHREADS;
                const int xl = v % gWp; int t = v / gWp;
                const int yl = t % gHp; const int zl = t / gHp;
                const int zi = z0 + zl - a.pz, yi = yl - a.py, xi = xl - a.px;
                if (v < nvox && zi >= 0 && zi < a.Din && yi >= 0 && yi < a.Hin && xi >= 0 && xi < a.Win) {
                    const int64_t base = fbase + ((int64_t)(zi * a.Hin + yi) * a.Win + xi) * a.Cin;
                    if (fast6) {
                        const float2* p2 = reinterpret_cast<const float2*>((const float*)a.in + base);
                        const float2 u0 = p2[0], u1 = p2[1], u2 = p2[2];
                        e[u][0] = u0.x; e[u][1] = u0.y; e[u][2] = u1.x; e[u][3] = u1.y; e[u][4] = u2.x; e[u][5] = u2.y;
                    } else {
#pragma unroll
                        for (int c = 0; c < 2 * NST; ++c)
                            if (c < a.Cin) e[u][c] = load_elem(a.in, a.dtype, base + c);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int v = vb + u * NTHREADS;
                if (v >= nvox) continue;
                float* rec = A + (size_t)v * REC;
                if (NST == 4) {
                    *reinterpret_cast<float4*>(rec) = make_float4(e[u][0], e[u][2], e[u][4], e[u][6]);
                    *reinterpret_cast<float4*>(rec + 4) = make_float4(e[u][1], e[u][3], e[u][5], e[u][7]);
                } else if (NST == 3) {
                    *reinterpret_cast<float2*>(rec) = make_float2(e[u][0], e[u][2]);
                    *reinterpret_cast<float2*>(rec + 2) = make_float2(e[u][1], e[u][3]);
                    *reinterpret_cast<float2*>(rec + 4) = make_float2(e[u][4], e[u][5]);
                } else if (NST == 2) {
                    *reinterpret_cast<float4*>(rec) = make_float4(e[u][0], e[u][2], e[u][1], e[u][3]);
                } else {
                    *reinterpret_cast<float2*>(rec) = make_float2(e[u][0], e[u][1]);
                }
            }
        }
    }
    __syncthreads();

    const int rounds = (a.n_mtiles + WAVES - 1) / WAVES;
    float* outb = a.out + f * a.out_fs + a.out_coff;
    const int co = j;
    const int cofs = a.out_blk_stride ? (co >> 2) * a.out_blk_stride + (co & 3) : co;
    const bool cok = co < a.Cout;
    const int cc = cok ? co : 0;
    const float bv = a.bias ? a.bias[cc] : 0.f;

    for (int rd = 0; rd < rounds; ++rd) {
        const int mt = rd * WAVES + wave;
        if (mt >= a.n_mtiles) break;  // wave-uniform; no barriers below
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const float* arow = A + (size_t)rowvox[mt * 32 + j] * REC;
        // fully unrolled over the 27 taps; tap t+1's LDS reads are issued before tap t's MFMAs (ping-pong
        // registers, pinned with sched_barrier) so the wave never sits on an LDS round trip
        auto fetch = [&](int dz, int dy, int dx, float (&av)[4]) {
            const float* rec = arow + ((dz * gHp + dy) * gWp + dx) * REC;
            if (NST == 4) {
                const float4 q = *reinterpret_cast<const float4*>(rec + 4 * h);
                av[0] = q.x; av[1] = q.y; av[2] = q.z; av[3] = q.w;
            } else if (NST == 3) {
                const float2 q = *reinterpret_cast<const float2*>(rec + 2 * h);
                av[0] = q.x; av[1] = q.y; av[2] = rec[4 + h];
            } else if (NST == 2) {
                const float2 q = *reinterpret_cast<const float2*>(rec + 2 * h);
                av[0] = q.x; av[1] = q.y;
            } else {
                av[0] = rec[h];
            }
        };
        float av[2][4];
        fetch(0, 0, 0, av[0]);
#pragma unroll
        for (int t = 0; t < kTaps; ++t) {
            if (t + 1 < kTaps) fetch((t + 1) / 9, ((t + 1) / 3) % 3, (t + 1) % 3, av[(t + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t & 1][0], breg[t].x, acc, 0, 0, 0);
            if (NST > 1) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t & 1][1], breg[t].y, acc, 0, 0, 0);
            if (NST > 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t & 1][2], breg[t].z, acc, 0, 0, 0);
            if (NST > 3) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t & 1][3], breg[t].w, acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue in registers (see conv_mfma.hip for the row/lane layout) ----------------------
        float x[16];
        if (POOL_FIRST) {
            // (the bias is added to the 2 pooled values, not to all 16 sums: max commutes with "+ bv" bit for bit)
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = acc[i];
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = acc[i] + bv;
        }
        if (POOL_FIRST) {
            // max-pooling commutes with a monotone non-decreasing chain (ELU/ReLU/BN with scale >= 0 ...): pool the raw
            // sums and evaluate the chain on the 4 pooled values of this lane instead of all 16 (TIMED block 1: 8x fewer
            // ELU exponentials)
            float m4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float m = fmaxf(fmaxf(x[4 * q], x[4 * q + 1]), fmaxf(x[4 * q + 2], x[4 * q + 3]));
                m4[q] = fmaxf(m, __shfl_xor(m, 32));
            }
            // lanes of half h own pooled voxels 2h and 2h+1 of the tile
            float y0 = (h ? m4[2] : m4[0]) + bv, y1 = (h ? m4[3] : m4[1]) + bv;
            th_post2(y0, y1, cc, a.post);
            const int o0 = cok ? rowout[mt * 4 + 2 * h] : -1, o1 = cok ? rowout[mt * 4 + 2 * h + 1] : -1;
            if (o0 >= 0) outb[o0 + cofs] = y0;
            if (o1 >= 0) outb[o1 + cofs] = y1;
            continue;
        }
        th_post16(x, cc, a.post);
        if (POOL == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int oo = cok ? rowout[mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h] : -1;
                if (oo >= 0) outb[oo + cofs] = x[i];
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float m;
                if (POOL == 1) m = fmaxf(fmaxf(x[4 * q], x[4 * q + 1]), fmaxf(x[4 * q + 2], x[4 * q + 3]));
                else m = (x[4 * q] + x[4 * q + 1]) + (x[4 * q + 2] + x[4 * q + 3]);
                const float o2 = __shfl_xor(m, 32);
                m = (POOL == 1) ? fmaxf(m, o2) : (m + o2) * 0.125f;
                const int oo = cok ? rowout[mt * 4 + q] : -1;
                if (oo >= 0 && (q >> 1) == h) outb[oo + cofs] = m;
            }
        }
    }
}


// ---- F(2,3) along x ---------------------------------------------------------------------------------------------------------
// The same brick, the same raw LDS image, the same register-resident weights — but a GEMM row is a PAIR of output voxels
// (2 tx, 2 tx + 1) and each (dz, dy) tap contributes 4 products per channel instead of 6 (3 x taps x 2 outputs): four transform
// points V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3 of the four input voxels d0..d3 = x 2 tx - 1 .. 2 tx + 2, formed
// in registers right after the LDS reads (one VALU op per MFMA), multiplied with U_p = G W (the halves folded into the weights on
// the host in double precision) into FOUR independent accumulators; out(2 tx) = M0 + M1 + M2, out(2 tx + 1) = M1 - M2 - M3.
// 1.5x fewer MFMAs, exact data transform (0 / +-1).  With a pool the pair is the window's x extent and the rows are grouped four
// (dz, dy) mates at a time, so the whole 2^3 window sits in one lane's registers (no cross-lane step at all).  The four
// accumulators also end the single-accumulator chain that made k_conv_first need three waves per SIMD; this one runs two
// workgroups per CU on 4-plane bricks (25 row tiles, 70 KB of image).
// KN: timing knock-outs (TH_FIRST_DBG, results are WRONG): 1 one LDS fetch per tile, 2 no transform adds, 4 no epilogue chain / stores
template <int NST, int PMODE, int GEO = 0, int KN = 0>
__global__ void __launch_bounds__(256, 2) k_conv_first_w(const ConvFirstArgs a) {
    constexpr int POOL = PMODE == 3 ? 1 : PMODE;
    constexpr bool POOL_FIRST = PMODE == 3;
    constexpr int NTHREADS = 256, WAVES = 4;
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;

    const int zb = blockIdx.x % a.nzb;
    const int64_t f = blockIdx.x / a.nzb;
    const int z0 = zb * a.ZB;
    const int gHp = GEO ? GEO : a.Hp, gWp = GEO ? GEO : a.Wp, gHc = GEO ? GEO - 2 : a.Hc, gWc = GEO ? GEO - 2 : a.Wc;
    const int TX = gWc >> 1;
    const int nvox = a.Zp * gHp * gWp;
    constexpr int REC = 2 * NST;
    float* A = reinterpret_cast<float*>(smem);
    int* rowvox = (int*)(reinterpret_cast<char*>(smem) + a.tab_off);
    int* rowout = rowvox + a.rows;


    // ---- row tables: rowvox = staged voxel of d0 at tap (0, 0); rowout = the pair's first output voxel / the pooled voxel
    {
        const int ZBv = min(a.ZB, a.Dc - z0);
        for (int r = tid; r < a.rows; r += NTHREADS) {
            int vox = 0, oo = -1;
            if (POOL == 0) {
                const int hw = gHc * TX;
                if (r < ZBv * hw) {
                    const int zl = r / hw, rem = r - zl * hw, y = rem / TX, tx = rem - y * TX;
                    vox = (zl * gHp + y) * gWp + 2 * tx;
                    oo = (((z0 + zl) * a.Ho + y) * a.Wo + 2 * tx) * a.out_cs;
                }
                rowout[r] = oo;
            } else {
                const int pq = r >> 2, mate = r & 3;
                const int PH = gHc >> 1;
                if (pq < (ZBv >> 1) * PH * TX) {
                    const int pzz = pq / (PH * TX), rem = pq - pzz * (PH * TX), pyy = rem / TX, pxx = rem - pyy * TX;
                    const int zl = 2 * pzz + (mate >> 1), y = 2 * pyy + (mate & 1);
                    vox = (zl * gHp + y) * gWp + 2 * pxx;
                    oo = ((((z0 >> 1) + pzz) * a.Ho + pyy) * a.Wo + pxx) * a.out_cs;
                }
                if (mate == 0) rowout[r >> 2] = oo;
            }
            rowvox[r] = vox;
        }
    }

    // ---- stage the haloed input brick straight from the caller's frames (as k_conv_first) ----------------------------
    {
        const int64_t fbase = f * (int64_t)a.Din * a.Hin * a.Win * a.Cin;
        const bool fast6 = (a.vec8 && NST == 3);
        constexpr int U = 12;  // voxels in flight per thread: a 6-plane brick of 22 x 22 voxels is 11.3 per thread — one batch of loads
        for (int vb = tid; vb < nvox; vb += NTHREADS * U) {
            float e[U][8];
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int c = 0; c < 8; ++c) e[u][c] = 0.f;
                const int v = vb + u * NTHREADS;
                const int xl = v % gWp; int t = v / gWp;
                const int yl = t % gHp; const int zl = t / gHp;
                const int zi = z0 + zl - a.pz, yi = yl - a.py, xi = xl - a.px;
                if (v < nvox && zi >= 0 && zi < a.Din && yi >= 0 && yi < a.Hin && xi >= 0 && xi < a.Win) {
                    const int64_t base = fbase + ((int64_t)(zi * a.Hin + yi) * a.Win + xi) * a.Cin;
                    if (fast6) {
                        const float2* p2 = reinterpret_cast<const float2*>((const float*)a.in + base);
                        const float2 u0 = p2[0], u1 = p2[1], u2 = p2[2];
                        e[u][0] = u0.x; e[u][1] = u0.y; e[u][2] = u1.x; e[u][3] = u1.y; e[u][4] = u2.x; e[u][5] = u2.y;
                    } else {
#pragma unroll
                        for (int c = 0; c < 2 * NST; ++c)
                            if (c < a.Cin) e[u][c] = load_elem(a.in, a.dtype, base + c);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int v = vb + u * NTHREADS;
                if (v >= nvox) continue;
                float* rec = A + (size_t)v * REC;
                if (NST == 4) {
                    *reinterpret_cast<float4*>(rec) = make_float4(e[u][0], e[u][2], e[u][4], e[u][6]);
                    *reinterpret_cast<float4*>(rec + 4) = make_float4(e[u][1], e[u][3], e[u][5], e[u][7]);
                } else if (NST == 3) {
                    *reinterpret_cast<float2*>(rec) = make_float2(e[u][0], e[u][2]);
                    *reinterpret_cast<float2*>(rec + 2) = make_float2(e[u][1], e[u][3]);
                    *reinterpret_cast<float2*>(rec + 4) = make_float2(e[u][4], e[u][5]);
                } else if (NST == 2) {
                    *reinterpret_cast<float4*>(rec) = make_float4(e[u][0], e[u][2], e[u][1], e[u][3]);
                } else {
                    *reinterpret_cast<float2*>(rec) = make_float2(e[u][0], e[u][1]);
                }
            }
        }
    }
    // ---- weights -> registers: U_p of tap (dz, dy), the k-steps in x / y / z / w (after the staging loop: its 12 voxels in
    // flight per thread need the registers, and the weights come from L2)
    float4 breg[36];
    {
        const float4* w4 = reinterpret_cast<const float4*>(a.wpk);
#pragma unroll
        for (int t = 0; t < 36; ++t) breg[t] = w4[(t * 2 + h) * 32 + j];
    }
    __syncthreads();

    const int rounds = (a.n_mtiles + WAVES - 1) / WAVES;
    float* outb = a.out + f * a.out_fs + a.out_coff;
    const int co = j;
    const int cofs = a.out_blk_stride ? (co >> 2) * a.out_blk_stride + (co & 3) : co;
    const bool cok = co < a.Cout;
    const int cc = cok ? co : 0;
    const float bv = a.bias ? a.bias[cc] : 0.f;

    // POOL_FIRST: a tile leaves 4 pooled sums per lane; their epilogue chain (exp / BatchNorm constants from memory), the output
    // offsets (LDS) and the stores are DEFERRED into the next tile's tap loop (a slot between two MFMA bursts) instead of standing
    // between the two tiles' MFMAs (knock-out: the chain + stores cost 12 % of the kernel there)
    float mprev[4] = {0.f, 0.f, 0.f, 0.f};
    int mtprev = -1;
    // the two chains TIMED / DenseCPD blocks use get straight-line code with their constants in registers (the generic th_post2
    // decodes the op list — kernel-argument loads, a switch per op — and fetches the BatchNorm constants for every tile: ~650
    // cycles per tile and wave): 1 = ELU -> BN-affine, 2 = BN-affine -> ReLU, 0 = generic
    int epi = 0;
    float esc = 1.f, esh = 0.f, ealpha = 1.f;
    if (POOL_FIRST && a.post.n == 2) {
        if (a.post.type[0] == POP_ACT && a.post.act[0] == ACT_ELU && a.post.type[1] == POP_AFFINE) {
            epi = 1; ealpha = a.post.alpha[0]; esc = a.post.scale[1][cc]; esh = a.post.shift[1][cc];
        } else if (a.post.type[0] == POP_AFFINE && a.post.type[1] == POP_ACT && a.post.act[1] == ACT_RELU) {
            epi = 2; esc = a.post.scale[0][cc]; esh = a.post.shift[0][cc];
        }
    }
    epi = __builtin_amdgcn_readfirstlane(epi);
    auto finish_prev = [&]() {
        if (mtprev < 0) return;
        if (KN & 4) { if (mprev[0] + mprev[1] + mprev[2] + mprev[3] == 1.2345e-30f) outb[0] = mprev[0]; mtprev = -1; return; }
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
            const int oo = cok ? rowout[mtprev * 8 + 2 * q + h] : -1;
            if (oo >= 0) outb[oo + cofs] = mprev[q];
        }
        mtprev = -1;
    };
    for (int rd = 0; rd < rounds; ++rd) {
        const int mt = rd * WAVES + wave;
        if (mt >= a.n_mtiles) break;  // wave-uniform; no barriers below
        f32x16 acc[4];
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[p][i] = 0.f;
        const float* arow = A + (size_t)rowvox[mt * 32 + j] * REC;
        // the four input voxels of tap t9 = 3 dz + dy, this lane's k-slot of every step
        // (two per-lane bases — the lane half's slot of the wide read and of NST = 3's odd third — so that every tap / voxel
        // displacement below is an immediate of the ds_read when the geometry is a template constant)
        const float* arow_a = arow + (NST == 4 ? 4 * h : NST == 1 ? h : 2 * h);
        const float* arow_b = arow + 4 + h;
        auto fetch = [&](int t9, float (&d)[4][4]) {
            const int off0 = (((t9 / 3) * gHp + (t9 % 3)) * gWp) * REC;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int off = off0 + i * REC;
                if (NST == 4) {
                    const float4 q = *reinterpret_cast<const float4*>(arow_a + off);
                    d[i][0] = q.x; d[i][1] = q.y; d[i][2] = q.z; d[i][3] = q.w;
                } else if (NST == 3) {
                    const float2 q = *reinterpret_cast<const float2*>(arow_a + off);
                    d[i][0] = q.x; d[i][1] = q.y; d[i][2] = arow_b[off];
                } else if (NST == 2) {
                    const float2 q = *reinterpret_cast<const float2*>(arow_a + off);
                    d[i][0] = q.x; d[i][1] = q.y;
                } else {
                    d[i][0] = arow_a[off];
                }
            }
        };
        // The wave alternates between a burst of 4 NST MFMAs (tap t, its four points already in registers) and a short group
        // of everything else (the points of tap t + 1 from the raw voxels read a tap ago, the reads of tap t + 2): whatever a
        // wave issues BETWEEN two of its MFMAs delays the second one by about its own issue time, whether or not it depends on
        // it (measured: a lone wave per SIMD ran this loop at 56 % of the pipe with one subtraction in front of every MFMA).
        float d[4][4], v[2][4][4];
        auto points = [&](float (&vv)[4][4]) {
#pragma unroll
            for (int s = 0; s < NST; ++s) {
                if (KN & 2) { vv[0][s] = d[0][s]; vv[1][s] = d[1][s]; vv[2][s] = d[2][s]; vv[3][s] = d[3][s]; continue; }
                vv[0][s] = d[0][s] - d[2][s];
                vv[1][s] = d[1][s] + d[2][s];
                vv[2][s] = d[2][s] - d[1][s];
                vv[3][s] = d[1][s] - d[3][s];
            }
        };
        fetch(0, d);
        points(v[0]);
        fetch(1, d);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (t + 1 < 9) points(v[(t + 1) & 1]);
            if (t + 2 < 9 && !(KN & 1)) fetch(t + 2, d);
            if (POOL_FIRST && t == 2) finish_prev();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < NST; ++s) {
                const float w0 = s == 0 ? breg[t].x : s == 1 ? breg[t].y : s == 2 ? breg[t].z : breg[t].w;
                const float w1 = s == 0 ? breg[9 + t].x : s == 1 ? breg[9 + t].y : s == 2 ? breg[9 + t].z : breg[9 + t].w;
                const float w2 = s == 0 ? breg[18 + t].x : s == 1 ? breg[18 + t].y : s == 2 ? breg[18 + t].z : breg[18 + t].w;
                const float w3 = s == 0 ? breg[27 + t].x : s == 1 ? breg[27 + t].y : s == 2 ? breg[27 + t].z : breg[27 + t].w;
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[t & 1][0][s], w0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[t & 1][1][s], w1, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[t & 1][2][s], w2, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[t & 1][3][s], w3, acc[3], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- epilogue in registers: the pair's two outputs, then as k_conv_first --------------------------------------
        float x0[16], x1[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            x0[i] = (acc[0][i] + acc[1][i]) + acc[2][i];
            x1[i] = (acc[1][i] - acc[2][i]) - acc[3][i];
        }
        if (POOL_FIRST) {
            // rows 4 q .. 4 q + 3 of this lane are the (dz, dy) mates of pooled voxel 2 q + h of the tile; the pair is its x extent
            float m4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                m4[q] = fmaxf(fmaxf(fmaxf(x0[4 * q], x0[4 * q + 1]), fmaxf(x0[4 * q + 2], x0[4 * q + 3])),
                              fmaxf(fmaxf(x1[4 * q], x1[4 * q + 1]), fmaxf(x1[4 * q + 2], x1[4 * q + 3]))) + bv;
#pragma unroll
            for (int q = 0; q < 4; ++q) mprev[q] = m4[q];
            mtprev = mt;
            continue;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) { x0[i] += bv; x1[i] += bv; }
        th_post16(x0, cc, a.post);
        th_post16(x1, cc, a.post);
        if (POOL == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int oo = cok ? rowout[mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h] : -1;
                if (oo >= 0) { outb[oo + cofs] = x0[i]; outb[oo + a.out_cs + cofs] = x1[i]; }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float m;
                if (POOL == 1) m = fmaxf(fmaxf(fmaxf(x0[4 * q], x0[4 * q + 1]), fmaxf(x0[4 * q + 2], x0[4 * q + 3])),
                                         fmaxf(fmaxf(x1[4 * q], x1[4 * q + 1]), fmaxf(x1[4 * q + 2], x1[4 * q + 3])));
                else m = (((x0[4 * q] + x0[4 * q + 1]) + (x0[4 * q + 2] + x0[4 * q + 3])) + ((x1[4 * q] + x1[4 * q + 1]) + (x1[4 * q + 2] + x1[4 * q + 3]))) * 0.125f;
                const int oo = cok ? rowout[mt * 8 + 2 * q + h] : -1;
                if (oo >= 0) outb[oo + cofs] = m;
            }
        }
    }
    if (POOL_FIRST) finish_prev();
}

typedef void (*FirstKernel)(const ConvFirstArgs);
constexpr int kWaves = 4;
#define ROW(NST) { k_conv_first<kWaves, NST, 0>, k_conv_first<kWaves, NST, 1>, k_conv_first<kWaves, NST, 2>, k_conv_first<kWaves, NST, 3> }
const FirstKernel kFirstKernels[4][4] = {ROW(1), ROW(2), ROW(3), ROW(4)};
// compile-time row geometry for 21^3 x 6 aposteriori frames: pooled first block (20 + 2) and unpooled 'same' (21 + 2)
struct FirstGeo { int nst, pmode, geo; FirstKernel k; };
const FirstGeo kFirstGeo[] = {
    {3, 3, 22, k_conv_first<kWaves, 3, 3, 22>},
    {3, 1, 22, k_conv_first<kWaves, 3, 1, 22>},
    {3, 0, 23, k_conv_first<kWaves, 3, 0, 23>},
};

#define ROWW(NST) { k_conv_first_w<NST, 0>, k_conv_first_w<NST, 1>, k_conv_first_w<NST, 2>, k_conv_first_w<NST, 3> }
const FirstKernel kFirstWKernels[4][4] = {ROWW(1), ROWW(2), ROWW(3), ROWW(4)};
const FirstGeo kFirstWGeo[] = {
    {3, 3, 22, k_conv_first_w<3, 3, 22>},
    {3, 1, 22, k_conv_first_w<3, 1, 22>},
};
const FirstKernel kFirstWDbg[8] = {nullptr, k_conv_first_w<3, 3, 22, 1>, k_conv_first_w<3, 3, 22, 2>, k_conv_first_w<3, 3, 22, 3>,
                                   k_conv_first_w<3, 3, 22, 4>, nullptr, nullptr, k_conv_first_w<3, 3, 22, 7>};

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

}  // namespace

bool conv_first_plan(int Din, int Hin, int Win, int Cin, const TView& oc, const ConvGeom& g, int Cout, int pool,
                     ConvMfmaPlan* p) {
    if (g.kd != 3 || g.kh != 3 || g.kw != 3) return false;
    if (g.sd != 1 || g.sh != 1 || g.sw != 1 || g.dd != 1 || g.dh != 1 || g.dw != 1) return false;
    // the kernel holds the weights of 32 output channels in registers; a wider first layer (33 .. 128 filters) runs it once per
    // block of 32 columns (nnb passes over the caller's frames: a 64-filter layer costs two launches of 2.3 ms per 4096 frames
    // where the generic brick kernel, which the layer would otherwise fall to, takes 11 — tools/plan_report.py, timed_w40)
    if (Cin < 1 || Cin > 8 || Cout > 128) return false;
    if (pool && (oc.D < 2 || oc.H < 2 || oc.W < 2)) return false;
    p->cfg = 100;  // marks the first-layer kernel
    p->CI = 8; p->CS = 8; p->BN = 32; p->nnb = (Cout + 31) / 32; p->nchunks = 1; p->pool = pool; p->bres = 1; p->FB = 1;
    p->Dc = pool ? (oc.D / 2) * 2 : oc.D;
    p->Hc = pool ? (oc.H / 2) * 2 : oc.H;
    p->Wc = pool ? (oc.W / 2) * 2 : oc.W;
    p->Hp = p->Hc + 2;
    p->Wp = p->Wc + 2;
    // F(2,3) along x (k_conv_first_w): 'same' padding, an even number of computed columns; rows are x pairs
    const ThKnobs& kn = th_knobs_planning();
    p->knobs = &kn;
    const bool no_wino = !kn.first_wino;
    const bool wino = !no_wino && g.pz == 1 && g.py == 1 && g.px == 1 && p->Wc % 2 == 0 && p->Wc >= 2;
    p->first_wino = wino ? 1 : 0;
    auto rows_for = [&](int zb) {
        if (wino) return round_up(pool ? 4 * ((zb / 2) * (p->Hc / 2) * (p->Wc / 2)) : zb * p->Hc * (p->Wc / 2), 32);
        return round_up(pool ? 8 * ((zb / 2) * (p->Hc / 2) * (p->Wc / 2)) : zb * p->Hc * p->Wc, 32);
    };
    const int nst = (Cin + 1) / 2;
    const size_t rec_bytes = (size_t)8 * nst;
    auto img_for = [&](int zb) { return ((size_t)(zb + 2) * p->Hp * p->Wp * rec_bytes + 15) / 16 * 16; };
    auto lds_for = [&](int zb) {
        const int rows = rows_for(zb);
        return img_for(zb) + (size_t)rows * 4 + (size_t)(pool ? rows / (wino ? 4 : 8) : rows) * 4;
    };
    const int step = pool ? 2 : 1;
    // prefer bricks that leave room for two workgroups per CU (<= 80 KiB); fall back to one per CU
    int ZB = 0;
    for (size_t limit : {wino ? (size_t)80 * 1024 : (size_t)160 * 1024 / 3, (size_t)80 * 1024, (size_t)160 * 1024}) {
        for (int zb = p->Dc; zb >= step; zb -= step)
            if (lds_for(zb) <= limit) { ZB = zb; break; }
        if (ZB) break;
    }
    if (!ZB) return false;
    for (int zb = ZB; zb >= step && zb * 10 >= ZB * 6; zb -= step)
        if (p->Dc % zb == 0) { ZB = zb; break; }
    if (kn.first_zb) {  // tuning experiments
        const int zb = kn.first_zb;
        if (zb >= step && zb % step == 0 && zb <= p->Dc && lds_for(zb) <= (size_t)160 * 1024) ZB = zb;
    }
    p->ZB = ZB;
    p->nzb = (p->Dc + ZB - 1) / ZB;
    p->Zp = ZB + 2;
    p->rows_pf = rows_for(ZB);
    p->lds_bytes = lds_for(ZB);
    p->tab_off = img_for(ZB);
    p->wpk_floats = (size_t)(wino ? 36 : kTaps) * 2 * 32 * 4;
    p->exec_flops = 2.0 * (double)p->nzb * p->rows_pf * 32.0 * (2.0 * nst) * (wino ? 36 : kTaps) * p->nnb;
    p->own_flops = wino ? 2.0 * (double)p->Dc * p->Hc * (p->Wc / 2) * 36.0 * Cin * Cout : 0.0;
    if (oc.fs > 0x7fffffffLL) return false;
    char buf[224];
    if (wino)
        snprintf(buf, sizeof buf, "conv_first_w<F(2,3) along x; nst%d,pool%d> ZB%d/%d rows%d lds%zuK (weights in VGPRs, direct input) [k_conv_first_w<%d,%d>]",
                 nst, pool, ZB, p->Dc, p->rows_pf, p->lds_bytes / 1024, nst, pool);
    else
    snprintf(buf, sizeof buf, "conv_first<w%d,nst%d,pool%d> ZB%d/%d rows%d lds%zuK (weights in VGPRs, direct input) [k_conv_first<%d,%d,%d>]",
             kWaves, nst, pool, ZB, p->Dc, p->rows_pf, p->lds_bytes / 1024, kWaves, nst, pool);
    p->label = buf;
    return true;
}

// Keras [3,3,3,Cin,Cout] -> [tap][h][co(32)][t(4)] with channel c = 2t + h
void conv_first_pack_weights(int Cin, int Cout, const float* w, float* dst) {
    std::memset(dst, 0, (size_t)kTaps * 2 * 32 * 4 * sizeof(float));
    for (int tap = 0; tap < kTaps; ++tap)
        for (int c = 0; c < Cin; ++c)
            for (int co = 0; co < Cout; ++co)
                dst[(((size_t)tap * 2 + (c & 1)) * 32 + co) * 4 + (c >> 1)] = w[((size_t)tap * Cin + c) * Cout + co];
}

// Keras [3,3,3,Cin,Cout] -> U_p = sum_k G[p][k] W[dz][dy][k] (double), [p][3 dz + dy][h][co(32)][t(4)] with channel c = 2t + h
void conv_first_w_pack_weights(int Cin, int Cout, const float* w, float* dst) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    std::memset(dst, 0, (size_t)36 * 2 * 32 * 4 * sizeof(float));
    for (int t9 = 0; t9 < 9; ++t9)
        for (int c = 0; c < Cin; ++c)
            for (int co = 0; co < Cout; ++co)
                for (int p = 0; p < 4; ++p) {
                    double u = 0;
                    for (int k = 0; k < 3; ++k) u += G[p][k] * (double)w[((size_t)(t9 * 3 + k) * Cin + c) * Cout + co];
                    dst[((((size_t)p * 9 + t9) * 2 + (c & 1)) * 32 + co) * 4 + (c >> 1)] = (float)u;
                }
}

namespace {
// the instantiation a plan runs with a given epilogue chain: pool-first (PMODE 3) when the chain is monotone, the
// compile-time geometry when there is one for this frame size
FirstKernel pick_first_kernel(const ConvMfmaPlan& p, int nst, const PostOps& post, int* pmode, int* geo) {
    const ThKnobs& kn = th_knobs_of(p.knobs);
    const bool no_pool_first = kn.no_pool_first != 0;   // A/B comparisons and tests
    *pmode = (p.pool == 1 && post.monotone && !no_pool_first) ? 3 : p.pool;
    *geo = 0;
    FirstKernel k = p.first_wino ? kFirstWKernels[nst - 1][*pmode] : kFirstKernels[nst - 1][*pmode];
    if (p.Hp == p.Wp && p.Hc == p.Hp - 2 && p.Wc == p.Wp - 2 && !kn.conv_nogeo) {
        if (p.first_wino) {
            for (const FirstGeo& ge : kFirstWGeo)
                if (ge.nst == nst && ge.pmode == *pmode && ge.geo == p.Hp) { k = ge.k; *geo = ge.geo; }
        } else {
            for (const FirstGeo& ge : kFirstGeo)
                if (ge.nst == nst && ge.pmode == *pmode && ge.geo == p.Hp) { k = ge.k; *geo = ge.geo; }
        }
    }
    return k;
}
}  // namespace

// the plan's label with the kernel that launch_conv_first will actually run for this epilogue chain (as rocprofv3 prints it)
std::string conv_first_label(const ConvMfmaPlan& p, int Cin, const PostOps& post) {
    int pmode, geo;
    pick_first_kernel(p, (Cin + 1) / 2, post, &pmode, &geo);
    std::string l = p.label;
    const size_t b = l.rfind('[');
    if (b != std::string::npos) {
        char buf[64];
        if (p.first_wino) snprintf(buf, sizeof buf, "[k_conv_first_w<%d,%d,%d,0>]", (Cin + 1) / 2, pmode, geo);
        else snprintf(buf, sizeof buf, "[k_conv_first<%d,%d,%d,%d>]", kWaves, (Cin + 1) / 2, pmode, geo);
        l = l.substr(0, b) + buf;
    }
    return l;
}

int launch_conv_first(hipStream_t s, int64_t n, const ConvMfmaPlan& p, const void* frames, int dtype, int Din, int Hin,
                      int Win, int Cin, TView out, ConvGeom g, int Cout, const float* wpk, const float* bias, PostOps post) {
    ConvFirstArgs a;
    std::memset(&a, 0, sizeof a);
    a.in = frames; a.dtype = dtype; a.Din = Din; a.Hin = Hin; a.Win = Win; a.Cin = Cin;
    a.pz = g.pz; a.py = g.py; a.px = g.px;
    a.Dc = p.Dc; a.Hc = p.Hc; a.Wc = p.Wc;
    a.ZB = p.ZB; a.nzb = p.nzb; a.Zp = p.Zp; a.Hp = p.Hp; a.Wp = p.Wp;
    a.rows = p.rows_pf; a.n_mtiles = p.rows_pf / 32; a.tab_off = (int)p.tab_off;
    a.wpk = wpk; a.Cout = Cout; a.bias = bias; a.post = post;
    a.out = out.p; a.out_fs = out.fs; a.out_cs = out.cs; a.out_coff = out.coff; a.Ho = out.H; a.Wo = out.W;
    if (out.blk) {
        if (out.blk != 4 || out.coff || out.cs != Cout || Cout % 4) TH_FAIL(TH_EINVAL, "conv_first: bad chunk-blocked output view");
        a.out_cs = 4;
        a.out_blk_stride = out.D * out.H * out.W * 4;
    }
    a.nframes = n;
    a.vec8 = (dtype == TH_F32 && Cin == 6 && ((uintptr_t)frames % 8) == 0) ? 1 : 0;
    const int64_t grid = n * p.nzb;
    if (grid > 0x7fffffffLL) TH_FAIL(TH_EINVAL, "conv_first: grid too large");
    const int nst = (Cin + 1) / 2;
    int pmode, geo;
    FirstKernel k = pick_first_kernel(p, nst, post, &pmode, &geo);
    {   // timing experiments only (tools/bench_layer.py): knock-out instantiations of k_conv_first_w<3,3,22>
        const int dbg = th_knobs_of(p.knobs).first_dbg;
        if (dbg > 0 && dbg < 8 && kFirstWDbg[dbg] && p.first_wino && nst == 3 && pmode == 3 && geo == 22) k = kFirstWDbg[dbg];
    }
    HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(kWaves * 64), p.lds_bytes, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) TH_FAIL(TH_EHIP, "conv_first launch failed: %s (%s)", hipGetErrorString(e), p.label.c_str());
    return TH_OK;
}
