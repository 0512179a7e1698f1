// Kernel templates of conv_pointwise.hip (k_conv_pw, k_conv_pw2) — in a header so that their ~80 instantiations compile as two
// translation units in parallel (conv_pointwise.hip: the plain kernels; conv_pointwise_b.hip: the chunk-blocked-output and
// straight-line-epilogue instantiations of k_conv_pw2).  See conv_pointwise.hip for the design notes.
#pragma once
#include "common.h"
#include "device_math.h"

#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

struct ConvPwArgs {
    const float* in; int64_t in_fs; int in_cs, in_coff, Cin, K8, vec_ok, in_dense;
    int V, H, W;        // conv extent per frame (input voxels), V = D*H*W
    int Vo, Ho, Wo;     // pooled extent (POOL != 0)
    const float* wpk;
    int Cout;
    const float* bias;
    PreOp pre;
    PostOps post;
    float* out; int64_t out_fs; int out_cs, out_coff, out_dense;
    int out_blk;        // POOL == 0 only: chunk-blocked output (TView::blk) — (f, v, co) at f*out_fs + (co >> 2)*V*4 + v*4 + (co & 3)
    unsigned nrows;     // GEMM rows: frames*V, or frames*Vo*8 when pooled
    unsigned ntiles;
    unsigned in_bytes, out_bytes;   // k_conv_pw2: byte spans of the activation views (buffer descriptors, < 4 GiB)
    int dbg;            // k_conv_pw2 timing knock-outs (TH_PW_DBG; results are WRONG when set): 1 no loads, 2 no stores
};

// this lane's input row for a tile: lane j supplies GEMM row tile*32 + j (POOL: pooled voxel tile*4 + j/8, mate j%8);
// rows past the end are clamped to the last one (their results are never stored)
template <int POOL>
__device__ __forceinline__ const float* pw_row_ptr(const ConvPwArgs& a, unsigned tile, int j, int h) {
    const float* src;
    if (POOL == 0) {
        const unsigned r = tile * 32 + j;
        const unsigned rc = r < a.nrows ? r : a.nrows - 1;
        if (a.in_dense) src = a.in + (int64_t)rc * a.in_cs;
        else { const unsigned f = rc / (unsigned)a.V; src = a.in + (int64_t)f * a.in_fs + (int64_t)(rc - f * a.V) * a.in_cs; }
    } else {
        const unsigned p = tile * 4 + (j >> 3), m = j & 7;
        const unsigned pc = p * 8 < a.nrows ? p : a.nrows / 8 - 1;
        const unsigned f = pc / (unsigned)a.Vo, vo = pc - f * a.Vo;
        const unsigned zo = vo / (unsigned)(a.Ho * a.Wo), rem = vo - zo * (a.Ho * a.Wo);
        const unsigned yo = rem / (unsigned)a.Wo, xo = rem - yo * a.Wo;
        const unsigned v = ((2 * zo + (m >> 2)) * a.H + 2 * yo + ((m >> 1) & 1)) * a.W + 2 * xo + (m & 1);
        src = a.in + (int64_t)f * a.in_fs + (int64_t)v * a.in_cs;
    }
    return src + a.in_coff + 4 * h;
}

// epilogue: lane holds output channel (nt*32 + j) of rows (r&3) + 8*(r>>2) + 4h
template <int NT, int POOL>
__device__ __forceinline__ void pw_epilogue(const ConvPwArgs& a, unsigned tile, const f32x16 (&acc)[NT], int j, int h) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = nt * 32 + j;
        const bool cok = co < a.Cout;
        const int cc = cok ? co : 0;
        float x[16];
        const float bv = a.bias ? a.bias[cc] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) x[r] = acc[nt][r] + bv;
        th_post16(x, cc, a.post);
        if (POOL == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const unsigned row = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (cok && row < a.nrows) {
                    int64_t off;
                    if (a.out_blk) {
                        const unsigned f = row / (unsigned)a.V;
                        a.out[(int64_t)f * a.out_fs + (int64_t)(co >> 2) * a.V * 4 + (int64_t)(row - f * a.V) * 4 + (co & 3)] = x[r];
                        continue;
                    }
                    if (a.out_dense) off = (int64_t)row * a.out_cs;
                    else { const unsigned f = row / (unsigned)a.V; off = (int64_t)f * a.out_fs + (int64_t)(row - f * a.V) * a.out_cs; }
                    a.out[off + a.out_coff + co] = x[r];
                }
            }
        } else {
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                float s;
                if (POOL == 1) s = fmaxf(fmaxf(x[4 * o], x[4 * o + 1]), fmaxf(x[4 * o + 2], x[4 * o + 3]));
                else s = (x[4 * o] + x[4 * o + 1]) + (x[4 * o + 2] + x[4 * o + 3]);
                const float t = __shfl_xor(s, 32, 64);
                s = POOL == 1 ? fmaxf(s, t) : (s + t) * 0.125f;
                const unsigned p = tile * 4 + o;
                if (h == 0 && cok && p * 8 < a.nrows) {
                    int64_t off;
                    if (a.out_dense) off = (int64_t)p * a.out_cs;
                    else { const unsigned f = p / (unsigned)a.Vo; off = (int64_t)f * a.out_fs + (int64_t)(p - f * a.Vo) * a.out_cs; }
                    a.out[off + a.out_coff + co] = s;
                }
            }
        }
    }
}

// KMAX: float4 A-slots a lane keeps per K pass (8 channels each); NT: 32-wide output tiles; POOL 0/1 max/2 avg
template <int KMAX, int NT, int POOL>
__global__ void __launch_bounds__(256, NT == 4 ? 2 : (KMAX == 16 && NT == 2 ? 3 : 4)) k_conv_pw(const ConvPwArgs a) {
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int K8 = a.K8;

    // ---- weights (fragment order [kk][nt][lane] float4) and prologue constants -> LDS ---------------
    float4* Bs = smem;
    float* psc = reinterpret_cast<float*>(smem + (size_t)K8 * NT * 64);
    float* psh = psc + K8 * 8;
    {
        const float4* src = reinterpret_cast<const float4*>(a.wpk);
        for (int i = tid; i < K8 * NT * 64; i += 256) Bs[i] = src[i];
        for (int i = tid; i < K8 * 8; i += 256) {
            psc[i] = (a.pre.scale && i < a.Cin) ? a.pre.scale[i] : 1.f;
            psh[i] = (a.pre.shift && i < a.Cin) ? a.pre.shift[i] : 0.f;
        }
    }
    __syncthreads();
    const bool has_pre = a.pre.scale != nullptr || a.pre.act != ACT_LINEAR;

    const unsigned wstride = gridDim.x * 4;
    for (unsigned tile = blockIdx.x * 4 + wave; tile < a.ntiles; tile += wstride) {
        const float* src = pw_row_ptr<POOL>(a, tile, j, h);

        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;

        for (int k0 = 0; k0 < K8; k0 += KMAX) {
            float4 av[KMAX];
#pragma unroll
            for (int u = 0; u < KMAX; ++u) {
                const int c = (k0 + u) * 8 + 4 * h;   // first of this lane's 4 channels
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k0 + u < K8) {
                    if (a.vec_ok && c + 3 < a.Cin) v = *reinterpret_cast<const float4*>(src + (k0 + u) * 8);
                    else {
                        const float* s = src + (k0 + u) * 8;
                        if (c + 0 < a.Cin) v.x = s[0];
                        if (c + 1 < a.Cin) v.y = s[1];
                        if (c + 2 < a.Cin) v.z = s[2];
                        if (c + 3 < a.Cin) v.w = s[3];
                    }
                }
                av[u] = v;
            }
#pragma unroll
            for (int u = 0; u < KMAX; ++u) {
                if (k0 + u < K8) {
                    float4 v = av[u];
                    if (has_pre) {
                        const float4 sc = *reinterpret_cast<const float4*>(psc + (k0 + u) * 8 + 4 * h);
                        const float4 sh = *reinterpret_cast<const float4*>(psh + (k0 + u) * 8 + 4 * h);
                        float y[4] = {fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w)};
                        th_act_vec<4>(y, a.pre.act, a.pre.alpha);     // the activation decoded once per vector, not per element
                        v = make_float4(y[0], y[1], y[2], y[3]);
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float4 b = Bs[((k0 + u) * NT + nt) * 64 + lane];
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, b.x, acc[nt], 0, 0, 0);
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, b.y, acc[nt], 0, 0, 0);
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, b.z, acc[nt], 0, 0, 0);
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, b.w, acc[nt], 0, 0, 0);
                    }
                }
            }
        }

        pw_epilogue<NT, POOL>(a, tile, acc, j, h);
    }
}

// ---- the same GEMM with buffer addressing and a software pipeline ----------------------------------------------
// k_conv_pw above takes the SUM of its phases: per 4096 frames of 96 -> 64 channels at 10^3 it needs 0.90 ms where the
// loads alone take 0.29, the MFMAs 0.36 and the epilogue 0.20 (knock-outs: 0.70 without the epilogue, 0.60 without the
// loads, 0.36 without both).  Every wave alternates "issue loads, wait" and "MFMAs, stores"; the matrix pipe serves the
// ready waves of a SIMD round-robin, so they finish together, reload together and wait together.  Prefetching the next
// tile inside the wave only helps if hipcc can COUNT the memory operations in flight (gfx9 has one vmcnt for loads and
// stores): one conditional load or store and it waits with vmcnt(0), i.e. for the prefetch and for every store of the
// previous tile.  So here every memory instruction is unconditional — buffer loads / stores whose out-of-range lanes
// carry an offset beyond the descriptor (the hardware returns 0 / drops the store) — the next tile's KMAX loads are
// issued before this tile's MFMAs, and the wait in front of a tile's MFMAs is vmcnt(KMAX + 16 NT), not 0.
// Requirements (launch_conv_pw): 16-byte aligned views, Cin % 8 == 0, one K pass (K8 <= KMAX), views under 4 GiB.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

// BLK: the chunk-blocked output form (TView::blk, POOL == 0 only) as its own instantiation — compiled into the plain kernels it
// cost them registers (k_conv_pw2<12,2,0> spilled, <4,2,0> lost a workgroup per CU: 0.29 -> 0.56 ms for DenseCPD's first bottleneck)
// EPI: 0 the generic epilogue chain; 2 "BN-affine -> ReLU" (every DenseNet / DenseCPD bottleneck) as straight-line code with this
// lane's constants in registers — as its own instantiation: next to the generic chain in one kernel it cost 40 registers and spills
template <int KMAX, int NT, int POOL, int BLK = 0, int EPI = 0>
__global__ void __launch_bounds__(256, (NT == 4 || KMAX == 16 || (BLK && KMAX == 12 && NT == 2)) ? 2 : ((BLK && KMAX == 4 && NT <= 2) ? 4 : 3)) k_conv_pw2(const ConvPwArgs a) {   // KMAX = 12: three per CU
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int K8 = a.K8;
    constexpr unsigned OOB = 0xffffffffu;

    float4* Bs = smem;
    float* psc = reinterpret_cast<float*>(smem + (size_t)K8 * NT * 64);
    float* psh = psc + K8 * 8;
    // chunk-blocked output only: a [32 rows][36] transpose tile per wave behind the prologue constants (launch_conv_pw adds it)
    float* const tsc = psh + K8 * 8 + wave * (32 * 36);
    {
        const float4* src = reinterpret_cast<const float4*>(a.wpk);
        for (int i = tid; i < K8 * NT * 64; i += 256) Bs[i] = src[i];
        for (int i = tid; i < K8 * 8; i += 256) {
            psc[i] = (a.pre.scale && i < a.Cin) ? a.pre.scale[i] : 1.f;
            psh[i] = (a.pre.shift && i < a.Cin) ? a.pre.shift[i] : 0.f;
        }
    }
    __syncthreads();
    // psc / psh hold (1, 0) where there is no BatchNorm, so "affine" is always applicable; what matters is the activation
    const int pre_kind = __builtin_amdgcn_readfirstlane(
        a.pre.act == ACT_RELU ? (a.pre.scale ? 1 : 2) : (a.pre.act == ACT_LINEAR && !a.pre.scale ? 0 : 3));
    const __amdgpu_buffer_rsrc_t rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, (int)a.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rout = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)a.out_bytes, 0x00020000);

    // byte offset of this lane's first float4 of a tile (OOB for rows past the end: they load zeros, are never stored)
    auto row_off = [&](unsigned tile) -> unsigned {
        unsigned e;   // element offset
        if (POOL == 0) {
            const unsigned r = tile * 32 + j;
            if (r >= a.nrows) return OOB;
            if (a.in_dense) e = r * (unsigned)a.in_cs;
            else { const unsigned f = r / (unsigned)a.V; e = f * (unsigned)a.in_fs + (r - f * a.V) * (unsigned)a.in_cs; }
        } else {
            const unsigned p = tile * 4 + (j >> 3), m = j & 7;
            if (p * 8 >= a.nrows) return OOB;
            const unsigned f = p / (unsigned)a.Vo, vo = p - f * a.Vo;
            const unsigned zo = vo / (unsigned)(a.Ho * a.Wo), rem = vo - zo * (a.Ho * a.Wo);
            const unsigned yo = rem / (unsigned)a.Wo, xo = rem - yo * a.Wo;
            const unsigned v = ((2 * zo + (m >> 2)) * a.H + 2 * yo + ((m >> 1) & 1)) * a.W + 2 * xo + (m & 1);
            e = f * (unsigned)a.in_fs + v * (unsigned)a.in_cs;
        }
        return (e + (unsigned)a.in_coff + 4u * h) * 4u;
    };
    auto load_tile = [&](u32x4 (&av)[KMAX], unsigned tile) {
        const unsigned base = row_off(tile);
#pragma unroll
        for (int u = 0; u < KMAX; ++u) {
            const unsigned off = (u < K8 && base != OOB && !(a.dbg & 1)) ? base + (unsigned)u * 32u : OOB;
            av[u] = __builtin_amdgcn_raw_buffer_load_b128(rin, off, 0, 0);
        }
    };
    // this lane's bias values, fetched ONCE: a global load inside the tile loop is the YOUNGEST memory operation of the wave,
    // so waiting for it is s_waitcnt vmcnt(0) — it would drain the next tile's prefetch in front of every epilogue (the
    // compiler cannot hoist it itself: the stores to `out` might alias)
    float bias_r[NT];
    const int npost = __builtin_amdgcn_readfirstlane(a.post.n);
    // the epilogue chains of the DenseNet-style / TIMED-style blocks as straight-line code with this lane's BatchNorm constants in
    // registers (a lane keeps its output channels for all its tiles): 1 = ELU -> BN-affine, 2 = BN-affine -> ReLU, 0 = generic
    // (th_post16: op list decoded and constants fetched from memory for every tile — ~650 cycles per (tile, n-tile) in front of the
    // stores of a tile that is worth 2 000 - 6 000 cycles of MFMAs)
    float esc[NT], esh[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int co = nt * 32 + j;
        bias_r[nt] = a.bias ? a.bias[co < a.Cout ? co : 0] : 0.f;
        esc[nt] = 1.f; esh[nt] = 0.f;
        if (EPI == 2) { esc[nt] = a.post.scale[0][co < a.Cout ? co : 0]; esh[nt] = a.post.shift[0][co < a.Cout ? co : 0]; }
    }
    auto compute_store = [&](const u32x4 (&av)[KMAX], unsigned tile) {
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        // The BN -> activation prologue is decoded ONCE per tile, outside the MFMA loop: one specialised copy of the loop for
        // "affine + ReLU" (every DenseNet/DenseCPD layer), one for "nothing", one generic.  With th_act's switch inlined per
        // element the kernel spent ~60 VALU/SALU instructions per MFMA and ran at 44 % of the matrix pipe with every load and
        // store knocked out (round 3, TH_PW_DBG=3).
        auto gemm = [&](auto kind) {
            constexpr int KIND = decltype(kind)::value;     // 0 none, 1 affine + ReLU, 2 ReLU, 3 generic
#pragma unroll
            for (int u = 0; u < KMAX; ++u) {
                if (u < K8) {
                    const f32x4v vv = __builtin_bit_cast(f32x4v, av[u]);   // whole-vector cast (element-wise bit_cast of a vector lvalue reads .x four times)
                    float4 v = make_float4(vv.x, vv.y, vv.z, vv.w);
                    if constexpr (KIND == 1 || KIND == 3) {
                        const float4 sc = *reinterpret_cast<const float4*>(psc + u * 8 + 4 * h);
                        const float4 sh = *reinterpret_cast<const float4*>(psh + u * 8 + 4 * h);
                        v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
                    }
                    if constexpr (KIND == 1 || KIND == 2) {
                        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
                    }
                    if constexpr (KIND == 3) {
                        v.x = th_act(v.x, a.pre.act, a.pre.alpha); v.y = th_act(v.y, a.pre.act, a.pre.alpha);
                        v.z = th_act(v.z, a.pre.act, a.pre.alpha); v.w = th_act(v.w, a.pre.act, a.pre.alpha);
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const float4 b = Bs[(u * NT + nt) * 64 + lane];
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, b.x, acc[nt], 0, 0, 0);
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, b.y, acc[nt], 0, 0, 0);
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, b.z, acc[nt], 0, 0, 0);
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, b.w, acc[nt], 0, 0, 0);
                    }
                }
            }
        };
        if (pre_kind == 1) gemm(std::integral_constant<int, 1>{});
        else if (pre_kind == 0) gemm(std::integral_constant<int, 0>{});
        else if (pre_kind == 2) gemm(std::integral_constant<int, 2>{});
        else gemm(std::integral_constant<int, 3>{});
        // epilogue: lane holds output channel (nt*32 + j) of rows (r&3) + 8*(r>>2) + 4h; every store is issued, masked by offset
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int co = nt * 32 + j;
            const bool cok = co < a.Cout;
            const int cc = cok ? co : 0;
            float x[16];
            const float bv = bias_r[nt];
#pragma unroll
            for (int r = 0; r < 16; ++r) x[r] = acc[nt][r] + bv;
            if (EPI == 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) x[r] = fmaxf(fmaf(x[r], esc[nt], esh[nt]), 0.f);
            } else if (npost) {
                th_post16(x, cc, a.post);    // (generic chain: its BatchNorm constants are loaded in the loop)
            }
            if (POOL == 0 && BLK) {
                // chunk-blocked output ([C/4][voxel][4]): the accumulator layout has a lane own ONE channel of 16 rows — stored
                // as it is, every instruction would write 16 separate 16-byte pieces.  The 32 x 32 tile goes through a per-wave
                // LDS tile instead (wave-synchronous: no barrier) and comes back with a lane owning one ROW and 16 channels:
                // four 16-byte stores whose 32 lanes cover 32 consecutive voxels of a chunk, 512 contiguous bytes
#pragma unroll
                for (int r = 0; r < 16; ++r) tsc[((r & 3) + 8 * (r >> 2) + 4 * h) * 36 + j] = x[r];
                const unsigned row = tile * 32 + (unsigned)j;
                const unsigned f = row / (unsigned)a.V, v = row - f * (unsigned)a.V;
                const unsigned ebase = f * (unsigned)a.out_fs + v * 4u;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const u32x4 y = *reinterpret_cast<const u32x4*>(tsc + j * 36 + 16 * h + 4 * k);
                    const unsigned chunk = (unsigned)(nt * 8 + h * 4 + k);
                    const unsigned off = (row < a.nrows && (int)(chunk * 4) < a.Cout && !(a.dbg & 2)) ? (ebase + chunk * (unsigned)a.V * 4u) * 4u : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128(y, rout, off, 0, 0);
                }
            } else if (POOL == 0 && !BLK && a.out_dense && tile * 32 + 32 <= a.nrows) {
                // whole tile in range, rows at a constant byte stride: ONE vector offset per (tile, n-tile) and the row
                // displacement as the instruction's scalar offset — no per-store address arithmetic (it was ~8 VALU
                // instructions x 32 stores per tile, a third of this kernel's VALU issue)
                const unsigned rs = (unsigned)a.out_cs * 4u;
                const unsigned vbase = cok ? ((tile * 32u + 4u * h) * (unsigned)a.out_cs + (unsigned)a.out_coff + (unsigned)co) * 4u : OOB;
                const unsigned vb = (a.dbg & 2) ? OOB : vbase;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x[r]), rout, vb, (unsigned)((r & 3) + 8 * (r >> 2)) * rs, 0);
            } else if (POOL == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned row = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    unsigned e;
                    if (a.out_dense) e = row * (unsigned)a.out_cs;
                    else { const unsigned f = row / (unsigned)a.V; e = f * (unsigned)a.out_fs + (row - f * a.V) * (unsigned)a.out_cs; }
                    const unsigned off = (cok && row < a.nrows && !(a.dbg & 2)) ? (e + (unsigned)a.out_coff + (unsigned)co) * 4u : OOB;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x[r]), rout, off, 0, 0);
                }
            } else {
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    float s;
                    if (POOL == 1) s = fmaxf(fmaxf(x[4 * o], x[4 * o + 1]), fmaxf(x[4 * o + 2], x[4 * o + 3]));
                    else s = (x[4 * o] + x[4 * o + 1]) + (x[4 * o + 2] + x[4 * o + 3]);
                    const float t = __shfl_xor(s, 32, 64);
                    s = POOL == 1 ? fmaxf(s, t) : (s + t) * 0.125f;
                    const unsigned p = tile * 4 + o;
                    unsigned e;
                    if (a.out_dense) e = p * (unsigned)a.out_cs;
                    else { const unsigned f = p / (unsigned)a.Vo; e = f * (unsigned)a.out_fs + (p - f * a.Vo) * (unsigned)a.out_cs; }
                    const unsigned off = (h == 0 && cok && p * 8 < a.nrows) ? (e + (unsigned)a.out_coff + (unsigned)co) * 4u : OOB;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, s), rout, off, 0, 0);
                }
            }
        }
    };

    const unsigned wstride = gridDim.x * 4;
    unsigned tile = blockIdx.x * 4 + wave;
    if (tile >= a.ntiles) return;
    u32x4 avA[KMAX], avB[KMAX];
    load_tile(avA, tile);
    while (true) {
        const unsigned t1 = tile + wstride;      // past the end: all offsets out of range, nothing moves
        load_tile(avB, t1);
        compute_store(avA, tile);
        if (t1 >= a.ntiles) break;
        const unsigned t2 = t1 + wstride;
        load_tile(avA, t2);
        compute_store(avB, t1);
        if (t2 >= a.ntiles) break;
        tile = t2;
    }
}

typedef void (*PwKernel)(const ConvPwArgs);

}  // namespace
