// Fused Conv3D for gfx950: implicit GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32),
// with the Keras block around it folded in:
//     [BN-affine -> act on the input]  ->  Conv3D(stride 1, any k, 'same'/'valid')  -> +bias
//     -> [act / BN-affine ...]  ->  [Max/AveragePooling3D 2x2x2]  -> store at a channel offset
// i.e. one launch per "Conv3D -> ELU -> BatchNorm (-> MaxPool)" block of TIMED (reference
// README.md:254) and per "BN -> ReLU -> Conv -> (Concat)" step of a DenseCPD dense layer.
//
// GEMM view (per workgroup):  M = output voxels of a brick (FB whole frames, or ZB z-planes of one
// frame), N = BN output channels, K = taps x Cin.  The brick's *input* voxels (+halo, zero padded =
// Keras 'same') are staged ONCE per Cin-chunk in LDS as [voxel][CS] fp32; every tap then reads the
// same LDS image at a wave-uniform voxel offset, so HBM/L2 sees each activation once per chunk
// instead of k^3 times.  Weights are pre-packed on the host as [nb][chunk][tap][BN][CS] so that a
// tap slab is one linear, coalesced copy into LDS (double buffered across taps, or fully resident
// when small).  Both MFMA operands are fetched with ds_read_b128:
//     lane l = (j = l&31, h = l>>5) reads 4 consecutive input channels  c = 8*kk + 4*h + t
//     A[i=j][k=h] = act[voxel(row j)][c],  B[k=h][j] = W[co j][c]   for MFMA t = 0..3
// (the K order inside a chunk is permuted, identically for A and B — a sum does not care).
// fp32 MFMA is an exact fmaf chain (cdna_hip_programming.md §3), so results match a scalar
// fp32 convolution to accumulation-order rounding.
//
// Row order inside a 32-row MFMA tile is chosen so the epilogue never leaves registers:
// for pooled layers rows are grouped 8 pool-mates at a time (row = 8*pooled + mate); in the
// 32x32 C layout (row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) a lane then holds mates 0-3 or 4-7 of
// 4 pooled voxels, so a 2x2x2 pool is 3 max ops + one cross-half exchange.
#include "common.h"
#include "device_math.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr size_t kLdsLimit = 160 * 1024;
#ifndef STAGE_U
#define STAGE_U 8  // staging loads in flight per thread
#endif
// -DCONV_PROF=1: per-wave s_memtime phase totals of k_conv_mfma printed by workgroup 0 (timing experiments)
#ifndef CONV_PROF
#define CONV_PROF 0
#endif

struct ConvMfmaArgs {
    const float* in; int64_t in_fs; int in_cs, in_coff, Din, Hin, Win, Cin, vec_ok;
    int kd, kh, kw, pz, py, px, ntaps;
    int Dc, Hc, Wc;
    int FB, ZB, nzb, Zp, Hp, Wp, rows_pf, nrows, n_mtiles;
    int CS, nchunks, nnb, tab_off, zmajor, dbg;  // dbg: timing experiments only (TH_CONV_DBG), results are wrong when set
    const float* wpk;
    int Cout;
    const float* bias;
    PreOp pre;
    PostOps post;
    float* out; int64_t out_fs; int out_cs, out_coff, Ho, Wo;
    int64_t nframes;
    const float* wx;   // k_conv_n16 XC > 0: weights of output channels 16..16+XC-1, [chunk][tap][q][c][4]
    int sd, sh, sw;    // convolution stride (k_conv_mfma only; 1 elsewhere)
    unsigned in_grp_bytes;   // k_conv_n16: bytes from a frame group's first input element to its last (buffer descriptor)
    int geo_compact;   // k_conv_mfma GEO kernels: chunks after the first re-stage real voxels only (see the staging loop)
};

// GEO > 0 (streamed 16-channel variants, 3x3x3, stride 1): Hp = Wp = GEO at compile time; see the tap loop.
template <int WAVES, int TM, int TN, int NT, int CI, int BRES, int POOL, int GEO = 0>
__global__ void __launch_bounds__(WAVES * 64, WAVES == 4 ? 2 : 1) k_conv_mfma(const ConvMfmaArgs a) {
    constexpr int NTHREADS = WAVES * 64;
    constexpr int BN = NT * 32;
    constexpr int CI4 = CI / 4;
    constexpr int KK = CI / 8;
    static_assert(NT % TN == 0, "NT must be a multiple of TN");
    extern __shared__ __attribute__((aligned(16))) float4 smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int CS4 = a.CS >> 2;
#if CONV_PROF
    const long long cp_t0 = clock64();
    long long cp_tab = 0, cp_stage = 0, cp_mfma = 0, cp_epi = 0;
#endif

    // ---- workgroup -> (frame group, z brick, channel block) ---------------------------------
    int bid = blockIdx.x;
    const int nb = bid % a.nnb; bid /= a.nnb;
    const int zb = bid % a.nzb; bid /= a.nzb;
    const int64_t f0 = (int64_t)bid * a.FB;
    const int z0 = zb * a.ZB;

    // ---- LDS carve-up ---------------------------------------------------------------------------
    // GEO kernels: 3x3x3, stride 1, Hp = Wp = GEO and Hc = Wc = GEO - 2, so the table arithmetic divides by constants
    const int gHp = GEO ? GEO : a.Hp, gWp = GEO ? GEO : a.Wp, gHc = GEO ? GEO - 2 : a.Hc, gWc = GEO ? GEO - 2 : a.Wc;
    const int nvox = a.FB * a.Zp * gHp * gWp;
    float4* A4 = smem;
    float4* B4 = A4 + (size_t)nvox * CS4;
    const int bslab4 = BN * CS4;  // one tap's weights
    int* rowvox = (int*)(reinterpret_cast<char*>(smem) + a.tab_off);
    int* rowout = rowvox + a.nrows;
    int* tapoff = rowout + (POOL ? a.nrows / 8 : a.nrows);  // staged-voxel offset of every tap
    for (int t = tid; t < a.ntaps; t += NTHREADS) {
        const int dz = t / (a.kh * a.kw), r2 = t - dz * (a.kh * a.kw), dy = r2 / a.kw, dx = r2 - dy * a.kw;
        tapoff[t] = (dz * gHp + dy) * gWp + dx;
    }
    int* voxsrc = tapoff + a.ntaps;  // staged voxel -> source offset (floats, relative to frame f0) or -1
    for (int v = tid; v < nvox; v += NTHREADS) {
        const int xl = v % gWp; int t = v / gWp;
        const int yl = t % gHp; t /= gHp;
        const int zl = t % a.Zp; const int f = t / a.Zp;
        const int zi = z0 * a.sd + zl - a.pz, yi = yl - a.py, xi = xl - a.px;   // staged plane zl of the brick = input plane z0*sd + zl - pz
        const bool ok = (f0 + f) < a.nframes && zi >= 0 && zi < a.Din && yi >= 0 && yi < a.Hin && xi >= 0 && xi < a.Win;
        voxsrc[v] = ok ? f * (int)a.in_fs + ((zi * a.Hin + yi) * a.Win + xi) * a.in_cs : -1;
    }

    // per m-tile bitmask over dz: set when EVERY row of the tile reads the zero halo for that dz
    int* tskip = voxsrc + nvox;
    for (int mt = tid; mt < a.n_mtiles; mt += NTHREADS) {
        int mask = 0;
        if (POOL == 0 && a.zmajor) {
            const int ZBv = min(a.ZB, a.Dc - z0);
            const int fhw = a.FB * gHc * gWc, total = ZBv * fhw;
            if (mt * 32 >= total) mask = 0xff;
            else {
                const int zlo = z0 + (mt * 32) / fhw, zhi = z0 + min(mt * 32 + 31, total - 1) / fhw;
                for (int dz = 0; dz < a.kd && dz < 8; ++dz)
                    if (zhi + dz - a.pz < 0 || zlo + dz - a.pz >= a.Din) mask |= 1 << dz;
            }
        }
        tskip[mt] = mask;
    }

    // ---- row tables: GEMM row -> staged voxel index, and -> output offset -----------------------
    {
        const int ZBv = min(a.ZB, a.Dc - z0);
        for (int r = tid; r < a.nrows; r += NTHREADS) {
            int f = r / a.rows_pf, q = r - f * a.rows_pf;
            bool fok = (f0 + f) < a.nframes;
            int vox = 0, oo = -1;
            if (POOL == 0) {
                const int hw = gHc * gWc;
                bool rok = q < ZBv * hw;
                if (a.zmajor) {
                    // rows ordered (z, frame, y, x): a 32-row tile then usually lies inside ONE z-plane, and
                    // tiles of the first/last plane can skip the taps that only ever see the zero halo
                    const int fhw = a.FB * hw;
                    rok = r < ZBv * fhw;
                    const int zl0 = r / fhw, rem0 = r - zl0 * fhw;
                    f = rem0 / hw;
                    q = zl0 * hw + (rem0 - f * hw);
                    fok = (f0 + f) < a.nframes;
                }
                if (fok && rok) {
                    const int zl = q / hw, rem = q - zl * hw, y = rem / gWc, x = rem - y * gWc;
                    vox = ((f * a.Zp + zl * a.sd) * gHp + y * a.sh) * gWp + x * a.sw;   // window origin of output voxel (zl, y, x)
                    oo = f * (int)a.out_fs + (((z0 + zl) * a.Ho + y) * a.Wo + x) * a.out_cs;
                }
                rowout[r] = oo;
            } else {
                const int pq = q >> 3, mate = q & 7;
                const int PH = gHc >> 1, PW = gWc >> 1;
                if (fok && pq < (ZBv >> 1) * PH * PW) {
                    const int pzz = pq / (PH * PW), rem = pq - pzz * (PH * PW), pyy = rem / PW, pxx = rem - pyy * PW;
                    const int zl = 2 * pzz + (mate >> 2), y = 2 * pyy + ((mate >> 1) & 1), x = 2 * pxx + (mate & 1);
                    vox = ((f * a.Zp + zl) * gHp + y) * gWp + x;
                    oo = f * (int)a.out_fs + ((((z0 >> 1) + pzz) * a.Ho + pyy) * a.Wo + pxx) * a.out_cs;
                }
                if (mate == 0) rowout[r >> 3] = oo;
            }
            rowvox[r] = vox;
        }
    }

    const int nblocks_n = NT / TN;
    const int total_blocks = ((a.n_mtiles + TM - 1) / TM) * nblocks_n;
    const int rounds = (total_blocks + WAVES - 1) / WAVES;
    const float* wbase = a.wpk + (size_t)nb * a.nchunks * a.ntaps * BN * a.CS;
    const float4* wbase4 = reinterpret_cast<const float4*>(wbase);

    for (int rd = 0; rd < rounds; ++rd) {
        const int blk = rd * WAVES + wave;
        const bool active = blk < total_blocks;
        const int mb = active ? blk / nblocks_n : 0;
        const int nbw = active ? blk - mb * nblocks_n : 0;

        f32x16 acc[TM][TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[tm][tn][i] = 0.f;

        if (rd == 0) __syncthreads();  // row tables visible
#if CONV_PROF
        if (rd == 0) cp_tab = clock64() - cp_t0;
#endif
        // a wave's TM m-tiles are interleaved (mb, mb + nmb, ...) so that every wave gets its share of the
        // boundary-plane tiles that skip taps
        const int nmb = (a.n_mtiles + TM - 1) / TM;
        int aidx[TM], bidx[TN], skp[TM];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int mt = a.zmajor ? mb + tm * nmb : mb * TM + tm;
            aidx[tm] = (mt < a.n_mtiles ? rowvox[mt * 32 + j] : 0) * CS4 + h;
            skp[tm] = __builtin_amdgcn_readfirstlane((active && mt < a.n_mtiles) ? tskip[mt] : 0xff);
        }
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) bidx[tn] = ((nbw * TN + tn) * 32 + j) * CS4 + h;

        for (int ch = 0; ch < a.nchunks; ++ch) {
            constexpr bool kLdsEpi = (TM * TN > 2);  // the LDS epilogue clobbers the staging area
            const bool need_a = (kLdsEpi || !(a.nchunks == 1 && rd > 0)) && !((a.dbg & 1) && ch > 0);
            const bool need_b = BRES == 2 ? false : (BRES ? need_a : true);
#if CONV_PROF
            const long long cp_a = clock64();
#endif
            __syncthreads();  // everyone is done reading the previous A image / B slabs
            if (need_a) {
                // ---- stage the haloed input brick for channels [ch*CI, ch*CI+CI) ------------------
                // voxsrc[v] (built once per workgroup) holds the source offset of staged voxel v or -1 for
                // halo / out-of-range: no index arithmetic here, and loads go out 4 at a time so their L2
                // latencies overlap instead of serialising.
                constexpr int U = STAGE_U;
                const int nvec = nvox * CI4;
                const float* inb = a.in + f0 * a.in_fs + a.in_coff + ch * CI;
                const bool has_pre = a.pre.scale || a.pre.act != ACT_LINEAR;
                if (GEO > 0 && a.geo_compact) {
                    // Only voxels that exist are staged from memory: real voxel rr of a frame is input voxel rr, and its
                    // staged position follows from constant divisions.  The halo (42 % of a 12^3 image, 64 % of two 7^3
                    // ones) is zeroed with the first chunk — plain LDS writes, no table lookups, no loads — and nothing
                    // writes it again until an LDS-scratch epilogue.  geo_compact (host): 'same' padding, the whole frame in
                    // one brick, Hin = Win = GEO - 2, float4-aligned full chunks.
                    constexpr int N = GEO - 2;
                    const int per_frame = a.Din * N * N;
                    const int nfv = (int)min((int64_t)a.FB, a.nframes - f0);
                    const int nreal4 = nfv * per_frame * CI4;
                    if (ch == 0) {
                        const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
                        for (int v = tid; v < nvox; v += NTHREADS) {
                            const int x = v % GEO, t = v / GEO, y = t % GEO, t2 = t / GEO, z = t2 % a.Zp, f = t2 / a.Zp;
                            if (x == 0 || x == GEO - 1 || y == 0 || y == GEO - 1 || z == 0 || z == a.Zp - 1 || f >= nfv) {
#pragma unroll
                                for (int g = 0; g < CI4; ++g) A4[(size_t)v * CS4 + g] = zero4;
                            }
                        }
                    }
                    for (int base = tid; base < nreal4; base += NTHREADS * U) {
                        float4 val[U];
                        int dst[U];
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            const int i = min(base + u * NTHREADS, nreal4 - 1);
                            const int r = i / CI4, g = i % CI4;
                            const int f = r / per_frame, rr = r - f * per_frame;
                            const int z = rr / (N * N), y = (rr / N) % N, x = rr % N;
                            dst[u] = (((f * a.Zp + z + 1) * GEO + y + 1) * GEO + x + 1) * CS4 + g;
                            val[u] = *reinterpret_cast<const float4*>(inb + f * a.in_fs + rr * a.in_cs + g * 4);
                        }
#pragma unroll
                        for (int u = 0; u < U; ++u) {
                            if (base + u * NTHREADS >= nreal4) continue;
                            if (has_pre) {
                                const int c0 = ch * CI + (dst[u] % CS4) * 4;
                                float e[4] = {val[u].x, val[u].y, val[u].z, val[u].w};
                                if (a.pre.scale) {
#pragma unroll
                                    for (int k = 0; k < 4; ++k) e[k] = fmaf(e[k], a.pre.scale[c0 + k], a.pre.shift[c0 + k]);
                                }
                                th_act_vec<4>(e, a.pre.act, a.pre.alpha);   // decoded once per vector (see device_math.h)
                                val[u] = make_float4(e[0], e[1], e[2], e[3]);
                            }
                            A4[dst[u]] = val[u];
                        }
                    }
                } else
                for (int base = tid; base < nvec; base += NTHREADS * U) {
                    int off[U];
                    float4 val[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int i = base + u * NTHREADS;
                        off[u] = (i < nvec) ? voxsrc[i / CI4] : -1;
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int i = base + u * NTHREADS;
                        const int g = i % CI4;
                        const int c0 = ch * CI + g * 4;
                        val[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (off[u] >= 0 && c0 < a.Cin) {
                            const float* src = inb + off[u] + g * 4;
                            if (a.vec_ok && c0 + 4 <= a.Cin) {
                                val[u] = *reinterpret_cast<const float4*>(src);
                            } else {
                                val[u].x = src[0];
                                if (c0 + 1 < a.Cin) val[u].y = src[1];
                                if (c0 + 2 < a.Cin) val[u].z = src[2];
                                if (c0 + 3 < a.Cin) val[u].w = src[3];
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int i = base + u * NTHREADS;
                        if (i >= nvec) continue;
                        const int v = i / CI4, g = i % CI4;
                        if (has_pre && off[u] >= 0) {
                            const int c0 = ch * CI + g * 4;
                            float e[4] = {val[u].x, val[u].y, val[u].z, val[u].w}, y[4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const int c = min(c0 + k, a.Cin - 1);
                                y[k] = a.pre.scale ? fmaf(e[k], a.pre.scale[c], a.pre.shift[c]) : e[k];
                            }
                            th_act_vec<4>(y, a.pre.act, a.pre.alpha);   // decoded once per vector (see device_math.h)
#pragma unroll
                            for (int k = 0; k < 4; ++k)
                                if (c0 + k < a.Cin) e[k] = y[k];
                            val[u] = make_float4(e[0], e[1], e[2], e[3]);
                        }
                        A4[(size_t)v * CS4 + g] = val[u];
                    }
                }
            }
            const float4* wch4 = wbase4 + (size_t)ch * a.ntaps * bslab4;
            if (need_b) {
                const int n4 = BRES ? a.ntaps * bslab4 : bslab4;
                for (int i = tid; i < n4; i += NTHREADS) B4[i] = wch4[i];
            }
            __syncthreads();
#if CONV_PROF
            const long long cp_b = clock64();
            cp_stage += cp_b - cp_a;
#endif

            if (BRES == 2) {
                // weights streamed L2 -> registers, one tap ahead: every lane fetches exactly its own MFMA
                // B fragments (host layout [nb][chunk][tap][kk][ntile][lane] float4, 1 KiB per wave-load), so
                // there is no weight slab in LDS and NO barrier inside a Cin-chunk — the 8 waves drift apart
                // and cover each other's LDS latency.
                if (active) {
                    constexpr int TAPSTRIDE = KK * NT * 64;
                    const float4* wf = reinterpret_cast<const float4*>(a.wpk) +
                                       ((size_t)(nb * a.nchunks + ch) * a.ntaps) * TAPSTRIDE + (nbw * TN) * 64 + lane;
                    // Software pipeline, pinned with sched_barrier so hipcc cannot sink the loads back next to
                    // their uses.  Per stage (= 8 input channels of one tap): the NEXT stage's A fragments are
                    // requested from LDS before this stage's TM*TN*4 MFMAs and consumed after them (ping-pong
                    // register sets, no copies); a stage's B fragments are re-loaded for the next tap right
                    // after their last use, so they have a whole stage (>= 2048 MFMA cycles) to arrive from L2.
                    // Tap offsets come from scalar counters: no memory access, no wait, on the address path.
                    int tz = 0, ty = 0, tx = 0;  // coordinates of the NEXT tap
                    auto advance = [&]() {       // returns the staged-voxel offset of the next tap (clamped at the end)
                        if (tx + 1 < a.kw) ++tx;
                        else if (ty + 1 < a.kh) { tx = 0; ++ty; }
                        else if (tz + 1 < a.kd) { tx = 0; ty = 0; ++tz; }
                        return ((tz * a.Hp + ty) * a.Wp + tx) * CS4;
                    };
                    if constexpr (GEO > 0 && KK == 2) {
                        // Compile-time geometry: the nine in-plane taps of a dz plane are unrolled and their staged-voxel
                        // offsets are IMMEDIATES of the ds_read_b128 instructions (relative to a per-dz base), the weight
                        // pointer just advances — no scalar tap counters, no v_add per fragment read.  Every non-MFMA
                        // instruction in this loop delays the next MFMA issue by about its own issue time: on the bare
                        // stage structure (tools/microbench/stage_bench.hip) the address arithmetic costs 2.4 % of the
                        // matrix pipe (151.8 -> 155.4 TFLOP/s).
                        constexpr int C5 = (CI + 4) / 4;          // float4 per staged voxel (CS = CI + 4)
                        constexpr int PLANE = GEO * GEO * C5;     // one dz step
                        float4 bc[2][TN], avA[TM], avB[TM];
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                            for (int tn = 0; tn < TN; ++tn) bc[kk][tn] = wf[(kk * NT + tn) * 64];
                        int az[TM];
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm) { az[tm] = aidx[tm]; avA[tm] = A4[az[tm]]; }
                        const float4* wt = wf;   // the current tap's fragments
                        for (int dz = 0; dz < 3; ++dz) {
                            const int nxt_plane = dz < 2 ? PLANE : (2 * GEO + 2) * C5;   // after the last tap: re-read it
#pragma unroll
                            for (int t9 = 0; t9 < 9; ++t9) {
                                const int cur = ((t9 / 3) * GEO + t9 % 3) * C5;
                                const int nxt = (((t9 + 1) / 3) * GEO + (t9 + 1) % 3) * C5;
                                const float4* wn = (t9 == 8 && dz == 2) ? wt : wt + TAPSTRIDE;
                                // ---- stage 0 ----
#pragma unroll
                                for (int tm = 0; tm < TM; ++tm) avB[tm] = A4[az[tm] + cur + 2];
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int tm = 0; tm < TM; ++tm) {
                                    if ((skp[tm] >> dz) & 1) continue;  // whole tile sees only the zero halo for this dz
#pragma unroll
                                    for (int tn = 0; tn < TN; ++tn) {
                                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(avA[tm].x, bc[0][tn].x, acc[tm][tn], 0, 0, 0);
                                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(avA[tm].y, bc[0][tn].y, acc[tm][tn], 0, 0, 0);
                                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(avA[tm].z, bc[0][tn].z, acc[tm][tn], 0, 0, 0);
                                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(avA[tm].w, bc[0][tn].w, acc[tm][tn], 0, 0, 0);
                                    }
                                }
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int tn = 0; tn < TN; ++tn) bc[0][tn] = wn[tn * 64];
                                // ---- stage 1 ----
#pragma unroll
                                for (int tm = 0; tm < TM; ++tm) avA[tm] = A4[az[tm] + (t9 < 8 ? nxt : nxt_plane)];
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int tm = 0; tm < TM; ++tm) {
                                    if ((skp[tm] >> dz) & 1) continue;
#pragma unroll
                                    for (int tn = 0; tn < TN; ++tn) {
                                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(avB[tm].x, bc[1][tn].x, acc[tm][tn], 0, 0, 0);
                                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(avB[tm].y, bc[1][tn].y, acc[tm][tn], 0, 0, 0);
                                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(avB[tm].z, bc[1][tn].z, acc[tm][tn], 0, 0, 0);
                                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(avB[tm].w, bc[1][tn].w, acc[tm][tn], 0, 0, 0);
                                    }
                                }
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int tn = 0; tn < TN; ++tn) bc[1][tn] = wn[(NT + tn) * 64];
                                wt = wn;
                            }
#pragma unroll
                            for (int tm = 0; tm < TM; ++tm) az[tm] += PLANE;
                        }
                    } else if (KK == 2) {
                        float4 bc[2][TN], avA[TM], avB[TM];
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                            for (int tn = 0; tn < TN; ++tn) bc[kk][tn] = wf[(kk * NT + tn) * 64];
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm) avA[tm] = A4[aidx[tm]];
                        int toff_cur = 0;
                        for (int tap = 0; tap < a.ntaps; ++tap) {
                            const int cz = tz;  // dz of the current tap
                            const int toff_nxt = advance();
                            const float4* wn = wf + (size_t)min(tap + 1, a.ntaps - 1) * TAPSTRIDE;
                            // ---- stage 0 ----
#pragma unroll
                            for (int tm = 0; tm < TM; ++tm) avB[tm] = A4[aidx[tm] + toff_cur + 2];
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int tm = 0; tm < TM; ++tm) {
                                if (__builtin_expect((skp[tm] >> cz) & 1, 0)) continue;  // whole tile sees only the zero halo for this dz
#pragma unroll
                                for (int tn = 0; tn < TN; ++tn) {
                                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(avA[tm].x, bc[0][tn].x, acc[tm][tn], 0, 0, 0);
                                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(avA[tm].y, bc[0][tn].y, acc[tm][tn], 0, 0, 0);
                                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(avA[tm].z, bc[0][tn].z, acc[tm][tn], 0, 0, 0);
                                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(avA[tm].w, bc[0][tn].w, acc[tm][tn], 0, 0, 0);
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int tn = 0; tn < TN; ++tn) bc[0][tn] = wn[tn * 64];
                            // ---- stage 1 ----
#pragma unroll
                            for (int tm = 0; tm < TM; ++tm) avA[tm] = A4[aidx[tm] + toff_nxt];
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int tm = 0; tm < TM; ++tm) {
                                if (__builtin_expect((skp[tm] >> cz) & 1, 0)) continue;  // whole tile sees only the zero halo for this dz
#pragma unroll
                                for (int tn = 0; tn < TN; ++tn) {
                                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(avB[tm].x, bc[1][tn].x, acc[tm][tn], 0, 0, 0);
                                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(avB[tm].y, bc[1][tn].y, acc[tm][tn], 0, 0, 0);
                                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(avB[tm].z, bc[1][tn].z, acc[tm][tn], 0, 0, 0);
                                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(avB[tm].w, bc[1][tn].w, acc[tm][tn], 0, 0, 0);
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int tn = 0; tn < TN; ++tn) bc[1][tn] = wn[(NT + tn) * 64];
                            toff_cur = toff_nxt;
                        }
                    } else {
                        float4 bcur[KK][TN], bnxt[KK][TN], av[TM], avn[TM];
#pragma unroll
                        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                            for (int tn = 0; tn < TN; ++tn) bcur[kk][tn] = wf[(kk * NT + tn) * 64];
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm) av[tm] = A4[aidx[tm]];
                        int toff_cur = 0;
                        for (int tap = 0; tap < a.ntaps; ++tap) {
                            const int toff_nxt = advance();
                            const float4* wn = wf + (size_t)min(tap + 1, a.ntaps - 1) * TAPSTRIDE;
#pragma unroll
                            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                                for (int tn = 0; tn < TN; ++tn) bnxt[kk][tn] = wn[(kk * NT + tn) * 64];
#pragma unroll
                            for (int kk = 0; kk < KK; ++kk) {
                                const int noff = (kk + 1 < KK) ? toff_cur + (kk + 1) * 2 : toff_nxt;
#pragma unroll
                                for (int tm = 0; tm < TM; ++tm) avn[tm] = A4[aidx[tm] + noff];
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                                    for (int tn = 0; tn < TN; ++tn) {
                                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].x, bcur[kk][tn].x, acc[tm][tn], 0, 0, 0);
                                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].y, bcur[kk][tn].y, acc[tm][tn], 0, 0, 0);
                                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].z, bcur[kk][tn].z, acc[tm][tn], 0, 0, 0);
                                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].w, bcur[kk][tn].w, acc[tm][tn], 0, 0, 0);
                                    }
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int tm = 0; tm < TM; ++tm) av[tm] = avn[tm];
                            }
#pragma unroll
                            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                                for (int tn = 0; tn < TN; ++tn) bcur[kk][tn] = bnxt[kk][tn];
                            toff_cur = toff_nxt;
                        }
                    }
                }
            } else if (BRES) {
                if (active) {
                    int tap = 0;
                    for (int dz = 0; dz < a.kd; ++dz)
                        for (int dy = 0; dy < a.kh; ++dy)
                            for (int dx = 0; dx < a.kw; ++dx, ++tap) {
                                const int toff = ((dz * a.Hp + dy) * a.Wp + dx) * CS4;
                                const float4* Bt = B4 + (size_t)tap * bslab4;
#pragma unroll
                                for (int kk = 0; kk < KK; ++kk) {
                                    float4 av[TM], bv[TN];
#pragma unroll
                                    for (int tm = 0; tm < TM; ++tm) av[tm] = A4[aidx[tm] + toff + kk * 2];
#pragma unroll
                                    for (int tn = 0; tn < TN; ++tn) bv[tn] = Bt[bidx[tn] + kk * 2];
#pragma unroll
                                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                                        for (int tn = 0; tn < TN; ++tn) {
                                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].x, bv[tn].x, acc[tm][tn], 0, 0, 0);
                                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].y, bv[tn].y, acc[tm][tn], 0, 0, 0);
                                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].z, bv[tn].z, acc[tm][tn], 0, 0, 0);
                                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].w, bv[tn].w, acc[tm][tn], 0, 0, 0);
                                        }
                                }
                            }
                }
            } else {
                // per-tap weight slabs, double buffered: prefetch tap+1 into registers while tap computes
                constexpr int BPF = (BN * ((CI + 4) / 4) + NTHREADS - 1) / NTHREADS;  // float4 per thread per slab
                int tap = 0;
                for (int dz = 0; dz < a.kd; ++dz)
                    for (int dy = 0; dy < a.kh; ++dy)
                        for (int dx = 0; dx < a.kw; ++dx, ++tap) {
                            const bool more = tap + 1 < a.ntaps;
                            // unconditional (clamped) loads keep the prefetch registers out of scratch and
                            // let the loads fly under this tap's MFMAs
                            float4 pf[BPF];
                            {
                                const float4* nxt = wch4 + (size_t)(more ? tap + 1 : tap) * bslab4;
#pragma unroll
                                for (int u = 0; u < BPF; ++u) pf[u] = nxt[min(tid + u * NTHREADS, bslab4 - 1)];
                            }
                            if (active) {
                                const int toff = ((dz * a.Hp + dy) * a.Wp + dx) * CS4;
                                const float4* Bt = B4 + (size_t)(tap & 1) * bslab4;
#pragma unroll
                                for (int kk = 0; kk < KK; ++kk) {
                                    float4 av[TM], bv[TN];
#pragma unroll
                                    for (int tm = 0; tm < TM; ++tm) av[tm] = A4[aidx[tm] + toff + kk * 2];
#pragma unroll
                                    for (int tn = 0; tn < TN; ++tn) bv[tn] = Bt[bidx[tn] + kk * 2];
#pragma unroll
                                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                                        for (int tn = 0; tn < TN; ++tn) {
                                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].x, bv[tn].x, acc[tm][tn], 0, 0, 0);
                                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].y, bv[tn].y, acc[tm][tn], 0, 0, 0);
                                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].z, bv[tn].z, acc[tm][tn], 0, 0, 0);
                                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm].w, bv[tn].w, acc[tm][tn], 0, 0, 0);
                                        }
                                }
                            }
                            if (more) {
                                float4* dstb = B4 + (size_t)((tap + 1) & 1) * bslab4;
#pragma unroll
                                for (int u = 0; u < BPF; ++u) {
                                    const int i = tid + u * NTHREADS;
                                    if (i < bslab4) dstb[i] = pf[u];
                                }
                                __syncthreads();
                            }
                        }
            }
#if CONV_PROF
            cp_mfma += clock64() - cp_b;
#endif
        }  // chunks
#if CONV_PROF
        const long long cp_e = clock64();
#endif

        // ---- epilogue: bias, activation / BN-affine chain, optional 2x2x2 pool, store ------------
        constexpr bool REG_EPI = (TM * TN <= 2);
        if (REG_EPI) {
            // small per-wave tiles: finish in registers.  In the 32x32 C layout a lane holds rows
            // (i&3) + 8*(i>>2) + 4*h, i.e. pool-mates 0-3 (h=0) or 4-7 (h=1) of 4 pooled voxels.
            if (active) {
                float* outb = a.out + f0 * a.out_fs + a.out_coff;
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    const int mt = a.zmajor ? mb + tm * nmb : mb * TM + tm;
                    const bool mt_ok = mt < a.n_mtiles;
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        const int co = nb * BN + (nbw * TN + tn) * 32 + j;
                        const bool cok = co < a.Cout && mt_ok;
                        const int cc = co < a.Cout ? co : 0;
                        const float bv = a.bias ? a.bias[cc] : 0.f;
                        float x[16];
#pragma unroll
                        for (int i = 0; i < 16; ++i) x[i] = acc[tm][tn][i] + bv;
                        th_post16(x, cc, a.post);
                        if (POOL == 0) {
#pragma unroll
                            for (int i = 0; i < 16; ++i) {
                                const int oo = cok ? rowout[mt * 32 + (i & 3) + 8 * (i >> 2) + 4 * h] : -1;
                                if (oo >= 0) outb[oo + co] = x[i];
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
                                if (oo >= 0 && (q >> 1) == h) outb[oo + co] = m;
                            }
                        }
                    }
                }
            }
        } else {
            // big per-wave tiles.  Max-pool in front of a monotone chain (TIMED block 2) needs no scratch at all:
            // a lane holds pool-mates 0-3 (h = 0) or 4-7 (h = 1) of 4 pooled voxels, so 3 max + one cross-half
            // exchange per voxel leave the pooled sums in registers, and the chain runs on 2 values per tile —
            // no LDS round trip, no workgroup barrier in front of the epilogue.
            const bool reg_pool = POOL == 1 && a.post.monotone;
            if (reg_pool) {
                if (active) {
                    float* outb = a.out + f0 * a.out_fs + a.out_coff;
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) {
                        const int mt = a.zmajor ? mb + tm * nmb : mb * TM + tm;
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn) {
                            const int co = nb * BN + (nbw * TN + tn) * 32 + j;
                            const bool cok = co < a.Cout && mt < a.n_mtiles;
                            const int cc = co < a.Cout ? co : 0;
                            const float bv = a.bias ? a.bias[cc] : 0.f;
                            float m[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float mq = fmaxf(fmaxf(acc[tm][tn][4 * q], acc[tm][tn][4 * q + 1]), fmaxf(acc[tm][tn][4 * q + 2], acc[tm][tn][4 * q + 3]));
                                m[q] = fmaxf(mq, __shfl_xor(mq, 32));
                            }
                            // lane half h finishes pooled voxels 2h and 2h + 1 (max commutes with the bias add bit for bit)
                            float v0 = (h ? m[2] : m[0]) + bv, v1 = (h ? m[3] : m[1]) + bv;
                            th_post2(v0, v1, cc, a.post);
                            const int o0 = cok ? rowout[mt * 4 + 2 * h] : -1, o1 = cok ? rowout[mt * 4 + 2 * h + 1] : -1;
                            if (o0 >= 0) outb[o0 + co] = v0;
                            if (o1 >= 0) outb[o1 + co] = v1;
                        }
                    }
                }
            } else {
            // otherwise each 32x32 accumulator tile goes registers -> a per-wave LDS scratch
            // tile -> a compact runtime loop (keeps the activation switch out of a 128-way unroll, so
            // the accumulators stay in VGPRs).  The scratch aliases the A/B staging area (hence the
            // barrier, and hence A is re-staged every round in this mode).
            __syncthreads();
            if (active) {
                float* outb = a.out + f0 * a.out_fs + a.out_coff;
                float* T = reinterpret_cast<float*>(smem) + wave * (32 * 33);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        const int mt = a.zmajor ? mb + tm * nmb : mb * TM + tm;
#pragma unroll
                        for (int i = 0; i < 16; ++i) T[((i & 3) + 8 * (i >> 2) + 4 * h) * 33 + j] = acc[tm][tn][i];
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                        const int co = nb * BN + (nbw * TN + tn) * 32 + j;
                        const bool cok = co < a.Cout && mt < a.n_mtiles;
                        const int cc = co < a.Cout ? co : 0;
                        const float bv = a.bias ? a.bias[cc] : 0.f;
                        if (POOL == 0) {
                            // lane (j, h) finishes rows h, h+2, ..., h+30 of column j
                            float x[16];
#pragma unroll
                            for (int k = 0; k < 16; ++k) x[k] = T[(2 * k + h) * 33 + j] + bv;
                            th_post16(x, cc, a.post);
#pragma unroll
                            for (int k = 0; k < 16; ++k) {
                                const int oo = cok ? rowout[mt * 32 + 2 * k + h] : -1;
                                if (oo >= 0) outb[oo + co] = x[k];
                            }
                        } else {
                            // lane (j, h) finishes pooled voxels q = h and h+2: their 8 mates each
                            float x[16];
#pragma unroll
                            for (int k = 0; k < 2; ++k)
#pragma unroll
                                for (int e = 0; e < 8; ++e) x[8 * k + e] = T[(8 * (2 * k + h) + e) * 33 + j] + bv;
                            if (POOL == 1 && a.post.monotone) {
                                // max-pool first, then the (monotone non-decreasing) chain on the two pooled values only
                                float m0 = x[0], m1 = x[8];
#pragma unroll
                                for (int e = 1; e < 8; ++e) { m0 = fmaxf(m0, x[e]); m1 = fmaxf(m1, x[8 + e]); }
                                th_post2(m0, m1, cc, a.post);
                                const int o0 = cok ? rowout[mt * 4 + h] : -1, o1 = cok ? rowout[mt * 4 + 2 + h] : -1;
                                if (o0 >= 0) outb[o0 + co] = m0;
                                if (o1 >= 0) outb[o1 + co] = m1;
                            } else {
                                th_post16(x, cc, a.post);
#pragma unroll
                                for (int k = 0; k < 2; ++k) {
                                    float m = x[8 * k];
#pragma unroll
                                    for (int e = 1; e < 8; ++e) m = (POOL == 1) ? fmaxf(m, x[8 * k + e]) : m + x[8 * k + e];
                                    if (POOL == 2) m *= 0.125f;
                                    const int oo = cok ? rowout[mt * 4 + 2 * k + h] : -1;
                                    if (oo >= 0) outb[oo + co] = m;
                                }
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                }
            }
            }
        }
#if CONV_PROF
        cp_epi += clock64() - cp_e;
#endif
    }  // rounds
#if CONV_PROF
    if (blockIdx.x == 0 && lane == 0)
        printf("conv_mfma prof wave %d: total %lld  tables %lld  staging+barriers %lld  mfma-phase %lld  epilogue %lld (s_memtime ticks)\n", wave,
               clock64() - cp_t0, cp_tab, cp_stage, cp_mfma, cp_epi);
#endif
}

// =====================================================================================================
// Narrow-output variant: Cout <= 16 (DenseNet/DenseCPD growth convolutions), 3x3x3, stride 1.
// A 32-wide MFMA tile would spend half its columns on zero padding, so this kernel uses
// v_mfma_f32_16x16x4_f32 (same 64 FLOP/clk/SIMD rate): lane (i = l&15, q = l>>4) supplies
// A[i][k=q] / B[k=q][i]; with one ds_read_b128 it holds channels 4q..4q+3 of its voxel, i.e. the
// operands of 4 consecutive MFMAs (K order permuted identically in the packed weights).
// Weight fragments stream L2 -> registers through a 9-tap ring (slot refilled right after its last
// use, 9 taps = several microseconds ahead).  Workgroups are PERSISTENT: the row/voxel tables are
// built once, then the workgroup walks its frame groups; while the MFMAs of one 16-channel chunk run,
// the global loads of the next chunk (or of chunk 0 of the next frame group) are already in flight
// into registers and are written to LDS between two barriers, so staging latency never sits on the
// MFMA pipe even with a single workgroup per CU.  The tap loop is fully unrolled with A registers
// pinned by sched_barrier.
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4n __attribute__((ext_vector_type(4)));

// Compile-time knock-outs for timing experiments on k_conv_n16's inner loop (results are WRONG when set; build a
// second library with -DN16_KNOCK=n and load it through TIMED_HIP_LIB): 1 no weight-ring refills, 2 A fragments read
// from conflict-free dummy addresses, 4 no A-fragment reads after tap 0.
#ifndef N16_KNOCK
#define N16_KNOCK 0
#endif

// XC > 0: output channels 16..16+XC-1 (Cout = 17..20, e.g. the 20 amino-acid classes of a TIMED head) ride on
// v_mfma_f32_4x4x1_16B_f32 from the SAME A registers: the instruction is 16 independent 4x4 outer products (lane
// 4b+i supplies A_b[i] and B_b[i]; VGPR r of lane 4b+j receives D_b[r][j] — checked on hardware,
// tools/microbench/mfma_4x4x1_layout.hip).  With lane (i16, q) holding channels 4q..4q+3 of tile row i16, block
// b = 4q + (i16>>2) multiplies rows 4(i16>>2)..+3 at channel 4q+t by W[4q+t][16 + (i16&3)]: exactly 4 rows x 4 extra
// channels per block, nothing padded, 25 % more matrix-pipe time for 25 % more channels (a 32-wide tile would spend
// 12 of 32 columns on zeros; the round-1 version ran these channels as 64 scalar FMAs per tap per wave on the VALU
// pipe and could not afford the next-chunk prefetch).  The four blocks of a row group hold partial sums over the
// channel quarters q; they are added with two cross-lane exchanges in the epilogue.
// GEO > 0: Hp = Wp = GEO at compile time.  The 27 tap offsets then are immediates of the ds_read_b128 instructions
// instead of a scalar offset + one v_add_u32 per read: measured on the bare tap loop (tools/microbench/
// n16_tap_bench.hip) the address arithmetic alone costs 5-6 % of the matrix pipe (92 -> 98.6 % at TM = 8).
template <int WAVES, int TM, int POOL, int XC = 0, int GEO = 0>
__global__ void __launch_bounds__(WAVES * 64, WAVES == 4 ? 2 : 1) k_conv_n16(const ConvMfmaArgs a) {
    static_assert(XC == 0 || (XC == 4 && TM <= 4 && POOL == 0), "side channels: 4 channels (one 4x4x1 block column), ping-pong variant, no pooling");
    constexpr int NTHREADS = WAVES * 64;
    constexpr int CI = 16, CI4 = 4, NTAPS = 27;
    constexpr int BR = 9;                       // weight-ring depth (taps in flight)
    constexpr int PF = (TM <= 4) ? 6 : 8;     // float4 per thread of next-chunk prefetch (REAL voxels only, see pf_off)
    extern __shared__ __attribute__((aligned(16))) float4 smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, q = lane >> 4;
    constexpr int CS4 = 5;                    // plan_n16: CS = 20 floats per staged voxel

    const int vox_pf = a.Zp * a.Hp * a.Wp;   // staged (haloed) voxels per frame and z slab
    const int nvox = a.FB * vox_pf;
    float4* A4 = smem;
    // The three tables live INSIDE the staged image: a voxel occupies CS = 20 floats of which the fragment reads and the
    // staging writes touch 16; floats 16, 17, 18 of voxel i hold rowvox[i], rowout[i], voxsrc[i] (nrows <= nvox).  No
    // LDS beyond the image itself, so two slabs of a 10^3 frame (2 x 80 640 B) share a CU.
    int* tabs = reinterpret_cast<int*>(smem);
    const int TS = a.CS;
#define ROWVOX(i) tabs[(i) * TS + 16]
#define ROWOUT(i) tabs[(i) * TS + 17]
#define VOXSRC(i) tabs[(i) * TS + 18]

    // z slabs (nzb > 1): a unit of work is (frame group, slab); gridDim.x is a multiple of nzb, so a persistent
    // workgroup keeps its slab and the tables are still built once.  The planner only emits nzb = 1 today: two 4-wave
    // workgroups per CU on 5-plane slabs of a 10^3 frame (2 x 80 640 B) were measured at 124.0 TFLOP/s against 125.2
    // for one 8-wave workgroup on the whole frame — the extra halo planes eat what the overlap gives.
    const int zb = blockIdx.x % a.nzb;
    const int z0 = zb * a.ZB;
    const int ZBv = min(a.ZB, a.Dc - z0);

    // ---- tables, relative to the first frame of a group: built ONCE per (persistent) workgroup -------
    for (int v = tid; v < nvox; v += NTHREADS) {
        const int xl = v % a.Wp; int t = v / a.Wp;
        const int yl = t % a.Hp; t /= a.Hp;
        const int zl = t % a.Zp; const int f = t / a.Zp;
        const int zi = z0 + zl - a.pz, yi = yl - a.py, xi = xl - a.px;
        const bool ok = zi >= 0 && zi < a.Din && yi >= 0 && yi < a.Hin && xi >= 0 && xi < a.Win;
        VOXSRC(v) = ok ? f * (int)a.in_fs + ((zi * a.Hin + yi) * a.Win + xi) * a.in_cs : -1;
    }
    for (int r = tid; r < a.nrows; r += NTHREADS) {
        const int f = r / a.rows_pf, qq = r - f * a.rows_pf;
        int vox = 0, oo = -1;
        if (POOL == 0) {
            const int hw = a.Hc * a.Wc;
            if (qq < ZBv * hw) {
                const int zl = qq / hw, rem = qq - zl * hw, y = rem / a.Wc, x = rem - y * a.Wc;
                vox = ((f * a.Zp + zl) * a.Hp + y) * a.Wp + x;
                oo = f * (int)a.out_fs + (((z0 + zl) * a.Ho + y) * a.Wo + x) * a.out_cs;
            }
            ROWOUT(r) = oo;
        } else {
            const int pq = qq >> 3, mate = qq & 7;
            const int PH = a.Hc >> 1, PW = a.Wc >> 1;
            if (pq < (ZBv >> 1) * PH * PW) {
                const int pzz = pq / (PH * PW), rem = pq - pzz * (PH * PW), pyy = rem / PW, pxx = rem - pyy * PW;
                const int zl = 2 * pzz + (mate >> 2), y = 2 * pyy + ((mate >> 1) & 1), x = 2 * pxx + (mate & 1);
                vox = ((f * a.Zp + zl) * a.Hp + y) * a.Wp + x;
                oo = f * (int)a.out_fs + ((((z0 >> 1) + pzz) * a.Ho + pyy) * a.Wo + pxx) * a.out_cs;
            }
            if (mate == 0) ROWOUT(r >> 3) = oo;
        }
        ROWVOX(r) = vox;
    }
    __syncthreads();

    const int n_mt = a.nrows / 16;
    const int total_blocks = (n_mt + TM - 1) / TM;
    const int rounds = (total_blocks + WAVES - 1) / WAVES;
    const float4* wpk4 = reinterpret_cast<const float4*>(a.wpk) + lane;  // [chunk][tap][lane]
    const int co = i16;
    const bool cvalid = co < a.Cout;
    const int cc = cvalid ? co : 0;
    const float bv = a.bias ? a.bias[cc] : 0.f;
    const int nvec = nvox * CI4;
    const bool has_pre = a.pre.scale || a.pre.act != ACT_LINEAR;
    const int64_t ngroups = (a.nframes + a.FB - 1) / a.FB * a.nzb;   // units: (frame group, slab), slab fastest
    // next-chunk prefetch: when the whole staged image is <= PF float4 per thread, the global loads of the
    // NEXT chunk (or of chunk 0 of this workgroup's next frame group) are issued before the MFMA phase of the
    // current chunk and land in registers underneath it
    // Only voxels that exist in the input are prefetched: the halo of the staged image is written (with zeros) by
    // the first, plain staging pass of the workgroup and never changes afterwards, so re-writing it every chunk
    // (42 % of a 12^3 image, 64 % of a 7^3 one) would only cost load issue slots, registers and LDS writes.
    const int zlo = max(a.pz - z0, 0), ylo = max(a.py, 0), xlo = max(a.px, 0);
    const int nzr = min(a.Zp, a.Din + a.pz - z0) - zlo, nyr = min(a.Hp, a.Hin + a.py) - ylo, nxr = min(a.Wp, a.Win + a.px) - xlo;
    const int real_pf = nzr * nyr * nxr;          // real voxels of one staged frame
    const int nreal4 = a.FB * real_pf * CI4;
    const bool can_pf = rounds == 1 && nreal4 <= PF * NTHREADS && nzr > 0 && nyr > 0 && nxr > 0 && a.vec_ok && (a.Cin & 3) == 0 && !(a.dbg & 64);

    // nvalid: frames of the group that exist (the last group of a batch may be ragged)
    auto load_vec = [&](const float* inb, int nvalid, int ch, int i, bool* okp) -> float4 {
        float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
        int off = (i < nvec) ? VOXSRC(i / CI4) : -1;
        if (nvalid < a.FB && off >= 0 && (i / CI4) / vox_pf >= nvalid) off = -1;
        const int g = i % CI4;
        const int c0 = ch * CI + g * 4;
        *okp = off >= 0;
        if (off >= 0 && c0 < a.Cin) {
            const float* src = inb + ch * CI + off + g * 4;
            if (a.vec_ok && c0 + 4 <= a.Cin) {
                val = *reinterpret_cast<const float4*>(src);
            } else {
                val.x = src[0];
                if (c0 + 1 < a.Cin) val.y = src[1];
                if (c0 + 2 < a.Cin) val.z = src[2];
                if (c0 + 3 < a.Cin) val.w = src[3];
            }
        }
        return val;
    };
    auto store_vec = [&](int ch, int i, float4 val, bool ok) {
        if (i >= nvec) return;
        const int v = i / CI4, g = i % CI4;
        if (has_pre && ok) {
            const int c0 = ch * CI + g * 4;
            float e[4] = {val.x, val.y, val.z, val.w}, y[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c = min(c0 + k, a.Cin - 1);
                y[k] = a.pre.scale ? fmaf(e[k], a.pre.scale[c], a.pre.shift[c]) : e[k];
            }
            th_act_vec<4>(y, a.pre.act, a.pre.alpha);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (c0 + k < a.Cin) e[k] = y[k];
            val = make_float4(e[0], e[1], e[2], e[3]);
        }
        A4[(size_t)v * CS4 + g] = val;
    };

    // weight ring: slot t % BR holds tap t's fragment and is refilled with tap t + BR right after its last
    // use.  The packed image is [chunk][tap][lane] + the first BR taps once more at the end, so the index simply runs
    // on (into the next chunk, and past the last chunk into the copy of chunk 0 for the next frame group) — and the
    // refill address is (uniform chunk base) + (one of 7 loop-invariant per-lane offsets) + (immediate): NO address
    // arithmetic in the tap loop.  On the bare loop 8 scalar + 1 vector instruction per refill cost 3-5 % of the
    // matrix pipe (tools/microbench/n16_tap_bench.hip).
    float4 breg[BR];
    unsigned voff[9];      // lane * 16 + k * 4096 bytes; (t + BR) * 1024 = voff[(t + BR) >> 2] + ((t + BR) & 3) * 1024
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        voff[k] = (unsigned)lane * 16u + (unsigned)k * 4096u;
        asm volatile("" : "+v"(voff[k]));   // keep them as registers: re-deriving them would put the adds back
    }
    // XC: this lane's B operand of the 4x4x1 blocks, xw[t % 3] = W_t[4q..4q+3][16 + (i16 & 3)] (packed image
    // [chunk][tap][q][c][4] + 3 taps of padding); a slot is refilled with tap t + 3 right after use (27 taps = 9 turns
    // of the ring, so slots line up across chunks)
    const f32x4* wx4 = reinterpret_cast<const f32x4*>(a.wx) + q * (XC ? XC : 1) + (XC ? (i16 & 3) : 0);
    unsigned voffx[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        voffx[k] = (unsigned)(q * (XC ? XC : 1) + (XC ? (i16 & 3) : 0)) * 16u + (unsigned)k * 4096u;
        asm volatile("" : "+v"(voffx[k]));
    }
    f32x4 xw[3];
    if (wave < total_blocks) {
#pragma unroll
        for (int t = 0; t < BR; ++t) breg[t] = wpk4[(size_t)t * 64];
        if (XC) {
#pragma unroll
            for (int r = 0; r < 3; ++r) xw[r] = wx4[(size_t)r * 4 * XC];
        }
    }

    // this thread's prefetch slots: BYTE offset (from the frame group's first input element) of the float4 it fetches for
    // real voxel (tid + u * NTHREADS) / 4, or OOB = nothing (the buffer load then returns zeros without touching memory)
    // — and where it goes in the staged image (float4 index).  The loads are buffer loads: a per-group descriptor, this
    // per-slot offset in a VGPR and the chunk's channel offset in an SGPR, i.e. NO address arithmetic and no select per
    // load in front of the MFMA phase (the 64-bit pointer form cost ~6 VALU instructions per load).
    constexpr unsigned OOB = 0xffffffffu;
    unsigned pf_boff[PF];
    int pf_dst[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) {
        const int i = tid + u * NTHREADS;
        pf_boff[u] = OOB; pf_dst[u] = 0;
        if (can_pf && i < nreal4) {
            const int r = i / CI4, f = r / real_pf, rr = r - f * real_pf;
            const int xa = rr % nxr, t = rr / nxr, ya = t % nyr, za = t / nyr;
            const int v = ((f * a.Zp + za + zlo) * a.Hp + ya + ylo) * a.Wp + xa + xlo;
            pf_boff[u] = (unsigned)(VOXSRC(v) + (i % CI4) * 4) * 4u;
            pf_dst[u] = v * CS4 + (i % CI4);
        }
    }
    bool staged = false;   // chunk 0 of the current group was already written to LDS from the prefetch registers
#if N16_KNOCK & 8
    long long prof_t0 = clock64(), prof_mfma = 0, prof_bar = 0, prof_epi = 0, prof_pf = 0, prof_issue = 0, prof_grp = 0;
#endif
    for (int64_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int64_t f0 = grp / a.nzb * a.FB;
        const int nvalid = (int)min((int64_t)a.FB, a.nframes - f0);
        const float* inb0 = a.in + f0 * a.in_fs + a.in_coff;
        float* outb = a.out + f0 * a.out_fs + a.out_coff;
        const int64_t gnext = grp + gridDim.x;   // same slab: gridDim.x % nzb == 0
        const bool has_next = gnext < ngroups;
        const int64_t f0_next = (has_next ? gnext : grp) / a.nzb * a.FB;
        const float* inb_next = a.in + f0_next * a.in_fs + a.in_coff;
        const int nvalid_next = has_next ? (int)min((int64_t)a.FB, a.nframes - f0_next) : nvalid;
        // a ragged group (the last one of a launch) takes the plain staging path, which knows about missing frames
        const bool grp_pf = can_pf && nvalid == a.FB && !(a.dbg & 1);
        const bool nxt_pf = can_pf && has_next && nvalid_next == a.FB && !(a.dbg & 1);
        const __amdgpu_buffer_rsrc_t rs_cur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(inb0), 0, (int)a.in_grp_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_nxt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(inb_next), 0, (int)a.in_grp_bytes, 0x00020000);

#if N16_KNOCK & 8
        long long prof_d = 0;
        const long long prof_g0 = clock64();
        bool prof_first = true;
#endif
        for (int rd = 0; rd < rounds; ++rd) {
            const int blk = rd * WAVES + wave;
            const bool active = blk < total_blocks;
            f32x4 acc[TM];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) acc[tm] = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x4 xacc[XC ? TM : 1];   // 4x4x1 accumulators: VGPR r = row 4*(i16>>2)+r, this lane = extra channel i16&3, channel quarter q
#pragma unroll
            for (int tm = 0; tm < (XC ? TM : 1); ++tm) xacc[tm] = f32x4{0.f, 0.f, 0.f, 0.f};
            int aidx[TM];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int mt = blk * TM + tm;
                aidx[tm] = ((active && mt < n_mt) ? ROWVOX(mt * 16 + i16) : 0) * CS4 + q;
            }
            for (int ch = 0; ch < a.nchunks; ++ch) {
                const bool need_a = !(a.nchunks == 1 && rd > 0) && !((a.dbg & 1) && (ch > 0 || grp != blockIdx.x));
#if N16_KNOCK & 8
                long long prof_a = clock64();
                if (prof_first) { prof_grp += prof_a - prof_g0; prof_first = false; }
#endif
                if (!(ch > 0 ? grp_pf : staged)) {   // otherwise this chunk was written from registers already
                    __syncthreads();
                    if (need_a) {
                        constexpr int U = 4;
                        for (int base = tid; base < nvec; base += NTHREADS * U) {
                            float4 val[U];
                            bool ok[U];
#pragma unroll
                            for (int u = 0; u < U; ++u) val[u] = load_vec(inb0, nvalid, ch, base + u * NTHREADS, &ok[u]);
#pragma unroll
                            for (int u = 0; u < U; ++u) store_vec(ch, base + u * NTHREADS, val[u], ok[u]);
                        }
                    }
                }
                __syncthreads();
#if N16_KNOCK & 8
                const long long prof_a2 = clock64();
                prof_bar += prof_a2 - prof_a;
#endif
                const bool last_ch = ch + 1 == a.nchunks;
                const bool do_pf = last_ch ? nxt_pf : grp_pf;
                const int pch = last_ch ? 0 : ch + 1;
                // The PF loads are issued UNCONDITIONALLY (also when their result will not be used): hipcc can then count
                // them and waits for the weight ring with vmcnt(N) instead of vmcnt(0), which would drain the prefetch in
                // front of the first MFMA.
                float4 pfv[PF];
                {
                    const int g = tid % CI4;                 // NTHREADS % CI4 == 0: the same channel group for every u
                    // nothing is fetched for channels past Cin in the last chunk, nor from a ragged group (frames that do not
                    // exist must not be touched; such a group is staged by the plain path)
                    const bool full = last_ch ? nvalid_next == a.FB : nvalid == a.FB;
                    const unsigned gmask = (full && pch * CI + g * 4 + 4 <= a.Cin) ? 0u : OOB;
                    const unsigned soff = (unsigned)(pch * CI) * 4u;
#pragma unroll
                    for (int u = 0; u < PF; ++u) {
                        const u32x4n raw = last_ch ? __builtin_amdgcn_raw_buffer_load_b128(rs_nxt, pf_boff[u] | gmask, soff, 0)
                                                   : __builtin_amdgcn_raw_buffer_load_b128(rs_cur, pf_boff[u] | gmask, soff, 0);
                        pfv[u] = __builtin_bit_cast(float4, raw);
                    }
                }
                // a thread's prefetched vectors all cover the same 4 channels (NTHREADS % 4 == 0): their BN constants
                // are fetched once, also unconditionally
                const int pc0 = min(pch * CI + (tid % CI4) * 4, a.Cin - 4);
                const bool pbn = a.pre.scale && can_pf;   // can_pf: Cin % 4 == 0, so pc0 is aligned and in bounds
                const float4 psc = *reinterpret_cast<const float4*>(pbn ? a.pre.scale + pc0 : a.wpk);
                const float4 psh = *reinterpret_cast<const float4*>(pbn ? a.pre.shift + pc0 : a.wpk);
#if N16_KNOCK & 8
                long long prof_b = clock64();
                prof_issue += prof_b - prof_a2;
#endif
                const char* wsc = reinterpret_cast<const char*>(a.wpk) + (size_t)ch * (NTAPS * 1024);          // uniform
                const char* wxs = reinterpret_cast<const char*>(a.wx) + (size_t)ch * (NTAPS * 4 * (XC ? XC : 1) * 16);
                if (active) {
                    // A fragments: TM <= 4 keeps two register sets (ping-pong by tap).  TM = 8 has ONE set and pipelines at
                    // half-tap granularity instead: as soon as the MFMAs of tiles 0..3 of tap t have issued, their
                    // registers are re-loaded with tap t+1 while the MFMAs of tiles 4..7 run (and vice versa), so every
                    // ds_read has 16 MFMAs (512 cycles) of cover without a second register set.
                    constexpr bool PING = (TM <= 4);
                    constexpr int HT = PING ? TM : TM / 2;
                    f32x4 av[PING ? 2 : 1][TM];
                    const f32x4* A4v = reinterpret_cast<const f32x4*>(A4);
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) av[0][tm] = A4v[aidx[tm]];
#pragma unroll
                    for (int t = 0; t < NTAPS; ++t) {
                        const int nt = t + 1;
                        int noff = (((nt / 9) * (GEO ? GEO : a.Hp) + (nt / 3) % 3) * (GEO ? GEO : a.Wp) + nt % 3) * CS4;
                        // runtime geometry: opaque to LICM, otherwise hipcc hoists all 27*TM read addresses out of the
                        // chunk loop and spills them to scratch
                        if (!GEO) asm volatile("" : "+s"(noff));
                        if (PING && nt < NTAPS && !(N16_KNOCK & 4)) {
#pragma unroll
                            for (int tm = 0; tm < TM; ++tm) av[nt & 1][tm] = (N16_KNOCK & 2) ? A4v[tid + tm * NTHREADS + (nt & 1) * 64] : A4v[aidx[tm] + noff];
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (!XC) {
#pragma unroll
                            for (int tm = 0; tm < HT; ++tm) {
                                const f32x4 aq = av[PING ? (t & 1) : 0][tm];
                                acc[tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq.x, breg[t % BR].x, acc[tm], 0, 0, 0);
                                acc[tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq.y, breg[t % BR].y, acc[tm], 0, 0, 0);
                                acc[tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq.z, breg[t % BR].z, acc[tm], 0, 0, 0);
                                acc[tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq.w, breg[t % BR].w, acc[tm], 0, 0, 0);
                            }
                        } else {
                            // k-major MFMA order (TM independent tiles between dependent accumulations); the 16-wide MFMA
                            // of (k, tm) is followed by the 4x4x1 MFMA of the same A register for the side channels
                            const float bk[4] = {breg[t % BR].x, breg[t % BR].y, breg[t % BR].z, breg[t % BR].w};
                            const f32x4 wv = xw[t % 3];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
#pragma unroll
                                for (int tm = 0; tm < TM; ++tm) {
                                    const f32x4 aq = av[t & 1][tm];
                                    acc[tm] = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[k], bk[k], acc[tm], 0, 0, 0);
                                    xacc[tm] = __builtin_amdgcn_mfma_f32_4x4x1f32(aq[k], wv[k], xacc[tm], 0, 0, 0);
                                }
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (XC) {
                            const int xb = (t + 3) * 4 * XC * 16;   // bytes from the chunk's first tap
                            xw[t % 3] = *reinterpret_cast<const f32x4*>(wxs + voffx[xb >> 12] + (xb & 4095));
                        }
                        if (!PING) {
                            if (nt < NTAPS && !(N16_KNOCK & 4)) {
#pragma unroll
                                for (int tm = 0; tm < HT; ++tm) av[0][tm] = (N16_KNOCK & 2) ? A4v[tid + tm * NTHREADS + (nt & 1) * 64] : A4v[aidx[tm] + noff];
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
                        if (!(N16_KNOCK & 1))
                            breg[t % BR] = *reinterpret_cast<const float4*>(wsc + voff[(t + BR) >> 2] + ((t + BR) & 3) * 1024);
                        if (!PING && nt < NTAPS && !(N16_KNOCK & 4)) {
#pragma unroll
                            for (int tm = HT; tm < TM; ++tm) av[0][tm] = (N16_KNOCK & 2) ? A4v[tid + tm * NTHREADS + (nt & 1) * 64] : A4v[aidx[tm] + noff];
                        }
                    }
                }
#if N16_KNOCK & 8
                prof_mfma += clock64() - prof_b;
                long long prof_c = clock64();
#endif
                if (do_pf) {
                    // BN -> activation prologue in registers BEFORE the barrier (overlaps the other waves' last MFMAs).  The
                    // activation is decoded ONCE for all 4 PF values (th_act_vec): th_act's switch inlined per element
                    // cost ~10 scalar + vector instructions per value in this serial phase of the chunk.
                    if (has_pre && pch * CI + (tid % CI4) * 4 < a.Cin) {   // (a thread's vectors all cover the same 4 channels)
                        float xv[4 * PF];
#pragma unroll
                        for (int u = 0; u < PF; ++u) {
                            xv[4 * u] = pfv[u].x; xv[4 * u + 1] = pfv[u].y; xv[4 * u + 2] = pfv[u].z; xv[4 * u + 3] = pfv[u].w;
                        }
                        if (a.pre.scale) {
#pragma unroll
                            for (int u = 0; u < PF; ++u) {
                                xv[4 * u] = fmaf(xv[4 * u], psc.x, psh.x); xv[4 * u + 1] = fmaf(xv[4 * u + 1], psc.y, psh.y);
                                xv[4 * u + 2] = fmaf(xv[4 * u + 2], psc.z, psh.z); xv[4 * u + 3] = fmaf(xv[4 * u + 3], psc.w, psh.w);
                            }
                        }
                        th_act_vec<4 * PF>(xv, a.pre.act, a.pre.alpha);
#pragma unroll
                        for (int u = 0; u < PF; ++u) pfv[u] = make_float4(xv[4 * u], xv[4 * u + 1], xv[4 * u + 2], xv[4 * u + 3]);   // slots that fetched nothing are never stored
                    }
                    __syncthreads();   // every wave is done reading chunk ch
#pragma unroll
                    for (int u = 0; u < PF; ++u)
                        if (pf_boff[u] != OOB) A4[pf_dst[u]] = pfv[u];
                }
#if N16_KNOCK & 8
                prof_pf += clock64() - prof_c;
#endif
            }
#if N16_KNOCK & 8
            prof_d = clock64();
#endif
            // ---- epilogue in registers: C layout col = lane&15, row = 4*(lane>>4) + reg --------------------
            if (active) {
#pragma unroll
                for (int g0 = 0; g0 < TM; g0 += 4) {
                    float x[16];
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int r = 0; r < 4; ++r) x[4 * g + r] = (g0 + g < TM) ? acc[(g0 + g < TM) ? g0 + g : 0][r] + bv : 0.f;
                    th_post16(x, cc, a.post);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        if (g0 + g >= TM) continue;
                        const int mt = blk * TM + g0 + g;
                        // rows of frames past the end of a ragged last group are dropped
                        const bool ok = cvalid && mt < n_mt && (nvalid == a.FB || (mt * 16 + 4 * q) / a.rows_pf < nvalid);
                        if (POOL == 0) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int oo = ok ? ROWOUT(mt * 16 + 4 * q + r) : -1;
                                if (oo >= 0) outb[oo + co] = x[4 * g + r];
                            }
                        } else {
                            // rows 4q..4q+3 of the tile = mates (4q)&7.. of pooled voxel q>>1: combine with lane q^1
                            float m;
                            if (POOL == 1) m = fmaxf(fmaxf(x[4 * g], x[4 * g + 1]), fmaxf(x[4 * g + 2], x[4 * g + 3]));
                            else m = (x[4 * g] + x[4 * g + 1]) + (x[4 * g + 2] + x[4 * g + 3]);
                            const float o2 = __shfl_xor(m, 16);
                            m = (POOL == 1) ? fmaxf(m, o2) : (m + o2) * 0.125f;
                            const int oo = ok ? ROWOUT(mt * 2 + (q >> 1)) : -1;
                            if (oo >= 0 && (q & 1) == 0) outb[oo + co] = m;
                        }
                    }
                }
                if (XC) {
                    // side channels: lane (q, g = i16>>2, j = i16&3) holds, in VGPR r, the partial sum of row 4g+r, channel
                    // 16+j over channel quarter q; add the four quarters (lanes 16 apart), then lane q finishes row 4g+q
                    const int cx = 16 + (i16 & 3);
                    const bool xvalid = cx < a.Cout;
                    const float bx = (a.bias && xvalid) ? a.bias[cx] : 0.f;
#pragma unroll
                    for (int tm = 0; tm < (XC ? TM : 1); ++tm) {
                        float mine = 0.f;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float v = xacc[tm][r];
                            v += __shfl_xor(v, 16);
                            v += __shfl_xor(v, 32);
                            mine = (q == r) ? v : mine;
                        }
                        const int mt = blk * TM + tm;
                        const int row = mt * 16 + 4 * (i16 >> 2) + q;
                        const bool ok = xvalid && mt < n_mt && (nvalid == a.FB || row / a.rows_pf < nvalid);
                        const int oo = ok ? ROWOUT(row) : -1;
                        if (oo >= 0) outb[oo + cx] = th_post(mine + bx, cx, a.post);
                    }
                }
            }
        }
#if N16_KNOCK & 8
        prof_epi += clock64() - prof_d;
#endif
        staged = nxt_pf;
    }
#if N16_KNOCK & 8
    if (blockIdx.x == 0 && lane == 0)
        printf("n16 prof wave %d: total %lld  mfma-phase %lld  barrier+stage %lld  prefetch-issue %lld  pf-write %lld  epilogue %lld  group-setup %lld (s_memtime ticks)\n", wave,
               clock64() - prof_t0, prof_mfma, prof_bar, prof_issue, prof_pf, prof_epi, prof_grp);
#endif
}

#undef ROWVOX
#undef ROWOUT
#undef VOXSRC

// ---- tile configurations ------------------------------------------------------------------------
struct CfgDesc { int WAVES, TM, TN, NT, CI, BRES; };
const CfgDesc kCfgs[] = {
    {8, 1, 1, 1, 8, 1},   // 0: Cin<=8,  Cout<=32  (first layer: everything resident, many rounds)
    {8, 4, 2, 2, 16, 0},  // 1: BN=64
    {8, 4, 2, 4, 16, 0},  // 2: BN=128
    {8, 2, 1, 1, 16, 0},  // 3: BN=32
    {8, 2, 2, 2, 8, 1},   // 4: Cin<=8, BN=64
    {8, 4, 2, 2, 16, 2},  // 5: BN=64,  weights streamed L2 -> registers
    {8, 4, 2, 4, 16, 2},  // 6: BN=128, streamed
    {8, 2, 1, 1, 16, 2},  // 7: BN=32,  streamed
    {4, 4, 2, 2, 16, 2},  // 8: 4-wave workgroups (two per CU: one's staging hides under the other's MFMAs)
    {4, 4, 2, 4, 16, 2},  // 9
    {4, 2, 1, 1, 16, 2},  // 10
    {8, 2, 1, 1, 8, 2},   // 11: Cin<=8 with weights streamed (large kernels on the input, e.g. 5^3: 125 taps do not fit LDS)
    {8, 2, 2, 2, 8, 2},   // 12: same, BN=64
    {4, 2, 3, 3, 16, 2},  // 13: BN=96 — the last Cout block of a wide layer (338 = 128 + 128 + 82: 352 columns instead of 384)
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

typedef void (*ConvKernel)(const ConvMfmaArgs);
#define CFG_ROW(W, TM, TN, NT, CI, BR) \
    { k_conv_mfma<W, TM, TN, NT, CI, BR, 0>, k_conv_mfma<W, TM, TN, NT, CI, BR, 1>, k_conv_mfma<W, TM, TN, NT, CI, BR, 2> }
const ConvKernel kKernels[kNumCfgs][3] = {
    CFG_ROW(8, 1, 1, 1, 8, 1), CFG_ROW(8, 4, 2, 2, 16, 0), CFG_ROW(8, 4, 2, 4, 16, 0),
    CFG_ROW(8, 2, 1, 1, 16, 0), CFG_ROW(8, 2, 2, 2, 8, 1),
    CFG_ROW(8, 4, 2, 2, 16, 2), CFG_ROW(8, 4, 2, 4, 16, 2), CFG_ROW(8, 2, 1, 1, 16, 2),
    CFG_ROW(4, 4, 2, 2, 16, 2), CFG_ROW(4, 4, 2, 4, 16, 2), CFG_ROW(4, 2, 1, 1, 16, 2),
    CFG_ROW(8, 2, 1, 1, 8, 2), CFG_ROW(8, 2, 2, 2, 8, 2),
    CFG_ROW(4, 2, 3, 3, 16, 2),
};

// compile-time staged row geometry (Hp = Wp) for the TIMED layers after the first: 3x3x3, stride 1
struct MfmaGeo { int cfg, pool, geo; ConvKernel k; };
const MfmaGeo kMfmaGeo[] = {
    {9, 0, 7, k_conv_mfma<4, 4, 2, 4, 16, 2, 0, 7>},     // 5^3 volumes, two frames per workgroup
    {5, 1, 12, k_conv_mfma<8, 4, 2, 2, 16, 2, 1, 12>},   // 10^3 + 2^3 max-pool
    {13, 0, 7, k_conv_mfma<4, 2, 3, 3, 16, 2, 0, 7>},    // 5^3, 96-column tail block
    {8, 0, 7, k_conv_mfma<4, 4, 2, 2, 16, 2, 0, 7>},     // 5^3, 64-column tail block
};

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

}  // namespace

static bool plan_with_cfg(int cfg, size_t lds_limit, bool whole_frames_only, const TView& in, const TView& oc,
                          const ConvGeom& g, int Cin, int Cout, int pool, ConvMfmaPlan* p) {
    const CfgDesc& c = kCfgs[cfg];
    p->knobs = &th_knobs_planning();
    p->cfg = cfg;
    p->CI = c.CI;
    p->CS = c.CI == 8 ? 8 : c.CI + 4;  // CI=8: unpadded (2-way ds_read_b128 conflict, but the brick fits)
    p->BN = c.NT * 32;
    p->nnb = (Cout + p->BN - 1) / p->BN;
    p->nchunks = (Cin + c.CI - 1) / c.CI;
    p->pool = pool;
    p->bres = c.BRES;
    p->Dc = pool ? (oc.D / 2) * 2 : oc.D;
    p->Hc = pool ? (oc.H / 2) * 2 : oc.H;
    p->Wc = pool ? (oc.W / 2) * 2 : oc.W;
    p->Hp = (p->Hc - 1) * g.sh + g.kh;   // staged (haloed) extent: windows of Hc outputs at stride sh
    p->Wp = (p->Wc - 1) * g.sw + g.kw;
    const int ntaps = g.kd * g.kh * g.kw;
    const size_t bbytes = c.BRES == 2 ? 0 : (size_t)(c.BRES ? ntaps : 2) * p->BN * p->CS * 4;
    auto rows_for = [&](int zb) {
        const int r = pool ? 8 * ((zb / 2) * (p->Hc / 2) * (p->Wc / 2)) : zb * p->Hc * p->Wc;
        return round_up(r, 32);
    };
    auto tab_bytes = [&](int fb, int zb) {
        const size_t nvox = (size_t)fb * ((zb - 1) * g.sd + g.kd) * p->Hp * p->Wp;
        const int nrows = fb * rows_for(zb);
        return (size_t)nrows * 4 + (size_t)(pool ? nrows / 8 : nrows) * 4 + (size_t)ntaps * 4 + nvox * 4 + (size_t)(nrows / 32) * 4;
    };
    auto lds_for = [&](int fb, int zb) {
        const size_t nvox = (size_t)fb * ((zb - 1) * g.sd + g.kd) * p->Hp * p->Wp;
        return std::max(nvox * p->CS * 4 + bbytes, (size_t)c.WAVES * 32 * 33 * 4) + tab_bytes(fb, zb);
    };
    const int max_mt = c.WAVES * c.TM * c.TN / c.NT;  // m-tiles one round covers
    int FB = 0, ZB = 0;
    if (lds_for(1, p->Dc) <= lds_limit && (p->nchunks == 1 || rows_for(p->Dc) / 32 <= max_mt || !whole_frames_only)) {
        ZB = p->Dc;
        FB = 1;
        while (FB < 16 && lds_for(FB + 1, ZB) <= lds_limit &&
               (p->nchunks == 1 || (FB + 1) * rows_for(ZB) / 32 <= max_mt))
            ++FB;
    } else {
        if (whole_frames_only) return false;
        FB = 1;
        const int step = pool ? 2 : 1;
        for (int zb = p->Dc - step; zb >= step; zb -= step)
            if (lds_for(1, zb) <= lds_limit) { ZB = zb; break; }
        if (ZB == 0) return false;
        // prefer a brick height that divides the extent evenly if one exists at >= 60% of the max
        for (int zb = ZB; zb >= step && zb * 10 >= ZB * 6; zb -= step)
            if (p->Dc % zb == 0) { ZB = zb; break; }
    }
    p->FB = FB;
    p->ZB = ZB;
    p->nzb = (p->Dc + ZB - 1) / ZB;
    p->Zp = (ZB - 1) * g.sd + g.kd;
    p->rows_pf = rows_for(ZB);
    p->lds_bytes = lds_for(FB, ZB);
    p->tab_off = p->lds_bytes - tab_bytes(FB, ZB);
    p->wpk_floats = (size_t)p->nnb * p->nchunks * ntaps * p->BN * (c.BRES == 2 ? c.CI : p->CS);
    const double rows_exec = (double)p->nzb * p->rows_pf;  // per frame
    p->exec_flops = 2.0 * rows_exec * (double)(p->nnb * p->BN) * (double)(p->nchunks * c.CI) * ntaps;
    if (p->pool == 0 && c.BRES == 2 && c.CI == 16 && g.kd <= 8 && g.sd == 1 && g.sh == 1 && g.sw == 1) {
        // z-major rows: m-tiles whose rows all read the zero halo for a dz skip that dz's taps
        // (same rule as the kernel's tskip table); exec_flops counts what is actually issued
        const int n_mtiles = FB * p->rows_pf / 32, fhw = FB * p->Hc * p->Wc;
        double issued = 0, full = 0;
        for (int zbk = 0; zbk < p->nzb; ++zbk) {
            const int z0 = zbk * ZB, ZBv = std::min(ZB, p->Dc - z0), total = ZBv * fhw;
            for (int mt = 0; mt < n_mtiles; ++mt) {
                full += g.kd;
                if (mt * 32 >= total) continue;   // whole tile is padding: never issued
                const int zlo = z0 + (mt * 32) / fhw, zhi = z0 + std::min(mt * 32 + 31, total - 1) / fhw;
                for (int dz = 0; dz < g.kd; ++dz)
                    if (!(zhi + dz - g.pz < 0 || zlo + dz - g.pz >= in.D)) issued += 1;
            }
        }
        if (full > 0) p->exec_flops *= issued / full;
    }
    if ((int64_t)FB * oc.fs > 0x7fffffffLL || (int64_t)FB * in.D * in.H * in.W * std::max(in.cs, in.C) > 0x7fffffffLL) return false;
    p->geo = 0;
    if (g.kd == 3 && g.kh == 3 && g.kw == 3 && g.sd == 1 && g.sh == 1 && g.sw == 1 && p->Hp == p->Wp && p->CS == 20 && !th_knobs_planning().conv_nogeo)
        for (const MfmaGeo& ge : kMfmaGeo)
            if (ge.cfg == cfg && ge.pool == pool && ge.geo == p->Hp) p->geo = ge.geo;
    char buf[224], geo[16];
    snprintf(geo, sizeof geo, ",%d", p->geo);   // the full instantiation, as rocprofv3 prints it
    snprintf(buf, sizeof buf, "conv_mfma<w%d,%dx%d,nt%d,ci%d,%s,pool%d> FB%d ZB%d/%d rows%d lds%zuK [k_conv_mfma<%d,%d,%d,%d,%d,%d,%d%s>]",
             c.WAVES, c.TM, c.TN, c.NT, c.CI, c.BRES == 2 ? "stream" : (c.BRES ? "res" : "dbuf"), pool, FB, ZB, p->Dc, p->rows_pf,
             p->lds_bytes / 1024, c.WAVES, c.TM, c.TN, c.NT, c.CI, c.BRES, pool, geo);
    p->label = buf;
    return true;
}


// ---- narrow-output (Cout <= 16) kernel: planning, packing, launch ------------------------------------------
namespace {
struct N16Cfg { int WAVES, TM; };
// [2] = {4,4} with 4 side channels on 4x4x1 MFMA blocks (Cout 17..20)
const N16Cfg kN16[] = {{8, 8}, {4, 4}, {4, 4}};
constexpr int kNumN16 = 3;
typedef void (*ConvKernelN16)(const ConvMfmaArgs);
// compile-time staged row geometry (Hp = Wp) for the shapes the DenseCPD / TIMED topologies use, unpooled
struct N16Geo { int variant, geo; ConvKernelN16 k; };
const N16Geo kN16Geo[] = {
    {0, 12, k_conv_n16<8, 8, 0, 0, 12>},
    {1, 7, k_conv_n16<4, 4, 0, 0, 7>},
    {2, 7, k_conv_n16<4, 4, 0, 4, 7>},
    {1, 4, k_conv_n16<4, 4, 0, 0, 4>},
};
const ConvKernelN16 kN16Kernels[kNumN16][3] = {
    {k_conv_n16<8, 8, 0>, k_conv_n16<8, 8, 1>, k_conv_n16<8, 8, 2>},
    {k_conv_n16<4, 4, 0>, k_conv_n16<4, 4, 1>, k_conv_n16<4, 4, 2>},
    {k_conv_n16<4, 4, 0, 4>, nullptr, nullptr},
};
bool plan_n16(int variant, size_t lds_limit, const TView& in, const TView& oc, const ConvGeom& g, int Cin, int Cout, int pool,
              ConvMfmaPlan* p) {
    const N16Cfg& c = kN16[variant];
    p->knobs = &th_knobs_planning();
    p->cfg = 200 + variant;
    p->CI = 16; p->CS = 20; p->BN = 16; p->nnb = 1; p->nchunks = (Cin + 15) / 16; p->pool = pool; p->bres = 3;
    p->Dc = pool ? (oc.D / 2) * 2 : oc.D;
    p->Hc = pool ? (oc.H / 2) * 2 : oc.H;
    p->Wc = pool ? (oc.W / 2) * 2 : oc.W;
    p->Hp = p->Hc + 2; p->Wp = p->Wc + 2;
    auto rows_for = [&](int zb) { return round_up(pool ? 8 * ((zb / 2) * (p->Hc / 2) * (p->Wc / 2)) : zb * p->Hc * p->Wc, 16); };
    // the row / voxel tables live in the 4 spare floats of every staged voxel (CS = 20): the image is all the LDS there is
    auto lds_for = [&](int fb, int zb) { return (size_t)fb * (zb + 2) * p->Hp * p->Wp * p->CS * 4; };
    const int max_mt = c.WAVES * c.TM;
    int FB = 1;
    const int ZB = p->Dc, nzb = 1;                                   // whole frames only (see the kernel on z slabs)
    if (lds_for(1, p->Dc) > lds_limit) return false;
    if (p->nchunks > 1 && rows_for(p->Dc) / 16 > max_mt) return false;
    while (FB < 32 && lds_for(FB + 1, p->Dc) <= lds_limit && (FB + 1) * rows_for(p->Dc) / 16 <= max_mt) ++FB;
    p->FB = FB; p->ZB = ZB; p->nzb = nzb; p->Zp = ZB + 2;
    p->rows_pf = rows_for(ZB);
    p->lds_bytes = lds_for(FB, ZB);
    p->tab_off = 0;
    if ((size_t)FB * p->rows_pf > (size_t)FB * p->Zp * p->Hp * p->Wp) return false;   // tables ride in the image: rows <= staged voxels
    const int xc = variant == 2 ? 4 : 0;   // VALU side channels
    if (xc && pool) return false;
    p->BN = 16 + xc;
    // both images carry the first taps once more at the end (9 resp. 3: the ring depths), so the kernel's refill index never wraps
    p->wpk_floats = ((size_t)p->nchunks * 27 + 9) * 64 * 4 + (xc ? ((size_t)p->nchunks * 27 + 3) * 4 * xc * 4 : 0);
    p->exec_flops = 2.0 * (double)p->rows_pf * p->nzb * 16.0 * (double)(p->nchunks * 16) * 27;   // MFMA only
    if ((int64_t)FB * oc.fs > 0x7fffffffLL || (int64_t)FB * in.D * in.H * in.W * std::max(in.cs, in.C) > 0x7fffffffLL) return false;
    p->geo = 0;
    if (pool == 0 && p->Hp == p->Wp && !th_knobs_planning().n16_nogeo)
        for (const N16Geo& ge : kN16Geo)
            if (ge.variant == variant && ge.geo == p->Hp) p->geo = ge.geo;
    char buf[224], geo[24];
    snprintf(geo, sizeof geo, "%s,%d", xc ? "" : ",0", p->geo);   // the full instantiation, as rocprofv3 prints it
    snprintf(buf, sizeof buf, "conv_n16<w%d,tm%d,pool%d%s> FB%d ZB%d/%d rows%d lds%zuK (16x16x4 MFMA, weight ring%s) [k_conv_n16<%d,%d,%d%s%s>]",
             c.WAVES, c.TM, pool, xc ? ",xc4" : "", FB, ZB, p->Dc, p->rows_pf, p->lds_bytes / 1024,
             xc ? ", channels 16.. on 4x4x1 MFMA blocks" : "", c.WAVES, c.TM, pool, xc ? ",4" : "", geo);
    p->label = buf;
    return true;
}
}  // namespace

bool conv_mfma_plan(const TView& in, const TView& oc, const ConvGeom& g, int Cin, int Cout, int pool, ConvMfmaPlan* p) {
    if (g.dd != 1 || g.dh != 1 || g.dw != 1) return false;
    const bool strided = g.sd != 1 || g.sh != 1 || g.sw != 1;
    if (strided && pool) return false;   // strided convolutions run on the brick kernel (row table carries the stride), unpooled
    if (pool && (oc.D < 2 || oc.H < 2 || oc.W < 2)) return false;
    int cfg;
    if (Cin <= 8) cfg = Cout <= 32 ? 0 : (Cout <= 64 ? 4 : 2);
    else cfg = Cout <= 32 ? 3 : (Cout <= 64 ? 1 : 2);
    // TH_CONV_BMODE=dbuf keeps the LDS double-buffered weight slabs, =stream8 the 8-wave streamed kernels
    // (A/B comparisons); default: weights streamed L2 -> registers, two 4-wave workgroups per CU when whole
    // frames fit in half the LDS
    const ThKnobs& kn = th_knobs_planning();
    p->knobs = &kn;
    const bool dbuf = kn.conv_bmode == 1, stream8 = kn.conv_bmode == 2, no16 = kn.conv_bmode == 3;
    if (!dbuf && !stream8 && !strided && Cout <= 16 && Cin > 8 && g.kd == 3 && g.kh == 3 && g.kw == 3 && !no16) {
        if (plan_n16(1, kLdsLimit / 2, in, oc, g, Cin, Cout, pool, p)) return true;   // two 4-wave workgroups per CU
        if (plan_n16(0, kLdsLimit, in, oc, g, Cin, Cout, pool, p)) return true;       // one 8-wave workgroup
    }
    // Cout 17..20 (a 20-class head): 16 channels on the matrix pipe + up to 4 on the VALU pipe underneath,
    // instead of a 32-wide tile with 12 zero columns
    if (!dbuf && !stream8 && !strided && Cout > 16 && Cout <= 20 && Cin > 8 && pool == 0 && g.kd == 3 && g.kh == 3 && g.kw == 3 && !no16 &&
        !kn.conv_noxc) {
        if (plan_n16(2, kLdsLimit / 2, in, oc, g, Cin, Cout, pool, p)) return true;
    }
    if (!dbuf && cfg >= 1 && cfg <= 3) {
        if (!stream8 && plan_with_cfg(cfg + 7, kLdsLimit / 2, true, in, oc, g, Cin, Cout, pool, p)) return true;
        cfg += 4;
    }
    if (plan_with_cfg(cfg, kLdsLimit, false, in, oc, g, Cin, Cout, pool, p)) return true;
    // Cin <= 8 with a kernel too large for LDS-resident weights: stream them
    if (cfg == 0 || cfg == 4) return plan_with_cfg(cfg == 0 ? 11 : 12, kLdsLimit, false, in, oc, g, Cin, Cout, pool, p);
    return false;
}

// Heterogeneous Cout blocks: a layer planned on the two-per-CU 128-column kernel (cfg 9) whose LAST block would be mostly
// zero columns gets that block from a narrower instantiation instead — 96 (cfg 13), 64 (cfg 8) or 32 (cfg 10) columns —
// launched right after the 128-column blocks on the same input.  Every block stages the input once either way (a block
// is a workgroup), so unlike a split over different kernels nothing is staged more often than before.  TIMED-rotamer's
// head (256 -> 338 at 5^3, half of that model's time): 128 + 128 + 96 = 352 columns instead of 3 x 128 = 384.
bool conv_mfma_plan_tail(const TView& in, const TView& oc, const ConvGeom& g, int Cin, int Cout, int pool, const ConvMfmaPlan& main,
                         ConvMfmaPlan* tail, int* cout_main) {
    const bool off = th_knobs_planning().conv_notail != 0;     // A/B comparisons and tests
    if (off || main.cfg != 9 || main.nnb < 2) return false;
    const int done = (main.nnb - 1) * main.BN, rest = Cout - done;
    if (rest <= 0 || rest > 96) return false;
    const int cfg = rest <= 32 ? 10 : (rest <= 64 ? 8 : 13);
    if (!plan_with_cfg(cfg, kLdsLimit / 2, true, in, oc, g, Cin, rest, pool, tail)) return false;
    if (tail->nnb != 1) return false;
    *cout_main = done;
    return true;
}

void conv_mfma_pack_weights(const ConvMfmaPlan& p, const ConvGeom& g, int Cin, int Cout, const float* w, float* dst) {
    const int ntaps = g.kd * g.kh * g.kw;
    std::memset(dst, 0, p.wpk_floats * sizeof(float));
    if (p.bres == 3) {  // k_conv_n16: [chunk][tap][q][j][t] with ci = chunk*16 + 4q + t, co = j
        for (int ch = 0; ch < p.nchunks; ++ch)
            for (int t = 0; t < ntaps; ++t)
                for (int q = 0; q < 4; ++q)
                    for (int j = 0; j < 16 && j < Cout; ++j)
                        for (int e = 0; e < 4; ++e) {
                            const int ci = ch * 16 + 4 * q + e;
                            if (ci < Cin) dst[((((size_t)ch * ntaps + t) * 4 + q) * 16 + j) * 4 + e] = w[((size_t)t * Cin + ci) * Cout + j];
                        }
        std::memcpy(dst + (size_t)p.nchunks * ntaps * 256, dst, (size_t)9 * 256 * sizeof(float));   // ring run-on: taps 0..8 again
        // side channels (BN = 16 + xc): [chunk][tap][q][c][e] appended after the MFMA image
        const int xc = p.BN - 16;
        float* wx = dst + ((size_t)p.nchunks * ntaps + 9) * 256;
        for (int ch = 0; ch < p.nchunks; ++ch)
            for (int t = 0; t < ntaps; ++t)
                for (int q = 0; q < 4; ++q)
                    for (int c = 0; c < xc; ++c)
                        for (int e = 0; e < 4; ++e) {
                            const int ci = ch * 16 + 4 * q + e, co = 16 + c;
                            if (ci < Cin && co < Cout) wx[((((size_t)ch * ntaps + t) * 4 + q) * xc + c) * 4 + e] = w[((size_t)t * Cin + ci) * Cout + co];
                        }
        if (xc) std::memcpy(wx + (size_t)p.nchunks * ntaps * 4 * xc * 4, wx, (size_t)3 * 4 * xc * 4 * sizeof(float));
        return;
    }
    if (p.bres == 2) {
        // fragment order: [nb][chunk][tap][kk][ntile][h][j][t]  with  ci = chunk*CI + kk*8 + 4h + t,  co = nb*BN + ntile*32 + j
        const int KK = p.CI / 8, NT = p.BN / 32;
        for (int nb = 0; nb < p.nnb; ++nb)
            for (int ch = 0; ch < p.nchunks; ++ch)
                for (int t = 0; t < ntaps; ++t)
                    for (int kk = 0; kk < KK; ++kk)
                        for (int nt = 0; nt < NT; ++nt) {
                            float* frag = dst + ((((size_t)(nb * p.nchunks + ch) * ntaps + t) * KK + kk) * NT + nt) * 256;
                            for (int h = 0; h < 2; ++h)
                                for (int j = 0; j < 32; ++j) {
                                    const int co = nb * p.BN + nt * 32 + j;
                                    if (co >= Cout) continue;
                                    for (int q = 0; q < 4; ++q) {
                                        const int ci = ch * p.CI + kk * 8 + 4 * h + q;
                                        if (ci < Cin) frag[(h * 32 + j) * 4 + q] = w[((size_t)t * Cin + ci) * Cout + co];
                                    }
                                }
                        }
        return;
    }
    for (int nb = 0; nb < p.nnb; ++nb)
        for (int ch = 0; ch < p.nchunks; ++ch)
            for (int t = 0; t < ntaps; ++t) {
                float* slab = dst + (((size_t)nb * p.nchunks + ch) * ntaps + t) * p.BN * p.CS;
                for (int n = 0; n < p.BN; ++n) {
                    const int co = nb * p.BN + n;
                    if (co >= Cout) break;
                    for (int c = 0; c < p.CI; ++c) {
                        const int ci = ch * p.CI + c;
                        if (ci >= Cin) break;
                        slab[(size_t)n * p.CS + c] = w[((size_t)t * Cin + ci) * Cout + co];
                    }
                }
            }
}

int launch_conv_mfma(hipStream_t s, int64_t n, const ConvMfmaPlan& p, TView in, TView out, ConvGeom g, int Cin, int Cout,
                     const float* wpk, const float* bias, PreOp pre, PostOps post) {
    const bool n16 = p.cfg >= 200 && p.cfg < 200 + kNumN16;
    if (!n16 && (p.cfg < 0 || p.cfg >= kNumCfgs)) TH_FAIL(TH_EINVAL, "conv_mfma: bad plan");
    const CfgDesc c16 = {n16 ? kN16[p.cfg - 200].WAVES : 0, 0, 0, 0, 16, 3};
    const CfgDesc& c = n16 ? c16 : kCfgs[p.cfg];
    ConvMfmaArgs a;
    std::memset(&a, 0, sizeof a);
    a.in = in.p; a.in_fs = in.fs; a.in_cs = in.cs; a.in_coff = in.coff;
    a.Din = in.D; a.Hin = in.H; a.Win = in.W; a.Cin = Cin;
    a.vec_ok = (in.cs % 4 == 0 && in.coff % 4 == 0 && in.fs % 4 == 0 && ((uintptr_t)in.p % 16) == 0) ? 1 : 0;
    a.kd = g.kd; a.kh = g.kh; a.kw = g.kw; a.pz = g.pz; a.py = g.py; a.px = g.px; a.ntaps = g.kd * g.kh * g.kw;
    a.sd = g.sd; a.sh = g.sh; a.sw = g.sw;
    a.Dc = p.Dc; a.Hc = p.Hc; a.Wc = p.Wc;
    a.FB = p.FB; a.ZB = p.ZB; a.nzb = p.nzb; a.Zp = p.Zp; a.Hp = p.Hp; a.Wp = p.Wp;
    a.rows_pf = p.rows_pf; a.nrows = p.FB * p.rows_pf; a.n_mtiles = a.nrows / 32;
    a.CS = p.CS; a.nchunks = p.nchunks; a.nnb = p.nnb;
    a.tab_off = (int)p.tab_off;
    const ThKnobs& kn = th_knobs_of(p.knobs);
    { const bool nozm = kn.conv_nozmajor != 0; a.zmajor = (!nozm && !n16 && p.pool == 0 && p.bres == 2 && c.CI == 16 && g.sd == 1 && g.sh == 1 && g.sw == 1) ? 1 : 0; }
    a.dbg = kn.conv_dbg;
    a.wpk = wpk; a.Cout = Cout; a.bias = bias; a.pre = pre; a.post = post;
    a.wx = n16 ? wpk + ((size_t)p.nchunks * 27 + 9) * 256 : wpk;
    a.in_grp_bytes = (unsigned)std::min<int64_t>(0xfffffff0LL, ((int64_t)(p.FB - 1) * in.fs + ((int64_t)in.D * in.H * in.W - 1) * in.cs + Cin) * 4);
    a.out = out.p; a.out_fs = out.fs; a.out_cs = out.cs; a.out_coff = out.coff; a.Ho = out.H; a.Wo = out.W;
    a.nframes = n;
    const int64_t groups = (n + p.FB - 1) / p.FB;
    int64_t grid = groups * p.nzb * p.nnb;
    if (n16) {  // persistent workgroups: one resident set, each walks groups blockIdx.x, +gridDim.x, ...
        static int ncu = 0;
        if (!ncu) {
            int dev = 0;
            HIP_TRY(hipGetDevice(&dev));
            HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
        }
        int64_t resident = (int64_t)ncu * (c.WAVES == 4 ? 2 : 1);
        if (kn.n16_resident) resident = std::max(1, kn.n16_resident);   // tests: force multi-trip workgroups
        // units = (frame group, slab); equal trip counts: ceil(units / ceil(units / resident)) workgroups, and a
        // workgroup keeps its slab: the stride is a multiple of nzb
        const int64_t units = groups * p.nzb;
        const int64_t trips = (units + resident - 1) / resident;
        grid = std::max<int64_t>(1, (units + trips - 1) / std::max<int64_t>(trips, 1));
        grid = (grid + p.nzb - 1) / p.nzb * p.nzb;
    }
    if (grid > 0x7fffffffLL) TH_FAIL(TH_EINVAL, "conv_mfma: grid too large");
    ConvKernel k = n16 ? kN16Kernels[p.cfg - 200][p.pool] : kKernels[p.cfg][p.pool];
    if (!n16 && p.geo)
        for (const MfmaGeo& ge : kMfmaGeo)
            if (ge.cfg == p.cfg && ge.pool == p.pool && ge.geo == p.geo) {
                k = ge.k;
                a.geo_compact = (g.pz == 1 && g.py == 1 && g.px == 1 && p.nzb == 1 && p.Zp == in.D + 2 && in.H == ge.geo - 2 && in.W == ge.geo - 2 &&
                                 a.vec_ok && Cin % 16 == 0 && !kn.conv_nocompact) ? 1 : 0;
            }
    if (n16 && p.geo)
        for (const N16Geo& ge : kN16Geo)
            if (ge.variant == p.cfg - 200 && ge.geo == p.geo) k = ge.k;
    HIP_TRY(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsLimit));
    const size_t lds_pad = (size_t)std::max(0, kn.conv_ldspad);  // occupancy experiments
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(c.WAVES * 64), std::min(p.lds_bytes + lds_pad, kLdsLimit), s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) TH_FAIL(TH_EHIP, "conv_mfma launch failed: %s (%s)", hipGetErrorString(e), p.label.c_str());
    return TH_OK;
}
