// conv_wfused.hip's layer — 3x3x3 'same' stride-1 on a volume with even in-plane extent (TIMED's conv3d_1: 32 -> 64 at 10^3,
// + MaxPool; SURVEY.md §8(a) P2a, the call served is reference predict.py:142) — taken OFF the fp32-input matrix pipe:
// the same F(2,3) x F(2,3) in-plane / direct-z algorithm, but every product runs on v_mfma_f32_32x32x16_bf16 with BOTH operands
// split exactly into three bf16 pieces (x = h + m + l, round to nearest even at each step), six of the nine piece products
// kept (l H, m H, h H, m M, h M, h L), fp32 accumulation — the scheme of conv_wino.hip's k_wino_gemm_b3 and of
// conv_first_b3.hip, error of the size of one fp32 rounding per product.  The fp32 pipe is 1/16 of the bf16 rate on gfx950:
// six bf16 products cost 3/8 of the matrix-pipe cycles of one fp32 product.
//
// What made conv_first_b3 VALU-bound — splitting the data operand in registers for every MFMA — does not happen here: a
// transform-domain value is split ONCE, when the transform writes it to LDS, and then feeds 3 z taps x 64 output columns.
// The price is LDS capacity (6 bytes per value instead of 4), so the loop nest is turned inside out relative to k_conv_wf:
//
//   workgroup (8 waves) = one frame x all (<= 64) output channels, persistent over frames
//     phase = 16 input channels (one k-step of the 32x32x16 MFMA); its raw slice of the WHOLE frame sits in LDS (R, fp32,
//             [4 chunks of 4 channels][z y rows + a zero row][x, even then odd, + a zero slot] x float4, filled by direct
//             global -> LDS loads: the halo is loaded from 16 bytes of zeros)
//       step = one of the 16 transform positions (a, b):
//         V[pos]  [piece 3][record = (z, tile) 256 + zero + dump][2 slots of 8 channels] bf16   24 KB, double-buffered
//         B[pos]  [z tap 3][column tile 2][piece 3][lane 64] x 8 bf16                          18 KB, double-buffered (LDS-DMA)
//         wave w multiplies row tile w (32 records) with all 64 columns: 3 z taps x 2 column tiles x 6 products = 36 MFMAs into
//         two 32x32 accumulators, while the transform of position pos + 1 (4 voxel reads, 3 adds, the split: ~8 VALU per value;
//         8 values per thread) runs in their shadow; then the accumulators are folded into the lane's OUTPUT registers:
//         y[o][p] += AT[o][a] AT[p][b] acc  (0 / +-1: 2.25 adds per accumulator register on average)
//     unit done: the 2 x 2 outputs of every tile are in registers (128 per lane): bias, epilogue chain, optional 2^3 pooling
//     (the tile is the in-plane window, the z mate is the neighbouring register), stores.
//
// One barrier per step, one more per phase (the first position's transform has nothing to hide behind: ~3 % bubble).
#include "common.h"
#include "device_math.h"

#include <algorithm>
#include <cstring>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr size_t kWfsLdsLimit = 160 * 1024;
constexpr int kWfsFrags = 18;                 // B fragments per (phase, position): [dz 3][column tile 2][piece 3]

struct ConvWfsArgs {
    const float* in; int64_t in_fs; int in_cs, in_coff, in_blk;
    int Cin, nkh;
    const uint4* wpk;                 // [pass][phase][pos 16][frag 18][lane 64] x 8 bf16
    const float* zero16;              // 16 bytes of zeros (DMA source of the halo slots)
    int Cout, ncp;
    const float* bias;
    PostOps post;
    float* out; int64_t out_fs; int out_cs, out_coff;
    int64_t nframes;
    unsigned nslots;                  // ceil(nframes / 8) * 8 * ncp logical units
};

__device__ __forceinline__ unsigned wfs_pk(float x, float y) {          // v_cvt_pk_bf16_f32: round to nearest even
    return __builtin_bit_cast(unsigned, __builtin_convertvector((v2f){x, y}, bf16x2));
}

// One LDS-DMA: 64 lanes x 16 bytes from (wave-uniform base + each lane's 32-bit byte offset) to LDS at lds_dst + 16 lane.
// Spelled in asm for the scalar-base form: through the builtin the compiler kept a 64-bit lane address per load, hoisted all
// of them out of the phase loop, spilled them, and re-loaded each from scratch in front of its DMA (a scratch re-load's
// vmcnt(0) also drains every DMA in flight).  hipcc does not count an asm load: the waits are written out (wfs_dma_wait).
template <bool NT>
__device__ __forceinline__ void wfs_glds(const void* base, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void wfs_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
template <class T>
__device__ __forceinline__ unsigned wfs_lds_addr(T* p) { return (unsigned)(uintptr_t)p; }       // LDS byte address of a __shared__ object

// DBG (TH_WF_DBG with the split kernel; results WRONG): 1 no transform, 2 no MFMAs, 4 weights loaded in the first phase only, 8 no fold,
// 16 slice loaded in the first phase only; 32 (results right) no interleave request inside the groups; 512 fragment reads in the first phase only, 1024 no step barrier
template <int D, int H, int W, int POOL, int DBG = 0>
__global__ void __launch_bounds__(512, 1) k_conv_wfs(const ConvWfsArgs a) {
    constexpr int TY = H / 2, TX = W / 2, NT = TY * TX, NR = D * NT, NVOX = D * H * W;
    constexpr int RCOL = W + 1;                       // x slots of a row: even x, odd x, one zero slot
    constexpr int RROWS = D * H + 1;                  // + the zero row
    constexpr int RSL = RROWS * RCOL;                 // float4 slots per 4-channel chunk
    constexpr int RDMA = (4 * RSL + 63) / 64;         // wave-wide 1 KB loads that fill R
    constexpr int RPAD = RDMA * 64;
    constexpr int VREC = 256 + 2;                     // records: 256 rows, the zero record, the dump record
    constexpr int VPIECE = VREC * 2;                  // uint4 per piece
    constexpr int VBUF = 3 * VPIECE;
    constexpr int BBUF = kWfsFrags * 64;
    static_assert(H % 2 == 0 && W % 2 == 0, "in-plane tiles are 2 x 2");
    static_assert(NR <= 256, "one 32-row tile per wave");
    static_assert(POOL == 0 || D % 2 == 0, "z pooling pairs");
    // three separate arrays (static LDS, 156 KB): the waits the compiler places between an LDS-DMA in flight and a later ds_read
    // depend on what it can prove about aliasing, and indices into ONE dynamic array it cannot tell apart
    __shared__ __attribute__((aligned(16))) float4 R4[RPAD];                   // [4][RSL] (+ pad to whole DMA instructions)
    __shared__ __attribute__((aligned(16))) uint4 V4[2 * VBUF];                // [2][3][VREC][2]
    __shared__ __attribute__((aligned(16))) uint4 B4a[BBUF];                   // [18][64]: even steps
    __shared__ __attribute__((aligned(16))) uint4 B4b[BBUF];                   // odd steps (two arrays: a load into one in flight must
                                                                               // not make the compiler drain it before reads of the other)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j32 = lane & 31, hh = lane >> 5;

    // ---- units: (frame, pass of 64 output channels); frames dealt to the 8 XCDs round-robin (as k_conv_wf)
    const unsigned G = gridDim.x, bid = blockIdx.x;
    if (bid >= a.nslots) return;
    const int my_units = (int)((a.nslots - bid + G - 1) / G);
    auto unit_of = [&](int k, int64_t& f, int& pass, bool& ok) {
        const unsigned u = bid + (unsigned)k * G;
        const unsigned xcd = u & 7u, jj = u >> 3;
        pass = (int)(jj % (unsigned)a.ncp);
        const int64_t ff = (int64_t)(jj / (unsigned)a.ncp) * 8 + xcd;
        ok = ff < a.nframes;
        f = ok ? ff : a.nframes - 1;
    };
    const int nphases = my_units * a.nkh;
    // (experiments, results right; ms per 4096 dense frames against 1.330 as shipped) 2048: the second-dispatched half of the workgroup
    // at priority 1: 1.329; 4096 / 8192: interleave quotas of 5 / 12 VALU per MFMA instead of 8: 1.354 / 1.367; 32: no quota: 1.364
    if ((DBG & 2048) && wave >= 4) __builtin_amdgcn_s_setprio(1);

    // ---- zero V (zero / dump records) and R (halo) once
    for (int k = tid; k < 2 * VBUF; k += 512) V4[k] = make_uint4(0u, 0u, 0u, 0u);
    for (int k = tid; k < RPAD; k += 512) R4[k] = make_float4(0.f, 0.f, 0.f, 0.f);          // the halo of R: never loaded, stays zero

    // record numbering: z * NT + tile; with a pool the two planes of a z pair interleave so that rows 2 t, 2 t + 1 — neighbouring
    // accumulator registers of one lane — are the z mates of one pooled voxel
    auto rec_of = [&](int z, int tile) { return POOL ? (z >> 1) * (2 * NT) + 2 * tile + (z & 1) : z * NT + tile; };
    auto zt_of = [&](int r, int& z, int& tile) {
        if (POOL) { const int pr = r >> 1; z = 2 * (pr / NT) + (r & 1); tile = pr % NT; }
        else { z = r / NT; tile = r % NT; }
    };
    auto vslot = [&](int rec, int half) { return rec * 2 + (half ^ ((rec >> 3) & 1)); };

    // ---- transform side: record tid & 255, channels 8 (tid >> 8) .. + 7 of the phase
    const int trec = tid & 255, th = tid >> 8;
    // patch rows y = 2 ty - 1 + i, columns x = 2 tx - 1 + j: rows / columns 1 and 2 always exist and lie a constant apart
    // (RCOL slots; W / 2 slots: x = 2 tx is even, 2 tx + 1 odd), 0 and 3 may be the zero row / zero slot
    int rb0, rb1, rb3, cs0, cs1, cs3, wslot;
    {
        const bool ok = trec < NR;
        int z, tile;
        zt_of(ok ? trec : 0, z, tile);
        const int ty = tile / TX, tx = tile % TX;
        const int hb = th * 2 * RSL;
        rb1 = (z * H + 2 * ty) * RCOL + hb;
        rb0 = ty > 0 ? rb1 - RCOL : D * H * RCOL + hb;
        rb3 = ty < TY - 1 ? rb1 + 2 * RCOL : D * H * RCOL + hb;
        cs1 = tx;
        cs0 = tx > 0 ? W / 2 + tx - 1 : W;
        cs3 = tx < TX - 1 ? tx + 1 : W;
        wslot = ok ? vslot(trec, th) : (257 * 2 + th);
    }
    // ---- A side: row 32 wave + j32; slots of the minus / centre / plus z tap
    int aoff[3];
    {
        const int row = 32 * wave + j32;
        const bool ok = row < NR;
        int z, tile;
        zt_of(ok ? row : 0, z, tile);
        aoff[1] = ok ? vslot(rec_of(z, tile), hh) : 256 * 2 + hh;
        aoff[0] = (ok && z > 0) ? vslot(rec_of(z - 1, tile), hh) : 256 * 2 + hh;
        aoff[2] = (ok && z < D - 1) ? vslot(rec_of(z + 1, tile), hh) : 256 * 2 + hh;
    }

    // plain floats, not 16-wide vectors: a vector add makes a NEW 16-register tuple before the old one dies, and with eight folds in
    // flight the allocator spilled whole tuples
    float y[2][4][16];                                // [column tile][2 o + p][accumulator register]
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) y[n][q][r] = 0.f;

    // ---- loads -------------------------------------------------------------------------------------------------------------
    const float* const in0 = a.in + a.in_coff;
    // slice of phase ph (unit ph / nkh, channels 16 (ph % nkh) ..): 4 chunks x RSL slots, lane-linear in LDS.  Sources are a
    // wave-uniform base plus a 32-bit lane offset, one load at a time (fenced): twelve 64-bit lane addresses computed side by
    // side next to 160 accumulator registers spilled them
    auto issue_R = [&](int ph) __attribute__((always_inline)) {
        if ((DBG & 16) && ph > 0) return;
        if (ph >= nphases) ph = nphases - 1;
        int64_t f; int pass; bool ok;
        unit_of(ph / a.nkh, f, pass, ok);
        const int kh = ph % a.nkh;
        const char* const fb = reinterpret_cast<const char*>(in0 + f * a.in_fs + (a.in_blk ? (int64_t)kh * 4 * NVOX * 4 : (int64_t)kh * 16));
        const unsigned vstride = a.in_blk ? 16u : (unsigned)a.in_cs * 4u;          // bytes between voxels
        const unsigned cstride = a.in_blk ? (unsigned)NVOX * 16u : 16u;            // bytes between 4-channel chunks
        unsigned lv = (unsigned)lane;
        asm volatile("" : "+v"(lv));                       // (opaque: the offsets below are recomputed here, not hoisted out of the loop and spilled)
#pragma unroll
        for (int t = 0; t < (RDMA + 7) / 8; ++t) {
            const int q = wave + 8 * t;
            const unsigned L = (unsigned)(q * 64) + lv;
            const unsigned chunk = L / (unsigned)RSL, s = L - chunk * (unsigned)RSL;
            const unsigned row = s / (unsigned)RCOL, col = s - row * (unsigned)RCOL;
            const unsigned x = col < (unsigned)(W / 2) ? 2u * col : 2u * (col - (unsigned)(W / 2)) + 1u;
            // halo slots (the zero row, the zero slot of every row, the padding behind the last chunk) are never loaded: zeroed
            // once when the kernel starts, they stay zero — their lanes sit the load out (EXEC)
            const bool real = q < RDMA && L < 4u * RSL && row < (unsigned)(D * H) && col < (unsigned)W;
            const unsigned off = chunk * cstride + (row * (unsigned)W + x) * vstride;
            if (real) wfs_glds<false>(fb, off, wfs_lds_addr(R4 + (q < RDMA ? q : 0) * 64));
        }
    };
    // (The slice of phase ph + 1 can only be loaded into R when the last transform of phase ph has read it: one burst of 70 KB
    // under step 15, which the barrier behind that step waits for.  Measured: 0.13 ms of 1.31 per 4096 frames.  Pulling its lines
    // into the L2 half a phase earlier with ordinary loads — one dword per 128-byte line — was 5 % SLOWER than the bare burst, and
    // the nt hint on the burst another 5 %; both removed.)
    // weight fragments of step (ph, pos) into B buffer `buf`: the stream is contiguous in (pass, phase, pos) order
    const unsigned lane16 = (unsigned)lane * 16u;
    auto issue_B = [&](int ph, int pos, int buf) __attribute__((always_inline)) {
        if ((DBG & 4) && ph > 0) return;
        if (ph >= nphases) ph = nphases - 1;
        int64_t f; int pass; bool ok;
        unit_of(ph / a.nkh, f, pass, ok);
        const int kh = ph % a.nkh;
        const char* const base = reinterpret_cast<const char*>(a.wpk) + ((size_t)((pass * a.nkh + kh) * 16 + pos) * kWfsFrags + wave) * 1024;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            // (third round: waves 2..7 load their second fragment again — every wave issues the same three loads, no branch)
            const int q0 = wave + 8 * t, q = q0 < kWfsFrags ? q0 : q0 - 8;
            wfs_glds<false>(base + (q - wave) * 1024, lane16, wfs_lds_addr((buf ? B4b : B4a) + q * 64));
        }
    };

    // ---- transform of position POS into V buffer `buf`: V = sum_i sum_j BT[a][i] BT[b][j] d[i][j], split into three pieces.
    // BT rows of F(2,3): (d0 - d2, d1 + d2, d2 - d1, d1 - d3).  In three parts so that a step can request a column's two voxels
    // BEFORE a group of MFMAs and use them behind it (tr_issue), keep only the column sum (tr_half), and split / store when both
    // columns are in (tr_finish): 8 registers of raw data in flight instead of 16, no LDS round trip the wave sits out
    float4 ta, tb;
    auto tr_issue = [&](auto POS, int c, int col) __attribute__((always_inline)) {
        constexpr int pos = decltype(POS)::value;
        constexpr int pa = pos >> 2, pb = pos & 3;
        constexpr int i1 = pa == 0 ? 0 : 1, i2 = pa == 3 ? 3 : 2;
        constexpr int j1 = pb == 0 ? 0 : 1, j2 = pb == 3 ? 3 : 2;
        const int ra = i1 == 0 ? rb0 : rb1, rbb = i2 == 3 ? rb3 : rb1 + RCOL;
        const int cc = col == 0 ? (j1 == 0 ? cs0 : cs1) : (j2 == 3 ? cs3 : cs1 + W / 2);
        const float4* const Rc = R4 + c * RSL;
        ta = Rc[ra + cc];
        tb = Rc[rbb + cc];
    };
    auto tr_half = [&](auto POS, float (&t)[4]) __attribute__((always_inline)) {
        constexpr int pos = decltype(POS)::value;
        constexpr int pa = pos >> 2;
        constexpr float si1 = pa == 2 ? -1.f : 1.f, si2 = (pa == 0 || pa == 3) ? -1.f : 1.f;
        t[0] = si1 * ta.x + si2 * tb.x; t[1] = si1 * ta.y + si2 * tb.y; t[2] = si1 * ta.z + si2 * tb.z; t[3] = si1 * ta.w + si2 * tb.w;
    };
    auto tr_finish = [&](auto POS, int buf, int c, const float (&t1)[4], const float (&t2)[4]) __attribute__((always_inline)) {
        constexpr int pos = decltype(POS)::value;
        constexpr int pb = pos & 3;
        constexpr float sj1 = pb == 2 ? -1.f : 1.f, sj2 = (pb == 0 || pb == 3) ? -1.f : 1.f;
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = sj1 * t1[k] + sj2 * t2[k];
        unsigned hp[2], mp[2], lp[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const float x0 = v[2 * k], x1 = v[2 * k + 1];
            hp[k] = wfs_pk(x0, x1);
            const float r0 = x0 - __builtin_bit_cast(float, hp[k] << 16), r1 = x1 - __builtin_bit_cast(float, hp[k] & 0xffff0000u);
            mp[k] = wfs_pk(r0, r1);
            const float q0 = r0 - __builtin_bit_cast(float, mp[k] << 16), q1 = r1 - __builtin_bit_cast(float, mp[k] & 0xffff0000u);
            lp[k] = wfs_pk(q0, q1);
        }
        u32x2* const dst = reinterpret_cast<u32x2*>(V4 + buf * VBUF + wslot) + c;
        dst[0] = (u32x2){hp[0], hp[1]};
        dst[2 * VPIECE] = (u32x2){mp[0], mp[1]};
        dst[4 * VPIECE] = (u32x2){lp[0], lp[1]};
    };
    // the whole transform of one position, unpipelined (the first position of a phase has nothing to hide behind)
    auto transform = [&](auto POS, int buf, int c) __attribute__((always_inline)) {
        float t1[4], t2[4];
        tr_issue(POS, c, 0); tr_half(POS, t1);
        tr_issue(POS, c, 1); tr_half(POS, t2);
        tr_finish(POS, buf, c, t1, t2);
    };

    // the two 32x32 accumulators of a step: acc[0] (complete after the fifth group of MFMAs) is folded into y beside the sixth
    // group, acc[1] lives across the step boundary and is folded beside the FIRST group of the next step (which writes acc[0]) —
    // not after the last MFMA, where nothing covered the adds (all eight waves leave a barrier in lockstep)
    f32x16 acc[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
    // y[o][p] += AT[o][a] AT[p][b] acc[n], AT = (1 1 1 0 / 0 1 -1 -1), (a, b) of position FPOS
    auto fold = [&](auto FPOS, int n) __attribute__((always_inline)) {
        constexpr int fpos = decltype(FPOS)::value;
        constexpr int fa = fpos >> 2, fb = fpos & 3;
        constexpr int ao0 = fa <= 2 ? 1 : 0, ao1 = fa == 0 ? 0 : (fa == 1 ? 1 : -1);
        constexpr int bp0 = fb <= 2 ? 1 : 0, bp1 = fb == 0 ? 0 : (fb == 1 ? 1 : -1);
        if (DBG & 8) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { y[n][0][r] += acc[n][r]; asm volatile("" : "+v"(y[n][0][r])); }
            return;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float m = acc[n][r];
            // (the empty asm pins every sum HERE: the adds hang on no chain, and instruction selection otherwise moved all
            // 1152 of a phase behind its last step — with each step's accumulators parked in scratch until then)
            if (ao0 * bp0 == 1) { y[n][0][r] += m; asm volatile("" : "+v"(y[n][0][r])); }
            if (ao0 * bp1 == 1) { y[n][1][r] += m; asm volatile("" : "+v"(y[n][1][r])); }
            if (ao0 * bp1 == -1) { y[n][1][r] -= m; asm volatile("" : "+v"(y[n][1][r])); }
            if (ao1 * bp0 == 1) { y[n][2][r] += m; asm volatile("" : "+v"(y[n][2][r])); }
            if (ao1 * bp0 == -1) { y[n][2][r] -= m; asm volatile("" : "+v"(y[n][2][r])); }
            if (ao1 * bp1 == 1) { y[n][3][r] += m; asm volatile("" : "+v"(y[n][3][r])); }
            if (ao1 * bp1 == -1) { y[n][3][r] -= m; asm volatile("" : "+v"(y[n][3][r])); }
        }
    };

    // ---- one step: 36 MFMAs of position POS out of V / B buffer (POS & 1); in their shadow the fold of the previous position and
    // the transform of the next one
    auto step = [&](auto POS, int ph) __attribute__((always_inline)) {
        constexpr int pos = decltype(POS)::value;
        constexpr int cur = pos & 1;
        using Prev = std::integral_constant<int, (pos + 15) & 15>;
        using Next = std::integral_constant<int, (pos + 1) & 15>;
        constexpr bool tr = pos < 15 && !(DBG & 1);
        issue_B(pos == 15 ? ph + 1 : ph, (pos + 1) & 15, cur ^ 1);
        if (pos == 15) issue_R(ph + 1);
        const uint4* const Vc = V4 + cur * VBUF;
        const uint4* const Bc = (cur ? B4b : B4a) + lane;
        // six groups g = 2 dz + n of six MFMAs; the fragments of group g + 1 are requested before the MFMAs of group g, and the
        // groups are fenced (sched_barrier): left alone hipcc hoists the 27 fragment reads of a step to its top — 108 registers —
        // and spills the output accumulators
        // ONE set of A registers (a second set is 12 registers the step does not have: lane constants went to scratch and came back
        // in every step): the products are ordered l H, m H, m M, h H, h M, h L, so that piece l is free after the first MFMA of a
        // z tap's second group, m after the third, h after the last — each is re-loaded for the next z tap right there
        bf16x8 Af[3], Bf[2][3];
        auto loadA1 = [&](int dz, int pc) __attribute__((always_inline)) {
            if ((DBG & 512) && ph > 0) { asm volatile("" : "+v"(Af[pc])); return; }
            Af[pc] = __builtin_bit_cast(bf16x8, Vc[pc * VPIECE + aoff[dz]]);
        };
        auto loadB = [&](int g, int set) __attribute__((always_inline)) {
            if ((DBG & 512) && ph > 0) { asm volatile("" : "+v"(Bf[set][0]), "+v"(Bf[set][1]), "+v"(Bf[set][2])); return; }
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) Bf[set][pc] = __builtin_bit_cast(bf16x8, Bc[(g * 3 + pc) * 64]);
        };
        float t1[4], t2[4];
        // Everything else a wave does is placed BETWEEN its own MFMAs (sched_group_barrier: "one MFMA, then up to V VALU and L LDS
        // instructions", six times per group): all eight waves leave the step's barrier in lockstep, so what a wave does after its
        // MFMAs the SIMD's other wave does at the same moment, and the matrix pipe idles (round 4's k_conv_wf).
        //   group 0: fold of the previous step's second accumulator (the group overwrites the first)   36 adds
        //   groups 1..4: the next position's transform, a column of a 4-channel half per group          4 / 26 / 4 / 26 VALU
        //   group 5: fold of THIS step's first accumulator (complete after group 4)                     36 adds
#define WFS_PIPE(V, L) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, V, 0); __builtin_amdgcn_sched_group_barrier(0x080, L, 0);
#define WFS_P1 { if (DBG & 4096) { WFS_PIPE(5, 2) } else if (DBG & 8192) { WFS_PIPE(12, 3) } else { WFS_PIPE(8, 2) } }
#define WFS_SLOT { if (!(DBG & 32)) { WFS_P1 WFS_P1 WFS_P1 WFS_P1 WFS_P1 WFS_P1 } __builtin_amdgcn_sched_barrier(0); }
        loadA1(0, 2); loadA1(0, 1); loadA1(0, 0);
        loadB(0, 0);
        if (tr) tr_issue(Next{}, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            const int dz = g >> 1, n = g & 1;
            const bool reload = n == 1 && dz < 2;
            if (g + 1 < 6) loadB(g + 1, (g + 1) & 1);
            if (!(DBG & 2)) {
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const bf16x8 (&B)[3] = Bf[g & 1];
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Af[2], B[0], dz == 0 ? zero : acc[n], 0, 0, 0);      // l H
                if (reload) loadA1(dz + 1, 2);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Af[1], B[0], acc[n], 0, 0, 0);                       // m H
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Af[1], B[1], acc[n], 0, 0, 0);                       // m M
                if (reload) loadA1(dz + 1, 1);
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Af[0], B[0], acc[n], 0, 0, 0);                       // h H
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Af[0], B[1], acc[n], 0, 0, 0);                       // h M
                acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Af[0], B[2], acc[n], 0, 0, 0);                       // h L
                if (reload) loadA1(dz + 1, 0);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[n][r] = (dz == 0 ? 0.f : acc[n][r]) + __builtin_bit_cast(float, __builtin_bit_cast(u32x4, Af[0])[r] ^ __builtin_bit_cast(u32x4, Bf[g & 1][n])[r]);
                if (reload) { loadA1(dz + 1, 2); loadA1(dz + 1, 1); loadA1(dz + 1, 0); }
            }
            if (g == 0) fold(Prev{}, 1);
            if (g == 5) fold(POS, 0);
            if (tr) {
                if (g == 1) { tr_half(Next{}, t1); tr_issue(Next{}, 0, 1); }
                if (g == 2) { tr_half(Next{}, t2); tr_finish(Next{}, cur ^ 1, 0, t1, t2); tr_issue(Next{}, 1, 0); }
                if (g == 3) { tr_half(Next{}, t1); tr_issue(Next{}, 1, 1); }
                if (g == 4) { tr_half(Next{}, t2); tr_finish(Next{}, cur ^ 1, 1, t1, t2); }
            }
            WFS_SLOT
        }
#undef WFS_SLOT
#undef WFS_P1
#undef WFS_PIPE
        wfs_dma_wait();                                   // this wave's pieces of the next step's weights (and slice) have landed
        if (!(DBG & 1024)) __syncthreads();               // V / B of the next step complete
    };

    // ---- unit done: bias, epilogue chain, (pool,) store; accumulators cleared.  C layout of the 32x32 MFMA: column = j32,
    // row = (r & 3) + 8 (r >> 2) + 4 hh
    auto epilogue = [&](int64_t f, int pass, bool uok) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int co = pass * 64 + 32 * n + j32;
            const bool cok = uok && co < a.Cout;
            const int cc = co < a.Cout ? co : 0;
            const float bv = a.bias ? a.bias[cc] : 0.f;
            float* const outb = a.out + f * a.out_fs + a.out_coff + cc;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) y[n][q][r] += bv;
            if (POOL == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = y[n][q][r];
                    th_post16(v, cc, a.post);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        if (!cok || row >= NR) continue;
                        const int z = row / NT, tile = row % NT, ty = tile / TX, tx = tile % TX;
                        outb[(int64_t)((z * H + 2 * ty + (q >> 1)) * W + 2 * tx + (q & 1)) * a.out_cs] = v[r];
                    }
                }
            } else {
                const bool first = POOL == 1 && a.post.monotone;           // pool the raw sums, run the chain on 1/8 of the values
                if (!first) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float v[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) v[r] = y[n][q][r];
                        th_post16(v, cc, a.post);
#pragma unroll
                        for (int r = 0; r < 16; ++r) y[n][q][r] = v[r];
                    }
                }
                float pv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int r = 2 * e;
                    if (POOL == 1) {
                        const float m0 = fmaxf(fmaxf(y[n][0][r], y[n][1][r]), fmaxf(y[n][2][r], y[n][3][r]));
                        const float m1 = fmaxf(fmaxf(y[n][0][r + 1], y[n][1][r + 1]), fmaxf(y[n][2][r + 1], y[n][3][r + 1]));
                        pv[e] = fmaxf(m0, m1);
                    } else {
                        pv[e] = (((y[n][0][r] + y[n][1][r]) + (y[n][2][r] + y[n][3][r])) +
                                 ((y[n][0][r + 1] + y[n][1][r + 1]) + (y[n][2][r + 1] + y[n][3][r + 1]))) * 0.125f;
                    }
                }
                if (first) {
#pragma unroll
                    for (int e = 0; e < 8; e += 2) th_post2(pv[e], pv[e + 1], cc, a.post);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int r = 2 * e;
                    const int row = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (!cok || row >= NR) continue;
                    outb[(int64_t)(row >> 1) * a.out_cs] = pv[e];           // pooled voxel (z pair, tile) = row >> 1
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) y[n][q][r] = 0.f;
        }
    };

    // ---- the persistent loop ---------------------------------------------------------------------------------------------
    __syncthreads();                                      // V zeroed
    issue_R(0);
    issue_B(0, 0, 0);
    wfs_dma_wait();
    for (int ph = 0; ph < nphases; ++ph) {
        __syncthreads();                                  // this phase's slice and its first weights have landed
        if (!(DBG & 1) || ph == 0) {
            transform(std::integral_constant<int, 0>{}, 0, 0);
            transform(std::integral_constant<int, 0>{}, 0, 1);
        }
        __syncthreads();
        step(std::integral_constant<int, 0>{}, ph);
        step(std::integral_constant<int, 1>{}, ph);
        step(std::integral_constant<int, 2>{}, ph);
        step(std::integral_constant<int, 3>{}, ph);
        step(std::integral_constant<int, 4>{}, ph);
        step(std::integral_constant<int, 5>{}, ph);
        step(std::integral_constant<int, 6>{}, ph);
        step(std::integral_constant<int, 7>{}, ph);
        step(std::integral_constant<int, 8>{}, ph);
        step(std::integral_constant<int, 9>{}, ph);
        step(std::integral_constant<int, 10>{}, ph);
        step(std::integral_constant<int, 11>{}, ph);
        step(std::integral_constant<int, 12>{}, ph);
        step(std::integral_constant<int, 13>{}, ph);
        step(std::integral_constant<int, 14>{}, ph);
        step(std::integral_constant<int, 15>{}, ph);
        if ((ph + 1) % a.nkh == 0) {
            int64_t f; int pass; bool uok;
            unit_of(ph / a.nkh, f, pass, uok);
            fold(std::integral_constant<int, 15>{}, 1);          // the unit's last accumulator (the next step's fold then adds zeros)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[1][r] = 0.f;
            epilogue(f, pass, uok);
        }
    }
}

typedef void (*WfsKernel)(const ConvWfsArgs);
struct WfsGeo { int D, H, W; WfsKernel k[3]; };      // [pool]
#define WFS_INST(D, H, W) {D, H, W, {k_conv_wfs<D, H, W, 0>, k_conv_wfs<D, H, W, 1>, k_conv_wfs<D, H, W, 2>}}
const WfsGeo kWfsGeo[] = {WFS_INST(10, 10, 10)};
#undef WFS_INST
struct WfsDbg { int code; WfsKernel k; };
#define WFS_DBG(c) {c, k_conv_wfs<10, 10, 10, 1, c>}
const WfsDbg kWfsDbg[] = {WFS_DBG(1), WFS_DBG(2), WFS_DBG(3), WFS_DBG(4), WFS_DBG(8), WFS_DBG(16), WFS_DBG(20), WFS_DBG(21), WFS_DBG(32), WFS_DBG(29), WFS_DBG(533), WFS_DBG(1045), WFS_DBG(1565), WFS_DBG(1024), WFS_DBG(2048), WFS_DBG(4096), WFS_DBG(8192), WFS_DBG(6144)};
#undef WFS_DBG

inline uint16_t wfs_bf16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline double wfs_bf16_val(uint16_t b) {
    const uint32_t u = (uint32_t)b << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return (double)f;
}

size_t wfs_lds_bytes(int D, int H, int W) {
    const int rsl = (D * H + 1) * (W + 1);
    const int rpad = (4 * rsl + 63) / 64 * 64;
    return (size_t)rpad * 16 + (size_t)2 * 3 * (256 + 2) * 2 * 16 + (size_t)2 * kWfsFrags * 64 * 16;
}

}  // namespace

// ---- host side -------------------------------------------------------------------------------------------------------------
// `base` is the layer's conv_wfused plan (geometry and padding already checked there); the split form serves it when the knobs
// allow, the layer has no input prologue (the slice goes from HBM to LDS without passing through registers), Cin is a multiple
// of 16 (one k-step per phase) and more than 32 output channels exist (a pass multiplies two 32-column tiles)
bool conv_wfs_plan(const ConvWfPlan& base, const TView& in, const PreOp& pre, ConvWfsPlan* p) {
    const ThKnobs& kn = th_knobs_planning();
    if (!kn.wf_split || base.geo < 0) return false;
    if (conv_wf_pre_kind(pre) != 0) return false;
    if (base.Cin % 16 != 0 || base.Cin < 16 || base.Cout <= 32) return false;
    int geo = -1;
    for (size_t k = 0; k < sizeof kWfsGeo / sizeof kWfsGeo[0]; ++k)
        if (kWfsGeo[k].D == in.D && kWfsGeo[k].H == in.H && kWfsGeo[k].W == in.W) geo = (int)k;
    if (geo < 0) return false;
    p->geo = geo; p->pool = base.pool; p->Cin = base.Cin; p->Cout = base.Cout;
    p->nkh = base.Cin / 16;
    p->ncp = (base.Cout + 63) / 64;
    p->knobs = &kn;
    p->wpk_floats = (size_t)p->ncp * p->nkh * 16 * kWfsFrags * 64 * 4 + 16;       // + 64 bytes of zeros (the halo's DMA source)
    p->own_flops = base.own_flops;
    // what the MFMAs issue: 8 row tiles x 2 column tiles x 3 z taps x 6 products of 32 x 32 x 16 per (position, phase)
    p->exec_flops = 2.0 * 32 * 32 * 16 * 6 * 3 * 2 * 8 * 16.0 * p->nkh * p->ncp;
    p->lds_bytes = wfs_lds_bytes(in.D, in.H, in.W);
    if (p->lds_bytes > kWfsLdsLimit) return false;
    char buf[320];
    snprintf(buf, sizeof buf, "conv_wf<F(2,3)^2 in-plane fused in LDS, z direct; pool%d; bf16x3 split operands, 6 products, fp32 accumulate> 64c x %d, K%d, "
             "lds%zuK (32x32x16 bf16 MFMA) [k_conv_wfs<%d,%d,%d,%d,0>]", p->pool, p->ncp, p->Cin, p->lds_bytes / 1024, in.D, in.H, in.W, p->pool);
    p->label = buf;
    return true;
}

// Keras [3][3][3][Cin][Cout] -> U[a][b][dz][ci][co] = sum_jk G[a][j] G[b][k] W[dz][j][k][ci][co] (double), G of F(2,3), each value
// split into three bf16 pieces (residuals in double), laid out as the B fragments of v_mfma_f32_32x32x16_bf16:
// [pass][phase][pos = 4 a + b][dz][column tile n][piece][lane = 32 hh + j][e]: ci = 16 phase + 8 hh + e, co = 64 pass + 32 n + j
void conv_wfs_pack_weights(const ConvWfsPlan& p, const float* w, float* dst_f) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int Cin = p.Cin, Cout = p.Cout;
    std::memset(dst_f, 0, p.wpk_floats * sizeof(float));
    uint16_t* dst = reinterpret_cast<uint16_t*>(dst_f);
    for (int ci = 0; ci < Cin; ++ci)
        for (int co = 0; co < Cout; ++co) {
            const int pass = co / 64, n = (co % 64) / 32, j = co % 32;
            const int kh = ci / 16, hh = (ci % 16) / 8, e = ci % 8;
            for (int dz = 0; dz < 3; ++dz) {
                double wk[3][3];
                for (int jj = 0; jj < 3; ++jj)
                    for (int k = 0; k < 3; ++k) wk[jj][k] = (double)w[((((size_t)dz * 3 + jj) * 3 + k) * Cin + ci) * Cout + co];
                for (int aa = 0; aa < 4; ++aa) {
                    double t[3];
                    for (int k = 0; k < 3; ++k) t[k] = G[aa][0] * wk[0][k] + G[aa][1] * wk[1][k] + G[aa][2] * wk[2][k];
                    for (int bb = 0; bb < 4; ++bb) {
                        const double u = G[bb][0] * t[0] + G[bb][1] * t[1] + G[bb][2] * t[2];
                        uint16_t pc[3];
                        pc[0] = wfs_bf16_rne((float)u);
                        const double r1 = u - wfs_bf16_val(pc[0]);
                        pc[1] = wfs_bf16_rne((float)r1);
                        const double r2 = r1 - wfs_bf16_val(pc[1]);
                        pc[2] = wfs_bf16_rne((float)r2);
                        const size_t stepi = (size_t)(pass * p.nkh + kh) * 16 + (aa * 4 + bb);
                        for (int piece = 0; piece < 3; ++piece) {
                            const size_t frag = stepi * kWfsFrags + (size_t)(dz * 2 + n) * 3 + piece;
                            dst[(frag * 64 + 32 * hh + j) * 8 + e] = pc[piece];
                        }
                    }
                }
            }
        }
}

std::string conv_wfs_label(const ConvWfsPlan& p) { return p.label; }

int launch_conv_wfs(hipStream_t s, int64_t n, const ConvWfsPlan& p, TView in, TView out, const float* wpk, const float* bias, PostOps post) {
    if (n <= 0) return TH_OK;
    if (p.geo < 0 || p.geo >= (int)(sizeof kWfsGeo / sizeof kWfsGeo[0])) TH_FAIL(TH_EINVAL, "conv_wfs: bad plan");
    if (in.cs % 4 || in.coff % 4 || in.fs % 4 || ((uintptr_t)in.p % 16)) TH_FAIL(TH_EINVAL, "conv_wfs: the input view is not 16-byte aligned");
    if (in.blk && (in.blk != 4 || in.coff || in.cs != p.Cin)) TH_FAIL(TH_EINVAL, "conv_wfs: bad chunk-blocked input view");
    ConvWfsArgs a;
    std::memset(&a, 0, sizeof a);
    a.in = in.p; a.in_fs = in.fs; a.in_cs = in.cs; a.in_coff = in.coff; a.in_blk = in.blk;
    a.Cin = p.Cin; a.nkh = p.nkh;
    a.wpk = reinterpret_cast<const uint4*>(wpk);
    a.zero16 = wpk + (p.wpk_floats - 16);
    a.Cout = p.Cout; a.ncp = p.ncp; a.bias = bias; a.post = post;
    a.out = out.p; a.out_fs = out.fs; a.out_cs = out.cs; a.out_coff = out.coff; a.nframes = n;
    const int64_t nslots = (n + 7) / 8 * 8 * p.ncp;
    if (nslots > 0x7fffffffLL) TH_FAIL(TH_EINVAL, "conv_wfs: too many frames per launch");
    a.nslots = (unsigned)nslots;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    const ThKnobs& kn = th_knobs_of(p.knobs);
    int64_t resident = ncu;                              // one 8-wave workgroup per CU
    if (kn.wf_resident) resident = std::max(1, kn.wf_resident);
    const int64_t trips = (nslots + resident - 1) / resident;
    int64_t grid = (nslots + trips - 1) / trips;
    grid = (grid + 7) / 8 * 8;
    WfsKernel k = kWfsGeo[p.geo].k[p.pool];
    if (kn.wf_dbg > 0 && p.geo == 0 && p.pool == 1)
        for (const WfsDbg& d : kWfsDbg) if (d.code == kn.wf_dbg) k = d.k;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(512), 0, s, a);            // (static LDS)
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) TH_FAIL(TH_EHIP, "conv_wfs launch failed: %s (%s)", hipGetErrorString(e), p.label.c_str());
    return TH_OK;
}
