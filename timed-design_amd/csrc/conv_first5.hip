// 5x5x5 'same' convolution on the model input (21^3 x <= 8 channels -> <= 16 filters, 2^3 max-pool behind it) — ProDCoNN's wide
// stem (the call served is reference predict.py:142) — on the bf16 matrix pipe with exactly split operands.
// It was the one benchmark layer without a fast path: 81 % of ProDCoNN-synth on the generic fp32 kernel at 0.25 of the pipe
// (6 channels padded to 8, 16 columns in a 32-wide tile).
//
// Direct form (125 taps: a minimal-filtering form would need 6 points per x pair and three times the staged data), products as
// v_mfma_f32_16x16x32_bf16 on operands split exactly into three bf16 pieces, six of nine piece products, fp32 accumulation — the
// scheme of conv_first_b3.hip / conv_wfsplit.hip, and like the latter the data is split ONCE, when it is staged:
//   D3  [piece 3][ring of 6 planes][14 rows][25 of 28 x] x 16 bytes   a voxel's (<= 8) channels as one bf16x8 record per piece, zero halo
//   B   [2 buffers][7 k-steps][piece 3][lane 64] x 16 bytes     the weights of ONE z tap, LDS-DMA, double-buffered over the 5 z taps
// A k-step (K = 32) is four voxel records: for a (dz, dy) row the taps dx = 0..3 (lane group kg <-> dx), and the fifth tap dx = 4
// of four different dy rides in a "tail" k-step (kg <-> dy): 7 k-steps per z tap, 750 of 1120 k-slots useful.
// Work unit = (frame, half of the y range, pooled z plane): 400 conv outputs = 25 tiles of 16 rows; a row is a MEMBER of a pool
// window (row = 8 pooled voxel + (dx2, dz2, dy2)), so that a lane's four accumulator rows and its neighbour 16 lanes away hold
// one window.  One persistent 8-wave workgroup per CU walks frame -> y half -> pooled plane; consecutive planes share four of
// their six input planes: the two new ones are requested when a unit starts and written (split) when it ends.
#include "common.h"
#include "device_math.h"

#include <hip/hip_fp16.h>

#include <algorithm>
#include <cstring>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int kF5D = 21;                       // frame extent
constexpr int kF5X = 25, kF5Y = 14;            // staged x (-2 .. 22) and rows (y0 - 2 .. y0 + 11) of a plane half
constexpr int kF5PlaneVox = kF5X * kF5Y;       // 350 staged voxels per plane half
// LDS strides in records: a row is 28 records and a plane 392 (= 8 mod 16), and a tile row is 8 pooled voxel + (dx2, dz2, dy2):
// with these the 16 lanes that one cycle of a ds_read_b128 serves (MI355X_MICROARCH.md §LDS) fall on 16 different 16-byte slots
// of the 256-byte bank row in every k-step (or on the same record).  With the dense strides 25 / 350 and the order (dz2, dy2, dx2)
// every A fragment read was a two-way bank conflict (5.60 -> 5.04 ms per 4096 frames).
constexpr int kF5RS = 28, kF5PS = kF5Y * kF5RS;
static_assert(kF5PS % 16 == 8, "plane stride");
constexpr int kF5Ring = 6;
constexpr int kF5Piece = kF5Ring * kF5PS;              // records per piece
constexpr int kF5KS = 7;                       // k-steps per z tap
constexpr int kF5Frags = kF5KS * 3;            // B fragments per z tap
constexpr int kF5Tiles = 25;
constexpr size_t kF5WpkFloats = (size_t)5 * kF5Frags * 64 * 4;

struct ConvF5Args {
    const void* in; int dtype; int Cin; int vec8;
    const uint4* wpk;                 // [dz 5][k-step 7][piece 3][lane 64] x 8 bf16
    int Cout;
    const float* bias;
    PostOps post;
    float* out; int64_t out_fs; int out_cs, out_coff, Ho, Wo;
    int64_t nframes;
};

__device__ __forceinline__ float f5_load_elem(const void* base, int dtype, int64_t i) {
    switch (dtype) {
        case TH_F32: return ((const float*)base)[i];
        case TH_F64: return (float)((const double*)base)[i];
        case TH_U8: return (float)((const unsigned char*)base)[i];
        case TH_BOOL: return ((const unsigned char*)base)[i] ? 1.f : 0.f;
        default: return __half2float(((const __half*)base)[i]);
    }
}
__device__ __forceinline__ unsigned f5_pk(float x, float y) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((v2f){x, y}, bf16x2));
}
__device__ __forceinline__ void f5_glds(const void* base, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}

// DBG (TH_FIRST_DBG; results WRONG): 1 no staging after the first unit, 2 no MFMAs, 4 weights loaded once
template <int DBG>
__global__ void __launch_bounds__(512, 1) k_conv_first5(const ConvF5Args a) {
    __shared__ __attribute__((aligned(16))) uint4 D3[3 * kF5Piece];
    __shared__ __attribute__((aligned(16))) uint4 Ba[kF5Frags * 64];
    __shared__ __attribute__((aligned(16))) uint4 Bb[kF5Frags * 64];
    __shared__ __attribute__((aligned(16))) f32x4 Px[kF5KS * 64];         // tile 24's partial sums, one per k-step owner

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i16 = lane & 15, kg = lane >> 4;
    const int64_t G = gridDim.x;
    const int nk = (int)((a.nframes - blockIdx.x + G - 1) / G);
    if (nk <= 0) return;
    const int64_t frame_elems = (int64_t)kF5D * kF5D * kF5D * a.Cin;

    // ---- staging: voxel idx = tid + 512 k of a run of planes; raw channels in registers between issue and commit -------------
    float e[2][8];
    int sdst[2];
    auto load_voxel = [&](int64_t f, int yh, int z, int v, float (&x)[8]) {
#pragma unroll
        for (int c = 0; c < 8; ++c) x[c] = 0.f;
        const int row = v / kF5X, xi = v - row * kF5X;
        const int y = 10 * yh - 2 + row, xx = xi - 2;
        if (z < 0 || z >= kF5D || y < 0 || y >= kF5D || xx < 0 || xx >= kF5D) return;
        const int64_t base = f * frame_elems + ((int64_t)(z * kF5D + y) * kF5D + xx) * a.Cin;
        if (a.vec8) {
            const float2* p2 = reinterpret_cast<const float2*>((const float*)a.in + base);
            const float2 u0 = p2[0], u1 = p2[1], u2 = p2[2];
            x[0] = u0.x; x[1] = u0.y; x[2] = u1.x; x[3] = u1.y; x[4] = u2.x; x[5] = u2.y;
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (c < a.Cin) x[c] = f5_load_elem(a.in, a.dtype, base + c);
        }
    };
    auto store_voxel = [&](int dst, const float (&x)[8]) {
        unsigned h[4], m[4], l[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float x0 = x[2 * k], x1 = x[2 * k + 1];
            h[k] = f5_pk(x0, x1);
            const float r0 = x0 - __builtin_bit_cast(float, h[k] << 16), r1 = x1 - __builtin_bit_cast(float, h[k] & 0xffff0000u);
            m[k] = f5_pk(r0, r1);
            const float q0 = r0 - __builtin_bit_cast(float, m[k] << 16), q1 = r1 - __builtin_bit_cast(float, m[k] & 0xffff0000u);
            l[k] = f5_pk(q0, q1);
        }
        D3[dst] = make_uint4(h[0], h[1], h[2], h[3]);
        D3[kF5Piece + dst] = make_uint4(m[0], m[1], m[2], m[3]);
        D3[2 * kF5Piece + dst] = make_uint4(l[0], l[1], l[2], l[3]);
    };
    auto slot_of = [&](int z) { return (z + 2) % kF5Ring; };
    auto lds_of = [&](int v) { const int row = v / kF5X; return row * kF5RS + (v - row * kF5X); };
    // the two planes z_lo, z_lo + 1 (700 voxels): requested here ...
    auto issue2 = [&](int64_t f, int yh, int z_lo) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int idx = tid + 512 * k;
            sdst[k] = -1;
            if (idx < 2 * kF5PlaneVox) {
                const int pl = idx >= kF5PlaneVox ? 1 : 0, v = idx - pl * kF5PlaneVox;
                load_voxel(f, yh, z_lo + pl, v, e[k]);
                sdst[k] = slot_of(z_lo + pl) * kF5PS + lds_of(v);
            }
        }
    };
    auto commit2 = [&]() {      // ... split and written here
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (sdst[k] >= 0) store_voxel(sdst[k], e[k]);
    };
    auto stage_sync = [&](int64_t f, int yh, int z_lo, int nz) {
        for (int idx = tid; idx < nz * kF5PlaneVox; idx += 512) {
            const int pl = idx / kF5PlaneVox, v = idx - pl * kF5PlaneVox;
            float x[8];
            load_voxel(f, yh, z_lo + pl, v, x);
            store_voxel(slot_of(z_lo + pl) * kF5PS + lds_of(v), x);
        }
    };
    // weights of z tap dz into buffer `buf`: 21 wave-wide 1 KB loads, three per wave (the ragged third round repeats the second)
    const unsigned lane16 = (unsigned)lane * 16u;
    auto issue_B = [&](int dz, int buf) {
        const char* const base = reinterpret_cast<const char*>(a.wpk) + (size_t)dz * kF5Frags * 1024;
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int q0 = wave + 8 * t, q = q0 < kF5Frags ? q0 : q0 - 8;
            f5_glds(base + q * 1024, lane16, (unsigned)(uintptr_t)((buf ? Bb : Ba) + q * 64));
        }
    };

    // ---- per-lane row geometry of the wave's tiles: tile = wave + 8 k; row = 16 tile + i16 = 8 q + (dx2, dz2, dy2) ------------
    // tiles wave, wave + 8, wave + 16 are this wave's; tile 24 (k = 3) is nobody's: wave w < 7 runs its k-step w in every phase (22 items
    // per wave and phase instead of 21 and 28 on one, for which the other seven waited at the phase's barrier: MFMA busy was 57 %), the
    // seven partial sums meet in LDS when the unit ends and wave 7 adds them in a fixed order
    int rowbase[4], zsel[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int tile = k < 3 ? wave + 8 * k : kF5Tiles - 1;
        const int r = 16 * tile + i16, q = r >> 3, mm = r & 7;
        const int pyl = q / 10, px = q - 10 * pyl;
        rowbase[k] = (2 * pyl + (mm & 1)) * kF5RS + 2 * px + (mm >> 2);
        zsel[k] = (mm >> 1) & 1;
    }
    // k-step offsets (records): main k-step ks = dy: + dy 25 + kg; tail 5: dy = kg, dx = 4; tail 6: dy = 4, dx = 4 (kg 0 only)
    int koff[kF5KS];
#pragma unroll
    for (int ks = 0; ks < kF5KS; ++ks) koff[ks] = ks < 5 ? ks * kF5RS + kg : (ks == 5 ? kg * kF5RS + 4 : 4 * kF5RS + 4);

    const int co = i16;
    const bool cok = co < a.Cout;
    const int cc = cok ? co : 0;
    const float bv = a.bias ? a.bias[cc] : 0.f;
    const bool pool_first = a.post.monotone != 0;

    int phase = 0;                                            // z-tap phases so far: the weight buffer in use is phase & 1
    issue_B(0, 0);
    for (int k = 0; k < nk; ++k) {
        const int64_t f = blockIdx.x + (int64_t)k * G;
        float* const outb = a.out + f * a.out_fs + a.out_coff + cc;
        for (int u = 0; u < 20; ++u) {
            const int yh = u / 10, pz = u - 10 * yh;
            if (pz == 0) {                                    // a new half: its first six planes z = -2 .. 3
                if (!(DBG & 1) || (k == 0 && u == 0)) stage_sync(f, yh, -2, 6);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            // the next unit's two new planes (z = 2 pz + 4, + 5) replace the two that die with this unit
            const bool more = pz < 9;
            if (more && !(DBG & 1)) issue2(f, yh, 2 * pz + 4);
            f32x4 acc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int dz = 0; dz < 5; ++dz, ++phase) {
                const int buf = phase & 1;
                // the weights of the next phase (the next unit's first tap behind the last one) into the other buffer
                if (!(DBG & 4) || phase == 0) issue_B(dz == 4 ? 0 : dz + 1, buf ^ 1);
                const uint4* const Bc = (buf ? Bb : Ba) + lane;
                int base[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) base[t] = ((2 * pz + zsel[t] + dz) % kF5Ring) * kF5PS + rowbase[t];
                // 21 items (k-step, own tile) of six MFMAs; the fragments of item it + 1 are requested before the MFMAs of item it and the
                // items are fenced: left alone hipcc reads a fragment right in front of its first use and every item starts with an
                // LDS round trip that only one other wave per SIMD can cover
                bf16x8 Af[2][3], Bf[2][3];
                auto loadA = [&](int t, int ks, int set) __attribute__((always_inline)) {
                    const int ad = base[t] + koff[ks];
                    Af[set][0] = __builtin_bit_cast(bf16x8, D3[ad]);
                    Af[set][1] = __builtin_bit_cast(bf16x8, D3[kF5Piece + ad]);
                    Af[set][2] = __builtin_bit_cast(bf16x8, D3[2 * kF5Piece + ad]);
                };
                auto loadB = [&](int ks, int set) __attribute__((always_inline)) {
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) Bf[set][pc] = __builtin_bit_cast(bf16x8, Bc[(ks * 3 + pc) * 64]);
                };
                auto mma6 = [&](f32x4& c, const bf16x8 (&A)[3], const bf16x8 (&B)[3]) __attribute__((always_inline)) {
                    if (DBG & 2) { c[0] += __builtin_bit_cast(float, __builtin_bit_cast(uint4, A[0]).x ^ __builtin_bit_cast(uint4, B[1]).y); return; }
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[2], B[0], c, 0, 0, 0);      // l H
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[1], B[0], c, 0, 0, 0);      // m H
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], B[0], c, 0, 0, 0);      // h H
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[1], B[1], c, 0, 0, 0);      // m M
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], B[1], c, 0, 0, 0);      // h M
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[0], B[2], c, 0, 0, 0);      // h L
                };
                loadB(0, 0);
                loadA(0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int it = 0; it < 3 * kF5KS; ++it) {
                    const int ks = it / 3, t = it % 3;
                    if (it + 1 < 3 * kF5KS) {
                        const int nks = (it + 1) / 3, nt = (it + 1) % 3;
                        if (nt == 0) loadB(nks, nks & 1);
                        loadA(nt, nks, (it + 1) & 1);
                    }
                    mma6(acc[t], Af[it & 1], Bf[ks & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (t == 2 && wave == ks) {               // this wave's k-step of tile 24, with the weights it holds anyway
                        bf16x8 Ax[3];
                        const int ad = base[3] + koff[ks];
                        Ax[0] = __builtin_bit_cast(bf16x8, D3[ad]);
                        Ax[1] = __builtin_bit_cast(bf16x8, D3[kF5Piece + ad]);
                        Ax[2] = __builtin_bit_cast(bf16x8, D3[2 * kF5Piece + ad]);
                        mma6(acc[3], Ax, Bf[ks & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                if (dz == 4 && wave < kF5KS) Px[wave * 64 + lane] = acc[3];
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the next weights (and the prefetched voxels)
                __syncthreads();                                       // everybody is done with this phase's buffer and, after dz = 4, the planes
            }
            if (wave == kF5KS) {
                acc[3] = Px[lane];
#pragma unroll
                for (int w = 1; w < kF5KS; ++w) acc[3] += Px[w * 64 + lane];
            }
            // ---- the unit's outputs: bias, (chain,) 2^3 max over a lane's four rows and its neighbour 16 lanes away, (chain,) store
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (t == 3 && wave != kF5KS) continue;
                float v[4] = {acc[t][0] + bv, acc[t][1] + bv, acc[t][2] + bv, acc[t][3] + bv};
                if (!pool_first) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = th_post(v[r], cc, a.post);
                }
                float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
                mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
                if (pool_first) mx = th_post(mx, cc, a.post);
                const int tile = t < 3 ? wave + 8 * t : kF5Tiles - 1, q = 2 * tile + (kg >> 1);
                const int pyl = q / 10, px = q - 10 * pyl;
                if (cok && !(kg & 1)) outb[(int64_t)((pz * a.Ho + 5 * yh + pyl) * a.Wo + px) * a.out_cs] = mx;
            }
            if (more && !(DBG & 1)) {
                commit2();
                __syncthreads();
            }
        }
    }
}

typedef void (*F5Kernel)(const ConvF5Args);

inline uint16_t f5_bf16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline double f5_bf16_val(uint16_t b) {
    const uint32_t u = (uint32_t)b << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return (double)f;
}

}  // namespace

// does the kernel serve this layer?  (asked by the planner before anything is packed)
bool conv_first5_ok(int Din, int Hin, int Win, int Cin, int Cout, const ConvGeom& g, int pool) {
    const ThKnobs& kn = th_knobs_planning();
    if (!kn.first_split || kn.no_pool_first) return false;
    if (Din != kF5D || Hin != kF5D || Win != kF5D || Cin < 1 || Cin > 8 || Cout < 1 || Cout > 16) return false;
    if (g.kd != 5 || g.kh != 5 || g.kw != 5 || g.sd != 1 || g.sh != 1 || g.sw != 1 || g.dd != 1 || g.dh != 1 || g.dw != 1) return false;
    if (g.pz != 2 || g.py != 2 || g.px != 2) return false;                      // 'same'
    return pool == 1;                                                           // 2^3 max pool behind it (conv outputs 0 .. 19 per axis)
}
size_t conv_first5_wpk_floats() { return kF5WpkFloats; }
// MFMA FLOPs issued per frame: 20 units x 25 tiles x 35 k-steps x 6 products of 16 x 16 x 32
double conv_first5_exec_flops() { return 20.0 * kF5Tiles * 5 * kF5KS * 6 * 2.0 * 16 * 16 * 32; }
std::string conv_first5_label() {
    char buf[256];
    snprintf(buf, sizeof buf, "conv_first5<5x5x5 direct, 2^3 max-pool> persistent, ring of %d plane halves split once at staging, lds%zuK; bf16x3 split "
             "operands, 6 products, fp32 accumulate (16x16x32 bf16 MFMA, one z tap of weights in LDS, direct input) [k_conv_first5]",
             kF5Ring, (sizeof(uint4) * (3 * kF5Piece + 2 * kF5Frags * 64)) / 1024);
    return buf;
}

// Keras [5][5][5][Cin][Cout] -> three bf16 pieces of every weight (residuals in double), as the B fragments of the 16x16x32 MFMA:
// [dz][k-step][piece][lane = 16 kg + co][e]; k-step ks < 5: tap (dz, dy = ks, dx = kg), channel e; ks = 5: (dz, dy = kg, dx = 4);
// ks = 6: (dz, dy = 4, dx = 4) for kg = 0, zero elsewhere
void conv_first5_pack_weights(int Cin, int Cout, const float* w, float* dst_f) {
    std::memset(dst_f, 0, kF5WpkFloats * sizeof(float));
    uint16_t* dst = reinterpret_cast<uint16_t*>(dst_f);
    for (int dz = 0; dz < 5; ++dz)
        for (int ks = 0; ks < kF5KS; ++ks)
            for (int kgi = 0; kgi < 4; ++kgi) {
                int dy, dx;
                if (ks < 5) { dy = ks; dx = kgi; }
                else if (ks == 5) { dy = kgi; dx = 4; }
                else { if (kgi) continue; dy = 4; dx = 4; }
                for (int c = 0; c < Cin; ++c)
                    for (int co = 0; co < Cout; ++co) {
                        const double u = (double)w[((((size_t)dz * 5 + dy) * 5 + dx) * Cin + c) * Cout + co];
                        uint16_t pc[3];
                        pc[0] = f5_bf16_rne((float)u);
                        const double r1 = u - f5_bf16_val(pc[0]);
                        pc[1] = f5_bf16_rne((float)r1);
                        pc[2] = f5_bf16_rne((float)(r1 - f5_bf16_val(pc[1])));
                        for (int piece = 0; piece < 3; ++piece)
                            dst[((((size_t)dz * kF5KS + ks) * 3 + piece) * 64 + 16 * kgi + co) * 8 + c] = pc[piece];
                    }
            }
}

int launch_conv_first5(hipStream_t s, int64_t n, const ThKnobs* knobs, const void* frames, int dtype, int Cin, TView out, int Cout,
                       const float* wpk, const float* bias, PostOps post) {
    if (n <= 0) return TH_OK;
    if (out.D != 10 || out.H != 10 || out.W != 10) TH_FAIL(TH_EINVAL, "conv_first5: the pooled output is not 10^3");
    ConvF5Args a;
    std::memset(&a, 0, sizeof a);
    a.in = frames; a.dtype = dtype; a.Cin = Cin;
    a.vec8 = (dtype == TH_F32 && Cin == 6 && ((uintptr_t)frames % 8) == 0) ? 1 : 0;
    a.wpk = reinterpret_cast<const uint4*>(wpk);
    a.Cout = Cout; a.bias = bias; a.post = post;
    a.out = out.p; a.out_fs = out.fs; a.out_cs = out.cs; a.out_coff = out.coff; a.Ho = out.H; a.Wo = out.W;
    a.nframes = n;
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    const ThKnobs& kn = th_knobs_of(knobs);
    int64_t resident = ncu;
    if (kn.wf_resident) resident = std::max(1, kn.wf_resident);       // tests: several frames per workgroup on small batches
    const int64_t trips = (n + resident - 1) / resident;
    const int64_t grid = (n + trips - 1) / trips;
    F5Kernel k = k_conv_first5<0>;
    if (kn.first_dbg == 1) k = k_conv_first5<1>;
    else if (kn.first_dbg == 2) k = k_conv_first5<2>;
    else if (kn.first_dbg == 4) k = k_conv_first5<4>;
    else if (kn.first_dbg == 7) k = k_conv_first5<7>;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(512), 0, s, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) TH_FAIL(TH_EHIP, "conv_first5 launch failed: %s", hipGetErrorString(e));
    return TH_OK;
}
