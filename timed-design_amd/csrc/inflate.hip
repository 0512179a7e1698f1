// DEFLATE (RFC 1951) / zlib (RFC 1950) decoding ON THE GPU, one lane per compressed chunk.
//
// Why: the reference's frames live in an aposteriori .hdf5 whose per-residue datasets are gzip-compressed chunks
// (reference design_utils/utils.py:514-529 reads them one h5py call at a time).  Inflating them on the host is the
// end-to-end limiter of predict.py on real datasets: one core inflates 1.3 k frames/s (444 KB of float64 each) and the GPU
// box's container has 16 usable cores -> 10-14 k frames/s in front of a GPU that consumes 226 k.  A
// 100 k-frame dataset is ~10^6 independent deflate streams: there is no parallelism inside a stream, but there is all
// the parallelism one could want across them.  So: the COMPRESSED bytes cross PCIe (8.7x fewer than the float64 frames),
// every lane of every wavefront decodes its own stream, and the chunks are placed into the channels-last frame tensor
// (float64 -> float32 on the way: Keras' own cast).
//
// Two kernels, because the two halves of DEFLATE parallelise differently:
//   k_inflate_tokens   one LANE per stream: the Huffman layer (inherently serial per stream, so the parallelism is across
//                      streams) — canonical decode by code length without a data-dependent loop, the two sorted symbol tables
//                      per lane in LDS (320 B per stream: eight wavefronts per CU), everything else in registers, a predicated
//                      straight-line symbol loop, all memory traffic at wave-synchronous points.  It does NOT produce bytes:
//                      it writes 32-bit tokens (a literal, or a match of length L at distance D), so it never reads its own
//                      output back; it also hands the zlib trailer (Adler-32) to the second kernel.
//   k_lz_resolve       one WAVEFRONT per stream: the LZ77 layer in LDS — literals of 64 tokens land in parallel (DPP prefix
//                      sum of the lengths), every match is copied in one step by the whole wavefront, the stream's Adler-32
//                      is verified, and a chunk that fits the LDS window whole is placed from there straight into its frame
//                      (float64 -> float32 on the way); longer streams go round a 64 KB ring and leave in 8-byte stores for
//                      k_place_chunks.
// A first, single-kernel version (a lane emitting bytes into HBM and reading far matches back from there) spent 64 ms on
// the 32 768 chunks of a 1 024-frame batch: every read-back of a lane's own earlier store waits for the wave's whole store
// queue, and with 64 independent streams per wave some lane is always at such a point.
// Every loop is bounded by the declared input / output lengths: corrupt data ends in a status code, never in a hang or an
// out-of-range access.
#include "common.h"

#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <type_traits>
#include <vector>

namespace {

constexpr int kLanes = 64;
constexpr int kMaxBits = 15, kMaxLCodes = 286, kMaxDCodes = 30, kFixLCodes = 288;

struct InfDesc { int64_t src_off, src_len, dst_off, dst_len, tok_off; };

// status per chunk
enum { INF_OK = 0, INF_EINPUT = 1, INF_EOUTPUT = 2, INF_EHEADER = 3, INF_ECODE = 4, INF_EDIST = 5, INF_ETABLE = 6, INF_ESTORED = 7, INF_ESHORT = 8, INF_ECHECK = 9 };

// Tables of one wavefront, [entry][lane]: the symbols of the two Huffman codes of the current block, sorted by (code length, symbol)
// — 288 + 32 bytes per stream and NOTHING else, because 64 streams x 320 B = 20 KB is what lets EIGHT wavefronts share the 160 KB of
// a CU: the kernel is bound by instruction latency with every lane on its own data-dependent path, and two wavefronts per SIMD hide
// each other (measured with 32 streams per wavefront: 2 x as many wavefronts in 1.22 x the time).  Everything else lives in
// registers or is not kept at all: bit 8 of the literal/length symbols (9 words), the per-length limits and bases of both codes
// (struct Code), the code-length code of a dynamic header (19 x 3 bits of lengths, 19 x 5 bits of sorted symbols) — and the 316
// code LENGTHS of a dynamic block are never stored: the header is decoded twice, once to count the codes per length (the counters
// borrow the first 64 rows of lsym, dead at that point) and once more, from the saved bit position, to put every symbol in its place.
// LPW = streams (active lanes) per wavefront: small batches use fewer, so that the streams spread over the CUs.
template <int LPW>
struct LdsT {
    unsigned char lsym[kFixLCodes][LPW];
    unsigned char dsym[32][LPW];
};
// bit 8 of the sorted literal/length symbols: 288 bits in nine NAMED registers (an array indexed per lane would live in scratch)
struct Hi9 { unsigned h0, h1, h2, h3, h4, h5, h6, h7, h8; };
__device__ __forceinline__ unsigned hi_bit(const Hi9& h, int idx) {       // idx in 0..287
    const int k = idx >> 5;
    const unsigned a = (k & 1) ? h.h1 : h.h0, bb = (k & 1) ? h.h3 : h.h2, c = (k & 1) ? h.h5 : h.h4, d = (k & 1) ? h.h7 : h.h6;
    const unsigned ab = (k & 2) ? bb : a, cd = (k & 2) ? d : c;
    const unsigned w = (k & 8) ? h.h8 : ((k & 4) ? cd : ab);
    return (w >> (idx & 31)) & 1u;
}
__device__ __forceinline__ void hi_set(Hi9& h, int idx) {
    const int k = idx >> 5;
    const unsigned bit = 1u << (idx & 31);
    h.h0 |= k == 0 ? bit : 0u; h.h1 |= k == 1 ? bit : 0u; h.h2 |= k == 2 ? bit : 0u; h.h3 |= k == 3 ? bit : 0u; h.h4 |= k == 4 ? bit : 0u;
    h.h5 |= k == 5 ? bit : 0u; h.h6 |= k == 6 ? bit : 0u; h.h7 |= k == 7 ? bit : 0u; h.h8 |= k == 8 ? bit : 0u;
}


// Bit reader: the compressed bytes arrive 16 at a time (one aligned global load per 16 bytes — a byte-wise reader pays the
// ~700-cycle global latency per input byte, with all 64 lanes of the wave waiting in lockstep), 32 bits at a time into the
// 64-bit bit buffer.  `remaining` counts the bits the stream still owns: going negative means the input was exhausted.
struct Bits {
    const uint4* blk;       // next aligned 16-byte block
    const uint4* blk_end;   // one past the last block that may be read
    const uint4* blk_first; // the block the stream begins in
    uint4 cur;              // the block being consumed
    uint4 nxt;              // the block after it, requested when `cur` was taken: a load issued only when its data is needed
                            // would have to wait — behind every token store the wavefront has queued before it — on each refill
    int widx;               // next 32-bit word of `cur` (4: none left)
    unsigned long long buf;
    int cnt;
    long long remaining;
    bool over;
};

// A block load must be a GLOBAL load: through a generic pointer it is a flat load, which also counts in lgkmcnt — every wait
// for an LDS table read then waits for the block in flight as well.  (HIP's uint4 is a class: its copy goes through a generic
// reference, so the block is loaded as a native vector through an address-space-1 pointer.)
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) v4u* GlobalBlockPtr;
__device__ __forceinline__ uint4 load_global_block(const uint4* p) {
    const v4u v = *(GlobalBlockPtr)(unsigned long long)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ unsigned next_word(Bits& b) {
    if (b.widx == 4) {
        b.cur = b.nxt;
        if (b.blk < b.blk_end) b.nxt = load_global_block(b.blk);
        else b.nxt = make_uint4(0, 0, 0, 0);
        ++b.blk;
        b.widx = 0;
    }
    const unsigned w = b.widx == 0 ? b.cur.x : (b.widx == 1 ? b.cur.y : (b.widx == 2 ? b.cur.z : b.cur.w));
    ++b.widx;
    return w;
}
__device__ __forceinline__ void refill(Bits& b) {
    if (b.cnt <= 32) {
        b.buf |= (unsigned long long)next_word(b) << b.cnt;
        b.cnt += 32;
    }
}
__device__ __forceinline__ void bits_init(Bits& b, const unsigned char* p, long long len, const unsigned char* buf_end) {
    const unsigned long long a = (unsigned long long)p;
    b.blk = reinterpret_cast<const uint4*>(a & ~15ull);
    b.blk_first = b.blk;
    b.blk_end = reinterpret_cast<const uint4*>(((unsigned long long)buf_end + 15ull) & ~15ull);
    const uint4* own_end = reinterpret_cast<const uint4*>(((unsigned long long)(p + len) + 15ull) & ~15ull);
    if (own_end < b.blk_end) b.blk_end = own_end;       // never read past the stream's last block (nor past the buffer)
    b.widx = 4; b.buf = 0; b.cnt = 0; b.over = false;
    b.remaining = 8 * len;
    b.cur = make_uint4(0, 0, 0, 0);
    if (b.blk < b.blk_end) b.nxt = load_global_block(b.blk);            // prime the pipeline: the first block
    else b.nxt = make_uint4(0, 0, 0, 0);
    ++b.blk;
    const int skip = (int)(a & 15);
    for (int k = 0; k < (skip >> 2); ++k) (void)next_word(b);   // whole words in front of the stream
    if (skip & 3) {                                             // and the leading bytes of its first word
        b.buf = (unsigned long long)next_word(b) >> (8 * (skip & 3));
        b.cnt = 32 - 8 * (skip & 3);
    }
}
__device__ __forceinline__ unsigned take(Bits& b, int n) {   // n <= 16
    refill(b);
    const unsigned v = (unsigned)(b.buf & ((1ull << n) - 1));
    b.buf >>= n;
    b.cnt -= n;
    b.remaining -= n;
    if (b.remaining < 0) b.over = true;                       // (zeros or foreign bits were returned; the caller checks `over`)
    return v;
}

// Canonical Huffman decode WITHOUT a data-dependent loop (64 lanes decode 64 different streams in lockstep: a bit-serial walk
// over the code lengths makes every lane pay for the longest code in the wave, branch by branch).  Per code, from the counts
// per length:  first[L] = first code of length L, offs[L] = index of its symbol;  LEFT-ALIGNED to 15 bits,
// limit[L] = (first[L] + count[L]) << (15 - L) is non-decreasing in L, and a 15-bit window v of the stream (first bit read =
// most significant) holds a code of length L exactly when limit[L-1] <= v < limit[L].  So L = 1 + #{L' : v >= limit[L']}, and
// the symbol index is (v >> (15 - L)) + base[L] with base[L] = offs[L] - first[L] — 15 compares, 14 selects, no branch.
struct Code { int lim[16]; int bas[16]; };     // [1..15]; bas[1] = base[1], bas[L > 1] = base[L] - base[L-1] (deltas: a select
                                               // chain over base[] itself is folded by hipcc into a register-array index = scratch)
template <int LPW>
__device__ __forceinline__ void code_from_counts(Code& c, const unsigned short (*cnt)[LPW], int lane) {
    int first = 0, offs = 0, prev_base = 0;
#pragma unroll
    for (int len = 1; len <= kMaxBits; ++len) {
        const int count = cnt[len][lane];
        c.lim[len] = (first + count) << (kMaxBits - len);
        c.bas[len] = (offs - first) - prev_base;
        prev_base = offs - first;
        offs += count;
        first = (first + count) << 1;
    }
}
// ---- dynamic block header ------------------------------------------------------------------------------------------------
// The code-length code (RFC 1951 3.2.7): 19 symbols with 3-bit lengths.  Lengths, limits / bases and the sorted symbol table all fit
// in registers (57 + 95 bits of table).
struct ClCode { int lim[8]; int bas[8]; unsigned long long tlo, thi; };      // [1..7]; tlo: sorted symbols 0..11 (5 bits each), thi: 12..18
__device__ __forceinline__ int cl_build(ClCode& c, unsigned long long cl) {   // cl: 3 bits per symbol; returns 0 or INF_ETABLE
    int cnt[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) cnt[k] = 0;
    for (int sy = 0; sy < 19; ++sy) {
        const int l = (int)(cl >> (3 * sy)) & 7;
#pragma unroll
        for (int k = 0; k < 8; ++k) cnt[k] += (l == k) ? 1 : 0;
    }
    int left = 1;
    bool over_sub = false;
#pragma unroll
    for (int len = 1; len <= 7; ++len) {
        left = (left << 1) - cnt[len];
        over_sub = over_sub || left < 0;
    }
    // (no codes at all: every decode fails with INF_ECODE, as with any other code that lacks the symbol; an incomplete or
    // over-subscribed code-length code is refused, as zlib does)
    if (cnt[0] != 19 && (over_sub || left != 0)) return INF_ETABLE;
    int first = 0, offs = 0, prev_base = 0;
    int at[8];
    at[0] = 0;
#pragma unroll
    for (int len = 1; len <= 7; ++len) {
        c.lim[len] = (first + cnt[len]) << (kMaxBits - len);
        c.bas[len] = (offs - first) - prev_base;
        prev_base = offs - first;
        at[len] = offs;
        offs += cnt[len];
        first = (first + cnt[len]) << 1;
    }
    c.tlo = 0; c.thi = 0;
    for (int sy = 0; sy < 19; ++sy) {
        const int l = (int)(cl >> (3 * sy)) & 7;
        int pos = 0;
#pragma unroll
        for (int k = 1; k <= 7; ++k) {
            const bool here = k == l;
            pos += here ? at[k] : 0;
            at[k] += here ? 1 : 0;
        }
        if (l != 0) {
            if (pos < 12) c.tlo |= (unsigned long long)sy << (5 * pos);
            else c.thi |= (unsigned long long)sy << (5 * (pos - 12));
        }
    }
    return INF_OK;
}
__device__ __forceinline__ int cl_decode(Bits& b, const ClCode& c) {         // a symbol 0..18, or -1 (then b.over tells why)
    refill(b);
    const int v = (int)(__brev((unsigned)(b.buf & 0x7fff)) >> 17);
    int len = 1, base = c.bas[1];
#pragma unroll
    for (int l = 1; l < 7; ++l) {
        const bool ge = v >= c.lim[l];
        len += ge ? 1 : 0;
        base += ge ? c.bas[l + 1] : 0;
    }
    if (v >= c.lim[7]) return -1;
    b.buf >>= len;
    b.cnt -= len;
    b.remaining -= len;
    if (b.remaining < 0) { b.over = true; return -1; }
    const int idx = min(max((v >> (kMaxBits - len)) + base, 0), 18);
    return (int)((idx < 12 ? c.tlo >> (5 * idx) : c.thi >> (5 * (idx - 12))) & 31);
}
// the nlen + ndist code lengths of a dynamic block, run-length coded with the code-length code: fn(i, length) for every one of
// them, in order.  Called twice per block with the same bit position: the lengths themselves are never stored.
template <class F>
__device__ __forceinline__ int walk_lengths(Bits& b, const ClCode& c, int total, F&& fn) {
    int i = 0, prev = 0;
    while (i < total) {
        const int sym = cl_decode(b, c);
        if (sym < 0) return b.over ? INF_EINPUT : INF_ECODE;
        int rep = 1, val = sym;
        if (sym >= 16) {
            if (sym == 16) {
                if (i == 0) return INF_ETABLE;
                val = prev;
                rep = 3 + (int)take(b, 2);
            } else if (sym == 17) { val = 0; rep = 3 + (int)take(b, 3); }
            else { val = 0; rep = 11 + (int)take(b, 7); }
            if (b.over) return INF_EINPUT;
            if (i + rep > total) return INF_ETABLE;
        }
        for (int r = 0; r < rep; ++r) fn(i++, val);
        prev = val;
    }
    return INF_OK;
}
// a code's counts per length (LDS, u16) checked like zlib's inflate_table: < 0 over-subscribed, > 0 incomplete, 0 complete
template <int LPW>
__device__ __forceinline__ int counts_left(const unsigned short (*cnt)[LPW], int n, int lane) {
    if (cnt[0][lane] == n) return 0;   // no codes: complete, but decoding will fail
    int left = 1;
    for (int len = 1; len <= kMaxBits; ++len) {
        left <<= 1;
        left -= cnt[len][lane];
        if (left < 0) return left;
    }
    return left;
}
// where the symbols of each length start in the sorted table (registers; updated through select chains)
struct Offs { int o[kMaxBits + 1]; };
template <int LPW>
__device__ __forceinline__ void offs_from_counts(Offs& f, const unsigned short (*cnt)[LPW], int lane) {
    f.o[0] = 0; f.o[1] = 0;
#pragma unroll
    for (int len = 1; len < kMaxBits; ++len) f.o[len + 1] = f.o[len] + cnt[len][lane];
}
__device__ __forceinline__ int offs_take(Offs& f, int l) {      // l in 1..15: the next free slot of that length
    int at = 0;
#pragma unroll
    for (int k = 1; k <= kMaxBits; ++k) {
        const bool here = k == l;
        at += here ? f.o[k] : 0;
        f.o[k] += here ? 1 : 0;
    }
    return at;
}

// token sink of pass 1: literal = the byte; match = bit 31 | length << 16 | (distance - 1)
struct Out {
    unsigned* tok;
    long long nt;       // tokens written (never more than `o`, hence never more than the `len` slots this stream owns)
    long long o, len;   // bytes the tokens stand for so far / declared output length
};
__device__ __forceinline__ void emit(Out& w, unsigned b) {
    w.tok[w.nt++] = b;
    ++w.o;
}

// Length / distance bases and extra-bit counts by formula (RFC 1951 §3.2.5) — as tables in constant memory they were global
// loads per match, and a load in this kernel waits behind every token store queued before it.
__device__ __forceinline__ void len_code(int s, int* base, int* extra) {      // s = symbol - 257 in 0..28
    const int e = s < 8 ? 0 : (s >> 2) - 1;
    *extra = s == 28 ? 0 : e;
    *base = s == 28 ? 258 : (s < 8 ? 3 + s : 3 + ((4 + (s & 3)) << e));
}
__device__ __forceinline__ void dist_code(int s, int* base, int* extra) {     // s in 0..29
    const int e = s < 4 ? 0 : (s >> 1) - 1;
    *extra = e;
    *base = s < 4 ? 1 + s : 1 + ((2 + (s & 1)) << e);
}
// order of the code-length code lengths (RFC 1951 §3.2.7), 5 bits each: 16 17 18 0 8 7 9 6 10 5 11 4 | 12 3 13 2 14 1 15
__device__ __forceinline__ int cl_order(int i) {
    const unsigned long long A = 16ull | 17ull << 5 | 18ull << 10 | 0ull << 15 | 8ull << 20 | 7ull << 25 | 9ull << 30 | 6ull << 35 | 10ull << 40 |
                                 5ull << 45 | 11ull << 50 | 4ull << 55;
    const unsigned long long B = 12ull | 3ull << 5 | 13ull << 10 | 2ull << 15 | 14ull << 20 | 1ull << 25 | 15ull << 30;
    return (int)((i < 12 ? A >> (5 * i) : B >> (5 * (i - 12))) & 31);
}

// Predicated variants for the symbol loop: no refill inside (the loop refills at two fixed points), bits are consumed only
// where `on`, an invalid code gives -1 and consumes nothing, the table index is clamped so the LDS read is always in range.
__device__ __forceinline__ unsigned take_nr(Bits& b, int n) {   // n <= 16, the buffer holds at least n bits
    const unsigned v = (unsigned)b.buf & ((1u << n) - 1u);
    b.buf >>= n;
    b.cnt -= n;
    b.remaining -= n;
    return v;
}
template <int NSYM, class Look>
__device__ __forceinline__ int decode_p(Bits& b, const Code& c, bool on, Look&& look) {
    const int v = (int)(__brev((unsigned)(b.buf & 0x7fff)) >> 17);
    int len = 1, base = c.bas[1];
#pragma unroll
    for (int l = 1; l < kMaxBits; ++l) {
        const bool ge = v >= c.lim[l];
        len += ge ? 1 : 0;
        base += ge ? c.bas[l + 1] : 0;
    }
    const bool bad = v >= c.lim[kMaxBits];          // not a code of this (incomplete) set
    const int use = (on && !bad) ? len : 0;
    b.buf >>= use;
    b.cnt -= use;
    b.remaining -= use;
    const int idx = min(max((v >> (kMaxBits - len)) + base, 0), NSYM - 1);
    const int s = look(idx);
    return bad ? -1 : s;
}

// Input queue of the symbol loop.  Global loads and stores retire in order behind ONE counter (vmcnt): with block switches
// wherever a lane happens to run dry, some lane of the 64 switches in almost every iteration, and its wait for the block it
// asked for 12 tokens ago is a wait for the load another lane issued a moment ago — a full memory latency per token (46 % of the
// wave cycles).  So all memory traffic of the loop happens at wave-synchronous points, every second iteration: the blocks asked
// for two iterations ago have arrived (nothing younger is in flight), the queue advances, the next block is requested and the
// (at most two) tokens of the last two iterations are stored.  Two iterations consume at most 4 words (2 refills of 32 bits
// each): `cur` + `nxt` (8 words, of which at most 3 were used up before) always cover them; `nx2` is the block in flight.
struct InQ { uint4 cur, nxt, nx2; int widx; };     // widx: next word to take, 0..3 in cur, 4..7 in nxt
// (no branch around the load and no zero for "past the end": a select or a merge after the load makes the compiler wait for the
// data on the spot.  Past its last block a stream re-reads that block; whatever is decoded from it is discarded, because
// `remaining` has gone negative by then.  blk_last >= the first block of the stream, which lies inside the buffer.)
__device__ __forceinline__ const uint4* clamp_block(const Bits& b, const uint4* last) { return b.blk < b.blk_end ? b.blk : last; }
__device__ __forceinline__ void refill_q(Bits& b, InQ& q) {     // branch-free: a word is taken where the buffer is at most half full
    const bool need = b.cnt <= 32;
    const bool hi = q.widx >= 4;
    const unsigned x = hi ? q.nxt.x : q.cur.x, y = hi ? q.nxt.y : q.cur.y, z = hi ? q.nxt.z : q.cur.z, t = hi ? q.nxt.w : q.cur.w;
    const int k = q.widx & 3;
    const unsigned wd = k == 0 ? x : (k == 1 ? y : (k == 2 ? z : t));
    b.buf |= need ? ((unsigned long long)wd << b.cnt) : 0ull;
    b.cnt += need ? 32 : 0;
    q.widx += need ? 1 : 0;
}

// literal / length + distance symbols of one compressed block.  The 64 lanes decode 64 different streams: a literal here, a
// match there, an end of block elsewhere.  Written with branches, every one of those cases (and every error exit) is a divergent
// region the wavefront walks through one after the other — 600 instructions and 48 exec-mask regions per token.  So the body is
// straight-line and PREDICATED: every lane runs the literal/length decode, the length extra bits (0 bits unless it holds a match),
// the distance decode and its extra bits (consuming nothing unless it holds a match); what a lane found is sorted out with
// selects at the end; the only branches are wave-uniform (no lane holds a match: skip the distance half; no lane is running: leave).
template <class LitLook, class DistLook>
__device__ __forceinline__ int codes(Bits& b, Out& w, const Code& lc, const Code& dc, LitLook&& lit, DistLook&& dist_sym) {
    InQ q;
    q.cur = b.cur; q.nxt = b.nxt; q.widx = b.widx;
    const uint4* last = b.blk_end - 1 > b.blk_first ? b.blk_end - 1 : b.blk_first;
    q.nx2 = load_global_block(clamp_block(b, last));
    ++b.blk;
    unsigned t0 = 0, t1 = 0;
    int held = 0;                                 // tokens of this lane waiting for the next synchronous point
    int st = -1;                                  // -1: this lane is still inside the block
    for (int it = 0;; ++it) {
        if ((it & 1) == 0) {                      // ---- the synchronous point (wave-uniform) ----
            const bool adv = q.widx >= 4;
            if (adv) {
                q.cur = q.nxt; q.nxt = q.nx2; q.widx -= 4;
                q.nx2 = load_global_block(clamp_block(b, last));
                ++b.blk;
            }
            if (held >= 1) w.tok[w.nt] = t0;
            if (held == 2) w.tok[w.nt + 1] = t1;
            w.nt += held;
            held = 0;
        }
        const bool run = st < 0;
        refill_q(b, q);                           // >= 33 bits: a literal/length code (15) and its extra bits (5)
        const int s = decode_p<kFixLCodes>(b, lc, run, lit);
        bool bad = s < 0;
        const bool is_m = s > 256, is_l = s >= 0 && s < 256, eob = s == 256;
        const int ms = s - 257;
        bad = bad || (is_m && ms >= 29);
        int lbase, lextra;
        len_code(min(max(ms, 0), 28), &lbase, &lextra);
        const bool mrun = run && is_m && !bad;
        const int len = lbase + (int)take_nr(b, mrun ? lextra : 0);
        long long dist = 0;
        if (__any(mrun)) {
            refill_q(b, q);                       // >= 33 bits again: a distance code (15) and its extra bits (13)
            const int ds = decode_p<32>(b, dc, mrun, dist_sym);
            bad = bad || (mrun && (ds < 0 || ds >= 30));
            int dbase, dextra;
            dist_code(min(max(ds, 0), 29), &dbase, &dextra);
            dist = dbase + (long long)take_nr(b, (mrun && !bad) ? dextra : 0);
        }
        const bool over = b.remaining < 0;        // (zeros or foreign bits were decoded: whatever came out does not count)
        const int n = is_m ? len : 1;
        int now = -1;
        if (bad) now = over ? INF_EINPUT : INF_ECODE;
        else if (over) now = INF_EINPUT;
        else if (eob) now = INF_OK;
        else if (is_m && dist > w.o) now = INF_EDIST;
        else if (w.o + n > w.len) now = INF_EOUTPUT;
        const bool emit = run && now < 0;
        const unsigned tk = is_l ? (unsigned)s : (0x80000000u | ((unsigned)len << 16) | (unsigned)(dist - 1));
        t0 = (emit && held == 0) ? tk : t0;
        t1 = (emit && held == 1) ? tk : t1;
        held += emit ? 1 : 0;
        w.o += emit ? n : 0;
        st = run ? now : st;
        if (!__any(st < 0)) break;
    }
    if (held >= 1) w.tok[w.nt] = t0;
    if (held == 2) w.tok[w.nt + 1] = t1;
    w.nt += held;
    // back to the block-at-a-time reader of the headers: it holds `cur`, `nxt` and the address of the block after them
    if (q.widx >= 4) { b.cur = q.nxt; b.nxt = q.nx2; b.widx = q.widx - 4; }
    else { b.cur = q.cur; b.nxt = q.nxt; b.widx = q.widx; --b.blk; }
    if (st == INF_EINPUT) b.over = true;
    return st;
}

template <int LPW>
__global__ void __launch_bounds__(kLanes) k_inflate_tokens(const unsigned char* comp, long long comp_len, const InfDesc* desc, long long n,
                                                        unsigned* tokens, long long* ntok, int* status, unsigned* adler, int zlib_wrapped) {
    typedef LdsT<LPW> Lds;
    __shared__ Lds L;
    const int lane = threadIdx.x;
    const long long idx = (long long)blockIdx.x * LPW + lane;
    if (lane >= LPW || idx >= n) return;
    const InfDesc d = desc[idx];
    Bits b;
    bits_init(b, comp + d.src_off, d.src_len, comp + comp_len);
    Out w;
    w.tok = tokens + d.tok_off; w.nt = 0; w.o = 0; w.len = d.dst_len;
    Code lc, dc;                   // the current block's two Huffman codes (limits / bases per length, in registers)
    Hi9 hi = Hi9{0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    // per-length counters of a block header: u16 [32][LPW] over the first 64 rows of lsym (dead while a header is being read)
    unsigned short (*cnt)[LPW] = reinterpret_cast<unsigned short (*)[LPW]>(&L.lsym[0][0]);
    auto lit = [&](int i) { return (int)L.lsym[i][lane] | (int)(hi_bit(hi, i) << 8); };
    auto dist_sym = [&](int i) { return (int)L.dsym[i][lane]; };
    int st = INF_OK;
    if (zlib_wrapped) {
        if (d.src_len < 6) st = INF_EHEADER;
        else {
            const unsigned cmf = take(b, 8), flg = take(b, 8);
            if ((cmf & 0x0f) != 8 || ((cmf << 8) | flg) % 31 != 0 || (flg & 0x20)) st = INF_EHEADER;
        }
    }
    int last = 0;
    while (st == INF_OK && !last) {
        last = (int)take(b, 1);
        const unsigned type = take(b, 2);
        if (b.over) { st = INF_EINPUT; break; }
        if (type == 0) {
            // stored block: skip to the byte boundary (the stream position is 8*src_len - remaining bits), LEN / ~LEN, raw bytes
            { const int pad = (int)(b.remaining & 7); refill(b); b.buf >>= pad; b.cnt -= pad; b.remaining -= pad; }
            const unsigned len = take(b, 16), nlen = take(b, 16);
            if (b.over || len != (~nlen & 0xffff)) { st = INF_ESTORED; break; }
            if (w.o + (long long)len > w.len) { st = INF_EOUTPUT; break; }
            for (unsigned k = 0; k < len; ++k) {
                const unsigned v = take(b, 8);
                if (b.over) break;
                emit(w, v);
            }
            if (b.over) { st = INF_EINPUT; break; }
        } else if (type == 1) {
            // fixed codes (RFC 1951 3.2.6), written out sorted by (length, symbol): 256..279 (7 bits), 0..143 and 280..287 (8), 144..255 (9);
            // 30 distance codes of 5 bits
            for (int k = 0; k < 32; ++k) cnt[k][lane] = 0;
            cnt[7][lane] = 24; cnt[8][lane] = 152; cnt[9][lane] = 112; cnt[16 + 5][lane] = kMaxDCodes;
            code_from_counts(lc, cnt, lane);
            code_from_counts(dc, cnt + 16, lane);
            for (int k = 0; k < 24; ++k) L.lsym[k][lane] = (unsigned char)k;                    // 256 + k
            for (int k = 0; k < 144; ++k) L.lsym[24 + k][lane] = (unsigned char)k;
            for (int k = 0; k < 8; ++k) L.lsym[168 + k][lane] = (unsigned char)(24 + k);         // 280 + k
            for (int k = 0; k < 112; ++k) L.lsym[176 + k][lane] = (unsigned char)(144 + k);
            for (int k = 0; k < kMaxDCodes; ++k) L.dsym[k][lane] = (unsigned char)k;
            hi = Hi9{0x00ffffffu, 0u, 0u, 0u, 0u, 0x0000ff00u, 0u, 0u, 0u};
            st = codes(b, w, lc, dc, lit, dist_sym);
        } else if (type == 2) {
            const int nlen = (int)take(b, 5) + 257, ndist = (int)take(b, 5) + 1, ncode = (int)take(b, 4) + 4;
            if (b.over) { st = INF_EINPUT; break; }
            if (nlen > kMaxLCodes || ndist > kMaxDCodes) { st = INF_ETABLE; break; }
            unsigned long long cl = 0;
            for (int i = 0; i < ncode; ++i) cl |= (unsigned long long)take(b, 3) << (3 * cl_order(i));
            if (b.over) { st = INF_EINPUT; break; }
            ClCode clc;
            if (cl_build(clc, cl) != INF_OK) { st = INF_ETABLE; break; }
            // pass A: count the codes per length (literal/length: cnt[0..15], distance: cnt[16..31])
            for (int k = 0; k < 32; ++k) cnt[k][lane] = 0;
            const Bits at_lengths = b;
            int eob_len = 0;
            st = walk_lengths(b, clc, nlen + ndist, [&](int i, int len) {
                cnt[(i < nlen ? 0 : 16) + len][lane]++;
                eob_len = i == 256 ? len : eob_len;
            });
            if (st != INF_OK) break;
            if (eob_len == 0) { st = INF_ETABLE; break; }                      // no end-of-block code
            int err = counts_left(cnt, nlen, lane);
            if (err < 0 || (err > 0 && nlen - cnt[0][lane] != 1)) { st = INF_ETABLE; break; }
            err = counts_left(cnt + 16, ndist, lane);
            if (err < 0 || (err > 0 && ndist - cnt[16][lane] != 1)) { st = INF_ETABLE; break; }
            code_from_counts(lc, cnt, lane);
            code_from_counts(dc, cnt + 16, lane);
            Offs ol, od;
            offs_from_counts(ol, cnt, lane);
            offs_from_counts(od, cnt + 16, lane);
            // pass B: the same bits again; every symbol goes to its place (the counters' rows of lsym are overwritten from here on)
            hi = Hi9{0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
            b = at_lengths;
            (void)walk_lengths(b, clc, nlen + ndist, [&](int i, int len) {
                if (len != 0) {
                    if (i < nlen) {
                        const int pos = offs_take(ol, len);
                        L.lsym[pos][lane] = (unsigned char)(i & 0xff);
                        if (i & 0x100) hi_set(hi, pos);
                    } else {
                        L.dsym[offs_take(od, len)][lane] = (unsigned char)(i - nlen);
                    }
                }
            });
            st = codes(b, w, lc, dc, lit, dist_sym);
        } else {
            st = INF_ECODE;
        }
    }
    if (st == INF_OK && w.o != w.len) st = INF_ESHORT;
    // zlib trailer (RFC 1950): the Adler-32 of the uncompressed data, big-endian, at the next byte boundary.  zlib itself — and with
    // it h5py's deflate filter — refuses a stream whose checksum does not match; k_lz_resolve holds the bytes and compares.
    unsigned expect = 1u;
    if (st == INF_OK && zlib_wrapped) {
        (void)take(b, (int)(b.remaining & 7));
        expect = 0;
        for (int k = 0; k < 4; ++k) expect = (expect << 8) | take(b, 8);
        if (b.over) st = INF_EINPUT;
    }
    adler[idx] = expect;
    ntok[idx] = st == INF_OK ? w.nt : 0;
    status[idx] = st;
}

// chunk -> frame placement: chunk `c` (chunk_bytes of raw data, C order over cdim[rank]) covers the box starting at
// coff[c][rank] of dataset ds[c]; elements inside the dataset's shape go to out[ds][...] (C order), float64 -> float32 when
// conv == 1.  One workgroup per chunk.
struct PlaceArgs {
    const unsigned char* raw; long long chunk_bytes;
    const int* ds; const int* coff;     // [n_chunks], [n_chunks][8]
    int rank; int shape[8]; int cdim[8]; int esz; int conv;
    unsigned char* out; long long out_elems;   // elements per dataset in the output
    unsigned celems;                           // elements per chunk
    int shuffle;                               // HDF5 shuffle filter under the deflate: byte b of element e is raw[b * celems + e]
};
// k_lz_resolve's workgroup IS one wavefront: the LDS executes a wavefront's operations in issue order, so what one lane wrote
// is visible to the next read of any lane — all that is needed is that the compiler keeps the order.  (__syncthreads() would
// also wait for every global load and store in flight: the prefetched tokens, the flushed words.)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// inclusive prefix sum over the 64 lanes on the DPP network (row shifts inside each row of 16, then the two row broadcasts):
// six VALU operations instead of six trips through the LDS crossbar
__device__ __forceinline__ int wave_scan_incl(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);     // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);     // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);     // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);     // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);     // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);     // row_bcast:31 into rows 2 and 3
    return v;
}

// ---- pass 2: tokens -> bytes, one wavefront per stream, the window of the output in LDS ---------------------------------------
// window: byte i of the output lives at ring[i & M].  A stream of up to 60 KB is kept whole (M = ~0, the LDS allocation is the
// stream length); a longer one goes round a 64 KB ring (M = 65535): bytes older than the 32 KB DEFLATE window are flushed to HBM in
// 8-byte words whenever a batch begins, so at most 32 KB + 7 + 64 * 258 bytes are live.  Per batch of 64 tokens: lane l owns token
// l; an inclusive wave scan of the token lengths gives every token its output position; literals are written at once; matches are
// then resolved one after the other (a match may copy what an earlier match of the same batch produced), each in one step by all
// 64 lanes.
__global__ void __launch_bounds__(kLanes) k_lz_resolve(const unsigned* tokens, const long long* ntok, const InfDesc* desc, long long n,
                                                       unsigned char* out, int* status, unsigned M, const PlaceArgs pa, int fused, const unsigned* adler, int check, unsigned dump_whole) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
    const long long idx = blockIdx.x;
    if (idx >= n || status[idx] != INF_OK) return;
    const int lane = threadIdx.x;
    const InfDesc d = desc[idx];
    const unsigned* tok = tokens + d.tok_off;
    const long long nt = ntok[idx];
    unsigned char* dst = out + d.dst_off;
    long long o = 0, flushed = 0;
    // Adler-32 of the output, block by block as the bytes become final: for a block of m bytes b_0..b_{m-1},
    //   s1' = s1 + sum b_j,   s2' = s2 + m s1 + sum (m - j) b_j        (mod 65521)
    // — both sums are plain sums over the block, so the lanes take 8 bytes each per step (v_sad_u8 / v_dot4 on the two words)
    // and one wave reduction per block gives every lane the same (s1, s2).
    unsigned s1 = 1u, s2 = 0u;
    long long summed = 0;                     // bytes [0, summed) are in (s1, s2)
    auto adler_to = [&](long long upto) {     // [summed, upto); summed is a multiple of 8
        if (!check || upto <= summed) return;
        const long long m = upto - summed;
        unsigned long long a = 0, wsum = 0;
        long long j = 8ll * lane;
        for (; j + 8 <= m; j += 8ll * kLanes) {
            const unsigned long long v = *reinterpret_cast<const unsigned long long*>(ring + ((unsigned)(summed + j) & M));
            const unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
            const unsigned sb = __builtin_amdgcn_sad_u8(lo, 0u, 0u) + __builtin_amdgcn_sad_u8(hi, 0u, 0u);
            const unsigned si = __builtin_amdgcn_udot4(lo, 0x03020100u, 0u, false) + __builtin_amdgcn_udot4(hi, 0x07060504u, 0u, false);
            a += sb;
            wsum += (unsigned long long)(m - j) * sb - si;
        }
        if (j < m) {                          // the lane that holds the ragged end (at most 7 bytes)
            for (long long k = j; k < m; ++k) {
                const unsigned bb = ring[(unsigned)(summed + k) & M];
                a += bb;
                wsum += (unsigned long long)(m - k) * bb;
            }
        }
#pragma unroll
        for (int sft = 32; sft >= 1; sft >>= 1) {
            a += (unsigned long long)__shfl_xor((unsigned)a, sft) | ((unsigned long long)__shfl_xor((unsigned)(a >> 32), sft) << 32);
            wsum += (unsigned long long)__shfl_xor((unsigned)wsum, sft) | ((unsigned long long)__shfl_xor((unsigned)(wsum >> 32), sft) << 32);
        }
        s2 = (unsigned)((s2 + (unsigned long long)(m % 65521) * s1 + wsum % 65521) % 65521);
        s1 = (unsigned)((s1 + a) % 65521);
        summed = upto;
    };
    auto flush_to = [&](long long upto) {     // [flushed, upto), both multiples of 8
        adler_to(upto);
        for (long long i = flushed + 8ll * lane; i < upto; i += 8ll * kLanes)
            *reinterpret_cast<unsigned long long*>(dst + i) = *reinterpret_cast<const unsigned long long*>(ring + ((unsigned)i & M));
        flushed = upto;
    };
    if (nt <= 0) {            // an empty stream has no bytes either: its Adler-32 is 1
        if (check && adler[idx] != 1u && lane == 0) status[idx] = INF_ECHECK;
        return;
    }
    unsigned tk_next = tok[min((long long)lane, nt - 1)];
    for (long long t0 = 0; t0 < nt; t0 += kLanes) {
        const bool valid = t0 + lane < nt;
        const unsigned tk = tk_next;
        const bool is_m = valid && (tk >> 31);
        const int len = is_m ? (int)((tk >> 16) & 0x1ff) : (valid ? 1 : 0);
        const int incl = wave_scan_incl(len);
        const int total = __builtin_amdgcn_readlane(incl, kLanes - 1);
        const long long pos = o + incl - len;
        // make room: everything older than the 32 KB window goes out.  Ring mode only: a whole-window stream (M = ~0) stays in LDS
        // until the end — on the fused path `dst` is not even a chunk-sized buffer (d_raw is 16 bytes there), and the final
        // adler_to(o) + placement / flush cover every byte
        if (M != 0xffffffffu && o - flushed > 32768 + 8) {
            wave_sync();
            flush_to((o - 32768) & ~7ll);
            wave_sync();
        }
        // the next batch's tokens: in flight while this batch is resolved (issued after the flush's stores, so that waiting
        // for it at the top of the loop waits for nothing else)
        tk_next = tok[min(t0 + kLanes + lane, nt - 1)];
        if (valid && !is_m) ring[(unsigned)pos & M] = (unsigned char)(tk & 0xff);
        wave_sync();
        // matches, one after the other (a match may copy what an earlier one of the batch produced), each in ONE step by the
        // whole wavefront: byte k of a match is byte (k mod distance) of the `distance` bytes in front of it, which are final —
        // so all (at most 258) bytes are independent; lane l takes bytes l, l + 64, ... : every read is issued before any write
        // what the serial loop needs of a match is worked out by its own lane, all 64 at once (the reciprocal above all), and
        // fetched with v_readlane: nothing but the copy itself is left on the one-match-after-the-other chain
        const int mdist_l = (int)(tk & 0x7fff) + 1;
        const float inv_l = 1.0f / (float)mdist_l;
        const unsigned to_l = (unsigned)pos;
        unsigned long long mm = __ballot(is_m);
        while (mm) {
            const int l = __ffsll((long long)mm) - 1;
            mm &= mm - 1;
            const int mlen = __builtin_amdgcn_readlane(len, l);
            const int mdist = __builtin_amdgcn_readlane(mdist_l, l);
            const unsigned to = (unsigned)__builtin_amdgcn_readlane((int)to_l, l);
            const unsigned from = to - (unsigned)mdist;
            // (k mod mdist by a float reciprocal + one correction step: exact for k < 320, mdist <= 32768; the reads of lanes
            // beyond the match fetch some byte of the ring and are dropped — no branch between the reads, so they overlap)
            const float inv = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, inv_l), l));
            // lanes beyond the match write to a dump byte (one select) instead of being masked out (a compare, an exec save /
            // restore and a branch per 64-byte block); a match that does not overlap itself (distance >= length: most matches
            // that are not runs) needs no folding at all
            // (ring: the batch ends at most 64 * 258 bytes after its start `o`, and nothing older than o - 32 776 is still
            // unflushed: o + 20 480 is neither)
            const unsigned dump = M == 0xffffffffu ? dump_whole : (((unsigned)o + 20480u) & M);
            auto copy = [&](auto nb, auto wrap_c) {
                constexpr int NB = decltype(nb)::value;
                constexpr bool WRAP = decltype(wrap_c)::value;
                unsigned char v[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int k = min(lane + kLanes * j, mlen - 1);
                    int r = k;
                    if (WRAP) {
                        r = k - (int)((float)k * inv) * mdist;
                        if (r < 0) r += mdist;
                        if (r >= mdist) r -= mdist;
                    }
                    v[j] = ring[(from + (unsigned)r) & M];
                }
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int k = lane + kLanes * j;
                    ring[k < mlen ? ((to + (unsigned)k) & M) : dump] = v[j];
                }
            };
            const std::true_type yes{};
            const std::false_type no{};
            if (mdist >= mlen) {
                if (mlen <= kLanes) copy(std::integral_constant<int, 1>{}, no);
                else if (mlen <= 2 * kLanes) copy(std::integral_constant<int, 2>{}, no);
                else copy(std::integral_constant<int, 5>{}, no);
            } else {
                if (mlen <= kLanes) copy(std::integral_constant<int, 1>{}, yes);
                else if (mlen <= 2 * kLanes) copy(std::integral_constant<int, 2>{}, yes);
                else copy(std::integral_constant<int, 5>{}, yes);
            }
            wave_sync();
        }
        o += total;
    }
    wave_sync();
    if (check) {
        adler_to(o);                          // (what has not been summed yet is still in the window, ragged end included)
        if (((s2 << 16) | s1) != adler[idx]) {
            if (lane == 0) status[idx] = INF_ECHECK;
            return;
        }
    }
    if (fused) {
        // the whole chunk sits in LDS (M = ~0): place it straight into its frame — float64 -> float32 on the way when asked —
        // instead of writing the raw bytes out for k_place_chunks to read back
        const int out_esz = pa.conv == 1 ? 4 : pa.esz;
        unsigned char* dd = pa.out + (long long)pa.ds[idx] * pa.out_elems * out_esz;
        const int* co = pa.coff + idx * 8;
        // element e of the chunk -> its coordinates (last dimension fastest).  Lane l takes e = l, l + 64, ...: the coordinates are
        // worked out once by division and then advanced by the mixed-radix digits of 64 with carries — no division per element.
        auto place = [&](auto rank_c) {
            constexpr int R = decltype(rank_c)::value;
            int cd[R], sh[R], c0[R], dg[R], ix[R];
            long long stride[R];
            unsigned step = kLanes, mine = (unsigned)lane;
            long long mul = 1;
#pragma unroll
            for (int dm = R - 1; dm >= 0; --dm) {
                cd[dm] = pa.cdim[dm]; sh[dm] = pa.shape[dm]; c0[dm] = co[dm];
                dg[dm] = (int)(step % (unsigned)cd[dm]); step /= (unsigned)cd[dm];
                ix[dm] = (int)(mine % (unsigned)cd[dm]); mine /= (unsigned)cd[dm];
                stride[dm] = mul; mul *= sh[dm];
            }
            for (unsigned e = lane; e < pa.celems; e += kLanes) {
                bool inside = true;
                long long oidx = 0;
#pragma unroll
                for (int dm = 0; dm < R; ++dm) {
                    const int g = c0[dm] + ix[dm];
                    inside = inside && g < sh[dm];
                    oidx += (long long)g * stride[dm];
                }
                if (inside) {
                    if (pa.conv == 1) {
                        double v;
                        if (pa.shuffle) {              // the 8 bytes of element e lie one in each byte plane
                            unsigned long long u = 0;
#pragma unroll
                            for (int k = 0; k < 8; ++k) u |= (unsigned long long)ring[(unsigned)k * pa.celems + e] << (8 * k);
                            v = __builtin_bit_cast(double, u);
                        } else {
                            v = *reinterpret_cast<const double*>(ring + 8ull * e);
                        }
                        reinterpret_cast<float*>(dd)[oidx] = (float)v;
                    } else if (pa.shuffle) {
                        for (int k = 0; k < pa.esz; ++k) dd[oidx * pa.esz + k] = ring[(unsigned)k * pa.celems + e];
                    } else {
                        for (int k = 0; k < pa.esz; ++k) dd[oidx * pa.esz + k] = ring[(unsigned long long)e * pa.esz + k];
                    }
                }
                int carry = 0;
#pragma unroll
                for (int dm = R - 1; dm >= 0; --dm) {
                    int v = ix[dm] + dg[dm] + carry;
                    carry = v >= cd[dm] ? 1 : 0;
                    ix[dm] = v - (carry ? cd[dm] : 0);
                }
            }
        };
        switch (pa.rank) {
            case 1: place(std::integral_constant<int, 1>{}); break;
            case 2: place(std::integral_constant<int, 2>{}); break;
            case 3: place(std::integral_constant<int, 3>{}); break;
            case 4: place(std::integral_constant<int, 4>{}); break;
            case 5: place(std::integral_constant<int, 5>{}); break;
            case 6: place(std::integral_constant<int, 6>{}); break;
            default: place(std::integral_constant<int, 7>{}); break;
        }
        return;
    }
    const long long whole = o & ~7ll;
    flush_to(whole);
    if (lane < (int)(o - whole)) dst[whole + lane] = ring[(unsigned)(whole + lane) & M];
}

__global__ void k_place_chunks(const PlaceArgs a, long long n_chunks) {
    const long long c = blockIdx.x;
    if (c >= n_chunks) return;
    long long celems = 1;
    for (int d = 0; d < a.rank; ++d) celems *= a.cdim[d];
    const unsigned char* src = a.raw + c * a.chunk_bytes;
    const int out_esz = a.conv == 1 ? 4 : a.esz;
    unsigned char* dst = a.out + (long long)a.ds[c] * a.out_elems * out_esz;
    for (long long e = threadIdx.x; e < celems; e += blockDim.x) {
        long long rem = e, oidx = 0;
        bool inside = true;
        long long mul = 1;
        // decompose e over the chunk dims (last fastest) and build the dataset index
        for (int d = a.rank - 1; d >= 0; --d) {
            const int i = (int)(rem % a.cdim[d]);
            rem /= a.cdim[d];
            const int g = a.coff[c * 8 + d] + i;
            if (g >= a.shape[d]) inside = false;
            oidx += (long long)g * mul;
            mul *= a.shape[d];
        }
        if (!inside) continue;
        if (a.conv == 1) {
            double v;
            memcpy(&v, src + e * 8, 8);
            reinterpret_cast<float*>(dst)[oidx] = (float)v;
        } else {
            for (int k = 0; k < a.esz; ++k) dst[oidx * a.esz + k] = src[e * a.esz + k];
        }
    }
}

// streams per wavefront (see LdsT)
inline void launch_tokens(hipStream_t stream, const unsigned char* comp, long long comp_len, const InfDesc* desc, long long n, unsigned* tok,
                          long long* ntok, int* st, unsigned* adler, int wrapped) {
    const char* env = getenv("TH_INFLATE_LPW");     // A/B and tests: 8, 16, 32 or 64 (read at every call)
    const int forced = env ? atoi(env) : 0;
    // measured (gzip float64 frames, 17 KB chunks, after the LDS diet): 64 streams per wavefront are the fastest or tied at every
    // batch size — 8 192 streams: 5.5 / 3.9 / 3.3 / 3.2 ms at LPW 8 / 16 / 32 / 64 (the chain of one stream is ~3.2 ms);
    // 32 768 streams: 6.5 / 3.7 / 3.7 / 3.2 ms; 131 072 streams: 3.9 ms at 64.  Fewer only when there are hardly any streams.
    const int lpw = forced ? forced : (n >= 512 ? 64 : (n >= 64 ? 16 : 8));
    if (lpw == 8)
        hipLaunchKernelGGL(k_inflate_tokens<8>, dim3((unsigned)((n + 7) / 8)), dim3(kLanes), 0, stream, comp, comp_len, desc, n, tok, ntok, st, adler, wrapped);
    else if (lpw == 32)
        hipLaunchKernelGGL(k_inflate_tokens<32>, dim3((unsigned)((n + 31) / 32)), dim3(kLanes), 0, stream, comp, comp_len, desc, n, tok, ntok, st, adler, wrapped);
    else if (lpw == 16)
        hipLaunchKernelGGL(k_inflate_tokens<16>, dim3((unsigned)((n + 15) / 16)), dim3(kLanes), 0, stream, comp, comp_len, desc, n, tok, ntok, st, adler, wrapped);
    else
        hipLaunchKernelGGL(k_inflate_tokens<64>, dim3((unsigned)((n + 63) / 64)), dim3(kLanes), 0, stream, comp, comp_len, desc, n, tok, ntok, st, adler, wrapped);
}

// LDS window of k_lz_resolve for streams of at most max_len bytes
// (a stream that fits is kept whole — its LDS is its length and positions are used as they are, mask ~0 —, so 17 KB chunks
// run 9 wavefronts per CU instead of the 5 a 32 KB power-of-two ring allows; longer streams go round a 64 KB ring)
struct RingGeom { unsigned lds, mask, spare; };      // spare: bytes behind the window (a whole stream gets a dump byte there)
inline RingGeom ring_geom(int64_t max_len) {
    if (max_len <= 61440) return {(unsigned)((std::max<int64_t>(max_len, 16) + 15) & ~15ll), 0xffffffffu, 16u};
    return {65536u, 65535u, 0u};
}

// upper bound of the decoder's token arena per device (inflate_place_device decodes in pieces that fit it)
constexpr size_t kTokArenaBytes = (size_t)8 << 30;

struct DevBuf {
    void* p = nullptr; size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return TH_OK;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        // 25 % of slack so that batches of slowly growing size do not re-allocate; the exact size when the slack does not fit
        for (const size_t want : {bytes + bytes / 4 + 4096, bytes}) {
            const hipError_t e = th_malloc_retry(&p, want);
            if (e == hipSuccess) { cap = want; return TH_OK; }
            p = nullptr;
            (void)hipGetLastError();
            if (e != hipErrorOutOfMemory) TH_FAIL(TH_EHIP, "th_malloc_retry(%zu): %s", want, hipGetErrorString(e));
        }
        TH_FAIL(TH_ENOMEM, "inflate: no device memory for a %zu-byte scratch buffer (decode fewer datasets per call)", bytes);
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
    }
};

}  // namespace

// ---- C ABI ------------------------------------------------------------------------------------------------------------
// th_inflate_many: n independent zlib (wrapped = 1) or raw deflate (wrapped = 0) streams, stream i = comp[src_off[i] ..
// + src_len[i]), inflated on `device` into out[dst_off[i] .. + dst_len[i]) (dst_off 8-byte aligned; dst_len = the exact
// uncompressed size).  comp / out are HOST buffers (copied through device memory).  status_out[i] = 0 or the reason stream
// i failed (1 input exhausted, 2 output overrun, 3 bad zlib header, 4 invalid code, 5 distance too far back, 6 bad code
// table, 7 bad stored block, 8 stream ended short of dst_len).  Returns TH_OK when every stream decoded, TH_EIO otherwise.
extern "C" int th_inflate_many(int device, const void* comp, int64_t comp_len, int64_t n, const int64_t* src_off, const int64_t* src_len,
                               const int64_t* dst_off, const int64_t* dst_len, void* out, int64_t out_len, int wrapped, int* status_out) {
    if (n < 0 || (n && (!comp || !src_off || !src_len || !dst_off || !dst_len || !out))) TH_FAIL(TH_EINVAL, "th_inflate_many: null argument");
    if (n == 0) return TH_OK;
    std::vector<InfDesc> desc((size_t)n);
    int64_t tok_total = 0, max_len = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (src_off[i] < 0 || src_len[i] < 0 || src_off[i] > comp_len - src_len[i] || dst_off[i] < 0 || dst_len[i] < 0 ||
            dst_off[i] > out_len - dst_len[i] || (dst_off[i] & 7))
            TH_FAIL(TH_EINVAL, "th_inflate_many: stream %lld lies outside its buffer (or its output is not 8-byte aligned)", (long long)i);
        desc[(size_t)i] = {src_off[i], src_len[i], dst_off[i], dst_len[i], tok_total};
        tok_total += dst_len[i];
        max_len = std::max<int64_t>(max_len, dst_len[i]);
    }
    HIP_TRY(hipSetDevice(device));
    unsigned char *d_comp = nullptr, *d_out = nullptr;
    InfDesc* d_desc = nullptr;
    int* d_st = nullptr;
    unsigned *d_tok = nullptr, *d_ad = nullptr;
    long long* d_nt = nullptr;
    int rc = TH_OK;
    auto fail = [&](hipError_t e, const char* what) { th_set_error("th_inflate_many: %s: %s", what, hipGetErrorString(e)); rc = TH_EHIP; };
    hipError_t e;
    if ((e = th_malloc_retry(&d_comp, (size_t)comp_len + 16)) != hipSuccess) fail(e, "hipMalloc");
    if (!rc && (e = th_malloc_retry(&d_out, (size_t)out_len + 16)) != hipSuccess) fail(e, "hipMalloc");
    if (!rc && (e = th_malloc_retry(&d_desc, (size_t)n * sizeof(InfDesc))) != hipSuccess) fail(e, "hipMalloc");
    if (!rc && (e = th_malloc_retry(&d_st, (size_t)n * sizeof(int))) != hipSuccess) fail(e, "hipMalloc");
    if (!rc && (e = th_malloc_retry(&d_tok, (size_t)(tok_total + 16) * sizeof(unsigned))) != hipSuccess) fail(e, "hipMalloc");
    if (!rc && (e = th_malloc_retry(&d_nt, (size_t)n * sizeof(long long))) != hipSuccess) fail(e, "hipMalloc");
    if (!rc && (e = th_malloc_retry(&d_ad, (size_t)n * sizeof(unsigned))) != hipSuccess) fail(e, "hipMalloc");
    if (!rc && (e = hipMemcpy(d_comp, comp, (size_t)comp_len, hipMemcpyHostToDevice)) != hipSuccess) fail(e, "copy in");
    if (!rc && (e = hipMemcpy(d_desc, desc.data(), (size_t)n * sizeof(InfDesc), hipMemcpyHostToDevice)) != hipSuccess) fail(e, "copy in");
    if (!rc && (e = hipMemset(d_st, 0xff, (size_t)n * sizeof(int))) != hipSuccess) fail(e, "memset");
    if (!rc) {
        launch_tokens(nullptr, d_comp, (long long)comp_len, d_desc, (long long)n, d_tok, d_nt, d_st, d_ad, wrapped);
        if ((e = hipGetLastError()) != hipSuccess) fail(e, "launch");
    }
    if (!rc) {
        const RingGeom rg = ring_geom(max_len);
        if ((e = hipFuncSetAttribute((const void*)k_lz_resolve, hipFuncAttributeMaxDynamicSharedMemorySize, 65536)) != hipSuccess) fail(e, "attribute");
        hipLaunchKernelGGL(k_lz_resolve, dim3((unsigned)n), dim3(kLanes), rg.lds + rg.spare, 0, d_tok, d_nt, d_desc, (long long)n, d_out, d_st, rg.mask, PlaceArgs{}, 0, d_ad, wrapped ? 1 : 0, rg.lds);
        if (!rc && (e = hipGetLastError()) != hipSuccess) fail(e, "launch");
    }
    if (!rc && (e = hipDeviceSynchronize()) != hipSuccess) fail(e, "kernel");
    if (!rc && (e = hipMemcpy(out, d_out, (size_t)out_len, hipMemcpyDeviceToHost)) != hipSuccess) fail(e, "copy out");
    std::vector<int> st((size_t)n, 0);
    if (!rc && (e = hipMemcpy(st.data(), d_st, (size_t)n * sizeof(int), hipMemcpyDeviceToHost)) != hipSuccess) fail(e, "copy out");
    for (void* p : {(void*)d_comp, (void*)d_out, (void*)d_desc, (void*)d_st, (void*)d_tok, (void*)d_nt, (void*)d_ad})
        if (p) (void)hipFree(p);
    if (rc) return rc;
    int64_t bad = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (status_out) status_out[i] = st[(size_t)i];
        if (st[(size_t)i] != 0) ++bad;
    }
    if (bad) TH_FAIL(TH_EIO, "th_inflate_many: %lld of %lld streams did not decode", (long long)bad, (long long)n);
    return TH_OK;
}

namespace {
struct Scratch {
    std::mutex mu;
    DevBuf d_comp, d_raw, d_desc, d_st, d_ds, d_coff, d_tok, d_nt, d_ad;
    void* h_st = nullptr; size_t h_st_cap = 0;
    hipStream_t own = nullptr;
    void release_device() {
        for (DevBuf* b : {&d_comp, &d_raw, &d_desc, &d_st, &d_ds, &d_coff, &d_tok, &d_nt, &d_ad}) b->release();
    }
};
constexpr int kMaxDevices = 16;         // one set per device (`predict.py --devices 0,1,...` decodes on the GPU that predicts)
Scratch sets[kMaxDevices];
}  // namespace

// th_h5_release_scratch: give back the decoder's device scratch of `device` (token arena: 5 bytes per uncompressed byte of the
// largest batch decoded so far) — predict.py calls it when a run ends; the next decode allocates again.
extern "C" int th_h5_release_scratch(int device) {
    if (device < 0 || device >= kMaxDevices) TH_FAIL(TH_EINVAL, "th_h5_release_scratch: device %d outside 0..%d", device, kMaxDevices - 1);
    Scratch* sc = &sets[device];
    std::unique_lock<std::mutex> lock(sc->mu);
    bool any = sc->h_st != nullptr;
    for (DevBuf* b : {&sc->d_comp, &sc->d_raw, &sc->d_desc, &sc->d_st, &sc->d_ds, &sc->d_coff, &sc->d_tok, &sc->d_nt, &sc->d_ad}) any = any || b->p;
    if (!any) return TH_OK;                 // nothing was ever decoded on this device: no HIP call at all
    HIP_TRY(hipSetDevice(device));
    sc->release_device();
    if (sc->h_st) (void)hipHostFree(sc->h_st);
    sc->h_st = nullptr; sc->h_st_cap = 0;
    return TH_OK;
}

// ---- HDF5 chunks -> device-resident frames (used by h5ingest.hip: th_h5_decode_device) -----------------------------------
// comp: host pointer to a byte span of the file that contains every chunk (chunk i at span offset src_off[i], csize[i]
// bytes); chunk i belongs to dataset ds[i] at element offsets coff[i][rank].  Everything is inflated into a scratch buffer
// and placed into d_out [n_datasets][shape...] (float32 when conv = 1 and the data is float64, the stored type otherwise).
int inflate_place_device(int device, hipStream_t stream, const void* span, int64_t span_len, int64_t n_chunks, const int64_t* src_off,
                         const int64_t* csize, const int* ds, const int* coff8, int rank, const int64_t* shape, const int64_t* chunk, int esz,
                         int conv, void* d_out, int64_t* n_bad, int shuffle) {
    // Scratch memory with its own (non-blocking) stream, one caller at a time per device.  (Measured with TWO sets per device and
    // two loader threads in predict.py, so that the pageable upload of one batch runs under the kernels of the other: each call then
    // took twice as long — the same 0.38 s for 40 k frames warm — and the first call of a process paid for a second 9 GB token
    // arena: 0.49 -> 1.09 s.  One set it is.)  `stream`: nullptr (the usual case) means "the set's own stream".
    if (device < 0 || device >= kMaxDevices) TH_FAIL(TH_EINVAL, "inflate: device %d outside 0..%d", device, kMaxDevices - 1);
    Scratch* sc = &sets[device];
    std::unique_lock<std::mutex> lock(sc->mu);
    HIP_TRY(hipSetDevice(device));
    if (!sc->own) HIP_TRY(hipStreamCreateWithFlags(&sc->own, hipStreamNonBlocking));
    if (!stream) stream = sc->own;
    DevBuf &d_comp = sc->d_comp, &d_raw = sc->d_raw, &d_desc = sc->d_desc, &d_st = sc->d_st, &d_ds = sc->d_ds, &d_coff = sc->d_coff,
           &d_tok = sc->d_tok, &d_nt = sc->d_nt, &d_ad = sc->d_ad;
    void*& h_st = sc->h_st;
    size_t& h_st_cap = sc->h_st_cap;
    int64_t chunk_bytes = esz;
    for (int d = 0; d < rank; ++d) chunk_bytes *= chunk[d];
    const int64_t cb8 = (chunk_bytes + 7) / 8 * 8;
    int rc;
    // a chunk that fits the LDS window whole is placed by k_lz_resolve itself: no raw bytes in HBM, no placement kernel
    const bool fused = ring_geom(chunk_bytes).mask == 0xffffffffu && chunk_bytes / esz < (1ll << 31);
    if (shuffle && !fused) TH_FAIL(TH_EUNSUP, "inflate: a shuffled chunk of %lld bytes does not fit the LDS window", (long long)chunk_bytes);
    // The token arena holds 4 bytes per UNCOMPRESSED byte (worst case: every byte a literal) — 7.3 GB for a 4096-frame call of
    // float64 frames, whose hipMalloc was most of a cold process's first call.  The two decode kernels therefore run over pieces of
    // the chunk list that need at most kTokArenaBytes of tokens (1152 float64 frames: 36 864 streams per launch, still 18 waves per
    // SIMD of work); the pieces queue on the stream back to back and reuse the arena.
    size_t arena = kTokArenaBytes;
    if (const char* e = getenv("TH_INFLATE_TOK_KB")) arena = (size_t)std::max(1, atoi(e)) << 10;    // tests: force many pieces (read at every call)
    const int64_t piece = std::max<int64_t>(1, std::min<int64_t>(n_chunks, (int64_t)(arena / sizeof(unsigned)) / std::max<int64_t>(chunk_bytes, 1)));
    if ((rc = d_comp.ensure((size_t)span_len + 16)) || (rc = d_raw.ensure(fused ? 16 : (size_t)(n_chunks * cb8) + 16)) ||
        (rc = d_desc.ensure((size_t)n_chunks * sizeof(InfDesc))) || (rc = d_st.ensure((size_t)n_chunks * sizeof(int))) ||
        (rc = d_ds.ensure((size_t)n_chunks * sizeof(int))) || (rc = d_coff.ensure((size_t)n_chunks * 8 * sizeof(int))) ||
        (rc = d_tok.ensure((size_t)(piece * chunk_bytes + 16) * sizeof(unsigned))) || (rc = d_nt.ensure((size_t)n_chunks * sizeof(long long))) ||
        (rc = d_ad.ensure((size_t)n_chunks * sizeof(unsigned)))) {
        // out of device memory (TH_ENOMEM) or a failed allocation: keep nothing — the caller retries with fewer datasets or reads
        // through the host, and the model's arenas / the pooled batch buffers get the memory back
        sc->release_device();
        return rc;
    }
    static const bool trace = getenv("TH_H5_TRACE") != nullptr;
    const auto t_begin = std::chrono::steady_clock::now();
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    std::vector<InfDesc> desc((size_t)n_chunks);
    for (int64_t i = 0; i < n_chunks; ++i) desc[(size_t)i] = {src_off[i], csize[i], i * cb8, chunk_bytes, (i % piece) * chunk_bytes};
    HIP_TRY(hipMemcpyAsync(d_comp.p, span, (size_t)span_len, hipMemcpyHostToDevice, stream));
    const double t_span = since();
    HIP_TRY(hipMemcpyAsync(d_desc.p, desc.data(), (size_t)n_chunks * sizeof(InfDesc), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(d_ds.p, ds, (size_t)n_chunks * sizeof(int), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(d_coff.p, coff8, (size_t)n_chunks * 8 * sizeof(int), hipMemcpyHostToDevice, stream));
    PlaceArgs a;
    a.raw = (const unsigned char*)d_raw.p; a.chunk_bytes = cb8; a.ds = (const int*)d_ds.p; a.coff = (const int*)d_coff.p;
    a.rank = rank; a.esz = esz; a.conv = conv; a.out = (unsigned char*)d_out; a.out_elems = 1;
    for (int d = 0; d < 8; ++d) { a.shape[d] = d < rank ? (int)shape[d] : 1; a.cdim[d] = d < rank ? (int)chunk[d] : 1; }
    for (int d = 0; d < rank; ++d) a.out_elems *= shape[d];
    a.celems = (unsigned)(chunk_bytes / esz);
    a.shuffle = shuffle;
    {
        const RingGeom rg = ring_geom(chunk_bytes);
        HIP_TRY(hipFuncSetAttribute((const void*)k_lz_resolve, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
        for (int64_t i0 = 0; i0 < n_chunks; i0 += piece) {       // both kernels index their per-chunk arrays from the piece's first chunk
            const int64_t cnt = std::min(piece, n_chunks - i0);
            launch_tokens(stream, (const unsigned char*)d_comp.p, (long long)span_len, (const InfDesc*)d_desc.p + i0, (long long)cnt, (unsigned*)d_tok.p,
                          (long long*)d_nt.p + i0, (int*)d_st.p + i0, (unsigned*)d_ad.p + i0, 1);
            HIP_TRY(hipGetLastError());
            PlaceArgs ap = a;
            ap.ds = a.ds + i0; ap.coff = a.coff + 8 * i0;
            hipLaunchKernelGGL(k_lz_resolve, dim3((unsigned)cnt), dim3(kLanes), rg.lds + rg.spare, stream, (const unsigned*)d_tok.p, (const long long*)d_nt.p + i0,
                               (const InfDesc*)d_desc.p + i0, (long long)cnt, (unsigned char*)d_raw.p, (int*)d_st.p + i0, rg.mask, ap, fused ? 1 : 0,
                               (const unsigned*)d_ad.p + i0, 1, rg.lds);
            HIP_TRY(hipGetLastError());
        }
    }
    if (!fused) {
        hipLaunchKernelGGL(k_place_chunks, dim3((unsigned)n_chunks), dim3(256), 0, stream, a, (long long)n_chunks);
        HIP_TRY(hipGetLastError());
    }
    if (h_st_cap < (size_t)n_chunks * sizeof(int)) {
        if (h_st) (void)hipHostFree(h_st);
        h_st = nullptr; h_st_cap = 0;
        HIP_TRY(hipHostMalloc(&h_st, (size_t)n_chunks * sizeof(int) * 2 + 4096, hipHostMallocDefault));
        h_st_cap = (size_t)n_chunks * sizeof(int) * 2 + 4096;
    }
    HIP_TRY(hipMemcpyAsync(h_st, d_st.p, (size_t)n_chunks * sizeof(int), hipMemcpyDeviceToHost, stream));
    const double t_enq = since();
    HIP_TRY(hipStreamSynchronize(stream));
    if (trace) fprintf(stderr, "[inflate] span upload returned after %.2f ms, launches enqueued after %.2f ms, stream drained after %.2f ms\n", t_span, t_enq, since());   // `span` / the descriptor vectors may go away now, and the statuses are needed
    int64_t bad = 0;
    for (int64_t i = 0; i < n_chunks; ++i)
        if (((const int*)h_st)[i] != 0) ++bad;
    if (n_bad) *n_bad = bad;
    return TH_OK;
}
