// th_fmt_e18_f32: '%.18e' of a float32, for host code (th_format_csv) and device code (k_format_csv_f32) alike.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define TH_HD __host__ __device__
#else
#define TH_HD
#endif

// '%.18e' of a float32 (widened exactly to double first, as NumPy hands np.float32 values to Python's '%' operator) WITHOUT the
// general double formatter, in a form a GPU lane can run: the rotamer matrix of predict.py:145-146 is 338 float32 values per residue
// (1 GB of text per 125 k residues), and with Ryu-printf on host threads the writer thread, not the GPU, set the pace of
// `predict.py --predict_rotamers` (0.82 s of formatting / appending against 0.49 s of GPU work).  A float32 below 2^24 is
// m * 2^-k with a 24-bit m and 0 <= k <= 149: its integer part is m >> k and its fraction a k-bit fixed-point number whose decimal
// digits come out nine at a time by multiplying with 10^9 (a handful of 32 x 32 -> 64 multiply-adds over at most five limbs, no
// division, exact).  The expansion ends after k digits, so the 19 significant digits are rounded on the EXACT value: to nearest,
// ties to even, like glibc's printf and CPython's dtoa (ties do occur: odd m * 2^-24 between 1e-5 and 1e-4 has exactly 20 digits).
// Integers >= 2^24, NaN and infinities are declined (-1): the caller formats them with the general double formatter.  Checked against Python's '%.18e' on random bit patterns, every
// exponent, subnormals and constructed ties (tests/test_textio.py).
TH_HD inline char* th_put9(uint32_t b, char* p) {              // nine digits of b < 10^9, leading zeros included
    const uint32_t hi = b / 10000u, lo = b % 10000u;  // hi < 10^5
    const uint32_t h1 = hi / 10u;                     // < 10^4
    p[0] = (char)('0' + h1 / 1000u); p[1] = (char)('0' + h1 / 100u % 10u); p[2] = (char)('0' + h1 / 10u % 10u); p[3] = (char)('0' + h1 % 10u);
    p[4] = (char)('0' + hi % 10u);
    p[5] = (char)('0' + lo / 1000u); p[6] = (char)('0' + lo / 100u % 10u); p[7] = (char)('0' + lo / 10u % 10u); p[8] = (char)('0' + lo % 10u);
    return p + 9;
}
TH_HD inline int th_fmt_e18_f32(float f, char* dst) {
    const uint32_t bits = __builtin_bit_cast(uint32_t, f);
    const uint32_t ex = (bits >> 23) & 0xffu, man = bits & 0x7fffffu;
    if (ex > 150u) return -1;                                               // NaN / inf; integers >= 2^24 (k < 0): the caller's general formatter
    char* p = dst;
    if (bits >> 31) *p++ = '-';
    if (ex == 0 && man == 0) {
        const char* z = "0.000000000000000000e+00";
        for (int i = 0; i < 24; ++i) p[i] = z[i];
        return (int)(p + 24 - dst);
    }
    uint32_t m = ex ? (man | 0x800000u) : man;
    int k = ex ? 150 - (int)ex : 149;                                       // value = m * 2^-k, 0 <= k <= 149
    const int tz0 = __builtin_ctz(m), tz = tz0 < k ? tz0 : k;
    m >>= tz; k -= tz;
    char dig[48];                                                           // significant digits, first one non-zero
    int nd = 0, e10 = 0;
    const uint32_t ip = k < 24 ? (m >> k) : 0u;                             // integer part (< 2^24: at most 8 digits)
    if (ip) {
        char t[9];
        th_put9(ip, t);
        int z = 0;
        while (t[z] == '0') ++z;
        nd = 9 - z;
        for (int i = 0; i < nd; ++i) dig[i] = t[z + i];
        e10 = nd - 1;
    }
    // the fraction, binary point moved up to a limb boundary: L limbs (little endian), K = 32 L fraction bits
    uint32_t limb[5] = {0, 0, 0, 0, 0};
    const int L = (k + 31) >> 5;
    bool rest = false;                                                      // fraction bits left
    if (k > 0) {
        const uint32_t fr = k < 24 ? (m & ((1u << k) - 1u)) : m;
        if (fr) {
            // fr / 2^k = (fr << sh) / 2^(32 L) with sh = 32 L - k in 0..31: the shifted fraction is below 2^55, i.e. it lies in the
            // two lowest limbs however many limbs of leading zero bits the number has above it
            const uint64_t v = (uint64_t)fr << (32 * L - k);
            limb[0] = (uint32_t)v;
            if (L > 1) limb[1] = (uint32_t)(v >> 32);
            rest = true;
        }
    }
    int lo = 0;                                                             // limbs below lo are zero (and stay zero)
    int zero_blocks = 0;
    while (nd < 20 && rest) {
        uint64_t carry = 0;
        while (lo < L && limb[lo] == 0) ++lo;
        for (int i = lo; i < L; ++i) {
            const uint64_t t = (uint64_t)limb[i] * 1000000000u + carry;
            limb[i] = (uint32_t)t;
            carry = t >> 32;
        }
        const uint32_t block = (uint32_t)carry;                             // the next nine decimal digits
        rest = false;
        for (int i = lo; i < L; ++i) rest = rest || limb[i];
        if (nd == 0) {
            if (block == 0) { ++zero_blocks; continue; }
            char t[9];
            th_put9(block, t);
            int z = 0;
            while (t[z] == '0') ++z;
            nd = 9 - z;
            for (int i = 0; i < nd; ++i) dig[i] = t[z + i];
            e10 = -(9 * zero_blocks + z + 1);
        } else {
            th_put9(block, dig + nd);
            nd += 9;
        }
    }
    bool up = false;
    if (nd >= 20) {
        bool sticky = rest;
        for (int i = 20; i < nd; ++i) sticky = sticky || dig[i] != '0';
        up = dig[19] > '5' || (dig[19] == '5' && (sticky || ((dig[18] - '0') & 1)));
    } else {
        for (int i = nd; i < 19; ++i) dig[i] = '0';                         // the expansion ended: exact
    }
    if (up) {
        int i = 18;
        while (i >= 0 && dig[i] == '9') dig[i--] = '0';
        if (i >= 0) ++dig[i];
        else { dig[0] = '1'; ++e10; }                                       // 9.99…9 -> 1.00…0e+1
    }
    *p++ = dig[0];
    *p++ = '.';
    for (int i = 1; i <= 18; ++i) *p++ = dig[i];
    *p++ = 'e';
    *p++ = e10 < 0 ? '-' : '+';
    const int ae = e10 < 0 ? -e10 : e10;                                    // <= 45: two digits
    *p++ = (char)('0' + ae / 10);
    *p++ = (char)('0' + ae % 10);
    return (int)(p - dst);
}

