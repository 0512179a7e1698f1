// Host-side text formatter for the prediction writers.  No device code.
//
// The reference writes probabilities with np.savetxt's default format: every value as '%.18e', ',' between
// columns, '\n' after each row (design_utils/utils.py:768-771 for the float16-rounded <model>.csv,
// predict.py:145-146 for the full-precision rotamer matrix).  At GPU rates that Python-level formatting is
// the bottleneck (1 M x 338 values = 8.5 GB of text), so th_format_csv produces the same bytes natively:
//   * float16 input (the caller rounds with NumPy, exactly like the reference's np.array(..., dtype=float16)):
//     a 65 536-entry table of preformatted strings, one memcpy per value;
//   * fp32 / fp64 output: snprintf("%.18e") (glibc and CPython both round correctly, so the digits agree),
//     rows split over host threads.
// NaN is written as 'nan' whatever its sign bit, like Python's % operator.
#include "common.h"

#include <hip/hip_fp16.h>

#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace {

// '%.18e' of v into dst (room for 32 bytes); returns the length.  std::to_chars(scientific, 18) is libstdc++'s
// Ryu-printf: correctly rounded like printf (and CPython's '%' operator), several times faster than glibc's
// multi-precision snprintf; it writes the same "d.ddde[+-]XX" shape (at least two exponent digits).
inline int fmt_e18(double v, char* dst) {
    if (std::isnan(v)) { std::memcpy(dst, "nan", 3); return 3; }
    if (std::isinf(v)) { const int n = v < 0 ? 4 : 3; std::memcpy(dst, v < 0 ? "-inf" : "inf", n); return n; }
    const std::to_chars_result r = std::to_chars(dst, dst + 32, v, std::chars_format::scientific, 18);
    if (r.ec != std::errc()) return snprintf(dst, 32, "%.18e", v);
    return (int)(r.ptr - dst);
}

float half_bits_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t)(h >> 15) << 31;
    uint32_t exp = (h >> 10) & 0x1f, man = h & 0x3ff, bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {  // subnormal: normalise
            int e = -1;
            do { ++e; man <<= 1; } while (!(man & 0x400));
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((man & 0x3ff) << 13);
        }
    } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

struct HalfTable {
    std::vector<char> text;     // 65536 x 32 bytes
    std::vector<uint8_t> len;
};
const HalfTable& half_table() {
    static HalfTable t;
    static std::once_flag once;
    std::call_once(once, [] {
        t.text.assign(65536 * 32, 0);
        t.len.assign(65536, 0);
        for (int h = 0; h < 65536; ++h) t.len[h] = (uint8_t)fmt_e18((double)half_bits_to_float((uint16_t)h), &t.text[(size_t)h * 32]);
    });
    return t;
}

inline double load_value(const void* data, int dtype, int64_t i) {
    switch (dtype) {
        case TH_F32: return (double)((const float*)data)[i];
        case TH_F64: return ((const double*)data)[i];
        default: return (double)half_bits_to_float(((const uint16_t*)data)[i]);
    }
}

}  // namespace

extern "C" int64_t th_format_csv(const void* data, int dtype, int64_t n, int64_t k, char* out, int64_t cap) {
    if (!data || n < 0 || k <= 0 || (!out && cap > 0)) { th_set_error("th_format_csv: bad argument"); return TH_EINVAL; }
    if (dtype != TH_F32 && dtype != TH_F64 && dtype != TH_F16) { th_set_error("th_format_csv: dtype must be f32, f64 or f16"); return TH_EINVAL; }
    if (n == 0) return 0;
    const bool as_half = dtype == TH_F16;
    const HalfTable* tab = as_half ? &half_table() : nullptr;
    unsigned hw = (unsigned)th_usable_cpus();
    const int64_t work = n * k;
    int nthreads = (int)std::min<int64_t>(hw ? std::min(hw, 32u) : 4, std::max<int64_t>(1, work / (as_half ? 200000 : 20000)));
    nthreads = std::max(1, std::min<int>(nthreads, (int)n));
    // every thread formats its rows straight into `out` at the worst-case offset of its first row (28 bytes per
    // value); the pieces are then closed up front to back — no intermediate buffers
    const int64_t per_row = k * 28;
    if (cap < n * per_row) { th_set_error("th_format_csv: buffer of %lld bytes, need %lld (28 per value)", (long long)cap, (long long)(n * per_row)); return TH_EINVAL; }
    std::vector<int64_t> len(nthreads, 0);
    auto run = [&](int t) {
        const int64_t r0 = n * t / nthreads, r1 = n * (t + 1) / nthreads;
        char* const base = out + r0 * per_row;
        char* p = base;
        for (int64_t r = r0; r < r1; ++r) {
            for (int64_t c = 0; c < k; ++c) {
                const int64_t i = r * k + c;
                if (as_half) {
                    const uint16_t h = ((const uint16_t*)data)[i];
                    std::memcpy(p, &tab->text[(size_t)h * 32], 28);   // fixed-size copy (entries are <= 25 chars), advance by the real length
                    p += tab->len[h];
                } else {
                    p += fmt_e18(load_value(data, dtype, i), p);
                }
                *p++ = (c + 1 < k) ? ',' : '\n';
            }
        }
        len[t] = p - base;
    };
    if (nthreads == 1) run(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t) th.emplace_back(run, t);
        for (auto& x : th) x.join();
    }
    int64_t total = 0;
    for (int t = 0; t < nthreads; ++t) {
        const int64_t r0 = n * t / nthreads;
        if (total != r0 * per_row) std::memmove(out + total, out + r0 * per_row, (size_t)len[t]);
        total += len[t];
    }
    return total;
}

// ---- argmax -> residue letters: replaces the per-residue Python loop of extract_sequence_from_pred_matrix —
// design_utils/utils.py:659 (max_idx = np.argmax(prediction_matrix, axis=1)) and :689-692 (res_dic[max_idx[idx]] appended
// to the chain's string one residue at a time).  np.argmax semantics: the FIRST maximum, and a NaN counts as the maximum
// (the first NaN of a row wins).  letters_out[i] = col_letters[argmax_i]; idx_out (optional) receives the indices.
extern "C" int th_argmax_letters(const void* matrix, int dtype, int64_t n, int64_t k, const char* col_letters, char* letters_out,
                                 int32_t* idx_out) {
    if (!matrix || n < 0 || k <= 0 || (!letters_out && !idx_out) || (letters_out && !col_letters))
        TH_FAIL(TH_EINVAL, "th_argmax_letters: bad argument");
    if (dtype != TH_F32 && dtype != TH_F64 && dtype != TH_F16) TH_FAIL(TH_EINVAL, "th_argmax_letters: dtype must be f32, f64 or f16");
    if (k > 0x7fffffff) TH_FAIL(TH_EINVAL, "th_argmax_letters: too many columns");
    if (n == 0) return TH_OK;
    // float16 -> float through a 64 Ki table (exact), so one code path compares floats / doubles
    static std::vector<float> half_lut;
    static std::once_flag once;
    if (dtype == TH_F16) std::call_once(once, [] { half_lut.resize(65536); for (int h = 0; h < 65536; ++h) half_lut[h] = half_bits_to_float((uint16_t)h); });
    const int hw = th_usable_cpus();
    int nthreads = (int)std::min<int64_t>(std::min(hw > 0 ? hw : 4, 32), std::max<int64_t>(1, n * k / 400000));
    nthreads = std::max(1, std::min<int>(nthreads, (int)n));
    auto run = [&](int t) {
        const int64_t r0 = n * t / nthreads, r1 = n * (t + 1) / nthreads;
        for (int64_t r = r0; r < r1; ++r) {
            int64_t best = 0;
            if (dtype == TH_F64) {
                const double* row = (const double*)matrix + r * k;
                double bv = row[0];
                if (!std::isnan(bv))
                    for (int64_t c = 1; c < k; ++c) { const double v = row[c]; if (std::isnan(v)) { best = c; break; } if (v > bv) { bv = v; best = c; } }
            } else {
                float bv = dtype == TH_F32 ? ((const float*)matrix)[r * k] : half_lut[((const uint16_t*)matrix)[r * k]];
                if (!std::isnan(bv))
                    for (int64_t c = 1; c < k; ++c) {
                        const float v = dtype == TH_F32 ? ((const float*)matrix)[r * k + c] : half_lut[((const uint16_t*)matrix)[r * k + c]];
                        if (std::isnan(v)) { best = c; break; }
                        if (v > bv) { bv = v; best = c; }
                    }
            }
            if (letters_out) letters_out[r] = col_letters[best];
            if (idx_out) idx_out[r] = (int32_t)best;
        }
    };
    if (nthreads == 1) run(0);
    else {
        std::vector<std::thread> th;
        for (int t = 0; t < nthreads; ++t) th.emplace_back(run, t);
        for (auto& x : th) x.join();
    }
    return TH_OK;
}

// ---- dataset-map text -> fixed-width string table: replaces np.genfromtxt(datasetmap.txt, delimiter=",", dtype=str) —
// predict.py:99 (0.2 s per 100 k rows in NumPy's Python tokenizer; the map is on the critical path of every run).
// Two passes over plain ASCII text with `cols` delimiter-separated fields per line: th_csv_shape validates and sizes
// (TH_EUNSUP for anything np.genfromtxt would treat specially — '#' comments, '\r', quotes, non-ASCII bytes, blank or
// ragged lines, leading / trailing blanks of a field — the caller then uses NumPy), th_csv_fill writes the fields as
// UCS-4 code units into out[rows][cols][width] (zero padded): the memory layout of a NumPy '<U{width}' array.
extern "C" int th_csv_shape(const char* text, int64_t len, char delim, int64_t* rows_out, int* cols_out, int* width_out) {
    if (!text || len < 0 || !rows_out || !cols_out || !width_out) TH_FAIL(TH_EINVAL, "th_csv_shape: null argument");
    int64_t rows = 0;
    int cols = -1, width = 0, c = 0, w = 0;
    bool any = false;
    for (int64_t i = 0; i <= len; ++i) {
        const bool end = i == len;
        const unsigned char ch = end ? '\n' : (unsigned char)text[i];
        if (end && !any) break;                       // the file ended with its last newline
        if (ch == '\n' || ch == (unsigned char)delim) {
            if (w == 0) TH_FAIL(TH_EUNSUP, "th_csv_shape: empty field or line");
            const unsigned char first = (unsigned char)text[i - w], last = (unsigned char)text[i - 1];
            if (first == ' ' || first == '\t' || last == ' ' || last == '\t') TH_FAIL(TH_EUNSUP, "th_csv_shape: blank-padded field");
            width = std::max(width, w);
            w = 0;
            ++c;
            if (ch == '\n') {
                if (cols < 0) cols = c;
                else if (c != cols) TH_FAIL(TH_EUNSUP, "th_csv_shape: ragged line %lld", (long long)rows);
                c = 0;
                ++rows;
                any = false;
            }
            continue;
        }
        if (ch >= 0x80 || ch == '#' || ch == '\r' || ch == '"' || ch == '\'' || ch == 0) TH_FAIL(TH_EUNSUP, "th_csv_shape: special byte 0x%02x", ch);
        ++w;
        any = true;
    }
    if (rows == 0) TH_FAIL(TH_EUNSUP, "th_csv_shape: no rows");
    *rows_out = rows; *cols_out = cols; *width_out = width;
    return TH_OK;
}

extern "C" int th_csv_fill(const char* text, int64_t len, char delim, int64_t rows, int cols, int width, uint32_t* out) {
    if (!text || !out || rows <= 0 || cols <= 0 || width <= 0) TH_FAIL(TH_EINVAL, "th_csv_fill: bad argument");
    std::memset(out, 0, (size_t)rows * cols * width * sizeof(uint32_t));
    int64_t r = 0;
    int c = 0, w = 0;
    for (int64_t i = 0; i < len && r < rows; ++i) {
        const unsigned char ch = (unsigned char)text[i];
        if (ch == '\n') { ++r; c = 0; w = 0; continue; }
        if (ch == (unsigned char)delim) { ++c; w = 0; continue; }
        if (c >= cols || w >= width) TH_FAIL(TH_EINVAL, "th_csv_fill: text does not match the shape th_csv_shape reported");
        out[((size_t)r * cols + c) * width + w++] = ch;
    }
    return TH_OK;
}
