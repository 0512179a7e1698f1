// Text formatter for the prediction writers: host code, plus one device kernel for the float32 matrix
// (th_format_csv_device, at the end of the file).
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
#include "fmt_e18_f32.h"

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

// float32 values: th_fmt_e18_f32 (fmt_e18_f32.h, shared with the device kernel below); what it declines goes through fmt_e18
inline int fmt_e18_f32(float f, char* dst) {
    const int n = th_fmt_e18_f32(f, dst);
    return n > 0 ? n : fmt_e18((double)f, dst);
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
                    p += dtype == TH_F32 ? fmt_e18_f32(((const float*)data)[i], p) : fmt_e18(load_value(data, dtype, i), p);
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

// ---- th_format_csv_device: the float32 matrix formatted ON THE GPU -------------------------------------------------------------
// Replaces np.savetxt(f, y_pred_batch, delimiter=",") of predict.py:145-146 for the full-precision rotamer matrix.  Every finite,
// non-negative float32 below 2^24 — every probability — is exactly 24 characters in '%.18e' form, so value i owns bytes
// [25 i, 25 i + 25) of the text (its separator included): no prefix sum, no compaction.  One lane per value (th_fmt_e18_f32, the
// same function the host path calls), a workgroup's 6 400 bytes assembled in LDS and stored as whole dwords.  A value that does not
// have the fixed form (negative, NaN, infinite, >= 2^24) raises a flag and the call returns TH_EUNSUP: the caller formats that
// block with th_format_csv.  HBM-bound byte work in principle (4 bytes in, 25 out per value); in practice the digit arrays live in
// scratch memory and the kernel takes ~0.1 ms per 338 000 values, against 2.2 ms for 13 host threads.
namespace {

constexpr int kFmtThreads = 256, kFmtBytes = 25;

__global__ void __launch_bounds__(kFmtThreads) k_format_csv_f32(const float* __restrict__ x, long long total, int k, char* __restrict__ out,
                                                                int* __restrict__ flag) {
    __shared__ __attribute__((aligned(16))) char sh[kFmtThreads * kFmtBytes];
    const long long i = (long long)blockIdx.x * kFmtThreads + threadIdx.x;
    if (i < total) {
        char t[32];
        const int len = th_fmt_e18_f32(x[i], t);
        char* d = sh + threadIdx.x * kFmtBytes;
        if (len == 24) {
            for (int j = 0; j < 24; ++j) d[j] = t[j];
        } else {
            atomicOr(flag, 1);
            for (int j = 0; j < 24; ++j) d[j] = '?';
        }
        d[24] = (i % k == k - 1) ? '\n' : ',';
    }
    __syncthreads();
    const long long base = (long long)blockIdx.x * (kFmtThreads * kFmtBytes);       // a multiple of 4
    const long long left = total * kFmtBytes - base;
    const int nbytes = (int)(left < (long long)(kFmtThreads * kFmtBytes) ? left : (long long)(kFmtThreads * kFmtBytes));
    const uint32_t* s4 = (const uint32_t*)sh;
    uint32_t* o4 = (uint32_t*)(out + base);
    for (int j = threadIdx.x; j < (nbytes >> 2); j += kFmtThreads) o4[j] = s4[j];
    for (int j = (nbytes & ~3) + threadIdx.x; j < nbytes; j += kFmtThreads) out[base + j] = sh[j];
}

struct FmtScratch {
    std::mutex mu;
    int device = -1;
    hipStream_t stream = nullptr;
    float* d_in = nullptr;
    char* d_out = nullptr;
    int* d_flag = nullptr;
    size_t cap_values = 0;
    void release() {
        if (device < 0) return;
        (void)hipSetDevice(device);
        if (stream) { (void)hipStreamSynchronize(stream); (void)hipStreamDestroy(stream); }
        (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(d_flag);
        stream = nullptr; d_in = nullptr; d_out = nullptr; d_flag = nullptr; cap_values = 0; device = -1;
    }
};
FmtScratch& fmt_scratch() {
    static FmtScratch* s = new FmtScratch;      // never destroyed: no HIP calls from static destructors
    return *s;
}

}  // namespace

extern "C" int64_t th_format_csv_device(int device, const float* rows, int64_t n, int64_t k, char* out, int64_t cap) {
    if (!rows || n < 0 || k <= 0 || k > 0x7fffffff || (!out && cap > 0)) { th_set_error("th_format_csv_device: bad argument"); return TH_EINVAL; }
    if (n == 0) return 0;
    const int64_t total = n * k, bytes = total * kFmtBytes;
    if (cap < bytes) { th_set_error("th_format_csv_device: buffer of %lld bytes, need %lld (25 per value)", (long long)cap, (long long)bytes); return TH_EINVAL; }
    FmtScratch& S = fmt_scratch();
    std::lock_guard<std::mutex> lock(S.mu);
    if (S.device >= 0 && S.device != device) S.release();
    HIP_TRY(hipSetDevice(device));
    if (S.device < 0) {
        S.device = device;
        // highest priority: the model's kernels keep every CU busy, and a formatter launch that queues behind them makes the
        // writer thread wait for milliseconds per group; with priority its few hundred workgroups take the next free slots
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        hipError_t e = hipStreamCreateWithPriority(&S.stream, hipStreamNonBlocking, greatest);
        if (e == hipSuccess) e = th_malloc_retry((void**)&S.d_flag, sizeof(int));
        if (e != hipSuccess) { S.release(); th_set_error("th_format_csv_device: %s", hipGetErrorString(e)); return TH_EHIP; }
    }
    if (S.cap_values < (size_t)total) {
        (void)hipFree(S.d_in); (void)hipFree(S.d_out);
        S.d_in = nullptr; S.d_out = nullptr; S.cap_values = 0;
        const size_t want = (size_t)total + (size_t)total / 4;
        hipError_t e = th_malloc_retry((void**)&S.d_in, want * sizeof(float));
        if (e == hipSuccess) e = th_malloc_retry((void**)&S.d_out, want * kFmtBytes + 16);
        if (e != hipSuccess) {
            (void)hipFree(S.d_in); S.d_in = nullptr;
            (void)hipGetLastError();
            th_set_error("th_format_csv_device: hipMalloc: %s", hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? TH_ENOMEM : TH_EHIP;
        }
        S.cap_values = want;
    }
    int flag = 0;
    HIP_TRY(hipMemsetAsync(S.d_flag, 0, sizeof(int), S.stream));
    HIP_TRY(hipMemcpyAsync(S.d_in, rows, (size_t)total * sizeof(float), hipMemcpyHostToDevice, S.stream));
    hipLaunchKernelGGL(k_format_csv_f32, dim3((unsigned)((total + kFmtThreads - 1) / kFmtThreads)), dim3(kFmtThreads), 0, S.stream, S.d_in,
                       (long long)total, (int)k, S.d_out, S.d_flag);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(&flag, S.d_flag, sizeof(int), hipMemcpyDeviceToHost, S.stream));
    HIP_TRY(hipStreamSynchronize(S.stream));
    if (flag) { th_set_error("th_format_csv_device: a value without the fixed 24-character form (negative, NaN, infinite or >= 2^24)"); return TH_EUNSUP; }
    HIP_TRY(hipMemcpyAsync(out, S.d_out, (size_t)bytes, hipMemcpyDeviceToHost, S.stream));
    HIP_TRY(hipStreamSynchronize(S.stream));
    return bytes;
}

extern "C" int th_format_csv_device_release(void) {
    FmtScratch& S = fmt_scratch();
    std::lock_guard<std::mutex> lock(S.mu);
    S.release();
    return TH_OK;
}
