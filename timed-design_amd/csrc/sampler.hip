// Monte-Carlo sequence sampler on gfx950 — replaces the NumPy kernel of
// design_utils/sampling_utils.py in the reference:
//   apply_temp_to_probs      (sampling_utils.py:139-161)  q = p**(1/t); q /= q.sum(axis=1)
//   random_choice_prob_index (sampling_utils.py:81-82)    r ~ U; idx = (q.cumsum(1) > r).argmax(1)
// Everything is fp64 like the reference.  Bit-exactness rules that the kernels restate:
//   * cumsum is a strictly sequential left-to-right fp64 running sum (what np.cumsum does);
//   * "(cumsum > r).argmax()" = FIRST index whose running sum exceeds r, and 0 when none does
//     (rows that sum to < r after float16 rounding, NaN rows) — SURVEY.md Appendix C-5;
//   * the row normaliser uses NumPy's pairwise summation order (8 strided partials, blocks of
//     128, recursive halving) so that q is bit-identical to NumPy whenever the power itself is
//     exact: NumPy maps x**2.0 to square and x**0.5 to sqrt (t = 0.5, t = 2), both IEEE-exact
//     here too.  For other exponents device pow() may differ from libm in the last ulp.
// Uniforms come from one of three sources (include/timed_hip.h TH_RNG_*): caller supplied,
// rocRAND Philox4x32-10 (one subsequence per draw), or an on-device MT19937 that replays
// np.random.seed(seed); np.random.rand(...) exactly (init_genrand + genrand_res53).
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include <rocrand/rocrand_kernel.h>

namespace {

// ---- NumPy pairwise sum (numpy/core/src/umath/loops_utils.h.src, DOUBLE_pairwise_sum) ---------
// recursion is unrolled at compile time (DEPTH halvings cover n <= 128 * 2^DEPTH classes)
template <int DEPTH>
__device__ double np_pairwise_sum(const double* a, int n) {
    if (n < 8) {
        double res = 0.;
        for (int i = 0; i < n; ++i) res += a[i];
        return res;
    } else if (n <= 128 || DEPTH == 0) {
        double r[8];
        for (int k = 0; k < 8; ++k) r[k] = a[k];
        int i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int k = 0; k < 8; ++k) r[k] += a[i + k];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        int n2 = n / 2;
        n2 -= n2 % 8;
        constexpr int D1 = DEPTH > 0 ? DEPTH - 1 : 0;
        return np_pairwise_sum<D1>(a, n2) + np_pairwise_sum<D1>(a + n2, n - n2);
    }
}

// temper + normalise + running sum.  Each row is inherently sequential (NumPy's pairwise normaliser order and a
// strictly left-to-right cumsum are what make the result bit-identical), so the parallelism is across rows: a
// workgroup stages ROWS rows through LDS with coalesced loads/stores (a thread walking its own row in global memory
// would touch one 8-byte word per 8*n_cls-byte stride), thread r then owns row r in LDS.  The LDS row stride is odd
// in doubles so the 64 row-walkers hit distinct banks.
//   mode 0 (TH_TEMPER_NONE): rows used as they are;   1: q = x**e here (e in {1, 2, 0.5} — exact IEEE ops, the same
//   fast paths NumPy takes);   2: rows already hold x**e (host pow);   modes 1, 2 renormalise.
// flags[row] bit 0: the running sum is finite and non-decreasing (k_draw may then bisect instead of scanning).
// cum_dtype: the running sum is rounded to this type after every addition (TH_F64 / TH_F32 / TH_F16) — np.cumsum
// accumulates in the array's own dtype, and the reference hands float16 rows straight from predict's
// pdb_to_probability to random_choice_prob_index (sampling_utils.py:82,125); float16 adds go through float like
// NumPy's HALF_add loop.  The comparison with r stays fp64 (NumPy promotes).
__global__ void __launch_bounds__(64) k_temper_cumsum(const double* __restrict__ p, int64_t n_res, int n_cls, double e, int mode,
                                                      int cum_dtype, int rows_per_wg, double* __restrict__ q, double* __restrict__ c,
                                                      unsigned char* __restrict__ flags) {
    extern __shared__ double lds[];
    const int ld = n_cls | 1;
    const int64_t row0 = (int64_t)blockIdx.x * rows_per_wg;
    const int rows = (int)min<int64_t>(rows_per_wg, n_res - row0);
    const int64_t base = row0 * n_cls;
    for (int k = threadIdx.x; k < rows * n_cls; k += 64) lds[(k / n_cls) * ld + k % n_cls] = p[base + k];
    __syncthreads();
    double* qr = lds + (size_t)threadIdx.x * ld;
    double* cr = lds + (size_t)rows_per_wg * ld + (size_t)threadIdx.x * ld;
    if ((int)threadIdx.x < rows) {
        if (mode != 0) {
            if (mode == 1 && e != 1.0)
                for (int j = 0; j < n_cls; ++j) { const double x = qr[j]; qr[j] = (e == 2.0) ? x * x : sqrt(x); }
            const double s = np_pairwise_sum<8>(qr, n_cls);
            for (int j = 0; j < n_cls; ++j) qr[j] = qr[j] / s;
        }
        double run = 0.;
        bool mono = true;
        for (int j = 0; j < n_cls; ++j) {
            double nx = (j == 0) ? qr[0] : run + qr[j];
            if (cum_dtype == TH_F32) nx = (j == 0) ? qr[0] : (double)((float)run + (float)qr[j]);
            else if (cum_dtype == TH_F16) nx = (j == 0) ? qr[0] : (double)(_Float16)((float)(_Float16)run + (float)(_Float16)qr[j]);
            mono = mono && (nx >= run || j == 0) && (nx - nx == 0.0);   // finite and not decreasing
            run = nx;
            cr[j] = run;
        }
        flags[row0 + threadIdx.x] = mono ? 1 : 0;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < rows * n_cls; k += 64) {
        q[base + k] = lds[(k / n_cls) * ld + k % n_cls];
        c[base + k] = lds[(size_t)rows_per_wg * ld + (k / n_cls) * ld + k % n_cls];
    }
}

// One thread per draw.  Draws are numbered in the reference's consumption order
//   for key in keys: for s in range(n_samples): r = np.random.rand(n_res[key])      (sampling_utils.py:118-125)
// so draw d of key k (rows row_off[k] .. row_off[k+1]) is d = n_samples*row_off[k] + s*n_res_k + i, which is also where
// its uniform sits in the caller's / the device generator's stream and where its index / letter is written.
__global__ void __launch_bounds__(256) k_draw(const double* __restrict__ c, const unsigned char* __restrict__ flags, int n_cls,
                                              const int64_t* __restrict__ row_off, int n_keys, int64_t n_samples, int rng_mode,
                                              uint64_t seed, uint64_t rng_offset, const double* __restrict__ uniforms,
                                              int32_t* __restrict__ idx, double* __restrict__ r_out,
                                              const char* __restrict__ letters, char* __restrict__ letters_out) {
    const int64_t total = n_samples * row_off[n_keys];
    for (int64_t d = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; d < total; d += (int64_t)gridDim.x * blockDim.x) {
        int lo = 0, hi = n_keys;                       // last key whose first draw is <= d
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (n_samples * row_off[mid] <= d) lo = mid; else hi = mid;
        }
        const int64_t r0 = row_off[lo], n_res = row_off[lo + 1] - r0;
        const int64_t i = (d - n_samples * r0) % n_res;
        double r;
        if (rng_mode == TH_RNG_PHILOX) {
            rocrand_state_philox4x32_10 st;
            rocrand_init(seed, rng_offset + (uint64_t)d, 0, &st);
            r = rocrand_uniform_double(&st);
        } else {
            r = uniforms[d];
        }
        const double* cr = c + (r0 + i) * n_cls;
        int first = 0;
        if (flags[r0 + i] && n_cls > 32) {             // monotone finite running sum: first index with cr[j] > r by bisection
            int a = 0, b = n_cls;                      // invariant: cr[<a] <= r, cr[>=b] > r
            while (a < b) {
                const int mid = (a + b) >> 1;
                if (cr[mid] > r) b = mid; else a = mid + 1;
            }
            first = a < n_cls ? a : 0;
        } else {
            for (int j = 0; j < n_cls; ++j) {
                if (cr[j] > r) { first = j; break; }
            }
        }
        idx[d] = first;
        if (r_out) r_out[d] = r;
        if (letters_out) letters_out[d] = letters[first];
    }
}

// ---- one launch for a whole run (th_sampler_run): running sums + draws + letters -----------------------------------------------
// k_temper_cumsum + k_draw of a resident sampler are two launches with the running sums written to and read back from HBM in
// between, and on a 300 x 20 matrix k_temper_cumsum is 5 workgroups of 64 sequential row-walkers: 16 us, half the device time
// of a config-5 call.  Here a workgroup owns R = 8 rows: eight threads walk them (the same strictly left-to-right sum in the
// rows' own dtype, the same finite / non-decreasing flag) into LDS, then all 256 threads draw for those rows — thread (row
// t % 8, sample t / 8 + 32 j) — with k_draw's own compare, bisection rule and Philox subsequence numbering (draw d of the
// reference's order), so indices and letters are bit-identical to the two-launch path.  Rows are used as they are
// (TH_TEMPER_NONE): sample.py tempers the whole matrix before it is cut into keys (sample.py:40-41).
constexpr int kFusedRows = 8;
__global__ void __launch_bounds__(256) k_cumsum_draw(const double* __restrict__ p, int64_t n_rows, int n_cls, int cum_dtype,
                                                     const int64_t* __restrict__ row_off, int n_keys, int64_t n_samples, int rng_mode,
                                                     uint64_t seed, uint64_t rng_offset, const double* __restrict__ uniforms,
                                                     int32_t* __restrict__ idx, const char* __restrict__ letters,
                                                     char* __restrict__ letters_out) {
    extern __shared__ double lds[];                        // [R][n_cls | 1] running sums, then R flags, R key starts, R key lengths
    const int ld = n_cls | 1;
    const int64_t row0 = (int64_t)blockIdx.x * kFusedRows;
    const int rows = (int)min<int64_t>(kFusedRows, n_rows - row0);
    int64_t* meta = reinterpret_cast<int64_t*>(lds + (size_t)kFusedRows * ld);     // [3][R]: monotone flag, first row of the key, rows of the key
    for (int k = threadIdx.x; k < rows * n_cls; k += 256) lds[(k / n_cls) * ld + k % n_cls] = p[row0 * n_cls + k];
    __syncthreads();
    if ((int)threadIdx.x < rows) {
        double* cr = lds + (size_t)threadIdx.x * ld;
        double run = 0.;
        bool mono = true;
        for (int j = 0; j < n_cls; ++j) {
            const double x = cr[j];
            double nx = (j == 0) ? x : run + x;
            if (cum_dtype == TH_F32) nx = (j == 0) ? x : (double)((float)run + (float)x);
            else if (cum_dtype == TH_F16) nx = (j == 0) ? x : (double)(_Float16)((float)(_Float16)run + (float)(_Float16)x);
            mono = mono && (nx >= run || j == 0) && (nx - nx == 0.0);
            run = nx;
            cr[j] = run;
        }
        const int64_t row = row0 + threadIdx.x;
        int lo = 0, hi = n_keys;                           // the key that owns this row
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (row_off[mid] <= row) lo = mid; else hi = mid;
        }
        meta[threadIdx.x] = mono ? 1 : 0;
        meta[kFusedRows + threadIdx.x] = row_off[lo];
        meta[2 * kFusedRows + threadIdx.x] = row_off[lo + 1] - row_off[lo];
    }
    __syncthreads();
    const int r = threadIdx.x % kFusedRows;
    if (r >= rows) return;
    const double* cr = lds + (size_t)r * ld;
    const bool mono = meta[r] != 0;
    const int64_t k0 = meta[kFusedRows + r], n_res = meta[2 * kFusedRows + r];
    const int64_t i = row0 + r - k0;
    // blockIdx.y cuts the samples into slices of gridDim.y-strided groups of 32: a 300 x 1000 run is 38 x 32 workgroups with ONE
    // draw per thread (a workgroup that looped over all samples of its rows ran 31 dependent uniform loads per thread: 85 us)
    for (int64_t s = (int64_t)blockIdx.y * (256 / kFusedRows) + threadIdx.x / kFusedRows; s < n_samples; s += (int64_t)gridDim.y * (256 / kFusedRows)) {
        const int64_t d = n_samples * k0 + s * n_res + i;
        double u;
        if (rng_mode == TH_RNG_PHILOX) {
            rocrand_state_philox4x32_10 st;
            rocrand_init(seed, rng_offset + (uint64_t)d, 0, &st);
            u = rocrand_uniform_double(&st);
        } else if (rng_mode == TH_RNG_MT_WORDS) {            // raw MT19937 words: temper + genrand_res53 here (exact: a 53-bit integer / 2^53)
            const uint2 w = reinterpret_cast<const uint2*>(uniforms)[d];
            uint32_t a = w.x, b = w.y;
            a ^= a >> 11; a ^= (a << 7) & 0x9d2c5680u; a ^= (a << 15) & 0xefc60000u; a ^= a >> 18;
            b ^= b >> 11; b ^= (b << 7) & 0x9d2c5680u; b ^= (b << 15) & 0xefc60000u; b ^= b >> 18;
            u = ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
        } else {
            u = uniforms[d];
        }
        int first = 0;
        if (mono && n_cls > 32) {
            int a = 0, b = n_cls;
            while (a < b) {
                const int mid = (a + b) >> 1;
                if (cr[mid] > u) b = mid; else a = mid + 1;
            }
            first = a < n_cls ? a : 0;
        } else {
            for (int j = 0; j < n_cls; ++j) {
                if (cr[j] > u) { first = j; break; }
            }
        }
        if (idx) idx[d] = first;
        if (letters_out) letters_out[d] = letters[first];
    }
}

// ---- sequence metrics (reference design_utils/analyse_utils.py:351-371, called per drawn sequence at
// sampling_utils.py:132) — one wavefront per sampled sequence: 20-bin residue histogram in LDS, then
//   charge(pH) = sum_c n_c * tab[pH][c] + term[pH]   (tab = signed Henderson-Hasselbalch partial charges, host-built)
//   out = (charge at pH 7.4, pI = first grid pH of minimum |charge| on np.arange(1, 13, 0.1), mass, eps280)
// Sums run over the 20 classes in alphabetical one-letter order, fp64, so the host restatement can be matched
// bit for bit.  Letters outside the 20 standard residues are not counted.
constexpr int kGrid = 120;
struct MetricTables {
    double tab74[20], term74;
    double mass[20], water;
    double ext[20];
    double grid[kGrid];
    double tab[kGrid][20];
    double term[kGrid];
};

__global__ void __launch_bounds__(256) k_seq_metrics(const char* __restrict__ letters, const int64_t* __restrict__ row_off,
                                                     int n_keys, int64_t n_samples, const MetricTables* __restrict__ T,
                                                     double* __restrict__ out) {
#pragma clang fp contract(off)       // separate multiply and add, like the host restatement (no FMA contraction)
    __shared__ int hist[4][20];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t seq = (int64_t)blockIdx.x * 4 + wave;          // sequence number: key-major, then sample
    const int64_t n_seq = (int64_t)n_keys * n_samples;
    if (lane < 20) hist[wave][lane] = 0;
    __syncthreads();
    if (seq < n_seq) {
        const int k = (int)(seq / n_samples);
        const int64_t s = seq % n_samples;
        const int64_t r0 = row_off[k], n_res = row_off[k + 1] - r0;
        const char* L = letters + n_samples * r0 + s * n_res;
        for (int64_t i = lane; i < n_res; i += 64) {
            int cls = -1;
            switch (L[i]) {
                case 'A': cls = 0; break;  case 'C': cls = 1; break;  case 'D': cls = 2; break;  case 'E': cls = 3; break;
                case 'F': cls = 4; break;  case 'G': cls = 5; break;  case 'H': cls = 6; break;  case 'I': cls = 7; break;
                case 'K': cls = 8; break;  case 'L': cls = 9; break;  case 'M': cls = 10; break; case 'N': cls = 11; break;
                case 'P': cls = 12; break; case 'Q': cls = 13; break; case 'R': cls = 14; break; case 'S': cls = 15; break;
                case 'T': cls = 16; break; case 'V': cls = 17; break; case 'W': cls = 18; break; case 'Y': cls = 19; break;
                default: break;
            }
            if (cls >= 0) atomicAdd(&hist[wave][cls], 1);
        }
    }
    __syncthreads();
    if (seq >= n_seq) return;
    double n[20];
    for (int c = 0; c < 20; ++c) n[c] = (double)hist[wave][c];
    // isoelectric point: lanes take grid points g = lane, lane + 64; first minimum of |charge| wins (np.argmin / min())
    double best = 1e300;
    int best_g = kGrid;
    for (int g = lane; g < kGrid; g += 64) {
        double ch = 0.;
        for (int c = 0; c < 20; ++c) ch += n[c] * T->tab[g][c];
        ch += T->term[g];
        const double a = fabs(ch);
        if (a < best) { best = a; best_g = g; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_xor(best, off);
        const int og = __shfl_xor(best_g, off);
        if (ob < best || (ob == best && og < best_g)) { best = ob; best_g = og; }
    }
    if (lane == 0) {
        double ch = 0., mw = 0., ex = 0.;
        for (int c = 0; c < 20; ++c) ch += n[c] * T->tab74[c];
        ch += T->term74;
        for (int c = 0; c < 20; ++c) mw += n[c] * T->mass[c];
        mw += T->water;
        for (int c = 0; c < 20; ++c) ex += n[c] * T->ext[c];
        double* o = out + seq * 4;
        o[0] = ch; o[1] = T->grid[best_g < kGrid ? best_g : 0]; o[2] = mw; o[3] = ex;
    }
}

// ---- MT19937 exactly as NumPy's legacy global RandomState ------------------------------------
// np.random.seed(s): init_genrand(s) (Knuth multiplier 1812433253), pos = 624.
// np.random.rand(): genrand_res53: a = u32()>>5, b = u32()>>6, (a*67108864+b)/9007199254740992.
constexpr int MT_N = 624, MT_M = 397;
__global__ void __launch_bounds__(256) k_mt19937_uniforms(uint32_t seed, uint64_t skip_doubles, int64_t n_doubles,
                                                          double* __restrict__ out) {
    __shared__ uint32_t mt[MT_N];
    __shared__ uint32_t tw[MT_N];
    const int tid = threadIdx.x;
    if (tid == 0) {
        mt[0] = seed;
        for (int i = 1; i < MT_N; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    }
    __syncthreads();
    const uint64_t w_begin = 2 * skip_doubles, w_end = 2 * (skip_doubles + (uint64_t)n_doubles);
    for (uint64_t base = 0; base < w_end; base += MT_N) {
        // regenerate the 624-word block in three dependency phases
        const int lo[3] = {0, MT_N - MT_M, 2 * (MT_N - MT_M)};
        const int hi[3] = {MT_N - MT_M, 2 * (MT_N - MT_M), MT_N};
        for (int ph = 0; ph < 3; ++ph) {
            uint32_t nv[3];
            int cnt = 0;
            for (int i = lo[ph] + tid; i < hi[ph]; i += 256, ++cnt) {
                const uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % MT_N] & 0x7fffffffu);
                nv[cnt] = mt[(i + MT_M) % MT_N] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            __syncthreads();
            cnt = 0;
            for (int i = lo[ph] + tid; i < hi[ph]; i += 256, ++cnt) mt[i] = nv[cnt];
            __syncthreads();
        }
        if (base + MT_N <= w_begin) continue;  // whole block skipped
        for (int i = tid; i < MT_N; i += 256) {
            uint32_t y = mt[i];
            y ^= (y >> 11);
            y ^= (y << 7) & 0x9d2c5680u;
            y ^= (y << 15) & 0xefc60000u;
            y ^= (y >> 18);
            tw[i] = y;
        }
        __syncthreads();
        for (int k = tid; k < MT_N / 2; k += 256) {
            const uint64_t w = base + 2 * (uint64_t)k;
            if (w >= w_begin && w < w_end) {
                const uint32_t a = tw[2 * k] >> 5, b = tw[2 * k + 1] >> 6;
                out[(w - w_begin) / 2] = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
            }
        }
        __syncthreads();
    }
}

// grow-only device / pinned-host scratch
struct Scratch {
    void* p = nullptr;
    size_t cap = 0;
    bool host = false;
    int ensure(size_t bytes) {
        if (cap >= bytes && p) return TH_OK;
        if (p) { if (host) (void)hipHostFree(p); else (void)hipFree(p); p = nullptr; cap = 0; }
        const size_t want = std::max<size_t>(bytes + bytes / 4, 1 << 16);
        hipError_t e = host ? hipHostMalloc(&p, want, hipHostMallocDefault) : th_malloc_retry(&p, want);
        if (e != hipSuccess) { p = nullptr; th_set_error("sampler: allocating %zu bytes failed: %s", want, hipGetErrorString(e)); return TH_ENOMEM; }
        cap = want;
        return TH_OK;
    }
    void release() { if (p) { if (host) (void)hipHostFree(p); else (void)hipFree(p); } p = nullptr; cap = 0; }
};

// ampal's tables as restated in design_utils/analyse_utils.py (PARITY UNPINNED: ampal 1.5.1 is not in the reference
// tree); class order = alphabetical one-letter codes ACDEFGHIKLMNPQRSTVWY.
const double kMass[20] = {71.0779, 103.1429, 115.0874, 129.114, 147.1739, 57.0513, 137.1393, 113.1576, 128.1723, 113.1576,
                          131.1961, 114.1026, 97.1152, 128.1292, 156.1857, 87.0773, 101.1039, 99.1311, 186.2099, 163.1733};
const double kWater = 18.01528;
const double kExt[20] = {0, 120, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 5690, 1280};
const int kSign[20] = {0, -1, -1, -1, 0, 0, +1, 0, +1, 0, 0, 0, 0, 0, +1, 0, 0, 0, 0, -1};
const double kPka[20] = {0, 8.3, 3.65, 4.25, 0, 0, 6.1, 0, 10.53, 0, 0, 0, 0, 0, 12.48, 0, 0, 0, 0, 10.1};
const double kPkaNterm = 8.0, kPkaCterm = 3.1;

double partial_charge(double pka, int sign, double ph) {
    double diff = ph - pka;
    if (sign > 0) diff = -diff;
    const double r = std::pow(10.0, diff);
    return r / (1.0 + r);
}

void build_metric_tables(MetricTables* T) {
    auto fill = [](double ph, double* tab, double* term) {
        for (int c = 0; c < 20; ++c) tab[c] = kSign[c] ? partial_charge(kPka[c], kSign[c], ph) * kSign[c] : 0.0;
        *term = partial_charge(kPkaNterm, +1, ph) * (+1) + partial_charge(kPkaCterm, -1, ph) * (-1);
    };
    fill(7.4, T->tab74, &T->term74);
    for (int c = 0; c < 20; ++c) { T->mass[c] = kMass[c]; T->ext[c] = kExt[c]; }
    T->water = kWater;
    for (int g = 0; g < kGrid; ++g) {
        const volatile double first = 1.0, next = 1.0 + 0.1;
        T->grid[g] = first + g * (next - first);   // np.arange(1, 13, 0.1) fills first + i*delta, delta = (start+step)-start
        fill(T->grid[g], T->tab[g], &T->term[g]);
    }
}

}  // namespace

// A sampler keeps the probability rows of a whole run resident on one device (tempered, normalised, with their
// running sums) and draws any number of sequences for any subset of keys in ONE launch sequence on its own stream.
struct th_sampler {
    int device = 0;
    hipStream_t stream = nullptr;
    int64_t n_rows = 0;
    int n_cls = 0;
    Scratch dp, dq, dc, dflags, drow, du, di, dr, dlet, dcat, dmet, dtab;
    Scratch hq;                     // host buffer for the powered rows
    Scratch hu;                     // page-locked buffer handed to the caller for its uniforms / raw generator words (th_sampler_uniform_buffer)
    Scratch hin, din, hout, dout;   // th_sampler_run: ONE page-locked input block / device copy, ONE device output block / page-locked copy
    bool tables = false;
    std::mutex mu;
};

namespace {

// Scope guard: every exit path after the first enqueue — a failed Scratch::ensure, a HIP error — synchronises the
// sampler's stream before returning, so no queued copy can still read a local vector / a caller buffer and no queued
// kernel can still use a scratch buffer that the next call reallocates.
struct StreamDrain {
    hipStream_t s;
    ~StreamDrain() { if (s) { (void)hipStreamSynchronize(s); } }
};

int sampler_load(th_sampler* S, const double* probs, int64_t n_rows, int n_cls, double t, int mode, int cum_dtype, double* q_out) {
    if (!S || !probs || n_rows <= 0 || n_cls <= 0) TH_FAIL(TH_EINVAL, "th_sampler_load: bad shape");
    if (cum_dtype != TH_F64 && cum_dtype != TH_F32 && cum_dtype != TH_F16) TH_FAIL(TH_EINVAL, "th_sampler_load: running-sum dtype %d", cum_dtype);
    if (mode < TH_TEMPER_NONE || mode > TH_TEMPER_PREPOWERED) TH_FAIL(TH_EINVAL, "th_sampler_load: temper mode %d", mode);
    if (mode != TH_TEMPER_NONE && t == 0.0)
        TH_FAIL(TH_EINVAL, "th_sample: temperature 0 (the reference divides by it: sampling_utils.py:159)");
    HIP_TRY(hipSetDevice(S->device));
    StreamDrain drain{S->stream};   // any return below (errors included) leaves nothing queued on caller / scratch memory
    const size_t cells = (size_t)n_rows * n_cls, bytes = cells * sizeof(double);
    int rc;
    if ((rc = S->dp.ensure(bytes)) || (rc = S->dq.ensure(bytes)) || (rc = S->dc.ensure(bytes)) ||
        (rc = S->dflags.ensure((size_t)n_rows)))
        return rc;
    const double e = mode == TH_TEMPER_POW ? 1.0 / t : 1.0;
    const double* src = probs;
    int kmode = mode;
    if (mode == TH_TEMPER_POW && e != 1.0 && e != 2.0 && e != 0.5) {
        // generic exponent: the power runs on the host with libm's pow — the function NumPy's float64 `**` loop calls
        // element by element (its AVX-512 SVML loop may differ from it in the last bit; callers that must match a
        // particular NumPy build pass rows they powered themselves, TH_TEMPER_PREPOWERED).  Device pow() is not
        // correctly rounded and an ulp in q can move a residue index when r lands on a CDF boundary.
        S->hq.host = true;
        if ((rc = S->hq.ensure(bytes))) return rc;
        double* h = (double*)S->hq.p;
        for (size_t k = 0; k < cells; ++k) h[k] = std::pow(probs[k], e);
        src = h;
        kmode = TH_TEMPER_PREPOWERED;
    }
    HIP_TRY(hipMemcpyAsync(S->dp.p, src, bytes, hipMemcpyHostToDevice, S->stream));
    // rows per workgroup: as many as fit 64 row-walkers and ~60 KB of LDS (two images: q and its running sum)
    const int ld = n_cls | 1;
    int rows = (int)std::min<int64_t>(64, std::max<int64_t>(1, (60 * 1024) / ((size_t)2 * ld * sizeof(double))));
    const size_t lds = (size_t)2 * rows * ld * sizeof(double);
    if (lds > 64 * 1024 + 0 && rows == 1) {
        // a single row wider than the LDS budget: raise the dynamic limit (160 KB per CU on gfx950)
        HIP_TRY(hipFuncSetAttribute((const void*)k_temper_cumsum, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    if (lds > 160 * 1024) TH_FAIL(TH_EUNSUP, "th_sampler_load: %d categories per row exceed the LDS budget", n_cls);
    hipLaunchKernelGGL(k_temper_cumsum, dim3((unsigned)((n_rows + rows - 1) / rows)), dim3(64), lds, S->stream,
                       (const double*)S->dp.p, n_rows, n_cls, e, kmode, cum_dtype, rows, (double*)S->dq.p, (double*)S->dc.p,
                       (unsigned char*)S->dflags.p);
    HIP_TRY(hipGetLastError());
    if (q_out) HIP_TRY(hipMemcpyAsync(q_out, S->dq.p, bytes, hipMemcpyDeviceToHost, S->stream));
    HIP_TRY(hipStreamSynchronize(S->stream));      // `probs` / the host pow buffer may be reused by the caller now
    S->n_rows = n_rows;
    S->n_cls = n_cls;
    return TH_OK;
}

int sampler_draw(th_sampler* S, int64_t n_keys, const int64_t* row_off, int64_t n_samples, int rng_mode, uint64_t seed,
                 uint64_t rng_offset, const double* uniforms, const char* cat_letters, int32_t* idx_out, double* r_out,
                 char* letters_out, double* metrics_out) {
    if (!S || !row_off || n_keys <= 0 || n_samples < 0) TH_FAIL(TH_EINVAL, "th_sampler_draw: bad argument");
    if (S->n_rows <= 0) TH_FAIL(TH_EINVAL, "th_sampler_draw: no probabilities loaded (th_sampler_load)");
    if (n_keys > 0x7fffffff) TH_FAIL(TH_EINVAL, "th_sampler_draw: too many keys");
    if (row_off[0] < 0) TH_FAIL(TH_EINVAL, "th_sampler_draw: negative row offset");
    for (int64_t k = 0; k < n_keys; ++k)
        if (row_off[k + 1] <= row_off[k]) TH_FAIL(TH_EINVAL, "th_sampler_draw: key %lld has no rows", (long long)k);
    if (row_off[n_keys] > S->n_rows) TH_FAIL(TH_EINVAL, "th_sampler_draw: rows beyond the loaded matrix");
    if (rng_mode < TH_RNG_HOST || rng_mode > TH_RNG_MT19937) TH_FAIL(TH_EINVAL, "th_sample: rng_mode %d", rng_mode);
    if (rng_mode == TH_RNG_HOST && n_samples > 0 && !uniforms) TH_FAIL(TH_EINVAL, "th_sample: rng_mode 0 needs uniforms");
    if (rng_mode == TH_RNG_MT19937 && seed > 0xffffffffULL) TH_FAIL(TH_EINVAL, "th_sample: MT19937 seed must fit 32 bits");
    if ((letters_out || metrics_out) && !cat_letters) TH_FAIL(TH_EINVAL, "th_sample: letters_out / metrics_out need cat_letters");
    HIP_TRY(hipSetDevice(S->device));
    // draws are numbered from the first requested row: shift the offsets so that key 0 starts at draw 0
    std::vector<int64_t> off(n_keys + 1);
    StreamDrain drain{S->stream};   // declared after `off`: the stream is drained before the vector (a copy source) dies
    const int64_t base_row = row_off[0];
    for (int64_t k = 0; k <= n_keys; ++k) off[k] = row_off[k] - base_row;
    const int64_t total = n_samples * off[n_keys];
    if (total == 0) return TH_OK;
    int rc;
    if ((rc = S->drow.ensure((size_t)(n_keys + 1) * sizeof(int64_t)))) return rc;
    HIP_TRY(hipMemcpyAsync(S->drow.p, off.data(), (size_t)(n_keys + 1) * sizeof(int64_t), hipMemcpyHostToDevice, S->stream));
    const bool need_u = rng_mode != TH_RNG_PHILOX;
    if (need_u) {
        if ((rc = S->du.ensure((size_t)total * sizeof(double)))) return rc;
        if (rng_mode == TH_RNG_HOST) {
            HIP_TRY(hipMemcpyAsync(S->du.p, uniforms, (size_t)total * sizeof(double), hipMemcpyHostToDevice, S->stream));
        } else {
            hipLaunchKernelGGL(k_mt19937_uniforms, dim3(1), dim3(256), 0, S->stream, (uint32_t)seed, rng_offset, total, (double*)S->du.p);
            HIP_TRY(hipGetLastError());
        }
    }
    if ((rc = S->di.ensure((size_t)total * sizeof(int32_t)))) return rc;
    if (r_out && (rc = S->dr.ensure((size_t)total * sizeof(double)))) return rc;
    const bool want_letters = letters_out || metrics_out;
    if (want_letters) {
        if ((rc = S->dcat.ensure((size_t)S->n_cls)) || (rc = S->dlet.ensure((size_t)total))) return rc;
        HIP_TRY(hipMemcpyAsync(S->dcat.p, cat_letters, (size_t)S->n_cls, hipMemcpyHostToDevice, S->stream));
    }
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    const size_t row_shift = (size_t)base_row * S->n_cls;
    hipLaunchKernelGGL(k_draw, dim3((unsigned)blocks), dim3(256), 0, S->stream, (const double*)S->dc.p + row_shift,
                       (const unsigned char*)S->dflags.p + base_row, S->n_cls, (const int64_t*)S->drow.p, (int)n_keys, n_samples,
                       rng_mode, seed, rng_offset, (const double*)S->du.p, (int32_t*)S->di.p, r_out ? (double*)S->dr.p : nullptr,
                       (const char*)S->dcat.p, want_letters ? (char*)S->dlet.p : nullptr);
    HIP_TRY(hipGetLastError());
    if (metrics_out) {
        if (!S->tables) {
            std::unique_ptr<MetricTables> T(new MetricTables);
            build_metric_tables(T.get());
            if ((rc = S->dtab.ensure(sizeof(MetricTables)))) return rc;
            HIP_TRY(hipMemcpy(S->dtab.p, T.get(), sizeof(MetricTables), hipMemcpyHostToDevice));
            S->tables = true;
        }
        const int64_t n_seq = n_keys * n_samples;
        if ((rc = S->dmet.ensure((size_t)n_seq * 4 * sizeof(double)))) return rc;
        hipLaunchKernelGGL(k_seq_metrics, dim3((unsigned)((n_seq + 3) / 4)), dim3(256), 0, S->stream, (const char*)S->dlet.p,
                           (const int64_t*)S->drow.p, (int)n_keys, n_samples, (const MetricTables*)S->dtab.p, (double*)S->dmet.p);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(metrics_out, S->dmet.p, (size_t)n_seq * 4 * sizeof(double), hipMemcpyDeviceToHost, S->stream));
    }
    if (idx_out) HIP_TRY(hipMemcpyAsync(idx_out, S->di.p, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost, S->stream));
    if (r_out) HIP_TRY(hipMemcpyAsync(r_out, S->dr.p, (size_t)total * sizeof(double), hipMemcpyDeviceToHost, S->stream));
    if (letters_out) HIP_TRY(hipMemcpyAsync(letters_out, S->dlet.p, (size_t)total, hipMemcpyDeviceToHost, S->stream));
    HIP_TRY(hipStreamSynchronize(S->stream));
    return TH_OK;
}

// th_sampler_run: a whole sample.py run as ONE submission.  Inputs (row offsets, category letters, probability rows) go through one
// page-locked block and one host->device copy; k_cumsum_draw and k_seq_metrics write into one device block — [metrics | idx |
// letters] — that comes back in one copy into a page-locked block owned by the sampler (valid until its next call).  With a
// device generator that is two copies and two kernels per call; caller-supplied uniforms add the copy of those.
inline size_t up16(size_t x) { return (x + 15) / 16 * 16; }

int sampler_run_fused(th_sampler* S, const double* probs, int64_t n_rows, int n_cls, int cum_dtype, int64_t n_keys, const int64_t* row_off,
                      int64_t n_samples, int rng_mode, uint64_t seed, uint64_t rng_offset, const double* uniforms, const char* cat_letters,
                      unsigned want, const void** block_out, int64_t* offsets_out) {
    if (!S || !probs || !row_off || !block_out || !offsets_out || n_rows <= 0 || n_cls <= 0 || n_keys <= 0 || n_samples < 0)
        TH_FAIL(TH_EINVAL, "th_sampler_run: bad argument");
    if (cum_dtype != TH_F64 && cum_dtype != TH_F32 && cum_dtype != TH_F16) TH_FAIL(TH_EINVAL, "th_sampler_run: running-sum dtype %d", cum_dtype);
    if (n_keys > 0x7fffffff) TH_FAIL(TH_EINVAL, "th_sampler_run: too many keys");
    if (row_off[0] != 0 || row_off[n_keys] != n_rows) TH_FAIL(TH_EINVAL, "th_sampler_run: the keys must cover rows 0..n_rows");
    for (int64_t k = 0; k < n_keys; ++k)
        if (row_off[k + 1] <= row_off[k]) TH_FAIL(TH_EINVAL, "th_sampler_run: key %lld has no rows", (long long)k);
    if (rng_mode < TH_RNG_HOST || rng_mode > TH_RNG_MT_WORDS) TH_FAIL(TH_EINVAL, "th_sampler_run: rng_mode %d", rng_mode);
    if ((rng_mode == TH_RNG_HOST || rng_mode == TH_RNG_MT_WORDS) && n_samples > 0 && !uniforms)
        TH_FAIL(TH_EINVAL, "th_sampler_run: rng_mode %d needs uniforms", rng_mode);
    if (rng_mode == TH_RNG_MT19937 && seed > 0xffffffffULL) TH_FAIL(TH_EINVAL, "th_sampler_run: MT19937 seed must fit 32 bits");
    const bool want_idx = want & 1u, want_let = (want & 2u) != 0, want_met = (want & 4u) != 0;
    if ((want_let || want_met) && !cat_letters) TH_FAIL(TH_EINVAL, "th_sampler_run: letters / metrics need cat_letters");
    if (!(want & 7u)) TH_FAIL(TH_EINVAL, "th_sampler_run: nothing requested");
    HIP_TRY(hipSetDevice(S->device));
    StreamDrain drain{S->stream};
    const int64_t total = n_samples * n_rows, n_seq = n_keys * n_samples;
    // ---- input block: [row_off][letters][rows]
    const size_t o_off = 0, o_cat = up16((size_t)(n_keys + 1) * 8), o_p = o_cat + up16((size_t)n_cls),
                 in_bytes = o_p + (size_t)n_rows * n_cls * 8;
    S->hin.host = true;
    int rc;
    if ((rc = S->hin.ensure(in_bytes)) || (rc = S->din.ensure(in_bytes))) return rc;
    char* hi = (char*)S->hin.p;
    std::memcpy(hi + o_off, row_off, (size_t)(n_keys + 1) * 8);
    if (cat_letters) std::memcpy(hi + o_cat, cat_letters, (size_t)n_cls);
    std::memcpy(hi + o_p, probs, (size_t)n_rows * n_cls * 8);
    HIP_TRY(hipMemcpyAsync(S->din.p, hi, in_bytes, hipMemcpyHostToDevice, S->stream));
    const char* di = (const char*)S->din.p;
    // ---- uniforms
    if (rng_mode != TH_RNG_PHILOX && total > 0) {
        if ((rc = S->du.ensure((size_t)total * sizeof(double)))) return rc;
        if (rng_mode == TH_RNG_HOST || rng_mode == TH_RNG_MT_WORDS) {      // doubles, or two raw 32-bit words per draw: 8 bytes either way
            HIP_TRY(hipMemcpyAsync(S->du.p, uniforms, (size_t)total * sizeof(double), hipMemcpyHostToDevice, S->stream));
        } else {
            hipLaunchKernelGGL(k_mt19937_uniforms, dim3(1), dim3(256), 0, S->stream, (uint32_t)seed, rng_offset, total, (double*)S->du.p);
            HIP_TRY(hipGetLastError());
        }
    }
    // ---- output block: [metrics n_seq x 4 doubles][idx total x int32][letters total bytes]; only the requested parts exist
    const size_t m_bytes = want_met ? up16((size_t)n_seq * 4 * sizeof(double)) : 0;
    const size_t i_bytes = want_idx ? up16((size_t)total * sizeof(int32_t)) : 0;
    const bool dev_letters = want_let || want_met;
    const size_t l_bytes = dev_letters ? up16((size_t)total) : 0;
    const size_t out_bytes = m_bytes + i_bytes + l_bytes;
    S->hout.host = true;
    if ((rc = S->dout.ensure(out_bytes + 16)) || (rc = S->hout.ensure(out_bytes + 16))) return rc;
    char* dout = (char*)S->dout.p;
    offsets_out[0] = want_idx ? (int64_t)m_bytes : -1;
    offsets_out[1] = want_let ? (int64_t)(m_bytes + i_bytes) : -1;
    offsets_out[2] = want_met ? 0 : -1;
    if (total > 0) {
        const size_t lds = ((size_t)kFusedRows * (n_cls | 1)) * sizeof(double) + 3 * kFusedRows * sizeof(int64_t);
        if (lds > 160 * 1024) TH_FAIL(TH_EUNSUP, "th_sampler_run: %d categories per row exceed the LDS budget", n_cls);
        if (lds > 48 * 1024) HIP_TRY(hipFuncSetAttribute((const void*)k_cumsum_draw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int64_t row_groups = (n_rows + kFusedRows - 1) / kFusedRows;
        const int64_t slices = std::max<int64_t>(1, std::min<int64_t>({(n_samples + 31) / 32, 65535, (8192 + row_groups - 1) / row_groups}));
        hipLaunchKernelGGL(k_cumsum_draw, dim3((unsigned)row_groups, (unsigned)slices), dim3(256), lds, S->stream,
                           (const double*)(di + o_p), n_rows, n_cls, cum_dtype, (const int64_t*)(di + o_off), (int)n_keys, n_samples, rng_mode,
                           seed, rng_offset, (const double*)S->du.p, want_idx ? (int32_t*)(dout + m_bytes) : nullptr, di + o_cat,
                           dev_letters ? dout + m_bytes + i_bytes : nullptr);
        HIP_TRY(hipGetLastError());
        if (want_met) {
            if (!S->tables) {
                std::unique_ptr<MetricTables> T(new MetricTables);
                build_metric_tables(T.get());
                if ((rc = S->dtab.ensure(sizeof(MetricTables)))) return rc;
                HIP_TRY(hipMemcpy(S->dtab.p, T.get(), sizeof(MetricTables), hipMemcpyHostToDevice));
                S->tables = true;
            }
            hipLaunchKernelGGL(k_seq_metrics, dim3((unsigned)((n_seq + 3) / 4)), dim3(256), 0, S->stream, (const char*)(dout + m_bytes + i_bytes),
                               (const int64_t*)(di + o_off), (int)n_keys, n_samples, (const MetricTables*)S->dtab.p, (double*)dout);
            HIP_TRY(hipGetLastError());
        }
        // one copy back: the letters are last, so a call that does not return them copies the prefix in front of them only
        const size_t back = want_let ? out_bytes : m_bytes + i_bytes;
        if (back) HIP_TRY(hipMemcpyAsync(S->hout.p, dout, back, hipMemcpyDeviceToHost, S->stream));
    }
    HIP_TRY(hipStreamSynchronize(S->stream));
    *block_out = S->hout.p;
    return TH_OK;
}

// one lazily created sampler per device behind the one-shot entry points (th_apply_temp / th_sample / th_sample_ex)
std::mutex g_default_mu;
std::vector<th_sampler*> g_default;

th_sampler* default_sampler(int device) {
    std::lock_guard<std::mutex> lock(g_default_mu);
    if (device < 0 || device >= 1024) { th_set_error("bad device index %d", device); return nullptr; }
    if ((int)g_default.size() <= device) g_default.resize(device + 1, nullptr);
    if (!g_default[device]) {
        th_sampler* s = nullptr;
        if (th_sampler_create(device, &s) != TH_OK) return nullptr;
        g_default[device] = s;
    }
    return g_default[device];
}

}  // namespace

extern "C" {

int th_sampler_create(int device, th_sampler** out) {
    if (!out) TH_FAIL(TH_EINVAL, "null argument");
    HIP_TRY(hipSetDevice(device));
    std::unique_ptr<th_sampler> s(new th_sampler);
    s->device = device;
    HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    *out = s.release();
    return TH_OK;
}

void th_sampler_free(th_sampler* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (Scratch* b : {&s->dp, &s->dq, &s->dc, &s->dflags, &s->drow, &s->du, &s->di, &s->dr, &s->dlet, &s->dcat, &s->dmet, &s->dtab, &s->hq,
                       &s->hin, &s->din, &s->hout, &s->dout, &s->hu})
        b->release();
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

int th_sampler_load(th_sampler* s, const double* probs, int64_t n_rows, int n_cls, double temperature, int temper_mode,
                    int cum_dtype, double* q_out) {
    if (!s) TH_FAIL(TH_EINVAL, "null sampler");
    std::lock_guard<std::mutex> lock(s->mu);
    return sampler_load(s, probs, n_rows, n_cls, temperature, temper_mode, cum_dtype, q_out);
}

int th_sampler_draw(th_sampler* s, int64_t n_keys, const int64_t* row_off, int64_t n_samples, int rng_mode, uint64_t seed,
                    uint64_t rng_offset, const double* uniforms, const char* cat_letters, int32_t* idx_out, double* r_out,
                    char* letters_out, double* metrics_out) {
    if (!s) TH_FAIL(TH_EINVAL, "null sampler");
    std::lock_guard<std::mutex> lock(s->mu);
    return sampler_draw(s, n_keys, row_off, n_samples, rng_mode, seed, rng_offset, uniforms, cat_letters, idx_out, r_out,
                        letters_out, metrics_out);
}

int th_sampler_uniform_buffer(th_sampler* s, size_t bytes, void** out) {
    if (!s || !out) TH_FAIL(TH_EINVAL, "null argument");
    std::lock_guard<std::mutex> lock(s->mu);
    HIP_TRY(hipSetDevice(s->device));
    s->hu.host = true;
    if (int rc = s->hu.ensure(bytes)) return rc;
    *out = s->hu.p;
    return TH_OK;
}

int th_sampler_run(th_sampler* s, const double* probs, int64_t n_rows, int n_cls, int cum_dtype, int64_t n_keys, const int64_t* row_off,
                   int64_t n_samples, int rng_mode, uint64_t seed, uint64_t rng_offset, const double* uniforms, const char* cat_letters,
                   unsigned want, const void** block_out, int64_t* offsets_out) {
    if (!s) TH_FAIL(TH_EINVAL, "null sampler");
    std::lock_guard<std::mutex> lock(s->mu);
    return sampler_run_fused(s, probs, n_rows, n_cls, cum_dtype, n_keys, row_off, n_samples, rng_mode, seed, rng_offset, uniforms, cat_letters,
                             want, block_out, offsets_out);
}

}  // extern "C"

int sampler_run(int device, const double* h_probs, int64_t n_res, int n_cls, int64_t n_samples, double temperature,
                int rng_mode, uint64_t seed, uint64_t rng_offset, const double* h_uniforms, int32_t* h_idx,
                double* h_r_out, const char* cat_letters, char* h_letters, double* h_q_out) {
    if (!h_probs || n_res <= 0 || n_cls <= 0 || n_samples < 0) TH_FAIL(TH_EINVAL, "th_sample: bad shape");
    if (temperature == 0.0) TH_FAIL(TH_EINVAL, "th_sample: temperature 0 (the reference divides by it: sampling_utils.py:159)");
    if (rng_mode == TH_RNG_HOST && n_samples > 0 && !h_uniforms) TH_FAIL(TH_EINVAL, "th_sample: rng_mode 0 needs uniforms");
    if (rng_mode < TH_RNG_HOST || rng_mode > TH_RNG_MT19937) TH_FAIL(TH_EINVAL, "th_sample: rng_mode %d", rng_mode);
    if (rng_mode == TH_RNG_MT19937 && seed > 0xffffffffULL) TH_FAIL(TH_EINVAL, "th_sample: MT19937 seed must fit 32 bits");
    if (h_letters && !cat_letters) TH_FAIL(TH_EINVAL, "th_sample: letters_out needs cat_letters");
    th_sampler* S = default_sampler(device);
    if (!S) return TH_EHIP;
    std::lock_guard<std::mutex> lock(S->mu);
    // sample.py:40 skips apply_temp_to_probs when t == 1 (rows stay un-normalised); apply_temp_to_probs itself
    // (n_samples == 0 with q_out) always renormalises, also at t == 1 where x**1.0 is x.
    const bool apply = temperature != 1.0 || (n_samples == 0 && h_q_out);
    int rc = sampler_load(S, h_probs, n_res, n_cls, temperature, apply ? TH_TEMPER_POW : TH_TEMPER_NONE, TH_F64, h_q_out);
    if (rc || n_samples == 0) return rc;
    const int64_t row_off[2] = {0, n_res};
    return sampler_draw(S, 1, row_off, n_samples, rng_mode, seed, rng_offset, h_uniforms, cat_letters, h_idx, h_r_out,
                        h_letters, nullptr);
}

// ---- th_mt19937_rand: np.random.rand(n) replayed natively (host code) ------------------------------------------------------------
// The reference draws its uniforms from NumPy's GLOBAL legacy generator — r = np.random.rand(n), sampling_utils.py:81 — i.e. MT19937
// and genrand_res53 ((a >> 5) * 2^26 + (b >> 6)) / 2^53 on one host core, value by value through NumPy's per-call machinery: 0.70 ms
// for the 300 000 uniforms of one config-5 call, two thirds of the whole API call.  Given the generator's state (np.random.get_state():
// 624 key words and the position in them) this fills `out` with the SAME n doubles and leaves key / pos as NumPy would have left them,
// so the caller puts the state back (np.random.set_state) and the stream continues as if np.random.rand had been called: whole
// 624-word blocks are regenerated with loops the compiler vectorises (the recurrence reaches back 227 words), tempered in bulk and
// converted in pairs.  Checked against np.random.rand for seeds, start positions around the block boundary (odd positions: a pair
// straddles two blocks) and lengths, including the values drawn AFTER the call (tests/test_host_utils.py).
namespace {
inline void mt19937_next_block(uint32_t* mt) {
    constexpr int N = 624, M = 397;
    constexpr uint32_t A = 0x9908b0dfU, UP = 0x80000000U, LO = 0x7fffffffU;
    int kk = 0;
    for (; kk < N - M; ++kk) { const uint32_t y = (mt[kk] & UP) | (mt[kk + 1] & LO); mt[kk] = mt[kk + M] ^ (y >> 1) ^ ((0u - (y & 1u)) & A); }
    for (; kk < N - 1; ++kk) { const uint32_t y = (mt[kk] & UP) | (mt[kk + 1] & LO); mt[kk] = mt[kk + (M - N)] ^ (y >> 1) ^ ((0u - (y & 1u)) & A); }
    const uint32_t y = (mt[N - 1] & UP) | (mt[0] & LO);
    mt[N - 1] = mt[M - 1] ^ (y >> 1) ^ ((0u - (y & 1u)) & A);
}
inline uint32_t mt19937_temper(uint32_t y) {
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680U; y ^= (y << 15) & 0xefc60000U; y ^= y >> 18;
    return y;
}
inline double res53(uint32_t a, uint32_t b) {       // a, b already shifted: 27 and 26 bits; the scale is a power of two (exact)
    return ((double)(int32_t)a * 67108864.0 + (double)(int32_t)b) * (1.0 / 9007199254740992.0);
}
}  // namespace

extern "C" int th_mt19937_words(uint32_t* key, int* pos_io, int64_t n, uint32_t* out) {
    if (!key || !pos_io || n < 0 || (!out && n > 0)) TH_FAIL(TH_EINVAL, "th_mt19937_words: null argument");
    int pos = *pos_io;
    if (pos < 0 || pos > 624) TH_FAIL(TH_EINVAL, "th_mt19937_words: position %d outside 0..624", pos);
    int64_t words = 2 * n;
    while (words > 0) {
        if (pos >= 624) { mt19937_next_block(key); pos = 0; }
        const int cnt = (int)std::min<int64_t>(words, 624 - pos);
        std::memcpy(out, key + pos, (size_t)cnt * sizeof(uint32_t));
        out += cnt; words -= cnt; pos += cnt;
    }
    *pos_io = pos;
    return TH_OK;
}

extern "C" int th_mt19937_rand(uint32_t* key, int* pos_io, int64_t n, double* out) {
    if (!key || !pos_io || n < 0 || (!out && n > 0)) TH_FAIL(TH_EINVAL, "th_mt19937_rand: null argument");
    int pos = *pos_io;
    if (pos < 0 || pos > 624) TH_FAIL(TH_EINVAL, "th_mt19937_rand: position %d outside 0..624", pos);
    while (n > 0) {
        if (pos >= 624) { mt19937_next_block(key); pos = 0; }
        const int64_t pairs = std::min<int64_t>(n, (624 - pos) / 2);
        if (pairs > 0) {
            uint32_t tw[624];
            const int cnt = 2 * (int)pairs;
            for (int i = 0; i < cnt; ++i) tw[i] = mt19937_temper(key[pos + i]);
            for (int64_t i = 0; i < pairs; ++i) out[i] = res53(tw[2 * i] >> 5, tw[2 * i + 1] >> 6);
            out += pairs; n -= pairs; pos += cnt;
        } else {                                    // one word left in this block: the pair straddles two blocks
            const uint32_t a = mt19937_temper(key[pos]) >> 5;
            mt19937_next_block(key);
            pos = 0;
            const uint32_t b = mt19937_temper(key[pos++]) >> 6;
            *out++ = res53(a, b);
            --n;
        }
    }
    *pos_io = pos;
    return TH_OK;
}
