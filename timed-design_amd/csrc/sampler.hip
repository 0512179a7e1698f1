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
#include <mutex>

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

// one thread per residue row: temper, normalise, running sum
__global__ void k_temper_cumsum(const double* __restrict__ p, int64_t n_res, int n_cls, double t, int force_norm,
                                double* __restrict__ q, double* __restrict__ c) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n_res) return;
    const double* pr = p + i * n_cls;
    double* qr = q + i * n_cls;
    double* cr = c + i * n_cls;
    // sample.py:40 skips apply_temp_to_probs when t == 1 (rows stay un-normalised); the function
    // itself (force_norm) always renormalises, also at t == 1 where x**1.0 is x.
    if (t != 1.0 || force_norm) {
        const double e = 1.0 / t;
        for (int j = 0; j < n_cls; ++j) {
            const double x = pr[j];
            qr[j] = (e == 1.0) ? x : (e == 2.0) ? x * x : (e == 0.5) ? sqrt(x) : pow(x, e);
        }
        const double s = np_pairwise_sum<8>(qr, n_cls);
        for (int j = 0; j < n_cls; ++j) qr[j] = qr[j] / s;
    } else {
        for (int j = 0; j < n_cls; ++j) qr[j] = pr[j];
    }
    double run = 0.;
    for (int j = 0; j < n_cls; ++j) {
        run = (j == 0) ? qr[0] : run + qr[j];
        cr[j] = run;
    }
}

// one thread per draw (sample s, residue i)
__global__ void k_draw(const double* __restrict__ c, int64_t n_res, int n_cls, int64_t n_samples, int rng_mode,
                       uint64_t seed, uint64_t rng_offset, const double* __restrict__ uniforms, int32_t* __restrict__ idx,
                       double* __restrict__ r_out, const char* __restrict__ letters, char* __restrict__ letters_out) {
    const int64_t total = n_samples * n_res;
    for (int64_t d = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; d < total; d += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = d % n_res;
        double r;
        if (rng_mode == TH_RNG_PHILOX) {
            rocrand_state_philox4x32_10 st;
            rocrand_init(seed, rng_offset + (uint64_t)d, 0, &st);
            r = rocrand_uniform_double(&st);
        } else {
            r = uniforms[d];
        }
        const double* cr = c + i * n_cls;
        int first = 0;
        for (int j = 0; j < n_cls; ++j) {
            if (cr[j] > r) { first = j; break; }
        }
        idx[d] = first;
        if (r_out) r_out[d] = r;
        if (letters_out) letters_out[d] = letters[first];
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

// Device scratch for one call.  hipMalloc/hipFree cost more than the kernels of a config-5 sized call (a few
// hundred microseconds each vs ~10 us), so the buffers are kept in a small grow-only pool: one cached allocation per
// DevBuf declared in sampler_run (claimed in declaration order), guarded by a mutex that serialises sampler calls.
struct Pool {
    std::mutex mu;
    struct Slot { void* p = nullptr; size_t cap = 0; int dev = -1; } slot[16];
    int next = 0;
};
Pool g_pool;

struct DevBuf {
    void* p = nullptr;
    Pool::Slot* s = nullptr;
    int alloc(size_t bytes) {
        if (!s) s = &g_pool.slot[g_pool.next++ % 16];
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (s->cap < bytes || s->dev != dev || !s->p) {
            if (s->p) { (void)hipSetDevice(s->dev); (void)hipFree(s->p); (void)hipSetDevice(dev); s->p = nullptr; s->cap = 0; }
            const size_t want = std::max<size_t>(bytes, 1 << 16);
            hipError_t e = hipMalloc(&s->p, want);
            if (e != hipSuccess) { s->p = nullptr; th_set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e)); return TH_ENOMEM; }
            s->cap = want; s->dev = dev;
        }
        p = s->p;
        return TH_OK;
    }
};

}  // namespace

int sampler_run(int device, const double* h_probs, int64_t n_res, int n_cls, int64_t n_samples, double temperature,
                int rng_mode, uint64_t seed, uint64_t rng_offset, const double* h_uniforms, int32_t* h_idx,
                double* h_r_out, const char* cat_letters, char* h_letters, double* h_q_out) {
    if (!h_probs || n_res <= 0 || n_cls <= 0 || n_samples < 0) TH_FAIL(TH_EINVAL, "th_sample: bad shape");
    if (temperature == 0.0) TH_FAIL(TH_EINVAL, "th_sample: temperature 0 (the reference divides by it: sampling_utils.py:159)");
    if (rng_mode == TH_RNG_HOST && n_samples > 0 && !h_uniforms) TH_FAIL(TH_EINVAL, "th_sample: rng_mode 0 needs uniforms");
    if (rng_mode < TH_RNG_HOST || rng_mode > TH_RNG_MT19937) TH_FAIL(TH_EINVAL, "th_sample: rng_mode %d", rng_mode);
    if (rng_mode == TH_RNG_MT19937 && seed > 0xffffffffULL) TH_FAIL(TH_EINVAL, "th_sample: MT19937 seed must fit 32 bits");
    if (h_letters && !cat_letters) TH_FAIL(TH_EINVAL, "th_sample: letters_out needs cat_letters");
    std::lock_guard<std::mutex> pool_lock(g_pool.mu);
    g_pool.next = 0;
    HIP_TRY(hipSetDevice(device));
    const int64_t total = n_samples * n_res;
    const size_t pbytes = (size_t)n_res * n_cls * sizeof(double);
    DevBuf dp, dq, dc, du, di, dl, dlo;
    int rc;
    if ((rc = dp.alloc(pbytes)) || (rc = dq.alloc(pbytes)) || (rc = dc.alloc(pbytes))) return rc;
    HIP_TRY(hipMemcpy(dp.p, h_probs, pbytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_temper_cumsum, dim3((unsigned)((n_res + 63) / 64)), dim3(64), 0, 0, (const double*)dp.p, n_res,
                       n_cls, temperature, (n_samples == 0 && h_q_out) ? 1 : 0, (double*)dq.p, (double*)dc.p);
    HIP_TRY(hipGetLastError());
    if (h_q_out) HIP_TRY(hipMemcpy(h_q_out, dq.p, pbytes, hipMemcpyDeviceToHost));
    if (total == 0) { HIP_TRY(hipDeviceSynchronize()); return TH_OK; }
    const bool need_u = rng_mode != TH_RNG_PHILOX;
    if (need_u) {
        if ((rc = du.alloc((size_t)total * sizeof(double)))) return rc;
        if (rng_mode == TH_RNG_HOST) {
            HIP_TRY(hipMemcpy(du.p, h_uniforms, (size_t)total * sizeof(double), hipMemcpyHostToDevice));
        } else {
            hipLaunchKernelGGL(k_mt19937_uniforms, dim3(1), dim3(256), 0, 0, (uint32_t)seed, rng_offset, total,
                               (double*)du.p);
            HIP_TRY(hipGetLastError());
        }
    }
    if ((rc = di.alloc((size_t)total * sizeof(int32_t)))) return rc;
    DevBuf dr;
    if (h_r_out && (rc = dr.alloc((size_t)total * sizeof(double)))) return rc;
    if (h_letters) {
        if ((rc = dl.alloc((size_t)n_cls)) || (rc = dlo.alloc((size_t)total))) return rc;
        HIP_TRY(hipMemcpy(dl.p, cat_letters, (size_t)n_cls, hipMemcpyHostToDevice));
    }
    int64_t blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(k_draw, dim3((unsigned)blocks), dim3(256), 0, 0, (const double*)dc.p, n_res, n_cls, n_samples,
                       rng_mode, seed, rng_offset, (const double*)du.p, (int32_t*)di.p, (double*)dr.p, (const char*)dl.p,
                       (char*)dlo.p);
    HIP_TRY(hipGetLastError());
    if (h_idx) HIP_TRY(hipMemcpy(h_idx, di.p, (size_t)total * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (h_r_out) HIP_TRY(hipMemcpy(h_r_out, dr.p, (size_t)total * sizeof(double), hipMemcpyDeviceToHost));
    if (h_letters) HIP_TRY(hipMemcpy(h_letters, dlo.p, (size_t)total, hipMemcpyDeviceToHost));
    HIP_TRY(hipDeviceSynchronize());
    return TH_OK;
}
