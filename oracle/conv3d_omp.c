/* CPU ORACLE (test infrastructure only) — the convolution of oracle/cnn_oracle.py as a blocked DIRECT convolution in C with
 * OpenMP over the host cores: the stand-in for the reference's TensorFlow-CPU path that BASELINE.md B1 specifies ("own C++ /
 * NumPy oracle, OpenMP over all cores") for bench.py's `cpu_baseline` leg.  Same arithmetic contract as cnn_oracle.conv3d —
 * Keras Conv3D, channels-last, cross-correlation, zero padding given as `pad before` per axis, any stride, dilation 1 (reference
 * call site predict.py:142; layer semantics SURVEY.md Appendix A) — fp32 throughout, sums in (tap, ci) order per output.
 * The NumPy im2col + sgemm form materialises a 27x copy of every activation and ran at ~5 % of what the cores sustain; here a
 * frame is copied once into a zero-padded image and every weight vector loaded from L1 feeds XT output voxels.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this; the product never does.
 * Built by oracle/Makefile into oracle/_build/liboracle_conv.so (gcc -O3 -fopenmp; the hot loop is cloned for AVX-512 / AVX2 /
 * baseline x86-64 and picked at run time, so the file built in the container runs on the GPU box's host). */
#include <omp.h>
#include <stdlib.h>
#include <string.h>

typedef float v16f __attribute__((vector_size(64), aligned(4)));
enum { XT = 6, CV = 2 };      /* output voxels along x and 16-float vectors of output channels per register tile */

/* one output row segment: XT voxels x (CV x 16) channels, all taps and input channels */
__attribute__((target_clones("avx512f", "avx2,fma", "default")))
static void tile(const float* pad, long rs, long ps, int Cin, const float* wp, int Coutp, int kd, int kh, int kw, int sw,
                 int nx, int cob, const float* biasp, float* out, int Cout) {
    v16f acc[XT][CV];
    for (int t = 0; t < XT; ++t)
        for (int v = 0; v < CV; ++v) acc[t][v] = *(const v16f*)(biasp + cob + 16 * v);
    for (int dz = 0; dz < kd; ++dz)
        for (int dy = 0; dy < kh; ++dy) {
            const float* xr = pad + dz * ps + dy * rs;
            const float* wr = wp + (long)((dz * kh + dy) * kw) * Cin * Coutp + cob;
            for (int dx = 0; dx < kw; ++dx)
                for (int ci = 0; ci < Cin; ++ci) {
                    const float* wv = wr + (long)(dx * Cin + ci) * Coutp;
                    const v16f w0 = *(const v16f*)wv, w1 = *(const v16f*)(wv + 16);
                    const float* xp = xr + (long)dx * Cin + ci;
                    for (int t = 0; t < XT; ++t) {
                        const float xv = xp[(long)t * sw * Cin];
                        acc[t][0] += xv * w0;
                        acc[t][1] += xv * w1;
                    }
                }
        }
    for (int t = 0; t < nx; ++t)
        for (int v = 0; v < CV; ++v)
            for (int k = 0; k < 16; ++k) {
                const int co = cob + 16 * v + k;
                if (co < Cout) out[(long)t * Cout + co] = acc[t][v][k];
            }
}

/* out[n][zo][yo][xo][co] = bias[co] + sum_{dz,dy,dx,ci} x[n][zo sd + dz - pz][yo sh + dy - py][xo sw + dx - px][ci] w[dz][dy][dx][ci][co]
 * returns 0, or -1 when memory could not be had */
int oracle_conv3d_f32(const float* x, const float* w, const float* bias, float* out, long N, int D, int H, int W, int Cin, int Cout,
                      int kd, int kh, int kw, int sd, int sh, int sw, int pz, int py, int px, int Do, int Ho, int Wo, int threads) {
    const int Coutp = (Cout + 16 * CV - 1) / (16 * CV) * (16 * CV);
    /* padded image large enough for every window the tile loop may touch (XT voxels past the last real one along x) */
    const int Wo_t = (Wo + XT - 1) / XT * XT;
    const long Dp = (long)(Do - 1) * sd + kd, Hp = (long)(Ho - 1) * sh + kh, Wp = (long)(Wo_t - 1) * sw + kw;
    const long DpA = Dp > D + pz ? Dp : D + pz, HpA = Hp > H + py ? Hp : H + py, WpA = Wp > W + px ? Wp : W + px;
    const long rs = WpA * Cin, ps = HpA * rs, img = DpA * ps;
    float* wp = (float*)aligned_alloc(64, sizeof(float) * (size_t)kd * kh * kw * Cin * Coutp);
    float* biasp = (float*)aligned_alloc(64, sizeof(float) * (size_t)Coutp);
    if (!wp || !biasp) { free(wp); free(biasp); return -1; }
    memset(wp, 0, sizeof(float) * (size_t)kd * kh * kw * Cin * Coutp);
    for (long r = 0; r < (long)kd * kh * kw * Cin; ++r) memcpy(wp + r * Coutp, w + r * Cout, sizeof(float) * (size_t)Cout);
    memset(biasp, 0, sizeof(float) * (size_t)Coutp);
    if (bias) memcpy(biasp, bias, sizeof(float) * (size_t)Cout);
    if (threads > 0) omp_set_num_threads(threads);
    int failed = 0;
#pragma omp parallel
    {
        float* pad = (float*)aligned_alloc(64, sizeof(float) * (size_t)(img + 64));
        if (!pad) {
#pragma omp atomic write
            failed = 1;
        }
        long cur = -1;          /* the frame whose padded image this thread holds */
#pragma omp for collapse(2) schedule(dynamic, 1)
        for (long n = 0; n < N; ++n)
            for (int zo = 0; zo < Do; ++zo) {
                if (!pad) continue;
                if (cur != n) {
                    memset(pad, 0, sizeof(float) * (size_t)img);
                    for (int z = 0; z < D; ++z)
                        for (int y = 0; y < H; ++y)
                            memcpy(pad + (z + pz) * ps + (y + py) * rs + (long)px * Cin, x + (((n * D + z) * H + y) * (long)W) * Cin,
                                   sizeof(float) * (size_t)W * Cin);
                    cur = n;
                }
                for (int yo = 0; yo < Ho; ++yo)
                    for (int xo = 0; xo < Wo; xo += XT) {
                        const int nx = Wo - xo < XT ? Wo - xo : XT;
                        const float* base = pad + (long)zo * sd * ps + (long)yo * sh * rs + (long)xo * sw * Cin;
                        float* o = out + ((((n * Do + zo) * Ho + yo) * (long)Wo) + xo) * Cout;
                        for (int cob = 0; cob < Coutp; cob += 16 * CV)
                            tile(base, rs, ps, Cin, wp, Coutp, kd, kh, kw, sw, nx, cob, biasp, o, Cout);
                    }
            }
        free(pad);
    }
    free(wp);
    free(biasp);
    return failed ? -1 : 0;
}
