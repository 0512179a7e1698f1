// Lossless sparse transport of float32 frames (the data the reference builds at design_utils/utils.py:487-530).
// A Gaussian aposteriori frame is ~8 % non-zero, and predict.py from a float32 frame pack was bound by PCIe at 222 KB per frame
// (0.37 of the device-resident rate, round 5).  A frame travels as
//     bitmap  [ceil(E / 32) words, rounded up to 4]   bit k of word w set <=> element 32 w + k is stored
//     values  the stored elements in element order (every element whose BIT PATTERN is not +0.0: -0.0, NaN payloads and
//             denormals are stored, so the expansion is bit-exact)
// and is expanded into the dense [E] float32 frame on the device, in front of the first layer: one workgroup per frame, the
// per-word ranks by a block scan into LDS, then every thread writes whole float4 (64 lanes x 16 bytes = 1 KB contiguous per
// store instruction).  Blob layout (host and device): include/timed_hip.h, th_predict_sparse_async.
#include "common.h"

namespace {

constexpr int kSpThreads = 256;
constexpr int kSpMaxWords = 4096;             // words of bitmap per frame held in LDS (131 072 elements; an aposteriori frame has 1 737)

__global__ void __launch_bounds__(kSpThreads) k_sparse_expand(const uint32_t* __restrict__ bits, const uint64_t* __restrict__ vidx,
                                                              const float* __restrict__ values, float* __restrict__ out, int E,
                                                              int W) {
    __shared__ uint32_t words[kSpMaxWords];
    __shared__ uint32_t rank[kSpMaxWords];    // stored elements in front of word w
    __shared__ uint32_t wave_tot[kSpThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t f = blockIdx.x;
    const uint32_t* const b = bits + f * W;
    const float* const v = values + (vidx[f] - vidx[0]);
    // pass 1: thread t owns words [t per, (t + 1) per): popcounts, exclusive prefix inside the thread, block scan of the totals
    const int per = (W + kSpThreads - 1) / kSpThreads;
    uint32_t mine = 0;
    for (int j = 0; j < per; ++j) {
        const int w = tid * per + j;
        if (w < W) {
            const uint32_t x = b[w];
            words[w] = x;
            rank[w] = mine;
            mine += (uint32_t)__popc(x);
        }
    }
    uint32_t incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t base = incl - mine;
    for (int k = 0; k < wave; ++k) base += wave_tot[k];
    for (int j = 0; j < per; ++j) {
        const int w = tid * per + j;
        if (w < W) rank[w] += base;
    }
    __syncthreads();
    // pass 2: the piece's dense frames are ONE contiguous array (frame f starts at element f E, which is 16-byte aligned only when
    // f E is a multiple of 4): a thread writes the aligned float4 of GLOBAL elements 4 Q .. 4 Q + 3; the quads that straddle the
    // frame's first or last element are written element by element
    const int64_t start = f * (int64_t)E;
    const int64_t Q0 = start >> 2;
    const int nq = (int)(((start + E - 1) >> 2) - Q0) + 1;
    float* const o = out + Q0 * 4;                         // 16-byte aligned (out is)
    const int head = (int)(start - Q0 * 4);                // elements of quad 0 that belong to the previous frame
    for (int q = tid; q < nq; q += kSpThreads) {
        const int e0 = 4 * q - head;                       // this quad's first element inside the frame (may be < 0)
        const int ef = e0 < 0 ? 0 : e0;
        uint32_t r = rank[ef >> 5] + (uint32_t)__popc(words[ef >> 5] & ((1u << (ef & 31)) - 1u));
        float y[4];
        bool ok[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = e0 + i;
            ok[i] = e >= 0 && e < E;
            const bool set = ok[i] && ((words[e >> 5] >> (e & 31)) & 1u);
            y[i] = 0.f;
            if (set) y[i] = v[r++];
        }
        if (ok[0] && ok[3]) {
            *reinterpret_cast<float4*>(o + 4 * q) = make_float4(y[0], y[1], y[2], y[3]);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (ok[i]) o[4 * q + i] = y[i];
        }
    }
}

}  // namespace

// bits [n][W] words, vidx [n + 1] cumulative counts (vidx[0] is the index of values[0]), values: the stored elements of the n
// frames in order; out: the n dense frames of E floats each, contiguous, the first one 16-byte aligned
int launch_sparse_expand(hipStream_t s, int64_t n, const uint32_t* bits, const uint64_t* vidx, const float* values, float* out, int E,
                         int W) {
    if (n <= 0) return TH_OK;
    if (E <= 0 || W < (E + 31) / 32 || W > kSpMaxWords) TH_FAIL(TH_EINVAL, "sparse frames: %d elements in %d bitmap words per frame (at most %d words)", E, W, kSpMaxWords);
    if ((uintptr_t)out % 16) TH_FAIL(TH_EINVAL, "sparse frames: the dense target is not 16-byte aligned");
    hipLaunchKernelGGL(k_sparse_expand, dim3((unsigned)n), dim3(kSpThreads), 0, s, bits, vidx, values, out, E, W);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) TH_FAIL(TH_EHIP, "k_sparse_expand launch failed: %s", hipGetErrorString(e));
    return TH_OK;
}
