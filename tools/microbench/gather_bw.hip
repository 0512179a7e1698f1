// Read bandwidth of k_conv_pw's activation access pattern vs a coalesced one (gfx950).  A "tile" is 32 rows of K floats
// (row stride CS floats).  GATHER: lane (j = l&31, h = l>>5) reads row j, 16 bytes at channel u*8 + 4h for u < K/8 — one
// instruction touches 32 rows x 32 bytes.  COALESCED: lane l reads 16 bytes at flat offset (u*64 + l)*16 of the tile's
// rows — one instruction covers 1 KiB of consecutive memory (8 whole rows when K = 32 and the rows are dense).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gather_bw gather_bw.hip && ./gather_bw
#include <hip/hip_runtime.h>
#include <cstdio>
template <int K8, int MODE>
__global__ void __launch_bounds__(256, 4) k(const float* __restrict__ in, float* out, unsigned ntiles, int cs) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (unsigned tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
        float4 v[K8];
        if (MODE == 0) {
            const float* src = in + ((size_t)tile * 32 + j) * cs + 4 * h;
#pragma unroll
            for (int u = 0; u < K8; ++u) v[u] = *reinterpret_cast<const float4*>(src + u * 8);
        } else {
            // K = 8*K8 floats per row = 2*K8 float4; lane covers float4 index (u*64 + lane) of the tile: row = idx / (2*K8)
#pragma unroll
            for (int u = 0; u < K8; ++u) {
                const int idx = u * 64 + lane, row = idx / (2 * K8), c4 = idx % (2 * K8);
                v[u] = *reinterpret_cast<const float4*>(in + ((size_t)tile * 32 + row) * cs + c4 * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < K8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    if (s.x + s.y + s.z + s.w == 12345.f) out[threadIdx.x] = s.x;
}
template <int K8, int MODE>
void run(const float* in, float* out, unsigned ntiles, int cs, const char* what) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<K8, MODE>), dim3(2048), dim3(256), 0, 0, in, out, ntiles, cs);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<K8, MODE>), dim3(2048), dim3(256), 0, 0, in, out, ntiles, cs);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-10s K=%3d row stride %3d floats: %6.2f TB/s\n", what, K8 * 8, cs, (double)ntiles * 32 * K8 * 32 / (ms * 1e-3) / 1e12);
}
int main() {
    const unsigned rows = 4096u * 1000u, ntiles = rows / 32;
    float *in, *out;
    hipMalloc(&in, (size_t)rows * 176 * 4); hipMalloc(&out, 4096);
    hipMemset(in, 0, (size_t)rows * 176 * 4);
    run<4, 0>(in, out, ntiles, 32, "gather");   run<4, 1>(in, out, ntiles, 32, "coalesced");
    run<4, 0>(in, out, ntiles, 176, "gather");  run<4, 1>(in, out, ntiles, 176, "coalesced");
    run<8, 0>(in, out, ntiles, 176, "gather");  run<8, 1>(in, out, ntiles, 176, "coalesced");
    run<12, 0>(in, out, ntiles, 176, "gather"); run<12, 1>(in, out, ntiles, 176, "coalesced");
    run<12, 0>(in, out, ntiles, 96, "gather");  run<12, 1>(in, out, ntiles, 96, "coalesced");
    return 0;
}
