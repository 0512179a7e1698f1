// Operand / result layout of v_mfma_f32_4x4x1_16B_f32 on gfx950, checked empirically:
//   hipcc --offload-arch=gfx950 -O2 tools/microbench/mfma_4x4x1_layout.hip -o /tmp/l && /tmp/l
// Expectation: 16 independent blocks; lane l = 4*b + i supplies A_b[i] and B_b[i]; afterwards VGPR r of lane 4*b + j
// holds D_b[r][j] = A_b[r] * B_b[j].
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) d[l * 4 + r] = acc[r];
}
int main() {
    float ha[64], hb[64], hd[256], *da, *db, *dd;
    for (int l = 0; l < 64; ++l) { ha[l] = 1.f + l; hb[l] = 100.f + 3.f * l; }
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dd, sizeof hd);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int blk = 0; blk < 16; ++blk)
        for (int j = 0; j < 4; ++j)
            for (int r = 0; r < 4; ++r) {
                const float want = ha[4 * blk + r] * hb[4 * blk + j];
                if (hd[(4 * blk + j) * 4 + r] != want) ++bad;
            }
    printf("layout D_b[r][j] in VGPR r of lane 4b+j: %s (%d mismatches)\n", bad ? "NO" : "yes", bad);
    return bad != 0;
}
