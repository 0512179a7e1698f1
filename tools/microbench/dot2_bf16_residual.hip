// Is x - bf16(x) through v_dot2_f32_bf16 (x + piece.lo * (-1) + piece.hi * 0: ONE instruction per value) bit-identical to the
// shift / mask / subtract form, and does v_pk_add_f32 with neg modifiers do the same subtraction?  Both: yes, on 65 536 values
// incl. zeros and 1e-20-scale ones.  (In a real kernel the dot2 form still lost: DESIGN.md 4.1e — the operand (bf16 -1, 0) must
// come from an SGPR, not hipcc's inline constant -1.0; a DOT result needs 3 wait states before a VALU reads it, which inline asm
// hides from the hazard recognizer; and next to MFMAs it was slower than the 4-instruction form.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/dot2_bf16_residual.hip -o /tmp/d2 && /tmp/d2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk(v2f x) { return __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2)); }
__device__ __forceinline__ v2f unpk(unsigned w) { return (v2f){__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xffff0000u)}; }
__global__ void k(const float* in, float* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    v2f x = {in[2 * i], in[2 * i + 1]};
    unsigned h = pk(x);
    v2f ref = x - unpk(h);
    v2f r1, r2;
    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r1.x) : "v"(h), "s"(0x0000bf80u), "v"(x.x));
    asm volatile("v_dot2_f32_bf16 %0, %1, %2, %3" : "=v"(r1.y) : "v"(h), "s"(0xbf800000u), "v"(x.y));
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r2) : "v"(x), "v"(unpk(h)));
    out[6 * i] = ref.x; out[6 * i + 1] = ref.y; out[6 * i + 2] = r1.x; out[6 * i + 3] = r1.y; out[6 * i + 4] = r2.x; out[6 * i + 5] = r2.y;
}
int main() {
    const int n = 1 << 16;
    float* h = (float*)malloc(n * 4); float* o = (float*)malloc(n * 12);
    srand(1);
    for (int i = 0; i < n; ++i) { float u = (rand() / (float)RAND_MAX - 0.5f) * 8.f; h[i] = (i % 7 == 0) ? 0.f : (i % 5 == 0 ? u * 1e-20f : u); }
    float *d, *dout; hipMalloc(&d, n * 4); hipMalloc(&dout, n * 12);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, d, dout, n);
    hipMemcpy(o, dout, n * 12, hipMemcpyDeviceToHost);
    int bad1 = 0, bad2 = 0;
    for (int i = 0; i < n / 2; ++i) for (int c = 0; c < 2; ++c) {
        if (o[6 * i + c] != o[6 * i + 2 + c]) { if (bad1 < 5) printf("dot2: x=%g ref=%g got=%g\n", h[2 * i + c], o[6 * i + c], o[6 * i + 2 + c]); ++bad1; }
        if (o[6 * i + c] != o[6 * i + 4 + c]) { if (bad2 < 5) printf("pk: x=%g ref=%g got=%g\n", h[2 * i + c], o[6 * i + c], o[6 * i + 4 + c]); ++bad2; }
    }
    printf("dot2 mismatches %d, pk_add mismatches %d of %d\n", bad1, bad2, n);
    return 0;
}
