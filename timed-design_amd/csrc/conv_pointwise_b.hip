// The chunk-blocked-output (BLK) and straight-line "BN-affine -> ReLU" epilogue (EPI = 2) instantiations of k_conv_pw2 — a second
// translation unit so that they compile next to conv_pointwise.hip's plain kernels, not after them.
#include "conv_pointwise_k.h"

namespace {
// chunk-blocked output, generic chain (no pooling): [kmax index][nt index], and the 12-slot pair
const PwKernel kPw2Blk[3][3] = {{k_conv_pw2<4, 1, 0, 1>, k_conv_pw2<4, 2, 0, 1>, k_conv_pw2<4, 4, 0, 1>},
                                {k_conv_pw2<8, 1, 0, 1>, k_conv_pw2<8, 2, 0, 1>, k_conv_pw2<8, 4, 0, 1>},
                                {k_conv_pw2<16, 1, 0, 1>, k_conv_pw2<16, 2, 0, 1>, nullptr}};
const PwKernel kPw2Blk_12[2] = {k_conv_pw2<12, 1, 0, 1>, k_conv_pw2<12, 2, 0, 1>};
// "BN-affine -> ReLU" epilogue (EPI = 2), no pooling: [blk][kmax index][nt index], and the 12-slot pairs
// (up to 64 output / 96 input channels: the other shapes keep the generic chain)
const PwKernel kPw2Relu[2][3][3] = {{{k_conv_pw2<4, 1, 0, 0, 2>, k_conv_pw2<4, 2, 0, 0, 2>, nullptr},
                                     {k_conv_pw2<8, 1, 0, 0, 2>, k_conv_pw2<8, 2, 0, 0, 2>, nullptr},
                                     {nullptr, nullptr, nullptr}},
                                    {{k_conv_pw2<4, 1, 0, 1, 2>, k_conv_pw2<4, 2, 0, 1, 2>, nullptr},
                                     {k_conv_pw2<8, 1, 0, 1, 2>, k_conv_pw2<8, 2, 0, 1, 2>, nullptr},
                                     {nullptr, nullptr, nullptr}}};
const PwKernel kPw2Relu_12[2][2] = {{k_conv_pw2<12, 1, 0, 0, 2>, k_conv_pw2<12, 2, 0, 0, 2>}, {k_conv_pw2<12, 1, 0, 1, 2>, k_conv_pw2<12, 2, 0, 1, 2>}};
}  // namespace

const void* conv_pw2_extra(int blk, int epi, int kmi, int nti, int slot12) {
    if (blk < 0 || blk > 1 || kmi < 0 || kmi > 2 || nti < 0 || nti > 2) return nullptr;
    if (epi == 2) return (const void*)(slot12 ? (nti <= 1 ? kPw2Relu_12[blk][nti] : nullptr) : kPw2Relu[blk][kmi][nti]);
    if (epi == 0 && blk == 1) return (const void*)(slot12 ? (nti <= 1 ? kPw2Blk_12[nti] : nullptr) : kPw2Blk[kmi][nti]);
    return nullptr;
}
