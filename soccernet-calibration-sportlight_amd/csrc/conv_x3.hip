// conv_kernel instantiations for the bf16x3 engine (x3_t: fp32 tensors, split-bf16 arithmetic; see conv.hpp / conv_variants.inc).
#include "conv.hpp"
#include "../../include/sncal.h"

namespace sncal {
static const ConvVariant k_variants_x3[] = {
#define V(KS, S, NI, MI, G) {SNCAL_BF16X3, KS, S, NI, MI, G, &conv_launch<x3_t, KS, S, NI, MI, G>, conv_group_fn<x3_t, KS, S, NI, MI, G>()},
#include "conv_variants.inc"
#undef V
};
void launch_conv_shared_s2_x3(const ConvSharedParams& sp, unsigned blocks, size_t lds, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_shared_s2_kernel<x3_t, 2, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    SNCAL_LAUNCH((conv_shared_s2_kernel<x3_t, 2, 3>), dim3(blocks), dim3(256), lds, s, sp);
}
const ConvVariant* conv_variants_x3(int* n) {
    *n = (int)(sizeof(k_variants_x3) / sizeof(k_variants_x3[0]));
    return k_variants_x3;
}
}  // namespace sncal
