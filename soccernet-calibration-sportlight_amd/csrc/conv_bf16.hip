// conv_kernel instantiations for __bf16 (see conv.hpp / conv_variants.inc).
#include "conv.hpp"
#include "../../include/sncal.h"

namespace sncal {
static const ConvVariant k_variants_bf16[] = {
#define V(KS, S, NI, MI, G) {SNCAL_BF16, KS, S, NI, MI, G, &conv_launch<__bf16, KS, S, NI, MI, G>, conv_group_fn<__bf16, KS, S, NI, MI, G>()},
#include "conv_variants.inc"
#undef V
};
const ConvVariant* conv_variants_bf16(int* n) {
    *n = (int)(sizeof(k_variants_bf16) / sizeof(k_variants_bf16[0]));
    return k_variants_bf16;
}
}  // namespace sncal
