// conv_kernel instantiations for float (see conv.hpp / conv_variants.inc).
#include "conv.hpp"
#include "../../include/sncal.h"

namespace sncal {
static const ConvVariant k_variants_f32[] = {
#define V(KS, S, NI, MI, G) {SNCAL_F32, KS, S, NI, MI, G, &conv_launch<float, KS, S, NI, MI, G>, conv_group_fn<float, KS, S, NI, MI, G>()},
#include "conv_variants.inc"
#undef V
};
const ConvVariant* conv_variants_f32(int* n) {
    *n = (int)(sizeof(k_variants_f32) / sizeof(k_variants_f32[0]));
    return k_variants_f32;
}
}  // namespace sncal
