// Two-team persistent 3x3 stride-1 convolution for the wide HRNet branches (96 / 192 / 384 channels), bf16.
//
// Replaces, for these layers, the conv3x3 + eval-BN (+ residual) (+ ReLU) of BasicBlock
// (/root/reference/src/models/hrnet/hrnet.py:42-58) that the generic kernel (conv.hpp) runs as
// conv<bf16,k3,s1,NI3,MI6,G4>.  Same GEMM view (D[cout, pixel] = sum_k W[cout, k] X[k, pixel], k = chunk of 32 input
// channels -> tap -> channel), same bf16 operands, fp32 accumulation started at the folded-BN shift, same epilogue
// arithmetic.  The MFMA shape is v_mfma_f32_32x32x16_bf16 (the generic kernel: 16x16x32): the products of one output are
// summed in a different order inside the matrix unit, so outputs agree with the generic kernel to fp32 rounding of the
// accumulator, not bit for bit (the first version of this kernel used 16x16x32 and WAS bit-identical; a phase trace
// showed 216 of them take 4.1k clk per wave where 108 of the 32x32x16 take 3.5k -- MI355X_MICROARCH.md: 2075 vs 2382 TF).
//
// Why another kernel.  The generic kernel stages a chunk (54 KB of weights + the halo tile), waits, multiplies, and
// relies on a second resident workgroup to fill the wait.  Measured (NOTES/design_history_r1_r5.md §4): the two workgroups drift into
// phase, the LDS-DMA issue of one (~100 clk per 1 KB piece on the issuing wave) lands on top of its own MFMA phase
// as often as under the other's, and the MFMA pipe is busy 40 % of the kernel although the work is MFMA-bound.
// Here the alternation is BUILT IN:
//   * one workgroup of 8 waves per CU = two TEAMS of 4 waves; wave i of team A and wave i of team B share SIMD i;
//   * one team MULTIPLIES the stage it has in LDS (9 k-steps x 24 MFMAs per wave, one wave per SIMD issuing back to
//     back) while the other team LOADS its next stage by LDS-DMA (its waves carry the whole issue cost and the wait);
//     a token in LDS makes the two MFMA phases mutually exclusive, so the roles swap by themselves (see the main loop);
//   * a team's tile is 8 x 32 output pixels x 96 output channels (4 x 6 accumulator fragments per wave), a stage is
//     32 input channels: 54 KB of weights + a 10 x 34 pixel halo (22.5 KB); two teams = 160 KB of LDS, all of it;
//   * the epilogue of a finished tile runs at the head of the team's next LOAD phase; each wave transposes its fragments
//     through the block of the weight region that it alone refills afterwards, so no team-level barrier is needed;
//   * weights per MAC drop by a quarter against the 192-pixel tile of the generic kernel, and no DMA instruction is
//     ever issued by a wave that has MFMAs to issue.
// Frames are STACKED: output row index s = f * (H + 1) + y, every frame followed by one shared all-zero row (it is the
// bottom padding of frame f and the top padding of frame f + 1).  Tiles of 8 stacked rows then cut the whole batch with
// 1 / (H + 1) waste instead of (ceil(H / 8) * 8 - H) / H per frame -- 6 % instead of 41 % on the 17 x 30 branch.
// The halo image in LDS is [pixel][4 x 16 B] at a 36-pixel row pitch with the four 8-channel groups of a pixel rotated
// by (pixel >> 2) & 3: every ds_read_b128 lane group of a B fragment (32 consecutive pixels, one channel group per
// half-wave) then touches 16 distinct 16-byte bank slots (enumerated against MI355X_MICROARCH.md "LDS"), without the 50 % pitch padding of the generic kernel; the
// rotation is applied on the DMA's per-lane SOURCE address, the LDS destination stays lane-linear.
// Work items (tile x 96-channel block, cost = Cin / 32 stages) of up to three member convolutions are dealt to the
// 2 x 256 teams on the host (conv_tt_plan in hrnet.cpp): contiguous slices per XCD, longest-processing-time first.
#include "common.hpp"
#include "conv_tt.hpp"
#include "x3.hpp"
#include <cstddef>

namespace sncal {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((address_space(3))) void lds_void;

// member m's parameters straight from the kernel-argument segment, by scalar loads at a computed offset (selecting among three
// by-value copies of the argument keeps ~80 SGPRs alive and spills; a dynamically indexed argument is copied to scratch)
__device__ __forceinline__ TTMember load_member(int m) {
    TTMember r;
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(sizeof(TTMember) % 4 == 0 && offsetof(TTParams, m) == 0, "the members must open the kernel-argument segment");
    const __attribute__((address_space(4))) unsigned* src =
        (const __attribute__((address_space(4))) unsigned*)__builtin_amdgcn_kernarg_segment_ptr() + m * (int)(sizeof(TTMember) / 4);
    unsigned* dst = reinterpret_cast<unsigned*>(&r);
#pragma unroll
    for (unsigned i = 0; i < sizeof(TTMember) / 4; ++i) dst[i] = src[i];
#else
    (void)m;
    r = TTMember{};
#endif
    return r;
}

// ---- tile configurations: (MB, NB) = (3, 2): 96 output channels x 8 stacked rows x 32 columns, every mode;
//      (2, 3): 64 x 12 x 32 for the bf16x3 engine's 48-channel branch (25 % instead of 50 % of padded rows)
#define TT_CFG_NS c32
#define TT_CFG_MB 3
#define TT_CFG_NB 2
#include "conv_tt_body.inc"
#undef TT_CFG_NS
#undef TT_CFG_MB
#undef TT_CFG_NB
#define TT_CFG_NS c23
#define TT_CFG_MB 2
#define TT_CFG_NB 3
#include "conv_tt_body.inc"
#undef TT_CFG_NS
#undef TT_CFG_MB
#undef TT_CFG_NB

// (3, 1): 96 x 4 x 32, split arithmetic only -- SMALL launches (round 5).  Below about two items per team the launch lasts as long as its
// longest item (twelve stages on the 384-channel branch) while most teams hold three-stage items or nothing: at batch 8, 98 us per
// grouped launch for 40 us of work per team.  Half the rows per item = twice the items at half a stage's multiplies each; the weight
// stream per stage is the same (L2 serves it; at these sizes nothing is bandwidth-bound).
#define TT_CFG_NS c31
#define TT_CFG_MB 3
#define TT_CFG_NB 1
#include "conv_tt_body.inc"
#undef TT_CFG_NS
#undef TT_CFG_MB
#undef TT_CFG_NB

void launch_conv_tt(const TTParams& p, int n_wgs, int mode, hipStream_t s, int cfg) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&c32::conv_tt_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&c32::conv_tt_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&c32::conv_tt_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&c23::conv_tt_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&c31::conv_tt_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    if (cfg == 2) {                                    // 96 output channels x 4 rows x 32 columns (split arithmetic, small launches)
        const size_t lds = (size_t)2 * c31::TEAM_BYTES + 64 + 2 * TT_TABLE_MAX * 4;
        SNCAL_LAUNCH(c31::conv_tt_kernel<2>, dim3((unsigned)n_wgs), dim3(512), lds, s, p);
        return;
    }
    if (cfg == 1) {                                    // 64 output channels x 12 rows x 32 columns (bf16x3: the 48-channel branch)
        const size_t lds = (size_t)2 * c23::TEAM_BYTES + 64 + 2 * TT_TABLE_MAX * 4;
        SNCAL_LAUNCH(c23::conv_tt_kernel<2>, dim3((unsigned)n_wgs), dim3(512), lds, s, p);
        return;
    }
    const size_t lds = (size_t)2 * c32::TEAM_BYTES + 64 + 2 * TT_TABLE_MAX * 4;
    if (mode == 1) SNCAL_LAUNCH(c32::conv_tt_kernel<1>, dim3((unsigned)n_wgs), dim3(512), lds, s, p);
    else if (mode == 2) SNCAL_LAUNCH(c32::conv_tt_kernel<2>, dim3((unsigned)n_wgs), dim3(512), lds, s, p);
    else SNCAL_LAUNCH(c32::conv_tt_kernel<0>, dim3((unsigned)n_wgs), dim3(512), lds, s, p);
}

}  // namespace sncal
