// Parameters / launcher / weight packing of the fused Bottleneck seam in split (hi + lo) arithmetic (bneckx3.hip): conv3 (64 -> 256, 1x1)
// + residual + ReLU of one Bottleneck and conv1 (256 -> 64, 1x1) + ReLU of the next one as ONE pass over the pixels.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>
#include "head.hpp"      // h32_row_channel

namespace sncal {

constexpr int BNP_MID = 64, BNP_WIDE = 256;                   // the reference's layer1 widths (hrnet.py:273-276: planes 64, expansion 4)
constexpr int BNP_W_BYTES = BNP_MID * BNP_WIDE * 2 * 2;      // one layer's hi + lo fragments: 64 KB

struct BneckPairParams {
    const float* h2;     // [P][64]  fp32: conv2's output of block b (ReLU applied)
    const float* res;    // [P][256] fp32: the residual of block b (its input, or block 0's downsample branch)
    float* y;            // [P][256] fp32: block b's output = ReLU(conv3(h2) + res) -- the next block's residual
    float* h1;           // [P][64]  fp32: ReLU(conv1(y)) of block b + 1
    const void* w3;      // bnp_pack_weights of block b's conv3 (256 x 64)
    const void* w1;      // bnp_pack_weights of block b + 1's conv1 (64 x 256)
    const float* b3;     // folded-BN shifts: 256
    const float* b1;     // 64
    // block 0's tail instead of a seam (w1 = null, h1 unused): the residual is the downsample branch Wds . x0 + bds, computed in the same pass
    const float* x0;     // [P][64] fp32: the block input
    const void* wds;     // bnp_pack_weights of downsample.0 (256 x 64)
    const float* bds;    // 256
    long long P;         // pixels = N * H * W (a 1x1 convolution does not see the image structure)
    unsigned* range;     // fp16x3: sticky counter of wavefronts that split a value beyond the fp16 range (x3.hpp x3_report), or null
    unsigned* ticket;    // two zeroed device words owned by the caller's stream: [0] pixel-group tickets, [1] waves that have left (the kernel re-arms both)
};

int launch_bneck_pair_x3(const BneckPairParams& p, int n_cus, hipStream_t s);

// A fragments of v_mfma_f32_32x32x16_{bf16,f16} for a 1x1 layer (w: [cout][cin] folded weights): [cout / 32][cin / 16][hi | lo][64 lanes][8];
// lane l, row l & 31 carries output channel 32 mb + h32_row_channel(l & 31) (head.hpp: a lane's accumulator registers 8 h .. 8 h + 7 are
// then the eight consecutive channels 32 mb + 16 h + 8 (l >> 5) + 0..7 -- at once the layout of a 32-byte store and of the next
// layer's B fragment), K octet l >> 5 = input channels 16 ks + 8 (l >> 5) + 0..7.  split(w, &hi, &lo) yields the two 16-bit codes.
template <class Split>
inline void bnp_pack_weights(const float* w, const float* scale, int cout, int cin, Split split, std::vector<uint16_t>& out) {
    const int MB = cout / 32, KS = cin / 16;
    out.assign((size_t)MB * KS * 2 * 64 * 8, 0);
    for (int mb = 0; mb < MB; ++mb)
        for (int ks = 0; ks < KS; ++ks)
            for (int lane = 0; lane < 64; ++lane) {
                const int co = 32 * mb + h32_row_channel(lane & 31);
                uint16_t* hi = out.data() + ((((size_t)mb * KS + ks) * 2 + 0) * 64 + lane) * 8;
                uint16_t* lo = out.data() + ((((size_t)mb * KS + ks) * 2 + 1) * 64 + lane) * 8;
                for (int e = 0; e < 8; ++e) {
                    const int ci = 16 * ks + 8 * (lane >> 5) + e;
                    split(w[(size_t)co * cin + ci] * scale[co], &hi[e], &lo[e]);
                }
            }
}

}  // namespace sncal
