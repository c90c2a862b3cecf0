// Parameters / launcher / weight packing of the fused 48-channel BasicBlock in split (hi + lo) arithmetic (bblockx3.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>

namespace sncal {

constexpr int BBX_STEPS = 14;                       // pair-steps per convolution (27 (tap, 16-channel group) units + one zero unit)
constexpr int BBX_STEP_BYTES = 6 * 1024;            // [unit 0: 3 blocks of 16 output channels][unit 1: 3 blocks] x 1 KB A fragments
constexpr int BBX_W_BYTES = BBX_STEPS * BBX_STEP_BYTES;

struct BBlockX3Params {
    const void* x;          // split twin of the block input: [N][H][W][3 groups][16 hi | 16 lo] 16-bit = 192 B per pixel (also the residual)
    void* out_twin;         // optional: split twin of the block output, same layout
    float* out;             // optional: fp32 output [N][H][W][out_cstride], channels at out_coff
    const void* w1;         // conv1 weights, bbx3_pack_weights
    const void* w2;         // conv2 weights
    const float* b1;        // folded-BN shifts (48 floats each)
    const float* b2;
    int N, H, W;
    int out_cstride, out_coff;
    int tiles_x, tiles_y;   // filled by the launcher (tiles_y: tile rows of the N (H + 1) stacked rows)
    unsigned h1_magic;      // filled by the launcher: floor(2^32 / (H + 1)) + 1 (stacked row -> frame by one multiplication)
    int run_max;            // filled by the launcher: longest run of vertically adjacent tiles per ticket (SNCAL_BBX_RUNS, default 8; 1 = single tiles)
    int dbg;                // tuning aid (SNCAL_BBX_DBG, timing only -- results are wrong): 1 = drop the output stores, 2 = no weight requests after the first steps, 4 = no halo requests
    unsigned long long* trace;   // tuning aid (SNCAL_BBX_TRACE=<file>): 8 clock sums per wave, or null
    unsigned* range;        // fp16x3: sticky counter of wavefronts that split a value beyond the fp16 range (x3.hpp x3_report), or null
    unsigned* ticket;       // nine zeroed device words owned by the caller's stream: tile tickets per XCD [0..8), workgroups that ran dry [8] (re-armed by the kernel)
};

int launch_bblockx3(const BBlockX3Params& p, hipStream_t s);

// Pair-step order of the K loop (shared by the kernel and the packing).  Unit = (tap (dy, dx), 16-channel group g); step s multiplies
// units u0(s) and u1(s): the two cross terms of each on its own (w_hi.x_lo + w_lo.x_hi in ONE K = 32 MFMA) and the two main terms
// together (w_hi(u0).x_hi(u0) + w_hi(u1).x_hi(u1) in one).  The partner of a unit is chosen so that the address distance between
// the two units' pixels is one of three constants (next row / next pixel / next group): the lanes that hold the second unit's half of
// a main-term B fragment carry that distance in their base address.
//   s = 0..8   : (0, dx, g) + (1, dx, g),  dx = s / 3, g = s % 3        (next row)
//   s = 9..11  : (2, 0, g)  + (2, 1, g),   g = s - 9                    (next pixel)
//   s = 12     : (2, 2, 0)  + (2, 2, 1)                                 (next group)
//   s = 13     : (2, 2, 2)  + a zero unit
struct BbxUnit { int dy, dx, g; };      // g < 0: the zero unit
constexpr BbxUnit bbx_unit(int s, int which) {
    if (s < 9) return BbxUnit{which, s / 3, s % 3};
    if (s < 12) return BbxUnit{2, which, s - 9};
    if (s == 12) return BbxUnit{2, 2, which};
    return which == 0 ? BbxUnit{2, 2, 2} : BbxUnit{0, 0, -1};
}

// A fragments of v_mfma_f32_16x16x32_{bf16,f16} for one convolution (w: [48][48][3][3] folded weights): per step and unit three 1 KB
// fragments (16 output channels each); lane l holds output channel 16 cb + (l & 15) and K octet l >> 4: octets 0, 1 = hi parts of
// input channels 16 g + 0..7 / 8..15 of the tap, octets 2, 3 = their lo parts.  (The main-term fragment [w_hi(u0) | w_hi(u1)] is read
// out of the two units' fragments with a per-lane address, it is not stored.)  split(w, &hi, &lo) yields the two 16-bit codes.
template <class Split>
inline void bbx3_pack_weights(const float* w, const float* scale, Split split, std::vector<uint16_t>& out) {
    out.assign((size_t)BBX_W_BYTES / 2, 0);
    for (int s = 0; s < BBX_STEPS; ++s)
        for (int which = 0; which < 2; ++which) {
            const BbxUnit u = bbx_unit(s, which);
            if (u.g < 0) continue;
            for (int cb = 0; cb < 3; ++cb)
                for (int lane = 0; lane < 64; ++lane) {
                    const int co = cb * 16 + (lane & 15), o = lane >> 4;
                    uint16_t* dst = out.data() + ((size_t)s * BBX_STEP_BYTES + (size_t)which * 3072 + (size_t)cb * 1024 + (size_t)lane * 16) / 2;
                    for (int e = 0; e < 8; ++e) {
                        const int ci = u.g * 16 + (o & 1) * 8 + e;
                        const float v = w[(((size_t)co * 48 + ci) * 3 + u.dy) * 3 + u.dx] * scale[co];
                        uint16_t hi, lo;
                        split(v, &hi, &lo);
                        dst[e] = o < 2 ? hi : lo;
                    }
                }
        }
}

}  // namespace sncal
