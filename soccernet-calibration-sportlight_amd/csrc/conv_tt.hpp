// Two-team persistent 3x3 convolution (conv_tt.hip): parameters, work items, launcher.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace sncal {

constexpr int TT_MAX_MEMBERS = 3;
constexpr int TT_TH = 8;            // output rows per tile (rows of the STACKED image, see conv_tt.hip)
constexpr int TT_TW = 32;           // output columns per tile (two 16-pixel MFMA fragments)
constexpr int TT_COUT = 96;         // output channels per work item (MI = 6 fragments)
constexpr int TT_CIN = 32;          // input channels per stage (G = 4 k-groups of 8 bf16)

struct TTMember {                   // one convolution of the launch: 3x3, stride 1, pad 1, bf16 NHWC, folded BN (+res)(+ReLU)
    const void* in;                 // [N][H][W][Cin] bf16 (fp8 variant: the e4m3 twin of the tensor, 1 byte per element)
    void* out;                      // [N][H][W][out_cstride] bf16, channels at out_coff (fp8 variant: may be NULL)
    const void* res;                // optional residual, indexed like out
    const void* w;                  // packed weights of the generic conv kernel for (MI = 6, G = 4): [nblk][chunk][9][6][64] x 16 B
    const float* bias;              // folded-BN shift, padded to nblk * 96
    int N, H, W, Cin, chunks;       // chunks = Cin / 32
    int cout, out_cstride, out_coff, relu;
    unsigned w_bytes, in_bytes;     // buffer-descriptor ranges
    unsigned hp1_magic;             // floor(2^32 / (H + 1)) + 1
    // fp8 variant only (conv_tt_kernel<true>): y = acc * oscale[c] + bias[c] with oscale = input scale x weight scale of channel c
    const float* oscale;            // [cout]
    void* out8;                     // optional e4m3 twin of the output, indexed like out at 1 byte per element
    float out8_inv_scale;           // 1 / (per-tensor scale of the output twin)
    int res_split;                  // bf16x3: res points at the residual's SPLIT TWIN ([16 hi | 16 lo] bf16 per 16-channel group, dense), not at fp32
};

struct TTItem {                     // one output tile x one 96-channel block
    uint16_t member, nb;
    int32_t row0;                   // first STACKED output row of the tile (frame f, row y  ->  f * (H + 1) + y)
    int32_t col0;
    int32_t pad_;
};

struct TTParams {
    TTMember m[TT_MAX_MEMBERS];
    const TTItem* items;            // grouped by XCD (workgroup b runs on XCD b % 8), most expensive member first inside an XCD
    const uint32_t* xcd_first;      // [9] offsets into items
    unsigned* queue;                // sixteen zeroed device words owned by the caller's stream: next item per XCD [0..8), teams that have left per XCD [8..16)
                                    // (the kernel re-arms them; launches that share the words must be ordered on one stream)
    unsigned* range;                // fp16x3: sticky counter of wavefronts that split a value beyond the fp16 range (x3.hpp x3_report), or null
    int lazy;                       // 1: small launch -- a workgroup draws its next ticket only when its pair is done (conv_tt_body.inc)
    unsigned long long* trace;      // tuning aid (SNCAL_TT_TRACE=<file>): 256 s_memtime stamps per team, or null
    int ablate;                     // tuning aid (SNCAL_TT_ABLATE, timing only, results invalid): 1 = no epilogue, 2 = no MFMAs, 4 = no DMA
};

void launch_conv_tt(const TTParams& p, int n_wgs, int mode, hipStream_t s, int cfg = 0);      // mode 0 bf16, 1 fp8 (e4m3), 2 x3 (split-bf16, fp32 in / out);
                                                                                                // cfg 0: 96 channels x 8 rows, 1: 64 x 12 (mode 2 only)
constexpr int TT_TABLE_MAX = 760;       // floats per LDS table (bias, output scale)
constexpr int TT_QUEUE_FLOATS = 16;     // the LAST 16 floats of the output-scale table are the ticket queue's four TTItem slots (conv_tt_body.inc q_slot)
constexpr int TT_COUT_MAX = TT_TABLE_MAX - TT_QUEUE_FLOATS;      // so a (grouped) launch may hold at most this many output channels in its tables

}  // namespace sncal
