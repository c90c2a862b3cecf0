// Parameters / launcher of the fused head kernel (head.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace sncal {

constexpr int HEAD_MAX_SRC = 5;
constexpr int HEAD_MAX_FOLD = 2;

struct HeadParams {
    const void* direct;            // [N][H][W][Cd] bf16: the tensor that already sits at head resolution
    int Cd;                        // its channel count (<= 64)
    int nfold;                     // narrow branches whose upsampled channels are appended to the stage-1 K dimension
    const void* fold[HEAD_MAX_FOLD];   // [N][Hf][Wf][Cf] bf16
    int Cf[HEAD_MAX_FOLD], Hf[HEAD_MAX_FOLD], Wf[HEAD_MAX_FOLD];
    float fsy[HEAD_MAX_FOLD], fsx[HEAD_MAX_FOLD];
    int ks1;                       // stage-1 k-steps: ceil((Cd + sum Cf) / 32), rounded up to an instantiated depth
    const void* w0;                // stage-1 A fragments [NQ][2][ks1][64 lanes] x 16 B (rows permuted, BN scale folded)
    const float* bias0;            // [HP] folded BN shift of last_layer.0 (zero on the padding)
    const void* w1;                // stage-2 A fragments [NQ][M2][64 lanes] x 16 B
    const void* w0_32;             // head32.hip: stage-1 A fragments of v_mfma_f32_32x32x16 [NQ][ks16][64 lanes] x 16 B (rows in tt_row_channel order), or null
    const void* w1_32;             // head32.hip: stage-2 A fragments [NQ][ceil(LC / 32)][2][64 lanes] x 16 B (class rows in tt_row_channel order)
    const void* w0_32_lo;          // headx3.hip (bf16x3 engine): the lo parts of the split weights, same layouts (w0_32 / w1_32 hold the hi parts)
    const void* w1_32_lo;
    int ks16;                      // (Cd + sum Cf) / 16 when that is exact, else 0
    const float* bias1;            // [LC] bias of last_layer.3 (zero on the padding)
    int nsrc;
    const void* src[HEAD_MAX_SRC]; // t_i = W0_i . branch_i at native resolution, [N][Hs][Ws][HP] bf16
    int Hs[HEAD_MAX_SRC], Ws[HEAD_MAX_SRC];
    float sy[HEAD_MAX_SRC], sx[HEAD_MAX_SRC];
    float* logits;                 // [N][H][W][LC] fp32
    // head32.hip, decode-fused form (nobody wants the heatmap): log-softmax and the tile's row / column maxima in the epilogue, no logits
    // written.  dec_row [N][C-1][H][tiles_x] and dec_col [N][tiles_y][C-1][W] feed kp_finish (decode.hip); both null = write logits
    float* dec_row;
    float* dec_col;
    int dec_C;                     // real class count incl. background (58)
    int N, H, W;
    int HP, NQ, LC;                // padded hidden width (800), HP/32, padded class count (M2*16)
    int tiles_x, tiles_y;          // filled by the launcher
    unsigned tiles_x_magic, tiles_y_magic;   // filled by the launcher: floor(2^32 / d) + 1
    int stage_folds;               // headx3.hip, filled by the launcher: the folded branches' source boxes of a tile go through LDS once (1) or every lane fetches its taps (0)
    unsigned* range;               // headx3.hip (fp16x3): sticky counter of wavefronts that split a value beyond the fp16 range (x3.hpp), or null
    unsigned long long* trace;     // head32.hip tuning aid (SNCAL_HEAD_TRACE=<file>): 8 phase sums per workgroup, or null
};

// head32.hip: row order of a 32-row block of A fragments.  The D registers of v_mfma_f32_32x32x16 give lane l rows 8 q + 4 (l >> 5) + j
// (q, j = 0..3) of column l & 31; MFMA row r carries channel h32_row_channel(r) of the block, so that the registers 8 h .. 8 h + 7 of a
// lane are the eight consecutive channels 16 h + 8 (l >> 5) + 0..7.
constexpr int h32_row_channel(int r) { return 16 * (r >> 4) + 8 * ((r >> 2) & 1) + 4 * ((r >> 3) & 1) + (r & 3); }

int launch_head_fused(const HeadParams& p, int m2, hipStream_t s);
bool launch_head32(const HeadParams& p, hipStream_t s);      // head32.hip; false = does not apply, nothing launched
bool head32_applies(const HeadParams& p);                  // the same test without launching
size_t head32_decode_scratch(int B, int C, int h, int w);   // bytes of dec_row + dec_col
void head32_decode_parts(int h, int w, int* row_parts, int* col_parts);
// headx3.hip: the same head in split-bf16 arithmetic on fp32 tensors (bf16x3 engine); direct / fold / src are fp32 there
bool launch_headx3(const HeadParams& p, hipStream_t s);
bool headx3_applies(const HeadParams& p);

}  // namespace sncal
