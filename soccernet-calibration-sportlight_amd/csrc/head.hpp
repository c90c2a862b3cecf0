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
    const float* bias1;            // [LC] bias of last_layer.3 (zero on the padding)
    int nsrc;
    const void* src[HEAD_MAX_SRC]; // t_i = W0_i . branch_i at native resolution, [N][Hs][Ws][HP] bf16
    int Hs[HEAD_MAX_SRC], Ws[HEAD_MAX_SRC];
    float sy[HEAD_MAX_SRC], sx[HEAD_MAX_SRC];
    float* logits;                 // [N][H][W][LC] fp32
    int N, H, W;
    int HP, NQ, LC;                // padded hidden width (800), HP/32, padded class count (M2*16)
    int tiles_x, tiles_y;          // filled by the launcher
    unsigned tiles_x_magic, tiles_y_magic;   // filled by the launcher: floor(2^32 / d) + 1
};

int launch_head_fused(const HeadParams& p, int m2, hipStream_t s);

}  // namespace sncal
