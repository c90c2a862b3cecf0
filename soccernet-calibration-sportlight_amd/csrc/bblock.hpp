// Parameters / launcher of the fused 48-channel BasicBlock kernel (bblock.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace sncal {

struct BBlockParams {
    const void* x;          // [N][H][W][48] bf16 block input (also the residual)
    void* out;              // [N][H][W][48] bf16
    const void* w1;         // conv1 weights, generic conv packing for (MI = 3, G = 3): [2 chunks][7 k-steps][3][64 lanes] x 16 B
    const void* w2;         // conv2 weights, same packing
    const float* b1;        // folded-BN shifts (48 floats each)
    const float* b2;
    int N, H, W;
    int tiles_x, tiles_y;   // filled by the launcher
    int dbg;                // tuning aid (SNCAL_BB_DBG): 1 = drop the output stores, 2 = request the next halo after conv2 (timing only)
    unsigned long long* trace;   // tuning aid (SNCAL_BB_TRACE=<file>): 16 s_memtime stamps per workgroup, or null
    unsigned* ticket;       // nine zeroed device words owned by the caller's stream: tile tickets per XCD [0..8), workgroups that ran dry [8] (re-armed by the kernel)
};

int launch_bblock48(const BBlockParams& p, hipStream_t s);

}  // namespace sncal
