// The seam between two Bottlenecks of layer1 as ONE kernel, in split (hi + lo) arithmetic on fp32 tensors (fp16x3 / bf16x3 engines):
//
//     y  = ReLU(W3 . h2 + shift3 + res)        conv3 + bn3 + residual + ReLU of block b      (/root/reference/src/models/hrnet/hrnet.py:91-99)
//     h1 = ReLU(W1 . y  + shift1)              conv1 + bn1 + ReLU of block b + 1             (hrnet.py:79-83)
//
// Why: both layers are 1x1 convolutions over 2.07 M pixels (64 frames x 135 x 240) and bound by HBM: as two launches the 256-channel
// fp32 tensor y (2.1 GB) is written by the first and read back by the second (7.4 GB per seam at ~4.4 TB/s = 1.7 ms); fused it is
// written once for the next residual and never re-read (5.3 GB).  VERDICT r3 item 5a.
//
// A 1x1 convolution does not see the image: the kernel walks the pixels linearly in groups of 32, one wavefront per group, no
// synchronisation after the weights are in LDS.  Per group (MFMA 32 x 32 x 16, M = channels, N = the 32 pixels):
//     B fragments of h2 (K = 64: 4 k-steps), split in registers
//     for each 32-channel block mb of y (8 of them):
//         acc  = shift3 ; acc += W3[mb] . h2          4 k-steps x 3 MFMAs (hi.hi + hi.lo + lo.hi)
//         y    = ReLU(acc + res)  -> stored (fp32), and split: with the row order of bnp_pack_weights the lane's 16 values ARE the B
//                fragments of k-steps 2 mb, 2 mb + 1 of the second layer
//         acc1[0..1] += W1[:, 32 mb .. 32 mb + 31] . y        2 blocks x 2 k-steps x 3 MFMAs
//     h1 = ReLU(acc1) -> stored
// 192 MFMAs per group = 0.16 ms of matrix-pipe time per seam chip-wide: the kernel is a memory stream (per pixel 256 + 1024 B read,
// 1024 + 256 B written), the residual of the next block is requested before the current one is multiplied.  All 128 KB of split
// weights stay in LDS (one workgroup of 8 wavefronts per CU, persistent).
#include "common.hpp"
#include "bneckx3.hpp"
#include "x3.hpp"
#include <algorithm>
#include <cstdlib>

#pragma clang fp contract(off)

namespace sncal {
namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int NW = 8;                                  // wavefronts per workgroup (12, three per SIMD, measured 8 % slower: 4.2 against 4.6 TB/s)
constexpr int OFF_W3 = 0, OFF_W1 = BNP_W_BYTES, OFF_B3 = 2 * BNP_W_BYTES, OFF_B1 = OFF_B3 + BNP_WIDE * 4, BNP_LDS = OFF_B1 + BNP_MID * 4;
static_assert(BNP_LDS <= 160 * 1024, "LDS");

__device__ __forceinline__ f32x16 mfma3(const x3h8& ah, const x3h8& al, const x3h8& bh, const x3h8& bl, f32x16 acc) {
    acc = X3_MFMA_32x32x16(al, bh, acc);
    acc = X3_MFMA_32x32x16(ah, bl, acc);
    return X3_MFMA_32x32x16(ah, bh, acc);
}

// NEXT: the seam (the next block's conv1 follows in the same pass).  DS: block 0's tail -- its residual is the downsample branch
// Wds . x0 + shift_ds of the block input x0 (64 channels; hrnet.py:93-94), computed here as four more k-steps of the same accumulator
// instead of a 2.1 GB tensor written by one launch and read by the next; Wds takes W1's place in LDS (all three do not fit).
template <bool DS, bool NEXT>
__global__ __launch_bounds__(64 * NW, 1) void bneck_pair_kernel(const BneckPairParams p) {
    static_assert(!(DS && NEXT), "W3 + Wds + W1 = 192 KB of split weights do not fit the LDS");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    {   // weights and shifts -> LDS, once
        const uint4* w3 = reinterpret_cast<const uint4*>(p.w3);
        const uint4* w1 = reinterpret_cast<const uint4*>(NEXT ? p.w1 : p.wds);
        uint4* d = reinterpret_cast<uint4*>(smem);
        constexpr int N16 = BNP_W_BYTES / 16;
        for (int i = threadIdx.x; i < N16; i += 64 * NW) { d[i] = w3[i]; d[N16 + i] = w1[i]; }
        float* b = reinterpret_cast<float*>(smem + OFF_B3);
        for (int i = threadIdx.x; i < BNP_WIDE; i += 64 * NW) b[i] = DS ? p.b3[i] + p.bds[i] : p.b3[i];
        if (NEXT) for (int i = threadIdx.x; i < BNP_MID; i += 64 * NW) b[BNP_WIDE + i] = p.b1[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long G = (p.P + 31) / 32;
    // fragment (f, hi | lo) of a layer: 1 KB, lane's 16 bytes
    auto frag = [&](int off, int f, int part) { return *reinterpret_cast<const x3h8*>(smem + off + ((f * 2 + part) * 64 + lane) * 16); };
    auto shift8 = [&](int off, int c, float (&v)[8]) {
        const float4 a = *reinterpret_cast<const float4*>(smem + off + c * 4), b = *reinterpret_cast<const float4*>(smem + off + c * 4 + 16);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    };
    // Pixel groups are taken from a device-wide ticket counter, one per wave and round, not dealt statically: the kernel is one
    // workgroup per CU, and a CU that another stream's work holds (the previous batch's camera solves sit on up to 16 + a few CUs for
    // ~15 ms) starts its workgroup late -- with a static deal the launch then lasts until that workgroup has walked its whole share
    // (measured: 1.26 -> 1.85 ms per seam beside the solves).  Tickets go out in address order, so the stream stays sequential in HBM.
    // The ticket for the NEXT round is requested at the top of a round and read at its end.  The last wave to leave re-arms the counter.
    float amax = 0.f;                                  // range tracker (x3.hpp)
    auto take = [&]() -> unsigned { unsigned t = 0; if (lane == 0) t = atomicAdd(p.ticket, 1u); return t; };
    unsigned t_next = take();
    for (;;) {
        const long long g = (long long)(unsigned)__builtin_amdgcn_readfirstlane((int)t_next);
        if (g >= G) break;
        t_next = take();
        const long long px_raw = g * 32 + l31;
        const bool ok = px_raw < p.P;
        const long long px = ok ? px_raw : p.P - 1;
        const float* rp = p.res + px * BNP_WIDE + 8 * hi;
        float* yp = p.y + px * BNP_WIDE + 8 * hi;
        // residual of channel block mb: the lane's channels 32 mb + 16 h + 8 hi + 0..7, h = 0, 1
        float4 r[4], rn[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) r[i] = rn[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!DS) {
#pragma unroll
            for (int h = 0; h < 2; ++h) { r[2 * h] = *reinterpret_cast<const float4*>(rp + 16 * h); r[2 * h + 1] = *reinterpret_cast<const float4*>(rp + 16 * h + 4); }
        }
        // B fragments of h2: lane (pixel l31, K octet hi) holds channels 16 ks + 8 hi + 0..7
        constexpr int KSA = DS ? 8 : 4;                 // k-steps of the first layer: h2, then (DS) x0
        x3h8 bh[KSA], bl[KSA];
#pragma unroll
        for (int ks = 0; ks < KSA; ++ks) {
            const float* hp = (ks < 4 ? p.h2 : p.x0) + px * BNP_MID + 8 * hi + 16 * (ks & 3);
            const float4 a = *reinterpret_cast<const float4*>(hp), b = *reinterpret_cast<const float4*>(hp + 4);
            const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            x3u4 hu, lu;
            x3_split8(v, x3_lower(false), hu, lu);          // (inputs: tracked by the kernels that wrote them)
            bh[ks] = __builtin_bit_cast(x3h8, hu); bl[ks] = __builtin_bit_cast(x3h8, lu);
        }
        f32x16 acc1[2];
        if (NEXT) {
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float v[8];
                    shift8(OFF_B1, 32 * m + 16 * h + 8 * hi, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc1[m][8 * h + e] = v[e];
                }
        }
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) {
            if (!DS && mb < 7) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    rn[2 * h] = *reinterpret_cast<const float4*>(rp + 32 * (mb + 1) + 16 * h);
                    rn[2 * h + 1] = *reinterpret_cast<const float4*>(rp + 32 * (mb + 1) + 16 * h + 4);
                }
            }
            f32x16 acc;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float v[8];
                shift8(OFF_B3, 32 * mb + 16 * h + 8 * hi, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[8 * h + e] = v[e];
            }
#pragma unroll
            for (int ks = 0; ks < KSA; ++ks) {
                const int off = ks < 4 ? OFF_W3 : OFF_W1, f = mb * 4 + (ks & 3);
                acc = mfma3(frag(off, f, 0), frag(off, f, 1), bh[ks], bl[ks], acc);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float rr[8] = {r[2 * h].x, r[2 * h].y, r[2 * h].z, r[2 * h].w, r[2 * h + 1].x, r[2 * h + 1].y, r[2 * h + 1].z, r[2 * h + 1].w};
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = x3_relu(acc[8 * h + e] + rr[e]);
#pragma unroll
                for (int e = 0; e < 8; e += 2) x3_track(amax, v[e], v[e + 1]);      // every OUTPUT is tracked where it is produced (x3.hpp)
                if (ok) {
                    *reinterpret_cast<float4*>(yp + 32 * mb + 16 * h) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(yp + 32 * mb + 16 * h + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
                if (NEXT) {
                    x3u4 yhu, ylu;
                    x3_split8(v, x3_lower(false), yhu, ylu);
                    const x3h8 yh = __builtin_bit_cast(x3h8, yhu), yl = __builtin_bit_cast(x3h8, ylu);
#pragma unroll
                    for (int m = 0; m < 2; ++m) acc1[m] = mfma3(frag(OFF_W1, m * 16 + 2 * mb + h, 0), frag(OFF_W1, m * 16 + 2 * mb + h, 1), yh, yl, acc1[m]);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) r[i] = rn[i];
        }
        if (NEXT && ok) {
            float* op = p.h1 + px * BNP_MID + 8 * hi;
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = x3_relu(acc1[m][8 * h + e]);
#pragma unroll
                    for (int e = 0; e < 8; e += 2) x3_track(amax, v[e], v[e + 1]);
                    *reinterpret_cast<float4*>(op + 32 * m + 16 * h) = make_float4(v[0], v[1], v[2], v[3]);
                    *reinterpret_cast<float4*>(op + 32 * m + 16 * h + 4) = make_float4(v[4], v[5], v[6], v[7]);
                }
        }
    }
    x3_report(amax, p.range);
    if (lane == 0 && atomicAdd(p.ticket + 1, 1u) == gridDim.x * NW - 1u) {      // every wave holds its one failing ticket: nobody touches the counter any more
        p.ticket[0] = 0u; p.ticket[1] = 0u;
        __threadfence();
    }
}

}  // namespace

int launch_bneck_pair_x3(const BneckPairParams& p, int n_cus, hipStream_t s) {
    if (p.P <= 0) return SNCAL_OK;
    const bool ds = p.wds != nullptr;
    if (!p.ticket) { set_error("launch_bneck_pair_x3: no ticket words"); return SNCAL_ERR_ARG; }
    if (ds == (p.w1 != nullptr)) { set_error("launch_bneck_pair_x3: exactly one of the next block's conv1 and the downsample branch"); return SNCAL_ERR_ARG; }
    static bool attr_done = false;
    if (!attr_done) {
        SNCAL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&bneck_pair_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, BNP_LDS));
        SNCAL_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&bneck_pair_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, BNP_LDS));
        attr_done = true;
    }
    const long long G = (p.P + 31) / 32;
    const unsigned blocks = (unsigned)std::min<long long>((G + NW - 1) / NW, n_cus > 0 ? n_cus : 256);
    if (ds) SNCAL_LAUNCH((bneck_pair_kernel<true, false>), dim3(blocks), dim3(64 * NW), (size_t)BNP_LDS, s, p);
    else SNCAL_LAUNCH((bneck_pair_kernel<false, true>), dim3(blocks), dim3(64 * NW), (size_t)BNP_LDS, s, p);
    SNCAL_CHECK_LAUNCH();
    return SNCAL_OK;
}

}  // namespace sncal
