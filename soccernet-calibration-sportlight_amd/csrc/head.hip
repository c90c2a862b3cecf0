// Fused HRNet head (bf16 path): upsample + concat + 1x1 conv + BN + ReLU + 1x1 conv in one kernel.
//
// Reference formulation (/root/reference/src/models/hrnet/hrnet.py:489-510, :316-329; line net
// /root/reference/src/models/line/hrnet.py:236-248, :86-102): every branch is bilinearly upsampled
// (align_corners=True) to the head resolution, concatenated with the stem features into a 784 (720)-channel
// tensor -- 203 MB per frame in bf16 at 270x480 -- which then goes through conv1x1(784->784)+BN+ReLU and
// conv1x1(784->58).  That single 784x784 GEMM is 31 % of the network's MACs.
//
// Restructuring used here (declared in DESIGN.md; SURVEY 8d "algebraic shortcut"): a 1x1 convolution commutes
// with bilinear interpolation (both are linear, the interpolation weights sum to one), so
//     W0 . concat(direct, up(b_i)...) = W0_d . direct + sum_i up(W0_i . b_i).
// The per-branch products t_i = W0_i . b_i run at the branches' NATIVE resolutions through the generic MFMA
// conv kernel (8.8 GMAC instead of 79.7), and this kernel finishes the head per output pixel without ever
// materialising the 784-channel tensors:
//     stage 1  MFMA   h = W0_d . direct            (K = 64 stem / 48 branch-0 channels)
//     gather   VALU   h += sum_i bilinear(t_i)     (4 taps x up to 5 sources; the 32-channel slice of every
//                                                   source box of the 8x16 tile is DMA'd to LDS, double-buffered)
//              VALU   h = relu(h + folded-BN shift)
//     stage 2  MFMA   logits += W1[:, 32-slice] . h
// Hidden channels are walked in slices of 32.  The rows of the stage-1 A fragments are permuted so that the
// lane that owns output (pixel, channels 8g..8g+7) in the stage-1 accumulators is exactly the lane that must
// hold the same 8 k-values in the stage-2 B fragment: the hand-off between the two GEMMs is a register
// repack -- no LDS round trip, no shuffles.  One barrier per 32-channel slice orders the source-box DMA.
#include "common.hpp"
#include "head.hpp"
#include <cstdlib>

#pragma clang fp contract(fast)

namespace sncal {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

typedef __attribute__((address_space(3))) void lds_void;
constexpr int HEAD_TH = 4;                 // tile = 4 rows x 16 columns of head pixels; one wave owns one row
constexpr int HEAD_SRC_LDS = 3072;         // LDS bytes per source per q-slice (<= 48 source pixels x 64 B)
constexpr int HEAD_MAX_DMA = 4;            // DMA instructions per wave per slice (<= 16 over the block)

template <int M2, int NSRC>
__global__ __launch_bounds__(256) void head_fused_kernel(const HeadParams p) {
    constexpr int KS1 = 2;
    constexpr int OFF_W0 = NSRC * HEAD_SRC_LDS, OFF_W1 = OFF_W0 + 2 * KS1 * 1024, OFF_B0 = OFF_W1 + M2 * 1024;
    constexpr int HEAD_BUF = OFF_B0 + 1024;     // per q-slice: source boxes, stage-1 / stage-2 A fragments, BN shift
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 2 * HEAD_BUF bytes
    const int lane = threadIdx.x & 63, g = lane >> 4, ln = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tile = blockIdx.x;
    const int tx = tile % p.tiles_x; tile /= p.tiles_x;
    const int ty = tile % p.tiles_y;
    const int n = tile / p.tiles_y;
    const int oy0 = ty * HEAD_TH, ox0 = tx * 16;
    const __bf16* direct = reinterpret_cast<const __bf16*>(p.direct);

    // block-uniform source boxes (the taps of every pixel of the tile fall inside).  The DMA work list of this
    // wave (which instruction of which source, per-lane byte offset) does not depend on the channel slice, so
    // it is computed once; the slice only moves the scalar offset.
    int by0[NSRC], bx0[NSRC], bw[NSRC];
    unsigned dma_voff[HEAD_MAX_DMA];
    int dma_src[HEAD_MAX_DMA], dma_lds[HEAD_MAX_DMA];
#pragma unroll
    for (int k = 0; k < HEAD_MAX_DMA; ++k) { dma_voff[k] = 0x80000000u; dma_src[k] = -1; dma_lds[k] = 0; }
    {
        int cursor = 0;     // running instruction index over all sources (block-uniform)
#pragma unroll
        for (int s = 0; s < NSRC; ++s) {
            const int ylast = min(oy0 + HEAD_TH - 1, p.H - 1), xlast = min(ox0 + 15, p.W - 1);
            by0[s] = (int)(p.sy[s] * (float)oy0); bx0[s] = (int)(p.sx[s] * (float)ox0);
            const int by1 = min((int)(p.sy[s] * (float)ylast) + 1, p.Hs[s] - 1);
            const int bx1 = min((int)(p.sx[s] * (float)xlast) + 1, p.Ws[s] - 1);
            bw[s] = bx1 - bx0[s] + 1;
            const int npx = (by1 - by0[s] + 1) * bw[s];
            const int ninstr = (npx * 4 + 63) / 64;
            for (int i = 0; i < ninstr; ++i, ++cursor) {
                if ((cursor & 3) != wave) continue;
                const int k = cursor >> 2;
                const int slot = i * 64 + lane, pi = slot >> 2, piece = slot & 3;
                const int ly = pi / bw[s], lx = pi - ly * bw[s];
                const unsigned v = pi < npx ? (unsigned)((((by0[s] + ly) * p.Ws[s] + bx0[s] + lx) * p.HP) * 2 + piece * 16) : 0x80000000u;
#pragma unroll
                for (int kk = 0; kk < HEAD_MAX_DMA; ++kk)
                    if (kk == k) { dma_voff[kk] = v; dma_src[kk] = s; dma_lds[kk] = s * HEAD_SRC_LDS + i * 1024; }
            }
        }
    }
    // everything the slice loop consumes comes through LDS-DMA: an ordinary global load inside the loop would
    // make hipcc wait vmcnt(0) at its first use and drain the prefetch every iteration
    const __amdgpu_buffer_rsrc_t rs_w0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w0), 0, p.NQ * 2 * KS1 * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w1), 0, p.NQ * M2 * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias0), 0, p.HP * 4, 0x00020000);
    auto issue_slice = [&](int q, int buf) {
        char* const base = smem + buf * HEAD_BUF;
#pragma unroll
        for (int k = 0; k < HEAD_MAX_DMA; ++k) {
#pragma unroll
            for (int s = 0; s < NSRC; ++s)
                if (dma_src[k] == s) {
                    // one image of source s (ranges stay < 2 GB); building the descriptor is 4 scalar moves
                    const size_t img = (size_t)p.Hs[s] * p.Ws[s] * p.HP * 2;
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<char*>(reinterpret_cast<const char*>(p.src[s])) + (size_t)n * img, 0, (int)img, 0x00020000);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(base + dma_lds[k]), 16,
                                                             dma_voff[k], (unsigned)(q * 64), 0, 0);
                }
        }
        // A fragments of the slice: wave w brings stage-1 piece w (4 pieces) and stage-2 piece w (M2 pieces)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w0, (lds_void*)(base + OFF_W0 + wave * 1024), 16, (unsigned)(lane * 16),
                                                 (unsigned)((q * 2 * KS1 + wave) * 1024), 0, 0);
        if (wave < M2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w1, (lds_void*)(base + OFF_W1 + wave * 1024), 16, (unsigned)(lane * 16),
                                                     (unsigned)((q * M2 + wave) * 1024), 0, 0);
        if (wave == 3)      // 32 shift values = 128 B; the other lanes read out of range
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b0, (lds_void*)(base + OFF_B0), 16,
                                                     lane < 8 ? (unsigned)(lane * 16) : 0x80000000u, (unsigned)(q * 128), 0, 0);
    };
    issue_slice(0, 0);

    // per-lane pixel bookkeeping: wave owns row `wave` of the tile; lane column = ln
    const int y = oy0 + wave, x = ox0 + ln;
    const bool valid = y < p.H && x < p.W;
    const int yc = min(y, p.H - 1), xc = min(x, p.W - 1);
    const long pix = ((long)n * p.H + yc) * p.W + xc;
    unsigned lo00[NSRC], ldx[NSRC], ldy[NSRC];     // LDS byte offsets of the taps
    bf16x2 wtop[NSRC], wbot[NSRC];     // (w00, w01) and (w10, w11) as bf16 pairs for v_dot2c_f32_bf16
#pragma unroll
    for (int s = 0; s < NSRC; ++s) {
        const float fy = p.sy[s] * (float)yc, fx = p.sx[s] * (float)xc;   // PyTorch align_corners=True index
        int iy = (int)fy, ix = (int)fx;
        iy = iy > p.Hs[s] - 1 ? p.Hs[s] - 1 : iy;
        ix = ix > p.Ws[s] - 1 ? p.Ws[s] - 1 : ix;
        const float ly1 = fy - (float)iy, lx1 = fx - (float)ix;
        wtop[s][0] = (__bf16)((1.f - lx1) * (1.f - ly1)); wtop[s][1] = (__bf16)(lx1 * (1.f - ly1));
        wbot[s][0] = (__bf16)((1.f - lx1) * ly1); wbot[s][1] = (__bf16)(lx1 * ly1);
        lo00[s] = (unsigned)(s * HEAD_SRC_LDS + ((iy - by0[s]) * bw[s] + (ix - bx0[s])) * 64 + g * 16);
        ldx[s] = ix < p.Ws[s] - 1 ? 64u : 0u;
        ldy[s] = iy < p.Hs[s] - 1 ? (unsigned)(bw[s] * 64) : 0u;
    }
    bf16x8 bD[KS1];
#pragma unroll
    for (int ks = 0; ks < KS1; ++ks) {
        const int ch = ks * 32 + g * 8;
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (ch < p.Cd) v = *reinterpret_cast<const bf16x8*>(direct + pix * p.Cd + ch);
        bD[ks] = v;
    }

    f32x4 acc2[M2];
#pragma unroll
    for (int mi = 0; mi < M2; ++mi) acc2[mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int q = 0; q < p.NQ; ++q) {
        const int buf = q & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my DMA pieces of slice q landed
        asm volatile("s_barrier" ::: "memory");              // everyone's did; everyone is done with slice q-1
        if (q + 1 < p.NQ) issue_slice(q + 1, buf ^ 1);       // lands while slice q is consumed
        const char* const sb = smem + buf * HEAD_BUF;
        // ---- stage 1: 32 hidden channels x 16 pixels, K = direct channels ------------------------------
        f32x4 acc1[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int ks = 0; ks < KS1; ++ks) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(sb + OFF_W0 + ((f * KS1 + ks) * 64 + lane) * 16);
                acc1[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bD[ks], acc1[f], 0, 0, 0);
            }
        // ---- gather (from LDS) + folded BN shift + ReLU: lane owns channels q*32 + g*8 .. +7 of its pixel ----
        const float4 bs0 = *reinterpret_cast<const float4*>(sb + OFF_B0 + g * 32);
        const float4 bs1 = *reinterpret_cast<const float4*>(sb + OFF_B0 + g * 32 + 16);
        float v[8] = {acc1[0][0] + bs0.x, acc1[0][1] + bs0.y, acc1[0][2] + bs0.z, acc1[0][3] + bs0.w,
                      acc1[1][0] + bs1.x, acc1[1][1] + bs1.y, acc1[1][2] + bs1.z, acc1[1][3] + bs1.w};
#pragma unroll
        for (int s = 0; s < NSRC; ++s) {
            const char* t = sb + lo00[s];
            // 4 taps x 8 channels: v_perm_b32 pairs the same channel of two taps, v_dot2c_f32_bf16 applies both weights
            const uint4 t00 = *reinterpret_cast<const uint4*>(t);
            const uint4 t01 = *reinterpret_cast<const uint4*>(t + ldx[s]);
            const uint4 t10 = *reinterpret_cast<const uint4*>(t + ldy[s]);
            const uint4 t11 = *reinterpret_cast<const uint4*>(t + ldy[s] + ldx[s]);
            const unsigned a0[4] = {t00.x, t00.y, t00.z, t00.w}, a1[4] = {t01.x, t01.y, t01.z, t01.w};
            const unsigned b0[4] = {t10.x, t10.y, t10.z, t10.w}, b1[4] = {t11.x, t11.y, t11.z, t11.w};
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
                const bf16x2 tl = __builtin_bit_cast(bf16x2, __builtin_amdgcn_perm(a1[pr], a0[pr], 0x05040100u));
                const bf16x2 th = __builtin_bit_cast(bf16x2, __builtin_amdgcn_perm(a1[pr], a0[pr], 0x07060302u));
                const bf16x2 bl = __builtin_bit_cast(bf16x2, __builtin_amdgcn_perm(b1[pr], b0[pr], 0x05040100u));
                const bf16x2 bh = __builtin_bit_cast(bf16x2, __builtin_amdgcn_perm(b1[pr], b0[pr], 0x07060302u));
                v[2 * pr] = __builtin_amdgcn_fdot2_f32_bf16(tl, wtop[s], v[2 * pr], false);
                v[2 * pr] = __builtin_amdgcn_fdot2_f32_bf16(bl, wbot[s], v[2 * pr], false);
                v[2 * pr + 1] = __builtin_amdgcn_fdot2_f32_bf16(th, wtop[s], v[2 * pr + 1], false);
                v[2 * pr + 1] = __builtin_amdgcn_fdot2_f32_bf16(bh, wbot[s], v[2 * pr + 1], false);
            }
        }
        bf16x8 bH;
#pragma unroll
        for (int e = 0; e < 8; ++e) bH[e] = (__bf16)fmaxf(v[e], 0.f);
        // ---- stage 2: logits += W1[:, q-slice] . h -------------------------------------------------------
#pragma unroll
        for (int mi = 0; mi < M2; ++mi) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(sb + OFF_W1 + (mi * 64 + lane) * 16);
            acc2[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bH, acc2[mi], 0, 0, 0);
        }
    }
    // ---- logits (+ conv bias) -> fp32 NHWC [P][LC]; lane holds classes mi*16 + g*4 .. +3 of its pixel -------
    if (valid) {
#pragma unroll
        for (int mi = 0; mi < M2; ++mi) {
            const int c = mi * 16 + g * 4;
            const float4 b = *reinterpret_cast<const float4*>(p.bias1 + c);
            *reinterpret_cast<float4*>(p.logits + pix * p.LC + c) =
                make_float4(acc2[mi][0] + b.x, acc2[mi][1] + b.y, acc2[mi][2] + b.z, acc2[mi][3] + b.w);
        }
    }
}

template <int M2>
void launch_nsrc(const HeadParams& q, unsigned blocks, hipStream_t s) {
    switch (q.nsrc) {
        case 3: hipLaunchKernelGGL((head_fused_kernel<M2, 3>), dim3(blocks), dim3(256), (size_t)2 * (q.nsrc * HEAD_SRC_LDS + (4 + M2 + 1) * 1024), s, q); break;
        case 4: hipLaunchKernelGGL((head_fused_kernel<M2, 4>), dim3(blocks), dim3(256), (size_t)2 * (q.nsrc * HEAD_SRC_LDS + (4 + M2 + 1) * 1024), s, q); break;
        default: hipLaunchKernelGGL((head_fused_kernel<M2, 5>), dim3(blocks), dim3(256), (size_t)2 * (q.nsrc * HEAD_SRC_LDS + (4 + M2 + 1) * 1024), s, q); break;
    }
}

int launch_head_fused(const HeadParams& p, int m2, hipStream_t s) {
    HeadParams q = p;
    q.tiles_x = (p.W + 15) / 16;
    q.tiles_y = (p.H + HEAD_TH - 1) / HEAD_TH;
    if (p.nsrc < 3 || p.nsrc > HEAD_MAX_SRC) { set_error("fused head: %d gather sources", p.nsrc); return SNCAL_ERR_ARG; }
    int total_instr = 0;
    for (int s2 = 0; s2 < p.nsrc; ++s2) {     // worst-case source box of a tile must fit its LDS slot / DMA list
        const int bh = (int)(p.sy[s2] * (HEAD_TH - 1)) + 3, bwid = (int)(p.sx[s2] * 15) + 3;
        if (bh * bwid * 64 > HEAD_SRC_LDS) { set_error("fused head: source %d is not a down-scaled branch", s2); return SNCAL_ERR_ARG; }
        total_instr += (bh * bwid * 4 + 63) / 64;
    }
    if (total_instr > 4 * HEAD_MAX_DMA) { set_error("fused head: source boxes need %d DMA instructions", total_instr); return SNCAL_ERR_ARG; }
    const unsigned blocks = (unsigned)(q.tiles_x * q.tiles_y * p.N);
    if (m2 == 2) launch_nsrc<2>(q, blocks, s);
    else if (m2 == 4) launch_nsrc<4>(q, blocks, s);
    else { set_error("fused head supports up to 64 classes"); return SNCAL_ERR_ARG; }
    SNCAL_CHECK_LAUNCH();
    return SNCAL_OK;
}

}  // namespace sncal
