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
//     stage 1  MFMA   h = W0_d . [direct | up(narrow branches)]
//                     The narrow branches (48 + 96 channels) are cheaper to upsample BEFORE the 1x1 conv: their
//                     bilinear taps are blended once per pixel into extra stage-1 B fragments (K = 64 + 48 + 96),
//                     so no 784-channel product of theirs is ever written (t_0 alone was 3.3 GB per 64 frames)
//                     or gathered per slice.
//     gather   VALU   h += sum_i bilinear(t_i)     (the two wide branches: 4 taps each; the 32-channel slice of
//                                                   every source box of the 4x16 tile is DMA'd to LDS, double-buffered)
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
// tile = 4*NP rows x 16 columns of head pixels; a wave owns NP rows (the A fragments of a slice -- ~19 KB of
// LDS-DMA per slice -- are shared by all of them: at NP = 1 their re-streaming per 64 pixels bound the kernel)
constexpr int HEAD_SRC_LDS = 3072;         // LDS bytes per source per q-slice (<= 48 source pixels x 64 B)
constexpr int HEAD_MAX_DMA = 4;            // DMA instructions per wave per slice (<= 16 over the block)

template <int M2, int NSRC, int KS1, int NP, int DB, int GM>
__global__ __launch_bounds__(256, DB ? (NP == 1 ? 3 : 2) : (NP == 1 ? 5 : 3)) void head_fused_kernel(const HeadParams p) {
    constexpr int HEAD_TH = 4 * NP;
    constexpr int OFF_W0 = NSRC * HEAD_SRC_LDS, OFF_W1 = OFF_W0 + 2 * KS1 * 1024, OFF_B0 = OFF_W1 + M2 * 1024;
    constexpr int HEAD_BUF = OFF_B0 + 1024;     // per q-slice: source boxes, stage-1 / stage-2 A fragments, BN shift
    extern __shared__ __attribute__((aligned(16))) char smem[];      // 2 * HEAD_BUF bytes
    const int lane = threadIdx.x & 63, g = lane >> 4, ln = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tile = blockIdx.x;
    // multiply-high by host reciprocals instead of emulated integer divisions (~25 VALU instructions each)
    const unsigned q1 = p.tiles_x == 1 ? (unsigned)tile : __umulhi((unsigned)tile, p.tiles_x_magic);
    const int tx = tile - (int)q1 * p.tiles_x;
    const unsigned q2 = p.tiles_y == 1 ? q1 : __umulhi(q1, p.tiles_y_magic);
    const int ty = (int)q1 - (int)q2 * p.tiles_y;
    const int n = (int)q2;
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
                const int ly = (pi * ((65536 / bw[s]) + 1)) >> 16, lx = pi - ly * bw[s];     // pi < 64, bw <= 64: exact; bw[s] is
                                                                                             // block-uniform, its reciprocal is scalar work
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
        // A fragments of the slice: stage-1 pieces (2 * KS1) round-robin over the waves, stage-2 piece w (M2 pieces)
#pragma unroll
        for (int i = 0; i < (2 * KS1 + 3) / 4; ++i)
            if (wave + 4 * i < 2 * KS1)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w0, (lds_void*)(base + OFF_W0 + (wave + 4 * i) * 1024), 16, (unsigned)(lane * 16),
                                                         (unsigned)((q * 2 * KS1 + wave + 4 * i) * 1024), 0, 0);
        if (wave < M2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w1, (lds_void*)(base + OFF_W1 + wave * 1024), 16, (unsigned)(lane * 16),
                                                     (unsigned)((q * M2 + wave) * 1024), 0, 0);
        if (wave == 3)      // 32 shift values = 128 B; the other lanes read out of range
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b0, (lds_void*)(base + OFF_B0), 16,
                                                     lane < 8 ? (unsigned)(lane * 16) : 0x80000000u, (unsigned)(q * 128), 0, 0);
    };
    issue_slice(0, 0);

    // per-lane pixel bookkeeping: wave owns rows wave*NP .. wave*NP+NP-1 of the tile; lane column = ln
    const int x = ox0 + ln, xc = min(x, p.W - 1);
    bool valid[NP];
    long pix[NP];
    unsigned lo00[NP][NSRC], ldx[NSRC], ldy[NP][NSRC];     // LDS byte offsets of the taps
    bf16x2 wtop[NP][NSRC], wbot[NP][NSRC];     // (w00, w01) and (w10, w11) as bf16 pairs for v_dot2c_f32_bf16
    bf16x8 bD[NP][KS1];
    bf16x8 wint[NP];
#pragma unroll
    for (int r = 0; r < NP; ++r) {
        const int y = oy0 + wave * NP + r;
        valid[r] = y < p.H && x < p.W;
        const int yc = min(y, p.H - 1);
        pix[r] = ((long)n * p.H + yc) * p.W + xc;
#pragma unroll
        for (int s = 0; s < NSRC; ++s) {
            const float fy = p.sy[s] * (float)yc, fx = p.sx[s] * (float)xc;   // PyTorch align_corners=True index
            int iy = (int)fy, ix = (int)fx;
            iy = iy > p.Hs[s] - 1 ? p.Hs[s] - 1 : iy;
            ix = ix > p.Ws[s] - 1 ? p.Ws[s] - 1 : ix;
            const float ly1 = fy - (float)iy, lx1 = fx - (float)ix;
            wtop[r][s][0] = (__bf16)((1.f - lx1) * (1.f - ly1)); wtop[r][s][1] = (__bf16)(lx1 * (1.f - ly1));
            wbot[r][s][0] = (__bf16)((1.f - lx1) * ly1); wbot[r][s][1] = (__bf16)(lx1 * ly1);
            lo00[r][s] = (unsigned)(s * HEAD_SRC_LDS + ((iy - by0[s]) * bw[s] + (ix - bx0[s])) * 64 + g * 16);
            ldx[s] = ix < p.Ws[s] - 1 ? 64u : 0u;
            ldy[r][s] = iy < p.Hs[s] - 1 ? (unsigned)(bw[s] * 64) : 0u;
        }
        if constexpr (GM) {
            // gather-by-MFMA (two sources, boxes of <= 16 pixels): B fragment of the interpolation GEMM
            //     h[ch, px] += sum_k t[ch, k] * wint[k, px],   k = 16 s + (pixel of source s's box),
            // lane (px = ln, k-block g) holds the bf16 bilinear weights of box pixels 8 (g & 1) .. + 7 of source g >> 1 (zero
            // where the pixel is not one of this output pixel's four taps).  Same bf16 weights as the VALU path's dot2 pairs.
            float wq[4] = {0.f, 0.f, 0.f, 0.f};
            int tq[4] = {-1, -1, -1, -1};
#pragma unroll
            for (int s2 = 0; s2 < NSRC; ++s2) {
                const bool mine = (g >> 1) == s2;
                const int t00 = (int)((lo00[r][s2] - (unsigned)(s2 * HEAD_SRC_LDS) - (unsigned)(g * 16)) >> 6);
                const int t01 = t00 + (int)(ldx[s2] >> 6), t10 = t00 + (int)(ldy[r][s2] >> 6), t11 = t10 + (int)(ldx[s2] >> 6);
                const float f00 = (float)wtop[r][s2][0], f01 = (float)wtop[r][s2][1], f10 = (float)wbot[r][s2][0], f11 = (float)wbot[r][s2][1];
                tq[0] = mine ? t00 : tq[0]; tq[1] = mine ? t01 : tq[1]; tq[2] = mine ? t10 : tq[2]; tq[3] = mine ? t11 : tq[3];
                wq[0] = mine ? f00 : wq[0]; wq[1] = mine ? f01 : wq[1]; wq[2] = mine ? f10 : wq[2]; wq[3] = mine ? f11 : wq[3];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int slot = 8 * (g & 1) + e;
                float w = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) w += slot == tq[k] ? wq[k] : 0.f;
                wint[r][e] = (__bf16)w;
            }
        }
        // stage-1 B fragments: K = [direct channels | upsampled narrow branches], 8 channels per lane and k-step.
        // Segment boundaries are multiples of 8 channels, so a lane's k-group lies in exactly one segment.
#pragma unroll
        for (int ks = 0; ks < KS1; ++ks) {
            const int kk = ks * 32 + g * 8;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (kk < p.Cd) {
                v = *reinterpret_cast<const bf16x8*>(direct + pix[r] * p.Cd + kk);
            } else {
                int seg0 = p.Cd;
#pragma unroll
                for (int f = 0; f < HEAD_MAX_FOLD; ++f) {
                    if (f < p.nfold) {
                        if (kk >= seg0 && kk < seg0 + p.Cf[f]) {
                            const float fy = p.fsy[f] * (float)yc, fx = p.fsx[f] * (float)xc;     // align_corners=True
                            int iy = (int)fy, ix = (int)fx;
                            iy = iy > p.Hf[f] - 1 ? p.Hf[f] - 1 : iy;
                            ix = ix > p.Wf[f] - 1 ? p.Wf[f] - 1 : ix;
                            const float ly1 = fy - (float)iy, lx1 = fx - (float)ix;
                            const int dx = ix < p.Wf[f] - 1 ? p.Cf[f] : 0, dy = iy < p.Hf[f] - 1 ? p.Wf[f] * p.Cf[f] : 0;
                            const __bf16* t = reinterpret_cast<const __bf16*>(p.fold[f]) +
                                              (((size_t)n * p.Hf[f] + iy) * p.Wf[f] + ix) * p.Cf[f] + (kk - seg0);
                            const bf16x8 t00 = *reinterpret_cast<const bf16x8*>(t), t01 = *reinterpret_cast<const bf16x8*>(t + dx);
                            const bf16x8 t10 = *reinterpret_cast<const bf16x8*>(t + dy), t11 = *reinterpret_cast<const bf16x8*>(t + dy + dx);
                            const float w00 = (1.f - lx1) * (1.f - ly1), w01 = lx1 * (1.f - ly1), w10 = (1.f - lx1) * ly1, w11 = lx1 * ly1;
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                v[e] = (__bf16)(w00 * (float)t00[e] + w01 * (float)t01[e] + w10 * (float)t10[e] + w11 * (float)t11[e]);
                        }
                        seg0 += p.Cf[f];
                    }
                }
            }
            bD[r][ks] = v;
        }
    }

    f32x4 acc2[NP][M2];
#pragma unroll
    for (int r = 0; r < NP; ++r)
#pragma unroll
        for (int mi = 0; mi < M2; ++mi) acc2[r][mi] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int q = 0; q < p.NQ; ++q) {
        const int buf = DB ? (q & 1) : 0;
        if (!DB && q > 0) {                                  // single buffer: more workgroups per CU hide the round instead
            asm volatile("s_barrier" ::: "memory");
            issue_slice(q, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my DMA pieces of slice q landed
        asm volatile("s_barrier" ::: "memory");              // everyone's did; everyone is done with slice q-1
        if (DB && q + 1 < p.NQ) issue_slice(q + 1, buf ^ 1);       // lands while slice q is consumed
        const char* const sb = smem + buf * HEAD_BUF;
        const float4 bs0 = *reinterpret_cast<const float4*>(sb + OFF_B0 + g * 32);
        const float4 bs1 = *reinterpret_cast<const float4*>(sb + OFF_B0 + g * 32 + 16);
#pragma unroll
        for (int r = 0; r < NP; ++r) {
            // ---- stage 1: 32 hidden channels x 16 pixels, K = direct + folded channels ---------------------
            f32x4 acc1[2] = {f32x4{bs0.x, bs0.y, bs0.z, bs0.w}, f32x4{bs1.x, bs1.y, bs1.z, bs1.w}};   // folded BN shift
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int ks = 0; ks < KS1; ++ks) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(sb + OFF_W0 + ((f * KS1 + ks) * 64 + lane) * 16);
                    acc1[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bD[r][ks], acc1[f], 0, 0, 0);
                }
            if constexpr (GM) {
                // ---- gather as two more MFMAs: A fragment = the sources' box pixels of this slice, transposed on the fly -- lane
                // (row m = ln -> hidden channel (m >> 2) * 8 + f * 4 + (m & 3), the stage-1 row order; k-block g) reads the 8 box
                // pixels 8 (g & 1) .. + 7 of source g >> 1 for its channel: eight 2-byte LDS reads at a 64-byte stride.  64 VALU
                // instructions (v_perm / v_dot2c) and eight ds_read_b128 per row and slice become 16 ds_read_u16 and 2 MFMAs.
                typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const unsigned short* tp = reinterpret_cast<const unsigned short*>(
                        sb + (g >> 1) * HEAD_SRC_LDS + (8 * (g & 1)) * 64 + (((ln >> 2) * 8 + f * 4 + (ln & 3)) * 2));
                    u16x8 t;
#pragma unroll
                    for (int e = 0; e < 8; ++e) t[e] = tp[e * 32];
                    acc1[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, t), wint[r], acc1[f], 0, 0, 0);
                }
            }
            // ---- gather (from LDS) + ReLU: lane owns channels q*32 + g*8 .. +7 of its pixel ------------------
            float v[8] = {acc1[0][0], acc1[0][1], acc1[0][2], acc1[0][3], acc1[1][0], acc1[1][1], acc1[1][2], acc1[1][3]};
            if constexpr (!GM) {
#pragma unroll
            for (int s = 0; s < NSRC; ++s) {
                const char* t = sb + lo00[r][s];
                // 4 taps x 8 channels: v_perm_b32 pairs the same channel of two taps, v_dot2c_f32_bf16 applies both weights
                const uint4 t00 = *reinterpret_cast<const uint4*>(t);
                const uint4 t01 = *reinterpret_cast<const uint4*>(t + ldx[s]);
                const uint4 t10 = *reinterpret_cast<const uint4*>(t + ldy[r][s]);
                const uint4 t11 = *reinterpret_cast<const uint4*>(t + ldy[r][s] + ldx[s]);
                const unsigned a0[4] = {t00.x, t00.y, t00.z, t00.w}, a1[4] = {t01.x, t01.y, t01.z, t01.w};
                const unsigned b0[4] = {t10.x, t10.y, t10.z, t10.w}, b1[4] = {t11.x, t11.y, t11.z, t11.w};
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                    const bf16x2 tl = __builtin_bit_cast(bf16x2, __builtin_amdgcn_perm(a1[pr], a0[pr], 0x05040100u));
                    const bf16x2 th = __builtin_bit_cast(bf16x2, __builtin_amdgcn_perm(a1[pr], a0[pr], 0x07060302u));
                    const bf16x2 bl = __builtin_bit_cast(bf16x2, __builtin_amdgcn_perm(b1[pr], b0[pr], 0x05040100u));
                    const bf16x2 bh = __builtin_bit_cast(bf16x2, __builtin_amdgcn_perm(b1[pr], b0[pr], 0x07060302u));
                    v[2 * pr] = __builtin_amdgcn_fdot2_f32_bf16(tl, wtop[r][s], v[2 * pr], false);
                    v[2 * pr] = __builtin_amdgcn_fdot2_f32_bf16(bl, wbot[r][s], v[2 * pr], false);
                    v[2 * pr + 1] = __builtin_amdgcn_fdot2_f32_bf16(th, wtop[r][s], v[2 * pr + 1], false);
                    v[2 * pr + 1] = __builtin_amdgcn_fdot2_f32_bf16(bh, wbot[r][s], v[2 * pr + 1], false);
                }
            }
            }
            bf16x8 bH;
#pragma unroll
            for (int e = 0; e < 8; ++e) bH[e] = (__bf16)fmaxf(v[e], 0.f);
            // ---- stage 2: logits += W1[:, q-slice] . h ---------------------------------------------------
#pragma unroll
            for (int mi = 0; mi < M2; ++mi) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(sb + OFF_W1 + (mi * 64 + lane) * 16);
                acc2[r][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bH, acc2[r][mi], 0, 0, 0);
            }
        }
    }
    // ---- logits (+ conv bias) -> fp32 NHWC [P][LC]; lane holds classes mi*16 + g*4 .. +3 of its pixel -------
#pragma unroll
    for (int r = 0; r < NP; ++r)
        if (valid[r]) {
#pragma unroll
            for (int mi = 0; mi < M2; ++mi) {
                const int c = mi * 16 + g * 4;
                const float4 b = *reinterpret_cast<const float4*>(p.bias1 + c);
                *reinterpret_cast<float4*>(p.logits + pix[r] * p.LC + c) =
                    make_float4(acc2[r][mi][0] + b.x, acc2[r][mi][1] + b.y, acc2[r][mi][2] + b.z, acc2[r][mi][3] + b.w);
            }
        }
}

template <int M2, int NSRC, int KS1, int NP>
void launch_one(const HeadParams& q, unsigned blocks, hipStream_t s) {
    // single-buffered slices by default: the phases of a slice (A-fragment DMA ~2 ms per 64 frames at the CU's
    // 58 B/clk DMA rate, stage-1 MFMA chains, gather, stage 2) serialise inside a wave, so what pays is MORE
    // resident waves (25 KB of LDS, 92 VGPRs -> 5 per SIMD), not prefetch depth (measured 5.8 vs 6.9 ms)
    static const int db = getenv("SNCAL_HEAD_DB") ? atoi(getenv("SNCAL_HEAD_DB")) : 0;     // tuning aid
    const size_t lds1 = (size_t)(NSRC * HEAD_SRC_LDS + (2 * KS1 + M2 + 1) * 1024);
    // gather by MFMA: two gather sources whose worst-case boxes hold at most 16 pixels each (one DMA piece, K slots 16 s .. 16 s + 15)
    static const int gm_env = getenv("SNCAL_HEAD_GM") ? atoi(getenv("SNCAL_HEAD_GM")) : 1;      // tuning aid: 0 = VALU gather
    bool gm = NSRC == 2 && gm_env != 0;
    for (int s2 = 0; s2 < NSRC && gm; ++s2) {
        const int bh = (int)(q.sy[s2] * (4 * NP - 1)) + 3, bwid = (int)(q.sx[s2] * 15) + 3;
        if (bh * bwid > 16) gm = false;
    }
    if constexpr (NSRC == 2) {
        if (gm) {
            if (db) SNCAL_LAUNCH((head_fused_kernel<M2, NSRC, KS1, NP, 1, 1>), dim3(blocks), dim3(256), 2 * lds1, s, q);
            else SNCAL_LAUNCH((head_fused_kernel<M2, NSRC, KS1, NP, 0, 1>), dim3(blocks), dim3(256), lds1, s, q);
            return;
        }
    }
    if (db) SNCAL_LAUNCH((head_fused_kernel<M2, NSRC, KS1, NP, 1, 0>), dim3(blocks), dim3(256), 2 * lds1, s, q);
    else SNCAL_LAUNCH((head_fused_kernel<M2, NSRC, KS1, NP, 0, 0>), dim3(blocks), dim3(256), lds1, s, q);
}

template <int M2, int NP>
int launch_m2(const HeadParams& q, unsigned blocks, hipStream_t s) {
    const int key = q.nsrc * 10 + q.ks1;
    switch (key) {
        case 22: launch_one<M2, 2, 2, NP>(q, blocks, s); break;
        case 25: launch_one<M2, 2, 5, NP>(q, blocks, s); break;
        case 27: launch_one<M2, 2, 7, NP>(q, blocks, s); break;
        case 32: launch_one<M2, 3, 2, NP>(q, blocks, s); break;
        case 35: launch_one<M2, 3, 5, NP>(q, blocks, s); break;
        case 37: launch_one<M2, 3, 7, NP>(q, blocks, s); break;
        case 42: launch_one<M2, 4, 2, NP>(q, blocks, s); break;
        default: set_error("fused head: %d gather sources with %d stage-1 k-steps is not instantiated", q.nsrc, q.ks1); return SNCAL_ERR_ARG;
    }
    return SNCAL_OK;
}

int launch_head_fused(const HeadParams& p, int m2, hipStream_t s) {
    if (launch_head32(p, s)) { SNCAL_CHECK_LAUNCH(); return SNCAL_OK; }      // the 32 x 32 x 16 version where it applies
    HeadParams q = p;
    if (p.nsrc < 2 || p.nsrc > 4) { set_error("fused head: %d gather sources", p.nsrc); return SNCAL_ERR_ARG; }
    static const int force_np = getenv("SNCAL_HEAD_NP") ? atoi(getenv("SNCAL_HEAD_NP")) : 0;     // tuning aid
    int np = 0;
    static const int np_max = getenv("SNCAL_HEAD_NP_MAX") ? atoi(getenv("SNCAL_HEAD_NP_MAX")) : 1;
    for (int cand = np_max; cand >= 1 && !np; --cand) {   // rows per wave: the worst-case source boxes must fit their LDS slots / DMA list
        if (force_np && cand != force_np) continue;
        const int th = 4 * cand;
        int total_instr = 0; bool ok = true;
        for (int s2 = 0; s2 < p.nsrc; ++s2) {
            const int bh = (int)(p.sy[s2] * (th - 1)) + 3, bwid = (int)(p.sx[s2] * 15) + 3;
            if (bh * bwid * 64 > HEAD_SRC_LDS) ok = false;
            total_instr += (bh * bwid * 4 + 63) / 64;
        }
        if (ok && total_instr <= 4 * HEAD_MAX_DMA) np = cand;
    }
    if (!np) { set_error("fused head: the gather sources are not down-scaled branches"); return SNCAL_ERR_ARG; }
    q.tiles_x = (p.W + 15) / 16;
    q.tiles_y = (p.H + 4 * np - 1) / (4 * np);
    q.tiles_x_magic = q.tiles_x <= 1 ? 0u : 0xFFFFFFFFu / (unsigned)q.tiles_x + 1u;
    q.tiles_y_magic = q.tiles_y <= 1 ? 0u : 0xFFFFFFFFu / (unsigned)q.tiles_y + 1u;
    const unsigned blocks = (unsigned)(q.tiles_x * q.tiles_y * p.N);
    int rc;
    if (m2 == 2) rc = np == 2 ? launch_m2<2, 2>(q, blocks, s) : launch_m2<2, 1>(q, blocks, s);
    else if (m2 == 4) rc = np == 2 ? launch_m2<4, 2>(q, blocks, s) : launch_m2<4, 1>(q, blocks, s);
    else { set_error("fused head supports up to 64 classes"); return SNCAL_ERR_ARG; }
    if (rc) return rc;
    SNCAL_CHECK_LAUNCH();
    return SNCAL_OK;
}

}  // namespace sncal
