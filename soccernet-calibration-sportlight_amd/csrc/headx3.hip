// Fused HRNet head of the bf16x3 engine: head32.hip's structure in SPLIT-bf16 arithmetic on fp32 tensors (keypoint network: stem
// features direct, two narrow branches folded into stage 1's K, two wide branches gathered; /root/reference/src/models/hrnet/hrnet.py
// :489-510, :316-329).
//
//     stage 1  MFMA x3  h = W0_d . [direct | up(narrow branches)]                    (K1 = 64 + 48 + 96 = 208 = 13 K-steps of 16)
//     gather   MFMA x3  h += sum_s t_s . wint_s     t_s = box pixels of the wide branch's fp32 product at native resolution
//              VALU     h = relu(h)                 (the folded-BN shift is stage 1's initial value), fp32
//     stage 2  MFMA x3  logits += W1[:, 32-slice] . h
// Every product a.b of two fp32 operands is a_hi.b_hi + a_hi.b_lo + a_lo.b_hi on v_mfma_f32_32x32x16_bf16 with hi = bf16(x),
// lo = bf16(x - hi) (2^-17 per operand), accumulated in fp32: the arithmetic of conv_tt's MODE 2 (NOTES/design_history_r1_r5.md §9.3).  The weights arrive
// split from the host (hi and lo A fragments side by side in the slice), activations are split in registers: the direct tensor and the
// bilinear blends of the folded branches once per tile (26 B fragments), the gathered box pixels and the hidden vector per slice.
// Why: in the fp32-class engines the head was three passes over fp32 tensors of 784 channels x 270 x 480 x 64 frames = 26 GB each
// (per-source products 19 ms, bilinear sum 33 ms, last conv ~8 ms per 64 frames); fused, the hidden vector never leaves registers.
// Tiling, row order (h32_row_channel), the register repack between the GEMMs and the decode-fused epilogue (per-pixel log-softmax, row
// and column maxima of the tile) are head32.hip's.  Slice pipeline (round 4): the kernel's 232 VGPRs allow two workgroups per CU, so a
// workgroup takes 78 KB of LDS: stage-1 weights double-buffered and requested a slice ahead, stage-2 weights requested at the top of their
// slice (they land under stage 1 + gather), a wave's gather boxes requested as soon as the wave has read the current ones.
#include "common.hpp"
#include "head.hpp"
#include <vector>
#include <cstdio>
#include "softmax_px.hpp"
#include "x3.hpp"
#include <cstdio>
#include <cstdlib>

#pragma clang fp contract(off)

namespace sncal {

typedef x3h8 bf16x8;            // (x3.hpp: the 16-bit type of the split, fp16 or bf16; the name is historical)
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void;

namespace {
constexpr int KS = 13, RB = 2;
constexpr int X_SRC = 16 * 1024;                    // 4 waves x 2 sources x 2 KB (16 box pixels x 32 channels fp32)
// Stage-1 weights and shift are DOUBLE-buffered (round 4): the kernel holds 232 VGPRs (104 of them the resident stage-1 B fragments), so
// two workgroups share a CU and 2 x 78 KB of LDS fit; slice q + 1's 27 KB land while slice q multiplies.
constexpr int W0_BUF = 2 * KS * 1024 + 1024;        // [W0 hi 13 KB][W0 lo 13 KB][shift 1 KB]
constexpr int OFF_W0 = X_SRC, OFF_W1H = OFF_W0 + 2 * W0_BUF, OFF_W1L = OFF_W1H + RB * 2 * 1024, X_LDS = OFF_W1L + RB * 2 * 1024;      // 79872 B
static_assert(2 * X_LDS <= 160 * 1024, "two workgroups per CU");

typedef int i32x4_t __attribute__((ext_vector_type(4)));
// One LDS-DMA piece as inline assembly (64 lanes x 16 bytes -> 1 KB at LDS byte address `lds_addr`): hipcc's wait-count pass would put
// vmcnt(0) in front of every fragment read behind a DMA builtin, i.e. wait for the NEXT slice's pieces; every wait here is explicit.
__device__ __forceinline__ void dma_piece(i32x4_t rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ i32x4_t raw_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    return i32x4_t{(int)(unsigned)a, (int)(unsigned)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}

__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& h, bf16x8& l) {
    x3u4 hu, lu;
    x3_split8(v, x3_lower(false), hu, lu);
    h = __builtin_bit_cast(bf16x8, hu); l = __builtin_bit_cast(bf16x8, lu);
}
__device__ __forceinline__ f32x16 mfma3(const bf16x8& ah, const bf16x8& al, const bf16x8& bh, const bf16x8& bl, f32x16 acc) {
    acc = X3_MFMA_32x32x16(al, bh, acc);
    acc = X3_MFMA_32x32x16(ah, bl, acc);
    return X3_MFMA_32x32x16(ah, bh, acc);
}
}  // namespace

template <int DEC>
__global__ __launch_bounds__(256, 2) void headx3_kernel(const HeadParams p) {
    const unsigned long long t_begin = p.trace ? __builtin_amdgcn_s_memtime() : 0ull;
    float amax = 0.f;                                  // range tracker of the hidden vector (x3.hpp; the inputs were tracked by their producers)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int tile = blockIdx.x;
    const unsigned q1 = p.tiles_x == 1 ? (unsigned)tile : __umulhi((unsigned)tile, p.tiles_x_magic);
    const int tx = tile - (int)q1 * p.tiles_x;
    const unsigned q2 = p.tiles_y == 1 ? q1 : __umulhi(q1, p.tiles_y_magic);
    const int ty = (int)q1 - (int)q2 * p.tiles_y;
    const int n = (int)q2;
    const int oy0 = ty * 4, ox0 = tx * 32;
    const int y = oy0 + wave, yc = min(y, p.H - 1);
    const int x = ox0 + l31, xc = min(x, p.W - 1);
    const bool valid = y < p.H && x < p.W;
    const long pix = ((long)n * p.H + yc) * p.W + xc;

    // ---- this wave's source boxes (its row, its 32 columns): two DMA pieces of 64 x 16 B per source and slice (a box pixel's slice is
    // 32 fp32 = 8 lanes) and the bilinear weights of this lane's pixel over the box as a split B fragment
    unsigned dma_voff[2][2];
    bf16x8 wih[2], wil[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int xlast = min(ox0 + 31, p.W - 1);
        const float fy = p.sy[s] * (float)yc;
        int by0 = (int)fy;
        by0 = by0 > p.Hs[s] - 1 ? p.Hs[s] - 1 : by0;
        const int nrows = by0 < p.Hs[s] - 1 ? 2 : 1;
        const int bx0 = (int)(p.sx[s] * (float)ox0);
        const int bx1 = min((int)(p.sx[s] * (float)xlast) + 1, p.Ws[s] - 1);
        const int bw = bx1 - bx0 + 1, npx = nrows * bw;            // <= 16: checked on the host for the worst case
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int slot = j * 64 + lane, pi = slot >> 3, part = slot & 7;
            const int ly = pi >= bw ? 1 : 0, lx = pi - ly * bw;
            dma_voff[s][j] = pi < npx ? (unsigned)((((by0 + ly) * p.Ws[s] + bx0 + lx) * p.HP) * 4 + part * 16) : 0x80000000u;
        }
        const float fx = p.sx[s] * (float)xc;
        int ix = (int)fx;
        ix = ix > p.Ws[s] - 1 ? p.Ws[s] - 1 : ix;
        const float ly1 = fy - (float)by0, lx1 = fx - (float)ix;
        const float w00 = (1.f - lx1) * (1.f - ly1), w01 = lx1 * (1.f - ly1), w10 = (1.f - lx1) * ly1, w11 = lx1 * ly1;
        const int t00 = ix - bx0, t01 = t00 + (ix < p.Ws[s] - 1 ? 1 : 0), t10 = t00 + (nrows == 2 ? bw : 0), t11 = t10 + (ix < p.Ws[s] - 1 ? 1 : 0);
        float wv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int slot = 8 * hi + e;
            float w = 0.f;
            w += slot == t00 ? w00 : 0.f;
            w += slot == t01 ? w01 : 0.f;
            w += slot == t10 ? w10 : 0.f;
            w += slot == t11 ? w11 : 0.f;
            wv[e] = w;
        }
        split8(wv, wih[s], wil[s]);
    }

    const i32x4_t rs_w0h = raw_rsrc(p.w0_32, (unsigned)(p.NQ * KS * 1024)), rs_w0l = raw_rsrc(p.w0_32_lo, (unsigned)(p.NQ * KS * 1024));
    const i32x4_t rs_w1h = raw_rsrc(p.w1_32, (unsigned)(p.NQ * RB * 2 * 1024)), rs_w1l = raw_rsrc(p.w1_32_lo, (unsigned)(p.NQ * RB * 2 * 1024));
    const i32x4_t rs_b0 = raw_rsrc(p.bias0, (unsigned)(p.HP * 4));
    i32x4_t rs_src[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const size_t img = (size_t)p.Hs[s] * p.Ws[s] * p.HP * 4;      // one fp32 image of source s
        rs_src[s] = raw_rsrc(reinterpret_cast<const char*>(p.src[s]) + (size_t)n * img, (unsigned)img);
    }
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(lds_void*)smem;
    // the three request groups of a slice.  Boxes: this wave's own region (private: requested as soon as the wave has read the old ones);
    // stage-1 weights + shift: buffer q & 1, requested one slice ahead; stage-2 weights: single-buffered, requested at the top of their slice
    auto issue_boxes = [&](int q) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 2; ++j) dma_piece(rs_src[s], lds0 + ((wave * 2 + s) * 2 + j) * 1024, dma_voff[s][j], (unsigned)(q * 128));
    };
    auto issue_w0 = [&](int q) {
        const unsigned base = lds0 + OFF_W0 + (q & 1) * W0_BUF;
#pragma unroll
        for (int i = 0; i < (KS + 3) / 4; ++i)
            if (wave + 4 * i < KS) {
                dma_piece(rs_w0h, base + (wave + 4 * i) * 1024, (unsigned)(lane * 16), (unsigned)((q * KS + wave + 4 * i) * 1024));
                dma_piece(rs_w0l, base + KS * 1024 + (wave + 4 * i) * 1024, (unsigned)(lane * 16), (unsigned)((q * KS + wave + 4 * i) * 1024));
            }
        if (wave == 3) dma_piece(rs_b0, base + 2 * KS * 1024, lane < 8 ? (unsigned)(lane * 16) : 0x80000000u, (unsigned)(q * 128));
    };
    auto issue_w1 = [&](int q) {       // RB * 2 = 4 pieces each: one per wave
        dma_piece(rs_w1h, lds0 + OFF_W1H + wave * 1024, (unsigned)(lane * 16), (unsigned)((q * RB * 2 + wave) * 1024));
        dma_piece(rs_w1l, lds0 + OFF_W1L + wave * 1024, (unsigned)(lane * 16), (unsigned)((q * RB * 2 + wave) * 1024));
    };
    unsigned long long tpro[2] = {0, 0}, tpro_prev = t_begin;      // tuning aid: prologue split (slot 6: set-up + first requests, slot 7: direct fragments + box wait)
    auto lap_pro = [&](int k) { if (p.trace) { const unsigned long long now = __builtin_amdgcn_s_memtime(); tpro[k] += now - tpro_prev; tpro_prev = now; } };
    issue_boxes(0);
    issue_w0(0);

    // ---- stage-1 B fragments, split: K = [direct channels | upsampled narrow branches]; lane (pixel l31, k-block hi) holds channels
    // 16 ks + 8 hi .. + 7 of its pixel.  Segment boundaries are multiples of 8 channels.
    // (A branch-free form of this loop -- every k-step a four-tap blend, the loads of 4 + 4 + 2 + 2 + 1 k-steps in flight together --
    // was measured in round 4: the prologue stayed at ~30k clocks of a workgroup's ~166k.  It is bound by the number of cache lines the
    // taps touch (64 B per pixel, tap and k-step), not by 13 serial trips to HBM.)
    // Round 5: the folded branches' taps come out of LDS.  A tile of 4 x 32 pixels blends 72 float4 per lane out of a 4 x 18 (x2 branch) and a
    // 3 x 10 (x4 branch) pixel box of the sources; fetched per lane those are 9 KB of requests per pixel row and k-step through the vector
    // memory path, four lanes to a cache line at best -- the prologue was 30k of a workgroup's 166k clocks and bound by exactly that count
    // (round 4).  Now the two boxes (<= 26 KB) are requested once per workgroup as LDS-DMA pieces into stage-1 weight buffer 1, which is idle
    // until the top of slice 0, while the direct tensor's fragments are fetched, and the blends read LDS.
    bf16x8 bDh[KS], bDl[KS];
    const float* direct = reinterpret_cast<const float*>(p.direct);
    int f_iy0[HEAD_MAX_FOLD], f_ix0[HEAD_MAX_FOLD], f_bw[HEAD_MAX_FOLD], f_off[HEAD_MAX_FOLD];
    if (p.stage_folds) {
        int off = 0;
        const int y0c = min(oy0, p.H - 1), y3c = min(oy0 + 3, p.H - 1), x31c = min(ox0 + 31, p.W - 1);
#pragma unroll
        for (int f = 0; f < HEAD_MAX_FOLD; ++f) {
            f_iy0[f] = f_ix0[f] = f_bw[f] = f_off[f] = 0;
            if (f < p.nfold) {
                const int iy0 = min((int)(p.fsy[f] * (float)y0c), p.Hf[f] - 1), iy3 = min((int)(p.fsy[f] * (float)y3c), p.Hf[f] - 1);
                const int ix0 = min((int)(p.fsx[f] * (float)ox0), p.Wf[f] - 1), ix1 = min((int)(p.fsx[f] * (float)x31c), p.Wf[f] - 1);
                const int nrows = iy3 - iy0 + 1 + (iy3 < p.Hf[f] - 1 ? 1 : 0), bw = ix1 - ix0 + 1 + (ix1 < p.Wf[f] - 1 ? 1 : 0);
                const unsigned rowb = (unsigned)(bw * p.Cf[f] * 4), total = (unsigned)nrows * rowb;       // (a box row is contiguous in the source)
                const int npieces = (int)((total + 1023u) / 1024u);
                const size_t img = (size_t)p.Hf[f] * p.Wf[f] * p.Cf[f] * 4;
                const i32x4_t rs_f = raw_rsrc(reinterpret_cast<const char*>(p.fold[f]) + (size_t)n * img, (unsigned)img);
                const unsigned lds_f = (unsigned)(__UINTPTR_TYPE__)(lds_void*)smem + OFF_W0 + W0_BUF + (unsigned)off;
                for (int i = wave; i < npieces; i += 4) {
                    const unsigned o = (unsigned)((i * 64 + lane) * 16);
                    const unsigned row = (o >= rowb ? 1u : 0u) + (o >= 2u * rowb ? 1u : 0u) + (o >= 3u * rowb ? 1u : 0u);      // (<= 4 rows: launcher)
                    const unsigned voff = o < total ? (unsigned)(((iy0 + (int)row) * p.Wf[f] + ix0) * p.Cf[f] * 4) + (o - row * rowb) : 0x80000000u;
                    dma_piece(rs_f, lds_f + (unsigned)(i * 1024), voff, 0u);
                }
                f_iy0[f] = iy0; f_ix0[f] = ix0; f_bw[f] = bw; f_off[f] = off;
                off += npieces * 1024;
            }
        }
    }
    // torch's bilinear order: rows blended in x first, then in y (upsample_add's fp32 path, ops.hip)
    auto blend8 = [&](const auto* t, int dx, int dy, float lx1, float ly1, float (&v)[8]) __attribute__((always_inline)) {
        const float lx0 = 1.f - lx1, ly0 = 1.f - ly1;
#pragma unroll
        for (int h4 = 0; h4 < 2; ++h4) {
            const float4 t00 = *reinterpret_cast<const float4*>(t + 4 * h4), t01 = *reinterpret_cast<const float4*>(t + dx + 4 * h4);
            const float4 t10 = *reinterpret_cast<const float4*>(t + dy + 4 * h4), t11 = *reinterpret_cast<const float4*>(t + dy + dx + 4 * h4);
            v[4 * h4 + 0] = (t00.x * lx0 + t01.x * lx1) * ly0 + (t10.x * lx0 + t11.x * lx1) * ly1;
            v[4 * h4 + 1] = (t00.y * lx0 + t01.y * lx1) * ly0 + (t10.y * lx0 + t11.y * lx1) * ly1;
            v[4 * h4 + 2] = (t00.z * lx0 + t01.z * lx1) * ly0 + (t10.z * lx0 + t11.z * lx1) * ly1;
            v[4 * h4 + 3] = (t00.w * lx0 + t01.w * lx1) * ly0 + (t10.w * lx0 + t11.w * lx1) * ly1;
        }
    };
    // per fold and lane, once: LDS float index of the lane's first tap, tap distances, blend weights (the staged path)
    int f_t0[HEAD_MAX_FOLD], f_dx[HEAD_MAX_FOLD], f_dy[HEAD_MAX_FOLD];
    float f_lx[HEAD_MAX_FOLD], f_ly[HEAD_MAX_FOLD];
#pragma unroll
    for (int f = 0; f < HEAD_MAX_FOLD; ++f) {
        f_t0[f] = f_dx[f] = f_dy[f] = 0; f_lx[f] = f_ly[f] = 0.f;
        if (p.stage_folds && f < p.nfold) {
            const float fy = p.fsy[f] * (float)yc, fx = p.fsx[f] * (float)xc;     // align_corners=True
            int iy = (int)fy, ix = (int)fx;
            iy = iy > p.Hf[f] - 1 ? p.Hf[f] - 1 : iy;
            ix = ix > p.Wf[f] - 1 ? p.Wf[f] - 1 : ix;
            f_ly[f] = fy - (float)iy; f_lx[f] = fx - (float)ix;
            f_dx[f] = ix < p.Wf[f] - 1 ? p.Cf[f] : 0;
            f_dy[f] = iy < p.Hf[f] - 1 ? f_bw[f] * p.Cf[f] : 0;
            f_t0[f] = (OFF_W0 + W0_BUF + f_off[f]) / 4 + ((iy - f_iy0[f]) * f_bw[f] + (ix - f_ix0[f])) * p.Cf[f];
        }
    }
    bool staged_ready = false;
    lap_pro(0);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int kk = ks * 16 + hi * 8;
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (p.stage_folds && !staged_ready && ks * 16 >= p.Cd) {      // (Cd is a multiple of 16 whenever the boxes are staged: launcher)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_barrier" ::: "memory");                   // everyone's pieces of the boxes
            staged_ready = true;
            lap_pro(1);
        }
        if (kk < p.Cd) {
            const float4 a = *reinterpret_cast<const float4*>(direct + pix * p.Cd + kk), b = *reinterpret_cast<const float4*>(direct + pix * p.Cd + kk + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
            int seg0 = p.Cd;
#pragma unroll
            for (int f = 0; f < HEAD_MAX_FOLD; ++f) {
                if (f < p.nfold) {
                    if (kk >= seg0 && kk < seg0 + p.Cf[f]) {
                        if (p.stage_folds) {
                            blend8(reinterpret_cast<const float*>(smem) + f_t0[f] + (kk - seg0), f_dx[f], f_dy[f], f_lx[f], f_ly[f], v);
                        } else {
                            const float fy = p.fsy[f] * (float)yc, fx = p.fsx[f] * (float)xc;     // align_corners=True
                            int iy = (int)fy, ix = (int)fx;
                            iy = iy > p.Hf[f] - 1 ? p.Hf[f] - 1 : iy;
                            ix = ix > p.Wf[f] - 1 ? p.Wf[f] - 1 : ix;
                            const float ly1 = fy - (float)iy, lx1 = fx - (float)ix;
                            const int dx = ix < p.Wf[f] - 1 ? p.Cf[f] : 0, dy = iy < p.Hf[f] - 1 ? p.Wf[f] * p.Cf[f] : 0;
                            const float* t = reinterpret_cast<const float*>(p.fold[f]) + (((size_t)n * p.Hf[f] + iy) * p.Wf[f] + ix) * p.Cf[f] + (kk - seg0);
                            blend8(t, dx, dy, lx1, ly1, v);
                        }
                    }
                    seg0 += p.Cf[f];
                }
            }
        }
        split8(v, bDh[ks], bDl[ks]);
    }

    f32x16 acc2[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[rb][e] = 0.f;
    const int tch = h32_row_channel(l31);         // hidden channel (within a slice) of this lane's row of a transposed box fragment

    // tuning aid (SNCAL_HEAD_TRACE=<file>, tools/head_trace.py): clocks of wave 0 in [0] barriers + wait, [1] requests, [2] stage 1,
    // [3] gather, [4] ReLU + stage 2, [5] prologue
    unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
    const bool tracing = p.trace != nullptr;
    auto lap = [&](int k) { if (tracing) { const unsigned long long now = __builtin_amdgcn_s_memtime(); tsum[k] += now - tprev; tprev = now; } };
    if (tracing) { tprev = t_begin; lap(5); }
    // Request order per wave: boxes(0), W0(0) | then per slice q: W1(q), W0(q + 1), boxes(q + 1).  Requests complete in order, so
    // "at most n newest in flight" (vmcnt(n)) names exactly which older groups have landed; a wave issues 6-9 pieces per W0 group
    // (the waits use the smallest count, i.e. wait for up to 3 pieces more than needed), 2 per W1 group, 4 per box group.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");
    for (int q = 0; q < p.NQ; ++q) {
        // here: everyone is done with slice q - 1 and everyone's pieces of W0(q) have landed (the barrier at the end of slice q - 1)
        lap(0);
        issue_w1(q);                                  // lands under stage 1 + gather
        if (q + 1 < p.NQ) issue_w0(q + 1);            // lands under the whole slice
        lap(1);
        const char* const w0b = smem + OFF_W0 + (q & 1) * W0_BUF;
        // ---- stage 1: 32 hidden channels x 32 pixels, started at the folded-BN shift
        f32x16 acc1;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 b0 = *reinterpret_cast<const float4*>(w0b + 2 * KS * 1024 + (16 * h + 8 * hi) * 4);
            const float4 b1 = *reinterpret_cast<const float4*>(w0b + 2 * KS * 1024 + (16 * h + 8 * hi + 4) * 4);
            acc1[8 * h + 0] = b0.x; acc1[8 * h + 1] = b0.y; acc1[8 * h + 2] = b0.z; acc1[8 * h + 3] = b0.w;
            acc1[8 * h + 4] = b1.x; acc1[8 * h + 5] = b1.y; acc1[8 * h + 6] = b1.z; acc1[8 * h + 7] = b1.w;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 ah = *reinterpret_cast<const bf16x8*>(w0b + (ks * 64 + lane) * 16);
            const bf16x8 al = *reinterpret_cast<const bf16x8*>(w0b + KS * 1024 + (ks * 64 + lane) * 16);
            acc1 = mfma3(ah, al, bDh[ks], bDl[ks], acc1);
        }
        lap(2);
        // my boxes of slice q (requested after the gather of slice q - 1; newer: W1(q), W0(q + 1)) -- the first slice's landed in the prologue
        if (q > 0) { if (q + 1 < p.NQ) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
        // ---- gather: A fragment = the box pixels of this slice, transposed on the fly and split: lane (row l31 -> channel tch, k-block
        // hi) reads box pixels 8 hi .. 8 hi + 7 of its channel (eight 4-byte reads, 128-byte stride)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float* tp = reinterpret_cast<const float*>(smem + (wave * 2 + s) * 2048) + (8 * hi) * 32 + tch;
            float tv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) tv[e] = tp[e * 32];
            bf16x8 th, tl;
            split8(tv, th, tl);
            acc1 = mfma3(th, tl, wih[s], wil[s], acc1);
        }
        if (q + 1 < p.NQ) {             // my boxes are read (tv went through the split): the next slice's travel under stage 2 and stage 1
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue_boxes(q + 1);
            asm volatile("s_waitcnt vmcnt(10)" ::: "memory");     // my pieces of W1(q) (newer: W0(q + 1) and the boxes)
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");                   // everyone's pieces of W1(q)
        lap(3);
        // ---- ReLU -> split stage-2 B fragments (a register repack), stage 2: logits += W1[:, q-slice] . h
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float hv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) hv[e] = acc1[8 * h + e];
            x3u4 bhu, blu;
            x3_split8(hv, x3_lower(true), bhu, blu, amax);           // ReLU + clamp + split
            const bf16x8 bh = __builtin_bit_cast(bf16x8, bhu), bl = __builtin_bit_cast(bf16x8, blu);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const bf16x8 ah = *reinterpret_cast<const bf16x8*>(smem + OFF_W1H + ((rb * 2 + h) * 64 + lane) * 16);
                const bf16x8 al = *reinterpret_cast<const bf16x8*>(smem + OFF_W1L + ((rb * 2 + h) * 64 + lane) * 16);
                acc2[rb] = mfma3(ah, al, bh, bl, acc2[rb]);
            }
        }
        lap(4);
        if (q + 1 < p.NQ) {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // my pieces of W0(q + 1) (newer: the boxes)
            asm volatile("s_barrier" ::: "memory");               // everyone's; and everyone is done with slice q
        }
    }
    x3_report(amax, p.range);
    if (tracing && threadIdx.x == 0 && blockIdx.x % 97 == 0)
    {
        for (int k = 0; k < 6; ++k) p.trace[(size_t)(blockIdx.x / 97) * 8 + k] = tsum[k];
        p.trace[(size_t)(blockIdx.x / 97) * 8 + 6] = tpro[0];
        p.trace[(size_t)(blockIdx.x / 97) * 8 + 7] = tpro[1];
    }
    if constexpr (DEC) {
        // decode-fused epilogue (head32.hip's): per-pixel log-softmax (softmax_px.hpp: the same arithmetic and summation order as the
        // softmax kernels), then the tile's maxima per class over its 32 columns for every row and over its 4 rows for every column
        float v[32], r[32];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = rb * 32 + 16 * h + 8 * hi;
                const float4 b0 = *reinterpret_cast<const float4*>(p.bias1 + c), b1 = *reinterpret_cast<const float4*>(p.bias1 + c + 4);
                const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) v[8 * (2 * rb + h) + e] = c + e < p.dec_C ? acc2[rb][8 * h + e] + bb[e] : -INFINITY;
            }
        logsoftmax_px32x2(v, hi, p.dec_C, r);
        asm volatile("s_barrier" ::: "memory");
        float* const s_lp = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) s_lp[(wave * 64 + 16 * k + 8 * hi + e) * 32 + l31] = valid ? r[8 * k + e] : -INFINITY;
        __syncthreads();
        const int C1 = p.dec_C - 1, t = threadIdx.x;
        {
            const int rw = t >> 6, c = t & 63, yy = oy0 + rw;
            if (c < C1 && yy < p.H) {
                const float4* qq = reinterpret_cast<const float4*>(s_lp + (rw * 64 + c) * 32);
                float m = -INFINITY;
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float4 u = qq[i]; m = fmaxf(m, fmaxf(fmaxf(u.x, u.y), fmaxf(u.z, u.w))); }
                p.dec_row[(((size_t)n * C1 + c) * p.H + yy) * p.tiles_x + tx] = m;
            }
        }
        for (int id = t; id < C1 * 32; id += 256) {
            const int c = id >> 5, xx = id & 31;
            if (ox0 + xx < p.W) {
                const float m = fmaxf(fmaxf(s_lp[(0 * 64 + c) * 32 + xx], s_lp[(1 * 64 + c) * 32 + xx]), fmaxf(s_lp[(2 * 64 + c) * 32 + xx], s_lp[(3 * 64 + c) * 32 + xx]));
                p.dec_col[(((size_t)n * p.tiles_y + ty) * C1 + c) * p.W + ox0 + xx] = m;
            }
        }
        return;
    }
    if (valid) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = rb * 32 + 16 * h + 8 * hi;
                if (c < p.LC) {
                    const float4 b0 = *reinterpret_cast<const float4*>(p.bias1 + c), b1 = *reinterpret_cast<const float4*>(p.bias1 + c + 4);
                    float* o = p.logits + pix * p.LC + c;
                    *reinterpret_cast<float4*>(o) = make_float4(acc2[rb][8 * h] + b0.x, acc2[rb][8 * h + 1] + b0.y, acc2[rb][8 * h + 2] + b0.z, acc2[rb][8 * h + 3] + b0.w);
                    *reinterpret_cast<float4*>(o + 4) = make_float4(acc2[rb][8 * h + 4] + b1.x, acc2[rb][8 * h + 5] + b1.y, acc2[rb][8 * h + 6] + b1.z, acc2[rb][8 * h + 7] + b1.w);
                }
            }
    }
}

// applies to the keypoint network's head: two gather sources whose per-wave boxes hold at most 16 pixels, K1 = 13 x 16, 33..64 classes
bool headx3_applies(const HeadParams& p) {
    static const int enabled = getenv("SNCAL_HEADX3") ? atoi(getenv("SNCAL_HEADX3")) : 1;      // 0 = the split head on the generic fp32 kernels
    if (!enabled || p.nsrc != 2 || !p.w0_32 || !p.w0_32_lo || !p.w1_32 || !p.w1_32_lo || p.ks16 != KS || p.LC != 64 || p.Cd % 8) return false;
    for (int s2 = 0; s2 < 2; ++s2) {
        const int bwid = (int)(p.sx[s2] * 31) + 3;
        if (2 * bwid > 16) return false;
    }
    return true;
}

bool launch_headx3(const HeadParams& p, hipStream_t s) {
    if (!headx3_applies(p)) return false;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&headx3_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, X_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&headx3_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, X_LDS);
        attr_done = true;
    }
    HeadParams q = p;
    // the folded branches' boxes through LDS (kernel prologue): at most four source rows per tile, both boxes within one stage-1 weight buffer
    static const int stage = getenv("SNCAL_HEAD_STAGE") ? atoi(getenv("SNCAL_HEAD_STAGE")) : 1;
    q.stage_folds = stage != 0 && p.Cd % 16 == 0 ? 1 : 0;
    {
        int bytes = 0;
        for (int f = 0; f < p.nfold; ++f) {
            const int rows = (int)(p.fsy[f] * 3.f) + 3, cols = (int)(p.fsx[f] * 31.f) + 3;
            if (rows > 4 || p.Cf[f] % 8) q.stage_folds = 0;
            bytes += (rows * cols * p.Cf[f] * 4 + 1023) / 1024 * 1024;
        }
        if (bytes > W0_BUF) q.stage_folds = 0;
    }
    q.tiles_x = (p.W + 31) / 32;
    q.tiles_y = (p.H + 3) / 4;
    q.tiles_x_magic = q.tiles_x <= 1 ? 0u : 0xFFFFFFFFu / (unsigned)q.tiles_x + 1u;
    q.tiles_y_magic = q.tiles_y <= 1 ? 0u : 0xFFFFFFFFu / (unsigned)q.tiles_y + 1u;
    const unsigned blocks = (unsigned)(q.tiles_x * q.tiles_y * p.N);
    static const char* trace_file = getenv("SNCAL_HEAD_TRACE");
    const size_t n_tr = (size_t)(blocks / 97 + 1) * 8;
    q.trace = nullptr;
    if (trace_file && hipMalloc(&q.trace, n_tr * 8) == hipSuccess) (void)hipMemsetAsync(q.trace, 0, n_tr * 8, s);
    if (p.dec_row && p.dec_col) SNCAL_LAUNCH((headx3_kernel<1>), dim3(blocks), dim3(256), (size_t)X_LDS, s, q);
    else SNCAL_LAUNCH((headx3_kernel<0>), dim3(blocks), dim3(256), (size_t)X_LDS, s, q);
    if (q.trace) {
        std::vector<unsigned long long> h(n_tr);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h.data(), q.trace, n_tr * 8, hipMemcpyDeviceToHost);
        (void)hipFree(q.trace);
        if (FILE* f = fopen(trace_file, "wb")) { fwrite(h.data(), 8, n_tr, f); fclose(f); }
    }
    return true;
}

}  // namespace sncal
