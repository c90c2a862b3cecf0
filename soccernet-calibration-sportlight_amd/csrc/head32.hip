// Fused HRNet head, 32 x 32 x 16 MFMA version (bf16 path; keypoint network: two gathered wide branches, K1 a multiple of 16).
//
// Same mathematics and the same restructuring as head.hip (read its header first):
//     stage 1  MFMA   h = W0_d . [direct | up(narrow branches)]                      (K1 = 64 + 48 + 96 = 208)
//     gather   MFMA   h += sum_s t_s . wint_s      t_s = box pixels of the wide branch's product at native resolution,
//                                                   wint_s = this output pixel's bilinear weights over the box (B fragment)
//              VALU   h = relu(h)                  (the folded-BN shift is stage 1's initial value)
//     stage 2  MFMA   logits += W1[:, 32-slice] . h
// for /root/reference/src/models/hrnet/hrnet.py:489-510, :316-329.  What changes is the tiling: a wave owns 32 pixels of one
// output row and every product is a v_mfma_f32_32x32x16_bf16 -- 32 hidden channels x 32 pixels per instruction.  Against the
// 16 x 16 x 32 version per 32 pixels and 32-channel slice: 19 MFMAs of 32 clk instead of 40 of ~19.4, 17 A-fragment reads from
// LDS instead of 36 (the fused head was LDS-bound after its gather moved to the matrix pipe), and a workgroup's slice of
// weights (17 KB of LDS-DMA) serves 128 pixels instead of 64.
// The hand-off between the two GEMMs is still a register repack: MFMA row r of a 32-row block carries channel
// h32_row_channel(r) (head.hpp), so registers 8 h .. 8 h + 7 of a lane's accumulator are the channels 16 h + 8 (lane >> 5) + 0..7
// of its pixel -- exactly the 8 k-values that lane must supply to stage 2's K = 16 step h.
#include "common.hpp"
#include "head.hpp"
#include "softmax_px.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>

#pragma clang fp contract(fast)

namespace sncal {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int H32_SRC = 8 * 1024;          // per slice buffer: 4 waves x 2 sources x one 1 KB DMA piece (16 box pixels x 64 B)

template <int RB, int KS, int DB, int HL, int DEC>
__global__ __launch_bounds__(256, DB ? 3 : 4) void head32_kernel(const HeadParams p) {
    constexpr int OFF_W0 = H32_SRC, OFF_W1 = OFF_W0 + KS * 1024, OFF_B0 = OFF_W1 + RB * 2 * 1024, BUF = OFF_B0 + 1024;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long t_begin = p.trace ? __builtin_amdgcn_s_memtime() : 0ull;
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

    // ---- this wave's source boxes (its row, its 32 columns) and its two DMA pieces per slice -------------------------------------
    unsigned dma_voff[2];
    bf16x8 wint[2];
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
        {
            const int pi = lane >> 2, piece = lane & 3;
            const int ly = pi >= bw ? 1 : 0, lx = pi - ly * bw;
            dma_voff[s] = pi < npx ? (unsigned)((((by0 + ly) * p.Ws[s] + bx0 + lx) * p.HP) * 2 + piece * 16) : 0x80000000u;
        }
        // bilinear taps of this lane's pixel inside the box (PyTorch align_corners=True index) -> B fragment of the interpolation GEMM:
        // lane (pixel l31, k-block hi) holds the weights of box pixels 8 hi .. 8 hi + 7; the same bf16 weights as head.hip
        const float fx = p.sx[s] * (float)xc;
        int ix = (int)fx;
        ix = ix > p.Ws[s] - 1 ? p.Ws[s] - 1 : ix;
        const float ly1 = fy - (float)by0, lx1 = fx - (float)ix;
        const float w00 = (float)(__bf16)((1.f - lx1) * (1.f - ly1)), w01 = (float)(__bf16)(lx1 * (1.f - ly1));
        const float w10 = (float)(__bf16)((1.f - lx1) * ly1), w11 = (float)(__bf16)(lx1 * ly1);
        const int t00 = ix - bx0, t01 = t00 + (ix < p.Ws[s] - 1 ? 1 : 0), t10 = t00 + (nrows == 2 ? bw : 0), t11 = t10 + (ix < p.Ws[s] - 1 ? 1 : 0);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int slot = 8 * hi + e;
            float w = 0.f;
            w += slot == t00 ? w00 : 0.f;
            w += slot == t01 ? w01 : 0.f;
            w += slot == t10 ? w10 : 0.f;
            w += slot == t11 ? w11 : 0.f;
            wint[s][e] = (__bf16)w;
        }
    }

    // everything the slice loop consumes comes through LDS-DMA (an ordinary global load inside the loop would make hipcc wait
    // vmcnt(0) at its first use and drain the prefetch every iteration)
    const __amdgpu_buffer_rsrc_t rs_w0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w0_32), 0, p.NQ * KS * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w1_32), 0, p.NQ * RB * 2 * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias0), 0, p.HP * 4, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_src[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const size_t img = (size_t)p.Hs[s] * p.Ws[s] * p.HP * 2;      // one image of source s (ranges stay < 2 GB)
        rs_src[s] = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.src[s])) + (size_t)n * img, 0, (int)img, 0x00020000);
    }
    auto issue_slice = [&](int q, int buf) {
        char* const base = smem + buf * BUF;
#pragma unroll
        for (int s = 0; s < 2; ++s)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_src[s], (lds_void*)(base + (wave * 2 + s) * 1024), 16, dma_voff[s], (unsigned)(q * 64), 0, 0);
#pragma unroll
        for (int i = 0; i < (KS + 3) / 4; ++i)
            if (wave + 4 * i < KS)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w0, (lds_void*)(base + OFF_W0 + (wave + 4 * i) * 1024), 16, (unsigned)(lane * 16),
                                                         (unsigned)((q * KS + wave + 4 * i) * 1024), 0, 0);
        if (wave < RB * 2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w1, (lds_void*)(base + OFF_W1 + wave * 1024), 16, (unsigned)(lane * 16),
                                                     (unsigned)((q * RB * 2 + wave) * 1024), 0, 0);
        if (wave == 3)      // 32 shift values = 128 B; the other lanes read out of range
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_b0, (lds_void*)(base + OFF_B0), 16,
                                                     lane < 8 ? (unsigned)(lane * 16) : 0x80000000u, (unsigned)(q * 128), 0, 0);
    };
    issue_slice(0, 0);

    // ---- stage-1 B fragments: K = [direct channels | upsampled narrow branches]; lane (pixel l31, k-block hi) holds channels
    // 16 ks + 8 hi .. + 7.  Segment boundaries are multiples of 8 channels, so a lane's k-group lies in exactly one segment.
    bf16x8 bD[KS];
    const __bf16* direct = reinterpret_cast<const __bf16*>(p.direct);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int kk = ks * 16 + hi * 8;
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (kk < p.Cd) {
            v = *reinterpret_cast<const bf16x8*>(direct + pix * p.Cd + kk);
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
        bD[ks] = v;
    }

    f32x16 acc2[RB];
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[rb][e] = 0.f;
    const int tch = h32_row_channel(l31);         // hidden channel (within a slice) of this lane's row of a transposed box fragment

    // tuning aid: clocks of wave 0 in [0] wait + barrier, [1] next slice requested, [2] stage 1, [3] gather, [4] ReLU + stage 2, [5] prologue
    unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0}, tprev = 0;
    const bool tracing = p.trace != nullptr;
    auto lap = [&](int k) { if (tracing) { const unsigned long long now = __builtin_amdgcn_s_memtime(); tsum[k] += now - tprev; tprev = now; } };
    if (tracing) { tprev = t_begin; lap(5); }
    for (int q = 0; q < p.NQ; ++q) {
        const int buf = DB ? (q & 1) : 0;
        if (!DB && q > 0) {
            asm volatile("s_barrier" ::: "memory");
            issue_slice(q, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my DMA pieces of slice q landed
        asm volatile("s_barrier" ::: "memory");              // everyone's did; everyone is done with slice q-1
        lap(0);
        if (DB && q + 1 < p.NQ) issue_slice(q + 1, buf ^ 1);       // lands while slice q is consumed
        lap(1);
        const char* const sb = smem + buf * BUF;
        // ---- stage 1: 32 hidden channels x 32 pixels; accumulator registers 8 h .. 8 h + 7 = channels 16 h + 8 hi + 0..7, started at
        // the folded-BN shift
        f32x16 acc1;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 b0 = *reinterpret_cast<const float4*>(sb + OFF_B0 + (16 * h + 8 * hi) * 4);
            const float4 b1 = *reinterpret_cast<const float4*>(sb + OFF_B0 + (16 * h + 8 * hi + 4) * 4);
            acc1[8 * h + 0] = b0.x; acc1[8 * h + 1] = b0.y; acc1[8 * h + 2] = b0.z; acc1[8 * h + 3] = b0.w;
            acc1[8 * h + 4] = b1.x; acc1[8 * h + 5] = b1.y; acc1[8 * h + 6] = b1.z; acc1[8 * h + 7] = b1.w;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(sb + OFF_W0 + (ks * 64 + lane) * 16);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bD[ks], acc1, 0, 0, 0);
        }
        lap(2);
        // ---- gather: one more MFMA per wide branch.  A fragment = the box pixels of this slice, transposed on the fly: lane (row l31 ->
        // channel h32_row_channel(l31), k-block hi) reads box pixels 8 hi .. 8 hi + 7 for its channel (eight 2-byte reads, 64-byte stride)
        typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const unsigned short* tp = reinterpret_cast<const unsigned short*>(sb + (wave * 2 + s) * 1024 + (8 * hi) * 64 + tch * 2);
            u16x8 t;
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = tp[e * 32];
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, t), wint[s], acc1, 0, 0, 0);
        }
        lap(3);
        // ---- ReLU -> stage-2 B fragments (a register repack), stage 2: logits += W1[:, q-slice] . h -----------------------------------
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8 bH, bL;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float hv = fmaxf(acc1[8 * h + e], 0.f);
                bH[e] = (__bf16)hv;
                if constexpr (HL) bL[e] = (__bf16)(hv - (float)bH[e]);      // experiment (SNCAL_HEAD_HILO=1): hidden vector as bf16 hi + lo
            }
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(sb + OFF_W1 + ((rb * 2 + h) * 64 + lane) * 16);
                acc2[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bH, acc2[rb], 0, 0, 0);
                if constexpr (HL) acc2[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bL, acc2[rb], 0, 0, 0);
            }
        }
        lap(4);
    }
    if (tracing && threadIdx.x == 0 && blockIdx.x % 97 == 0)
        for (int k = 0; k < 6; ++k) p.trace[(size_t)(blockIdx.x / 97) * 8 + k] = tsum[k];
    if constexpr (DEC) {
        // ---- decode-fused epilogue: log-softmax per pixel (softmax_px.hpp: bit-identical to the softmax kernels), then the tile's maxima
        // per class -- over its 32 columns for every row, over its 4 rows for every column -- which is all the keypoint decode needs
        // (transforms.py:230-238: argmax of the column maxima / row maxima).  The (N,58,h,w) log-probabilities and the logits are never
        // written: 2 x 2.1 GB per 64 frames less HBM traffic and one kernel less.
        static_assert(!DEC || RB == 2, "the two-lane softmax holds 64 channel slots per pixel");
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
        // [row of the tile][class][pixel] in LDS (the slice buffers are free: everyone is past the last slice)
        asm volatile("s_barrier" ::: "memory");
        float* const s_lp = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) s_lp[(wave * 64 + 16 * k + 8 * hi + e) * 32 + l31] = valid ? r[8 * k + e] : -INFINITY;
        __syncthreads();
        const int C1 = p.dec_C - 1, t = threadIdx.x;
        {   // row maxima: thread -> (row t >> 6, class t & 63)
            const int rw = t >> 6, c = t & 63, yy = oy0 + rw;
            if (c < C1 && yy < p.H) {
                const float4* q = reinterpret_cast<const float4*>(s_lp + (rw * 64 + c) * 32);
                float m = -INFINITY;
#pragma unroll
                for (int i = 0; i < 8; ++i) { const float4 u = q[i]; m = fmaxf(m, fmaxf(fmaxf(u.x, u.y), fmaxf(u.z, u.w))); }
                p.dec_row[(((size_t)n * C1 + c) * p.H + yy) * p.tiles_x + tx] = m;
            }
        }
        for (int id = t; id < C1 * 32; id += 256) {      // column maxima: (class id >> 5, column id & 31)
            const int c = id >> 5, xx = id & 31;
            if (ox0 + xx < p.W) {
                const float m = fmaxf(fmaxf(s_lp[(0 * 64 + c) * 32 + xx], s_lp[(1 * 64 + c) * 32 + xx]), fmaxf(s_lp[(2 * 64 + c) * 32 + xx], s_lp[(3 * 64 + c) * 32 + xx]));
                p.dec_col[(((size_t)n * p.tiles_y + ty) * C1 + c) * p.W + ox0 + xx] = m;
            }
        }
        return;
    }
    // ---- logits (+ conv bias) -> fp32 NHWC [P][LC]; registers 8 h .. 8 h + 7 of block rb = classes 32 rb + 16 h + 8 hi + 0..7 ----------
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

// applies when: two gather sources whose per-wave boxes (one output row x 32 columns) hold at most 16 pixels, K1 = ks16 * 16 with an
// instantiated depth, LC a multiple of 8 and at most 64.  Returns false (nothing launched) otherwise: the 16 x 16 x 32 kernel runs.
static bool finish_trace(const HeadParams& q, size_t n_tr, const char* trace_file, hipStream_t s) {
    if (q.trace) {
        std::vector<unsigned long long> h(n_tr);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h.data(), q.trace, n_tr * 8, hipMemcpyDeviceToHost);
        (void)hipFree(q.trace);
        if (FILE* f = fopen(trace_file, "wb")) { fwrite(h.data(), 8, n_tr, f); fclose(f); }
    }
    return true;
}

bool head32_applies(const HeadParams& p) {
    static const int enabled = getenv("SNCAL_HEAD32") ? atoi(getenv("SNCAL_HEAD32")) : 1;      // tuning aid: 0 = head.hip
    if (!enabled || p.nsrc != 2 || !p.w0_32 || !p.w1_32 || p.ks16 != 13 || p.LC > 64 || p.LC % 8) return false;
    for (int s2 = 0; s2 < 2; ++s2) {
        const int bwid = (int)(p.sx[s2] * 31) + 3;                 // worst-case box: 2 rows x bwid columns
        if (2 * bwid > 16) return false;
    }
    return true;
}
void head32_decode_parts(int h, int w, int* row_parts, int* col_parts) { *row_parts = (w + 31) / 32; *col_parts = (h + 3) / 4; }
size_t head32_decode_scratch(int B, int C, int h, int w) {
    int rp, cp;
    head32_decode_parts(h, w, &rp, &cp);
    return ((size_t)B * (C - 1) * h * rp + (size_t)B * cp * (C - 1) * w) * sizeof(float);
}

bool launch_head32(const HeadParams& p, hipStream_t s) {
    if (!head32_applies(p)) return false;
    HeadParams q = p;
    q.tiles_x = (p.W + 31) / 32;
    q.tiles_y = (p.H + 3) / 4;
    q.tiles_x_magic = q.tiles_x <= 1 ? 0u : 0xFFFFFFFFu / (unsigned)q.tiles_x + 1u;
    q.tiles_y_magic = q.tiles_y <= 1 ? 0u : 0xFFFFFFFFu / (unsigned)q.tiles_y + 1u;
    const unsigned blocks = (unsigned)(q.tiles_x * q.tiles_y * p.N);
    const int rb = (p.LC + 31) / 32;
    static const char* trace_file = getenv("SNCAL_HEAD_TRACE");
    const size_t n_tr = (size_t)(blocks / 97 + 1) * 8;
    q.trace = nullptr;
    if (trace_file && hipMalloc(&q.trace, n_tr * 8) == hipSuccess) (void)hipMemsetAsync(q.trace, 0, n_tr * 8, s);
    // single-buffered slices by default: 26 KB of LDS and 128 VGPRs let FOUR workgroups share a CU, and a workgroup's prologue (boxes,
    // interpolation weights, 13 B fragments: 22 % of its life) and slice waits hide under the others -- 3.8 ms against 4.2 ms double-buffered at three
    static const int db = getenv("SNCAL_HEAD_DB") ? atoi(getenv("SNCAL_HEAD_DB")) : 0;
    const size_t lds1 = (size_t)(H32_SRC + (13 + rb * 2 + 1) * 1024);
    if (p.dec_row && p.dec_col && rb == 2) {       // decode-fused form: 32 KB of LDS for the tile's log-probabilities
        SNCAL_LAUNCH((head32_kernel<2, 13, 0, 0, 1>), dim3(blocks), dim3(256), (size_t)32 * 1024, s, q);
        return finish_trace(q, n_tr, trace_file, s);
    }
    // SNCAL_HEAD_HILO=1 (experiment): stage 2 multiplies the hidden vector as bf16 hi + bf16 lo (16 mantissa
    // bits instead of 8) -- what "hidden -> logits in higher precision" buys is measured with tests/test_parity_gpu.py, NOTES/design_history_r1_r5.md §8
    static const int hilo = getenv("SNCAL_HEAD_HILO") ? atoi(getenv("SNCAL_HEAD_HILO")) : 0;
    if (rb == 2) {
        if (hilo) SNCAL_LAUNCH((head32_kernel<2, 13, 1, 1, 0>), dim3(blocks), dim3(256), 2 * lds1, s, q);
        else if (db) SNCAL_LAUNCH((head32_kernel<2, 13, 1, 0, 0>), dim3(blocks), dim3(256), 2 * lds1, s, q);
        else SNCAL_LAUNCH((head32_kernel<2, 13, 0, 0, 0>), dim3(blocks), dim3(256), lds1, s, q);
    } else {
        if (db) SNCAL_LAUNCH((head32_kernel<1, 13, 1, 0, 0>), dim3(blocks), dim3(256), 2 * lds1, s, q);
        else SNCAL_LAUNCH((head32_kernel<1, 13, 0, 0, 0>), dim3(blocks), dim3(256), lds1, s, q);
    }
    return finish_trace(q, n_tr, trace_file, s);
}

}  // namespace sncal
