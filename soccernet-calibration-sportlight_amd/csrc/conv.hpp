// Implicit-GEMM convolution on gfx950 MFMA: NHWC activations, fragment-ordered packed weights.
//
// Replaces the cuDNN/ATen conv + eval BatchNorm + ReLU (+ residual add) sequences the reference runs for
// BasicBlock / Bottleneck / transition / fuse / head layers (/root/reference/src/models/hrnet/hrnet.py
// :42-58, :79-99, :183-214, :316-329, :357-391, :450-456).  BatchNorm is folded into the packed weights
// (scale) and the epilogue bias (shift) at load time.
//
// GEMM view (operands swapped so that each lane ends up with 4 consecutive output channels of one pixel
// and can store 8/16 contiguous bytes):
//      D[cout, pixel] = sum_k  W[cout, k] * X[k, pixel],   k = (tap, cin)
//   A operand = weights  (M = 16 output channels per MFMA tile)
//   B operand = pixels   (N = 16 consecutive output x of one row per MFMA tile)
// K is walked in 16-byte "k-groups" (8 bf16 / 4 fp32 input channels of one tap); one k-step = 4 k-groups =
// one v_mfma_f32_16x16x32_bf16, or four v_mfma_f32_16x16x4_f32 for the exact-fp32 parity path.  The four
// k-groups of a k-step may belong to different taps, so Cin = 48 (6 k-groups per tap) packs 54 -> 56
// k-groups instead of padding the channel count.
//
// Workgroup = 256 threads = 4 waves.  It owns a spatial tile of 4*NI pixel fragments (TH rows x TWF
// fragments wide, chosen per layer so odd sizes like 135x240, 68x120, 34x60, 17x30 waste < 7 %) times
// MI*16 output channels.  Per input-channel chunk (G k-groups per pixel) it stages
//   - the input halo tile  [(TH-1)*S+KS] x [(16*TWF-1)*S+KS] pixels x G*16 bytes   (read once, reused by
//     all KS*KS taps and all MI channel tiles: this is where the 9x im2col redundancy is absorbed), with
//     a per-(G,stride) pixel pitch that makes the B-fragment ds_read_b128 bank-conflict-free;
//   - the weight chunk, already in fragment order [k-step][mi][lane][16 B] (lane-linear, conflict-free).
// Each wave computes NI pixel fragments x MI channel fragments: per k-step MI + NI ds_read_b128 feed
// MI*NI MFMAs.
//
// Staging is all LDS-DMA (buffer_load_dwordx4 ... lds, 1 KB per wave-instruction, no VGPR round trip): the
// weight chunk is a lane-linear copy; the halo tile is written as [pixel][SLOTS x 16 B] where lanes that
// fall on the pitch padding, outside the image or beyond Cin fetch out of bounds and the buffer descriptor
// returns zeros -- zero padding costs nothing.  Barriers are raw s_barrier + explicit s_waitcnt vmcnt so
// that hipcc does not drain the DMA queue early.
//
// What bounds the kernel (per-workgroup s_memtime traces, profiles/): a staging round is LATENCY-bound -- weights
// are L2 hits (~58 B/clk/CU), the halo comes from HBM/Infinity Cache (2-3k clk under load) -- and a workgroup
// that issues a round and then waits leaves the CU without MFMA work ~50 % of the time even with a second
// resident workgroup.  Explicit prefetch of the next chunk (halo only, or halo + weights, double-buffered in LDS)
// was built and measured: the waits disappear but the DMA issue cost (~100 clk per 1 KB piece on the issuing wave)
// moves into the MFMA phase and the doubled LDS footprint costs a resident workgroup -- never faster than single
// buffers with two or three workgroups per CU, so the kernel keeps single buffers only.  An L2 prefetch of a future
// workgroup's halo (one dword per 128-byte line through LDS-DMA, issued under the last MFMA phase) left the first
// round's wait unchanged: that wait is the prologue's own VALU work and DMA issue, not a cold-miss latency.
// Progressive landing (halo first, weights in k-step order, the MFMAs of k-steps 0..2 starting when the first third of the
// weights has landed: partial s_waitcnt vmcnt(N) + two extra barriers per chunk) was also built: bit-identical, the explicit
// wait fell from 1.4k to 0.2k clk but the ISSUE phase grew by the same amount -- the issuing waves are throttled by the CU's
// LDS-DMA rate (75 KB per chunk at <= 58 B/clk/CU; 54 KB of it weights re-fetched per 192-pixel tile), so a round is bound
// by DMA throughput during its burst rather than by the latency of its last piece; workgroup lifetime unchanged, removed.
#pragma once
#include "x3.hpp"
#include <hip/hip_runtime.h>
#include <cstdint>
#include "common.hpp"

namespace sncal {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

// element tag of the bf16x3 engine's generic convolutions: fp32 tensors staged exactly like the float variant, multiplied in split-bf16
// arithmetic (x = hi + lo bf16, y = hi.hi + hi.lo + lo.hi in fp32 -- conv_tt_body.inc MODE 2 has the numbers).  The packed weights hold
// [4 hi | 4 lo] bf16 in the 16 bytes where the float variant holds 4 fp32; a B fragment (4 fp32 of one k-group) is split in registers.
struct x3_t {};
template <typename T> struct Elem;
template <> struct Elem<__bf16> { static constexpr int GE = 8; using frag = bf16x8; static constexpr bool X3 = false; };
template <> struct Elem<float> { static constexpr int GE = 4; using frag = f32x4; static constexpr bool X3 = false; };
template <> struct Elem<x3_t> { static constexpr int GE = 4; using frag = f32x4; static constexpr bool X3 = true; };

struct ConvParams {
    const void* in;        // [N][Hin][Win][Cin] T
    void* out;             // [N][Hout][Wout][out_cstride] T (or float when out_f32), channels at out_coff
    const void* res;       // optional residual, same indexing as out (may alias out)
    const void* w;         // packed weights: [nblk][chunk][kstep][mi][lane] x 16 B
    const float* bias;     // [cout_frags*16] fp32 (folded BN shift / conv bias)
    int N, Hin, Win, Cin;
    int Hout, Wout, cout_frags;      // cout_frags*16 = padded Cout
    int cout;                        // real Cout (stores are masked beyond it)
    int out_cstride, out_coff;
    int cin_chunks;
    int nblk;                        // output-channel blocks of MI*16 channels
    unsigned n_work, per_xcd;        // work items = tiles * nblk; items per XCD (grid = 8 * per_xcd workgroups)
    unsigned nblk_magic, tiles_x_magic, tiles_y_magic;   // conv_magic() of nblk, tiles_x, tiles_y
    int twf, twf_log2;               // fragments per tile row (a power of two); tile rows TH = 4*NI/twf
    unsigned halo_w_magic;           // floor(2^32 / halo width) + 1: pix / halo_w == umulhi(pix, magic) for pix < 2^16
    int tiles_x, tiles_y;
    int relu, out_f32;
    unsigned w_bytes;                // size of the packed weight buffer (buffer descriptor range)
    int ablate;                      // tuning aid: bit0 skip MFMA phase, bit1 skip DMA, bit2 skip weight DMA, bit3 skip halo DMA (results invalid)
    int epi_lds;                     // 1: transpose the output tile through LDS for 16-byte coalesced stores
    unsigned long long* trace;       // tuning aid (SNCAL_CONV_TRACE): 16 timestamps per workgroup, or null
    unsigned* range;                 // x3_t only: sticky counter of wavefronts that split a value beyond the fp16 range (x3.hpp x3_report), or null
    void* out_twin;                  // x3_t only: split twin of the (dense, out_coff 0) output for the next two-team convolution, or null;
                                     // out may then be null (nobody reads the fp32 form)
};

// pixel pitch (bytes) of the LDS halo tile that makes the B-fragment reads conflict-free
// (found by enumerating the ds_read_b128 lane groups of MI355X_MICROARCH.md "LDS")
__host__ __device__ constexpr int halo_pitch(int G, int stride) {
    int slots = G;
    if (G == 3) return 48;            // 24-channel chunks trade a 2-way read conflict for LDS space (3 workgroups/CU)
    if (stride == 1) { if (G > 1) while (slots % 4 != 2) ++slots; }
    else { if (G > 1 && slots % 2 == 0) ++slots; }
    return slots * 16;
}

// pixel fragments per wave that the coalesced epilogue stages through LDS at a time
__host__ __device__ constexpr int epi_frags(int NI, int wgs_per_cu) { return wgs_per_cu >= 3 ? 1 : NI > 4 ? 2 : NI; }

__host__ __device__ constexpr int conv_nks(int KS, int G) { return (KS * KS * G + 3) / 4; }


// x / d for wave-uniform runtime d without the ~25-instruction emulated integer division: multiply-high by the
// host-computed reciprocal floor(2^32 / d) + 1 (exact while x * d < 2^32); d = 1 has no 32-bit reciprocal
__host__ __device__ inline unsigned conv_magic(unsigned d) { return d <= 1 ? 0u : 0xFFFFFFFFu / d + 1u; }
__device__ __forceinline__ unsigned conv_udiv(unsigned x, unsigned d, unsigned magic) { return d == 1 ? x : __umulhi(x, magic); }

// upper bound of the 1 KB halo pieces of a tile, over the tile shapes the host may pick (TWF = 1, 2, 4, ...)
__host__ __device__ constexpr int conv_max_halo_pieces(int KS, int S, int NI, int G) {
    int best = 0;
    for (int twf = 1; twf <= 4 * NI; twf *= 2) {
        if ((4 * NI) % twf) continue;
        const int th = 4 * NI / twf;
        const int hh = (th - 1) * S + KS, hw = (16 * twf - 1) * S + KS;
        const int pieces = (hh * hw * halo_pitch(G, S) + 1023) / 1024;
        if (pieces <= 64 && pieces > best) best = pieces;      // tiles above 64 KB of halo are never chosen
    }
    return best;
}

// waves per SIMD the kernel is compiled for (HIP's second __launch_bounds__ argument; 2 -> 256 VGPRs, 3 -> 168):
// small weight chunks (<= 32 KB) with <= 18 accumulator tiles fit three workgroups per CU (<= 53 KB LDS each) --
// a staging round is latency-bound (~3k clk), more workgroups in flight hide it.  (A fifth, DMA-only producer
// wave per workgroup was tried: two 5-wave workgroups only co-reside on a CU at <= 128 VGPRs -- wave placement
// starts at the same SIMD -- which this register tile cannot meet.)
__host__ __device__ constexpr int conv_wgs_per_cu(int KS, int NI, int MI, int G) {
    return (conv_nks(KS, G) * MI <= 32 && MI * NI <= 18) ? 3 : 2;
}

// Register budget of a kernel (its __launch_bounds__): the split-arithmetic variants keep more live registers per tile (both halves of the
// B fragments, the paired main terms), and at the 168 VGPRs of three workgroups per CU the 16-fragment tiles spilled 57 registers to
// scratch -- the 1x1 256 -> 64 convolutions of layer1 wrote 1.5 GB per launch where the output is 0.5 GB (PMC WRITE_SIZE).  They get the
// 256-VGPR budget of two workgroups (651 -> 578 us per launch); the LDS staging stays sized by conv_wgs_per_cu (host and device agree
// on that one).  The 12-fragment variants (MI 6 x NI 2: 32 spilled registers) are faster WITH their spills at three workgroups per CU
// (48 against 53 us) and keep the rule.
template <typename T>
__host__ __device__ constexpr int conv_launch_wgs(int KS, int NI, int MI, int G) {
    return (Elem<T>::X3 && MI * NI > 12) ? 2 : conv_wgs_per_cu(KS, NI, MI, G);
}

// workgroups per CU / epilogue staging depth of a variant (host and device agree through these)
__host__ __device__ constexpr int conv_resident_wgs(int KS, int NI, int MI, int G) {
    return conv_wgs_per_cu(KS, NI, MI, G);
}
__host__ __device__ constexpr int conv_epi_frags(int KS, int NI, int MI, int G) {
    return epi_frags(NI, conv_resident_wgs(KS, NI, MI, G));
}

inline size_t conv_stage_bytes(int KS, int S, int NI, int MI, int G, int twf) {
    const int th = 4 * NI / twf;
    const int hh = (th - 1) * S + KS, hw = (16 * twf - 1) * S + KS;
    const size_t halo = ((size_t)hh * hw * halo_pitch(G, S) + 1023) / 1024 * 1024;   // DMA writes whole KBs
    return (size_t)conv_nks(KS, G) * MI * 1024 + halo;
}


typedef __attribute__((address_space(3))) void lds_void;

// One work item (output tile x n-block) of a convolution; the two kernels below map workgroups to work items.
template <typename T, int KS, int STRIDE, int NI, int MI, int G>
__device__ __forceinline__ void conv_body(const ConvParams& p, const unsigned w) {
    using frag = typename Elem<T>::frag;
    constexpr int GE = Elem<T>::GE;
    constexpr int NKG = KS * KS * G, NKS = (NKG + 3) / 4;
    constexpr int PS = halo_pitch(G, STRIDE);
    constexpr int PAD = KS / 2;
    constexpr int SLOTS = PS / 16;
    constexpr int ESIZE = 16 / GE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    [[maybe_unused]] float amax = 0.f;       // x3_t: range tracker of the input and twin splits (x3.hpp)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Work item w = tile * nblk + n-block (n-block fastest: the workgroups that share an input tile and write the other
    // channel slices of the same pixels run together).  XCD-aware order: block b runs on XCD b % 8 (observed dispatch
    // rule, used for speed only), so XCD k walks the contiguous range [k, k+1) * per_xcd of work items -- neighbouring
    // tiles (shared halo rows) and the n-blocks of one tile meet in the same L2 instead of eight different ones.
    unsigned tile = conv_udiv(w, (unsigned)p.nblk, p.nblk_magic);
    const int nb = (int)(w - tile * (unsigned)p.nblk);
    unsigned q = conv_udiv(tile, (unsigned)p.tiles_x, p.tiles_x_magic);
    const int tx = (int)(tile - q * (unsigned)p.tiles_x);
    const unsigned n_u = conv_udiv(q, (unsigned)p.tiles_y, p.tiles_y_magic);
    const int ty = (int)(q - n_u * (unsigned)p.tiles_y);
    const int n = (int)n_u;
    const int TWF = p.twf, TWF_LOG2 = p.twf_log2, TH = (4 * NI) >> TWF_LOG2;     // TWF is a power of two: shifts, not the
                                                                                 // ~25-instruction emulated integer division
    const int HALO_W = (16 * TWF - 1) * STRIDE + KS, HALO_H = (TH - 1) * STRIDE + KS;
    const int oy00 = ty * TH, ox0 = tx * 16 * TWF;
    const int ix0 = ox0 * STRIDE - PAD;
    const int g = lane >> 4, ln = lane & 15;
    const int npix = HALO_H * HALO_W;
    const int halo_bytes = (npix * PS + 1023) / 1024 * 1024;

    // accumulators start at the folded-BN shift of the lane's 4 output channels (no bias pass in the epilogue; the
    // load latency hides under the first staging round)
    f32x4 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const float4 bs = *reinterpret_cast<const float4*>(p.bias + (nb * MI + mi) * 16 + g * 4);   // padded to nblk*MI*16, zeros beyond Cout
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[mi][j] = f32x4{bs.x, bs.y, bs.z, bs.w};
    }

    int boff[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int f = wave * NI + j;
        const int fr = f >> TWF_LOG2, fx = f & (TWF - 1);
        boff[j] = ((fr * STRIDE) * HALO_W + (fx * 16 + ln) * STRIDE) * PS;
    }
    const int row_pitch = HALO_W * PS;

    // buffer descriptors: one image of the input (so that ranges stay < 2 GB), the packed weights
    const size_t img_bytes = (size_t)p.Hin * p.Win * p.Cin * ESIZE;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.in)) + (size_t)n * img_bytes, 0, (int)img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);
    const int n_halo_instr = halo_bytes / 1024;

    // LDS: [weight chunk][halo tile]
    constexpr int W_BYTES = NKS * MI * 1024;
    char* const s_halo0 = smem + W_BYTES;
    auto issue_weights = [&](int c, int buf) {   // lane-linear 1 KB pieces, round-robin over the issuing waves
        if (p.ablate & 6) return;
        const unsigned wbase = (unsigned)(((size_t)nb * p.cin_chunks + c) * W_BYTES);
        char* const sw = smem + buf * W_BYTES;
        for (int i = wave; i < NKS * MI; i += 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(sw + i * 1024), 16, (unsigned)(lane * 16), wbase + i * 1024, 0, 0);
    };
    // Halo pieces: piece j covers 64 consecutive 16-byte slots of the [pixel][SLOTS] image.  A lane's global
    // offset does not depend on the chunk (the chunk's channel offset rides in the scalar offset), so the
    // address arithmetic (two divisions, bounds tests) is done ONCE per workgroup and kept in registers: while a
    // co-resident wave runs MFMAs a VALU instruction issues every ~8 clk, and recomputing ~60 of them per piece
    // per chunk was the longest part of a staging round (per-workgroup trace, profiles/).  Padding slots and
    // outside-image pixels get an out-of-range offset -> the DMA writes zeros.
    // (Kept for register tiles that leave room: <= 8 pieces per wave and <= 18 accumulator tiles, or <= 12 and <= 16.)
    constexpr int MAXH_ALL = (conv_max_halo_pieces(KS, STRIDE, NI, G) + 3) / 4;
    constexpr bool HOIST = (MAXH_ALL <= 8 && MI * NI <= 18) || (MAXH_ALL <= 12 && MI * NI <= 16);
    constexpr int MAXH = HOIST ? MAXH_ALL : 1;
    const int cin_groups = (p.Cin + GE - 1) / GE;
    auto halo_voff = [&](int j, int c_lo) -> unsigned {      // c_lo: first k-group of the chunk, or 0 when hoisted
        // branch-free, divisions by multiplication (pix < 2^16): every VALU instruction here competes with a
        // co-resident wave's MFMA issue
        const unsigned slot = (unsigned)(j * 64 + lane);
        const unsigned pix = slot / (unsigned)SLOTS, cg = slot - pix * SLOTS;
        const unsigned hy = __umulhi(pix, p.halo_w_magic), hx = pix - hy * HALO_W;
        const int iy = oy00 * STRIDE - PAD + (int)hy, ix = ix0 + (int)hx;
        const bool ok = (cg < (unsigned)G) & (pix < (unsigned)npix) & ((unsigned)iy < (unsigned)p.Hin) &
                        ((unsigned)ix < (unsigned)p.Win) & ((int)(c_lo + cg) < cin_groups);
        return ok ? (unsigned)(((iy * p.Win + ix) * p.Cin) * ESIZE + cg * 16) : 0x80000000u;
    };
    unsigned hv[MAXH];
    if (HOIST) {
#pragma unroll
        for (int jj = 0; jj < MAXH; ++jj) hv[jj] = halo_voff(wave + 4 * jj, 0);
    }
    const bool has_tail = cin_groups % G != 0;       // last chunk holds fewer than G k-groups per pixel
    auto issue_halo = [&](int c, int buf) {
        if (p.ablate & 10) return;
        char* const si = s_halo0 + buf * halo_bytes;
        const unsigned cbase = (unsigned)(c * G * 16);
        if (HOIST) {
            const bool tail = has_tail && c == p.cin_chunks - 1;
#pragma unroll
            for (int jj = 0; jj < MAXH; ++jj) {
                const int j = wave + 4 * jj;
                if (j < n_halo_instr) {
                    unsigned voff = hv[jj];
                    if (tail) { const int slot = j * 64 + lane; if (c * G + slot % SLOTS >= cin_groups) voff = 0x80000000u; }
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void*)(si + j * 1024), 16, voff, cbase, 0, 0);
                }
            }
        } else {
            for (int j = wave; j < n_halo_instr; j += 4) {
                const unsigned voff = halo_voff(j, c * G);   // (a call inside the builtin's arguments loses the host stub)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void*)(si + j * 1024), 16, voff, cbase, 0, 0);
            }
        }
    };

    unsigned long long* const trc = p.trace ? p.trace + (size_t)blockIdx.x * 16 : nullptr;
    auto stamp = [&](int slot) { if (trc && tid == 0 && slot < 16) trc[slot] = __builtin_amdgcn_s_memtime(); };
    if (trc && tid == 0) trc[0] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 20);
    stamp(1);
    issue_weights(0, 0); issue_halo(0, 0);      // chunk 0 is in flight: everything below until the wait is free.  (Weights
                                                // first measured 1.5 % faster than halo first, same box A/B.)

    // Coalesced epilogue (A) bookkeeping, done ONCE up front: the lane -> (pixel, 8-channel group) map and the 32-bit
    // element offsets of the wave's NI fragments inside its image serve the residual prefetch here and the stores at
    // the end (0xFFFFFFFF = nothing to store).  With registers to spare (RES_PF) the residual tile is requested now
    // and arrives under the main loop.
    constexpr int EPI_CO = MI * 16, EPI_GROUPS = EPI_CO / 8, EPI_ITEMS = 16 * EPI_GROUPS, EPI_ITERS = (EPI_ITEMS + 63) / 64;
    constexpr bool RES_PF = MI * NI <= 24 && conv_wgs_per_cu(KS, NI, MI, G) == 2;   // big accumulator sets leave no registers for the prefetch
    bf16x8 res_pf[RES_PF ? NI : 1][EPI_ITERS];
    unsigned eoff[RES_PF ? NI : 1][EPI_ITERS];
    const bool epi_a = GE == 8 && p.epi_lds && !p.out_f32;
    const size_t img_out = (size_t)n * p.Hout * p.Wout * p.out_cstride;
    auto epi_offset = [&](int j, int it) -> unsigned {
        const int f = wave * NI + j;
        const int fr = f >> TWF_LOG2, fx = f & (TWF - 1);
        const int oy = oy00 + fr;
        const int id = it * 64 + lane;
        const int px = id / EPI_GROUPS, grp = id - px * EPI_GROUPS;
        const int ox = ox0 + fx * 16 + px;
        const int co = nb * EPI_CO + grp * 8;
        const bool ok = (id < EPI_ITEMS) & (oy < p.Hout) & (ox < p.Wout) & (co < p.cout);
        return ok ? (unsigned)(((oy * p.Wout + ox) * p.out_cstride + p.out_coff + co) * 2) : 0x80000000u;      // bytes inside the image; out of range = nothing
    };
    // branch-free residual loads / output stores of epilogue A: per-image buffer descriptors, a missing residual is a zero-sized one
    // (loads return zeros), an item outside the image carries an out-of-range offset (epilogue F has the measurements)
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    const int img_b16 = p.Hout * p.Wout * p.out_cstride * 2;
    const __amdgpu_buffer_rsrc_t rs_res16 = __builtin_amdgcn_make_buffer_rsrc(
        p.res ? const_cast<char*>(reinterpret_cast<const char*>(p.res)) + img_out * 2 : const_cast<char*>(reinterpret_cast<const char*>(p.in)), 0,
        p.res ? img_b16 : 0, 0x00020000);
    if constexpr (GE == 8 && RES_PF) {
        if (epi_a) {
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int it = 0; it < EPI_ITERS; ++it) {
                    const unsigned o = epi_offset(j, it);
                    eoff[j][it] = o;
                    res_pf[j][it] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_res16, o, 0, 0));
                }
        }
    }

    for (int c = 0; c < p.cin_chunks; ++c) {
        if (c > 0) {
            asm volatile("s_barrier" ::: "memory");                       // everyone finished reading the buffers
            issue_weights(c, 0); issue_halo(c, 0);
            if (c == 1) stamp(12);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // my pieces of chunk c have landed
        if (c == 1) stamp(13);
        asm volatile("s_barrier" ::: "memory");                           // ... everyone's; the MFMAs of chunk c-1 are over
        stamp(2 + 2 * c);
        const char* const s_w = smem;
        const char* const s_in = s_halo0;
        // fragment offsets of k-step s (compile-time tap arithmetic when the 4 k-groups of a step share a tap)
        auto frag_off = [&](int s) -> int {
            if constexpr (G % 4 == 0) {
                const int tap = (4 * s) / G, cg0 = (4 * s) % G;
                return (tap / KS) * row_pitch + (tap % KS) * PS + (cg0 + g) * 16;
            } else {
                int kg = 4 * s + g;
                kg = kg < NKG ? kg : NKG - 1;     // padded k-groups: weights are zero, address must stay valid
                const int tap = kg / G, cg = kg - tap * G;
                const int dy = tap / KS, dx = tap - dy * KS;
                return dy * row_pitch + dx * PS + cg * 16;
            }
        };
        if (!(p.ablate & 1)) {
        auto mma = [&](f32x4& d, const frag& av, const frag& bv) {
            if constexpr (GE == 8) {
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, d, 0, 0, 0);
            } else if constexpr (Elem<T>::X3) {
                (void)av; (void)bv;
            } else {
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], bv[0], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], bv[1], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], bv[2], d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], bv[3], d, 0, 0, 0);
            }
        };
        // software pipeline over k-steps: fragments of step s+1 are fetched from LDS while step s multiplies
        frag a[2][MI], b[2][NI];
        x3h x3_wprev[Elem<T>::X3 ? MI : 1][4], x3_hprev[Elem<T>::X3 ? NI : 1][4];      // x3_t: hi halves of the even k-step of a pair
        {
            const int off = frag_off(0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[0][mi] = *reinterpret_cast<const frag*>(s_w + (mi * 64 + lane) * 16);
#pragma unroll
            for (int j = 0; j < NI; ++j) b[0][j] = *reinterpret_cast<const frag*>(s_in + boff[j] + off);
        }
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            if (s + 1 < NKS) {
                const int off = frag_off(s + 1);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    a[nxt][mi] = *reinterpret_cast<const frag*>(s_w + (((s + 1) * MI + mi) * 64 + lane) * 16);
#pragma unroll
                for (int j = 0; j < NI; ++j) b[nxt][j] = *reinterpret_cast<const frag*>(s_in + boff[j] + off);
            }
            if constexpr (Elem<T>::X3) {
                // one k-step = 4 k-groups x 4 channels: lane group g holds k-group 4 s + g.  Two v_mfma_f32_16x16x32_bf16 per tile with
                // A = [w_hi | w_lo] as packed (element e of A meets element e of B): B = [x_lo | x_hi] gives the cross terms
                // w_hi.x_lo + w_lo.x_hi, B = [x_hi | 0] the main term -- small terms first.
                // (The main term as v_mfma_f32_16x16x16_bf16 on the low halves was built first: same issue time, two registers fewer per
                // B fragment -- and WRONG on gfx950 as hipcc 7.2 schedules it: back to back behind the K = 32 instruction whose vDst it
                // reads as SrcC, with no wait states; accumulator registers 0 and 1 of such tiles came out wrong.  Found by the per-kernel
                // test, isolated by tools/dev/mfma_pair_probe.hip.)
                // Round 4: the main terms of TWO consecutive k-steps share one instruction -- A = [w_hi(s) | w_hi(s + 1)] against
                // B = [x_hi(s) | x_hi(s + 1)], both halves taken from registers the steps hold anyway -- so a pair of steps costs 3 MFMAs
                // instead of 4 (an odd last step keeps its own main term against [x_hi | 0]).
                x3h8 bx[NI];
                x3h hcur[NI][4];
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    x3h l[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) X3_SPLIT1(b[cur][j][e], hcur[j][e], l[e]);
                    bx[j] = x3h8{l[0], l[1], l[2], l[3], hcur[j][0], hcur[j][1], hcur[j][2], hcur[j][3]};
                }
                const bool second = (s & 1) != 0, alone = !second && s + 1 == NKS;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const x3h8 aw = __builtin_bit_cast(x3h8, a[cur][mi]);
#pragma unroll
                    for (int j = 0; j < NI; ++j) acc[mi][j] = X3_MFMA_16x16x32(aw, bx[j], acc[mi][j]);
                    if (second || alone) {
                        const x3h z = (x3h)0.0f;
                        const x3h8 am = second ? x3h8{x3_wprev[mi][0], x3_wprev[mi][1], x3_wprev[mi][2], x3_wprev[mi][3], aw[0], aw[1], aw[2], aw[3]} : aw;
#pragma unroll
                        for (int j = 0; j < NI; ++j) {
                            const x3h8 bm = second ? x3h8{x3_hprev[j][0], x3_hprev[j][1], x3_hprev[j][2], x3_hprev[j][3], hcur[j][0], hcur[j][1], hcur[j][2], hcur[j][3]}
                                                   : x3h8{hcur[j][0], hcur[j][1], hcur[j][2], hcur[j][3], z, z, z, z};
                            acc[mi][j] = X3_MFMA_16x16x32(am, bm, acc[mi][j]);
                        }
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) x3_wprev[mi][e] = aw[e];
                    }
                }
                if (!second && !alone) {
#pragma unroll
                    for (int j = 0; j < NI; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) x3_hprev[j][e] = hcur[j][e];
                }
            } else {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int j = 0; j < NI; ++j) mma(acc[mi][j], a[cur][mi], b[cur][j]);
            }
            if (s + 1 < NKS) __builtin_amdgcn_sched_barrier(0);    // keep the prefetch ahead of the next step's MFMAs
        }
        }
        stamp(3 + 2 * c);
    }

    // ---- epilogue A (bf16 outputs): transpose through LDS (the staging
    // buffers are free now), then every lane handles 8 consecutive channels of one pixel: residual load and
    // output store are 16 bytes per lane and contiguous per pixel row (a fragment row is one 16 x Cout*2 B
    // contiguous span) instead of 8-byte pieces scattered over 16 pixel rows.
    if constexpr (GE == 8) {
        if (epi_a) {
            constexpr int CO = MI * 16, PITCH = CO + 4;                   // floats per staged pixel row
            asm volatile("s_barrier" ::: "memory");                        // all waves are done with the staging buffers
            stamp(10);
            constexpr int JB = conv_epi_frags(KS, NI, MI, G);                              // fragments staged at a time
            float* stg = reinterpret_cast<float*>(smem) + wave * (JB * 16 * PITCH);
            constexpr int GROUPS = EPI_GROUPS, EITERS = EPI_ITERS;
            const __amdgpu_buffer_rsrc_t rs_out16 = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(p.out) + img_out * 2, 0, img_b16, 0x00020000);
            // item offsets: kept from the prologue when the residual was prefetched; otherwise computed here and the
            // residual is requested for ALL fragments first -- the fragment registers of the main loop are dead --
            // so its latency is paid once, not once per staged block
            unsigned off[NI][EITERS];
            bf16x8 rr[RES_PF ? 1 : NI][RES_PF ? 1 : EITERS];
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int it = 0; it < EITERS; ++it) {
                    if constexpr (RES_PF) {
                        off[j][it] = eoff[j][it];
                    } else {
                        off[j][it] = epi_offset(j, it);
                        rr[j][it] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_res16, off[j][it], 0, 0));
                    }
                }
#pragma unroll
            for (int j0 = 0; j0 < NI; j0 += JB) {
#pragma unroll
                for (int jj = 0; jj < JB; ++jj)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        const int j = j0 + jj;
                        *reinterpret_cast<float4*>(stg + (jj * 16 + ln) * PITCH + mi * 16 + g * 4) =
                            make_float4(acc[mi][j][0], acc[mi][j][1], acc[mi][j][2], acc[mi][j][3]);
                    }
                // wave-local hand-off: LDS operations of one wave complete in order
                if (j0 == 0) stamp(11);
#pragma unroll
                for (int jj = 0; jj < JB; ++jj) {
#pragma unroll
                    for (int it = 0; it < EITERS; ++it) {
                        {
                            const int id = it * 64 + lane;
                            const int px = id / GROUPS, grp = id - px * GROUPS;
                            const float* sp = stg + (jj * 16 + (EPI_ITEMS % 64 ? min(px, 15) : px)) * PITCH + grp * 8;      // (lanes beyond the last item store nothing)
                            const float4 lo = *reinterpret_cast<const float4*>(sp), hi = *reinterpret_cast<const float4*>(sp + 4);
                            float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                            if (p.res) {
                                bf16x8 r;
                                if constexpr (RES_PF) r = res_pf[j0 + jj][it]; else r = rr[j0 + jj][it];
#pragma unroll
                                for (int e = 0; e < 8; ++e) v[e] += (float)r[e];
                            }
                            bf16x8 q;
#pragma unroll
                            for (int e = 0; e < 8; ++e) q[e] = (__bf16)v[e];
                            if (p.relu) {
                                // ReLU on the packed bf16 pairs: rounding keeps the sign, so relu(round(x)) == round(relu(x)); a bf16
                                // with the sign bit set is a negative int16 -> v_pk_max_i16 against 0, one instruction per two values
                                typedef short s16x8 __attribute__((ext_vector_type(8)));
                                const s16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
                                q = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, q), zero));
                            }
                            u32x4_t qd = __builtin_bit_cast(u32x4_t, q);
                            asm volatile("" : "+v"(qd));
                            __builtin_amdgcn_raw_buffer_store_b128(qd, rs_out16, off[j0 + jj][it], 0, 0);
                            asm volatile("s_nop 1" :: "v"(qd) : "memory");     // store data stays untouched behind the store (NOTES/design_history_r1_r5.md §9.1)
                        }
                    }
                }
            }
            stamp(15);
            if constexpr (Elem<T>::X3) x3_report(amax, p.range);
            return;
        }
    }
    // ---- epilogue F (fp32 outputs: the exact-fp32 and the bf16x3 engine): the same transpose through LDS, every lane then handles 4
    // consecutive channels of one pixel -- residual load and output store are 16 bytes per lane and MI * 64 contiguous bytes per pixel
    // (the direct epilogue below writes 64-byte pieces per pixel and instruction; with it the 1x1 convolutions of layer1, 2 GB in and
    // 2 GB out per launch, ran at 3 TB/s with or without their MFMA phase).  Same arithmetic, same bits.  BRANCH-FREE like conv_tt's:
    // loads and stores go through buffer descriptors of one image, an item outside the image / beyond Cout carries an out-of-range
    // offset (loads return zeros, stores are dropped), a missing residual / output / twin is a zero-sized descriptor.  (With `if (ok)`
    // around every load hipcc branched around each one and spilled the accumulators: 164-392 bytes of scratch per lane.)
    if constexpr (GE == 4) {
        if (p.epi_lds) {
            constexpr int CO = MI * 16, PITCH = CO + 4, GROUPS = CO / 4, EITERS = MI;      // 16 pixels x GROUPS items = 64 * MI
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            asm volatile("s_barrier" ::: "memory");                        // all waves are done with the staging buffers
            constexpr bool PRE_ALL = NI * MI <= 8;
            constexpr int JBH = conv_epi_frags(KS, NI, MI, G);                   // fragments per wave the host reserved staging for
            constexpr int JB = PRE_ALL ? JBH : 1;                                // big register tiles: one fragment at a time, the next one's residual in flight
            float* stg = reinterpret_cast<float*>(smem) + wave * (JBH * 16 * PITCH);
            int lane_l = lane;               // laundered: hipcc would hoist the lane-only item arithmetic above the main loop
            asm volatile("" : "+v"(lane_l));
            const int img_b = p.Hout * p.Wout * p.out_cstride * 4;              // bytes of one image of the output tensor (< 2 GB)
            char* const in_c = const_cast<char*>(reinterpret_cast<const char*>(p.in));
            const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
                p.res ? const_cast<char*>(reinterpret_cast<const char*>(p.res)) + img_out * 4 : in_c, 0, p.res ? img_b : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
                p.out ? reinterpret_cast<char*>(p.out) + img_out * 4 : in_c, 0, p.out ? img_b : 0, 0x00020000);
            const bool has_twin = Elem<T>::X3 && p.out_twin != nullptr;         // dense output (out_coff 0, stride = cout): same image size
            const __amdgpu_buffer_rsrc_t rs_tw = __builtin_amdgcn_make_buffer_rsrc(
                has_twin ? reinterpret_cast<char*>(p.out_twin) + img_out * 4 : in_c, 0, has_twin ? img_b : 0, 0x00020000);
            // item offsets + residuals: for ALL fragments up front when the registers allow (the latency is paid once), else one
            // fragment ahead
            constexpr int NR = PRE_ALL ? NI : 2 * JB;
            unsigned off[NR][EITERS];
            u32x4 rr[NR][EITERS];
            auto fetch = [&](int j, int slot) __attribute__((always_inline)) {
                const int f = wave * NI + j;
                const int fr = f >> TWF_LOG2, fx = f & (TWF - 1);
                const int oy = oy00 + fr;
#pragma unroll
                for (int it = 0; it < EITERS; ++it) {
                    const int id = it * 64 + lane_l;
                    const int px = id / GROUPS, grp = id - px * GROUPS;
                    const int ox = ox0 + fx * 16 + px;
                    const int co = nb * CO + grp * 4;
                    const bool ok = (oy < p.Hout) & (ox < p.Wout) & (co < p.cout);
                    off[slot][it] = ok ? (unsigned)(((oy * p.Wout + ox) * p.out_cstride + p.out_coff + co) * 4) : 0x80000000u;
                    rr[slot][it] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, off[slot][it], 0, 0);
                }
            };
            if constexpr (PRE_ALL) {
#pragma unroll
                for (int j = 0; j < NI; ++j) fetch(j, j);
            } else {
#pragma unroll
                for (int jj = 0; jj < JB; ++jj) fetch(jj, jj);
            }
#pragma unroll
            for (int j0 = 0; j0 < NI; j0 += JB) {
                const int base = PRE_ALL ? j0 : ((j0 / JB) & 1) * JB;              // slots of this block
                if constexpr (!PRE_ALL) {
                    if (j0 + JB < NI) {
#pragma unroll
                        for (int jj = 0; jj < JB; ++jj) fetch(j0 + JB + jj, (((j0 / JB) + 1) & 1) * JB + jj);      // the next block's, under this block's work
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int jj = 0; jj < JB; ++jj)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        const int j = j0 + jj;
                        *reinterpret_cast<float4*>(stg + (jj * 16 + ln) * PITCH + mi * 16 + g * 4) =
                            make_float4(acc[mi][j][0], acc[mi][j][1], acc[mi][j][2], acc[mi][j][3]);
                    }
                // wave-local hand-off: LDS operations of one wave complete in order
#pragma unroll
                for (int jj = 0; jj < JB; ++jj) {
#pragma unroll
                    for (int it = 0; it < EITERS; ++it) {
                        const unsigned o = off[base + jj][it];
                        const int id = it * 64 + lane_l;
                        const int px = id / GROUPS, grp = id - px * GROUPS;
                        const float4 a4 = *reinterpret_cast<const float4*>(stg + (jj * 16 + px) * PITCH + grp * 4);
                        const u32x4 r = rr[base + jj][it];
                        float v[4] = {a4.x + __uint_as_float(r[0]), a4.y + __uint_as_float(r[1]), a4.z + __uint_as_float(r[2]), a4.w + __uint_as_float(r[3])};
                        if (p.relu) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                        }
                        u32x4 ov = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                        if constexpr (Elem<T>::X3) {
                            x3_track(amax, v[0], v[1]); x3_track(amax, v[2], v[3]);      // every OUTPUT is tracked where it is produced (x3.hpp)
                            if (has_twin) {        // wave-uniform: [16 hi | 16 lo] bf16 per pixel and 16-channel group, this lane's 4 channels are 8 + 8 bytes
                                x3h4 th, tl;
#pragma unroll
                                for (int e = 0; e < 4; ++e) X3_SPLIT1(v[e], th[e], tl[e]);
                                const unsigned c16 = (unsigned)(grp * 4) & 15u;          // (nb * CO is a multiple of 16)
                                const unsigned toff = o == 0x80000000u ? o : o - c16 * 2;  // byte offset of the group's hi half + this lane's 8 bytes
                                u32x2 t0 = __builtin_bit_cast(u32x2, th), t1 = __builtin_bit_cast(u32x2, tl);
                                asm volatile("" : "+v"(t0), "+v"(t1), "+v"(ov));
                                __builtin_amdgcn_raw_buffer_store_b64(t0, rs_tw, toff, 0, 0);
                                __builtin_amdgcn_raw_buffer_store_b64(t1, rs_tw, toff, 32, 0);
                                __builtin_amdgcn_raw_buffer_store_b128(ov, rs_out, o, 0, 0);
                                asm volatile("s_nop 3" :: "v"(t0), "v"(t1), "v"(ov) : "memory");     // store data stays untouched behind the stores (NOTES/design_history_r1_r5.md §9.1)
                                continue;
                            }
                        }
                        asm volatile("" : "+v"(ov));
                        __builtin_amdgcn_raw_buffer_store_b128(ov, rs_out, o, 0, 0);
                        asm volatile("s_nop 1" :: "v"(ov) : "memory");
                    }
                }
                if constexpr (!PRE_ALL) __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (Elem<T>::X3) x3_report(amax, p.range);
            return;
        }
    }
    // epilogue: (+ residual) (ReLU) -> store 4 consecutive channels per lane
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int f = wave * NI + j;
        const int fr = f >> TWF_LOG2, fx = f & (TWF - 1);
        const int oy = oy00 + fr, ox = ox0 + fx * 16 + ln;
        if (oy >= p.Hout || ox >= p.Wout) continue;
        const size_t pix = ((size_t)n * p.Hout + oy) * p.Wout + ox;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int cf = nb * MI + mi;
            if (cf >= p.cout_frags) continue;
            const int co = cf * 16 + g * 4;
            if (co >= p.cout) continue;                    // padded output channels are never stored
            float v0 = acc[mi][j][0], v1 = acc[mi][j][1], v2 = acc[mi][j][2], v3 = acc[mi][j][3];
            const size_t o = pix * p.out_cstride + p.out_coff + co;
            if (p.res) {
                if constexpr (GE == 8) {
                    const bf16x4 r = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const __bf16*>(p.res) + o);
                    v0 += (float)r[0]; v1 += (float)r[1]; v2 += (float)r[2]; v3 += (float)r[3];
                } else {
                    const float4 r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res) + o);
                    v0 += r.x; v1 += r.y; v2 += r.z; v3 += r.w;
                }
            }
            if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
            if constexpr (Elem<T>::X3) {
                x3_track(amax, v0, v1); x3_track(amax, v2, v3);              // every OUTPUT is tracked where it is produced (x3.hpp)
                // [16 hi | 16 lo] bf16 per pixel and 16-channel group (conv_tt_body.inc): this lane's 4 channels are 8 + 8 bytes
                if (p.out_twin) {
                    const float v[4] = {v0, v1, v2, v3};
                    x3h4 th, tl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) X3_SPLIT1(v[e], th[e], tl[e]);
                    char* const tw = reinterpret_cast<char*>(p.out_twin) + (pix * p.cout + (co & ~15)) * 4 + (co & 15) * 2;
                    *reinterpret_cast<x3h4*>(tw) = th;
                    *reinterpret_cast<x3h4*>(tw + 32) = tl;
                }
                if (!p.out) continue;
            }
            if (p.out_f32 || GE == 4) {
                float* dst = reinterpret_cast<float*>(p.out) + o;
                if (co + 4 <= p.cout) *reinterpret_cast<float4*>(dst) = make_float4(v0, v1, v2, v3);
                else { dst[0] = v0; if (co + 1 < p.cout) dst[1] = v1; if (co + 2 < p.cout) dst[2] = v2; }
            } else {
                bf16x4 q;
                q[0] = (__bf16)v0; q[1] = (__bf16)v1; q[2] = (__bf16)v2; q[3] = (__bf16)v3;
                *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(p.out) + o) = q;
            }
        }
    }
    if constexpr (Elem<T>::X3) x3_report(amax, p.range);
}

template <typename T, int KS, int STRIDE, int NI, int MI, int G>
__global__ __launch_bounds__(256, conv_launch_wgs<T>(KS, NI, MI, G)) void conv_kernel(const ConvParams p) {
    const unsigned w = (blockIdx.x & 7u) * p.per_xcd + (blockIdx.x >> 3);
    if (w >= p.n_work) return;
    conv_body<T, KS, STRIDE, NI, MI, G>(p, w);
}

// Up to three INDEPENDENT convolutions of the same kernel variant in one launch (the same-depth convs of the parallel
// HRNet branches): their work items are concatenated, so the small grids of the low-resolution branches fill the tail of
// the big one and two launch gaps (~5 us each) disappear.  Same code per work item -> same bits.
struct ConvGroupParams {
    ConvParams p[3];        // members, most expensive work items first (they start first on every XCD)
    unsigned per_xcd[3];    // ceil(n_work / 8) of each member
    int n;
};

// Block b runs on XCD b % 8 and takes the (b / 8)-th item of that XCD's list: the XCD's contiguous slice of member 0, then
// of member 1, then of member 2 -- every XCD gets the same mix of cheap and expensive items (a plain concatenation gave
// whole XCDs to the 384-channel member, 3x the cost per item: 20 ms instead of 13), and each member keeps the
// neighbouring-tiles-in-one-L2 order of the single launch.
template <typename T, int KS, int STRIDE, int NI, int MI, int G>
__global__ __launch_bounds__(256, conv_launch_wgs<T>(KS, NI, MI, G)) void conv_group_kernel(const ConvGroupParams gp) {
    const unsigned k = blockIdx.x & 7u;
    unsigned j = blockIdx.x >> 3;
    int m = 0;
    if (j >= gp.per_xcd[0]) {
        j -= gp.per_xcd[0]; m = 1;
        if (gp.n < 2) return;
        if (j >= gp.per_xcd[1]) { j -= gp.per_xcd[1]; m = 2; if (gp.n < 3 || j >= gp.per_xcd[2]) return; }
    }
    // member parameters by scalar selects (a dynamically indexed kernel argument would be copied to scratch)
    ConvParams q = gp.p[0];
    unsigned px = gp.per_xcd[0];
    if (m == 1) { q = gp.p[1]; px = gp.per_xcd[1]; }
    if (m == 2) { q = gp.p[2]; px = gp.per_xcd[2]; }
    const unsigned w = k * px + j;
    if (w >= q.n_work) return;
    conv_body<T, KS, STRIDE, NI, MI, G>(q, w);
}

// Up to three stride-2 3x3 convolutions that read the SAME input tensor (the first convolutions of a HighResolutionModule's
// fuse-down chains: hrnet.py:195-214 builds, from branch j, one chain per lower-resolution output, and every chain starts on x[j]):
// launched one by one they fetched the 48-channel 135 x 240 tensor three times from HBM (PMC 1.72 x the algorithmic bytes for the class,
// VERDICT r4 weak 4).  Here the work items are ordered tile-major, member-minor: the workgroups that need one input tile are dispatched
// back to back on one XCD (block b runs on XCD b % 8, tiles in contiguous slices per XCD as everywhere), so one of them misses to HBM
// and the others hit that XCD's L2.  Members may be of different n-block variants (MI = 6 for 96-channel blocks, MI = 3 for the 48-channel
// chains); the tile grid is common (same input, same stride).  Same work items, same code per item: same bits as the single launches.
struct ConvSharedParams {
    ConvParams p[3];
    int mi[3];                  // MI of each member's variant (6 or 3)
    unsigned first[4];          // first item-in-tile of each member: {0, nblk0, nblk0 + nblk1, items per tile}
    unsigned tiles, tiles_per_xcd;
    int n;
};
template <typename T, int NI, int G>
__global__ __launch_bounds__(256, conv_launch_wgs<T>(3, NI, 6, G)) void conv_shared_s2_kernel(const ConvSharedParams sp) {
    const unsigned k = blockIdx.x & 7u, j = blockIdx.x >> 3;
    const unsigned ipt = sp.first[3];
    const unsigned tl = j / ipt, r = j - tl * ipt;
    const unsigned tile = k * sp.tiles_per_xcd + tl;
    if (tl >= sp.tiles_per_xcd || tile >= sp.tiles) return;
    const int m = r >= sp.first[2] ? 2 : r >= sp.first[1] ? 1 : 0;
    ConvParams q = sp.p[0];
    unsigned f0 = sp.first[0];
    int mi = sp.mi[0];
    if (m == 1) { q = sp.p[1]; f0 = sp.first[1]; mi = sp.mi[1]; }
    if (m == 2) { q = sp.p[2]; f0 = sp.first[2]; mi = sp.mi[2]; }
    const unsigned w = tile * (unsigned)q.nblk + (r - f0);
    if (mi == 6) conv_body<T, 3, 2, NI, 6, G>(q, w);
    else conv_body<T, 3, 2, NI, 3, G>(q, w);
}
void launch_conv_shared_s2_x3(const ConvSharedParams& sp, unsigned blocks, size_t lds, hipStream_t s);      // conv_x3.hip: x3_t, NI 2, G 3
// conv_s2p.hip: the same work (one to three stride-2 convolutions on one input; a single convolution = one member) on the pipelined
// persistent kernel; ticket = nine zeroed device words of the caller's stream
int launch_conv_s2p_x3(const ConvSharedParams& sp, unsigned* ticket, hipStream_t s);

// host-visible launcher table ---------------------------------------------------------------------
typedef void (*ConvLaunchFn)(const ConvParams&, dim3 grid, size_t lds, hipStream_t s);
typedef void (*ConvGroupLaunchFn)(const ConvGroupParams&, dim3 grid, size_t lds, hipStream_t s);

struct ConvVariant {
    int dtype;      // SNCAL_F32 / SNCAL_BF16
    int ks, stride, ni, mi, g;
    ConvLaunchFn launch;
    ConvGroupLaunchFn launch_group;     // nullptr: this variant has no grouped instantiation
};

template <typename T, int KS, int STRIDE, int NI, int MI, int G>
void conv_launch(const ConvParams& p, dim3 grid, size_t lds, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {   // > 64 KB dynamic LDS needs the opt-in attribute
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_kernel<T, KS, STRIDE, NI, MI, G>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    SNCAL_LAUNCH((conv_kernel<T, KS, STRIDE, NI, MI, G>), grid, dim3(256), lds, s, p);
}

template <typename T, int KS, int STRIDE, int NI, int MI, int G>
void conv_group_launch(const ConvGroupParams& gp, dim3 grid, size_t lds, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_group_kernel<T, KS, STRIDE, NI, MI, G>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    SNCAL_LAUNCH((conv_group_kernel<T, KS, STRIDE, NI, MI, G>), grid, dim3(256), lds, s, gp);
}

// grouped instantiations exist for the branch convolutions (3x3 stride 1, 96-channel n-blocks) and the fuse-up 1x1 convs
// only: compile time
template <typename T, int KS, int STRIDE, int NI, int MI, int G>
constexpr ConvGroupLaunchFn conv_group_fn() {
    if constexpr ((Elem<T>::GE == 8 && STRIDE == 1 && ((KS == 3 && MI == 6) || (KS == 1 && G == 8 && (MI == 3 || MI == 6)))) ||
                  (Elem<T>::X3 && STRIDE == 1 && KS == 1 && G == 8 && (MI == 3 || MI == 6)))      // bf16x3: the fuse-up 1x1 convs
        return &conv_group_launch<T, KS, STRIDE, NI, MI, G>;
    else return nullptr;
}

// registries filled by conv_bf16.hip / conv_f32.hip
const ConvVariant* conv_variants_bf16(int* n);
const ConvVariant* conv_variants_f32(int* n);
const ConvVariant* conv_variants_x3(int* n);       // conv_x3.hip: x3_t

}  // namespace sncal
