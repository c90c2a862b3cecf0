// Two-team persistent 3x3 stride-1 convolution for the wide HRNet branches (96 / 192 / 384 channels), bf16.
//
// Replaces, for these layers, the conv3x3 + eval-BN (+ residual) (+ ReLU) of BasicBlock
// (/root/reference/src/models/hrnet/hrnet.py:42-58) that the generic kernel (conv.hpp) runs as
// conv<bf16,k3,s1,NI3,MI6,G4>.  Same GEMM view (D[cout, pixel] = sum_k W[cout, k] X[k, pixel], k = chunk of 32 input
// channels -> tap -> channel), same bf16 operands, fp32 accumulation started at the folded-BN shift, same epilogue
// arithmetic.  The MFMA shape is v_mfma_f32_32x32x16_bf16 (the generic kernel: 16x16x32): the products of one output are
// summed in a different order inside the matrix unit, so outputs agree with the generic kernel to fp32 rounding of the
// accumulator, not bit for bit (the first version of this kernel used 16x16x32 and WAS bit-identical; a phase trace
// showed 216 of them take 4.1k clk per wave where 108 of the 32x32x16 take 3.5k -- MI355X_MICROARCH.md: 2075 vs 2382 TF).
//
// Why another kernel.  The generic kernel stages a chunk (54 KB of weights + the halo tile), waits, multiplies, and
// relies on a second resident workgroup to fill the wait.  Measured: the two workgroups drift into
// phase, the LDS-DMA issue of one (~100 clk per 1 KB piece on the issuing wave) lands on top of its own MFMA phase
// as often as under the other's, and the MFMA pipe is busy 40 % of the kernel although the work is MFMA-bound.
// Here the alternation is BUILT IN:
//   * one workgroup of 8 waves per CU = two TEAMS of 4 waves; wave i of team A and wave i of team B share SIMD i;
//   * one team MULTIPLIES the stage it has in LDS (9 k-steps x 24 MFMAs per wave, one wave per SIMD issuing back to
//     back) while the other team LOADS its next stage by LDS-DMA (its waves carry the whole issue cost and the wait);
//     a token in LDS makes the two MFMA phases mutually exclusive, so the roles swap by themselves (see the main loop);
//   * a team's tile is 8 x 32 output pixels x 96 output channels (4 x 6 accumulator fragments per wave), a stage is
//     32 input channels: 54 KB of weights + a 10 x 34 pixel halo (22.5 KB); two teams = 160 KB of LDS, all of it;
//   * the epilogue of a finished tile runs at the head of the team's next LOAD phase; each wave transposes its fragments
//     through the block of the weight region that it alone refills afterwards, so no team-level barrier is needed;
//   * weights per MAC drop by a quarter against the 192-pixel tile of the generic kernel, and no DMA instruction is
//     ever issued by a wave that has MFMAs to issue.
// Frames are STACKED: output row index s = f * (H + 1) + y, every frame followed by one shared all-zero row (it is the
// bottom padding of frame f and the top padding of frame f + 1).  Tiles of 8 stacked rows then cut the whole batch with
// 1 / (H + 1) waste instead of (ceil(H / 8) * 8 - H) / H per frame -- 6 % instead of 41 % on the 17 x 30 branch.
// The halo image in LDS is [pixel][4 x 16 B] at a 36-pixel row pitch with the four 8-channel groups of a pixel rotated
// by (pixel >> 2) & 3: every ds_read_b128 lane group of a B fragment (32 consecutive pixels, one channel group per
// half-wave) then touches 16 distinct 16-byte bank slots (enumerated against MI355X_MICROARCH.md "LDS"), without the 50 % pitch padding of the generic kernel; the
// rotation is applied on the DMA's per-lane SOURCE address, the LDS destination stays lane-linear.
// Work items (tile x 96-channel block, cost = Cin / 32 stages) of up to three member convolutions are dealt to the
// 2 x 256 teams on the host (conv_tt_plan in hrnet.cpp): contiguous slices per XCD, longest-processing-time first.
#include "common.hpp"
#include "conv_tt.hpp"
#include <cstddef>

namespace sncal {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((address_space(3))) void lds_void;

namespace {
constexpr int MB = 3, NB = 2, NT = 18;              // 32-channel blocks, 32-pixel blocks (= tile rows) per wave, K = 16 steps per stage
constexpr int NKS = 9, MI = 6;                      // taps; 1 KB weight pieces per tap
constexpr int W_BYTES = NKS * MI * 1024;            // 55296: one stage of weights, fragment order [k-step][mi][lane] x 16 B
constexpr int HP = 36;                              // halo row pitch in pixels (34 used)
constexpr int HROWS = TT_TH + 2;                    // 10
constexpr int HALO_PIECES = (HROWS * HP * 64 + 1023) / 1024;    // 23 DMA pieces of 1 KB
constexpr int H_BYTES = HALO_PIECES * 1024;         // 23552
constexpr int TEAM_BYTES = W_BYTES + H_BYTES;       // 78848; two teams = 157696 B of the CU's 163840
// weight pieces per wave: wave tw owns pieces [wp_first(tw), wp_first(tw + 1)) of the 54 = 14, 14, 13, 13 -- a CONTIGUOUS
// block, because the wave also stages its epilogue there (below)
__device__ __host__ constexpr int wp_first(int tw) { return tw * 14 - (tw > 3 ? 2 : tw > 2 ? 1 : 0); }
constexpr int EPI_PITCH = TT_COUT + 4;              // floats per staged pixel row
constexpr int EPI_GROUPS = TT_COUT / 8, EPI_ITEMS = 32 * EPI_GROUPS, EPI_ITERS = EPI_ITEMS / 64;   // 12, 384, 6
static_assert(wp_first(4) == NKS * MI && 32 * EPI_PITCH * 4 <= 13 * 1024, "a wave's epilogue staging must fit its own block of the weight region");
}  // namespace

// member m's parameters straight from the kernel-argument segment, by scalar loads at a computed offset (selecting among three
// by-value copies of the argument keeps ~80 SGPRs alive and spills; a dynamically indexed argument is copied to scratch)
__device__ __forceinline__ TTMember load_member(int m) {
    TTMember r;
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(sizeof(TTMember) % 4 == 0 && offsetof(TTParams, m) == 0, "the members must open the kernel-argument segment");
    const __attribute__((address_space(4))) unsigned* src =
        (const __attribute__((address_space(4))) unsigned*)__builtin_amdgcn_kernarg_segment_ptr() + m * (int)(sizeof(TTMember) / 4);
    unsigned* dst = reinterpret_cast<unsigned*>(&r);
#pragma unroll
    for (unsigned i = 0; i < sizeof(TTMember) / 4; ++i) dst[i] = src[i];
#else
    (void)m;
    r = TTMember{};
#endif
    return r;
}

// FP8 = false: bf16 operands, v_mfma_f32_32x32x16_bf16, stages of 32 input channels.
// FP8 = true (BASELINE config C5): OCP e4m3 operands -- the input is the e4m3 twin of the activation tensor (per-tensor scale),
// the weights carry one scale per output channel -- on v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales: stages of 64
// input channels in the same 54 KB + 22.5 KB of LDS, i.e. half the stages and half the DMA bytes per MAC, twice the MACs per
// matrix-pipe cycle.  y = sum * (input scale x weight scale) + folded-BN shift is applied in the epilogue; the output goes out
// as bf16 and / or as the e4m3 twin the next fp8 convolution reads.
// MODE 2 (X3, the fp32-class engine `bf16x3`): fp32 activations and weights split into bf16 hi + bf16 lo (x = hi + lo to 2^-17),
// y = hi.hi + hi.lo + lo.hi accumulated in fp32 -- three v_mfma_f32_32x32x16_bf16 per product where the exact-fp32 engine's
// v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 rate.  Measured against the plain fp32 forward (tools/x3_sim.py arithmetic):
// |dlogp| 5e-6 mean / 1.4e-4 max, identical keypoint indices on every row.  The operand tensors are "split twins": per pixel and
// 16-channel group [16 hi | 16 lo] bf16 -- byte for byte a bf16 tensor of 2 C pseudo-channels, so the halo / weight staging of the
// bf16 mode is used unchanged (a stage = 16 real channels); only the MFMA pairing and the fp32 epilogue differ.
template <int MODE>
__global__ __launch_bounds__(512, 2) void conv_tt_kernel(const TTParams P) {
    constexpr bool FP8 = MODE == 1, X3 = MODE == 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = wv >> 2, tw = wv & 3;
    const unsigned T = blockIdx.x * 2u + (unsigned)team;
    char* const s_w = smem + team * TEAM_BYTES;
    char* const s_h = s_w + W_BYTES;
    float* const s_bias = reinterpret_cast<float*>(smem + 2 * TEAM_BYTES + 64);     // folded-BN shifts of the members, back to back
    float* const s_osc = s_bias + TT_TABLE_MAX;                                     // fp8 variant: output scales, same indexing
    const int tab1 = P.m[0].cout, tab2 = P.m[0].cout + P.m[1].cout;                 // table offsets of members 1 and 2

    unsigned it = P.team_first[T];
    const unsigned SA = P.team_stages[blockIdx.x * 2u], SB = P.team_stages[blockIdx.x * 2u + 1u];

    // B fragment of K = 16 step (tap (dy, dx), channel half h) for tile row jr of this wave: pixel (tw * 2 + jr + dy, l31 + dx),
    // channel group cg = 2 h + hi, stored in slot (rot + cg) & 3 with rot = (pixel >> 2) & 3 = (row + ((l31 + dx) >> 2)) & 3
    // (row pitch 36 = 9 * 4).  bptr[dx][k] carries everything that depends on the lane for k = (jr + dy + 2 h) & 3; the rest of
    // the address, (jr + dy) * 36 * 64, is an immediate of the ds_read.
    const char* bptr[3][4];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            bptr[dx][k] = s_h + ((tw * 2 * HP + l31 + dx) * 64 + ((tw * 2 + ((l31 + dx) >> 2) + (FP8 ? 2 * hi : hi) + k) & 3) * 16);
    const char* const aptr = s_w + lane * 16;

    unsigned long long* const trc = P.trace ? P.trace + (size_t)T * 256 : nullptr;
    unsigned tix = 0, eix = 192;          // [0,192): six stamps per stage; [192,256): four stamps per epilogue
    auto stamp = [&]() { if (trc && tid == team * 256 && tix < 192) trc[tix++] = __builtin_amdgcn_s_memtime(); };
    auto estamp = [&]() { if (trc && tid == team * 256 && eix < 256) trc[eix++] = __builtin_amdgcn_s_memtime(); };

    // ---- state of the team's current item ---------------------------------------------------------------------------
    // Beside a multiplying partner wave, a dependent VALU instruction of the loading wave completes every 5-8 clk and a
    // vector-memory instruction costs ~160 clk (phase traces, tools/tt_trace.py): per-lane address arithmetic and the 24
    // loads / stores of an epilogue are the expensive parts of a LOAD phase.  So the halo offsets are computed once per
    // item (not per stage), and the epilogue's addresses are split into a wave-uniform scalar part (row, tile column,
    // channel block: SALU) and per-lane constants.
    f32x16 acc[MB][NB];
    unsigned hv[6];                       // per-lane byte offsets of this wave's halo DMA pieces (stage-independent)
    TTMember M = load_member(0);
    int nb = 0, c = 0, row0 = 0, col0 = 0, tab = 0;

    // byte offset of this lane's 16 bytes of halo piece `piece`: slot q of the [pixel][4] image holds channel group
    // (q & 3) - (pixel >> 2) mod 4; pitch padding, rows / columns outside the frame and the shared zero row between
    // stacked frames get an out-of-range offset -> the DMA writes zeros.  (Splitting this into per-lane kernel constants and
    // a wave-uniform scalar part per item -- 42 instead of ~200 VALU instructions -- was built and measured SLOWER: the
    // scalar part pushed the kernel over the 102-SGPR budget and the spills cost more than the VALU work saved.)
    auto halo_voff = [&](int piece, int lane) -> unsigned {
        const unsigned q = (unsigned)(piece * 64 + lane);
        const unsigned p = q >> 2;
        const unsigned hrow = (p * 1821u) >> 16, hcol = p - hrow * HP;           // p / 36 for p < 2048
        const unsigned cg = ((q & 3u) - (p >> 2)) & 3u;
        const int s = row0 - 1 + (int)hrow;
        const unsigned f = __umulhi((unsigned)max(s, 0), M.hp1_magic);
        const int y = s - (int)f * (M.H + 1);
        const int x = col0 - 1 + (int)hcol;
        const bool ok = (hrow < (unsigned)HROWS) & (hcol < 34u) & (s >= 0) & ((int)f < M.N) & (y < M.H) & ((unsigned)x < (unsigned)M.W);
        return ok ? (unsigned)((((int)f * M.H + y) * M.W + x) * M.Cin * (FP8 ? 1 : 2)) + cg * 16u : 0x80000000u;
    };

    auto setup_item = [&](const TTItem I) {
        M = load_member(I.member);
        nb = I.nb; row0 = I.row0; col0 = I.col0;
        int lane_l = lane;                 // laundered: hipcc would hoist the lane-only parts out of the stage loop and spill them
        asm volatile("" : "+v"(lane_l));
#pragma unroll
        for (int jj = 0; jj < 6; ++jj) hv[jj] = tw + 4 * jj < HALO_PIECES ? halo_voff(tw + 4 * jj, lane_l) : 0x80000000u;
        if (eix > 192) estamp();
        // accumulators start at the folded-BN shift: register quad q of block mb holds channels mb * 32 + 8 q + 4 hi + 0..3 of
        // pixel l31.  The shifts come from a table in LDS (filled once per workgroup): a global load here would sit behind the
        // old tile's stores and its wait, vmcnt(0), would last until every store has drained
        tab = I.member == 0 ? 0 : I.member == 1 ? tab1 : tab2;
        if constexpr (FP8) {               // fp8: plain sums; scale and shift are applied in the epilogue
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int jr = 0; jr < NB; ++jr)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mb][jr][r] = 0.f;
        } else {
            const float* const bt = s_bias + tab + nb * TT_COUT + 4 * hi;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 bs = *reinterpret_cast<const float4*>(bt + mb * 32 + 8 * q);
#pragma unroll
                    for (int jr = 0; jr < NB; ++jr) {
                        acc[mb][jr][4 * q + 0] = bs.x; acc[mb][jr][4 * q + 1] = bs.y; acc[mb][jr][4 * q + 2] = bs.z; acc[mb][jr][4 * q + 3] = bs.w;
                    }
                }
        }
    };

    auto setup_done_stamp = [&]() { if (eix > 192) estamp(); };
    auto issue_stage = [&](int cc) {
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(M.in), 0, (int)M.in_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(M.w), 0, (int)M.w_bytes, 0x00020000);
        const unsigned wbase = (unsigned)((nb * M.chunks + cc) * W_BYTES);
        const int i1 = wp_first(tw + 1);
        for (int i = wp_first(tw); i < i1; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(s_w + i * 1024), 16, (unsigned)(lane * 16), wbase + i * 1024, 0, 0);
        const unsigned cbase = (unsigned)(cc * 64);
#pragma unroll
        for (int jj = 0; jj < 6; ++jj)
            if (tw + 4 * jj < HALO_PIECES)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void*)(s_h + (tw + 4 * jj) * 1024), 16, hv[jj], cbase, 0, 0);
    };

    // (+ residual) (ReLU) -> bf16, through a wave-private LDS transpose so that every lane stores 8 consecutive channels.
    // BRANCH-FREE: residual loads and output stores go through buffer descriptors, an item with nothing to store carries an
    // out-of-range offset (loads return zeros, stores are dropped), a missing residual is a zero-sized descriptor.  (The
    // first version guarded every 16-byte load with `if (res && valid)`: hipcc then branches around each load and waits for
    // it before the next -- twelve dependent HBM round trips, 12k clk per tile against a 4k clk MFMA phase of the other team.)
    auto epilogue_x3 = [&]() __attribute__((always_inline)) {
        // fp32 in, fp32 out: (+ residual) (ReLU), 8 channels = 32 bytes per lane and item = two b128 accesses each way; channel blocks
        // beyond the layer's width (a 48-channel layer runs as one 96-channel item with zero weights above 48) are not stored
        const unsigned out_bytes = (unsigned)(M.N * M.H * M.W * M.out_cstride * 4);
        const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(M.res ? M.res : M.in), 0, M.res ? (int)out_bytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(M.out ? M.out : const_cast<void*>(M.in), 0, M.out ? (int)out_bytes : 0, 0x00020000);
        // split twin of the output for the next bf16x3 convolution ([16 hi | 16 lo] bf16 per 16-channel group: 4 bytes per element like
        // the fp32 tensor; written here the consumer needs no separate split pass).  Twin tensors are dense: stride = the layer's width.
        const unsigned twin_bytes = (unsigned)(M.N * M.H * M.W * M.cout * 4);
        const __amdgpu_buffer_rsrc_t rs_tw = __builtin_amdgcn_make_buffer_rsrc(M.out8 ? M.out8 : const_cast<void*>(M.in), 0, M.out8 ? (int)twin_bytes : 0, 0x00020000);
        const bool has_twin = M.out8 != nullptr;
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const bool has_res = M.res != nullptr;
        estamp();
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));
        float* const stg = reinterpret_cast<float*>(s_w + wp_first(tw) * 1024);
        estamp();
#pragma unroll
        for (int jr = 0; jr < NB; ++jr) {
            const int srow = row0 + tw * 2 + jr;
            const unsigned f = __umulhi((unsigned)srow, M.hp1_magic);
            const int y = srow - (int)f * (M.H + 1);
            const bool row_ok = ((int)f < M.N) & (y < M.H);
            const unsigned soff = row_ok ? (unsigned)(((((int)f * M.H + y) * M.W + col0) * M.out_cstride + M.out_coff + nb * TT_COUT) * 4) : 0u;
            const unsigned soff_tw = row_ok ? (unsigned)(((((int)f * M.H + y) * M.W + col0) * M.cout + nb * TT_COUT) * 4) : 0u;
            if (jr == 1) estamp();
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(stg + l31 * EPI_PITCH + mb * 32 + 8 * q + 4 * hi) =
                        make_float4(acc[mb][jr][4 * q], acc[mb][jr][4 * q + 1], acc[mb][jr][4 * q + 2], acc[mb][jr][4 * q + 3]);
#pragma unroll
            for (int e = 0; e < EPI_ITERS; ++e) {
                const int id = e * 64 + lane_l;
                const int px = id / EPI_GROUPS, grp = id - px * EPI_GROUPS;
                const bool ok = row_ok & (px < M.W - col0) & (nb * TT_COUT + grp * 8 < M.cout);
                const unsigned voff = ok ? (unsigned)((px * M.out_cstride + grp * 8) * 4) : 0x80000000u;
                const float* sp = stg + px * EPI_PITCH + grp * 8;
                const float4 lo = *reinterpret_cast<const float4*>(sp), hi4 = *reinterpret_cast<const float4*>(sp + 4);
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi4.x, hi4.y, hi4.z, hi4.w};
                if (has_res) {
                    const u32x4 r0 = __builtin_amdgcn_raw_buffer_load_b128(rs_res, voff, soff, 0);
                    const u32x4 r1 = __builtin_amdgcn_raw_buffer_load_b128(rs_res, voff, soff + 16u, 0);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { v[k] += __uint_as_float(r0[k]); v[4 + k] += __uint_as_float(r1[k]); }
                }
                if (M.relu) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
                }
                u32x4 o0 = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                u32x4 o1 = {__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])};
                // hipcc leaves ONE wait state between a 16-byte store and the next write of its data registers (it reused the first data
                // register of store 1 as the address of store 2, and rewrote store 2's data right behind it); under memory load gfx950
                // reads store data later than that: isolated elements came out holding address bit patterns (the fp8 epilogue below met
                // the same).  So: both vectors complete in registers of their own, no VALU between the stores (the +16 rides in the
                // scalar offset), and the data registers stay live and untouched for four more wait states.
                asm volatile("" : "+v"(o0), "+v"(o1));
                __builtin_amdgcn_raw_buffer_store_b128(o0, rs_out, voff, soff, 0);
                __builtin_amdgcn_raw_buffer_store_b128(o1, rs_out, voff, soff + 16u, 0);
                asm volatile("s_nop 3" :: "v"(o0), "v"(o1) : "memory");
                if (has_twin) {                  // wave-uniform: hi = bf16(y), lo = bf16(y - hi), 16 bytes each, lo 32 bytes behind hi
                    bf16x8 th, tl;
#pragma unroll
                    for (int k = 0; k < 8; ++k) { th[k] = (__bf16)v[k]; tl[k] = (__bf16)(v[k] - (float)th[k]); }
                    u32x4 t0 = __builtin_bit_cast(u32x4, th), t1 = __builtin_bit_cast(u32x4, tl);
                    const unsigned voff_tw = ok ? (unsigned)(px * M.cout * 4 + (grp >> 1) * 64 + (grp & 1) * 16) : 0x80000000u;
                    asm volatile("" : "+v"(t0), "+v"(t1));
                    __builtin_amdgcn_raw_buffer_store_b128(t0, rs_tw, voff_tw, soff_tw, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(t1, rs_tw, voff_tw, soff_tw + 32u, 0);
                    asm volatile("s_nop 3" :: "v"(t0), "v"(t1) : "memory");
                }
            }
        }
        estamp();
    };
    auto epilogue = [&]() __attribute__((always_inline)) {
        if (P.ablate & 1) return;
        if constexpr (X3) { epilogue_x3(); return; }
        const unsigned out_bytes = (unsigned)(M.N * M.H * M.W * M.out_cstride * 2);
        const __amdgpu_buffer_rsrc_t rs_out8 = __builtin_amdgcn_make_buffer_rsrc(FP8 && M.out8 ? M.out8 : const_cast<void*>(M.in), 0,
                                                                                   FP8 && M.out8 ? (int)(out_bytes / 2) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(M.res ? M.res : M.in), 0,
                                                                                  M.res ? (int)out_bytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(M.out ? M.out : const_cast<void*>(M.in), 0,
                                                                                  M.out ? (int)out_bytes : 0, 0x00020000);     // no bf16 output: zero-sized
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const bool has_res = M.res != nullptr;      // the first conv of a BasicBlock has none: no loads, no unpack / add (a VALU
        estamp();                                   // instruction beside a multiplying partner costs ~8 clk, a load ~160)
        // item e of a tile row: lane -> pixel px of the row's 32 and 8-channel group grp (lane constants of the kernel)
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));
        unsigned lane_off[EPI_ITERS];
        int px_e[EPI_ITERS];
#pragma unroll
        for (int e = 0; e < EPI_ITERS; ++e) {
            const int id = e * 64 + lane_l;
            const int px = id / EPI_GROUPS, grp = id - px * EPI_GROUPS;
            px_e[e] = px;
            lane_off[e] = (unsigned)((px * M.out_cstride + grp * 8) * 2);
        }
        u32x4 rr[NB][EPI_ITERS];
        unsigned voff[NB][EPI_ITERS], soff[NB];
#pragma unroll
        for (int jr = 0; jr < NB; ++jr) {
            // wave-uniform part (SALU): stacked row -> (frame, row), channel block
            const int srow = row0 + tw * 2 + jr;
            const unsigned f = __umulhi((unsigned)srow, M.hp1_magic);
            const int y = srow - (int)f * (M.H + 1);
            const bool row_ok = ((int)f < M.N) & (y < M.H);
            soff[jr] = row_ok ? (unsigned)(((((int)f * M.H + y) * M.W + col0) * M.out_cstride + M.out_coff + nb * TT_COUT) * 2) : 0u;
#pragma unroll
            for (int e = 0; e < EPI_ITERS; ++e) {
                voff[jr][e] = (row_ok & (px_e[e] < M.W - col0)) ? lane_off[e] : 0x80000000u;      // out of range: loads 0, stores nothing
                rr[jr][e] = u32x4{0u, 0u, 0u, 0u};
                if (has_res) rr[jr][e] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, voff[jr][e], soff[jr], 0);     // wave-uniform branch
            }
        }
        float* const stg = reinterpret_cast<float*>(s_w + wp_first(tw) * 1024);      // this wave's own block of the weight region
        estamp();
#pragma unroll
        for (int jr = 0; jr < NB; ++jr) {
            if (jr == 1) estamp();
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(stg + l31 * EPI_PITCH + mb * 32 + 8 * q + 4 * hi) =
                        make_float4(acc[mb][jr][4 * q], acc[mb][jr][4 * q + 1], acc[mb][jr][4 * q + 2], acc[mb][jr][4 * q + 3]);
            // wave-local hand-off: the LDS operations of one wave complete in order
#pragma unroll
            for (int e = 0; e < EPI_ITERS; ++e) {
                const float* sp = stg + (lane_off[e] >> 1) - px_e[e] * (M.out_cstride - EPI_PITCH);     // px * PITCH + grp * 8
                const float4 lo = *reinterpret_cast<const float4*>(sp), hi4 = *reinterpret_cast<const float4*>(sp + 4);
                float v[8] = {lo.x, lo.y, lo.z, lo.w, hi4.x, hi4.y, hi4.z, hi4.w};
                if constexpr (FP8) {             // y = sum * (input scale x weight scale of the channel) + folded-BN shift
                    __builtin_amdgcn_sched_barrier(0);       // (hipcc hoists the table reads of all twelve items otherwise: 190 registers)
                    const int ch = tab + nb * TT_COUT + (int)(lane_off[e] >> 1) - px_e[e] * M.out_cstride;      // ... + grp * 8
                    const float4 s0 = *reinterpret_cast<const float4*>(s_osc + ch), s1 = *reinterpret_cast<const float4*>(s_osc + ch + 4);
                    const float4 b0 = *reinterpret_cast<const float4*>(s_bias + ch), b1 = *reinterpret_cast<const float4*>(s_bias + ch + 4);
                    const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, bi[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = v[k] * sc[k] + bi[k];
                }
                if (has_res) {
                    const bf16x8 r = __builtin_bit_cast(bf16x8, rr[jr][e]);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] += (float)r[k];
                }
                bf16x8 q;
#pragma unroll
                for (int k = 0; k < 8; ++k) q[k] = (__bf16)v[k];
                if (M.relu) {       // relu(round(x)) == round(relu(x)); a negative bf16 is a negative int16
                    typedef short s16x8 __attribute__((ext_vector_type(8)));
                    const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                    q = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, q), z));
                }
                if constexpr (FP8) {             // e4m3 twin for the next fp8 convolution: the ROUNDED bf16 value / scale, saturated.
                    // Packed BEFORE the bf16 store is issued: with the conversion after it, v_cvt_pk_fp8_f32 was allocated the
                    // store's first data register and, under memory load, overwrote it before the store had read it -- isolated
                    // wrong bf16 elements (first of an 8-channel group, magnitudes of fp8 codes), traced with per-launch checksums.
                    float w8[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) w8[k] = fminf(fmaxf((float)q[k] * M.out8_inv_scale, -448.f), 448.f);
                    int p0 = __builtin_amdgcn_cvt_pk_fp8_f32(w8[0], w8[1], 0, false);
                    p0 = __builtin_amdgcn_cvt_pk_fp8_f32(w8[2], w8[3], p0, true);
                    int p1 = __builtin_amdgcn_cvt_pk_fp8_f32(w8[4], w8[5], 0, false);
                    p1 = __builtin_amdgcn_cvt_pk_fp8_f32(w8[6], w8[7], p1, true);
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    const unsigned v8 = voff[jr][e] == 0x80000000u ? 0x80000000u : voff[jr][e] >> 1;
                    u32x4 qd = __builtin_bit_cast(u32x4, q);
                    u32x2 pd = u32x2{(unsigned)p0, (unsigned)p1};
                    asm volatile("" : "+v"(qd), "+v"(pd));      // both packs are complete, in registers of their own, before either store
                    __builtin_amdgcn_raw_buffer_store_b128(qd, rs_out, voff[jr][e], soff[jr], 0);
                    __builtin_amdgcn_raw_buffer_store_b64(pd, rs_out8, v8, soff[jr] >> 1, 0);
                    asm volatile("s_nop 1" ::: "memory");        // ... and two wait states before anything may reuse the data registers
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, q), rs_out, voff[jr][e], soff[jr], 0);
                }
            }
        }
        estamp();
    };

    auto multiply_stage = [&](auto&& near_end) {
        // pin the accumulators where they are: without this hipcc copies the 96 registers on entry (two reaching
        // definitions: the start value of a new item / the previous stage) and spills some of the originals around the phase
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int jr = 0; jr < NB; ++jr) asm volatile("" : "+v"(acc[mb][jr]));
        if constexpr (FP8) {
            // fp8 stage = 64 input channels: 9 K = 64 steps (one per tap) x 6 v_mfma_scale_f32_32x32x64_f8f6f4 with unit block
            // scales (E8M0 127).  Lane l holds 32 consecutive K bytes of row / column l & 31: channels 32 (l >> 5) .. + 31 of
            // the tap = two 16-byte slots of the pixel, two lane-linear 1 KB pieces of the weights (layout probed on hardware:
            // tools/dev/mx_probe.hip).
            i32x8 a[2][MB], b[2][NB];
            auto load_frags = [&](int s, int buf) {
                const int dy = s / 3, dx = s - dy * 3;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    const i32x4 lo = *reinterpret_cast<const i32x4*>(aptr + ((s * MB + mb) * 2 + 0) * 1024);
                    const i32x4 hi4 = *reinterpret_cast<const i32x4*>(aptr + ((s * MB + mb) * 2 + 1) * 1024);
                    a[buf][mb] = i32x8{lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                }
#pragma unroll
                for (int jr = 0; jr < NB; ++jr) {
                    const i32x4 lo = *reinterpret_cast<const i32x4*>(bptr[dx][(jr + dy) & 3] + (jr + dy) * HP * 64);
                    const i32x4 hi4 = *reinterpret_cast<const i32x4*>(bptr[dx][(jr + dy + 1) & 3] + (jr + dy) * HP * 64);
                    b[buf][jr] = i32x8{lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                }
            };
            load_frags(0, 0);
#pragma unroll
            for (int t = 0; t < NKS; ++t) {
                const int cur = t & 1;
                if (t + 1 < NKS) {
                    load_frags(t + 1, cur ^ 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int jr = 0; jr < NB; ++jr)
                        acc[mb][jr] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[cur][mb], b[cur][jr], acc[mb][jr], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
                // anchor the step: hipcc sinks the scaled MFMAs of ALL steps below the last reads otherwise (sched_barrier does
                // not hold them), keeps nine steps of fragments alive and spills ~290 registers
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int jr = 0; jr < NB; ++jr) asm volatile("" : "+v"(acc[mb][jr]) :: "memory");
                if (t + 1 < NKS) __builtin_amdgcn_sched_barrier(0);
            }
            near_end();
        } else if constexpr (X3) {
            // 9 taps x (10 fragment reads, 18 MFMAs): K-step h = 0 holds the hi parts, h = 1 the lo parts of the stage's 16 channels;
            // acc += w_lo.x_hi + w_hi.x_lo + w_hi.x_hi (small terms first).  Fragments of tap s + 1 are fetched while tap s multiplies.
            bf16x8 a[2][2][MB], b[2][2][NB];
            auto load_frags = [&](int sidx, int buf) {
                const int dy = sidx / 3, dx = sidx - dy * 3;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) a[buf][h][mb] = *reinterpret_cast<const bf16x8*>(aptr + ((2 * sidx + h) * MB + mb) * 1024);
#pragma unroll
                    for (int jr = 0; jr < NB; ++jr)
                        b[buf][h][jr] = *reinterpret_cast<const bf16x8*>(bptr[dx][(jr + dy + 2 * h) & 3] + (jr + dy) * HP * 64);
                }
            };
            load_frags(0, 0);
#pragma unroll
            for (int sidx = 0; sidx < NKS; ++sidx) {
                const int cur = sidx & 1;
                if (sidx == NKS - 1) near_end();
                if (sidx + 1 < NKS) load_frags(sidx + 1, cur ^ 1);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int jr = 0; jr < NB; ++jr) {
                        acc[mb][jr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][1][mb], b[cur][0][jr], acc[mb][jr], 0, 0, 0);
                        acc[mb][jr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][0][mb], b[cur][1][jr], acc[mb][jr], 0, 0, 0);
                        acc[mb][jr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][0][mb], b[cur][0][jr], acc[mb][jr], 0, 0, 0);
                    }
                if (sidx + 1 < NKS) {                              // one fragment read behind each of the first ten MFMAs of the tap
#pragma unroll
                    for (int i = 0; i < 2 * (MB + NB); ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 3 * MB * NB - 2 * (MB + NB), 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            // 18 K = 16 steps (tap, channel half) x 6 MFMAs of 32 x 32 x 16; fragments of step t + 1 are fetched while step t
            // multiplies.  `near_end` runs before the last two steps (the token is handed on while ~400 clk of MFMAs are queued).
            bf16x8 a[2][MB], b[2][NB];
            auto load_frags = [&](int t, int buf) {
                const int s = t >> 1, h = t & 1, dy = s / 3, dx = s - dy * 3;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) a[buf][mb] = *reinterpret_cast<const bf16x8*>(aptr + (t * MB + mb) * 1024);
#pragma unroll
                for (int jr = 0; jr < NB; ++jr)
                    b[buf][jr] = *reinterpret_cast<const bf16x8*>(bptr[dx][(jr + dy + 2 * h) & 3] + (jr + dy) * HP * 64);
            };
            load_frags(0, 0);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int cur = t & 1;
                if (t == NT - 2) near_end();
                if (t + 1 < NT) load_frags(t + 1, cur ^ 1);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                    for (int jr = 0; jr < NB; ++jr)
                        acc[mb][jr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][mb], b[cur][jr], acc[mb][jr], 0, 0, 0);
                if (t + 1 < NT) {                                  // one fragment read behind each of the first five MFMAs of the step
#pragma unroll
                    for (int i = 0; i < MB + NB; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, MB * NB - (MB + NB), 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    };

    // Both teams run the same program -- per stage: LOAD (DMA, and at an item boundary the previous tile's epilogue), team
    // barrier, take the CU's MULTIPLY token, 108 MFMAs per wave, release.  The token makes the MFMA phases of the two teams
    // mutually exclusive (two multiplying waves on one SIMD would only halve each other), so the teams alternate by
    // themselves: while one multiplies the other loads.  They are NOT locked phase by phase: the first version closed every
    // phase with one s_barrier over both teams, and a team whose LOAD phase carried an epilogue (~6k clk: 24 vector-memory
    // instructions per wave at ~160 clk each beside a multiplying partner) held the partner's next multiply back by 2-5k
    // clk per tile.  Now the partner takes the token again as soon as its own next stage has landed.
    // Team-level synchronisation goes through LDS words (one wave-instruction each way, s_sleep while polling):
    //   arrive += 1 per wave when its DMA pieces have landed            go    = k + 1 once wave 0 holds the token
    //   early  += 1 per wave two K-steps before the end of its MFMAs;   the last one frees the token
    //   done   += 1 per wave when its MFMAs (and LDS reads) are over:   the stage's buffers may be overwritten
    unsigned* const ctrl = reinterpret_cast<unsigned*>(smem + 2 * TEAM_BYTES);       // [0] token, [4 + 4 team + {0,1,2,3}] arrive, go, done, early
    unsigned* const w_token = ctrl;
    unsigned* const w_arrive = ctrl + 4 + 4 * team;
    unsigned* const w_go = w_arrive + 1;
    unsigned* const w_done = w_arrive + 2;
    unsigned* const w_early = w_arrive + 3;
    if (tid < 16) ctrl[tid] = 0u;
    for (int i = tid; i < 2 * TT_TABLE_MAX; i += 512) s_bias[i] = 0.f;      // (rows of a padded channel block read table slots nobody fills)
    __syncthreads();
#pragma unroll
    for (int m = 0; m < TT_MAX_MEMBERS; ++m) {
        const int o = m == 0 ? 0 : m == 1 ? tab1 : tab2;
        if (P.m[m].bias && tid < P.m[m].cout) {
            s_bias[o + tid] = P.m[m].bias[tid];
            if constexpr (FP8) s_osc[o + tid] = P.m[m].oscale[tid];
        }
    }
    __syncthreads();
    auto poll = [&](unsigned* p) -> unsigned { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto spin_until = [&](unsigned* p, unsigned target) {
        while ((int)(poll(p) - target) < 0) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
    };
    const unsigned S = team == 0 ? SA : SB;
    const unsigned it_last = max(P.team_first[T + 1], it + 1u) - 1u;
    TTItem I_next = P.items[it];
    for (unsigned st = 0; st < S; ++st) {
        stamp();                                                  // [0] LOAD begins
        if (c == 0) {
            if (st > 0) epilogue();                               // the finished tile, out of registers that the new one needs
            setup_item(I_next);
            setup_done_stamp();
            // the item after this one is fetched now (a scalar load that misses every cache: 2-3k clk if waited for on the spot)
            I_next = P.items[min(it + 1u, it_last)];
        }
        stamp();                                                  // [1] epilogue / setup done
        if (!(P.ablate & 4)) issue_stage(c);
        stamp();                                                  // [2] DMA issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // my DMA pieces have landed (and my stores are out)
        stamp();                                                  // [3] landed
        if (lane == 0) __hip_atomic_fetch_add(w_arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (tw == 0) {
            spin_until(w_arrive, 4u * (st + 1u));                 // the whole stage is in LDS
            for (;;) {                                            // take the token
                unsigned got = 0;
                if (lane == 0) {
                    unsigned expect = 0u;
                    got = __hip_atomic_compare_exchange_strong(w_token, &expect, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) ? 1u : 0u;
                }
                if (__builtin_amdgcn_readfirstlane(got)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (lane == 0) __hip_atomic_store(w_go, st + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            spin_until(w_go, st + 1u);
        }
        asm volatile("" ::: "memory");
        stamp();                                                  // [4] MULTIPLY begins
        if (P.ablate & 2) {
            unsigned old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(w_early, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (lane == 0 && old == 4u * st + 3u) __hip_atomic_store(w_token, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else
        multiply_stage([&]() {
            unsigned old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(w_early, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (lane == 0 && old == 4u * st + 3u) __hip_atomic_store(w_token, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        });
        if (++c == M.chunks) { c = 0; ++it; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp();                                                  // [5] MFMAs issued
        if (lane == 0) __hip_atomic_fetch_add(w_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        spin_until(w_done, 4u * (st + 1u));                       // every wave of the team is done reading this stage
    }
    if (S > 0) { epilogue(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
}

void launch_conv_tt(const TTParams& p, int n_wgs, int mode, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_tt_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_tt_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_tt_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    const size_t lds = (size_t)2 * TEAM_BYTES + 64 + 2 * TT_TABLE_MAX * 4;
    if (mode == 1) SNCAL_LAUNCH(conv_tt_kernel<1>, dim3((unsigned)n_wgs), dim3(512), lds, s, p);
    else if (mode == 2) SNCAL_LAUNCH(conv_tt_kernel<2>, dim3((unsigned)n_wgs), dim3(512), lds, s, p);
    else SNCAL_LAUNCH(conv_tt_kernel<0>, dim3((unsigned)n_wgs), dim3(512), lds, s, p);
}

}  // namespace sncal
