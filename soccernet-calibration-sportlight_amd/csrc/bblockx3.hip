// Fused BasicBlock of the 48-channel full-resolution branch in SPLIT arithmetic (the fp32-class engine `bf16x3`):
//     out = ReLU( BN2(conv2( ReLU(BN1(conv1(x))) )) + x )          /root/reference/src/models/hrnet/hrnet.py:42-58
// in ONE persistent kernel, every product as w_hi.x_hi + w_hi.x_lo + w_lo.x_hi on the 16-bit matrix pipe with fp32 accumulation.
//
// Why (round 4).  As two launches of the two-team kernel (conv_tt, 64-channel x 12-row tile) the block cost 2 x 357 us per 64 frames at
// 0.27 of the split-arithmetic roof: 48 channels ran as a padded 64-row MFMA block (25 % zero rows), every tile of three 16-channel
// stages paid a tile boundary as long as its multiplies (phase traces: epilogue 10k clk, 24k with the residual reads; the x halo of every
// stage fetched from HBM by a team that cannot load while it multiplies), and the pair moved 5.5 tensor passes through HBM (x halo, mid
// out, mid halo, residual, out) where 2.3 are needed.  Here:
//   * v_mfma_f32_16x16x32: three 16-row blocks = 48 output channels exactly, no padded rows;
//   * the split form costs 1.5 MFMAs per (16 x 16 outputs, 16 channels of one tap) instead of 2: the two CROSS terms of a unit share
//     one K = 32 instruction (A = [w_hi | w_lo] against B = [x_lo | x_hi]) and the MAIN terms of two units share another
//     (A = [w_hi(u0) | w_hi(u1)], B = [x_hi(u0) | x_hi(u1)]); units are paired so that the second unit's pixels sit at one of three
//     constant distances from the first's (bblockx3.hpp);
//   * x halo (20 x 18 pixels) and the mid tile (18 x 16) live in LDS as separate hi and lo PLANES of 96 bytes per pixel: the K octets
//     of a B fragment are then adjacent 16-byte slots at a pixel pitch of 6 slots, which is conflict-free for ds_read_b128's lane groups
//     ({ln 0-3,12-15 | octet 0} + {ln 4-11 | octet 1} hit the even / the odd slots) without padding; the twin's [16 hi | 16 lo] groups
//     are de-interleaved by the DMA's per-lane source address;
//   * weights (2 x 84 KB, more than the LDS left beside the tiles) stream through a 4-slot ring of 6 KB pair-steps, requested by a NINTH
//     wave, the x halos by a TENTH, that do nothing else: no multiplying wave ever issues a DMA or waits on vmcnt, and the halo's HBM
//     latency never stands in front of a weight step (requests of one wave complete in order); the hand-offs are counters in LDS
//     (w_ready / w_done per step, x_ready / x_free per tile), the eight multiplying waves meet only at the mid tile;
//   * the residual is the centre of the x halo (registers, read before the halo region is handed back), the next tile's halo lands
//     under conv2, the output is stored straight from the accumulators as the split twin the next block reads and / or as fp32.
//   * (round 5) the frames are tiled as ONE stack of N (H + 1) rows and a workgroup walks RUNS of up to eight tiles down a 14-column
//     strip, keeping the four x rows and two mid rows a tile shares with the one above it in LDS (kernel body: "Geometry"): 576 -> 544
//     tile rows per 64 frames of 135 rows, conv1 on 16 instead of 18 mid rows for three tiles in four, 20 % fewer halo requests;
//     632 -> 595 us per block on the same box, outputs bit for bit the same.  What the phase trace shows is left: the multiply loops run
//     at 257 clk per sub-step (12 MFMAs per SIMD = 192 clk nominal) with every DMA switched off (SNCAL_BBX_DBG=6: 27.1k against 28.5k clk
//     per tile) -- 40 ds_read_b128 per sub-step and CU are 160 of those clocks on the LDS port; 3 x J register blocking at J = 2.
// Arithmetic: hi = rne16(v), lo = rne16(v - hi); mid is split exactly like a tensor that conv_tt would have written, so the block
// equals conv1 -> twin -> conv2 of the two-kernel path up to the summation order inside the matrix unit.
#include "bblockx3.hpp"
#include "common.hpp"
#include "x3.hpp"
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace sncal {
namespace {

typedef x3h8 h16x8;                                                   // (x3.hpp: the 16-bit type of the split, fp16 or bf16)
typedef x3h4 h16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));
#define BBX_MFMA(a, b, c) X3_MFMA_16x16x32(a, b, c)

constexpr int TH = 16, TW = 14;                     // output tile
constexpr int MH = TH + 2, MW = 16;                 // mid tile (rows, fragment width)
constexpr int XH = TH + 4, XW = 18;                 // x halo tile
constexpr int PXB = 96;                             // bytes per pixel and plane (48 channels x 16 bit)
constexpr int X_PITCH = XW * PXB, X_PIECES = 34, X_PLANE = X_PIECES * 1024;     // 34,560 B of pixels per plane in whole 1 KB DMA pieces
constexpr int M_PITCH = MW * PXB, M_PLANE = MH * M_PITCH;
constexpr int RING_SLOTS = 4;                       // 2 x BBX_STEPS is a multiple: a step's slot is a compile-time constant
constexpr int OFF_X = 0, OFF_M = OFF_X + 2 * X_PLANE, OFF_RING = OFF_M + 2 * M_PLANE;
constexpr int OFF_BIAS = OFF_RING + RING_SLOTS * BBX_STEP_BYTES, OFF_CTRL = OFF_BIAS + 96 * 4, LDS_BYTES = OFF_CTRL + 96;
#ifndef BBX_NW
#define BBX_NW 8
#endif
constexpr int NW = BBX_NW;                          // multiplying waves (8: two per SIMD; 4: one per SIMD, twice the rows each); wave NW streams the weights, wave NW + 1 the x halos
constexpr int J1 = (MH + NW - 1) / NW, J2 = TH / NW;                          // pixel fragments (tile rows) per wave: conv1 (at most), conv2
constexpr int NSUB = 3 * BBX_STEPS - 1;             // sub-steps of a convolution: (cross u0, cross u1, main) per step, no cross for the zero unit
static_assert(XH * X_PITCH <= X_PLANE && (2 * BBX_STEPS) % RING_SLOTS == 0 && TH % NW == 0 && J1 == J2 + 1 && LDS_BYTES <= 160 * 1024, "layout");
enum { C_WREADY = 0, C_XREADY = 2, C_XFREE = 3, C_MID = 4, C_TKNOWN = 5, C_MCOPY = 6, C_WDONE = 8, C_TILE = 16 };       // C_TILE + (round & 3): the round's tile, or TILE_END
constexpr unsigned TILE_END = 0xffffffffu;          // C_WDONE + w: steps wave w has finished reading (a SUM over waves
                                                                                       // would let seven fast waves vouch for a slow one)
constexpr unsigned TILE_CONT = 0x40000000u;         // flag in a tile word: the tile directly BELOW the workgroup's previous one (see "runs")
// A continuing tile keeps what it shares with the tile above: x halo rows 16..19 become rows 0..3 (LDS slots 0..383 of either plane, whole
// 1 KB pieces; the piece that straddles rows 3 | 4 is simply fetched again) and mid rows 16, 17 become rows 0, 1.
constexpr int KEEP_XROWS = 4, KEEP_PIECES = KEEP_XROWS * XW * 6 / 64, KEEP_SRC = (XH - KEEP_XROWS) * X_PITCH;
constexpr unsigned OOB = 0xfffffe00u;               // a buffer offset beyond any tensor the launcher accepts (reads return zero, stores are dropped)
constexpr int RUN_MAX = 8;                          // longest run of vertically adjacent tiles a workgroup takes with one ticket
static_assert(KEEP_PIECES == 6 && KEEP_SRC + KEEP_PIECES * 1024 <= XH * X_PITCH && KEEP_PIECES * 1024 <= KEEP_XROWS * X_PITCH, "kept halo rows");

// One LDS-DMA piece (64 lanes x 16 bytes -> 1 KB at LDS byte address `lds_addr`) as inline assembly: hipcc's wait-count pass does not
// see it, so the loader's counter polls are not preceded by vmcnt(0); every wait of this kernel is explicit.
__device__ __forceinline__ void dma_piece(i32x4_t rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ i32x4_t raw_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    return i32x4_t{(int)(unsigned)a, (int)(unsigned)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ unsigned poll(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void spin_until(unsigned* p, unsigned target) {
    while ((int)(poll(p) - target) < 0) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

__global__ __launch_bounds__(64 * (NW + 2)) void bblockx3_kernel(const BBlockX3Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned* const ctrl = reinterpret_cast<unsigned*>(smem + OFF_CTRL);
    float* const s_bias = reinterpret_cast<float*>(smem + OFF_BIAS);
    if (tid < 24) ctrl[tid] = 0u;
    if (tid < 96) s_bias[tid] = tid < 48 ? p.b1[tid] : p.b2[tid - 48];
    __syncthreads();

    // Geometry.  The frames are walked as ONE image of N (H + 1) rows -- frame n at rows n (H + 1) .., one separator row between frames that
    // reads as zeros and is never stored: with per-frame tiles a 135-row frame cost 9 x 16 = 144 tile rows (6.25 % of the tiles' rows empty),
    // stacked it costs 136.  Tile (ys, tx) covers stacked rows 16 ys .. + 15 and columns 14 tx .. + 13.
    // Tiles of this workgroup: the tile ROWS are cut into 8 contiguous ranges, one per XCD (workgroup b runs on XCD b % 8); the workgroups of
    // an XCD draw RUNS from the XCD's ticket counter (p.ticket[xcd]): a run = up to RUN_MAX vertically adjacent tiles of one 14-column strip,
    // walked downwards.  From the second tile of a run on, conv1 computes 16 mid rows instead of 18 (two per wave: the 18-row tile gives two
    // waves three rows and the SIMDs that host them set conv1's pace, 630 against 504 MFMAs) and the halo wave fetches 28 pieces per plane
    // instead of 34.  Runs are dealt row-block-major, strip-minor, so the workgroups of an XCD walk neighbouring strips side by side and
    // the column overlap of their halos meets in the XCD's L2.  Long runs leave a long tail: the run length halves (8, 4, 2, 1) towards the
    // end of the XCD's range, each length while at least one run per workgroup of it is left ("guided" deal); launches with less than a
    // handful of tiles per workgroup come out as single tiles.  A workgroup whose CU another stream's work holds for a while (the camera
    // solves of the previous batch) simply takes fewer runs (round 5; the static deal of rounds 3 / 4 made the launch wait for it: +4 %).
    // The halo wave takes the ticket (it is a round ahead of everybody) and publishes the tiles through LDS; the last workgroup to run dry
    // re-arms the counters.
    const int xcd = (int)blockIdx.x & 7;
    const int h1 = p.H + 1, rows_s = p.N * h1;                     // stacked rows (the last separator is never reached by a valid pixel)
    const int tiles_ys = (rows_s - 1 + TH - 1) / TH;
    const int y_lo = (int)((long)tiles_ys * xcd / 8), y_hi = (int)((long)tiles_ys * (xcd + 1) / 8);
    const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) char*)smem;
    const unsigned img_twin = (unsigned)(p.H * p.W * 192);
    // stacked row v -> (frame, row in frame): v / h1 by multiplication (exact for v h1 < 2^32: the launcher bounds N)
    const unsigned h1_magic = p.h1_magic;
    auto frame_of = [&](unsigned v) -> unsigned { return __umulhi(v, h1_magic); };

    auto publish = [&](int word, unsigned v) { if (lane == 0) __hip_atomic_store(ctrl + word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    if (wave == NW + 1) {
        // ---------------------------------------------------------------- the halo wave ---------------------------------------------
        // One tile's x halo per round: 34 + 34 requests that come from HBM.  It has a wave (= a request queue) of its own: requests of one
        // wave complete in order, and riding with the weight stream -- all at once, or a few per step -- the halo's HBM latency stood in
        // front of every weight step that followed it (conv2: 8k clk of MFMAs took 18k / 23k clk, phase trace).
        // small launches (fewer than 8 tile rows per XCD: up to batch 7 at 540p): whole tile rows per XCD are too coarse a cut (batch 1: nine
        // tile rows for eight XCDs, one of them with two rounds of tiles) -- the row-major tile list is cut into eight equal shares instead
        const bool flat = tiles_ys < 64;
        const int n_tiles = tiles_ys * p.tiles_x;
        const int t_lo = (int)((long)n_tiles * xcd / 8), t_hi = (int)((long)n_tiles * (xcd + 1) / 8);
        // phases of the XCD's deal: rows_of[k] tile rows in runs of RUN_MAX >> k tiles
        int rows_of[4];
        {
            const int wgs = ((int)gridDim.x - xcd + 7) / 8;
            int rem = y_hi - y_lo;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int L = RUN_MAX >> k, keep = (wgs * L + p.tiles_x - 1) / p.tiles_x;
                rows_of[k] = p.run_max >= L && rem > keep ? (rem - keep) / L * L : 0;
                rem -= rows_of[k];
            }
            rows_of[3] = rem;
        }
        const i32x4_t rs = raw_rsrc(p.x, (unsigned)p.N * img_twin);
        int k = 0, len = 0, row0 = 0, strip = 0;
        unsigned n_cont = 0;
        bool prev_cont = false;
        for (int ti = 0;; ++ti) {
            if (k == len) {                                                               // the next run
                unsigned tk = 0;
                if (lane == 0) tk = atomicAdd(p.ticket + xcd, 1u);
                int r = __builtin_amdgcn_readfirstlane((int)tk);
                row0 = y_lo; len = 0; k = 0;
                if (flat) {                                                               // single tiles of the XCD's share of the row-major tile list
                    r += t_lo;
                    if (r < t_hi) { len = 1; row0 = r / p.tiles_x; }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int L = RUN_MAX >> q, runs = rows_of[q] / L * p.tiles_x;
                        if (len == 0) {
                            if (r < runs) { len = L; row0 += r / p.tiles_x * L; }
                            else { r -= runs; row0 += rows_of[q]; }
                        }
                    }
                }
                strip = r % p.tiles_x;
            }
            const bool dry = len == 0;
            // (slot ti & 3 last held round ti - 4; every reader is in round ti - 1 or later: the XFREE wait below)
            publish(C_TILE + (ti & 3), dry ? TILE_END : (unsigned)((row0 + k) * p.tiles_x + strip) | (k > 0 ? TILE_CONT : 0u));
            publish(C_TKNOWN, (unsigned)ti + 1u);
            if (prev_cont) {
                // tile ti - 1 continues tile ti - 2: the upper tile's last two mid rows are the lower tile's first two.  Moved HERE, while the
                // multiplying waves are in conv1 of tile ti - 1 (they moved them themselves at first: +1k clk on waves 6 / 7 in front of the mid
                // barrier): every wave is done with conv2 of tile ti - 2, and nobody writes or reads mid before C_MCOPY says so
                const unsigned need = 2u * BBX_STEPS * (unsigned)(ti - 1);
                while (__builtin_amdgcn_ballot_w64((int)(poll(ctrl + C_WDONE + (lane & (NW - 1))) - need) >= 0) != ~0ull) __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
                static_assert(2 * M_PITCH == 3 * 1024, "two mid rows are 3 KB per plane");
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    u32x4 part[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) part[i] = *reinterpret_cast<const u32x4*>(smem + OFF_M + pl * M_PLANE + (MH - 2) * M_PITCH + i * 1024 + lane * 16);
#pragma unroll
                    for (int i = 0; i < 3; ++i) *reinterpret_cast<u32x4*>(smem + OFF_M + pl * M_PLANE + i * 1024 + lane * 16) = part[i];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                publish(C_MCOPY, ++n_cont);
                prev_cont = false;
            }
            if (dry) {
                if (lane == 0 && atomicAdd(p.ticket + 8, 1u) == gridDim.x - 1u) {     // every workgroup holds its one failing ticket
#pragma unroll
                    for (int i = 0; i < 9; ++i) p.ticket[i] = 0u;
                    __threadfence();
                }
                break;
            }
            const int oy0 = (row0 + k) * TH - 2, ox0 = strip * TW - 2;
            if (ti > 0) spin_until(ctrl + C_XFREE, (unsigned)NW * (unsigned)ti);          // everyone has read its residual out of the old halo
            int piece0 = 0;
            if (k > 0) {                                                                  // the four rows shared with the tile above stay in LDS
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {                                          // (a plane at a time: 24 registers)
                    u32x4 keep[KEEP_PIECES];
#pragma unroll
                    for (int i = 0; i < KEEP_PIECES; ++i)
                        keep[i] = *reinterpret_cast<const u32x4*>(smem + OFF_X + pl * X_PLANE + KEEP_SRC + i * 1024 + lane * 16);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   // (read before the requests below overwrite rows 16..19)
#pragma unroll
                    for (int i = 0; i < KEEP_PIECES; ++i)
                        *reinterpret_cast<u32x4*>(smem + OFF_X + pl * X_PLANE + i * 1024 + lane * 16) = keep[i];
                }
                piece0 = KEEP_PIECES;
            }
            for (int piece = piece0; piece < X_PIECES; ++piece) {
                // LDS slot q of a plane = pixel q / 6 of the 20 x 18 halo (row-major), 16-byte slot q % 6 = (group, half) of its 48 channels;
                // the twin keeps [hi half 0 | hi half 1 | lo half 0 | lo half 1] per group: hi plane from +0 / +16, lo plane from +32 / +48
                const unsigned q = (unsigned)(piece * 64 + lane);
                const unsigned px = (q * 43691u) >> 18, sl = q - px * 6u;                  // q / 6 for q < 2^16
                const unsigned row = (px * 3641u) >> 16, col = px - row * 18u;            // px / 18 for px < 2^12
                const int vs = oy0 + (int)row, ix = ox0 + (int)col;
                const unsigned n = frame_of((unsigned)(vs < 0 ? 0 : vs));
                const int iy = vs - (int)n * h1;
                const bool ok = (px < (unsigned)(XH * XW)) & (vs >= 0) & (n < (unsigned)p.N) & (iy < p.H) & ((unsigned)ix < (unsigned)p.W);
                const unsigned voff = ok ? n * img_twin + (unsigned)((iy * p.W + ix) * 192) + (sl >> 1) * 64u + (sl & 1u) * 16u : OOB;
                if (p.dbg & 4) continue;
                dma_piece(rs, lds0 + OFF_X + piece * 1024, voff, 0u);
                dma_piece(rs, lds0 + OFF_X + X_PLANE + piece * 1024, voff, 32u);
            }
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            publish(C_XREADY, (unsigned)ti + 1u);
            prev_cont = k > 0;
            ++k;
        }
        return;
    }
    if (wave == NW) {
        // ---------------------------------------------------------------- the weight wave -------------------------------------------
        // Every request of this wave completes in order.  After the six pieces of step g are issued, `vmcnt(12)` leaves at most the
        // pieces of steps g - 1 and g in flight: steps <= g - 2 have landed.
        const i32x4_t rs_w1 = raw_rsrc(p.w1, BBX_W_BYTES), rs_w2 = raw_rsrc(p.w2, BBX_W_BYTES);
        unsigned g = 0;
        for (int ti = 0;; ++ti) {
            spin_until(ctrl + C_TKNOWN, (unsigned)ti + 1u);          // (known a round ahead: the halo wave takes round ti + 1's ticket as round ti begins)
            if (poll(ctrl + C_TILE + (ti & 3)) == TILE_END) break;
            for (int s = 0; s < 2 * BBX_STEPS; ++s, ++g) {
                if (g >= (unsigned)RING_SLOTS) {                                        // every wave is done with the slot's previous step
                    const unsigned need = g - RING_SLOTS + 1u;
                    while (__builtin_amdgcn_ballot_w64((int)(poll(ctrl + C_WDONE + (lane & (NW - 1))) - need) >= 0) != ~0ull) __builtin_amdgcn_s_sleep(1);
                    asm volatile("" ::: "memory");
                }
                const unsigned slot = g & (RING_SLOTS - 1);
                const i32x4_t rs = s < BBX_STEPS ? rs_w1 : rs_w2;
                const unsigned src = (unsigned)((s < BBX_STEPS ? s : s - BBX_STEPS) * BBX_STEP_BYTES);
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    if (!(p.dbg & 2) || g < 4u) dma_piece(rs, lds0 + OFF_RING + slot * BBX_STEP_BYTES + i * 1024, (unsigned)(lane * 16), src + i * 1024);
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                if (g >= 1u) publish(C_WREADY, g - 1u);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        publish(C_WREADY, g);
        return;
    }

    // -------------------------------------------------------------------- the multiplying waves -----------------------------------
    const int ln = lane & 15, o = lane >> 4, ln2 = ln < TW ? ln : TW - 1;         // conv2: lanes 14 / 15 recompute column 13 and are not stored
    const unsigned lane_in = (unsigned)((o & 1) * 16);
    // per-lane LDS byte offsets of the B fragments: [row j] cross-term fragment ([x_lo | x_hi]: octets 0, 1 from the lo plane) and the four
    // main-term variants (octets 2, 3 = the partner unit: same pixel / next row / next pixel / next group)
    unsigned bc1[J1], bh1[J1][4], bc2[J2], bh2[J2][4];
#pragma unroll
    for (int j = 0; j < J1; ++j) {
        const int f = wave + NW * j < MH ? wave + NW * j : MH - 1;
        const unsigned h0 = (unsigned)(OFF_X + f * X_PITCH + ln * PXB) + lane_in;
        bc1[j] = h0 + (o < 2 ? (unsigned)X_PLANE : 0u);
        bh1[j][0] = h0; bh1[j][1] = h0 + (o >= 2 ? (unsigned)X_PITCH : 0u); bh1[j][2] = h0 + (o >= 2 ? (unsigned)PXB : 0u); bh1[j][3] = h0 + (o >= 2 ? 32u : 0u);
    }
#pragma unroll
    for (int j = 0; j < J2; ++j) {
        const int r = wave + NW * j;
        const unsigned h0 = (unsigned)(OFF_M + r * M_PITCH + ln2 * PXB) + lane_in;
        bc2[j] = h0 + (o < 2 ? (unsigned)M_PLANE : 0u);
        bh2[j][0] = h0; bh2[j][1] = h0 + (o >= 2 ? (unsigned)M_PITCH : 0u); bh2[j][2] = h0 + (o >= 2 ? (unsigned)PXB : 0u); bh2[j][3] = h0 + (o >= 2 ? 32u : 0u);
    }
    const unsigned pa = (unsigned)(OFF_RING + lane * 16), pah = pa + (o >= 2 ? 2560u : 0u);     // A: a unit's fragment / [w_hi(u0) | w_hi(u1)]

    f32x4 acc[3][J1];
    auto init_acc = [&](int conv) {
#pragma unroll
        for (int cb = 0; cb < 3; ++cb) {
            const float4 b = *reinterpret_cast<const float4*>(s_bias + conv * 48 + cb * 16 + o * 4);
#pragma unroll
            for (int j = 0; j < J1; ++j) acc[cb][j] = f32x4{b.x, b.y, b.z, b.w};
        }
    };
    auto signal = [&](int word) { if (lane == 0) __hip_atomic_fetch_add(ctrl + word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };

    // One convolution: 14 pair-steps of (cross u0, cross u1, main) sub-steps; J pixel fragments of this wave against the three 16-channel
    // blocks.  Operand fragments are read one sub-step ahead, the reads interleaved with the MFMAs.  gs0 = global index of its first step.
    auto conv = [&](auto j_c, auto pitch_c, auto s0_c, const unsigned* bc, const unsigned (*bh)[4], unsigned gs0) __attribute__((always_inline)) {
        constexpr int J = decltype(j_c)::value, PITCH = decltype(pitch_c)::value, S0 = decltype(s0_c)::value;
        h16x8 a[2][3], b[2][J];
        unsigned peek = 0;
        auto load = [&](int k, int buf) __attribute__((always_inline)) {
            const int s = k < 39 ? k / 3 : 13, part = k < 39 ? k % 3 : (k == 39 ? 0 : 2);
            const int slot = (S0 + s) & (RING_SLOTS - 1);
            // the step's weights have landed?  Asked one sub-step EARLIER (the poll rides behind the previous step's last fragment reads
            // and is back when they are): a poll on the spot drains this wave's LDS queue, a bubble in front of 28 steps per tile
            if (part == 0 && (s == 0 || (int)(peek - (gs0 + (unsigned)s + 1u)) < 0)) spin_until(ctrl + C_WREADY, gs0 + (unsigned)s + 1u);
            const BbxUnit u0 = bbx_unit(s, 0);
            if (part < 2) {
                const BbxUnit u = bbx_unit(s, part);
                const int off = u.dy * PITCH + u.dx * PXB + u.g * 32;
#pragma unroll
                for (int cb = 0; cb < 3; ++cb) a[buf][cb] = *reinterpret_cast<const h16x8*>(smem + pa + (slot * BBX_STEP_BYTES + part * 3072 + cb * 1024));
#pragma unroll
                for (int j = 0; j < J; ++j) b[buf][j] = *reinterpret_cast<const h16x8*>(smem + bc[j] + off);
            } else {
                const int ty = s < 9 ? 1 : s < 12 ? 2 : s == 12 ? 3 : 0;
                const int off = u0.dy * PITCH + u0.dx * PXB + u0.g * 32;
#pragma unroll
                for (int cb = 0; cb < 3; ++cb) a[buf][cb] = *reinterpret_cast<const h16x8*>(smem + pah + (slot * BBX_STEP_BYTES + cb * 1024));
#pragma unroll
                for (int j = 0; j < J; ++j) b[buf][j] = *reinterpret_cast<const h16x8*>(smem + bh[j][ty] + off);
                // (LDS operations of a wave execute in order: the step's last reads are ahead of this)
                if (lane == 0) __hip_atomic_store(ctrl + C_WDONE + wave, gs0 + (unsigned)s + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                peek = poll(ctrl + C_WREADY);
            }
        };
        load(0, 0);
#pragma unroll
        for (int k = 0; k < NSUB; ++k) {
            const int cur = k & 1;
            if (k + 1 < NSUB) load(k + 1, cur ^ 1);
#pragma unroll
            for (int cb = 0; cb < 3; ++cb)
#pragma unroll
                for (int j = 0; j < J; ++j) acc[cb][j] = BBX_MFMA(a[cur][cb], b[cur][j], acc[cb][j]);
            if (k + 1 < NSUB) {
#pragma unroll
                for (int i = 0; i < 3 + J; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one LDS read
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 3 * J - (3 + J), 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    const bool tracing = p.trace != nullptr;
    float amax = 0.f;                                  // range tracker of the mid and output splits (x3.hpp)
    auto lap = [&](int k) { if (tracing) { const unsigned long long now = __builtin_amdgcn_s_memtime(); tsum[k] += now - tprev; tprev = now; } };

    bool at_cont = false;
    unsigned n_cont = 0;                               // continuing tiles so far
    for (int ti = 0;; ++ti) {
        spin_until(ctrl + C_TKNOWN, (unsigned)ti + 1u);
        const unsigned tu = (unsigned)__builtin_amdgcn_readfirstlane((int)poll(ctrl + C_TILE + (ti & 3)));
        if (tu == TILE_END) break;
        const bool cont = (tu & TILE_CONT) != 0u;
        const int t = (int)(tu & ~TILE_CONT);
        const int tx = t % p.tiles_x, ys = t / p.tiles_x;
        const int oy0 = ys * TH, ox0 = tx * TW;                       // (oy0: stacked row)
        const unsigned gs = (unsigned)ti * 2u * BBX_STEPS;
        if (tracing) tprev = __builtin_amdgcn_s_memtime();
        init_acc(0);
        spin_until(ctrl + C_XREADY, (unsigned)ti + 1u);                   // this tile's halo has landed
        lap(0);
        // conv1.  First tile of a run: mid rows f = wave + 8 j of all 18 (waves 0 and 1 own three, the others two); continuing tile: rows
        // 0, 1 are the tile above's rows 16, 17 (copied below), f = 2 + wave + 8 j of the 16 new ones
        const int f0 = cont ? 2 : 0;
        if (cont != at_cont) {                                        // the conv1 fragment addresses follow f0 (a run's second tile, a new run's first)
            const unsigned d = cont ? 2u * X_PITCH : 0u - 2u * X_PITCH;
#pragma unroll
            for (int j = 0; j < J1; ++j) {
                bc1[j] += d;
#pragma unroll
                for (int i = 0; i < 4; ++i) bh1[j][i] += d;
            }
            at_cont = cont;
        }
        if (!cont && wave + NW * (J1 - 1) < MH)
            conv(std::integral_constant<int, J1>{}, std::integral_constant<int, X_PITCH>{}, std::integral_constant<int, 0>{}, bc1, bh1, gs);
        else
            conv(std::integral_constant<int, J1 - 1>{}, std::integral_constant<int, X_PITCH>{}, std::integral_constant<int, 0>{}, bc1, bh1, gs);
        lap(1);
        // residual = centre of the x halo -> registers (hi + lo), then the halo region is free for the next tile
        f32x4 res[J2][3];
#pragma unroll
        for (int j = 0; j < J2; ++j)
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) {
                const char* px = smem + OFF_X + ((wave + NW * j + 2) * XW + ln + 2) * PXB + cb * 32 + o * 8;
                const h16x4 rh = *reinterpret_cast<const h16x4*>(px), rl = *reinterpret_cast<const h16x4*>(px + X_PLANE);
#pragma unroll
                for (int e = 0; e < 4; ++e) res[j][cb][e] = (float)rl[e] + (float)rh[e];
            }
        signal(C_XFREE);
        // continuing tile: mid rows 0, 1 are the tile above's rows 16, 17, moved by the halo wave during conv1 (long done: the poll is a formality)
        if (cont) spin_until(ctrl + C_MCOPY, ++n_cont);
        // mid = ReLU(conv1) as hi / lo planes; positions outside the image (separator rows between frames included) are conv2's zero padding
#pragma unroll
        for (int j = 0; j < J1; ++j) {
            const int f = f0 + wave + NW * j;
            if (f < MH) {
                const int vs = oy0 - 1 + f, ix = ox0 - 1 + ln;
                const unsigned n = frame_of((unsigned)(vs < 0 ? 0 : vs));
                const int iy = vs - (int)n * h1;
                const bool inimg = (vs >= 0) & (n < (unsigned)p.N) & (iy < p.H) & ((unsigned)ix < (unsigned)p.W);
#pragma unroll
                for (int cb = 0; cb < 3; ++cb) {
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = inimg ? acc[cb][j][e] : 0.f;
                    x3u2 hi, lo;
                    x3_split4(v, x3_lower(true), hi, lo, amax);        // ReLU = the split's lower clamp bound
                    char* dst = smem + OFF_M + (f * MW + ln) * PXB + cb * 32 + o * 8;
                    *reinterpret_cast<x3u2*>(dst) = hi;
                    *reinterpret_cast<x3u2*>(dst + M_PLANE) = lo;
                }
            }
        }
        signal(C_MID);
        init_acc(1);
        lap(2);
        spin_until(ctrl + C_MID, (unsigned)NW * ((unsigned)ti + 1u));     // everyone's mid rows are written
        lap(3);
        conv(std::integral_constant<int, J2>{}, std::integral_constant<int, M_PITCH>{}, std::integral_constant<int, BBX_STEPS % RING_SLOTS>{}, bc2, bh2, gs + BBX_STEPS);
        lap(4);
        // epilogue: + x, ReLU -> split twin (8 + 8 bytes per lane and block: the four lanes of a pixel complete 32-byte halves) and / or fp32
        const bool has_tw = p.out_twin != nullptr, has_f = p.out != nullptr;
        const unsigned img_f32 = (unsigned)(p.H * p.W * p.out_cstride * 4);
        const __amdgpu_buffer_rsrc_t rs_tw = __builtin_amdgcn_make_buffer_rsrc(has_tw ? p.out_twin : const_cast<void*>(p.x), 0,
                                                                                 has_tw && !(p.dbg & 1) ? (int)((unsigned)p.N * img_twin) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_f = __builtin_amdgcn_make_buffer_rsrc(has_f ? static_cast<void*>(p.out) : const_cast<void*>(p.x), 0,
                                                                                has_f && !(p.dbg & 1) ? (int)((unsigned)p.N * img_f32) : 0, 0x00020000);
#pragma unroll
        for (int j = 0; j < J2; ++j) {
            const int vs = oy0 + wave + NW * j, ox = ox0 + ln;
            const unsigned n = frame_of((unsigned)vs);
            const int oy = vs - (int)n * h1;
            const bool ok = (ln < TW) & (n < (unsigned)p.N) & (oy < p.H) & (ox < p.W);
            const unsigned vt = ok ? n * img_twin + (unsigned)((oy * p.W + ox) * 192 + o * 8) : OOB;
            const unsigned vf = ok ? n * img_f32 + (unsigned)(((oy * p.W + ox) * p.out_cstride + p.out_coff + o * 4) * 4) : OOB;
#pragma unroll
            for (int cb = 0; cb < 3; ++cb) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[cb][j][e] + res[j][cb][e], 0.f);
                if (has_tw) {
                    x3u2 d0, d1;
                    x3_split4(v, x3_lower(false), d0, d1, amax);     // (v is past the ReLU: the fp32 output needs it too)
                    asm volatile("" : "+v"(d0), "+v"(d1));
                    __builtin_amdgcn_raw_buffer_store_b64(d0, rs_tw, vt, cb * 64, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(d1, rs_tw, vt, cb * 64 + 32, 0);
                    asm volatile("s_nop 3" :: "v"(d0), "v"(d1) : "memory");
                }
                if (has_f) {
                    u32x4 d = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                    asm volatile("" : "+v"(d));
                    __builtin_amdgcn_raw_buffer_store_b128(d, rs_f, vf, cb * 64, 0);
                    asm volatile("s_nop 3" :: "v"(d) : "memory");
                }
            }
        }
        lap(5);
        tsum[7] += 1;
    }
    x3_report(amax, p.range);
    if (tracing && lane == 0)
        for (int k = 0; k < 8; ++k) p.trace[((size_t)blockIdx.x * NW + wave) * 8 + k] = tsum[k];
}

}  // namespace

int launch_bblockx3(const BBlockX3Params& p0, hipStream_t s) {
    BBlockX3Params p = p0;
    // one launch addresses its tensors with 32-bit buffer offsets: more frames than fit go in several launches
    const size_t img = (size_t)p.H * p.W * (size_t)(p.out && p.out_cstride * 4 > 192 ? p.out_cstride * 4 : 192);
    const char* lim_env = getenv("SNCAL_BBX_MAX_BYTES");             // (test hook: a small limit sends a small batch down the several-launches path)
    const size_t lim = lim_env ? (size_t)atoll(lim_env) : (size_t)0xf0000000u;
    const size_t n_max = img ? lim / img : 0;
    if (n_max == 0 || (size_t)(p.H + 1) * (p.H + 1) * (n_max < (size_t)p.N ? n_max : (size_t)p.N) >= ((size_t)1 << 32)) {
        set_error("launch_bblockx3: image too large for 32-bit offsets");
        return SNCAL_ERR_ARG;
    }
    if ((size_t)p.N > n_max) {
        for (int n0 = 0; n0 < p0.N; n0 += (int)n_max) {
            BBlockX3Params q = p0;
            q.N = p0.N - n0 < (int)n_max ? p0.N - n0 : (int)n_max;
            q.x = static_cast<const char*>(p0.x) + (size_t)n0 * p.H * p.W * 192;
            if (p0.out_twin) q.out_twin = static_cast<char*>(p0.out_twin) + (size_t)n0 * p.H * p.W * 192;
            if (p0.out) q.out = p0.out + (size_t)n0 * p.H * p.W * p.out_cstride;
            const int rc = launch_bblockx3(q, s);
            if (rc) return rc;
        }
        return SNCAL_OK;
    }
    p.tiles_x = (p.W + TW - 1) / TW;
    p.tiles_y = (p.N * (p.H + 1) - 1 + TH - 1) / TH;
    p.h1_magic = (unsigned)((((unsigned long long)1 << 32) / (unsigned)(p.H + 1)) + 1ull);
    static const int run_max = getenv("SNCAL_BBX_RUNS") ? atoi(getenv("SNCAL_BBX_RUNS")) : RUN_MAX;
    p.run_max = run_max < 1 ? 1 : run_max;
    p.trace = nullptr;
    static const int dbg = getenv("SNCAL_BBX_DBG") ? atoi(getenv("SNCAL_BBX_DBG")) : 0;
    p.dbg = dbg;
    static int n_wgs = 0;
    if (!n_wgs) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bblockx3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        int dev = 0, cus = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        n_wgs = cus >= 8 ? cus / 8 * 8 : 256;                 // one workgroup per CU, a multiple of the 8 XCDs
    }
    static const char* trace_file = getenv("SNCAL_BBX_TRACE");
    if (trace_file && hipMalloc(&p.trace, (size_t)n_wgs * NW * 64) == hipSuccess) (void)hipMemsetAsync(p.trace, 0, (size_t)n_wgs * NW * 64, s);
    if (!p.ticket) { set_error("launch_bblockx3: no ticket words"); return SNCAL_ERR_ARG; }
    SNCAL_LAUNCH(bblockx3_kernel, dim3((unsigned)n_wgs), dim3(64 * (NW + 2)), (size_t)LDS_BYTES, s, p);
    SNCAL_CHECK_LAUNCH();
    if (p.trace) {      // every launch overwrites the dump: the file holds the last fused block of the run
        std::vector<unsigned long long> h((size_t)n_wgs * NW * 8);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h.data(), p.trace, h.size() * 8, hipMemcpyDeviceToHost);
        (void)hipFree(p.trace);
        if (FILE* f = fopen(trace_file, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
    }
    return SNCAL_OK;
}

}  // namespace sncal
