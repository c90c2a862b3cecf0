// Fused BasicBlock for the 48-channel full-resolution branch (bf16 path):
//     out = ReLU( BN2(conv2( ReLU(BN1(conv1(x))) )) + x )          /root/reference/src/models/hrnet/hrnet.py:42-58
// in ONE kernel.  The 135x240x48 branch is the HBM-bound class of the network (216 FLOP/B per conv): run
// as two convolutions it moves x (read) + mid (write) + mid (read) + x (residual read) + out (write) = 5 tensor
// passes; fused it reads x once (with a 2-pixel halo that the caches absorb) and writes out once.
//
// Workgroup = 4 waves (one per SIMD), PERSISTENT, one per CU; output tile 16 rows x 14 columns:
//   weights  conv1 + conv2, four 21 KB chunks (conv x 24-channel K-chunk), RESIDENT in LDS for the life of the workgroup, in the
//            SAME fragment-ordered packing the generic conv kernel uses for (MI = 3, G = 3): the layers' packed weights and
//            folded-BN shifts are shared and the results are bit-identical to the two-kernel path (same MFMA order, same bf16
//            rounding of mid)
//   x halo   20 x 18 pixels x 48 ch  -> LDS (LDS-DMA, zero fill outside the image); the next tile's halo is requested as soon as
//            conv1 and the residual read are done with this one and lands under conv2
//   conv1    mid = 18 x 16 pixels (the output tile + 1 pixel ring) in 18 row-fragments of 16 pixels; positions outside the
//            image are written as zeros (conv2's zero padding), ReLU'd, bf16 -> LDS
//   conv2    16 row-fragments (lanes 14/15 of a fragment recompute column 13 and are not stored)
//   stores   packed in registers and issued after the NEXT tile's opening barrier, so that the wait for the halo DMA (vmcnt
//            counts loads and stores together) never waits for a store
// LDS 84 + 40 + 31.5 = 155.5 KB -> one workgroup per CU; two barriers per tile.
#include "bblock.hpp"
#include "common.hpp"
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace sncal {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int BB_TH = 16, BB_TW = 14;            // output tile
constexpr int BB_MH = BB_TH + 2, BB_MW = 16;     // mid tile (rows, fragment width)
constexpr int BB_XH = BB_TH + 4, BB_XW = 18;     // x halo tile
constexpr int BB_PS = 112;                       // LDS bytes per pixel: 6 k-groups + 1 padding slot (bank spread)
constexpr int BB_NKS = 7, BB_MI = 3;             // k-steps per 24-channel chunk (27 k-groups -> 28), 48 output channels
constexpr int BB_WBYTES = BB_NKS * BB_MI * 1024; // one (conv, K-chunk) of packed weights: 21 KB
constexpr int BB_XROW = 2048;                    // LDS bytes per halo row: two 1 KB DMA pieces (126 of 128 slots used)
constexpr int BB_XBYTES = BB_XH * BB_XROW;
constexpr int BB_MIDBYTES = BB_MH * BB_MW * BB_PS;
constexpr int BB_LDS = 4 * BB_WBYTES + BB_XBYTES + BB_MIDBYTES;
constexpr int BB_PF = 1;                         // operand prefetch distance in k-steps
constexpr int BB_NW = 8;                         // waves per workgroup: two per SIMD
constexpr int BB_J1 = (BB_MH + BB_NW - 1) / BB_NW, BB_J2 = BB_TH / BB_NW;      // pixel fragments (tile rows) per wave: conv1 (at most), conv2
static_assert(BB_TH % BB_NW == 0 && BB_J1 == BB_J2 + 1 && BB_LDS + 16 <= 160 * 1024, "tile rows split evenly over the waves; one workgroup per CU");

// PERSISTENT (round 2): one workgroup of four waves per CU walks a contiguous range of tiles with the block's whole weight set
// (conv1 + conv2, 4 x 21 KB) RESIDENT in LDS.  The round-1 kernel launched one workgroup per tile and staged those 84 KB for
// every 12 x 14 output pixels (13,824 workgroups, 1.16 GB of weight traffic per launch, four staging rounds with their barriers
// per tile): a workgroup spent 80 % of its life outside its MFMA phases.  Now a tile costs its x halo (one DMA round that lands
// under the previous tile's conv2), two barriers and its MFMAs.  Same packed weights, same K order, same bf16 rounding of
// mid as the two-kernel path -> bit-identical results (tested).
// One LDS-DMA piece (64 lanes x 16 bytes -> 1 KB at LDS byte address `lds_addr`) as inline assembly.  The builtin form makes hipcc's
// wait-count pass treat every later ds_read as a possible reader of the DMA's destination: it put `s_waitcnt vmcnt(0)` in front of
// conv2's first fragment reads -- i.e. waited for the NEXT tile's halo to land before multiplying -- and, because the loop-top wait
// is inline assembly it cannot see, `vmcnt(2..0)` into conv1's first k-step, which waited for the previous tile's STORES.  Together
// ~3k of 13k clk per tile.  Ordering is explicit here (vmcnt(0) + barrier at the top of every tile).
typedef int i32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma_piece(i32x4_t rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}
__device__ __forceinline__ i32x4_t raw_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    return i32x4_t{(int)(unsigned)a, (int)(unsigned)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}

__global__ __launch_bounds__(64 * BB_NW, 1) void bblock48_kernel(const BBlockParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const s_w = smem;                                       // W1c0, W1c1, W2c0, W2c1
    char* const s_x = smem + 4 * BB_WBYTES;
    char* const s_mid = s_x + BB_XBYTES;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, ln = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // tiles of this workgroup: the launch's tiles are cut into 8 contiguous ranges, one per XCD (workgroup b runs on XCD b % 8);
    // the workgroups of an XCD take consecutive tiles of its range, so neighbouring halos meet in one L2
    // Round 5: IN ORDER from a ticket counter of the XCD (p.ticket[xcd]) instead of a fixed stride -- a workgroup whose CU another stream's
    // work holds (the camera solves of the previous batch, CU-masked to one CU per XCD) takes fewer tiles instead of walking its whole
    // share after everybody else has left (the launch then lasted twice as long: 276 us on average against 200 beside the solves).
    // Thread 0 draws the ticket of the NEXT tile at the top of a tile and publishes it behind conv1 (the atomic's round trip rides under
    // the multiplies); everybody reads it behind the mid barrier, where the next halo is requested.
    const int xcd = (int)blockIdx.x & 7;
    const int n_tiles = p.N * p.tiles_y * p.tiles_x;
    const int t_lo = (int)((long)n_tiles * xcd / 8), t_hi = (int)((long)n_tiles * (xcd + 1) / 8);
    unsigned* const s_next = reinterpret_cast<unsigned*>(smem + BB_LDS);        // [2]: the next tile, by parity of the tile count
    if (tid == 0) s_next[0] = (unsigned)t_lo + atomicAdd(p.ticket + xcd, 1u);
    __syncthreads();
    int t = __builtin_amdgcn_readfirstlane((int)s_next[0]);

    const size_t img_bytes = (size_t)p.H * p.W * 48 * 2;
    const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w1), 0, 2 * BB_WBYTES, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w2), 0, 2 * BB_WBYTES, 0x00020000);
    for (int i = wave; i < 2 * BB_NKS * BB_MI; i += BB_NW) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w1, (lds_void*)(s_w + i * 1024), 16, (unsigned)(lane * 16), (unsigned)(i * 1024), 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w2, (lds_void*)(s_w + 2 * BB_WBYTES + i * 1024), 16, (unsigned)(lane * 16), (unsigned)(i * 1024), 0, 0);
    }
    // x halo of tile `tile`, row-aligned: a halo row (18 pixels x 7 slots = 126 slots) is two 64-slot DMA pieces, so a lane's
    // part of the address (pixel-in-row, k-group, left/right bounds) is the same for every row and the row rides in the scalar
    // offset -- no per-piece VALU work.  Slot 6, slots 126/127 and outside-image pixels read out of range -> zeros.
    const unsigned x_lds = (unsigned)(__UINTPTR_TYPE__)(lds_void*)s_x;      // LDS byte address of the halo region
    unsigned xlane[2];                          // lane part of a piece's source offset, relative to the tile's first halo column
    int xpx[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const unsigned slot = (unsigned)(k * 64 + lane);
        const unsigned px = slot / 7u, cg = slot - px * 7u;
        xpx[k] = (cg < 6u && px < (unsigned)BB_XW) ? (int)px : (1 << 20);      // padding slots: never inside the image
        xlane[k] = px * 96u + cg * 16u;
    }
    i32x4_t xrs = raw_rsrc(p.x, (unsigned)img_bytes);       // state of the halo request in flight: image descriptor, lane offsets, first row
    unsigned xv[2] = {0x80000000u, 0x80000000u};
    int xoy = 0;
    auto prepare_x = [&](int n, int oy0, int ox0) {
        xrs = raw_rsrc(reinterpret_cast<const char*>(p.x) + (size_t)n * img_bytes, (unsigned)img_bytes);
        xoy = oy0 - 2;
#pragma unroll
        for (int k = 0; k < 2; ++k)
            xv[k] = (unsigned)(ox0 - 2 + xpx[k]) < (unsigned)p.W ? xlane[k] + (unsigned)((ox0 - 2) * 96) : 0x80000000u;
    };
    auto piece_x = [&](int j) {                  // piece j = half (j & 1) of halo row j >> 1
        const int iy = xoy + (j >> 1);
        const bool rowok = (unsigned)iy < (unsigned)p.H;
        const unsigned voff = rowok ? ((j & 1) ? xv[1] : xv[0]) : 0x80000000u;
        dma_piece(xrs, x_lds + (unsigned)j * 1024u, voff, rowok ? (unsigned)(iy * p.W * 96) : 0u);
    };
    auto issue_x = [&](int n, int oy0, int ox0, int first, int stride) {
        prepare_x(n, oy0, ox0);
        for (int j = first; j < 2 * BB_XH; j += stride) piece_x(j);
    };
    // tile coordinates: decoded once (scalar divisions), then advanced by the stride of the walk
    int tx = t % p.tiles_x, ty = (t / p.tiles_x) % p.tiles_y, tn = t / (p.tiles_x * p.tiles_y);
    if (t < t_hi) issue_x(tn, ty * BB_TH, tx * BB_TW, wave, BB_NW);

    // fragment offsets of k-step s within a 24-channel chunk: k-group kg = 4s + g -> (tap, cg); same order as pack_layer
    auto frag_off = [&](int s, int row_pitch, int chunk) -> int {
        int kg = 4 * s + g;
        kg = kg < 27 ? kg : 26;                  // padded k-group: zero weights, any valid address
        const int tap = kg / 3, cg = kg - tap * 3;
        const int dy = tap / 3, dx = tap - dy * 3;
        return dy * row_pitch + dx * BB_PS + (chunk * 3 + cg) * 16;
    };
    f32x4 acc[BB_MI][BB_J1];
    // folded-BN shifts of the lane's four channels per 16-channel block, loaded ONCE (a global load per tile sat in front of each
    // conv's first MFMA: ~1k clk of exposed L2 latency twice per tile)
    float4 bias1[BB_MI], bias2[BB_MI];
#pragma unroll
    for (int mi = 0; mi < BB_MI; ++mi) {
        bias1[mi] = *reinterpret_cast<const float4*>(p.b1 + mi * 16 + g * 4);
        bias2[mi] = *reinterpret_cast<const float4*>(p.b2 + mi * 16 + g * 4);
    }
    auto init_acc = [&](const float4 (&bias)[BB_MI]) {
#pragma unroll
        for (int mi = 0; mi < BB_MI; ++mi)
#pragma unroll
            for (int j = 0; j < BB_J1; ++j) acc[mi][j] = f32x4{bias[mi].x, bias[mi].y, bias[mi].z, bias[mi].w};
    };
    // one K-chunk of one conv: NJ pixel fragments per wave from the LDS image `src`
    auto mma_chunk = [&](const char* wbuf, const char* src, const int (&boff)[BB_J1], int row_pitch, int chunk, auto nj_c, auto&& between) {
        constexpr int nj = decltype(nj_c)::value;
        // Operand fragments are read one k-step ahead of their MFMAs, and the reads are INTERLEAVED with the MFMAs (one ds_read
        // behind each of the first MFMAs, sched_group_barrier): this wave is alone on its SIMD, so while it issues a block of seven
        // or eight reads (plus their address adds) the matrix pipe drains -- with "all reads, then all MFMAs" a k-step took ~330 clk
        // whether it held 15 MFMAs or 12.  Read order = order of first use (A0, the B fragments, A1, A2).
        bf16x8 a[2][BB_MI], b[2][nj];
        auto load = [&](int s, int buf) {
            const int off = frag_off(s, row_pitch, chunk);
            a[buf][0] = *reinterpret_cast<const bf16x8*>(wbuf + ((s * BB_MI + 0) * 64 + lane) * 16);
#pragma unroll
            for (int j = 0; j < nj; ++j) b[buf][j] = *reinterpret_cast<const bf16x8*>(src + boff[j] + off);
#pragma unroll
            for (int mi = 1; mi < BB_MI; ++mi) a[buf][mi] = *reinterpret_cast<const bf16x8*>(wbuf + ((s * BB_MI + mi) * 64 + lane) * 16);
        };
        load(0, 0);
#pragma unroll
        for (int s = 0; s < BB_NKS; ++s) {
            const int cur = s & 1;
            if (s + 1 < BB_NKS) load(s + 1, cur ^ 1);
#pragma unroll
            for (int mi = 0; mi < BB_MI; ++mi)
#pragma unroll
                for (int j = 0; j < nj; ++j)
                    acc[mi][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[cur][mi], b[cur][j], acc[mi][j], 0, 0, 0);
            if (s + 1 < BB_NKS) {
#pragma unroll
                for (int i = 0; i < BB_MI + nj; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // one LDS read
                }
                __builtin_amdgcn_sched_group_barrier(0x008, BB_MI * nj - (BB_MI + nj), 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            between(s);                         // hook between k-steps (the halo DMA pieces ride here)
        }
    };
    // conv1: mid rows f = wave + 8j (waves 0 and 1 own three, the others two)
    // conv2: output rows r = wave + 8j
    int boff1[BB_J1], boff2[BB_J1];
#pragma unroll
    for (int j = 0; j < BB_J1; ++j) {
        const int f = min(wave + BB_NW * j, BB_MH - 1);
        boff1[j] = f * BB_XROW + ln * BB_PS;
        const int r = min(wave + BB_NW * j, BB_TH - 1), c = min(ln, BB_TW - 1);
        boff2[j] = (r * BB_MW + c) * BB_PS;
    }

    // finished tile waiting for its stores: out-of-range offsets drop the lanes / rows outside the image
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    bf16x4 oq[BB_J2][BB_MI];
    unsigned ovoff[BB_J2];
    __amdgpu_buffer_rsrc_t rs_out = rs_w1;
    bool pending = false;
    auto flush = [&]() {
#pragma unroll
        for (int j = 0; j < BB_J2; ++j)
#pragma unroll
            for (int mi = 0; mi < BB_MI; ++mi)
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, oq[j][mi]), rs_out, ovoff[j], mi * 32, 0);
    };

    // tuning aid (SNCAL_BB_TRACE=<file>): per wave, clocks spent in [0] halo wait + opening barrier, [1] conv1, [2] residual read
    // + mid write + barrier + next halo request, [3] conv2, [4] epilogue arithmetic, [5] tiles
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = 0;
    const bool tracing = p.trace != nullptr;
    auto lap = [&](int k) { if (tracing) { const unsigned long long now = __builtin_amdgcn_s_memtime(); tsum[k] += now - tprev; tprev = now; } };

    auto nohook = [](int) {};
    int t_next = t_hi;
    for (unsigned ti = 0; t < t_hi; ++ti, t = t_next) {
        if (tracing) tprev = __builtin_amdgcn_s_memtime();
        const int n = tn, oy0 = ty * BB_TH, ox0 = tx * BB_TW;
        init_acc(bias1);
        __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0), as a builtin so that hipcc's wait-count pass sees it: my pieces
        asm volatile("" ::: "memory");                        // of this tile's x halo (first tile: and of the weights) have landed
        asm volatile("s_barrier" ::: "memory");               // ... then everyone's, and everyone is past the old mid
        unsigned tk_next = 0;
        if (tid == 0) tk_next = atomicAdd(p.ticket + xcd, 1u);                // (in flight under conv1)
        lap(0);
        // the previous tile's outputs are stored between conv1's two K-chunks: behind the opening wait (whose vmcnt(0) they would
        // otherwise prolong) and while the SIMD's other wave keeps the matrix pipe busy
        if (wave + BB_NW * (BB_J1 - 1) < BB_MH) {             // the first waves own one mid row more than the others
            mma_chunk(s_w, s_x, boff1, BB_XROW, 0, std::integral_constant<int, BB_J1>{}, nohook);
            if (pending && !(p.dbg & 1)) flush();
            mma_chunk(s_w + BB_WBYTES, s_x, boff1, BB_XROW, 1, std::integral_constant<int, BB_J1>{}, nohook);
        } else {
            mma_chunk(s_w, s_x, boff1, BB_XROW, 0, std::integral_constant<int, BB_J1 - 1>{}, nohook);
            if (pending && !(p.dbg & 1)) flush();
            mma_chunk(s_w + BB_WBYTES, s_x, boff1, BB_XROW, 1, std::integral_constant<int, BB_J1 - 1>{}, nohook);
        }
        lap(1);
        // residual = centre of the x halo -> registers: the halo region takes the NEXT tile's halo while conv2 runs
        bf16x4 rx[BB_J2][BB_MI];
#pragma unroll
        for (int j = 0; j < BB_J2; ++j)
#pragma unroll
            for (int mi = 0; mi < BB_MI; ++mi)
                rx[j][mi] = *reinterpret_cast<const bf16x4*>(s_x + (wave + BB_NW * j + 2) * BB_XROW + (ln + 2) * BB_PS + (mi * 16 + g * 4) * 2);
#pragma unroll
        for (int j = 0; j < BB_J1; ++j) {
            const int f = wave + BB_NW * j;
            if (f < BB_MH) {
                const int iy = oy0 - 1 + f, ix = ox0 - 1 + ln;
                const bool inimg = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
#pragma unroll
                for (int mi = 0; mi < BB_MI; ++mi) {
                    // relu(round(x)) == round(relu(x)) and a negative bf16 is a negative int16: ReLU on the packed pairs
                    bf16x4 q;
#pragma unroll
                    for (int e = 0; e < 4; ++e) q[e] = (__bf16)acc[mi][j][e];
                    typedef short s16x4 __attribute__((ext_vector_type(4)));
                    const s16x4 z = {0, 0, 0, 0};
                    const s16x4 r = __builtin_elementwise_max(__builtin_bit_cast(s16x4, q), z);
                    *reinterpret_cast<s16x4*>(s_mid + (f * BB_MW + ln) * BB_PS + (mi * 16 + g * 4) * 2) = inimg ? r : z;
                }
            }
        }
        init_acc(bias2);
        if (tid == 0) s_next[(ti + 1u) & 1u] = (unsigned)t_lo + tk_next;      // (slot of tile ti + 1: last read behind tile ti - 1's mid barrier)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // my mid rows are written, my residual is in registers
        lap(2);
        asm volatile("s_barrier" ::: "memory");               // ... everyone's: mid is complete, nobody reads this x halo any more
        t_next = __builtin_amdgcn_readfirstlane((int)s_next[(ti + 1u) & 1u]);
        tx = t_next % p.tiles_x; ty = (t_next / p.tiles_x) % p.tiles_y; tn = t_next / (p.tiles_x * p.tiles_y);
        lap(6);
        // The next tile's halo lands under conv2.  The four YOUNGER waves (4..7) request it, ten pieces each, before their conv2: the
        // matrix pipe favours the older wave of a SIMD, so waves 0..3 go straight into conv2 (per-wave phase trace, tools/bb_trace.py:
        // requested by all eight waves it cost ~0.9k clk per tile between the barrier and the first conv2 MFMA; spread between the
        // younger waves' k-steps the last pieces were requested too late and the next tile waited for them).
        if (!(p.dbg & 2) && t_next < t_hi && wave >= BB_NW / 2) issue_x(tn, ty * BB_TH, tx * BB_TW, wave - BB_NW / 2, BB_NW / 2);
        lap(7);
        mma_chunk(s_w + 2 * BB_WBYTES, s_mid, boff2, BB_MW * BB_PS, 0, std::integral_constant<int, BB_J2>{}, nohook);
        mma_chunk(s_w + 3 * BB_WBYTES, s_mid, boff2, BB_MW * BB_PS, 1, std::integral_constant<int, BB_J2>{}, nohook);

        if ((p.dbg & 2) && t_next < t_hi) issue_x(tn, ty * BB_TH, tx * BB_TW, wave, BB_NW);
        lap(3);
        // epilogue: + x, ReLU -> packed bf16 in registers (4 channels = 8 bytes per lane and fragment); stored by `flush`
        rs_out = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(p.out) + (size_t)n * img_bytes, 0, (int)img_bytes, 0x00020000);
#pragma unroll
        for (int j = 0; j < BB_J2; ++j) {
            const int r = wave + BB_NW * j, oy = oy0 + r, ox = ox0 + ln;
            ovoff[j] = (ln < BB_TW && oy < p.H && ox < p.W) ? (unsigned)(((oy * p.W + ox) * 48 + g * 4) * 2) : 0x80000000u;
#pragma unroll
            for (int mi = 0; mi < BB_MI; ++mi)
#pragma unroll
                for (int e = 0; e < 4; ++e) oq[j][mi][e] = (__bf16)fmaxf(acc[mi][j][e] + (float)rx[j][mi][e], 0.f);
        }
        pending = true;
        lap(4);
        tsum[5] += 1;
    }
    if (pending) flush();
    // every workgroup holds exactly one failing ticket when it leaves; the last one re-arms the counters for the next launch on this stream
    if (tid == 0 && atomicAdd(p.ticket + 8, 1u) == gridDim.x - 1u) {
#pragma unroll
        for (int i = 0; i < 9; ++i) p.ticket[i] = 0u;
        __threadfence();
    }
    if (tracing && lane == 0)
        for (int k = 0; k < 8; ++k) p.trace[((size_t)blockIdx.x * BB_NW + wave) * 8 + k] = tsum[k];
}

int launch_bblock48(const BBlockParams& p0, hipStream_t s) {
    BBlockParams p = p0;
    p.tiles_x = (p.W + BB_TW - 1) / BB_TW;
    p.tiles_y = (p.H + BB_TH - 1) / BB_TH;
    p.trace = nullptr;
    static const int dbg = getenv("SNCAL_BB_DBG") ? atoi(getenv("SNCAL_BB_DBG")) : 0;
    p.dbg = dbg;
    static int n_wgs = 0;
    if (!n_wgs) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bblock48_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        int dev = 0, cus = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        n_wgs = cus >= 8 ? cus / 8 * 8 : 256;                 // one workgroup per CU, a multiple of the 8 XCDs
    }
    static const char* trace_file = getenv("SNCAL_BB_TRACE");
    if (trace_file && hipMalloc(&p.trace, (size_t)n_wgs * BB_NW * 64) == hipSuccess) (void)hipMemsetAsync(p.trace, 0, (size_t)n_wgs * BB_NW * 64, s);
    if (!p.ticket) { set_error("launch_bblock48: no ticket words"); return SNCAL_ERR_ARG; }
    SNCAL_LAUNCH(bblock48_kernel, dim3((unsigned)n_wgs), dim3(64 * BB_NW), (size_t)BB_LDS + 16, s, p);
    SNCAL_CHECK_LAUNCH();
    if (p.trace) {      // every launch overwrites the dump: the file holds the last fused block of the run
        std::vector<unsigned long long> h((size_t)n_wgs * BB_NW * 8);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h.data(), p.trace, h.size() * 8, hipMemcpyDeviceToHost);
        (void)hipFree(p.trace);
        if (FILE* f = fopen(trace_file, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
    }
    return SNCAL_OK;
}

}  // namespace sncal
