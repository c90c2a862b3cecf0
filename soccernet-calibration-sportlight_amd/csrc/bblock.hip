// Fused BasicBlock for the 48-channel full-resolution branch (bf16 path):
//     out = ReLU( BN2(conv2( ReLU(BN1(conv1(x))) )) + x )          /root/reference/src/models/hrnet/hrnet.py:42-58
// in ONE kernel.  The 135x240x48 branch is the HBM-bound class of the network (216 FLOP/B per conv): run
// as two convolutions it moves x (read) + mid (write) + mid (read) + x (residual read) + out (write) = 5 tensor
// passes; fused it reads x once (with a 2-pixel halo that the caches absorb) and writes out once.
//
// Workgroup = 4 waves, output tile 12 rows x 14 columns:
//   x halo   16 x 18 pixels x 48 ch  -> LDS once (LDS-DMA, zero fill outside the image), also serves as the residual
//   conv1    mid = 14 x 16 pixels (the output tile + 1 pixel ring) in 14 row-fragments of 16 pixels; positions
//            outside the image are written as zeros (conv2's zero padding), ReLU'd, bf16 -> LDS
//   conv2    12 row-fragments (lanes 14/15 of a fragment recompute column 13 and are not stored)
// Weights come in four 21 KB chunks (conv1 / conv2 x two 24-channel K-chunks), in the SAME fragment-ordered packing
// the generic conv kernel uses for (MI = 3, G = 3), so the layers' packed weights and folded-BN shifts are shared and
// the results are bit-identical to the two-kernel path (same MFMA order, same bf16 rounding of mid).
// Only the first staging round is exposed: LDS regions are time-shared so that every later chunk is in flight under
// MFMAs --   W: W1c0 -> W2c0        M: W1c1 -> mid        X: x halo -> W2c1 (after the residual moved to registers)
// LDS 77.5 KB -> two workgroups per CU.
#include "bblock.hpp"
#include "common.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace sncal {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((address_space(3))) void lds_void;

constexpr int BB_TH = 12, BB_TW = 14;            // output tile
constexpr int BB_MH = BB_TH + 2, BB_MW = 16;     // mid tile (rows, fragment width)
constexpr int BB_XH = BB_TH + 4, BB_XW = 18;     // x halo tile
constexpr int BB_PS = 112;                       // LDS bytes per pixel: 6 k-groups + 1 padding slot (bank spread)
constexpr int BB_NKS = 7, BB_MI = 3;             // k-steps per 24-channel chunk (27 k-groups -> 28), 48 output channels
constexpr int BB_WBYTES = BB_NKS * BB_MI * 1024;
constexpr int BB_XROW = 2048;                    // LDS bytes per halo row: two 1 KB DMA pieces (126 of 128 slots used)
constexpr int BB_XBYTES = BB_XH * BB_XROW;
constexpr int BB_MIDBYTES = BB_MH * BB_MW * BB_PS;
constexpr int BB_LDS = BB_XBYTES + BB_MIDBYTES + BB_WBYTES;

__global__ __launch_bounds__(256, 2) void bblock48_kernel(const BBlockParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const s_x = smem;
    char* const s_mid = smem + BB_XBYTES;
    char* const s_w = smem + BB_XBYTES + BB_MIDBYTES;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, ln = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile = blockIdx.x;
    const int tx = tile % p.tiles_x; tile /= p.tiles_x;
    const int ty = tile % p.tiles_y;
    const int n = tile / p.tiles_y;
    const int oy0 = ty * BB_TH, ox0 = tx * BB_TW;

    unsigned long long* const trc = p.trace ? p.trace + (size_t)blockIdx.x * 16 : nullptr;
    auto stamp = [&](int slot) { if (trc && tid == 0) trc[slot] = __builtin_amdgcn_s_memtime(); };
    if (trc && tid == 0) trc[0] = ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32) | __builtin_amdgcn_s_getreg((31 << 11) | 20);
    stamp(1);
    const size_t img_bytes = (size_t)p.H * p.W * 48 * 2;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.x)) + (size_t)n * img_bytes, 0, (int)img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w1), 0, 2 * BB_WBYTES, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w2), 0, 2 * BB_WBYTES, 0x00020000);
    auto issue_w = [&](const __amdgpu_buffer_rsrc_t rs, char* dst, int chunk) {
        for (int i = wave; i < BB_NKS * BB_MI; i += 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(dst + i * 1024), 16, (unsigned)(lane * 16),
                                                     (unsigned)(chunk * BB_WBYTES + i * 1024), 0, 0);
    };
    issue_w(rs_w1, s_w, 0);
    // x halo, row-aligned: a halo row (18 pixels x 7 slots = 126 slots) is two 64-slot DMA pieces, so a lane's part of
    // the address (pixel-in-row, k-group, left/right bounds) is the same for every row and the row rides in the scalar
    // offset -- no per-piece VALU work.  Slot 6, slots 126/127 and outside-image pixels read out of range -> zeros.
    unsigned xv[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const unsigned slot = (unsigned)(k * 64 + lane);
        const unsigned px = slot / 7u, cg = slot - px * 7u;
        const int ix = ox0 - 2 + (int)px;
        const bool ok = (cg < 6u) & (px < (unsigned)BB_XW) & ((unsigned)ix < (unsigned)p.W);
        xv[k] = ok ? (unsigned)(ix * 96 + (int)cg * 16) : 0x80000000u;
    }
    for (int j = wave; j < 2 * BB_XH; j += 4) {
        const int hy = j >> 1, iy = oy0 - 2 + hy;
        const bool rowok = (unsigned)iy < (unsigned)p.H;
        const unsigned voff = rowok ? ((j & 1) ? xv[1] : xv[0]) : 0x80000000u;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void*)(s_x + j * 1024), 16, voff, rowok ? (unsigned)(iy * p.W * 96) : 0u, 0, 0);
    }
    // conv1's second chunk goes LAST, into the (still unused) mid region: the first MFMA phase only waits for what was
    // issued before it (loads complete in order; wave 0 owns 6 of the 21 pieces, the others 5) and W1c1 lands under it
    issue_w(rs_w1, s_mid, 1);

    // fragment offsets of k-step s within a 24-channel chunk: k-group kg = 4s + g -> (tap, cg); same order as pack_layer
    auto frag_off = [&](int s, int row_pitch, int chunk) -> int {
        int kg = 4 * s + g;
        kg = kg < 27 ? kg : 26;                  // padded k-group: zero weights, any valid address
        const int tap = kg / 3, cg = kg - tap * 3;
        const int dy = tap / 3, dx = tap - dy * 3;
        return dy * row_pitch + dx * BB_PS + (chunk * 3 + cg) * 16;
    };
    f32x4 acc[BB_MI][4];
    auto init_acc = [&](const float* bias) {
#pragma unroll
        for (int mi = 0; mi < BB_MI; ++mi) {
            const float4 bs = *reinterpret_cast<const float4*>(bias + mi * 16 + g * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[mi][j] = f32x4{bs.x, bs.y, bs.z, bs.w};
        }
    };
    // one K-chunk of one conv: NJ pixel fragments per wave from the LDS image `src`
    auto mma_chunk = [&](const char* wbuf, const char* src, const int (&boff)[4], int row_pitch, int chunk, int nj) {
        bf16x8 a[2][BB_MI], b[2][4];
        {
            const int off = frag_off(0, row_pitch, chunk);
#pragma unroll
            for (int mi = 0; mi < BB_MI; ++mi) a[0][mi] = *reinterpret_cast<const bf16x8*>(wbuf + (mi * 64 + lane) * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) if (j < nj) b[0][j] = *reinterpret_cast<const bf16x8*>(src + boff[j] + off);
        }
#pragma unroll
        for (int s = 0; s < BB_NKS; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            if (s + 1 < BB_NKS) {
                const int off = frag_off(s + 1, row_pitch, chunk);
#pragma unroll
                for (int mi = 0; mi < BB_MI; ++mi)
                    a[nxt][mi] = *reinterpret_cast<const bf16x8*>(wbuf + (((s + 1) * BB_MI + mi) * 64 + lane) * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) if (j < nj) b[nxt][j] = *reinterpret_cast<const bf16x8*>(src + boff[j] + off);
            }
#pragma unroll
            for (int mi = 0; mi < BB_MI; ++mi)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < nj) acc[mi][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[cur][mi], b[cur][j], acc[mi][j], 0, 0, 0);
            if (s + 1 < BB_NKS) __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto round_done = [&]() {                    // my DMA pieces landed and my LDS writes completed; then everyone's
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
    };

    // ---- conv1: mid rows f = wave + 4j (row 13 is recomputed by the waves that own fewer rows; only f < 14 is stored)
    int boff1[4], boff2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = min(wave + 4 * j, BB_MH - 1);
        boff1[j] = f * BB_XROW + ln * BB_PS;
        const int r = min(wave + 4 * j, BB_TH - 1), c = min(ln, BB_TW - 1);
        boff2[j] = (r * BB_MW + c) * BB_PS;
    }
    init_acc(p.b1);
    if (wave == 0) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");      // my x-halo and W1c0 pieces landed
    else asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");      // ... everyone's
    stamp(2);
    mma_chunk(s_w, s_x, boff1, BB_XROW, 0, 4);
    stamp(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my W1c1 pieces landed
    asm volatile("s_barrier" ::: "memory");      // everyone's did, and everyone finished reading W1c0
    issue_w(rs_w2, s_w, 0);                      // conv2's first chunk lands under conv1's second
    stamp(4);
    mma_chunk(s_mid, s_x, boff1, BB_XROW, 1, 4);
    stamp(5);
    asm volatile("s_barrier" ::: "memory");      // everyone finished reading W1c1 (mid region) and the x halo
    // residual = centre of the x halo -> registers, so that the halo region can take conv2's second chunk
    bf16x4 rx[3][BB_MI];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int mi = 0; mi < BB_MI; ++mi)
            rx[j][mi] = *reinterpret_cast<const bf16x4*>(s_x + (wave + 4 * j + 2) * BB_XROW + (ln + 2) * BB_PS + (mi * 16 + g * 4) * 2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int f = wave + 4 * j;
        if (f < BB_MH) {
            const int iy = oy0 - 1 + f, ix = ox0 - 1 + ln;
            const bool inimg = ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
#pragma unroll
            for (int mi = 0; mi < BB_MI; ++mi) {
                bf16x4 q;
#pragma unroll
                for (int e = 0; e < 4; ++e) q[e] = (__bf16)(inimg ? fmaxf(acc[mi][j][e], 0.f) : 0.f);
                *reinterpret_cast<bf16x4*>(s_mid + (f * BB_MW + ln) * BB_PS + (mi * 16 + g * 4) * 2) = q;
            }
        }
    }
    // ---- conv2: output rows r = wave + 4j, j < 3
    init_acc(p.b2);
    stamp(6);
    round_done();                                // W2c0 landed, mid is visible, everyone holds its residual
    issue_w(rs_w2, s_x, 1);                      // conv2's second chunk lands in the halo region under conv2's first
    stamp(7);
    mma_chunk(s_w, s_mid, boff2, BB_MW * BB_PS, 0, 3);
    stamp(8);
    round_done();
    stamp(9);
    mma_chunk(s_x, s_mid, boff2, BB_MW * BB_PS, 1, 3);
    stamp(10);

    // ---- epilogue: + x, ReLU, bf16 store (4 channels = 8 bytes per lane and fragment)
    __bf16* const out = reinterpret_cast<__bf16*>(p.out) + (size_t)n * p.H * p.W * 48;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int r = wave + 4 * j, oy = oy0 + r, ox = ox0 + ln;
        if (ln < BB_TW && oy < p.H && ox < p.W) {
#pragma unroll
            for (int mi = 0; mi < BB_MI; ++mi) {
                bf16x4 q;
#pragma unroll
                for (int e = 0; e < 4; ++e) q[e] = (__bf16)fmaxf(acc[mi][j][e] + (float)rx[j][mi][e], 0.f);
                *reinterpret_cast<bf16x4*>(out + ((size_t)oy * p.W + ox) * 48 + mi * 16 + g * 4) = q;
            }
        }
    }
    stamp(11);
}

int launch_bblock48(const BBlockParams& p0, hipStream_t s) {
    BBlockParams p = p0;
    p.tiles_x = (p.W + BB_TW - 1) / BB_TW;
    p.tiles_y = (p.H + BB_TH - 1) / BB_TH;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&bblock48_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    static const char* trace_file = getenv("SNCAL_BB_TRACE");
    const size_t nwg = (size_t)p.tiles_x * p.tiles_y * p.N;
    p.trace = nullptr;
    if (trace_file && hipMalloc(&p.trace, nwg * 128) == hipSuccess) (void)hipMemsetAsync(p.trace, 0, nwg * 128, s);
    SNCAL_LAUNCH(bblock48_kernel, dim3((unsigned)nwg), dim3(256), (size_t)BB_LDS, s, p);
    SNCAL_CHECK_LAUNCH();
    if (p.trace) {      // every launch overwrites the dump: the file holds the last fused block of the run
        std::vector<unsigned long long> h(nwg * 16);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(h.data(), p.trace, nwg * 128, hipMemcpyDeviceToHost);
        (void)hipFree(p.trace);
        if (FILE* f = fopen(trace_file, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
    }
    return SNCAL_OK;
}

}  // namespace sncal
