// 3x3 stride-2 convolutions of the split-arithmetic engine as a PIPELINED persistent kernel (round 6).
// STATUS: built, bit-identical to the generic kernel on every launch (per-kernel tests, goldens), and SLOWER on MI355X: 5.5 against 4.6 ms
// per step for the single launches, 4.7 against 3.9 for the shared ones -- so it is OFF by default (SNCAL_S2P=1 switches it on;
// tests/test_kernels_gpu.py keeps it under the per-kernel oracle).  The ablation that explains it is at the bottom of this comment and in
// profiles/r06_kernel_ab.md.
//
// Replaces, for the fp16x3 engine, the generic kernel's launches of the stride-2 class: transitions and the fuse-down chains of every
// HighResolutionModule (/root/reference/src/models/hrnet/hrnet.py:183-214, 357-391): conv3x3 stride 2 + folded BatchNorm (+ ReLU).
//
// Why.  conv_kernel<x3_t, 3, 2, 2, MI, 3> (conv.hpp) stages a 12-channel chunk (42 KB of weight fragments + a 9 x 65 pixel halo), waits
// for it, multiplies, and only then may request the next chunk: per chunk and workgroup 4.2k clk of stage wait in front of 4.9k clk of
// multiplies (2.0k of them MFMAs), the matrix pipe busy 0.30 (NOTES/design_history_r1_r5.md 11.5, last paragraph).  It relies on the
// CU's second workgroup to multiply meanwhile, and the two drift into phase.  The class's loss is OVERLAP, not bandwidth.
//
// Here one workgroup per CU runs twelve waves with fixed roles:
//   * eight COMPUTE waves (two per SIMD, so that one wave's split VALU work and LDS reads run beside the other's MFMAs),
//     one 16-pixel fragment each (the same 8-fragment tile, the same packed weights, the same k-step order and MFMA pairing as the
//     generic variant: bit-identical results);
//   * four LOADER waves that do nothing but LDS-DMA: chunk k + 1 lands in the second stage buffer while chunk k multiplies, and since the
//     kernel is persistent the NEXT item's first chunk lands under the current item's epilogue -- the cold first stage that every
//     workgroup of the generic kernel waits for;
//   * hand-offs are LDS counters, one PER WAVE (ready[l]: chunks whose pieces of loader l have landed; done[w]: chunks compute wave w has
//     finished reading -- a sum over waves would let fast waves vouch for a slow one), no s_barrier; work items come from the XCD's ticket counter (a CU held by a camera-solve wavefront takes fewer), published
//     through a four-slot ring by loader wave 0.
// The epilogue stores straight from the accumulators (fp32 and / or the split twin): both stage buffers belong to the loaders.
// Items are ordered tile-major, member-minor over up to three convolutions that read the SAME input (conv_shared_s2's order: one of
// the workgroups that need an input tile misses to HBM, the others hit the XCD's L2); a single convolution is the one-member case.
//
// Measured (SNCAL_S2P_ABLATE, ms per step of the 27 single launches; timing only): as built 5.53; without the multiplies 4.01; without
// any DMA 4.25; without the stores 4.32; without multiplies and DMA 1.83; without all three 1.03.  So: multiplies ~2.0, DMA ~1.8, stores
// ~1.2 (16-byte pieces scattered over 16 pixel rows: the generic kernel transposes through LDS, which here belongs to the loaders),
// skeleton ~1.0 (38 us per launch: two LDS hand-offs per chunk at ~300 clk each) -- and multiplies + DMA add up instead of overlapping:
// beside two multiplying waves per SIMD a loader wave issues a vector-memory instruction every ~160 clk (17 pieces per chunk and loader:
// 2.8k clk + the landing latency, against 2.6k clk of multiplies), and with two stage buffers a loader cannot run further ahead.  Eight
// loader waves (128 VGPRs, 14 spilled) measured 5.73: the period is the landing latency then.  What it would take: a third stage buffer
// (8-channel chunks: 49 KB per stage) or weights kept across tiles -- a different packing, not built.
#include "conv.hpp"
#include "x3.hpp"
#include "common.hpp"
#include <cstdlib>

namespace sncal {
namespace {

constexpr int S2_KS = 3, S2_G = 3, S2_GE = 4;
constexpr int S2_NKG = S2_KS * S2_KS * S2_G, S2_NKS = (S2_NKG + 3) / 4;            // 27 k-groups, 7 k-steps per chunk
constexpr int S2_PS = halo_pitch(S2_G, 2), S2_SLOTS = S2_PS / 16;                 // 48 B per halo pixel
constexpr int S2_NCW = 8, S2_NLW = 4;                                             // compute / loader waves
constexpr int S2_W_MAX = S2_NKS * 6 * 1024;                                       // weight fragments of one chunk at MI = 6
constexpr int S2_H_MAX = conv_max_halo_pieces(3, 2, 2, 3) * 1024;                 // largest halo over the tile shapes the host may pick
constexpr int S2_STAGE = S2_W_MAX + S2_H_MAX;
constexpr int S2_CTRL = 2 * S2_STAGE, S2_LDS = S2_CTRL + 128;      // 32 control words
constexpr int S2_MAXH = (conv_max_halo_pieces(3, 2, 2, 3) + S2_NLW - 1) / S2_NLW; // halo pieces per loader wave
static_assert(S2_LDS <= 160 * 1024, "two stage buffers fit the CU's LDS");
static_assert((S2_NLW & (S2_NLW - 1)) == 0 && (S2_NCW & (S2_NCW - 1)) == 0 && S2_NLW + S2_NCW + 8 <= 32, "per-wave counters + ring fit the control words");
enum { C_READY = 0, C_DONE = C_READY + S2_NLW, C_KNOWN = C_DONE + S2_NCW, C_ITEM = C_KNOWN + 4 };                      // C_READY + l, C_DONE + w: chunks delivered / finished; C_ITEM + (round & 3): the round's item, or ITEM_END
constexpr unsigned ITEM_END = 0xffffffffu;

__device__ __forceinline__ unsigned poll(unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void spin_until(unsigned* p, unsigned target) {
    while ((int)(poll(p) - target) < 0) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}
// all of the N (a power of two) per-wave counters at p have reached `target`
template <int N>
__device__ __forceinline__ void spin_until_all(unsigned* p, unsigned target, int lane) {
    while (__builtin_amdgcn_ballot_w64((int)(poll(p + (lane & (N - 1))) - target) >= 0) != ~0ull) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}

struct S2Item { ConvParams q; int mi, nb, n, ty, tx; };

// item j of the XCD's list -> member, n-block, tile
__device__ __forceinline__ bool s2_decode(const ConvSharedParams& sp, unsigned xcd, unsigned j, S2Item& it) {
    const unsigned ipt = sp.first[3];
    const unsigned tl = j / ipt, r = j - tl * ipt;
    const unsigned tile = xcd * sp.tiles_per_xcd + tl;
    if (tl >= sp.tiles_per_xcd || tile >= sp.tiles) return false;
    const int m = r >= sp.first[2] ? 2 : r >= sp.first[1] ? 1 : 0;
    it.q = sp.p[0];
    unsigned f0 = sp.first[0];
    it.mi = sp.mi[0];
    if (m == 1) { it.q = sp.p[1]; f0 = sp.first[1]; it.mi = sp.mi[1]; }
    if (m == 2) { it.q = sp.p[2]; f0 = sp.first[2]; it.mi = sp.mi[2]; }
    it.nb = (int)(r - f0);
    const unsigned qy = conv_udiv(tile, (unsigned)it.q.tiles_x, it.q.tiles_x_magic);
    it.tx = (int)(tile - qy * (unsigned)it.q.tiles_x);
    const unsigned n_u = conv_udiv(qy, (unsigned)it.q.tiles_y, it.q.tiles_y_magic);
    it.ty = (int)(qy - n_u * (unsigned)it.q.tiles_y);
    it.n = (int)n_u;
    return true;
}

// ---- one item on a compute wave: accumulate its fragment over all chunks, then store ----------------------------------------------
template <int MI>
__device__ __forceinline__ void s2_compute_item(const S2Item& it, char* smem, unsigned* ctrl, int cw, int lane, unsigned& kc, float& amax, int ablate) {
    const ConvParams& p = it.q;
    const int g = lane >> 4, ln = lane & 15;
    const int TWF = p.twf, TWF_LOG2 = p.twf_log2, TH = S2_NCW >> TWF_LOG2;
    const int HALO_W = (16 * TWF - 1) * 2 + S2_KS;
    const int oy00 = it.ty * TH, ox0 = it.tx * 16 * TWF;
    const int fr = cw >> TWF_LOG2, fx = cw & (TWF - 1);
    const int boff = ((fr * 2) * HALO_W + (fx * 16 + ln) * 2) * S2_PS;
    const int row_pitch = HALO_W * S2_PS;
    f32x4 acc[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const float4 bs = *reinterpret_cast<const float4*>(p.bias + (it.nb * MI + mi) * 16 + g * 4);
        acc[mi] = f32x4{bs.x, bs.y, bs.z, bs.w};
    }
    auto frag_off = [&](int s) -> int {
        int kg = 4 * s + g;
        kg = kg < S2_NKG ? kg : S2_NKG - 1;              // padded k-groups: weights are zero, the address must stay valid
        const int tap = kg / S2_G, cg = kg - tap * S2_G;
        const int dy = tap / S2_KS, dx = tap - dy * S2_KS;
        return dy * row_pitch + dx * S2_PS + cg * 16;
    };
    for (int c = 0; c < p.cin_chunks; ++c, ++kc) {
        spin_until_all<S2_NLW>(ctrl + C_READY, kc + 1u, lane);                // every loader wave's pieces of this chunk have landed
        const char* const s_w = smem + (kc & 1u) * S2_STAGE;
        const char* const s_in = s_w + S2_W_MAX;
        // the generic kernel's k-step loop (conv.hpp, Elem<T>::X3) at NI = 1: fragments of step s + 1 are fetched while step s multiplies
        if (!(ablate & 1)) {
        f32x4 a[2][MI], b[2];
        x3h x3_wprev[MI][4], x3_hprev[4];
        {
            const int off = frag_off(0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[0][mi] = *reinterpret_cast<const f32x4*>(s_w + (mi * 64 + lane) * 16);
            b[0] = *reinterpret_cast<const f32x4*>(s_in + boff + off);
        }
#pragma unroll
        for (int s = 0; s < S2_NKS; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            if (s + 1 < S2_NKS) {
                const int off = frag_off(s + 1);
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) a[nxt][mi] = *reinterpret_cast<const f32x4*>(s_w + (((s + 1) * MI + mi) * 64 + lane) * 16);
                b[nxt] = *reinterpret_cast<const f32x4*>(s_in + boff + off);
            }
            x3h hcur[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) X3_SPLIT1(b[cur][e], hcur[e], l[e]);
            const x3h8 bx = x3h8{l[0], l[1], l[2], l[3], hcur[0], hcur[1], hcur[2], hcur[3]};
            const bool second = (s & 1) != 0, alone = !second && s + 1 == S2_NKS;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                const x3h8 aw = __builtin_bit_cast(x3h8, a[cur][mi]);
                acc[mi] = X3_MFMA_16x16x32(aw, bx, acc[mi]);
                if (second || alone) {
                    const x3h z = (x3h)0.0f;
                    const x3h8 am = second ? x3h8{x3_wprev[mi][0], x3_wprev[mi][1], x3_wprev[mi][2], x3_wprev[mi][3], aw[0], aw[1], aw[2], aw[3]} : aw;
                    const x3h8 bm = second ? x3h8{x3_hprev[0], x3_hprev[1], x3_hprev[2], x3_hprev[3], hcur[0], hcur[1], hcur[2], hcur[3]}
                                           : x3h8{hcur[0], hcur[1], hcur[2], hcur[3], z, z, z, z};
                    acc[mi] = X3_MFMA_16x16x32(am, bm, acc[mi]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) x3_wprev[mi][e] = aw[e];
                }
            }
            if (!second && !alone) {
#pragma unroll
                for (int e = 0; e < 4; ++e) x3_hprev[e] = hcur[e];
            }
            if (s + 1 < S2_NKS) __builtin_amdgcn_sched_barrier(0);
        }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    // my reads of this stage are over
        if (lane == 0) __hip_atomic_store(ctrl + C_DONE + cw, kc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    // epilogue (conv.hpp's direct form): (+ residual) (ReLU) -> 4 consecutive channels per lane, fp32 and / or the split twin
    const int oy = oy00 + fr, ox = ox0 + fx * 16 + ln;
    if (oy >= p.Hout || ox >= p.Wout || (ablate & 8)) return;
    const size_t pix = ((size_t)it.n * p.Hout + oy) * p.Wout + ox;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int cf = it.nb * MI + mi;
        if (cf >= p.cout_frags) continue;
        const int co = cf * 16 + g * 4;
        if (co >= p.cout) continue;
        float v0 = acc[mi][0], v1 = acc[mi][1], v2 = acc[mi][2], v3 = acc[mi][3];
        const size_t o = pix * p.out_cstride + p.out_coff + co;
        if (p.res) {
            const float4 r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res) + o);
            v0 += r.x; v1 += r.y; v2 += r.z; v3 += r.w;
        }
        if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
        x3_track(amax, v0, v1); x3_track(amax, v2, v3);
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
        float* dst = reinterpret_cast<float*>(p.out) + o;
        if (co + 4 <= p.cout) *reinterpret_cast<float4*>(dst) = make_float4(v0, v1, v2, v3);
        else { dst[0] = v0; if (co + 1 < p.cout) dst[1] = v1; if (co + 2 < p.cout) dst[2] = v2; }
    }
}

// ---- one item on a loader wave: its share of the DMA pieces of every chunk ----------------------------------------------------------
__device__ __forceinline__ void s2_load_item(const S2Item& it, char* smem, unsigned* ctrl, int lw, int lane, unsigned& kc, int ablate) {
    const ConvParams& p = it.q;
    const int MI = it.mi;
    const int TWF = p.twf, TH = S2_NCW >> p.twf_log2;
    const int HALO_W = (16 * TWF - 1) * 2 + S2_KS, HALO_H = (TH - 1) * 2 + S2_KS;
    const int npix = HALO_H * HALO_W;
    const int n_halo = (npix * S2_PS + 1023) / 1024;
    const int oy00 = it.ty * TH, ix0 = it.tx * 16 * TWF * 2 - 1;
    const int cin_groups = (p.Cin + S2_GE - 1) / S2_GE;
    const size_t img_bytes = (size_t)p.Hin * p.Win * p.Cin * 4;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.in)) + (size_t)it.n * img_bytes, 0, (int)img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);
    // per-lane source offsets of this wave's halo pieces (conv.hpp halo_voff: padding slots, pixels outside the image and channel
    // groups beyond Cin get an out-of-range offset -> the DMA writes zeros); the chunk's channel offset rides in the scalar offset
    unsigned hv[S2_MAXH];
#pragma unroll
    for (int jj = 0; jj < S2_MAXH; ++jj) {
        const unsigned slot = (unsigned)((lw + S2_NLW * jj) * 64 + lane);
        const unsigned pix = slot / (unsigned)S2_SLOTS, cg = slot - pix * S2_SLOTS;
        const unsigned hy = __umulhi(pix, p.halo_w_magic), hx = pix - hy * HALO_W;
        const int iy = oy00 * 2 - 1 + (int)hy, ix = ix0 + (int)hx;
        const bool ok = (cg < (unsigned)S2_G) & (pix < (unsigned)npix) & ((unsigned)iy < (unsigned)p.Hin) & ((unsigned)ix < (unsigned)p.Win) &
                        ((int)cg < cin_groups);
        hv[jj] = ok ? (unsigned)(((iy * p.Win + ix) * p.Cin) * 4 + cg * 16) : 0x80000000u;
    }
    const bool has_tail = cin_groups % S2_G != 0;
    const int w_chunk = S2_NKS * MI * 1024;
    for (int c = 0; c < p.cin_chunks; ++c, ++kc) {
        if (kc >= 2u) spin_until_all<S2_NCW>(ctrl + C_DONE, kc - 1u, lane);    // every compute wave is done with chunk kc - 2, which held this buffer
        char* const sw = smem + (kc & 1u) * S2_STAGE;
        char* const si = sw + S2_W_MAX;
        const unsigned wbase = (unsigned)(((size_t)it.nb * p.cin_chunks + c) * w_chunk);
        if (!(ablate & 2))
        for (int i = lw; i < S2_NKS * MI; i += S2_NLW)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(sw + i * 1024), 16, (unsigned)(lane * 16), wbase + i * 1024, 0, 0);
        const unsigned cbase = (unsigned)(c * S2_G * 16);
        const bool tail = has_tail && c == p.cin_chunks - 1;
#pragma unroll
        for (int jj = 0; jj < S2_MAXH; ++jj) {
            const int j = lw + S2_NLW * jj;
            if (j < n_halo && !(ablate & 4)) {
                unsigned voff = hv[jj];
                if (tail) { const int slot = j * 64 + lane; if (c * S2_G + slot % S2_SLOTS >= cin_groups) voff = 0x80000000u; }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void*)(si + j * 1024), 16, voff, cbase, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // my pieces have landed
        if (lane == 0) __hip_atomic_store(ctrl + C_READY + lw, kc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}

__global__ __launch_bounds__(64 * (S2_NCW + S2_NLW)) void conv_s2p_kernel(const ConvSharedParams sp, unsigned* __restrict__ ticket, int ablate) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned* const ctrl = reinterpret_cast<unsigned*>(smem + S2_CTRL);
    if (tid < 32) ctrl[tid] = 0u;
    __syncthreads();
    const unsigned xcd = blockIdx.x & 7u;
    unsigned kc = 0;                                   // running chunk index of this workgroup (loaders and compute waves count alike)
    float amax = 0.f;
    const bool loader = wave >= S2_NCW;
    const int lw = wave - S2_NCW;
    // loader wave 0 draws the tickets, ONE ROUND AHEAD: the atomic's round trip (~2 us) runs under the current item's loading instead of
    // in front of every item (first version: 38 us of skeleton per launch, most of it tickets waited for on the spot)
    unsigned t_next = 0;
    if (loader && lw == 0 && lane == 0) t_next = atomicAdd(ticket + xcd, 1u);
    for (unsigned ti = 0;; ++ti) {
        if (loader && lw == 0) {                       // the round's item, published to everybody
            const unsigned t = (unsigned)__builtin_amdgcn_readfirstlane((int)t_next);
            if (lane == 0) t_next = atomicAdd(ticket + xcd, 1u);           // (in flight until the next round reads it)
            S2Item probe;
            const bool ok = s2_decode(sp, xcd, t, probe);
            // (slot ti & 3 last held round ti - 4; the slowest reader is at most two chunks, i.e. at most two items, behind)
            if (lane == 0) {
                __hip_atomic_store(ctrl + C_ITEM + (ti & 3u), ok ? t : ITEM_END, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(ctrl + C_KNOWN, ti + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        spin_until(ctrl + C_KNOWN, ti + 1u);
        const unsigned j = (unsigned)__builtin_amdgcn_readfirstlane((int)poll(ctrl + C_ITEM + (ti & 3u)));
        if (j == ITEM_END) break;
        S2Item it;
        s2_decode(sp, xcd, j, it);
        if (loader) s2_load_item(it, smem, ctrl, lw, lane, kc, ablate);
        else if (it.mi == 6) s2_compute_item<6>(it, smem, ctrl, wave, lane, kc, amax, ablate);
        else s2_compute_item<3>(it, smem, ctrl, wave, lane, kc, amax, ablate);
    }
    if (!loader) x3_report(amax, sp.p[0].range);
    // every workgroup holds its one failing ticket: the last one to leave re-arms the counters for the next launch on this stream
    if (tid == 64 * S2_NCW && atomicAdd(ticket + 8, 1u) == gridDim.x - 1u) {
#pragma unroll
        for (int i = 0; i < 9; ++i) ticket[i] = 0u;
        __threadfence();
    }
}

}  // namespace

// sp: one to three convolutions on one input (conv_shared_s2's parameters; a single convolution = one member); ticket: nine zeroed
// device words owned by the caller's stream (re-armed by the kernel)
int launch_conv_s2p_x3(const ConvSharedParams& sp, unsigned* ticket, hipStream_t s) {
    static int n_wgs = 0;
    if (!n_wgs) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_s2p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        int dev = 0, cus = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        n_wgs = cus >= 8 ? cus / 8 * 8 : 256;           // one workgroup per CU, a multiple of the 8 XCDs
    }
    if (!ticket) { set_error("launch_conv_s2p_x3: no ticket words"); return SNCAL_ERR_ARG; }
    const unsigned items = 8u * sp.tiles_per_xcd * sp.first[3];
    const unsigned grid = items < (unsigned)n_wgs ? ((items + 7u) / 8u) * 8u : (unsigned)n_wgs;
    static const int ablate = getenv("SNCAL_S2P_ABLATE") ? atoi(getenv("SNCAL_S2P_ABLATE")) : 0;      // tuning aid (timing only, results invalid): 1 = no multiplies, 2 = no weight DMA, 4 = no halo DMA, 8 = no stores
    SNCAL_LAUNCH(conv_s2p_kernel, dim3(grid), dim3(64 * (S2_NCW + S2_NLW)), (size_t)S2_LDS, s, sp, ticket, ablate);
    SNCAL_CHECK_LAUNCH();
    return SNCAL_OK;
}

}  // namespace sncal
