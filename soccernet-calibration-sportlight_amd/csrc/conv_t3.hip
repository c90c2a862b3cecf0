// EXPERIMENT (round 3, OFF by default: SNCAL_CONV_T3=1 selects it; measured 13.3 ms per step against conv_tt's 12.4 -- see the end of
// this comment).  Three-team persistent 3x3 stride-1 convolution for the wide HRNet branches (96 / 192 / 384 channels), bf16: a variant
// of conv_tt.hip for the bf16 engine (BasicBlock conv3x3 + eval-BN (+ residual) (+ ReLU), /root/reference/src/models/hrnet/hrnet.py:42-58).
//
// Why a third team.  conv_tt's ablation in situ showed its phases ADD UP instead of overlapping: per 32-channel stage a
// team multiplies 3.9k clk and spends ~4.5k on load + synchronisation, plus ~11k of epilogue / set-up per tile; a team is in its MULTIPLY
// phase 35-45 % of its life, two teams keep the matrix pipe busy ~70 % of the time at best (PMC: 0.50 with tails and padding).  The
// duty cycle of a team cannot be raised much (the loading wave's instructions issue at 1/2 .. 1/8 rate beside a multiplying wave on its
// SIMD), but a THIRD team raises the sum: with three teams at ~33 % each the pipe always finds a team whose stage has landed.
// What that costs: three waves per SIMD (<= 168 VGPRs: the residual prefetch of the epilogue is per tile row, the halo offsets are
// three instead of six) and three stage buffers in 160 KB of LDS -> stages of 16 input channels (27 KB of weights + 11.25 KB of
// halo per team), one v_mfma_f32_32x32x16_bf16 K-step per tap, 54 MFMAs per wave and stage.
// Everything else is conv_tt's design: tile 8 x 32 pixels x 96 output channels on the STACKED frames (row s = f (H + 1) + y, one shared
// zero row between frames), accumulators started at the folded-BN shift from an LDS table, LDS tokens (arrive / go / early / done
// counters per team, one MULTIPLY token per CU), host-dealt work items per team, branch-free epilogue through buffer descriptors.
// The epilogue stages one (tile row, 32-channel block) at a time -- 32 pixels x 32 channels fp32, 4.5 KB -- through the wave's OWN block
// of the team's weight region (6-7 KB), so no team-level barrier is needed and nothing is staged where a DMA may land.
// Halo image in LDS: [pixel][2 x 16 B] (16 channels), row pitch 36 pixels: a B fragment read (32 consecutive pixels x two 8-channel
// groups) is one contiguous 1 KB run -- conflict-free without the slot rotation conv_tt needs for its 64-byte pixels.
// Same operands, same fp32 accumulation as conv_tt / the generic kernel; products are summed in another order (16-channel stages), so
// outputs agree to fp32 rounding of the accumulator (tests/test_kernels_gpu.py checked every launch against torch fp32: green).
// RESULT: 161 VGPRs, no spills, three waves per SIMD resident -- and 7 % SLOWER than two teams.  Ablation in situ (ms per step, conv_tt in
// brackets): as built 13.3 (12.4); without the epilogue 11.2 (10.2); without the MFMAs 9.3 (8.6); without the DMA 10.5 (10.0); neither
// MFMAs nor DMA 5.5 (5.0); skeleton only 3.4 (2.5).  With two teams or three, the phases ADD UP: a third team does not make the loading
// waves' work overlap the multiplying wave's, and the doubled number of (16-channel) stages costs 0.9 ms of synchronisation.  Whatever
// serialises DMA issue, epilogue and MFMA issue sits below the team level (per-SIMD issue / the LDS pipeline shared by fragment reads and
// DMA writes), so more teams per CU is not the lever.
#include "common.hpp"
#include "conv_tt.hpp"
#include <cstddef>

namespace sncal {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void;

namespace {
constexpr int NTEAMS = 3;
constexpr int MB = 3, NB = 2;                       // 32-channel blocks, tile rows per wave
constexpr int NKS = 9;                              // taps = K = 16 steps per stage
constexpr int W_PIECES = NKS * MB;                  // 27 pieces of 1 KB: [tap][mb][lane] x 16 B
constexpr int W_BYTES = W_PIECES * 1024;            // 27648
constexpr int HP = 36;                              // halo row pitch in pixels (34 used)
constexpr int HROWS = TT_TH + 2;                    // 10
constexpr int PX_BYTES = 32;                        // 16 channels bf16
constexpr int HALO_PIECES = (HROWS * HP * PX_BYTES + 1023) / 1024;      // 12 (11.25 used)
constexpr int H_BYTES = HALO_PIECES * 1024;         // 12288
constexpr int TEAM_BYTES = W_BYTES + H_BYTES;       // 39936; three teams = 119808
__device__ __host__ constexpr int wp_first(int tw) { return tw * 7 - (tw > 3 ? 1 : 0); }      // 7, 7, 7, 6 weight pieces per wave
constexpr int EPI_PITCH = 36;                       // floats per staged pixel (32 channels + 4)
static_assert(wp_first(4) == W_PIECES && 32 * EPI_PITCH * 4 <= 6 * 1024, "a wave's epilogue staging fits its own block of the weight region");
}  // namespace

__device__ __forceinline__ TTMember t3_load_member(int m) {
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

__global__ __launch_bounds__(256 * NTEAMS, 1) void conv_t3_kernel(const TTParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = wv >> 2, tw = wv & 3;
    const unsigned T = blockIdx.x * (unsigned)NTEAMS + (unsigned)team;
    char* const s_w = smem + team * TEAM_BYTES;
    char* const s_h = s_w + W_BYTES;
    unsigned* const ctrl = reinterpret_cast<unsigned*>(smem + NTEAMS * TEAM_BYTES);       // [0] token, [4 + 4 team + {0,1,2,3}] arrive, go, done, early
    float* const s_bias = reinterpret_cast<float*>(smem + NTEAMS * TEAM_BYTES + 64);
    const int tab1 = P.m[0].cout, tab2 = P.m[0].cout + P.m[1].cout;

    unsigned it = P.team_first[T];
    const unsigned S = P.team_stages[T];

    // B fragment of tap (dy, dx) for tile row jr of this wave: pixel (tw * 2 + jr + dy, l31 + dx), channel group hi
    const char* const bbase = s_h + ((tw * 2 * HP + l31) * PX_BYTES + hi * 16);
    const char* const aptr = s_w + lane * 16;

    f32x16 acc[MB][NB];
    unsigned hv[3];                       // per-lane byte offsets of this wave's halo DMA pieces (stage-independent)
    TTMember M = t3_load_member(0);
    int nb = 0, c = 0, row0 = 0, col0 = 0, tab = 0;

    auto halo_voff = [&](int piece, int ln) -> unsigned {
        const unsigned q = (unsigned)(piece * 64 + ln);
        const unsigned p = q >> 1, cg = q & 1u;
        const unsigned hrow = (p * 1821u) >> 16, hcol = p - hrow * HP;           // p / 36 for p < 2048
        const int s = row0 - 1 + (int)hrow;
        const unsigned f = __umulhi((unsigned)max(s, 0), M.hp1_magic);
        const int y = s - (int)f * (M.H + 1);
        const int x = col0 - 1 + (int)hcol;
        const bool ok = (hrow < (unsigned)HROWS) & (hcol < 34u) & (s >= 0) & ((int)f < M.N) & (y < M.H) & ((unsigned)x < (unsigned)M.W);
        return ok ? (unsigned)((((int)f * M.H + y) * M.W + x) * M.Cin * 2) + cg * 16u : 0x80000000u;
    };

    auto setup_item = [&](const TTItem I) {
        M = t3_load_member(I.member);
        nb = I.nb; row0 = I.row0; col0 = I.col0;
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) hv[jj] = halo_voff(tw + 4 * jj, lane_l);
        tab = I.member == 0 ? 0 : I.member == 1 ? tab1 : tab2;
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
    };

    auto issue_stage = [&](int cc) {
        const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(M.in), 0, (int)M.in_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(M.w), 0, (int)M.w_bytes, 0x00020000);
        const unsigned wbase = (unsigned)((nb * M.chunks + cc) * W_BYTES);
        const int i1 = wp_first(tw + 1);
        for (int i = wp_first(tw); i < i1; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(s_w + i * 1024), 16, (unsigned)(lane * 16), wbase + i * 1024, 0, 0);
        const unsigned cbase = (unsigned)(cc * PX_BYTES);
#pragma unroll
        for (int jj = 0; jj < 3; ++jj)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void*)(s_h + (tw + 4 * jj) * 1024), 16, hv[jj], cbase, 0, 0);
    };

    // (+ residual) (ReLU) -> bf16, one (tile row, 32-channel block) at a time through a wave-private LDS transpose: every lane then
    // stores 8 consecutive channels.  Branch-free: out-of-range offsets drop lanes / rows outside the image, a missing residual is a
    // zero-sized descriptor.
    auto epilogue = [&]() __attribute__((always_inline)) {
        if (P.ablate & 1) return;
        const unsigned out_bytes = (unsigned)(M.N * M.H * M.W * M.out_cstride * 2);
        const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(M.res ? M.res : M.in), 0, M.res ? (int)out_bytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(M.out ? M.out : const_cast<void*>(M.in), 0, M.out ? (int)out_bytes : 0, 0x00020000);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const bool has_res = M.res != nullptr;
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));
        float* const stg = reinterpret_cast<float*>(s_w + wp_first(tw) * 1024);
        // item e of a (row, block): lane -> pixel px = id / 4, 8-channel group grp = id & 3 (two iterations cover 32 x 4)
#pragma unroll
        for (int jr = 0; jr < NB; ++jr) {
            const int srow = row0 + tw * 2 + jr;
            const unsigned f = __umulhi((unsigned)srow, M.hp1_magic);
            const int y = srow - (int)f * (M.H + 1);
            const bool row_ok = ((int)f < M.N) & (y < M.H);
            const unsigned soff = row_ok ? (unsigned)(((((int)f * M.H + y) * M.W + col0) * M.out_cstride + M.out_coff + nb * TT_COUT) * 2) : 0u;
            unsigned voff[2];
            u32x4 rr[MB][2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int id = e * 64 + lane_l, px = id >> 2, grp = id & 3;
                voff[e] = (row_ok & (px < M.W - col0)) ? (unsigned)((px * M.out_cstride + grp * 8) * 2) : 0x80000000u;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    rr[mb][e] = u32x4{0u, 0u, 0u, 0u};
                    if (has_res) rr[mb][e] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, voff[e], soff + (unsigned)(mb * 64), 0);
                }
            }
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<float4*>(stg + l31 * EPI_PITCH + 8 * q + 4 * hi) =
                        make_float4(acc[mb][jr][4 * q], acc[mb][jr][4 * q + 1], acc[mb][jr][4 * q + 2], acc[mb][jr][4 * q + 3]);
                // wave-local hand-off: the LDS operations of one wave complete in order
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int id = e * 64 + lane_l, px = id >> 2, grp = id & 3;
                    const float* sp = stg + px * EPI_PITCH + grp * 8;
                    const float4 lo = *reinterpret_cast<const float4*>(sp), hi4 = *reinterpret_cast<const float4*>(sp + 4);
                    float v[8] = {lo.x, lo.y, lo.z, lo.w, hi4.x, hi4.y, hi4.z, hi4.w};
                    if (has_res) {
                        const bf16x8 r = __builtin_bit_cast(bf16x8, rr[mb][e]);
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] += (float)r[k];
                    }
                    bf16x8 q;
#pragma unroll
                    for (int k = 0; k < 8; ++k) q[k] = (__bf16)v[k];
                    if (M.relu) {
                        typedef short s16x8 __attribute__((ext_vector_type(8)));
                        const s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
                        q = __builtin_bit_cast(bf16x8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, q), z));
                    }
                    u32x4 qd = __builtin_bit_cast(u32x4, q);
                    asm volatile("" : "+v"(qd));
                    __builtin_amdgcn_raw_buffer_store_b128(qd, rs_out, voff[e], soff + (unsigned)(mb * 64), 0);
                    asm volatile("s_nop 1" :: "v"(qd) : "memory");     // store data registers stay untouched behind the store
                }
                // the next block overwrites the staging area: reads above are complete in program order (same wave, LDS in order)
            }
        }
    };

    auto multiply_stage = [&](auto&& near_end) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int jr = 0; jr < NB; ++jr) asm volatile("" : "+v"(acc[mb][jr]));
        bf16x8 a[2][MB], b[2][NB];
        auto load_frags = [&](int t, int buf) {
            const int dy = t / 3, dx = t - dy * 3;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) a[buf][mb] = *reinterpret_cast<const bf16x8*>(aptr + (t * MB + mb) * 1024);
#pragma unroll
            for (int jr = 0; jr < NB; ++jr) b[buf][jr] = *reinterpret_cast<const bf16x8*>(bbase + ((jr + dy) * HP + dx) * PX_BYTES);
        };
        load_frags(0, 0);
#pragma unroll
        for (int t = 0; t < NKS; ++t) {
            const int cur = t & 1;
            if (t == NKS - 2) near_end();
            if (t + 1 < NKS) load_frags(t + 1, cur ^ 1);
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int jr = 0; jr < NB; ++jr)
                    acc[mb][jr] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cur][mb], b[cur][jr], acc[mb][jr], 0, 0, 0);
            if (t + 1 < NKS) {
#pragma unroll
                for (int i = 0; i < MB + NB; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, MB * NB - (MB + NB), 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    unsigned* const w_token = ctrl;
    unsigned* const w_arrive = ctrl + 4 + 4 * team;
    unsigned* const w_go = w_arrive + 1;
    unsigned* const w_done = w_arrive + 2;
    unsigned* const w_early = w_arrive + 3;
    if (tid < 16) ctrl[tid] = 0u;
    for (int i = tid; i < TT_TABLE_MAX; i += 256 * NTEAMS) s_bias[i] = 0.f;
    __syncthreads();
#pragma unroll
    for (int m = 0; m < TT_MAX_MEMBERS; ++m) {
        const int o = m == 0 ? 0 : m == 1 ? tab1 : tab2;
        if (P.m[m].bias && tid < P.m[m].cout) s_bias[o + tid] = P.m[m].bias[tid];
    }
    __syncthreads();
    auto poll = [&](unsigned* p) -> unsigned { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto spin_until = [&](unsigned* p, unsigned target) {
        while ((int)(poll(p) - target) < 0) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
    };
    const unsigned it_last = max(P.team_first[T + 1], it + 1u) - 1u;
    TTItem I_next = P.items[it];
    for (unsigned st = 0; st < S; ++st) {
        if (c == 0) {
            if (st > 0) epilogue();
            setup_item(I_next);
            I_next = P.items[min(it + 1u, it_last)];
        }
        if (!(P.ablate & 4)) issue_stage(c);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // my DMA pieces have landed (and my stores are out)
        if (lane == 0) __hip_atomic_fetch_add(w_arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (tw == 0) {
            spin_until(w_arrive, 4u * (st + 1u));
            for (;;) {
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
        auto release = [&]() {
            unsigned old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(w_early, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (lane == 0 && old == 4u * st + 3u) __hip_atomic_store(w_token, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        };
        if (P.ablate & 2) release(); else multiply_stage(release);
        if (++c == M.chunks) { c = 0; ++it; }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(w_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        spin_until(w_done, 4u * (st + 1u));                       // every wave of the team is done reading this stage
    }
    if (S > 0) { epilogue(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
}

void launch_conv_t3(const TTParams& p, int n_wgs, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_t3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    const size_t lds = (size_t)NTEAMS * TEAM_BYTES + 64 + TT_TABLE_MAX * 4;
    SNCAL_LAUNCH(conv_t3_kernel, dim3((unsigned)n_wgs), dim3(256 * NTEAMS), lds, s, p);
}

}  // namespace sncal
