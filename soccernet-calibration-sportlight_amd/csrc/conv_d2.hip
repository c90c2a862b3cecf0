// EXPERIMENT (round 3; SNCAL_CONV_D2=1 selects it).  Two-team persistent 3x3 stride-1 convolution with DOUBLE-BUFFERED 16-channel
// stages, bf16 -- the variant VERDICT r2 item 5 asked for: "16-input-channel stages (27 KB weights + 11 KB halo => four buffers in
// 153 KB, each team double-buffered, its own LOAD hidden under its own multiply and the tile-boundary chain no longer exposes the first
// stage)".  BasicBlock conv3x3 + eval-BN (+ residual) (+ ReLU), /root/reference/src/models/hrnet/hrnet.py:42-58, like conv_tt.hip.
//
// conv_tt's team is serial: LOAD (DMA issue, landing latency) -> MULTIPLY -> LOAD ...; only the partner's multiply hides a LOAD, and at a
// tile boundary the chain epilogue -> set-up -> first stage of the next tile is exposed.  Here a team owns TWO stage buffers
// (2 x (27 KB weights + 12 KB halo); two teams = 156 KB): it requests stage s + 1 BEFORE it multiplies stage s, so a stage has a whole
// period (its own multiply + the partner's) to land, and the first stage of the next tile is already in LDS when the epilogue ends.  The
// load side of a team therefore runs one stage ahead of its compute side and carries its own item state (descriptors, halo offsets).
// Everything else is conv_t3's / conv_tt's: tile 8 x 32 pixels x 96 channels on the stacked frames, [pixel][2 x 16 B] halo image (a B
// fragment read is one contiguous 1 KB run), accumulators started at the folded-BN shift, LDS counters (arrive / go / early / done) and
// one MULTIPLY token per CU, host-dealt items, branch-free epilogue staged through the wave's own block of the weight buffer that was
// multiplied last.  Weights: conv_t3's packing [nb][chunk16][tap 9][mb 3][lane 64] x 16 B.
// RESULT (same box A/B, ms per step of the 64 launches; every launch checked by tests/test_kernels_gpu.py under SNCAL_CONV_D2=1: green):
// 14.1 against conv_tt's 12.1 -- 15 % SLOWER.  168 VGPRs, no spills.  Requesting stage s + 1 after the multiply instead of before it
// (under the partner's multiply; SNCAL_TT_ABLATE bit 8): 14.3.  Ablation: without the epilogue 11.4, without the MFMAs 10.3 (conv_tt:
// 8.6), without the DMA 11.2.  The landing waits are gone, as intended, but they were not what bounds the kernel: twice as many stages
// cost twice as many team hand-offs (arrive / token / go / done through LDS words, ~2k clk per stage), and the load side's item set-up
// plus the DMA issue now sit in front of every token request.  With conv_t3 (three teams) this is the second 16-channel-stage design
// that loses to 32-channel single-buffered stages; the kernel stays OFF by default.
#include "common.hpp"
#include "conv_tt.hpp"
#include <cstddef>

namespace sncal {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((address_space(3))) void lds_void;

namespace {
constexpr int NTEAMS = 2;
constexpr int MB = 3, NB = 2;                       // 32-channel blocks, tile rows per wave
constexpr int NKS = 9;                              // taps = K = 16 steps per stage
constexpr int W_PIECES = NKS * MB;                  // 27 pieces of 1 KB: [tap][mb][lane] x 16 B
constexpr int W_BYTES = W_PIECES * 1024;            // 27648
constexpr int HP = 36;                              // halo row pitch in pixels (34 used)
constexpr int HROWS = TT_TH + 2;                    // 10
constexpr int PX_BYTES = 32;                        // 16 channels bf16
constexpr int HALO_PIECES = (HROWS * HP * PX_BYTES + 1023) / 1024;      // 12 (11.25 used)
constexpr int H_BYTES = HALO_PIECES * 1024;         // 12288
constexpr int BUF_BYTES = W_BYTES + H_BYTES;        // 39936
constexpr int TEAM_BYTES = 2 * BUF_BYTES;           // two stage buffers per team; two teams = 159744 of the CU's 163840 B
__device__ __host__ constexpr int wp_first(int tw) { return tw * 7 - (tw > 3 ? 1 : 0); }      // 7, 7, 7, 6 weight pieces per wave
constexpr int EPI_PITCH = 36;                       // floats per staged pixel (32 channels + 4)
static_assert(wp_first(4) == W_PIECES && 32 * EPI_PITCH * 4 <= 6 * 1024, "a wave's epilogue staging fits its own block of the weight region");
}  // namespace

__device__ __forceinline__ TTMember d2_load_member(int m) {
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

__global__ __launch_bounds__(256 * NTEAMS, 2) void conv_d2_kernel(const TTParams P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int team = wv >> 2, tw = wv & 3;
    const unsigned T = blockIdx.x * (unsigned)NTEAMS + (unsigned)team;
    char* const s_team = smem + team * TEAM_BYTES;            // buffer b: weights at b * BUF_BYTES, halo W_BYTES behind
    unsigned* const ctrl = reinterpret_cast<unsigned*>(smem + NTEAMS * TEAM_BYTES);       // [0] token, [4 + 4 team + {0,1,2,3}] arrive, go, done, early
    float* const s_bias = reinterpret_cast<float*>(smem + NTEAMS * TEAM_BYTES + 64);
    const int tab1 = P.m[0].cout, tab2 = P.m[0].cout + P.m[1].cout;

    const unsigned it0 = P.team_first[T];
    const unsigned S = P.team_stages[T];

    // B fragment of tap (dy, dx) for tile row jr of this wave: pixel (tw * 2 + jr + dy, l31 + dx), channel group hi
    const int b_lane = W_BYTES + ((tw * 2 * HP + l31) * PX_BYTES + hi * 16);
    const int a_lane = lane * 16;

    // ---- compute side: the item whose stages are being multiplied ----------------------------------------------------------------
    f32x16 acc[MB][NB];
    TTMember M = d2_load_member(0);
    int nb = 0, c = 0, row0 = 0, col0 = 0, tab = 0;
    // ---- load side: the item whose stages are being requested, one stage ahead ------------------------------------------------------
    unsigned hv[3];                       // per-lane byte offsets of this wave's halo DMA pieces (stage-independent)
    __amdgpu_buffer_rsrc_t rs_in_l = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(M.in), 0, 0, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_w_l = rs_in_l;
    unsigned wbase_l = 0;                 // byte offset of the next stage's weights
    int cl = 0, chunks_l = 1;
    TTItem I_pending = TTItem{};          // the item the load side set up last = the item of the compute side's next tile

    auto load_setup = [&](const TTItem I) {
        const TTMember Ml = d2_load_member(I.member);
        I_pending = I;
        rs_in_l = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(Ml.in), 0, (int)Ml.in_bytes, 0x00020000);
        rs_w_l = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(Ml.w), 0, (int)Ml.w_bytes, 0x00020000);
        wbase_l = (unsigned)(I.nb * Ml.chunks * W_BYTES);
        chunks_l = Ml.chunks; cl = 0;
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) {
            const unsigned q = (unsigned)((tw + 4 * jj) * 64 + lane_l);
            const unsigned p = q >> 1, cg = q & 1u;
            const unsigned hrow = (p * 1821u) >> 16, hcol = p - hrow * HP;           // p / 36 for p < 2048
            const int s = I.row0 - 1 + (int)hrow;
            const unsigned f = __umulhi((unsigned)max(s, 0), Ml.hp1_magic);
            const int y = s - (int)f * (Ml.H + 1);
            const int x = I.col0 - 1 + (int)hcol;
            const bool ok = (hrow < (unsigned)HROWS) & (hcol < 34u) & (s >= 0) & ((int)f < Ml.N) & (y < Ml.H) & ((unsigned)x < (unsigned)Ml.W);
            hv[jj] = ok ? (unsigned)((((int)f * Ml.H + y) * Ml.W + x) * Ml.Cin * 2) + cg * 16u : 0x80000000u;
        }
    };
    auto issue_stage = [&](int buf) {     // stage cl of the load side's item into buffer buf
        if (P.ablate & 4) return;
        char* const sw = s_team + buf * BUF_BYTES;
        const int i1 = wp_first(tw + 1);
        for (int i = wp_first(tw); i < i1; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w_l, (lds_void*)(sw + i * 1024), 16, (unsigned)(lane * 16), wbase_l + i * 1024, 0, 0);
        const unsigned cbase = (unsigned)(cl * PX_BYTES);
#pragma unroll
        for (int jj = 0; jj < 3; ++jj)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in_l, (lds_void*)(sw + W_BYTES + (tw + 4 * jj) * 1024), 16, hv[jj], cbase, 0, 0);
        wbase_l += W_BYTES;
    };

    auto compute_setup = [&](const TTItem I) {
        M = d2_load_member(I.member);
        nb = I.nb; row0 = I.row0; col0 = I.col0;
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

    // (+ residual) (ReLU) -> bf16, one (tile row, 32-channel block) at a time through a wave-private LDS transpose in the wave's own
    // block of weight buffer `buf` (the one multiplied last: nobody reads it any more, and only this wave refills that block)
    auto epilogue = [&](int buf) __attribute__((always_inline)) {
        if (P.ablate & 1) return;
        const unsigned out_bytes = (unsigned)(M.N * M.H * M.W * M.out_cstride * 2);
        const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(M.res ? M.res : M.in), 0, M.res ? (int)out_bytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(M.out ? M.out : const_cast<void*>(M.in), 0, M.out ? (int)out_bytes : 0, 0x00020000);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const bool has_res = M.res != nullptr;
        int lane_l = lane;
        asm volatile("" : "+v"(lane_l));
        float* const stg = reinterpret_cast<float*>(s_team + buf * BUF_BYTES + wp_first(tw) * 1024);
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
            }
        }
    };

    auto multiply_stage = [&](int buf, auto&& near_end) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int jr = 0; jr < NB; ++jr) asm volatile("" : "+v"(acc[mb][jr]));
        const char* const aptr = s_team + buf * BUF_BYTES + a_lane;
        const char* const bbase = s_team + buf * BUF_BYTES + b_lane;
        bf16x8 a[2][MB], b[2][NB];
        auto load_frags = [&](int t, int bsel) {
            const int dy = t / 3, dx = t - dy * 3;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) a[bsel][mb] = *reinterpret_cast<const bf16x8*>(aptr + (t * MB + mb) * 1024);
#pragma unroll
            for (int jr = 0; jr < NB; ++jr) b[bsel][jr] = *reinterpret_cast<const bf16x8*>(bbase + ((jr + dy) * HP + dx) * PX_BYTES);
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
    if (S == 0) return;
    const unsigned it_last = max(P.team_first[T + 1], it0 + 1u) - 1u;
    unsigned it_l = it0;
    load_setup(P.items[it_l]);
    TTItem I_ln = P.items[min(it_l + 1u, it_last)];           // fetched one item ahead: a scalar load that misses every cache
    issue_stage(0);
    if (++cl == chunks_l) cl = 0;
    for (unsigned st = 0; st < S; ++st) {
        const int buf = (int)(st & 1u);
        if (c == 0) {
            if (st > 0) epilogue(buf ^ 1);                        // the finished tile, staged in the buffer multiplied last
            compute_setup(I_pending);                             // = the item whose first stage was requested a stage ago
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // my pieces of stage st have landed (requested a whole period ago); my stores are out
        if (lane == 0) __hip_atomic_fetch_add(w_arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (!(P.ablate & 8)) {
            if (st + 1u < S) {                                        // request stage st + 1 into the other buffer (everybody is done with stage st - 1: see below)
                if (cl == 0) { ++it_l; load_setup(I_ln); I_ln = P.items[min(it_l + 1u, it_last)]; }
                issue_stage(buf ^ 1);
                if (++cl == chunks_l) cl = 0;
            }
        }
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
        if (P.ablate & 2) release(); else multiply_stage(buf, release);
        if ((P.ablate & 8) && st + 1u < S) {                      // variant (SNCAL_TT_ABLATE bit 8): request stage st + 1 AFTER the multiply, under the partner's
            if (cl == 0) { ++it_l; load_setup(I_ln); I_ln = P.items[min(it_l + 1u, it_last)]; }
            issue_stage(buf ^ 1);
            if (++cl == chunks_l) cl = 0;
        }
        if (++c == M.chunks) c = 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(w_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        spin_until(w_done, 4u * (st + 1u));                       // every wave of the team is done reading stage st: its buffer may be staged in / refilled
    }
    epilogue((int)((S - 1u) & 1u));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

void launch_conv_d2(const TTParams& p, int n_wgs, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_d2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_done = true;
    }
    const size_t lds = (size_t)NTEAMS * TEAM_BYTES + 64 + TT_TABLE_MAX * 4;
    SNCAL_LAUNCH(conv_d2_kernel, dim3((unsigned)n_wgs), dim3(256 * NTEAMS), lds, s, p);
}

}  // namespace sncal
