// LDS-DMA throughput microbenchmark (dev aid): bytes/clk/CU for synchronous rounds vs. pipelined issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((address_space(3))) void lds_void;
template <int P, bool PIPE>
__global__ __launch_bounds__(256, 2) void k(const char* src, size_t region, int shared, int rounds, unsigned long long* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* base = shared ? src : src + (size_t)blockIdx.x * region;
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)region, 0x00020000);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rounds; ++r) {
        const unsigned rbase = (unsigned)(((size_t)r * 4 * P * 1024) % (region - 4 * P * 1024 + 1)) & ~1023u;
#pragma unroll
        for (int i = 0; i < P; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + ((r & 1) * 4 * P + wave * P + i) * 1024), 16,
                                                     (unsigned)(lane * 16), rbase + (wave * P + i) * 1024, 0, 0);
        if (PIPE) { if (P <= 8) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(P) : "memory"); else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(P) : "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
template <int P, bool PIPE>
void run(const char* d, size_t region, int shared, int wgs_per_cu, unsigned long long* dout) {
    const int rounds = 200, grid = 256 * wgs_per_cu;
    hipFuncSetAttribute((const void*)&k<P, PIPE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    size_t lds = std::max<size_t>((size_t)2 * 4 * P * 1024, wgs_per_cu == 1 ? 81 * 1024 : wgs_per_cu == 2 ? 54 * 1024 : 33 * 1024);
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k<P, PIPE>), dim3(grid), dim3(256), lds, 0, d, region, shared, rounds, dout);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(grid);
    hipMemcpy(h.data(), dout, grid * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += v; mean /= grid;
    const double bytes = (double)rounds * 4 * P * 1024;
    printf("P=%2d/wave (%3d KB/round) %s %s wgs/cu=%d: %7.0f clk/round  %.1f B/clk/WG  %.1f B/clk/CU\n", P, 4 * P, PIPE ? "pipelined" : "sync     ",
           shared ? "shared(L2)" : "private   ", wgs_per_cu, mean / rounds, bytes / mean, bytes / mean * wgs_per_cu);
}
int main() {
    const size_t region = 1 << 20;   // private: 1 MB per WG (768 MB total > L2+MALL); shared: 1 MB total
    char* d; unsigned long long* dout;
    hipMalloc(&d, region * 768); hipMemset(d, 1, region * 768); hipMalloc(&dout, 768 * 8);
    for (int shared = 1; shared >= 0; --shared)
        for (int w = 1; w <= 3; ++w) {
            run<5, false>(d, region, shared, w, dout); run<10, false>(d, region, shared, w, dout); run<20, false>(d, region, shared, w, dout);
            run<5, true>(d, region, shared, w, dout); run<10, true>(d, region, shared, w, dout); run<20, true>(d, region, shared, w, dout);
        }
    return 0;
}
