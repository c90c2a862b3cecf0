// Dev probe: which SIMD does wave w of a 512-thread workgroup run on?  (HW_REG_HW_ID bits [5:4] on gfx9)
// hipcc --offload-arch=gfx950 -O2 tools/dev/simd_probe.hip -o /tmp/simd_probe && /tmp/simd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void probe(unsigned* out) {
    extern __shared__ char smem[];
    unsigned id = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = id;
}
int main() {
    unsigned* d; hipMalloc(&d, 4096 * 8 * 4);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    probe<<<256, 512, 160 * 1024>>>(d);
    static unsigned h[4096 * 8]; hipMemcpy(h, d, 256 * 8 * 4, hipMemcpyDeviceToHost);
    int paired = 0, total = 0, hist[4][4] = {};
    for (int b = 0; b < 256; ++b) {
        for (int w = 0; w < 4; ++w) { int s0 = (h[b * 8 + w] >> 4) & 3, s1 = (h[b * 8 + w + 4] >> 4) & 3; paired += s0 == s1; ++total; }
        for (int w = 0; w < 8; ++w) hist[w & 3][(h[b * 8 + w] >> 4) & 3]++;
        if (b < 6) { printf("wg %d simd:", b); for (int w = 0; w < 8; ++w) printf(" %u", (h[b * 8 + w] >> 4) & 3); printf("  cu %u se %u\n", (h[b*8] >> 8) & 15, (h[b*8] >> 13) & 7); }
    }
    printf("wave w and w+4 on the same SIMD: %d of %d\n", paired, total);
    return 0;
}
