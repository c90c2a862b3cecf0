// Dev probe behind NOTES/design_history_r1_r5.md §9.3: a v_mfma_f32_16x16x16_bf16 that takes as SrcC the vDst of the v_mfma_f32_16x16x32_bf16 issued right
// before it (hipcc 7.2 emits the pair without wait states), with identical and with half-overlapping vDst / SrcC registers, against the
// same arithmetic with s_nop 15 between the two instructions.  Prints how many of the 256 accumulator elements differ.
//   hipcc --offload-arch=gfx950 -O2 tools/dev/mfma_pair_probe.hip -o /tmp/mfma_pair_probe && /tmp/mfma_pair_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>      // 0: back to back, same registers; 1: s_nop 15 between; 2: back to back, vDst of the second overlaps its SrcC by half
__global__ void probe(float* out) {
    const int lane = threadIdx.x;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)((lane * 7 + i * 3) % 11 - 5); b[i] = (__bf16)(float)((lane * 5 + i) % 13 - 6); }
    f32x4 c = {1.f, 2.f, 3.f, 4.f}, d;
    if constexpr (MODE == 0) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\t"
                     "v_mfma_f32_16x16x16_bf16 %0, %3, %4, %0\n\t"
                     "s_nop 15\n\ts_nop 15"
                     : "+v"(c) : "v"(a), "v"(b), "v"(*(bf16x4*)&a), "v"(*(bf16x4*)&b));
        d = c;
    } else if constexpr (MODE == 1) {
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0\n\t"
                     "s_nop 15\n\ts_nop 15\n\t"
                     "v_mfma_f32_16x16x16_bf16 %0, %3, %4, %0\n\t"
                     "s_nop 15\n\ts_nop 15"
                     : "+v"(c) : "v"(a), "v"(b), "v"(*(bf16x4*)&a), "v"(*(bf16x4*)&b));
        d = c;
    } else {
        // registers v[40:43] = SrcC / first vDst, second vDst = v[38:41]
        asm volatile("v_mov_b32 v40, %1\n\tv_mov_b32 v41, %2\n\tv_mov_b32 v42, %3\n\tv_mov_b32 v43, %4\n\t"
                     "s_nop 4\n\t"
                     "v_mfma_f32_16x16x32_bf16 v[40:43], %5, %6, v[40:43]\n\t"
                     "v_mfma_f32_16x16x16_bf16 v[38:41], %7, %8, v[40:43]\n\t"
                     "s_nop 15\n\ts_nop 15\n\t"
                     "v_mov_b32 %0, v38\n\t"
                     : "=v"(d[0]) : "v"(c[0]), "v"(c[1]), "v"(c[2]), "v"(c[3]), "v"(a), "v"(b), "v"(*(bf16x4*)&a), "v"(*(bf16x4*)&b)
                     : "v38", "v39", "v40", "v41", "v42", "v43");
        asm volatile("v_mov_b32 %0, v39\n\tv_mov_b32 %1, v40\n\tv_mov_b32 %2, v41" : "=v"(d[1]), "=v"(d[2]), "=v"(d[3]));
    }
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = d[e];
}

int main() {
    float *d, h[3][256];
    hipMalloc(&d, 256 * 4);
    probe<0><<<1, 64>>>(d); hipMemcpy(h[0], d, 1024, hipMemcpyDeviceToHost);
    probe<1><<<1, 64>>>(d); hipMemcpy(h[1], d, 1024, hipMemcpyDeviceToHost);
    probe<2><<<1, 64>>>(d); hipMemcpy(h[2], d, 1024, hipMemcpyDeviceToHost);
    int bad0 = 0, bad2 = 0;
    for (int i = 0; i < 256; ++i) { bad0 += h[0][i] != h[1][i]; bad2 += h[2][i] != h[1][i]; }
    printf("back to back, same registers: %d of 256 elements differ from the spaced pair (by accumulator register: ", bad0);
    for (int e = 0; e < 4; ++e) { int n = 0; for (int l = 0; l < 64; ++l) n += h[0][l * 4 + e] != h[1][l * 4 + e]; printf("%d ", n); }
    printf(")\n");
    printf("back to back, second vDst overlapping its SrcC by half: %d of 256 differ (elements by register: ", bad2);
    for (int e = 0; e < 4; ++e) { int n = 0; for (int l = 0; l < 64; ++l) n += h[2][l * 4 + e] != h[1][l * 4 + e]; printf("%d ", n); }
    printf(")\n");
    return 0;
}
