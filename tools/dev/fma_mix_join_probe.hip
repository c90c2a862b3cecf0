#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include "../../soccernet-calibration-sportlight_amd/csrc/x3.hpp"
using namespace sncal;
__global__ void k(const unsigned* hi, const unsigned* lo, float* a, float* b, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x; if (i >= n) return;
    a[2 * i] = x3_join<0>(hi[i], lo[i]); a[2 * i + 1] = x3_join<1>(hi[i], lo[i]);
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 h = __builtin_bit_cast(h2, hi[i]), l = __builtin_bit_cast(h2, lo[i]);
    b[2 * i] = (float)l[0] + (float)h[0]; b[2 * i + 1] = (float)l[1] + (float)h[1];
}
int main() {
    const int n = 1 << 22; std::vector<unsigned> hi(n), lo(n); srand(3);
    for (int i = 0; i < n; ++i) { hi[i] = ((unsigned)rand() << 16) ^ rand() ^ ((unsigned)rand() << 31); lo[i] = ((unsigned)rand() << 16) ^ rand() ^ ((unsigned)rand() << 31); }
    unsigned *dh, *dl; float *a, *b;
    (void)hipMalloc(&dh, n * 4); (void)hipMalloc(&dl, n * 4); (void)hipMalloc(&a, n * 8); (void)hipMalloc(&b, n * 8);
    (void)hipMemcpy(dh, hi.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dl, lo.data(), n * 4, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dh, dl, a, b, n);
    std::vector<unsigned> ha(2 * n), hb(2 * n);
    (void)hipMemcpy(ha.data(), a, n * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(hb.data(), b, n * 8, hipMemcpyDeviceToHost);
    long bad = 0, nan = 0;
    for (int i = 0; i < 2 * n; ++i) if (ha[i] != hb[i]) { float x, y; memcpy(&x, &ha[i], 4); memcpy(&y, &hb[i], 4); if (x != x && y != y) { ++nan; continue; } if (bad < 5) printf("%d: %08x %08x (hi %08x lo %08x)\n", i, ha[i], hb[i], hi[i / 2], lo[i / 2]); ++bad; }
    printf("x3_join: %d values, mismatches %ld (NaN-payload-only differences %ld)\n", 2 * n, bad, nan);
}
