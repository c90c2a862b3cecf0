// Does rocprofv3's FETCH_SIZE on gfx950 need the x2 read correction for ISOLATED 64-byte pieces?  (run on the GPU box)
//   hipcc --offload-arch=gfx950 -O2 tools/dev/fetch_probe.hip -o /tmp/fetch_probe
//   cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/fp -- /tmp/fetch_probe
// The guide's correction (FETCH_SIZE reports half of a wide coalesced read stream) was calibrated on contiguous streams, where the
// memory side issues 128-byte requests.  The fused head reads 64-byte box pieces (32 channels x 2 B of one source pixel, next piece
// 1600 B away).  Three kernels read the SAME number of useful bytes (256 MB, far beyond L2 + Infinity Cache, each byte once):
//   stream   lanes read consecutive 16 B: a wave covers 1 KB contiguous
//   piece64  groups of 4 lanes read one 64-byte piece, pieces 1600 B apart (the head's pattern)
//   piece128 groups of 8 lanes read one 128-byte piece, pieces 1600 B apart
// FETCH_SIZE (KB) x 1024 / 256 MB = 0.5 means the x2 correction applies to that pattern, 1.0 means it does not.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void stream_k(const u32x4* __restrict__ x, size_t n16, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const u32x4 v = x[i]; acc ^= v[0] ^ v[1] ^ v[2] ^ v[3]; }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int LANES>       // LANES x 16 B per piece
__global__ void piece_k(const char* __restrict__ x, size_t n_pieces, size_t stride, unsigned* out) {
    unsigned acc = 0;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x, nt = (size_t)gridDim.x * blockDim.x;
    for (size_t i = t; i < n_pieces * LANES; i += nt) {
        const size_t piece = i / LANES, part = i % LANES;
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + piece * stride + part * 16);
        acc ^= v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main() {
    const size_t useful = 256ull << 20, stride = 1600;
    const size_t n64 = useful / 64, n128 = useful / 128;
    char* buf; unsigned* out;
    const size_t span = n64 * stride + 4096;                 // 6.7 GB
    if (hipMalloc(&buf, span) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 1, span);
    hipDeviceSynchronize();
    stream_k<<<4096, 256>>>(reinterpret_cast<const u32x4*>(buf), useful / 16, out);
    piece_k<4><<<4096, 256>>>(buf, n64, stride, out);
    piece_k<8><<<4096, 256>>>(buf, n128, stride, out);
    hipDeviceSynchronize();
    printf("useful bytes per kernel: %zu (stream_k, piece_k<4>, piece_k<8>)\n", useful);
    return 0;
}
