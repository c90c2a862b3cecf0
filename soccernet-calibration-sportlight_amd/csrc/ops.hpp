// Launchers of the memory-bound helper kernels (ops.hip) used by the HRNet plan executor.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <algorithm>

namespace sncal {

struct UpsampleAddParams {
    const void* base;        // optional [N][H][W][C] tensor added to the sum (NULL = 0)
    const void* src[4];      // up to 4 low-resolution sources [N][Hs][Ws][C]
    int Hs[4], Ws[4];
    float sy[4], sx[4];      // align_corners=True scales (in-1)/(out-1)
    int nsrc;
    void* out;               // [N][H][W][out_cstride], channels written at out_coff
    int N, H, W, C;
    int out_cstride, out_coff;
    int relu;
    unsigned cg_magic, w_magic;   // filled by the launcher: reciprocals of the odd parts of C/GE and W for the index decode
    unsigned cg_shift, w_shift;   // ... and their power-of-two parts (x / d = (x >> shift) / odd)
    unsigned* range;         // fp32 path, fp16x3 engine: sticky counter of wavefronts that split a value beyond the fp16 range (x3.hpp), or null
    void* out_twin;          // fp32 path, bf16x3 engine: split twin ([16 hi | 16 lo] bf16 per 16-channel group, dense, out_coff 0) of the output, or null;
                             // out may then be null
};

int launch_nchw_to_nhwc(int dtype, const float* x, void* y, int N, int C, int H, int W, hipStream_t s, unsigned* nonfinite = nullptr);   // nonfinite: counter of NaN / inf inputs, or null
int launch_u8hwc_to_nhwc(int dtype, const unsigned char* x, void* y, int N, int H, int W, hipStream_t s);
int launch_upsample_add(int dtype, const UpsampleAddParams& p, hipStream_t s);
int launch_softmax_nchw(const float* logits, int cstride, int C, size_t npix_total, size_t hw, int log_mode,
                        float* out, hipStream_t s);

// fused log-softmax + D1 keypoint decode (decode.hip): NHWC fp32 logits -> (B,C-1,3) keypoints without the heatmap;
// `scratch` holds logsoftmax_decode_scratch() bytes
size_t logsoftmax_decode_scratch(int B, int C, int h, int w);
int launch_kp_finish(const float* rowpart, int row_parts, const float* colpart, int col_parts, int C, int B, int h, int w, int img_h, int img_w,
                     float* kpts, hipStream_t s);
int launch_logsoftmax_decode(const float* logits, int cstride, int C, int B, int h, int w, int img_h, int img_w, float* scratch,
                             float* kpts, hipStream_t s);

// fp8 activation helpers (quant.hip): per-tensor |x| maximum (atomicMax into *d_out, a float's bit pattern; zero it first) and
// bf16 -> fp8 e4m3 quantisation x / scale saturated to +-448
int launch_absmax_bf16(const void* x, size_t n, unsigned* d_out, hipStream_t s);
int launch_quantize_fp8(const void* x, void* y, size_t n, float scale, hipStream_t s);
// fp32 -> split twin ([16 hi | 16 lo] bf16 per 16-channel group) for the bf16x3 convolutions; n % 16 == 0 (C % 16 == 0)
int launch_split_f32(const void* x, void* y, size_t n, hipStream_t s, unsigned* range = nullptr);      // range: x3.hpp x3_report, or null

}  // namespace sncal
