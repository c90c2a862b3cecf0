/*
 * sncal.h -- C ABI of libsncal.so: the MI355X (gfx950) implementation of the per-frame camera
 * calibration hot path of NikolasEnt/soccernet-calibration-sportlight.
 *
 * The reference has no FFI: the path sits behind three Python call surfaces.  Each
 * entry point below names the reference interface it replaces (file:line under /root/reference);
 * the Python host mirror (the .py files of soccernet-calibration-sportlight_amd) binds them with ctypes and
 * INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative sncal_status; the message of the last
 *     failure on the calling thread is available from sncal_last_error();
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); nothing synchronises
 *     the stream or the device unless stated;
 *   - pointers named d_* are DEVICE pointers, h_* are HOST pointers; the library never allocates
 *     behind the caller's back on the data path: scratch comes from caller-provided workspaces
 *     whose size is returned by the matching *_workspace() query (the one exception is the
 *     convenience form sncal_calibrate, documented there; sncal_calibrate_ws follows the rule);
 *   - no torch / C++ types cross the boundary: plain pointers, ints, floats, doubles.
 */
#ifndef SNCAL_H
#define SNCAL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SNCAL_VERSION 100

typedef enum {
    SNCAL_OK = 0,
    SNCAL_ERR_ARG = -1,        /* invalid argument (shape, null pointer, unsupported size)  */
    SNCAL_ERR_HIP = -2,        /* a HIP runtime call or kernel launch failed                */
    SNCAL_ERR_STATE = -3,      /* object not finalised / weights missing                    */
    SNCAL_ERR_WORKSPACE = -4,  /* caller workspace too small                                */
    SNCAL_ERR_UNSUPPORTED = -5,/* well-formed input that this build does not handle (e.g. progressive JPEG) */
    SNCAL_ERR_RANGE = -6       /* fp16x3 engine: a folded weight (finalize) or an activation (sncal_hrnet_range_status) outside what
                                  fp16 hi + lo halves represent; the reference's fp32 predict() has no such limit: use SNCAL_F32 */
} sncal_status;

typedef enum { SNCAL_F32 = 0, SNCAL_BF16 = 1, SNCAL_FP8 = 2, SNCAL_BF16X3 = 3 } sncal_dtype;

int sncal_version(void);
const char* sncal_last_error(void);
/* Name of the split-arithmetic engine this library was built with (SNCAL_BF16X3): "fp16x3" (fp16 hi + fp16 lo, the default since round 4)
 * or "bf16x3" (bf16 hi + lo, -DSNCAL_X3_F16=0).  No reference counterpart: the reference's predict() is plain fp32
 * (src/models/hrnet/metamodel.py:127-134); the engine reproduces it on the 16-bit matrix pipe (NOTES/design_history_r1_r5.md §9.3 / 10). */
const char* sncal_x3_name(void);

/* ------------------------------------------------------------------------------------------------
 * D1  keypoint heatmap decode
 * replaces HRNetPredictionTransform.__call__  src/models/hrnet/transforms.py:228-239
 *   d_logp  (B,C,h,w) fp32 log-probabilities, NCHW contiguous
 *   d_out   (B,C-1,3) fp32 rows [x_px, y_px, conf]; the last (background) channel is dropped
 *   x = first column holding the channel maximum of exp(logp), y = first row holding it (the two
 *   come from separate reductions, exactly like the reference), conf = that maximum,
 *   x_px = x*img_w/w, y_px = y*img_h/h.   exp is float32(exp(float64(.))) -- see oracle/decode.py.
 * ---------------------------------------------------------------------------------------------- */
int sncal_heatmap_decode(const float* d_logp, int B, int C, int h, int w, int img_h, int img_w,
                         float* d_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * L2  line heatmap 2-peak decode
 * replaces EHMPredictionTransform.__call__ / mask_heat_points_gauss
 *          src/models/line/transforms.py:217-280
 *   d_heat (B,C,h,w) fp32;  d_out (B,C,2,3) fp32 rows [x*scale, y*scale, value]
 * ---------------------------------------------------------------------------------------------- */
int sncal_line_decode(const float* d_heat, int B, int C, int h, int w, float sigma, float scale,
                      float* d_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * L3 + L4  line peaks -> line equations -> the 30 intersection keypoint candidates
 * replaces get_line_data / calculate_slope_intercept  src/utils/export_line_result.py:51-131,
 *          CameraCreator.__init__ lines ingestion     src/models/hrnet/prediction.py:105-124,
 *          line_eq_intersection                       src/models/hrnet/prediction.py:643-653
 *   d_peaks (B,23,2,3) fp32 rows [x, y, p] in heatmap units (sncal_line_decode with scale 1; LINE_CLS order);
 *   d_out   (B,30,3) fp32 rows [x, y, valid] = sncal_calibrate's d_line_pts (ids of LINE_INTERSECTIONS).
 * Arithmetic as in the reference's pinned numpy 1.24.2: float32 coordinates, float64 from `+ 1e-5` on.
 * ---------------------------------------------------------------------------------------------- */
int sncal_lines_to_points(const float* d_peaks, int B, float scale, double prob_thre, float* d_out,
                          void* stream);

/* ------------------------------------------------------------------------------------------------
 * M1-M5, L1  HRNet keypoint / line network
 * replaces HighResolutionNet.__init__/forward  src/models/hrnet/hrnet.py:255-355, 437-511,
 *          src/models/line/hrnet.py:30-249, HRNetHeatmap.forward src/models/hrnet/model.py:143-150
 * The descriptor carries the fields of the reference's model_config yaml
 * (src/models/hrnet/model_config/hrnet_w48.yaml).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int num_classes;            /* 58 keypoint net, 23 line net                                  */
    int stem_width;             /* 64                                                            */
    int upscale;                /* 2: bilinear x2 of all branches + stem concat; 1: line net     */
    int head_softmax;           /* 0 = LogSoftmax(dim=1) head, 1 = Softmax(dim=1) head           */
    int stage1_blocks;          /* number of Bottleneck blocks in layer1                         */
    int stage1_channels;        /* Bottleneck planes (output = 4x)                               */
    int num_modules[3];         /* stage2..4                                                     */
    int num_branches[3];        /* 2,3,4                                                         */
    int num_blocks[3];          /* BasicBlocks per branch (uniform per stage in every config)    */
    int num_channels[3][4];     /* branch widths per stage                                       */
} sncal_hrnet_desc;

typedef struct sncal_hrnet sncal_hrnet;

/* Build the execution plan (no weights yet).  dtype selects the arithmetic of the conv kernels:
 * SNCAL_BF16 = bf16 activations/weights with fp32 MFMA accumulation (fast path),
 * SNCAL_F32  = fp32 activations/weights on the exact-fp32 MFMA (parity path),
 * SNCAL_BF16X3 = the fp32 engine (fp32 activations, fp32 accumulation) with SPLIT-bf16 arithmetic in the 3x3 stride-1 convolutions of
 *              stages 2-4: x = hi + lo, w = hi + lo in bf16, hi.hi + hi.lo + lo.hi on the bf16 MFMA -- fp32-class results (|dlogp| ~ 5e-6
 *              against the exact engine, same keypoint indices) at a multiple of the fp32 MFMA rate,
 * SNCAL_FP8  = the bf16 engine with OCP e4m3 arithmetic (CDNA4 block-scaled MFMA, K = 64) in the wide 3x3 stride-1
 *              convolutions of stages 2-4 (BASELINE config 5); needs sncal_hrnet_calibrate_fp8 before the first forward. */
int sncal_hrnet_create(const sncal_hrnet_desc* desc, int dtype, sncal_hrnet** out);
void sncal_hrnet_destroy(sncal_hrnet* net);

/* The plan's conv units in the reference's registration order, with the state-dict prefixes the
 * host needs to fetch weights ("model.stage3.2.branches.1.0.conv1" / "...bn1"). */
int sncal_hrnet_num_convs(const sncal_hrnet* net);
int sncal_hrnet_conv_info(const sncal_hrnet* net, int idx, char* name, int name_cap, char* bn_name,
                          int bn_cap, int* cin, int* cout, int* ksize, int* stride, int* has_bias);
/* Give conv `idx` its weights: h_weight (Cout,Cin,k,k) fp32 host, h_scale/h_shift (Cout) fp32 host
 * = the eval-mode BatchNorm folded to y = conv(x)*scale + shift (scale NULL => 1). */
int sncal_hrnet_set_conv(sncal_hrnet* net, int idx, const float* h_weight, const float* h_scale,
                         const float* h_shift);
/* Pack + upload all weights (synchronous).  Every conv must have been set.
 * SNCAL_BF16X3 built with fp16 halves ("fp16x3", the default engine) carries every operand as fp16 hi + lo: exact to 2^-22 relative for
 * |v| in [2^-3, 65504], to 2^-25 absolute below, CLAMPED at +-65504 above.  The reference's predict() is plain fp32 with no such limits
 * (src/models/hrnet/metamodel.py:127-134), so the engine guards both ends of the clamp:
 *   here      SNCAL_ERR_RANGE when a folded weight (w * gamma / sqrt(var + eps)) exceeds 65504 in magnitude, is not finite, or when most of
 *             a layer's weight mass lies below 2^-14 (fp16's smallest normal: the halves keep fewer than 11 bits there); the message
 *             names the layer.  Callers fall back to SNCAL_F32 (the host mirror's load_model does so by itself, with a warning);
 *   forward   every kernel that splits activations counts the wavefronts that met |v| > 65504: sncal_hrnet_range_status below.
 *   balance   (fp16x3 only, default ON) before packing, finalize rebalances block-internal channels by exact powers of two: the tensor
 *             between conv1 and conv2 of a BasicBlock (conv1 / conv2, conv2 / conv3 of a Bottleneck; src/models/hrnet/hrnet.py:42-58, 79-99)
 *             has ONE consumer, so producer row c x 2^-l and consumer column c x 2^l is the same fp32 network bit for bit, and choosing l so
 *             that weight and activation are of one size keeps the fp16 halves at their full 22 bits on checkpoints whose BatchNorm scales
 *             spread over decades (csrc/hrnet.cpp equalize_blocks; only channels >= 2^4 away from an ordinary checkpoint's balance move).
 *             sncal_hrnet_set_equalize(net, 0) switches it off; sncal_hrnet_equalize runs the same step on its own (host only: no GPU
 *             needed) and reports the number of channels moved; sncal_hrnet_get_conv reads a conv's host parameters back (between
 *             sncal_hrnet_set_conv and sncal_hrnet_finalize, which releases them).  No reference counterpart (the reference is fp32).
 * A network handle is SINGLE-STREAM: forwards of one handle on two streams at once would share its work-ticket words. */
int sncal_hrnet_finalize(sncal_hrnet* net);
int sncal_hrnet_set_equalize(sncal_hrnet* net, int enable);
int sncal_hrnet_equalize(sncal_hrnet* net, int* moved);
int sncal_hrnet_get_conv(const sncal_hrnet* net, int idx, float* h_weight, float* h_scale, float* h_shift);
/* Range flag of the split-fp16 engine since the last clear: *overflow = wavefronts that split (and clamped) an activation beyond +-65504,
 * *nonfinite = workgroups of the input layout kernel that met a NaN / infinite frame value.  Returns SNCAL_OK when both are zero,
 * SNCAL_ERR_RANGE otherwise (sncal_last_error says what to do: such forwards are not the reference's fp32 result).  Synchronises
 * `stream`; clear != 0 re-arms the counters.  Engines without the clamp (SNCAL_F32, SNCAL_BF16, SNCAL_FP8, the bf16x3 build) report zeros.
 * No reference counterpart: the reference computes in fp32 (metamodel.py:127-134) and cannot overflow where this engine can. */
int sncal_hrnet_range_status(sncal_hrnet* net, unsigned* overflow, unsigned* nonfinite, int clear, void* stream);

/* Output spatial size and workspace bytes for a (B,3,H,W) input. */
int sncal_hrnet_output_size(const sncal_hrnet* net, int H, int W, int* out_h, int* out_w);
int sncal_hrnet_workspace(const sncal_hrnet* net, int B, int H, int W, size_t* bytes);

/* Forward.  d_x (B,3,H,W) fp32 NCHW in [0,1] (what make_submit.py:66-67 builds).
 *   d_heat  optional (B,num_classes,h,w) fp32 NCHW head output (log-softmax / softmax)
 *   d_kpts  optional (B,num_classes-1,3) fp32 decoded keypoints (keypoint net only; D1 semantics
 *           with size = (img_h,img_w))
 * At least one of d_heat / d_kpts must be non-NULL. */
int sncal_hrnet_forward(sncal_hrnet* net, const float* d_x, int B, int H, int W, float* d_heat,
                        float* d_kpts, int img_h, int img_w, void* d_ws, size_t ws_bytes,
                        void* stream);

/* C5 (BASELINE.json configs[4]: "HRNet-W48 fp8 (CDNA4 fp8 MFMA) 1920x1080 ... reproj-error tolerance sweep").  No reference
 * counterpart: HRNetMetaModel.predict is fp32 (src/models/hrnet/metamodel.py:127-134); the reference's AMP autocast exists
 * only in its train / val steps (:34, :67).
 * calibrate: one bf16 forward of d_x (B,3,H,W) that records max |x| of every tensor feeding an fp8-capable convolution
 *            (BasicBlock conv1 / conv2 of the 96 / 192 / 384-channel branches, src/models/hrnet/hrnet.py:42-58) ->
 *            per-tensor activation scales amax / 448; weights carry one scale per output channel.
 * set_fp8_layers: which of those convolutions run in fp8 -- "all", "none", or a comma list of stage2..stage4 and c<width>
 *            (a layer is selected when its stage AND its width are selected; an empty class selects all of it).  The
 *            tolerance sweep of tests/test_fp8_gpu.py walks this selection. */
int sncal_hrnet_calibrate_fp8(sncal_hrnet* net, const float* d_x, int B, int H, int W, void* d_ws, size_t ws_bytes, void* stream);
int sncal_hrnet_set_fp8_layers(sncal_hrnet* net, const char* spec);

/* Same forward from the frames as the reference's harness holds them BEFORE torchvision's ToTensor
 * (make_submit.py:62-66: cv2.imread -> BGR uint8 (H,W,3) -> ToTensor = float32 x/255, CHW): d_x (B,H,W,3) uint8.
 * Bit-identical to sncal_hrnet_forward on ToTensor's output; the input read is 3 bytes per pixel instead of 12
 * (at thousands of frames per second the float frames are the PCIe / HBM bottleneck). */
int sncal_hrnet_forward_u8(sncal_hrnet* net, const unsigned char* d_x, int B, int H, int W, float* d_heat,
                           float* d_kpts, int img_h, int img_w, void* d_ws, size_t ws_bytes,
                           void* stream);

/* Diagnostics (measurement only, no reference counterpart): per-kernel timing of forwards with HIP events
 * recorded on the caller's stream between the plan's launches.  Enable, run forwards, then read the
 * accumulated per-kernel-variant totals (the call synchronises the recorded events). */
typedef struct {
    char kernel[96];        /* e.g. "conv<bf16,k3,s1,NI4,MI6,G4>", "upsample_add", "softmax_nchw", "kp_decode" */
    double flops;           /* algorithmic FLOPs (2*MACs of the direct formulation) summed over launches  */
    double bytes;           /* algorithmic HBM bytes (inputs read once + outputs written once) summed     */
    double ms;              /* summed event-to-event time                                                */
    int launches;
} sncal_kernel_stat;
/* enable: 0 off; 1 time every launch; 2 time only the launches of the kernel variant that led the profile recorded so
 * far (needs a preceding mode-1 profile; the timing events cost ~4 us per launch, so a timed region that only needs
 * its dominant kernel's duration should not pay for the other ~150 launches). Every call clears the recorded profile. */
int sncal_hrnet_set_profiling(sncal_hrnet* net, int enable);
int sncal_hrnet_get_profile(sncal_hrnet* net, sncal_kernel_stat* out, int cap, int* count);

/* Plan introspection + taps (test instrumentation, no reference counterpart): the per-kernel parity tests
 * (tests/test_kernels_gpu.py) check EVERY launch of a forward -- each convolution of src/models/hrnet/hrnet.py:42-58, 79-99,
 * 183-246, 357-391, the fuse sums of :229-244, the head of :316-329, 489-510 -- against torch fp32 on the operands that launch
 * really read.  The plan is the flat op list the executor walks; tensors are NHWC slots of the caller's workspace.  A tap
 * copies one tensor of the FIRST sub-batch into a caller buffer (device-to-device, on the forward's stream) when the
 * executor passes the given op -- after the launch that ran it, also when that launch was a grouped / fused one issued at
 * an earlier op -- while the tensor is still alive.  Taps never change what is launched. */
typedef struct {
    int type;                  /* 0 input layout, 1 conv, 2 upsample_add, 3 softmax, 4 decode, 5 fused head                    */
    int active;                /* part of the plan at the current layout (three head formulations live side by side)         */
    int conv;                  /* conv unit (index as in sncal_hrnet_conv_info; >= num_convs: head-internal slice of last_layer.0) */
    char name[96];             /* its name, "" for non-conv ops                                                                */
    int cin, cout, ksize, stride, col_off;   /* col_off: first column of last_layer.0 of a head-internal slice               */
    int in, res, out, base;    /* tensor ids (-1 = none)                                                                       */
    int src[4], nsrc;          /* upsample_add sources                                                                         */
    int head_direct, head_src[5], head_nsrc, head_fold[2], head_nfold;
    int relu, out_coff, out_f32, fp8;        /* fp8: 1 = this conv runs in e4m3 at the current layout, 2 = in split bf16 on the two-team kernel (bf16x3), 3 = in split bf16 on the generic kernel */
    char kernel[96];           /* label of the launch that executed it in the last mode-1 profiled forward ("" = unknown or
                                  executed by the launch of an earlier op: grouped members, second conv of a fused block)      */
    int res_twin;              /* bf16x3: this conv reads its residual from the split twin of tensor `res`, not from its fp32 form */
} sncal_plan_op;
typedef struct {
    int C, H, W;               /* NHWC, per frame                                                                              */
    int dtype;                 /* 0 fp32, 1 bf16, 2 e4m3                                                                       */
    int twin;                  /* id of the tensor's e4m3 twin or -1                                                           */
    int alive;                 /* allocated at the current layout                                                              */
    float scale;               /* e4m3 twins: value = code * scale                                                             */
    size_t bytes;              /* of the first sub-batch (sub_batch frames)                                                    */
    int sub_batch;
} sncal_plan_tensor;
int sncal_hrnet_plan_num_ops(const sncal_hrnet* net);
int sncal_hrnet_plan_num_tensors(const sncal_hrnet* net);
/* Valid after sncal_hrnet_workspace / a forward at the shape of interest (the layout decides sizes, twins and head variant). */
int sncal_hrnet_plan_op(const sncal_hrnet* net, int idx, sncal_plan_op* out);
int sncal_hrnet_plan_tensor(const sncal_hrnet* net, int id, sncal_plan_tensor* out);
/* Register a tap (op_idx >= 0) or clear all taps (op_idx < 0).  d_dst must hold sncal_plan_tensor.bytes. */
int sncal_hrnet_plan_tap(sncal_hrnet* net, int op_idx, int tensor_id, void* d_dst);

/* ------------------------------------------------------------------------------------------------
 * S0-S11  camera solve (batched, one wavefront per frame)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {              /* one solved camera; mirrors baseline/camera.py:79-90 attributes     */
    double position[3];
    double rotation[9];       /* row-major R, X_cam = R (X_world - position)                       */
    double fx, fy;            /* calibration[0,0], calibration[1,1]                                */
    double cx, cy;            /* calibration[0,2], [1,2] used by solve_pnp/refine (Q3)             */
    double rmse;              /* Camera.projection_rmse of the selected camera (mean L2, px)       */
    int32_t status;           /* 0 = no camera (reference returns None), >0 = which branch produced it */
    int32_t n_points;         /* matched points used                                               */
} sncal_camera;

enum {                        /* sncal_camera.status                                               */
    SNCAL_CAM_NONE = 0,
    SNCAL_CAM_ORIGINAL = 1,       /* original_voter calibrate path  prediction.py:396-433          */
    SNCAL_CAM_ORIGINAL_HOM = 2,   /* original_voter homography fallback  :434-436                  */
    SNCAL_CAM_VOTER_REL = 3, SNCAL_CAM_VOTER_ACC = 4, SNCAL_CAM_VOTER_ALL = 5,
    SNCAL_CAM_VOTER_GROUND = 6,   /* voter winners  :293-323                                       */
    SNCAL_CAM_VOTER_HOM = 7       /* voter homography fallback  :327-329                           */
};

#define SNCAL_MAX_CONF_THRESHS 16   /* the reference loops over any number of thresholds (prediction.py:245-257); make_submit.py passes 3 */
typedef struct {              /* CameraCreator kwargs, make_submit.py:45-50                        */
    int algorithm;            /* 0 iterative_voter, 1 original_voter, 2 voter,
                                 3 opencv_calibration, 4 opencv_calibration_multiplane            */
    int n_conf_threshs;
    double conf_thresh;       /* compared in double, like numpy's float32-scalar > python-float      */
    double conf_threshs[SNCAL_MAX_CONF_THRESHS];   /* iterative_voter's thresholds in the order they are tried; the first n_conf_threshs count */
    double max_rmse, max_rmse_rel;
    int min_points, min_points_per_plane, min_points_for_refinement, reliable_thresh;
    double min_focal_length;
    int img_w, img_h;
    int lm_schedule;          /* 0 (default) = the minimisers follow OpenCV 4.7's own schedules as far as they are known:
                                 LMSolver for solvePnPRefineLM (lambda_0 = 1, D fixed, gain ratio, stop 1e-5), CvLevMarq for the
                                 extrinsics refits (20 iterations / FLT_EPSILON) and calibrateCamera's joint fit (30 / DBL_EPSILON);
                                 1 = run every minimiser to convergence (the build's round-1/2 specification).  OpenCV parity is
                                 unpinned either way (cv2 is not installable offline).  ABI note: 0 became the OpenCV schedules in
                                 round 3 -- a caller that zero-initialises this struct gets them; set 1 for the round-1/2 behaviour */
    int refine_max_iters;     /* iteration cap of Camera.refine_camera's LM under lm_schedule 0; 0 = 20000 = what the reference passes
                                 (camera.py:116; rounds 1-3 defaulted to 200).  Well-posed frames stop on the 1e-5 step test within
                                 ~25 iterations; the runs that used to need the cap were voter candidates calibrated to f ~ 0.04 px,
                                 which good_camera discards whatever their pose: those are no longer refined at all              */
} sncal_voter_cfg;

/* Camera.refine_camera  baseline/camera.py:105-119 (cv.solvePnPRefineLM, K fixed, 6-DoF pose LM; LMSolver's schedule, see
 * sncal_voter_cfg.lm_schedule; max_iters <= 0 / eps <= 0 select the reference's (20000, 1e-5), see sncal_voter_cfg.refine_max_iters; the environment variable
 * SNCAL_SOLVE_SCHEDULE=converged switches this entry and sncal_solve_pnp to the run-to-convergence minimisers).
 *   d_K (B,4) fx,fy,cx,cy   d_pts3d (B,N,3) fp64   d_pts2d (B,N,2) fp64   d_npts (B) int32
 *   d_rt (B,12) in/out: rotation row-major (9) + position (3)   d_rmse (B) out mean-L2 px */
int sncal_pnp_refine_lm(const double* d_K, const double* d_pts3d, const double* d_pts2d,
                        const int32_t* d_npts, int B, int N, double* d_rt, double* d_rmse,
                        int max_iters, double eps, void* stream);

/* Camera.solve_pnp  baseline/camera.py:92-103 (cv.solvePnPRansac + Rodrigues), same layouts. */
int sncal_solve_pnp(const double* d_K, const double* d_pts3d, const double* d_pts2d,
                    const int32_t* d_npts, int B, int N, double* d_rt, void* stream);

/* CameraCreator.__call__  src/models/hrnet/prediction.py:130-136 with every algorithm of :90-96.
 *   d_kpts (B,57,3) fp32 decoded keypoints   d_line_pts (B,30,3) fp32 [x,y,valid] or NULL
 *   d_out (B) sncal_camera.  Never fails per frame: status 0 == the reference's `None`.
 * Asynchronous on `stream`.  iterative_voter runs as stages on that stream: the original_voter pass (one wavefront per frame and half:
 * its homography camera and its calibrated camera), then, for the frames it left without a camera, one wavefront per threshold x voter
 * camera and a selection in the reference's threshold order (prediction.py:250-256).
 * Scratch (per-frame slots of the two stages):
 *   sncal_calibrate_ws   the library's convention: the caller provides d_ws of at least sncal_calibrate_workspace(B, cfg) bytes
 *                        (16-byte aligned device memory, free to reuse once the work enqueued on `stream` has passed it); nothing
 *                        persists between calls.  The host mirror (CameraCreator.solve_device) uses this form.
 *   sncal_calibrate      convenience form without a workspace argument: the library keeps ONE scratch block per (device, stream),
 *                        hipMalloc'ed at the stream's first call, grown when a larger batch arrives, and reused by later calls on that
 *                        stream (stream order makes that safe; rounds 3-4 took it from hipMallocAsync per call, which made the HOST wait
 *                        for other streams' solves).  The block is released by sncal_stream_destroy(stream) for streams created here, and
 *                        by sncal_shutdown() for all streams -- call it before destroying a stream created elsewhere whose handle value
 *                        may be reused.
 * RANSAC sampling -- a KNOWN DEVIATION from the reference's numbers: cv2.findHomography (src/datatools/ellipse.py:497, used through
 * prediction.py:487-500) and cv.solvePnPRansac (baseline/camera.py:100-101) draw their minimal samples from OpenCV's fixed-seed RNG
 * (a multiply-with-carry generator); this library draws the four indices of hypothesis h from its own counter hash of (h, draw) --
 * the same 128 samples for every frame with the same point count, evaluated lane-parallel -- and takes the hypothesis with the most
 * inliers (ties: smaller squared error, then lower h) where OpenCV keeps the first best and adapts its iteration count.  Both are
 * deterministic and both refit on the winner's inliers, so on frames whose keypoints are all inliers of one model the refit -- hence the camera -- does not
 * depend on the draw; on frames with gross outliers a different draw may find a different inlier set and therefore a different camera
 * than OpenCV would.  Unmeasurable here (no cv2 on this image), stated so that nobody takes it for parity. */
int sncal_calibrate(const float* d_kpts, const float* d_line_pts, int B, const sncal_voter_cfg* cfg,
                    sncal_camera* d_out, void* stream);
int sncal_calibrate_workspace(int B, const sncal_voter_cfg* cfg, size_t* bytes);
int sncal_calibrate_ws(const float* d_kpts, const float* d_line_pts, int B, const sncal_voter_cfg* cfg,
                       sncal_camera* d_out, void* d_ws, size_t ws_bytes, void* stream);
/* Free what the library holds outside its handles (the scratch blocks of sncal_calibrate above); synchronises those streams. */
int sncal_shutdown(void);

/* ------------------------------------------------------------------------------------------------
 * H2 / N2  batched camera evaluation (accuracy@t of the SoccerNet calibration benchmark)
 * replaces get_polylines / distance_to_polyline / evaluate_camera_prediction and the mirrored-label accuracy choice
 *          baseline/evaluate_camera.py:14-229, 293-320 (the inner loop of src/models/hrnet/metrics.py:97-229)
 *   d_cams (B) solved cameras (status 0 = missed frame); d_field (n_pts,3) fp64 sampled pitch model in class order,
 *   d_class_start (n_cls+1) int32, d_mirror (n_cls) int32 = index of the left/right-symmetric class
 *   (SoccerPitch.symetric_classes); d_gt (B,n_cls,max_gt,2) fp64 annotated points in pixels with counts
 *   d_gt_cnt (B,n_cls); d_gt_extra (B) = annotated classes outside the pitch model (always false negatives).
 *   d_out (B,12) fp32: confusion [TP,FP,FN,0] with plain labels, the same with mirrored labels, accuracy plain,
 *   accuracy mirrored, chosen pass (1 or 2), evaluated flag (0 for a missed frame).
 * The principal point is (img_w/2, img_h/2) as in Camera.from_json_parameters / project_point.
 * ---------------------------------------------------------------------------------------------- */
int sncal_evaluate_cameras(const sncal_camera* d_cams, int B, const double* d_field, const int* d_class_start,
                           const int* d_mirror, int n_cls, const double* d_gt, const int* d_gt_cnt,
                           const int* d_gt_extra, int max_gt, double threshold, int img_w, int img_h,
                           float* d_out, void* stream);
/* The same evaluation with the per-class outputs of evaluate_camera_prediction (evaluate_camera.py:172-226), both
 * optional:  d_err (B,2,n_cls,max_gt) fp64 = `dict_errors`: distance of annotated point k to the predicted polyline
 * of class c, [b][0] plain labels, [b][1] mirrored labels (the points come from class mirror[c]); NaN = no such
 * point or class not predicted.  d_class_conf (B,2,n_cls,4) int32 = `per_class_confusion` entries
 * {[0,0] points within the threshold, [0,1] points beyond it, [1,0] points of an annotated class that was not
 * predicted, flag: class predicted but not annotated -- the reference books 2 (lines) or 9 (circles) at [0,1]}. */
int sncal_evaluate_cameras_detail(const sncal_camera* d_cams, int B, const double* d_field, const int* d_class_start,
                                  const int* d_mirror, int n_cls, const double* d_gt, const int* d_gt_cnt,
                                  const int* d_gt_extra, int max_gt, double threshold, int img_w, int img_h,
                                  float* d_out, double* d_err, int* d_class_conf, void* stream);

/* ------------------------------------------------------------------------------------------------
 * N3  input stage: JPEG bytes -> BGR uint8 frames on the device
 * replaces cv2.imread(path)            src/utils/make_submit.py:62, src/utils/export_line_result.py:176
 *          (opencv-python==4.7.0.72, requirements.txt:5 -> libjpeg-turbo defaults: JDCT_ISLOW, fancy upsampling)
 * The serial part of a JPEG -- marker parsing and Huffman decoding -- runs on host threads into pinned memory;
 * everything with pixel parallelism runs on the device, bit-exactly as libjpeg-turbo computes it: dequantisation +
 * jidctint.c jpeg_idct_islow, jdsample.c h2v2/h2v1 fancy upsampling, jdcolor.c YCbCr->RGB.  The output (B,H,W,3) BGR
 * is the input of sncal_hrnet_forward_u8.
 * Supported: 8-bit sequential DCT (SOF0/SOF1), Huffman, one interleaved scan, grey or YCbCr 4:4:4 / 4:2:2 / 4:2:0,
 * restart intervals.  Progressive / arithmetic / CMYK / 4:4:0 / multi-scan -> SNCAL_ERR_UNSUPPORTED; damaged
 * streams -> SNCAL_ERR_ARG; the message names the frame.  A file whose EXIF orientation tag asks for a rotation / flip (cv2.imread applies it)
 * -> SNCAL_ERR_UNSUPPORTED as well: the decoder never returns pixels cv2.imread would not.
 * ---------------------------------------------------------------------------------------------- */
typedef struct sncal_jpeg sncal_jpeg;
typedef struct {
    int32_t width, height;
    int32_t components;          /* 1 (grey) or 3 (YCbCr) */
    int32_t h_samp, v_samp;      /* luma sampling factors: 1x1 = 4:4:4, 2x1 = 4:2:2, 2x2 = 4:2:0 */
    int32_t restart_interval;    /* MCUs, 0 = none */
    int32_t blocks[3];           /* 8x8 coefficient blocks per component (MCU padded) */
} sncal_jpeg_info;

/* Host only (no GPU needed): parse the headers of one JPEG. */
int sncal_jpeg_probe(const unsigned char* data, size_t len, sncal_jpeg_info* info);
/* Host only: headers + Huffman decode of one JPEG into quantised coefficient blocks, int16, natural (row-major)
 * order, component after component, each (block_rows, block_cols, 64) -- the layout the device kernels read.
 * `cap` = capacity of `coef` in int16 elements; info->blocks tells how much was written. */
int sncal_jpeg_entropy_decode(const unsigned char* data, size_t len, int16_t* coef, size_t cap,
                              sncal_jpeg_info* info);
/* Decoder for batches of up to max_batch frames of exactly height x width pixels (the reference stacks its frames
 * into one tensor, make_submit.py:63).  n_threads host workers for the entropy decode (0 = hardware concurrency,
 * capped at 32).  Owns pinned staging (double-buffered) and the device coefficient / sample-plane buffers. */
int sncal_jpeg_create(int max_batch, int height, int width, int n_threads, sncal_jpeg** out);
void sncal_jpeg_destroy(sncal_jpeg* dec);
/* data[b], len[b]: host pointers to B encoded frames.  d_bgr: (B,height,width,3) uint8 on the device.  The host
 * part runs inside the call; the copy and the two kernels are enqueued on `stream`. */
int sncal_jpeg_decode(sncal_jpeg* dec, const unsigned char* const* data, const size_t* len, int B,
                      unsigned char* d_bgr, void* stream);

/* ------------------------------------------------------------------------------------------------
 * N4 (heatmap half)  training-target synthesis
 * replaces HRNetLoss.create_target = create_heatmaps + background channel     src/models/hrnet/loss.py:7-52, 81-87
 *   d_kpts (B,N,3) fp32 [x, y, visibility] in heatmap pixels (already divided by the loss stride, loss.py:92)
 *   d_out  (B,N+1,h,w) fp32: N separable amplitude-1 Gaussians (zero where the point is not "visible" by the
 *          reference's own test any(keypoints == 1, dim=-1), loss.py:49) and 1 - max over them as channel N.
 * ---------------------------------------------------------------------------------------------- */
int sncal_create_target(const float* d_kpts, int B, int N, float sigma, int h, int w, float* d_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pipeline plumbing: a stream confined to `cus_per_xcd` compute units of each XCD (for the camera solves)
 * replaces the 16-process CPU pool of make_submit.py:25,53-54,69 (ProcessPoolExecutor workers beside the GPU loop): the solves of
 * up to four batches run beside the network on these streams; what they may occupy is bounded by the mask instead of by a process
 * count.  BLOCKING stream (synchronises with the legacy null stream): see csrc/api.cpp.  Destroy with sncal_stream_destroy.
 * ---------------------------------------------------------------------------------------------- */
int sncal_stream_create_cu_mask(int cus_per_xcd, void** stream);
int sncal_stream_destroy(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SNCAL_H */
