/*
 * ssp_hip.h - C ABI of the MI355X-native Semantic-SuperPoint pair-training path.
 *
 * Every pointer named *_dev is a DEVICE pointer owned by the caller (PyTorch tensors in the
 * Python shims); the library never frees caller memory and keeps no host threads.  All entry
 * points return 0 on success or a negative code; ssp_last_error() gives the message.  `stream`
 * is a hipStream_t passed as void* (0 = default stream).  One handle per GPU / stream.
 *
 * Reference interfaces replaced (paths relative to the reference repository):
 *   ssp_forward      <- models/SuperPointNet_gauss2.py:42-69, models/SuperPointNet_gauss2_ssmall.py:58-99
 *                       (+ models/unet_parts.py:10-48)
 *   ssp_backward     <- autograd of the above (loss.backward(), Train_model_heatmap_all.py:407)
 *   ssp_pair_step    <- Train_model_heatmap_all.py:195-413 (train_val_sample: 2 forwards, labels2Dto3D
 *                       utils/utils.py:408-440, getMasks Train_model_frontend_all.py:373-386,
 *                       detector_loss :155-179, sem_loss :181-193, batch_descriptor_loss_sparse
 *                       utils/loss_functions/sparse_loss.py:267-284, MultiTaskLoss :46-77, backward)
 *                       or, with ssp_pair_inputs.dense_loss, the dense descriptor_loss utils/utils.py:779-893
 *   ssp_adam_step    <- optimizer.step() of Train_model_frontend_all.py:183-198 (Adam, constant LR)
 *   ssp_sample_indices <- the stochastic half of descriptor_loss_sparse (sparse_loss.py:184-246,
 *                       correspondence_finder.py:191-320), device RNG
 */
#ifndef SSP_HIP_H
#define SSP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ssp_handle ssp_handle;

enum { SSP_ARCH_GAUSS2 = 0, SSP_ARCH_GAUSS2_SSMALL = 1 };
enum { SSP_NREP = 32 }; /* replicas of each fp64 statistics accumulator (spreads same-address atomics) */

typedef struct {
  int arch;      /* SSP_ARCH_* */
  int n_classes; /* 133 for the ssmall seg head; ignored for gauss2 */
  int max_batch; /* largest N of any forward (1..1024); ssp_pair_step additionally requires batch <= 128
                    (per-image accumulator arrays of the sparse descriptor loss, SSP_MAX_PAIRS) */
  int height;    /* H, multiple of 8 */
  int width;     /* W, multiple of 8 */
  int n_match;   /* num_matching_attempts (1000) */
  int n_non;     /* num_masked_non_matches_per_match (100) */
  int dense_loss; /* 1: reserve the [B, cells, cells] coefficient matrix of the dense descriptor loss
                     (model.dense_loss.enable, utils/utils.py:779-893) in the workspace */
} ssp_config;

/* Device buffers bound once after create (all float32 unless noted).
 * params/grads/adam_m/adam_v: [n_params + 3]  (net.parameters() order = state_dict order without
 * buffers, OIHW conv weights; the 3 trailing floats are MultiTaskLoss.eta).
 * bn_running: [2 * n_bn_channels] = all running_mean (layer order) then all running_var.
 * num_batches_tracked: int64 [n_bn_layers]. */
typedef struct {
  float* params_dev;
  float* grads_dev;
  float* adam_m_dev;
  float* adam_v_dev;
  float* bn_running_dev;
  int64_t* num_batches_tracked_dev;
  void* workspace_dev;
  size_t workspace_bytes;
} ssp_buffers;

/* One micro-batch of pairs (Train_model_heatmap_all.py:212-251 `sample`). NCHW with C==1.
 * warped_image_dev == NULL selects the SINGLE-VIEW step of `data.warped_pair.enable: false` (Train_model_heatmap_all.py:207,
 * 237-262, 330-332; configs/magicpoint_shapes_pair.yaml): one forward, detector (+ segmentation) loss of the image only,
 * loss_det_warp = loss_sem_warp = 0; the other warped_* pointers and homographies_dev are then ignored and lambda_loss must
 * be 0 (the reference asserts "need a pair of images"). */
typedef struct {
  int batch;
  const float* image_dev;             /* [B,1,H,W] */
  const float* warped_image_dev;      /* [B,1,H,W], or NULL: single-view step */
  const float* labels_dev;            /* [B,1,H,W] labels_2D(_gaussian) */
  const float* warped_labels_dev;     /* [B,1,H,W] */
  const float* valid_mask_dev;        /* [B,1,H,W] */
  const float* warped_valid_mask_dev; /* [B,1,H,W] */
  const float* homographies_dev;      /* [B,3,3] normalised coords, image -> warped */
  const int64_t* semantic_dev;        /* [B,H,W] or NULL */
  const int64_t* warped_semantic_dev; /* [B,H,W] or NULL */
  /* sparse-loss indices, either given (parity mode) or sampled on device when match_a_dev==NULL */
  const int32_t* match_a_dev;    /* [B,n_match] cell index u+v*Wc in image a */
  const int32_t* match_b_dev;    /* [B,n_match] cell index in image b */
  const int32_t* nonmatch_b_dev; /* [B,n_match*n_non] cell index in image b */
  uint64_t seed;                 /* device sampler seed (used when match_a_dev == NULL) */
  float lambda_loss;             /* model.lambda_loss; 0 disables the descriptor loss */
  float lamda_d;                 /* sparse_loss.params.lamda_d */
  int multi_task;                /* model.multi_task_loss */
  int train;                     /* 1: accumulate gradients; 0: forward + losses only */
  /* dense descriptor loss instead of the sparse one (Train_model_heatmap_all.py:131-137,348-350): no indices needed.
   * dense_lamda_d: descriptor_loss's `lamda_d` (250: the shipped configs spell it `lambda_d`, which the reference
   * swallows in **config); descriptor_dist: dense_loss.params.descriptor_dist (4) */
  int dense_loss;
  float dense_lamda_d;
  float descriptor_dist;
  /* optional [B,3,3]: the CELL-space homographies scale_homography_torch(H, (Hc, Wc)) (utils/homographies.py:270-276)
   * computed by the caller on the host with the reference's own fp32 op sequence; used by the captured device sampler
   * (ssp_pair_step_graph) - see ssp_sample_indices_cell */
  const float* cell_homographies_dev;
  /* sparse_loss.params.method / dist (sparse_loss.py:76-77, pixelwise_contrastive_loss.py:140): 0 / 0 = "2d" / "cos", what every
   * shipped config selects; method 1 = "1d" (matches by index_select at the cell), dist 1 = "euclidean" (squared distance of the
   * matches, (max(0, ||a - b|| - 0.2))^2 of the non-matches - nearly every non-match of unit descriptors is then a hard one) */
  int sparse_method;
  int sparse_dist;
} ssp_pair_inputs;

/* indices into the float scalars[SSP_N_SCALARS] array filled by ssp_pair_step (the reference's
 * scalar_dict, Train_model_heatmap_all.py:415-441) */
enum {
  SSP_S_LOSS = 0, SSP_S_LOSS_DET, SSP_S_LOSS_DET_WARP, SSP_S_LOSS_DESC, SSP_S_LOSS_SEM, SSP_S_LOSS_SEM_WARP,
  SSP_S_POSITIVE_DIST, SSP_S_NEGATIVE_DIST, SSP_S_ETA_DET, SSP_S_ETA_DESC, SSP_S_ETA_SEM,
  SSP_N_SCALARS = 16
};

const char* ssp_last_error(void);
/* sha256 of the library's source files (csrc/*.hip, csrc/*.hip.h, include/ssp_hip.h) at build time: hipbuild.source_id() */
const char* ssp_build_id(void);
/* Bit-reproducible accumulation (process-wide; also SSP_DETERMINISTIC=1 in the environment): floating-point atomics of the path
 * become order-independent - fp64 accumulators take addends rounded to a fixed quantum (exact sums), fp32 scatter targets go
 * through 64-bit fixed-point shadows (csrc/det.hip.h).  Switch it BEFORE ssp_bind: the shadows are allocated there.  The
 * reference has no counterpart (torch.use_deterministic_algorithms is never set: train4.py). */
int ssp_set_deterministic(int on);
int ssp_get_deterministic(void);
/* Measurement aid (no counterpart in the reference): the shader clock in MHz that the device sustains while every CU runs
 * fp32 matrix-core instructions for ~`ms` milliseconds on `stream` (blocking).  bench.py reports it before and after the timed
 * region: the boxes of a pool differ in the clock their power controller grants. */
int ssp_clock_probe(float ms, double* mhz_out, void* stream);
int ssp_create(const ssp_config* cfg, ssp_handle** out);
void ssp_destroy(ssp_handle* h);
size_t ssp_param_count(const ssp_handle* h);       /* net parameters (without eta) */
size_t ssp_bn_channel_count(const ssp_handle* h);  /* sum of C over BatchNorm layers */
int ssp_bn_layer_count(const ssp_handle* h);
size_t ssp_workspace_bytes(const ssp_handle* h);
int ssp_bind(ssp_handle* h, const ssp_buffers* b, void* stream);

/* forward of one batch into activation slot 0/1; outputs (may be NULL) are NCHW like the reference:
 * semi [N,65,H/8,W/8], desc [N,256,H/8,W/8], sem [N,n_classes,H,W]. train: 1 = batch statistics +
 * running-stat update (module default), 0 = eval() (running statistics). */
int ssp_forward(ssp_handle* h, int slot, const float* x_dev, int n, int height, int width, int train,
                float* semi_dev, float* desc_dev, float* sem_dev, void* stream);
/* backward of slot given dL/d(outputs) in NCHW (NULL = zero); ACCUMULATES into grads_dev. */
int ssp_backward(ssp_handle* h, int slot, const float* dsemi_dev, const float* ddesc_dev, const float* dsem_dev,
                 void* stream);
int ssp_zero_grad(ssp_handle* h, void* stream);
int ssp_pair_step(ssp_handle* h, const ssp_pair_inputs* in, float* scalars_dev, void* stream);
int ssp_adam_step(ssp_handle* h, float lr, int step, void* stream);
/* the same Adam step on grads * grad_scale (data parallel: 1 / world_size after an all-reduce SUM; grads_dev is left
 * untouched, so gradient accumulation over micro-batches keeps working) */
int ssp_adam_step_scaled(ssp_handle* h, float lr, int step, float grad_scale, void* stream);

/* ---- data-parallel overlap ("all-reduce overlapped with encoder backward") ------------------
 * ssp_pair_step_phase: phase 0 = ssp_pair_step.  Phase 1 runs the step up to the point where every gradient from
 * ssp_grad_early_offset(h) to the end of the flat vector (encoder layers >= 2, all heads, eta: 97.7 % of the bytes) is
 * FINAL; phase 2 runs the rest of the backward pass (the two 240x320 layers, 40 % of the backward time).  Between the
 * two calls the host starts the all-reduce of the early bucket on another stream; it overlaps with phase 2. */
int ssp_pair_step_phase(ssp_handle* h, const ssp_pair_inputs* in, float* scalars_dev, int phase, void* stream);
size_t ssp_grad_early_offset(const ssp_handle* h); /* first float of the early-final gradient bucket */

/* ---- hipGraph form of the pair step (north star: "one graph per image pair") --------------------------------------
 * The first call with a given (inputs, phase, algorithm) signature captures the launches of ssp_pair_step_phase into a
 * hipGraph on `stream` (must not be the default stream); later calls replay it with ONE hipGraphLaunch.  With
 * sample_indices != 0 the device index sampler (ssp_sample_indices into in->match_a/match_b/nonmatch_b_dev, which must
 * be caller-owned buffers of the usual sizes) is part of the graph; its seed `in->seed` is kept in device memory and may
 * change from call to call.  Everything else of `in` is part of the signature.  Up to 16 graphs are cached per handle;
 * ssp_bind drops them.  ssp_profile_enable must be off (hipEvents cannot be recorded into the capture). */
int ssp_pair_step_graph(ssp_handle* h, const ssp_pair_inputs* in, float* scalars_dev, int phase, int sample_indices,
                        void* stream);
int ssp_sample_indices(ssp_handle* h, const float* homographies_dev, int batch, uint64_t seed, int32_t* match_a_dev,
                       int32_t* match_b_dev, int32_t* nonmatch_b_dev, void* stream);
/* Index parity: every kernel that warps integer coordinates takes the homography in the space of those coordinates.  With
 * the *_cell / *_px entry points the caller passes T^-1 H T computed on the HOST exactly like the reference
 * (torch.inverse(trans) @ H @ trans: scale_homography_torch / homography_scaling_torch) and the device applies the
 * reference's own fp32 operation order (fma chain of torch's CPU matmul, correctly rounded division, round half to even):
 * warped integer indices are then bit-identical to the reference's.  The plain entry points derive T^-1 H T analytically
 * on the device; coordinates within ~1e-4 of a .5 boundary may then round the other way. */
int ssp_sample_indices_cell(ssp_handle* h, const float* cell_homographies_dev, int batch, uint64_t seed, int32_t* match_a_dev,
                            int32_t* match_b_dev, int32_t* nonmatch_b_dev, void* stream);

/* timing hook for bench.py: when enabled, every launch of the tagged kernel family is bracketed by
 * hipEvents on `stream`; ssp_profile_read returns accumulated milliseconds, launches and FLOPs. */
enum { SSP_PROF_NONE = 0, SSP_PROF_CONV3X3_FWD = 1, SSP_PROF_CONV3X3_DGRAD = 2, SSP_PROF_CONV3X3_WGRAD = 3,
       SSP_PROF_CONV_BIG_FWD = 4, SSP_PROF_CONV3X3_ALL = 5 /* 3x3 forward + data-gradient launches */,
       SSP_PROF_CONV3X3_EVERY = 6 /* 3x3 forward + data-gradient + weight-gradient launches */ };
int ssp_profile_enable(ssp_handle* h, int family);
int ssp_profile_read(ssp_handle* h, double* ms, int64_t* launches, double* flops, double* bytes);
/* The same measurement split by the KERNEL that ran each tagged launch (bench.py's per-kernel roofline entries). */
enum { SSP_PROF_K_CONV_WINO4 = 0 /* conv_wino4_kernel, Winograd F(4x4,3x3) */, SSP_PROF_K_CONV_WINO_PIPE = 1, SSP_PROF_K_CONV_WINO_P2 = 2,
       SSP_PROF_K_WGRAD_WINO = 3 /* wgrad_wino_kernel, F(3x3,2x2) */, SSP_PROF_K_WGRAD_WINO4 = 4 /* wgrad_wino4_kernel, F(3x3,4x4) */,
       SSP_PROF_K_OTHER = 5 /* direct implicit GEMM, bf16-operand Winograd kernels */,
       SSP_PROF_K_CONV_BF16 = 6 /* conv_bf16_kernel (forward + data gradient of the bf16 path) */,
       SSP_PROF_K_WGRAD_BF16 = 7 /* wgrad_bf16_kernel */, SSP_PROF_K_COUNT = 8 };
/* Suspends (paused != 0) / resumes the bracketing without resetting the counters: bench.py brackets every n-th step of the
 * timed region only (the event records of ~30 launches per step cost ~1.7 % of the step when every step carries them). */
int ssp_profile_pause(ssp_handle* h, int paused);
int ssp_profile_read_kernel(ssp_handle* h, int kernel, double* ms, int64_t* launches, double* flops, double* executed_flops,
                            double* bytes);
/* FLOPs the tagged launches since ssp_profile_enable EXECUTED on the matrix cores (ssp_profile_read's `flops` are the
 * algorithmic, direct-convolution FLOPs): x 1/4 for a Winograd F(4x4,3x3) launch, x 16/36 for F(2x2,3x3), x 1 otherwise. */
int ssp_profile_read_executed(ssp_handle* h, double* executed_flops);

/* ---- operator-level entry points (used by the unit parity tests; same kernels as above) ---- */
/* 3x3 / 1x1 convolution, NHWC fp32, stride 1, "same" padding, weights OIHW (reference layout).
 * in_mode: 0 raw input, 1 input = relu(in*scale+shift), 2 = maxpool2(relu(in*scale+shift)) where `in`
 * is [N,2H,2W,Cin]. stats_dev (double [SSP_NREP][2*Cout]: partial sum, sumsq replicas; zeroed by the caller)
 * may be NULL. */
int ssp_op_conv(const float* in_dev, const float* w_oihw_dev, const float* bias_dev, float* out_dev, int n, int h,
                int w, int cin, int cout, int ksize, int in_mode, const float* in_scale_dev,
                const float* in_shift_dev, double* stats_dev, int transpose_flip, void* workspace_dev,
                size_t workspace_bytes, void* stream);
int ssp_op_conv_wgrad(const float* in_dev, const float* dout_dev, float* dw_oihw_dev, int n, int h, int w, int cin,
                      int cout, int ksize, int in_mode, const float* in_scale_dev, const float* in_shift_dev,
                      void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---- bf16 path (conv algorithm 12, BASELINE configs[3]): bf16 NHWC activations in HBM, v_mfma_f32_32x32x16_bf16 ----
 * Convolution of models/unet_parts.py:14-21 as a direct implicit GEMM (csrc/conv_bf16.hip.h).  in: bf16 (fp32 when in_f32)
 * NHWC [n,h,w,cin]; w: OIHW fp32 (rounded to bf16 when packed; transpose_flip: O = cin, the data-gradient convolution);
 * in_mode 1: operand = bf16(relu(in * scale + shift)); out: bf16 (fp32 when out_f32) NHWC [n,h,w,cout] = round(acc + bias);
 * stats (optional, double [SSP_NREP][2 cout], zeroed by the caller): sum / sum of squares of the STORED output;
 * pool_out (optional, bf16 [n,h/2,w/2,cout]): per-channel max (pool_gamma >= 0) / min (< 0) of every 2x2 window of out.
 * workspace: >= ceil(cout/64) * ceil(cin/32) * ksize^2 * 4096 bytes (the packed bf16 weight image). */
int ssp_op_conv_bf16(const void* in_dev, const float* w_oihw_dev, const float* bias_dev, void* out_dev, int n, int h, int w,
                     int cin, int cout, int ksize, int in_mode, const float* in_scale_dev, const float* in_shift_dev,
                     double* stats_dev, int transpose_flip, int in_f32, int out_f32, void* pool_out_dev,
                     const float* pool_gamma_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* Weight gradient of the same convolution (csrc/wgrad_bf16.hip.h): x bf16 NHWC [n,h,w,cin] (in_mode 1: operand =
 * bf16(relu(x * scale + shift))), dy bf16 (fp32 when dy_f32, rounded to bf16 on load) NHWC [n,h,w,cout]; the fp32 OIHW gradient is
 * ACCUMULATED into dw_oihw_dev.  workspace: partial slabs, >= ceil(cin/64) * ceil(cout/64) * ksize^2 * 16 KiB (more = more
 * workgroups, up to 2 per CU). */
int ssp_op_conv_wgrad_bf16(const void* x_dev, const void* dy_dev, float* dw_oihw_dev, int n, int h, int w, int cin, int cout,
                           int ksize, int in_mode, const float* in_scale_dev, const float* in_shift_dev, int dy_f32,
                           void* workspace_dev, size_t workspace_bytes, void* stream);

/* ssp_op_bn_bwd_strided on bf16 tensors (y, dout, dy: bf16 NHWC, channel stride cs; ReLU layers only). */
int ssp_op_bn_bwd_bf16(const void* y_dev, const void* dout_dev, const float* gamma_dev, const float* stats4_dev, void* dy_dev,
                       float* dgamma_dev, float* dbeta_dev, float* dbias_dev, double* sums_dev, int n, int h, int w, int c, int cs,
                       int relu, int pool, void* stream);

/* labels2Dto3D (utils/utils.py:408-440, add_dustbin=True) -> target [B,65,H/8,W/8] NCHW and getMasks
 * (Train_model_frontend_all.py:373-386) -> cellmask [B,H/8,W/8]; either pair of pointers may be NULL. */
int ssp_op_labels(const float* labels2d_dev, const float* mask2d_dev, float* target_dev, float* cellmask_dev, int b,
                  int h, int w, void* stream);

/* detector_loss (Train_model_heatmap_all.py:155-179, softmax branch) as an operator: semi NHWC [b][h/8*w/8][cs] (65 logits,
 * channel stride cs >= 65), labels2d / mask2d [b,1,h,w] -> loss_dev[0] and (optional) d loss / d semi in the same layout.
 * scratch: 64 KiB + 4 bytes per cell. */
int ssp_op_detector_loss(const float* semi_nhwc_dev, int cs, const float* labels2d_dev, const float* mask2d_dev, int b, int h,
                         int w, void* scratch_dev, size_t scratch_bytes, float* loss_dev, float* dsemi_nhwc_dev, void* stream);

/* sem_loss (Train_model_heatmap_all.py:181-193: CrossEntropyLoss(ignore_index = n_classes) of the bilinear upsample,
 * align_corners=False, models/SuperPointNet_gauss2_ssmall.py:87-91) as an operator, without the [b,n_classes,h,w] logits: convSout NHWC
 * [b][h/8*w/8][cs] (n_classes logits, channel stride cs >= n_classes), labels int64 [b,h,w] (values outside [0, n_classes) are
 * ignored) -> loss_dev[0] = mean NLL over the counted pixels and (optional, OVERWRITTEN) d loss / d convSout in the same layout.
 * algo: 0 = the kernel the training step would take, 1 = pixels-then-classes lanes (any h, w; <= 192 classes), 2 = (x, class) lanes
 * (<= 144 classes).  scratch: 64 KiB. */
int ssp_op_sem_loss(const float* sout_nhwc_dev, int cs, const int64_t* labels_dev, int b, int h, int w, int n_classes, int algo,
                    void* scratch_dev, size_t scratch_bytes, float* loss_dev, float* dsout_nhwc_dev, void* stream);

/* ---- pair construction on the device (dataset side of the reference, datasets/Coco.py:341-392) ----
 * ssp_op_warp_image : inv_warp_image_batch (utils/utils.py:347-385): out[p] = sample(img, inv_h * p), p on the
 *                     linspace(-1,1) grid, zeros padding, align_corners=True; nearest != 0 selects mode="nearest".
 * ssp_op_erode      : the erosion of compute_valid_mask (utils/utils.py:737-740): MORPH_ELLIPSE(2r,2r), anchor (r,r).
 * ssp_op_warp_labels: warpLabels (datasets/data_tools.py:37-63) on a keypoint MAP: every non-zero pixel (x,y) is warped
 *                     with T^-1 h T (utils/utils.py:297-300), kept if inside, rounded half-to-even, scattered as 1. */
int ssp_op_warp_image(const float* img_dev, const float* inv_h_dev, float* out_dev, int b, int h, int w, int nearest,
                      void* stream);
int ssp_op_erode(const float* mask_dev, float* out_dev, int b, int h, int w, int radius, void* stream);
int ssp_op_warp_labels(const float* labels_dev, const float* h_dev, float* out_dev, int b, int h, int w, void* stream);
int ssp_op_warp_labels_px(const float* labels_dev, const float* hpx_dev, float* out_dev, int b, int h, int w, void* stream);

/* Pair construction for real data:
 * ssp_op_sample_homographies: sample_homography_np (utils/homographies.py:12-141) with the keys of
 *     `warped_pair.params` + the inversion of datasets/Coco.py:342-350, counter-based device RNG (distribution-level
 *     equivalent of the numpy / scipy streams); h_dev = `homographies` (image -> warped), inv_h_dev = `inv_homographies`.
 * ssp_op_warp_labels_full : warpLabels(..., bilinear=True) (datasets/data_tools.py:37-63) on a keypoint map:
 *     labels [b,1,h,w], res [b,2,h,w] (warped - round(warped) at the rounded position), labels_bi [b,1,h,w]
 *     (get_labels_bi :26-34); any output may be NULL; last-write-wins scatters like torch.
 * ssp_op_sem_finalize     : datasets/Coco_sem.py:447-448: float class map -> int64, invalid pixels -> n_classes. */
typedef struct ssp_homography_params {
  int32_t perspective, scaling, rotation, translation, allow_artifacts;
  int32_t n_scales, n_angles; /* 5, 25 */
  float scaling_amplitude, perspective_amplitude_x, perspective_amplitude_y, patch_ratio, max_angle, translation_overflow;
} ssp_homography_params;
int ssp_op_sample_homographies(uint64_t seed, const ssp_homography_params* p, int b, float* h_dev, float* inv_h_dev,
                               void* stream);
int ssp_op_warp_labels_full(const float* labels_dev, const float* h_dev, float* labels_out_dev, float* res_out_dev,
                            float* bi_out_dev, int b, int h, int w, void* stream);
int ssp_op_warp_labels_full_px(const float* labels_dev, const float* hpx_dev, float* labels_out_dev, float* res_out_dev,
                               float* bi_out_dev, int b, int h, int w, void* stream);
/* ssp_op_label_quantize : the `*_gaussian` label maps of datasets/Coco.py:378,400 (ImgAugTransform with GaussianBlur sigma 0.2,
 * utils/photometric.py:59-78): uint8 quantisation floor(x * 255) / 255; the sigma-0.2 blur itself is the identity on 8 bits. */
int ssp_op_label_quantize(const float* in_dev, float* out_dev, size_t n, void* stream);
int ssp_op_sem_finalize(const float* sem_warped_dev, const float* valid_dev, int64_t* out_dev, size_t n, int n_classes,
                        void* stream);

/* ---- homography-adaptation export (export.py:192-352) ------------------------------
 * One image = n_views warped copies that form ONE BatchNorm batch (the reference leaves the net in train mode,
 * models/model_wrap.py:120).  ssp_export_points replaces the body of the export loop (export.py:296-309):
 *   fe.run(img, onlyHeatmap=True)   -> forward (detector head) + flattenDetection (utils/utils.py:515-560)
 *   combine_heatmap                 -> export.py:49-60; unwarp_h = the matrices the reference passes as
 *                                      `inv_homographies`, i.e. sample["homographies"] (export.py:281-284 swaps the keys)
 *   fe.getPtsFromHeatmap            -> models/model_wrap.py:266-293 (threshold, nms_fast :129-192, border removal)
 *   fe.soft_argmax_points           -> models/model_wrap.py:212-249 (when subpixel != 0)
 *   pts[:top_k]                     -> export.py:303-309
 * pts_dev[k]: [ssp_export_max_points][5] rows (x, y, confidence, sx, sy), descending confidence; the reference's
 * float64 point is (x + sx - 2, y + sy - 2, confidence) (sx = sy = 2 when subpixel == 0); count_dev[k]: rows written.
 * Equal confidences are ordered by the lower row-major pixel index (numpy's quicksort order is unspecified there).
 * workspace_dev[k]: ssp_export_workspace_bytes(p) bytes per image.  heatmap_out_dev (or its entries) may be NULL. */
typedef struct ssp_export_params {
  int32_t n_views;       /* data.homography_adaptation.num */
  int32_t height, width; /* multiples of 8 */
  float conf_thresh;     /* model.detection_threshold (compared in fp32, as numpy does) */
  int32_t nms_dist;      /* model.nms */
  int32_t border_remove; /* SuperPointFrontend_torch.border_remove = 4 (models/model_wrap.py:70) */
  int32_t top_k;         /* model.top_k, 0 = all */
  int32_t subpixel;      /* model.subpixel.enable */
} ssp_export_params;

size_t ssp_export_workspace_bytes(const ssp_export_params* p);
int ssp_export_max_points(const ssp_export_params* p);
int ssp_export_points(ssp_handle* h, const ssp_export_params* p, int n_images, const float* const* views_dev,
                      const float* const* masks_dev, const float* const* unwarp_h_dev, void* const* workspace_dev,
                      float* const* heatmap_out_dev, float* const* pts_dev, int32_t* const* count_dev, void* stream);

/* the stages as operators (parity tests):
 * ssp_op_homoadapt_views  : datasets/Coco.py:279-288: n warped copies of ONE [h,w] image (inv_warp_image_batch,
 *                           bilinear) and the nearest-warped all-ones masks (compute_valid_mask before erosion)
 * ssp_op_flatten_detection: flattenDetection on a public NCHW `semi` [n,65,hc,wc] (times mask [n,8hc,8wc] if given)
 * ssp_op_combine_heatmap  : combine_heatmap on heat = heatmap*mask and mask, both [n,h,w]
 * ssp_op_heatmap_points   : getPtsFromHeatmap + soft_argmax_points + top-k on one [h,w] heatmap
 * ssp_op_soft_argmax_points: soft_argmax_points for explicit points xy [n,2] (x, y; truncated to int) -> (sx, sy) [n,2] */
int ssp_op_homoadapt_views(const float* img_dev, const float* inv_h_dev, float* views_dev, float* masks_dev, int n, int h,
                           int w, void* stream);
int ssp_op_flatten_detection(const float* semi_nchw_dev, const float* mask_dev, float* heat_dev, int n, int hc, int wc,
                             void* stream);
int ssp_op_combine_heatmap(const float* heat_dev, const float* mask_dev, const float* unwarp_h_dev, float* out_dev, int n,
                           int h, int w, void* stream);
int ssp_op_heatmap_points(const float* heat_dev, const ssp_export_params* p, void* workspace_dev, float* pts_dev,
                          int32_t* count_dev, void* stream);
int ssp_op_soft_argmax_points(const float* heat_dev, const float* xy_dev, float* out_dev, int n, int h, int w,
                              void* stream);

/* ---- logging branch of train_val_sample (Train_model_heatmap_all.py:447-568) ------
 * ssp_detector_heatmap: get_heatmap (Train_model_frontend_all.py:664-669) = flattenDetection of the detector logits
 *                       of the last forward / pair step in `slot` (0 = image, 1 = warped image) -> heat [n,h,w].
 * ssp_op_heatmap_nms  : heatmap_to_nms / heatmap_nms (Train_model_heatmap_all.py:574-587,693-707) for n_maps heatmaps
 *                       [n_maps,h,w] (p->top_k / subpixel / n_views ignored) and batch_precision_recall's terms
 *                       (:614-622, utils/utils.py:929-941): nms_map_dev [n_maps,h,w] 0/1 (optional),
 *                       pr_dev [n_maps][2] = (precision, recall) against labels_dev [n_maps,h,w] (both optional). */
int ssp_detector_heatmap(ssp_handle* h, int slot, float* heat_dev, void* stream);
int ssp_op_heatmap_nms(const float* heat_dev, const ssp_export_params* p, int n_maps, void* workspace_dev,
                       const float* labels_dev, float* nms_map_dev, float* pr_dev, void* stream);

/* BatchNorm2d(train) (+ReLU (+MaxPool2d(2))) backward. y: raw conv output NHWC; dout: gradient wrt the activated
 * (and pooled) output; stats4 = scale|shift|mean|invstd ([4*C]); dgamma/dbeta/dbias are accumulated;
 * sums_dev: double [SSP_NREP][2*C] scratch. */
int ssp_op_bn_bwd(const float* y_dev, const float* dout_dev, const float* gamma_dev, const float* stats4_dev,
                  float* dy_dev, float* dgamma_dev, float* dbeta_dev, float* dbias_dev, double* sums_dev, int n, int h,
                  int w, int c, int relu, int pool, void* stream);
/* The same on channel-padded NHWC tensors (pixel stride cs >= c, cs % 4 == 0: the 65-channel detector head lives in
 * 68-float pixels); the per-channel vectors stay [c]. */
int ssp_op_bn_bwd_strided(const float* y_dev, const float* dout_dev, const float* gamma_dev, const float* stats4_dev,
                          float* dy_dev, float* dgamma_dev, float* dbeta_dev, float* dbias_dev, double* sums_dev, int n,
                          int h, int w, int c, int cs, int relu, int pool, void* stream);

/* Algorithm of the 3x3 forward / data-gradient convolutions whose input channels are a multiple of 16 (process-wide
 * DEFAULT, copied into a handle at ssp_create; takes effect at the next forward, which re-packs the weights): 1 (default) = Winograd on the fp32 matrix
 * cores, fp32 throughout: F(4x4,3x3) (4x fewer multiplies, conv_wino4_kernel) on maps of >= 60x80 pixels with >= 4 tile
 * blocks per CU, F(2x2,3x3) (2.25x fewer, software-pipelined kernel whose weight fragments come straight from L2)
 * elsewhere; results within ~2e-6 / ~3e-7 relative of the direct form, 9 = F(2x2,3x3) only (the default of rounds 1-2),
 * 10 = F(4x4,3x3) wherever legal (tests), 5 = the F(2x2,3x3) pipeline with the weights staged through LDS, 6 = its
 * two-workgroups-per-CU variant everywhere,
 * 2 = Winograd without the software pipeline (both for A/B measurements), 0 = direct implicit GEMM,
 * 3 = EXPERIMENTAL reduced precision: the Winograd kernels with bf16 matrix-core operands (fp32 storage, transforms,
 * accumulation and master weights); outputs within ~4e-3 relative RMS of fp32, gradients of the first layers up to
 * ~25 % off per step; never used for a reported fp32 number,
 * 7 = the same with split-bf16 (hi + lo) operands everywhere, 8 = MIXED (BASELINE configs[3], bench.py --dtype bf16): fp32
 * forward (algorithm 1), data / weight gradients of the 3x3 layers with one-part bf16 operands; losses equal the fp32 step's,
 * gradients within ~1e-2 per tensor, 11 = algorithm 1 with the Winograd F(3x3,4x4) weight gradient. */
int ssp_set_conv_algo(int algo);
/* the same choice for ONE handle (a new handle starts with the process-wide value of ssp_set_conv_algo, which also
 * governs the handle-less ssp_op_conv / ssp_op_conv_wgrad) */
int ssp_handle_set_conv_algo(ssp_handle* h, int algo);

/* perf-debug hook: ablate bits (1 no global loads, 2 no LDS writes, 4 no stores, 8 no MFMA) and grid override of
 * conv_mfma_kernel; (0, 0) restores the product behaviour. */
int ssp_debug_conv_knobs(int ablate, int grid);
/* perf-debug hook: workgroups per CU admitted by hipOccupancyMaxActiveBlocksPerMultiprocessor for a kernel family
 * (0 conv_wino_p2_kernel, 1 conv_wino_pipe_kernel, 2 wgrad_wino_kernel); negative = error */
int ssp_debug_occupancy(int which);

/* test hook: device pointer of an internal buffer ("gP","gQ","dsemi","ddesc","desc","dsout","Y<l>","A<l>" (pooled copy of layer l),
 * "scale<l>","shift<l>","mean<l>","invstd<l>"); under the bf16 path Y<l> / A<l> of the 3x3 layers, gP and gQ hold bf16 elements */
int ssp_debug_buffer(ssp_handle* h, int slot, const char* name, float** ptr, size_t* nfloats);

/* batch_descriptor_loss_sparse (utils/loss_functions/sparse_loss.py:267-284) on NHWC descriptor maps [B][hc*wc][256] with explicit
 * indices; out2_dev = {mean positive_dist, mean negative_dist}.  method: 0 = "2d" (bilinear grid_sample at normPts), 1 = "1d"
 * (index_select at the cell); dist: 0 = "cos", 1 = "euclidean" (pixelwise_contrastive_loss.py:140,185-210,247-258).  With dd_a /
 * dd_b (both or none; same layout, OVERWRITTEN) also the gradient of coef_pos * positive_dist + coef_neg * negative_dist. */
int ssp_op_sparse_loss(const float* desc_a_nhwc_dev, const float* desc_b_nhwc_dev, const int32_t* match_a_dev,
                       const int32_t* match_b_dev, const int32_t* nonmatch_b_dev, int b, int hc, int wc, int n_match,
                       int n_non, int method, int dist, float coef_pos, float coef_neg, float* dd_a_nhwc_dev, float* dd_b_nhwc_dev,
                       float* out2_dev, void* stream);

/* Dense descriptor loss as an operator (utils/utils.py:779-893) on NHWC descriptor maps [B][hc*wc][256]:
 * out3_dev = {loss_desc, pos_sum, neg_sum}; with dda_dev / ddb_dev (both or none) also the gradients of
 * scale * loss_desc (multi_task == 0) or scale * (pos_sum + neg_sum) (multi_task != 0) wrt both maps.
 * scratch: 64 KiB + B * cells * cells floats when gradients are requested. */
int ssp_op_dense_loss(const float* desc_a_nhwc_dev, const float* desc_b_nhwc_dev, const float* homographies_dev,
                      const float* valid_dev, int b, int hc, int wc, float lamda_d, float descriptor_dist, int multi_task,
                      float scale, void* scratch_dev, size_t scratch_bytes, float* out3_dev, float* dda_dev,
                      float* ddb_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SSP_HIP_H */
