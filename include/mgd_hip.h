/*
 * libmgd_hip.so — C-ABI of the MI355X-native MultiGridDet hot path (gfx950 / CDNA4).
 *
 * The reference (solufast-cvprojects/multigriddet) is pure Python on TensorFlow and has no
 * FFI/plugin boundary of its own; its drop-in boundary is the Python API.
 * Every entry point below therefore replaces a TF/Keras op or a numpy routine of the reference;
 * the reference interface each one stands in for is cited as file:line (paths relative to the
 * reference root).  The host-side mirror of the reference's Python API lives in
 * `multigriddet_amd/` and calls these through ctypes.
 *
 * Conventions
 *   - every pointer is a caller-owned DEVICE pointer unless the name ends in `_host`;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises;
 *   - no hidden allocation: scratch is a caller-provided workspace, sized by a *_workspace_size query;
 *   - return value 0 = ok, negative = MGD_E*; `mgd_last_error()` returns a thread-local message;
 *   - tensors are NHWC; bf16 is the 16-bit brain float (uint16_t storage); sizes are element counts.
 */
#ifndef MGD_HIP_H
#define MGD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGD_OK 0
#define MGD_EINVAL (-1)  /* bad argument (shape, alignment, unsupported option) */
#define MGD_ELAUNCH (-2) /* HIP launch failure                                   */
#define MGD_ENOSPC (-3)  /* workspace too small                                  */

const char* mgd_last_error(void);
int mgd_version(void);

/* Name of the kernel family this thread's last launch went to, e.g. "conv_gather_gemm(counted pipeline, ping-pong)":
 * measurement code tags its event brackets with it (bench.py).  No reference counterpart. */
const char* mgd_last_kernel(void);

/* ----------------------------------------------------------------------------------------------
 * Convolution engine (implicit GEMM on bf16 MFMA, fp32 accumulate).
 * Replaces: Keras Conv2D as used by DarknetConv2D / DarknetConv2D_BN_Leaky
 *           (multigriddet/models/layers.py:43-49, 88-95), its autodiff (dgrad, wgrad), and
 *           ZeroPadding2D(((1,0),(1,0))) + 'valid' stride-2 conv (models/backbones/darknet.py:33-34).
 *
 * One generic "gather-GEMM" descriptor covers forward, data-gradient (incl. the four output-parity
 * classes of the stride-2 transposed conv) and 1x1: for iteration point (n, i, j) and tap t the
 * source pixel is (i*in_stride + dh[t], j*in_stride + dw[t]) (zero outside the source) and the
 * destination pixel is (i*out_stride + out_off_h, j*out_stride + out_off_w).
 *   dst[n, ., ., co] = sum_{t, ci} src[n, ., ., ci] * wpk[co][t*Ci + ci]  (+ bias[co]) (+ addend)
 * `wpk` is a packed bf16 weight image [Co_pad][K_pad] produced by mgd_pack_weights / mgd_pack_weights_batch
 * (opaque: row-major for 32- and 64-row tiles, MFMA-fragment order when Co_pad % 128 == 0, see mgd_pack_weights).
 * Optional epilogues: per-channel sum / sum-of-squares of the (bf16-rounded) result into
 * `stats[(block % stats_replicas)][2][Co]` for training-mode BatchNorm; fp32 output.
 * Limits (MGD_EINVAL otherwise): N*Hg*Wg < 2^31 pixels; the kernels address `src` and `wpk` with 32-bit byte
 * offsets from a scalar base, so N*Hs*Ws*Ci*2 < 2^32 and Co_pad*K_pad*2 < 2^32 (4 GiB per operand; at 608x608 the
 * 64-channel 304x304 maps reach that at N ~ 360 images per call).
 * ---------------------------------------------------------------------------------------------- */
typedef struct mgd_conv_desc {
  const void* src;      /* bf16 [N, Hs, Ws, Ci]                                   */
  const void* wpk;      /* bf16 packed weights [Co_pad][K_pad]                    */
  void* dst;            /* bf16 (or f32 if dst_f32) [N, Hd, Wd, Co]               */
  const float* bias;    /* optional f32 [Co]                                      */
  const void* addend;   /* optional bf16 [N, Hd, Wd, Co], added before the store  */
  float* stats;         /* optional f32 [stats_replicas][2][Co], accumulated      */
  int32_t N, Hs, Ws, Ci;
  int32_t Hg, Wg;       /* iteration grid                                         */
  int32_t Hd, Wd, Co;
  int32_t in_stride, out_stride, out_off_h, out_off_w;
  int32_t ntaps;
  int32_t dh[9], dw[9];
  int32_t K_pad, Co_pad; /* of wpk                                                */
  int32_t dst_f32;
  int32_t stats_replicas;
  /* Optional: when dst is the activation gradient `da` of a BatchNorm+LeakyReLU layer, fold that layer's
   * backward reduction into this launch: bn_sums[(block % stats_replicas)][2][Co] += { sum dyh, sum dyh*yhat }
   * over the tile, dyh = da * leaky'(bn_y*bn_scale+bn_shift), yhat = (bn_y - bn_mean)*bn_invstd (what
   * mgd_bn_act_bwd_reduce computes in a separate pass).  bn_y has dst's shape. */
  const void* bn_y;
  const float* bn_scale;
  const float* bn_shift;
  const float* bn_mean;
  const float* bn_invstd;
  float* bn_sums;
  float bn_slope;
  /* inference with BatchNorm folded into the conv (weights pre-scaled by gamma/sqrt(var+eps), `bias` = the BN shift):
   * act_slope != 0 applies LeakyReLU(act_slope) to (acc + bias) before the optional `addend` (= the residual input),
   * i.e. the whole DarknetConv2D_BN_Leaky (+ Add) of models/layers.py:88-95 in one launch.  bf16 output only. */
  float act_slope;
  /* K ranges of the latency form (below): 0 or 1 = none.  `partial` / `partial_bytes`: its caller-owned workspace. */
  int32_t splitk;
  float* partial;
  int64_t partial_bytes;
  /* Latency form (latency != 0) for launches of a few thousand pixels (small-batch inference: 75 dependent launches, each
   * far too small to fill 256 CUs with 128 x 128 tiles): blocks of 128 channels x 64 pixels x one of max(splitk, 1) K
   * ranges, all of a block's K-steps in flight at once, and the ranges added INSIDE the kernel by the last block to arrive
   * at each tile (in range order: deterministic) - no second launch.  With splitk > 1 `partial` is the workspace: 16 KiB of
   * tile tickets that start at zero (launches leave them at zero), then splitk * tiles * 32 KiB of fp32
   * partial tiles, tiles = Co_pad/128 * ceil(N*Hg*Wg / 64) <= 4096; partial_bytes covers both (mgd_latency_workspace_size).
   * The workspace must be UNCACHED device memory (mgd_uncached_alloc) owned by the caller; launches that share one must be
   * ordered on one stream.  Needs Co_pad % 128 == 0, bf16 output, no stats / bn_y, and ntaps == 1 or
   * Ci % 64 == 0; anything else is refused. */
  int32_t latency;
  /* Kernel form.  0 (MGD_CONV_AUTO): the library chooses from the geometry (measured rules).  Any other
   * value forces one form and fails with MGD_EINVAL when that form cannot run the geometry - for tests and A/B measurements;
   * the library reads no environment variables.  form_arg: low 8 bits = the form's tile selector (0 = its default); with a
   * forced form + 256 = tap-major K order, + 512 = chunk-major K order (each 64-channel chunk of the pixel rows through all
   * its taps before the next chunk) where the form's default is the other one. */
  int32_t form;
  int32_t form_arg;
} mgd_conv_desc;

enum {
  MGD_CONV_AUTO = 0,
  MGD_CONV_THIN = 2,        /* conv_gemm2_kernel: 64- / 32-channel tiles, both operands in a 2-stage LDS-DMA ring        */
  MGD_CONV_PRODCONS = 6,    /* conv_gemm6_kernel: 4 loader + 4 MFMA waves, 128 x 128 tiles                                */
  MGD_CONV_GLOBALW = 8,     /* conv_gemm8_kernel: weight fragments straight from global memory, three blocks per CU      */
  MGD_CONV_COUNTED = 9,     /* conv_gemm9_kernel, 4 waves; form_arg = pixel tile / 16 (12, 8, 6, 4)                       */
  MGD_CONV_PINGPONG = 10,   /* conv_gemm9_kernel, 8 waves in two groups; form_arg = pixel tile / 16 (8, 12)               */
  MGD_CONV_PHASED = 12,     /* conv_gemm12_kernel: 8 waves, counted LDS-DMA across barriers; form_arg: 0 = 256 x 256,     */
                            /*   1 = 256 x 192, 2 = 128 channels x 384 pixels                                             */
  MGD_CONV_PATCH = 13       /* conv_patch_kernel (32 -> 64 and 64 -> 32 channel 3x3 layers)                               */
};

int mgd_conv_gather_gemm(const mgd_conv_desc* d, void* stream);

/* n = 2..4 descriptors of ONE geometry that differ in wpk, K_pad, the taps and the output offset only - the four output-parity
 * classes of a stride-2 data gradient (ZeroPadding2D + 'valid' stride-2 conv, models/backbones/darknet.py:33-34) - in ONE
 * launch: a block works on class (block / tiles).  128-channel weight tiles, Ci % 64 == 0, bf16 output, library dispatch;
 * MGD_EINVAL otherwise (launch the classes one by one with mgd_conv_gather_gemm). */
int mgd_conv_gather_gemm_classes(const mgd_conv_desc* d, int n, void* stream);

/* Workspace of the latency form with K ranges (mgd_conv_desc.latency, splitk > 1).  Partial tiles written by blocks on one
 * XCD are read by a block on another inside the same kernel, which ordinary (L2-cached) device memory does not guarantee, so
 * the workspace is UNCACHED device memory - and it is the CALLER's: the library keeps no buffer and no other state, so two
 * callers (two models, two streams) never share tickets or partial tiles.  One workspace serves the launches of ONE stream.
 *   mgd_latency_workspace_size(tiles, ranges): bytes for launches of up to `tiles` tiles (128 channels x 64 pixels) x `ranges`;
 *   mgd_uncached_alloc / mgd_uncached_free: hipExtMallocWithFlags(hipDeviceMallocUncached) on the current device, the first
 *     16 KiB (the tickets) zero-filled; both synchronise the device - call them outside stream capture;
 *   mgd_latency_tickets: test hook, waits for `stream` and copies the 4096 tickets to the host (all zero between launches).
 * No reference counterpart. */
int64_t mgd_latency_workspace_size(int tiles, int ranges);
int mgd_uncached_alloc(int64_t bytes, void** out);
int mgd_uncached_free(void* p);
int mgd_latency_tickets(const void* workspace, unsigned* out4096, void* stream);


/* Stride-2 data gradient of a 3x3 conv with 32 input / 64 output channels (the first down-sampling layer,
 * models/backbones/darknet.py:33-34 with ZeroPadding2D(((1,0),(1,0))): all four output-parity classes in one launch,
 * dy read once (the generic route is four mgd_conv_gather_gemm launches).  wpk[c] / K_pad[c]: the packed transposed
 * images of class c = ph*2 + pw as mgd_pack_weights writes them (taps in (kh, kw) order of the class).
 * dx[n, 2i+ph, 2j+pw, :] (+ addend); optional fused BatchNorm-backward sums as in mgd_conv_desc. */
typedef struct mgd_dgrad_s2_desc {
  const void* dy;      /* bf16 [N, Ho, Wo, Co]  */
  const void* wpk[4];  /* bf16 [32][K_pad[c]]    */
  void* dx;            /* bf16 [N, H, W, Ci]     */
  const void* addend;  /* optional bf16, dx layout */
  int32_t K_pad[4];
  int32_t N, Ho, Wo, Co, H, W, Ci;
  int32_t stats_replicas;
  const void* bn_y;
  const float *bn_scale, *bn_shift, *bn_mean, *bn_invstd;
  float* bn_sums;
  float bn_slope;
} mgd_dgrad_s2_desc;
int mgd_conv_dgrad_s2_patch(const mgd_dgrad_s2_desc* d, void* stream);

/* Weight gradient: dW[co][t][ci] += sum_p dy[p][co] * src[p (+) tap t][ci]   (fp32 atomics).
 * Replaces the Conv2D kernel gradient of Keras autodiff (layers.py:43-49).  Geometry fields have
 * the forward conv's meaning (src = forward input, dy = gradient of the forward output, whose
 * pixel grid is the iteration grid Hg x Wg; in_stride, dh, dw as in the forward descriptor).
 * `dw` is fp32 [Co][ntaps][Ci] (the master OHWI layout) and must be zeroed by the caller before the
 * first accumulation of a step. `splits` = number of pixel-range slices (>=1). */
typedef struct mgd_wgrad_desc {
  const void* src;  /* bf16 [N, Hs, Ws, Ci]     */
  const void* dy;   /* bf16 [N, Hg, Wg, Co]     */
  float* dw;        /* f32 [Co][ntaps][Ci]      */
  int32_t N, Hs, Ws, Ci, Hg, Wg, Co;
  int32_t in_stride, ntaps;
  int32_t dh[9], dw_off[9];
  int32_t splits;
  /* Kernel form: 0 = the library's choice from the geometry; otherwise that form or MGD_EINVAL (tests, A/B runs).
   * form_arg: MGD_WGRAD_DESC - ring depth 2 / 3 / 4 (0 = by layer). */
  int32_t form;
  int32_t form_arg;
  /* Optional workspace of the kernel-row form (MGD_WGRAD_ROW): with at least mgd_conv_wgrad_workspace_size(d) bytes its blocks
   * store per-split fp32 slabs with plain stores and a second launch adds them into dw (the fp32 atomics of 256 blocks x 192 KiB
   * run at a fifth of the store rate); without it the form uses atomics.  Caller-owned, one per stream. */
  float* partial;
  int64_t partial_bytes;
} mgd_wgrad_desc;

enum {
  MGD_WGRAD_AUTO = 0,
  MGD_WGRAD_PERTAP = 2,   /* conv_wgrad2_kernel: per-tap blocks, carried source coordinates (any stride)            */
  MGD_WGRAD_PATCH = 3,    /* conv_wgrad3_kernel: all nine taps of a [64 co] x [Ci] slice per block (Ci = 32 / 64)   */
  MGD_WGRAD_DESC = 4,     /* conv_wgrad4_kernel: per-tap blocks, raw buffer descriptors (stride-1 'same' layers)    */
  MGD_WGRAD_ROW = 5       /* conv_wgrad5_kernel: 128 x 128 x (three taps of a kernel row) blocks, 8 waves           */
};

int mgd_conv_wgrad(const mgd_wgrad_desc* d, void* stream);
int64_t mgd_conv_wgrad_workspace_size(const mgd_wgrad_desc* d);

/* Stem conv 3x3, Cin=3 -> Cout=32, stride 1, 'same' (models/backbones/darknet.py:21).  image f32 [N,H,W,3];
 * w f32 [32][3][3][3] (OHWI); y bf16 [N,H,W,32].  mgd_stem_fwd runs on the matrix cores straight from the fp32 image
 * (image and weights rounded to bf16, fp32 accumulation, K = 27 of one 32-deep MFMA) with the BatchNorm statistics
 * epilogue; mgd_stem_wgrad is the direct fp32 kernel (the engine uses mgd_stem_im2col + mgd_conv_wgrad instead). */
int mgd_stem_fwd(const float* image, const float* w, void* y, float* stats, int stats_replicas, int N, int H,
                 int W, void* stream);
/* BatchNorm-folded inference form of the stem (w pre-scaled by gamma / sqrt(moving_var + eps), bias = the BN shift):
 * y = LeakyReLU(act_slope)(conv + bias), bf16 - the DarknetConv2D_BN_Leaky of models/layers.py:88-95 in one launch. */
int mgd_stem_fwd_act(const float* image, const float* w, const float* bias, float act_slope, void* y, int N, int H, int W,
                     void* stream);
int mgd_stem_wgrad(const float* image, const void* dy, float* dw, int N, int H, int W, void* stream);
/* mgd_stem_wgrad with the stem's BatchNorm + LeakyReLU backward (layers.py:94-95) applied on the fly: takes da (gradient
 * wrt the activated stem output) and y (raw stem output) instead of dy, the per-channel scale/shift/mean/invstd of the
 * forward pass and the replicated sums [replicas][2][32] of (dyh, dyh*yhat) that the producer of da accumulated;
 * adds dbeta / dgamma (+=).  dy itself is never written. */
int mgd_stem_wgrad_bn(const float* image, const void* da, const void* y, const float* scale, const float* shift,
                      const float* save_mean, const float* save_invstd, const float* sums, int replicas, float* dgamma,
                      float* dbeta, float slope, float* dw, int N, int H, int W, void* stream);

/* Pack fp32 master weights W[Co][T][Ci] (OHWI) into a bf16 gather-GEMM image.
 * out[r][t'*Cin' + c] = transpose ? W[c][src_tap[t']][r] : W[r][src_tap[t']][c], zero padded to
 * [rows_pad][K_pad].  (transpose=1 builds data-gradient images.)
 * Element (r, k) is stored row-major when rows_pad % 128 != 0.  When rows_pad % 128 == 0 (K_pad % 64 == 0) the image is in
 * MFMA-fragment order: per (128-row tile cot, 64-deep K-step ks) one 16-KiB block of 1024 16-byte chunks, chunk
 * ((g*2 + kk)*64 + lane) holding row cot*128 + g*16 + (lane & 15), columns ks*64 + (kk*4 + (lane >> 4))*8 .. +8
 * (g = 16-row group 0..7, kk = 32-deep half) - a wave's A operand of a K-step is then eight coalesced 16-byte loads. */
int mgd_pack_weights(const float* w, void* out, int Co, int T, int Ci, int transpose, int ntaps_out,
                     const int32_t* src_tap_host, int rows_pad, int K_pad, void* stream);

/* All packed images of a network in ONE launch: `jobs_dev` is a device array of njobs descriptors sorted by
 * `begin` = index of the job's first 32-row x 64-column tile, a job having ntaps_out * ceil(cin/64) * ceil(rows/32) tiles
 * (rows, cin = Co, Ci or swapped when transpose); total = number of tiles.  Only the valid region of each
 * image is written: the caller zero-fills the images once at allocation. */
typedef struct mgd_pack_job {
  const float* w;      /* fp32 master weights [Co][T][Ci]          */
  void* out;           /* bf16 image [rows_pad][K_pad]             */
  int32_t Co, T, Ci, transpose, ntaps_out, rows_pad, K_pad, pad_;
  uint64_t srccode;    /* 4 bits per output tap: source tap index  */
  int64_t begin;
} mgd_pack_job;
int mgd_pack_weights_batch(const mgd_pack_job* jobs_dev, int njobs, int64_t total, void* stream);

/* Stem as a GEMM: bf16 im2col [N*H*W][32] of the fp32 image (k = (kh*3+kw)*3+c, k >= 27 zero) so that the
 * 3->32 conv (models/backbones/darknet.py:21) and its weight gradient run on the MFMA kernels as a 1x1
 * conv with Ci = 32. */
int mgd_stem_im2col(const float* image, void* out, int N, int H, int W, void* stream);

/* ----------------------------------------------------------------------------------------------
 * BatchNorm (training mode, Keras defaults eps 1e-3 / momentum 0.99) + LeakyReLU(0.1) + residual
 * Replaces: BatchNormalization() + LeakyReLU(alpha=0.1) (layers.py:94-95), Add (darknet.py:39)
 * and their gradients.
 * ---------------------------------------------------------------------------------------------- */
/* stats [R][2][C] -> mean,var; scale=gamma*rsqrt(var+eps), shift=beta-mean*scale; saves
 * mean/invstd for backward; moving = moving*momentum + batch*(1-momentum).  If training==0 the
 * moving statistics are used and nothing is updated. */
int mgd_bn_finalize(const float* stats, int replicas, int C, float count, const float* gamma, const float* beta,
                    float* moving_mean, float* moving_var, float* scale, float* shift, float* save_mean,
                    float* save_invstd, float eps, float momentum, int training, void* stream);
/* a = leaky(y*scale+shift) (+ residual); bf16 [P][C] */
int mgd_bn_act_fwd(const void* y, const float* scale, const float* shift, const void* residual, void* a,
                   int64_t P, int C, float slope, void* stream);
/* mgd_bn_finalize + mgd_bn_act_fwd in one launch (every block folds the replicas of its channel window). */
int mgd_bn_act_fwd_fused(const float* stats, int replicas, float count, const float* gamma, const float* beta,
                         float* moving_mean, float* moving_var, float* scale, float* shift, float* save_mean,
                         float* save_invstd, float eps, float momentum, int training, const void* y,
                         const void* residual, void* a, int64_t P, int C, float slope, void* stream);
/* sums[R][2][C] += { sum dyh, sum dyh*yhat },  dyh = da * leaky'(y*scale+shift) */
int mgd_bn_act_bwd_reduce(const void* da, const void* y, const float* scale, const float* shift,
                          const float* save_mean, const float* save_invstd, float* sums, int replicas,
                          int64_t P, int C, float slope, void* stream);
/* dgamma,dbeta <- sums; dy = scale*(dyh - mean(dyh) - yhat*mean(dyh*yhat));  if frozen: dy = scale*dyh */
int mgd_bn_act_bwd_apply(const void* da, const void* y, const float* scale, const float* shift,
                         const float* save_mean, const float* save_invstd, const float* sums, int replicas,
                         float* dgamma, float* dbeta, void* dy, int64_t P, int C, float slope, int frozen,
                         void* stream);

/* UpSampling2D(2) nearest + Concatenate([up, skip]) (heads/multigrid_head.py:296-298) and backward. */
int mgd_upsample_concat_fwd(const void* u, const void* skip, void* out, int N, int h, int w, int Cu, int Cs,
                            void* stream);
int mgd_upsample_concat_bwd(const void* dout, void* du, void* dskip, int N, int h, int w, int Cu, int Cs,
                            void* stream);
/* dbias[c] = sum_p dy[p][c] for the three linear prediction convs (multigrid_head.py:71). */
int mgd_bias_grad(const void* dy_bf16, float* dbias, int64_t P, int C, void* stream);
int mgd_f32_to_bf16(const float* in, void* out, int64_t n, void* stream);
int mgd_bf16_to_f32(const void* in, float* out, int64_t n, void* stream);

/* Adam with Keras semantics (config/model_builder.py:86-96): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
 * p -= lr_t*m/(sqrt(v)+eps).  grad_scale multiplies g first (1/world for DP averaging).
 * weight_decay > 0 gives AdamW (decoupled, Keras: p -= lr*wd*p). */
int mgd_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, int step, float grad_scale, float weight_decay, void* stream);
/* Same update with lr_t = hyper[0] and lr*weight_decay = hyper[1] read from DEVICE memory, so that a
 * captured hipGraph of the whole step can be replayed while the schedule / bias correction advance. */
int mgd_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, float beta1,
                      float beta2, float eps, float grad_scale, void* stream);
int mgd_sgd_step(float* p, const float* g, float* mom, int64_t n, float lr, float momentum, int nesterov,
                 float grad_scale, void* stream);

/* ----------------------------------------------------------------------------------------------
 * 3x3 multi-grid y_true target builder.
 * mode 0 (T1) replaces tf_preprocess_true_boxes (multigriddet/data/generators.py:2696-3390);
 * mode 1 (T2) replaces preprocess_true_boxes      (multigriddet/data/generators.py:3393-3473).
 * boxes f32 [B][M][5] (x1,y1,x2,y2,cls); anchors f32 [L][A][2]; y_true[l] f32 [B][gh_l][gw_l][5+A+C]
 * (fully written, zeros included).  assign (optional) int32 [B][M][4] = layer, anchor, row, col.
 * ---------------------------------------------------------------------------------------------- */
size_t mgd_build_targets_workspace_size(int B, int M, int L, const int32_t* grid_hw_host);
int mgd_build_targets(const float* boxes, int B, int M, const float* anchors, int L, int A, int C, int in_h,
                      int in_w, const int32_t* grid_hw_host, float* const* y_true_host, int32_t* assign,
                      int mode, void* ws, size_t ws_bytes, void* stream);

/* ----------------------------------------------------------------------------------------------
 * MultiGridLoss forward + backward (multigriddet/losses/multigrid_loss.py:233-443 and helpers
 * :445-1043).  One call per step handles all L scales.  components[8] (device) receives
 * {loc, obj, anchor(pre-scaled), cls, consensus_coord, consensus_obj, consensus_cls, total}.
 * grad_f32[l] / grad_bf16[l] (either may be NULL) receive d total / d y_pred[l].
 * ---------------------------------------------------------------------------------------------- */
typedef struct mgd_loss_cfg {
  int32_t L, A, C, B;
  int32_t in_h, in_w;
  int32_t grid_h[4], grid_w[4];
  float anchors[4][8][2];
  float ignore_thresh, label_smoothing;
  int32_t loss_option;
  float coord_scale, object_scale, no_object_scale, class_scale, anchor_scale;
  int32_t norm_batch, norm_positives, norm_grid; /* multiplicities of each entry in loss_normalization */
  int32_t use_iou_aware_objectness;
  float iou_objectness_power, iou_objectness_ratio;
  float trainable_nms_weight, trainable_nms_power;
  int32_t use_consensus_loss;
  float consensus_iou_power, consensus_min_iou, consensus_coord_scale, consensus_obj_scale,
      consensus_class_scale, consensus_center_tolerance;
  int32_t consensus_stop_gradient;
  int32_t use_focal_loss; /* 0 = BCE, 1 = sigmoid focal (losses/focal_loss.py:40-77) */
  float focal_alpha, focal_gamma;
  float grad_out_scale;   /* multiplies the gradient only (loss scaling); 1.0 normally */
  /* loss_option 3 localisation (losses/iou_losses.py:36-237 via multigrid_loss.py:353-364): 0 = MSE (no flag set),
   * 1 = GIoU, 2 = DIoU, 3 = CIoU.  iou_compat 0 = "tf_ref": the reference's arithmetic, including its
   * [B,H,W] * [B,H,W,1] broadcast (defined for H == W and B == 1 or B == H only - MGD_EINVAL otherwise) on the raw
   * offset / log-ratio tensors; 1 = "fixed": per-cell mask, boxes decoded to grid-cell units. */
  int32_t iou_loss, iou_compat;
  /* SoftmaxFocalLoss classification (losses/focal_loss.py:80-114 via multigrid_loss.py:815-828); softmax_compat as
   * iou_compat (tf_ref additionally needs C == 1 or C == W: class_weights [1,1,1,C] multiplies along W). */
  int32_t use_softmax_focal, softmax_compat;
} mgd_loss_cfg;

size_t mgd_loss_workspace_size(const mgd_loss_cfg* cfg);
int mgd_loss_fwd_bwd(const mgd_loss_cfg* cfg, const float* const* y_pred_host, const float* const* y_true_host,
                     const float* class_weights, float* const* grad_f32_host, void* const* grad_bf16_host,
                     float* components, void* ws, size_t ws_bytes, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Decode + correct_boxes + confidence filter (multigriddet/postprocess/multigrid_decode.py:100-235,
 * 262-283) and greedy / soft NMS + top-k + xyxy (postprocess/nms.py:83-385,
 * multigrid_decode.py:300-345, 397-422).  Batched over images.
 * cand layout per image: boxes f32 [cap][4] (top-left xywh, image px), score f32 [cap], cls i32 [cap],
 * count i32.  Candidates are emitted in the reference's row order (scale-major, then cell).  The output buffers may be
 * uninitialised: the kernels write every count and zero every slot past it (candidates and detections alike).
 * ---------------------------------------------------------------------------------------------- */
typedef struct mgd_decode_cfg {
  int32_t L, A, C, B;
  int32_t in_h, in_w;
  int32_t grid_h[4], grid_w[4];
  float anchors[4][8][2];
  int32_t use_softmax, rescore;
  float confidence;
  int32_t cap; /* candidate capacity per image (>= total cells to be safe) */
  int32_t tag_scale; /* 1: cand_cls = class | scale << 16, for mgd_nms(method | MGD_NMS_PER_SCALE) */
} mgd_decode_cfg;

size_t mgd_decode_workspace_size(const mgd_decode_cfg* cfg);
int mgd_decode(const mgd_decode_cfg* cfg, const float* const* y_pred_host, const float* image_hw /*[B][2] dev*/,
               float* cand_boxes, float* cand_scores, int32_t* cand_cls, int32_t* cand_count, void* ws,
               size_t ws_bytes, void* stream);

/* method: 0 = standard/cluster (IoU, nms.py:83-148, 320-385), 1 = DIoU (nms.py:151-231),
 * 2 = soft (nms.py:234-317: sigma 0.5, score threshold 1e-3; `threshold` is ignored, survivors leave in
 * original candidate order with their decayed scores, or the top max_boxes by decayed score if more survive).
 * out_boxes i32 [B][max_boxes][4] (xyxy, clipped, floor(v+0.5)) or f32 xywh if !return_xyxy;
 * out_count[b] = number of detections.  Workspace: mgd_nms_workspace_size(B, cap) bytes. */
/* method | MGD_NMS_PER_SCALE (greedy methods): "per-scale NMS" of the north-star - a box only suppresses boxes decoded
 * from its own scale (the scale id is read from cand_cls >> 16, see mgd_decode_cfg.tag_scale, and stripped from the
 * output); ranking and the top-max_boxes cut stay global, all in the same single launch.  NOT the reference's
 * semantics (multigrid_decode.py:98 concatenates the scales before NMS): an option, off by default. */
#define MGD_NMS_PER_SCALE 0x100
size_t mgd_nms_workspace_size(int B, int cap);
int mgd_nms(const float* cand_boxes, const float* cand_scores, const int32_t* cand_cls,
            const int32_t* cand_count, int B, int cap, int method, float threshold, int max_boxes,
            const float* image_hw, int return_xyxy, void* out_boxes, float* out_scores, int32_t* out_cls,
            int32_t* out_count, void* ws, size_t ws_bytes, void* stream);

/* Weighted Boxes Fusion on one model's detections, as MultiGridDecoder.handle_predictions(use_wbf=True) runs it
 * (multigriddet/postprocess/wbf.py:79-199 via multigrid_decode.py:281-287, iou_thr = nms_threshold,
 * skip_box_thr 0, conf_type 'avg'): per class, greedy clusters around the highest-scoring unused box
 * (IoU against the seed), score-weighted mean box (float64), mean score; output in (class ascending, seed score
 * descending) order, or the top max_boxes by score if there are more.  Same argument meaning as mgd_nms. */
size_t mgd_wbf_workspace_size(int B, int cap);
int mgd_wbf(const float* cand_boxes, const float* cand_scores, const int32_t* cand_cls,
            const int32_t* cand_count, int B, int cap, float iou_threshold, int max_boxes,
            const float* image_hw, int return_xyxy, void* out_boxes, float* out_scores, int32_t* out_cls,
            int32_t* out_count, void* ws, size_t ws_bytes, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Evaluation (mAP) hot spot: multigriddet/evaluation/metrics.py:28-71 (calculate_iou_matrix) and :73-219
 * (match_predictions_to_gt / match_predictions_to_gt_cached).  float64 like the reference.
 * mode 0: boxes are xyxy (the cached-IoU path); mode 1: the un-cached path's BoxUtils.box_iou
 * (utils/boxes.py:16-57), which reads the same four numbers as (cx, cy, w, h) - the reference's per-scale
 * metrics always go that way, so it is reproduced.
 * mgd_eval_match: predictions are grouped by (image, class); group g owns predictions
 * [group_pred_start[g], group_pred_start[g+1]) in DESCENDING score order and ground truths
 * [group_gt_start[g], group_gt_start[g+1]).  For each of the num_thresholds (<= 16) IoU thresholds a prediction
 * takes the unmatched ground truth of its group with the highest IoU (first maximum, IoU > 0) and is a true
 * positive if that IoU >= threshold: tp[t][p] in {0,1}, tp is [num_thresholds][num_preds] bytes. */
int mgd_iou_matrix(const double* boxes1 /*[n][4]*/, const double* boxes2 /*[m][4]*/, double* out /*[n][m]*/, int n, int m,
                   int mode, void* stream);
int mgd_eval_match(const double* pred_boxes, const int32_t* group_pred_start, const double* gt_boxes,
                   const int32_t* group_gt_start, int num_groups, int max_group_gts, const double* thresholds,
                   int num_thresholds, int mode, uint8_t* tp, long long num_preds, void* stream);

/* ----------------------------------------------------------------------------------------------
 * On-device batch augmentation (multigriddet/data/generators.py:561-1009 Mosaic, :1164-1282
 * GridMask, :1012-1161 MixUp).  Random draws are made on the host by the caller (so that the
 * oracle can consume the same draws) and passed in as small parameter arrays.
 * ---------------------------------------------------------------------------------------------- */
int mgd_mosaic(const float* images, const float* boxes, int B, int S, int M_in, const int32_t* src_idx /*[B][4]*/,
               const int32_t* crop_xy /*[B][2]*/, float min_wh, float* out_images, float* out_boxes, int M_out,
               int32_t* overflow, void* stream);
int mgd_gridmask(float* images, float* boxes, int B, int S, int M, const int32_t* apply /*[B]*/,
                 const int32_t* d_l_off /*[B][3]: d, l, offset*/, float keep_frac, void* stream);
int mgd_mixup(const float* images, const float* boxes, int B, int S, int M_in, const int32_t* partner /*[B]*/,
              const float* lam /*[B]*/, float* out_images, float* out_boxes, int M_out, void* stream);

/* ---- data-parallel exchange (the reference has no multi-GPU path, trainers/trainer.py:430-594 runs one
 * device).  One process per GPU; the only collective of the path is the SUM of the flat fp32 gradient buffer, cut into
 * buckets that are launched while backward is still running.  Thin RCCL binding (librccl is opened lazily): rank 0
 * draws a 128-byte id with mgd_comm_unique_id and hands it to the other ranks out of band (torch.distributed store,
 * MPI, a file); every rank then calls mgd_comm_init on its own current device.  mgd_comm_allreduce_bucket is in
 * place, asynchronous on `stream`, and must be called in the same bucket order on every rank. */
#define MGD_COMM_ID_BYTES 128
int mgd_comm_unique_id(void* id128);
int mgd_comm_init(void** comm, int rank, int world, const void* id128);
int mgd_comm_allreduce_bucket(void* comm, float* grads, int64_t count, void* stream);
int mgd_comm_destroy(void* comm);

/* ----------------------------------------------------------------------------------------------
 * Inference-side letterbox on the device (multigriddet/utils/preprocessing.py:12-90: PIL Image.resize(BICUBIC) +
 * centred paste on a (128,128,128) canvas + /255).  src: uint8 [H][W][3] device frame; dst: f32 [Hd][Wd][3] (one image
 * slot of the NHWC model input).  (nw, nh) = int(W*scale), int(H*scale), scale = min(Wd/W, Hd/H); (dx, dy) = centred
 * offset.  kx/bx, ky/by: PIL's fixed-point resampling tables for the horizontal (W -> nw) and vertical (H -> nh) pass:
 * k[out][ks] int32 coefficients (22 fractional bits), b[out][2] = (first source index, count); device pointers, built
 * by the host (multigriddet_amd/utils/preprocessing.py: resample_tables).  Result equals PIL's bit for bit.
 * Workspace: mgd_letterbox_workspace_size(H, nw) bytes (the 8-bit intermediate of the horizontal pass).
 * ---------------------------------------------------------------------------------------------- */
size_t mgd_letterbox_workspace_size(int H, int nw);
int mgd_letterbox_u8(const uint8_t* src, int H, int W, float* dst, int Hd, int Wd, int nh, int nw, int dy, int dx,
                     const int32_t* kx, const int32_t* bx, int ksx, const int32_t* ky, const int32_t* by, int ksy,
                     float fill, void* ws, size_t ws_bytes, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Strict-parity fp32 execution (the reference's default numeric type: Keras float32 layers, models/layers.py:43-95).
 * fp32 NHWC activations, fp32 OHWI weights [Co][k*k][Ci] (the master layout, no packing), fp32 accumulation; direct
 * kernels, no attempt at speed - for end-to-end comparison of the 69-conv graph with the oracle under a tight bound
 * (Network(precision="fp32")).  Geometry as Conv2D in the reference: 'same' for stride 1; stride 2 =
 * ZeroPadding2D(((1,0),(1,0))) + 'valid' (models/backbones/darknet.py:33-34).
 * dgrad: dx = (addend) + conv^T(dy); wgrad: dw += x (*) dy; bn_stats: stats[0..C) += sum y, stats[C..2C) += sum y^2 (feed
 * mgd_bn_finalize with replicas = 1); bn_act_bwd: sums is a zeroed [2C] scratch; dgamma / dbeta are accumulated.
 * ---------------------------------------------------------------------------------------------- */
int mgd_conv2d_f32_fwd(const float* x, const float* w, float* y, const float* bias, int N, int H, int W, int Ci, int Co,
                       int k, int s, void* stream);
int mgd_conv2d_f32_dgrad(const float* dy, const float* w, float* dx, const float* addend, int N, int H, int W, int Ci,
                         int Co, int k, int s, void* stream);
int mgd_conv2d_f32_wgrad(const float* x, const float* dy, float* dw, int N, int H, int W, int Ci, int Co, int k, int s,
                         void* stream);
int mgd_bn_stats_f32(const float* y, int64_t P, int C, float* stats, void* stream);
int mgd_bn_act_fwd_f32(const float* y, const float* scale, const float* shift, const float* residual, float* a, int64_t P,
                       int C, float slope, void* stream);
int mgd_bn_act_bwd_f32(const float* da, const float* y, const float* scale, const float* shift, const float* save_mean,
                       const float* save_invstd, float* sums, float* dgamma, float* dbeta, float* dy, int64_t P, int C,
                       float slope, int frozen, void* stream);
int mgd_upsample_concat_fwd_f32(const float* u, const float* skip, float* out, int N, int h, int w, int Cu, int Cs,
                                void* stream);
int mgd_upsample_concat_bwd_f32(const float* dout, float* du, float* dskip, int N, int h, int w, int Cu, int Cs,
                                void* stream);
int mgd_bias_grad_f32(const float* dy, float* dbias, int64_t P, int C, void* stream);

/* ----------------------------------------------------------------------------------------------
 * Launch plans (csrc/plan.cpp).  A step's calls into this library are recorded once - entry point, arguments, stream slot,
 * cross-stream waits - and replayed by ONE call that needs no interpreter: the counterpart of what `model.fit` does per
 * batch in the reference (multigriddet/trainers/trainer.py:572-581), where Keras' executor, not Python, issues the kernels.
 * Not a hipGraph: the streams stay separate hardware queues; only the host side of the launches moves to C.
 *   mgd_plan_add_call: `name` = an int-returning entry point declared above; argument i is words[i] read by kinds[i]:
 *     0 integer / pointer value, 1 float (bit pattern in the low 32 bits), 2 double (bit pattern), 3 pointer to
 *     blob + words[i] (the blob - descriptors the call takes by pointer - is copied into the plan), 4 stream slot
 *     (streams[words[i]] at replay), 5 parameter slot (params[words[i]] at replay: pointers that change from run to run).
 *   mgd_plan_add_wait: at this point of the replay streams[waiting] waits for what streams[signalling] holds so far.
 *   mgd_plan_run: replays in recording order; returns the first failing call's code (mgd_last_error has its message).
 * A plan is immutable while it runs; one thread replays it at a time.  mgd_memset_async: hipMemsetAsync as an entry point,
 * so that a step's buffer clears are part of its plan. */
typedef struct mgd_plan mgd_plan;
int mgd_plan_create(mgd_plan** out);
int mgd_plan_destroy(mgd_plan* plan);
int mgd_plan_size(const mgd_plan* plan);
int mgd_plan_add_call(mgd_plan* plan, const char* name, const int64_t* words, const uint8_t* kinds, int nargs, const void* blob,
                      int64_t blob_bytes);
int mgd_plan_add_wait(mgd_plan* plan, int waiting, int signalling);
int mgd_plan_run(const mgd_plan* plan, void* const* streams, int nstreams, void* const* params, int nparams);
int mgd_memset_async(void* p, int value, int64_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MGD_HIP_H */
