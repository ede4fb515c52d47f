/*
 * gaiaseg_hip.h — C-ABI of the MI355X (gfx950) kernels behind the GAIA-seg supernet hot path.
 *
 * The reference (GAIA-seg) has no FFI: every kernel it runs is reached through torch ops called
 * from Python nn.Modules.  This header therefore declares one entry point
 * per torch-op call site on the hot path; each comment cites the
 * reference call site the entry point replaces.  All functions
 *   - take raw device pointers, sizes, element strides and a hipStream_t passed as void*,
 *   - allocate nothing (workspaces are passed in; query sizes with the *_workspace_bytes calls),
 *   - are re-entrant, keep no global state, never synchronise the device,
 *   - return 0 on success, a negative GS_E_* code for argument errors and a positive hipError_t
 *     value when a launch failed.
 *
 * Tensor conventions (fp32 everywhere, like the reference: fp16_enabled=False,
 * gaiaseg/models/decode_heads/dynamic_fcn_head.py:81):
 *   activations : NHWC, element (n,h,w,c) at n*H*W*ld + (h*W + w)*ld + c, ld >= C ("pixel stride"),
 *                 ld and C multiples of 4 (the only exceptions are the 3-channel input image, read
 *                 through explicit strides, and num_classes, which the host pads to 4 with zeros);
 *   conv weight : physical [KH][KW][Ci_max][Co_ld] (HWIO) — the max-size supernet tensor; the
 *                 kernels read the leading slice [:, :, :Ci, :Co] in place (DynConv2d semantics,
 *                 SURVEY.md Appendix A1), so no per-step repacking of the sliced weight exists;
 *   labels      : int64 [N,H,W], ignore_index (255) allowed.
 */
#ifndef GAIASEG_HIP_H
#define GAIASEG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_OK 0
#define GS_E_BADARG (-1)     /* inconsistent / unsupported descriptor */
#define GS_E_ALIGN (-2)      /* pointer or stride not aligned as documented */
#define GS_E_WORKSPACE (-3)  /* workspace too small */
#define GS_E_NULL (-4)       /* required pointer is NULL */

/* ABI version of this header; bumped on any signature change. */
int gs_abi_version(void);
/* Human-readable text for a return code of any function below. */
const char* gs_error_string(int code);
/* gfx target the library was compiled for ("gfx950"). */
const char* gs_target_arch(void);

/* ------------------------------------------------------------------------------------------ */
/* Dynamic (slimmable) convolution as implicit GEMM on fp32 MFMA — K1/K2/K3/K6/K10/K11/K12/K16 */
/* ------------------------------------------------------------------------------------------ */
/* Replaces F.conv2d(x, weight[:Co,:Ci], bias[:Co], stride, padding, dilation) of gaiavision
 * DynConv2d as called from DynamicBottleneck / DynamicConvModule / conv_seg
 * (gaiaseg/models/utils/dynamic_res_layer.py:84-125, gaiaseg/models/backbones/dynamic_resnet.py:255-302,
 *  gaiaseg/models/decode_heads/dynamic_fcn_head.py:76-126, dynamic_psp_head.py:53-59,123,140-147,
 *  dynamic_uper_head.py:40-79) and its autograd backward (dgrad / wgrad). */
typedef struct gs_conv_desc {
  int32_t N, H, W;        /* input batch and spatial size                                   */
  int32_t Ci, Co;         /* ACTIVE channels this step (Co padded to a multiple of 4)       */
  int32_t Ci_max, Co_ld;  /* weight physical dims: [KH][KW][Ci_max][Co_ld], Co_ld % 4 == 0   */
  int32_t KH, KW;
  int32_t stride, pad, dil;
  int32_t Ho, Wo;         /* floor((H + 2p - d(k-1) - 1)/s) + 1, checked by the library      */
  int64_t x_sn, x_sh, x_sw, x_sc; /* element strides of x (NHWC: x_sc = 1; image: NCHW)     */
  int32_t ldy;            /* pixel stride of y / dy (>= Co, % 4 == 0)                        */
  int32_t ld_add;         /* pixel stride of the optional addend (0 if unused)               */
  int32_t role;           /* GS_CONV_ROLE_*: profiling label only, results are identical     */
  int32_t reserved;       /* must be 0                                                       */
  /* Optional: x is the INPUT of a BatchNorm + ReLU whose output the convolution consumes
   * (bottleneck bn1 -> conv2, bn2 -> conv3).  in_affine = that BatchNorm's coefficient block
   * [scale | beta | mean | invstd][Ci] (as written by gs_conv_bn_forward / gs_bn_finalize) and the
   * forward and weight-gradient kernels evaluate relu((x - mean) * scale + beta) in their operand
   * loaders: the normalised activation is never written to or read from HBM, zero padding applies
   * to the activation.  gs_conv2d_dgrad ignores it (its result is the gradient w.r.t. the
   * activation).  Only where gs_conv2d_in_affine_supported(d) != 0; NULL = x is used as it is. */
  const float* in_affine;
} gs_conv_desc;
/* 1 if gs_conv2d_forward / gs_conv2d_wgrad / gs_conv_bn_* accept d->in_affine for this shape
 * (NHWC x, 1x1 or 3x3, Ci % 16 == 0, Ci <= 640: the fast kernels with 64-row tiles). */
int gs_conv2d_in_affine_supported(const gs_conv_desc* d);
/* role = GS_CONV_ROLE_BOTTLENECK3X3 marks conv2 of DynamicBottleneck (SURVEY.md K3,
 * gaiaseg/models/utils/dynamic_res_layer.py:96-106): the forward dispatches an identically compiled
 * but separately NAMED kernel instantiation (igemm_rows_fast_kernel<..., 1> and
 * splitk_reduce_kernel<false, 1>), so a rocprofv3 kernel trace reports the headline kernel of
 * bench.py's roofline on its own rows. */
#define GS_CONV_ROLE_GENERIC 0
#define GS_CONV_ROLE_BOTTLENECK3X3 1

/* bytes of split-K scratch the three calls below may use for this descriptor (max of the three) */
size_t gs_conv2d_workspace_bytes(const gs_conv_desc* d);

/* y[n,ho,wo,:Co] = conv(x, w[:, :, :Ci, :Co]) (+ bias[:Co]) (+ addend).  bias/addend may be NULL.
 * The network's stem -- 7x7, stride 2, pad 3, Ci = Ci_max = 3 on the NCHW image, Co in {32, 48, 64},
 * Wo % 128 == 0 (gaiaseg/models/backbones/dynamic_resnet.py:290-297) -- has its own forward and
 * weight-gradient kernels (csrc/stem.hip); gs_debug_last_conv_launch reports them as 128-row tiles. */
int gs_conv2d_forward(const gs_conv_desc* d, const float* x, const float* w, const float* bias,
                      const float* addend, float* y, void* workspace, size_t workspace_bytes,
                      void* stream);
/* dx[n,h,w,:Ci] (= or +=) sum_{taps,co} dy * w.   dx is NHWC with pixel stride x_sw (x_sc == 1).
 * accumulate != 0 adds to the existing dx (residual / multi-consumer gradients). */
int gs_conv2d_dgrad(const gs_conv_desc* d, const float* dy, const float* w, float* dx,
                    int accumulate, void* workspace, size_t workspace_bytes, void* stream);
/* dw[:, :, :Ci, :Co] = sum_pixels x (gathered) * dy, written into the max-size gradient tensor
 * (same physical layout as w); the rest of dw is left untouched (zero by the host's contract). */
int gs_conv2d_wgrad(const gs_conv_desc* d, const float* x, const float* dy, float* dw,
                    void* workspace, size_t workspace_bytes, void* stream);

/* out[c] = sum over rows of src[r*ld + c], c < C  (conv_seg bias gradient; also used by tests).
 * workspace >= gs_colsum_workspace_bytes(rows, C). */
size_t gs_colsum_workspace_bytes(int64_t rows, int32_t C);
int gs_colsum(const float* src, int64_t rows, int32_t C, int32_t ld, float* out, void* workspace,
              size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Dynamic BatchNorm (+ residual add, + ReLU) — K4, K8                                         */
/* ------------------------------------------------------------------------------------------ */
/* Replaces F.batch_norm on the leading C-slice of max-size parameters (gaiavision DynBN /
 * DynSyncBN; call sites dynamic_resnet.py:267-300,411,
 * dynamic_res_layer.py:92, DynamicBottleneck norm1-3, DynamicConvModule norm) fused with the
 * following nn.ReLU(inplace=True) and the bottleneck's `out += identity`.
 *
 * Training forward is three launches:
 *   gs_bn_stats     : per-channel shifted sums  S1 = sum(x - x[0,c]),  S2 = sum (x - x[0,c])^2
 *                     reduced in a fixed order (deterministic) into sums[2*C]
 *   (optional all-reduce of sums[] + count across ranks for SyncBN — done by the host)
 *   gs_bn_finalize  : mean/var -> coeffs {scale = gamma*invstd, beta, mean, invstd}; updates
 *                     running_mean/var (momentum, unbiased var)
 *   gs_bn_apply     : y = relu?((x - mean)*scale + beta (+ residual))
 */
size_t gs_bn_stats_workspace_bytes(int64_t rows, int32_t C);
/* sums: [3*C] floats = {S1[C], S2[C], shift[C]} (shift = x[0,c], the conditioning offset). */
int gs_bn_stats(const float* x, int64_t rows, int32_t C, int32_t ldx, float* sums,
                void* workspace, size_t workspace_bytes, void* stream);
/* count = number of samples per channel that produced sums (N*H*W, or the global count for SyncBN).
 * coeffs: [4*C] = {scale[C], beta[C], mean[C], invstd[C]}.  running_* may be NULL (no update).
 * gamma/beta may be NULL (treated as 1/0). */
int gs_bn_finalize(const float* sums, double count, int32_t C, const float* gamma,
                   const float* beta, float eps, float momentum, float* running_mean,
                   float* running_var, float* coeffs, void* stream);
/* SyncBN exchange (torch.nn.SyncBatchNorm's all_gather of [mean, var, count]; heads
 * with norm_cfg SyncBN): gs_bn_sync_local turns the shifted sums of gs_bn_stats into this rank's
 * payload local[2C+1] = {mean[C], biased var[C], count} (double); after the host's all_gather into
 * gathered[world][2C+1], gs_bn_sync_merge produces the merged sums {0, gvar*total, gmean}[3C] that
 * gs_bn_finalize takes (count = total).  One launch each. */
int gs_bn_sync_local(const float* sums, double count, int32_t C, double* local, void* stream);
int gs_bn_sync_merge(const double* gathered, int32_t world, int32_t C, float* merged, void* stream);
/* gs_bn_stats + gs_bn_finalize in two launches instead of three, for rank-local statistics
 * (count = rows).  Bit-identical to the separate calls. */
int gs_bn_stats_finalize(const float* x, int64_t rows, int32_t C, int32_t ldx, const float* gamma,
                         const float* beta, float eps, float momentum, float* running_mean,
                         float* running_var, float* coeffs, void* workspace,
                         size_t workspace_bytes, void* stream);
/* Eval-mode coefficients from running statistics (norm_eval / inference). coeffs as above. */
int gs_bn_eval_coeffs(const float* running_mean, const float* running_var, int32_t C,
                      const float* gamma, const float* beta, float eps, float* coeffs,
                      void* stream);
/* y = act((x - mean)*scale + beta + residual).  residual may be NULL; relu != 0 applies max(.,0).
 * x and y may alias (in-place).  */
int gs_bn_apply(const float* x, int64_t rows, int32_t C, int32_t ldx, const float* coeffs,
                const float* residual, int32_t ld_res, int32_t relu, float* y, int32_t ldy,
                void* stream);
/* gs_bn_apply with ReLU that also writes the mask bytes mask[rows][C/4] (bit e of byte q set <=>
 * y[r][4q+e] > 0), see gs_bn_args::relu_mask. */
int gs_bn_apply_mask(const float* x, int64_t rows, int32_t C, int32_t ldx, const float* coeffs,
                     const float* residual, int32_t ld_res, float* y, int32_t ldy, uint8_t* mask,
                     void* stream);

/* Backward.  mask_mode: 0 = none; 1 = ReLU directly after this BN (mask recomputed from
 * (x-mean)*scale+beta > 0); 2 = mask from a saved post-activation tensor `act` (> 0), used for the
 * bottleneck output relu(out + identity) where `act` is the block output.
 *   gs_bn_bwd_reduce : sums[2*C] = { sum g , sum g*xhat }, g = dy * mask.  If g_out != NULL the
 *                      masked gradient g is also written there (may alias dy): it is the gradient
 *                      of the identity branch.
 *   gs_bn_bwd_apply  : dx = scale * (g - sum_g/count - xhat * sum_gx/count)   (training stats)
 *                      or dx = scale * g when use_batch_stats == 0 (eval-mode BN in training).
 *                      Also writes dgamma[:C] = sum_gx, dbeta[:C] = sum_g when non-NULL. */
size_t gs_bn_bwd_workspace_bytes(int64_t rows, int32_t C);
int gs_bn_bwd_reduce(const float* dy, int32_t ld_dy, const float* x, int32_t ldx,
                     const float* act, int32_t ld_act, int64_t rows, int32_t C,
                     const float* coeffs, int32_t mask_mode, float* g_out, int32_t ld_g,
                     float* sums, void* workspace, size_t workspace_bytes, void* stream);
int gs_bn_bwd_apply(const float* dy, int32_t ld_dy, const float* x, int32_t ldx,
                    const float* act, int32_t ld_act, int64_t rows, int32_t C,
                    const float* coeffs, const float* sums, double count, int32_t mask_mode,
                    int32_t use_batch_stats, float* dx, int32_t ld_dx, float* dgamma,
                    float* dbeta, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* conv -> BatchNorm (+ residual) (+ ReLU) issued by ONE host call per direction               */
/* ------------------------------------------------------------------------------------------ */
/* Replaces the module chain conv -> norm -> activate of gaiavision DynamicConvModule and of each
 * conv/bn pair of DynamicBottleneck (gaiaseg/models/utils/dynamic_res_layer.py:84-125,
 * gaiaseg/models/backbones/dynamic_resnet.py:255-302, decode heads' ConvModules) for rank-local
 * BatchNorm.  Same kernels as gs_conv2d_*, gs_bn_* and gs_stream_fork; it exists because one host
 * call per module costs as much host time as the kernels cost GPU time on this path, and because
 * the batch statistics can then come from the conv itself: without split-K the conv epilogue writes
 * per-tile partial sums that are merged exactly (Chan et al., double, fixed order — equal to
 * gs_bn_stats up to rounding); with split-K one kernel sums the slabs, stores y and accumulates the
 * statistics (bit-identical to gs_conv2d_forward + gs_bn_stats_finalize).  Backward is a pure
 * composition.  workspace must be >= gs_conv_bn_workspace_bytes(d). */
typedef struct gs_bn_args {
  const float* gamma;        /* [>= C] or NULL (1)                                              */
  const float* beta;         /* [>= C] or NULL (0)                                              */
  float* running_mean;       /* [>= C] or NULL                                                  */
  float* running_var;        /* [>= C] or NULL                                                  */
  float eps, momentum;
  int32_t use_batch_stats;   /* 1: statistics of this batch (training); 0: running statistics   */
  int32_t update_running;    /* 1: update running_mean / running_var (training mode)            */
  int32_t relu;              /* ReLU after the BN (+ residual)                                  */
  int32_t reserved;          /* must be 0                                                       */
  uint8_t* relu_mask;        /* forward, relu != 0, z != NULL: if not NULL the apply pass also
                                writes the ReLU mask [rows][Co/4], one byte per channel quad
                                (bit e set <=> z[.., 4q+e] > 0) -- 1/16 of z, for the consumer's
                                fused BatchNorm-backward epilogue (gs_bn_bwd_fuse mode 3)          */
  const float* residual_coeffs; /* forward, residual != NULL: if not NULL, `residual` is the RAW
                                output of another conv (the projection shortcut of a stage's first
                                block, gaiaseg/models/utils/dynamic_res_layer.py:70-94) and these
                                are that conv's BatchNorm coefficients [scale | beta | mean | ..][C]
                                (layout of `coeffs`): the apply pass adds
                                (residual - mean) * scale + beta, so the shortcut's normalised
                                output is never written or read                                     */
} gs_bn_args;
/* max(gs_conv2d_workspace_bytes, gs_bn_stats_workspace_bytes) for this conv's output */
size_t gs_conv_bn_workspace_bytes(const gs_conv_desc* d);
/* y = conv(x, w) [N*Ho*Wo][ldy] (kept for backward); coeffs[4*Co] (kept for backward);
 * z = relu?(BN(y) (+ residual)) with pixel stride ldz.  The conv has no bias (norm follows).
 * z == NULL (residual must be NULL): only y and coeffs are produced; the consumer evaluates
 * relu(BN(y)) in its operand loader (gs_conv_desc::in_affine = coeffs).  d->in_affine of THIS conv is
 * honoured in forward and in the weight gradient of gs_conv_bn_backward. */
int gs_conv_bn_forward(const gs_conv_desc* d, const float* x, const float* w, const gs_bn_args* bn,
                       const float* residual, int32_t ld_res, float* y, float* coeffs, float* z,
                       int32_t ldz, void* workspace, size_t workspace_bytes, void* stream);
/* dz: gradient of z (in/out: overwritten with the masked gradient when write_g != 0 — that is the
 * gradient of the identity branch of a bottleneck); mask_mode as gs_bn_bwd_reduce (z is the saved
 * post-activation tensor for mode 2).  dy: scratch [N*Ho*Wo][ldy] receiving the gradient of the conv
 * output; bsums: scratch [2*Co].  dgamma/dbeta/dw/dx may be NULL (not needed).  side_stream != NULL
 * runs the weight gradient there (forked after dy is complete; the caller joins it later with
 * gs_stream_fork(side_stream, stream)) using side_workspace.
 *
 * Cross-layer fusion of the BatchNorm backward reduction (the bn_bwd_partial pass over dz and y):
 *   input_bn != NULL: x is the output of relu(bn_prev(y_prev) [+ residual]) and this call's data
 *     gradient is the LAST contribution to its gradient (accumulate_dx included).  Where the dgrad
 *     kernel allows it (stride 1, no split-K, fast path) its epilogue masks the gradient with that
 *     ReLU's mask, stores the masked gradient g into dx and reduces {sum g, sum g * xhat_prev} into
 *     input_bn->sums; *input_bn->fused is set to 1 (else 0 and dx holds the plain gradient).
 *   sums_ready != 0: THIS layer's dz already is the masked gradient and bsums already holds its
 *     sums (a consumer's call did the above): the reduction pass is skipped, mask_mode is ignored. */
typedef struct gs_bn_bwd_fuse {
  const float* y;        /* bn_prev's input [rows][ldy]                                         */
  const float* act;      /* mode 2: bn_prev's post-activation output [rows][ldact], else NULL   */
  const float* coeffs;   /* bn_prev's coefficient block [scale | beta | mean | invstd][Ci]      */
  float* sums;           /* out: [2*Ci] = {sum g, sum g * xhat}                                  */
  int32_t* fused;        /* out (host): 1 if the fused epilogue ran                             */
  int32_t ldy, ldact;
  int32_t mode;          /* 1: mask = bn_prev(y) > 0 ; 2: mask = act > 0 ; 3: mask bits (below)  */
  int32_t reserved;
  const uint8_t* mask;   /* mode 3: gs_bn_args::relu_mask of bn_prev's forward [rows][ldmask]:
                            the same mask as mode 2 without reading the activation again         */
  int32_t ldmask;        /* bytes per pixel row, >= Ci / 4                                       */
  int32_t reserved2;     /* must be 0                                                            */
} gs_bn_bwd_fuse;
int gs_conv_bn_backward(const gs_conv_desc* d, const float* x, const float* w, const float* y,
                        const float* z, int32_t ldz, const float* coeffs, const gs_bn_args* bn,
                        float* dz, int32_t ld_dz, int32_t mask_mode, int32_t write_g, float* dy,
                        float* bsums, float* dgamma, float* dbeta, float* dw, float* dx,
                        int32_t accumulate_dx, void* workspace, size_t workspace_bytes,
                        void* side_workspace, size_t side_workspace_bytes, void* stream,
                        void* side_stream, const gs_bn_bwd_fuse* input_bn, int32_t sums_ready);

/* Live timer of the forward launches with role GS_CONV_ROLE_BOTTLENECK3X3: bench.py's `roofline`.
 * One pair of HIP events per op on the launch stream.  Where the op is ONE kernel (unsplit, or split-K
 * combined inside the launch: the default) the pair is attached to that kernel's own dispatch
 * (hipExtLaunchKernelGGL start / stop events: the dispatch's begin / end timestamps, what a rocprofv3
 * kernel trace reports); where a reduce launch follows, or with GS_K3_TIMER_MARKERS set, the events are
 * markers recorded in front of the conv and behind the last launch (2-3 us more per op).
 * enable(1) resets and starts, enable(0) stops; read after a device synchronize. */
int gs_k3_timer_enable(int32_t on);
int gs_k3_timer_read(int64_t* launches, double* total_ms, double* total_flops);

/* ------------------------------------------------------------------------------------------ */
/* Pooling — K5, K7                                                                            */
/* ------------------------------------------------------------------------------------------ */
/* nn.AvgPool2d(kernel_size=s, stride=s, ceil_mode=True, count_include_pad=False): the shortcut
 * pooling of avg_down=True (gaiaseg/models/utils/dynamic_res_layer.py:75-82).  Output size
 * ceil(H/s) x ceil(W/s); border windows average their in-bounds pixels only.
 * backward: dx[n,h,w,:] (+)= dy[n,h/s,w/s,:] / count(window). */
int gs_avgpool_ceil_forward(const float* x, int32_t N, int32_t H, int32_t W, int32_t C, int32_t ldx,
                            int32_t s, float* y, int32_t ldy, void* stream);
int gs_avgpool_ceil_backward(const float* dy, int32_t ld_dy, int32_t N, int32_t H, int32_t W,
                             int32_t C, int32_t s, float* dx, int32_t ld_dx, int32_t accumulate,
                             void* stream);
/* nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (dynamic_resnet.py:302,413), generic k/s/p.
 * idx (uint8 tap index of the first maximum in (kh,kw) scan order, as ATen) is saved for bwd. */
int gs_maxpool_forward(const float* x, int32_t N, int32_t H, int32_t W, int32_t C, int32_t ldx,
                       int32_t k, int32_t s, int32_t p, int32_t Ho, int32_t Wo, float* y,
                       int32_t ldy, uint8_t* idx, void* stream);
/* dx (= or += when accumulate) gather of dy through idx.  */
int gs_maxpool_backward(const float* dy, int32_t ld_dy, const uint8_t* idx, int32_t N, int32_t H,
                        int32_t W, int32_t C, int32_t k, int32_t s, int32_t p, int32_t Ho,
                        int32_t Wo, float* dx, int32_t ld_dx, int32_t accumulate, void* stream);

/* nn.AdaptiveAvgPool2d(s) for all pool_scales of DynamicPPM in ONE read of x
 * (dynamic_psp_head.py:48-51).  scales[nscales] (e.g. {1,2,3,6}); y is the concatenation over
 * scales of [N][s][s][C] blocks (pixel stride C).  Bin edges floor(i*H/s), ceil((i+1)*H/s). */
size_t gs_adaptive_avgpool_workspace_bytes(int32_t N, int32_t H, int32_t W, int32_t C,
                                           const int32_t* scales, int32_t nscales);
int gs_adaptive_avgpool_forward(const float* x, int32_t N, int32_t H, int32_t W, int32_t C,
                                int32_t ldx, const int32_t* scales, int32_t nscales, float* y,
                                void* workspace, size_t workspace_bytes, void* stream);
/* dx (= or +=) sum over scales of dy_bin / bin_area for every bin containing the pixel. */
int gs_adaptive_avgpool_backward(const float* dy, int32_t N, int32_t H, int32_t W, int32_t C,
                                 const int32_t* scales, int32_t nscales, float* dx, int32_t ld_dx,
                                 int32_t accumulate, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Bilinear resize (align_corners False/True) — K9, K15, K16, K17                              */
/* ------------------------------------------------------------------------------------------ */
/* mmseg.ops.resize == F.interpolate(mode='bilinear') (dynamic_psp_head.py:67-71,
 * dynamic_uper_head.py:108-112,123-127).  Writes into a channel slice of a wider buffer (ldy) so
 * that torch.cat never materialises (PPM / FPN concat fusion); accumulate != 0 gives the UPer
 * top-down `laterals[i-1] += resize(laterals[i])` in one pass. */
int gs_bilinear_forward(const float* x, int32_t N, int32_t Hi, int32_t Wi, int32_t C, int32_t ldx,
                        int32_t Ho, int32_t Wo, int32_t align_corners, float* y, int32_t ldy,
                        int32_t accumulate, void* stream);
/* Adjoint (deterministic gather form): dx (= or +=) sum_p w_p * dy_p.  Large footprints (small
 * source maps) are split over destination-row slices into the workspace and summed in order. */
size_t gs_bilinear_backward_workspace_bytes(int32_t N, int32_t Hi, int32_t Wi, int32_t C,
                                            int32_t Ho, int32_t Wo);
int gs_bilinear_backward(const float* dy, int32_t ld_dy, int32_t N, int32_t Hi, int32_t Wi,
                         int32_t C, int32_t Ho, int32_t Wo, int32_t align_corners, float* dx,
                         int32_t ld_dx, int32_t accumulate, void* workspace,
                         size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Elementwise helpers                                                                         */
/* ------------------------------------------------------------------------------------------ */
/* dst[r, :C] (= or +=) src[r, :C] * alpha for strided 2-D views (concat slices, grad accumulation). */
int gs_copy2d(const float* src, int32_t ld_src, float* dst, int32_t ld_dst, int64_t rows,
              int32_t C, float alpha, int32_t accumulate, void* stream);
/* nn.Dropout2d (fcn_head.py:248-253): y[n,p,c] = x[n,p,c] * mask[n*C + c]; mask already holds
 * keep/(1-p).  Same kernel is its own backward.  x, y may alias. */
int gs_scale_nc(const float* x, int32_t ldx, const float* mask, int32_t N, int64_t pixels_per_image,
                int32_t C, float* y, int32_t ldy, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Pixel-wise cross entropy with on-the-fly bilinear upsampling — K13, K14                     */
/* ------------------------------------------------------------------------------------------ */
/* Replaces resize(seg_logit -> label size) + F.cross_entropy(reduction='none', ignore_index) +
 * weight_reduce_loss(mean over ALL pixels) + accuracy(top-1)
 * (gaiaseg/models/decode_heads/dynamic_fcn_head.py:137-159; spec copies
 *  gaiaseg/models/losses/cross_entropy_loss.py:81-92, utils.py:26-55, accuracy.py:38-49).
 * logits: [N,h,w,Cls] through element strides (l_sn,l_sh,l_sw,l_sc); labels int64 [N,H,W].
 * pixel_weight: optional float [N,H,W] (OHEM sampler weights), class_weight optional [Cls].
 * out[0] = sum_i w_i * ce_i (host divides by N*H*W and multiplies loss_weight),
 * out[1] = number of pixels whose argmax == label.     (both accumulated in double, fixed order)
 * The full-resolution logits are never materialised. */
typedef struct gs_ce_desc {
  int32_t N, h, w, Cls;      /* logits                                                        */
  int32_t H, W;              /* label / loss resolution                                       */
  int64_t l_sn, l_sh, l_sw, l_sc;
  int32_t ignore_index;
  int32_t align_corners;
} gs_ce_desc;
size_t gs_ce_workspace_bytes(const gs_ce_desc* d);
/* lse: optional float [N,H,W]; when non-NULL the per-pixel log-sum-exp of the resized logits is
 * saved there for gs_ce_backward (which requires it). */
int gs_ce_forward(const gs_ce_desc* d, const float* logits, const int64_t* labels,
                  const float* pixel_weight, const float* class_weight, float* lse, double* out,
                  void* workspace, size_t workspace_bytes, void* stream);
/* Same, with the host-side epilogue of `losses` folded in: out2[0] = float(out[0]) * loss_scale
 * (= loss_weight / (N*H*W)), out2[1] = float(out[1]) * acc_scale (= 100 / (N*H*W)), both fp32 —
 * dynamic_fcn_head.py:149-159 computes exactly these two scalars. */
int gs_ce_forward_scaled(const gs_ce_desc* d, const float* logits, const int64_t* labels,
                         const float* pixel_weight, const float* class_weight, float* lse,
                         float loss_scale, float acc_scale, float* out2, void* workspace,
                         size_t workspace_bytes, void* stream);
/* dlogits[n,y,x,c] = grad_scale * sum_p bilinear_w(p->(y,x)) * w_p * cw * (softmax_p[c] - [c==label_p])
 * written densely (pixel stride ld_d, columns Cls..ld_d-1 zeroed).  Deterministic gather form:
 * one workgroup per low-resolution logit pixel walks its bilinear footprint. */
int gs_ce_backward(const gs_ce_desc* d, const float* logits, const int64_t* labels,
                   const float* pixel_weight, const float* class_weight, const float* lse,
                   float grad_scale, float* dlogits, int32_t ld_d, void* stream);
/* Same result through the tile form when the up-scaling is a power-of-two integer factor with
 * align_corners = 0 (every mmseg head of this path: x8 / x16 / x32): one workgroup per tile of
 * full-resolution pixels between four low-resolution logit pixels evaluates each softmax term once
 * (the gather form of gs_ce_backward evaluates it once per neighbour, i.e. four times), then a
 * fixed-order gather adds the four corner sums per low-resolution pixel.  Any OTHER up-scaling of at
 * most 64 full-resolution pixels per tile on average (config 4's 193 -> 769, align_corners either way)
 * takes the row-tile form (r04): the same tiles, bounds found with the resize's own source-index
 * arithmetic, one 16-lane DPP row per tile.  Falls back to gs_ce_backward otherwise (larger
 * non-power-of-two ratios, down-scaling).  workspace >= gs_ce_backward_workspace_bytes(d, ld_d), which
 * is 0 exactly when the gather form will be used. */
size_t gs_ce_backward_workspace_bytes(const gs_ce_desc* d, int32_t ld_d);
int gs_ce_backward_ws(const gs_ce_desc* d, const float* logits, const int64_t* labels,
                      const float* pixel_weight, const float* class_weight, const float* lse,
                      float grad_scale, float* dlogits, int32_t ld_d, void* workspace,
                      size_t workspace_bytes, void* stream);
/* OHEMPixelSampler support: prob[n,Y,X] = softmax(resized logit)[label]
 * for valid pixels, 2.0 for ignored ones (so they sort last). */
int gs_ce_label_prob(const gs_ce_desc* d, const float* logits, const int64_t* labels, float* prob,
                     void* stream);

/* OHEMPixelSampler.sample (mmseg): from the label probabilities written by
 * gs_ce_label_prob, weight[i] = 1 for the hard valid pixels, else 0.
 *   use_thresh != 0 : threshold = max(thresh, sorted_valid_prob[min(batch_kept, n_valid-1)]),
 *                     keep prob < threshold          (OHEM with `thresh`)
 *   use_thresh == 0 : keep the batch_kept valid pixels of smallest probability (= largest loss)
 * The rank statistic is an exact 3-pass radix select (integer atomics only; reproducible). */
size_t gs_ohem_workspace_bytes(void);
int gs_ohem_weights(const float* prob, int64_t n, int64_t batch_kept, float thresh,
                    int32_t use_thresh, float* weight, void* workspace, size_t workspace_bytes,
                    void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Inference epilogue — K17                                                                    */
/* ------------------------------------------------------------------------------------------ */
/* argmax over classes of bilinearly resized logits (softmax is monotone):
 * seg[n,Y,X] = argmax_c resize(logits)[n,c,Y,X]  (dynamic_distiller.py:461-521 whole_inference +
 * simple_test).  If probs != NULL also writes softmax probabilities [N,H,W,Cls] (aug_test). */
int gs_resize_argmax(const gs_ce_desc* d, const float* logits, int64_t* seg, float* probs,
                     void* stream);

/* Whole-image and sliding-window test mode as ONE gather kernel
 * (dynamic_distiller.py:416-459 slide_inference, :461-473 whole_inference, :475-508 inference,
 * :510-540 simple_test / aug_test).  The reference accumulates every window's up-sampled logits
 * into a [N,C,H,W] tensor (159 MB at 1024x2048), counts, divides, resizes to ori_shape, applies
 * softmax, flips and takes the argmax in separate passes.  Here the low-resolution logits of all
 * windows stay resident and each output pixel gathers them: nothing of size C*H*W exists unless
 * the caller asks for the probabilities.
 *   logits : [ny*nx][N][hl][wl][ld] fp32 (NHWC per window, window index row-major, ld % 4 == 0,
 *            ld >= C): the decode head's output for every window of hc x wc image pixels
 *   win_y / win_x : HOST arrays of the ny window-row / nx window-column origins (<= 64 each); the
 *            window list is their product.  Whole mode: ny = nx = 1, origin 0, hc = H, wc = W.
 *   pred(y,x) = sum over covering windows of bilinear(logits_win)(y - y0, x - x0) / cover count
 *   out = resize(pred, (Ho, Wo)) (identity when Ho == H and Wo == W: `rescale` to ori_shape)
 *   p = softmax_c(out), read at the mirrored position when flip = 1 (horizontal) / 2 (vertical)
 *   probs_out[N][C][Ho][Wo] = (probs_in ? probs_in : 0) + p     (either may be NULL; may alias)
 *   labels[N][Ho][Wo]       = argmax_c of that sum (of the logits when no probabilities are used)
 * Returns GS_E_BADARG if a window leaves the image or a pixel is not covered. */
typedef struct gs_slide_desc {
  int32_t N, C, ld;       /* images per window, classes, padded class stride of logits */
  int32_t hl, wl;         /* low-resolution logits size of one window */
  int32_t hc, wc;         /* window size in image pixels */
  int32_t H, W;           /* image size */
  int32_t Ho, Wo;         /* output size */
  int32_t ny, nx;         /* window rows / columns */
  int32_t align_corners;
  int32_t flip;           /* 0 none, 1 horizontal, 2 vertical */
  int32_t reserved;       /* must be 0 */
} gs_slide_desc;
int gs_slide_fuse(const gs_slide_desc* d, const int32_t* win_y, const int32_t* win_x,
                  const float* logits, const float* probs_in, float* probs_out, int64_t* labels,
                  void* stream);
/* Test / A-B hook: label-only, un-rescaled gs_slide_fuse calls walk strips of 4 output pixels that
 * share their low-resolution cell's corner vectors (mode 1, the default) or use the per-pixel kernel
 * (mode 0); -1 = back to the GS_SLIDE_STRIP environment value.  Both give bit-identical label maps. */
int gs_debug_set_slide_strip(int32_t mode);

/* mIoU evaluation support: conf[label*C + pred] += 1 over the pixels
 * whose label != ignore_index; conf is [C*C] uint64, accumulated (zero it before the first call).
 * Replaces mmseg's intersect_and_union histogramming behind
 * gaiaseg/core/evaluation/cross_arch_eval_hooks.py:85-92. */
int gs_confusion_matrix(const int64_t* pred, const int64_t* label, int64_t n, int32_t num_classes,
                        int32_t ignore_index, uint64_t* conf, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Optimiser — K18                                                                             */
/* ------------------------------------------------------------------------------------------ */
/* torch.optim.SGD(momentum, weight_decay, dampening=0, nesterov=False) over flat fp32 arenas
 * (cfg optimizer: configs/_dynamic_/models/pspnet_ar50to101v2_gsync.py:175).  One call updates
 * one contiguous element range [0,n) of the arenas; the host merges the parameters used this step
 * into a few ranges (parameters of depth-skipped blocks are not in any range: no update at all).
 * momentum_buf starts at zero, so the first use gives buf = g exactly as torch does.
 *   g = grad*grad_scale + weight_decay*p ; buf = momentum*buf + g ; p -= lr*buf
 * grad_scale is 1/world_size for the data-parallel mean. n % 4 == 0, pointers 16-byte aligned.
 * zero_grad != 0 clears grad after use (optimizer.zero_grad() of the NEXT iteration folded in:
 * OptimizerHook's zero_grad -> backward -> step order). */
int gs_sgd_step(float* param, float* grad, float* momentum_buf, int64_t n, float lr,
                float momentum, float weight_decay, float grad_scale, int32_t zero_grad, void* stream);
/* Same update with the hyper-parameters read from DEVICE memory when the kernel runs:
 * hyper = {lr, momentum, weight_decay, grad_scale} (16-byte aligned).  The launch carries no
 * step-dependent value, so a hipGraph captured around a whole training step (core/runner.py) can be
 * replayed under the poly learning-rate schedule (PolyLrUpdaterHook) by rewriting 16 bytes. */
/* Writes {lr, momentum, weight_decay, grad_scale} into that buffer in stream order (the values travel
 * as kernel arguments: unlike an asynchronous copy from a reused host buffer, a launch that is still
 * queued when the host moves on to the next step keeps its own values). */
int gs_sgd_set_hyper(float* hyper, float lr, float momentum, float weight_decay, float grad_scale,
                     void* stream);
int gs_sgd_step_hyper(float* param, float* grad, float* momentum_buf, int64_t n, const float* hyper,
                      int32_t zero_grad, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Training input pipeline                                             */
/* ------------------------------------------------------------------------------------------ */
/* One sample of the reference's train_pipeline (configs/_dynamic_/models/
 * pspnet_ar50to101v2_gsync.py:60-75: Resize(ratio_range) -> RandomCrop -> RandomFlip ->
 * PhotoMetricDistortion -> Normalize(to_rgb) -> Pad -> DefaultFormatBundle) as one gather kernel
 * from the ORIGINAL uint8 image [src_h][src_w][3] and label map [src_h][src_w] on the device to
 * out_img fp32 [3][out_h][out_w] and out_label int64 [out_h][out_w] (may be NULL).  The random
 * decisions are made by the caller and passed in the descriptor:
 *   res_h, res_w      size of the (virtual) resized image        (mmcv.imrescale)
 *   crop_y/x, crop_h/w the crop window inside it, crop <= out     (RandomCrop; the rest is padding)
 *   flip              horizontal flip of the crop                 (RandomFlip)
 *   pm_*              PhotoMetricDistortion: brightness delta, contrast alpha (first = mode 1),
 *                     saturation alpha, integer hue delta (uint8 HSV, H in [0,180))
 *   mean/std/to_rgb   Normalize; pad_val / seg_pad_val            (Pad)
 * Resize: image bilinear with half-pixel centres rounded to uint8, label nearest. */
typedef struct gs_augment_desc {
  int32_t src_h, src_w, src_is_rgb;
  int32_t res_h, res_w;
  int32_t crop_y, crop_x, crop_h, crop_w;
  int32_t out_h, out_w;
  int32_t flip;
  int32_t pm_enable, pm_brightness, pm_contrast, pm_contrast_first, pm_saturation, pm_hue;
  float pm_delta, pm_alpha, pm_sat_alpha;
  int32_t pm_hue_delta;
  int32_t to_rgb;
  float mean[3], std[3];
  float pad_val;
  int32_t seg_pad_val;
} gs_augment_desc;
int gs_seg_augment(const gs_augment_desc* d, const uint8_t* img, const uint8_t* label,
                   float* out_img, int64_t* out_label, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Tuning hook (tools/sweep_conv_plans.py): force tile rows (64|128), tile columns            */
/* (32|48|64|80|96|128) and split-K factor of the following gs_conv2d_* calls; bm = 0 restores */
/* the planner.  Process-global, not thread-safe, results stay exact (only the fixed summation */
/* order of split-K changes).                                                                  */
/* ------------------------------------------------------------------------------------------ */
int gs_debug_force_plan(int32_t bm, int32_t bn, int32_t splits);
/* The tile / split-K plan the library would choose for an M x N x K implicit GEMM (max_splits: 64 for
 * forward / dgrad, 512 for wgrad).  Host arithmetic only (no GPU needed). */
int gs_debug_query_plan(int32_t M, int32_t N, int32_t K, int32_t max_splits, int32_t* bm,
                        int32_t* bn, int32_t* splits, int32_t* ksteps_per_split);

/* Which kernel family and K loop an implicit-GEMM launch used.  The parity tests assert it, so that a
 * comparison against F.conv2d is known to have exercised the loop it claims to (the stride-1 data
 * gradient of every bottleneck conv, gaiaseg/models/utils/dynamic_res_layer.py:105-125, contracts on
 * GS_KLOOP_BF16X3 when its grid is large enough and on the fp32 loops otherwise). */
#define GS_OP_FORWARD 0
#define GS_OP_DGRAD 1
#define GS_OP_WGRAD 2
#define GS_KLOOP_GENERIC 0      /* igemm_rows_kernel / igemm_wgrad_kernel: any shape, sectioned loop  */
#define GS_KLOOP_FP32 1         /* fast kernels, v_mfma_f32_16x16x4_f32, one K step per barrier       */
#define GS_KLOOP_FP32_PAIRS 2   /* the same, two K steps per barrier                                  */
#define GS_KLOOP_BF16X3 3       /* six v_mfma_f32_16x16x32_bf16 over an exact 3-way bf16 split        */
#define GS_KLOOP_STREAM 4       /* 1x1 streaming kernel: weights resident in LDS, persistent over rows  */
#define GS_KLOOP_COUNT 5
typedef struct gs_debug_launch {
  int32_t op;                   /* GS_OP_*                                                            */
  int32_t kloop;                /* GS_KLOOP_*                                                         */
  int32_t bm, bn, splits, ksteps_per_split;
  int32_t in_affine;            /* 1: relu(bn(x)) evaluated in the operand loader                     */
  int32_t bn_bwd_mode;          /* dgrad: gs_bn_bwd_fuse mode folded into the tile epilogue (0 none)   */
} gs_debug_launch;
/* The most recent conv launch issued by the calling thread (a strided dgrad reports its last parity
 * class).  GS_E_BADARG if the thread has not launched any. */
int gs_debug_last_conv_launch(gs_debug_launch* out);
/* counts[(op * GS_KLOOP_COUNT + kloop) * 3 + bn_bwd_mode] = launches since the last reset
 * (process-wide, 45 entries); reset != 0 clears after reading.  counts may be NULL (reset only). */
int gs_debug_conv_launch_counts(int64_t* counts, int32_t reset);
/* flops[op * GS_KLOOP_COUNT + kloop] (15 entries) = algorithmic FLOPs (2 * M * N * K of the implicit GEMM, padding not counted)
 * launched since the last reset: bench.py states which share of a step's contraction work ran on
 * which MFMA path, so that its roofline fractions name the right bound.  The forward thread and the
 * autograd thread may both launch: the counters are updated atomically. */
int gs_debug_conv_launch_flops(double* flops, int32_t reset);
/* Forward convolutions on the bf16x3 K loop: 0 = never (fp32 MFMA loops), 1 = the shapes where it
 * measured ahead (3x3, not where the two-steps-per-barrier fp32 loop runs unsplit), 2 = every launch
 * the loop's gate admits (the operator tests), 3 = the split-K 3x3s only (the default: see
 * csrc/igemm_core.h x3_fwd_mode for the parity margins behind that choice), -1 = back to the
 * GS_X3_FWD environment value.  Process-global. */
int gs_debug_set_x3_fwd(int32_t mode);
/* Split-K launches of the forward / data-gradient row kernels: 1 = every output tile's partial slabs
 * are summed inside the launch by the tile's last-arriving workgroup, which then runs the unsplit
 * epilogue (default; csrc/igemm_core.h splitk_publish), 0 = a separate reduce launch sums them,
 * -1 = back to the GS_SPLITK_INKERNEL environment value.  Both sum in split order: the outputs are
 * bit-identical.  Process-global. */
int gs_debug_set_splitk_inkernel(int32_t mode);
/* Split-K row launches since the last reset that combined their slabs inside the launch (the parity
 * tests check that the shapes they compare really took that path); reset != 0 clears the count. */
int64_t gs_debug_splitk_combined(int32_t reset);
/* The per-tile partials a fused conv + BatchNorm call leaves (BatchNorm statistics of the forward,
 * BatchNorm-backward sums of a data gradient with gs_bn_bwd_fuse).  mode is a mask — 1: the forward
 * statistics, 2: the data gradient's sums, 3: both —; a set bit means launches with at most
 * GS_COL_FINALIZE_MAX (160) row tiles merge them inside the launch — the last workgroup of every
 * column tile writes the coefficients / sums (csrc/igemm_core.h column_finalize_*), 0 = a separate
 * launch merges them (the default: the in-launch form measured neutral on the training step, every
 * workgroup pays the arrival hand-off), -1 = back to the GS_COL_FINALIZE environment value.
 * Process-global.
 * gs_debug_col_finalized: launches since the last reset that merged their partials themselves. */
int gs_debug_set_col_finalize(int32_t mode);
int64_t gs_debug_col_finalized(int32_t reset);
/* Compute units the planners assume: hipDeviceProp::multiProcessorCount of the current device, read
 * once at first use (256 on an MI355X; 256 is also assumed when no device is present). */
int gs_debug_num_cu(void);
/* Dispatch of the streaming 1x1 kernel (csrc/igemm_stream.h): 0 = never, 1 = the shapes where it
 * measured ahead of the tile kernels (default), 2 = every shape whose weights fit in LDS (the operator
 * tests cover all of its code paths this way), -1 = back to the GS_STREAM environment value.
 * Process-global, not thread-safe. */
int gs_debug_set_stream_mode(int32_t mode);
/* The same for the forward launches with role GS_CONV_ROLE_BOTTLENECK3X3 only: flops[kloop], 5 entries
 * (which bound bench.py's headline `roofline` is priced against). */
int gs_debug_k3_flops(double* flops, int32_t reset);
/* What gs_conv2d_forward / _dgrad / _wgrad (op = GS_OP_*) WOULD launch for this descriptor: host
 * arithmetic only, no GPU needed (honours gs_debug_force_plan and the GS_X3 switches).  For a strided
 * dgrad it describes the parity class (0, 0). */
int gs_debug_query_conv_launch(const gs_conv_desc* d, int32_t op, gs_debug_launch* out);

/* ------------------------------------------------------------------------------------------ */
/* Stream fork / join: work enqueued on `to` after this call waits for everything enqueued on  */
/* `from` before it (hipEventRecord + hipStreamWaitEvent on an internal event ring).  Used to   */
/* run the weight-gradient kernels on a side stream beside BN-backward / dgrad (the reference   */
/* runs all of backward on one stream: torch autograd, gaiaseg/apis/train.py:88-96).            */
/* ------------------------------------------------------------------------------------------ */
int gs_stream_fork(void* from_stream, void* to_stream);

#ifdef __cplusplus
}
#endif
#endif /* GAIASEG_HIP_H */
