/*
 * amyloid_yolo.h -- C ABI of libamyloid_yolo_hip.so (gfx950 / MI355X).
 *
 * The reference (keiserlab/amyloid-yolo-paper) is pure Python/PyTorch and has no FFI for this path
 * (SURVEY.md §8b); its "operator interface" is the Python call surface of models.py / utils/utils.py.
 * Each entry point below names the reference code it replaces.  The host-side mirror of that Python
 * surface lives in amyloid_yolo_paper_amd/{models,utils}.py and binds these symbols with ctypes
 * (INTEGRATION.md shows the stub).
 *
 * Conventions: every function returns 0 on success or a negative AY_ERR_* code; ay_last_error() gives
 * a thread-local message.  All buffers are caller-allocated DEVICE pointers (the library never
 * allocates or frees), every call is stream-ordered on `stream` (a hipStream_t passed as void*) and
 * performs no host synchronisation.  No torch types appear here.
 *
 * Activation layouts
 *   "blocked bf16"  [B][C/16][H][W][16] bfloat16   -- the MFMA path ("c16" planes; C padded to 16)
 *   "blocked f32"   [B][C/16][H][W][16] float      -- head outputs of the bf16 path
 *   "nchw f32"      [B][C][H][W] float             -- the reference's layout; fp32 parity path + I/O
 */
#ifndef AMYLOID_YOLO_H
#define AMYLOID_YOLO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AY_OK 0
#define AY_ERR_ARG (-1)      /* bad argument / unsupported shape */
#define AY_ERR_LAUNCH (-2)   /* HIP launch error */
#define AY_ERR_WORKSPACE (-3)

typedef void* ay_stream_t; /* hipStream_t */

/* 16-bit storage type of the MFMA path's activations and packed filters.  Same layouts, bytes and MFMA rate; bfloat16 keeps
 * fp32's exponent range (training, default), IEEE half (inference only: the *_f16 entry points below) has an 11-bit
 * significand -- an 8x smaller rounding step per stored activation (BASELINE.json configs[4] "fp16 MFMA path") -- and is
 * finite up to 65504: stored activations are post-BatchNorm, heads leave the path as fp32. */
#define AY_DT_BF16 0
#define AY_DT_F16 1

/* ABI version of this header: bumped whenever an exported signature changes (2: ay_plan_create gained `act_dtype` in the middle of
 * its argument list).  A binding compares ay_version() with the version it was written against before the first call, so that a
 * stale libamyloid_yolo_hip.so (or an out-of-tree caller of the old 7-argument form) fails at load time instead of passing a pointer
 * as a dtype (amyloid_yolo_paper_amd/_lib.py: ABI_VERSION). */
#define AY_ABI_VERSION 2
int ay_version(void);
const char* ay_last_error(void);

/* One convolutional block: conv -> (BN affine | bias) -> LeakyReLU(0.1)? -> (+ residual)?
 * Replaces models.py:26-45 executed at models.py:242-243, with the following shortcut
 * (models.py:246-248) fused as `residual`. */
typedef struct ay_conv_desc {
    int32_t batch;
    int32_t cin, cout;         /* logical channels */
    int32_t hin, win;          /* input spatial size */
    int32_t hout, wout;        /* output spatial size = floor((hin + 2*pad - k)/stride) + 1, pad=(k-1)/2 */
    int32_t ksize, stride;     /* 1|3 ; 1|2 */
    int32_t leaky;             /* 1: LeakyReLU(0.1) after the affine */
    int32_t out_f32;           /* bf16 path only: 1 = store blocked f32 (linear heads), 0 = blocked bf16 */
    int32_t cout_pad;          /* channels in scale/shift/packed weights/output planes: multiple of 16 (bf16 path: 32) */
} ay_conv_desc;

/* ---- bf16 MFMA path -------------------------------------------------------------------------- */

/* OIHW fp32 weights -> packed bf16 [cin/16][k*k][2][cout_pad][8] (zero rows for cout..cout_pad). */
int ay_pack_conv_weights_bf16(const float* w_oihw, void* packed, int cout, int cout_pad, int cin, int ksize,
                              ay_stream_t stream);
size_t ay_packed_weight_bytes(int cout_pad, int cin, int ksize);

/* BN (eval) -> per-channel scale/shift (models.py:43 semantics, eps passed in); bias-only layers use
 * gamma=NULL: scale=1, shift=bias.  Pads [n, n_pad) with scale=0, shift=0. */
int ay_fold_bn(const float* gamma, const float* beta, const float* mean, const float* var, const float* bias,
               float eps, float* scale, float* shift, int n, int n_pad, ay_stream_t stream);

/* Stem: nchw f32 image [B,3,H,W] -> blocked bf16 [B][2][H][W][16] (3x3 s1 conv 3->32, fp32 math). */
int ay_stem_conv_fwd(const float* x_nchw, const float* w_oihw, const float* scale, const float* shift,
                     void* out_blocked, int batch, int h, int w, int leaky, ay_stream_t stream);

/* Layer 0 + layer 1 of the Darknet-53 stem fused (3x3 s1 3->32, then 3x3 s2 32->64, each with BN affine + leaky): the
 * 32-channel full-resolution intermediate stays in LDS.  stem_w_bf16: [32][32] bf16, k = ci*9+kh*3+kw (27..31 zero);
 * w1_packed: ay_pack_conv_weights_bf16 of the 64x32x3x3 filters (cout_pad 64); out: blocked bf16 [B][4][H/2][W/2][16].
 * The image and the stem filters enter the MFMA as bf16 (fp32 accumulate). */
int ay_stem_s2_fused_fwd(const float* x_nchw, const void* stem_w_bf16, const float* scale0, const float* shift0, int leaky0,
                         const void* w1_packed, const float* scale1, const float* shift1, int leaky1, void* out_blocked,
                         int batch, int h, int w, ay_stream_t stream);

/* 3x3 (stride 1|2) and 1x1 convolution, blocked bf16 in, MFMA 32x32x16 bf16, fp32 accumulate,
 * fused scale/shift + leaky + residual epilogue; `residual` (blocked bf16, output shape) may be NULL. */
int ay_conv_fwd_bf16(const ay_conv_desc* d, const void* src, const void* w_packed, const float* scale,
                     const float* shift, const void* residual, void* out, ay_stream_t stream);

/* The 3x3 stride-1 member of ay_conv_fwd_bf16 for cout_pad % 128 == 0, cin % 32 == 0, at least 16 output rows: the same
 * block (models.py:26-45, 246-248) on v_mfma_f32_16x16x32_bf16, which the chip clocks higher than the 32x32x16 form on
 * non-trivial data.  ay_conv_fwd_bf16 dispatches to it (AY_M16=0 in the environment keeps the 32x32x16 kernel); same
 * arguments, same rounding contract (bf16 operands, fp32 accumulation in a different order, one rounding per output). */
int ay_conv3x3_m16_fwd_bf16(const ay_conv_desc* d, const void* src, const void* w_packed, const float* scale, const float* shift,
                            const void* residual, void* out, ay_stream_t stream);

/* The same 1x1 convolution over a route that is never materialised (models.py:86-96,244-245): input channels
 * [0, c1) come from src1_halfres ([B][c1/16][H/2][W/2][16], nearest x2 upsample folded into the loader), channels
 * [c1, cin) from src2 ([B][(cin-c1)/16][H][W][16]); c1 and cin-c1 multiples of 64, cout_pad a multiple of 128. */
int ay_conv1x1_cat_fwd_bf16(const ay_conv_desc* d, const void* src1_halfres, int c1, const void* src2, const void* w_packed,
                            const float* scale, const float* shift, void* out, ay_stream_t stream);

/* Fused Darknet-53 residual block (models.py:26-45 twice + the shortcut at :246-248):
 * out = leaky(bn2(conv3x3(leaky(bn1(conv1x1(x)))))) + x, channels C -> C/2 -> C, in one kernel: the C/2-channel
 * intermediate stays in LDS.  x/out blocked bf16 [B][C/16][H][W][16] (out != x); w1_packed = ay_pack_conv_weights_bf16 of
 * the [C/2][C][1][1] filters (cout_pad C/2), w2_packed of the [C][C/2][3][3] filters (cout_pad C); scale/shift from
 * ay_fold_bn.  Same rounding points as the two ay_conv_fwd_bf16 calls it replaces (one bf16 rounding of the intermediate,
 * one of the output).  ay_resblock_supported(C) says whether C has a fused kernel (64 and 128); other blocks use the
 * two-call path. */
int ay_resblock_supported(int channels);
int ay_resblock_fwd_bf16(const void* x, const void* w1_packed, const float* scale1, const float* shift1, int leaky1,
                         const void* w2_packed, const float* scale2, const float* shift2, int leaky2, void* out, int batch,
                         int channels, int h, int w, ay_stream_t stream);

/* route (models.py:244-245) + nearest x2 upsample (models.py:86-96) as one gather into a blocked bf16
 * tensor [B][(c1+c2)/16][H][W][16]: src1 [B][c1/16][H>>up1][W>>up1][16], src2 [B][c2/16][H][W][16]|NULL. */
int ay_concat_upsample_bf16(const void* src1, int c1, int up1, const void* src2, int c2, void* out,
                            int batch, int h, int w, ay_stream_t stream);

/* layout converters (tests, fp32<->bf16 path bridges) */
int ay_blocked_bf16_to_nchw_f32(const void* src, float* dst, int batch, int c, int h, int w, ay_stream_t stream);
int ay_blocked_f32_to_nchw_f32(const float* src, float* dst, int batch, int c, int h, int w, ay_stream_t stream);
int ay_nchw_f32_to_blocked_bf16(const float* src, void* dst, int batch, int c, int h, int w, ay_stream_t stream);

/* ---- the same inference entry points on IEEE half ("blocked f16": [B][C/16][H][W][16] half; packed filters half) ----------
 * Identical arguments, layouts, fusion and rounding points (operands 16-bit, fp32 accumulate, fp32 epilogue, ONE rounding per
 * stored activation -- to half instead of bfloat16); v_mfma_f32_{32x32x16,16x16x32}_f16 in place of the _bf16 forms.
 * ay_concat_upsample_bf16 moves 16-bit elements untouched and serves both types.  Replaces the same reference lines as the
 * _bf16 entry point of the same name (models.py:26-45, 86-96, 244-248). */
int ay_pack_conv_weights_f16(const float* w_oihw, void* packed, int cout, int cout_pad, int cin, int ksize, ay_stream_t stream);
int ay_stem_conv_fwd_f16(const float* x_nchw, const float* w_oihw, const float* scale, const float* shift,
                         void* out_blocked, int batch, int h, int w, int leaky, ay_stream_t stream);
/* stem_w_f16: [32][32] half, k = ci*9+kh*3+kw (27..31 zero) */
int ay_stem_s2_fused_fwd_f16(const float* x_nchw, const void* stem_w_f16, const float* scale0, const float* shift0, int leaky0,
                             const void* w1_packed, const float* scale1, const float* shift1, int leaky1, void* out_blocked,
                             int batch, int h, int w, ay_stream_t stream);
int ay_conv_fwd_f16(const ay_conv_desc* d, const void* src, const void* w_packed, const float* scale,
                    const float* shift, const void* residual, void* out, ay_stream_t stream);
int ay_conv3x3_m16_fwd_f16(const ay_conv_desc* d, const void* src, const void* w_packed, const float* scale, const float* shift,
                           const void* residual, void* out, ay_stream_t stream);
int ay_conv1x1_cat_fwd_f16(const ay_conv_desc* d, const void* src1_halfres, int c1, const void* src2, const void* w_packed,
                           const float* scale, const float* shift, void* out, ay_stream_t stream);
int ay_resblock_fwd_f16(const void* x, const void* w1_packed, const float* scale1, const float* shift1, int leaky1,
                        const void* w2_packed, const float* scale2, const float* shift2, int leaky2, void* out, int batch,
                        int channels, int h, int w, ay_stream_t stream);
int ay_blocked_f16_to_nchw_f32(const void* src, float* dst, int batch, int c, int h, int w, ay_stream_t stream);
int ay_nchw_f32_to_blocked_f16(const float* src, void* dst, int batch, int c, int h, int w, ay_stream_t stream);

/* ---- fp32 parity path (reference layout) ------------------------------------------------------ */

/* Same block in nchw f32 with OIHW fp32 weights; src2/c-split/up1 implement route+upsample in the
 * loader: channels [0,cin1) come from src1 read at (y>>up1, x>>up1), [cin1,cin) from src2. */
int ay_conv_fwd_f32(const ay_conv_desc* d, const float* src1, int cin1, int up1, const float* src2,
                    const float* w_oihw, const float* scale, const float* shift, const float* residual,
                    float* out, ay_stream_t stream);
/* The same block restricted to the VALU kernel (one fmaf chain per output in (ci, kh, kw) order; no matrix cores): the fp32
 * training engine's forward, whose gradients are pinned element-wise by the reference's training fixtures on that summation
 * order.  ay_conv_fwd_f32 itself runs the cfg format's shapes on exact-fp32 MFMA (v_mfma_f32_32x32x2_f32). */
int ay_conv_fwd_f32_valu(const ay_conv_desc* d, const float* src1, int cin1, int up1, const float* src2,
                    const float* w_oihw, const float* scale, const float* shift, const float* residual,
                    float* out, ay_stream_t stream);

/* ---- YOLO head decode: models.py:127-172 ------------------------------------------------------- */
/* head: blocked f32 [B][cpad/16][G][G][16], cpad = A*(5+C) rounded up to 32 (layout=1) or nchw f32 [B][A*(5+C)][G][G] (layout=0).
 * Writes rows [row_offset, row_offset + A*G*G) of out [B][n_total][5+C]:
 * (sig(tx)+gx)*s, (sig(ty)+gy)*s, exp(tw)*aw, exp(th)*ah, sig(conf), sig(cls...) ; s = img_dim/G. */
int ay_yolo_decode(const float* head, int layout, float* out, int batch, int num_anchors, int num_classes,
                   int grid, int img_dim, const float* anchors_wh /* host, A*2, pixels */, int n_total,
                   int row_offset, ay_stream_t stream);
/* A detection head AND its decode in one launch (models.py:33-40 for the linear 1x1 head convolution 81 / 93 / 105, then
 * models.py:127-172): the head tensor never goes to memory, the convolution's epilogue writes rows [row_offset, row_offset + A*G*G)
 * of pred [B][n_total][5+C] -- the same bits ay_conv_fwd_bf16 (out_f32) + ay_yolo_decode produce.  d: ksize 1, stride 1, leaky 0,
 * cout = A*(5+C), cout_pad = cout rounded up to 32; scale = ones, shift = the bias (ay_fold_bn without BatchNorm). */
int ay_head_decode_fwd_bf16(const ay_conv_desc* d, const void* src_blocked_bf16, const void* w_packed, const float* scale,
                            const float* shift, int num_anchors, int num_classes, int img_dim,
                            const float* anchors_wh /* host, A*2, pixels */, float* pred, int n_total, int row_offset,
                            ay_stream_t stream);
int ay_head_decode_fwd_f16(const ay_conv_desc* d, const void* src_blocked_f16, const void* w_packed, const float* scale,
                           const float* shift, int num_anchors, int num_classes, int img_dim, const float* anchors_wh, float* pred,
                           int n_total, int row_offset, ay_stream_t stream);

/* ---- box math: utils/utils.py:53-59,193-232 ---------------------------------------------------- */
int ay_xywh2xyxy(float* boxes, int64_t n_rows, int row_stride, ay_stream_t stream); /* in place, first 4 cols */
/* mode 0: IoU with the reference's +1-pixel rule (bbox_iou); mode 1: GIoU (no +1; new, unpinned).
 * n1 == n2 (elementwise) or n1 == 1 (broadcast); xyxy=0 means (cx,cy,w,h) inputs. out[n2]. */
int ay_box_iou(const float* box1, int n1, const float* box2, int n2, int xyxy, int mode, float* out,
               ay_stream_t stream);
/* all pairs: out[n1][n2] */
int ay_box_iou_pairwise(const float* box1, int n1, const float* box2, int n2, int mode, float* out,
                        ay_stream_t stream);

/* ---- merge-NMS: utils/utils.py:235-273 ---------------------------------------------------------- */
/* pred [B][N][5+C] (cx,cy,w,h,conf,cls..) is converted to corners IN PLACE (reference :244).
 * Per image: rows with conf >= conf_thres, ordered by conf*max(cls) descending (ties: lower row first),
 * greedy class-aware suppression (IoU > nms_thres, strict) with confidence-weighted box merge.
 * out_rows [B][max_det][7] (x1,y1,x2,y2,conf,cls_conf,cls_pred), keep_idx [B][max_det] original row of
 * each emitted cluster head, count[B] (clamped to max_det), cand_count[B] candidates after the filter. */
size_t ay_nms_workspace_bytes(int batch, int n_rows);
/* step 1: corners in place + conf filter + sort keys -> cand_count[B] (lets the caller size max_det exactly) */
int ay_nms_filter(float* pred, int batch, int n_rows, int num_classes, float conf_thres, int32_t* cand_count,
                  void* workspace, size_t workspace_bytes, ay_stream_t stream);
/* step 2: per-image sort + greedy merge scan; pred must already hold corners (step 1) */
int ay_nms_sort_merge(const float* pred, int batch, int n_rows, int num_classes, float nms_thres, int max_det,
                      float* out_rows, int32_t* keep_idx, int32_t* count, const int32_t* cand_count,
                      void* workspace, size_t workspace_bytes, ay_stream_t stream);
/* both steps back to back, no host sync; count[b] > max_det tells the caller rows were dropped */
int ay_nms_merge(float* pred, int batch, int n_rows, int num_classes, float conf_thres, float nms_thres,
                 int max_det, float* out_rows, int32_t* keep_idx, int32_t* count, int32_t* cand_count,
                 void* workspace, size_t workspace_bytes, ay_stream_t stream);

/* ---- training step, fp32 reference-precision path (nchw f32) -------------------------------------- */

/* Train-mode BatchNorm2d + LeakyReLU forward (models.py:43-45): batch statistics over (B,H,W), biased variance for the
 * normalisation, running stats updated with PyTorch semantics running = (1-momentum)*running + momentum*batch
 * (unbiased variance), momentum = 0.9 in the reference.  z = raw conv output, y = block output. */
int ay_bn_train_fwd_f32(const float* z, const float* gamma, const float* beta, float* running_mean, float* running_var,
                        float momentum, float eps, int leaky, float* y, float* save_mean, float* save_invstd,
                        int batch, int channels, int hw, ay_stream_t stream);
/* its backward (autograd of the above): dy -> dz, dgamma, dbeta (written, not accumulated) */
int ay_bn_train_bwd_f32(const float* dy, const float* y, const float* z, const float* gamma, const float* save_mean,
                        const float* save_invstd, int leaky, float* dz, float* dgamma, float* dbeta, int batch,
                        int channels, int hw, ay_stream_t stream);
int ay_bias_grad_f32(const float* dz, float* dbias, int batch, int channels, int hw, ay_stream_t stream);
int ay_bias_grad_f32_acc(const float* dz, float* dbias, int accumulate, int batch, int channels, int hw, ay_stream_t stream);
/* autograd of nn.Conv2d (models.py:33-40): input gradient (optionally accumulated into dx) and weight gradient */
int ay_conv_dgrad_f32(const ay_conv_desc* d, const float* dz, const float* w_oihw, float* dx, int accumulate,
                      ay_stream_t stream);
int ay_conv_wgrad_f32(const ay_conv_desc* d, const float* x, const float* dz, float* dw, ay_stream_t stream);
/* shortcut add (models.py:246-248), gradient accumulation, route/upsample copy (models.py:86-96,244-245) and its backward */
int ay_add_f32(const float* a, const float* b, float* out, size_t n, ay_stream_t stream);
int ay_accumulate_f32(float* dst, const float* src, size_t n, ay_stream_t stream);
int ay_copy_channels_f32(const float* src, float* out, int batch, int csrc, int ctotal, int c0, int h, int w, int up,
                         ay_stream_t stream);
int ay_slice_accumulate_f32(const float* dout, float* dsrc, int batch, int csrc, int ctotal, int c0, int h, int w, int up,
                            ay_stream_t stream);
/* YOLO layer loss (models.py:174-222) with build_targets (utils/utils.py:276-330) fused: target assignment
 * (best-of-A anchor by wh-IoU, ignore threshold, last-writer-wins scatter in target order), the six loss terms and the
 * gradient w.r.t. the raw head tensor [B][A*(5+C)][G][G].  sums_out (device, 16 floats): [0..3] sum sq err x,y,w,h @obj,
 * [4] BCE conf @obj, [5] BCE conf @noobj, [6] BCE cls @obj, [7] n_obj, [8] n_noobj, [9] class hits, [10] sum conf @obj,
 * [11] sum conf @noobj, [12] #conf>0.5, [13] #(iou>0.5 & detected), [14] #(iou>0.75 & detected).
 * loss = (s0+s1+s2+s3)/n_obj + s4/n_obj + 100*s5/n_noobj + s6/(n_obj*C); dhead = grad_scale * dloss/dhead. */
size_t ay_yolo_loss_workspace_bytes(int batch, int num_anchors, int num_classes, int grid);
int ay_yolo_loss_fwd_bwd(const float* head_nchw, const float* targets, int n_targets, int batch, int num_anchors,
                         int num_classes, int grid, int img_dim, const float* anchors_wh /* host */, float ignore_thres,
                         float grad_scale, float* dhead, float* sums_out, void* workspace, size_t workspace_bytes,
                         ay_stream_t stream);
/* GIoU variant of the box term (BASELINE.json configs[4]; new feature, no reference counterpart): same call, same sums
 * layout, but sums[0] = sum over object cells of 1 - GIoU(decoded box, target box) (grid units, no +1 rule) and
 * sums[1..3] = 0; loss = s0/n_obj + s4/n_obj + 100*s5/n_noobj + s6/(n_obj*C). */
int ay_yolo_loss_giou_fwd_bwd(const float* head_nchw, const float* targets, int n_targets, int batch, int num_anchors,
                              int num_classes, int grid, int img_dim, const float* anchors_wh /* host */, float ignore_thres,
                              float grad_scale, float* dhead, float* sums_out, void* workspace, size_t workspace_bytes,
                              ay_stream_t stream);
/* utils/utils.py:276-330 build_targets on device, as the dense 10-tuple the reference returns (same order):
 * pred_boxes [B,A,G,G,4] cxcywh in grid units, pred_cls [B,A,G,G,C], targets [nT,6] (sample, class, cx, cy, w, h in [0,1]),
 * anchors_grid (HOST) [A,2] = anchors / stride (models.py:123).  Masks are bytes (0/1).  Duplicate (sample, anchor, cell)
 * targets: the last one in target order wins, classes accumulate (multi-hot), as on the reference's CPU path. */
size_t ay_build_targets_workspace_bytes(int batch, int num_anchors, int grid);
int ay_build_targets(const float* pred_boxes, const float* pred_cls, const float* targets, int n_targets, int batch,
                     int num_anchors, int num_classes, int grid, const float* anchors_grid /* host */, float ignore_thres,
                     float* iou_scores, float* class_mask, uint8_t* obj_mask, uint8_t* noobj_mask, float* tx, float* ty,
                     float* tw, float* th, float* tcls, float* tconf, void* workspace, size_t workspace_bytes,
                     ay_stream_t stream);
/* torch.optim.Adam step (train.py:81,118) on one flat buffer; grads are multiplied by grad_scale first (1/world size) */
int ay_adam_flat(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1,
                 float beta2, float eps, int step, float grad_scale, ay_stream_t stream);

/* ---- evaluation statistics ------------------------------------------------------- */
/* Greedy true-positive matching of get_batch_statistics (utils/utils.py:154-190) for a batch: rows [B][max_det][7]
 * (x1,y1,x2,y2,conf,cls_conf,cls_pred, detections in descending-score order as non_max_suppression emits them), count[B],
 * targets [nT][6] = (sample, class, x1, y1, x2, y2) in pixels -> tp [B][max_det] (1.0 = true positive).  *overflow is set
 * to 1 if an image has more than 2048 targets (only the first 2048 are matched). */
int ay_match_detections(const float* rows, const int32_t* count, int batch, int max_det, const float* targets, int n_targets,
                        float iou_thres, float* tp, int32_t* overflow, ay_stream_t stream);

/* ---- tile ingest ----------------------------------------------------------------- */
/* uint8 HWC RGB tiles [B,H,W,3] -> float32 NCHW [B,3,S,S]: x/255 (utils/transforms.py:96), centre zero pad to square
 * (utils/datasets.py:22-32), nearest resize src = min(floor(dst * (float)in/out), in-1) (utils/datasets.py:35-37), one pass. */
int ay_ingest_tiles_u8(const void* img_hwc_u8, int batch, int h, int w, int out_size, float pad_value, float* out_nchw,
                       ay_stream_t stream);

/* WSI -> tile streaming (crop.py:13-25,44-47): tile t = (ty, tx) of the tile_size grid dzsave(layout='google')
 * lays over the slide, cut out of a resident uint8 HWC region (rows `row_stride_bytes` apart; region_h x region_w source pixels),
 * edge tiles padded with the background 255; shrink 2 = the 40x -> 20x halving first (2x2 mean, round half up; the grid then lies
 * on the region_h/2 x region_w/2 image); then x/255 and the nearest resize to out_size as in ay_ingest_tiles_u8.
 * out [tiles_y*tiles_x][3][out_size][out_size] fp32. */
int ay_ingest_region_tiles_u8(const void* region_hwc_u8, int region_h, int region_w, size_t row_stride_bytes, int shrink,
                              int tile, int tiles_y, int tiles_x, int out_size, float* out_nchw, ay_stream_t stream);

/* ---- training step, bf16 MFMA path (blocked bf16 activations and activation gradients) ------------- */

/* Train-mode BatchNorm + LeakyReLU (+ fused shortcut add of `skip`) around the MFMA convolution: statistics pass (fp64
 * atomics into sums_ws[2*C]), finalize (mean/invstd/running stats, PyTorch momentum semantics), apply pass -> y. */
int ay_bn_train_fwd_bf16(const void* z, const float* gamma, const float* beta, float* running_mean, float* running_var,
                         float momentum, float eps, int leaky, const void* skip, void* y, float* save_mean, float* save_invstd,
                         double* sums_ws, int batch, int channels, int h, int w, ay_stream_t stream);
/* dy (gradient of the block output, before the shortcut add) -> dz, dgamma, dbeta; leaky' is taken from the recomputed
 * pre-activation gamma*xhat+beta */
int ay_bn_train_bwd_bf16(const void* dy, const void* z, const float* gamma, const float* beta, const float* save_mean,
                         const float* save_invstd, int leaky, void* dz, float* dgamma, float* dbeta, double* sums_ws, int batch,
                         int channels, int h, int w, ay_stream_t stream);
/* The `_acc` forms with accumulate != 0 ADD the parameter gradients to what dgamma / dbeta / dw / dbias hold: the parameter
 * gradients of a step go straight into the caller's (flat) gradient buffer, also across the batches of a gradient
 * accumulation (train.py:116-119), instead of through per-layer temporaries and autograd's accumulation pass. */
int ay_bn_train_bwd_bf16_acc(const void* dy, const void* z, const float* gamma, const float* beta, const float* save_mean,
                             const float* save_invstd, int leaky, void* dz, float* dgamma, float* dbeta, double* sums_ws,
                             int accumulate, int batch, int channels, int h, int w, ay_stream_t stream);
/* The two BatchNorm calls for a caller that has CLEARED sums_ws itself (fp64 zeros on entry): a training step of Darknet-53
 * zeroes the workspaces of all 72 layers and both passes with one memset instead of 144 small fill launches.  Otherwise
 * identical to ay_bn_train_fwd_bf16 / ay_bn_train_bwd_bf16_acc (models.py:43 train-mode semantics and its backward). */
int ay_bn_train_fwd_bf16_zeroed_ws(const void* z, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                   float momentum, float eps, int leaky, const void* skip, void* y, float* save_mean, float* save_invstd,
                                   double* sums_ws_zeroed, int batch, int channels, int h, int w, ay_stream_t stream);
int ay_bn_train_bwd_bf16_acc_zeroed_ws(const void* dy, const void* z, const float* gamma, const float* beta, const float* save_mean,
                                       const float* save_invstd, int leaky, void* dz, float* dgamma, float* dbeta, double* sums_ws_zeroed,
                                       int accumulate, int batch, int channels, int h, int w, ay_stream_t stream);
int ay_accumulate_bf16(void* dst, const void* src, size_t n_elems, ay_stream_t stream);
/* route / nearest-upsample backward on blocked tensors (channel counts multiples of 16) */
int ay_slice_accumulate_bf16(const void* dout, void* dsrc, int batch, int csrc, int ctotal, int c0, int h, int w, int up,
                             int accumulate, ay_stream_t stream);
/* stride-2 data gradient = stride-1 convolution of the zero-inserted output gradient: out[2y][2x] = in[y][x] */
int ay_zero_insert_bf16(const void* in, void* out, int batch, int channels, int h, int w, int ho, int wo, ay_stream_t stream);
/* filters that make ay_conv_fwd_bf16 compute the data gradient: W'[ci][co][kh][kw] = W[co][ci][k-1-kh][k-1-kw], packed
 * [ceil(cout/16)][k*k][2][cin_pad][8]; use with desc{cin=ceil16(cout), cout=cin, cout_pad=cin_pad, stride 1}. */
int ay_pack_dgrad_weights_bf16(const float* w_oihw, void* packed, int cout, int cin, int cin_pad, int ksize, ay_stream_t stream);
/* Every filter image a training step needs, re-packed in ONE launch after the optimiser moved the weights (the reference has no
 * counterpart: its convolutions read nn.Conv2d.weight directly; this replaces ~145 ay_pack_*_bf16 launches per step).  `jobs` and
 * `work` live in DEVICE memory and are built once per weight layout: job j packs `total` elements of image kind `kind` (0 =
 * ay_pack_conv_weights_bf16, `cin` = channels of the source tensor; 1 = ay_pack_dgrad_weights_bf16; 2 =
 * ay_pack_dgrad_s2_weights_bf16) exactly as the single calls do; work item w = (job, first_block) covers elements
 * [first_block * ay_pack_batch_block(), +ay_pack_batch_block()) of that job. */
typedef struct ay_pack_job {
    const float* src;
    void* dst;
    int32_t kind, cout, cout_pad, cin, cin_pad, ksize;
    uint64_t total;
} ay_pack_job;
typedef struct ay_pack_work {
    uint32_t job, first_block;
} ay_pack_work;
int ay_pack_batch_block(void);
int ay_pack_batch_bf16(const void* jobs_device, const void* work_device, int n_work, ay_stream_t stream);
/* The stem Conv2d(3, 32, 3, 1, 1) (models.py:33-41, layer 0) on the bf16 training path, straight from the fp32 NCHW image (W % 4 == 0,
 * 16-byte aligned): forward z = bf16(conv(bf16(x), bf16(w))) with fp32 accumulation -> blocked bf16 [B][2][H][W][16]
 * (w_bf16: [32][32] bf16, index ci*9 + kh*3 + kw, entries 27..31 unused), and the filter gradient dW[32][3][3][3] (fp32,
 * overwritten or, accumulate != 0, added to) from the blocked bf16 output gradient; partial sums per workgroup go to
 * `workspace` (ay_stem_train_wgrad_workspace_bytes()) and are added in a fixed order.  What loss.backward() (train.py:113)
 * computes for that layer, on bf16-rounded operands. */
int ay_stem_train_fwd_bf16(const float* x_nchw, const void* w_bf16, void* z_blocked, int batch, int h, int w, ay_stream_t stream);
/* ... and the layer's BatchNorm batch statistics gathered where z is produced ("stats reduce fused with the conv
 * epilogue"): sums[0..31] = sum z, sums[32..63] = sum z^2 over the batch (of the bf16-rounded values, fp64, fixed summation order), to
 * be followed by ay_bn_train_apply_bf16 -- ay_bn_train_fwd_bf16 without its statistics pass. */
size_t ay_stem_train_stats_workspace_bytes(void);
int ay_stem_train_fwd_stats_bf16(const float* x_nchw, const void* w_bf16, void* z_blocked, double* sums, void* workspace, size_t workspace_bytes,
                                 int batch, int h, int w, ay_stream_t stream);
int ay_bn_train_apply_bf16(const void* z, const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                           float eps, int leaky, const void* skip, void* y, float* save_mean, float* save_invstd, const double* sums_ws,
                           int batch, int channels, int h, int w, ay_stream_t stream);
size_t ay_stem_train_wgrad_workspace_bytes(void);
int ay_stem_train_wgrad_bf16(const float* x_nchw, const void* dz_blocked, float* dw_oihw, int accumulate, void* workspace,
                             size_t workspace_bytes, int batch, int h, int w, ay_stream_t stream);
/* Data gradient of a 3x3 stride-2 convolution (the reference gets it from autograd: loss.backward(), train.py:113, through
 * nn.Conv2d(stride=2), models.py:33-41) WITHOUT zero insertion: output pixel (y, x) only receives the filter taps with
 * (y + 1 - kh) and (x + 1 - kw) even, so each parity class (y & 1, x & 1) of dx is a stride-1 convolution of dz with a
 * 2x2 window of 1, 2, 2 or 4 live taps -- 16 tap slots per dz pixel instead of the 36 of the zero-inserted form, and dz is
 * read at its own (quarter) size.  `d` is the FORWARD convolution's descriptor (ksize 3, stride 2, even hin/win; cout_pad =
 * channels of dz's planes); dx has cin_pad (multiple of 32) channel planes of hin x win; residual (may alias dx) is
 * added before the bf16 rounding; ones / zeros: cin_pad floats of 1 / 0 (the kernel's affine epilogue).
 * Filters from ay_pack_dgrad_s2_weights_bf16: [class py*2+px][cout_pad/16][window tap][2][cin_pad][8] bf16. */
size_t ay_packed_dgrad_s2_weight_bytes(int cout_pad, int cin_pad);
int ay_pack_dgrad_s2_weights_bf16(const float* w_oihw, void* packed, int cout, int cout_pad, int cin, int cin_pad, ay_stream_t stream);
int ay_conv_dgrad_s2_bf16(const ay_conv_desc* d, const void* dz, const void* w_s2_packed, const float* ones, const float* zeros,
                          const void* residual, void* dx, int cin_pad, ay_stream_t stream);
/* weight gradient on the MFMA path: dW (OIHW fp32, overwritten) from blocked bf16 input x and output gradient dz
 * (desc as in the forward; cout_pad = channels of dz's planes) */
int ay_conv_wgrad_bf16(const ay_conv_desc* d, const void* x_blocked, const void* dz_blocked, float* dw_oihw, ay_stream_t stream);
int ay_conv_wgrad_bf16_acc(const ay_conv_desc* d, const void* x_blocked, const void* dz_blocked, float* dw_oihw, int accumulate,
                           ay_stream_t stream);
/* The same with a caller-owned workspace of ay_conv_wgrad_workspace_bytes(d) bytes: the split-K partial filters are written to
 * slabs and summed in a fixed order by a second small kernel -- bit-identical results from run to run, and faster than the
 * fp32 atomics of the forms above (which remain for callers without a workspace; a too small workspace falls back to them). */
size_t ay_conv_wgrad_workspace_bytes(const ay_conv_desc* d);
int ay_conv_wgrad_bf16_ws(const ay_conv_desc* d, const void* x_blocked, const void* dz_blocked, float* dw_oihw, int accumulate,
                          void* workspace, size_t workspace_bytes, ay_stream_t stream);

/* ---- inference plan: Darknet.forward (models.py:237-255) lowered to a flat op list ------------------------------
 * The host lowers the cfg graph once (which layers fuse, which routes fold into a loader) and hands the ops over; the
 * library lays the layer outputs ("values") out in ONE caller-allocated workspace -- a value's bytes are reused once its
 * last reader has been issued (first-fit over lifetimes; everything is stream-ordered) -- and ay_plan_forward issues the
 * whole network: one C call per batch instead of ~85.  Weight/scale/shift pointers are device memory owned by the caller
 * and must outlive the plan.  src/res/dst are value ids (0 .. n_values-1); AY_PLAN_INPUT names the network input. */
#define AY_PLAN_INPUT (-1)
#define AY_PLAN_NONE (-2)
enum {
    AY_OP_STEM_S2_FUSED = 1, /* ay_stem_s2_fused_fwd: input -> dst; w = stem filters bf16, w2 = packed layer-1 filters */
    AY_OP_STEM = 2,          /* ay_stem_conv_fwd: input -> dst; w = OIHW fp32 filters */
    AY_OP_CONV = 3,          /* ay_conv_fwd_bf16: src (+ res) -> dst */
    AY_OP_RESBLOCK = 4,      /* ay_resblock_fwd_bf16: src -> dst; (w, scale, shift, conv.leaky) = 1x1, (w2, ..2, leaky2) = 3x3 */
    AY_OP_CONV1X1_CAT = 5,   /* ay_conv1x1_cat_fwd_bf16: src = half-resolution source (c1 channels), src2 = direct source */
    AY_OP_CONCAT_UPSAMPLE = 6, /* ay_concat_upsample_bf16: src (c1 channels, up1) [+ src2 (c2 channels)] -> dst, conv.hout x wout */
    AY_OP_DECODE = 7         /* ay_yolo_decode of a blocked-f32 head: src -> rows [row_offset, ..) of the output */
};
typedef struct ay_plan_op {
    int32_t kind;
    int32_t src, src2, res, dst;   /* value ids; AY_PLAN_NONE where unused */
    ay_conv_desc conv;             /* shapes (all kinds use batch/hout/wout; RESBLOCK: cin = channels) */
    int32_t c1, c2, up1;           /* CONV1X1_CAT: c1; CONCAT_UPSAMPLE: c1, c2, up1 */
    int32_t leaky2;                /* STEM_S2_FUSED / RESBLOCK: activation of the second convolution */
    int32_t num_anchors, num_classes, grid, row_offset; /* DECODE */
    float anchors_wh[12];          /* DECODE: up to 6 anchors (w,h) in pixels */
    const void* w;
    const float* scale;
    const float* shift;
    const void* w2;
    const float* scale2;
    const float* shift2;
} ay_plan_op;
typedef struct ay_plan ay_plan;
/* value_bytes[i] = size of value i.  Fails (AY_ERR_ARG) on a value read before it is written or never written.
 * act_dtype: AY_DT_BF16 | AY_DT_F16 -- the storage type of the blocked activations and of every packed filter image in `ops`
 * (the plan then issues the _bf16 or the _f16 entry points). */
int ay_plan_create(const ay_plan_op* ops, int n_ops, const size_t* value_bytes, int n_values, int img_dim, int n_total_rows,
                   int act_dtype, ay_plan** out_plan);
void ay_plan_destroy(ay_plan* plan);
size_t ay_plan_workspace_bytes(const ay_plan* plan);       /* 256-byte aligned arena the caller allocates */
size_t ay_plan_value_offset(const ay_plan* plan, int value); /* where a value lives in the arena (tests) */
/* x: [B][3][S][S] fp32 NCHW; out_rows: [B][n_total_rows][5+C] fp32.  Stream-ordered, no host synchronisation. */
int ay_plan_forward(const ay_plan* plan, const float* x_nchw, void* workspace, float* out_rows, ay_stream_t stream);
/* Profiling without a host synchronisation inside the measured region: between begin and end every ay_plan_forward records
 * an event pair around each selected op (op_selected[n_ops] of 0/1, NULL = all; an event record costs the stream a few
 * microseconds), on the stream it issues to; end waits for them and returns, per op, the time summed over the recorded
 * forwards (0 for unselected ops).  A plan is driven by one host thread at a time. */
int ay_plan_profile_begin(ay_plan* plan, const unsigned char* op_selected);
/* the same with event pairs on every `every`-th forward only (the first one included): 62 event records cost a 24-ms step 0.17 ms;
 * ay_plan_profile_end then returns the number of forwards that were RECORDED */
int ay_plan_profile_begin_every(ay_plan* plan, const unsigned char* op_selected, int every);
int ay_plan_profile_end(ay_plan* plan, float* op_ms_sum /* n_ops */, int* n_forwards);
/* one forward with a HIP event pair around every op on `stream`; synchronises the stream and fills op_ms[n_ops] */
int ay_plan_forward_timed(const ay_plan* plan, const float* x_nchw, void* workspace, float* out_rows, float* op_ms,
                          ay_stream_t stream);

/* ---- union-merge of overlapping same-class detections (core.py:366-423 mergeDetections, :326-364) ----------
 * rows [batch][max_rows][7] = (x1, y1, x2, y2, conf, cls_conf, cls_pred) as ay_nms_merge leaves them (after rescaling), count[batch]
 * valid rows per image -> rows_out [batch][max_rows][7], count_out[batch]: pairs of rows of class 0 or 1 whose truncated integer pixel
 * rectangles share a pixel are replaced by the rectangle of the covered pixels (min of the confidences), pass after pass until
 * nothing changes, exactly as the reference does -- with the pair order the reference leaves to a Python set made explicit: rows
 * in input order, merged rows appended.  One wavefront per image, rows in LDS; max_rows <= ay_merge_detections_max_rows(). */
int ay_merge_detections_max_rows(void);
int ay_merge_detections(const float* rows, const int* count, int batch, int max_rows, float* rows_out, int* count_out,
                        ay_stream_t stream);

/* Replaying a captured HIP graph of these calls.  Every entry point is plain stream work -- kernel launches only: no allocation,
 * no host copy, no memset node, no symbol access inside a call -- so a stream capture of a step (ay_plan_forward + ay_nms_merge ...)
 * replays like any other graph (scripts/micro/graph_sync.hip, graph_coherence.hip, graph_input_coherence.hip: every wait covers a
 * replayed graph, a kernel behind a replay sees its writes, a replay sees eager writes to its inputs).  The persistent kernels rely on
 * stream order between launches (a launch hands its work-counter set back zeroed for a later launch on that stream, the plan's
 * arena reuses a block once its last reader has been issued): replay a graph on ONE stream at a time and do not run other library
 * work on the capture stream concurrently.  ay_stream_fence records a library-owned event on `stream` and makes the stream wait for
 * it: an optional stream-ordered fence (utils.graph_replay() places it behind a replay; the product test replays without it). */
int ay_stream_fence(ay_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* AMYLOID_YOLO_H */
