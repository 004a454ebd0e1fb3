/* libmnyolo — C ABI of the MI355X-native MobileNet-YOLO hot path (gfx950 only).
 *
 * Drop-in boundary B2 of SURVEY §8(b).  The reference (eric612/Mobilenet-YOLO-Pytorch)
 * owns no kernels: its hot path is stock torch.nn ops + torchvision.ops.nms.  Each entry
 * point below therefore cites the reference CALL SITE whose arithmetic it replaces
 * (paths relative to the reference repo root).
 *
 * Conventions
 *  - Activations are channels-last (NHWC) fp32 unless stated otherwise.
 *  - "View" arguments (x, in_scale, in_shift, in_act): the kernel reads
 *        a = act(x * in_scale[c] + in_shift[c])      (in_scale == NULL -> a = act(x))
 *    i.e. the BatchNorm-apply + activation of the PRODUCING layer is fused into the
 *    consumer's load path; padding taps are 0 in the activated domain.
 *  - `stats` (optional, may be NULL): per-channel partial sums of the raw output,
 *    laid out [parts][2][C] (sum, sum of squares); `parts` is returned by the matching
 *    *_stat_parts() query.  mny_bn_finalize() reduces them in fp64.
 *  - Ownership: the caller allocates every buffer (inputs, outputs, workspaces).  The
 *    library never allocates, frees or retains device pointers.
 *  - All work is enqueued on `stream` (a hipStream_t passed as void*); no entry point
 *    synchronises the device.
 *  - Return value: 0 on success, a negative MNY_E* code otherwise; mny_last_error()
 *    returns a thread-local message.  No C++ exception crosses this boundary.
 *  - Re-entrant; deterministic (no floating-point atomics anywhere).
 */
#ifndef MNYOLO_H
#define MNYOLO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MNY_OK 0
#define MNY_EINVAL (-1)   /* bad argument (shape, alignment, null pointer) */
#define MNY_EHIP (-2)     /* a HIP runtime call / kernel launch failed */
#define MNY_EUNSUPPORTED (-3)

/* activation codes (mobilenetv2.py:42 ReLU6, mbv2_yolo.py:24 LeakyReLU(0.1),
 * mobilenetv3.py:14-23 hswish / ReLU) */
#define MNY_ACT_NONE 0
#define MNY_ACT_RELU6 1
#define MNY_ACT_LEAKY 2
#define MNY_ACT_RELU 3
#define MNY_ACT_HSWISH 4
#define MNY_ACT_HSIGMOID 5   /* relu6(z+3)/6, mobilenetv3.py:20-23; only as the gate operand of mny_mul_views */

int mny_version(void);
const char* mny_last_error(void);
/* max number of stat/partial rows any kernel will write (upper bound for workspace sizing) */
int mny_max_parts(void);

/* ---- stem: 3x3 stride-2 pad-1 conv 3->Cout on NCHW input, NHWC output -------------------
 * replaces nn.Conv2d(3,32,3,2,1) at models/mobilenetv2.py:40 (used :113).               */
int mny_stem_fwd(const float* x_nchw, const float* w /*[Cout,3,3,3]*/, float* y /*[N,H/2,W/2,Cout]*/,
                 float* stats, int N, int H, int W, int Cout, void* stream);
int mny_stem_stat_parts(int N, int H, int W, int Cout);
/* dW = sum x * dy; `ws` holds [mny_stem_wgrad_parts()][Cout*27] floats */
int mny_stem_wgrad(const float* x_nchw, const float* dy, float* dw, float* ws,
                   int N, int H, int W, int Cout, void* stream);
int mny_stem_wgrad_parts(int N, int H, int W, int Cout);

/* weight gradient of the stem conv straight from the stem unit's OUTPUT gradient g (what autograd hands to
 * BatchNorm's backward, mobilenetv2.py:40-42): dY = ca*g*act'(scale*y+shift) + cb*y + cc is rebuilt on load from
 * g and the raw conv output y (coef = mny_bn_bwd_finalize's [3][Cout]), so mny_bn_bwd_apply's dY tensor is
 * neither written nor re-read.  ws as for mny_stem_wgrad.  Cout with 256 % (Cout/4) == 0. */
int mny_stem_bnwgrad_supported(int Cout);
int mny_stem_bnwgrad(const float* x_nchw, const float* g, const float* y, const float* scale,
                     const float* shift, int act, const float* coef, float* dw, float* ws, int N, int H, int W,
                     int Cout, void* stream);

/* ---- depthwise KxK (K=3|5), stride 1|2, pad K/2, no bias ---------------------------------
 * replaces nn.Conv2d(C,C,3,s,1,groups=C) at models/mobilenetv2.py:65,79 and
 * models/mbv2_yolo.py:22 (BasicConv depthwise); K=5 for models/mobilenetv3.py:54.        */
int mny_dw_fwd(const float* x, const float* in_scale, const float* in_shift, int in_act,
               const float* w /*[C,K,K]*/, float* y, float* stats,
               int N, int H, int W, int C, int K, int stride, void* stream);
int mny_dw_stat_parts(int N, int H, int W, int C, int K, int stride);
/* the same count for a given storage type: flags bit 0 = bf16 storage.  On bf16 storage the 5x5 stride-1 launches with >= 120 channels (last
 * 64-channel chunk at least 3/4 full) run the TILE form of csrc/dwtile.hip (every thread loads and activates its own column once, the
 * window comes back from an LDS ring; MNY_DWTF=0 / 1 forces never / always), which has its own grid. */
int mny_dw_stat_parts_x(int N, int H, int W, int C, int K, int stride, int flags);
/* dx[N,H,W,C] = (addend ? addend : 0) + conv_transpose(dy, w); H,W are the INPUT extents */
int mny_dw_bwd_data(const float* dy, const float* w, const float* addend, float* dx,
                    int N, int H, int W, int C, int K, int stride, void* stream);
/* dw[C,K,K]; ws holds [mny_dw_wgrad_parts()][C*K*K] floats */
int mny_dw_bwd_weight(const float* x, const float* in_scale, const float* in_shift, int in_act,
                      const float* dy, float* dw, float* ws,
                      int N, int H, int W, int C, int K, int stride, void* stream);
int mny_dw_wgrad_parts(int N, int H, int W, int C, int K, int stride);

/* ---- pointwise 1x1 conv == row-major GEMM on fp32 MFMA ----------------------------------
 * y[M,Nc] = act_in(x)[M,K] * w[Nc,K]^T (+ bias) (+ addend)
 * replaces nn.Conv2d(Cin,Cout,1) at models/mobilenetv2.py:48,69,75,83,
 * models/mbv2_yolo.py:20 (BasicConv 1x1) and :82 (biased head conv).
 * The same entry point computes the data gradient: dx[M,K] = dy[M,Nc] * wT[K,Nc]^T with
 * wT = mny_transpose(w) and `addend` = the other gradient contribution (residual path). */
int mny_pw_fwd(const float* x, const float* in_scale, const float* in_shift, int in_act,
               const float* w, const float* bias, const float* addend, float* y, float* stats,
               int64_t M, int K, int Nc, void* stream);
int mny_pw_stat_parts(int64_t M, int K, int Nc);
/* dw[Nc,K] = dy[M,Nc]^T * act_in(x)[M,K]; dbias[Nc] (optional) = column sums of dy.
 * ws holds mny_pw_wgrad_ws_floats() floats. */
int mny_pw_wgrad(const float* x, const float* in_scale, const float* in_shift, int in_act,
                 const float* dy, float* dw, float* dbias, float* ws,
                 int64_t M, int K, int Nc, void* stream);
size_t mny_pw_wgrad_ws_floats(int64_t M, int K, int Nc);
/* Fused BatchNorm-backward + weight gradient + data gradient of a pointwise unit whose input has <= 32 channels
 * (the HBM-bound "expand" layers of models/mobilenetv2.py:75-77, e.g. 16->96 @176x176): replaces
 * mny_bn_bwd_reduce/finalize/apply + mny_pw_wgrad + mny_transpose + mny_pw_fwd(dgrad) and never materialises dY
 * (DESIGN.md §4).  g = dL/d act(scale*y+shift); x is read through its view; dx (optional) = data gradient wrt the
 * activated input (+ addend).  ws: mny_pw_bnbwd_ws_floats() floats.  mny_pw_bnbwd_supported() tells whether a shape qualifies. */
int mny_pw_bnbwd_supported(int64_t M, int K, int Nc);
size_t mny_pw_bnbwd_ws_floats(int64_t M, int K, int Nc);
int mny_pw_bnbwd(const float* g, const float* y, const float* scale, const float* shift, int act,
                 const float* mean, const float* invstd, const float* gamma,
                 const float* x, const float* in_scale, const float* in_shift, int in_act,
                 const float* w, const float* addend, float* dx, float* dw, float* dgamma, float* dbeta,
                 float* ws, int64_t M, int K, int Nc, void* stream);
/* mny_pw_bnbwd whose data gradient completes the output gradient of the conv+BN+act unit in front of the expand unit (the project conv of the previous
 * block, directly or through the residual add that hands it the gradient unchanged): that unit's BN-backward sums (sum dz, sum dz*xhat over the STORED dx;
 * ry = its raw output [M][K], r_scale / r_shift / r_act = its view, clamp family) leave with the second stage, red[mny_pw_bnbwd_red_parts()][2][K] —
 * no mny_bn_bwd_reduce pass over (dx, ry).  fp32 storage, N in {96, 144, 192} (mny_pw_bnbwd_red_supported). */
int mny_pw_bnbwd_red_supported(int64_t M, int K, int Nc);
int mny_pw_bnbwd_red_parts(int64_t M, int K, int Nc);
int mny_pw_bnbwd_red(const float* g, const float* y, const float* scale, const float* shift, int act, const float* mean, const float* invstd,
                     const float* gamma, const float* x, const float* in_scale, const float* in_shift, int in_act, const float* w,
                     const float* addend, float* dx, float* dw, float* dgamma, float* dbeta, float* ws, const float* ry, const float* r_scale,
                     const float* r_shift, int r_act, const float* r_mean, const float* r_invstd, float* red, int64_t M, int K, int Nc, void* stream);
int mny_transpose(const float* src /*[R,Cc]*/, float* dst /*[Cc,R]*/, int R, int Cc, void* stream);

/* ---- BatchNorm (training / eval), eps 1e-5, momentum 0.1 ---------------------------------
 * replaces nn.BatchNorm2d at models/mobilenetv2.py:41,49,66,70,76,80,84 and
 * models/mbv2_yolo.py:23.  Training: stats partials -> batch mean/var (biased for the
 * normalisation, unbiased for running_var like torch) -> scale/shift consumed by views. */
int mny_bn_finalize(const float* stats, int parts, int64_t count,
                    const float* gamma, const float* beta, float eps, float momentum,
                    float* running_mean, float* running_var,   /* updated in place; may be NULL */
                    float* scale, float* shift, float* mean, float* invstd, int C, void* stream);
int mny_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* scale, float* shift,
                       int C, void* stream);
/* backward, step 1: partial sums of dz = g * act'(scale*y+shift) and dz*yhat,
 * yhat = (y-mean)*invstd  -> red[parts][2][C] */
int mny_bn_bwd_reduce(const float* g, const float* y, const float* scale, const float* shift, int act,
                      const float* mean, const float* invstd, float* red, int64_t M, int C, void* stream);
int mny_bn_bwd_parts(int64_t M, int C);
/* step 2: dgamma, dbeta and the per-channel coefficients of dy = ca*dz + cb*y + cc */
int mny_bn_bwd_finalize(const float* red, int parts, int64_t count, const float* gamma,
                        const float* mean, const float* invstd,
                        float* dgamma, float* dbeta, float* coef /*[3][C]*/, int C, void* stream);
/* step 3: dy = ca * g*act'(scale*y+shift) + cb*y + cc   (dy may alias g).
 * coef == NULL: dy = g*act'(...) only (activation backward of a non-BN view). */
int mny_bn_bwd_apply(const float* g, const float* y, const float* scale, const float* shift, int act,
                     const float* coef, float* dy, int64_t M, int C, void* stream);

/* ---- residual / upsample glue -----------------------------------------------------------
 * out = view(a) + view(b) [+ nearest-2x-upsample(up)]   (all [N,H,W,C]; up is [N,H/2,W/2,C])
 * replaces `x + self.conv(x)` (mobilenetv2.py:89), torch.add (mbv2_yolo.py:103,151) and
 * nn.Upsample(scale_factor=2,'nearest') (mbv2_yolo.py:52).  b may be NULL (materialise a). */
int mny_add_views(const float* a, const float* a_scale, const float* a_shift, int a_act,
                  const float* b, const float* b_scale, const float* b_shift, int b_act,
                  const float* up, float* out, int N, int H, int W, int C, void* stream);
/* out = view(a) * view(b): the per-pixel "SE" gate of models/mobilenetv3.py:40-41 (x * hsigmoid(bn(conv(..)))).
 * backward of one operand: dst = (addend ? addend : 0) + g * view(other). */
int mny_mul_views(const float* a, const float* a_scale, const float* a_shift, int a_act,
                  const float* b, const float* b_scale, const float* b_shift, int b_act,
                  float* out, int64_t M, int C, void* stream);
int mny_mul_views_bwd(const float* g, const float* o, const float* o_scale, const float* o_shift, int o_act,
                      const float* addend, float* dst, int64_t M, int C, void* stream);
/* PartAdd(a, upsample2x(up)) of models/mbv3_yolo.py:85-96,135 for Ca <= Cb:
 * out[..., :Ca] = view(a) + up2x[..., :Ca];  out[..., Ca:] = up2x[..., Ca:].   out/up2x have Cb channels. */
int mny_partadd_up(const float* a, const float* a_scale, const float* a_shift, int a_act, const float* up,
                   float* out, int N, int H, int W, int Ca, int Cb, void* stream);
/* dst[M,Ca] = (accumulate ? dst : 0) + src[M,Cb][:, :Ca]   (gradient of the PartAdd's first operand) */
int mny_slice_channels(const float* src, float* dst, int accumulate, int64_t M, int Ca, int Cb, void* stream);
/* dst[N,H/2,W/2,C] = (accumulate ? dst : 0) + sum of the 2x2 children of src[N,H,W,C] */
int mny_upsample_bwd(const float* src, float* dst, int accumulate, int N, int H, int W, int C, void* stream);
/* dst = (accumulate ? dst : 0) + alpha[0] * src   (alpha: device scalar or NULL => 1) */
int mny_axpy(const float* src, const float* alpha, float* dst, int accumulate, int64_t n, void* stream);

/* ---- detection math ---------------------------------------------------------------------
 * One head, channels-last [N,g,g,A*(5+C)] (channel = a*(5+C)+attr, yolo_loss.py:84).
 * anchors_all: [n_anchors_all][2] already divided by img_size (yolo_loss.py:214);
 * mask: the A indices of this head's anchors.
 * targets: packed [T,5] = (label 1..C, cx, cy, w, h); t_off[N+1] = per-image offsets.  */
typedef struct {
    int N, g, A, C;            /* batch, grid, anchors of this head, classes */
    int n_anchors_all;
    float ignore_thresh, iou_thresh, iou_weighting;
} mny_yolo_head;

/* Training: replaces YOLOLoss.forward/get_target (models/yolo_loss.py:77-178,206-236),
 * box_ciou (:257-293), class_loss (:425-434), weighted_mse_loss (:53-60) and their autograd
 * graph.  out7 = (loss, recall, avg_iou, obj, no_obj, cls_score, count/N); dhead = dLoss/dhead.
 * ws: mny_yolo_loss_ws_bytes() bytes. */
int mny_yolo_loss(const float* head, const float* targets, const int32_t* t_off,
                  const float* anchors_all, const int32_t* mask, const mny_yolo_head* hp,
                  float* out7, float* dhead, void* ws, void* stream);
size_t mny_yolo_loss_ws_bytes(const mny_yolo_head* hp, int total_targets);

/* Eval: replaces YOLOLoss.get_pred_boxes (models/yolo_loss.py:180-204): decode every cell to
 * (x1,y1,x2,y2,conf,cls_score,cls_idx), keep conf > val_conf, compact per image preserving
 * (anchor,row,col) order.  Image n writes rows[n*row_stride + base + i]; base = base_counts[n]
 * (NULL -> 0) so a second head can append behind the first (utils/box.py:17 `cat`);
 * counts[n] = base + kept.  row_stride >= A*g*g (+ what earlier heads may have written). */
int mny_yolo_decode(const float* head, const float* anchors_all, const int32_t* mask,
                    const mny_yolo_head* hp, float val_conf, float* rows, int row_stride,
                    const int32_t* base_counts, int32_t* counts, void* stream);

/* Per-class NMS: replaces utils/box.py:11-31 + torchvision.ops.nms(boxes, score*conf, thr).
 * rows: [capacity,7]; segment s (one image) = rows[seg_begin[s] .. seg_begin[s]+seg_count[s]).
 * out_idx: kept row indices (into rows); segment s occupies out_idx[seg_begin[s] ..
 * +out_counts[s]), class-major then descending score (stable, ties by original order).
 * out_rows (optional): the kept rows gathered densely, segment after segment.
 * thr is a double and the float IoU is promoted before the strict `>` compare, like torchvision.
 * max_seg_rows: caller's upper bound on rows in any one segment (sizes the LDS sort buffer;
 * 0 = use `capacity`).  No bucket-size limit (utils/box.py:20-29 has none): a (segment,class)
 * bucket beyond the 8192-row LDS image is sorted and resolved through global scratch.  Only a
 * WRONG bound (a real segment larger than max_seg_rows) is an error: such a bucket keeps nothing
 * and the int32 at ws+mny_nms_status_offset() receives its size (0 = ok).
 * The int32[S+1] exclusive prefix of out_counts is left at ws+mny_nms_prefix_offset().
 * ws: mny_nms_ws_bytes() bytes. */
int mny_nms_per_class(const float* rows, const int32_t* seg_begin, const int32_t* seg_count, int S,
                      int capacity, int max_seg_rows, int num_classes, double thr,
                      int32_t* out_idx, int32_t* out_counts, float* out_rows, void* ws, void* stream);
size_t mny_nms_ws_bytes(int S, int capacity, int num_classes);
size_t mny_nms_status_offset(int S, int capacity, int num_classes);
size_t mny_nms_prefix_offset(int S, int capacity, int num_classes);

/* ---- fused backward of a depthwise 3x3 stride-1 conv + BN + activation unit ----------------------------
 * One pass over (g, y, x) instead of bn_bwd_apply + dw_bwd_weight + dw_bwd_data (7 tensor passes -> 4): rebuilds
 * dY = ca*g*act'(scale*y+shift) + cb*y + cc in registers from the coefficients `coef` = [3][C] written by
 * mny_bn_bwd_finalize, and produces dx (gradient wrt the ACTIVATED input view, + optional addend) and dw [C,1,3,3].
 * `ws`: [mny_dw_bnbwd_parts_k()][C*K*K] floats.  K == 3 or K == 5, stride == 1 (mny_dw_bnbwd_supported()).
 * replaces the backward of mobilenetv2.py:65-67,79-81 / mbv2_yolo.py:22-24 for those units.
 * K == 5 (MobileNetV3's 5x5 units, mobilenetv3.py:54-56,68-69) runs the TILE form (csrc/dwtile.hip): a workgroup owns a tile of
 * columns x channel groups, dY is rebuilt once per element and shared through a ring of K+1 rows in LDS, every window element
 * feeds the data gradient (gather) and the weight gradient (over the input pixels a thread owns); dw is [C,1,5,5].  On bf16 storage
 * the 3x3 units the tile form is faster on take it too (dwt_use() in dwtile.hip states the rule and the measurements).
 * MNY_NO_DWT5=1 reports K == 5 unsupported (bn_bwd_apply + dw_bwd_weight + dw_bwd_data); MNY_DWT3=1 / 0 forces 3x3 onto / off the tile form.
 * mny_dw_bnbwd_parts_k(..., flags): bit 0 = bf16 storage, bit 1 = called as mny_dw_bnbwd_red (the row count depends on the form that runs);
 * mny_dw_bnbwd_parts() == mny_dw_bnbwd_parts_k(..., 3, 0). */
int mny_dw_bnbwd_supported(int K, int stride);
int mny_dw_bnbwd_parts(int N, int H, int W, int C);
int mny_dw_bnbwd_parts_k(int N, int H, int W, int C, int K, int flags);
int mny_dw_bnbwd(const float* g, const float* y, const float* scale, const float* shift, int act, const float* coef,
                 const float* x, const float* in_scale, const float* in_shift, int in_act, const float* w,
                 const float* addend, float* dx, float* dw, float* ws, int N, int H, int W, int C, int K, int stride,
                 void* stream);
/* The same fusion for a 3x3 STRIDE-2 depthwise unit (x is [N,H,W,C], g / y are [N,Ho,Wo,C] with Ho = (H-1)/2+1): replaces
 * mny_bn_bwd_apply + mny_dw_bwd_weight + mny_dw_bwd_data of the four down-sampling units (models/mobilenetv2.py:65-67 at
 * stride 2).  ws: [mny_dw_bnbwd_s2_parts()][C*9] floats; dw == NULL leaves the partials to mny_reduce_batch. */
int mny_dw_bnbwd_s2_parts(int N, int H, int W, int C);
int mny_dw_bnbwd_s2(const float* g, const float* y, const float* scale, const float* shift, int act, const float* coef,
                    const float* x, const float* in_scale, const float* in_shift, int in_act, const float* w,
                    const float* addend, float* dx, float* dw, float* ws, int N, int H, int W, int C, void* stream);
/* mny_dw_bnbwd + the BN-backward sums of the unit that PRODUCED the input: when `x` is the raw output of a conv+BN+act unit
 * (in_scale/in_shift/in_act = its view, in_mean/in_invstd = its batch statistics) whose only consumer is this depthwise unit,
 * the dX written here is that unit's complete output gradient, and the kernel also leaves its sums (sum dz, sum dz*xhat per
 * channel) as partial rows in_red[mny_dw_bnbwd_parts()][2][C] — the input of mny_bn_bwd_finalize that a separate
 * mny_bn_bwd_reduce pass over (dX, x) would have produced (autograd of BatchNorm2d + ReLU6 at models/mobilenetv2.py:69-71). */
int mny_dw_bnbwd_red(const float* g, const float* y, const float* scale, const float* shift, int act, const float* coef,
                     const float* x, const float* in_scale, const float* in_shift, int in_act, const float* in_mean,
                     const float* in_invstd, const float* w, const float* addend, float* dx, float* dw, float* ws, float* in_red,
                     int N, int H, int W, int C, int K, int stride, void* stream);

/* ---- low-rank BatchNorm backward of a WIDE expand unit (csrc/lrbwd.hip) --------------------------------------------------------
 * autograd of nn.Conv2d(K, C, 1) + nn.BatchNorm2d + ReLU6 / ReLU / h-swish with C >= 2K (models/mobilenetv2.py:73-78,
 * models/mobilenetv3.py:49-51,67) WITHOUT the bn_bwd_apply pass over the C-wide tensors.  The depthwise unit behind it stores
 * dzc = ca o dL/da o act'(z) (ca = gamma * invstd = the unit's forward scale) instead of dL/da (mny_dw_bnbwd_red_dz: same arguments and
 * partial rows as mny_dw_bnbwd_red); then with X the unit's viewed K-wide input and Y = X W^T:
 *     dX = dzc W + X Q + r,    Q = W^T diag(cb) W,  r = cc^T W          (mny_pw_fwd data gradient on dzc, then mny_pw_lr_fix)
 *     dW = dzc^T X + cb o (W G) + cc (x) s,   G = X^T X,  s = colsum(X)   (mny_pw_wgrad on dzc, then mny_lr_wfix)
 * with (ca, cb, cc) = the coefficient rows of mny_bn_bwd_finalize.
 * mny_lr_gram : partial rows [mny_lr_gram_parts(M, K)][K*K + K] = (G | s) of the VIEWED input (combine with mny_reduce_batch).
 * mny_lr_prep : q[K][K] = Q (symmetric, so it is its own NT operand) and r[K], on the fp32 matrix cores.
 * mny_pw_lr_fix : dx = view(x) Q + r + addend (addend may be dx; in_scale / in_shift = the LINEAR view of the unit's input, or NULL, NULL);
 *               red != NULL: the BN-backward sums of the unit whose complete output gradient dx now is (ry = its raw output [M][K], r_* = its
 *               view / statistics), rows [mny_pw_lr_fix_parts(M, K, r_act)][2][K].
 * mny_lr_wfix : dw[C][K] += cb o (W G) + cc (x) s in place; gram_sums = the combined [K*K + K] row of mny_lr_gram.               */
int mny_lr_supported(int64_t M, int K, int C);
int mny_lr_gram_parts(int64_t M, int K);
int mny_lr_gram(const float* x, const float* in_scale, const float* in_shift, int in_act, float* parts, int64_t M, int K, void* stream);
int mny_lr_gram_bf16(const void* x, const float* in_scale, const float* in_shift, int in_act, float* parts, int64_t M, int K, void* stream);
int mny_lr_prep(const float* coef, const float* w, float* q, float* r, int C, int K, void* stream);
int mny_pw_lr_fix_parts(int64_t M, int K, int r_act);
int mny_pw_lr_fix(const float* x, const float* in_scale, const float* in_shift, const float* q, const float* r, const float* addend, float* dx,
                  const float* ry, const float* r_scale, const float* r_shift, int r_act, const float* r_mean, const float* r_invstd, float* red,
                  int64_t M, int K, void* stream);
int mny_lr_wfix(float* dw, const float* gram_sums, const float* coef, const float* w, int C, int K, void* stream);
/* ... and for a 3x3 STRIDE-2 unit (fp32 storage, ReLU6 view of the producer): mny_dw_bnbwd_s2 with dx = in_scale o dX o relu6'(z) and the producer's
 * BN-backward sums as partial rows in_red[mny_dw_bnbwd_s2_parts()][2][C] (models/mobilenetv2.py:73-81 at stride 2: block 96 -> 576 -> 160). */
int mny_dw_bnbwd_s2_red_dz(const float* g, const float* y, const float* scale, const float* shift, int act, const float* coef,
                           const float* x, const float* in_scale, const float* in_shift, int in_act, const float* in_mean,
                           const float* in_invstd, const float* w, const float* addend, float* dx, float* dw, float* ws, float* in_red,
                           int N, int H, int W, int C, void* stream);
/* Fused backward of a depthwise 5x5 STRIDE-2 conv+BN+act unit (models/mobilenetv3.py:88,100: Block(5, 24->72->40, s2), Block(5, 160->672->160, s2);
 * replaces the autograd backward of nn.Conv2d(groups=C, 5, stride 2, pad 2) + nn.BatchNorm2d + ReLU / h-swish): dY rebuilt from (g, y, coef) on
 * chip, one pass over (g, y, x) yields dx (+ addend) and the weight-gradient partials ws[mny_dw_bnbwd_s2k5_parts()][C*25] (dw == NULL leaves them to
 * mny_reduce_batch).  in_red != NULL (with in_mean / in_invstd / in_scale / in_shift of the unit that produced x, consumed only here): that unit's
 * BN-backward sums as partial rows in_red[parts][2][C].  C even. */
int mny_dw_bnbwd_s2k5_parts(int N, int H, int W, int C);
int mny_dw_bnbwd_s2k5(const float* g, const float* y, const float* scale, const float* shift, int act, const float* coef,
                      const float* x, const float* in_scale, const float* in_shift, int in_act, const float* in_mean,
                      const float* in_invstd, const float* w, const float* addend, float* dx, float* dw, float* ws, float* in_red,
                      int N, int H, int W, int C, void* stream);
int mny_dw_bnbwd_s2k5_bf16(const void* g, const void* y, const float* scale, const float* shift, int act, const float* coef,
                           const void* x, const float* in_scale, const float* in_shift, int in_act, const float* in_mean,
                           const float* in_invstd, const float* w, const void* addend, void* dx, float* dw, float* ws, float* in_red,
                           int N, int H, int W, int C, void* stream);
int mny_dw_bnbwd_red_dz_supported(int K, int C, int bf16);
int mny_dw_bnbwd_red_dz(const float* g, const float* y, const float* scale, const float* shift, int act, const float* coef,
                        const float* x, const float* in_scale, const float* in_shift, int in_act, const float* in_mean,
                        const float* in_invstd, const float* w, const float* addend, float* dx, float* dw, float* ws, float* in_red,
                        int N, int H, int W, int C, int K, int stride, void* stream);

/* ---- row padding for the detection heads' backward ---------------------------------------------------------
 * The head gradient dL/dhead is [M][75] (yolo_loss.py:84 channel layout): rows are neither 16-B aligned nor a multiple
 * of 4 wide, which would put its weight- and data-gradient GEMMs on the register-staged kernels.  mny_pad_rows copies it
 * (times *alpha, the upstream dL/dloss; alpha may be NULL) into [M][Cp] with zeroed pad columns, mny_transpose_pad
 * writes W^T as [Cin][Cp] with zeroed pad columns; the GEMMs then run with Cp channels (pad products are 0, the pad rows
 * of dW land in the slack of the caller's gradient slot).                                                      */
int mny_pad_rows(const float* src, const float* alpha, float* dst, int64_t M, int C, int Cp, void* stream);
int mny_transpose_pad(const float* src /*[R][Cc]*/, float* dst /*[Cc][Rp]*/, int R, int Cc, int Rp, void* stream);

/* ---- all weight transposes of a backward pass in one launch ----------------------------------------------------
 * The data-gradient GEMMs read W^T ([Cin][Cout] rows; mny_transpose / mny_transpose_pad per layer).  A static plan
 * knows every (W, W^T) pair up front: jobs (DEVICE array) describe them, block_job (DEVICE int32[nblocks]) maps a
 * 32x32-tile workgroup to its job (job.block0 = its first workgroup; a job has ceil(Cc/32)*ceil(Rp/32) of them). */
/* Deferred partial combines.  Every weight-gradient entry point (mny_pw_wgrad, mny_dw_bwd_weight, mny_dw_bnbwd[_red],
 * mny_stem_wgrad, mny_stem_bnwgrad and their _bf16 twins) writes per-workgroup partial sums [nparts][n] to `ws` and then
 * combines them (fp64, fixed order) into dW.  Called with dw == NULL it stops after the partials (mny_pw_wgrad: only without
 * dbias); a static plan gives each layer its own `ws` and combines a whole backward segment in ONE launch: jobs (DEVICE
 * array) name the partials, block_job (DEVICE int32[nblocks]) maps a workgroup (32 outputs) to its job (job.block0 = its first
 * workgroup, ceil(n/32) of them).  Same arithmetic and order as the per-layer combine. */
typedef struct mny_reduce_job {
    const float* parts; /* [nparts][n] */
    float* out;         /* [n] */
    int64_t n;
    int32_t nparts, block0;
} mny_reduce_job;
int mny_reduce_batch(const mny_reduce_job* jobs, const int32_t* block_job, int nblocks, void* stream);
int mny_pw_wgrad_splits(int64_t M, int K, int Nc);      /* nparts of mny_pw_wgrad's partials (n = Nc * K) */

typedef struct mny_transpose_job {
    const float* src; /* [R][Cc] fp32 */
    void* dst;        /* [Cc][Rp] fp32 (or bf16 for the _bf16 entry point), columns R..Rp-1 zeroed */
    int32_t R, Cc, Rp, block0;
} mny_transpose_job;
int mny_transpose_batch(const mny_transpose_job* jobs, const int32_t* block_job, int nblocks, void* stream);

/* ---- fused multi-tensor AdamW --------------------------------------------------------------
 * replaces optim.AdamW(...).step() at train.py:134,283 (torch semantics: decoupled weight decay, bias-corrected
 * moments, amsgrad off).  `table_dev` is a DEVICE array of chunks; one workgroup updates one chunk (callers split
 * large tensors into chunks of <= 64 Ki elements).  `vec4` = all four pointers 16-B aligned.  `step` is the
 * 1-based step count used for the bias corrections.                                                            */
typedef struct mny_adamw_chunk {
    float* p;
    const float* g;
    float* m;
    float* v;
    int n;
    int vec4;
} mny_adamw_chunk;
int mny_adamw_step(const mny_adamw_chunk* table_dev, int nchunks, double lr, double beta1, double beta2, double eps,
                   double weight_decay, int64_t step, void* stream);

/* ---- data-gradient GEMM + BN-backward reduction of the unit it feeds ------------------------------------------
 * dx[M,Nc] = dy[M,K] * W, with W^T given as [Nc][K] rows (mny_transpose of the conv weight) — the autograd
 * data gradient of nn.Conv2d(Nc,K,1) (mobilenetv2.py:69,83) — when dx is the complete gradient of a
 * conv+BN+act unit's output: the epilogue also forms that unit's BN-backward sums from the tile it just
 * produced and the unit's raw output y[M,Nc]:  red[part][0][c] = sum dz, red[part][1][c] = sum dz*xhat,
 * dz = dx * act'(scale*y+shift), xhat = (y-mean)*invstd — exactly what mny_bn_bwd_reduce(dx, y, ...) writes,
 * in the same [parts][2][Nc] layout (parts = mny_pw_dgrad_bnred_parts(M,K,Nc)); feed it to mny_bn_bwd_finalize.
 * Saves the reduce pass's read of dx.  K % 4 == 0 (K % 8 for the bf16 twin), any activation. */
int mny_pw_dgrad_bnred_supported(int64_t M, int K, int Nc, int act);
int mny_pw_dgrad_bnred_parts(int64_t M, int K, int Nc); /* partial rows it writes */
int mny_pw_dgrad_bnred(const float* dy, const float* wT, float* dx, const float* y, const float* scale,
                       const float* shift, int act, const float* mean, const float* invstd, float* red,
                       int64_t M, int K, int Nc, void* stream);
/* ... with an addend: dx = dy * W + addend is the LAST contribution to that unit's output gradient (the others — a residual
 * branch — were accumulated into `addend`), and the sums are taken over the complete dx (torch.add at mobilenetv2.py:89 followed
 * by the BatchNorm backward of the block's project conv). */
int mny_pw_dgrad_bnred_add_supported(int64_t M, int K, int Nc, int act);
int mny_pw_dgrad_bnred_add(const float* dy, const float* wT, const float* addend, float* dx, const float* y, const float* scale,
                           const float* shift, int act, const float* mean, const float* invstd, float* red, int64_t M, int K, int Nc,
                           void* stream);

/* ---- fp32 GEMMs on pre-cut weight planes ---------------------------------------------------------------------------
 * The MFMA-bound fp32 GEMMs run on the bf16 matrix cores as six partial products per fp32 product (every operand cut exactly into
 * three bf16 pieces).  The weight operand is the same for every row tile: a static plan cuts all of its GEMM weights
 * (and their transposes) ONCE per pass with mny_cut3_batch — jobs (DEVICE array) name fp32 matrices [R][C] and their plane buffers of
 * mny_pw_w6_bytes(C, R) bytes, block_job (DEVICE int32[nblocks]) maps a workgroup (256 16-byte chunks) to its job (job.block0 = its
 * first workgroup, ceil(R * ceil(C/16) * 2 / 256) of them) — and calls the _w6 twins, which take the planes where mny_pw_fwd /
 * mny_pw_dgrad_bnred[_add] take the fp32 matrix.  Same tiling, same partial rows (mny_pw_stat_parts / mny_pw_dgrad_bnred_parts), same
 * results up to fp32 rounding order.  mny_pw_w6_supported(M, K, Nc): the shape takes this form (fp32, K % 4 == 0, >= 20 FLOP per byte). */
typedef struct mny_cut3_job {
    const float* src; /* [R][C] fp32 */
    void* dst;        /* three bf16 planes, mny_pw_w6_bytes(C, R) bytes */
    int32_t R, C, block0, pad;
} mny_cut3_job;
int mny_pw_w6_supported(int64_t M, int K, int Nc);
size_t mny_pw_w6_bytes(int K, int Nc);
int mny_cut3_batch(const mny_cut3_job* jobs, const int32_t* block_job, int nblocks, void* stream);
int mny_pw_fwd_w6(const float* x, const float* in_scale, const float* in_shift, int in_act, const void* w6, const float* bias,
                  const float* addend, float* y, float* stats, int64_t M, int K, int Nc, void* stream);
int mny_pw_dgrad_bnred_w6(const float* dy, const void* wT6, const float* addend, float* dx, const float* y, const float* scale,
                          const float* shift, int act, const float* mean, const float* invstd, float* red, int64_t M, int K, int Nc,
                          void* stream);

/* ---- which kernel family a pointwise-conv call takes (pure host query; tests and tools/plan_stats.py use it to PROVE that a
 * plan compared with the oracle contains the kernels the benchmark runs) ----------------------------------------------------------
 * op: 0 = mny_pw_fwd (forward / plain data gradient of nn.Conv2d(K,Nc,1), mobilenetv2.py:63-85), 1 = mny_pw_dgrad_bnred[_add],
 * 2 = mny_pw_wgrad (no bias gradient).  bf16: the storage type of the plan (0 = fp32).  Returns one of MNY_ROUTE_*. */
enum {
    MNY_ROUTE_TILE_V1 = 0,      /* register-staged tile kernels (unaligned channel counts) */
    MNY_ROUTE_DMA_F32 = 1,      /* LDS-DMA tile kernel, fp32 MFMA (or the bf16 MFMA with bf16 storage) */
    MNY_ROUTE_DMA_X6 = 2,       /* LDS-DMA tile kernel, six-product bf16 form */
    MNY_ROUTE_THIN = 3,         /* short-reduction vector-ALU stream kernel (pwthin.hip) */
    MNY_ROUTE_WIDE = 4,         /* barrier-free wide-output kernel (pwwide.hip) */
    MNY_ROUTE_WGRAD_STREAM = 5, /* barrier-free stream weight-gradient kernel (pwwgs.hip) */
    MNY_ROUTE_WAVE16 = 6        /* bf16 storage, K <= 48 at >= 131072 pixels: a wave per 16 pixels on the bf16 matrix cores (gate.hip pwt_fwd_kernel) */
};
int mny_pw_route(int op, int bf16, int64_t M, int K, int Nc);   /* PREDICTION for a plain view (no h-swish input, no bias gradient) */
/* the family the LAST pointwise-conv call of the calling thread actually took (recorded by the dispatchers where they decide; -1 before
 * the first call).  engine.py records it for every call of a plan's first replay: NetPlan.kernel_routes() reports what ran. */
int mny_pw_last_route(void);

/* ---- evaluation consumer: VOC07 11-point mAP on the device -----------------------------
 * Replaces utils/eval_mAP.py:134-187 (calculate_mAP), :69-132 (eval_class_ap), :8-65
 * (eval_single_image_recall) and utils/iou.py:4-48 (find_jaccard_overlap) on a PACKED layout: the
 * reference's per-image tensor lists laid back to back, image i = [off[i], off[i+1]).
 * n_classes counts the background entry (train.py:58); classes 1..n_classes-1 are evaluated, labels are
 * float like the reference's tensors (a label that is not an integer in that range is never evaluated).
 * Semantics kept: within an image detections are matched in STORED order; best ground truth = first
 * maximum of the fp32 IoU (NaN -> false positive); IoU > 0.5; a hit on a `difficult` (true_diff != 0)
 * object counts as neither; a second hit on an object is a false positive.  Score ties in the global sort
 * keep (image, stored) order (torch.sort leaves them unspecified).
 * Outputs (device): ap[n_classes-1], tp_sum / fp_sum [n_classes-1] (exact counts as float),
 * prec11 [(n_classes-1)*11] (the 11 interpolated precisions), mean_ap[1].
 * Limits: n_classes <= 255, D <= 2^30.  ws: mny_map_ws_bytes(D, T) bytes.  No host sync. */
size_t mny_map_ws_bytes(int64_t D, int64_t T);
int mny_map_eval(const float* det_boxes /*[D,4]*/, const float* det_labels, const float* det_scores,
                 const int32_t* det_off /*[n_images+1]*/, const float* true_boxes /*[T,4]*/,
                 const float* true_labels, const float* true_diff, const int32_t* true_off, int n_images,
                 int64_t D, int64_t T, int n_classes, float* ap, float* tp_sum, float* fp_sum, float* prec11,
                 float* mean_ap, void* ws, void* stream);
/* train.py:371-385 (the glue of test() between the detector and calculate_mAP): detection rows
 * [D,7] (x1,y1,x2,y2,obj,cls_conf,cls) -> boxes, label = cls+1, score = obj*cls_conf; targets [T,5]
 * (cls,cx,cy,w,h) -> corner boxes, label = cls as stored, difficulty 0.  Either half may be empty. */
int mny_eval_pack(const float* rows, int64_t D, const float* targets, int64_t T, float* det_boxes,
                  float* det_labels, float* det_scores, float* true_boxes, float* true_labels,
                  float* true_diff, void* stream);

/* ---- segmentation head of the BDD100K config ------------------------------------------------
 * Replaces SegLoss.forward (models/seg_loss.py:51-80) and its autograd on the channels-last seg head
 * [N,h,w,C] (the reference permutes seg_maps [N,h,w,C] to NCHW instead, :54): n = N*h*w*C elements.
 * out3 = (0.05 * mean((sigmoid(x)-t)^2), mean sigmoid where t >= 0.5, mean sigmoid where t < 0.5) — an empty
 * selection gives NaN like torch.mean; dhead = 0.05*2*(sigmoid(x)-t)/n: the reference's custom sigmoid passes the
 * gradient through unchanged (:24-32).  ws: mny_seg_loss_ws_bytes(n).  Deterministic (fixed-order fp64 sums). */
size_t mny_seg_loss_ws_bytes(int64_t n);
int mny_seg_loss(const float* head, const float* seg_maps, int64_t n, float* out3, float* dhead, void* ws,
                 void* stream);
/* eval branch (:77-80): sigmoid of image 0 only, written channel-major [C,h,w] like the reference's numpy result */
int mny_seg_sigmoid(const float* head /*[N,h,w,C], image 0 is read*/, int h, int w, int C, float* out,
                    void* stream);

/* ---- device-side batch input preparation ------------------------------------------------------
 * Replaces the image half of collate_fn (folder2lmdb.py:223-256): per image transforms.Resize(size, BILINEAR) on
 * the decoded PIL image, ToTensor, Normalize(mean, std), torch.stack — with the batch's one
 * random.choice(train_img_size) made by the caller.  The resize is Pillow's ImagingResample (third-party, not
 * vendored): antialiased triangle filter whose support grows with the down-scale factor, 22-bit fixed-point taps,
 * horizontal pass then vertical pass, each rounded to uint8 — reproduced bit for bit; then
 * out = (u8/255 - mean)/std in fp32 (same divisions as torch).
 * src: decoded RGB uint8 images, HWC, anywhere in one device buffer; desc (DEVICE, [N]): byte offset + size of
 * each.  max_in_h/max_in_w: caller's upper bounds on the source sizes (size the workspace and the tap tables); an
 * image outside them is written as zeros and the int32 at ws+0 receives 1+its index (0 = ok).
 * mean3/std3: HOST arrays of 3 floats, read during the call.  out: [N,3,out_h,out_w] fp32 (NCHW, what
 * mny_stem_fwd reads).  ws: mny_prep_ws_bytes().  No host sync. */
typedef struct mny_image_desc {
    int64_t offset; /* bytes from src to pixel (0,0) */
    int32_t h, w;
} mny_image_desc;
size_t mny_prep_ws_bytes(int N, int max_in_h, int max_in_w, int out_h, int out_w);
int mny_prep_batch(const uint8_t* src, const mny_image_desc* desc, int N, int max_in_h, int max_in_w, int out_h,
                   int out_w, const float* mean3, const float* std3, float* out, void* ws, void* stream);

/* ---- bf16 STORAGE twins (BASELINE config 4: MobileNetV3-YOLO 512x512 bf16) -----------------
 * Every `mny_X_bf16` has the contract of `mny_X` above with ONE difference: the activation-sized tensors (the
 * `void*` parameters: raw conv outputs, materialised sums, gradients wrt activations) are bf16 in HBM.  Kernels
 * widen on load, compute and accumulate in fp32 and round once (RNE) on store; BN statistics are taken over the
 * ROUNDED outputs (what the consumer will read).  Weights, biases, view coefficients, statistics, workspaces and
 * parameter gradients stay fp32, so optimizers/checkpoints are unchanged.  The detection heads are converted to
 * fp32 (mny_cvt_bf16_f32) before mny_yolo_loss / mny_yolo_decode, whose gradient re-enters through
 * mny_cvt_f32_bf16.  Channel counts need the same alignment as the fp32 entry points (C % 4 == 0 for the
 * stencil / elementwise kernels; any K, N for the pointwise GEMMs).                                          */
int mny_stem_fwd_bf16(const float* x_nchw, const float* w, void* y, float* stats, int N, int H, int W, int Cout, void* stream);
int mny_stem_wgrad_bf16(const float* x_nchw, const void* dy, float* dw, float* ws, int N, int H, int W, int Cout, void* stream);
int mny_dw_fwd_bf16(const void* x, const float* in_scale, const float* in_shift, int in_act, const float* w, void* y,
                    float* stats, int N, int H, int W, int C, int K, int stride, void* stream);
int mny_dw_bwd_data_bf16(const void* dy, const float* w, const void* addend, void* dx, int N, int H, int W, int C, int K,
                         int stride, void* stream);
int mny_dw_bwd_weight_bf16(const void* x, const float* in_scale, const float* in_shift, int in_act, const void* dy,
                           float* dw, float* ws, int N, int H, int W, int C, int K, int stride, void* stream);
/* the pointwise GEMM also takes its WEIGHTS in bf16 ([Nc][K], e.g. mny_cvt_f32_bf16 of the fp32 master copy, or
 * mny_transpose_bf16 for the data-gradient): K % 8 == 0 runs LDS-DMA + v_mfma_f32_32x32x16_bf16, other K the
 * register-staged fp32-MFMA kernel */
int mny_pw_fwd_bf16(const void* x, const float* in_scale, const float* in_shift, int in_act, const void* w_bf16,
                    const float* bias, const void* addend, void* y, float* stats, int64_t M, int K, int Nc, void* stream);
int mny_transpose_bf16(const float* src /*[R][Cc] fp32*/, void* dst /*[Cc][R] bf16*/, int R, int Cc, void* stream);
int mny_pw_stat_parts_bf16(int64_t M, int K, int Nc);
int mny_pw_wgrad_bf16(const void* x, const float* in_scale, const float* in_shift, int in_act, const void* dy, float* dw,
                      float* dbias, float* ws, int64_t M, int K, int Nc, void* stream);
int mny_bn_bwd_reduce_bf16(const void* g, const void* y, const float* scale, const float* shift, int act, const float* mean,
                           const float* invstd, float* red, int64_t M, int C, void* stream);
int mny_bn_bwd_apply_bf16(const void* g, const void* y, const float* scale, const float* shift, int act, const float* coef,
                          void* dy, int64_t M, int C, void* stream);
int mny_add_views_bf16(const void* a, const float* a_scale, const float* a_shift, int a_act, const void* b,
                       const float* b_scale, const float* b_shift, int b_act, const void* up, void* out, int N, int H, int W,
                       int C, void* stream);
int mny_mul_views_bf16(const void* a, const float* a_scale, const float* a_shift, int a_act, const void* b,
                       const float* b_scale, const float* b_shift, int b_act, void* out, int64_t M, int C, void* stream);
int mny_mul_views_bwd_bf16(const void* g, const void* o, const float* o_scale, const float* o_shift, int o_act,
                           const void* addend, void* dst, int64_t M, int C, void* stream);
int mny_partadd_up_bf16(const void* a, const float* a_scale, const float* a_shift, int a_act, const void* up, void* out,
                        int N, int H, int W, int Ca, int Cb, void* stream);
int mny_slice_channels_bf16(const void* src, void* dst, int accumulate, int64_t M, int Ca, int Cb, void* stream);
int mny_upsample_bwd_bf16(const void* src, void* dst, int accumulate, int N, int H, int W, int C, void* stream);
int mny_dw_bnbwd_bf16(const void* g, const void* y, const float* scale, const float* shift, int act, const float* coef,
                      const void* x, const float* in_scale, const float* in_shift, int in_act, const float* w,
                      const void* addend, void* dx, float* dw, float* ws, int N, int H, int W, int C, int K, int stride,
                      void* stream);
int mny_dw_bnbwd_s2_bf16(const void* g, const void* y, const float* scale, const float* shift, int act, const float* coef,
                         const void* x, const float* in_scale, const float* in_shift, int in_act, const float* w,
                         const void* addend, void* dx, float* dw, float* ws, int N, int H, int W, int C, void* stream);
int mny_dw_bnbwd_red_bf16(const void* g, const void* y, const float* scale, const float* shift, int act, const float* coef,
                          const void* x, const float* in_scale, const float* in_shift, int in_act, const float* in_mean,
                          const float* in_invstd, const float* w, const void* addend, void* dx, float* dw, float* ws, float* in_red,
                          int N, int H, int W, int C, int K, int stride, void* stream);
int mny_pad_rows_bf16(const float* src, const float* alpha, void* dst, int64_t M, int C, int Cp, void* stream);
int mny_transpose_pad_bf16(const float* src, void* dst, int R, int Cc, int Rp, void* stream);
int mny_stem_bnwgrad_bf16(const float* x_nchw, const void* g, const void* y, const float* scale, const float* shift, int act,
                          const float* coef, float* dw, float* ws, int N, int H, int W, int Cout, void* stream);
/* fused BN-backward + weight gradient + data gradient of a thin expand unit (mny_pw_bnbwd) under bf16 storage: g, y, x, addend, dx are
 * bf16; w, statistics, dw / dgamma / dbeta and the workspace (mny_pw_bnbwd_ws_floats) fp32.  K in {8,16,24,32}, N in {64,72,96,144,192}:
 * the ReLU expand units of models/mobilenetv3.py:44-74 (16->64, 24->72) and the ReLU6 ones of models/mobilenetv2.py:75-77. */
int mny_pw_bnbwd_supported_bf16(int64_t M, int K, int Nc);
int mny_pw_bnbwd_bf16(const void* g, const void* y, const float* scale, const float* shift, int act, const float* mean, const float* invstd,
                      const float* gamma, const void* x, const float* in_scale, const float* in_shift, int in_act, const float* w,
                      const void* addend, void* dx, float* dw, float* dgamma, float* dbeta, float* ws, int64_t M, int K, int Nc, void* stream);
int mny_pw_dgrad_bnred_supported_bf16(int64_t M, int K, int Nc, int act);
int mny_pw_dgrad_bnred_parts_bf16(int64_t M, int K, int Nc);
int mny_pw_dgrad_bnred_bf16(const void* dy, const void* wT, void* dx, const void* y, const float* scale, const float* shift, int act,
                            const float* mean, const float* invstd, float* red, int64_t M, int K, int Nc, void* stream);
int mny_pw_dgrad_bnred_add_supported_bf16(int64_t M, int K, int Nc, int act);
int mny_pw_dgrad_bnred_add_bf16(const void* dy, const void* wT, const void* addend, void* dx, const void* y, const float* scale,
                                const float* shift, int act, const float* mean, const float* invstd, float* red, int64_t M, int K,
                                int Nc, void* stream);
int mny_transpose_batch_bf16(const mny_transpose_job* jobs, const int32_t* block_job, int nblocks, void* stream);
int mny_pw_wgrad_splits_bf16(int64_t M, int K, int Nc);
int mny_axpy_bf16(const void* src, const float* alpha, void* dst, int accumulate, int64_t n, void* stream);
/* all bf16 weight shadows of a forward pass in one launch: block_job maps a workgroup (4096 elements) to its job */
typedef struct mny_cvt_job {
    const float* src;
    void* dst; /* bf16 */
    int64_t n;
    int32_t block0, pad_;
} mny_cvt_job;
int mny_cvt_batch_f32_bf16(const mny_cvt_job* jobs, const int32_t* block_job, int nblocks, void* stream);
/* element-wise storage conversion, n elements (RNE to bf16, exact widening back) */
int mny_cvt_f32_bf16(const float* src, void* dst, int64_t n, void* stream);
int mny_cvt_bf16_f32(const void* src, float* dst, int64_t n, void* stream);

/* ---- frozen BatchNorm: a module in eval() whose losses are differentiated (models/mbv2_yolo.py:157 returns losses with a graph under
 * model.eval(); nn.BatchNorm2d then normalises with the running statistics, which are constants of the step).  mny_bn_eval_stats fills
 * the mean / invstd the backward kernels read (mean = running_mean, invstd = 1 / sqrt(running_var + eps)); mny_bn_bwd_finalize_frozen
 * has mny_bn_bwd_finalize's argument list and yields dgamma = sum dz * yhat, dbeta = sum dz, coef = (gamma * invstd, 0, 0). */
int mny_bn_eval_stats(const float* running_mean, const float* running_var, float eps, float* mean_out, float* invstd_out, int C,
                      void* stream);
int mny_bn_bwd_finalize_frozen(const float* red, int parts, int64_t count, const float* gamma, const float* mean, const float* invstd,
                               float* dgamma, float* dbeta, float* coef, int C, void* stream);
/* ---- expand + depthwise as one unit (csrc/exdw.hip): the front half of an inverted-residual block whose depthwise conv has
 * stride 2 — 1x1 conv K -> C = 6K (K in {16, 24, 32}) + BN + ReLU6 + depthwise 3x3 stride 2 (models/mobilenetv2.py:73-85) — with the
 * 6x-wide expand output and its gradient NEVER materialised: every pass recomputes it from the thin input x[N,H,W,K] (a view:
 * in_scale / in_shift / in_act as everywhere), bit-identical to what mny_pw_fwd would have stored.  fp32 storage, even H and W.
 *   mny_exdw_stats : BN batch statistics of the expand output -> partial rows [mny_exdw_stat_parts()][2][C] for mny_bn_finalize;
 *                    derived from the K x K second-moment matrix and the column sums of the viewed input (sum y^2 = w^T (x^T x) w, fp64
 *                    per workgroup): one read of x, no recomputation of y.  MNY_EXDW_STATS=direct sums the recomputed y instead.
 *   mny_exdw_fwd   : z[N,H/2,W/2,C] = dw3x3_s2(relu6(e_scale * (x w_exp^T) + e_shift)), z statistics -> [mny_exdw_fwd_parts()][2][C]
 *   mny_exdw_bwd   : given gz = dL/d act(z_scale z + z_shift) and the depthwise unit's BN-backward coefficients z_coef[3][C]
 *                    (mny_bn_bwd_finalize): dw_dw[C,3,3] (or, dw_dw == NULL, partial rows [mny_exdw_bwd_parts()][C*9] left in dw_ws),
 *                    the expand unit's dw_exp[C,K], dgamma_e, dbeta_e, and dx[N,H,W,K] = data gradient wrt the viewed input (+ addend).
 *                    ws: mny_exdw_bwd_ws_floats() floats; dw_ws: mny_exdw_bwd_parts() * C * 9 floats.  addend must NOT alias dx
 *                    (two passes: dx is overwritten before the addend is read) — MNY_EINVAL. */
int mny_exdw_supported(int N, int H, int W, int K, int C, int stride);
int mny_exdw_stat_parts(int64_t M, int K, int C);
int mny_exdw_stats(const float* x, const float* in_scale, const float* in_shift, int in_act, const float* w_exp, float* stats,
                   int64_t M, int K, int C, void* stream);
int mny_exdw_fwd_parts(int N, int H, int W, int K, int C, int stride);
int mny_exdw_fwd(const float* x, const float* in_scale, const float* in_shift, int in_act, const float* w_exp,
                 const float* e_scale, const float* e_shift, const float* w_dw, float* z, float* z_stats,
                 int N, int H, int W, int K, int C, int stride, void* stream);
int mny_exdw_bwd_parts(int N, int H, int W, int K, int C, int stride);
size_t mny_exdw_bwd_ws_floats(int N, int H, int W, int K, int C, int stride);
int mny_exdw_bwd(const float* gz, const float* z, const float* z_scale, const float* z_shift, int z_act, const float* z_coef,
                 const float* x, const float* in_scale, const float* in_shift, int in_act, const float* w_exp,
                 const float* e_scale, const float* e_shift, const float* e_mean, const float* e_invstd, const float* e_gamma,
                 const float* w_dw, const float* addend, float* dx, float* dw_exp, float* dgamma_e, float* dbeta_e,
                 float* dw_dw, float* dw_ws, float* ws, int N, int H, int W, int K, int C, int stride, void* stream);
/* the same where x is the RAW output of a conv+BN(+act) unit consumed only by this one (view in_scale / in_shift / in_act, statistics
 * in_mean / in_invstd): dx is then that unit's complete output gradient, and its BN-backward sums (mny_bn_bwd_reduce's: sum dz, sum dz*yhat)
 * are left in in_red as partial rows [mny_exdw_bwd_red_parts()][2][K] for mny_bn_bwd_finalize — no separate reduce pass over dx and x
 * (the project conv in front of the first expand unit, models/mobilenetv2.py:69-85).  Like mny_dw_bnbwd_red. */
int mny_exdw_bwd_red_parts(int N, int H, int W, int K, int C, int stride);
int mny_exdw_bwd_red(const float* gz, const float* z, const float* z_scale, const float* z_shift, int z_act, const float* z_coef,
                     const float* x, const float* in_scale, const float* in_shift, int in_act, const float* in_mean, const float* in_invstd,
                     const float* w_exp, const float* e_scale, const float* e_shift, const float* e_mean, const float* e_invstd,
                     const float* e_gamma, const float* w_dw, const float* addend, float* dx, float* dw_exp, float* dgamma_e, float* dbeta_e,
                     float* dw_dw, float* dw_ws, float* ws, float* in_red, int N, int H, int W, int K, int C, int stride, void* stream);

/* ---- stem + first depthwise unit, backward as one pass (csrc/stemdw.hip): the autograd of nn.Conv2d(3,32,3,2,1) + BN + ReLU6
 * (models/mobilenetv2.py:40,113) followed by the depthwise nn.Conv2d(32,32,3,1,1,groups=32) + BN + ReLU6 of the first InvertedResidual
 * (:65-67), given gd = dL/d act(d_scale d + d_shift) (the depthwise unit's output gradient), the depthwise unit's BN-backward
 * coefficients d_coef[3][32] (mny_bn_bwd_finalize), the raw outputs d (depthwise) and s (stem) with their BN coefficients, and the
 * NCHW fp32 image.  Replaces mny_dw_bnbwd_red + mny_bn_bwd_finalize + mny_stem_bnwgrad for that pair: the stem's output gradient is
 * never written (dW_stem = ca o (dz^T P) + cb o (W P^T P) + cc (x) colsum(P), P = the 27-value image patches; fp64 finalize).
 * Outputs: dw_stem[32,3,3,3], dgamma_s[32], dbeta_s[32], dw_dw[32,3,3] (or, dw_dw == NULL, partial rows
 * [mny_stemdw_bwd_parts()][32*9] left in dw_ws for mny_reduce_batch).  ws: mny_stemdw_bwd_ws_floats() floats; dw_ws:
 * mny_stemdw_bwd_parts() * 288 floats.  fp32 storage, Cout = 32, activations of the ReLU / ReLU6 / leaky family. */
int mny_stemdw_supported(int N, int H, int W, int Cout, int s_act, int d_act);
int mny_stemdw_bwd_parts(int N, int H, int W, int Cout);
size_t mny_stemdw_bwd_ws_floats(int N, int H, int W, int Cout);
int mny_stemdw_bwd(const float* gd, const float* d, const float* d_scale, const float* d_shift, int d_act, const float* d_coef,
                   const float* s, const float* s_scale, const float* s_shift, const float* s_mean, const float* s_invstd,
                   const float* s_gamma, int s_act, const float* x_nchw, const float* w_stem, const float* w_dw,
                   float* dw_stem, float* dgamma_s, float* dbeta_s, float* dw_dw, float* dw_ws, float* ws,
                   int N, int H, int W, int Cout, void* stream);

/* ---- project-conv unit, backward as one pass (csrc/pjbwd.hip): the autograd of the linear bottleneck nn.Conv2d(Ki,No,1) + BN that
 * closes an inverted-residual block (models/mobilenetv2.py:69-70,83-84), given g = dL/d(BN output) [M,No], the unit's raw conv output
 * y [M,No], its BN-backward coefficients coef[3][No] (mny_bn_bwd_finalize), and the raw output d [M,Ki] of the unit in front (the
 * depthwise unit; its view = d_scale / d_shift / d_act, its statistics d_mean / d_invstd), consumed only here.  Replaces
 * mny_bn_bwd_apply + mny_pw_dgrad_bnred + mny_pw_wgrad for that unit: gd[M,Ki] = dY W (the gradient wrt the activated d), the
 * BN-backward sums of the unit in front as partial rows red[mny_pj_bwd_parts()][2][Ki] (for mny_bn_bwd_finalize), and
 * dw[No,Ki] = dY^T act(BN(d)) (or, dw == NULL, partial rows [mny_pj_bwd_parts()][No*Ki] left in dw_ws).  d is read once, dY never
 * touches HBM.  No in {16,24,32}, Ki in {32,96,144,192}, fp32 storage. */
int mny_pj_bwd_supported(int64_t M, int Ki, int No, int d_act);
int mny_pj_bwd_parts(int64_t M, int Ki, int No);
int mny_pj_bwd(const float* g, const float* y, const float* coef, const float* d, const float* d_scale, const float* d_shift,
               const float* d_mean, const float* d_invstd, int d_act, const float* w, float* gd, float* dw, float* dw_ws, float* red,
               int64_t M, int Ki, int No, void* stream);
/* the same on bf16 storage (csrc/gate.hip, the wave-per-16-pixels matrix-core machinery of the gate unit): MobileNetV3's thin project
 * convs (models/mobilenetv3.py:57-58,69), (Ki, No) in {(16,16), (64,24), (72,24), (72,40), (120,40)}, any view activation but h-sigmoid;
 * the BN-backward sums of the unit in front are taken over the STORED (bf16) gd like mny_pw_dgrad_bnred_bf16's; dw's operands are the
 * bf16 values the forward GEMM multiplied. */
int mny_pj_bwd_supported_bf16(int64_t M, int Ki, int No, int d_act);
int mny_pj_bwd_parts_bf16(int64_t M, int Ki, int No);
int mny_pj_bwd_bf16(const void* g, const void* y, const float* coef, const void* d, const float* d_scale, const float* d_shift,
                    const float* d_mean, const float* d_invstd, int d_act, const float* w, void* gd, float* dw, float* dw_ws, float* red,
                    int64_t M, int Ki, int No, void* stream);

/* ---- MobileNetV3's per-pixel gate as one unit (csrc/gate.hip), bf16 storage: models/mobilenetv3.py:26-41 (SeModule whose avg_pool is
 * never called: the gate is per pixel) applied to the project conv's output (:69-71) —
 *     t = s3 * y3 + b3;  h = relu(BN1(W1 t));  gate = relu6(BN2(W2 h) + 3) / 6;  out = t * gate (+ view(add_x): the residual add of :72)
 * with the hidden tensors (and, backward, their gradients) never written.  Training-mode BatchNorm needs the batch statistics of W1 t
 * and of W2 h before they are used, so the forward is three streaming passes over y3 [M,C] with mny_bn_finalize in between:
 *   mny_gate_stats1_bf16 -> partial rows [mny_gate_parts(M)][2][R] of W1 t   (then mny_bn_finalize(..., R) -> sc1, sh1, mean1, invstd1)
 *   mny_gate_stats2_bf16 -> partial rows [mny_gate_parts(M)][2][C] of W2 h   (then mny_bn_finalize(..., C) -> sc2, sh2, mean2, invstd2)
 *   mny_gate_fwd_bf16    -> out [M,C]
 * and the backward, given dout = dL/d(out) [M,C] (the residual operand's gradient is dout itself), three more with
 * mny_bn_bwd_finalize in between:
 *   mny_gate_bwd1_bf16 -> BN2's backward sums, partial rows red2 [mny_gate_bwd_parts(M)][2][C]   (finalize -> dgamma2, dbeta2, coef2[3][C])
 *   mny_gate_bwd2_bf16 -> BN1's backward sums red1 [..][2][R] (finalize -> dgamma1, dbeta1, coef1[3][R]) and dW2 as partial
 *                         rows dw2_parts [mny_gate_bwd_parts(M)][C*R] (mny_reduce_batch)
 *   mny_gate_bwd3_bf16 -> dt [M,C] = dL/d(t) (both uses of t), dW1 as partial rows dw1_parts [..][R*C], and optionally
 *                         (red3 != NULL, mny_gate_bwd_red3_supported) the project unit's own BN-backward sums red3 [..][2][C] over
 *                         (dt, y3) — what a separate mny_bn_bwd_reduce pass would compute (mean3 / invstd3: its statistics).
 * wq: the two 1x1 conv weights (w1 [R,C], w2 [C,R], fp32 masters) cut into bf16 matrix-core operands, mny_gate_wq_bytes(C,R) bytes,
 * refreshed once per pass for all gates of a plan by mny_gate_cut_batch_bf16 (the gate's counterpart of the GEMM path's bf16 shadows).
 * (C, R) in {(40,10), (112,28), (160,40)} = MobileNetV3-Large's gates. */
typedef struct mny_gate_cut_job {
    const float* w1; /* [R][C] */
    const float* w2; /* [C][R] */
    void* wq;        /* mny_gate_wq_bytes(C, R) bytes */
    int32_t C, R;
} mny_gate_cut_job;
int mny_gate_supported(int64_t M, int C, int R);
int mny_gate_parts(int64_t M);
int mny_gate_bwd_parts(int64_t M);
int mny_gate_bwd_red3_supported(int C, int R);
size_t mny_gate_wq_bytes(int C, int R);
int mny_gate_cut_batch_bf16(const mny_gate_cut_job* jobs, int njobs, void* stream);
int mny_gate_stats1_bf16(const void* y3, const float* s3, const float* b3, const void* wq, float* stats, int64_t M, int C, int R,
                         void* stream);
int mny_gate_stats2_bf16(const void* y3, const float* s3, const float* b3, const void* wq, const float* sc1, const float* sh1,
                         float* stats, int64_t M, int C, int R, void* stream);
int mny_gate_fwd_bf16(const void* y3, const float* s3, const float* b3, const void* wq, const float* sc1, const float* sh1,
                      const float* sc2, const float* sh2, const void* add_x, const float* add_scale, const float* add_shift,
                      int add_act, void* out, int64_t M, int C, int R, void* stream);
int mny_gate_bwd1_bf16(const void* y3, const float* s3, const float* b3, const void* dout, const void* wq, const float* sc1,
                       const float* sh1, const float* sc2, const float* sh2, const float* mean2, const float* invstd2, float* red2,
                       int64_t M, int C, int R, void* stream);
int mny_gate_bwd2_bf16(const void* y3, const float* s3, const float* b3, const void* dout, const void* wq, const float* sc1,
                       const float* sh1, const float* mean1, const float* invstd1, const float* sc2, const float* sh2,
                       const float* coef2, float* red1, float* dw2_parts, int64_t M, int C, int R, void* stream);
int mny_gate_bwd3_bf16(const void* y3, const float* s3, const float* b3, const void* dout, const void* wq, const float* sc1,
                       const float* sh1, const float* sc2, const float* sh2, const float* coef2, const float* coef1,
                       const float* mean3, const float* invstd3, void* dt, float* dw1_parts, float* red3, int64_t M, int C, int R,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MNYOLO_H */
