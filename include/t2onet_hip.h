/* t2onet_hip.h -- C ABI of libt2onet_hip.so, the MI355X (gfx950) implementation of
 * the T2ONet executor/operator hot path.
 *
 * Drop-in boundary.  The reference has no FFI; the interface these entry points
 * replace is the Python one, cited per function (paths relative to the reference):
 *     Executor.execute            executors/executor.py:33-55
 *     Operator.execute / process  models/operators.py:112-131 and the process()
 *                                 bodies :240-245 :277-283 :351-358 :473-479 :509-511
 *                                 :571-585 :607-616
 *     L1 step                     experiments/t2onet/train_seq2seqL1.py:78-85
 *     Attention.forward           models/attention.py:37-40 (score / softmax / mix)
 *     BatchNorm2d + add + ReLU    models/actor_resnet.py:38-44, :99-100 (training mode)
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - all tensors are fp32, contiguous, DEVICE pointers: images (B,3,H,W) NCHW RGB in
 *     [0,1]; params (B,param_stride) rows; masks (B,mask_ch,H,W), mask_ch in {1,3};
 *   - `stream` is a hipStream_t passed as void*; every call only enqueues work on it:
 *     no allocation, no host synchronisation, safe to capture in a hipGraph;
 *   - caller-owned scratch: `workspace` of at least t2o_workspace_bytes(B,H,W) bytes;
 *   - return value: 0 = ok, otherwise a T2O_E* code; t2o_last_error() gives the text
 *     (thread-local).  Nothing throws.
 *
 * Operator indices (executors/executor.py:30):
 *   0 brightness 1 contrast 2 saturation 3 color-curve 4 inpaint(unsupported)
 *   5 tone-curve 6 sharpness 7 white;  -1 = identity (executor.py:44-46).
 */
#ifndef T2ONET_HIP_H
#define T2ONET_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define T2O_OK 0
#define T2O_EINVAL 1        /* bad pointer / shape / stride */
#define T2O_EUNSUPPORTED 2  /* operator 4 (inpaint) or unknown operator index */
#define T2O_EWORKSPACE 3    /* workspace too small */
#define T2O_ELAUNCH 4       /* HIP reported a launch error */

#define T2O_OP_IDENTITY (-1)
#define T2O_MAX_PARAM 24

int t2o_abi_version(void);
const char* t2o_last_error(void);
/* sha256 (hex) of the sources + compiler flags this binary was built from (compiled in by
 * t2onet_amd/build.py); the loader refuses a library whose digest differs from the tree's. */
const char* t2o_source_digest(void);

/* number of parameters of operator `op` (Executor.get_param_num, executor.py:61-63); -1 if unknown */
int t2o_op_num_params(int op);

/* bytes of scratch needed by the *_bwd / *_l1 / sequence calls for images of this shape */
size_t t2o_workspace_bytes(int B, int H, int W);

/* ---- one operator over a whole (sub)batch: Operator.execute once `param` is known ----
 * out = clamp(process(img, param) * mask + img * (1 - mask), 0, 1)   (operators.py:128-130)
 * mask may be NULL (mask_ch = 0).  op = -1 copies img to out (no clamp). */
int t2o_op_fwd(int op, const float* img, const float* param, int param_stride,
               const float* mask, int mask_ch, float* out, int B, int H, int W, void* stream);

/* gradient of the above: gimg (nullable) and gparam (B,gparam_stride; first n columns written) */
int t2o_op_bwd(int op, const float* img, const float* param, int param_stride,
               const float* mask, int mask_ch, const float* gout,
               float* gimg, float* gparam, int gparam_stride,
               void* workspace, size_t workspace_bytes, int B, int H, int W, void* stream);

/* ---- per-sample operators in ONE launch: replaces Actor.divide_op_group + the per-group
 * execute + index_select regrouping (models/actor.py:100-114, :244-259).
 * op_id: (B) int32 DEVICE array; param rows padded to param_stride (24 in the actor). */
int t2o_apply_fwd(const int* op_id, const float* img, const float* param, int param_stride,
                  const float* mask, int mask_ch, float* out, int B, int H, int W, void* stream);
int t2o_apply_bwd(const int* op_id, const float* img, const float* param, int param_stride,
                  const float* mask, int mask_ch, const float* gout,
                  float* gimg, float* gparam, int gparam_stride,
                  void* workspace, size_t workspace_bytes, int B, int H, int W, void* stream);

/* ---- L1 loss, train_seq2seqL1.py:85: loss[0] = mean |pred - target| over n floats ---- */
int t2o_l1_fwd(const float* pred, const float* target, float* loss, size_t n,
               void* workspace, size_t workspace_bytes, void* stream);
/* gpred = sign(pred - target) * gloss[0] / n      (gloss: device scalar) */
int t2o_l1_bwd(const float* pred, const float* target, const float* gloss, float* gpred,
               size_t n, void* stream);

/* ---- END select fused with the L1 loss (train_seq2seqL1.py:78-85): pred[b] = imgs[first[b]][b], first[b] = the step of
 * sample b's first END token (else the last step), loss[0] = mean |pred - target| over B * row floats.  imgs / gimgs: HOST
 * arrays of T <= 8 device pointers to (B, row) step images / their gradients; first: device int64 (B).  The backward
 * writes all T gradients in one launch: sign(pred - target) * gloss[0] / (B row) where selected, zero elsewhere.
 * Replaces torch.stack + advanced indexing + t2o_l1_* (and their zero-filled scatter in the backward); the loss equals
 * t2o_l1_fwd on the gathered images bit for bit.  workspace: as t2o_l1_fwd for n = B * row. */
int t2o_end_select_l1_fwd(const float* const* imgs, int T, const long long* first, const float* target, float* loss, int B,
                          size_t row, void* workspace, size_t workspace_bytes, void* stream);
int t2o_end_select_l1_bwd(const float* const* imgs, float* const* gimgs, int T, const long long* first, const float* target,
                          const float* gloss, int B, size_t row, void* stream);

/* ---- operator fused with the L1 loss on its output (last operator of a sequence) ----
 * forward also reads `target` and writes loss[0] = mean |out - target|;
 * backward takes the target instead of gout: gout = sign(out - target) * gloss[0] / (B*3*H*W). */
int t2o_op_fwd_l1(int op, const float* img, const float* param, int param_stride,
                  const float* mask, int mask_ch, const float* target, float* out, float* loss,
                  void* workspace, size_t workspace_bytes, int B, int H, int W, void* stream);
int t2o_op_bwd_l1(int op, const float* img, const float* param, int param_stride,
                  const float* mask, int mask_ch, const float* target, const float* gloss,
                  float* gimg, float* gparam, int gparam_stride,
                  void* workspace, size_t workspace_bytes, int B, int H, int W, void* stream);

/* ---- a known operator sequence (executor benchmark / planner: utils/beam_search.py:79) ----
 * ops: K HOST ints; params: (K,B,24) device; acts: (K,B,3,H,W) device, acts[k] = output of
 * operator k (every intermediate image is materialised, as Executor.execute returns it);
 * loss[0] = mean |acts[K-1] - target|.  One host call, 2K+2 kernel launches, no sync. */
int t2o_sequence_fwd(const int* ops, int K, const float* img, const float* params,
                     const float* target, float* acts, float* loss,
                     void* workspace, size_t workspace_bytes, int B, int H, int W, void* stream);
/* gimg (nullable), gparams (K,B,24); gbuf: scratch of 2 images (2,B,3,H,W) */
int t2o_sequence_bwd(const int* ops, int K, const float* img, const float* params,
                     const float* target, const float* acts, const float* gloss,
                     float* gimg, float* gparams, float* gbuf,
                     void* workspace, size_t workspace_bytes, int B, int H, int W, void* stream);

/* ---- the same, FUSED: runs of pointwise operators execute in registers in one kernel pair;
 * only the image at segment boundaries (before/after each sharpness, every 8 operators) goes
 * through HBM.  out (B,3,H,W) = final image.  seg_bufs: t2o_fused_sequence_buffers(ops,K) images
 * of scratch (B,3,H,W each), written by fwd and read by bwd.  Same results as t2o_sequence_*
 * (bit-identical images; parameter gradients equal up to summation order). */
int t2o_fused_sequence_buffers(const int* ops, int K);
int t2o_fused_sequence_fwd(const int* ops, int K, const float* img, const float* params,
                           const float* target, float* out, float* loss, float* seg_bufs,
                           void* workspace, size_t workspace_bytes, int B, int H, int W, void* stream);
/* backward: either (target, gloss) -- fused L1, gout = NULL -- or target = NULL and gout =
 * gradient w.r.t. the final image.  gbuf: scratch of 2 images. */
int t2o_fused_sequence_bwd(const int* ops, int K, const float* img, const float* params,
                           const float* target, const float* gloss, const float* gout,
                           float* gimg, float* gparams, const float* seg_bufs, float* gbuf,
                           void* workspace, size_t workspace_bytes, int B, int H, int W, void* stream);

/* value AND gradient of loss = mean|sequence(img) - target| in one call (what one train iteration of the planner's and
 * the L1 trainer's inner loop needs: Executor.execute K times, executors/executor.py:33-55, then L1Loss + backward,
 * experiments/t2onet/train_seq2seqL1.py:82-86, utils/beam_search.py:65-91).  The forward of the LAST segment is not
 * launched: its backward kernels recompute the final pixel anyway (for sign(out - target)) and also emit |out - target|
 * partials and, when out != NULL, the final image.  BASELINE configs[1] (5 per-pixel operators + sharpness): 3 launches
 * instead of 4; a list without sharpness: ONE launch.  gimg, gparams and out are bit-identical to
 * t2o_fused_sequence_fwd(target) + t2o_fused_sequence_bwd(target, gloss) (the same kernels in the same geometry); loss
 * is the same sum in another order (partials per backward workgroup), equal to ~1e-7 relative.  out may be NULL;
 * seg_bufs / gbuf as for the two-call form.  Image sizes whose sharpness runs on the LDS-tile kernels (W % 4 != 0 ...)
 * fall back to the two calls internally (then out is required when the list is a single sharpness segment). */
int t2o_fused_sequence_l1_value_grad(const int* ops, int K, const float* img, const float* params,
                                     const float* target, const float* gloss, float* out, float* loss,
                                     float* gimg, float* gparams, float* seg_bufs, float* gbuf,
                                     void* workspace, size_t workspace_bytes, int B, int H, int W, void* stream);

/* ---- run-time specialisation of the fused chain kernels for an arbitrary operator list (t2o_jit.hip).
 * The fastest kernels behind t2o_fused_sequence_fwd/bwd take the operator list as a compile-time constant; ahead of time
 * only BASELINE.json's two lists are instantiated.  t2o_fused_sequence_prepare(ops, K) compiles (hipRTC, from the headers
 * embedded in this library, same flags as the ahead-of-time build) and loads the kernels for every per-pixel segment of
 * this list that has none; later t2o_fused_sequence_* calls with the same list pick them up.  Results are bit-identical
 * to the run-time-loop kernels (same arithmetic per pixel; parameter gradients up to summation order).  Blocking (seconds
 * per new list), HOST side: call it outside any stream capture.  T2O_EUNSUPPORTED when libhiprtc is not on the machine
 * (the sequence calls then keep using the run-time-loop kernels).  t2o_jit_set_cache_dir: directory where compiled code
 * objects are kept (keyed by source digest + operator list), NULL/"" = no disk cache.  Executor.execute permits any order
 * (executors/executor.py:33-55); the planner enumerates them (utils/beam_search.py:218-231). */
int t2o_fused_sequence_prepare(const int* ops, int K);
int t2o_jit_set_cache_dir(const char* dir);
int t2o_jit_specialisations(void);      /* operator lists specialised so far in this process */

/* ---- planner candidate sweep (utils/beam_search.py:65-91: one executor call + .item() per
 * evaluated parameter): loss[c] = mean |clamp(process(img, params[c])) - target| for C candidate
 * parameter rows of ONE per-pixel operator (0,1,2,3,5,7) against ONE image pair (3,H,W), in a
 * single launch; the image is read once per 8 candidates. */
size_t t2o_candidates_workspace_bytes(int C, int H, int W);
int t2o_op_candidates_l1(int op, const float* img, const float* target, const float* params, int C,
                         int param_stride, float* loss, void* workspace, size_t workspace_bytes,
                         int H, int W, void* stream);

/* The same for up to 64 (image, operator) jobs in ONE launch -- a whole beam-search step of the planner
 * (utils/beam_search.py:218-231: every beam image x every operation): job j evaluates its C candidate rows
 * params[j] of operator ops[j] on image imgs[img_index[j]] against the common target.  ops / img_index are HOST
 * arrays (copied into the kernel arguments); loss is (J, C). */
size_t t2o_candidates_multi_workspace_bytes(int J, int C, int H, int W);
int t2o_op_candidates_multi_l1(const int* ops, const int* img_index, int J, const float* imgs, int n_img,
                               const float* target, const float* params, int C, int param_stride, float* loss,
                               void* workspace, size_t workspace_bytes, int H, int W, void* stream);

/* ---- SSIM: utils/ssim/__init__.py:20-40 (forward: the evaluation metric; backward: its closed-form gradient) ----
 * 11x11 Gaussian window (sigma 1.5), zero padding, C1 = 1e-4, C2 = 9e-4.
 * out[b] = mean over (C,H,W) of the SSIM map of sample b (size_average=False of the reference;
 * its size_average=True is the mean of out).  workspace: t2o_ssim_workspace_bytes(B,C,H,W). */
size_t t2o_ssim_workspace_bytes(int B, int C, int H, int W);
int t2o_ssim_fwd(const float* img1, const float* img2, float* out,
                 void* workspace, size_t workspace_bytes, int B, int C, int H, int W, void* stream);
/* Backward of t2o_ssim_fwd (the reference differentiates utils/ssim/__init__.py:20-40 by autograd; there is no reference
 * kernel): gout (B) = gradient w.r.t. out[b]; g1 / g2 (B,C,H,W) = gradients w.r.t. img1 / img2, either may be NULL.
 * One launch, one workgroup per 32 x 32 tile of a plane; the four derivative maps live in LDS only.  Writes (does not
 * accumulate); no workspace; deterministic. */
int t2o_ssim_bwd(const float* img1, const float* img2, const float* gout, float* g1, float* g2,
                 int B, int C, int H, int W, void* stream);

/* ---- dot-product attention core, models/attention.py:37-40 ----
 * q (B,D), ctx (B,L,D) -> attn (B,L) = softmax_l(q . ctx_l) over ALL L rows (no padding
 * mask, as the reference), mix (B,D) = sum_l attn_l ctx_l.   D % 64 == 0, D <= 1024, L <= 64 */
int t2o_attn_fwd(const float* q, const float* ctx, float* attn, float* mix,
                 int B, int L, int D, void* stream);
/* gmix (B,D), gattn (B,L) nullable -> gq (B,D), gctx (B,L,D) */
int t2o_attn_bwd(const float* q, const float* ctx, const float* attn, const float* gmix,
                 const float* gattn, float* gq, float* gctx, int B, int L, int D, void* stream);

/* ---- choice of the next operator in the free-running decode, models/actor.py:222-236 (one launch for: exp, explore
 * mix, op-mask, renormalisation, Categorical draw or arg-max, op-mask update).  logp (B,n) log-probabilities, op_mask
 * (B,n) in/out (the chosen entry is cleared), u (B) uniform [0,1) numbers or NULL for arg-max, pred_op (B) int64
 * operator-vocabulary ids, exec_op (B) int32 = pred_op - 3 (the executor index t2o_apply_* takes; negative = identity). */
int t2o_choose_op(const float* logp, float* op_mask, const float* u, float explore_prob, long long* pred_op, int* exec_op,
                  int B, int n_cls, void* stream);

/* ---- training-mode BatchNorm2d fused with the residual add and ReLU that follow it in the image
 * encoder: models/actor_resnet.py:38-44 (BasicBlock.forward: relu(bn1(conv1(x))), relu(bn2(conv2(.)) +
 * shortcut(x))) and :99-100 (stem).  x, res, out: (N,C,H*W) contiguous NCHW; HW = H*W.
 *   out = max((x - mean_c) * invstd_c * weight_c + bias_c (+ res), 0), batch statistics per channel
 *   (biased variance), running_mean / running_var (nullable, both or neither) updated in place with
 *   `momentum` and the unbiased variance, as torch.nn.BatchNorm2d in training mode.
 * save_mean / save_invstd (C) are written for the backward.  workspace: t2o_bn_workspace_bytes(N, C). */
size_t t2o_bn_workspace_bytes(int N, int C);
int t2o_bn_relu_fwd(const float* x, const float* res, const float* weight, const float* bias,
                    float* running_mean, float* running_var, float* save_mean, float* save_invstd, float* out,
                    float momentum, float eps, void* workspace, size_t workspace_bytes,
                    int N, int C, int HW, void* stream);
/* dy = gradient w.r.t. out.  has_res = 1: a residual was added (y = the forward's out is needed for the ReLU
 * mask; dres (nullable) receives the residual's gradient); has_res = 0: the mask is recomputed from x, y may
 * be NULL.  dx (N,C,HW), dweight / dbias (C, nullable). */
int t2o_bn_relu_bwd(const float* x, const float* y, const float* dy, const float* weight, const float* bias,
                    const float* save_mean, const float* save_invstd, float* dx, float* dres,
                    float* dweight, float* dbias, int has_res, void* workspace, size_t workspace_bytes,
                    int N, int C, int HW, void* stream);

/* The same pair on channels-last activations: x, res, out, y, dy, dx, dres are (M, C) with M = N*H*W rows of C
 * contiguous channels (torch.channels_last storage of an (N,C,H,W) tensor), the layout the convolutions run in
 * natively on MI355X.  C must be a power of two in [4, 1024].  Same arithmetic, same running-statistics update.
 * relu = 0: plain batch norm without the activation (the shortcut branch, models/actor_resnet.py:33-36).
 * workspace: t2o_bn_nhwc_workspace_bytes(M, C). */
size_t t2o_bn_nhwc_workspace_bytes(int M, int C);
int t2o_bn_relu_nhwc_fwd(const float* x, const float* res, const float* weight, const float* bias,
                         float* running_mean, float* running_var, float* save_mean, float* save_invstd, float* out,
                         float momentum, float eps, int relu, void* workspace, size_t workspace_bytes,
                         int M, int C, void* stream);
/* t2o_bn_relu_nhwc_fwd with the batch statistics taken from per-tile partial sums a producer already holds
 * (t2o_conv3x3_fwd_stats_nhwc: the convolution's accumulators) instead of a pass over x: partial is
 * (partial_rows, 2, C) fp32 -- per row the channels' sums, then their sums of squares -- combined in fp64 in row order.
 * Everything else as t2o_bn_relu_nhwc_fwd. */
int t2o_bn_relu_nhwc_fwd_partials(const float* x, const float* res, const float* weight, const float* bias, float* running_mean,
                                  float* running_var, float* save_mean, float* save_invstd, float* out, float momentum,
                                  float eps, int relu, const float* partial, int partial_rows, void* workspace,
                                  size_t workspace_bytes, int M, int C, void* stream);
int t2o_bn_relu_nhwc_bwd(const float* x, const float* y, const float* dy, const float* weight, const float* bias,
                         const float* save_mean, const float* save_invstd, float* dx, float* dres,
                         float* dweight, float* dbias, int has_res, int relu, void* workspace,
                         size_t workspace_bytes, int M, int C, void* stream);

/* ---- 3x3 stride-1 padding-1 convolution of the image encoder, weight gradient (models/actor_resnet.py:27-44,
 * the BasicBlock convolutions) on the fp32 matrix cores (t2o_conv.hip).  Replaces the library's weight-gradient call
 * (torch.ops.aten.convolution_backward, output_mask [0,1,0]) for these layers.
 *   x  (N,H,W,Ci) and dy (N,H,W,Co): NHWC = torch.channels_last storage of (N,C,H,W) tensors
 *   dw (Co,3,3,Ci)                 = channels_last storage of a (Co,Ci,3,3) weight gradient
 *   dw[co][kh][kw][ci] = sum_{n,h,w} dy[n][h][w][co] * x[n][h+kh-1][w+kw-1][ci]   (zero padding)
 * Ci and Co must be multiples of 64, W a multiple of 4.  Deterministic: split-K partial sums in `workspace`
 * (t2o_conv3x3_wgrad_workspace_bytes) are added in a fixed order. */
size_t t2o_conv3x3_wgrad_workspace_bytes(int N, int H, int W, int Ci, int Co);
int t2o_conv3x3_wgrad_nhwc(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                           int N, int H, int W, int Ci, int Co, void* stream);

/* ---- the same convolutions, forward and data gradient (t2o_conv.hip: one implicit-GEMM kernel on the fp32 matrix
 * cores; the data gradient runs it on dy with the transposed, tap-mirrored weight).  Replace F.conv2d(x, w, None, 1, 1)
 * (models/actor_resnet.py:27-30 conv3x3) and torch.ops.aten.convolution_backward(..., output_mask [1,0,0]).
 *   y[n][h][w][co]  = sum_{kh,kw,ci} x[n][h+kh-1][w+kw-1][ci] * w[co][kh][kw][ci]        (zero padding)
 *   dx[n][h][w][ci] = sum_{kh,kw,co} dy[n][h-kh+1][w-kw+1][co] * w[co][kh][kw][ci]
 * Tensors as above (NHWC activations, (Co,3,3,Ci) weight).  Forward: Ci % 32 == 0, Co % 64 == 0; data gradient:
 * Co % 32 == 0, Ci % 64 == 0; both: W % 8 == 0.  `workspace` holds a zero region (padding source of the LDS-DMA)
 * and, for the data gradient, the transformed weight; it is (re)written by every call. */
size_t t2o_conv3x3_fwd_workspace_bytes(int N, int H, int W, int Ci, int Co);
int t2o_conv3x3_fwd_nhwc(const float* x, const float* w, float* y, void* workspace, size_t workspace_bytes,
                         int N, int H, int W, int Ci, int Co, void* stream);
size_t t2o_conv3x3_dgrad_workspace_bytes(int N, int H, int W, int Ci, int Co);
int t2o_conv3x3_dgrad_nhwc(const float* dy, const float* w, float* dx, void* workspace, size_t workspace_bytes,
                           int N, int H, int W, int Ci, int Co, void* stream);

/* ---- forward of the stride-2 3x3 convolutions (F.conv2d(x, w, None, 2, 1), models/actor_resnet.py:32-36 with stride 2):
 *   y[n][a][b][co] = sum_{kh,kw,ci} x[n][2a+kh-1][2b+kw-1][ci] * w[co][kh][kw][ci]      (zero padding)
 *   x (N,2Ho,2Wo,Ci), w (Co,3,3,Ci), y (N,Ho,Wo,Co).  Ci % 32 == 0, Co % 64 == 0, Wo % 8 == 0. */
size_t t2o_conv3x3s2_fwd_workspace_bytes(int N, int Ho, int Wo, int Ci, int Co);
int t2o_conv3x3s2_fwd_nhwc(const float* x, const float* w, float* y, void* workspace, size_t workspace_bytes,
                           int N, int Ho, int Wo, int Ci, int Co, void* stream);

/* The forward convolution (stride 1 or 2, shapes as t2o_conv3x3_fwd_nhwc / t2o_conv3x3s2_fwd_nhwc; Ho, Wo = output
 * grid) that also leaves the batch-norm statistics of its output: stats is (rows, 2, Co) fp32 with rows =
 * t2o_conv3x3_fwd_stats_rows(...) -- per pixel tile the per-channel sum and sum of squares of y, summed in a fixed
 * order from the accumulators -- the input of t2o_bn_relu_nhwc_fwd_partials.  Replaces conv2d + the statistics half
 * of BatchNorm2d (models/actor_resnet.py:38-44: bn(conv(x))): the activation is not read back for its mean. */
int t2o_conv3x3_fwd_stats_rows(int N, int Ho, int Wo, int Co, int stride);
int t2o_conv3x3_fwd_stats_nhwc(const float* x, const float* w, float* y, float* stats, void* workspace, size_t workspace_bytes,
                               int N, int Ho, int Wo, int Ci, int Co, int stride, void* stream);

/* Forward of the image encoder's stem, conv2d(x, w, None, stride 2, padding 1) with 3 input and 32 / 64 output channels
 * (models/actor_resnet.py:99): x (N, 2Ho, 2Wo, 3), w (Co,3,3,3) channels-last, y (N, Ho, Wo, Co).  stats: null, or
 * (t2o_stem_fwd_stats_rows(N, Ho, Wo), 2, Co) per-workgroup channel sums / sums of squares of y for
 * t2o_bn_relu_nhwc_fwd_partials (the stem's batch norm then makes no statistics pass over its 268 MB at bs=64). */
int t2o_stem_fwd_stats_rows(int N, int Ho, int Wo);
int t2o_stem_fwd_nhwc(const float* x, const float* w, float* y, float* stats, int N, int Ho, int Wo, int Co, void* stream);

/* Weight gradient of the same stem convolution: x (N, 2Ho, 2Wo, 3), dy (N, Ho, Wo, Co) -> dw (Co,3,3,3) channels-last.
 * Deterministic (per-workgroup partial blocks in the workspace, added in order), unlike the atomic kernel the library
 * picks for this layer.  Workspace: t2o_stem_wgrad_workspace_bytes. */
size_t t2o_stem_wgrad_workspace_bytes(int N, int Ho, int Wo, int Co);
int t2o_stem_wgrad_nhwc(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes, int N, int Ho, int Wo,
                        int Co, void* stream);

/* ---- weight gradient of the same stride-2 convolutions (t2o_conv.hip, the stride-1 kernel with a two-plane x tile):
 *   dw[co][kh][kw][ci] = sum_{n,a,b} dy[n][a][b][co] * x[n][2a+kh-1][2b+kw-1][ci]   (zero padding)
 *   x (N,2Ho,2Wo,Ci), dy (N,Ho,Wo,Co), dw (Co,3,3,Ci).  Ci, Co multiples of 64, Wo a multiple of 4.  Deterministic
 *   (fixed-order split-K), replaces convolution_backward(..., stride 2, output_mask [0,1,0]). */
size_t t2o_conv3x3s2_wgrad_workspace_bytes(int N, int Ho, int Wo, int Ci, int Co);
int t2o_conv3x3s2_wgrad_nhwc(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                             int N, int Ho, int Wo, int Ci, int Co, void* stream);

/* Optional: register `bytes` of device memory on `device` that hold zeros and that nobody writes while convolutions
 * run (NULL unregisters).  The t2o_conv3x3* calls then read their padding from it instead of clearing the zero
 * region at the start of their workspace on every call (it must be at least that large, 20 KiB covers the
 * encoder; smaller: ignored).  Process-wide state, set before the calls it should affect. */
int t2o_conv_set_zero_region(int device, const void* zeros, size_t bytes);

/* ---- data gradient of the encoder's STRIDE-2 3x3 convolutions (the first convolution of each stage,
 * models/actor_resnet.py:32-36 with stride 2; replaces torch.ops.aten.convolution_backward(..., output_mask [1,0,0])):
 *   dx[n][i][j][ci] = sum_{kh,kw,co} dy[n][(i+1-kh)/2][(j+1-kw)/2][co] * w[co][kh][kw][ci]   over the taps for which
 *   both quotients are integers inside the (Ho, Wo) grid.   dy (N,Ho,Wo,Co), w (Co,3,3,Ci), dx (N,2Ho,2Wo,Ci), NHWC.
 * Co % 32 == 0, Ci % 64 == 0, Wo % 8 == 0.  `workspace`: zero region + the transposed weight, rewritten per call. */
size_t t2o_conv3x3s2_dgrad_workspace_bytes(int N, int Ho, int Wo, int Ci, int Co);
int t2o_conv3x3s2_dgrad_nhwc(const float* dy, const float* w, float* dx, void* workspace, size_t workspace_bytes,
                             int N, int Ho, int Wo, int Ci, int Co, void* stream);

/* ---- operator parameter heads for a batch whose samples use different operators: models/operators.py:73-88
 * (param = op_param_regressor(fc2(LeakyReLU_0.01(fc1(features))))) as called per operator group by
 * models/actor.py:244-255.  op_id (B) device int32: executor index of each sample, < 0 or 4 -> no head, zeros.
 * w1/b1/w2/b2: HOST arrays of 8 device pointers indexed by executor index (entry 4 may be NULL): fc1 (512,512) /
 * (512), fc2 (n,512) / (n), n = 1, 8 (tone) or 24 (color).  ctx (B,512) -> param (B,24) zero padded; hidden (B,512)
 * and raw (B,24) are written for the backward.  Regressor constants: brightness_range, saturation_range (lo, hi),
 * sharpness_range of the options.  D must be 512. */
int t2o_param_heads_fwd(const int* op_id, const float* ctx, const float* const* w1, const float* const* b1,
                        const float* const* w2, const float* const* b2, float* hidden, float* raw, float* param,
                        float brightness_range, float sat_lo, float sat_hi, float sharpness_range, int B, int D,
                        void* stream);
/* gparam (B,24) -> gctx (B,512) and DENSE gradients of every head (gw1 .. gb2: HOST arrays of 8 device pointers;
 * heads no sample selected receive zeros).  dpre (B,512): scratch.  Deterministic (batch-order sums). */
int t2o_param_heads_bwd(const int* op_id, const float* ctx, const float* const* w1, const float* const* b1,
                        const float* const* w2, const float* const* b2, const float* hidden, const float* raw,
                        const float* gparam, float* gctx, float* dpre, float* const* gw1, float* const* gb1,
                        float* const* gw2, float* const* gb2, float brightness_range, float sat_lo, float sat_hi,
                        float sharpness_range, int B, int D, void* stream);
/* The same with accumulate != 0: the heads' weight gradients are ADDED to gw1 / gb1 / gw2 / gb2 (every element has
 * exactly one writer: still deterministic).  For a trainer whose gradient buffers are persistent and zeroed before
 * the backward: the 28 gradient tensors of a decoder step then need no accumulation launches of their own
 * (140 per episode step under autograd). */
int t2o_param_heads_bwd_acc(const int* op_id, const float* ctx, const float* const* w1, const float* const* b1,
                        const float* const* w2, const float* const* b2, const float* hidden, const float* raw,
                        const float* gparam, float* gctx, float* dpre, float* const* gw1, float* const* gb1,
                        float* const* gw2, float* const* gb2, float brightness_range, float sat_lo, float sat_hi,
                        float sharpness_range, int B, int D, int accumulate, void* stream);

/* ---- round 3: the forms the encoder's explicit forward/backward schedule (t2onet_amd/encoder.py) calls ----
 * Gradient buffers of a trainer are persistent and zeroed once per step, so the weight-gradient kernels can ADD their
 * result (accumulate != 0; every element has one writer: still deterministic) instead of handing a fresh tensor to an
 * accumulation launch; transformed weights are made once per optimiser step, not once per data-gradient call. */

/* t2o_bn_relu_nhwc_bwd with accumulate: dweight / dbias += instead of = (models/actor_resnet.py:38-44 backward). */
int t2o_bn_relu_nhwc_bwd_acc(const float* x, const float* y, const float* dy, const float* weight, const float* bias,
                             const float* save_mean, const float* save_invstd, float* dx, float* dres, float* dweight,
                             float* dbias, int has_res, int relu, int accumulate, void* workspace, size_t workspace_bytes,
                             int M, int C, void* stream);
/* ... with the sums supplied by the producer of dy (see t2o_conv3x3_dgrad_pre_bnsums_nhwc): finalize + apply only; no residual. */
int t2o_bn_relu_nhwc_bwd_partials_acc(const float* x, const float* dy, const float* weight, const float* bias,
                                      const float* save_mean, const float* save_invstd, float* dx, float* dweight,
                                      float* dbias, int relu, int accumulate, const float* partial, int partial_rows,
                                      void* workspace, size_t workspace_bytes, int M, int C, void* stream);

/* Weight gradient of a 3x3 convolution, stride 1 or 2 ((N,Ho,Wo) = the dy grid; shapes / workspace as
 * t2o_conv3x3_wgrad_nhwc / t2o_conv3x3s2_wgrad_nhwc); accumulate != 0: dw += . */
int t2o_conv3x3_wgrad_acc_nhwc(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                               int N, int Ho, int Wo, int Ci, int Co, int stride, int accumulate, void* stream);

/* wt[ci][t'][co] = w[co][t][ci] for a (Co, taps, Ci) weight, t' = taps-1-t when flip != 0 else t.  taps = 9, flip = 1: the
 * weight the stride-1 data gradient runs the forward kernel with; taps = 9, flip = 0: the stride-2 data gradient's;
 * taps = 1: the transposed 1x1 shortcut weight.  Co, Ci multiples of 32. */
int t2o_conv_weight_transform(const float* w, float* wt, int Co, int Ci, int taps, int flip, void* stream);
/* n <= 32 of them in one launch (HOST arrays of length n): an encoder's 20 weights once per optimiser step */
int t2o_conv_weight_transform_batch(const float* const* w, float* const* wt, const int* Co, const int* Ci, const int* taps,
                                    const int* flip, int n, void* stream);

/* t2o_conv3x3_dgrad_nhwc with the transformed weight supplied (wt from t2o_conv_weight_transform(w, wt, Co, Ci, 9, 1))
 * and an optional addend (N,H,W,Ci) added to dx in the kernel's epilogue: the gradient a BasicBlock's input receives
 * through the identity shortcut (models/actor_resnet.py:43 `out += self.shortcut(x)`), so no separate add pass.
 * workspace: t2o_conv3x3_fwd_workspace_bytes(N, H, W, Co, Ci) (the zero region only). */
int t2o_conv3x3_dgrad_pre_nhwc(const float* dy, const float* wt, const float* addend, float* dx, void* workspace,
                               size_t workspace_bytes, int N, int H, int W, int Ci, int Co, void* stream);
/* The same data gradient (no addend) in front of y = relu(bn(bn_x)) -- a BasicBlock's second convolution, models/actor_resnet.py:
 * 38-44: the epilogue ALSO leaves the batch norm's backward sums, rows (t2o_conv3x3_dgrad_bnsums_rows(...), 2, Ci): per pixel
 * tile the channels' sums of g = dx * [bn_x * gamma * invstd + (beta - mean * gamma * invstd) > 0] and of g * xhat, the gate
 * evaluated exactly as t2o_bn_relu_nhwc_bwd does.  Feed them to t2o_bn_relu_nhwc_bwd_partials_acc: that batch norm's own
 * sums pass (one read of dx and one of bn_x) is not launched; bn_x is fetched under the last loop iteration's MFMAs. */
int t2o_conv3x3_dgrad_bnsums_rows(int N, int H, int W, int Ci, int Co);
int t2o_conv3x3_dgrad_pre_bnsums_nhwc(const float* dy, const float* wt, float* dx, const float* bn_x,
                                      const float* save_mean, const float* save_invstd, const float* weight,
                                      const float* bias, float* rows, void* workspace, size_t workspace_bytes,
                                      int N, int H, int W, int Ci, int Co, void* stream);
/* t2o_conv3x3s2_dgrad_nhwc with wt = t2o_conv_weight_transform(w, wt, Co, Ci, 9, 0) supplied. */
int t2o_conv3x3s2_dgrad_pre_nhwc(const float* dy, const float* wt, float* dx, void* workspace, size_t workspace_bytes,
                                 int N, int Ho, int Wo, int Ci, int Co, void* stream);

/* The stem (models/actor_resnet.py:99, conv 3 -> 32 / 64, stride 2) reading / writing the image in its own layout:
 * planar != 0: x / dx are (N,3,2Ho,2Wo) NCHW (what the operators produce: no channels-last copy of the image per encoder
 * call); planar == 0: (N,2Ho,2Wo,3).  y / dy are (N,Ho,Wo,Co) NHWC, w (Co,3,3,3) channels-last.  accumulate != 0: the
 * weight gradient is added to dw, the data gradient to dx (the image gradient already holds the operator's part). */
int t2o_stem_fwd(const float* x, const float* w, float* y, float* stats, int N, int Ho, int Wo, int Co, int planar, void* stream);
/* forward for an input of ANY size (Hi, Wi; Ho = (Hi+1)/2): full-resolution inference images have odd sizes */
int t2o_stem_fwd_any(const float* x, const float* w, float* y, float* stats, int N, int Hi, int Wi, int Co, int planar, void* stream);
int t2o_stem_wgrad(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes, int N, int Ho, int Wo,
                   int Co, int planar, int accumulate, void* stream);
int t2o_stem_dgrad(const float* dy, const float* w, float* dx, int N, int Ho, int Wo, int Co, int planar, int accumulate, void* stream);

/* ---- the 1x1 stride-2 shortcut convolutions of the encoder's stages (models/actor_resnet.py:33-36: nn.Conv2d(in, out,
 * kernel_size=1, stride=stride, bias=False)), fp32 matrix cores (t2o_conv1x1.hip).  Ho = (H+1)/2, Wo = (W+1)/2.
 *   forward        y[n][a][b][co]    = sum_ci x[n][2a][2b][ci] * w[co][ci]                x (N,H,W,Ci), w (Co,Ci), y (N,Ho,Wo,Co)
 *   data gradient  dx[n][2a][2b][ci] += sum_co dy[n][a][b][co] * w[co][ci]                ADDS into an existing dx (the 3x3
 *                  branch's data gradient, which covers every pixel); wt (Ci,Co) = t2o_conv_weight_transform(w, wt, Co, Ci, 1, 0)
 *   weight gradient dw[co][ci]       (+)= sum_{n,a,b} dy[n][a][b][co] * x[n][2a][2b][ci]   deterministic split-K
 * Ci, Co multiples of 64.  Replace F.conv2d(x, w, None, 2) and both halves of convolution_backward for these layers
 * (the library's kernels add atomically and need their output cleared first). */
int t2o_conv1x1s2_fwd_nhwc(const float* x, const float* w, float* y, int N, int H, int W, int Ci, int Co, void* stream);
int t2o_conv1x1s2_dgrad_acc_nhwc(const float* dy, const float* wt, float* dx, int N, int H, int W, int Ci, int Co, void* stream);
size_t t2o_conv1x1s2_wgrad_workspace_bytes(int N, int H, int W, int Ci, int Co);
int t2o_conv1x1s2_wgrad_nhwc(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                             int N, int H, int W, int Ci, int Co, int accumulate, void* stream);

/* ---- the same 3x3 convolutions (padding 1, stride 1 or 2) for ANY image size (t2o_conv_generic.hip): gathered-row
 * implicit GEMMs on the fp32 matrix cores, about half the rate of the LDS-DMA kernels above, for the layers those
 * cannot take (image width not a multiple of 8 / 4, odd sizes under stride 2: the 4 x 4 stage of a 128 x 128 training
 * image, every stage of a full-resolution inference image) -- replaces the library fall-back there, deterministic.
 *   forward        x (N,H,W,Ci), w (Co,3,3,Ci) -> y (N,Ho,Wo,Co), Ho = (H-1)/stride + 1
 *   data gradient  dy (N,Ho,Wo,Co), wt = t2o_conv_weight_transform(w, wt, Co, Ci, 9, stride == 1) -> dx (N,H,W,Ci)
 *                  (+ addend (N,H,W,Ci), nullable)
 *   weight gradient x, dy -> dw (Co,3,3,Ci), split-K partials in `workspace`, accumulate != 0: dw +=
 * Ci % 32 == 0 (weight gradient: % 64), Co % 64 == 0 (data gradient: Co % 32, Ci % 64). */
int t2o_conv3x3_any_fwd_nhwc(const float* x, const float* w, float* y, int N, int H, int W, int Ci, int Co, int stride, void* stream);
int t2o_conv3x3_any_dgrad_nhwc(const float* dy, const float* wt, const float* addend, float* dx, int N, int H, int W, int Ci, int Co,
                               int stride, void* stream);
size_t t2o_conv3x3_any_wgrad_workspace_bytes(int N, int H, int W, int Ci, int Co, int stride);
int t2o_conv3x3_any_wgrad_nhwc(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                               int N, int H, int W, int Ci, int Co, int stride, int accumulate, void* stream);

/* ---- Winograd F(2x2, 3x3) for the stride-1 3x3 convolutions of the deep encoder stages (t2o_winograd.hip transforms,
 * t2o_conv.hip GEMMs; models/actor_resnet.py:24-44, the 256- and 512-channel BasicBlocks).  y = conv2d(x, w, None, 1, 1),
 * NHWC, H and W even, T = N * H/2 * W/2 output tiles, Tpad = t2o_wino_padded_tiles(N,H,W) (T rounded up to 256):
 *   U (16, Co, Ci)   = t2o_wino_weight_transform(w (Co,3,3,Ci), Cn = Co, Ck = Ci)                once per weight update
 *   V (16, Tpad, Ci) = t2o_wino_input_transform(x (N,H,W,Ci))                                     rows T.. are zero
 *   M (16, T, Co)    = t2o_gemm_nt_batched(V, U, M, 16, T, Co, Ci, Tpad)                          M[xi] = V[xi] U[xi]^T
 *   y (N,H,W,Co)     = t2o_wino_output_transform(M, addend, y, stats)
 * addend: null or (N,H,W,Co), added to y (the gradient through a block's identity shortcut, as in
 * t2o_conv3x3_dgrad_pre_nhwc); stats: null or (t2o_wino_stats_rows(N,H,W,Co), 2, Co) per-workgroup channel sums / sums
 * of squares of y for t2o_bn_relu_nhwc_fwd_partials (as t2o_conv3x3_fwd_stats_nhwc).  The data gradient is the same
 * pipeline on dy with U = t2o_wino_weight_transform(wt, Cn = Ci, Ck = Co), wt = t2o_conv_weight_transform(w, ., Co, Ci, 9, 1).
 * Weight gradient (in the transformed domain, deterministic):
 *   Ad (16, Tpad, Co)      = t2o_wino_dy_transform(dy (N,H,W,Co))                                 A dY A^T, rows T.. zero
 *   dU (splits, 16, Co, Ci) = t2o_gemm_tn_batched(Ad, V, dU, 16, Tpad, Co, Ci, splits)            dU[xi] = Ad[xi]^T V[xi]
 *                            splits = t2o_gemm_tn_splits(16, Tpad, Co, Ci): pieces of the tile range (fixed partition)
 *   dw (Co,3,3,Ci)  (+)=   t2o_wino_dw_transform(dU, dw, Co, Ci, splits, accumulate)              G^T (sum of the pieces) G
 * C: a power of two in [4, 1024] for the transforms; the GEMMs: t2o_gemm_nt_batched N % 64 == 0, K % 32 == 0;
 * t2o_gemm_tn_batched M, N % 128 == 0, rows % (64 * splits) == 0; all tensors 16-byte aligned. */
int t2o_wino_padded_tiles(int N, int H, int W);
int t2o_wino_weight_transform(const float* w, float* U, int Cn, int Ck, void* stream);
int t2o_wino_weight_transform_batch(const float* const* w, float* const* U, const int* Cn, const int* Ck, int n, void* stream);   /* n <= 32 banks, one launch */
/* ... straight into the chunk-major layout (Ck/8, 16, Cn, 8) of t2o_wino_fused_conv_nhwc (= t2o_wino_u_chunked of the above) */
int t2o_wino_weight_transform_chunked_batch(const float* const* w, float* const* Uc, const int* Cn, const int* Ck, int n, void* stream);
int t2o_wino_input_transform(const float* x, float* V, int N, int H, int W, int C, void* stream);
/* the same into rows [0, Tpad) of every plane of a LARGER (16, plane_rows, C) tensor whose first plane starts at V (a train
 * step's passes side by side: the weight gradient then runs once over all of them); plane_rows = 0: Tpad */
int t2o_wino_input_transform_ld(const float* x, float* V, int N, int H, int W, int C, int plane_rows, void* stream);
int t2o_wino_stats_rows(int N, int H, int W, int C);
int t2o_wino_output_transform(const float* M, const float* addend, float* y, float* stats, int N, int H, int W, int C, void* stream);
int t2o_wino_dy_transform(const float* dy, float* Ad, int N, int H, int W, int C, void* stream);
/* t2o_wino_input_transform(dy) and t2o_wino_dy_transform(dy) in one pass over dy (a layer's backward needs both) */
int t2o_wino_dy_transforms(const float* dy, float* V, float* Ad, int N, int H, int W, int C, void* stream);
int t2o_wino_dy_transforms_ld(const float* dy, float* V, float* Ad, int N, int H, int W, int C, int ad_plane_rows, void* stream);
/* A dY A^T alone into a row range of a larger (16, plane_rows, C) tensor (the data gradient then runs on t2o_wino_fused_conv_nhwc) */
int t2o_wino_dy_transform_ld(const float* dy, float* Ad, int N, int H, int W, int C, int plane_rows, void* stream);   /* Ad as a row range of a (16, ad_plane_rows, C) tensor */
int t2o_wino_dw_transform(const float* dU, float* dw, int Co, int Ci, int splits, int accumulate, void* stream);
/* the batched fp32 matrix-core GEMMs behind them (t2o_conv.hip k_gemm_nt / k_gemm_tn: the forward convolution's LDS-DMA
 * machinery without taps), dense row-major operands:
 *   nt: C[b] (M,N) = A[b] (a_rows >= M rows of K) * B[b] (N,K)^T          for b < batches
 *   tn: C[s][b] (M,N) = sum over the rows t of piece s of A[b] (rows,M)[t]^T B[b] (rows,N)[t]   for s < splits, b < batches */
int t2o_gemm_nt_batched(const float* A, const float* B, float* C, int batches, int M, int N, int K, int a_rows, void* stream);
int t2o_gemm_tn_splits(int batches, int rows, int M, int N);
int t2o_gemm_tn_batched(const float* A, const float* B, float* C, int batches, int rows, int M, int N, int splits, void* stream);
/* ... with A / B row ranges [0, rows) of every plane of LARGER (batches, a_plane_rows | b_plane_rows, .) tensors: the weight
 * gradient over the first passes of a train step's V / A dY A^T arenas when a step used fewer passes than the arena holds */
int t2o_gemm_tn_batched_ld(const float* A, const float* B, float* C, int batches, int rows, int a_plane_rows, int b_plane_rows, int M, int N,
                           int splits, void* stream);

/* ---- LSTM layers of the request encoder (models/lang_encoder.py:70-113: 2-layer bidirectional LSTM over packed, i.e.
 * per-sample-length, sequences; nn.LSTM gate order i, f, g, o), one launch per time step for both directions (t2o_rnn.hip).
 *   gi    (B, L, D*4H)  x W_ih^T for every step and direction (a library GEMM), no bias
 *   whh_t (D, H, H, 4)  whh_t[d][k][j][g] = W_hh[d][g*H + j][k]  (forward);   whh (D, H, H, 4): whh[d][c][k][q] =
 *                       W_hh[d][4*c + q][k]  (backward) -- both 16-byte-load repackings of nn.LSTM's (4H, H) weight;
 *                       b_ih / b_hh (D, 4H) nullable
 *   len   (B) int64     valid lengths on the DEVICE: a sample's state stops changing at its last token (direction 0) /
 *                       starts from zero there (direction 1), outputs are zero at pads -- what pack_padded_sequence ->
 *                       LSTM -> pad_packed_sequence computes, without the length sort and without host-side lengths
 *   out   (B, L, D*H);  hnew, cnew (L, D, B, H): state after processing time t (final state: t = L-1 for direction 0,
 *                       t = 0 for direction 1);  gates (L, D, B, 4H): post-activation gates, kept for the backward.
 * Backward: dout (B, L, D*H), dhn / dcn (D, B, H) nullable -> dgates (L, D, B, 4H) = gradients of the gate pre-activations
 * (weight, bias and input gradients are GEMMs / sums over them: host side); carry_h, dc (D, B, H): scratch.
 * H % 64 == 0, H <= 256, D in {1, 2}. */
int t2o_lstm_layer_fwd(const float* gi, const float* whh_t, const float* b_ih, const float* b_hh, const long long* len,
                       float* out, float* hnew, float* cnew, float* gates, int B, int L, int H, int D, void* stream);
int t2o_lstm_layer_bwd(const float* whh, const long long* len, const float* cnew, const float* gates, const float* dout,
                       const float* dhn, const float* dcn, float* dgates, float* carry_h, float* dc,
                       int B, int L, int H, int D, void* stream);

/* ---- image feature head: models/actor.py:50,142-143,215-216  feat = relu(bn1(vis_encoder.fc(pooled))) -- Linear(K -> D),
 * BatchNorm1d(D) in training (batch statistics over the B rows, running statistics updated as torch does: momentum,
 * unbiased variance, num_batches_tracked += 1) or evaluation mode, ReLU -- one launch (a workgroup owns 16 feature
 * columns of the whole batch, so the statistics never leave it).  B <= 64, K % 4 == 0; rows 16-byte aligned.
 * Backward: g_feat -> d_fc (gradient of the Linear's output, kept for the caller's weight-gradient product
 * d_fc^T . pooled), d_bn (2, D) = [d weight; d bias] of the batch norm, d_pooled = d_fc . fc_w (skipped when NULL). */
typedef struct t2o_image_feature {
  const float *fc_w, *fc_b;                    /* (D, K), (D) nullable */
  const float *bn_w, *bn_b;                    /* (D) nullable (affine=False) */
  float *running_mean, *running_var;           /* (D); nullable in training mode (no update) */
  long long* num_batches_tracked;              /* nullable */
  const float* pooled;                         /* (B, K) */
  float* pooled_copy;                          /* nullable: (B, K) copy for the caller's weight-gradient product */
  float *fc_out, *stats, *feat;                /* (B, D) pre-norm, (2, D) = mean, 1/std, (B, D) */
  const float* g_feat;                         /* backward: (B, D) */
  float *d_fc, *d_bn, *d_pooled;               /* (B, D), (2, D), (B, K) nullable */
  float momentum, eps;
  int training, B, K, D;
} t2o_image_feature_t;
int t2o_image_feature_fwd(const t2o_image_feature_t* args, void* stream);
int t2o_image_feature_bwd(const t2o_image_feature_t* args, void* stream);

/* ---- one decoding step: models/action_decoder.py:38-64 Decoder.forward_step with models/attention.py:17-44 inside --
 *   step_in = [embedding[prev_op] (E) | relu(vis_linear(feat)) (D)]
 *   2-layer LSTM(E + D -> D) one step (gate order i, f, g, o; biases nullable)  ->  q = h1n
 *   attn = softmax_l(q . enc_l) over ALL L rows, mix = sum_l attn_l enc_l          (t2o_attn_fwd)
 *   ctx = tanh(linear_out([mix | q]));  logp = log_softmax(out_linear(ctx))
 * as 6 launches.  Everything a backward needs is left in caller storage (`saved`): the activated gates, both new
 * states, copies hp0 / hp1 of the previous hidden states (nullable), mix, ctx, logp.
 * Backward (9 launches): gradients of logp / ctx / the new states (each nullable; at least one of g_ctx, g_logp) ->
 * d_feat, d_h0, d_c0, d_h1, d_c1, d_enc.  Only DATA gradients: the pre-activation gradients d_logits (B, V; written
 * only when g_logp is given), d_lin (B, D), d_gates1, d_gates0 (B, 4D), d_step_in (B, E + D), d_vis (B, D) stay in
 * caller storage and the caller forms each weight gradient as ONE product over all the steps of a train step
 * (out_linear: d_logits^T ctx; linear_out: d_lin^T [mix | h1n]; LSTM layer 1: d_gates1^T h0n, d_gates1^T hp1; layer 0:
 * d_gates0^T step_in, d_gates0^T hp0; vis_linear: d_vis^T feat_copy; embedding: rows prev_op += d_step_in[:, :E]).
 * D % 64 == 0, D <= 1024, E % 4 == 0, V <= 16, L <= 64; every buffer 16-byte aligned, rows dense. */
typedef struct t2o_decoder_step {
  /* parameters */
  const float *emb;                            /* (V, E) */
  const float *vis_w, *vis_b;                  /* (D, D), (D) nullable */
  const float *w_ih0, *w_hh0, *b_ih0, *b_hh0;  /* (4D, E + D), (4D, D), (4D), (4D) */
  const float *w_ih1, *w_hh1, *b_ih1, *b_hh1;  /* (4D, D), (4D, D), (4D), (4D) */
  const float *lo_w, *lo_b;                    /* attention.linear_out (D, 2D), (D) nullable */
  const float *out_w, *out_b;                  /* out_linear (V, D), (V) nullable */
  /* forward inputs */
  const long long* prev_op;                    /* (B) token of the previous operator */
  const float *feat, *h0, *c0, *h1, *c1;       /* (B, D) each */
  const float* enc;                            /* (B, L, D) request encoding */
  /* forward outputs and saved values */
  float *step_in;                              /* (B, E + D) */
  float *feat_copy, *hp0, *hp1;                /* (B, D) each, nullable: copies of feat / h0 / h1 (the X of their products) */
  long long* prev_op_copy;                     /* (B) nullable */
  float *gates0, *gates1;                      /* (B, 4D) activated i, f, g, o */
  float *h0n, *c0n, *h1n, *c1n;                /* (B, D) new states */
  float *attn, *mix, *ctx, *logp;              /* (B, L), (B, D), (B, D), (B, V) */
  /* backward inputs (nullable) */
  const float *g_logp, *g_ctx, *g_h0n, *g_c0n, *g_h1n, *g_c1n;
  /* backward: kept for the weight gradients */
  float *d_logits, *d_lin, *d_gates1, *d_gates0, *d_step_in, *d_vis;
  /* backward scratch */
  float *d_ctx, *d_mix, *d_qa, *d_q, *d_x1;    /* (B, D) each */
  /* backward outputs */
  float *d_feat, *d_h0, *d_c0, *d_h1, *d_c1;   /* (B, D) each */
  float* d_enc;                                /* (B, L, D) */
  int B, L, D, E, V;
} t2o_decoder_step_t;
int t2o_decoder_step_fwd(const t2o_decoder_step_t* args, void* stream);
int t2o_decoder_step_bwd(const t2o_decoder_step_t* args, void* stream);

/* ---- the two batch norms of a shortcut block in one pass each way: models/actor_resnet.py:33-36, 42-44
 *   out = relu(bn2(x) + bn_s(xs))        x = conv2's output, xs = the 1x1 stride-2 shortcut convolution's output, both (M, C)
 * Forward (3 launches instead of 5: the normalised shortcut is never stored): `partial` = the main branch's statistics rows
 * left by its producing convolution (t2o_conv3x3_fwd_stats_nhwc / t2o_wino_output_transform; NULL: a statistics pass is
 * made here), the shortcut branch's statistics are always made here.  Both batch norms get their batch statistics
 * (save_*), running statistics and — backward, 3 launches instead of 6: the gated gradient shared by both branches is
 * never stored — their input gradients dx / dxs and parameter gradients (accumulate != 0: added).  Same arithmetic in the
 * same order as t2o_bn_relu_nhwc_fwd(relu = 0) followed by t2o_bn_relu_nhwc_fwd_partials(res): bit-identical results. */
size_t t2o_bn_dual_nhwc_workspace_bytes(int M, int C);
int t2o_bn_dual_relu_nhwc_fwd(const float* x, const float* partial, int partial_rows, const float* xs,
                              const float* weight, const float* bias, float* running_mean, float* running_var, float* save_mean,
                              float* save_invstd, const float* weight_s, const float* bias_s, float* running_mean_s,
                              float* running_var_s, float* save_mean_s, float* save_invstd_s, float* out, float momentum, float eps,
                              float momentum_s, float eps_s, void* workspace, size_t workspace_bytes, int M, int C, void* stream);
int t2o_bn_dual_relu_nhwc_bwd_acc(const float* x, const float* xs, const float* y, const float* dy, const float* weight,
                                  const float* bias, const float* save_mean, const float* save_invstd, const float* weight_s,
                                  const float* bias_s, const float* save_mean_s, const float* save_invstd_s, float* dx, float* dxs,
                                  float* dweight, float* dbias, float* dweight_s, float* dbias_s, int accumulate, void* workspace,
                                  size_t workspace_bytes, int M, int C, void* stream);

/* Rewrites a captured, not yet instantiated hipGraph (hipGraph_t) in place: every memset node becomes a kernel node
 * doing the same fill, with the same dependencies and dependents; *replaced = how many.  Memset nodes were seen to
 * run out of order with neighbouring kernel nodes on replay (ROCm 7.2 / gfx950): t2onet_amd/graphs.py calls this on
 * the encoder's graphs before instantiating them (library calls inside them may enqueue hipMemsetAsync). */
int t2o_graph_memsets_to_kernels(void* graph, int* replaced);

/* ---- Adam over one flat fp32 buffer: experiments/t2onet/train_seq2seqL1.py:169 (torch.optim.Adam, default betas and
 * eps, no weight decay) -- param, grad, exp_avg, exp_avg_sq: n floats each, 16-byte aligned; `step` = 1 for the first
 * update (bias corrections are computed on the host in double).  In place; one streaming pass. */
int t2o_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr, float beta1,
                  float beta2, float eps, int step, void* stream);

/* ---- WEIGHT GRADIENT of the same layers in the Winograd domain with both transforms on chip (t2o_wino_wgrad.hip;
 * models/actor_resnet.py:24-44 under autograd in the reference): dw (Co,3,3,Ci) (+)= d loss / d w of y = conv2d(x, w, None, 1, 1)
 * for x (n_img,H,W,Ci), dy (n_img,H,W,Co) NHWC, H and W multiples of 16, Ci and Co multiples of 64 (<= 512), fewer than 2^31
 * pixels.  n_img may span several encoder passes laid side by side (encoder.WgradArena).  16 of the direct kernel's 36
 * multiplies; V = B^T d B and A dY A^T live in LDS only.  Deterministic: per-workgroup partial sums in the workspace
 * (t2o_wino_fused_wgrad_workspace_bytes), added in a fixed order, then G^T dU G (t2o_wino_dw_transform).  zeros: >= 17 * Ci * 4 +
 * 256 bytes of zeros, 16-byte aligned (t2o_conv_set_zero_region's block serves). */
int t2o_wino_fused_wgrad_supported(int n_img, int H, int W, int Ci, int Co);
size_t t2o_wino_fused_wgrad_workspace_bytes(int n_img, int H, int W, int Ci, int Co);
int t2o_wino_fused_wgrad_nhwc(const float* x, const float* dy, float* dw, const float* zeros, void* workspace, size_t workspace_bytes,
                              int n_img, int H, int W, int Ci, int Co, int accumulate, void* stream);

/* ---- the train step's one collective (the reference runs one process: train_seq2seqL1.py:74-88 has no
 * counterpart).  t2o_allreduce: in-place SUM all-reduce of n fp32 values over the caller's communicator (an ncclComm_t, passed
 * as void*), stream-ordered on `stream`, no host synchronisation; t2o_allreduce_mean: the same followed by x 1/nranks (the
 * data-parallel gradient average; flat 16-byte aligned).  RCCL is resolved at first use (dlopen: the copy the process already
 * holds, else librccl.so) -- not a link-time dependency; without it every entry point here returns T2O_EUNSUPPORTED and
 * t2o_comm_available() is 0.  For a host that has no communicator of its own: rank 0 calls t2o_comm_unique_id (128 bytes),
 * hands them to every rank by any means, all ranks call t2o_comm_init_rank (collective; the calling thread's current HIP
 * device is the rank's GPU) and, at the end, t2o_comm_destroy. */
int t2o_comm_available(void);
int t2o_comm_unique_id(void* id128);
int t2o_comm_init_rank(void** comm, int nranks, const void* id128, int rank);
int t2o_comm_destroy(void* comm);
int t2o_allreduce(float* flat, size_t n, void* comm, void* stream);
int t2o_allreduce_mean(float* flat, size_t n, void* comm, void* stream);

/* ---- the general fp32 matrix-core GEMM (t2o_gemm.hip) behind the dense products no specialised kernel takes: the request
 * encoder's input projection over all time steps and its weight / input gradients (models/lang_encoder.py:91-102, nn.LSTM's
 * x W_ih^T and the gradients of W_ih, W_hh), the decoder tape's weight gradients over all decoder steps of a train step
 * (models/action_decoder.py:52-63: vis_linear, both LSTM cells, attention.linear_out, out_linear) and fc (models/
 * actor_resnet.py:107).   C (M,N) = [C +] op(A) op(B), row-major, leading dimensions in floats:
 *   a_kmajor = 1: A is stored (K, M) (a "dy^T x" product sums over the rows of both operands), 0: (M, K);
 *   b_kmajor = 1: B is stored (K, N), 0: (N, K) (nn.Linear's weight).
 * Any M, N, K >= 1; operands / C may be column slices of larger matrices.  One workgroup per 64 x 64 tile of C; its 1, 2 or 4
 * contraction groups (chosen by K alone) walk their 32-deep chunks of K front to back and are added in group order: the rounding
 * depends on the shape only (no split-K across workgroups, no atomics) -- bitwise the same on every run and machine.
 * t2o_colsum: out[n] = [out[n] +] sum over the rows of X (R, N) (bias gradients), fixed order. */
int t2o_gemm(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int a_kmajor, int b_kmajor,
             int accumulate, void* stream);
int t2o_colsum(const float* X, float* out, int R, int N, int ldx, int accumulate, void* stream);

/* ---- Winograd F(2x2,3x3) with V and M kept on chip (t2o_wino_fused.hip): conv2d(x, w, None, 1, 1) on NHWC activations for the
 * stride-1 3x3 layers of the 64- / 128-channel stages (models/actor_resnet.py:24-44), H and W multiples of 16, Ci of 8, Co of
 * 64 -- ONE launch per layer where t2o_wino_input_transform + t2o_gemm_nt_batched + t2o_wino_output_transform move 8x the
 * activation through HBM.  uc = t2o_wino_u_chunked(U) with U = t2o_wino_weight_transform(w) (or of the mirrored transpose
 * for the data gradient): (Ci/8, 16, Co, 8).  addend (N,H,W,Co) or NULL is added in the epilogue; stats or NULL receives
 * (t2o_wino_fused_stats_rows, 2, Co) partial sums / sums of squares of y for t2o_bn_relu_nhwc_fwd_partials.  zeros: >= Ci*4 +
 * 32 bytes of zeros, 16-byte aligned (t2o_conv_set_zero_region's block serves); Ci, Co <= 1024 (so 4,128 bytes always suffice). */
int t2o_wino_fused_supported(int N, int H, int W, int Ci, int Co);
int t2o_wino_fused_stats_rows(int N, int H, int W);
int t2o_wino_u_chunked(const float* U, float* Uc, int Cn, int Ck, void* stream);
int t2o_wino_fused_conv_nhwc(const float* x, const float* uc, const float* addend, float* y, float* stats,
                             const float* zeros, int N, int H, int W, int Ci, int Co, void* stream);
/* ... as a data gradient in front of y = relu(bn(bn_x)) (see t2o_conv3x3_dgrad_pre_bnsums_nhwc): rows (t2o_wino_fused_stats_rows,
 * 2, Co) receive that batch norm's backward sums for t2o_bn_relu_nhwc_bwd_partials_acc; bn_x is fetched under the last chunk. */
int t2o_wino_fused_conv_bnsums_nhwc(const float* x, const float* uc, float* y, const float* bn_x, const float* save_mean,
                                    const float* save_invstd, const float* weight, const float* bias, float* rows,
                                    const float* zeros, int N, int H, int W, int Ci, int Co, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* T2ONET_HIP_H */
