/*
 * retinanet_hip.h -- C ABI of libretinanet_hip.so, the MI355X (gfx950) dense-head
 * path of RetinaNet: anchor emission, IoU + anchor/GT matching, fused focal +
 * smooth-L1 loss (value and gradient in one pass), box decode/clip and batched
 * per-class NMS + top-k.
 *
 * The reference (benihime91/pytorch_retinanet, pure Python) has no FFI of its
 * own; each entry point below replaces the torch / torchvision op sequence at the
 * cited reference lines (paths relative to the reference root).  Signatures are
 * plain C: raw device pointers, sizes, one opaque hipStream_t.  No torch types.
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer unless marked (host).  The caller owns all
 *     buffers including workspaces; the library allocates nothing and keeps no state.
 *   - All work is enqueued on `stream` (a hipStream_t, NULL = default stream); no
 *     entry point synchronises with the host.
 *   - Return value: 0 = RN_OK, negative = argument error (RN_E*), positive =
 *     hipError_t from a launch.
 *   - Tensors are dense, row-major, in the layouts written next to each argument.
 *     `cls`/`grad_cls` base pointers must be 16-byte aligned, `box`/`grad_box`
 *     8-byte (16-bit dtypes) or 16-byte (fp32) aligned.
 *   - `anchor_bstride`: element stride between images' anchor sets; 0 = one anchor
 *     set [A][4] shared by every image (the reference recomputes identical anchors
 *     per image, retinanet/anchors.py:223-228).
 *   - gt_off: int32[B+1] prefix offsets of each image's rows in gt_boxes/gt_labels.
 */
#ifndef RETINANET_HIP_H
#define RETINANET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RN_ABI_VERSION 10
#define RN_MAX_LEVELS 8

enum rn_dtype { RN_F32 = 0, RN_BF16 = 1, RN_F16 = 2 };

enum rn_status {
    RN_OK = 0,
    RN_EINVAL = -1,       /* null pointer / non-positive size / bad enum */
    RN_EALIGN = -2,       /* pointer alignment requirement not met */
    RN_EWORKSPACE = -3,   /* workspace too small */
    RN_EUNSUPPORTED = -4, /* shape outside the supported range (see entry point) */
    RN_ETHRESH = -5       /* fg_thr <= bg_thr (retinanet/box_utils.py:66 assert) */
};

/* One pyramid level: feature-map size, stride and anchors per location. */
typedef struct rn_level { int32_t H, W, stride, num_cell; } rn_level;

/* retinanet/config.py:85-87 (gamma, alpha, beta), losses.py:84 (logit_shift = 1),
 * box_utils.py:32 (log_eps = 1e-8), config.py:67 (reg_w). */
typedef struct rn_loss_params {
    float alpha, gamma, beta, logit_shift, log_eps;
    float reg_w[4];
} rn_loss_params;

/* retinanet/config.py:71-75 (score_thr, nms_thr, max_det), models.py:203 (min_box = 1e-2). */
typedef struct rn_detect_params {
    float score_thr, min_box, nms_thr;
    int32_t max_det;
    float reg_w[4];
} rn_detect_params;

int rn_version(void);
const char *rn_status_string(int status);

/* Node census of a captured hipGraph (`graph`: hipGraph_t): counts[0] kernel, [1] memset, [2] memcpy, [3] other node types.
 * Why it exists: memset NODES of a replayed hipGraph misbehave on ROCm 7.0 -- after the process has synchronised with the device
 * and enqueued other blit work (fills, small copies) a replayed memset writes garbage (round 4: a 32-byte hipMemsetAsync in front
 * of the matcher scaled every later loss by 1 / garbage).  This library issues no hipMemsetAsync; a host that captures its calls
 * together with other work (MIOpen's split-K weight gradients clear their output with one at some shapes) can check its graph. */
int rn_hipgraph_node_census(void *graph, int64_t counts[4]);
/* The repair: every memset node of `graph` (not yet instantiated) is replaced by a kernel node with the same destination, value,
 * element size, width, height and pitch, the same dependencies and the same dependents; *replaced (nullable) = their number.
 * graph.CapturedTrainStep calls it between capture_end and instantiate (torch.cuda.CUDAGraph(keep_graph=True)). */
int rn_hipgraph_replace_memset_nodes(void *graph, int64_t *replaced);

/* ---- K1 anchors_emit --------------------------------------------------------
 * Replaces AnchorGenerator._compute_grid_offsets / grid_anchors / forward's cat,
 * retinanet/anchors.py:151-170, :172-197, :228.  Bit-exact with the CPU path.
 * levels (host)[L]; cell_anchors (host array of L device pointers, each
 * f32[num_cell][4], = the module's `cell_anchors.{l}` buffers); out f32[A][4]
 * with A = rn_anchors_count(levels, L), levels concatenated in order, row-major
 * (y, x), cell anchor fastest. */
int64_t rn_anchors_count(const rn_level *levels, int L);
int rn_anchors_emit(const rn_level *levels, int L, const float *const *cell_anchors,
                    double offset, float *out, void *stream);

/* ---- K2 iou_match -----------------------------------------------------------
 * Replaces matcher(), retinanet/box_utils.py:51-80, and the torchvision
 * box_iou it calls (:74); the [T,A] IoU matrix is never materialised.
 * matches i64[B][A]: -2 ignore, -1 background, >=0 index of the matched GT row
 * within its image (lowest index on ties).  num_fg (nullable) i32[B] = count of
 * matches >= 0 per image.  Bit-exact with the CPU path for fp32 inputs. */
int rn_iou_match(const float *anchors, int64_t anchor_bstride,
                 const float *gt_boxes, const int32_t *gt_off, int B, int64_t A,
                 float fg_thr, float bg_thr, int64_t *matches, int32_t *num_fg, void *stream);
/* Same, with a host-side hint: total_gt = gt_off[B] - gt_off[0] when the caller knows it (tensor shapes), -1
 * otherwise.  gt_off lives on the device and the call never syncs, so only the hint can select the batch-shaped
 * kernel (shared anchors, <= 1024 GT boxes in the batch, <= 32 per image on average: one thread = one anchor x
 * all images).  A wrong hint never corrupts memory, but the matches of GT rows beyond it are undefined. */
int rn_iou_match_ex(const float *anchors, int64_t anchor_bstride,
                    const float *gt_boxes, const int32_t *gt_off, int B, int64_t A,
                    float fg_thr, float bg_thr, int64_t *matches, int32_t *num_fg,
                    int64_t total_gt, void *stream);

/* rn_iou_match_ex that also writes the loss kernel's shortcut to the rows that are not plain background:
 * special_rows u64[B][(A + 63) / 64] (rn_iou_match_special_bytes), bit (a & 63) of word a >> 6 = [matches[b][a] != -1]
 * (matched or ignored rows, ~0.3 % at the reference's thresholds).  rn_loss_fwd_bwd_levels_ex then fetches a 64-row chunk's
 * flags with two scalar loads and reads `matches` only where a bit is set, instead of streaming all of it (NULL: not written). */
size_t rn_iou_match_special_bytes(int B, int64_t A);
int rn_iou_match_special(const float *anchors, int64_t anchor_bstride,
                         const float *gt_boxes, const int32_t *gt_off, int B, int64_t A,
                         float fg_thr, float bg_thr, int64_t *matches, int32_t *num_fg,
                         uint64_t *special_rows, int64_t total_gt, void *stream);
/* ... with flags (ABI 8), for the caller that feeds the results straight to rn_loss_fwd_bwd_levels_ex (the training path):
 *   RN_MATCH_FLAGGED_ONLY   `matches` is written ONLY at rows whose flag bit is set (matched or ignored rows); every other entry
 *                           keeps what the buffer held.  The loss kernel reads `matches` through the flag words, so the B * A * 8
 *                           bytes of int64 codes -- 12.9 of the 16.1 MB this call moves at the train shape -- are never stored.
 *                           Needs special_rows; kernels that cannot skip the stores (large GT sets) write everything as before.
 *   RN_MATCH_NUM_FG_ZEROED  num_fg[0..B) already holds zeros (the caller cleared it with its own input copies): no clear launch. */
enum { RN_MATCH_NUM_FG_ZEROED = 1, RN_MATCH_FLAGGED_ONLY = 2 };
int rn_iou_match_special_ex(const float *anchors, int64_t anchor_bstride,
                            const float *gt_boxes, const int32_t *gt_off, int B, int64_t A,
                            float fg_thr, float bg_thr, int64_t *matches, int32_t *num_fg,
                            uint64_t *special_rows, int64_t total_gt, int flags, void *stream);

/* ---- K3 loss_fwd_bwd --------------------------------------------------------
 * Replaces RetinaNetLosses.forward / calc_loss / focal_loss / smooth_l1_loss,
 * retinanet/losses.py:19-145, and bbox_2_activ, retinanet/box_utils.py:25-34:
 * one pass over cls [B][A][K] and box [B][A][4] (dtype in {f32,bf16,f16}) that
 * produces out_loss f32[2] = {classification_loss, regression_loss} (batch means
 * of the per-image normalised sums) AND the gradients of those two scalars w.r.t.
 * cls / box (same dtype and shape; grad_* may be NULL for value-only).
 * gt_labels i64 in 1..K.  num_fg i32[B] as produced by rn_iou_match.
 * workspace: rn_loss_workspace_bytes(B, A, K) bytes, 16-byte aligned. */
size_t rn_loss_workspace_bytes(int B, int64_t A, int K);
int rn_loss_fwd_bwd(const void *cls, const void *box, int dtype, int B, int64_t A, int K,
                    const float *anchors, int64_t anchor_bstride,
                    const float *gt_boxes, const int64_t *gt_labels, const int32_t *gt_off,
                    const int64_t *matches, const int32_t *num_fg, const rn_loss_params *params,
                    float *out_loss, void *grad_cls, void *grad_box,
                    void *workspace, size_t workspace_bytes, void *stream);

/* Same, reading the head outputs where the convolutions left them: L per-level tensors
 * cls_levels[l] [B][A_l][K], box_levels[l] [B][A_l][4] (host arrays of L device pointers;
 * level_anchors (host) = A_l; sum A_l = A, the row length of `matches` and of `anchors`, levels in
 * anchor order) instead of their concatenation -- removes the reference's torch.cat at
 * retinanet/layers.py:195, :259 and its backward.  grad_*_levels: per-level outputs (both NULL for
 * value-only). */
int rn_loss_fwd_bwd_levels(const void *const *cls_levels, const void *const *box_levels,
                           const int64_t *level_anchors, int L, int dtype, int B, int K,
                           const float *anchors, int64_t anchor_bstride,
                           const float *gt_boxes, const int64_t *gt_labels, const int32_t *gt_off,
                           const int64_t *matches, const int32_t *num_fg, const rn_loss_params *params,
                           float *out_loss, void *const *grad_cls_levels, void *const *grad_box_levels,
                           void *workspace, size_t workspace_bytes, void *stream);
/* grad_cls_levels[l] must not alias cls_levels[l] (RN_EINVAL): the repair phase re-reads logits the stream has passed. */
/* Same call; additionally records the caller's HIP events (hipEvent_t, may be NULL) on `stream` immediately before and
 * after the streaming kernel -- the dominant kernel of the call -- so a benchmark can time that kernel alone (the
 * one-block finalize that follows is outside the pair). */
/* rn_loss_fwd_bwd_levels with the special-row words of rn_iou_match_special (nullable: then `matches` is streamed) and an
 * optional pair of HIP events recorded on `stream` right around the streaming kernel (nullable; bench.py's roofline). */
int rn_loss_fwd_bwd_levels_ex(const void *const *cls_levels, const void *const *box_levels,
                              const int64_t *level_anchors, int L, int dtype, int B, int K,
                              const float *anchors, int64_t anchor_bstride, const float *gt_boxes,
                              const int64_t *gt_labels, const int32_t *gt_off, const int64_t *matches,
                              const uint64_t *special_rows, const int32_t *num_fg, const rn_loss_params *params,
                              float *out_loss, void *const *grad_cls_levels, void *const *grad_box_levels, void *workspace,
                              size_t workspace_bytes, void *stream, void *event_start, void *event_stop);
/* rn_loss_fwd_bwd_levels_ex WITHOUT the one-block finalize launch (ABI 8): every workgroup of the streaming kernel adds its two
 * partial sums -- as 2^-32 fixed point, so that the total is an integer sum and therefore the same bits in any order -- to two words
 * of one of 64 cache lines of `state` with relaxed device-scope atomics and bumps that line's arrival counter; one wave of workgroup 0
 * polls the 64 counters, sums the lines when everybody has arrived, writes out_loss and leaves the words zeroed.  The finalize KERNEL
 * of the other entry points uses the same arithmetic, so both give the same bits.  `state`: >= 256 + 64 * 64 = 4 352 bytes, 64-byte
 * aligned (a rn_loss_match_state_bytes(>= 1 072) buffer serves: this form uses bytes 256 .. 4 351, the fused form words 0, 1 and its
 * counters, all zero between calls); the caller zero-fills it ONCE, every completed call leaves it zero-filled, and
 * it must not be shared by calls that can run concurrently (one per stream).  A workgroup that never arrives (a faulted launch) makes
 * out_loss NaN after a bounded wait instead of hanging the stream. */
int rn_loss_fwd_bwd_levels_fin(const void *const *cls_levels, const void *const *box_levels,
                               const int64_t *level_anchors, int L, int dtype, int B, int K,
                               const float *anchors, int64_t anchor_bstride, const float *gt_boxes,
                               const int64_t *gt_labels, const int32_t *gt_off, const int64_t *matches,
                               const uint64_t *special_rows, const int32_t *num_fg, const rn_loss_params *params,
                               float *out_loss, void *const *grad_cls_levels, void *const *grad_box_levels, void *workspace,
                               size_t workspace_bytes, void *state, void *stream, void *event_start, void *event_stop);
/* rn_loss_fwd_bwd_levels_fin with two more arguments (ABI 10; reference: retinanet/losses.py:49-111, and the fp16 run of
 * demo.ipynb, Lightning precision = 16):
 *   grad_prescale  nullable device f32[1].  Every GRADIENT the call writes (not the two losses) is multiplied by *grad_prescale
 *                  BEFORE it is rounded to the I/O dtype.  A torch.amp.GradScaler multiplies the fp32 loss by its scale before
 *                  backward, so the reference's fp16 class-head gradients (0.25 p^3 / (num_fg B) ~ 4e-10 for a background
 *                  element at the prior) are stored as ~2.6e-5; written unscaled they would flush to zero below fp16's
 *                  smallest subnormal (6e-8) before any later multiplication.  The caller multiplies by upstream / prescale
 *                  in backward (a no-op when upstream == prescale).
 *   form           how the special rows (matched / ignored: the flag words) are repaired.  Same arithmetic in all three; gradients
 *                  bit for bit, losses to the last bits of the 2^-32 fixed-point sums:
 *                  RN_LOSS_FORM_CHUNKS (0)       one launch; every streaming wave repairs its own range before / after its stream,
 *                                                64-row chunk by chunk (the _fin form).  Best at the train shape (~0.3 % special rows).
 *                  RN_LOSS_FORM_REPAIR_PASS (1)  TWO launches -- a pure background stream over the logits (no row logic) and a repair
 *                                                kernel that walks the flag words (`special_rows` required), one special row per lane,
 *                                                and finishes the sums; `workspace` may be NULL; the event pair brackets both.
 *                                                Measured slower on MI355X (the repair's dependent loads have nothing to hide under);
 *                                                kept for A/B.
 *                  RN_LOSS_FORM_LIST (2)         one launch; the flagged rows of ALL of a wave's chunks go through one compact list
 *                                                (the dependent loads run once per 64 special rows, not once per chunk) and ignored rows
 *                                                move as 16-byte pieces.  Best from ~32 GT boxes per image on (BASELINE configs[4]:
 *                                                500 per image, ~19 k special rows per image). */
#define RN_LOSS_FORM_CHUNKS 0
#define RN_LOSS_FORM_REPAIR_PASS 1
#define RN_LOSS_FORM_LIST 2
int rn_loss_fwd_bwd_levels_rp(const void *const *cls_levels, const void *const *box_levels,
                              const int64_t *level_anchors, int L, int dtype, int B, int K,
                              const float *anchors, int64_t anchor_bstride, const float *gt_boxes,
                              const int64_t *gt_labels, const int32_t *gt_off, const int64_t *matches,
                              const uint64_t *special_rows, const int32_t *num_fg, const rn_loss_params *params,
                              const float *grad_prescale, int form, float *out_loss, void *const *grad_cls_levels,
                              void *const *grad_box_levels, void *workspace, size_t workspace_bytes, void *state,
                              void *stream, void *event_start, void *event_stop);
/* K2 + K3 in ONE launch (round 4): the matcher of retinanet/box_utils.py:51-80 runs in the loss kernel's prologue -- every wave
 * matches the anchor rows of its own range against the image's GT boxes (one box per lane, so max_gt_per_image <= 64), the
 * per-image foreground counts meet in device-scope counters behind a grid barrier (the launch uses the resident grid only), and
 * the match codes never leave the chip unless `matches_out` (nullable, i64[B][A]) asks for them.  Same results as
 * rn_iou_match + rn_loss_fwd_bwd_levels bit for bit (match codes, num_fg) / to the last ulp of the same arithmetic (losses,
 * gradients).  num_fg_out i32[B] is written by the finalize kernel.  `state` (rn_loss_match_state_bytes(B), 64-byte aligned)
 * holds the barrier word and the counters: the caller zero-fills it ONCE; every completed call leaves it zero-filled; it must
 * not be shared by calls that can run concurrently (one per stream) NOR overlap kernels of other streams that occupy wave slots:
 * the grid barrier needs every workgroup co-resident (the grid is sized from an occupancy estimate); a barrier that is not
 * complete after a bounded wait poisons out_loss with NaN instead of hanging.  RN_EUNSUPPORTED: more than 64 GT boxes in an image, or a
 * shape whose per-wave row range does not fit the kernel's lists (then call rn_iou_match_special + rn_loss_fwd_bwd_levels_ex).
 * event_start / event_stop as in rn_loss_fwd_bwd_levels_ex. */
size_t rn_loss_match_state_bytes(int B);
int rn_loss_match_fwd_bwd_levels(const void *const *cls_levels, const void *const *box_levels,
                                 const int64_t *level_anchors, int L, int dtype, int B, int K,
                                 const float *anchors, int64_t anchor_bstride, const float *gt_boxes,
                                 const int64_t *gt_labels, const int32_t *gt_off, int max_gt_per_image,
                                 float fg_thr, float bg_thr, int64_t *matches_out, int32_t *num_fg_out,
                                 const rn_loss_params *params, float *out_loss, void *const *grad_cls_levels,
                                 void *const *grad_box_levels, void *workspace, size_t workspace_bytes, void *state,
                                 size_t state_bytes, void *stream, void *event_start, void *event_stop);
int rn_loss_fwd_bwd_levels_timed(const void *const *cls_levels, const void *const *box_levels,
                                 const int64_t *level_anchors, int L, int dtype, int B, int K,
                                 const float *anchors, int64_t anchor_bstride, const float *gt_boxes,
                                 const int64_t *gt_labels, const int32_t *gt_off, const int64_t *matches,
                                 const int32_t *num_fg, const rn_loss_params *params, float *out_loss,
                                 void *const *grad_cls_levels, void *const *grad_box_levels, void *workspace,
                                 size_t workspace_bytes, void *stream, void *event_start, void *event_stop);

/* In-place data[i] *= *scale (device scalar); returns immediately on the device
 * when *scale == 1.  Used by autograd's backward to apply the upstream gradient to
 * the gradients rn_loss_fwd_bwd already wrote, without a host sync. */
int rn_scale_inplace(void *data, int dtype, int64_t n, const float *scale, void *stream);
/* The same for up to 16 tensors of one dtype in one launch: data[k][0..n[k]) *= *scales[k] (device scalars). */
int rn_scale_inplace_batched(void *const *data, const int64_t *n, const float *const *scales, int count, int dtype, void *stream);

/* ---- fused BatchNorm2d (+ residual) (+ ReLU), channels-last ---------------------------------
 * Conv-stack widening.  Replaces the bn -> (+identity) -> relu sequences of the
 * reference's residual blocks, retinanet/backbone.py:70-80 and :118-136, and the stem :248-250:
 *   y = relu?( (x - mean) * invstd * gamma + beta  (+ residual) )
 * x, residual, y: [M = N*H*W][C] (channels_last [N,C,H,W]), dtype in {f32, bf16, f16}, C % 8 == 0;
 * gamma, beta, running_*, save_*: f32[C] (gamma/beta may be NULL = 1/0).  training != 0: batch
 * statistics, running stats updated with `momentum` (unbiased variance), *num_batches_tracked += 1
 * (nullable); training == 0: running statistics.  save_mean / save_invstd / coef ([2][C] forward,
 * [3][C] backward scratch) are caller-owned; workspace: rn_bn_workspace_bytes(C).
 * Backward returns dx, dresidual (nullable; = gradient after the ReLU mask), dgamma, dbeta.  With relu != 0
 * and y == NULL the ReLU mask is recomputed from x and fwd_coef (= the coef array the forward call filled),
 * which saves one activation read per backward kernel; only valid when the forward had no residual.
 * relu_mask (forward, nullable; u8[M*C/8]): one bit per element, set where y > 0.  Backward with relu == 2 takes that
 * mask through the `y` argument instead of the activation (1/16 of its bytes) -- the residual layers' variant. */
size_t rn_bn_workspace_bytes(int C);
int rn_bn_act_forward(const void *x, const void *residual, void *y, int dtype, int64_t M, int C,
                      const float *gamma, const float *beta, float *running_mean, float *running_var,
                      int64_t *num_batches_tracked, int training, float momentum, float eps, int relu,
                      float *save_mean, float *save_invstd, float *coef, uint8_t *relu_mask, void *workspace,
                      size_t workspace_bytes, void *stream);
int rn_bn_act_backward(const void *dy, const void *y, const void *x, void *dx, void *dresidual, int dtype,
                       int64_t M, int C, const float *gamma, const float *save_mean, const float *save_invstd,
                       const float *fwd_coef, int training, int relu, float *dgamma, float *dbeta, float *coef,
                       void *workspace, size_t workspace_bytes, void *stream);

/* The same BatchNorm arithmetic in pieces, for layers whose statistics come out of a GEMM epilogue and whose normalisation
 * goes into a GEMM operand load (rn_pw_conv_*): every piece is the kernel rn_bn_act_* runs, bit for bit.
 *   rn_bn_stats            statistics pass only (partial + final): fills save_mean / save_invstd / coef [2][C], updates the
 *                          running statistics; the activation is not written.
 *   rn_bn_stats_finalize   the final step alone, from `nblocks` rows of partial sums f32[nblocks][2][C] (sum, sum of squares)
 *                          produced elsewhere (rn_pw_conv_forward, epilogue RN_PW_EPI_STATS).
 *   rn_bn_apply            y = relu?(x * coef_a + coef_b (+ residual)) (+ relu_mask bits).
 *   rn_bn_bwd_reduce       backward sums (partial + final): dgamma, dbeta and coef3 [3][C] = (a, k0, k1) of
 *                          dx = a * g' + k1 * x + k0; relu as in rn_bn_act_backward (0 / 1 / 2).
 *   rn_bn_bwd_finalize     the final step alone, from partial sums f32[nblocks][2][C] = (sum g', sum g' * xhat).
 *   rn_bn_bwd_apply        dx (and dresidual = g', nullable) from coef3; relu_mode 0 none, 1 mask from y, 2 recomputed from
 *                          x and fwd_coef, 3 bits in `y`. */
int rn_bn_stats(const void *x, int dtype, int64_t M, int C, const float *gamma, const float *beta, float *running_mean,
                float *running_var, int64_t *num_batches_tracked, float momentum, float eps, float *save_mean,
                float *save_invstd, float *coef, void *workspace, size_t workspace_bytes, void *stream);
int rn_bn_stats_finalize(const float *partial, int nblocks, int64_t M, int C, const float *gamma, const float *beta,
                         float *running_mean, float *running_var, int64_t *num_batches_tracked, float momentum, float eps,
                         float *save_mean, float *save_invstd, float *coef, void *stream);
int rn_bn_apply(const void *x, const void *residual, void *y, int dtype, int64_t M, int C, const float *coef, int relu,
                uint8_t *relu_mask, void *stream);
/* rn_bn_apply with relu = 1 whose residual is itself a BatchNorm output that was never written: `residual` holds that
 * layer's INPUT and res_coef [2][C] its coefficients; y = relu(x * a + b + round(residual * ra + rb)) -- the value a
 * separate rn_bn_apply pass would have stored (the downsample branch of a bottleneck,
 * /root/reference/retinanet/backbone.py:122-136). */
int rn_bn_apply_res_affine(const void *x, const void *residual, const float *res_coef, void *y, int dtype, int64_t M, int C,
                           const float *coef, uint8_t *relu_mask, void *stream);
int rn_bn_bwd_reduce(const void *dy, const void *y, const void *x, int dtype, int64_t M, int C, const float *gamma,
                     const float *save_mean, const float *save_invstd, const float *fwd_coef, int training, int relu,
                     float *dgamma, float *dbeta, float *coef3, void *workspace, size_t workspace_bytes, void *stream);
int rn_bn_bwd_finalize(const float *partial, int nblocks, int64_t M, int C, const float *gamma, const float *save_mean,
                       const float *save_invstd, int training, float *dgamma, float *dbeta, float *coef3, void *stream);
int rn_bn_bwd_apply(const void *dy, const void *y, const void *x, void *dx, void *dresidual, int dtype, int64_t M, int C,
                    const float *coef3, const float *fwd_coef, int relu_mode, void *stream);

/* ---- backbone / FPN convolutions as MFMA GEMMs with the surrounding BatchNorm fused in (csrc/pw.hip) --------------------
 * Replaces the conv -> bn -> relu chains of the reference's Bottleneck (retinanet/backbone.py:105-136: conv1x1, conv3x3,
 * conv1x1, the strided 1x1 downsample) and their autograd backward, bf16 channels-last, f32 accumulation.
 *   x [rows][Cin]  (= [Nimg][H][W][Cin]),  w [N][taps][Cin] (= channels-last [N][Cin][kh][kw]),  y [M][N] (= [Nimg][Ho][Wo][N]),
 *   M = Nimg * Ho * Wo; taps 1 (1x1, pad 0) or 9 (3x3, pad 1); stride 1 or 2; Cin % 64 == 0, N % 64 == 0.
 * Prologue (applied to the activation operand on its way into the GEMM):
 *   RN_PW_PRO_AFFINE_RELU  x' = relu(x * a[c] + b[c])                        -- the previous layer's BatchNorm + ReLU; padding
 *                          positions of a 3x3 conv stay zero (they pad the ACTIVATION)
 *   RN_PW_PRO_BN_BWD       x' = a[c] * g' + c[c] * x2 + b[c], g' = x masked by the ReLU (relu_mode 0 none; 2: where
 *                          fma(x2, fa, fb) rounds to a positive bf16; 3: bits [rows][Cin / 8]) -- BatchNorm backward of the layer
 *                          whose gradient this GEMM consumes (a, b, c = coef3 of rn_bn_bwd_reduce / _finalize)
 * Epilogue:
 *   RN_PW_EPI_STATS        partial f32[rn_pw_walkers(M)][2][N]: column sums and sums of squares of the bf16 output
 *   RN_PW_EPI_RESID        y += resid * rbits  (the identity branch's gradient: resid [M][N], rbits [M][N / 8])
 *   RN_PW_EPI_RELU_BWD     y = y * [fma(zprev, ea, eb) > 0 in bf16]; partial f32[walkers][2][N] = (sum y, sum y * (zprev - emean) * einv)
 *                          -- ReLU backward + the two sums of the BatchNorm backward of the layer BELOW this data gradient
 *   RN_PW_EPI_BIAS         y = act(y + bias[n] (+ resid)), act = ReLU when `relu`: alone or OR-ed with RN_PW_EPI_RESID (unmasked, no prologue) --
 *                          inference: bn(conv(x)) with the BatchNorm folded into w and bias, + identity, + ReLU in the GEMM's epilogue
 *                          (retinanet/backbone.py:118-136 under eval(): one pass over the block's largest tensor less per convolution)
 * rn_pw_conv_wgrad: dw [N][taps][Cin] = sum_m gpro(g)[m][N] x xpro(x)[pos(m, tap)][Cin]; gpro: none / BN_BWD, xpro: none / AFFINE_RELU;
 * workspace rn_pw_wgrad_workspace_bytes(d) (f32 partials of the position splits, summed in a fixed order). */
enum { RN_PW_PRO_NONE = 0, RN_PW_PRO_AFFINE_RELU = 1, RN_PW_PRO_BN_BWD = 2 };
enum { RN_PW_EPI_NONE = 0, RN_PW_EPI_STATS = 1, RN_PW_EPI_RESID = 2, RN_PW_EPI_RELU_BWD = 4, RN_PW_EPI_BIAS = 8 };
/* dtype: element type of x / w / y / g / dw -- RN_BF16 or RN_F16 (0, what a caller from before ABI version 8 leaves in the struct's
 * tail padding, means RN_BF16); the struct's size did not change. */
typedef struct rn_pw_conv { int64_t M; int32_t Cin, N, taps, stride, pad, Ho, Wo, H, W, dtype; } rn_pw_conv;
typedef struct rn_pw_prologue {
    int32_t kind, relu_mode;
    const float *a, *b, *c, *fa, *fb;
    const void *x2;
    const uint8_t *bits;
} rn_pw_prologue;
typedef struct rn_pw_epilogue {
    int32_t kind;
    float *partial;
    const void *resid;
    const uint8_t *rbits;
    const void *zprev;
    const float *ea, *eb, *emean, *einv;
    /* RN_PW_EPI_RESID: rbits may be NULL (y += resid, no mask).  res_stride == 2: resid lives on the stride-2 grid
     * [n][ceil(res_h / 2)][ceil(res_w / 2)][N] of the output grid [n][res_h][res_w] and is added at the rows with even
     * (y, x) only -- the data gradient of a 1x1 / stride-2 convolution (the bottleneck's downsample branch,
     * /root/reference/retinanet/backbone.py:179-186) joining conv1's data gradient without being scattered first.
     * 0 / 1: resid is [M][N] like y. */
    int32_t res_stride, res_h, res_w;
    int32_t relu;           /* RN_PW_EPI_BIAS: ReLU after the bias (and the residual) */
    const float *bias;      /* RN_PW_EPI_BIAS: f32 [N], 16-byte aligned */
} rn_pw_epilogue;
int rn_pw_walkers(int64_t M);
int rn_pw_conv_forward(const rn_pw_conv *d, const void *x, const void *w, void *y, const rn_pw_prologue *pro,
                       const rn_pw_epilogue *epi, void *stream);
size_t rn_pw_wgrad_workspace_bytes(const rn_pw_conv *d);
int rn_pw_conv_wgrad(const rn_pw_conv *d, const void *g, const void *x, void *dw, const rn_pw_prologue *gpro,
                     const rn_pw_prologue *xpro, void *workspace, size_t workspace_bytes, void *stream);
/* rn_pw_conv_wgrad in two steps, so that the split reductions of several weight gradients (a bottleneck's three or four) share ONE
 * launch: _partial runs the position-contraction kernel only (f32 partials of *splits position splits in `workspace`, which must
 * stay untouched until the reduction), rn_pw_wgrad_reduce_many sums up to 8 of them into their bf16 gradients (n_elems[i] =
 * N * taps * Cin of gradient i; HOST arrays). */
int rn_pw_wgrad_reduce_many_dt(const void *const *partials, const int *splits, const int64_t *n_elems, void *const *dws, int n, int dtype,
                               void *stream);
int rn_pw_conv_wgrad_partial(const rn_pw_conv *d, const void *g, const void *x, const rn_pw_prologue *gpro, const rn_pw_prologue *xpro,
                             void *workspace, size_t workspace_bytes, int *splits, void *stream);
int rn_pw_wgrad_reduce_many(const void *const *partials, const int *splits, const int64_t *n_elems, void *const *dws, int n, void *stream);
/* conv3 of a bottleneck (1x1 / stride 1, Cm -> C4 channels; /root/reference/retinanet/backbone.py:114, applied at :131-132), BOTH
 * gradients of its autograd backward in one pass over the block-output gradient (ABI 9).  Equivalent to
 *   rn_pw_conv_forward(g, w3t; prologue RN_PW_PRO_BN_BWD(a3, k0, k1, x2 = z3, bits, relu_mode 3); epilogue RN_PW_EPI_RELU_BWD(partial_bn,
 *                      zprev = z2, ea, eb, emean, einv)) -> dy2       and
 *   rn_pw_conv_wgrad_partial(g, z2; gpro = the same prologue, xpro = RN_PW_PRO_AFFINE_RELU(ea, eb)) -> f32 partials in `workspace`,
 * which each stream g, z3 and the ReLU bits -- the block's largest tensors -- once; here they are read once for both.
 *   g, z3 [M][C4], bits [M][C4 / 8], w3t [Cm][C4] (w3 transposed), z2, dy2 [M][Cm], 16-bit elements of `dtype`; a3 / k0 / k1 f32 [C4],
 *   ea / eb / emean / einv f32 [Cm] (16-byte aligned); partial_bn f32 [walkers][2][Cm] with walkers = rn_pw_conv3_backward_walkers(M, Cm, C4)
 *   (the row count rn_bn_bwd_finalize is given); workspace: rn_pw_conv3_backward_workspace_bytes(M, Cm, C4) bytes = [walkers][C4][Cm] f32,
 *   *splits = walkers: the (partials, splits) pair rn_pw_wgrad_reduce_many_dt sums into dW3 [C4][Cm].
 * Shapes: (Cm, C4) = (64, 256) and (128, 512) -- layer1 / layer2 of the ResNet-50 trunk; walkers() returns 0 and the call
 * RN_EUNSUPPORTED for anything else (the caller keeps the two separate launches).  dy2 is bit-identical to the separate launch. */
/* The end of one bottleneck and the start of the next in one pass (forward, ABI 9): the block output
 *   y = relu(fma(z3, oa, ob) + r),  r = resid (identity block) or round(fma(resid, res_a, res_b)) (resid = the INPUT of the downsample
 *   branch's BatchNorm; res_a / res_b both NULL otherwise),  ybits [M][C4 / 8] = [y alive]
 * -- exactly rn_bn_apply(z3, resid, relu = 1, bits) / rn_bn_apply_res_affine (/root/reference/retinanet/backbone.py:132-136) -- is
 * written AND multiplied, from LDS, into the next block's conv1 (1x1, C4 -> CN; backbone.py:118):  z1 [M][CN] = y . w1^T with
 * partial f32 [walkers][2][CN] = column sums / sums of squares of z1 as stored (what rn_pw_conv_forward(y, w1, RN_PW_EPI_STATS) returns;
 * z1 is bit-identical to it).  The separate launches write y and read it straight back.  16-bit elements of `dtype`; oa / ob / res_a /
 * res_b f32 [C4], 16-byte aligned; w1 [CN][C4].  (C4, CN) = (256, 64), (256, 128), (512, 128): walkers() is 0 for anything else. */
int rn_pw_block_out_conv1_walkers(int64_t M, int C4, int CN);
int rn_pw_block_out_conv1(int64_t M, int C4, int CN, int dtype, const void *z3, const void *resid, const float *res_a, const float *res_b,
                          const float *oa, const float *ob, const void *w1, void *y, uint8_t *ybits, void *z1, float *partial, void *stream);
/* The start of one bottleneck's backward and of the one before it in one pass (ABI 9): conv1's data gradient joined by the identity
 * branch's,  dx [M][C4] = dz1 [M][Cm] . w1t^T + resid * rbits  -- rn_pw_conv_forward(dz1, w1t, epilogue RN_PW_EPI_RESID(resid, rbits,
 * res_stride, res_h, res_w)), bit for bit -- and, on the tile just stored, the sums of the PREVIOUS block's bn3 backward over dx (which is
 * the gradient at that block's output):  partial f32 [walkers][2][C4] = (sum g', sum g' * (prev_z3 - prev_mean) * prev_invstd),
 * g' = dx * prev_bits -- what rn_bn_bwd_reduce(dx, prev_bits, prev_z3, ..) sums before it finalizes; rn_bn_bwd_finalize(partial, walkers, ..)
 * completes it.  Saves that launch's read of dx.  Cm = 64 or 128, C4 a multiple of 128; w1t [C4][Cm]; prev_mean / prev_invstd f32 [C4]. */
int rn_pw_dgrad_resid_sums_walkers(int64_t M, int Cm, int C4);
int rn_pw_dgrad_resid_sums(int64_t M, int Cm, int C4, int dtype, const void *dz1, const void *w1t, const void *resid, const uint8_t *rbits,
                           int res_stride, int res_h, int res_w, const void *prev_z3, const uint8_t *prev_bits, const float *prev_mean,
                           const float *prev_invstd, void *dx, float *partial, void *stream);
/* conv3 of a bottleneck, forward, on the row-tile walker kernel of rn_pw_dgrad_resid_sums (ABI 9): z3 [M][C4] = relu(fma(z2, fa, fb)) . w3^T
 * with fwd_coef f32 [2][Cm] = (fa | fb) bn2's forward coefficients, partial f32 [walkers][2][C4] = column sums / sums of squares of z3 as
 * stored -- rn_pw_conv_forward(z2, w3, prologue RN_PW_PRO_AFFINE_RELU, epilogue RN_PW_EPI_STATS), z3 bit for bit.  Cm = 64 / 128, C4 % 128 == 0. */
int rn_pw_conv3_forward_walkers(int64_t M, int Cm, int C4);
int rn_pw_conv3_forward(int64_t M, int Cm, int C4, int dtype, const void *z2, const float *fwd_coef, const void *w3, void *z3, float *partial,
                        void *stream);
int rn_pw_conv3_backward_walkers(int64_t M, int Cm, int C4);
size_t rn_pw_conv3_backward_workspace_bytes(int64_t M, int Cm, int C4);
int rn_pw_conv3_backward(int64_t M, int Cm, int C4, int dtype, const void *g, const void *z3, const uint8_t *bits, const float *a3,
                         const float *k0, const float *k1, const void *w3t, const void *z2, const float *ea, const float *eb,
                         const float *emean, const float *einv, void *dy2, float *partial_bn, void *workspace, size_t workspace_bytes,
                         int *splits, void *stream);


/* ---- the ResNet stem convolution (7x7 / stride 2 / pad 3, 3 -> 64 channels, bias-free) -------------------------------
 * Replaces `self.conv1` of /root/reference/retinanet/backbone.py:152 (applied at :246) for bf16 channels-last tensors.
 *   x  [B][H][W][3] bf16 (channels-last memory of [B, 3, H, W]),  w [64][7][7][3] bf16 (channels-last memory of [64, 3, 7, 7]),
 *   y  [B][Ho][Wo][64] bf16, Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1.
 * xp: scratch of rn_stem_padded_bytes(B, H, W) bytes (the zero-bordered NHWC4 copy of x the MFMA kernel reads),
 * wk: scratch of 64 * 7 * 32 * 2 bytes (the weights in the kernel's k order); both are rewritten by every call.
 * partial (nullable): f32 [rn_stem_partial_rows(B, H, W)][2][64] = per-workgroup sum / sum of squares of the STORED
 * (bf16-rounded) outputs, the input of rn_bn_stats_finalize for the BatchNorm that follows (backbone.py:153). */
size_t rn_stem_padded_bytes(int B, int H, int W);
int rn_stem_partial_rows(int B, int H, int W);
int rn_stem_conv_forward(const void *x, const void *w, void *xp, void *wk, void *y, float *partial, int dtype, int B, int H, int W,
                         void *stream);
/* Its weight gradient: dw [64][7][7][3] bf16 (channels-last memory of a [64, 3, 7, 7] gradient) from g = the gradient at the
 * conv output [B][Ho][Wo][64] bf16 and the padded copy xp that rn_stem_conv_forward wrote for the same x (the image itself is
 * not read again).  workspace: rn_stem_wgrad_workspace_bytes(B, H, W) bytes of f32 partials. */
size_t rn_stem_wgrad_workspace_bytes(int B, int H, int W);
int rn_stem_conv_wgrad(const void *g, const void *xp, void *dw, int dtype, int B, int H, int W, void *workspace,
                       size_t workspace_bytes, void *stream);
/* The same with the BatchNorm + ReLU backward of the layer above in the operand load (ABI 9): g is the gradient at the output of
 * relu(bn1(conv)) -- /root/reference/retinanet/backbone.py:246-248 -- z the conv output [B][Ho][Wo][64], coef3 f32 [3][64] = (a | k0 | k1)
 * of rn_bn_bwd_reduce / rn_bn_bwd_finalize, fwd_coef f32 [2][64] the forward (a | b); the conv-output gradient
 * round(fma(a, g * [fma(z, fa, fb) alive], fma(k1, z, k0))) -- what rn_bn_bwd_apply(relu_mode 2) would have stored -- is formed per
 * staged row and never written (dw is bit-identical to the two-launch form). */
int rn_stem_conv_wgrad_bn(const void *g, const void *z, const float *coef3, const float *fwd_coef, const void *xp, void *dw, int dtype,
                          int B, int H, int W, void *workspace, size_t workspace_bytes, void *stream);

/* Weight gradient of a NARROW 3x3 / stride-1 / pad-1 convolution without bias -- conv2 of the layer1 / layer2 / layer4 bottlenecks
 * (retinanet/backbone.py:112,128, autograd's weight gradient of F.conv2d there): Cout, Cin multiples of 64, bf16 channels-last.
 *   g  [N][H][W][Cout] gradient at the conv output,  x [N][H][W][Cin] the conv input,
 *   dw [Cout][3][3][Cin] bf16 (channels-last memory of a [Cout, Cin, 3, 3] gradient), fp32 accumulation.
 * All nine taps of a 64 x 64 block of dw are held by one workgroup (csrc/wgrad3x3.hip); zero_page: >= 128 zero bytes on the device
 * (what pixels outside the image read); workspace: f32 partials, rn_conv3x3_wgrad_narrow_workspace_bytes(Cout, Cin) bytes. */
size_t rn_conv3x3_wgrad_narrow_workspace_bytes(int Cout, int Cin);
int rn_conv3x3_wgrad_narrow(const void *g, const void *x, void *dw, int dtype, int N, int H, int W, int Cout, int Cin,
                            const void *zero_page, void *workspace, size_t workspace_bytes, void *stream);

/* The forward product of the same NARROW 3x3 / stride-1 / pad-1 convolution at C = 64 input and output channels (conv2 of the layer1
 * bottlenecks, retinanet/backbone.py:112,128 -- what F.conv2d(x, w, None, 1, 1) returns there), bf16 channels-last, fp32 accumulation:
 *   x, y [N][H][W][64],  w [64][3][3][64] (channels-last memory of a [64, 64, 3, 3] weight).
 * Issued with the tap-reversed, role-swapped weight (rn_conv3x3_levels_dgrad_weight's layout) it is the data gradient of that
 * convolution.  bias (f32 [64], may be NULL) and relu: y = act(conv + bias) in the epilogue -- inference with the BatchNorm folded in.
 * Weights in registers, input rows in an LDS ring (csrc/narrow3x3.hip); zero_page: >= 128 zero bytes on the device (what
 * pixels outside the image read).  RN_EUNSUPPORTED for other C or dtypes. */
int rn_conv3x3_narrow_forward(const void *x, const void *w, const float *bias, void *y, int dtype, int N, int H, int W, int C, int relu,
                              const void *zero_page, void *stream);

/* n device-to-device copies (dsts[i] <- srcs[i], nbytes[i] bytes, non-overlapping) in one launch per 64: the inputs of a step
 * into the static buffers of its captured hipGraph (graph.CapturedTrainStep).  srcs / dsts / nbytes are HOST arrays. */
int rn_copy_many(const void *const *srcs, void *const *dsts, const int64_t *nbytes, int n, void *stream);
/* n widening copies dsts[i] (f32) <- srcs[i] (src_dtype: RN_BF16 or RN_F16), counts[i] elements each, one launch per 64: the
 * gather of 16-bit parameter gradients into the fp32 buckets of the gradient exchange (no reference analogue: Lightning's DDP
 * exchanges fp32 gradients of fp32 parameters).  HOST arrays. */
int rn_cast_many_to_f32(const void *const *srcs, void *const *dsts, const int64_t *counts, int n, int src_dtype, void *stream);

/* dsts[i] [cols[i]][rows[i]] = transpose of srcs[i] [rows[i]][cols[i]], 16-bit elements, n <= 16 matrices in one launch (the
 * data-gradient weights of a bottleneck's 1x1 convolutions).  srcs / dsts / rows / cols are HOST arrays. */
int rn_transpose_many(const void *const *srcs, void *const *dsts, const int *rows, const int *cols, int n, void *stream);
/* The weight rn_conv3x3_levels_to_canvas expects, from the forward weight w [Cout][3][3][Cin] (16-bit): out [Cin][3][3][Kpad],
 * taps reversed, channel roles swapped, contraction axis laid out as that kernel walks it (the piece that straddles the end
 * of a Cout % 8 != 0 row carries the row's last 8 channels, zero weights on the repeated ones; zeros up to Kpad). */
int rn_conv3x3_levels_dgrad_weight(const void *w, void *out, int Cout, int Cin, int Kpad, void *stream);

/* ---- K4 decode_clip ---------------------------------------------------------
 * Replaces activ_2_bbox, retinanet/box_utils.py:37-48 (including its use of
 * dx,dy for the sizes, :46) and torchvision clip_boxes_to_image at
 * retinanet/models.py:189.  deltas [B][A][4] (dtype), image_hw i32[B][2] = resized
 * unpadded (h, w) per image, NULL = no clipping.  out f32[B][A][4]. */
int rn_decode_clip(const void *deltas, int dtype, int B, int64_t A,
                   const float *anchors, int64_t anchor_bstride, const int32_t *image_hw,
                   const float reg_w[4], float *out, void *stream);

/* ---- conv epilogue: bias (+ ReLU) (+ position mask), channels-last ------------------------------
 * y = mask[m % HW] ? act(x + bias[c]) : 0 on [M][C] activations (C % 8 == 0), act = ReLU when relu != 0;
 * replaces the bias add + nn.ReLU pairs of the reference's head towers (retinanet/layers.py:143-171,
 * :213-241) and their backward (ReLU mask + bias-gradient reduction, one pass).  mask: u8[HW] shared by
 * all images, NULL = keep everything (used to zero the gaps of a packed level canvas).  Backward:
 * dx = dy where y > 0 (relu) and mask, else 0; dbias[c] = sum_m dx[m][c] (f32, deterministic).  dx may be
 * NULL when relu == 0 and mask == NULL (dx == dy).  workspace: rn_bn_workspace_bytes(C). */
int rn_bias_act_forward(const void *x, const float *bias, const uint8_t *mask, void *y, int dtype,
                        int64_t M, int C, int64_t HW, int relu, void *stream);
int rn_bias_act_backward(const void *dy, const void *y, const uint8_t *mask, void *dx, float *dbias,
                         int dtype, int64_t M, int C, int64_t HW, int relu,
                         void *workspace, size_t workspace_bytes, void *stream);

/* ---- 3x3 conv of the head towers on the packed level canvas (MFMA implicit GEMM) -------------------
 * y = mask * act( conv3x3(x, w, stride 1, pad 1) + bias ) for the 3x3 conv + ReLU pairs of the reference's
 * towers (retinanet/layers.py:143-171, :213-241), on a canvas that carries a one-pixel ZERO border:
 * x, y: [M = N*Hp*Wp][C] bf16 (channels-last [N, C, Hp, Wp]); w: [Cout][3][3][Cin] bf16 (the channels-last
 * memory of a [Cout, Cin, 3, 3] weight); bias f32[Cout] or NULL; mask u8[HWp = Hp*Wp] or NULL -- it must be 0 on
 * the border (border outputs are computed from wrapped neighbours and are only correct as zeros).  Wp = row
 * pitch in positions.  dtype: RN_BF16 or RN_F16 (every conv3x3 / stem / narrow entry point of this library: the same kernels instantiated on v_mfma_..._bf16 / _f16); Cin % 64 == 0, Cout % 256 == 0 (else RN_EUNSUPPORTED: the caller
 * keeps its MIOpen path).  The data gradient is the same call with the taps reversed and the channel roles
 * swapped (w' = w.flip(2, 3).transpose(0, 1)), relu = 0, bias = NULL. */
int rn_conv3x3_canvas(const void *x, const void *w, const float *bias, const uint8_t *mask, void *y, int dtype,
                      int64_t M, int64_t HWp, int Wp, int Cin, int Cout, int relu, void *stream);
/* P <= 4 convolutions of identical geometry in ONE launch (HOST arrays of device pointers; biases nullable as a
 * whole): the cls and box towers run the same shapes side by side, and their tiles together fill the chip's
 * workgroup waves better than two launches (2 x 749 tiles of two-image sheets at B = 8: 6 rounds of 256 CUs). */
int rn_conv3x3_canvas_batched(const void *const *xs, const void *const *ws, const float *const *biases,
                              const uint8_t *mask, void *const *ys, int P, int dtype, int64_t M, int64_t HWp,
                              int Wp, int Cin, int Cout, int relu, void *stream);
/* y = conv3x3(x, w[.., 0:C]) + conv3x3(x2, w[.., C:2C]) on the canvas in ONE accumulator (ABI 9): the contraction walks the C channels of x, then
 * those of x2, against w [Cout][3][3][2 C] (no bias, no ReLU; `mask` as above).  The two head towers read the same FPN canvas in their
 * first layer (/root/reference/retinanet/layers.py:163-167, 235-251), so its gradient is the SUM of their first-layer data gradients:
 * this is that sum without the two separate outputs and autograd's add pass over them. */
int rn_conv3x3_canvas_sum2(const void *x, const void *x2, const void *w, const uint8_t *mask, void *y, int dtype, int64_t M, int64_t HWp,
                           int Wp, int C, int Cout, void *stream);

/* Weight gradient of the canvas convolution for P <= 4 problems (256 -> 256 channels, bf16):
 *   dw[p][n][3][3][c] = sum_m g[p][m][n] * x[p][m + tap offset][c]
 * g: gradient at the conv OUTPUT (already masked: zero on border / gap positions), x: the conv input canvas; both
 * [M][256] bf16; dw: [256][3][3][256] bf16 (channels-last memory of a [Cout, Cin, 3, 3] weight gradient).  A position-
 * contraction MFMA GEMM (transposed LDS fragment reads) split over the positions, plus a reduction of the splits;
 * workspace: rn_conv3x3_wgrad_workspace_bytes(P, M); zeros: >= 256 bytes of zeros. */
size_t rn_conv3x3_wgrad_workspace_bytes(int P, int64_t M);
int rn_conv3x3_canvas_wgrad_batched(const void *const *gs, const void *const *xs, void *const *dws, int P, int dtype,
                                    int64_t M, int Wp, int Cin, int Cout, const void *zeros, void *workspace,
                                    size_t workspace_bytes, void *stream);

/* The same kernels on plain DENSE tensors, for P <= 4 convolutions that each have their own geometry and weights, in
 * one launch (the FPN's 3x3 output convolutions on P3 / P4 / P5 -- /root/reference/retinanet/layers.py:34-38 applied at
 * :62-64 -- which the reference runs level by level through nn.Conv2d):
 *   ys[p][n][y][x][co] = biases[p][co] + sum_{r,s,ci} xs[p][n][y + r - 1][x + s - 1][ci] * ws[p][co][r][s][ci]   (zero padding)
 * xs[p]: [N][hs[p]][wds[p]][Cin], ys[p]: [N][hs[p]][wds[p]][Cout], ws[p]: [Cout][3][3][Cin], all bf16 (channels-last
 * memory of NCHW tensors), biases[p]: f32 [Cout] or NULL (biases itself may be NULL).  Cin % 64 == 0, Cout % 256 == 0,
 * N * h * w < 2^22 per problem.  The row tiles of all problems form one grid; a tap that leaves its image reads `zeros`
 * (>= 16 bytes of zeros, 16-byte aligned).  The data gradient is the same call on the output gradients with the weights
 * of rn_conv3x3_dgrad_weight_batched and no bias. */
int rn_conv3x3_dense_batched(const void *const *xs, const void *const *ws, const float *const *biases, void *const *ys, int P,
                             int dtype, int N, const int *hs, const int *wds, int Cin, int Cout, const void *zeros,
                             void *stream);
/* One dense 3x3 / stride-1 / pad-1 convolution (no bias) whose K walk -- Cin / 64 channel chunks x 9 taps -- is cut into 2 or 3
 * contiguous ranges run by different workgroups (ABI 9): f32 partials in `workspace`, summed into y by a second launch.  For
 * convolutions with few 256-row tiles (conv2 of the layer4 bottlenecks, /root/reference/retinanet/backbone.py:112,128: 8 400 positions x
 * 512 channels = 66 workgroups of 72 K-tiles) this fills the chip.  _workspace_bytes() returns 0 when a split would not help (the
 * launch already covers half the CUs): the caller then uses rn_conv3x3_dense_batched or its own fallback.  Cin % 64 == 0, Cout % 256 == 0. */
size_t rn_conv3x3_dense_splitk_workspace_bytes(int N, int h, int w, int Cout);
int rn_conv3x3_dense_splitk(const void *x, const void *w, void *y, int dtype, int N, int h, int wd, int Cin, int Cout, const void *zeros,
                            void *workspace, size_t workspace_bytes, void *stream);
/* A dense 3x3 / stride-1 / pad-1 convolution (no bias), Cout a multiple of 128, on the band-staged kernel (ABI 9): per (channel chunk,
 * kernel row) ONE band of 258 consecutive positions feeds the three horizontal taps; taps that leave their image are zeroed at the
 * MFMA fragment.  conv2 of the layer2 bottlenecks (/root/reference/retinanet/backbone.py:112,128; 128 -> 128), forward and -- with the
 * flipped, transposed weights -- data gradient.  x [N][h][wd][Cin], w [Cout][3][3][Cin], y [N][h][wd][Cout]; Cin % 64 == 0. */
int rn_conv3x3_dense_band(const void *x, const void *w, void *y, int dtype, int N, int h, int wd, int Cin, int Cout, const void *zeros,
                          void *stream);
/* The same with the per-channel sums of the output AS STORED in the epilogue (no second pass over y): partial f32 [tiles][2][Cout] =
 * (sum y, sum y^2) per 256-position row tile, tiles = rn_conv3x3_dense_band_tiles(N, h, wd) -- what rn_bn_stats' first launch computes;
 * rn_bn_stats_finalize(partial, tiles, N*h*wd, ..) completes the statistics of the BatchNorm after the convolution (bn2 of a
 * bottleneck, /root/reference/retinanet/backbone.py:129).  y is bit-identical to rn_conv3x3_dense_band's. */
int rn_conv3x3_dense_band_tiles(int N, int h, int wd);
int rn_conv3x3_dense_band_stats(const void *x, const void *w, void *y, float *partial, int dtype, int N, int h, int wd, int Cin, int Cout,
                                const void *zeros, void *stream);
/* The same with ys[p] = relu(...) when `relu` (inference: conv2 of the layer3 bottlenecks with the folded BatchNorm as bias and the
 * ReLU of retinanet/backbone.py:132 in the epilogue). */
int rn_conv3x3_dense_batched_act(const void *const *xs, const void *const *ws, const float *const *biases, void *const *ys, int P,
                                 int dtype, int N, const int *hs, const int *wds, int Cin, int Cout, const void *zeros, int relu,
                                 void *stream);
/* Its weight gradient (256 -> 256): dws[p][co][3][3][ci] = sum over positions of gs[p][pos][co] * xs[p][pos + tap][ci];
 * every problem gets a number of position splits proportional to its size (one round of workgroups in total), a second
 * kernel sums the f32 partials.  workspace: rn_conv3x3_dense_wgrad_workspace_bytes(P); zeros: >= 256 bytes of zeros. */
size_t rn_conv3x3_dense_wgrad_workspace_bytes(int P);
int rn_conv3x3_dense_wgrad_batched(const void *const *gs, const void *const *xs, void *const *dws, int P, int dtype, int N,
                                   const int *hs, const int *wds, int Cin, int Cout, const void *zeros, void *workspace,
                                   size_t workspace_bytes, void *stream);

/* rn_conv3x3_canvas_batched that also writes, per problem, the ReLU bits of its outputs: relu_mask_outs[p] = [M][Cout / 8]
 * bytes, bit j of byte (m, c / 8) = [ys[p][m][c + j] > 0] (relu must be set; 16-byte aligned; NULL = plain
 * rn_conv3x3_canvas_batched). */
int rn_conv3x3_canvas_batched_ex(const void *const *xs, const void *const *ws, const float *const *biases,
                                 const uint8_t *mask, void *const *ys, uint8_t *const *relu_mask_outs, int P, int dtype, int64_t M,
                                 int64_t HWp, int Wp, int Cin, int Cout, int relu, void *stream);
/* Data gradient of P tower convs whose INPUT was the ReLU output of the layer below (retinanet/layers.py:143-171: conv +
 * ReLU pairs), with that layer's ReLU backward and bias gradient fused into the epilogue:
 *   ys[p] = conv3x3(gs[p], ws[p]) * mask * relu_bits[p],   dbiases[p][c] = sum over positions of ys[p][., c]
 * gs: gradients at the conv outputs [M][Cin]; ws: the forward weights with taps reversed and channel roles swapped,
 * [Cout][3][3][Cin]; relu_masks[p]: the ReLU bits of the conv's forward INPUT (= the layer below's relu_mask_outs),
 * [M][Cout / 8]; dbiases: f32 [Cout] each.  ys[p] is the gradient at the PRE-activation of the layer below (what
 * rn_bias_act_backward would produce from the plain data gradient), so that layer needs no pass of its own.  Column sums:
 * one partial row per 256-position tile in `workspace` (rn_conv3x3_colsum_workspace_bytes), reduced in double in a fixed
 * order. */
size_t rn_conv3x3_colsum_workspace_bytes(int P, int64_t M, int Cout);
int rn_conv3x3_canvas_dgrad_relu_batched(const void *const *gs, const void *const *ws, const uint8_t *const *relu_masks,
                                         const uint8_t *mask, void *const *ys, float *const *dbiases, int P, int dtype,
                                         int64_t M, int64_t HWp, int Wp, int Cin, int Cout, void *workspace,
                                         size_t workspace_bytes, void *stream);
/* The data gradient's weights of P convs: outs[p] [Cin][3][3][Cout] = ws[p] [Cout][3][3][Cin] with taps reversed and channel
 * roles swapped (16-bit elements, Cout % 32 == Cin % 32 == 0); one launch. */
int rn_conv3x3_dgrad_weight_batched(const void *const *ws, void *const *outs, int P, int Cout, int Cin, void *stream);
/* The same for n convolutions of different widths (couts[i], cins[i], multiples of 32) in one launch per 48: a training step
 * flips the weights of all its 3x3 convolutions at once (pytorch_retinanet_amd/biasact.py: refresh_dgrad_weights). */
int rn_conv3x3_dgrad_weight_many(const void *const *ws, void *const *outs, const int *couts, const int *cins, int n, void *stream);
/* out[c] = sum over l and rows of xs[l][row][c] for L <= 6 dense row-major bf16 / f16 tensors [rows[l]][C] (C even: the
 * 810-channel logit gradients, the 36-channel box-delta gradients): the bias gradient of the class- / box-output conv.
 * f32 out[C], deterministic. */
size_t rn_colsum_rows_workspace_bytes(int L, int C);
int rn_colsum_rows(const void *const *xs, const int64_t *rows, int L, int C, int dtype, float *out, void *workspace,
                   size_t workspace_bytes, void *stream);

/* ---- class-output conv on the canvas with DENSE per-level results -----------------------------------------------
 * The last 3x3 conv of the classification subnet (retinanet/layers.py:163-167: 256 -> 9*K channels) reads the tower
 * output where it lies -- on the zero-bordered canvas -- and writes, per pyramid level, the dense channels-last
 * tensor ys[l] = [n_images][h_l][w_l][Cout] bf16, which IS the [N][h*w*9][K] logits tensor of retinanet/layers.py:189-191
 * that rn_loss_fwd_bwd_levels / rn_detect_levels stream (no dead classes, no unpack copy).  Also used for the 36-channel
 * box-output conv (layers.py:235-251); a ragged last tile of <= 64 output channels runs on a narrow kernel variant.
 * Canvas layout: the canvas is N sheets of [Hp][Wp] positions; a sheet carries `slots` images (image = sheet * slots +
 * slot; the last sheet may have unused slots when n_images is not a multiple of slots), each pyramid level of each slot in
 * its own rectangle with at least one empty row / column around it.  map (int32, DEVICE memory, [Hp * Wp]) says what lies
 * at every position of a sheet: -1 = border / gap, else (slot << 28) | (tensor << 24) | (y * w + x), i.e. position (y, x)
 * of that slot's image in per-level tensor `tensor` (T <= 6 tensors, hw[t] = h_t * w_t < 2^24).  Entries that point
 * outside their tensor are treated as gaps.
 * Cout: any even number (rows of Cout elements start on 4-byte boundaries); Cin % 64 == 0; w [Cout][3][3][Cin] bf16;
 * bias f32[Cout] or NULL; zeros: >= 256 bytes of zeros, 16-byte aligned. */
typedef struct rn_canvas_layout { const int32_t *map; int32_t slots, n_images, T; int32_t hw[6]; } rn_canvas_layout;
/* Pack (to_canvas != 0) the per-level tensors levels[t] = [n_images][h_t][w_t][C] onto the canvas [N sheets][Hp][Wp][C] with
 * zeros in the gaps, or unpack the canvas into them (to_canvas == 0), as `layout` places them; bf16 / f16, C even; one launch. */
int rn_canvas_pack(void *const *levels, const rn_canvas_layout *layout, void *canvas, int dtype, int N, int Hp, int Wp, int C,
                   int to_canvas, void *stream);
int rn_conv3x3_canvas_to_levels(const void *x, const void *w, const float *bias, const rn_canvas_layout *layout,
                                void *const *ys, int dtype, int N, int Hp, int Wp, int Cin, int Cout,
                                const void *zeros, void *stream);
/* Its data gradient: y[M = N*Hp*Wp][Cout] (canvas, masked) = conv of the dense per-level gradients gs[l] =
 * [N][h_l][w_l][row_elems] with w [Cout][3][3][Kpad] bf16 = the forward weight with taps reversed and channel roles
 * swapped, its contraction axis laid out in Kpad (next multiple of 64) slots: slot k = channel k for k < e =
 * row_elems - row_elems % 8; when row_elems % 8 != 0 the kernel fetches the LAST 8 channels of a row for slots e .. e+7
 * (it never reads past a row), so those slots carry channel row_elems - 8 + (k - e) with ZERO weight on the ones that
 * repeat (< e); every later slot is zero.  row_elems even and >= 8. */
int rn_conv3x3_levels_to_canvas(const void *const *gs, const rn_canvas_layout *layout, int row_elems,
                                const void *w, const uint8_t *mask, void *y, int dtype, int N, int Hp, int Wp,
                                int Kpad, int Cout, const void *zeros, void *stream);
/* The same when the conv's INPUT was a ReLU output that feeds nothing else (the last tower layer, layers.py:147-171 /
 * :217-241, in front of class_subnet_output / box_subnet_output): y is also multiplied by relu_mask ([M][Cout / 8] bytes, the
 * bits rn_conv3x3_canvas_batched_ex wrote for that activation) and its column sums -- the tower layer's bias gradient -- go to
 * dbias f32[Cout]; workspace: rn_conv3x3_colsum_workspace_bytes(1, N * Hp * Wp, Cout). */
int rn_conv3x3_levels_to_canvas_relu(const void *const *gs, const rn_canvas_layout *layout, int row_elems, const void *w,
                                     const uint8_t *relu_mask, const uint8_t *mask, void *y, float *dbias, int dtype, int N, int Hp,
                                     int Wp, int Kpad, int Cout, const void *zeros, void *workspace, size_t workspace_bytes,
                                     void *stream);
/* Its weight gradient: dw [row_elems][3][3][Cin = 256] bf16 = sum over canvas positions of gs (gathered) x the
 * tapped canvas input x [M][256].  workspace: rn_conv3x3_wgrad_workspace_bytes((row_elems + 255) / 256, M). */
int rn_conv3x3_levels_wgrad(const void *const *gs, const rn_canvas_layout *layout, int row_elems,
                            const void *x, void *dw, int dtype, int N, int Hp, int Wp, int Cin,
                            const void *zeros, void *workspace, size_t workspace_bytes, void *stream);

/* ---- stem max pooling, channels-last, no index tensor -------------------------------------------------
 * nn.MaxPool2d(kernel_size=3, stride=2, padding=1) of the reference's stem (retinanet/backbone.py:251) on
 * [N][H][W][C] activations (C % 8 == 0), y: [N][(H-1)/2+1][(W-1)/2+1][C].  argmax (u8, shape of y, nullable for
 * inference): position 0..8 of the maximum inside its window with PyTorch's scan rule (first maximum; the last
 * NaN wins) -- one byte per element instead of PyTorch's int64 index; the backward gathers from it. */
int rn_maxpool3x3s2_forward(const void *x, void *y, uint8_t *argmax, int dtype, int N, int H, int W, int C, void *stream);
/* The same pooling of relu(x * coef[0][c] + coef[1][c]) rounded to dtype -- the stem's BatchNorm + ReLU (backbone.py:247-251)
 * applied on the fly, its output never written; y and argmax equal those of rn_bn_apply followed by rn_maxpool3x3s2_forward. */
int rn_bn_relu_maxpool3x3s2_forward(const void *x, const float *coef, void *y, uint8_t *argmax, int dtype, int N, int H, int W, int C,
                                    void *stream);
int rn_maxpool3x3s2_backward(const uint8_t *argmax, const void *dy, void *dx, int dtype,
                             int N, int H, int W, int C, void *stream);
/* The same for the stem (ABI 9), where the pooled tensor was relu(bn1(z)) (/root/reference/retinanet/backbone.py:246-251): besides dx the kernel
 * takes the two sums of bn1's backward over the gradient it has just formed -- partial f32 [rows][2][C], rows =
 * rn_maxpool3x3s2_backward_bn_rows(N, H, W, C): what rn_bn_bwd_reduce(dx, NULL, z, .., fwd_coef, training, relu = 1) sums before it finalizes
 * (g' = dx * [fma(z, fa, fb) alive]; sum g', sum g' * (z - mean) * invstd) -- so that pass does not re-read dx; complete with
 * rn_bn_bwd_finalize(partial, rows, ..).  fwd_coef f32 [2][C] = (a | b) of the forward.  C % 8 == 0 and (C / 8) | 256. */
int rn_maxpool3x3s2_backward_bn_rows(int N, int H, int W, int C);
int rn_maxpool3x3s2_backward_bn(const uint8_t *argmax, const void *dy, const void *z, const float *fwd_coef, const float *mean,
                                const float *invstd, void *dx, float *partial, int dtype, int N, int H, int W, int C, void *stream);

/* ---- FPN top-down step, channels-last ----------------------------------------------------------------------
 * out = lat + nearest_upsample_2x(top)   (retinanet/layers.py:36,52-53: lateral 1x1 conv + nn.Upsample(scale_factor=2) of the
 * level above): lat, out [N][H][W][C], top [N][H/2][W/2][C] (H, W even, C % 8 == 0), one pass.  Backward for `top`:
 * dtop = the 2 x 2 block sums of g (f32 accumulation); the lateral's gradient is g itself. */
int rn_fpn_add_upsample2x(const void *lat, const void *top, void *out, int dtype, int N, int H, int W, int C, void *stream);
int rn_fpn_upsample2x_backward(const void *g, void *dtop, int dtype, int N, int Ht, int Wt, int C, void *stream);

/* ---- optimizer step: SGD on fp32 masters with a bf16 working copy, multi-tensor ------------------------------
 * The update torch.optim.SGD performs (the reference's optimizer, hparams.yaml:63-68), in its order, in fp32:
 *   g = grad + weight_decay * w;  buf = first_step ? g : momentum * buf + (1 - dampening) * g;
 *   g = nesterov ? g + momentum * buf : buf;  w -= lr * g
 * for n_tensors tensors in one launch per 48 (HOST arrays of device pointers / element counts).  params16[i]
 * (nullable): bf16 copy of tensor i, rewritten as bf16(w) -- the conv weights the bf16 forward consumes, so that
 * neither autocast's per-step weight casts nor the bf16 -> fp32 gradient casts are needed; when grads16 != 0 the
 * gradient of a tensor WITH a 16-bit copy is bf16, every other gradient is f32. */
int rn_sgd_master_step(float *const *masters, float *const *momenta, const void *const *grads, void *const *params16,
                       const int64_t *numels, int n_tensors, int grads16, float lr, float momentum, float dampening,
                       float weight_decay, int nesterov, int first_step, void *stream);
/* The same step for fp16 working copies and for fp16 autocast under a loss scale (ABI 8): dtype16 = RN_BF16 or RN_F16 is the type of
 * params16[i] (and of the 16-bit gradients); grad_scale / found_inf (nullable DEVICE scalars, f32) are what torch.amp.GradScaler hands
 * an optimizer with `_step_supports_amp_scaling`: every gradient is divided by grad_scale[0] first, and when found_inf[0] != 0 the
 * launch changes nothing (no parameter, no momentum buffer) -- no host synchronisation, so the step stays capturable. */
int rn_sgd_master_step_ex(float *const *masters, float *const *momenta, const void *const *grads, void *const *params16,
                          const int64_t *numels, int n_tensors, int grads16, int dtype16, float lr, float momentum, float dampening,
                          float weight_decay, int nesterov, int first_step, const float *grad_scale, const float *found_inf, void *stream);

/* ---- T1 transform (normalise + resize + pad + batch) -------------------------------------------
 * Replaces torchvision's GeneralizedRCNNTransform as the reference runs it at
 * retinanet/models.py:116 (construction), :262 and :279 (calls): per image (x - mean) / std, bilinear
 * resize (align_corners=False, scale recomputed from the integer sizes) to out_hw[b], zero padding
 * into one batch [B][3][Hp][Wp].  images: HOST array of B device pointers, each f32 [3][h][w]
 * contiguous; in_hw / out_hw: HOST i32[B][2] = (h, w) before / after the resize (the caller computes
 * the sizes exactly as torchvision does, floor(size * scale) in double); mean / std: HOST f32[3].
 * out: dtype out_dtype, NCHW or (channels_last != 0) NHWC memory order; Wp % 4 == 0. */
int rn_transform_batch(const void *const *images, const int32_t *in_hw, const int32_t *out_hw, int B,
                       const float mean[3], const float std[3], int Hp, int Wp,
                       void *out, int out_dtype, int channels_last, void *stream);

/* ---- K6 nms (op boundary) ---------------------------------------------------
 * Replaces torchvision.ops.nms as called at retinanet/models.py:210, batched over
 * S independent segments: seg_off i32[S+1]; boxes f32[N][4], scores f32[N].
 * keep i64[N]: for segment s, keep[seg_off[s] .. +keep_count[s]) are the kept
 * indices RELATIVE to the segment start, in descending score order (stable).
 * workspace: rn_nms_workspace_bytes(N, S). */
size_t rn_nms_workspace_bytes(int64_t N, int S);
int rn_nms_segments(const float *boxes, const float *scores, const int32_t *seg_off, int S, int64_t N,
                    float iou_thr, int64_t *keep, int32_t *keep_count,
                    void *workspace, size_t workspace_bytes, void *stream);

/* ---- K4-K7 detect -----------------------------------------------------------
 * Replaces Retinanet.process_detections, retinanet/models.py:160-243: sigmoid,
 * decode + clip, per class {score > thr, remove_small_boxes, nms}, class-major
 * concat, labels + 1, stable sort by score, first max_det.
 * Outputs are padded to max_det rows per image: out_boxes f32[B][max_det][4],
 * out_scores f32[B][max_det], out_labels i64[B][max_det], out_count i32[B].
 * max_candidates = per-image capacity for (anchor, class) pairs passing the score
 * and size filters; if exceeded, out_status[b] (i32[B]) is set to 1 and that
 * image's result is truncated -- the caller re-runs with a larger capacity
 * (A*K always suffices). */
size_t rn_detect_workspace_bytes(int B, int64_t A, int K, int64_t max_candidates);
int rn_detect(const void *cls, const void *deltas, int dtype, int B, int64_t A, int K,
              const float *anchors, int64_t anchor_bstride, const int32_t *image_hw,
              const rn_detect_params *params, int64_t max_candidates,
              float *out_boxes, float *out_scores, int64_t *out_labels, int32_t *out_count,
              int32_t *out_status, void *workspace, size_t workspace_bytes, void *stream);

/* Same chain on the per-level head outputs: cls_levels[l] [B][A_l][K], box_levels[l] [B][A_l][4]
 * (level_anchors[l] = A_l, levels in anchor order, L <= RN_MAX_LEVELS), i.e. the conv outputs of
 * retinanet/layers.py:189-195 / :253-259 before their torch.cat.  Results are
 * identical to rn_detect on the concatenation; workspace: rn_detect_workspace_bytes(B, sum A_l, K, max_candidates). */
int rn_detect_levels(const void *const *cls_levels, const void *const *box_levels,
                     const int64_t *level_anchors, int L, int dtype, int B, int K,
                     const float *anchors, int64_t anchor_bstride, const int32_t *image_hw,
                     const rn_detect_params *params, int64_t max_candidates,
                     float *out_boxes, float *out_scores, int64_t *out_labels, int32_t *out_count,
                     int32_t *out_status, void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RETINANET_HIP_H */
