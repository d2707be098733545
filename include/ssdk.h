/*
 * ssdk.h -- C ABI of libssdk, the MI355X (gfx950) implementation of the single-shot-detection hot path.
 *
 * One entry point per reference function on the path; the comment on each declaration cites the
 * reference interface it replaces (paths relative to the reference repository root).  Rules of the boundary:
 *   - every pointer marked DEV is device memory owned by the caller (torch); the library never allocates or
 *     frees device memory on the hot path and keeps no pointer past the call; workspaces are caller-provided,
 *     sized by the matching *_workspace_bytes() query;
 *   - every launch goes to the hipStream_t passed as `stream` (void*; NULL = the null stream); no call
 *     synchronises the device;
 *   - return value: 0 ok, <0 invalid argument (SSDK_E_*), >0 a hipError_t; no exception crosses the boundary;
 *     ssdk_last_error_string() gives the thread's last message;
 *   - fp32 everywhere unless stated; layouts are the reference's: scores [B, A*C] anchor-major/class-minor,
 *     locs [B, A*4], anchors/priors [A,4] = (cx, cy, w, h) in pixels, target [B, A, 6] =
 *     (x1, y1, x2, y2, class, score), ground-truth rows (x1, y1, x2, y2, class, score[, difficult]).
 */
#ifndef SSDK_H_
#define SSDK_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSDK_VERSION 114

#define SSDK_OK 0
#define SSDK_E_INVALID (-1)   /* bad argument / shape */
#define SSDK_E_WORKSPACE (-2) /* workspace missing or too small */
#define SSDK_E_UNSUPPORTED (-3)
#define SSDK_E_STREAMK_TIMEOUT (-4) /* a stream-K head GEMM gave up on a parked partial tile (see ssdk_heads_fwd_timeouts) */

/* detection/matcher.py:4-5 */
#define SSDK_NOT_MATCHED (-2)
#define SSDK_IGNORE (-1)

/* classification loss kinds (bf/modules/losses.py) */
#define SSDK_CLS_CROSS_ENTROPY 0 /* torch.nn.CrossEntropyLoss(reduction='sum', ignore_index=-1), losses.py:4 */
#define SSDK_CLS_SIGMOID_FOCAL 1 /* SigmoidFocalLoss, losses.py:34-54 */
#define SSDK_CLS_SOFTMAX_FOCAL 2 /* SoftmaxFocalLoss, losses.py:56-78 */
#define SSDK_CLS_CE_SOFT 3       /* CrossEntropyWithSoftTargetsLoss, losses.py:80-93 (target of multibox_loss.py:68-71) */
#define SSDK_CLS_BCE_SOFT 4      /* BinaryCrossEntropyWithSoftTargetsLoss, losses.py:95-106 (target of multibox_loss.py:64-67) */
/* localisation loss kinds */
#define SSDK_LOC_SMOOTH_L1 0 /* torch.nn.SmoothL1Loss(reduction='sum') on box_coder-encoded targets, multibox_loss.py:81-86 */
#define SSDK_LOC_GIOU 1      /* GeneralizedIoULoss, losses.py:109-114, on decoded corners (multibox_loss.py:77-79) */

int ssdk_version(void);

/*
 * bf/utils/box_utils.py:16-194 as a callable surface (every array DEV fp32; boxes 16-byte aligned rows of 4; no workspace unless said).
 *   ssdk_box_to_corners    :16-23   [cx,cy,w,h] -> [c - wh/2, c + wh/2]                                  box, out [n,4] (out may be box)
 *   ssdk_box_to_centroids  :25-36   inplace_form = 0: [(max+min)/2, max-min] (:36); != 0: wh = max-min, c = min + wh/2 (:33-34) -- the two
 *                                   forms round the centre differently, both as the reference does
 *   ssdk_box_area          :38-46   clamp(x2-x1, 0) * clamp(y2-y1, 0)                                     out [n]
 *   ssdk_box_intersection  :49-80   cat([max of the min corners, min of the max corners]); cartesian != 0: out [na,nb,4], else na == nb and
 *                                   out [na,4]; zero_incorrect != 0: rows with max < min in any coordinate become 0
 *   ssdk_box_iou           :83-101 (generalized = 0) / :104-143 (generalized != 0): cartesian != 0: out [na,nb], else out [na].  No +1, no
 *                                   epsilon: two degenerate boxes give NaN like the reference.  The [G,A] matrix ssdk_match_per_prediction takes.
 *   ssdk_nms               :145-194 ONE problem (n <= 65536; the batched fused path is ssdk_postprocess): max_per_class in 1..n-1 first keeps
 *                                   the max_per_class best scores (:186-188 topk(sorted=False): the reference defines the SET; here it is kept
 *                                   in descending score order, ties by ascending index); then soft == 0: hard NMS per torchvision.ops.nms's
 *                                   documented contract (:193; PARITY UNPINNED: torchvision is not vendored), soft != 0: _soft_nms (:145-163,
 *                                   incl. its loop condition mask.nonzero().sum()).  picked DEV int64 [>= min(n, cap)]: positions, in pick
 *                                   order, in the array the NMS ran on (the input, or the top-k subset when a cap applied); picked_boxes
 *                                   DEV [.,4] / picked_scores DEV [.] (NULL: not wanted): boxes[picked] / scores[picked]; count DEV int32.
 */
int ssdk_box_to_corners(const float* box, float* out, long long n, void* stream);
int ssdk_box_to_centroids(const float* box, float* out, long long n, int inplace_form, void* stream);
int ssdk_box_area(const float* box, float* out, long long n, void* stream);
int ssdk_box_intersection(const float* a, int na, const float* b, int nb, int cartesian, int zero_incorrect, float* out, void* stream);
int ssdk_box_iou(const float* a, int na, const float* b, int nb, int cartesian, int generalized, float* out, void* stream);
size_t ssdk_nms_workspace_bytes(int n);
int ssdk_nms(const float* boxes, const float* scores, int n, float overlap_threshold, float score_threshold, int max_per_class, int soft,
             float sigma, long long* picked, float* picked_boxes, float* picked_scores, int* count, void* workspace, size_t workspace_bytes,
             void* stream);

/*
 * DETERMINISTIC MODE (process-wide; default off, or SSDK_DETERMINISTIC=1 in the environment).  The reference runs every GPU job with
 * torch.backends.cudnn.deterministic = True / benchmark = False (bf/training/env.py:74-76).  Off, several kernels add fp32 partial
 * results with atomics in the order the hardware happens to retire them: split-K convolutions (forward too), the scatter form of the
 * data gradients, the K-split weight gradients, the bias-gradient column sums.  On, every fp32 sum is taken in an order fixed by the
 * launch: convolutions are not split over K with atomics (stream-K's parked partial tiles are already added in range order), the data
 * gradient of a strided convolution is a row matrix of contributions plus a per-pixel sum in a fixed order, weight gradients write one
 * partial tile per K split and a second kernel adds the partials in split order, bias gradients likewise.  (The heads' backward is the
 * same in both modes since round 5: rows in pixel order, plain stores, ordered sums -- no atomics either way.)  Same inputs ->
 * the same bits, run to run and eager vs HIP-graph replay; the cost is reported by bench.py (per_config[*].deterministic_ms_per_step).
 * (The BatchNorm sums stay fp64 atomics over per-workgroup fp32 partials: such sums are exact -- hence order-independent -- unless the
 * partials span more than ~2^18 in magnitude.)  The workspace sizes of ssdk_heads_bwd / ssdk_conv2d_bwd depend on the mode: size and
 * call under the same setting.  Returns the previous setting.
 */
int ssdk_set_deterministic(int enabled);
int ssdk_get_deterministic(void);
const char* ssdk_last_error_string(void);

/* ---- anchors ------------------------------------------------------------------------------------------------ */

/*
 * Host helper: per-level box sizes of detection/anchor_generators/ssd.py:55-104,125-136 (SsdAnchorGenerator with
 * num_branches = 1, flip = True).  `ratios` are the level's aspect_ratios before flipping; min_scale/max_scale are
 * the fp32 values scales[i], scales[i+1] of ssd.py:33.  Writes nb x (w, h) to hws (host, capacity hws_cap pairs)
 * and returns nb (>0) or an error (<0).
 */
int ssdk_anchor_sizes_ssd(const double* ratios, int nratio, float min_scale, float max_scale, int img_w, int img_h,
                          float* hws, int hws_cap);

/* Host helper: detection/anchor_generators/retina_net.py:18-26,40-43 box sizes (scale-major, ratio-minor). */
int ssdk_anchor_sizes_retina(const double* ratios, int nratio, int level, double scale, int scales_per_level,
                             float* hws, int hws_cap);

/* Host helper: torch.linspace(start, end, steps) fp32 as used by ssd.py:33 for the scale table. */
int ssdk_linspace_f32(float start, float end, int steps, float* out);

/*
 * Device: one level of _generate_anchors (ssd.py:106-151 / retina_net.py:28-54): out DEV [H, W, nb, 4] with
 * centres linspace((0.5)*step, (0.5 + n - 1)*step, n), step = img / n, and the given (w, h) table (host, nb pairs).
 */
int ssdk_anchors_level(float* out, int layer_h, int layer_w, int nb, const float* hws_host, int img_w, int img_h,
                       void* stream);
/* The same with the generator's `step` and `offset` given (ssd.py:111-118,138-139: step_w = step or img_w / layer_w; centres
 * linspace(offset * step, (offset + n - 1) * step, n)); ssdk_anchors_level is step = img / n, offset = (.5, .5). */
int ssdk_anchors_level_ex(float* out, int layer_h, int layer_w, int nb, const float* hws_host, double step_w, double step_h,
                          double offset_x, double offset_y, void* stream);

/* ---- target assignment (T1 + T2 + T3) ------------------------------------------------------------------------- */

size_t ssdk_encode_ground_truth_workspace_bytes(int batch, int total_gt);

/*
 * detection/target_assigner.py:22-63 TargetAssigner.encode_ground_truth, with bf/utils/box_utils.py:83-101 (iou)
 * and detection/matcher.py:33-56 (match_per_prediction, force_match_for_each_target=True) fused.
 *   gt_rows    DEV [total_gt, gt_stride] rows of all images back to back, gt_stride >= 6.  total_gt may be a fixed CAPACITY larger than
 *              gt_offsets[batch] (static buffers of a captured HIP graph): rows past gt_offsets[batch] are padding and ignored
 *   gt_offsets DEV int32 [batch + 1], image i owns rows gt_offsets[i] .. gt_offsets[i+1]
 *   anchors    DEV [A, 4] centroid form
 *   target     DEV [batch, A, 6] (written in full)
 *   box_idx    DEV int32 [batch, A] or NULL: the matcher's box_idx (-2 not matched, -1 ignore, >= 0 GT index)
 */
int ssdk_encode_ground_truth(const float* gt_rows, int gt_stride, const int32_t* gt_offsets, int batch, int total_gt,
                             const float* anchors, int num_anchors, float matched_threshold,
                             float unmatched_threshold, float* target, int32_t* box_idx, void* workspace,
                             size_t workspace_bytes, void* stream);

size_t ssdk_match_per_prediction_workspace_bytes(int num_boxes);
/*
 * detection/matcher.py:33-56 match_per_prediction on a given weight matrix: weights DEV [num_boxes, num_anchors] (any values; NaN
 * ranks above everything, as in torch.max / argmax); box_idx DEV int64 [num_anchors] = argmax over the boxes (first maximum),
 * SSDK_NOT_MATCHED below unmatched_threshold, SSDK_IGNORE in [unmatched, matched); force_match_for_each_target != 0: every box's
 * best anchor (first maximum) is given to it, the highest box index winning an anchor several boxes claim (:52-54).
 */
int ssdk_match_per_prediction(const float* weights, int num_boxes, int num_anchors, float matched_threshold,
                              float unmatched_threshold, int force_match_for_each_target, int64_t* box_idx, void* workspace,
                              size_t workspace_bytes, void* stream);

/* ---- sampler + loss (S1 + L1 + L2 + L3) ----------------------------------------------------------------------- */

size_t ssdk_multibox_loss_workspace_bytes(int batch, int num_anchors, int num_classes);

/*
 * detection/sampler.py:12-25 hard_negative_mining.  scores DEV [batch, A, C]; target_classes DEV fp32, the class
 * of anchor r at target_classes[r * class_stride] (class_stride = 6 and pointer = target + 4 reads the class column
 * of a [batch, A, 6] target in place; 1 = a dense [batch, A] array); sampled DEV uint8 [batch, A] (1 = positive or
 * selected hard negative).  Leaves the per-anchor log-sum-exp of the scores in the workspace for
 * ssdk_multibox_loss_fwd to reuse (lse_valid).
 */
int ssdk_hard_negative_mining(const float* scores, const float* target_classes, int class_stride, int batch,
                              int num_anchors, int num_classes, double negative_per_positive_ratio,
                              int64_t min_negative_per_image, uint8_t* sampled, void* workspace, size_t workspace_bytes,
                              void* stream);

/* detection/sampler.py:9-10 naive_sampler: positives only. */
int ssdk_naive_sampler(const float* target_classes, int class_stride, int batch, int num_anchors, uint8_t* sampled,
                       void* stream);

/*
 * Loss configuration of detection/losses/multibox_loss.py:11-33 (what the constructed loss objects carry).
 *   focal_alpha < 0 means None (SoftmaxFocalLoss only).  reduce_mean == 1: the focal losses divide by the number of
 *   sampled rows -- the reference's constructor drops reduction='sum' for classes whose __init__ takes **kwargs
 *   (bf/utils/misc_utils.py:22-29).  reduce_mean == 2: out3 holds the plain sums, NOT divided by
 *   max(1, #positives) -- a loss module of bf/modules/losses.py:34-106 on its own (forward(prediction, target) outside
 *   MultiboxLoss).  soft_epsilon: label smoothing of the soft-target losses.
 */
typedef struct ssdk_loss_params {
    int cls_kind;  /* SSDK_CLS_* */
    int loc_kind;  /* SSDK_LOC_* */
    float focal_gamma;
    float focal_alpha;
    int reduce_mean;
    float soft_epsilon;
    float classification_weight;
    float localization_weight;
    float xy_scale, wh_scale, eps; /* BoxCoder */
    float smooth_l1_beta;
} ssdk_loss_params;

/*
 * detection/losses/multibox_loss.py:35-94 MultiboxLoss.forward for a given sampled mask.
 *   classification: sum over the sampled rows of the loss selected by cls_kind (ignore_index = -1 for the hard-label
 *   kinds), localisation: loc_kind over the positives.
 *   target DEV [batch, A, 6]: with SSDK_LOC_SMOOTH_L1 it is MUTATED like the reference -- columns 0..3 become the
 *          encoded regression targets (to_centroids + encode_box in place, multibox_loss.py:81-82 / box_coder.py:22-30),
 *          every anchor; with SSDK_LOC_GIOU it is left alone (multibox_loss.py:77-79).
 *   out3   DEV float[3] = (loss, class_loss, loc_loss), each already divided by max(1, #positives).
 *   lse_valid != 0: the workspace already holds this batch's per-anchor log-sum-exp (left there by
 *          ssdk_hard_negative_mining on the same scores), so the scores are not read again for sampled negatives.
 * The workspace keeps what ssdk_multibox_loss_bwd needs (divider, scale, per-anchor log-sum-exp); pass the same one.
 */
int ssdk_multibox_loss_fwd(const ssdk_loss_params* params, const float* scores, const float* locs, const float* anchors,
                           float* target, const uint8_t* sampled, int batch, int num_anchors, int num_classes, int lse_valid,
                           float* out3, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Backward of ssdk_multibox_loss_fwd.  grad_out DEV float[2] = (dL/dclass_loss, dL/dloc_loss).
 * target is the target as the forward call left it.  dscores DEV [batch, A, C] and dlocs DEV [batch, A, 4] are
 * written in full (zeros off the sampled / positive rows).
 */
int ssdk_multibox_loss_bwd(const ssdk_loss_params* params, const float* scores, const float* locs, const float* anchors,
                           const float* target, const uint8_t* sampled, const float* grad_out, int batch, int num_anchors,
                           int num_classes, float* dscores, float* dlocs, void* workspace, size_t workspace_bytes,
                           void* stream);
/* ... which also writes row_mask DEV uint8 [batch][num_anchors] (NULL: not wanted): 1 where the anchor's dscores / dlocs rows can be
 * non-zero (it contributes a classification term, or it is a positive and contributes a box term), 0 where both rows were written as
 * zeros -- the guarantee ssdk_heads_bwd_ex takes.  grad_single != 0: grad_out is ONE device scalar, the upstream gradient of
 * loss = class_loss + loc_loss (out3[0] of the forward), applied to both terms; 0: grad_out[2] = (d class_loss, d loc_loss). */
int ssdk_multibox_loss_bwd_ex(const ssdk_loss_params* params, const float* scores, const float* locs, const float* anchors,
                              const float* target, const uint8_t* sampled, const float* grad_out, int grad_single, int batch, int num_anchors,
                              int num_classes, float* dscores, float* dlocs, uint8_t* row_mask, void* workspace, size_t workspace_bytes,
                              void* stream);

/* detection/box_coder.py:13-34 encode_box (inplace != 0: :22-30, eps after the divide; else :32-34). boxes [n_batch, A, 4]. */
int ssdk_encode_box(const float* boxes, const float* priors, float* out, int batch, int num_anchors, float xy_scale,
                    float wh_scale, float eps, int inplace_semantics, void* stream);
/* detection/box_coder.py:37-57 decode_box -> centroid boxes (inplace_semantics != 0: op order of :45-53, else :55-57). */
int ssdk_decode_box(const float* locs, const float* priors, float* out, int batch, int num_anchors, float xy_scale,
                    float wh_scale, int inplace_semantics, void* stream);

/* ---- postprocess (P1 + P2) ----------------------------------------------------------------------------------- */

size_t ssdk_postprocess_workspace_bytes(int batch, int num_anchors, int num_classes, int softmax, int max_per_class,
                                        int max_total);
/* ... for a call with this soft_nms: soft-NMS without max_per_class (or above 256) keeps a decaying score per candidate and up to
 * min(max_per_class, num_anchors) rows per class, which the size above (= soft_nms 0) does not hold. */
size_t ssdk_postprocess_workspace_bytes_ex(int batch, int num_anchors, int num_classes, int softmax, int max_per_class,
                                           int max_total, int soft_nms);

/*
 * detection/postprocessor.py:24-78 Postprocessor.postprocess with bf/utils/box_utils.py:166-194 (nms wrapper:
 * per-class top max_per_class, then hard NMS per torchvision.ops.nms's contract, or -- soft_nms != 0 -- the gaussian
 * soft-NMS of box_utils.py:145-163 with sigma = soft_sigma) fused.
 *   scores DEV [batch, A, C] logits; locs DEV [batch, A, 4]; priors DEV [A, 4]
 *   softmax != 0: F.softmax and drop column 0 (classes 1..C-1); else sigmoid (classes 1..C)
 *   max_per_class: 1..256, or <= 0 for None (every candidate of a class enters NMS, box_utils.py:186), or > 256 -- the last two take
 *   the greedy kernels (hard: post_nms_any_kernel, a class then yields at most max_total rows, more can never reach the final
 *   top-max_total; soft: post_softnms_any_kernel, workspace from ssdk_postprocess_workspace_bytes_ex) and at most 131 072 anchors;
 *   max_total <= 0 means None
 *   out    DEV [batch, out_cap, 6] rows (x1, y1, x2, y2, class, score); counts DEV int32 [batch];
 *          out_cap >= (max_total > 0 ? max_total : ncls * max_per_class)   (max_per_class None: ncls * num_anchors)
 *   nms_candidates DEV int64 [batch] or NULL: number of boxes that entered NMS per image
 */
int ssdk_postprocess(const float* scores, const float* locs, const float* priors, int batch, int num_anchors,
                     int num_classes, int softmax, float score_threshold, int max_per_class, float nms_threshold,
                     int soft_nms, float soft_sigma, int max_total, float xy_scale, float wh_scale, float* out, int out_cap,
                     int32_t* counts,
                     int64_t* nms_candidates, void* workspace, size_t workspace_bytes, void* stream);

/* ---- multi-scale heads (H1) ----------------------------------------------------------------------------------- */

/*
 * One pyramid level of detection/detector.py:50-63: score = Conv2d(Cin, n_score, 3, padding=1, bias) and
 * loc = Conv2d(Cin, n_loc, 3, padding=1, bias) (detection/detector_builder.py:111-137), each followed by
 * permute(0,2,3,1).contiguous().view(B,-1), written straight into the concatenated outputs of detector.py:65-66.
 *   x        DEV [batch, H, W, Cin]  NHWC (= torch channels_last memory of the [B,Cin,H,W] source map)
 *   w_score  DEV [n_score, 3, 3, Cin] (= channels_last memory of the [n_score,Cin,3,3] parameter); b_score [n_score] or NULL
 *   w_loc    DEV [n_loc, 3, 3, Cin]; b_loc [n_loc] or NULL (n_loc may be 0)
 *   scores_offset / locs_offset: this level's first element inside one image's row of scores / locs
 *            (= C resp. 4 times the anchors of the preceding levels); element (image b, pixel p = y*W+x, channel n) lives at
 *            scores[b*scores_batch_stride + scores_offset + p*n_score + n], locs alike.
 *   backward outputs (ssdk_heads_bwd only; NULL = skip): dx DEV [batch,H,W,Cin]; dw_score/dw_loc like the weights
 *            (both or neither); db_score/db_loc.  All are overwritten, not accumulated.
 */
/* A level that is a SINGLE head (n_loc == 0: the score tower's or the loc tower's convolution of a SharedConvPredictor level, which run as
 * two calls because they read different maps) cannot tell the library how many anchor types a pixel has; its caller may pass that count
 * in locs_offset, which means nothing else for such a level (0 = unknown).  With it the sparse backward of ssdk_heads_bwd takes the
 * ordered anchor-row form (rows of n_score / types columns) in both modes; without it the level falls back to round 4's forms, dense in
 * deterministic mode. */
typedef struct ssdk_head_level {
    const float* x;
    int h, w, cin;
    const float* w_score;
    const float* b_score;
    int n_score;
    const float* w_loc;
    const float* b_loc;
    int n_loc;
    long long scores_offset;
    long long locs_offset;
    float* dx;
    float* dw_score;
    float* db_score;
    float* dw_loc;
    float* db_loc;
} ssdk_head_level;

/*
 * All levels of detector.py:50-66 in one grouped launch (n_levels <= 8): scores DEV [batch, scores_batch_stride],
 * locs DEV [batch, locs_batch_stride].  fp32 in, fp32 accumulate on the matrix cores (v_mfma_f32_32x32x2_f32).
 */
/* workspace DEV (optional, NULL = none): ssdk_heads_fwd_workspace_bytes() bytes, ZERO-FILLED ONCE by the caller before its first use and then
 * left to the library between calls (it holds parked partial tiles and their ready flags, told apart from call to call by a launch
 * counter).  With it, a launch of a few rounds of whole tiles runs in stream-K form: the K slices of the whole launch are cut into equal
 * ranges, one per resident workgroup, and a tile that straddles two ranges is summed through the workspace -- same results up to fp32
 * summation order of that one K split. */
/* The stream-K fix-up wait is bounded (a workgroup that never runs must not hang the GPU), and a wait that runs out is LOUD: the owner fills
 * its output tile with NaN instead of storing an incomplete sum, counts the event in the workspace, and sets a sticky word in pinned host
 * memory that makes this and every later ssdk_heads_fwd of the process return SSDK_E_STREAMK_TIMEOUT.  ssdk_heads_fwd_timeouts reads the
 * workspace's counter back (it SYNCHRONISES `stream`: a test / end-of-run check, not a hot-path call); *timeouts_host == 0 is the only
 * healthy value.  (Reference: none -- torch.nn.Conv2d has no such failure mode; detection/detector.py:50-63.) */
size_t ssdk_heads_fwd_workspace_bytes(void);
int ssdk_heads_fwd_timeouts(const void* workspace, size_t workspace_bytes, void* stream, unsigned* timeouts_host);
/* The sticky host word alone, without synchronising anything: 1 once any stream-K launch of this process has given up on a parked
 * partial tile.  For callers that replay captured HIP graphs (a replay does not pass through ssdk_heads_fwd's own check): test it before
 * every replay.  The kernel is loud on its own as well: from the first loss on, every launch on that workspace -- replayed or not --
 * stores NaN in every tile. */
int ssdk_streamk_poisoned(void);
/* TEST HOOK (fault injection, process-wide, synchronises the device): the stream-K workgroup with range index `drop_workgroup` never
 * raises its flag (-1: none), and an owner gives up after `spin_limit` polls (0: the default 2^22).  Used by
 * tests/streamk_fault_worker.py in a child process; never by the product path. */
int ssdk_debug_streamk_fault(int drop_workgroup, unsigned spin_limit);   /* refused (-3) unless SSDK_ENABLE_FAULT_INJECTION=1 is in the environment */
/* Recovery after SSDK_E_STREAMK_TIMEOUT (e.g. a trainer that goes back to a checkpoint inside the process): synchronises `stream`, clears
 * the flags and the timeout counter of this ssdk_heads_fwd / ssdk_conv2d_fwd_ws workspace and the process-wide sticky word, so that
 * ssdk_heads_fwd runs again and captured graphs replay healthy launches.  The outputs of the step that timed out are lost (NaN). */
int ssdk_streamk_reset(void* workspace, size_t workspace_bytes, void* stream);
/* TEST HOOK: byte offsets, inside the workspace of ssdk_heads_bwd / ssdk_heads_bwd_ex, of the intermediates the ordered anchor-row backward
 * keeps for level `level` (fp32 mode): out[0..7] = ga (rows [type][B*H*W][Jpad]), T (rows [row][9*Cin]), aidx (int [type][B*H*W]: T row or -1),
 * apix (int [type][B*H*W]: pixel of row r), acounts (int[16]: rows per type), plan (int[34]: first T row per type, then first 128-row
 * tile per type), mode (int: 2 = anchor rows, 0 = dense), T capacity in rows.  -3 when these levels do not take that pipeline. */
int ssdk_debug_heads_bwd_layout(const ssdk_head_level* levels, int n_levels, int batch, int level, unsigned long long* out);
int ssdk_heads_fwd(const ssdk_head_level* levels, int n_levels, int batch, float* scores, long long scores_batch_stride,
                   float* locs, long long locs_batch_stride, void* workspace, size_t workspace_bytes, void* stream);
/* The same with a cap on the persistent workgroups of the stream-K form (0 = the library's choice: 512 = every 64 KB LDS slot of the
 * chip; otherwise rounded down to a multiple of 8, at least 256): a caller that runs the levels taken from the backbone
 * (detection/detector.py:36-38) on one stream and the pyramid tail (detector.py:39-43) with its levels' heads on another leaves the
 * tail's kernels room beside this launch.  Several calls may fill disjoint level slices of the same scores / locs rows concurrently
 * on different streams, each with its own workspace. */
int ssdk_heads_fwd_ex(const ssdk_head_level* levels, int n_levels, int batch, float* scores, long long scores_batch_stride,
                      float* locs, long long locs_batch_stride, int max_workgroups, void* workspace, size_t workspace_bytes, void* stream);

/*
 * FAST MODE of ssdk_heads_fwd (opt-in, never the default): the role of apex AMP O1 in the reference (bf/training/env.py:87-95 runs these
 * convolutions in half precision; detection/postprocessor.py:39-40 casts back to fp32).  Every fp32 operand is split into bf16 pieces and
 * the product is the sum of its `terms` = 3 largest cross terms (a_hi b_hi + a_hi b_mid + a_mid b_hi) on v_mfma_f32_32x32x16_bf16 with
 * fp32 accumulation: a product is wrong by ~2^-16 relative instead of exact, logits by ~1e-5 of their scale.  Same arguments, layouts
 * and outputs as ssdk_heads_fwd; Cin % 32 == 0 on every level; workspace = ssdk_heads_fwd_fast_workspace_bytes() bytes (the split
 * weights of this call; nothing is kept between calls).  The backward pass stays ssdk_heads_bwd (fp32).
 */
size_t ssdk_heads_fwd_fast_workspace_bytes(const ssdk_head_level* levels, int n_levels);
int ssdk_heads_fwd_fast(const ssdk_head_level* levels, int n_levels, int batch, float* scores, long long scores_batch_stride, float* locs,
                        long long locs_batch_stride, int terms, void* workspace, size_t workspace_bytes, void* stream);

size_t ssdk_heads_bwd_workspace_bytes(const ssdk_head_level* levels, int n_levels, int batch);

/* FAST MODE of the backward pass (opt-in, like ssdk_heads_fwd_fast / ssdk_conv2d_fwd_fast; the reference's AMP covers forward AND
 * backward: bf/training/env.py:87-95, bf/training/callbacks.py:34-40): the DATA gradients that are dense stride-1 convolutions -- the
 * heads' dense form (every level of a focal-loss step), the 3 x 3 / 1 x 1 stride-1 layers of towers, necks and tails -- run as the forward
 * convolution of dy with the mirrored kernel on the split-bf16 GEMM (three cross terms on v_mfma_f32_32x32x16_bf16, fp32 accumulate);
 * the weights' two bf16 planes are split per call from the re-laid-out fp32 weights -- and the WEIGHT gradients (dense, pixel-row and
 * anchor-row forms alike) run on the split-bf16 form of the K-split GEMM, both operands split in registers after the LDS reads.
 * Everything else of ssdk_heads_bwd / ssdk_conv2d_bwd (the sparse and strided data gradients, bias gradients, the pack) is unchanged
 * fp32, as are the arguments; the
 * workspaces are larger (the planes): size them with the _fast_ functions.  A level / layer the kernel cannot take (channels not a
 * multiple of 32, operands beyond 2 GiB) silently stays on the fp32 kernel. */
size_t ssdk_heads_bwd_fast_workspace_bytes(const ssdk_head_level* levels, int n_levels, int batch);
int ssdk_heads_bwd_fast(const ssdk_head_level* levels, int n_levels, int batch, const float* dscores, long long scores_batch_stride,
                        const float* dlocs, long long locs_batch_stride, int terms, void* workspace, size_t workspace_bytes, void* stream);

/* ssdk_heads_bwd with what the producer of the gradient knows about its sparsity: row_mask DEV uint8 [batch][num_anchors] (NULL: nothing
 * known), 0 = the caller GUARANTEES that the score and loc gradient rows of that anchor (anchors numbered like the rows of scores / locs:
 * level after level, pixel-major, anchor type minor) are entirely zero -- ssdk_multibox_loss_bwd_ex writes exactly this mask for the
 * gradient it produces (hard-negative mining: ~4 % of the anchors).  The pack pass then reads only the pixels that have a marked anchor
 * instead of scanning all of dscores for non-zero rows (SSD-300 / 81 classes, batch 32: 88 MB).  fast_terms: 0 = fp32 (ssdk_heads_bwd),
 * 3 = the dense data gradients in fast mode (ssdk_heads_bwd_fast; size the workspace accordingly). */
int ssdk_heads_bwd_ex(const ssdk_head_level* levels, int n_levels, int batch, const float* dscores, long long scores_batch_stride,
                      const float* dlocs, long long locs_batch_stride, const unsigned char* row_mask, int num_anchors, int fast_terms,
                      void* workspace, size_t workspace_bytes, void* stream);

/* Backward of ssdk_heads_fwd (what autograd derives for detector.py:50-66): dscores / dlocs are the gradients of the
 * concatenated outputs, addressed like scores / locs. */
int ssdk_heads_bwd(const ssdk_head_level* levels, int n_levels, int batch, const float* dscores,
                   long long scores_batch_stride, const float* dlocs, long long locs_batch_stride, void* workspace,
                   size_t workspace_bytes, void* stream);

/* ---- pyramid tail (H2) and RetinaNet tower (H3): generic NHWC convolution + BatchNorm --------------------------- */

/*
 * torch.nn.Conv2d(cin, cout, ksize, stride, padding, bias) on NHWC buffers, as used by bf/modules/conv.py:4-36
 * (Conv2dBn of detection/detector_builder.py:57-109) and detection/modules/predictors.py:28-31,60-66.
 *   x [batch,hin,win,cin]; w [cout,ksize,ksize,cin] (channels_last memory of the OIHW parameter); bias [cout] or NULL;
 *   ksize 1 or 3, stride 1 or 2; relu != 0 fuses max(.,0) into the epilogue (the tower's conv -> ReLU);
 *   y [batch,hout,wout,cout].  Backward fields (ssdk_conv2d_bwd): dy like y (gradient w.r.t. the convolution output,
 *   i.e. with a fused ReLU already undone by ssdk_relu_bwd); dx like x, dw like w, db [cout]; NULL = skip.
 */
typedef struct ssdk_conv_desc {
    const float* x;
    int hin, win, cin;
    const float* w;
    const float* bias;
    int cout, ksize, stride, pad, relu;
    float* y;
    const float* dy;
    float* dx;
    float* dw;
    float* db;
    const float* w_t;   /* ssdk_conv2d_bwd: the weights as re-laid out by ssdk_conv2d_transpose_weights, or NULL (the call does it itself) */
    double* stats;      /* ssdk_conv2d_fwd: NULL, or a BatchNorm `sums` buffer [2 * cout + 2] (fp64) into which the per-channel sums of the
                           output and of its squares are ADDED (the values stored, i.e. after bias and ReLU), [2 * cout] = rows: the
                           statistics half of the BatchNorm that follows (bf/modules/conv.py:33-35), taken in the GEMM epilogue when the
                           convolution is not split over K, else by a pass over the output; cout % 4 == 0 */
} ssdk_conv_desc;

/* n <= 8 convolutions (e.g. the five pyramid levels of one shared tower layer) in one grouped launch. */
int ssdk_conv2d_fwd(const ssdk_conv_desc* descs, int n, int batch, void* stream);
/* The same with the stream-K workspace of ssdk_heads_fwd (ssdk_heads_fwd_workspace_bytes() bytes, zero-filled once by the caller, shared with
 * ssdk_heads_fwd on the same stream; NULL = none): a launch of one or two rounds of whole tiles -- the large layers of a pyramid tail
 * (detection/detector_builder.py:73-82 at 18 x 18 / 9 x 9) -- then runs in stream-K form instead of splitting K with atomics into a
 * zero-filled output.  Same results up to fp32 summation order; a fix-up wait that runs out is loud (ssdk_heads_fwd_timeouts). */
int ssdk_conv2d_fwd_ws(const ssdk_conv_desc* descs, int n, int batch, void* workspace, size_t workspace_bytes, void* stream);

/* The same split-bf16 GEMM for a group of convolutions (ssdk_conv2d_fwd's descriptors: RetinaNet's towers, detection/modules/predictors.py:60-76,
 * and any tail / neck convolution with Cin % 32 == 0).  Bias, ReLU and outputs as ssdk_conv2d_fwd; descriptors that share a weight tensor
 * (a tower layer over five pyramid levels) share its split planes; ssdk_conv_desc::stats is honoured by a statistics pass after the launch. */
size_t ssdk_conv2d_fwd_fast_workspace_bytes(const ssdk_conv_desc* descs, int n);
int ssdk_conv2d_fwd_fast(const ssdk_conv_desc* descs, int n, int batch, int terms, void* workspace, size_t workspace_bytes, void* stream);
size_t ssdk_conv2d_bwd_workspace_bytes(const ssdk_conv_desc* descs, int n, int batch);
/* accumulate != 0: dw / db are added to what the buffers hold; else they are overwritten (descriptors that name the
 * same dw / db -- weights shared across levels -- are summed into it). */
int ssdk_conv2d_bwd(const ssdk_conv_desc* descs, int n, int batch, int accumulate, void* workspace, size_t workspace_bytes,
                    void* stream);
/* ... with the stream-K state of ssdk_heads_fwd / ssdk_conv2d_fwd_ws (ssdk_heads_fwd_workspace_bytes() bytes, zeroed once, kept between
 * calls): with SSDK_CONV_STREAMK_BWD=1 in the environment a stride-1 data-gradient launch of a few rounds of tiles whose last round is
 * partly filled (the RetinaNet towers) runs in stream-K form like the forward launch of the same layers (off by default: measured no
 * faster).  Same results up to fp32 summation order inside a tile cut in two. */
int ssdk_conv2d_bwd_sk(const ssdk_conv_desc* descs, int n, int batch, int accumulate, void* workspace, size_t workspace_bytes,
                       void* sk_workspace, size_t sk_workspace_bytes, void* stream);
/* ... with the stride-1 data gradients in fast mode (see ssdk_heads_bwd_fast above). */
size_t ssdk_conv2d_bwd_fast_workspace_bytes(const ssdk_conv_desc* descs, int n, int batch);
int ssdk_conv2d_bwd_fast(const ssdk_conv_desc* descs, int n, int batch, int accumulate, int terms, void* workspace, size_t workspace_bytes,
                         void* stream);
/* The weights of n <= 24 convolutions in the layout ssdk_conv2d_bwd's backward-data GEMM reads (stride 1: [cin][tap][cout], else
 * [tap][cin][cout]), outs[i] = cin * ksize^2 * cout floats, 16-byte aligned, ONE launch: called once per training step for all the
 * layers of a chain, the results passed as ssdk_conv_desc::w_t (the reference has no counterpart: cuDNN re-lays weights out inside
 * every backward call, bf/modules/conv.py:30-36 through torch.nn.Conv2d). */
int ssdk_conv2d_transpose_weights(const ssdk_conv_desc* descs, int n, float* const* outs, void* stream);
/* dx = y > 0 ? dy : 0 (n floats, n % 4 == 0). */
int ssdk_relu_bwd(const float* y, const float* dy, long long n, float* dx, void* stream);

size_t ssdk_batchnorm_workspace_bytes(int channels); /* = the [2 * channels + 2] doubles of a `sums` buffer (below), 256-byte rounded */
/*
 * torch.nn.BatchNorm2d forward on [rows = batch*H*W][channels] (NHWC), optional fused ReLU after it (conv.py:33-35).
 * training != 0: batch statistics (biased variance), running_mean / running_var updated with `momentum` (unbiased
 * variance) when non-NULL; else running statistics.  save_mean / save_rstd [channels] are kept for the backward.
 * num_batches_tracked DEV int64 [1] or NULL: the module's counter, incremented by one in training mode
 * (torch.nn.modules.batchnorm._BatchNorm.forward does `self.num_batches_tracked.add_(1)` -- a launch of its own per layer).
 */
int ssdk_batchnorm_fwd(const float* x, long long rows, int channels, const float* gamma, const float* beta,
                       float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                       int training, int relu, float* y, float* save_mean, float* save_rstd, void* workspace,
                       size_t workspace_bytes, void* stream);
/* y is the forward output (needed only when relu bit 0 is set, for the mask).  Backward entry points read `relu` as two bits: bit 0 = the
 * norm's own fused ReLU; bit 1 = the norm's INPUT x is the output of a ReLU (conv -> ReLU -> BatchNorm, detection/modules/predictors.py:60-76)
 * whose gradient is taken in the same pass: dx = 0 where x <= 0, and the producer skips its own ReLU-gradient pass. */
int ssdk_batchnorm_bwd(const float* x, const float* y, const float* dy, long long rows, int channels, const float* gamma,
                       const float* save_mean, const float* save_rstd, int relu, int training, float* dx, float* dgamma,
                       float* dbeta, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Synchronised BatchNorm (the reference converts every BatchNorm with apex `convert_syncbn_model`, detection/init.py:85, in
 * --distributed mode): ssdk_batchnorm_fwd / _bwd cut in two, with the caller's all-reduce(sum) over the ranks in between.
*   sums DEV double [2 * channels + 2]: forward  = (sum x, sum x^2, rows, 0); backward = (sum dy', sum dy' * xhat, rows, 0), dy' = dy masked
 *   by the fused ReLU (the last double is padding: buffers are zero-filled in 16-byte units).  One all-reduce(sum) of the whole buffer -- or of several layers' buffers laid back to back, e.g. the five
 *   per-level norms of one RetinaNet tower layer -- makes them the statistics of the global batch.
 *   ssdk_batchnorm_apply: training-mode forward from such sums (count_in_sums != 0: the row count is sums[2 * channels], else `rows`);
 *     running statistics, num_batches_tracked, save_mean / save_rstd as ssdk_batchnorm_fwd.
 *   ssdk_batchnorm_bwd_apply: dx from `sums` (global; total_rows = sums + 2 * channels or NULL for `rows`), dgamma / dbeta from
 *     `sums_local` (this rank's own sums, NULL = sums): parameter gradients stay per-rank sums, the gradient exchange averages them
 *     like every other parameter's (torch.nn.SyncBatchNorm does the same).
 * With one rank the two halves are exactly ssdk_batchnorm_fwd / ssdk_batchnorm_bwd (which are implemented as their composition).
 */
int ssdk_batchnorm_stats(const float* x, long long rows, int channels, double* sums, void* stream);
int ssdk_batchnorm_apply(const float* x, long long rows, int channels, const float* gamma, const float* beta, float* running_mean,
                         float* running_var, int64_t* num_batches_tracked, float momentum, float eps, int relu, float* y,
                         float* save_mean, float* save_rstd, const double* sums, int count_in_sums, void* stream);
int ssdk_batchnorm_bwd_stats(const float* x, const float* y, const float* dy, long long rows, int channels, const float* save_mean,
                             const float* save_rstd, int relu, double* sums, void* stream);
int ssdk_batchnorm_bwd_apply(const float* x, const float* y, const float* dy, long long rows, int channels, const float* gamma,
                             const float* save_mean, const float* save_rstd, int relu, int training, const double* sums,
                             const double* sums_local, const double* total_rows, float* dx, float* dgamma, float* dbeta, void* stream);

/*
 * Chained form for a layer that keeps two `sums` buffers of its own (forward, backward), so that no zero-fill launch runs in front of
 * the statistics: `sums` must hold zeros on entry; `zero_after` (NULL = none; a buffer of the same size that nothing else touches during
 * the call) is zero-filled by the apply launch -- the forward call zeroes the layer's backward buffer, the backward call its forward
 * buffer.  Training mode, statistics of this process's rows; otherwise exactly ssdk_batchnorm_fwd / ssdk_batchnorm_bwd.
 */
int ssdk_batchnorm_fwd_chained(const float* x, long long rows, int channels, const float* gamma, const float* beta, float* running_mean,
                               float* running_var, int64_t* num_batches_tracked, float momentum, float eps, int relu, float* y,
                               float* save_mean, float* save_rstd, double* sums, double* zero_after, void* stream);
int ssdk_batchnorm_bwd_chained(const float* x, const float* y, const float* dy, long long rows, int channels, const float* gamma,
                               const float* save_mean, const float* save_rstd, int relu, float* dx, float* dgamma, float* dbeta,
                               double* sums, double* zero_after, void* stream);
/* The statistics half alone, ADDING into `sums` (no zero-fill; sums[2 * channels] = rows), and the apply half of the chained form for sums
 * that are already complete -- accumulated by the producing convolution's epilogue (ssdk_conv_desc::stats): conv -> BatchNorm then costs one
 * launch for the norm instead of two, and the statistics pass over the activation is gone (bf/modules/conv.py:30-36). */
int ssdk_batchnorm_stats_accumulate(const float* x, long long rows, int channels, double* sums, void* stream);
int ssdk_batchnorm_apply_chained(const float* x, long long rows, int channels, const float* gamma, const float* beta, float* running_mean,
                                 float* running_var, int64_t* num_batches_tracked, float momentum, float eps, int relu, float* y,
                                 float* save_mean, float* save_rstd, const double* sums, double* zero_after, void* stream);

/* ---- FPN top-down step (next-row f1) -------------------------------------------------------------------------------- */

/* bf/modules/features.py:106-107: out = fine + F.interpolate(coarse, size=(hf, wf), mode='nearest'); NHWC, channels % 4 == 0.
 * fine == NULL: plain nearest upsampling (features.py:371, M2Det base features). */
int ssdk_upsample_nearest_add_fwd(const float* fine, const float* coarse, int batch, int hf, int wf, int hc, int wc,
                                  int channels, float* out, void* stream);
/* gradient w.r.t. coarse: each coarse pixel sums dout over the fine pixels it was copied to (the gradient w.r.t. fine is dout). */
int ssdk_upsample_nearest_add_bwd(const float* dout, int batch, int hf, int wf, int hc, int wc, int channels,
                                  float* dcoarse, void* stream);

/* ---- M2Det scale-wise feature aggregation (next-row f1), bf/modules/features.py:273-300 ----------------------------- */

/* F.adaptive_avg_pool2d(x, 1): x [batch, hw, channels] (NHWC) -> out [batch, channels]; and its backward. */
int ssdk_global_avgpool_fwd(const float* x, int batch, int hw, int channels, float* out, void* stream);
int ssdk_global_avgpool_bwd(const float* dout, int batch, int hw, int channels, float* dx, void* stream);
/* out = x * sigmoid(z), z [batch, channels] broadcast over the pixels (features.py:296-298); backward gives dx and dz. */
int ssdk_sigmoid_gate_fwd(const float* x, const float* z, int batch, int hw, int channels, float* out, void* stream);
int ssdk_sigmoid_gate_bwd(const float* x, const float* z, const float* dout, int batch, int hw, int channels, float* dx,
                          float* dz, void* stream);

/* The same gate over PIECES (round 4): the maps the reference concatenates in front of the gate (features.py:385, torch.cat of the eight
 * TUM outputs of a scale, each [batch, hw, piece_channels]) are read where they are -- the concatenated map is never built.
 * `pieces`: n_pieces (<= 8) device pointers, 16-byte aligned, all of the same shape; columns k * piece_channels .. of pooled / z / out / dout /
 * dpool belong to piece k.
 *   pool_fwd:         pooled [batch, n * pc] = mean over the pixels                                  (features.py:288)
 *   gate_fwd:         out [batch, hw, n * pc] = piece * sigmoid(z)                                   (:296-298)
 *   gate_bwd_reduce:  dz [batch, n * pc] = sigmoid'(z) * sum_hw dout * piece                         (first half of the backward: feeds fc2 / fc1)
 *   gate_bwd_apply:   dpieces[k] [batch, hw, pc] = dout[.., k * pc ..] * sigmoid(z) + dpool / hw     (second half: the gate's and the pool's
 *                     gradient of a piece in one pass, each piece's gradient a contiguous map) */
int ssdk_sfam_pool_fwd(const float* const* pieces, int n_pieces, int batch, int hw, int piece_channels, float* pooled, void* stream);
int ssdk_sfam_gate_fwd(const float* const* pieces, int n_pieces, int batch, int hw, int piece_channels, const float* z, float* out,
                       void* stream);
int ssdk_sfam_gate_bwd_reduce(const float* const* pieces, int n_pieces, int batch, int hw, int piece_channels, const float* z,
                              const float* dout, float* dz, void* stream);
int ssdk_sfam_gate_bwd_apply(float* const* dpieces, int n_pieces, int batch, int hw, int piece_channels, const float* z, const float* dout,
                             const float* dpool, void* stream);

/* ---- depthwise convolution (bf/modules/conv.py:39-85 DepthwiseConv2dBn: depthwise k x k, groups = channels, then a 1x1 pointwise
 * convolution; `use_depthwise` configs such as samples/ssd_mb2_voc.py).  NHWC activations, weights [channels][k*k] (= the memory of
 * torch's [C,1,k,k] parameter), optional bias.  HBM-bound stencil; the pointwise half is ssdk_conv2d_*.
 *   fwd: y[b,yo,xo,c] = bias[c] + sum_tap w[c][tap] * x[b, yo*stride - pad + ky, xo*stride - pad + kx, c]
 *   bwd: dx (may be NULL), dw [channels][k*k] and db (may be NULL) are OVERWRITTEN (accumulate == 0) or added to. */
int ssdk_depthwise_conv2d_fwd(const float* x, const float* w, const float* bias, int batch, int hin, int win, int channels, int ksize,
                              int stride, int pad, float* y, void* stream);
int ssdk_depthwise_conv2d_bwd(const float* x, const float* w, const float* dy, int batch, int hin, int win, int channels, int ksize,
                              int stride, int pad, float* dx, float* dw, float* db, int accumulate, void* stream);

/* ---- device-resident input side ---------------------------------------------------------------------
 * bf/core/batch_container.py:25-45  BatchContainer.mixup_ with the random draws (lam ~ Beta(alpha, alpha), index = randperm(B),
 * roll = rand(B) < p) made by the caller exactly as the reference makes them on the host.
 *   images: out[i] = roll[i] ? lam * in[i] + (1 - lam) * in[index[i]] : in[i]   (out-of-place; fp32, two roundings like the
 *   reference's separate multiply and add; lam and 1 - lam are rounded to fp32 first, as torch does for a python scalar).
 *   ground truth (packed rows [total, gt_stride] + int32 offsets [B + 1], as for ssdk_encode_ground_truth): image i keeps its
 *   rows with score (column 5) * lam followed by the rows of image index[i] with score * (1 - lam) when roll[i], unchanged
 *   otherwise (:33-43).  rows_out must hold 2 * total rows; offsets_out [B + 1] is written on the device. */
int ssdk_mixup_images(const float* in, float* out, int batch, long long per_image, const int* index, const unsigned char* roll,
                      double lam, void* stream);
int ssdk_mixup_ground_truth(const float* rows_in, int gt_stride, const int* offsets_in, int batch, const int* index,
                            const unsigned char* roll, double lam, float* rows_out, int* offsets_out, void* stream);

/* ---- evaluation metric -----------------------------------------------------------------------------
 * detection/metrics/mean_average_precision.py:10-116  mean_average_precision(predictions, gts, class_labels, iou_threshold, voc)
 *   predictions [n_pred, 7] device: image id, corner box, class id, score (the rows bf/eval.py:58-66 builds from the
 *   postprocessor's output).  Ground truth as for ssdk_encode_ground_truth: rows [total_gt, gt_stride] + offsets
 *   int32[num_images + 1]; class at column 4; difficult flag at column 6 when gt_stride > 6 (reference :22).
 *   Class ids are integers in [0, num_classes).  ap_out[num_classes]: AP of every class with at least one counted ground
 *   truth, NaN for the others; map_out[1] (double): their mean (:111).  voc != 0: 11-point interpolation (:99-103).
 *   Score ties (the reference's argsort is unstable): lower prediction row first.  The reference's 0/0 -> NaN precision
 *   for a class whose best prediction hits a difficult box is reproduced. */
size_t ssdk_mean_average_precision_workspace_bytes(long long n_pred, long long total_gt, int num_classes);
int ssdk_mean_average_precision(const float* predictions, long long n_pred, const float* gt_rows, int gt_stride,
                                const int* gt_offsets, int num_images, long long total_gt, int num_classes,
                                float iou_threshold, int voc, float* ap_out, double* map_out, void* workspace,
                                size_t workspace_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SSDK_H_ */
