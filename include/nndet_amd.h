/*
 * nndet_amd.h -- C ABI of the MI355X-native (gfx950) RetinaUNet hot path for nnDetection.
 *
 * This is the drop-in boundary below nnDetection's Python plugin surface. The
 * reference has exactly one native entry point, `nndet._C.nms` (nndet/csrc/ops.cpp:13-15,
 * nndet/csrc/cpu/nms.cpp:18-34, nndet/csrc/cuda/nms.cu:148-221); every other function of the hot
 * path is PyTorch-Python there. Each entry below names the reference function it replaces.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all pointers are DEVICE pointers unless noted;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - the caller owns every buffer (PyTorch's caching allocator in the Python binding);
 *   - return value: 0 on success, a hipError_t (>0) for HIP failures, negative for argument errors
 *     (NNDET_EINVAL ...). Nothing is thrown.
 *   - boxes are fp32 (x1, y1, x2, y2, z1, z2), nndet/core/boxes/ops.py:131-159;
 *   - activations are NDHWC (channels-last-3d) with the channel count padded to a multiple of 32,
 *     dtype NNDET_BF16, NNDET_F16 or NNDET_F32; accumulation is always fp32. NNDET_F16 (IEEE half, round to nearest even) is the
 *     storage type of the reference's own mixed-precision training: pl.Trainer(precision=16, amp_backend='native'),
 *     scripts/train.py:277-278 -- convolutions under torch.autocast(float16), loss scaled by a GradScaler. The kernels neither
 *     scale nor clamp: a value beyond 65504 becomes inf exactly as in the reference, and the caller's GradScaler skips the step.
 */
#ifndef NNDET_AMD_H
#define NNDET_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NNDET_OK 0
#define NNDET_EINVAL (-1)      /* bad argument (shape, alignment, unsupported channel count) */
#define NNDET_EWORKSPACE (-2)  /* workspace too small */

#define NNDET_F32 0
#define NNDET_BF16 1
#define NNDET_F16 2

const char* nndet_version(void);
/* Number of bytes of LDS / registers are compile-time; this reports the arch the library was built for. */
const char* nndet_arch(void);

/* Stream plumbing for the data-parallel / multi-stream wiring (replaces nothing in the reference, which leaves streams to
 * Lightning, scripts/train.py:265-289): a HIP stream restricted to the compute units whose bits are set in cu_mask (`words` 32-bit
 * words, bit i of word w = CU 32 w + i; hipExtStreamCreateWithCUMask). The caller owns the stream and destroys it with
 * nndet_stream_destroy. */
int nndet_stream_create_cumask(const uint32_t* cu_mask, int32_t words, void** stream_out);
int nndet_stream_destroy(void* stream);
/* Measurement hook (bench.py `roofline`): the NEXT launch of the 3x3x3 / stride 1 weight-gradient kernel (k_wgrad3d, uniform form) whose
 * problem has exactly `total_tiles` tiles of 4 x 8 x 8 output points records ev_start right before and ev_stop right after that ONE
 * kernel on the stream it is launched on (the caller's hipEvent_t handles, created with timing enabled). One shot: disarmed by the
 * launch that matched, or by total_tiles = 0. Returns 0. No effect on results. */
int nndet_probe_wgrad3d(int64_t total_tiles, void* ev_start, void* ev_stop);

/* ------------------------------------------------------------------------------------------------
 * 3D NMS -- replaces nndet._C.nms (nms_cuda, nndet/csrc/cuda/nms.cu:148-221; kernel :99-145;
 * IoU :36-51) including its score sort and, unlike the reference, the greedy scan stays on the GPU
 * (no mask D2H, no host loop).
 *   boxes [n,6] fp32, scores [n] fp32 (input order).
 *   keep_out [n] int64: indices into the INPUT order of the kept boxes, by decreasing score
 *                       (ties: lower index first); entries >= *n_keep are set to -1.
 *   n_keep_out [1] int64 (device).
 * Box j is suppressed iff IoU(kept i, j) > iou_threshold (NaN never suppresses).
 * workspace: nndet_nms3d_workspace_bytes(n) bytes, 256-B aligned.
 * ---------------------------------------------------------------------------------------------- */
size_t nndet_nms3d_workspace_bytes(int64_t n);
int nndet_nms3d_f32(const float* boxes, const float* scores, int64_t n, float iou_threshold,
                    int64_t* keep_out, int64_t* n_keep_out, void* workspace, size_t workspace_bytes,
                    void* stream);
/* 2D boxes [n,4] (x1, y1, x2, y2): nndet._C.nms dispatches on dets.size(1) (nms_kernel / devIoU, nndet/csrc/cuda/nms.cu:22-34,54-96,172-180;
 * the Python wrapper sends 2D boxes to torchvision.ops.nms, nndet/core/boxes/nms.py:70-72 -- same rule). Same outputs / workspace as
 * nndet_nms3d_f32; decisions are bit-identical to devIoU: the boxes are widened to (x1, y1, x2, y2, 0, 1), whose 3D IoU multiplies the
 * 2D intersection and areas by exactly 1.0f. */
int nndet_nms2d_f32(const float* boxes, const float* scores, int64_t n, float iou_threshold,
                    int64_t* keep_out, int64_t* n_keep_out, void* workspace, size_t workspace_bytes,
                    void* stream);
/* Same, but the caller supplies the order (descending score) -- the mask + scan part only.
 * `order` [n] int32 indices into boxes. Used by batched post-processing that already sorted. */
int nndet_nms3d_sorted_f32(const float* boxes, const int32_t* order, int64_t n, float iou_threshold,
                           int64_t* keep_out, int64_t* n_keep_out, void* workspace,
                           size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pairwise IoU / GIoU -- replace box_iou / generalized_box_iou
 * (nndet/core/boxes/ops.py:75-102,106-128,131-185). out [n,m] fp32 row-major.
 * `eps` is added to the intersection (IoU) resp. to the hull volume only (GIoU), as the reference.
 * ---------------------------------------------------------------------------------------------- */
int nndet_iou3d_pairwise_f32(const float* a, int64_t n, const float* b, int64_t m, float eps,
                             float* out, void* stream);
int nndet_giou3d_pairwise_f32(const float* a, int64_t n, const float* b, int64_t m, float eps,
                              float* out, void* stream);
/* out[i] = max_j IoU(a[i], b[j]) (NaN propagates like torch.max) without the [n, m] matrix: the objective of the planner's
 * anchor optimisation, box_iou(gt_boxes, anchors).max(dim=1)[0].mean() (nndet/planning/architecture/boxes/base.py:424-484). m >= 1. */
int nndet_iou3d_rowmax_f32(const float* a, int64_t n, const float* b, int64_t m, float eps, float* out, void* stream);
/* Element-wise (diagonal) GIoU with gradient w.r.t. `a`: what GIoULoss needs
 * (nndet/losses/regression.py:147-162 takes torch.diag of the [P,P] matrix). */
int nndet_giou3d_diag_fwd_f32(const float* a, const float* b, int64_t n, float eps, float* out, void* stream);
int nndet_giou3d_diag_bwd_f32(const float* a, const float* b, const float* grad_out, int64_t n, float eps,
                              float* grad_a, void* stream);

/* Gradient of the full [n, m] GIoU matrix w.r.t. BOTH box sets: generalized_box_iou is an autograd expression in the reference
 * (nndet/core/boxes/ops.py:106-128,162-185) and GIoULoss back-propagates through it (nndet/losses/regression.py:158-161).
 * grad_out [n, m] fp32 row-major (the cotangent of nndet_giou3d_pairwise_f32's `out`); grad_a [n, 6] and / or grad_b [m, 6] fp32,
 * either may be NULL (not wanted). Sub-gradient conventions of torch autograd: max / min split 0.5 / 0.5 on ties, clamp(min=0)
 * passes the gradient where its argument is >= 0. Deterministic (fixed summation order, float64 partial sums). */
int nndet_giou3d_pairwise_bwd_f32(const float* a, int64_t n, const float* b, int64_t m, const float* grad_out, float eps,
                                  float* grad_a, float* grad_b, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Anchor grid -- replaces AnchorGenerator3D.grid_anchors (nndet/core/boxes/anchors.py:337-377).
 *   cell [A,6] fp32 (AnchorGenerator3DS.generate_anchors, anchors.py:526-549)
 *   out  [sx*sy*sz*A, 6] fp32, x-major over the grid then the A cell anchors.
 * ---------------------------------------------------------------------------------------------- */
int nndet_anchors3d_grid_f32(const float* cell, int32_t A, int32_t sx, int32_t sy, int32_t sz,
                             int32_t stride_x, int32_t stride_y, int32_t stride_z, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ATSS matching -- replaces ATSSMatcher.compute_matches with center_in_gt=False
 * (nndet/core/boxes/matcher/atss.py:48-122) + Matcher.__call__ (matcher/base.py:27-63) and the
 * label/box gather of BaseRetinaNet.assign_targets_to_anchors (nndet/core/retina.py:258-288).
 * Never materialises the [G,M] distance / IoU matrices.
 *   gt [G,6], anchors [M,6]; level_offsets [L+1] int64 HOST array (prefix sums of anchors per level);
 *   k = num_candidates * num_anchors_per_loc (clamped per level to the level size).
 *   matches [M] int64: index of the matched GT or -1 (BELOW_LOW_THRESHOLD).
 * Candidate rule: the k anchors with the smallest (centre distance, anchor index) per GT and level
 * (the reference's torch.topk leaves ties implementation-defined); positives: IoU >= mean + std
 * (unbiased) over the GT's candidates; an anchor positive for several GTs takes the highest IoU
 * (ties: lowest GT index).
 * G == 0 -> all -1. Any G >= 0 is accepted (GTs are processed in tiles of 16); at most NNDET_ATSS_MAX_BATCH images per
 * batched call.
 * ---------------------------------------------------------------------------------------------- */
#define NNDET_ATSS_MAX_BATCH 64
size_t nndet_atss3d_workspace_bytes(int64_t G, int64_t M, int32_t L, int32_t k);
int nndet_atss3d_match_f32(const float* gt, int64_t G, const float* anchors, int64_t M,
                           const int64_t* level_offsets_host, int32_t L, int32_t k,
                           int64_t* matches, void* workspace, size_t workspace_bytes, void* stream);
/* Whole batch in one pass over the anchors (all images of a batch share the anchor tensor, anchors.py:230-237):
 * gt [G,6] = the GT boxes of all images concatenated, img_off_host [B+1] their offsets (HOST array), matches [B,M] with GT
 * indices LOCAL to the image. Same result as B single-image calls; 4x fewer launches at batch 4. Workspace as for G GTs. */
int nndet_atss3d_match_batched_f32(const float* gt, int64_t G, const int32_t* img_off_host, int32_t B,
                                   const float* anchors, int64_t M, const int64_t* level_offsets_host, int32_t L,
                                   int32_t k, int64_t* matches, void* workspace, size_t workspace_bytes, void* stream);
/* nndet_atss3d_match_batched_f32 that also writes the anchors' TRAINING LABELS: labels_out [B,M] = gt_classes[matched GT] + 1, 0 for an
 * unmatched anchor -- what BaseRetinaNet.assign_targets_to_anchors forms from the matches with a clamp / gather / compare / multiply
 * chain over [B, M] tensors (nndet/core/retina.py:262-287; ATSS produces no BETWEEN_THRESHOLDS = -2). gt_classes [G] float (NULL: all
 * class 0). The matched BOXES are not gathered at all: nndet_detloss_matched_f32 reads them through `matches` at the <= 42 sampled
 * positives (the reference's matched_gt_boxes is a [B, M, 6] gather, 114 MB per step at 160x160x96 / batch 4).
 * center_in_gt != 0 (ATSSMatcher(center_in_gt=True), nndet/core/boxes/matcher/atss.py:101-107; RetinaUNetV001 sets it False): a candidate
 * only becomes a positive of a GT box if the anchor's centre lies inside that box, more than min_dist (the reference: 0.01) from every
 * face (center_in_boxes, nndet/core/boxes/ops.py:290-311). */
int nndet_atss3d_assign_batched_f32(const float* gt, const float* gt_classes, int64_t G, const int32_t* img_off_host, int32_t B,
                                    const float* anchors, int64_t M, const int64_t* level_offsets_host, int32_t L,
                                    int32_t k, int32_t center_in_gt, float min_dist, int64_t* matches, float* labels_out,
                                    void* workspace, size_t workspace_bytes, void* stream);
/* IoU-threshold matcher -- replaces IoUMatcher.compute_matches (nndet/core/boxes/matcher/iou.py:43-107; the matcher of the reference's
 * skeleton module, RetinaUNetV001 configures ATSS): matches [M] = index of the GT with the highest IoU (ties: lowest index),
 * -1 (BELOW_LOW_THRESHOLD) if that IoU < low_threshold, -2 (BETWEEN_THRESHOLDS) if low <= IoU < high_threshold. With
 * allow_low_quality_matches every GT also claims its best anchor (ties: lowest anchor index; two GTs on one anchor: the higher GT index,
 * as the reference's sequential assignment); workspace: G * 8 bytes then. G == 0 -> all -1. The [G, M] IoU matrix is never formed. */
int nndet_iou_match3d_f32(const float* gt, int64_t G, const float* anchors, int64_t M, float low_threshold, float high_threshold,
                          int32_t allow_low_quality_matches, int64_t* matches, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Box decode + clip -- replaces decode_single (nndet/core/boxes/coder.py:90-155, weights = 1) followed
 * by clip_boxes_to_image_3d_ (nndet/core/boxes/clip.py:83-101). img_* <= 0 disables clipping.
 * rel [n,6], anchors [n_anchor,6] (anchor index = row % n_anchor), out [n,6].
 * ---------------------------------------------------------------------------------------------- */
int nndet_decode_clip3d_f32(const float* rel, const float* anchors, int64_t n, int64_t n_anchor,
                            float clip_exp, float img_x, float img_y, float img_z, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused post-processing front end (whole batch, no host round trip) -- replaces
 * DetectionHeadHNM.postprocess_for_inference (decode + sigmoid of ALL anchors, nndet/arch/heads/comb.py:140-158) followed by
 * BaseRetinaNet.postprocess_detections_single_image (nndet/core/retina.py:332-379: clip, descending sort, top-k, score
 * threshold, remove_small_boxes, batched_nms, first detections_per_img) and batched_nms' class offsets
 * (nndet/core/boxes/nms.py:81-106).
 *   scores [B, M*C]: logits (scores_are_probs == 0: sigmoid is applied, nndet/arch/heads/classifier.py box_logits_to_probs)
 *                    or probabilities (scores_are_probs != 0);
 *   deltas [B, M, 6]: regression deltas when anchors != NULL (decoded against anchors [M,6], shared by the images, with
 *                    clip_exp as nndet_decode_clip3d_f32) or already decoded boxes when anchors == NULL;
 *   img_* <= 0 disables clipping; topk <= 0 = no top-k (all M*C candidates); K = min(topk, M) candidates survive per image
 *   ordered by (score descending, flat index a*C + c ascending -- the reference's sort leaves ties implementation-defined);
 *   use_score_thresh: keep score > score_thresh; use_min_size: keep boxes whose every side >= min_size;
 *   outputs [B, max_det, ...]: rows >= out_counts[b] are zero boxes / zero scores / label -1. out_labels = flat index % C.
 * Only the K survivors are decoded; the NMS runs on the already sorted survivors.
 * workspace: nndet_postprocess3d_workspace_bytes(B, M, C, topk) bytes (0 = invalid arguments).
 * ---------------------------------------------------------------------------------------------- */
size_t nndet_postprocess3d_workspace_bytes(int32_t B, int64_t M, int32_t C, int32_t topk);
int nndet_postprocess3d_f32(const float* scores, int32_t scores_are_probs, const float* deltas, const float* anchors,
                            int32_t B, int64_t M, int32_t C, float clip_exp, float img_x, float img_y, float img_z,
                            int32_t topk, float score_thresh, int32_t use_score_thresh, float min_size,
                            int32_t use_min_size, float nms_thresh, int32_t max_det, float* out_boxes,
                            float* out_scores, int64_t* out_labels, int64_t* out_counts, void* workspace,
                            size_t workspace_bytes, void* stream);

/* Same pipeline for ROWS that already are (box, probability, label) triples -- replaces the single-model stage of the
 * inference ensembler, BoxEnsemblerSelective.postprocess_image (nndet/inference/ensembler/detection.py:166-217: sort by
 * probability -> first model_topk -> score threshold -> clip_boxes_to_image -> remove_small_boxes -> model_nms_fn =
 * batched_nms_model, nndet/inference/detection/model.py:25-54 -> first model_detections_per_image):
 *   probs [B, M], boxes [B, M, 6] decoded, labels [B, M] int32 (the class offsets of batched_nms use them);
 *   out_index [B, max_det]: row (0 .. M-1) each kept detection came from, -1 padding -- the caller gathers whatever else
 *   travels with a row (the ensembler's per-box weights). workspace: nndet_postprocess3d_workspace_bytes(B, M, 1, topk). */
int nndet_postprocess3d_rows_f32(const float* probs, const float* boxes, const int32_t* labels, int32_t B, int64_t M,
                                 float img_x, float img_y, float img_z, int32_t topk, float score_thresh,
                                 int32_t use_score_thresh, float min_size, int32_t use_min_size, float nms_thresh,
                                 int32_t max_det, float* out_boxes, float* out_scores, int64_t* out_labels,
                                 int64_t* out_index, int64_t* out_counts, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Weighted box clustering -- replaces wbc / batched_wbc (nndet/inference/detection/wbc.py:22-160,163-199), the cross-model /
 * cross-tile consolidation of the ensemblers (nndet/inference/ensembler/detection.py:166-217,476-537). No [N, N] IoU matrix,
 * no host loop: the cluster heads come from the NMS bit mask + on-device greedy scan (the same recurrence), members are
 * attached to their FIRST head, one thread per cluster consolidates.
 *   boxes [n,6], scores [n], weights [n], n_exp_preds [n] fp32; labels [n] int64 or NULL (NULL = one class: `wbc`; else
 *   clustering per label: `batched_wbc`);
 *   cluster head order: descending score (ties: lower index); members: IoU(head, box) > iou_thresh among the boxes not yet in a
 *   cluster; score = sum(iou w s) / (sum(iou w) + max(0, mean(n_exp) - n_found) * mean(iou w) * missing_weight),
 *   box = sum(box * iou w s) / sum(iou w s), w = weights (* box volume if use_area); clusters with score <= score_thresh are dropped.
 *   out_boxes [n,6] / out_scores [n] / out_labels [n] int64: clusters in head order; out_count [1] int64 (device).
 * Sums are fp32 in pool order (the reference's torch.sum order is unspecified: compare at 1e-5 relative).
 * ---------------------------------------------------------------------------------------------- */
size_t nndet_wbc3d_workspace_bytes(int64_t n);
int nndet_wbc3d_f32(const float* boxes, const float* scores, const int64_t* labels, const float* weights,
                    const float* n_exp_preds, int64_t n, float iou_thresh, float score_thresh, int32_t use_area,
                    float missing_weight, float* out_boxes, float* out_scores, int64_t* out_labels,
                    int64_t* out_count, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Device-side target preparation -- replaces FindInstances -> Instances2Boxes -> Instances2Segmentation
 * (nndet/io/transforms/instances.py:26-41,44-142,207-296), which every training / validation step runs on the GPU before the
 * network (nndet/ptmodule/retinaunet/base.py:141,163): one pass over the instance volume instead of a Python loop with one
 * nonzero() / mask per instance.
 *   inst [B, D*H*W] fp32 instance ids (0 = background; values are truncated to int like `.to(torch.int)`);
 *   cls_table [B, max_id] int32 (device): class of instance id per image, -1 = id not in the image's instance_mapping;
 *   seg_out [B, D*H*W] fp32: class + 1 on instance voxels, else 0;
 *   per image, instances in ascending id order: boxes_out [B, max_id, 6] fp32 = (min0-1, min1-1, max0+1, max1+1, min2-1, max2+1)
 *   over the voxel coordinates (axis 0, 1, 2 = D, H, W), classes_out [B, max_id] int64, ids_out [B, max_id] int32,
 *   counts_out [B] int32. ext_ws: [B, max_id, 6] int32 scratch. err_out [1] int32 bit mask: 1 = an id >= max_id occurred,
 *   2 = an instance has no class (the reference raises KeyError).
 * ---------------------------------------------------------------------------------------------- */
int nndet_instances_to_targets_f32(const float* inst, int32_t B, int32_t D, int32_t H, int32_t W,
                                   const int32_t* cls_table, int32_t max_id, float* seg_out, int32_t* ext_ws,
                                   float* boxes_out, int64_t* classes_out, int32_t* ids_out, int32_t* counts_out,
                                   int32_t* err_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Convolution stack (implicit GEMM on MFMA; NDHWC; fp32 accumulate) -- replaces the
 * torch.nn.Conv3d / ConvTranspose3d calls inside ConvInstanceRelu / ConvGroupRelu
 * (nndet/arch/conv.py:54-143,297-348) for the encoder (nndet/arch/encoder/modular.py:110-126),
 * decoder (nndet/arch/decoder/base.py:391-417) and heads (nndet/arch/heads/classifier.py:160-181,
 * regressor.py:153-173, segmenter.py:167-182).
 *
 * A NndetConv describes one (possibly transposed) convolution problem; the three entry points are
 * the forward pass, the data gradient and the weight gradient. Weights are passed PACKED (see
 * nndet_pack_weight) in the activation dtype; the weight gradient is accumulated in fp32 directly in
 * PyTorch's [Cout, Cin, kd, kh, kw] (or [Cin, Cout, ...] for transposed) layout.
 * ---------------------------------------------------------------------------------------------- */
#define NNDET_STATS_REPLICAS 32   /* stats buffers are [NNDET_STATS_REPLICAS][N][C_p][2] fp64 (atomic-contention spreading) */

typedef struct NndetConv {
    int32_t dtype;            /* NNDET_F32 | NNDET_BF16 | NNDET_F16 (activations + packed weights) */
    int32_t transposed;       /* 0: Conv3d, 1: ConvTranspose3d (requires kernel == stride, padding 0) */
    int32_t batch;
    int32_t cin, cout;        /* logical channels */
    int32_t cin_p, cout_p;    /* physical (padded) channels of the in / out tensors, multiples of 32 (cin_p == 1 allowed for the stem) */
    int32_t in_d, in_h, in_w;     /* input spatial size (x, y, z of the reference = d, h, w here) */
    int32_t out_d, out_h, out_w;  /* output spatial size */
    int32_t k[3], s[3], p[3];     /* kernel, stride, padding per axis */
    /* Deferred normalisation of the INPUT (may be NULL): [batch, cin_p, 2] fp32 (scale, shift) per image and channel. The forward
     * and the weight-gradient kernels then read x as  relu?(x * scale + shift)  while staging it (zero padding stays zero), i.e.
     * they consume the PRE-norm output of the previous conv + the coefficients of nndet_norm_finalize, and the normalised
     * activation is never written to HBM (nndet/arch/conv.py:195,271: conv -> norm -> ReLU -> next conv). Arithmetic = the
     * materialising nndet_norm_apply (fmaf, max, round to dtype), so both routes are bit-identical. Not for transposed convs. */
    const float* in_affine;
    int32_t in_relu;
    int32_t reserved_;
} NndetConv;

/* ------------------------------------------------------------------------------------------------
 * Fused stem block: Conv3d(1 -> C, 3x3x3, stride 1, pad 1, no bias) -> InstanceNorm3d(affine) -> ReLU, the first block of the
 * encoder (nndet/arch/conv.py:146-217 instantiated by nndet/arch/encoder/modular.py:79-108 with in_channels = 1), 16-bit
 * activation types only. With ONE input channel the convolution costs 27 MACs per output value, so it is recomputed instead of
 * stored: forward = statistics pass (reads the image only) + finalize + recompute-normalise-store pass; the pre-norm tensor never
 * exists. Backward needs no input gradient (the image), so dW / dgamma / dbeta come from ONE pass over d_out (the gradient w.r.t. the
 * block output) + the image, followed by a one-block combine (csrc/conv_stem.hip: k_stem_bwd3) -- instead of norm-backward reduce +
 * apply + stem weight gradient (6 passes over the largest activation of the network).
 *   x [N, D, H, W, 1] (dtype); w_f32 [C][27] fp32 (PyTorch [C,1,3,3,3]); out / d_out [N, D, H, W, C_p] (dtype);
 *   stats [NNDET_STATS_REPLICAS][N][C_p][2] fp64 ZEROED scratch; mean_rstd_out [N][C_p][2] fp32 (kept for backward);
 *   dw [C][27], dgamma [C], dbeta [C] fp32 (overwritten); workspace: nndet_stem_block_backward_workspace_bytes(c).
 * nndet_stem_block_supported(c) != 0 iff the descriptor is such a block. */
int32_t nndet_stem_block_supported(const NndetConv* c);
int nndet_stem_block_forward(const NndetConv* c, const void* x, const float* w_f32, const float* gamma, const float* beta, float eps,
                             int32_t relu, void* out, double* stats, float* mean_rstd_out, void* stream);
size_t nndet_stem_block_backward_workspace_bytes(const NndetConv* c);
int nndet_stem_block_backward(const NndetConv* c, const void* x, const void* d_out, const float* w_f32, const float* mean_rstd,
                              const float* gamma, const float* beta, int32_t relu, float* dw, float* dgamma, float* dbeta,
                              void* workspace, size_t workspace_bytes, void* stream);

/* pack W (fp32, PyTorch layout) -> [taps][rows_p][k_p] in `dtype`, zero padded.
 * mode 0: rows = Cout, k = Cin (forward of Conv3d)
 * mode 1: rows = Cin,  k = Cout (data gradient of Conv3d)
 * For ConvTranspose3d (weight [Cin, Cout, ...]) mode 0 gives rows = Cout, k = Cin (its forward) and
 * mode 1 rows = Cin, k = Cout (its data gradient). */
size_t nndet_packed_weight_elems(const NndetConv* c, int32_t mode);
int nndet_pack_weight(const NndetConv* c, int32_t mode, const float* w, void* packed, void* stream);

/* The same packing for n (convolution, mode) pairs in one or two launches: what a training step does after every optimizer
 * step for all layers of the model. convs / modes / w / out are HOST arrays of length n (w[i], out[i] device pointers). */
int nndet_pack_weights_batched(const NndetConv* convs, const int32_t* modes, const float* const* w, void* const* out,
                               int32_t n, void* stream);

/* y[N,od,oh,ow,cout_p] = conv(x[N,id,ih,iw,cin_p]) + bias ; bias may be NULL ([cout_p] fp32, zero padded).
 * Stem special case cin_p == 1: `w_packed_mode0` is the UNPACKED fp32 weight [cout, 1, kd, kh, kw].
 * If stats != NULL ([NNDET_STATS_REPLICAS, N, cout_p, 2] fp64, zeroed by the caller) the epilogue accumulates per-(n, channel)
 * sum and sum of squares of the ROUNDED outputs: the InstanceNorm / GroupNorm statistics pass is fused away. */
int nndet_conv3d_forward(const NndetConv* c, const void* x, const void* w_packed_mode0, const float* bias,
                         const void* residual /* NULL or [N,od,oh,ow,cout_p]: y = conv + bias + residual (UFPN top-down add,
                                                nndet/arch/decoder/base.py:410) */,
                         void* y, double* stats, void* stream);
/* The forward convolution of a block whose INPUT arrives pre-norm (c->in_affine set: the coefficient table of nndet_norm_finalize,
 * nndet/arch/conv.py:195,271 conv -> norm -> ReLU -> next conv), for a producer whose normalised output is ALSO needed in HBM (the
 * decoder's lateral, the segmentation branch and the backward pass read it): ONE call that
 *   writes  x_norm = relu?(x_pre * scale + shift)   (what nndet_norm_apply / nndet_affine_apply write, bit for bit) and
 *   returns y = conv(x_norm) (+ bias, + statistics) exactly as nndet_conv3d_forward(x_norm) would.
 * For the 32 -> 64 stride-2 3x3x3 transition on 16-bit types (nndet_conv3d_forward_norm_input_fused(c) != 0) this is ONE launch
 * (k_ig3s<.., PRE>: the halo is transformed in LDS and the tile's core voxels are stored on the way), i.e. the separate 1.26 GB
 * normalisation pass of the largest activation disappears; every other problem runs nndet_affine_apply + nndet_conv3d_forward. */
int32_t nndet_conv3d_forward_norm_input_fused(const NndetConv* c);
int nndet_conv3d_forward_norm_input(const NndetConv* c, const void* x_pre, void* x_norm, const void* w_packed_mode0, const float* bias,
                                    void* y, double* stats, void* stream);
/* dx[N,id,ih,iw,cin_p] = conv^T(dy[N,od,oh,ow,cout_p]) */
int nndet_conv3d_backward_data(const NndetConv* c, const void* dy, const void* w_packed_mode1,
                               void* dx, void* stream);
/* Split-K variants of nndet_conv3d_forward / nndet_conv3d_backward_data for SMALL problems (the deep encoder stages and the small
 * pyramid levels: nndet/arch/encoder/modular.py:79-108 stages 3-5, nndet/arch/decoder/base.py:243-270 levels 3-5): with a few hundred
 * workgroups that each walk serially through all channel chunks x 27 taps most of the 256 CUs idle. Given `ws` of at least
 * nndet_conv3d_splitk_workspace_bytes(c, kind) bytes (kind 0 forward, 1 data gradient; 0 = this problem is not split) the channel
 * chunks are spread over several workgroups per tile (fp32 partial sums in ws) and a second small launch adds them in a fixed order,
 * applies bias / residual, rounds and accumulates the norm statistics. Same mathematics, fp32 summation order differs from the
 * unsplit launch. With ws == NULL or too small they behave exactly like the plain entry points. */
size_t nndet_conv3d_splitk_workspace_bytes(const NndetConv* c, int32_t kind);
int nndet_conv3d_forward_ws(const NndetConv* c, const void* x, const void* w, const float* bias, const void* residual, void* y,
                            double* stats, void* ws, size_t ws_bytes, void* stream);
int nndet_conv3d_backward_data_ws(const NndetConv* c, const void* dy, const void* w, void* dx, void* ws, size_t ws_bytes, void* stream);

/* Data gradient that also accumulates the bias gradient dbias[cout] += sum over voxels of dy (fp32, zero it first): the pointwise
 * kernels (1x1x1 convolutions, transposed k == s) read every dy element exactly once anyway, which saves the separate
 * column-sum pass over dy (629 MB at full resolution). nndet_conv3d_dgrad_fuses_bias() tells whether a problem is covered;
 * the caller then passes dbias = NULL to nndet_conv3d_backward_weight. */
int32_t nndet_conv3d_dgrad_fuses_bias(const NndetConv* c);
int nndet_conv3d_backward_data_bias(const NndetConv* c, const void* dy, const void* w_packed_mode1, void* dx, float* dbias, void* stream);
/* dx += conv^T(dy): the data gradient ADDED to what dx already holds -- the gradient an earlier consumer of the same activation
 * wrote (an encoder stage output feeds the next stage and the decoder lateral; autograd would add the two gradients in a third pass:
 * two reads + one write of the activation size). dbias as in nndet_conv3d_backward_data_bias, or NULL. Transposed convolutions are
 * not covered (NNDET_EINVAL). */
int nndet_conv3d_backward_data_acc(const NndetConv* c, const void* dy, const void* w_packed_mode1, void* dx_inout, float* dbias,
                                   void* stream);
/* The same accumulation, and -- when dx_inout is then the COMPLETE gradient w.r.t. the output of a conv -> norm -> (ReLU) block (this
 * launch adds the last contribution) -- the two sums per (image, channel) that block's norm backward needs from it, accumulated in the
 * epilogue from the values being stored: S1 = sum g, S2 = sum g * xhat, g = dx * [ReLU mask], into red_ws's replicas (layout and zeroing
 * as for nndet_norm_backward). y_norm = the block's pre-norm tensor (shape of dx), mean_rstd / gamma / beta / relu / c_norm = that norm's.
 * nndet_norm_backward_presummed() then finishes the norm backward without its reduction pass (one read of y_norm here instead of a
 * read of y_norm and of dx there). nndet_conv3d_dgrad_fuses_norm_reduce() tells whether a problem is covered: the strided 3x3x3 data
 * gradients in 16-bit types with 32 input channels (the 32 <- 64 transition at full resolution). */
int32_t nndet_conv3d_dgrad_fuses_norm_reduce(const NndetConv* c);
int nndet_conv3d_backward_data_acc_normred(const NndetConv* c, const void* dy, const void* w_packed_mode1, void* dx_inout,
                                           const void* y_norm, const float* mean_rstd, const float* gamma, const float* beta,
                                           int32_t relu, int32_t c_norm, double* red_ws, void* stream);
/* The same sums from a data gradient that WRITES dx (one consumer: conv -> norm -> ReLU -> conv inside an encoder stage,
 * nndet/arch/blocks/basic.py:45-151): dx = conv^T(dy) and, in the epilogue, S1 / S2 of the block that produced this convolution's
 * input, from the values being stored and that block's pre-norm tensor -- the k_norm_bwd_reduce launch (a read of dx and of y_norm that
 * runs 2-3x slower than alone next to the weight-gradient stream) leaves the serial chain of the backward pass.
 * nndet_conv3d_dgrad_normred_supported(): 3x3x3 / stride 1 / padding 1 in 16-bit types on the k_ig3 configurations with 32 or 64 rows
 * per workgroup (every stride-1 layer of the encoder from 64 channels on; not the 32 -> 32 full-resolution layers, whose input is the
 * fused stem block). */
int32_t nndet_conv3d_dgrad_normred_supported(const NndetConv* c);
int nndet_conv3d_backward_data_normred(const NndetConv* c, const void* dy, const void* w_packed_mode1, void* dx, const void* y_norm,
                                       const float* mean_rstd, const float* gamma, const float* beta, int32_t relu, int32_t c_norm,
                                       double* red_ws, void* stream);
/* dw (fp32, PyTorch layout, ACCUMULATED into: zero it first) ; dbias ([cout] fp32, accumulated) may be NULL.
 * Two-stage reduction through `workspace` (nndet_conv3d_wgrad_workspace_bytes(c) bytes): deterministic, no atomics on dw. */
size_t nndet_conv3d_wgrad_workspace_bytes(const NndetConv* c);
int nndet_conv3d_backward_weight(const NndetConv* c, const void* x, const void* dy, float* dw, float* dbias,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * RAGGED batches ("items"): feature maps of DIFFERENT spatial sizes that share the channel count and go through the SAME layer --
 * the pyramid levels P2..P5 through the shared detection-head trunks (nndet/arch/heads/classifier.py:160-181, regressor.py:153-173,
 * comb.py:85-109 loop over the levels and call the same convs once per level). One launch covers every (level, image) pair:
 * the small levels (75 .. 4800 positions per image) ride along in the half-empty last round of workgroups of the largest level
 * instead of costing a latency-bound launch each, and the weight gradient of the shared parameters is summed over the levels inside
 * the kernel instead of by 3 extra additions per parameter.
 *   storage: ONE [rows, C_p] buffer; item i is the NDHWC volume dims[i] = (D, H, W) starting at voxel row row_off[i]
 *   (row_off[i] * C_p * sizeof(dtype) must be a multiple of 16; items must not overlap; every item < 2^31 bytes).
 * Only the 3x3x3 / stride 1 / padding 1 Conv3d is covered (what the head trunks are); NndetConv.batch / in_* / out_* are ignored,
 * in_affine must be NULL. Per-item statistics use the item index where the uniform entry points use the image index:
 * stats [NNDET_STATS_REPLICAS, n_items, cout_p, 2], mean_rstd [n_items, C_p, 2].
 * ---------------------------------------------------------------------------------------------- */
#define NNDET_MAX_ITEMS 32
typedef struct NndetItems {
    int32_t n_items, reserved_;
    int32_t dims[NNDET_MAX_ITEMS][3];
    int64_t row_off[NNDET_MAX_ITEMS];
} NndetItems;
int nndet_conv3d_forward_items(const NndetConv* c, const NndetItems* items, const void* x, const void* w_packed_mode0,
                               const float* bias, void* y, double* stats, void* stream);
int nndet_conv3d_backward_data_items(const NndetConv* c, const NndetItems* items, const void* dy, const void* w_packed_mode1,
                                     void* dx, void* stream);
/* dw / dbias accumulate the sum over ALL items; workspace: nndet_conv3d_wgrad_workspace_bytes(c) */
int nndet_conv3d_backward_weight_items(const NndetConv* c, const NndetItems* items, const void* x, const void* dy, float* dw,
                                       float* dbias, void* workspace, size_t workspace_bytes, void* stream);
/* nndet_norm_apply / nndet_norm_backward per item (statistics over the item's own voxels); red_ws as in nndet_norm_backward with
 * N = n_items */
int nndet_norm_apply_items(int32_t dtype, const void* x, const double* stats, const float* gamma, const float* beta,
                           const NndetItems* items, int32_t c, int32_t c_p, int32_t groups, float eps, int32_t relu, void* y,
                           float* mean_rstd_out, void* stream);
int nndet_norm_backward_items(int32_t dtype, const void* x, const void* dy, const float* mean_rstd, const float* gamma,
                              const float* beta, const NndetItems* items, int32_t c, int32_t c_p, int32_t groups, int32_t relu,
                              void* dx, float* dgamma, float* dbeta, double* red_ws, void* stream);
/* Split I/O (round 5): the FIRST layer of the classifier and of the regressor trunk read the same pyramid batch
 * (nndet/arch/heads/comb.py:85-109 -> classifier.py:160-181, regressor.py:153-173), so they run as ONE convolution 128 -> 2 x 128 and ONE
 * GroupNorm over the 256 channels (groups of 16 never straddle the two halves). The normalised output is written as two dense tensors --
 * channels [0, split) to y_lo [rows][split], [split, c_p) to y_hi [rows][c_p - split] -- which the two branches' next layers read; the
 * backward pass takes the incoming gradient from two such tensors. split: a multiple of 32. */
int nndet_norm_apply_items_split(int32_t dtype, const void* x, const double* stats, const float* gamma, const float* beta,
                                 const NndetItems* items, int32_t c, int32_t c_p, int32_t groups, float eps, int32_t relu, void* y_lo,
                                 void* y_hi, int32_t split, float* mean_rstd_out, void* stream);
int nndet_norm_backward_items_split(int32_t dtype, const void* x, const void* dy_lo, const void* dy_hi, int32_t split,
                                    const float* mean_rstd, const float* gamma, const float* beta, const NndetItems* items, int32_t c,
                                    int32_t c_p, int32_t groups, int32_t relu, void* dx, float* dgamma, float* dbeta, double* red_ws,
                                    void* stream);

/* ------------------------------------------------------------------------------------------------
 * InstanceNorm3d / GroupNorm (+ReLU) on NDHWC -- replaces nn.InstanceNorm3d / nndet GroupNorm + nn.ReLU
 * inside ConvInstanceRelu / ConvGroupRelu (nndet/arch/conv.py:195,271; nndet/arch/layers/norm.py:26-50).
 * Statistics are per (n, group) over the group's channels and all voxels; groups == C is InstanceNorm.
 *   stats [NNDET_STATS_REPLICAS, N, C_p, 2] fp64 per-channel (sum, sumsq) -- produced by the conv epilogue or
 *   nndet_norm_stats; consumers add the replicas.
 * ---------------------------------------------------------------------------------------------- */
int nndet_norm_stats(int32_t dtype, const void* x, int32_t batch, int64_t spatial, int32_t c_p,
                     double* stats /* zeroed */, void* stream);
/* Coefficients only: mean_rstd_out [N, C_p, 2] (for backward) and scale_shift_out [N, C_p, 2] = (rstd * gamma, beta - mean * rstd *
 * gamma) per image and channel (padded channels: 0, 0) -- what NndetConv.in_affine of the CONSUMER convolution takes. */
int nndet_norm_finalize(const double* stats, const float* gamma, const float* beta, int32_t batch, int64_t spatial, int32_t c,
                        int32_t c_p, int32_t groups, float eps, float* mean_rstd_out, float* scale_shift_out, void* stream);
/* y = relu?(x * scale + shift) with scale_shift [N, C_p, 2] from nndet_norm_finalize: materialises a deferred activation (for a
 * consumer that is not one of the convolution entry points above). */
int nndet_affine_apply(int32_t dtype, const void* x, const float* scale_shift, int32_t batch, int64_t spatial, int32_t c_p,
                       int32_t relu, void* y, void* stream);
/* y = relu?((x - mean) * rstd * gamma + beta); mean_rstd_out [N, C_p, 2] fp32 is written for backward. */
int nndet_norm_apply(int32_t dtype, const void* x, const double* stats, const float* gamma, const float* beta,
                     int32_t batch, int64_t spatial, int32_t c, int32_t c_p, int32_t groups, float eps,
                     int32_t relu, void* y, float* mean_rstd_out, void* stream);
/* backward of y = relu?(norm(x)): dx, and dgamma / dbeta ([c] fp32, accumulated -> zero first).
 * red_ws: fp64 scratch, zeroed: [NNDET_STATS_REPLICAS, N, C_p, 2] partial sums followed by N more doubles (the tickets by which the
 * last workgroup of an image finishes the reduction inside the first pass). */
int nndet_norm_backward(int32_t dtype, const void* x, const void* dy, const float* mean_rstd,
                        const float* gamma, const float* beta, int32_t batch, int64_t spatial, int32_t c,
                        int32_t c_p, int32_t groups, int32_t relu, void* dx, float* dgamma, float* dbeta,
                        double* red_ws, void* stream);
/* The same with the replica sums of red_ws already accumulated by nndet_conv3d_backward_data_acc_normred (uniform batches only). */
int nndet_norm_backward_presummed(int32_t dtype, const void* x, const void* dy, const float* mean_rstd,
                                  const float* gamma, const float* beta, int32_t batch, int64_t spatial, int32_t c,
                                  int32_t c_p, int32_t groups, int32_t relu, void* dx, float* dgamma, float* dbeta,
                                  double* red_ws, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Segmentation loss (2-class) -- replaces DiCESegmenterFgBg.compute_loss: 0.5*CE + 0.5*SoftDice(softmax,
 * batch_dice, no background, smooth 1e-5) (nndet/arch/heads/segmenter.py:184-206,273-289;
 * nndet/losses/segmentation.py:32-151). logits [N, spatial, c_p] (channels 0,1 used); target uint8 (>0 = fg).
 *   sums_out [4] fp64 (zeroed): {sum CE, tp, fp, fn} of the foreground channel.
 *   backward: dlogits = g_ce * dCEsum/dl + g_tp*dtp/dl + g_fp*dfp/dl + g_fn*dfn/dl (padded channels get 0).
 * ---------------------------------------------------------------------------------------------- */
int nndet_segloss_forward(int32_t dtype, const void* logits, const uint8_t* target, int64_t nvox, int32_t c_p,
                          double* sums_out, void* stream);
int nndet_segloss_backward(int32_t dtype, const void* logits, const uint8_t* target, int64_t nvox, int32_t c_p,
                           const float* coeffs /* device [4]: g_ce, g_tp, g_fp, g_fn */, void* dlogits, void* stream);

/* Fused segmentation HEAD for training: the 1x1x1 output conv of DiCESegmenterFgBg (self.conv_out, 2 classes, bias;
 * nndet/arch/heads/segmenter.py:160-182) + the loss sums above in ONE pass over the decoder feature map, and its backward
 * (recomputes the logits, writes dx, accumulates dW / dbias) in one more. The [N, spatial, 32]-padded logits and their
 * gradient are never written (4 x 629 MB per step at 160x160x96, batch 4). Only c_p == 32 (cin <= 32).
 *   x [nvox][32] NDHWC (dtype); w [2][cin] fp32 (already rounded to dtype by the caller, like the packed conv weights);
 *   bias [2] fp32. The logits are rounded to dtype before the loss, as the unfused conv would store them.
 *   backward: dwb_out [2*cin + 2] fp64 (zeroed): dW[0][:], dW[1][:], dbias[0], dbias[1]; dx [nvox][32] (dtype).
 * Inference still materialises the logits through nndet_conv3d_forward. */
int nndet_seghead_forward(int32_t dtype, const void* x, int32_t c_p, int32_t cin, const float* w, const float* bias,
                          const uint8_t* target, int64_t nvox, double* sums_out, void* stream);
int nndet_seghead_backward(int32_t dtype, const void* x, int32_t c_p, int32_t cin, const float* w, const float* bias,
                           const uint8_t* target, int64_t nvox, const float* coeffs, void* dx, double* dwb_out, void* stream);
/* Same backward pass in FACTORISED form: the gradient w.r.t. x is the outer product d1[voxel] * (w[1][:] - w[0][:]) (the two
 * logit gradients of the 2-class softmax are d1 and -d1), so only d1_out [nvox] (dtype) is written -- 2 bytes instead of 64 per voxel.
 * The convolution that produced x (decoder.out.P0, nndet/arch/decoder/base.py:243-270, whose output only the segmenter reads,
 * nndet/arch/heads/segmenter.py:167-182) then gets its data gradient as a ONE-input-channel convolution of d1
 * (nndet_conv3d_forward with cin_p == 1 and the composed, flipped kernel) and its weight gradient as (w1 - w0) (x) the
 * one-channel weight gradient of (d1, conv input) (nndet_conv3d_backward_weight with cin_p == 1): nndetection_amd/arch/conv.py. */
int nndet_seghead_backward_rank1(int32_t dtype, const void* x, int32_t c_p, int32_t cin, const float* w, const float* bias,
                                 const uint8_t* target, int64_t nvox, const float* coeffs, void* d1_out, double* dwb_out, void* stream);
/* The whole segmentation branch of a TRAINING step -- decoder.out.P0 (Conv3d 3x3x3 C -> C + bias, no norm / activation:
 * nndet/arch/decoder/base.py:243-270) followed by DiCESegmenterFgBg's 1x1x1 output conv and loss (nndet/arch/heads/segmenter.py:
 * 184-206,273-289) -- as ONE convolution with a single output channel. Both layers are linear and, when no prediction is asked
 * for, nothing but the loss reads their outputs; the fg / bg loss depends on z = logit_1 - logit_0 only, so
 *     z[p] = c0 + sum_{t, cin} wc[t][cin] x[p + t - 1][cin],  wc[t][cin] = sum_c (w_head[1] - w_head[0])[c] W_out[c][cin][t],
 *     c0 = (w_head[1] - w_head[0]) . b_out + (b_head[1] - b_head[0]).
 * forward:  x [N, D, H, W, 32] (dtype: NNDET_BF16 / NNDET_F16; NNDET_F32 -> NNDET_EINVAL, the fp32 path keeps the two layers),
 *           w_packed [27][32] values of dtype (tap t = (kd * 3 + kh) * 3 + kw, i.e. 16 dwords per tap), c0 (device scalar), target
 *           uint8 [N, D, H, W] -> z_out fp32 [N, D, H, W] and sums_out fp64 [nndet_segbranch_replicas()][4] (zeroed by the caller;
 *           the row sum is {sum CE, tp, fp, fn} as produced by nndet_seghead_forward).
 * backward: d1_out [N * D * H * W] (dtype) = dL/dz from z, target and coeffs [4] = dL/d{sum CE, tp, fp, fn}; dsum_out fp64
 *           [replicas] (zeroed) accumulates sum(d1). The gradients of x, W_out, b_out, w_head, b_head follow from d1 as for
 *           nndet_seghead_backward_rank1 (nndetection_amd/arch/segmenter.py: _SegBranchFn). */
int nndet_segbranch_replicas(void);
int nndet_segbranch_forward(int32_t dtype, const void* x, int32_t N, int32_t D, int32_t H, int32_t W, int32_t c_p, const void* w_packed,
                            const float* c0, const uint8_t* target, float* z_out, double* sums_out, void* stream);
/* The same with the decoder's level-0 lateral absorbed too (nndet/arch/decoder/base.py:243-270, 405-413: x_0 = lateral_0(a_0) +
 * up_1(x_1), lateral_0 = Conv3d 1x1x1 C -> C + bias): z = c0 + conv3(a_0; wc . W_lat) + conv3(u; wc) with u = up_1(x_1) + b_lat
 * (the lateral bias rides on the transposed convolution's bias, which keeps the zero padding of the 3x3x3 convolution exact).
 * x = a_0 with w_packed = the composition wc . W_lat, x2 = u with w2_packed = wc; both [27][32] values of dtype. */
int nndet_segbranch_forward2(int32_t dtype, const void* x, const void* w_packed, const void* x2, const void* w2_packed, int32_t N, int32_t D,
                             int32_t H, int32_t W, int32_t c_p, const float* c0, const uint8_t* target, float* z_out, double* sums_out,
                             void* stream);
/* ... and with the last top-down step absorbed as well (nndet/arch/decoder/base.py:272-304,405-413: u = ConvTranspose3d(k = s = 2)(x_1)):
 * conv3(u; wc) is one 3x3x3 / stride-1 convolution 64 -> 8 of the HALF-resolution map x_1 -- one output channel per parity class of the
 * output voxel, run by nndet_conv3d_forward on composed weights (nndetection_amd/arch/segmenter.py: up_compose) -- plus a bias that
 * depends on the voxel's border class only. x = a_0 with w_packed = wc . W_lat as above; zup [N, D/2, H/2, W/2, 32] (dtype) = that
 * convolution's output (channel (pd * 2 + ph) * 2 + pw of voxel m belongs to output voxel 2 m + (pd, ph, pw)); cb [27] fp32 = the
 * bias of border class (cd * 3 + ch) * 3 + cw (0: first plane of the axis, 1: inside, 2: last plane). D, H, W even. u never exists. */
int nndet_segbranch_forward_up(int32_t dtype, const void* x, const void* w_packed, const void* zup, const float* cb, int32_t N, int32_t D,
                               int32_t H, int32_t W, int32_t c_p, const float* c0, const uint8_t* target, float* z_out, double* sums_out,
                               void* stream);
/* Backward side of that: d1 [N, D, H, W] (dtype) -> dzs_out [N, D/2, H/2, W/2, 32] (dtype; channel (pd * 2 + ph) * 2 + pw of voxel m =
 * d1[2 m + (pd, ph, pw)], channels 8 .. 31 zero: the output gradient of the composed convolution) and csum_out fp64
 * [nndet_segbranch_replicas()][27] (zeroed by the caller): the sum of d1 per border class. */
int nndet_segbranch_s2d(int32_t dtype, const void* d1, int32_t N, int32_t D, int32_t H, int32_t W, void* dzs_out, double* csum_out,
                        void* stream);
/* The parameter-only part of the fused segmentation branch with the absorbed level-0 lateral and last top-down step
 * (nndet/arch/decoder/base.py:243-270 `lateral.P0`, `up.P1`, `out.P0`; nndet/arch/heads/segmenter.py:184-206) in ONE launch: the composed
 * kernels wc [27][32] / wqa (16 bit) / wfa (flipped, [32][27]), the head difference wd [32], the constant c0, bsum = b_up + b_lat, the
 * composed half-resolution kernel wc_up [8][cin1][27] (output channel = parity class of the full-resolution voxel) and the border-class
 * bias cb [27]. t_table [216][216] / k3_table [27][27] are the constant 0/1 selection tables (which tap of which parity class reads which
 * position of the k = s = 2 transposed kernel; which taps stay inside the volume per border class). All inputs / outputs fp32 except wqa. */
int nndet_segbranch_compose_up(int32_t dtype, const float* w_out, const float* b_out, const float* w_head, const float* b_head,
                               const float* w_lat, const float* w_up, const float* b_up, const float* b_lat, const float* t_table,
                               const float* k3_table, int32_t cin1, float* wd, float* wc, float* c0, void* wqa, float* wfa, float* bsum,
                               float* wc_up, float* cb, void* stream);
/* All parameter gradients of the branch from the one-channel correlations (32 channels; every tensor fp32, contiguous):
 * w_out [32][32][27], b_out [32] or NULL, w_lat [32][32] or NULL (then e_a, dw_lat NULL too), wd [32] = w_head[1] - w_head[0],
 * e_x / e_a [32][27] = nndet_conv3d_backward_weight(cin_p == 1) of (d1, top-down term) / (d1, a_0), dsum [n_dsum] fp64 (its sum =
 * sum(d1), as nndet_segbranch_backward leaves it) -> dw_out [32][32][27], db_out [32] or NULL, dw_lat [32][32], dw_head [2][32],
 * db_head [2] or NULL (all WRITTEN). */
int nndet_segbranch_param_grads(const float* w_out, const float* b_out, const float* w_lat, const float* wd, const float* e_x,
                                const float* e_a, const double* dsum, int32_t n_dsum, float* dw_out, float* db_out, float* dw_lat,
                                float* dw_head, float* db_head, void* stream);
int nndet_segbranch_backward(int32_t dtype, const float* z, const uint8_t* target, int64_t nvox, const float* coeffs, void* d1_out,
                             double* dsum_out, void* stream);
/* Scalar tail of the loss: sums [4] fp32 (what the forward entry points above produce, cast to fp32) ->
 * losses_out [2] = {alpha * CE_sum / nvox, (1 - alpha) * (1 - (2 tp + smooth_nom) / (2 tp + fp + fn + smooth_denom))} and
 * coeffs_out [2, 4] = d losses / d sums, in one launch instead of ~45 one-element torch launches (forward + autograd). */
int nndet_segloss_tail_f32(const float* sums, int64_t nvox, float alpha, float smooth_nom, float smooth_denom, float* losses_out,
                           float* coeffs_out, void* stream);


/* ------------------------------------------------------------------------------------------------
 * Batch-level hard-negative sampling on the device -- replaces DetectionHeadHNM.select_indices (nndet/arch/heads/comb.py:247-276)
 * + HardNegativeSamplerBatched (nndet/core/boxes/sampler.py:57-98,154-185,237-270): no torch.where / topk / randperm round trips.
 *   labels [N] fp32 (concatenated over the batch: >= 1 foreground, 0 background, < 0 ignored);
 *   scores [N, C] logits (scores_are_probs == 0; fg probability = sigmoid(max_c)) or [N] probabilities (!= 0, C == 1);
 *   pos_cap = int(batch_size_per_image * B * positive_fraction); neg_pos_ratio = |1 - 1 / positive_fraction|;
 *   num_pos = min(#pos, pos_cap); num_neg = min(#neg, max(int(max(1, num_pos) * neg_pos_ratio), min_neg));
 *   pool = min(#neg, int(num_neg * pool_size)) highest-probability negatives (ties: lowest index);
 *   positives = a uniformly random num_pos-subset of the foreground anchors, negatives = a uniformly random num_neg-subset of
 *   the pool (keys hash(seed, .)); deterministic != 0 selects what randperm(n) := (n-1, ..., 0) selects in the reference (the
 *   LAST num_pos positives, the num_neg LOWEST-scoring pool members) -- used by the parity tests.
 *   pos_idx [pos_cap] / neg_idx [nndet_hnm_neg_capacity(...)] int64: anchor indices in ASCENDING order (the reference reads its
 *   masks back with torch.where), padded with -1; counts [4] int64 = {num_pos, num_neg, #pos, #neg}.
 * ---------------------------------------------------------------------------------------------- */
size_t nndet_hnm_sample_workspace_bytes(int32_t pos_cap, double neg_pos_ratio, int32_t min_neg, double pool_size);
int32_t nndet_hnm_neg_capacity(int32_t pos_cap, double neg_pos_ratio, int32_t min_neg);
int nndet_hnm_sample_f32(const float* labels, const float* scores, int32_t scores_are_probs, int64_t N, int32_t C,
                         int32_t pos_cap, double neg_pos_ratio, int32_t min_neg, double pool_size, uint64_t seed,
                         int32_t deterministic, int64_t* pos_idx, int64_t* neg_idx, int64_t* counts,
                         void* workspace, size_t workspace_bytes, void* stream);

/* sigmoid + max over classes of the (padded-free) logits [n, C] fp32 -> probs [n] ; used by the hard-negative
 * sampler (DetectionHeadHNM.select_indices, nndet/arch/heads/comb.py:247-276). */
int nndet_sigmoid_max_f32(const float* logits, int64_t n, int32_t C, float* out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Detection-head output gather -- replaces, for all pyramid levels of one head branch in ONE pass, the
 * `permute(0, 2, 3, 4, 1).contiguous().view(N, -1, last)` of the classifier / regressor outputs
 * (nndet/arch/heads/classifier.py:176-181, regressor.py:165-172), the level-specific `Scale` of the regressor
 * (regressor.py:163-164, nndet/arch/layers/scale.py:21-43) and the `torch.cat(levels, dim=1)` of
 * DetectionHead.forward (nndet/arch/heads/comb.py:107-108).
 *   y[l]     : conv_out of level l, NDHWC [N, points[l], cout_p] (dtype), channels padded to cout_p (multiple of 32);
 *   scale[l] : device pointer to the level's scalar (fp32) or NULL;  out : fp32 [N, sum_l points[l], cout] (row = position,
 *              i.e. [N, sum_l points[l] * A, last] for cout = A * last), level blocks in order inside every image.
 * Backward: grad_out (same shape as out) -> dy[l] (padded NDHWC gradient of conv_out, zeros in the padding) and, where dscale[l] is
 * not NULL, dscale[l] += sum(grad_out_l * y_l) (fp32 atomics; the caller zeroes it).
 * ---------------------------------------------------------------------------------------------- */
#define NNDET_HEAD_MAX_LEVELS 8
typedef struct NndetHeadLevels {
    int32_t nlev;
    int32_t reserved_;
    const void* y[NNDET_HEAD_MAX_LEVELS];
    void* dy[NNDET_HEAD_MAX_LEVELS];            /* backward only */
    const float* scale[NNDET_HEAD_MAX_LEVELS];
    float* dscale[NNDET_HEAD_MAX_LEVELS];       /* backward only */
    int64_t points[NNDET_HEAD_MAX_LEVELS];
} NndetHeadLevels;
int nndet_head_gather_f32(int32_t dtype, const NndetHeadLevels* levels, int32_t N, int32_t cout, int32_t cout_p, float* out,
                          void* stream);
int nndet_head_gather_backward(int32_t dtype, const NndetHeadLevels* levels, int32_t N, int32_t cout, int32_t cout_p,
                               const float* grad_out, void* stream);

/* Sparse backward of the detection head's OUTPUT convolutions (training): the loss touches <= batch_size_per_image * batch sampled
 * anchors (nndet/arch/heads/comb.py:351-405, nndet/core/boxes/sampler.py:237-270), so the gradient w.r.t. box_logits / box_deltas has
 * <= 170 non-zero rows. Instead of the dense backward of permute / view / cat (classifier.py:176-181, regressor.py:163-172,
 * comb.py:107-108) and of the two 3x3x3 output convolutions over all pyramid levels:
 *   nndet_head_out_sparse_scatter: K entries (idx[k] = anchor index in the [N, sum_l points_l * A] order of nndet_head_gather_f32, -1 =
 *     unused slot; val [K][G] = gradient of the G = num_classes (classifier) or 6 (regressor) values of that anchor) -> row of the ragged
 *     head buffer, first channel a * G, values x the level's Scale (levels->scale, may be NULL); also written into dy_zeroed
 *     [rows][cout_p] (dtype, zero-filled by the caller), which thereby is the complete dense gradient; levels->dscale[l] += d(Scale)
 *     from y = the raw conv output. level_row0_host[l] = first row of level l (rows of level l: row0 + n * points + position).
 *   nndet_conv_out_sparse_backward: data gradient dx_zeroed [rows][cin_p] (dtype, ZERO-FILLED by the caller; only the rows the entries
 *     touch are written, via dx32_scratch, an fp32 scratch of the same shape that needs no initialisation), weight
 *     gradient dw [cout][cin][27] and dbias [cout] (fp32, ACCUMULATED with atomics) of the 3x3x3 / stride 1 / pad 1 convolution from the
 *     entries; x = the conv input (ragged, NndetItems), w_f32 = its weights [cout][cin][27] fp32. */
int nndet_head_out_sparse_scatter(int32_t dtype, const NndetHeadLevels* levels, int32_t N, int32_t A, int32_t G,
                                  const int64_t* level_row0_host, const int64_t* idx, const float* val, int32_t K, const void* y,
                                  int32_t cout_p, void* dy_zeroed, int32_t* rows_out, int32_t* c0_out, float* vals_out, void* stream);
int nndet_conv_out_sparse_backward(const NndetConv* c, const NndetItems* items, const int32_t* rows, const int32_t* c0,
                                   const float* vals, int32_t K, int32_t G, const void* x, const float* w_f32, float* dx32_scratch,
                                   void* dx_zeroed, float* dw, float* dbias, void* stream);
/* Forward of such an output convolution at the K anchors idx[] only: out [K][G] = Scale_l * (conv(x)[row, a*G .. a*G+G-1] + bias),
 * raw_out [K][G] the unscaled values (for d(Scale)), rows_out / c0_out as nndet_head_out_sparse_scatter emits them and level_out the
 * pyramid level of each entry (-1 for unused slots, whose outputs are 0). Used for the regressor in training steps: nndet/arch/heads/comb.py:383-401 reads box_deltas at
 * sampled_pos_inds and nowhere else. */
/* Scale backward for the entries of nndet_conv_out_sparse_forward (nndet/arch/layers/scale.py:21-43: out = scale_l * raw):
 * vals_out [K][G] = grad [K][G] * scale_l (what nndet_conv_out_sparse_backward takes), levels->dscale[l] (not NULL) = sum over the
 * entries of level l of grad . raw (WRITTEN, not accumulated); level [K] as returned by the forward (-1 = unused slot -> zeros). */
int nndet_conv_out_sparse_scale_backward(const NndetHeadLevels* levels, const float* grad, const float* raw, const int32_t* level,
                                         int32_t K, int32_t G, float* vals_out, void* stream);
int nndet_conv_out_sparse_forward(const NndetConv* c, const NndetItems* items, const NndetHeadLevels* levels, int32_t N, int32_t A,
                                  int32_t G, const int64_t* level_row0_host, const int64_t* idx, int32_t K, const void* x,
                                  const float* w_f32, const float* bias, float* out, float* raw_out, int32_t* rows_out,
                                  int32_t* c0_out, int32_t* level_out, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused detection loss on the sampled anchors -- replaces the tail of DetectionHeadHNM.compute_loss (nndet/arch/heads/comb.py:
 * 351-405): decode of the sampled positives (nndet/core/boxes/coder.py:90-155, unit weights, clamp at bbox_xform_clip) ->
 * GIoULoss (nndet/losses/regression.py:118-162: -sum diag GIoU(pred, target, eps)) / max(1, num_pos), and
 * BCEWithLogitsLossOneHot (nndet/losses/classification.py:137-181: one-hot without the background column) of positives +
 * negatives -- forward and the per-row gradients in one single-workgroup launch instead of ~250 element-wise launches.
 *   logits [M_tot, C], deltas [M_tot, 6] fp32 (all anchors of the batch); pos [pos_cap] / neg [neg_cap] int64 anchor indices padded
 *   with -1 and counts = {num_pos, num_neg, ...} as nndet_hnm_sample_f32 writes them; labels [M_tot] (>= 1 foreground class + 1);
 *   matched_gt [M_tot, 6]; anchors [m_anchors, 6] (anchor of index i = anchors[i % m_anchors]: the images share one anchor set).
 *   reg = reg_weight * -1 * S / max(num_pos, 1) with S = sum GIoU (reg_mean: S / max(num_pos, 1));
 *   cls = cls_weight * sum BCE (cls_mean: / (max(num_pos + num_neg, 1) * C)).
 *   losses_out [2] = {reg, cls}; g_deltas_out [pos_cap, 6], g_logits_out [pos_cap + neg_cap, C]: d loss / d row (padding rows 0).
 * nndet_detloss_scatter_f32: backward -- d_deltas [M_tot, 6] / d_logits [M_tot, C] (ZEROED by the caller) receive the sampled rows
 * scaled by upstream = {d/d reg, d/d cls} (device).
 * ---------------------------------------------------------------------------------------------- */
int nndet_detloss_f32(const float* logits, const float* deltas, const int64_t* pos, int32_t pos_cap, const int64_t* neg,
                      int32_t neg_cap, const int64_t* counts, const float* labels, const float* matched_gt, const float* anchors,
                      int64_t m_anchors, int32_t C, float eps, float clip, float reg_weight, int32_t reg_mean, float cls_weight,
                      int32_t cls_mean, float* losses_out, float* g_deltas_out, float* g_logits_out, void* stream);
int nndet_detloss_scatter_f32(const int64_t* pos, int32_t pos_cap, const int64_t* neg, int32_t neg_cap, int32_t C,
                              const float* g_deltas, const float* g_logits, const float* upstream, float* d_deltas, float* d_logits,
                              void* stream);
/* nndet_detloss_scatter_f32 with the upstream gradients as two device scalars (NULL = 0) and, written on the way, the scaled compact rows
 * val_deltas [pos_cap, 6] / val_logits [pos_cap + neg_cap, C] and the index list idx_out [pos_cap + neg_cap] = pos ++ neg (each may be
 * NULL): what the sparse consumers of the two gradients read instead of the dense tensors. One launch instead of six. */
int nndet_detloss_scatter2_f32(const int64_t* pos, int32_t pos_cap, const int64_t* neg, int32_t neg_cap, int32_t C,
                               const float* g_deltas, const float* g_logits, const float* up_reg, const float* up_cls, float* d_deltas,
                               float* d_logits, float* val_deltas, float* val_logits, int64_t* idx_out, void* stream);
/* nndet_detloss_f32 with the box deltas of the sampled positives ONLY: deltas_compact [pos_cap][6], row r belongs to anchor pos[r]
 * (produced by nndet_conv_out_sparse_forward: the regressor's output convolution evaluated at the <= 42 sampled positives instead of
 * at all 4.75 M anchors of a batch). g_deltas_out has the same compact layout as before; nndet_detloss_scatter_f32 accepts
 * d_deltas == NULL in that case (there is no dense delta gradient to scatter into). */
int nndet_detloss_compact_f32(const float* logits, const float* deltas_compact, const int64_t* pos, int32_t pos_cap, const int64_t* neg,
                              int32_t neg_cap, const int64_t* counts, const float* labels, const float* matched_gt, const float* anchors,
                              int64_t m_anchors, int32_t C, float eps, float clip, float reg_weight, int32_t reg_mean, float cls_weight,
                              int32_t cls_mean, float* losses_out, float* g_deltas_out, float* g_logits_out, void* stream);
/* nndet_detloss_f32 / nndet_detloss_compact_f32 with the matched GT boxes given INDIRECTLY: gt_all [G][6] = the GT boxes of the batch
 * concatenated, matches [B * m_anchors] = the ATSS output (index local to the image, -1 unmatched), gt_base_host [B] (HOST array) = first
 * row of each image in gt_all; B <= 64. deltas_compact != 0: deltas are the [pos_cap][6] rows of the sampled positives. Replaces the
 * `matched_gt_boxes[sampled_pos_inds]` read of DetectionHeadHNM.compute_loss (nndet/arch/heads/comb.py:383-391) without the dense
 * [B * M, 6] tensor behind it. */
int nndet_detloss_matched_f32(const float* logits, const float* deltas, int32_t deltas_compact, const int64_t* pos, int32_t pos_cap,
                              const int64_t* neg, int32_t neg_cap, const int64_t* counts, const float* labels, const float* gt_all,
                              const int64_t* matches, const int32_t* gt_base_host, int32_t B, const float* anchors, int64_t m_anchors,
                              int32_t C, float eps, float clip, float reg_weight, int32_t reg_mean, float cls_weight, int32_t cls_mean,
                              float* losses_out, float* g_deltas_out, float* g_logits_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NNDET_AMD_H */
