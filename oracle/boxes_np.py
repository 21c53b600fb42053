"""CPU oracle for the 3D box ops of the RetinaUNet hot path (numpy, fp32 arithmetic).

TEST INFRASTRUCTURE ONLY: imported by `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` as the *checker*. The product (`nndetection_amd/`) never
imports this module and has no CPU fallback.

Every function restates one reference function (paths relative to /root/reference) with the same
operation order so that fp32 results are bit-identical to the reference's CPU path; it is pinned
against the real reference by `tests/golden/make_golden.py` (fixtures in `tests/golden/*.npz`,
checked by `tests/test_oracle_golden.py`).

Box layout everywhere: (x1, y1, x2, y2, z1, z2)  [nndet/core/boxes/ops.py:131-159].
"""
from itertools import product
from typing import List, Sequence, Tuple

import numpy as np

F32 = np.float32
INF = 100.0  # nndet/core/boxes/matcher/atss.py uses INF = 100 as the "not a candidate" fill value
BELOW_LOW_THRESHOLD = -1  # nndet/core/boxes/matcher/base.py:14
BETWEEN_THRESHOLDS = -2   # nndet/core/boxes/matcher/base.py:15
BBOX_XFORM_CLIP = float(np.log(1000.0 / 16))  # torchvision BoxCoder default: 4.135166...


def _f(a):
    return np.ascontiguousarray(a, dtype=F32)


# --------------------------------------------------------------------------------------
# pairwise IoU / GIoU / centre distance
# --------------------------------------------------------------------------------------
def box_area_3d(b: np.ndarray) -> np.ndarray:
    """nndet/core/boxes/ops.py:27-38 : ((x2-x1)*(y2-y1))*(z2-z1)"""
    b = _f(b)
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) * (b[:, 5] - b[:, 4])


def box_iou_union_3d(b1, b2, eps: float = 0.0) -> Tuple[np.ndarray, np.ndarray]:
    """nndet/core/boxes/ops.py:131-159"""
    b1, b2 = _f(b1), _f(b2)
    vol1, vol2 = box_area_3d(b1), box_area_3d(b2)
    x1 = np.maximum(b1[:, None, 0], b2[:, 0])
    y1 = np.maximum(b1[:, None, 1], b2[:, 1])
    x2 = np.minimum(b1[:, None, 2], b2[:, 2])
    y2 = np.minimum(b1[:, None, 3], b2[:, 3])
    z1 = np.maximum(b1[:, None, 4], b2[:, 4])
    z2 = np.minimum(b1[:, None, 5], b2[:, 5])
    inter = (np.maximum(x2 - x1, F32(0)) * np.maximum(y2 - y1, F32(0)) * np.maximum(z2 - z1, F32(0))) + F32(eps)
    union = (vol1[:, None] + vol2) - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        return (inter / union).astype(F32), union.astype(F32)


def box_iou(b1, b2, eps: float = 0.0) -> np.ndarray:
    """nndet/core/boxes/ops.py:75-102 (3D branch). Empty input -> empty 1-D array (like tensor([]))."""
    b1, b2 = _f(b1), _f(b2)
    if b1.size == 0 or b2.size == 0:
        return np.zeros((0,), F32)
    return box_iou_union_3d(b1, b2, eps)[0]


def generalized_box_iou(b1, b2, eps: float = 0.0) -> np.ndarray:
    """nndet/core/boxes/ops.py:162-185. NOTE eps is NOT forwarded to the inner IoU (ops.py:175)."""
    b1, b2 = _f(b1), _f(b2)
    if b1.size == 0 or b2.size == 0:
        return np.zeros((0,), F32)
    iou, union = box_iou_union_3d(b1, b2)
    x1 = np.minimum(b1[:, None, 0], b2[:, 0])
    y1 = np.minimum(b1[:, None, 1], b2[:, 1])
    x2 = np.maximum(b1[:, None, 2], b2[:, 2])
    y2 = np.maximum(b1[:, None, 3], b2[:, 3])
    z1 = np.minimum(b1[:, None, 4], b2[:, 4])
    z2 = np.maximum(b1[:, None, 5], b2[:, 5])
    vol = (np.maximum(x2 - x1, F32(0)) * np.maximum(y2 - y1, F32(0)) * np.maximum(z2 - z1, F32(0))) + F32(eps)
    with np.errstate(divide="ignore", invalid="ignore"):
        return (iou - (vol - union) / vol).astype(F32)


def box_center(b) -> np.ndarray:
    """nndet/core/boxes/ops.py:314-327 : (hi + lo) / 2"""
    b = _f(b)
    return np.stack([(b[:, 2] + b[:, 0]) / F32(2), (b[:, 3] + b[:, 1]) / F32(2), (b[:, 5] + b[:, 4]) / F32(2)], 1)


def box_center_dist(b1, b2) -> np.ndarray:
    """nndet/core/boxes/ops.py:262-287 (euclidean): sqrt((dx^2 + dy^2) + dz^2) in fp32."""
    c1, c2 = box_center(b1), box_center(b2)
    d = c1[:, None, :] - c2[None, :, :]
    d = d * d
    return np.sqrt((d[..., 0] + d[..., 1]) + d[..., 2]).astype(F32)


# --------------------------------------------------------------------------------------
# anchors
# --------------------------------------------------------------------------------------
def cell_anchors_3ds(width: Sequence[float], height: Sequence[float], depth: Sequence[float]) -> np.ndarray:
    """AnchorGenerator3DS.generate_anchors, nndet/core/boxes/anchors.py:526-549 (depth fastest)."""
    s = np.asarray(list(product(width, height, depth)), dtype=F32) / F32(2)
    return np.stack([-s[:, 0], -s[:, 1], s[:, 0], s[:, 1], -s[:, 2], s[:, 2]], 1).astype(F32)


def grid_anchors_3d(grid_sizes, strides, cells) -> Tuple[List[np.ndarray], List[int]]:
    """AnchorGenerator3D.grid_anchors, nndet/core/boxes/anchors.py:337-377.
    Order: x-major (x, y, z) over the grid, then the A cell anchors; per level [Sx*Sy*Sz*A, 6]."""
    out, per_level = [], []
    for size, stride, base in zip(grid_sizes, strides, cells):
        sx = np.arange(size[0], dtype=F32) * F32(stride[0])
        sy = np.arange(size[1], dtype=F32) * F32(stride[1])
        sz = np.arange(size[2], dtype=F32) * F32(stride[2])
        gx, gy, gz = np.meshgrid(sx, sy, sz, indexing="ij")
        gx, gy, gz = gx.reshape(-1), gy.reshape(-1), gz.reshape(-1)
        shifts = np.stack([gx, gy, gx, gy, gz, gz], 1)
        a = (shifts[:, None, :] + base[None, :, :]).reshape(-1, 6).astype(F32)
        out.append(a)
        per_level.append(a.shape[0])
    return out, per_level


def anchors_for_image(image_size, fmap_sizes, width, height, depth) -> Tuple[np.ndarray, List[int]]:
    """AnchorGenerator2D.forward, anchors.py:211-242: stride = int(image/fmap) per axis; levels concatenated."""
    strides = [[int(i / s) for i, s in zip(image_size, fm)] for fm in fmap_sizes]
    cells = [cell_anchors_3ds(w, h, d) for w, h, d in zip(width, height, depth)]
    per, npl = grid_anchors_3d(fmap_sizes, strides, cells)
    return np.concatenate(per, 0), npl


# --------------------------------------------------------------------------------------
# ATSS matcher
# --------------------------------------------------------------------------------------
def center_in_boxes(center, boxes, eps=0.01):
    """nndet/core/boxes/ops.py:290-311: min over the six centre-to-face distances > eps (row i of `center` against row i of `boxes`)."""
    center, boxes = _f(center), _f(boxes)
    ax = np.stack([center[:, 0] - boxes[:, 0], center[:, 1] - boxes[:, 1], boxes[:, 2] - center[:, 0], boxes[:, 3] - center[:, 1],
                   center[:, 2] - boxes[:, 4], boxes[:, 5] - center[:, 2]], 1)
    return ax.min(1) > F32(eps)


def atss_match(boxes, anchors, num_anchors_per_level: Sequence[int], num_anchors_per_loc: int,
               num_candidates: int = 4, center_in_gt: bool = False, min_dist: float = 0.01) -> Tuple[np.ndarray, np.ndarray]:
    """ATSSMatcher.compute_matches (center_in_gt=False: V001; True: atss.py:101-107), nndet/core/boxes/matcher/atss.py:48-122,
    with Matcher.__call__'s no-GT fast path (matcher/base.py:51-56).

    Tie rule (the reference's torch.topk leaves ties implementation-defined): candidates are the
    k smallest by (distance, anchor index) -- lowest index wins. Returns (iou[G,M] fp32, matches[M] int64).
    """
    boxes, anchors = _f(boxes).reshape(-1, 6), _f(anchors)
    G, M = boxes.shape[0], anchors.shape[0]
    if G == 0:
        return np.zeros((0,), F32), np.full((M,), BELOW_LOW_THRESHOLD, np.int64)
    dist = box_center_dist(boxes, anchors)
    cand, start = [], 0
    for apl in num_anchors_per_level:
        k = min(num_candidates * num_anchors_per_loc, apl)
        idx = np.argsort(dist[:, start:start + apl], axis=1, kind="stable")[:, :k]
        cand.append(idx + start)
        start += apl
    cand = np.concatenate(cand, 1)                               # [G, K]
    iou = box_iou(boxes, anchors)                                # [G, M]
    cov = np.take_along_axis(iou, cand, 1)                       # [G, K]
    mean = cov.astype(np.float64).mean(1)
    std = cov.astype(np.float64).std(1, ddof=1) if cov.shape[1] > 1 else np.full((G,), np.nan)
    thr = mean.astype(F32) + std.astype(F32)                     # atss.py:97-99: fp32 mean + fp32 std
    is_pos = cov >= thr[:, None]
    if center_in_gt:
        ac = np.stack([(anchors[:, 2] + anchors[:, 0]) / F32(2), (anchors[:, 3] + anchors[:, 1]) / F32(2),
                       (anchors[:, 5] + anchors[:, 4]) / F32(2)], 1).astype(F32)                   # box_center, ops.py:314-327
        inside = center_in_boxes(ac[cand.reshape(-1)], np.repeat(boxes, cand.shape[1], 0), min_dist).reshape(cand.shape)
        is_pos = is_pos & inside
    best = np.full((G, M), -INF, F32)
    for g in range(G):
        sel = cand[g][is_pos[g]]
        best[g, sel] = iou[g, sel]
    matches = best.argmax(0).astype(np.int64)                    # first max -> lowest GT index on ties
    matches[best.max(0) == -INF] = BELOW_LOW_THRESHOLD
    return iou, matches


def atss_match_blocked(boxes, anchors, num_anchors_per_level: Sequence[int], num_anchors_per_loc: int,
                       num_candidates: int = 4, rows: int = 100, threads: int = 1) -> np.ndarray:
    """`atss_match` (center_in_gt=False; nndet/core/boxes/matcher/atss.py:48-122) for sizes whose [G, M] matrices do not fit
    (2 000 GT x 5 levels x 100 000 anchors = 4 GB per matrix): the GT boxes are walked in blocks of `rows`.
    Per GT the steps are those of `atss_match` -- the k candidates per level with the smallest (distance, anchor index), IoU of the
    candidates only (the same element-wise fp32 expression as `box_iou`), threshold = fp32 mean + fp32 unbiased std -- and the arg-max over
    the GTs (atss.py:109-118: first maximum = lowest GT index) is carried as a running (value, index) pair per anchor. -> matches [M]."""
    boxes, anchors = _f(boxes).reshape(-1, 6), _f(anchors)
    G, M = boxes.shape[0], anchors.shape[0]
    if G == 0:
        return np.full((M,), BELOW_LOW_THRESHOLD, np.int64)
    def block(g0):
        b = boxes[g0:g0 + rows]
        c1, c2 = box_center(b), box_center(anchors)
        dx, dy, dz = (c1[:, None, q] - c2[None, :, q] for q in range(3))
        dist = np.sqrt((dx * dx + dy * dy) + dz * dz)            # == box_center_dist(b, anchors), without the [r, M, 3] temporary
        del dx, dy, dz
        cand, start = [], 0
        for apl in num_anchors_per_level:
            k = min(num_candidates * num_anchors_per_loc, apl)
            d = dist[:, start:start + apl]
            kth = np.partition(d, k - 1, axis=1)[:, k - 1]       # value of the k-th smallest distance per row
            idx = np.empty((b.shape[0], k), np.int64)
            for r in range(b.shape[0]):
                c = np.nonzero(d[r] <= kth[r])[0]                # ascending index; ties of the k-th value included
                idx[r] = c[np.argsort(d[r][c], kind="stable")[:k]]   # (distance, index) order, lowest index first
            cand.append(idx + start)
            start += apl
        cand = np.concatenate(cand, 1)                           # [r, K]
        del dist
        out = []
        for r in range(b.shape[0]):
            cov = box_iou(b[r:r + 1], anchors[cand[r]])[0]       # [K]
            mean = cov.astype(np.float64).mean()
            std = cov.astype(np.float64).std(ddof=1) if cov.shape[0] > 1 else np.nan
            thr = F32(mean) + F32(std)
            pos = cov >= thr
            out.append((cand[r][pos], cov[pos]))
        return out

    starts = list(range(0, G, rows))
    if threads > 1 and len(starts) > 1:                          # numpy releases the GIL in the heavy parts; blocks are independent
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(threads) as ex:
            per_block = list(ex.map(block, starts))
    else:
        per_block = [block(g0) for g0 in starts]
    best = np.full((M,), -INF, F32)
    matches = np.full((M,), BELOW_LOW_THRESHOLD, np.int64)
    g = 0
    for blk in per_block:                                        # the arg-max over the GTs, in GT order
        for sel, v in blk:
            upd = v > best[sel]                                  # strict: an equal IoU keeps the lower GT index
            best[sel[upd]] = v[upd]
            matches[sel[upd]] = g
            g += 1
    return matches


def nms2d(boxes, scores, thr):
    """2D NMS: devIoU + nms_kernel + the host scan of nndet/csrc/cuda/nms.cu:22-34,54-96,203-215 (what nndet._C.nms does for [N, 4]
    boxes; the Python wrapper uses torchvision.ops.nms, same rule) -- greedy over descending score (ties: lower index first),
    suppress iff IoU > thr. -> kept indices into the input order."""
    b, s = _f(boxes).reshape(-1, 4), _f(scores)
    order = np.argsort(-s.astype(np.float64), kind="stable")
    b = b[order]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    alive = np.ones(len(b), bool)
    keep = []
    for i in range(len(b)):
        if not alive[i]:
            continue
        keep.append(order[i])
        bottom, top = np.maximum(b[i, 0], b[i + 1:, 0]), np.minimum(b[i, 2], b[i + 1:, 2])
        left, right = np.maximum(b[i, 1], b[i + 1:, 1]), np.minimum(b[i, 3], b[i + 1:, 3])
        inter = np.maximum(right - left, F32(0)) * np.maximum(top - bottom, F32(0))
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = inter / (area[i] + area[i + 1:] - inter)
        alive[i + 1:] &= ~(iou > F32(thr))
    return np.asarray(keep, np.int64)


def iou_match(boxes, anchors, low: float, high: float, allow_low_quality_matches: bool):
    """IoUMatcher.compute_matches, nndet/core/boxes/matcher/iou.py:43-107 (+ the no-GT fast path, matcher/base.py:51-56).
    Ties (torch.max leaves the index of equal maxima implementation-defined): lowest GT index per anchor, lowest anchor index per GT;
    two GTs claiming the same anchor through the low-quality rule: the later (higher) GT index, as the in-order assignment."""
    boxes, anchors = _f(boxes).reshape(-1, 6), _f(anchors)
    G, M = boxes.shape[0], anchors.shape[0]
    if G == 0:
        return np.zeros((0,), F32), np.full((M,), BELOW_LOW_THRESHOLD, np.int64)
    iou = box_iou(boxes, anchors)
    vals, matches = iou.max(0), iou.argmax(0).astype(np.int64)
    all_matches = matches.copy()
    below, between = vals < F32(low), (vals >= F32(low)) & (vals < F32(high))
    matches[below] = BELOW_LOW_THRESHOLD
    matches[between] = BETWEEN_THRESHOLDS
    if allow_low_quality_matches:
        best = iou.argmax(1)
        for g in range(G):
            matches[best[g]] = g
    return iou, matches


def assign_targets(matches: np.ndarray, gt_boxes, gt_classes, num_anchors: int):
    """BaseRetinaNet.assign_targets_to_anchors for one image, nndet/core/retina.py:258-288.
    labels: fp32, 0 background, c+1 foreground; matched boxes [M,6] (zeros when no GT)."""
    gt_boxes = _f(gt_boxes).reshape(-1, 6)
    if gt_boxes.shape[0] > 0:
        m = np.clip(matches, 0, None)
        mb = gt_boxes[m]
        lab = _f(gt_classes)[m] + F32(1)
    else:
        mb = np.zeros((num_anchors, 6), F32)
        lab = np.zeros((num_anchors,), F32)
    lab = lab.copy()
    lab[matches == -1] = 0.0
    lab[matches == -2] = -1.0
    return lab, mb


# --------------------------------------------------------------------------------------
# NMS
# --------------------------------------------------------------------------------------
def _iou_one_vs_many(a, b):
    """devIoU_3d, nndet/csrc/cuda/nms.cu:36-51 (same fp32 ops as ops.py:131-159 with eps=0)."""
    x1 = np.maximum(a[0], b[:, 0]); y1 = np.maximum(a[1], b[:, 1])
    x2 = np.minimum(a[2], b[:, 2]); y2 = np.minimum(a[3], b[:, 3])
    z1 = np.maximum(a[4], b[:, 4]); z2 = np.minimum(a[5], b[:, 5])
    inter = np.maximum(x2 - x1, F32(0)) * np.maximum(y2 - y1, F32(0)) * np.maximum(z2 - z1, F32(0))
    sa = (a[2] - a[0]) * (a[3] - a[1]) * (a[5] - a[4])
    sb = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) * (b[:, 5] - b[:, 4])
    with np.errstate(divide="ignore", invalid="ignore"):
        return inter / ((sa + sb) - inter)


def nms(boxes, scores, thr: float) -> np.ndarray:
    """Greedy 3D NMS = nms_cuda (nndet/csrc/cuda/nms.cu:148-221) == nms_cpu (nndet/core/boxes/nms.py:31-53):
    sort by descending score (ties: lower index first -- the reference's sort is unstable there),
    box j is suppressed iff IoU(kept i, j) > thr (NaN never suppresses, as in the CUDA kernel).
    Returns int64 indices into the input, in decreasing score order."""
    boxes, scores = _f(boxes).reshape(-1, 6), _f(scores)
    n = boxes.shape[0]
    if n == 0:
        return np.zeros((0,), np.int64)
    order = np.argsort(-scores, kind="stable")
    b = boxes[order]
    alive = np.ones(n, bool)
    keep = []
    thr = F32(thr)
    for i in range(n):
        if not alive[i]:
            continue
        keep.append(i)
        if i + 1 < n:
            sup = _iou_one_vs_many(b[i], b[i + 1:]) > thr
            alive[i + 1:] &= ~sup
    return order[np.asarray(keep, np.int64)]


def batched_nms(boxes, scores, idxs, thr: float) -> np.ndarray:
    """nndet/core/boxes/nms.py:81-106: offset boxes by class * (max_coordinate + 1) in fp32, then nms."""
    boxes = _f(boxes).reshape(-1, 6)
    if boxes.size == 0:
        return np.zeros((0,), np.int64)
    off = _f(idxs) * (boxes.max() + F32(1))
    return nms(boxes + off[:, None], scores, thr)


# --------------------------------------------------------------------------------------
# box coder / clip / post-processing front end
# --------------------------------------------------------------------------------------
def decode_single(rel_codes, boxes, weights=(1.,) * 6, clip: float = BBOX_XFORM_CLIP) -> np.ndarray:
    """nndet/core/boxes/coder.py:90-155 for [N,6] codes (dx, dy, dw, dh, dz, dd)."""
    r, b = _f(rel_codes).reshape(-1, 6), _f(boxes)
    w = b[:, 2] - b[:, 0]; h = b[:, 3] - b[:, 1]; d = b[:, 5] - b[:, 4]
    cx = b[:, 0] + F32(0.5) * w; cy = b[:, 1] + F32(0.5) * h; cz = b[:, 4] + F32(0.5) * d
    dx = r[:, 0] / F32(weights[0]); dy = r[:, 1] / F32(weights[1])
    dw = np.minimum(r[:, 2] / F32(weights[2]), F32(clip)); dh = np.minimum(r[:, 3] / F32(weights[3]), F32(clip))
    dz = r[:, 4] / F32(weights[4]); dd = np.minimum(r[:, 5] / F32(weights[5]), F32(clip))
    pcx = dx * w + cx; pcy = dy * h + cy; pcz = dz * d + cz
    pw = np.exp(dw) * w; ph = np.exp(dh) * h; pd = np.exp(dd) * d
    half = F32(0.5)
    return np.stack([pcx - half * pw, pcy - half * ph, pcx + half * pw, pcy + half * ph,
                     pcz - half * pd, pcz + half * pd], 1).astype(F32)


def clip_boxes_to_image(boxes, img_shape) -> np.ndarray:
    """clip_boxes_to_image_3d_, nndet/core/boxes/clip.py:83-101 (x,y,z limits = s0,s1,s2)."""
    b = _f(boxes).copy()
    s0, s1, s2 = [F32(s) for s in img_shape]
    b[:, 0] = np.clip(b[:, 0], 0, s0); b[:, 2] = np.clip(b[:, 2], 0, s0)
    b[:, 1] = np.clip(b[:, 1], 0, s1); b[:, 3] = np.clip(b[:, 3], 0, s1)
    b[:, 4] = np.clip(b[:, 4], 0, s2); b[:, 5] = np.clip(b[:, 5], 0, s2)
    return b


def remove_small_boxes(boxes, min_size: float) -> np.ndarray:
    """nndet/core/boxes/ops.py:241-259 -> indices to keep."""
    b = _f(boxes)
    m = F32(min_size)
    k = ((b[:, 2] - b[:, 0]) >= m) & ((b[:, 3] - b[:, 1]) >= m) & ((b[:, 5] - b[:, 4]) >= m)
    return np.nonzero(k)[0]


def sigmoid(x):
    x = np.asarray(x, np.float64)
    return (1.0 / (1.0 + np.exp(-x))).astype(F32)


def postprocess_single_image(boxes, probs, image_shape, num_classes=1, topk=10000, score_thresh=0.0,
                             min_size=0.01, nms_thresh=0.6, detections_per_img=100):
    """BaseRetinaNet.postprocess_detections_single_image, nndet/core/retina.py:332-379.
    boxes: decoded [M,6]; probs [M,C]. Sort ties: lower flat index first."""
    boxes = clip_boxes_to_image(boxes, image_shape)
    p = _f(probs).reshape(-1)
    k = min(topk, boxes.shape[0])
    idx = np.argsort(-p, kind="stable")[:k]
    p = p[idx]
    keep = p > F32(score_thresh)
    p, idx = p[keep], idx[keep]
    a_idx, labels = idx // num_classes, idx % num_classes
    b = boxes[a_idx]
    keep = remove_small_boxes(b, min_size)
    b, p, labels = b[keep], p[keep], labels[keep]
    keep = batched_nms(b, p, labels, nms_thresh)[:detections_per_img]
    return b[keep], p[keep], labels[keep].astype(np.int64)


def ensembler_postprocess_image(boxes, probs, labels, weights, shape, model_topk=1000, model_score_thresh=0.0,
                                min_size=0.01, model_iou=0.1, model_detections_per_image=100):
    """BoxEnsemblerSelective.postprocess_image (nndet/inference/ensembler/detection.py:166-217) with model_nms_fn =
    batched_nms_model (nndet/inference/detection/model.py:25-54). Probability ties: lower row first."""
    b, p, l, w = _f(boxes).reshape(-1, 6), _f(probs), np.asarray(labels), _f(weights)
    idx = np.argsort(-p, kind="stable")[:model_topk]
    idx = idx[p[idx] > F32(model_score_thresh)]
    b, p, l, w = b[idx], p[idx], l[idx], w[idx]
    b = clip_boxes_to_image(b, shape)
    keep = remove_small_boxes(b, min_size)
    b, p, l, w = b[keep], p[keep], l[keep], w[keep]
    keep = batched_nms(b, p, l, model_iou)[:model_detections_per_image]
    return b[keep], p[keep], l[keep], w[keep]


# --------------------------------------------------------------------------------------
# hard-negative sampler bookkeeping (RNG itself stays in torch on both sides)
# --------------------------------------------------------------------------------------
def hnm_counts(num_positive: int, num_negative: int, batch_size: int, batch_size_per_image: int = 32,
               positive_fraction: float = 0.33, min_neg: int = 1, pool_size: float = 20):
    """HardNegativeSamplerBatched counts, nndet/core/boxes/sampler.py:154-185,237-262."""
    bs = batch_size_per_image * batch_size
    num_pos = min(num_positive, int(bs * positive_fraction))
    num_neg = int(max(1, num_pos) * abs(1 - 1. / float(positive_fraction)))
    num_neg = min(num_negative, max(num_neg, min_neg))
    pool = min(num_negative, int(num_neg * pool_size))
    return num_pos, num_neg, pool


# --------------------------------------------------------------------------------------
# target preparation (instances -> boxes / classes / semantic map)
# --------------------------------------------------------------------------------------
def instances_to_targets(target, mapping):
    """One image. FindInstances + instances_to_boxes + get_instance_class_from_properties + instances_to_segmentation,
    nndet/io/transforms/instances.py:26-41,93-126,166-181,251-296. target [D, H, W] instance ids, mapping {id: class}.
    -> boxes [n, 6] fp32 (min0-1, min1-1, max0+1, max1+1, min2-1, max2+1), classes [n] int64, ids [n], semantic [D, H, W]."""
    t = np.asarray(target)
    ti = t.astype(np.int32)                                  # .to(dtype=torch.int)
    ids = np.unique(ti)
    ids = ids[ids > 0]
    m = {int(k): int(v) for k, v in mapping.items()}
    boxes, classes = [], []
    sem = np.zeros_like(t)
    for i in ids:
        idx = np.stack(np.nonzero(t == i), 1)
        mn, mx = idx.min(0), idx.max(0)
        boxes.append([mn[0] - 1, mn[1] - 1, mx[0] + 1, mx[1] + 1, mn[2] - 1, mx[2] + 1])
        classes.append(m[int(i)])
        sem[t == i] = m[int(i)] + 1
    return (np.asarray(boxes, F32).reshape(-1, 6), np.asarray(classes, np.int64), ids.astype(np.int32), sem)


def hnm_select_reversed(labels, fg_probs, batch_size: int, batch_size_per_image: int = 32, positive_fraction: float = 0.33,
                        min_neg: int = 1, pool_size: float = 20):
    """HardNegativeSamplerBatched.__call__ + DetectionHeadHNM.select_indices (nndet/core/boxes/sampler.py:67-98,187-270,
    nndet/arch/heads/comb.py:268-276) with torch.randperm(n) := (n-1, ..., 0), the permutation the parity tests patch in.
    Pool ties (torch.topk leaves them open): lower index first. -> (pos indices, neg indices, pool indices), ascending."""
    lab, p = _f(labels), _f(fg_probs)
    positive = np.nonzero(lab >= 1)[0]
    negative = np.nonzero(lab == 0)[0]
    num_pos, num_neg, pool = hnm_counts(len(positive), len(negative), batch_size, batch_size_per_image, positive_fraction, min_neg, pool_size)
    pos = positive[::-1][:num_pos]
    order = np.argsort(-p[negative], kind="stable")[:pool]
    pool_idx = negative[order]
    neg = pool_idx[::-1][:num_neg]
    return np.sort(pos), np.sort(neg), pool_idx


# --------------------------------------------------------------------------------------
# weighted box clustering (inference ensembling)
# --------------------------------------------------------------------------------------
def wbc(boxes, scores, weights, n_exp_preds, iou_thresh, score_thresh, use_area=True, missing_weight=1.0):
    """wbc + compute_cluster_consolidation, nndet/inference/detection/wbc.py:94-199, one class. Score ties: lower index first."""
    b, s, w, ne = _f(boxes).reshape(-1, 6), _f(scores), _f(weights), _f(n_exp_preds)
    if b.shape[0] == 0:
        return np.zeros((0, 6), F32), np.zeros((0,), F32)
    ious = box_iou(b, b)
    if use_area:
        w = w * box_area_3d(b)
    pool = np.argsort(-s, kind="stable")
    nb, ns = [], []
    while pool.size > 0:
        h = pool[0]
        with np.errstate(invalid="ignore"):
            row = ious[h][pool]
            idx = pool[row > F32(iou_thresh)]
            rest = pool[row <= F32(iou_thresh)]
        if idx.size:
            iou_c = ious[h][idx]
            msw = iou_c * w[idx]
            ms = msw * s[idx]
            n_missing = max(F32(0), F32(ne[idx].astype(F32).mean()) - F32(len(idx)))
            denom = msw.sum(dtype=F32) + F32(n_missing) * F32(msw.mean(dtype=F32)) * F32(missing_weight)
            sc = ms.sum(dtype=F32) / denom
            bb = (b[idx] * ms[:, None]).sum(0, dtype=F32) / ms.sum(dtype=F32)
            if sc > F32(score_thresh):
                nb.append(bb); ns.append(sc)
        pool = rest
    if not nb:
        return np.zeros((0, 6), F32), np.zeros((0,), F32)
    return np.stack(nb).astype(F32), np.asarray(ns, F32)


def batched_wbc(boxes, scores, labels, weights, iou_thresh, n_exp_preds, score_thresh, use_area=False, missing_weight=1.0):
    """batched_wbc, wbc.py:22-91: per label (ascending), concatenated."""
    labels = np.asarray(labels)
    ob, os_, ol = [], [], []
    for l in np.unique(labels):
        m = labels == l
        bb, ss = wbc(_f(boxes)[m], _f(scores)[m], _f(weights)[m], _f(n_exp_preds)[m], iou_thresh, score_thresh, use_area, missing_weight)
        ob.append(bb); os_.append(ss); ol.append(np.full((len(ss),), l, F32))
    if not ob:
        return np.zeros((0, 6), F32), np.zeros((0,), F32), np.zeros((0,), F32)
    return np.concatenate(ob), np.concatenate(os_), np.concatenate(ol)
