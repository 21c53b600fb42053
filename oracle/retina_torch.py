"""CPU oracle for the RetinaUNetV001 network + train/inference step (plain PyTorch fp32).

TEST INFRASTRUCTURE ONLY (see oracle/README.md): the checker for the floating-point part of the
hot path and the `cpu_baseline` ("port") timed by bench.py. The product never imports it.

It restates, with stock `torch.nn` layers and `oracle.boxes_np`, what the reference builds in
`RetinaUNetModule.from_config_plan` (nndet/ptmodule/retinaunet/base.py:338-466) for
`RetinaUNetV001` (nndet/ptmodule/retinaunet/v001.py:29-38) with `nndet/conf/train/v001.yaml:61-107`:

  encoder   nndet/arch/encoder/modular.py:28-126 + blocks/basic.py:127-151 (2 x [conv3 -> IN -> ReLU] per stage)
  decoder   nndet/arch/decoder/base.py:391-417 (lateral 1^3, top-down ConvTranspose k=s, out 3^3; bias, no norm/act)
  heads     nndet/arch/heads/classifier.py:116-181, regressor.py:99-173 (conv3+GN+ReLU x2 -> conv3; Scale per level)
  segmenter nndet/arch/heads/segmenter.py:121-182 (1^3 conv on P0 -> 2 logits)
  step      nndet/core/retina.py:86-159,228-379 ; losses nndet/arch/heads/comb.py:351-405,
            nndet/losses/{classification.py:137-181, regression.py:118-162, segmentation.py:32-151}

The module tree reproduces the reference's state-dict keys exactly so one state
dict loads into the reference, this oracle and the HIP model. Pinned against the real reference by
tests/golden/make_golden.py (losses, gradients, detections at fixed weights/inputs).
"""
import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import boxes_np as bx


def _cna(cin, cout, k, stride=1, pad=0, norm=None, act=False, bias=None, transposed=False, cpg=16):
    """conv -> norm -> act with children named conv / norm / act (nndet/arch/conv.py:54-143)."""
    m = nn.Sequential()
    bias = (norm is None) if bias is None else bias
    conv_cls = nn.ConvTranspose3d if transposed else nn.Conv3d
    m.add_module("conv", conv_cls(cin, cout, k, stride=stride, padding=pad, bias=bias))
    if norm == "instance":
        m.add_module("norm", nn.InstanceNorm3d(cout, eps=1e-5, affine=True))
    elif norm == "group":
        m.add_module("norm", nn.GroupNorm(cout // cpg, cout, eps=1e-5, affine=True))
    if act:
        m.add_module("act", nn.ReLU(inplace=norm is not None))
    return m


class _Holder(nn.Module):
    """Attribute container so nested names match the reference tree."""


class OracleRetinaUNet(nn.Module):
    def __init__(self, plan_arch: dict, plan_anchors: dict, model_cfg: dict):
        super().__init__()
        pa = plan_arch
        self.plan_arch, self.plan_anchors, self.model_cfg = pa, plan_anchors, model_cfg
        kernels, strides = pa["conv_kernels"], pa["strides"]
        nst = len(kernels)
        max_ch = pa.get("max_channels", 320)
        # ---- encoder (modular.py:79-108): channels double, capped
        self.encoder = _Holder()
        stages, chans, cin = [], [], pa["in_channels"]
        for s in range(nst):
            cout = pa["start_channels"] if s == 0 else min(cin * 2, max_ch)
            k = tuple(kernels[s]); pad = tuple((i - 1) // 2 for i in k)
            st = 1 if s == 0 else tuple(strides[s - 1])
            blk = _Holder()
            blk.convs = nn.Sequential(nn.Sequential(
                _cna(cin, cout, k, st, pad, "instance", True),
                _cna(cout, cout, k, 1, pad, "instance", True)))
            stages.append(blk); chans.append(cout); cin = cout
        self.encoder.stages = nn.ModuleList(stages)
        self.enc_channels = chans
        # ---- decoder (decoder/base.py:28-313)
        dl = tuple(pa["decoder_levels"])
        self.decoder_levels = dl
        dk = model_cfg["decoder_kwargs"]
        outc = [pa["fpn_channels"]] * nst
        for ol in [l for l in range(nst) if l < min(dl)][::-1]:
            outc[ol] = max(dk.get("min_out_channels", 8), outc[ol + 1] // 2)
        self.dec_channels = outc
        self.decoder = _Holder()
        self.decoder.lateral = nn.ModuleDict({f"P{l}": nn.Sequential(_cna(chans[l], outc[l], 1)) for l in range(nst)})
        self.decoder.out = nn.ModuleDict({
            f"P{l}": nn.Sequential(_cna(outc[l], outc[l], tuple(kernels[l]), 1, tuple((i - 1) // 2 for i in kernels[l])))
            for l in range(nst)})
        self.decoder.up = nn.ModuleDict({
            f"P{l}": _cna(outc[l], outc[l - 1], tuple(strides[l - 1]), tuple(strides[l - 1]), 0, transposed=True)
            for l in range(1, nst)})
        # ---- heads
        self.A = len(plan_anchors["width"][0]) * len(plan_anchors["height"][0]) * len(plan_anchors["depth"][0])
        self.C = pa["classifier_classes"]
        fc, hc = pa["fpn_channels"], pa["head_channels"]
        ck, rk = model_cfg["head_classifier_kwargs"], model_cfg["head_regressor_kwargs"]

        def trunk(n, cpg):
            t = nn.Sequential()
            t.add_module("c_in", _cna(fc, hc, 3, 1, 1, "group", True, cpg=cpg))
            for i in range(n):
                t.add_module(f"c_internal{i}", _cna(hc, hc, 3, 1, 1, "group", True, cpg=cpg))
            return t
        self.head = _Holder()
        self.head.classifier = _Holder()
        self.head.classifier.conv_internal = trunk(ck["num_convs"], ck["norm_channels_per_group"])
        self.head.classifier.conv_out = _cna(hc, self.A * self.C, 3, 1, 1, bias=True)
        self.head.regressor = _Holder()
        self.head.regressor.conv_internal = trunk(rk["num_convs"], rk["norm_channels_per_group"])
        self.head.regressor.conv_out = _cna(hc, self.A * 6, 3, 1, 1, bias=True)
        sc = []
        for _ in dl:
            h = _Holder(); h.scale = nn.Parameter(torch.tensor(1.0)); sc.append(h)
        self.head.regressor.scales = nn.ModuleList(sc)
        # init (classifier.py:210-228, regressor.py:194-201)
        for hd in (self.head.classifier, self.head.regressor):
            for layer in hd.modules():
                if isinstance(layer, nn.Conv3d):
                    nn.init.normal_(layer.weight, mean=0, std=0.01)
                    if layer.bias is not None:
                        nn.init.constant_(layer.bias, 0)
        prior = ck.get("prior_prob", 0.01)
        nn.init.constant_(self.head.classifier.conv_out.conv.bias, -math.log((1 - prior) / prior))
        # ---- segmenter (DiCESegmenterFgBg -> 2 logits)
        self.segmenter = _Holder()
        self.segmenter.conv_out = _cna(outc[0], 2, 1, 1, 0, bias=True)
        # post-processing constants (retinaunet/base.py:436-440)
        self.detections_per_img = pa.get("detections_per_img", 100)
        self.score_thresh = pa.get("score_thresh", 0)
        self.topk_candidates = pa.get("topk_candidates", 10000)
        self.remove_small_boxes = pa.get("remove_small_boxes", 0.01)
        self.nms_thresh = pa.get("nms_thresh", 0.6)
        self._anchor_cache = {}

    # ------------------------------------------------------------------ forward
    def features(self, x):
        enc = []
        for st in self.encoder.stages:
            x = st.convs(x)
            enc.append(x)
        lat = [self.decoder.lateral[f"P{l}"](f) for l, f in enumerate(enc)]
        n = len(lat)
        xs: List[Optional[torch.Tensor]] = [None] * n
        up = None
        for l in range(n - 1, -1, -1):
            x = lat[l] if up is None else lat[l] + up
            if l > 0:
                up = self.decoder.up[f"P{l}"](x)
            xs[l] = x
        return [self.decoder.out[f"P{l}"](xs[l]) for l in range(n)]

    def head_forward(self, fmaps):
        logits, deltas = [], []
        for lvl, p in enumerate(fmaps):
            n = p.shape[0]
            c = self.head.classifier.conv_out(self.head.classifier.conv_internal(p))
            logits.append(c.permute(0, 2, 3, 4, 1).contiguous().view(n, -1, self.C))
            r = self.head.regressor.conv_out(self.head.regressor.conv_internal(p))
            r = r * self.head.regressor.scales[lvl].scale
            deltas.append(r.permute(0, 2, 3, 4, 1).contiguous().view(n, -1, 6))
        return {"box_deltas": torch.cat(deltas, 1).reshape(-1, 6), "box_logits": torch.cat(logits, 1).flatten(0, -2)}

    def anchors(self, image_size, fmap_sizes):
        key = str((tuple(image_size), tuple(map(tuple, fmap_sizes))))
        if key not in self._anchor_cache:
            p = self.plan_anchors
            self._anchor_cache[key] = bx.anchors_for_image(image_size, fmap_sizes, p["width"], p["height"], p["depth"])
        return self._anchor_cache[key]

    def forward(self, x):
        fm = self.features(x)
        fm_head = [fm[i] for i in self.decoder_levels]
        pred = self.head_forward(fm_head)
        anchors, npl = self.anchors(x.shape[2:], [f.shape[2:] for f in fm_head])
        seg = {"seg_logits": self.segmenter.conv_out(fm[0])}
        return pred, anchors, npl, seg

    # ------------------------------------------------------------------ training step
    def select_indices(self, labels_cat: torch.Tensor, logits: torch.Tensor, batch_size: int):
        """comb.py:247-276 + sampler.py:237-270 (RNG = torch.randperm, as the reference)."""
        sk = self.model_cfg["head_sampler_kwargs"]
        probs = torch.sigmoid(logits).max(dim=1)[0]
        positive = torch.where(labels_cat >= 1)[0]
        negative = torch.where(labels_cat == 0)[0]
        num_pos, num_neg, pool = bx.hnm_counts(positive.numel(), negative.numel(), batch_size,
                                               sk["batch_size_per_image"], sk["positive_fraction"],
                                               sk.get("min_neg", 0), sk.get("pool_size", 10))
        perm1 = torch.randperm(positive.numel(), device=positive.device)[:num_pos]
        pos_mask = torch.zeros_like(labels_cat, dtype=torch.uint8); pos_mask[positive[perm1]] = 1
        _, pool_idx = probs[negative].topk(pool, sorted=True)
        negp = negative[pool_idx]
        perm2 = torch.randperm(negp.numel(), device=negp.device)[:num_neg]
        neg_mask = torch.zeros_like(labels_cat, dtype=torch.uint8); neg_mask[negp[perm2]] = 1
        return torch.where(pos_mask)[0], torch.where(neg_mask)[0]

    def assign(self, images_shape, targets):
        """ATSS target assignment of a batch with the numpy restatement (retina.py:228-290): (labels, matched boxes, anchors), each
        concatenated over the batch. Split from `train_step` so that bench.py's same-GPU leg (stock torch modules on the GPU) can
        compute it once on the host and hand the device copies in."""
        P = tuple(images_shape[2:])
        st = [[1, 1, 1]]
        for s_ in self.plan_arch["strides"]:
            st.append([a * b for a, b in zip(st[-1], s_)])
        fm = [tuple(-(-P[a] // st[l][a]) for a in range(3)) for l in self.decoder_levels]
        anchors, npl = self.anchors(P, fm)
        labels, matched = [], []
        for gb, gc in zip(targets["target_boxes"], targets["target_classes"]):
            gb, gc = gb.detach().cpu().numpy(), gc.detach().cpu().numpy()
            _, m = bx.atss_match(gb, anchors, npl, self.A, self.model_cfg["matcher_kwargs"]["num_candidates"])
            lab, mb = bx.assign_targets(m, gb, gc, anchors.shape[0])
            labels.append(torch.from_numpy(lab)); matched.append(torch.from_numpy(mb))
        return torch.cat(labels), torch.cat(matched), torch.from_numpy(anchors).repeat(len(labels), 1)

    def train_step(self, images, targets, evaluation: bool = False, assigned=None):
        pred, anchors, npl, seg = self(images)
        B = images.shape[0]
        if assigned is None:
            labels, matched = [], []
            for gb, gc in zip(targets["target_boxes"], targets["target_classes"]):
                _, m = bx.atss_match(gb.numpy(), anchors, npl, self.A, self.model_cfg["matcher_kwargs"]["num_candidates"])
                lab, mb = bx.assign_targets(m, gb.numpy(), gc.numpy(), anchors.shape[0])
                labels.append(torch.from_numpy(lab)); matched.append(torch.from_numpy(mb))
            labels_cat = torch.cat(labels); matched_cat = torch.cat(matched)
            anchors_t = torch.from_numpy(anchors).repeat(B, 1)
        else:
            labels_cat, matched_cat, anchors_t = assigned
        logits, deltas = pred["box_logits"], pred["box_deltas"]
        with torch.no_grad():
            pos, neg = self.select_indices(labels_cat, logits, B)
            sampled = torch.cat([pos, neg])
        losses = {}
        if pos.numel() > 0:
            pb = decode_single_t(deltas[pos], anchors_t[pos])
            giou = giou_t(pb, matched_cat[pos], eps=1e-7)
            losses["reg"] = -1 * torch.diag(giou).sum() / max(1, pos.numel())
        onehot = F.one_hot(labels_cat[sampled].long(), self.C + 1)[:, 1:].float()
        losses["cls"] = F.binary_cross_entropy_with_logits(logits[sampled], onehot)
        tseg = (targets["target_seg"] > 0).long()
        sl = seg["seg_logits"]
        losses["seg_ce"] = 0.5 * F.cross_entropy(sl, tseg)
        p = torch.softmax(sl, 1)
        oh = torch.zeros_like(p).scatter_(1, tseg[:, None], 1)
        ax = [0, 2, 3, 4]
        tp = (p * oh).sum(ax); fp = (p * (1 - oh)).sum(ax); fn = ((1 - p) * oh).sum(ax)
        dc = (2 * tp + 1e-5) / (2 * tp + fp + fn + 1e-5)
        losses["seg_dice"] = 0.5 * (1 - dc[1:].mean())
        prediction = self.postprocess(images, pred, anchors, seg) if evaluation else None
        return losses, prediction

    @torch.no_grad()
    def postprocess(self, images, pred, anchors, seg):
        B = images.shape[0]
        M = anchors.shape[0]
        deltas = pred["box_deltas"].detach().numpy().reshape(B, M, 6)
        probs = torch.sigmoid(pred["box_logits"].detach()).numpy().reshape(B, M, self.C)
        out = {"pred_boxes": [], "pred_scores": [], "pred_labels": []}
        for b in range(B):
            boxes = bx.decode_single(deltas[b], anchors)
            bb, pp, ll = bx.postprocess_single_image(
                boxes, probs[b], images.shape[2:], self.C, self.topk_candidates, self.score_thresh,
                self.remove_small_boxes, self.nms_thresh, self.detections_per_img)
            out["pred_boxes"].append(bb); out["pred_scores"].append(pp); out["pred_labels"].append(ll)
        out["pred_seg"] = torch.softmax(seg["seg_logits"], 1)
        return out

    @torch.no_grad()
    def inference_step(self, images):
        pred, anchors, npl, seg = self(images)
        return self.postprocess(images, pred, anchors, seg)


def decode_single_t(rel, boxes, clip=bx.BBOX_XFORM_CLIP):
    """differentiable decode (coder.py:90-155, weights all 1)."""
    w = boxes[:, 2] - boxes[:, 0]; h = boxes[:, 3] - boxes[:, 1]; d = boxes[:, 5] - boxes[:, 4]
    cx = boxes[:, 0] + 0.5 * w; cy = boxes[:, 1] + 0.5 * h; cz = boxes[:, 4] + 0.5 * d
    dw = torch.clamp(rel[:, 2], max=clip); dh = torch.clamp(rel[:, 3], max=clip); dd = torch.clamp(rel[:, 5], max=clip)
    pcx = rel[:, 0] * w + cx; pcy = rel[:, 1] * h + cy; pcz = rel[:, 4] * d + cz
    pw = torch.exp(dw) * w; ph = torch.exp(dh) * h; pd = torch.exp(dd) * d
    return torch.stack([pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph, pcz - 0.5 * pd, pcz + 0.5 * pd], 1)


def giou_t(b1, b2, eps=0.0):
    """differentiable generalized_box_iou_3d (ops.py:162-185; eps only on the hull volume)."""
    v1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1]) * (b1[:, 5] - b1[:, 4])
    v2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1]) * (b2[:, 5] - b2[:, 4])
    lo = lambda i: torch.max(b1[:, None, i], b2[:, i])
    hi = lambda i: torch.min(b1[:, None, i], b2[:, i])
    inter = (hi(2) - lo(0)).clamp(min=0) * (hi(3) - lo(1)).clamp(min=0) * (hi(5) - lo(4)).clamp(min=0)
    union = v1[:, None] + v2 - inter
    iou = inter / union
    LO = lambda i: torch.min(b1[:, None, i], b2[:, i])
    HI = lambda i: torch.max(b1[:, None, i], b2[:, i])
    vol = (HI(2) - LO(0)).clamp(min=0) * (HI(3) - LO(1)).clamp(min=0) * (HI(5) - LO(4)).clamp(min=0) + eps
    return iou - (vol - union) / vol
