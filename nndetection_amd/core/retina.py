"""Retina(U)Net detector: forward, target assignment, losses, post-processing. Mirrors BaseRetinaNet
(nndet/core/retina.py:30-414): same constructor, `train_step(images, targets, evaluation, batch_num)`,
`inference_step(images)`, `forward(inp)`, loss keys (reg, cls, seg_ce, seg_dice) and prediction keys."""
from typing import Any, Dict, List, Optional, Tuple

import os
import torch
import torch.nn as nn
from torch import Tensor

from .. import _lib as L
from . import boxes as box_utils


# NNDET_REPACK_EVERY_STEP=0: trust the parameters' version counters alone (valid with optimizers that advance them: torch's foreach /
# single-tensor ones and nndetection_amd.optim.SGDNesterov), saves the re-pack launch of steps that follow no optimizer step.
REPACK_EVERY_STEP = os.environ.get("NNDET_REPACK_EVERY_STEP", "1") != "0"


class MatchedBoxes:
    """The reference's `matched_gt_boxes` (one [M, 6] tensor per image: the GT box each anchor was matched to, retina.py:262-287)
    WITHOUT the gather: the GT boxes of the batch, the ATSS matches [B, M] (index local to the image, -1 = none) and the first GT row of
    each image. The detection loss reads the boxes of the <= 42 sampled positives through the matches (nndet_detloss_matched_f32);
    `materialize()` gives the reference's list for any other consumer (114 MB per step at 160x160x96, batch 4)."""

    def __init__(self, gt_all: Tensor, matches: Tensor, offsets):
        self.gt_all, self.matches, self.offsets = gt_all, matches, [int(o) for o in offsets]

    def __len__(self):
        return self.matches.shape[0]

    def tensors(self):
        return [self.gt_all, self.matches]

    def materialize(self) -> List[Tensor]:
        B, M = self.matches.shape
        if self.gt_all.shape[0] == 0:
            return [torch.zeros((M, 6), dtype=torch.float32, device=self.matches.device) for _ in range(B)]
        last = self.gt_all.shape[0] - 1
        base = torch.tensor([min(o, last) for o in self.offsets[:-1]], dtype=torch.int64, device=self.matches.device)[:, None]
        boxes = self.gt_all[self.matches.clamp(min=0) + base]
        for b in range(B):
            if self.offsets[b + 1] == self.offsets[b]:
                boxes[b].zero_()
        return list(boxes.unbind(0))


# NNDET_LAZY_TARGETS=0: labels / matched boxes of the batched target assignment as the clamp / gather / compare chain of round 3
LAZY_TARGETS = os.environ.get("NNDET_LAZY_TARGETS", "1") != "0"
# NNDET_SEG_FIRST=1: the segmentation branch + loss queued BEFORE the detection head, so that the backward pass issues the head's nodes
# first (the engine runs ready nodes in decreasing creation order). Built in round 5 from the profiler's timeline, where the main stream
# idles ~1.1 ms while the branch's ~45 small backward launches are issued -- but that gap is the PROFILER's: unprofiled, the host is two to
# three steps ahead of the GPU (tools/host_lead.py). Measured 13.05 / 13.01 ms with it vs 12.54 / 12.55 without
# (profiles/round5_ab_seg_first.txt): the branch's HBM-bound forward kernels then run under the head trunks instead of under the sampler's
# latency-bound passes. OFF by default.
SEG_FIRST = os.environ.get("NNDET_SEG_FIRST", "0") != "0"


class BaseRetinaNet(nn.Module):
    def __init__(self, dim: int, encoder, decoder, head, num_classes: int, anchor_generator, matcher,
                 decoder_levels: tuple = (2, 3, 4, 5), score_thresh: float = None, detections_per_img: int = 100,
                 topk_candidates: int = 10000, remove_small_boxes: float = 1e-2, nms_thresh: float = 0.9,
                 segmenter=None):
        super().__init__()
        assert dim == 3
        self.dim = dim
        self.decoder_levels = decoder_levels
        self.encoder, self.decoder, self.head = encoder, decoder, head
        self.num_foreground_classes = num_classes
        self.anchor_generator = anchor_generator
        self.proposal_matcher = matcher
        self.score_thresh, self.topk_candidates = score_thresh, topk_candidates
        self.detections_per_img, self.remove_small_boxes, self.nms_thresh = detections_per_img, remove_small_boxes, nms_thresh
        self.segmenter = segmenter
        if hasattr(self.decoder, "used_levels") and self.decoder.used_levels is None:
            used = set(decoder_levels)
            if segmenter is not None:
                used.add(0)
            self.decoder.used_levels = used          # out convs nobody reads are not computed
        from ..arch.decoder import UFPNModular
        if isinstance(self.decoder, UFPNModular) and hasattr(self.encoder, "set_defer_outputs"):
            self.encoder.set_defer_outputs(True)     # the decoder's lateral convs apply the encoder's norm + ReLU on load
            if hasattr(self.encoder, "set_fuse_grad_accum"):
                self.encoder.set_fuse_grad_accum(os.environ.get("NNDET_FUSE_GRAD_ACC", "1") != "0")
            if hasattr(self.encoder, "set_early_consumer"):
                self.encoder.set_early_consumer(True)    # (arch/conv.py: only with NNDET_EARLY_CONSUMER=1, measured slower)
            if hasattr(self.encoder, "stage_hook") and hasattr(self.decoder, "early_lateral") and \
                    list(getattr(self.encoder, "out_stages", [])) == list(range(self.decoder.num_level)):
                self.encoder.stage_hook = self.decoder.early_lateral      # laterals start under the deeper encoder stages

    def train(self, mode: bool = True):
        """Switching between training and evaluation drops every packed-weight cache: the parameters may have been written since by a
        kernel that does not advance their version counters (fused optimizers, see arch/conv.py: prepack_all)."""
        if mode != self.training:
            self.__dict__.pop("_nndet_conv_blocks", None)      # (arch/conv.py prepack_all: the cached list of conv blocks is rebuilt)
            self.__dict__.pop("_nndet_pack_args", None)
            for m in self.modules():
                if hasattr(m, "_pack_cache"):
                    m._pack_cache.clear()
            from ..arch.conv import PACK_EPOCH
            PACK_EPOCH[0] += 1                   # (caches keyed on pairs of blocks, arch/pyramid.py)
        return super().train(mode)

    def never_used_parameters(self) -> List[nn.Parameter]:
        """Parameters that exist for state-dict parity with the reference but never receive a gradient: the decoder output convs
        of pyramid levels that neither the detection head nor the segmenter reads (`decoder.out.P1.*` for RetinaUNetV001)."""
        used = getattr(self.decoder, "used_levels", None)
        out = getattr(self.decoder, "out", None)
        if used is None or out is None or not getattr(self.decoder, "skip_unused_out", True):
            return []
        res = []
        for name, mod in out.items():
            if int(str(name).lstrip("P")) not in used:
                res.extend(mod.parameters())
        return res

    # Which side streams share one of the runtime's four hardware queues decides 0.1 - 1.6 ms of a 11.6 ms step (profiles/round6_stream_mapping.txt),
    # and the runtime binds a stream to a queue when it is created / first used: the FIRST forward pass on a device creates and uses every
    # side stream of the package in the order a training step would (whatever this first call is -- a validation pass that touches only two
    # of them would otherwise leave the rest to be bound later, in another order). NNDET_BIND_STREAMS=0: bound as they come.
    _bound_devices: set = set()
    _bind_order = [k for k in os.environ.get("NNDET_BIND_STREAMS", "aux0,aux1,tail,head,wgrad").split(",") if k and k != "0"]

    def _bind_streams(self, device) -> None:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if idx in BaseRetinaNet._bound_devices:
            return
        BaseRetinaNet._bound_devices.add(idx)
        from ..arch.decoder import UFPNModular
        for kind in BaseRetinaNet._bind_order:
            st = None
            if kind in ("aux0", "aux1"):
                if self.overlap_aux and (kind == "aux0" or self.segmenter is not None):
                    st = self._aux(device, int(kind[3]))
            elif kind == "tail" and isinstance(self.decoder, UFPNModular) and self.decoder.split_tail:
                st = UFPNModular._tail_streams.get(idx)
                if st is None:
                    st = UFPNModular._tail_streams[idx] = L.new_stream("tail", device, int(os.environ.get("NNDET_PRIO_TAIL", "0")))
            elif kind == "head" and hasattr(self.head, "_side_streams") and getattr(self.head, "multi_stream", False):
                st = self.head._side_streams(device, 1)[0]
            elif kind == "wgrad" and L.wgrad_streams.enabled:
                st = L.wgrad_streams.stream_for(device)
            if st is not None:
                with torch.cuda.stream(st):
                    torch.zeros(1, device=device)                # (first use)

    # ------------------------------------------------------------------ forward (retina.py:198-226)
    def forward(self, inp: Tensor):
        if inp.is_cuda and BaseRetinaNet._bind_order:
            self._bind_streams(inp.device)
        L.arena_reset(inp.device)                # one fill for all per-layer statistics buffers of the previous step
        inp = L.autocast_input(inp)              # under torch.autocast: the image in the autocast dtype (B2: conv in half precision)
        if inp.is_cuda:
            from ..arch.conv import prepack_all
            grad = torch.is_grad_enabled()
            # all packed weights in one launch. In training mode they are re-packed unconditionally: an optimizer stepped since, and
            # fused optimizer kernels do not advance the parameters' version counters (arch/conv.py: prepack_all)
            prepack_all(self, inp.dtype if inp.dtype in L._DT else torch.float32,
                        modes=(0, 1) if grad else (0,), force=self.training and REPACK_EVERY_STEP)
            if grad:                                           # one zero fill for every parameter-gradient accumulator of the step
                if getattr(self, "_grad_numel", None) is None:
                    self._grad_numel = sum(p.numel() + 64 for p in self.parameters() if p.requires_grad)
                L.grad_pool.begin(self._grad_numel, inp.device, owner=self)
                # side channels of the PREVIOUS backward pass (sparse-gradient hints, the factorised segmentation gradient): entries
                # nobody consumed would pin large tensors across steps. Their entries live from one node of a backward
                # pass to a later node of the SAME pass, so a forward pass that runs INSIDE a backward pass (activation checkpointing
                # recomputes, a second model driven from a hook) must leave them alone: cleared only at top level.
                if L.graph_task_id() == -1:
                    L.grad_hints.clear()
                    from ..arch.conv import _rank1_grads, _norm_presums
                    _rank1_grads.clear()
                    _norm_presums.clear()
        if hasattr(self.decoder, "defer_out0"):                # decoder.out.P0 + segmentation head + loss as one 32 -> 1 convolution?
            self.decoder.defer_out0 = self._seg_branch_ok(inp)
            self.decoder.absorb_lat0 = self.decoder.defer_out0 and self._seg_lateral_ok()
            self.decoder.absorb_up0 = self.decoder.absorb_lat0 and self._seg_up_ok(inp)
        try:
            features_maps_all = self.decoder(self.encoder(inp))
        finally:
            if hasattr(self.decoder, "defer_out0"):
                self.decoder.defer_out0 = self.decoder.absorb_lat0 = self.decoder.absorb_up0 = False
        feature_maps_head = [features_maps_all[i] for i in self.decoder_levels]
        tail_ev = getattr(self.decoder, "tail_event", None)    # level 0 (the segmenter's input) comes from the decoder's side stream
        if getattr(self, "_seg_side", None) is not None:       # train_step: the segmentation branch forks HERE (before the head is queued)
            if tail_ev is not None:
                self._dec_event = tail_ev
            else:
                self._dec_event = torch.cuda.Event()
                self._dec_event.record()
        fused = self.segmenter is not None and (getattr(self, "_fuse_seg_head", False) or
                                                (getattr(self, "_seg_infer", False) and getattr(features_maps_all[0], "_nndet_pre_out", None) is not None))
        # does the segmenter consume level 0 on its side stream (fused head + loss, train_step) or on THIS stream (inference,
        # evaluation, NNDET_SEG_FUSED=0, wide level 0)? Decided from what it will actually do, not from a flag a previous step may
        # have left behind: only the former may skip the join with the decoder's tail stream.
        seg_on_side = fused and getattr(self, "_seg_side", None) is not None and \
            bool(getattr(self.segmenter, "takes_fused_route", lambda fm: False)(features_maps_all))
        pred_seg = None
        # Segmentation branch FIRST (round 5): its autograd nodes are created before the head's, so the backward pass -- the engine
        # runs ready nodes in decreasing creation order -- issues the whole detection-head backward before the ~45 small launches of
        # the branch's backward. In the reference's order (loss of the branch computed last = its backward first) the main stream had
        # nothing queued for ~1.1 ms while the host issued those launches onto the branch's stream (profiles/round4_v3_timeline_one_step.txt
        # @5967-7135 us). Only on the route train_step forks onto the side stream; NNDET_SEG_FIRST=0 restores the old order.
        seg_target = getattr(self, "_seg_target", None)
        if SEG_FIRST and seg_on_side and seg_target is not None and torch.is_grad_enabled():
            if features_maps_all[0] is not None and self._seg_rank1_ok():
                features_maps_all[0]._nndet_rank1_ok = True
            pred_seg = self.segmenter(features_maps_all, fused=True)
            if isinstance(pred_seg, dict) and "seg_input" in pred_seg:
                self._seg_losses_early = self._seg_loss_on_side(pred_seg, seg_target)
        pred_detection = self.head(feature_maps_head)
        anchors = self.anchor_generator(inp, feature_maps_head)
        if tail_ev is not None and inp.is_cuda:
            if not seg_on_side:                                # level 0 is consumed on this stream: join here,
                torch.cuda.current_stream(inp.device).wait_event(tail_ev)      # behind the head that was queued in the meantime
            for t in features_maps_all[:1]:
                if t is not None:
                    t.record_stream(torch.cuda.current_stream(inp.device))
        if self.segmenter is not None and pred_seg is None:
            if fused and features_maps_all[0] is not None and self._seg_rank1_ok():
                features_maps_all[0]._nndet_rank1_ok = True      # its gradient may travel as d1 (x) (w1 - w0): arch/conv.py
            pred_seg = self.segmenter(features_maps_all, fused=True) if fused else self.segmenter(features_maps_all)
        return pred_detection, anchors, pred_seg

    def _seg_loss_on_side(self, pred_seg, target_seg):
        """The fused segmentation branch + loss on its side stream (forked behind the decoder, not behind the head)."""
        seg_s = self._seg_side
        seg_s.wait_event(self._dec_event)            # the decoder output is ready; the head queued behind it is not waited for
        for v in pred_seg.values():
            v.record_stream(seg_s)
            lat = getattr(v, "_nndet_pre_lat", None)     # the absorbed lateral's input is read on that stream too
            if lat is not None:
                lat[1].record_stream(seg_s)
        target_seg.record_stream(seg_s)
        with torch.cuda.stream(seg_s):
            return self.segmenter.compute_loss(pred_seg, target_seg)

    def _seg_branch_ok(self, inp: Tensor) -> bool:
        """May this forward pass skip decoder.out.P0 and leave the whole segmentation branch to `_SegBranchFn` (csrc/segbranch.hip)?
        A training step without prediction (`_fuse_seg_head`), 16-bit activations on the GPU, a plain 32 -> 32 3x3x3 output
        convolution whose result only the 2-class segmenter reads."""
        from ..arch import segmenter as S
        training_step = getattr(self, "_fuse_seg_head", False) and torch.is_grad_enabled()
        if not (S.SEG_BRANCH and self.segmenter is not None and (training_step or getattr(self, "_seg_infer", False)) and inp.is_cuda
                and inp.dtype in (torch.bfloat16, torch.float16) and self._seg_rank1_ok()):
            return False
        mod = self.decoder.out["P0"][0]
        seg = self.segmenter
        return bool(mod.in_channels == 32 and mod.out_channels == 32 and list(getattr(seg, "in_channels", [0]))[0] == 32
                    and getattr(seg, "conv_out", None) is not None and seg.conv_out.out_channels == 2
                    and os.environ.get("NNDET_SEG_FUSED", "1") != "0")

    def _seg_lateral_ok(self) -> bool:
        """... and the level-0 lateral with it (arch/segmenter.py: SEG_LATERAL): a plain 1x1x1 32 -> 32 convolution without norm."""
        from ..arch import segmenter as S
        from ..arch.conv import BaseConvNormAct
        lat = getattr(self.decoder, "lateral", None)
        blk = lat["P0"] if (lat is not None and "P0" in lat) else None
        if not S.SEG_LATERAL or blk is None or len(list(blk.children())) != 1 or not isinstance(blk[0], BaseConvNormAct):
            return False
        m, up = blk[0], getattr(self.decoder, "up", {})
        return bool(m.norm_groups == 0 and not m.transposed and m.k == (1, 1, 1) and m.s == (1, 1, 1) and m.in_channels == 32
                    and m.out_channels == 32 and not getattr(m, "relu", False) and "P1" in up and up["P1"].norm_groups == 0)

    def _seg_up_ok(self, inp: Tensor) -> bool:
        """... and the last top-down step (arch/segmenter.py: SEG_UP): a plain k = s = 2 transposed convolution onto the 32 channels of
        level 0 whose input (decoder level 1) nobody else reads, even patch dims."""
        from ..arch import segmenter as S
        up = getattr(self.decoder, "up", {})
        if not S.SEG_UP or "P1" not in up or inp.dim() != 5 or any(int(d) % 2 for d in inp.shape[2:]):
            return False
        m = up["P1"]
        used = getattr(self.decoder, "used_levels", None)
        return bool(getattr(m, "transposed", False) and m.norm_groups == 0 and m.k == (2, 2, 2) and m.s == (2, 2, 2) and m.p == (0, 0, 0)
                    and m.out_channels == 32 and m.in_channels % 32 == 0 and not getattr(m, "relu", False)
                    and 1 not in tuple(self.decoder_levels) and used is not None and 1 not in used
                    and getattr(self.decoder, "skip_unused_out", True))

    def _seg_rank1_ok(self) -> bool:
        """True if decoder level 0 is produced by one of our plain 3x3x3 / stride-1 convolutions (no norm) and read by the segmenter
        ONLY -- then the segmentation head may hand its input gradient back in factorised form (arch/conv.py: _rank1_backward)."""
        ok = getattr(self, "_seg_rank1_cached", None)
        if ok is None:
            from ..arch.conv import BaseConvNormAct
            out = getattr(self.decoder, "out", None)
            blk = out["P0"] if (out is not None and "P0" in out) else None
            mods = [m for m in blk.modules() if isinstance(m, BaseConvNormAct)] if blk is not None else []
            ok = (0 not in tuple(self.decoder_levels) and len(mods) == 1 and len(list(blk.children())) == 1
                  and mods[0].norm_groups == 0 and not mods[0].transposed and mods[0].k == (3, 3, 3) and mods[0].s == (1, 1, 1)
                  and mods[0].p == (1, 1, 1) and mods[0].out_channels % 32 == 0 and mods[0].in_channels % 32 == 0)
            self._seg_rank1_cached = bool(ok)
            if ok:
                mods[0]._nndet_rank1_consumer = True     # (arch/conv.py: a COPY of the factorised gradient arriving there is an error)
        return ok

    def _lazy_targets_ok(self) -> bool:
        return bool(getattr(self.head, "accepts_matched_boxes", False))

    # Target assignment (ATSS on the anchors + GT boxes) does not depend on the network: with the anchors of the previous step with
    # the same image shape (the generator caches them) it runs on a side stream UNDER the forward pass instead of between the
    # forward and the backward pass (0.3-0.45 ms of small launches per step). The segmentation branch (fused output conv + loss,
    # two HBM-bound streaming passes) runs on another side stream next to the MFMA-bound detection head and its loss.
    # NNDET_OVERLAP_AUX=0: everything on the main stream in the reference's order.
    overlap_aux = os.environ.get("NNDET_OVERLAP_AUX", "1") != "0"
    _aux_streams: Dict[int, list] = {}

    def _aux(self, device, i: int):
        pool = BaseRetinaNet._aux_streams.setdefault(device.index or 0, [])
        while len(pool) <= i:
            pool.append(L.new_stream("aux%d" % len(pool), device, int(os.environ.get("NNDET_PRIO_AUX%d" % len(pool), os.environ.get("NNDET_PRIO_AUX", "0")))))
        return pool[i]

    # ------------------------------------------------------------------ train step (retina.py:86-159)
    def train_step(self, images: Tensor, targets: dict, evaluation: bool, batch_num: int = 0):
        """`targets`: the reference's dict (target_boxes, target_classes, target_seg) or a `core.targets.DeferredTargets`, whose
        box lists become available on the host only after `resolve()`: then the forward pass is queued FIRST and the target
        assignment afterwards (still under the forward pass on the GPU), so the one device -> host read of a plugin training step
        does not drain the queue."""
        from .targets import DeferredTargets
        lazy = targets if isinstance(targets, DeferredTargets) else None
        if lazy is None:
            target_boxes: List[Tensor] = targets["target_boxes"]
            target_classes: List[Tensor] = targets["target_classes"]
            target_seg: Tensor = targets["target_seg"]
        else:
            target_boxes = target_classes = None
            target_seg = lazy.target_seg
        # the segmentation logits are only materialised when a prediction is asked for (evaluation); else conv + loss are fused
        self._fuse_seg_head = (not evaluation) and torch.is_grad_enabled()
        self.head._defer_reg_out = self._fuse_seg_head     # no prediction asked for: box_deltas only at the sampled positives (arch/heads.py)
        overlap = self.overlap_aux and images.is_cuda
        pre, main, cached = None, None, None
        if overlap:
            main = torch.cuda.current_stream(images.device)
            cached = getattr(self, "_anchors_by_shape", {}).get(tuple(images.shape))
            if cached is not None and lazy is None:
                side = self._aux(images.device, 0)
                side.wait_stream(main)                   # the targets (and everything of the previous step) are ready
                with torch.cuda.stream(side):
                    pre = self.assign_targets_to_anchors(cached, target_boxes, target_classes, lazy=self._lazy_targets_ok())
            elif cached is not None:
                self._lazy_fork = torch.cuda.Event()
                self._lazy_fork.record(main)             # the target tensors are ready here; the forward pass is queued behind it
        self._seg_side = self._aux(images.device, 1) if (overlap and self.segmenter is not None and self._fuse_seg_head) else None
        self._seg_target = target_seg if self._seg_side is not None else None
        self._seg_losses_early = None
        try:
            return self._train_step_body(images, lazy, target_boxes, target_classes, target_seg, evaluation, overlap, pre, main, cached)
        finally:                                          # never leave the fork state behind for a later forward() / inference_step()
            self._fuse_seg_head = False
            self.head._defer_reg_out = False
            self._seg_side = None
            self._seg_target = None
            self._seg_losses_early = None

    def _train_step_body(self, images, lazy, target_boxes, target_classes, target_seg, evaluation, overlap, pre, main, cached):
        pred_detection, anchors, pred_seg = self(images)
        if self._seg_side is not None and not (isinstance(pred_seg, dict) and "seg_input" in pred_seg):
            # the segmenter did not take the fused route and read decoder level 0 on the main stream: forward() has joined the
            # decoder's tail stream for it (see there); the loss stays on the main stream as well
            self._seg_side = None
        if lazy is not None:                             # the forward pass is queued: now read the instance counts
            tg = lazy.resolve()
            target_boxes, target_classes = tg["target_boxes"], tg["target_classes"]
            if overlap and cached is not None and len(anchors) == len(cached) and all(a is c for a, c in zip(anchors, cached)):
                side = self._aux(images.device, 0)
                side.wait_event(self._lazy_fork)
                with torch.cuda.stream(side):
                    pre = self.assign_targets_to_anchors(cached, target_boxes, target_classes, lazy=self._lazy_targets_ok())
                    for t in list(target_boxes) + list(target_classes):
                        t.record_stream(side)
        if pre is not None and len(anchors) == len(cached) and all(a is c for a, c in zip(anchors, cached)):
            main.wait_stream(self._aux(images.device, 0))
            labels, matched_gt_boxes = pre
            for t in list(labels) + (matched_gt_boxes.tensors() if isinstance(matched_gt_boxes, MatchedBoxes) else list(matched_gt_boxes)):
                t.record_stream(main)
        else:
            labels, matched_gt_boxes = self.assign_targets_to_anchors(anchors, target_boxes, target_classes, lazy=self._lazy_targets_ok())
        if overlap:
            if not hasattr(self, "_anchors_by_shape"):
                self._anchors_by_shape = {}
            self._anchors_by_shape[tuple(images.shape)] = list(anchors)
        losses = {}
        seg_losses = self._seg_losses_early              # (forward(): the branch was queued before the head, NNDET_SEG_FIRST)
        self._seg_losses_early = None
        if seg_losses is None and self._seg_side is not None:      # segmentation loss on its side stream, next to the detection loss
            seg_losses = self._seg_loss_on_side(pred_seg, target_seg)
        head_losses, pos_idx, neg_idx = self.head.compute_loss(pred_detection, labels, matched_gt_boxes, anchors)
        losses.update(head_losses)
        if seg_losses is not None:
            main.wait_stream(self._seg_side)
            for v in seg_losses.values():
                v.record_stream(main)
            losses.update(seg_losses)
        elif self.segmenter is not None:
            losses.update(self.segmenter.compute_loss(pred_seg, target_seg))
        prediction = None
        if evaluation:
            prediction = self.postprocess_for_inference(images=images, pred_detection=pred_detection,
                                                        pred_seg=pred_seg, anchors=anchors)
        return losses, prediction

    @torch.no_grad()
    def assign_targets_to_anchors(self, anchors: List[Tensor], target_boxes: List[Tensor], target_classes: List[Tensor], lazy: bool = False):
        """retina.py:228-290 with the fused ATSS kernel as matcher. `lazy` (train_step, when the head accepts it): the labels come
        straight out of the ATSS kernel and the matched boxes stay a `MatchedBoxes` (no [B, M, 6] gather)."""
        labels, matched_gt_boxes = [], []
        npl = None
        if len(anchors) > 1 and all(a is anchors[0] for a in anchors) and hasattr(self.proposal_matcher, "match_batch") \
                and len(anchors) <= 64:
            return self._assign_targets_batched(anchors[0], target_boxes, target_classes, lazy=lazy and LAZY_TARGETS)
        for anchors_per_image, gt_boxes, gt_classes in zip(anchors, target_boxes, target_classes):
            if npl is None:
                npl = self.anchor_generator.get_num_acnhors_per_level()
            dev = anchors_per_image.device
            gt_boxes = gt_boxes.to(dev, torch.float32)
            gt_classes = gt_classes.to(dev)
            mq, matched_idxs = self.proposal_matcher(gt_boxes, anchors_per_image, num_anchors_per_level=npl,
                                                     num_anchors_per_loc=self.anchor_generator.num_anchors_per_location()[0])
            matched_idxs = matched_idxs.to(dev)
            if mq.numel() > 0:
                clamped = matched_idxs.clamp(min=0)
                matched_gt_boxes_per_image = gt_boxes[clamped]
                labels_per_image = gt_classes[clamped].to(dtype=anchors_per_image.dtype) + 1
            else:
                matched_gt_boxes_per_image = torch.zeros_like(anchors_per_image)
                labels_per_image = torch.zeros(anchors_per_image.shape[0], dtype=anchors_per_image.dtype, device=dev)
            labels_per_image[matched_idxs == self.proposal_matcher.BELOW_LOW_THRESHOLD] = 0.0
            labels_per_image[matched_idxs == self.proposal_matcher.BETWEEN_THRESHOLDS] = -1.0
            labels.append(labels_per_image)
            matched_gt_boxes.append(matched_gt_boxes_per_image)
        return labels, matched_gt_boxes

    def _assign_targets_batched(self, anchors: Tensor, target_boxes: List[Tensor], target_classes: List[Tensor], lazy: bool = False):
        """Same result as the per-image loop above, with one ATSS pass over the shared anchors for the whole batch."""
        dev, B, M = anchors.device, len(target_boxes), anchors.shape[0]
        npl = self.anchor_generator.get_num_acnhors_per_level()
        if lazy:
            gt_all, matches, offs, labels = self.proposal_matcher.match_batch(
                target_boxes, anchors, npl, self.anchor_generator.num_anchors_per_location()[0], classes=target_classes)
            return list(labels.unbind(0)), MatchedBoxes(gt_all, matches, offs)
        gt_all, matches, offs = self.proposal_matcher.match_batch(
            target_boxes, anchors, npl, self.anchor_generator.num_anchors_per_location()[0])
        if gt_all.shape[0] == 0:
            z = torch.zeros((B, M), dtype=anchors.dtype, device=dev)
            return list(z.unbind(0)), [torch.zeros_like(anchors) for _ in range(B)]
        cls_all = torch.cat([c.to(dev).reshape(-1) for c, b in zip(target_classes, target_boxes) if b.numel() > 0], 0)
        # (pinned + non_blocking: torch.tensor(..., device=dev) copies through the stream and blocks the host until the forward pass
        # in front of it has drained)
        last = gt_all.shape[0] - 1
        base = torch.tensor([min(o, last) for o in offs[:-1]], dtype=torch.int64)
        base = (base.pin_memory().to(dev, non_blocking=True) if dev.type == "cuda" else base.to(dev))[:, None]
        glob = matches.clamp(min=0) + base                                   # [B, M] rows of gt_all
        boxes = gt_all[glob]                                                 # [B, M, 6]
        labels = (cls_all[glob].to(anchors.dtype) + 1) * (matches >= 0).to(anchors.dtype)
        for b in range(B):
            if offs[b + 1] == offs[b]:                                       # image without objects (retina.py:274-281)
                boxes[b].zero_()
                labels[b].zero_()
        return list(labels.unbind(0)), list(boxes.unbind(0))

    # ------------------------------------------------------------------ post-processing (retina.py:161-196,292-379)
    @torch.no_grad()
    def postprocess_for_inference(self, images: Tensor, pred_detection: Dict[str, Tensor], pred_seg, anchors: List[Tensor]):
        image_shapes = [images.shape[2:]] * images.shape[0]
        boxes, probs, labels = self.postprocess_detections(pred_detection, anchors, image_shapes)
        prediction = {"pred_boxes": boxes, "pred_scores": probs, "pred_labels": labels}
        if self.segmenter is not None:
            prediction["pred_seg"] = self.segmenter.postprocess_for_inference(pred_seg)["pred_seg"]
        return prediction

    def postprocess_detections(self, pred_detection: Dict[str, Tensor], anchors: List[Tensor], image_shapes):
        """retina.py:292-330. All images of a batch share the anchors and (patch-based inference) the image shape, so the whole
        batch goes through ONE fused pass (csrc/postproc.hip): top-k on the logits, decode + clip of the <= topk survivors
        only, filters, batched NMS, first `detections_per_img` -- one read of the per-image counts at the end."""
        B = len(anchors)
        same = all(a is anchors[0] for a in anchors) and all(tuple(sh) == tuple(image_shapes[0]) for sh in image_shapes)
        if not same:                                   # images with their own anchors / shapes: one fused pass per image
            per = [len(a) for a in anchors]
            deltas = pred_detection["box_deltas"].split(per, 0)
            logits = pred_detection["box_logits"].split(per, 0)
            res = [self._postprocess_fused(l[None], d[None], a, sh) for l, d, a, sh in zip(logits, deltas, anchors, image_shapes)]
            return [r[0][0] for r in res], [r[1][0] for r in res], [r[2][0] for r in res]
        M = anchors[0].shape[0]
        return self._postprocess_fused(pred_detection["box_logits"].view(B, M, -1), pred_detection["box_deltas"].view(B, M, -1),
                                       anchors[0], image_shapes[0])

    def _postprocess_fused(self, logits: Tensor, deltas: Tensor, anchors: Tensor, image_shape):
        return box_utils.postprocess_batch(
            logits, deltas, anchors, image_shape, self.num_foreground_classes, self.topk_candidates, self.score_thresh,
            self.remove_small_boxes, self.nms_thresh, self.detections_per_img,
            bbox_xform_clip=getattr(self.head.coder, "bbox_xform_clip", box_utils.BBOX_XFORM_CLIP))

    def postprocess_detections_single_image(self, boxes: Tensor, probs: Tensor, image_shape):
        """Same contract as retina.py:332-379 (decoded boxes [M, 6] + probabilities [M, C] of ONE image) on the fused kernel."""
        assert boxes.shape[0] == probs.shape[0]
        b, p, l = box_utils.postprocess_batch(
            probs.reshape(1, boxes.shape[0], -1), boxes.reshape(1, -1, 6), None, image_shape, self.num_foreground_classes,
            self.topk_candidates, self.score_thresh, self.remove_small_boxes, self.nms_thresh, self.detections_per_img,
            scores_are_probs=True)
        return b[0], p[0], l[0]

    # inference_step in 16 bits: the segmentation probabilities come from the training step's composed convolution (decoder level 0 --
    # output convolution, lateral, last top-down step -- and the logits never exist; arch/segmenter.py: logit_difference).
    # NNDET_SEG_INFER_FUSED=0: the separate layers.
    seg_infer_fused = os.environ.get("NNDET_SEG_INFER_FUSED", "1") != "0"

    @torch.no_grad()
    def inference_step(self, images: Tensor, **kwargs) -> Dict[str, Any]:
        self._seg_infer = bool(self.seg_infer_fused and not self.training)
        try:
            pred_detection, anchors, pred_seg = self(images)
        finally:
            self._seg_infer = False
        return self.postprocess_for_inference(images=images, pred_detection=pred_detection, pred_seg=pred_seg, anchors=anchors)
