"""Foreground/background segmentation head with fused CE + SoftDice loss. Mirrors DiCESegmenterFgBg
(nndet/arch/heads/segmenter.py:30-289): one 1x1x1 conv on the finest decoder map -> 2 logits per voxel;
loss = alpha * CE + (1 - alpha) * SoftDice(softmax, batch_dice, no background, smooth 1e-5)
(nndet/losses/segmentation.py:32-151). The per-voxel part of the loss (softmax, CE, tp/fp/fn sums and their
gradient) is two streaming HIP kernels (csrc/segloss.hip); the scalar algebra on the four sums is autograd."""
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn
from torch import Tensor

from .. import _lib as L
from ..layout import phys, logical


class _SegSums(torch.autograd.Function):
    """logits [N,2,D,H,W] (padded NDHWC), target uint8 [N,D,H,W] -> fp32 [4] = (sum CE, tp, fp, fn)."""

    @staticmethod
    def forward(ctx, logits, target_u8):
        lp, c = phys(logits)
        if c != 2:
            raise L.NndetError("the fused segmentation loss is the 2-class (fg/bg) loss of DiCESegmenterFgBg")
        nvox = lp.shape[0] * lp.shape[1] * lp.shape[2] * lp.shape[3]
        sums = torch.zeros((4,), dtype=torch.float64, device=lp.device)
        L.call("nndet_segloss_forward", L.dtype_code(lp), L.ptr(lp), L.ptr(target_u8), nvox, lp.shape[4], L.ptr(sums), L.stream())
        ctx.save_for_backward(lp, target_u8)
        return sums.float()

    @staticmethod
    def backward(ctx, g):
        lp, tgt = ctx.saved_tensors
        nvox = lp.shape[0] * lp.shape[1] * lp.shape[2] * lp.shape[3]
        coeffs = g.detach().float().contiguous()
        dl = torch.empty_like(lp)
        L.call("nndet_segloss_backward", L.dtype_code(lp), L.ptr(lp), L.ptr(tgt), nvox, lp.shape[4], L.ptr(coeffs), L.ptr(dl), L.stream())
        return logical(dl, 2), None


class DiCESegmenterFgBg(nn.Module):
    def __init__(self, conv, seg_classes: int, in_channels: Sequence[int], decoder_levels: Sequence[int],
                 internal_channels: Optional[int] = None, num_internal: int = 0, add_norm: bool = True, add_act: bool = True,
                 kernel_size=3, alpha: float = 0.5, ce_kwargs: Optional[dict] = None, dice_kwargs: Optional[dict] = None, **kwargs):
        super().__init__()
        if num_internal != 0 or ce_kwargs:
            raise NotImplementedError("intermediate segmenter convs / CE kwargs are not used by RetinaUNetV001")
        self.seg_classes = 1 + 1                     # FgBg: always one foreground class (segmenter.py:256-268, :39)
        self.in_channels, self.decoder_levels, self.alpha = in_channels, decoder_levels, alpha
        dk = dict(dice_kwargs or {})
        self.smooth_nom, self.smooth_denom = dk.get("smooth_nom", 1e-5), dk.get("smooth_denom", 1e-5)
        if not dk.get("batch_dice", False) or dk.get("do_bg", False):
            raise NotImplementedError("only batch_dice=True, do_bg=False (v001.yaml:101-103) is fused")
        self.conv_out = conv(in_channels[0], self.seg_classes, kernel_size=1, padding=0, add_norm=None, add_act=None, bias=True)
        self.conv_intermediate = None

    def forward(self, x: List[Tensor]) -> Dict[str, Tensor]:
        return {"seg_logits": self.conv_out(x[0])}

    def compute_loss(self, pred_seg: Dict[str, Tensor], target: Tensor) -> Dict[str, Tensor]:
        logits = pred_seg["seg_logits"]
        tgt = (target > 0).to(torch.uint8).contiguous()          # segmenter.py:288 binarises in place
        s = _SegSums.apply(logits, tgt)
        nvox = float(tgt.numel())
        ce = s[0] / nvox                                          # CrossEntropyLoss mean over voxels
        tp, fp, fn = s[1], s[2], s[3]
        dc = (2 * tp + self.smooth_nom) / (2 * tp + fp + fn + self.smooth_denom)
        return {"seg_ce": self.alpha * ce, "seg_dice": (1 - self.alpha) * (1 - dc)}

    def postprocess_for_inference(self, prediction: Dict[str, Tensor], *args, **kwargs) -> Dict[str, Tensor]:
        return {"pred_seg": torch.softmax(prediction["seg_logits"].float(), dim=1)}
