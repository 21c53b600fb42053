"""Detection heads. Mirrors BCECLassifier / GIoURegressor / DetectionHeadHNMNative
(nndet/arch/heads/classifier.py:64-292, regressor.py:51-310, comb.py:85-158,170-276,351-405) and `Scale`
(nndet/arch/layers/scale.py:21-43); same constructor arguments, module names and loss definitions.
The conv trunks are ConvGroupRelu blocks (HIP), the GIoU of the sampled positives and the fg-probability
used by the sampler run in HIP kernels; the remaining arithmetic acts on <= 32*B sampled rows."""
import math
import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from .. import _lib as L
from ..core.boxes.ops import giou_diag
from ..core.boxes.coder import decode_clip
from ..layout import phys, logical
import ctypes

CONV_TYPES = (nn.Conv2d, nn.Conv3d)


class Scale(nn.Module):
    def __init__(self, scale: float = 1.):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

    def forward(self, inp: Tensor) -> Tensor:
        return inp * self.scale


def _trunk(conv, in_channels, internal_channels, num_convs, **kwargs) -> nn.Sequential:
    t = nn.Sequential()
    t.add_module("c_in", conv(in_channels, internal_channels, kernel_size=3, stride=1, padding=1, **kwargs))
    for i in range(num_convs):
        t.add_module(f"c_internal{i}", conv(internal_channels, internal_channels, kernel_size=3, stride=1, padding=1, **kwargs))
    for m in t:                          # consumed by the next trunk conv / conv_out only: norm + ReLU applied by the consumer on load
        if hasattr(m, "defer_output"):
            m.defer_output = True
    return t


def _flatten_head(x: Tensor, last: int) -> Tensor:
    """[N, A*last, X, Y, Z] -> [N, X*Y*Z*A, last] fp32 (classifier.py:176-181): a plain slice of the NDHWC buffer."""
    n = x.size(0)
    return x.permute(0, 2, 3, 4, 1).contiguous().view(n, -1, last).float()


class _HeadGatherFn(torch.autograd.Function):
    """All pyramid levels of one head branch -> fp32 [N, sum_l positions_l, cout] in one launch (csrc/headio.hip): the
    permute/contiguous/view of classifier.py:176-181 / regressor.py:165-172, the per-level Scale (regressor.py:163-164) and the
    torch.cat over the levels (comb.py:107-108). args = n_scales scalar parameters followed by the per-level conv outputs."""

    @staticmethod
    def forward(ctx, cout: int, n_scales: int, *args):
        scales, xs = args[:n_scales], args[n_scales:]
        ys = [phys(x)[0] for x in xs]
        N, cout_p = ys[0].shape[0], ys[0].shape[4]
        pts = [y.shape[1] * y.shape[2] * y.shape[3] for y in ys]
        sc = [s.detach().float().contiguous() for s in scales]
        out = torch.empty((N, sum(pts), cout), dtype=torch.float32, device=ys[0].device)
        lv = L.NndetHeadLevels()
        lv.nlev = len(ys)
        for l, y in enumerate(ys):
            lv.y[l], lv.points[l] = y.data_ptr(), pts[l]
            lv.scale[l] = sc[l].data_ptr() if n_scales else None
        L.call("nndet_head_gather_f32", L.dtype_code(ys[0]), ctypes.byref(lv), N, cout, cout_p, L.ptr(out), L.stream())
        ctx.cout, ctx.n_scales, ctx.pts = cout, n_scales, pts
        ctx.save_for_backward(*ys, *sc)
        return out

    @staticmethod
    def backward(ctx, g):
        nl = len(ctx.pts)
        ys, sc = ctx.saved_tensors[:nl], ctx.saved_tensors[nl:]
        N, cout_p = ys[0].shape[0], ys[0].shape[4]
        g = g.contiguous().float()
        dys = [torch.empty_like(y) for y in ys]
        dsc = torch.zeros((max(ctx.n_scales, 1),), dtype=torch.float32, device=g.device)
        lv = L.NndetHeadLevels()
        lv.nlev = nl
        for l, y in enumerate(ys):
            lv.y[l], lv.dy[l], lv.points[l] = y.data_ptr(), dys[l].data_ptr(), ctx.pts[l]
            if ctx.n_scales:
                lv.scale[l], lv.dscale[l] = sc[l].data_ptr(), dsc.data_ptr() + 4 * l
        L.call("nndet_head_gather_backward", L.dtype_code(ys[0]), ctypes.byref(lv), N, ctx.cout, cout_p, L.ptr(g), L.stream())
        return (None, None) + tuple(dsc[l].reshape(()) for l in range(ctx.n_scales)) + tuple(logical(d, ctx.cout) for d in dys)


def _cat_rows(ts) -> Tensor:
    """torch.cat(ts, dim=0) -- without the copy when the tensors are the rows base[0], base[1], ... of one contiguous tensor (what
    the batched target assignment returns: labels [B, M] / boxes [B, M, 6] unbound into per-image lists)."""
    if isinstance(ts, Tensor):
        return ts
    if len(ts) == 1:
        return ts[0]
    t0 = ts[0]
    if t0.dim() >= 1 and t0.is_contiguous() and not t0.requires_grad:
        nb, st = t0.numel() * t0.element_size(), t0.untyped_storage().data_ptr()
        if all(t.shape == t0.shape and t.dtype == t0.dtype and t.is_contiguous() and not t.requires_grad
               and t.untyped_storage().data_ptr() == st and t.data_ptr() == t0.data_ptr() + i * nb for i, t in enumerate(ts)):
            return torch.as_strided(t0, (len(ts) * t0.shape[0],) + tuple(t0.shape[1:]), t0.stride(), t0.storage_offset())
    return torch.cat(list(ts), dim=0)


# Sparse backward of the two head OUTPUT convolutions (csrc/sparse_out.hip): the detection loss touches <= 170 sampled anchors, so
# the dense data / weight gradients of conv_out (0.47 TFLOP per step) are replaced by per-entry scatter kernels. NNDET_SPARSE_OUT=0:
# dense backward as in the reference.
SPARSE_OUT = os.environ.get("NNDET_SPARSE_OUT", "1") != "0"


# The regression loss reads box_deltas at the <= 42 sampled positives and nowhere else (comb.py:383-401). In a TRAINING step (no
# prediction asked for) the regressor's output convolution 128 -> 162 is therefore not run over the 4.75 M anchors of the batch at all:
# `_forward_items` hands on its input (the trunk output) as `DeferredDeltas`, and once the sampler has picked the positives the
# convolution is evaluated at exactly those anchors (csrc/sparse_out.hip: k_ho_forward; backward: k_ho_backward). 197 GFLOP forward
# + the [4.75 M, 6] fp32 flatten / cat and its zero-filled gradient disappear. NNDET_SPARSE_REG=0: dense deltas as in the reference.
SPARSE_REG = os.environ.get("NNDET_SPARSE_REG", "1") != "0"
FUSE_CIN = os.environ.get("NNDET_HEAD_FUSE_CIN", "1") != "0"      # one launch for the first layer of both head trunks (arch/pyramid.py)


class DeferredDeltas:
    """box_deltas that have not been computed: the regressor trunk output of the ragged pyramid batch + what is needed to evaluate
    the output convolution + Scale either at sampled anchors (`_RegSparseFn`) or densely (`materialize`, any other consumer)."""

    def __init__(self, t2d, meta, regressor, scales, cout, sdim):
        self.t2d, self.meta, self.regressor, self.scales, self.cout, self.sdim = t2d, meta, regressor, list(scales), cout, sdim

    def materialize(self) -> Tensor:
        from . import pyramid as P
        y = P.items_block(self.regressor.conv_out, self.t2d, self.meta)
        return P.head_gather_items(y, self.meta, self.cout, self.scales).view(-1, self.sdim * 2)


class _RegSparseFn(torch.autograd.Function):
    """Regressor conv_out (+ Scale) at the sampled positives `pos` [P] (-1 = unused slot) -> compact deltas [P, 6] fp32."""

    @staticmethod
    def forward(ctx, t2d, weight, bias, pos, dd, *scales):
        from . import pyramid as P
        meta, mod = dd.meta, dd.regressor.conv_out
        t2d = t2d.contiguous()
        desc = P._items_desc(t2d, mod, meta)
        dev = t2d.device
        K, G = int(pos.shape[0]), dd.sdim * 2
        A = dd.cout // G
        w32 = P.rounded_w32(mod, weight, t2d.dtype)                           # what the dense kernel would multiply with
        b32 = bias.detach().float().contiguous() if bias is not None else None
        sc = [s_.detach().float().contiguous() for s_ in scales]
        pts = [d * h * w for (_, d, h, w) in meta.level_shapes]
        lv = L.NndetHeadLevels()
        lv.nlev = len(pts)
        for l in range(len(pts)):
            lv.points[l] = pts[l]
            lv.scale[l] = sc[l].data_ptr() if sc else None
        row0 = (ctypes.c_int64 * len(pts))(*[r0 for (r0, _) in meta.level_rows])
        out = torch.empty((K, G), dtype=torch.float32, device=dev)
        raw = torch.empty((K, G), dtype=torch.float32, device=dev)
        rows = torch.empty((K,), dtype=torch.int32, device=dev)
        c0 = torch.empty((K,), dtype=torch.int32, device=dev)
        lvl = torch.empty((K,), dtype=torch.int32, device=dev)
        L.call("nndet_conv_out_sparse_forward", ctypes.byref(desc), ctypes.byref(meta.items), ctypes.byref(lv), meta.batch, A, G, row0,
               L.ptr(pos.contiguous()), K, L.ptr(t2d), L.ptr(w32), L.ptr(b32), L.ptr(out), L.ptr(raw), L.ptr(rows), L.ptr(c0), L.ptr(lvl),
               L.stream())
        ctx.desc, ctx.meta, ctx.G, ctx.n_scales, ctx.has_bias = desc, meta, G, len(sc), bias is not None
        ctx.bias_param = bias                        # (the Parameter itself: its gradient region in a static pool is looked up by identity)
        ctx.save_for_backward(t2d, weight, w32, raw, rows, c0, lvl, *sc)
        return out

    @staticmethod
    def backward(ctx, g):
        desc, meta, G = ctx.desc, ctx.meta, ctx.G
        t2d, weight, w32, raw, rows, c0, lvl = ctx.saved_tensors[:7]
        sc = ctx.saved_tensors[7:]
        dev = t2d.device
        g = g.detach().float().contiguous()
        dsc_out = ()
        if ctx.n_scales:
            # out = scale_l * raw  =>  vals = g * scale_l, d(scale_l) = sum over the level's anchors of g . raw: one small launch
            # (as torch ops: 13 launches on the critical chain between the loss and the head trunks' backward pass)
            vals = torch.empty_like(g)
            dsc = torch.empty((ctx.n_scales,), dtype=torch.float32, device=dev)
            lv = L.NndetHeadLevels()
            lv.nlev = ctx.n_scales
            for l in range(ctx.n_scales):
                lv.scale[l], lv.dscale[l] = sc[l].data_ptr(), dsc.data_ptr() + 4 * l
            L.call("nndet_conv_out_sparse_scale_backward", ctypes.byref(lv), L.ptr(g), L.ptr(raw), L.ptr(lvl), int(lvl.numel()), G, L.ptr(vals),
                   L.stream())
            dsc_out = tuple(dsc[l].reshape(()) for l in range(ctx.n_scales))
        else:
            vals = g
        nw, cout = weight.numel(), desc.cout
        g_w, g_b = L.grad_pool.take_for([(weight, nw), (ctx.bias_param if ctx.has_bias else None, cout if ctx.has_bias else 0)], dev)
        dw = g_w.view(weight.shape)
        dbias = g_b if ctx.has_bias else None
        dx32 = torch.empty((meta.rows, desc.cin_p), dtype=torch.float32, device=dev)     # scratch: only the touched rows are used
        dx = torch.zeros_like(t2d)
        L.call("nndet_conv_out_sparse_backward", ctypes.byref(desc), ctypes.byref(meta.items), L.ptr(rows), L.ptr(c0), L.ptr(vals),
               int(rows.numel()), G, L.ptr(t2d), L.ptr(w32), L.ptr(dx32), L.ptr(dx), L.ptr(dw), L.ptr(dbias), L.stream())
        return (dx, dw.to(weight.dtype), dbias, None, None) + dsc_out


class _DetLossFn(torch.autograd.Function):
    """(reg, cls) losses of the sampled anchors and their gradients in one launch each way (csrc/boxes.hip: k_detloss,
    k_detloss_scatter; include/nndet_amd.h: nndet_detloss_f32). Arithmetic = _compute_loss_sync_free below (decode_single,
    giou_diag, binary_cross_entropy_with_logits on the one-hot labels, masked sums, the reference's divisors)."""

    @staticmethod
    def forward(ctx, box_logits, box_deltas, pos, neg, counts, labels, matched_gt, anchors, cfg):
        dev = box_logits.device
        lg, dl = box_logits.detach().float().contiguous(), box_deltas.detach().float().contiguous()
        C = lg.shape[1]
        P, Q = pos.shape[0], neg.shape[0]
        losses = torch.empty((2,), dtype=torch.float32, device=dev)
        g_d = torch.empty((P, 6), dtype=torch.float32, device=dev)
        g_l = torch.empty((P + Q, C), dtype=torch.float32, device=dev)
        lab, an = labels.detach().float().contiguous(), anchors.detach().float().contiguous()
        ctx.compact = bool(cfg.get("compact", False))        # box_deltas = the [P, 6] rows of the sampled positives (_RegSparseFn)
        if hasattr(matched_gt, "materialize"):               # core.retina.MatchedBoxes: GT boxes + ATSS matches, no [B, M, 6] gather
            mb = matched_gt
            nb = len(mb)
            base = (ctypes.c_int32 * nb)(*[min(o, max(mb.gt_all.shape[0] - 1, 0)) for o in mb.offsets[:nb]])
            gta = mb.gt_all.detach().float().contiguous()
            if gta.shape[0] == 0:                            # no object in the whole batch: nothing is positive, nothing is read
                gta = torch.zeros((1, 6), dtype=torch.float32, device=dev)
            L.call("nndet_detloss_matched_f32", L.ptr(lg), L.ptr(dl), int(ctx.compact), L.ptr(pos), P, L.ptr(neg), Q, L.ptr(counts), L.ptr(lab),
                   L.ptr(gta), L.ptr(mb.matches.contiguous()), base, nb, L.ptr(an), an.shape[0], C, float(cfg["eps"]), float(cfg["clip"]),
                   float(cfg["reg_w"]), int(cfg["reg_mean"]), float(cfg["cls_w"]), int(cfg["cls_mean"]), L.ptr(losses), L.ptr(g_d), L.ptr(g_l),
                   L.stream())
        else:
            gt = matched_gt.detach().float().contiguous()
            L.call("nndet_detloss_compact_f32" if ctx.compact else "nndet_detloss_f32", L.ptr(lg), L.ptr(dl), L.ptr(pos), P, L.ptr(neg), Q,
                   L.ptr(counts), L.ptr(lab), L.ptr(gt), L.ptr(an),
                   an.shape[0], C, float(cfg["eps"]), float(cfg["clip"]), float(cfg["reg_w"]), int(cfg["reg_mean"]), float(cfg["cls_w"]),
                   int(cfg["cls_mean"]), L.ptr(losses), L.ptr(g_d), L.ptr(g_l), L.stream())
        ctx.save_for_backward(pos, neg, g_d, g_l)
        ctx.shapes = (tuple(box_logits.shape), box_logits.dtype, tuple(box_deltas.shape), box_deltas.dtype)
        # two 0-dim outputs (not one [2] tensor the caller indexes: autograd would then build the incoming gradient with two zero
        # fills, two copies and an add -- five launches on the critical chain between the loss and the backward pass)
        return losses[0], losses[1]

    @staticmethod
    def backward(ctx, g_reg, g_cls):
        pos, neg, g_d, g_l = ctx.saved_tensors
        ls, lt, ds, dt = ctx.shapes
        dev = g_d.device

        def scalar(g):                                       # the upstream gradient of a loss as a device fp32 scalar (None: that loss is not part of the sum)
            if g is None:
                return None
            g = g.detach()
            return g if (g.dtype == torch.float32 and g.is_contiguous()) else g.float().contiguous()

        u_reg, u_cls = scalar(g_reg), scalar(g_cls)
        P, Q, C = pos.shape[0], neg.shape[0], int(g_l.shape[1])
        sparse = SPARSE_OUT and lt == torch.float32 and (ctx.compact or dt == torch.float32)
        # ONE launch: dense scatter + (for the sparse consumers) the scaled compact rows and pos ++ neg. Before round 6 the node formed the
        # scalars' stack, the two products and the concatenation with six torch launches on the chain between the loss and the backward pass.
        d_logits = torch.zeros(ls, dtype=torch.float32, device=dev)
        d_deltas = None if ctx.compact else torch.zeros(ds, dtype=torch.float32, device=dev)
        val_d = torch.empty((P, 6), dtype=torch.float32, device=dev)
        val_l = torch.empty((P + Q, C), dtype=torch.float32, device=dev) if sparse else None
        idx = torch.empty((P + Q,), dtype=torch.int64, device=dev) if sparse else None
        L.call("nndet_detloss_scatter2_f32", L.ptr(pos), P, L.ptr(neg), Q, C, L.ptr(g_d), L.ptr(g_l), L.ptr(u_reg), L.ptr(u_cls),
               L.ptr(d_deltas), L.ptr(d_logits), L.ptr(val_d), L.ptr(val_l), L.ptr(idx), L.stream())
        if ctx.compact:                                      # compact deltas: their gradient is compact too (no dense scatter)
            d_logits = d_logits.to(lt)
            if sparse:
                L.grad_hints.put(d_logits, {"idx": idx, "val": val_l, "G": C})
            return d_logits, val_d.to(dt), None, None, None, None, None, None, None
        d_logits, d_deltas = d_logits.to(lt), d_deltas.to(dt)
        if sparse:
            # both gradients are zero except at the <= P + Q sampled rows: tell the consumers (arch/pyramid.py: the gather / output
            # convolution backward then touch those rows only). The dense tensors above stay complete, valid gradients.
            L.grad_hints.put(d_deltas, {"idx": pos, "val": val_d, "G": 6})
            L.grad_hints.put(d_logits, {"idx": idx, "val": val_l, "G": C})
        return d_logits, d_deltas, None, None, None, None, None, None, None


class BCECLassifier(nn.Module):
    def __init__(self, conv, in_channels: int, internal_channels: int, num_classes: int, anchors_per_pos: int,
                 num_levels: int, num_convs: int = 3, add_norm: bool = True, prior_prob: Optional[float] = None,
                 weight: Optional[Tensor] = None, reduction: str = "mean", smoothing: float = 0.0,
                 loss_weight: float = 1., **kwargs):
        super().__init__()
        if smoothing != 0.0 or weight is not None:
            raise NotImplementedError("label smoothing / class weights are not used by RetinaUNetV001")
        self.dim, self.num_levels, self.num_convs = conv.dim, num_levels, num_convs
        self.num_classes, self.anchors_per_pos = num_classes, anchors_per_pos
        self.in_channels, self.internal_channels = in_channels, internal_channels
        self.prior_prob, self.reduction, self.loss_weight = prior_prob, reduction, loss_weight
        self.conv_internal = _trunk(conv, in_channels, internal_channels, num_convs, add_norm=add_norm, **kwargs)
        self.conv_out = conv(internal_channels, num_classes * anchors_per_pos, kernel_size=3, stride=1, padding=1,
                             add_norm=False, add_act=False, bias=True)
        self.init_weights()

    def forward(self, x: Tensor, level: int, **kwargs) -> Tensor:
        return _flatten_head(self.conv_out(self.conv_internal(x)), self.num_classes)

    def forward_raw(self, x: Tensor) -> Tensor:
        """conv output of one level before the flatten; DetectionHeadHNMNative gathers all levels in one pass (_HeadGatherFn)"""
        return self.conv_out(self.conv_internal(x))

    def compute_loss(self, pred_logits: Tensor, targets: Tensor, **kwargs) -> Tensor:
        """BCEWithLogitsLossOneHot (nndet/losses/classification.py:137-181): one-hot without the background column."""
        onehot = F.one_hot(targets.long(), self.num_classes + 1)[:, 1:].float()
        return self.loss_weight * F.binary_cross_entropy_with_logits(pred_logits, onehot, reduction=self.reduction)

    def box_logits_to_probs(self, box_logits: Tensor) -> Tensor:
        return torch.sigmoid(box_logits)

    def init_weights(self) -> None:
        """classifier.py:210-228"""
        if self.prior_prob is not None:
            for layer in self.modules():
                if isinstance(layer, CONV_TYPES):
                    nn.init.normal_(layer.weight, mean=0, std=0.01)
                    if layer.bias is not None:
                        nn.init.constant_(layer.bias, 0)
            bias_value = -math.log((1 - self.prior_prob) / self.prior_prob)
            for layer in self.conv_out.modules():
                if isinstance(layer, CONV_TYPES):
                    nn.init.constant_(layer.bias, bias_value)


class GIoURegressor(nn.Module):
    def __init__(self, conv, in_channels: int, internal_channels: int, anchors_per_pos: int, num_levels: int,
                 num_convs: int = 3, add_norm: bool = True, reduction: Optional[str] = "sum", loss_weight: float = 1.,
                 learn_scale: bool = False, **kwargs):
        super().__init__()
        self.dim, self.num_levels, self.num_convs = conv.dim, num_levels, num_convs
        self.learn_scale, self.anchors_per_pos = learn_scale, anchors_per_pos
        self.in_channels, self.internal_channels = in_channels, internal_channels
        self.reduction, self.loss_weight, self.eps = reduction, loss_weight, 1e-7
        self.conv_internal = _trunk(conv, in_channels, internal_channels, num_convs, add_norm=add_norm, **kwargs)
        self.conv_out = conv(internal_channels, anchors_per_pos * self.dim * 2, kernel_size=3, stride=1, padding=1,
                             add_norm=False, add_act=False, bias=True)
        if self.learn_scale:
            self.scales = nn.ModuleList([Scale() for _ in range(num_levels)])
        self.init_weights()

    def forward(self, x: Tensor, level: int, **kwargs) -> Tensor:
        bb = _flatten_head(self.conv_out(self.conv_internal(x)), self.dim * 2)
        if self.learn_scale:
            bb = self.scales[level](bb)          # a scalar multiply commutes with the permute/view of regressor.py:165-172
        return bb

    def forward_raw(self, x: Tensor) -> Tensor:
        """conv output of one level before Scale and flatten (both applied by _HeadGatherFn)"""
        return self.conv_out(self.conv_internal(x))

    def compute_loss(self, pred_boxes: Tensor, target_boxes: Tensor, **kwargs) -> Tensor:
        """GIoULoss (nndet/losses/regression.py:118-162): -sum(diag(GIoU(pred, target, eps=1e-7)))."""
        g = giou_diag(pred_boxes, target_boxes, eps=self.eps)
        red = g.sum() if self.reduction == "sum" else (g.mean() if self.reduction == "mean" else g)
        return self.loss_weight * -1 * red

    def init_weights(self) -> None:
        """regressor.py:194-201"""
        for layer in self.modules():
            if isinstance(layer, CONV_TYPES):
                nn.init.normal_(layer.weight, mean=0, std=0.01)
                if layer.bias is not None:
                    nn.init.constant_(layer.bias, 0)


class DetectionHeadHNMNative(nn.Module):
    """classifier + regressor + hard-negative sampling; loss on decoded boxes (comb.py:351-405)."""

    accepts_matched_boxes = True      # compute_loss takes core.retina.MatchedBoxes in place of the list of [M, 6] tensors

    def __init__(self, classifier, regressor, coder, sampler, log_num_anchors=None):
        super().__init__()
        self.classifier, self.regressor, self.coder, self.fg_bg_sampler = classifier, regressor, coder, sampler

    # The (classifier, regressor) x level branches are independent until the loss. Run sequentially, the small pyramid levels
    # leave most of the 256 CUs idle (40 workgroups per conv at 10x10x6) and the 600-workgroup P2 launches have a half-empty
    # second round; on side streams they fill each other's gaps. Set NNDET_HEAD_STREAMS=0 for the sequential order.
    multi_stream = os.environ.get("NNDET_HEAD_STREAMS", "1") != "0"
    _streams: Dict[int, list] = {}

    def _side_streams(self, device, n: int):
        pool = DetectionHeadHNMNative._streams.setdefault(device.index or 0, [])
        while len(pool) < n:
            pool.append(L.new_stream("head", device, int(os.environ.get("NNDET_PRIO_HEAD", "0"))))
        return pool[:n]

    gather_levels = os.environ.get("NNDET_HEAD_GATHER", "1") != "0"      # one flatten + Scale + cat launch per branch (csrc/headio.hip)
    # all pyramid levels through a trunk layer in ONE launch (arch/pyramid.py: the levels are the items of a ragged batch);
    # NNDET_HEAD_ITEMS=0 selects the per-level launches on side streams below
    items_levels = os.environ.get("NNDET_HEAD_ITEMS", "1") != "0"

    def _forward_items(self, fmaps: List[Tensor]) -> Dict[str, Tensor]:
        """comb.py:85-109 with the level loop folded into the kernels: 3 conv + 2 norm launches per branch for ALL levels."""
        from . import pyramid as P
        x2d, meta = P.cat_levels(fmaps)
        sdim = fmaps[0].ndim - 2
        nc, a = self.classifier.num_classes, self.classifier.anchors_per_pos
        outs = {}
        # the regressor trunk is issued first, on a side stream forked here; the classifier trunk follows on the main stream, so the
        # two (independent until the loss) fill each other's last round of workgroups
        branches = (("reg", self.regressor, self.regressor.anchors_per_pos * sdim * 2,
                     [sc.scale for sc in self.regressor.scales[:len(fmaps)]] if self.regressor.learn_scale else []),
                    ("cls", self.classifier, nc * a, []))
        two_streams = self.multi_stream
        main = torch.cuda.current_stream(x2d.device)
        side = self._side_streams(x2d.device, 1)[0] if two_streams else None
        if two_streams and torch.is_grad_enabled() and hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)   # intended: one trunk per stream
        # the FIRST trunk layer of both branches reads x2d: one convolution 128 -> 2 x 128 + one GroupNorm on the main stream (round 5,
        # arch/pyramid.py: _FusedItemsBlockFn); the branches fork behind it. NNDET_HEAD_FUSE_CIN=0: two launches on two streams.
        first = {}
        c_in_r, c_in_c = self.regressor.conv_internal[0], self.classifier.conv_internal[0]
        if FUSE_CIN and P.fusable_pair(c_in_c, c_in_r, x2d):
            first["cls"], first["reg"] = P.fused_items_blocks(c_in_c, c_in_r, x2d, meta)
        for name, head, cout, scales in branches:
            on_side = two_streams and name == "reg"
            if on_side:
                side.wait_stream(main)
                x2d.record_stream(side)
                if name in first:
                    first[name].record_stream(side)
            defer = (name == "reg" and SPARSE_REG and SPARSE_OUT and getattr(self, "_defer_reg_out", False) and torch.is_grad_enabled())
            with torch.cuda.stream(side if on_side else main):
                t = x2d
                for bi, blk in enumerate(head.conv_internal):
                    t = first[name] if (bi == 0 and name in first) else P.items_block(blk, t, meta)
                if defer:                                    # training step: conv_out + Scale only at the sampled positives, later
                    o = DeferredDeltas(t, meta, head, scales, cout, sdim)
                else:
                    t = P.items_block(head.conv_out, t, meta)
                    o = P.head_gather_items(t, meta, cout, scales)
            if on_side:
                (o.t2d if defer else o).record_stream(main)
            outs[name] = o
        if two_streams:
            main.wait_stream(side)
        deltas = outs["reg"] if isinstance(outs["reg"], DeferredDeltas) else outs["reg"].view(-1, sdim * 2)
        return {"box_deltas": deltas, "box_logits": outs["cls"].view(-1, nc)}

    def _items_ok(self, fmaps: List[Tensor]) -> bool:
        if not (self.items_levels and self.gather_levels and fmaps[0].is_cuda and len(fmaps) <= L.HEAD_MAX_LEVELS
                and type(self.classifier) is BCECLassifier and type(self.regressor) is GIoURegressor):
            return False
        from . import pyramid as P
        from .conv import DEFER_NORM
        blocks = list(self.classifier.conv_internal) + [self.classifier.conv_out] + list(self.regressor.conv_internal) + [self.regressor.conv_out]
        return (not DEFER_NORM) and P.supports(fmaps, blocks)

    def forward(self, fmaps: List[Tensor]) -> Dict[str, Tensor]:
        if self._items_ok(fmaps):
            return self._forward_items(fmaps)
        logits, offsets = [None] * len(fmaps), [None] * len(fmaps)
        fused = (self.gather_levels and fmaps[0].is_cuda and len(fmaps) <= L.HEAD_MAX_LEVELS
                 and type(self.classifier) is BCECLassifier and type(self.regressor) is GIoURegressor)
        run_cls = (lambda p, level: self.classifier.forward_raw(p)) if fused else (lambda p, level: self.classifier(p, level=level))
        run_reg = (lambda p, level: self.regressor.forward_raw(p)) if fused else (lambda p, level: self.regressor(p, level=level))
        if self.multi_stream and fmaps[0].is_cuda and len(fmaps) > 1:
            from .conv import BaseConvNormAct, prepack
            main = torch.cuda.current_stream(fmaps[0].device)
            grad = torch.is_grad_enabled()
            if grad and hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
                torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)   # intended: shared weights, many streams
            for head in (self.classifier, self.regressor):          # shared weights: pack once, before the fork
                for m in head.modules():
                    if isinstance(m, BaseConvNormAct):
                        prepack(m, fmaps[0], modes=(0, 1) if grad else (0,))
            streams = self._side_streams(fmaps[0].device, 2 * len(fmaps))
            for level, p in enumerate(fmaps):
                for hi, (run, outs) in enumerate(((run_cls, logits), (run_reg, offsets))):
                    s = streams[2 * level + hi]
                    s.wait_stream(main)
                    p.record_stream(s)
                    with torch.cuda.stream(s):
                        o = run(p, level)
                    o.record_stream(main)
                    outs[level] = o
            for s in streams:
                main.wait_stream(s)
        else:
            for level, p in enumerate(fmaps):
                logits[level] = run_cls(p, level)
                offsets[level] = run_reg(p, level)
        sdim = fmaps[0].ndim - 2
        if fused:
            nc, a = self.classifier.num_classes, self.classifier.anchors_per_pos
            scales = [sc.scale for sc in self.regressor.scales[:len(fmaps)]] if self.regressor.learn_scale else []
            box_logits = _HeadGatherFn.apply(nc * a, 0, *logits).view(-1, nc)
            box_deltas = _HeadGatherFn.apply(self.regressor.anchors_per_pos * sdim * 2, len(scales), *scales, *offsets).view(-1, sdim * 2)
            return {"box_deltas": box_deltas, "box_logits": box_logits}
        return {"box_deltas": torch.cat(offsets, dim=1).reshape(-1, sdim * 2),
                "box_logits": torch.cat(logits, dim=1).flatten(0, -2)}

    @torch.no_grad()
    def select_indices(self, target_labels: List[Tensor], boxes_scores: Tensor) -> Tuple[Tensor, Tensor]:
        """comb.py:247-276 on the device (csrc/sampler.hip): fg probability, counts, pool top-k and the random picks in one
        fused op; the only host read is the pair of counts."""
        labels = target_labels[0] if len(target_labels) == 1 else torch.cat(target_labels, dim=0)
        if hasattr(self.fg_bg_sampler, "sample_indices"):
            return self.fg_bg_sampler.sample_indices(labels, boxes_scores, len(target_labels))
        probs = torch.sigmoid(boxes_scores.detach().float()).max(dim=1)[0]       # a sampler class from elsewhere: mask contract
        pos, neg = self.fg_bg_sampler(target_labels, probs)
        return torch.where(torch.cat(pos, dim=0))[0], torch.where(torch.cat(neg, dim=0))[0]

    # The sampler's counts are the only data-dependent sizes of a training step. Reading them (select_indices) blocks the host
    # until the whole forward pass has drained, and everything after it -- loss, backward of the heads, i.e. hundreds of small
    # launches -- is then issued into an empty queue. The sync-free path keeps the fixed-capacity, -1 padded index lists of the
    # device sampler and masks the padding out of the two reductions instead (same sums, same divisors; summation order aside).
    # NNDET_SYNCFREE_LOSS=0, a patched / overridden select_indices, a foreign sampler or reduction=None select the compact path.
    sync_free = os.environ.get("NNDET_SYNCFREE_LOSS", "1") != "0"
    fused_loss = os.environ.get("NNDET_FUSED_DETLOSS", "1") != "0"        # (sync-free path only) the loss tail as one kernel
    _unit_box: Dict[torch.device, Tensor] = {}

    def _use_sync_free(self) -> bool:
        return (self.sync_free and "select_indices" not in self.__dict__
                and type(self).select_indices is DetectionHeadHNMNative.select_indices
                and hasattr(self.fg_bg_sampler, "sample_device")
                and type(self.classifier) is BCECLassifier and self.classifier.reduction in ("mean", "sum")
                and type(self.regressor) is GIoURegressor and self.regressor.reduction in ("mean", "sum"))

    def compute_loss(self, prediction: Dict[str, Tensor], target_labels: List[Tensor], matched_gt_boxes: List[Tensor],
                     anchors: List[Tensor]):
        """-> (losses dict, sampled positive indices, sampled negative indices), comb.py:351-405.

        Contract of the DEFAULT (sync-free) route on the GPU -- it differs from the reference in two documented ways so that the
        host never reads the sampler's counts between the forward and the backward pass:
          * the dict ALWAYS has the key "reg"; with no positive anchor in the batch it is an exact 0 with zero gradients, where the
            reference leaves the key out (comb.py:397-401). `sum(losses.values())` -- all the reference's callers do with it -- is
            unchanged; code that iterates the KEYS (a logger) sees one more entry on such batches;
          * the two index tensors have the sampler's fixed capacity and are padded with -1; the reference returns compact lists.
            Nothing in nnDetection consumes them (`train_step` drops them, retina.py:123-131); a caller that indexes with them must
            drop the -1 entries first (`idx[idx >= 0]`).
        `NNDET_SYNCFREE_LOSS=0`, a patched `select_indices`, a foreign sampler class or `reduction=None` select the compact route,
        which reproduces the reference's key set and index lists exactly (at the price of one host synchronisation per step)."""
        box_logits, box_deltas = prediction["box_logits"], prediction["box_deltas"]
        if box_logits.is_cuda and self._use_sync_free():
            return self._compute_loss_sync_free(box_logits, box_deltas, target_labels, matched_gt_boxes, anchors)
        if hasattr(matched_gt_boxes, "materialize"):                 # core.retina.MatchedBoxes: the reference's list for this route
            matched_gt_boxes = matched_gt_boxes.materialize()
        if isinstance(box_deltas, DeferredDeltas):
            box_deltas = box_deltas.materialize()
        losses = {}
        sampled_pos_inds, sampled_neg_inds = self.select_indices(target_labels, box_logits)
        sampled_inds = torch.cat([sampled_pos_inds, sampled_neg_inds], dim=0)
        labels = torch.cat(target_labels, dim=0)
        n_img, m = len(anchors), anchors[0].shape[0]
        same = all(a is anchors[0] for a in anchors)
        if same:                                                   # no [B*M, 6] concatenation of identical anchors
            anchors_pos = anchors[0][sampled_pos_inds % m]
        else:
            anchors_pos = torch.cat(anchors, dim=0)[sampled_pos_inds]
        pred_boxes_sampled = self.coder.decode_single(box_deltas[sampled_pos_inds], anchors_pos)
        if isinstance(matched_gt_boxes, (list, tuple)):
            matched_gt_boxes = torch.cat(matched_gt_boxes, dim=0)
        target_boxes_sampled = matched_gt_boxes[sampled_pos_inds]
        if sampled_pos_inds.numel() > 0:
            losses["reg"] = self.regressor.compute_loss(pred_boxes_sampled, target_boxes_sampled) / max(1, sampled_pos_inds.numel())
        elif torch.is_grad_enabled():
            # comb.py:397-401: no "reg" key -> the 12 regressor tensors get no gradient on this rank. Tell a gradient reducer now, so
            # that their bucket is not held back until the end of the backward pass (nndetection_amd.ddp.GradAllReducer.mark_no_grad)
            L.notify_no_grad(list(self.regressor.parameters()))
        losses["cls"] = self.classifier.compute_loss(box_logits[sampled_inds], labels[sampled_inds])
        return losses, sampled_pos_inds, sampled_neg_inds

    def _compute_loss_sync_free(self, box_logits: Tensor, box_deltas: Tensor, target_labels: List[Tensor], matched_gt_boxes,
                                anchors: List[Tensor]):
        """comb.py:351-405 without a host read. Returns the PADDED index lists (-1 = unused slot) in place of the compact ones.
        With no positive anchor the reference leaves "reg" out of the dict; here it is an exact 0 (the total is the same)."""
        labels = _cat_rows(target_labels)
        pos, neg, counts = self.fg_bg_sampler.sample_device(labels, box_logits, len(target_labels))
        from ..core.boxes.coder import BoxCoderND
        deferred = box_deltas if isinstance(box_deltas, DeferredDeltas) else None
        fusable = self.fused_loss and type(self.coder) is BoxCoderND and box_logits.dim() == 2 and pos.numel() > 0
        if deferred is not None:
            if fusable and deferred.sdim == 3:
                conv = deferred.regressor.conv_out.conv
                box_deltas = _RegSparseFn.apply(deferred.t2d, conv.weight, conv.bias, pos, deferred, *deferred.scales)
            else:
                box_deltas, deferred = deferred.materialize(), None
        if (fusable and box_deltas.dim() == 2 and box_deltas.shape[1] == 6):
            # one launch for both losses and their gradients instead of ~250 element-wise ones (csrc/boxes.hip: k_detloss)
            same = all(a is anchors[0] for a in anchors)
            an = anchors[0] if same else torch.cat(anchors, dim=0)
            indirect = hasattr(matched_gt_boxes, "materialize") and same
            gt = matched_gt_boxes if indirect else _cat_rows(matched_gt_boxes.materialize() if hasattr(matched_gt_boxes, "materialize")
                                                              else matched_gt_boxes)
            cfg = {"eps": self.regressor.eps, "clip": getattr(self.coder, "bbox_xform_clip", math.log(1000. / 16)),
                   "reg_w": self.regressor.loss_weight, "reg_mean": self.regressor.reduction == "mean",
                   "cls_w": self.classifier.loss_weight, "cls_mean": self.classifier.reduction == "mean",
                   "compact": deferred is not None}
            reg_l, cls_l = _DetLossFn.apply(box_logits, box_deltas, pos, neg, counts, labels, gt, an, cfg)
            return {"reg": reg_l, "cls": cls_l}, pos, neg
        if hasattr(matched_gt_boxes, "materialize"):
            matched_gt_boxes = matched_gt_boxes.materialize()
        n_pos, n_neg = counts[0], counts[1]
        pos_ok, neg_ok = pos >= 0, neg >= 0
        pos_c, neg_c = pos.clamp(min=0), neg.clamp(min=0)
        m = anchors[0].shape[0]
        if all(a is anchors[0] for a in anchors):
            anchors_pos = anchors[0][pos_c % m]
        else:
            anchors_pos = torch.cat(anchors, dim=0)[pos_c]
        if isinstance(matched_gt_boxes, (list, tuple)):
            matched_gt_boxes = matched_gt_boxes[0] if len(matched_gt_boxes) == 1 else torch.cat(matched_gt_boxes, dim=0)
        pred = self.coder.decode_single(box_deltas[pos_c], anchors_pos)
        tgt = matched_gt_boxes[pos_c]
        # unused slots point at anchor 0, whose target may be the all-zero box of a background anchor (GIoU 0/0): give them a
        # harmless pair; torch.where also keeps their (zero) gradient away from row 0 of box_deltas
        unit = self._unit_box.get(pred.device)
        if unit is None:                                            # built once per device (a host -> device copy blocks the host)
            unit = self._unit_box[pred.device] = torch.tensor([0., 0., 1., 1., 0., 1.], device=pred.device)
        pred = torch.where(pos_ok[:, None], pred.float(), unit)
        tgt = torch.where(pos_ok[:, None], tgt.float(), unit)
        g = giou_diag(pred, tgt, eps=self.regressor.eps)
        g_sum = torch.where(pos_ok, g, torch.zeros_like(g)).sum()
        n_pos_f = n_pos.clamp(min=1).to(g_sum.dtype)
        red = g_sum if self.regressor.reduction == "sum" else g_sum / n_pos_f
        losses = {"reg": self.regressor.loss_weight * -1 * red / n_pos_f}
        idx = torch.cat([pos_c, neg_c], dim=0)
        ok = torch.cat([pos_ok, neg_ok], dim=0)
        lab = torch.where(ok, labels[idx], torch.zeros((), dtype=labels.dtype, device=labels.device))
        nc = self.classifier.num_classes
        onehot = F.one_hot(lab.long(), nc + 1)[:, 1:].float()
        bce = F.binary_cross_entropy_with_logits(box_logits[idx], onehot, reduction="none")
        b_sum = torch.where(ok[:, None], bce, torch.zeros_like(bce)).sum()
        if self.classifier.reduction == "mean":
            b_sum = b_sum / ((n_pos + n_neg).clamp(min=1) * nc).to(b_sum.dtype)
        losses["cls"] = self.classifier.loss_weight * b_sum
        return losses, pos, neg

    def postprocess_for_inference(self, prediction: Dict[str, Tensor], anchors: List[Tensor]) -> Dict[str, Tensor]:
        return {"pred_boxes": self.coder.decode(prediction["box_deltas"], anchors),
                "pred_probs": self.classifier.box_logits_to_probs(prediction["box_logits"])}
