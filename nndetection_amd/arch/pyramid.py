"""Pyramid levels as ONE ragged batch through the shared detection-head trunks.

The reference calls the same classifier / regressor convolutions once per pyramid level
(nndet/arch/heads/comb.py:85-109 -> classifier.py:160-181, regressor.py:153-173): per level 3 convs + 2 GroupNorms per branch,
forward and backward, i.e. ~350 launches per training step, most of them on levels of 75 ... 4800 positions per image that
cannot fill 256 CUs. Here the levels P2..P5 of all images are the ITEMS of one [rows, C_p] buffer (level-major, so every level is
still a contiguous NDHWC tensor) and each layer of a trunk is ONE launch over all items (include/nndet_amd.h: NndetItems,
nndet_conv3d_*_items, nndet_norm_*_items). The weight gradient of the shared parameters is summed over the levels inside the
kernel. Per element the arithmetic is that of the per-level path (same tiles, same tap / accumulation order): outputs and data
gradients are bit-identical, weight gradients differ by fp32 summation order only (tests/test_pyramid_gpu.py).
"""
import ctypes
from typing import Dict, List, Sequence, Tuple

import torch

from .. import _lib as L
from ..layout import cpad, phys, logical
from .conv import _packed, _padded_bias, BaseConvNormAct

__all__ = ["PyramidMeta", "pyramid_meta", "cat_levels", "items_block", "head_gather_items", "supports"]


class PyramidMeta:
    """Static description of a ragged batch: level shapes (N, D, H, W), the C-ABI item table, row offsets of the levels."""

    def __init__(self, level_shapes: Sequence[Tuple[int, int, int, int]]):
        self.level_shapes = [tuple(int(v) for v in s) for s in level_shapes]
        items = L.NndetItems()
        rows, idx = 0, 0
        self.level_rows: List[Tuple[int, int]] = []
        for (n, d, h, w) in self.level_shapes:
            self.level_rows.append((rows, n * d * h * w))
            for _ in range(n):
                items.dims[idx][0], items.dims[idx][1], items.dims[idx][2] = d, h, w
                items.row_off[idx] = rows
                rows += d * h * w
                idx += 1
        items.n_items = idx
        self.items, self.n_items, self.rows = items, idx, rows
        self.batch = self.level_shapes[0][0]


_META: Dict[tuple, PyramidMeta] = {}


def pyramid_meta(level_shapes) -> PyramidMeta:
    key = tuple(tuple(int(v) for v in s) for s in level_shapes)
    m = _META.get(key)
    if m is None:
        m = _META[key] = PyramidMeta(key)
    return m


def supports(fmaps: Sequence[torch.Tensor], blocks: Sequence[torch.nn.Module]) -> bool:
    """True if the feature maps can go through `blocks` as one ragged batch: same batch / channels / dtype on the GPU, at most
    NNDET_MAX_ITEMS (level, image) pairs, every block a 3x3x3 / stride 1 / padding 1 Conv3d (+ GroupNorm / InstanceNorm + ReLU)."""
    if len(fmaps) < 2 or not all(f.is_cuda and f.dim() == 5 for f in fmaps):
        return False
    f0 = fmaps[0]
    if f0.dtype not in L._DT:
        return False
    if any(f.dtype != f0.dtype or f.shape[:2] != f0.shape[:2] for f in fmaps):
        return False
    if f0.shape[0] * len(fmaps) > L.MAX_ITEMS or f0.shape[1] == 1:
        return False
    esz = f0.element_size()
    cmax = cpad(f0.shape[1])
    for b in blocks:
        if not isinstance(b, BaseConvNormAct) or b.transposed or b.k != (3, 3, 3) or b.s != (1, 1, 1) or b.p != (1, 1, 1):
            return False
        cmax = max(cmax, cpad(b.out_channels))
    return all(f.shape[2] * f.shape[3] * f.shape[4] * cmax * esz < 2 ** 31 for f in fmaps)


class _CatLevelsFn(torch.autograd.Function):
    """Per-level logical [N, C, D, H, W] tensors -> the ragged [rows, C_p] buffer (one copy launch); backward hands every level a
    VIEW of the gradient buffer (no kernel)."""

    @staticmethod
    def forward(ctx, meta: PyramidMeta, *fmaps):
        ps = [phys(f)[0] for f in fmaps]
        ctx.meta, ctx.c = meta, fmaps[0].shape[1]
        cp = ps[0].shape[4]
        # consecutive slices of one allocation (arch/decoder.py: _ragged_out): the concatenation is a view
        p0 = ps[0]
        if p0.is_cuda and all(p.is_contiguous() and p.dtype == p0.dtype and p.shape[4] == cp for p in ps):
            st, off, ok = p0.untyped_storage().data_ptr(), p0.data_ptr(), True
            for p in ps:
                ok = ok and p.untyped_storage().data_ptr() == st and p.data_ptr() == off
                off += p.numel() * p.element_size()
            if ok and off - p0.data_ptr() <= p0.untyped_storage().nbytes() - (p0.data_ptr() - st):
                total = sum(p.numel() for p in ps) // cp
                return torch.empty((0,), dtype=p0.dtype, device=p0.device).set_(p0.untyped_storage(), p0.storage_offset(), (total, cp), (cp, 1))
        return torch.cat([p.reshape(-1, cp) for p in ps], dim=0)

    @staticmethod
    def backward(ctx, g):
        meta = ctx.meta
        g = g.contiguous()
        cp = g.shape[1]
        outs = []
        for (n, d, h, w), (r0, nr) in zip(meta.level_shapes, meta.level_rows):
            outs.append(logical(g[r0:r0 + nr].view(n, d, h, w, cp), ctx.c))
        return (None,) + tuple(outs)


def cat_levels(fmaps: Sequence[torch.Tensor]) -> Tuple[torch.Tensor, PyramidMeta]:
    meta = pyramid_meta([(f.shape[0], f.shape[2], f.shape[3], f.shape[4]) for f in fmaps])
    return _CatLevelsFn.apply(meta, *fmaps), meta


def _items_desc(x2d: torch.Tensor, mod, meta: PyramidMeta) -> L.NndetConv:
    d = L.NndetConv()
    d.dtype = L.dtype_code(x2d)
    d.transposed = 0
    d.batch = meta.n_items
    d.cin, d.cout, d.cin_p, d.cout_p = mod.in_channels, mod.out_channels, x2d.shape[1], cpad(mod.out_channels)
    _, dd, hh, ww = meta.level_shapes[0]
    d.in_d, d.in_h, d.in_w = dd, hh, ww
    d.out_d, d.out_h, d.out_w = dd, hh, ww
    d.k = (ctypes.c_int32 * 3)(3, 3, 3); d.s = (ctypes.c_int32 * 3)(1, 1, 1); d.p = (ctypes.c_int32 * 3)(1, 1, 1)
    return d


class _ItemsConvFn(torch.autograd.Function):
    """3x3x3 conv (+bias) of every item of the ragged batch in one launch; side output = the per-(item, channel) sum / sum of
    squares for the norm (conv epilogue). Backward: one data-gradient launch, one weight-gradient launch (+ its reduction) whose
    result is already summed over the levels that share `weight`."""

    @staticmethod
    def forward(ctx, x2d, weight, bias, mod, meta, want_stats):
        if x2d.dim() != 2 or x2d.shape[0] != meta.rows or x2d.shape[1] != cpad(mod.in_channels):
            raise L.NndetError(f"ragged batch: expected [{meta.rows}, {cpad(mod.in_channels)}], got {tuple(x2d.shape)}")
        x2d = x2d.contiguous()
        desc = _items_desc(x2d, mod, meta)
        dev, dt = x2d.device, x2d.dtype
        w0 = _packed(mod, 0, weight, desc, dt)
        y = torch.empty((meta.rows, desc.cout_p), dtype=dt, device=dev)
        stats = L.arena_zeros((L.STATS_REPLICAS, meta.n_items, desc.cout_p, 2), torch.float64, dev) if want_stats else None
        b_p = _padded_bias(mod, bias, desc.cout_p)
        L.call("nndet_conv3d_forward_items", ctypes.byref(desc), ctypes.byref(meta.items), L.ptr(x2d), L.ptr(w0), L.ptr(b_p),
               L.ptr(y), L.ptr(stats), L.stream())
        ctx.desc, ctx.mod, ctx.meta, ctx.has_bias = desc, mod, meta, bias is not None
        ctx.save_for_backward(x2d, weight)
        if want_stats:
            ctx.mark_non_differentiable(stats)
            return y, stats
        return y, None

    @staticmethod
    def backward(ctx, grad_out, _grad_stats=None):
        desc, mod, meta = ctx.desc, ctx.mod, ctx.meta
        x2d, weight = ctx.saved_tensors
        dev, dt = x2d.device, x2d.dtype
        dconv = grad_out.to(dt).contiguous()
        nw, cout = weight.numel(), desc.cout
        g_w, g_b = L.grad_pool.take_for([(weight, nw), (mod.conv.bias if ctx.has_bias else None, cout if ctx.has_bias else 0)], dev)
        dw = g_w.view(weight.shape)
        dbias = g_b if ctx.has_bias else None
        hint = L.grad_hints.pop(dconv)                      # set by _HeadGatherItemsFn.backward: dconv is zero except at <= 170 rows
        if hint is not None:
            # sparse data + weight gradient of this output convolution (csrc/sparse_out.hip); dw / dbias are views of the zeroed pool
            w32 = rounded_w32(mod, weight, dt)                               # the values the forward kernel multiplied with
            dx32 = torch.empty((meta.rows, desc.cin_p), dtype=torch.float32, device=dev)     # scratch: only the touched rows are used
            dx = torch.zeros_like(x2d)
            L.call("nndet_conv_out_sparse_backward", ctypes.byref(desc), ctypes.byref(meta.items), L.ptr(hint["rows"]), L.ptr(hint["c0"]),
                   L.ptr(hint["vals"]), int(hint["rows"].numel()), int(hint["G"]), L.ptr(x2d), L.ptr(w32), L.ptr(dx32), L.ptr(dx),
                   L.ptr(dw), L.ptr(dbias), L.stream())
            return (dx if ctx.needs_input_grad[0] else None), dw.to(weight.dtype), dbias, None, None, None
        side = L.wgrad_streams.side(dev, weight)            # weight-gradient stream, forked before the data gradient is queued
        dx = None
        if ctx.needs_input_grad[0]:
            w1 = _packed(mod, 1, weight, desc, dt)
            dx = torch.empty_like(x2d)
            L.call("nndet_conv3d_backward_data_items", ctypes.byref(desc), ctypes.byref(meta.items), L.ptr(dconv), L.ptr(w1),
                   L.ptr(dx), L.stream())
        ws_bytes = L.load().nndet_conv3d_wgrad_workspace_bytes(ctypes.byref(desc))
        if side is not None:
            x2d.record_stream(side); dconv.record_stream(side)
        raw = side.cuda_stream if side is not None else L.stream()
        ws = L.workspace(ws_bytes, dev, raw_stream=raw if side is not None else None)
        L.call("nndet_conv3d_backward_weight_items", ctypes.byref(desc), ctypes.byref(meta.items), L.ptr(x2d), L.ptr(dconv),
               L.ptr(dw), L.ptr(dbias), L.ptr(ws), ws_bytes, raw)
        return dx, dw.to(weight.dtype), dbias, None, None, None


class _ItemsNormFn(torch.autograd.Function):
    """GroupNorm / InstanceNorm (+ReLU) per item of the ragged batch from the conv-epilogue statistics: finalize + apply in the
    forward pass, reduce (+ finalize by the last workgroup of an item) + apply in the backward pass, each ONE launch for all items."""

    @staticmethod
    def forward(ctx, y2d, gamma, beta, stats, mod, meta):
        dev = y2d.device
        cout, cout_p = mod.out_channels, y2d.shape[1]
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        mean_rstd = torch.empty((meta.n_items, cout_p, 2), dtype=torch.float32, device=dev)
        out = torch.empty_like(y2d)
        code = L.dtype_code(y2d)
        L.call("nndet_norm_apply_items", code, L.ptr(y2d), L.ptr(stats), L.ptr(g32), L.ptr(b32), ctypes.byref(meta.items), cout,
               cout_p, mod.norm_groups, float(mod.norm_eps), int(mod.relu), L.ptr(out), L.ptr(mean_rstd), L.stream())
        ctx.mod, ctx.meta, ctx.code, ctx.dims = mod, meta, code, (cout, cout_p)
        ctx.save_for_backward(y2d, mean_rstd, g32, b32)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        mod, meta = ctx.mod, ctx.meta
        cout, cout_p = ctx.dims
        y2d, mean_rstd, g32, b32 = ctx.saved_tensors
        dev = y2d.device
        g = grad_out.to(y2d.dtype).contiguous()
        dgamma, dbeta = L.grad_pool.take_for([(mod.norm.weight, cout), (mod.norm.bias, cout)], dev)
        dconv = torch.empty_like(y2d)
        red = L.arena_zeros((L.STATS_REPLICAS * meta.n_items * cout_p * 2 + meta.n_items,), torch.float64, dev)
        L.call("nndet_norm_backward_items", ctx.code, L.ptr(y2d), L.ptr(g), L.ptr(mean_rstd), L.ptr(g32), L.ptr(b32),
               ctypes.byref(meta.items), cout, cout_p, mod.norm_groups, int(mod.relu), L.ptr(dconv), L.ptr(dgamma), L.ptr(dbeta),
               L.ptr(red), L.stream())
        return dconv, dgamma, dbeta, None, None, None


def rounded_w32(mod: BaseConvNormAct, weight: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    """fp32 copy of `weight` rounded to the activation dtype (what the packed 16-bit weights of the dense kernels hold), cached per
    parameter version like the packed weights (arch/conv.py: _packed; dropped by the forced re-pack of a training pass)."""
    if dtype == torch.float32:
        return weight.detach().float().contiguous()
    from .conv import _pver
    key, ver = ("w32r", dtype), _pver(weight)
    hit = mod._pack_cache.get(key)
    if hit is None or hit[0] != ver:
        hit = mod._pack_cache[key] = (ver, weight.detach().to(dtype).float().contiguous())
    return hit[1]


def items_block(mod: BaseConvNormAct, x2d: torch.Tensor, meta: PyramidMeta) -> torch.Tensor:
    """conv -> norm -> ReLU block `mod` applied to the ragged batch [rows, Cin_p] -> [rows, Cout_p]."""
    has_norm = mod.norm_groups > 0
    y, stats = _ItemsConvFn.apply(x2d, mod.conv.weight, mod.conv.bias, mod, meta, has_norm)
    if not has_norm:
        return y
    return _ItemsNormFn.apply(y, mod.norm.weight, mod.norm.bias, stats, mod, meta)


class _HeadGatherItemsFn(torch.autograd.Function):
    """The ragged conv output of one head branch -> fp32 [N, sum_l positions_l, cout] (flatten + Scale + cat of heads._HeadGatherFn);
    the level pointers are offsets into the one buffer, and the backward pass writes ONE gradient buffer."""

    @staticmethod
    def forward(ctx, cout: int, n_scales: int, meta: PyramidMeta, *args):
        scales, y2d = args[:n_scales], args[n_scales]
        y2d = y2d.contiguous()
        cout_p, esz = y2d.shape[1], y2d.element_size()
        N = meta.batch
        pts = [d * h * w for (_, d, h, w) in meta.level_shapes]
        sc = [s.detach().float().contiguous() for s in scales]
        out = torch.empty((N, sum(pts), cout), dtype=torch.float32, device=y2d.device)
        lv = L.NndetHeadLevels()
        lv.nlev = len(pts)
        base = y2d.data_ptr()
        for l, (r0, _) in enumerate(meta.level_rows):
            lv.y[l], lv.points[l] = base + r0 * cout_p * esz, pts[l]
            lv.scale[l] = sc[l].data_ptr() if n_scales else None
        L.call("nndet_head_gather_f32", L.dtype_code(y2d), ctypes.byref(lv), N, cout, cout_p, L.ptr(out), L.stream())
        ctx.cout, ctx.n_scales, ctx.pts, ctx.meta = cout, n_scales, pts, meta
        ctx.save_for_backward(y2d, *sc)
        return out

    @staticmethod
    def backward(ctx, g):
        meta = ctx.meta
        y2d, sc = ctx.saved_tensors[0], ctx.saved_tensors[1:]
        cout_p, esz = y2d.shape[1], y2d.element_size()
        g = g.contiguous().float()
        hint = L.grad_hints.pop(g)                     # set by heads._DetLossFn.backward: g is zero except at <= 170 sampled anchors
        dsc = torch.zeros((max(ctx.n_scales, 1),), dtype=torch.float32, device=g.device)
        lv = L.NndetHeadLevels()
        lv.nlev = len(ctx.pts)
        if hint is not None and hint["idx"].numel() > 0 and ctx.cout % hint["G"] == 0:
            # sparse route (csrc/sparse_out.hip): entries -> rows of the ragged buffer; dy = zeros + those rows (still the complete
            # dense gradient), and a hint for the output convolution's backward pass
            dy = torch.zeros_like(y2d)
            K, G = int(hint["idx"].numel()), int(hint["G"])
            rows = torch.empty((K,), dtype=torch.int32, device=g.device)
            c0 = torch.empty((K,), dtype=torch.int32, device=g.device)
            vals = torch.empty((K, G), dtype=torch.float32, device=g.device)
            for l in range(len(ctx.pts)):
                lv.points[l] = ctx.pts[l]
                if ctx.n_scales:
                    lv.scale[l], lv.dscale[l] = sc[l].data_ptr(), dsc.data_ptr() + 4 * l
            row0 = (ctypes.c_int64 * len(ctx.pts))(*[r0 for (r0, _) in meta.level_rows])
            L.call("nndet_head_out_sparse_scatter", L.dtype_code(y2d), ctypes.byref(lv), meta.batch, ctx.cout // G, G, row0,
                   L.ptr(hint["idx"].contiguous()), L.ptr(hint["val"].float().contiguous()), K, L.ptr(y2d), cout_p, L.ptr(dy),
                   L.ptr(rows), L.ptr(c0), L.ptr(vals), L.stream())
            L.grad_hints.put(dy, {"rows": rows, "c0": c0, "vals": vals, "G": G})
            return (None, None, None) + tuple(dsc[l].reshape(()) for l in range(ctx.n_scales)) + (dy,)
        dy = torch.empty_like(y2d)
        yb, db = y2d.data_ptr(), dy.data_ptr()
        for l, (r0, _) in enumerate(meta.level_rows):
            lv.y[l], lv.dy[l], lv.points[l] = yb + r0 * cout_p * esz, db + r0 * cout_p * esz, ctx.pts[l]
            if ctx.n_scales:
                lv.scale[l], lv.dscale[l] = sc[l].data_ptr(), dsc.data_ptr() + 4 * l
        L.call("nndet_head_gather_backward", L.dtype_code(y2d), ctypes.byref(lv), meta.batch, ctx.cout, cout_p, L.ptr(g), L.stream())
        return (None, None, None) + tuple(dsc[l].reshape(()) for l in range(ctx.n_scales)) + (dy,)


def head_gather_items(y2d: torch.Tensor, meta: PyramidMeta, cout: int, scales: Sequence[torch.Tensor] = ()) -> torch.Tensor:
    return _HeadGatherItemsFn.apply(cout, len(scales), meta, *scales, y2d)
