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

__all__ = ["PyramidMeta", "pyramid_meta", "cat_levels", "items_block", "head_gather_items", "supports", "fused_items_blocks", "fusable_pair"]


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


class _FusedPair:
    """Stand-in for ONE conv block made of two blocks that read the same input (the first layers of the classifier and the regressor
    trunk): the channel counts / kernel / norm parameters of the concatenation, and its own packed-weight cache."""

    def __init__(self, a: BaseConvNormAct, b: BaseConvNormAct):
        self.in_channels, self.out_channels = a.in_channels, a.out_channels + b.out_channels
        self.split = a.out_channels
        self.k, self.s, self.p, self.transposed = a.k, a.s, a.p, False
        cpg = a.out_channels // a.norm_groups
        self.norm_groups, self.norm_eps, self.relu = self.out_channels // cpg, a.norm_eps, a.relu
        self._pack_cache = {}


def fusable_pair(a, b, x2d: torch.Tensor) -> bool:
    """Can blocks `a` and `b` (same input) run as one convolution + one GroupNorm? Same 3x3x3 / stride-1 convolution shape without bias,
    the same kind of norm with whole groups inside each half, output channel counts that are multiples of 32 (the halves are written
    as two dense tensors, csrc/norm.hip: split I/O)."""
    ok = (isinstance(a, BaseConvNormAct) and isinstance(b, BaseConvNormAct) and x2d.is_cuda
          and a.in_channels == b.in_channels and a.out_channels == b.out_channels and a.out_channels % 32 == 0
          and a.k == b.k == (3, 3, 3) and a.s == b.s == (1, 1, 1) and a.p == b.p == (1, 1, 1) and not a.transposed and not b.transposed
          and a.conv.bias is None and b.conv.bias is None and a.norm_groups > 0 and a.norm_groups == b.norm_groups
          and a.out_channels % a.norm_groups == 0 and a.norm_eps == b.norm_eps and a.relu == b.relu
          and a.conv.weight.dtype == b.conv.weight.dtype == torch.float32
          #  affine norms on both sides, and the same trainability: a frozen half would still get its weight-gradient work and
          # a gradient it must not receive
          and getattr(a.norm, "weight", None) is not None and getattr(a.norm, "bias", None) is not None
          and getattr(b.norm, "weight", None) is not None and getattr(b.norm, "bias", None) is not None
          and a.conv.weight.requires_grad == b.conv.weight.requires_grad
          and a.norm.weight.requires_grad == b.norm.weight.requires_grad == a.norm.bias.requires_grad == b.norm.bias.requires_grad
          and (a.conv.weight.requires_grad or not torch.is_grad_enabled()))
    return bool(ok)


class _FusedItemsBlockFn(torch.autograd.Function):
    """conv -> norm -> ReLU of TWO blocks that read the same ragged batch, as one 3x3x3 convolution Cin -> 2 x C and one norm over the 2 C
    channels (round 5; nndet/arch/heads/comb.py:85-109: the classifier's and the regressor's `conv_internal.c_in` see the same feature
    maps). Forward: the input is read once (the two launches re-fetched its halo tiles twice, 3.6 x the algorithmic bytes), the
    normalised halves come out as two dense tensors. Backward: one norm backward over both incoming gradients, ONE data-gradient
    launch whose accumulator sums both branches (no dX_cls + dX_reg add pass), one weight-gradient launch over the concatenated dY.
    Per element the convolution / norm arithmetic is that of the separate launches (same tiles, same order): bit-identical outputs."""

    @staticmethod
    def forward(ctx, x2d, w_a, w_b, g_a, b_a, g_b, b_b, pair, mods, meta):
        x2d = x2d.contiguous()
        dev, dt = x2d.device, x2d.dtype
        desc = _items_desc(x2d, pair, meta)
        # packed weights of the concatenation, cached until one of the two parameters changes
        from .conv import _pver, PACK_EPOCH
        ver = (_pver(w_a), _pver(w_b), PACK_EPOCH[0])
        hit = pair._pack_cache.get(("w", dt))
        if hit is None or hit[0] != ver:
            w32 = torch.cat((w_a.detach().float(), w_b.detach().float()), 0).contiguous()
            bufs = []
            for mode in (0, 1):
                n = L.load().nndet_packed_weight_elems(ctypes.byref(desc), mode)
                buf = torch.empty((n,), dtype=dt, device=dev)
                L.call("nndet_pack_weight", ctypes.byref(desc), mode, L.ptr(w32), L.ptr(buf), L.stream())
                bufs.append(buf)
            hit = pair._pack_cache[("w", dt)] = (ver, bufs)
        w0, w1 = hit[1]
        cout, cout_p, split = pair.out_channels, desc.cout_p, pair.split
        y = torch.empty((meta.rows, cout_p), dtype=dt, device=dev)
        stats = L.arena_zeros((L.STATS_REPLICAS, meta.n_items, cout_p, 2), torch.float64, dev)
        L.call("nndet_conv3d_forward_items", ctypes.byref(desc), ctypes.byref(meta.items), L.ptr(x2d), L.ptr(w0), None, L.ptr(y), L.ptr(stats),
               L.stream())
        g32 = torch.cat((g_a.detach().float(), g_b.detach().float())).contiguous()
        b32 = torch.cat((b_a.detach().float(), b_b.detach().float())).contiguous()
        mean_rstd = torch.empty((meta.n_items, cout_p, 2), dtype=torch.float32, device=dev)
        out_a = torch.empty((meta.rows, split), dtype=dt, device=dev)
        out_b = torch.empty((meta.rows, cout_p - split), dtype=dt, device=dev)
        code = L.dtype_code(y)
        L.call("nndet_norm_apply_items_split", code, L.ptr(y), L.ptr(stats), L.ptr(g32), L.ptr(b32), ctypes.byref(meta.items), cout, cout_p,
               pair.norm_groups, float(pair.norm_eps), int(pair.relu), L.ptr(out_a), L.ptr(out_b), split, L.ptr(mean_rstd), L.stream())
        ctx.desc, ctx.pair, ctx.mods, ctx.meta, ctx.code, ctx.w1 = desc, pair, mods, meta, code, w1
        ctx.save_for_backward(x2d, y, mean_rstd, g32, b32, w_a, w_b)
        ctx.set_materialize_grads(False)        # a branch that is not part of the loss arrives as None (its parameters then get NO gradient)
        return out_a, out_b

    @staticmethod
    def backward(ctx, ga, gb):
        desc, pair, meta = ctx.desc, ctx.pair, ctx.meta
        mod_a, mod_b = ctx.mods
        x2d, y, mean_rstd, g32, b32, w_a, w_b = ctx.saved_tensors
        dev, dt = x2d.device, x2d.dtype
        cout, cout_p, split = pair.out_channels, desc.cout_p, pair.split
        if ga is None and gb is None:
            return (None,) * 10
        # One branch without a gradient (the regressor on a batch without positive anchors, nndet/arch/heads/comb.py:397-401): its half
        # of the incoming gradient is zero and its parameters are handed None, as the separate block would leave them
        has_a, has_b = ga is not None, gb is not None
        ga = ga.to(dt).contiguous() if has_a else torch.zeros((meta.rows, split), dtype=dt, device=dev)
        gb = gb.to(dt).contiguous() if has_b else torch.zeros((meta.rows, cout_p - split), dtype=dt, device=dev)
        nw = w_a.numel()
        # one contiguous [2 C][Cin][27] weight gradient (the kernel's row stride spans both halves); handed to autograd as its two halves
        gw, gg, gbt = L.grad_pool.take_for([(None, 2 * nw), (None, cout), (None, cout)], dev)
        dconv = torch.empty_like(y)
        red = L.arena_zeros((L.STATS_REPLICAS * meta.n_items * cout_p * 2 + meta.n_items,), torch.float64, dev)
        L.call("nndet_norm_backward_items_split", ctx.code, L.ptr(y), L.ptr(ga), L.ptr(gb), split, L.ptr(mean_rstd), L.ptr(g32), L.ptr(b32),
               ctypes.byref(meta.items), cout, cout_p, pair.norm_groups, int(pair.relu), L.ptr(dconv), L.ptr(gg), L.ptr(gbt), L.ptr(red),
               L.stream())
        side = L.wgrad_streams.side(dev, w_a)               # weight-gradient stream, forked before the data gradient is queued
        if side is not None:
            L.wgrad_streams.side(dev, w_b)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x2d)
            L.call("nndet_conv3d_backward_data_items", ctypes.byref(desc), ctypes.byref(meta.items), L.ptr(dconv), L.ptr(ctx.w1), L.ptr(dx),
                   L.stream())
        ws_bytes = L.load().nndet_conv3d_wgrad_workspace_bytes(ctypes.byref(desc))
        if side is not None:
            x2d.record_stream(side); dconv.record_stream(side)
        raw = side.cuda_stream if side is not None else L.stream()
        ws = L.workspace(ws_bytes, dev, raw_stream=raw if side is not None else None)
        dw = gw.view(2 * w_a.shape[0], *w_a.shape[1:])
        L.call("nndet_conv3d_backward_weight_items", ctypes.byref(desc), ctypes.byref(meta.items), L.ptr(x2d), L.ptr(dconv), L.ptr(dw), None,
               L.ptr(ws), ws_bytes, raw)
        ca = mod_a.out_channels
        return (dx, dw[:ca].to(w_a.dtype) if has_a else None, dw[ca:].to(w_b.dtype) if has_b else None,
                gg[:ca] if has_a else None, gbt[:ca] if has_a else None, gg[ca:] if has_b else None, gbt[ca:] if has_b else None,
                None, None, None)


_PAIRS: Dict[tuple, _FusedPair] = {}


def fused_items_blocks(a: BaseConvNormAct, b: BaseConvNormAct, x2d: torch.Tensor, meta: PyramidMeta):
    """(a(x), b(x)) for two conv -> norm -> ReLU blocks on the same ragged batch, as one launch per kernel (see _FusedItemsBlockFn)."""
    key = (id(a), id(b))
    pair = _PAIRS.get(key)
    if pair is None or pair.owner() != (a, b):
        pair = _PAIRS[key] = _FusedPair(a, b)
        import weakref
        drop = lambda _r, k=key: _PAIRS.pop(k, None)          #  the entry dies with either block: ids are recycled
        ra, rb = weakref.ref(a, drop), weakref.ref(b, drop)
        pair.owner = lambda: (ra(), rb())
    return _FusedItemsBlockFn.apply(x2d, a.conv.weight, b.conv.weight, a.norm.weight, a.norm.bias, b.norm.weight, b.norm.bias, pair, (a, b), meta)


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
