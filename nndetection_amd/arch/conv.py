"""Conv -> norm -> activation blocks on hand-written HIP kernels, behind the reference's conv factory
contract (nndet/arch/conv.py:28-51,54-143,146-294).

`ConvInstanceRelu(dim, in_ch, out_ch, kernel_size, stride, padding, ..., add_norm, add_act, transposed)` and
`ConvGroupRelu(..., norm_channels_per_group=16)` are `nn.Sequential`s whose children are named `conv`,
`norm`, `act` exactly like the reference (state-dict keys `...conv.weight`, `...norm.weight`), `conv` IS an
`nn.Conv3d` / `nn.ConvTranspose3d` (so the heads' `init_weights` isinstance loops work) and `norm` IS an
`nn.InstanceNorm3d` / `nn.GroupNorm` (so `get_params_no_wd_on_norm` finds it). Only `forward` differs: it runs
ONE autograd Function over the whole block:

    forward : implicit-GEMM conv (MFMA, NDHWC, stats in the epilogue) -> fused norm-apply + ReLU
    backward: fused norm/ReLU backward -> data gradient (MFMA) + weight gradient (MFMA) + bias gradient

Tensors crossing module boundaries are logical [N, C, D, H, W] views (what the reference's modules exchange)
of NDHWC storage whose channel count is padded to a multiple of 32 (`layout.py`). Parameters stay fp32 in
PyTorch's layout; they are re-packed (and cast to the activation dtype) once per optimizer step.
"""
import ctypes
import os
from typing import Optional, Sequence, Union

import torch
import torch.nn as nn

from .. import _lib as L
from ..layout import cpad, phys, logical, mark_padded

__all__ = ["Generator", "ConvInstanceRelu", "ConvGroupRelu", "conv_kwargs_helper", "compute_padding_for_kernel", "deferred", "materialize"]


class Generator:
    """Factory helper (nndet/arch/conv.py:28-51): Generator(conv_cls, dim)(*args, **kw) -> conv_cls(dim, *args, **kw)."""

    def __init__(self, conv_cls, dim: int):
        self.dim = dim
        self.conv_cls = conv_cls

    def __call__(self, *args, **kwargs):
        return self.conv_cls(self.dim, *args, **kwargs)


def _triple(v):
    return tuple(int(i) for i in v) if isinstance(v, (tuple, list)) else (int(v),) * 3


def _desc(x_p: torch.Tensor, cin: int, cout: int, k, s, p, transposed: bool) -> L.NndetConv:
    N, D, H, W, cin_p = x_p.shape
    d = L.NndetConv()
    d.dtype = L.dtype_code(x_p)
    d.transposed = int(transposed)
    d.batch, d.cin, d.cout, d.cin_p, d.cout_p = N, cin, cout, cin_p, cpad(cout)
    d.in_d, d.in_h, d.in_w = D, H, W
    sp = (D, H, W)
    out = [sp[i] * s[i] if transposed else (sp[i] + 2 * p[i] - k[i]) // s[i] + 1 for i in range(3)]
    d.out_d, d.out_h, d.out_w = out
    d.k = (ctypes.c_int32 * 3)(*k); d.s = (ctypes.c_int32 * 3)(*s); d.p = (ctypes.c_int32 * 3)(*p)
    return d


# Everything cached from a parameter is keyed on (its autograd version, its address, PARAM_GENERATION[0]). Fused optimizer kernels
# (torch._fused_sgd_ / torch.optim.*(fused=True)) write parameters WITHOUT advancing `_version`; whoever installs such an optimizer
# registers `bump_param_generation` as an optimizer step post-hook (ptmodule.amd_fuse_sgd does), so that the caches are invalidated by
# the step itself and correctness does not depend on the unconditional re-pack of a training-mode forward (NNDET_REPACK_EVERY_STEP).
PARAM_GENERATION = [0]


def bump_param_generation(*_args, **_kwargs) -> None:
    PARAM_GENERATION[0] += 1


def _pver(t: torch.Tensor):
    return (t._version, t.data_ptr(), PARAM_GENERATION[0])


# Caches that do not hang off a conv block (arch/pyramid.py: the packed weights of two FUSED blocks) are additionally keyed on this
# counter: it advances whenever the per-module caches are dropped wholesale -- the forced re-pack of a training-mode forward pass
# (prepack_all(force=True): a foreign fused optimizer may have written the parameters without touching their version counters) and a
# train() <-> eval() switch of the detector (core/retina.py).
PACK_EPOCH = [0]


def _packed(mod, mode: int, weight: torch.Tensor, desc: L.NndetConv, dtype: torch.dtype) -> torch.Tensor:
    """Packed + cast weights, cached per (mode, dtype) until the parameter changes (optimizer step)."""
    key = (mode, dtype)
    ver = _pver(weight)
    hit = mod._pack_cache.get(key)
    if hit is not None and hit[0] == ver:
        return hit[1]
    n = L.load().nndet_packed_weight_elems(ctypes.byref(desc), mode)
    buf = torch.empty((n,), dtype=dtype, device=weight.device)
    w32 = weight.detach().float().contiguous()
    L.call("nndet_pack_weight", ctypes.byref(desc), mode, L.ptr(w32), L.ptr(buf), L.stream())
    mod._pack_cache[key] = (ver, buf)
    return buf


def _padded_bias(mod, bias: Optional[torch.Tensor], cout_p: int) -> Optional[torch.Tensor]:
    """fp32 bias padded to the physical channel count. The padded copy (zeros + copy = 2 tiny launches) is made once per parameter
    version on the stream that first needs it (prepack / prepack_all: before the head forks its side streams), not once per call."""
    if bias is None:
        return None
    if bias.numel() == cout_p and bias.dtype == torch.float32:
        return bias.detach()
    key, ver = ("bias", cout_p), _pver(bias)
    hit = mod._pack_cache.get(key)
    if hit is None or hit[0] != ver:
        hit = mod._pack_cache[key] = (ver, _pad1d(bias, cout_p))
    return hit[1]


def prepack(mod, x: torch.Tensor, modes=(0, 1)) -> None:
    """Fill the packed-weight cache of conv block `mod` for input `x` on the CURRENT stream. Shared modules that are about
    to run on several side streams (detection head levels) must be packed before the fork: the cache itself is not
    stream-aware."""
    x_p, _ = phys(x)
    if mod.in_channels == 1:
        return
    desc = _desc(x_p, mod.in_channels, mod.out_channels, mod.k, mod.s, mod.p, mod.transposed)
    desc.cin_p = cpad(mod.in_channels)       # only the channel counts and the kernel enter the packing
    for mode in modes:
        _packed(mod, mode, mod.conv.weight, desc, x_p.dtype)
    _padded_bias(mod, mod.conv.bias, cpad(mod.out_channels))


def prepack_all(model: nn.Module, dtype: torch.dtype, modes=(0, 1), force: bool = False) -> int:
    """Refresh the packed-weight cache of EVERY conv block of `model` whose parameter changed (i.e. after an optimizer step) with
    one batched launch (nndet_pack_weights_batched) instead of one launch per layer and mode. Returns the number of jobs.
    `force`: re-pack everything regardless of the cached version -- a training-mode forward pass does this, because FUSED optimizer
    kernels (torch._fused_sgd_ / _fused_adam_, i.e. torch.optim.*(fused=True)) write the parameters without advancing `_version`
    (nndetection_amd.optim advances it by hand; a foreign optimizer may not).
    Host cost (round 6: 1.2 ms of the ~10 ms a step takes to enqueue went into this function): the list of conv blocks is cached on the
    model, the descriptor and size of a (block, mode, dtype) job in the block's cache, and a block's packed buffer is re-written IN PLACE
    (every kernel that read the previous contents was enqueued before this call on streams the caller's stream has joined: the packs run
    at the start of a forward pass, behind the optimizer step)."""
    jobs = []
    if force:
        PACK_EPOCH[0] += 1
    blocks = model.__dict__.get("_nndet_conv_blocks")
    n_mod = len(model._modules)
    if blocks is None or blocks[0] != n_mod or any(r() is None for r in blocks[1]):
        import weakref
        blocks = (n_mod, [weakref.ref(m) for m in model.modules() if isinstance(m, BaseConvNormAct)])
        model.__dict__["_nndet_conv_blocks"] = blocks      # (not a registered attribute: no submodule, nothing for state_dict)
    for ref in blocks[1]:
        mod = ref()
        if force:                                                    # EVERY block, the 1-channel stem included: its forward
            mod._pack_cache.pop(("bias", cpad(mod.out_channels)), None)      # reads the padded-bias cache too (_ConvFn.forward)
            mod._pack_cache.pop(("w32r", dtype), None)               # (arch/pyramid.py: rounded_w32)
        if mod.in_channels == 1:
            continue
        w = mod.conv.weight
        if not w.is_cuda:
            return 0
        _padded_bias(mod, mod.conv.bias, cpad(mod.out_channels))
        ver = _pver(w)
        for mode in modes:
            hit = mod._pack_cache.get((mode, dtype))
            if hit is not None and hit[0] == ver and not force:
                continue
            job = mod._pack_cache.get(("job", mode, dtype))
            if job is None:
                d = L.NndetConv()
                d.dtype = L._DT[dtype]
                d.transposed = int(mod.transposed)
                d.batch, d.cin, d.cout, d.cin_p, d.cout_p = 1, mod.in_channels, mod.out_channels, cpad(mod.in_channels), cpad(mod.out_channels)
                d.k = (ctypes.c_int32 * 3)(*mod.k); d.s = (ctypes.c_int32 * 3)(*mod.s); d.p = (ctypes.c_int32 * 3)(*mod.p)
                job = mod._pack_cache[("job", mode, dtype)] = (d, int(L.load().nndet_packed_weight_elems(ctypes.byref(d), mode)))
            d, n = job
            buf = hit[1] if (hit is not None and hit[1].numel() == n and hit[1].dtype == dtype and hit[1].device == w.device) else None
            if buf is None:
                buf = torch.empty((n,), dtype=dtype, device=w.device)
            w32 = w.detach()
            if w32.dtype != torch.float32 or not w32.is_contiguous():
                w32 = w32.float().contiguous()
            jobs.append((mod, mode, ver, d, w32, buf))
    if not jobs:
        return 0
    n = len(jobs)
    sig = tuple(j[4].data_ptr() for j in jobs) + tuple(j[5].data_ptr() for j in jobs) + tuple(j[1] for j in jobs) + (L._DT[dtype],)
    args = model.__dict__.get("_nndet_pack_args")
    if args is None or args[0] != sig:                      # (the same sources, destinations and modes as last step: the argument arrays are kept)
        convs = (L.NndetConv * n)(*[j[3] for j in jobs])
        modes_c = (ctypes.c_int32 * n)(*[j[1] for j in jobs])
        wp = (ctypes.c_void_p * n)(*[j[4].data_ptr() for j in jobs])
        op = (ctypes.c_void_p * n)(*[j[5].data_ptr() for j in jobs])
        args = model.__dict__["_nndet_pack_args"] = (sig, convs, modes_c, wp, op)
    L.call("nndet_pack_weights_batched", args[1], args[2], args[3], args[4], n, L.stream())
    for mod, mode, ver, _, _, buf in jobs:
        mod._pack_cache[(mode, dtype)] = (ver, buf)
    return n


def _pad1d(v: Optional[torch.Tensor], n: int) -> Optional[torch.Tensor]:
    if v is None:
        return None
    v = v.detach().float()
    return v.contiguous() if v.numel() == n else torch.nn.functional.pad(v, (0, n - v.numel()))


# Deferred normalisation is implemented and parity-tested on both routes, but OFF by default: every activation element is staged by
# at least two consumers (forward + weight gradient, each with ~2x halo amplification), so the norm arithmetic is done ~4x instead
# of once, in staging phases that are VALU / issue-bound rather than HBM-bound. Measured on one MI355X in the same process
# (profiles/round2_defer_norm_ab.txt): 148.6 / 146.6 patches/s deferred vs 153.1 / 147.8 materialised; peak HBM 4.8 vs 6.5 GiB.
DEFER_NORM = os.environ.get("NNDET_DEFER_NORM", "0") != "0"


# Early consumer (NNDET_EARLY_CONSUMER=1 enables it; OFF by default: measured 0.8 % slower, see below): the full-resolution output of
# encoder stage 0 is normalised in a 1.26 GB pass
# (k_norm_apply, 0.22 ms) that sits on the serial chain of the forward pass between the stage's last convolution and the first
# (stride-2) convolution of stage 1. With this on, that convolution reads the PRE-norm tensor and applies relu(x * scale + shift) while
# staging (the AFF kernels of the deferred route: the same arithmetic, bit-identical inputs), and the materialising pass -- still
# needed: the decoder, the segmentation branch and every backward kernel read the normalised tensor -- runs on an auxiliary stream next
# to it. The producer tags its output with `_nndet_pre = (pre-norm tensor, scale/shift table, relu, event)`; the consumer and the
# encoder make the current stream wait for the event before anything else can touch the normalised tensor.
# Measured (one gpurun session, alternating, luna160 batch 4 bf16): 14.94 / 14.93 ms per step with it, 14.80 / 14.83 without
# (profiles/round3_ab_early_consumer.txt) -- the stride-2 convolution and the norm pass are both HBM-heavy, side by side they take as
# long as one after the other, and the staging arithmetic comes on top. Bit-identical results (tests/test_model_gpu.py).
EARLY_CONSUMER = os.environ.get("NNDET_EARLY_CONSUMER", "0") != "0"
_pre_event = [None]

# Round 6: the consumer WRITES the normalised tensor (NNDET_NORM_INPUT_FUSE=0 disables it). Same producer / consumer pair, but no second
# pass at all: the producer only computes the coefficient table (nndet_norm_finalize) and hands on an UNWRITTEN output buffer tagged
# `_nndet_pre = (pre-norm tensor, table, relu, None, state)`; the stride-2 convolution reads the pre-norm tensor, transforms the halo
# in LDS and stores each tile's core voxels of the normalised tensor on the way (nndet_conv3d_forward_norm_input -> k_ig3s<.., PRE>,
# csrc/conv_ig3s.hip). 1.26 GB read + written once less on the serial chain of the forward pass; bit-identical values. Anybody else
# who meets the tag first (`ensure_materialized`: the encoder after the stage, a hook, a consumer the fused launch does not cover)
# writes the buffer with the plain nndet_affine_apply pass.
NORM_INPUT_FUSE = os.environ.get("NNDET_NORM_INPUT_FUSE", "1") != "0"
# The unwritten buffer may only exist between the two stages INSIDE the encoder's forward loop, which opens this scope around them and
# materialises whatever is left when the consumer stage returns. A stage called on its own, or one whose output somebody observes through a
# forward hook, takes the plain route.
_fill_scope = [False]


def _observed(mod) -> bool:
    """Forward hooks on this module (or registered globally) would see its output before the consumer has written it."""
    import torch.nn.modules.module as M
    return bool(mod._forward_hooks) or bool(getattr(M, "_global_forward_hooks", None))


def ensure_materialized(x: torch.Tensor) -> None:
    """Write the normalised tensor behind an unconsumed `_nndet_pre` fill tag (see NORM_INPUT_FUSE) on the current stream."""
    pre = getattr(x, "_nndet_pre", None)
    if pre is None or len(pre) < 5 or pre[4]["done"]:
        return
    y_p, _ = phys(pre[0])
    x_p, _ = phys(x)
    N, cp = y_p.shape[0], y_p.shape[4]
    spatial = y_p.shape[1] * y_p.shape[2] * y_p.shape[3]
    L.call("nndet_affine_apply", L.dtype_code(y_p), L.ptr(y_p), L.ptr(pre[1]), N, spatial, cp, int(pre[2]), L.ptr(x_p), L.stream())
    pre[4]["done"] = True


def deferred(x: torch.Tensor):
    """(scale_shift [N, C_p, 2] fp32, relu) if `x` is a DEFERRED activation -- the pre-norm output of a conv block whose
    InstanceNorm / GroupNorm (+ReLU) is applied by the CONSUMER convolution while it stages its input -- else None."""
    return getattr(x, "_nndet_deferred", None)


# ---- factorised (rank-1) output gradients -------------------------------------------------------------------------------------
# The fused segmentation head's gradient w.r.t. its input is d1[voxel] * wd[channel] (2-class softmax: the logit gradients are d1
# and -d1). Its autograd node hands on an UNWRITTEN tensor of the right shape and registers the factors here under that tensor's
# address; the backward pass of the convolution that produced the head's input (decoder.out.P0: nobody else reads its output) then
#   * gets its data gradient as a ONE-input-channel convolution of d1 with the kernel  Wf[cin][t] = sum_c wd[c] W[c][cin][2 - t]
#     (the stem kernels: 27 MACs per output instead of 27 x 32, reads 2 bytes per voxel instead of 64), and
#   * its weight gradient as  dW[c][cin][t] = wd[c] * E[cin][2 - t],  E = the one-channel weight gradient of (d1, conv input),
# i.e. 1.09 TFLOP and ~2.4 GB of the 8.9 TFLOP / 27 GB of a 160x160x96 training step disappear (NNDET_SEG_RANK1=0 disables it).
RANK1 = os.environ.get("NNDET_SEG_RANK1", "1") != "0"
_rank1_grads = {}
# Norm-backward sums that arrive with the gradient ("the reduce in the epilogue of the data gradient that produces its
# input"). An activation with two consumers of ours gets its complete gradient from the SECOND consumer's in-place accumulation
# (encoder.py set_fuse_grad_accum). When that consumer is the strided 3x3x3 data gradient (k_dgs) and the activation is the output of a
# materialised conv -> norm -> ReLU block, the kernel also accumulates the two per-channel sums that block's norm backward needs
# (nndet_conv3d_backward_data_acc_normred) and registers them here under the gradient buffer's address; _NormFn.backward then skips its
# reduction pass (nndet_norm_backward_presummed). At full resolution: one read of the pre-norm tensor (315 MB at batch 2) inside the
# data gradient instead of the 0.38 ms k_norm_bwd_reduce launch on the tail of the step. NNDET_NORM_RED_FUSE=0: the separate pass.
NORM_RED_FUSE = os.environ.get("NNDET_NORM_RED_FUSE", "1") != "0"
WGRAD_ALL = os.environ.get("NNDET_WGRAD_ALL", "1") != "0"      # 0: nodes whose bias gradient comes out of the data-gradient launch keep their weight gradient on the main stream
# Round 6: the same for the plain chains conv -> norm -> ReLU -> conv inside an encoder stage (ONE consumer, the stride-1 3x3x3 data
# gradient in k_ig3 writes dx and takes the sums: nndet_conv3d_backward_data_normred). The k_norm_bwd_reduce launches it replaces sit
# on the serial chain of the backward pass and run 2-3 x slower than alone next to the weight-gradient stream. NNDET_NORM_RED_CHAIN=0: off.
NORM_RED_CHAIN = os.environ.get("NNDET_NORM_RED_CHAIN", "1") != "0"
_norm_presums = {}          # gradient buffer address -> (red_ws, mean_rstd of the norm they belong to)
_last_mean_rstd = [None]    # _NormFn.forward -> BaseConvNormAct.forward (the tag of the block's output)
norm_red_fused = [0]        # how many norm backward passes took their sums from a data gradient (tests)


def rank1_register(fake: torch.Tensor, d1: torch.Tensor, wd: torch.Tensor, sum_d1: torch.Tensor) -> None:
    """fake: the unwritten [N, D, H, W, C_p] gradient tensor; d1 [N, D, H, W, 1]; wd [C] fp32; sum_d1: 0-dim fp32 (= sum of d1)."""
    _rank1_grads.clear()                                   # at most one live entry (one segmentation branch per backward pass)
    # the entry HOLDS `fake` (its address cannot be handed to another tensor while the entry lives) and its version counter: an
    # in-place write by anybody else (the engine accumulating a second contribution, a hook) voids it -- loudly, see below
    _rank1_grads[fake.data_ptr()] = (d1, wd, sum_d1, fake, fake._version)


def _rank1_backward(ctx, dconv, weight, x_p, desc, dw, dbias, side, raw):
    """Backward of a 3x3x3 / stride 1 / pad 1 convolution whose output gradient is d1 (x) wd (see above). Returns dx (physical)."""
    d1, wd, sum_d1, fake, ver = _rank1_grads.pop(dconv.data_ptr())
    if dconv._version != ver or fake._version != ver or dconv.numel() != fake.numel():
        raise L.NndetError("the factorised gradient of the fused segmentation head was modified in place before it reached decoder.out.P0 "
                           "(gradient accumulation / a hook on an unwritten tensor); set NNDET_SEG_RANK1=0 for the dense gradient")
    dev, dt = x_p.device, x_p.dtype
    cout, cin = desc.cout, desc.cin
    w_r = weight.detach().to(dt).float()                                     # what the forward kernels multiplied with
    wf = torch.einsum("c,cidhw->idhw", wd[:cout], w_r).flip(1, 2, 3).reshape(cin, 1, 3, 3, 3).contiguous()
    sd = L.NndetConv()
    sd.dtype, sd.transposed, sd.batch = desc.dtype, 0, desc.batch
    sd.cin, sd.cout, sd.cin_p, sd.cout_p = 1, cin, 1, desc.cin_p
    sd.in_d, sd.in_h, sd.in_w = desc.in_d, desc.in_h, desc.in_w
    sd.out_d, sd.out_h, sd.out_w = desc.in_d, desc.in_h, desc.in_w
    sd.k = (ctypes.c_int32 * 3)(3, 3, 3); sd.s = (ctypes.c_int32 * 3)(1, 1, 1); sd.p = (ctypes.c_int32 * 3)(1, 1, 1)
    dx_p = None
    if ctx.needs_input_grad[0]:
        dx_p = torch.empty_like(x_p)
        L.call("nndet_conv3d_forward", ctypes.byref(sd), L.ptr(d1), L.ptr(wf), None, None, L.ptr(dx_p), None, L.stream())
    # weight gradient on the weight-gradient stream: E[cin][t] = sum_p x[p][cin] d1[p + t - 1]  ->  dW[c][cin][t] = wd[c] E[cin][2 - t]
    if side is not None:
        for t in (x_p, d1, wd, sum_d1):
            t.record_stream(side)
    cur = torch.cuda.current_stream(dev)
    with torch.cuda.stream(side if side is not None else cur):
        e = torch.zeros((cin, 1, 3, 3, 3), dtype=torch.float32, device=dev)
        L.call("nndet_conv3d_backward_weight", ctypes.byref(sd), L.ptr(d1), L.ptr(x_p), L.ptr(e), None, None, 0, raw)
        torch.mul(wd[:cout].view(cout, 1, 1, 1, 1), e.flip(2, 3, 4).view(1, cin, 3, 3, 3), out=dw)
        if dbias is not None:
            torch.mul(wd[:cout], sum_d1, out=dbias)
    return dx_p


def _splitk_bytes(mod, desc, kind: int) -> int:
    """Workspace a split-K launch of this convolution wants (0: not a split-K problem); cached per module and problem."""
    key = ("splitk", kind, desc.dtype, desc.batch, desc.in_d, desc.in_h, desc.in_w, bool(desc.in_affine), os.environ.get("NNDET_IGEMM_SPLITK"),
           os.environ.get("NNDET_IGEMM_SMALLWG"))
    n = mod._pack_cache.get(key)
    if n is None:
        n = mod._pack_cache[key] = int(L.load().nndet_conv3d_splitk_workspace_bytes(ctypes.byref(desc), kind))
    return n


def _query(mod, desc, name: str, plan_env: bool = False) -> int:
    """A property of this convolution PROBLEM that the library answers from the descriptor alone (nndet_conv3d_<name>(desc): which kernel
    takes it, a workspace size), cached per module, problem geometry and the switches that steer the plan: a backward node asked three
    of them on every call, each a ctypes round trip and one a full plan construction (round 6: 82 us of host time per conv backward).
    plan_env: the answer depends on switches the library reads per call (tests flip them): they become part of the key."""
    key = ("q", name, desc.dtype, desc.batch, desc.in_d, desc.in_h, desc.in_w, bool(desc.in_affine), _PLAN_ENV() if plan_env else None)
    v = mod._pack_cache.get(key)
    if v is None:
        v = mod._pack_cache[key] = int(getattr(L.load(), "nndet_conv3d_" + name)(ctypes.byref(desc)))
    return v


def _PLAN_ENV():
    e = os.environ
    return (e.get("NNDET_IGEMM_SPLITK"), e.get("NNDET_IGEMM_SMALLWG"), e.get("NNDET_IGEMM_SPEC"), e.get("NNDET_IG3R"), e.get("NNDET_IG3S"), e.get("NNDET_DGSP"))


def rank1_branch_backward(x_p, a_p, w_out, b_out, w_lat, w_head, b_head, wd, wf, wfa, d1, dsum, need_dx, need_da, e_x_fn=None):
    """Backward of the composed segmentation branch (arch/segmenter.py: _SegBranchFn) from d1 = dL/d(l1 - l0) per voxel. The level-0
    map x (+ W_lat a), the output of `conv3x3x3(.; w_out) + b_out` and the logits were never computed; with wd = w_head[1] - w_head[0]:
        dx  = the ONE-input-channel convolution of d1 with wf[cin][t] = sum_c wd[c] W_out[c][cin][2 - t]        (stem forward kernel)
        da  = the same with wfa[k][t] = sum_c W_lat[c][k] wf[c][t]                                              (lateral absorbed)
        E_x[cin][t] = sum_p d1[p] x[p + t - 1][cin], E_a likewise for a                                         (stem weight gradient)
        every parameter gradient from E_x, E_a, sum(d1) in one small launch (csrc/segbranch.hip: k_segbranch_params).
    Returns (dx_p, da_p, dW_out, db_out, dW_lat, dW_head, db_head); the parameter gradients are views of the per-step gradient pool
    and are produced on the weight-gradient stream like every other weight gradient."""
    # x_p None (the top-down step absorbed too, arch/segmenter.py: NNDET_SEG_UP): the correlation E_x of d1 with the never-formed
    # top-down term comes from `e_x_fn(raw stream)`, called inside the weight-gradient stream context; no dx.
    ref_p = x_p if x_p is not None else a_p
    dev, dt = ref_p.device, ref_p.dtype
    N, D, H, W, cin_p = ref_p.shape
    cin = cout = w_out.shape[0]
    sd = L.NndetConv()
    sd.dtype, sd.transposed, sd.batch = L._DT[dt], 0, N
    sd.cin, sd.cout, sd.cin_p, sd.cout_p = 1, cin, 1, cin_p
    sd.in_d, sd.in_h, sd.in_w = D, H, W
    sd.out_d, sd.out_h, sd.out_w = D, H, W
    sd.k = (ctypes.c_int32 * 3)(3, 3, 3); sd.s = (ctypes.c_int32 * 3)(1, 1, 1); sd.p = (ctypes.c_int32 * 3)(1, 1, 1)
    nw, nb = w_out.numel(), (cout if b_out is not None else 0)
    nl = w_lat.numel() if w_lat is not None else 0
    nh, nhb = w_head.numel(), (2 if b_head is not None else 0)
    g_out, g_bo, g_lat, g_head, g_bh = L.grad_pool.take_for(
        [(w_out, nw), (b_out, nb), (w_lat, nl), (w_head, nh), (b_head, nhb)], dev)
    dw = g_out.view(w_out.shape)
    dbias = g_bo if nb else None
    dw_lat = g_lat.view(w_lat.shape) if nl else None
    dw_head = g_head.view(w_head.shape)
    db_head = g_bh if nhb else None
    side = L.wgrad_streams.side(dev, w_out)
    if side is not None:
        L.wgrad_streams.side(dev, w_head)                    # their gradients are produced on that stream as well
        if w_lat is not None:
            L.wgrad_streams.side(dev, w_lat)
    dx_p = da_p = None
    if need_dx and x_p is not None:
        dx_p = torch.empty_like(x_p)
        L.call("nndet_conv3d_forward", ctypes.byref(sd), L.ptr(d1), L.ptr(wf), None, None, L.ptr(dx_p), None, L.stream())
    if need_da and a_p is not None:
        da_p = torch.empty_like(a_p)
        L.call("nndet_conv3d_forward", ctypes.byref(sd), L.ptr(d1), L.ptr(wfa), None, None, L.ptr(da_p), None, L.stream())
    if side is not None:
        for t in (d1, wd, dsum) + ((x_p,) if x_p is not None else ()) + ((a_p,) if a_p is not None else ()):
            t.record_stream(side)
    cur = torch.cuda.current_stream(dev)
    raw = side.cuda_stream if side is not None else L.stream()
    with torch.cuda.stream(side if side is not None else cur):
        e = torch.zeros((2, cin, 27), dtype=torch.float32, device=dev)
        if x_p is not None:
            L.call("nndet_conv3d_backward_weight", ctypes.byref(sd), L.ptr(d1), L.ptr(x_p), L.ptr(e[0]), None, None, 0, raw)
        else:
            e[0].copy_(e_x_fn(side, raw))
        if a_p is not None:
            L.call("nndet_conv3d_backward_weight", ctypes.byref(sd), L.ptr(d1), L.ptr(a_p), L.ptr(e[1]), None, None, 0, raw)
        L.call("nndet_segbranch_param_grads", L.ptr(w_out.detach()), L.ptr(b_out.detach()) if b_out is not None else None,
               L.ptr(w_lat.detach()) if w_lat is not None else None, L.ptr(wd), L.ptr(e[0]), L.ptr(e[1]) if a_p is not None else None,
               L.ptr(dsum), int(dsum.numel()), L.ptr(dw), L.ptr(dbias), L.ptr(dw_lat), L.ptr(dw_head), L.ptr(db_head), raw)
    return dx_p, da_p, dw, dbias, dw_lat, dw_head, db_head


class _ConvFn(torch.autograd.Function):
    """conv (+bias, + fused residual). Input: a plain activation or a deferred one (then `x_ss` = its scale/shift table and the
    kernels read relu?(x * scale + shift): NndetConv.in_affine). Output: y and -- for blocks with a norm -- the per-(image, channel)
    sum / sum of squares of y accumulated by the conv epilogue (non-differentiable side output for _NormFn)."""

    @staticmethod
    def forward(ctx, x, x_ss, x_relu, weight, bias, mod, residual, want_stats, pre=None):
        x_p, cin = phys(x)
        if cin != mod.in_channels:
            raise L.NndetError(f"expected {mod.in_channels} input channels, got {cin}")
        desc = _desc(x_p, mod.in_channels, mod.out_channels, mod.k, mod.s, mod.p, mod.transposed)
        if x_ss is not None:
            if mod.transposed or desc.cin_p == 1 or tuple(x_ss.shape) != (desc.batch, desc.cin_p, 2):
                raise L.NndetError("deferred input normalisation: unsupported consumer or coefficient table shape")
            desc.in_affine, desc.in_relu = x_ss.data_ptr(), int(x_relu)
        x_bwd, desc_bwd = x_p, desc
        x_fill = None
        if pre is not None:                      # early consumer: THIS launch reads the pre-norm tensor + table; backward sees the plain input
            pre_p, _ = phys(pre[0])
            if x_ss is not None or tuple(pre_p.shape) != tuple(x_p.shape) or pre_p.dtype != x_p.dtype or tuple(pre[1].shape) != (desc.batch, desc.cin_p, 2):
                raise L.NndetError("early consumer: the pre-norm tensor does not match the input")
            desc = _desc(x_p, mod.in_channels, mod.out_channels, mod.k, mod.s, mod.p, mod.transposed)
            desc.in_affine, desc.in_relu = pre[1].data_ptr(), int(pre[2])
            x_fill = x_p if len(pre) > 4 else None   # NORM_INPUT_FUSE: this launch also WRITES the normalised input
            x_p = pre_p
        dev, dt = x_p.device, x_p.dtype
        N, cout, cout_p = desc.batch, desc.cout, desc.cout_p
        stem = desc.cin_p == 1
        w_arg = weight.detach().float().contiguous() if stem else _packed(mod, 0, weight, desc, dt)
        y = None
        dst = getattr(mod, "_out_buf", None)      # (arch/decoder.py: the head levels' outputs are written straight into the ragged batch buffer)
        if dst is not None:
            mod._out_buf = None
            if tuple(dst.shape) == (N, desc.out_d, desc.out_h, desc.out_w, cout_p) and dst.dtype == dt and dst.device == dev and dst.is_contiguous():
                y = dst
        if y is None:
            y = torch.empty((N, desc.out_d, desc.out_h, desc.out_w, cout_p), dtype=dt, device=dev)
        stats = L.arena_zeros((L.STATS_REPLICAS, N, cout_p, 2), torch.float64, dev) if want_stats else None
        b_p = _padded_bias(mod, bias, cout_p)
        r_p = None
        if residual is not None:
            if want_stats:
                raise L.NndetError("a fused residual is only defined for convolutions without norm")
            r_p, _ = phys(residual, dtype=dt, cp=cout_p)
            if tuple(r_p.shape) != tuple(y.shape):
                raise L.NndetError(f"residual shape {tuple(residual.shape)} does not match the conv output")
        sk = 0 if stem else _splitk_bytes(mod, desc, 0)
        if x_fill is not None:
            if residual is not None or stem:
                raise L.NndetError("norm-input fusion: unsupported consumer")
            L.call("nndet_conv3d_forward_norm_input", ctypes.byref(desc), L.ptr(x_p), L.ptr(x_fill), L.ptr(w_arg), L.ptr(b_p), L.ptr(y), L.ptr(stats),
                   L.stream())
            pre[4]["done"] = True
        elif sk:                                   # small problem: the channel chunks are spread over several workgroups per tile
            ws = L.workspace(sk, dev)
            L.call("nndet_conv3d_forward_ws", ctypes.byref(desc), L.ptr(x_p), L.ptr(w_arg), L.ptr(b_p), L.ptr(r_p), L.ptr(y), L.ptr(stats),
                   L.ptr(ws), sk, L.stream())
        else:
            L.call("nndet_conv3d_forward", ctypes.byref(desc), L.ptr(x_p), L.ptr(w_arg), L.ptr(b_p), L.ptr(r_p), L.ptr(y), L.ptr(stats), L.stream())
        ctx.desc, ctx.mod, ctx.has_bias, ctx.has_res = desc_bwd, mod, bias is not None, residual is not None
        ctx.gacc = None if mod.transposed else getattr(x, "_nndet_gacc", None)   # fused accumulation of the input gradient (encoder.py)
        ctx.norm_src = getattr(x, "_nndet_norm_src", None)    # (see _norm_presums; with or without a fused gradient accumulation)
        ctx.x_ss = x_ss                      # (tiny) keeps the table alive for the weight gradient
        ctx.save_for_backward(x_bwd, weight)
        out = logical(y, cout)
        if want_stats:
            ctx.mark_non_differentiable(stats)
            return out, stats
        return out, None

    @staticmethod
    def backward(ctx, grad_out, _grad_stats=None):
        desc, mod = ctx.desc, ctx.mod
        cout, cout_p = desc.cout, desc.cout_p
        x_p, weight = ctx.saved_tensors
        dev, dt = x_p.device, x_p.dtype
        dconv, _ = phys(grad_out, dtype=dt, cp=cout_p)
        # dW and dbias of this node are views of the per-step gradient pool (ONE zero fill per step, _lib.grad_pool)
        nw = weight.numel()
        g_w, g_b = L.grad_pool.take_for([(weight, nw), (mod.conv.bias if ctx.has_bias else None, cout if ctx.has_bias else 0)], dev)
        # (NNDET_WGRAD_ALL=0: nodes whose bias gradient comes out of the data-gradient launch keep their weight gradient on this stream)
        keep = (not WGRAD_ALL and ctx.has_bias and ctx.needs_input_grad[0] and desc.cin_p != 1 and
                bool(_query(mod, desc, "dgrad_fuses_bias")))
        ctx.wg_side = None if keep else L.wgrad_streams.side(dev, weight)
        dw = g_w.view(weight.shape)
        dbias = g_b if ctx.has_bias else None
        dx = None
        bias_from_dgrad = False
        if _rank1_grads and dconv.data_ptr() not in _rank1_grads and getattr(mod, "_nndet_rank1_consumer", False):
            # The segmentation head handed on a FACTORISED gradient (an unwritten tensor registered under its address, see
            # rank1_register) and something between the two nodes -- a tensor hook, gradient checkpointing, a wrapper that clones or
            # accumulates gradients -- gave this node another tensor: its content is undefined. Fail loudly instead of training on it.
            _rank1_grads.clear()
            raise L.NndetError("the factorised gradient of the fused segmentation head reached decoder.out.P0 as a COPY (a hook / "
                               "checkpointing / gradient accumulation in between?); set NNDET_SEG_RANK1=0 for the dense gradient")
        if dconv.data_ptr() in _rank1_grads:               # factorised output gradient (fused segmentation head): see _rank1_backward
            side = ctx.wg_side
            raw = side.cuda_stream if side is not None else L.stream()
            dx_p = _rank1_backward(ctx, dconv, weight, x_p, desc, dw, dbias, side, raw)
            return (logical(dx_p, desc.cin) if dx_p is not None else None), None, None, dw.to(weight.dtype), dbias, None, None, None, None
        if ctx.needs_input_grad[0]:
            if desc.cin_p == 1:
                raise L.NndetError("gradient w.r.t. the 1-channel input image is not implemented (never needed in training)")
            w1 = _packed(mod, 1, weight, desc, dt)
            gacc = ctx.gacc
            fuses_bias = dbias is not None and bool(_query(mod, desc, "dgrad_fuses_bias"))
            if gacc is not None and gacc["buf"] is not None and gacc["buf"].shape == x_p.shape and gacc["buf"].dtype == dt:
                # second consumer of this activation: add into the gradient the first consumer wrote and hand autograd nothing
                if gacc.get("ev") is not None:               # the first consumer may have run on another stream (decoder tail)
                    torch.cuda.current_stream(dev).wait_event(gacc["ev"])
                    gacc["buf"].record_stream(torch.cuda.current_stream(dev))
                ns = ctx.norm_src if NORM_RED_FUSE else None
                if (ns is not None and not fuses_bias and ns[0].shape == x_p.shape and ns[0].dtype == dt
                        and ns[2].norm.weight.dtype == torch.float32 and ns[2].norm.bias.dtype == torch.float32     # (read as raw fp32)
                        and ns[2].norm.weight.is_contiguous() and ns[2].norm.bias.is_contiguous()
                        and _query(mod, desc, "dgrad_fuses_norm_reduce")):
                    ny_p, nmr, nmod = ns
                    red = L.arena_zeros((L.STATS_REPLICAS * desc.batch * desc.cin_p * 2 + desc.batch,), torch.float64, dev)
                    L.call("nndet_conv3d_backward_data_acc_normred", ctypes.byref(desc), L.ptr(dconv), L.ptr(w1), L.ptr(gacc["buf"]),
                           L.ptr(ny_p), L.ptr(nmr), L.ptr(nmod.norm.weight.detach()), L.ptr(nmod.norm.bias.detach()), int(nmod.relu),
                           nmod.out_channels, L.ptr(red), L.stream())
                    _norm_presums.clear()                      # at most one live entry
                    # the entry holds the buffer (no address reuse while it lives) and its version counter: a further contribution
                    # that autograd adds IN PLACE into this storage (a third consumer, a hook) advances it and voids the sums
                    _norm_presums[gacc["buf"].data_ptr()] = (red, nmr, gacc["buf"], gacc["buf"]._version)
                else:
                    L.call("nndet_conv3d_backward_data_acc", ctypes.byref(desc), L.ptr(dconv), L.ptr(w1), L.ptr(gacc["buf"]),
                           L.ptr(dbias) if fuses_bias else None, L.stream())
                # This node hands autograd NO gradient (the sum lives in the first consumer's buffer), so the engine inserts no
                # stream synchronisation for it: order the producer's backward stream (= its forward stream) behind the accumulation
                # ourselves, whichever stream this node runs on.
                ps = gacc.get("stream")
                if ps is not None and x_p.is_cuda and ps != torch.cuda.current_stream(dev):
                    ev2 = torch.cuda.Event()
                    ev2.record()
                    ps.wait_event(ev2)
                gacc["buf"] = None
                bias_from_dgrad = fuses_bias
                dx_p = None
            else:
                dx_p = torch.empty_like(x_p)
            ns = ctx.norm_src if (NORM_RED_FUSE and NORM_RED_CHAIN and gacc is None and dx_p is not None) else None
            if dx_p is None:
                pass
            elif (ns is not None and ns[0].shape == x_p.shape and ns[0].dtype == dt and not _splitk_bytes(mod, desc, 1)
                    and ns[2].norm.weight.dtype == torch.float32 and ns[2].norm.bias.dtype == torch.float32     # (read as raw fp32)
                    and ns[2].norm.weight.is_contiguous() and ns[2].norm.bias.is_contiguous()
                    and _query(mod, desc, "dgrad_normred_supported", plan_env=True)):
                # x is the output of a materialised conv -> norm -> ReLU block and (as far as this node can know) read by this
                # convolution only: the data gradient also takes that block's norm-backward sums (k_ig3<.., NB>). If autograd adds another
                # contribution after all, the sum is another tensor or an in-place write (version counter): the entry is not used.
                ny_p, nmr, nmod = ns
                red = L.arena_zeros((L.STATS_REPLICAS * desc.batch * desc.cin_p * 2 + desc.batch,), torch.float64, dev)
                L.call("nndet_conv3d_backward_data_normred", ctypes.byref(desc), L.ptr(dconv), L.ptr(w1), L.ptr(dx_p),
                       L.ptr(ny_p), L.ptr(nmr), L.ptr(nmod.norm.weight.detach()), L.ptr(nmod.norm.bias.detach()), int(nmod.relu),
                       nmod.out_channels, L.ptr(red), L.stream())
                _norm_presums.clear()                      # at most one live entry
                _norm_presums[dx_p.data_ptr()] = (red, nmr, dx_p, dx_p._version)
            elif dbias is not None and _query(mod, desc, "dgrad_fuses_bias"):
                # pointwise kernels read every dy element once: the bias gradient comes out of the same pass
                L.call("nndet_conv3d_backward_data_bias", ctypes.byref(desc), L.ptr(dconv), L.ptr(w1), L.ptr(dx_p), L.ptr(dbias), L.stream())
                bias_from_dgrad = True
            else:
                sk = _splitk_bytes(mod, desc, 1)
                if sk:
                    ws = L.workspace(sk, dev)
                    L.call("nndet_conv3d_backward_data_ws", ctypes.byref(desc), L.ptr(dconv), L.ptr(w1), L.ptr(dx_p), L.ptr(ws), sk, L.stream())
                else:
                    L.call("nndet_conv3d_backward_data", ctypes.byref(desc), L.ptr(dconv), L.ptr(w1), L.ptr(dx_p), L.stream())
            if dx_p is not None:
                dx = logical(dx_p, desc.cin)  # gradient w.r.t. the input AS THE CONV SAW IT (i.e. after a deferred norm + ReLU)
                if gacc is not None:
                    gacc["buf"] = dx_p        # first consumer: the second one adds into this buffer
                    if dx_p.is_cuda:
                        gacc["ev"] = torch.cuda.Event()
                        gacc["ev"].record()   # (on this node's stream; the second consumer waits for it if it runs elsewhere)
        ws_bytes = L.load().nndet_conv3d_wgrad_workspace_bytes(ctypes.byref(desc))     # (not cached: the weight-gradient switches are read per call)
        side = ctx.wg_side                                 # weight-gradient stream (forked at the top of backward, before the data gradient)
        if side is not None:
            x_p.record_stream(side); dconv.record_stream(side)
        raw = side.cuda_stream if side is not None else L.stream()
        ws = L.workspace(ws_bytes, dev, raw_stream=raw if side is not None else None)
        L.call("nndet_conv3d_backward_weight", ctypes.byref(desc), L.ptr(x_p), L.ptr(dconv), L.ptr(dw),
               None if bias_from_dgrad else L.ptr(dbias), L.ptr(ws), ws_bytes, raw)
        # d(residual) = grad_out: the same NDHWC buffer is handed to both consumers (no copy, no add kernel)
        return dx, None, None, dw.to(weight.dtype), dbias, None, (logical(dconv, cout) if ctx.has_res else None), None, None


class _NormFn(torch.autograd.Function):
    """InstanceNorm / GroupNorm (+ReLU) of a conv output from the epilogue statistics.
    materialize=True : writes relu?(norm(y)) (one read + one write of the activation);
    materialize=False: DEFERRED -- only the coefficients are computed (nndet_norm_finalize); the returned tensor aliases y and is
                       tagged by the caller; every consumer convolution applies relu?(y * scale + shift) while staging its input,
                       so the normalised activation never exists in HBM. Backward is the same in both cases: from the gradient
                       w.r.t. the normalised activation (what the consumers' data gradients deliver) to dy, dgamma, dbeta."""

    @staticmethod
    def forward(ctx, y, gamma, beta, stats, mod, materialize):
        y_p, cout = phys(y)
        N, cout_p = y_p.shape[0], y_p.shape[4]
        spatial = y_p.shape[1] * y_p.shape[2] * y_p.shape[3]
        dev = y_p.device
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        mean_rstd = torch.empty((N, cout_p, 2), dtype=torch.float32, device=dev)
        code = L.dtype_code(y_p)
        if materialize and materialize not in (2, 3):
            out_p = torch.empty_like(y_p)
            L.call("nndet_norm_apply", code, L.ptr(y_p), L.ptr(stats), L.ptr(g32), L.ptr(b32), N, spatial, cout, cout_p,
                   mod.norm_groups, float(mod.norm_eps), int(mod.relu), L.ptr(out_p), L.ptr(mean_rstd), L.stream())
            ss = None
            out = logical(out_p, cout)
        else:
            ss = torch.empty((N, cout_p, 2), dtype=torch.float32, device=dev)
            L.call("nndet_norm_finalize", L.ptr(stats), L.ptr(g32), L.ptr(b32), N, spatial, cout, cout_p, mod.norm_groups,
                   float(mod.norm_eps), L.ptr(mean_rstd), L.ptr(ss), L.stream())
            if materialize == 3:                              # the first consumer writes it (see NORM_INPUT_FUSE)
                out = logical(torch.empty_like(y_p), cout)
            elif materialize == 2:                            # early consumer: materialise on the auxiliary stream (see EARLY_CONSUMER)
                out_p = torch.empty_like(y_p)
                cur, aux = torch.cuda.current_stream(dev), L.aux_stream(dev)
                ev = torch.cuda.Event()
                ev.record(cur)
                aux.wait_event(ev)
                L.call("nndet_affine_apply", code, L.ptr(y_p), L.ptr(ss), N, spatial, cout_p, int(mod.relu), L.ptr(out_p), aux.cuda_stream)
                done = torch.cuda.Event()
                done.record(aux)
                for t in (y_p, ss, out_p):
                    t.record_stream(aux)
                _pre_event[0] = done
                out = logical(out_p, cout)
            else:
                out = logical(y_p.view(y_p.shape), cout)      # an alias of y's storage (no kernel)
        ctx.mod, ctx.code, ctx.dims = mod, code, (N, spatial, cout, cout_p)
        ctx.save_for_backward(y_p, mean_rstd, g32, b32)
        _last_mean_rstd[0] = mean_rstd
        if ss is not None:
            ctx.mark_non_differentiable(ss)
        return out, ss

    @staticmethod
    def backward(ctx, grad_out, _grad_ss=None):
        mod = ctx.mod
        N, spatial, cout, cout_p = ctx.dims
        y_p, mean_rstd, g32, b32 = ctx.saved_tensors
        dev, dt = y_p.device, y_p.dtype
        g_p, _ = phys(grad_out, dtype=dt, cp=cout_p)
        dgamma, dbeta = L.grad_pool.take_for([(mod.norm.weight, cout), (mod.norm.bias, cout)], dev)
        dconv = torch.empty_like(y_p)
        pre = _norm_presums.pop(g_p.data_ptr(), None) if _norm_presums else None
        if pre is not None and (pre[2]._version != pre[3] or g_p._version != pre[3]):
            pre = None                                     # modified in place since the sums were taken: reduce again from the tensor
        if pre is not None and pre[1].data_ptr() == mean_rstd.data_ptr():      # the sums came with the gradient (see _norm_presums)
            norm_red_fused[0] += 1
            L.call("nndet_norm_backward_presummed", ctx.code, L.ptr(y_p), L.ptr(g_p), L.ptr(mean_rstd), L.ptr(g32), L.ptr(b32), N, spatial,
                   cout, cout_p, mod.norm_groups, int(mod.relu), L.ptr(dconv), L.ptr(dgamma), L.ptr(dbeta), L.ptr(pre[0]), L.stream())
            return logical(dconv, cout), dgamma, dbeta, None, None, None
        red = L.arena_zeros((L.STATS_REPLICAS * N * cout_p * 2 + N,), torch.float64, dev)     # replica sums + N ticket slots
        L.call("nndet_norm_backward", ctx.code, L.ptr(y_p), L.ptr(g_p), L.ptr(mean_rstd), L.ptr(g32), L.ptr(b32), N, spatial,
               cout, cout_p, mod.norm_groups, int(mod.relu), L.ptr(dconv), L.ptr(dgamma), L.ptr(dbeta), L.ptr(red), L.stream())
        return logical(dconv, cout), dgamma, dbeta, None, None, None


# Fused stem block (csrc/conv_stem.hip, nndet_stem_block_*): the first encoder block conv(1 -> C) -> InstanceNorm -> ReLU in 16-bit
# types recomputes its (27-MAC) convolution instead of storing the pre-norm tensor, and its backward pass is ONE pass over the
# incoming gradient. NNDET_STEM_FUSED=0 restores conv + norm as separate kernels (the fp32 path always uses those).
FUSED_STEM = os.environ.get("NNDET_STEM_FUSED", "1") != "0"


STEM_BWD_MAIN = os.environ.get("NNDET_STEM_BWD_MAIN", "1") != "0"


class _StemBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, gamma, beta, mod):
        x_p, cin = phys(x)
        desc = _desc(x_p, 1, mod.out_channels, mod.k, mod.s, mod.p, False)
        dev, dt = x_p.device, x_p.dtype
        N, cout, cout_p = desc.batch, desc.cout, desc.cout_p
        w32 = weight.detach().float().contiguous()
        g32, b32 = gamma.detach().float().contiguous(), beta.detach().float().contiguous()
        out = torch.empty((N, desc.out_d, desc.out_h, desc.out_w, cout_p), dtype=dt, device=dev)
        stats = L.arena_zeros((L.STATS_REPLICAS, N, cout_p, 2), torch.float64, dev)
        mean_rstd = torch.empty((N, cout_p, 2), dtype=torch.float32, device=dev)
        L.call("nndet_stem_block_forward", ctypes.byref(desc), L.ptr(x_p), L.ptr(w32), L.ptr(g32), L.ptr(b32), float(mod.norm_eps),
               int(mod.relu), L.ptr(out), L.ptr(stats), L.ptr(mean_rstd), L.stream())
        ctx.desc, ctx.mod = desc, mod
        ctx.save_for_backward(x_p, w32, g32, b32, mean_rstd, weight)
        return logical(out, cout)

    @staticmethod
    def backward(ctx, grad_out):
        desc, mod = ctx.desc, ctx.mod
        x_p, w32, g32, b32, mean_rstd, weight = ctx.saved_tensors
        dev, dt = x_p.device, x_p.dtype
        cout = desc.cout
        g_p, _ = phys(grad_out, dtype=dt, cp=desc.cout_p)
        nw = weight.numel()
        g_w, dgamma, dbeta = L.grad_pool.take_for([(weight, nw), (mod.norm.weight, cout), (mod.norm.bias, cout)], dev)
        dw = g_w.view(weight.shape)
        # Parameter gradients only -- but this is the LAST node of a backward pass: nothing is queued behind it on the data-gradient
        # chain, while the weight-gradient stream still has the full-resolution 32 -> 32 weight gradient in front of it. On THIS stream
        # the two run side by side (tools/phase_times.py: the step ended 0.3 ms later with it queued behind that weight gradient).
        # NNDET_STEM_BWD_MAIN=0: on the weight-gradient stream like every other weight gradient.
        side = None if STEM_BWD_MAIN else L.wgrad_streams.side(dev, weight)
        raw = side.cuda_stream if side is not None else L.stream()
        if side is not None:
            for t in (x_p, g_p, mean_rstd, w32, g32, b32):
                t.record_stream(side)
        ws_bytes = L.load().nndet_stem_block_backward_workspace_bytes(ctypes.byref(desc))
        ws = L.workspace(ws_bytes, dev, raw_stream=raw if side is not None else None)
        L.call("nndet_stem_block_backward", ctypes.byref(desc), L.ptr(x_p), L.ptr(g_p), L.ptr(w32), L.ptr(mean_rstd), L.ptr(g32),
               L.ptr(b32), int(mod.relu), L.ptr(dw), L.ptr(dgamma), L.ptr(dbeta), L.ptr(ws), ws_bytes, raw)
        return None, dw.to(weight.dtype), dgamma, dbeta, None


class BaseConvNormAct(nn.Sequential):
    def __init__(self, dim: int, in_channels: int, out_channels: int, norm: Optional[str], act: Optional[str],
                 kernel_size, stride=1, padding=0, dilation=1, groups: int = 1, bias: Optional[bool] = None,
                 transposed: bool = False, norm_kwargs: Optional[dict] = None, act_inplace: Optional[bool] = None,
                 initializer=None):
        super().__init__()
        if dim != 3:
            raise L.NndetError("only 3D convolutions are on the MI355X hot path")
        if _triple(dilation) != (1, 1, 1) or groups != 1:
            raise L.NndetError("dilation / grouped convolutions are not used by RetinaUNetV001 and not implemented")
        norm_kwargs = {} if norm_kwargs is None else dict(norm_kwargs)
        bias = bool(norm is None) if bias is None else bias               # conv.py:113
        conv_cls = nn.ConvTranspose3d if transposed else nn.Conv3d
        self.add_module("conv", conv_cls(in_channels, out_channels, kernel_size, stride=stride, padding=padding, bias=bias))
        self.in_channels, self.out_channels, self.transposed = in_channels, out_channels, bool(transposed)
        self.k, self.s, self.p = _triple(kernel_size), _triple(stride), _triple(padding)
        self.norm_groups, self.norm_eps, self.relu = 0, 1e-5, False
        if norm is not None:
            eps = norm_kwargs.get("eps", 1e-5)
            if not norm_kwargs.get("affine", True):
                raise L.NndetError("non-affine norms are not used by RetinaUNetV001")
            if norm.lower() == "instance":
                self.add_module("norm", nn.InstanceNorm3d(out_channels, eps=eps, affine=True))
                self.norm_groups = out_channels
            elif norm.lower() == "group":
                cpg = norm_kwargs.get("channels_per_group", 16)
                self.add_module("norm", nn.GroupNorm(out_channels // cpg, out_channels, eps=eps, affine=True))
                self.norm_groups = out_channels // cpg
            else:
                raise L.NndetError(f"norm {norm} is not on the hot path")
            self.norm_eps = eps
        if act is not None:
            if act != "ReLU":
                raise L.NndetError(f"activation {act} is not on the hot path")
            if norm is None:
                raise L.NndetError("an activation without a norm does not occur in RetinaUNetV001")
            self.add_module("act", nn.ReLU(inplace=bool(norm is not None) if act_inplace is None else act_inplace))
            self.relu = True
        self._pack_cache = {}
        self.defer_output = False        # set by our containers: the norm + ReLU of the output is applied by the consumer convs
        self.early_output = False        # set by the encoder: the output's first consumer reads the pre-norm tensor (see EARLY_CONSUMER)
        self.early_input = False         # ... and this is that consumer
        if initializer is not None:
            self.apply(initializer)

    def _norm_input_fused(self, x: torch.Tensor, pre) -> bool:
        """Does ONE launch cover this convolution reading `pre` and writing the normalised tensor (nndet_conv3d_forward_norm_input_fused)?"""
        x_p, _ = phys(x)
        desc = _desc(x_p, self.in_channels, self.out_channels, self.k, self.s, self.p, self.transposed)
        if tuple(pre[1].shape) != (desc.batch, desc.cin_p, 2):
            return False
        desc.in_affine, desc.in_relu = pre[1].data_ptr(), int(pre[2])
        return bool(L.load().nndet_conv3d_forward_norm_input_fused(ctypes.byref(desc)))

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        """residual (optional, only without norm): returns conv(x) + residual from ONE kernel (epilogue add).
        `x` may be a deferred activation (see `deferred`); with `self.defer_output` (set by OUR containers for outputs that only
        our own convolutions consume) the result is one: its norm + ReLU is applied by the consumers on load."""
        has_norm = self.norm_groups > 0
        d = deferred(x)
        pre = getattr(x, "_nndet_pre", None) if d is None else None
        if d is None:
            x = L.autocast_input(x)
            if residual is not None and residual.dtype != x.dtype:
                residual = residual.to(x.dtype)
        if (FUSED_STEM and d is None and residual is None and self.in_channels == 1 and has_norm and x.is_cuda and x.dim() == 5
                and x.dtype in (torch.bfloat16, torch.float16) and self.norm_groups == self.out_channels and not self.transposed
                and self.k == (3, 3, 3) and self.s == (1, 1, 1) and self.p == (1, 1, 1) and self.conv.bias is None
                and not (bool(self.defer_output) and DEFER_NORM)):
            return _StemBlockFn.apply(x, self.conv.weight, self.norm.weight, self.norm.bias, self)
        if d is not None and (self.transposed or self.in_channels == 1):
            x, d = materialize(x), None
        x_ss, x_relu = d if d is not None else (None, False)
        if pre is not None:
            use = (self.early_input and x.is_cuda and not self.transposed and self.in_channels != 1 and pre[0].dtype == x.dtype
                   and pre[0].shape == x.shape)
            if len(pre) > 4:                     # NORM_INPUT_FUSE: only the launch that writes the tensor while it stages may take it unwritten
                use = use and residual is None and not pre[4]["done"] and self._norm_input_fused(x, pre)
                if not use:
                    ensure_materialized(x)
                    pre = None
            elif not use:                        # not the consumer this was meant for: only order behind the materialising pass
                torch.cuda.current_stream(x.device).wait_event(pre[3])
                pre = None
        y, stats = _ConvFn.apply(x, x_ss, x_relu, self.conv.weight, self.conv.bias, self, residual, has_norm, pre)
        if pre is not None and pre[3] is not None:
            torch.cuda.current_stream(x.device).wait_event(pre[3])    # from here on the normalised tensor is complete for this stream
        if not has_norm:
            return y
        defer = bool(self.defer_output) and DEFER_NORM and y.is_cuda
        early = (not defer) and bool(self.early_output) and EARLY_CONSUMER and y.is_cuda
        fill = ((not defer) and (not early) and bool(self.early_output) and NORM_INPUT_FUSE and _fill_scope[0] and y.is_cuda
                and y.dtype in (torch.bfloat16, torch.float16) and not _observed(self))
        out, ss = _NormFn.apply(y, self.norm.weight, self.norm.bias, stats, self, 3 if fill else (2 if early else (not defer)))
        if defer:
            mark_padded(out)
            out._nndet_deferred = (ss, self.relu)
        elif fill:
            out._nndet_pre = (y, ss, self.relu, None, {"done": False})
        elif early:
            out._nndet_pre = (y, ss, self.relu, _pre_event[0])
        if NORM_RED_FUSE and not defer and y.is_cuda and out.requires_grad and self.norm_groups == self.out_channels:
            out._nndet_norm_src = (phys(y)[0].detach(), _last_mean_rstd[0], self)   # (see _norm_presums)
        _last_mean_rstd[0] = None
        return out


def materialize(x: torch.Tensor) -> torch.Tensor:
    """A deferred activation as a plain tensor (for a consumer that cannot apply the norm on load): one k_norm_apply pass,
    differentiable (the gradient flows back into the deferred tensor)."""
    d = deferred(x)
    if d is None:
        return x
    return _Materialize.apply(x, d[0], d[1])


class _Materialize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ss, relu):
        x_p, c = phys(x)
        out = torch.empty_like(x_p)
        N, cp = x_p.shape[0], x_p.shape[4]
        spatial = x_p.shape[1] * x_p.shape[2] * x_p.shape[3]
        L.call("nndet_affine_apply", L.dtype_code(x_p), L.ptr(x_p), L.ptr(ss), N, spatial, cp, int(relu), L.ptr(out), L.stream())
        ctx.save_for_backward(x_p, ss)
        ctx.relu, ctx.c = relu, c
        return logical(out, c)

    @staticmethod
    def backward(ctx, g):
        # d/d(a) -> d/d(a): the consumers of a deferred tensor hand back the gradient w.r.t. the NORMALISED activation, and so
        # does this node (the ReLU mask is applied by the producer's norm backward from y itself)
        return g, None, None


class ConvInstanceRelu(BaseConvNormAct):
    """conv -> InstanceNorm3d(affine) -> ReLU (nndet/arch/conv.py:146-217)."""

    def __init__(self, dim, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=None,
                 transposed=False, add_norm: bool = True, add_act: bool = True, act_inplace=None, norm_eps: float = 1e-5,
                 norm_affine: bool = True, initializer=None):
        super().__init__(dim, in_channels, out_channels, "Instance" if add_norm else None, "ReLU" if add_act else None,
                         kernel_size, stride, padding, dilation, groups, bias, transposed,
                         {"eps": norm_eps, "affine": norm_affine}, act_inplace, initializer)


class ConvGroupRelu(BaseConvNormAct):
    """conv -> GroupNorm(C / channels_per_group groups) -> ReLU (nndet/arch/conv.py:220-294)."""

    def __init__(self, dim, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias=None,
                 transposed=False, add_norm: bool = True, add_act: bool = True, act_inplace=None, norm_eps: float = 1e-5,
                 norm_affine: bool = True, norm_channels_per_group: int = 16, initializer=None):
        super().__init__(dim, in_channels, out_channels, "Group" if add_norm else None, "ReLU" if add_act else None,
                         kernel_size, stride, padding, dilation, groups, bias, transposed,
                         {"eps": norm_eps, "affine": norm_affine, "channels_per_group": norm_channels_per_group},
                         act_inplace, initializer)


def compute_padding_for_kernel(kernel_size):
    """nndet/arch/conv.py:455-470"""
    return tuple((i - 1) // 2 for i in kernel_size) if isinstance(kernel_size, Sequence) else (kernel_size - 1) // 2


def conv_kwargs_helper(norm: bool, activation: bool):
    """nndet/arch/conv.py:473-489"""
    return {"add_norm": norm, "add_act": activation}
