"""Foreground/background segmentation head with fused CE + SoftDice loss. Mirrors DiCESegmenterFgBg
(nndet/arch/heads/segmenter.py:30-289): one 1x1x1 conv on the finest decoder map -> 2 logits per voxel;
loss = alpha * CE + (1 - alpha) * SoftDice(softmax, batch_dice, no background, smooth 1e-5)
(nndet/losses/segmentation.py:32-151). The per-voxel part of the loss (softmax, CE, tp/fp/fn sums and their
gradient) is two streaming HIP kernels (csrc/segloss.hip); the scalar algebra on the four sums is autograd."""
import os
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


class _SegHeadFused(torch.autograd.Function):
    """Training-only fusion of the 1x1x1 output conv with the loss sums (csrc/segloss.hip, nndet_seghead_*): x = finest decoder
    map [N,C<=32,D,H,W], weight [2,C,1,1,1], bias [2], target uint8 -> fp32 [4] = (sum CE, tp, fp, fn). The padded logits and
    their gradient never reach HBM."""

    @staticmethod
    def forward(ctx, x, weight, bias, target_u8):
        xp, cin = phys(x)
        if xp.shape[-1] != 32 or weight.shape[0] != 2 or weight.shape[1] != cin:
            raise L.NndetError("fused segmentation head: needs <= 32 input channels and the 2-class 1x1x1 output conv")
        nvox = xp.shape[0] * xp.shape[1] * xp.shape[2] * xp.shape[3]
        w32 = weight.detach().reshape(2, cin).to(xp.dtype).float().contiguous()      # rounded like the packed conv weights
        b32 = bias.detach().float().contiguous() if bias is not None else None
        sums = torch.zeros((4,), dtype=torch.float64, device=xp.device)
        L.call("nndet_seghead_forward", L.dtype_code(xp), L.ptr(xp), 32, cin, L.ptr(w32), L.ptr(b32), L.ptr(target_u8), nvox,
               L.ptr(sums), L.stream())
        ctx.save_for_backward(xp, w32, target_u8)
        ctx.b32, ctx.cin, ctx.wshape, ctx.wdtype = b32, cin, tuple(weight.shape), weight.dtype
        # may the input gradient travel in factorised form (d1 (x) (w1 - w0), arch/conv.py: _rank1_backward)? Only if the tensor
        # was produced by one of OUR plain 3x3x3 / stride-1 convolutions and nobody else consumes it -- the caller says so
        ctx.rank1 = bool(getattr(x, "_nndet_rank1_ok", False)) and xp.dtype != torch.float32 and cin == xp.shape[-1]
        return sums.float()

    @staticmethod
    def backward(ctx, g):
        xp, w32, tgt = ctx.saved_tensors
        cin = ctx.cin
        nvox = xp.shape[0] * xp.shape[1] * xp.shape[2] * xp.shape[3]
        coeffs = g.detach().float().contiguous()
        dx = torch.empty_like(xp)
        dwb = torch.zeros((2 * cin + 2,), dtype=torch.float64, device=xp.device)
        from .conv import RANK1, rank1_register
        if ctx.rank1 and RANK1:
            d1 = torch.empty(xp.shape[:4] + (1,), dtype=xp.dtype, device=xp.device)
            L.call("nndet_seghead_backward_rank1", L.dtype_code(xp), L.ptr(xp), 32, cin, L.ptr(w32), L.ptr(ctx.b32), L.ptr(tgt), nvox,
                   L.ptr(coeffs), L.ptr(d1), L.ptr(dwb), L.stream())
            wd = torch.zeros((32,), dtype=torch.float32, device=xp.device)
            wd[:cin] = w32[1] - w32[0]
            # dx itself stays unwritten: its consumer reads the factors. Its first voxel is poisoned so that anything that DID read
            # it as a dense gradient (a tensor hook, a consumer the tag should have excluded) shows up as NaN instead of as noise
            dx[0, 0, 0, 0].fill_(float("nan"))
            rank1_register(dx, d1, wd, dwb[2 * cin + 1].float())
        else:
            L.call("nndet_seghead_backward", L.dtype_code(xp), L.ptr(xp), 32, cin, L.ptr(w32), L.ptr(ctx.b32), L.ptr(tgt), nvox,
                   L.ptr(coeffs), L.ptr(dx), L.ptr(dwb), L.stream())
        dw = dwb[:2 * cin].float().view(ctx.wshape).to(ctx.wdtype)
        db = dwb[2 * cin:].float() if ctx.b32 is not None else None
        return logical(dx, cin), dw, db, None


# decoder.out.P0 + the output conv + the loss as ONE 32 -> 1 convolution (csrc/segbranch.hip): the decoder hands its level-0 map
# on BEFORE the output convolution (tagged with that module, arch/decoder.py), `_SegBranchFn` does the rest. NNDET_SEG_BRANCH=0: the
# output convolution runs in the decoder and the head + loss as `_SegHeadFused`.
SEG_BRANCH = os.environ.get("NNDET_SEG_BRANCH", "1") != "0"
# ... and the decoder's level-0 lateral 1x1x1 convolution with it (the level-0 map is then never formed). NNDET_SEG_LATERAL=0: the
# lateral and the top-down add run in the decoder.
SEG_LATERAL = os.environ.get("NNDET_SEG_LATERAL", "1") != "0"


# ---- the last top-down step absorbed as well (NNDET_SEG_UP) ----------------------------------------------------------------------------
# With the lateral absorbed, the branch's second input is u = up_1(x_1) + b_up + b_lat: a k = s = 2 transposed convolution of the
# half-resolution map x_1 (64 channels), written (629 MB at 160x160x96, batch 4), read once by the branch, and in the backward pass its
# gradient written and read twice. conv3(u; wc) is linear in x_1: for the output voxel p = 2 m + pi (pi = its parity class) tap t reads
# u[p + t - 1] = u[2 (m + delta) + par], so
#     conv3(u; wc)[2 m + pi] = sum_{delta in {-1,0,1}^3, i} Wc[pi][i][delta] x_1[m + delta][i]  +  sum_{t: p + t - 1 inside} B[t]
#     Wc[pi][i][delta] = sum_{t -> (delta, par)} sum_k wc[t][k] W_up[i][k][par],      B[t] = wc[t] . (b_up + b_lat)
# -- ONE 3x3x3 / stride-1 convolution 64 -> 8 (one output channel per parity class) at HALF resolution on the existing MFMA kernels, plus
# a bias that depends on the border class of p only (27 values). Per axis: pi = 0: t = 0 -> (delta -1, par 1), 1 -> (0, 0), 2 -> (0, 1);
# pi = 1: t = 0 -> (0, 0), 1 -> (0, 1), 2 -> (+1, 0). Backward, from d1 = dL/dz: dz_up[m][pi] = d1[2 m + pi] (space-to-depth),
# dx_1 and dWc are that convolution's data / weight gradient, and with S[t] = sum of d1 over the voxels whose tap t stays inside:
#     dW_up[i][k][par] = sum_{(pi, delta) -> (t, par)} wc[t][k] dWc[pi][i][delta]
#     d(b_up)[k] = d(b_lat)[k] = sum_t wc[t][k] S[t]
#     Ec_u[t][k] = sum_p d1[p] u[p + t - 1][k] = sum_{(pi, delta) -> (t, par), i} W_up[i][k][par] dWc[pi][i][delta] + S[t] (b_up + b_lat)[k]
# (Ec_u is what k_segbranch_params needs of u for the gradients of decoder.out.P0 and the head). u never exists.
SEG_UP = os.environ.get("NNDET_SEG_UP", "1") != "0"


_up_consts = {}


def _up_tables(dtype, device):
    """T [8 * 27, 27 * 8]: T[(pi, delta), (t, par)] = 1 where tap t (3 axes) of an output voxel of parity class pi reads the transposed
    convolution's kernel position par at the half-resolution offset delta; K3 [27, 27]: K3[t, cls] = 1 if tap t of a voxel of border
    class cls stays inside the volume. Built once per (dtype, device)."""
    key = (dtype, str(device))
    hit = _up_consts.get(key)
    if hit is None:
        m = torch.zeros((2, 3, 3, 2), dtype=torch.float64)            # one axis: [pi][t][delta + 1][par]
        for pi, t, e, par in ((0, 0, 0, 1), (0, 1, 1, 0), (0, 2, 1, 1), (1, 0, 1, 0), (1, 1, 1, 1), (1, 2, 2, 0)):
            m[pi, t, e, par] = 1.0
        T = torch.einsum("adxp,beyq,cfzr->abcxyzdefpqr", m, m, m).reshape(8 * 27, 27 * 8)
        k = torch.ones((3, 3), dtype=torch.float64)                    # one axis: [t][cls]
        k[0, 0] = 0.0
        k[2, 2] = 0.0
        K3 = torch.einsum("ad,be,cf->abcdef", k, k, k).reshape(27, 27)
        hit = _up_consts[key] = (T.to(dtype=dtype, device=device), K3.to(dtype=dtype, device=device))
    return hit


def up_compose(wc: torch.Tensor, w_up: torch.Tensor, bsum: Optional[torch.Tensor]):
    """wc [27, C] (tap-major composed kernel), w_up [I, C, 2, 2, 2] (ConvTranspose3d layout), bsum [C] or None ->
    Wc [8, I, 3, 3, 3] (output channel = parity class (pd * 2 + ph) * 2 + pw) and the border-class bias cb [27]."""
    I, C = w_up.shape[0], wc.shape[1]
    T, K3 = _up_tables(wc.dtype, wc.device)
    A = torch.matmul(wc, w_up.to(wc.dtype).permute(1, 2, 3, 4, 0).reshape(C, 8 * I))          # [t, (par, i)] = sum_k wc[t, k] W_up[i, k, par]
    Wc = torch.matmul(T, A.reshape(27 * 8, I)).reshape(8, 27, I).permute(0, 2, 1).reshape(8, I, 3, 3, 3)
    cb = None
    if bsum is not None:
        cb = torch.matmul(torch.mv(wc, bsum.to(wc.dtype)), K3)           # cb[cls] = sum_t B[t] K3[t, cls]
    return Wc, cb


def up_param_grads(wc: torch.Tensor, w_up: torch.Tensor, bsum: Optional[torch.Tensor], dWc: torch.Tensor, cls_sums: torch.Tensor):
    """dWc [8, I, 3, 3, 3] (weight gradient of the composed convolution), cls_sums [27] (sum of d1 per border class) ->
    (dW_up [I, C, 2, 2, 2], dbsum [C], Ec_u [27, C]); see the derivation above."""
    I, C = w_up.shape[0], wc.shape[1]
    T, K3 = _up_tables(wc.dtype, wc.device)
    d = dWc.to(wc.dtype).reshape(8, I, 27).permute(0, 2, 1).reshape(8 * 27, I)
    dA = torch.matmul(T.t(), d).reshape(27, 8, I)                        # [t, par, i] = sum_{(pi, delta) -> (t, par)} dWc[pi][i][delta]
    S = torch.mv(K3, cls_sums.to(wc.dtype).reshape(27))                  # S[t] = sum over the classes whose tap t stays inside
    # .contiguous(): einsum hands back a PERMUTED view of its [C, 8 I] product. A gradient that does not obey autograd's layout contract
    # is not adopted by AccumulateGrad but CLONED -- on the parameter's stream, which the engine orders behind this node's stream only,
    # not behind the weight-gradient stream this runs on: the clone then reads the tensor before it is written (round 4: at batch 4
    # decoder.up.P1.conv.weight.grad came out as zeros / stale memory in some runs; found by
    # tests/test_parity_full_gpu.py::test_luna160_b4_benchmarked_route_vs_fp32_and_reference).
    dw_up = torch.einsum("tk,tpi->ikp", wc, dA).reshape(I, C, 2, 2, 2).contiguous()
    ec = torch.einsum("tpi,ikp->tk", dA, w_up.to(wc.dtype).reshape(I, C, 8))
    dbsum = torch.mv(wc.t(), S)
    if bsum is not None:
        ec = ec + S.unsqueeze(-1) * bsum.to(wc.dtype)
    return dw_up, dbsum, ec


_compose_cache = {}
# Round 6: the parameter-only part of the absorbed branch as ONE kernel (nndet_segbranch_compose_up) + the two weight packs instead of ~25
# torch launches of 4-6 us each on the branch's critical path (0.13 ms of chip time per step wherever they ran).
# NNDET_SEG_COMPOSE_FUSED=0: the torch expressions below (which stay the definition the kernel is tested against).
COMPOSE_FUSED = os.environ.get("NNDET_SEG_COMPOSE_FUSED", "1") != "0"


def _compose_fusable(params, dt) -> bool:
    w_lat, w_out, b_out, w_head, b_head, w_up, b_up, b_lat = params
    ts = [t for t in params if t is not None]
    return bool(w_out.is_cuda and dt in (torch.bfloat16, torch.float16) and all(t.dtype == torch.float32 and t.is_contiguous() for t in ts)
                and tuple(w_out.shape) == (32, 32, 3, 3, 3) and w_head.numel() == 64 and w_lat.numel() == 32 * 32
                and w_up.dim() == 5 and tuple(w_up.shape[1:]) == (32, 2, 2, 2) and 4 <= w_up.shape[0] <= 64 and w_up.shape[0] % 4 == 0
                and (b_out is None or b_out.numel() == 32) and (b_head is None or b_head.numel() == 2)
                and (b_up is None or b_up.numel() == 32) and (b_lat is None or b_lat.numel() == 32))


def _compose_up_branch_fused(params, dt):
    import ctypes
    w_lat, w_out, b_out, w_head, b_head, w_up, b_up, b_lat = params
    dev, I = w_out.device, int(w_up.shape[0])
    T, K3 = _up_tables(torch.float32, dev)
    seg = lambda n: (n + 63) // 64 * 64                      # every output starts on a 256-byte boundary of ONE allocation
    sizes = [32, 27 * 32, 1, 32 * 27, 32, 8 * I * 27, 27]
    offs, tot = [], 0
    for n in sizes:
        offs.append(tot); tot += seg(n)
    flat = torch.empty((tot,), dtype=torch.float32, device=dev)
    wd, wc, c0, wfa, bsum, Wc, cb = (flat[o:o + n] for o, n in zip(offs, sizes))
    wqa = torch.empty((27, 32), dtype=dt, device=dev)
    det = lambda t: t.detach() if t is not None else None
    L.call("nndet_segbranch_compose_up", L._DT[dt], L.ptr(det(w_out)), L.ptr(det(b_out)), L.ptr(det(w_head)), L.ptr(det(b_head)), L.ptr(det(w_lat)),
           L.ptr(det(w_up)), L.ptr(det(b_up)), L.ptr(det(b_lat)), L.ptr(T), L.ptr(K3), I, L.ptr(wd), L.ptr(wc), L.ptr(c0), L.ptr(wqa), L.ptr(wfa),
           L.ptr(bsum), L.ptr(Wc), L.ptr(cb), L.stream())
    Wc = Wc.view(8, I, 3, 3, 3)
    sd = L.NndetConv()                              # (packing depends on the channel counts / kernel only)
    sd.dtype, sd.transposed, sd.batch = L._DT[dt], 0, 1
    sd.cin, sd.cout, sd.cin_p, sd.cout_p = I, 8, (I + 31) // 32 * 32, 32
    sd.k = (ctypes.c_int32 * 3)(3, 3, 3); sd.s = (ctypes.c_int32 * 3)(1, 1, 1); sd.p = (ctypes.c_int32 * 3)(1, 1, 1)
    lib = L.load()
    pk = []
    for mode in (0, 1):
        buf = torch.empty((int(lib.nndet_packed_weight_elems(ctypes.byref(sd), mode)),), dtype=dt, device=dev)
        L.call("nndet_pack_weight", ctypes.byref(sd), mode, L.ptr(Wc), L.ptr(buf), L.stream())
        pk.append(buf)
    return {"wd": wd, "wc": wc.view(27, 32), "c0": c0, "wqa": wqa, "wfa": wfa.view(32, 1, 3, 3, 3), "bsum": bsum, "w_up32": w_up.detach(), "cb": cb, "pk": pk}


def _compose_up_branch(params, dt):
    """Everything of the absorbed branch that depends on the parameters only (a few dozen small launches): the composed kernels, the
    constant, the border-class bias and the two packed copies of the half-resolution convolution's weights. On the CURRENT stream."""
    import ctypes
    w_lat, w_out, b_out, w_head, b_head, w_up, b_up, b_lat = params
    if not torch.is_grad_enabled():                 # inference: the parameters do not change between calls
        from .conv import _pver
        key = (dt,) + tuple(_pver(t) if t is not None else None for t in params)
        if _compose_cache.get("key") == key:
            return _compose_cache["val"]
        with torch.enable_grad():                   # (only to take the uncached route below)
            val = _compose_up_branch(params, dt)
        _compose_cache["key"], _compose_cache["val"] = key, val
        return val
    dev = w_out.device
    cout, cin, cin1 = w_out.shape[0], w_out.shape[1], w_up.shape[0]
    if COMPOSE_FUSED and _compose_fusable(params, dt):
        return _compose_up_branch_fused(params, dt)
    wh = w_head.detach().reshape(2, cout).float()
    wd = (wh[1] - wh[0]).contiguous()
    wc = torch.einsum("c,cidhw->dhwi", wd, w_out.detach().float()).reshape(27, cin).contiguous()
    c0 = torch.zeros((), dtype=torch.float32, device=dev)
    if b_out is not None:
        c0 = c0 + (wd * b_out.detach().float()).sum()
    if b_head is not None:
        c0 = c0 + (b_head.detach()[1] - b_head.detach()[0]).float()
    c0 = c0.reshape(1).contiguous()
    wca = torch.matmul(wc, w_lat.detach().float().reshape(cin, 32))
    wqa = wca.to(dt).contiguous()
    wfa = wca.flip(0).t().contiguous().view(32, 1, 3, 3, 3)
    bs = [b.detach().float() for b in (b_up, b_lat) if b is not None]
    bsum = (bs[0] + bs[1]) if len(bs) == 2 else (bs[0] if bs else torch.zeros((cin,), dtype=torch.float32, device=dev))
    w_up32 = w_up.detach().float().contiguous()
    Wc, cb = up_compose(wc, w_up32, bsum)
    Wc, cb = Wc.contiguous(), cb.contiguous()
    sd = L.NndetConv()                              # (packing depends on the channel counts / kernel only)
    sd.dtype, sd.transposed, sd.batch = L._DT[dt], 0, 1
    sd.cin, sd.cout, sd.cin_p, sd.cout_p = cin1, 8, (cin1 + 31) // 32 * 32, 32
    sd.k = (ctypes.c_int32 * 3)(3, 3, 3); sd.s = (ctypes.c_int32 * 3)(1, 1, 1); sd.p = (ctypes.c_int32 * 3)(1, 1, 1)
    lib = L.load()
    pk = []
    for mode in (0, 1):
        buf = torch.empty((int(lib.nndet_packed_weight_elems(ctypes.byref(sd), mode)),), dtype=dt, device=dev)
        L.call("nndet_pack_weight", ctypes.byref(sd), mode, L.ptr(Wc), L.ptr(buf), L.stream())
        pk.append(buf)
    return {"wd": wd, "wc": wc, "c0": c0, "wqa": wqa, "wfa": wfa, "bsum": bsum, "w_up32": w_up32, "cb": cb, "pk": pk}


class _SegBranchFn(torch.autograd.Function):
    """x = the decoder's level-0 map BEFORE decoder.out.P0 [N, 32, D, H, W] (16-bit), w_out [32, 32, 3, 3, 3] / b_out of that
    convolution, w_head [2, 32, 1, 1, 1] / b_head of the segmenter's output conv, target uint8 -> fp32 [4] = (sum CE, tp, fp, fn).
    Forward: z = l1 - l0 per voxel by one composed 32 -> 1 convolution; backward: d1 = dL/dz, everything else from d1
    (arch/conv.py: rank1_branch_backward). Neither the 32-channel output of decoder.out.P0 nor the logits ever exist.
    With (a0, w_lat) the decoder's level-0 LATERAL is absorbed as well: the level-0 map is x + W_lat a0 (a0 = the encoder's
    full-resolution output, x = the top-down term up_1(x_1) carrying both biases), which never exists either:
    z = conv3(a0; wc . W_lat) + conv3(x; wc) + c0 (nndet_segbranch_forward2)."""

    @staticmethod
    def forward(ctx, x, a0, w_lat, w_out, b_out, w_head, b_head, target_u8, w_up=None, b_up=None, b_lat=None):
        ctx.up = w_up is not None
        if ctx.up:
            return _SegBranchFn._forward_up(ctx, x, a0, w_lat, w_out, b_out, w_head, b_head, target_u8, w_up, b_up, b_lat)
        xp, cin = phys(x)
        dev, dt = xp.device, xp.dtype
        N, D, H, W, cp = xp.shape
        cout = w_out.shape[0]
        if cp != 32 or cin != 32 or tuple(w_out.shape[1:]) != (32, 3, 3, 3) or tuple(w_head.shape[:2]) != (2, cout) or dt == torch.float32:
            raise L.NndetError("fused segmentation branch: 32 channels in 16 bits, a 3x3x3 output conv and the 2-class head")
        wh = w_head.detach().reshape(2, cout).float()
        wd = (wh[1] - wh[0]).contiguous()                                              # [cout]
        wc = torch.einsum("c,cidhw->dhwi", wd, w_out.detach().float()).reshape(27, cin)   # composed kernel, tap-major
        wq = wc.to(dt).contiguous()
        # the flipped kernels of the backward pass (data gradients as one-input-channel convolutions of d1), made here where the host
        # has time to spare: wf[cin][t] = wc[26 - t][cin]
        wf = wc.flip(0).t().contiguous().view(cin, 1, 3, 3, 3)
        wfa = wd                                                                        # (placeholder, replaced with the lateral)
        c0 = torch.zeros((), dtype=torch.float32, device=dev)
        if b_out is not None:
            c0 = c0 + (wd * b_out.detach().float()).sum()
        if b_head is not None:
            c0 = c0 + (b_head.detach()[1] - b_head.detach()[0]).float()
        c0 = c0.reshape(1).contiguous()
        z = torch.empty((N, D, H, W), dtype=torch.float32, device=dev)
        R = int(L.load().nndet_segbranch_replicas())
        sums = torch.zeros((R, 4), dtype=torch.float64, device=dev)
        ap = None
        if a0 is not None:
            ap, ka = phys(a0)
            if tuple(ap.shape) != tuple(xp.shape) or ap.dtype != dt or ka != 32 or tuple(w_lat.shape[:2]) != (cin, 32):
                raise L.NndetError("fused segmentation branch: the absorbed lateral needs a 32 -> 32 1x1x1 convolution of a same-sized map")
            wca = torch.matmul(wc, w_lat.detach().float().reshape(cin, 32))                 # [27][k] = sum_c wc[t][c] W_lat[c][k]
            wqa = wca.to(dt).contiguous()
            wfa = wca.flip(0).t().contiguous().view(32, 1, 3, 3, 3)
            L.call("nndet_segbranch_forward2", L.dtype_code(xp), L.ptr(ap), L.ptr(wqa), L.ptr(xp), L.ptr(wq), N, D, H, W, cp, L.ptr(c0),
                   L.ptr(target_u8), L.ptr(z), L.ptr(sums), L.stream())
        else:
            L.call("nndet_segbranch_forward", L.dtype_code(xp), L.ptr(xp), N, D, H, W, cp, L.ptr(wq), L.ptr(c0), L.ptr(target_u8), L.ptr(z),
                   L.ptr(sums), L.stream())
        ctx.save_for_backward(xp, w_out, b_out if b_out is not None else wd, w_head, b_head if b_head is not None else wd, target_u8, z, wd,
                              ap if ap is not None else wd, w_lat if w_lat is not None else wd, wf, wfa)
        ctx.has_b_out, ctx.has_b_head, ctx.R, ctx.cin, ctx.has_lat = b_out is not None, b_head is not None, R, cin, ap is not None
        ctx.gacc = getattr(a0, "_nndet_gacc", None) if a0 is not None else None       # fused accumulation of a0's gradient (encoder.py)
        return sums.sum(0).float()

    # ---- the top-down step absorbed too: x = x_1 [N, I, D/2, H/2, W/2] (see the derivation above SEG_UP)
    @staticmethod
    def _up_desc(x1p, cin1):
        import ctypes
        sd = L.NndetConv()
        N, D2, H2, W2, cp1 = x1p.shape
        sd.dtype, sd.transposed, sd.batch = L._DT[x1p.dtype], 0, N
        sd.cin, sd.cout, sd.cin_p, sd.cout_p = cin1, 8, cp1, 32
        sd.in_d, sd.in_h, sd.in_w = D2, H2, W2
        sd.out_d, sd.out_h, sd.out_w = D2, H2, W2
        sd.k = (ctypes.c_int32 * 3)(3, 3, 3); sd.s = (ctypes.c_int32 * 3)(1, 1, 1); sd.p = (ctypes.c_int32 * 3)(1, 1, 1)
        return sd

    @staticmethod
    def _forward_up(ctx, x1, a0, w_lat, w_out, b_out, w_head, b_head, target_u8, w_up, b_up, b_lat):
        import ctypes
        x1p, cin1 = phys(x1)
        ap, ka = phys(a0)
        dev, dt = ap.device, ap.dtype
        N, D, H, W, cp = ap.shape
        cout = w_out.shape[0]
        cin = 32
        if (cp != 32 or ka != 32 or tuple(w_out.shape[1:]) != (32, 3, 3, 3) or tuple(w_head.shape[:2]) != (2, cout) or dt == torch.float32
                or x1p.dtype != dt or tuple(x1p.shape[:4]) != (N, D // 2, H // 2, W // 2) or (D | H | W) & 1
                or tuple(w_up.shape) != (cin1, cin, 2, 2, 2) or tuple(w_lat.shape[:2]) != (cin, 32)):
            raise L.NndetError("fused segmentation branch with the top-down step: 16 bits, even dims, a k = s = 2 transposed convolution C1 -> 32")
        params = (w_lat, w_out, b_out, w_head, b_head, w_up, b_up, b_lat)
        pre = _compose_up_branch(params, dt)
        wd, wc, c0, wqa, wfa, bsum, w_up32, cb, pk = (pre[k] for k in ("wd", "wc", "c0", "wqa", "wfa", "bsum", "w_up32", "cb", "pk"))
        sd = _SegBranchFn._up_desc(x1p, cin1)
        lib = L.load()
        zup = torch.empty((N, D // 2, H // 2, W // 2, 32), dtype=dt, device=dev)
        L.call("nndet_conv3d_forward", ctypes.byref(sd), L.ptr(x1p), L.ptr(pk[0]), None, None, L.ptr(zup), None, L.stream())
        z = torch.empty((N, D, H, W), dtype=torch.float32, device=dev)
        R = int(lib.nndet_segbranch_replicas())
        sums = torch.zeros((R, 4), dtype=torch.float64, device=dev)
        L.call("nndet_segbranch_forward_up", L.dtype_code(ap), L.ptr(ap), L.ptr(wqa), L.ptr(zup), L.ptr(cb), N, D, H, W, cp, L.ptr(c0),
               L.ptr(target_u8), L.ptr(z), L.ptr(sums), L.stream())
        ctx.save_for_backward(x1p, w_out, b_out if b_out is not None else wd, w_head, b_head if b_head is not None else wd, target_u8, z, wd,
                              ap, w_lat, wfa, wc, w_up32, bsum, pk[1], w_up)
        ctx.has_b_out, ctx.has_b_head, ctx.R, ctx.cin, ctx.cin1 = b_out is not None, b_head is not None, R, cin, cin1
        ctx.has_b_up, ctx.has_b_lat = b_up is not None, b_lat is not None
        ctx.b_refs = [b for b in (b_up, b_lat) if b is not None]             # (parameters: their gradients come from the weight-gradient stream)
        ctx.gacc = getattr(a0, "_nndet_gacc", None)
        return sums.sum(0).float()

    @staticmethod
    def _backward_up(ctx, g):
        import ctypes
        (x1p, w_out, b_out, w_head, b_head, tgt, z, wd, ap, w_lat, wfa, wc, w_up32, bsum, pk1, w_up) = ctx.saved_tensors
        dev, dt = ap.device, ap.dtype
        N, D, H, W, _ = ap.shape
        coeffs = g.detach().float().contiguous()
        d1 = torch.empty(ap.shape[:4] + (1,), dtype=dt, device=dev)
        dsum = torch.zeros((ctx.R,), dtype=torch.float64, device=dev)
        L.call("nndet_segbranch_backward", L.dtype_code(ap), L.ptr(z), L.ptr(tgt), z.numel(), L.ptr(coeffs), L.ptr(d1), L.ptr(dsum), L.stream())
        dzs = torch.empty((N, D // 2, H // 2, W // 2, 32), dtype=dt, device=dev)
        csum = torch.zeros((ctx.R, 27), dtype=torch.float64, device=dev)
        L.call("nndet_segbranch_s2d", L.dtype_code(ap), L.ptr(d1), N, D, H, W, L.ptr(dzs), L.ptr(csum), L.stream())
        sd = _SegBranchFn._up_desc(x1p, ctx.cin1)
        dx1_p = None
        if ctx.needs_input_grad[0]:
            dx1_p = torch.empty_like(x1p)
            L.call("nndet_conv3d_backward_data", ctypes.byref(sd), L.ptr(dzs), L.ptr(pk1), L.ptr(dx1_p), L.stream())
        out = {}

        def e_x_fn(side, raw):                       # runs inside the weight-gradient stream context (arch/conv.py)
            if side is not None:
                for t in (x1p, dzs, csum, wc, w_up32, bsum):
                    t.record_stream(side)
                L.wgrad_streams.side(dev, w_up)
                for b in ctx.b_refs:
                    L.wgrad_streams.side(dev, b)
            dWc = torch.zeros((8, ctx.cin1, 3, 3, 3), dtype=torch.float32, device=dev)
            ws_bytes = L.load().nndet_conv3d_wgrad_workspace_bytes(ctypes.byref(sd))
            ws = L.workspace(ws_bytes, dev, raw_stream=raw if side is not None else None)
            L.call("nndet_conv3d_backward_weight", ctypes.byref(sd), L.ptr(x1p), L.ptr(dzs), L.ptr(dWc), None, L.ptr(ws), ws_bytes, raw)
            dw_up, dbsum, ec = up_param_grads(wc, w_up32, bsum, dWc, csum.sum(0).float())
            out["dw_up"], out["dbsum"] = dw_up, dbsum
            out["dbsum2"] = dbsum.clone() if (ctx.has_b_up and ctx.has_b_lat) else dbsum     # two parameters never share one gradient tensor
            return ec.flip(0).t()                   # Ec[t][k] -> the kernels' E[k][26 - t]

        from .conv import rank1_branch_backward
        _, da_p, dw_out, db_out, dw_lat, dw_head, db_head = rank1_branch_backward(
            None, ap, w_out, b_out if ctx.has_b_out else None, w_lat, w_head, b_head if ctx.has_b_head else None, wd, None, wfa, d1, dsum,
            False, ctx.needs_input_grad[1], e_x_fn=e_x_fn)
        da = _SegBranchFn._a0_grad(ctx, da_p, dev)
        dbs = out["dbsum"]
        for t_, p_ in ((out["dw_up"], w_up), (dbs, None), (out["dbsum2"], None)):
            # produced on the weight-gradient stream: autograd must ADOPT these tensors (see up_param_grads), never clone them
            if not t_.is_contiguous() or (p_ is not None and tuple(t_.shape) != tuple(p_.shape)):
                raise L.NndetError("a parameter gradient of the absorbed segmentation branch violates autograd's layout contract")
        return ((logical(dx1_p, ctx.cin1) if dx1_p is not None else None), da, dw_lat.to(w_lat.dtype), dw_out.to(w_out.dtype), db_out,
                dw_head.to(w_head.dtype), db_head, None, out["dw_up"].to(w_up.dtype), dbs if ctx.has_b_up else None,
                out["dbsum2"] if ctx.has_b_lat else None)

    @staticmethod
    def _a0_grad(ctx, da_p, dev):
        """The gradient of a0 through the fused-accumulation protocol of the encoder outputs (encoder.py: set_fuse_grad_accum)."""
        if da_p is None:
            return None
        gacc = ctx.gacc
        if gacc is not None and gacc["buf"] is not None and gacc["buf"].shape == da_p.shape and gacc["buf"].dtype == da_p.dtype:
            # another consumer of a0 wrote its gradient first (not the usual order: this node is the last one created): add
            if gacc.get("ev") is not None:
                torch.cuda.current_stream(dev).wait_event(gacc["ev"])
                gacc["buf"].record_stream(torch.cuda.current_stream(dev))
            gacc["buf"].add_(da_p)
            ps = gacc.get("stream")
            if ps is not None and ps != torch.cuda.current_stream(dev):
                ev2 = torch.cuda.Event(); ev2.record(); ps.wait_event(ev2)
            gacc["buf"] = None
            return None
        if gacc is not None:                       # first consumer: the other one adds into this buffer (conv.py: _ConvFn.backward)
            gacc["buf"] = da_p
            gacc["ev"] = torch.cuda.Event()
            gacc["ev"].record()
        return logical(da_p, 32)

    @staticmethod
    def backward(ctx, g):
        if ctx.up:
            return _SegBranchFn._backward_up(ctx, g)
        xp, w_out, b_out, w_head, b_head, tgt, z, wd, ap, w_lat, wf, wfa = ctx.saved_tensors
        dev = xp.device
        nvox = z.numel()
        coeffs = g.detach().float().contiguous()
        d1 = torch.empty(xp.shape[:4] + (1,), dtype=xp.dtype, device=dev)
        dsum = torch.zeros((ctx.R,), dtype=torch.float64, device=dev)
        L.call("nndet_segbranch_backward", L.dtype_code(xp), L.ptr(z), L.ptr(tgt), nvox, L.ptr(coeffs), L.ptr(d1), L.ptr(dsum), L.stream())
        from .conv import rank1_branch_backward
        dx_p, da_p, dw_out, db_out, dw_lat, dw_head, db_head = rank1_branch_backward(
            xp, ap if ctx.has_lat else None, w_out, b_out if ctx.has_b_out else None, w_lat if ctx.has_lat else None, w_head,
            b_head if ctx.has_b_head else None, wd, wf, wfa, d1, dsum, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        da = _SegBranchFn._a0_grad(ctx, da_p, dev)
        return ((logical(dx_p, ctx.cin) if dx_p is not None else None), da, (dw_lat.to(w_lat.dtype) if ctx.has_lat else None),
                dw_out.to(w_out.dtype), db_out, dw_head.to(w_head.dtype), db_head, None, None, None, None)


class _SegTail(torch.autograd.Function):
    """(sum CE, tp, fp, fn) fp32 [4] -> (seg_ce, seg_dice) fp32 [2] and, for backward, their 2 x 4 Jacobian: one launch
    (csrc/segloss.hip k_segloss_tail) instead of the scalar algebra as ~45 one-element torch launches."""

    @staticmethod
    def forward(ctx, s, nvox, alpha, sn, sd):
        s = s.detach().float().contiguous()
        out = torch.empty((2,), dtype=torch.float32, device=s.device)
        jac = torch.empty((2, 4), dtype=torch.float32, device=s.device)
        L.call("nndet_segloss_tail_f32", L.ptr(s), int(nvox), float(alpha), float(sn), float(sd), L.ptr(out), L.ptr(jac), L.stream())
        ctx.save_for_backward(jac)
        return out

    @staticmethod
    def backward(ctx, g):
        jac, = ctx.saved_tensors
        return torch.mv(jac.t(), g.float()), None, None, None, None


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
        self._zero_target = {}                       # (shape, device) -> uint8 zeros for logit_difference

    fused_tail = os.environ.get("NNDET_SEG_TAIL", "1") != "0"      # the scalar algebra on the four sums as one kernel (_SegTail)

    def takes_fused_route(self, x: List[Tensor]) -> bool:
        """Would forward(x, fused=True) hand the decoder map on unread (so that nothing of this module touches it on the caller's
        stream)? BaseRetinaNet.forward decides its stream join from this."""
        return bool(x[0].is_cuda and self.in_channels[0] <= 32 and os.environ.get("NNDET_SEG_FUSED", "1") != "0")

    def forward(self, x: List[Tensor], fused: bool = False) -> Dict[str, Tensor]:
        """fused=True (training steps that do not need the logits): hand the decoder map to compute_loss, which runs the output
        conv and the loss in one pass; else the logits as in the reference."""
        if fused and self.takes_fused_route(x):
            return {"seg_input": x[0]}
        return {"seg_logits": self.conv_out(x[0])}

    def compute_loss(self, pred_seg: Dict[str, Tensor], target: Tensor) -> Dict[str, Tensor]:
        tgt = (target > 0).to(torch.uint8).contiguous()          # segmenter.py:288 binarises in place
        pre = getattr(pred_seg.get("seg_input"), "_nndet_pre_out", None) if "seg_input" in pred_seg else None
        if pre is not None:                                       # the decoder skipped its output convolution: the whole branch here
            lat = getattr(pred_seg["seg_input"], "_nndet_pre_lat", None)      # (lateral module, its input): absorbed as well
            up = getattr(pred_seg["seg_input"], "_nndet_pre_up", None) if lat else None   # ... and the last top-down step: seg_input is x_1
            if up is not None:
                s = _SegBranchFn.apply(pred_seg["seg_input"], lat[1], lat[0].conv.weight, pre.conv.weight, pre.conv.bias,
                                       self.conv_out.conv.weight, self.conv_out.conv.bias, tgt, up.conv.weight, up.conv.bias,
                                       lat[0].conv.bias)
            else:
                s = _SegBranchFn.apply(pred_seg["seg_input"], lat[1] if lat else None, lat[0].conv.weight if lat else None,
                                       pre.conv.weight, pre.conv.bias, self.conv_out.conv.weight, self.conv_out.conv.bias, tgt)
        elif "seg_input" in pred_seg:
            s = _SegHeadFused.apply(pred_seg["seg_input"], self.conv_out.conv.weight, self.conv_out.conv.bias, tgt)
        else:
            s = _SegSums.apply(pred_seg["seg_logits"], tgt)
        if s.is_cuda and self.fused_tail:
            both = _SegTail.apply(s, tgt.numel(), self.alpha, self.smooth_nom, self.smooth_denom)
            return {"seg_ce": both[0], "seg_dice": both[1]}
        nvox = float(tgt.numel())
        ce = s[0] / nvox                                          # CrossEntropyLoss mean over voxels
        tp, fp, fn = s[1], s[2], s[3]
        dc = (2 * tp + self.smooth_nom) / (2 * tp + fp + fn + self.smooth_denom)
        return {"seg_ce": self.alpha * ce, "seg_dice": (1 - self.alpha) * (1 - dc)}

    def postprocess_for_inference(self, prediction: Dict[str, Tensor], *args, **kwargs) -> Dict[str, Tensor]:
        if "seg_logits" not in prediction:                        # the decoder left the whole branch to us (BaseRetinaNet.inference_step)
            z = self.logit_difference(prediction["seg_input"])
            p1 = torch.sigmoid(z)                                  # softmax over two logits = (sigmoid(l0 - l1), sigmoid(l1 - l0))
            return {"pred_seg": torch.stack((torch.sigmoid(-z), p1), dim=1)}
        return {"pred_seg": torch.softmax(prediction["seg_logits"].float(), dim=1)}

    @torch.no_grad()
    def logit_difference(self, seg_input: Tensor) -> Tensor:
        """z = l1 - l0 per voxel [N, D, H, W] fp32 of the two segmentation logits from the decoder map the decoder stopped at (level 0 before
        its output convolution / before its lateral / level 1 before the last top-down step: `_nndet_pre_out`, `_nndet_pre_lat`,
        `_nndet_pre_up`), by the composed convolution of the training step's fused branch (`_SegBranchFn`, csrc/segbranch.hip) -- the
        32-channel level-0 maps and the logits never exist. Inference only (round 4): the kernel's loss sums are computed against an
        all-background target and dropped."""
        pre = getattr(seg_input, "_nndet_pre_out", None)
        if pre is None:
            raise L.NndetError("logit_difference: the decoder did not defer its level-0 output convolution")
        lat = getattr(seg_input, "_nndet_pre_lat", None)
        up = getattr(seg_input, "_nndet_pre_up", None) if lat else None
        ref = lat[1] if lat else seg_input                        # a full-resolution map: the target's shape
        rp, _ = phys(ref)
        key = (tuple(rp.shape[:4]), rp.device)
        tgt = self._zero_target.get(key)
        if tgt is None:
            self._zero_target.clear()
            tgt = self._zero_target[key] = torch.zeros(rp.shape[:4], dtype=torch.uint8, device=rp.device)

        class _Ctx:                                               # what _SegBranchFn.forward stores for its backward pass; z is entry 6
            needs_input_grad = ()
            def save_for_backward(self, *t):
                self.saved = t
        ctx = _Ctx()
        if up is not None:
            _SegBranchFn.forward(ctx, seg_input, lat[1], lat[0].conv.weight, pre.conv.weight, pre.conv.bias, self.conv_out.conv.weight,
                                 self.conv_out.conv.bias, tgt, up.conv.weight, up.conv.bias, lat[0].conv.bias)
        else:
            _SegBranchFn.forward(ctx, seg_input, lat[1] if lat else None, lat[0].conv.weight if lat else None, pre.conv.weight, pre.conv.bias,
                                 self.conv_out.conv.weight, self.conv_out.conv.bias, tgt)
        return ctx.saved[6]
