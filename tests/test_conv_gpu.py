"""GPU parity of the convolution / norm / loss kernels against plain PyTorch fp32 on the CPU.

fp32 kernels (exact-fp32 MFMA): relative error <= 2e-5 of the tensor's max (summation order only).
bf16 kernels: the reference is fed the SAME bf16-rounded inputs / weights, so the only differences are the
fp32 accumulation order and the final rounding of the output to bf16: <= 1 bf16 ulp = 2^-8 relative to the
tensor's max; gradients of weights are fp32 sums of bf16 products: 5e-3.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.gpu_util import relerr

pytestmark = pytest.mark.gpu

TOL = {torch.float32: dict(fwd=2e-5, dx=2e-5, dw=5e-5), torch.bfloat16: dict(fwd=6e-3, dx=8e-3, dw=8e-3),
       torch.float16: dict(fwd=8e-4, dx=1e-3, dw=1e-3)}       # fp16: 1 ulp = 2^-11 of the tensor's max (8 x tighter than bf16)

# (cin, cout, kernel, stride, padding, transposed, spatial)
CONVS = [
    ("c32_k3", 32, 32, 3, 1, 1, False, (9, 10, 12)),            # cfg (2,8), ragged tiles
    ("c64_k3", 64, 64, 3, 1, 1, False, (8, 8, 8)),              # cfg (4,4)
    ("c128_k3", 128, 128, 3, 1, 1, False, (5, 5, 6)),           # 4 K-chunks, tiny map
    ("c32to64_s2", 32, 64, 3, 2, 1, False, (10, 9, 12)),        # strided: cfg (4,2); dgrad parity classes, odd dims
    ("c32to32_s2", 32, 32, 3, 2, 1, False, (8, 8, 8)),          # strided, rows % 64 != 0: cfg (2,2)
    ("c64_s221", 64, 64, 3, (2, 2, 1), 1, False, (10, 10, 6)),  # anisotropic stride of the last encoder stage
    ("head_cls", 128, 27, 3, 1, 1, False, (5, 6, 6)),           # padded output channels 27 -> 32
    ("head_reg", 64, 162, 3, 1, 1, False, (5, 6, 6)),           # 162 -> 192
    ("lateral", 64, 32, 1, 1, 0, False, (6, 7, 8)),             # 1x1x1
    ("seg_out", 32, 2, 1, 1, 0, False, (8, 8, 8)),              # 2 -> 32 padded
    ("up_222", 64, 32, 2, 2, 0, True, (4, 5, 6)),               # ConvTranspose k = s
    ("up_221", 128, 128, (2, 2, 1), (2, 2, 1), 0, True, (3, 3, 6)),
    ("up_222_c32", 64, 32, 2, 2, 0, True, (5, 7, 9)),           # pointwise kernel, odd dims (ragged 16-point tiles), 8 positions
    ("lateral_c64", 64, 64, 1, 1, 0, False, (5, 7, 9)),
    ("lateral_256to128", 256, 128, 1, 1, 0, False, (4, 5, 3)),
    ("stem", 1, 32, 3, 1, 1, False, (9, 10, 12)),
    ("c32_k3_tiles", 32, 32, 3, 1, 1, False, (17, 16, 9)),      # several (8,8,8) tiles of k_ig3 + ragged edges in every axis
    ("c64_k3_tiles", 64, 64, 3, 1, 1, False, (9, 17, 16)),      # several (4,8,8) tiles, 2 K-chunks
    ("c64to128_s2_tiles", 64, 128, 3, 2, 1, False, (21, 19, 35)),   # k_wgrad3s: 6 x 3 x 3 tiles of (2,4,8) lattice points per image, odd dims, 2 x 2 block pairs
    ("c32to64_s2_tiles", 32, 64, 3, 2, 1, False, (21, 19, 35)),     # k_ig3s (forward) + k_wgrad3s: the same tiling for the 32 -> 64 transition
]
SPEC3 = ["c32_k3", "c64_k3", "c128_k3", "head_cls", "head_reg", "c32_k3_tiles", "c64_k3_tiles"]   # 3x3x3 stride 1
STRIDED = ["c32to64_s2", "c32to32_s2", "c64_s221", "up_222", "up_221", "c64to128_s2_tiles", "c32to64_s2_tiles"]                             # strided gathers (fwd or dgrad)
POINTWISE = ["lateral", "seg_out", "up_222", "up_221", "up_222_c32", "lateral_c64", "lateral_256to128"]   # k_pw's layers (1x1x1, k = s transposed)


def _mk(name, dtype, norm=None, act=False):
    from nndetection_amd.arch.conv import ConvInstanceRelu, ConvGroupRelu
    cfg = {c[0]: c for c in CONVS}[name]
    _, cin, cout, k, s, p, tr, sp = cfg
    cls = ConvGroupRelu if norm == "group" else ConvInstanceRelu
    import zlib
    torch.manual_seed(zlib.crc32(name.encode()) % 1000)      # (hash(str) is randomised per process: runs were not reproducible)
    m = cls(3, cin, cout, k, stride=s, padding=p, transposed=tr, add_norm=norm is not None, add_act=act)
    with torch.no_grad():
        for pname, prm in m.named_parameters():
            if prm.ndim == 5:
                prm.copy_(torch.randn_like(prm) * (1.0 / (prm[0].numel() ** 0.5)))
            else:
                prm.copy_(torch.randn_like(prm) * 0.3 + (1.0 if pname.endswith("norm.weight") else 0.0))
    x = torch.randn(2, cin, *sp)
    return m, x, cfg


def _ref_forward(m, x, cfg, dtype, norm, act):
    """plain torch fp32 on CPU with the same (rounded) operands"""
    _, cin, cout, k, s, p, tr, sp = cfg
    rd = (lambda t: t.detach().to(dtype).float().clone())
    w = rd(m.conv.weight.detach()).requires_grad_(True)
    b = m.conv.bias.detach().clone().requires_grad_(True) if m.conv.bias is not None else None
    xr = rd(x).requires_grad_(True)
    y = F.conv_transpose3d(xr, w, b, stride=s) if tr else F.conv3d(xr, w, b, stride=s, padding=p)
    g = be = None
    if norm is not None:
        if dtype != torch.float32:
            y = y + (y.to(dtype).float() - y).detach()          # the kernel normalises the bf16-rounded conv output
        g = m.norm.weight.detach().clone().requires_grad_(True)
        be = m.norm.bias.detach().clone().requires_grad_(True)
        if norm == "instance":
            y = F.instance_norm(y, weight=g, bias=be, eps=1e-5)
        else:
            y = F.group_norm(y, cout // 16, g, be, eps=1e-5)
    if act:
        y = F.relu(y)
    return xr, w, b, g, be, y


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("name,spec", [(c[0], "lib") for c in CONVS] + [(n, v) for v in ("generic", "ig3-nt8", "ig3-nt16") for n in SPEC3] +
                         [(n, v) for v in ("strided-0", "strided-1", "strided-2") for n in STRIDED] + [(n, "pw") for n in POINTWISE])
def test_conv_fwd_bwd(name, spec, dtype, monkeypatch):
    """spec: "lib" = the library's own kernel choice; "generic" = k_igemm only (NNDET_IGEMM_SPEC=0); "ig3-nt8" / "ig3-nt16" =
    force the compile-time-tile kernel k_ig3 for every 3x3x3 stride-1 convolution (forward and backward-data,
    NNDET_IGEMM_SPEC=2) with 8 / 16 point tiles per wave for the 64-row layers (NNDET_IGEMM_NT); "pw" = the pointwise kernel k_pw
    whatever the size (NNDET_PW_MINPTS=0: by default layers this small go to the implicit-GEMM kernel)."""
    if spec == "pw":
        monkeypatch.setenv("NNDET_PW_MINPTS", "0")
    elif spec.startswith("strided"):                      # tile variants of the strided implicit-GEMM configurations (NNDET_IGEMM_STRIDED)
        monkeypatch.setenv("NNDET_IGEMM_STRIDED", spec[8:])
    elif spec == "generic":
        monkeypatch.setenv("NNDET_IGEMM_SPEC", "0")
        monkeypatch.setenv("NNDET_WGRAD_SPEC", "0")
    elif spec.startswith("ig3"):
        monkeypatch.setenv("NNDET_IGEMM_SPEC", "2")
        monkeypatch.setenv("NNDET_WGRAD_SPEC", "2")      # k_wgrad3 (compile-time tile) for the weight gradient
        monkeypatch.setenv("NNDET_IGEMM_NT", spec[6:])
    m, x, cfg = _mk(name, dtype)
    tol = TOL[dtype]
    xr, w, b, _, _, yref = _ref_forward(m, x, cfg, dtype, None, False)
    gy = torch.randn_like(yref)
    if dtype != torch.float32:
        gy = gy.to(dtype).float()
    yref.backward(gy)
    m = m.cuda()
    xg = x.detach().clone().cuda().to(dtype).requires_grad_(cfg[1] != 1)
    y = m(xg)
    assert y.shape == yref.shape and y.dtype == dtype
    e = relerr(y.float(), yref)
    assert e <= tol["fwd"], f"forward rel err {e:.3e}"
    y.backward(gy.cuda().to(dtype))
    torch.cuda.synchronize()
    if cfg[1] != 1:
        e = relerr(xg.grad.float(), xr.grad)
        assert e <= tol["dx"], f"dgrad rel err {e:.3e}"
    e = relerr(m.conv.weight.grad, w.grad)
    assert e <= tol["dw"], f"wgrad rel err {e:.3e}"
    if b is not None:
        e = relerr(m.conv.bias.grad, b.grad)
        assert e <= tol["dw"], f"bias grad rel err {e:.3e}"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("wgs", ["2", "6", "256"], ids=["wgs2", "wgs6", "wgs256"])
@pytest.mark.parametrize("form", ["2", "1"], ids=["wgrad3e", "wgrad3d"])
@pytest.mark.parametrize("name", ["c64to128_s2_tiles", "c32to64_s2", "c32to64_s2_tiles", "c32_k3_tiles", "c64_k3_tiles", "head_reg"])
def test_lds_dma_weight_gradient_kernels_tile_walk(name, form, wgs, dtype, monkeypatch):
    """k_wgrad3d / k_wgrad3s (csrc/conv_wgrad.hip): persistent workgroups that stage the NEXT tile by LDS-DMA while the current one is
    in the MFMAs. The tile walk (k_wgrad3s: a mixed-radix increment by the grid size, a full decode for the ragged last round;
    k_wgrad3d: magic-multiplier decode) is exercised with grids of 1-2 workgroups (many rounds per workgroup, carries in every digit),
    3-6 (ragged last rounds) and the default; the reference is fp32 torch on the same rounded operands."""
    if form == "1" and "_s2" in name:
        pytest.skip("the stride-2 kernel has one form")
    monkeypatch.setenv("NNDET_WGRAD3D", form)             # 2: k_wgrad3e (round 6: role-split waves, incremental tile walk), 1: k_wgrad3d
    monkeypatch.setenv("NNDET_WGRAD3S_WGS", wgs)
    monkeypatch.setenv("NNDET_WGRAD3D_WGS", wgs)
    monkeypatch.setenv("NNDET_IG3S_WGS", wgs)             # k_ig3s (forward of the 32 -> 64 transition) walks its tiles the same way
    m, x, cfg = _mk(name, dtype)
    tol = TOL[dtype]
    xr, w, b, _, _, yref = _ref_forward(m, x, cfg, dtype, None, False)
    gy = torch.randn_like(yref).to(dtype).float()
    yref.backward(gy)
    m = m.cuda()
    xg = x.detach().clone().cuda().to(dtype).requires_grad_(True)
    y = m(xg)
    e = relerr(y.float(), yref)
    assert e <= tol["fwd"], f"forward rel err {e:.3e}"
    y.backward(gy.cuda().to(dtype))
    torch.cuda.synchronize()
    e = relerr(m.conv.weight.grad, w.grad)
    assert e <= tol["dw"], f"wgrad rel err {e:.3e}"
    if b is not None:
        e = relerr(m.conv.bias.grad, b.grad)
        assert e <= tol["dw"], f"bias grad rel err {e:.3e}"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("name,norm", [("c32_k3", "instance"), ("c64_k3", "instance"), ("c32to64_s2", "instance"), ("c32to64_s2_tiles", "instance"), ("c64to128_s2_tiles", "instance"),
                                        ("stem", "instance"), ("c128_k3", "group"), ("c64_k3", "group")])
def test_conv_norm_relu_block(name, norm, dtype):
    m, x, cfg = _mk(name, dtype, norm, True)
    xr, w, b, g, be, yref = _ref_forward(m, x, cfg, dtype, norm, True)
    gy = torch.randn_like(yref)
    if dtype != torch.float32:
        gy = gy.to(dtype).float()
    yref.backward(gy)
    m = m.cuda()
    xg = x.detach().clone().cuda().to(dtype).requires_grad_(cfg[1] != 1)
    y = m(xg)
    ft = {torch.float32: 2e-5, torch.bfloat16: 1.2e-2, torch.float16: 1.5e-3}[dtype]
    # bf16: the tiny test volumes make sums like dbeta = sum(g) cancel to ~sqrt(n), one flipped ReLU-mask element of the
    # bf16-rounded conv output is worth ~3 % of that sum -> 6e-2 (the fp32 path pins the arithmetic at 1e-4)
    gt = {torch.float32: 1e-4, torch.bfloat16: 1e-1, torch.float16: 2e-2}[dtype]
    e = relerr(y.float(), yref)
    assert e <= ft, f"block forward rel err {e:.3e}"
    y.backward(gy.cuda().to(dtype))
    torch.cuda.synchronize()
    errs = {"dw": relerr(m.conv.weight.grad, w.grad), "dgamma": relerr(m.norm.weight.grad, g.grad),
            "dbeta": relerr(m.norm.bias.grad, be.grad)}
    if cfg[1] != 1:
        errs["dx"] = relerr(xg.grad.float(), xr.grad)
    bad = {k: v for k, v in errs.items() if v > gt}
    assert not bad, f"block backward rel errs {errs}"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("shape,cout", [((3, 13, 17, 22), 32), ((2, 8, 16, 16), 64), ((1, 32, 40, 24), 32)], ids=["ragged", "c64", "tiles"])
def test_fused_stem_block_matches_separate_kernels_and_fp32(shape, cout, dtype, monkeypatch):
    """nndet_stem_block_forward / _backward (conv(1 -> C) + InstanceNorm + ReLU with the convolution recomputed, backward as ONE
    pass + combine) against (a) the separate kernels conv -> statistics -> norm apply / norm backward -> stem weight gradient on the
    same 16-bit inputs and (b) plain PyTorch fp32 on the CPU with the same rounded operands. Ragged volumes (tiles that stick out
    in every axis), several images, 64 output channels (two channel blocks per workgroup column)."""
    from nndetection_amd.arch import conv as conv_mod
    from nndetection_amd.arch.conv import ConvInstanceRelu
    torch.manual_seed(7)
    N, D, H, W = shape
    m = ConvInstanceRelu(3, 1, cout, 3, stride=1, padding=1)
    with torch.no_grad():
        m.conv.weight.copy_(torch.randn_like(m.conv.weight) * 0.25)
        m.norm.weight.copy_(1.0 + 0.3 * torch.randn_like(m.norm.weight))
        m.norm.bias.copy_(0.3 * torch.randn_like(m.norm.bias))
    x = torch.randn(N, 1, D, H, W)
    gy = torch.randn(N, cout, D, H, W).to(dtype).float()
    # (b) fp32 reference with the rounded operands
    rd = lambda t_: t_.detach().to(dtype).float().clone()
    xr = rd(x)
    w = rd(m.conv.weight).requires_grad_(True)
    g = m.norm.weight.detach().clone().requires_grad_(True); be = m.norm.bias.detach().clone().requires_grad_(True)
    yref = F.relu(F.instance_norm(F.conv3d(xr, w, None, padding=1), weight=g, bias=be, eps=1e-5))
    yref.backward(gy)
    res = {}
    for fused in (True, False):
        monkeypatch.setattr(conv_mod, "FUSED_STEM", fused)
        mg = ConvInstanceRelu(3, 1, cout, 3, stride=1, padding=1)
        mg.load_state_dict(m.state_dict())
        mg = mg.cuda()
        y = mg(x.cuda().to(dtype))
        assert y.dtype == dtype and y.shape == yref.shape
        y.backward(gy.cuda().to(dtype))
        torch.cuda.synchronize()
        res[fused] = (y.float().cpu(), mg.conv.weight.grad.cpu(), mg.norm.weight.grad.cpu(), mg.norm.bias.grad.cpu())
    ft, gt = (1.2e-2, 4e-2) if dtype == torch.bfloat16 else (1.5e-3, 6e-3)
    for name, i, ref in (("forward", 0, yref.detach()), ("dW", 1, w.grad), ("dgamma", 2, g.grad), ("dbeta", 3, be.grad)):
        tol = ft if i == 0 else gt
        e_ref, e_sep, e_sepref = relerr(res[True][i], ref), relerr(res[True][i], res[False][i]), relerr(res[False][i], ref)
        assert e_ref <= tol, f"{name}: fused vs fp32 reference {e_ref:.3e}"
        # the two HIP routes agree as far as their distances from the fp32 result allow (the sums over these small volumes cancel,
        # which amplifies the 16-bit rounding of the separate kernels' stored norm-backward output: measured up to 2.6e-2 for dW)
        assert e_sep <= 1.05 * (e_ref + e_sepref) + 1e-6, f"{name}: fused vs separate kernels {e_sep:.3e} ({e_ref:.3e} + {e_sepref:.3e})"
        # the fused path keeps the pre-norm values and the norm-backward output in fp32: it must not be further from fp32 than the
        # separate kernels are (which round both to 16 bits), up to noise
        assert e_ref <= 1.5 * e_sepref + 0.25 * tol, f"{name}: fused {e_ref:.3e} vs separate {e_sepref:.3e} from fp32"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("shape", [(2, 9, 10, 12), (1, 16, 24, 32)], ids=["ragged", "tiles"])
def test_rank1_segmentation_gradient_through_the_producer_conv(shape, dtype, monkeypatch):
    """decoder.out.P0 (3x3x3, 32 -> 32, bias) -> fused segmentation head + loss. The head's input gradient is d1 (x) (w1 - w0); with
    NNDET_SEG_RANK1 the producer convolution's backward pass consumes the FACTORS (data gradient = one-input-channel convolution of
    d1, weight gradient = (w1 - w0) (x) one-channel weight gradient) instead of the dense 32-channel tensor. Same gradients for the
    conv input, conv weight / bias and the head's weight / bias as the dense route, and as plain PyTorch fp32 on the CPU."""
    from nndetection_amd.arch import conv as conv_mod
    from nndetection_amd.arch import Generator, ConvInstanceRelu, DiCESegmenterFgBg
    torch.manual_seed(11)
    N, D, H, W = shape
    conv = ConvInstanceRelu(3, 32, 32, 3, stride=1, padding=1, add_norm=False, add_act=False)
    seg = DiCESegmenterFgBg(Generator(ConvInstanceRelu, 3), seg_classes=1, in_channels=[32], decoder_levels=[0],
                            dice_kwargs={"batch_dice": True})
    with torch.no_grad():
        conv.conv.weight.copy_(torch.randn_like(conv.conv.weight) / 29.4); conv.conv.bias.copy_(torch.randn(32) * 0.2)
        seg.conv_out.conv.weight.copy_(torch.randn_like(seg.conv_out.conv.weight) * 0.3); seg.conv_out.conv.bias.copy_(torch.tensor([0.2, -0.1]))
    x0 = torch.randn(N, 32, D, H, W)
    tgt = (torch.rand(N, D, H, W) > 0.75).float()
    rd = lambda t_: t_.detach().to(dtype).float().clone()
    # fp32 CPU reference on the rounded operands (the conv output and the logits are rounded to the 16-bit type like the kernels store them)
    xr = rd(x0).requires_grad_(True)
    w = rd(conv.conv.weight).requires_grad_(True); b = conv.conv.bias.detach().clone().requires_grad_(True)
    ws = rd(seg.conv_out.conv.weight).requires_grad_(True); bs = seg.conv_out.conv.bias.detach().clone().requires_grad_(True)
    o = F.conv3d(xr, w, b, padding=1)
    o = o + (o.to(dtype).float() - o).detach()
    sl = F.conv3d(o, ws, bs)
    sl = sl + (sl.to(dtype).float() - sl).detach()
    t = (tgt > 0).long()
    p = torch.softmax(sl, 1)
    oh = torch.zeros_like(p).scatter_(1, t[:, None], 1)
    ax = [0, 2, 3, 4]
    tp = (p * oh).sum(ax); fp = (p * (1 - oh)).sum(ax); fn = ((1 - p) * oh).sum(ax)
    loss = 0.5 * F.cross_entropy(sl, t) + 0.5 * (1 - ((2 * tp + 1e-5) / (2 * tp + fp + fn + 1e-5))[1:].mean())
    (loss * 64.0).backward()                                # (a loss scale, as a GradScaler would apply for fp16)
    res = {}
    for rank1 in (True, False):
        monkeypatch.setattr(conv_mod, "RANK1", rank1)
        cg, sg = ConvInstanceRelu(3, 32, 32, 3, stride=1, padding=1, add_norm=False, add_act=False), \
            DiCESegmenterFgBg(Generator(ConvInstanceRelu, 3), seg_classes=1, in_channels=[32], decoder_levels=[0], dice_kwargs={"batch_dice": True})
        cg.load_state_dict(conv.state_dict()); sg.load_state_dict(seg.state_dict())
        cg, sg = cg.cuda(), sg.cuda()
        xg = x0.cuda().to(dtype).requires_grad_(True)
        og = cg(xg)
        og._nndet_rank1_ok = True                           # what BaseRetinaNet.forward asserts for decoder level 0
        out = sg.compute_loss(sg([og], fused=True), tgt.cuda())
        ((out["seg_ce"] + out["seg_dice"]) * 64.0).backward()
        torch.cuda.synchronize()
        assert (len(conv_mod._rank1_grads) == 0)             # consumed (or never registered)
        res[rank1] = [xg.grad.float().cpu(), cg.conv.weight.grad.cpu(), cg.conv.bias.grad.cpu(), sg.conv_out.conv.weight.grad.cpu(),
                      sg.conv_out.conv.bias.grad.cpu()]
    refs = [xr.grad, w.grad, b.grad, ws.grad, bs.grad]
    tol = 2.5e-2 if dtype == torch.bfloat16 else 4e-3
    for name, a, d_, r in zip(("dx", "dW", "db", "dW_seg", "db_seg"), res[True], res[False], refs):
        e_ref, e_dense, e_denseref = relerr(a, r), relerr(a, d_), relerr(d_, r)
        assert e_ref <= tol, f"{name}: rank-1 route vs fp32 {e_ref:.3e}"
        assert e_dense <= 1.05 * (e_ref + e_denseref) + 1e-6, f"{name}: rank-1 vs dense route {e_dense:.3e}"
        assert e_ref <= 1.5 * e_denseref + 0.25 * tol, f"{name}: rank-1 {e_ref:.3e} vs dense {e_denseref:.3e} from fp32"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("case", [(320, 320, 1, (2, 10, 10, 6)), (128, 256, 2, (2, 20, 20, 12)), (64, 96, 1, (1, 5, 5, 6)), (256, 320, (2, 2, 1), (2, 10, 10, 6))],
                         ids=["e4", "e3s2", "odd-rows", "s221"])
def test_split_k_matches_unsplit(case, dtype, monkeypatch):
    """Split-K launches of the small / deep layers (k_igemm + k_ig_splitk_reduce, nndet_conv3d_forward_ws / _backward_data_ws): same
    outputs, norm statistics (through the normalised block output) and data gradients as the unsplit launch -- fp32 summation order
    only -- for automatic and forced split counts; and the split path really is the one that runs."""
    from nndetection_amd.arch import ConvInstanceRelu
    from nndetection_amd import _lib as L
    cin, cout, stride, shape = case
    torch.manual_seed(5)
    N, D, H, W = shape
    x0, r0 = torch.randn(N, cin, D, H, W), torch.randn(N, cout, D, H, W)
    blk = ConvInstanceRelu(3, cin, cout, 3, stride=stride, padding=1)                        # conv -> IN -> ReLU (epilogue statistics)
    plain = ConvInstanceRelu(3, cin, cout, 3, stride=1, padding=1, add_norm=False, add_act=False)   # conv + bias (+ residual)
    calls = []
    real = L.call
    monkeypatch.setattr(L, "call", lambda name, *a: (calls.append(name), real(name, *a))[1])
    res = {}
    for mode in ("0", None, "2", "5"):
        if mode is None:
            monkeypatch.delenv("NNDET_IGEMM_SPLITK", raising=False)
        else:
            monkeypatch.setenv("NNDET_IGEMM_SPLITK", mode)
        b, p = ConvInstanceRelu(3, cin, cout, 3, stride=stride, padding=1), ConvInstanceRelu(3, cin, cout, 3, stride=1, padding=1, add_norm=False, add_act=False)
        b.load_state_dict(blk.state_dict()); p.load_state_dict(plain.state_dict())
        b, p = b.cuda(), p.cuda()
        xg = x0.cuda().to(dtype).requires_grad_(True)
        calls.clear()
        y = b(xg)
        z = p(xg, residual=r0.cuda().to(dtype))
        (y.float().square().sum() * 1e-2 + (z.float() * 0.37).sum()).backward()
        torch.cuda.synchronize()
        used = "nndet_conv3d_forward_ws" in calls
        if mode is None:                         # automatic: small grids split when there are >= 4 channel chunks (32 / 16 channels each)
            assert used == (dtype != torch.float32 and cin // 32 >= 4), (case, calls)      # (fp32 = the parity path: never automatic)
        else:
            assert used == (mode != "0") and ("nndet_conv3d_backward_data_ws" in calls) == (mode != "0"), (mode, calls)
        res[mode] = [t.detach().float().cpu() for t in (y, z, xg.grad, b.conv.weight.grad, b.norm.weight.grad, p.conv.bias.grad)]
    tol = 2e-5 if dtype == torch.float32 else (1.6e-2 if dtype == torch.bfloat16 else 2e-3)     # 16-bit: one ulp of the stored output
    for mode in (None, "2", "5"):
        for name, a, r0 in zip(("y", "z", "dx", "dW", "dgamma", "db"), res[mode], res["0"]):
            assert relerr(a, r0) <= tol, (mode, name, relerr(a, r0))


@pytest.mark.parametrize("lateral", [False, True], ids=["out", "out+lateral"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("shape", [(2, 9, 10, 12), (1, 16, 24, 32), (1, 5, 17, 35)], ids=["ragged", "tiles", "odd"])
def test_segmentation_branch_as_one_convolution(shape, dtype, lateral, monkeypatch):
    """csrc/segbranch.hip: decoder.out.P0 (3x3x3, 32 -> 32, bias) + the segmenter's 1x1x1 output conv + CE / SoftDice as ONE composed
    32 -> 1 convolution (`_SegBranchFn`): losses and the gradients of the input, both weights and both biases against plain PyTorch
    fp32 of the two layers (16-bit input, fp32 weights -- the fused route never rounds the 32-channel map or the logits, so it must be
    at least as close to fp32 as the two-layer HIP route, which does) and against that two-layer route (`_SegHeadFused` + rank-1).
    lateral: the level-0 map is x + W_lat a0 with a 1x1x1 lateral absorbed into the same kernel (nndet_segbranch_forward2): gradients
    of a0 and W_lat as well."""
    from nndetection_amd.arch import conv as conv_mod, segmenter as seg_mod
    from nndetection_amd.arch import Generator, ConvInstanceRelu, DiCESegmenterFgBg
    from nndetection_amd import _lib as L
    torch.manual_seed(13)
    N, D, H, W = shape
    mk = lambda: (ConvInstanceRelu(3, 32, 32, 3, stride=1, padding=1, add_norm=False, add_act=False),
                  DiCESegmenterFgBg(Generator(ConvInstanceRelu, 3), seg_classes=1, in_channels=[32], decoder_levels=[0], dice_kwargs={"batch_dice": True}),
                  ConvInstanceRelu(3, 32, 32, 1, stride=1, padding=0, add_norm=False, add_act=False))
    conv, seg, latm = mk()
    with torch.no_grad():
        conv.conv.weight.copy_(torch.randn_like(conv.conv.weight) / 29.4); conv.conv.bias.copy_(torch.randn(32) * 0.2)
        seg.conv_out.conv.weight.copy_(torch.randn_like(seg.conv_out.conv.weight) * 0.3); seg.conv_out.conv.bias.copy_(torch.tensor([0.2, -0.1]))
        latm.conv.weight.copy_(torch.randn_like(latm.conv.weight) / 5.7); latm.conv.bias.zero_()
    x0, a00 = torch.randn(N, 32, D, H, W), torch.randn(N, 32, D, H, W).relu()
    tgt = (torch.rand(N, D, H, W) > 0.75).float()
    xr = x0.to(dtype).float().requires_grad_(True)
    ar = a00.to(dtype).float().requires_grad_(True)
    w = conv.conv.weight.detach().clone().requires_grad_(True); b = conv.conv.bias.detach().clone().requires_grad_(True)
    ws = seg.conv_out.conv.weight.detach().clone().requires_grad_(True); bs = seg.conv_out.conv.bias.detach().clone().requires_grad_(True)
    wl = latm.conv.weight.detach().clone().requires_grad_(True)
    sl = F.conv3d(F.conv3d(xr + F.conv3d(ar, wl) if lateral else xr, w, b, padding=1), ws, bs)
    t = (tgt > 0).long()
    p = torch.softmax(sl, 1)
    oh = torch.zeros_like(p).scatter_(1, t[:, None], 1)
    ax = [0, 2, 3, 4]
    tp = (p * oh).sum(ax); fp = (p * (1 - oh)).sum(ax); fn = ((1 - p) * oh).sum(ax)
    ce = 0.5 * F.cross_entropy(sl, t)
    dice = 0.5 * (1 - ((2 * tp + 1e-5) / (2 * tp + fp + fn + 1e-5))[1:].mean())
    ((ce + dice) * 64.0).backward()
    calls = []
    real = L.call
    monkeypatch.setattr(L, "call", lambda name, *a: (calls.append(name), real(name, *a))[1])
    res = {}
    for fused in (True, False):
        cg, sg, lg = mk()
        cg.load_state_dict(conv.state_dict()); sg.load_state_dict(seg.state_dict()); lg.load_state_dict(latm.state_dict())
        cg, sg, lg = cg.cuda(), sg.cuda(), lg.cuda()
        xg = x0.cuda().to(dtype).requires_grad_(True)
        ag = a00.cuda().to(dtype).requires_grad_(True)
        calls.clear()
        if fused:
            xin = xg * 1.0                                      # (a non-leaf, like the decoder's level-0 map)
            xin._nndet_pre_out = cg                             # what UFPNModular._out0 attaches when the detector defers out.P0
            if lateral:
                xin._nndet_pre_lat = (lg, ag * 1.0)             # ... and UFPNModular._top_down0 when the lateral is absorbed
            out = sg.compute_loss(sg([xin], fused=True), tgt.cuda())
            assert ("nndet_segbranch_forward2" if lateral else "nndet_segbranch_forward") in calls and "nndet_seghead_forward" not in calls
        else:
            og = cg(xg + lg(ag) if lateral else xg)
            og._nndet_rank1_ok = True
            out = sg.compute_loss(sg([og], fused=True), tgt.cuda())
            assert "nndet_seghead_forward" in calls and "nndet_segbranch_forward" not in calls
        ((out["seg_ce"] + out["seg_dice"]) * 64.0).backward()
        torch.cuda.synchronize()
        res[fused] = [out["seg_ce"].detach().cpu(), out["seg_dice"].detach().cpu(), xg.grad.float().cpu(), cg.conv.weight.grad.cpu(),
                      cg.conv.bias.grad.cpu(), sg.conv_out.conv.weight.grad.cpu(), sg.conv_out.conv.bias.grad.cpu()]
        if lateral:
            res[fused] += [ag.grad.float().cpu(), lg.conv.weight.grad.cpu()]
    refs = [ce.detach(), dice.detach(), xr.grad, w.grad, b.grad, ws.grad, bs.grad] + ([ar.grad, wl.grad] if lateral else [])
    ltol = 2e-3 if dtype == torch.bfloat16 else 3e-4
    tol = 2.5e-2 if dtype == torch.bfloat16 else 4e-3
    for i, name in enumerate(("seg_ce", "seg_dice")):
        assert abs(float(res[True][i]) - float(refs[i])) <= ltol, (name, float(res[True][i]), float(refs[i]))
    for name, a, d_, r in zip(("dx", "dW", "db", "dW_seg", "db_seg", "da0", "dW_lat"), res[True][2:], res[False][2:], refs[2:]):
        e_ref, e_two, e_tworef = relerr(a, r), relerr(a, d_), relerr(d_, r)
        assert e_ref <= tol, f"{name}: fused branch vs fp32 {e_ref:.3e}"
        assert e_two <= 1.05 * (e_ref + e_tworef) + 1e-6, f"{name}: fused branch vs two-layer route {e_two:.3e}"
        assert e_ref <= 1.25 * e_tworef + 0.25 * tol, f"{name}: fused {e_ref:.3e} vs two-layer {e_tworef:.3e} from fp32"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("shape", [(2, 8, 10, 12), (1, 16, 24, 32), (1, 6, 18, 34)], ids=["ragged", "tiles", "odd-tiles"])
def test_segmentation_branch_absorbs_the_top_down_step(shape, dtype, monkeypatch):
    """NNDET_SEG_UP: with the lateral absorbed, the last top-down step up.P1 (ConvTranspose3d k = s = 2, 64 -> 32) goes into the branch as
    well -- conv3(up(x1) + b; wc) as ONE half-resolution 3x3x3 convolution 64 -> 8 (nndet_conv3d_forward on composed weights) + a
    border-class bias (nndet_segbranch_forward_up); backward through nndet_segbranch_s2d and that convolution's data / weight gradient.
    Losses and the gradients of x1, a0 and of ALL parameters (up.P1 weight + bias, lateral weight + bias, out.P0, head) against plain
    PyTorch fp32 of the four layers and against the route that forms the top-down term (nndet_segbranch_forward2)."""
    from nndetection_amd.arch import Generator, ConvInstanceRelu, DiCESegmenterFgBg
    from nndetection_amd import _lib as L
    torch.manual_seed(17)
    N, D, H, W = shape
    mk = lambda: (ConvInstanceRelu(3, 32, 32, 3, stride=1, padding=1, add_norm=False, add_act=False),
                  DiCESegmenterFgBg(Generator(ConvInstanceRelu, 3), seg_classes=1, in_channels=[32], decoder_levels=[0], dice_kwargs={"batch_dice": True}),
                  ConvInstanceRelu(3, 32, 32, 1, stride=1, padding=0, add_norm=False, add_act=False),
                  ConvInstanceRelu(3, 64, 32, 2, stride=2, padding=0, add_norm=False, add_act=False, transposed=True))
    conv, seg, latm, upm = mk()
    with torch.no_grad():
        conv.conv.weight.copy_(torch.randn_like(conv.conv.weight) / 29.4); conv.conv.bias.copy_(torch.randn(32) * 0.2)
        seg.conv_out.conv.weight.copy_(torch.randn_like(seg.conv_out.conv.weight) * 0.3); seg.conv_out.conv.bias.copy_(torch.tensor([0.2, -0.1]))
        latm.conv.weight.copy_(torch.randn_like(latm.conv.weight) / 5.7); latm.conv.bias.copy_(torch.randn(32) * 0.3)
        upm.conv.weight.copy_(torch.randn_like(upm.conv.weight) / 8.0); upm.conv.bias.copy_(torch.randn(32) * 0.3)
    x10, a00 = torch.randn(N, 64, D // 2, H // 2, W // 2), torch.randn(N, 32, D, H, W).relu()
    tgt = (torch.rand(N, D, H, W) > 0.75).float()
    xr, ar = x10.to(dtype).float().requires_grad_(True), a00.to(dtype).float().requires_grad_(True)
    P = {"w": conv.conv.weight, "b": conv.conv.bias, "ws": seg.conv_out.conv.weight, "bs": seg.conv_out.conv.bias, "wl": latm.conv.weight,
         "bl": latm.conv.bias, "wu": upm.conv.weight, "bu": upm.conv.bias}
    R = {k: v.detach().clone().requires_grad_(True) for k, v in P.items()}
    x0r = F.conv3d(ar, R["wl"], R["bl"]) + F.conv_transpose3d(xr, R["wu"], R["bu"], stride=2)
    sl = F.conv3d(F.conv3d(x0r, R["w"], R["b"], padding=1), R["ws"], R["bs"])
    t = (tgt > 0).long()
    p = torch.softmax(sl, 1)
    oh = torch.zeros_like(p).scatter_(1, t[:, None], 1)
    ax = [0, 2, 3, 4]
    tp = (p * oh).sum(ax); fp = (p * (1 - oh)).sum(ax); fn = ((1 - p) * oh).sum(ax)
    ce = 0.5 * F.cross_entropy(sl, t)
    dice = 0.5 * (1 - ((2 * tp + 1e-5) / (2 * tp + fp + fn + 1e-5))[1:].mean())
    ((ce + dice) * 64.0).backward()
    calls = []
    real = L.call
    monkeypatch.setattr(L, "call", lambda name, *a: (calls.append(name), real(name, *a))[1])
    res = {}
    for up_mode in (True, False):
        cg, sg, lg, ug = mk()
        for m_, r_ in ((cg, conv), (sg, seg), (lg, latm), (ug, upm)):
            m_.load_state_dict(r_.state_dict())
        cg, sg, lg, ug = cg.cuda(), sg.cuda(), lg.cuda(), ug.cuda()
        xg = x10.cuda().to(dtype).requires_grad_(True)
        ag = a00.cuda().to(dtype).requires_grad_(True)
        calls.clear()
        if up_mode:
            xin = xg * 1.0                                       # (a non-leaf, like the decoder's level-1 map)
            xin._nndet_pre_out, xin._nndet_pre_lat, xin._nndet_pre_up = cg, (lg, ag * 1.0), ug      # what UFPNModular attaches
            out = sg.compute_loss(sg([xin], fused=True), tgt.cuda())
            assert "nndet_segbranch_forward_up" in calls and "nndet_segbranch_forward2" not in calls
        else:
            from nndetection_amd.arch.conv import _ConvFn
            u, _ = _ConvFn.apply(xg * 1.0, None, False, ug.conv.weight, ug.conv.bias + lg.conv.bias, ug, None, False)
            u._nndet_pre_out, u._nndet_pre_lat = cg, (lg, ag * 1.0)
            out = sg.compute_loss(sg([u], fused=True), tgt.cuda())
            assert "nndet_segbranch_forward2" in calls and "nndet_segbranch_forward_up" not in calls
        ((out["seg_ce"] + out["seg_dice"]) * 64.0).backward()
        torch.cuda.synchronize()
        if up_mode:
            assert "nndet_segbranch_s2d" in calls
        res[up_mode] = [out["seg_ce"].detach().cpu(), out["seg_dice"].detach().cpu(), xg.grad.float().cpu(), ag.grad.float().cpu(),
                        cg.conv.weight.grad.cpu(), cg.conv.bias.grad.cpu(), sg.conv_out.conv.weight.grad.cpu(), sg.conv_out.conv.bias.grad.cpu(),
                        lg.conv.weight.grad.cpu(), lg.conv.bias.grad.cpu(), ug.conv.weight.grad.cpu(), ug.conv.bias.grad.cpu()]
    refs = [ce.detach(), dice.detach(), xr.grad, ar.grad, R["w"].grad, R["b"].grad, R["ws"].grad, R["bs"].grad, R["wl"].grad, R["bl"].grad,
            R["wu"].grad, R["bu"].grad]
    ltol = 2e-3 if dtype == torch.bfloat16 else 3e-4
    tol = 2.5e-2 if dtype == torch.bfloat16 else 4e-3
    for i, name in enumerate(("seg_ce", "seg_dice")):
        assert abs(float(res[True][i]) - float(refs[i])) <= ltol, (name, float(res[True][i]), float(refs[i]))
    names = ("dx1", "da0", "dW", "db", "dW_seg", "db_seg", "dW_lat", "db_lat", "dW_up", "db_up")
    for name, a, d_, r in zip(names, res[True][2:], res[False][2:], refs[2:]):
        e_ref, e_two, e_tworef = relerr(a, r), relerr(a, d_), relerr(d_, r)
        assert e_ref <= tol, f"{name}: absorbed top-down step vs fp32 {e_ref:.3e} (the route that forms it: {e_tworef:.3e})"
        assert e_two <= 1.05 * (e_ref + e_tworef) + 1e-6, f"{name}: vs the route that forms the top-down term {e_two:.3e}"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_segloss(dtype):
    from nndetection_amd.arch import Generator, ConvInstanceRelu, DiCESegmenterFgBg
    torch.manual_seed(0)
    seg = DiCESegmenterFgBg(Generator(ConvInstanceRelu, 3), seg_classes=1, in_channels=[32, 64], decoder_levels=(1,),
                            dice_kwargs={"batch_dice": True})
    x = torch.randn(2, 32, 8, 9, 10)
    tgt = (torch.rand(2, 8, 9, 10) > 0.8).float() * 3
    rd = lambda t_: t_.detach().to(dtype).float().clone()
    xr = rd(x).requires_grad_(True)
    w = rd(seg.conv_out.conv.weight.detach()).requires_grad_(True)
    b = seg.conv_out.conv.bias.detach().clone().requires_grad_(True)
    sl = F.conv3d(xr, w, b)
    if dtype != torch.float32:
        sl = sl + (sl.to(dtype).float() - sl).detach()
    t = (tgt > 0).long()
    ce = 0.5 * F.cross_entropy(sl, t)
    p = torch.softmax(sl, 1)
    oh = torch.zeros_like(p).scatter_(1, t[:, None], 1)
    ax = [0, 2, 3, 4]
    tp = (p * oh).sum(ax); fp = (p * (1 - oh)).sum(ax); fn = ((1 - p) * oh).sum(ax)
    dice = 0.5 * (1 - ((2 * tp + 1e-5) / (2 * tp + fp + fn + 1e-5))[1:].mean())
    (ce + dice).backward()
    seg = seg.cuda()
    xg = x.detach().clone().cuda().to(dtype).requires_grad_(True)
    out = seg.compute_loss(seg([xg]), tgt.cuda())
    assert abs(out["seg_ce"].item() - ce.item()) < 2e-5 and abs(out["seg_dice"].item() - dice.item()) < 2e-5, (out, ce, dice)
    (out["seg_ce"] + out["seg_dice"]).backward()
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    assert relerr(xg.grad.float(), xr.grad) < tol
    assert relerr(seg.conv_out.conv.weight.grad, w.grad) < tol
    assert relerr(seg.conv_out.conv.bias.grad, b.grad) < tol
    pr = seg.postprocess_for_inference(seg([xg]))["pred_seg"]
    assert torch.allclose(pr.sum(1).cpu(), torch.ones(2, 8, 9, 10), atol=1e-5)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("grid", ["8", "16", "0"], ids=["grid8", "grid16", "gridall"])
@pytest.mark.parametrize("norm", [None, "instance"], ids=["bias", "instnorm"])
def test_ig3r_persistent_32to32(norm, grid, dtype, monkeypatch):
    """k_ig3r (persistent, register-resident weights, LDS-DMA staging; bf16 32 -> 32 with every dim a multiple of 8): forward
    with bias / with the fused InstanceNorm statistics + ReLU, data gradient (mirrored taps), weight gradient of the block.
    NNDET_IG3R=2 forces the kernel for small problems; NNDET_IG3R_GRID=8/16 makes every workgroup walk over 6-12 tiles that
    span both images (tile pipeline, buffer toggling, statistics flush at the image change); grid 0 = one tile per workgroup."""
    monkeypatch.setenv("NNDET_IG3R", "2")
    if grid != "0":
        monkeypatch.setenv("NNDET_IG3R_GRID", grid)
    from nndetection_amd.arch.conv import ConvInstanceRelu
    torch.manual_seed(5)
    m = ConvInstanceRelu(3, 32, 32, 3, stride=1, padding=1, add_norm=norm is not None, add_act=norm is not None)
    with torch.no_grad():
        m.conv.weight.copy_(torch.randn_like(m.conv.weight) / 29.4)
        if m.conv.bias is not None:
            m.conv.bias.copy_(torch.randn_like(m.conv.bias) * 0.3)
        if norm is not None:
            m.norm.weight.copy_(1.0 + 0.3 * torch.randn_like(m.norm.weight))
            m.norm.bias.copy_(0.3 * torch.randn_like(m.norm.bias))
    x = torch.randn(2, 32, 16, 24, 32)
    cfg = ("c32r", 32, 32, 3, 1, 1, False, (16, 24, 32))
    xr, w, b, g, be, yref = _ref_forward(m, x, cfg, dtype, norm, norm is not None)
    gy = torch.randn_like(yref).to(dtype).float()
    yref.backward(gy)
    outs = []
    for ig3r in ("2", "0"):                              # the persistent kernel, then k_ig3 on the same inputs
        monkeypatch.setenv("NNDET_IG3R", ig3r)
        mg = ConvInstanceRelu(3, 32, 32, 3, stride=1, padding=1, add_norm=norm is not None, add_act=norm is not None)
        mg.load_state_dict(m.state_dict())
        mg = mg.cuda()
        xg = x.detach().clone().cuda().to(dtype).requires_grad_(True)
        y = mg(xg)
        y.backward(gy.cuda().to(dtype))
        torch.cuda.synchronize()
        outs.append((y.float().cpu(), xg.grad.float().cpu(), mg.conv.weight.grad.cpu()))
    tol = TOL[dtype]
    y, dx, dw = outs[0]
    assert relerr(y, yref) <= (2e-2 if norm else tol["fwd"])
    assert relerr(dx, xr.grad) <= (2e-2 if norm else tol["dx"])
    assert relerr(dw, w.grad) <= (2e-2 if norm else tol["dw"])
    # against k_ig3: same products, same fp32 accumulation order per output -> identical conv outputs; the statistics are summed in
    # a different order (fp32 partial sums per lane), so the normalised block may differ by an output ulp
    y0, dx0, dw0 = outs[1]
    if norm is None:
        assert torch.equal(y, y0) and torch.equal(dx, dx0)
    else:
        assert relerr(y, y0) <= 8e-3 and relerr(dx, dx0) <= 8e-3


def test_full_size_layer_against_torch_gpu():
    """Full-resolution layer shapes of BASELINE config 2 (too slow for the CPU oracle at batch 4): compare against
    PyTorch's own GPU conv (MIOpen) in fp32 on a 1-patch slice, plus linearity as a size-independent property."""
    from nndetection_amd.arch.conv import ConvInstanceRelu
    torch.manual_seed(1)
    m = ConvInstanceRelu(3, 32, 32, 3, stride=1, padding=1, add_norm=False, add_act=False).cuda()
    x = torch.randn(1, 32, 160, 160, 96, device="cuda")
    y = m(x)
    yref = F.conv3d(x, m.conv.weight, m.conv.bias, padding=1)
    assert relerr(y, yref) < 2e-5
    y2 = m(2.5 * x)
    b = m.conv.bias.view(1, -1, 1, 1, 1)
    assert relerr(y2 - b, 2.5 * (y - b)) < 1e-5


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("route", ["lib", "pw"])
def test_transposed_conv_with_fused_residual(dtype, route, monkeypatch):
    """decoder top-down step x_l = lateral_l + up(x_{l+1}) (nndet/arch/decoder/base.py:405-413) from one kernel ("pw": the pointwise
    kernel whatever the size, "lib": the library's choice -- the implicit-GEMM kernel for a layer this small)."""
    if route == "pw":
        monkeypatch.setenv("NNDET_PW_MINPTS", "0")
    m, x, cfg = _mk("up_222", dtype)
    res = torch.randn(2, 32, 8, 10, 12)
    rd = lambda t_: t_.detach().to(dtype).float().clone()
    xr, rr = rd(x).requires_grad_(True), rd(res).requires_grad_(True)
    w = rd(m.conv.weight).requires_grad_(True); b = m.conv.bias.detach().clone().requires_grad_(True)
    yref = F.conv_transpose3d(xr, w, b, stride=2) + rr
    gy = rd(torch.randn_like(yref))
    yref.backward(gy)
    m = m.cuda()
    xg = x.detach().clone().cuda().to(dtype).requires_grad_(True)
    rg = res.detach().clone().cuda().to(dtype).contiguous(memory_format=torch.channels_last_3d).requires_grad_(True)
    y = m(xg, residual=rg)
    tol = {torch.float32: 2e-5, torch.bfloat16: 8e-3, torch.float16: 1e-3}[dtype]
    assert relerr(y.float(), yref) <= tol
    y.backward(gy.cuda().to(dtype))
    assert relerr(xg.grad.float(), xr.grad) <= tol
    assert relerr(rg.grad.float(), rr.grad) <= tol
    assert relerr(m.conv.weight.grad, w.grad) <= tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_fused_seg_head_matches_unfused(dtype):
    """nndet_seghead_forward / _backward (1x1x1 output conv + loss sums in one pass) == conv (nndet_conv3d_forward) followed by
    nndet_segloss_*: the four sums, d(input), dW, dbias. fp32: summation order only; bf16: the logits may differ in the last
    bf16 bit where the fp32 accumulation order of the conv differs."""
    from nndetection_amd.arch.conv import ConvInstanceRelu, Generator
    from nndetection_amd.arch.segmenter import DiCESegmenterFgBg
    torch.manual_seed(3)
    seg = DiCESegmenterFgBg(Generator(ConvInstanceRelu, 3), seg_classes=1, in_channels=[32], decoder_levels=[0],
                            dice_kwargs={"batch_dice": True, "smooth_nom": 1e-5, "smooth_denom": 1e-5}).cuda()
    with torch.no_grad():
        seg.conv_out.conv.weight.copy_(torch.randn_like(seg.conv_out.conv.weight) * 0.3)
        seg.conv_out.conv.bias.copy_(torch.tensor([0.2, -0.1]))
    x0 = (torch.randn(2, 32, 9, 10, 12, device="cuda") * 1.5).to(dtype)
    target = (torch.rand(2, 9, 10, 12, device="cuda") > 0.7).float()
    out = {}
    for fused in (False, True):
        seg.zero_grad(set_to_none=True)
        x = x0.detach().clone().requires_grad_(True)
        pred = seg([x], fused=fused)
        assert ("seg_input" in pred) == fused
        losses = seg.compute_loss(pred, target)
        (losses["seg_ce"] * 1.3 + losses["seg_dice"] * 0.7).backward()
        torch.cuda.synchronize()
        out[fused] = (float(losses["seg_ce"].detach()), float(losses["seg_dice"].detach()), x.grad.float().clone(),
                      seg.conv_out.conv.weight.grad.clone(), seg.conv_out.conv.bias.grad.clone())
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    a, b = out[False], out[True]
    assert abs(a[0] - b[0]) <= tol * max(1.0, abs(a[0])) and abs(a[1] - b[1]) <= tol, (a[:2], b[:2])
    for k, name in ((2, "dx"), (3, "dW"), (4, "dbias")):
        e = relerr(b[k], a[k])
        assert e <= (5e-5 if dtype == torch.float32 else 4e-2), f"{name} rel err {e:.3e}"


# ---------------------------------------------------------------------------------------------- full size (load-dependent hazards)
FULL = [
    # name, cin, cout, k, s, p, transposed, input spatial, env
    ("e0_32x32_ig3r", 32, 32, 3, 1, 1, False, (160, 160, 96), {}),
    ("e0_32x32_ig3", 32, 32, 3, 1, 1, False, (160, 160, 96), {"NNDET_IG3R": "0"}),
    ("e0_32x32_ig3r_ragged", 32, 32, 3, 1, 1, False, (152, 160, 88), {}),          # 4180 tiles on 256 workgroups: 16 or 17 each
    ("e1_32to64_s2", 32, 64, 3, 2, 1, False, (160, 160, 96), {}),
    ("e1_64x64", 64, 64, 3, 1, 1, False, (80, 80, 48), {}),
    ("lat_p0_1x1", 32, 32, 1, 1, 0, False, (160, 160, 96), {}),
    ("up_p1_64to32", 64, 32, 2, 2, 0, True, (80, 80, 48), {}),
]


@pytest.mark.parametrize("lp", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("name", [c[0] for c in FULL])
def test_full_size_bf16_kernels_against_fp32_kernels(name, lp, monkeypatch):
    """The 16-bit (bf16 / fp16) kernels at the benchmarked layer sizes (one 160x160x96 patch) against the exact-fp32 kernels on the same rounded
    inputs and weights, forward and data gradient, every element: <= 1 bf16 ulp of the tensor maximum. Small-size parity cannot
    see hazards that only bite when all 256 CUs saturate the memory pipe (round 2: the last tile of every k_ig3r workgroup had
    4 points x 8 channels overwritten after a 16-byte store; tools/diag_ig3r.py)."""
    from nndetection_amd.arch.conv import ConvInstanceRelu
    _, cin, cout, k, s, p, tr, sp, env = {c[0]: c for c in FULL}[name]
    for kk, vv in env.items():
        monkeypatch.setenv(kk, vv)
    torch.manual_seed(3)
    m = ConvInstanceRelu(3, cin, cout, k, stride=s, padding=p, transposed=tr, add_norm=False, add_act=False).cuda()
    with torch.no_grad():
        m.conv.weight.copy_((torch.randn_like(m.conv.weight) / (m.conv.weight[0].numel() ** 0.5)).to(lp).float())
        m.conv.bias.copy_(torch.randn_like(m.conv.bias) * 0.3)
    x16 = torch.randn(1, cin, *sp, device="cuda").to(lp)
    res = {}
    for dt in (lp, torch.float32):
        x = x16.detach().to(dt).clone().requires_grad_(True)
        y = m(x)
        if dt == lp:
            gy16 = torch.randn(y.shape, device="cuda").to(lp)
        y.backward(gy16.to(dt))
        torch.cuda.synchronize()
        res[dt] = (y.detach().float(), x.grad.float())
        m.zero_grad(set_to_none=True)
    for (a, b), what in zip(zip(res[lp], res[torch.float32]), ("forward", "data gradient")):
        assert torch.isfinite(a).all(), what
        err = (a - b).abs()
        tol = (2.0 ** -7 if lp == torch.bfloat16 else 2.0 ** -10) * float(b.abs().max())
        nbad = int((err > tol).sum())
        assert nbad == 0, f"{name} {what}: {nbad} elements off by more than {tol:.4f} (max {float(err.max()):.4f})"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32], ids=["bf16", "f16", "f32"])
def test_batched_weight_packing_equals_single(dtype):
    """nndet_pack_weights_batched (round 4: tiles transposed through LDS) against nndet_pack_weight (element-wise gather), bit for bit,
    for every kind of layer of the network: 3x3x3 with padded channel counts (27 -> 32, 162 -> 192), 1x1x1, transposed k = s = 2 and
    (2, 2, 1), a 320-channel layer, both orientations (forward / data gradient)."""
    import ctypes
    from nndetection_amd import _lib as L
    from nndetection_amd.arch.conv import ConvInstanceRelu
    specs = [(32, 32, 3, 1, False), (128, 27, 3, 1, False), (128, 162, 3, 1, False), (64, 64, 1, 1, False), (32, 2, 1, 1, False),
             (128, 64, (2, 2, 2), (2, 2, 2), True), (128, 128, (2, 2, 1), (2, 2, 1), True), (320, 320, 3, 1, False), (256, 320, 3, 2, False)]
    mods, convs, modes, ws, outs, refs = [], [], [], [], [], []
    for cin, cout, k, s_, tr in specs:
        m = ConvInstanceRelu(3, cin, cout, k, stride=s_, padding=0 if tr or k == 1 else 1, transposed=tr, add_norm=False, add_act=False).cuda()
        with torch.no_grad():
            m.conv.weight.copy_(torch.randn_like(m.conv.weight))
        for mode in (0, 1):
            d = L.NndetConv()
            d.dtype, d.transposed, d.batch = L._DT[dtype], int(tr), 1
            d.cin, d.cout, d.cin_p, d.cout_p = cin, cout, (cin + 31) // 32 * 32, (cout + 31) // 32 * 32
            d.k = (ctypes.c_int32 * 3)(*m.k); d.s = (ctypes.c_int32 * 3)(*m.s); d.p = (ctypes.c_int32 * 3)(*m.p)
            n = L.load().nndet_packed_weight_elems(ctypes.byref(d), mode)
            w32 = m.conv.weight.detach().float().contiguous()
            ref = torch.full((n,), 7.0, device="cuda").to(dtype)
            L.call("nndet_pack_weight", ctypes.byref(d), mode, L.ptr(w32), L.ptr(ref), L.stream())
            out = torch.full((n,), 9.0, device="cuda").to(dtype)                       # (every element must be written, padding included)
            mods.append(m); convs.append(d); modes.append(mode); ws.append(w32); outs.append(out); refs.append(ref)
    n = len(convs)
    L.call("nndet_pack_weights_batched", (L.NndetConv * n)(*convs), (ctypes.c_int32 * n)(*modes), (ctypes.c_void_p * n)(*[w.data_ptr() for w in ws]),
           (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs]), n, L.stream())
    torch.cuda.synchronize()
    for i, (o, r) in enumerate(zip(outs, refs)):
        assert torch.equal(o, r), (specs[i // 2], modes[i], int((o != r).sum()))


@pytest.mark.parametrize("cin,cout", [(32, 64), (64, 128)])
def test_strided_forward_kernels_rounding_quality(cin, cout, monkeypatch):
    """bound the per-layer error of k_ig3s / k_ig3s2 (round 4) and of the k_igemm route they replaced, instead of only the
    end-to-end trajectory: bf16 outputs against the CORRECTLY ROUNDED float64 convolution of the same bf16 operands (measured with
    tools/ig3s_accuracy.py: 0.009 % / 0.016 % of the outputs differ from it, no bias). Bounds: <= 0.05 % misrounded, each by one bf16 ulp (of the correctly rounded value) at
    most, |mean signed error| <= 2 % of the rms error, epilogue statistics within 1e-5 of float64 sums of the kernel's own outputs."""
    import ctypes
    import torch.nn.functional as F
    from nndetection_amd import _lib as L
    from nndetection_amd.arch.conv import ConvInstanceRelu, _desc, _packed
    torch.manual_seed(1)
    sp, B = (33, 31, 35), 2
    m = ConvInstanceRelu(3, cin, cout, 3, stride=2, padding=1, add_norm=False, add_act=False).cuda()
    x = (torch.randn(B, *sp, cin, device="cuda") * 1.5).to(torch.bfloat16)
    d = _desc(x, cin, cout, m.k, m.s, m.p, False)
    w0 = _packed(m, 0, m.conv.weight, d, torch.bfloat16)
    wq = m.conv.weight.detach().to(torch.bfloat16).double().cpu()
    ref = F.conv3d(x.double().cpu().permute(0, 4, 1, 2, 3), wq, None, stride=2, padding=1).permute(0, 2, 3, 4, 1)
    ref_r = ref.to(torch.bfloat16).double()
    for ig in ("0", "1"):
        monkeypatch.setenv("NNDET_IG3S", ig)
        y = torch.empty((B, d.out_d, d.out_h, d.out_w, cout), dtype=torch.bfloat16, device="cuda")
        stats = torch.zeros((32, B, cout, 2), dtype=torch.float64, device="cuda")
        L.call("nndet_conv3d_forward", ctypes.byref(d), L.ptr(x), L.ptr(w0), None, None, L.ptr(y), L.ptr(stats), L.stream())
        torch.cuda.synchronize()
        yd = y.double().cpu()
        mis = (yd != ref_r)
        frac = float(mis.double().mean())
        assert frac <= 5e-4, (ig, frac)
        ulp = 2.0 ** (torch.floor(torch.log2(ref_r.abs().clamp_min(1e-30))) - 7)         # bf16: 8 significant bits
        # (plus the fp32 accumulation error of the kernels' sums, which matters where the products cancel to a result near zero)
        tol = ulp * 1.0001 + 2e-6 * float(ref.abs().max())
        assert bool(((yd - ref_r).abs() <= tol)[mis].all()), (ig, "an output is off by more than one bf16 ulp",
                                                               float(((yd - ref_r).abs() / tol)[mis].max()))
        err = yd - ref
        assert abs(float(err.mean())) <= 0.02 * float(err.pow(2).mean().sqrt()), (ig, float(err.mean()))
        st = stats.sum(0).cpu()
        s_ref = torch.stack((yd.sum((1, 2, 3)), (yd * yd).sum((1, 2, 3))), -1)
        assert float(((st - s_ref).abs() / s_ref.abs().clamp_min(1e-9)).max()) <= 1e-5, ig


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("cn,relu,shape", [(32, True, (21, 19, 35)), (24, True, (16, 24, 40)), (32, False, (9, 33, 17))])
def test_norm_backward_sums_from_the_strided_data_gradient(cn, relu, shape, dtype, monkeypatch):
    """conv -> InstanceNorm -> ReLU block whose output feeds a 1x1x1 convolution (first consumer in backward) and a stride-2 3x3x3
    convolution (second consumer: k_dgs adds its data gradient in place). With NORM_RED_FUSE the second consumer also accumulates the
    block's norm-backward sums (nndet_conv3d_backward_data_acc_normred) and the block's norm backward skips its reduction pass
    (nndet_norm_backward_presummed). Same rounded gradient values, same expressions: dgamma / dbeta agree to summation order, the
    gradient behind the norm to the rounding of a few elements."""
    from nndetection_amd.arch import conv as CV
    from nndetection_amd.arch.conv import ConvInstanceRelu
    torch.manual_seed(5)
    b0 = ConvInstanceRelu(3, 32, cn, 3, padding=1, add_norm=True, add_act=relu).cuda()
    c1 = ConvInstanceRelu(3, cn, 64, 3, stride=2, padding=1, add_norm=True, add_act=True).cuda()
    lat = ConvInstanceRelu(3, cn, 32, 1, add_norm=False, add_act=False).cuda()
    with torch.no_grad():
        b0.norm.weight.copy_(torch.randn(cn).cuda() * 0.3 + 1.0)
        b0.norm.bias.copy_(torch.randn(cn).cuda() * 0.3)
    x0 = torch.randn(2, 32, *shape, device="cuda").to(dtype)
    g1 = torch.randn(2, 64, *[(s + 1) // 2 for s in shape], device="cuda").to(dtype)
    g2 = torch.randn(2, 32, *shape, device="cuda").to(dtype)
    res = {}
    for mode in (False, True):
        monkeypatch.setattr(CV, "NORM_RED_FUSE", mode)
        CV.norm_red_fused[0] = 0
        for m in (b0, c1, lat):
            m.zero_grad(set_to_none=True)
        x = x0.clone().requires_grad_(True)
        a = b0(x)
        a._nndet_gacc = {"buf": None, "stream": torch.cuda.current_stream()}
        y1 = c1(a)
        y2 = lat(a)
        torch.autograd.backward([y2, y1], [g2, g1])
        torch.cuda.synchronize()
        assert CV.norm_red_fused[0] == (1 if mode else 0)
        res[mode] = [t.detach().float().clone() for t in (b0.norm.weight.grad, b0.norm.bias.grad, b0.conv.weight.grad, x.grad,
                                                          c1.conv.weight.grad, lat.conv.weight.grad)]
    names = ["dgamma", "dbeta", "dw(block)", "dx", "dw(strided)", "dw(lateral)"]
    tol = [2e-5, 2e-5, 2e-3, 2e-3, 1e-6, 1e-6]
    for n, t, a_, b_ in zip(names, tol, res[False], res[True]):
        e = relerr(b_, a_)
        assert e <= t, f"{n}: fused vs separate reduction rel err {e:.3e}"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("batch,shape", [(2, (21, 19, 35)), (1, (40, 32, 24)), (3, (9, 33, 17)), (2, (64, 48, 32))], ids=["odd", "even", "odd3", "tiles"])
def test_persistent_strided_data_gradient_is_bit_identical_to_k_dgs(batch, shape, dtype, monkeypatch):
    """k_dgsp (csrc/conv_dgs.hip, round 6: persistent workgroups, weights in LDS, LDS-DMA halo, two register sets of prefetched residual /
    pre-norm values) against k_dgs (NNDET_DGSP=0) through the three C entry points of the 32 <- 64 stride-2 data gradient -- plain,
    accumulating, accumulating + norm-backward sums: same taps, chunks and accumulation order per output element, so dx must be
    BIT-IDENTICAL (ragged volumes, several images per workgroup walk, grids of 3 workgroups = many tiles and image changes per
    workgroup); the sums agree to fp32 summation order and with a float64 evaluation of the stored gradient."""
    import ctypes
    from nndetection_amd import _lib as L
    from nndetection_amd.arch.conv import ConvInstanceRelu, _desc, _packed
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    m = ConvInstanceRelu(3, 32, 64, 3, stride=2, padding=1, add_norm=False, add_act=False).to(dev)
    x = torch.randn(batch, *shape, 32, device=dev).to(dtype)
    d = _desc(x, 32, 64, m.k, m.s, m.p, False)
    w1 = _packed(m, 1, m.conv.weight, d, dtype)
    dy = torch.randn(batch, d.out_d, d.out_h, d.out_w, 64, device=dev).to(dtype)
    res0 = torch.randn(batch, *shape, 32, device=dev).to(dtype)
    ny = (torch.randn(batch, *shape, 32, device=dev) * 1.5 + 0.3).to(dtype)
    mr = torch.stack((ny.float().mean((1, 2, 3)), 1.0 / (ny.float().var((1, 2, 3), unbiased=False) + 1e-5).sqrt()), -1).contiguous()
    gam, bet = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.3
    st = L.stream()
    out = {}
    for form, wgs in (("0", "256"), ("1", "256"), ("1", "3")):
        monkeypatch.setenv("NNDET_DGSP", form)
        monkeypatch.setenv("NNDET_DGSP_WGS", wgs)
        dx_plain = torch.full_like(x, float("nan"))
        L.call("nndet_conv3d_backward_data", ctypes.byref(d), L.ptr(dy), L.ptr(w1), L.ptr(dx_plain), st)
        dx_acc = res0.clone()
        L.call("nndet_conv3d_backward_data_acc", ctypes.byref(d), L.ptr(dy), L.ptr(w1), L.ptr(dx_acc), None, st)
        dx_nb = res0.clone()
        red = torch.zeros(L.STATS_REPLICAS * batch * 32 * 2 + batch, dtype=torch.float64, device=dev)
        L.call("nndet_conv3d_backward_data_acc_normred", ctypes.byref(d), L.ptr(dy), L.ptr(w1), L.ptr(dx_nb), L.ptr(ny), L.ptr(mr), L.ptr(gam), L.ptr(bet), 1, 32,
               L.ptr(red), st)
        torch.cuda.synchronize()
        out[(form, wgs)] = (dx_plain, dx_acc, dx_nb, red[:L.STATS_REPLICAS * batch * 32 * 2].view(L.STATS_REPLICAS, batch, 32, 2).sum(0))
    ref = out[("0", "256")]
    assert not torch.isnan(ref[0].float()).any()
    for key in (("1", "256"), ("1", "3")):
        got = out[key]
        for i, what in enumerate(("plain", "accumulating", "accumulating + norm sums")):
            assert torch.equal(got[i], ref[i]), f"{key}: dx of the {what} form differs from k_dgs"
        # (S2 = rstd * (sum g y - mean * sum g) cancels: compared against the largest sum of its kind, as the float64 check below)
        assert float(((got[3] - ref[3]).abs() / ref[3].abs().amax((0, 1), keepdim=True)).max()) <= 2e-5, key
    # the sums against float64 from the stored gradient (the definition: S1 = sum g [mask], S2 = sum g [mask] xhat)
    xh = (ny.double() - mr[:, :, 0].double().view(batch, 1, 1, 1, 32)) * mr[:, :, 1].double().view(batch, 1, 1, 1, 32)
    sc = (mr[:, :, 1] * gam).view(batch, 1, 1, 1, 32); sh = (bet - mr[:, :, 0] * (mr[:, :, 1] * gam)).view(batch, 1, 1, 1, 32)
    gm = out[("1", "3")][2].double() * (torch.addcmul(sh, ny.float(), sc) > 0)
    s64 = torch.stack((gm.sum((1, 2, 3)), (gm * xh).sum((1, 2, 3))), -1)
    assert float(((out[("1", "3")][3] - s64).abs() / s64.abs().amax((0, 1), keepdim=True)).max()) <= 1e-6


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("relu", [1, 0], ids=["relu", "linear"])
@pytest.mark.parametrize("batch,shape", [(1, (8, 8, 16)), (3, (13, 19, 37)), (2, (4, 9, 50)), (2, (22, 32, 48))],
                         ids=["one_tile", "ragged", "thin", "tiles"])
def test_stride2_forward_that_writes_its_normalised_input(batch, shape, relu, dtype, monkeypatch):
    """nndet_conv3d_forward_norm_input (k_ig3s<.., PRE>, round 6): the 32 -> 64 stride-2 transition reads the PRE-norm tensor + the
    coefficient table, transforms the halo in LDS and stores the normalised tensor on the way. Against the two launches it replaces
    (nndet_affine_apply, then nndet_conv3d_forward = the plain k_ig3s on the materialised tensor): the normalised tensor and the
    convolution output must be BIT-IDENTICAL (same fmaf / pack / ReLU, same taps and accumulators), every voxel of the normalised
    tensor written exactly by its one owner (NaN canary), the statistics equal to fp64 atomics' order; grids of 1 / 3 / 7 workgroups
    walk many tiles, rounds with and without a successor, and image changes (coefficient reload) per workgroup."""
    import ctypes
    from nndetection_amd import _lib as L
    from nndetection_amd.arch.conv import ConvInstanceRelu, _desc, _packed
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    m = ConvInstanceRelu(3, 32, 64, 3, stride=2, padding=1, add_norm=False, add_act=False, bias=True).to(dev)
    y0 = (torch.randn(batch, *shape, 32, device=dev) * 1.7 + 0.2).to(dtype)              # the pre-norm tensor
    ss = torch.stack((torch.rand(batch, 32, device=dev) + 0.5, torch.randn(batch, 32, device=dev) * 0.4), -1).contiguous()
    d = _desc(y0, 32, 64, m.k, m.s, m.p, False)
    w0 = _packed(m, 0, m.conv.weight, d, dtype)
    bias = m.conv.bias.detach().float().contiguous()
    bias_p = torch.zeros(64, device=dev); bias_p[:64] = bias
    st = L.stream()
    code = L.dtype_code(y0)
    spatial = shape[0] * shape[1] * shape[2]
    # the two launches
    a_ref = torch.empty_like(y0)
    L.call("nndet_affine_apply", code, L.ptr(y0), L.ptr(ss), batch, spatial, 32, relu, L.ptr(a_ref), st)
    out_ref = torch.full((batch, d.out_d, d.out_h, d.out_w, 64), float("nan"), device=dev, dtype=dtype)
    st_ref = torch.zeros(L.STATS_REPLICAS, batch, 64, 2, dtype=torch.float64, device=dev)
    L.call("nndet_conv3d_forward", ctypes.byref(d), L.ptr(a_ref), L.ptr(w0), L.ptr(bias_p), None, L.ptr(out_ref), L.ptr(st_ref), st)
    torch.cuda.synchronize()
    assert not torch.isnan(out_ref.float()).any()
    dp = _desc(y0, 32, 64, m.k, m.s, m.p, False)
    dp.in_affine, dp.in_relu = ss.data_ptr(), relu
    assert L.load().nndet_conv3d_forward_norm_input_fused(ctypes.byref(dp)) == 1
    for wgs in ("256", "1", "3", "7"):
        monkeypatch.setenv("NNDET_IG3S_WGS", wgs)
        for with_stats in (True, False):
            a = torch.full_like(y0, float("nan"))
            out = torch.full_like(out_ref, float("nan"))
            stt = torch.zeros_like(st_ref) if with_stats else None
            L.call("nndet_conv3d_forward_norm_input", ctypes.byref(dp), L.ptr(y0), L.ptr(a), L.ptr(w0), L.ptr(bias_p), L.ptr(out), L.ptr(stt), st)
            torch.cuda.synchronize()
            assert torch.equal(a.view(torch.int16), a_ref.view(torch.int16)), f"wgs {wgs}: the normalised tensor differs"
            assert torch.equal(out.view(torch.int16), out_ref.view(torch.int16)), f"wgs {wgs}: the convolution output differs"
            if with_stats:
                s, r = stt.sum(0), st_ref.sum(0)
                assert float(((s - r).abs() / (r.abs() + 1.0)).max()) <= 2e-6     # (fp32 partial sums per workgroup and image: the walk differs with the grid)
    # a problem the fused launch does not cover runs the two launches behind the same entry point
    m2 = ConvInstanceRelu(3, 32, 32, 3, stride=1, padding=1, add_norm=False, add_act=False).to(dev)
    d2 = _desc(y0, 32, 32, m2.k, m2.s, m2.p, False)
    w2 = _packed(m2, 0, m2.conv.weight, d2, dtype)
    o_ref = torch.empty(batch, *shape, 32, device=dev, dtype=dtype)
    L.call("nndet_conv3d_forward", ctypes.byref(d2), L.ptr(a_ref), L.ptr(w2), None, None, L.ptr(o_ref), None, st)
    d2p = _desc(y0, 32, 32, m2.k, m2.s, m2.p, False)
    d2p.in_affine, d2p.in_relu = ss.data_ptr(), relu
    assert L.load().nndet_conv3d_forward_norm_input_fused(ctypes.byref(d2p)) == 0
    a2, o2 = torch.full_like(y0, float("nan")), torch.full_like(o_ref, float("nan"))
    L.call("nndet_conv3d_forward_norm_input", ctypes.byref(d2p), L.ptr(y0), L.ptr(a2), L.ptr(w2), None, L.ptr(o2), None, st)
    torch.cuda.synchronize()
    assert torch.equal(a2.view(torch.int16), a_ref.view(torch.int16)) and torch.equal(o2.view(torch.int16), o_ref.view(torch.int16))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("cin,cn,relu,batch,shape", [(64, 64, 1, 2, (21, 19, 35)), (64, 48, 1, 3, (8, 24, 40)), (128, 128, 0, 2, (9, 17, 13)), (32, 32, 1, 1, (16, 16, 24)),
                                                      (64, 64, 1, 4, (40, 40, 24))],
                         ids=["c64_ragged", "c48_padded", "c128_linear", "c32_small", "c64_tiles"])
def test_norm_backward_sums_from_the_stride1_data_gradient(cin, cn, relu, batch, shape, dtype, monkeypatch):
    """nndet_conv3d_backward_data_normred (k_ig3<.., NB>, round 6): the stride-1 3x3x3 data gradient also accumulates S1 = sum g [mask],
    S2 = sum g [mask] xhat of the conv -> norm -> ReLU block that produced its input, from the values it stores and that block's pre-norm
    tensor. dx must be BIT-IDENTICAL to nndet_conv3d_backward_data; the sums agree with a float64 evaluation from the stored gradient
    (1e-6 of the largest sum of their kind) and with what k_norm_bwd_reduce takes from the same tensors; through the module route the
    block's norm backward skips its reduction pass and dgamma / dbeta / the gradient behind the norm agree with the separate pass."""
    import ctypes
    from nndetection_amd import _lib as L
    from nndetection_amd.arch import conv as CV
    from nndetection_amd.arch.conv import ConvInstanceRelu, _desc, _packed
    from nndetection_amd.layout import cpad
    dev = torch.device("cuda:0")
    torch.manual_seed(17)
    cp = cpad(cn)
    m = ConvInstanceRelu(3, cn, cin, 3, stride=1, padding=1, add_norm=False, add_act=False).to(dev)     # consumer: cn -> cin channels
    x = torch.randn(batch, *shape, cp, device=dev).to(dtype)
    d = _desc(x, cn, cin, m.k, m.s, m.p, False)
    # (test volumes are small and ragged: take the compile-time-tile kernel k_ig3 whatever the padding / workgroup-count rules of build_plan say)
    monkeypatch.setenv("NNDET_IGEMM_SPEC", "2")
    monkeypatch.setenv("NNDET_IGEMM_SMALLWG", "0")
    assert L.load().nndet_conv3d_dgrad_normred_supported(ctypes.byref(d)) == 1
    w1 = _packed(m, 1, m.conv.weight, d, dtype)
    dy = torch.randn(batch, *shape, d.cout_p, device=dev).to(dtype)
    ny = (torch.randn(batch, *shape, cp, device=dev) * 1.5 + 0.3).to(dtype)
    if cp > cn:
        ny[..., cn:] = 0
    mr = torch.zeros(batch, cp, 2, device=dev)
    mr[:, :cn, 0] = ny.float()[..., :cn].mean((1, 2, 3)); mr[:, :cn, 1] = 1.0 / (ny.float()[..., :cn].var((1, 2, 3), unbiased=False) + 1e-5).sqrt()
    gam, bet = torch.rand(cn, device=dev) + 0.5, torch.randn(cn, device=dev) * 0.3
    st = L.stream()
    dx_ref = torch.full_like(x, float("nan"))
    L.call("nndet_conv3d_backward_data", ctypes.byref(d), L.ptr(dy), L.ptr(w1), L.ptr(dx_ref), st)
    dx = torch.full_like(x, float("nan"))
    red = torch.zeros(L.STATS_REPLICAS * batch * cp * 2 + batch, dtype=torch.float64, device=dev)
    L.call("nndet_conv3d_backward_data_normred", ctypes.byref(d), L.ptr(dy), L.ptr(w1), L.ptr(dx), L.ptr(ny), L.ptr(mr), L.ptr(gam), L.ptr(bet), relu, cn,
           L.ptr(red), st)
    torch.cuda.synchronize()
    assert not torch.isnan(dx_ref.float()).any()
    assert torch.equal(dx.view(torch.int16), dx_ref.view(torch.int16)), "dx differs from the plain data gradient"
    sums = red[:L.STATS_REPLICAS * batch * cp * 2].view(L.STATS_REPLICAS, batch, cp, 2).sum(0)[:, :cn]
    xh = (ny.double()[..., :cn] - mr[:, :cn, 0].double().view(batch, 1, 1, 1, cn)) * mr[:, :cn, 1].double().view(batch, 1, 1, 1, cn)
    sc = (mr[:, :cn, 1] * gam).view(batch, 1, 1, 1, cn); sh = (bet - mr[:, :cn, 0] * (mr[:, :cn, 1] * gam)).view(batch, 1, 1, 1, cn)
    mask = (torch.addcmul(sh, ny.float()[..., :cn], sc) > 0) if relu else torch.ones_like(xh, dtype=torch.bool)
    gm = dx.double()[..., :cn] * mask
    s64 = torch.stack((gm.sum((1, 2, 3)), (gm * xh).sum((1, 2, 3))), -1)
    assert float(((sums - s64).abs() / s64.abs().amax((0, 1), keepdim=True)).max()) <= 1e-6
    if cp > cn:
        assert float(red[:L.STATS_REPLICAS * batch * cp * 2].view(L.STATS_REPLICAS, batch, cp, 2)[:, :, cn:].abs().max()) == 0.0
    # module route: block -> consumer, the block's norm backward with and without the sums from the consumer's data gradient
    b0 = ConvInstanceRelu(3, 32, cn, 3, padding=1, add_norm=True, add_act=bool(relu)).to(dev)
    c1 = ConvInstanceRelu(3, cn, cin, 3, padding=1, add_norm=True, add_act=True).to(dev)
    with torch.no_grad():
        b0.norm.weight.copy_(torch.randn(cn, device=dev) * 0.3 + 1.0); b0.norm.bias.copy_(torch.randn(cn, device=dev) * 0.3)
    x0 = torch.randn(batch, 32, *shape, device=dev).to(dtype)
    g1 = torch.randn(batch, cin, *shape, device=dev).to(dtype)
    res = {}
    for mode in (False, True):
        monkeypatch.setattr(CV, "NORM_RED_CHAIN", mode)
        CV.norm_red_fused[0] = 0
        for mm in (b0, c1):
            mm.zero_grad(set_to_none=True)
        xx = x0.clone().requires_grad_(True)
        c1(b0(xx)).backward(g1)
        torch.cuda.synchronize()
        assert CV.norm_red_fused[0] == (1 if mode else 0)
        res[mode] = [t.detach().float().clone() for t in (b0.norm.weight.grad, b0.norm.bias.grad, b0.conv.weight.grad, xx.grad, c1.conv.weight.grad)]
    for nme, t, a_, b_ in zip(["dgamma", "dbeta", "dw(block)", "dx", "dw(consumer)"], [2e-5, 2e-5, 2e-3, 2e-3, 1e-6], res[False], res[True]):
        e = relerr(b_, a_)
        assert e <= t, f"{nme}: fused vs separate reduction rel err {e:.3e}"


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16], ids=["bf16", "f16"])
@pytest.mark.parametrize("cin1,biases", [(64, (1, 1, 1, 1)), (32, (0, 1, 0, 1)), (48, (1, 0, 1, 0)), (8, (0, 0, 0, 0))])
def test_parameter_composition_of_the_absorbed_branch_in_one_kernel(cin1, biases, dtype, monkeypatch):
    """nndet_segbranch_compose_up (round 6): the parameter-only part of the fused segmentation branch -- composed kernels, constant, summed
    biases, the composed half-resolution kernel over the parity classes, the border-class bias, the two packed weight copies -- from ONE
    kernel against the torch expressions of arch/segmenter.py (_compose_up_branch with COMPOSE_FUSED off): fp32 results to 2e-6 of each
    tensor's largest element (summation order), the 16-bit ones to one unit in the last place."""
    from nndetection_amd.arch import segmenter as S
    dev = torch.device("cuda:0")
    torch.manual_seed(31 + cin1)
    r = lambda *s: torch.randn(*s, device=dev) * 0.2
    w_lat, w_out, w_head, w_up = r(32, 32, 1, 1, 1), r(32, 32, 3, 3, 3), r(2, 32, 1, 1, 1), r(cin1, 32, 2, 2, 2)
    b_out, b_head, b_up, b_lat = (r(32) if biases[0] else None, r(2) if biases[1] else None, r(32) if biases[2] else None, r(32) if biases[3] else None)
    params = (w_lat, w_out, b_out, w_head, b_head, w_up, b_up, b_lat)
    assert S._compose_fusable(params, dtype)
    with torch.enable_grad():
        monkeypatch.setattr(S, "COMPOSE_FUSED", False)
        ref = S._compose_up_branch(params, dtype)
        monkeypatch.setattr(S, "COMPOSE_FUSED", True)
        got = S._compose_up_branch(params, dtype)
    torch.cuda.synchronize()
    for k in ("wd", "wc", "c0", "wfa", "bsum", "cb", "w_up32"):
        a, b = got[k].float().reshape(-1), ref[k].float().reshape(-1)
        assert a.shape == b.shape and tuple(got[k].shape) == tuple(ref[k].shape), k
        assert float((a - b).abs().max()) <= 2e-6 * max(float(b.abs().max()), 1e-3), k
    ulp = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10          # (the spacing of the 16-bit type relative to a value at the bottom of its binade)
    for a, b, k in ((got["wqa"], ref["wqa"], "wqa"), (got["pk"][0], ref["pk"][0], "pk0"), (got["pk"][1], ref["pk"][1], "pk1")):
        assert a.shape == b.shape and a.dtype == b.dtype == dtype, k
        d = (a.float() - b.float()).abs()
        assert bool((d <= ulp * b.float().abs() + 2e-6 * float(b.float().abs().max())).all()), (k, float(d.max()))
        assert float((d > 0).float().mean()) < 0.01, k                   # (a rounding flip here and there, not another tensor)
