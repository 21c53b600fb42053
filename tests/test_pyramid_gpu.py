"""GPU parity of the ragged-batch ("items") path -- all pyramid levels through a shared head layer in ONE launch
(nndetection_amd/arch/pyramid.py, include/nndet_amd.h: NndetItems) -- against the per-level launches of the same kernels,
which are themselves checked against plain PyTorch fp32 in tests/test_conv_gpu.py.

Convolution outputs and data gradients: BIT-IDENTICAL (same compile-time tiles, same tap and accumulation order per element).
Norm outputs: the fp64 statistics are accumulated by atomics in a different order -> 1e-6 (fp32) / 1 bf16 ulp.
Weight / bias gradients: summed over the levels inside the kernel instead of by autograd -> fp32 summation order, 2e-5 / 5e-3.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

# level shapes (D, H, W): the luna160 pyramid P2..P5 scaled down, with ragged tiles in every axis and a level smaller than a tile
LEVELS = [(12, 17, 9), (6, 9, 5), (3, 5, 3), (2, 3, 3)]


def _block(cin, cout, norm, seed):
    from nndetection_amd.arch.conv import ConvGroupRelu
    torch.manual_seed(seed)
    m = ConvGroupRelu(3, cin, cout, 3, stride=1, padding=1, add_norm=norm, add_act=norm, bias=None if norm else True)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.ndim == 5:
                p.copy_(torch.randn_like(p) / (p[0].numel() ** 0.5))
            else:
                p.copy_(torch.randn_like(p) * 0.3 + (1.0 if n.endswith("norm.weight") else 0.0))
    return m.cuda()


def _fmaps(cin, dtype, batch, seed, requires_grad=True):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(batch, cin, *sp, generator=g).cuda().to(dtype).requires_grad_(requires_grad) for sp in LEVELS]


def _split(y2d, meta, c):
    from nndetection_amd.layout import logical
    out = []
    for (n, d, h, w), (r0, nr) in zip(meta.level_shapes, meta.level_rows):
        out.append(logical(y2d[r0:r0 + nr].view(n, d, h, w, y2d.shape[1]), c))
    return out


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
@pytest.mark.parametrize("cin,cout", [(128, 128), (128, 27), (64, 162), (32, 32)], ids=["128to128", "cls_out", "reg_out", "32to32"])
def test_items_conv_matches_per_level(dtype, cin, cout):
    """conv + bias: forward and data gradient bit-identical to the per-level launches, dW / dbias summed over the levels."""
    from nndetection_amd.arch import pyramid as P
    from nndetection_amd import _lib as L
    mod = _block(cin, cout, False, 1)
    batch = 3
    fm_a = _fmaps(cin, dtype, batch, 2)
    fm_b = [f.detach().clone().requires_grad_(True) for f in fm_a]
    assert P.supports(fm_a, [mod])
    # per level (forced onto the same compile-time-tile kernels the ragged launch uses)
    import os
    os.environ["NNDET_IGEMM_SPEC"] = "2"; os.environ["NNDET_WGRAD_SPEC"] = "2"; os.environ["NNDET_IGEMM_SMALLWG"] = "0"
    try:
        ys = [mod(f) for f in fm_a]
        gy = [torch.randn(y.shape, generator=torch.Generator().manual_seed(7 + i)).cuda().to(dtype) for i, y in enumerate(ys)]
        torch.autograd.backward(ys, gy)
        ref_dw, ref_db = mod.conv.weight.grad.clone(), mod.conv.bias.grad.clone()
        mod.zero_grad(set_to_none=True)
    finally:
        for k in ("NNDET_IGEMM_SPEC", "NNDET_WGRAD_SPEC", "NNDET_IGEMM_SMALLWG"):
            os.environ.pop(k, None)
    # ragged
    x2d, meta = P.cat_levels(fm_b)
    assert meta.n_items == batch * len(LEVELS) and x2d.shape == (meta.rows, P.cpad(cin))
    y2d = P.items_block(mod, x2d, meta)
    yi = _split(y2d, meta, cout)
    for a, b in zip(ys, yi):
        assert a.shape == b.shape and torch.equal(a.detach(), b.detach())
    # gradient w.r.t. the ragged output, padded channels zero (what the head gather / norm backward produce)
    g2d = torch.zeros_like(y2d)
    for gl, (r0, nr), (n, d, h, w) in zip(gy, meta.level_rows, meta.level_shapes):
        g2d[r0:r0 + nr, :cout] = gl.permute(0, 2, 3, 4, 1).reshape(nr, cout)
    y2d.backward(g2d)
    torch.cuda.synchronize()
    for fa, fb in zip(fm_a, fm_b):
        assert torch.equal(fa.grad, fb.grad)
    tol = 2e-5 if dtype == torch.float32 else 5e-3
    dw, db = mod.conv.weight.grad, mod.conv.bias.grad
    assert float((dw - ref_dw).abs().max()) <= tol * float(ref_dw.abs().max())
    assert float((db - ref_db).abs().max()) <= tol * float(ref_db.abs().max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_items_trunk_with_groupnorm_matches_per_level(dtype):
    """conv -> GroupNorm -> ReLU -> conv -> GroupNorm -> ReLU -> conv (a head branch): outputs, input gradients and parameter
    gradients against the per-level path."""
    from nndetection_amd.arch import pyramid as P
    blocks = [_block(64, 64, True, 11), _block(64, 64, True, 12), _block(64, 27, False, 13)]
    fm_a = _fmaps(64, dtype, 2, 3)
    fm_b = [f.detach().clone().requires_grad_(True) for f in fm_a]

    def run_levels():
        outs = []
        for f in fm_a:
            t = f
            for b in blocks:
                t = b(t)
            outs.append(t)
        return outs

    ys = run_levels()
    gy = [torch.randn(y.shape, generator=torch.Generator().manual_seed(70 + i)).cuda().to(dtype) for i, y in enumerate(ys)]
    torch.autograd.backward(ys, gy)
    ref = {f"{i}.{n}": p.grad.clone() for i, b in enumerate(blocks) for n, p in b.named_parameters()}
    for b in blocks:
        b.zero_grad(set_to_none=True)

    x2d, meta = P.cat_levels(fm_b)
    t = x2d
    for b in blocks:
        t = P.items_block(b, t, meta)
    yi = _split(t, meta, 27)
    ftol = 2e-5 if dtype == torch.float32 else 2 ** -7
    for a, b_ in zip(ys, yi):
        scale = float(a.detach().float().abs().max())
        assert float((a.detach().float() - b_.detach().float()).abs().max()) <= ftol * scale
    g2d = torch.zeros_like(t)
    for gl, (r0, nr) in zip(gy, meta.level_rows):
        g2d[r0:r0 + nr, :27] = gl.permute(0, 2, 3, 4, 1).reshape(nr, 27)
    t.backward(g2d)
    torch.cuda.synchronize()
    gtol = 5e-5 if dtype == torch.float32 else 2e-2
    for fa, fb in zip(fm_a, fm_b):
        scale = float(fa.grad.float().abs().max())
        assert float((fa.grad.float() - fb.grad.float()).abs().max()) <= gtol * scale
    got = {f"{i}.{n}": p.grad for i, b in enumerate(blocks) for n, p in b.named_parameters()}
    for n in ref:
        scale = float(ref[n].abs().max()) + 1e-12
        assert float((got[n] - ref[n]).abs().max()) <= gtol * scale, (n, float((got[n] - ref[n]).abs().max()), scale)


def test_items_rejects_unsupported():
    """Strided / transposed / 1x1x1 blocks and too many items fall back to the per-level path (supports() is False); the C entry
    points return NNDET_EINVAL for them instead of computing something else."""
    import ctypes
    from nndetection_amd.arch import pyramid as P
    from nndetection_amd.arch.conv import ConvGroupRelu
    from nndetection_amd import _lib as L
    fm = _fmaps(32, torch.float32, 2, 5, requires_grad=False)
    assert not P.supports(fm, [ConvGroupRelu(3, 32, 32, 3, stride=2, padding=1).cuda()])
    assert not P.supports(fm, [ConvGroupRelu(3, 32, 32, 1, stride=1, padding=0).cuda()])
    assert not P.supports(_fmaps(32, torch.float32, 9, 5, requires_grad=False), [_block(32, 32, False, 1)])     # 36 items > 32
    mod = ConvGroupRelu(3, 32, 32, 1, stride=1, padding=0).cuda()
    x2d, meta = P.cat_levels(fm)
    d = P._items_desc(x2d, mod, meta)
    d.k = (ctypes.c_int32 * 3)(1, 1, 1); d.p = (ctypes.c_int32 * 3)(0, 0, 0)
    y = torch.empty_like(x2d)
    w = torch.zeros(32 * 32, device="cuda")
    rc = L.load().nndet_conv3d_forward_items(ctypes.byref(d), ctypes.byref(meta.items), x2d.data_ptr(), w.data_ptr(), None,
                                             y.data_ptr(), None, None)
    assert rc == -1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_detection_head_items_equals_per_level(dtype, monkeypatch):
    """DetectionHeadHNMNative.forward with the levels as one ragged batch (default) == the per-level launches: identical logits /
    deltas up to the GroupNorm statistics order, parameter gradients up to the summation order."""
    monkeypatch.setenv("NNDET_IGEMM_SPLITK", "0")     # (the per-level launches of the small levels would split K: another fp32 summation order)
    from nndetection_amd.plans import get_plan, MODEL_CFG_V001
    from nndetection_amd.ptmodule import build_model
    from nndetection_amd.arch.heads import DetectionHeadHNMNative
    plan = get_plan("toy64")
    torch.manual_seed(0)
    net = build_model(plan).cuda()
    head = net.head
    with torch.no_grad():
        for i, sc in enumerate(head.regressor.scales):
            sc.scale.fill_(1.0 + 0.25 * i)
        for m in head.modules():                                    # the default init (std 0.01) gives vanishing activations
            if isinstance(m, torch.nn.Conv3d):
                m.weight.copy_(torch.randn_like(m.weight) / (m.weight[0].numel() ** 0.5))
    c = plan["arch"]["fpn_channels"]
    g = torch.Generator().manual_seed(1)
    shapes = [(16, 16, 16), (8, 8, 8), (4, 4, 4), (2, 2, 2)]
    fm = [torch.randn(2, c, *s, generator=g).cuda().to(dtype) for s in shapes]
    res = {}
    old = DetectionHeadHNMNative.items_levels
    try:
        for mode in (True, False):
            DetectionHeadHNMNative.items_levels = mode
            net.zero_grad(set_to_none=True)
            fm_m = [f.detach().clone().requires_grad_(True) for f in fm]
            assert head._items_ok(fm_m) == mode
            pred = head(fm_m)
            gl = torch.randn(pred["box_logits"].shape, generator=torch.Generator().manual_seed(3)).cuda()
            gd = torch.randn(pred["box_deltas"].shape, generator=torch.Generator().manual_seed(4)).cuda()
            ((pred["box_logits"] * gl).sum() + (pred["box_deltas"] * gd).sum()).backward()
            torch.cuda.synchronize()
            res[mode] = (pred, [f.grad.clone() for f in fm_m],
                         {n: p.grad.detach().clone() for n, p in head.named_parameters() if p.grad is not None})
    finally:
        DetectionHeadHNMNative.items_levels = old
    (p1, x1, g1), (p0, x0, g0) = res[True], res[False]
    ftol = 5e-5 if dtype == torch.float32 else 2e-2
    for k in ("box_logits", "box_deltas"):
        assert p1[k].shape == p0[k].shape and p1[k].dtype == torch.float32
        assert float((p1[k] - p0[k]).detach().abs().max()) <= ftol * float(p0[k].detach().abs().max()), k
    gtol = 1e-4 if dtype == torch.float32 else 3e-2
    for a, b in zip(x1, x0):
        assert float((a.float() - b.float()).abs().max()) <= gtol * float(b.float().abs().max())
    assert set(g1) == set(g0) and any("scales" in n for n in g1)
    for n in g0:
        scale = float(g0[n].abs().max()) + 1e-12
        # d(scale_l) = sum(g * y) over a level of 8 .. 4096 positions: in bf16 the two routes' y differ by an ulp on some elements
        # (GroupNorm statistics order) and nothing averages that out on the small levels
        tol = 0.15 if (dtype != torch.float32 and "scales" in n) else gtol
        assert float((g1[n] - g0[n]).abs().max()) <= tol * scale, (n, float((g1[n] - g0[n]).abs().max()), scale)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16], ids=["f32", "bf16", "f16"])
def test_fused_first_trunk_layer_of_both_branches(dtype):
    """Round 5: the classifier's and the regressor's first trunk layer read the same ragged batch and run as ONE
    convolution 128 -> 2 x 128 + ONE GroupNorm over 256 channels with split output (arch/pyramid.py: _FusedItemsBlockFn,
    nndet_norm_apply_items_split / nndet_norm_backward_items_split). Against the two separate `items_block` launches: conv -> norm -> ReLU
    outputs BIT-IDENTICAL up to the norm statistics' atomics (1e-6 / 1 ulp), the input gradient = the SUM of the two branches' data
    gradients from one accumulator (closer to fp32 than the rounded sum of two rounded gradients), weight / norm gradients per branch."""
    from nndetection_amd.arch import pyramid as P
    a, b = _block(128, 128, True, 11), _block(128, 128, True, 12)
    batch = 2
    fm1 = _fmaps(128, dtype, batch, 5)
    fm2 = [f.detach().clone().requires_grad_(True) for f in fm1]
    x1, meta = P.cat_levels(fm1)
    x2, _ = P.cat_levels(fm2)
    assert P.fusable_pair(a, b, x1)
    ya, yb = P.items_block(a, x1, meta), P.items_block(b, x1, meta)
    fa, fb = P.fused_items_blocks(a, b, x2, meta)
    tol = 2e-6 if dtype == torch.float32 else 0.0
    for u, v, n in ((ya, fa, "a"), (yb, fb, "b")):
        assert u.shape == v.shape
        d = (u.float() - v.float()).abs()
        ulp = u.float().abs().clamp_min(1e-3) * (2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10 if dtype == torch.float16 else 0.0)
        assert bool((d <= ulp + tol * float(u.float().abs().max())).all()), (n, float(d.max()))
        assert float((d > 0).float().mean()) < 1e-3, (n, "more than 0.1 % of the outputs differ")
    g = torch.Generator().manual_seed(3)
    ga, gb = torch.randn(ya.shape, generator=g).cuda().to(dtype), torch.randn(yb.shape, generator=g).cuda().to(dtype)
    torch.autograd.backward([ya, yb], [ga, gb])
    ref = {n: p.grad.clone() for blk, tag in ((a, "a"), (b, "b")) for n, p in ((tag + "." + k, v) for k, v in blk.named_parameters())}
    ref_dx = [f.grad.clone() for f in fm1]
    a.zero_grad(set_to_none=True); b.zero_grad(set_to_none=True)
    torch.autograd.backward([fa, fb], [ga, gb])
    torch.cuda.synchronize()
    rel = 2e-5 if dtype == torch.float32 else (2e-2 if dtype == torch.bfloat16 else 3e-3)
    for blk, tag in ((a, "a"), (b, "b")):
        for k, p in blk.named_parameters():
            r = ref[tag + "." + k]
            assert p.grad is not None and float((p.grad - r).abs().max()) <= rel * float(r.abs().max()) + 1e-7, (tag, k)
    for f, r in zip(fm2, ref_dx):
        assert float((f.grad.float() - r.float()).abs().max()) <= rel * float(r.float().abs().max()), "input gradient"


def test_head_forward_uses_the_fused_first_layer(monkeypatch):
    """DetectionHeadHNMNative._forward_items takes the fused route by default and NNDET_HEAD_FUSE_CIN=0 restores the two launches; both
    give the same predictions."""
    from nndetection_amd.plans import get_plan
    from nndetection_amd.ptmodule import build_model
    from nndetection_amd.arch import heads as H
    from nndetection_amd import _lib as L
    plan = get_plan("toy64")
    torch.manual_seed(0)
    net = build_model(plan).cuda().eval()
    x = torch.randn(2, 1, *plan["patch_size"], device="cuda").to(torch.bfloat16)
    calls = []
    real = L.call
    monkeypatch.setattr(L, "call", lambda name, *a: (calls.append(name), real(name, *a))[1])
    outs = {}
    for fuse in (True, False):
        monkeypatch.setattr(H, "FUSE_CIN", fuse)
        calls.clear()
        with torch.no_grad():
            pred, _, _ = net(x)
        assert ("nndet_norm_apply_items_split" in calls) == fuse
        outs[fuse] = (pred["box_logits"].float().clone(), pred["box_deltas"].float().clone())
    for u, v in zip(outs[True], outs[False]):
        assert float((u - v).abs().max()) <= 2e-2 * float(v.abs().max())
