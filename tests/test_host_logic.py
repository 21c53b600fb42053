"""CPU: host-side logic that needs no GPU -- plans, state-dict parity with the reference-shaped oracle, NDHWC layout
helpers, optimizer groups / LR schedule, and the N > 1 gradient all-reduce path on gloo (world_size 2)."""
import os
import sys

import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plans_and_state_dict_keys_match_reference_shape():
    from nndetection_amd.plans import get_plan, MODEL_CFG_V001
    from nndetection_amd.ptmodule import build_model
    from oracle.retina_torch import OracleRetinaUNet
    for name, nparam in (("tiny", None), ("toy64", 13_356_771), ("luna160", 18_902_019), ("lidc192", 18_967_555)):
        p = get_plan(name)
        m = build_model(p)
        o = OracleRetinaUNet(p["arch"], p["anchors"], MODEL_CFG_V001)
        assert list(m.state_dict().keys()) == list(o.state_dict().keys())
        assert all(a.shape == b.shape for a, b in zip(m.state_dict().values(), o.state_dict().values()))
        if nparam is not None:                       # parameter counts of the reference architecture
            assert sum(x.numel() for x in m.parameters()) == nparam
    keys = list(build_model(get_plan("luna160")).state_dict().keys())
    assert len(keys) == 92
    for k in ("encoder.stages.0.convs.0.0.conv.weight", "encoder.stages.5.convs.0.1.norm.bias", "decoder.lateral.P3.0.conv.bias",
              "decoder.up.P5.conv.weight", "decoder.out.P1.0.conv.weight", "head.classifier.conv_internal.c_in.norm.weight",
              "head.regressor.conv_out.conv.bias", "head.regressor.scales.3.scale", "segmenter.conv_out.conv.weight"):
        assert k in keys, k


def test_factory_contract_types():
    """conv child IS nn.Conv3d, norm child IS an InstanceNorm3d / GroupNorm; bias only without norm."""
    from nndetection_amd.arch import Generator, ConvInstanceRelu, ConvGroupRelu
    c = Generator(ConvInstanceRelu, 3)(32, 64, kernel_size=3, stride=2, padding=1)
    assert isinstance(c.conv, nn.Conv3d) and isinstance(c.norm, nn.InstanceNorm3d) and isinstance(c.act, nn.ReLU)
    assert c.conv.bias is None and c.norm.weight.shape == (64,)
    g = Generator(ConvGroupRelu, 3)(128, 128, kernel_size=3, padding=1, norm_channels_per_group=16)
    assert isinstance(g.norm, nn.GroupNorm) and g.norm.num_groups == 8
    u = Generator(ConvInstanceRelu, 3)(64, 32, kernel_size=(2, 2, 1), stride=(2, 2, 1), transposed=True, add_norm=False, add_act=False)
    assert isinstance(u.conv, nn.ConvTranspose3d) and u.conv.bias is not None and list(u._modules) == ["conv"]


def test_classifier_prior_init():
    from nndetection_amd.ptmodule import build_model
    from nndetection_amd.plans import get_plan
    m = build_model(get_plan("tiny"))
    b = m.head.classifier.conv_out.conv.bias
    assert torch.allclose(b, torch.full_like(b, -4.59511985))       # -log(99), classifier.py:223
    assert abs(m.head.regressor.conv_out.conv.weight.std().item() - 0.01) < 2e-3


def test_layout_roundtrip_cpu():
    from nndetection_amd.layout import phys, logical, cpad
    assert [cpad(c) for c in (1, 2, 27, 32, 33, 162, 320)] == [32, 32, 32, 32, 64, 192, 320]
    x = torch.randn(2, 27, 3, 4, 5)
    p, c = phys(x)
    assert p.shape == (2, 3, 4, 5, 32) and c == 27 and p.is_contiguous()
    assert torch.equal(p[..., :27].permute(0, 4, 1, 2, 3), x) and (p[..., 27:] == 0).all()
    v = logical(p, 27)
    assert v.shape == x.shape and torch.equal(v, x)
    p2, _ = phys(v)                                   # our own padded view: zero-copy
    assert p2.data_ptr() == p.data_ptr()
    y = torch.randn(2, 64, 3, 4, 5).to(memory_format=torch.channels_last_3d)
    p3, _ = phys(y)
    assert p3.data_ptr() == y.data_ptr()              # channels_last_3d tensors are already NDHWC
    img = torch.randn(2, 1, 3, 4, 5)
    pi, ci = phys(img, dtype=torch.bfloat16)
    assert pi.shape == (2, 3, 4, 5, 1) and ci == 1 and pi.dtype == torch.bfloat16


def test_optimizer_groups_and_schedule():
    from nndetection_amd.ptmodule import build_model, configure_optimizer
    from nndetection_amd.plans import get_plan
    m = build_model(get_plan("luna160"))
    opt, sched = configure_optimizer(m, lean=False)
    no_wd, wd = opt.param_groups
    assert no_wd["weight_decay"] == 0.0 and wd["weight_decay"] == 3e-5
    assert len(no_wd["params"]) == 2 * 16 and len(no_wd["params"]) + len(wd["params"]) == 92     # 16 norm modules
    assert opt.defaults["nesterov"] and opt.defaults["momentum"] == 0.9
    # reference schedule: lr(k) = warm_lr + (lr0 - warm_lr) * k / 4000 during warm-up, k = _step_count (starts at 1)
    assert abs(opt.param_groups[0]["lr"] - (1e-6 + (0.01 - 1e-6) / 4000)) < 1e-12


def test_lean_optimizer_matches_torch_sgd():
    """nndetection_amd.optim.SGDNesterov + LinearWarmupPolyLR == torch.optim.SGD + reference schedule, step by step,
    including parameters without gradient in some steps (unused heads)."""
    from nndetection_amd.ptmodule import configure_optimizer
    cfg = {"initial_lr": 0.01, "sgd_momentum": 0.9, "sgd_nesterov": True, "weight_decay": 3e-5, "warm_iterations": 3,
           "warm_lr": 1e-6, "poly_gamma": 0.9, "max_num_epochs": 1, "num_train_batches_per_epoch": 10}
    torch.manual_seed(0)
    def make():
        torch.manual_seed(1)
        return nn.Sequential(nn.Conv3d(2, 4, 3), nn.InstanceNorm3d(4, affine=True), nn.Conv3d(4, 2, 1), nn.Linear(3, 3))
    a, b = make(), make()
    oa, sa = configure_optimizer(a, cfg, lean=True)
    ob, sb = configure_optimizer(b, cfg, lean=False)
    for it in range(8):
        x = torch.randn(2, 2, 5, 5, 5)
        for m, o, s in ((a, oa, sa), (b, ob, sb)):
            y = m[2](m[1](m[0](x))).sum()
            if it % 3 != 1:
                y = y + m[3](torch.ones(3)).sum()       # m[3] has no gradient in some steps
            y.backward()
            o.step(); s.step(); o.zero_grad(set_to_none=True)
        assert abs(oa.param_groups[0]["lr"] - ob.param_groups[0]["lr"]) < 1e-15
        for pa, pb in zip(a.parameters(), b.parameters()):
            assert torch.allclose(pa, pb, atol=1e-7, rtol=1e-6), it


def _ddp_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from nndetection_amd.ddp import GradAllReducer
    torch.manual_seed(rank)                            # different init per rank: must be broadcast from rank 0
    model = nn.Sequential(nn.Linear(8, 16), nn.ReLU(), nn.Linear(16, 4), nn.Linear(4, 4))
    ddp = GradAllReducer(model, first_bucket_mb=1e-4, bucket_mb=2e-4)
    w0 = model[0].weight.detach().clone()
    torch.manual_seed(100 + rank)
    x = torch.randn(5, 8)
    h = model[2](model[1](model[0](x)))
    loss = h.sum() if rank == 0 else (model[3](h)).sum()   # rank 0 never uses the last layer: its grads are None there
    loss.backward()
    ddp.finish()
    # numpy arrays (pickled by value): torch tensors travel as file descriptors, and the parent's fd-receive raced with this
    # process exiting under load (ConnectionResetError / FileNotFoundError)
    q.put((rank, w0.numpy(), [p.grad.detach().numpy().copy() for p in model.parameters()], len(ddp.buckets)))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_allreduce_gloo_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    res = [(r, torch.from_numpy(w), [torch.from_numpy(g) for g in gs], nb) for r, w, gs, nb in res]
    (r0, w0a, g0, nb), (r1, w0b, g1, _) = res
    assert torch.equal(w0a, w0b), "parameters were not broadcast from rank 0"
    assert nb >= 2, "expected several buckets"
    for a, b in zip(g0, g1):
        assert torch.allclose(a, b, atol=1e-6), "ranks disagree after the all-reduce"
    # reference: average of the two local gradients with the unused layer zero-filled on rank 0
    torch.manual_seed(0)
    model = nn.Sequential(nn.Linear(8, 16), nn.ReLU(), nn.Linear(16, 4), nn.Linear(4, 4))
    grads = []
    for rank in range(2):
        model.zero_grad()
        torch.manual_seed(100 + rank)
        x = torch.randn(5, 8)
        h = model[2](model[1](model[0](x)))
        (h.sum() if rank == 0 else model[3](h).sum()).backward()
        grads.append([torch.zeros_like(p) if p.grad is None else p.grad.clone() for p in model.parameters()])
    for got, a, b in zip(g0, *grads):
        assert torch.allclose(got, (a + b) / 2, atol=1e-6)


def _ddp_static_unused_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from nndetection_amd.ddp import GradAllReducer
    torch.manual_seed(0)
    model = nn.Sequential(nn.Linear(8, 16), nn.ReLU(), nn.Linear(16, 4), nn.Linear(4, 4))    # model[3] is never used
    unused = list(model[3].parameters())
    launched = {}
    for tag, su in (("declared", unused), ("undeclared", [])):
        ddp = GradAllReducer(model, first_bucket_mb=1e-4, bucket_mb=2e-4, static_unused=su)
        model.zero_grad(set_to_none=True)
        torch.manual_seed(100 + rank)
        model[2](model[1](model[0](torch.randn(5, 8)))).sum().backward()
        launched[tag] = (ddp._next, len(ddp.buckets))          # buckets already in flight when backward returns
        ddp.finish()
        for h in ddp._hooks:
            h.remove()
    q.put((rank, launched, [p.grad.detach().numpy().copy() for p in model.parameters()]))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_allreduce_static_unused_does_not_block_overlap():
    """A parameter that never gets a gradient (decoder.out.P1 of RetinaUNet) sits in the FIRST bucket (reverse registration
    order). Undeclared, no bucket can be launched before finish(); declared as static_unused, every bucket is in flight when
    backward returns, and the unused parameter's gradient is zero on every rank."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=_ddp_static_unused_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda r: r[0])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    res = [(r, l, [torch.from_numpy(g) for g in gs]) for r, l, gs in res]
    for rank, launched, grads in res:
        n_decl, nb = launched["declared"]
        n_undecl, _ = launched["undeclared"]
        assert nb >= 2
        assert n_decl == nb, f"rank {rank}: only {n_decl}/{nb} buckets launched during backward"
        assert n_undecl == 0, "without the declaration the first bucket (and so all of them) waits for finish()"
        assert float(grads[-1].abs().max()) == 0.0 and float(grads[-2].abs().max()) == 0.0
    for a, b in zip(res[0][2], res[1][2]):
        assert torch.allclose(a, b, atol=1e-6)


def test_never_used_parameters_of_retina_unet():
    from nndetection_amd.plans import get_plan
    from nndetection_amd.ptmodule import build_model
    assert build_model(get_plan("tiny")).never_used_parameters() == []      # every level of `tiny` is read by the head / segmenter
    net = build_model(get_plan("luna160"))                                   # levels (2,3,4,5) + 0: P1 is never read
    names = {n for n, p in net.named_parameters() if any(p is q for q in net.never_used_parameters())}
    assert names == {"decoder.out.P1.0.conv.weight", "decoder.out.P1.0.conv.bias"}, names
    assert names and all(n.startswith("decoder.out.P") for n in names), names
    used = net.decoder.used_levels
    assert all(int(n.split(".")[2][1:]) not in used for n in names)


def test_force_overlap_world1_hook_bucket_view_path():
    """force_overlap=True: the post-accumulate hooks, bucket copies and gradient views run without a process group (what the
    one-GPU `-m gpu` test uses to exercise the overlapped path against the multi-stream head)."""
    from nndetection_amd.ddp import GradAllReducer
    torch.manual_seed(0)
    model = nn.Sequential(nn.Linear(8, 16), nn.ReLU(), nn.Linear(16, 4), nn.Linear(4, 4))
    ref = [None] * 6
    x = torch.randn(5, 8)
    model[2](model[1](model[0](x))).sum().backward()
    ref = [None if p.grad is None else p.grad.clone() for p in model.parameters()]
    model.zero_grad(set_to_none=True)
    ddp = GradAllReducer(model, first_bucket_mb=1e-4, bucket_mb=2e-4, force_overlap=True, static_unused=list(model[3].parameters()))
    model[2](model[1](model[0](x))).sum().backward()
    assert ddp._next == len(ddp.buckets) >= 2            # everything was "launched" from the hooks
    ddp.finish()
    for p, r in zip(model.parameters(), ref):
        assert p.grad is not None and p.grad._base is not None       # a view of its bucket
        assert torch.equal(p.grad, torch.zeros_like(p) if r is None else r)


def test_lean_sgd_state_dict_roundtrip_matches_torch_layout():
    """SGDNesterov / LinearWarmupPolyLR can be checkpointed; the layout is torch.optim.SGD's, so the state
    loads into a torch.optim.SGD over the same groups and continues identically."""
    from nndetection_amd.optim import SGDNesterov, LinearWarmupPolyLR
    torch.manual_seed(0)
    def make():
        torch.manual_seed(1)
        m = nn.Sequential(nn.Linear(6, 5), nn.GroupNorm(1, 5), nn.Linear(5, 3))
        groups = [{"params": list(m[1].parameters()), "weight_decay": 0.0},
                  {"params": list(m[0].parameters()) + list(m[2].parameters()), "weight_decay": 3e-5}]
        return m, groups
    def run(m, opt, sched, n, seed):
        torch.manual_seed(seed)
        for _ in range(n):
            m(torch.randn(4, 6)).pow(2).sum().backward()
            opt.step(); sched.step(); opt.zero_grad(set_to_none=True)
    m1, g1 = make(); o1 = SGDNesterov(g1, 0.01); s1 = LinearWarmupPolyLR(o1, 5, 1e-6, 0.9, 50)
    run(m1, o1, s1, 4, 7)
    import copy
    sd_o, sd_s, sd_m = copy.deepcopy(o1.state_dict()), s1.state_dict(), {k: v.clone() for k, v in m1.state_dict().items()}   # (like torch, state_dict() holds references)
    assert set(sd_o) == {"state", "param_groups"} and all("momentum_buffer" in v for v in sd_o["state"].values())
    run(m1, o1, s1, 3, 8)
    # resume the lean pair from the checkpoint
    m2, g2 = make(); m2.load_state_dict(sd_m); o2 = SGDNesterov(g2, 0.01); s2 = LinearWarmupPolyLR(o2, 5, 1e-6, 0.9, 50)
    o2.load_state_dict(sd_o); s2.load_state_dict(sd_s)
    run(m2, o2, s2, 3, 8)
    for a, b in zip(m1.parameters(), m2.parameters()):
        assert torch.equal(a, b)
    # the same checkpoint loads into torch.optim.SGD (same layout)
    m3, g3 = make(); m3.load_state_dict(sd_m)
    o3 = torch.optim.SGD(g3, 0.01, momentum=0.9, nesterov=True)
    o3.load_state_dict(sd_o)
    assert len(o3.state) == len(sd_o["state"])
    lr_now = sd_o["param_groups"][0]["lr"]
    assert abs(o3.param_groups[0]["lr"] - lr_now) < 1e-15


def test_cat_rows_without_copy():
    """arch/heads._cat_rows: per-image lists that are the rows of one contiguous tensor (what the batched target assignment returns)
    are concatenated as a view; anything else falls back to torch.cat."""
    import torch
    from nndetection_amd.arch.heads import _cat_rows
    a = torch.arange(24.).view(4, 6)
    r = _cat_rows(list(a.unbind(0)))
    assert r.data_ptr() == a.data_ptr() and torch.equal(r, a.reshape(-1))
    b = torch.arange(48.).view(2, 4, 6)
    r = _cat_rows(list(b.unbind(0)))
    assert r.shape == (8, 6) and r.data_ptr() == b.data_ptr() and torch.equal(r, b.reshape(-1, 6))
    r = _cat_rows([b[1], b[0]])                                  # not consecutive: a copy, in list order
    assert torch.equal(r, torch.cat([b[1], b[0]])) and r.data_ptr() != b.data_ptr()
    r = _cat_rows([torch.zeros(3), torch.ones(3)])               # separate storages
    assert torch.equal(r, torch.tensor([0., 0, 0, 1, 1, 1]))
    assert _cat_rows(a) is a and _cat_rows([a[0]]).data_ptr() == a.data_ptr()


def test_pyramid_meta_item_table():
    """arch/pyramid.PyramidMeta: level-major items, running voxel-row offsets, the C struct of include/nndet_amd.h (NndetItems)."""
    import ctypes
    from nndetection_amd import _lib as L
    from nndetection_amd.arch.pyramid import pyramid_meta
    shapes = [(4, 40, 40, 24), (4, 20, 20, 12), (4, 10, 10, 6), (4, 5, 5, 6)]
    m = pyramid_meta(shapes)
    assert m is pyramid_meta(shapes)                              # cached per shape key
    assert m.n_items == 16 and m.batch == 4 and m.rows == 4 * (38400 + 4800 + 600 + 150)
    assert m.level_rows == [(0, 153600), (153600, 19200), (172800, 2400), (175200, 600)]
    assert list(m.items.dims[0]) == [40, 40, 24] and list(m.items.dims[5]) == [20, 20, 12] and list(m.items.dims[15]) == [5, 5, 6]
    assert m.items.row_off[0] == 0 and m.items.row_off[4] == 153600 and m.items.row_off[5] == 153600 + 4800 and m.items.row_off[15] == 175200 + 450
    assert ctypes.sizeof(L.NndetItems) == 8 + L.MAX_ITEMS * 12 + L.MAX_ITEMS * 8 and L.MAX_ITEMS == 32


def test_fused_sgd_version_counters_are_advanced_by_hand():
    """torch._fused_sgd_ writes parameters without incrementing `_version` (the packed convolution weights are cached per version):
    nndetection_amd.optim advances the counters itself. Pins both facts -- if torch starts bumping, the helper just bumps once more."""
    from nndetection_amd.optim import _bump_versions
    ps = [torch.nn.Parameter(torch.ones(4)), torch.nn.Parameter(torch.ones(2, 3))]
    before = [p._version for p in ps]
    _bump_versions(ps)
    assert [p._version for p in ps] == [v + 1 for v in before]
    assert all(bool((p == 1).all()) for p in ps)                      # values untouched
    if hasattr(torch, "_fused_sgd_"):
        p, g, b = torch.ones(4), torch.ones(4), torch.empty(4)
        v = p._version
        try:
            torch._fused_sgd_([p], [g], [b], weight_decay=0.0, momentum=0.9, lr=0.1, dampening=0.0, nesterov=True, maximize=False,
                              is_first_step=True, grad_scale=None, found_inf=None)
        except (RuntimeError, NotImplementedError):
            return                                                        # no CPU kernel in this build
        assert float(p[0]) != 1.0 and p._version in (v, v + 1)


def test_absorbed_top_down_step_algebra():
    """arch/segmenter.py (NNDET_SEG_UP): conv3(up(x1) + b; wc) as one half-resolution 3x3x3 convolution with 8 parity output channels
    plus a border-class bias, and its parameter gradients from that convolution's weight gradient -- against plain torch in fp64."""
    import torch.nn.functional as F
    from nndetection_amd.arch.segmenter import up_compose, up_param_grads
    torch.manual_seed(3)
    dd = torch.float64
    N, I, C, D2, H2, W2 = 2, 6, 5, 3, 4, 2
    x1 = torch.randn(N, I, D2, H2, W2, dtype=dd, requires_grad=True)
    w_up = torch.randn(I, C, 2, 2, 2, dtype=dd, requires_grad=True)
    bsum = torch.randn(C, dtype=dd, requires_grad=True)
    wc = torch.randn(27, C, dtype=dd, requires_grad=True)
    G = torch.randn(N, 1, 2 * D2, 2 * H2, 2 * W2, dtype=dd)
    u = F.conv_transpose3d(x1, w_up, bias=bsum, stride=2)
    u.retain_grad()
    zr = F.conv3d(u, wc.t().reshape(1, C, 3, 3, 3), padding=1)
    (zr * G).sum().backward()
    # ---- forward
    Wc, cb = up_compose(wc.detach(), w_up.detach(), bsum.detach())
    zup = F.conv3d(x1.detach(), Wc, padding=1)                               # [N, 8, D2, H2, W2]
    z = zup.reshape(N, 2, 2, 2, D2, H2, W2).permute(0, 4, 1, 5, 2, 6, 3).reshape(N, 1, 2 * D2, 2 * H2, 2 * W2).clone()

    def cls(n):
        c = torch.ones(n, dtype=torch.long); c[0] = 0; c[-1] = 2
        return c
    cd, ch, cw = cls(2 * D2), cls(2 * H2), cls(2 * W2)
    z += cb.reshape(3, 3, 3)[cd][:, ch][:, :, cw]
    assert torch.allclose(z, zr.detach(), rtol=1e-10, atol=1e-10)
    # ---- backward from d1 = G
    d1 = G[:, 0]
    dzs = d1.reshape(N, D2, 2, H2, 2, W2, 2).permute(0, 2, 4, 6, 1, 3, 5).reshape(N, 8, D2, H2, W2)
    dx1 = torch.nn.grad.conv3d_input(x1.shape, Wc, dzs, padding=1)
    dWc = torch.nn.grad.conv3d_weight(x1.detach(), Wc.shape, dzs, padding=1)
    csum = torch.zeros(3, 3, 3, dtype=dd)
    csum.index_put_((cd[:, None, None].expand(2 * D2, 2 * H2, 2 * W2), ch[None, :, None].expand(2 * D2, 2 * H2, 2 * W2),
                     cw[None, None, :].expand(2 * D2, 2 * H2, 2 * W2)), d1.sum(0), accumulate=True)
    dw_up, dbsum, ec = up_param_grads(wc.detach(), w_up.detach(), bsum.detach(), dWc, csum.reshape(27))
    assert torch.allclose(dx1, x1.grad, rtol=1e-10, atol=1e-10)
    assert torch.allclose(dw_up, w_up.grad, rtol=1e-10, atol=1e-10)
    assert torch.allclose(dbsum, bsum.grad, rtol=1e-10, atol=1e-10)
    # Ec_u[t][k] = sum_p d1[p] u[p + t - 1][k] = the gradient of the composed kernel wc
    assert torch.allclose(ec, wc.grad, rtol=1e-10, atol=1e-10)


def test_which_steps_may_absorb_the_top_down_step():
    """core/retina.py `_seg_up_ok` / `_seg_lateral_ok`: the last top-down step may only go into the segmentation branch when decoder
    level 1 has no other reader (the head starts at level 2, out.P1 is skipped), up.P1 is a plain k = s = 2 transposed convolution onto
    32 channels and the patch dimensions are even."""
    from nndetection_amd.plans import get_plan
    from nndetection_amd.ptmodule import build_model
    from nndetection_amd.arch import segmenter as S
    luna = build_model(get_plan("luna160"))
    assert luna._seg_lateral_ok()
    assert luna._seg_up_ok(torch.zeros(1, 1, 160, 160, 96))
    assert luna._seg_up_ok(torch.zeros(1, 1, 192, 64, 32))
    assert not luna._seg_up_ok(torch.zeros(1, 1, 160, 160, 95))           # odd depth: the parity classes would be ragged
    assert not luna._seg_up_ok(torch.zeros(1, 160, 160, 96))              # not a 5-D batch
    old = S.SEG_UP
    try:
        S.SEG_UP = False
        assert not luna._seg_up_ok(torch.zeros(1, 1, 160, 160, 96))
    finally:
        S.SEG_UP = old
    toy = build_model(get_plan("toy64"))                                    # its detection head reads level 1
    assert 1 in tuple(toy.decoder_levels) and not toy._seg_up_ok(torch.zeros(1, 1, 64, 64, 64))


# ---- round 5: gradients written IN PLACE into the reducer's buckets (static gradient-pool layout), gradient-free parameters declared
# ---- during the forward pass (a rank without positive anchors)
class _PoolLinearFn(torch.autograd.Function):
    """A linear layer whose backward takes its parameter-gradient memory from _lib.grad_pool exactly like the convolution nodes do
    (arch/conv.py: take_for): accumulates into zeroed memory and hands autograd views of it."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w, b)
        return x @ w.t() + b

    @staticmethod
    def backward(ctx, g):
        from nndetection_amd import _lib as L
        x, w, b = ctx.saved_tensors
        gw, gb = L.grad_pool.take_for([(w, w.numel()), (b, b.numel())], g.device)
        dw = gw.view(w.shape)
        dw += g.t() @ x
        gb += g.sum(0)
        return g @ w, dw, gb


class _PoolNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.a, self.b, self.c, self.reg = nn.Linear(8, 16), nn.Linear(16, 4), nn.Linear(4, 4), nn.Linear(4, 4)

    def forward(self, x, use_reg=True):
        from nndetection_amd import _lib as L
        L.grad_pool.begin(sum(p.numel() + 64 for p in self.parameters()), x.device, owner=self)
        h = x
        for m in (self.a, self.b, self.c):
            h = torch.relu(_PoolLinearFn.apply(h, m.weight, m.bias))
        if use_reg:
            return h.sum() + _PoolLinearFn.apply(h, self.reg.weight, self.reg.bias).sum()
        L.notify_no_grad(list(self.reg.parameters()))          # what arch/heads.py does on a batch without positive anchors
        return h.sum()


def _ddp_inplace_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from nndetection_amd.ddp import GradAllReducer
    from nndetection_amd import _lib as L
    torch.manual_seed(0)
    model = _PoolNet()
    ddp = GradAllReducer(model, first_bucket_mb=1e-4, bucket_mb=2e-4)
    out = {}
    for step in range(2):
        torch.manual_seed(100 + rank + 10 * step)
        x = torch.randn(5, 8)
        loss = model(x, use_reg=(rank == 0))                  # rank 1: no gradient for `reg` (declared during the forward pass)
        loss.backward()
        n_before = ddp._next                                    # buckets in flight when backward() returns
        ddp.finish()
        lo, hi = ddp._flat_all.data_ptr(), ddp._flat_all.data_ptr() + ddp._flat_all.numel() * 4
        out[step] = dict(launched=list(ddp.launched_from_hooks), n_before=n_before, nb=len(ddp.buckets), copied=ddp.copied_last,
                         inside=all(lo <= p.grad.data_ptr() < hi for p in model.parameters()),
                         grads=[p.grad.detach().numpy().copy() for p in model.parameters()])
        model.zero_grad(set_to_none=True)
    ddp.close()
    assert L.grad_pool.static is None and not L.no_grad_listeners
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_grad_allreduce_in_place_and_gradient_free_rank_gloo_world2():
    """VERDICT r4 item 8a/8b. (a) The nodes write their parameter gradients into the reducer's bucket memory (static pool layout):
    no bucket copy (`copied_last == 0`), p.grad lives inside the flat buffer. (b) Rank 1 has no gradient for the `reg` layer (a rank
    without positive anchors, nndet/arch/heads/comb.py:397-401) and says so during its forward pass: on BOTH ranks every bucket is
    launched from the gradient hooks, none from finish(). Values: the mean of the two ranks' local gradients, zeros for `reg` on rank 1."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + os.getpid() % 2000
    procs = [ctx.Process(target=_ddp_inplace_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = dict(q.get(timeout=120) for _ in procs)
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    for step in range(2):
        local = []
        for rank in range(2):                                  # the same two local gradients with stock autograd
            torch.manual_seed(0)
            ref = _PoolNet()
            torch.manual_seed(100 + rank + 10 * step)
            x = torch.randn(5, 8)
            h = x
            for m in (ref.a, ref.b, ref.c):
                h = torch.relu(m(h))
            (h.sum() + (ref.reg(h).sum() if rank == 0 else 0.0)).backward()
            local.append([torch.zeros_like(p) if p.grad is None else p.grad.clone() for p in ref.parameters()])
        for rank in range(2):
            r = res[rank][step]
            assert r["nb"] >= 3 and r["launched"] == [True] * r["nb"] and r["n_before"] == r["nb"], (rank, step, r["launched"], r["n_before"])
            assert r["copied"] == 0, f"rank {rank} step {step}: {r['copied']} gradients were copied into their bucket"
            assert r["inside"], "a gradient lives outside the reducer's flat buffer"
            for got, a, b in zip(r["grads"], *local):
                assert torch.allclose(torch.from_numpy(got), (a + b) / 2, atol=1e-6), (rank, step)


def test_static_pool_hands_a_region_out_once_and_only_without_grad():
    """The static layout must not alias two contributions: a second node of the same parameter in one backward pass, and a second
    backward pass while `.grad` is still set (gradient accumulation), get fresh memory; autograd adds them in. Another model in the
    same process never sees the layout (owner check)."""
    from nndetection_amd.ddp import GradAllReducer
    from nndetection_amd import _lib as L
    torch.manual_seed(0)
    model = _PoolNet()
    ddp = GradAllReducer(model, first_bucket_mb=1e-4, bucket_mb=2e-4, force_overlap=True)
    try:
        x = torch.randn(5, 8)
        ref = _PoolNet(); ref.load_state_dict(model.state_dict())
        h = x
        for m in (ref.a, ref.b, ref.c):
            h = torch.relu(m(h))
        (h.sum() + ref.reg(h).sum() + ref.reg(2 * h).sum()).backward()
        want = [p.grad.clone() for p in ref.parameters()]
        # `reg` used twice in one pass
        L.grad_pool.begin(0, x.device, owner=model)
        h = x
        for m in (model.a, model.b, model.c):
            h = torch.relu(_PoolLinearFn.apply(h, m.weight, m.bias))
        (h.sum() + _PoolLinearFn.apply(h, model.reg.weight, model.reg.bias).sum()
         + _PoolLinearFn.apply(2 * h, model.reg.weight, model.reg.bias).sum()).backward()
        ddp.finish()
        for p, w in zip(model.parameters(), want):
            assert torch.allclose(p.grad, w, atol=1e-5)
        # two forward passes, then their two backward passes, no zero_grad in between: the second pass finds `.grad` set (it lives in the
        # parameter's region) and must accumulate through fresh memory, not write into the region again
        model.zero_grad(set_to_none=True)
        la, lb = model(x), model(3 * x)
        la.backward(); lb.backward()
        ddp.finish()
        ref.zero_grad(set_to_none=True)
        for xx in (x, 3 * x):
            h = xx
            for m in (ref.a, ref.b, ref.c):
                h = torch.relu(m(h))
            (h.sum() + ref.reg(h).sum()).backward()
        for p, g in zip(model.parameters(), ref.parameters()):
            assert torch.allclose(p.grad, g.grad, atol=1e-4)
        model.zero_grad(set_to_none=True)
        other = _PoolNet()
        other(x).backward()                                     # not the owner: per-step pool, gradients outside the buckets
        lo, hi = ddp._flat_all.data_ptr(), ddp._flat_all.data_ptr() + ddp._flat_all.numel() * 4
        assert not any(lo <= p.grad.data_ptr() < hi for p in other.parameters())
    finally:
        ddp.close()


def test_static_pool_never_wipes_gradients_that_live_in_it():
    """after a backward pass -- and after finish() -- the parameters' gradients ARE the reducer's flat memory. A
    grad-enabled forward pass must not zero-fill it while they live there: (a) a forward / backward micro-batch loop without zero_grad
    (gradient accumulation), (b) a forward pass between finish() and optimizer.step(), (c) a forward pass INSIDE a backward pass
    (checkpoint recompute). In all three the layout stays un-armed, the per-step pool serves the nodes and autograd adds in place."""
    from nndetection_amd.ddp import GradAllReducer
    from nndetection_amd import _lib as L
    torch.manual_seed(0)
    model = _PoolNet()
    ref = _PoolNet(); ref.load_state_dict(model.state_dict())
    ddp = GradAllReducer(model, first_bucket_mb=1e-4, bucket_mb=2e-4, force_overlap=True)

    def ref_step(xx):
        h = xx
        for m in (ref.a, ref.b, ref.c):
            h = torch.relu(m(h))
        (h.sum() + ref.reg(h).sum()).backward()

    try:
        x1, x2, x3 = torch.randn(5, 8), torch.randn(5, 8), torch.randn(5, 8)
        # (a) two micro-batches, finish() after each backward, NO zero_grad in between
        model(x1).backward(); ddp.finish()
        assert not L.grad_pool.armed
        g1 = [p.grad.clone() for p in model.parameters()]
        model(x2).backward(); ddp.finish()
        ref_step(x1); ref_step(x2)
        for p, g in zip(model.parameters(), ref.parameters()):
            assert torch.allclose(p.grad, g.grad, atol=1e-4), "accumulated gradient differs from stock autograd"
        # (b) a grad-enabled forward pass between finish() and optimizer.step(): the reduced gradients survive
        before = [p.grad.clone() for p in model.parameters()]
        loss_unused = model(x3)                                  # begin() runs here
        assert not L.grad_pool.armed, "the static layout was armed on top of live gradients"
        for p, b in zip(model.parameters(), before):
            assert torch.equal(p.grad, b), "a forward pass wiped the gradients the optimizer is about to read"
        del loss_unused
        model.zero_grad(set_to_none=True); ref.zero_grad(set_to_none=True)
        assert g1[0].abs().sum() > 0
        # (c) a forward pass of the owner INSIDE a backward pass leaves the armed layout (and what was written into it) alone
        inner = {}

        class Recompute(torch.autograd.Function):
            @staticmethod
            def forward(ctx, t):
                return t * 1.0

            @staticmethod
            def backward(ctx, g):
                with torch.enable_grad():
                    inner["armed_before"], inner["taken_before"] = L.grad_pool.armed, set(L.grad_pool.taken)
                    model(x2)                                   # checkpoint-style recompute: calls grad_pool.begin inside the pass
                    inner["armed_after"], inner["taken_after"] = L.grad_pool.armed, set(L.grad_pool.taken)
                return g

        h = x1
        L.grad_pool.begin(0, x1.device, owner=model)
        assert L.grad_pool.armed
        for m in (model.a, model.b, model.c):
            h = torch.relu(_PoolLinearFn.apply(h, m.weight, m.bias))
        (Recompute.apply(h).sum() + _PoolLinearFn.apply(h, model.reg.weight, model.reg.bias).sum()).backward()
        ddp.finish()
        assert inner["armed_before"] and inner["taken_before"], "the reg node ran before the recompute node and took its regions"
        assert not inner["armed_after"], "a forward pass inside a backward pass must not re-arm (and zero) the static layout"
        ref_step(x1)
        for p, g in zip(model.parameters(), ref.parameters()):
            assert torch.allclose(p.grad, g.grad, atol=1e-4)
    finally:
        ddp.close()


def test_no_grad_declaration_of_an_abandoned_step_does_not_leak_into_the_next():
    """a no-positives forward pass declares the regressor gradient-free; if that step is abandoned (skipped, exception, NaN
    guard: no backward, no finish) the next step WITH positives must neither raise from the gradient hook nor launch a bucket early."""
    from nndetection_amd.ddp import GradAllReducer
    torch.manual_seed(0)
    model = _PoolNet()
    ddp = GradAllReducer(model, first_bucket_mb=1e-4, bucket_mb=2e-4, force_overlap=True)
    try:
        x = torch.randn(5, 8)
        model(x, use_reg=False)                                  # declares `reg` gradient-free ... and is abandoned
        assert ddp._marked
        model(x, use_reg=True).backward()                        # a new top-level forward pass: the declarations are dropped
        ddp.finish()
        assert ddp.launched_from_hooks == [True] * len(ddp.buckets)
        assert all(b.pending == b.expected for b in ddp.buckets) and not ddp._marked
        ref = _PoolNet(); ref.load_state_dict(model.state_dict())
        h = x
        for m in (ref.a, ref.b, ref.c):
            h = torch.relu(m(h))
        (h.sum() + ref.reg(h).sum()).backward()
        for p, g in zip(model.parameters(), ref.parameters()):
            assert torch.allclose(p.grad, g.grad, atol=1e-5)
    finally:
        ddp.close()


def test_gradient_side_channel_is_void_after_an_in_place_write_and_cannot_alias():
    """VERDICT r5 (weak: side channels keyed by data_ptr()). The key is an address, but (a) the entry holds its tensor, so the allocator
    cannot hand that address to another tensor while the entry lives, (b) the entry carries the version counter -- an in-place
    accumulation by the autograd engine (it adds a second contribution in place when it holds the only reference) or by a hook voids
    it, (c) a sub-view at the same address with another size is not a match, (d) views of the same storage and size still match."""
    from nndetection_amd._lib import _GradHints
    h = _GradHints()
    t = torch.zeros(4, 6)
    h.put(t, {"k": 1})
    assert h.pop(t.view(24)) == {"k": 1} and h.pop(t) is None          # a view of all of it matches once
    h.put(t, {"k": 2})
    assert h.pop(t[:2]) is None                                         # same address, other extent: not ours (and the entry is consumed)
    h.put(t, {"k": 3})
    t.add_(1.0)                                                         # what InputBuffer::accumulate does with a uniquely held gradient
    assert h.pop(t) is None
    h.put(t, {"k": 4})
    v = t.view(24)
    v.mul_(2.0)                                                         # through a view: the version counter is shared
    assert h.pop(t) is None
    # (a) while an entry lives its address is taken: a new tensor of the same size never lands on it
    h.put(t, {"k": 5})
    addr = t.data_ptr()
    del t, v
    others = [torch.zeros(4, 6) for _ in range(64)]
    assert all(o.data_ptr() != addr for o in others)
    h.clear()
    assert not h.d


def test_private_autograd_hooks_are_probed_once_and_degrade_gracefully():
    """The multi-stream backward pass hangs off two private torch hooks ("one torch upgrade from breaking"): both are probed at
    import; this torch has them, the graph-task id tells a backward pass from the outside, and without them the weight-gradient stream
    is simply off (the product then stays on the calling stream -- same results)."""
    from nndetection_amd import _lib as L
    assert L.PRIVATE_AUTOGRAD_HOOKS
    assert L.graph_task_id() == -1
    seen = []

    class Probe(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 2.0

        @staticmethod
        def backward(ctx, g):
            seen.append(L.graph_task_id())
            return g * 2.0

    x = torch.ones(3, requires_grad=True)
    Probe.apply(x).sum().backward()
    assert len(seen) == 1 and seen[0] >= 0 and L.graph_task_id() == -1
    assert L._WgradStreams.enabled == (os.environ.get("NNDET_WGRAD_STREAM", "1") != "0")
    assert L.wgrad_streams.side(torch.device("cpu"), torch.nn.Parameter(torch.zeros(1))) is None      # CPU tensors never leave the stream
