"""Diagnostic: the ragged head path at the luna160 pyramid sizes, stage by stage (finite checks + comparison with the per-level path)."""
import sys
import torch
from nndetection_amd.arch import pyramid as P
from nndetection_amd.arch.conv import ConvGroupRelu

LEV = [(40, 40, 24), (20, 20, 12), (10, 10, 6), (5, 5, 6)]


def block(cin, cout, norm, seed):
    torch.manual_seed(seed)
    m = ConvGroupRelu(3, cin, cout, 3, stride=1, padding=1, add_norm=norm, add_act=norm, bias=None if norm else True)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.ndim == 5:
                p.copy_(torch.randn_like(p) / (p[0].numel() ** 0.5))
    return m.cuda()


def run(batch, dtype):
    blocks = [block(128, 128, True, 1), block(128, 128, True, 2), block(128, 162, False, 3)]
    g = torch.Generator().manual_seed(0)
    fm = [torch.randn(batch, 128, *s, generator=g).cuda().to(dtype).requires_grad_(True) for s in LEV]
    fm2 = [f.detach().clone().requires_grad_(True) for f in fm]
    x2d, meta = P.cat_levels(fm)
    ts = [x2d]
    for b in blocks:
        ts.append(P.items_block(b, ts[-1], meta))
        ts[-1].retain_grad()
    print(f"batch {batch} {dtype}: forward finite:", [bool(torch.isfinite(t.float()).all()) for t in ts])
    out = P.head_gather_items(ts[-1], meta, 162, [])
    go = torch.randn(out.shape, generator=torch.Generator().manual_seed(5)).cuda()
    (out * go).sum().backward()
    torch.cuda.synchronize()
    print("  grads finite (stages):", [bool(torch.isfinite(t.grad.float()).all()) for t in ts[1:]],
          "inputs:", [bool(torch.isfinite(f.grad.float()).all()) for f in fm])
    for i, b in enumerate(blocks):
        for n, p in b.named_parameters():
            ok = bool(torch.isfinite(p.grad).all())
            if not ok:
                bad = (~torch.isfinite(p.grad)).sum().item()
                print(f"  block {i} {n}: {bad} of {p.grad.numel()} non-finite")
    ref = {}
    for i, b in enumerate(blocks):
        for n, p in b.named_parameters():
            ref[(i, n)] = p.grad.clone()
        b.zero_grad(set_to_none=True)
    # per level
    outs = []
    for f in fm2:
        t = f
        for b in blocks:
            t = b(t)
        outs.append(t)
    off = 0
    tot = 0.0
    for o, s in zip(outs, LEV):
        n = s[0] * s[1] * s[2]
        gl = go[:, off:off + n].reshape(batch, *s, 162).permute(0, 4, 1, 2, 3).to(dtype)
        off += n
        tot = tot + (o.float() * gl.float()).sum()
    tot.backward()
    torch.cuda.synchronize()
    for i, b in enumerate(blocks):
        for n, p in b.named_parameters():
            r = ref[(i, n)]
            d = float((r - p.grad).abs().max()) if torch.isfinite(r).all() else float("nan")
            print(f"  block {i} {n}: per-level finite {bool(torch.isfinite(p.grad).all())}, |items - levels| {d:.3e} of {float(p.grad.abs().max()):.3e}")
    for a, b_ in zip(fm, fm2):
        print("  input grad diff", float((a.grad.float() - b_.grad.float()).abs().max()), "of", float(b_.grad.float().abs().max()))


for bt in (1, 4):
    for dt in (torch.float32, torch.bfloat16):
        run(bt, dt)
