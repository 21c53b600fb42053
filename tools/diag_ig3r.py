"""Diagnostic: k_ig3r against k_ig3 (bit-identical by construction) on shapes with ragged tile counts per workgroup."""
import os, sys
import torch
from nndetection_amd.arch.conv import ConvInstanceRelu


def run(shape, grid=None, bias=True):
    N, D, H, W = shape
    torch.manual_seed(1)
    m = ConvInstanceRelu(3, 32, 32, 3, stride=1, padding=1, add_norm=False, add_act=False).cuda()
    with torch.no_grad():
        m.conv.weight.copy_(torch.randn_like(m.conv.weight) / 29.4)
        m.conv.bias.copy_(torch.randn_like(m.conv.bias) * 0.3)
    x = torch.randn(N, 32, D, H, W, device="cuda").to(torch.bfloat16)
    outs = []
    for mode in ("2", "0"):
        os.environ["NNDET_IG3R"] = mode
        if grid and mode == "2":
            os.environ["NNDET_IG3R_GRID"] = str(grid)
        else:
            os.environ.pop("NNDET_IG3R_GRID", None)
        with torch.no_grad():
            y = m(x)
        torch.cuda.synchronize()
        outs.append(y.permute(0, 2, 3, 4, 1).float())
    a, b = outs
    bad = (a != b) | ~torch.isfinite(a)
    nb = int(bad.any(dim=-1).sum())
    msg = f"shape {shape} grid {grid}: {nb} bad voxels of {N * D * H * W}, non-finite {int((~torch.isfinite(a)).sum())}"
    if nb:
        idx = bad.any(dim=-1).nonzero()
        tiles = torch.unique(torch.stack([idx[:, 0], idx[:, 1] // 8, idx[:, 2] // 8, idx[:, 3] // 8], 1), dim=0)
        nt = (D // 8) * (H // 8) * (W // 8)
        lin = (tiles[:, 0] * nt + (tiles[:, 1] * (H // 8) + tiles[:, 2]) * (W // 8) + tiles[:, 3]).tolist()
        msg += f"; {len(lin)} bad tiles, first {lin[:24]} last {lin[-6:]}; d-planes in tile of first bad voxels {sorted(set((idx[:200, 1] % 8).tolist()))}"
    print(msg, flush=True)
    if nb and D >= 160:
        t0 = tiles[0]
        sel = (idx[:, 0] == t0[0]) & (idx[:, 1] // 8 == t0[1]) & (idx[:, 2] // 8 == t0[2]) & (idx[:, 3] // 8 == t0[3])
        vox = idx[sel]
        print("   tile", t0.tolist(), "bad voxels (d,h,w in tile):", [(int(v[1]) % 8, int(v[2]) % 8, int(v[3]) % 8) for v in vox][:40])
        for v in vox[:3]:
            va, vb = a[v[0], v[1], v[2], v[3]], b[v[0], v[1], v[2], v[3]]
            ch = (va != vb).nonzero().flatten().tolist()
            print("   voxel", v.tolist(), "bad channels", ch, "ig3r", [round(float(z), 3) for z in va[ch][:8]], "ig3", [round(float(z), 3) for z in vb[ch][:8]])
        # does the wrong value equal the right value of another position? (misdirected store)
        v = vox[0]
        va = a[v[0], v[1], v[2], v[3]]
        tile_b = b[v[0], (v[1] // 8) * 8:(v[1] // 8) * 8 + 8, (v[2] // 8) * 8:(v[2] // 8) * 8 + 8, (v[3] // 8) * 8:(v[3] // 8) * 8 + 8]
        m = (tile_b[..., :8] == va[:8]).all(-1).nonzero()
        print("   first bad voxel's channels 0-7 equal the reference of in-tile positions:", m.tolist()[:5])


run((1, 24, 24, 40), 16)
run((1, 24, 24, 40), 8)
run((1, 24, 24, 40), None)
run((2, 24, 24, 40), 16)
run((2, 16, 24, 32), 16)
run((1, 160, 160, 96))
run((1, 80, 80, 96))
run((1, 160, 160, 48))
