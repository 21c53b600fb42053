#!/usr/bin/env python
"""Norm backward (k_norm_bwd_reduce + k_norm_bwd_apply) and forward apply of TWO builds of the library on identical inputs at the real
layer shapes: dgamma / dbeta / dx of both against a float64 torch evaluation. Usage: tools/norm_ab_check.py <other lib .so>"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nndetection_amd import _lib as L                                   # noqa: E402

SHAPES = [("enc0 32ch", 4, 160 * 160 * 96, 32, 32), ("enc1 64ch", 4, 80 * 80 * 48, 64, 64), ("enc2 128ch", 4, 40 * 40 * 24, 128, 128),
          ("enc3 256ch", 4, 20 * 20 * 12, 256, 256), ("enc4 320ch", 4, 10 * 10 * 6, 320, 320), ("head P2 gn8", 4, 40 * 40 * 24, 128, 8),
          ("enc0 b1", 1, 160 * 160 * 96, 32, 32), ("enc1 b1", 1, 80 * 80 * 48, 64, 64), ("enc2 b1", 1, 40 * 40 * 24, 128, 128),
          ("enc3 b1", 1, 20 * 20 * 12, 256, 256), ("enc1 b2", 2, 80 * 80 * 48, 64, 64), ("odd 64ch", 3, 77 * 31 * 13, 64, 64),
          ("odd 96ch", 2, 12345, 96, 96), ("lidc e1 b2", 2, 96 * 96 * 64, 64, 64)]


def bind(path):
    lib = C.CDLL(path)
    for name in ("nndet_norm_backward", "nndet_norm_apply"):
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = L.SIGNATURES[name]
    return lib


def main():
    cur, other = L.load(), bind(sys.argv[1])
    dev = torch.device("cuda:0")
    for dt in (torch.bfloat16, torch.float16, torch.float32):
        code = L.dtype_code(torch.empty(0, dtype=dt))
        for name, n, sp, c, groups in SHAPES:
            if dt == torch.float32 and sp > 2e6:
                continue
            if dt != torch.bfloat16 and len(sys.argv) > 2:
                continue
            torch.manual_seed(0)
            cp = (c + 31) // 32 * 32
            y = (torch.randn(n, sp, cp, device=dev) * 1.3 + 0.2).to(dt)
            g = (torch.randn(n, sp, cp, device=dev) * 1e-3).to(dt)
            gam, bet = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.3
            stats = torch.zeros(L.STATS_REPLICAS, n, cp, 2, dtype=torch.float64, device=dev)
            yd = y.double()
            stats[0, :, :, 0], stats[0, :, :, 1] = yd.sum(1), (yd * yd).sum(1)
            res = {}
            for tag, lib in (("cur", cur), ("other", other)):
                out, mr = torch.empty_like(y), torch.empty(n, cp, 2, device=dev)
                rc = lib.nndet_norm_apply(code, L.ptr(y), L.ptr(stats), L.ptr(gam), L.ptr(bet), n, sp, c, cp, groups, 1e-5, 1, L.ptr(out), L.ptr(mr), L.stream())
                assert rc == 0
                dx, dg, db = torch.empty_like(y), torch.zeros(c, device=dev), torch.zeros(c, device=dev)
                red = torch.zeros(L.STATS_REPLICAS * n * cp * 2 + n, dtype=torch.float64, device=dev)
                rc = lib.nndet_norm_backward(code, L.ptr(y), L.ptr(g), L.ptr(mr), L.ptr(gam), L.ptr(bet), n, sp, c, cp, groups, 1, L.ptr(dx), L.ptr(dg),
                                             L.ptr(db), L.ptr(red), L.stream())
                assert rc == 0
                torch.cuda.synchronize()
                res[tag] = (out, mr, dx, dg, db)
            # float64 reference of dgamma / dbeta from the library's own (mean, rstd)
            mr = res["cur"][1].double()
            mu, rs = mr[:, :c, 0].unsqueeze(1), mr[:, :c, 1].unsqueeze(1)
            xh = (yd[..., :c] - mu) * rs
            z = xh * gam.double() + bet.double()
            gm = g.double()[..., :c] * (z > 0)
            db64, dg64 = gm.sum((0, 1)), (gm * xh).sum((0, 1))
            line = f"{str(dt)[6:]:9s} {name:12s}"
            for tag in ("cur", "other"):
                out, _, dx, dg, db = res[tag]
                line += (f" | {tag}: dgamma err {float((dg.double() - dg64).abs().max() / dg64.abs().max()):.2e} dbeta err "
                         f"{float((db.double() - db64).abs().max() / db64.abs().max()):.2e}")
            m_ = float(sp * (c // groups))
            xg = xh.reshape(n, sp, groups, c // groups); gg = (gm * gam.double()).reshape(n, sp, groups, c // groups)
            k1 = gg.sum((1, 3), keepdim=True) / m_; k2 = (gg * xg).sum((1, 3), keepdim=True) / m_
            dx64 = (rs.reshape(n, 1, groups, c // groups) * (gg - k1 - xg * k2)).reshape(n, sp, c)
            for tag in ("cur", "other"):
                e = (res[tag][2].double()[..., :c] - dx64).abs().max() / dx64.abs().max()
                line += f" | {tag} dx err {float(e):.2e}"
            same = [bool(torch.equal(a, b)) for a, b in zip(res["cur"], res["other"])]
            dxd = float((res["cur"][2].double() - res["other"][2].double()).abs().max())
            line += f" | identical out/mr/dx/dg/db: {same} max |dx diff| {dxd:.2e}"
            print(line, flush=True)
            del y, g, yd, xh, z, gm, res


if __name__ == "__main__":
    main()
