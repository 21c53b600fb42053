"""GPU: two-build differential of the normalisation kernels. Round 5 found a hipcc (ROCm 7.2) miscompile of
k_norm_bwd_reduce that no unit test saw (profiles/round5_norm_reduce_miscompile.txt); besides the in-step re-evaluations
(tests/test_parity_full_gpu.py) this test compiles csrc/norm.hip a second time at -O1 on the GPU box, links it with the other objects of
the shipped library and runs forward apply + backward of BOTH builds on identical inputs at real layer shapes (InstanceNorm, GroupNorm,
odd sizes, all three storage types): element-wise outputs must be bit-identical, the reduced sums equal to summation-order noise, and
both within 1e-5 of a float64 evaluation. A compiler upgrade that breaks one optimisation level shows up as a difference."""
import ctypes as C
import glob
import os
import shutil
import subprocess

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nndetection_amd", "csrc")

SHAPES = [("enc1 64ch", 2, 80 * 80 * 48, 64, 64), ("enc2 128ch", 4, 40 * 40 * 24, 128, 128), ("enc4 320ch", 4, 10 * 10 * 6, 320, 320),
          ("head P2 gn8", 4, 40 * 40 * 24, 128, 8), ("odd 64ch", 3, 77 * 31 * 13, 64, 64), ("odd 96ch gn6", 2, 12345, 96, 6)]


@pytest.fixture(scope="module")
def o1_lib(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    objs = [o for o in glob.glob(os.path.join(CSRC, "_obj", "*.o")) if os.path.basename(o) != "norm.o"]
    if not os.path.isfile(hipcc) or len(objs) < 15:
        pytest.skip("needs hipcc and the object files of the shipped library (csrc/_obj)")
    d = tmp_path_factory.mktemp("norm_o1")
    obj, lib = str(d / "norm_o1.o"), str(d / "libnndet_amd_norm_o1.so")
    flags = ["--offload-arch=gfx950", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"]
    subprocess.check_call([hipcc, "-O1"] + flags + ["-c", os.path.join(CSRC, "norm.hip"), "-o", obj])
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--no-undefined", "-o", lib, obj] + objs)
    return lib


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32], ids=["bf16", "f16", "f32"])
def test_norm_apply_and_backward_O1_vs_O3_builds(o1_lib, dtype):
    from nndetection_amd import _lib as L
    cur = L.load()
    other = C.CDLL(o1_lib)
    for name in ("nndet_norm_backward", "nndet_norm_apply"):
        fn = getattr(other, name)
        fn.restype, fn.argtypes = L.SIGNATURES[name]
    dev = torch.device("cuda:0")
    code = L.dtype_code(torch.empty(0, dtype=dtype))
    for name, n, sp, c, groups in SHAPES:
        torch.manual_seed(0)
        cp = (c + 31) // 32 * 32
        y = (torch.randn(n, sp, cp, device=dev) * 1.3 + 0.2).to(dtype)
        y[..., c:] = 0
        g = (torch.randn(n, sp, cp, device=dev) * 1e-3).to(dtype)
        gam, bet = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.3
        yd = y.double()
        stats = torch.zeros(L.STATS_REPLICAS, n, cp, 2, dtype=torch.float64, device=dev)
        stats[0, :, :, 0], stats[0, :, :, 1] = yd.sum(1), (yd * yd).sum(1)
        res = {}
        for tag, lib in (("O3", cur), ("O1", other)):
            out, mr = torch.empty_like(y), torch.empty(n, cp, 2, device=dev)
            assert lib.nndet_norm_apply(code, L.ptr(y), L.ptr(stats), L.ptr(gam), L.ptr(bet), n, sp, c, cp, groups, 1e-5, 1, L.ptr(out), L.ptr(mr), L.stream()) == 0
            dx, dg, db = torch.empty_like(y), torch.zeros(c, device=dev), torch.zeros(c, device=dev)
            red = torch.zeros(L.STATS_REPLICAS * n * cp * 2 + n, dtype=torch.float64, device=dev)
            assert lib.nndet_norm_backward(code, L.ptr(y), L.ptr(g), L.ptr(mr), L.ptr(gam), L.ptr(bet), n, sp, c, cp, groups, 1, L.ptr(dx), L.ptr(dg),
                                           L.ptr(db), L.ptr(red), L.stream()) == 0
            torch.cuda.synchronize()
            res[tag] = (out, mr, dx, dg, db)
        a, b = res["O3"], res["O1"]
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]), f"{name}: forward apply / (mean, rstd) differ between the builds"
        # float64 reference from the library's own (mean, rstd)
        mr = a[1].double()
        mu, rs = mr[:, :c, 0].unsqueeze(1), mr[:, :c, 1].unsqueeze(1)
        xh = (yd[..., :c] - mu) * rs
        sc = (a[1][:, :c, 1] * gam).unsqueeze(1); sh = bet - a[1][:, :c, 0].unsqueeze(1) * sc
        mask = torch.addcmul(sh, y[..., :c].float(), sc) > 0
        gm = g.double()[..., :c] * mask
        db64, dg64 = gm.sum((0, 1)), (gm * xh).sum((0, 1))
        for tag in ("O3", "O1"):
            _, _, dx, dg, db = res[tag]
            assert float((dg.double() - dg64).abs().max() / dg64.abs().max()) <= 1e-5, (name, tag, "dgamma")
            assert float((db.double() - db64).abs().max() / db64.abs().max()) <= 1e-5, (name, tag, "dbeta")
        # the two builds against each other: sums to summation-order noise, dx to one rounding of the storage type (the per-group constants
        # come from atomically accumulated sums)
        assert float((a[3] - b[3]).abs().max() / a[3].abs().max()) <= 2e-6 and float((a[4] - b[4]).abs().max() / a[4].abs().max()) <= 2e-6, name
        ulp = {torch.bfloat16: 2.0 ** -8, torch.float16: 2.0 ** -11, torch.float32: 2.0 ** -22}[dtype]
        dmax = float(a[2].float().abs().max())
        assert float((a[2].float() - b[2].float()).abs().max()) <= 2 * ulp * dmax, f"{name}: dx differs between the builds"
        del y, g, yd, xh, gm, res, a, b
