"""Isolated timing of the norm backward passes (k_norm_bwd_reduce + k_norm_bwd_apply) and the forward passes (stats + apply) at the
shapes of the luna160 step, with the algorithmic traffic next to them: reduce reads x and dy, apply reads both and writes dx.
    python tools/norm_microbench.py [dtype]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from nndetection_amd import _lib as L                                   # noqa: E402

SHAPES = [  # (name, batch, spatial, channels, groups)
    ("enc0 32ch 160x160x96", 4, 160 * 160 * 96, 32, 32),
    ("enc1 64ch 80x80x96", 4, 80 * 80 * 96, 64, 64),
    ("enc2 128ch 40x40x48", 4, 40 * 40 * 48, 128, 128),
    ("enc3 256ch 20x20x24", 4, 20 * 20 * 24, 256, 256),
    ("enc4 320ch 10x10x12", 4, 10 * 10 * 12, 320, 320),
    ("enc5 320ch 5x5x6", 4, 5 * 5 * 6, 320, 320),
    ("head P2 128ch gn8 40x40x24", 4, 40 * 40 * 24, 128, 8),
]


def main():
    dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
    code = L.dtype_code(torch.empty(0, dtype=dt))
    dev = torch.device("cuda:0")
    for name, n, sp, c, groups in SHAPES:
        cp = (c + 31) // 32 * 32
        x = torch.randn(n, sp, cp, device=dev).to(dt)
        dy = torch.randn(n, sp, cp, device=dev).to(dt)
        dx = torch.empty_like(x)
        g = torch.rand(c, device=dev) + 0.5
        b = torch.randn(c, device=dev)
        mr = torch.zeros(n, cp, 2, device=dev)
        mr[..., 1] = 1.0
        dg, db = torch.zeros(c, device=dev), torch.zeros(c, device=dev)
        nred = L.STATS_REPLICAS * n * cp * 2 + n

        def run():
            red = torch.zeros(nred, dtype=torch.float64, device=dev)
            L.call("nndet_norm_backward", code, L.ptr(x), L.ptr(dy), L.ptr(mr), L.ptr(g), L.ptr(b), n, sp, c, cp, groups, 1,
                   L.ptr(dx), L.ptr(dg), L.ptr(db), L.ptr(red), L.stream())

        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        nbytes = x.numel() * x.element_size()
        print(f"{name:32s} tensor {nbytes / 1e6:7.1f} MB | backward (reduce + apply) {ms:7.3f} ms = {5 * nbytes / ms / 1e9:6.2f} TB/s of 5 passes",
              flush=True)


if __name__ == "__main__":
    main()
