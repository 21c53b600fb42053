#!/usr/bin/env python
"""Does the 256 MiB Infinity Cache keep a tensor a kernel has just WRITTEN, so that the next kernel reads it without HBM traffic?
Producer (copy kernel, writes `size` MB) -> consumer (elementwise kernel reading it, writing `size` MB), HIP-event timed:
  hot      consumer right after the producer, same traversal order
  hot-rev  consumer walks the tensor in REVERSE block order (most recently written data first) -- emulated by flipping halves
  cold     1.5 GB of unrelated traffic between producer and consumer
Decides whether per-image chaining of the full-resolution layers (157 MB per image and tensor at 160x160x96x32 bf16, 629 MB per
batch of 4) is worth building."""
import torch

dev = torch.device("cuda")
flush = torch.empty(1536 << 20, dtype=torch.uint8, device=dev)


def run(mb, mode, iters=10):
    n = (mb << 20) // 2
    src = torch.randn(n, device=dev, dtype=torch.bfloat16)
    y = torch.empty_like(src)
    z = torch.empty_like(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    h = n // 2
    for _ in range(iters + 2):
        y.copy_(src)                                   # producer: writes y front to back
        if mode == "cold":
            flush.fill_(1)
        e0.record()
        if mode == "hot-rev":                          # second half first: the part of y written last
            torch.mul(y[h:], 2, out=z[h:]); torch.mul(y[:h], 2, out=z[:h])
        else:
            torch.mul(y, 2, out=z)
        e1.record()
        torch.cuda.synchronize()
        if _ >= 2:
            tot += e0.elapsed_time(e1)
    ms = tot / iters
    return ms, 2 * mb / 1024 / (ms * 1e-3) / 1e3       # TB/s of (read + write)


for mb in (64, 157, 314, 629):
    res = {m: run(mb, m) for m in ("hot", "hot-rev", "cold")}
    print(f"{mb:4d} MB tensor: " + "  ".join(f"{m} {v[0]:.3f} ms ({v[1]:.2f} TB/s r+w)" for m, v in res.items()), flush=True)
