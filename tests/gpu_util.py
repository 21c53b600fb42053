import numpy as np
import torch


def rand_boxes(rng, n, extent=(160, 160, 96), smin=2, smax=26):
    c = rng.uniform(0, 1, (n, 3)) * np.asarray(extent)
    s = rng.uniform(smin, smax, (n, 3))
    lo, hi = c - s / 2, c + s / 2
    return np.stack([lo[:, 0], lo[:, 1], hi[:, 0], hi[:, 1], lo[:, 2], hi[:, 2]], 1).astype(np.float32)


def distinct_scores(rng, n):
    return ((rng.permutation(n).astype(np.float64) + 1) / (n + 1)).astype(np.float32)


def dev():
    return torch.device("cuda:0")


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def det_randperm(n, *a, **k):
    return torch.arange(n - 1, -1, -1, device=k.get("device", None))


def relerr(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / max(1e-30, b.abs().max()))
