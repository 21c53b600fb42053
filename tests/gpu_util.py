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


det_randperm.nndet_reversed_arange = True      # the device sampler (core/boxes/sampler.py) then selects what this permutation selects


def relerr(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / max(1e-30, b.abs().max()))


def synth_inputs(plan, seed=0):
    """Deterministic synthetic batch of a plan (CPU tensors): image ~ N(0,1) from a seeded generator, 1-2 boxes per image with
    a consistent instance segmentation. Shared by tests/golden/make_golden.py and the parity tests."""
    import numpy as np
    import torch
    P, B = plan["patch_size"], plan["batch_size"]
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 1, *P, generator=g)
    boxes, classes = [], []
    seg = torch.zeros(B, *P)
    rng = np.random.default_rng(seed + 1)
    for b in range(B):
        n = 2 if b % 2 == 0 else 1
        c = rng.uniform(0.25, 0.75, (n, 3)) * np.asarray(P) + 0.137
        s = rng.uniform(5, 11, (n, 3))
        lo, hi = np.maximum(c - s / 2, 0), np.minimum(c + s / 2, np.asarray(P))
        bb = np.stack([lo[:, 0], lo[:, 1], hi[:, 0], hi[:, 1], lo[:, 2], hi[:, 2]], 1).astype(np.float32)
        boxes.append(torch.from_numpy(bb)); classes.append(torch.zeros(n))
        for q in bb:
            seg[b, int(q[0]):int(q[2]) + 1, int(q[1]):int(q[3]) + 1, int(q[4]):int(q[5]) + 1] = 1
    return x, {"target_boxes": boxes, "target_classes": classes, "target_seg": seg}
