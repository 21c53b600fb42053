#!/usr/bin/env python
"""box-op micro-benchmark on one MI355X through the C ABI wrappers, with the numpy oracle timed on the
host for a bounded sample. Prints one line per op: size, ms, rate, algorithmic GB/s (DESIGN.md section 4 byte counts).
Usage: tools/box_microbench.py [--no-cpu]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def rb(rng, n, extent=160.0, smin=2.0, smax=26.0):
    c = rng.uniform(0, extent, (n, 3)); s = rng.uniform(smin, smax, (n, 3))
    return np.stack([c[:, 0] - s[:, 0] / 2, c[:, 1] - s[:, 1] / 2, c[:, 0] + s[:, 0] / 2, c[:, 1] + s[:, 1] / 2,
                     c[:, 2] - s[:, 2] / 2, c[:, 2] + s[:, 2] / 2], 1).astype(np.float32)


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


def main():
    from nndetection_amd.core.boxes import box_iou, generalized_box_iou, nms, ATSSMatcher
    from nndetection_amd.plans import get_plan
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    a_np, g_np = rb(rng, 100000), rb(rng, 2000)
    a, g = torch.from_numpy(a_np).to(dev), torch.from_numpy(g_np).to(dev)
    for name, fn in (("box_iou", box_iou), ("generalized_box_iou", generalized_box_iou)):
        dt = timeit(lambda: fn(g, a))
        pairs = g.shape[0] * a.shape[0]
        print(f"{name:22s} [2000 x 100000]  {dt * 1e3:8.3f} ms  {pairs / dt / 1e9:8.2f} G pairs/s  {(4 * pairs + 24 * (g.shape[0] + a.shape[0])) / dt / 1e9:8.1f} GB/s (writes)", flush=True)
    for n in (1000, 2000, 10000, 100000):
        b = torch.from_numpy(rb(rng, n)).to(dev)
        sc = torch.from_numpy(((rng.permutation(n) + 1) / (n + 1)).astype(np.float32)).to(dev)
        dt = timeit(lambda: nms(b, sc, 0.6))
        k = nms(b, sc, 0.6)
        print(f"{'nms thr 0.6':22s} N = {n:6d}        {dt * 1e3:8.3f} ms  {n / dt / 1e6:8.2f} M boxes/s  kept {k.numel()}", flush=True)
    # ATSS on the config-2 anchor set (1 186 650 anchors per image, 4 levels), 3 / 20 GT boxes, batch of 4 in one pass
    plan = get_plan("luna160")
    from nndetection_amd.ptmodule import build_model
    net = build_model(plan).to(dev)
    x = torch.zeros(1, 1, *plan["patch_size"], device=dev, dtype=torch.bfloat16)
    with torch.no_grad():
        fm = net.decoder(net.encoder(x))
        fm_head = [fm[i] for i in net.decoder_levels]
        anchors = net.anchor_generator(x, fm_head)[0]
    npl = net.anchor_generator.get_num_acnhors_per_level()
    m = ATSSMatcher(num_candidates=4, center_in_gt=False)
    for G in (3, 20):
        gts = [torch.from_numpy(rb(rng, G, smin=4, smax=24)).to(dev) for _ in range(4)]
        dt1 = timeit(lambda: m(gts[0], anchors, npl, 27))
        dt4 = timeit(lambda: m.match_batch(gts, anchors, npl, 27))
        print(f"{'atss match':22s} M = {anchors.shape[0]}, G = {G:2d}: 1 image {dt1 * 1e3:7.3f} ms; batch of 4 in one pass {dt4 * 1e3:7.3f} ms "
              f"({4 * anchors.shape[0] / dt4 / 1e9:.2f} G anchors/s)", flush=True)
    # SURVEY 8d config 5, ATSS leg: 2 000 GT x 5 levels x 100 000 anchors, 27 anchors per location, 4 candidates (k = 108 per level)
    rng5 = np.random.default_rng(0)
    a5_np = np.concatenate([rb(rng5, 100000) for _ in range(5)], 0); g5_np = rb(rng5, 2000)
    a5, g5 = torch.from_numpy(a5_np).to(dev), torch.from_numpy(g5_np).to(dev)
    npl5 = [100000] * 5
    for G in (40, 500, 2000):
        dt = timeit(lambda: m(g5[:G], a5, npl5, 27), iters=3)
        print(f"{'atss match (config 5)':22s} M = 5 x 100000, G = {G:4d}: {dt * 1e3:8.3f} ms  {G * a5.shape[0] / dt / 1e9:8.2f} G (GT, anchor) pairs/s  "
              f"{a5.shape[0] / dt / 1e6:8.2f} M anchors/s", flush=True)
    if "--no-cpu" not in sys.argv:
        from oracle import boxes_np as bx
        th = min(16, os.cpu_count() or 1)
        t0 = time.perf_counter(); ref5 = bx.atss_match_blocked(g5_np, a5_np, npl5, 27, 4, rows=50, threads=th); dt = time.perf_counter() - t0
        got5 = m(g5, a5, npl5, 27)[1].cpu().numpy()
        print(f"cpu oracle atss        M = 5 x 100000, G = 2000: {dt * 1e3:8.1f} ms (numpy, blocked, {th} threads); GPU matches bit-exact: "
              f"{bool(np.array_equal(ref5, got5))}, positives {int((ref5 >= 0).sum())}", flush=True)
    if "--no-cpu" not in sys.argv:
        from oracle import boxes_np as bx
        t0 = time.perf_counter(); bx.box_iou(g_np[:200], a_np); dt = time.perf_counter() - t0
        print(f"cpu oracle box_iou     [200 x 100000]   {dt * 1e3:8.1f} ms  {200 * 100000 / dt / 1e9:8.3f} G pairs/s (numpy, host)", flush=True)
        b = rb(rng, 10000); sc = ((rng.permutation(10000) + 1) / 10001).astype(np.float32)
        t0 = time.perf_counter(); bx.nms(b, sc, 0.6); dt = time.perf_counter() - t0
        print(f"cpu oracle nms         N = 10000        {dt * 1e3:8.1f} ms  {10000 / dt / 1e6:8.4f} M boxes/s (numpy, host)", flush=True)


if __name__ == "__main__":
    main()
