#!/usr/bin/env python
"""3D NMS at N boxes (default 10 000, IoU 0.6, distinct scores) a few times: the workload of the NMS PMC passes
(`tools/gpu_round.sh nmspmc`: waves, wave cycles, busy cycles -> occupancy of k_nms_mask / k_nms_scan_super, north_star).
Usage: tools/nms_microbench.py [N] [iters]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def main():
    from nndetection_amd.core.boxes import nms
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    rng = np.random.default_rng(0)
    c = rng.uniform(0, 160, (n, 3)); s = rng.uniform(2, 26, (n, 3))
    b = np.stack([c[:, 0] - s[:, 0] / 2, c[:, 1] - s[:, 1] / 2, c[:, 0] + s[:, 0] / 2, c[:, 1] + s[:, 1] / 2,
                  c[:, 2] - s[:, 2] / 2, c[:, 2] + s[:, 2] / 2], 1).astype(np.float32)
    sc = ((rng.permutation(n) + 1) / (n + 1)).astype(np.float32)
    bt, st = torch.from_numpy(b).cuda(), torch.from_numpy(sc).cuda()
    k = nms(bt, st, 0.6); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        k = nms(bt, st, 0.6)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print(f"nms N={n} thr=0.6: {dt * 1e3:.3f} ms, {n / dt / 1e6:.2f} M boxes/s, kept {k.numel()}")


if __name__ == "__main__":
    main()
