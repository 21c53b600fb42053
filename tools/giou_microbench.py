#!/usr/bin/env python
"""box_iou and generalized_box_iou [2000 x 100 000] a few times each (for rocprofv3 --pmc: why does GIoU take twice the time for the
same bytes written?). Usage: tools/giou_microbench.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tools.box_microbench import rb
from nndetection_amd.core.boxes import box_iou, generalized_box_iou

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rng = np.random.default_rng(0)
a, g = torch.from_numpy(rb(rng, 100000)).cuda(), torch.from_numpy(rb(rng, 2000)).cuda()
for _ in range(reps):
    box_iou(g, a)
    generalized_box_iou(g, a)
torch.cuda.synchronize()
