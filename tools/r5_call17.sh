#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --deselect tests/test_boxes_gpu.py --deselect tests/test_conv_gpu.py 2>&1 | tail -15 | cut -c1-300 | tee gpurun_out/t_full2.txt
