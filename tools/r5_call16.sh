#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -15 | cut -c1-300 | tee gpurun_out/t_full.txt
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 | tee gpurun_out/bench_v3.json | cut -c1-600
