#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
run() { env "$@" timeout 600 python bench.py --steps 60 --warmup 15 --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
echo "== A/B R128 x FUSE_CIN" | tee $O/ab_r128.txt
for rep in 1 2; do
for cfg in "NNDET_HEAD_FUSE_CIN=0" "NNDET_HEAD_FUSE_CIN=1" "NNDET_HEAD_FUSE_CIN=1 NNDET_IGEMM_R128=1" "NNDET_HEAD_FUSE_CIN=0 NNDET_IGEMM_R128=1"; do echo "$cfg $(run $cfg)" | tee -a $O/ab_r128.txt; done; done
