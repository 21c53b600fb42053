#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
echo "== new tests"
timeout 900 python -m pytest tests/test_pyramid_gpu.py tests/test_parity_full_gpu.py tests/test_abi.py -m gpu -q --tb=short -p no:cacheprovider -k "fused_first or head_forward_uses or norm_backward_inside or header or lidc192_b2" 2>&1 | tail -25 | cut -c1-300
run() { env "$@" timeout 600 python bench.py --steps 60 --warmup 15 --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
echo "== A/B NNDET_HEAD_FUSE_CIN" | tee $O/ab_fuse_cin.txt
for v in 1 0 1 0 1 0; do echo "FUSE_CIN=$v $(run NNDET_HEAD_FUSE_CIN=$v)" | tee -a $O/ab_fuse_cin.txt; done
