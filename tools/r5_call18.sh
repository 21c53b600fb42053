#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -6 | cut -c1-300
run() { env "$@" timeout 600 python bench.py --steps 60 --warmup 15 --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
echo "== A/B NNDET_PACK_SIDE (1 = weights behind encoder stage 0 packed on the auxiliary stream)" | tee $O/ab_pack_side.txt
for r in 1 2 3; do
  echo "PACK_SIDE=0 $(run NNDET_PACK_SIDE=0)" | tee -a $O/ab_pack_side.txt
  echo "PACK_SIDE=1 $(run NNDET_PACK_SIDE=1)" | tee -a $O/ab_pack_side.txt
done
