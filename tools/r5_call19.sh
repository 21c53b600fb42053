#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
run() { env "$@" timeout 600 python bench.py --steps 60 --warmup 15 --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
echo "== A/B NNDET_PACK_SIDE (0 = all weights packed on the main stream; 1 = L.aux_stream; 2 = weight-gradient stream; 3 = segmentation-branch stream)" | tee $O/ab_pack_side2.txt
run NNDET_PACK_SIDE=0 > /dev/null
for r in 1 2; do
  for v in 0 1 2 3; do echo "PACK_SIDE=$v $(run NNDET_PACK_SIDE=$v)" | tee -a $O/ab_pack_side2.txt; done
done
