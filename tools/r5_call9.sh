#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_sampler_gpu.py tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "sampler or ddp_in_place" 2>&1 | tail -6 | cut -c1-300
run() { env "$@" timeout 600 python bench.py --steps 60 --warmup 15 --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
echo "== A/B NNDET_SP_TAIL_FUSED" | tee $O/ab_sp_tail.txt
for v in 1 0 1 0 1 0; do echo "SP_TAIL_FUSED=$v $(run NNDET_SP_TAIL_FUSED=$v)" | tee -a $O/ab_sp_tail.txt; done
