#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_sampler_gpu.py tests/test_model_gpu.py tests/test_parity_full_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "sampler or tiny_fp32 or luna160_b4_fp32" 2>&1 | tail -6 | cut -c1-300
run() { env "$@" timeout 600 python bench.py --steps 60 --warmup 15 --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
echo "== A/B NNDET_SP_FUSED_PICK" | tee $O/ab_sp_pick.txt
for v in 1 0 1 0 1 0; do echo "SP_FUSED_PICK=$v $(run NNDET_SP_FUSED_PICK=$v)" | tee -a $O/ab_sp_pick.txt; done
