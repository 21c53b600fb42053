#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_conv_gpu.py tests/test_pyramid_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "norm or block or trunk or items" 2>&1 | tail -4 | cut -c1-300
timeout 900 python -m pytest tests/test_parity_full_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "norm_backward_inside or luna160_b4_fp32 or luna160_absorbed" 2>&1 | tail -4 | cut -c1-300
run() { env "$@" timeout 600 python bench.py --steps 60 --warmup 15 --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
P=$PWD/nndetection_amd/csrc/libnndet_amd_prev.so
echo "== A/B norm reduce epilogue: xor shuffles before the LDS atomics (prev = HEAD before)" | tee $O/ab_norm_shfl.txt
run NNDET_AMD_LIB=$P > /dev/null
for r in 1 2 3; do
  echo "prev $(run NNDET_AMD_LIB=$P)" | tee -a $O/ab_norm_shfl.txt
  echo "cur  $(run X=1)" | tee -a $O/ab_norm_shfl.txt
done
