#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "s2 or strided" 2>&1 | tail -3 | cut -c1-300
P=$PWD/nndetection_amd/csrc/libnndet_amd_prev.so
echo "== conv_microbench e2_64to128_s2 (forward = k_ig3s2): prev lib, then current" | tee $O/ig3s2_swz.txt
NNDET_AMD_LIB=$P timeout 300 python tools/conv_microbench.py e2_64to128_s2 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/ig3s2_swz.txt
timeout 300 python tools/conv_microbench.py e2_64to128_s2 2>&1 | grep -v amdgpu.ids | tail -1 | tee -a $O/ig3s2_swz.txt
timeout 900 python -m pytest tests/test_parity_full_gpu.py tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "luna160 or toy64 or tiny" 2>&1 | tail -3 | cut -c1-300
run() { env "$@" timeout 600 python bench.py --steps 80 --warmup 15 --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
echo "== A/B (prev = HEAD before the norm shuffle, k_stem_fwd3, k_ig3s, k_ig3s2 changes)" | tee $O/ab_fwd_chain.txt
run NNDET_AMD_LIB=$P > /dev/null
for r in 1 2 3 4; do
  echo "prev $(run NNDET_AMD_LIB=$P)" | tee -a $O/ab_fwd_chain.txt
  echo "cur  $(run X=1)" | tee -a $O/ab_fwd_chain.txt
done
