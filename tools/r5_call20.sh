#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "stem" 2>&1 | tail -4 | cut -c1-300
run() { env "$@" timeout 600 python bench.py --steps 60 --warmup 15 --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
P=$PWD/nndetection_amd/csrc/libnndet_amd_prev.so
echo "== A/B k_stem_bwd3 LDS 60.8 -> 51 KB (prev = the build before; WGS = total workgroups of the kernel)" | tee $O/ab_stem_lds.txt
run NNDET_AMD_LIB=$P > /dev/null
for r in 1 2 3; do
  echo "prev          $(run NNDET_AMD_LIB=$P)" | tee -a $O/ab_stem_lds.txt
  echo "51 KB         $(run X=1)" | tee -a $O/ab_stem_lds.txt
  echo "51 KB WGS=768 $(run NNDET_STEM_BWD_WGS=768)" | tee -a $O/ab_stem_lds.txt
done
