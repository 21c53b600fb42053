#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
run() { env "$@" timeout 600 python bench.py --steps 60 --warmup 15 --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
echo "== A/B library cur (norm reductions, relu template) vs prev (round 4)" | tee $O/ab_normlib.txt
for v in cur prev cur prev; do lib=$PWD/nndetection_amd/csrc/libnndet_amd.so; [ $v = prev ] && lib=$PWD/nndetection_amd/csrc/libnndet_amd_prev.so
  echo "lib=$v $(run NNDET_AMD_LIB=$lib)" | tee -a $O/ab_normlib.txt; done
echo "== suite"
timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/t_suite.txt 2>&1; tail -12 $O/t_suite.txt; grep -n "^E  " $O/t_suite.txt | cut -c1-300 | head -20
