#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 900 python -m pytest tests/test_conv_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "stem or rank1 or segmentation or conv_fwd_bwd" 2>&1 | tail -4 | cut -c1-300
: > $O/stem_fwd_variants.txt
for w in 0 1024 2048; do
  rm -rf $O/prof
  (cd /tmp && NNDET_STEM_FWD_WGS=$w timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -- python $OLDPWD/tools/stem_bwd_microbench.py 6 > $OLDPWD/$O/prof.txt 2>&1)
  db=$(find $O/prof -name "*_results.db" | head -1)
  echo "== FWD_WGS=$w" | tee -a $O/stem_fwd_variants.txt
  [ -n "$db" ] && python tools/rocpd_stats.py "$db" 40 | grep -i "k_stem_fwd3" | cut -c1-150 | tee -a $O/stem_fwd_variants.txt
  rm -rf $O/prof
done
timeout 900 python -m pytest tests/test_parity_full_gpu.py tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "luna160 or toy64 or tiny" 2>&1 | tail -4 | cut -c1-300
run() { env "$@" timeout 600 python bench.py --steps 60 --warmup 15 --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
P=$PWD/nndetection_amd/csrc/libnndet_amd_prev.so
echo "== A/B k_stem_fwd3 recipe (prev = HEAD before the norm-shuffle and this)" | tee $O/ab_stem_fwd.txt
run NNDET_AMD_LIB=$P > /dev/null
for r in 1 2 3; do
  echo "prev     $(run NNDET_AMD_LIB=$P)" | tee -a $O/ab_stem_fwd.txt
  echo "cur      $(run X=1)" | tee -a $O/ab_stem_fwd.txt
  echo "cur 1024 $(run NNDET_STEM_FWD_WGS=1024)" | tee -a $O/ab_stem_fwd.txt
done
