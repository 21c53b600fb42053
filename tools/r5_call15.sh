#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "norm_backward_sums" 2>&1 | tail -8 | cut -c1-250
for v in 3 4; do NNDET_DGS_NB_LATE=$v timeout 900 python -m pytest tests/test_parity_full_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "norm_backward_inside" 2>&1 | tail -3 | cut -c1-400; done
: > $O/dgs_variants.txt
for cfg in "0 0" "1 0" "1 3" "1 4"; do
  set -- $cfg
  rm -rf $O/prof
  (cd /tmp && NNDET_NORM_RED_FUSE=$1 NNDET_DGS_NB_LATE=$2 timeout 300 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -- python $OLDPWD/tools/dgs_normred_microbench.py 6 > $OLDPWD/$O/prof.txt 2>&1)
  db=$(find $O/prof -name "*_results.db" | head -1)
  echo "== FUSE=$1 variant=$2  $(grep 'fused norm' $O/prof.txt | cut -c1-150)" | tee -a $O/dgs_variants.txt
  [ -n "$db" ] && python tools/rocpd_stats.py "$db" 40 | grep -i "k_dgs\|k_norm_bwd" | cut -c1-150 | tee -a $O/dgs_variants.txt
  rm -rf $O/prof
done
run() { env "$@" timeout 600 python bench.py --steps 60 --warmup 15 --no-extras 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | head -1; }
echo "== step A/B" | tee $O/ab_norm_red_fuse2.txt
for r in 1 2; do
  echo "FUSE=0       $(run NNDET_NORM_RED_FUSE=0)" | tee -a $O/ab_norm_red_fuse2.txt
  for v in 0 3 4; do echo "FUSE=1 var=$v $(run NNDET_NORM_RED_FUSE=1 NNDET_DGS_NB_LATE=$v)" | tee -a $O/ab_norm_red_fuse2.txt; done
done
