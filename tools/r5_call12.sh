#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
timeout 600 python -m pytest tests/test_conv_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x -k "norm_backward_sums and prefetch" 2>&1 | tail -30 | cut -c1-300
for v in 1 0; do
  rm -rf $O/prof
  (cd /tmp && NNDET_NORM_RED_FUSE=$v timeout 900 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof -- python $OLDPWD/bench.py --steps 3 --warmup 2 --no-extras > $OLDPWD/$O/prof.txt 2>&1)
  db=$(find $O/prof -name "*_results.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_stats.py "$db" 5 --by-grid --timeline $O/timeline_fuse$v.txt > $O/kernel_stats_fuse$v.txt
  echo "== FUSE=$v"; grep -i "k_dgs\|k_norm_bwd" $O/kernel_stats_fuse$v.txt | cut -c1-200 | head -30
  rm -rf $O/prof
done
